"""CPU-side check of the BUILT library's kernels (no GPU, no hipcc): the resource table hipcc wrote into the code objects.

Every kernel the shared object carries can be reached by some dispatcher path (instantiations the dispatchers cannot pick are not
compiled: csrc/pc_gemm_skinny.h ``launch_w8`` / ``launch_one``, csrc/pc_gemm_rows.hip), so "no reachable kernel spills" is
checked as "no kernel of the library has a private (scratch) segment"."""
import os

from promptcache_amd import _native, codeobj


def _rows():
    assert os.path.exists(_native.lib_path()), "run __graft_entry__.build() first"
    return codeobj.kernels(_native.lib_path())


def test_no_kernel_of_the_library_uses_scratch():
    rows = _rows()
    assert len(rows) > 100, "the code-object metadata was not found in the library"
    spilling = sorted(((r["scratch"], r["vgpr"], r["demangled"]) for r in rows if r["scratch"] > 0), reverse=True)
    assert not spilling, "kernels with a scratch segment (bytes per lane, VGPRs, name):\n" + "\n".join(map(str, spilling[:40]))


def test_kernel_count_and_the_hot_path_kernels_are_there():
    rows = _rows()
    names = [r["demangled"] for r in rows]
    # round 2 shipped 748 instantiations of the weight-streaming template alone
    # (round 5: + 8 gemm_dense_kernel<.., LO8> and 4 quant_rows_kernel instantiations; + 70 gemm_q8p_kernel, 3 gemm_q8f_kernel and 4 gemm_q8c_kernel:
    # the LLM.int8 projections with the quantiser inside, csrc/pc_gemm_q8.hip)
    assert len(rows) <= 725, len(rows)
    for must in ("kv_copy_kernel", "attn_small_kernel<128, false, 0, false, 1>", "attn_small_kernel<128, false, 0, true, 1>", "attn_small_kernel<128, false, 0, true, 2>", "kv_row_table_kernel", "pca::attn_ring_kernel<true, false, false>", "pca::attn_ring_kernel<true, false, true>", "gemm_skinny_ks_kernel", "gemm_q8p_kernel<3, 2, 1, 2, 2>", "gemm_q8p_kernel<3, 3, 1, 2, 2>", "gemm_q8p_kernel<1, 1, 0, 2, 2>", "gemm_q8p_kernel<1, 1, 2, 2, 2>", "gemm_q8p_kernel<3, 2, 1, 2, 8>", "gemm_q8f_kernel<4, 2>", "gemm_q8c_kernel<1, 11>", "gemm_part_kernel", "gemm_q8p_kernel<3, 2, 3, 2, 2>", "gemm_q8p_kernel<1, 1, 3, 2, 2>",
                 "pcg::gemm_rows_kernel<4, 3, 2, true,", "pcg::gemm_rows_kernel<8, 2, 2, true, 3, 3, 2, 768", "gemm_dense_kernel<2, 2, true, false, false>", "rope_append_kernel", "pca::attn_wide_kernel<true, 3>", "pca::attn_wide_kernel<false, 2>"):
        assert any(must in n for n in names), must
    # register budgets the launch bounds promise: 512-thread kernels at most 256 registers, 768-thread ones 168, 1024-thread ones 128
    for r in rows:
        if r["max_threads"] > 768:
            assert r["vgpr"] + r["agpr"] <= 128, r
        elif r["max_threads"] > 512:
            assert r["vgpr"] + r["agpr"] <= 168, r
        elif r["max_threads"] > 256:
            assert r["vgpr"] + r["agpr"] <= 256, r


def test_timed_step_kernels_keep_two_waves_per_simd():
    """The kernels of the timed step (cached prefill, <= 16 rows) and the many-row attention: at most 256 unified registers."""
    rows = {r["demangled"]: r for r in _rows()}
    for name, r in rows.items():
        if name.startswith("attn_small_kernel<") and (", true, " in name.split("(")[0][-12:] or name.split("(")[0].endswith(", 2>")):
            # the staging variant (pc_attn gather_rows) keeps the K fragments alive until they are stored; its launch is sized to
            # ONE workgroup per CU (small_nstream), i.e. one wave per SIMD: the 512-register budget, and never a spill
            # (likewise the two-row-tile instantiations for 17..32 new rows, "..., 2>")
            assert r["vgpr"] + r["agpr"] <= 512, (name, r["vgpr"], r["agpr"])
        elif name.startswith(("attn_small_kernel<128", "pca::attn_ring_kernel", "pca::attn_wide_kernel")):
            assert r["vgpr"] + r["agpr"] <= 256, (name, r["vgpr"], r["agpr"])
