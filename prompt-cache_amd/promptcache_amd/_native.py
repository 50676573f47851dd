"""ctypes binding of ``libpromptcache_hip.so`` (C-ABI declared in ``include/promptcache_hip.h``).

The product path has NO fallback: if the shared library is missing or a call fails, a
``RuntimeError`` is raised.  PyTorch-ROCm is used only for device memory and the stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

_LIB_NAME = "libpromptcache_hip.so"
_lib: Optional[C.CDLL] = None

_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
_pi32 = C.POINTER(C.c_int32)



class AttnArgs(C.Structure):
    """``pc_attn_args`` of include/promptcache_hip.h (field for field)."""
    _fields_ = [("struct_bytes", C.c_uint32),
                ("q", _vp), ("q_lo", _vp), ("q_batch_stride", _i64), ("q_token_stride", _i64),
                ("k", _vp), ("v", _vp), ("kv_batch_stride", _i64), ("kv_head_stride", _i64),
                ("out", _vp), ("out_lo", _vp), ("out_batch_stride", _i64), ("out_token_stride", _i64),
                ("out_frag_hi", _vp), ("out_frag_lo", _vp),
                ("B", _i32), ("H", _i32), ("Hkv", _i32), ("D", _i32), ("q_len", _i32), ("past_len", _i32),
                ("softmax_scale", _f32),
                ("workspace", _vp), ("workspace_bytes", _i64),
                ("past_len_dev", _vp), ("past_lens", _vp),
                ("key_pos", _vp), ("key_pos_batch_stride", _i64), ("slopes_log2", _vp),
                ("k_lo", _vp), ("v_lo", _vp), ("lo_batch_stride", _i64), ("lo_head_stride", _i64), ("lo_row0", _i32),
                ("counters", _vp),
                ("prefix_k", _vp), ("prefix_v", _vp), ("prefix_k_lo", _vp), ("prefix_v_lo", _vp), ("prefix_head_stride", _i64),
                ("gather_rows", _vp), ("gather_k_plane", _i32), ("gather_v_plane", _i32),
                ("defer_merge", _i32), ("nsplit_out", _vp)]


class KvSeg(C.Structure):
    """``pc_kv_seg``: one staged segment of a prompt (module store, first staged row, rows)."""
    _fields_ = [("src", _vp), ("dst_row", _i32), ("len", _i32)]


class KvRow(C.Structure):
    """``pc_kv_row``: where one staged row lies (pc_kv_row_table)."""
    _fields_ = [("base", C.c_uint64), ("plane_stride16", C.c_uint32), ("flags", C.c_uint32)]


KV_ROW_STAGED = 1
KV_ROW_TABLE_MAX_SEG = 1024


class DenseQkvArgs(C.Structure):
    """``pc_dense_qkv_args`` of include/promptcache_hip.h (field for field)."""
    _fields_ = [("struct_bytes", C.c_uint32),
                ("x_hi", _vp), ("x_lo", _vp), ("ldx", _i64),
                ("w", _vp), ("ldw", _i64), ("K", _i32),
                ("cs", _vp),
                ("q_hi", _vp), ("q_lo", _vp), ("q_token_stride", _i64),
                ("k_arena", _vp), ("v_arena", _vp), ("arena_batch_stride", _i64), ("arena_head_stride", _i64),
                ("k_lo", _vp), ("v_lo", _vp), ("lo_batch_stride", _i64), ("lo_head_stride", _i64), ("lo_row0", _i32),
                ("B", _i32), ("H", _i32), ("Hkv", _i32), ("D", _i32), ("q_len", _i32), ("past_len", _i32), ("cap", _i32),
                ("past_lens", _vp),
                ("x_lo8", _vp), ("x_lo8_scale", _vp), ("ldx8", _i64), ("w8", _vp), ("w8_scale", _vp), ("ldw8", _i64)]


class GemmArgs(C.Structure):
    """``pc_gemm_args`` of include/promptcache_hip.h (field for field)."""
    _fields_ = [("struct_bytes", C.c_uint32), ("epilogue", _i32),
                ("wf", _vp), ("w_scale", _vp), ("xf_hi", _vp), ("xf_lo", _vp),
                ("x", _vp), ("norm_weight", _vp), ("eps", _f32),
                ("M", _i32), ("N", _i32), ("K", _i32), ("rows_dev", _vp),
                ("y", _vp), ("ldy", _i64), ("of_hi", _vp), ("of_lo", _vp), ("kslices", _i32),
                ("ks_tiles", _i32), ("ks_scratch", _vp), ("ks_scratch_bytes", _i64), ("ks_counters", _vp),
                ("x_scale", _vp), ("corr", _vp), ("ldc", _i64), ("corr_has", _vp),
                ("flags", _vp), ("x_raw", _vp), ("w_codes_t", _vp), ("ldt", _i64), ("row_perm", _vp),
                ("cs", _vp), ("q_hi", _vp), ("q_lo", _vp), ("q_token_stride", _i64), ("k_arena", _vp), ("v_arena", _vp),
                ("arena_batch_stride", _i64), ("arena_head_stride", _i64),
                ("B", _i32), ("H", _i32), ("Hkv", _i32), ("D", _i32), ("q_len", _i32), ("past_len", _i32), ("cap", _i32),
                ("past_len_dev", _vp), ("k_lo", _vp), ("v_lo", _vp), ("lo_batch_stride", _i64), ("lo_head_stride", _i64),
                ("lo_base", _i32), ("x_codes8", _vp),
                ("row_max_out", _vp), ("flags_out", _vp), ("out_threshold", _f32)]


class GemmQ8Args(C.Structure):
    """``pc_gemm_q8_args`` of include/promptcache_hip.h (field for field)."""
    _fields_ = [("struct_bytes", C.c_uint32), ("epilogue", _i32),
                ("wf", _vp), ("w_scale", _vp), ("w_codes_t", _vp), ("ldt", _i64), ("row_perm", _vp),
                ("threshold", _f32),
                ("x", _vp), ("norm_weight", _vp), ("eps", _f32),
                ("xf_hi", _vp),
                ("row_max", _vp), ("row_max_units", _i32), ("flags_in", _vp),
                ("M", _i32), ("N", _i32), ("K", _i32),
                ("y", _vp), ("ldy", _i64), ("of_hi", _vp), ("of_lo", _vp),
                ("row_max_out", _vp), ("flags_out", _vp),
                ("flags_clear", _vp), ("clear_bytes", _i32),
                ("kslices", _i32), ("ks_tiles", _i32), ("ks_scratch", _vp), ("ks_scratch_bytes", _i64), ("ks_counters", _vp),
                ("cs", _vp), ("q_hi", _vp), ("q_lo", _vp), ("q_token_stride", _i64), ("k_arena", _vp), ("v_arena", _vp),
                ("arena_batch_stride", _i64), ("arena_head_stride", _i64),
                ("B", _i32), ("H", _i32), ("Hkv", _i32), ("D", _i32), ("q_len", _i32), ("past_len", _i32), ("cap", _i32),
                ("past_len_dev", _vp), ("k_lo", _vp), ("v_lo", _vp), ("lo_batch_stride", _i64), ("lo_head_stride", _i64),
                ("lo_base", _i32),
                ("dbg_codes", _vp), ("dbg_scale", _vp), ("dbg_flags", _vp),
                ("part_o", _vp), ("part_ml", _vp), ("part_nsplit", _i32), ("part_head_dim", _i32),
                ("x_codes8", _vp), ("x_scale", _vp), ("x_flags", _vp)]


# name -> (restype, argtypes); mirrors include/promptcache_hip.h one to one
SIGNATURES = {
    "pc_attn": (C.c_int, [C.POINTER(AttnArgs), _vp]),
    "pc_gemm": (C.c_int, [C.POINTER(GemmArgs), _vp]),
    "pc_gemm_q8": (C.c_int, [C.POINTER(GemmQ8Args), _vp]),
    "pc_gemm_part": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "pc_dev_gemm_trace": (C.c_int, [_vp]),
    "pc_dev_attn_trace": (C.c_int, [_vp]),
    "pc_gemm_skinny_ks_scratch_bytes": (C.c_int64, [_i32, _i32]),
    "pc_version": (C.c_int, []),
    "pc_last_error_string": (C.c_char_p, []),
    "pc_kv_gather": (C.c_int, [C.POINTER(_vp), _pi32, _pi32, _i32, _vp, _i32, _i32, _i32, _i32, _vp]),
    "pc_kv_slice_store": (C.c_int, [_vp, _i32, _pi32, _pi32, C.POINTER(_vp), _i32, _i32, _i32, _i32, _vp]),
    "pc_kv_row_table": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "pc_attn_gather_ok": (C.c_int, [C.POINTER(AttnArgs)]),
    "pc_rope_table": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp]),
    "pc_rope_append": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _i64, _i64, _vp, _vp, _i64, _i64, _vp, _vp, _i64, _i64, _vp,
                                 _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "pc_attn_workspace_bytes": (C.c_int64, [_i32, _i32, _i32, _i32, _i32]),
    "pc_rope_append_ex": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _i64, _i64, _vp, _vp, _i64, _i64, _vp, _vp, _i64, _i64, _vp,
                                    _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i64, _i64, _i32, _i64, _vp]),
    "pc_rmsnorm_split": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _f32, _vp]),
    "pc_layernorm_split": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _vp]),
    "pc_silu_mul_split": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "pc_gelu_split": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "pc_add3": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "pc_rmsnorm_frag": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _f32, _vp, _i32, _vp]),
    "pc_rmsnorm": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _f32, _i32, _vp]),
    "pc_layernorm": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _f32, _vp]),
    "pc_layernorm_frag": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _vp, _i32, _vp]),
    "pc_gelu": (C.c_int, [_vp, _vp, _i64, _vp]),
    "pc_silu_mul": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "pc_embed_gather": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "pc_probe_layouts": (C.c_int, [_vp, _vp, _vp]),
    "pc_fetch_block": (C.c_int, [_vp, _vp, _i32, _vp]),
    "pc_prefill_prologue": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _vp]),
    "pc_rope_append_var": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _i64, _i64, _vp, _vp, _i64, _i64, _vp, _vp, _i64, _i64, _vp,
                                     _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i64, _i64, _i32, _vp]),
    "pc_greedy_advance": (C.c_int, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "pc_quant_act_i8": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _f32, _vp, _vp]),
    "pc_rmsnorm_quant_i8": (C.c_int, [_vp, _vp, _f32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _f32, _vp, _vp]),
    "pc_outlier_corr": (C.c_int, [_vp, _i32, _vp, _vp, _i64, _i32, _vp, _vp, _i64, _vp, _vp, _i32, _i32, _vp, _i64, _vp, _vp]),
    "pc_gemm_dense_a8": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _vp, _i64, _vp]),
    "pc_gemm_dense": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _vp, _i64, _vp]),
    "pc_gemm_dense_qkv_rope": (C.c_int, [_vp, _vp]),
    "pc_gemm_dense_ws": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp]),
}

# Entry points of -DPC_DEV_SWEEPS builds only (csrc/pc_dev.h: measured negative results kept for A/B; not in the product library
# and not part of include/promptcache_hip.h): bound when the loaded library has them.
DEV_SIGNATURES = {
    "pc_chain_sync_words": (C.c_int32, []),
    "pc_chain_sync_err_word": (C.c_int32, []),
    "pc_gemm_chain": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _i32, _i32, _vp, _vp, _f32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                _i64, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i64, _i64,
                                _i32, _vp, _vp]),
    "pc_quant_rows_i8": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _i64, _vp, _vp]),
    "pc_gemm_part_rows": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i64, _i32, _vp, _i64, _vp, _vp]),
    "pc_gemm_dense_lo8": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _i64, _vp, _vp,
                                    _i64, _vp, _i64, _vp]),
}


def lib_path() -> str:
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), _LIB_NAME)


def load() -> C.CDLL:
    """Load the HIP extension (once).  Fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{_LIB_NAME} not found at {path}: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the prompt-cache hot path.")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in DEV_SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    _lib = lib
    return lib


def has(name: str) -> bool:
    """Whether the loaded library exports ``name`` (the dev-only entry points of csrc/pc_dev.h exist in -DPC_DEV_SWEEPS builds)."""
    return hasattr(load(), name)


def _dev(name: str):
    fn = getattr(load(), name, None)
    if fn is None:
        raise RuntimeError(f"{name} exists only in a dev build of the library: PC_BUILD_FLAGS=-DPC_DEV_SWEEPS python __graft_entry__.py --force")
    return fn


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().pc_last_error_string()
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


def _ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


# ------------------------------------------------------------------------------------------------
# thin typed wrappers (tensor.data_ptr() + sizes; no torch types cross the ABI)
# ------------------------------------------------------------------------------------------------

def kv_gather(seg_ptrs: Sequence[int], seg_lens: Sequence[int], seg_dst_off: Sequence[int], dst,
              n_layers: int, n_kv_heads: int, head_dim: int, max_ctx: int, stream: Optional[int] = None) -> None:
    n = len(seg_ptrs)
    a_ptr = (_vp * n)(*seg_ptrs)
    a_len = (C.c_int32 * n)(*seg_lens)
    a_off = (C.c_int32 * n)(*seg_dst_off)
    rc = load().pc_kv_gather(a_ptr, a_len, a_off, n, dst.data_ptr(), n_layers, n_kv_heads, head_dim, max_ctx,
                             current_stream() if stream is None else stream)
    check(rc, "pc_kv_gather")


def kv_slice_store(src, src_cap: int, seg_src_off: Sequence[int], seg_lens: Sequence[int], seg_dst_ptrs: Sequence[int],
                   n_layers: int, n_kv_heads: int, head_dim: int, stream: Optional[int] = None) -> None:
    n = len(seg_dst_ptrs)
    a_off = (C.c_int32 * n)(*seg_src_off)
    a_len = (C.c_int32 * n)(*seg_lens)
    a_ptr = (_vp * n)(*seg_dst_ptrs)
    rc = load().pc_kv_slice_store(src.data_ptr(), src_cap, a_off, a_len, a_ptr, n, n_layers, n_kv_heads, head_dim,
                                  current_stream() if stream is None else stream)
    check(rc, "pc_kv_slice_store")


def kv_row_table(segs_dev, nseg_dev, max_seg: int, total_rows_dev, dst, n_kv_heads: int, head_dim: int, max_ctx: int, rows,
                 stream: Optional[int] = None) -> None:
    """Expand the staging plan (device array of ``pc_kv_seg``, count and total row count in device words) to one ``pc_kv_row`` per
    staged row of ``dst`` (the arena buffer of one batch row) -- what ``attn_fwd(..., gather=...)`` reads."""
    rc = load().pc_kv_row_table(segs_dev.data_ptr(), nseg_dev.data_ptr(), max_seg, total_rows_dev.data_ptr(), dst.data_ptr(),
                                n_kv_heads, head_dim, max_ctx, rows.data_ptr(), current_stream() if stream is None else stream)
    check(rc, "pc_kv_row_table")


def rope_table(pos_i32, inv_freq, cs_out, n_tok: int, head_dim: int, stream: Optional[int] = None) -> None:
    rc = load().pc_rope_table(pos_i32.data_ptr(), inv_freq.data_ptr(), cs_out.data_ptr(), n_tok, head_dim,
                              current_stream() if stream is None else stream)
    check(rc, "pc_rope_table")


def rope_append(q, q_bs, q_ts, q_out, qo_bs, qo_ts, k_new, v_new, n_bs, n_ts, k_arena, v_arena, a_bs, a_hs, cs,
                B, H, Hkv, D, q_len, past_len, cap, in_is_f32: bool, past_len_dev=None, q_out_lo=None,
                stream: Optional[int] = None, kv_lo=None, in2_offset: int = 0, past_lens=None) -> None:
    """``in2_offset`` (elements, with ``kv_lo``): inputs are ``x[i] + x[i + in2_offset]`` (the halves of a [hi; lo] GEMM).
    ``past_lens`` (device int32 [B]): one past length per batch row (``past_len`` = their maximum), pc_rope_append_var."""
    if past_lens is not None:
        assert in2_offset == 0 and past_len_dev is None and (kv_lo is None or kv_lo[4] == 0)
        rc = load().pc_rope_append_var(q.data_ptr(), q_bs, q_ts, q_out.data_ptr(), _ptr(q_out_lo), qo_bs, qo_ts, k_new.data_ptr(),
                                       v_new.data_ptr(), n_bs, n_ts, k_arena.data_ptr(), v_arena.data_ptr(), a_bs, a_hs,
                                       cs.data_ptr(), B, H, Hkv, D, q_len, past_len, cap, int(in_is_f32), past_lens.data_ptr(),
                                       None if kv_lo is None else kv_lo[0].data_ptr(), None if kv_lo is None else kv_lo[1].data_ptr(),
                                       0 if kv_lo is None else kv_lo[2], 0 if kv_lo is None else kv_lo[3], 0,
                                       current_stream() if stream is None else stream)
        check(rc, "pc_rope_append_var")
        return
    if kv_lo is not None:
        rc = load().pc_rope_append_ex(q.data_ptr(), q_bs, q_ts, q_out.data_ptr(), _ptr(q_out_lo), qo_bs, qo_ts, k_new.data_ptr(),
                                      v_new.data_ptr(), n_bs, n_ts, k_arena.data_ptr(), v_arena.data_ptr(), a_bs, a_hs,
                                      cs.data_ptr(), B, H, Hkv, D, q_len, past_len, cap, int(in_is_f32), _ptr(past_len_dev),
                                      kv_lo[0].data_ptr(), kv_lo[1].data_ptr(), kv_lo[2], kv_lo[3], kv_lo[4], in2_offset,
                                      current_stream() if stream is None else stream)
        check(rc, "pc_rope_append_ex")
        return
    assert in2_offset == 0, "in2_offset goes through pc_rope_append_ex (pass kv_lo)"
    rc = load().pc_rope_append(q.data_ptr(), q_bs, q_ts, q_out.data_ptr(), _ptr(q_out_lo), qo_bs, qo_ts, k_new.data_ptr(),
                               v_new.data_ptr(), n_bs, n_ts, k_arena.data_ptr(), v_arena.data_ptr(), a_bs, a_hs,
                               cs.data_ptr(), B, H, Hkv, D, q_len, past_len, cap, int(in_is_f32), _ptr(past_len_dev),
                               current_stream() if stream is None else stream)
    check(rc, "pc_rope_append")


def attn_workspace_bytes(B: int, H: int, D: int, q_len: int, kv_len_max: int) -> int:
    return int(load().pc_attn_workspace_bytes(B, H, D, q_len, kv_len_max))


def _attn_args(q, q_bs, q_ts, k, v, kv_bs, kv_hs, out, o_bs, o_ts, B, H, Hkv, D, q_len, past_len, scale,
               workspace=None, past_len_dev=None, out_frag=None, q_lo=None, alibi=None, out_lo=None, kv_lo=None, past_lens=None,
               counters=None, prefix=None, gather=None) -> AttnArgs:
    ws_bytes = 0 if workspace is None else workspace.numel() * workspace.element_size()
    fh, fl = (None, None) if out_frag is None else out_frag
    kpos, slopes = (None, None) if alibi is None else alibi
    lo = (None, None, 0, 0, 0) if kv_lo is None else kv_lo
    pf = (None, None, None, None, 0) if prefix is None else prefix
    gr = (None, 0, 0) if gather is None else gather
    return AttnArgs(C.sizeof(AttnArgs), q.data_ptr(), _ptr(q_lo), q_bs, q_ts, k.data_ptr(), v.data_ptr(), kv_bs, kv_hs,
                    _ptr(out), _ptr(out_lo), o_bs, o_ts, _ptr(fh), _ptr(fl), B, H, Hkv, D, q_len, past_len, scale,
                    _ptr(workspace), ws_bytes, _ptr(past_len_dev), _ptr(past_lens), _ptr(kpos),
                    0 if kpos is None else kpos.stride(0), _ptr(slopes), _ptr(lo[0]), _ptr(lo[1]), lo[2], lo[3], lo[4],
                    _ptr(counters), _ptr(pf[0]), _ptr(pf[1]), _ptr(pf[2]), _ptr(pf[3]), pf[4], _ptr(gr[0]), gr[1], gr[2])


def attn_gather_ok(*args, **kw) -> bool:
    """Whether ``attn_fwd`` with these arguments runs on a kernel that takes ``gather`` (same arguments, nothing is launched)."""
    a = _attn_args(*args, **kw)
    return bool(load().pc_attn_gather_ok(C.byref(a)))


def attn_fwd(q, q_bs, q_ts, k, v, kv_bs, kv_hs, out, o_bs, o_ts, B, H, Hkv, D, q_len, past_len, scale,
             workspace=None, past_len_dev=None, out_frag=None, q_lo=None, stream: Optional[int] = None,
             alibi=None, out_lo=None, kv_lo=None, past_lens=None, counters=None, prefix=None, gather=None,
             defer_merge: bool = False) -> int:
    """The attention of one layer through ``pc_attn`` (struct entry; every option is a field).
    ``defer_merge``: a split-KV launch of <= 16 rows leaves its partials in ``workspace`` for ``gemm_q8(part=...)``; returns the
    partials per row (1: merged as usual, the output planes are final).
    ``past_lens`` (device int32 [B]): one past length per batch row (``past_len`` = their maximum).
    ``out_frag=(hi, lo)``: write split-precision fragment planes for the o_proj launch instead of ``out``.
    ``alibi=(key_pos fp32 [B, stride], slopes_log2 fp32 [H])``: MPT's additive position bias.
    ``out_lo``: row-major residual plane of ``out``; ``kv_lo=(k_lo, v_lo, batch_stride, head_stride, row0)``: residuals of K/V rows
    from key index row0 on (written by ``rope_append(..., kv_lo=...)`` / the q|k|v projection).
    ``counters`` (int32 [B*H], zero before first use): merge the split-KV partials inside the launch.
    ``prefix=(k, v, k_lo | None, v_lo | None, head_stride)`` with ``past_lens``: batch row b attends to rows [0, past_lens[b]) of
    these shared planes, then to its own rows, which ``k`` / ``v`` (and ``kv_lo``) hold from row 0 on.
    ``gather=(row table, k plane, v plane)`` (B = 1, ``attn_gather_ok``): stage while reading -- key rows below ``past_len`` are
    read from where the table (``kv_row_table``) says they lie and written to ``k`` / ``v`` unless they are there already."""
    a = _attn_args(q, q_bs, q_ts, k, v, kv_bs, kv_hs, out, o_bs, o_ts, B, H, Hkv, D, q_len, past_len, scale, workspace,
                   past_len_dev, out_frag, q_lo, alibi, out_lo, kv_lo, past_lens, counters, prefix, gather)
    ns = C.c_int32(1)
    if defer_merge:
        a.defer_merge, a.nsplit_out = 1, C.addressof(ns)
    rc = load().pc_attn(C.byref(a), current_stream() if stream is None else stream)
    check(rc, "pc_attn")
    return int(ns.value)


EPI_STORE, EPI_ADD, EPI_SILU, EPI_GELU = 0, 1, 2, 4


EPI_QKV_ROPE = 3


def _gemm(stream=None, **f) -> None:
    """One ``pc_gemm`` call: keyword = field of ``pc_gemm_args`` (tensors become pointers, everything else 0 / NULL)."""
    a = GemmArgs()
    a.struct_bytes = C.sizeof(GemmArgs)
    a.kslices, a.lo_base = 1, -1
    for k, v in f.items():
        if v is None:
            continue
        setattr(a, k, v.data_ptr() if hasattr(v, "data_ptr") else v)
    rc = load().pc_gemm(C.byref(a), current_stream() if stream is None else stream)
    check(rc, "pc_gemm")


def _qkv_fields(cs, q_hi, q_lo, q_ts, k_arena, v_arena, a_bs, a_hs, B, H, Hkv, D, q_len, past_len, cap, past_len_dev, kv_lo, lo_base):
    lo = (None, None, 0, 0) if kv_lo is None else kv_lo
    return dict(epilogue=EPI_QKV_ROPE, cs=cs, q_hi=q_hi, q_lo=q_lo, q_token_stride=q_ts, k_arena=k_arena, v_arena=v_arena,
                arena_batch_stride=a_bs, arena_head_stride=a_hs, B=B, H=H, Hkv=Hkv, D=D, q_len=q_len, past_len=past_len, cap=cap,
                past_len_dev=past_len_dev, k_lo=lo[0], v_lo=lo[1], lo_batch_stride=lo[2], lo_head_stride=lo[3], lo_base=lo_base)


def gemm_skinny(wf, xf_hi, xf_lo, M: int, N: int, K: int, epilogue: int, y=None, ldy: int = 0, of_hi=None, of_lo=None,
                kslices: int = 1, stream: Optional[int] = None, wscale=None, rows_dev=None) -> None:
    """``kslices > 1`` (plain-store epilogue): ``y`` is ``[kslices][M][ldy]`` slabs of partial sums.
    ``wscale`` (fp32 [N]): ``wf`` is an int8 fragment image (``to_weight_frags_i8``)."""
    _gemm(stream, epilogue=epilogue, wf=wf, w_scale=wscale, xf_hi=xf_hi, xf_lo=xf_lo, M=M, N=N, K=K, y=y, ldy=ldy, of_hi=of_hi,
          of_lo=of_lo, kslices=kslices, rows_dev=rows_dev)


def gemm_skinny_ks_scratch_bytes(N: int, kslices: int) -> int:
    return int(load().pc_gemm_skinny_ks_scratch_bytes(N, kslices))


def gemm_skinny_ks(wf, xf_hi, xf_lo, M: int, N: int, K: int, y, ldy: int, kslices: int, tiles_per_wg: int, scratch, counters,
                   stream: Optional[int] = None, rows_dev=None) -> None:
    """``y += x @ W^T`` (M <= 16) with K split over ``kslices`` workgroup slices and the reduction inside the launch;
    ``counters`` (int32 [ceil(N/16/tiles_per_wg)]) must be zero before the first launch (launches leave them zero)."""
    _gemm(stream, epilogue=EPI_ADD, wf=wf, xf_hi=xf_hi, xf_lo=xf_lo, M=M, N=N, K=K, y=y, ldy=ldy, kslices=kslices,
          ks_tiles=tiles_per_wg, ks_scratch=scratch, ks_scratch_bytes=scratch.numel() * scratch.element_size(), ks_counters=counters,
          rows_dev=rows_dev)


def gemm_qkv_rope(wf_perm, xf_hi, xf_lo, M, K, cs, q_hi, q_lo, q_ts, k_arena, v_arena, a_bs, a_hs, B, H, Hkv, D, q_len,
                  past_len, cap, past_len_dev=None, stream: Optional[int] = None, kv_lo=None, wscale=None,
                  lo_base: int = -1, rows_dev=None, kslices: int = 1, scratch=None) -> None:
    """``kv_lo=(k_lo, v_lo, batch_stride, head_stride)``: also write the fp16 residuals of the new K / V rows; ``lo_base``
    places them (-1: compact ``[B][Hkv][q_len][D]`` rows of this pass; >= 0 / -2: a residual tail that outlives the pass).
    ``kslices > 1`` (65..512 rows): K is cut across workgroups, the partial sums go through ``scratch`` (fp32, >= kslices * M * N
    elements) and the rotation / append runs as a second launch over them."""
    _gemm(stream, wf=wf_perm, w_scale=wscale, xf_hi=xf_hi, xf_lo=xf_lo, M=M, K=K, rows_dev=rows_dev, kslices=kslices,
          ks_scratch=scratch, ks_scratch_bytes=0 if scratch is None else scratch.numel() * scratch.element_size(),
          **_qkv_fields(cs, q_hi, q_lo, q_ts, k_arena, v_arena, a_bs, a_hs, B, H, Hkv, D, q_len, past_len, cap, past_len_dev, kv_lo, lo_base))


def layernorm(x_f32, weight, bias, out, rows: int, hidden: int, eps: float, stream: Optional[int] = None) -> None:
    rc = load().pc_layernorm(x_f32.data_ptr(), weight.data_ptr(), _ptr(bias), out.data_ptr(), rows, hidden, eps,
                             current_stream() if stream is None else stream)
    check(rc, "pc_layernorm")


def layernorm_frag(x_f32, weight, bias, xf_hi, xf_lo, rows: int, hidden: int, eps: float, slabs=None, nslabs: int = 0,
                   stream: Optional[int] = None) -> None:
    rc = load().pc_layernorm_frag(x_f32.data_ptr(), weight.data_ptr(), _ptr(bias), xf_hi.data_ptr(), xf_lo.data_ptr(),
                                  rows, hidden, eps, _ptr(slabs), nslabs, current_stream() if stream is None else stream)
    check(rc, "pc_layernorm_frag")


def gelu(x_f32, out, n: int, stream: Optional[int] = None) -> None:
    rc = load().pc_gelu(x_f32.data_ptr(), out.data_ptr(), n, current_stream() if stream is None else stream)
    check(rc, "pc_gelu")


def gemm_skinny_norm(wf, x_f32, norm_weight, eps: float, M: int, N: int, K: int, epilogue: int, y=None, ldy: int = 0,
                     of_hi=None, of_lo=None, stream: Optional[int] = None, wscale=None, rows_dev=None) -> None:
    """RMSNorm folded into the projection (M <= 16): y = W . (norm_weight * x) * rsqrt(mean(x^2) + eps)."""
    _gemm(stream, epilogue=epilogue, wf=wf, w_scale=wscale, x=x_f32, norm_weight=norm_weight, eps=eps, M=M, N=N, K=K, y=y, ldy=ldy,
          of_hi=of_hi, of_lo=of_lo, rows_dev=rows_dev)


def gemm_qkv_rope_norm(wf_perm, x_f32, norm_weight, eps: float, M, K, cs, q_hi, q_lo, q_ts, k_arena, v_arena, a_bs, a_hs, B, H,
                       Hkv, D, q_len, past_len, cap, past_len_dev=None, stream: Optional[int] = None, kv_lo=None,
                       wscale=None, lo_base: int = -1, rows_dev=None) -> None:
    _gemm(stream, wf=wf_perm, w_scale=wscale, x=x_f32, norm_weight=norm_weight, eps=eps, M=M, K=K, rows_dev=rows_dev,
          **_qkv_fields(cs, q_hi, q_lo, q_ts, k_arena, v_arena, a_bs, a_hs, B, H, Hkv, D, q_len, past_len, cap, past_len_dev, kv_lo, lo_base))


def qkv_rope_row_perm(n_heads_total: int, D: int):
    """Row permutation of the fused [q;k;v] weight for pc_gemm_qkv_rope: tile j of a head = features
    8j..8j+7 then D/2+8j..D/2+8j+7."""
    import torch
    idx = []
    for h in range(n_heads_total):
        for j in range(D // 16):
            idx += [h * D + 8 * j + r for r in range(8)] + [h * D + D // 2 + 8 * j + r for r in range(8)]
    return torch.tensor(idx, dtype=torch.long)


def rmsnorm_frag(x_f32, weight, xf_hi, xf_lo, rows: int, hidden: int, eps: float, slabs=None, nslabs: int = 0,
                 stream: Optional[int] = None) -> None:
    """``slabs`` ([nslabs][rows][hidden] fp32): added to ``x_f32`` in place (fixed order) before the norm."""
    rc = load().pc_rmsnorm_frag(x_f32.data_ptr(), weight.data_ptr(), xf_hi.data_ptr(), xf_lo.data_ptr(), rows, hidden, eps,
                                _ptr(slabs), nslabs if slabs is not None else 0,
                                current_stream() if stream is None else stream)
    check(rc, "pc_rmsnorm_frag")


def to_weight_frags(w):
    """nn.Linear weight [N, K] (fp16) -> fragment-major Wf[N/16][K/32][64][8] (see include/promptcache_hip.h)."""
    N, K = w.shape
    assert N % 16 == 0 and K % 32 == 0, (N, K)
    return w.view(N // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous()


def quantize_rows_int8(w):
    """Row-wise absmax int8 (the weight quantiser of LLM.int8): ``q = round_half_even(w * (127 / absmax_row))`` in
    [-127, 127], ``scale = absmax_row / 127`` (all-zero row: q = 0, scale = 1), on the weight's device.  The two per-row
    quotients are formed as tensor / tensor in fp64 and rounded to fp32 -- which is the correctly rounded fp32 quotient
    (53 >= 2 * 24 + 2 bits) on any device; ``scalar / tensor`` would be lowered to ``reciprocal * scalar`` and lands one ulp
    off often enough to flip exact ties (w = absmax / 2) on real weights.  The per-element step is one correctly rounded
    fp32 product, so the result is bit-identical to ``oracle/int8_oracle.py``.
    Returns ``(q int8 [N, K], scale fp32 [N])``."""
    import torch
    wf = w.float()
    amax = wf.abs().amax(dim=1)
    ok = amax > 0
    one = torch.ones_like(amax)
    safe = torch.where(ok, amax, one).double()
    c127 = torch.full_like(safe, 127.0)
    scale = torch.where(ok, (safe / c127).float(), one)
    inv = torch.where(ok, (c127 / safe).float(), torch.zeros_like(amax))
    q = torch.round(wf * inv[:, None]).clamp_(-127, 127).to(torch.int8)
    return q, scale.contiguous()


def to_weight_frags_i8(q):
    """int8 weight [N, K] -> fragment-major image [N/16][K/64][64][16] of OFFSET-BINARY bytes (q + 128) for the W8 kernels:
    lane 16 g + m of tile t holds, for the k-step pair s, row 16 t + m's features 64 s + 8 g .. + 7 followed by
    64 s + 32 + 8 g .. + 7 (one 16-byte load feeds two k-steps).  Returned as ``[N/16][K/64][4][16][2][8]``."""
    import torch
    N, K = q.shape
    assert N % 16 == 0 and K % 64 == 0 and q.dtype == torch.int8, (N, K, q.dtype)
    u = (q.to(torch.int16) + 128).to(torch.uint8)
    return u.view(N // 16, 16, K // 64, 2, 4, 8).permute(0, 2, 4, 1, 3, 5).contiguous()


def to_act_frags(x):
    """[M, K] float tensor -> (hi, lo) fp16 fragment planes [ceil(M/16)][K/32][64][8] (test/utility helper; the
    hot path produces these planes directly in the kernels)."""
    import torch
    M, K = x.shape
    mt = (M + 15) // 16
    xp = torch.zeros((mt * 16, K), dtype=torch.float32, device=x.device)
    xp[:M] = x.float()
    hi = xp.half()
    lo = (xp - hi.float()).half()
    f = lambda t: t.view(mt, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous()
    return f(hi), f(lo)


def from_act_frags(f, M: int):
    """Inverse of the plane layout: [mt][K/32][64][8] -> [M, K] (test helper)."""
    mt, ks = f.shape[0], f.shape[1]
    return f.view(mt, ks, 4, 16, 8).permute(0, 3, 1, 2, 4).reshape(mt * 16, ks * 32)[:M]


def rmsnorm(x, weight, out, rows: int, hidden: int, eps: float, x_is_f32: bool, stream: Optional[int] = None) -> None:
    rc = load().pc_rmsnorm(x.data_ptr(), weight.data_ptr(), out.data_ptr(), rows, hidden, eps, int(x_is_f32),
                           current_stream() if stream is None else stream)
    check(rc, "pc_rmsnorm")


def silu_mul(gate_up, out, rows: int, inter: int, in_is_f32: bool = False, stream: Optional[int] = None) -> None:
    rc = load().pc_silu_mul(gate_up.data_ptr(), out.data_ptr(), rows, inter, int(in_is_f32),
                            current_stream() if stream is None else stream)
    check(rc, "pc_silu_mul")


def rmsnorm_split(x_f32, weight, out_hi, out_lo, rows: int, hidden: int, eps: float, stream: Optional[int] = None) -> None:
    rc = load().pc_rmsnorm_split(x_f32.data_ptr(), weight.data_ptr(), out_hi.data_ptr(), out_lo.data_ptr(), rows, hidden, eps,
                                 current_stream() if stream is None else stream)
    check(rc, "pc_rmsnorm_split")


def layernorm_split(x_f32, weight, bias, out_hi, out_lo, rows: int, hidden: int, eps: float, stream: Optional[int] = None) -> None:
    rc = load().pc_layernorm_split(x_f32.data_ptr(), weight.data_ptr(), _ptr(bias), out_hi.data_ptr(), out_lo.data_ptr(), rows,
                                   hidden, eps, current_stream() if stream is None else stream)
    check(rc, "pc_layernorm_split")


def silu_mul_split(gate_up, gate_up2, out_hi, out_lo, rows: int, inter: int, stream: Optional[int] = None) -> None:
    rc = load().pc_silu_mul_split(gate_up.data_ptr(), _ptr(gate_up2), out_hi.data_ptr(), out_lo.data_ptr(), rows, inter,
                                  current_stream() if stream is None else stream)
    check(rc, "pc_silu_mul_split")


def gelu_split(x, x2, out_hi, out_lo, n: int, stream: Optional[int] = None) -> None:
    rc = load().pc_gelu_split(x.data_ptr(), _ptr(x2), out_hi.data_ptr(), out_lo.data_ptr(), n,
                              current_stream() if stream is None else stream)
    check(rc, "pc_gelu_split")


def add3(x, a, b, n: int, stream: Optional[int] = None) -> None:
    rc = load().pc_add3(x.data_ptr(), a.data_ptr(), b.data_ptr(), n, current_stream() if stream is None else stream)
    check(rc, "pc_add3")


def embed_gather(table, ids_i64, out, n_tok: int, hidden: int, vocab: int, stream: Optional[int] = None) -> None:
    rc = load().pc_embed_gather(table.data_ptr(), ids_i64.data_ptr(), out.data_ptr(), n_tok, hidden, vocab,
                                current_stream() if stream is None else stream)
    check(rc, "pc_embed_gather")


def fetch_block(host_pinned, dst, nbytes: int, stream: Optional[int] = None) -> None:
    """Graph node: pull ``nbytes`` of a pinned host tensor into the device tensor ``dst`` (``pc_fetch_block``)."""
    rc = load().pc_fetch_block(host_pinned.data_ptr(), dst.data_ptr(), nbytes, current_stream() if stream is None else stream)
    check(rc, "pc_fetch_block")


def prefill_prologue(host_pinned, dev_block, nbytes: int, n_tok: int, o_pos: int, o_words: int, o_segs: int, max_seg: int, embed_table,
                     hidden: int, vocab: int, x_out, inv_freq, head_dim: int, cs_out, rows=None, dst=None, max_ctx: int = 0,
                     stream: Optional[int] = None) -> None:
    """Graph node: host block -> device block, embedding rows -> fp32 residual stream, (cos, sin) rows, row table (``pc_prefill_prologue``)."""
    rc = load().pc_prefill_prologue(host_pinned.data_ptr(), dev_block.data_ptr(), nbytes, n_tok, o_pos, o_words, o_segs, max_seg,
                                    embed_table.data_ptr(), hidden, vocab, x_out.data_ptr(), inv_freq.data_ptr(), head_dim,
                                    cs_out.data_ptr(), _ptr(rows), _ptr(dst), max_ctx, current_stream() if stream is None else stream)
    check(rc, "pc_prefill_prologue")


def probe_layouts(out_mfma, out_tr, stream: Optional[int] = None) -> None:
    rc = load().pc_probe_layouts(out_mfma.data_ptr(), out_tr.data_ptr(), current_stream() if stream is None else stream)
    check(rc, "pc_probe_layouts")


def gemm_dense(x_hi, x_lo, w, M: int, N: int, K: int, epilogue: int, y=None, out_hi=None, out_lo=None, wscale=None,
               ldx: Optional[int] = None, ldw: Optional[int] = None, ldy: Optional[int] = None, ldo: Optional[int] = None,
               stream: Optional[int] = None, workspace=None) -> None:
    """Many-row projection ``(x_hi + x_lo) @ w^T`` (pc_gemm_dense.hip): fp16 row-major operands, fp32 accumulation,
    epilogue EPI_STORE (y = .), EPI_ADD (y += .), EPI_SILU (out = silu(gate) * up as hi / lo planes, w = [gate; up]) or
    EPI_GELU (out = gelu(.) planes).  ``x_lo`` may be None (single-precision-plane activations).
    ``workspace`` (any device tensor): scratch for the split-K form of few-row launches (pc_gemm_dense_ws)."""
    ld_o = 0
    if out_hi is not None:
        ld_o = ldo if ldo is not None else out_hi.stride(-2)
    if workspace is not None:
        rc = load().pc_gemm_dense_ws(x_hi.data_ptr(), _ptr(x_lo), x_hi.stride(-2) if ldx is None else ldx, w.data_ptr(),
                                     w.stride(-2) if ldw is None else ldw, _ptr(wscale), M, N, K, epilogue, _ptr(y),
                                     (0 if y is None else y.stride(-2)) if ldy is None else ldy, _ptr(out_hi), _ptr(out_lo),
                                     ld_o, workspace.data_ptr(), workspace.numel() * workspace.element_size(),
                                     current_stream() if stream is None else stream)
        check(rc, "pc_gemm_dense_ws")
        return
    rc = load().pc_gemm_dense(x_hi.data_ptr(), _ptr(x_lo), x_hi.stride(-2) if ldx is None else ldx, w.data_ptr(),
                              w.stride(-2) if ldw is None else ldw, _ptr(wscale), M, N, K, epilogue, _ptr(y),
                              (0 if y is None else y.stride(-2)) if ldy is None else ldy, _ptr(out_hi), _ptr(out_lo), ld_o,
                              current_stream() if stream is None else stream)
    check(rc, "pc_gemm_dense")


def quant_rows_i8(x_lo, M: int, K: int, codes, scale, stream: Optional[int] = None) -> None:
    """A residual activation plane fp16 ``[M, K]`` -> row-wise absmax int8 codes ``codes [M, K]`` + ``scale [M]`` fp32
    (``pc_quant_rows_i8``): the second operand plane of ``gemm_dense(..., lo8=...)``."""
    rc = _dev("pc_quant_rows_i8")(x_lo.data_ptr(), x_lo.stride(-2), M, K, codes.data_ptr(), codes.stride(-2), scale.data_ptr(),
                                 current_stream() if stream is None else stream)
    check(rc, "pc_quant_rows_i8")


def gemm_dense_lo8(x_hi, x_lo8, x_lo8_scale, w, w8, w8_scale, M: int, N: int, K: int, epilogue: int, y=None, out_hi=None, out_lo=None,
                   workspace=None, stream: Optional[int] = None) -> None:
    """``gemm_dense`` with the residual activation plane on the int8 MFMA (``pc_gemm_dense_lo8``): ``x_lo8`` int8 ``[M, K]`` +
    ``x_lo8_scale [M]`` from ``quant_rows_i8``, ``w8`` int8 ``[N, K]`` + ``w8_scale [N]`` from ``quantize_rows_int8(w)``."""
    rc = _dev("pc_gemm_dense_lo8")(x_hi.data_ptr(), x_hi.stride(-2), x_lo8.data_ptr(), x_lo8_scale.data_ptr(), x_lo8.stride(-2),
                                  w.data_ptr(), w.stride(-2), w8.data_ptr(), w8_scale.data_ptr(), w8.stride(-2), M, N, K, epilogue,
                                  _ptr(y), 0 if y is None else y.stride(-2), _ptr(out_hi), _ptr(out_lo),
                                  0 if out_hi is None else out_hi.stride(-2), _ptr(workspace),
                                  0 if workspace is None else workspace.numel() * workspace.element_size(),
                                  current_stream() if stream is None else stream)
    check(rc, "pc_gemm_dense_lo8")


def gemm_dense_qkv_rope(x_hi, x_lo, w, K: int, cs, q_hi, q_lo, q_ts: int, k_arena, v_arena, a_bs: int, a_hs: int, B: int, H: int,
                        Hkv: int, D: int, q_len: int, past_len: int, cap: int, kv_lo=None, past_lens=None,
                        stream: Optional[int] = None, lo8=None) -> None:
    """The fused many-row q|k|v projection (``pc_gemm_dense_qkv_rope``): ``(x_hi + x_lo) @ [q; k; v]^T``, RoPE from the table
    ``cs``, rotated q into ``q_hi`` / ``q_lo``, rotated k and v into the arena planes behind each batch row's past (and their
    residuals into ``kv_lo = (k_lo, v_lo, batch_stride, head_stride, row0)``).  ``lo8 = (x_lo8, x_lo8_scale, w8, w8_scale)``
    instead of ``x_lo``: the residual plane on the int8 MFMA (``pc_gemm_dense_lo8``)."""
    lo = (None, None, 0, 0, 0) if kv_lo is None else kv_lo
    l8 = (None, None, 0, None, None, 0) if lo8 is None else (lo8[0], lo8[1], lo8[0].stride(-2), lo8[2], lo8[3], lo8[2].stride(-2))
    a = DenseQkvArgs(C.sizeof(DenseQkvArgs), x_hi.data_ptr(), _ptr(x_lo), x_hi.stride(-2), w.data_ptr(), w.stride(-2), K,
                     cs.data_ptr(), q_hi.data_ptr(), _ptr(q_lo), q_ts, k_arena.data_ptr(), v_arena.data_ptr(), a_bs, a_hs,
                     _ptr(lo[0]), _ptr(lo[1]), lo[2], lo[3], lo[4], B, H, Hkv, D, q_len, past_len, cap, _ptr(past_lens),
                     _ptr(l8[0]), _ptr(l8[1]), l8[2], _ptr(l8[3]), _ptr(l8[4]), l8[5])
    rc = load().pc_gemm_dense_qkv_rope(C.byref(a), current_stream() if stream is None else stream)
    check(rc, "pc_gemm_dense_qkv_rope")


def chain_sync_state(device) -> "torch.Tensor":
    """Zeroed sync state of pc_gemm_chain (owned by the launches afterwards; one per stream of chained launches)."""
    import torch
    return torch.zeros(_dev("pc_chain_sync_words")(), dtype=torch.int32, device=device)


def chain_sync_error(state) -> int:
    """Non-zero: an in-kernel wait of pc_gemm_chain timed out (synchronises)."""
    return int(state[_dev("pc_chain_sync_err_word")()].item())


def gemm_chain(wo_f, attn_hi, attn_lo, attn_width: int, x, M: int, hidden: int, wgu_f, ln2, eps: float, inter: int, act_hi, act_lo,
               wdown_f, sync_state, qkv=None, stream: Optional[int] = None) -> None:
    """o_proj -> gate|up -> down_proj (-> the next layer's q|k|v) as one persistent launch (pc_gemm_chain).
    ``qkv`` = dict(wqkv_f, ln1, cs, q_hi, q_lo, q_ts, k_arena, v_arena, a_bs, a_hs, B, H, Hkv, D, q_len, past_len, cap,
    past_len_dev=None, kv_lo=None (k_lo, v_lo, bs, hs), lo_base=-1) or None."""
    q = qkv or {}
    lo = q.get("kv_lo") or (None, None, 0, 0)
    rc = _dev("pc_gemm_chain")(wo_f.data_ptr(), attn_hi.data_ptr(), attn_lo.data_ptr(), attn_width, x.data_ptr(), M, hidden,
                              wgu_f.data_ptr(), ln2.data_ptr(), eps, inter, act_hi.data_ptr(), act_lo.data_ptr(),
                              wdown_f.data_ptr(), _ptr(q.get("wqkv_f")), _ptr(q.get("ln1")), _ptr(q.get("cs")),
                              _ptr(q.get("q_hi")), _ptr(q.get("q_lo")), q.get("q_ts", 0), _ptr(q.get("k_arena")),
                              _ptr(q.get("v_arena")), q.get("a_bs", 0), q.get("a_hs", 0), q.get("B", 0), q.get("H", 0),
                              q.get("Hkv", 0), q.get("D", 0), q.get("q_len", 0), q.get("past_len", 0), q.get("cap", 0),
                              _ptr(q.get("past_len_dev")), _ptr(lo[0]), _ptr(lo[1]), lo[2], lo[3], q.get("lo_base", -1),
                              sync_state.data_ptr(), current_stream() if stream is None else stream)
    check(rc, "pc_gemm_chain")


def greedy_advance(logits, vocab: int, ids, pos, past, ring, counter, stream: Optional[int] = None) -> None:
    """Tail of a captured greedy decode step (pc_greedy_advance): argmax -> the graph's own input words + the token ring."""
    rc = load().pc_greedy_advance(logits.data_ptr(), vocab, ids.data_ptr(), pos.data_ptr(), past.data_ptr(), ring.data_ptr(),
                                  counter.data_ptr(), ring.numel(), current_stream() if stream is None else stream)
    check(rc, "pc_greedy_advance")


# ---- LLM.int8 (pc_int8.hip) -------------------------------------------------------------------------------------------
LLM_INT8_THRESHOLD = 6.0        # transformers' llm_int8_threshold default, what load_in_8bit=True runs with


def quant_act_i8(x, frag: bool, T: int, K: int, codes, x_scale, flags_set, flags_clear=None, threshold: float = LLM_INT8_THRESHOLD,
                 ldx: Optional[int] = None, stream: Optional[int] = None, codes8=None) -> None:
    """LLM.int8 activation quantiser: fp16 ``x`` (row-major [T, ld] or a fragment plane) -> ``codes`` (fp16, same layout),
    ``x_scale`` [T] fp32, outlier-column flag bytes; ``codes8`` (optional, int8 [ceil(T/16), K/64, 64, 16]): the int8 MFMA's
    operand image of the same codes (``pc_gemm`` ``x_codes8``)."""
    rc = load().pc_quant_act_i8(x.data_ptr(), (0 if frag else x.stride(-2)) if ldx is None else ldx, int(frag), T, K, codes.data_ptr(),
                                x_scale.data_ptr(), flags_set.data_ptr(), _ptr(flags_clear),
                                0 if flags_clear is None else flags_clear.numel(), threshold, _ptr(codes8),
                                current_stream() if stream is None else stream)
    check(rc, "pc_quant_act_i8")


def gemm_skinny_a8c(wf8, w_scale, xq, zeros, x_scale, flags, x_raw, w_codes_t, M: int, N: int, K: int, epilogue: int, y=None,
                    ldy: int = 0, of_hi=None, of_lo=None, stream: Optional[int] = None, codes8=None, row_max_out=None, flags_out=None) -> None:
    """LLM.int8 projection over the code plane ``xq`` with the outlier correction inside the launch (``flags``: >= 16384 bytes).
    ``row_max_out`` / ``flags_out`` (SiLU epilogue, M <= 16): the per-tile row maxima and outlier flags ``gemm_q8``'s down_proj form reads."""
    _gemm(stream, epilogue=epilogue, wf=wf8, w_scale=w_scale, xf_hi=xq, xf_lo=zeros, x_scale=x_scale, x_codes8=codes8, flags=flags, x_raw=x_raw,
          w_codes_t=w_codes_t, ldt=w_codes_t.stride(-2), M=M, N=N, K=K, y=y, ldy=ldy, of_hi=of_hi, of_lo=of_lo, row_max_out=row_max_out,
          flags_out=flags_out, out_threshold=LLM_INT8_THRESHOLD if row_max_out is not None else 0.0)


def gemm_qkv_rope_a8c(wf8_perm, w_scale_perm, xq, zeros, x_scale, flags, x_raw, w_codes_t, row_perm, M, K, cs, q_hi, q_lo, q_ts,
                      k_arena, v_arena, a_bs, a_hs, B, H, Hkv, D, q_len, past_len, cap, past_len_dev=None, kv_lo=None,
                      lo_base: int = -1, stream: Optional[int] = None, codes8=None) -> None:
    _gemm(stream, wf=wf8_perm, w_scale=w_scale_perm, xf_hi=xq, xf_lo=zeros, x_scale=x_scale, x_codes8=codes8, flags=flags, x_raw=x_raw,
          w_codes_t=w_codes_t, ldt=w_codes_t.stride(-2), row_perm=row_perm, M=M, K=K,
          **_qkv_fields(cs, q_hi, q_lo, q_ts, k_arena, v_arena, a_bs, a_hs, B, H, Hkv, D, q_len, past_len, cap, past_len_dev, kv_lo, lo_base))


def gemm_part(wf, part_o, part_ml, nsplit: int, H: int, D: int, N: int, y, stream: Optional[int] = None) -> None:
    """o_proj + residual of a one-row step on the split-KV partials ``attn_fwd(defer_merge=True)`` left (``nsplit`` per head)."""
    rc = load().pc_gemm_part(wf.data_ptr(), part_o.data_ptr(), part_ml.data_ptr(), nsplit, H, D, N, y.data_ptr(),
                             current_stream() if stream is None else stream)
    check(rc, "pc_gemm_part")


def gemm_part_rows(wf, part_o, part_ml, nsplit: int, H: int, D: int, N: int, M: int, y, ldy: int, kslices: int, scratch, counters,
                   rows_dev=None, stream: Optional[int] = None) -> None:
    """o_proj + residual of a 1..16-row step on the split-KV partials ``attn_fwd(defer_merge=True)`` left for ``q_len = M`` rows: K cut
    into ``kslices`` workgroup slices, reduced inside the launch (``pc_gemm_part_rows``; -DPC_DEV_SWEEPS builds only: measured slower than the merge launch + o_proj)."""
    rc = _dev("pc_gemm_part_rows")(wf.data_ptr(), part_o.data_ptr(), part_ml.data_ptr(), nsplit, H, D, N, M,
                                  None if rows_dev is None else rows_dev.data_ptr(), y.data_ptr(), ldy, kslices, scratch.data_ptr(),
                                  scratch.numel() * scratch.element_size(), counters.data_ptr(),
                                  current_stream() if stream is None else stream)
    check(rc, "pc_gemm_part_rows")


def gemm_q8(stream=None, **f) -> None:
    """One ``pc_gemm_q8`` call (LLM.int8 projection of <= 16 rows, activation quantiser inside the launch): keyword = field of
    ``pc_gemm_q8_args``; tensors become pointers, everything else 0 / NULL.  ``qkv=dict(...)`` takes ``_qkv_fields``' arguments."""
    a = GemmQ8Args()
    a.struct_bytes = C.sizeof(GemmQ8Args)
    a.kslices, a.lo_base, a.threshold = 1, -1, LLM_INT8_THRESHOLD
    for k, v in f.items():
        if v is None:
            continue
        setattr(a, k, v.data_ptr() if hasattr(v, "data_ptr") else v)
    if f.get("w_codes_t") is not None and "ldt" not in f:
        a.ldt = f["w_codes_t"].stride(-2)
    rc = load().pc_gemm_q8(C.byref(a), current_stream() if stream is None else stream)
    check(rc, "pc_gemm_q8")


def rmsnorm_quant_i8(x, norm_weight, eps: float, T: int, hidden: int, x_hi, codes, x_scale, flags_set, flags_clear=None,
                     threshold: float = LLM_INT8_THRESHOLD, stream: Optional[int] = None, codes8=None) -> None:
    """RMSNorm + LLM.int8 activation quantiser in one launch (fragment planes, T <= 64): pc_rmsnorm_frag + pc_quant_act_i8."""
    rc = load().pc_rmsnorm_quant_i8(x.data_ptr(), norm_weight.data_ptr(), eps, T, hidden, x_hi.data_ptr(), codes.data_ptr(),
                                    x_scale.data_ptr(), flags_set.data_ptr(), _ptr(flags_clear),
                                    0 if flags_clear is None else flags_clear.numel(), threshold, _ptr(codes8),
                                    current_stream() if stream is None else stream)
    check(rc, "pc_rmsnorm_quant_i8")


def outlier_corr(flags, K: int, x, codes, frag: bool, x_scale, w_codes_t, w_scale, row_perm, T: int, N: int, corr, has,
                 ldx: Optional[int] = None, stream: Optional[int] = None) -> None:
    """``w_codes_t``: the int8 weight codes transposed, [K, N] (``q.t().contiguous()``)."""
    rc = load().pc_outlier_corr(flags.data_ptr(), K, x.data_ptr(), codes.data_ptr(), (0 if frag else x.stride(-2)) if ldx is None else ldx,
                                int(frag), x_scale.data_ptr(), w_codes_t.data_ptr(), w_codes_t.stride(-2), w_scale.data_ptr(),
                                _ptr(row_perm), T, N, corr.data_ptr(), corr.stride(-2), has.data_ptr(),
                                current_stream() if stream is None else stream)
    check(rc, "pc_outlier_corr")


def gemm_skinny_a8(wf8, w_scale, xq, zeros, x_scale, corr, has, M: int, N: int, K: int, epilogue: int, y=None, ldy: int = 0,
                   of_hi=None, of_lo=None, stream: Optional[int] = None, codes8=None) -> None:
    _gemm(stream, epilogue=epilogue, wf=wf8, w_scale=w_scale, xf_hi=xq, xf_lo=zeros, x_scale=x_scale, x_codes8=codes8, corr=corr, ldc=corr.stride(-2),
          corr_has=has, M=M, N=N, K=K, y=y, ldy=ldy, of_hi=of_hi, of_lo=of_lo)


def gemm_qkv_rope_a8(wf8, w_scale, xq, zeros, x_scale, corr, has, M: int, K: int, cs, q_hi, q_lo, q_ts, k_arena, v_arena, a_bs, a_hs,
                     B, H, Hkv, D, q_len, past_len, cap, past_len_dev=None, kv_lo=None, lo_base: int = -1,
                     stream: Optional[int] = None, codes8=None) -> None:
    _gemm(stream, wf=wf8, w_scale=w_scale, xf_hi=xq, xf_lo=zeros, x_scale=x_scale, x_codes8=codes8, corr=corr, ldc=corr.stride(-2), corr_has=has, M=M, K=K,
          **_qkv_fields(cs, q_hi, q_lo, q_ts, k_arena, v_arena, a_bs, a_hs, B, H, Hkv, D, q_len, past_len, cap, past_len_dev,
                        kv_lo[:4] if kv_lo else None, lo_base))


def gemm_dense_a8(xq, w_codes, w_scale, x_scale, corr, has, M: int, N: int, K: int, epilogue: int, y=None, out_hi=None, out_lo=None,
                  stream: Optional[int] = None) -> None:
    rc = load().pc_gemm_dense_a8(xq.data_ptr(), xq.stride(-2), w_codes.data_ptr(), w_codes.stride(-2), w_scale.data_ptr(),
                                 x_scale.data_ptr(), corr.data_ptr(), corr.stride(-2), has.data_ptr(), M, N, K, epilogue, _ptr(y),
                                 0 if y is None else y.stride(-2), _ptr(out_hi), _ptr(out_lo),
                                 0 if out_hi is None else out_hi.stride(-2), current_stream() if stream is None else stream)
    check(rc, "pc_gemm_dense_a8")
