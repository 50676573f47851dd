"""Stress tests of the in-launch hand-offs on the default path: the K reduction inside
gemm_skinny_ks_kernel (down_proj of the timed step: partial tiles written through, arrival counter, last arriver adds them in
slice order) and the split-KV merge inside attn_small_kernel (pc_attn `counters`).  Both publish with agent-scope atomic stores
+ `s_waitcnt vmcnt(0)` + a relaxed arrival counter (pc_gemm_ks.hip: the hand-off contract).  A lost or stale partial would show
as a wrong word once in many launches, so: 1e5 launches each (2e4 for the int8 down_proj form of pc_gemm_q8, which carries the same hand-off), under UNEVEN load (a copy stream hammering HBM and the L2s next to
them), the consumer's caches warm, EVERY output word compared with the first launch's."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
LAUNCHES = int(os.environ.get("PC_HANDOFF_LAUNCHES", "100000"))     # (the pair of default-form tests takes ~4 s at 1e5 on MI355X)
PER_GRAPH = 50


@pytest.fixture(params=[False, True], ids=["relaxed+vmcnt0", "formal_acq_rel"])
def formal(request):
    """Both forms of the arrival live in the one binary; PC_FORMAL_HANDOFF=1 in the environment selects the acq_rel fetch-add (the
    C++-memory-model form) at launch time -- the captured graphs below keep what their capture read (the
    formal form runs in every pass of the suite, through the same stress as the default)."""
    old = os.environ.get("PC_FORMAL_HANDOFF")
    os.environ["PC_FORMAL_HANDOFF"] = "1" if request.param else "0"
    yield request.param
    if old is None:
        os.environ.pop("PC_FORMAL_HANDOFF", None)
    else:
        os.environ["PC_FORMAL_HANDOFF"] = old


def _n():
    from promptcache_amd import _native
    return _native


class _Load:
    """Uneven background load: big device-to-device copies on a side stream for as long as the context is open."""

    def __enter__(self):
        self.s = torch.cuda.Stream()
        self.a = torch.empty(192 << 20, dtype=torch.uint8, device=DEV)
        self.b = torch.empty_like(self.a)
        self.n = 0
        return self

    def pump(self, k=3):
        with torch.cuda.stream(self.s):
            for i in range(k):
                lo = (self.n % 3) * (64 << 20)
                self.b[lo:lo + (64 << 20) - 4096 * (self.n % 7)].copy_(self.a[lo:lo + (64 << 20) - 4096 * (self.n % 7)])
                self.n += 1

    def __exit__(self, *exc):
        self.s.synchronize()
        return False


def _replay_and_count(graph, outs, ref, load):
    """Replay `graph` (PER_GRAPH launches into outs[j]) LAUNCHES / PER_GRAPH times; -> number of mismatching words, on the device."""
    bad = torch.zeros((), dtype=torch.int64, device=DEV)
    for r in range(LAUNCHES // PER_GRAPH):
        if r % 4 == 0:
            load.pump()
        graph.replay()
        bad += (outs.view(torch.int32) != ref.view(torch.int32)).sum()
    torch.cuda.synchronize()
    return int(bad)


def test_k_reduction_inside_the_launch_1e5_launches_bit_stable(formal):
    n = _n()
    rng = np.random.default_rng(41)
    M, N, K, tiles, slices = 12, 4096, 11008, 2, 2                      # down_proj of the timed step as it is dispatched
    w = torch.from_numpy((0.05 * rng.standard_normal((N, K), dtype=np.float32)).astype(np.float16)).to(DEV)
    x = torch.from_numpy(rng.standard_normal((M, K), dtype=np.float32)).to(DEV)
    wf = n.to_weight_frags(w)
    hi, lo = n.to_act_frags(x)
    y0 = torch.from_numpy(rng.standard_normal((M, N), dtype=np.float32)).to(DEV)
    scratch = torch.empty(n.gemm_skinny_ks_scratch_bytes(N, 8) // 4, dtype=torch.float32, device=DEV)
    counters = torch.zeros(N // 16, dtype=torch.int32, device=DEV)
    outs = torch.empty((PER_GRAPH, M, N), dtype=torch.float32, device=DEV)

    def body():
        for j in range(PER_GRAPH):
            outs[j].copy_(y0)
            n.gemm_skinny_ks(wf, hi, lo, M, N, K, outs[j], N, slices, tiles, scratch, counters)

    body()
    torch.cuda.synchronize()
    ref = outs[0:1].clone()
    exact = (x.double() @ w.double().t()).float() + y0
    assert (ref[0] - exact).abs().max().item() < 2e-4 * float(exact.abs().max())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    with _Load() as load:
        bad = _replay_and_count(g, outs, ref.expand_as(outs), load)
    assert bad == 0, f"{bad} words differed over {LAUNCHES} launches"
    assert int(counters.abs().sum()) == 0                             # every launch left its arrival counters at zero
    if formal:
        # the two forms of the arrival give the same bits (the partials are added in slice order either way)
        os.environ["PC_FORMAL_HANDOFF"] = "0"
        outs[0].copy_(y0)
        n.gemm_skinny_ks(wf, hi, lo, M, N, K, outs[0], N, slices, tiles, scratch, counters)
        torch.cuda.synchronize()
        assert torch.equal(outs[0].view(torch.int32), ref[0].view(torch.int32))


def test_int8_down_proj_k_reduction_inside_the_launch_bit_stable(formal):
    """pc_gemm_q8's F form (down_proj of the LLM.int8 cached step as it is dispatched: 4 tiles x 4 slices, the quantiser and the outlier
    correction inside, round 5): the same hand-off as gemm_skinny_ks_kernel in its own kernel -- same stress, a fifth of the launches."""
    n = _n()
    rng = np.random.default_rng(43)
    M, N, K, tiles, slices = 12, 4096, 11008, 4, 4
    x = np.clip(rng.standard_normal((M, K)).astype(np.float32) * 1.5, -5.9, 5.9)
    cols = rng.permutation(K)[:200]
    x[rng.integers(0, M, size=200), cols] = 9.0                         # flagged columns in every K slice: the correction runs
    x = x.astype(np.float16).astype(np.float32)
    w = (0.03 * rng.standard_normal((N, K))).astype(np.float32)
    q, sc = n.quantize_rows_int8(torch.from_numpy(w).to(DEV))
    wf8, qt = n.to_weight_frags_i8(q), q.t().contiguous()
    hi, _ = n.to_act_frags(torch.from_numpy(x).to(DEV))
    outl = np.abs(x) >= 6.0
    pm = np.zeros((K // 16, 16), dtype=np.float32)
    pm[:, :M] = np.where(outl, 0.0, np.abs(x)).reshape(M, K // 16, 16).max(axis=2).T
    pmd = torch.from_numpy(pm).to(DEV)
    fl = torch.zeros(16384, dtype=torch.uint8, device=DEV)
    fl[:K] = torch.from_numpy(outl.any(axis=0).astype(np.uint8)).to(DEV)
    y0 = torch.from_numpy(rng.standard_normal((M, N), dtype=np.float32)).to(DEV)
    scratch = torch.empty(n.gemm_skinny_ks_scratch_bytes(N, 8) // 4, dtype=torch.float32, device=DEV)
    counters = torch.zeros(N // 16, dtype=torch.int32, device=DEV)
    outs = torch.empty((PER_GRAPH, M, N), dtype=torch.float32, device=DEV)

    def one(dst):
        dst.copy_(y0)
        n.gemm_q8(epilogue=n.EPI_ADD, wf=wf8, w_scale=sc, w_codes_t=qt, xf_hi=hi, row_max=pmd, row_max_units=K // 16, flags_in=fl, M=M, N=N, K=K,
                  y=dst, ldy=N, ks_tiles=tiles, kslices=slices, ks_scratch=scratch, ks_scratch_bytes=scratch.numel() * 4, ks_counters=counters)

    def body():
        for j in range(PER_GRAPH):
            one(outs[j])

    body()
    torch.cuda.synchronize()
    ref = outs[0:1].clone()
    from oracle import int8_oracle as io
    from oracle import llmint8_oracle as lo
    qo, so = io.quantize_rows_int8(w)
    exact = y0.cpu().numpy() + lo.linear(x, qo, so)
    assert np.abs(ref[0].cpu().numpy() - exact).max() < 3e-5 * max(1.0, np.abs(exact).max())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    global LAUNCHES
    keep, LAUNCHES = LAUNCHES, max(PER_GRAPH, LAUNCHES // 5)
    try:
        with _Load() as load:
            bad = _replay_and_count(g, outs, ref.expand_as(outs), load)
    finally:
        LAUNCHES = keep
    assert bad == 0, f"{bad} words differed"
    assert int(counters.abs().sum()) == 0
    if formal:
        os.environ["PC_FORMAL_HANDOFF"] = "0"
        one(outs[0])
        torch.cuda.synchronize()
        assert torch.equal(outs[0].view(torch.int32), ref[0].view(torch.int32))


def test_split_kv_merge_inside_the_launch_1e5_launches_bit_stable(formal):
    n = _n()
    rng = np.random.default_rng(42)
    B, H, Hkv, D, q_len, past = 1, 32, 32, 128, 12, 1725               # one layer of the persona cached prefill
    cap = past + q_len + 3
    f16 = lambda a: torch.from_numpy(a.astype(np.float16)).to(DEV)     # noqa: E731
    q32 = rng.standard_normal((B, q_len, H, D), dtype=np.float32)
    q = f16(q32)
    ql = f16(q32 - q.float().cpu().numpy())
    k = f16(rng.standard_normal((B, Hkv, cap, D), dtype=np.float32))
    v = f16(rng.standard_normal((B, Hkv, cap, D), dtype=np.float32))
    klo = f16(1e-4 * rng.standard_normal((B, Hkv, q_len, D), dtype=np.float32))
    vlo = f16(1e-4 * rng.standard_normal((B, Hkv, q_len, D), dtype=np.float32))
    kv_lo = (klo, vlo, Hkv * q_len * D, q_len * D, -1)
    ws = torch.empty(max(n.attn_workspace_bytes(B, H, D, q_len, past + q_len), 4) // 4, dtype=torch.float32, device=DEV)
    counters = torch.zeros(B * H, dtype=torch.int32, device=DEV)
    mt = (B * q_len + 15) // 16
    outs = torch.zeros((PER_GRAPH, 2, mt, H * D // 32, 64, 8), dtype=torch.float16, device=DEV)
    scale = 1.0 / np.sqrt(D)

    def body():
        for j in range(PER_GRAPH):
            n.attn_fwd(q, q_len * H * D, H * D, k, v, Hkv * cap * D, cap * D, None, 0, 0, B, H, Hkv, D, q_len, past, scale, ws,
                       out_frag=(outs[j, 0], outs[j, 1]), q_lo=ql, kv_lo=kv_lo, counters=counters)

    body()
    torch.cuda.synchronize()
    ref = outs[0:1].clone()
    # the two-launch form (attn_combine_kernel) merges the same partials with the same arithmetic
    two = torch.zeros_like(ref)
    n.attn_fwd(q, q_len * H * D, H * D, k, v, Hkv * cap * D, cap * D, None, 0, 0, B, H, Hkv, D, q_len, past, scale, ws,
               out_frag=(two[0, 0], two[0, 1]), q_lo=ql, kv_lo=kv_lo)
    torch.cuda.synchronize()
    live = n.from_act_frags(ref[0, 0], q_len).double() + n.from_act_frags(ref[0, 1], q_len).double()
    other = n.from_act_frags(two[0, 0], q_len).double() + n.from_act_frags(two[0, 1], q_len).double()
    # (the two kernels may contract a * b + c differently: fp32 round-off, see test_attn_single_launch_merge_equals_...)
    assert torch.isfinite(live).all() and float((live - other).abs().max()) <= 4e-7 * float(other.abs().max()) + 1e-9
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    with _Load() as load:
        bad = _replay_and_count(g, outs, ref.expand_as(outs), load)
    assert bad == 0, f"{bad} words differed over {LAUNCHES} launches"
    assert int(counters.abs().sum()) == 0
