#ifdef PC_DEV_SWEEPS      // (dev builds only: see pc_dev.h)
#include "pc_dev.h"
// pc_gemm_chain: the projections between two attention calls of a <= 16-row forward as ONE persistent launch (opt-in,
// PC_CHAIN=1).  Kernel templates shared with the stand-alone launches: pc_gemm_skinny.h.
#include "pc_gemm_skinny.h"

using namespace pcg;

namespace {

// ---------------------------------------------------------------------------------------------------
// pc_gemm_chain: the projections between two attention calls of a <= 16-row forward as ONE persistent launch --
//     phase 0  x += attn @ Wo^T                    (EPI_ADD)              llama2.py:405, :638
//     phase 1  act = silu(gate(n2(x))) * up(n2(x)) (RMSNorm folded, SiLU)  llama2.py:640-643, :242
//     phase 2  x += act @ Wdown^T                  (EPI_ADD)              llama2.py:242, :644
//     phase 3  q|k|v of the NEXT layer: n1(x) @ Wqkv^T, RoPE, in-place KV append (optional)  llama2.py:345-364
// Every phase is the body of the stand-alone launch (gemm_skinny_body: same tiles, same K split, same reduction order ->
// bit-identical results); between phases the grid meets at a barrier.  What the fusion buys is not the barrier -- a grid
// barrier costs about what a kernel boundary costs -- but what happens AROUND it: a stand-alone launch spends ~4 us outside
// its stream (launch, first-load latency, reduction + epilogue: o_proj takes 9.4 us for 5.3 us of HBM time); here each
// wave issues the first k-block of the next phase's weights as soon as its own K loop is done (weights depend on no
// activation), so the HBM stream keeps running through reduction, epilogue and barrier.
// Grid: one workgroup per CU, all co-resident (512 threads, <= 256 VGPRs, < 160 KB LDS: the launcher takes
// min(CUs, 256)); phase blocks are dealt round-robin (bx = wg, wg + grid, ...).
// Sync state (pc_chain_sync_words() uint32 words, zeroed once by the caller): a launch-epoch word, 8 sharded arrival
// counters + a top counter (all monotonic across launches: no memset between graph replays), 8 generation words the
// workgroups poll, an error word.  Hand-off recipe of the guide (G16): every storing wave drains its stores, workgroup
// barrier, ONE lane does the agent-scope release (L2 write-back) + arrive, polls relaxed with s_sleep, ONE agent-scope
// acquire, workgroup barrier, then plain vector loads.  Every spin is bounded: on a timeout the error word is set and the
// launch runs to its end (results are then garbage, pc_chain_sync_error reports it; the state must be re-zeroed).
constexpr int kSyStride = 32;                 // uint32 words between hot words (128 B apart)
enum { SY_BASE = 0, SY_SHARD = 1, SY_TOP = 9, SY_GEN = 10, SY_ERR = 18, SY_SLOTS = 19 };
constexpr uint32_t kSpinLimit = 1u << 21;

constexpr int kTraceSlots = 16;              // uint64 timestamps per workgroup (PC_CHAIN_TRACE=1; tools/chain_trace.py)
struct ChainParams {
    GemmParams ph[4];
    int32_t nblk[4];
    int32_t nphases;
    uint32_t* sync;
    unsigned long long* trace;                // null unless tracing: [grid][kTraceSlots] wall_clock64() stamps
    int32_t pf_mode;                          // when the next phase's first block is fetched: 0 as early as possible (after the
                                              // K loop / the store drain / the arrive), 1 after the arrive, 2 after the release
};

typedef __attribute__((address_space(1))) uint32_t gu32;

struct GridSync {
    gu32* st; uint32_t base; int wg, nwg;
    unsigned long long* tr;
    int pf_mode;
    __device__ __forceinline__ void stamp(int slot) const { if (tr) tr[wg * kTraceSlots + slot] = wall_clock64(); }
    __device__ __forceinline__ void init(uint32_t* s) {
        st = (gu32*)s; wg = blockIdx.x; nwg = gridDim.x;
        base = __hip_atomic_load(st + SY_BASE * kSyStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // sync-wave half 1: publish this workgroup's stores and arrive at seam k (1-based inside the launch)
    __device__ __forceinline__ void arrive(uint32_t k, bool last_seam) const {
        const uint32_t target = base + k;
        const int shard = wg & 7;
        const uint32_t nshard = (uint32_t)((nwg - shard + 7) >> 3), nsh = (uint32_t)(nwg < 8 ? nwg : 8);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t old = __hip_atomic_fetch_add(st + (SY_SHARD + shard) * kSyStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == target * nshard) {                 // last of its shard at this seam
            const uint32_t o2 = __hip_atomic_fetch_add(st + SY_TOP * kSyStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (o2 + 1 == target * nsh) {                 // last of all: release everyone
                if (last_seam) __hip_atomic_store(st + SY_BASE * kSyStride, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (uint32_t j = 0; j < nsh; ++j)
                    __hip_atomic_store(st + (SY_GEN + j) * kSyStride, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    // sync-wave half 2: wait until every workgroup has arrived at seam k, then acquire
    __device__ __forceinline__ void wait(uint32_t k) const {
        const uint32_t target = base + k;
        gu32* gen = st + (SY_GEN + (wg & 7)) * kSyStride;
        uint32_t spins = 0;
        while ((int32_t)(__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > kSpinLimit) {
                __hip_atomic_store(st + SY_ERR * kSyStride, 0x100u + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
};

constexpr int kSyncWave = kWaves - 1;          // the wave that arrives / polls (it never runs an epilogue)

// One phase of the chain for this workgroup.  PF: the next phase's first-block prefetch as a callable (wave-uniform
// caller decides when): waves without stores call it right after their K loop, storing waves after their stores have
// drained, the sync wave after it has arrived.
template <int T, int EPI, int U, bool NORM, bool HAVE_PRE, class PF>
__device__ __forceinline__ void chain_phase(const GemmParams& p, int nblk, const GridSync& gs, uint32_t seam, bool last_seam,
                                            const h8 (*pre)[(EPI == EPI_SILU) ? 2 * T : T], PF prefetch_next,
                                            float* red, float (*ssl)[32], _Float16* gam_lds) {
    static_assert(T <= 4, "one reduction round, and the sync wave never stores");
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool storing = wave < T;                        // single-round launches (MT = 1): output item w is wave w's
    const bool early_wave = !storing && wave != kSyncWave && gs.pf_mode == 0;
    const int wg = gs.wg, nwg = gs.nwg;
    const int ts = seam == 0 ? 12 : 4 * ((int)seam - 1);              // trace slots of this phase (seam 0 = the last one)
    if (threadIdx.x == 0) gs.stamp(ts);                   // phase start
    if (wg < nblk) {
        // the workgroup's first block takes the prefetched fragments (dead afterwards); further rounds are plain
        {
            const bool lastb = wg + nwg >= nblk;
            auto early = [&]() { if (lastb && early_wave) prefetch_next(); };
            if constexpr (HAVE_PRE) gemm_skinny_body<1, T, EPI, true, U, NORM, false, true>(p, wg, 0, red, ssl, pre, early, gam_lds);
            else gemm_skinny_body<1, T, EPI, true, U, NORM, false, false>(p, wg, 0, red, ssl, nullptr, early, gam_lds);
        }
        for (int bx = wg + nwg; bx < nblk; bx += nwg) {
            const bool lastb = bx + nwg >= nblk;
            auto early = [&]() { if (lastb && early_wave) prefetch_next(); };
            lds_barrier();                                // the reduction buffer of the previous block is free
            gemm_skinny_body<1, T, EPI, true, U, NORM, false, false>(p, bx, 0, red, ssl, nullptr, early, gam_lds);
        }
    } else if (early_wave) {
        prefetch_next();                                  // no block in this phase: still prefetch the next one
    }
    if (threadIdx.x == 0) gs.stamp(ts + 1);               // bodies done (wave 0: after its epilogue stores were issued)
    if (seam == 0) return;                                // last phase of the launch: the kernel boundary publishes
    if (storing) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's epilogue stores have reached L2
        if (gs.pf_mode == 0) prefetch_next();
    }
    lds_barrier();                                        // (the prefetch loads stay in flight across both barriers)
    if (wave == kSyncWave) {
        if ((threadIdx.x & 63) == 0) { gs.stamp(ts + 2); gs.arrive(seam, last_seam); }     // workgroup complete -> arrive
        if (gs.pf_mode <= 1) prefetch_next();
        if ((threadIdx.x & 63) == 0) { gs.wait(seam); gs.stamp(ts + 3); }                   // released
        if (gs.pf_mode == 2) prefetch_next();
    } else if (gs.pf_mode == 1) {
        prefetch_next();
    }
    lds_barrier();
    if (gs.pf_mode == 2 && wave != kSyncWave) prefetch_next();
}

// Tile widths per phase: (TO, TG, TD, TQ) tiles / gate-up pairs per workgroup, chosen by the launcher like choose_T does.
template <int TO, int TG, int TD, int TQ, int UO, int UG, int UD, int UQ>
__global__ __launch_bounds__(kThreads) void gemm_chain_kernel(const ChainParams c) {
    __shared__ __attribute__((aligned(16))) float red[kWaves * 8 * 64 * 4];      // 64 KiB: the widest phase's reduction buffer
    __shared__ float ssl[kWaves][32];
    __shared__ __attribute__((aligned(16))) _Float16 gam_lds[kWaves * kGamHalfs];
    GridSync gs;
    gs.init(c.sync);
    gs.tr = c.trace;
    gs.pf_mode = c.pf_mode;
    const int wg = blockIdx.x;
    const bool qkv = c.nphases == 4;
    h8 pre_g[UG][2 * TG], pre_d[UD][TD], pre_q[UQ][TQ];
    auto pf_g = [&]() { if (wg < c.nblk[1]) prefetch_first_block<TG, EPI_SILU, UG>(c.ph[1], wg, pre_g); };
    auto pf_d = [&]() { if (wg < c.nblk[2]) prefetch_first_block<TD, EPI_ADD, UD>(c.ph[2], wg, pre_d); };
    auto pf_q = [&]() { if (qkv && wg < c.nblk[3]) prefetch_first_block<TQ, EPI_ROPE, UQ>(c.ph[3], wg, pre_q); };
    auto pf_none = [&]() {};
    chain_phase<TO, EPI_ADD, UO, false, false>(c.ph[0], c.nblk[0], gs, 1, false, nullptr, pf_g, red, ssl, gam_lds);
    chain_phase<TG, EPI_SILU, UG, true, true>(c.ph[1], c.nblk[1], gs, 2, !qkv, pre_g, pf_d, red, ssl, gam_lds);
    if (qkv) {
        chain_phase<TD, EPI_ADD, UD, false, true>(c.ph[2], c.nblk[2], gs, 3, true, pre_d, pf_q, red, ssl, gam_lds);
        chain_phase<TQ, EPI_ROPE, UQ, true, true>(c.ph[3], c.nblk[3], gs, 0, false, pre_q, pf_none, red, ssl, gam_lds);
    } else {
        chain_phase<TD, EPI_ADD, UD, false, true>(c.ph[2], c.nblk[2], gs, 0, false, pre_d, pf_none, red, ssl, gam_lds);
    }
}

}  // namespace

// ---- pc_gemm_chain: o_proj -> gate|up -> down_proj (-> the next layer's q|k|v) as one persistent launch ---------------
PC_EXPORT int32_t pc_chain_sync_words(void) { return SY_SLOTS * kSyStride + 256 * kTraceSlots * 2; }   // (+ the trace area)
PC_EXPORT int32_t pc_chain_sync_err_word(void) { return SY_ERR * kSyStride; }

namespace {
template <int TO, int TG, int TD, int TQ, int UO, int UG, int UD, int UQ>
int launch_chain(const ChainParams& c, int grid, hipStream_t s) {
    hipLaunchKernelGGL((gemm_chain_kernel<TO, TG, TD, TQ, UO, UG, UD, UQ>), dim3(grid), dim3(kThreads), 0, s, c);
    return pc_check_launch("gemm_chain_kernel");
}
int chain_grid() {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        return n;
    }();
    return cus < 256 ? cus : 256;
}
}  // namespace

PC_EXPORT int pc_gemm_chain(const void* wo_f, const void* attn_hi, const void* attn_lo, int32_t attn_width, float* x,
                            int32_t M, int32_t hidden, const void* wgu_f, const void* ln2_weight, float eps, int32_t inter,
                            void* act_hi, void* act_lo, const void* wdown_f, const void* wqkv_f_next,
                            const void* ln1_weight_next, const float* cs, void* q_hi, void* q_lo, int64_t q_token_stride,
                            void* k_arena, void* v_arena, int64_t arena_batch_stride, int64_t arena_head_stride, int32_t B,
                            int32_t H, int32_t Hkv, int32_t D, int32_t q_len, int32_t past_len, int32_t cap,
                            const int32_t* past_len_dev, void* k_lo, void* v_lo, int64_t lo_batch_stride,
                            int64_t lo_head_stride, int32_t lo_base, void* sync_state, void* stream) {
    PC_REQUIRE(wo_f && attn_hi && attn_lo && x && wgu_f && ln2_weight && act_hi && act_lo && wdown_f && sync_state, PC_ERR_ARG,
               "pc_gemm_chain: null pointer");
    PC_REQUIRE(M > 0 && M <= 16, PC_ERR_ARG, "pc_gemm_chain: M=%d outside 1..16", M);
    PC_REQUIRE(hidden <= 32 * kGamSteps * kWaves, PC_ERR_ARG, "pc_gemm_chain: hidden %d > 16384", hidden);
    PC_REQUIRE(hidden > 0 && hidden % 32 == 0 && inter > 0 && inter % 32 == 0 && attn_width > 0 && attn_width % 32 == 0, PC_ERR_ARG,
               "pc_gemm_chain: hidden, inter and the attention width must be multiples of 32");
    ChainParams c;
    memset(&c, 0, sizeof(c));
    c.sync = (uint32_t*)sync_state;
    static const bool trace = [] { const char* e = getenv("PC_CHAIN_TRACE"); return e && atoi(e) != 0; }();
    if (trace) c.trace = (unsigned long long*)((uint32_t*)sync_state + SY_SLOTS * kSyStride);
    static const int pf_mode = [] { const char* e = getenv("PC_CHAIN_PF"); return e ? atoi(e) : 0; }();
    c.pf_mode = pf_mode;
    // phase 0: x += attn @ Wo^T
    GemmParams& o = c.ph[0];
    o.wf = (const _Float16*)wo_f; o.xf_hi = (const _Float16*)attn_hi; o.xf_lo = (const _Float16*)attn_lo;
    o.y = x; o.ldy = hidden; o.M = M; o.ntiles = hidden / 16; o.KS = attn_width / 32; o.kslices = 1; o.slab_stride = (int64_t)M * hidden;
    // phase 1: act = silu(gate(n2(x))) * up(n2(x))
    GemmParams& g = c.ph[1];
    g.wf = (const _Float16*)wgu_f; g.xn = x; g.gamma = (const _Float16*)ln2_weight; g.eps = eps;
    g.of_hi = (_Float16*)act_hi; g.of_lo = (_Float16*)act_lo; g.M = M; g.ntiles = 2 * inter / 16; g.npairs = inter / 16;
    g.KS = hidden / 32; g.KSo = inter / 32; g.kslices = 1;
    // phase 2: x += act @ Wdown^T
    GemmParams& d = c.ph[2];
    d.wf = (const _Float16*)wdown_f; d.xf_hi = (const _Float16*)act_hi; d.xf_lo = (const _Float16*)act_lo;
    d.y = x; d.ldy = hidden; d.M = M; d.ntiles = hidden / 16; d.KS = inter / 32; d.kslices = 1; d.slab_stride = (int64_t)M * hidden;
    c.nphases = 3;
    int tq = 1;
    if (wqkv_f_next) {
        const int N = (H + 2 * Hkv) * D;
        PC_REQUIRE(ln1_weight_next && cs && q_hi && q_lo && k_arena && v_arena, PC_ERR_ARG, "pc_gemm_chain: null q|k|v pointer");
        PC_REQUIRE(M == B * q_len && D % 16 == 0 && H > 0 && Hkv > 0, PC_ERR_ARG, "pc_gemm_chain: bad q|k|v shape");
        PC_REQUIRE((int64_t)past_len + q_len <= cap, PC_ERR_BOUNDS, "pc_gemm_chain: past_len %d + q_len %d exceeds arena rows %d",
                   past_len, q_len, cap);
        PC_REQUIRE(q_token_stride % 4 == 0 && arena_head_stride % 4 == 0, PC_ERR_ARG, "pc_gemm_chain: strides must keep 8-byte alignment");
        PC_REQUIRE((k_lo == nullptr) == (v_lo == nullptr) && (!k_lo || lo_head_stride % 4 == 0), PC_ERR_ARG,
                   "pc_gemm_chain: k_lo / v_lo go together, strides must keep 8-byte alignment");
        PC_REQUIRE(lo_base >= -2 && (lo_base != -2 || past_len_dev) && (lo_base < 0 || lo_base <= past_len), PC_ERR_ARG,
                   "pc_gemm_chain: lo_base must be -1, -2 (past_len_dev[1]) or lie in [0, past_len]");
        GemmParams& q = c.ph[3];
        q.wf = (const _Float16*)wqkv_f_next; q.xn = x; q.gamma = (const _Float16*)ln1_weight_next; q.eps = eps;
        q.M = M; q.ntiles = N / 16; q.KS = hidden / 32; q.kslices = 1;
        q.rope.cs = (const float2*)cs; q.rope.q_hi = (_Float16*)q_hi; q.rope.q_lo = (_Float16*)q_lo; q.rope.q_ts = q_token_stride;
        q.rope.k_arena = (_Float16*)k_arena; q.rope.v_arena = (_Float16*)v_arena; q.rope.a_bs = arena_batch_stride;
        q.rope.a_hs = arena_head_stride; q.rope.past_len_dev = past_len_dev;
        q.rope.k_lo = (_Float16*)k_lo; q.rope.v_lo = (_Float16*)v_lo; q.rope.lo_bs = lo_batch_stride; q.rope.lo_hs = lo_head_stride;
        q.rope.lo_base = lo_base; q.rope.H = H; q.rope.Hkv = Hkv; q.rope.D = D; q.rope.q_len = q_len; q.rope.past_len = past_len;
        c.nphases = 4;
        tq = choose_T(q.ntiles);
        c.nblk[3] = pc_ceil_div(q.ntiles, tq);
    }
    const int to = choose_T(o.ntiles), tg = choose_T(g.npairs);
    c.nblk[0] = pc_ceil_div(o.ntiles, to);
    c.nblk[1] = pc_ceil_div(g.npairs, tg);
    c.nblk[2] = pc_ceil_div(d.ntiles, to);
    const int grid = chain_grid();
    PC_REQUIRE(grid >= 8, PC_ERR_ARG, "pc_gemm_chain: could not query the CU count");
    hipStream_t s = (hipStream_t)stream;
    // every wave's K range (the last one is the shortest) must hold one full first block of each prefetched phase
    auto last_range = [](int KS) { return KS - (kWaves - 1) * pc_ceil_div(KS, kWaves); };
    const int lg = last_range(g.KS), ld = last_range(d.KS), lq = c.nphases == 4 ? last_range(c.ph[3].KS) : 1 << 20;
    // (k-steps per block 8 / 4 / 8 / 4: 16 for the two N = hidden phases spills ~200 VGPRs next to the prefetch registers)
    if (to == 1 && tg == 3 && (c.nphases == 3 || tq == 3) && lg >= 4 && ld >= 8 && lq >= 4)
        return launch_chain<1, 3, 1, 3, 8, 4, 8, 4>(c, grid, s);        // 7b shapes (hidden 4096, inter 11008)
    if (to == 2 && tg == 4 && (c.nphases == 3 || tq == 4) && lg >= 2 && ld >= 8 && lq >= 4)
        return launch_chain<2, 4, 2, 4, 8, 2, 8, 4>(c, grid, s);        // 13b shapes (hidden 5120, inter 13824)
    pc_set_error("pc_gemm_chain: no instantiation for these tile widths (o/down %d, gate|up %d, q|k|v %d)", to, tg, tq);
    return PC_ERR_ARG;
}

#endif  // PC_DEV_SWEEPS
