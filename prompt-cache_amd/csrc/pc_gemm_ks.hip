// N = hidden projections of a <= 16-row forward (o_proj, down_proj: llama2.py:405 + :638, :242 + :644) with K split ACROSS
// workgroups and the reduction INSIDE the launch.
//
// Why.  These launches have only N / 16 = 256 output tiles.  With one tile per workgroup (gemm_skinny_kernel<1, 1, EPI_ADD>)
// every workgroup walks all of K, and per k-step a wave issues one 1-KiB weight load next to TWO 1-KiB activation loads (hi and
// lo plane, L2 hits): two thirds of what the CU's vector memory pipe carries is activations that every one of the 256 CUs
// re-reads -- 0.41 / 0.49 of the HBM roof for o_proj / down_proj.  Giving a workgroup T tiles and 1 / S of K keeps S * 256 / T
// workgroups in flight while a k-step's activation loads serve T weight fragments (round 2 measured the stream alone at 18.2 us
// instead of 23.2 us for down_proj at T = 8, S = 8, profiles/r02_gemm_n4096_sweep.txt).  The S partial tiles must then be added:
//   * every workgroup writes its reduced partial tile(s) THROUGH to memory (8-byte agent-scope stores, lane-linear: 1 KiB per
//     tile), drains them, and one lane adds 1 to the arrival counter of its tile group;
//   * the LAST arriver of a group reads all S partials back (agent-scope loads: they bypass its L1), adds them in slice order
//     0 .. S-1 -- the result does not depend on who arrived last -- adds the residual stream and stores y; it leaves the counter
//     at zero for the next launch.  No spin, no fence, no second launch.
#include "pc_gemm_skinny.h"

using namespace pcg;

namespace {

constexpr int kMaxSlices = 8;

struct KsParams {
    GemmParams g;            // wf, xf_hi, xf_lo, y, ldy, M, ntiles, KS, kslices
    float* slabs;            // [kslices][ntiles][64][4] fp32 partial tiles (MFMA C layout, lane-linear)
    uint32_t* counters;      // [ceil(ntiles / T)] arrival counters, zero between launches
    int32_t formal;          // acq_rel arrival (PC_FORMAL_HANDOFF=1) instead of relaxed + vmcnt(0)
};

// DB: the K loop keeps TWO blocks of U k-steps in flight (block b + 1 is requested before block b is consumed, block b + 2 into
// b's registers right after), addresses as a wave-uniform base + one 32-bit lane offset per k-step (pc_gemm_q8.hip's loop)
// MT: row tiles (1: M <= 16; 2: M <= 32 -- the 17..32-row questions of BASELINE config 3 keep their K slices and the in-launch residual
// add instead of leaving slabs to a pc_rmsnorm_frag launch); an item = (row tile a, weight tile t), MT * T <= 8 items, one per wave
template <int T, int U, bool DB = false, int MT = 1>
__global__ __launch_bounds__(kThreads) void gemm_skinny_ks_kernel(const KsParams kp) {
    const GemmParams& p = kp.g;
    static_assert(MT * T <= kWaves && (!DB || MT == 1), "one wave per (row tile, weight tile) item");
    constexpr int NI = MT * T;
    __shared__ __attribute__((aligned(16))) float red_raw[kWaves * NI * 64 * 4];
    __shared__ int s_last;
    float (*red)[NI][64][4] = (float (*)[NI][64][4])red_raw;
    const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bx = blockIdx.x, by = blockIdx.y, KS = p.KS, S = p.kslices;
    int ks0, ks1;
    wave_k_range<false>(p, by, wave, ks0, ks1);
    int tile[T];
    wg_tiles<T, EPI_ADD>(p, bx, tile);
    f4 acc[MT][T];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int t = 0; t < T; ++t) { f4 z = {0.f, 0.f, 0.f, 0.f}; acc[a][t] = z; }
    const _Float16* wbase[T];
#pragma unroll
    for (int t = 0; t < T; ++t) wbase[t] = p.wf + ((int64_t)tile[t] * KS * 64 + lane) * 8;
    const _Float16* xh_base = p.xf_hi + lane * 8;
    const _Float16* xl_base = p.xf_lo + lane * 8;
    bool row_ok[MT];
    const int rows_live = p.m_dev ? (*p.m_dev < p.M ? *p.m_dev : p.M) : p.M;
#pragma unroll
    for (int a = 0; a < MT; ++a) row_ok[a] = a * 16 + m < rows_live;
    // this wave's item: row tile ia, weight tile it (waves behind the items repeat the last one and store nothing)
    const int item = wave < NI ? wave : NI - 1;
    const int ia = item / T, it = item - ia * T;
    const int my_row = ia * 16 + m;
    // the residual tile the last arriver will add to, fetched now (clamped, unconditional; written by that lane only)
    f4 yold;
    {
        const int unit = bx * T + it < p.ntiles ? bx * T + it : p.ntiles - 1;
        yold = *(const f4*)(p.y + (int64_t)(my_row < p.M ? my_row : p.M - 1) * p.ldy + unit * 16 + g * 4);
    }
    if constexpr (DB) {
        const char* wt[T];
#pragma unroll
        for (int t = 0; t < T; ++t) wt[t] = (const char*)p.wf + (int64_t)tile[t] * KS * 1024;
        const char* xhb = (const char*)p.xf_hi;
        const char* xlb = (const char*)p.xf_lo;
        h8 wA[U][T], wB[U][T], hA[U], hB[U], lA[U], lB[U];
        const bool rok = row_ok[0];
        auto issue = [&](h8 (&w)[U][T], h8 (&xh)[U], h8 (&xl)[U], int kb) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                int k = kb + u < ks1 ? kb + u : ks1 - 1;
                k = k < 0 ? 0 : k;
                const uint32_t voff = (uint32_t)k * 1024u + (uint32_t)lane * 16u;
#pragma unroll
                for (int t = 0; t < T; ++t) w[u][t] = __builtin_bit_cast(h8, __builtin_nontemporal_load((const u32x4*)(wt[t] + voff)));
                h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                xh[u] = z; xl[u] = z;
                if (rok) {
                    xh[u] = *(const h8*)(xhb + voff);
                    xl[u] = *(const h8*)(xlb + voff);
                }
            }
        };
        auto consume = [&](const h8 (&w)[U][T], const h8 (&xh)[U], const h8 (&xl)[U], int kb) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (kb + u >= ks1) continue;             // (wave-uniform)
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[u][t], xh[u], acc[0][t], 0, 0, 0);
                    acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[u][t], xl[u], acc[0][t], 0, 0, 0);
                }
            }
        };
        const int nb = (ks1 - ks0 + U - 1) / U;
        issue(wA, hA, lA, ks0);
        if (nb > 1) issue(wB, hB, lB, ks0 + U);
        int b = 0;
        bool done = false;
        while (b + 2 < nb) {
            consume(wA, hA, lA, ks0 + b * U);
            issue(wA, hA, lA, ks0 + (b + 2) * U);
            if (!(b + 3 < nb)) {
                consume(wB, hB, lB, ks0 + (b + 1) * U);
                consume(wA, hA, lA, ks0 + (b + 2) * U);
                done = true;
                break;
            }
            consume(wB, hB, lB, ks0 + (b + 1) * U);
            issue(wB, hB, lB, ks0 + (b + 3) * U);
            b += 2;
        }
        if (!done) {
            if (b < nb) consume(wA, hA, lA, ks0 + b * U);
            if (b + 1 < nb) consume(wB, hB, lB, ks0 + (b + 1) * U);
        }
    } else {
    int ks = ks0;
    for (; ks + U <= ks1; ks += U) k_block<MT, T, true, U, false>(wbase, xh_base, xl_base, KS, ks, U, row_ok, acc);
    if (ks < ks1) k_block<MT, T, true, U, true>(wbase, xh_base, xl_base, KS, ks, ks1 - ks, row_ok, acc);
    }

    // ---- the eight waves' K shares through LDS, fixed order; wave t then holds this workgroup's partial of tile t ----
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int t = 0; t < T; ++t) *(f4*)red[wave][a * T + t][lane] = acc[a][t];
    lds_barrier();
    f4 v = {0.f, 0.f, 0.f, 0.f};
    const bool mine = wave < NI && bx * T + it < p.ntiles;           // (clamped duplicate tiles are not stored)
    if (wave < NI) {
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            const f4 x = *(const f4*)red[w][wave][lane];
            v[0] += x[0]; v[1] += x[1]; v[2] += x[2]; v[3] += x[3];
        }
    }
    const int my_tile = bx * T + it;
    const int64_t slab_tiles = (int64_t)MT * p.ntiles;               // partial tiles per slice: [row tile][weight tile]
    const int64_t my_slot = (int64_t)ia * p.ntiles + my_tile;
    if (mine) {
        float* dst = kp.slabs + (((int64_t)by * slab_tiles + my_slot) * 64 + lane) * 4;
        st_wt2(dst, v[0], v[1]);
        st_wt2(dst + 2, v[2], v[3]);
    }
    // Hand-off contract (gfx9 / CDNA: this is the MI355X guide's "8-byte agent-scope atomics on both sides" form).  Payload =
    // agent-scope atomic stores: they write THROUGH the XCD's L2 to the coherence point and count in vmcnt like any vector
    // memory operation on gfx9, so `s_waitcnt vmcnt(0)` below means "acknowledged by memory"; the barrier orders every wave's
    // drain before lane 0's arrival; the last arriver reads with agent-scope atomic loads, which bypass its L1 / L2.  No release
    // / acquire pair appears in the source because none is needed on this ISA -- and a compiler or ISA that tracked stores in a
    // separate counter (gfx10+: vscnt) would break it silently, hence the guard:
#if !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__) && defined(__HIP_DEVICE_COMPILE__)
#error "gemm_skinny_ks_kernel's in-launch hand-off relies on gfx9 vmcnt semantics (stores counted in vmcnt); re-derive it for this target"
#endif
    // PC_FORMAL_HANDOFF=1 (environment, read at launch: kp.formal): the C++-memory-model form (release on the arrival, acquire in
    // the last arriver) for A/B: +1.7 us per launch on MI355X (buffer_wbl2 + buffer_inv on the critical path;
    // profiles/r04_variants.txt).  tests/test_gpu_handoff.py hammers the default form (launches under uneven load, every word
    // compared) and runs the formal form through the same stress.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // every storing wave: its stores are acknowledged
    __syncthreads();
    if (tid == 0) {
        gu32* c = (gu32*)(kp.counters + bx);
        const uint32_t old = kp.formal ? __hip_atomic_fetch_add(c, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT)
                                       : __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (old + 1u == (uint32_t)S) ? 1 : 0;
        if (last) __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = last;
    }
    __syncthreads();
    if (!s_last || !mine) return;
    // ---- last arriver: the S partials of its tiles, slice order ----
    float2 a[kMaxSlices], b[kMaxSlices];
#pragma unroll
    for (int s = 0; s < kMaxSlices; ++s) {
        const int sc = s < S ? s : S - 1;                             // clamped re-read instead of a branch around the loads
        const float* src = kp.slabs + (((int64_t)sc * slab_tiles + my_slot) * 64 + lane) * 4;
        a[s] = ld_wt2(src);
        b[s] = ld_wt2(src + 2);
    }
    f4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < kMaxSlices; ++s)
        if (s < S) { r[0] += a[s].x; r[1] += a[s].y; r[2] += b[s].x; r[3] += b[s].y; }
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    tile_epilogue<EPI_ADD>(p, r, zero, my_row, my_tile, g, 0, false, zero, zero, true, yold);
}

template <int T, int U, bool DB = false, int MT = 1>
int launch_ks(const KsParams& kp, hipStream_t s) {
    const dim3 grid(pc_ceil_div(kp.g.ntiles, T), kp.g.kslices);
    hipLaunchKernelGGL((gemm_skinny_ks_kernel<T, U, DB, MT>), grid, dim3(kThreads), 0, s, kp);
    return pc_check_launch("gemm_skinny_ks_kernel");
}

}  // namespace

PC_EXPORT int64_t pc_gemm_skinny_ks_scratch_bytes(int32_t N, int32_t kslices) {
    if (N <= 0 || kslices <= 0) return 0;
    return (int64_t)kslices * (N / 16) * 64 * 4 * (int64_t)sizeof(float);
}

namespace pcg {
// y[m][n] += sum_k x[m][k] W[n][k]   (M <= 16 rows, split-precision activation planes, fp16 weight image), K cut into
// `kslices` (1..8) workgroup slices with `tiles_per_wg` (1, 2, 4 or 8) output tiles per workgroup; scratch >=
// pc_gemm_skinny_ks_scratch_bytes(N, kslices) bytes; counters: ceil(N / 16 / tiles_per_wg) uint32 words, zero before the first
// launch (every launch leaves them zero).  Deterministic: the partials are added in slice order whoever arrives last.
int launch_skinny_ks(const void* wf, const void* xf_hi, const void* xf_lo, int M, int N, int K, float* y, int64_t ldy, int kslices,
                     int tiles_per_wg, void* scratch, int64_t scratch_bytes, void* counters, const int32_t* rows_dev, hipStream_t s) {
    PC_REQUIRE(wf && xf_hi && xf_lo && y && scratch && counters, PC_ERR_ARG, "pc_gemm (ks): null pointer");
    PC_REQUIRE(M > 0 && M <= 32 && N > 0 && N % 16 == 0 && K > 0 && K % 32 == 0 && ldy >= N && ldy % 4 == 0, PC_ERR_ARG,
               "pc_gemm (ks): need 1 <= M <= 32, N %% 16 == 0, K %% 32 == 0");
    PC_REQUIRE(kslices >= 1 && kslices <= kMaxSlices, PC_ERR_ARG, "pc_gemm (ks): kslices %d outside 1..8", kslices);
    const int mt = M > 16 ? 2 : 1;
    PC_REQUIRE(scratch_bytes >= mt * pc_gemm_skinny_ks_scratch_bytes(N, kslices) && ((uintptr_t)scratch & 15) == 0, PC_ERR_WORKSPACE,
               "pc_gemm (ks): scratch too small (17..32 rows: twice pc_gemm_skinny_ks_scratch_bytes) or misaligned");
    KsParams kp;
    memset(&kp, 0, sizeof(kp));
    kp.g.wf = (const _Float16*)wf; kp.g.xf_hi = (const _Float16*)xf_hi; kp.g.xf_lo = (const _Float16*)xf_lo;
    kp.g.y = y; kp.g.ldy = ldy; kp.g.M = M; kp.g.m_dev = rows_dev; kp.g.ntiles = N / 16; kp.g.KS = K / 32; kp.g.kslices = kslices;
    kp.slabs = (float*)scratch; kp.counters = (uint32_t*)counters; kp.formal = pc_formal_handoff();
    // k-steps per block: a wave's K share is K / 32 / (8 kslices) k-steps -- keep the whole share in flight where it fits
    const int share = pc_ceil_div(pc_ceil_div(K / 32, kslices), kWaves);
#ifdef PC_DEV_SWEEPS      // (two blocks in flight: measured no gain, profiles/r05_ks_db_ab.txt -- dev builds only)
    static const int db = [] { const char* e = getenv("PC_KS_DB"); return e ? atoi(e) : 0; }();
#endif
    if (mt == 2) {                                       // two row tiles: at most four weight tiles per workgroup
        switch (tiles_per_wg) {
            case 1: return share > 4 ? launch_ks<1, 8, false, 2>(kp, s) : launch_ks<1, 4, false, 2>(kp, s);
            case 2: return share > 4 ? launch_ks<2, 6, false, 2>(kp, s) : launch_ks<2, 4, false, 2>(kp, s);
            case 4: return launch_ks<4, 2, false, 2>(kp, s);
            default: break;
        }
        pc_set_error("pc_gemm (ks): 17..32 rows take ks_tiles 1, 2 or 4 (got %d)", tiles_per_wg);
        return PC_ERR_ARG;
    }
#ifdef PC_DEV_SWEEPS
    if (db) {                                            // two blocks in flight (PC_KS_DB=1: A/B against the single-block loop)
        switch (tiles_per_wg) {
            case 1: return launch_ks<1, 4, true>(kp, s);
            case 2: return launch_ks<2, 3, true>(kp, s);
            case 4: return launch_ks<4, 2, true>(kp, s);
            case 8: return launch_ks<8, 1, true>(kp, s);
            default: break;
        }
    }
#endif
    switch (tiles_per_wg) {
        case 1: return share > 8 ? launch_ks<1, 16>(kp, s) : launch_ks<1, 8>(kp, s);
        case 2: return share > 8 ? launch_ks<2, 12>(kp, s) : launch_ks<2, 8>(kp, s);
        case 4: return share > 4 ? launch_ks<4, 8>(kp, s) : launch_ks<4, 4>(kp, s);
        case 8: return share > 3 ? launch_ks<8, 4>(kp, s) : launch_ks<8, 3>(kp, s);
        default: break;
    }
    pc_set_error("pc_gemm (ks): ks_tiles %d not in {1, 2, 4, 8}", tiles_per_wg);
    return PC_ERR_ARG;
}
}  // namespace pcg
