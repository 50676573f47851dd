// o_proj + residual (llama2.py:405, :638) of a ONE-ROW forward -- a decode step -- that takes the attention's split-KV PARTIALS as
// its activation source (pc_attn defer_merge) and merges them in its prologue: the merge launch between the attention and o_proj
// (attn_combine_kernel, 4.8 us of a 95 us layer) disappears.
//
//   prologue   every workgroup merges the row's H * D values from the 2..8 partials per head with attn_combine_kernel's arithmetic
//              (csrc/pc_part_merge.h: same order, explicit fmas -> the same bits), splits them into fp16 hi / lo and leaves both as
//              MFMA B-operand fragments of row 0 in LDS ([K/32][4 g][16 B] per plane, 2 * K * 2 bytes); the wave's whole weight share
//              (K / 8 = 16 k-steps of 1 KiB at K = 4096) was requested before, so the stream runs while the row is merged
//   K loop     weight fragments from registers / HBM (two blocks in flight, scalar base + lane offset), activation operands from LDS
//              (every lane reads row 0's fragment of its group g: lanes of rows >= 1 compute columns that are never stored);
//              hi then lo per k-step, k ascending inside wave_k_range's share -- gemm_skinny_kernel<1, 1, EPI_ADD>'s order
//   epilogue   the eight waves' shares through LDS in wave order, + residual, store: bit-identical to pc_attn + attn_combine_kernel +
//              pc_gemm (EPI_ADD) on the merged planes (tests/test_gpu_kernels.py)
#include "pc_gemm_skinny.h"
#include "pc_part_merge.h"
#ifdef PC_DEV_SWEEPS
#include "pc_dev.h"
#endif

using namespace pcg;

namespace {

struct PartGemmParams {
    GemmParams g;            // wf (fp16 image), y, ldy, M = 1, ntiles, KS
    pcm::PartSrc part;
};

constexpr int kPU = 8;       // k-steps per block

__global__ __launch_bounds__(kThreads) void gemm_part_kernel(const PartGemmParams pp) {
    const GemmParams& p = pp.g;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // [K * 2] hi fragments | [K * 2] lo fragments
    __shared__ __attribute__((aligned(16))) float red[kWaves][64][4];
    const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bx = blockIdx.x, KS = p.KS, K = KS * 32, nv = K >> 3;
    // ---- 1. loads: the partials of this thread's chunks, then the weight share ----
    pcm::PartLoads pl;
    pcm::part_issue(pp.part, 0, (tid < nv ? tid : nv - 1) * 8, pl);          // (K <= 4096: one 8-feature chunk per thread)
    int ks0, ks1;
    wave_k_range<false>(p, 0, wave, ks0, ks1);
    const int tile = bx < p.ntiles ? bx : p.ntiles - 1;
    const char* wt = (const char*)p.wf + (int64_t)tile * KS * 1024;
    h8 wA[kPU], wB[kPU];
    auto issue = [&](h8 (&w)[kPU], int kb) {
#pragma unroll
        for (int u = 0; u < kPU; ++u) {
            int k = kb + u < ks1 ? kb + u : ks1 - 1;
            k = k < 0 ? 0 : k;
            w[u] = __builtin_bit_cast(h8, __builtin_nontemporal_load((const u32x4*)(wt + (uint32_t)k * 1024u + (uint32_t)lane * 16u)));
        }
    };
    const int nb = (ks1 - ks0 + kPU - 1) / kPU;
    issue(wA, ks0);
    if (nb > 1) issue(wB, ks0 + kPU);
    const f4 yold = *(const f4*)(p.y + tile * 16 + g * 4);                   // row 0 of the residual stream (written by this lane only)
    // ---- 2. merge, split, fragments of row 0 into LDS ----
    {
        h8 lo;
        const h8 hi = __builtin_bit_cast(h8, pcm::part_merge(pp.part, pl, (pcm::h8*)&lo));
        if (tid < nv) {                                   // chunk c = tid: k-step c >> 2, group c & 3
            *(h8*)(smem + tid * 16) = hi;
            *(h8*)(smem + (size_t)K * 2 + tid * 16) = lo;
        }
    }
    lds_barrier();
    // ---- 3. K loop ----
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const h8* xh = (const h8*)smem + g;
    const h8* xl = (const h8*)(smem + (size_t)K * 2) + g;
    auto consume = [&](const h8 (&w)[kPU], int kb) {
        h8 a[kPU], b[kPU];
#pragma unroll
        for (int u = 0; u < kPU; ++u) {
            const int k = kb + u < ks1 ? kb + u : ks1 - 1;
            a[u] = xh[(k < 0 ? 0 : k) * 4];
            b[u] = xl[(k < 0 ? 0 : k) * 4];
        }
#pragma unroll
        for (int u = 0; u < kPU; ++u) {
            if (kb + u >= ks1) continue;                 // (wave-uniform)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[u], a[u], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[u], b[u], acc, 0, 0, 0);
        }
    };
    {
        int b = 0;
        bool done = false;
        while (b + 2 < nb) {
            consume(wA, ks0 + b * kPU);
            issue(wA, ks0 + (b + 2) * kPU);
            if (!(b + 3 < nb)) {
                consume(wB, ks0 + (b + 1) * kPU);
                consume(wA, ks0 + (b + 2) * kPU);
                done = true;
                break;
            }
            consume(wB, ks0 + (b + 1) * kPU);
            issue(wB, ks0 + (b + 3) * kPU);
            b += 2;
        }
        if (!done) {
            if (b < nb) consume(wA, ks0 + b * kPU);
            if (b + 1 < nb) consume(wB, ks0 + (b + 1) * kPU);
        }
    }
    // ---- 4. the eight waves' shares in wave order, residual, store (row 0 = lanes m == 0) ----
    *(f4*)red[wave][lane] = acc;
    lds_barrier();
    if (wave == 0 && bx < p.ntiles) {
        f4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            const f4 x = *(const f4*)red[w][lane];
            v[0] += x[0]; v[1] += x[1]; v[2] += x[2]; v[3] += x[3];
        }
        if (m == 0) {
            v[0] += yold[0]; v[1] += yold[1]; v[2] += yold[2]; v[3] += yold[3];
            *(f4*)(p.y + tile * 16 + g * 4) = v;
        }
    }
}


#ifdef PC_DEV_SWEEPS
// ---------------------------------------------------------------------------------------------------
// (dev builds only: measured slower than the merge launch + o_proj, csrc/pc_dev.h)  The same for 2..16 rows (the cached prefill of a short question: BASELINE config 1's 12 rows): K is cut across workgroups, the
// reduction runs inside the launch (pc_gemm_ks.hip's hand-off), and no lane reads an activation plane at all:
//   * workgroup (bx, by) owns T = S output tiles and slice by of K; its eight waves cut the slice eight ways -- at K = 4096, S = 8
//     a wave holds TWO k-steps of eight weight tiles, all sixteen 1-KiB loads requested up front;
//   * lane (row m, group g) needs, per k-step, the eight merged values of row m at features 32 ks + 8 g .. + 8 as its MFMA B operand:
//     it merges exactly those from the 2..8 partials (pcm::part_merge -- attn_combine_kernel's arithmetic, so the operand has the
//     bits the merge launch would have written to the planes) while the weight loads are in flight.  Nothing goes through LDS,
//     and the K loop issues no activation load: the two thirds of a one-tile o_proj launch's vector-memory traffic that were
//     activation re-reads (pc_gemm_ks.hip) are gone together with the merge launch (4.8 us + a launch boundary per layer);
//   * a (row, chunk) is merged by the ntiles / T workgroups of its K slice: 2..8 x 40 B per lane and k-step from L2 / MALL.
// Summation order: k ascending inside a wave's share, waves in order, slices in order -- deterministic, not gemm_skinny_kernel's
// order (the result differs from the three-launch form in the last fp32 bits; tests/test_gpu_part_rows.py bounds it).
struct PartKsParams {
    GemmParams g;            // wf, y, ldy, M (= q_len of the partial layout), m_dev (live rows), ntiles, KS, kslices
    pcm::PartSrc part;
    float* slabs;            // [kslices][ntiles][64][4] fp32 partial tiles
    uint32_t* counters;      // [ceil(ntiles / T)], zero between launches
    int32_t formal;
};

template <int T, int KW>     // T tiles per workgroup (one per wave in the reduction), at most KW k-steps per wave
__global__ __launch_bounds__(kThreads) void gemm_part_ks_kernel(const PartKsParams kp) {
    const GemmParams& p = kp.g;
    static_assert(T <= kWaves, "one wave per tile in the reduction");
    __shared__ __attribute__((aligned(16))) float red_raw[kWaves * T * 64 * 4];
    __shared__ int s_last;
    float (*red)[T][64][4] = (float (*)[T][64][4])red_raw;
    const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bx = blockIdx.x, by = blockIdx.y, KS = p.KS, S = p.kslices;
    int ks0, ks1;
    wave_k_range<false>(p, by, wave, ks0, ks1);
    int tile[T];
    wg_tiles<T, EPI_ADD>(p, bx, tile);
    // ---- 1. the wave's whole weight share (k-steps behind the share re-read a valid one and are not multiplied) ----
    h8 w[KW][T];
#pragma unroll
    for (int j = 0; j < KW; ++j) {
        const int k = ks0 + j < KS ? ks0 + j : KS - 1;
#pragma unroll
        for (int t = 0; t < T; ++t)
            w[j][t] = __builtin_bit_cast(h8, __builtin_nontemporal_load(
                (const u32x4*)((const char*)p.wf + ((int64_t)tile[t] * KS + k) * 1024 + (uint32_t)lane * 16u)));
    }
    const int rows_live = p.m_dev ? (*p.m_dev < p.M ? *p.m_dev : p.M) : p.M;
    const bool row_ok = m < rows_live;
    const int it = wave < T ? wave : T - 1;
    f4 yold;
    {
        const int unit = bx * T + it < p.ntiles ? bx * T + it : p.ntiles - 1;
        yold = *(const f4*)(p.y + (int64_t)(m < p.M ? m : p.M - 1) * p.ldy + unit * 16 + g * 4);
    }
    // ---- 2. this lane's operands: merged from the partials, one k-step at a time ----
    h8 xh[KW], xl[KW];
    const int mrow = m < kp.part.q_len ? m : kp.part.q_len - 1;
    constexpr int PB = KW <= 2 ? KW : 1;                 // k-steps whose partial loads are in flight together (80 registers each)
#pragma unroll
    for (int j0 = 0; j0 < KW; j0 += PB) {
        pcm::PartLoads pl[PB];
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int k = ks0 + j0 + u < KS ? ks0 + j0 + u : KS - 1;
            pcm::part_issue(kp.part, mrow, k * 32 + g * 8, pl[u]);
        }
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            h8 lo;
            const h8 hi = __builtin_bit_cast(h8, pcm::part_merge(kp.part, pl[u], (pcm::h8*)&lo));
            const h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            xh[j0 + u] = row_ok ? hi : z;
            xl[j0 + u] = row_ok ? lo : z;
        }
    }
    // ---- 3. MFMAs: hi then lo per k-step, k ascending ----
    f4 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) { f4 z = {0.f, 0.f, 0.f, 0.f}; acc[t] = z; }
#pragma unroll
    for (int j = 0; j < KW; ++j) {
        if (ks0 + j >= ks1) continue;                    // (wave-uniform)
#pragma unroll
        for (int t = 0; t < T; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[j][t], xh[j], acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[j][t], xl[j], acc[t], 0, 0, 0);
        }
    }
    // ---- 4. the eight waves' shares through LDS in wave order; wave t then holds the workgroup's partial of tile t ----
#pragma unroll
    for (int t = 0; t < T; ++t) *(f4*)red[wave][t][lane] = acc[t];
    lds_barrier();
    f4 v = {0.f, 0.f, 0.f, 0.f};
    const bool mine = wave < T && bx * T + it < p.ntiles;
    if (wave < T) {
#pragma unroll
        for (int ww = 0; ww < kWaves; ++ww) {
            const f4 x = *(const f4*)red[ww][wave][lane];
            v[0] += x[0]; v[1] += x[1]; v[2] += x[2]; v[3] += x[3];
        }
    }
    const int my_tile = bx * T + it;
    if (mine) {
        float* dst = kp.slabs + (((int64_t)by * p.ntiles + my_tile) * 64 + lane) * 4;
        st_wt2(dst, v[0], v[1]);
        st_wt2(dst + 2, v[2], v[3]);
    }
    // ---- 5. hand-off (pc_gemm_ks.hip: write-through stores, vmcnt(0), one arrival per workgroup; the last arriver adds in slice order) ----
#if !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__) && defined(__HIP_DEVICE_COMPILE__)
#error "gemm_part_ks_kernel's in-launch hand-off relies on gfx9 vmcnt semantics (stores counted in vmcnt); re-derive it for this target"
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
    constexpr int kMaxS = 8;
    float2 a[kMaxS], b[kMaxS];
#pragma unroll
    for (int s2 = 0; s2 < kMaxS; ++s2) {
        const int sc = s2 < S ? s2 : S - 1;
        const float* src = kp.slabs + (((int64_t)sc * p.ntiles + my_tile) * 64 + lane) * 4;
        a[s2] = ld_wt2(src);
        b[s2] = ld_wt2(src + 2);
    }
    f4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2 = 0; s2 < kMaxS; ++s2)
        if (s2 < S) { r[0] += a[s2].x; r[1] += a[s2].y; r[2] += b[s2].x; r[3] += b[s2].y; }
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    tile_epilogue<EPI_ADD>(p, r, zero, m, my_tile, g, 0, false, zero, zero, true, yold);
}

template <int T, int KW>
int launch_part_ks(const PartKsParams& kp, hipStream_t s) {
    hipLaunchKernelGGL((gemm_part_ks_kernel<T, KW>), dim3(pc_ceil_div(kp.g.ntiles, T), kp.g.kslices), dim3(kThreads), 0, s, kp);
    return pc_check_launch("gemm_part_ks_kernel");
}
#endif  // PC_DEV_SWEEPS

}  // namespace

// y[0][n] += sum_k merged[k] W[n][k]: include/promptcache_hip.h
PC_EXPORT int pc_gemm_part(const void* wf, const float* part_o, const float* part_ml, int32_t nsplit, int32_t H, int32_t D, int32_t N,
                           float* y, void* stream) {
    const int K = H * D;
    PC_REQUIRE(wf && part_o && part_ml && y, PC_ERR_ARG, "pc_gemm_part: null pointer");
    PC_REQUIRE(nsplit >= 2 && nsplit <= pcm::kPartNS && H > 0 && D > 0 && D % 8 == 0 && K % 32 == 0 && K <= 4096 && N > 0 && N % 16 == 0 &&
               (((uintptr_t)part_o | (uintptr_t)part_ml | (uintptr_t)y) & 15) == 0, PC_ERR_ARG,
               "pc_gemm_part: need 2..8 partials per head, K = H * D <= 4096 (K %% 32 == 0), N %% 16 == 0, 16-byte aligned pointers");
    PartGemmParams pp;
    memset(&pp, 0, sizeof(pp));
    pp.g.wf = (const _Float16*)wf; pp.g.y = y; pp.g.ldy = N; pp.g.M = 1; pp.g.ntiles = N / 16; pp.g.KS = K / 32; pp.g.kslices = 1;
    pp.part.part_o = part_o; pp.part.part_ml = part_ml; pp.part.nsplit = nsplit; pp.part.D = D; pp.part.q_len = 1;
    hipLaunchKernelGGL(gemm_part_kernel, dim3(N / 16), dim3(kThreads), (size_t)K * 4, (hipStream_t)stream, pp);
    return pc_check_launch("gemm_part_kernel");
}


#ifdef PC_DEV_SWEEPS
// y[m][n] += sum_k merged[m][k] W[n][k], m < M <= 16: csrc/pc_dev.h
PC_EXPORT int pc_gemm_part_rows(const void* wf, const float* part_o, const float* part_ml, int32_t nsplit, int32_t H, int32_t D, int32_t N,
                                int32_t M, const int32_t* rows_dev, float* y, int64_t ldy, int32_t kslices, void* scratch,
                                int64_t scratch_bytes, void* counters, void* stream) {
    const int K = H * D;
    PC_REQUIRE(wf && part_o && part_ml && y && scratch && counters, PC_ERR_ARG, "pc_gemm_part_rows: null pointer");
    PC_REQUIRE(nsplit >= 2 && nsplit <= pcm::kPartNS && H > 0 && D > 0 && D % 8 == 0 && K % 32 == 0 && N > 0 && N % 16 == 0 && M >= 1 &&
               M <= 16 && ldy >= N && ldy % 4 == 0 && (((uintptr_t)part_o | (uintptr_t)part_ml | (uintptr_t)y | (uintptr_t)scratch) & 15) == 0,
               PC_ERR_ARG, "pc_gemm_part_rows: need 2..8 partials per head, 1..16 rows, K = H * D with K %% 32 == 0, N %% 16 == 0, 16-byte aligned pointers");
    PC_REQUIRE(kslices == 2 || kslices == 4 || kslices == 8, PC_ERR_ARG, "pc_gemm_part_rows: kslices %d not in {2, 4, 8}", kslices);
    PC_REQUIRE(scratch_bytes >= (int64_t)kslices * (N / 16) * 64 * 4 * (int64_t)sizeof(float), PC_ERR_WORKSPACE,
               "pc_gemm_part_rows: scratch smaller than pc_gemm_skinny_ks_scratch_bytes(N, kslices)");
    PartKsParams kp;
    memset(&kp, 0, sizeof(kp));
    kp.g.wf = (const _Float16*)wf; kp.g.y = y; kp.g.ldy = ldy; kp.g.M = M; kp.g.m_dev = rows_dev; kp.g.ntiles = N / 16; kp.g.KS = K / 32;
    kp.g.kslices = kslices;
    kp.part.part_o = part_o; kp.part.part_ml = part_ml; kp.part.nsplit = nsplit; kp.part.D = D; kp.part.q_len = M;
    kp.slabs = (float*)scratch; kp.counters = (uint32_t*)counters; kp.formal = pc_formal_handoff();
    const int share = pc_ceil_div(pc_ceil_div(K / 32, kslices), kWaves);      // k-steps per wave
    hipStream_t s = (hipStream_t)stream;
    // T = kslices tiles per workgroup: N / 16 workgroups whatever the cut (one per CU at N = 4096)
    if (kslices == 8) {
        if (share <= 2) return launch_part_ks<8, 2>(kp, s);
        if (share <= 3) return launch_part_ks<8, 3>(kp, s);
    } else if (kslices == 4) {
        if (share <= 4) return launch_part_ks<4, 4>(kp, s);
        if (share <= 5) return launch_part_ks<4, 5>(kp, s);
    } else {
        if (share <= 8) return launch_part_ks<2, 8>(kp, s);
        if (share <= 10) return launch_part_ks<2, 10>(kp, s);
    }
    pc_set_error("pc_gemm_part_rows: no instantiation for K = %d with %d slices (%d k-steps per wave)", K, kslices, share);
    return PC_ERR_ARG;
}
#endif  // PC_DEV_SWEEPS
