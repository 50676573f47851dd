// 65..512-row weight-streaming projections (gemm_rows_kernel): long questions in front of a staged cache
// (replaces the nn.Linear calls of promptcache/model/llama2.py:345-347, :405, :242 at q = 65..512; see pc_gemm_skinny.h).
#include "pc_gemm_skinny.h"

namespace pcg {


// ---------------------------------------------------------------------------------------------------
// 65..512 rows ("mid M": long questions in front of a staged cache).  Still weight streaming -- the weights are
// read once -- but a workgroup can no longer afford to split K across its waves: every workgroup would re-read
// all the activation planes and hold MT*TT accumulators per wave.  Here the waves split the ROWS instead:
//   * compute wave w (w < ceil(M/64)) owns rows [64w, 64w+64) x all TT weight tiles of the workgroup
//     (4 x TT accumulators), reads its own activation fragments straight from L2 (prefetched one k-step ahead)
//     and the weight fragments from LDS;
//   * kStageWaves extra waves do nothing but stream the workgroup's weight tiles HBM -> registers -> LDS, three
//     stages (24-32 KiB each) deep.  Their load queues hold only weight loads, the compute waves' queues only
//     activation loads: a wave's loads complete in order, so one wave issuing both would make every L2-hit
//     activation load wait behind ~2 us HBM weight loads.
// One raw s_barrier per stage (LDS-only wait: `__syncthreads()` would drain the prefetch queues).
// Activations: hi and lo planes when the caller passes both (two MFMAs per weight fragment), the hi plane only when
// xf_lo is NULL.
constexpr int kStageWaves = 4;


// SW staging waves, KCV k-steps per stage, PDV activation prefetch distance, LB launch bound (0 / -1 / 0: the defaults)
// NSD > 0: the staging waves move the stages HBM -> LDS by LDS-DMA (`global_load_lds`, no registers in between) through a ring of
// NSD stages instead of three register sets and two LDS buffers
// XT (with NSD): an ODD number of row tiles leaves the last one to the staging waves -- with LDS-DMA staging they hold no weight
// registers and issue a handful of instructions per stage, so each multiplies that row tile against a third of the workgroup's weight
// tiles.  (Otherwise the odd tile is a ninth compute wave on the SIMD that already runs two: 5 row tiles against 4 on the others --
// 259 rows = 17 tiles cost 8 % more than 256.)
template <int TT, int EPI, int MTW, bool TWO, int SW = kStageWaves, int KCV = 0, int PDV = -1, int LB = 0, int NSD = 0, bool XT = false>
__global__ __launch_bounds__(LB ? LB : (MTW == 2 ? 1024 : 768)) void gemm_rows_kernel(const GemmParams p) {
    constexpr int T = (EPI == EPI_SILU) ? TT / 2 : TT;   // output units (tiles, or gate/up pairs) per workgroup
    constexpr int KC = KCV ? KCV : (TT <= 4) ? 8 : 4;    // k-steps per stage
    constexpr int F = TT * KC;                           // 1-KiB fragments per stage
    constexpr int FPW = F / SW;
    constexpr int PD = PDV >= 0 ? PDV : (MTW == 2) ? 3 : 1, NX = PD + 1;  // activation prefetch distance (k-steps) / register sets
    static_assert(F % SW == 0 && KC % NX == 0, "stage must split evenly over the staging waves");
    constexpr int NB = NSD ? NSD : 2;                    // LDS stages
    constexpr int FA = XT ? 2 * KC : 0;                  // (XT) the odd row tile's activation fragments of a stage, hi then lo, behind the weights
    static_assert(!XT || FA % SW == 0, "the activation fragments split evenly over the staging waves");
    __shared__ __attribute__((aligned(16))) _Float16 wbuf[NB][F + FA][64][8];

    const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // rows of this workgroup: all of them, or block blockIdx.z of p.zrows rows (289..512 rows: two row blocks per column panel)
    const int row0 = p.zrows ? (int)blockIdx.z * p.zrows : 0;
    const int Mz = p.zrows ? ((p.M - row0 < p.zrows) ? p.M - row0 : p.zrows) : p.M;
    // compute waves: what the LAUNCH was sized for -- the rows of a full block.  (Counting them from this block's own Mz gave the
    // shorter second block of a two-block launch one compute wave fewer than the host launched: a fourth wave then took the
    // staging branch of a three-wave staging split and wrote fragment slot 24 of a 24-slot stage -- M = 321..336, 385..400,
    // 449..464.)  A compute wave whose rows all lie behind Mz only keeps the barriers: nva <= 0, nothing stored.
    static_assert(!XT || (NSD > 0 && MTW == 2 && TWO), "the staging waves take the odd row tile of a two-plane, two-tiles-per-wave launch with ring staging");
    const int RW = XT ? ((p.M + 15) >> 4) / MTW : ((p.zrows ? p.zrows : p.M) + 16 * MTW - 1) / (16 * MTW);   // (XT: one block, odd tile count)
    const int KS = p.KS;
    const int ksq = (KS + p.kslices - 1) / p.kslices;
    const int kq0 = blockIdx.y * ksq;
    const int kq1 = (kq0 + ksq < KS) ? kq0 + ksq : KS;
    const int nst = (kq1 - kq0 + KC - 1) / KC;
    const int nunits = (EPI == EPI_SILU) ? p.npairs : p.ntiles;
    // (Rotating the K walk per workgroup, so that workgroups do not read the same activation fragments from L2 in
    // lockstep, measured no gain: L2 channel conflicts are not what bounds this kernel.)

    if (wave >= RW) {
        // ---------------- staging wave ----------------
        if (nst <= 0 && !XT) return;                     // empty K slice (kslices > k-steps): nothing to stream
        const int sidx = wave - RW;
        const _Float16* src[FPW];
        int kst[FPW];
#pragma unroll
        for (int i = 0; i < FPW; ++i) {
            const int f = sidx + SW * i;                 // fragment of the stage: k-step-major, tile-minor
            const int kk = f / TT, tt = f - kk * TT;
            int unit = blockIdx.x * T + (EPI == EPI_SILU ? (tt < T ? tt : tt - T) : tt);
            if (unit >= nunits) unit = nunits - 1;       // clamped duplicates are computed and never stored
            const int tile = (EPI == EPI_SILU && tt >= T) ? p.npairs + unit : unit;
            kst[i] = kk;
            src[i] = p.wf + (((int64_t)tile * KS + kq0 + kk) * 64 + lane) * 8;
        }
        if constexpr (NSD > 0) {
            // Ring of NSD stages: stage st lives in slot st % NSD.  Compute waves pass barrier #st when they are done with
            // stage st-1, so after it slot (st-1) % NSD is free: that is where stage st+NSD-1 goes.  Before barrier #st this
            // wave's DMAs of stage st must have landed: all but the NSD-2 stages issued after it (vmcnt counts in order).
            // K-steps of the last stage past the end of the K range re-read the last valid one; the compute waves skip them.
            const uint32_t w0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)&wbuf[0][0][0][0];
            auto issue = [&](int st, int slot) __attribute__((always_inline)) {
                const int se = st < nst ? st : nst - 1;
#pragma unroll
                for (int i = 0; i < FPW; ++i) {
                    const int kabs = kq0 + se * KC + kst[i];
                    const int back = kabs < kq1 ? 0 : kabs - (kq1 - 1);
                    const _Float16* g = src[i] + ((int64_t)se * KC - back) * 512;
                    const uint32_t dst = w0 + (uint32_t)((slot * (F + FA) + sidx + SW * i) * 1024);
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" ::"v"(g), "s"(dst) : "memory", "m0");
                }
            };
            if constexpr (!XT) {
            int slot = 0;
            for (int st = 0; st < NSD - 1 && st < nst; ++st) { issue(st, slot); slot = slot + 1 == NSD ? 0 : slot + 1; }
            for (int st = 0; st < nst; ++st) {
                const int ahead = nst - 1 - st < NSD - 2 ? nst - 1 - st : NSD - 2;   // stages issued behind st: they may stay in flight
#pragma unroll
                for (int r = 0; r <= NSD - 2; ++r)
                    if (ahead == r) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(r * FPW) : "memory");
                asm volatile("s_barrier" ::: "memory");
                if (st + NSD - 1 < nst) {
                    issue(st + NSD - 1, slot);
                    slot = slot + 1 == NSD ? 0 : slot + 1;
                }
            }
            } else {
            // ---- XT: stream AND multiply the odd last row tile against this wave's share of the weight tiles ----
            // units (tiles, or gate/up pairs) sidx, sidx + SW, ...  The row tile's activation fragments travel with the stage: 2 KC more
            // 1-KiB DMAs behind the weights (hi plane, then lo), shared by the three waves -- every vector-memory operation of this
            // wave is an LDS-DMA under the counted waits below, the operands are LDS reads.
            constexpr int NU = (T + SW - 1) / SW;        // units per staging wave, at most
            constexpr int TPU = (EPI == EPI_SILU) ? 2 : 1;
            constexpr int APW = FA / SW;                 // activation fragments per staging wave and stage
            constexpr int OPS = FPW + APW;               // DMA instructions per wave and stage
            const int xt_tile = MTW * RW;                // (row0 = 0: one block)
            const _Float16* xa = p.xf_hi + (((int64_t)xt_tile * KS + kq0) * 64 + lane) * 8;
            const int64_t lo_delta = p.xf_lo - p.xf_hi;
            const int klast = kq1 - 1 - kq0;
            f4 acc[NU][TPU];
#pragma unroll
            for (int i = 0; i < NU; ++i)
#pragma unroll
                for (int q = 0; q < TPU; ++q) { f4 z = {0.f, 0.f, 0.f, 0.f}; acc[i][q] = z; }
            auto issue_all = [&](int st, int slot) __attribute__((always_inline)) {
                issue(st, slot);
#pragma unroll
                for (int i = 0; i < APW; ++i) {
                    const int a = sidx + SW * i, pl = a / KC, kk = a - pl * KC;
                    const int kn = st * KC + kk < klast ? st * KC + kk : klast;
                    const _Float16* g = xa + (pl ? lo_delta : 0) + (int64_t)kn * 512;
                    const uint32_t dst = w0 + (uint32_t)((slot * (F + FA) + F + a) * 1024);
                    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(dst) : "memory", "m0");
                }
            };
            int slot = 0, rslot = 0;
            for (int st = 0; st < NSD - 1 && st < nst; ++st) { issue_all(st, slot); slot = slot + 1 == NSD ? 0 : slot + 1; }
            for (int st = 0; st < nst; ++st) {
                const int ahead = nst - 1 - st < NSD - 2 ? nst - 1 - st : NSD - 2;
#pragma unroll
                for (int r = 0; r <= NSD - 2; ++r)
                    if (ahead == r) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(r * OPS) : "memory");
                asm volatile("s_barrier" ::: "memory");
                if (st + NSD - 1 < nst) {
                    issue_all(st + NSD - 1, slot);
                    slot = slot + 1 == NSD ? 0 : slot + 1;
                }
                const _Float16* wst = &wbuf[rslot][0][lane][0];
                rslot = rslot + 1 == NSD ? 0 : rslot + 1;
#pragma unroll
                for (int j = 0; j < KC; ++j) {
                    if (st * KC + j > klast) continue;                  // (wave-uniform: re-read k-steps behind the end of the K range)
                    const h8 xh = *(const h8*)(wst + (F + j) * 512);
                    const h8 xl = *(const h8*)(wst + (F + KC + j) * 512);
#pragma unroll
                    for (int i = 0; i < NU; ++i) {
                        const int u = sidx + SW * i;
                        if (u >= T) continue;
#pragma unroll
                        for (int q = 0; q < TPU; ++q) {
                            const h8 w = *(const h8*)(wst + (j * TT + u + q * T) * 512);
                            acc[i][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, xh, acc[i][q], 0, 0, 0);
                            acc[i][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, xl, acc[i][q], 0, 0, 0);
                        }
                    }
                }
            }
            const int lrow = xt_tile * 16 + m;
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                const int u = sidx + SW * i;
                if (u >= T) continue;
                f4 uu = {0.f, 0.f, 0.f, 0.f};
                if (EPI == EPI_SILU) uu = acc[i][TPU - 1];
                tile_epilogue<EPI>(p, acc[i][0], uu, lrow < Mz ? lrow : p.M, (int)blockIdx.x * T + u, g, (int)blockIdx.y);
            }
            }
            return;
        }
        // (dev timing probes, tools/rows_exp.sh: -DPC_ROWS_EXP=2 streams nothing, =1 computes nothing, =3 loads no activations)
#if defined(PC_ROWS_EXP) && PC_ROWS_EXP == 2
#define PC_ROWS_STAGE_SRC(dst, ptr) { (void)(ptr); dst = h8{1, 1, 1, 1, 1, 1, 1, 1}; }
#else
#define PC_ROWS_STAGE_SRC(dst, ptr) dst = ldg_h8_nt(ptr);
#endif
        h8 r0[FPW], r1[FPW], r2[FPW];
        const h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
        // loads of stage st into a register set; k-steps past the end of the K range re-read the last valid
        // one (never out of bounds) and are zeroed when they are written to LDS
#define PC_STAGE_LOAD(R, ST)                                                                  \
        {                                                                                     \
            const int se = (ST) < nst ? (ST) : nst - 1;   /* stages past the end: the last one */ \
            _Pragma("unroll") for (int i = 0; i < FPW; ++i) {                                 \
                const int kabs = kq0 + se * KC + kst[i];                                      \
                const int back = kabs < kq1 ? 0 : kabs - (kq1 - 1);                           \
                PC_ROWS_STAGE_SRC(R[i], src[i] + ((int64_t)se * KC - back) * 512)            \
            }                                                                                 \
        }
#define PC_STAGE_WRITE(R, ST)                                                                 \
        {                                                                                     \
            const int se = (ST);                                                              \
            _Pragma("unroll") for (int i = 0; i < FPW; ++i) {                                 \
                const bool ok = kq0 + se * KC + kst[i] < kq1;                                 \
                *(h8*)wbuf[(ST) & 1][sidx + SW * i][lane] = ok ? R[i] : zero;                 \
            }                                                                                 \
        }
        // Loads are issued unconditionally (stages past the end re-read the last valid k-step): a load under a
        // branch makes hipcc's vmcnt bookkeeping assume the not-taken path and wait for the NEWEST loads
        // before each LDS write, which would drain the three-stage prefetch every stage.
        PC_STAGE_LOAD(r0, 0)
        PC_STAGE_LOAD(r1, 1)
        for (int st = 0; st < nst; st += 3) {
            PC_STAGE_LOAD(r2, st + 2)
            PC_STAGE_WRITE(r0, st)
            lds_barrier();
            PC_STAGE_LOAD(r0, st + 3)
            if (st + 1 < nst) {
                PC_STAGE_WRITE(r1, st + 1)
                lds_barrier();
            }
            PC_STAGE_LOAD(r1, st + 4)
            if (st + 2 < nst) {
                PC_STAGE_WRITE(r2, st + 2)
                lds_barrier();
            }
        }
#undef PC_STAGE_LOAD
#undef PC_STAGE_WRITE
#undef PC_ROWS_STAGE_SRC
        return;
    }

    // ---------------- compute wave ----------------
    f4 acc[MTW][TT];
#pragma unroll
    for (int a = 0; a < MTW; ++a)
#pragma unroll
        for (int t = 0; t < TT; ++t) { f4 z = {0.f, 0.f, 0.f, 0.f}; acc[a][t] = z; }
    // Activation loads are unconditional as well (same vmcnt reason): row tiles past the end of the planes are
    // clamped to the last one, pad rows inside it are read as they are -- output column m depends on
    // activation row m only, and rows >= M are never stored.
    const _Float16* xa[MTW];
    const int64_t lo_delta = TWO ? (p.xf_lo - p.xf_hi) : 0;          // lo plane = hi plane + lo_delta (same layout)
    const int mt_last = ((Mz + 15) >> 4) - 1;            // (row tiles count from this block's first row)
#pragma unroll
    for (int a = 0; a < MTW; ++a) {
        int mt = MTW * wave + a;
        mt = mt < mt_last ? mt : mt_last;
        xa[a] = p.xf_hi + (((int64_t)((row0 >> 4) + mt) * KS + kq0) * 64 + lane) * 8;
    }
    // row tiles this wave really owns (wave-uniform; <= 0 for a wave behind the block's last row): the MFMAs of the clamped duplicates behind the last tile are skipped
    const int nva = __builtin_amdgcn_readfirstlane((mt_last + 1 - MTW * wave) < MTW ? (mt_last + 1 - MTW * wave) : MTW);
    const int klast = kq1 - 1 - kq0;                     // last valid k-step, relative to kq0
    // k-step (relative to kq0) loaded for sequence position i; positions past the end repeat the last one
    auto kseq = [&](int i) { return i < klast ? i : klast; };
    h8 xs[NX][MTW], xsl[TWO ? NX : 1][MTW];
    if (nst > 0) {                                       // (an empty K slice stores zeros below)
#pragma unroll
        for (int d = 0; d < PD; ++d) {
            const int kn = kseq(d);
#pragma unroll
            for (int a = 0; a < MTW; ++a) {
                xs[d][a] = ldg_h8(xa[a] + (int64_t)kn * 512);
                if (TWO) xsl[d][a] = ldg_h8(xa[a] + lo_delta + (int64_t)kn * 512);
            }
        }
    }
    int slot = 0;
    for (int st = 0; st < nst; ++st) {
        lds_barrier();                                   // stage st is in wbuf[st & 1] (ring: wbuf[st % NSD])
        const _Float16* wst = &wbuf[NSD ? slot : (st & 1)][0][lane][0];
        if (NSD) slot = slot + 1 == NSD ? 0 : slot + 1;
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            // prefetch the activation fragments PD k-steps ahead (clamped at the end of the K range)
            const int kn = kseq(st * KC + j + PD);
#pragma unroll
            for (int a = 0; a < MTW; ++a) {
#if defined(PC_ROWS_EXP) && PC_ROWS_EXP == 3
                (void)kn;
#else
                xs[(j + PD) % NX][a] = ldg_h8(xa[a] + (int64_t)kn * 512);
                if (TWO) xsl[(j + PD) % NX][a] = ldg_h8(xa[a] + lo_delta + (int64_t)kn * 512);
#endif
            }
            __builtin_amdgcn_sched_barrier(0);           // keep the prefetch ahead of this k-step's MFMAs (see k_block)
            h8 w[TT];
#pragma unroll
            for (int t = 0; t < TT; ++t) w[t] = *(const h8*)(wst + (j * TT + t) * 512);
            const bool kok = !NSD || st * KC + j <= klast;   // (ring stages hold re-read k-steps behind the end of the K range, not zeros)
#pragma unroll
            for (int a = 0; a < MTW; ++a) {
#if defined(PC_ROWS_EXP) && PC_ROWS_EXP == 1
                if (a < nva && p.M < 0) {
#else
                if (a < nva && kok) {
#endif
#pragma unroll
                    for (int t = 0; t < TT; ++t) {
                        acc[a][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[t], xs[j % NX][a], acc[a][t], 0, 0, 0);
                        if (TWO) acc[a][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[t], xsl[j % NX][a], acc[a][t], 0, 0, 0);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int a = 0; a < MTW; ++a)
#pragma unroll
        for (int t = 0; t < T; ++t) {
            f4 u = {0.f, 0.f, 0.f, 0.f};
            if (EPI == EPI_SILU) u = acc[a][T + t];
            // (rows past this block's end belong to the next block or to nobody: they are not stored)
            const int lrow = (MTW * wave + a) * 16 + m;
            tile_epilogue<EPI>(p, acc[a][t], u, lrow < Mz ? row0 + lrow : p.M, (int)blockIdx.x * T + t, g, (int)blockIdx.y);
        }
}

// (Tried and dropped: also splitting the rows of a column group over 2-4 workgroups placed on one XCD, with the shape
// chosen by bytes-per-workgroup: 10-15 % faster on a few isolated shapes (7b q|k|v at 259 / 512 rows), but the forward
// as a whole did not gain -- q = 98: 5.8 -> 6.2 ms, config 4: 19.4 -> 20.1 ms.)
// Launch shape of the rows kernel.  Rows per compute wave: 32 while that needs <= 12 compute waves (M <= 384; more,
// narrower waves hide the L2 latency of the activation loads and spread evenly over the four SIMDs), else 64.
// Weight tiles per workgroup: the smallest of {3,4,6} ({2,3} gate/up pairs) that fits the grid into one round (else 6).
template <int EPI>
int launch_rows(const GemmParams& p_in, int units, hipStream_t s) {
    static const int forced = [] { const char* e = getenv("PC_GEMM_ROWS_TT"); return e ? atoi(e) : 0; }();
    const GemmParams& p = p_in;
    const bool narrow = pc_ceil_div(p.M, 32) <= 12;
    const bool two = p.xf_lo != nullptr;                 // split-precision activations: <= 4 tiles per workgroup (registers)
    const int RW = narrow ? pc_ceil_div(p.M, 32) : pc_ceil_div(p.M, 64);
    const dim3 block((RW + kStageWaves) * 64);
#define PC_ROWS(TTV)                                                                                       \
    do {                                                                                                   \
        constexpr int TV = (EPI == EPI_SILU) ? (TTV) / 2 : (TTV);                                          \
        const dim3 grid(pc_ceil_div(units, TV), p.kslices);                                                \
        if (two) {                                                                                         \
            if constexpr ((TTV) <= 4) {                                                                    \
                if (narrow) hipLaunchKernelGGL((gemm_rows_kernel<TTV, EPI, 2, true>), grid, block, 0, s, p); \
                else hipLaunchKernelGGL((gemm_rows_kernel<TTV, EPI, 4, true>), grid, block, 0, s, p);      \
            }                                                                                              \
        } else if (narrow) hipLaunchKernelGGL((gemm_rows_kernel<TTV, EPI, 2, false>), grid, block, 0, s, p); \
        else if constexpr ((TTV) <= 6) hipLaunchKernelGGL((gemm_rows_kernel<TTV, EPI, 4, false>), grid, block, 0, s, p); \
        return pc_check_launch("gemm_rows_kernel");                                                        \
    } while (0)
    // Wide panels (round 3): up to 288 rows (two blocks of them up to 512) with split-precision planes, a workgroup takes EIGHT weight tiles (128 columns; four
    // gate/up pairs) instead of four.  What bounds this kernel at a few hundred rows is the activation planes every workgroup
    // re-reads from L2 (13b, 256 rows: 5.2 MB per workgroup against 0.65 MB of weights; launch time = HBM time of the weights +
    // 1.2 us per row): half as many workgroups re-read half as much.  The accumulators of eight tiles need 156 registers, i.e.
    // at most 12 waves per workgroup: 9 compute waves (288 rows) + THREE staging waves, 3 k-steps per stage, activation prefetch
    // distance 2.  The caller fills the chip with K slices instead of narrow panels (o_proj / down_proj: slabs; see
    // rows_kslices in model/llama_hip.py).  13b layer at 259 rows: 491 -> 338 us (tools/rows_bench.py).
    {
        static const bool no_wide = [] { const char* e = getenv("PC_ROWS_WIDE"); return e && e[0] == '0'; }();
        static const bool wide_rope = [] { const char* e = getenv("PC_ROWS_WIDE_ROPE"); return e && e[0] == '1'; }();   // (q|k|v keeps four tiles: no K split under its epilogue)
        const int mt = pc_ceil_div(p.M, 16);
        // 289..512 rows: two row blocks per column panel (grid.z), each within the 9 compute waves -- the weights of a panel are
        // then read twice (once from L2), the activations of a row block by half as many workgroups
        const int nz = mt <= 18 ? 1 : 2;
        if (two && !no_wide && !forced && mt <= 36 && (EPI != EPI_ROPE || wide_rope || nz == 2)) {
            constexpr int TV = (EPI == EPI_SILU) ? 4 : 8;
            GemmParams pz = p;
            pz.zrows = nz == 1 ? 0 : 16 * pc_ceil_div(mt, 2);
            const int rw = pc_ceil_div(pc_ceil_div(mt, nz), 2);
#ifdef PC_DEV_SWEEPS
            // (LDS-DMA ring staging, round 6: the same time as register staging in the step -- config 4 21.96 / 21.93 ms against
            // 21.93 / 21.83 -- because a CU's weight stream is capped near 13.5 GB/s whatever is in flight; profiles/r06_variants.txt r6s)
            static const int ring = [] { const char* e = getenv("PC_ROWS_RING"); return e ? atoi(e) : 0; }();
            if (ring == 6)
                hipLaunchKernelGGL((gemm_rows_kernel<8, EPI, 2, true, 3, 3, 2, 768, 6>), dim3(pc_ceil_div(units, TV), p.kslices, nz),
                                   dim3((rw + 3) * 64), 0, s, pz);
            else if (ring == 3)
                hipLaunchKernelGGL((gemm_rows_kernel<8, EPI, 2, true, 3, 6, 2, 768, 3>), dim3(pc_ceil_div(units, TV), p.kslices, nz),
                                   dim3((rw + 3) * 64), 0, s, pz);
            else
#endif
            // an odd number of row tiles in one block (259 rows = 17 tiles): the odd tile goes to the three staging waves (XT: LDS-DMA
            // ring of five 30-KiB stages) instead of a ninth compute wave on the SIMD that already runs two.  PC_ROWS_XT=0: off
            static const bool xt_on = [] { const char* e = getenv("PC_ROWS_XT"); return !(e && e[0] == '0'); }();
            if (xt_on && nz == 1 && (mt & 1) && mt >= 5) {
                hipLaunchKernelGGL((gemm_rows_kernel<8, EPI, 2, true, 3, 3, 2, 768, 5, true>), dim3(pc_ceil_div(units, TV), p.kslices, 1),
                                   dim3((mt / 2 + 3) * 64), 0, s, pz);
                return pc_check_launch("gemm_rows_kernel");
            }
            hipLaunchKernelGGL((gemm_rows_kernel<8, EPI, 2, true, 3, 3, 2, 768>), dim3(pc_ceil_div(units, TV), p.kslices, nz),
                               dim3((rw + 3) * 64), 0, s, pz);
            return pc_check_launch("gemm_rows_kernel");
        }
    }
    const int work = units * p.kslices;
    if constexpr (EPI == EPI_SILU) {
        if (forced == 4 || two || (!forced && pc_ceil_div(work, 2) <= 256)) PC_ROWS(4);
        PC_ROWS(6);          // (8 tiles per workgroup spill at the 128 registers a 16-wave workgroup leaves)
    } else {
        if (forced == 3 || (!forced && pc_ceil_div(work, 3) <= 256)) PC_ROWS(3);
        if (forced == 4 || two || (!forced && pc_ceil_div(work, 4) <= 256)) PC_ROWS(4);
        PC_ROWS(6);
    }
#undef PC_ROWS
}
int launch_rows_epi(int epi, const GemmParams& p, int units, hipStream_t s) {
    switch (epi) {
        case EPI_STORE: return launch_rows<EPI_STORE>(p, units, s);
        case EPI_ADD: return launch_rows<EPI_ADD>(p, units, s);
        case EPI_SILU: return launch_rows<EPI_SILU>(p, units, s);
        case EPI_ROPE: return launch_rows<EPI_ROPE>(p, units, s);
        default: return launch_rows<EPI_GELU>(p, units, s);
    }
}

}  // namespace pcg
