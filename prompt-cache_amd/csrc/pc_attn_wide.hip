// A long question over a LONG staged cache (BASELINE config 4: 259 new rows over 8 258 staged keys at the 13b shape): the STAGED
// keys of a many-row split-precision launch at head_dim 128 -- ALL query rows of a head (up to 288) in one workgroup.
//
// What attn_ring_kernel (pc_attn_ring.hip) pays at that shape: 128 query rows per workgroup make 259 rows three q-blocks per head
// (128 + 128 + 3), each of which streams the head's whole K / V through its own ring -- three times the L2 -> LDS traffic, DMA
// issue and stage barriers for the same MFMAs, a third block whose eight waves carry three rows, and two KV splits where 256 rows
// run three (profiles/r05_config4_kernel_stats.txt: 167 us, 28 % MFMA-busy).  Here a workgroup is every query row of one head
// (more than 288 rows: balanced q-blocks) against one slice of the staged keys:
//   * 8 waves, two per SIMD; wave w owns the 16-row groups w, w + 8, w + 16 (17 groups = 3 + 2 + ... + 2: five per SIMD on one
//     SIMD, four on the others).  A K / V tile crosses L2 -> LDS once per head, and the stage barrier and the DMA issue are paid
//     once per 17 groups instead of once per 8.
//   * the first two groups of a wave are one 32-row UNIT: every K / V^T fragment read out of LDS feeds both (a wave issues one
//     instruction per ~4 cycles, and with one group per read the fragment reads, their addresses and waits made a group-tile ~330
//     instructions for 64 MFMAs: issue-bound above the matrix pipe's time).
//   * a wave's 256 registers hold the O^T accumulators of its three groups (96) and the hi plane of their Q fragments (48); the lo
//     plane of Q waits in LDS in fragment order, and the QK^T walk keeps one k-step of the K tile live at a time.
//   * the softmax defers the running-max update until a row's maximum has grown by more than 2^8 (P stays below 256: exact in the
//     (hi, lo) fp16 pair, fp32 accumulation): the O^T rescale all but disappears; cross-lane maxima by v_permlane16/32_swap (VALU)
//     instead of ds_bpermute round trips; row sums stay per lane until the epilogue.
//   * tiles travel by LDS-DMA into a two-stage ring of 32-KiB stages (one 64-key tile: K, V), one barrier per tile;
//     stage-while-reading (pc_attn gather_rows) as in the ring kernel: sources from the row table, rows not in the arena yet are
//     stored from the stage that just landed.
//   * only STAGED keys (exact fp16, every one visible to every row): no residual planes, no causal mask -- the keys this pass
//     appended go to attn_ring_kernel in its own-rows mode as one more partial, and attn_combine_kernel merges (pc_attn.hip).
// Replaces LlamaAttention.forward's core over the cached keys, promptcache/model/llama2.py:368-398.
// Roofline: MFMA.  flops = 4 * H * D * q * staged keys * (2 planes of Q / P).
#include <type_traits>
#include <utility>

#include "pc_attn_common.h"

namespace pca {
namespace {

// dev (tools/plane_audit.py, precision audit only): the staged keys' QK^T without Q's lo plane / V^T P^T without P's lo plane
#ifndef PC_WIDE_NOQLO
#define PC_WIDE_NOQLO 0
#endif
#ifndef PC_WIDE_NOPLO
#define PC_WIDE_NOPLO 0
#endif
#ifndef PC_WIDE_EXP
#define PC_WIDE_EXP 0          // dev probes (timing attribution only; results are wrong): 1 no V^T P^T MFMAs, 2 no softmax arithmetic
#endif
constexpr int WD = 128, WKS = WD / 32, WDB = WD / 16, WCPR = WD / 8;
constexpr int kWThreads = 512, kWaves = 8;
constexpr int kWR = 3;                       // row groups per wave (at most; the launch's count is a template parameter)
constexpr int kWGroups = 18;                 // row groups per workgroup (at most: 288 rows; what the Q planes in LDS hold)
constexpr int kWPlane = kTK * WD * 2;        // 16 KiB: one 64-key plane
constexpr int kWStage = 2 * kWPlane;         // K, V
constexpr int kWNst = 2;
constexpr float kDeferLog2 = 8.f;            // the running max moves when a row's maximum has grown by more than this (log2 units)

template <int I> using ic = std::integral_constant<int, I>;
template <class F, int... Is> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(ic<Is>{}), ...); }
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// (single instructions: plain fmaxf on MFMA results gets a canonicalising v_max_f32 x, x, x in front of it)
__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// R: row groups per wave of this launch = ceil(groups per q-block / 8); groups are dealt round robin, so a wave owns R or R - 1
// of them (fewer only in a short last q-block): the walk is written for R groups, and what a wave computes for a group it does
// not own (zero Q fragments) is never stored.
template <bool GATHER, int R>
__global__ __launch_bounds__(kWThreads) void attn_wide_kernel(const AttnParams p) {
    __shared__ __attribute__((aligned(16))) char lds[kWNst * kWStage];
    constexpr int D = WD, KS = WKS, DB = WDB, CPR = WCPR;

    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, gq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // grid: x = q-block, y = head, z = slice of the staged keys
    const int qblk = blockIdx.x, h = blockIdx.y, split = blockIdx.z;
    const int hkv = h / (p.H / p.Hkv);
    const int q_len = p.q_len;
    int past_len_v = p.past_len;
    asm volatile("" : "+s"(past_len_v));
    if (p.past_len_dev) past_len_v = *p.past_len_dev;
    const int past_len = __builtin_amdgcn_readfirstlane(past_len_v);

    // ---- this workgroup's rows: groups of 16, dealt to the waves round robin ----
    const int ngroups = (q_len + 15) >> 4;
    const int gpb = (ngroups + p.nqblk - 1) / p.nqblk;          // groups per q-block (host: <= kWaves * kWR)
    const int g0 = qblk * gpb;
    int ng_blk = ngroups - g0;
    ng_blk = ng_blk < gpb ? ng_blk : gpb;
    const int my_ng = (ng_blk - wave + kWaves - 1) / kWaves;     // groups of this wave (wave-uniform; may be 0)

    // ---- this slice's tiles ----
    const int ntiles_all = (past_len + kTK - 1) / kTK;
    const int tps = (ntiles_all + p.wide_nsplit - 1) / p.wide_nsplit;
    const int t0 = split * tps;
    int nt = ntiles_all - t0;
    nt = nt < tps ? nt : tps;
    nt = nt < 0 ? 0 : nt;
    const int last_key = past_len - 1;

    // ---- Q fragments of every group of this wave.  Registers: the hi plane of its first two groups.  LDS, in FRAGMENT order (each
    // lane reads back the 16 bytes it wrote: wave-private regions, no bank conflicts, no barrier): the lo plane of all of them and
    // both planes of the third group (only waves 0 and 1 can have one: kWGroups = 18) ----
    constexpr int RQ = R < 2 ? R : 2;
    constexpr int NQL = kWaves * R < kWGroups ? kWaves * R : kWGroups;
    __shared__ __attribute__((aligned(16))) char qlo_lds[NQL * KS * 1024];
    __shared__ __attribute__((aligned(16))) char qhi2_lds[R == 3 ? (kWGroups - 2 * kWaves) * KS * 1024 : 16];
    char* const my_qlo = qlo_lds + wave * (KS * 1024) + lane * 16;        // + group g of this wave: (g * kWaves) * KS * 1024
    [[maybe_unused]] char* const my_qhi2 = qhi2_lds + (wave & 1) * (KS * 1024) + lane * 16;
    h8 qf[RQ][KS];
    static_for<R>([&](auto gi) {
        constexpr int g = decltype(gi)::value;
        const int qi = (g0 + g * kWaves + wave) * 16 + n;
        const bool visited = g + 1 < R || my_ng == R;          // (a group the walk visits: the planes hold kWGroups groups, not 8 R)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            h8 z = {0, 0, 0, 0, 0, 0, 0, 0}, zl = z;
            if (g < my_ng && qi < q_len) {
                const int64_t off = (int64_t)qi * p.q_ts + (int64_t)h * D + ks * 32 + gq * 8;
                z = *(const h8*)(p.q + off);
                zl = *(const h8*)(p.q_lo + off);
            }
            if constexpr (g < RQ) qf[g][ks] = z;
            else if (visited) *(h8*)(my_qhi2 + ks * 1024) = z;
            if (visited) *(h8*)(my_qlo + (g * kWaves * KS + ks) * 1024) = zl;
        }
    });
    f4 o[R][DB];
    float m_run[R], l_run[R];
    static_for<R>([&](auto gi) {
        constexpr int g = decltype(gi)::value;
#pragma unroll
        for (int db = 0; db < DB; ++db) { f4 z = {0.f, 0.f, 0.f, 0.f}; o[g][db] = z; }
        m_run[g] = kNegBig; l_run[g] = 0.f;
    });

    // ---- staging: wave-instruction j (0, 1) of a plane covers rows 4 (8 j + wave) .. + 4 (1 KiB); lane l: row + (l >> 4), position
    // l & 15.  (row & 15 = (4 wave + (l >> 4)) & 15 for both j: one swizzle per lane)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    __shared__ __attribute__((aligned(16))) char gtab[GATHER ? kWaves * 2048 : 16];
    [[maybe_unused]] const uint32_t gtab0 = GATHER ? (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)gtab + wave * 2048 : 0;
    [[maybe_unused]] uint64_t g_nost[3][2] = {{~0ull, ~0ull}, {~0ull, ~0ull}, {~0ull, ~0ull}};    // do-not-store lane masks: [0] the tile being multiplied, [1] landing, [2] just issued
    auto lane_row = [&](int l) { return 4 * wave + (l >> 4); };                     // + 32 j
    auto lane_koff = [&](int l) { const int r = lane_row(l), c = l & 15; return (c ^ (r & 15)) << 3; };           // K: position c holds chunk c ^ (row & 15)
    auto lane_voff = [&](int l) { const int r = lane_row(l), c = l & 15; return ((c - 2 * (r & 7)) & 15) << 3; }; // V: chunk (c - 2 (row & 7)) mod 16
    const _Float16* kbase = p.k + (int64_t)hkv * p.kv_hs;
    const _Float16* vbase = p.v + (int64_t)hkv * p.kv_hs;
    [[maybe_unused]] auto fetch_entries = [&](int t) {       // row-table entries of tile t0 + t -> gtab (2 x 1 KiB per wave)
        if constexpr (GATHER) {
            if (t >= nt) return;
            int l = lane;
            asm volatile("" : "+v"(l));
            const int key0 = (t0 + t) * kTK;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int key = key0 + 32 * j + lane_row(l);
                glds16_raw((const _Float16*)(p.rows + (key < last_key ? key : last_key)), gtab0 + j * 1024);
            }
        }
    };
    auto issue = [&](int t, uint32_t buf) {
        const int key0 = (t0 + t) * kTK;
        int l = lane;
        asm volatile("" : "+v"(l));                     // (recomputed from the lane id at each use: no registers held across the arithmetic)
        const int row = lane_row(l), koff = lane_koff(l), voff = lane_voff(l);
        [[maybe_unused]] u32x4 ent[GATHER ? 2 : 1];
        if constexpr (GATHER) {
#pragma unroll
            for (int j = 0; j < 2; ++j) ent[j] = *(const u32x4*)(gtab + wave * 2048 + j * 1024 + lane * 16);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t dst = buf + (8 * j + wave) * 1024;
            const int key = key0 + 32 * j + row;
            if constexpr (GATHER) {
                const u32x4 e = ent[j];
                const uint64_t base = ((uint64_t)e[1] << 32) | e[0];
                const uint64_t ka = base + (((uint64_t)(uint32_t)(p.g_kplane + hkv) * e[2]) << 4) + (uint32_t)(koff * 2);
                const uint64_t va = base + (((uint64_t)(uint32_t)(p.g_vplane + hkv) * e[2]) << 4) + (uint32_t)(voff * 2);
                const uint32_t nostore = (key <= last_key && !(e[3] & PC_KV_ROW_STAGED)) ? 0u : 1u;
                g_nost[2][j] = __ballot(nostore);
                glds16_raw((const _Float16*)(uintptr_t)ka, dst);
                glds16_raw((const _Float16*)(uintptr_t)va, dst + kWPlane);
            } else {
                const int kc = key < last_key ? key : last_key;
                glds16_raw(kbase + (int64_t)kc * D + koff, dst);
                glds16_raw(vbase + (int64_t)kc * D + voff, dst + kWPlane);
            }
        }
    };

    // dev (pc_dev_attn_trace): wall-clock stamps of ONE tile of workgroup (0, 0, 0), per wave: [wave][16]
    int tr_slot = -1;
    auto stamp = [&]() {
        if (p.trace && tr_slot >= 0 && tr_slot < 16 && lane == 0) p.trace[wave * 16 + tr_slot] = __builtin_readcyclecounter();
        if (tr_slot >= 0) ++tr_slot;
    };
    // ---- a UNIT of NG (1 or 2) row groups against one tile: the groups of a unit share every K and V^T fragment read ----
    // A wave issues one instruction per ~4 cycles: with one group per fragment read a group-tile is ~330 instructions for its 64
    // MFMAs (K / V^T / Q-lo fragment reads, their address arithmetic and waits are half of them) and the wave is ISSUE-bound at
    // ~1 300 cycles where the matrix pipe needs 1 024 (per-phase stamps: tools/wide_trace.py).  Two groups per read halve that part.
    const int v_rowb = (gq * 4 + (n >> 2)) * (D * 2), v_colb = (n & 3) * 8 + 32 * ((gq * 4 + (n >> 2)) & 7);      // (bytes; see pv_unit)
    auto qk_unit = [&](const _Float16* Kl, auto g0i, auto ngi, f4 (&acc)[2][4]) {
        constexpr int G0 = decltype(g0i)::value, NG = decltype(ngi)::value;
        // (the lane id goes through an opaque register per unit: otherwise hipcc merges the fragment reads of all units of a tile
        // -- the same LDS addresses -- and keeps a whole tile live next to the groups' state)
        int nk = n;
        asm volatile("" : "+v"(nk));
#pragma unroll
        for (int u = 0; u < NG; ++u)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) { f4 z = {0.f, 0.f, 0.f, 0.f}; acc[u][kb] = z; }
        h8 ka[2][4], ql[2][NG];
        [[maybe_unused]] h8 qh2[2];
        auto reads = [&](int ks, int set) {
#pragma unroll
            for (int u = 0; u < NG; ++u) ql[set][u] = *(const h8*)(my_qlo + ((G0 + u) * kWaves * KS + ks) * 1024);
            if constexpr (G0 >= 2) qh2[set] = *(const h8*)(my_qhi2 + ks * 1024);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const int row = kb * 16 + nk;
                ka[set][kb] = *(const h8*)(Kl + row * D + (((ks * 4 + gq) ^ (row & (CPR - 1))) << 3));
            }
        };
        // k-step outer: 4 NG independent accumulate chains (consecutive MFMAs never share an accumulator), the fragments of
        // k-step ks + 1 are read while k-step ks is multiplied
        reads(0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4 + NG + (G0 >= 2 ? 1 : 0), 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) { reads(ks + 1, (ks + 1) & 1); __builtin_amdgcn_sched_group_barrier(0x100, 4 + NG + (G0 >= 2 ? 1 : 0), 1); }
#pragma unroll
            for (int pl_ = 0; pl_ < (PC_WIDE_NOQLO ? 1 : 2); ++pl_)
#pragma unroll
                for (int u = 0; u < NG; ++u)
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb)
                    {
                        h8 bq;
                        if constexpr (G0 >= 2) bq = pl_ ? ql[ks & 1][u] : qh2[ks & 1];
                        else bq = pl_ ? ql[ks & 1][u] : qf[G0 + u][ks];
                        acc[u][kb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka[ks & 1][kb], bq, acc[u][kb], 0, 0, 0);
                    }
            __builtin_amdgcn_sched_group_barrier(0x008, 8 * NG, 1);          // (the issue order as written: hipcc sinks the reads to their uses otherwise)
        }
    };
    // online softmax of one group's raw scores, branch-free part: P as (hi, lo) operand pairs, this lane's row-sum share, the new
    // running max (it moves only past kDeferLog2) -- written to sit in ONE scheduling region beside the next group's QK^T MFMAs
    auto soft_core = [&](const f4 (&acc)[4], auto gi, h8 (&pb)[2], h8 (&pbl)[2], float& rs, float& m_new, bool& moved) {
        constexpr int g = decltype(gi)::value;
        const float c_ = p.scale_log2;
        float mx = vmax3(vmax3(acc[0][0], acc[0][1], acc[0][2]), vmax3(acc[0][3], acc[1][0], acc[1][1]), vmax(acc[1][2], acc[1][3]));
        mx = vmax3(mx, vmax3(acc[2][0], acc[2][1], acc[2][2]), vmax3(acc[2][3], acc[3][0], acc[3][1]));
        mx = vmax3(mx, acc[3][2], acc[3][3]);
        {
            const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = vmax(__uint_as_float(s16[0]), __uint_as_float(s16[1]));
            const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = vmax(__uint_as_float(s32[0]), __uint_as_float(s32[1]));
        }
        const float m_old = m_run[g];
        moved = (mx - m_old) * c_ > kDeferLog2;
        m_new = moved ? mx : m_old;
        const float mc = m_new * c_;
        rs = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            float e[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                e[r] = fast_exp2(__builtin_fmaf(acc[kb][r], c_, -mc));
                rs += e[r];
            }
            uint32_t h01, l01, h23, l23;
            split_pair(e[0], e[1], h01, l01);
            split_pair(e[2], e[3], h23, l23);
            const u32x2 hv = {h01, h23}, lv = {l01, l23};
            const h4 hq = __builtin_bit_cast(h4, hv), lq = __builtin_bit_cast(h4, lv);
#pragma unroll
            for (int r = 0; r < 4; ++r) { pb[kb >> 1][(kb & 1) * 4 + r] = hq[r]; pbl[kb >> 1][(kb & 1) * 4 + r] = lq[r]; }
        }
    };
    // ... and the rare part: rows whose running max moved rescale their accumulators
    auto soft_fin = [&](auto gi, const float rs, const float m_new, const bool moved) {
        constexpr int g = decltype(gi)::value;
        if (__any(moved)) {
            const float alpha = fast_exp2((m_run[g] - m_new) * p.scale_log2);          // (1 for the rows that did not move)
            l_run[g] = l_run[g] * alpha;
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                o[g][db][0] *= alpha; o[g][db][1] *= alpha; o[g][db][2] *= alpha; o[g][db][3] *= alpha;
            }
            m_run[g] = m_new;
        }
        l_run[g] += rs;                                              // (this lane's 16 keys; summed over the row's four lanes in the epilogue)
    };
    auto mask_scores = [&](f4 (&acc)[4], const int key0) {          // the one tile of the launch that holds the last staged key
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (key0 + kb * 16 + gq * 4 + r > last_key) acc[kb][r] = -INFINITY;
    };
    auto pv_unit = [&](const _Float16* Vl, auto g0i, auto ngi, const h8 (&pb)[2][2], const h8 (&pbl)[2][2]) {
        constexpr int G0 = decltype(g0i)::value, NG = decltype(ngi)::value;
        // register sets of V^T fragments: the next pair of head-dim blocks is read while this one is multiplied; a ONE-group unit
        // multiplies a pair in 8 MFMAs (128 cycles: half an LDS round trip) and reads two pairs ahead where the registers allow
        // (R < 3: 94.7 -> 92.0 us at 130 rows, profiles/r06_variants.txt)
        constexpr int NS = (NG == 1 && R < 3) ? 3 : 2;
        h4 va[NS][2][4];
        // V^T fragment of head-dim block db, key step t: row t * 32 + gq * 4 + (n >> 2) of the plane, 8 bytes at column
        // (db * 32 + (n & 3) * 8 + 32 (row & 7)) mod 256 -- one per-lane row base (opaque per unit: see qk_unit) and a column that
        // walks 32 bytes per block; the key step and the second half of the fragment are immediate offsets
        int vb = v_rowb;
        asm volatile("" : "+v"(vb));
        const char* const vrow_p = (const char*)Vl + vb;
        auto load_pair = [&](int dp) {
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const char* q = vrow_p + ((v_colb + (2 * dp + d) * 32) & 255);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    va[dp % NS][d][2 * t] = lds_tr_read((const _Float16*)(q + t * 32 * D * 2));
                    va[dp % NS][d][2 * t + 1] = lds_tr_read((const _Float16*)(q + t * 32 * D * 2 + 16 * D * 2));
                }
            }
        };
        // head-dim blocks in PAIRS: the 8 NG MFMAs of a pair walk its 2 NG accumulators in turn
#pragma unroll
        for (int dp = 0; dp < NS - 1; ++dp) { load_pair(dp); __builtin_amdgcn_sched_group_barrier(0x100, 8, 2); }
#pragma unroll
        for (int dp = 0; dp < DB / 2; ++dp) {
            if (dp + NS - 1 < DB / 2) { load_pair(dp + NS - 1); __builtin_amdgcn_sched_group_barrier(0x100, 8, 2); }
#if PC_WIDE_EXP != 1
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int pl_ = 0; pl_ < (PC_WIDE_NOPLO ? 1 : 2); ++pl_)
#pragma unroll
                    for (int u = 0; u < NG; ++u)
#pragma unroll
                        for (int d = 0; d < 2; ++d) {
                            const h4 lo = va[dp % NS][d][2 * t], hi = va[dp % NS][d][2 * t + 1];
                            const h8 a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                            o[G0 + u][2 * dp + d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pl_ ? pbl[u][t] : pb[u][t], o[G0 + u][2 * dp + d], 0, 0, 0);
                        }
#else
            asm volatile("" :: "v"(va[dp % NS][0][0]), "v"(va[dp % NS][0][3]), "v"(va[dp % NS][1][0]), "v"(va[dp % NS][1][3]));
#endif
            __builtin_amdgcn_sched_group_barrier(0x008, 8 * NG, 2);
        }
    };
    // one unit against one tile: scores of its groups, their softmax one after the other, then V^T P^T
    auto unit = [&](const _Float16* Kl, const _Float16* Vl, const bool masked, const int key0, auto g0i, auto ngi) {
        constexpr int G0 = decltype(g0i)::value, NG = decltype(ngi)::value;
        f4 acc[2][4];
        h8 pb[2][2], pbl[2][2];
        stamp();
        qk_unit(Kl, g0i, ngi, acc);
        stamp();
        static_for<NG>([&](auto ui) {
            constexpr int u = decltype(ui)::value;
            float rs, m_new;
            bool moved;
            if (masked) mask_scores(acc[u], key0);
            soft_core(acc[u], ic<G0 + u>{}, pb[u], pbl[u], rs, m_new, moved);
            soft_fin(ic<G0 + u>{}, rs, m_new, moved);
        });
        stamp();
        pv_unit(Vl, g0i, ngi, pb, pbl);
        stamp();
    };
    // all groups of this wave against one tile
    auto tile = [&](const _Float16* Kl, const _Float16* Vl, const bool masked, const int key0) {
        if constexpr (R == 1) {
            unit(Kl, Vl, masked, key0, ic<0>{}, ic<1>{});
        } else {
            // (R = 2: a wave that owns one group walks one -- nine groups are 2 + 1 + ... + 1, not eight pairs)
            if (R == 3 || my_ng == 2) unit(Kl, Vl, masked, key0, ic<0>{}, ic<2>{});
            else unit(Kl, Vl, masked, key0, ic<0>{}, ic<1>{});
            if constexpr (R == 3) {
                if (my_ng == 3) unit(Kl, Vl, masked, key0, ic<2>{}, ic<1>{});
            }
        }
    };

    // ---- the ring: two stages; while tile i is multiplied, tile i + 1 is in flight; one barrier per tile ----
    if (nt > 0) {
        if constexpr (GATHER) { fetch_entries(0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        issue(0, lds0);
        if constexpr (GATHER) { g_nost[0][0] = g_nost[2][0]; g_nost[0][1] = g_nost[2][1]; }
        if (nt > 1) {
            if constexpr (GATHER) { fetch_entries(1); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            issue(1, lds0 + kWStage);
            if constexpr (GATHER) {
                g_nost[1][0] = g_nost[2][0]; g_nost[1][1] = g_nost[2][1];
                fetch_entries(2);
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");    // (the entry DMA sits behind the tile DMA in the queue: drain)
            } else {
                asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");    // tile 0 landed, everyone's
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
#pragma unroll
        for (int g = 0; g < RQ; ++g)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) asm volatile("" ::"v"(qf[g][ks]));     // (hipcc's vmcnt(0) for the Q loads lands here, not inside the loop)
        for (int i = 0; i < nt; ++i) {
            const char* buf = lds + (i & 1) * kWStage;
            const int key0 = (t0 + i) * kTK;
            if constexpr (GATHER) {
                // ---- tile i has landed: the rows of it that are not in the arena yet leave for it (each wave the 4 KiB it DMA'd) ----
                _Float16* kd = const_cast<_Float16*>(p.k) + (int64_t)hkv * p.kv_hs;
                _Float16* vd = const_cast<_Float16*>(p.v) + (int64_t)hkv * p.kv_hs;
                if (h % (p.H / p.Hkv) == 0 && qblk == 0) {
                    int l = lane;
                    asm volatile("" : "+v"(l));
                    const int row = lane_row(l), koff = lane_koff(l), voff = lane_voff(l);
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if (!((g_nost[0][j] >> lane) & 1ull)) {
                            const char* src = buf + (8 * j + wave) * 1024 + lane * 16;
                            const u32x4 kc = *(const u32x4*)(src);
                            const u32x4 vc = *(const u32x4*)(src + kWPlane);
                            const int64_t key = key0 + 32 * j + row;
                            __builtin_nontemporal_store(kc, (u32x4*)(kd + key * D + koff));
                            __builtin_nontemporal_store(vc, (u32x4*)(vd + key * D + voff));
                        }
                    }
                }
            }
            tr_slot = (p.trace && i == 8 && blockIdx.x + blockIdx.y + blockIdx.z == 0) ? 0 : -1;
            if (my_ng > 0) tile((const _Float16*)buf, (const _Float16*)(buf + kWPlane), key0 + kTK - 1 > last_key, key0);
            if (tr_slot >= 0) tr_slot = 12;
            stamp();
            if (i + 1 < nt) {
                // tile i + 1 has landed (this wave's DMA drained, everyone's via the barrier) and every wave is done reading tile i:
                // its stage takes tile i + 2, which then has the whole of tile i + 1's arithmetic to arrive
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                stamp();
                asm volatile("s_barrier" ::: "memory");
                stamp();
                if (i + 2 < nt) issue(i + 2, lds0 + (i & 1) * kWStage);
                stamp();
                if constexpr (GATHER) {
                    g_nost[0][0] = g_nost[1][0]; g_nost[0][1] = g_nost[1][1];
                    g_nost[1][0] = g_nost[2][0]; g_nost[1][1] = g_nost[2][1];
                    fetch_entries(i + 3);
                }
            }
        }
    }

    // ---- epilogue: one partial per (row, slice) for attn_combine_kernel ----
    // (lane-derived indices recomputed from an opaque copy of the lane id: hipcc otherwise computes the slots in front of the loop
    // and parks them in scratch across it)
    int le = lane;
    asm volatile("" : "+v"(le));
    const int ne = le & 15, gqe = le >> 4;
    static_for<R>([&](auto gi) {
        constexpr int g = decltype(gi)::value;
        const int qi = (g0 + g * kWaves + wave) * 16 + ne;
        if (g < my_ng && qi < q_len) {
            float l = l_run[g];
            l += __shfl_xor(l, 16);
            l += __shfl_xor(l, 32);
            const int64_t slot = ((int64_t)h * p.nsplit + split) * q_len + qi;
            float* po = p.part_o + slot * D + gqe * 4;
#pragma unroll
            for (int db = 0; db < DB; ++db) *(f4*)(po + db * 16) = o[g][db];
            if (gqe == 0) { p.part_ml[slot * 2] = m_run[g] * p.scale_log2; p.part_ml[slot * 2 + 1] = l; }
        }
    });
}

}  // namespace

// Launches the wide kernel takes the STAGED keys of: what the ring kernel takes (head_dim 128, split-precision Q, no ALiBi), one batch
// row, one past length, no shared prefix, residual planes for the pass's own rows -- and a shape where it pays: the three launches
// (slices, own rows, merge of up to 16 partials) cost ~25 us more fixed time than the ring kernel's two, which the shared K / V pass
// earns back from ~6 k staged keys and 128 rows on (tools/wide_sweep.py, profiles/r06_wide_sweep.txt: 40 heads over 8 258 keys,
// 130 / 259 / 288 / 512 rows: 0.89 / 0.84 / 0.84 / 0.92 x the ring kernel's time; 32 heads over 4 390 keys: 1.0-1.2 x, over
// 2 048: 1.2-1.4 x).  PC_ATTN_WIDE_MIN / PC_ATTN_WIDE_MIN_ROWS move the thresholds, PC_ATTN_NO_WIDE=1 switches it off.
bool wide_eligible(const AttnParams& p, int D, int B) {
    const char* e = getenv("PC_ATTN_NO_WIDE");            // (read per call: tests and probes switch it inside one process)
    if (e && e[0] == '1') return false;
    static const int min_rows = [] { const char* m = getenv("PC_ATTN_WIDE_MIN_ROWS"); return m ? atoi(m) : 128; }();
    return ring_eligible(p, D) && B == 1 && !p.past_lens && !p.pre_k && p.past_len >= wide_min_keys() && p.q_len >= min_rows &&
           p.k_lo != nullptr && p.lo_row0 < 0;
}
int wide_min_keys() {
    const char* m = getenv("PC_ATTN_WIDE_MIN");           // (per call, like PC_ATTN_NO_WIDE)
    return m ? atoi(m) : 6144;
}

int wide_nqblk(int q_len) { return pc_ceil_div(pc_ceil_div(q_len, 16), kWGroups); }      // (18 groups = 288 rows: the Q planes in LDS)

// slices of the staged keys: one workgroup per CU, at least two tiles per slice, at most 15 (+ the own-rows partial = 16)
int wide_nsplit(int H, int q_len, int past_len) {
    static const int forced = [] { const char* e = getenv("PC_ATTN_WIDE_NSPLIT"); return e ? atoi(e) : 0; }();
    const int units = H * wide_nqblk(q_len);
    int s = 256 / (units > 0 ? units : 1);
    const int by_len = pc_ceil_div(past_len, kTK) / 2;
    s = s < by_len ? s : by_len;
    if (forced > 0) s = forced;
    s = s < 15 ? s : 15;
    return s < 1 ? 1 : s;
}

// p.nsplit = wide slices + 1 (pc_attn.hip sized the workspace and points part_o / part_ml at it)
int launch_attn_wide(const AttnParams& p0, hipStream_t stream) {
    AttnParams p = p0;
    p.wide_nsplit = p.nsplit - 1;
    p.nqblk = wide_nqblk(p.q_len);
    const dim3 grid(p.nqblk, p.H, p.wide_nsplit), block(kWThreads);
    const int ngroups = pc_ceil_div(p.q_len, 16), gpb = pc_ceil_div(ngroups, p.nqblk), R = pc_ceil_div(gpb, kWaves);
#define PC_WIDE(RV)                                                                                       \
    do {                                                                                                  \
        if (p.rows) hipLaunchKernelGGL((attn_wide_kernel<true, RV>), grid, block, 0, stream, p);          \
        else hipLaunchKernelGGL((attn_wide_kernel<false, RV>), grid, block, 0, stream, p);                \
    } while (0)
    switch (R) {
        case 1: PC_WIDE(1); break;
        case 2: PC_WIDE(2); break;
        default: PC_WIDE(3); break;
    }
#undef PC_WIDE
    int rc = pc_check_launch("attn_wide_kernel");
    if (rc != PC_OK) return rc;
    // the keys this pass appended (residual planes, causal mask): the ring kernel over [past_len, kv_len) as the last partial
    AttnParams r = p0;
    r.rows = nullptr;
    r.own_only = 1;
    return launch_attn_ring(r, 1, stream);
}

}  // namespace pca
