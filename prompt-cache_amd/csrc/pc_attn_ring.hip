// Many-row split-precision attention for gfx950 (q_len > 64 at head_dim 128: long questions over a staged cache, the schema
// encode, no-cache prefill): 128 query rows per workgroup, K / V tiles by LDS-DMA into a two-stage ring.
//
// Same math and operand layout as attn_fwd_kernel (pc_attn.hip: S^T = K . Q^T, O^T += V^T . P^T, 16 query rows per wave,
// Q and P as split-precision pairs, residual tiles of the K / V rows that carry them) -- what changes is how the keys get to
// the matrix cores:
//   * workgroup = 8 waves = 128 query rows of one head (attn_fwd_kernel: 4 waves, 64 rows): a K / V tile crosses L2 -> LDS
//     once per 128 rows.  At 259 rows over 8.3 k staged keys (BASELINE config 4) attn_fwd_kernel moves 872 MB of tiles per
//     layer for 45 GFLOP -- it is bound by that stream, not by the MFMAs (19 % busy).
//   * tiles travel by global_load_lds (16 B per lane, lane-linear destination; the K XOR swizzle and the V row rotation are
//     applied to the SOURCE address of each lane): no staging registers (attn_fwd_kernel holds 64 VGPRs of them and sits at
//     exactly 256).
//   * one 64-KiB ring stage holds either TWO plain 64-key tiles (K0 V0 K1 V1: staged rows, exact fp16) or ONE tile with
//     its residual planes (K V Klo Vlo: rows this pass or this encode appended): always 8 DMA instructions per wave and
//     stage, so the vmcnt bookkeeping is the same for both.  Two stages = 128 KiB, one workgroup per CU, 2 waves per SIMD.
//     While stage i is multiplied, stage i+1 is in flight; one barrier per stage.
//   * the key range is walked in REGIONS with workgroup-uniform base pointers: [own rows, plain | own rows, with residuals]
//     or, behind a shared prefix, [prefix rows (all plain or all with residuals) | own rows, with residuals].  A region ends exactly where the next begins (tiles need no
//     64-key alignment: there is no ALiBi here), so no tile mixes rows with and without residuals and nothing is zero-filled;
//     rows past the end of a region are clamped to its last row (finite data) and masked.
// Replaces LlamaAttention.forward's core, promptcache/model/llama2.py:368-398 (see pc_attn.hip).
// Roofline: MFMA.  flops = 4 * H * D * q * keys_visible * (2 planes of Q / P) (+ 1/2 more on residual tiles).
#include <type_traits>

#include "pc_attn_common.h"

#ifndef PC_RING_EXP
#define PC_RING_EXP 0
#endif
#ifndef PC_RING_PRIO
#define PC_RING_PRIO 0
#endif
#ifndef PC_RING_PAIR
#define PC_RING_PAIR 1
#endif

namespace pca {
namespace {

constexpr int RD = 128;                 // head dim
constexpr int RKS = RD / 32, RDB = RD / 16, RCPR = RD / 8;
constexpr int kRingThreads = 512, kRingQB = 128;
constexpr int kPlane = kTK * RD * 2;    // bytes of one 64-key plane (16 KiB)
constexpr int kStage = 4 * kPlane;      // 64 KiB

// one region of the key walk (all workgroup-uniform): keys [a, e) are rows of k / v indexed BY KEY (bases are pre-shifted);
// lo != 0: every key of the region has a residual row in kl / vl
struct Region { const _Float16* k; const _Float16* v; const _Float16* kl; const _Float16* vl; int a, e, lo; };

// GATHER (pc_attn gather_rows; B = 1, no shared prefix): STAGE WHILE READING, as attn_small_kernel does for short prompts.  Keys of
// the PLAIN region are fetched from where their row-table entries say they lie (module stores for rows a fresh prompt stages, the
// arena itself for the rest); the workgroups of q-block 0 -- one per (kv head, KV split): together they walk every staged key once
// -- write the rows that are not in the arena yet, from the ring stage they just waited for (each wave stores the 8 KiB it
// DMA'd: its own lane-linear chunks, back through the swizzle).  Entries are fetched one stage ahead of the DMA that uses them
// (right behind the previous issue, consumed behind the loop's vmcnt(0)), so the ring's timing does not see the extra load level;
// the stores are waited for at the end of the stage, a stage's worth of MFMAs later.
template <bool KVLO, bool PRE, bool GATHER = false>
__global__ __launch_bounds__(kRingThreads) void attn_ring_kernel(const AttnParams p) {
    static_assert(!(GATHER && PRE), "the staging stream has no shared prefix");
    __shared__ __attribute__((aligned(16))) char lds[2 * kStage];
    constexpr int D = RD, KS = RKS, DB = RDB, CPR = RCPR;
    constexpr int NREG = (PRE || KVLO) ? 2 : 1;      // [prefix | own rows]  or  [own rows, plain | own rows, with residuals]

    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int qblk, h, b, split;
    // own_only (launch_attn_wide, pc_attn_wide.hip): one workgroup per (q-block, head) walks the keys THIS PASS appended,
    // [past_len, kv_len), and leaves the result as partial nsplit - 1 of nsplit (the staged keys are the wide kernel's slices)
    const int gns = p.own_only ? 1 : p.nsplit;
    if (p.xcd_remap) {
        // 1-D grid, XCD-aware (see attn_fwd_kernel): heads are dealt to XCDs, an XCD walks the q-blocks of one head after
        // another, heaviest first, so that head's K / V is served from that XCD's L2
        // (with KV splits a "pair" is (batch row, head, split): its q-blocks stream the same key range)
        const int nqb = p.nqblk, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int npairs = p.H * p.nbatch * gns;
        const int per_xcd = (npairs + 7) >> 3;
        const int pair = (slot / nqb) * 8 + xcd;
        if (slot / nqb >= per_xcd || pair >= npairs) return;
        qblk = nqb - 1 - (slot % nqb);
        const int bh = pair / gns;
        split = pair - bh * gns;
        b = bh / p.H; h = bh - b * p.H;
    } else {
        qblk = blockIdx.x; h = blockIdx.y;
        b = blockIdx.z / gns; split = blockIdx.z - b * gns;
    }
    const int hkv = h / (p.H / p.Hkv);
    const int q_len = p.q_len;
    // (the host value goes through an opaque register first: hipcc otherwise turns this into ONE load through a select
    // between the device pointers and the address of the kernel argument, which it has to park in a scratch slot for that)
    int past_len_v = p.past_len;
    asm volatile("" : "+s"(past_len_v));
    if (p.past_lens) past_len_v = p.past_lens[b];
    else if (p.past_len_dev) past_len_v = *p.past_len_dev;
    const int past_len = __builtin_amdgcn_readfirstlane(past_len_v);
    const int kv_len = past_len + q_len;

    // this split's key range clipped by what the workgroup can causally see
    int kps = (kv_len + gns - 1) / gns;
    kps = (kps + kTK - 1) / kTK * kTK;
    const int ks0 = p.own_only ? past_len : split * kps;
    if (p.own_only) split = p.nsplit - 1;
    const int wg_rows_end = (qblk * kRingQB + kRingQB < q_len) ? qblk * kRingQB + kRingQB : q_len;
    int kend = ks0 + kps;
    kend = kend < kv_len ? kend : kv_len;
    kend = kend < past_len + wg_rows_end ? kend : past_len + wg_rows_end;

    const int qrow0 = qblk * kRingQB + wave * 16;
    const int qi = qrow0 + n;
    const bool wave_active = qrow0 < q_len;
    const int wave_rows_end = (qrow0 + 16 < q_len) ? qrow0 + 16 : q_len;
    const int wave_vis_end = past_len + wave_rows_end;
    const int row_vis_end = (qi < q_len) ? past_len + qi + 1 : 0;

    // ---- regions ----
    // own rows: key `key` is row (key - pre) of k / v (pre = 0 without a shared prefix), residual rows from own_lo0 on
    const int pre = PRE ? past_len : 0;
    const int own_lo0 = !KVLO ? 0x7fffffff
                              : (PRE ? pre : (p.lo_row0 == -2 ? __builtin_amdgcn_readfirstlane(p.past_len_dev[1])
                                                              : (p.lo_row0 < 0 ? past_len : p.lo_row0)));
    Region reg[NREG];
    {
        int r = 0;
        const int a0 = ks0, e0 = kend < pre ? kend : pre;            // prefix keys of this split
        const int a1 = ks0 > pre ? ks0 : pre, e1 = kend;             // own keys of this split
        if (PRE) {
            const _Float16* pk = p.pre_k + (int64_t)hkv * p.pre_hs;
            const _Float16* pv = p.pre_v + (int64_t)hkv * p.pre_hs;
            if (KVLO) {
                // the prefix has residuals for all of its rows or for none (one region either way)
                const bool plo = p.pre_k_lo != nullptr;
                const _Float16* pkl = plo ? p.pre_k_lo + (int64_t)hkv * p.pre_hs : pk;
                const _Float16* pvl = plo ? p.pre_v_lo + (int64_t)hkv * p.pre_hs : pv;
                reg[r++] = Region{pk, pv, pkl, pvl, a0, e0, plo ? 1 : 0};
            } else {
                reg[r++] = Region{pk, pv, pk, pv, a0, e0, 0};
            }
        }
        const _Float16* ok = p.k + b * p.kv_bs + (int64_t)hkv * p.kv_hs - (int64_t)pre * D;
        const _Float16* ov = p.v + b * p.kv_bs + (int64_t)hkv * p.kv_hs - (int64_t)pre * D;
        if (KVLO) {
            const _Float16* okl = p.k_lo + b * p.lo_bs + (int64_t)hkv * p.lo_hs - (int64_t)own_lo0 * D;
            const _Float16* ovl = p.v_lo + b * p.lo_bs + (int64_t)hkv * p.lo_hs - (int64_t)own_lo0 * D;
            if (PRE) {
                reg[r++] = Region{ok, ov, okl, ovl, a1, e1, 1};      // (behind a shared prefix every own row has its residual)
            } else {
                int m = own_lo0 < a1 ? a1 : own_lo0;
                m = m > e1 ? e1 : m;
                reg[r++] = Region{ok, ov, ok, ov, a1, m, 0};
                reg[r++] = Region{ok, ov, okl, ovl, m, e1, 1};
            }
        } else {
            reg[r++] = Region{ok, ov, ok, ov, a1, e1, 0};
        }
    }
    // stages per region: plain regions advance 128 keys per stage (two tiles), residual regions 64
    int nst_r[NREG], nst = 0;
#pragma unroll
    for (int r = 0; r < NREG; ++r) {
        const int len = reg[r].e - reg[r].a;
        nst_r[r] = len <= 0 ? 0 : (reg[r].lo ? (len + kTK - 1) / kTK : (len + 2 * kTK - 1) / (2 * kTK));
        nst += nst_r[r];
    }

    // ---- Q fragments (hi, lo) ----
    h8 qf[KS], qfl[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        qf[ks] = z; qfl[ks] = z;
        if (wave_active && qi < q_len) {
            const int64_t off = b * p.q_bs + (int64_t)qi * p.q_ts + (int64_t)h * D + ks * 32 + g * 8;
            qf[ks] = *(const h8*)(p.q + off);
            if (p.q_lo) qfl[ks] = *(const h8*)(p.q_lo + off);
        }
    }
    f4 o[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) { f4 z = {0.f, 0.f, 0.f, 0.f}; o[db] = z; }
    float m_run = kNegBig, l_run = 0.f;

    // ---- staging: lane -> (row of the plane, chunk position), source chunk by the tile's swizzle ----
    // wave-instruction j of a plane covers rows 4 * (8 j + wave) .. + 4 (1 KiB); lane l: row + (l >> 4), position l & 15
    // (GATHER: recomputed from the lane id at each use -- six registers the arithmetic phase needs more)
    struct LaneSlot { int srow[2], skoff[2], svoff[2]; };
    auto lane_slots = [&](int l) {
        LaneSlot z;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = 4 * (8 * j + wave) + (l >> 4), c = l & 15;
            z.srow[j] = r;
            z.skoff[j] = (c ^ (r & 15)) << 3;                 // K: position c holds chunk c ^ (row & 15)
            z.svoff[j] = ((c - 2 * (r & 7)) & 15) << 3;       // V: position c holds chunk (c - 2 (row & 7)) mod 16
        }
        return z;
    };
    auto fresh_slots = [&]() {
        int l = lane;
        if constexpr (GATHER) asm volatile("" : "+v"(l));
        return lane_slots(l);
    };
    [[maybe_unused]] const LaneSlot slots0 = lane_slots(lane);
    // stage s of the walk -> its region and first key (workgroup-uniform; NREG <= 4, unrolled selects)
    auto locate = [&](int s, Region& x, int& key0) {
        x = reg[0];
        key0 = reg[0].a;
        int base = 0;
#pragma unroll
        for (int r = 0; r < NREG; ++r) {
            if (s >= base && s < base + nst_r[r]) {
                x = reg[r];
                key0 = reg[r].a + (s - base) * (reg[r].lo ? kTK : 2 * kTK);
            }
            base += nst_r[r];
        }
    };
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    // GATHER: the row-table entries of this lane's four DMA slots of the NEXT stage to issue are themselves fetched by LDS-DMA
    // (4 x 1 KiB per wave and stage, right behind the previous stage's tile DMA; landed by the loop's next vmcnt(0)): no register
    // holds them across a stage's arithmetic and hipcc sees no load it would wait for in the wrong place.  128 KiB of ring +
    // 32 KiB of entries = all of the CU's LDS, one workgroup per CU as before.  g_nost: do-not-store lane masks (ballots, in SGPRs) of the
    // three stages in flight: [0] the stage being multiplied, [1] the one landing, [2] the one just issued.
    __shared__ __attribute__((aligned(16))) char gtab[GATHER ? kRingThreads * 64 : 16];
    [[maybe_unused]] const uint32_t gtab0 = GATHER ? (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)gtab + wave * 4096 : 0;
    [[maybe_unused]] uint64_t g_nost[3][4] = {{~0ull, ~0ull, ~0ull, ~0ull}, {~0ull, ~0ull, ~0ull, ~0ull}, {~0ull, ~0ull, ~0ull, ~0ull}};
    [[maybe_unused]] auto g_rotate = [&]() {
#pragma unroll
        for (int k = 0; k < 4; ++k) { g_nost[0][k] = g_nost[1][k]; g_nost[1][k] = g_nost[2][k]; }
    };
    // Who stores: every workgroup of a kv head's group streams the same staged keys (they lie in front of every query row), so
    // the landed stages are dealt out over all of them -- stage i leaves with writer i mod g_nw.  The stores are issued in front
    // of the stage's arithmetic, which hides their completion; a light last q-block (config 4's 259 rows = 128 + 128 + 3) has no
    // arithmetic to hide it behind but is one memory round trip per stage ahead of the others anyway.  (One writer per kv head:
    // +47 us per launch at config 4, all stages to the light block: +43 us, dealt out: see profiles/r04_variants.txt.)
    [[maybe_unused]] int g_w = -1, g_nw = 1;
    if constexpr (GATHER) {
        const int group = p.H / p.Hkv, nqb = p.nqblk;
        g_nw = group * nqb;
        g_w = (h % group) * nqb + qblk;
    }
    [[maybe_unused]] auto fetch_entries = [&](int stage) {
        if constexpr (GATHER) {
            if (stage >= nst) return;
            Region x;
            int skey0;
            locate(stage, x, skey0);
            if (x.lo) return;                                  // (workgroup-uniform) the pass's own rows are read from the arena as always
#ifdef PC_RING_G_NOFETCH
            return;
#endif
            const int last = x.e - 1;
            const LaneSlot z = fresh_slots();
            const int* srow = z.srow;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int key = skey0 + t * kTK + srow[j];
                    glds16_raw((const _Float16*)(p.rows + (key < last ? key : last)), gtab0 + (2 * t + j) * 1024);
                }
        }
    };
    auto issue = [&](int stage, uint32_t buf) {
        Region x;
        int skey0;
        locate(stage, x, skey0);
        const int last = x.e - 1;
        const LaneSlot z = GATHER ? fresh_slots() : slots0;
        const int *srow = z.srow, *skoff = z.skoff, *svoff = z.svoff;
        // (GATHER: the four entries of the stage come out of LDS in one batch -- one LDS round trip in front of the eight DMA
        // instructions, not one per slot: 0.4 us per stage otherwise, tools/ring_gather_micro.py)
        [[maybe_unused]] u32x4 ent[GATHER ? 4 : 1];
        if constexpr (GATHER) {
            if (!x.lo) {
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) ent[sl] = *(const u32x4*)(gtab + wave * 4096 + sl * 1024 + lane * 16);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint32_t dst = buf + (8 * j + wave) * 1024;
#ifndef PC_RING_G_PLAINADDR     // (dev probe: the staging instantiation with the plain launch's addressing -- all rows must be staged)
                if constexpr (GATHER) {
                    if (!x.lo) {
                        const u32x4 e = ent[2 * t + j];
                        const uint64_t base = ((uint64_t)e[1] << 32) | e[0];
                        const uint64_t ka = base + (((uint64_t)(uint32_t)(p.g_kplane + hkv) * e[2]) << 4) + (uint32_t)(skoff[j] * 2);
                        const uint64_t va = base + (((uint64_t)(uint32_t)(p.g_vplane + hkv) * e[2]) << 4) + (uint32_t)(svoff[j] * 2);
                        const int key = skey0 + t * kTK + srow[j];
                        const uint32_t nostore = (key <= last && !(e[3] & PC_KV_ROW_STAGED)) ? 0u : 1u;
                        g_nost[2][2 * t + j] = __ballot(nostore);
                        glds16_raw((const _Float16*)(uintptr_t)ka, dst + (2 * t) * kPlane);
                        glds16_raw((const _Float16*)(uintptr_t)va, dst + (2 * t + 1) * kPlane);
                        continue;
                    }
                    g_nost[2][2 * t + j] = ~0ull;              // (the pass's own rows: nothing to store)
                }
#endif
                if (x.lo) {
                    // one tile with residuals: planes K V Klo Vlo (t = 0: K, V; t = 1: Klo, Vlo)
                    int key = skey0 + srow[j];
                    key = key < last ? key : last;
                    const _Float16* kb = t ? x.kl : x.k;
                    const _Float16* vb = t ? x.vl : x.v;
                    glds16_raw(kb + (int64_t)key * D + skoff[j], dst + (2 * t) * kPlane);
                    glds16_raw(vb + (int64_t)key * D + svoff[j], dst + (2 * t + 1) * kPlane);
                } else {
                    int key = skey0 + t * kTK + srow[j];
                    key = key < last ? key : last;
                    glds16_raw(x.k + (int64_t)key * D + skoff[j], dst + (2 * t) * kPlane);
                    glds16_raw(x.v + (int64_t)key * D + svoff[j], dst + (2 * t + 1) * kPlane);
                }
            }
        }
    };

    // one 64-key tile out of LDS: scores, online softmax, O^T update (the arithmetic of attn_fwd_kernel's tile body).
    // The fragment reads run ONE BLOCK AHEAD of the MFMAs that consume them (two register sets, issue order pinned with
    // sched_group_barrier): left alone hipcc puts every read right in front of its first use and the wave sits out one LDS
    // round trip per 16-key block and per head-dim block (12 per tile).  attn_fwd_kernel has no registers for the second
    // set; this kernel, without staging registers, does.
    // MASKED = false: every key of the tile is visible to every row of the wave and inside the region -- no compares.
    auto tile = [&](const _Float16* Kl, const _Float16* Vl, const _Float16* Kll, const _Float16* Vll, auto lo_tag, auto mask_tag,
                    const int key0, const int key_end) {
        constexpr bool tile_lo = KVLO && decltype(lo_tag)::value;
        constexpr bool MASKED = decltype(mask_tag)::value;
        constexpr int NRK = tile_lo ? 2 * KS : KS;          // K-fragment reads per 16-key block
        constexpr int NMK = tile_lo ? 3 * KS : 2 * KS;      // MFMAs per 16-key block
        float sv[4][4];
        float mx = -INFINITY;
        h8 ka[2][KS];
        [[maybe_unused]] h8 kal[2][tile_lo ? KS : 1];
        auto load_k = [&](int kb, int set) {
            const int row = kb * 16 + n;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int off = row * D + (((ks * 4 + g) ^ (row & (CPR - 1))) << 3);
                ka[set][ks] = *(const h8*)(Kl + off);
                if constexpr (tile_lo) kal[set][ks] = *(const h8*)(Kll + off);
            }
        };
        load_k(0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NRK, 0);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            if (kb + 1 < 4) load_k(kb + 1, (kb + 1) & 1);
            f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka[kb & 1][ks], qf[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka[kb & 1][ks], qfl[ks], acc, 0, 0, 0);
                if constexpr (tile_lo) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kal[kb & 1][ks], qf[ks], acc, 0, 0, 0);
            }
            if (kb + 1 < 4) __builtin_amdgcn_sched_group_barrier(0x100, NRK, 0);      // the next block's reads first ...
            __builtin_amdgcn_sched_group_barrier(0x008, NMK, 0);                      // ... then this block's MFMAs
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // (raw scores: the softmax scale is folded into the exponent below -- one fma per score instead of mul + sub)
                float s_ = acc[r];
                if constexpr (MASKED) {
                    const int key = key0 + kb * 16 + g * 4 + r;
                    s_ = (key < row_vis_end && key < key_end) ? s_ : -INFINITY;
                }
                sv[kb][r] = s_;
                mx = fmaxf(mx, s_);
            }
        }
#if PC_RING_EXP == 1      // dev probe: no cross-lane max, no exponentials (timing attribution only; results are wrong)
        const float m_new = m_run;
#else
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
#endif
        const float c_ = p.scale_log2, mc = m_new * c_;       // m_run / m_new are in raw-score units, the partials' m in log2 units
        const float alpha = fast_exp2(m_run * c_ - mc);
        float rs = 0.f;
        h8 pb[2], pbl[2];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            float e[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#if PC_RING_EXP == 1
                e[r] = sv[kb][r];
#else
                e[r] = fast_exp2(__builtin_fmaf(sv[kb][r], c_, -mc));
#endif
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
#if PC_RING_EXP != 1
        rs += __shfl_xor(rs, 16);
        rs += __shfl_xor(rs, 32);
#endif
        l_run = l_run * alpha + rs;
        if (__any(m_new > m_run)) {
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                o[db][0] *= alpha; o[db][1] *= alpha; o[db][2] *= alpha; o[db][3] *= alpha;
            }
        }
        m_run = m_new;
        // O^T += V^T . P^T: per 16-wide head-dim block two 32-key steps; its V^T fragments are read one block ahead
        constexpr int NRV = tile_lo ? 8 : 4, NMV = tile_lo ? 6 : 4;
        h4 va[2][4];
        [[maybe_unused]] h4 val[2][tile_lo ? 4 : 1];
        auto load_v = [&](int db, int set) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int vrow = t * 32 + g * 4 + (n >> 2);
                const int off = vrow * D + ((db * 16 + (n & 3) * 4 + 16 * (vrow & 7)) & (D - 1));
                va[set][2 * t] = lds_tr_read(Vl + off);
                va[set][2 * t + 1] = lds_tr_read(Vl + off + 16 * D);
                if constexpr (tile_lo) {
                    val[set][2 * t] = lds_tr_read(Vll + off);
                    val[set][2 * t + 1] = lds_tr_read(Vll + off + 16 * D);
                }
            }
        };
        load_v(0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NRV, 1);
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            if (db + 1 < DB) load_v(db + 1, (db + 1) & 1);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const h4 lo = va[db & 1][2 * t], hi = va[db & 1][2 * t + 1];
                const h8 a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                o[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pb[t], o[db], 0, 0, 0);
                o[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pbl[t], o[db], 0, 0, 0);
                if constexpr (tile_lo) {
                    const h4 llo = val[db & 1][2 * t], lhi = val[db & 1][2 * t + 1];
                    const h8 al = {llo[0], llo[1], llo[2], llo[3], lhi[0], lhi[1], lhi[2], lhi[3]};
                    o[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, pb[t], o[db], 0, 0, 0);
                }
            }
            if (db + 1 < DB) __builtin_amdgcn_sched_group_barrier(0x100, NRV, 1);
            __builtin_amdgcn_sched_group_barrier(0x008, NMV, 1);
        }
    };
    // Two PLAIN, fully visible tiles (A, B) of one stage as a software pipeline: the MFMAs of one tile run beside the softmax
    // arithmetic of the other -- QK(A) | QK(B) + softmax(A) | PV(A) + softmax(B) | PV(B) -- instead of QK, softmax, PV twice in a row
    // (a wave issues in order: an MFMA occupies the matrix pipe for 16 cycles after its 4-cycle issue, and independent VALU
    // instructions of the same wave fill that shadow; back to back, QK -> softmax -> PV leaves the pipe idle through every softmax).
    auto pair = [&](const _Float16* KA, const _Float16* VA, const _Float16* KB, const _Float16* VB) {
        const float c_ = p.scale_log2;
        auto qk = [&](const _Float16* Kl, f4 (&acc)[4], int sync) {
            h8 ka[2][KS];
            auto load_k = [&](int kb, int set) {
                const int row = kb * 16 + n;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
#if PC_RING_EXP == 4      // dev probe: no K reads
                    ka[set][ks] = h8{(_Float16)row, (_Float16)ks, 0, 0, 0, 0, 0, 0};
#else
                    ka[set][ks] = *(const h8*)(Kl + row * D + (((ks * 4 + g) ^ (row & (CPR - 1))) << 3));
#endif
                }
            };
            load_k(0, 0);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                if (kb + 1 < 4) load_k(kb + 1, (kb + 1) & 1);
                f4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    a = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka[kb & 1][ks], qf[ks], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka[kb & 1][ks], qfl[ks], a, 0, 0, 0);
                }
                acc[kb] = a;
            }
            (void)sync;
        };
        // online softmax of one tile's raw scores: P as hi / lo operand pairs, the row sums and maxima, alpha for the rescale
        auto soft = [&](const f4 (&acc)[4], h8 (&pb)[2], h8 (&pbl)[2], float& alpha, bool& moved) {
            float mx = fmaxf(fmaxf(fmaxf(acc[0][0], acc[0][1]), fmaxf(acc[0][2], acc[0][3])), fmaxf(fmaxf(acc[1][0], acc[1][1]), fmaxf(acc[1][2], acc[1][3])));
            mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(acc[2][0], acc[2][1]), fmaxf(acc[2][2], acc[2][3])), fmaxf(fmaxf(acc[3][0], acc[3][1]), fmaxf(acc[3][2], acc[3][3]))));
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx), mc = m_new * c_;
            alpha = fast_exp2(m_run * c_ - mc);
            moved = m_new > m_run;
            float rs = 0.f;
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
            rs += __shfl_xor(rs, 16);
            rs += __shfl_xor(rs, 32);
            l_run = l_run * alpha + rs;
            m_run = m_new;
        };
        auto rescale = [&](float alpha, bool moved) {
            if (__any(moved)) {
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    o[db][0] *= alpha; o[db][1] *= alpha; o[db][2] *= alpha; o[db][3] *= alpha;
                }
            }
        };
        auto pv = [&](const _Float16* Vl, const h8 (&pb)[2], const h8 (&pbl)[2]) {
            h4 va[2][4];
            auto load_v = [&](int db, int set) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int vrow = t * 32 + g * 4 + (n >> 2);
                    const int off = vrow * D + ((db * 16 + (n & 3) * 4 + 16 * (vrow & 7)) & (D - 1));
#if PC_RING_EXP == 3      // dev probe: no V^T reads (LDS-bandwidth attribution; results are wrong)
                    va[set][2 * t] = h4{(_Float16)off, 0, 0, 0};
                    va[set][2 * t + 1] = h4{0, (_Float16)off, 0, 0};
#else
                    va[set][2 * t] = lds_tr_read(Vl + off);
                    va[set][2 * t + 1] = lds_tr_read(Vl + off + 16 * D);
#endif
                }
            };
            load_v(0, 0);
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                if (db + 1 < DB) load_v(db + 1, (db + 1) & 1);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const h4 lo = va[db & 1][2 * t], hi = va[db & 1][2 * t + 1];
                    const h8 a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    o[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pb[t], o[db], 0, 0, 0);
                    o[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pbl[t], o[db], 0, 0, 0);
                }
            }
        };
        f4 accA[4], accB[4];
        h8 pA[2], pAl[2], pB[2], pBl[2];
        float alA, alB;
        bool mvA, mvB;
        qk(KA, accA, 2);
        qk(KB, accB, 3);                 // } one scheduling region: B's MFMAs beside A's exponentials
        soft(accA, pA, pAl, alA, mvA);   // }
        rescale(alA, mvA);
        pv(VA, pA, pAl);                 // } and A's V^T . P^T beside B's exponentials
        soft(accB, pB, pBl, alB, mvB);   // }
        rescale(alB, mvB);
        pv(VB, pB, pBl);
    };

    using T_ = std::true_type;
    using F_ = std::false_type;
    // a tile whose every key is visible to every row of this wave (and lies inside its region) skips the mask compares
    auto run_tile = [&](const _Float16* Kl, const _Float16* Vl, const _Float16* Kll, const _Float16* Vll, auto lo_tag, const int key0,
                        const int key_end) {
        const int full_end = past_len + qrow0 + 1;           // keys visible to the FIRST row of the wave
        if (key0 + kTK <= key_end && key0 + kTK <= full_end && qrow0 + 16 <= q_len)
            tile(Kl, Vl, Kll, Vll, lo_tag, F_{}, key0, key_end);
        else
            tile(Kl, Vl, Kll, Vll, lo_tag, T_{}, key0, key_end);
    };

    // ---- the ring ----
#if PC_RING_PRIO
    // static priority for the second-dispatched half of the workgroup: at equal priority the older wave of a SIMD wins every
    // VALU arbitration and waves 4-7 start each phase late
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
    if (nst > 0) {
        if constexpr (GATHER) { fetch_entries(0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        issue(0, lds0);
        if constexpr (GATHER) { g_rotate(); g_rotate(); }                       // stage 0's masks -> [0]
        if (nst > 1) {
            if constexpr (GATHER) { fetch_entries(1); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            issue(1, lds0 + kStage);
            if constexpr (GATHER) {
#pragma unroll
                for (int k = 0; k < 4; ++k) g_nost[1][k] = g_nost[2][k];          // stage 1's masks -> [1]
                fetch_entries(2);
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // (the entry DMA sits behind the tile DMA in the queue: drain)
            } else {
                asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");  // stage 0 landed, everyone's
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        // (the Q fragments are this kernel's only loads hipcc knows about: consumed here, its vmcnt(0) for them lands in
        // front of the loop instead of in front of the first MFMA of every tile, where it would also drain the ring)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" ::"v"(qf[ks]), "v"(qfl[ks]));
        for (int i = 0; i < nst; ++i) {
            const char* buf = lds + (i & 1) * kStage;
            Region x;
            int key0;
            locate(i, x, key0);
            if constexpr (GATHER) {
                // ---- stage i has landed: the rows of it that are not in the arena yet leave for it ----
                if (!x.lo && i % g_nw == g_w) {
                    _Float16* kd = const_cast<_Float16*>(p.k) + (int64_t)hkv * p.kv_hs;
                    _Float16* vd = const_cast<_Float16*>(p.v) + (int64_t)hkv * p.kv_hs;
                    const LaneSlot z = fresh_slots();
                    const int *srow = z.srow, *skoff = z.skoff, *svoff = z.svoff;
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            if (!((g_nost[0][2 * t + j] >> lane) & 1ull)) {
                                const char* src = buf + (8 * j + wave) * 1024 + lane * 16;
                                const u32x4 kc = *(const u32x4*)(src + (2 * t) * kPlane);
                                const u32x4 vc = *(const u32x4*)(src + (2 * t + 1) * kPlane);
                                const int64_t key = key0 + t * kTK + srow[j];
                                __builtin_nontemporal_store(kc, (u32x4*)(kd + key * D + skoff[j]));
                                __builtin_nontemporal_store(vc, (u32x4*)(vd + key * D + svoff[j]));
                            }
                        }
                }
            }
            if (wave_active && key0 < wave_vis_end) {
                if (KVLO && x.lo) {
                    run_tile((const _Float16*)buf, (const _Float16*)(buf + kPlane), (const _Float16*)(buf + 2 * kPlane),
                             (const _Float16*)(buf + 3 * kPlane), T_{}, key0, x.e);
                } else {
                    const int full_end = (x.e < past_len + qrow0 + 1) ? x.e : past_len + qrow0 + 1;
                    if (PC_RING_PAIR && key0 + 2 * kTK <= full_end && qrow0 + 16 <= q_len) {
                        // both tiles whole and visible to every row of the wave: the pipelined pair
                        pair((const _Float16*)buf, (const _Float16*)(buf + kPlane), (const _Float16*)(buf + 2 * kPlane),
                             (const _Float16*)(buf + 3 * kPlane));
                    } else {
                        run_tile((const _Float16*)buf, (const _Float16*)(buf + kPlane), nullptr, nullptr, F_{}, key0, x.e);
                        if (key0 + kTK < x.e && key0 + kTK < wave_vis_end)
                            run_tile((const _Float16*)(buf + 2 * kPlane), (const _Float16*)(buf + 3 * kPlane), nullptr, nullptr, F_{},
                                     key0 + kTK, x.e);
                    }
                }
            }
            if (i + 1 < nst) {
                // stage i+1 has landed (this wave's DMA drained, everyone's via the barrier) and every wave is done reading
                // stage i: its buffer takes stage i+2, which then has the whole of stage i+1's arithmetic to arrive
#if PC_RING_EXP != 2      // (2: dev probe without the ring's synchronisation and refills)
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (i + 2 < nst) issue(i + 2, lds0 + (i & 1) * kStage);
                if constexpr (GATHER) { g_rotate(); fetch_entries(i + 3); }
#endif
            }
        }
    }

    // ---- epilogue (as attn_fwd_kernel) ----
    if (!(wave_active && qi < q_len)) return;
    if (p.nsplit == 1) {
        const float inv = 1.0f / l_run;
        if (p.of_hi) {
            const int row = b * q_len + qi, KSo = p.H * D / 32;
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                h4 hi, lo;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    _Float16 vh, vl;
                    pc_split(o[db][r] * inv, vh, vl);
                    hi[r] = vh; lo[r] = vl;
                }
                const int64_t off = frag_off(row, h * D + db * 16 + g * 4, KSo);
                *(h4*)(p.of_hi + off) = hi;
                *(h4*)(p.of_lo + off) = lo;
            }
            return;
        }
        const int64_t ooff = b * p.o_bs + (int64_t)qi * p.o_ts + (int64_t)h * D + g * 4;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            h4 r, rl;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                _Float16 vh, vl;
                pc_split(o[db][j] * inv, vh, vl);
                r[j] = vh; rl[j] = vl;
            }
            *(h4*)(p.out + ooff + db * 16) = r;
            if (p.out_lo) *(h4*)(p.out_lo + ooff + db * 16) = rl;
        }
    } else {
        const int64_t slot = (((int64_t)b * p.H + h) * p.nsplit + split) * q_len + qi;
        float* po = p.part_o + slot * D + g * 4;
#pragma unroll
        for (int db = 0; db < DB; ++db) *(f4*)(po + db * 16) = o[db];
        if (g == 0) { p.part_ml[slot * 2] = m_run * p.scale_log2; p.part_ml[slot * 2 + 1] = l_run; }
    }
}

}  // namespace

// Launches the ring kernel takes: head_dim 128, split-precision Q (q_lo), more than 32 query rows, no ALiBi, not the tail /
// small-q modes.  PC_ATTN_NO_RING=1: off (A/B against attn_fwd_kernel).
// fewest query rows the ring kernel takes (PC_ATTN_RING_MIN; pc_attn_workspace_bytes sizes the split-KV workspace with the same number)
int ring_min_rows() {
    static const int min_rows = [] { const char* m = getenv("PC_ATTN_RING_MIN"); return m ? atoi(m) : 33; }();
    return min_rows;
}
bool ring_eligible(const AttnParams& p, int D) {
    const char* e = getenv("PC_ATTN_NO_RING");            // (read per call: tests and probes switch it inside one process)
    const bool off = e && e[0] == '1';
    return !off && D == RD && p.q_lo && p.q_len >= ring_min_rows() && !p.key_pos && !p.tail && !p.small;
}

// KV splits of a ring launch: one workgroup per CU; minimise rounds x (stages per split + 1) + the merge
int ring_nsplit(int B, int H, int q_len, int kv_len) {
    const int units = B * H * pc_ceil_div(q_len, kRingQB);
    if (units >= 256) return 1;             // the grid fills the chip by itself: splits only add partials to merge
    // (rounds 3-4 kept >= 4 stages per split; for 33..128 rows over a 1.7 k cache that left half of the CUs idle: 7b q = 36 / 50 /
    // 66 / 102: 5.10 / 5.41 / 6.05 / 6.65 -> 4.91 / 5.27 / 5.85 / 6.48 ms without the floor, profiles/r04_variants.txt)
    static const int min_stages = [] { const char* e = getenv("PC_RING_MIN_STAGES"); return e ? atoi(e) : 1; }();
    // (Round 5 tried pricing a THIN last q-block -- 259 = 128 + 128 + 3 rows, BASELINE config 4: seven of its eight waves only help
    // with the DMA -- at a third of a full block so that 259 rows take the 3 splits 256 rows take: 360 workgroups instead of 240,
    // the attention launch 155 -> 202 us and the step 21.3 -> 22.8 ms on one box (profiles/r05_variants.txt): the second round of
    // workgroups costs more than the shorter key ranges save.  The count below stands.)
    int best = 1;
    double best_cost = 1e30;
    for (int s = 1; s <= 16; ++s) {
        const int stages = pc_ceil_div(pc_ceil_div(kv_len, s), 2 * kTK);
        if (s > 1 && stages < min_stages) break;
        const int rounds = pc_ceil_div(units * s, 256);
        const double cost = rounds * (stages + 1.0) + (s > 1 ? 2.0 + 0.25 * s : 0.0);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = s; }
    }
    return best;
}

int launch_attn_ring(const AttnParams& p0, int B, hipStream_t stream) {
    AttnParams p = p0;
    static const bool no_remap = [] { const char* e = getenv("PC_ATTN_NO_XCD"); return e && e[0] == '1'; }();
    p.nqblk = pc_ceil_div(p.q_len, kRingQB);
    p.nbatch = B;
    p.xcd_remap = (p.nqblk >= 2 && !no_remap) ? 1 : 0;
    const int gns = p.own_only ? 1 : p.nsplit;
    dim3 grid(p.nqblk, p.H, B * gns);
    if (p.xcd_remap) grid = dim3(8 * p.nqblk * ((p.H * B * gns + 7) / 8), 1, 1);
    const dim3 block(kRingThreads);
    if (p.pre_k) {
        if (p.k_lo) hipLaunchKernelGGL((attn_ring_kernel<true, true>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((attn_ring_kernel<false, true>), grid, block, 0, stream, p);
    } else if (p.rows) {         // stage while reading (pc_attn gather_rows)
        if (p.k_lo) hipLaunchKernelGGL((attn_ring_kernel<true, false, true>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((attn_ring_kernel<false, false, true>), grid, block, 0, stream, p);
    } else if (p.k_lo) hipLaunchKernelGGL((attn_ring_kernel<true, false>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((attn_ring_kernel<false, false>), grid, block, 0, stream, p);
    return pc_check_launch("attn_ring_kernel");
}

}  // namespace pca
