// Prefill attention over a staged KV arena (flash-style, MFMA, wave64) for gfx950.
//
// Replaces  LlamaAttention.forward core   promptcache/model/llama2.py:368-398
//             repeat_kv (:368-369), QK^T/sqrt(D) (:371), +mask (:384), softmax fp32 (:387), PV (:388),
//             transpose/reshape (:396-398)
//           _make_causal_mask / _prepare_decoder_attention_mask   llama2.py:62-76, :798-819  (implicit)
//
// Mask semantics (index order, NOT position order): new token i sees every staged key j < past_len and
// the new keys past_len + i' with i' <= i.  Position ids only steer RoPE (pc_rope.hip).
//
// Formulation ("doubly swapped" so all softmax state is lane-local, no cross-lane P shuffles):
//     S^T[key][q] = K[key][:] . Q[q][:]        mfma_f32_16x16x32_f16   A = K rows,   B = Q^T
//     O^T[d][q]  += V^T[d][key] . P^T[key][q]  mfma_f32_16x16x32_f16   A = V^T,      B = P^T
//   With the 16x16 C/D map (col = lane&15, row = 4*(lane>>4)+reg) a lane owns ONE query column n=lane&15:
//   its 4 accumulator registers are 4 keys of that query (S^T) or 4 head-dims of that query (O^T), so the
//   running max / sum / rescale never leave the lane group {n, n+16, n+32, n+48}.
//   The P^T B-operand wants key slot (g, j), g = lane>>4; a lane's own S^T values of two adjacent 16-key
//   blocks are keys {4g..4g+3} and {16+4g..16+4g+3}: we DEFINE slot (g, j<4) = key 4g+j, (g, j>=4) =
//   key 16+4g+(j-4) and feed V^T with the same permutation, so P goes register -> MFMA directly.
//   V^T fragments come from the row-major LDS V tile through ds_read_b64_tr_b16 (hardware transpose).
//
// Tiling: workgroup = 4 waves = 64 query rows (16 per wave) of one head; KV tile = 64 keys, loaded
// cooperatively (16 B/lane, full 256-B rows -> coalesced) into LDS (K XOR-swizzled against the 16-way
// ds_read_b128 bank conflict of a 256-B row stride).  When B*H*q-blocks cannot fill 256 CUs (cached
// prefill: q_len ~ 10..50) the KV axis is split across workgroups and a second kernel merges the
// (m, l, O) partials.
//
// Roofline: cached prefill (q << S) is HBM-bound on the K/V stream: bytes = 2*Hkv*(S+q)*D*2 per layer;
// encode / no-cache (q = S) is MFMA-bound: flops = 4*H*D*q*(S + (q+1)/2) per layer.
#include "pc_attn_common.h"

namespace {

using namespace pca;

// The past length of batch row b: per-row device word, the graph's device word, or the host value.  (The host value goes
// through an opaque register first: hipcc otherwise makes this ONE load through a select between the device pointers and the
// address of the kernel argument, and parks the argument in a scratch slot to have such an address.)
__device__ __forceinline__ int row_past_len(const AttnParams& p, int b) {
    int v = p.past_len;
    asm volatile("" : "+s"(v));
    if (p.past_lens) v = p.past_lens[b];
    else if (p.past_len_dev) v = *p.past_len_dev;
    return v;
}

#ifndef PC_TAIL_MAX
#define PC_TAIL_MAX 32
#endif
constexpr int kTailMax = PC_TAIL_MAX;   // most rows of one prefill pass the tail workgroup handles (NT = 16 or 32 below)

// The attention of <= NT new query rows over the <= NT rows their own pass appended (keys past_len + j, j <= qi), for one
// head, as a split-KV partial (m, l, O) in the log2 domain.  All fp32 FMAs on (q_hi + q_lo), (K + K_lo), (V + V_lo): the
// reference computes the pass in fp32 (llama2.py:361-388), only the STAGED rows are fp16 there.  It runs in the
// workgroup of an extra split while the other splits stream the staged keys, so its 2-5 us hide under their ~10 us.
// (Residual tiles inside the streaming kernel instead cost 32 VGPRs -- the second resident workgroup per CU, or spills --
// and a straggler split: 13.3 -> 15.9 .. 25 us per launch on the persona prompt; the same arithmetic inside the merge
// kernel lengthened the serial merge by 1.1 us per layer.)
// 256 threads = NT rows x LPR lanes; a lane owns JPT = NT / LPR keys in the score phase and D / LPR output dims after it.
// LDS: qs / ks / vs are [NT][D] fp32 (qs, ks with the float4 column XOR-swizzled by the row: the lanes of a row group
// read different rows of one column; unswizzled they share a bank), ps is [NT][NT].
template <int D, int NT, bool ALIBI, bool WT = false>   // WT: the partial leaves through write-through stores (fused merge)
__device__ __forceinline__ void attn_tail_block(const AttnParams& p, float* __restrict__ qs, float* __restrict__ ks,
                                                float* __restrict__ vs, float* __restrict__ ps, int b, int h, int split) {
    constexpr int CPR = D / 8;             // 16-byte fp16 chunks per row
    constexpr int F4R = D / 4;             // float4 columns per fp32 row
    constexpr int SWZ = F4R >= 16 ? 15 : F4R - 1;
    constexpr int LPR = kThreads / NT;     // lanes per query row (16 / 8)
    constexpr int JPT = NT / LPR;          // keys per lane (1 / 4)
    constexpr int DPT = D / LPR;           // output dims per lane in the O phase
    static_assert(NT * LPR == kThreads && JPT * LPR == NT && DPT * LPR == D && DPT % 2 == 0, "tail thread map");
    const int tid = threadIdx.x, q_len = p.q_len;
    const int past = row_past_len(p, 0);
    const int hkv = h / (p.H / p.Hkv);
    const _Float16* kb = p.k + b * p.kv_bs + (int64_t)hkv * p.kv_hs + (int64_t)past * D;
    const _Float16* vb = p.v + b * p.kv_bs + (int64_t)hkv * p.kv_hs + (int64_t)past * D;
    const _Float16* klb = p.k_lo + b * p.lo_bs + (int64_t)hkv * p.lo_hs;
    const _Float16* vlb = p.v_lo + b * p.lo_bs + (int64_t)hkv * p.lo_hs;
    for (int idx = tid; idx < NT * CPR; idx += kThreads) {
        const int row = idx / CPR, c = idx - row * CPR;
        const int rc = row < q_len ? row : q_len - 1;          // clamped: unconditional loads
        const int64_t qoff = b * p.q_bs + (int64_t)rc * p.q_ts + (int64_t)h * D + c * 8;
        const h8 qh = *(const h8*)(p.q + qoff);
        h8 ql = {0, 0, 0, 0, 0, 0, 0, 0};
        if (p.q_lo) ql = *(const h8*)(p.q_lo + qoff);
        const h8 kk = *(const h8*)(kb + (int64_t)rc * D + c * 8), kl = *(const h8*)(klb + (int64_t)rc * D + c * 8);
        const h8 vv = *(const h8*)(vb + (int64_t)rc * D + c * 8), vl = *(const h8*)(vlb + (int64_t)rc * D + c * 8);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f4 q4, k4, v4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                q4[e] = (float)qh[half * 4 + e] + (float)ql[half * 4 + e];
                k4[e] = (float)kk[half * 4 + e] + (float)kl[half * 4 + e];
                v4[e] = (float)vv[half * 4 + e] + (float)vl[half * 4 + e];
            }
            const int col = ((2 * c + half) ^ (row & SWZ)) * 4;
            *(f4*)(qs + row * D + col) = q4;
            *(f4*)(ks + row * D + col) = k4;
            *(f4*)(vs + row * D + (2 * c + half) * 4) = v4;      // read row-wise below: no swizzle
        }
    }
    __syncthreads();
    const int qi = tid / LPR, jl = tid - qi * LPR;               // (query row, lane in the row group)
    float acc[JPT];
#pragma unroll
    for (int u = 0; u < JPT; ++u) acc[u] = 0.f;
#pragma unroll 8
    for (int d4 = 0; d4 < F4R; ++d4) {
        const f4 a = *(const f4*)(qs + qi * D + ((d4 ^ (qi & SWZ)) * 4));
#pragma unroll
        for (int u = 0; u < JPT; ++u) {
            const int j = jl + u * LPR;
            const f4 k4 = *(const f4*)(ks + j * D + ((d4 ^ (j & SWZ)) * 4));
            acc[u] += a[0] * k4[0] + a[1] * k4[1] + a[2] * k4[2] + a[3] * k4[3];
        }
    }
    float sv[JPT];
    float m = kNegBig;
#pragma unroll
    for (int u = 0; u < JPT; ++u) {
        const int j = jl + u * LPR;
        sv[u] = acc[u] * p.scale_log2;
        if (ALIBI) sv[u] += p.slopes[h] * p.key_pos[b * p.kp_bs + past + (j < q_len ? j : q_len - 1)];
        const bool vis = qi < q_len && j <= qi;                  // causal inside the pass
        sv[u] = vis ? sv[u] : kNegBig;
        m = fmaxf(m, sv[u]);
    }
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    float l = 0.f;
#pragma unroll
    for (int u = 0; u < JPT; ++u) {
        const int j = jl + u * LPR;
        const float pj = (qi < q_len && j <= qi) ? exp2f(sv[u] - m) : 0.f;
        l += pj;
        ps[qi * NT + j] = pj;
    }
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1) l += __shfl_xor(l, off);
    const int64_t slot = (((int64_t)b * p.H + h) * p.nsplit + split) * q_len + qi;
    if (jl == 0 && qi < q_len) {
        if (WT) st_wt2(p.part_ml + slot * 2, m, l);
        else { p.part_ml[slot * 2] = m; p.part_ml[slot * 2 + 1] = l; }
    }
    __syncthreads();
    float o[DPT];
#pragma unroll
    for (int e = 0; e < DPT; ++e) o[e] = 0.f;
#pragma unroll 4
    for (int jj = 0; jj < NT; ++jj) {
        const float w = ps[qi * NT + jj];
#pragma unroll
        for (int e = 0; e < DPT; ++e) o[e] += w * vs[jj * D + jl * DPT + e];
    }
    if (qi < q_len) {
        if (WT) {
#pragma unroll
            for (int e = 0; e < DPT; e += 2) st_wt2(p.part_o + slot * D + jl * DPT + e, o[e], o[e + 1]);
        } else {
#pragma unroll
            for (int e = 0; e < DPT; ++e) p.part_o[slot * D + jl * DPT + e] = o[e];
        }
    }
}

// (Tried and dropped for q_len <= 16: the four waves of a workgroup taking different key tiles -- four tiles in
// flight per workgroup, one HBM round trip for a 4-tile range, partials merged through LDS.  In the captured forward
// it ran 13.6 us + 5.0 us combine (7 splits) against 12.4 us + 6.2 us (10 splits) for this kernel: the launch is
// bound by streaming 28.5 MB of K/V (~6 us) plus one latency and the serial tail, not by the number of round trips.)
// (Tried and dropped: 32 query rows per wave -- two 16-row sub-tiles sharing every K / V^T fragment read.  It halves
// LDS reads per flop but needs 256 VGPRs (1 wave per SIMD) and halves the workgroup count: 322 us vs 323 us at
// q = S = 4.4k, 30 us vs 16 us at q = 456.)
// HP ("high precision", used when q_len <= 64 where the kernel is HBM-bound and MFMA time is free): Q and P
// enter the MFMAs as split-precision pairs (hi = fp16(x), lo = fp16(x - hi)), i.e. two MFMAs per fragment.
// Against the reference's fp32 CPU path this removes the two largest rounding terms of the kernel (fp16 Q:
// 2.5e-3, fp16 P: 1.7e-3 max |delta logit| on a 7b-shaped layer); K/V stay fp16 as staged.
// PRE: the launch carries a shared key prefix (AttnParams::pre_k): its own instantiation, so that the others keep their
// register allocation (the prefix walk costs the residual variant its second wave per SIMD).
// GATHER (pc_attn gather_rows; tail mode, one q-block, B = 1: prompts of 17..32 new tokens): stage while reading, as in
// attn_small_kernel -- every staged key row is fetched from where its row-table entry says it lies and, unless it is in the arena
// already, written there when the tile goes to LDS (the staging registers hold it anyway).  A (kv head, key row) belongs to exactly
// one workgroup of the launch.  The tile's stores are waited for one tile later, behind a tile's worth of MFMAs.
#ifdef PC_GATHER_STORE_PLAIN
#define PC_GATHER_ST(v, p) (*(p) = (v))
#else
#define PC_GATHER_ST(v, p) __builtin_nontemporal_store((v), (p))
#endif
template <int D, bool HP, bool ALIBI, bool KVLO, bool PRE, bool GATHER = false>
__device__ __forceinline__ void attn_fwd_body(const AttnParams p) {
    static_assert(!KVLO || HP, "K/V residual planes go with split-precision Q and P");
    static_assert(!PRE || !ALIBI, "a shared prefix excludes ALiBi");
    static_assert(!GATHER || (HP && !KVLO && !PRE && !ALIBI), "the staging stream is the tail-mode instantiation");
    constexpr int KS = D / 32;   // MFMA k-steps across the head dim (QK^T)
    constexpr int DB = D / 16;   // 16-wide head-dim blocks of O^T
    constexpr int CPR = D / 8;   // 16-byte chunks per K/V row
    constexpr int LPT = kTK * CPR / kThreads;  // 16-byte loads per thread per tile per tensor
    static_assert(LPT >= 1, "tile too small for 256 threads");

    // (the split-precision variants without residual tiles also host attn_tail_block's fp32 buffers in these two)
    constexpr int kKlElems = (HP && !KVLO && 4 * kTailMax * D > kTK * D) ? 4 * kTailMax * D : kTK * D;
    constexpr int kVlElems = (HP && !KVLO && 2 * (kTailMax * D + kTailMax * kTailMax) > kTK * D)
                                 ? 2 * (kTailMax * D + kTailMax * kTailMax) : kTK * D;
    __shared__ __attribute__((aligned(16))) _Float16 Kl[kKlElems];
    __shared__ __attribute__((aligned(16))) _Float16 Vl[kVlElems];
    __shared__ __attribute__((aligned(16))) _Float16 Kll[KVLO ? kTK * D : 8];     // residual tiles (KVLO)
    __shared__ __attribute__((aligned(16))) _Float16 Vll[KVLO ? kTK * D : 8];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, g = lane >> 4;
    int qblk, h, b, split;
    if (p.xcd_remap) {
        // 1-D grid, XCD-aware (q >> 64, no KV split): the dispatcher places block i on XCD i % 8, and every
        // q-block of a head streams the same K/V.  Heads are dealt to XCDs (h % 8) and an XCD walks the q-blocks
        // of one head after another, heaviest (last) q-block first, so a head's K/V (2.25 MB at S = 4.4k) is
        // served from that XCD's 4 MiB L2 instead of crossing the fabric once per q-block.
        const int nqb = p.nqblk, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int per_xcd = (p.H * p.nbatch + 7) >> 3;          // (batch, head) pairs per XCD
        const int pair = (slot / nqb) * 8 + xcd;                // pair index = b * H + h
        if (slot / nqb >= per_xcd || pair >= p.H * p.nbatch) return;
        qblk = nqb - 1 - (slot % nqb);
        b = pair / p.H; h = pair - b * p.H; split = 0;
    } else {
        qblk = blockIdx.x; h = blockIdx.y;
        b = blockIdx.z / p.nsplit; split = blockIdx.z - b * p.nsplit;
    }
    const int hkv = h / (p.H / p.Hkv);
    const int q_len = p.q_len;
    // (the residual-tile variant keeps the select form: it sits at exactly 256 registers and the branchy form costs it its
    // second wave per SIMD; the select's scratch slot only shows up in the variants without residual tiles)
    int past_len_sel;
    if constexpr (KVLO && !PRE) past_len_sel = p.past_lens ? p.past_lens[b] : (p.past_len_dev ? *p.past_len_dev : p.past_len);
    else past_len_sel = row_past_len(p, b);
    const int past_len = past_len_sel;
    int nsp = p.nsplit;
    if constexpr (HP && !KVLO) {
        if (p.tail) {
            nsp = p.nsplit - 1;
            if (split == nsp) {           // workgroup-uniform: the extra split takes the pass's own rows
                static_assert(sizeof(Kl) >= 2 * kTailMax * D * sizeof(float) &&
                              sizeof(Vl) >= (kTailMax * D + kTailMax * kTailMax) * sizeof(float), "tail buffers live in the K / V tiles");
                if (q_len <= 16 || kTailMax == 16)
                    attn_tail_block<D, 16, ALIBI>(p, (float*)Kl, (float*)Kl + 16 * D, (float*)Vl, (float*)Vl + 16 * D, b, h, split);
                else
                    attn_tail_block<D, kTailMax, ALIBI>(p, (float*)Kl, (float*)Kl + kTailMax * D, (float*)Vl, (float*)Vl + kTailMax * D, b, h, split);
                return;
            }
        }
    }
    const int kv_len = p.tail ? past_len : past_len + q_len;   // tail mode: staged keys only, all visible to every row

    // this split's key range (tile-aligned) clipped by what the workgroup can causally see
    int kps = (kv_len + nsp - 1) / nsp;
    kps = (kps + kTK - 1) / kTK * kTK;
    const int ks0 = split * kps;
    const int wg_rows_end = (qblk * kQB + kQB < q_len) ? qblk * kQB + kQB : q_len;
    int kend = ks0 + kps;
    kend = kend < kv_len ? kend : kv_len;
    kend = kend < past_len + wg_rows_end ? kend : past_len + wg_rows_end;

    const int qrow0 = qblk * kQB + wave * 16;
    const int qi = qrow0 + n;                       // this lane's query row (new-token index)
    const bool wave_active = qrow0 < q_len;         // wave-uniform
    const int wave_rows_end = (qrow0 + 16 < q_len) ? qrow0 + 16 : q_len;
    const int wave_vis_end = p.tail ? past_len : past_len + wave_rows_end;
    const int row_vis_end = (qi < q_len) ? (p.tail ? past_len : past_len + qi + 1) : 0;  // keys [0, row_vis_end) are visible

    h8 qf[KS], qfl[HP ? KS : 1];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        qf[ks] = z;
        if (HP) qfl[ks] = z;
        if (wave_active && qi < q_len) {
            const int64_t off = b * p.q_bs + (int64_t)qi * p.q_ts + (int64_t)h * D + ks * 32 + g * 8;
            qf[ks] = *(const h8*)(p.q + off);
            if (HP && p.q_lo) qfl[ks] = *(const h8*)(p.q_lo + off);
        }
    }

    f4 o[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db) { f4 z = {0.f, 0.f, 0.f, 0.f}; o[db] = z; }
    float m_run = kNegBig, l_run = 0.f;

    // a shared prefix shifts the pass's own planes: key index `pre + r` is their row r
    const int pre = PRE ? past_len : 0;
    const _Float16* kbase = p.k + b * p.kv_bs + (int64_t)hkv * p.kv_hs;
    const _Float16* vbase = p.v + b * p.kv_bs + (int64_t)hkv * p.kv_hs;
    [[maybe_unused]] const float slope = ALIBI ? p.slopes[h] : 0.f;
    [[maybe_unused]] const float* kpos = ALIBI ? p.key_pos + b * p.kp_bs : nullptr;

    // Register-staged, software-pipelined tiles: the global loads of tile i+1 are issued right after tile i
    // has been written to LDS and stay in flight while tile i is consumed (HBM latency hides under the MFMAs).
    u32x4 kr[LPT], vr[LPT];
    [[maybe_unused]] u32x4 krl[KVLO ? LPT : 1], vrl[KVLO ? LPT : 1];
    [[maybe_unused]] const _Float16* klb = KVLO ? p.k_lo + b * p.lo_bs + (int64_t)hkv * p.lo_hs : nullptr;   // (per segment)
    [[maybe_unused]] const _Float16* vlb = KVLO ? p.v_lo + b * p.lo_bs + (int64_t)hkv * p.lo_hs : nullptr;
    // lo_row0: -1 = the rows of this pass (past_len); -2 = the device word next to past_len (decode over a residual tail
    // that started at an earlier pass; hipGraph replays read both from the same static buffer)
    [[maybe_unused]] int lo_row0 = !KVLO ? 0 : (PRE ? pre : (p.lo_row0 == -2 ? p.past_len_dev[1] : (p.lo_row0 < 0 ? past_len : p.lo_row0)));
    // The key range is walked in up to two SEGMENTS with workgroup-uniform base pointers: the shared prefix [0, pre) out of
    // its own planes (when there is one), then the rows behind it out of k / v.  `kend` is the end of the current segment.
    const int kend_all = kend;
    int seg_key0 = ks0;
    if (PRE) {
        kend = kend_all < pre ? kend_all : pre;
        kbase = p.pre_k + (int64_t)hkv * p.pre_hs; vbase = p.pre_v + (int64_t)hkv * p.pre_hs;
        if (KVLO) {
            lo_row0 = p.pre_k_lo ? 0 : 0x7fffffff;
            if (p.pre_k_lo) { klb = p.pre_k_lo + (int64_t)hkv * p.pre_hs; vlb = p.pre_v_lo + (int64_t)hkv * p.pre_hs; }
        }
    }
    [[maybe_unused]] uint32_t gflag[GATHER ? LPT : 1];
    [[maybe_unused]] const bool g_writer = GATHER && (h % (p.H / p.Hkv)) == 0;
    [[maybe_unused]] const uint64_t g_koff = GATHER ? (uint64_t)(uint32_t)(p.g_kplane + hkv) : 0;
    [[maybe_unused]] const uint64_t g_voff = GATHER ? (uint64_t)(uint32_t)(p.g_vplane + hkv) : 0;
    auto issue_loads = [&](int key0) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int c = tid + i * kThreads;
            const int row = c / CPR, col = c - row * CPR;
            const int key = key0 + row;
            u32x4 z = {0u, 0u, 0u, 0u};
            kr[i] = z; vr[i] = z;   // zero-fill rows at/after kend: a garbage V row would turn 0 * NaN into NaN
            if (KVLO) { krl[i] = z; vrl[i] = z; }
            if constexpr (GATHER) {
                gflag[i] = PC_KV_ROW_STAGED;
                if (key < kend) {
                    const u32x4 e = *(const u32x4*)(p.rows + key);
                    const uint64_t base = ((uint64_t)e[1] << 32) | e[0];
                    gflag[i] = e[3];
                    kr[i] = __builtin_nontemporal_load((const u32x4*)(uintptr_t)(base + ((g_koff * e[2]) << 4) + col * 16));
                    vr[i] = __builtin_nontemporal_load((const u32x4*)(uintptr_t)(base + ((g_voff * e[2]) << 4) + col * 16));
                }
                continue;
            }
            if (key < kend) {
                kr[i] = *(const u32x4*)(kbase + (int64_t)key * D + col * 8);
                vr[i] = *(const u32x4*)(vbase + (int64_t)key * D + col * 8);
                if (KVLO && key >= lo_row0) {            // rows before lo_row0 are exact fp16 (staged module KV)
                    krl[i] = *(const u32x4*)(klb + (int64_t)(key - lo_row0) * D + col * 8);
                    vrl[i] = *(const u32x4*)(vlb + (int64_t)(key - lo_row0) * D + col * 8);
                }
            }
        }
    };
  for (int seg = PRE ? 0 : 1; seg < 2; ++seg) {
    if (PRE && seg == 1) {           // (workgroup-uniform) the pass's own rows behind the prefix
        kend = kend_all;
        seg_key0 = ks0 > pre ? ks0 : pre;
        kbase = p.k + b * p.kv_bs + (int64_t)hkv * p.kv_hs - (int64_t)pre * D;     // key index pre + r is row r
        vbase = p.v + b * p.kv_bs + (int64_t)hkv * p.kv_hs - (int64_t)pre * D;
        if (KVLO) {
            lo_row0 = pre;
            klb = p.k_lo + b * p.lo_bs + (int64_t)hkv * p.lo_hs; vlb = p.v_lo + b * p.lo_bs + (int64_t)hkv * p.lo_hs;
        }
    }
    if (seg_key0 < kend) issue_loads(seg_key0);

    for (int key0 = seg_key0; key0 < kend; key0 += kTK) {
        // (workgroup-uniform) does this tile hold any key with a residual?  Most tiles of a long staged cache do not: their
        // residual tiles are neither written nor read (a question of 259 rows over 8.3 k staged keys: 129 of 134 tiles)
        [[maybe_unused]] const bool tile_lo = KVLO && key0 + kTK > lo_row0;
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int c = tid + i * kThreads;
            const int row = c / CPR, col = c - row * CPR;
            *(u32x4*)(Kl + row * D + ((col ^ (row & (CPR - 1))) << 3)) = kr[i];
            // V rows are rotated by 32 B per row (mod the row): the 8 rows a 32-lane half of ds_read_b64_tr_b16
            // touches then sit in 8 different 32-B windows of the 256-B bank row instead of the same one (the
            // unrotated tile measured SQ_LDS_BANK_CONFLICT = 68 % of SQ_LDS_IDX_ACTIVE: an 8-way conflict)
            *(u32x4*)(Vl + row * D + (((col + 2 * (row & 7)) & (CPR - 1)) << 3)) = vr[i];
            if constexpr (GATHER) {
                if (g_writer && !(gflag[i] & PC_KV_ROW_STAGED)) {      // (gflag is STAGED for rows at / behind kend)
                    _Float16* kd = const_cast<_Float16*>(kbase) + (int64_t)(key0 + row) * D + col * 8;
                    _Float16* vd = const_cast<_Float16*>(vbase) + (int64_t)(key0 + row) * D + col * 8;
                    PC_GATHER_ST(kr[i], (u32x4*)kd);
                    PC_GATHER_ST(vr[i], (u32x4*)vd);
                }
            }
            if (KVLO) {
                if (tile_lo) {
                    *(u32x4*)(Kll + row * D + ((col ^ (row & (CPR - 1))) << 3)) = krl[i];
                    *(u32x4*)(Vll + row * D + (((col + 2 * (row & 7)) & (CPR - 1)) << 3)) = vrl[i];
                }
            }
        }
        __syncthreads();
        if (key0 + kTK < kend) issue_loads(key0 + kTK);

        if (wave_active && key0 < wave_vis_end) {
            // ---- S^T = K . Q^T : four 16-key blocks ----
            float sv[4][4];
            float mx = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                f4 acc = {0.f, 0.f, 0.f, 0.f};
                const int row = kb * 16 + n;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int chunk = ks * 4 + g;
                    const h8 a = *(const h8*)(Kl + row * D + ((chunk ^ (row & (CPR - 1))) << 3));
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, qf[ks], acc, 0, 0, 0);
                    if (HP) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, qfl[ks], acc, 0, 0, 0);
                    if (KVLO) {
                        if (tile_lo) {
                            const h8 al = *(const h8*)(Kll + row * D + ((chunk ^ (row & (CPR - 1))) << 3));
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, qf[ks], acc, 0, 0, 0);
                        }
                    }
                }
                f4 kb4 = {0.f, 0.f, 0.f, 0.f};
                if (ALIBI) {        // positions of this lane's 4 keys (rows past kend are masked below; the buffer
                                    // is padded to a multiple of the tile, see pc_attn_fwd_alibi)
                    kb4 = *(const f4*)(kpos + key0 + kb * 16 + g * 4);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = key0 + kb * 16 + g * 4 + r;
                    float sc = acc[r] * p.scale_log2;
                    if (ALIBI) sc += slope * kb4[r];
                    const float s = (key < row_vis_end && key < kend) ? sc : -INFINITY;
                    sv[kb][r] = s;
                    mx = fmaxf(mx, s);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);       // stays finite (m_run starts at -1e30)
            const float alpha = fast_exp2(m_run - m_new);
            float rs = 0.f;
            h8 pb[2], pbl[HP ? 2 : 1];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = fast_exp2(sv[kb][r] - m_new);   // exp2(-inf) = 0 for masked keys
                    rs += e;
                    const _Float16 eh = (_Float16)e;
                    pb[kb >> 1][(kb & 1) * 4 + r] = eh;
                    if (HP) pbl[kb >> 1][(kb & 1) * 4 + r] = (_Float16)(e - (float)eh);
                }
            }
            rs += __shfl_xor(rs, 16);
            rs += __shfl_xor(rs, 32);
            l_run = l_run * alpha + rs;
            // Rescale O only when some row's running max actually moved (alpha == 1 otherwise: identical result).
            // The accumulators live in AGPRs; an unconditional multiply costs an accvgpr read + write per register
            // per tile (176 moves in the ISA), and after the first few tiles the max rarely changes.
            if (__any(m_new > m_run)) {
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    o[db][0] *= alpha; o[db][1] *= alpha; o[db][2] *= alpha; o[db][3] *= alpha;
                }
            }
            m_run = m_new;
            // ---- O^T += V^T . P^T : two 32-key steps x DB head-dim blocks ----
#pragma unroll
            for (int db = 0; db < DB; ++db) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int vrow = t * 32 + g * 4 + (n >> 2);          // (vrow + 16) & 7 == vrow & 7: same rotation
                    const _Float16* vp = Vl + vrow * D + ((db * 16 + (n & 3) * 4 + 16 * (vrow & 7)) & (D - 1));
                    const h4 lo = lds_tr_read(vp);            // keys 32t + 4g + {0..3}
                    const h4 hi = lds_tr_read(vp + 16 * D);   // keys 32t + 16 + 4g + {0..3}
                    const h8 a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    o[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pb[t], o[db], 0, 0, 0);
                    if (HP) o[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pbl[t], o[db], 0, 0, 0);
                    if (KVLO) {
                        if (tile_lo) {
                            const _Float16* vpl = Vll + (vp - Vl);
                            const h4 llo = lds_tr_read(vpl);
                            const h4 lhi = lds_tr_read(vpl + 16 * D);
                            const h8 al = {llo[0], llo[1], llo[2], llo[3], lhi[0], lhi[1], lhi[2], lhi[3]};
                            o[db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, pb[t], o[db], 0, 0, 0);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }

  }   // segments

    if (!(wave_active && qi < q_len)) return;
    if (p.nsplit == 1) {
        const float inv = 1.0f / l_run;
        if (p.of_hi) {
            // o_proj consumes split-precision fragment planes: row = token index over B*q_len, k = h*D + d
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
        if (g == 0) { p.part_ml[slot * 2] = m_run; p.part_ml[slot * 2 + 1] = l_run; }
    }
}

template <int D, bool HP, bool ALIBI = false, bool KVLO = false, bool GATHER = false>
__global__ __launch_bounds__(kThreads) void attn_fwd_kernel(const AttnParams p) {
    attn_fwd_body<D, HP, ALIBI, KVLO, false, GATHER>(p);
}

// the shared-prefix walk for passes of <= 64 rows (longer ones: pc_attn_ring.hip); its residual variant runs one wave per SIMD
// (292 registers) rather than spill
template <int D, bool HP, bool KVLO>
__global__ __launch_bounds__(kThreads) void attn_fwd_pre_kernel(const AttnParams p) {
    attn_fwd_body<D, HP, false, KVLO, true>(p);
}

// ---------------------------------------------------------------------------------------------------
// Large q (q_len > 64: schema encode, no-cache prefill, long questions): the MFMA-bound regime.
// The 16-row-per-wave kernel above re-reads every K / V^T fragment from LDS once per 16 query rows: 32 KiB of LDS reads
// per 32 MFMAs of 16 cycles, i.e. the LDS (128 B/clk/CU) is asked for twice what the MFMAs can consume.  This variant
// gives a wave 32 query rows and uses mfma_f32_32x32x16_f16: the same operand bytes per MFMA feed twice the flops.
//   workgroup = 4 waves = 128 query rows of one head; KV tile = 64 keys (same LDS tiles, swizzles and loaders);
//   S^T[key][q] (32 x 32 blocks, A = K rows, B = Q^T), O^T[d][q] += V^T[d][key] . P^T[key][q] (A = V^T via
//   ds_read_b64_tr_b16, B = P^T straight from the S^T accumulators).
// 32x32 C/D map: lane l owns query column l & 31; register r holds row 8*(r/4) + 4*(l>>5) + (r%4).  A/B operand:
// lane l holds row/column l & 31, k = 8*(l>>5) + j (j < 8).  As in the 16x16 kernel the key slots of the P^T operand
// are DEFINED by where the S^T accumulators already sit: for the 16-key step (blk, hh) slot (l>>5, j) is key
// 32 blk + 16 hh + (j < 4 ? 4 (l>>5) + j : 8 + 4 (l>>5) + j - 4) = accumulator registers 8 hh .. 8 hh + 7, and V^T is
// read with the same permutation.
typedef float f16v __attribute__((ext_vector_type(16)));
constexpr int kQB32 = 128;

template <int D, bool ALIBI = false>
__global__ __launch_bounds__(kThreads) void attn_fwd32_kernel(const AttnParams p) {
    constexpr int KS = D / 16;   // 16-wide k-steps across the head dim (QK^T)
    constexpr int DB = D / 32;   // 32-wide head-dim blocks of O^T
    constexpr int CPR = D / 8;
    constexpr int LPT = kTK * CPR / kThreads;
    // (two tile buffers with one barrier per tile measured 3 % slower than this single-buffered form)
    __shared__ __attribute__((aligned(16))) _Float16 Kl[kTK * D];
    __shared__ __attribute__((aligned(16))) _Float16 Vl[kTK * D];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 31, kg = lane >> 5;
    int qblk, h, b, split;
    if (p.xcd_remap) {
        const int nqb = p.nqblk, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int per_xcd = (p.H * p.nbatch + 7) >> 3;
        const int pair = (slot / nqb) * 8 + xcd;
        if (slot / nqb >= per_xcd || pair >= p.H * p.nbatch) return;
        qblk = nqb - 1 - (slot % nqb);
        b = pair / p.H; h = pair - b * p.H; split = 0;
    } else {
        qblk = blockIdx.x; h = blockIdx.y;
        b = blockIdx.z / p.nsplit; split = blockIdx.z - b * p.nsplit;
    }
    const int hkv = h / (p.H / p.Hkv);
    const int q_len = p.q_len;
    const int past_len = row_past_len(p, b);
    const int kv_len = past_len + q_len;
    int kps = (kv_len + p.nsplit - 1) / p.nsplit;
    kps = (kps + kTK - 1) / kTK * kTK;
    const int ks0 = split * kps;
    const int wg_rows_end = (qblk * kQB32 + kQB32 < q_len) ? qblk * kQB32 + kQB32 : q_len;
    int kend = ks0 + kps;
    kend = kend < kv_len ? kend : kv_len;
    kend = kend < past_len + wg_rows_end ? kend : past_len + wg_rows_end;

    const int qrow0 = qblk * kQB32 + wave * 32;
    const int qi = qrow0 + n;
    const bool wave_active = qrow0 < q_len;
    const int wave_rows_end = (qrow0 + 32 < q_len) ? qrow0 + 32 : q_len;
    const int wave_vis_end = past_len + wave_rows_end;
    const int row_vis_end = (qi < q_len) ? past_len + qi + 1 : 0;

    h8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        qf[ks] = z;
        if (wave_active && qi < q_len)
            qf[ks] = *(const h8*)(p.q + b * p.q_bs + (int64_t)qi * p.q_ts + (int64_t)h * D + ks * 16 + kg * 8);
    }
    f16v o[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
    float m_run = kNegBig, l_run = 0.f;
    [[maybe_unused]] const float slope = ALIBI ? p.slopes[h] : 0.f;
    [[maybe_unused]] const float* kpos = ALIBI ? p.key_pos + b * p.kp_bs : nullptr;

    const _Float16* kbase = p.k + b * p.kv_bs + (int64_t)hkv * p.kv_hs;
    const _Float16* vbase = p.v + b * p.kv_bs + (int64_t)hkv * p.kv_hs;
    u32x4 kr[LPT], vr[LPT];
    auto issue_loads = [&](int key0) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int c = tid + i * kThreads;
            const int row = c / CPR, col = c - row * CPR;
            const int key = key0 + row;
            u32x4 z = {0u, 0u, 0u, 0u};
            kr[i] = z; vr[i] = z;
            if (key < kend) {
                kr[i] = *(const u32x4*)(kbase + (int64_t)key * D + col * 8);
                vr[i] = *(const u32x4*)(vbase + (int64_t)key * D + col * 8);
            }
        }
    };
    if (ks0 < kend) issue_loads(ks0);

    for (int key0 = ks0; key0 < kend; key0 += kTK) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int c = tid + i * kThreads;
            const int row = c / CPR, col = c - row * CPR;
            *(u32x4*)(Kl + row * D + ((col ^ (row & (CPR - 1))) << 3)) = kr[i];
            // V rows rotated by 64 B per row (mod 4 rows): ds_read_b64_tr_b16 serves lanes 0-31 in one LDS cycle and
            // they touch 4 consecutive rows x 64 contiguous bytes here, which must fall into 4 disjoint 64-B windows
            // of the 256-B bank row (the 32-B rotation of the 16-row kernel left them overlapping pairwise:
            // SQ_LDS_BANK_CONFLICT = 24 % of SQ_LDS_IDX_ACTIVE)
            *(u32x4*)(Vl + row * D + (((col + 4 * (row & 3)) & (CPR - 1)) << 3)) = vr[i];
        }
        __syncthreads();
        if (key0 + kTK < kend) issue_loads(key0 + kTK);

        if (wave_active && key0 < wave_vis_end) {
            // ---- S^T = K . Q^T : two 32-key blocks ----
            // tiles every row of this wave sees in full (all but the diagonal and the last tile) skip the mask tests
            const bool full_tile = (key0 + kTK <= kend) && (key0 + kTK <= past_len + qrow0 + 1);
            f16v sacc[2];
            float mx = -INFINITY;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {                  // (alternating the two accumulators per k-step: 6 % slower)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[blk][r] = 0.f;
                const int row = blk * 32 + n;                    // A operand: key row of this lane
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int chunk = ks * 2 + kg;               // 16-byte chunk: d = 16 ks + 8 kg .. + 8
                    const h8 a = *(const h8*)(Kl + row * D + ((chunk ^ (row & (CPR - 1))) << 3));
                    sacc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qf[ks], sacc[blk], 0, 0, 0);
                }
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    f4 kb4 = {0.f, 0.f, 0.f, 0.f};
                    if (ALIBI) kb4 = *(const f4*)(kpos + key0 + blk * 32 + rg * 8 + kg * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int key = key0 + blk * 32 + rg * 8 + kg * 4 + j;
                        float sc = sacc[blk][rg * 4 + j] * p.scale_log2;
                        if (ALIBI) sc += slope * kb4[j];
                        if (!full_tile) sc = (key < row_vis_end && key < kend) ? sc : -INFINITY;
                        sacc[blk][rg * 4 + j] = sc;
                        mx = fmaxf(mx, sc);
                    }
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = fast_exp2(m_run - m_new);
            float rs = 0.f;
            h8 pb[4];                                            // P^T operands of the four 16-key steps
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = fast_exp2(sacc[blk][r] - m_new);
                    rs += e;
                    pb[blk * 2 + (r >> 3)][r & 7] = (_Float16)e;
                }
            rs += __shfl_xor(rs, 32);
            l_run = l_run * alpha + rs;
            if (__any(m_new > m_run)) {
#pragma unroll
                for (int db = 0; db < DB; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
            }
            m_run = m_new;
            // ---- O^T += V^T . P^T : four 16-key steps x DB head-dim blocks ----
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int k0 = t * 16 + kg * 4;                  // keys k0 .. k0+3 (slots j < 4) and k0+8 .. (j >= 4)
                const int vrow = k0 + ((n & 15) >> 2);           // source row of this lane inside its 16-lane group
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    const int dcol = db * 32 + (n & 16) + (n & 3) * 4;
                    const _Float16* vp = Vl + vrow * D + ((dcol + 32 * (vrow & 3)) & (D - 1));
                    const h4 lo = lds_tr_read(vp);               // keys k0 .. k0+3 at d = db*32 + (n & 31)
                    const h4 hi = lds_tr_read(vp + 8 * D);       // keys k0+8 .. : (vrow + 8) & 3 == vrow & 3, same rotation
                    const h8 a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pb[t], o[db], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    if (!(wave_active && qi < q_len)) return;
    // lane holds O^T[d][qi] for d = db*32 + 8*rg + 4*kg + j
    if (p.nsplit == 1) {
        const float inv = 1.0f / l_run;
        const int row = b * q_len + qi, KSo = p.H * D / 32;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int d = db * 32 + rg * 8 + kg * 4;
                if (p.of_hi) {
                    h4 hi, lo;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        _Float16 vh, vl;
                        pc_split(o[db][rg * 4 + j] * inv, vh, vl);
                        hi[j] = vh; lo[j] = vl;
                    }
                    const int64_t off = frag_off(row, h * D + d, KSo);
                    *(h4*)(p.of_hi + off) = hi;
                    *(h4*)(p.of_lo + off) = lo;
                } else {
                    h4 r, rl;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        _Float16 vh, vl;
                        pc_split(o[db][rg * 4 + j] * inv, vh, vl);
                        r[j] = vh; rl[j] = vl;
                    }
                    const int64_t ooff = b * p.o_bs + (int64_t)qi * p.o_ts + (int64_t)h * D + d;
                    *(h4*)(p.out + ooff) = r;
                    if (p.out_lo) *(h4*)(p.out_lo + ooff) = rl;
                }
            }
    } else {
        const int64_t slot = (((int64_t)b * p.H + h) * p.nsplit + split) * q_len + qi;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                f4 v = {o[db][rg * 4], o[db][rg * 4 + 1], o[db][rg * 4 + 2], o[db][rg * 4 + 3]};
                *(f4*)(p.part_o + slot * D + db * 32 + rg * 8 + kg * 4) = v;
            }
        if (kg == 0) { p.part_ml[slot * 2] = m_run; p.part_ml[slot * 2 + 1] = l_run; }
    }
}

// ---------------------------------------------------------------------------------------------------
// q_len <= 16 over a long staged cache (the cached prefill proper: q ~ 12 new tokens over S ~ 1.7 k staged keys).
// HBM-bound on the K/V stream, and at this size LATENCY-bound: the 64-rows-per-workgroup kernel above gives one active
// wave per workgroup a serial chain of 3 tiles behind barriers (12.4 us + a 5 us merge launch for 28.5 MB).  Here every
// WAVE owns a contiguous slice of the keys -- ceil(kv_len / (4 * nstream)) of them, one 64-key tile for the persona
// shape -- and runs with no barrier at all:
//   * K fragments go straight from global memory into the MFMA A operand (a lane's 16 bytes are exactly its fragment:
//     key row n, head dims 32 ks + 8 g ..; no LDS round trip for K);
//   * V rows are loaded coalesced, pass through a wave-private LDS tile and come back transposed (ds_read_b64_tr_b16);
//   * all of a tile's loads (32 KiB per wave, 128 KiB per workgroup) are issued before the first wait, so the whole
//     K/V stream of the launch is in flight after one round trip;
//   * the four waves' (m, l, O) partials are merged through LDS and leave as ONE partial per workgroup, so the merge
//     kernel reads nstream (+1) partials per row instead of one per 3-tile split.
// Q and P are split-precision pairs (HP).  Tail mode (see attn_tail_block) adds one workgroup per head for the pass's own
// rows; without it the new rows are part of the stream under the index-order causal mask.

// Fused split-KV merge: the workgroup has written its (m, l, O) partial through to memory; it arrives at the (batch row,
// head) counter and the LAST arriver merges all nsplit partials in split order (the result does not depend on who is last):
//     out = sum_i 2^(m_i - m*) O_i / sum_i 2^(m_i - m*) l_i           (what attn_combine_kernel does in a second launch)
// Hand-off: write-through payload stores, every storing wave drains (vmcnt 0), workgroup barrier, ONE relaxed agent-scope
// fetch-add; the last arriver reads the partials with agent-scope loads (they bypass its L1): no fence on either side, no
// spin anywhere.  The last arriver leaves the counter at zero for the next launch.
template <int D, int NS>
__device__ __forceinline__ void small_arrive_merge(const AttnParams& p, int b, int h, int* s_last) {
    const int tid = threadIdx.x;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        gu32* c = (gu32*)(p.counters + b * p.H + h);
        // (same hand-off contract as gemm_skinny_ks_kernel, pc_gemm_ks.hip: gfx9 vmcnt semantics; PC_FORMAL_HANDOFF=1 in the
        // environment selects the acq_rel arrival at launch time -- both forms live in the one binary, the suite runs both)
        const uint32_t old = p.formal_handoff ? __hip_atomic_fetch_add(c, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT)
                                              : __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (old + 1u == (uint32_t)p.nsplit) ? 1 : 0;
        if (last) __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_last = last;
    }
    __syncthreads();
    if (!*s_last) return;
    constexpr int DPT = D / 16;                          // head dims per thread: 16 threads per query row
    const int qi = tid >> 4, c = tid & 15, q_len = p.q_len, nsplit = p.nsplit;
    if (qi >= q_len) return;
    const int64_t base = ((int64_t)b * p.H + h) * nsplit;
    float mv[NS], lv[NS], ov[NS][DPT];
#pragma unroll
    for (int s = 0; s < NS; ++s) {                       // one batch of independent loads (clamped re-reads behind nsplit)
        const int sc = s < nsplit ? s : nsplit - 1;
        const int64_t slot = (base + sc) * q_len + qi;
        const float2 ml = ld_wt2(p.part_ml + slot * 2);
        mv[s] = s < nsplit ? ml.x : kNegBig;
        lv[s] = ml.y;
#pragma unroll
        for (int e = 0; e < DPT; e += 2) {
            const float2 t = ld_wt2(p.part_o + slot * D + c * DPT + e);
            ov[s][e] = t.x; ov[s][e + 1] = t.y;
        }
    }
    float mstar = kNegBig;
#pragma unroll
    for (int s = 0; s < NS; ++s) mstar = fmaxf(mstar, mv[s]);
    float num[DPT], den = 0.f;
#pragma unroll
    for (int e = 0; e < DPT; ++e) num[e] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const float w = s < nsplit ? exp2f(mv[s] - mstar) : 0.f;
        den += w * lv[s];
#pragma unroll
        for (int e = 0; e < DPT; ++e) num[e] += w * ov[s][e];
    }
    _Float16 hi[DPT], lo[DPT];
#pragma unroll
    for (int e = 0; e < DPT; ++e) pc_split(num[e] / den, hi[e], lo[e]);
    _Float16 *dh, *dl;
    if (p.of_hi) {
        const int64_t off = frag_off(b * q_len + qi, h * D + c * DPT, p.H * D / 32);   // DPT consecutive halfs of one fragment
        dh = p.of_hi + off; dl = p.of_lo + off;
    } else {
        const int64_t off = b * p.o_bs + (int64_t)qi * p.o_ts + (int64_t)h * D + c * DPT;
        dh = p.out + off; dl = p.out_lo ? p.out_lo + off : nullptr;
    }
    if constexpr (DPT == 8) {
        *(h8*)dh = h8{hi[0], hi[1], hi[2], hi[3], hi[4], hi[5], hi[6], hi[7]};
        if (dl) *(h8*)dl = h8{lo[0], lo[1], lo[2], lo[3], lo[4], lo[5], lo[6], lo[7]};
    } else if constexpr (DPT == 4) {
        *(h4*)dh = h4{hi[0], hi[1], hi[2], hi[3]};
        if (dl) *(h4*)dl = h4{lo[0], lo[1], lo[2], lo[3]};
    } else {
#pragma unroll
        for (int e = 0; e < DPT; ++e) { dh[e] = hi[e]; if (dl) dl[e] = lo[e]; }
    }
}

// NS > 0: ONE launch -- the split-KV partials are merged by the last-arriving workgroup of each head (small_arrive_merge,
// at most NS partials per row); NS = 0: the partials are left for attn_combine_kernel.
// GATHER (pc_attn gather_rows; B = 1): STAGE WHILE READING.  Every key row of the stream is read from where its entry of the row
// table (pc_kv_row_table) says it lies -- a module store for rows a fresh prompt stages, the arena itself for rows that are
// already there -- and rows that are not staged yet are written to the arena as they pass: K from the MFMA operand registers,
// V from the wave's LDS tile.  Each (kv head, key row) belongs to exactly one wave of one workgroup, so after the launch the
// arena planes hold what pc_kv_gather would have put there, bit for bit; the module K/V crossed the chip once.  The 64 table
// entries of a tile go through a wave-private LDS slice (one 16-byte load per lane, then the 4 + LPW entries a lane needs for
// its K fragments and V chunks come back as ds_read_b128).  Stores are issued behind the tile's last wait: on gfx9 stores
// count in vmcnt with the loads, and a store in front of the V wait would put its write acknowledgement on the critical path.
// (The launch is sized to ONE workgroup per CU -- small_nstream -- so the staging variant, which keeps the K fragments alive until
// they are stored, takes the registers of a one-wave-per-SIMD kernel instead of spilling at 256.)
// RT: 16-row tiles of query rows per wave (1: <= 16 new rows; 2: 17..32 -- every K fragment and V tile read serves both tiles;
// one workgroup per CU, as the staging variant).
template <int D, bool ALIBI, int NS, bool GATHER = false, int RT = 1>
__global__ __launch_bounds__(kThreads, (GATHER || RT > 1) ? 1 : 2) void attn_small_kernel(const AttnParams p) {
    constexpr bool FUSE = NS > 0;
    static_assert(!(FUSE && GATHER), "the in-launch merge and the staging stream are separate instantiations");
    static_assert(RT == 1 || (RT == 2 && !FUSE && 16 * RT <= kTailMax), "two row tiles: the two-launch form");
    constexpr int KS = D / 32, DB = D / 16, CPR = D / 8;
    constexpr int LPW = kTK * CPR / 64;              // 16-byte V chunks per lane per tile
    constexpr int kTileHalfs = kTK * D;
    constexpr int kTailRows = 16 * RT;
    constexpr int kTailBytes = (2 * kTailRows * D + kTailRows * D + kTailRows * kTailRows) * 4;
    constexpr int kMergeBytes = RT * (4 * (D / 16) * 64 * 4 + 4 * 16 * 2) * 4;      // the four waves' (O, m, l) per row tile
    constexpr int kLds0 = 4 * kTileHalfs * 2 > kTailBytes ? 4 * kTileHalfs * 2 : kTailBytes;
    constexpr int kLdsBytes = kLds0 > kMergeBytes ? kLds0 : kMergeBytes;
    constexpr int kTabBytes = GATHER ? 4 * kTK * 16 : 0;           // row-table entries of the four waves' tiles
    __shared__ __attribute__((aligned(16))) char smem[kLdsBytes + 16 + kTabBytes];
    int* s_last = (int*)(smem + kLdsBytes);

    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = blockIdx.y;
    const int b = blockIdx.z / p.nsplit, split = blockIdx.z - b * p.nsplit;
    const int hkv = h / (p.H / p.Hkv);
    const int q_len = p.q_len;
    const int past_len = row_past_len(p, 0);
    const int nstream = p.tail ? p.nsplit - 1 : p.nsplit;
    auto stamp = [&](int slot) {
        if (p.trace && lane == 0)
            p.trace[((((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * 4 + wave) * 4) + slot] = wall_clock64();
    };
    stamp(0);
    if (p.tail && split == nstream) {                 // workgroup-uniform: the pass's own rows, fp32
        float* f = (float*)smem;
        attn_tail_block<D, kTailRows, ALIBI, FUSE>(p, f, f + kTailRows * D, f + 2 * kTailRows * D, f + 3 * kTailRows * D, b, h, split);
        if constexpr (FUSE) small_arrive_merge<D, NS>(p, b, h, s_last);
        return;
    }
    const int kv_len = p.tail ? past_len : past_len + q_len;
    int cpw = (kv_len + nstream * 4 - 1) / (nstream * 4);          // keys per wave
    cpw = (cpw + 15) & ~15;
    const int k0 = (split * 4 + wave) * cpw;
    const int k1 = (k0 + cpw < kv_len) ? k0 + cpw : kv_len;
    int qi[RT], row_vis_end[RT];                                   // this lane's query row of each row tile
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        qi[rt] = rt * 16 + n;
        row_vis_end[rt] = qi[rt] < q_len ? (p.tail ? past_len : past_len + qi[rt] + 1) : 0;
    }

    // Every load below is unconditional (rows / keys past the end are clamped to the last valid one and masked later): a load
    // under a branch makes hipcc's vmcnt bookkeeping wait for the NEWEST loads at the first use, which would put the whole
    // V stream behind the K round trip.
    h8 qf[RT][KS], qfl[RT][KS];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int qc = qi[rt] < q_len ? qi[rt] : q_len - 1;
        const _Float16* qlo = p.q_lo ? p.q_lo : p.q;          // (no lo plane: a finite stand-in, multiplied by zero below)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int64_t off = b * p.q_bs + (int64_t)qc * p.q_ts + (int64_t)h * D + ks * 32 + g * 8;
            qf[rt][ks] = *(const h8*)(p.q + off);
            qfl[rt][ks] = *(const h8*)(qlo + off);
        }
        if (!p.q_lo) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { h8 z = {0, 0, 0, 0, 0, 0, 0, 0}; qfl[rt][ks] = z; }
        }
    }
    f4 o[RT][DB];
    float m_run[RT], l_run[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
        for (int db = 0; db < DB; ++db) { f4 z = {0.f, 0.f, 0.f, 0.f}; o[rt][db] = z; }
        m_run[rt] = kNegBig; l_run[rt] = 0.f;
    }
    const _Float16* kbase = p.k + b * p.kv_bs + (int64_t)hkv * p.kv_hs;
    const _Float16* vbase = p.v + b * p.kv_bs + (int64_t)hkv * p.kv_hs;
    [[maybe_unused]] const float slope = ALIBI ? p.slopes[h] : 0.f;
    [[maybe_unused]] const float* kpos = ALIBI ? p.key_pos + b * p.kp_bs : nullptr;
    _Float16* Vw = (_Float16*)smem + wave * kTileHalfs;            // this wave's V tile
    char* Vwb = smem + wave * kTileHalfs * 2;                      // (the same, as the wave-uniform LDS-DMA base)
    [[maybe_unused]] u32x4* etab = (u32x4*)(smem + kLdsBytes + 16) + wave * kTK;   // GATHER: this wave's 64 row-table entries
    [[maybe_unused]] const bool g_writer = GATHER && (h % (p.H / p.Hkv)) == 0;        // one query head per kv head stages
    [[maybe_unused]] const uint64_t g_koff = GATHER ? (uint64_t)(uint32_t)(p.g_kplane + hkv) : 0;
    [[maybe_unused]] const uint64_t g_voff = GATHER ? (uint64_t)(uint32_t)(p.g_vplane + hkv) : 0;

    for (int key0 = k0; key0 < k1; key0 += kTK) {
        // ---- every load of the tile first: K as MFMA fragments (registers), V by LDS-DMA into the wave's tile ----
        u32x4 kr[4][KS];
        [[maybe_unused]] uint32_t kflag[4];
        if constexpr (GATHER) {
            // lane = row of the tile: its entry becomes {address of the row in this head's K plane | STAGED flag in bit 0,
            // address in the V plane} (rows are 16-byte aligned: the low address bits are free)
            const int ek = key0 + lane < k1 ? key0 + lane : k1 - 1;
            const u32x4 e = *(const u32x4*)(p.rows + ek);
            const uint64_t base = ((uint64_t)e[1] << 32) | e[0];
            const uint64_t ka = (base + ((g_koff * e[2]) << 4)) | (e[3] & PC_KV_ROW_STAGED);
            const uint64_t va = base + ((g_voff * e[2]) << 4);
            etab[lane] = u32x4{(uint32_t)ka, (uint32_t)(ka >> 32), (uint32_t)va, (uint32_t)(va >> 32)};
            __builtin_amdgcn_wave_barrier();             // (a wave's LDS operations execute in order: no wait needed)
        }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const int key = key0 + kb * 16 + n < k1 ? key0 + kb * 16 + n : k1 - 1;     // (keys past k1 are masked below)
            if constexpr (GATHER) {
                const uint64_t ka = *(const uint64_t*)(etab + (key - key0));
                const char* src = (const char*)(uintptr_t)(ka & ~(uint64_t)15);
                kflag[kb] = (uint32_t)ka & PC_KV_ROW_STAGED;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
#ifndef PC_GATHER_LOAD_PLAIN
                    kr[kb][ks] = __builtin_nontemporal_load((const u32x4*)(src + (ks * 32 + g * 8) * 2));
#else
                    kr[kb][ks] = *(const u32x4*)(src + (ks * 32 + g * 8) * 2);
#endif
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) kr[kb][ks] = *(const u32x4*)(kbase + (int64_t)key * D + ks * 32 + g * 8);
            }
        }
        // V rows go straight to LDS, rotated by 32 B per row (see attn_fwd_kernel) -- the permutation sits on the per-lane
        // SOURCE address, LDS-DMA writes lane-linearly: chunk c = 64 i + lane of the tile = (row c / CPR, position c % CPR)
        // holds source column (position - 2 (row & 7)) mod CPR.  Rows past k1 re-read the last valid row (their P is 0; a
        // finite value times 0, where uninitialised arena rows could hold NaNs).
#pragma unroll
        for (int i = 0; i < LPW; ++i) {
            const int c = lane + i * 64, row = c / CPR, pos = c - row * CPR;
            const int col = (pos - 2 * (row & 7)) & (CPR - 1);
            const int rr = key0 + row < k1 ? key0 + row : k1 - 1;
            if constexpr (GATHER) {
                const char* src = (const char*)(uintptr_t)((const uint64_t*)(etab + (rr - key0)))[1];
#ifndef PC_GATHER_LOAD_PLAIN
                glds16_nt((const _Float16*)(src + col * 16), Vwb + i * 1024);
#else
                glds16((const _Float16*)(src + col * 16), Vwb + i * 1024);
#endif
            } else {
                glds16(vbase + (int64_t)rr * D + col * 8, Vwb + i * 1024);
            }
        }
        __builtin_amdgcn_sched_barrier(0);               // the whole tile is in flight before the first wait
        // ---- S^T = K . Q^T, online softmax: per row tile, on the same K fragments ----
        h8 pb[RT][2], pbl[RT][2];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            float sv[4][4];
            float mx = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const h8 a = __builtin_bit_cast(h8, kr[kb][ks]);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, qf[rt][ks], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, qfl[rt][ks], acc, 0, 0, 0);
                }
                f4 kb4 = {0.f, 0.f, 0.f, 0.f};
                if (ALIBI) kb4 = *(const f4*)(kpos + key0 + kb * 16 + g * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = key0 + kb * 16 + g * 4 + r;
                    float sc = acc[r] * p.scale_log2;
                    if (ALIBI) sc += slope * kb4[r];
                    const float s = (key < row_vis_end[rt] && key < k1) ? sc : -INFINITY;
                    sv[kb][r] = s;
                    mx = fmaxf(mx, s);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run[rt], mx);
            const float alpha = fast_exp2(m_run[rt] - m_new);
            float rs = 0.f;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = fast_exp2(sv[kb][r] - m_new);
                    rs += e;
                    const _Float16 eh = (_Float16)e;
                    pb[rt][kb >> 1][(kb & 1) * 4 + r] = eh;
                    pbl[rt][kb >> 1][(kb & 1) * 4 + r] = (_Float16)(e - (float)eh);
                }
            rs += __shfl_xor(rs, 16);
            rs += __shfl_xor(rs, 32);
            if (rt == 0 && key0 == k0) stamp(1);         // K arrived, scores + softmax of the first tile done
            l_run[rt] = l_run[rt] * alpha + rs;
#pragma unroll
            for (int db = 0; db < DB; ++db) { o[rt][db][0] *= alpha; o[rt][db][1] *= alpha; o[rt][db][2] *= alpha; o[rt][db][3] *= alpha; }
            m_run[rt] = m_new;
        }
        // ---- the V tile has landed (this wave's own DMA: vmcnt covers it, no barrier), back transposed ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int db = 0; db < DB; ++db) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int vrow = t * 32 + g * 4 + (n >> 2);
                const _Float16* vp = Vw + vrow * D + ((db * 16 + (n & 3) * 4 + 16 * (vrow & 7)) & (D - 1));
                const h4 lo = lds_tr_read(vp);
                const h4 hi = lds_tr_read(vp + 16 * D);
                const h8 a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    o[rt][db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pb[rt][t], o[rt][db], 0, 0, 0);
                    o[rt][db] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, pbl[rt][t], o[rt][db], 0, 0, 0);
                }
            }
        }
        if constexpr (GATHER) {
            // ---- stage the tile: rows that are not in the arena yet leave for it (K from registers, V from the LDS tile) ----
            if (g_writer) {
                _Float16* kdst = const_cast<_Float16*>(kbase);
                _Float16* vdst = const_cast<_Float16*>(vbase);
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    const int key = key0 + kb * 16 + n;
                    if (key < k1 && !(kflag[kb] & PC_KV_ROW_STAGED)) {
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks)
                            PC_GATHER_ST(kr[kb][ks], (u32x4*)(kdst + (int64_t)key * D + ks * 32 + g * 8));
                    }
                }
#pragma unroll
                for (int i = 0; i < LPW; ++i) {
                    const int c = lane + i * 64, row = c / CPR, pos = c - row * CPR;
                    const int col = (pos - 2 * (row & 7)) & (CPR - 1);
                    const u32x4 chunk = *(const u32x4*)(Vwb + i * 1024 + lane * 16);
                    const uint32_t fl = (key0 + row < k1) ? ((const uint32_t*)(etab + row))[0] : PC_KV_ROW_STAGED;
                    if (!(fl & PC_KV_ROW_STAGED))
                        PC_GATHER_ST(chunk, (u32x4*)(vdst + (int64_t)(key0 + row) * D + col * 8));
                }
            }
        }
    }

    stamp(2);                                            // the wave's key slice is done
    // ---- merge the four waves' partials through LDS: one (m, l, O) partial per workgroup ----
    __syncthreads();                                     // every wave is done with its V tile
    float* mo = (float*)smem;                            // [RT][4][DB][64][4]
    float* mml = mo + RT * 4 * DB * 64 * 4;              // [RT][4][16][2]
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
        for (int db = 0; db < DB; ++db) *(f4*)(mo + (((rt * 4 + wave) * DB + db) * 64 + lane) * 4) = o[rt][db];
        if (g == 0) { mml[((rt * 4 + wave) * 16 + n) * 2] = m_run[rt]; mml[((rt * 4 + wave) * 16 + n) * 2 + 1] = l_run[rt]; }
    }
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        float mw[4], lw[4], mstar = kNegBig;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            mw[w] = mml[((rt * 4 + w) * 16 + n) * 2]; lw[w] = mml[((rt * 4 + w) * 16 + n) * 2 + 1];
            mstar = fmaxf(mstar, mw[w]);
        }
        float wt[4], lsum = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { wt[w] = fast_exp2(mw[w] - mstar); lsum += wt[w] * lw[w]; }
        if (qi[rt] < q_len) {
            const int64_t slot = (((int64_t)b * p.H + h) * p.nsplit + split) * q_len + qi[rt];
            constexpr int DPW = (DB + 3) / 4;                // head-dim blocks merged by one wave
#pragma unroll
            for (int j = 0; j < DPW; ++j) {
                const int db = wave * DPW + j;
                if (db < DB) {
                    f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const f4 x = *(const f4*)(mo + (((rt * 4 + w) * DB + db) * 64 + lane) * 4);
                        acc[0] += wt[w] * x[0]; acc[1] += wt[w] * x[1]; acc[2] += wt[w] * x[2]; acc[3] += wt[w] * x[3];
                    }
                    float* dst = p.part_o + slot * D + db * 16 + g * 4;
                    if constexpr (FUSE) { st_wt2(dst, acc[0], acc[1]); st_wt2(dst + 2, acc[2], acc[3]); }
                    else *(f4*)dst = acc;
                }
            }
            if (wave == 0 && g == 0) {
                if constexpr (FUSE) st_wt2(p.part_ml + slot * 2, mstar, lsum);
                else { p.part_ml[slot * 2] = mstar; p.part_ml[slot * 2 + 1] = lsum; }
            }
        }
    }
    if (p.trace) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(3); }
    if constexpr (FUSE) small_arrive_merge<D, NS>(p, b, h, s_last);
}

// Streaming workgroups per head of attn_small_kernel (each = 4 key slices): ONE workgroup per CU in total, the tail workgroup
// of each head included -- 8 + 1 splits x 32 heads = 288 workgroups put a second workgroup on 32 of the 256 CUs and the launch
// waits for those: 7 + 1 (exactly 256) runs the persona step 1.7 % faster end to end (3.908 -> 3.844 ms, profiles/r03_*).
int small_nstream(int B, int H, int tail = 0) {
    static const int forced = [] { const char* e = getenv("PC_ATTN_SMALL_WG"); return e ? atoi(e) : 0; }();
    if (forced > 0) return forced < 15 ? forced : 15;
    int ns = 256 / (B * H) - tail;
    if (ns < 1) ns = 1;
    if (ns > 15) ns = 15;
    return ns;
}

// Merge split-KV partials: out = sum_i 2^(m_i - m*) O_i / sum_i 2^(m_i - m*) l_i.
// One workgroup per (query row, head), one thread per head dim.  (A variant with one workgroup per head and
// float4 items measured 8.0 us vs 6.3 us per launch inside the captured forward: more parallel, shorter chains win.)
// RPW > 1 (64 and more rows): a workgroup merges RPW rows one after the other -- at 259 rows x 40 heads the one-row form is
// 10 360 two-wave workgroups and the launch is bound by their dispatch (9 us per layer of BASELINE config 4)
template <int D, int NS, int RPW = 1>   // NS: compile-time bound on nsplit, so every partial is loaded before anything is used
__global__ void attn_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                    _Float16* __restrict__ out, int64_t o_bs, int64_t o_ts,
                                    _Float16* __restrict__ of_hi, _Float16* __restrict__ of_lo, int H, int q_len,
                                    int nsplit, _Float16* __restrict__ out_lo) {
    const int h = blockIdx.y, b = blockIdx.z, d = threadIdx.x;
    const int64_t base = ((int64_t)b * H + h) * nsplit;
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int qi = (int)blockIdx.x * RPW + rr;
    if (RPW > 1 && qi >= q_len) break;
    // one batch of independent loads (a loop over a runtime nsplit serialises them: max first, then one dependent
    // (m, l, o) round trip per split -- 6.2 us per launch for 1.9 MB of partials)
    float mv[NS], lv[NS], ov[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int sc = s < nsplit ? s : nsplit - 1;           // clamped re-read instead of a branch around the load
        const int64_t slot = (base + sc) * q_len + qi;
        const float2 ml = *(const float2*)(part_ml + slot * 2);
        mv[s] = s < nsplit ? ml.x : kNegBig;
        lv[s] = ml.y;
        ov[s] = part_o[slot * D + d];
    }
    float mstar = kNegBig;
#pragma unroll
    for (int s = 0; s < NS; ++s) mstar = fmaxf(mstar, mv[s]);
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const float w = s < nsplit ? exp2f(mv[s] - mstar) : 0.f;
        den = __builtin_fmaf(w, lv[s], den);         // (explicit: pc_gemm_q8.hip's consumer-side merge forms the same bits)
        num = __builtin_fmaf(w, ov[s], num);
    }
    const float v = num / den;
    if (of_hi) {
        const int64_t off = frag_off(b * q_len + qi, h * D + d, H * D / 32);
        _Float16 hi, lo;
        pc_split(v, hi, lo);
        of_hi[off] = hi;
        of_lo[off] = lo;
    } else {
        _Float16 hi, lo;
        pc_split(v, hi, lo);
        const int64_t off = b * o_bs + (int64_t)qi * o_ts + (int64_t)h * D + d;
        out[off] = hi;
        if (out_lo) out_lo[off] = lo;
    }
  }
}

// The 32-rows-per-wave kernel (attn_fwd32_kernel) halves the LDS traffic per flop but also the number of workgroups:
// it wins once 128-row q-blocks alone fill the chip (q = S = 1737: 81 -> 69 us, 4404: 322 -> 274 us) or when a long
// staged past supplies the parallelism through KV splits (q = 260 over S = 8000: 144 -> 131 us); with fewer blocks the
// causal triangle leaves too few, too unequal workgroups (q = S = 456: 16 -> 25 us).
bool use_rows32(int B, int H, int q_len, int kv_len) {
    static const bool off = [] { const char* e = getenv("PC_ATTN_NO32"); return e && e[0] == '1'; }();
    if (off || q_len <= kQB) return false;
    return B * H * pc_ceil_div(q_len, kQB32) >= 256 || kv_len >= 8 * q_len;
}

// hp: the launch carries split-precision Q (the 64-rows-per-workgroup kernel whatever the size)
int choose_nsplit(int B, int H, int q_len, int kv_len, bool hp) {
    static const int forced = [] { const char* e = getenv("PC_ATTN_NSPLIT"); return e ? atoi(e) : 0; }();
    const int nqblk = pc_ceil_div(q_len, (!hp && use_rows32(B, H, q_len, kv_len)) ? kQB32 : kQB);
    const int base = B * H * nqblk;
    // ~1.25 workgroups per CU: fewer, longer KV streams per workgroup beat many short ones (measured on the
    // persona shape: 10 splits x 32 heads = 18.7 us vs 13 splits 20.3 us vs 4 splits 23.8 us per layer)
    int ns = forced > 0 ? forced : (320 + base / 2) / (base > 0 ? base : 1);
    const int max_by_len = kv_len / (2 * kTK);                   // keep >= 2 tiles per split
    if (forced <= 0 && q_len > kSmallQ && base > 0 && kv_len >= 4 * q_len) {
        // Many query rows over a much longer cache (a long question in front of staged modules: 259 rows over 8.3 k keys at the
        // 13b shape; not the causal q ~ kv regime of an encode, whose key ranges differ per query block): a workgroup walks kv_len / (64 ns) key tiles and two workgroups share a CU, so what counts is how evenly
        // base * ns workgroups fill rounds of 512 slots.  Cost in key-tile units: rounds x (tiles per split + 1) + the merge.
        // Measured (tools/attn_mid.py, 40 heads, q = 259, 8.3 k keys): 2 splits 252 us, 3: 276, 4: 247, 5: 218, 10: 231;
        // (32 heads, q = 100, 1.7 k keys) 2: 52, 4: 34.5, 5: 33, 8: 31 -- the rule below picks 5 and 8.
        const int T = pc_ceil_div(kv_len, kTK);
        int best = 1;
        float best_cost = 1e30f;
        const int lim = max_by_len < kMaxSplit ? (max_by_len < 1 ? 1 : max_by_len) : kMaxSplit;
        for (int c = 1; c <= lim; ++c) {
            const float cost = (float)pc_ceil_div(base * c, 512) * (float)(pc_ceil_div(T, c) + 1) + (c > 1 ? 3.f + 0.5f * c : 0.f);
            if (cost < best_cost) { best_cost = cost; best = c; }
        }
        ns = best;
    }
    if (ns > max_by_len) ns = max_by_len;
    if (ns > kMaxSplit) ns = kMaxSplit;
    if (ns < 1) ns = 1;
    return ns;
}

// Streaming splits of a tail-mode launch of the 64-row kernel (<= 32 new rows, their own keys in the tail workgroup's fp32 pass):
// the merge launch runs anyway, the streams cover the staged keys only, and what a short question over a SHORT cache pays is the
// serial walk of one workgroup over its key tiles -- choose_nsplit's "two tiles per split at least" left 5 tiles to one workgroup
// at 300 staged keys.  Here: as many streams as there are tiles while the grid stays within two workgroups per CU, then the
// fewest streams with the same tiles per stream.  7b, 22 / 30 new rows: 300 staged keys 4.50 -> 4.24 ms, 1 727: 4.57 -> 4.50 ms
// (profiles/r04_variants.txt).
int tail_stream_splits(int B, int H, int past_len) {
    static const int forced = [] { const char* e = getenv("PC_ATTN_NSPLIT"); return e ? atoi(e) : 0; }();
    const int T = pc_ceil_div(past_len > 0 ? past_len : 1, kTK);
    int cmax = 512 / (B * H > 0 ? B * H : 1) - 1;
    if (cmax > kMaxSplit - 1) cmax = kMaxSplit - 1;
    if (cmax < 1) cmax = 1;
    if (forced > 0) return forced < cmax ? forced : cmax;
    int c = T < cmax ? T : cmax;
    const int per = pc_ceil_div(T, c);
    while (c > 1 && pc_ceil_div(T, c - 1) == per) --c;
    return c;
}

template <int D>
int launch_attn(const AttnParams& p0, int B, hipStream_t stream) {
    AttnParams p = p0;
    static const bool no_remap = [] { const char* e = getenv("PC_ATTN_NO_XCD"); return e && e[0] == '1'; }();
    // a q_lo plane asks for split-precision Q and P at any q_len (the many-row path in its precise mode): 16-row kernel
    const bool want_hp = p.q_len <= kQB || p.q_lo != nullptr;
    const bool rows32 = !p.q_lo && !p.k_lo && !p.pre_k && use_rows32(B, p.H, p.q_len, p.past_len + p.q_len);
    p.nqblk = pc_ceil_div(p.q_len, rows32 ? kQB32 : kQB);
    p.nbatch = B;
    p.xcd_remap = (!p.small && p.nsplit == 1 && p.nqblk >= (rows32 ? 2 : 4) && !no_remap) ? 1 : 0;
    dim3 grid(p.nqblk, p.H, B * p.nsplit);
    if (p.xcd_remap) grid = dim3(8 * p.nqblk * ((p.H * B + 7) / 8), 1, 1);
    const bool ring = D == 128 && ring_eligible(p, D);      // (attn_fwd_impl chose p.nsplit for it)
#ifdef PC_DEV_SWEEPS      // (the merge inside the launch: built, bit-reproducible, and no faster than the second launch in three rounds of
                          // measurements -- dev builds only; the product library ignores `counters` and merges in a second launch)
    if (!ring && p.small && p.counters) {
        // one launch: the last-arriving workgroup of each head merges the partials (no attn_combine_kernel)
#define PC_SMALL_FUSED(NSV)                                                                                          \
        do {                                                                                                       \
            if (p.key_pos) hipLaunchKernelGGL((attn_small_kernel<D, true, NSV>), grid, dim3(kThreads), 0, stream, p);  \
            else hipLaunchKernelGGL((attn_small_kernel<D, false, NSV>), grid, dim3(kThreads), 0, stream, p);          \
        } while (0)
        if (p.nsplit <= 4) PC_SMALL_FUSED(4); else if (p.nsplit <= 8) PC_SMALL_FUSED(8); else PC_SMALL_FUSED(16);
#undef PC_SMALL_FUSED
        return pc_check_launch("attn_small_kernel");
    }
#else
    p.counters = nullptr;
#endif
    if (ring) {
        // > 64 split-precision rows at head_dim 128: 128 rows per workgroup, tiles by LDS-DMA (pc_attn_ring.hip)
        const int rrc = p.wide ? launch_attn_wide(p, stream) : launch_attn_ring(p, B, stream);
        if (rrc != PC_OK) return rrc;
    } else if (p.small && p.rows) {       // stage while reading (pc_attn gather_rows)
        if (p.key_pos) hipLaunchKernelGGL((attn_small_kernel<D, true, 0, true>), grid, dim3(kThreads), 0, stream, p);
        else if (p.q_len > kSmallQ) hipLaunchKernelGGL((attn_small_kernel<D, false, 0, true, 2>), grid, dim3(kThreads), 0, stream, p);
        else hipLaunchKernelGGL((attn_small_kernel<D, false, 0, true>), grid, dim3(kThreads), 0, stream, p);
    } else if (p.small) {
        if (p.key_pos) hipLaunchKernelGGL((attn_small_kernel<D, true, 0>), grid, dim3(kThreads), 0, stream, p);
        else if (p.q_len > kSmallQ) hipLaunchKernelGGL((attn_small_kernel<D, false, 0, false, 2>), grid, dim3(kThreads), 0, stream, p);
        else hipLaunchKernelGGL((attn_small_kernel<D, false, 0>), grid, dim3(kThreads), 0, stream, p);
    } else if (rows32) {
        if (p.key_pos) hipLaunchKernelGGL((attn_fwd32_kernel<D, true>), grid, dim3(kThreads), 0, stream, p);
        else hipLaunchKernelGGL((attn_fwd32_kernel<D, false>), grid, dim3(kThreads), 0, stream, p);
    } else if (p.pre_k) {        // shared prefix (pc_attn checked: past_lens, no ALiBi, many-row kernel)
        if (p.k_lo) hipLaunchKernelGGL((attn_fwd_pre_kernel<D, true, true>), grid, dim3(kThreads), 0, stream, p);
        else if (want_hp) hipLaunchKernelGGL((attn_fwd_pre_kernel<D, true, false>), grid, dim3(kThreads), 0, stream, p);
        else hipLaunchKernelGGL((attn_fwd_pre_kernel<D, false, false>), grid, dim3(kThreads), 0, stream, p);
    } else if (p.k_lo && !p.tail) {
        if (p.key_pos) hipLaunchKernelGGL((attn_fwd_kernel<D, true, true, true>), grid, dim3(kThreads), 0, stream, p);
        else hipLaunchKernelGGL((attn_fwd_kernel<D, true, false, true>), grid, dim3(kThreads), 0, stream, p);
    } else if (p.key_pos) {
        if (want_hp) hipLaunchKernelGGL((attn_fwd_kernel<D, true, true>), grid, dim3(kThreads), 0, stream, p);
        else hipLaunchKernelGGL((attn_fwd_kernel<D, false, true>), grid, dim3(kThreads), 0, stream, p);
    } else if (want_hp && p.rows) hipLaunchKernelGGL((attn_fwd_kernel<D, true, false, false, true>), grid, dim3(kThreads), 0, stream, p);
    else if (want_hp) hipLaunchKernelGGL((attn_fwd_kernel<D, true>), grid, dim3(kThreads), 0, stream, p);
    else hipLaunchKernelGGL((attn_fwd_kernel<D, false>), grid, dim3(kThreads), 0, stream, p);
    int rc = pc_check_launch("attn_fwd_kernel");
    if (rc != PC_OK) return rc;
    // defer_merge: the <= 16-row streaming kernel leaves its partials (part_o [B*H*nsplit*q_len][D], part_ml behind them, at the
    // start of the workspace) to a consumer that merges them in its own prologue; every other shape merges as usual and says so
    const bool deferred = p.defer_merge && p.small && !ring && p.nsplit > 1 && p.nsplit <= 8 && B == 1 && !p.out_lo;
    if (p.nsplit_out) *p.nsplit_out = deferred ? p.nsplit : 1;
    if (p.nsplit > 1 && !deferred) {
#define PC_COMBINE(NSV)                                                                                         \
        do {                                                                                                    \
            if (p.q_len >= 64)                                                                                  \
                hipLaunchKernelGGL((attn_combine_kernel<D, NSV, 4>), dim3(pc_ceil_div(p.q_len, 4), p.H, B), dim3(D), 0, stream, \
                                   p.part_o, p.part_ml, p.out, p.o_bs, p.o_ts, p.of_hi, p.of_lo, p.H, p.q_len, p.nsplit, p.out_lo); \
            else                                                                                                \
                hipLaunchKernelGGL((attn_combine_kernel<D, NSV>), dim3(p.q_len, p.H, B), dim3(D), 0, stream, p.part_o, \
                                   p.part_ml, p.out, p.o_bs, p.o_ts, p.of_hi, p.of_lo, p.H, p.q_len, p.nsplit, p.out_lo); \
        } while (0)
        if (p.nsplit <= 4) PC_COMBINE(4);
        else if (p.nsplit <= 8) PC_COMBINE(8);
        else if (p.nsplit <= 16) PC_COMBINE(16);
        else PC_COMBINE(32);
#undef PC_COMBINE
        rc = pc_check_launch("attn_combine_kernel");
    }
    return rc;
}

}  // namespace

PC_EXPORT int64_t pc_attn_workspace_bytes(int32_t B, int32_t H, int32_t D, int32_t q_len, int32_t kv_len_max) {
    if (B <= 0 || H <= 0 || D <= 0 || q_len <= 0) return 0;
    // (the launch may or may not carry split-precision Q: room for the larger split count)
    const int ns_a = choose_nsplit(B, H, q_len, kv_len_max, false), ns_b = choose_nsplit(B, H, q_len, kv_len_max, true);
    int ns = ns_a > ns_b ? ns_a : ns_b;
    // passes of <= kTailMax rows may run in tail mode (pc_attn_fwd_ex with lo_row0 = -1): one more split, always merged
    if (q_len <= kTailMax) {
        const int nt = tail_stream_splits(B, H, kv_len_max);
        ns = ns > nt ? ns : nt;
        ns = (ns < kMaxSplit ? ns : kMaxSplit - 1) + 1;
    }
    // passes of <= 16 rows may take attn_small_kernel: one partial per workgroup (+ the tail's)
    if (q_len <= 2 * kSmallQ && ns < small_nstream(B, H) + 1) ns = small_nstream(B, H) + 1;   // (17..32 rows: two row tiles per wave)
    if (D == 128 && q_len >= ring_min_rows()) { const int nr = ring_nsplit(B, H, q_len, kv_len_max); ns = ns > nr ? ns : nr; }   // (pc_attn_ring.hip)
    if (D == 128 && B == 1 && q_len >= ring_min_rows() && kv_len_max - q_len >= wide_min_keys()) {      // (pc_attn_wide.hip: staged keys = kv_len - q_len)
        const int nw = wide_nsplit(H, q_len, kv_len_max - q_len) + 1;
        ns = ns > nw ? ns : nw;
    }
    if (ns <= 1) return 0;
    return (int64_t)B * H * ns * q_len * (D + 2) * (int64_t)sizeof(float);
}

// dev hook: the next attention launches of this thread stamp per-wave wall-clock times into `buf` (NULL: off)
static thread_local unsigned long long* g_attn_trace = nullptr;
PC_EXPORT int pc_dev_attn_trace(void* buf) { g_attn_trace = (unsigned long long*)buf; return PC_OK; }

namespace {
int attn_fwd_impl(const void* q, const void* q_lo, int64_t q_batch_stride, int64_t q_token_stride, const void* k,
                  const void* v, int64_t kv_batch_stride, int64_t kv_head_stride, void* out,
                  int64_t out_batch_stride, int64_t out_token_stride, int32_t B, int32_t H, int32_t Hkv,
                  int32_t D, int32_t q_len, int32_t past_len, float softmax_scale, void* workspace,
                  int64_t workspace_bytes, const int32_t* past_len_dev, void* out_frag_hi, void* out_frag_lo,
                  const float* key_pos, int64_t key_pos_batch_stride, const float* slopes, void* out_lo,
                  const void* k_lo, const void* v_lo, int64_t lo_bs, int64_t lo_hs, int32_t lo_row0, void* stream,
                  const int32_t* past_lens = nullptr, uint32_t* counters = nullptr, const void* pre_k = nullptr,
                  const void* pre_v = nullptr, const void* pre_k_lo = nullptr, const void* pre_v_lo = nullptr, int64_t pre_hs = 0,
                  const pc_kv_row* gather_rows = nullptr, int32_t g_kplane = 0, int32_t g_vplane = 0, int* gather_ok = nullptr,
                  int32_t defer_merge = 0, int32_t* nsplit_out = nullptr) {
#ifndef PC_DEV_SWEEPS
    counters = nullptr;       // (the in-launch merge exists in dev builds only: the product merges in a second launch, bit for bit the same)
#endif
    // gather_ok != NULL: dry run -- *gather_ok = whether this launch shape would take gather_rows; nothing is launched
    PC_REQUIRE(B > 0 && H > 0 && Hkv > 0 && H % Hkv == 0 && q_len >= 0 && past_len >= 0, PC_ERR_ARG,
               "pc_attn_fwd: bad sizes");
    PC_REQUIRE(D == 32 || D == 64 || D == 128, PC_ERR_ARG, "pc_attn_fwd: head_dim %d unsupported (32/64/128)", D);
    if (q_len == 0) return PC_OK;
    PC_REQUIRE(q && k && v && (out || (out_frag_hi && out_frag_lo)), PC_ERR_ARG, "pc_attn_fwd: null pointer");
    PC_REQUIRE((out_frag_hi == nullptr) == (out_frag_lo == nullptr), PC_ERR_ARG, "pc_attn_fwd: need both fragment planes");
    PC_REQUIRE(!out_frag_hi || (B * q_len <= 512 && (H * D) % 32 == 0), PC_ERR_ARG,
               "pc_attn_fwd: fragment-plane output is for the weight-streaming regime (B*q_len <= 512)");
    PC_REQUIRE(q_token_stride % 8 == 0 && kv_head_stride % 8 == 0 && out_token_stride % 4 == 0, PC_ERR_ARG,
               "pc_attn_fwd: strides must keep 16-byte (q, kv) / 8-byte (out) alignment");
    AttnParams p;
    p.q = (const _Float16*)q; p.q_bs = q_batch_stride; p.q_ts = q_token_stride;
    p.q_lo = (const _Float16*)q_lo;
    p.k = (const _Float16*)k; p.v = (const _Float16*)v; p.kv_bs = kv_batch_stride; p.kv_hs = kv_head_stride;
    p.out = (_Float16*)out; p.o_bs = out_batch_stride; p.o_ts = out_token_stride;
    p.out_lo = (_Float16*)out_lo;
    p.of_hi = (_Float16*)out_frag_hi; p.of_lo = (_Float16*)out_frag_lo;
    p.past_len_dev = past_len_dev;
    p.past_lens = past_lens;
    p.counters = counters;
    p.defer_merge = defer_merge; p.nsplit_out = nsplit_out;
    if (nsplit_out) *nsplit_out = 1;
    p.formal_handoff = pc_formal_handoff();
    p.pre_k = (const _Float16*)pre_k; p.pre_v = (const _Float16*)pre_v;
    p.pre_k_lo = (const _Float16*)pre_k_lo; p.pre_v_lo = (const _Float16*)pre_v_lo; p.pre_hs = pre_hs;
    p.trace = g_attn_trace;
    p.key_pos = key_pos; p.kp_bs = key_pos_batch_stride; p.slopes = slopes;
    p.k_lo = (const _Float16*)k_lo; p.v_lo = (const _Float16*)v_lo; p.lo_bs = lo_bs; p.lo_hs = lo_hs; p.lo_row0 = lo_row0;
    p.H = H; p.Hkv = Hkv; p.q_len = q_len; p.past_len = past_len;
    p.scale_log2 = softmax_scale * 1.4426950408889634f;
    p.nsplit = choose_nsplit(B, H, q_len, past_len + q_len, q_len <= kQB || q_lo != nullptr);
    p.tail = 0; p.small = 0;
    // (a pass the ring kernel takes carries its own rows as residual tiles in the stream: no fp32 tail workgroup)
    const bool ring_first = ring_eligible(p, D);
    p.tail = (!ring_first && k_lo && lo_row0 == -1 && q_len <= kTailMax && !past_lens) ? 1 : 0;
    // <= 16 rows over a long cache: one key slice per WAVE, partials merged per workgroup (attn_small_kernel).  Not for
    // launches that carry residual tiles in the stream (decode over a residual tail, lo_row0 != -1).
    static const bool small_off = [] { const char* e = getenv("PC_ATTN_NO_SMALL"); return e && e[0] == '1'; }();
    // 17..32 rows in tail mode take the same kernel with two row tiles per wave (every K fragment and V tile serves both;
    // two-launch form, no ALiBi instantiation); PC_ATTN_SMALL2=0: the 64-row kernel, as in rounds 1-3
    static const bool small2_off = [] { const char* e = getenv("PC_ATTN_SMALL2"); return e && e[0] == '0'; }();
    const bool small2 = !small2_off && q_len > kSmallQ && q_len <= 2 * kSmallQ && p.tail && !key_pos && !counters;
    bool small = !small_off && (q_len <= kSmallQ || small2) && !past_lens && (!k_lo || p.tail) && past_len + q_len >= 256;
    if (small) {
        const int ns = small_nstream(B, H, p.tail) + p.tail;
        if (ns >= 2) p.nsplit = ns; else small = false;
    }
    p.small = small ? 1 : 0;
    // staging while reading lives in attn_small_kernel's two-launch form and in the tail-mode instantiation of the 64-row kernel
    // (17..32 new rows: one q-block, the streaming splits cover staged keys only); one batch row
    const bool mid_gather = !small && p.tail && !ring_first && !key_pos && !pre_k && q_len <= kQB;
    // ... and in the ring kernel (more than 32 split-precision rows at head_dim 128, no shared prefix, one past length)
    const bool ring_gather = !small && ring_eligible(p, D) && !pre_k && !past_lens;
    const bool can_gather = (small || mid_gather || ring_gather) && (small ? !counters : true) && B == 1 && past_len > 0;
    if (gather_ok) { *gather_ok = can_gather ? 1 : 0; return PC_OK; }
    PC_REQUIRE(!gather_rows || can_gather, PC_ERR_ARG,
               "pc_attn: gather_rows needs B = 1 and a launch of <= %d query rows in tail mode, <= %d rows over >= 256 keys without counters, or a split-precision launch of >= %d rows at head_dim 128 (ask pc_attn_gather_ok)", kTailMax, kSmallQ, ring_min_rows());
    PC_REQUIRE(!gather_rows || (((uintptr_t)gather_rows & 15) == 0 && g_kplane >= 0 && g_vplane >= 0), PC_ERR_ARG,
               "pc_attn: gather_rows must be 16-byte aligned, planes non-negative");
    p.rows = gather_rows; p.g_kplane = g_kplane; p.g_vplane = g_vplane;
    if (ring_eligible(p, D)) p.nsplit = ring_nsplit(B, H, q_len, past_len + q_len);
    p.wide = 0; p.wide_nsplit = 0; p.own_only = 0;
    if (!small && wide_eligible(p, D, B)) { p.wide = 1; p.nsplit = wide_nsplit(H, q_len, past_len) + 1; }
    if (p.tail && !small) {
        p.nsplit = tail_stream_splits(B, H, past_len);
        // one more split for the tail workgroup -- taken from the streaming splits when the total would cross into the
        // next instantiation of the merge kernel (4 / 8 / 16 / 32 partials per row)
        int ms = p.nsplit;
        if (ms >= 4 && (ms & (ms - 1)) == 0) ms -= 1;
        p.nsplit = ms + 1;
    }
    p.part_o = nullptr; p.part_ml = nullptr;
    if (p.nsplit > 1) {
        const int64_t slots = (int64_t)B * H * p.nsplit * q_len;
        const int64_t need = slots * (D + 2) * (int64_t)sizeof(float);
        PC_REQUIRE(workspace && workspace_bytes >= need, PC_ERR_WORKSPACE,
                   "pc_attn_fwd: workspace %lld B < %lld B", (long long)workspace_bytes, (long long)need);
        p.part_o = (float*)workspace;
        p.part_ml = p.part_o + slots * D;
    }
    switch (D) {
        case 32: return launch_attn<32>(p, B, (hipStream_t)stream);
        case 64: return launch_attn<64>(p, B, (hipStream_t)stream);
        default: return launch_attn<128>(p, B, (hipStream_t)stream);
    }
}
}  // namespace

// pc_attn: the one struct-taking entry of the attention family (include/promptcache_hip.h; the round 1-2 entry points are inline
// wrappers over it until round 4).
// `counters` (optional, B*H zeroed uint32 words, left zero by every launch): launches of <= 16 query rows over a long staged
// cache then merge their split-KV partials INSIDE the launch (last-arriving workgroup per head) instead of through a second one.
PC_EXPORT int pc_attn(const pc_attn_args* a, void* stream) {
    PC_REQUIRE(a && a->struct_bytes == (uint32_t)sizeof(pc_attn_args), PC_ERR_ARG,
               "pc_attn: args is NULL or struct_bytes != sizeof(pc_attn_args) (ABI mismatch)");
    PC_REQUIRE((a->k_lo == nullptr) == (a->v_lo == nullptr) && (!a->k_lo || a->q_lo), PC_ERR_ARG,
               "pc_attn: k_lo / v_lo go together and need q_lo (split-precision Q)");
    PC_REQUIRE(!a->k_lo || a->lo_head_stride % 8 == 0, PC_ERR_ARG, "pc_attn: the lo strides must keep 16-byte alignment");
    PC_REQUIRE((a->key_pos == nullptr) == (a->slopes_log2 == nullptr), PC_ERR_ARG, "pc_attn: key_pos and slopes go together");
    PC_REQUIRE(!a->key_pos || (a->key_pos_batch_stride % 4 == 0 && ((uintptr_t)a->key_pos & 15) == 0), PC_ERR_ARG,
               "pc_attn: key_pos rows not 16-byte aligned");
    if (a->past_lens) {
        PC_REQUIRE(!a->past_len_dev && !a->key_pos && !a->out_frag_hi && (!a->k_lo || a->lo_row0 == 0), PC_ERR_ARG,
                   "pc_attn: per-row past lengths exclude past_len_dev, ALiBi and fragment output; residual planes must be arena-shaped");
    } else {
        PC_REQUIRE(!a->k_lo || a->lo_row0 == -1 || (a->lo_row0 == -2 && a->past_len_dev) ||
                   (a->lo_row0 >= 0 && a->lo_row0 <= a->past_len && !a->past_len_dev), PC_ERR_ARG,
                   "pc_attn: lo_row0 must be -1 (= past_len), -2 (= past_len_dev[1]) or lie in [0, past_len] of a host past_len");
    }
    PC_REQUIRE(!a->counters || ((uintptr_t)a->counters & 3) == 0, PC_ERR_ARG, "pc_attn: counters not 4-byte aligned");
    PC_REQUIRE(!a->defer_merge || (a->nsplit_out && !a->counters), PC_ERR_ARG, "pc_attn: defer_merge needs nsplit_out and excludes counters");
    if (a->prefix_k) {
        PC_REQUIRE(a->prefix_v && a->past_lens && a->prefix_head_stride % 8 == 0 && (a->q_lo || a->q_len > 16), PC_ERR_ARG,
                   "pc_attn: a shared prefix needs prefix_v, past_lens, a 16-byte aligned head stride and the many-row kernel");
        PC_REQUIRE((a->prefix_k_lo == nullptr) == (a->prefix_v_lo == nullptr) && (!a->prefix_k_lo || a->k_lo), PC_ERR_ARG,
                   "pc_attn: prefix_k_lo / prefix_v_lo go together and with k_lo / v_lo");
    }
    return attn_fwd_impl(a->q, a->q_lo, a->q_batch_stride, a->q_token_stride, a->k, a->v, a->kv_batch_stride, a->kv_head_stride,
                         a->out, a->out_batch_stride, a->out_token_stride, a->B, a->H, a->Hkv, a->D, a->q_len, a->past_len,
                         a->softmax_scale, a->workspace, a->workspace_bytes, a->past_len_dev, a->out_frag_hi, a->out_frag_lo,
                         a->key_pos, a->key_pos_batch_stride, a->slopes_log2, a->out_lo, a->k_lo, a->v_lo, a->lo_batch_stride,
                         a->lo_head_stride, a->lo_row0, stream, a->past_lens, a->counters, a->prefix_k, a->prefix_v,
                         a->prefix_k_lo, a->prefix_v_lo, a->prefix_head_stride, a->gather_rows, a->gather_k_plane, a->gather_v_plane, nullptr,
                         a->defer_merge, a->nsplit_out);
}

// Whether pc_attn would run this launch shape on a kernel that implements gather_rows (the field itself is ignored here).
PC_EXPORT int pc_attn_gather_ok(const pc_attn_args* a) {
    if (!a || a->struct_bytes != (uint32_t)sizeof(pc_attn_args)) return 0;
    if (a->past_lens || a->prefix_k || a->q_len <= 0) return 0;
    if (!(a->D == 32 || a->D == 64 || a->D == 128) || a->B <= 0 || a->H <= 0 || a->Hkv <= 0 || a->H % a->Hkv) return 0;
    int ok = 0;
    const int rc = attn_fwd_impl(a->q, a->q_lo, a->q_batch_stride, a->q_token_stride, a->k, a->v, a->kv_batch_stride, a->kv_head_stride,
                                 a->out, a->out_batch_stride, a->out_token_stride, a->B, a->H, a->Hkv, a->D, a->q_len, a->past_len,
                                 a->softmax_scale, a->workspace, a->workspace_bytes, a->past_len_dev, a->out_frag_hi, a->out_frag_lo,
                                 a->key_pos, a->key_pos_batch_stride, a->slopes_log2, a->out_lo, a->k_lo, a->v_lo, a->lo_batch_stride,
                                 a->lo_head_stride, a->lo_row0, nullptr, nullptr, a->counters, nullptr, nullptr, nullptr, nullptr, 0,
                                 nullptr, 0, 0, &ok);
    return rc == PC_OK ? ok : 0;
}
