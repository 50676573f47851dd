// (kernel templates of pc_gemm_q8: included by pc_gemm_q8.hip -- entry point, o_proj / down_proj shapes -- and pc_gemm_q8_norm.hip)
// LLM.int8 projections of a <= 16-row forward (the timed cached step, decode) with the ACTIVATION QUANTISER INSIDE the launch.
//
// Replaces, for load_in_8bit models (demo.py:27-29, eval.py:36-42 -> bitsandbytes Linear8bitLt, threshold 6.0), the pairs
//   pc_rmsnorm_quant_i8 + pc_gemm(q|k|v, a8c)      input_layernorm  + q/k/v_proj   llama2.py:629, :345-347
//   pc_quant_act_i8     + pc_gemm(o_proj, a8c)     o_proj + residual               llama2.py:405, :638
//   pc_rmsnorm_quant_i8 + pc_gemm(gate|up, a8c)    post_attention_layernorm + MLP  llama2.py:641, :242
//   pc_quant_act_i8     + pc_gemm(down, a8c)       down_proj + residual            llama2.py:242, :644
// Round 4 ran the vector-wise quantiser (Dettmers et al. 2022, section 3: outlier test, row absmax without the outliers, codes)
// as a launch of its own in front of every projection: four launches of 5.7 .. 8.9 us per layer (28 us of a 110 us layer:
// profiles/r05_int8_before.txt) whose arithmetic is a few thousand operations.  A row's scale needs the whole row, which is why
// it was a launch; here every workgroup of the consumer derives it itself:
//
//   P form (q|k|v, gate|up, o_proj; K <= 6144)  The workgroup reads the activation rows once -- the fp32 residual stream (the
//       RMSNorm folded in: sum of squares, gain, fp16 rounding) or the fp16 plane the attention wrote -- with rmsnorm_quant_kernel's
//       / quant_act_kernel's thread-to-chunk mapping, reduction order and arithmetic (256 threads per row, two rows per pass), so
//       codes, scales and outlier flags are bit-identical to the stand-alone quantisers'; the codes land in LDS as the int8
//       MFMA's operand image and the K loop reads its activation operands from there: no vector-memory request per k-step for
//       activations at all (those requests, not HBM, bound the round-4 launches).  The first block of weight fragments is in
//       flight while the prologue runs, and the K loop double-buffers the rest.
//   F form (down_proj; any K)  The producer's SiLU epilogue leaves per tile and row the largest non-outlier |value| and a flag byte
//       per outlier column (GemmParams::pmax_out); the consumer reduces 16 * inter/16 partial maxima (a maximum is exact in any
//       order) and quantises its fp16 fragments on the fly -- a lane's operand slots belong to its own row, so the scale is a
//       lane-local scalar.  K is split across workgroups with the reduction inside the launch (pc_gemm_ks.hip's hand-off):
//       each slice's partial tile is rescaled and carries its slice's share of the outlier correction, the last arriver adds the
//       slices in order, adds the residual stream and stores.
// The outlier correction (fp16 part of the decomposition) runs inside both forms exactly as in gemm_skinny_body (pc_gemm_skinny.h).
#pragma once
#include "pc_gemm_skinny.h"
#include "pc_part_merge.h"

namespace pcq {
using namespace pcg;

constexpr int kQ8RI = 8;                      // most prologue passes: two rows each (256 threads per row) -> 16 rows

struct Q8Params {
    GemmParams g;                             // wf (int8 image), wscale, cbt / ldt / row_perm, y / of_* / rope, M, ntiles, KS, npairs, kslices
    float threshold;
    const float* pmax_in; int32_t pmax_units; // F form: the producer's partial row maxima [units][16]
    const unsigned char* flags_in;            // F form: the producer's outlier-column flag bytes [>= 16384]
    unsigned char* flags_clear; int32_t clear_bytes;   // a flag buffer this launch zeroes (a multiple of 16 bytes)
    float* slabs; uint32_t* counters; int32_t formal;  // F form: in-launch K reduction (pc_gemm_ks.hip)
    signed char* dbg_codes; float* dbg_scale; unsigned char* dbg_flags;   // tests: workgroup 0's codes image / scales / flags
    // P form, source "partials" (M = 1): the split-KV partials pc_attn left (defer_merge) are merged in the prologue
    pcm::PartSrc part;
    // P form, source "image" (5..16 rows): a quantiser launch left the codes as the operand image, the row scales and the flag bytes
    const signed char* img8; const float* img_scale; const unsigned char* img_flags;
};
using pcm::PartLoads;

__device__ __forceinline__ int nz4(uint32_t w) {
    return ((w & 0xffu) ? 1 : 0) + ((w & 0xff00u) ? 1 : 0) + ((w & 0xff0000u) ? 1 : 0) + ((w >> 24) ? 1 : 0);
}

// eight values -> eight signed code bytes, quant_act_kernel's arithmetic: entries at or above the threshold count as zero,
// round_half_even(a * inv) (|a| <= the row maximum, so the product cannot pass 127; no clamp needed).  The integer goes under
// the mantissa of 1.5 * 2^23: the low byte of the sum's bit pattern is its two's complement.
__device__ __forceinline__ u32x2 quant8(h8 x, float inv, float thr) {
    uint32_t t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float f = (float)x[e];
        const float a = fabsf(f) >= thr ? 0.f : f;
        t[e] = __float_as_uint(rintf(a * inv) + 12582912.0f);
    }
    u32x2 r;
    r[0] = __builtin_amdgcn_perm(t[1], t[0], 0x0c0c0400u) | __builtin_amdgcn_perm(t[3], t[2], 0x04000c0cu);
    r[1] = __builtin_amdgcn_perm(t[5], t[4], 0x0c0c0400u) | __builtin_amdgcn_perm(t[7], t[6], 0x04000c0cu);
    return r;
}
__device__ __forceinline__ float code_of(float f, float inv, float thr) {
    const float a = fabsf(f) >= thr ? 0.f : f;
    return rintf(a * inv);
}

// byte offset of code (row, feature k) in the one-row-tile operand image [K/64][64 lanes][16 B]
__device__ __forceinline__ int img_off(int row, int k) {
    const int c = k >> 3;
    return (((c >> 3) * 64 + (c & 3) * 16 + row) << 4) + (((c >> 2) & 1) << 3) + (k & 7);
}

// (st_wt2 / ld_wt2, the write-through hand-off helpers: pc_gemm_skinny.h)

// ---- the outlier correction of one workgroup's tiles (gemm_skinny_body's, with the flags / operands behind accessors) ----
// corr[t][n] = sum over the flagged columns k of  X[t][k] * fp16(CB[n][k] * s[n])  -  CA[t][k] * CB[n][k] * xs[t] * s[n].
// `fw`: this thread's flag bytes (NF dwords = columns [tid * 4 NF, ...)), zero outside the columns the workgroup answers for.
// xraw(row, k) / code(row, k): the fp16 activation and its code.  cols: the (idle) reduction buffer; wtot: kWaves ints.
template <int TT, int NF, class XRaw, class Code>
__device__ __forceinline__ bool q8_correction(const GemmParams& p, const uint32_t (&fw)[NF], const int (&tile)[TT], bool row_ok, float xs_row,
                                              unsigned short* cols, int cols_cap, int* wtot, f4 (&cacc)[TT], XRaw xraw, Code code) {
    const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int mine = 0;
#pragma unroll
    for (int j = 0; j < NF; ++j) mine += nz4(fw[j]);
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    if (lane == 63) wtot[wave] = incl;
    lds_barrier();
    int off = incl - mine, total = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) { off += (w < wave) ? wtot[w] : 0; total += wtot[w]; }
    if (total == 0) return false;                        // (workgroup-uniform) the usual case behind a norm
    if (mine) {
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if ((fw[j] >> (8 * b)) & 0xffu) { if (off < cols_cap) cols[off] = (unsigned short)(tid * 4 * NF + 4 * j + b); ++off; }
    }
    lds_barrier();
    if (total > cols_cap) total = cols_cap;
    // MFMA form: 32 compacted columns are one k-step; see gemm_skinny_body
    float wsc[TT][4], wsa[TT];
    int nra[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) wsc[t][r] = p.wscale[tile[t] * 16 + g * 4 + r];
        const int na = tile[t] * 16 + m;
        wsa[t] = p.wscale[na];
        nra[t] = p.row_perm ? p.row_perm[na] : na;
    }
    const float xsr = row_ok ? xs_row : 0.f;
    f4 iacc[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) { f4 z = {0.f, 0.f, 0.f, 0.f}; iacc[t] = z; }
    const int nks = (total + 31) >> 5;
    // SETS column blocks per pass: every gather of both blocks is issued before the first MFMA (one memory round trip instead of one
    // per block -- the random-init bench model flags ~460 columns of down_proj's input: 15 blocks, two per wave); the blocks enter the
    // accumulators in the same order as one at a time
    constexpr int SETS = TT <= 4 ? 2 : 1;
    for (int s0 = wave; s0 < nks; s0 += SETS * kWaves) {
        int cj[SETS][8];
        bool okc[SETS][8];
        h8 xb[SETS], cb[SETS];
        signed char qb[SETS][TT][8];
#pragma unroll
        for (int z = 0; z < SETS; ++z) {
            const int sblk = s0 + z * kWaves;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = sblk * 32 + g * 8 + e;
                okc[z][e] = sblk < nks && j < total;
                cj[z][e] = cols[okc[z][e] ? j : 0];
            }
        }
#pragma unroll
        for (int z = 0; z < SETS; ++z) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool ok = okc[z][e] && row_ok;
                xb[z][e] = ok ? xraw(m, cj[z][e]) : (_Float16)0;
                cb[z][e] = ok ? code(m, cj[z][e]) : (_Float16)0;
            }
#pragma unroll
            for (int t = 0; t < TT; ++t)
#pragma unroll
                for (int e = 0; e < 8; ++e) qb[z][t][e] = okc[z][e] ? p.cbt[(int64_t)cj[z][e] * p.ldt + nra[t]] : (signed char)0;
        }
#pragma unroll
        for (int z = 0; z < SETS; ++z) {
            if (s0 + z * kWaves >= nks) continue;        // (wave-uniform)
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                h8 wa, qa;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float q = (float)qb[z][t][e];
                    qa[e] = (_Float16)q;
                    wa[e] = (_Float16)(q * wsa[t]);
                }
                cacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, xb[z], cacc[t], 0, 0, 0);
                iacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa, cb[z], iacc[t], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) cacc[t][r] -= iacc[t][r] * (xsr * wsc[t][r]);
    lds_barrier();                                       // the column list is dead: the buffer goes back to the reduction
    return true;
}

// =====================================================================================================================
// P form
// =====================================================================================================================
template <int TT> struct Q8Depth { static constexpr int NW = TT == 1 ? 8 : (TT == 2 ? 4 : (TT <= 6 ? 2 : 1)); };

// RI: prologue passes (two rows each): 2 for M <= 4 (decode; both K-loop buffers are then in flight across the prologue), 8 for M <= 16
// SRC: 0 the fp16 plane, 1 the fp32 residual stream under an RMSNorm, 2 pc_attn's split-KV partials of one row (M = 1),
// 3 the operand image + scales + flags of a quantiser launch (pc_quant_act_i8 / pc_rmsnorm_quant_i8 codes8): copied into LDS, no
// quantiser arithmetic here -- the K loop of this file (activation operands from LDS, two blocks in flight) for 5..16 rows
template <int T, int EPI, int SRC, int G, int RI>
__global__ __launch_bounds__(kThreads) void gemm_q8p_kernel(const Q8Params qp) {
    const GemmParams& p = qp.g;
    constexpr bool NORM = SRC == 1, PART = SRC == 2, IMG = SRC == 3;
    constexpr int GP = (G + 1) / 2;                      // PART: chunks per thread (all 512 threads on the one row)
    static_assert(!PART || (RI == 2 && GP == 1), "the partials source is for one row of K <= 4096");
    constexpr int TT = (EPI == EPI_SILU) ? 2 * T : T;
    constexpr int TPI = (EPI == EPI_SILU) ? 2 : 1;
    constexpr int kRT = (TT < 8) ? TT : 8;
    constexpr int IPR = kRT / TPI;
    constexpr int NW = Q8Depth<TT>::NW;                  // k-step pairs per block of the K loop
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // [K * 16] operand image | [G * 2048] flag bytes
    __shared__ __attribute__((aligned(16))) float red_raw[kWaves * kRT * 64 * 4];
    __shared__ float lred[kWaves][kQ8RI], lredq[kWaves][kQ8RI];
    constexpr bool PF2 = RI <= 2;                        // both buffers requested before the prologue
    __shared__ float lrs[16], lxs[16];
    __shared__ int wtot[kWaves];
    float (*red)[kRT][64][4] = (float (*)[kRT][64][4])red_raw;

    const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bx = blockIdx.x, KS = p.KS, K = KS * 32, nv = K >> 3, M = p.M;
    unsigned char* img = smem;
    unsigned char* lflag = smem + (size_t)K * 16;
    const float thr = qp.threshold > 0.f ? qp.threshold : __builtin_inff();

    // ---- 1. every load that does not wait for anything: the activation rows, the gain, then the first weight block ----
    const int hsel = __builtin_amdgcn_readfirstlane(tid >> 8), t8 = tid & 255;
    // (KEEP: the fp32 rows stay in registers between the two norm passes; at three chunks per thread -- K > 4096 -- sixteen rows of
    // them do not fit next to the weight prefetch, and the second pass reads them again: L2 hits)
    constexpr bool KEEP = NORM && RI * G <= 16;
    h8 hv[RI][G];
    [[maybe_unused]] f4 va[KEEP ? RI : 1][G], vb[KEEP ? RI : 1][G];
    [[maybe_unused]] h8 gw[G];
    auto load_x = [&](int row, f4 (&a)[G], f4 (&b)[G]) {
#pragma unroll
        for (int k = 0; k < G; ++k) {
            const int i = t8 + k * 256;
            f4 z = {0.f, 0.f, 0.f, 0.f};
            a[k] = z; b[k] = z;
            if (row < M && i < nv) {
                const float* xr = p.xn + (int64_t)row * K + i * 8;
                a[k] = *(const f4*)xr;
                b[k] = *(const f4*)(xr + 4);
            }
        }
    };
#pragma unroll
    for (int it = 0; it < RI; ++it) {
        const int row = 2 * it + hsel;
        if constexpr (KEEP) load_x(row, va[it], vb[it]);
#pragma unroll
        for (int k = 0; k < G; ++k) {
            const int i = t8 + k * 256;
            if constexpr (NORM) {
                h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                hv[it][k] = z;
            } else {
                h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                hv[it][k] = z;
                if constexpr (!PART && !IMG) {
                    if (row < M && i < nv) hv[it][k] = *(const h8*)(p.xf_hi + frag_off(row, i * 8, KS));
                }
            }
        }
    }
    [[maybe_unused]] PartLoads pl;
    if constexpr (PART) pcm::part_issue(qp.part, 0, (tid < nv ? tid : nv - 1) * 8, pl);
    [[maybe_unused]] u32x4 ci[IMG ? 4 * G : 1];          // IMG: this thread's 16-byte entries of the image (K entries), its flag dwords
    [[maybe_unused]] uint32_t cf[IMG ? G : 1];
    if constexpr (IMG) {
#pragma unroll
        for (int j = 0; j < 4 * G; ++j) {
            const int e = tid + j * kThreads;
            ci[j] = ((const u32x4*)qp.img8)[e < K ? e : K - 1];
        }
#pragma unroll
        for (int j = 0; j < G; ++j) cf[j] = ((const uint32_t*)qp.img_flags)[tid * G + j];
    }
    if constexpr (NORM) {
#pragma unroll
        for (int k = 0; k < G; ++k) {
            const int i = t8 + k * 256;
            h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            gw[k] = z;
            if (i < nv) gw[k] = *(const h8*)(p.gamma + i * 8);
        }
    }
    int tile[TT];
    wg_tiles<T, EPI>(p, bx, tile);
    int ks0, ks1;
    wave_k_range<true>(p, 0, wave, ks0, ks1);
    const int p0 = ks0 >> 1, p1 = ks1 >> 1;              // this wave's k-step pairs
    // (addresses as a wave-uniform tile base + one 32-bit lane offset per k-step pair: the loads take the scalar base and the
    // offset register directly -- a 64-bit lane address per load had the allocator park addresses in registers whose loads were
    // still in flight, which serialised the blocks)
    const char* wt[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) wt[t] = (const char*)p.wf + (int64_t)tile[t] * (KS >> 1) * 1024;
    u32x4 rawA[NW][TT], rawB[NW][TT];
    auto issue = [&](u32x4 (&raw)[NW][TT], int pb) {
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            int pr = pb + u < p1 ? pb + u : p1 - 1;
            pr = pr < 0 ? 0 : pr;
            const uint32_t voff = (uint32_t)pr * 1024u + (uint32_t)lane * 16u;
#pragma unroll
            for (int t = 0; t < TT; ++t) raw[u][t] = __builtin_nontemporal_load((const u32x4*)(wt[t] + voff));
        }
    };
    issue(rawA, p0);
    if (PF2 && p0 + NW < p1) issue(rawB, p0 + NW);
    // EPI_ADD: the residual tile wave w will add to (written by this lane only)
    constexpr bool kPreY = (EPI == EPI_ADD) && (T <= IPR);
    [[maybe_unused]] f4 yold = {0.f, 0.f, 0.f, 0.f};
    if constexpr (kPreY) {
        const int t = wave < T ? wave : T - 1;
        const int unit = bx * T + t < p.ntiles ? bx * T + t : p.ntiles - 1;
        yold = *(const f4*)(p.y + (int64_t)(m < M ? m : M - 1) * p.ldy + unit * 16 + g * 4);
    }
    // the flag bytes start at zero; a flag buffer of the NEXT producer is cleared by the whole grid on the side
    if constexpr (IMG) {
#pragma unroll
        for (int j = 0; j < 4 * G; ++j) {
            const int e = tid + j * kThreads;
            if (e < K) ((u32x4*)img)[e] = ci[j];
        }
#pragma unroll
        for (int j = 0; j < G; ++j) ((uint32_t*)lflag)[tid * G + j] = cf[j];
        if (tid < M) lxs[tid] = qp.img_scale[tid];
    } else {
        for (int i = tid; i < G * 512; i += kThreads) ((uint32_t*)lflag)[i] = 0u;
    }
    if (qp.flags_clear) {
        const u32x4 z4 = {0u, 0u, 0u, 0u};
        for (int i = bx * kThreads + tid; i < (qp.clear_bytes >> 4); i += gridDim.x * kThreads) ((u32x4*)qp.flags_clear)[i] = z4;
    }

    // ---- 2. RMSNorm (rmsnorm_quant_kernel's order: per-thread partial sums, wave shuffles, the row's four waves in order) ----
    if constexpr (NORM) {
#pragma unroll
        for (int it = 0; it < RI; ++it) {
            float ss = 0.f;
            f4 ta[G], tb[G];
            if constexpr (KEEP) {
#pragma unroll
                for (int k = 0; k < G; ++k) { ta[k] = va[it][k]; tb[k] = vb[it][k]; }
            } else {
                load_x(2 * it + hsel, ta, tb);
            }
#pragma unroll
            for (int k = 0; k < G; ++k)
                ss += ta[k][0] * ta[k][0] + ta[k][1] * ta[k][1] + ta[k][2] * ta[k][2] + ta[k][3] * ta[k][3] +
                      tb[k][0] * tb[k][0] + tb[k][1] * tb[k][1] + tb[k][2] * tb[k][2] + tb[k][3] * tb[k][3];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
            if (lane == 0) lred[wave][it] = ss;
        }
    }
    lds_barrier();
    if constexpr (NORM) {
#pragma unroll
        for (int it = 0; it < RI; ++it) {
            const int row = 2 * it + hsel;
            if (row < M) {                               // (wave-uniform)
                const int w0 = 4 * hsel;
                const float rs = rsqrtf((lred[w0][it] + lred[w0 + 1][it] + lred[w0 + 2][it] + lred[w0 + 3][it]) / (float)K + p.eps);
                if (t8 == 0) lrs[row] = rs;
                f4 ta[G], tb[G];
                if constexpr (KEEP) {
#pragma unroll
                    for (int k = 0; k < G; ++k) { ta[k] = va[it][k]; tb[k] = vb[it][k]; }
                } else {
                    load_x(row, ta, tb);
                }
#pragma unroll
                for (int k = 0; k < G; ++k)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float v = (float)gw[k][e] * ((e < 4 ? ta[k][e] : tb[k][e - 4]) * rs);
                        _Float16 vh, vl;
                        pc_split(v, vh, vl);
                        hv[it][k][e] = vh;
                    }
            }
        }
    }
    if constexpr (IMG) {
        // (image, flags and scales are in LDS behind the barrier above)
    } else if constexpr (PART) {
        // ---- 3'. / 4'. one row, one chunk per thread: merge, flags, the row maximum over the eight waves, codes ----
        const h8 hp = __builtin_bit_cast(h8, pcm::part_merge(qp.part, pl));
        float mx = 0.f;
        if (tid < nv) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = fabsf((float)hp[e]);
                if (f >= thr) lflag[tid * 8 + e] = 1;
                else mx = fmaxf(mx, f);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if (lane == 0) lredq[wave][0] = mx;
        lds_barrier();
        float sca = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) sca = fmaxf(sca, lredq[w][0]);
        const float inv = sca > 0.f ? 127.0f / sca : 0.f;
        if (tid == 0) lxs[0] = sca / 127.0f;
        if (tid < nv) *(u32x2*)(img + img_off(0, tid * 8)) = quant8(hp, inv, thr);
        lds_barrier();
    } else {
    // ---- 3. outlier flags + row maxima without the outliers (quant_act_kernel) ----
#pragma unroll
    for (int it = 0; it < RI; ++it) {
        const int row = 2 * it + hsel;
        float mx = 0.f;
        if (row < M) {
#pragma unroll
            for (int k = 0; k < G; ++k) {
                const int i = t8 + k * 256;
                if (i < nv) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float f = fabsf((float)hv[it][k][e]);
                        if (f >= thr) lflag[i * 8 + e] = 1;
                        else mx = fmaxf(mx, f);
                    }
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if (lane == 0) lredq[wave][it] = mx;
    }
    lds_barrier();
    // ---- 4. codes into the operand image ----
#pragma unroll
    for (int it = 0; it < RI; ++it) {
        const int row = 2 * it + hsel;
        if (row < M) {
            const int w0 = 4 * hsel;
            const float sca = fmaxf(fmaxf(lredq[w0][it], lredq[w0 + 1][it]), fmaxf(lredq[w0 + 2][it], lredq[w0 + 3][it]));
            const float inv = sca > 0.f ? 127.0f / sca : 0.f;
            if (t8 == 0) lxs[row] = sca / 127.0f;
#pragma unroll
            for (int k = 0; k < G; ++k) {
                const int i = t8 + k * 256;
                if (i < nv) {
                    const u32x2 q = quant8(hv[it][k], inv, thr);
                    *(u32x2*)(img + img_off(row, i * 8)) = q;
                }
            }
        }
    }
    lds_barrier();
    }
    if (qp.dbg_codes && bx == 0) {                       // tests: what this workgroup computed
        for (int i = tid; i < K * 4; i += kThreads) ((uint32_t*)qp.dbg_codes)[i] = ((const uint32_t*)img)[i];
        for (int i = tid; i < K; i += kThreads) qp.dbg_flags[i] = lflag[i];
        if (tid < M) qp.dbg_scale[tid] = lxs[tid];
    }

    // ---- 5. the K loop: weight fragments streamed (double-buffered blocks of NW pairs), activation operands from LDS ----
    const bool row_ok = m < M;
    i32x4 acc[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) acc[t] = i32x4{0, 0, 0, 0};
    const u32x4* xi = (const u32x4*)img + lane;
    auto consume = [&](const u32x4 (&raw)[NW][TT], int pb) {
        u32x4 xq[NW];
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            xq[u] = u32x4{0u, 0u, 0u, 0u};
            if (pb + u < p1) xq[u] = xi[(pb + u) * 64];   // (lanes of rows behind M read stale LDS: their output columns are never stored)
        }
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            if (pb + u >= p1) continue;                  // (wave-uniform)
            const i32x4 xa = {(int)xq[u][0], (int)xq[u][1], (int)xq[u][2], (int)xq[u][3]};
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                const u32x4 wq = raw[u][t] ^ 0x80808080u;
                const i32x4 wv = {(int)wq[0], (int)wq[1], (int)wq[2], (int)wq[3]};
                acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wv, xa, acc[t], 0, 0, 0);
            }
        }
    };
    // Two blocks in flight: block b + 1 was requested before block b is consumed, block b + 2 is requested into b's registers right
    // after.  Every path through the loop body issues and consumes in the same order, so the compiler's vmcnt counts are exact (a
    // conditional issue made it wait for the block it had just requested).
    {
        const int nb = (p1 - p0 + NW - 1) / NW;
        if (!PF2 && nb > 1) issue(rawB, p0 + NW);
        int b = 0;
        bool done = false;
        while (b + 2 < nb) {
            consume(rawA, p0 + b * NW);
            issue(rawA, p0 + (b + 2) * NW);
            if (!(b + 3 < nb)) {
                consume(rawB, p0 + (b + 1) * NW);
                consume(rawA, p0 + (b + 2) * NW);
                done = true;
                break;
            }
            consume(rawB, p0 + (b + 1) * NW);
            issue(rawB, p0 + (b + 3) * NW);
            b += 2;
        }
        if (!done) {
            if (b < nb) consume(rawA, p0 + b * NW);
            if (b + 1 < nb) consume(rawB, p0 + (b + 1) * NW);
        }
    }
    f4 facc[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) facc[t] = f4{(float)acc[t][0], (float)acc[t][1], (float)acc[t][2], (float)acc[t][3]};

    // ---- 6. the outlier correction, dealt over the eight waves ----
    f4 cacc[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) { f4 z = {0.f, 0.f, 0.f, 0.f}; cacc[t] = z; }
    uint32_t fw[G];
#pragma unroll
    for (int j = 0; j < G; ++j) fw[j] = ((const uint32_t*)lflag)[tid * G + j];
    const float xs_row = row_ok ? lxs[m] : 0.f;
    const float rs_row = (NORM && row_ok) ? lrs[m] : 0.f;
    auto xraw = [&](int row, int k) -> _Float16 {
        if constexpr (NORM) {
            const float v = (float)p.gamma[k] * (p.xn[(int64_t)row * K + k] * rs_row);
            _Float16 vh, vl;
            pc_split(v, vh, vl);
            return vh;
        } else if constexpr (PART) {
            PartLoads L;
            pcm::part_issue(qp.part, 0, k & ~7, L);
            return pcm::part_merge(qp.part, L)[k & 7];
        } else {
            return p.xf_hi[frag_off(row, k, KS)];
        }
    };
    auto code = [&](int row, int k) -> _Float16 { return (_Float16)(float)((const signed char*)img)[img_off(row, k)]; };
    const bool fused = q8_correction<TT, G>(p, fw, tile, row_ok, xs_row, (unsigned short*)red_raw, kWaves * kRT * 64 * 4 * 4 / 2, wtot,
                                            cacc, xraw, code);

    // ---- 7. split-K reduction through LDS (fixed order) + epilogue: gemm_skinny_body's rounds ----
    constexpr int TE = (EPI == EPI_SILU) ? T : TT;
    constexpr int ROUNDS = (TE + IPR - 1) / IPR;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        if (r > 0) lds_barrier();
        f4 csv = {0.f, 0.f, 0.f, 0.f}, csu = {0.f, 0.f, 0.f, 0.f};
        if (fused) {
#pragma unroll
            for (int i = 0; i < IPR; ++i) {
                const int t = r * IPR + i;
                if (t < TE) {
                    *(f4*)red[wave][i * TPI][lane] = cacc[t];
                    if (EPI == EPI_SILU) *(f4*)red[wave][i * TPI + 1][lane] = cacc[T + t];
                }
            }
            lds_barrier();
            if (wave < IPR && r * IPR + wave < TE) {
#pragma unroll
                for (int w = 0; w < kWaves; ++w) {
                    const f4 x = *(const f4*)red[w][wave * TPI][lane];
                    csv[0] += x[0]; csv[1] += x[1]; csv[2] += x[2]; csv[3] += x[3];
                    if (EPI == EPI_SILU) {
                        const f4 y = *(const f4*)red[w][wave * TPI + 1][lane];
                        csu[0] += y[0]; csu[1] += y[1]; csu[2] += y[2]; csu[3] += y[3];
                    }
                }
            }
            lds_barrier();
        }
#pragma unroll
        for (int i = 0; i < IPR; ++i) {
            const int t = r * IPR + i;
            if (t < TE) {
                *(f4*)red[wave][i * TPI][lane] = facc[t];
                if (EPI == EPI_SILU) *(f4*)red[wave][i * TPI + 1][lane] = facc[T + t];
            }
        }
        lds_barrier();
        const int t = r * IPR + wave;
        if (wave < IPR && t < TE) {
            f4 v = {0.f, 0.f, 0.f, 0.f}, u = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < kWaves; ++w) {
                const f4 x = *(const f4*)red[w][wave * TPI][lane];
                v[0] += x[0]; v[1] += x[1]; v[2] += x[2]; v[3] += x[3];
                if (EPI == EPI_SILU) {
                    const f4 y = *(const f4*)red[w][wave * TPI + 1][lane];
                    u[0] += y[0]; u[1] += y[1]; u[2] += y[2]; u[3] += y[3];
                }
            }
            tile_epilogue<EPI>(p, v, u, m, bx * T + t, g, 0, fused, csv, csu, kPreY, yold, true, xs_row);
        }
    }
}

// =====================================================================================================================
// F form
// =====================================================================================================================
constexpr int kMaxSlices = 8;

template <int T, int NW>
__global__ __launch_bounds__(kThreads) void gemm_q8f_kernel(const Q8Params qp) {
    const GemmParams& p = qp.g;
    __shared__ __attribute__((aligned(16))) float red_raw[kWaves * T * 64 * 4];
    __shared__ float lpm[kWaves][16];
    __shared__ int wtot[kWaves];
    __shared__ int s_last;
    float (*red)[T][64][4] = (float (*)[T][64][4])red_raw;
    const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bx = blockIdx.x, by = blockIdx.y, KS = p.KS, S = p.kslices, M = p.M;
    const float thr = qp.threshold > 0.f ? qp.threshold : __builtin_inff();
    const bool row_ok = m < M;

    // ---- 1. the producer's partial row maxima (thread: rows 4 rq .. 4 rq + 3 of units part, part + 128, ...) ----
    const int rq = tid & 3, part = tid >> 2;
    f4 pm = {0.f, 0.f, 0.f, 0.f};
    for (int u = part; u < qp.pmax_units; u += kThreads / 4) {
        const f4 x = *(const f4*)(qp.pmax_in + (int64_t)u * 16 + rq * 4);
        pm[0] = fmaxf(pm[0], x[0]); pm[1] = fmaxf(pm[1], x[1]); pm[2] = fmaxf(pm[2], x[2]); pm[3] = fmaxf(pm[3], x[3]);
    }
    // ---- 2. the first block's operands ----
    int ks0, ks1;
    wave_k_range<true>(p, by, wave, ks0, ks1);
    const int p0 = ks0 >> 1, p1 = ks1 >> 1;
    int tile[T];
    wg_tiles<T, EPI_ADD>(p, bx, tile);
    const char* wt[T];
#pragma unroll
    for (int t = 0; t < T; ++t) wt[t] = (const char*)p.wf + (int64_t)tile[t] * (KS >> 1) * 1024;
    const char* xbase = (const char*)p.xf_hi;
    u32x4 rawA[NW][T], rawB[NW][T];
    h8 xA[NW][2], xB[NW][2];
    auto issue = [&](u32x4 (&raw)[NW][T], h8 (&xh)[NW][2], int pb) {
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            int pr = pb + u < p1 ? pb + u : p1 - 1;
            pr = pr < 0 ? 0 : pr;
            const uint32_t voff = (uint32_t)pr * 1024u + (uint32_t)lane * 16u;
#pragma unroll
            for (int t = 0; t < T; ++t) raw[u][t] = __builtin_nontemporal_load((const u32x4*)(wt[t] + voff));
            h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            xh[u][0] = z; xh[u][1] = z;
            if (row_ok) {                                // (one fp16 k-step of the plane = 1 KiB, like a weight pair)
                xh[u][0] = *(const h8*)(xbase + 2u * voff - (uint32_t)lane * 16u);
                xh[u][1] = *(const h8*)(xbase + 2u * voff - (uint32_t)lane * 16u + 1024u);
            }
        }
    };
    issue(rawA, xA, p0);
    if (p0 + NW < p1) issue(rawB, xB, p0 + NW);
    f4 yold;
    {
        const int tw = wave < T ? wave : T - 1;
        const int unit = bx * T + tw < p.ntiles ? bx * T + tw : p.ntiles - 1;
        yold = *(const f4*)(p.y + (int64_t)(m < M ? m : M - 1) * p.ldy + unit * 16 + g * 4);
    }
    if (qp.flags_clear) {                                // (a flag buffer of a LATER launch, zeroed on the side by the whole grid)
        const u32x4 z4 = {0u, 0u, 0u, 0u};
        for (int i = (by * gridDim.x + bx) * kThreads + tid; i < (qp.clear_bytes >> 4); i += gridDim.x * gridDim.y * kThreads)
            ((u32x4*)qp.flags_clear)[i] = z4;
    }
    // ---- 3. row scales: a maximum is exact in any order ----
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) {
        pm[0] = fmaxf(pm[0], __shfl_xor(pm[0], o)); pm[1] = fmaxf(pm[1], __shfl_xor(pm[1], o));
        pm[2] = fmaxf(pm[2], __shfl_xor(pm[2], o)); pm[3] = fmaxf(pm[3], __shfl_xor(pm[3], o));
    }
    if (lane < 4) *(f4*)&lpm[wave][lane * 4] = pm;
    lds_barrier();
    float sca = 0.f;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) sca = fmaxf(sca, lpm[w][m]);
    const float inv = (row_ok && sca > 0.f) ? 127.0f / sca : 0.f;
    const float xs_row = row_ok ? sca / 127.0f : 0.f;
    if (qp.dbg_scale && bx == 0 && by == 0 && tid < M) qp.dbg_scale[tid] = xs_row;

    // ---- 4. the K loop: fp16 fragments quantised in registers (a lane's operand slots are its own row's) ----
    i32x4 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = i32x4{0, 0, 0, 0};
    auto consume = [&](const u32x4 (&raw)[NW][T], const h8 (&xh)[NW][2], int pb) {
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            if (pb + u >= p1) continue;                  // (wave-uniform)
            const u32x2 lo = quant8(xh[u][0], inv, thr), hi = quant8(xh[u][1], inv, thr);
            const i32x4 xa = {(int)lo[0], (int)lo[1], (int)hi[0], (int)hi[1]};
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const u32x4 wq = raw[u][t] ^ 0x80808080u;
                const i32x4 wv = {(int)wq[0], (int)wq[1], (int)wq[2], (int)wq[3]};
                acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wv, xa, acc[t], 0, 0, 0);
            }
        }
    };
    {   // (two blocks in flight, as in the P form)
        const int nb = (p1 - p0 + NW - 1) / NW;
        int b = 0;
        bool done = false;
        while (b + 2 < nb) {
            consume(rawA, xA, p0 + b * NW);
            issue(rawA, xA, p0 + (b + 2) * NW);
            if (!(b + 3 < nb)) {
                consume(rawB, xB, p0 + (b + 1) * NW);
                consume(rawA, xA, p0 + (b + 2) * NW);
                done = true;
                break;
            }
            consume(rawB, xB, p0 + (b + 1) * NW);
            issue(rawB, xB, p0 + (b + 3) * NW);
            b += 2;
        }
        if (!done) {
            if (b < nb) consume(rawA, xA, p0 + b * NW);
            if (b + 1 < nb) consume(rawB, xB, p0 + (b + 1) * NW);
        }
    }
    f4 facc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) facc[t] = f4{(float)acc[t][0], (float)acc[t][1], (float)acc[t][2], (float)acc[t][3]};

    // ---- 5. this slice's share of the outlier correction: the flagged columns inside the slice's K range ----
    f4 cacc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) { f4 z = {0.f, 0.f, 0.f, 0.f}; cacc[t] = z; }
    bool fused = false;
    if (qp.flags_in) {
        int ksq = (KS + S - 1) / S;
        ksq = (ksq + 1) & ~1;
        const int kq0 = by * ksq, kq1 = (kq0 + ksq < KS) ? kq0 + ksq : KS;
        const int c0 = kq0 * 32, c1 = kq1 * 32;
        uint32_t fw[8];
        const u32x4 f0 = *(const u32x4*)(qp.flags_in + tid * 32), f1 = *(const u32x4*)(qp.flags_in + tid * 32 + 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) { fw[j] = f0[j]; fw[4 + j] = f1[j]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) {                    // (slice boundaries are multiples of 64 columns)
            const int c = tid * 32 + 4 * j;
            if (c < c0 || c >= c1) fw[j] = 0u;
        }
        auto xraw = [&](int row, int k) -> _Float16 { return p.xf_hi[frag_off(row, k, KS)]; };
        auto code = [&](int row, int k) -> _Float16 { return (_Float16)code_of((float)p.xf_hi[frag_off(row, k, KS)], inv, thr); };
        fused = q8_correction<T, 8>(p, fw, tile, row_ok, xs_row, (unsigned short*)red_raw, kWaves * T * 64 * 4 * 4 / 2, wtot, cacc, xraw, code);
    }

    // ---- 6. the eight waves' shares through LDS, fixed order; wave t then holds the slice's partial of tile t, rescaled ----
    f4 cs = {0.f, 0.f, 0.f, 0.f};
    if (fused) {
#pragma unroll
        for (int t = 0; t < T; ++t) *(f4*)red[wave][t][lane] = cacc[t];
        lds_barrier();
        if (wave < T) {
#pragma unroll
            for (int w = 0; w < kWaves; ++w) {
                const f4 x = *(const f4*)red[w][wave][lane];
                cs[0] += x[0]; cs[1] += x[1]; cs[2] += x[2]; cs[3] += x[3];
            }
        }
        lds_barrier();
    }
#pragma unroll
    for (int t = 0; t < T; ++t) *(f4*)red[wave][t][lane] = facc[t];
    lds_barrier();
    f4 v = {0.f, 0.f, 0.f, 0.f};
    const bool mine = wave < T && bx * T + wave < p.ntiles;
    const int my_tile = bx * T + wave;
    if (wave < T) {
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            const f4 x = *(const f4*)red[w][wave][lane];
            v[0] += x[0]; v[1] += x[1]; v[2] += x[2]; v[3] += x[3];
        }
    }
    if (mine) {
        const f4 sv = *(const f4*)(p.wscale + my_tile * 16 + g * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (v[r] * sv[r]) * xs_row + cs[r];
        float* dst = qp.slabs + (((int64_t)by * p.ntiles + my_tile) * 64 + lane) * 4;
        st_wt2(dst, v[0], v[1]);
        st_wt2(dst + 2, v[2], v[3]);
    }
    // hand-off: pc_gemm_ks.hip's contract (agent-scope write-through stores, vmcnt(0), barrier, one arrival per workgroup)
#if !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__) && defined(__HIP_DEVICE_COMPILE__)
#error "gemm_q8f_kernel's in-launch hand-off relies on gfx9 vmcnt semantics (stores counted in vmcnt); re-derive it for this target"
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        gu32* c = (gu32*)(qp.counters + bx);
        const uint32_t old = qp.formal ? __hip_atomic_fetch_add(c, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT)
                                       : __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (old + 1u == (uint32_t)S) ? 1 : 0;
        if (last) __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = last;
    }
    __syncthreads();
    if (!s_last || !mine) return;
    float2 a[kMaxSlices], b[kMaxSlices];
#pragma unroll
    for (int s = 0; s < kMaxSlices; ++s) {
        const int sc = s < S ? s : S - 1;
        const float* src = qp.slabs + (((int64_t)sc * p.ntiles + my_tile) * 64 + lane) * 4;
        a[s] = ld_wt2(src);
        b[s] = ld_wt2(src + 2);
    }
    f4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < kMaxSlices; ++s)
        if (s < S) { r[0] += a[s].x; r[1] += a[s].y; r[2] += b[s].x; r[3] += b[s].y; }
    if (row_ok) {
        float* yp = p.y + (int64_t)m * p.ldy + my_tile * 16 + g * 4;
        r[0] += yold[0]; r[1] += yold[1]; r[2] += yold[2]; r[3] += yold[3];
        *(f4*)yp = r;
    }
}

// =====================================================================================================================
// C form: the F form's inputs (fp16 plane + the producer's per-tile row maxima and flag bytes) at M <= 4 rows (decode).  Four
// rows of codes fit the LDS for any K <= 16384 as a COMPACT operand image [K/64][4 g][4 rows][16 B] (lanes of rows m >= 4 read
// row m & 3: their output columns are never stored), so the workgroup quantises its rows once, up front, and keeps all of K: no
// K slices, no hand-off through memory behind the K loop (the F form's tail: write-through partials, arrival, read-back ~3 us),
// no quantiser arithmetic inside the K loop.
// =====================================================================================================================
template <int T, int CI>
__global__ __launch_bounds__(kThreads) void gemm_q8c_kernel(const Q8Params qp) {
    const GemmParams& p = qp.g;
    constexpr int NW = Q8Depth<T>::NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // [K * 4] compact operand image
    __shared__ __attribute__((aligned(16))) float red_raw[kWaves * T * 64 * 4];
    __shared__ __attribute__((aligned(16))) float lpm[kWaves][4];
    __shared__ int wtot[kWaves];
    float (*red)[T][64][4] = (float (*)[T][64][4])red_raw;
    const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bx = blockIdx.x, KS = p.KS, K = KS * 32, nv = K >> 3, M = p.M;
    const float thr = qp.threshold > 0.f ? qp.threshold : __builtin_inff();
    const bool row_ok = m < M;
    unsigned char* img = smem;

    // ---- 1. loads that wait for nothing, in the order they are needed: row maxima, this thread's activation chunks, weights ----
    f4 pm = {0.f, 0.f, 0.f, 0.f};
    for (int u = tid; u < qp.pmax_units; u += kThreads) {
        const f4 x = *(const f4*)(qp.pmax_in + (int64_t)u * 16);             // rows 0 .. 3 of unit u
        pm[0] = fmaxf(pm[0], x[0]); pm[1] = fmaxf(pm[1], x[1]); pm[2] = fmaxf(pm[2], x[2]); pm[3] = fmaxf(pm[3], x[3]);
    }
    h8 hv[CI];
    const int nwork = M * nv;                             // (row, chunk) items; item w = tid + j * 512: row = w / nv
#pragma unroll
    for (int j = 0; j < CI; ++j) {
        const int w = tid + j * kThreads;
        h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        hv[j] = z;
        if (w < nwork) {
            const int row = (w >= nv) + (w >= 2 * nv) + (w >= 3 * nv), i = w - row * nv;
            hv[j] = *(const h8*)(p.xf_hi + frag_off(row, i * 8, KS));
        }
    }
    int tile[T];
    wg_tiles<T, EPI_ADD>(p, bx, tile);
    int ks0, ks1;
    wave_k_range<true>(p, 0, wave, ks0, ks1);
    const int p0 = ks0 >> 1, p1 = ks1 >> 1;
    const char* wt[T];
#pragma unroll
    for (int t = 0; t < T; ++t) wt[t] = (const char*)p.wf + (int64_t)tile[t] * (KS >> 1) * 1024;
    u32x4 rawA[NW][T], rawB[NW][T];
    auto issue = [&](u32x4 (&raw)[NW][T], int pb) {
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            int pr = pb + u < p1 ? pb + u : p1 - 1;
            pr = pr < 0 ? 0 : pr;
            const uint32_t voff = (uint32_t)pr * 1024u + (uint32_t)lane * 16u;
#pragma unroll
            for (int t = 0; t < T; ++t) raw[u][t] = __builtin_nontemporal_load((const u32x4*)(wt[t] + voff));
        }
    };
    issue(rawA, p0);
    if (p0 + NW < p1) issue(rawB, p0 + NW);
    f4 yold;
    {
        const int t = wave < T ? wave : T - 1;
        const int unit = bx * T + t < p.ntiles ? bx * T + t : p.ntiles - 1;
        yold = *(const f4*)(p.y + (int64_t)(m < M ? m : M - 1) * p.ldy + unit * 16 + g * 4);
    }
    if (qp.flags_clear) {
        const u32x4 z4 = {0u, 0u, 0u, 0u};
        for (int i = bx * kThreads + tid; i < (qp.clear_bytes >> 4); i += gridDim.x * kThreads) ((u32x4*)qp.flags_clear)[i] = z4;
    }
    // ---- 2. row scales (a maximum is exact in any order) ----
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        pm[0] = fmaxf(pm[0], __shfl_xor(pm[0], o)); pm[1] = fmaxf(pm[1], __shfl_xor(pm[1], o));
        pm[2] = fmaxf(pm[2], __shfl_xor(pm[2], o)); pm[3] = fmaxf(pm[3], __shfl_xor(pm[3], o));
    }
    if (lane == 0) *(f4*)&lpm[wave][0] = pm;
    lds_barrier();
    float sca4[4], inv4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float sv = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) sv = fmaxf(sv, lpm[w][r]);
        sca4[r] = sv;
        inv4[r] = sv > 0.f ? 127.0f / sv : 0.f;
    }
    const int mr = m & 3;
    const float sca_m = mr == 0 ? sca4[0] : (mr == 1 ? sca4[1] : (mr == 2 ? sca4[2] : sca4[3]));
    const float inv_m = (row_ok && sca_m > 0.f) ? 127.0f / sca_m : 0.f;
    const float xs_row = row_ok ? sca_m / 127.0f : 0.f;
    if (qp.dbg_scale && bx == 0 && tid < M) qp.dbg_scale[tid] = xs_row;
    // ---- 3. codes into the compact image ----
#pragma unroll
    for (int j = 0; j < CI; ++j) {
        const int w = tid + j * kThreads;
        if (w < nwork) {
            const int row = (w >= nv) + (w >= 2 * nv) + (w >= 3 * nv), i = w - row * nv;
            const float inv = row == 0 ? inv4[0] : (row == 1 ? inv4[1] : (row == 2 ? inv4[2] : inv4[3]));
            const u32x2 q = quant8(hv[j], inv, thr);
            *(u32x2*)(img + (((((i >> 3) * 4 + (i & 3)) * 4 + row) << 4) + (((i >> 2) & 1) << 3))) = q;
        }
    }
    lds_barrier();
    // ---- 4. the K loop ----
    i32x4 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = i32x4{0, 0, 0, 0};
    const u32x4* xi = (const u32x4*)img + (g * 4 + mr);
    auto consume = [&](const u32x4 (&raw)[NW][T], int pb) {
        u32x4 xq[NW];
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            xq[u] = u32x4{0u, 0u, 0u, 0u};
            if (pb + u < p1) xq[u] = xi[(pb + u) * 16];
        }
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            if (pb + u >= p1) continue;
            const i32x4 xa = {(int)xq[u][0], (int)xq[u][1], (int)xq[u][2], (int)xq[u][3]};
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const u32x4 wq = raw[u][t] ^ 0x80808080u;
                const i32x4 wv = {(int)wq[0], (int)wq[1], (int)wq[2], (int)wq[3]};
                acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wv, xa, acc[t], 0, 0, 0);
            }
        }
    };
    {
        const int nb = (p1 - p0 + NW - 1) / NW;
        int b = 0;
        bool done = false;
        while (b + 2 < nb) {
            consume(rawA, p0 + b * NW);
            issue(rawA, p0 + (b + 2) * NW);
            if (!(b + 3 < nb)) {
                consume(rawB, p0 + (b + 1) * NW);
                consume(rawA, p0 + (b + 2) * NW);
                done = true;
                break;
            }
            consume(rawB, p0 + (b + 1) * NW);
            issue(rawB, p0 + (b + 3) * NW);
            b += 2;
        }
        if (!done) {
            if (b < nb) consume(rawA, p0 + b * NW);
            if (b + 1 < nb) consume(rawB, p0 + (b + 1) * NW);
        }
    }
    f4 facc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) facc[t] = f4{(float)acc[t][0], (float)acc[t][1], (float)acc[t][2], (float)acc[t][3]};
    // ---- 5. outlier correction (the producer's flag bytes), reduction, residual add ----
    f4 cacc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) { f4 z = {0.f, 0.f, 0.f, 0.f}; cacc[t] = z; }
    uint32_t fw[8];
    {
        const u32x4 f0 = *(const u32x4*)(qp.flags_in + tid * 32), f1 = *(const u32x4*)(qp.flags_in + tid * 32 + 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) { fw[j] = f0[j]; fw[4 + j] = f1[j]; }
    }
    auto xraw = [&](int row, int k) -> _Float16 { return p.xf_hi[frag_off(row, k, KS)]; };
    auto code = [&](int row, int k) -> _Float16 { return (_Float16)code_of((float)p.xf_hi[frag_off(row, k, KS)], inv_m, thr); };
    const bool fused = q8_correction<T, 8>(p, fw, tile, row_ok, xs_row, (unsigned short*)red_raw, kWaves * T * 64 * 4 * 4 / 2, wtot, cacc, xraw, code);
    f4 csv = {0.f, 0.f, 0.f, 0.f};
    if (fused) {
#pragma unroll
        for (int t = 0; t < T; ++t) *(f4*)red[wave][t][lane] = cacc[t];
        lds_barrier();
        if (wave < T) {
#pragma unroll
            for (int w = 0; w < kWaves; ++w) {
                const f4 x = *(const f4*)red[w][wave][lane];
                csv[0] += x[0]; csv[1] += x[1]; csv[2] += x[2]; csv[3] += x[3];
            }
        }
        lds_barrier();
    }
#pragma unroll
    for (int t = 0; t < T; ++t) *(f4*)red[wave][t][lane] = facc[t];
    lds_barrier();
    if (wave < T) {
        f4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            const f4 x = *(const f4*)red[w][wave][lane];
            v[0] += x[0]; v[1] += x[1]; v[2] += x[2]; v[3] += x[3];
        }
        const f4 zero = {0.f, 0.f, 0.f, 0.f};
        tile_epilogue<EPI_ADD>(p, v, zero, m, bx * T + wave, g, 0, fused, csv, zero, true, yold, true, xs_row);
    }
}

template <int T, int CI>
int launch_q8c(const Q8Params& qp, hipStream_t s) {
    const int K = qp.g.KS * 32;
    static const hipError_t attr = hipFuncSetAttribute((const void*)gemm_q8c_kernel<T, CI>, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 4);
    (void)attr;
    hipLaunchKernelGGL((gemm_q8c_kernel<T, CI>), dim3(pc_ceil_div(qp.g.ntiles, T)), dim3(kThreads), (size_t)K * 4, s, qp);
    return pc_check_launch("gemm_q8c_kernel");
}

template <int T, int EPI, int SRC, int G, int RI>
int launch_q8p_one(const Q8Params& qp, int units, int K, hipStream_t s) {
    constexpr int TT = (EPI == EPI_SILU) ? 2 * T : T;
    constexpr int kRT = (TT < 8) ? TT : 8;
    constexpr size_t kStatic = (size_t)kWaves * kRT * 64 * 16 + 1024;          // the reduction buffer + the small arrays
    constexpr size_t kLdsMax = 160 * 1024;
    const size_t lds = (size_t)K * 16 + (size_t)G * 2048;
    PC_REQUIRE(lds + kStatic <= kLdsMax, PC_ERR_ARG, "pc_gemm_q8: K = %d with %d weight tiles per workgroup does not fit the LDS", K, TT);
    static const hipError_t attr = hipFuncSetAttribute((const void*)gemm_q8p_kernel<T, EPI, SRC, G, RI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       (int)(kLdsMax - kStatic));
    (void)attr;
    hipLaunchKernelGGL((gemm_q8p_kernel<T, EPI, SRC, G, RI>), dim3(pc_ceil_div(units, T)), dim3(kThreads), lds, s, qp);
    return pc_check_launch("gemm_q8p_kernel");
}

template <int T, int EPI, bool NORM>
int launch_q8p_g(const Q8Params& qp, int units, int K, hipStream_t s) {
    if constexpr (!NORM && EPI == EPI_ADD) {
        if (qp.part.part_o) return launch_q8p_one<T, EPI, 2, 2, 2>(qp, units, K, s);           // (pc_gemm_q8 checked: M = 1, K <= 4096)
    }
    if constexpr (EPI != EPI_STORE) {
        if (qp.img8) return K <= 4096 ? launch_q8p_one<T, EPI, 3, 2, 2>(qp, units, K, s) : launch_q8p_one<T, EPI, 3, 3, 2>(qp, units, K, s);
    }
    if (qp.g.M <= 4) return K <= 4096 ? launch_q8p_one<T, EPI, NORM ? 1 : 0, 2, 2>(qp, units, K, s) : launch_q8p_one<T, EPI, NORM ? 1 : 0, 3, 2>(qp, units, K, s);
    return K <= 4096 ? launch_q8p_one<T, EPI, NORM ? 1 : 0, 2, kQ8RI>(qp, units, K, s) : launch_q8p_one<T, EPI, NORM ? 1 : 0, 3, kQ8RI>(qp, units, K, s);
}

template <int EPI, bool NORM, int kMaxT>
int launch_q8p(const Q8Params& qp, int T, int units, int K, hipStream_t s) {
    if (T > kMaxT) T = kMaxT;
    if constexpr (kMaxT >= 4) { if (T >= 4) return launch_q8p_g<4, EPI, NORM>(qp, units, K, s); }
    if constexpr (kMaxT >= 3) { if (T == 3) return launch_q8p_g<3, EPI, NORM>(qp, units, K, s); }
    if (T == 2) return launch_q8p_g<2, EPI, NORM>(qp, units, K, s);
    return launch_q8p_g<1, EPI, NORM>(qp, units, K, s);
}

template <int T, int NW>
int launch_q8f(const Q8Params& qp, hipStream_t s) {
    const dim3 grid(pc_ceil_div(qp.g.ntiles, T), qp.g.kslices);
    hipLaunchKernelGGL((gemm_q8f_kernel<T, NW>), grid, dim3(kThreads), 0, s, qp);
    return pc_check_launch("gemm_q8f_kernel");
}

// the q|k|v and gate|up launch shapes are instantiated in their own translation unit (pc_gemm_q8_norm.hip) so that the two halves of
// this template family compile side by side
int launch_q8p_rope(const Q8Params& qp, int T, int units, int K, hipStream_t s);
int launch_q8p_silu(const Q8Params& qp, int T, int units, int K, hipStream_t s);

}  // namespace pcq
