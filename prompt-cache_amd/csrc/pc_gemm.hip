// Weight-streaming projections for the small-q prefill / decode regime (M = B*q_len <= 64 rows).
//
// Replaces  q_proj/k_proj/v_proj   promptcache/model/llama2.py:345-347   (one fused [q|k|v] GEMM)
//           o_proj + residual       promptcache/model/llama2.py:405, :638
//           gate/up + SiLU*up       promptcache/model/llama2.py:242       (act_fn(gate(x)) * up(x))
//           down_proj + residual    promptcache/model/llama2.py:242, :644
//           lm_head                 promptcache/model/llama2.py:1050
//           LlamaRMSNorm (producer) promptcache/model/llama2.py:103-108   (pc_rmsnorm_frag)
//
// Regime.  With q ~ 10..50 new tokens the projections are pure weight streaming: 2*P bytes (13.5 GB for
// 7b) once per forward, ~1 FLOP/B per row -- HBM-bound, far below the MFMA roof.  The design goal is
// therefore one perfectly sequential HBM stream per wave and nothing else on the critical path.
//
// Layouts (all "fragment-major", i.e. the exact register image of mfma_f32_16x16x32_f16 operands):
//   weights  Wf[N/16][K/32][64 lanes][8 halfs]: lane l = 16*g + n holds W[16*tile + n][32*ks + 8*g .. +8].
//            Built ONCE at model load from the nn.Linear [N][K] matrix; a wave-instruction then reads 1 KiB
//            contiguous and consecutive k-steps are consecutive KiBs (the row-major layout would give 16 rows x
//            64 B per instruction).
//   activations  Xf[plane][M/16][K/32][64][8]: lane l = 16*g + m holds X[16*mt + m][32*ks + 8*g .. +8];
//            written directly in this form by the producers (pc_rmsnorm_frag, the attention epilogue, the
//            SiLU epilogue below).  Two planes, hi = fp16(x) and lo = fp16(x - hi): the MFMA is issued twice
//            per weight fragment, which is free under the HBM roof and keeps ~22 bits of the activations
//            (the parity target is the reference's fp32 CPU path; weights are exact in fp16 by construction).
//
// Kernel.  D[n][m] = sum_k W[n][k] X[m][k] with A = weight fragment, B = activation fragment, so a lane owns
// token m = lane&15 and 4 consecutive output features (C/D map: col = lane&15, row = 4*(lane>>4)+reg).
// Workgroup = 8 waves that split K eight ways over the same T output tiles (every projection then yields
// >= ~256 workgroups even for N = 4096); partial tiles are reduced through LDS in fixed wave order
// (deterministic, no atomics), then the epilogue runs on the reduced tile:
//   EPI_STORE  y[m][n]  = v                     (fp32 row-major: qkv for RoPE, logits)
//   EPI_ADD    y[m][n] += v                     (fp32 residual stream, o_proj / down_proj)
//   EPI_SILU   Of[m][j] = silu(gate_j) * up_j   (gate tile i and up tile inter/16 + i reduced in the same
//                                                workgroup; written as hi/lo fragment planes for down_proj)
//   EPI_ROPE   fused q|k|v projection epilogue: RoPE at the supplied position ids on q and k
//              (apply_rotary_pos_emb, llama2.py:202-210), rotated q -> split-precision fp16 planes for the
//              attention kernel, rotated k and v -> appended IN PLACE to the layer's KV arena (the torch.cat
//              of llama2.py:361-364).  The weight rows are permuted at load so that a 16-row tile holds 8
//              rotary pairs (rows 0-7: features 8j..8j+7, rows 8-15: the partners D/2 + 8j..): the partner of a
//              lane's value sits in lane ^ 32, one shuffle, no extra pass and no fp32 q|k|v round trip.
// Algorithmic bytes per launch: N*K*2 (weights once) [+ M*K*4 activations from L2 per workgroup].
#include <hip/hip_fp16.h>
#include <string.h>

#include "pc_common.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWaves = 8;
constexpr int kThreads = kWaves * 64;
enum { EPI_STORE = 0, EPI_ADD = 1, EPI_SILU = 2, EPI_ROPE = 3, EPI_GELU = 4 };

struct RopeEpi {          // EPI_ROPE outputs
    const float2* cs;     // [B*q_len][D/2] (cos, sin) from pc_rope_table
    _Float16* q_hi; _Float16* q_lo; int64_t q_ts;        // [B*q_len][H*D] planes, token stride q_ts
    _Float16* k_arena; _Float16* v_arena; int64_t a_bs, a_hs;
    _Float16* k_lo; _Float16* v_lo; int64_t lo_bs, lo_hs;   // optional fp16 residuals of the new K / V rows, [B][Hkv][rows][D]
    int32_t lo_base;      // residual row of token tt: tt (lo_base = -1), past + tt - lo_base (>= 0), past + tt - past_len_dev[1] (-2)
    const int32_t* past_len_dev;
    int32_t H, Hkv, D, q_len, past_len;
};

struct GemmParams {
    const _Float16* wf;      // [ntiles][KS][64][8]
    const _Float16* xf_hi;   // [MT][KS][64][8]
    const _Float16* xf_lo;   // same, may be null (single pass)
    float* y; int64_t ldy;   // EPI_STORE / EPI_ADD
    _Float16* of_hi; _Float16* of_lo; int32_t KSo;  // EPI_SILU: output planes [MT][KSo][64][8]
    int32_t M, ntiles, KS, npairs;
    int32_t kslices; int64_t slab_stride;   // EPI_STORE only: grid.y K-slices, slice s writes y + s*slab_stride
    // NORM activation source (M <= 16): the fp32 residual stream itself; RMSNorm is folded into the launch
    const float* xn; const _Float16* gamma; float eps;
    // int8 weights (W8 kernels, M <= 64): wf is the fragment image of OFFSET-BINARY bytes (q + 128), 8 B per lane per
    // k-step, wscale[n] the fp32 scale of output feature n (row order of the image); y = scale * sum_k q[n][k] x[k]
    const float* wscale; int32_t w8;
    // LLM.int8 activations (pc_int8.hip): the activation planes hold int8 CODES, xscale[m] = SCA[m] / 127 rescales row m, and
    // corr[m][n] (row stride ldc, output-feature index in the image's row order) is added before the epilogue's
    // nonlinearity when *corr_has != 0 (the fp16 outlier part of the decomposition)
    const float* xscale; const float* corr; int64_t ldc; const int32_t* corr_has;
    // ... or the correction is computed INSIDE the launch (pc_gemm_*_a8c): oflags = the K outlier-column flag bytes of
    // pc_quant_act_i8 (buffer of >= 16384 bytes, zero behind K), xraw = the fp16 activations (fragment plane, same layout as the
    // code plane xf_hi), cbt = the int8 weight codes transposed [K][ldt] in ORIGINAL row order, row_perm = image row -> original
    // row (q|k|v's rotary permutation) or null
    const unsigned char* oflags; const _Float16* xraw; const signed char* cbt; int64_t ldt; const int32_t* row_perm;
    RopeEpi rope;
};

__device__ __forceinline__ h8 ldg_h8(const _Float16* p) { return *(const h8*)p; }
__device__ __forceinline__ h8 ldg_h8_nt(const _Float16* p) {
    return __builtin_bit_cast(h8, __builtin_nontemporal_load((const u32x4*)p));
}

// Workgroup barrier that waits for this wave's LDS traffic only: `__syncthreads()` also drains vmcnt, i.e. every global
// load the wave has in flight -- the row-split kernel's staging loads, pc_gemm_chain's cross-barrier weight prefetch.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// int8 weights: 8 offset-binary bytes (two dwords, one k-step of this lane) -> fp16, exactly: the byte u goes under the
// exponent of 1024.0 (0x6400 | u = 1024 + u), then 1152 = 1024 + 128 is subtracted (v_perm_b32 + v_pk_add_f16: 8 VALU
// ops per fragment next to two 8-pass MFMAs).
__device__ __forceinline__ h8 cvt_w8(uint32_t d0, uint32_t d1) {
    const h2v bias = {(_Float16)1152.0f, (_Float16)1152.0f};
    const h2v a0 = __builtin_bit_cast(h2v, __builtin_amdgcn_perm(0x64646464u, d0, 0x04010400u)) - bias;
    const h2v b0 = __builtin_bit_cast(h2v, __builtin_amdgcn_perm(0x64646464u, d0, 0x04030402u)) - bias;
    const h2v a1 = __builtin_bit_cast(h2v, __builtin_amdgcn_perm(0x64646464u, d1, 0x04010400u)) - bias;
    const h2v b1 = __builtin_bit_cast(h2v, __builtin_amdgcn_perm(0x64646464u, d1, 0x04030402u)) - bias;
    const h8 out = {a0[0], a0[1], b0[0], b0[1], a1[0], a1[1], b1[0], b1[1]};
    return out;
}

// position of element (row m, feature k) in a fragment-major plane with KS k-steps
__device__ __forceinline__ int64_t frag_off(int m, int k, int KS) {
    return ((((int64_t)(m >> 4) * KS + (k >> 5)) * 64) + ((k & 31) >> 3) * 16 + (m & 15)) * 8 + (k & 7);
}

// One k-block of U k-steps: issue every load (U * (TT + 2*MT) KiB per wave in flight), then the MFMAs.
// The activation loads are predicated per lane on "row m exists" (one exec-masked region per block): pad
// rows cost no L2 traffic -- at M = 12 that is 25 % of the activation bytes, at M = 1 (decode) 94 % -- and
// their stale register contents only reach output columns that are never stored.
// TAIL: the last, partial block of a wave's K range (nvalid < U k-steps): the missing steps re-read the last
// valid one and their weight fragments are zeroed, so the tail keeps the same load depth as a full block
// instead of degenerating into nvalid serial load->wait->MFMA round trips.
// PRE: the block's weight fragments were fetched earlier (wpre, by prefetch_first_block in front of a grid barrier of
// pc_gemm_chain); only the activation loads are issued here.
template <int MT, int TT, bool TWO, int U, bool TAIL, bool W8 = false, bool PRE = false>
__device__ __forceinline__ void k_block(const _Float16* const (&wbase)[TT], const _Float16* xh_base,
                                        const _Float16* xl_base, int KS, int ks, int nvalid, const bool (&row_ok)[MT],
                                        f4 (&acc)[MT][TT], const h8 (*wpre)[TT] = nullptr) {
    static_assert(!PRE || (!TAIL && !W8), "prefetched first blocks are full fp16 blocks");
    // W8: the image holds k-step PAIRS -- a lane's 16 bytes are its 8 values of k-step 2s and of 2s + 1 -- so one
    // global_load_dwordx4 feeds four MFMAs; ks, nvalid and U are even (the K ranges are cut on pair boundaries).  The
    // raw bytes wait in registers (half of what fp16 fragments take) and are converted right before their MFMAs.
    constexpr int NW = W8 ? U / 2 : U;
    static_assert(!W8 || U % 2 == 0, "int8 weights come in k-step pairs");
    h8 w[W8 ? 1 : U][TT], xh[U][MT], xl[U][MT];
    u32x4 raw[W8 ? NW : 1][TT];
#pragma unroll
    for (int u = 0; u < NW; ++u)
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            if constexpr (W8) {
                const int uu = (TAIL && 2 * u >= nvalid) ? nvalid / 2 - 1 : u;
                raw[u][t] = __builtin_nontemporal_load((const u32x4*)(wbase[t] + (int64_t)((ks >> 1) + uu) * 512));
            } else if constexpr (PRE) {
                w[u][t] = wpre[u][t];
            } else {
                const int uu = (TAIL && u >= nvalid) ? nvalid - 1 : u;
                w[u][t] = ldg_h8_nt(wbase[t] + (int64_t)(ks + uu) * 512);   // 1 KiB / wave, streamed once
            }
        }
#pragma unroll
    for (int a = 0; a < MT; ++a) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            xh[u][a] = z;
            if (TWO) xl[u][a] = z;
        }
        if (row_ok[a]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int uu = (TAIL && u >= nvalid) ? nvalid - 1 : u;
                const int64_t off = ((int64_t)a * KS + ks + uu) * 512;
                xh[u][a] = ldg_h8(xh_base + off);                                                // L2-resident
                if (TWO) xl[u][a] = ldg_h8(xl_base + off);
            }
        }
    }
    // keep the whole block's loads in flight: hipcc otherwise sinks each load next to its MFMA and waits
    // vmcnt(0) per k-step (measured in the ISA), which turns a streaming kernel into a latency chain
    __builtin_amdgcn_sched_barrier(0);
    if (TAIL && !W8) {
#pragma unroll
        for (int u = 1; u < U; ++u)
            if (u >= nvalid) {
                h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int t = 0; t < TT; ++t) w[u][t] = z;
            }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (W8 && TAIL && u >= nvalid) continue;          // (wave-uniform) the re-read pair contributes nothing
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            h8 wv;
            if constexpr (W8) wv = (u & 1) ? cvt_w8(raw[u >> 1][t][2], raw[u >> 1][t][3]) : cvt_w8(raw[u >> 1][t][0], raw[u >> 1][t][1]);
            else wv = w[u][t];
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                acc[a][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, xh[u][a], acc[a][t], 0, 0, 0);
                if (TWO) acc[a][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, xl[u][a], acc[a][t], 0, 0, 0);
            }
        }
    }
}

// NORM variant of k_block for M <= 16: the activation operand is produced on the fly from the fp32 residual stream
// x[M][K] and the RMSNorm weight: the wave loads 8 floats of its row per k-step (the same bytes as the two fp16
// planes), forms g*x as a split-precision pair and accumulates sum(x^2) of its K slice in `ss`.  The 1/rms factor
// is a per-row scalar and the GEMM is linear in the activations, so it is applied to the reduced tile in the
// epilogue -- LlamaRMSNorm (llama2.py:103-108) costs no launch and no pass over x of its own.  Needs |g*x| < 65504.
template <int TT, int U, bool TAIL, bool W8 = false, bool PRE = false>
__device__ __forceinline__ void k_block_norm(const _Float16* const (&wbase)[TT], const float* xrow, const _Float16* gam,
                                             int ks, int nvalid, bool row_ok, f4 (&acc)[1][TT], float& ss,
                                             const h8 (*wpre)[TT] = nullptr) {
    static_assert(!PRE || (!TAIL && !W8), "prefetched first blocks are full fp16 blocks");
    constexpr int NW = W8 ? U / 2 : U;                   // W8: k-step pairs per 16-byte load, see k_block
    static_assert(!W8 || U % 2 == 0, "int8 weights come in k-step pairs");
    h8 w[W8 ? 1 : U][TT], gw[U];
    u32x4 raw[W8 ? NW : 1][TT];
    f4 xa[U][2];
#pragma unroll
    for (int u = 0; u < NW; ++u)
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            if constexpr (W8) {
                const int uu = (TAIL && 2 * u >= nvalid) ? nvalid / 2 - 1 : u;
                raw[u][t] = __builtin_nontemporal_load((const u32x4*)(wbase[t] + (int64_t)((ks >> 1) + uu) * 512));
            } else if constexpr (PRE) {
                w[u][t] = wpre[u][t];
            } else {
                const int uu = (TAIL && u >= nvalid) ? nvalid - 1 : u;
                w[u][t] = ldg_h8_nt(wbase[t] + (int64_t)(ks + uu) * 512);
            }
        }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int uu = (TAIL && u >= nvalid) ? nvalid - 1 : u;
        f4 z = {0.f, 0.f, 0.f, 0.f};
        xa[u][0] = z; xa[u][1] = z;
        gw[u] = *(const h8*)(gam + (ks + uu) * 32);
    }
    if (row_ok) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int uu = (TAIL && u >= nvalid) ? nvalid - 1 : u;
            xa[u][0] = *(const f4*)(xrow + (ks + uu) * 32);
            xa[u][1] = *(const f4*)(xrow + (ks + uu) * 32 + 4);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (TAIL && u >= nvalid) continue;              // the re-read k-step contributes nothing
        h8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xv = e < 4 ? xa[u][0][e] : xa[u][1][e - 4];
            ss += xv * xv;
            const float v = xv * (float)gw[u][e];
            _Float16 vh, vl;
            pc_split(v, vh, vl);
            hi[e] = vh; lo[e] = vl;
        }
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            h8 wv;
            if constexpr (W8) wv = (u & 1) ? cvt_w8(raw[u >> 1][t][2], raw[u >> 1][t][3]) : cvt_w8(raw[u >> 1][t][0], raw[u >> 1][t][1]);
            else wv = w[u][t];
            acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, hi, acc[0][t], 0, 0, 0);
            acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, lo, acc[0][t], 0, 0, 0);
        }
    }
}

// Epilogue of one reduced 16 x 16 tile.  v (and u = the "up" tile for EPI_SILU) follow the MFMA C/D map: this lane
// holds output row `row` (token) and features unit*16 + 4*g .. +3.  Must be called by all 64 lanes of a wave
// (EPI_ROPE exchanges rotary partners across lanes).
template <int EPI>
__device__ __forceinline__ void tile_epilogue(const GemmParams& p, f4 v, f4 u, int row, int unit, int g, int slice,
                                              bool fused_corr = false, f4 fused_cv = f4{0.f, 0.f, 0.f, 0.f},
                                              f4 fused_cu = f4{0.f, 0.f, 0.f, 0.f}) {
    const int nunits = (EPI == EPI_SILU) ? p.npairs : p.ntiles;
    if (p.wscale && unit < nunits) {        // int8 weights: per-output-feature scale (linear, so K-sliced partials scale too)
        const f4 sv = *(const f4*)(p.wscale + unit * 16 + g * 4);
        v[0] *= sv[0]; v[1] *= sv[1]; v[2] *= sv[2]; v[3] *= sv[3];
        if (EPI == EPI_SILU) {
            const f4 su = *(const f4*)(p.wscale + (p.npairs + unit) * 16 + g * 4);
            u[0] *= su[0]; u[1] *= su[1]; u[2] *= su[2]; u[3] *= su[3];
        }
    }
    if (p.xscale && unit < nunits && row < p.M) {
        const float xs = p.xscale[row];
        v[0] *= xs; v[1] *= xs; v[2] *= xs; v[3] *= xs;
        if (EPI == EPI_SILU) { u[0] *= xs; u[1] *= xs; u[2] *= xs; u[3] *= xs; }
        if (fused_corr) {                   // the correction was accumulated inside this launch
            v[0] += fused_cv[0]; v[1] += fused_cv[1]; v[2] += fused_cv[2]; v[3] += fused_cv[3];
            if (EPI == EPI_SILU) { u[0] += fused_cu[0]; u[1] += fused_cu[1]; u[2] += fused_cu[2]; u[3] += fused_cu[3]; }
        } else if (*p.corr_has) {
            const f4 cv = *(const f4*)(p.corr + (int64_t)row * p.ldc + unit * 16 + g * 4);
            v[0] += cv[0]; v[1] += cv[1]; v[2] += cv[2]; v[3] += cv[3];
            if (EPI == EPI_SILU) {
                const f4 cu = *(const f4*)(p.corr + (int64_t)row * p.ldc + (p.npairs + unit) * 16 + g * 4);
                u[0] += cu[0]; u[1] += cu[1]; u[2] += cu[2]; u[3] += cu[3];
            }
        }
    }
    if (unit < nunits && row < p.M) {
        if (EPI == EPI_SILU) {
            const int j0 = unit * 16 + g * 4;   // intermediate feature index of v[0]
            h4 hi, lo;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float s = (v[r] / (1.0f + __expf(-v[r]))) * u[r];
                _Float16 sh, sl;
                pc_split(s, sh, sl);
                hi[r] = sh; lo[r] = sl;
            }
            const int64_t off = frag_off(row, j0, p.KSo);
            *(h4*)(p.of_hi + off) = hi;
            *(h4*)(p.of_lo + off) = lo;
        } else if (EPI == EPI_GELU) {
            // nn.GELU() (falcon.py:726, exact erf form) of the reduced tile, as split-precision planes for dense_4h_to_h
            h4 hi, lo;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float s = 0.5f * v[r] * (1.0f + erff(v[r] * 0.70710678118654752f));
                _Float16 sh, sl;
                pc_split(s, sh, sl);
                hi[r] = sh; lo[r] = sl;
            }
            const int64_t off = frag_off(row, unit * 16 + g * 4, p.KSo);
            *(h4*)(p.of_hi + off) = hi;
            *(h4*)(p.of_lo + off) = lo;
        } else if (EPI == EPI_ROPE) {
            // handled below (needs the cross-lane exchange from every lane, valid or not)
        } else {
            float* yp = p.y + (int64_t)slice * p.slab_stride + (int64_t)row * p.ldy + unit * 16 + g * 4;
            if (EPI == EPI_ADD) {
                const f4 old = *(const f4*)yp;
                v[0] += old[0]; v[1] += old[1]; v[2] += old[2]; v[3] += old[3];
            }
            *(f4*)yp = v;
        }
    }
    if (EPI == EPI_ROPE) {
        const RopeEpi& e = p.rope;
        // partner half of every value lives in lane ^ 32 (rows 8..15 of the permuted tile)
        f4 pv;
        pv[0] = __shfl_xor(v[0], 32); pv[1] = __shfl_xor(v[1], 32); pv[2] = __shfl_xor(v[2], 32); pv[3] = __shfl_xor(v[3], 32);
        if (unit < nunits && row < p.M) {
            const int tpd = e.D >> 4;                       // tiles per head
            const int hh = unit / tpd, j = unit - hh * tpd;
            const bool is_hi = g >= 2;
            const int i0 = 8 * j + 4 * (g & 1);            // rotary frequency index of v[0]
            const int d0 = i0 + (is_hi ? (e.D >> 1) : 0);  // feature index inside the head
            const int bb = row / e.q_len, tt = row - bb * e.q_len;
            if (hh < e.H + e.Hkv) {
                const float2* cs = e.cs + (int64_t)row * (e.D >> 1) + i0;
                h4 hi, lo;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float2 w = cs[r];
                    // q*cos + rotate_half(q)*sin (llama2.py:208): low half pairs with -high, high with +low
                    const float o = is_hi ? (v[r] * w.x + pv[r] * w.y) : (v[r] * w.x - pv[r] * w.y);
                    _Float16 oh, ol;
                    pc_split(o, oh, ol);
                    hi[r] = oh; lo[r] = ol;
                }
                if (hh < e.H) {
                    const int64_t off = (int64_t)row * e.q_ts + (int64_t)hh * e.D + d0;
                    *(h4*)(e.q_hi + off) = hi;
                    *(h4*)(e.q_lo + off) = lo;
                } else {
                    const int past = e.past_len_dev ? *e.past_len_dev : e.past_len;
                    *(h4*)(e.k_arena + bb * e.a_bs + (int64_t)(hh - e.H) * e.a_hs + (int64_t)(past + tt) * e.D + d0) = hi;
                    if (e.k_lo) {
                        const int lr = e.lo_base == -1 ? tt : past + tt - (e.lo_base == -2 ? e.past_len_dev[1] : e.lo_base);
                        *(h4*)(e.k_lo + bb * e.lo_bs + (int64_t)(hh - e.H) * e.lo_hs + (int64_t)lr * e.D + d0) = lo;
                    }
                }
            } else {
                const int past = e.past_len_dev ? *e.past_len_dev : e.past_len;
                h4 hv, lv;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    _Float16 oh, ol;
                    pc_split(v[r], oh, ol);
                    hv[r] = oh; lv[r] = ol;
                }
                *(h4*)(e.v_arena + bb * e.a_bs + (int64_t)(hh - e.H - e.Hkv) * e.a_hs + (int64_t)(past + tt) * e.D + d0) = hv;
                if (e.v_lo) {
                    const int lr = e.lo_base == -1 ? tt : past + tt - (e.lo_base == -2 ? e.past_len_dev[1] : e.lo_base);
                    *(h4*)(e.v_lo + bb * e.lo_bs + (int64_t)(hh - e.H - e.Hkv) * e.lo_hs + (int64_t)lr * e.D + d0) = lv;
                }
            }
        }
    }
}

// K range [ks0, ks1) of wave `wave` in K-slice `by` of a launch (grid.y slices K across workgroups, then eight ways across
// the waves of a workgroup)
template <bool W8>
__device__ __forceinline__ void wave_k_range(const GemmParams& p, int by, int wave, int& ks0, int& ks1) {
    const int KS = p.KS;
    int ksq = (KS + p.kslices - 1) / p.kslices;
    if (W8) ksq = (ksq + 1) & ~1;                         // int8 images are cut on k-step pairs (KS is even)
    const int kq0 = by * ksq;
    const int kq1 = (kq0 + ksq < KS) ? kq0 + ksq : KS;
    int ksw = (kq1 - kq0 + kWaves - 1) / kWaves;
    if (W8) ksw = (ksw + 1) & ~1;
    ks0 = kq0 + wave * ksw;
    ks1 = (ks0 + ksw < kq1) ? ks0 + ksw : kq1;
}

// weight tile ids of workgroup bx (clamped: a clamped duplicate tile recomputes a valid tile and is not stored)
template <int T, int EPI>
__device__ __forceinline__ void wg_tiles(const GemmParams& p, int bx, int (&tile)[(EPI == EPI_SILU) ? 2 * T : T]) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int i = bx * T + t;
        if (EPI == EPI_SILU) {
            const int ic = i < p.npairs ? i : p.npairs - 1;
            tile[t] = ic;                 // gate rows
            tile[T + t] = p.npairs + ic;  // up rows
        } else {
            tile[t] = i < p.ntiles ? i : p.ntiles - 1;
        }
    }
}

// The first k-block's weight fragments of workgroup bx (fp16 image), issued ahead of time: they do not depend on any
// activation, so a persistent workgroup (pc_gemm_chain) fetches them BEFORE it waits at the grid barrier in front of the
// phase -- the HBM stream keeps running while the previous phase's results cross the chip.
template <int T, int EPI, int U>
__device__ __forceinline__ void prefetch_first_block(const GemmParams& p, int bx, h8 (&wpre)[U][(EPI == EPI_SILU) ? 2 * T : T]) {
    constexpr int TT = (EPI == EPI_SILU) ? 2 * T : T;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int ks0, ks1;
    wave_k_range<false>(p, 0, wave, ks0, ks1);
    int tile[TT];
    wg_tiles<T, EPI>(p, bx, tile);
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int t = 0; t < TT; ++t)
            wpre[u][t] = ldg_h8_nt(p.wf + ((int64_t)tile[t] * p.KS * 64 + lane) * 8 + (int64_t)(ks0 + u) * 512);
}

// One workgroup's share of a launch: block (bx, by) of the grid.  PRE: wpre holds the first k-block's weight fragments
// (prefetch_first_block; needs >= U k-steps per wave, the launcher checks).
struct NoHook { __device__ __forceinline__ void operator()() const {} };

// after_k() runs between the K loop and the reduction (pc_gemm_chain: the early prefetch of the next phase's weights).
template <int MT, int T, int EPI, bool TWO, int U, bool NORM = false, bool W8 = false, bool PRE = false, class AfterK = NoHook>
__device__ __forceinline__ void gemm_skinny_body(const GemmParams& p, const int bx, const int by,
                                                 float* red_raw, float (*ssl)[16],
                                                 const h8 (*wpre)[(EPI == EPI_SILU) ? 2 * T : T] = nullptr,
                                                 AfterK after_k = AfterK()) {
    static_assert(!NORM || (MT == 1 && TWO), "the fused-RMSNorm source is for one row tile");
    constexpr int TT = (EPI == EPI_SILU) ? 2 * T : T;   // weight tiles reduced per workgroup
    constexpr int TPI = (EPI == EPI_SILU) ? 2 : 1;                 // tiles per output item
    constexpr int kRT = (MT * TT < 8) ? MT * TT : 8;               // tiles per wave in the reduction buffer (<= 64 KiB)
    constexpr int IPR = kRT / TPI;                                 // items per reduction round
    float (*red)[kRT][64][4] = (float (*)[kRT][64][4])red_raw;     // [kWaves][kRT][64][4] fp32, caller-owned LDS

    const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: scalar loop control
    const int KS = p.KS;
    // K range of this workgroup (grid.y slices K across workgroups; partial sums then go to per-slice slabs
    // that the consumer -- pc_rmsnorm_frag -- adds up in fixed order), then eight ways across the waves
    int ks0, ks1;
    wave_k_range<W8>(p, by, wave, ks0, ks1);
    int tile[TT];
    wg_tiles<T, EPI>(p, bx, tile);

    f4 acc[MT][TT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int t = 0; t < TT; ++t) { f4 z = {0.f, 0.f, 0.f, 0.f}; acc[a][t] = z; }

    const _Float16* wbase[TT];
#pragma unroll
    // (W8: [tile][KS/2][64][16 B] -- the same 512 halfs per unit as fp16, the unit being a k-step pair)
    for (int t = 0; t < TT; ++t) wbase[t] = p.wf + ((int64_t)tile[t] * (W8 ? KS / 2 : KS) * 64 + lane) * 8;
    const _Float16* xh_base = p.xf_hi + lane * 8;
    const _Float16* xl_base = TWO ? p.xf_lo + lane * 8 : nullptr;

    bool row_ok[MT];
#pragma unroll
    for (int a = 0; a < MT; ++a) row_ok[a] = a * 16 + m < p.M;
    int ks = ks0;
    [[maybe_unused]] float ss = 0.f;
    if constexpr (NORM) {
        const float* xrow = p.xn + (int64_t)m * (KS * 32) + g * 8;
        const _Float16* gam = p.gamma + g * 8;
        if constexpr (PRE) {
            k_block_norm<TT, U, false, W8, true>(wbase, xrow, gam, ks, U, row_ok[0], acc, ss, wpre);
            ks += U;
        }
        for (; ks + U <= ks1; ks += U) k_block_norm<TT, U, false, W8>(wbase, xrow, gam, ks, U, row_ok[0], acc, ss);
        if (ks < ks1) k_block_norm<TT, U, true, W8>(wbase, xrow, gam, ks, ks1 - ks, row_ok[0], acc, ss);
        ss += __shfl_xor(ss, 16);
        ss += __shfl_xor(ss, 32);
        if (g == 0) ssl[wave][m] = ss;                   // this wave's share of sum(x^2) of row m
    } else {
        if constexpr (PRE) {
            k_block<MT, TT, TWO, U, false, W8, true>(wbase, xh_base, xl_base, KS, ks, U, row_ok, acc, wpre);
            ks += U;
        }
        for (; ks + U <= ks1; ks += U) k_block<MT, TT, TWO, U, false, W8>(wbase, xh_base, xl_base, KS, ks, U, row_ok, acc);
        if (ks < ks1) k_block<MT, TT, TWO, U, true, W8>(wbase, xh_base, xl_base, KS, ks, ks1 - ks, row_ok, acc);
    }
    after_k();

    // ---- LLM.int8 outlier correction inside the launch (pc_gemm_*_a8c) ----
    // corr[t][n] = sum over the outlier columns k of  X[t][k] * fp16(CB[n][k] * s[n])  -  CA[t][k] * CB[n][k] * xs[t] * s[n]
    // (pc_int8.hip).  Every workgroup compacts the flag bytes into a column list in the (still idle) reduction buffer -- 32
    // bytes per thread, ascending, prefix sums by shuffles -- and its eight waves deal the columns among themselves, each
    // accumulating its share for the workgroup's tiles in the MFMA C layout; the shares meet in the split-K reduction below.
    // A stand-alone correction launch costs ~4 us even when there is nothing to correct (the usual case behind a norm).
    [[maybe_unused]] f4 cacc[MT][TT];
    [[maybe_unused]] bool fused = false;
    if constexpr (W8) {
        fused = p.oflags != nullptr;
        if (fused) {
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int t = 0; t < TT; ++t) { f4 z = {0.f, 0.f, 0.f, 0.f}; cacc[a][t] = z; }
            unsigned short* cols = (unsigned short*)red_raw;
            constexpr int kColsCap = kWaves * kRT * 64 * 4 * 4 / 2;         // u16 entries that fit the reduction buffer
            int* wtot = (int*)&ssl[0][0];
            const u32x4 f0 = *(const u32x4*)(p.oflags + tid * 32), f1 = *(const u32x4*)(p.oflags + tid * 32 + 16);
            auto nz4 = [](uint32_t w) { return ((w & 0xffu) ? 1 : 0) + ((w & 0xff00u) ? 1 : 0) + ((w & 0xff0000u) ? 1 : 0) + ((w >> 24) ? 1 : 0); };
            const int mine = nz4(f0[0]) + nz4(f0[1]) + nz4(f0[2]) + nz4(f0[3]) + nz4(f1[0]) + nz4(f1[1]) + nz4(f1[2]) + nz4(f1[3]);
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
            if (mine) {
                auto put4 = [&](uint32_t w, int base) {
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        if ((w >> (8 * b)) & 0xffu) { if (off < kColsCap) cols[off] = (unsigned short)(base + b); ++off; }
                };
                const int c0 = tid * 32;
                put4(f0[0], c0); put4(f0[1], c0 + 4); put4(f0[2], c0 + 8); put4(f0[3], c0 + 12);
                put4(f1[0], c0 + 16); put4(f1[1], c0 + 20); put4(f1[2], c0 + 24); put4(f1[3], c0 + 28);
            }
            lds_barrier();
            if (total > kColsCap) total = kColsCap;
            if (total > 0) {
                // MFMA form: 32 compacted columns are one k-step.  A lane gathers the operands of its fragment slots -- weight
                // lane (n, g): codes CB[n][cols[32 s + 8 g + e]] (-> the fp16 dequantised weights and the codes themselves),
                // activation lane (m, g): X and CA at the same columns -- for two MFMAs per tile: sum X * fp16(CB * s) and the
                // integer sum CA * CB (exact in fp32), which leaves as  - sum * xs[m] * s[n]  in the C layout.  Slots behind
                // the last column hold zeros.  (A scalar loop over the columns cost 33 us at 460 columns.)
                float wsc[TT][4], xsr[MT], wsa[TT];
                int nra[TT];
#pragma unroll
                for (int t = 0; t < TT; ++t) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) wsc[t][r] = p.wscale[tile[t] * 16 + g * 4 + r];   // (tile ids are clamped: valid)
                    const int na = tile[t] * 16 + m;                     // the weight row of this lane's A-operand slot
                    wsa[t] = p.wscale[na];
                    nra[t] = p.row_perm ? p.row_perm[na] : na;
                }
#pragma unroll
                for (int a = 0; a < MT; ++a) xsr[a] = row_ok[a] ? p.xscale[a * 16 + m] : 0.f;
                f4 iacc[MT][TT];
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int t = 0; t < TT; ++t) { f4 z = {0.f, 0.f, 0.f, 0.f}; iacc[a][t] = z; }
                const int nks = (total + 31) >> 5;
                for (int sblk = wave; sblk < nks; sblk += kWaves) {
                    int cj[8];
                    bool okc[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int j = sblk * 32 + g * 8 + e;
                        okc[e] = j < total;
                        cj[e] = cols[okc[e] ? j : 0];
                    }
                    h8 xb[MT], cb[MT];
#pragma unroll
                    for (int a = 0; a < MT; ++a)
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const bool ok = okc[e] && row_ok[a];
                            const int64_t xo = frag_off(a * 16 + m, cj[e], KS);
                            xb[a][e] = ok ? p.xraw[xo] : (_Float16)0;
                            cb[a][e] = ok ? p.xf_hi[xo] : (_Float16)0;
                        }
#pragma unroll
                    for (int t = 0; t < TT; ++t) {
                        h8 wa, qa;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float q = okc[e] ? (float)p.cbt[(int64_t)cj[e] * p.ldt + nra[t]] : 0.f;
                            qa[e] = (_Float16)q;
                            wa[e] = (_Float16)(q * wsa[t]);              // fp16(CB * SCB / 127)
                        }
#pragma unroll
                        for (int a = 0; a < MT; ++a) {
                            cacc[a][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, xb[a], cacc[a][t], 0, 0, 0);
                            iacc[a][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa, cb[a], iacc[a][t], 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int a = 0; a < MT; ++a)
#pragma unroll
                    for (int t = 0; t < TT; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) cacc[a][t][r] -= iacc[a][t][r] * (xsr[a] * wsc[t][r]);
            }
            lds_barrier();                                   // the column list is dead: the buffer goes to the reduction
        }
    }

    // ---- split-K reduction through LDS, fixed order ----
    // An output item is one reduced tile (a gate/up pair of tiles for the SiLU epilogue).  Up to kRT tiles per
    // wave fit the LDS buffer, so the items go through it in rounds of IPR, one item per wave per round; the
    // unrolled round/slot indices keep every acc[][] access compile-time (no runtime-indexed register arrays).
    constexpr int TE = (EPI == EPI_SILU) ? T : TT;   // epilogue items per M-tile
    constexpr int NOUT = MT * TE;
    constexpr int ROUNDS = (NOUT + IPR - 1) / IPR;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
    if (r > 0) lds_barrier();                        // the previous round's readers are done with the buffer
    [[maybe_unused]] f4 csv = {0.f, 0.f, 0.f, 0.f}, csu = {0.f, 0.f, 0.f, 0.f};
    if constexpr (W8) {
        if (fused) {                                 // the waves' correction shares first (same slots, same order)
#pragma unroll
            for (int i = 0; i < IPR; ++i) {
                const int item = r * IPR + i;
                if (item < NOUT) {
                    const int a = item / TE, t = item - a * TE;
                    *(f4*)red[wave][i * TPI][lane] = cacc[a][t];
                    if (EPI == EPI_SILU) *(f4*)red[wave][i * TPI + 1][lane] = cacc[a][T + t];
                }
            }
            lds_barrier();
            if (wave < IPR && r * IPR + wave < NOUT) {
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
    }
#pragma unroll
    for (int i = 0; i < IPR; ++i) {
        const int item = r * IPR + i;
        if (item < NOUT) {
            const int a = item / TE, t = item - a * TE;
            *(f4*)red[wave][i * TPI][lane] = acc[a][t];
            if (EPI == EPI_SILU) *(f4*)red[wave][i * TPI + 1][lane] = acc[a][T + t];
        }
    }
    lds_barrier();       // (not __syncthreads: a chained caller has the next phase's weight loads in flight here)

    const int item = r * IPR + wave;
    if (wave < IPR && item < NOUT) {
        const int a = item / TE;
        const int t = item - a * TE;
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
        if constexpr (NORM) {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) tot += ssl[w][m];
            const float rs = rsqrtf(tot / (float)(KS * 32) + p.eps);
            v[0] *= rs; v[1] *= rs; v[2] *= rs; v[3] *= rs;
            u[0] *= rs; u[1] *= rs; u[2] *= rs; u[3] *= rs;
        }
        if constexpr (W8) tile_epilogue<EPI>(p, v, u, a * 16 + m, bx * T + t, g, by, fused, csv, csu);
        else tile_epilogue<EPI>(p, v, u, a * 16 + m, bx * T + t, g, by);
    }
    }   // rounds
}

template <int MT, int T, int EPI, bool TWO, int U, bool NORM = false, bool W8 = false>
__global__ __launch_bounds__(kThreads) void gemm_skinny_kernel(const GemmParams p) {
    constexpr int TT = (EPI == EPI_SILU) ? 2 * T : T;
    constexpr int kRT = (MT * TT < 8) ? MT * TT : 8;
    __shared__ __attribute__((aligned(16))) float red[kWaves * kRT * 64 * 4];
    __shared__ float ssl[kWaves][16];
    gemm_skinny_body<MT, T, EPI, TWO, U, NORM, W8>(p, (int)blockIdx.x, (int)blockIdx.y, red, ssl);
}

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
                                            float* red, float (*ssl)[16]) {
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
            if constexpr (HAVE_PRE) gemm_skinny_body<1, T, EPI, true, U, NORM, false, true>(p, wg, 0, red, ssl, pre, early);
            else gemm_skinny_body<1, T, EPI, true, U, NORM, false, false>(p, wg, 0, red, ssl, nullptr, early);
        }
        for (int bx = wg + nwg; bx < nblk; bx += nwg) {
            const bool lastb = bx + nwg >= nblk;
            auto early = [&]() { if (lastb && early_wave) prefetch_next(); };
            lds_barrier();                                // the reduction buffer of the previous block is free
            gemm_skinny_body<1, T, EPI, true, U, NORM, false, false>(p, bx, 0, red, ssl, nullptr, early);
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
    __shared__ float ssl[kWaves][16];
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
    chain_phase<TO, EPI_ADD, UO, false, false>(c.ph[0], c.nblk[0], gs, 1, false, nullptr, pf_g, red, ssl);
    chain_phase<TG, EPI_SILU, UG, true, true>(c.ph[1], c.nblk[1], gs, 2, !qkv, pre_g, pf_d, red, ssl);
    if (qkv) {
        chain_phase<TD, EPI_ADD, UD, false, true>(c.ph[2], c.nblk[2], gs, 3, true, pre_d, pf_q, red, ssl);
        chain_phase<TQ, EPI_ROPE, UQ, true, true>(c.ph[3], c.nblk[3], gs, 0, false, pre_q, pf_none, red, ssl);
    } else {
        chain_phase<TD, EPI_ADD, UD, false, true>(c.ph[2], c.nblk[2], gs, 0, false, pre_d, pf_none, red, ssl);
    }
}

// RMSNorm producing split-precision fragment planes: one workgroup per row.
// Optional prologue: x[row] += slab[0][row] + slab[1][row] + ... (fixed order), the K-sliced partial sums a
// preceding pc_gemm_skinny left behind -- the residual add of llama2.py:638 / :644 happens here, in place.
// LN = true: torch.nn.LayerNorm instead (Falcon, falcon.py:757): mean removed (two passes over the register-resident
// row, no cancellation), bias `b` added after the gain.
template <int G, bool LN = false>   // G = 8-element groups per thread: the whole row stays in registers between the passes
__global__ __launch_bounds__(256) void rmsnorm_frag_kernel(float* __restrict__ x, const _Float16* __restrict__ w,
                                                           _Float16* __restrict__ of_hi, _Float16* __restrict__ of_lo,
                                                           int hidden, float eps, const float* __restrict__ slabs,
                                                           int nslabs, int64_t slab_stride,
                                                           const _Float16* __restrict__ b = nullptr) {
    __shared__ float red[4];
    __shared__ float redm[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const int nv = hidden >> 3;
    float* xr = x + (int64_t)row * hidden;
    f4 va[G], vb[G];
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < G; ++k) {
        const int i = tid + k * 256;
        f4 z = {0.f, 0.f, 0.f, 0.f};
        va[k] = z; vb[k] = z;
        if (i < nv) {
            va[k] = *(const f4*)(xr + i * 8);
            vb[k] = *(const f4*)(xr + i * 8 + 4);
        }
    }
    if (nslabs > 0) {
        // CH slabs at a time: all of their loads are issued before the first add (clamped, unconditional), so a chunk
        // costs one L2 round trip instead of one per slab (one workgroup per row: with 12..64 rows nothing else hides
        // the latency -- 6.5 -> ~4 us per launch at 22 rows); the adds keep the fixed slab order
        constexpr int CH = G <= 2 ? 4 : (G <= 4 ? 2 : 1);
        for (int s0 = 0; s0 < nslabs; s0 += CH) {
            f4 c[CH][G], d[CH][G];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int sj = s0 + j < nslabs ? s0 + j : nslabs - 1;
#pragma unroll
                for (int k = 0; k < G; ++k) {
                    const int i = tid + k * 256;
                    const float* sp = slabs + sj * slab_stride + (int64_t)row * hidden + (i < nv ? i : nv - 1) * 8;
                    c[j][k] = *(const f4*)sp;
                    d[j][k] = *(const f4*)(sp + 4);
                }
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                if (s0 + j < nslabs) {
#pragma unroll
                    for (int k = 0; k < G; ++k) {
                        if (tid + k * 256 < nv) {
                            va[k][0] += c[j][k][0]; va[k][1] += c[j][k][1]; va[k][2] += c[j][k][2]; va[k][3] += c[j][k][3];
                            vb[k][0] += d[j][k][0]; vb[k][1] += d[j][k][1]; vb[k][2] += d[j][k][2]; vb[k][3] += d[j][k][3];
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < G; ++k) {
            const int i = tid + k * 256;
            if (i < nv) {
                *(f4*)(xr + i * 8) = va[k];
                *(f4*)(xr + i * 8 + 4) = vb[k];
            }
        }
    }
    float mu = 0.f;
    if (LN) {
        float sm = 0.f;
#pragma unroll
        for (int k = 0; k < G; ++k)       // lanes past the end of the row hold zeros
            sm += va[k][0] + va[k][1] + va[k][2] + va[k][3] + vb[k][0] + vb[k][1] + vb[k][2] + vb[k][3];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
        if ((tid & 63) == 0) redm[tid >> 6] = sm;
        __syncthreads();
        mu = (redm[0] + redm[1] + redm[2] + redm[3]) / (float)hidden;
#pragma unroll
        for (int k = 0; k < G; ++k) {
            if (tid + k * 256 < nv) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { va[k][e] -= mu; vb[k][e] -= mu; }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < G; ++k)
        ss += va[k][0] * va[k][0] + va[k][1] * va[k][1] + va[k][2] * va[k][2] + va[k][3] * va[k][3] +
              vb[k][0] * vb[k][0] + vb[k][1] * vb[k][1] + vb[k][2] * vb[k][2] + vb[k][3] * vb[k][3];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    const float rs = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)hidden + eps);
    const int KS = hidden >> 5;
#pragma unroll
    for (int k = 0; k < G; ++k) {
        const int i = tid + k * 256;
        if (i < nv) {
            const h8 gw = *(const h8*)(w + i * 8);
            h8 bw = {0, 0, 0, 0, 0, 0, 0, 0};
            if (LN && b) bw = *(const h8*)(b + i * 8);
            h8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = (float)gw[e] * ((e < 4 ? va[k][e] : vb[k][e - 4]) * rs);
                if (LN) v += (float)bw[e];
                _Float16 vh, vl;
                pc_split(v, vh, vl);
                hi[e] = vh; lo[e] = vl;
            }
            const int64_t off = frag_off(row, i * 8, KS);
            *(h8*)(of_hi + off) = hi;
            *(h8*)(of_lo + off) = lo;
        }
    }
}

template <int MT, int T, int EPI>
int launch_one(const GemmParams& p, int units, hipStream_t s) {
    constexpr int TT = (EPI == EPI_SILU) ? 2 * T : T;
    constexpr int UD = (MT * TT >= 16) ? 1 : (MT * TT >= 8) ? 2 : (MT * TT >= 3 ? 4 : 8);     // default depth
    static const int forced = [] { const char* e = getenv("PC_GEMM_U"); return e ? atoi(e) : 0; }();
    const dim3 grid(pc_ceil_div(units, T), p.kslices), block(kThreads);
    const bool two = p.xf_lo != nullptr;
#define PC_GO(UV)                                                                                     \
    do {                                                                                              \
        if (p.w8) {   /* int8 weights: split-precision activations only.  Same k-steps per block as fp16 (half the   */ \
                      /* bytes in flight): doubling them measured slower, 23.4 vs 21.6 us on the 7b gate|up launch */ \
            constexpr int UW = ((UV) < 2) ? 2 : (((UV) > 8) ? 8 : (UV));                              \
            if constexpr (MT == 1 && EPI != EPI_ADD) {                                                \
                if (p.xn) {                                                                           \
                    hipLaunchKernelGGL((gemm_skinny_kernel<MT, T, EPI, true, UW, true, true>), grid, block, 0, s, p); \
                    break;                                                                            \
                }                                                                                     \
            }                                                                                         \
            if (p.xscale) {   /* LLM.int8 codes: ONE activation plane (the "lo" plane of the a8 calls is all zeros) */ \
                hipLaunchKernelGGL((gemm_skinny_kernel<MT, T, EPI, false, UW, false, true>), grid, block, 0, s, p); \
                break;                                                                                \
            }                                                                                         \
            hipLaunchKernelGGL((gemm_skinny_kernel<MT, T, EPI, true, UW, false, true>), grid, block, 0, s, p); \
            break;                                                                                    \
        }                                                                                             \
        if constexpr (MT == 1 && EPI != EPI_ADD) {                                                    \
            if (p.xn) {                                                                               \
                hipLaunchKernelGGL((gemm_skinny_kernel<MT, T, EPI, true, UV, true>), grid, block, 0, s, p); \
                break;                                                                                \
            }                                                                                         \
        }                                                                                             \
        if (two) hipLaunchKernelGGL((gemm_skinny_kernel<MT, T, EPI, true, UV>), grid, block, 0, s, p); \
        else hipLaunchKernelGGL((gemm_skinny_kernel<MT, T, EPI, false, UV>), grid, block, 0, s, p);    \
    } while (0)
    if constexpr (MT * TT <= 2) {
        if (forced == 16) { PC_GO(16); return pc_check_launch("gemm_skinny_kernel"); }
    }
    if constexpr (MT * TT <= 4) {
        if (forced == 8) { PC_GO(8); return pc_check_launch("gemm_skinny_kernel"); }
        if (forced == 4) { PC_GO(4); return pc_check_launch("gemm_skinny_kernel"); }
    }
    if (forced == 2) { PC_GO(2); return pc_check_launch("gemm_skinny_kernel"); }
    if constexpr (MT * TT <= 2) {
        // one or two tiles per workgroup (the N = hidden projections): measured in-graph on the 7b shapes
        // (tools/gemm_n4096_sweep.py), k-steps per block 4 / 8 / 16: o_proj 9.3 / 9.8 / 9.5 us at 12 rows, 8.0 / 8.9 / 8.9 at
        // one row; down_proj 24.3 / 24.7 / 23.2 at 12 rows, 19.1 / 20.9 / 21.8 at one row -- shallow blocks win except for
        // the long-K launch with its activation loads (more than 4 rows), which wants the deep one
        if (!forced) {
            if (p.M > 4 && p.KS >= 256) { PC_GO(16); } else { PC_GO(4); }
            return pc_check_launch("gemm_skinny_kernel");
        }
    }
    PC_GO(UD);
#undef PC_GO
    return pc_check_launch("gemm_skinny_kernel");
}

// Weight tiles per workgroup (TT = 2T for the SiLU epilogue) are limited by registers: MT * TT accumulators of 4
// VGPRs each next to the in-flight operands.  Wide workgroups matter most for MT > 1: every workgroup reads all of
// the activation planes, so that traffic is (#workgroups x planes) and at MT = 4 it exceeds the weights'.
template <int MT, int EPI>
int launch_T(const GemmParams& p, int T, int units, hipStream_t s) {
    constexpr int kMaxT = (MT == 4 ? 6 : 8) / (EPI == EPI_SILU ? 2 : 1);
    if (T > kMaxT) T = kMaxT;
    if constexpr (kMaxT >= 8) { if (T >= 8) return launch_one<MT, 8, EPI>(p, units, s); }
    if constexpr (kMaxT >= 4) { if (T >= 4) return launch_one<MT, 4, EPI>(p, units, s); }
    if constexpr (kMaxT >= 3) { if (T == 3) return launch_one<MT, 3, EPI>(p, units, s); }
    if constexpr (kMaxT >= 2) { if (T >= 2) return launch_one<MT, 2, EPI>(p, units, s); }
    return launch_one<MT, 1, EPI>(p, units, s);
}


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
constexpr int kRowsMaxM = 512;


template <int TT, int EPI, int MTW, bool TWO>
__global__ __launch_bounds__(MTW == 2 ? 1024 : 768) void gemm_rows_kernel(const GemmParams p) {
    constexpr int T = (EPI == EPI_SILU) ? TT / 2 : TT;   // output units (tiles, or gate/up pairs) per workgroup
    constexpr int KC = (TT <= 4) ? 8 : 4;                // k-steps per stage
    constexpr int F = TT * KC;                           // 1-KiB fragments per stage
    constexpr int FPW = F / kStageWaves;
    constexpr int PD = (MTW == 2) ? 3 : 1, NX = PD + 1;  // activation prefetch distance (k-steps) / register sets
    static_assert(F % kStageWaves == 0 && KC % NX == 0, "stage must split evenly over the staging waves");
    __shared__ __attribute__((aligned(16))) _Float16 wbuf[2][F][64][8];

    const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int RW = (p.M + 16 * MTW - 1) / (16 * MTW);
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
        if (nst <= 0) return;                            // empty K slice (kslices > k-steps): nothing to stream
        const int sidx = wave - RW;
        const _Float16* src[FPW];
        int kst[FPW];
#pragma unroll
        for (int i = 0; i < FPW; ++i) {
            const int f = sidx + kStageWaves * i;        // fragment of the stage: k-step-major, tile-minor
            const int kk = f / TT, tt = f - kk * TT;
            int unit = blockIdx.x * T + (EPI == EPI_SILU ? (tt < T ? tt : tt - T) : tt);
            if (unit >= nunits) unit = nunits - 1;       // clamped duplicates are computed and never stored
            const int tile = (EPI == EPI_SILU && tt >= T) ? p.npairs + unit : unit;
            kst[i] = kk;
            src[i] = p.wf + (((int64_t)tile * KS + kq0 + kk) * 64 + lane) * 8;
        }
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
                R[i] = ldg_h8_nt(src[i] + ((int64_t)se * KC - back) * 512);                   \
            }                                                                                 \
        }
#define PC_STAGE_WRITE(R, ST)                                                                 \
        {                                                                                     \
            const int se = (ST);                                                              \
            _Pragma("unroll") for (int i = 0; i < FPW; ++i) {                                 \
                const bool ok = kq0 + se * KC + kst[i] < kq1;                                 \
                *(h8*)wbuf[(ST) & 1][sidx + kStageWaves * i][lane] = ok ? R[i] : zero;        \
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
    const int mt_last = ((p.M + 15) >> 4) - 1;
#pragma unroll
    for (int a = 0; a < MTW; ++a) {
        int mt = MTW * wave + a;
        mt = mt < mt_last ? mt : mt_last;
        xa[a] = p.xf_hi + (((int64_t)mt * KS + kq0) * 64 + lane) * 8;
    }
    // row tiles this wave really owns (wave-uniform, >= 1): the MFMAs of the clamped duplicates behind the last tile are skipped
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
    for (int st = 0; st < nst; ++st) {
        lds_barrier();                                   // stage st is in wbuf[st & 1]
        const _Float16* wst = &wbuf[st & 1][0][lane][0];
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            // prefetch the activation fragments PD k-steps ahead (clamped at the end of the K range)
            const int kn = kseq(st * KC + j + PD);
#pragma unroll
            for (int a = 0; a < MTW; ++a) {
                xs[(j + PD) % NX][a] = ldg_h8(xa[a] + (int64_t)kn * 512);
                if (TWO) xsl[(j + PD) % NX][a] = ldg_h8(xa[a] + lo_delta + (int64_t)kn * 512);
            }
            __builtin_amdgcn_sched_barrier(0);           // keep the prefetch ahead of this k-step's MFMAs (see k_block)
            h8 w[TT];
#pragma unroll
            for (int t = 0; t < TT; ++t) w[t] = *(const h8*)(wst + (j * TT + t) * 512);
#pragma unroll
            for (int a = 0; a < MTW; ++a) {
                if (a < nva) {
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
            tile_epilogue<EPI>(p, acc[a][t], u, (MTW * wave + a) * 16 + m, (int)blockIdx.x * T + t, g, (int)blockIdx.y);
        }
}

// (Tried and dropped: also splitting the rows of a column group over 2-4 workgroups placed on one XCD, with the shape
// chosen by bytes-per-workgroup: 10-15 % faster on a few isolated shapes (7b q|k|v at 259 / 512 rows), but the forward
// as a whole did not gain -- q = 98: 5.8 -> 6.2 ms, config 4: 19.4 -> 20.1 ms.)
// Launch shape of the rows kernel.  Rows per compute wave: 32 while that needs <= 12 compute waves (M <= 384; more,
// narrower waves hide the L2 latency of the activation loads and spread evenly over the four SIMDs), else 64.
// Weight tiles per workgroup: the smallest of {3,4,6,8} ({2,3,4} gate/up pairs) that fits the grid into one round.
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
    const int work = units * p.kslices;
    if constexpr (EPI == EPI_SILU) {
        if (forced == 4 || two || (!forced && pc_ceil_div(work, 2) <= 256)) PC_ROWS(4);
        if (forced == 6 || !narrow || (!forced && pc_ceil_div(work, 3) <= 256)) PC_ROWS(6);
        PC_ROWS(8);
    } else {
        if (forced == 3 || (!forced && pc_ceil_div(work, 3) <= 256)) PC_ROWS(3);
        if (forced == 4 || two || (!forced && pc_ceil_div(work, 4) <= 256)) PC_ROWS(4);
        if (forced == 6 || !narrow || (!forced && pc_ceil_div(work, 6) <= 256)) PC_ROWS(6);
        PC_ROWS(8);
    }
#undef PC_ROWS
}

template <int EPI>
int launch_MT(const GemmParams& p, int T, int units, hipStream_t s) {
    const int mt = pc_ceil_div(p.M, 16);
    if (mt > 4) return launch_rows<EPI>(p, units, s);
    if (mt <= 1) return launch_T<1, EPI>(p, T, units, s);
    if (mt == 2) return launch_T<2, EPI>(p, T, units, s);
    return launch_T<4, EPI>(p, T, units, s);
}

// tiles (or gate/up pairs) per workgroup: fill ~256 CUs with one round of workgroups where possible
int choose_T(int units) {
    static const int forced = [] { const char* e = getenv("PC_GEMM_T"); return e ? atoi(e) : 0; }();
    if (forced > 0) return forced;
    // smallest T in {1,2,3,4,8} whose grid fits one round of 256 CUs (a partial second round idles most of
    // the chip: 344 workgroups ran at 4.4 TB/s where 230 run the same bytes in one round)
    const int cand[5] = {1, 2, 3, 4, 8};
    for (int i = 0; i < 5; ++i)
        if (pc_ceil_div(units, cand[i]) <= 256) return cand[i];
    return 8;
}

}  // namespace

namespace {
// operands of the in-launch LLM.int8 outlier correction (pc_gemm_*_a8c)
struct A8Fused {
    const void* flags; const void* xraw; const void* cbt; int64_t ldt; const int32_t* row_perm;
};

int gemm_skinny_impl(const void* wf, const void* xf_hi, const void* xf_lo, const float* xn, const void* gamma, float eps,
                     int32_t M, int32_t N, int32_t K, int32_t epilogue, float* y, int64_t ldy, void* of_hi, void* of_lo,
                     int32_t kslices, void* stream, const float* wscale = nullptr, const float* xscale = nullptr,
                     const float* corr = nullptr, int64_t ldc = 0, const int32_t* corr_has = nullptr,
                     const A8Fused* fz = nullptr) {
    PC_REQUIRE(M > 0 && M <= kRowsMaxM, PC_ERR_ARG, "pc_gemm_skinny: M=%d outside 1..512 (use a dense GEMM above)", M);
    PC_REQUIRE(!xscale || fz || (wscale && corr && corr_has && ldc >= N && ldc % 4 == 0 && ((uintptr_t)corr & 15) == 0 && kslices == 1),
               PC_ERR_ARG, "pc_gemm_skinny_a8: int8 activations need int8 weights, x_scale, corr (16-byte aligned, ldc >= N), corr_has, no K-slices");
    PC_REQUIRE(!fz || (xscale && wscale && kslices == 1 && fz->flags && fz->xraw && fz->cbt && fz->ldt >= N && K <= 16384 &&
                       ((uintptr_t)fz->flags & 15) == 0), PC_ERR_ARG,
               "pc_gemm_skinny_a8c: the fused correction needs x_scale, w_scale, 16-byte aligned flags (>= 16384 bytes), x_raw, w_codes_t (ldt >= N), K <= 16384");
    PC_REQUIRE(N > 0 && N % 16 == 0 && K > 0 && K % 32 == 0, PC_ERR_ARG, "pc_gemm_skinny: need N%%16==0 and K%%32==0");
    PC_REQUIRE(wf && (xf_hi || xn), PC_ERR_ARG, "pc_gemm_skinny: null pointer");
    PC_REQUIRE(!xn || (gamma && M <= 16 && kslices == 1 && (epilogue == EPI_STORE || epilogue == EPI_SILU)), PC_ERR_ARG,
               "pc_gemm_skinny_norm: the fused-RMSNorm source needs M <= 16, no K-slicing, epilogue 0 or 2");
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.xn = xn; p.gamma = (const _Float16*)gamma; p.eps = eps;
    p.wf = (const _Float16*)wf; p.xf_hi = (const _Float16*)xf_hi; p.xf_lo = (const _Float16*)xf_lo;
    PC_REQUIRE(!wscale || (M <= 64 && (xn || xf_lo) && ((uintptr_t)wscale & 15) == 0 && K % 64 == 0), PC_ERR_ARG,
               "pc_gemm_skinny_w8: int8 weights need M <= 64, K %% 64 == 0, split-precision activations and 16-byte aligned scales");
    p.wscale = wscale; p.w8 = wscale ? 1 : 0;
    p.xscale = xscale; p.corr = corr; p.ldc = ldc; p.corr_has = corr_has;
    if (fz) {
        p.oflags = (const unsigned char*)fz->flags; p.xraw = (const _Float16*)fz->xraw; p.cbt = (const signed char*)fz->cbt;
        p.ldt = fz->ldt; p.row_perm = fz->row_perm;
    }
    p.y = y; p.ldy = ldy; p.of_hi = (_Float16*)of_hi; p.of_lo = (_Float16*)of_lo;
    p.M = M; p.ntiles = N / 16; p.KS = K / 32; p.npairs = 0; p.KSo = 0;
    PC_REQUIRE(kslices >= 1 && kslices <= 16 && (kslices == 1 || epilogue == EPI_STORE), PC_ERR_ARG,
               "pc_gemm_skinny: K-slicing (kslices=%d) is available for the plain-store epilogue only", kslices);
    p.kslices = kslices; p.slab_stride = (int64_t)M * ldy;
    hipStream_t s = (hipStream_t)stream;
    if (epilogue == EPI_SILU) {
        PC_REQUIRE(N % 64 == 0, PC_ERR_ARG, "pc_gemm_skinny: SiLU epilogue needs N = 2*inter with inter%%32==0");
        PC_REQUIRE(of_hi && of_lo, PC_ERR_ARG, "pc_gemm_skinny: SiLU epilogue needs output planes");
        p.npairs = N / 32;          // inter / 16
        p.KSo = (N / 2) / 32;       // k-steps of the consumer (down_proj, K = inter)
        return launch_MT<EPI_SILU>(p, choose_T(p.npairs), p.npairs, s);
    }
    if (epilogue == EPI_GELU) {
        PC_REQUIRE(N % 32 == 0 && of_hi && of_lo, PC_ERR_ARG, "pc_gemm_skinny: GELU epilogue needs N%%32==0 and output planes");
        p.KSo = N / 32;             // k-steps of the consumer (dense_4h_to_h, K = N)
        return launch_MT<EPI_GELU>(p, choose_T(p.ntiles), p.ntiles, s);
    }
    PC_REQUIRE(y && ldy >= N && ldy % 4 == 0, PC_ERR_ARG, "pc_gemm_skinny: bad output");
    if (epilogue == EPI_ADD) return launch_MT<EPI_ADD>(p, choose_T(p.ntiles), p.ntiles, s);
    PC_REQUIRE(epilogue == EPI_STORE, PC_ERR_ARG, "pc_gemm_skinny: unknown epilogue %d", epilogue);
    return launch_MT<EPI_STORE>(p, choose_T(p.ntiles * kslices) , p.ntiles, s);
}

int gemm_qkv_rope_impl(const void* wf_perm, const void* xf_hi, const void* xf_lo, const float* xn, const void* gamma,
                       float eps, int32_t M, int32_t K, const float* cs, void* q_hi, void* q_lo, int64_t q_token_stride,
                       void* k_arena, void* v_arena, int64_t arena_batch_stride, int64_t arena_head_stride, int32_t B,
                       int32_t H, int32_t Hkv, int32_t D, int32_t q_len, int32_t past_len, int32_t cap,
                       const int32_t* past_len_dev, void* k_lo, void* v_lo, int64_t lo_bs, int64_t lo_hs, void* stream,
                       const float* wscale = nullptr, int32_t lo_base = -1, const float* xscale = nullptr,
                       const float* corr = nullptr, int64_t ldc = 0, const int32_t* corr_has = nullptr, const A8Fused* fz = nullptr);
}  // namespace

PC_EXPORT int pc_gemm_skinny(const void* wf, const void* xf_hi, const void* xf_lo, int32_t M, int32_t N, int32_t K,
                             int32_t epilogue, float* y, int64_t ldy, void* of_hi, void* of_lo, int32_t kslices,
                             void* stream) {
    PC_REQUIRE(xf_hi, PC_ERR_ARG, "pc_gemm_skinny: null pointer");
    return gemm_skinny_impl(wf, xf_hi, xf_lo, nullptr, nullptr, 0.f, M, N, K, epilogue, y, ldy, of_hi, of_lo, kslices, stream);
}

PC_EXPORT int pc_gemm_skinny_norm(const void* wf, const float* x, const void* norm_weight, float eps, int32_t M, int32_t N,
                                  int32_t K, int32_t epilogue, float* y, int64_t ldy, void* of_hi, void* of_lo,
                                  void* stream) {
    PC_REQUIRE(x && norm_weight, PC_ERR_ARG, "pc_gemm_skinny_norm: null pointer");
    return gemm_skinny_impl(wf, nullptr, nullptr, x, norm_weight, eps, M, N, K, epilogue, y, ldy, of_hi, of_lo, 1, stream);
}

PC_EXPORT int pc_gemm_qkv_rope(const void* wf_perm, const void* xf_hi, const void* xf_lo, int32_t M, int32_t K,
                               const float* cs, void* q_hi, void* q_lo, int64_t q_token_stride, void* k_arena,
                               void* v_arena, int64_t arena_batch_stride, int64_t arena_head_stride, int32_t B,
                               int32_t H, int32_t Hkv, int32_t D, int32_t q_len, int32_t past_len, int32_t cap,
                               const int32_t* past_len_dev, void* k_lo, void* v_lo, int64_t lo_batch_stride,
                               int64_t lo_head_stride, void* stream) {
    PC_REQUIRE(xf_hi, PC_ERR_ARG, "pc_gemm_qkv_rope: null pointer");
    return gemm_qkv_rope_impl(wf_perm, xf_hi, xf_lo, nullptr, nullptr, 0.f, M, K, cs, q_hi, q_lo, q_token_stride, k_arena,
                              v_arena, arena_batch_stride, arena_head_stride, B, H, Hkv, D, q_len, past_len, cap,
                              past_len_dev, k_lo, v_lo, lo_batch_stride, lo_head_stride, stream);
}

PC_EXPORT int pc_gemm_qkv_rope_norm(const void* wf_perm, const float* x, const void* norm_weight, float eps, int32_t M,
                                    int32_t K, const float* cs, void* q_hi, void* q_lo, int64_t q_token_stride,
                                    void* k_arena, void* v_arena, int64_t arena_batch_stride, int64_t arena_head_stride,
                                    int32_t B, int32_t H, int32_t Hkv, int32_t D, int32_t q_len, int32_t past_len,
                                    int32_t cap, const int32_t* past_len_dev, void* k_lo, void* v_lo,
                                    int64_t lo_batch_stride, int64_t lo_head_stride, void* stream) {
    PC_REQUIRE(x && norm_weight && M <= 16, PC_ERR_ARG, "pc_gemm_qkv_rope_norm: needs x, the norm weight and M <= 16");
    return gemm_qkv_rope_impl(wf_perm, nullptr, nullptr, x, norm_weight, eps, M, K, cs, q_hi, q_lo, q_token_stride, k_arena,
                              v_arena, arena_batch_stride, arena_head_stride, B, H, Hkv, D, q_len, past_len, cap,
                              past_len_dev, k_lo, v_lo, lo_batch_stride, lo_head_stride, stream);
}

namespace {
int gemm_qkv_rope_impl(const void* wf_perm, const void* xf_hi, const void* xf_lo, const float* xn, const void* gamma,
                       float eps, int32_t M, int32_t K, const float* cs, void* q_hi, void* q_lo, int64_t q_token_stride,
                       void* k_arena, void* v_arena, int64_t arena_batch_stride, int64_t arena_head_stride, int32_t B,
                       int32_t H, int32_t Hkv, int32_t D, int32_t q_len, int32_t past_len, int32_t cap,
                       const int32_t* past_len_dev, void* k_lo, void* v_lo, int64_t lo_bs, int64_t lo_hs, void* stream,
                       const float* wscale, int32_t lo_base, const float* xscale, const float* corr, int64_t ldc,
                       const int32_t* corr_has, const A8Fused* fz) {
    const int N = (H + 2 * Hkv) * D;
    PC_REQUIRE(M > 0 && M <= kRowsMaxM && M == B * q_len, PC_ERR_ARG, "pc_gemm_qkv_rope: M=%d must equal B*q_len and be <= 512", M);
    PC_REQUIRE(D % 16 == 0 && K > 0 && K % 32 == 0 && H > 0 && Hkv > 0, PC_ERR_ARG, "pc_gemm_qkv_rope: bad shape");
    PC_REQUIRE(wf_perm && (xf_hi || xn) && cs && q_hi && q_lo && k_arena && v_arena, PC_ERR_ARG, "pc_gemm_qkv_rope: null pointer");
    PC_REQUIRE((int64_t)past_len + q_len <= cap, PC_ERR_BOUNDS,
               "pc_gemm_qkv_rope: past_len %d + q_len %d exceeds arena rows %d", past_len, q_len, cap);
    PC_REQUIRE(q_token_stride % 4 == 0 && arena_head_stride % 4 == 0, PC_ERR_ARG, "pc_gemm_qkv_rope: strides must keep 8-byte alignment");
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.wf = (const _Float16*)wf_perm; p.xf_hi = (const _Float16*)xf_hi; p.xf_lo = (const _Float16*)xf_lo;
    p.xn = xn; p.gamma = (const _Float16*)gamma; p.eps = eps;
    PC_REQUIRE(!wscale || (M <= 64 && (xn || xf_lo) && ((uintptr_t)wscale & 15) == 0 && K % 64 == 0), PC_ERR_ARG,
               "pc_gemm_qkv_rope_w8: int8 weights need M <= 64, K %% 64 == 0, split-precision activations and 16-byte aligned scales");
    p.wscale = wscale; p.w8 = wscale ? 1 : 0;
    PC_REQUIRE(!xscale || fz || (wscale && corr && corr_has && ldc >= N && ldc % 4 == 0 && ((uintptr_t)corr & 15) == 0), PC_ERR_ARG,
               "pc_gemm_qkv_rope_a8: int8 activations need int8 weights, x_scale, corr (16-byte aligned, ldc >= N) and corr_has");
    PC_REQUIRE(!fz || (xscale && wscale && fz->flags && fz->xraw && fz->cbt && fz->ldt >= N && K <= 16384 && ((uintptr_t)fz->flags & 15) == 0),
               PC_ERR_ARG, "pc_gemm_qkv_rope_a8c: the fused correction needs x_scale, w_scale, 16-byte aligned flags (>= 16384 bytes), x_raw, w_codes_t (ldt >= N), K <= 16384");
    p.xscale = xscale; p.corr = corr; p.ldc = ldc; p.corr_has = corr_has;
    if (fz) {
        p.oflags = (const unsigned char*)fz->flags; p.xraw = (const _Float16*)fz->xraw; p.cbt = (const signed char*)fz->cbt;
        p.ldt = fz->ldt; p.row_perm = fz->row_perm;
    }
    p.y = nullptr; p.ldy = 0; p.of_hi = nullptr; p.of_lo = nullptr; p.KSo = 0;
    p.M = M; p.ntiles = N / 16; p.KS = K / 32; p.npairs = 0; p.kslices = 1; p.slab_stride = 0;
    p.rope.cs = (const float2*)cs; p.rope.q_hi = (_Float16*)q_hi; p.rope.q_lo = (_Float16*)q_lo; p.rope.q_ts = q_token_stride;
    p.rope.k_arena = (_Float16*)k_arena; p.rope.v_arena = (_Float16*)v_arena; p.rope.a_bs = arena_batch_stride;
    p.rope.a_hs = arena_head_stride; p.rope.past_len_dev = past_len_dev;
    PC_REQUIRE((k_lo == nullptr) == (v_lo == nullptr) && (!k_lo || lo_hs % 4 == 0), PC_ERR_ARG,
               "pc_gemm_qkv_rope: k_lo / v_lo go together, strides must keep 8-byte alignment");
    p.rope.k_lo = (_Float16*)k_lo; p.rope.v_lo = (_Float16*)v_lo; p.rope.lo_bs = lo_bs; p.rope.lo_hs = lo_hs;
    PC_REQUIRE(lo_base >= -2 && (lo_base != -2 || past_len_dev) && (lo_base < 0 || lo_base <= past_len), PC_ERR_ARG,
               "pc_gemm_qkv_rope: lo_base must be -1 (pass-relative rows), -2 (past_len_dev[1]) or lie in [0, past_len]");
    p.rope.lo_base = lo_base;
    p.rope.H = H; p.rope.Hkv = Hkv; p.rope.D = D; p.rope.q_len = q_len; p.rope.past_len = past_len;
    return launch_MT<EPI_ROPE>(p, choose_T(p.ntiles), p.ntiles, (hipStream_t)stream);
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

PC_EXPORT int pc_rmsnorm_frag(float* x, const void* weight, void* xf_hi, void* xf_lo, int32_t rows,
                              int32_t hidden, float eps, const float* slabs, int32_t nslabs, void* stream) {
    PC_REQUIRE(rows > 0 && rows <= kRowsMaxM && hidden > 0 && hidden % 32 == 0, PC_ERR_ARG, "pc_rmsnorm_frag: bad sizes");
    PC_REQUIRE(x && weight && xf_hi && xf_lo && nslabs >= 0 && (nslabs == 0 || slabs), PC_ERR_ARG,
               "pc_rmsnorm_frag: null pointer");
    PC_REQUIRE(hidden <= 16384, PC_ERR_ARG, "pc_rmsnorm_frag: hidden %d > 16384", hidden);
    const int groups = pc_ceil_div(hidden / 8, 256);
#define PC_RMS(GV)                                                                                                   \
    hipLaunchKernelGGL(rmsnorm_frag_kernel<GV>, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, (const _Float16*)weight, \
                       (_Float16*)xf_hi, (_Float16*)xf_lo, hidden, eps, slabs, nslabs, (int64_t)rows * hidden,       \
                       (const _Float16*)nullptr)
    if (groups <= 1) PC_RMS(1); else if (groups <= 2) PC_RMS(2); else if (groups <= 4) PC_RMS(4); else PC_RMS(8);
#undef PC_RMS
    return pc_check_launch("rmsnorm_frag_kernel");
}

PC_EXPORT int pc_layernorm_frag(float* x, const void* weight, const void* bias, void* xf_hi, void* xf_lo, int32_t rows,
                                int32_t hidden, float eps, const float* slabs, int32_t nslabs, void* stream) {
    PC_REQUIRE(rows > 0 && rows <= kRowsMaxM && hidden > 0 && hidden % 32 == 0 && hidden <= 16384, PC_ERR_ARG,
               "pc_layernorm_frag: bad sizes");
    PC_REQUIRE(x && weight && xf_hi && xf_lo && nslabs >= 0 && (nslabs == 0 || slabs), PC_ERR_ARG,
               "pc_layernorm_frag: null pointer");   /* bias may be NULL */
    const int groups = pc_ceil_div(hidden / 8, 256);
#define PC_LN(GV)                                                                                                    \
    hipLaunchKernelGGL((rmsnorm_frag_kernel<GV, true>), dim3(rows), dim3(256), 0, (hipStream_t)stream, x,            \
                       (const _Float16*)weight, (_Float16*)xf_hi, (_Float16*)xf_lo, hidden, eps, slabs, nslabs,      \
                       (int64_t)rows * hidden, (const _Float16*)bias)
    if (groups <= 1) PC_LN(1); else if (groups <= 2) PC_LN(2); else if (groups <= 4) PC_LN(4); else PC_LN(8);
#undef PC_LN
    return pc_check_launch("layernorm_frag_kernel");
}

// ---- int8 weights (the reference's GPU configs load the model with load_in_8bit) ----------------
// Weight-only int8: wf8 is the fragment image [N/16][K/64][64][16] of OFFSET-BINARY bytes -- a lane's 16 bytes are its 8
// values of k-step 2s followed by those of k-step 2s + 1 -- (q + 128, q = round(w / scale)
// per output row, scale = absmax / 127), w_scale[N] fp32 in the row order of the image.  Activations stay split-precision
// fp16 pairs and accumulation fp32, so y = scale[n] * sum_k q[n][k] * x[k] exactly as an fp32 GEMM over the dequantised
// weights would give it -- half the weight bytes per launch.  M <= 64 (the weight-streaming regime proper).
PC_EXPORT int pc_gemm_skinny_w8(const void* wf8, const float* w_scale, const void* xf_hi, const void* xf_lo, int32_t M,
                                int32_t N, int32_t K, int32_t epilogue, float* y, int64_t ldy, void* of_hi, void* of_lo,
                                int32_t kslices, void* stream) {
    PC_REQUIRE(xf_hi && xf_lo && w_scale, PC_ERR_ARG, "pc_gemm_skinny_w8: null pointer");
    return gemm_skinny_impl(wf8, xf_hi, xf_lo, nullptr, nullptr, 0.f, M, N, K, epilogue, y, ldy, of_hi, of_lo, kslices, stream,
                            w_scale);
}

PC_EXPORT int pc_gemm_skinny_norm_w8(const void* wf8, const float* w_scale, const float* x, const void* norm_weight, float eps,
                                     int32_t M, int32_t N, int32_t K, int32_t epilogue, float* y, int64_t ldy, void* of_hi,
                                     void* of_lo, void* stream) {
    PC_REQUIRE(x && norm_weight && w_scale, PC_ERR_ARG, "pc_gemm_skinny_norm_w8: null pointer");
    return gemm_skinny_impl(wf8, nullptr, nullptr, x, norm_weight, eps, M, N, K, epilogue, y, ldy, of_hi, of_lo, 1, stream,
                            w_scale);
}

PC_EXPORT int pc_gemm_qkv_rope_w8(const void* wf8_perm, const float* w_scale_perm, const void* xf_hi, const void* xf_lo,
                                  const float* x, const void* norm_weight, float eps, int32_t M, int32_t K, const float* cs,
                                  void* q_hi, void* q_lo, int64_t q_token_stride, void* k_arena, void* v_arena,
                                  int64_t arena_batch_stride, int64_t arena_head_stride, int32_t B, int32_t H, int32_t Hkv,
                                  int32_t D, int32_t q_len, int32_t past_len, int32_t cap, const int32_t* past_len_dev,
                                  void* k_lo, void* v_lo, int64_t lo_batch_stride, int64_t lo_head_stride, void* stream) {
    PC_REQUIRE(w_scale_perm && ((xf_hi && xf_lo && !x) || (x && norm_weight && !xf_hi && M <= 16)), PC_ERR_ARG,
               "pc_gemm_qkv_rope_w8: pass either the activation planes or (x, norm_weight) with M <= 16");
    return gemm_qkv_rope_impl(wf8_perm, xf_hi, xf_lo, x, norm_weight, eps, M, K, cs, q_hi, q_lo, q_token_stride, k_arena,
                              v_arena, arena_batch_stride, arena_head_stride, B, H, Hkv, D, q_len, past_len, cap,
                              past_len_dev, k_lo, v_lo, lo_batch_stride, lo_head_stride, stream, w_scale_perm);
}

// pc_gemm_qkv_rope_ex: the union of the q|k|v entry points (fp16 or int8 weights: w_scale_perm NULL or not; activation
// planes or the fused-RMSNorm source) plus lo_base, which places the residual rows in a buffer that outlives the pass:
// row of token tt = tt (-1), past_len + tt - lo_base (>= 0) or past_len + tt - past_len_dev[1] (-2, decode under a hipGraph).
PC_EXPORT int pc_gemm_qkv_rope_ex(const void* wf_perm, const float* w_scale_perm, const void* xf_hi, const void* xf_lo,
                                  const float* x, const void* norm_weight, float eps, int32_t M, int32_t K, const float* cs,
                                  void* q_hi, void* q_lo, int64_t q_token_stride, void* k_arena, void* v_arena,
                                  int64_t arena_batch_stride, int64_t arena_head_stride, int32_t B, int32_t H, int32_t Hkv,
                                  int32_t D, int32_t q_len, int32_t past_len, int32_t cap, const int32_t* past_len_dev,
                                  void* k_lo, void* v_lo, int64_t lo_batch_stride, int64_t lo_head_stride, int32_t lo_base,
                                  void* stream) {
    PC_REQUIRE((xf_hi && !x) || (x && norm_weight && !xf_hi && M <= 16), PC_ERR_ARG,
               "pc_gemm_qkv_rope_ex: pass either the activation planes or (x, norm_weight) with M <= 16");
    return gemm_qkv_rope_impl(wf_perm, xf_hi, xf_lo, x, norm_weight, eps, M, K, cs, q_hi, q_lo, q_token_stride, k_arena,
                              v_arena, arena_batch_stride, arena_head_stride, B, H, Hkv, D, q_len, past_len, cap,
                              past_len_dev, k_lo, v_lo, lo_batch_stride, lo_head_stride, stream, w_scale_perm, lo_base);
}

// ---- LLM.int8 (int8 weights AND int8 activations, pc_int8.hip) -------------------------------------------------------
// xq_hi holds the activation CODES of pc_quant_act_i8 as an fp16 fragment plane (|code| <= 127: exact), xq_lo a plane of zeros;
// y = (sum_k code_w[n][k] * code_x[m][k]) * w_scale[n] * x_scale[m]  (+ corr[m][n] when *corr_has) through the epilogue.
// The integer dot product is accumulated in fp32 by the fp16 MFMAs: exact below 2^24 per accumulator.
PC_EXPORT int pc_gemm_skinny_a8(const void* wf8, const float* w_scale, const void* xq_hi, const void* xq_lo, const float* x_scale,
                                const float* corr, int64_t ldc, const int32_t* corr_has, int32_t M, int32_t N, int32_t K,
                                int32_t epilogue, float* y, int64_t ldy, void* of_hi, void* of_lo, void* stream) {
    PC_REQUIRE(xq_hi && xq_lo && w_scale && x_scale, PC_ERR_ARG, "pc_gemm_skinny_a8: null pointer");
    return gemm_skinny_impl(wf8, xq_hi, xq_lo, nullptr, nullptr, 0.f, M, N, K, epilogue, y, ldy, of_hi, of_lo, 1, stream, w_scale,
                            x_scale, corr, ldc, corr_has);
}

PC_EXPORT int pc_gemm_qkv_rope_a8(const void* wf8_perm, const float* w_scale_perm, const void* xq_hi, const void* xq_lo,
                                  const float* x_scale, const float* corr, int64_t ldc, const int32_t* corr_has, int32_t M,
                                  int32_t K, const float* cs, void* q_hi, void* q_lo, int64_t q_token_stride, void* k_arena,
                                  void* v_arena, int64_t arena_batch_stride, int64_t arena_head_stride, int32_t B, int32_t H,
                                  int32_t Hkv, int32_t D, int32_t q_len, int32_t past_len, int32_t cap,
                                  const int32_t* past_len_dev, void* k_lo, void* v_lo, int64_t lo_batch_stride,
                                  int64_t lo_head_stride, int32_t lo_base, void* stream) {
    PC_REQUIRE(xq_hi && xq_lo && w_scale_perm && x_scale, PC_ERR_ARG, "pc_gemm_qkv_rope_a8: null pointer");
    return gemm_qkv_rope_impl(wf8_perm, xq_hi, xq_lo, nullptr, nullptr, 0.f, M, K, cs, q_hi, q_lo, q_token_stride, k_arena,
                              v_arena, arena_batch_stride, arena_head_stride, B, H, Hkv, D, q_len, past_len, cap,
                              past_len_dev, k_lo, v_lo, lo_batch_stride, lo_head_stride, stream, w_scale_perm, lo_base,
                              x_scale, corr, ldc, corr_has);
}

// pc_gemm_skinny_a8 / pc_gemm_qkv_rope_a8 with the outlier correction computed INSIDE the launch (M <= 64): instead of corr /
// corr_has the call takes what pc_outlier_corr would have read -- the flag bytes of pc_quant_act_i8 (a buffer of >= 16384 bytes,
// zero behind K), the fp16 activations x_raw (fragment plane), the transposed int8 weight codes [K][ldt] (original row order) and,
// for q|k|v, the image-row -> original-row permutation.  Same result up to the fp32 summation order over the outlier columns.
PC_EXPORT int pc_gemm_skinny_a8c(const void* wf8, const float* w_scale, const void* xq_hi, const void* xq_lo, const float* x_scale,
                                 const void* flags, const void* x_raw, const void* w_codes_t, int64_t ldt, int32_t M, int32_t N,
                                 int32_t K, int32_t epilogue, float* y, int64_t ldy, void* of_hi, void* of_lo, void* stream) {
    PC_REQUIRE(xq_hi && xq_lo && w_scale && x_scale, PC_ERR_ARG, "pc_gemm_skinny_a8c: null pointer");
    const A8Fused fz = {flags, x_raw, w_codes_t, ldt, nullptr};
    return gemm_skinny_impl(wf8, xq_hi, xq_lo, nullptr, nullptr, 0.f, M, N, K, epilogue, y, ldy, of_hi, of_lo, 1, stream, w_scale,
                            x_scale, nullptr, 0, nullptr, &fz);
}

PC_EXPORT int pc_gemm_qkv_rope_a8c(const void* wf8_perm, const float* w_scale_perm, const void* xq_hi, const void* xq_lo,
                                   const float* x_scale, const void* flags, const void* x_raw, const void* w_codes_t, int64_t ldt,
                                   const int32_t* row_perm, int32_t M, int32_t K, const float* cs, void* q_hi, void* q_lo,
                                   int64_t q_token_stride, void* k_arena, void* v_arena, int64_t arena_batch_stride,
                                   int64_t arena_head_stride, int32_t B, int32_t H, int32_t Hkv, int32_t D, int32_t q_len,
                                   int32_t past_len, int32_t cap, const int32_t* past_len_dev, void* k_lo, void* v_lo,
                                   int64_t lo_batch_stride, int64_t lo_head_stride, int32_t lo_base, void* stream) {
    PC_REQUIRE(xq_hi && xq_lo && w_scale_perm && x_scale && row_perm, PC_ERR_ARG, "pc_gemm_qkv_rope_a8c: null pointer");
    const A8Fused fz = {flags, x_raw, w_codes_t, ldt, row_perm};
    return gemm_qkv_rope_impl(wf8_perm, xq_hi, xq_lo, nullptr, nullptr, 0.f, M, K, cs, q_hi, q_lo, q_token_stride, k_arena,
                              v_arena, arena_batch_stride, arena_head_stride, B, H, Hkv, D, q_len, past_len, cap,
                              past_len_dev, k_lo, v_lo, lo_batch_stride, lo_head_stride, stream, w_scale_perm, lo_base,
                              x_scale, nullptr, 0, nullptr, &fz);
}
