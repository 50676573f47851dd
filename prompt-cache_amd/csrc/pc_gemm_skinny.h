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
// This header holds the kernel templates; the translation units pc_gemm_mt{1,2,4}.hip instantiate the launch shapes of one
// row regime each (they compile in parallel), pc_gemm.hip holds the C-ABI entry points.
#pragma once
#include <hip/hip_fp16.h>
#include <string.h>

#include "pc_common.h"

namespace pcg {


typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWaves = 8;
constexpr int kThreads = kWaves * 64;
constexpr int kRowsMaxM = 512;              // most rows any weight-streaming launch takes (gemm_rows_kernel, pc_gemm_rows.hip)
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
    const int32_t* m_dev;    // optional device word: rows that really carry tokens (<= M): pad rows behind it are not loaded
    int32_t kslices; int64_t slab_stride;   // EPI_STORE only: grid.y K-slices, slice s writes y + s*slab_stride
    int32_t zrows;           // gemm_rows_kernel: rows per grid.z block (a multiple of 16; 0: one block takes all M rows)
    // NORM activation source (M <= 16): the fp32 residual stream itself; RMSNorm is folded into the launch
    const float* xn; const _Float16* gamma; float eps;
    // int8 weights (W8 kernels, M <= 64): wf is the fragment image of OFFSET-BINARY bytes (q + 128), 8 B per lane per
    // k-step, wscale[n] the fp32 scale of output feature n (row order of the image); y = scale * sum_k q[n][k] x[k]
    const float* wscale; int32_t w8;
    // LLM.int8 activations (pc_int8.hip): the activation planes hold int8 CODES, xscale[m] = SCA[m] / 127 rescales row m, and
    // corr[m][n] (row stride ldc, output-feature index in the image's row order) is added before the epilogue's
    // nonlinearity when *corr_has != 0 (the fp16 outlier part of the decomposition)
    const float* xscale; const float* corr; int64_t ldc; const int32_t* corr_has;
    const signed char* xq8;            // (optional) the codes as the int8 MFMA's operand image, [M/16][KS/2][64][16] (pc_quant_act_i8 codes8)
    // ... or the correction is computed INSIDE the launch (pc_gemm_*_a8c): oflags = the K outlier-column flag bytes of
    // pc_quant_act_i8 (buffer of >= 16384 bytes, zero behind K), xraw = the fp16 activations (fragment plane, same layout as the
    // code plane xf_hi), cbt = the int8 weight codes transposed [K][ldt] in ORIGINAL row order, row_perm = image row -> original
    // row (q|k|v's rotary permutation) or null
    const unsigned char* oflags; const _Float16* xraw; const signed char* cbt; int64_t ldt; const int32_t* row_perm;
    // EPI_SILU as the PRODUCER of an LLM.int8 projection input quantised inside its consumer (pc_gemm_q8.hip): per output tile
    // and row the largest |fp16(silu(g) * u)| below the outlier threshold, pmax_out[unit][16], and one flag byte per feature
    // holding an entry at or above it (set-only; the buffer is zero when the launch starts)
    float* pmax_out; unsigned char* oflags_out; float thr_out;
    RopeEpi rope;
    // K split inside a workgroup: the four waves dispatched first (one per SIMD, the OLDER wave of each SIMD) win the arbitration
    // for the CU's memory pipe against their younger SIMD partners and finish an equal share ~30 % earlier; the younger half
    // then streams on alone at half the memory parallelism (tools/gemm_trace.py: down_proj 11.8 vs 17.6 us).  kskew / 64 of an
    // equal share moves from each younger wave to an older one (static: the summation order stays fixed).  prio_alt: the two
    // halves swap s_setprio every k-block instead (fair on average, no tuning).
    int32_t kskew, prio_alt;
    // dev: per-wave wall-clock stamps [grid.x * grid.y][kWaves][4] (entry, K loop done, reduced, done); pc_dev_gemm_trace
    unsigned long long* trace;
};

__device__ __forceinline__ void trace_stamp(const GemmParams& p, int bx, int by, int wave, int slot) {
    if (p.trace && (threadIdx.x & 63) == 0)
        p.trace[(((int64_t)by * gridDim.x + bx) * kWaves + wave) * 4 + slot] = wall_clock64();
}

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

// LLM.int8 activations on the int8 MFMA: the codes plane holds integers in [-127, 127] as fp16 values (pc_quant_act_i8); the
// eight of k-step 2s and the eight of 2s + 1 of a lane become the 16 signed bytes of one v_mfma_i32_16x16x64_i8 operand, in the
// byte order of the weight image's 16 bytes (so the lanes of the two operands agree on which k sits where; the sum over k does
// not care which).  Exactly: the code goes under the exponent of 1024.0 as an offset-binary byte (code + 1152 = 0x6400 | (code +
// 128)), v_perm_b32 collects the low bytes, the XOR turns offset binary into two's complement.  6 VALU ops per k-step of a row
// tile -- once per ROW tile; the weight fragments, which are what cost eight conversions per tile and k-step on the fp16 MFMA,
// need one XOR per dword.
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x2 pack_codes8(h8 c) {
    const h2v bias = {(_Float16)1152.0f, (_Float16)1152.0f};
    const h2v p0 = h2v{c[0], c[1]} + bias, p1 = h2v{c[2], c[3]} + bias, p2 = h2v{c[4], c[5]} + bias, p3 = h2v{c[6], c[7]} + bias;
    u32x2 r;
    r[0] = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, p1), __builtin_bit_cast(uint32_t, p0), 0x06040200u) ^ 0x80808080u;
    r[1] = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, p3), __builtin_bit_cast(uint32_t, p2), 0x06040200u) ^ 0x80808080u;
    return r;
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
                                        f4 (&acc)[MT][TT], const h8 (*wpre)[TT] = nullptr, const signed char* x8_base = nullptr) {
    static_assert(!PRE || (!TAIL && !W8), "prefetched first blocks are full fp16 blocks");
    // W8: the image holds k-step PAIRS -- a lane's 16 bytes are its 8 values of k-step 2s and of 2s + 1 -- so one
    // global_load_dwordx4 feeds four MFMAs; ks, nvalid and U are even (the K ranges are cut on pair boundaries).  The
    // raw bytes wait in registers (half of what fp16 fragments take) and are converted right before their MFMAs.
    constexpr int NW = W8 ? U / 2 : U;
    static_assert(!W8 || U % 2 == 0, "int8 weights come in k-step pairs");
    constexpr bool A8 = W8 && !TWO;                      // LLM.int8 codes on both sides
    h8 w[W8 ? 1 : U][TT], xh[U][MT], xl[U][MT];
    u32x4 raw[W8 ? NW : 1][TT];
    [[maybe_unused]] u32x4 xq[A8 ? NW : 1][MT];          // x8_base: the activations' int8 operand image, one load per k-step pair
    const bool img = A8 && x8_base != nullptr;           // (wave-uniform)
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
        if constexpr (A8) {
#pragma unroll
            for (int u = 0; u < NW; ++u) xq[u][a] = u32x4{0u, 0u, 0u, 0u};
            if (img) {
                if (row_ok[a]) {
#pragma unroll
                    for (int u = 0; u < NW; ++u) {
                        const int uu = (TAIL && 2 * u >= nvalid) ? nvalid / 2 - 1 : u;
                        xq[u][a] = *(const u32x4*)(x8_base + ((int64_t)a * (KS >> 1) + (ks >> 1) + uu) * 1024);
                    }
                }
                continue;
            }
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
    if constexpr (A8) {
        // int8 codes on both sides (LLM.int8; launch_w8 instantiates TWO = false for x_scale launches only): one
        // v_mfma_i32_16x16x64_i8 per weight tile and k-step PAIR, int32 accumulators kept in acc's registers (the caller converts
        // them once, behind the K loop) -- the arithmetic Linear8bitLt's igemmlt does, exact for any K
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            if (TAIL && 2 * u >= nvalid) continue;        // (wave-uniform) the re-read pair contributes nothing
            i32x4 xa[MT];
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                if (img) {
                    xa[a] = i32x4{(int)xq[u][a][0], (int)xq[u][a][1], (int)xq[u][a][2], (int)xq[u][a][3]};
                } else {
                    const u32x2 lo = pack_codes8(xh[2 * u][a]), hi = pack_codes8(xh[2 * u + 1][a]);
                    xa[a] = i32x4{(int)lo[0], (int)lo[1], (int)hi[0], (int)hi[1]};
                }
            }
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                const u32x4 wq = raw[u][t] ^ 0x80808080u;
                const i32x4 wv = {(int)wq[0], (int)wq[1], (int)wq[2], (int)wq[3]};
#pragma unroll
                for (int a = 0; a < MT; ++a)
                    acc[a][t] = __builtin_bit_cast(f4, __builtin_amdgcn_mfma_i32_16x16x64_i8(wv, xa[a], __builtin_bit_cast(i32x4, acc[a][t]), 0, 0, 0));
            }
        }
        return;
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
// The RMSNorm gain of a wave's K range, staged ONCE per wave in LDS: a per-k-step global load of it was one of six (q|k|v)
// / nine (gate|up) vector-memory instructions per k-step of a workgroup -- and what bounds these launches is the rate at which
// a CU's vector memory pipe takes requests (~35-45 GB/s per CU, L2 hits included), not HBM itself.  The loads are issued in
// front of the first k-block's weight loads and committed to the wave-private LDS slice after that block's loads are all in
// flight (no workgroup barrier: the slice is read by the wave that wrote it); k-steps then read it with ds_read_b128 (four
// distinct addresses per instruction, one per 16-lane group: broadcast, conflict-free).
constexpr int kGamSteps = 64;                      // k-steps of gain per wave the LDS slice holds (K <= 16384)
constexpr int kGamHalfs = kGamSteps * 32;
constexpr int kGamJ = kGamHalfs / 512;            // 1-KiB wave-loads to fill it
struct GammaStage {
    h8 pre[kGamJ];
    _Float16* dst;                                 // this wave's LDS slice
    int nh, lane;                                  // halfs of gain in the wave's K range
    bool done;
    __device__ __forceinline__ void issue(const _Float16* src, _Float16* lds_slice, int nhalfs, int ln) {
        dst = lds_slice; nh = nhalfs; lane = ln; done = false;
#pragma unroll
        for (int j = 0; j < kGamJ; ++j) {
            h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            pre[j] = z;
            if (j * 512 < nh) {                    // (wave-uniform)
                const int idx = j * 512 + lane * 8;
                pre[j] = *(const h8*)(src + (idx < nh ? idx : nh - 8));
            }
        }
    }
    __device__ __forceinline__ void commit() {
        if (done) return;
        done = true;
#pragma unroll
        for (int j = 0; j < kGamJ; ++j)
            if (j * 512 < nh) {
                const int idx = j * 512 + lane * 8;
                *(h8*)(dst + (idx < nh ? idx : nh - 8)) = pre[j];
            }
    }
};

template <int MT, int TT, int U, bool TAIL, bool W8 = false, bool PRE = false>
__device__ __forceinline__ void k_block_norm(const _Float16* const (&wbase)[TT], const float* const (&xrow)[MT], const _Float16* gam,
                                             int ks, int nvalid, const bool (&row_ok)[MT], f4 (&acc)[MT][TT], float (&ss)[MT],
                                             GammaStage& gst, const h8 (*wpre)[TT] = nullptr) {
    static_assert(!PRE || (!TAIL && !W8), "prefetched first blocks are full fp16 blocks");
    constexpr int NW = W8 ? U / 2 : U;                   // W8: k-step pairs per 16-byte load, see k_block
    static_assert(!W8 || U % 2 == 0, "int8 weights come in k-step pairs");
    h8 w[W8 ? 1 : U][TT], gw[U];
    u32x4 raw[W8 ? NW : 1][TT];
    f4 xa[MT][U][2];
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
    for (int a = 0; a < MT; ++a) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            f4 z = {0.f, 0.f, 0.f, 0.f};
            xa[a][u][0] = z; xa[a][u][1] = z;
        }
        if (row_ok[a]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int uu = (TAIL && u >= nvalid) ? nvalid - 1 : u;
                xa[a][u][0] = *(const f4*)(xrow[a] + (ks + uu) * 32);
                xa[a][u][1] = *(const f4*)(xrow[a] + (ks + uu) * 32 + 4);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    gst.commit();                                  // (first block of the wave only) the gain slice goes to LDS
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int uu = (TAIL && u >= nvalid) ? nvalid - 1 : u;
        gw[u] = *(const h8*)(gam + (ks + uu) * 32);    // gam: the wave's LDS slice, rebased to absolute k-steps
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (TAIL && u >= nvalid) continue;              // the re-read k-step contributes nothing
        h8 hi[MT], lo[MT];
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xv = e < 4 ? xa[a][u][0][e] : xa[a][u][1][e - 4];
                ss[a] += xv * xv;
                const float v = xv * (float)gw[u][e];
                _Float16 vh, vl;
                pc_split(v, vh, vl);
                hi[a][e] = vh; lo[a][e] = vl;
            }
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            h8 wv;
            if constexpr (W8) wv = (u & 1) ? cvt_w8(raw[u >> 1][t][2], raw[u >> 1][t][3]) : cvt_w8(raw[u >> 1][t][0], raw[u >> 1][t][1]);
            else wv = w[u][t];
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                acc[a][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, hi[a], acc[a][t], 0, 0, 0);
                acc[a][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, lo[a], acc[a][t], 0, 0, 0);
            }
        }
    }
}

// Epilogue of one reduced 16 x 16 tile.  v (and u = the "up" tile for EPI_SILU) follow the MFMA C/D map: this lane
// holds output row `row` (token) and features unit*16 + 4*g .. +3.  Must be called by all 64 lanes of a wave
// (EPI_ROPE exchanges rotary partners across lanes).
template <int EPI>
__device__ __forceinline__ void tile_epilogue(const GemmParams& p, f4 v, f4 u, int row, int unit, int g, int slice,
                                              bool fused_corr = false, f4 fused_cv = f4{0.f, 0.f, 0.f, 0.f},
                                              f4 fused_cu = f4{0.f, 0.f, 0.f, 0.f}, bool have_old = false,
                                              f4 old_pre = f4{0.f, 0.f, 0.f, 0.f}, bool have_xs = false, float xs_in = 0.f) {
    const int nunits = (EPI == EPI_SILU) ? p.npairs : p.ntiles;
    if (p.wscale && unit < nunits) {        // int8 weights: per-output-feature scale (linear, so K-sliced partials scale too)
        const f4 sv = *(const f4*)(p.wscale + unit * 16 + g * 4);
        v[0] *= sv[0]; v[1] *= sv[1]; v[2] *= sv[2]; v[3] *= sv[3];
        if (EPI == EPI_SILU) {
            const f4 su = *(const f4*)(p.wscale + (p.npairs + unit) * 16 + g * 4);
            u[0] *= su[0]; u[1] *= su[1]; u[2] *= su[2]; u[3] *= su[3];
        }
    }
    if ((p.xscale || have_xs) && unit < nunits && row < p.M) {
        const float xs = have_xs ? xs_in : p.xscale[row];      // (have_xs: the row scale was computed inside this launch)
        v[0] *= xs; v[1] *= xs; v[2] *= xs; v[3] *= xs;
        if (EPI == EPI_SILU) { u[0] *= xs; u[1] *= xs; u[2] *= xs; u[3] *= xs; }
        if (fused_corr) {                   // the correction was accumulated inside this launch
            v[0] += fused_cv[0]; v[1] += fused_cv[1]; v[2] += fused_cv[2]; v[3] += fused_cv[3];
            if (EPI == EPI_SILU) { u[0] += fused_cu[0]; u[1] += fused_cu[1]; u[2] += fused_cu[2]; u[3] += fused_cu[3]; }
        } else if (p.corr_has && *p.corr_has) {
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
            if (p.of_lo) *(h4*)(p.of_lo + off) = lo;
            if (p.pmax_out) {                   // (the four lanes of a row enter together: same row, same unit)
                float mxl = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float f = fabsf((float)hi[r]);
                    if (p.thr_out > 0.f && f >= p.thr_out) p.oflags_out[j0 + r] = 1;
                    else mxl = fmaxf(mxl, f);
                }
                mxl = fmaxf(mxl, __shfl_xor(mxl, 16));
                mxl = fmaxf(mxl, __shfl_xor(mxl, 32));
                if (g == 0) p.pmax_out[unit * 16 + row] = mxl;
            }
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
                // (have_old: the residual tile was fetched at kernel entry -- no dependent load behind the reduction)
                const f4 old = have_old ? old_pre : *(const f4*)yp;
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

// write-through stores / L1-bypassing loads of the in-launch K reduction's partial tiles (pc_gemm_ks.hip, pc_gemm_part.hip)
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) uint32_t gu32;
__device__ __forceinline__ void st_wt2(float* p, float a, float b) {
    const unsigned long long x = ((unsigned long long)__float_as_uint(b) << 32) | __float_as_uint(a);
    __hip_atomic_store((gu64*)p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float2 ld_wt2(const float* p) {
    const unsigned long long x = __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((uint32_t)x), __uint_as_float((uint32_t)(x >> 32)));
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
    if (p.kskew > 0) {
        // older half (waves 0..3): (64 + kskew) / 64 of an equal share each; the younger half divides the rest
        const int n = kq1 - kq0;
        int ko = (n * (64 + p.kskew) + 8 * 64 - 1) / (8 * 64);
        if (W8) ko = (ko + 1) & ~1;
        int rest = n - 4 * ko;
        if (rest < 0) rest = 0;
        int ky = (rest + 3) / 4;
        if (W8) ky = (ky + 1) & ~1;
        ks0 = wave < 4 ? kq0 + wave * ko : kq0 + 4 * ko + (wave - 4) * ky;
        const int len = wave < 4 ? ko : ky;
        if (ks0 > kq1) ks0 = kq1;
        ks1 = (ks0 + len < kq1) ? ks0 + len : kq1;
        return;
    }
    ks0 = kq0 + wave * ksw;
    ks1 = (ks0 + ksw < kq1) ? ks0 + ksw : kq1;
}

// (prio_alt) the halves of the workgroup take turns at s_setprio 1, one k-block each
__device__ __forceinline__ void alt_prio(const GemmParams& p, int wave, int& blk) {
    if (p.prio_alt) {
        if (((blk++) ^ (wave >> 2)) & 1) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
    }
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
                                                 float* red_raw, float (*ssl)[32],
                                                 const h8 (*wpre)[(EPI == EPI_SILU) ? 2 * T : T] = nullptr,
                                                 AfterK after_k = AfterK(), _Float16* gam_lds = nullptr) {
    static_assert(!NORM || (MT <= 2 && TWO), "the fused-RMSNorm source is for one or two row tiles");
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
    trace_stamp(p, bx, by, wave, 0);
    int ks0, ks1;
    wave_k_range<W8>(p, by, wave, ks0, ks1);
    int tile[TT];
    wg_tiles<T, EPI>(p, bx, tile);
    // EPI_ADD, one reduction round: the residual tile wave w will add to is fetched NOW (clamped, unconditional), so the
    // epilogue does not end on a dependent load -> add -> store chain (the tile is written by this lane only)
    constexpr bool kPreY = (EPI == EPI_ADD) && (MT * T <= IPR) && !W8;
    [[maybe_unused]] f4 yold = {0.f, 0.f, 0.f, 0.f};
    if constexpr (kPreY) {
        const int item = wave < MT * T ? wave : MT * T - 1;
        const int a = item / T, t = item - a * T;
        const int unit = bx * T + t < p.ntiles ? bx * T + t : p.ntiles - 1;
        const int row = a * 16 + m < p.M ? a * 16 + m : p.M - 1;
        yold = *(const f4*)(p.y + (int64_t)row * p.ldy + unit * 16 + g * 4);
    }

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
    [[maybe_unused]] const signed char* x8_base = (W8 && !TWO && p.xq8) ? p.xq8 + lane * 16 : nullptr;

    bool row_ok[MT];
    const int rows_live = p.m_dev ? (*p.m_dev < p.M ? *p.m_dev : p.M) : p.M;
#pragma unroll
    for (int a = 0; a < MT; ++a) row_ok[a] = a * 16 + m < rows_live;
    int ks = ks0;
    [[maybe_unused]] float ss[MT];
#pragma unroll
    for (int a = 0; a < MT; ++a) ss[a] = 0.f;
    if constexpr (NORM) {
        const float* xrow[MT];
#pragma unroll
        for (int a = 0; a < MT; ++a) xrow[a] = p.xn + (int64_t)(a * 16 + m) * (KS * 32) + g * 8;   // (rows behind rows_live are never read)
        // the RMSNorm gain of this wave's K range through its LDS slice (GammaStage; the launcher guarantees <= kGamSteps k-steps)
        _Float16* gslice = gam_lds + wave * kGamHalfs;
        GammaStage gst;
        gst.issue(p.gamma + (int64_t)ks0 * 32, gslice, (ks1 - ks0) * 32, lane);
        const _Float16* gam = gslice + g * 8 - ks0 * 32;
        if constexpr (PRE) {
            k_block_norm<MT, TT, U, false, W8, true>(wbase, xrow, gam, ks, U, row_ok, acc, ss, gst, wpre);
            ks += U;
        }
        int blk = 0;
        for (; ks + U <= ks1; ks += U) { alt_prio(p, wave, blk); k_block_norm<MT, TT, U, false, W8>(wbase, xrow, gam, ks, U, row_ok, acc, ss, gst); }
        if (ks < ks1) { alt_prio(p, wave, blk); k_block_norm<MT, TT, U, true, W8>(wbase, xrow, gam, ks, ks1 - ks, row_ok, acc, ss, gst); }
        if (p.prio_alt) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            ss[a] += __shfl_xor(ss[a], 16);
            ss[a] += __shfl_xor(ss[a], 32);
            if (g == 0) ssl[wave][a * 16 + m] = ss[a];   // this wave's share of sum(x^2) of row 16 a + m
        }
    } else {
        if constexpr (PRE) {
            k_block<MT, TT, TWO, U, false, W8, true>(wbase, xh_base, xl_base, KS, ks, U, row_ok, acc, wpre, x8_base);
            ks += U;
        }
        int blk = 0;
        for (; ks + U <= ks1; ks += U) { alt_prio(p, wave, blk); k_block<MT, TT, TWO, U, false, W8>(wbase, xh_base, xl_base, KS, ks, U, row_ok, acc, nullptr, x8_base); }
        if (ks < ks1) { alt_prio(p, wave, blk); k_block<MT, TT, TWO, U, true, W8>(wbase, xh_base, xl_base, KS, ks, ks1 - ks, row_ok, acc, nullptr, x8_base); }
        if (p.prio_alt) __builtin_amdgcn_s_setprio(0);
    }
    if constexpr (W8 && !TWO) {                      // the int8 MFMA left int32 sums in acc's registers (k_block)
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                const i32x4 c = __builtin_bit_cast(i32x4, acc[a][t]);
                acc[a][t] = f4{(float)c[0], (float)c[1], (float)c[2], (float)c[3]};
            }
    }
    after_k();
    trace_stamp(p, bx, by, wave, 1);

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
    if (r == 0) trace_stamp(p, bx, by, wave, 2);

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
            for (int w = 0; w < kWaves; ++w) tot += ssl[w][a * 16 + m];
            const float rs = rsqrtf(tot / (float)(KS * 32) + p.eps);
            v[0] *= rs; v[1] *= rs; v[2] *= rs; v[3] *= rs;
            u[0] *= rs; u[1] *= rs; u[2] *= rs; u[3] *= rs;
        }
        if constexpr (W8) tile_epilogue<EPI>(p, v, u, a * 16 + m, bx * T + t, g, by, fused, csv, csu);
        else if constexpr (kPreY) tile_epilogue<EPI>(p, v, u, a * 16 + m, bx * T + t, g, by, false, f4{0.f, 0.f, 0.f, 0.f},
                                                     f4{0.f, 0.f, 0.f, 0.f}, true, yold);
        else tile_epilogue<EPI>(p, v, u, a * 16 + m, bx * T + t, g, by);
    }
    }   // rounds
    if (p.trace) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); trace_stamp(p, bx, by, wave, 3); }
}

template <int MT, int T, int EPI, bool TWO, int U, bool NORM = false, bool W8 = false>
__global__ __launch_bounds__(kThreads) void gemm_skinny_kernel(const GemmParams p) {
    constexpr int TT = (EPI == EPI_SILU) ? 2 * T : T;
    constexpr int kRT = (MT * TT < 8) ? MT * TT : 8;
    __shared__ __attribute__((aligned(16))) float red[kWaves * kRT * 64 * 4];
    __shared__ float ssl[kWaves][32];
    __shared__ __attribute__((aligned(16))) _Float16 gam_lds[NORM ? kWaves * kGamHalfs : 8];
    gemm_skinny_body<MT, T, EPI, TWO, U, NORM, W8>(p, (int)blockIdx.x, (int)blockIdx.y, red, ssl, nullptr, NoHook(), gam_lds);
}

// int8 weight images (W8): more than 8 accumulator tiles next to the dequantised k-step pairs spill (36 .. 716 B per lane at
// 12 .. 16 tiles), so those shapes are not instantiated -- launch_T keeps int8 launches at or below 8.
// UA: k-steps per block of the LLM.int8 instantiation (codes on both sides: a k-step PAIR is one 16-byte load per operand, so
// the one-tile launches can keep sixteen k-steps in flight where the fp16-activation form stops at eight)
template <int MT, int T, int EPI, int UW, int UA = UW>
void launch_w8(const GemmParams& p, dim3 grid, dim3 block, hipStream_t s) {
    constexpr int TT = (EPI == EPI_SILU) ? 2 * T : T;
    if constexpr (MT * TT <= 8) {
        if constexpr (MT == 1 && EPI != EPI_ADD) {
            if (p.xn) {
                hipLaunchKernelGGL((gemm_skinny_kernel<MT, T, EPI, true, UW, true, true>), grid, block, 0, s, p);
                return;
            }
        }
        if (p.xscale)     // LLM.int8 codes: ONE activation plane (the "lo" plane of the a8 calls is all zeros)
            hipLaunchKernelGGL((gemm_skinny_kernel<MT, T, EPI, false, UA, false, true>), grid, block, 0, s, p);
        else
            hipLaunchKernelGGL((gemm_skinny_kernel<MT, T, EPI, true, UW, false, true>), grid, block, 0, s, p);
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
            constexpr int UA = (MT * TT <= 2 && (UV) >= 16) ? 16 : UW;                                \
            launch_w8<MT, T, EPI, UW, UA>(p, grid, block, s);                                         \
            break;                                                                                    \
        }                                                                                             \
        if constexpr (MT <= 2 && EPI != EPI_ADD) {                                                    \
            if (p.xn) {   /* (two row tiles of fp32 rows: eight k-steps of them in flight spill) */       \
                constexpr int UN = (MT == 2 && (UV) > 4) ? 4 : (UV);                                  \
                hipLaunchKernelGGL((gemm_skinny_kernel<MT, T, EPI, true, UN, true>), grid, block, 0, s, p); \
                break;                                                                                \
            }                                                                                         \
        }                                                                                             \
        if (two) hipLaunchKernelGGL((gemm_skinny_kernel<MT, T, EPI, true, UV>), grid, block, 0, s, p); \
        else hipLaunchKernelGGL((gemm_skinny_kernel<MT, T, EPI, false, UV>), grid, block, 0, s, p);    \
    } while (0)
#ifdef PC_DEV_SWEEPS   // block-depth sweeps (tools/gemm_n4096_sweep.py): ~500 extra kernel instantiations, not in the product build
    if constexpr (MT * TT <= 2) {
        if (forced == 16) { PC_GO(16); return pc_check_launch("gemm_skinny_kernel"); }
    }
    if constexpr (MT * TT <= 4) {
        if (forced == 8) { PC_GO(8); return pc_check_launch("gemm_skinny_kernel"); }
        if (forced == 4) { PC_GO(4); return pc_check_launch("gemm_skinny_kernel"); }
    }
    if (forced == 2) { PC_GO(2); return pc_check_launch("gemm_skinny_kernel"); }
#endif
    if constexpr (MT * TT <= 2) {
        // one or two tiles per workgroup (the N = hidden projections): measured in-graph on the 7b shapes
        // (tools/gemm_n4096_sweep.py), k-steps per block 4 / 8 / 16: o_proj 9.3 / 9.8 / 9.5 us at 12 rows, 8.0 / 8.9 / 8.9 at
        // one row; down_proj 24.3 / 24.7 / 23.2 at 12 rows, 19.1 / 20.9 / 21.8 at one row -- shallow blocks win except for
        // the long-K launch with its activation loads (more than 4 rows), which wants the deep one
        if (!forced) {
            // (the deep block only where it was measured: one tile, one row tile, residual add = down_proj; every other
            // shape with U = 16 spills 100 .. 480 B per lane)
            if constexpr (MT == 1 && TT == 1 && EPI == EPI_ADD) {
                if (p.M > 4 && p.KS >= 256) { PC_GO(16); return pc_check_launch("gemm_skinny_kernel"); }
            }
            PC_GO(4);
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
    constexpr int kMaxT8 = 8 / MT / (EPI == EPI_SILU ? 2 : 1);     // int8 images: at most 8 accumulator tiles (launch_w8)
    if (p.w8 && T > kMaxT8) T = kMaxT8;
    if constexpr (kMaxT >= 8) { if (T >= 8) return launch_one<MT, 8, EPI>(p, units, s); }
    if constexpr (kMaxT >= 4) { if (T >= 4) return launch_one<MT, 4, EPI>(p, units, s); }
    if constexpr (kMaxT >= 3) { if (T == 3) return launch_one<MT, 3, EPI>(p, units, s); }
    if constexpr (kMaxT >= 2) { if (T >= 2) return launch_one<MT, 2, EPI>(p, units, s); }
    return launch_one<MT, 1, EPI>(p, units, s);
}

// launch_T<MT, EPI> for a run-time epilogue (defined once per row regime in pc_gemm_mt{1,2,4}.hip)
int launch_skinny_mt1(int epi, const GemmParams& p, int T, int units, hipStream_t s);
int launch_skinny_mt2(int epi, const GemmParams& p, int T, int units, hipStream_t s);
int launch_skinny_mt4(int epi, const GemmParams& p, int T, int units, hipStream_t s);
// gemm_rows_kernel (65..512 rows), pc_gemm_rows.hip
int launch_rows_epi(int epi, const GemmParams& p, int units, hipStream_t s);
// residual add with K split across workgroups, reduced inside the launch (pc_gemm_ks.hip)
int launch_skinny_ks(const void* wf, const void* xf_hi, const void* xf_lo, int M, int N, int K, float* y, int64_t ldy, int kslices,
                     int tiles_per_wg, void* scratch, int64_t scratch_bytes, void* counters, const int32_t* rows_dev, hipStream_t s);
int choose_T(int units);

#define PC_SKINNY_MT_DEFINE(NAME, MTV)                                                    \
    int NAME(int epi, const GemmParams& p, int T, int units, hipStream_t s) {             \
        switch (epi) {                                                                    \
            case EPI_STORE: return launch_T<MTV, EPI_STORE>(p, T, units, s);              \
            case EPI_ADD: return launch_T<MTV, EPI_ADD>(p, T, units, s);                  \
            case EPI_SILU: return launch_T<MTV, EPI_SILU>(p, T, units, s);                \
            case EPI_ROPE: return launch_T<MTV, EPI_ROPE>(p, T, units, s);                \
            default: return launch_T<MTV, EPI_GELU>(p, T, units, s);                      \
        }                                                                                 \
    }
// the same dispatcher over TWO translation units (the one-row-tile regime is the longest compile of the library: its epilogues with
// an output plane on one side, the residual-stream ones on the other)
#define PC_SKINNY_MT_DEFINE_A(NAME, MTV)                                                  \
    int NAME(int epi, const GemmParams& p, int T, int units, hipStream_t s) {             \
        switch (epi) {                                                                    \
            case EPI_SILU: return launch_T<MTV, EPI_SILU>(p, T, units, s);                \
            default: return launch_T<MTV, EPI_ROPE>(p, T, units, s);                      \
        }                                                                                 \
    }
#define PC_SKINNY_MT_DEFINE_B(NAME, NAME_A, MTV)                                          \
    int NAME_A(int epi, const GemmParams& p, int T, int units, hipStream_t s);            \
    int NAME(int epi, const GemmParams& p, int T, int units, hipStream_t s) {             \
        switch (epi) {                                                                    \
            case EPI_STORE: return launch_T<MTV, EPI_STORE>(p, T, units, s);              \
            case EPI_ADD: return launch_T<MTV, EPI_ADD>(p, T, units, s);                  \
            case EPI_SILU: case EPI_ROPE: return NAME_A(epi, p, T, units, s);             \
            default: return launch_T<MTV, EPI_GELU>(p, T, units, s);                      \
        }                                                                                 \
    }

}  // namespace pcg
