// LLM.int8() activation side (the reference's GPU configs load every model with load_in_8bit=True: demo.py:27-29,
// eval.py:36-42, config/llm_config_*.json:5 -> transformers -> bitsandbytes.nn.Linear8bitLt(threshold = 6.0)).
// bitsandbytes is a third-party dependency that is not in the reference tree; what is implemented is its published
// algorithm (Dettmers et al., "LLM.int8()", NeurIPS 2022, section 3; restated on the CPU in oracle/llmint8_oracle.py):
//
//   X  = fp16(input)                                   [T][K]
//   O  = {k : |X[t][k]| >= threshold for some t}       outlier columns
//   X0 = X with every entry |x| >= threshold zeroed;   SCA[t] = max_k |X0[t][k]|
//   CA = round_half_even(X0 * 127 / SCA), columns in O zeroed
//   Y  = (CA . CB^T) * SCA[t] * SCB[n] / 127^2  +  X[:, O] . fp16(CB[:, O] * SCB / 127)^T
//
// Two kernels around the projection launch (pc_gemm_skinny_a8 / pc_gemm_qkv_rope_a8 below 65 rows, pc_gemm_dense above):
//   quant_act_kernel     one workgroup per row: outlier test, row absmax without the outliers, codes (held in fp16: |code|
//                        <= 127 is exact, the projection kernels run their fp16 MFMAs on them with fp32 accumulation --
//                        exact integers below 2^24 per accumulator), x_scale[t] = SCA[t] / 127, and one flag byte per
//                        outlier column.  Only the outlier ENTRIES are zeroed here.
//   outlier_corr_kernel  compacts the flagged columns (deterministic order) and writes
//                            corr[t][n] = sum_{k in O} ( X[t][k] * fp16(CB[n][k] * scale[n])  -  CA[t][k] * CB[n][k] * x_scale[t] * scale[n] )
//                        (CB read from a transposed int8 image [K][N]: a column of W is contiguous there)
//                        i.e. the fp16 part of the decomposition MINUS what the int8 product still carries in those columns
//                        for rows whose own entry is not an outlier -- algebraically the "zero the whole column" of the
//                        published form, without a second pass over CA.  The projection's epilogue adds corr before its
//                        nonlinearity when *has != 0.  With no outlier (the usual case behind a norm) the kernel ends after
//                        the flag scan.
// Layouts: activations either row-major [T][ld] or the fragment-major planes of pc_gemm.hip ([M/16][K/32][64][8]).
#include <hip/hip_fp16.h>

#include "pc_common.h"
#ifdef PC_DEV_SWEEPS
#include "pc_dev.h"
#endif

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int64_t frag_off(int m, int k, int KS) {
    return ((((int64_t)(m >> 4) * KS + (k >> 5)) * 64) + ((k & 31) >> 3) * 16 + (m & 15)) * 8 + (k & 7);
}

// The int8 operand image of the codes (pc_gemm x_codes8): [T/16][K/64][64 lanes][16 bytes], lane (row & 15, g) holding its eight
// codes of k-step 2s (k = 64 s + 8 g + e) and then of 2s + 1 (k = 64 s + 32 + 8 g + e) as signed bytes -- the byte order of the
// int8 weight image, i.e. one v_mfma_i32_16x16x64_i8 operand per 16-byte load.  r[e]: the rounded, clamped codes of chunk c.
__device__ __forceinline__ void store_codes8(signed char* __restrict__ codes8, int row, int c, int KS, const float (&r)[8]) {
    const int s = c >> 2, g = c & 3;
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        lo |= ((uint32_t)(int)r[e] & 0xffu) << (8 * e);
        hi |= ((uint32_t)(int)r[4 + e] & 0xffu) << (8 * e);
    }
    const int64_t off = ((((int64_t)(row >> 4) * (KS >> 1) + (s >> 1)) * 64 + g * 16 + (row & 15)) << 4) + ((s & 1) << 3);
    typedef uint32_t u2v __attribute__((ext_vector_type(2)));
    *(u2v*)(codes8 + off) = u2v{lo, hi};
}

template <int G, bool FRAG>
__global__ __launch_bounds__(256) void quant_act_kernel(const _Float16* __restrict__ x, int64_t ld, int K, _Float16* __restrict__ codes,
                                                        float* __restrict__ x_scale, unsigned char* __restrict__ flags_set,
                                                        unsigned char* __restrict__ flags_clear, int clear_len, float threshold,
                                                        signed char* __restrict__ codes8) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x, nchunk = K >> 3, KS = K >> 5;
    // the flag bytes of the NEXT activation slot are cleared here (its quantiser runs after this launch, its last reader ran
    // long before): no memset node, no race with this slot's own flags
    if (flags_clear)
        for (int k = row * 256 + tid; k < clear_len; k += gridDim.x * 256) flags_clear[k] = 0;
    float a[G][8];
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int c = tid + i * 256;
#pragma unroll
        for (int e = 0; e < 8; ++e) a[i][e] = 0.f;
        if (c < nchunk) {
            const int64_t off = FRAG ? frag_off(row, c * 8, KS) : (int64_t)row * ld + c * 8;
            const h8 v = *(const h8*)(x + off);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = (float)v[e];
                const bool outl = threshold > 0.f && fabsf(f) >= threshold;
                if (outl) flags_set[c * 8 + e] = 1;
                a[i][e] = outl ? 0.f : f;
                mx = fmaxf(mx, fabsf(a[i][e]));
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    const float sca = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float inv = sca > 0.f ? 127.0f / sca : 0.f;
    if (tid == 0) x_scale[row] = sca / 127.0f;
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int c = tid + i * 256;
        if (c < nchunk) {
            h8 q;
            float rr[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float r = rintf(a[i][e] * inv);
                r = fminf(fmaxf(r, -127.f), 127.f);
                q[e] = (_Float16)r;
                rr[e] = r;
            }
            const int64_t off = FRAG ? frag_off(row, c * 8, KS) : (int64_t)row * ld + c * 8;
            *(h8*)(codes + off) = q;
            if (codes8) store_codes8(codes8, row, c, KS, rr);
        }
    }
}

// pc_quant_rows_i8 (round 5): a residual activation plane x_lo fp16 [M][K] as row-wise absmax int8 codes [M][K] + scale[M] for the
// int8 MFMA of pc_gemm_dense_lo8.  One workgroup per row; no outlier logic (this is not LLM.int8: the plane is a rounding residual).
template <int G>
__global__ __launch_bounds__(256) void quant_rows_kernel(const _Float16* __restrict__ x, int64_t ldx, int K, signed char* __restrict__ codes,
                                                         int64_t ld8, float* __restrict__ scale) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x, nchunk = K >> 3;
    float a[G][8];
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int c = tid + i * 256;
#pragma unroll
        for (int e = 0; e < 8; ++e) a[i][e] = 0.f;
        if (c < nchunk) {
            const h8 v = *(const h8*)(x + (int64_t)row * ldx + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) { a[i][e] = (float)v[e]; mx = fmaxf(mx, fabsf(a[i][e])); }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    const float amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float inv = amax > 0.f ? 127.0f / amax : 0.f;
    if (tid == 0) scale[row] = amax / 127.0f;
#pragma unroll
    for (int i = 0; i < G; ++i) {
        const int c = tid + i * 256;
        if (c < nchunk) {
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float r0 = fminf(fmaxf(rintf(a[i][e] * inv), -127.f), 127.f);
                const float r1 = fminf(fmaxf(rintf(a[i][4 + e] * inv), -127.f), 127.f);
                lo |= ((uint32_t)(int)r0 & 0xffu) << (8 * e);
                hi |= ((uint32_t)(int)r1 & 0xffu) << (8 * e);
            }
            typedef uint32_t u2v __attribute__((ext_vector_type(2)));
            *(u2v*)(codes + (int64_t)row * ld8 + c * 8) = u2v{lo, hi};
        }
    }
}

// RMSNorm + quantiser in one launch for the two projection inputs that come out of a norm (q|k|v, gate|up; <= 64-row path):
// exactly rmsnorm_frag_kernel's arithmetic (pc_gemm.hip: same per-thread partial sums, same reduction order, v = g * (x * rs),
// hi = fp16(v)) followed by quant_act_kernel's on the hi values, so the pair of launches and this one are bit-identical.
// Writes the hi plane (the fp16 activations the outlier correction reads), the codes plane, x_scale and the outlier flags.
typedef float f4 __attribute__((ext_vector_type(4)));

template <int G>
__global__ __launch_bounds__(256) void rmsnorm_quant_kernel(const float* __restrict__ x, const _Float16* __restrict__ w,
                                                            _Float16* __restrict__ of_hi, _Float16* __restrict__ codes,
                                                            float* __restrict__ x_scale, unsigned char* __restrict__ flags_set,
                                                            unsigned char* __restrict__ flags_clear, int clear_len, float threshold,
                                                            int hidden, float eps, signed char* __restrict__ codes8) {
    __shared__ float red[4];
    __shared__ float redq[4];
    const int row = blockIdx.x, tid = threadIdx.x, nv = hidden >> 3, KS = hidden >> 5;
    if (flags_clear)
        for (int k = row * 256 + tid; k < clear_len; k += gridDim.x * 256) flags_clear[k] = 0;
    const float* xr = x + (int64_t)row * hidden;
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
#pragma unroll
    for (int k = 0; k < G; ++k)
        ss += va[k][0] * va[k][0] + va[k][1] * va[k][1] + va[k][2] * va[k][2] + va[k][3] * va[k][3] +
              vb[k][0] * vb[k][0] + vb[k][1] * vb[k][1] + vb[k][2] * vb[k][2] + vb[k][3] * vb[k][3];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    const float rs = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)hidden + eps);
    float a[G][8];
    float mx = 0.f;
#pragma unroll
    for (int k = 0; k < G; ++k) {
        const int i = tid + k * 256;
#pragma unroll
        for (int e = 0; e < 8; ++e) a[k][e] = 0.f;
        if (i < nv) {
            const h8 gw = *(const h8*)(w + i * 8);
            h8 hi;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = (float)gw[e] * ((e < 4 ? va[k][e] : vb[k][e - 4]) * rs);
                _Float16 vh, vl;
                pc_split(v, vh, vl);
                hi[e] = vh;
                const float f = (float)vh;
                const bool outl = threshold > 0.f && fabsf(f) >= threshold;
                if (outl) flags_set[i * 8 + e] = 1;
                a[k][e] = outl ? 0.f : f;
                mx = fmaxf(mx, fabsf(a[k][e]));
            }
            *(h8*)(of_hi + frag_off(row, i * 8, KS)) = hi;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((tid & 63) == 0) redq[tid >> 6] = mx;
    __syncthreads();
    const float sca = fmaxf(fmaxf(redq[0], redq[1]), fmaxf(redq[2], redq[3]));
    const float inv = sca > 0.f ? 127.0f / sca : 0.f;
    if (tid == 0) x_scale[row] = sca / 127.0f;
#pragma unroll
    for (int k = 0; k < G; ++k) {
        const int i = tid + k * 256;
        if (i < nv) {
            h8 q;
            float rr[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float r = rintf(a[k][e] * inv);
                r = fminf(fmaxf(r, -127.f), 127.f);
                q[e] = (_Float16)r;
                rr[e] = r;
            }
            *(h8*)(codes + frag_off(row, i * 8, KS)) = q;
            if (codes8) store_codes8(codes8, row, i, KS, rr);
        }
    }
}

constexpr int kMaxCols = 512;       // outlier columns handled per batch of the correction kernel (each batch costs ~4 us of
                                    // dependent staging phases: 460 columns took 56 us in four batches of 128)
constexpr int kCorrRows = 16;       // activation rows staged per pass (one per row slot of the workgroup)
constexpr int kCorrCols = 16;       // output columns per workgroup of the correction kernel

template <bool FRAG>
__global__ __launch_bounds__(256) void outlier_corr_kernel(const unsigned char* __restrict__ flags, int K, const _Float16* __restrict__ x,
                                                           const _Float16* __restrict__ codes, int64_t ldx, const float* __restrict__ x_scale,
                                                           const signed char* __restrict__ cbt, int64_t ldt, const float* __restrict__ w_scale,
                                                           const int32_t* __restrict__ row_perm, int T, int N, float* __restrict__ corr,
                                                           int64_t ldc, int32_t* __restrict__ has) {
    __shared__ int cols[kMaxCols];
    __shared__ int cnt[257];
    __shared__ __attribute__((aligned(16))) unsigned char lf[16384];       // the flag bytes (K <= 16384)
    const int tid = threadIdx.x, KS = K >> 5;
    // ---- flags into LDS with coalesced 16-byte loads; the usual case (no outlier anywhere) ends right here ----
    typedef unsigned int u4v __attribute__((ext_vector_type(4)));
    int any = 0;
    for (int c = tid; c < (K >> 4); c += 256) {
        const u4v v = ((const u4v*)flags)[c];
        ((u4v*)lf)[c] = v;
        any |= (v[0] | v[1] | v[2] | v[3]) != 0;
    }
    if (!__syncthreads_or(any)) {
        if (blockIdx.x == 0 && tid == 0) has[0] = 0;
        return;
    }
    // ---- deterministic compaction of the flagged columns: thread i owns the contiguous range [i*per, (i+1)*per) ----
    const int per = (K + 255) / 256;
    int mine = 0;
    for (int k = tid * per; k < (tid + 1) * per && k < K; ++k) mine += lf[k] ? 1 : 0;
    // exclusive prefix sum of the 256 per-thread counts: shuffles inside a wave, the four wave totals through LDS (a
    // single thread walking cnt[] cost ~12 us of serial LDS round trips in every call that had any outlier column)
    {
        __shared__ int wtot[4];
        const int lane = tid & 63, wv = tid >> 6;
        int incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        if (lane == 63) wtot[wv] = incl;
        __syncthreads();
        int woff = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) woff += (w < wv) ? wtot[w] : 0;
        cnt[tid + 1] = woff + incl;                      // inclusive: cnt[i] = flagged columns owned by threads < i
        if (tid == 0) cnt[0] = 0;
    }
    __syncthreads();
    const int total = cnt[256];
    if (blockIdx.x == 0 && tid == 0) has[0] = 1;
    // 16 output columns x 16 row slots per workgroup: four times the workgroups of a 64-column split (N = 4096 gave 64
    // workgroups on 256 CUs) and a 12-row step keeps 12 of the 16 slots busy instead of 3 of 4 at three rows each
    const int nl = tid & (kCorrCols - 1), tg = tid / kCorrCols;
    const int n = blockIdx.x * kCorrCols + nl;
    const int nrow = n < N ? (row_perm ? row_perm[n] : n) : 0;
    const float ws = n < N ? w_scale[nrow] : 0.f;
    __shared__ _Float16 wl[kMaxCols][kCorrCols];              // this workgroup's weight codes of the batch's columns
    __shared__ _Float16 xl[kCorrRows][kMaxCols];              // fp16 activations / their codes at the batch's columns
    __shared__ _Float16 cl[kCorrRows][kMaxCols];
    for (int base = 0; base < total; base += kMaxCols) {
        __syncthreads();
        {   // this batch of the column list (in column order)
            int w = cnt[tid] - base;
            for (int k = tid * per; k < (tid + 1) * per && k < K; ++k)
                if (lf[k]) { if (w >= 0 && w < kMaxCols) cols[w] = k; ++w; }
        }
        __syncthreads();
        const int nb = total - base < kMaxCols ? total - base : kMaxCols;
        // the weight codes of the batch's columns, from the TRANSPOSED int8 image [K][N] (a column of W is a contiguous row
        // there: gathering 600 columns of down_proj from the row-major image cost 400 us per launch), once per workgroup
        // and batch, then reused by every activation row
        for (int e = tid; e < nb * kCorrCols; e += 256) {
            const int j = e / kCorrCols, l = e & (kCorrCols - 1);
            const int nn = blockIdx.x * kCorrCols + l;
            const int nr = nn < N ? (row_perm ? row_perm[nn] : nn) : 0;
            wl[j][l] = (_Float16)(float)cbt[(int64_t)cols[j] * ldt + nr];
        }
        for (int t0 = 0; t0 < T; t0 += kCorrRows) {
            __syncthreads();
            const int nt = T - t0 < kCorrRows ? T - t0 : kCorrRows;
            // activations and codes at the flagged columns into LDS (independent loads), so that the sum below runs
            // out of LDS: with the loads inside the column loop a 600-column batch took 400 us
            for (int e = tid; e < nt * nb; e += 256) {
                const int tt = e / nb, j = e - tt * nb;
                const int64_t xo = FRAG ? frag_off(t0 + tt, cols[j], KS) : (int64_t)(t0 + tt) * ldx + cols[j];
                xl[tt][j] = x[xo];
                cl[tt][j] = codes[xo];
            }
            __syncthreads();
            for (int tt = tg; tt < nt; tt += 256 / kCorrCols) {
                float acc = 0.f;
                const float xs = x_scale[t0 + tt] * ws;
#pragma unroll 8
                for (int j = 0; j < nb; ++j) {       // (unrolled: eight columns' LDS reads in flight instead of one round trip each)
                    const float wq = (float)wl[j][nl];
                    const float wd = (float)(_Float16)(wq * ws);        // fp16(CB * SCB / 127): the fp16 weight column
                    acc += (float)xl[tt][j] * wd - (float)cl[tt][j] * wq * xs;
                }
                if (n < N) {
                    float* cp = corr + (int64_t)(t0 + tt) * ldc + n;
                    *cp = base == 0 ? acc : *cp + acc;
                }
            }
        }
    }
}

}  // namespace

PC_EXPORT int pc_quant_act_i8(const void* x, int64_t ldx, int32_t frag, int32_t T, int32_t K, void* codes, float* x_scale,
                              void* flags_set, void* flags_clear, int32_t clear_len, float threshold, void* codes8, void* stream) {
    PC_REQUIRE(!codes8 || (K % 64 == 0 && ((uintptr_t)codes8 & 15) == 0), PC_ERR_ARG,
               "pc_quant_act_i8: the int8 operand image needs K %% 64 == 0 and a 16-byte aligned buffer");
    PC_REQUIRE(T > 0 && K > 0 && K % 32 == 0 && K <= 16384, PC_ERR_ARG, "pc_quant_act_i8: need T > 0, K %% 32 == 0, K <= 16384");
    PC_REQUIRE(x && codes && x_scale && flags_set, PC_ERR_ARG, "pc_quant_act_i8: null pointer");
    PC_REQUIRE(frag || (ldx >= K && ldx % 8 == 0), PC_ERR_ARG, "pc_quant_act_i8: row-major planes need ldx >= K, ldx %% 8 == 0");
    const int groups = pc_ceil_div(K / 8, 256);
    hipStream_t s = (hipStream_t)stream;
#define PC_Q(GV)                                                                                                          \
    do {                                                                                                                  \
        if (frag) hipLaunchKernelGGL((quant_act_kernel<GV, true>), dim3(T), dim3(256), 0, s, (const _Float16*)x, ldx, K,  \
                                     (_Float16*)codes, x_scale, (unsigned char*)flags_set, (unsigned char*)flags_clear, clear_len, threshold, (signed char*)codes8); \
        else hipLaunchKernelGGL((quant_act_kernel<GV, false>), dim3(T), dim3(256), 0, s, (const _Float16*)x, ldx, K,       \
                                (_Float16*)codes, x_scale, (unsigned char*)flags_set, (unsigned char*)flags_clear, clear_len, threshold, (signed char*)codes8); \
    } while (0)
    if (groups <= 1) PC_Q(1); else if (groups <= 2) PC_Q(2); else if (groups <= 4) PC_Q(4); else PC_Q(8);
#undef PC_Q
    return pc_check_launch("quant_act_kernel");
}

PC_EXPORT int pc_rmsnorm_quant_i8(const float* x, const void* norm_weight, float eps, int32_t T, int32_t hidden, void* x_hi,
                                  void* codes, float* x_scale, void* flags_set, void* flags_clear, int32_t clear_len,
                                  float threshold, void* codes8, void* stream) {
    PC_REQUIRE(!codes8 || (hidden % 64 == 0 && ((uintptr_t)codes8 & 15) == 0), PC_ERR_ARG,
               "pc_rmsnorm_quant_i8: the int8 operand image needs hidden %% 64 == 0 and a 16-byte aligned buffer");
    PC_REQUIRE(T > 0 && T <= 64 && hidden > 0 && hidden % 32 == 0 && hidden <= 16384, PC_ERR_ARG,
               "pc_rmsnorm_quant_i8: need 1 <= T <= 64, hidden %% 32 == 0, hidden <= 16384");
    PC_REQUIRE(x && norm_weight && x_hi && codes && x_scale && flags_set, PC_ERR_ARG, "pc_rmsnorm_quant_i8: null pointer");
    const int groups = pc_ceil_div(hidden / 8, 256);
    hipStream_t s = (hipStream_t)stream;
#define PC_RQ(GV)                                                                                                          \
    hipLaunchKernelGGL(rmsnorm_quant_kernel<GV>, dim3(T), dim3(256), 0, s, x, (const _Float16*)norm_weight, (_Float16*)x_hi,  \
                       (_Float16*)codes, x_scale, (unsigned char*)flags_set, (unsigned char*)flags_clear, clear_len, threshold, \
                       hidden, eps, (signed char*)codes8)
    if (groups <= 1) PC_RQ(1); else if (groups <= 2) PC_RQ(2); else if (groups <= 4) PC_RQ(4); else PC_RQ(8);
#undef PC_RQ
    return pc_check_launch("rmsnorm_quant_kernel");
}

PC_EXPORT int pc_outlier_corr(const void* flags, int32_t K, const void* x, const void* codes, int64_t ldx, int32_t frag,
                              const float* x_scale, const void* w_codes_t, int64_t ldt, const float* w_scale, const int32_t* row_perm,
                              int32_t T, int32_t N, float* corr, int64_t ldc, int32_t* has, void* stream) {
    PC_REQUIRE(T > 0 && N > 0 && K > 0 && K % 32 == 0 && K <= 16384 && ((uintptr_t)flags & 15) == 0, PC_ERR_ARG,
               "pc_outlier_corr: bad sizes (K %% 32 == 0, K <= 16384, flags 16-byte aligned)");
    PC_REQUIRE(flags && x && codes && x_scale && w_codes_t && w_scale && corr && has && ldc >= N && ldt >= N, PC_ERR_ARG,
               "pc_outlier_corr: null pointer or short strides");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(pc_ceil_div(N, kCorrCols));
    if (frag)
        hipLaunchKernelGGL((outlier_corr_kernel<true>), grid, dim3(256), 0, s, (const unsigned char*)flags, K, (const _Float16*)x,
                           (const _Float16*)codes, ldx, x_scale, (const signed char*)w_codes_t, ldt, w_scale, row_perm, T, N, corr, ldc, has);
    else
        hipLaunchKernelGGL((outlier_corr_kernel<false>), grid, dim3(256), 0, s, (const unsigned char*)flags, K, (const _Float16*)x,
                           (const _Float16*)codes, ldx, x_scale, (const signed char*)w_codes_t, ldt, w_scale, row_perm, T, N, corr, ldc, has);
    return pc_check_launch("outlier_corr_kernel");
}

#ifdef PC_DEV_SWEEPS      // (dev builds only: see pc_dev.h)
PC_EXPORT int pc_quant_rows_i8(const void* x, int64_t ldx, int32_t M, int32_t K, void* codes, int64_t ld8, float* scale, void* stream) {
    PC_REQUIRE(x && codes && scale && M > 0 && K > 0 && K % 8 == 0 && ldx >= K && ldx % 8 == 0 && ld8 >= K && ld8 % 8 == 0 &&
               ((uintptr_t)x & 15) == 0 && ((uintptr_t)codes & 7) == 0, PC_ERR_ARG,
               "pc_quant_rows_i8: K %% 8 == 0, fp16 rows 16-byte aligned, code rows 8-byte aligned");
    const int g = pc_ceil_div(K >> 3, 256);
    PC_REQUIRE(g <= 8, PC_ERR_ARG, "pc_quant_rows_i8: K up to 16384");
    hipStream_t s = (hipStream_t)stream;
    const _Float16* xp = (const _Float16*)x;
    signed char* cp = (signed char*)codes;
    switch (g) {
        case 1: hipLaunchKernelGGL((quant_rows_kernel<1>), dim3(M), dim3(256), 0, s, xp, ldx, K, cp, ld8, scale); break;
        case 2: hipLaunchKernelGGL((quant_rows_kernel<2>), dim3(M), dim3(256), 0, s, xp, ldx, K, cp, ld8, scale); break;
        case 3: case 4: hipLaunchKernelGGL((quant_rows_kernel<4>), dim3(M), dim3(256), 0, s, xp, ldx, K, cp, ld8, scale); break;
        default: hipLaunchKernelGGL((quant_rows_kernel<8>), dim3(M), dim3(256), 0, s, xp, ldx, K, cp, ld8, scale); break;
    }
    return pc_check_launch("quant_rows_kernel");
}
#endif  // PC_DEV_SWEEPS
