// Library plumbing (version, thread-local error string), the small elementwise / reduction pieces of the
// layer stack, and the hardware-layout probe the GPU tests use to pin the two lane maps pc_attn.hip
// depends on.
//
// Replaces  LlamaRMSNorm.forward       promptcache/model/llama2.py:103-108   (pc_rmsnorm)
//           act_fn(gate) * up          promptcache/model/llama2.py:242       (pc_silu_mul)
//           embed_tokens(input_ids)    promptcache/model/llama2.py:869       (pc_embed_gather)
#include <hip/hip_fp16.h>
#include <string.h>

#include "pc_common.h"

static thread_local char g_err[512] = "";

void pc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

PC_EXPORT int pc_version(void) { return PC_ABI_VERSION; }
PC_EXPORT const char* pc_last_error_string(void) { return g_err; }

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// One workgroup (256 threads) per row; 8 elements (16 B of fp16 / 32 B of fp32) per lane per pass.
// fp32 statistics and fp32 gain multiply, ONE fp16 rounding at the end.  (The reference casts the normalised
// value back to the input dtype before the gain, llama2.py:108 -- a no-op on its fp32 CPU path, which is the
// parity target, so rounding there would only add error.)
template <bool XF32>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const void* __restrict__ xin, const _Float16* __restrict__ w,
                                                      _Float16* __restrict__ out, int hidden, float eps,
                                                      _Float16* __restrict__ out_lo = nullptr) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const int nv = hidden >> 3;
    float ss = 0.f;
    for (int i = tid; i < nv; i += 256) {
        if (XF32) {
            const f4* p = (const f4*)((const float*)xin + (int64_t)row * hidden + i * 8);
            const f4 a = p[0], b = p[1];
            ss += a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3] + b[0] * b[0] + b[1] * b[1] + b[2] * b[2] + b[3] * b[3];
        } else {
            const h8 a = *(const h8*)((const _Float16*)xin + (int64_t)row * hidden + i * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) ss += (float)a[e] * (float)a[e];
        }
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    const float tot = red[0] + red[1] + red[2] + red[3];
    const float rs = rsqrtf(tot / (float)hidden + eps);
    for (int i = tid; i < nv; i += 256) {
        const h8 g = *(const h8*)(w + i * 8);
        h8 o;
        if (XF32) {
            const f4* p = (const f4*)((const float*)xin + (int64_t)row * hidden + i * 8);
            const f4 a = p[0], b = p[1];
            if (out_lo) {      // split-precision output: hi to `out`, the fp16 residual to `out_lo`
                h8 ol;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    _Float16 vh, vl;
                    pc_split((float)g[e] * ((e < 4 ? a[e] : b[e - 4]) * rs), vh, vl);
                    o[e] = vh; ol[e] = vl;
                }
                *(h8*)(out_lo + (int64_t)row * hidden + i * 8) = ol;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = (_Float16)((float)g[e] * (a[e] * rs));
                    o[e + 4] = (_Float16)((float)g[e + 4] * (b[e] * rs));
                }
            }
        } else {
            const h8 a = *(const h8*)((const _Float16*)xin + (int64_t)row * hidden + i * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (_Float16)((float)g[e] * ((float)a[e] * rs));
        }
        *(h8*)(out + (int64_t)row * hidden + i * 8) = o;
    }
}

// gate_up: [rows][2*inter] (gate columns first, then up), fp32 or fp16 -> out[rows][inter] = silu(gate) * up,
// fp32 math, one fp16 rounding.
template <bool F32>
__global__ __launch_bounds__(256) void silu_mul_kernel(const void* __restrict__ gu, _Float16* __restrict__ out,
                                                       int inter, const float* __restrict__ gu2 = nullptr,
                                                       _Float16* __restrict__ out_lo = nullptr) {
    const int row = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i * 8 >= inter) return;
    float gt[8], up[8];
    if (F32) {
        const float* g = (const float*)gu + (int64_t)row * 2 * inter + i * 8;
        const f4 a = *(const f4*)g, b = *(const f4*)(g + 4), c = *(const f4*)(g + inter), d = *(const f4*)(g + inter + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { gt[e] = a[e]; gt[e + 4] = b[e]; up[e] = c[e]; up[e + 4] = d[e]; }
        if (gu2) {             // second addend (the lo-plane half of a split-precision projection)
            const float* g2 = gu2 + (int64_t)row * 2 * inter + i * 8;
            const f4 a2 = *(const f4*)g2, b2 = *(const f4*)(g2 + 4), c2 = *(const f4*)(g2 + inter), d2 = *(const f4*)(g2 + inter + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { gt[e] += a2[e]; gt[e + 4] += b2[e]; up[e] += c2[e]; up[e + 4] += d2[e]; }
        }
    } else {
        const _Float16* g = (const _Float16*)gu + (int64_t)row * 2 * inter + i * 8;
        const h8 a = *(const h8*)g, c = *(const h8*)(g + inter);
#pragma unroll
        for (int e = 0; e < 8; ++e) { gt[e] = (float)a[e]; up[e] = (float)c[e]; }
    }
    h8 o;
    if (out_lo) {
        h8 ol;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            _Float16 vh, vl;
            pc_split((gt[e] / (1.0f + __expf(-gt[e]))) * up[e], vh, vl);
            o[e] = vh; ol[e] = vl;
        }
        *(h8*)(out_lo + (int64_t)row * inter + i * 8) = ol;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (_Float16)((gt[e] / (1.0f + __expf(-gt[e]))) * up[e]);
    }
    *(h8*)(out + (int64_t)row * inter + i * 8) = o;
}

// torch.nn.LayerNorm over the last dim (falcon.py:757, :1020): fp32 statistics (two passes: mean, then the centred
// sum of squares -- no E[x^2] - mean^2 cancellation), affine in fp32, one fp16 rounding.  x fp32 [rows][hidden].
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const _Float16* __restrict__ w,
                                                        const _Float16* __restrict__ b, _Float16* __restrict__ out,
                                                        int hidden, float eps, _Float16* __restrict__ out_lo) {
    __shared__ float red[2][4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const int nv = hidden >> 3;
    const float* xr = x + (int64_t)row * hidden;
    float sm = 0.f;
    for (int i = tid; i < nv; i += 256) {
        const f4 a = *(const f4*)(xr + i * 8), c = *(const f4*)(xr + i * 8 + 4);
        sm += a[0] + a[1] + a[2] + a[3] + c[0] + c[1] + c[2] + c[3];
    }
    sm = wave_sum(sm);
    if ((tid & 63) == 0) red[0][tid >> 6] = sm;
    __syncthreads();
    const float mu = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (float)hidden;
    float ss = 0.f;
    for (int i = tid; i < nv; i += 256) {
        const f4 a = *(const f4*)(xr + i * 8), c = *(const f4*)(xr + i * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { ss += (a[e] - mu) * (a[e] - mu) + (c[e] - mu) * (c[e] - mu); }
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red[1][tid >> 6] = ss;
    __syncthreads();
    const float rs = rsqrtf((red[1][0] + red[1][1] + red[1][2] + red[1][3]) / (float)hidden + eps);
    for (int i = tid; i < nv; i += 256) {
        const f4 a = *(const f4*)(xr + i * 8), c = *(const f4*)(xr + i * 8 + 4);
        const h8 g = *(const h8*)(w + i * 8);
        h8 bb = {0, 0, 0, 0, 0, 0, 0, 0};
        if (b) bb = *(const h8*)(b + i * 8);       // MPT's LayerNorm has no bias (mpt.py:207, :215)
        h8 o, ol;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = ((e < 4 ? a[e] : c[e - 4]) - mu) * rs * (float)g[e] + (float)bb[e];
            _Float16 vh, vl;
            pc_split(v, vh, vl);
            o[e] = vh; ol[e] = vl;
        }
        *(h8*)(out + (int64_t)row * hidden + i * 8) = o;
        if (out_lo) *(h8*)(out_lo + (int64_t)row * hidden + i * 8) = ol;
    }
}

// nn.GELU() (falcon.py:726), the exact erf form: fp32 in -> fp16 out
__global__ __launch_bounds__(256) void gelu_kernel(const float* __restrict__ x, _Float16* __restrict__ out, int64_t n8,
                                                   const float* __restrict__ x2, _Float16* __restrict__ out_lo) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    f4 a = *(const f4*)(x + i * 8), c = *(const f4*)(x + i * 8 + 4);
    if (x2) {
        const f4 a2 = *(const f4*)(x2 + i * 8), c2 = *(const f4*)(x2 + i * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] += a2[e]; c[e] += c2[e]; }
    }
    h8 o, ol;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = e < 4 ? a[e] : c[e - 4];
        _Float16 vh, vl;
        pc_split(0.5f * v * (1.0f + erff(v * 0.70710678118654752f)), vh, vl);
        o[e] = vh; ol[e] = vl;
    }
    *(h8*)(out + i * 8) = o;
    if (out_lo) *(h8*)(out_lo + i * 8) = ol;
}

// x += a + b (fp32): the two halves of a split-precision projection folded into the residual stream in one pass
__global__ __launch_bounds__(256) void add3_kernel(float* __restrict__ x, const float* __restrict__ a,
                                                   const float* __restrict__ b, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f4 v = *(const f4*)(x + i * 4);
    const f4 p = *(const f4*)(a + i * 4), q = *(const f4*)(b + i * 4);
    v[0] += p[0] + q[0]; v[1] += p[1] + q[1]; v[2] += p[2] + q[2]; v[3] += p[3] + q[3];
    *(f4*)(x + i * 4) = v;
}

__global__ __launch_bounds__(256) void embed_gather_kernel(const _Float16* __restrict__ table,
                                                           const int64_t* __restrict__ ids, _Float16* __restrict__ out,
                                                           int hidden, int vocab) {
    const int t = blockIdx.x;
    int64_t id = ids[t];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    for (int i = threadIdx.x; i < (hidden >> 3); i += 256)
        *(h8*)(out + (int64_t)t * hidden + i * 8) = *(const h8*)(table + id * hidden + i * 8);
}

// ---- layout probe -------------------------------------------------------------------------------------
// out_mfma[lane*4+r]: D register r of lane for D = A.B with A[m][slot0] = m+1, B[slot0][n] = n+17 (all other
//   k-slots zero) -> expected (row+1)*(col+17) with row = 4*(lane>>4)+r, col = lane&15 (asymmetric on purpose).
// out_tr[0..255]:   ds_read_b64_tr_b16 with lane l addressing elements [4l, 4l+4) of an iota array.
// out_tr[256..511]: the same with the attention kernel's addressing (row stride 128 elements).
__global__ void probe_kernel(float* out_mfma, float* out_tr) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[4096];
    const int l = threadIdx.x, n = l & 15, g = l >> 4;
    for (int i = l; i < 4096; i += 64) lds[i] = (_Float16)(float)(i & 2047);
    __syncthreads();
    h8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    if (g == 0) { a[0] = (_Float16)(float)(n + 1); b[0] = (_Float16)(float)(n + 17); }
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out_mfma[l * 4 + r] = acc[r];
    s4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + l * 4));
    h4 h0 = __builtin_bit_cast(h4, t0);
    for (int j = 0; j < 4; ++j) out_tr[l * 4 + j] = (float)h0[j];
    s4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s4*)(lds + (g * 4 + (n >> 2)) * 128 + (n & 3) * 4));
    h4 h1 = __builtin_bit_cast(h4, t1);
    for (int j = 0; j < 4; ++j) out_tr[256 + l * 4 + j] = (float)h1[j];
}


// ---- greedy decode step tail, on the device ------------------------------------------------------------------------
// token = argmax(logits[0..V)) -- the lowest index among equal maxima, torch.argmax's answer on a contiguous row
// (generation_engine.py:159: int(torch.argmax(last_token_logits))) -- stored where the NEXT replay of the captured
// decode graph reads its inputs: ids[0] = token, pos[0] += 1, past[0] += 1; the token also goes to ring[ctr % cap] and
// ctr is incremented.  One workgroup; the whole step stays on the GPU (no host round trip between decode steps).
__global__ __launch_bounds__(1024) void greedy_advance_kernel(const float* __restrict__ logits, int V, int64_t* ids,
                                                              int32_t* pos, int32_t* past, int32_t* ring, int32_t* ctr,
                                                              int ring_cap) {
    __shared__ float sv[16];
    __shared__ int si[16];
    const int tid = threadIdx.x;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid * 4; i < V; i += 4096) {
        if (i + 3 < V) {
            const f4 x = *(const f4*)(logits + i);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (x[e] > best) { best = x[e]; bi = i + e; }
        } else {
            for (int e = 0; i + e < V; ++e)
                if (logits[i + e] > best) { best = logits[i + e]; bi = i + e; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if ((tid & 63) == 0) { sv[tid >> 6] = best; si[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
        if (bi == 0x7fffffff) bi = 0;                 // all -inf / NaN: torch returns index 0 for an all-equal row
        ids[0] = bi;
        pos[0] += 1;
        past[0] += 1;
        const int c = ctr[0];
        ring[c % ring_cap] = bi;
        ctr[0] = c + 1;
    }
}

}  // namespace

PC_EXPORT int pc_rmsnorm(const void* x, const void* weight, void* out, int32_t rows, int32_t hidden, float eps,
                         int32_t x_is_f32, void* stream) {
    PC_REQUIRE(rows >= 0 && hidden > 0 && hidden % 8 == 0, PC_ERR_ARG, "pc_rmsnorm: hidden must be a multiple of 8");
    if (rows == 0) return PC_OK;
    PC_REQUIRE(x && weight && out, PC_ERR_ARG, "pc_rmsnorm: null pointer");
    if (x_is_f32)
        hipLaunchKernelGGL(rmsnorm_kernel<true>, dim3(rows), dim3(256), 0, (hipStream_t)stream, x,
                           (const _Float16*)weight, (_Float16*)out, hidden, eps);
    else
        hipLaunchKernelGGL(rmsnorm_kernel<false>, dim3(rows), dim3(256), 0, (hipStream_t)stream, x,
                           (const _Float16*)weight, (_Float16*)out, hidden, eps);
    return pc_check_launch("rmsnorm_kernel");
}

PC_EXPORT int pc_rmsnorm_split(const float* x, const void* weight, void* out_hi, void* out_lo, int32_t rows, int32_t hidden,
                               float eps, void* stream) {
    PC_REQUIRE(rows >= 0 && hidden > 0 && hidden % 8 == 0, PC_ERR_ARG, "pc_rmsnorm_split: hidden must be a multiple of 8");
    if (rows == 0) return PC_OK;
    PC_REQUIRE(x && weight && out_hi && out_lo, PC_ERR_ARG, "pc_rmsnorm_split: null pointer");
    hipLaunchKernelGGL(rmsnorm_kernel<true>, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const void*)x,
                       (const _Float16*)weight, (_Float16*)out_hi, hidden, eps, (_Float16*)out_lo);
    return pc_check_launch("rmsnorm_kernel");
}

PC_EXPORT int pc_silu_mul_split(const float* gate_up, const float* gate_up2, void* out_hi, void* out_lo, int32_t rows,
                                int32_t inter, void* stream) {
    PC_REQUIRE(rows >= 0 && inter > 0 && inter % 8 == 0, PC_ERR_ARG, "pc_silu_mul_split: inter must be a multiple of 8");
    if (rows == 0) return PC_OK;
    PC_REQUIRE(gate_up && out_hi && out_lo, PC_ERR_ARG, "pc_silu_mul_split: null pointer");
    dim3 grid(pc_ceil_div(inter / 8, 256), rows);
    hipLaunchKernelGGL(silu_mul_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const void*)gate_up, (_Float16*)out_hi,
                       inter, gate_up2, (_Float16*)out_lo);
    return pc_check_launch("silu_mul_kernel");
}

PC_EXPORT int pc_add3(float* x, const float* a, const float* b, int64_t n, void* stream) {
    PC_REQUIRE(n >= 0 && n % 4 == 0, PC_ERR_ARG, "pc_add3: element count must be a multiple of 4");
    if (n == 0) return PC_OK;
    PC_REQUIRE(x && a && b, PC_ERR_ARG, "pc_add3: null pointer");
    const int64_t n4 = n / 4;
    hipLaunchKernelGGL(add3_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, a, b, n4);
    return pc_check_launch("add3_kernel");
}

PC_EXPORT int pc_silu_mul(const void* gate_up, void* out, int32_t rows, int32_t inter, int32_t in_is_f32, void* stream) {
    PC_REQUIRE(rows >= 0 && inter > 0 && inter % 8 == 0, PC_ERR_ARG, "pc_silu_mul: inter must be a multiple of 8");
    if (rows == 0) return PC_OK;
    PC_REQUIRE(gate_up && out, PC_ERR_ARG, "pc_silu_mul: null pointer");
    dim3 grid(pc_ceil_div(inter / 8, 256), rows);
    if (in_is_f32)
        hipLaunchKernelGGL(silu_mul_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, gate_up, (_Float16*)out, inter);
    else
        hipLaunchKernelGGL(silu_mul_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, gate_up, (_Float16*)out, inter);
    return pc_check_launch("silu_mul_kernel");
}

PC_EXPORT int pc_layernorm(const float* x, const void* weight, const void* bias, void* out, int32_t rows, int32_t hidden,
                           float eps, void* stream) {
    PC_REQUIRE(rows >= 0 && hidden > 0 && hidden % 8 == 0, PC_ERR_ARG, "pc_layernorm: hidden must be a multiple of 8");
    if (rows == 0) return PC_OK;
    PC_REQUIRE(x && weight && out, PC_ERR_ARG, "pc_layernorm: null pointer");   /* bias may be NULL */
    hipLaunchKernelGGL(layernorm_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, (const _Float16*)weight,
                       (const _Float16*)bias, (_Float16*)out, hidden, eps, (_Float16*)nullptr);
    return pc_check_launch("layernorm_kernel");
}

PC_EXPORT int pc_layernorm_split(const float* x, const void* weight, const void* bias, void* out_hi, void* out_lo,
                                 int32_t rows, int32_t hidden, float eps, void* stream) {
    PC_REQUIRE(rows >= 0 && hidden > 0 && hidden % 8 == 0, PC_ERR_ARG, "pc_layernorm_split: hidden must be a multiple of 8");
    if (rows == 0) return PC_OK;
    PC_REQUIRE(x && weight && out_hi && out_lo, PC_ERR_ARG, "pc_layernorm_split: null pointer");
    hipLaunchKernelGGL(layernorm_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, (const _Float16*)weight,
                       (const _Float16*)bias, (_Float16*)out_hi, hidden, eps, (_Float16*)out_lo);
    return pc_check_launch("layernorm_kernel");
}

PC_EXPORT int pc_gelu(const float* x, void* out, int64_t n, void* stream) {
    PC_REQUIRE(n >= 0 && n % 8 == 0, PC_ERR_ARG, "pc_gelu: element count must be a multiple of 8");
    if (n == 0) return PC_OK;
    PC_REQUIRE(x && out, PC_ERR_ARG, "pc_gelu: null pointer");
    const int64_t n8 = n / 8;
    hipLaunchKernelGGL(gelu_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (_Float16*)out, n8,
                       (const float*)nullptr, (_Float16*)nullptr);
    return pc_check_launch("gelu_kernel");
}

PC_EXPORT int pc_gelu_split(const float* x, const float* x2, void* out_hi, void* out_lo, int64_t n, void* stream) {
    PC_REQUIRE(n >= 0 && n % 8 == 0, PC_ERR_ARG, "pc_gelu_split: element count must be a multiple of 8");
    if (n == 0) return PC_OK;
    PC_REQUIRE(x && out_hi && out_lo, PC_ERR_ARG, "pc_gelu_split: null pointer");
    const int64_t n8 = n / 8;
    hipLaunchKernelGGL(gelu_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (_Float16*)out_hi,
                       n8, x2, (_Float16*)out_lo);
    return pc_check_launch("gelu_kernel");
}

PC_EXPORT int pc_embed_gather(const void* table, const int64_t* ids, void* out, int32_t n_tok, int32_t hidden,
                              int32_t vocab, void* stream) {
    PC_REQUIRE(n_tok >= 0 && hidden > 0 && hidden % 8 == 0 && vocab > 0, PC_ERR_ARG, "pc_embed_gather: bad sizes");
    if (n_tok == 0) return PC_OK;
    PC_REQUIRE(table && ids && out, PC_ERR_ARG, "pc_embed_gather: null pointer");
    hipLaunchKernelGGL(embed_gather_kernel, dim3(n_tok), dim3(256), 0, (hipStream_t)stream, (const _Float16*)table,
                       ids, (_Float16*)out, hidden, vocab);
    return pc_check_launch("embed_gather_kernel");
}

// pc_fetch_block: the per-call inputs of a captured forward (token ids, position ids, past length, staging plan) come out of
// ONE pinned host block that the graph's first node pulls into device memory itself -- the host only writes the block and
// replays the graph; no copy is enqueued per call (the reference uploads ids and positions with two pageable copies per call,
// generation_engine.py:96-97).  System-scope loads: the block is rewritten by the host between replays and must never be
// served from a device cache.
__global__ __launch_bounds__(256) void fetch_block_kernel(const unsigned long long* __restrict__ src, unsigned long long* __restrict__ dst,
                                                          int n8) {
    typedef __attribute__((address_space(1))) unsigned long long g64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += gridDim.x * blockDim.x)
        dst[i] = __hip_atomic_load((g64*)(src + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

PC_EXPORT int pc_fetch_block(const void* host_src, void* dst, int32_t nbytes, void* stream) {
    PC_REQUIRE(host_src && dst && nbytes > 0 && nbytes % 8 == 0 && nbytes <= (1 << 20), PC_ERR_ARG,
               "pc_fetch_block: need non-null pointers and 8 | nbytes <= 1 MiB");
    PC_REQUIRE((((uintptr_t)host_src | (uintptr_t)dst) & 7) == 0, PC_ERR_ARG, "pc_fetch_block: pointers must be 8-byte aligned");
    const int n8 = nbytes / 8;
    hipLaunchKernelGGL(fetch_block_kernel, dim3(pc_ceil_div(n8, 1024) < 8 ? pc_ceil_div(n8, 1024) : 8), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned long long*)host_src, (unsigned long long*)dst, n8);
    return pc_check_launch("fetch_block_kernel");
}

PC_EXPORT int pc_probe_layouts(float* out_mfma, float* out_tr, void* stream) {
    PC_REQUIRE(out_mfma && out_tr, PC_ERR_ARG, "pc_probe_layouts: null pointer");
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out_mfma, out_tr);
    return pc_check_launch("probe_kernel");
}

PC_EXPORT int pc_greedy_advance(const float* logits, int32_t vocab, int64_t* ids, int32_t* pos, int32_t* past_len, int32_t* ring,
                                int32_t* counter, int32_t ring_cap, void* stream) {
    PC_REQUIRE(logits && ids && pos && past_len && ring && counter && vocab > 0 && ring_cap > 0, PC_ERR_ARG,
               "pc_greedy_advance: null pointer or bad sizes");
    PC_REQUIRE(((uintptr_t)logits & 15) == 0, PC_ERR_ARG, "pc_greedy_advance: logits must be 16-byte aligned");
    hipLaunchKernelGGL(greedy_advance_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits, vocab, ids, pos, past_len,
                       ring, counter, ring_cap);
    return pc_check_launch("greedy_advance_kernel");
}
