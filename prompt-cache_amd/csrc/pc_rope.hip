// Position-id-aware RoPE and in-place KV append.
//
// Replaces  LlamaRotaryEmbedding._set_cos_sin_cache / forward   promptcache/model/llama2.py:129-147
//           apply_rotary_pos_emb (cos[position_ids] gather)      promptcache/model/llama2.py:202-210
//           torch.cat([past, new], dim=2) for K and V            promptcache/model/llama2.py:361-364
//
// The reference sizes its cos/sin table by max(position_ids)+1 (llama2.py:357) and gathers rows by the
// supplied ids; here only the gathered rows are ever computed (pc_rope_table, once per forward, fp32),
// and every layer applies them to q (in place) and k while writing k/v straight into rows
// [past_len, past_len+q_len) of the layer's KV arena -- the staged past is never re-copied.
//
// Both kernels are tiny (q_len x (H+2*Hkv) x D elements) and launch-latency-bound.
#include <hip/hip_fp16.h>

#include "pc_common.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void rope_table_kernel(const int32_t* __restrict__ pos, const float* __restrict__ inv_freq,
                                  float2* __restrict__ cs, int n_tok, int half_dim) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tok * half_dim) return;
    const int t = i / half_dim, f = i - t * half_dim;
    // angle = fp32(pos) * inv_freq, one fp32 rounding, as torch.einsum("i,j->ij", t, inv_freq) (llama2.py:133)
    const float ang = __fmul_rn((float)pos[t], inv_freq[f]);
    float s, c;
    sincosf(ang, &s, &c);  // full-range argument reduction (positions reach 1e4 rad)
    cs[i] = make_float2(c, s);
}

// One work item = 8 rotary pairs of one (token, head): 8 values from the low half + 8 from the high half.
// Items per token: H*D/16 for q, Hkv*D/16 for k, then Hkv*D/8 plain copies for v.  TIn is the projection
// output type: fp32 (GEMMs run with fp32 outputs so q/k/v are rounded to fp16 once, after the rotation)
// or fp16 (q may then be rotated in place, q_out == q).
template <typename TIn>
__device__ __forceinline__ void load8(const TIn* p, float (&x)[8]);
template <>
__device__ __forceinline__ void load8<_Float16>(const _Float16* p, float (&x)[8]) {
    const h8 v = *(const h8*)p;
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (float)v[e];
}
template <>
__device__ __forceinline__ void load8<float>(const float* p, float (&x)[8]) {
    const f4 a = *(const f4*)p, b = *(const f4*)(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { x[e] = a[e]; x[e + 4] = b[e]; }
}

template <typename TIn>
__global__ __launch_bounds__(256) void rope_append_kernel(
    const TIn* __restrict__ q, int64_t q_bs, int64_t q_ts, _Float16* __restrict__ q_out, _Float16* __restrict__ q_out_lo,
    int64_t qo_bs, int64_t qo_ts,
    const TIn* __restrict__ k_new, const TIn* __restrict__ v_new, int64_t n_bs, int64_t n_ts,
    _Float16* __restrict__ k_arena, _Float16* __restrict__ v_arena, int64_t a_bs, int64_t a_hs,
    const float2* __restrict__ cs, int H, int Hkv, int D, int q_len, int past_len,
    const int32_t* __restrict__ past_len_dev, _Float16* __restrict__ k_lo, _Float16* __restrict__ v_lo, int64_t lo_bs,
    int64_t lo_hs, int lo_row0, int64_t in2, const int32_t* __restrict__ past_lens) {
    // past_lens (optional, [B]): every batch row appends behind its OWN past length (ragged prefixes: scaffold suffixes
    // of different unions encoded in one batch over their trunk prefixes, cache_engine.py SchemaCache._process)
    // in2 != 0: every input element is the SUM x[i] + x[i + in2] -- the two row halves a stacked [hi; lo] projection leaves
    // k_lo / v_lo (optional): fp16 residuals of the appended K / V rows, [B][Hkv][rows][D] with strides lo_bs / lo_hs,
    // row = key index - lo_row0 (lo_row0 = past_len: compact, new rows only; 0: arena-shaped) -- the pass's own
    // keys in split precision for the attention of that pass (the arena keeps the fp16 value the reference stages)
    const int t = blockIdx.x, b = blockIdx.y;
    if (past_len_dev) past_len = *past_len_dev;
    if (past_lens) past_len = past_lens[b];
    const int half = D >> 1;
    const int cph = D >> 4;  // 8-pair chunks per head
    const int nq = H * cph, nk = Hkv * cph, nv = Hkv * (D >> 3);
    const float2* csr = cs + (int64_t)(b * q_len + t) * half;
    for (int it = threadIdx.x; it < nq + nk + nv; it += blockDim.x) {
        if (it < nq + nk) {
            const bool is_q = it < nq;
            const int j = is_q ? it : it - nq;
            const int h = j / cph, c = j - h * cph;
            const TIn* src = is_q ? q + b * q_bs + t * q_ts + (int64_t)h * D
                                  : k_new + b * n_bs + t * n_ts + (int64_t)h * D;
            _Float16* dst = is_q ? q_out + b * qo_bs + t * qo_ts + (int64_t)h * D
                                 : k_arena + b * a_bs + h * a_hs + (int64_t)(past_len + t) * D;
            float lo[8], hi[8];
            load8<TIn>(src + c * 8, lo);
            load8<TIn>(src + half + c * 8, hi);
            if (in2) {
                float lo2[8], hi2[8];
                load8<TIn>(src + in2 + c * 8, lo2);
                load8<TIn>(src + in2 + half + c * 8, hi2);
#pragma unroll
                for (int e = 0; e < 8; ++e) { lo[e] += lo2[e]; hi[e] += hi2[e]; }
            }
            h8 olo, ohi, rlo, rhi;   // r*: low-order residuals (second plane of the split-precision q)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float2 w = csr[c * 8 + e];
                // q*cos + rotate_half(q)*sin  (llama2.py:208): low half pairs with -high, high with +low
                const float a = lo[e] * w.x - hi[e] * w.y, b2 = hi[e] * w.x + lo[e] * w.y;
                _Float16 t0, t1, t2, t3;
                pc_split(a, t0, t1);
                pc_split(b2, t2, t3);
                olo[e] = t0; rlo[e] = t1;
                ohi[e] = t2; rhi[e] = t3;
            }
            *(h8*)(dst + c * 8) = olo;
            *(h8*)(dst + half + c * 8) = ohi;
            if (is_q && q_out_lo) {
                _Float16* dl = q_out_lo + b * qo_bs + t * qo_ts + (int64_t)h * D;
                *(h8*)(dl + c * 8) = rlo;
                *(h8*)(dl + half + c * 8) = rhi;
            }
            if (!is_q && k_lo) {
                _Float16* dl = k_lo + b * lo_bs + h * lo_hs + (int64_t)(past_len + t - lo_row0) * D;
                *(h8*)(dl + c * 8) = rlo;
                *(h8*)(dl + half + c * 8) = rhi;
            }
        } else {
            const int j = it - nq - nk;
            const int cpv = D >> 3;
            const int h = j / cpv, c = j - h * cpv;
            float x[8];
            load8<TIn>(v_new + b * n_bs + t * n_ts + (int64_t)h * D + c * 8, x);
            if (in2) {
                float x2[8];
                load8<TIn>(v_new + in2 + b * n_bs + t * n_ts + (int64_t)h * D + c * 8, x2);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] += x2[e];
            }
            h8 o, ol;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[e] = (_Float16)x[e];
                ol[e] = (_Float16)(x[e] - (float)o[e]);
            }
            *(h8*)(v_arena + b * a_bs + h * a_hs + (int64_t)(past_len + t) * D + c * 8) = o;
            if (v_lo) *(h8*)(v_lo + b * lo_bs + h * lo_hs + (int64_t)(past_len + t - lo_row0) * D + c * 8) = ol;
        }
    }
}

}  // namespace

PC_EXPORT int pc_rope_table(const int32_t* pos, const float* inv_freq, float* cs, int32_t n_tok, int32_t head_dim,
                            void* stream) {
    PC_REQUIRE(n_tok >= 0 && head_dim > 0 && head_dim % 2 == 0, PC_ERR_ARG, "pc_rope_table: bad sizes");
    if (n_tok == 0) return PC_OK;
    PC_REQUIRE(pos && inv_freq && cs, PC_ERR_ARG, "pc_rope_table: null pointer");
    const int n = n_tok * (head_dim / 2);
    hipLaunchKernelGGL(rope_table_kernel, dim3(pc_ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, pos,
                       inv_freq, (float2*)cs, n_tok, head_dim / 2);
    return pc_check_launch("rope_table_kernel");
}

namespace {
int rope_append_impl(const void* q, int64_t q_batch_stride, int64_t q_token_stride, void* q_out, void* q_out_lo,
                     int64_t qo_batch_stride, int64_t qo_token_stride, const void* k_new,
                     const void* v_new, int64_t kv_new_batch_stride, int64_t kv_new_token_stride,
                     void* k_arena, void* v_arena, int64_t arena_batch_stride, int64_t arena_head_stride,
                     const float* cs, int32_t B, int32_t H, int32_t Hkv, int32_t D, int32_t q_len,
                     int32_t past_len, int32_t cap, int32_t in_is_f32, const int32_t* past_len_dev,
                     void* k_lo, void* v_lo, int64_t lo_bs, int64_t lo_hs, int32_t lo_row0, int64_t in2, void* stream,
                     const int32_t* past_lens = nullptr) {
    PC_REQUIRE(B > 0 && H > 0 && Hkv > 0 && q_len >= 0 && past_len >= 0, PC_ERR_ARG, "pc_rope_append: bad sizes");
    PC_REQUIRE(D > 0 && D % 16 == 0, PC_ERR_ARG, "pc_rope_append: head_dim must be a multiple of 16");
    if (q_len == 0) return PC_OK;
    PC_REQUIRE(q && q_out && k_new && v_new && k_arena && v_arena && cs, PC_ERR_ARG, "pc_rope_append: null pointer");
    PC_REQUIRE((int64_t)past_len + q_len <= cap, PC_ERR_BOUNDS,
               "pc_rope_append: past_len %d + q_len %d exceeds arena rows %d", past_len, q_len, cap);
    PC_REQUIRE(q_token_stride % 8 == 0 && qo_token_stride % 8 == 0 && kv_new_token_stride % 8 == 0 &&
                   arena_head_stride % 8 == 0, PC_ERR_ARG, "pc_rope_append: strides must keep 16-byte alignment");
    if (in_is_f32)
        hipLaunchKernelGGL(rope_append_kernel<float>, dim3(q_len, B), dim3(256), 0, (hipStream_t)stream,
                           (const float*)q, q_batch_stride, q_token_stride, (_Float16*)q_out, (_Float16*)q_out_lo, qo_batch_stride,
                           qo_token_stride, (const float*)k_new, (const float*)v_new, kv_new_batch_stride,
                           kv_new_token_stride, (_Float16*)k_arena, (_Float16*)v_arena, arena_batch_stride,
                           arena_head_stride, (const float2*)cs, H, Hkv, D, q_len, past_len, past_len_dev, (_Float16*)k_lo,
                           (_Float16*)v_lo, lo_bs, lo_hs, lo_row0, in2, past_lens);
    else
        hipLaunchKernelGGL(rope_append_kernel<_Float16>, dim3(q_len, B), dim3(256), 0, (hipStream_t)stream,
                           (const _Float16*)q, q_batch_stride, q_token_stride, (_Float16*)q_out, (_Float16*)q_out_lo, qo_batch_stride,
                           qo_token_stride, (const _Float16*)k_new, (const _Float16*)v_new, kv_new_batch_stride,
                           kv_new_token_stride, (_Float16*)k_arena, (_Float16*)v_arena, arena_batch_stride,
                           arena_head_stride, (const float2*)cs, H, Hkv, D, q_len, past_len, past_len_dev, (_Float16*)k_lo,
                           (_Float16*)v_lo, lo_bs, lo_hs, lo_row0, in2, past_lens);
    return pc_check_launch("rope_append_kernel");
}
}  // namespace

PC_EXPORT int pc_rope_append(const void* q, int64_t q_batch_stride, int64_t q_token_stride, void* q_out, void* q_out_lo,
                             int64_t qo_batch_stride, int64_t qo_token_stride, const void* k_new,
                             const void* v_new, int64_t kv_new_batch_stride, int64_t kv_new_token_stride,
                             void* k_arena, void* v_arena, int64_t arena_batch_stride, int64_t arena_head_stride,
                             const float* cs, int32_t B, int32_t H, int32_t Hkv, int32_t D, int32_t q_len,
                             int32_t past_len, int32_t cap, int32_t in_is_f32, const int32_t* past_len_dev,
                             void* stream) {
    return rope_append_impl(q, q_batch_stride, q_token_stride, q_out, q_out_lo, qo_batch_stride, qo_token_stride, k_new, v_new,
                            kv_new_batch_stride, kv_new_token_stride, k_arena, v_arena, arena_batch_stride,
                            arena_head_stride, cs, B, H, Hkv, D, q_len, past_len, cap, in_is_f32, past_len_dev, nullptr,
                            nullptr, 0, 0, 0, 0, stream);
}

PC_EXPORT int pc_rope_append_ex(const void* q, int64_t q_batch_stride, int64_t q_token_stride, void* q_out, void* q_out_lo,
                                int64_t qo_batch_stride, int64_t qo_token_stride, const void* k_new,
                                const void* v_new, int64_t kv_new_batch_stride, int64_t kv_new_token_stride,
                                void* k_arena, void* v_arena, int64_t arena_batch_stride, int64_t arena_head_stride,
                                const float* cs, int32_t B, int32_t H, int32_t Hkv, int32_t D, int32_t q_len,
                                int32_t past_len, int32_t cap, int32_t in_is_f32, const int32_t* past_len_dev,
                                void* k_lo, void* v_lo, int64_t lo_batch_stride, int64_t lo_head_stride, int32_t lo_row0,
                                int64_t in2_offset, void* stream) {
    PC_REQUIRE((k_lo == nullptr) == (v_lo == nullptr), PC_ERR_ARG, "pc_rope_append_ex: k_lo and v_lo go together");
    PC_REQUIRE(in2_offset % 8 == 0, PC_ERR_ARG, "pc_rope_append_ex: in2_offset must keep 16-byte alignment");
    PC_REQUIRE(!k_lo || (lo_row0 >= 0 && lo_row0 <= past_len && lo_head_stride % 8 == 0), PC_ERR_ARG,
               "pc_rope_append_ex: lo_row0 must lie in [0, past_len] and the lo strides keep 16-byte alignment");
    return rope_append_impl(q, q_batch_stride, q_token_stride, q_out, q_out_lo, qo_batch_stride, qo_token_stride, k_new, v_new,
                            kv_new_batch_stride, kv_new_token_stride, k_arena, v_arena, arena_batch_stride,
                            arena_head_stride, cs, B, H, Hkv, D, q_len, past_len, cap, in_is_f32, past_len_dev, k_lo, v_lo,
                            lo_batch_stride, lo_head_stride, lo_row0, in2_offset, stream);
}

// pc_rope_append_ex with one past length PER BATCH ROW (device int32[B]; `past_len` is then their maximum, used for the
// bounds check only): row b appends its q_len new K / V rows at arena rows [past_lens[b], past_lens[b] + q_len).
PC_EXPORT int pc_rope_append_var(const void* q, int64_t q_batch_stride, int64_t q_token_stride, void* q_out, void* q_out_lo,
                                 int64_t qo_batch_stride, int64_t qo_token_stride, const void* k_new,
                                 const void* v_new, int64_t kv_new_batch_stride, int64_t kv_new_token_stride,
                                 void* k_arena, void* v_arena, int64_t arena_batch_stride, int64_t arena_head_stride,
                                 const float* cs, int32_t B, int32_t H, int32_t Hkv, int32_t D, int32_t q_len,
                                 int32_t past_len, int32_t cap, int32_t in_is_f32, const int32_t* past_lens,
                                 void* k_lo, void* v_lo, int64_t lo_batch_stride, int64_t lo_head_stride, int32_t lo_row0,
                                 void* stream) {
    PC_REQUIRE(past_lens, PC_ERR_ARG, "pc_rope_append_var: past_lens is required");
    PC_REQUIRE((k_lo == nullptr) == (v_lo == nullptr), PC_ERR_ARG, "pc_rope_append_var: k_lo and v_lo go together");
    PC_REQUIRE(!k_lo || (lo_row0 == 0 && lo_head_stride % 8 == 0), PC_ERR_ARG,
               "pc_rope_append_var: residual planes must be arena-shaped (lo_row0 = 0) with 16-byte aligned strides");
    return rope_append_impl(q, q_batch_stride, q_token_stride, q_out, q_out_lo, qo_batch_stride, qo_token_stride, k_new, v_new,
                            kv_new_batch_stride, kv_new_token_stride, k_arena, v_arena, arena_batch_stride,
                            arena_head_stride, cs, B, H, Hkv, D, q_len, past_len, cap, in_is_f32, nullptr, k_lo, v_lo,
                            lo_batch_stride, lo_head_stride, lo_row0, 0, stream, past_lens);
}
