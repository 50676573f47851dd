/*
 * promptcache_hip_compat.h -- the round 1-2 entry-point NAMES of the attention and weight-streaming projection families as
 * inline wrappers over the two struct-taking entry points that libpromptcache_hip.so exports now (pc_attn, pc_gemm;
 * promptcache_hip.h).  Nothing here is exported by the library; every wrapper only fills the struct.  Each one replaces the
 * same reference ops as the field set it fills (promptcache_hip.h cites them: llama2.py:345-347, :357-364, :368-398, :405,
 * :242, :638, :644, :1050).
 */
#ifndef PROMPTCACHE_HIP_COMPAT_H
#define PROMPTCACHE_HIP_COMPAT_H

#include <string.h>

#include "promptcache_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- attention: pc_attn_fwd / _alibi / _ex / _var -------------------------------------------------------------------- */
static inline pc_attn_args pc_compat_attn_(const void* q, const void* q_lo, int64_t q_bs, int64_t q_ts, const void* k, const void* v,
                                           int64_t kv_bs, int64_t kv_hs, void* out, void* out_lo, int64_t o_bs, int64_t o_ts,
                                           int32_t B, int32_t H, int32_t Hkv, int32_t D, int32_t q_len, int32_t past_len,
                                           float softmax_scale, void* workspace, int64_t workspace_bytes) {
    pc_attn_args a;
    memset(&a, 0, sizeof(a));
    a.struct_bytes = (uint32_t)sizeof(a);
    a.q = q; a.q_lo = q_lo; a.q_batch_stride = q_bs; a.q_token_stride = q_ts;
    a.k = k; a.v = v; a.kv_batch_stride = kv_bs; a.kv_head_stride = kv_hs;
    a.out = out; a.out_lo = out_lo; a.out_batch_stride = o_bs; a.out_token_stride = o_ts;
    a.B = B; a.H = H; a.Hkv = Hkv; a.D = D; a.q_len = q_len; a.past_len = past_len; a.softmax_scale = softmax_scale;
    a.workspace = workspace; a.workspace_bytes = workspace_bytes;
    return a;
}

static inline int pc_attn_fwd(const void* q, const void* q_lo, int64_t q_batch_stride, int64_t q_token_stride, const void* k,
                              const void* v, int64_t kv_batch_stride, int64_t kv_head_stride, void* out,
                              int64_t out_batch_stride, int64_t out_token_stride, int32_t B, int32_t H, int32_t Hkv, int32_t D,
                              int32_t q_len, int32_t past_len, float softmax_scale, void* workspace, int64_t workspace_bytes,
                              const int32_t* past_len_dev, void* out_frag_hi, void* out_frag_lo, void* stream) {
    pc_attn_args a = pc_compat_attn_(q, q_lo, q_batch_stride, q_token_stride, k, v, kv_batch_stride, kv_head_stride, out, NULL,
                                     out_batch_stride, out_token_stride, B, H, Hkv, D, q_len, past_len, softmax_scale, workspace,
                                     workspace_bytes);
    a.past_len_dev = past_len_dev; a.out_frag_hi = out_frag_hi; a.out_frag_lo = out_frag_lo;
    return pc_attn(&a, stream);
}

static inline int pc_attn_fwd_alibi(const void* q, const void* q_lo, int64_t q_batch_stride, int64_t q_token_stride,
                                    const void* k, const void* v, int64_t kv_batch_stride, int64_t kv_head_stride, void* out,
                                    int64_t out_batch_stride, int64_t out_token_stride, int32_t B, int32_t H, int32_t Hkv,
                                    int32_t D, int32_t q_len, int32_t past_len, float softmax_scale, void* workspace,
                                    int64_t workspace_bytes, const int32_t* past_len_dev, void* out_frag_hi, void* out_frag_lo,
                                    const float* key_pos, int64_t key_pos_batch_stride, const float* slopes_log2, void* stream) {
    pc_attn_args a = pc_compat_attn_(q, q_lo, q_batch_stride, q_token_stride, k, v, kv_batch_stride, kv_head_stride, out, NULL,
                                     out_batch_stride, out_token_stride, B, H, Hkv, D, q_len, past_len, softmax_scale, workspace,
                                     workspace_bytes);
    a.past_len_dev = past_len_dev; a.out_frag_hi = out_frag_hi; a.out_frag_lo = out_frag_lo;
    a.key_pos = key_pos; a.key_pos_batch_stride = key_pos_batch_stride; a.slopes_log2 = slopes_log2;
    return pc_attn(&a, stream);
}

static inline int pc_attn_fwd_ex(const void* q, const void* q_lo, int64_t q_batch_stride, int64_t q_token_stride, const void* k,
                                 const void* v, int64_t kv_batch_stride, int64_t kv_head_stride, void* out, void* out_lo,
                                 int64_t out_batch_stride, int64_t out_token_stride, int32_t B, int32_t H, int32_t Hkv, int32_t D,
                                 int32_t q_len, int32_t past_len, float softmax_scale, void* workspace, int64_t workspace_bytes,
                                 const int32_t* past_len_dev, const float* key_pos, int64_t key_pos_batch_stride,
                                 const float* slopes_log2, const void* k_lo, const void* v_lo, int64_t lo_batch_stride,
                                 int64_t lo_head_stride, int32_t lo_row0, void* out_frag_hi, void* out_frag_lo, void* stream) {
    pc_attn_args a = pc_compat_attn_(q, q_lo, q_batch_stride, q_token_stride, k, v, kv_batch_stride, kv_head_stride, out, out_lo,
                                     out_batch_stride, out_token_stride, B, H, Hkv, D, q_len, past_len, softmax_scale, workspace,
                                     workspace_bytes);
    a.past_len_dev = past_len_dev; a.out_frag_hi = out_frag_hi; a.out_frag_lo = out_frag_lo;
    a.key_pos = key_pos; a.key_pos_batch_stride = key_pos_batch_stride; a.slopes_log2 = slopes_log2;
    a.k_lo = k_lo; a.v_lo = v_lo; a.lo_batch_stride = lo_batch_stride; a.lo_head_stride = lo_head_stride; a.lo_row0 = lo_row0;
    return pc_attn(&a, stream);
}

static inline int pc_attn_fwd_var(const void* q, const void* q_lo, int64_t q_batch_stride, int64_t q_token_stride, const void* k,
                                  const void* v, int64_t kv_batch_stride, int64_t kv_head_stride, void* out, void* out_lo,
                                  int64_t out_batch_stride, int64_t out_token_stride, int32_t B, int32_t H, int32_t Hkv, int32_t D,
                                  int32_t q_len, int32_t past_len, const int32_t* past_lens, float softmax_scale, void* workspace,
                                  int64_t workspace_bytes, const void* k_lo, const void* v_lo, int64_t lo_batch_stride,
                                  int64_t lo_head_stride, void* stream) {
    pc_attn_args a = pc_compat_attn_(q, q_lo, q_batch_stride, q_token_stride, k, v, kv_batch_stride, kv_head_stride, out, out_lo,
                                     out_batch_stride, out_token_stride, B, H, Hkv, D, q_len, past_len, softmax_scale, workspace,
                                     workspace_bytes);
    a.past_lens = past_lens;
    a.k_lo = k_lo; a.v_lo = v_lo; a.lo_batch_stride = lo_batch_stride; a.lo_head_stride = lo_head_stride; a.lo_row0 = 0;
    return pc_attn(&a, stream);
}

/* ---- projections: pc_gemm_skinny / _norm / _w8 / _norm_w8 / _a8 / _a8c / _ks, pc_gemm_qkv_rope / _norm / _w8 / _ex / _a8 / _a8c --- */
static inline pc_gemm_args pc_compat_gemm_(int32_t epilogue, const void* wf, const float* w_scale, const void* xf_hi,
                                           const void* xf_lo, const float* x, const void* norm_weight, float eps, int32_t M,
                                           int32_t N, int32_t K) {
    pc_gemm_args a;
    memset(&a, 0, sizeof(a));
    a.struct_bytes = (uint32_t)sizeof(a);
    a.epilogue = epilogue; a.wf = wf; a.w_scale = w_scale; a.xf_hi = xf_hi; a.xf_lo = xf_lo; a.x = x; a.norm_weight = norm_weight;
    a.eps = eps; a.M = M; a.N = N; a.K = K; a.kslices = 1; a.lo_base = -1;
    return a;
}

static inline int pc_gemm_skinny(const void* wf, const void* xf_hi, const void* xf_lo, int32_t M, int32_t N, int32_t K,
                                 int32_t epilogue, float* y, int64_t ldy, void* of_hi, void* of_lo, int32_t kslices, void* stream) {
    pc_gemm_args a = pc_compat_gemm_(epilogue, wf, NULL, xf_hi, xf_lo, NULL, NULL, 0.f, M, N, K);
    a.y = y; a.ldy = ldy; a.of_hi = of_hi; a.of_lo = of_lo; a.kslices = kslices;
    return pc_gemm(&a, stream);
}

static inline int pc_gemm_skinny_norm(const void* wf, const float* x, const void* norm_weight, float eps, int32_t M, int32_t N,
                                      int32_t K, int32_t epilogue, float* y, int64_t ldy, void* of_hi, void* of_lo, void* stream) {
    pc_gemm_args a = pc_compat_gemm_(epilogue, wf, NULL, NULL, NULL, x, norm_weight, eps, M, N, K);
    a.y = y; a.ldy = ldy; a.of_hi = of_hi; a.of_lo = of_lo;
    return pc_gemm(&a, stream);
}

static inline int pc_gemm_skinny_w8(const void* wf8, const float* w_scale, const void* xf_hi, const void* xf_lo, int32_t M,
                                    int32_t N, int32_t K, int32_t epilogue, float* y, int64_t ldy, void* of_hi, void* of_lo,
                                    int32_t kslices, void* stream) {
    pc_gemm_args a = pc_compat_gemm_(epilogue, wf8, w_scale, xf_hi, xf_lo, NULL, NULL, 0.f, M, N, K);
    a.y = y; a.ldy = ldy; a.of_hi = of_hi; a.of_lo = of_lo; a.kslices = kslices;
    return pc_gemm(&a, stream);
}

static inline int pc_gemm_skinny_norm_w8(const void* wf8, const float* w_scale, const float* x, const void* norm_weight, float eps,
                                         int32_t M, int32_t N, int32_t K, int32_t epilogue, float* y, int64_t ldy, void* of_hi,
                                         void* of_lo, void* stream) {
    pc_gemm_args a = pc_compat_gemm_(epilogue, wf8, w_scale, NULL, NULL, x, norm_weight, eps, M, N, K);
    a.y = y; a.ldy = ldy; a.of_hi = of_hi; a.of_lo = of_lo;
    return pc_gemm(&a, stream);
}

static inline int pc_gemm_skinny_a8(const void* wf8, const float* w_scale, const void* xq_hi, const void* xq_lo,
                                    const float* x_scale, const float* corr, int64_t ldc, const int32_t* corr_has, int32_t M,
                                    int32_t N, int32_t K, int32_t epilogue, float* y, int64_t ldy, void* of_hi, void* of_lo,
                                    void* stream) {
    pc_gemm_args a = pc_compat_gemm_(epilogue, wf8, w_scale, xq_hi, xq_lo, NULL, NULL, 0.f, M, N, K);
    a.y = y; a.ldy = ldy; a.of_hi = of_hi; a.of_lo = of_lo;
    a.x_scale = x_scale; a.corr = corr; a.ldc = ldc; a.corr_has = corr_has;
    return pc_gemm(&a, stream);
}

static inline int pc_gemm_skinny_a8c(const void* wf8, const float* w_scale, const void* xq_hi, const void* xq_lo,
                                     const float* x_scale, const void* flags, const void* x_raw, const void* w_codes_t,
                                     int64_t ldt, int32_t M, int32_t N, int32_t K, int32_t epilogue, float* y, int64_t ldy,
                                     void* of_hi, void* of_lo, void* stream) {
    pc_gemm_args a = pc_compat_gemm_(epilogue, wf8, w_scale, xq_hi, xq_lo, NULL, NULL, 0.f, M, N, K);
    a.y = y; a.ldy = ldy; a.of_hi = of_hi; a.of_lo = of_lo;
    a.x_scale = x_scale; a.flags = flags; a.x_raw = x_raw; a.w_codes_t = w_codes_t; a.ldt = ldt;
    return pc_gemm(&a, stream);
}

static inline int pc_gemm_skinny_ks(const void* wf, const void* xf_hi, const void* xf_lo, int32_t M, int32_t N, int32_t K,
                                    float* y, int64_t ldy, int32_t kslices, int32_t tiles_per_wg, void* scratch,
                                    int64_t scratch_bytes, void* counters, void* stream) {
    pc_gemm_args a = pc_compat_gemm_(PC_GEMM_EPI_ADD, wf, NULL, xf_hi, xf_lo, NULL, NULL, 0.f, M, N, K);
    a.y = y; a.ldy = ldy; a.kslices = kslices; a.ks_tiles = tiles_per_wg; a.ks_scratch = scratch;
    a.ks_scratch_bytes = scratch_bytes; a.ks_counters = counters;
    return pc_gemm(&a, stream);
}

/* pc_gemm_qkv_rope_ex is the union of the q|k|v forms (w_scale_perm NULL or not; planes or x + norm_weight; lo_base) */
static inline int pc_gemm_qkv_rope_ex(const void* wf_perm, const float* w_scale_perm, const void* xf_hi, const void* xf_lo,
                                      const float* x, const void* norm_weight, float eps, int32_t M, int32_t K, const float* cs,
                                      void* q_hi, void* q_lo, int64_t q_token_stride, void* k_arena, void* v_arena,
                                      int64_t arena_batch_stride, int64_t arena_head_stride, int32_t B, int32_t H, int32_t Hkv,
                                      int32_t D, int32_t q_len, int32_t past_len, int32_t cap, const int32_t* past_len_dev,
                                      void* k_lo, void* v_lo, int64_t lo_batch_stride, int64_t lo_head_stride, int32_t lo_base,
                                      void* stream) {
    pc_gemm_args a = pc_compat_gemm_(PC_GEMM_EPI_QKV_ROPE, wf_perm, w_scale_perm, xf_hi, xf_lo, x, norm_weight, eps, M, 0, K);
    a.cs = cs; a.q_hi = q_hi; a.q_lo = q_lo; a.q_token_stride = q_token_stride; a.k_arena = k_arena; a.v_arena = v_arena;
    a.arena_batch_stride = arena_batch_stride; a.arena_head_stride = arena_head_stride;
    a.B = B; a.H = H; a.Hkv = Hkv; a.D = D; a.q_len = q_len; a.past_len = past_len; a.cap = cap; a.past_len_dev = past_len_dev;
    a.k_lo = k_lo; a.v_lo = v_lo; a.lo_batch_stride = lo_batch_stride; a.lo_head_stride = lo_head_stride; a.lo_base = lo_base;
    return pc_gemm(&a, stream);
}

static inline int pc_gemm_qkv_rope(const void* wf_perm, const void* xf_hi, const void* xf_lo, int32_t M, int32_t K,
                                   const float* cs, void* q_hi, void* q_lo, int64_t q_token_stride, void* k_arena, void* v_arena,
                                   int64_t arena_batch_stride, int64_t arena_head_stride, int32_t B, int32_t H, int32_t Hkv,
                                   int32_t D, int32_t q_len, int32_t past_len, int32_t cap, const int32_t* past_len_dev,
                                   void* k_lo, void* v_lo, int64_t lo_batch_stride, int64_t lo_head_stride, void* stream) {
    return pc_gemm_qkv_rope_ex(wf_perm, NULL, xf_hi, xf_lo, NULL, NULL, 0.f, M, K, cs, q_hi, q_lo, q_token_stride, k_arena, v_arena,
                               arena_batch_stride, arena_head_stride, B, H, Hkv, D, q_len, past_len, cap, past_len_dev, k_lo, v_lo,
                               lo_batch_stride, lo_head_stride, -1, stream);
}

static inline int pc_gemm_qkv_rope_norm(const void* wf_perm, const float* x, const void* norm_weight, float eps, int32_t M,
                                        int32_t K, const float* cs, void* q_hi, void* q_lo, int64_t q_token_stride,
                                        void* k_arena, void* v_arena, int64_t arena_batch_stride, int64_t arena_head_stride,
                                        int32_t B, int32_t H, int32_t Hkv, int32_t D, int32_t q_len, int32_t past_len, int32_t cap,
                                        const int32_t* past_len_dev, void* k_lo, void* v_lo, int64_t lo_batch_stride,
                                        int64_t lo_head_stride, void* stream) {
    return pc_gemm_qkv_rope_ex(wf_perm, NULL, NULL, NULL, x, norm_weight, eps, M, K, cs, q_hi, q_lo, q_token_stride, k_arena,
                               v_arena, arena_batch_stride, arena_head_stride, B, H, Hkv, D, q_len, past_len, cap, past_len_dev,
                               k_lo, v_lo, lo_batch_stride, lo_head_stride, -1, stream);
}

static inline int pc_gemm_qkv_rope_w8(const void* wf8_perm, const float* w_scale_perm, const void* xf_hi, const void* xf_lo,
                                      const float* x, const void* norm_weight, float eps, int32_t M, int32_t K, const float* cs,
                                      void* q_hi, void* q_lo, int64_t q_token_stride, void* k_arena, void* v_arena,
                                      int64_t arena_batch_stride, int64_t arena_head_stride, int32_t B, int32_t H, int32_t Hkv,
                                      int32_t D, int32_t q_len, int32_t past_len, int32_t cap, const int32_t* past_len_dev,
                                      void* k_lo, void* v_lo, int64_t lo_batch_stride, int64_t lo_head_stride, void* stream) {
    return pc_gemm_qkv_rope_ex(wf8_perm, w_scale_perm, xf_hi, xf_lo, x, norm_weight, eps, M, K, cs, q_hi, q_lo, q_token_stride,
                               k_arena, v_arena, arena_batch_stride, arena_head_stride, B, H, Hkv, D, q_len, past_len, cap,
                               past_len_dev, k_lo, v_lo, lo_batch_stride, lo_head_stride, -1, stream);
}

static inline int pc_gemm_qkv_rope_a8(const void* wf8_perm, const float* w_scale_perm, const void* xq_hi, const void* xq_lo,
                                      const float* x_scale, const float* corr, int64_t ldc, const int32_t* corr_has, int32_t M,
                                      int32_t K, const float* cs, void* q_hi, void* q_lo, int64_t q_token_stride, void* k_arena,
                                      void* v_arena, int64_t arena_batch_stride, int64_t arena_head_stride, int32_t B, int32_t H,
                                      int32_t Hkv, int32_t D, int32_t q_len, int32_t past_len, int32_t cap,
                                      const int32_t* past_len_dev, void* k_lo, void* v_lo, int64_t lo_batch_stride,
                                      int64_t lo_head_stride, int32_t lo_base, void* stream) {
    pc_gemm_args a = pc_compat_gemm_(PC_GEMM_EPI_QKV_ROPE, wf8_perm, w_scale_perm, xq_hi, xq_lo, NULL, NULL, 0.f, M, 0, K);
    a.cs = cs; a.q_hi = q_hi; a.q_lo = q_lo; a.q_token_stride = q_token_stride; a.k_arena = k_arena; a.v_arena = v_arena;
    a.arena_batch_stride = arena_batch_stride; a.arena_head_stride = arena_head_stride;
    a.B = B; a.H = H; a.Hkv = Hkv; a.D = D; a.q_len = q_len; a.past_len = past_len; a.cap = cap; a.past_len_dev = past_len_dev;
    a.k_lo = k_lo; a.v_lo = v_lo; a.lo_batch_stride = lo_batch_stride; a.lo_head_stride = lo_head_stride; a.lo_base = lo_base;
    a.x_scale = x_scale; a.corr = corr; a.ldc = ldc; a.corr_has = corr_has;
    return pc_gemm(&a, stream);
}

static inline int pc_gemm_qkv_rope_a8c(const void* wf8_perm, const float* w_scale_perm, const void* xq_hi, const void* xq_lo,
                                       const float* x_scale, const void* flags, const void* x_raw, const void* w_codes_t,
                                       int64_t ldt, const int32_t* row_perm, int32_t M, int32_t K, const float* cs, void* q_hi,
                                       void* q_lo, int64_t q_token_stride, void* k_arena, void* v_arena,
                                       int64_t arena_batch_stride, int64_t arena_head_stride, int32_t B, int32_t H, int32_t Hkv,
                                       int32_t D, int32_t q_len, int32_t past_len, int32_t cap, const int32_t* past_len_dev,
                                       void* k_lo, void* v_lo, int64_t lo_batch_stride, int64_t lo_head_stride, int32_t lo_base,
                                       void* stream) {
    pc_gemm_args a = pc_compat_gemm_(PC_GEMM_EPI_QKV_ROPE, wf8_perm, w_scale_perm, xq_hi, xq_lo, NULL, NULL, 0.f, M, 0, K);
    a.cs = cs; a.q_hi = q_hi; a.q_lo = q_lo; a.q_token_stride = q_token_stride; a.k_arena = k_arena; a.v_arena = v_arena;
    a.arena_batch_stride = arena_batch_stride; a.arena_head_stride = arena_head_stride;
    a.B = B; a.H = H; a.Hkv = Hkv; a.D = D; a.q_len = q_len; a.past_len = past_len; a.cap = cap; a.past_len_dev = past_len_dev;
    a.k_lo = k_lo; a.v_lo = v_lo; a.lo_batch_stride = lo_batch_stride; a.lo_head_stride = lo_head_stride; a.lo_base = lo_base;
    a.x_scale = x_scale; a.flags = flags; a.x_raw = x_raw; a.w_codes_t = w_codes_t; a.ldt = ldt; a.row_perm = row_perm;
    return pc_gemm(&a, stream);
}

#ifdef __cplusplus
}
#endif

#endif /* PROMPTCACHE_HIP_COMPAT_H */
