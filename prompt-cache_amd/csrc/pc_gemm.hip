// C-ABI entry points of the weight-streaming projections (M = B*q_len <= 512 rows): pc_gemm_skinny*, pc_gemm_qkv_rope*,
// pc_rmsnorm_frag / pc_layernorm_frag.  The kernel templates live in pc_gemm_skinny.h; their launch shapes are instantiated per
// row regime in pc_gemm_mt1.hip / pc_gemm_mt2.hip / pc_gemm_mt4.hip (<= 16 / <= 32 / <= 64 rows) and pc_gemm_rows.hip
// (65..512 rows), the persistent o_proj -> gate|up -> down_proj (-> q|k|v) launch in pc_gemm_chain.hip.
//
// Replaces  q_proj/k_proj/v_proj   promptcache/model/llama2.py:345-347   (one fused [q|k|v] GEMM)
//           o_proj + residual       promptcache/model/llama2.py:405, :638
//           gate/up + SiLU*up       promptcache/model/llama2.py:242
//           down_proj + residual    promptcache/model/llama2.py:242, :644
//           lm_head                 promptcache/model/llama2.py:1050
//           LlamaRMSNorm (producer) promptcache/model/llama2.py:103-108   (pc_rmsnorm_frag)
#include "pc_gemm_skinny.h"

namespace pcg {

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

}  // namespace pcg

using namespace pcg;

// dev hook: the next weight-streaming launches of this thread stamp per-wave wall-clock times into `buf` (NULL: off)
static thread_local unsigned long long* g_gemm_trace = nullptr;
PC_EXPORT int pc_dev_gemm_trace(void* buf) { g_gemm_trace = (unsigned long long*)buf; return PC_OK; }

namespace {

// intra-workgroup K skew / priority alternation of the <= 64-row launches (GemmParams::kskew, prio_alt); dev overrides
void set_k_balance(GemmParams& p, int K) {
    static const int skew = [] { const char* e = getenv("PC_GEMM_KSKEW"); return e ? atoi(e) : 0; }();
    static const int alt = [] { const char* e = getenv("PC_GEMM_PRIO_ALT"); return e ? atoi(e) : 0; }();
    p.prio_alt = alt;
    p.kskew = skew;
    // the fused-RMSNorm source stages a wave's gain slice in LDS: at most kGamSteps k-steps per wave
    const int n = pc_ceil_div(K / 32, p.kslices > 0 ? p.kslices : 1);
    if (p.xn && (n * (64 + p.kskew) + 8 * 64 - 1) / (8 * 64) > kGamSteps) p.kskew = 0;
    if (p.M > 64) p.kskew = 0;                               // (the row-split kernel has no K split inside a workgroup)
}

int launch_MT(int epi, const GemmParams& p, int T, int units, hipStream_t s) {
    const int mt = pc_ceil_div(p.M, 16);
    if (mt > 4) return launch_rows_epi(epi, p, units, s);
    if (mt <= 1) return launch_skinny_mt1(epi, p, T, units, s);
    if (mt == 2) return launch_skinny_mt2(epi, p, T, units, s);
    return launch_skinny_mt4(epi, p, T, units, s);
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
    PC_REQUIRE(!xn || (gamma && M <= 16 && kslices == 1 && (epilogue == EPI_STORE || epilogue == EPI_SILU) && K <= 32 * kGamSteps * kWaves),
               PC_ERR_ARG, "pc_gemm_skinny_norm: the fused-RMSNorm source needs M <= 16, K <= 16384, no K-slicing, epilogue 0 or 2");
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
    p.trace = g_gemm_trace;
    set_k_balance(p, K);
    hipStream_t s = (hipStream_t)stream;
    if (epilogue == EPI_SILU) {
        PC_REQUIRE(N % 64 == 0, PC_ERR_ARG, "pc_gemm_skinny: SiLU epilogue needs N = 2*inter with inter%%32==0");
        PC_REQUIRE(of_hi && of_lo, PC_ERR_ARG, "pc_gemm_skinny: SiLU epilogue needs output planes");
        p.npairs = N / 32;          // inter / 16
        p.KSo = (N / 2) / 32;       // k-steps of the consumer (down_proj, K = inter)
        return launch_MT(EPI_SILU, p, choose_T(p.npairs), p.npairs, s);
    }
    if (epilogue == EPI_GELU) {
        PC_REQUIRE(N % 32 == 0 && of_hi && of_lo, PC_ERR_ARG, "pc_gemm_skinny: GELU epilogue needs N%%32==0 and output planes");
        p.KSo = N / 32;             // k-steps of the consumer (dense_4h_to_h, K = N)
        return launch_MT(EPI_GELU, p, choose_T(p.ntiles), p.ntiles, s);
    }
    PC_REQUIRE(y && ldy >= N && ldy % 4 == 0, PC_ERR_ARG, "pc_gemm_skinny: bad output");
    if (epilogue == EPI_ADD) return launch_MT(EPI_ADD, p, choose_T(p.ntiles), p.ntiles, s);
    PC_REQUIRE(epilogue == EPI_STORE, PC_ERR_ARG, "pc_gemm_skinny: unknown epilogue %d", epilogue);
    return launch_MT(EPI_STORE, p, choose_T(p.ntiles * kslices) , p.ntiles, s);
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
    PC_REQUIRE(!xn || K <= 32 * kGamSteps * kWaves, PC_ERR_ARG, "pc_gemm_qkv_rope_norm: the fused-RMSNorm source needs K <= 16384");
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
    p.trace = g_gemm_trace;
    set_k_balance(p, K);
    return launch_MT(EPI_ROPE, p, choose_T(p.ntiles), p.ntiles, (hipStream_t)stream);
}
}  // namespace


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
