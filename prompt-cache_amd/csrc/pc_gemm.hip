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
        constexpr int CH = G <= 4 ? 4 : 1;     // (4 slabs -- the 65..512-row stack at every hidden size up to 8192 -- in ONE round trip; 6 slabs at hidden 5120: two)
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

// q|k|v with K split across workgroups (65..288 rows, wide panels): the projection leaves `nslabs` fp32 slabs [M][N] of partial
// sums in the row-PERMUTED column order of the q|k|v weight image (inside every head, tile j = features 8j..8j+7 then their rotary
// partners D/2+8j..); this kernel adds the slabs in slice order and does what tile_epilogue<EPI_ROPE> does on a tile: RoPE at the
// row's table entries for q and k heads, hi / lo split, q into the planes, k and v (+ residuals) into the arena behind the past.
// One thread = one (row, head, tile j, half-tile quad): 4 low-half values and their 4 partners.
__global__ __launch_bounds__(256) void qkv_rope_slabs_kernel(const float* __restrict__ slabs, int nslabs, int64_t slab_stride,
                                                             int M, int N, const RopeEpi e) {
    const int tpd = e.D >> 4;                               // tiles per head
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int per_row = (N >> 4) * 2;                       // (tile, quad) pairs per row
    if (idx >= (int64_t)M * per_row) return;
    const int row = (int)(idx / per_row), rem = (int)(idx - (int64_t)row * per_row);
    const int unit = rem >> 1, q4 = rem & 1;                // tile of the permuted image, which 4 of its 8 frequency slots
    const int hh = unit / tpd, j = unit - hh * tpd;
    const float* sp = slabs + (int64_t)row * N + unit * 16 + q4 * 4;
    f4 lo4 = *(const f4*)sp, hi4 = *(const f4*)(sp + 8);
    for (int z = 1; z < nslabs; ++z) {
        const f4 a = *(const f4*)(sp + z * slab_stride), b = *(const f4*)(sp + z * slab_stride + 8);
        lo4[0] += a[0]; lo4[1] += a[1]; lo4[2] += a[2]; lo4[3] += a[3];
        hi4[0] += b[0]; hi4[1] += b[1]; hi4[2] += b[2]; hi4[3] += b[3];
    }
    const int i0 = 8 * j + 4 * q4;                          // rotary frequency index of element 0
    const int bb = row / e.q_len, tt = row - bb * e.q_len;
    const bool rot = hh < e.H + e.Hkv;
    h4 ah, al, ch, cl;
    const float2* cs = e.cs + (int64_t)row * (e.D >> 1) + i0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float a = lo4[r], c = hi4[r];
        if (rot) {                                          // q*cos + rotate_half(q)*sin (llama2.py:208)
            const float2 w = cs[r];
            a = lo4[r] * w.x - hi4[r] * w.y;
            c = hi4[r] * w.x + lo4[r] * w.y;
        }
        _Float16 t0, t1, t2, t3;
        pc_split(a, t0, t1);
        pc_split(c, t2, t3);
        ah[r] = t0; al[r] = t1; ch[r] = t2; cl[r] = t3;
    }
    const int half = e.D >> 1;
    if (hh < e.H) {
        const int64_t off = (int64_t)row * e.q_ts + (int64_t)hh * e.D + i0;
        *(h4*)(e.q_hi + off) = ah; *(h4*)(e.q_hi + off + half) = ch;
        *(h4*)(e.q_lo + off) = al; *(h4*)(e.q_lo + off + half) = cl;
        return;
    }
    const int past = e.past_len_dev ? *e.past_len_dev : e.past_len;
    const bool is_k = hh < e.H + e.Hkv;
    const int kh = is_k ? hh - e.H : hh - e.H - e.Hkv;
    _Float16* dst = (is_k ? e.k_arena : e.v_arena) + bb * e.a_bs + (int64_t)kh * e.a_hs + (int64_t)(past + tt) * e.D + i0;
    *(h4*)dst = ah; *(h4*)(dst + half) = ch;
    if (e.k_lo) {
        const int lr = e.lo_base == -1 ? tt : past + tt - (e.lo_base == -2 ? e.past_len_dev[1] : e.lo_base);
        _Float16* dl = (is_k ? e.k_lo : e.v_lo) + bb * e.lo_bs + (int64_t)kh * e.lo_hs + (int64_t)lr * e.D + i0;
        *(h4*)dl = al; *(h4*)(dl + half) = cl;
    }
}

}  // namespace


// ---------------------------------------------------------------------------------------------------
// pc_gemm: THE entry point of the weight-streaming projections (M = B*q_len <= 512 rows).  One struct (include/promptcache_hip.h,
// pc_gemm_args) covers what round 1-2 exported as thirteen functions: plain / residual-add / SiLU*up / GELU / q|k|v + RoPE + append
// epilogues, activation planes or the fused-RMSNorm source, fp16 or int8 weight images, LLM.int8 code planes with a separate or
// an in-launch outlier correction, K slices as slabs or reduced inside the launch.
PC_EXPORT int pc_gemm(const pc_gemm_args* a, void* stream) {
    PC_REQUIRE(a && a->struct_bytes == (uint32_t)sizeof(pc_gemm_args), PC_ERR_ARG,
               "pc_gemm: args is NULL or struct_bytes != sizeof(pc_gemm_args) (ABI mismatch)");
    const int epi = a->epilogue;
    const bool qkv = epi == PC_GEMM_EPI_QKV_ROPE;
    const int M = a->M, K = a->K;
    const int N = qkv ? (a->H + 2 * a->Hkv) * a->D : a->N;
    PC_REQUIRE(epi >= 0 && epi <= PC_GEMM_EPI_GELU, PC_ERR_ARG, "pc_gemm: unknown epilogue %d", epi);
    PC_REQUIRE(M > 0 && M <= kRowsMaxM, PC_ERR_ARG, "pc_gemm: M=%d outside 1..512 (use pc_gemm_dense above)", M);
    PC_REQUIRE(N > 0 && N % 16 == 0 && K > 0 && K % 32 == 0, PC_ERR_ARG, "pc_gemm: need N%%16==0 and K%%32==0");
    const bool norm = a->x != nullptr;
    PC_REQUIRE(a->wf && (norm ? (a->norm_weight && !a->xf_hi) : (a->xf_hi != nullptr)), PC_ERR_ARG,
               "pc_gemm: pass the weight image and either the activation planes or (x, norm_weight)");
    PC_REQUIRE(!norm || (M <= (a->w_scale ? 16 : 32) && K <= 32 * kGamSteps * kWaves &&
                         (epi == PC_GEMM_EPI_STORE || epi == PC_GEMM_EPI_SILU || qkv) && a->kslices <= 1), PC_ERR_ARG,
               "pc_gemm: the fused-RMSNorm source needs M <= 32 (16 with int8 weights), K <= 16384, no K-slicing, epilogue store / SiLU / q|k|v");
    const int kslices = a->kslices < 1 ? 1 : a->kslices;
    const bool fused = a->flags != nullptr;                       // in-launch LLM.int8 outlier correction
    PC_REQUIRE(!a->w_scale || (M <= 64 && (norm || a->xf_lo) && ((uintptr_t)a->w_scale & 15) == 0 && K % 64 == 0), PC_ERR_ARG,
               "pc_gemm: int8 weights need M <= 64, K %% 64 == 0, split-precision activations and 16-byte aligned scales");
    PC_REQUIRE(!a->x_scale || fused || (a->w_scale && a->corr && a->corr_has && a->ldc >= N && a->ldc % 4 == 0 &&
                                        ((uintptr_t)a->corr & 15) == 0 && kslices == 1), PC_ERR_ARG,
               "pc_gemm: int8 activations need int8 weights, x_scale, corr (16-byte aligned, ldc >= N), corr_has, no K-slices");
    PC_REQUIRE(!fused || (a->x_scale && a->w_scale && kslices == 1 && a->x_raw && a->w_codes_t && a->ldt >= N && K <= 16384 &&
                          ((uintptr_t)a->flags & 15) == 0 && (!qkv || a->row_perm)), PC_ERR_ARG,
               "pc_gemm: the fused correction needs x_scale, w_scale, 16-byte aligned flags (>= 16384 bytes), x_raw, w_codes_t "
               "(ldt >= N), K <= 16384 (and row_perm for q|k|v)");
    PC_REQUIRE(!a->x_codes8 || (a->x_scale && M <= 64 && ((uintptr_t)a->x_codes8 & 15) == 0), PC_ERR_ARG,
               "pc_gemm: x_codes8 goes with x_scale (LLM.int8 activations), M <= 64, 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;

    // ---- residual add with K split across workgroups and the reduction inside the launch (pc_gemm_ks.hip) ----
    if (epi == PC_GEMM_EPI_ADD && a->ks_counters) {
        PC_REQUIRE(!a->w_scale && !norm && a->xf_lo && M <= 32, PC_ERR_ARG,
                   "pc_gemm: the in-launch K reduction takes fp16 weights, both activation planes, M <= 32");
        return launch_skinny_ks(a->wf, a->xf_hi, a->xf_lo, M, N, K, a->y, a->ldy, kslices, a->ks_tiles, a->ks_scratch,
                                a->ks_scratch_bytes, a->ks_counters, a->rows_dev, s);
    }

    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.wf = (const _Float16*)a->wf; p.xf_hi = (const _Float16*)a->xf_hi; p.xf_lo = (const _Float16*)a->xf_lo;
    p.xn = a->x; p.gamma = (const _Float16*)a->norm_weight; p.eps = a->eps;
    p.wscale = a->w_scale; p.w8 = a->w_scale ? 1 : 0;
    p.xscale = a->x_scale; p.corr = a->corr; p.ldc = a->ldc; p.corr_has = a->corr_has;
    p.xq8 = (const signed char*)a->x_codes8;
    if (fused) {
        p.oflags = (const unsigned char*)a->flags; p.xraw = (const _Float16*)a->x_raw; p.cbt = (const signed char*)a->w_codes_t;
        p.ldt = a->ldt; p.row_perm = a->row_perm;
    }
    p.M = M; p.m_dev = a->rows_dev; p.ntiles = N / 16; p.KS = K / 32;
    p.trace = g_gemm_trace;

    if (qkv) {
        PC_REQUIRE(M == a->B * a->q_len, PC_ERR_ARG, "pc_gemm: q|k|v needs M=%d == B*q_len", M);
        PC_REQUIRE(a->D % 16 == 0 && a->H > 0 && a->Hkv > 0, PC_ERR_ARG, "pc_gemm: bad q|k|v shape");
        PC_REQUIRE(a->cs && a->q_hi && a->q_lo && a->k_arena && a->v_arena, PC_ERR_ARG, "pc_gemm: null q|k|v pointer");
        PC_REQUIRE((int64_t)a->past_len + a->q_len <= a->cap, PC_ERR_BOUNDS,
                   "pc_gemm: past_len %d + q_len %d exceeds arena rows %d", a->past_len, a->q_len, a->cap);
        PC_REQUIRE(a->q_token_stride % 4 == 0 && a->arena_head_stride % 4 == 0, PC_ERR_ARG, "pc_gemm: strides must keep 8-byte alignment");
        PC_REQUIRE((a->k_lo == nullptr) == (a->v_lo == nullptr) && (!a->k_lo || a->lo_head_stride % 4 == 0), PC_ERR_ARG,
                   "pc_gemm: k_lo / v_lo go together, strides must keep 8-byte alignment");
        PC_REQUIRE(a->lo_base >= -2 && (a->lo_base != -2 || a->past_len_dev) && (a->lo_base < 0 || a->lo_base <= a->past_len), PC_ERR_ARG,
                   "pc_gemm: lo_base must be -1 (pass-relative rows), -2 (past_len_dev[1]) or lie in [0, past_len]");
        p.kslices = 1;
        p.rope.cs = (const float2*)a->cs; p.rope.q_hi = (_Float16*)a->q_hi; p.rope.q_lo = (_Float16*)a->q_lo; p.rope.q_ts = a->q_token_stride;
        p.rope.k_arena = (_Float16*)a->k_arena; p.rope.v_arena = (_Float16*)a->v_arena; p.rope.a_bs = a->arena_batch_stride;
        p.rope.a_hs = a->arena_head_stride; p.rope.past_len_dev = a->past_len_dev;
        p.rope.k_lo = (_Float16*)a->k_lo; p.rope.v_lo = (_Float16*)a->v_lo; p.rope.lo_bs = a->lo_batch_stride; p.rope.lo_hs = a->lo_head_stride;
        p.rope.lo_base = a->lo_base;
        p.rope.H = a->H; p.rope.Hkv = a->Hkv; p.rope.D = a->D; p.rope.q_len = a->q_len; p.rope.past_len = a->past_len;
        if (kslices > 1) {
            // K slices under the q|k|v epilogue (row-split kernel, wide panels): partial slabs in ks_scratch, then the rotation /
            // append as a second launch over the slabs (qkv_rope_slabs_kernel)
            PC_REQUIRE(M > 64 && a->xf_lo && !a->w_scale && !norm && kslices <= 8, PC_ERR_ARG,
                       "pc_gemm: K slices under the q|k|v epilogue are for 65..512 rows, fp16 weights, both activation planes");
            PC_REQUIRE(a->ks_scratch && ((uintptr_t)a->ks_scratch & 15) == 0 &&
                       a->ks_scratch_bytes >= (int64_t)kslices * M * N * (int64_t)sizeof(float), PC_ERR_WORKSPACE,
                       "pc_gemm: q|k|v with %d K slices needs ks_scratch >= %lld bytes (16-byte aligned)", kslices,
                       (long long)kslices * M * N * (long long)sizeof(float));
            p.y = (float*)a->ks_scratch; p.ldy = N;
            p.kslices = kslices; p.slab_stride = (int64_t)M * N;
            const RopeEpi e = p.rope;
            int rc = launch_MT(EPI_STORE, p, choose_T(p.ntiles * kslices), p.ntiles, s);
            if (rc != PC_OK) return rc;
            const int64_t items = (int64_t)M * (N / 16) * 2;
            hipLaunchKernelGGL(qkv_rope_slabs_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s,
                               (const float*)a->ks_scratch, kslices, (int64_t)M * N, M, N, e);
            return pc_check_launch("qkv_rope_slabs_kernel");
        }
        set_k_balance(p, K);
        return launch_MT(EPI_ROPE, p, choose_T(p.ntiles), p.ntiles, s);
    }

    p.y = a->y; p.ldy = a->ldy; p.of_hi = (_Float16*)a->of_hi; p.of_lo = (_Float16*)a->of_lo;
    PC_REQUIRE(kslices <= 16 && (kslices == 1 || epi == PC_GEMM_EPI_STORE), PC_ERR_ARG,
               "pc_gemm: K slices as slabs (kslices=%d) go with the plain-store epilogue (residual add: pass ks_counters)", kslices);
    p.kslices = kslices; p.slab_stride = (int64_t)M * a->ldy;
    set_k_balance(p, K);
    if (epi == PC_GEMM_EPI_SILU) {
        PC_REQUIRE(N % 64 == 0 && a->of_hi && a->of_lo, PC_ERR_ARG, "pc_gemm: the SiLU epilogue needs N = 2*inter with inter%%32==0 and output planes");
        p.npairs = N / 32;          // inter / 16
        p.KSo = (N / 2) / 32;       // k-steps of the consumer (down_proj, K = inter)
        PC_REQUIRE((a->row_max_out == nullptr) == (a->flags_out == nullptr) && (!a->row_max_out || M <= 16), PC_ERR_ARG,
                   "pc_gemm: row_max_out and flags_out go together (M <= 16)");
        p.pmax_out = a->row_max_out; p.oflags_out = (unsigned char*)a->flags_out; p.thr_out = a->out_threshold;
        return launch_MT(EPI_SILU, p, choose_T(p.npairs), p.npairs, s);
    }
    if (epi == PC_GEMM_EPI_GELU) {
        PC_REQUIRE(N % 32 == 0 && a->of_hi && a->of_lo, PC_ERR_ARG, "pc_gemm: the GELU epilogue needs N%%32==0 and output planes");
        p.KSo = N / 32;             // k-steps of the consumer (dense_4h_to_h, K = N)
        return launch_MT(EPI_GELU, p, choose_T(p.ntiles), p.ntiles, s);
    }
    PC_REQUIRE(a->y && a->ldy >= N && a->ldy % 4 == 0, PC_ERR_ARG, "pc_gemm: bad output");
    if (epi == PC_GEMM_EPI_ADD) return launch_MT(EPI_ADD, p, choose_T(p.ntiles), p.ntiles, s);
    return launch_MT(EPI_STORE, p, choose_T(p.ntiles * kslices), p.ntiles, s);
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

