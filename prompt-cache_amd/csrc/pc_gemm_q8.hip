// pc_gemm_q8: C-ABI entry of the LLM.int8 projections with the activation quantiser inside the launch (kernel templates and the
// description of the three forms: pc_gemm_q8.h; include/promptcache_hip.h, pc_gemm_q8_args).  This unit instantiates the o_proj /
// down_proj launch shapes (residual add / store, F and C forms); q|k|v and gate|up live in pc_gemm_q8_norm.hip.
#include "pc_gemm_q8.h"

using namespace pcg;
using namespace pcq;

// ---------------------------------------------------------------------------------------------------
// pc_gemm_q8: LLM.int8 projection of <= 16 rows, quantiser inside (include/promptcache_hip.h, pc_gemm_q8_args).
PC_EXPORT int pc_gemm_q8(const pc_gemm_q8_args* a, void* stream) {
    PC_REQUIRE(a && a->struct_bytes == (uint32_t)sizeof(pc_gemm_q8_args), PC_ERR_ARG,
               "pc_gemm_q8: args is NULL or struct_bytes != sizeof(pc_gemm_q8_args) (ABI mismatch)");
    const int epi = a->epilogue;
    const bool qkv = epi == PC_GEMM_EPI_QKV_ROPE;
    const int M = a->M, K = a->K;
    const int N = qkv ? (a->H + 2 * a->Hkv) * a->D : a->N;
    PC_REQUIRE(epi == PC_GEMM_EPI_STORE || epi == PC_GEMM_EPI_ADD || epi == PC_GEMM_EPI_SILU || qkv, PC_ERR_ARG,
               "pc_gemm_q8: epilogue %d not in {store, add, SiLU, q|k|v}", epi);
    PC_REQUIRE(M > 0 && M <= 16 && N > 0 && N % 16 == 0 && K > 0 && K % 64 == 0, PC_ERR_ARG,
               "pc_gemm_q8: need 1 <= M <= 16, N %% 16 == 0, K %% 64 == 0");
    PC_REQUIRE(a->wf && a->w_scale && ((uintptr_t)a->w_scale & 15) == 0 && a->w_codes_t && a->ldt >= N && (!qkv || a->row_perm), PC_ERR_ARG,
               "pc_gemm_q8: needs the int8 weight image, 16-byte aligned w_scale, w_codes_t (ldt >= N) (and row_perm for q|k|v)");
    const bool norm = a->x != nullptr, fform = a->row_max != nullptr, part = a->part_o != nullptr, image = a->x_codes8 != nullptr;
    PC_REQUIRE(norm ? (a->norm_weight && !a->xf_hi && !fform && !part && !image) : ((a->xf_hi != nullptr) != part), PC_ERR_ARG,
               "pc_gemm_q8: pass exactly one of (x, norm_weight), the fp16 activation plane xf_hi, or pc_attn's partials part_o");
    PC_REQUIRE(!image || (a->xf_hi && a->x_scale && a->x_flags && !fform && !part && epi != PC_GEMM_EPI_STORE &&
                          (((uintptr_t)a->x_codes8 | (uintptr_t)a->x_flags) & 15) == 0), PC_ERR_ARG,
               "pc_gemm_q8: x_codes8 (a quantiser launch's operand image) goes with xf_hi (the fp16 plane), x_scale, x_flags (>= 16384 bytes), 16-byte aligned");
    PC_REQUIRE(!part || (a->part_ml && M == 1 && K <= 4096 && epi == PC_GEMM_EPI_ADD && !fform && a->part_nsplit >= 2 && a->part_nsplit <= pcm::kPartNS &&
                         a->part_head_dim > 0 && a->part_head_dim % 8 == 0 && K % a->part_head_dim == 0 &&
                         (((uintptr_t)a->part_o | (uintptr_t)a->part_ml) & 15) == 0), PC_ERR_ARG,
               "pc_gemm_q8: part_o (pc_attn defer_merge) needs part_ml, M = 1, K = H * part_head_dim <= 4096, 2..8 partials, residual-add epilogue");
    PC_REQUIRE(!a->flags_clear || (a->clear_bytes > 0 && a->clear_bytes % 16 == 0 && ((uintptr_t)a->flags_clear & 15) == 0), PC_ERR_ARG,
               "pc_gemm_q8: flags_clear needs a 16-byte aligned buffer and a multiple of 16 bytes");
    hipStream_t s = (hipStream_t)stream;
    Q8Params qp;
    memset(&qp, 0, sizeof(qp));
    GemmParams& p = qp.g;
    p.wf = (const _Float16*)a->wf; p.wscale = a->w_scale; p.w8 = 1;
    p.cbt = (const signed char*)a->w_codes_t; p.ldt = a->ldt; p.row_perm = a->row_perm;
    p.xn = a->x; p.gamma = (const _Float16*)a->norm_weight; p.eps = a->eps; p.xf_hi = (const _Float16*)a->xf_hi;
    p.M = M; p.ntiles = N / 16; p.KS = K / 32; p.kslices = 1;
    p.y = a->y; p.ldy = a->ldy; p.of_hi = (_Float16*)a->of_hi; p.of_lo = (_Float16*)a->of_lo;
    p.pmax_out = a->row_max_out; p.oflags_out = (unsigned char*)a->flags_out; p.thr_out = a->threshold;
    qp.threshold = a->threshold;
    qp.flags_clear = (unsigned char*)a->flags_clear; qp.clear_bytes = a->clear_bytes;
    qp.dbg_codes = (signed char*)a->dbg_codes; qp.dbg_scale = a->dbg_scale; qp.dbg_flags = (unsigned char*)a->dbg_flags;
    qp.img8 = (const signed char*)a->x_codes8; qp.img_scale = a->x_scale; qp.img_flags = (const unsigned char*)a->x_flags;
    qp.part.part_o = a->part_o; qp.part.part_ml = a->part_ml; qp.part.nsplit = a->part_nsplit; qp.part.D = a->part_head_dim; qp.part.q_len = 1;
    PC_REQUIRE(!a->dbg_codes || (a->dbg_scale && a->dbg_flags && !fform), PC_ERR_ARG, "pc_gemm_q8: dbg_codes goes with dbg_scale and dbg_flags (P form)");

    if (fform) {
        // ---- F form / C form: down_proj ----
        const int S = a->kslices < 1 ? 1 : a->kslices;
        const bool cform = S == 1 && M <= 4 && a->ks_tiles <= 2;
        PC_REQUIRE(epi == PC_GEMM_EPI_ADD && a->flags_in && a->row_max_units > 0 && K <= 16384 && (cform || (a->ks_counters && a->ks_scratch)),
                   PC_ERR_ARG, "pc_gemm_q8: row_max goes with the residual-add epilogue, flags_in, K <= 16384 (and ks_scratch / ks_counters "
                   "unless M <= 4 runs in one slice)");
        PC_REQUIRE(((uintptr_t)a->row_max & 15) == 0 && ((uintptr_t)a->flags_in & 15) == 0, PC_ERR_ARG, "pc_gemm_q8: row_max / flags_in must be 16-byte aligned");
        PC_REQUIRE(S <= kMaxSlices && a->y && a->ldy >= N && a->ldy % 4 == 0, PC_ERR_ARG, "pc_gemm_q8: kslices %d outside 1..8 or bad output", S);
        qp.pmax_in = a->row_max; qp.pmax_units = a->row_max_units; qp.flags_in = (const unsigned char*)a->flags_in;
        if (cform) {
            // one slice, the rows' codes staged once in LDS (decode)
            p.kslices = 1;
            if (K <= 11264) return a->ks_tiles == 2 ? launch_q8c<2, 11>(qp, s) : launch_q8c<1, 11>(qp, s);
            return a->ks_tiles == 2 ? launch_q8c<2, 16>(qp, s) : launch_q8c<1, 16>(qp, s);
        }
        PC_REQUIRE(a->ks_scratch_bytes >= pc_gemm_skinny_ks_scratch_bytes(N, S) && ((uintptr_t)a->ks_scratch & 15) == 0, PC_ERR_WORKSPACE,
                   "pc_gemm_q8: ks_scratch too small or misaligned");
        p.kslices = S;
        qp.slabs = (float*)a->ks_scratch; qp.counters = (uint32_t*)a->ks_counters; qp.formal = pc_formal_handoff();
        switch (a->ks_tiles) {
            case 2: return launch_q8f<2, 2>(qp, s);
            case 4: return launch_q8f<4, 2>(qp, s);
            case 8: return launch_q8f<8, 1>(qp, s);
            default: break;
        }
        pc_set_error("pc_gemm_q8: ks_tiles %d not in {2, 4, 8}", a->ks_tiles);
        return PC_ERR_ARG;
    }
    // ---- P form ----
    PC_REQUIRE(K <= 6144, PC_ERR_ARG, "pc_gemm_q8: the in-launch quantiser stages K <= 6144 features in LDS (K = %d)", K);
    if (qkv) {
        PC_REQUIRE(norm || image, PC_ERR_ARG, "pc_gemm_q8: q|k|v takes the fused-RMSNorm source or a quantiser launch's image");
        PC_REQUIRE(M == a->B * a->q_len && a->D % 16 == 0 && a->H > 0 && a->Hkv > 0, PC_ERR_ARG, "pc_gemm_q8: bad q|k|v shape");
        PC_REQUIRE(a->cs && a->q_hi && a->q_lo && a->k_arena && a->v_arena, PC_ERR_ARG, "pc_gemm_q8: null q|k|v pointer");
        PC_REQUIRE((int64_t)a->past_len + a->q_len <= a->cap, PC_ERR_BOUNDS,
                   "pc_gemm_q8: past_len %d + q_len %d exceeds arena rows %d", a->past_len, a->q_len, a->cap);
        PC_REQUIRE(a->q_token_stride % 4 == 0 && a->arena_head_stride % 4 == 0, PC_ERR_ARG, "pc_gemm_q8: strides must keep 8-byte alignment");
        PC_REQUIRE((a->k_lo == nullptr) == (a->v_lo == nullptr) && (!a->k_lo || a->lo_head_stride % 4 == 0), PC_ERR_ARG,
                   "pc_gemm_q8: k_lo / v_lo go together, strides must keep 8-byte alignment");
        PC_REQUIRE(a->lo_base >= -2 && (a->lo_base != -2 || a->past_len_dev) && (a->lo_base < 0 || a->lo_base <= a->past_len), PC_ERR_ARG,
                   "pc_gemm_q8: lo_base must be -1, -2 (past_len_dev[1]) or lie in [0, past_len]");
        p.rope.cs = (const float2*)a->cs; p.rope.q_hi = (_Float16*)a->q_hi; p.rope.q_lo = (_Float16*)a->q_lo; p.rope.q_ts = a->q_token_stride;
        p.rope.k_arena = (_Float16*)a->k_arena; p.rope.v_arena = (_Float16*)a->v_arena; p.rope.a_bs = a->arena_batch_stride;
        p.rope.a_hs = a->arena_head_stride; p.rope.past_len_dev = a->past_len_dev;
        p.rope.k_lo = (_Float16*)a->k_lo; p.rope.v_lo = (_Float16*)a->v_lo; p.rope.lo_bs = a->lo_batch_stride; p.rope.lo_hs = a->lo_head_stride;
        p.rope.lo_base = a->lo_base;
        p.rope.H = a->H; p.rope.Hkv = a->Hkv; p.rope.D = a->D; p.rope.q_len = a->q_len; p.rope.past_len = a->past_len;
        return launch_q8p_rope(qp, choose_T(p.ntiles), p.ntiles, K, s);
    }
    if (epi == PC_GEMM_EPI_SILU) {
        PC_REQUIRE((norm || image) && N % 64 == 0 && a->of_hi, PC_ERR_ARG, "pc_gemm_q8: the SiLU epilogue takes the fused-RMSNorm source, N = 2*inter (inter %% 32 == 0), of_hi");
        PC_REQUIRE((a->row_max_out == nullptr) == (a->flags_out == nullptr), PC_ERR_ARG, "pc_gemm_q8: row_max_out and flags_out go together");
        p.npairs = N / 32; p.KSo = (N / 2) / 32;
        return launch_q8p_silu(qp, choose_T(p.npairs), p.npairs, K, s);
    }
    PC_REQUIRE(a->y && a->ldy >= N && a->ldy % 4 == 0, PC_ERR_ARG, "pc_gemm_q8: bad output");
    PC_REQUIRE(!norm, PC_ERR_ARG, "pc_gemm_q8: the store / residual-add epilogues take the fp16 plane xf_hi");
    const int T = choose_T(p.ntiles);
    if (epi == PC_GEMM_EPI_ADD) return launch_q8p<EPI_ADD, false, 2>(qp, T, p.ntiles, K, s);
    return launch_q8p<EPI_STORE, false, 2>(qp, T, p.ntiles, K, s);
}
