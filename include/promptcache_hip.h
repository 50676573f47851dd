/*
 * promptcache_hip.h -- C-ABI of the MI355X-native prompt-cache prefill path (libpromptcache_hip.so).
 *
 * The reference (yale-sys/prompt-cache) is pure Python with no FFI; the seam this library sits
 * under is the `LanguageModel` adapter (promptcache/model/__init__.py:90-161) as consumed by
 * `CacheEngine` (promptcache/cache_engine.py:331-522) and `GenerationEngine`
 * (promptcache/generation_engine.py:61-209).  Each entry point below names the reference
 * op sequence (file:line) it replaces.  See INTEGRATION.md for the ctypes binding.
 *
 * Conventions
 *   - every data pointer is a DEVICE pointer owned by the caller (PyTorch-ROCm allocations);
 *     tables marked "host" are small host arrays read during the call;
 *   - `stream` is a hipStream_t passed as void*; nothing allocates, nothing synchronises, all
 *     launches are asynchronous on `stream` (hipGraph-capturable);
 *   - KV element type is IEEE fp16 (the reference stages KV as torch.half, cache_engine.py:105-106);
 *   - return 0 on success, a negative code on error; pc_last_error_string() describes the last
 *     error on the calling thread.
 */
#ifndef PROMPTCACHE_HIP_H
#define PROMPTCACHE_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PC_OK 0
#define PC_ERR_ARG (-1001)      /* bad argument (null pointer, unsupported head_dim, ...) */
#define PC_ERR_BOUNDS (-1002)   /* a segment / append would run past the destination capacity */
#define PC_ERR_WORKSPACE (-1003)/* workspace too small (query pc_attn_workspace_bytes) */
#define PC_ERR_HIP(e) (-(int)(e)) /* -hipError_t from a launch */

#define PC_ABI_VERSION 1

int pc_version(void);
const char* pc_last_error_string(void);

/* ---------------------------------------------------------------------------------------------
 * pc_kv_gather -- replaces PromptCache.update's copy loop, cache_engine.py:135-151
 *   (`tgt[:, st:ed, :].copy_(src)` x 2 x n_layers per segment) with ONE launch over a segment table.
 *
 *   seg_src[s]     device ptr to segment s's module KV, contiguous [n_layers][2][n_kv_heads][seg_len[s]][head_dim]
 *                  (index 0 of the "2" axis = K, 1 = V)                                    (host array)
 *   seg_len[s]     tokens in segment s                                                      (host array)
 *   seg_dst_off[s] first token row of segment s in the staged buffer                        (host array)
 *   dst            staged buffer [n_layers][2][n_kv_heads][max_ctx][head_dim] fp16 (cache_engine.py:104-107,
 *                  all layers in one allocation; layer i's K plane is the reference's device_cache[i][0])
 *   Segments with seg_dst_off + seg_len > max_ctx -> PC_ERR_BOUNDS (nothing is launched).
 * ------------------------------------------------------------------------------------------- */
int pc_kv_gather(const void* const* seg_src, const int32_t* seg_len, const int32_t* seg_dst_off,
                 int32_t nseg, void* dst, int32_t n_layers, int32_t n_kv_heads, int32_t head_dim,
                 int32_t max_ctx, void* stream);

/* ---------------------------------------------------------------------------------------------
 * pc_kv_row_table -- the same staging plan as pc_kv_gather, expanded to ONE ENTRY PER STAGED ROW, for the attention
 *   that stages while it reads (pc_attn `gather_rows`): the first forward over a freshly assembled prompt then reads every
 *   key / value row from its module store and writes the staged row as it goes, so PromptCache.update's copy
 *   (cache_engine.py:135-151) and the attention's first read of the staged rows (llama2.py:361-388) move the K/V across the
 *   chip once -- read 1x + write 1x -- instead of read + write + read.
 *
 *   segs      DEVICE array of segment descriptors in staging order (dst_row ascending, rows of a segment contiguous):
 *             src = the segment's module KV [n_layers][2][n_kv_heads][len][head_dim] (as pc_kv_gather's seg_src)
 *   nseg_dev  DEVICE int32: number of descriptors (<= max_seg) -- both come out of the caller's packed per-call input
 *             block, so a captured hipGraph replays the expansion for every new prompt
 *   dst, max_ctx   the staged buffer of pc_kv_gather.  Rows [0, total_rows_dev[0]) that no segment covers (rows a previous
 *             prompt staged at the same place and this one keeps; rows the model itself appended) get an entry that points
 *             at the staged buffer itself with PC_KV_ROW_STAGED set: read in place, not written
 *   rows      out: max_ctx entries; entry r describes staged row r: `base` = address of its head_dim fp16 values in plane 0
 *             (layer 0, K, head 0) of its source, `plane_stride16` = distance between consecutive planes of that source in
 *             16-byte units (plane p = (layer * 2 + k|v) * n_kv_heads + head)
 * ------------------------------------------------------------------------------------------- */
typedef struct pc_kv_seg { const void* src; int32_t dst_row; int32_t len; } pc_kv_seg;
typedef struct pc_kv_row { uint64_t base; uint32_t plane_stride16; uint32_t flags; } pc_kv_row;
#define PC_KV_ROW_STAGED 1u   /* the row already lies in the staged buffer: the attention does not write it */
int pc_kv_row_table(const pc_kv_seg* segs, const int32_t* nseg_dev, int32_t max_seg, const int32_t* total_rows_dev,
                    const void* dst, int32_t n_kv_heads, int32_t head_dim, int32_t max_ctx, pc_kv_row* rows, void* stream);

/* ---------------------------------------------------------------------------------------------
 * pc_kv_slice_store -- replaces SchemaCache._process's slice-and-store, cache_engine.py:283-296
 *   (`k_cache[j, :, st:ed, :]` per layer per TokenSequence, then `.cpu()`): the module KV stays in HBM.
 *
 *   src            the encode pass's KV arena [n_layers][2][n_kv_heads][src_cap][head_dim] (one batch row)
 *   seg_src_off[s] first token row of segment s inside the arena (= position_ids.index(offset), :278)
 *   seg_dst[s]     device ptr to the segment store, contiguous [n_layers][2][n_kv_heads][seg_len[s]][head_dim]
 * ------------------------------------------------------------------------------------------- */
int pc_kv_slice_store(const void* src, int32_t src_cap, const int32_t* seg_src_off, const int32_t* seg_len,
                      void* const* seg_dst, int32_t nseg, int32_t n_layers, int32_t n_kv_heads,
                      int32_t head_dim, void* stream);

/* ---------------------------------------------------------------------------------------------
 * pc_rope_table -- replaces LlamaRotaryEmbedding's cos/sin table + the `cos[position_ids]` gather,
 *   llama2.py:129-147 and :204-207.  Only the rows the supplied position ids select are produced
 *   (fp32, angle = float(pos) * inv_freq, as the reference builds them before its dtype cast).
 *
 *   pos       int32 [n_tok]        position ids of the new tokens (arbitrary, gaps allowed)
 *   inv_freq  fp32  [head_dim/2]   theta^(-2i/D), computed by the caller exactly as llama2.py:121
 *   cs        fp32  [n_tok][head_dim/2][2]  (cos, sin) out
 * ------------------------------------------------------------------------------------------- */
int pc_rope_table(const int32_t* pos, const float* inv_freq, float* cs, int32_t n_tok, int32_t head_dim,
                  void* stream);

/* ---------------------------------------------------------------------------------------------
 * pc_rope_append -- replaces apply_rotary_pos_emb (llama2.py:202-210) on q and k plus the
 *   `torch.cat([past, new], dim=2)` of llama2.py:361-364: rotated q goes to q_out (fp16), rotated k and
 *   v are written IN PLACE at token rows [past_len, past_len+q_len) of this layer's KV arena (the past
 *   is not re-copied).
 *
 *   q      [B][q_len][H][D] projection output, fp32 when in_is_f32 else fp16; token stride
 *          q_token_stride, batch stride q_batch_stride (elements of the input type)
 *   q_out  fp16 [B][q_len][H][D], strides qo_* (may alias q when the input is fp16: in-place rotation)
 *   q_out_lo  optional fp16 plane, same strides: fp16(q_rot - fp16(q_rot)), the low-order half of the
 *          split-precision q that pc_attn consumes in its small-q (HBM-bound) instantiation
 *   k_new  [B][q_len][Hkv][D] (pre-RoPE, same type as q), v_new likewise; strides kv_new_*
 *   k_arena, v_arena  fp16 [B][Hkv][cap][D]: head stride arena_head_stride, batch stride arena_batch_stride
 *   cs     from pc_rope_table, [B*q_len][D/2][2]
 *   past_len_dev: optional device int32*; when non-null the kernel reads past_len from it (graph replay)
 * ------------------------------------------------------------------------------------------- */
int pc_rope_append(const void* q, int64_t q_batch_stride, int64_t q_token_stride,
                   void* q_out, void* q_out_lo, int64_t qo_batch_stride, int64_t qo_token_stride,
                   const void* k_new, const void* v_new, int64_t kv_new_batch_stride, int64_t kv_new_token_stride,
                   void* k_arena, void* v_arena, int64_t arena_batch_stride, int64_t arena_head_stride,
                   const float* cs, int32_t B, int32_t H, int32_t Hkv, int32_t D, int32_t q_len,
                   int32_t past_len, int32_t cap, int32_t in_is_f32, const int32_t* past_len_dev, void* stream);

/* ---------------------------------------------------------------------------------------------
 * pc_attn -- replaces llama2.py:368-398: repeat_kv, QK^T/sqrt(D), + mask, softmax(fp32), PV,
 *   transpose/reshape.  The mask of llama2.py:62-76 / :798-819 is implicit: new token i (input order)
 *   sees every staged key j < past_len and new keys past_len + i' with i' <= i.  fp32 softmax and
 *   accumulation, MFMA fp16 x fp16 -> fp32 for both contractions, flash-style (scores are never
 *   materialised), split over the KV axis when (heads x q-blocks) cannot fill the chip.
 *   ONE struct-taking entry point (rounds 1-2 exported pc_attn_fwd / _alibi / _ex / _var; they had no caller left and are
 *   gone since round 4).  Optional pointers are NULL when unused.
 *
 *   struct_bytes  sizeof(pc_attn_args) of the caller's header (ABI check)
 *   q, q_lo    fp16 [B][q_len][H][D], RoPE applied; q_lo: optional low-order plane (q = q_hi + q_lo: split precision)
 *   k, v       fp16 arena planes [B][Hkv][cap][D] holding past_len + q_len valid rows
 *   out, out_lo   fp16 [B][q_len][H*D] (token stride out_token_stride); out_lo: optional residual plane of out
 *   out_frag_hi / out_frag_lo   optional (both or neither; B*q_len <= 512): instead of `out`, the result as
 *              split-precision fragment planes [ceil(B*q_len/16)][H*D/32][64][8] consumed by pc_gemm (o_proj)
 *   workspace  >= pc_attn_workspace_bytes(...) bytes of device memory (split-KV partials)
 *   past_len_dev   optional device int32* (graph replay; `past_len` is then the upper bound that sizes the launch)
 *   past_lens  optional device int32[B]: ONE PAST LENGTH PER BATCH ROW (ragged-prefix batches of the schema encode: scaffold
 *              suffixes of different unions in one batch, each over its own trunk prefix -- the batched form of
 *              cache_engine.py:217-304's per-scaffold forwards); past_len = their maximum.  Batch row b attends to keys
 *              [0, past_lens[b]) plus the rows it appended, up to its own.  Excludes past_len_dev, ALiBi, fragment output;
 *              residual planes must be arena-shaped (lo_row0 = 0)
 *   key_pos, slopes_log2   the additive ALiBi term of the reference's MPT attention (promptcache/model/mpt.py:90-110 slopes,
 *              :160-175 bias gathered at the POSITION IDS of the keys): score[q][key] = q.k * softmax_scale + slope[h] *
 *              position_id[key] (the reference's extra -slope * max_pos is constant along a softmax row and drops out).
 *              key_pos: fp32 [B][key_pos_batch_stride] position id of every cached and new key, padded with anything finite
 *              up to a multiple of 64 entries past past_len + q_len, rows 16-byte aligned; slopes_log2: fp32 [H] = slope * log2 e
 *   k_lo, v_lo fp16 residuals of K / V rows from key index lo_row0 on ([B][Hkv][rows][D] with the lo strides; written by
 *              pc_rope_append_ex / pc_gemm): those rows enter the contractions in split precision, as in the reference's
 *              fp32 pass (llama2.py:361-388); staged rows (< lo_row0) are the fp16 values the reference stages
 *              (cache_engine.py:105-106).  lo_row0 = -1: "the rows of this pass" (= past_len, read from past_len_dev when
 *              given; passes of <= 32 rows then run one extra KV split whose workgroup computes the attention over the
 *              pass's own rows in fp32); -2: the residual tail starts at key past_len_dev[1] (decode steps)
 *   counters   optional: B * H uint32 words, ZERO before the first launch that sees them; every launch leaves them
 *              zero.  With them a pass of <= 16 query rows over a long staged cache is ONE launch: each workgroup writes its
 *              split-KV partial through to memory, arrives at its head's counter, and the last arriver merges the partials in
 *              split order (the result does not depend on the arrival order).  Without them the partials are merged by a
 *              second launch -- which measures no slower on MI355X, so since round 6 the PRODUCT library ignores
 *              `counters` (same bits, two launches) and only -DPC_DEV_SWEEPS builds carry the in-launch merge.  One launch at a
 *              time per workspace / counters.
 *   prefix_k, prefix_v   optional (with past_lens): a SHARED KEY PREFIX held somewhere else -- fp16 planes [Hkv][prefix rows][D]
 *              with head stride prefix_head_stride (one layer of the root scaffold's arena).  Batch row b then attends to
 *              prefix rows [0, past_lens[b]) followed by the rows of ITS pass, which `k` / `v` hold from row 0 on: a suffix
 *              batch of the schema encode reads the trunk in place instead of carrying a copy of it in every batch row.
 *              prefix_k_lo / prefix_v_lo: optional residual planes of the prefix rows (same strides); k_lo / v_lo then
 *              cover the pass's rows from their row 0 (lo_row0 = 0).  Many-row kernel only (q_lo given or q_len > 16)
 *   gather_rows, gather_k_plane, gather_v_plane   optional (B = 1; pc_attn_gather_ok says whether this launch shape takes
 *              it): STAGE WHILE READING.  Key row r < past_len is read from where gather_rows[r] (pc_kv_row_table) says it
 *              lies -- plane gather_k_plane + kv_head for K, gather_v_plane + kv_head for V: (layer * 2 + 0|1) * Hkv of this
 *              layer -- and, unless the entry carries PC_KV_ROW_STAGED, written to row r of `k` / `v`.  After the launch
 *              rows [0, past_len) of `k` / `v` hold exactly what pc_kv_gather would have left there.  Rows from past_len on
 *              (this pass's own) are read from `k` / `v` as always.  Implemented by the streaming kernel of <= 32-row passes
 *              (tail mode or >= 256 keys), the tail-mode 64-row kernel, and the ring kernel (33..512 split-precision rows, D = 128)
 *   defer_merge, nsplit_out   optional (B = 1, no counters): a launch of <= 16 query rows that splits the keys (2..8 splits) LEAVES
 *              its partials for a consumer that merges them in its own prologue (pc_gemm_q8 part_o: the o_proj of an LLM.int8
 *              decode step) instead of running the merge launch: part_o fp32 [H * nsplit * q_len][D] at the start of the
 *              workspace, part_ml (running maximum in log2 units, denominator) [H * nsplit * q_len][2] right behind;
 *              slot = (h * nsplit + split) * q_len + row.  *nsplit_out (a HOST word, written before pc_attn returns) = the
 *              partials per row, or 1 when this launch shape merged as usual and the output planes are final
 * ------------------------------------------------------------------------------------------- */
int64_t pc_attn_workspace_bytes(int32_t B, int32_t H, int32_t D, int32_t q_len, int32_t kv_len_max);

typedef struct pc_attn_args {
    uint32_t struct_bytes;
    const void* q; const void* q_lo; int64_t q_batch_stride, q_token_stride;
    const void* k; const void* v; int64_t kv_batch_stride, kv_head_stride;
    void* out; void* out_lo; int64_t out_batch_stride, out_token_stride;
    void* out_frag_hi; void* out_frag_lo;
    int32_t B, H, Hkv, D, q_len, past_len;
    float softmax_scale;
    void* workspace; int64_t workspace_bytes;
    const int32_t* past_len_dev;
    const int32_t* past_lens;
    const float* key_pos; int64_t key_pos_batch_stride; const float* slopes_log2;
    const void* k_lo; const void* v_lo; int64_t lo_batch_stride, lo_head_stride; int32_t lo_row0;
    uint32_t* counters;
    const void* prefix_k; const void* prefix_v; const void* prefix_k_lo; const void* prefix_v_lo; int64_t prefix_head_stride;
    const pc_kv_row* gather_rows; int32_t gather_k_plane, gather_v_plane;
    int32_t defer_merge; int32_t* nsplit_out;
} pc_attn_args;
int pc_attn(const pc_attn_args* args, void* stream);
/* 1 when pc_attn would run `args` (gather_rows ignored) on a kernel that implements gather_rows, else 0 */
int pc_attn_gather_ok(const pc_attn_args* args);

/* ---------------------------------------------------------------------------------------------
 * Elementwise / reduction pieces of the layer stack, fused for the small-q prefill (q_len rows):
 *   pc_rmsnorm      -- LlamaRMSNorm.forward, llama2.py:103-108 (fp32 statistics)
 *   pc_silu_mul     -- act_fn(gate) * up of LlamaMLP.forward, llama2.py:242 (gate_up: [rows][2*inter], fp32 or fp16)
 *   pc_embed_gather -- embed_tokens lookup, llama2.py:869
 * ------------------------------------------------------------------------------------------- */
int pc_rmsnorm(const void* x, const void* weight, void* out, int32_t rows, int32_t hidden, float eps,
               int32_t x_is_f32, void* stream);
int pc_silu_mul(const void* gate_up, void* out, int32_t rows, int32_t inter, int32_t in_is_f32, void* stream);
int pc_embed_gather(const void* table, const int64_t* ids, void* out, int32_t n_tok, int32_t hidden,
                    int32_t vocab, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Weight-streaming projections for the small-q regime (M = B*q_len <= 512), csrc/pc_gemm*.hip: ONE entry point, pc_gemm.
 *   M <= 64:  eight waves split K over the same output tiles; split-precision (hi + lo) activations.
 *   M <= 512: the waves split the rows, weight tiles are staged through LDS by dedicated waves (hi + lo planes when both given).
 *
 * Fragment-major layouts (register image of mfma_f32_16x16x32_f16 operands, fp16):
 *   weights      Wf[N/16][K/32][64][8]   lane l = 16*g + n  holds W[16*tile + n][32*ks + 8*g .. +8]
 *                (W is the nn.Linear [N][K] matrix; build once at load: view(N/16,16,K/32,4,8).permute(0,2,3,1,4))
 *   activations  Xf[M/16 rounded up][K/32][64][8]   lane l = 16*g + m  holds X[16*mt + m][32*ks + 8*g .. +8]
 *                two planes: hi = fp16(x), lo = fp16(x - hi)  (lo may be NULL: single-precision pass)
 *
 * pc_gemm replaces the nn.Linear calls of llama2.py:345-347 (q|k|v fused), :405 (+ residual add :638), :242 (gate/up +
 *   SiLU*up; down + residual add :644) and :1050 (lm_head) when M <= 512.  Fields of pc_gemm_args (NULL / 0 when unused):
 *   struct_bytes  sizeof(pc_gemm_args) of the caller's header (ABI check)
 *   epilogue   PC_GEMM_EPI_STORE     y[m][n]  = sum_k X[m][k] W[n][k]   fp32 [M][ldy]; kslices > 1: the K axis is also split
 *                                    across workgroups, slice s writes its partial sums to y + s*M*ldy (slabs [kslices][M][ldy];
 *                                    pc_rmsnorm_frag adds them to the residual stream in fixed order)
 *              PC_GEMM_EPI_ADD       y[m][n] += ...   fp32 residual stream, in place.  With ks_counters (M <= 32, fp16 weights;
 *                                    17..32 rows: ks_tiles 1, 2 or 4 and twice the scratch bytes):
 *                                    K is cut into `kslices` (1..8) slices that run as separate workgroups of `ks_tiles`
 *                                    (1, 2, 4, 8) output tiles each, and the partial tiles are added INSIDE the launch -- every
 *                                    workgroup writes its partial through to ks_scratch (>= pc_gemm_skinny_ks_scratch_bytes(N,
 *                                    kslices) bytes, 16-byte aligned), arrives at its tile group's counter, the last arriver adds
 *                                    the partials in slice order (deterministic), adds y and stores.  ks_counters: ceil(N/16/
 *                                    ks_tiles) uint32 words, ZERO before the first launch; every launch leaves them zero
 *              PC_GEMM_EPI_SILU      W = [gate ; up] (N = 2*inter): of[m][j] = silu(gate_j) * up_j as fragment planes
 *                                    [M/16][inter/32][64][8] (of_hi, of_lo) for the down projection
 *              PC_GEMM_EPI_GELU      of[m][n] = gelu(...) as fragment planes (Falcon / MPT MLP, falcon.py:726)
 *              PC_GEMM_EPI_QKV_ROPE  the fused q|k|v projection: llama2.py:345-347 (projections), :357-359 (RoPE at the
 *                                    supplied position ids) and :361-364 (KV concat) in ONE launch; N = (H + 2 Hkv) D, M = B q_len.
 *                                    wf: fragment image of the row-PERMUTED [q;k;v] weight: inside every head, tile j (16 rows)
 *                                    holds features 8j..8j+7 followed by their rotary partners D/2+8j..D/2+8j+7.  Outputs:
 *                                    rotated q as split planes q_hi/q_lo [B*q_len][H*D] (token stride q_token_stride); rotated k
 *                                    and v in place at rows [past_len, past_len+q_len) of the arena planes (cs: pc_rope_table);
 *                                    k_lo / v_lo (both or NULL): fp16 residuals of those new rows; the residual row of token tt
 *                                    is tt (lo_base -1: a buffer of this pass's rows), past_len + tt - lo_base (>= 0) or
 *                                    past_len + tt - past_len_dev[1] (-2: ONE residual tail per layer across a prefill and the
 *                                    decode steps after it; llama2.py:361-388 keeps those rows in fp32 for the whole generation)
 *   wf, w_scale   fp16 fragment image; or (w_scale != NULL, M <= 64, K % 64 == 0) an INT8 image [N/16][K/64][64][16] of
 *              offset-binary bytes (q + 128; a lane's 16 bytes = its 8 values of k-step 2s then of 2s + 1) with w_scale fp32 [N]
 *              (row order of the image, 16-byte aligned): q[n][k] = round(w / scale[n]), scale[n] = absmax_k |w[n][k]| / 127 --
 *              the row-wise quantiser LLM.int8 applies to weights (the reference's GPU configs pass load_in_8bit=True,
 *              config/llm_config_*.json:5, eval.py:36-42); y = scale[n] * sum_k q[n][k] x[k]
 *   xf_hi, xf_lo  activation planes; OR
 *   x, norm_weight, eps   (M <= 16, K <= 16384; epilogues STORE / SILU / QKV_ROPE) the fp32 residual stream x [M][K] itself:
 *              y = W . (norm_weight * x) * rsqrt(mean(x^2) + eps) -- LlamaRMSNorm (llama2.py:103-108) folded into the projection
 *              that consumes it, no launch and no pass over x of its own.  Needs |norm_weight * x| < 65504
 *   rows_dev   optional device int32*: the rows that really carry tokens (<= M; B = 1).  Rows behind it are pad rows of a
 *              row-bucketed hipGraph: their results are unspecified and their activations are not loaded
 *   x_scale, corr, ldc, corr_has   LLM.int8 activations (pc_quant_act_i8): the planes hold int8 CODES, x_scale[m] = SCA[m] / 127;
 *              y = (sum_k code_w code_x) * w_scale[n] * x_scale[m] (+ corr[m][n] when *corr_has: pc_outlier_corr); OR
 *              The products run on the int8 MFMA (v_mfma_i32_16x16x64_i8, int32 sums: igemmlt's arithmetic, exact for any K).
 *   x_codes8   (optional, with x_scale, M <= 64) the codes as the int8 operand image pc_quant_act_i8 / pc_rmsnorm_quant_i8 write
 *              next to the fp16 codes plane: the K loop then reads it instead of xf_hi (half the activation bytes, no packing)
 *   row_max_out, flags_out, out_threshold   (PC_GEMM_EPI_SILU, M <= 16) what pc_gemm_q8's down_proj form reads instead of running a
 *              quantiser launch: per output pair-tile and row the largest |fp16 value| below out_threshold, [N/32][16] floats, and one
 *              flag byte per intermediate feature holding an entry at or above it (set-only: the buffer must be zero)
 *   flags, x_raw, w_codes_t, ldt, row_perm   the outlier correction computed INSIDE the launch (M <= 64): the flag bytes of
 *              pc_quant_act_i8 (>= 16384 bytes, zero behind K), the fp16 activations (fragment plane), the transposed int8 weight
 *              codes [K][ldt] (original row order) and, for q|k|v, the image-row -> original-row permutation
 * ------------------------------------------------------------------------------------------- */
#define PC_GEMM_EPI_STORE 0
#define PC_GEMM_EPI_ADD 1
#define PC_GEMM_EPI_SILU 2
#define PC_GEMM_EPI_QKV_ROPE 3
#define PC_GEMM_EPI_GELU 4
typedef struct pc_gemm_args {
    uint32_t struct_bytes;
    int32_t epilogue;
    const void* wf; const float* w_scale;
    const void* xf_hi; const void* xf_lo;
    const float* x; const void* norm_weight; float eps;
    int32_t M, N, K;
    const int32_t* rows_dev;
    float* y; int64_t ldy; void* of_hi; void* of_lo;
    int32_t kslices;
    int32_t ks_tiles; void* ks_scratch; int64_t ks_scratch_bytes; void* ks_counters;
    const float* x_scale; const float* corr; int64_t ldc; const int32_t* corr_has;
    const void* flags; const void* x_raw; const void* w_codes_t; int64_t ldt; const int32_t* row_perm;
    const float* cs; void* q_hi; void* q_lo; int64_t q_token_stride; void* k_arena; void* v_arena;
    int64_t arena_batch_stride, arena_head_stride;
    int32_t B, H, Hkv, D, q_len, past_len, cap;
    const int32_t* past_len_dev; void* k_lo; void* v_lo; int64_t lo_batch_stride, lo_head_stride; int32_t lo_base;
    const void* x_codes8;
    float* row_max_out; void* flags_out; float out_threshold;
} pc_gemm_args;
int pc_gemm(const pc_gemm_args* args, void* stream);
int64_t pc_gemm_skinny_ks_scratch_bytes(int32_t N, int32_t kslices);
/* pc_gemm_part -- o_proj + residual (llama2.py:405, :638) of a ONE-ROW forward (a decode step) on the attention's split-KV
 *   PARTIALS: y[0][n] += sum_k merged[k] W[n][k], where merged = the H * D values attn_combine_kernel would have produced from
 *   part_o / part_ml (pc_attn with defer_merge: nsplit = *nsplit_out in 2..8, layout there), split into fp16 hi + lo as pc_attn's
 *   fragment planes hold them.  wf: the fp16 fragment image of W [N][K = H * D <= 4096].  Bit-identical to pc_attn + its merge
 *   launch + pc_gemm(PC_GEMM_EPI_ADD) on the merged planes; one launch less per layer and decode step (csrc/pc_gemm_part.hip). */
int pc_gemm_part(const void* wf, const float* part_o, const float* part_ml, int32_t nsplit, int32_t H, int32_t D, int32_t N, float* y,
                 void* stream);
/* pc_gemm_q8 -- LLM.int8 projection of M <= 16 rows with the vector-wise activation quantiser INSIDE the launch
 *   (csrc/pc_gemm_q8.h).  Same arithmetic as pc_rmsnorm_quant_i8 / pc_quant_act_i8 followed by pc_gemm with x_scale + flags
 *   (bitsandbytes Linear8bitLt as demo.py:27-29 loads it; llama2.py:345-347, :405, :242 are the projections), without the
 *   quantiser launches: codes, row scales and outlier flags are derived by every workgroup of the projection itself.
 *   wf, w_scale, w_codes_t, ldt, row_perm   the int8 weight image, its scales, the transposed codes (outlier correction) and, for
 *              q|k|v, the image-row -> original-row permutation: as in pc_gemm_args
 *   threshold  the outlier threshold (6.0; <= 0: no outlier columns)
 *   source     (x, norm_weight, eps): the fp32 residual stream, RMSNorm folded in (epilogues QKV_ROPE, SILU), K <= 6144; or
 *              xf_hi: the fp16 activations as a fragment plane [1][K/32][64][8] (epilogues ADD, STORE), K <= 6144; or
 *              xf_hi + row_max [row_max_units][16] + flags_in [>= 16384 bytes, zero behind K] (epilogue ADD, any K <= 16384):
 *              the producer already left the per-tile row maxima and the outlier-column flags (row_max_out / flags_out of the
 *              launch with the SiLU epilogue); K is then cut into kslices (1..8) workgroup slices of ks_tiles (2, 4, 8) output
 *              tiles with the reduction inside the launch (ks_scratch >= pc_gemm_skinny_ks_scratch_bytes(N, kslices), ks_counters:
 *              ceil(N/16/ks_tiles) zeroed uint32 words; as PC_GEMM_EPI_ADD with ks_counters in pc_gemm).  M <= 4 with kslices = 1
 *              and ks_tiles 1 or 2 (decode): the rows' codes are staged once in LDS, every workgroup keeps all of K, no scratch
 *              part_o, part_ml, part_nsplit, part_head_dim (epilogue ADD, M = 1, K = H * part_head_dim <= 4096): the split-KV partials a
 *              pc_attn launch with defer_merge left (2..8 per row) -- the o_proj of a decode step merges them in its prologue
 *              with attn_combine_kernel's arithmetic (same bits), and the merge launch disappears
 *              xf_hi + x_codes8, x_scale, x_flags (epilogues QKV_ROPE, SILU, ADD; K <= 6144): what a quantiser launch left
 *              (pc_quant_act_i8 / pc_rmsnorm_quant_i8: fp16 plane, codes8 operand image, row scales, >= 16384 flag bytes) -- no quantiser
 *              arithmetic in this launch, but this file's K loop (image copied to LDS once, activation operands read from there, two
 *              weight blocks in flight); bit-identical to pc_gemm with x_scale + flags.  The form of 5..16 rows, where quantising all
 *              rows in every workgroup costs more than the launch
 *   outputs    y / ldy (STORE, ADD), of_hi (+ optional of_lo) fragment planes (SILU), the q|k|v fields (QKV_ROPE) as in pc_gemm_args
 *   row_max_out, flags_out   (SILU) per output pair-tile and row the largest |fp16 value| below the threshold, [N/32][16] floats, and
 *              one flag byte per intermediate feature holding an entry at or above it (set-only: the buffer must be zero)
 *   flags_clear, clear_bytes   a flag buffer this launch zeroes on the side (a multiple of 16 bytes) -- the o_proj launch clears
 *              the buffer the following SiLU launch sets; the down_proj form clears the first quantiser's of the next layer
 *   dbg_codes, dbg_scale, dbg_flags   (tests) workgroup 0 writes its operand image [K/64][64][16] int8, x_scale [M] and the K flag bytes */
typedef struct pc_gemm_q8_args {
    uint32_t struct_bytes;
    int32_t epilogue;
    const void* wf; const float* w_scale; const void* w_codes_t; int64_t ldt; const int32_t* row_perm;
    float threshold;
    const float* x; const void* norm_weight; float eps;
    const void* xf_hi;
    const float* row_max; int32_t row_max_units; const void* flags_in;
    int32_t M, N, K;
    float* y; int64_t ldy; void* of_hi; void* of_lo;
    float* row_max_out; void* flags_out;
    void* flags_clear; int32_t clear_bytes;
    int32_t kslices, ks_tiles; void* ks_scratch; int64_t ks_scratch_bytes; void* ks_counters;
    const float* cs; void* q_hi; void* q_lo; int64_t q_token_stride; void* k_arena; void* v_arena;
    int64_t arena_batch_stride, arena_head_stride;
    int32_t B, H, Hkv, D, q_len, past_len, cap;
    const int32_t* past_len_dev; void* k_lo; void* v_lo; int64_t lo_batch_stride, lo_head_stride; int32_t lo_base;
    void* dbg_codes; float* dbg_scale; void* dbg_flags;
    const float* part_o; const float* part_ml; int32_t part_nsplit, part_head_dim;
    const void* x_codes8; const float* x_scale; const void* x_flags;
} pc_gemm_q8_args;
int pc_gemm_q8(const pc_gemm_q8_args* args, void* stream);
/* pc_rmsnorm_frag -- LlamaRMSNorm (llama2.py:103-108) on the fp32 residual stream, output as fragment planes;
 *   optional prologue x += slabs[0] + ... + slabs[nslabs-1] (each [rows][hidden], the residual adds of
 *   llama2.py:638 / :644 fed by a K-sliced pc_gemm), written back to x in place. */
int pc_rmsnorm_frag(float* x, const void* weight, void* xf_hi, void* xf_lo, int32_t rows, int32_t hidden,
                    float eps, const float* slabs, int32_t nslabs, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Falcon adapter (promptcache/model/falcon.py; multi-query, parallel attention + MLP): the ops the Llama path does
 * not have.  Everything else (pc_kv_gather with Hkv = 1, pc_rope_*, pc_attn with H/Hkv = H, pc_gemm*) is shared.
 *   pc_layernorm       torch.nn.LayerNorm (falcon.py:757, :1020) of fp32 x [rows][hidden] -> fp16
 *   pc_layernorm_frag  the same, written as split-precision fragment planes, with the slab-folding prologue of
 *                      pc_rmsnorm_frag (x += slabs[0] + ... first: the o_proj and dense_4h_to_h partial sums)
 *                      (bias may be NULL in both: MPT's LayerNorm carries no bias, mpt.py:207, :215)
 *   pc_gelu            nn.GELU() (falcon.py:726, mpt.py:187: erf form) fp32 -> fp16, n elements (n % 8 == 0)
 *   pc_gemm epilogue PC_GEMM_EPI_GELU: of[m][j] = gelu(y[m][j]) as fragment planes [M/16][N/32][64][8] (dense_h_to_4h)
 * ------------------------------------------------------------------------------------------- */
int pc_layernorm(const float* x, const void* weight, const void* bias, void* out, int32_t rows, int32_t hidden,
                 float eps, void* stream);
int pc_layernorm_frag(float* x, const void* weight, const void* bias, void* xf_hi, void* xf_lo, int32_t rows,
                      int32_t hidden, float eps, const float* slabs, int32_t nslabs, void* stream);
int pc_gelu(const float* x, void* out, int64_t n, void* stream);


/* Split-precision variants for the dense (many-row) path.  fp16 activations cost 2^-11 per projection input; over 32
 * layers that alone moves 7b-shape logits by 2-3e-2 against the reference's fp32 path.  These write an activation as
 * hi = fp16(v) and lo = fp16(v - hi); the caller stacks [hi; lo] along the rows of one GEMM and adds the two halves of
 * the product (gate_up2 / x2 / pc_add3 take the second half):
 *   pc_rmsnorm_split, pc_layernorm_split   norm -> (hi, lo) [rows][hidden]
 *   pc_silu_mul_split   silu(g) * u of (gate_up + gate_up2) -> (hi, lo);  pc_gelu_split  gelu(x + x2) -> (hi, lo)
 *   pc_add3             x += a + b  (fp32 residual stream)
 *   (the attention takes the split planes through pc_attn: q_lo, out_lo, k_lo / v_lo) */
int pc_rmsnorm_split(const float* x, const void* weight, void* out_hi, void* out_lo, int32_t rows, int32_t hidden,
                     float eps, void* stream);
int pc_layernorm_split(const float* x, const void* weight, const void* bias, void* out_hi, void* out_lo, int32_t rows,
                       int32_t hidden, float eps, void* stream);
int pc_silu_mul_split(const float* gate_up, const float* gate_up2, void* out_hi, void* out_lo, int32_t rows,
                      int32_t inter, void* stream);
int pc_gelu_split(const float* x, const float* x2, void* out_hi, void* out_lo, int64_t n, void* stream);
int pc_add3(float* x, const float* a, const float* b, int64_t n, void* stream);
/* pc_rope_append_ex -- pc_rope_append that also writes the fp16 residuals of the appended K / V rows (k_lo, v_lo,
 *   both or neither: [B][Hkv][rows][D] with the given strides, row = key index - lo_row0; lo_row0 = past_len for a
 *   compact buffer of the new rows, 0 for an arena-shaped one that also carries residuals of earlier rows).  pc_attn consumes them: the keys / values a pass appends enter
 *   (in2_offset != 0: every q / k_new / v_new input element is x[i] + x[i + in2_offset], the two row halves a stacked
 *   [hi; lo] projection leaves)
 *   its own attention in split precision, as in the reference's fp32 pass (llama2.py:361-388), while the arena keeps the
 *   fp16 value the reference stages (cache_engine.py:105-106).  Staged rows (< past_len) have no residual. */
int pc_rope_append_ex(const void* q, int64_t q_batch_stride, int64_t q_token_stride, void* q_out, void* q_out_lo,
                      int64_t qo_batch_stride, int64_t qo_token_stride, const void* k_new, const void* v_new,
                      int64_t kv_new_batch_stride, int64_t kv_new_token_stride, void* k_arena, void* v_arena,
                      int64_t arena_batch_stride, int64_t arena_head_stride, const float* cs, int32_t B, int32_t H,
                      int32_t Hkv, int32_t D, int32_t q_len, int32_t past_len, int32_t cap, int32_t in_is_f32,
                      const int32_t* past_len_dev, void* k_lo, void* v_lo, int64_t lo_batch_stride,
                      int64_t lo_head_stride, int32_t lo_row0, int64_t in2_offset, void* stream);


/* Ragged-prefix batches (schema encode: scaffold suffixes of different unions in one batch, each over its own trunk
 * prefix -- the batched form of cache_engine.py:217-304's per-scaffold forwards): pc_rope_append_ex (and pc_attn.past_lens) with
 * ONE PAST LENGTH PER BATCH ROW.  past_lens: device int32[B]; past_len = their maximum (bounds check, KV-split sizing).
 * Batch row b appends its q_len new rows at arena rows [past_lens[b], past_lens[b] + q_len) and attends to keys
 * [0, past_lens[b]) plus the new rows up to its own (index-order mask, llama2.py:62-76).  Residual planes, when given,
 * are arena-shaped (lo_row0 = 0). */
int pc_rope_append_var(const void* q, int64_t q_batch_stride, int64_t q_token_stride, void* q_out, void* q_out_lo,
                       int64_t qo_batch_stride, int64_t qo_token_stride, const void* k_new, const void* v_new,
                       int64_t kv_new_batch_stride, int64_t kv_new_token_stride, void* k_arena, void* v_arena,
                       int64_t arena_batch_stride, int64_t arena_head_stride, const float* cs, int32_t B, int32_t H,
                       int32_t Hkv, int32_t D, int32_t q_len, int32_t past_len, int32_t cap, int32_t in_is_f32,
                       const int32_t* past_lens, void* k_lo, void* v_lo, int64_t lo_batch_stride, int64_t lo_head_stride,
                       int32_t lo_row0, void* stream);




/* Many-row projection (schema encode / no-cache prefill / long questions; MFMA-bound), csrc/pc_gemm_dense.hip:
 *     acc[m][n] = sum_k (x_hi[m][k] + x_lo[m][k]) * w[n][k]        fp16 operands, fp32 accumulation
 * x_hi / x_lo: split-precision activation planes [M][K] (row stride ldx halfs; x_lo may be NULL), w: the nn.Linear
 * weight [N][K] (row stride ldw), w_scale: optional fp32 per-output-row scale (int8 codes held in fp16).
 * Replaces the nn.Linear calls of promptcache/model/llama2.py:345-347 (q|k|v), :405 (o_proj), :242 (gate / up / down),
 * :1050 (lm_head) -- and falcon.py:393-405,:726-731 / mpt.py -- for passes of many rows, together with the ops
 * that follow them:
 *   epilogue 0  y[m][n]  = acc                                   (fp32, row stride ldy)
 *   epilogue 1  y[m][n] += acc                                   residual add, llama2.py:638 / :644
 *   epilogue 2  out[m][j] = silu(acc[m][j]) * acc[m][N/2 + j]    w = [gate; up] rows, llama2.py:242; out_hi / out_lo
 *                                                                split-precision planes [M][N/2] (row stride ldo)
 *   epilogue 4  out[m][n] = gelu(acc)                            nn.GELU(), falcon.py:726; planes [M][N]
 * K % 8 == 0, N % 4 == 0; operands 16-byte aligned.  Returns 0 or a negative PC_ERR_* code. */
int pc_gemm_dense(const void* x_hi, const void* x_lo, int64_t ldx, const void* w, int64_t ldw, const float* w_scale,
                  int32_t M, int32_t N, int32_t K, int32_t epilogue, float* y, int64_t ldy, void* out_hi, void* out_lo,
                  int64_t ldo, void* stream);

/* The fused q|k|v projection of a many-row pass (head_dim 128): replaces llama2.py:345-347 (q / k / v projections), :357-359 (RoPE at the
 * supplied position ids) and :361-364 (KV concat) for schema-encode passes, no-cache prefill and long questions -- one launch,
 * no [M][(H + 2 Hkv) D] fp32 intermediate.
 *   x_hi, x_lo   fp16 [M = B*q_len][K] activation planes (x_lo may be NULL), row stride ldx
 *   w            fp16 [(H + 2 Hkv) * 128][K] = [q; k; v] weight rows as nn.Linear holds them (NOT permuted), row stride ldw
 *   cs           fp32 [M][64][2] (cos, sin) from pc_rope_table at the position id of every row
 *   q_hi, q_lo   rotated q as split planes fp16 [M][q_token_stride] (q_lo may be NULL)
 *   k_arena, v_arena   layer planes [B][Hkv][cap][128] with the arena strides: rotated k and v of batch row b land at rows
 *                past + t (past = past_lens[b] when given, else past_len)
 *   k_lo, v_lo   optional residual planes of those rows ([B][Hkv][rows][128] with the lo strides, row = past + t - lo_row0) */
typedef struct pc_dense_qkv_args {
    uint32_t struct_bytes;
    const void* x_hi; const void* x_lo; int64_t ldx;
    const void* w; int64_t ldw; int32_t K;
    const float* cs;
    void* q_hi; void* q_lo; int64_t q_token_stride;
    void* k_arena; void* v_arena; int64_t arena_batch_stride, arena_head_stride;
    void* k_lo; void* v_lo; int64_t lo_batch_stride, lo_head_stride; int32_t lo_row0;
    int32_t B, H, Hkv, D, q_len, past_len, cap;
    const int32_t* past_lens;
    /* optional (round 5, instead of x_lo): the residual plane as int8 codes + row scales against an int8 weight image + row
     * scales (dev builds only: csrc/pc_dev.h, pc_gemm_dense_lo8) */
    const void* x_lo8; const float* x_lo8_scale; int64_t ldx8;
    const void* w8; const float* w8_scale; int64_t ldw8;
} pc_dense_qkv_args;
int pc_gemm_dense_qkv_rope(const pc_dense_qkv_args* args, void* stream);

/* pc_gemm_dense with a caller-owned scratch buffer (16-byte aligned, workspace_bytes long).  A plain-store / residual-add
 * launch whose tile grid would leave most of the 256 CUs idle -- the N = hidden projections (llama2.py:405 o_proj, :242
 * down_proj) at a few hundred rows -- cuts K into up to 8 slices: partial slabs [slices][M][N] fp32 in the workspace, added
 * in slice order by a second launch (deterministic).  When the slabs do not fit, or the grid is large enough, the call is
 * exactly pc_gemm_dense. */
int pc_gemm_dense_ws(const void* x_hi, const void* x_lo, int64_t ldx, const void* w, int64_t ldw, const float* w_scale,
                     int32_t M, int32_t N, int32_t K, int32_t epilogue, float* y, int64_t ldy, void* out_hi, void* out_lo,
                     int64_t ldo, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- LLM.int8(): int8 weights AND int8 activations with the fp16 outlier decomposition -------------------------------
 * What load_in_8bit=True means in the reference's GPU runs (demo.py:27-29, eval.py:36-42, config/llm_config_*.json:5 ->
 * transformers -> bitsandbytes.nn.Linear8bitLt(threshold = 6.0)).  bitsandbytes is not part of the reference tree: these
 * entry points implement its published algorithm (Dettmers et al., NeurIPS 2022, section 3; csrc/pc_int8.hip; CPU
 * restatement oracle/llmint8_oracle.py) and replace, per decoder-layer nn.Linear, the call Linear8bitLt.forward makes.
 *
 * pc_quant_act_i8: x = fp16 activations, row-major [T][ldx] (frag = 0) or a fragment-major plane [T/16][K/32][64][8]
 *   (frag = 1; pc_gemm.hip).  Per row: entries |x| >= threshold are outliers (their column gets flags_set[k] = 1), the
 *   rest is quantised vector-wise: codes = round_half_even(x * 127 / absmax_row) as fp16 values in the layout of x,
 *   x_scale[t] = absmax_row / 127.  flags_clear (optional, clear_len bytes): zeroed for the next activation slot.
 * pc_outlier_corr: corr[t][n] = sum over flagged columns k of x[t][k] * fp16(w_codes[r][k] * w_scale[r])
 *   - codes[t][k] * w_codes[r][k] * x_scale[t] * w_scale[r],  r = row_perm ? row_perm[n] : n  (w_codes_t: the int8 weight
 *   codes TRANSPOSED, [K][ldt], so that a weight column is contiguous); *has = 1 when any column is flagged, else 0 and
 *   corr is left untouched.
 * pc_gemm (x_scale != NULL) / pc_gemm_dense_a8: the projections over the codes,
 *   y = (sum_k w_code[n][k] * x_code[m][k]) * w_scale[n] * x_scale[m] + (*corr_has ? corr[m][n] : 0), then the epilogue
 *   (same epilogue codes and outputs as with fp16 operands). */
int pc_quant_act_i8(const void* x, int64_t ldx, int32_t frag, int32_t T, int32_t K, void* codes, float* x_scale, void* flags_set,
                    void* flags_clear, int32_t clear_len, float threshold, void* codes8, void* stream);
/* codes8 (optional; K % 64 == 0, 16-byte aligned, ceil(T/16) * K * 16 bytes): the same codes as the int8 MFMA's operand image
 * [T/16][K/64][64][16] -- a lane's 16 signed bytes = its eight codes of k-step 2s, then of 2s + 1, the byte order of the int8
 * weight image -- for pc_gemm's x_codes8: the projection then loads half the activation bytes and converts nothing. */
/* (pc_gemm with `flags` computes the outlier correction INSIDE the projection launch, M <= 64: every workgroup compacts the flags
 * and its eight waves share the outlier columns; equal to pc_outlier_corr + the corr form up to the fp32 summation order over the
 * outlier columns.  Saves one launch per projection, ~4 us even when no column is flagged.) */
/* pc_rmsnorm_frag + pc_quant_act_i8 in one launch (T <= 64 rows, fragment planes), for the two projection inputs that come out of
 * an RMSNorm (input_layernorm -> q|k|v, post_attention_layernorm -> gate|up; llama2.py:628, :640): bit-identical to the pair.
 * x: fp32 residual stream [T][hidden]; x_hi: the normalised fp16 activations (read by pc_outlier_corr), codes / x_scale / flags as
 * pc_quant_act_i8. */
int pc_rmsnorm_quant_i8(const float* x, const void* norm_weight, float eps, int32_t T, int32_t hidden, void* x_hi, void* codes,
                        float* x_scale, void* flags_set, void* flags_clear, int32_t clear_len, float threshold, void* codes8,
                        void* stream);
int pc_outlier_corr(const void* flags, int32_t K, const void* x, const void* codes, int64_t ldx, int32_t frag,
                    const float* x_scale, const void* w_codes_t, int64_t ldt, const float* w_scale, const int32_t* row_perm,
                    int32_t T, int32_t N, float* corr, int64_t ldc, int32_t* has, void* stream);
int pc_gemm_dense_a8(const void* xq, int64_t ldx, const void* w_codes, int64_t ldw, const float* w_scale, const float* x_scale,
                     const float* corr, int64_t ldc, const int32_t* corr_has, int32_t M, int32_t N, int32_t K, int32_t epilogue,
                     float* y, int64_t ldy, void* out_hi, void* out_lo, int64_t ldo, void* stream);

/* pc_fetch_block -- replaces the per-call uploads `torch.tensor(ids, device=)` / `torch.tensor(position_ids, device=)` of
 * generation_engine.py:96-97 (and :125-132 for every decode step): the FIRST node of a captured forward pulls the call's whole
 * input block -- token ids, position ids, past length, staging plan -- out of one pinned (device-mapped) host buffer into device
 * memory; per call the host writes that buffer and replays the graph, nothing else.  host_src: pinned host memory (8-byte
 * aligned, read with system-scope loads); nbytes: multiple of 8, <= 1 MiB.  The host must not rewrite the buffer before the
 * replay that reads it has finished. */
int pc_fetch_block(const void* host_src, void* dst, int32_t nbytes, void* stream);

/* pc_prefill_prologue -- everything a captured small-q forward does before its first layer, as ONE launch reading the call's pinned
 * host block (layout: int64 ids[n_tok] | int32 pos[n_tok] at o_pos | int32 words[8] at o_words = {past_len, tail base, live rows,
 * segments, rows of the row table, ...} | pc_kv_seg[max_seg] at o_segs): the block copied to its device twin (pc_fetch_block), the
 * embedding rows of the tokens as the fp32 residual stream x_out [n_tok][hidden] (embed_tokens, llama2.py:869), the (cos, sin)
 * rows of the supplied positions cs_out [n_tok][head_dim/2][2] (pc_rope_table; llama2.py:129-147, :204-207) and -- when `rows` is
 * given -- the staging plan expanded per staged row (pc_kv_row_table; words[3] segments, words[4] rows).  words[5] != 0: the
 * caller's ids / positions are DEVICE tensors (the reference's convention, generation_engine.py:96-97) that it copied into the
 * ids | pos region of dev_block on `stream` before this launch -- they are read there and that region is not overwritten (no
 * host read-back of device inputs).  Replaces the uploads of generation_engine.py:96-97 and four small launches per forward. */
int pc_prefill_prologue(const void* host_block, void* dev_block, int32_t nbytes, int32_t n_tok, int32_t o_pos, int32_t o_words,
                        int32_t o_segs, int32_t max_seg, const void* embed_table, int32_t hidden, int32_t vocab, float* x_out,
                        const float* inv_freq, int32_t head_dim, float* cs_out, pc_kv_row* rows, const void* dst, int32_t max_ctx,
                        void* stream);

/* Greedy decode without a host round trip per token (generation_engine.py:123-168, greedy branch :159): the tail of a
 * captured decode step.  token = argmax(logits[0..vocab)) (lowest index among equal maxima); ids[0] = token, pos[0] += 1,
 * past_len[0] += 1 -- the device words the NEXT replay of the same hipGraph reads its token id, position id and past
 * length from -- and ring[counter % ring_cap] = token, counter += 1 for the host to collect tokens when it wants them. */
int pc_greedy_advance(const float* logits, int32_t vocab, int64_t* ids, int32_t* pos, int32_t* past_len, int32_t* ring,
                      int32_t* counter, int32_t ring_cap, void* stream);

/* Diagnostics used by the GPU test-suite: dumps the MFMA C/D lane map and the LDS transpose-read
 * map the attention kernel relies on (probe_kernel in csrc/pc_misc.hip). */
/* dev hook (tools/gemm_trace.py): weight-streaming launches issued by this thread stamp per-wave wall-clock times
 * (entry, K loop done, reduced, done) into buf [workgroups][8][4] uint64; NULL switches it off.  No reference counterpart. */
int pc_dev_gemm_trace(void* buf);
int pc_dev_attn_trace(void* buf);   /* the same for attn_small_kernel: [workgroups][4][4] (entry, first tile scored, slice done, done) */
int pc_probe_layouts(float* out_mfma /*[16*16]*/, float* out_tr /*[512]*/, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PROMPTCACHE_HIP_H */
