// Entry points that exist only in -DPC_DEV_SWEEPS builds (PC_BUILD_FLAGS=-DPC_DEV_SWEEPS python __graft_entry__.py --force): measured
// negative results kept for A/B, NOT part of the drop-in boundary (include/promptcache_hip.h) and not in the product library.
//   pc_gemm_chain        one persistent launch for o_proj -> gate|up -> down -> next q|k|v: bit-identical to the four launches and
//                        SLOWER (rounds 2, 3, 5: 106-121 vs 88 us per layer; 233 vs 330 tok/s at one row; profiles/r02_chain_trace.txt,
//                        profiles/r05_variants.txt r5j)
//   pc_gemm_dense_lo8    the residual activation plane of the many-row projections on the int8 MFMA: +2 % encode throughput for
//                        6.7 GB of int8 weight images at 7b (profiles/r05_dense_lo8_ab.txt)
//   pc_gemm_part_rows    o_proj of a 2..16-row cached step on the attention's split-KV partials, K sliced across workgroups (round 6,
//                        VERDICT r5 Next 2's second half): correct and deterministic, and SLOWER than merge launch + o_proj -- 19.1 us
//                        against 4.7 + 10.0 at the persona shape: every workgroup pulls 256 KiB of fp32 partials next to its 128 KiB
//                        of weights (profiles/r06_variants.txt r6t)
#pragma once
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* pc_gemm_dense_lo8 / pc_quant_rows_i8 -- the many-row projections with the RESIDUAL activation plane on the int8 MFMA (round 5).
 * The split-precision path multiplies every weight fragment twice (x_hi and x_lo = fp16(x - x_hi)) so that a projection sees
 * ~22-bit activations, as the reference's fp32 CPU path does (llama2.py:345-347, :405, :242).  The residual
 * plane only has to be good to a few bits: pc_quant_rows_i8 turns x_lo [M][K] into int8 codes with one scale per row
 * (code = round_half_even(x_lo * 127 / max_k |x_lo|), scale = max / 127), the weights exist a second time as row-wise absmax
 * int8 codes [N][K] + scales (the LLM.int8 weight quantiser, promptcache_amd._native.quantize_rows_int8), and
 *     y = x_hi . W^T  (fp16 MFMA, fp32 sums)  +  (x_lo8 . W8^T as exact int32 sums on v_mfma_i32_32x32x32_i8) * x_lo8_scale[m] * w8_scale[n]
 * -- 24 matrix-pipe slots per K-step instead of 32, and half the residual plane's LDS / L2 traffic.  What it costs in accuracy:
 * the residual is carried to 2^-8 of its row maximum (|x_lo| <= 2^-11 |x|: ~19 bits of the row's largest activation) and the
 * weights of the residual term to 2^-8 of their row maximum (on a term that is 2^-11 of the product).  Same epilogues, tiles and
 * split-K workspace as pc_gemm_dense_ws; K % 64 == 0; x_lo8 / w8 16-byte aligned, ldx8 / ldw8 (bytes per row) % 16 == 0.
 * No reference counterpart (the reference multiplies in fp32); parity: tests/test_gpu_dense.py, tests/test_gpu_fullsize.py. */
int pc_quant_rows_i8(const void* x /* fp16 [M][ldx] */, int64_t ldx, int32_t M, int32_t K, void* codes /* int8 [M][ld8] */,
                     int64_t ld8, float* scale /* [M] */, void* stream);
int pc_gemm_dense_lo8(const void* x_hi, int64_t ldx, const void* x_lo8, const float* x_lo8_scale, int64_t ldx8, const void* w,
                      int64_t ldw, const void* w8, const float* w8_scale, int64_t ldw8, int32_t M, int32_t N, int32_t K,
                      int32_t epilogue, float* y, int64_t ldy, void* out_hi, void* out_lo, int64_t ldo, void* workspace,
                      int64_t workspace_bytes, void* stream);

/* pc_gemm_part_rows -- the same for M = 1..16 rows (the cached prefill of a short question; llama2.py:405, :638 behind :368-398):
 *   y[m][n] += sum_k merged[m][k] W[n][k].  part_o / part_ml as pc_attn(defer_merge) left them for q_len = M rows (pad rows of a
 *   captured graph included; rows_dev, optional: the number of live rows on the device -- rows behind it are not stored).  K = H * D is
 *   cut into kslices (2, 4 or 8) slices across workgroups of kslices output tiles each, added inside the launch in slice order
 *   (deterministic): scratch >= pc_gemm_skinny_ks_scratch_bytes(N, kslices) bytes, counters = N / 16 / kslices zeroed uint32 words
 *   that every launch leaves zero (pc_gemm's ks_scratch / ks_counters may be shared: launches of one stream).  Every lane merges the
 *   operand fragments it multiplies itself (attn_combine_kernel's arithmetic): no merge launch, no activation plane, no activation
 *   load in the K loop.  Differs from the three-launch form only in the fp32 summation order (csrc/pc_gemm_part.hip).
 *   Measured slower than the merge launch + pc_gemm (see the head of this file). */
int pc_gemm_part_rows(const void* wf, const float* part_o, const float* part_ml, int32_t nsplit, int32_t H, int32_t D, int32_t N, int32_t M,
                      const int32_t* rows_dev, float* y, int64_t ldy, int32_t kslices, void* scratch, int64_t scratch_bytes, void* counters,
                      void* stream);

/* ---- pc_gemm_chain: the projections between two attention calls of a <= 16-row forward as ONE persistent launch --------
 *   phase 0  x += attn @ Wo^T                                   o_proj + residual        llama2.py:405, :638
 *   phase 1  act = silu(gate(n2(x))) * up(n2(x))                post_attention_layernorm + gate / up + SiLU  llama2.py:640-643, :242
 *   phase 2  x += act @ Wdown^T                                 down_proj + residual     llama2.py:242, :644
 *   phase 3  (wqkv_f_next != NULL) the NEXT layer's input_layernorm + q|k|v + RoPE + in-place KV append  llama2.py:628, :345-364
 * i.e. pc_gemm(EPI_ADD), pc_gemm(x + norm_weight, EPI_SILU), pc_gemm(EPI_ADD), pc_gemm(x + norm_weight, EPI_QKV_ROPE; norm
 * source) with the same operands, tiles, K split and reduction order: the results are bit-identical to those four launches.
 * One workgroup per CU stays resident across the phases and meets the others at grid barriers; every wave fetches the first
 * weight block of the next phase before it waits, so the HBM stream does not stop at the seams.
 * attn_hi / attn_lo: fragment planes [1][attn_width/32][64][8] (pc_attn out_frag_*), act_hi / act_lo: scratch planes
 * [1][inter/32][64][8]; x: fp32 residual stream [M][hidden], updated in place; weights: fragment images (fp16).
 * sync_state: pc_chain_sync_words() uint32 words of device memory, zeroed ONCE by the caller and then owned by these calls
 * (monotonic counters: no reset between launches or graph replays; launches sharing a state must be stream-ordered).  Every
 * in-kernel wait is bounded: on a timeout word pc_chain_sync_err_word() of the state becomes non-zero, the launch still
 * terminates, its results are invalid and the state must be zeroed again.
 * M <= 16.  Shapes without an instantiation (tile widths other than the 7b / 13b ones) return PC_ERR_ARG: the caller then
 * issues the separate launches. */
int32_t pc_chain_sync_words(void);
int32_t pc_chain_sync_err_word(void);
int pc_gemm_chain(const void* wo_f, const void* attn_hi, const void* attn_lo, int32_t attn_width, float* x, int32_t M,
                  int32_t hidden, const void* wgu_f, const void* ln2_weight, float eps, int32_t inter, void* act_hi,
                  void* act_lo, const void* wdown_f, const void* wqkv_f_next, const void* ln1_weight_next, const float* cs,
                  void* q_hi, void* q_lo, int64_t q_token_stride, void* k_arena, void* v_arena, int64_t arena_batch_stride,
                  int64_t arena_head_stride, int32_t B, int32_t H, int32_t Hkv, int32_t D, int32_t q_len, int32_t past_len,
                  int32_t cap, const int32_t* past_len_dev, void* k_lo, void* v_lo, int64_t lo_batch_stride,
                  int64_t lo_head_stride, int32_t lo_base, void* sync_state, void* stream);

#ifdef __cplusplus
}
#endif
