// Shared by the attention translation units (pc_attn.hip, pc_attn_ring.hip): launch parameters, operand types and the small
// device helpers every attention kernel uses.  See pc_attn.hip for the formulation.
#pragma once
#include <hip/hip_fp16.h>

#include "pc_common.h"

namespace pca {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kTK = 64;        // keys per LDS tile
constexpr int kThreads = 256;  // 4 waves
constexpr int kQB = 64;        // query rows per workgroup
constexpr int kMaxSplit = 32;
constexpr int kSmallQ = 16;    // most query rows per batch row attn_small_kernel takes
constexpr float kNegBig = -1.0e30f;  // finite "-inf" for the running max

struct AttnParams {
    const _Float16* q; int64_t q_bs, q_ts;
    const _Float16* q_lo;   // optional low-order plane of q (same strides), consumed when HP
    const _Float16* k; const _Float16* v; int64_t kv_bs, kv_hs;
    _Float16* out; int64_t o_bs, o_ts;
    _Float16* out_lo;       // optional: fp16 residual of `out` (same strides): split-precision row-major output
    _Float16* of_hi; _Float16* of_lo;   // optional: fragment-major split-precision output planes (pc_gemm.hip)
    float* part_o; float* part_ml;
    // fused split-KV merge (attn_small_kernel): one arrival counter per (batch row, head), zero before the first launch; every
    // launch leaves them zero.  NULL: the partials are merged by attn_combine_kernel in a second launch.
    uint32_t* counters;
    unsigned long long* trace;   // dev (pc_dev_attn_trace): per-wave wall-clock stamps [workgroup][4 waves][4]
    const int32_t* past_len_dev;
    const int32_t* past_lens;   // optional [B]: one past length per batch row (ragged prefixes); p.past_len = their maximum
    // ALiBi (MPT, promptcache/model/mpt.py:90-110, :160-175): score += slope[h] * key_pos[b][key]; both pre-scaled to
    // the log2 domain by the host (slope * log2 e), key_pos = the POSITION ID of each cached / new key
    const float* key_pos; int64_t kp_bs; const float* slopes;
    // optional fp16 residuals of the NEW keys / values of this pass (rows past_len ..), compact [B][Hkv][q_len][D]:
    // the pass's own K/V then enter the MFMAs in split precision (staged rows are exact fp16 as the reference stages them)
    const _Float16* k_lo; const _Float16* v_lo; int64_t lo_bs, lo_hs; int32_t lo_row0;   // row = key - lo_row0
    // optional shared key prefix (attn_fwd_kernel, with past_lens): keys [0, past_lens[b]) of batch row b are rows of these
    // planes ([Hkv][rows][D], head stride pre_hs: the root scaffold's arena), keys from past_lens[b] on are rows 0.. of k / v
    // (and of k_lo / v_lo).  pre_k_lo / pre_v_lo: residuals of the prefix rows (NULL: the prefix is plain fp16).
    const _Float16* pre_k; const _Float16* pre_v; const _Float16* pre_k_lo; const _Float16* pre_v_lo; int64_t pre_hs;
    // optional (attn_small_kernel<.., GATHER>): one entry per key row -- where the row lies and whether the launch writes it to
    // k / v (pc_kv_row_table); planes g_kplane + kv head / g_vplane + kv head of the entry's source
    const pc_kv_row* rows; int32_t g_kplane, g_vplane;
    int32_t H, Hkv, q_len, past_len, nsplit;
    // tail != 0 (lo_row0 < 0 and q_len <= kTailMax: prefill of a short prompt over a staged cache): splits
    // 0 .. nsplit-2 stream the STAGED keys [0, past_len) only; the workgroup of split nsplit-1 computes the attention over
    // the rows this pass appended in fp32 (attn_tail_block) and leaves it as one more partial for the merge kernel.
    int32_t tail;
    int32_t small;          // attn_small_kernel launch (host-side dispatch flag)
    int32_t formal_handoff; // fused merge: acq_rel arrival (the C++-memory-model form, env PC_FORMAL_HANDOFF=1) instead of relaxed + vmcnt(0)
    int32_t xcd_remap, nqblk, nbatch;
    // (host) the staged keys go to attn_wide_kernel in `nsplit - 1` slices, the keys this pass appended to attn_ring_kernel in its
    // own_only mode as partial nsplit - 1 (pc_attn_wide.hip)
    int32_t wide, wide_nsplit, own_only;
    int32_t defer_merge;    // (host) leave the split-KV partials in the workspace: the consumer merges them (pc_gemm_q8 part_o)
    int32_t* nsplit_out;    // (host) where pc_attn reports how many partials per row it left (1: the output planes are final)
    float scale_log2;
};

// position of element (row m, feature k) in a fragment-major plane with KS k-steps (see pc_gemm.hip)
__device__ __forceinline__ int64_t frag_off(int m, int k, int KS) {
    return ((((int64_t)(m >> 4) * KS + (k >> 5)) * 64) + ((k & 31) >> 3) * 16 + (m & 15)) * 8 + (k & 7);
}

// v_exp_f32 directly: arguments here are <= 0 (or -inf), so the denormal-range scaling that exp2f() wraps
// around the instruction (v_cmp + v_cndmask + v_ldexp per call) buys nothing.
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Inter-workgroup hand-off inside one launch (fused split-KV merge): 8-byte agent-scope relaxed atomics on BOTH sides --
// the stores go through to the coherence point (no release fence), the loads bypass this CU's L1 (no acquire fence).
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

__device__ __forceinline__ h4 lds_tr_read(const _Float16* p) {
    // ds_read_b64_tr_b16: within a 16-lane group, lane i receives sub-element (i%4) of the 8 bytes
    // addressed by lanes {i/4, 4+i/4, 8+i/4, 12+i/4} (verified by pc_probe_layouts on hardware).
    s4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(p));
    return __builtin_bit_cast(h4, r);
}

__device__ __forceinline__ void glds16(const _Float16* g, char* lds_wave_base) {
    // LDS-DMA: 16 bytes per lane, lane-linear destination (1 KiB per wave-instruction), any per-lane source address
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ void glds16_nt(const _Float16* g, char* lds_wave_base) {
    // the same with the non-temporal cache policy (aux = 2): rows that are read exactly once by exactly one CU
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 2);
}

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
// (hi, lo) fp16 pair planes of two fp32 values: hi = fp16(e), lo = fp16(e - hi), packed two per register.  The residual is ONE
// v_fma_mix per value (f16 source widened inside the fma, result rounded to f16 into the low / high half) instead of
// v_cvt_f32_f16 + v_sub_f32 + v_cvt_f16_f32 + a pack -- the softmax is a third of this kernel's issue slots.
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair(float e0, float e1, uint32_t& hi, uint32_t& lo) {
    const h2 hh = {(_Float16)e0, (_Float16)e1};
    hi = __builtin_bit_cast(uint32_t, hh);
    uint32_t d;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(d) : "v"(hi), "v"(e0), "v"(e1));
    lo = d;
}

// LDS-DMA issued as raw instructions: 16 bytes per lane from `g` to LDS byte address `lds_addr` + 16 * lane.  The builtin form
// makes hipcc wait vmcnt(0) in front of every ds_read_b64_tr_b16 that follows (it cannot tell the transposing reads from
// the buffer the DMA is filling), which would serialise the ring; here the waits are the explicit ones in the ring loop.
__device__ __forceinline__ void glds16_raw(const _Float16* g, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_addr) : "memory", "m0");
}

// the many-row ring kernel (pc_attn_ring.hip): 128 query rows per workgroup, K / V tiles by LDS-DMA
bool ring_eligible(const AttnParams& p, int D);
int ring_min_rows();
int ring_nsplit(int B, int H, int q_len, int kv_len);
int launch_attn_ring(const AttnParams& p, int B, hipStream_t stream);
// a long question over a long staged cache (pc_attn_wide.hip): all query rows of a head per workgroup, slices of the staged keys
bool wide_eligible(const AttnParams& p, int D, int B);
int wide_nsplit(int H, int q_len, int past_len);
int wide_min_keys();
int launch_attn_wide(const AttnParams& p, hipStream_t stream);

}  // namespace pca
