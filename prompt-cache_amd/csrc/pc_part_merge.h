// The split-KV merge of pc_attn's partials (attn_combine_kernel, pc_attn.hip) as a device function, for consumers that merge in
// their own prologue (pc_attn defer_merge): the o_proj launch of an LLM.int8 decode step (pc_gemm_q8.hip, part_o).  Same
// arithmetic, same order, explicit fmas.  (Merging inside the o_proj input QUANTISER of the 12-row cached step was built and
// measured: no gain -- the quantiser has one workgroup per row, the merge launch 384; profiles/r05_variants.txt.)
//     out = sum_s 2^(m_s - m*) O_s / sum_s 2^(m_s - m*) l_s        (split order; llama2.py:385-399's softmax . V, re-associated)
// Partials (B = 1): part_o fp32 [H * nsplit * q_len][D], part_ml [H * nsplit * q_len][2] = (running maximum in log2 units,
// denominator); slot = (h * nsplit + split) * q_len + row.
#pragma once
#include "pc_common.h"

namespace pcm {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kPartNS = 8;                               // most partials per row a consumer-side merge takes

struct PartSrc { const float* part_o; const float* part_ml; int32_t nsplit, D, q_len; };
struct PartLoads { float mv[kPartNS], lv[kPartNS]; f4 oa[kPartNS], ob[kPartNS]; };

// every load of the 8-feature chunk starting at feature k0 of query row `row` (one batch of independent loads; splits behind
// nsplit re-read the last one and get weight zero)
__device__ __forceinline__ void part_issue(const PartSrc& ps, int row, int k0, PartLoads& L) {
    const int D = ps.D, ns = ps.nsplit;
    const int h = k0 / D, d0 = k0 - h * D;
#pragma unroll
    for (int s = 0; s < kPartNS; ++s) {
        const int sc = s < ns ? s : ns - 1;
        const int64_t slot = ((int64_t)h * ns + sc) * ps.q_len + row;
        const float2 ml = *(const float2*)(ps.part_ml + slot * 2);
        L.mv[s] = s < ns ? ml.x : -1.0e30f;
        L.lv[s] = ml.y;
        L.oa[s] = *(const f4*)(ps.part_o + slot * D + d0);
        L.ob[s] = *(const f4*)(ps.part_o + slot * D + d0 + 4);
    }
}

// -> the fp16 hi parts of the eight merged values (and their residuals when lo != nullptr)
__device__ __forceinline__ h8 part_merge(const PartSrc& ps, const PartLoads& L, h8* lo = nullptr) {
    const int ns = ps.nsplit;
    float mstar = -1.0e30f;
#pragma unroll
    for (int s = 0; s < kPartNS; ++s) mstar = fmaxf(mstar, L.mv[s]);
    float num[8], den = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) num[e] = 0.f;
#pragma unroll
    for (int s = 0; s < kPartNS; ++s) {
        const float w = s < ns ? exp2f(L.mv[s] - mstar) : 0.f;
        den = __builtin_fmaf(w, L.lv[s], den);
#pragma unroll
        for (int e = 0; e < 8; ++e) num[e] = __builtin_fmaf(w, e < 4 ? L.oa[s][e] : L.ob[s][e - 4], num[e]);
    }
    h8 out;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        _Float16 vh, vl;
        pc_split(num[e] / den, vh, vl);
        out[e] = vh;
        if (lo) (*lo)[e] = vl;
    }
    return out;
}

}  // namespace pcm
