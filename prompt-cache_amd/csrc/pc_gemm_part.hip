// o_proj + residual (llama2.py:405, :638) of a ONE-ROW forward -- a decode step -- that takes the attention's split-KV PARTIALS as
// its activation source (pc_attn defer_merge) and merges them in its prologue: the merge launch between the attention and o_proj
// (attn_combine_kernel, 4.8 us of a 95 us layer) disappears.
//
//   prologue   every workgroup merges the row's H * D values from the 2..8 partials per head with attn_combine_kernel's arithmetic
//              (csrc/pc_part_merge.h: same order, explicit fmas -> the same bits), splits them into fp16 hi / lo and leaves both as
//              MFMA B-operand fragments of row 0 in LDS ([K/32][4 g][16 B] per plane, 2 * K * 2 bytes); the wave's whole weight share
//              (K / 8 = 16 k-steps of 1 KiB at K = 4096) was requested before, so the stream runs while the row is merged
//   K loop     weight fragments from registers / HBM (two blocks in flight, scalar base + lane offset), activation operands from LDS
//              (every lane reads row 0's fragment of its group g: lanes of rows >= 1 compute columns that are never stored);
//              hi then lo per k-step, k ascending inside wave_k_range's share -- gemm_skinny_kernel<1, 1, EPI_ADD>'s order
//   epilogue   the eight waves' shares through LDS in wave order, + residual, store: bit-identical to pc_attn + attn_combine_kernel +
//              pc_gemm (EPI_ADD) on the merged planes (tests/test_gpu_kernels.py)
#include "pc_gemm_skinny.h"
#include "pc_part_merge.h"

using namespace pcg;

namespace {

struct PartGemmParams {
    GemmParams g;            // wf (fp16 image), y, ldy, M = 1, ntiles, KS
    pcm::PartSrc part;
};

constexpr int kPU = 8;       // k-steps per block

__global__ __launch_bounds__(kThreads) void gemm_part_kernel(const PartGemmParams pp) {
    const GemmParams& p = pp.g;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // [K * 2] hi fragments | [K * 2] lo fragments
    __shared__ __attribute__((aligned(16))) float red[kWaves][64][4];
    const int tid = threadIdx.x, lane = tid & 63, m = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bx = blockIdx.x, KS = p.KS, K = KS * 32, nv = K >> 3;
    // ---- 1. loads: the partials of this thread's chunks, then the weight share ----
    pcm::PartLoads pl;
    pcm::part_issue(pp.part, 0, (tid < nv ? tid : nv - 1) * 8, pl);          // (K <= 4096: one 8-feature chunk per thread)
    int ks0, ks1;
    wave_k_range<false>(p, 0, wave, ks0, ks1);
    const int tile = bx < p.ntiles ? bx : p.ntiles - 1;
    const char* wt = (const char*)p.wf + (int64_t)tile * KS * 1024;
    h8 wA[kPU], wB[kPU];
    auto issue = [&](h8 (&w)[kPU], int kb) {
#pragma unroll
        for (int u = 0; u < kPU; ++u) {
            int k = kb + u < ks1 ? kb + u : ks1 - 1;
            k = k < 0 ? 0 : k;
            w[u] = __builtin_bit_cast(h8, __builtin_nontemporal_load((const u32x4*)(wt + (uint32_t)k * 1024u + (uint32_t)lane * 16u)));
        }
    };
    const int nb = (ks1 - ks0 + kPU - 1) / kPU;
    issue(wA, ks0);
    if (nb > 1) issue(wB, ks0 + kPU);
    const f4 yold = *(const f4*)(p.y + tile * 16 + g * 4);                   // row 0 of the residual stream (written by this lane only)
    // ---- 2. merge, split, fragments of row 0 into LDS ----
    {
        h8 lo;
        const h8 hi = __builtin_bit_cast(h8, pcm::part_merge(pp.part, pl, (pcm::h8*)&lo));
        if (tid < nv) {                                   // chunk c = tid: k-step c >> 2, group c & 3
            *(h8*)(smem + tid * 16) = hi;
            *(h8*)(smem + (size_t)K * 2 + tid * 16) = lo;
        }
    }
    lds_barrier();
    // ---- 3. K loop ----
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const h8* xh = (const h8*)smem + g;
    const h8* xl = (const h8*)(smem + (size_t)K * 2) + g;
    auto consume = [&](const h8 (&w)[kPU], int kb) {
        h8 a[kPU], b[kPU];
#pragma unroll
        for (int u = 0; u < kPU; ++u) {
            const int k = kb + u < ks1 ? kb + u : ks1 - 1;
            a[u] = xh[(k < 0 ? 0 : k) * 4];
            b[u] = xl[(k < 0 ? 0 : k) * 4];
        }
#pragma unroll
        for (int u = 0; u < kPU; ++u) {
            if (kb + u >= ks1) continue;                 // (wave-uniform)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[u], a[u], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[u], b[u], acc, 0, 0, 0);
        }
    };
    {
        int b = 0;
        bool done = false;
        while (b + 2 < nb) {
            consume(wA, ks0 + b * kPU);
            issue(wA, ks0 + (b + 2) * kPU);
            if (!(b + 3 < nb)) {
                consume(wB, ks0 + (b + 1) * kPU);
                consume(wA, ks0 + (b + 2) * kPU);
                done = true;
                break;
            }
            consume(wB, ks0 + (b + 1) * kPU);
            issue(wB, ks0 + (b + 3) * kPU);
            b += 2;
        }
        if (!done) {
            if (b < nb) consume(wA, ks0 + b * kPU);
            if (b + 1 < nb) consume(wB, ks0 + (b + 1) * kPU);
        }
    }
    // ---- 4. the eight waves' shares in wave order, residual, store (row 0 = lanes m == 0) ----
    *(f4*)red[wave][lane] = acc;
    lds_barrier();
    if (wave == 0 && bx < p.ntiles) {
        f4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            const f4 x = *(const f4*)red[w][lane];
            v[0] += x[0]; v[1] += x[1]; v[2] += x[2]; v[3] += x[3];
        }
        if (m == 0) {
            v[0] += yold[0]; v[1] += yold[1]; v[2] += yold[2]; v[3] += yold[3];
            *(f4*)(p.y + tile * 16 + g * 4) = v;
        }
    }
}

}  // namespace

// y[0][n] += sum_k merged[k] W[n][k]: include/promptcache_hip.h
PC_EXPORT int pc_gemm_part(const void* wf, const float* part_o, const float* part_ml, int32_t nsplit, int32_t H, int32_t D, int32_t N,
                           float* y, void* stream) {
    const int K = H * D;
    PC_REQUIRE(wf && part_o && part_ml && y, PC_ERR_ARG, "pc_gemm_part: null pointer");
    PC_REQUIRE(nsplit >= 2 && nsplit <= pcm::kPartNS && H > 0 && D > 0 && D % 8 == 0 && K % 32 == 0 && K <= 4096 && N > 0 && N % 16 == 0 &&
               (((uintptr_t)part_o | (uintptr_t)part_ml | (uintptr_t)y) & 15) == 0, PC_ERR_ARG,
               "pc_gemm_part: need 2..8 partials per head, K = H * D <= 4096 (K %% 32 == 0), N %% 16 == 0, 16-byte aligned pointers");
    PartGemmParams pp;
    memset(&pp, 0, sizeof(pp));
    pp.g.wf = (const _Float16*)wf; pp.g.y = y; pp.g.ldy = N; pp.g.M = 1; pp.g.ntiles = N / 16; pp.g.KS = K / 32; pp.g.kslices = 1;
    pp.part.part_o = part_o; pp.part.part_ml = part_ml; pp.part.nsplit = nsplit; pp.part.D = D; pp.part.q_len = 1;
    hipLaunchKernelGGL(gemm_part_kernel, dim3(N / 16), dim3(kThreads), (size_t)K * 4, (hipStream_t)stream, pp);
    return pc_check_launch("gemm_part_kernel");
}
