// Many-row projections (schema encode, no-cache prefill, long questions): the MFMA-bound regime of the path.
//
// Replaces  q_proj/k_proj/v_proj   promptcache/model/llama2.py:345-347   (one fused [q|k|v] GEMM)
//           o_proj + residual       promptcache/model/llama2.py:405, :638
//           gate/up + SiLU*up       promptcache/model/llama2.py:242
//           down_proj + residual    promptcache/model/llama2.py:242, :644
//           lm_head                 promptcache/model/llama2.py:1050
//           (Falcon / MPT: query_key_value, dense, dense_h_to_4h + GELU, dense_4h_to_h; falcon.py:393-405, :726-731)
// inside SchemaCache._process (cache_engine.py:243-248): the module-KV precompute.
//
// Y[m][n] = sum_k (Xhi[m][k] + Xlo[m][k]) * W[n][k]       X: split-precision fp16 planes [M][K] (hi = fp16(x),
//                                                          lo = fp16(x - hi); lo optional), W: nn.Linear [N][K] fp16
// Why two planes: an fp16 activation costs 2^-12 per projection input, and through 32 layers that alone moves
// 7b-shape logits by 2-3e-2 against the reference's fp32 CPU path.  Round 1 ran [hi; lo] stacked
// along the rows of a vendor GEMM and added the halves in separate fp32 passes.  Here both planes meet the SAME weight
// fragment inside one tile pass: a weight fragment is read from LDS once and feeds the hi and the lo MFMA, the two
// partial products meet in the fp32 accumulator, and the epilogue (residual add, SiLU*up, GELU, fp32 store) runs on the
// accumulator tile -- no [2T][N] fp32 intermediate ever exists.
//
// Tile: workgroup = 8 waves (2 per SIMD), block tile 128 (M) x BN (N) x 64 (K), BN = 256 (waves 2 x 4, wave tile
// 64 x 64) or 128 (waves 4 x 2, wave tile 32 x 64; for launches whose 256-wide grid would leave CUs idle).
// mfma_f32_32x32x16_f16: per 16-deep k-slab a wave reads 2*MT activation fragments and 2 weight fragments (16 B per
// lane each, ds_read_b128) for 4*MT MFMAs.
// Staging: global_load_lds (16 B per lane, 1 KiB per wave-instruction = 8 rows x 128 B of a tile) straight into a
// double-buffered LDS image, one barrier per K-step: the loads of K-step t+1 are in flight while step t is multiplied.
// The LDS image keeps 128-byte rows; the 16-byte chunk c of row r sits at chunk position c ^ ((r >> 1) & 7), which
// makes every ds_read_b128 lane group of a fragment read hit 16 distinct 16-byte bank slots (row-major 128-B rows
// would be 8-way conflicted).  LDS-DMA writes lane-linearly, so the permutation is applied to the per-lane SOURCE
// address and again on the read (guide: rule 21, both sides or neither).
// Which global weight row lands in which LDS row is free as well (per-lane source address): the SiLU launch puts 32
// gate rows and the 32 matching up rows side by side in one wave tile, so gate_j and up_j meet in the same lane.
// Epilogue: the accumulator tile goes through the (now idle) LDS once so that every global store is 16 B per lane on
// full rows.
// Grid: 1-D, XCD-aware: block id b runs on XCD b % 8; an XCD owns the weight panels xcd, xcd + 8, ... and walks all
// M-blocks of a panel before the next one, so a panel is fetched from HBM once and served from that XCD's L2.
// Roofline: MFMA.  flops per launch = 2 * (TWO ? 2 : 1) * M * N * K against the 2.5 PFLOP/s dense fp16 peak.
#include <hip/hip_fp16.h>
#include <string.h>

#include <type_traits>

#include "pc_common.h"
#ifdef PC_DEV_SWEEPS
#include "pc_dev.h"
#endif

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BK = 64;
constexpr int kXT = BM * BK * 2;                 // bytes of one activation plane tile (16 KiB)
enum { EPI_STORE = 0, EPI_ADD = 1, EPI_SILU = 2, EPI_ROPE = 3, EPI_GELU = 4 };

struct DenseParams {
    const _Float16* xh; const _Float16* xl; int64_t ldx;   // [M][K] planes (xl may be null), row stride in halfs
    const _Float16* w; int64_t ldw;                        // [N][K]
    const float* wscale;                                   // optional per-output-row scale (int8 codes held in fp16)
    // LLM.int8 (pc_int8.hip): xh holds activation CODES, xscale[m] = SCA[m] / 127; corr[m][n] (global weight-row index n) is
    // added before the epilogue's nonlinearity when *corr_has != 0
    const float* xscale; const float* corr; int64_t ldc; const int32_t* corr_has;
    // LO8 (round 5): the lo activation plane as INT8 codes [M][K] with one scale per row (lo ~ code * xl8_scale[m]) against an int8
    // image of the weights [N][K] with one scale per row (W ~ code * w8_scale[n]): the residual product runs on
    // v_mfma_i32_32x32x32_i8 -- twice the fp16 MFMA rate, exact integer sums -- and joins the fp32 accumulator in the epilogue
    const signed char* xl8; const float* xl8_scale; int64_t ldx8;
    const signed char* w8; const float* w8_scale; int64_t ldw8;
    const _Float16* zeros;                                 // >= 16 B of zeros: source of activation chunks past K
    float* y; int64_t ldy;                                 // EPI_STORE / EPI_ADD
    _Float16* oh; _Float16* ol; int64_t ldo;               // EPI_SILU / EPI_GELU output planes
    int32_t M, N, K, mb, nb, nfeat;                        // nfeat: SiLU features (= N / 2)
    // split-K (EPI_STORE instantiation only): grid.y slices of `kper` K-steps, slice z stores its partial sums to
    // y + z * slab_stride; dense_reduce_kernel then adds the slabs in fixed order (launch_dense_splitk)
    int32_t kper; int64_t slab_stride;
    // EPI_ROPE (the fused q|k|v projection at head_dim 128, llama2.py:345-364): rotation table [M][64] (cos, sin), rotated q as
    // hi / lo planes [M][q_ts], rotated k and plain v into the arena planes (+ their residual planes) behind each batch row's past
    const float2* cs; _Float16* q_hi; _Float16* q_lo; int64_t q_ts;
    _Float16* k_arena; _Float16* v_arena; int64_t a_bs, a_hs;
    _Float16* k_lo; _Float16* v_lo; int64_t lo_bs, lo_hs; int32_t lo_row0;
    int32_t H, Hkv, q_len, past_len; const int32_t* past_lens;
};

__device__ __forceinline__ void glds16(const _Float16* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

typedef int i4v __attribute__((ext_vector_type(4)));
typedef int i16v __attribute__((ext_vector_type(16)));

template <int WM, int EPI, bool TWO, bool PP, bool LO8 = false>
__global__ __launch_bounds__(512) void gemm_dense_kernel(const DenseParams p) {
    static_assert(!(LO8 && (TWO || PP)), "LO8 replaces the fp16 lo plane; lockstep main loop only");
    constexpr int WN = 8 / WM, MT = BM / (32 * WM), NT = 2, BN = 64 * WN;
    constexpr int kWT = BN * BK * 2;             // bytes of the weight tile
    // stage image: Xhi | Xlo | W   (LO8: Xhi | Xlo8 (64-byte rows) | W | W8 (64-byte rows))
    constexpr int oW = LO8 ? kXT + kXT / 2 : 2 * kXT;
    constexpr int oW8 = oW + kWT;
    constexpr int kStage = LO8 ? oW8 + kWT / 2 : 2 * kXT + kWT;
    constexpr int NXI = 2, NWI = BN / 64;        // staging instructions per wave: 2 per activation plane, BN/64 for W
    constexpr int NW8 = BN / 128;                // ... and for the int8 weight tile (16 rows x 64 B per instruction); Xlo8: one
    __shared__ __attribute__((aligned(16))) char lds[2 * kStage];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    // block -> (M-block, weight panel), XCD-aware
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3;
    const int nloc = slot / p.mb, mi = slot - nloc * p.mb, ni = nloc * 8 + xcd;
    if (ni >= p.nb) return;
    const int m0 = mi * BM;
    const int K = p.K, nkt = (K + BK - 1) / BK;          // K-steps of the whole product
    const int t0 = blockIdx.y * p.kper;                  // this slice's first K-step (kper = nkt without split-K)
    const int nk = (p.kper < nkt - t0) ? p.kper : nkt - t0;

    // global weight row of LDS weight-tile row r; -1 = past the end (clamped source, never stored)
    auto wrow = [&](int r) -> int {
        if (EPI == EPI_SILU) {
            const int f = (ni * WN + (r >> 6)) * 32 + (r & 31);
            return f < p.nfeat ? ((r >> 5) & 1) * p.nfeat + f : -1;
        }
        if (EPI == EPI_ROPE) {
            // a wave tile (64 columns) = 32 features d of one head's first half next to their rotary partners d + 64, so
            // x[d] and x[d + 64] meet in the same lane and accumulator index of the two 32-wide MFMA blocks
            const int wt = r >> 6, c = r & 63;
            const int hd = ni * (BN / 128) + (wt >> 1);
            const int d = (c < 32) ? (wt & 1) * 32 + c : 64 + (wt & 1) * 32 + (c - 32);
            const int n = hd * 128 + d;
            return n < p.N ? n : -1;
        }
        const int n = ni * BN + r;
        return n < p.N ? n : -1;
    };

    // ---- staging sources: per lane one 16-byte chunk of one row per instruction ----
    const int srow = lane >> 3, scp = lane & 7;          // row inside the 8-row group, chunk POSITION in the LDS row
    const _Float16* gx[NXI];
    const _Float16* gw[NWI];
    int cx[NXI], cw[NWI];                                // source chunk index (for the K tail)
    const int64_t lo_delta = TWO ? (p.xl - p.xh) : 0;
#pragma unroll
    for (int j = 0; j < NXI; ++j) {
        const int r = wave * 16 + j * 8 + srow;
        const int gm = (m0 + r < p.M) ? m0 + r : p.M - 1;
        cx[j] = scp ^ ((r >> 1) & 7);
        gx[j] = p.xh + (int64_t)gm * p.ldx + cx[j] * 8;
    }
#pragma unroll
    for (int j = 0; j < NWI; ++j) {
        const int r = wave * (BN / 8) + j * 8 + srow;
        int gr = wrow(r);
        gr = gr < 0 ? 0 : gr;
        cw[j] = scp ^ ((r >> 1) & 7);
        gw[j] = p.w + (int64_t)gr * p.ldw + cw[j] * 8;
    }
    // LO8: one instruction of 16 rows x 64 B per wave for the activation codes, NW8 for the weight codes; lane -> (row, chunk
    // POSITION); position c of row r holds source chunk c ^ ((r >> 2) & 3): the 16 lanes of every ds_read_b128 group then fall
    // into 16 distinct 16-byte bank slots (64-byte rows: slot = 4 (r & 3) + position)
    [[maybe_unused]] const signed char* gx8 = nullptr;
    [[maybe_unused]] const signed char* gw8[NW8 > 0 ? NW8 : 1];
    if constexpr (LO8) {
        const int r = wave * 16 + (lane >> 2);
        const int gm = (m0 + r < p.M) ? m0 + r : p.M - 1;
        gx8 = p.xl8 + (int64_t)gm * p.ldx8 + (((lane & 3) ^ ((r >> 2) & 3)) << 4);
#pragma unroll
        for (int j = 0; j < NW8; ++j) {
            const int rw = wave * (BN / 8) + j * 16 + (lane >> 2);
            int gr = wrow(rw);
            gr = gr < 0 ? 0 : gr;
            gw8[j] = p.w8 + (int64_t)gr * p.ldw8 + (((lane & 3) ^ ((rw >> 2) & 3)) << 4);
        }
    }
    const bool ktail = (K % BK) != 0;
    auto stage = [&](int tl, char* buf) {
        const int t = t0 + tl;
        const bool tail = ktail && t == nkt - 1;
        const int64_t ko = (int64_t)t * BK;
        const int kchunks = tail ? (K - t * BK) / 8 : 8;          // valid 16-byte chunks in this K-step
#pragma unroll
        for (int j = 0; j < NXI; ++j) {
            const bool ok = !tail || cx[j] < kchunks;
            const _Float16* sh = ok ? gx[j] + ko : p.zeros;
            glds16(sh, buf + (wave * 16 + j * 8) * 128);
            if (TWO) glds16(ok ? gx[j] + lo_delta + ko : p.zeros, buf + kXT + (wave * 16 + j * 8) * 128);
        }
#pragma unroll
        for (int j = 0; j < NWI; ++j) {
            // chunks past K pair with zero activations: any finite weight bytes do (the row's first chunk)
            const _Float16* sw = (!tail || cw[j] < kchunks) ? gw[j] + ko : gw[j] - cw[j] * 8;
            glds16(sw, buf + oW + (wave * (BN / 8) + j * 8) * 128);
        }
        if constexpr (LO8) {                           // (K % 64 == 0 with LO8: no tail)
            glds16((const _Float16*)(gx8 + ko), buf + kXT + wave * 1024);
#pragma unroll
            for (int j = 0; j < NW8; ++j) glds16((const _Float16*)(gw8[j] + ko), buf + oW8 + (wave * (BN / 8) + j * 16) * 64);
        }
    };

    // ---- fragment addressing: lane = (row fr of the 32-row subtile, k-half fh) ----
    const int fr = lane & 31, fh = lane >> 5;
    int foff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) foff[ks] = fr * 128 + (((2 * ks + fh) ^ ((fr >> 1) & 7)) << 4);
    const int a_base = (wm * 32 * MT) * 128;              // + 32*i rows, + kXT for the lo plane
    const int b_base = oW + (wn * 64) * 128;              // + 32*j rows
    // LO8: a lane's 16 codes of a 32-deep block kb are chunk 2 kb + fh of its row (A and B pair slot by slot, so which sixteen k
    // of the block a lane holds is free as long as both sides agree)
    [[maybe_unused]] int foff8[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) foff8[kb] = fr * 64 + (((2 * kb + fh) ^ ((fr >> 2) & 3)) << 4);
    [[maybe_unused]] const int a8_base = kXT + (wm * 32 * MT) * 64, b8_base = oW8 + (wn * 64) * 64;
    [[maybe_unused]] i16v iacc[LO8 ? MT : 1][LO8 ? NT : 1];
    if constexpr (LO8) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) iacc[i][j][r] = 0;
    }

    f16v acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- main loop: one barrier per K-step, fragment reads one 16-deep slab ahead of the MFMAs ----
    // Two fragment register sets alternate over the 4 slabs of a K-step.  The reads of slab s+1 are issued in front of
    // the MFMAs of slab s, so their LDS latency passes under 4*MT MFMAs of 32 cycles each; the last slab of K-step t is
    // multiplied AFTER the barrier, next to the first reads of K-step t+1 and the LDS-DMA issue of K-step t+2, so the
    // matrix pipe has work while the workgroup re-synchronises.
    struct Frags { h8 ah[MT], al[TWO ? MT : 1], bw[NT]; };
    auto load_frags = [&](Frags& f, const char* buf, int ks) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            f.ah[i] = *(const h8*)(buf + a_base + i * 32 * 128 + foff[ks]);
            if (TWO) f.al[i] = *(const h8*)(buf + kXT + a_base + i * 32 * 128 + foff[ks]);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) f.bw[j] = *(const h8*)(buf + b_base + j * 32 * 128 + foff[ks]);
    };
    auto mfmas = [&](const Frags& f) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bw[j], acc[i][j], 0, 0, 0);
        if (TWO) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[i], f.bw[j], acc[i][j], 0, 0, 0);
        }
    };
    // Pin the issue order of one (reads of the next slab, MFMAs of this slab) pair: a read in front of every MFMA until
    // the reads run out.  Left alone, hipcc sinks the ds_reads next to their first use and waits lgkmcnt(0) right
    // behind them (seen in the ISA: four exposed LDS round trips per K-step).
    auto interleave = [&]() {
        // one MFMA, then two reads, until the reads run out: every read of the next slab is in flight at least three
        // MFMAs (~100 cycles) before the wait in front of that slab's first MFMA
        constexpr int R = MT * (TWO ? 2 : 1) + NT, MF = MT * NT * (TWO ? 2 : 1);
#pragma unroll
        for (int k = 0; k < MF; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          // one MFMA
            if (2 * k + 1 < R) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);       // two DS reads
            else if (2 * k < R) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // (an odd one left)
        }
    };
    // The slab multiplied right behind the K-step barrier also carries the LDS-DMA issue of K-step t + 2 (its buffer was just
    // freed).  Rounds 2-4 issued the 8 (LO8: 11) DMA instructions in one burst in front of that slab's MFMAs: both waves of a SIMD
    // spend ~60-100 issue cycles per instruction there while the matrix pipe, drained by the barrier, has nothing to do (seen in
    // the ISA: barrier, 8 x global_load_lds, then the first MFMA).  Round 5 spreads them BETWEEN that slab's MFMAs: one MFMA, two
    // fragment reads (until they run out), the slab's share of the DMA instructions, next MFMA -- for launches with two fp16 planes
    // (-0.6 ... -2.6 % per layer at 800 ... 6000 rows); one-plane and int8-residual launches measured 1-5 % SLOWER that way and keep
    // the burst (profiles/r05_variants.txt).  -DPC_DENSE_SPREAD_DMA=0: the burst everywhere.
#ifndef PC_DENSE_SPREAD_DMA
#define PC_DENSE_SPREAD_DMA 1
#endif
    // one MFMA / one fragment read / a range of the DMA instructions of a K-step, by index (compile-time after unrolling)
    auto mfma_one = [&](const Frags& f, int idx) {
        const int pl = idx / (MT * NT), i = (idx / NT) % MT, j = idx % NT;
        if (TWO && pl == 1) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[i], f.bw[j], acc[i][j], 0, 0, 0);
        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bw[j], acc[i][j], 0, 0, 0);
    };
    auto read_one = [&](Frags& f, const char* buf, int ks, int idx) {
        constexpr int NA = MT * (TWO ? 2 : 1);
        if (idx < NA) {
            const int i = TWO ? idx >> 1 : idx;
            if (TWO && (idx & 1)) f.al[i] = *(const h8*)(buf + kXT + a_base + i * 32 * 128 + foff[ks]);
            else f.ah[i] = *(const h8*)(buf + a_base + i * 32 * 128 + foff[ks]);
        } else {
            const int j = idx - NA;
            f.bw[j] = *(const h8*)(buf + b_base + j * 32 * 128 + foff[ks]);
        }
    };
    auto stage_range = [&](int tl, char* buf, int q0, int q1) {
        const int t = t0 + tl;
        const bool tail = ktail && t == nkt - 1;
        const int64_t ko = (int64_t)t * BK;
        const int kchunks = tail ? (K - t * BK) / 8 : 8;
        int q = 0;
#pragma unroll
        for (int j = 0; j < NXI; ++j) {
            const bool ok = !tail || cx[j] < kchunks;
            if (q >= q0 && q < q1) glds16(ok ? gx[j] + ko : p.zeros, buf + (wave * 16 + j * 8) * 128);
            ++q;
            if (TWO) {
                if (q >= q0 && q < q1) glds16(ok ? gx[j] + lo_delta + ko : p.zeros, buf + kXT + (wave * 16 + j * 8) * 128);
                ++q;
            }
        }
#pragma unroll
        for (int j = 0; j < NWI; ++j) {
            if (q >= q0 && q < q1)
                glds16((!tail || cw[j] < kchunks) ? gw[j] + ko : gw[j] - cw[j] * 8, buf + oW + (wave * (BN / 8) + j * 8) * 128);
            ++q;
        }
        if constexpr (LO8) {
            if (q >= q0 && q < q1) glds16((const _Float16*)(gx8 + ko), buf + kXT + wave * 1024);
            ++q;
#pragma unroll
            for (int j = 0; j < NW8; ++j) {
                if (q >= q0 && q < q1) glds16((const _Float16*)(gw8[j] + ko), buf + oW8 + (wave * (BN / 8) + j * 16) * 64);
                ++q;
            }
        }
    };
    if constexpr (PP) {
        // ---- ping-pong main loop (round 5; the guide's two-waves-per-SIMD regime).  MEASURED SLOWER than the lockstep loop below
        // on this kernel (0.87-0.90 x with both planes, profiles/r05_dense_pp_ab.txt) in all three forms tried -- fragment reads
        // waited for inside their load segment; read one phase ahead with hipcc's waits (lgkmcnt(0) every other phase: its
        // scoreboard merges conservatively at the group branches); read ahead as raw ds_read_b128 with explicit counted waits
        // (this form) -- kept for A/B behind -DPC_DEV_SWEEPS.  Eight barriers per K-step and one wave per SIMD feeding the
        // matrix pipe at a time lose to two waves per SIMD interleaving their MFMAs with one barrier per K-step; what is left
        // between the lockstep loop (1.09-1.15 PFLOP/s executed on random data) and the ~1.3 the guide's 8-phase template
        // reaches at 4096^3 is not in the wave pairing. ----
        // The loop above keeps the two waves of a SIMD in lockstep: both issue MFMAs at once (54 % of wave time were issue stalls
        // behind the shared pipe, profiles/r02_pmc_gemm_dense.txt) and both sit out the barrier and the first fragment reads of
        // the next K-step together (MFMA pipes busy 63 % of the kernel's cycles).  Here the workgroup's eight waves are two GROUPS
        // of four (one wave of each group per SIMD) that run the same program ONE BARRIER APART: a K-step is four phases (one
        // 16-deep slab each), a phase is a LOAD segment (the slab's 6 fragment reads, waited for) and a COMPUTE segment (its 8
        // MFMAs = 256 matrix-pipe cycles), every segment ends in a raw s_barrier -- so while group 0 is in its compute segment
        // group 1 is in its load segment and vice versa: the SIMD's matrix pipe always has exactly one wave feeding it, and the
        // LDS reads, the waits and the barrier latency of one wave pass under the MFMAs of the other.  One fragment register
        // set instead of two.
        //   barrier n separates interval n from n + 1.  Group 0: load (t, s) in interval 8t + 2s, compute in 8t + 2s + 1; group 1
        //   one interval later.  A load segment only ISSUES the fragment reads of the NEXT slab (two register sets); they are
        //   consumed one phase later, so the segment is a few dozen issue cycles and the LDS latency never sits in front of a
        //   barrier (first version: reads waited for inside their load segment -- the load segment then took longer than the
        //   partner's 256-cycle compute segment and the loop ran at ITS pace: 0.81-0.93 x the lockstep loop).
        //   Slab 0 of K-step t + 1 is read in load (t, 3): group 0 in interval 8t + 6, so K-step t + 1 must have landed, for all
        //   waves, by barrier 8t + 6: the LDS-DMA goes out in two pieces between MFMAs -- group 0 in compute (t, 0), (t, 1), waited
        //   for at the end of compute (t, 2); group 1 in load (t, 0) and compute (t, 0), waited for at the end of load (t, 2) -- at
        //   least two intervals of flight for the last piece.  The buffer it lands in held K-step t - 1, whose last reads (group
        //   1, issued in its load (t - 1, 2)) are complete at the start of interval 8t: every issue point above lies behind
        //   barrier 8t + 1.
        // Same MFMA order per accumulator as the loop above: bit-identical results.
        constexpr int NI = NXI * (TWO ? 2 : 1) + NWI;          // LDS-DMA instructions per wave and K-step
        auto stage_piece = [&](int tl, char* buf, int piece) {
            const int t = t0 + tl;
            const bool tail = ktail && t == nkt - 1;
            const int64_t ko = (int64_t)t * BK;
            const int kchunks = tail ? (K - t * BK) / 8 : 8;
            const int q0 = piece * NI / 2, q1 = (piece + 1) * NI / 2;
            int q = 0;
#pragma unroll
            for (int j = 0; j < NXI; ++j) {
                const bool ok = !tail || cx[j] < kchunks;
                if (q >= q0 && q < q1) glds16(ok ? gx[j] + ko : p.zeros, buf + (wave * 16 + j * 8) * 128);
                ++q;
                if (TWO) {
                    if (q >= q0 && q < q1) glds16(ok ? gx[j] + lo_delta + ko : p.zeros, buf + kXT + (wave * 16 + j * 8) * 128);
                    ++q;
                }
            }
#pragma unroll
            for (int j = 0; j < NWI; ++j) {
                if (q >= q0 && q < q1)
                    glds16((!tail || cw[j] < kchunks) ? gw[j] + ko : gw[j] - cw[j] * 8, buf + 2 * kXT + (wave * (BN / 8) + j * 8) * 128);
                ++q;
            }
        };
        const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
        Frags fr2[2];                                           // slab s lives in set s & 1: read one phase ahead of its MFMAs
        // The fragment reads of the ping-pong loop are raw ds_read_b128 statements hipcc does not count: a read issued in load
        // segment s is consumed in compute segment s + 1 behind an explicit `s_waitcnt lgkmcnt(R)` (R = reads per slab: only
        // the NEWER slab's reads may still be in flight).  Left to hipcc, the uniform `if (group)` branches between a read and
        // its use make the waitcnt pass merge its scoreboards conservatively: lgkmcnt(0) in every other phase, i.e. the LDS
        // latency of the just-issued reads in front of the MFMAs -- what the read-ahead is there to hide.  The loop holds no
        // other LDS or scalar-memory instruction (checked in the ISA: lgkmcnt counts in order only without SMEM in flight).
        constexpr int NR = MT * (TWO ? 2 : 1) + NT;
        const uint32_t lds32 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
        auto load_frags_raw = [&](Frags& f, uint32_t buf_off, int ks) {
            const uint32_t aa = lds32 + buf_off + a_base + foff[ks], ab = lds32 + buf_off + b_base + foff[ks];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.ah[i]) : "v"(aa), "i"(i * 32 * 128));
                if (TWO) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.al[i]) : "v"(aa), "i"(kXT + i * 32 * 128));
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.bw[j]) : "v"(ab), "i"(j * 32 * 128));
        };
        stage(0, lds);
        stage(nk > 1 ? 1 : 0, lds + kStage);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // both K-steps (the prologue is not where the time goes)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        load_frags_raw(fr2[0], 0, 0);
        if (grp) {                                              // group 1 runs one interval behind
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        auto half_a = [&](const Frags& f) {                     // first half of a slab's MFMAs (hi plane; without a lo plane: the first row block)
#pragma unroll
            for (int i = 0; i < (TWO ? MT : (MT + 1) / 2); ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bw[j], acc[i][j], 0, 0, 0);
        };
        auto half_b = [&](const Frags& f) {
            if (TWO) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[i], f.bw[j], acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = (MT + 1) / 2; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[i], f.bw[j], acc[i][j], 0, 0, 0);
            }
        };
        for (int t = 0; t < nk; ++t) {
            char* cur = lds + (t & 1) * kStage;
            char* nxt = lds + ((t + 1) & 1) * kStage;
            const int u1 = t + 1 < nk ? t + 1 : nk - 1;         // (clamped: past the end the last K-step goes into the idle buffer again)
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                // ---- load segment: only ISSUES the next slab's fragment reads (slab 0 of K-step t + 1 behind slab 3: its DMA was
                // waited for two barriers ago); they complete under the partner group's MFMAs and this group's own ----
                if (sl < 3) load_frags_raw(fr2[(sl + 1) & 1], (t & 1) * kStage, sl + 1);
                else load_frags_raw(fr2[0], ((t + 1) & 1) * kStage, 0);     // (past the last K-step: the clamped re-load of it, never multiplied)
                if (grp) {
                    if (sl == 0) stage_piece(u1, nxt, 0);
                    if (sl == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- compute segment: the slab read one phase ago; the newer slab's NR reads stay in flight ----
                static_assert(NR == 6 || NR == 4 || NR == 3, "reads per slab");
                if constexpr (NR == 6) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
                else if constexpr (NR == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
                half_a(fr2[sl & 1]);
                __builtin_amdgcn_sched_barrier(0);
                if (!grp) {
                    if (sl < 2) stage_piece(u1, nxt, sl);
                } else {
                    if (sl == 0) stage_piece(u1, nxt, 1);
                }
                __builtin_amdgcn_sched_barrier(0);
                half_b(fr2[sl & 1]);
                __builtin_amdgcn_s_setprio(0);
                if (sl == 2 && !grp) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the read-ahead behind the last slab: landed before its registers die)
        if (!grp) {                                             // group 0 waits for group 1's last compute segment
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if constexpr (LO8) {
        // ---- lockstep main loop with the lo plane on the int8 MFMA (round 5) ----
        // Per K-step: 16 fp16 MFMAs for the hi plane (as without a lo plane) + 8 v_mfma_i32_32x32x32_i8 for the residual plane
        // (two 32-deep blocks x MT x NT) instead of 16 more fp16 ones: 24 matrix-pipe slots instead of 32.  One int8 fragment set:
        // block kb is read one slab ahead of its MFMAs like the fp16 fragments (kb 0 next to slab 0, kb 1 next to slab 2).
        struct F8 { i4v a[MT], b[NT]; };
        F8 f8;
        auto load8 = [&](const char* buf, int kb) {
#pragma unroll
            for (int i = 0; i < MT; ++i) f8.a[i] = *(const i4v*)(buf + a8_base + i * 32 * 64 + foff8[kb]);
#pragma unroll
            for (int j = 0; j < NT; ++j) f8.b[j] = *(const i4v*)(buf + b8_base + j * 32 * 64 + foff8[kb]);
        };
        auto imfmas = [&]() {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) iacc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f8.a[i], f8.b[j], iacc[i][j], 0, 0, 0);
        };
        // (reads of the next slab, MFMAs of this one): one MFMA, then two reads, until the reads run out
        auto pin = [&](auto r_, auto mf_) {
            constexpr int R = decltype(r_)::value, MF = decltype(mf_)::value;
#pragma unroll
            for (int k = 0; k < MF; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (2 * k + 1 < R) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                else if (2 * k < R) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        };
        using RH = std::integral_constant<int, MT + NT>;             // reads of one fp16 slab
        using RB = std::integral_constant<int, 2 * (MT + NT)>;       // ... plus one int8 block
        using MH = std::integral_constant<int, MT * NT>;             // MFMAs of one fp16 slab
        using MB = std::integral_constant<int, 2 * MT * NT>;         // ... plus one int8 block
        Frags fa, fb;
        stage(0, lds);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (nk > 1) stage(1, lds + kStage);
        load_frags(fa, lds, 0);
        load8(lds, 0);
        for (int t = 0; t + 1 < nk; ++t) {
            const char* cur = lds + (t & 1) * kStage;
            load_frags(fb, cur, 1);
            mfmas(fa);
            imfmas();
            pin(RH{}, MB{});
            load_frags(fa, cur, 2);
            load8(cur, 1);
            mfmas(fb);
            pin(RB{}, MH{});
            load_frags(fb, cur, 3);
            mfmas(fa);
            imfmas();
            pin(RH{}, MB{});
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const int tn = t + 2 < nk ? t + 2 : nk - 1;
            // (the DMA issue spread between this slab's MFMAs -- see the fp16 loop below -- measured SLOWER here: 4177 vs 4136 us per
            // layer at 6000 rows, profiles/r05_variants.txt)
            stage(tn, lds + (t & 1) * kStage);
            load_frags(fa, lds + ((t + 1) & 1) * kStage, 0);
            load8(lds + ((t + 1) & 1) * kStage, 0);
            mfmas(fb);
            pin(RB{}, MH{});
        }
        {
            const char* cur = lds + ((nk - 1) & 1) * kStage;
            load_frags(fb, cur, 1);
            mfmas(fa);
            imfmas();
            pin(RH{}, MB{});
            load_frags(fa, cur, 2);
            load8(cur, 1);
            mfmas(fb);
            pin(RB{}, MH{});
            load_frags(fb, cur, 3);
            mfmas(fa);
            imfmas();
            pin(RH{}, MB{});
            mfmas(fb);
        }
        // the residual product joins the fp32 accumulator: y += int_sum * xl8_scale[row] * w8_scale[col]
        {
            float wsc[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                int gr = wrow(wn * 64 + 32 * j + fr);
                wsc[j] = p.w8_scale[gr < 0 ? 0 : gr];
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * 32 * MT + 32 * i + 8 * (r >> 2) + 4 * fh + (r & 3);
                    const float xs = p.xl8_scale[m < p.M ? m : p.M - 1];
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j][r] = __builtin_fmaf((float)iacc[i][j][r], xs * wsc[j], acc[i][j][r]);
                }
        }
    } else {
    Frags fa, fb;
    stage(0, lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (nk > 1) stage(1, lds + kStage);
    load_frags(fa, lds, 0);
    for (int t = 0; t + 1 < nk; ++t) {
        const char* cur = lds + (t & 1) * kStage;
        load_frags(fb, cur, 1);
        mfmas(fa);
        interleave();
        load_frags(fa, cur, 2);
        mfmas(fb);
        interleave();
        load_frags(fb, cur, 3);
        mfmas(fa);
        interleave();
        // K-step t+1 has landed (this wave's LDS-DMA drained, then everyone's via the barrier), and every wave's reads
        // of K-step t are complete (__syncthreads waits lgkmcnt(0)): its buffer may be overwritten.  The stage index
        // is clamped instead of branched on (past the end it re-loads the last K-step into the idle buffer): a branch
        // here splits the block and hipcc then waits for the fresh reads below before the MFMAs that do not need them.
        __builtin_amdgcn_sched_barrier(0);      // slab 2's MFMAs stay in front of the barrier: they cover slab 3's reads
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int tn = t + 2 < nk ? t + 2 : nk - 1;
        if constexpr (PC_DENSE_SPREAD_DMA && TWO) {
            constexpr int MF = MT * NT * (TWO ? 2 : 1), R = MT * (TWO ? 2 : 1) + NT, V = NXI * (TWO ? 2 : 1) + NWI;
            const char* nb = lds + ((t + 1) & 1) * kStage;
#pragma unroll
            for (int k = 0; k < MF; ++k) {
                mfma_one(fb, k);
                __builtin_amdgcn_sched_barrier(0);
                if (2 * k < R) read_one(fa, nb, 0, 2 * k);
                if (2 * k + 1 < R) read_one(fa, nb, 0, 2 * k + 1);
                stage_range(tn, lds + (t & 1) * kStage, k * V / MF, (k + 1) * V / MF);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            stage(tn, lds + (t & 1) * kStage);
            load_frags(fa, lds + ((t + 1) & 1) * kStage, 0);
            mfmas(fb);
            interleave();
        }
    }
    {
        const char* cur = lds + ((nk - 1) & 1) * kStage;
        load_frags(fb, cur, 1);
        mfmas(fa);
        interleave();
        load_frags(fa, cur, 2);
        mfmas(fb);
        interleave();
        load_frags(fb, cur, 3);
        mfmas(fa);
        interleave();
        mfmas(fb);
    }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the clamped re-load of the last K-step)

    // ---- epilogue: accumulators -> wave-private LDS tile [32*MT][64] fp32 -> full-row global stores ----
    __syncthreads();                                       // every wave is done with the staging buffers
    float* st = (float*)(lds + wave * (32 * MT * 64 * 4));
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                st[(32 * i + 8 * (r >> 2) + 4 * fh + (r & 3)) * 64 + 32 * j + fr] = acc[i][j][r];
    const int mrow0 = m0 + wm * 32 * MT;
    if (EPI == EPI_SILU) {
        // the wave's 64 columns are [32 gate features | the same 32 up features]: 8 lanes per row, 4 features per lane
        const int fbase = (ni * WN + wn) * 32 + (lane & 7) * 4;
#pragma unroll
        for (int it = 0; it < 4 * MT; ++it) {
            const int row = it * 8 + (lane >> 3), m = mrow0 + row;
            f4 g = *(const f4*)(st + row * 64 + (lane & 7) * 4);
            f4 u = *(const f4*)(st + row * 64 + 32 + (lane & 7) * 4);
            if (m < p.M && fbase < p.nfeat) {
                if (p.wscale) {
                    const f4 sg = *(const f4*)(p.wscale + fbase), su = *(const f4*)(p.wscale + p.nfeat + fbase);
                    g[0] *= sg[0]; g[1] *= sg[1]; g[2] *= sg[2]; g[3] *= sg[3];
                    u[0] *= su[0]; u[1] *= su[1]; u[2] *= su[2]; u[3] *= su[3];
                }
                if (p.xscale) {
                    const float xs = p.xscale[m];
                    g[0] *= xs; g[1] *= xs; g[2] *= xs; g[3] *= xs;
                    u[0] *= xs; u[1] *= xs; u[2] *= xs; u[3] *= xs;
                    if (*p.corr_has) {
                        const f4 cg = *(const f4*)(p.corr + (int64_t)m * p.ldc + fbase);
                        const f4 cu = *(const f4*)(p.corr + (int64_t)m * p.ldc + p.nfeat + fbase);
                        g[0] += cg[0]; g[1] += cg[1]; g[2] += cg[2]; g[3] += cg[3];
                        u[0] += cu[0]; u[1] += cu[1]; u[2] += cu[2]; u[3] += cu[3];
                    }
                }
                h4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float s = (g[e] / (1.0f + __expf(-g[e]))) * u[e];      // act_fn(gate) * up, llama2.py:242
                    _Float16 sh, sl;
                    pc_split(s, sh, sl);
                    hi[e] = sh; lo[e] = sl;
                }
                *(h4*)(p.oh + (int64_t)m * p.ldo + fbase) = hi;
                if (p.ol) *(h4*)(p.ol + (int64_t)m * p.ldo + fbase) = lo;
            }
        }
        return;
    }
    if (EPI == EPI_ROPE) {
        // RoPE at the supplied positions (llama2.py:200-213, :357-359) and the KV append (:361-364) on the accumulator tile:
        // 8 lanes per row, a lane holds features d0 .. d0+3 and d0+64 .. d0+67 of head `hd`
        const int hd = ni * (BN / 128) + (wn >> 1);
        const int d0 = (wn & 1) * 32 + (lane & 7) * 4;
        if (hd >= p.H + 2 * p.Hkv) return;
        const bool rot = hd < p.H + p.Hkv;
#pragma unroll
        for (int it = 0; it < 4 * MT; ++it) {
            const int row = it * 8 + (lane >> 3), m = mrow0 + row;
            const f4 x0 = *(const f4*)(st + row * 64 + (lane & 7) * 4);
            const f4 x1 = *(const f4*)(st + row * 64 + 32 + (lane & 7) * 4);
            if (m >= p.M) continue;
            f4 a = x0, c = x1;
            if (rot) {
                const f4 w01 = *(const f4*)(p.cs + (int64_t)m * 64 + d0);          // (cos, sin) of pairs d0, d0+1
                const f4 w23 = *(const f4*)(p.cs + (int64_t)m * 64 + d0 + 2);      // ... d0+2, d0+3
                const float cs_[4] = {w01[0], w01[2], w23[0], w23[2]}, sn_[4] = {w01[1], w01[3], w23[1], w23[3]};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // q*cos + rotate_half(q)*sin (llama2.py:208): the low half pairs with -high, the high half with +low
                    a[e] = x0[e] * cs_[e] - x1[e] * sn_[e];
                    c[e] = x1[e] * cs_[e] + x0[e] * sn_[e];
                }
            }
            h4 ah, al, ch, cl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                _Float16 t0, t1, t2, t3;
                pc_split(a[e], t0, t1);
                pc_split(c[e], t2, t3);
                ah[e] = t0; al[e] = t1; ch[e] = t2; cl[e] = t3;
            }
            _Float16* dh;
            _Float16* dl = nullptr;
            if (hd < p.H) {
                const int64_t off = (int64_t)m * p.q_ts + hd * 128 + d0;
                dh = p.q_hi + off;
                if (p.q_lo) dl = p.q_lo + off;
            } else {
                const int b = m / p.q_len, t = m - b * p.q_len;
                const int past = p.past_lens ? p.past_lens[b] : p.past_len;
                const bool is_k = hd < p.H + p.Hkv;
                const int kh = is_k ? hd - p.H : hd - p.H - p.Hkv;
                dh = (is_k ? p.k_arena : p.v_arena) + b * p.a_bs + kh * p.a_hs + (int64_t)(past + t) * 128 + d0;
                if (p.k_lo) dl = (is_k ? p.k_lo : p.v_lo) + b * p.lo_bs + kh * p.lo_hs + (int64_t)(past + t - p.lo_row0) * 128 + d0;
            }
            *(h4*)dh = ah;
            *(h4*)(dh + 64) = ch;
            if (dl) { *(h4*)dl = al; *(h4*)(dl + 64) = cl; }
        }
        return;
    }
    const int n = ni * BN + wn * 64 + (lane & 15) * 4;
#pragma unroll
    for (int it = 0; it < 8 * MT; ++it) {
        const int row = it * 4 + (lane >> 4), m = mrow0 + row;
        f4 v = *(const f4*)(st + row * 64 + (lane & 15) * 4);
        if (m < p.M && n < p.N) {
            if (p.wscale) {
                const f4 sv = *(const f4*)(p.wscale + n);
                v[0] *= sv[0]; v[1] *= sv[1]; v[2] *= sv[2]; v[3] *= sv[3];
            }
            if (p.xscale) {
                const float xs = p.xscale[m];
                v[0] *= xs; v[1] *= xs; v[2] *= xs; v[3] *= xs;
                if (*p.corr_has) {
                    const f4 cv = *(const f4*)(p.corr + (int64_t)m * p.ldc + n);
                    v[0] += cv[0]; v[1] += cv[1]; v[2] += cv[2]; v[3] += cv[3];
                }
            }
            if (EPI == EPI_GELU) {
                h4 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float s = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752f));   // nn.GELU(), falcon.py:726
                    _Float16 sh, sl;
                    pc_split(s, sh, sl);
                    hi[e] = sh; lo[e] = sl;
                }
                *(h4*)(p.oh + (int64_t)m * p.ldo + n) = hi;
                if (p.ol) *(h4*)(p.ol + (int64_t)m * p.ldo + n) = lo;
            } else {
                float* yp = p.y + (int64_t)blockIdx.y * p.slab_stride + (int64_t)m * p.ldy + n;
                if (EPI == EPI_ADD) {
                    const f4 old = *(const f4*)yp;
                    v[0] += old[0]; v[1] += old[1]; v[2] += old[2]; v[3] += old[3];
                }
                *(f4*)yp = v;
            }
        }
    }
}

// Split-K tail: y[m][n] (+)= sum_z slab[z][m][n], slabs added in slice order (deterministic), 16 B per lane.
template <bool ADD>
__global__ __launch_bounds__(256) void dense_reduce_kernel(const float* __restrict__ slab, int64_t slab_stride, int ks,
                                                           float* __restrict__ y, int64_t ldy, int M, int N) {
    const int n4 = N >> 2;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)M * n4) return;
    const int m = (int)(i / n4), n = (int)(i - (int64_t)m * n4) * 4;
    const float* sp = slab + (int64_t)m * N + n;
    f4 v = *(const f4*)sp;
    for (int z = 1; z < ks; ++z) {
        const f4 x = *(const f4*)(sp + z * slab_stride);
        v[0] += x[0]; v[1] += x[1]; v[2] += x[2]; v[3] += x[3];
    }
    float* yp = y + (int64_t)m * ldy + n;
    if (ADD) {
        const f4 old = *(const f4*)yp;
        v[0] += old[0]; v[1] += old[1]; v[2] += old[2]; v[3] += old[3];
    }
    *(f4*)yp = v;
}

// 16 bytes of zeros for activation chunks past K (device constant: no allocation, graph-capturable)
__device__ __attribute__((aligned(16))) _Float16 g_zero_chunk[8];

template <int WM, int EPI>
int launch_dense(DenseParams& p, hipStream_t s, int kslices = 1) {
    constexpr int BN = 64 * (8 / WM);
    const int units = (EPI == EPI_SILU) ? pc_ceil_div(p.nfeat, BN / 2) : pc_ceil_div(p.N, BN);
    p.mb = pc_ceil_div(p.M, BM);
    p.nb = units;
    const int nkt = pc_ceil_div(p.K, BK);
    p.kper = pc_ceil_div(nkt, kslices);
    const dim3 grid(8 * p.mb * pc_ceil_div(p.nb, 8), pc_ceil_div(nkt, p.kper)), block(512);
#ifdef PC_DEV_SWEEPS
    // The ping-pong main loop (PP = true) is a MEASURED NEGATIVE RESULT and only built for A/B (PC_BUILD_FLAGS=-DPC_DEV_SWEEPS,
    // then PC_DENSE_PP=1; tools/dense_pp_ab.py): bit-identical to the lockstep loop and 0.87-0.90 x its speed at the encode's
    // shapes with both planes, 0.66-0.83 x on one plane (profiles/r05_dense_pp_ab.txt).
    const char* e = getenv("PC_DENSE_PP");
    if (e && e[0] == '1') {
        if (p.xl) hipLaunchKernelGGL((gemm_dense_kernel<WM, EPI, true, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((gemm_dense_kernel<WM, EPI, false, true>), grid, block, 0, s, p);
        return pc_check_launch("gemm_dense_kernel");
    }
#endif
    if (p.xl8) {
#ifdef PC_DEV_SWEEPS
        if constexpr (EPI == EPI_GELU) {               // (no caller: the Falcon / MPT stacks keep the fp16 residual plane)
            pc_set_error("pc_gemm_dense_lo8: no GELU instantiation");
            return PC_ERR_ARG;
        } else {
            hipLaunchKernelGGL((gemm_dense_kernel<WM, EPI, false, false, true>), grid, block, 0, s, p);
        }
#else
        pc_set_error("pc_gemm_dense: the int8 residual plane (x_lo8) needs a -DPC_DEV_SWEEPS build (csrc/pc_dev.h)");
        return PC_ERR_ARG;
#endif
    } else if (p.xl) hipLaunchKernelGGL((gemm_dense_kernel<WM, EPI, true, false>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm_dense_kernel<WM, EPI, false, false>), grid, block, 0, s, p);
    return pc_check_launch("gemm_dense_kernel");
}

// Few-row launches of the N = hidden projections (o_proj, down_proj at a few hundred rows: 64-128 tiles of 128 x 256 for
// 256 CUs): the K range is cut into slices that run as separate workgroups and meet in a fixed-order reduction.
// Returns 0 when split-K does not apply (the caller then launches the plain form).
template <int EPI>
int launch_dense_splitk(DenseParams& p, hipStream_t s, float* ws, int64_t ws_bytes, int* launched) {
    static const int forced_ks = [] { const char* e = getenv("PC_DENSE_KS"); return e ? atoi(e) : 0; }();
    static const int forced_bn = [] { const char* e = getenv("PC_DENSE_BN"); return e ? atoi(e) : 0; }();
    *launched = 0;
    if (!ws || p.xscale || (EPI != EPI_ADD && EPI != EPI_STORE)) return 0;
    const int nkt = pc_ceil_div(p.K, BK), mb = pc_ceil_div(p.M, BM);
    // widest panel whose slices still fill the chip: 256-wide tiles do twice the MFMAs per LDS byte of 128-wide ones
    const int b256 = mb * pc_ceil_div(p.N, 256), b128 = mb * pc_ceil_div(p.N, 128);
    bool narrow = forced_bn ? forced_bn == 128 : false;
    int ks = forced_ks ? forced_ks : 256 / b256;
    if (!forced_ks && !forced_bn && ks < 2 && 256 / b128 >= 2) { narrow = true; ks = 256 / b128; }
    if (ks > 8) ks = 8;
    while (ks > 1 && nkt / ks < 8) --ks;                     // keep >= 8 K-steps per slice
    if (ks < 2 || (int64_t)ks * p.M * p.N * 4 > ws_bytes) return 0;
    float* y = p.y; const int64_t ldy = p.ldy;
    p.y = ws; p.ldy = p.N; p.slab_stride = (int64_t)p.M * p.N;
    const int rc = narrow ? launch_dense<4, EPI_STORE>(p, s, ks) : launch_dense<2, EPI_STORE>(p, s, ks);
    if (rc) return rc;
    const int nslab = pc_ceil_div(pc_ceil_div(p.K, BK), p.kper);
    const int64_t items = (int64_t)p.M * (p.N >> 2);
    const dim3 grid((unsigned)((items + 255) / 256)), block(256);
    if (EPI == EPI_ADD) hipLaunchKernelGGL((dense_reduce_kernel<true>), grid, block, 0, s, ws, p.slab_stride, nslab, y, ldy, p.M, p.N);
    else hipLaunchKernelGGL((dense_reduce_kernel<false>), grid, block, 0, s, ws, p.slab_stride, nslab, y, ldy, p.M, p.N);
    *launched = 1;
    return pc_check_launch("dense_reduce_kernel");
}

template <int EPI>
int launch_dense_tile(DenseParams& p, hipStream_t s) {
    static const int forced = [] { const char* e = getenv("PC_DENSE_BN"); return e ? atoi(e) : 0; }();
    // 256-wide weight panels (twice the MFMAs per LDS byte) unless the 128-wide grid needs fewer tile-rounds of the 256
    // CUs: a 128-wide tile pass takes 0.62 of a 256-wide one (measured at K = 4096 and 11008,
    // profiles/r02_dense_splitk.txt), e.g. 1000 x 4096: 128 wide tiles = one round of half the chip, 256 narrow tiles =
    // one round of all of it
    const int cols = (EPI == EPI_SILU) ? p.nfeat * 2 : p.N;
    const int mb = pc_ceil_div(p.M, BM);
    const int b256 = mb * pc_ceil_div(cols, 256), b128 = mb * pc_ceil_div(cols, 128);
    const bool narrow = forced ? forced == 128 : (b256 <= 512 && 0.62f * pc_ceil_div(b128, 256) < 1.0f * pc_ceil_div(b256, 256));
    return narrow ? launch_dense<4, EPI>(p, s) : launch_dense<2, EPI>(p, s);
}

}  // namespace

namespace {
struct Lo8Operands { const void* x_lo8; const float* x_lo8_scale; int64_t ldx8; const void* w8; const float* w8_scale; int64_t ldw8; };
int gemm_dense_impl(const void* x_hi, const void* x_lo, int64_t ldx, const void* w, int64_t ldw, const float* w_scale,
                    const float* x_scale, const float* corr, int64_t ldc, const int32_t* corr_has, int32_t M, int32_t N,
                    int32_t K, int32_t epilogue, float* y, int64_t ldy, void* out_hi, void* out_lo, int64_t ldo, void* stream,
                    void* workspace = nullptr, int64_t ws_bytes = 0, const Lo8Operands* lo8 = nullptr) {
    PC_REQUIRE(M > 0 && N > 0 && K > 0 && K % 8 == 0 && N % 4 == 0, PC_ERR_ARG, "pc_gemm_dense: need M, N, K > 0, K%%8==0, N%%4==0");
    PC_REQUIRE(x_hi && w, PC_ERR_ARG, "pc_gemm_dense: null pointer");
    PC_REQUIRE(ldx >= K && ldw >= K && ldx % 8 == 0 && ldw % 8 == 0 && ((uintptr_t)x_hi & 15) == 0 && ((uintptr_t)w & 15) == 0 &&
               ((uintptr_t)x_lo & 15) == 0, PC_ERR_ARG, "pc_gemm_dense: operands must be 16-byte aligned with row strides %% 8 == 0");
    PC_REQUIRE(!w_scale || ((uintptr_t)w_scale & 15) == 0, PC_ERR_ARG, "pc_gemm_dense: w_scale must be 16-byte aligned");
    DenseParams p;
    memset(&p, 0, sizeof(p));
    p.xh = (const _Float16*)x_hi; p.xl = (const _Float16*)x_lo; p.ldx = ldx;
    p.w = (const _Float16*)w; p.ldw = ldw; p.wscale = w_scale;
    PC_REQUIRE(!x_scale || (w_scale && !x_lo && corr && corr_has && ldc >= N && ldc % 4 == 0 && ((uintptr_t)corr & 15) == 0), PC_ERR_ARG,
               "pc_gemm_dense_a8: int8 activations need w_scale, no lo plane, corr (16-byte aligned, ldc >= N) and corr_has");
    p.xscale = x_scale; p.corr = corr; p.ldc = ldc; p.corr_has = corr_has;
    p.M = M; p.N = N; p.K = K;
    if (lo8) {
        PC_REQUIRE(lo8->x_lo8 && lo8->x_lo8_scale && lo8->w8 && lo8->w8_scale && !x_lo && !x_scale && !w_scale, PC_ERR_ARG,
                   "pc_gemm_dense_lo8: needs the int8 residual plane, its row scales, the int8 weight image and its row scales (and no "
                   "fp16 lo plane / LLM.int8 operands)");
        PC_REQUIRE(K % 64 == 0 && lo8->ldx8 >= K && lo8->ldw8 >= K && lo8->ldx8 % 16 == 0 && lo8->ldw8 % 16 == 0 &&
                   (((uintptr_t)lo8->x_lo8 | (uintptr_t)lo8->w8) & 15) == 0, PC_ERR_ARG,
                   "pc_gemm_dense_lo8: K %% 64 == 0, int8 planes 16-byte aligned with row strides %% 16 == 0");
        p.xl8 = (const signed char*)lo8->x_lo8; p.xl8_scale = lo8->x_lo8_scale; p.ldx8 = lo8->ldx8;
        p.w8 = (const signed char*)lo8->w8; p.w8_scale = lo8->w8_scale; p.ldw8 = lo8->ldw8;
    }
    void* z = nullptr;
    if (hipGetSymbolAddress(&z, HIP_SYMBOL(g_zero_chunk)) != hipSuccess || !z) {
        pc_set_error("pc_gemm_dense: hipGetSymbolAddress failed");
        return PC_ERR_ARG;
    }
    p.zeros = (const _Float16*)z;
    hipStream_t s = (hipStream_t)stream;
    if (epilogue == EPI_SILU || epilogue == EPI_GELU) {
        PC_REQUIRE(out_hi && ldo % 4 == 0 && ((uintptr_t)out_hi & 7) == 0 && ((uintptr_t)out_lo & 7) == 0, PC_ERR_ARG,
                   "pc_gemm_dense: activation epilogues need an 8-byte aligned out_hi plane (out_lo optional)");
        p.oh = (_Float16*)out_hi; p.ol = (_Float16*)out_lo; p.ldo = ldo;
        if (epilogue == EPI_SILU) {
            PC_REQUIRE(N % 8 == 0 && ldo >= N / 2, PC_ERR_ARG, "pc_gemm_dense: SiLU epilogue needs N = 2*inter, inter %% 4 == 0, ldo >= inter");
            p.nfeat = N / 2;
            return launch_dense_tile<EPI_SILU>(p, s);
        }
        PC_REQUIRE(ldo >= N, PC_ERR_ARG, "pc_gemm_dense: GELU epilogue needs ldo >= N");
        return launch_dense_tile<EPI_GELU>(p, s);
    }
    PC_REQUIRE(y && ldy >= N && ldy % 4 == 0 && ((uintptr_t)y & 15) == 0, PC_ERR_ARG, "pc_gemm_dense: bad fp32 output");
    p.y = y; p.ldy = ldy;
    PC_REQUIRE(epilogue == EPI_ADD || epilogue == EPI_STORE, PC_ERR_ARG, "pc_gemm_dense: unknown epilogue %d", epilogue);
    PC_REQUIRE(!workspace || ((uintptr_t)workspace & 15) == 0, PC_ERR_ARG, "pc_gemm_dense_ws: workspace must be 16-byte aligned");
    int launched = 0;
    const int rc = epilogue == EPI_ADD ? launch_dense_splitk<EPI_ADD>(p, s, (float*)workspace, ws_bytes, &launched)
                                       : launch_dense_splitk<EPI_STORE>(p, s, (float*)workspace, ws_bytes, &launched);
    if (rc || launched) return rc;
    p.y = y; p.ldy = ldy; p.slab_stride = 0;
    if (epilogue == EPI_ADD) return launch_dense_tile<EPI_ADD>(p, s);
    return launch_dense_tile<EPI_STORE>(p, s);
}
}  // namespace

PC_EXPORT int pc_gemm_dense(const void* x_hi, const void* x_lo, int64_t ldx, const void* w, int64_t ldw,
                            const float* w_scale, int32_t M, int32_t N, int32_t K, int32_t epilogue, float* y, int64_t ldy,
                            void* out_hi, void* out_lo, int64_t ldo, void* stream) {
    return gemm_dense_impl(x_hi, x_lo, ldx, w, ldw, w_scale, nullptr, nullptr, 0, nullptr, M, N, K, epilogue, y, ldy, out_hi,
                           out_lo, ldo, stream);
}

// pc_gemm_dense with a caller-owned scratch buffer: plain-store / residual-add launches whose tile grid would leave most
// of the 256 CUs idle (a few hundred rows against N = hidden) cut K into slices -- partial slabs [slices][M][N] fp32 in
// `workspace`, added in slice order by a second launch.  Without room for the slabs the call is pc_gemm_dense.
PC_EXPORT int pc_gemm_dense_ws(const void* x_hi, const void* x_lo, int64_t ldx, const void* w, int64_t ldw,
                               const float* w_scale, int32_t M, int32_t N, int32_t K, int32_t epilogue, float* y, int64_t ldy,
                               void* out_hi, void* out_lo, int64_t ldo, void* workspace, int64_t ws_bytes, void* stream) {
    return gemm_dense_impl(x_hi, x_lo, ldx, w, ldw, w_scale, nullptr, nullptr, 0, nullptr, M, N, K, epilogue, y, ldy, out_hi,
                           out_lo, ldo, stream, workspace, ws_bytes);
}

#ifdef PC_DEV_SWEEPS      // (dev builds only: see pc_dev.h)
// pc_gemm_dense with the residual activation plane as int8 codes (pc_quant_rows_i8) against an int8 image of the weights: the
// second plane's product runs on v_mfma_i32_32x32x32_i8 at twice the fp16 MFMA rate (include/promptcache_hip.h).
PC_EXPORT int pc_gemm_dense_lo8(const void* x_hi, int64_t ldx, const void* x_lo8, const float* x_lo8_scale, int64_t ldx8, const void* w,
                                int64_t ldw, const void* w8, const float* w8_scale, int64_t ldw8, int32_t M, int32_t N, int32_t K,
                                int32_t epilogue, float* y, int64_t ldy, void* out_hi, void* out_lo, int64_t ldo, void* workspace,
                                int64_t ws_bytes, void* stream) {
    const Lo8Operands lo8{x_lo8, x_lo8_scale, ldx8, w8, w8_scale, ldw8};
    return gemm_dense_impl(x_hi, nullptr, ldx, w, ldw, nullptr, nullptr, nullptr, 0, nullptr, M, N, K, epilogue, y, ldy, out_hi, out_lo,
                           ldo, stream, workspace, ws_bytes, &lo8);
}
#endif  // PC_DEV_SWEEPS

// The fused q|k|v projection of a many-row pass at head_dim 128 (include/promptcache_hip.h: pc_dense_qkv_args): projection,
// RoPE at the supplied positions, rotated q as hi / lo planes, rotated k and v appended to the arena (+ residual planes) -- the
// [M][(H + 2 Hkv) D] fp32 intermediate and the separate pc_rope_append launch of the round-2 encode are gone.
PC_EXPORT int pc_gemm_dense_qkv_rope(const pc_dense_qkv_args* a, void* stream) {
    PC_REQUIRE(a && a->struct_bytes == (uint32_t)sizeof(pc_dense_qkv_args), PC_ERR_ARG,
               "pc_gemm_dense_qkv_rope: args is NULL or struct_bytes != sizeof(pc_dense_qkv_args) (ABI mismatch)");
    PC_REQUIRE(a->D == 128 && a->B > 0 && a->H > 0 && a->Hkv > 0 && a->q_len > 0 && a->K > 0 && a->K % 8 == 0, PC_ERR_ARG,
               "pc_gemm_dense_qkv_rope: head_dim must be 128 (got %d); B, H, Hkv, q_len, K > 0; K %% 8 == 0", a->D);
    PC_REQUIRE(a->x_hi && a->w && a->cs && a->q_hi && a->k_arena && a->v_arena, PC_ERR_ARG, "pc_gemm_dense_qkv_rope: null pointer");
    PC_REQUIRE(a->ldx >= a->K && a->ldw >= a->K && a->ldx % 8 == 0 && a->ldw % 8 == 0 && ((uintptr_t)a->x_hi & 15) == 0 &&
               ((uintptr_t)a->w & 15) == 0 && ((uintptr_t)a->x_lo & 15) == 0 && ((uintptr_t)a->cs & 15) == 0, PC_ERR_ARG,
               "pc_gemm_dense_qkv_rope: operands and the rotation table must be 16-byte aligned with row strides %% 8 == 0");
    PC_REQUIRE(a->q_token_stride % 4 == 0 && a->arena_head_stride % 4 == 0 && a->arena_batch_stride % 4 == 0 &&
               ((uintptr_t)a->q_hi & 7) == 0 && ((uintptr_t)a->q_lo & 7) == 0 && ((uintptr_t)a->k_arena & 7) == 0 &&
               ((uintptr_t)a->v_arena & 7) == 0, PC_ERR_ARG, "pc_gemm_dense_qkv_rope: outputs must keep 8-byte alignment");
    PC_REQUIRE((a->k_lo == nullptr) == (a->v_lo == nullptr), PC_ERR_ARG, "pc_gemm_dense_qkv_rope: k_lo and v_lo go together");
    PC_REQUIRE(!a->k_lo || (a->lo_head_stride % 4 == 0 && a->lo_batch_stride % 4 == 0 && a->lo_row0 >= 0 &&
                            (a->past_lens ? a->lo_row0 == 0 : a->lo_row0 <= a->past_len)), PC_ERR_ARG,
               "pc_gemm_dense_qkv_rope: residual planes: 8-byte aligned strides, lo_row0 in [0, past_len] (0 with past_lens)");
    PC_REQUIRE((int64_t)a->past_len + a->q_len <= a->cap, PC_ERR_BOUNDS,
               "pc_gemm_dense_qkv_rope: past_len %d + q_len %d exceeds arena rows %d", a->past_len, a->q_len, a->cap);
    DenseParams p;
    memset(&p, 0, sizeof(p));
    p.xh = (const _Float16*)a->x_hi; p.xl = (const _Float16*)a->x_lo; p.ldx = a->ldx;
    p.w = (const _Float16*)a->w; p.ldw = a->ldw;
    p.M = a->B * a->q_len; p.N = (a->H + 2 * a->Hkv) * 128; p.K = a->K;
    if (a->x_lo8) {
        PC_REQUIRE(!a->x_lo && a->x_lo8_scale && a->w8 && a->w8_scale && a->K % 64 == 0 && a->ldx8 >= a->K && a->ldw8 >= a->K &&
                   a->ldx8 % 16 == 0 && a->ldw8 % 16 == 0 && (((uintptr_t)a->x_lo8 | (uintptr_t)a->w8) & 15) == 0, PC_ERR_ARG,
                   "pc_gemm_dense_qkv_rope: the int8 residual plane replaces x_lo; K %% 64 == 0; int8 planes 16-byte aligned, row strides %% 16 == 0");
        p.xl8 = (const signed char*)a->x_lo8; p.xl8_scale = a->x_lo8_scale; p.ldx8 = a->ldx8;
        p.w8 = (const signed char*)a->w8; p.w8_scale = a->w8_scale; p.ldw8 = a->ldw8;
    }
    void* z = nullptr;
    if (hipGetSymbolAddress(&z, HIP_SYMBOL(g_zero_chunk)) != hipSuccess || !z) {
        pc_set_error("pc_gemm_dense_qkv_rope: hipGetSymbolAddress failed");
        return PC_ERR_ARG;
    }
    p.zeros = (const _Float16*)z;
    p.cs = (const float2*)a->cs; p.q_hi = (_Float16*)a->q_hi; p.q_lo = (_Float16*)a->q_lo; p.q_ts = a->q_token_stride;
    p.k_arena = (_Float16*)a->k_arena; p.v_arena = (_Float16*)a->v_arena; p.a_bs = a->arena_batch_stride; p.a_hs = a->arena_head_stride;
    p.k_lo = (_Float16*)a->k_lo; p.v_lo = (_Float16*)a->v_lo; p.lo_bs = a->lo_batch_stride; p.lo_hs = a->lo_head_stride;
    p.lo_row0 = a->lo_row0;
    p.H = a->H; p.Hkv = a->Hkv; p.q_len = a->q_len; p.past_len = a->past_len; p.past_lens = a->past_lens;
    return launch_dense_tile<EPI_ROPE>(p, (hipStream_t)stream);
}

// LLM.int8 form of pc_gemm_dense (pc_int8.hip): xq = activation CODES of pc_quant_act_i8 held in fp16 (row-major [M][K]), w =
// weight codes held in fp16, y = (sum_k w[n][k] * xq[m][k]) * w_scale[n] * x_scale[m] (+ corr[m][n] when *corr_has), then
// the epilogue.  The integer dot product is accumulated in fp32 by the fp16 MFMAs: exact below 2^24 per accumulator.
PC_EXPORT int pc_gemm_dense_a8(const void* xq, int64_t ldx, const void* w_codes, int64_t ldw, const float* w_scale,
                               const float* x_scale, const float* corr, int64_t ldc, const int32_t* corr_has, int32_t M,
                               int32_t N, int32_t K, int32_t epilogue, float* y, int64_t ldy, void* out_hi, void* out_lo,
                               int64_t ldo, void* stream) {
    PC_REQUIRE(w_scale && x_scale, PC_ERR_ARG, "pc_gemm_dense_a8: null pointer");
    return gemm_dense_impl(xq, nullptr, ldx, w_codes, ldw, w_scale, x_scale, corr, ldc, corr_has, M, N, K, epilogue, y, ldy,
                           out_hi, out_lo, ldo, stream);
}
