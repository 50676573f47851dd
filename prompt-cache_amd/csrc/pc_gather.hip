// kv_copy: the module-KV gather (staging) and its inverse (slice-and-store), as ONE batched-memcpy
// kernel over a segment table.
//
// Replaces  PromptCache.update      promptcache/cache_engine.py:135-151  (pc_kv_gather)
//           SchemaCache._process    promptcache/cache_engine.py:283-296  (pc_kv_slice_store)
//
// Data layout in HBM (fp16 everywhere, D = head_dim):
//   staged buffer / encode arena   [n_layers][2][n_kv_heads][cap][D]       plane p = (layer, k|v, head)
//   segment store (module library) [n_layers][2][n_kv_heads][len_s][D]
// Inside one plane the tokens of a segment are contiguous on both sides, so a (segment, plane) pair is
// a flat copy of len_s*D*2 bytes; a segment contributes `planes` such runs whose bases advance by a
// per-side plane stride.  The kernel is HBM-bound: algorithmic bytes = 2 * S * planes * D * 2
// (read once + write once).
//
// Work decomposition: tile = <=32 KiB of one segment-plane (128 tokens at D=128); a workgroup owns one
// tile index for 2 consecutive planes.  Lanes move 16 B each (global_load/store_dwordx4, 1 KiB per
// wave-instruction, fully coalesced); 16 loads are issued before the first store so each lane keeps
// 256 B in flight.  Short segments (the 1-token whitespace runs between PML tags: 17 of the 25
// segments of the persona prompt) fold several planes into one pass so no lane idles.
#include "pc_common.h"

namespace {

constexpr int kMaxSeg = 40;          // descriptors per launch (kernarg-resident, 40 B each)
// tile size / planes per workgroup / loads in flight: see launch_batches (env-overridable for sweeps)
constexpr int kThreads = 256;

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct SegDesc {
    const char* src;
    char* dst;
    int64_t src_plane_stride;  // bytes
    int64_t dst_plane_stride;  // bytes
    int32_t bytes_per_plane;
    int32_t tile_start;        // first tile index of this segment in the launch
};

struct CopyArgs {
    SegDesc seg[kMaxSeg];
    int32_t nseg;
    int32_t planes;
    int32_t tile_bytes;      // per plane per workgroup (multiple of 16)
    int32_t planes_per_wg;
};

template <bool NT, int kUnroll>
__global__ __launch_bounds__(kThreads) void kv_copy_kernel(const CopyArgs a) {
    const int kTileBytes = a.tile_bytes, kPlanesPerWG = a.planes_per_wg;
    const int tile = blockIdx.x;
    // Wave-uniform scan of the (<=40 entry) kernarg table: scalar loads, no divergence.
    int s = 0;
    for (int i = 1; i < a.nseg; ++i) s = (a.seg[i].tile_start <= tile) ? i : s;
    const SegDesc d = a.seg[s];
    const int64_t byte0 = (int64_t)(tile - d.tile_start) * kTileBytes;
    int rem = d.bytes_per_plane - (int)byte0;
    rem = rem > kTileBytes ? kTileBytes : rem;
    const int nchunk = rem >> 4;  // 16-byte chunks in this tile, 1..1024

    // Lane -> (plane-in-pass, chunk) with a power-of-two chunk width w >= min(nchunk, 256).
    const int lg = nchunk >= kThreads ? 8 : (nchunk <= 1 ? 0 : 32 - __builtin_clz(nchunk - 1));
    const int w = 1 << lg;
    const int planes_per_pass = kThreads >> lg;
    const int tid = threadIdx.x;
    const int pl = tid >> lg;
    const int c0 = tid & (w - 1);
    const int nci = (nchunk + w - 1) >> lg;

    const int p0 = blockIdx.y * kPlanesPerWG;
    const int pend = (p0 + kPlanesPerWG < a.planes) ? p0 + kPlanesPerWG : a.planes;
    const int npi = (pend - p0 + planes_per_pass - 1) / planes_per_pass;
    const int niter = npi * nci;

    int pi = 0, ci = 0;  // uniform counters: iteration k = pi * nci + ci
    for (int k0 = 0; k0 < niter; k0 += kUnroll) {
        u32x4 v[kUnroll];
        int64_t doff[kUnroll];
        bool ok[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int p = p0 + pi * planes_per_pass + pl;
            const int c = c0 + ci * w;
            ok[u] = (k0 + u < niter) && (p < pend) && (c < nchunk);
            const int64_t inner = byte0 + (int64_t)c * 16;
            doff[u] = (int64_t)p * d.dst_plane_stride + inner;
            if (ok[u]) {
                const u32x4* sp = (const u32x4*)(d.src + (int64_t)p * d.src_plane_stride + inner);
                v[u] = NT ? __builtin_nontemporal_load(sp) : *sp;
            }
            if (++ci == nci) { ci = 0; ++pi; }
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            if (ok[u]) {
                u32x4* dp = (u32x4*)(d.dst + doff[u]);
                if (NT) __builtin_nontemporal_store(v[u], dp); else *dp = v[u];
            }
        }
    }
}

int launch_batches(SegDesc* descs, int nseg, int planes, hipStream_t stream) {
    // Non-temporal loads/stores (PC_GATHER_NT=1) look +2..5 % in an isolated loop but measured 373 us vs 355 us per
    // launch inside the real TTFT step (the attention that follows re-reads the staged rows): default off.
    static const bool nt = [] { const char* e = getenv("PC_GATHER_NT"); return e && e[0] == '1'; }();
    // Tiling swept inside the real TTFT step on MI355X (tools/gather_sweep.py; persona prompt, 1.81 GB per launch):
    //   16 KiB x 8 planes, 8 loads in flight  : 361 us (5.0 TB/s)      8 KiB x 8, 8 : 336 us (5.4 TB/s)
    //   64 KiB x 2 planes, 16 loads in flight : 327 us (5.5 TB/s)     32 KiB x 2, 16 : 318 us (5.7 TB/s)  <- default
    // i.e. long contiguous runs per workgroup (32 KiB per plane) and 256 B per lane outstanding.
    static const int tile_bytes = [] { const char* e = getenv("PC_GATHER_TILE"); return e ? atoi(e) : 32768; }();
    static const int ppw = [] { const char* e = getenv("PC_GATHER_PPW"); return e ? atoi(e) : 2; }();
    static const int unroll = [] { const char* e = getenv("PC_GATHER_UNROLL"); return e ? atoi(e) : 16; }();
    for (int base = 0; base < nseg; base += kMaxSeg) {
        CopyArgs a;
        const int n = (nseg - base < kMaxSeg) ? nseg - base : kMaxSeg;
        int tiles = 0;
        for (int i = 0; i < n; ++i) {
            a.seg[i] = descs[base + i];
            a.seg[i].tile_start = tiles;
            tiles += (a.seg[i].bytes_per_plane + tile_bytes - 1) / tile_bytes;
        }
        a.nseg = n;
        a.planes = planes;
        a.tile_bytes = tile_bytes;
        a.planes_per_wg = ppw;
        if (tiles == 0) continue;
        dim3 grid(tiles, pc_ceil_div(planes, ppw));
        if (nt) hipLaunchKernelGGL((kv_copy_kernel<true, 8>), grid, dim3(kThreads), 0, stream, a);
        else if (unroll == 8) hipLaunchKernelGGL((kv_copy_kernel<false, 8>), grid, dim3(kThreads), 0, stream, a);
        else if (unroll == 4) hipLaunchKernelGGL((kv_copy_kernel<false, 4>), grid, dim3(kThreads), 0, stream, a);
        else    hipLaunchKernelGGL((kv_copy_kernel<false, 16>), grid, dim3(kThreads), 0, stream, a);
        int rc = pc_check_launch("kv_copy_kernel");
        if (rc != PC_OK) return rc;
    }
    return PC_OK;
}

}  // namespace

PC_EXPORT int pc_kv_gather(const void* const* seg_src, const int32_t* seg_len, const int32_t* seg_dst_off,
                           int32_t nseg, void* dst, int32_t n_layers, int32_t n_kv_heads, int32_t head_dim,
                           int32_t max_ctx, void* stream) {
    PC_REQUIRE(nseg >= 0 && n_layers > 0 && n_kv_heads > 0 && max_ctx > 0, PC_ERR_ARG, "pc_kv_gather: bad sizes");
    PC_REQUIRE(head_dim > 0 && head_dim % 8 == 0, PC_ERR_ARG, "pc_kv_gather: head_dim must be a multiple of 8");
    if (nseg == 0) return PC_OK;
    PC_REQUIRE(seg_src && seg_len && seg_dst_off && dst, PC_ERR_ARG, "pc_kv_gather: null pointer");
    const int64_t row = (int64_t)head_dim * 2;
    SegDesc* descs = (SegDesc*)alloca(sizeof(SegDesc) * (size_t)nseg);
    int n = 0;
    for (int s = 0; s < nseg; ++s) {
        PC_REQUIRE(seg_len[s] >= 0 && seg_dst_off[s] >= 0, PC_ERR_ARG, "pc_kv_gather: negative segment field");
        PC_REQUIRE((int64_t)seg_dst_off[s] + seg_len[s] <= max_ctx, PC_ERR_BOUNDS,
                   "pc_kv_gather: segment %d (off %d, len %d) exceeds max_ctx %d", s, seg_dst_off[s], seg_len[s], max_ctx);
        if (seg_len[s] == 0) continue;
        PC_REQUIRE(seg_src[s] != nullptr, PC_ERR_ARG, "pc_kv_gather: null segment %d", s);
        PC_REQUIRE((int64_t)seg_len[s] * row < (1ll << 31), PC_ERR_ARG, "pc_kv_gather: segment too long");
        SegDesc& d = descs[n++];
        d.src = (const char*)seg_src[s];
        d.dst = (char*)dst + (int64_t)seg_dst_off[s] * row;
        d.bytes_per_plane = (int32_t)(seg_len[s] * row);
        d.src_plane_stride = d.bytes_per_plane;
        d.dst_plane_stride = (int64_t)max_ctx * row;
        d.tile_start = 0;
    }
    return launch_batches(descs, n, n_layers * 2 * n_kv_heads, (hipStream_t)stream);
}

PC_EXPORT int pc_kv_slice_store(const void* src, int32_t src_cap, const int32_t* seg_src_off, const int32_t* seg_len,
                                void* const* seg_dst, int32_t nseg, int32_t n_layers, int32_t n_kv_heads,
                                int32_t head_dim, void* stream) {
    PC_REQUIRE(nseg >= 0 && n_layers > 0 && n_kv_heads > 0 && src_cap > 0, PC_ERR_ARG, "pc_kv_slice_store: bad sizes");
    PC_REQUIRE(head_dim > 0 && head_dim % 8 == 0, PC_ERR_ARG, "pc_kv_slice_store: head_dim must be a multiple of 8");
    if (nseg == 0) return PC_OK;
    PC_REQUIRE(src && seg_src_off && seg_len && seg_dst, PC_ERR_ARG, "pc_kv_slice_store: null pointer");
    const int64_t row = (int64_t)head_dim * 2;
    SegDesc* descs = (SegDesc*)alloca(sizeof(SegDesc) * (size_t)nseg);
    int n = 0;
    for (int s = 0; s < nseg; ++s) {
        PC_REQUIRE(seg_len[s] >= 0 && seg_src_off[s] >= 0, PC_ERR_ARG, "pc_kv_slice_store: negative segment field");
        PC_REQUIRE((int64_t)seg_src_off[s] + seg_len[s] <= src_cap, PC_ERR_BOUNDS,
                   "pc_kv_slice_store: segment %d (off %d, len %d) exceeds arena rows %d", s, seg_src_off[s], seg_len[s], src_cap);
        if (seg_len[s] == 0) continue;
        PC_REQUIRE(seg_dst[s] != nullptr, PC_ERR_ARG, "pc_kv_slice_store: null segment %d", s);
        PC_REQUIRE((int64_t)seg_len[s] * row < (1ll << 31), PC_ERR_ARG, "pc_kv_slice_store: segment too long");
        SegDesc& d = descs[n++];
        d.src = (const char*)src + (int64_t)seg_src_off[s] * row;
        d.dst = (char*)seg_dst[s];
        d.bytes_per_plane = (int32_t)(seg_len[s] * row);
        d.src_plane_stride = (int64_t)src_cap * row;
        d.dst_plane_stride = d.bytes_per_plane;
        d.tile_start = 0;
    }
    return launch_batches(descs, n, n_layers * 2 * n_kv_heads, (hipStream_t)stream);
}

// ---- the staging plan as one entry per staged row (pc_attn `gather_rows`: the attention stages while it reads) --------------
namespace {
constexpr int kRowTabMaxSeg = 1024;   // descriptors one expansion handles (16 KiB of LDS)

// entry of staged row r given the plan's descriptors in LDS (staging order) -- shared by kv_row_table_kernel and the prologue
__device__ __forceinline__ pc_kv_row row_entry(const pc_kv_seg* s_seg, int nseg, int r, uint64_t dst, uint32_t row_bytes,
                                               uint32_t dst_plane_stride16) {
    int lo = 0, hi = nseg;
    while (lo < hi) {                                       // last descriptor with dst_row <= r
        const int mid = (lo + hi) >> 1;
        if (s_seg[mid].dst_row <= r) lo = mid + 1; else hi = mid;
    }
    pc_kv_row e;
    e.base = dst + (uint64_t)r * row_bytes;
    e.plane_stride16 = dst_plane_stride16;
    e.flags = PC_KV_ROW_STAGED;
    if (lo > 0) {
        const pc_kv_seg sg = s_seg[lo - 1];
        if (r < sg.dst_row + sg.len) {
            e.base = (uint64_t)sg.src + (uint64_t)(r - sg.dst_row) * row_bytes;
            e.plane_stride16 = (uint32_t)(((uint64_t)sg.len * row_bytes) >> 4);
            e.flags = 0;
        }
    }
    return e;
}

__global__ __launch_bounds__(256) void kv_row_table_kernel(const pc_kv_seg* __restrict__ segs, const int32_t* __restrict__ nseg_dev,
                                                           int max_seg, const int32_t* __restrict__ total_dev, uint64_t dst,
                                                           uint32_t row_bytes, uint32_t dst_plane_stride16, int max_ctx,
                                                           pc_kv_row* __restrict__ rows) {
    __shared__ pc_kv_seg s_seg[kRowTabMaxSeg];
    int nseg = *nseg_dev;
    nseg = nseg < 0 ? 0 : (nseg > max_seg ? max_seg : nseg);
    for (int i = threadIdx.x; i < nseg; i += blockDim.x) s_seg[i] = segs[i];
    __syncthreads();
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    int total = *total_dev;
    total = total > max_ctx ? max_ctx : total;
    if (r >= total) return;
    rows[r] = row_entry(s_seg, nseg, r, dst, row_bytes, dst_plane_stride16);
}

// ---- pc_prefill_prologue: everything a captured small-q forward does before its first layer, in ONE launch ------------------
// Block roles (all of them read the call's PINNED HOST block directly, with system-scope loads: no ordering inside the launch is
// needed, and the host rewrites the block between replays):
//   [0, n_tok)            embedding row of token t as the fp32 residual stream          (llama2.py:869; ids from the host block)
//   n_tok                 the whole block -> its device twin (past length, live rows, ... for the layers' kernels) and the
//                         (cos, sin) rows of the supplied positions                         (llama2.py:129-147, :204-207)
//   n_tok + 1 ..          the staging plan expanded to one pc_kv_row per staged row (pc_kv_row_table), when `rows` is given
typedef __attribute__((address_space(1))) unsigned long long g64;
__device__ __forceinline__ unsigned long long host_ld8(const void* p) {
    return __hip_atomic_load((g64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct PrologueArgs {
    const char* host; char* dev; int32_t nbytes, n_tok, o_pos, o_words, o_segs, max_seg;
    const _Float16* table; int32_t hidden, vocab; float* x;
    const float* inv_freq; int32_t half_dim; float2* cs;
    pc_kv_row* rows; uint64_t dst; uint32_t row_bytes, dst_plane_stride16; int32_t max_ctx;
};

typedef _Float16 h8g __attribute__((ext_vector_type(8)));
typedef float f4g __attribute__((ext_vector_type(4)));

typedef __attribute__((address_space(1))) unsigned int g32;
__device__ __forceinline__ unsigned int host_ld4(const void* p) {
    return __hip_atomic_load((g32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void prefill_prologue_kernel(const PrologueArgs a) {
    __shared__ pc_kv_seg s_seg[kRowTabMaxSeg];
    const int b = blockIdx.x, tid = threadIdx.x;
    // words[5] != 0: the caller's token ids and position ids are DEVICE tensors (the reference's calling convention,
    // generation_engine.py:96-97) that it copied into the device twin's ids | pos region on this stream before the launch:
    // read them there, and leave that region alone when the block is fetched
    const bool dev_in = host_ld4(a.host + a.o_words + 20) != 0u;
    if (b < a.n_tok) {
        long long id = dev_in ? ((const long long*)a.dev)[b] : (long long)host_ld8(a.host + 8 * b);
        id = id < 0 ? 0 : (id >= a.vocab ? a.vocab - 1 : id);
        for (int i = tid; i < (a.hidden >> 3); i += 256) {
            const h8g v = *(const h8g*)(a.table + id * a.hidden + i * 8);
            float* o = a.x + (int64_t)b * a.hidden + i * 8;
            *(f4g*)o = f4g{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
            *(f4g*)(o + 4) = f4g{(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
        }
        return;
    }
    if (b == a.n_tok) {
        if (!dev_in) {
            for (int i = tid; i < a.nbytes / 8; i += 256) ((unsigned long long*)a.dev)[i] = host_ld8(a.host + 8 * i);
        } else {          // (o_words is only 4-byte aligned: the words and the plan go over in 4-byte pieces)
            for (int i = a.o_words / 4 + tid; i < a.nbytes / 4; i += 256) ((unsigned int*)a.dev)[i] = host_ld4(a.host + 4 * i);
        }
        for (int i = tid; i < a.n_tok * a.half_dim; i += 256) {
            const int t = i / a.half_dim, f = i - t * a.half_dim;
            int pos;
            if (dev_in) {
                pos = ((const int*)(a.dev + a.o_pos))[t];
            } else {
                const unsigned long long w = host_ld8(a.host + a.o_pos + 8 * (t >> 1));  // two int32 positions per 8 bytes
                pos = (int)((t & 1) ? (w >> 32) : (w & 0xffffffffu));
            }
            const float ang = __fmul_rn((float)pos, a.inv_freq[f]);                      // as rope_table_kernel (pc_rope.hip)
            float sn, c;
            sincosf(ang, &sn, &c);
            a.cs[i] = make_float2(c, sn);
        }
        return;
    }
    if (!a.rows) return;
    // words[3] = segments, words[4] = rows of the table (o_words is 4-byte aligned: read the two int32 words separately)
    const unsigned long long w34a = host_ld8(a.host + ((a.o_words + 12) & ~7));
    const unsigned long long w34b = host_ld8(a.host + ((a.o_words + 16) & ~7));
    int nseg = (int)(((a.o_words + 12) & 7) ? (w34a >> 32) : (w34a & 0xffffffffu));
    int total = (int)(((a.o_words + 16) & 7) ? (w34b >> 32) : (w34b & 0xffffffffu));
    nseg = nseg < 0 ? 0 : (nseg > a.max_seg ? a.max_seg : nseg);
    total = total > a.max_ctx ? a.max_ctx : total;
    for (int i = tid; i < nseg; i += 256) {
        const unsigned long long p0 = host_ld8(a.host + a.o_segs + 16 * i), p1 = host_ld8(a.host + a.o_segs + 16 * i + 8);
        pc_kv_seg sg;
        sg.src = (const void*)(uintptr_t)p0; sg.dst_row = (int32_t)(p1 & 0xffffffffu); sg.len = (int32_t)(p1 >> 32);
        s_seg[i] = sg;
    }
    __syncthreads();
    const int r = (b - a.n_tok - 1) * 256 + tid;
    if (r < total) a.rows[r] = row_entry(s_seg, nseg, r, a.dst, a.row_bytes, a.dst_plane_stride16);
}
}  // namespace

PC_EXPORT int pc_prefill_prologue(const void* host_block, void* dev_block, int32_t nbytes, int32_t n_tok, int32_t o_pos,
                                  int32_t o_words, int32_t o_segs, int32_t max_seg, const void* embed_table, int32_t hidden,
                                  int32_t vocab, float* x_out, const float* inv_freq, int32_t head_dim, float* cs_out,
                                  pc_kv_row* rows, const void* dst, int32_t max_ctx, void* stream) {
    PC_REQUIRE(host_block && dev_block && embed_table && x_out && inv_freq && cs_out, PC_ERR_ARG, "pc_prefill_prologue: null pointer");
    PC_REQUIRE(nbytes > 0 && nbytes % 8 == 0 && nbytes <= (1 << 20) && n_tok > 0 && n_tok <= 512, PC_ERR_ARG,
               "pc_prefill_prologue: need 8 | nbytes <= 1 MiB and 1 <= n_tok <= 512");
    PC_REQUIRE(o_pos % 8 == 0 && o_words % 4 == 0 && o_segs % 8 == 0 && o_pos >= 8 * n_tok && o_words >= o_pos + 4 * n_tok &&
               o_words + 32 <= nbytes && (!rows || o_segs + 16 * max_seg <= nbytes), PC_ERR_ARG,
               "pc_prefill_prologue: block layout (ids | pos | words[8] | segs) does not fit nbytes");
    PC_REQUIRE(hidden > 0 && hidden % 8 == 0 && vocab > 0 && head_dim > 0 && head_dim % 2 == 0, PC_ERR_ARG, "pc_prefill_prologue: bad sizes");
    PC_REQUIRE(!rows || (dst && max_ctx > 0 && max_seg > 0 && max_seg <= kRowTabMaxSeg), PC_ERR_ARG,
               "pc_prefill_prologue: the row table needs the staged buffer, max_ctx and 1 <= max_seg <= %d", kRowTabMaxSeg);
    PrologueArgs a;
    a.host = (const char*)host_block; a.dev = (char*)dev_block; a.nbytes = nbytes; a.n_tok = n_tok; a.o_pos = o_pos; a.o_words = o_words;
    a.o_segs = o_segs; a.max_seg = max_seg; a.table = (const _Float16*)embed_table; a.hidden = hidden; a.vocab = vocab; a.x = x_out;
    a.inv_freq = inv_freq; a.half_dim = head_dim / 2; a.cs = (float2*)cs_out; a.rows = rows; a.dst = (uint64_t)(uintptr_t)dst;
    a.row_bytes = (uint32_t)head_dim * 2u; a.dst_plane_stride16 = rows ? (uint32_t)(((uint64_t)max_ctx * a.row_bytes) >> 4) : 0u;
    a.max_ctx = max_ctx;
    const int nblk = n_tok + 1 + (rows ? pc_ceil_div(max_ctx, 256) : 0);
    hipLaunchKernelGGL(prefill_prologue_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, a);
    return pc_check_launch("prefill_prologue_kernel");
}

namespace {
}  // namespace

PC_EXPORT int pc_kv_row_table(const pc_kv_seg* segs, const int32_t* nseg_dev, int32_t max_seg, const int32_t* total_rows_dev,
                              const void* dst, int32_t n_kv_heads, int32_t head_dim, int32_t max_ctx, pc_kv_row* rows, void* stream) {
    PC_REQUIRE(segs && nseg_dev && total_rows_dev && dst && rows, PC_ERR_ARG, "pc_kv_row_table: null pointer");
    PC_REQUIRE(max_seg > 0 && max_seg <= kRowTabMaxSeg, PC_ERR_ARG, "pc_kv_row_table: max_seg must lie in [1, %d]", kRowTabMaxSeg);
    PC_REQUIRE(head_dim > 0 && head_dim % 8 == 0 && max_ctx > 0 && n_kv_heads > 0, PC_ERR_ARG, "pc_kv_row_table: bad sizes");
    const uint32_t row_bytes = (uint32_t)head_dim * 2u;
    PC_REQUIRE(((uint64_t)max_ctx * row_bytes >> 4) < (1ull << 32), PC_ERR_ARG, "pc_kv_row_table: max_ctx too large");
    (void)n_kv_heads;
    hipLaunchKernelGGL(kv_row_table_kernel, dim3(pc_ceil_div(max_ctx, 256)), dim3(256), 0, (hipStream_t)stream, segs, nseg_dev,
                       max_seg, total_rows_dev, (uint64_t)(uintptr_t)dst, row_bytes, (uint32_t)(((uint64_t)max_ctx * row_bytes) >> 4),
                       max_ctx, rows);
    return pc_check_launch("kv_row_table_kernel");
}
