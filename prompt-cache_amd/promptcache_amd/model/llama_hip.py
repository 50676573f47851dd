"""Llama-2-class forward pass on MI355X: HIP kernels (C-ABI) for the prompt-cache hot path, torch
(hipBLASLt) for the dense projections.

Replaces the reference's patched HF model for this path:
  ``LlamaForCausalLM.forward``   promptcache/model/llama2.py:986-1076
  ``LlamaModel.forward``         promptcache/model/llama2.py:822-951
  ``LlamaDecoderLayer.forward``  promptcache/model/llama2.py:600-654
  ``LlamaAttention.forward``     promptcache/model/llama2.py:315-410   (RoPE, KV concat, mask, softmax, PV)
  ``LlamaMLP.forward``           promptcache/model/llama2.py:242

Call surface (what ``CacheEngine`` / ``GenerationEngine`` use):
  ``model(input_ids=[B,q] long, position_ids=[B,q] long, past_key_values=None|seq of (K,V) [B,Hkv,S,D],
          attention_mask=None|[B,q], use_cache=True)`` -> object with ``.logits`` [B,q,V] fp32 and
  ``.past_key_values`` (indexable ``[layer][0|1]`` -> [B,Hkv,S+q,D]).

Differences by design (same numbers, less traffic):
  * new K/V are appended in place to the arena that backs ``past_key_values`` (no ``torch.cat``);
  * the additive mask is never materialised; it is implicit in the attention kernel;
  * the residual stream is kept in fp32 (the parity target is the reference's fp32 CPU path);
  * right-padded batches (``attention_mask`` with trailing zeros, ``cache_engine.py:240-246``) need no
    mask: causality already hides trailing pads from every real token, and pad rows are never read.

There is no CPU / eager fallback: without the HIP extension this module raises.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Dict, Optional

import torch

import numpy as np

from .. import _native
from .config import LlamaShape
from .kv_arena import KVArena, StagedKV, arena_from_past

_SEG_DTYPE = np.dtype([("src", "<u8"), ("dst_row", "<i4"), ("len", "<i4")])      # pc_kv_seg


class _InputBlock:
    """Everything a captured small-q forward reads per call, in ONE pinned host block that the graph's first node
    (``pc_fetch_block``) pulls into its device twin:

        ids   int64 [T] | pos int32 [T] | words int32 [8] = {past_len, residual-tail base, live rows, segments, rows of the row
        table, ids / pos already on the device (0 | 1), -, -} | pc_kv_seg [max_seg]  (the staging plan, when the forward stages
        while it reads)

    Per call the host writes the block (numpy views) and replays the graph: no copy is enqueued, no fill kernel launched.  The
    block may be rewritten once the replay that read it has finished (``acquire`` waits for the event ``release`` records)."""

    def __init__(self, device, T: int, max_seg: int):
        self.T, self.max_seg = T, max_seg
        self.o_pos = 8 * T
        self.o_words = 12 * T
        self.o_segs = (self.o_words + 32 + 15) // 16 * 16
        self.nbytes = self.o_segs + 16 * max_seg
        self.dev = torch.zeros(self.nbytes, dtype=torch.uint8, device=device)
        self.ids = self.dev[:8 * T].view(torch.int64)
        self.pos = self.dev[self.o_pos:self.o_pos + 4 * T].view(torch.int32)
        self.words = self.dev[self.o_words:self.o_words + 32].view(torch.int32)
        self.segs = self.dev[self.o_segs:]
        self.host = torch.zeros(self.nbytes, dtype=torch.uint8, pin_memory=True)
        a = self.host.numpy()
        self.h_ids = a[:8 * T].view(np.int64)
        self.h_pos = a[self.o_pos:self.o_pos + 4 * T].view(np.int32)
        self.h_words = a[self.o_words:self.o_words + 32].view(np.int32)
        self.h_segs = a[self.o_segs:].view(_SEG_DTYPE) if max_seg else None
        self._done = torch.cuda.Event()
        self._busy = False

    def acquire(self) -> None:
        if self._busy:
            self._done.synchronize()
            self._busy = False

    def fetch(self) -> None:
        """(graph node) host block -> device block"""
        _native.fetch_block(self.host, self.dev, self.nbytes)

    def release(self) -> None:
        self._done.record()
        self._busy = True


_GRAPH_RNG_PRIMED: dict = {}


def prime_graph_capture(device) -> None:
    """torch allocates the CUDA generator's graph-capture state (seed / offset words) at the FIRST capture of the process and
    updates it in place at every later one.  The forwards of this package capture under ``torch.inference_mode()``; if theirs is
    the first capture, those words are inference tensors and any capture the CALLER later takes outside inference mode dies in
    ``capture_begin`` ("Inplace update to inference tensor outside InferenceMode").  So the first capture of the process is a
    trivial one taken with inference mode switched off (once, ~1 ms) -- and KEPT: torch frees those words again when the last
    registered graph dies and would re-allocate them under whatever mode the next capture runs in."""
    # (the capture state is per DEVICE: one primed graph per device index this process drives)
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx in _GRAPH_RNG_PRIMED:
        return
    with torch.inference_mode(False), torch.cuda.device(idx):
        t = torch.zeros(8, device=torch.device("cuda", idx))
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            t.add_(1.0)
    _GRAPH_RNG_PRIMED[idx] = (g, t)


def lw0_fp16(layers) -> bool:
    """fp16 weight images (no int8 scales) in the layer stack."""
    return len(layers) > 0 and layers[0].get("wqkv_s") is None


@dataclass
class CausalLMOutput:
    logits: torch.Tensor
    past_key_values: Optional[StagedKV]


class GreedyLoop:
    """Greedy decode steps that never leave the GPU (reference loop: generation_engine.py:123-168, greedy branch).

    Every step is one replay of a captured hipGraph whose last node (``pc_greedy_advance``) takes the argmax of the
    step's logits and writes it -- with position + 1 and past length + 1 -- into the device words the next replay reads
    its inputs from.  The host only enqueues replays (as far ahead as it likes) and picks tokens up from a small ring
    through pinned memory when it needs them for stop conditions; there is no ``int(argmax)`` sync, no H2D fill and no
    logits copy per token.  ``enqueue()`` -> slot index; ``token(i)`` waits for slot i only."""

    RING = 4096

    @torch.inference_mode()
    def __init__(self, model: "LlamaHIP", arena: KVArena, token: int, position: int, max_new: int):
        self.m = model
        past_len = arena.length
        need = past_len + max_new + 2
        if need > arena.cap:
            arena = arena.grown(max(need, 2 * arena.cap))
        self.arena = arena
        st = model._loop_state()
        self.st = st
        # The loop state (token / position / past words, ring, counter) is ONE set of device words per model: the newest
        # loop owns it.  A loop that was superseded (two generate() generators advanced alternately on one model) must
        # not keep replaying over the other's words -- enqueue() refuses instead of corrupting both sequences.
        model._live_loop = self
        dev = model.device
        st["ids"].copy_(torch.tensor([token], dtype=torch.int64), non_blocking=True)
        st["pos"].copy_(torch.tensor([position], dtype=torch.int32), non_blocking=True)
        st["ctr"].zero_()
        self.len0 = past_len
        self.n = 0                       # replays enqueued
        self._last_ent = None
        self.events = []                 # (start, end) per slot
        self.host = torch.empty(self.RING, dtype=torch.int32, pin_memory=True)

    @torch.inference_mode()
    def enqueue(self) -> int:
        m, a, st = self.m, self.arena, self.st
        if getattr(m, "_live_loop", None) is not self:
            raise RuntimeError("GreedyLoop: another greedy loop on this model took over the device loop state "
                               "(one device-side greedy generation per model at a time; set "
                               "GenerationEngine.device_greedy_loop = False to interleave generations)")
        past_len = self.len0 + self.n
        m._lo_mode = m._tail_mode(a, 1, past_len)
        ent = m._loop_graph(a, past_len)
        # host-known words of the step (the kernels read past_len / tail base from the device)
        if self.n == 0 or ent is not self._last_ent:
            st["past"][0:1].fill_(past_len)
            if m._lo_mode == 2:
                st["past"][1:2].fill_(a.tail_base)
        self._last_ent = ent
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ent[0].replay()
        e1.record()
        i = self.n
        self.host[i % self.RING: i % self.RING + 1].copy_(st["ring"][i % self.RING: i % self.RING + 1], non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        self.events.append((e0, e1, done))
        a.length = past_len + 1
        m._tail_done(a, m._lo_mode, 1, past_len)
        self.n += 1
        return i

    def token(self, i: int) -> int:
        self.events[i][2].synchronize()
        return int(self.host[i % self.RING])

    def close(self, consumed: int) -> None:
        """The generation ended after ``consumed`` of the enqueued steps were used: a look-ahead replay enqueued past the
        stop wrote one row too many -- the arena's length goes back to the real sequence, and the loop state is released."""
        if consumed < self.n:
            self.arena.length = self.len0 + consumed
            if self.arena.tail_base >= 0:            # (the residual tail is valid while tail_base + tail_len == length)
                self.arena.tail_len = max(0, self.arena.length - self.arena.tail_base)
        if getattr(self.m, "_live_loop", None) is self:
            self.m._live_loop = None

    def elapsed_ms(self, i: int) -> float:
        e0, e1, done = self.events[i]
        done.synchronize()
        return e0.elapsed_time(e1)


class LlamaHIP:
    """Weights live on one MI355X in fp16; ``wqkv`` and ``wgu`` are the row-concatenated q|k|v and
    gate|up projections so each is one GEMM."""

    SKINNY_MAX_ROWS = 64   # B*q_len at or below this: split-precision weight-streaming kernels inside one hipGraph
    # ... and at or below this the RMSNorms are folded into the projections.  The kernels take two row tiles as well
    # (PC_NORM_FUSED_MAX=32), but at 17..32 rows the six-launch loop measured SLOWER than slabs + pc_rmsnorm_frag (7b, 24 rows:
    # 110 vs 107 us of projections per layer -- o_proj / down_proj lose their K slices; profiles/r04_variants.txt)
    NORM_FUSED_MAX_ROWS = int(os.environ.get("PC_NORM_FUSED_MAX", "16"))
    # ... and up to here: the row-split weight-streaming kernel (gemm_rows_kernel, pc_gemm.hip), launched eagerly; above,
    # pc_gemm_dense.  Crossover against pc_gemm_dense_ws (tools/dense_vs_rows.py, profiles/r02_dense_splitk.txt, 7b layer,
    # split-precision planes on both sides): 259 rows 322 vs 402 us per layer, 512 rows 455 vs 441 us -- the streaming
    # kernel holds up to its own limit of 512 rows (round 1's 256 was measured against the vendor GEMM this tree no
    # longer carries)
    MID_MAX_ROWS = int(os.environ.get("PC_MID_MAX_ROWS", "512"))

    def _setup(self, shape, device, decode_headroom: int) -> None:
        """State every architecture shares: device, KV-arena headroom, workspace and the hipGraph cache."""
        _native.load()  # fail loudly if the extension is missing
        self.config = shape
        self.device = torch.device(device)
        self.dtype = torch.float16
        self.decode_headroom = decode_headroom
        self._ws = None
        # hipGraph cache for the small-q (prefill over staged KV / decode) forward: one captured graph per
        # (B, q_len, arena, split count); past_len, token ids and positions are read from device buffers so
        # every decode step and every same-shaped prompt replays the same graph.
        self.use_graphs = True
        self._graphs = {}
        self.max_graphs = 256   # one per (q_len, split count, mode): a serving mix of prompt lengths stays resident
        self.kslices = 4        # K-slices of the o_proj / down_proj launches (see _forward_skinny)
        # pc_gemm_chain (one persistent launch for o_proj -> gate|up -> down -> next q|k|v): correct and bit-identical, but
        # measured SLOWER than the four launches on MI355X (106-121 vs 88 us per layer, profiles/r02_chain_trace.txt): opt-in
        self.use_chain = os.environ.get("PC_CHAIN", "0") == "1" and _native.has("pc_gemm_chain")      # (dev builds only: csrc/pc_dev.h)
        self._chain_sync = None
        # split-precision activations in the many-row path (see _forward_dense_split); PC_FAST_DENSE=1 trades the
        # full-depth parity for 2x fewer GEMM flops
        self.precise_dense = os.environ.get("PC_FAST_DENSE", "0") != "1"
        # Many-row projections that run on the hi activation plane only (a tile pass of 1 plane instead of 2: half the MFMAs of that
        # launch).  Measured at full depth against the numpy oracle (tests/test_gpu_fullsize.py, 32 layers at the 7b shape; plain
        # init / six 60x outlier channels), max |dlogit| by which projection drops its lo plane: none 1.1e-3 / 7.0e-4, gate|up
        # 3.4e-3 / 3.1e-3, down 2.7e-3 / 2.4e-3, o 2.9e-3 / 2.8e-3, gate|up + down 4.4e-3 / 3.9e-3, q|k|v 1.5e-2 / 7.5e-3 (FAILS:
        # its outputs are the stored K / V and the attention's Q).  gate|up is 45 % of a layer's projection MACs: default.
        # PC_DENSE_LO_SKIP= (empty): every projection on both planes, as in rounds 1-3.
        # Depth matters (the error grows with every layer): the 40-layer 13b stack measured 7.2e-3 with gate|up on one plane against
        # 1.7e-3 on two -- inside the bar, but with little room -- so the default applies up to 32 layers; deeper stacks keep both
        # planes everywhere unless PC_DENSE_LO_SKIP says otherwise.
        _skip = os.environ.get("PC_DENSE_LO_SKIP")
        if _skip is None:
            _skip = "gu" if getattr(shape, "num_hidden_layers", 99) <= 32 else ""
        self.dense_lo_skip = tuple(t for t in _skip.split(",") if t)
        self.mid_lo_skip = tuple(t for t in os.environ.get("PC_MID_LO_SKIP", "").split(",") if t)
        self.fused_dense_qkv = os.environ.get("PC_FUSED_DENSE_QKV", "1") != "0"   # RoPE + KV append in the many-row q|k|v epilogue
        # Round 5: the many-row projections that keep their residual activation plane run it on the INT8 MFMA (pc_gemm_dense_lo8:
        # x_lo as row-wise absmax int8 codes against an int8 image of the weights, exact int32 sums at twice the fp16 MFMA rate:
        # 24 matrix-pipe slots per K-step instead of 32).  The residual only has to be good to a few bits -- it is 2^-11 of the
        # activation -- and is carried to 2^-8 of its row maximum.  MEASURED (profiles/r05_variants.txt): the launches are 3-8 %
        # faster than with the fp16 residual plane (not the 25 % the slot count promises: at ~1.15 PFLOP/s executed the kernel sits at
        # ~75 % of the matrix peak at the clock the chip holds under this load), the persona encode 101.9 -> 104.2 k tok/s, full-depth
        # parity unchanged at 7b (5.1e-3) and 3.3e-3 instead of 2.0e-3 at 13b -- for 6.7 GB of int8 weight images at 7b.  Not worth a
        # second weight image by default: OPT-IN (PC_DENSE_LO8=1).
        self.dense_lo8 = os.environ.get("PC_DENSE_LO8", "0") == "1" and _native.has("pc_gemm_dense_lo8")      # (dev builds only: csrc/pc_dev.h)
        self.encode_mid = os.environ.get("PC_ENC_MID", "1") != "0"    # encode passes of 65..512 rows on the row-split stack
        # keep the fp16 residuals of the K / V rows appended behind a staged cache -- the prompt's own tokens and every
        # decoded token -- in the arena's residual tail and feed them to the attention (the reference keeps those rows in
        # fp32 for the whole generation, llama2.py:361-388, generation_engine.py:123-147; the arena still holds the fp16
        # values).  See _tail_mode / _dense_pass_lo.
        self.new_kv_lo = os.environ.get("PC_NEW_KV_LO", "1") != "0"
        # ... and keep doing so through the DECODE steps of a generation (every decoded row gets a residual, every step's
        # attention reads residual tiles): decode logits 5e-5 from the oracle instead of 2-4e-3 -- both far inside the
        # 1e-2 bar -- for ~5 % of the decode rate.  Opt-in (PC_DECODE_TAIL=1 or model.decode_tail = True).
        self.decode_tail = os.environ.get("PC_DECODE_TAIL", "0") == "1"
        self.defer_merge = os.environ.get("PC_DEFER_MERGE", "1") != "0"    # one-row steps: o_proj merges the attention's partials
        # arrival counters of the single-launch split-KV merge (pc_attn `counters`: zero now, every launch leaves them zero).
        # Opt-in (PC_ATTN_FUSED=1): measured on MI355X the in-launch hand-off (write-through partials, drain, arrival counter,
        # the last arriver's read-back: ~4.5 us on the critical path) costs what the second launch costs -- persona step 3.922 /
        # 3.933 ms fused against 3.907 / 3.908 ms with attn_combine_kernel (gpurun_out/r3b)
        self._attn_counters = torch.zeros(8192, dtype=torch.int32, device=self.device) \
            if os.environ.get("PC_ATTN_FUSED", "0") == "1" else None
        # N = hidden projections of the <= 16-row stack with K split across workgroups and the reduction inside the launch
        # (pc_gemm_skinny_ks): "tiles,slices" per projection, empty = the one-tile-per-workgroup EPI_ADD launch
        def _ks(env, default):
            v = os.environ.get(env, default)
            return tuple(int(t) for t in v.split(",")) if v else None
        # Measured at 12 rows (tools/ks_micro.py, in-graph): down_proj (K = 11008) 22.75 -> 20.16 us at 2 tiles x 2 slices, every
        # other shape of either projection loses to the one-tile launch (o_proj 9.4 -> 10.4 .. 13.5 us; one row: 19.0 -> 20.3 us):
        # the in-launch reduction costs ~3 us, only the long-K launch with its 2:1 activation traffic earns it back.
        self.ks_o, self.ks_down = _ks("PC_KS_O", ""), _ks("PC_KS_DOWN", "2,2")
        self.ks_min_rows = 5       # ... and only with more than 4 rows (decode keeps the one-tile launch)
        self._ks_state = None      # (scratch, counters), lazily
        self._kv_only = False      # set per call (see __call__)
        self._past_lens = None     # set per call: per-row past lengths of a ragged-prefix encode batch
        self.supports_ragged_past = True    # the many-row path takes past_lens (see __call__)
        self._shared = None        # set per call: (trunk arena, per-row prefix lengths int32 [B], their maximum)
        self.supports_greedy_loop = True    # decode steps can run as a device-side loop (GreedyLoop)
        self.llm_int8 = False
        self._last_qt = None
        self.batch_invariant = True         # a row's result does not depend on the other rows of the forward (see llm_int8)
        self.tail_supported = True  # a subclass whose layer loops do not thread `_tail_for` through must switch this off
        self._gather = None         # set per forward: the row table while the attention launches stage as they read
        self._pre = None            # set per captured forward: (fp32 residual stream, rotation table) from pc_prefill_prologue
        self.fused_prologue = os.environ.get("PC_FUSED_PROLOGUE", "1") != "0"
        self.stats = {"fused_gather": 0}     # forwards that carried out a pending staging inside their attention launches
        self._gather_ok_cache, self._nsplit_cache = {}, {}
        # hipGraphs for the 65..512-row forward as well (B = 1, 16-row buckets): nine launches per layer from Python are host-bound
        # at 7b and ~100 rows (PC_GRAPH_MID=0: launched eagerly, as in rounds 1-3)
        self.graph_mid = os.environ.get("PC_GRAPH_MID", "1") != "0"
        self.max_mid_graphs = 8

    # the many-row layer loop of THIS class hands the attention a shared key prefix (__call__'s ``shared_prefix``); a subclass
    # with its own loop says so itself
    _shared_prefix_loop = True
    # The <= 16-row cached prefill of THIS class's layer loops can carry out a pending staging (KVArena.pending) inside its
    # attention launches (pc_attn gather_rows): CacheEngine then defers PromptCache.update's copy to the first lm() call.
    supports_fused_gather = True
    GATHER_MAX_SEG = 512            # segments of a staging plan the captured forward has room for (longer plans: pc_kv_gather)

    @property
    def supports_shared_prefix(self) -> bool:
        return bool(self._shared_prefix_loop and self.precise_dense and not self.llm_int8)

    def __init__(self, shape: LlamaShape, weights: Dict[str, torch.Tensor], device="cuda:0",
                 decode_headroom: int = 256, skinny: bool = True, int8_weights: bool = False):
        """``int8_weights`` (the adapters' ``load_in_8bit=True``): the decoder-layer linears are quantised row-wise to
        int8 (``_native.quantize_rows_int8``); passes of <= 64 rows stream the int8 fragment images (half the bytes, exact
        arithmetic on the dequantised values); longer passes run hipBLASLt on the int8 codes held in fp16 (exact) and
        scale the product per output feature (``_proj``)."""
        self._setup(shape, device, decode_headroom)
        c = shape
        self.H, self.Hkv, self.D, self.L = c.num_attention_heads, c.num_key_value_heads, c.head_dim, c.num_hidden_layers
        dev = self.device

        def w(name):
            t = weights[name]
            if not isinstance(t, torch.Tensor):
                t = torch.from_numpy(t)
            return t.to(device=dev, dtype=self.dtype).contiguous()

        self.embed = w("embed")
        self.norm = w("norm")
        self.lm_head = w("lm_head")
        # Two resident images of every projection (288 GB of HBM: 2 x 13.5 GB for 7b is cheap):
        # row-major [N][K] for the dense (large q: encode / no-cache) GEMMs, and the MFMA-fragment-major
        # image the weight-streaming kernels read as one sequential stream per wave (small q: cached
        # prefill, decode).  Shapes that do not tile (N%16, K%32) simply keep the dense path.
        self.skinny = bool(skinny) and c.hidden_size % 32 == 0 and c.intermediate_size % 32 == 0 and \
            c.vocab_size % 16 == 0 and (self.H * self.D) % 32 == 0
        fr = _native.to_weight_frags if self.skinny else (lambda t: None)
        self.lm_head_f = fr(self.lm_head)
        self.layers = []
        # (the int8 images are cut on pairs of 32-feature k-steps: every GEMM K must be a multiple of 64)
        self.int8_weights = bool(int8_weights) and self.skinny and c.hidden_size % 64 == 0 and \
            c.intermediate_size % 64 == 0 and (self.H * self.D) % 64 == 0
        # load_in_8bit on the Llama family = LLM.int8() as published (int8 weights AND vector-wise int8 activations with the
        # fp16 outlier decomposition, threshold 6.0: what the reference's GPU runs execute through bitsandbytes;
        # csrc/pc_int8.hip, oracle/llmint8_oracle.py).  PC_INT8_WEIGHT_ONLY=1 keeps round 1's weight-only mode instead
        # (split-precision fp16 activations over the int8 weights) -- the mode the Falcon / MPT adapters still run.
        self.llm_int8 = self.int8_weights and os.environ.get("PC_INT8_WEIGHT_ONLY", "0") != "1"
        if self.int8_weights:
            self.MID_MAX_ROWS = self.SKINNY_MAX_ROWS      # the row-split kernel has no int8 variant: 65+ rows go dense
        if self.llm_int8:
            kmax = max(c.hidden_size, c.intermediate_size, self.H * self.D)
            # outlier-column flags, one row per activation slot; 16384 bytes each: the in-launch correction (pc_gemm_*_a8c)
            # scans a whole row, 32 bytes per thread
            self._i8_flags = torch.zeros((4, max(kmax, 16384)), dtype=torch.uint8, device=dev)
            self.i8_fused_corr = os.environ.get("PC_INT8_FUSED_CORR", "1") != "0"
            # <= 16 rows (the cached step, decode): the activation quantisers run INSIDE the projections (pc_gemm_q8): six launches
            # per layer instead of ten.  PC_INT8_INLAUNCH=0: round 4's separate quantiser launches.  PC_Q8_DOWN="tiles,slices" of the
            # K-sliced down_proj (2 / 4 / 8 output tiles per workgroup, 1..8 slices)
            self.i8_inlaunch = os.environ.get("PC_INT8_INLAUNCH", "1") != "0" and c.hidden_size <= 6144 and \
                self.H * self.D <= 6144 and c.intermediate_size <= 16384
            self.q8_down = tuple(int(v) for v in os.environ.get("PC_Q8_DOWN", "4,4").split(","))
            self.q8_down_small = tuple(int(v) for v in os.environ.get("PC_Q8_DOWN_SMALL", "1,1").split(","))   # <= 4 rows (decode)
            self.q8_p_max_rows = int(os.environ.get("PC_Q8_P_MAX_ROWS", "4"))
            self.q8_defer_merge = os.environ.get("PC_Q8_DEFER_MERGE", "1") != "0"
            self.q8_image = os.environ.get("PC_Q8_IMAGE", "1") != "0"    # 5..16 rows: pc_gemm_q8 on the quantiser launches' images
            self._q8_flags = torch.zeros(16384, dtype=torch.uint8, device=dev)
            self._q8_pmax = torch.zeros((c.intermediate_size // 16, 16), dtype=torch.float32, device=dev)
            self._i8_zero = torch.zeros(((self.SKINNY_MAX_ROWS + 15) // 16) * 16 * kmax, dtype=self.dtype, device=dev)
            self.fuse_norm = False                        # activations are quantised between the norm and the projection
            self.batch_invariant = False                  # the fp16 outlier columns are chosen over ALL rows of a call

        prep = self._prep_linear

        for i in range(self.L):
            wqkv = torch.cat([w(f"l{i}.wq"), w(f"l{i}.wk"), w(f"l{i}.wv")], dim=0).contiguous()
            wgu = torch.cat([w(f"l{i}.gate"), w(f"l{i}.up")], dim=0).contiguous()
            wo, wdown = w(f"l{i}.wo"), w(f"l{i}.down")
            if self.skinny and i == 0:
                self._qkv_perm = _native.qkv_rope_row_perm(self.H + 2 * self.Hkv, self.D).to(dev)
            # q|k|v fragment image: rotary pairs share a 16-row tile (pc_gemm_qkv_rope)
            wqkv, wqkv_f, wqkv_s = prep(wqkv, self._qkv_perm if self.skinny else None); t_qkv = self._last_qt
            wo, wo_f, wo_s = prep(wo); t_o = self._last_qt
            wgu, wgu_f, wgu_s = prep(wgu); t_gu = self._last_qt
            wdown, wdown_f, wdown_s = prep(wdown); t_d = self._last_qt
            q8 = {}
            if self.dense_lo8 and self.precise_dense and not self.int8_weights:
                # int8 images (+ row scales) of the projections whose many-row launches keep the residual plane (dense_lo_skip)
                for key, wt, tag in (("wqkv", wqkv, "qkv"), ("wo", wo, "o"), ("wgu", wgu, "gu"), ("wdown", wdown, "down")):
                    if tag not in self.dense_lo_skip and wt.shape[1] % 64 == 0:
                        q8[key + "_q8"], q8[key + "_q8s"] = _native.quantize_rows_int8(wt)
            self.layers.append(dict(ln1=w(f"l{i}.ln1"), ln2=w(f"l{i}.ln2"), wqkv=wqkv, wo=wo, wgu=wgu, wdown=wdown, **q8,
                                    wqkv_t8=t_qkv, wo_t8=t_o, wgu_t8=t_gu, wdown_t8=t_d,
                                    wqkv_f=wqkv_f, wo_f=wo_f, wgu_f=wgu_f, wdown_f=wdown_f,
                                    wqkv_s=wqkv_s[0], wo_s=wo_s[0], wgu_s=wgu_s[0], wdown_s=wdown_s[0],
                                    wqkv_ds=wqkv_s[1], wo_ds=wo_s[1], wgu_ds=wgu_s[1], wdown_ds=wdown_s[1]))
        # exactly the reference formula, evaluated on the CPU like the reference does (llama2.py:121)
        self.inv_freq_cpu = 1.0 / (c.rope_theta ** (torch.arange(0, self.D, 2).float() / self.D))
        self.inv_freq = self.inv_freq_cpu.to(dev)
        self.softmax_scale = 1.0 / math.sqrt(self.D)
        self.fuse_norm = os.environ.get("PC_FUSE_NORM", "1") != "0" and not getattr(self, "llm_int8", False)
        if self.skinny:
            self._qkv_perm_i32 = self._qkv_perm.to(torch.int32).contiguous()

    # ------------------------------------------------------------------------------------------
    def new_arena(self, batch: int, cap: int) -> KVArena:
        return KVArena(batch, self.L, self.Hkv, cap, self.D, self.device, self.dtype)

    def _prep_linear(self, wt: torch.Tensor, perm=None):
        """One projection -> (row-major image for the many-row GEMMs, fragment image for the streaming kernels,
        (fragment-order scales, row-order scales) | (None, None)).  int8 mode: both images hold the int8 codes."""
        fr = _native.to_weight_frags if self.skinny else (lambda t: None)
        if not self.int8_weights:
            src = wt if perm is None else wt[perm].contiguous()
            return wt, fr(src), (None, None)
        q, sc = _native.quantize_rows_int8(wt)
        dense = q.to(self.dtype)                      # the int8 codes, exact in fp16; _proj applies the scales (many-row paths)
        dsc = sc
        self._last_qt = q.t().contiguous() if getattr(self, "llm_int8", False) else None   # [K][N] int8: columns for pc_outlier_corr
        if perm is not None:
            q, sc = q[perm].contiguous(), sc[perm].contiguous()
        return dense, _native.to_weight_frags_i8(q), (sc, dsc)

    def _linear_entries(self, name: str, prepped) -> dict:
        dense, frag, (sc, dsc) = prepped
        return {name: dense, name + "_f": frag, name + "_s": sc, name + "_ds": dsc, name + "_t8": getattr(self, "_last_qt", None)}

    TAIL_HEADROOM = 256     # decoded rows a generation's residual tail has room for past the prompt's own

    def _tail_mode(self, arena: KVArena, q_len: int, past_len: int) -> int:
        """How the weight-streaming paths treat the fp16 residuals of the rows they append (``KVArena.tail_lo``):
        1 = a prefill pass starts a new tail at ``past_len`` (rows [0, q_len)); 2 = a decode step continues the tail the
        prefill started, so every row since the staged cache ended -- prompt tokens and decoded tokens alike -- reaches
        the attention in split precision, as in the reference's fp32 generation (generation_engine.py:123-147);
        0 = no residuals (switched off, a tail that does not cover the rows since its base, or no room left)."""
        if not self.new_kv_lo or not self.tail_supported:
            return 0
        if q_len > 1:
            arena.ensure_tail(q_len + self.TAIL_HEADROOM)
            return 1
        t = arena.tail_lo
        if not self.decode_tail:
            return 0
        # (a caller may rewind the arena by a few rows: the tail then still covers [tail_base, past_len))
        if t is not None and 0 <= arena.tail_base <= past_len <= arena.tail_base + arena.tail_len and \
                past_len - arena.tail_base + 1 <= t.shape[4]:
            return 2
        return 0

    def _dense_pass_lo(self, arena: KVArena, B: int, Hkv: int, q_len: int, past_len: int):
        """Residual planes for the K / V rows a many-row pass appends: ``layer -> (k_lo, v_lo, bs, hs, row0)``.
        An encode arena (``arena.lo``) keeps residuals for all of its rows; a prefill in front of a generation writes them
        into the arena's residual tail so the decode steps after it find them (``_tail_mode``); a batch-padded or kv-only
        pass gets a scratch buffer shared by the layers (each layer's attention consumes it before the next overwrites it)."""
        if arena.lo is not None and arena.lo_len == past_len:
            return (lambda li: arena.lo_planes(li)), True
        if self.new_kv_lo and self.tail_supported and not self._kv_only:
            arena.ensure_tail(q_len + self.TAIL_HEADROOM)
            self._lo_mode = 1
            return (lambda li: arena.tail_planes(li) + (past_len,)), False
        lo = torch.empty((2, B, Hkv, q_len, self.D), dtype=self.dtype, device=self.device)
        scratch = (lo[0], lo[1], Hkv * q_len * self.D, q_len * self.D, past_len)
        return (lambda li: scratch), False

    @staticmethod
    def _tail_done(arena: KVArena, mode: int, q_len: int, past_len: int) -> None:
        if mode == 1:
            arena.tail_base, arena.tail_len = past_len, q_len
        elif mode == 2:
            arena.tail_len = past_len + q_len - arena.tail_base
        else:
            arena.tail_base, arena.tail_len = -1, 0          # rows appended without residuals: the tail no longer covers

    def _workspace(self, nbytes: int) -> Optional[torch.Tensor]:
        if nbytes <= 0:
            return None
        if self._ws is None or self._ws.numel() * 4 < nbytes:
            self._ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=self.device)
        return self._ws

    def _resolve_arena(self, past, B: int, q_len: int, keep_pending: bool = False):
        """-> (arena, past_len) with room for q_len more rows.  A staging the arena still owes (``KVArena.pending``) is carried
        out here unless the caller takes care of it itself (``keep_pending``)."""
        if past is None:
            return self.new_arena(B, q_len + self.decode_headroom), 0
        found = arena_from_past(past, self.L, self.Hkv, self.D)
        if found is not None and not keep_pending:
            found[0].materialize()
        if found is None:
            # foreign tensors (e.g. a caller-built legacy cache): copy once into an arena
            k0 = past[0][0]
            S = k0.shape[-2]
            arena = self.new_arena(B, S + q_len + self.decode_headroom)
            for i in range(self.L):
                k, v = past[i][0], past[i][1]
                if k.dim() == 3:
                    k, v = k.unsqueeze(0), v.unsqueeze(0)
                arena.buf[:, i, 0, :, :S].copy_(k)
                arena.buf[:, i, 1, :, :S].copy_(v)
            arena.length = S
            return arena, S
        arena, S = found
        if arena.B != B:
            raise ValueError(f"past_key_values batch {arena.B} != input batch {B}")
        if S + q_len > arena.cap:
            arena.length = S
            arena = arena.grown(max(S + q_len + self.decode_headroom, 2 * arena.cap))
        return arena, S

    # ------------------------------------------------------------------------------------------
    @torch.inference_mode()
    def __call__(self, input_ids: torch.Tensor, position_ids: Optional[torch.Tensor] = None,
                 past_key_values=None, attention_mask: Optional[torch.Tensor] = None, use_cache: bool = True,
                 last_token_only: bool = False, num_layers: Optional[int] = None, many_rows: bool = False,
                 kv_only: bool = False, past_lens: Optional[torch.Tensor] = None, shared_prefix=None,
                 **_unused) -> CausalLMOutput:
        """``shared_prefix = (trunk arena, [n_pre per batch row])`` (schema encode, split-precision many-row path): batch row b
        attends to rows [0, n_pre[b]) of the trunk arena IN PLACE and then to its own rows, which go into a fresh arena from
        row 0 on -- the suffix passes of an encode without a copy of the trunk per batch row.
        ``past_lens`` (int32 [B], many-row path over an encode arena): one past length per batch row -- row b appends
        behind its own ``past_lens[b]`` rows and attends to those plus its new rows (scaffold suffixes of different
        unions batched over their trunk prefixes); the arena's ``length`` must be their maximum.
        ``kv_only`` (many-row path): stop after the last layer's K / V are in the arena and return no logits -- all a
        schema-encode pass is run for (the reference discards the rest, cache_engine.py:243-296).
        ``many_rows``: take the stacked-GEMM path for more than 64 rows even where the row-split kernel would be
        faster -- it alone keeps the pass's own K/V in split precision, which is what a schema encode wants (its K/V
        are the product)."""
        n = _native
        dev = self.device
        B, q_len = input_ids.shape
        self._shared = None
        if shared_prefix is not None:
            trunk, n_pre = shared_prefix
            if not (many_rows and past_key_values is None and self.supports_shared_prefix and position_ids is not None
                    and len(n_pre) == B and trunk.B == 1 and max(n_pre) <= trunk.length):
                raise ValueError("shared_prefix goes with many_rows=True, explicit position ids, no past_key_values and a one-row "
                                 "trunk arena that holds every prefix")
            self._shared = (trunk, torch.tensor(list(n_pre), device=dev, dtype=torch.int32), int(max(n_pre)))
        arena, past_len = self._resolve_arena(past_key_values, B, q_len, keep_pending=True)
        if many_rows and self.precise_dense and past_key_values is None:
            arena.with_lo()           # a schema-encode pass: keep the residuals of every row it appends
        if position_ids is None:  # llama2.py:859-864
            position_ids = torch.arange(past_len, past_len + q_len, device=input_ids.device).unsqueeze(0).expand(B, q_len)
        position_ids = position_ids.view(-1, q_len)
        if attention_mask is not None:
            # (checked where the mask LIVES: a host mask -- what CacheEngine passes since round 5 -- costs no upload and no
            # device sync; a device mask, the reference's convention (cache_engine.py:246), is read back for the test: one
            # pipeline drain per forward, which is what made 10 % of the schema encode's wall time idle in rounds 1-4)
            am = attention_mask
            # only right padding is expressible without an explicit mask (cache_engine.py:38-47 pads right)
            if am.dim() == 2 and am.shape[1] == q_len and bool((am[:, 1:] > am[:, :-1]).any()):
                raise NotImplementedError("left / interior padding masks are not supported by the HIP path")

        T = B * q_len
        self._kv_only = bool(kv_only)
        self._past_lens = None
        if past_lens is not None:
            if not (many_rows and past_key_values is not None):
                raise ValueError("past_lens goes with many_rows=True over an encode arena")
            self._past_lens = past_lens.to(device=dev, dtype=torch.int32).contiguous()
        # a schema-encode pass (fresh arena, or a suffix over a trunk arena that carries residuals) always takes the
        # many-row path, however few rows it has: that path alone reads the prefix's residual planes and stops after the
        # last layer's K / V; and a pass that is not the serving prefill / decode must not capture a throwaway hipGraph
        encode_pass = many_rows and (self._past_lens is not None or
                                     self.precise_dense and (past_key_values is None or arena.lo is not None))
        graphed = self.skinny and T <= self.SKINNY_MAX_ROWS and self.use_graphs and not many_rows and not kv_only
        mid = self.skinny and not encode_pass and not kv_only and T <= self.MID_MAX_ROWS and \
            not (many_rows and self.precise_dense and T > self.SKINNY_MAX_ROWS)   # (the streaming stacks have no kv_only exit)
        streaming = graphed or mid
        self._lo_mode = self._tail_mode(arena, q_len, past_len) if streaming else 0
        if mid and not graphed and self.graph_mid and self.use_graphs and B == 1 and not many_rows and not self.llm_int8 and \
                T > self.SKINNY_MAX_ROWS and (self._shared_prefix_loop or type(self)._forward_skinny is LlamaHIP._forward_skinny):
            graphed = True                    # a long question over a staged cache: the row-split stack, captured per 16-row bucket
        if graphed:
            # Token ids and positions on the HOST (what GenerationEngine hands over) travel with the call's other words in one
            # pinned copy (_InputBlock); device tensors (the reference's calling convention, generation_engine.py:96-97) are
            # copied into the graph's input block on the device.  A pending staging is consumed by the attention launches.
            logits = self._graphed_skinny(input_ids.reshape(-1), position_ids.reshape(-1), arena, B, q_len, past_len,
                                          last_token_only, num_layers)
            arena.length = past_len + q_len
            self._tail_done(arena, self._lo_mode, q_len, past_len)
            return CausalLMOutput(logits=logits, past_key_values=arena.views() if use_cache else None)
        arena.materialize()           # every other path reads staged rows from the arena itself
        input_ids, position_ids = input_ids.to(dev), position_ids.to(dev)
        pos32 = position_ids.reshape(-1).to(torch.int32).contiguous()
        ids = input_ids.reshape(-1).to(torch.int64).contiguous()
        # an encode pass of 65..512 rows without per-row prefixes (the trunk of a schema, small whole scaffolds) runs the
        # row-split weight-streaming stack as well: same split-precision arithmetic, residuals of every row into arena.lo
        if (encode_pass and self.skinny and self.encode_mid and self._shared_prefix_loop and self._shared is None and self._past_lens is None
                and not self.llm_int8 and self.SKINNY_MAX_ROWS < T <= self.MID_MAX_ROWS and arena.lo is not None
                and arena.lo_len == past_len):
            self._lo_mode = 3                                # residual rows go to arena.lo (arena-shaped, row = key index)
            logits = self._forward_skinny(ids, pos32, None, arena, B, q_len, past_len, last_token_only, num_layers)
            arena.length = past_len + q_len
            arena.lo_len = past_len + q_len
            self._tail_done(arena, 0, q_len, past_len)
            return CausalLMOutput(logits=logits, past_key_values=arena.views() if use_cache else None)
        if mid:
            logits = self._forward_skinny(ids, pos32, None, arena, B, q_len, past_len, last_token_only, num_layers)
            arena.length = past_len + q_len
            self._tail_done(arena, self._lo_mode, q_len, past_len)
            return CausalLMOutput(logits=logits, past_key_values=arena.views() if use_cache else None)

        logits = self._forward_dense(ids, pos32, arena, B, q_len, past_len, last_token_only, num_layers)
        arena.length = past_len + q_len
        self._tail_done(arena, self._lo_mode, q_len, past_len)       # (_dense_pass_lo sets mode 1 when it starts a tail)
        return CausalLMOutput(logits=logits, past_key_values=arena.views() if use_cache else None)

    # ------------------------------------------------------------------------------------------
    def rows_kslices(self, T: int, N: int) -> int:
        """K slices of the N = hidden projections (o_proj, down_proj) of a T-row pass.  65..288 rows run the wide-panel form of
        the row-split kernel (128 columns per workgroup, csrc/pc_gemm_rows.hip): N / 128 panels, so the slices are what fills the
        256 CUs (13b: 40 panels x 6).  Other row counts keep 4."""
        forced = os.environ.get("PC_ROWS_KQ")
        if forced:
            return int(forced)
        if self.SKINNY_MAX_ROWS < T <= 512 and os.environ.get("PC_ROWS_WIDE", "1") != "0":
            blocks = 1 if T <= 288 else 2                     # (289..512 rows: two row blocks per column panel)
            # (above 128 rows six slices beat eight at the 7b shape -- 192 workgroups, fewer slabs for the norm to fold:
            # q = 130 / 194 / 258 / 288: 7.56 / 9.31 / 11.49 / 12.53 -> 7.45 / 9.09 / 10.97 / 11.92 ms; profiles/r04_variants.txt)
            return max(1, min(8 if T <= 128 else 6, 256 // (blocks * -(-N // 128))))
        return self.kslices

    def _proj(self, a_hi, a_lo, lw: dict, key: str, M: int, N: int, K: int, epi: int, **out) -> None:
        """One many-row projection on the hand-written MFMA kernel (pc_gemm_dense.hip): ``(a_hi + a_lo) @ W^T`` with the
        epilogue fused.  int8 mode: ``lw[key]`` holds the int8 codes as fp16 (exact) and the per-output scales
        (``key_ds``) are applied to the accumulator tile -- the many-row paths then compute with exactly the dequantised
        weights the streaming kernels use."""
        ws = None
        if epi in (_native.EPI_ADD, _native.EPI_STORE) and M <= 2048:
            # few-row launches of the N = hidden projections cut K into slices (pc_gemm_dense_ws): the slabs of the widest
            # split it ever takes are 256 tiles x 128 x 256 x 4 B
            ws = getattr(self, "_dense_ws", None)
            if ws is None:
                ws = self._dense_ws = torch.empty(34 << 20, dtype=torch.uint8, device=self.device)
        w8 = lw.get(key + "_q8") if (a_lo is not None and self.dense_lo8) else None
        if w8 is not None:
            # the residual plane on the int8 MFMA: one quantiser launch (M x K x 3 bytes of traffic), then the projection
            codes, sc = self._lo8_codes(a_lo, M, K)
            _native.gemm_dense_lo8(a_hi, codes, sc, lw[key], w8, lw[key + "_q8s"], M, N, K, epi, workspace=ws, **out)
            return
        _native.gemm_dense(a_hi, a_lo, lw[key], M, N, K, epi, wscale=lw.get(key + "_ds"), workspace=ws, **out)

    def _lo8_codes(self, a_lo, M: int, K: int):
        """Row-wise absmax int8 codes + scales of a residual activation plane (pc_quant_rows_i8)."""
        codes = torch.empty((M, K), dtype=torch.int8, device=self.device)
        sc = torch.empty(M, dtype=torch.float32, device=self.device)
        _native.quant_rows_i8(a_lo, M, K, codes, sc)
        return codes, sc

    def _forward_dense(self, ids, pos32, arena, B, q_len, past_len, last_token_only, num_layers):
        """Layer stack for many rows (schema encode, no-cache prefill): every projection is a pc_gemm_dense launch with
        its residual add / SiLU*up fused; RMSNorm, RoPE + KV append and the attention are HIP kernels between them.
        precise_dense (default): split-precision activations (hi, lo fp16 planes) -- an fp16 activation costs 2^-12 per
        projection input and through 32 layers that alone moves 7b-shape logits by 2-3e-2 against the reference's fp32
        path (the module KV an encode stores drifts the same way).  Both planes meet the same weight fragments inside
        the GEMM tile; Q, P and the pass's own K / V rows are split-precision in the attention as well.
        PC_FAST_DENSE=1: the hi plane only (half the MFMA work, fp16-activation accuracy)."""
        if self.llm_int8:
            return self._forward_dense_int8(ids, pos32, arena, B, q_len, past_len, last_token_only, num_layers)
        n = _native
        dev = self.device
        two = self.precise_dense
        H, Hkv, D, hid = self.H, self.Hkv, self.D, self.config.hidden_size
        inter = self.config.intermediate_size
        T = B * q_len
        W = (H + 2 * Hkv) * D
        eps = self.config.rms_norm_eps
        f32 = torch.float32
        cs = torch.empty((T, D // 2, 2), dtype=f32, device=dev)
        n.rope_table(pos32, self.inv_freq, cs, T, D)
        h2 = torch.empty((2, T, hid), dtype=self.dtype, device=dev)          # (hi, lo) of the normalised stream
        n.embed_gather(self.embed, ids, h2[0], T, hid, self.config.vocab_size)
        x = h2[0].float()  # fp32 residual stream
        attn2 = torch.empty((2, T, H * D), dtype=self.dtype, device=dev)
        act2 = torch.empty((2, T, inter), dtype=self.dtype, device=dev)
        q16 = torch.empty((T, H * D), dtype=self.dtype, device=dev)
        q16l = torch.empty((T, H * D), dtype=self.dtype, device=dev) if two else None
        lo_for, full_lo = self._dense_pass_lo(arena, B, Hkv, q_len, past_len) if two else ((lambda li: None), False)
        # shared prefix: the rows of this pass sit at arena rows [0, q_len) and the attention walks the trunk's planes first
        trunk, pre_lens, pre_max = self._shared if self._shared is not None else (None, self._past_lens, past_len)
        trunk_lo = trunk is not None and trunk.lo is not None and trunk.lo_len >= pre_max

        def prefix_of(li):
            if trunk is None:
                return None
            return (trunk.buf[0, li, 0], trunk.buf[0, li, 1], trunk.lo[0, li, 0] if trunk_lo else None,
                    trunk.lo[0, li, 1] if trunk_lo else None, trunk.head_stride)
        ws = self._workspace(n.attn_workspace_bytes(B, H, D, q_len, pre_max + q_len))
        lo = (lambda t: t[1]) if two else (lambda t: None)
        layers = self.layers if num_layers is None else self.layers[:num_layers]

        def norm(src, gain, rows, want_lo=True):
            if two and want_lo:
                n.rmsnorm_split(src, gain, h2[0], h2[1], rows, hid, eps)
            else:
                n.rmsnorm(src, gain, h2[0], rows, hid, eps, True)

        # head_dim 128, fp16 weights: RoPE and the KV append run in the q|k|v projection's epilogue (pc_gemm_dense_qkv_rope);
        # otherwise the projection leaves fp32 [T, W] for pc_rope_append
        fused_qkv = D == 128 and len(layers) > 0 and layers[0].get("wqkv_ds") is None and self.fused_dense_qkv
        qkv = None if fused_qkv else torch.empty((T, W), dtype=f32, device=dev)
        skip = self.dense_lo_skip                  # dev probe (PC_DENSE_LO_SKIP): projections that run on the hi plane only
        lo_q = (lambda t: None) if "qkv" in skip else lo
        lo_o = (lambda t: None) if "o" in skip else lo
        lo_g = (lambda t: None) if "gu" in skip else lo
        lo_d = (lambda t: None) if "down" in skip else lo
        for li, lw in enumerate(layers):
            norm(x, lw["ln1"], T)
            kp, vp = arena.k_plane(li), arena.v_plane(li)
            kv_lo = lo_for(li)
            if fused_qkv:
                xlo, lo8 = lo_q(h2), None
                if xlo is not None and self.dense_lo8 and lw.get("wqkv_q8") is not None:
                    codes, sc = self._lo8_codes(xlo, T, hid)
                    xlo, lo8 = None, (codes, sc, lw["wqkv_q8"], lw["wqkv_q8s"])
                n.gemm_dense_qkv_rope(h2[0], xlo, lw["wqkv"], hid, cs, q16, q16l, H * D, kp, vp, arena.batch_stride,
                                      arena.head_stride, B, H, Hkv, D, q_len, past_len, arena.cap, kv_lo=kv_lo,
                                      past_lens=self._past_lens, lo8=lo8)
            else:
                self._proj(h2[0], lo_q(h2), lw, "wqkv", T, W, hid, n.EPI_STORE, y=qkv)
                n.rope_append(qkv, q_len * W, W, q16, q_len * H * D, H * D, qkv[:, H * D:], qkv[:, (H + Hkv) * D:], q_len * W, W,
                              kp, vp, arena.batch_stride, arena.head_stride, cs, B, H, Hkv, D, q_len, past_len, arena.cap, True,
                              q_out_lo=q16l, kv_lo=kv_lo, past_lens=self._past_lens)
            if self._kv_only and li == len(layers) - 1:
                break             # schema encode: the K / V of the last layer are written; nothing after them is used
            # q_lo: split-precision Q and P in the attention as well (fp16 Q alone costs 1.6e-2 on 32-layer logits)
            n.attn_fwd(q16, q_len * H * D, H * D, kp, vp, arena.batch_stride, arena.head_stride, attn2[0],
                       q_len * H * D, H * D, B, H, Hkv, D, q_len, pre_max, self.softmax_scale, ws, q_lo=q16l,
                       out_lo=lo(attn2), kv_lo=kv_lo, past_lens=pre_lens, prefix=prefix_of(li))
            self._proj(attn2[0], lo_o(attn2), lw, "wo", T, hid, H * D, n.EPI_ADD, y=x)                  # x += attn @ Wo^T
            norm(x, lw["ln2"], T, want_lo="gu" not in skip)
            self._proj(h2[0], lo_g(h2), lw, "wgu", T, 2 * inter, hid, n.EPI_SILU, out_hi=act2[0], out_lo=lo_d(act2))
            self._proj(act2[0], lo_d(act2), lw, "wdown", T, hid, inter, n.EPI_ADD, y=x)                 # x += act @ Wd^T
        if full_lo:
            arena.lo_len = past_len + q_len
        if self._kv_only:
            return None
        head = {"lm_head": self.lm_head}
        V = self.config.vocab_size
        if last_token_only:
            xl = x.view(B, q_len, hid)[:, -1, :].contiguous()
            norm(xl, self.norm, B)
            logits = torch.empty((B, V), dtype=f32, device=dev)
            self._proj(h2[0, :B], h2[1, :B] if two else None, head, "lm_head", B, V, hid, n.EPI_STORE, y=logits)
            return logits.view(B, 1, V)
        norm(x, self.norm, T)
        logits = torch.empty((T, V), dtype=f32, device=dev)
        self._proj(h2[0], lo(h2), head, "lm_head", T, V, hid, n.EPI_STORE, y=logits)   # llama2.py:1050-1051: every row
        return logits.view(B, q_len, V)

    # ------------------------------------------------------------------------------------------
    # LLM.int8 layer stacks (load_in_8bit=True): every decoder-layer projection input is quantised vector-wise to int8 codes
    # (pc_quant_act_i8), the projection runs over weight codes x activation codes and is rescaled by w_scale[n] * x_scale[m]
    # in its epilogue, and columns holding an activation >= 6 are carried in fp16 (pc_outlier_corr) -- Dettmers et al. 2022 as
    # bitsandbytes' Linear8bitLt applies it.  Norms, RoPE, attention, residual stream and lm_head are the fp16-mode kernels.
    def _i8_lin_frag(self, slot, act_hi, K, lw, key, perm, T, N, bufs, norm=None):
        """Quantise a fragment-plane activation (its hi plane is the fp16 value bitsandbytes would see) and prepare the outlier
        correction for projection ``key``: -> (codes, x_scale, corr, has).  ``norm=(x_f32, gain, eps)``: the activation is
        RMSNorm(x) -- normalised, written to ``act_hi`` and quantised in ONE launch (pc_rmsnorm_quant_i8)."""
        n = _native
        codes, xs, corr, has, c8 = bufs
        if norm is not None:
            n.rmsnorm_quant_i8(norm[0], norm[1], norm[2], T, K, act_hi, codes, xs, self._i8_flags[slot], self._i8_flags[(slot + 1) % 4],
                               codes8=c8)
        else:
            n.quant_act_i8(act_hi, True, T, K, codes, xs, self._i8_flags[slot], self._i8_flags[(slot + 1) % 4], codes8=c8)
        n.outlier_corr(self._i8_flags[slot], K, act_hi, codes, True, xs, lw[key + "_t8"], lw[key + "_ds"], perm, T, N, corr, has)
        return codes, xs, corr, has

    def _forward_skinny_int8(self, ids, pos32, past_dev, arena, B, q_len, past_len, last_token_only, num_layers):
        n = _native
        dev = self.device
        c = self.config
        H, Hkv, D, hid, inter = self.H, self.Hkv, self.D, c.hidden_size, c.intermediate_size
        T = B * q_len
        W = (H + 2 * Hkv) * D
        eps = c.rms_norm_eps
        mt = (T + 15) // 16
        f32 = torch.float32
        cs = torch.empty((T, D // 2, 2), dtype=f32, device=dev)
        n.rope_table(pos32, self.inv_freq, cs, T, D)
        h16 = torch.empty((T, hid), dtype=self.dtype, device=dev)
        n.embed_gather(self.embed, ids, h16, T, hid, c.vocab_size)
        x = h16.float()
        q16 = torch.empty((T, H * D), dtype=self.dtype, device=dev)
        q16l = torch.empty((T, H * D), dtype=self.dtype, device=dev)
        ws = torch.empty(max(n.attn_workspace_bytes(B, H, D, q_len, past_len + q_len), 4) // 4, dtype=f32, device=dev)
        tail = self._tail_for(arena, past_dev)

        def planes(k, cnt=2):
            return tuple(torch.empty((mt, k // 32, 64, 8), dtype=self.dtype, device=dev) for _ in range(cnt))

        xh, xl, xq = planes(hid, 3)
        ah, al, aq = planes(H * D, 3)
        ch, cl, cq = planes(inter, 3)
        zero = self._i8_zero
        q8_all = self.i8_inlaunch and T <= self.q8_p_max_rows          # every quantiser inside its projection
        q8_down = self.i8_inlaunch and self.i8_fused_corr and T <= 16    # at least down_proj's
        # (the correction-has words of the two-launch forms: one fill node per forward, not needed when the projections quantise)
        has = None if q8_all else torch.zeros(4, dtype=torch.int32, device=dev)

        def image8(k):        # the int8 MFMA's operand image of a code plane (pc_quant_act_i8 codes8; K % 64 == 0 with int8 weights)
            return torch.empty((mt, k // 64, 64, 16), dtype=torch.int8, device=dev)

        x8, a8, c8 = image8(hid), image8(H * D), image8(inter)
        bufs = None if q8_all else \
            [(xq, torch.empty(T, dtype=f32, device=dev), torch.empty((T, W), dtype=f32, device=dev), has[0:1], x8),
             (aq, torch.empty(T, dtype=f32, device=dev), torch.empty((T, hid), dtype=f32, device=dev), has[1:2], a8),
             (xq, torch.empty(T, dtype=f32, device=dev), torch.empty((T, 2 * inter), dtype=f32, device=dev), has[2:3], x8),
             (cq, torch.empty(T, dtype=f32, device=dev), torch.empty((T, hid), dtype=f32, device=dev), has[3:4], c8)]
        layers = self.layers if num_layers is None else self.layers[:num_layers]
        # Outlier flags are set-only and slot s is cleared by the quantiser of slot s - 1: a pass that stopped behind a
        # layer's q|k|v (kv_only encodes) left slot 0 set, and this pass's first quantiser would OR onto it -- results would
        # depend on the call history.  One memset node (graph-capturable) makes every forward start clean.
        if not q8_all:
            self._i8_flags[0].zero_()
        if q8_all:
            # every projection derives its input's codes / row scales / outlier flags itself (csrc/pc_gemm_q8.hip): q|k|v and gate|up
            # from the fp32 residual stream (RMSNorm folded in), o_proj from the attention's fp16 plane, down_proj from the plane, the
            # per-tile row maxima and the flag bytes gate|up's SiLU epilogue leaves (the o_proj launch zeroes those flag bytes)
            qf, pm = self._q8_flags, self._q8_pmax
            sc, ctr = self._ks_buffers(hid)
            for li, lw in enumerate(layers):
                kp, vp = arena.k_plane(li), arena.v_plane(li)
                kvlo, lo_base = tail(li)
                lo4 = (None, None, 0, 0) if not kvlo else kvlo[:4]
                n.gemm_q8(epilogue=n.EPI_QKV_ROPE, wf=lw["wqkv_f"], w_scale=lw["wqkv_s"], w_codes_t=lw["wqkv_t8"], row_perm=self._qkv_perm_i32,
                          x=x, norm_weight=lw["ln1"], eps=eps, M=T, K=hid, cs=cs, q_hi=q16, q_lo=q16l, q_token_stride=H * D, k_arena=kp,
                          v_arena=vp, arena_batch_stride=arena.batch_stride, arena_head_stride=arena.head_stride, B=B, H=H, Hkv=Hkv, D=D,
                          q_len=q_len, past_len=past_len, cap=arena.cap, past_len_dev=past_dev, k_lo=lo4[0], v_lo=lo4[1],
                          lo_batch_stride=lo4[2], lo_head_stride=lo4[3], lo_base=lo_base)
                # (one row -- a decode step: the attention leaves its split-KV partials in `ws` and the o_proj launch merges them in
                # its prologue; the merge launch, 4.8 us of a 75 us layer, disappears)
                ns = n.attn_fwd(q16, q_len * H * D, H * D, kp, vp, arena.batch_stride, arena.head_stride, None, 0, 0,
                                B, H, Hkv, D, q_len, past_len, self.softmax_scale, ws, past_len_dev=past_dev, out_frag=(ah, al),
                                q_lo=q16l, kv_lo=kvlo,
                                gather=None if self._gather is None else (self._gather, li * 2 * Hkv, (li * 2 + 1) * Hkv),
                                defer_merge=self.q8_defer_merge and T == 1 and H * D <= 4096)
                src = dict(xf_hi=ah) if ns <= 1 else dict(part_o=ws, part_ml=ws[H * ns * D:], part_nsplit=ns, part_head_dim=D)
                n.gemm_q8(epilogue=n.EPI_ADD, wf=lw["wo_f"], w_scale=lw["wo_s"], w_codes_t=lw["wo_t8"], M=T, N=hid, K=H * D,
                          y=x, ldy=hid, flags_clear=qf, clear_bytes=qf.numel(), **src)
                n.gemm_q8(epilogue=n.EPI_SILU, wf=lw["wgu_f"], w_scale=lw["wgu_s"], w_codes_t=lw["wgu_t8"], x=x, norm_weight=lw["ln2"],
                          eps=eps, M=T, N=2 * inter, K=hid, of_hi=ch, row_max_out=pm, flags_out=qf)
                dn = self.q8_down_small if T <= 4 else self.q8_down
                n.gemm_q8(epilogue=n.EPI_ADD, wf=lw["wdown_f"], w_scale=lw["wdown_s"], w_codes_t=lw["wdown_t8"], xf_hi=ch, row_max=pm,
                          row_max_units=inter // 16, flags_in=qf, M=T, N=hid, K=inter, y=x, ldy=hid, ks_tiles=dn[0],
                          kslices=dn[1], ks_scratch=sc, ks_scratch_bytes=sc.numel() * 4, ks_counters=ctr)
            layers = []
        elif self.i8_fused_corr:
            # the outlier correction runs INSIDE the projection launches (pc_gemm_*_a8c): 10 launches per layer instead of 14
            fl = self._i8_flags
            pm = self._q8_pmax
            ksc, kctr = self._ks_buffers(hid) if q8_down else (None, None)
            q8_img = q8_down and self.q8_image and hid <= 6144

            def quant(slot, act_hi, K, buf, norm=None):
                codes, xs = buf[0], buf[1]
                if norm is not None:
                    n.rmsnorm_quant_i8(norm[0], norm[1], eps, T, K, act_hi, codes, xs, fl[slot], fl[(slot + 1) % 4], codes8=buf[4])
                else:
                    n.quant_act_i8(act_hi, True, T, K, codes, xs, fl[slot], fl[(slot + 1) % 4], codes8=buf[4])
                return codes, xs

            for li, lw in enumerate(layers):
                kp, vp = arena.k_plane(li), arena.v_plane(li)
                kvlo, lo_base = tail(li)
                cd, xs = quant(0, xh, hid, bufs[0], norm=(x, lw["ln1"]))
                if q8_img:
                    # (5..16 rows: the quantiser stays a launch, but the projection is pc_gemm_q8's -- the image copied to LDS once,
                    # activation operands read from there, two weight blocks in flight; same bits as pc_gemm with x_scale + flags)
                    lo4 = (None, None, 0, 0) if not kvlo else kvlo[:4]
                    n.gemm_q8(epilogue=n.EPI_QKV_ROPE, wf=lw["wqkv_f"], w_scale=lw["wqkv_s"], w_codes_t=lw["wqkv_t8"], row_perm=self._qkv_perm_i32,
                              xf_hi=xh, x_codes8=x8, x_scale=xs, x_flags=fl[0], M=T, K=hid, cs=cs, q_hi=q16, q_lo=q16l, q_token_stride=H * D,
                              k_arena=kp, v_arena=vp, arena_batch_stride=arena.batch_stride, arena_head_stride=arena.head_stride, B=B, H=H,
                              Hkv=Hkv, D=D, q_len=q_len, past_len=past_len, cap=arena.cap, past_len_dev=past_dev, k_lo=lo4[0], v_lo=lo4[1],
                              lo_batch_stride=lo4[2], lo_head_stride=lo4[3], lo_base=lo_base)
                else:
                    n.gemm_qkv_rope_a8c(lw["wqkv_f"], lw["wqkv_s"], cd, zero, xs, fl[0], xh, lw["wqkv_t8"], self._qkv_perm_i32, T, hid, cs,
                                        q16, q16l, H * D, kp, vp, arena.batch_stride, arena.head_stride, B, H, Hkv, D, q_len, past_len,
                                        arena.cap, past_dev, kv_lo=kvlo and kvlo[:4], lo_base=lo_base, codes8=x8)
                n.attn_fwd(q16, q_len * H * D, H * D, kp, vp, arena.batch_stride, arena.head_stride, None, 0, 0,
                           B, H, Hkv, D, q_len, past_len, self.softmax_scale, ws, past_len_dev=past_dev, out_frag=(ah, al),
                           q_lo=q16l, kv_lo=kvlo,
                           gather=None if self._gather is None else (self._gather, li * 2 * Hkv, (li * 2 + 1) * Hkv))
                cd, xs = quant(1, ah, H * D, bufs[1])
                if q8_img and H * D <= 6144:
                    n.gemm_q8(epilogue=n.EPI_ADD, wf=lw["wo_f"], w_scale=lw["wo_s"], w_codes_t=lw["wo_t8"], xf_hi=ah, x_codes8=a8, x_scale=xs,
                              x_flags=fl[1], M=T, N=hid, K=H * D, y=x, ldy=hid)
                else:
                    n.gemm_skinny_a8c(lw["wo_f"], lw["wo_s"], cd, zero, xs, fl[1], ah, lw["wo_t8"], T, hid, H * D, n.EPI_ADD, y=x, ldy=hid,
                                      codes8=a8)
                cd, xs = quant(2, xh, hid, bufs[2], norm=(x, lw["ln2"]))
                # (more than q8_p_max_rows rows: the quantisers of q|k|v, o_proj and gate|up stay launches -- a 12-row prologue in
                # every workgroup costs more vector ALU time than the launch it saves -- but down_proj reads what this SiLU
                # epilogue leaves: flag slot 3 was zeroed by the quantiser above, slot 0 is zeroed by the down_proj launch)
                if q8_img:
                    n.gemm_q8(epilogue=n.EPI_SILU, wf=lw["wgu_f"], w_scale=lw["wgu_s"], w_codes_t=lw["wgu_t8"], xf_hi=xh, x_codes8=x8, x_scale=xs,
                              x_flags=fl[2], M=T, N=2 * inter, K=hid, of_hi=ch, of_lo=cl, row_max_out=pm, flags_out=fl[3])
                else:
                    n.gemm_skinny_a8c(lw["wgu_f"], lw["wgu_s"], cd, zero, xs, fl[2], xh, lw["wgu_t8"], T, 2 * inter, hid, n.EPI_SILU,
                                      of_hi=ch, of_lo=cl, codes8=x8, row_max_out=pm if q8_down else None, flags_out=fl[3] if q8_down else None)
                if q8_down:
                    n.gemm_q8(epilogue=n.EPI_ADD, wf=lw["wdown_f"], w_scale=lw["wdown_s"], w_codes_t=lw["wdown_t8"], xf_hi=ch, row_max=pm,
                              row_max_units=inter // 16, flags_in=fl[3], M=T, N=hid, K=inter, y=x, ldy=hid, ks_tiles=self.q8_down[0],
                              kslices=self.q8_down[1], ks_scratch=ksc, ks_scratch_bytes=ksc.numel() * 4, ks_counters=kctr,
                              flags_clear=fl[0], clear_bytes=16384)
                    continue
                cd, xs = quant(3, ch, inter, bufs[3])
                n.gemm_skinny_a8c(lw["wdown_f"], lw["wdown_s"], cd, zero, xs, fl[3], ch, lw["wdown_t8"], T, hid, inter, n.EPI_ADD,
                                  y=x, ldy=hid, codes8=c8)
            layers = []
        for li, lw in enumerate(layers):
            kp, vp = arena.k_plane(li), arena.v_plane(li)
            kvlo, lo_base = tail(li)
            cd, xs, corr, hs = self._i8_lin_frag(0, xh, hid, lw, "wqkv", self._qkv_perm_i32, T, W, bufs[0], norm=(x, lw["ln1"], eps))
            n.gemm_qkv_rope_a8(lw["wqkv_f"], lw["wqkv_s"], cd, zero, xs, corr, hs, T, hid, cs, q16, q16l, H * D, kp, vp,
                               arena.batch_stride, arena.head_stride, B, H, Hkv, D, q_len, past_len, arena.cap, past_dev,
                               kv_lo=kvlo and kvlo[:4], lo_base=lo_base, codes8=x8)
            n.attn_fwd(q16, q_len * H * D, H * D, kp, vp, arena.batch_stride, arena.head_stride, None, 0, 0,
                       B, H, Hkv, D, q_len, past_len, self.softmax_scale, ws, past_len_dev=past_dev, out_frag=(ah, al),
                       q_lo=q16l, kv_lo=kvlo, counters=self._counters_for(B, H),
                       gather=None if self._gather is None else (self._gather, li * 2 * Hkv, (li * 2 + 1) * Hkv))
            cd, xs, corr, hs = self._i8_lin_frag(1, ah, H * D, lw, "wo", None, T, hid, bufs[1])
            n.gemm_skinny_a8(lw["wo_f"], lw["wo_s"], cd, zero, xs, corr, hs, T, hid, H * D, n.EPI_ADD, y=x, ldy=hid, codes8=a8)
            cd, xs, corr, hs = self._i8_lin_frag(2, xh, hid, lw, "wgu", None, T, 2 * inter, bufs[2], norm=(x, lw["ln2"], eps))
            n.gemm_skinny_a8(lw["wgu_f"], lw["wgu_s"], cd, zero, xs, corr, hs, T, 2 * inter, hid, n.EPI_SILU, of_hi=ch, of_lo=cl, codes8=x8)
            cd, xs, corr, hs = self._i8_lin_frag(3, ch, inter, lw, "wdown", None, T, hid, bufs[3])
            n.gemm_skinny_a8(lw["wdown_f"], lw["wdown_s"], cd, zero, xs, corr, hs, T, hid, inter, n.EPI_ADD, y=x, ldy=hid, codes8=c8)
        V = c.vocab_size
        if last_token_only:
            xlast = x.view(B, q_len, hid)[:, -1, :].contiguous()
            lh, ll = planes(hid)
            n.rmsnorm_frag(xlast, self.norm, lh, ll, B, hid, eps)
            logits = torch.empty((B, V), dtype=f32, device=dev)
            n.gemm_skinny(self.lm_head_f, lh, ll, B, V, hid, n.EPI_STORE, y=logits, ldy=V)
            return logits.view(B, 1, V)
        n.rmsnorm_frag(x, self.norm, xh, xl, T, hid, eps)
        logits = torch.empty((T, V), dtype=f32, device=dev)
        n.gemm_skinny(self.lm_head_f, xh, xl, T, V, hid, n.EPI_STORE, y=logits, ldy=V)
        return logits.view(B, q_len, V)

    def _forward_dense_int8(self, ids, pos32, arena, B, q_len, past_len, last_token_only, num_layers):
        """Many-row LLM.int8 stack: row-major activations, pc_gemm_dense_a8 over the codes.  Q, P and the pass's own K / V rows
        stay split-precision in the attention, as in the fp16 mode."""
        n = _native
        dev = self.device
        H, Hkv, D, hid = self.H, self.Hkv, self.D, self.config.hidden_size
        inter = self.config.intermediate_size
        T = B * q_len
        W = (H + 2 * Hkv) * D
        eps = self.config.rms_norm_eps
        f32 = torch.float32
        cs = torch.empty((T, D // 2, 2), dtype=f32, device=dev)
        n.rope_table(pos32, self.inv_freq, cs, T, D)
        h16 = torch.empty((T, hid), dtype=self.dtype, device=dev)
        n.embed_gather(self.embed, ids, h16, T, hid, self.config.vocab_size)
        x = h16.float()
        hq = torch.empty((T, hid), dtype=self.dtype, device=dev)
        attn2 = torch.empty((2, T, H * D), dtype=self.dtype, device=dev)
        aq = torch.empty((T, H * D), dtype=self.dtype, device=dev)
        act = torch.empty((T, inter), dtype=self.dtype, device=dev)
        cq = torch.empty((T, inter), dtype=self.dtype, device=dev)
        q16 = torch.empty((T, H * D), dtype=self.dtype, device=dev)
        q16l = torch.empty((T, H * D), dtype=self.dtype, device=dev)
        qkv = torch.empty((T, W), dtype=f32, device=dev)
        xs = torch.empty(T, dtype=f32, device=dev)
        corr = torch.empty(T * max(W, 2 * inter), dtype=f32, device=dev)
        has = torch.zeros(4, dtype=torch.int32, device=dev)
        lo_for, full_lo = self._dense_pass_lo(arena, B, Hkv, q_len, past_len)
        ws = self._workspace(n.attn_workspace_bytes(B, H, D, q_len, past_len + q_len))
        layers = self.layers if num_layers is None else self.layers[:num_layers]
        fl = self._i8_flags
        fl[0].zero_()                            # (see _forward_skinny_int8: a kv_only pass leaves slot 0 set)

        def lin(slot, a16, codes, K, lw, key, N, epi, **out):
            n.quant_act_i8(a16, False, T, K, codes, xs, fl[slot], fl[(slot + 1) % 4])
            cv = corr[:T * N].view(T, N)
            n.outlier_corr(fl[slot], K, a16, codes, False, xs, lw[key + "_t8"], lw[key + "_ds"], None, T, N, cv, has[slot:slot + 1])
            n.gemm_dense_a8(codes, lw[key], lw[key + "_ds"], xs, cv, has[slot:slot + 1], T, N, K, epi, **out)

        for li, lw in enumerate(layers):
            n.rmsnorm(x, lw["ln1"], h16, T, hid, eps, True)
            lin(0, h16, hq, hid, lw, "wqkv", W, n.EPI_STORE, y=qkv)
            kp, vp = arena.k_plane(li), arena.v_plane(li)
            kv_lo = lo_for(li)
            n.rope_append(qkv, q_len * W, W, q16, q_len * H * D, H * D, qkv[:, H * D:], qkv[:, (H + Hkv) * D:], q_len * W, W,
                          kp, vp, arena.batch_stride, arena.head_stride, cs, B, H, Hkv, D, q_len, past_len, arena.cap, True,
                          q_out_lo=q16l, kv_lo=kv_lo, past_lens=self._past_lens)
            if self._kv_only and li == len(layers) - 1:
                break
            n.attn_fwd(q16, q_len * H * D, H * D, kp, vp, arena.batch_stride, arena.head_stride, attn2[0],
                       q_len * H * D, H * D, B, H, Hkv, D, q_len, past_len, self.softmax_scale, ws, q_lo=q16l,
                       out_lo=attn2[1], kv_lo=kv_lo, past_lens=self._past_lens)
            lin(1, attn2[0], aq, H * D, lw, "wo", hid, n.EPI_ADD, y=x)
            n.rmsnorm(x, lw["ln2"], h16, T, hid, eps, True)
            lin(2, h16, hq, hid, lw, "wgu", 2 * inter, n.EPI_SILU, out_hi=act)
            lin(3, act, cq, inter, lw, "wdown", hid, n.EPI_ADD, y=x)
        if full_lo:
            arena.lo_len = past_len + q_len
        if self._kv_only:
            return None
        V = self.config.vocab_size
        head = {"lm_head": self.lm_head}
        h2 = torch.empty((2, T, hid), dtype=self.dtype, device=dev)
        if last_token_only:
            xl = x.view(B, q_len, hid)[:, -1, :].contiguous()
            n.rmsnorm_split(xl, self.norm, h2[0], h2[1], B, hid, eps)
            logits = torch.empty((B, V), dtype=f32, device=dev)
            self._proj(h2[0, :B], h2[1, :B], head, "lm_head", B, V, hid, n.EPI_STORE, y=logits)
            return logits.view(B, 1, V)
        n.rmsnorm_split(x, self.norm, h2[0], h2[1], T, hid, eps)
        logits = torch.empty((T, V), dtype=f32, device=dev)
        self._proj(h2[0], h2[1], head, "lm_head", T, V, hid, n.EPI_STORE, y=logits)
        return logits.view(B, q_len, V)

    # ------------------------------------------------------------------------------------------
    def _layers_norm_fused(self, x, cs, q16, q16l, ws, ah, al, ch, cl, arena, layers, B, q_len, past_len, past_dev,
                           last_token_only, tail=None):
        """T <= 32 rows (NORM_FUSED_MAX_ROWS): six launches per layer.  Both RMSNorms are folded into the projections that consume them
        (pc_gemm_*_norm read the fp32 residual stream directly) and both residual adds into the o_proj / down_proj
        epilogues, so x is the only activation that round-trips through memory in fp32."""
        n = _native
        c = self.config
        H, Hkv, D, hid, inter = self.H, self.Hkv, self.D, c.hidden_size, c.intermediate_size
        T, eps, V = B * q_len, c.rms_norm_eps, c.vocab_size
        # fp16 weights: everything between two attention calls -- o_proj, gate|up, down_proj and the NEXT layer's q|k|v -- is
        # one persistent launch (pc_gemm_chain: the same four bodies, bit-identical, but the weight stream runs through
        # the seams); three launches per layer instead of six.  Shapes without an instantiation fall back here.
        rows_dev = past_dev[2:3] if (past_dev is not None and past_dev.numel() > 2 and B == 1) else None
        chain = self.use_chain and layers and layers[0]["wqkv_s"] is None and T <= 16
        if chain and self._chain_sync is None:
            self._chain_sync = n.chain_sync_state(self.device)
        qkv_done = False
        for li, lw in enumerate(layers):
            kp, vp = arena.k_plane(li), arena.v_plane(li)
            kvlo, lo_base = tail(li) if tail is not None else (None, -1)
            if not qkv_done:
                n.gemm_qkv_rope_norm(lw["wqkv_f"], x, lw["ln1"], eps, T, hid, cs, q16, q16l, H * D, kp, vp, arena.batch_stride,
                                     arena.head_stride, B, H, Hkv, D, q_len, past_len, arena.cap, past_dev,
                                     kv_lo=kvlo and kvlo[:4], wscale=lw["wqkv_s"], lo_base=lo_base, rows_dev=rows_dev)
            qkv_done = False
            # (one row -- a decode step: the attention leaves its split-KV partials and o_proj merges them in its prologue,
            # pc_gemm_part; PC_DEFER_MERGE=0 keeps the merge launch)
            # (rows_dev: a one-row graph has one live row)
            part_ok = self.defer_merge and T == 1 and B == 1 and H * D <= 4096 and lw["wo_s"] is None and not chain and \
                self._attn_counters is None
            ns = n.attn_fwd(q16, q_len * H * D, H * D, kp, vp, arena.batch_stride, arena.head_stride, None, 0, 0,
                            B, H, Hkv, D, q_len, past_len, self.softmax_scale, ws, past_len_dev=past_dev, out_frag=(ah, al),
                            q_lo=q16l, kv_lo=kvlo, counters=self._counters_for(B, H),
                            gather=None if self._gather is None else (self._gather, li * 2 * Hkv, (li * 2 + 1) * Hkv),
                            defer_merge=part_ok)
            if chain:
                nxt = None
                if li + 1 < len(layers):
                    nl = layers[li + 1]
                    nkv, nbase = tail(li + 1) if tail is not None else (None, -1)
                    nxt = dict(wqkv_f=nl["wqkv_f"], ln1=nl["ln1"], cs=cs, q_hi=q16, q_lo=q16l, q_ts=H * D,
                               k_arena=arena.k_plane(li + 1), v_arena=arena.v_plane(li + 1), a_bs=arena.batch_stride,
                               a_hs=arena.head_stride, B=B, H=H, Hkv=Hkv, D=D, q_len=q_len, past_len=past_len, cap=arena.cap,
                               past_len_dev=past_dev, kv_lo=nkv and nkv[:4], lo_base=nbase)
                try:
                    n.gemm_chain(lw["wo_f"], ah, al, H * D, x, T, hid, lw["wgu_f"], lw["ln2"], eps, inter, ch, cl, lw["wdown_f"],
                                 self._chain_sync, qkv=nxt)
                    qkv_done = nxt is not None
                    continue
                except RuntimeError:
                    chain = self.use_chain = False       # no instantiation for this shape: nothing was launched
            if ns > 1:
                n.gemm_part(lw["wo_f"], ws, ws[H * ns * D:], ns, H, D, hid, x)
            elif self.ks_o and lw["wo_s"] is None and self.ks_min_rows <= T <= 32 and (T <= 16 or (self.ks_o[1] <= 4 and self.ks_o[0] in (1, 2, 4))):   # (in-launch K reduction)
                sc, ctr = self._ks_buffers(hid)
                n.gemm_skinny_ks(lw["wo_f"], ah, al, T, hid, H * D, x, hid, self.ks_o[1], self.ks_o[0], sc, ctr, rows_dev=rows_dev)
            else:
                n.gemm_skinny(lw["wo_f"], ah, al, T, hid, H * D, n.EPI_ADD, y=x, ldy=hid, wscale=lw["wo_s"], rows_dev=rows_dev)  # x += attn @ Wo^T
            n.gemm_skinny_norm(lw["wgu_f"], x, lw["ln2"], eps, T, 2 * inter, hid, n.EPI_SILU, of_hi=ch, of_lo=cl,
                               wscale=lw["wgu_s"], rows_dev=rows_dev)
            if self.ks_down and lw["wdown_s"] is None and self.ks_min_rows <= T <= 32 and (T <= 16 or (self.ks_down[1] <= 4 and self.ks_down[0] in (1, 2, 4))) and inter >= 2 * hid:
                sc, ctr = self._ks_buffers(hid)
                n.gemm_skinny_ks(lw["wdown_f"], ch, cl, T, hid, inter, x, hid, self.ks_down[1], self.ks_down[0], sc, ctr, rows_dev=rows_dev)
            else:
                n.gemm_skinny(lw["wdown_f"], ch, cl, T, hid, inter, n.EPI_ADD, y=x, ldy=hid, wscale=lw["wdown_s"], rows_dev=rows_dev)  # x += act @ Wd^T
        if last_token_only:
            xs = x.view(B, q_len, hid)[:, -1, :].contiguous()
            logits = torch.empty((B, V), dtype=torch.float32, device=self.device)
            n.gemm_skinny_norm(self.lm_head_f, xs, self.norm, eps, B, V, hid, n.EPI_STORE, y=logits, ldy=V)
            return logits.view(B, 1, V)
        logits = torch.empty((T, V), dtype=torch.float32, device=self.device)
        n.gemm_skinny_norm(self.lm_head_f, x, self.norm, eps, T, V, hid, n.EPI_STORE, y=logits, ldy=V, rows_dev=rows_dev)
        return logits.view(B, q_len, V)

    # rows a captured small-q graph is padded to (B = 1, q > 1): a prompt of q new tokens replays the graph of its bucket with
    # pad tokens BEHIND its own (under the causal mask nothing reaches back from them; their K/V rows lie past the arena's
    # length and are overwritten by the first decode steps), so a question length seen for the first time does not pay an
    # eager pass + capture (cold-shape TTFT 9.5-10.7 ms against ~4 ms warm) as long as its bucket was seen.  0 = exact q.
    # The projections read the number of LIVE rows from a device word next to past_len (pc_gemm rows_dev) and do not load the
    # pad rows' activations, so one graph per row tile (16) costs the 12-row headline prompt nothing measurable.
    graph_row_bucket = int(os.environ.get("PC_GRAPH_BUCKET", "16"))

    def _graph_rows(self, arena, B: int, q_len: int, past_len: int, last_token_only: bool) -> int:
        g = self.graph_row_bucket
        if g <= 1 or B != 1 or q_len == 1 or last_token_only or self.llm_int8 or self.use_chain:
            return q_len          # (LLM.int8 picks its outlier columns over all rows of a call: no pad rows there)
        qb = (q_len + g - 1) // g * g
        if qb > (self.MID_MAX_ROWS if self.graph_mid else self.SKINNY_MAX_ROWS) or past_len + qb > arena.cap:
            return q_len
        if (qb + 15) // 16 != (q_len + 15) // 16 or (q_len <= 32) != (qb <= 32):
            return q_len          # never across a row-tile count or the residual-tail regime of the attention
        return qb

    def _gather_ok(self, arena, B: int, q_len: int, past_len: int, num_layers: Optional[int] = None) -> bool:
        """Will the attention launches of this (bucketed) forward run on the kernel that stages while it reads?  Asked of the
        library itself (pc_attn_gather_ok) with the arguments _forward_skinny passes; the answer only depends on the row count,
        the residual mode and whether the key range reaches the streaming kernel's minimum, so it is remembered."""
        plan = arena.pending
        if plan is None or plan.total != past_len or len(plan.segs) > self.GATHER_MAX_SEG:
            return False
        if num_layers is not None and num_layers < self.L:
            return False      # a partial-depth forward would stage only its own layers' planes: materialise the whole plan instead
        return self._gather_shape_ok(arena, B, q_len, past_len)

    def _gather_shape_ok(self, arena, B: int, q_len: int, past_len: int) -> bool:
        if not self.supports_fused_gather or B != 1 or self.use_chain or past_len <= 0:
            return False
        ck = (q_len, self._lo_mode, past_len + q_len >= 256)
        ok = self._gather_ok_cache.get(ck)
        if ok is None:
            buf = arena.buf
            kvlo = (arena.tail_planes(0) + (-1,)) if self._lo_mode == 1 else None
            H, D = self.H, self.D
            ok = _native.attn_gather_ok(buf, q_len * H * D, H * D, arena.k_plane(0), arena.v_plane(0), arena.batch_stride,
                                        arena.head_stride, None, 0, 0, B, H, self.Hkv, D, q_len, past_len, self.softmax_scale,
                                        None, past_len_dev=buf, out_frag=(buf, buf), q_lo=buf, kv_lo=kvlo)
            self._gather_ok_cache[ck] = ok
        return ok

    def _nsplit_key(self, B: int, q_len: int, kv_len: int) -> int:
        k = (B, q_len, kv_len)
        v = self._nsplit_cache.get(k)
        if v is None:
            if len(self._nsplit_cache) > 4096:
                self._nsplit_cache.clear()
            v = self._nsplit_cache[k] = _native.attn_workspace_bytes(B, self.H, self.D, q_len, kv_len)   # monotone in the split count
        return v

    def _graph_key(self, arena, B, q_len, past_len, last_token_only, num_layers, gather):
        mode = self._lo_mode
        tail = arena.tail_lo if mode else None
        # (a staging forward captures the address of the arena's row table as well: a NEW arena that happens to land on the freed
        # buffer address of an old one must not replay a graph that writes through the old arena's freed row table)
        return (B, q_len, arena.buf.data_ptr(), arena.cap, self._nsplit_key(B, q_len, past_len + q_len), bool(last_token_only), num_layers,
                self.fuse_norm, self.use_chain, mode, tail.data_ptr() if mode else 0, tail.shape[4] if mode else 0,
                arena.row_table().data_ptr() if gather else 0)

    def _graph_entry(self, key, arena, B, q_len, past_len, last_token_only, num_layers, gather, eager_first=True):
        """The captured forward for ``key`` -> ``[graph, input block, logits buffer]`` (captured now when it does not exist yet:
        the block's host side must already hold valid inputs; ``eager_first`` runs the forward once before capturing)."""
        ent = self._graphs.pop(key, None)
        if ent is not None:
            self._graphs[key] = ent                          # LRU: a hit moves the entry to the young end
            return ent, False
        n = _native
        T = B * q_len
        if len(self._graphs) >= self.max_graphs:
            self._graphs.pop(next(iter(self._graphs)))       # evict the least recently used
        if T > self.SKINNY_MAX_ROWS:
            # the captured 65..512-row forwards hold their slabs and planes in the graph's pool (a few hundred MB each): a handful
            big = [k for k in self._graphs if isinstance(k[0], int) and k[0] * k[1] > self.SKINNY_MAX_ROWS]
            if len(big) >= self.max_mid_graphs:
                self._graphs.pop(big[0])
        blk = _InputBlock(self.device, T, self.GATHER_MAX_SEG if gather else 0)
        ent = [None, blk, None]
        self._graphs[key] = ent
        return ent, True

    def _fused_pro(self) -> bool:
        """Does a captured forward of this model start with pc_prefill_prologue (one launch: block fetch, embedding, rotation
        table, row table)?"""
        return self.fused_prologue and type(self)._forward_skinny is LlamaHIP._forward_skinny and not self.llm_int8

    def _capture(self, ent, arena, B, q_len, past_len, last_token_only, num_layers, gather, eager_first=True):
        n = _native
        blk = ent[1]

        fused_pro = self._fused_pro()

        def run():
            if fused_pro:
                # ONE launch: block fetch + embedding rows (fp32 residual stream) + rotation table + row table
                T = B * q_len
                x = torch.empty((T, self.config.hidden_size), dtype=torch.float32, device=self.device)
                cs = torch.empty((T, self.D // 2, 2), dtype=torch.float32, device=self.device)
                n.prefill_prologue(blk.host, blk.dev, blk.nbytes, T, blk.o_pos, blk.o_words, blk.o_segs, blk.max_seg, self.embed,
                                   self.config.hidden_size, self.config.vocab_size, x, self.inv_freq, self.D, cs,
                                   rows=arena.row_table() if gather else None, dst=arena.buf if gather else None,
                                   max_ctx=arena.cap if gather else 0)
                self._pre = (x, cs)
            else:
                blk.fetch()
                if gather:
                    n.kv_row_table(blk.segs, blk.words[3:4], self.GATHER_MAX_SEG, blk.words[4:5], arena.buf, self.Hkv, self.D, arena.cap,
                                   arena.row_table())
            if gather:
                self._gather = arena.row_tab
            try:
                return self._forward_skinny(blk.ids, blk.pos, blk.words, arena, B, q_len, past_len, last_token_only, num_layers)
            finally:
                self._gather = None
                self._pre = None

        if eager_first:
            # one eager pass first (loads code objects / sizes the allocator), then capture
            run()
            torch.cuda.synchronize()
        prime_graph_capture(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = run()
        ent[0], ent[2] = g, out

    def _graphed_skinny(self, ids, pos, arena, B, q_len, past_len, last_token_only, num_layers):
        """Replay (capturing on first use) the hipGraph of the small-q forward for this shape.  ``ids`` / ``pos``: flat integer
        tensors; host tensors are the fast path (device tensors -- the reference's calling convention -- are read back first)."""
        q_real = q_len
        q_len = self._graph_rows(arena, B, q_len, past_len, last_token_only)
        if q_len != q_real:
            self._lo_mode = self._tail_mode(arena, q_len, past_len)
        gather = self._gather_ok(arena, B, q_len, past_len, num_layers)
        plan = arena.pending if gather else None
        if not gather:
            arena.materialize()
        mode = self._lo_mode
        key = self._graph_key(arena, B, q_len, past_len, last_token_only, num_layers, gather)
        ent, fresh = self._graph_entry(key, arena, B, q_len, past_len, last_token_only, num_layers, gather)
        T = B * q_len
        blk = ent[1]
        # ---- this call's inputs: written into the pinned block the graph's first node fetches ----
        blk.acquire()
        n_real = q_real * B
        w = blk.h_words
        dev_in = ids.is_cuda or pos.is_cuda
        if dev_in and self._fused_pro():
            # device inputs (the reference's calling convention, generation_engine.py:96-97): copied into the device twin's
            # ids | pos region on this stream; the prologue reads them there (words[5]) -- no read-back, no host sync
            blk.ids[:n_real].copy_(ids, non_blocking=True)
            blk.pos[:n_real].copy_(pos, non_blocking=True)
            if q_len != q_real:                              # pad rows BEHIND the prompt's own (see _graph_rows)
                blk.ids[n_real:].zero_()
                blk.pos[n_real:] = blk.pos[n_real - 1] + torch.arange(1, T - n_real + 1, dtype=torch.int32, device=self.device)
            w[5] = 1
        else:
            if dev_in:                                       # (stacks without the fused prologue: correct, one read-back)
                ids, pos = ids.cpu(), pos.cpu()
            blk.h_ids[:n_real] = ids.numpy()
            blk.h_pos[:n_real] = pos.numpy()
            if q_len != q_real:                              # pad rows BEHIND the prompt's own (see _graph_rows)
                blk.h_ids[n_real:] = 0
                blk.h_pos[n_real:] = int(blk.h_pos[n_real - 1]) + np.arange(1, T - n_real + 1, dtype=np.int32)
            w[5] = 0
        w[0] = past_len
        w[1] = arena.tail_base if mode == 2 else 0
        w[2] = n_real                                        # rows that carry tokens: the projections do not load the pad rows' activations
        w[4] = min(past_len + q_len, arena.cap)              # rows of the row table (this pass's own rows: entries that point at the arena)
        if gather:
            w[3] = len(plan.segs)
            blk.h_segs[:len(plan.segs)] = plan.seg_array(_SEG_DTYPE)
        if fresh:
            try:
                self._capture(ent, arena, B, q_len, past_len, last_token_only, num_layers, gather)
            except BaseException:
                self._graphs.pop(key, None)                  # no half-built entry: the next call captures again
                raise
        g, out = ent[0], ent[2]
        g.replay()
        blk.release()
        if gather:
            arena.pending = None                             # the staged rows are in the arena now
            self.stats["fused_gather"] += 1
        res = out[:, :q_real].clone() if q_len != q_real else out.clone()
        if fresh and self.prewarm_tiles and B == 1 and q_real > 1 and T <= self.SKINNY_MAX_ROWS and not last_token_only:
            self._prewarm(arena, past_len + q_real, num_layers)
        return res

    # Capture the sibling row tiles of a prompt-sized forward as soon as the first one is captured: a question whose tile count
    # (16 / 32 / 48 / 64 rows) is new then replays instead of paying an eager pass + capture on its own TTFT (cold-shape TTFT
    # 14.7 ms against 4.7 ms warm in round 3).  Capture only -- nothing is executed, so the arena and the residual tail are not
    # touched; the host spends ~5 ms per tile, once per arena, behind the replay of the request that triggered it.
    prewarm_tiles = os.environ.get("PC_PREWARM_TILES", "1") != "0"

    def _prewarm(self, arena, past_len: int, num_layers) -> None:
        if self.llm_int8 or self.use_chain or self.graph_row_bucket != 16 or self.decode_tail:
            return
        saved = self._lo_mode
        try:
            for q in (16, 32, 48, 64):
                if past_len + q > arena.cap:
                    break
                self._lo_mode = self._tail_mode(arena, q, past_len)
                # both ways a prompt of this tile may arrive: rows already staged, or a staging plan for the attention to carry out
                for gather in ((False, True) if self._gather_shape_ok(arena, 1, q, past_len) else (False,)):
                    key = self._graph_key(arena, 1, q, past_len, False, num_layers, gather)
                    if key in self._graphs:
                        continue
                    ent, fresh = self._graph_entry(key, arena, 1, q, past_len, False, num_layers, gather)
                    if fresh:
                        try:
                            self._capture(ent, arena, 1, q, past_len, False, num_layers, gather, eager_first=False)
                        except Exception:                       # a capture that cannot be taken cold is simply taken on first use
                            self._graphs.pop(key, None)
                            self.prewarm_tiles = False
                            return
        finally:
            self._lo_mode = saved

    @torch.inference_mode()
    def _loop_state(self) -> dict:
        """Device words shared by every captured decode-loop graph of this model: token id, position id, {past length,
        residual-tail base}, the token ring and its counter (see GreedyLoop)."""
        st = getattr(self, "_loop_st", None)
        if st is None:
            dev = self.device
            st = dict(ids=torch.zeros(1, dtype=torch.int64, device=dev), pos=torch.zeros(1, dtype=torch.int32, device=dev),
                      past=torch.zeros(2, dtype=torch.int32, device=dev), ring=torch.zeros(GreedyLoop.RING, dtype=torch.int32, device=dev),
                      ctr=torch.zeros(1, dtype=torch.int32, device=dev))
            self._loop_st = st
        return st

    def _loop_graph(self, arena: KVArena, past_len: int):
        """The captured graph of ONE greedy decode step over ``arena`` (forward of the token in the loop state + argmax +
        state advance), keyed like ``_graphed_skinny``."""
        n = _native
        nsplit_key = n.attn_workspace_bytes(1, self.H, self.D, 1, past_len + 1)
        mode = self._lo_mode
        key = ("loop", arena.buf.data_ptr(), arena.cap, nsplit_key, self.fuse_norm, self.use_chain, mode,
               arena.tail_lo.data_ptr() if mode else 0, arena.tail_lo.shape[4] if mode else 0)
        ent = self._graphs.pop(key, None)
        if ent is not None:
            self._graphs[key] = ent
            return ent
        if len(self._graphs) >= self.max_graphs:
            self._graphs.pop(next(iter(self._graphs)))
        st = self._loop_state()
        st["past"][0:1].fill_(past_len)
        if mode == 2:
            st["past"][1:2].fill_(arena.tail_base)
        V = self.config.vocab_size
        # one eager pass first (loads code objects / sizes the allocator): it rewrites the K / V row the first replay
        # writes again and does not touch the loop state
        self._forward_skinny(st["ids"], st["pos"], st["past"], arena, 1, 1, past_len, False, None)
        torch.cuda.synchronize()
        prime_graph_capture(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = self._forward_skinny(st["ids"], st["pos"], st["past"], arena, 1, 1, past_len, False, None)
            n.greedy_advance(out, V, st["ids"], st["pos"], st["past"], st["ring"], st["ctr"])
        ent = (g, out)
        self._graphs[key] = ent
        return ent

    def greedy_loop(self, past, token: int, position: int, max_new: int) -> Optional["GreedyLoop"]:
        """A device-side greedy decode loop over the arena behind ``past`` (None when this model / cache cannot host one:
        the caller then steps through ``__call__``)."""
        if not (self.skinny and self.use_graphs and self.supports_greedy_loop):
            return None
        found = arena_from_past(past, self.L, self.Hkv, self.D)
        if found is None or found[0].B != 1:
            return None
        arena, S = found
        arena.length = S
        return GreedyLoop(self, arena, token, position, max_new)

    def _ks_buffers(self, hid: int):
        """Scratch slabs + arrival counters of pc_gemm_skinny_ks (shared by every such launch of the model: they run one
        after another on one stream; the counters are zero between launches)."""
        if self._ks_state is None:
            sc = torch.empty(_native.gemm_skinny_ks_scratch_bytes(hid, 8) // 4, dtype=torch.float32, device=self.device)
            self._ks_state = (sc, torch.zeros(hid // 16, dtype=torch.int32, device=self.device))
        return self._ks_state

    def _counters_for(self, B: int, H: int):
        """Arrival counters for the in-launch split merge (PC_ATTN_FUSED=1) -- not for a forward that stages while it reads: that
        one runs the two-launch form of the streaming kernel (pc_attn_gather_ok)."""
        c = self._attn_counters
        return c if c is not None and self._gather is None and B * H <= c.numel() else None

    def _tail_for(self, arena, past_dev):
        """Per-layer ``((k_lo, v_lo, batch_stride, head_stride, lo_row0) | None, lo_base)`` for the current tail mode."""
        mode = self._lo_mode
        if mode == 0:
            return lambda li: (None, -1)
        if mode == 3:                                        # encode pass: the arena's own residual planes, row = key index
            return lambda li: (arena.lo_planes(li), 0)
        base = -1 if mode == 1 else (-2 if past_dev is not None else arena.tail_base)
        return lambda li: (arena.tail_planes(li) + (base,), base)

    def _forward_skinny(self, ids, pos32, past_dev, arena, B, q_len, past_len, last_token_only, num_layers):
        """Layer stack for T = B*q_len <= 64 rows: every projection is a weight-streaming pc_gemm_skinny
        launch over fragment-major weights with split-precision (hi/lo fp16) activations; residual adds and
        SiLU*up are fused into GEMM epilogues; RMSNorm and attention emit the fragment planes directly.
        ``past_dev`` (device int32[1]) makes the pass graph-capturable: the kernels read past_len from it."""
        if self.llm_int8:
            return self._forward_skinny_int8(ids, pos32, past_dev, arena, B, q_len, past_len, last_token_only, num_layers)
        n = _native
        dev = self.device
        c = self.config
        H, Hkv, D, hid, inter = self.H, self.Hkv, self.D, c.hidden_size, c.intermediate_size
        T = B * q_len
        W = (H + 2 * Hkv) * D
        eps = c.rms_norm_eps
        mt = (T + 15) // 16
        if self._pre is not None:                     # (captured forward: pc_prefill_prologue already produced both)
            x, cs = self._pre
        else:
            cs = torch.empty((T, D // 2, 2), dtype=torch.float32, device=dev)
            n.rope_table(pos32, self.inv_freq, cs, T, D)
            h16 = torch.empty((T, hid), dtype=self.dtype, device=dev)
            n.embed_gather(self.embed, ids, h16, T, hid, c.vocab_size)
            x = h16.float()  # fp32 residual stream
        q16 = torch.empty((T, H * D), dtype=self.dtype, device=dev)
        q16l = torch.empty((T, H * D), dtype=self.dtype, device=dev)       # low-order plane of the split-precision q
        ws_bytes = n.attn_workspace_bytes(B, H, D, q_len, past_len + q_len)
        ws = torch.empty(max(ws_bytes, 4) // 4, dtype=torch.float32, device=dev)
        tail = self._tail_for(arena, past_dev)

        def planes(k):
            return (torch.empty((mt, k // 32, 64, 8), dtype=self.dtype, device=dev),
                    torch.empty((mt, k // 32, 64, 8), dtype=self.dtype, device=dev))

        xh, xl = planes(hid)
        ah, al = planes(H * D)
        ch, cl = planes(inter)
        # The two N = hidden projections (o_proj, down_proj) split K over KQ workgroup slices and leave KQ slabs of
        # partial sums; the next RMSNorm launch folds them into the residual stream (x += sum of slabs).
        KQ = self.rows_kslices(T, hid)
        slabs = torch.empty((KQ, T, hid), dtype=torch.float32, device=dev)
        # ... and so does q|k|v there: its 128-column panels are half as many as the CUs, so K is cut in two and the rotation /
        # append runs over the two slabs (pc_gemm: q|k|v epilogue with kslices = 2)
        QS = int(os.environ.get("PC_ROWS_QKV_KS", "2")) if (self.SKINNY_MAX_ROWS < T <= 288 and KQ != self.kslices and lw0_fp16(self.layers)) else 1
        qkv_slabs = torch.empty((QS, T, W), dtype=torch.float32, device=dev) if QS > 1 else None
        pending = 0                                   # slabs waiting to be added to x
        layers = self.layers if num_layers is None else self.layers[:num_layers]
        if T <= (16 if self.int8_weights else self.NORM_FUSED_MAX_ROWS) and self.fuse_norm:   # (int8 weight images: one row tile)
            return self._layers_norm_fused(x, cs, q16, q16l, ws, ah, al, ch, cl, arena, layers, B, q_len, past_len, past_dev,
                                           last_token_only, tail)
        for li, lw in enumerate(layers):
            n.rmsnorm_frag(x, lw["ln1"], xh, xl, T, hid, eps, slabs, pending)
            kp, vp = arena.k_plane(li), arena.v_plane(li)
            kvlo, lo_base = tail(li)
            # q|k|v projection + RoPE + in-place KV append in one weight-streaming launch
            n.gemm_qkv_rope(lw["wqkv_f"], xh, xl, T, hid, cs, q16, q16l, H * D, kp, vp, arena.batch_stride,
                            arena.head_stride, B, H, Hkv, D, q_len, past_len, arena.cap, past_dev, kv_lo=kvlo and kvlo[:4],
                            wscale=lw["wqkv_s"], lo_base=lo_base, kslices=QS, scratch=qkv_slabs)
            if self._kv_only and li == len(layers) - 1:
                return None       # schema encode: the K / V of the last layer are written; nothing after them is used
            n.attn_fwd(q16, q_len * H * D, H * D, kp, vp, arena.batch_stride, arena.head_stride, None, 0, 0,
                       B, H, Hkv, D, q_len, past_len, self.softmax_scale, ws, past_len_dev=past_dev, out_frag=(ah, al),
                       q_lo=q16l, kv_lo=kvlo, counters=self._counters_for(B, H),
                       gather=None if self._gather is None else (self._gather, li * 2 * Hkv, (li * 2 + 1) * Hkv))
            mid_skip = self.mid_lo_skip if (T > self.SKINNY_MAX_ROWS and self._lo_mode != 3) else ()      # (precision audit: tools/plane_audit.py)
            n.gemm_skinny(lw["wo_f"], ah, None if "o" in mid_skip else al, T, hid, H * D, n.EPI_STORE, y=slabs, ldy=hid, kslices=KQ,
                          wscale=lw["wo_s"])                                                    # attn @ Wo^T
            n.rmsnorm_frag(x, lw["ln2"], xh, xl, T, hid, eps, slabs, KQ)                       # x += ...; norm
            # (an encode pass on this stack follows the many-row stack's plane policy: dense_lo_skip; PC_MID_LO_SKIP=gu extends it to
            # the long-question forward)
            gu_lo = None if (T > self.SKINNY_MAX_ROWS and ((self._lo_mode == 3 and "gu" in self.dense_lo_skip) or
                                                           (self._lo_mode != 3 and "gu" in self.mid_lo_skip))) else xl
            n.gemm_skinny(lw["wgu_f"], xh, gu_lo, T, 2 * inter, hid, n.EPI_SILU, of_hi=ch, of_lo=cl, wscale=lw["wgu_s"])  # silu(g)*u
            n.gemm_skinny(lw["wdown_f"], ch, None if "down" in mid_skip else cl, T, hid, inter, n.EPI_STORE, y=slabs, ldy=hid, kslices=KQ,
                          wscale=lw["wdown_s"])                                                 # act @ Wd^T
            pending = KQ
        V = c.vocab_size
        if last_token_only:
            if pending:
                x.add_(slabs.sum(dim=0))
            xlast = x.view(B, q_len, hid)[:, -1, :].contiguous()
            lh, ll = planes(hid)
            n.rmsnorm_frag(xlast, self.norm, lh, ll, B, hid, eps)
            logits = torch.empty((B, V), dtype=torch.float32, device=dev)
            n.gemm_skinny(self.lm_head_f, lh, ll, B, V, hid, n.EPI_STORE, y=logits, ldy=V)
            return logits.view(B, 1, V)
        n.rmsnorm_frag(x, self.norm, xh, xl, T, hid, eps, slabs, pending)
        logits = torch.empty((T, V), dtype=torch.float32, device=dev)
        n.gemm_skinny(self.lm_head_f, xh, xl, T, V, hid, n.EPI_STORE, y=logits, ldy=V)          # llama2.py:1050-1051
        return logits.view(B, q_len, V)
