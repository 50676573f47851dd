"""MPT-7b-class forward pass on MI355X (the reference's ``Mpt`` adapter: ``promptcache/model/mpt.py``).

What differs from the Llama stack (``llama_hip.py``):

* no rotary embedding: positions enter through ALiBi only.  The reference gathers the bias row at the POSITION IDS of
  all keys, cached and new (``mpt.py:172``; the adapter sets ``use_full_position_ids``, so ``lm()`` receives
  ``past_len + q_len`` position ids) -- ``score = q.k / sqrt(D) + slope[h] * (pos[key] - max_pos)``.  The
  ``-slope * max_pos`` part is constant along a softmax row, so the kernel adds ``slope[h] * pos[key]`` only
  (``pc_attn_fwd_alibi``);
* LayerNorm without bias (``:205-215``), GELU MLP (``:196-203``), fused ``Wqkv`` = [q | k | v] (``:143-147``).

The q|k|v projection reuses the fused projection + KV-append launch of the Llama path with an identity rotation table
(cos = 1, sin = 0).  Arena management, dispatch by row count and the hipGraph cache are inherited.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from .. import _native
from .config import MptShape
from .llama_hip import CausalLMOutput, LlamaHIP, prime_graph_capture

_LOG2E = 1.4426950408889634


def alibi_slopes(num_heads: int, alibi_bias_max: int = 8) -> torch.Tensor:
    """``build_mpt_alibi_tensor`` (mpt.py:98-104): 1 / 2**(i * bias_max / n), i = 1..n, fp32.  Power-of-two head counts
    only -- the reference's own table raises for anything else (its ``view`` at :104)."""
    if num_heads & (num_heads - 1):
        raise ValueError(f"MPT ALiBi slopes need a power-of-two head count (got {num_heads}): mpt.py:104")
    base = torch.arange(1, num_heads + 1, dtype=torch.float32) * (alibi_bias_max / num_heads)
    return 1.0 / torch.pow(2, base)


class MptHIP(LlamaHIP):
    _shared_prefix_loop = False     # (its many-row loop keeps a copy of the trunk per batch row)
    supports_fused_gather = False   # (its layer loops read staged rows from the arena: PromptCache.update copies at once)

    def __init__(self, shape: MptShape, weights: Dict[str, torch.Tensor], device="cuda:0", decode_headroom: int = 256,
                 skinny: bool = True, int8_weights: bool = False):
        self._setup(shape, device, decode_headroom)
        c = shape
        self.H = self.Hkv = c.num_attention_heads
        self.D, self.L = c.head_dim, c.num_hidden_layers
        dev = self.device

        def w(name):
            t = weights[name]
            if not isinstance(t, torch.Tensor):
                t = torch.from_numpy(t)
            return t.to(device=dev, dtype=self.dtype).contiguous()

        self.embed = w("embed")
        self.lnf = w("lnf")
        self.lm_head = self.embed if c.tie_word_embeddings and "lm_head" not in weights else w("lm_head")
        hid = c.hidden_size
        self.skinny = bool(skinny) and hid % 32 == 0 and c.vocab_size % 16 == 0 and self.D % 16 == 0
        fr = _native.to_weight_frags if self.skinny else (lambda t: None)
        self.lm_head_f = fr(self.lm_head)
        self.layers = []
        # load_in_8bit: weight-only int8 decoder linears, see LlamaHIP (every GEMM K must be a multiple of 64)
        self.int8_weights = bool(int8_weights) and self.skinny and hid % 64 == 0
        if self.int8_weights:
            self.MID_MAX_ROWS = self.SKINNY_MAX_ROWS
        for i in range(self.L):
            if self.skinny and i == 0:
                self._qkv_perm = _native.qkv_rope_row_perm(3 * self.H, self.D).to(dev)
            ent = dict(ln1=w(f"l{i}.ln1"), ln2=w(f"l{i}.ln2"))
            ent.update(self._linear_entries("wqkv", self._prep_linear(w(f"l{i}.wqkv"), self._qkv_perm if self.skinny else None)))
            for name in ("wo", "w1", "w2"):
                ent.update(self._linear_entries(name, self._prep_linear(w(f"l{i}.{name}"))))
            self.layers.append(ent)
        self.slopes_log2 = (alibi_slopes(self.H, c.alibi_bias_max) * _LOG2E).to(dev)
        self.inv_freq_cpu = torch.zeros(1)                      # no rotary table (kept for interface symmetry)
        self.softmax_scale = 1.0 / math.sqrt(self.D)            # mpt.py:139-140
        self.fuse_norm = False       # LayerNorm is not a per-row scale: no norm folding into the projections
        self.supports_ragged_past = False   # ALiBi takes one position row per batch row, laid out for ONE past length
        self.supports_greedy_loop = False   # every step re-bases the per-key position row on the host (see __call__)

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def _kpos_cols(cap: int) -> int:
        return (cap + 63) // 64 * 64 + 64       # the kernel reads whole 64-key tiles of positions

    @torch.inference_mode()
    def __call__(self, input_ids: torch.Tensor, position_ids: Optional[torch.Tensor] = None, past_key_values=None,
                 attention_mask: Optional[torch.Tensor] = None, use_cache: bool = True, last_token_only: bool = False,
                 num_layers: Optional[int] = None, many_rows: bool = False, kv_only: bool = False,
                 **_unused) -> CausalLMOutput:
        dev = self.device
        self._kv_only = bool(kv_only)
        input_ids = input_ids.to(dev)
        B, q_len = input_ids.shape
        arena, past_len = self._resolve_arena(past_key_values, B, q_len)
        if many_rows and self.precise_dense and past_key_values is None:
            arena.with_lo()
        kv_len = past_len + q_len
        if position_ids is None:
            if past_len:
                raise ValueError("MPT needs the position id of every cached key (use_full_position_ids, mpt.py:172)")
            position_ids = torch.arange(q_len, device=dev).unsqueeze(0).expand(B, q_len)
        position_ids = position_ids.to(dev).reshape(B, -1)
        if position_ids.shape[1] != kv_len:
            raise ValueError(f"MPT got {position_ids.shape[1]} position ids for {kv_len} keys: pass the full position ids "
                             "(CacheEngine.process(..., return_full_position_ids=True), mpt.py:172)")
        if attention_mask is not None:
            # (checked where the mask LIVES: a host mask -- what CacheEngine passes since round 5 -- costs no upload and no
            # device sync; a device mask, the reference's convention (cache_engine.py:246), is read back for the test: one
            # pipeline drain per forward, which is what made 10 % of the schema encode's wall time idle in rounds 1-4)
            am = attention_mask
            if am.dim() == 2 and am.shape[1] == q_len and bool((am[:, 1:] > am[:, :-1]).any()):
                raise NotImplementedError("left / interior padding masks are not supported by the HIP path")
        T = B * q_len
        ids = input_ids.reshape(-1).to(torch.int64).contiguous()
        # like the reference (mpt.py:97: arange(1 - max_len, 1)) the bias is slope * (pos - max_pos) <= 0: the keys that
        # carry the weight sit near max_pos, where the term is small and fp32 resolves it finely
        position_ids = position_ids - position_ids.amax(dim=1, keepdim=True)
        # (same routing as LlamaHIP.__call__: encode passes always take the many-row path, only the serving prefill and
        # the decode steps capture hipGraphs)
        encode_pass = many_rows and self.precise_dense and (past_key_values is None or arena.lo is not None)
        graphed = self.skinny and T <= self.SKINNY_MAX_ROWS and self.use_graphs and not many_rows and not kv_only
        mid = self.skinny and not encode_pass and T <= self.MID_MAX_ROWS and \
            not (many_rows and self.precise_dense and T > self.SKINNY_MAX_ROWS)
        self._lo_mode = self._tail_mode(arena, q_len, past_len) if (graphed or mid) else 0
        if graphed:
            logits = self._graphed_mpt(ids, position_ids, arena, B, q_len, past_len, last_token_only, num_layers)
        else:
            kpos = torch.zeros((B, self._kpos_cols(arena.cap)), dtype=torch.float32, device=dev)
            kpos[:, :kv_len] = position_ids.to(torch.float32)
            if mid:
                logits = self._forward_skinny(ids, kpos, None, arena, B, q_len, past_len, last_token_only, num_layers)
            else:
                logits = self._forward_dense(ids, kpos, arena, B, q_len, past_len, last_token_only, num_layers)
        arena.length = kv_len
        self._tail_done(arena, self._lo_mode, q_len, past_len)
        return CausalLMOutput(logits=logits, past_key_values=arena.views() if use_cache else None)

    def _graphed_mpt(self, ids, position_ids, arena, B, q_len, past_len, last_token_only, num_layers):
        n = _native
        kv_len = past_len + q_len
        nsplit_key = n.attn_workspace_bytes(B, self.H, self.D, q_len, kv_len)
        mode = self._lo_mode
        key = (B, q_len, arena.buf.data_ptr(), arena.cap, nsplit_key, bool(last_token_only), num_layers, mode,
               arena.tail_lo.data_ptr() if mode else 0, arena.tail_lo.shape[4] if mode else 0)
        ent = self._graphs.pop(key, None)
        fresh = ent is None
        if not fresh:
            self._graphs[key] = ent                      # LRU: a hit moves the entry to the young end
        else:
            if len(self._graphs) >= self.max_graphs:
                self._graphs.pop(next(iter(self._graphs)))   # evict the least recently used
            st_ids = torch.zeros(B * q_len, dtype=torch.int64, device=self.device)
            st_kpos = torch.zeros((B, self._kpos_cols(arena.cap)), dtype=torch.float32, device=self.device)
            st_past = torch.zeros(2, dtype=torch.int32, device=self.device)      # {past_len, base of the residual tail}
            ent = [None, st_ids, st_kpos, st_past, None]
        _, st_ids, st_kpos, st_past, out = ent
        st_ids.copy_(ids)
        st_kpos[:, :kv_len].copy_(position_ids)           # int64 -> fp32 on the fly
        st_past[0:1].fill_(past_len)
        if mode == 2:
            st_past[1:2].fill_(arena.tail_base)
        if fresh:
            self._forward_skinny(st_ids, st_kpos, st_past, arena, B, q_len, past_len, last_token_only, num_layers)
            torch.cuda.synchronize()
            prime_graph_capture(self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = self._forward_skinny(st_ids, st_kpos, st_past, arena, B, q_len, past_len, last_token_only, num_layers)
            ent[0], ent[4] = g, out
            self._graphs[key] = ent
        ent[0].replay()
        return ent[4].clone()

    def _identity_rotation(self, T: int) -> torch.Tensor:
        cs = torch.zeros((T, self.D // 2, 2), dtype=torch.float32, device=self.device)
        cs[..., 0] = 1.0                                        # cos = 1, sin = 0: q and k pass through unrotated
        return cs

    # ------------------------------------------------------------------------------------------
    def _forward_dense(self, ids, kpos, arena, B, q_len, past_len, last_token_only, num_layers):
        """Many-row path (schema encode / no-cache prefill): Wqkv, out_proj + residual, up_proj + GELU, down_proj +
        residual on pc_gemm_dense; split-precision activations unless PC_FAST_DENSE=1 (see LlamaHIP._forward_dense)."""
        n = _native
        dev = self.device
        c = self.config
        two = self.precise_dense
        H, D, hid = self.H, self.D, c.hidden_size
        T = B * q_len
        W = 3 * hid
        eps = c.layer_norm_epsilon
        f32 = torch.float32
        cs = self._identity_rotation(T)
        h2 = torch.empty((2, T, hid), dtype=self.dtype, device=dev)
        n.embed_gather(self.embed, ids, h2[0], T, hid, c.vocab_size)
        x = h2[0].float()
        attn2 = torch.empty((2, T, hid), dtype=self.dtype, device=dev)
        act2 = torch.empty((2, T, 4 * hid), dtype=self.dtype, device=dev)
        q16 = torch.empty((T, hid), dtype=self.dtype, device=dev)
        q16l = torch.empty((T, hid), dtype=self.dtype, device=dev) if two else None
        qkv = torch.empty((T, W), dtype=f32, device=dev)
        lo_for, full_lo = self._dense_pass_lo(arena, B, H, q_len, past_len) if two else ((lambda li: None), False)
        ws = self._workspace(n.attn_workspace_bytes(B, H, D, q_len, past_len + q_len))
        alibi = (kpos, self.slopes_log2)
        lo = (lambda t: t[1]) if two else (lambda t: None)
        layers = self.layers if num_layers is None else self.layers[:num_layers]

        def norm(src, gw, rows):
            if two:
                n.layernorm_split(src, gw, None, h2[0], h2[1], rows, hid, eps)
            else:
                n.layernorm(src, gw, None, h2[0], rows, hid, eps)

        for li, lw in enumerate(layers):
            norm(x, lw["ln1"], T)                                                               # mpt.py:240
            self._proj(h2[0], lo(h2), lw, "wqkv", T, W, hid, n.EPI_STORE, y=qkv)                # [T, 3*hid]  :143
            kp, vp = arena.k_plane(li), arena.v_plane(li)
            kv_lo = lo_for(li)
            n.rope_append(qkv, q_len * W, W, q16, q_len * hid, hid, qkv[:, hid:], qkv[:, 2 * hid:], q_len * W, W,
                          kp, vp, arena.batch_stride, arena.head_stride, cs, B, H, H, D, q_len, past_len, arena.cap, True,
                          q_out_lo=q16l, kv_lo=kv_lo)
            if self._kv_only and li == len(layers) - 1:
                break             # schema encode: the K / V of the last layer are written; nothing after them is used
            n.attn_fwd(q16, q_len * hid, hid, kp, vp, arena.batch_stride, arena.head_stride, attn2[0],
                       q_len * hid, hid, B, H, H, D, q_len, past_len, self.softmax_scale, ws, alibi=alibi, q_lo=q16l,
                       out_lo=lo(attn2), kv_lo=kv_lo)
            self._proj(attn2[0], lo(attn2), lw, "wo", T, hid, hid, n.EPI_ADD, y=x)               # :185, :254
            norm(x, lw["ln2"], T)                                                               # :256
            self._proj(h2[0], lo(h2), lw, "w1", T, 4 * hid, hid, n.EPI_GELU, out_hi=act2[0], out_lo=lo(act2))
            self._proj(act2[0], lo(act2), lw, "w2", T, hid, 4 * hid, n.EPI_ADD, y=x)            # :197-201
        if full_lo:
            arena.lo_len = past_len + q_len
        if self._kv_only:
            return None
        head = {"lm_head": self.lm_head}
        V = c.vocab_size
        if last_token_only:
            xl = x.view(B, q_len, hid)[:, -1, :].contiguous()
            norm(xl, self.lnf, B)
            logits = torch.empty((B, V), dtype=f32, device=dev)
            self._proj(h2[0, :B], h2[1, :B] if two else None, head, "lm_head", B, V, hid, n.EPI_STORE, y=logits)
            return logits.view(B, 1, V)
        norm(x, self.lnf, T)
        logits = torch.empty((T, V), dtype=f32, device=dev)
        self._proj(h2[0], lo(h2), head, "lm_head", T, V, hid, n.EPI_STORE, y=logits)
        return logits.view(B, q_len, V)

    def _forward_skinny(self, ids, kpos, past_dev, arena, B, q_len, past_len, last_token_only, num_layers):
        n = _native
        dev = self.device
        c = self.config
        H, D, hid = self.H, self.D, c.hidden_size
        inter = 4 * hid
        T = B * q_len
        eps = c.layer_norm_epsilon
        mt = (T + 15) // 16
        f32 = torch.float32
        cs = self._identity_rotation(T)
        h16 = torch.empty((T, hid), dtype=self.dtype, device=dev)
        n.embed_gather(self.embed, ids, h16, T, hid, c.vocab_size)
        x = h16.float()
        q16 = torch.empty((T, hid), dtype=self.dtype, device=dev)
        q16l = torch.empty((T, hid), dtype=self.dtype, device=dev)
        ws_bytes = n.attn_workspace_bytes(B, H, D, q_len, past_len + q_len)
        ws = torch.empty(max(ws_bytes, 4) // 4, dtype=f32, device=dev)
        alibi = (kpos, self.slopes_log2)

        def planes(k):
            return (torch.empty((mt, k // 32, 64, 8), dtype=self.dtype, device=dev),
                    torch.empty((mt, k // 32, 64, 8), dtype=self.dtype, device=dev))

        xh, xl = planes(hid)
        ah, al = planes(hid)
        ch, cl = planes(inter)
        KQ = self.rows_kslices(T, hid)
        slabs = torch.empty((KQ, T, hid), dtype=f32, device=dev)
        pending = 0
        tail = self._tail_for(arena, past_dev)
        layers = self.layers if num_layers is None else self.layers[:num_layers]
        for li, lw in enumerate(layers):
            n.layernorm_frag(x, lw["ln1"], None, xh, xl, T, hid, eps, slabs, pending)
            kp, vp = arena.k_plane(li), arena.v_plane(li)
            kvlo, lo_base = tail(li)
            n.gemm_qkv_rope(lw["wqkv_f"], xh, xl, T, hid, cs, q16, q16l, hid, kp, vp, arena.batch_stride,
                            arena.head_stride, B, H, H, D, q_len, past_len, arena.cap, past_dev, kv_lo=kvlo and kvlo[:4],
                            lo_base=lo_base, wscale=lw["wqkv_s"])
            n.attn_fwd(q16, q_len * hid, hid, kp, vp, arena.batch_stride, arena.head_stride, None, 0, 0,
                       B, H, H, D, q_len, past_len, self.softmax_scale, ws, past_len_dev=past_dev, out_frag=(ah, al),
                       q_lo=q16l, alibi=alibi, kv_lo=kvlo)
            n.gemm_skinny(lw["wo_f"], ah, al, T, hid, hid, n.EPI_STORE, y=slabs, ldy=hid, kslices=KQ, wscale=lw["wo_s"])
            n.layernorm_frag(x, lw["ln2"], None, xh, xl, T, hid, eps, slabs, KQ)
            n.gemm_skinny(lw["w1_f"], xh, xl, T, inter, hid, n.EPI_GELU, of_hi=ch, of_lo=cl, wscale=lw["w1_s"])
            n.gemm_skinny(lw["w2_f"], ch, cl, T, hid, inter, n.EPI_STORE, y=slabs, ldy=hid, kslices=KQ, wscale=lw["w2_s"])
            pending = KQ
        V = c.vocab_size
        if last_token_only:
            if pending:
                x.add_(slabs.sum(dim=0))
            xlast = x.view(B, q_len, hid)[:, -1, :].contiguous()
            lh, ll = planes(hid)
            n.layernorm_frag(xlast, self.lnf, None, lh, ll, B, hid, eps)
            logits = torch.empty((B, V), dtype=f32, device=dev)
            n.gemm_skinny(self.lm_head_f, lh, ll, B, V, hid, n.EPI_STORE, y=logits, ldy=V)
            return logits.view(B, 1, V)
        n.layernorm_frag(x, self.lnf, None, xh, xl, T, hid, eps, slabs, pending)
        logits = torch.empty((T, V), dtype=f32, device=dev)
        n.gemm_skinny(self.lm_head_f, xh, xl, T, V, hid, n.EPI_STORE, y=logits, ldy=V)
        return logits.view(B, q_len, V)
