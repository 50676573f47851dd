"""Falcon-7b-class forward pass on MI355X (the reference's ``Falcon`` adapter: ``promptcache/model/falcon.py`` with
``multi_query``, ``parallel_attn``, rotary positions, no linear biases).

What differs from the Llama stack (``llama_hip.py``), per ``FalconDecoderLayer.forward`` (falcon.py:761-816):

* ONE LayerNorm per layer feeds both the attention and the MLP (``parallel_attn``, :797-798), and the layer output is
  ``x + attn + mlp`` (:810-813);
* multi-query attention: the fused ``query_key_value`` projection yields H query heads and a single K and V head
  (:393-396), so the module KV the cache engine stores / gathers is ``[L][2][1][len][D]`` -- 1/H of the Llama volume
  per token (``Falcon.get_cache_shape``, ``promptcache/model/__init__.py:256-258``);
* the MLP is ``dense_4h_to_h(gelu(dense_h_to_4h(x)))`` (:730-733).

Arena management, the hipGraph cache and the dispatch by row count are inherited; the kernels are the same ones
(``pc_kv_gather`` with Hkv = 1, ``pc_gemm_qkv_rope`` with H + 2 heads, ``pc_attn_fwd`` with a group size of H) plus
``pc_layernorm`` / ``pc_layernorm_frag`` / ``pc_gelu`` and the GELU GEMM epilogue.
"""
from __future__ import annotations

import math
from typing import Dict

import torch

from .. import _native
from .config import FalconShape
from .llama_hip import LlamaHIP


class FalconHIP(LlamaHIP):
    _shared_prefix_loop = False     # (its many-row loop keeps a copy of the trunk per batch row)
    supports_fused_gather = True    # (the weight-streaming loop hands pc_attn the row table: the first forward stages, as in LlamaHIP)

    def __init__(self, shape: FalconShape, weights: Dict[str, torch.Tensor], device="cuda:0", decode_headroom: int = 256,
                 skinny: bool = True, int8_weights: bool = False):
        self._setup(shape, device, decode_headroom)
        c = shape
        self.H, self.Hkv, self.D, self.L = c.num_attention_heads, 1, c.head_dim, c.num_hidden_layers
        dev = self.device

        def w(name):
            t = weights[name]
            if not isinstance(t, torch.Tensor):
                t = torch.from_numpy(t)
            return t.to(device=dev, dtype=self.dtype).contiguous()

        self.embed = w("embed")
        self.lnf_w, self.lnf_b = w("lnf_w"), w("lnf_b")
        self.lm_head = self.embed if c.tie_word_embeddings and "lm_head" not in weights else w("lm_head")
        hid = c.hidden_size
        self.skinny = bool(skinny) and hid % 32 == 0 and c.vocab_size % 16 == 0 and ((self.H + 2) * self.D) % 16 == 0
        fr = _native.to_weight_frags if self.skinny else (lambda t: None)
        self.lm_head_f = fr(self.lm_head)
        self.layers = []
        # load_in_8bit: weight-only int8 decoder linears, see LlamaHIP (every GEMM K must be a multiple of 64)
        self.int8_weights = bool(int8_weights) and self.skinny and hid % 64 == 0 and (self.H * self.D) % 64 == 0
        if self.int8_weights:
            self.MID_MAX_ROWS = self.SKINNY_MAX_ROWS
        for i in range(self.L):
            if self.skinny and i == 0:
                self._qkv_perm = _native.qkv_rope_row_perm(self.H + 2, self.D).to(dev)
            ent = dict(ln_w=w(f"l{i}.ln_w"), ln_b=w(f"l{i}.ln_b"))
            ent.update(self._linear_entries("wqkv", self._prep_linear(w(f"l{i}.wqkv"), self._qkv_perm if self.skinny else None)))
            for name in ("wo", "w1", "w2"):
                ent.update(self._linear_entries(name, self._prep_linear(w(f"l{i}.{name}"))))
            self.layers.append(ent)
        # falcon.py:99: the same formula as the Llama table, evaluated on the CPU in fp32
        self.inv_freq_cpu = 1.0 / (c.rope_theta ** (torch.arange(0, self.D, 2).float() / self.D))
        self.inv_freq = self.inv_freq_cpu.to(dev)
        self.softmax_scale = 1.0 / math.sqrt(self.D)        # inv_norm_factor, falcon.py:316
        self.fuse_norm = False       # LayerNorm is not a per-row scale: no norm folding into the projections

    # ------------------------------------------------------------------------------------------
    def _forward_dense(self, ids, pos32, arena, B, q_len, past_len, last_token_only, num_layers):
        """Many-row path (schema encode / no-cache prefill): the four projections of a Falcon block on pc_gemm_dense
        (q|k|v store, dense_h_to_4h + GELU, dense + residual, dense_4h_to_h + residual), split-precision activations
        unless PC_FAST_DENSE=1 (see LlamaHIP._forward_dense)."""
        n = _native
        dev = self.device
        c = self.config
        two = self.precise_dense
        H, D, hid = self.H, self.D, c.hidden_size
        T = B * q_len
        W = (H + 2) * D
        eps = c.layer_norm_epsilon
        f32 = torch.float32
        cs = torch.empty((T, D // 2, 2), dtype=f32, device=dev)
        n.rope_table(pos32, self.inv_freq, cs, T, D)
        h2 = torch.empty((2, T, hid), dtype=self.dtype, device=dev)
        n.embed_gather(self.embed, ids, h2[0], T, hid, c.vocab_size)
        x = h2[0].float()  # fp32 residual stream
        attn2 = torch.empty((2, T, H * D), dtype=self.dtype, device=dev)
        act2 = torch.empty((2, T, 4 * hid), dtype=self.dtype, device=dev)
        q16 = torch.empty((T, H * D), dtype=self.dtype, device=dev)
        q16l = torch.empty((T, H * D), dtype=self.dtype, device=dev) if two else None
        qkv = torch.empty((T, W), dtype=f32, device=dev)
        lo_for, full_lo = self._dense_pass_lo(arena, B, 1, q_len, past_len) if two else ((lambda li: None), False)
        ws = self._workspace(n.attn_workspace_bytes(B, H, D, q_len, past_len + q_len))
        lo = (lambda t: t[1]) if two else (lambda t: None)
        layers = self.layers if num_layers is None else self.layers[:num_layers]

        def norm(src, gw, gb, rows):
            if two:
                n.layernorm_split(src, gw, gb, h2[0], h2[1], rows, hid, eps)
            else:
                n.layernorm(src, gw, gb, h2[0], rows, hid, eps)

        for li, lw in enumerate(layers):
            norm(x, lw["ln_w"], lw["ln_b"], T)                                                     # falcon.py:779
            self._proj(h2[0], lo(h2), lw, "wqkv", T, W, hid, n.EPI_STORE, y=qkv)                   # [T, (H+2)*D]
            kp, vp = arena.k_plane(li), arena.v_plane(li)
            kv_lo = lo_for(li)
            n.rope_append(qkv, q_len * W, W, q16, q_len * H * D, H * D, qkv[:, H * D:], qkv[:, (H + 1) * D:], q_len * W, W,
                          kp, vp, arena.batch_stride, arena.head_stride, cs, B, H, 1, D, q_len, past_len, arena.cap, True,
                          q_out_lo=q16l, kv_lo=kv_lo, past_lens=self._past_lens)
            if self._kv_only and li == len(layers) - 1:
                break             # schema encode: the K / V of the last layer are written; nothing after them is used
            n.attn_fwd(q16, q_len * H * D, H * D, kp, vp, arena.batch_stride, arena.head_stride, attn2[0],
                       q_len * H * D, H * D, B, H, 1, D, q_len, past_len, self.softmax_scale, ws, q_lo=q16l,
                       out_lo=lo(attn2), kv_lo=kv_lo, past_lens=self._past_lens)
            # parallel attention + MLP on the same LayerNorm output (:798)
            self._proj(h2[0], lo(h2), lw, "w1", T, 4 * hid, hid, n.EPI_GELU, out_hi=act2[0], out_lo=lo(act2))
            self._proj(attn2[0], lo(attn2), lw, "wo", T, hid, H * D, n.EPI_ADD, y=x)
            self._proj(act2[0], lo(act2), lw, "w2", T, hid, 4 * hid, n.EPI_ADD, y=x)
        if full_lo:
            arena.lo_len = past_len + q_len
        if self._kv_only:
            return None
        head = {"lm_head": self.lm_head}
        V = c.vocab_size
        if last_token_only:
            xl = x.view(B, q_len, hid)[:, -1, :].contiguous()
            norm(xl, self.lnf_w, self.lnf_b, B)
            logits = torch.empty((B, V), dtype=f32, device=dev)
            self._proj(h2[0, :B], h2[1, :B] if two else None, head, "lm_head", B, V, hid, n.EPI_STORE, y=logits)
            return logits.view(B, 1, V)
        norm(x, self.lnf_w, self.lnf_b, T)
        logits = torch.empty((T, V), dtype=f32, device=dev)
        self._proj(h2[0], lo(h2), head, "lm_head", T, V, hid, n.EPI_STORE, y=logits)
        return logits.view(B, q_len, V)

    def _forward_skinny(self, ids, pos32, past_dev, arena, B, q_len, past_len, last_token_only, num_layers):
        """T <= 512 rows: weight-streaming projections (pc_gemm.hip).  The o_proj and dense_4h_to_h launches both leave
        K-sliced slabs of partial sums; the next layer's LayerNorm launch folds all of them into the residual stream."""
        n = _native
        dev = self.device
        c = self.config
        H, D, hid = self.H, self.D, c.hidden_size
        inter = 4 * hid
        T = B * q_len
        eps = c.layer_norm_epsilon
        mt = (T + 15) // 16
        f32 = torch.float32
        cs = torch.empty((T, D // 2, 2), dtype=f32, device=dev)
        n.rope_table(pos32, self.inv_freq, cs, T, D)
        h16 = torch.empty((T, hid), dtype=self.dtype, device=dev)
        n.embed_gather(self.embed, ids, h16, T, hid, c.vocab_size)
        x = h16.float()
        q16 = torch.empty((T, H * D), dtype=self.dtype, device=dev)
        q16l = torch.empty((T, H * D), dtype=self.dtype, device=dev)
        ws_bytes = n.attn_workspace_bytes(B, H, D, q_len, past_len + q_len)
        ws = torch.empty(max(ws_bytes, 4) // 4, dtype=f32, device=dev)

        def planes(k):
            return (torch.empty((mt, k // 32, 64, 8), dtype=self.dtype, device=dev),
                    torch.empty((mt, k // 32, 64, 8), dtype=self.dtype, device=dev))

        xh, xl = planes(hid)
        ah, al = planes(H * D)
        ch, cl = planes(inter)
        KQ = self.rows_kslices(T, hid)
        slabs = torch.empty((2 * KQ, T, hid), dtype=f32, device=dev)      # [0:KQ] attention branch, [KQ:] MLP branch
        pending = 0
        tail = self._tail_for(arena, past_dev)
        layers = self.layers if num_layers is None else self.layers[:num_layers]
        for li, lw in enumerate(layers):
            n.layernorm_frag(x, lw["ln_w"], lw["ln_b"], xh, xl, T, hid, eps, slabs, pending)
            kp, vp = arena.k_plane(li), arena.v_plane(li)
            kvlo, lo_base = tail(li)
            n.gemm_qkv_rope(lw["wqkv_f"], xh, xl, T, hid, cs, q16, q16l, H * D, kp, vp, arena.batch_stride,
                            arena.head_stride, B, H, 1, D, q_len, past_len, arena.cap, past_dev, kv_lo=kvlo and kvlo[:4],
                            lo_base=lo_base, wscale=lw["wqkv_s"])
            n.attn_fwd(q16, q_len * H * D, H * D, kp, vp, arena.batch_stride, arena.head_stride, None, 0, 0,
                       B, H, 1, D, q_len, past_len, self.softmax_scale, ws, past_len_dev=past_dev, out_frag=(ah, al),
                       q_lo=q16l, kv_lo=kvlo,
                       gather=None if self._gather is None else (self._gather, li * 2, li * 2 + 1))     # (one K/V head: planes 2 li, 2 li + 1)
            n.gemm_skinny(lw["wo_f"], ah, al, T, hid, H * D, n.EPI_STORE, y=slabs[:KQ], ldy=hid, kslices=KQ, wscale=lw["wo_s"])
            n.gemm_skinny(lw["w1_f"], xh, xl, T, inter, hid, n.EPI_GELU, of_hi=ch, of_lo=cl, wscale=lw["w1_s"])
            n.gemm_skinny(lw["w2_f"], ch, cl, T, hid, inter, n.EPI_STORE, y=slabs[KQ:], ldy=hid, kslices=KQ, wscale=lw["w2_s"])
            pending = 2 * KQ
        V = c.vocab_size
        if last_token_only:
            if pending:
                x.add_(slabs.sum(dim=0))
            xlast = x.view(B, q_len, hid)[:, -1, :].contiguous()
            lh, ll = planes(hid)
            n.layernorm_frag(xlast, self.lnf_w, self.lnf_b, lh, ll, B, hid, eps)
            logits = torch.empty((B, V), dtype=f32, device=dev)
            n.gemm_skinny(self.lm_head_f, lh, ll, B, V, hid, n.EPI_STORE, y=logits, ldy=V)
            return logits.view(B, 1, V)
        n.layernorm_frag(x, self.lnf_w, self.lnf_b, xh, xl, T, hid, eps, slabs, pending)
        logits = torch.empty((T, V), dtype=f32, device=dev)
        n.gemm_skinny(self.lm_head_f, xh, xl, T, V, hid, n.EPI_STORE, y=logits, ldy=V)
        return logits.view(B, q_len, V)
