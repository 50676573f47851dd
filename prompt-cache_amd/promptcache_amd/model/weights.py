"""Weight sources for the Llama-2-class adapter.

* ``make_weights_np``  -- seeded N(0, initializer_range) weights (the reference's ``_init_weights``,
  ``promptcache/model/llama2.py:686-695``) drawn with numpy's PCG64 so the same numbers can be
  produced in any process (tests, golden generation, the GPU box) and rounded to fp16.
* ``load_hf_safetensors`` -- real checkpoints (HF layout) when a directory is available.
* ``random_weights_device`` -- seeded weights drawn directly on the GPU at true shapes (bench).

Key names: ``embed``, ``l{i}.ln1|wq|wk|wv|wo|ln2|gate|up|down``, ``norm``, ``lm_head``;
linear weights are ``[out, in]`` (``nn.Linear`` layout).
"""
from __future__ import annotations

import glob
import os
from typing import Dict

import numpy as np

from .config import LlamaShape


def weight_shapes(cfg: LlamaShape) -> Dict[str, tuple]:
    hid, inter, D = cfg.hidden_size, cfg.intermediate_size, cfg.head_dim
    H, Hkv = cfg.num_attention_heads, cfg.num_key_value_heads
    shapes = {"embed": (cfg.vocab_size, hid)}
    for i in range(cfg.num_hidden_layers):
        shapes[f"l{i}.ln1"] = (hid,)
        shapes[f"l{i}.wq"] = (H * D, hid)
        shapes[f"l{i}.wk"] = (Hkv * D, hid)
        shapes[f"l{i}.wv"] = (Hkv * D, hid)
        shapes[f"l{i}.wo"] = (hid, H * D)
        shapes[f"l{i}.ln2"] = (hid,)
        shapes[f"l{i}.gate"] = (inter, hid)
        shapes[f"l{i}.up"] = (inter, hid)
        shapes[f"l{i}.down"] = (hid, inter)
    shapes["norm"] = (hid,)
    shapes["lm_head"] = (cfg.vocab_size, hid)
    return shapes


def make_weights_np(cfg: LlamaShape, seed: int = 0, scale: float = 1.0) -> Dict[str, np.ndarray]:
    """fp16 numpy weights.  Norm gains are 1 + 0.1*N(0,1) (not all-ones) so a test notices a
    dropped or misapplied gain.  ``scale`` multiplies the linear-weight std (tests use >1 to make
    the attention logits non-trivial at tiny hidden sizes)."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shp in weight_shapes(cfg).items():
        if len(shp) == 1:
            w = 1.0 + 0.1 * rng.standard_normal(shp, dtype=np.float32)
        else:
            w = (cfg.initializer_range * scale) * rng.standard_normal(shp, dtype=np.float32)
        out[name] = w.astype(np.float16)
    return out


def falcon_weight_shapes(cfg) -> Dict[str, tuple]:
    """Key names for the Falcon adapter: ``embed``, ``l{i}.ln_w|ln_b|wqkv|wo|w1|w2``, ``lnf_w``, ``lnf_b``, ``lm_head``;
    ``wqkv`` is the fused ``query_key_value`` matrix ``[(H + 2) * D, hidden]`` = H query heads, then the shared key
    head, then the shared value head (``falcon.py:393-396``)."""
    hid, D, H = cfg.hidden_size, cfg.head_dim, cfg.num_attention_heads
    shapes = {"embed": (cfg.vocab_size, hid)}
    for i in range(cfg.num_hidden_layers):
        shapes[f"l{i}.ln_w"] = (hid,)
        shapes[f"l{i}.ln_b"] = (hid,)
        shapes[f"l{i}.wqkv"] = ((H + 2) * D, hid)
        shapes[f"l{i}.wo"] = (hid, hid)
        shapes[f"l{i}.w1"] = (4 * hid, hid)
        shapes[f"l{i}.w2"] = (hid, 4 * hid)
    shapes["lnf_w"] = (hid,)
    shapes["lnf_b"] = (hid,)
    shapes["lm_head"] = (cfg.vocab_size, hid)
    return shapes


def make_falcon_weights_np(cfg, seed: int = 0, scale: float = 1.0) -> Dict[str, np.ndarray]:
    """fp16 numpy weights for a FalconShape: LayerNorm gains 1 + 0.1 N(0,1), LayerNorm biases 0.1 N(0,1)."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shp in falcon_weight_shapes(cfg).items():
        if name.endswith("_b"):
            w = 0.1 * rng.standard_normal(shp, dtype=np.float32)
        elif len(shp) == 1:
            w = 1.0 + 0.1 * rng.standard_normal(shp, dtype=np.float32)
        else:
            w = (cfg.initializer_range * scale) * rng.standard_normal(shp, dtype=np.float32)
        out[name] = w.astype(np.float16)
    if cfg.tie_word_embeddings:
        out["lm_head"] = out["embed"]
    return out


def mpt_weight_shapes(cfg) -> Dict[str, tuple]:
    """Key names for the MPT adapter: ``embed``, ``l{i}.ln1|wqkv|wo|ln2|w1|w2``, ``lnf``, ``lm_head``; ``wqkv`` is
    ``Wqkv`` [3*hid, hid] = all query heads, then all key heads, then all value heads (``mpt.py:144``)."""
    hid = cfg.hidden_size
    shapes = {"embed": (cfg.vocab_size, hid)}
    for i in range(cfg.num_hidden_layers):
        shapes[f"l{i}.ln1"] = (hid,)
        shapes[f"l{i}.wqkv"] = (3 * hid, hid)
        shapes[f"l{i}.wo"] = (hid, hid)
        shapes[f"l{i}.ln2"] = (hid,)
        shapes[f"l{i}.w1"] = (4 * hid, hid)
        shapes[f"l{i}.w2"] = (hid, 4 * hid)
    shapes["lnf"] = (hid,)
    shapes["lm_head"] = (cfg.vocab_size, hid)
    return shapes


def make_mpt_weights_np(cfg, seed: int = 0, scale: float = 1.0) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    out = {}
    for name, shp in mpt_weight_shapes(cfg).items():
        if len(shp) == 1:
            w = 1.0 + 0.1 * rng.standard_normal(shp, dtype=np.float32)
        else:
            w = (cfg.initializer_range * scale) * rng.standard_normal(shp, dtype=np.float32)
        out[name] = w.astype(np.float16)
    if cfg.tie_word_embeddings:
        out["lm_head"] = out["embed"]
    return out


def read_checkpoint_dir(path: str):
    """Every tensor of an HF checkpoint directory, by its HF name, on the host: the ``*.safetensors`` shards when
    there are any, else the ``pytorch_model*.bin`` shards (``torch.load(weights_only=True)``) that older uploads of
    the models the reference lists ship instead."""
    files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    raw = {}
    if files:
        from safetensors import safe_open
        for fn in files:
            with safe_open(fn, framework="pt", device="cpu") as f:
                for k in f.keys():
                    raw[k] = f.get_tensor(k)
        return raw
    bins = sorted(glob.glob(os.path.join(path, "pytorch_model*.bin")))
    if not bins:
        raise FileNotFoundError(f"no *.safetensors and no pytorch_model*.bin under {path}")
    import torch
    for fn in bins:
        raw.update(torch.load(fn, map_location="cpu", weights_only=True))
    return raw


def random_mpt_weights_device(cfg, device, dtype, seed: int = 0):
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = {}
    for name, shp in mpt_weight_shapes(cfg).items():
        if len(shp) == 1:
            w = 1.0 + 0.1 * torch.randn(shp, generator=g, device=device, dtype=torch.float32)
        else:
            w = torch.empty(shp, device=device, dtype=torch.float32).normal_(0.0, cfg.initializer_range, generator=g)
        out[name] = w.to(dtype)
        del w
    if cfg.tie_word_embeddings:
        out["lm_head"] = out["embed"]
    return out


def load_mpt_safetensors(path: str, cfg):
    """HF mpt-7b-class checkpoint directory -> the key layout of ``mpt_weight_shapes``."""
    raw = read_checkpoint_dir(path)
    out = {"embed": raw["transformer.wte.weight"], "lnf": raw["transformer.norm_f.weight"]}
    out["lm_head"] = raw.get("lm_head.weight", out["embed"])
    names = {"ln1": "norm_1.weight", "wqkv": "attn.Wqkv.weight", "wo": "attn.out_proj.weight", "ln2": "norm_2.weight",
             "w1": "ffn.up_proj.weight", "w2": "ffn.down_proj.weight"}
    for i in range(cfg.num_hidden_layers):
        for k, hf in names.items():
            out[f"l{i}.{k}"] = raw[f"transformer.blocks.{i}.{hf}"]
    return out


_HF_MAP = {
    "ln1": "input_layernorm.weight", "wq": "self_attn.q_proj.weight", "wk": "self_attn.k_proj.weight",
    "wv": "self_attn.v_proj.weight", "wo": "self_attn.o_proj.weight",
    "ln2": "post_attention_layernorm.weight", "gate": "mlp.gate_proj.weight",
    "up": "mlp.up_proj.weight", "down": "mlp.down_proj.weight",
}


def load_hf_safetensors(path: str, cfg: LlamaShape, device, dtype):
    """Read an HF Llama checkpoint directory (``*.safetensors``) into the key layout above."""
    raw = read_checkpoint_dir(path)
    out = {"embed": raw["model.embed_tokens.weight"], "norm": raw["model.norm.weight"],
           "lm_head": raw.get("lm_head.weight", raw["model.embed_tokens.weight"])}
    for i in range(cfg.num_hidden_layers):
        for short, hf in _HF_MAP.items():
            out[f"l{i}.{short}"] = raw[f"model.layers.{i}.{hf}"]
    return {k: v.to(device=device, dtype=dtype) for k, v in out.items()}


def random_weights_device(cfg: LlamaShape, device, dtype, seed: int = 0):
    """Seeded N(0, 0.02) weights generated on ``device`` (true-shape benchmarks: 7b = 13.5 GB fp16)."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = {}
    for name, shp in weight_shapes(cfg).items():
        if len(shp) == 1:
            w = 1.0 + 0.1 * torch.randn(shp, generator=g, device=device, dtype=torch.float32)
        else:
            w = torch.empty(shp, device=device, dtype=torch.float32).normal_(0.0, cfg.initializer_range, generator=g)
        out[name] = w.to(dtype)
        del w
    return out


def random_falcon_weights_device(cfg, device, dtype, seed: int = 0):
    """Seeded weights for a FalconShape generated on ``device`` (true-shape runs: falcon-7b = 14.4 GB fp16)."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = {}
    for name, shp in falcon_weight_shapes(cfg).items():
        if name.endswith("_b"):
            w = 0.1 * torch.randn(shp, generator=g, device=device, dtype=torch.float32)
        elif len(shp) == 1:
            w = 1.0 + 0.1 * torch.randn(shp, generator=g, device=device, dtype=torch.float32)
        else:
            w = torch.empty(shp, device=device, dtype=torch.float32).normal_(0.0, cfg.initializer_range, generator=g)
        out[name] = w.to(dtype)
        del w
    if cfg.tie_word_embeddings:
        out["lm_head"] = out["embed"]
    return out


def load_falcon_safetensors(path: str, cfg):
    """Read an HF falcon-7b-class checkpoint directory (``*.safetensors``) into the key layout of
    ``falcon_weight_shapes`` (HF names: ``transformer.h.{i}.self_attention.query_key_value`` etc.)."""
    raw = read_checkpoint_dir(path)
    out = {"embed": raw["transformer.word_embeddings.weight"], "lnf_w": raw["transformer.ln_f.weight"],
           "lnf_b": raw["transformer.ln_f.bias"]}
    out["lm_head"] = raw.get("lm_head.weight", out["embed"])
    names = {"ln_w": "input_layernorm.weight", "ln_b": "input_layernorm.bias", "wqkv": "self_attention.query_key_value.weight",
             "wo": "self_attention.dense.weight", "w1": "mlp.dense_h_to_4h.weight", "w2": "mlp.dense_4h_to_h.weight"}
    for i in range(cfg.num_hidden_layers):
        for k, hf in names.items():
            out[f"l{i}.{k}"] = raw[f"transformer.h.{i}.{hf}"]
    return out
