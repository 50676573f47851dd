"""Model shape descriptions for the Llama-2-class adapters.

The reference reads these from HF ``config.json`` via ``LlamaConfig`` (``promptcache/model/llama2.py:262-283``);
the shape constants it quotes for its system benchmarks are at ``eval_sys.py:92-100``.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, asdict


@dataclass
class LlamaShape:
    vocab_size: int = 32000
    hidden_size: int = 4096
    intermediate_size: int = 11008
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    num_key_value_heads: int = 32
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    max_position_embeddings: int = 4096
    initializer_range: float = 0.02
    name: str = "llama"

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def kv_bytes_per_token(self) -> int:
        """fp16 K+V bytes per token over all layers (7b = 524288)."""
        return 2 * self.num_hidden_layers * self.num_key_value_heads * self.head_dim * 2

    def to_dict(self):
        return asdict(self)

    @classmethod
    def from_hf_dir(cls, path: str) -> "LlamaShape":
        with open(os.path.join(path, "config.json")) as f:
            c = json.load(f)
        return cls(
            vocab_size=c["vocab_size"], hidden_size=c["hidden_size"],
            intermediate_size=c["intermediate_size"], num_hidden_layers=c["num_hidden_layers"],
            num_attention_heads=c["num_attention_heads"],
            num_key_value_heads=c.get("num_key_value_heads", c["num_attention_heads"]),
            rms_norm_eps=c.get("rms_norm_eps", 1e-5), rope_theta=c.get("rope_theta", 10000.0),
            max_position_embeddings=c.get("max_position_embeddings", 4096),
            name=os.path.basename(os.path.normpath(path)))


@dataclass
class FalconShape:
    """Falcon-7b-class decoder (``promptcache/model/falcon.py``: multi-query attention with ONE shared K/V head,
    parallel attention + MLP behind a single LayerNorm, GELU MLP of width 4*hidden, rotary positions, no biases in
    the linears).  The reference reads these from HF ``config.json`` via ``FalconConfig``; its cache shape is
    ``(L, 1, head_dim)`` (``promptcache/model/__init__.py:256-258``)."""
    vocab_size: int = 65024
    hidden_size: int = 4544
    num_hidden_layers: int = 32
    num_attention_heads: int = 71
    layer_norm_epsilon: float = 1e-5
    rope_theta: float = 10000.0
    initializer_range: float = 0.02
    tie_word_embeddings: bool = False
    name: str = "falcon"

    num_key_value_heads = 1            # multi_query

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def intermediate_size(self) -> int:
        return 4 * self.hidden_size

    @property
    def kv_bytes_per_token(self) -> int:
        return 2 * self.num_hidden_layers * self.head_dim * 2

    def to_dict(self):
        return asdict(self)

    @classmethod
    def from_hf_dir(cls, path: str) -> "FalconShape":
        with open(os.path.join(path, "config.json")) as f:
            c = json.load(f)
        if c.get("new_decoder_architecture") or not c.get("multi_query", True) or not c.get("parallel_attn", True) \
                or c.get("alibi") or c.get("bias"):
            raise ValueError("only the falcon-7b architecture (multi_query, parallel_attn, rotary, no bias) is supported")
        return cls(vocab_size=c["vocab_size"], hidden_size=c["hidden_size"],
                   num_hidden_layers=c.get("num_hidden_layers", c.get("n_layer")),
                   num_attention_heads=c.get("num_attention_heads", c.get("n_head")),
                   layer_norm_epsilon=c.get("layer_norm_epsilon", 1e-5), rope_theta=c.get("rope_theta", 10000.0),
                   tie_word_embeddings=c.get("tie_word_embeddings", True),
                   name=os.path.basename(os.path.normpath(path)))


FALCON_SHAPES = {
    "falcon-tiny": FalconShape(vocab_size=1024, hidden_size=128, num_hidden_layers=2, num_attention_heads=4, name="falcon-tiny"),
    # head_dim 64 like falcon-7b, an odd head count like its 71
    "falcon-mid": FalconShape(vocab_size=2048, hidden_size=448, num_hidden_layers=2, num_attention_heads=7, name="falcon-mid"),
    "falcon-7b": FalconShape(name="falcon-7b"),
}

@dataclass
class MptShape:
    """MPT-7b-class decoder (``promptcache/model/mpt.py``): multi-head attention with ALiBi position biases (no
    rotary), fused ``Wqkv`` = [q | k | v], bias-free LayerNorms, GELU MLP of width 4*d_model, lm_head tied to the
    embedding.  The reference reads these from ``MptConfig``; its cache shape is ``(n_layers, n_heads, head_dim)``
    (``promptcache/model/__init__.py:284-286``)."""
    vocab_size: int = 50432
    hidden_size: int = 4096
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    layer_norm_epsilon: float = 1e-5
    alibi_bias_max: int = 8
    initializer_range: float = 0.02
    tie_word_embeddings: bool = False
    name: str = "mpt"

    @property
    def num_key_value_heads(self) -> int:
        return self.num_attention_heads

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def intermediate_size(self) -> int:
        return 4 * self.hidden_size

    @property
    def kv_bytes_per_token(self) -> int:
        return 2 * self.num_hidden_layers * self.num_attention_heads * self.head_dim * 2

    def to_dict(self):
        return asdict(self)

    @classmethod
    def from_hf_dir(cls, path: str) -> "MptShape":
        with open(os.path.join(path, "config.json")) as f:
            c = json.load(f)
        ac = c.get("attn_config", {})
        if not ac.get("alibi", True) or not c.get("no_bias", True) or ac.get("clip_qkv") or ac.get("qk_ln"):
            raise ValueError("only the mpt-7b architecture (alibi, no biases, no qkv clipping / qk layernorm) is supported")
        return cls(vocab_size=c["vocab_size"], hidden_size=c["d_model"], num_hidden_layers=c["n_layers"],
                   num_attention_heads=c["n_heads"], layer_norm_epsilon=c.get("layer_norm_epsilon", 1e-5),
                   alibi_bias_max=ac.get("alibi_bias_max", 8), tie_word_embeddings=True,
                   name=os.path.basename(os.path.normpath(path)))


MPT_SHAPES = {
    "mpt-tiny": MptShape(vocab_size=1024, hidden_size=128, num_hidden_layers=2, num_attention_heads=4, name="mpt-tiny"),
    # head_dim 64.  (Head counts must be powers of two: the reference's slope table raises otherwise, mpt.py:104.)
    "mpt-mid": MptShape(vocab_size=2048, hidden_size=512, num_hidden_layers=2, num_attention_heads=8, name="mpt-mid"),
    "mpt-7b": MptShape(name="mpt-7b"),
}

SHAPES = {
    # test-sized
    "tiny": LlamaShape(vocab_size=1024, hidden_size=128, intermediate_size=344, num_hidden_layers=2,
                       num_attention_heads=4, num_key_value_heads=4, rms_norm_eps=1e-6, name="tiny"),
    # D=128 like the real models (MHA: the reference's cache engine cannot stage GQA, its
    # get_cache_shape returns num_attention_heads, promptcache/model/__init__.py:110-114)
    "mid": LlamaShape(vocab_size=2048, hidden_size=512, intermediate_size=1376, num_hidden_layers=3,
                      num_attention_heads=4, num_key_value_heads=4, rms_norm_eps=1e-5, name="mid"),
    # every GEMM K a multiple of 64 (what the int8-weight images need), D = 128, MHA and GQA
    "mid64": LlamaShape(vocab_size=2048, hidden_size=512, intermediate_size=1408, num_hidden_layers=3,
                        num_attention_heads=4, num_key_value_heads=4, rms_norm_eps=1e-5, name="mid64"),
    "mid64_gqa": LlamaShape(vocab_size=2048, hidden_size=512, intermediate_size=1408, num_hidden_layers=3,
                            num_attention_heads=4, num_key_value_heads=2, rms_norm_eps=1e-5, name="mid64_gqa"),
    # GQA 2:1 so the head-broadcast index math is exercised (oracle-checked only)
    "mid_gqa": LlamaShape(vocab_size=2048, hidden_size=512, intermediate_size=1376, num_hidden_layers=3,
                          num_attention_heads=4, num_key_value_heads=2, rms_norm_eps=1e-5, name="mid_gqa"),
    # D=128, MHA, 2 layers at hidden 256 (cheap full-path checks)
    "mid_mha": LlamaShape(vocab_size=2048, hidden_size=256, intermediate_size=688, num_hidden_layers=2,
                          num_attention_heads=2, num_key_value_heads=2, rms_norm_eps=1e-5, name="mid_mha"),
    # BASELINE.json configs
    "llama2-7b": LlamaShape(name="llama2-7b"),
    "llama2-13b": LlamaShape(hidden_size=5120, intermediate_size=13824, num_hidden_layers=40,
                             num_attention_heads=40, num_key_value_heads=40, name="llama2-13b"),
    "codellama-7b": LlamaShape(vocab_size=32016, rope_theta=1e6, max_position_embeddings=16384,
                               name="codellama-7b"),
}
