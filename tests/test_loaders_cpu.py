"""Round trips of the real-checkpoint path (CPU): a generated tiny HF checkpoint directory -- ``config.json`` plus the
tensors under their HF names, as ``*.safetensors`` shards or as ``pytorch_model*.bin`` -- must come back through
``<Shape>.from_hf_dir`` / ``load_*_safetensors`` as exactly the weight dict the adapters consume
(reference: ``AutoModelForCausalLM.from_pretrained`` call sites, promptcache/model/__init__.py:167-169, :209-211, :264-266)."""
import json
import os

import numpy as np
import pytest
import torch

from promptcache_amd.model import weights as W
from promptcache_amd.model.config import FalconShape, LlamaShape, MptShape


def _llama_dir(tmp, shape, w, shards=1, fmt="safetensors"):
    cfg = dict(vocab_size=shape.vocab_size, hidden_size=shape.hidden_size, intermediate_size=shape.intermediate_size,
               num_hidden_layers=shape.num_hidden_layers, num_attention_heads=shape.num_attention_heads,
               num_key_value_heads=shape.num_key_value_heads, rms_norm_eps=shape.rms_norm_eps, rope_theta=shape.rope_theta,
               max_position_embeddings=shape.max_position_embeddings)
    with open(os.path.join(tmp, "config.json"), "w") as f:
        json.dump(cfg, f)
    hf = {"model.embed_tokens.weight": w["embed"], "model.norm.weight": w["norm"], "lm_head.weight": w["lm_head"]}
    for i in range(shape.num_hidden_layers):
        for short, name in W._HF_MAP.items():
            hf[f"model.layers.{i}.{name}"] = w[f"l{i}.{short}"]
    hf = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in hf.items()}
    keys = sorted(hf)
    per = (len(keys) + shards - 1) // shards
    for s in range(shards):
        part = {k: hf[k] for k in keys[s * per:(s + 1) * per]}
        if fmt == "safetensors":
            from safetensors.torch import save_file
            save_file(part, os.path.join(tmp, f"model-{s + 1:05d}-of-{shards:05d}.safetensors"))
        else:
            torch.save(part, os.path.join(tmp, f"pytorch_model-{s + 1:05d}-of-{shards:05d}.bin"))


@pytest.mark.parametrize("fmt,shards", [("safetensors", 1), ("safetensors", 3), ("bin", 2)])
def test_llama_checkpoint_dir_round_trip(tmp_path, fmt, shards):
    shape = LlamaShape(vocab_size=320, hidden_size=64, intermediate_size=176, num_hidden_layers=3, num_attention_heads=4,
                       num_key_value_heads=2, rms_norm_eps=1e-6, rope_theta=1e6, max_position_embeddings=16384, name="x")
    w = W.make_weights_np(shape, seed=3)
    _llama_dir(str(tmp_path), shape, w, shards, fmt)
    got_shape = LlamaShape.from_hf_dir(str(tmp_path))
    for f in ("vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads",
              "num_key_value_heads", "rms_norm_eps", "rope_theta", "max_position_embeddings"):
        assert getattr(got_shape, f) == getattr(shape, f), f
    got = W.load_hf_safetensors(str(tmp_path), got_shape, "cpu", torch.float16)
    assert sorted(got) == sorted(w)
    for k, v in w.items():
        assert got[k].dtype == torch.float16 and np.array_equal(got[k].numpy(), v), k


def test_llama_tied_embeddings_and_missing_dir(tmp_path):
    shape = LlamaShape(vocab_size=64, hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2,
                       num_key_value_heads=2, name="t")
    w = W.make_weights_np(shape, seed=1)
    _llama_dir(str(tmp_path), shape, w)
    # drop lm_head from the shard: tied checkpoints omit it and the loader falls back to the embedding
    from safetensors.torch import load_file, save_file
    fn = [f for f in os.listdir(tmp_path) if f.endswith(".safetensors")][0]
    t = load_file(os.path.join(tmp_path, fn))
    del t["lm_head.weight"]
    save_file(t, os.path.join(tmp_path, fn))
    got = W.load_hf_safetensors(str(tmp_path), shape, "cpu", torch.float16)
    assert torch.equal(got["lm_head"], got["embed"])
    empty = tmp_path / "empty"
    empty.mkdir()
    with pytest.raises(FileNotFoundError):
        W.read_checkpoint_dir(str(empty))


def test_falcon_checkpoint_dir_round_trip(tmp_path):
    shape = FalconShape(vocab_size=128, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, name="f")
    w = W.make_falcon_weights_np(shape, 5, 1.0)
    cfg = dict(vocab_size=128, hidden_size=64, n_layer=2, n_head=4, multi_query=True, parallel_attn=True, bias=False,
               alibi=False, layer_norm_epsilon=shape.layer_norm_epsilon)
    with open(tmp_path / "config.json", "w") as f:
        json.dump(cfg, f)
    names = {"ln_w": "input_layernorm.weight", "ln_b": "input_layernorm.bias", "wqkv": "self_attention.query_key_value.weight",
             "wo": "self_attention.dense.weight", "w1": "mlp.dense_h_to_4h.weight", "w2": "mlp.dense_4h_to_h.weight"}
    hf = {"transformer.word_embeddings.weight": w["embed"], "transformer.ln_f.weight": w["lnf_w"],
          "transformer.ln_f.bias": w["lnf_b"], "lm_head.weight": w["lm_head"]}
    for i in range(2):
        for k, n in names.items():
            hf[f"transformer.h.{i}.{n}"] = w[f"l{i}.{k}"]
    from safetensors.torch import save_file
    save_file({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in hf.items()}, str(tmp_path / "model.safetensors"))
    got_shape = FalconShape.from_hf_dir(str(tmp_path))
    assert (got_shape.hidden_size, got_shape.num_hidden_layers, got_shape.num_attention_heads) == (64, 2, 4)
    got = W.load_falcon_safetensors(str(tmp_path), got_shape)
    for k, v in w.items():
        assert np.array_equal(got[k].numpy(), v), k
    bad = dict(cfg, alibi=True)
    with open(tmp_path / "config.json", "w") as f:
        json.dump(bad, f)
    with pytest.raises(ValueError):
        FalconShape.from_hf_dir(str(tmp_path))


def test_mpt_checkpoint_dir_round_trip(tmp_path):
    shape = MptShape(vocab_size=128, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, name="m")
    w = W.make_mpt_weights_np(shape, 6, 1.0)
    cfg = dict(vocab_size=128, d_model=64, n_layers=2, n_heads=4, no_bias=True, attn_config=dict(alibi=True, alibi_bias_max=8))
    with open(tmp_path / "config.json", "w") as f:
        json.dump(cfg, f)
    names = {"ln1": "norm_1.weight", "wqkv": "attn.Wqkv.weight", "wo": "attn.out_proj.weight", "ln2": "norm_2.weight",
             "w1": "ffn.up_proj.weight", "w2": "ffn.down_proj.weight"}
    hf = {"transformer.wte.weight": w["embed"], "transformer.norm_f.weight": w["lnf"], "lm_head.weight": w["lm_head"]}
    for i in range(2):
        for k, n in names.items():
            hf[f"transformer.blocks.{i}.{n}"] = w[f"l{i}.{k}"]
    torch.save({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in hf.items()}, str(tmp_path / "pytorch_model.bin"))
    got_shape = MptShape.from_hf_dir(str(tmp_path))
    assert (got_shape.hidden_size, got_shape.num_hidden_layers, got_shape.num_attention_heads) == (64, 2, 4)
    got = W.load_mpt_safetensors(str(tmp_path), got_shape)
    for k, v in w.items():
        assert np.array_equal(got[k].numpy(), v), k
