"""Debug helper: dense vs skinny LLM.int8 paths against the oracle on one forward (no engine)."""
import sys
import numpy as np
import torch
sys.path[:0] = [".", "prompt-cache_amd"]
from oracle.llama_oracle import OracleConfig
from oracle.llmint8_oracle import LlamaInt8Oracle
from promptcache_amd.model import Llama2
from promptcache_amd.model.config import SHAPES
from promptcache_amd.model.weights import make_weights_np

shape = SHAPES["mid64"]
w16 = make_weights_np(shape, 21, 2.0)
lm = Llama2(name="x", shape=shape, weights=w16, device="cuda:0", load_in_8bit=True)
cfg = OracleConfig(vocab_size=shape.vocab_size, hidden_size=shape.hidden_size, intermediate_size=shape.intermediate_size,
                   num_hidden_layers=shape.num_hidden_layers, num_attention_heads=shape.num_attention_heads,
                   num_key_value_heads=shape.num_key_value_heads, rms_norm_eps=shape.rms_norm_eps,
                   rope_theta=shape.rope_theta, inv_freq=lm.hf_model.inv_freq_cpu.numpy())
orc = LlamaInt8Oracle(cfg, {k: v.astype(np.float32) for k, v in w16.items()})
rng = np.random.default_rng(0)
for T in (40, 64, 65, 70):
    ids = rng.integers(3, shape.vocab_size, size=(1, T))
    pos = np.arange(T)[None]
    ref, present = orc.forward(ids, pos)
    for many in (True,):
        for nl in (1, None):
            out = lm.hf_model(input_ids=torch.tensor(ids, device="cuda"), position_ids=torch.tensor(pos, device="cuda"), use_cache=True,
                              many_rows=many, num_layers=nl)
            r = ref if nl is None else orc.forward(ids, pos, n_layers=nl)[0]
            k0 = out.past_key_values[0][0][0].float().cpu().numpy()
            print(f"T={T} many_rows={many} layers={nl}: max|dlogit| {np.abs(out.logits[0].cpu().numpy() - r[0]).max():.3e}  "
                  f"layer0 |dK| {np.abs(k0 - present[0][0][0]).max():.3e}")
