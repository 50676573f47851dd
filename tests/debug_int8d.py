import sys
import numpy as np
import torch
sys.path[:0] = [".", "prompt-cache_amd"]
from oracle.llama_oracle import OracleConfig
from oracle import llmint8_oracle as lo
from promptcache_amd.model import Llama2
from promptcache_amd.model.config import SHAPES
from promptcache_amd.model.weights import make_weights_np

shape = SHAPES["mid64"]
import os
w16 = make_weights_np(shape, 21, float(os.environ.get("WSCALE", "2.0")))
lm = Llama2(name="x", shape=shape, weights=w16, device="cuda:0", load_in_8bit=True)
cfg = OracleConfig(vocab_size=shape.vocab_size, hidden_size=shape.hidden_size, intermediate_size=shape.intermediate_size,
                   num_hidden_layers=shape.num_hidden_layers, num_attention_heads=shape.num_attention_heads,
                   num_key_value_heads=shape.num_key_value_heads, rms_norm_eps=shape.rms_norm_eps,
                   rope_theta=shape.rope_theta, inv_freq=lm.hf_model.inv_freq_cpu.numpy())


class Spy(lo.LlamaInt8Oracle):
    def _lin(self, x, key):
        if key in self.q:
            _, sca, cols, _ = lo.quantize_activations(x.reshape(-1, x.shape[-1]))
            print(f"   oracle {key}: K={x.shape[-1]} outlier cols {cols.size} max|x| {np.abs(x).max():.2f}")
        return super()._lin(x, key)


orc = Spy(cfg, {k: v.astype(np.float32) for k, v in w16.items()})
import io, contextlib
for seed in range(8):
    rng = np.random.default_rng(seed)
    T = 40
    ids = rng.integers(3, shape.vocab_size, size=(1, T))
    pos = np.arange(T)[None]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ref, present = orc.forward(ids, pos, n_layers=1)
    out = lm.hf_model(input_ids=torch.tensor(ids, device="cuda"), position_ids=torch.tensor(pos, device="cuda"), use_cache=True,
                      many_rows=True, num_layers=1)
    d = np.abs(out.logits[0].cpu().numpy() - ref[0]).max()
    out_s = lm.hf_model(input_ids=torch.tensor(ids, device="cuda"), position_ids=torch.tensor(pos, device="cuda"), use_cache=True,
                        many_rows=False, num_layers=1)
    ds = np.abs(out_s.logits[0].cpu().numpy() - ref[0]).max()
    kd = np.abs(out.past_key_values[0][1][0].float().cpu().numpy() - present[0][1][0]).max()
    print("seed", seed, "dense dlogit", d, "skinny dlogit", ds, "dense |dV|", kd)
