"""Layer-by-layer parity probe at the 7b layer shape: GPU k-layer stack vs numpy oracle, k = 0..K."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "prompt-cache_amd"))
from promptcache_amd.model.config import LlamaShape
from promptcache_amd.model.llama_hip import LlamaHIP
from promptcache_amd.model.weights import make_weights_np
from oracle.llama_oracle import LlamaOracle, OracleConfig

K = int(os.environ.get("K", "3"))
shape = LlamaShape(vocab_size=4096, num_hidden_layers=K, name="dbg")
w16 = make_weights_np(shape, 0, 1.0)
m = LlamaHIP(shape, w16, device="cuda:0")
m.use_graphs = False
cfg = OracleConfig(vocab_size=shape.vocab_size, hidden_size=shape.hidden_size, intermediate_size=shape.intermediate_size,
                   num_hidden_layers=K, num_attention_heads=32, num_key_value_heads=32, rms_norm_eps=shape.rms_norm_eps,
                   rope_theta=shape.rope_theta, inv_freq=m.inv_freq_cpu.numpy())
orc = LlamaOracle(cfg, {k: v.astype(np.float32) for k, v in w16.items()})
rng = np.random.default_rng(0)
S, q = 300, 12
past = rng.standard_normal((K, 2, 32, S, 128), dtype=np.float32).astype(np.float16)
ids = rng.integers(3, 4096, size=(1, q)); pos = np.arange(1000, 1000 + q)[None]
for skinny in (True, False):
    m.skinny = skinny
    for k in range(0, K + 1):
        arena = m.new_arena(1, S + q + 8)
        arena.buf[0, :K, :, :, :S] = torch.from_numpy(past).cuda()
        arena.length = S
        out = m(input_ids=torch.from_numpy(ids).cuda(), position_ids=torch.from_numpy(pos).cuda(),
                past_key_values=arena.views(S), num_layers=k)
        pk = [(past[i, 0][None], past[i, 1][None]) for i in range(k)]
        lg, _ = orc.forward(ids, pos, past=pk if k else None, n_layers=k) if k else (None, None)
        if k == 0:
            from oracle.llama_oracle import rmsnorm
            x = orc.w["embed"][ids]
            lg = rmsnorm(x, orc.w["norm"], cfg.rms_norm_eps) @ orc.w["lm_head"].T
        d = np.abs(out.logits[0].cpu().numpy() - lg[0])
        print(f"skinny={skinny} layers={k}: max|dlogit|={d.max():.2e} mean={d.mean():.2e}  max|logit|={np.abs(lg).max():.2f}")
