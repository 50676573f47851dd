"""Where does the full-depth (32-layer, 7b-shape) logit error come from?  Diagnostic, run by hand on the GPU box:
    python -m tests.probes.full_depth [layers]
Compares, against the numpy oracle: (a) the module KV the GPU encode stored, per layer; (b) the cached prefill run on
the ORACLE's staged KV (isolates the prefill); (c) the end-to-end logits."""
import os
import sys
import time

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "prompt-cache_amd"))

import numpy as np
import torch

from oracle import engine_oracle as eo
from oracle.llama_oracle import LlamaOracle, OracleConfig, kv_gather


def main(layers=32, qlen=8):
    import dataclasses
    from promptcache_amd import CacheEngine, Prompt, synth
    from promptcache_amd.model import Llama2
    from promptcache_amd.model.config import SHAPES
    from promptcache_amd.model.kv_arena import KVArena
    from promptcache_amd.model.weights import random_weights_device
    shape = dataclasses.replace(SHAPES[os.environ.get("PC_DBG_SHAPE", "llama2-7b")], num_hidden_layers=layers)
    seed = int(os.environ.get("PC_DBG_SEED", "5"))
    w = random_weights_device(shape, "cuda:0", torch.float16, seed=seed)
    int8 = os.environ.get("PC_DBG_INT8") == "1"          # load_in_8bit: the oracle then runs on the dequantised weights
    lm = Llama2(name="llama2-7b", shape=shape, weights=w, device="cuda:0", load_in_8bit=int8)
    if os.environ.get("PC_DBG_BENCH_WORKLOAD") == "1":     # the bench.py workload itself: 29 passes, S = 1725, q = 12
        sp, pp = synth.persona_like("p7")
    else:
        sp, pp = synth.persona_like("p7", system_len=120, intro_len=30,
                                    traits=(("age", (40, 35, 44)), ("home", (60, 52, 57)), ("job", (45, 50, 41))), question_len=qlen, seed=seed + 4)
    fmt = lm.get_formatter()
    eng = CacheEngine(4096 if os.environ.get('PC_DBG_BENCH_WORKLOAD') == '1' else 2048, lm)
    eng.add_schema(fmt(sp))
    prompt = Prompt(pp, [fmt])
    ids, pos, _, cache = eng.process(prompt)
    out = lm(input_ids=torch.tensor([ids], device="cuda"), position_ids=torch.tensor([pos], device="cuda"),
             past_key_values=cache, use_cache=True)
    got = out.logits[0].cpu().numpy()
    t0 = time.perf_counter()
    cfg = OracleConfig(vocab_size=shape.vocab_size, hidden_size=shape.hidden_size, intermediate_size=shape.intermediate_size,
                       num_hidden_layers=shape.num_hidden_layers, num_attention_heads=shape.num_attention_heads,
                       num_key_value_heads=shape.num_key_value_heads, rms_norm_eps=shape.rms_norm_eps,
                       rope_theta=shape.rope_theta, inv_freq=lm.hf_model.inv_freq_cpu.numpy())
    wnp = {k: v.float().cpu().numpy() for k, v in w.items()}
    if int8:
        from oracle import int8_oracle as io
        wnp = io.dequantized_llama_weights(wnp)
    model = LlamaOracle(cfg, wnp)
    sc = eng.get_schema("p7")
    jobs = []
    for p in sc.encode_paths():
        sf = sc.get_scaffold(p)
        jobs.append(dict(token_ids=sf.token_ids(), position_ids=sf.position_ids(), targets=sf.select(p).all_token_sequences()))
    lib = eo.encode_schema(model, jobs)
    used = [m.token_sequence for m in eng.prompt_cache.staged]
    staged, S, (logits, _) = eo.cached_prefill(model, lib, used, ids, pos, eng.prompt_cache.max_ctx_length)
    print(f"layers={layers} S={S} q={len(ids)} oracle {time.perf_counter()-t0:.0f}s  max|logit|={np.abs(logits).max():.2f}")
    print(f"(c) end to end      max|dlogit| = {np.abs(got - logits[0]).max():.3e}")
    # (a) stored module KV per layer
    kerr = np.zeros(layers); verr = np.zeros(layers); kmag = np.zeros(layers)
    for m in eng.prompt_cache.staged:
        o = lib[id(m.token_sequence)]
        st = m.store.float().cpu().numpy()              # [L,2,H,len,D]
        for l in range(layers):
            kerr[l] = max(kerr[l], np.abs(st[l, 0] - o[l][0]).max()); verr[l] = max(verr[l], np.abs(st[l, 1] - o[l][1]).max())
            kmag[l] = max(kmag[l], np.abs(o[l][0]).max())
    for l in sorted(set([0, 1, 2, 3, layers // 4, layers // 2, layers - 1])):
        print(f"(a) layer {l:2d}: max|dK| {kerr[l]:.2e}  max|dV| {verr[l]:.2e}  (max|K| {kmag[l]:.1f})")
    # (b) prefill on the oracle's staged KV
    arena = KVArena(1, layers, shape.num_key_value_heads, eng.prompt_cache.max_ctx_length, shape.head_dim, "cuda:0")
    for l, (k, v) in enumerate(staged):
        arena.buf[0, l, 0, :, :S] = torch.from_numpy(k).cuda()
        arena.buf[0, l, 1, :, :S] = torch.from_numpy(v).cuda()
    arena.length = S
    out2 = lm(input_ids=torch.tensor([ids], device="cuda"), position_ids=torch.tensor([pos], device="cuda"),
              past_key_values=arena.views(), use_cache=True)
    print(f"(b) prefill on oracle-staged KV  max|dlogit| = {np.abs(out2.logits[0].cpu().numpy() - logits[0]).max():.3e}")
    # (e) decode steps after (b), teacher-forced with the oracle's greedy tokens (generation_engine.py:123-147: the i-th
    #     decoded token sits at position max(position_ids) + 1 + i)
    steps = int(os.environ.get("PC_DBG_DECODE", "6"))
    if steps:
        _, present = model.forward(np.asarray([ids]), np.asarray([pos]), past=[(k[None], v[None]) for k, v in staged])
        olog, gpast, worst = logits, out2.past_key_values, 0.0
        for i in range(steps):
            tok = int(np.argmax(olog[0, -1]))
            p1 = max(pos) + 1 + i
            olog, present = model.forward(np.array([[tok]]), np.array([[p1]]), past=present)
            go = lm(input_ids=torch.tensor([[tok]], device="cuda"), position_ids=torch.tensor([[p1]], device="cuda"),
                    past_key_values=gpast, use_cache=True)
            gpast = go.past_key_values
            d = np.abs(go.logits[0, -1].cpu().numpy() - olog[0, -1]).max()
            worst = max(worst, d)
            print(f"(e) decode step {i}: max|dlogit| = {d:.3e}  argmax equal: {int(np.argmax(olog[0, -1])) == int(go.logits[0, -1].argmax())}")
        arena.length = S
    # (b') the same through the stacked-GEMM many-row path (no weight-streaming kernels)
    lm.hf_model.skinny = False
    arena.length = S
    out2b = lm(input_ids=torch.tensor([ids], device="cuda"), position_ids=torch.tensor([pos], device="cuda"),
               past_key_values=arena.views(S), use_cache=True)
    lm.hf_model.skinny = True
    print(f"(b') same, many-row path          max|dlogit| = {np.abs(out2b.logits[0].cpu().numpy() - logits[0]).max():.3e}")


    # (d) the no-cache path: every token re-encoded, positions range(N) (cache_engine.py:476-493)
    nids, npos, _, _ = eng.process(prompt, no_cache=True)
    out3 = lm(input_ids=torch.tensor([list(nids)], device="cuda"), position_ids=torch.tensor([npos], device="cuda"), use_cache=True)
    lg3, _ = model.forward(np.asarray([list(nids)]), np.asarray([npos]))
    print(f"(d) no-cache, {len(nids)} tokens      max|dlogit| = {np.abs(out3.logits[0].cpu().numpy() - lg3[0]).max():.3e}")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 32, int(sys.argv[2]) if len(sys.argv) > 2 else 8)
