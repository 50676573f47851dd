"""Round 6 plane audit (Next 1): BASELINE config 4's cached step (13b shape, 40 layers, 259 new rows over 8 258 staged
keys) -- what does it cost at the logits to drop a split-precision plane?  Every variant against the numpy oracle on identical
weights and staged module KV (max |dlogit| over all 259 rows x the vocabulary), and the variant's TTFT on the same box.
  python tools/plane_audit.py mid       # the 65..512-row projections: lo plane of gate|up / down / o inputs dropped (PC_MID_LO_SKIP)
  python tools/plane_audit.py attn TAG  # the product .so as built (tools/plane_audit.sh rebuilds pc_attn_wide.hip with -DPC_WIDE_NOQLO / NOPLO)
The oracle's logits are computed once (first call) and kept in /tmp."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prompt-cache_amd")]
import bench
from promptcache_amd import CacheEngine, Prompt
from promptcache_amd.model import Llama2

mode = sys.argv[1] if len(sys.argv) > 1 else "mid"
tag = sys.argv[2] if len(sys.argv) > 2 else "product"
name, max_ctx, max_tokens, entries, label = bench.config_workload(4)
dev = "cuda:0"
lm = Llama2(name, device=dev, random_init=True, seed=0)
eng = CacheEngine(max_ctx, lm)
fmt = lm.get_formatter()
sp, pp = entries[0]
eng.add_schema(fmt(sp), max_tokens=max_tokens)
prompt = Prompt(pp, [fmt])
ORACLE = "/tmp/plane_audit_oracle.npy"


def run():
    eng.prompt_cache.reset()
    ids, pos, _, cache = eng.process(prompt)
    out = lm(input_ids=torch.tensor([ids], device=dev), position_ids=torch.tensor([pos], device=dev), past_key_values=cache, use_cache=True)
    torch.cuda.synchronize()
    return ids, pos, out.logits[0].float().cpu().numpy()


def ttft(reps=6):
    ts = []
    for _ in range(reps):
        eng.prompt_cache.reset()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ids, pos, _, cache = eng.process(prompt)
        lm(input_ids=torch.tensor([ids], device=dev), position_ids=torch.tensor([pos], device=dev), past_key_values=cache, use_cache=True)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts[1:])


ids, pos, base = run()
if not os.path.exists(ORACLE):
    t0 = time.perf_counter()
    from oracle.llama_oracle import LlamaOracle, OracleConfig, kv_gather
    m, c = lm.hf_model, lm.hf_model.config
    L, H, Hkv, D = c.num_hidden_layers, m.H, m.Hkv, m.D
    w = {"embed": m.embed.float().cpu().numpy(), "norm": m.norm.float().cpu().numpy(), "lm_head": m.lm_head.float().cpu().numpy()}
    for i in range(L):
        lw = m.layers[i]
        wqkv = lw["wqkv"].float().cpu().numpy()
        w[f"l{i}.wq"], w[f"l{i}.wk"], w[f"l{i}.wv"] = wqkv[:H * D], wqkv[H * D:(H + Hkv) * D], wqkv[(H + Hkv) * D:]
        wgu = lw["wgu"].float().cpu().numpy()
        w[f"l{i}.gate"], w[f"l{i}.up"] = wgu[:c.intermediate_size], wgu[c.intermediate_size:]
        for k in ("ln1", "ln2", "wo"):
            w[f"l{i}.{k}"] = lw[k].float().cpu().numpy()
        w[f"l{i}.down"] = lw["wdown"].float().cpu().numpy()
    cfg = OracleConfig(vocab_size=c.vocab_size, hidden_size=c.hidden_size, intermediate_size=c.intermediate_size, num_hidden_layers=L,
                       num_attention_heads=H, num_key_value_heads=Hkv, rms_norm_eps=c.rms_norm_eps, rope_theta=c.rope_theta,
                       inv_freq=m.inv_freq_cpu.numpy())
    from threadpoolctl import threadpool_limits
    segs = []
    for sc in eng.prompt_cache.staged:                 # (ONE host copy per module store: 6.5 GB for the 8 k-token context)
        st = sc.store.cpu().numpy()
        segs.append([(st[i, 0], st[i, 1]) for i in range(L)])
    with threadpool_limits(limits=16, user_api="blas"):
        staged, S = kv_gather(segs, eng.prompt_cache.max_ctx_length)
        ref, _ = LlamaOracle(cfg, w).forward(np.asarray([ids]), np.asarray([pos]), past=[(k[None], v[None]) for k, v in staged], n_layers=L)
    np.save(ORACLE, ref[0].astype(np.float32))
    print(f"[oracle] {time.perf_counter() - t0:.0f} s (host copy of the weights: {sum(v.nbytes for v in w.values()) / 2**30:.0f} GiB)", flush=True)
    del w, segs, staged
ref = np.load(ORACLE)
print(f"config 4: q={len(ids)} staged={len(eng.prompt_cache)} layers={lm.hf_model.config.num_hidden_layers} max|logit| {np.abs(ref).max():.2f}")


def report(label):
    _, _, lg = run()
    print(f"  {label:44s} max|dlogit| vs oracle {np.abs(lg - ref).max():.2e}   vs all planes {np.abs(lg - base).max():.2e}   TTFT {ttft():.2f} ms", flush=True)


if mode == "mid":
    for skip in ((), ("gu",), ("down",), ("o",), ("gu", "down"), ("gu", "down", "o")):
        lm.hf_model.mid_lo_skip = skip
        lm.hf_model._graphs.clear()
        report("projection inputs on one plane: " + (",".join(skip) or "none"))
else:
    report(f"attention over the staged keys [{tag}]")
