"""LLM.int8 cached prefill + decode of the bench prompt, for rocprofv3 --kernel-trace --stats."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "prompt-cache_amd")]
from promptcache_amd import CacheEngine, Prompt, synth  # noqa: E402
from promptcache_amd.model import Llama2  # noqa: E402

lm = Llama2("llama2-7b", device="cuda:0", random_init=True, seed=0)
eng = CacheEngine(4096, lm)
fmt = lm.get_formatter()
sp, pp = synth.persona_like()
eng.add_schema(fmt(sp))
lm8 = Llama2("llama2-7b", device="cuda:0", random_init=True, seed=0, load_in_8bit=True)
prompt = Prompt(pp, [fmt])
for i in range(8):
    eng.prompt_cache.reset()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ids, pos, _, cache = eng.process(prompt)
    o = lm8(input_ids=torch.tensor([ids], device="cuda"), position_ids=torch.tensor([pos], device="cuda"), past_key_values=cache, use_cache=True)
    torch.cuda.synchronize(); print(f"ttft {i}: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
if os.environ.get("PC_I8_DECODE", "1") == "1":
    past, tok = o.past_key_values, int(torch.argmax(o.logits[0, -1]))
    for phase in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(32):
            o = lm8(input_ids=torch.tensor([[tok]], device="cuda"), position_ids=torch.tensor([[max(pos) + 2 + phase * 32 + i]], device="cuda"),
                    past_key_values=past, use_cache=True)
            past, tok = o.past_key_values, int(torch.argmax(o.logits[0, -1]))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"decode phase {phase}: {32 / dt:.1f} tok/s ({dt / 32 * 1e3:.3f} ms/token)", flush=True)
    loop = lm8.hf_model.greedy_loop(past, tok, max(pos) + 2 + 64, 128)
    if loop is not None and os.environ.get("PC_I8_LOOP", "1") == "1":
        for _ in range(32):
            loop.enqueue()
        loop.token(31)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(64):
            loop.enqueue()
        loop.token(95)
        dt = time.perf_counter() - t0
        print(f"decode device loop: {64 / dt:.1f} tok/s ({dt / 64 * 1e3:.3f} ms/token)", flush=True)
