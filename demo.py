"""Counterpart of the reference's ``demo.py`` (:23-103) on the MI355X path: same call order --
model adapter -> ``CacheEngine(max_ctx, lm)`` -> ``add_schema(read_file(xml, [formatter]), max_tokens)`` -> ``Prompt`` ->
``process`` (with the adapter's ``use_full_position_ids``) -> ``GenerationEngine.generate`` -- run once with the prompt
cache and once with ``no_cache=True``, printing the two timed intervals of each (gather, first forward).

    python demo.py                                   # llama2-7b shape, random weights, synthetic game-like schema
    python demo.py --model falcon --schema my.xml --prompt-file my_prompt.xml
    python demo.py --model /path/to/hf/checkpoint    # real weights + tokenizer when a directory is available

There is no network on the build / bench machines, so without a checkpoint directory the adapters use seeded random
weights at the named shape and the deterministic stand-in tokenizer: the text is noise, the timings are real.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "prompt-cache_amd"))

from promptcache_amd import CacheEngine, GenerationEngine, GenerationParameters, Prompt, read_file, synth  # noqa: E402
from promptcache_amd.model import CodeLlama, Falcon, Llama2, Mpt  # noqa: E402

ADAPTERS = {"llama": (Llama2, "llama2-7b"), "codellama": (CodeLlama, "codellama-7b"), "falcon": (Falcon, "falcon-7b"),
            "mpt": (Mpt, "mpt-7b")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama", help="llama | codellama | falcon | mpt | <HF checkpoint directory>")
    ap.add_argument("--schema", help="PML schema file (default: a synthetic game-like schema, 7 modules)")
    ap.add_argument("--prompt-file", help="PML prompt file referencing the schema")
    ap.add_argument("--max-ctx", type=int, default=5000)
    ap.add_argument("--max-tokens", type=int, default=800)
    ap.add_argument("--max-new-tokens", type=int, default=32)
    a = ap.parse_args()

    if os.path.isdir(a.model):
        import json
        arch = json.load(open(os.path.join(a.model, "config.json"))).get("model_type", "llama")
        lm = {"falcon": Falcon, "mpt": Mpt}.get(arch, Llama2)(a.model)
    else:
        cls, shape = ADAPTERS[a.model]
        lm = cls(shape, random_init=True)
    preproc = [lm.get_formatter()]
    cache_engine = CacheEngine(a.max_ctx, lm)
    gen_engine = GenerationEngine(lm)
    if a.schema:
        cache_engine.add_schema(read_file(a.schema, preproc), max_tokens=a.max_tokens)
        prompt_text = open(a.prompt_file).read()
    else:
        schema_text, prompt_text = synth.flat_docs("code-generation-game", 30, (306, 76, 800, 800, 800, 800, 800), 12)
        cache_engine.add_schema(lm.get_formatter()(schema_text), max_tokens=a.max_tokens)
    params = GenerationParameters(temperature=1.0, repetition_penalty=1.0, top_p=0.95, top_k=-1,
                                  max_new_tokens=a.max_new_tokens, stop_token_ids=lm.stop_token_ids, stop_str=lm.stop_str)
    prompt = Prompt(prompt_text, preproc)
    for no_cache in (False, True, False):       # the first pass also warms the hipGraph / allocator
        token_ids, position_ids, cache_time, cache = cache_engine.process(
            prompt, no_cache=no_cache, return_full_position_ids=lm.use_full_position_ids)
        last = None
        for last in gen_engine.generate(token_ids, position_ids, params, cache, stream_interval=2,
                                        use_full_position_ids=lm.use_full_position_ids):
            pass
        staged = 0 if cache is None else cache[0][0].shape[1]
        print(f"{'no_cache' if no_cache else 'cached  '}: staged {staged:5d} + new {len(token_ids):5d} tokens | "
              f"gather {cache_time:7.3f} ms | first forward {last.elapsed_time:8.3f} ms | "
              f"TTFT {cache_time + last.elapsed_time:8.3f} ms | {a.max_new_tokens} tokens in {last.response_time:8.2f} ms")
    print("Assistant:", (last.new_text[:120] + " ...") if last else "")


if __name__ == "__main__":
    main()
