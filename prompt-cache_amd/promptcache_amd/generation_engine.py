"""Generation engine: one prefill over the staged KV (TTFT), then token-by-token decode.

Public surface of ``promptcache/generation_engine.py`` in the reference -- ``GenerationParameters`` (:21-42),
``is_partial_stop`` (:45-50), ``Output`` (:53-58), ``GenerationEngine.generate`` (:67-209) -- kept, including
its observable quirks:

* the first decoded token is placed at position ``max(position_ids) + 2``: the reference adds the loop index,
  which starts at 1 for the first decode step, to ``max(position_ids) + 1`` (:82, :132);
* ``Output`` is constructed positionally as ``Output(text, new_text, inference_time, response_time)`` (:201), so
  the field called ``response_time`` carries the running total and ``elapsed_time`` the prefill latency (TTFT).

Underneath, every forward is the HIP path: K/V of new tokens are appended in place to the arena behind ``cache``
(no per-layer ``torch.cat`` of the whole past, ``llama2.py:361-364``) and each decode step replays one captured
hipGraph (``model/llama_hip.py``).
"""
from __future__ import annotations

import gc
import os
from dataclasses import dataclass, field
from typing import Generator, List, Optional, Tuple

import torch

from .model import LanguageModel
from .model.kv_arena import StagedKV


@dataclass
class GenerationParameters:
    temperature: float = 1.0
    repetition_penalty: float = 1.0
    top_p: float = 1.0
    top_k: int = -1
    max_new_tokens: int = 256
    stop_token_ids: List[int] = field(default_factory=list)
    stop_str: List[str] = field(default_factory=list)
    echo: bool = True

    @property
    def greedy(self) -> bool:
        return self.temperature < 1e-5 or self.top_p < 1e-8

    def get_logits_processor(self):
        """Same processor chain, same order and thresholds as the reference (:32-42)."""
        from transformers.generation import logits_process as lp
        chain = lp.LogitsProcessorList()
        if self.temperature >= 1e-5 and self.temperature != 1.0:
            chain.append(lp.TemperatureLogitsWarper(self.temperature))
        if self.repetition_penalty > 1.0:
            chain.append(lp.RepetitionPenaltyLogitsProcessor(self.repetition_penalty))
        if 1e-8 <= self.top_p < 1.0:
            chain.append(lp.TopPLogitsWarper(self.top_p))
        if self.top_k > 0:
            chain.append(lp.TopKLogitsWarper(self.top_k))
        return chain


def is_partial_stop(output: str, stop_str: str) -> bool:
    """True while the tail of ``output`` may still grow into ``stop_str`` (reference :45-50; note that
    ``output[-0:]`` is the whole string, so the first probe compares against everything generated so far)."""
    return any(stop_str.startswith(output[-i:]) for i in range(0, min(len(output), len(stop_str))))


@dataclass
class Output:
    text: str
    new_text: str
    response_time: float = 0.0
    elapsed_time: float = 0.0


class _Timer:
    """HIP-event stopwatch around one ``lm()`` call (the reference's measuring recipe, :104-115)."""

    def __enter__(self):
        self.t0 = torch.cuda.Event(enable_timing=True)
        self.t1 = torch.cuda.Event(enable_timing=True)
        self.t0.record()
        return self

    def __exit__(self, *exc):
        self.t1.record()
        torch.cuda.synchronize()
        self.ms = self.t0.elapsed_time(self.t1)
        return False


class GenerationEngine:
    # decode greedy generations as a device-side loop when the model offers one (PC_DEVICE_GREEDY=0: step through lm())
    device_greedy_loop = os.environ.get("PC_DEVICE_GREEDY", "1") != "0"

    def __init__(self, lm: LanguageModel, verbose: bool = False):
        self.lm = lm
        self.verbose = verbose

    # -- pieces of the loop ---------------------------------------------------------------------
    def _forward(self, ids: List[int], positions: List[int], past) -> Tuple[torch.Tensor, object, float]:
        # Host tensors: the model's captured small-q forward takes token ids, positions, past length and a pending staging plan
        # in ONE pinned host-to-device copy (model/llama_hip.py _InputBlock); the reference uploads ids and positions as two
        # pageable copies in front of the timed call (:96-97).  Paths that want device tensors move them themselves.
        ids_t = torch.tensor([ids], dtype=torch.long)
        pos_t = torch.tensor([positions], dtype=torch.long)
        with _Timer() as t:
            out = self.lm(input_ids=ids_t, position_ids=pos_t, past_key_values=past, use_cache=True)
        return out.logits, out.past_key_values, t.ms

    @staticmethod
    def _pick(last_logits: torch.Tensor, greedy: bool) -> int:
        if greedy:
            return int(torch.argmax(last_logits))
        return int(torch.multinomial(torch.softmax(last_logits, dim=-1), num_samples=1))

    def _render(self, output_ids: List[int], new_ids: List[int], stop_strs: List[str]) -> Tuple[str, str, bool, bool]:
        """-> (text, new_text, hit_stop_string, partially_matched)."""
        text, new_text = self.lm.decode(output_ids), self.lm.decode(new_ids)
        for stop in stop_strs:
            cut = new_text.rfind(stop, 0)
            if cut != -1:
                return text, new_text[:cut], True, False
            if is_partial_stop(text, stop):
                return text, new_text, False, True
        return text, new_text, False, False

    # -- the generator ----------------------------------------------------------------------------
    @torch.inference_mode()
    def generate(self, token_ids: List[int], position_ids: List[int], params: GenerationParameters,
                 cache=None, stream_interval: int = 2, use_full_position_ids: bool = False
                 ) -> Generator[Output, None, None]:
        processors = params.get_logits_processor()
        prompt_positions = list(position_ids)
        first_free = max(prompt_positions) + 1
        output_ids, new_ids = list(token_ids), []
        total_ms = ttft_ms = 0.0
        past = None

        # Greedy decoding whose processed argmax is the raw argmax (no repetition penalty: temperature, top-p and top-k
        # all keep the largest logit largest) runs as a device-side loop when the model offers one: every step is one
        # hipGraph replay that also picks the token and feeds it to the next replay (model/llama_hip.py GreedyLoop).
        # The host keeps ONE replay in flight ahead of the token it is looking at, so stop conditions are evaluated
        # exactly as below while the GPU never waits for the host; a step enqueued past a stop is simply discarded.
        loop = None
        want_loop = params.greedy and params.repetition_penalty <= 1.0 and not use_full_position_ids and \
            hasattr(getattr(self.lm, "hf_model", None), "greedy_loop") and self.device_greedy_loop

        try:
            for step in range(params.max_new_tokens):
                if loop is not None:
                    if loop.n <= step and loop.n < params.max_new_tokens - 1:
                        loop.enqueue()                             # the replay AFTER the one whose token is read below
                    token = loop.token(step - 1)
                    total_ms += loop.elapsed_ms(step - 1)
                    output_ids.append(token)
                    new_ids.append(token)
                    done = token in params.stop_token_ids
                    if step % stream_interval == 0 or step == params.max_new_tokens - 1 or done:
                        text, new_text, hit, partial = self._render(output_ids, new_ids, params.stop_str)
                        done = done or hit
                        if not partial:
                            yield Output(text, new_text, total_ms, ttft_ms)
                    if done:
                        break
                    continue
                if step == 0:
                    if cache is not None and not isinstance(cache, StagedKV):
                        # a plain list of [Hkv, S, D] views: add the batch dim like the reference does (:101-102)
                        cache = [(k.unsqueeze(0), v.unsqueeze(0)) if k.dim() == 3 else (k, v) for k, v in cache]
                    logits, past, ms = self._forward(list(token_ids), prompt_positions, cache)
                    ttft_ms = ms
                    if self.verbose:
                        print(f"Prefill latency: {ms:.2f} ms")
                else:
                    positions = (prompt_positions + list(range(first_free, first_free + step))) if use_full_position_ids \
                        else [first_free + step]
                    logits, past, ms = self._forward([new_ids[-1]], positions, past)
                total_ms += ms

                history = torch.as_tensor([output_ids], device=self.lm.device) if params.repetition_penalty > 1.0 else None
                token = self._pick(processors(history, logits[:, -1, :])[0], params.greedy)
                output_ids.append(token)
                new_ids.append(token)

                done = token in params.stop_token_ids
                if step % stream_interval == 0 or step == params.max_new_tokens - 1 or done:
                    text, new_text, hit, partial = self._render(output_ids, new_ids, params.stop_str)
                    done = done or hit
                    if not partial:
                        yield Output(text, new_text, total_ms, ttft_ms)
                if done:
                    break
                if step == 0 and want_loop and params.max_new_tokens > 1:
                    # the first decoded token sits at position first_free + 1 (the reference's loop index starts at 1, :132)
                    loop = self.lm.hf_model.greedy_loop(past, token, first_free + 1, params.max_new_tokens)
                    if loop is not None:
                        loop.enqueue()
        finally:
            # also when the consumer abandons the generator at a yield (GeneratorExit): a look-ahead replay enqueued past the last
            # token the caller saw wrote one arena row too many, and the model's loop state must be released
            if loop is not None:
                loop.close(len(new_ids) - 1)                 # steps of the loop that produced a token (the first came from the prefill)
            del past, loop
            gc.collect()
