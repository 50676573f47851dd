"""Generation engine: one prefill over the staged KV (TTFT), then token-by-token decode.

Mirrors ``promptcache/generation_engine.py`` of the reference -- ``GenerationParameters`` (:21-42),
``is_partial_stop`` (:45-50), ``Output`` (:53-58), ``GenerationEngine.generate`` (:67-209) -- including
its observable quirks: the first decoded token is placed at position ``max(position_ids) + 2``
(``position_offset + i`` with ``i`` starting at 1, :82/:132), and ``Output`` is built as
``Output(text, new_text, inference_time, response_time)`` (:201), so the field named
``elapsed_time`` carries the prefill latency (TTFT) and ``response_time`` the running total.

What changes underneath: the model appends K/V in place to the arena behind ``cache`` (no ``torch.cat``
of the whole past per layer per step, ``llama2.py:361-364``) and only the last row goes through
``lm_head`` during decode.
"""
from __future__ import annotations

import gc
from dataclasses import dataclass, field
from typing import Generator, List, Optional

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

    def get_logits_processor(self):
        from transformers.generation.logits_process import (
            LogitsProcessorList, RepetitionPenaltyLogitsProcessor, TemperatureLogitsWarper, TopKLogitsWarper,
            TopPLogitsWarper)
        p = LogitsProcessorList()
        if self.temperature >= 1e-5 and self.temperature != 1.0:
            p.append(TemperatureLogitsWarper(self.temperature))
        if self.repetition_penalty > 1.0:
            p.append(RepetitionPenaltyLogitsProcessor(self.repetition_penalty))
        if 1e-8 <= self.top_p < 1.0:
            p.append(TopPLogitsWarper(self.top_p))
        if self.top_k > 0:
            p.append(TopKLogitsWarper(self.top_k))
        return p


def is_partial_stop(output: str, stop_str: str) -> bool:
    """Whether the tail of ``output`` could still grow into ``stop_str`` (reference :45-50)."""
    for i in range(0, min(len(output), len(stop_str))):
        if stop_str.startswith(output[-i:]):
            return True
    return False


@dataclass
class Output:
    text: str
    new_text: str
    response_time: float = 0.0
    elapsed_time: float = 0.0


class GenerationEngine:
    def __init__(self, lm: LanguageModel, verbose: bool = False):
        self.lm = lm
        self.verbose = verbose

    @torch.inference_mode()
    def generate(self, token_ids: List[int], position_ids: List[int], params: GenerationParameters,
                 cache=None, stream_interval: int = 2, use_full_position_ids: bool = False
                 ) -> Generator[Output, None, None]:
        lm = self.lm
        device = lm.device
        processors = params.get_logits_processor()
        greedy = params.temperature < 1e-5 or params.top_p < 1e-8
        output_ids = list(token_ids)
        new_output_ids: List[int] = []
        position_offset = max(position_ids) + 1
        prompt_positions = list(position_ids)
        past = None
        inference_time = 0.0
        response_time = 0.0
        new_token_id = 0

        for i in range(params.max_new_tokens):
            start = torch.cuda.Event(enable_timing=True)
            end = torch.cuda.Event(enable_timing=True)
            if past is None:
                ids_t = torch.tensor([list(token_ids)], device=device, dtype=torch.long)
                pos_t = torch.tensor([prompt_positions], device=device, dtype=torch.long)
                if cache is not None and not isinstance(cache, StagedKV):
                    # plain list of [Hkv,S,D] views: add the batch dim like the reference (:101-102)
                    cache = [(k.unsqueeze(0), v.unsqueeze(0)) if k.dim() == 3 else (k, v) for k, v in cache]
                start.record()
                out = lm(input_ids=ids_t, position_ids=pos_t, past_key_values=cache, use_cache=True)
                end.record()
                torch.cuda.synchronize()
                inference_time += start.elapsed_time(end)
                response_time = inference_time            # TTFT as the reference reports it (:114-118)
                if self.verbose:
                    print(f"Prefill latency: {inference_time:.2f} ms")
            else:
                ids_t = torch.tensor([[new_token_id]], device=device, dtype=torch.long)
                if use_full_position_ids:
                    pos_t = torch.tensor([prompt_positions + list(range(position_offset, position_offset + i))],
                                         device=device, dtype=torch.long)
                else:
                    pos_t = torch.tensor([[position_offset + i]], device=device, dtype=torch.long)
                start.record()
                out = lm(input_ids=ids_t, position_ids=pos_t, past_key_values=past, use_cache=True)
                end.record()
                torch.cuda.synchronize()
                inference_time += start.elapsed_time(end)
            logits = out.logits
            past = out.past_key_values

            history = torch.as_tensor([output_ids], device=device) if params.repetition_penalty > 1.0 else None
            last = processors(history, logits[:, -1, :])[0]
            if greedy:
                new_token_id = int(torch.argmax(last))
            else:
                new_token_id = int(torch.multinomial(torch.softmax(last, dim=-1), num_samples=1))
            output_ids.append(new_token_id)
            new_output_ids.append(new_token_id)

            stopped = new_token_id in params.stop_token_ids
            if i % stream_interval == 0 or i == params.max_new_tokens - 1 or stopped:
                text = lm.decode(output_ids)
                new_text = lm.decode(new_output_ids)
                partial = False
                for stop in params.stop_str:
                    pos = new_text.rfind(stop, 0)
                    if pos != -1:
                        new_text = new_text[:pos]
                        stopped = True
                        break
                    partial = is_partial_stop(text, stop)
                    if partial:
                        break
                if not partial:
                    yield Output(text, new_text, inference_time, response_time)
            if stopped:
                break

        del past
        gc.collect()
