"""Model adapters: the drop-in boundary of the prompt-cache hot path.

Mirrors ``promptcache/model/__init__.py`` of the reference: ``FormatConversation`` (:48-77),
``LanguageModel`` (:90-161), ``CodeLlama`` (:164-182), ``Llama2`` (:185-203).  ``CacheEngine`` and
``GenerationEngine`` only ever talk to this surface, so an adapter whose ``hf_model`` is the native
``LlamaHIP`` forward makes the whole path run on the HIP kernels.

``Falcon`` (reference :206-258) runs the same kernels through ``falcon_hip.FalconHIP`` (multi-query cache shape
``(L, 1, D)``) and ``Mpt`` (reference :261-288) through ``mpt_hip.MptHIP`` (ALiBi at the keys' position ids,
``use_full_position_ids``).
"""
from __future__ import annotations

import abc
import os
import re
from typing import Callable, List, Optional, Tuple

import torch

from ..pml import Preprocessor, PreprocessorList, escape_xml  # noqa: F401  (re-exported like the reference)
from .config import FALCON_SHAPES, MPT_SHAPES, SHAPES, FalconShape, LlamaShape, MptShape
from .tokenizer import StandInTokenizer

# HF hub ids the reference's drivers use (demo.py:27, eval.py:36, config/*.json) -> shape presets
_HUB_TO_SHAPE = {
    "meta-llama/Llama-2-7b-chat-hf": "llama2-7b", "meta-llama/Llama-2-7b-hf": "llama2-7b",
    "meta-llama/Llama-2-13b-chat-hf": "llama2-13b", "meta-llama/Llama-2-13b-hf": "llama2-13b",
    "codellama/CodeLlama-7b-Instruct-hf": "codellama-7b", "codellama/CodeLlama-7b-hf": "codellama-7b",
}


class FormatConversation(Preprocessor):
    """Literal replacement of the chat tags by the model's chat strings, XML-escaped, applied to the
    raw text BEFORE XML parsing (reference :48-77)."""

    def __init__(self, system: Tuple[str, str, str], user: Tuple[str, str], assistant: Tuple[str, str]):
        self.system = tuple(escape_xml(s) for s in system)
        self.user = tuple(escape_xml(s) for s in user)
        self.assistant = tuple(escape_xml(s) for s in assistant)

    def __call__(self, prompt: str) -> str:
        prompt = re.sub(r" +<system>", "<system>", prompt)
        for old, new in (("<system>", self.system[0]), ("</system>", self.system[1]), ("<system/>", self.system[2]),
                         ("<user>", self.user[0]), ("</user>", self.user[1]),
                         ("<assistant>", self.assistant[0]), ("</assistant>", self.assistant[1])):
            prompt = prompt.replace(old, new)
        return prompt


class LanguageModel(abc.ABC):
    """What the engines need from a model (reference :90-161)."""
    use_full_position_ids: bool = False

    def __init__(self, name: str, model, tokenizer, stop_token_ids: Optional[List[int]] = None,
                 stop_str: Optional[List[str]] = None):
        self.name = name
        self.hf_tokenizer = tokenizer
        self.hf_model = model
        self.stop_token_ids = stop_token_ids if stop_token_ids is not None else [self.eos_token_id]
        self.stop_str = stop_str if stop_str is not None else []

    @abc.abstractmethod
    def get_formatter(self) -> Callable[[str], str]:
        ...

    def get_cache_shape(self) -> Tuple[int, int, int]:
        """(n_layers, n_kv_heads, head_dim).  The reference returns ``num_attention_heads`` (:110-114),
        equal to the KV head count for every Llama-2 7b/13b-class model it supports."""
        c = self.config
        return c.num_hidden_layers, c.num_key_value_heads, c.hidden_size // c.num_attention_heads

    # identity hooks for Llama (reference :116-126)
    def store_k_hook(self, k_cache):
        return k_cache

    def store_v_hook(self, v_cache):
        return v_cache

    def read_k_hook(self, k_cache):
        return k_cache

    def read_v_hook(self, v_cache):
        return v_cache

    def __call__(self, **kwargs):
        return self.hf_model(**kwargs)

    def encode(self, text: str) -> List[int]:
        return self.hf_tokenizer.encode(text, add_special_tokens=False)   # no BOS (reference :131-134)

    def decode(self, token_ids: List[int]) -> str:
        return self.hf_tokenizer.decode(token_ids, skip_special_tokens=False, spaces_between_special_tokens=False)

    @property
    def unk_token(self):
        return self.hf_tokenizer.unk_token

    @property
    def unk_token_id(self) -> int:
        return self.hf_tokenizer.unk_token_id

    @property
    def eos_token(self):
        return self.hf_tokenizer.eos_token

    @property
    def eos_token_id(self) -> int:
        return self.hf_tokenizer.eos_token_id

    @property
    def device(self) -> torch.device:
        return self.hf_model.device

    @property
    def config(self):
        return self.hf_model.config


def _llama_formatter() -> FormatConversation:
    # reference :170-173 and :191-194 (identical strings for CodeLlama and Llama2)
    return FormatConversation(system=("<s> [INST] <<SYS>>\n", "<</SYS>>\n\n", "<s> [INST] "),
                              user=("", "[/INST]"), assistant=("", "</s><s> [INST] "))


def _falcon_formatter() -> PreprocessorList:
    """Reference :212-235: newline normalisation, then the chat strings."""
    def rep(prompt: str) -> str:
        return prompt.replace("\r\n", "\n").replace("\n\n", "\n")
    conv = FormatConversation(system=("", "\n\n", ""), user=("User: ", "\n\nAssistant:"), assistant=(" ", "\n\n"))
    return PreprocessorList([rep, conv])


def _mpt_formatter() -> FormatConversation:
    """Reference :269-272."""
    return FormatConversation(system=("<|im_start|>system\n", "<|im_end|>\n", ""),
                              user=("<|im_start|>user\n", "<|im_end|>\n<|im_start|>assistant\n"),
                              assistant=("", "<|im_end|>\n"))


class _LlamaFamily(LanguageModel):
    """Shared constructor: a local HF checkpoint directory when one exists, else (explicitly requested)
    seeded random weights at the named shape with the deterministic stand-in tokenizer -- the build and
    benchmark machines have neither checkpoints nor network."""

    _default_name = "meta-llama/Llama-2-7b-chat-hf"

    def __init__(self, name: Optional[str] = None, device: str = "cuda:0", shape: Optional[LlamaShape] = None,
                 weights=None, tokenizer=None, random_init: bool = False, seed: int = 0, **_hf_kwargs):
        from .llama_hip import LlamaHIP
        from . import weights as W

        name = name or self._default_name
        if weights is not None:                                       # caller-supplied tensors (tests)
            assert shape is not None, "pass shape= together with weights="
        elif os.path.isdir(name):                                     # real checkpoint directory
            shape = shape or LlamaShape.from_hf_dir(name)
            weights = W.load_hf_safetensors(name, shape, device, torch.float16)
            if tokenizer is None:
                from transformers import AutoTokenizer
                tokenizer = AutoTokenizer.from_pretrained(name)
        else:
            key = _HUB_TO_SHAPE.get(name, name)
            if shape is None:
                if key not in SHAPES:
                    raise ValueError(f"unknown model {name!r}: pass a checkpoint directory or one of {sorted(SHAPES)}")
                shape = SHAPES[key]
            if not random_init:
                raise FileNotFoundError(
                    f"no checkpoint directory {name!r} (there is no network here); pass random_init=True to use "
                    f"seeded N(0, {shape.initializer_range}) weights at the {shape.name} shape")
            weights = W.random_weights_device(shape, device, torch.float16, seed)
        if tokenizer is None:
            tokenizer = StandInTokenizer(shape.vocab_size)
        # load_in_8bit (the reference's GPU configs, config/llm_config_*.json:5): weight-only int8 for the decoder linears
        model = LlamaHIP(shape, weights, device=device, int8_weights=bool(_hf_kwargs.get("load_in_8bit", False)))
        self.formatter = _llama_formatter()
        super().__init__(name, model, tokenizer, [tokenizer.eos_token_id], ["</s>"])

    def get_formatter(self) -> Callable[[str], str]:
        return self.formatter


class Llama2(_LlamaFamily):
    _default_name = "meta-llama/Llama-2-7b-chat-hf"


class CodeLlama(_LlamaFamily):
    _default_name = "codellama/CodeLlama-13b-Instruct-hf"


class Falcon(LanguageModel):
    """Reference ``Falcon`` adapter (:206-258): newline-normalising formatter + chat strings, stop tokens 0..11,
    stop strings ``<|endoftext|>`` / ``\\nUser``, and the multi-query cache shape ``(L, 1, head_dim)``."""

    _default_name = "tiiuae/falcon-7b-instruct"

    def __init__(self, name: Optional[str] = None, device: str = "cuda:0", shape: Optional[FalconShape] = None,
                 weights=None, tokenizer=None, random_init: bool = False, seed: int = 0, **_hf_kwargs):
        from .falcon_hip import FalconHIP
        from . import weights as W

        name = name or self._default_name
        if weights is not None:
            assert shape is not None, "pass shape= together with weights="
        elif os.path.isdir(name):
            shape = shape or FalconShape.from_hf_dir(name)
            weights = W.load_falcon_safetensors(name, shape)
            if tokenizer is None:
                from transformers import AutoTokenizer
                tokenizer = AutoTokenizer.from_pretrained(name)
        else:
            key = {"tiiuae/falcon-7b-instruct": "falcon-7b", "tiiuae/falcon-7b": "falcon-7b"}.get(name, name)
            if shape is None:
                if key not in FALCON_SHAPES:
                    raise ValueError(f"unknown model {name!r}: pass a checkpoint directory or one of {sorted(FALCON_SHAPES)}")
                shape = FALCON_SHAPES[key]
            if not random_init:
                raise FileNotFoundError(
                    f"no checkpoint directory {name!r} (there is no network here); pass random_init=True to use "
                    f"seeded N(0, {shape.initializer_range}) weights at the {shape.name} shape")
            weights = W.random_falcon_weights_device(shape, device, torch.float16, seed)
        if tokenizer is None:
            tokenizer = StandInTokenizer(shape.vocab_size)
        model = FalconHIP(shape, weights, device=device, int8_weights=bool(_hf_kwargs.get("load_in_8bit", False)))
        self.formatter = _falcon_formatter()
        super().__init__(name, model, tokenizer, list(range(12)), ["<|endoftext|>", "\nUser"])

    def get_formatter(self) -> Callable[[str], str]:
        return self.formatter

    def get_cache_shape(self) -> Tuple[int, int, int]:
        c = self.hf_model.config
        return c.num_hidden_layers, 1, c.head_dim


class Mpt(LanguageModel):
    """Reference ``Mpt`` adapter (:261-288): ChatML-style chat strings, stop tokens [50278, 0], and
    ``use_full_position_ids = True`` -- drivers pass ``return_full_position_ids`` / ``use_full_position_ids`` to the
    engines (demo.py:81-84) so the model sees the position id of every cached key (ALiBi, mpt.py:172)."""

    _default_name = "mosaicml/mpt-7b-chat"
    use_full_position_ids = True

    def __init__(self, name: Optional[str] = None, device: str = "cuda:0", shape: Optional[MptShape] = None,
                 weights=None, tokenizer=None, random_init: bool = False, seed: int = 0, **_hf_kwargs):
        from .mpt_hip import MptHIP
        from . import weights as W

        name = name or self._default_name
        if weights is not None:
            assert shape is not None, "pass shape= together with weights="
        elif os.path.isdir(name):
            shape = shape or MptShape.from_hf_dir(name)
            weights = W.load_mpt_safetensors(name, shape)
            if tokenizer is None:
                from transformers import AutoTokenizer
                tokenizer = AutoTokenizer.from_pretrained(name)
        else:
            key = {"mosaicml/mpt-7b-chat": "mpt-7b", "mosaicml/mpt-7b": "mpt-7b", "mosaicml/mpt-7b-instruct": "mpt-7b"}.get(name, name)
            if shape is None:
                if key not in MPT_SHAPES:
                    raise ValueError(f"unknown model {name!r}: pass a checkpoint directory or one of {sorted(MPT_SHAPES)}")
                shape = MPT_SHAPES[key]
            if not random_init:
                raise FileNotFoundError(
                    f"no checkpoint directory {name!r} (there is no network here); pass random_init=True to use "
                    f"seeded N(0, {shape.initializer_range}) weights at the {shape.name} shape")
            weights = W.random_mpt_weights_device(shape, device, torch.float16, seed)
        if tokenizer is None:
            tokenizer = StandInTokenizer(shape.vocab_size)
        model = MptHIP(shape, weights, device=device, int8_weights=bool(_hf_kwargs.get("load_in_8bit", False)))
        self.formatter = _mpt_formatter()
        super().__init__(name, model, tokenizer, [50278, 0], [])

    def get_formatter(self) -> Callable[[str], str]:
        return self.formatter

    def get_cache_shape(self) -> Tuple[int, int, int]:
        c = self.hf_model.config
        return c.num_hidden_layers, c.num_attention_heads, c.head_dim
