"""Deterministic stand-in tokenizer (no tokenizer.model / network in the build or GPU image).

The reference obtains token ids from HF tokenizers (``promptcache/model/__init__.py:131-137``:
``encode`` without BOS, ``decode`` with special tokens kept).  When a real HF tokenizer directory is
available the adapters in ``promptcache_amd.model`` use it; otherwise this stand-in provides
Llama-like granularity (~4 characters / token of English) with the two properties the cache engine
relies on:

* ``encode`` never returns an empty list for non-empty text -- whitespace-only text yields one token,
  as SentencePiece does (the reference's ``position_ids.index(offset)`` at ``cache_engine.py:278``
  depends on every TokenSequence having at least one token);
* it is a pure function of the text (stable ids across processes / machines: CRC32, not ``hash``).
"""
from __future__ import annotations

import re
import zlib
from typing import Dict, List

_PIECE = re.compile(
    r"</s>|<s>"                 # literal special tokens, as the Llama tokenizer parses them
    r"|\s?[A-Za-z]+"            # a word with its leading space
    r"|\s?[0-9]"                # digits one by one (Llama splits numbers per digit)
    r"|\s?[^\sA-Za-z0-9]"       # one punctuation mark
    r"|\s+"                     # remaining whitespace run
)

_RESERVED = 259  # 0 unk, 1 bos, 2 eos, 3..258 byte-fallback range (unused here)


class StandInTokenizer:
    unk_token = "<unk>"
    bos_token = "<s>"
    eos_token = "</s>"
    unk_token_id = 0
    bos_token_id = 1
    eos_token_id = 2

    def __init__(self, vocab_size: int = 32000, max_piece_chars: int = 5):
        if vocab_size <= _RESERVED + 1:
            raise ValueError("vocab_size too small for the stand-in tokenizer")
        self.vocab_size = vocab_size
        self.max_piece_chars = max_piece_chars
        self._rev: Dict[int, str] = {0: "<unk>", 1: "<s>", 2: "</s>"}

    def _id(self, piece: str) -> int:
        if piece == "<s>":
            return 1
        if piece == "</s>":
            return 2
        tid = _RESERVED + zlib.crc32(piece.encode("utf-8")) % (self.vocab_size - _RESERVED)
        self._rev.setdefault(tid, piece)
        return tid

    def encode(self, text: str, add_special_tokens: bool = False) -> List[int]:
        if not isinstance(text, str):
            # the contract of the HF (Python) tokenizers the reference loads (LlamaTokenizer / CodeLlamaTokenizer,
            # promptcache/model/__init__.py:167,188; transformers 4.34 PreTrainedTokenizer._encode_plus.get_input_ids): anything
            # that is not a string or a list of strings / ints is refused -- the reference reaches this with the BYTES that
            # lxml.etree.tostring returns for an XML comment or unknown tag among a module's children (schema.py:362-363)
            raise ValueError(f"Input {text} is not valid. Should be a string, a list/tuple of strings or a list/tuple of integers.")
        ids: List[int] = [1] if add_special_tokens else []
        n = self.max_piece_chars
        for m in _PIECE.finditer(text):
            p = m.group(0)
            if len(p) > n and p not in ("<s>", "</s>") and not p.isspace():
                # long words -> sub-word pieces, the first keeps the leading space
                for i in range(0, len(p), n):
                    ids.append(self._id(p[i:i + n]))
            else:
                ids.append(self._id(p))
        return ids

    def decode(self, token_ids, skip_special_tokens: bool = False, **_kw) -> str:
        out = []
        for t in token_ids:
            t = int(t)
            if skip_special_tokens and t in (0, 1, 2):
                continue
            out.append(self._rev.get(t, f" t{t}"))
        return "".join(out)
