"""Import the *reference* implementation (read-only checkout at /root/reference) in the build
container, with harness-side shims only -- the reference files are untouched and never copied.

TEST INFRASTRUCTURE (used by ``oracle/gen_golden.py`` to produce the committed fixtures under
``tests/golden/``).  Nothing here exists on the GPU box; nothing in ``-m gpu`` tests, ``smoke()`` or
``bench.py`` imports this module.

Shims:
  * ``lxml`` / ``lxml.etree`` -> stdlib ElementTree with comments kept as child nodes (as lxml keeps them);
    ``XMLParser(recover=True)`` is emulated by a pre-pass through the system libxml2 (``xmlReadMemory`` with
    XML_PARSE_RECOVER), the same library lxml wraps, then a strict ElementTree parse of the repaired dump;
    ``tostring`` returns bytes with the tail, like lxml's;
  * ``termcolor`` -> identity ``colored``;
  * ``transformers.utils.is_flash_attn_available`` -> ``False`` (needed by the falcon import chain);
  * ``torch.cuda.Event`` / ``torch.cuda.synchronize`` -> perf_counter based fakes (no GPU here).
"""
from __future__ import annotations

import ctypes
import ctypes.util
import importlib
import sys
import time
import types
import xml.etree.ElementTree as ET

REFERENCE_ROOT = "/root/reference"


def libxml2_recover(text: str) -> str:
    """Return ``text`` re-serialised by libxml2 after a recovering parse."""
    lib = ctypes.CDLL(ctypes.util.find_library("xml2") or "libxml2.so.2")
    lib.xmlReadMemory.restype = ctypes.c_void_p
    lib.xmlReadMemory.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]
    lib.xmlDocDumpMemoryEnc.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_int),
                                        ctypes.c_char_p]
    lib.xmlFreeDoc.argtypes = [ctypes.c_void_p]
    data = text.encode("utf-8")
    XML_PARSE_RECOVER, XML_PARSE_NOERROR, XML_PARSE_NOWARNING = 1, 32, 64
    doc = lib.xmlReadMemory(data, len(data), b"noname.xml", b"UTF-8",
                            XML_PARSE_RECOVER | XML_PARSE_NOERROR | XML_PARSE_NOWARNING)
    if not doc:
        raise ValueError("libxml2 could not recover the document")
    out = ctypes.c_char_p()
    n = ctypes.c_int()
    lib.xmlDocDumpMemoryEnc(doc, ctypes.byref(out), ctypes.byref(n), b"UTF-8")
    s = ctypes.string_at(out, n.value).decode("utf-8")
    lib.xmlFreeDoc(doc)
    if s.startswith("<?xml"):
        s = s[s.index("?>") + 2:].lstrip("\n")
    return s


class _XMLParser:
    def __init__(self, recover: bool = False, **_kw):
        self.recover = recover


def _et_parse(text: str):
    """ElementTree parse that KEEPS comments as child nodes, as lxml does (its XMLParser has remove_comments=False): a comment is
    an element whose ``tag`` is the ``ET.Comment`` factory -- not a string, exactly like ``lxml.etree.Comment`` -- so the
    reference's ``match e.tag`` falls through to ``case _`` (schema.py:362-363) and ``e.tag != "module"`` is true in a union
    (schema.py:207-208).  Comments outside the root element are dropped by the tree builder, as ``lxml.etree.fromstring`` returns
    the root element only."""
    return ET.fromstring(text, parser=ET.XMLParser(target=ET.TreeBuilder(insert_comments=True)))


def _fromstring(text, parser=None):
    if isinstance(text, bytes):
        text = text.decode("utf-8")
    try:
        return _et_parse(text)
    except ET.ParseError:
        if parser is not None and getattr(parser, "recover", False):
            return _et_parse(libxml2_recover(text))
        raise


def _tostring(e, **_kw):
    """``lxml.etree.tostring(e)`` with default arguments: BYTES (ASCII, character references for the rest), no declaration, the
    element's tail included; a comment renders as ``<!--text-->`` + tail.  ``ET.tostring`` does all of that for both node kinds
    (its serialiser writes ``<!--%s-->`` for ``ET.Comment`` nodes and appends the tail of the node it is given)."""
    return ET.tostring(e)


def install_shims() -> None:
    import torch
    import transformers.utils

    if "lxml" not in sys.modules:
        lxml = types.ModuleType("lxml")
        etree = types.ModuleType("lxml.etree")
        etree.XMLParser = _XMLParser
        etree.fromstring = _fromstring
        etree.tostring = _tostring
        etree.Element = ET.Element
        lxml.etree = etree
        sys.modules["lxml"] = lxml
        sys.modules["lxml.etree"] = etree
    if "termcolor" not in sys.modules:
        tc = types.ModuleType("termcolor")
        tc.colored = lambda s, *a, **k: s
        sys.modules["termcolor"] = tc
    if not hasattr(transformers.utils, "is_flash_attn_available"):
        transformers.utils.is_flash_attn_available = lambda: False

    if not torch.cuda.is_available():
        class _FakeEvent:
            def __init__(self, enable_timing=False):
                self.t = 0.0

            def record(self, *a):
                self.t = time.perf_counter()

            def elapsed_time(self, other):
                return (other.t - self.t) * 1e3

        torch.cuda.Event = _FakeEvent
        torch.cuda.synchronize = lambda *a, **k: None


def import_reference():
    """Returns the reference ``promptcache`` package (imported from /root/reference)."""
    install_shims()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        pc = importlib.import_module("promptcache")
    return pc


def make_reference_llama(cfg_dict: dict, weights_fp32: dict):
    """Instantiate the reference ``LlamaForCausalLM`` (promptcache/model/llama2.py:954) for a shape and
    load the given weights (our key layout -> HF parameter names)."""
    import torch
    import transformers

    import_reference()
    llama2 = importlib.import_module("promptcache.model.llama2")
    cfg = transformers.LlamaConfig(
        vocab_size=cfg_dict["vocab_size"], hidden_size=cfg_dict["hidden_size"],
        intermediate_size=cfg_dict["intermediate_size"], num_hidden_layers=cfg_dict["num_hidden_layers"],
        num_attention_heads=cfg_dict["num_attention_heads"], num_key_value_heads=cfg_dict["num_key_value_heads"],
        rms_norm_eps=cfg_dict["rms_norm_eps"], max_position_embeddings=cfg_dict["max_position_embeddings"],
        hidden_act="silu", pad_token_id=None, bos_token_id=1, eos_token_id=2, tie_word_embeddings=False)
    # transformers 5.x moved these into a rope_parameters dict; the 4.34-era reference reads attributes
    cfg.rope_theta = cfg_dict["rope_theta"]
    cfg.rope_scaling = None
    cfg.pretraining_tp = 1
    cfg.use_cache = True
    for k, v in (("output_attentions", False), ("output_hidden_states", False), ("use_return_dict", True)):
        if not hasattr(cfg, k):
            try:
                setattr(cfg, k, v)
            except Exception:
                pass
    model = llama2.LlamaForCausalLM(cfg)
    sd = {"model.embed_tokens.weight": weights_fp32["embed"], "model.norm.weight": weights_fp32["norm"],
          "lm_head.weight": weights_fp32["lm_head"]}
    names = {"ln1": "input_layernorm.weight", "wq": "self_attn.q_proj.weight", "wk": "self_attn.k_proj.weight",
             "wv": "self_attn.v_proj.weight", "wo": "self_attn.o_proj.weight", "ln2": "post_attention_layernorm.weight",
             "gate": "mlp.gate_proj.weight", "up": "mlp.up_proj.weight", "down": "mlp.down_proj.weight"}
    for i in range(cfg_dict["num_hidden_layers"]):
        for s, hf in names.items():
            sd[f"model.layers.{i}.{hf}"] = weights_fp32[f"l{i}.{s}"]
    sd = {k: torch.from_numpy(v.astype("float32")) for k, v in sd.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    missing = [m for m in missing if "rotary_emb" not in m]
    assert not missing and not unexpected, (missing, unexpected)
    model.eval()
    return model


def make_reference_falcon(cfg_dict: dict, weights_fp32: dict):
    """Instantiate the reference ``FalconForCausalLM`` (promptcache/model/falcon.py:1224) in the falcon-7b
    architecture (multi_query, parallel_attn, rotary, no bias) and load the given weights.

    Harness-side compatibility shims for transformers 5.x (the reference pins 4.34): ``rope_scaling`` /
    ``rope_theta`` attributes on the config (falcon.py:333-339 reads them), a dict-typed ``_tied_weights_keys``, and
    ``get_head_mask`` (removed from ``PreTrainedModel``; falcon.py:1108 only needs ``[None] * n_layers``)."""
    import torch
    import transformers

    import_reference()
    falcon = importlib.import_module("promptcache.model.falcon")
    cfg = transformers.FalconConfig(
        vocab_size=cfg_dict["vocab_size"], hidden_size=cfg_dict["hidden_size"],
        num_hidden_layers=cfg_dict["num_hidden_layers"], num_attention_heads=cfg_dict["num_attention_heads"],
        layer_norm_epsilon=cfg_dict["layer_norm_epsilon"], multi_query=True, parallel_attn=True, bias=False,
        new_decoder_architecture=False, alibi=False, tie_word_embeddings=False, bos_token_id=1, eos_token_id=2)
    cfg.rope_theta = cfg_dict["rope_theta"]
    cfg.rope_scaling = None
    cfg.use_cache = True
    falcon.FalconForCausalLM._tied_weights_keys = {}
    if not hasattr(falcon.FalconModel, "get_head_mask"):
        falcon.FalconModel.get_head_mask = lambda self, head_mask, n, **_k: [None] * n
    model = falcon.FalconForCausalLM(cfg)
    sd = {"transformer.word_embeddings.weight": weights_fp32["embed"], "transformer.ln_f.weight": weights_fp32["lnf_w"],
          "transformer.ln_f.bias": weights_fp32["lnf_b"], "lm_head.weight": weights_fp32["lm_head"]}
    names = {"ln_w": "input_layernorm.weight", "ln_b": "input_layernorm.bias",
             "wqkv": "self_attention.query_key_value.weight", "wo": "self_attention.dense.weight",
             "w1": "mlp.dense_h_to_4h.weight", "w2": "mlp.dense_4h_to_h.weight"}
    for i in range(cfg_dict["num_hidden_layers"]):
        for s_, hf in names.items():
            sd[f"transformer.h.{i}.{hf}"] = weights_fp32[f"l{i}.{s_}"]
    sd = {k: torch.from_numpy(v.astype("float32")) for k, v in sd.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    missing = [m for m in missing if "rotary" not in m and "inv_freq" not in m]
    assert not missing and not unexpected, (missing, unexpected)
    model.eval()
    return model


def make_reference_mpt(cfg_dict: dict, weights_fp32: dict):
    """Instantiate the reference ``MptForCausalLM`` (promptcache/model/mpt.py:594) and load the given weights.
    Harness-side shim for transformers 5.x: dict-typed ``_tied_weights_keys``."""
    import torch
    import transformers

    import_reference()
    mpt = importlib.import_module("promptcache.model.mpt")
    cfg = transformers.MptConfig(d_model=cfg_dict["hidden_size"], n_heads=cfg_dict["num_attention_heads"],
                                 n_layers=cfg_dict["num_hidden_layers"], vocab_size=cfg_dict["vocab_size"],
                                 max_seq_len=8192, layer_norm_epsilon=cfg_dict["layer_norm_epsilon"], no_bias=True,
                                 tie_word_embeddings=False)
    cfg.attn_config.alibi_bias_max = cfg_dict["alibi_bias_max"]
    cfg.use_cache = True
    mpt.MptForCausalLM._tied_weights_keys = {}
    model = mpt.MptForCausalLM(cfg)
    sd = {"transformer.wte.weight": weights_fp32["embed"], "transformer.norm_f.weight": weights_fp32["lnf"],
          "lm_head.weight": weights_fp32["lm_head"]}
    names = {"ln1": "norm_1.weight", "wqkv": "attn.Wqkv.weight", "wo": "attn.out_proj.weight", "ln2": "norm_2.weight",
             "w1": "ffn.up_proj.weight", "w2": "ffn.down_proj.weight"}
    for i in range(cfg_dict["num_hidden_layers"]):
        for s_, hf in names.items():
            sd[f"transformer.blocks.{i}.{hf}"] = weights_fp32[f"l{i}.{s_}"]
    sd = {k: torch.from_numpy(v.astype("float32")) for k, v in sd.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    model.eval()
    return model
