"""Generate the committed golden fixtures by running the REFERENCE implementation (imported from
/root/reference via ``oracle/ref_shim.py``) in the build container.

    python oracle/gen_golden.py            # rewrites tests/golden/*
    python oracle/gen_golden.py --layout-only   # only pml_layout.json / pml_recover.json / pml/*.xml

TEST INFRASTRUCTURE.  The reference never ships to the GPU box: only these fixtures (data: inputs and
expected outputs) and this script are committed.  Inputs are reproducible without the reference:
weights come from ``promptcache_amd.model.weights.make_weights_np(shape, seed, scale)`` (numpy PCG64),
token ids from the deterministic stand-in tokenizer, PML text from files under ``tests/golden/pml/``
(written by this script; the three synthetic schemas are authored here, not taken from the reference).

Fixtures
  pml_layout.json    integer layout of every parseable reference schema (examples/ and
                     benchmark/schema/test/): schema length, encode paths, per-scaffold CRC32 of token /
                     position ids, prompt assembly results (process() with and without cache)
  pml_recover.json   libxml2 recovery behaviour on malformed snippets (pins pml_xml)
  model_<case>.npz   tensors from the reference CacheEngine / GenerationEngine / LlamaForCausalLM on CPU:
                     cached-prefill logits, no-cache logits, layer-0 attention output, sampled rows of the
                     staged KV and of the stored module KV, greedy tokens, inv_freq
"""
from __future__ import annotations

import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "prompt-cache_amd"))

from oracle import ref_shim  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
REF = ref_shim.REFERENCE_ROOT


def crc(xs) -> int:
    return zlib.crc32(np.asarray(list(xs), dtype=np.int64).tobytes())


# ---------------------------------------------------------------------------------------------------
# synthetic PML (authored for this repo) -- small enough that tensor fixtures stay tiny
# ---------------------------------------------------------------------------------------------------

SYN_UNION = """<schema name="trip">
    <system>You are a careful travel planner. Answer briefly.</system>
    <user>
        Plan a trip with the following constraints.
        <module name="length">
            <union scaffold="weekend">
                <module name="weekend">The trip lasts two days, Saturday and Sunday, with one night away.</module>
                <module name="week">The trip lasts seven days and may include two different cities.</module>
            </union>
        </module>
        <module name="budget">
            The budget is <parameter name="amount" length="6" scaffold="unknown"/> dollars in total,
            <union>
                <module name="frugal">and every meal should be cheap street food.</module>
                <module name="lavish">and at least one dinner should be a tasting menu
                    <module name="wine">with paired wines</module>.
                </module>
            </union>
        </module>
        <module name="pets">A small dog travels with us.</module>
    </user>
</schema>
"""
SYN_UNION_PROMPT = """<prompt schema='trip'>
    <length><week/></length>
    <budget amount="1500"><lavish><wine/></lavish></budget>
    <user>Where should we go in October?</user>
</prompt>"""
SYN_UNION_PROMPT2 = """<prompt schema='trip'>
    <pets/>
    <budget amount="90"><frugal/></budget>
    <user>Suggest two places.</user>
</prompt>"""

SYN_FLAT = """<schema name="doc">
    <system>Answer using the documents.</system>
    <user>
        <module name="a">Document A. The river Ouse rises in the hills and runs east for ninety miles before it meets the sea near a small harbour town.</module>
        <module name="b">Document B. Copper prices were stable through the spring, then fell sharply once the new mine opened in the north.</module>
        <module name="c">Document C. The committee meets on the first Monday of each month unless that day is a public holiday.</module>
    </user>
</schema>
"""
SYN_FLAT_PROMPT = """<prompt schema='doc'><a/><b/><c/><user>When does the committee meet?</user></prompt>"""

# Children the reference serialises as text (schema.py:362-363): XML comments (under lxml a comment is a child node whose tag is not
# a string) and unknown elements go through ``lxml.etree.tostring`` -- which returns BYTES -- into ``lm.encode``; a comment inside a
# union fails the union's tag check (schema.py:207-208).  Authored here; what the reference does with them is recorded.
SYN_COMMENT_MODULE = """<schema name="c1">
    <module name="a">Alpha text. <!-- a note for the schema author --> Beta text.</module>
</schema>
"""
SYN_COMMENT_UNION = """<schema name="c2">
    <module name="m"><union><!-- pick one --><module name="x">X.</module><module name="y">Y.</module></union></module>
</schema>
"""
SYN_UNKNOWN_TAG = """<schema name="c3">
    <module name="a">Some <b>bold</b> words.</module>
</schema>
"""
SYN_COMMENT_OUTSIDE = """<!-- scenario: 1 -->
<schema name="c4">
    <module name="a">Alpha text.</module>
</schema>
<!-- trailing -->
"""
SYN_EXTRA = {"syn:comment_module": ("comment_module.xml", SYN_COMMENT_MODULE), "syn:comment_union": ("comment_union.xml", SYN_COMMENT_UNION),
             "syn:unknown_tag": ("unknown_tag.xml", SYN_UNKNOWN_TAG), "syn:comment_outside": ("comment_outside.xml", SYN_COMMENT_OUTSIDE)}

PERSONA_PROMPT = """<prompt schema='persona'>
    <age><young-adult/></age>
    <residence><seaside/></residence>
    <education><doctorate/></education>
    <occupation><technology/></occupation>
    <martial-status><married/></martial-status>
    <personality><introverted/></personality>
    <user>Introduce about yourself.</user>
</prompt>"""   # README.md:60-82 of the reference

GAME_PROMPT = """<prompt schema='code-generation-game'>
    <unit.py/><map.py/><player.py/><game.py/><database.py/>
    <user>Create a main entry for the game:</user>
</prompt>"""    # demo.py:66-77 of the reference


def build_ref_lm(pc, shape_name: str, seed: int, scale: float):
    import importlib
    from promptcache_amd.model.config import SHAPES
    from promptcache_amd.model.tokenizer import StandInTokenizer
    from promptcache_amd.model.weights import make_weights_np

    shape = SHAPES[shape_name]
    w16 = make_weights_np(shape, seed, scale)
    model = ref_shim.make_reference_llama(shape.to_dict(), {k: v.astype(np.float32) for k, v in w16.items()})
    rm = importlib.import_module("promptcache.model")

    class RefLM(rm.LanguageModel):
        def __init__(self):
            tok = StandInTokenizer(shape.vocab_size)
            super().__init__("ref", model, tok, [tok.eos_token_id], ["</s>"])
            self.formatter = rm.FormatConversation(system=("<s> [INST] <<SYS>>\n", "<</SYS>>\n\n", "<s> [INST] "),
                                                   user=("", "[/INST]"), assistant=("", "</s><s> [INST] "))

        def get_formatter(self):
            return self.formatter

    return RefLM(), shape


def build_ref_falcon_lm(pc, shape_name: str, seed: int, scale: float):
    """The reference ``Falcon`` adapter surface (promptcache/model/__init__.py:206-258: formatter, stop strings,
    ``get_cache_shape`` = (L, 1, D)) around the reference ``FalconForCausalLM`` with seeded weights."""
    import importlib
    from promptcache_amd.model.config import FALCON_SHAPES
    from promptcache_amd.model.tokenizer import StandInTokenizer
    from promptcache_amd.model.weights import make_falcon_weights_np

    shape = FALCON_SHAPES[shape_name]
    w16 = make_falcon_weights_np(shape, seed, scale)
    model = ref_shim.make_reference_falcon(shape.to_dict(), {k: v.astype(np.float32) for k, v in w16.items()})
    rm = importlib.import_module("promptcache.model")
    rprompt = importlib.import_module("promptcache.prompt")

    class RefFalconLM(rm.LanguageModel):
        def __init__(self):
            tok = StandInTokenizer(shape.vocab_size)
            super().__init__("ref-falcon", model, tok, list(range(12)), ["<|endoftext|>", "\nUser"])
            conv = rm.FormatConversation(system=("", "\n\n", ""), user=("User: ", "\n\nAssistant:"), assistant=(" ", "\n\n"))
            self.formatter = rprompt.PreprocessorList([lambda t: t.replace("\r\n", "\n").replace("\n\n", "\n"), conv])

        def get_formatter(self):
            return self.formatter

        def get_cache_shape(self):
            c = self.hf_model.config
            return c.num_hidden_layers, 1, c.hidden_size // c.num_attention_heads

    return RefFalconLM(), shape


def build_ref_mpt_lm(pc, shape_name: str, seed: int, scale: float):
    """The reference ``Mpt`` adapter surface (promptcache/model/__init__.py:261-288: chat strings,
    ``use_full_position_ids = True``, cache shape (L, H, D)) around the reference ``MptForCausalLM``."""
    import importlib
    from promptcache_amd.model.config import MPT_SHAPES
    from promptcache_amd.model.tokenizer import StandInTokenizer
    from promptcache_amd.model.weights import make_mpt_weights_np

    shape = MPT_SHAPES[shape_name]
    w16 = make_mpt_weights_np(shape, seed, scale)
    model = ref_shim.make_reference_mpt(shape.to_dict(), {k: v.astype(np.float32) for k, v in w16.items()})
    rm = importlib.import_module("promptcache.model")

    class RefMptLM(rm.LanguageModel):
        def __init__(self):
            tok = StandInTokenizer(shape.vocab_size)
            super().__init__("ref-mpt", model, tok, [50278, 0], [])
            self.formatter = rm.FormatConversation(system=("<|im_start|>system\n", "<|im_end|>\n", ""),
                                                   user=("<|im_start|>user\n", "<|im_end|>\n<|im_start|>assistant\n"),
                                                   assistant=("", "<|im_end|>\n"))
            self.use_full_position_ids = True

        def get_formatter(self):
            return self.formatter

        def get_cache_shape(self):
            c = self.hf_model.config
            return c.n_layers, c.n_heads, c.d_model // c.n_heads

    return RefMptLM(), shape


class TokOnlyLM:
    """Tokenizer-only stand-in for layout goldens (no model needed to lay out a schema)."""

    def __init__(self, vocab=32000):
        from promptcache_amd.model.tokenizer import StandInTokenizer
        self.tok = StandInTokenizer(vocab)
        self.unk_token_id = 0
        self.eos_token_id = 2

    def encode(self, text):
        return self.tok.encode(text)


def layout_goldens(pc):
    import importlib
    rm = importlib.import_module("promptcache.model")
    rs = importlib.import_module("promptcache.schema")
    rp = importlib.import_module("promptcache.prompt")
    rce = importlib.import_module("promptcache.cache_engine")
    fmt = rm.FormatConversation(system=("<s> [INST] <<SYS>>\n", "<</SYS>>\n\n", "<s> [INST] "),
                                user=("", "[/INST]"), assistant=("", "</s><s> [INST] "))
    lm = TokOnlyLM()
    files = [("examples/persona_generation.xml", None), ("examples/code_generation_game.xml", 800),
             ("examples/code_generation_bookstore.xml", None), ("examples/parameterized_prompts.xml", None),
             ("examples/personalization-education.xml", None), ("benchmark/schema/test/schema_mbti.xml", None),
             ("benchmark/schema/test/schema_mbti_short.xml", None), ("benchmark/schema/test/schema_persona.xml", None),
             ("benchmark/schema/test/schema_persona_long.xml", None), ("benchmark/schema/test/empty.xml", None),
             # the two reference schemas that carry XML comments among module children (schema.py:362-363)
             ("benchmark/schema/test/schema_code_generation.xml", None), ("benchmark/schema/test/schema_long_task_1.xml", None),
             ("syn:trip", None), ("syn:doc", None), ("syn:trip", 12)] + [(k, None) for k in SYN_EXTRA]
    out = {}
    for fn, max_tokens in files:
        if fn.startswith("syn:"):
            text = fmt({"syn:trip": SYN_UNION, "syn:doc": SYN_FLAT, **{k: v[1] for k, v in SYN_EXTRA.items()}}[fn])
        else:
            text = rp.read_file(os.path.join(REF, fn), [fmt])
        key = fn + (f"@{max_tokens}" if max_tokens else "")
        try:
            sc = rs.Schema(text, lm, max_tokens=max_tokens)
        except Exception as e:  # schema the reference itself rejects
            out[key] = {"error": type(e).__name__, "message": str(e)}
            continue
        # L1 path enumeration exactly as SchemaCache._process does it (cache_engine.py:188-210)
        stack, paths = [], [rs.Path()]
        if sc.contains_union():
            stack.append((list(), True, sc))
        while stack:
            path, is_default_parent, u = stack.pop()
            for e in u.children:
                if type(e) == rs.Module and e.contains_union():
                    stack.append((path + [u.name], is_default_parent, e))
                elif type(e) == rs.UnionModule:
                    for n in e.modules:
                        is_default = e.scaffold_name == n.name and is_default_parent
                        if n.contains_union():
                            stack.append((path + [u.name], is_default, n))
                        if not is_default:
                            paths.append(rs.Path(path + [u.name, n.name]).next)
        rec = {"name": sc.name, "length": len(sc), "paths": [], "max_tokens": max_tokens}
        for p in paths:
            sf = sc.get_scaffold(p)
            tgt = sf.select(p).all_token_sequences()
            rec["paths"].append({"path": str(p), "n": len(sf.token_ids()), "ids_crc": crc(sf.token_ids()),
                                 "pos_crc": crc(sf.position_ids()),
                                 "targets": [[t.offset, len(t)] for t in tgt]})
        out[key] = rec
    # prompt assembly through the reference CacheEngine.process (no model: no_cache schema caches)
    prompts = {"examples/persona_generation.xml": PERSONA_PROMPT, "examples/code_generation_game.xml@800": GAME_PROMPT,
               "syn:trip": SYN_UNION_PROMPT, "syn:trip#2": SYN_UNION_PROMPT2, "syn:doc": SYN_FLAT_PROMPT}

    class NoModelLM(TokOnlyLM):
        device = "cpu"

        def get_cache_shape(self):
            return 1, 1, 8

    for key, ptxt in prompts.items():
        skey = key.split("#")[0]
        fn, _, mt = skey.partition("@")
        max_tokens = int(mt) if mt else None
        text = fmt({"syn:trip": SYN_UNION, "syn:doc": SYN_FLAT}[fn]) if fn.startswith("syn:") else \
            rp.read_file(os.path.join(REF, fn), [fmt])
        nlm = NoModelLM()
        eng = rce.CacheEngine(64, nlm, target_device="cpu")
        eng.add_schema(text, max_tokens=max_tokens, no_cache=True)
        prompt = rp.Prompt(ptxt, [fmt])
        ids, pos, _, _ = eng.process(prompt, no_cache=True)
        # the cached branch needs cache_l1; replay its integer part only (used sequences, args, text)
        used, arg_ids, arg_pos = [], [], []
        sc = eng.get_schema(prompt.schema)
        stack = [(prompt, sc)]
        while stack:
            ref, module = stack.pop()
            used += [(m.offset, len(m)) for m in module.token_sequences()]
            for arg in ref.args:
                prm = [p for p in module.parameters() if p.name == arg.name][0]
                a = nlm.encode(arg.value)
                arg_ids += a
                arg_pos += prm.position_ids()[:len(a)]
            for m in ref.modules:
                stack.append((m, module.select(m.name)))
        if len(prompt.text) > 0:
            t = nlm.encode(prompt.text)
            arg_ids += t
            arg_pos += list(range(len(sc), len(sc) + len(t)))
        out.setdefault("prompts", {})[key] = {
            "prompt": ptxt, "text": prompt.text, "nocache_n": len(ids), "nocache_ids_crc": crc(ids),
            "nocache_pos_crc": crc(pos), "used": used, "new_ids": arg_ids, "new_pos": arg_pos}
    return out


# ---------------------------------------------------------------------------------------------------
# PML fuzz: random schemas + prompts (generated HERE, seeded), laid out by the imported reference
# ---------------------------------------------------------------------------------------------------

_FWORDS = ("alpha beta gamma delta kilo lima mike echo seven eleven 42 x y zebra quick brown fox jumps over the lazy dog "
           "if a <= b: return c&d \"quoted\" it's 3.14 [INST] </s> résumé naïve <- -> != ==").split()


def _fuzz_text(rnd, lo=1, hi=12):
    ws = rnd.choice(["", " ", "  ", "\n", "\n    ", " \n\t "])
    we = rnd.choice(["", " ", "\n", "\n  ", "   "])
    words = [rnd.choice(_FWORDS) for _ in range(rnd.randint(lo, hi))]
    sep = rnd.choice([" ", " ", "  ", "\n"])
    txt = sep.join(words)
    if rnd.random() < 0.7:                       # mostly well-formed text; sometimes raw '<' / '&' for the recovering parser
        txt = txt.replace("&", "&amp;").replace("<", "&lt;").replace(">", "&gt;")
    return ws + txt + we


def _fuzz_module(rnd, names, depth, allow_param=True):
    name = rnd.choice(["m", "mod", "Doc", "a_b", "x-1", "p.q", "_u"]) + str(len(names))
    names.append(name)
    attrs = f' name="{name}"' + (rnd.choice(["", ' cache="false"', ' cache="true"']) if rnd.random() < 0.2 else "")
    parts, kids, params = [], [], []
    for _ in range(rnd.randint(0, 3 if depth < 2 else 1)):
        r = rnd.random()
        if r < 0.45:
            parts.append(_fuzz_text(rnd))
        elif r < 0.65 and allow_param:
            pn = "arg" + str(len(names)) + str(len(params))
            params.append((pn, rnd.randint(1, 8)))
            sc = rnd.choice(["", f' scaffold="{rnd.choice(_FWORDS[:8])}"', ' scaffold="one two three four five six seven eight nine"'])
            parts.append(f'<parameter name="{pn}" length="{params[-1][1]}"{sc}/>')
        elif r < 0.85 and depth < 3:
            sub, info = _fuzz_module(rnd, names, depth + 1)
            parts.append(sub)
            kids.append(info)
        elif depth < 3:
            members = []
            minfo = []
            for _ in range(rnd.randint(1, 3)):
                sub, info = _fuzz_module(rnd, names, depth + 1)
                members.append(sub)
                minfo.append(info)
            sc = f' scaffold="{minfo[rnd.randrange(len(minfo))]["name"]}"' if rnd.random() < 0.5 else ""
            gap = rnd.choice(["", "\n", " "])
            parts.append(f"<union{sc}>{gap}" + gap.join(members) + f"{gap}</union>")
            kids.append({"union": minfo})
        parts.append(rnd.choice(["", "", " ", "\n    ", _fuzz_text(rnd, 1, 4)]))
    return f"<module{attrs}>" + "".join(parts) + "</module>", {"name": name, "kids": kids, "params": params}


def _fuzz_prompt(rnd, schema_name, tops):
    """A prompt over the schema: picks some top-level modules (one member per union), fills some parameters, trailing text."""
    def ref(info):
        args = "".join(f' {pn}="{rnd.choice(_FWORDS[:14])}{" more words here" if rnd.random() < 0.15 else ""}"'
                       for pn, _ in info["params"] if rnd.random() < 0.7)
        inner = ""
        for k in info["kids"]:
            if "union" in k:
                if rnd.random() < 0.8:
                    inner += ref(rnd.choice(k["union"]))
            elif rnd.random() < 0.6:
                inner += ref(k)
        return f"<{info['name']}{args}>{inner}</{info['name']}>" if inner else f"<{info['name']}{args}/>"
    body = ""
    for t in tops:
        if "union" in t:
            if rnd.random() < 0.8:
                body += ref(rnd.choice(t["union"]))
        elif rnd.random() < 0.75:
            body += ref(t)
        body += rnd.choice(["", "\n", "  "])
    tail = rnd.choice(["", " What now? ", "\n  <user>Please answer.</user>\n", " tail text "])
    if not body and not tail.strip():
        tail = " only text "
    return f"<prompt schema='{schema_name}'>{body}{tail}</prompt>"


def fuzz_goldens(pc, count=80, seed=20260929):
    """``count`` random PML schemas (nested modules, unions with / without scaffold, parameters, raw whitespace, escapes and the odd
    malformed character) each with two random prompts, laid out by the REFERENCE: schema length, encode paths, per-scaffold CRCs,
    prompt assembly -- or the exception the reference raises.  The texts are generated here (seeded), not reference text."""
    import importlib
    import random
    rm = importlib.import_module("promptcache.model")
    rs = importlib.import_module("promptcache.schema")
    rp = importlib.import_module("promptcache.prompt")
    rce = importlib.import_module("promptcache.cache_engine")
    fmt = rm.FormatConversation(system=("<s> [INST] <<SYS>>\n", "<</SYS>>\n\n", "<s> [INST] "),
                                user=("", "[/INST]"), assistant=("", "</s><s> [INST] "))

    class NoModelLM(TokOnlyLM):
        device = "cpu"

        def get_cache_shape(self):
            return 1, 1, 8

    rnd = random.Random(seed)
    cases = []
    for ci in range(count):
        names, tops, parts = [], [], []
        parts.append(rnd.choice(["", "\n", "<system>You are a helpful assistant.</system>", "  <system/>  "]))
        wrap_user = rnd.random() < 0.5
        if wrap_user:
            parts.append("<user>")
        for _ in range(rnd.randint(1, 4)):
            r = rnd.random()
            if r < 0.25:
                parts.append(_fuzz_text(rnd))
            elif r < 0.8:
                sub, info = _fuzz_module(rnd, names, 1)
                parts.append(sub)
                tops.append(info)
            else:
                members, minfo = [], []
                for _ in range(rnd.randint(1, 3)):
                    sub, info = _fuzz_module(rnd, names, 1)
                    members.append(sub)
                    minfo.append(info)
                sc = f' scaffold="{minfo[0]["name"]}"' if rnd.random() < 0.5 else ""
                parts.append(f"<union{sc}>" + "\n".join(members) + "</union>")
                tops.append({"union": minfo})
            parts.append(rnd.choice(["", "\n", "    "]))
        if wrap_user:
            parts.append("</user>")
        if rnd.random() < 0.3:
            parts.append("<assistant>Sure.</assistant>")
        sname = f"fz{ci}"
        schema_text = f'<schema name="{sname}">' + "".join(parts) + "</schema>"
        max_tokens = rnd.choice([None, None, None, 6, 20])
        rec = {"schema": schema_text, "max_tokens": max_tokens, "prompts": []}
        lm = TokOnlyLM()
        try:
            sc = rs.Schema(fmt(schema_text), lm, max_tokens=max_tokens)
        except Exception as e:  # noqa: BLE001
            rec["error"] = {"type": type(e).__name__, "message": str(e)}
            cases.append(rec)
            continue
        stack, paths = [], [rs.Path()]
        if sc.contains_union():
            stack.append((list(), True, sc))
        while stack:
            path, is_default_parent, u = stack.pop()
            for e in u.children:
                if type(e) == rs.Module and e.contains_union():
                    stack.append((path + [u.name], is_default_parent, e))
                elif type(e) == rs.UnionModule:
                    for n in e.modules:
                        is_default = e.scaffold_name == n.name and is_default_parent
                        if n.contains_union():
                            stack.append((path + [u.name], is_default, n))
                        if not is_default:
                            paths.append(rs.Path(path + [u.name, n.name]).next)
        rec["length"] = len(sc)
        rec["paths"] = []
        for p in paths:
            sf = sc.get_scaffold(p)
            sel = sf.select(p)
            # (select() returning None makes the reference's SchemaCache._process die with AttributeError at cache_engine.py:268:
            # recorded as targets = None)
            rec["paths"].append({"path": str(p), "n": len(sf.token_ids()), "ids_crc": crc(sf.token_ids()), "pos_crc": crc(sf.position_ids()),
                                 "targets": None if sel is None else [[t.offset, len(t)] for t in sel.all_token_sequences()]})
        for _ in range(2):
            ptxt = _fuzz_prompt(rnd, sname, tops)
            pr = {"prompt": ptxt}
            try:
                nlm = NoModelLM()
                eng = rce.CacheEngine(64, nlm, target_device="cpu")
                eng.add_schema(fmt(schema_text), max_tokens=max_tokens, no_cache=True)
                prompt = rp.Prompt(ptxt, [fmt])
                ids, pos, _, _ = eng.process(prompt, no_cache=True)
                pr.update(text=prompt.text, nocache_n=len(ids), nocache_ids_crc=crc(ids), nocache_pos_crc=crc(pos))
            except Exception as e:  # noqa: BLE001
                pr["error"] = {"type": type(e).__name__, "message": str(e)}
            rec["prompts"].append(pr)
        cases.append(rec)
    return cases


def recover_goldens():
    import xml.etree.ElementTree as ET
    snippets = ["<a>x < y</a>", "<a>x <= y</a>", "<a>1 <2 3</a>", "<a>p & q</a>", "<a>p &foo; q</a>", "<a>x <b>y</a>",
                "<a>a < b > c</a>", "<a>x<</a>", "<a>t</b></a>", "<a><b>t</a>", "<a>x <- y</a>",
                "<a>&lt;s&gt; [INST] &lt;&lt;SYS&gt;&gt;\n x &amp; y &quot;q&quot; &apos;z&apos; &#65;&#x42;</a>",
                "<s n='1'><m name=\"k\">if a <= b and c < d:\n    pass</m> tail <u/> t2</s>",
                # after the first error libxml2 2.9 drops predefined entities but keeps character references
                "<a>x < y &lt;b&gt; z &amp; w &quot;q&quot;</a>", "<a>&lt;b&gt; x < y &lt;b&gt;</a>",
                "<a><m>x < y</m><n>&lt;/s&gt;&lt;s&gt; [INST] &amp;</n></a>", "<a>x < y &#60;k&#62; &#x3c;</a>",
                "<a>if a <= b: &gt; &lt; x</a>", "<a>p &foo; q &lt; r</a>", "<a>p & q &gt; r</a>",
                # round 5 (found by the PML fuzz): unterminated references are dropped with the name / digits libxml2 had read
                "<a>c&d e</a>", "<a>R&D dept, AT&T</a>", "<a>c&d</a>", "<a>x&1y z</a>", "<a>x&#y z</a>", "<a>x&#12y z</a>", "<a>x&;y</a>",
                "<a>x &amp y</a>", "<a>x&lt y</a>", "<a>c&d-e.f:g_h i</a>", "<a>&d</a>", "<a>x&#x4g;</a>", "<a>x&#;y</a>", "<a>x&#x;y</a>",
                "<a>x&\u00e9 y</a>", "<a b='c&d e'>t</a>", "<a>x&d<b/>y</a>", "<a>x&a&b;c</a>",
                # ... ANY end tag closes the innermost element; what follows the root's end is ignored
                "<a>1</c>;</a>", "<a><b>x</c>y</b>z</a>", "<a><b>x</a>y", "<a><b><c>x</b>y</c>z</a>w", "<r><a>1</c>;</a>t<d/></r>",
                "<r>p<a>x</b>y</a>z<e/>q</r>", "<a>_<b/>x</</c>a</a>", "<a><d x='1'>:</&#65;& y='2';</a>", "<a><c>t</>u</a>",
                "<a><e>t</ y='2'<c/>u</a>", "<a><c>t</c x>u</a>",
                # ... and a start tag that cannot be finished is closed where it breaks
                "<a><_ x>t</_>u</a>", "<a><_ =>t</a>", "<a><_ <b/>t</a>", "<a><_ x='1' <b/>t</a>", "<a><_ x='1' y>t</_>u</a>",
                "<a><_ x='1'y='2'>t</_>u</a>", "<a><_ x=1>t</_>u</a>", "<a>q<_ x</a>", "<a><b/ >t</a>", "<a><b / >t</a>",
                "<r><a b=\"x<y\">t</a>u</r>", "<a><\u00e9<_>&lt;</c></a>", "<a b='1' c=\"2\"  d = '3' >t</a>",
                ]
    # round 6: character references outside XML's Char production are dropped (and count as the first error)
    extra = ["<r>a&#12;b&#0;c</r>", "<r>x&#x110000;y</r>", "<r>x&#xD800;y &lt; z</r>", "<r>x&#xFFFE;y</r>", "<r>a&#9;b&#10;c&#13;d</r>",
             "<r>x&#99999999999;y</r>", "<r>p&#1;q &amp; r</r>", "<r a='u&#0;v'>t</r>"]

    def dump(e):
        return {"tag": e.tag, "attrib": dict(e.attrib), "text": e.text, "tail": e.tail, "children": [dump(c) for c in e]}

    import random
    rnd = random.Random(20260929)                 # + 150 random malformed snippets (seeded; generated here)
    alphabet = ["word ", " ", "\n  ", "&amp;", "&lt;=", "R&D ", "a<b ", "x > y ", "<module name=\"m\">", "</module>", "<union>", "</union>",
                "<parameter name='p' length=\"3\"/>", "\"q\" ", "it's ", "</x>", "<m n=v>", "\u00e9 ", "&#233;", "&nbsp;", "</>", "<b/ >", "&#12", "&x"]
    while len(snippets) < 64 + 150:      # (64 hand-written + 150 random; `extra` goes behind them so the random ones stay as they were)
        cand = "<schema name='s'>" + "".join(rnd.choice(alphabet) for _ in range(rnd.randint(2, 10))) + "</schema>"
        try:
            ET.fromstring(ref_shim.libxml2_recover(cand))
        except Exception:  # noqa: BLE001  (libxml2 kept a duplicate attribute or produced nothing: not a tree to compare)
            continue
        snippets.append(cand)
    return [{"src": s, "tree": dump(ET.fromstring(ref_shim.libxml2_recover(s)))} for s in snippets + extra]


def model_golden(pc, case: str, shape_name: str, seed: int, scale: float, schema_text: str, prompt_text: str,
                 max_ctx: int, max_tokens=None, n_greedy: int = 4, family: str = "llama"):
    import importlib
    import torch
    rce = importlib.import_module("promptcache.cache_engine")
    rge = importlib.import_module("promptcache.generation_engine")
    rp = importlib.import_module("promptcache.prompt")
    lm, shape = {"falcon": build_ref_falcon_lm, "mpt": build_ref_mpt_lm, "llama": build_ref_lm}[family](pc, shape_name, seed, scale)
    full = bool(getattr(lm, "use_full_position_ids", False))     # MPT: position ids of every key (cache_engine.py:517-519)
    fmt = lm.get_formatter()
    eng = rce.CacheEngine(max_ctx, lm, target_device="cpu")
    eng.add_schema(fmt(schema_text), max_tokens=max_tokens)
    prompt = rp.Prompt(prompt_text, [fmt])
    rng = np.random.default_rng(1234)

    # ---- cached path ----
    ids, pos, _, cache = eng.process(prompt, no_cache=False, return_full_position_ids=full)
    S = cache[0][0].shape[1]
    staged_rows = np.sort(rng.choice(S, size=min(S, 48), replace=False))
    staged_k = np.stack([c[0][:, staged_rows].numpy() for c in cache])      # [L,H,rows,D] fp16
    staged_v = np.stack([c[1][:, staged_rows].numpy() for c in cache])
    staged_sum = np.array([[float(c[0].float().sum()), float(c[1].float().sum())] for c in cache])
    seg_table = [(m.token_sequence.offset, len(m)) for m in eng.prompt_cache.staged]
    past = [(k.unsqueeze(0), v.unsqueeze(0)) for k, v in cache]
    attn0 = {}
    if family == "falcon":
        layer0 = lm.hf_model.transformer.h[0].self_attention
        o_proj, rotary = layer0.dense, layer0.maybe_rotary
    elif family == "mpt":
        layer0 = lm.hf_model.transformer.blocks[0].attn
        o_proj, rotary = layer0.out_proj, None
    else:
        layer0 = lm.hf_model.model.layers[0].self_attn
        o_proj, rotary = layer0.o_proj, layer0.rotary_emb
    hook = o_proj.register_forward_pre_hook(lambda mod, inp: attn0.__setitem__("x", inp[0].detach().clone()))
    with torch.inference_mode():
        out = lm(input_ids=torch.tensor([ids]), position_ids=torch.tensor([pos]), past_key_values=past, use_cache=True)
    hook.remove()
    logits_cached = out.logits[0].numpy()
    new_k0 = out.past_key_values[0][0][0, :, S:].numpy()       # roped new keys of layer 0 (fp32)

    params = rge.GenerationParameters(temperature=0.0, max_new_tokens=n_greedy, stop_token_ids=[], stop_str=[])
    gen = rge.GenerationEngine(lm)
    toks = None
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        # fresh engine state for the generate run (process() increments usage counters, layout unchanged)
        ids2, pos2, _, cache2 = eng.process(prompt, no_cache=False, return_full_position_ids=full)
        outs = list(gen.generate(ids2, pos2, params, cache2, stream_interval=1, use_full_position_ids=full))
    # recover greedy token ids from the decoded stream is lossy; recompute them with the reference model
    with torch.inference_mode():
        pk = [(k.unsqueeze(0), v.unsqueeze(0)) for k, v in cache2]
        o = lm(input_ids=torch.tensor([ids2]), position_ids=torch.tensor([pos2]), past_key_values=pk, use_cache=True)
        toks = []
        offset = max(pos2) + 1
        pkv = o.past_key_values
        lg = o.logits
        for i in range(n_greedy):
            t = int(torch.argmax(lg[0, -1]))
            toks.append(t)
            if i == n_greedy - 1:
                break
            # generation_engine.py:127-132: full mode appends range(offset, offset + loop_index), else offset + loop_index
            step_pos = (pos2 + list(range(offset, offset + i + 1))) if full else [offset + i + 1]
            o = lm(input_ids=torch.tensor([[t]]), position_ids=torch.tensor([step_pos]), past_key_values=pkv,
                   use_cache=True)
            pkv, lg = o.past_key_values, o.logits
    assert lm.decode(toks) == outs[-1].new_text, (lm.decode(toks), outs[-1].new_text)

    # ---- no-cache path ----
    nids, npos, _, _ = eng.process(prompt, no_cache=True)
    with torch.inference_mode():
        out_nc = lm(input_ids=torch.tensor([list(nids)]), position_ids=torch.tensor([npos]), use_cache=True)
    logits_nc_last = out_nc.logits[0, -1].numpy()

    # ---- stored module KV (sampled) ----
    sc = eng.schemas[prompt.schema]
    mods = sorted(sc.cache_l1.values(), key=lambda m: (m.token_sequence.offset, len(m)))
    mod_table = [(m.token_sequence.offset, len(m)) for m in mods]
    def heads_first(t):     # multi-query stores are [len, D]: cache_engine.py:286-287 squeezes the single head away
        return t if t.dim() == 3 else t.unsqueeze(0)
    mod_k_first = np.stack([heads_first(m.host_cache[0][0])[:, 0].numpy() for m in mods])     # layer 0, first token  [M,H,D]
    mod_v_last = np.stack([heads_first(m.host_cache[-1][1])[:, -1].numpy() for m in mods])    # last layer, last token
    inv_freq = rotary.inv_freq.numpy() if rotary is not None else np.zeros(1, np.float32)

    np.savez_compressed(
        os.path.join(GOLD, f"model_{case}.npz"),
        shape_name=shape_name, seed=seed, scale=scale, max_ctx=max_ctx, max_tokens=-1 if max_tokens is None else max_tokens,
        schema_text=schema_text, prompt_text=prompt_text,
        input_ids=np.array(ids), position_ids=np.array(pos), seg_table=np.array(seg_table), S=S,
        staged_rows=staged_rows, staged_k=staged_k, staged_v=staged_v, staged_sum=staged_sum,
        logits_cached=logits_cached.astype(np.float32), attn0=attn0["x"][0].numpy().astype(np.float32),
        new_k0=new_k0.astype(np.float32), greedy=np.array(toks),
        nocache_ids=np.array(nids), nocache_pos=np.array(npos), logits_nocache_last=logits_nc_last.astype(np.float32),
        mod_table=np.array(mod_table), mod_k_first=mod_k_first.astype(np.float32), mod_v_last=mod_v_last.astype(np.float32),
        inv_freq=inv_freq.astype(np.float32), full_position_ids=int(full))
    print(f"[golden] model_{case}: S={S} q={len(ids)} greedy={toks} max|logit|={np.abs(logits_cached).max():.3f}")


def main():
    os.makedirs(GOLD, exist_ok=True)
    pc = ref_shim.import_reference()
    with open(os.path.join(GOLD, "pml_layout.json"), "w") as f:
        json.dump(layout_goldens(pc), f, indent=0, sort_keys=True)
    with open(os.path.join(GOLD, "pml_recover.json"), "w") as f:
        json.dump(recover_goldens(), f, indent=0)
    os.makedirs(os.path.join(GOLD, "pml"), exist_ok=True)
    for name, text in (("trip.xml", SYN_UNION), ("doc.xml", SYN_FLAT), *SYN_EXTRA.values()):
        with open(os.path.join(GOLD, "pml", name), "w") as f:
            f.write(text)
    with open(os.path.join(GOLD, "pml_fuzz.json"), "w") as f:
        json.dump(fuzz_goldens(pc), f, indent=0)
    if "--layout-only" in sys.argv:          # the integer-layout fixtures alone (seconds; the model fixtures take minutes)
        return
    model_golden(pc, "tiny_trip", "tiny", seed=0, scale=4.0, schema_text=SYN_UNION, prompt_text=SYN_UNION_PROMPT, max_ctx=256)
    model_golden(pc, "tiny_trip2", "tiny", seed=0, scale=4.0, schema_text=SYN_UNION, prompt_text=SYN_UNION_PROMPT2, max_ctx=256)
    model_golden(pc, "mid_trip", "mid", seed=1, scale=2.0, schema_text=SYN_UNION, prompt_text=SYN_UNION_PROMPT, max_ctx=300)
    model_golden(pc, "mid_mha_doc", "mid_mha", seed=2, scale=3.0, schema_text=SYN_FLAT, prompt_text=SYN_FLAT_PROMPT, max_ctx=200)
    # persona-structured synthetic schema (authored by promptcache_amd.synth, not reference text):
    # 3 traits x 3 members, whitespace 1-token segments between tags, 10 encode passes
    from promptcache_amd import synth
    sp, pp = synth.persona_like("persona", system_len=40, intro_len=20,
                                traits=(("age", (30, 26, 33)), ("home", (41, 37, 44)), ("job", (25, 29, 22))),
                                question_len=6, seed=5)
    model_golden(pc, "tiny_personalike", "tiny", seed=3, scale=4.0, schema_text=sp, prompt_text=pp, max_ctx=400)
    falcon_goldens(pc)
    mpt_goldens(pc)
    sampling_goldens(pc)
    write_manifest()


def write_manifest():
    """tests/golden/manifest.json: the key set of every .npz fixture as THIS script writes it.  tests/test_oracle_golden.py
    fails when a committed fixture and the manifest disagree -- a fixture written by an older revision of this script
    (a missing or surplus array) cannot pass for a changed reference."""
    import glob
    man = {}
    for fn in sorted(glob.glob(os.path.join(GOLD, "*.npz"))):
        with np.load(fn, allow_pickle=False) as z:
            man[os.path.basename(fn)] = sorted(z.files)
    man["_provenance"] = {
        "model fixtures (*.npz)": "captured from the imported reference (promptcache/ + its patched llama2.py / falcon.py / mpt.py) "
                                  "running on CPU in the build container, seeded random-init weights, stand-in tokenizer",
        "pml_layout.json, pml_recover.json, pml/*.xml": "the reference's schema.py / prompt.py run over an lxml STAND-IN "
            "(oracle/ref_shim.py: xml.etree.ElementTree + the system libxml2.so.2 through ctypes, XML_PARSE_RECOVER) because lxml "
            "is not installed in the image: libxml2's recovery behaviour is the real one (lxml wraps the same library), the tree "
            "API between it and the reference is the builder's -- a real-lxml pin is not possible offline",
        "LLM.int8": "no fixture: bitsandbytes==0.41.1 (requirements.txt) is absent and has no CPU path; oracle/llmint8_oracle.py "
                    "restates the published algorithm (parity unpinned for that mode)"}
    with open(os.path.join(GOLD, "manifest.json"), "w") as f:
        json.dump(man, f, indent=1, sort_keys=True)
    print(f"[golden] manifest: {len(man) - 1} fixtures")


SAMPLING_CASES = [
    # temperature, repetition_penalty, top_p, top_k   (generation_engine.py:22-42)
    (1.0, 1.0, 1.0, -1), (0.7, 1.0, 1.0, -1), (1.3, 1.2, 1.0, -1), (0.7, 1.0, 0.9, -1), (0.7, 1.0, 1.0, 5),
    (0.5, 1.3, 0.8, 7), (1.0, 1.0, 0.5, 1), (0.0, 1.0, 1.0, -1), (1.0, 1.0, 1e-9, -1), (2.0, 1.5, 0.95, 40),
]


def sampling_goldens(pc):
    """The reference's logits-processor chain (``GenerationParameters.get_logits_processor``, generation_engine.py:32-42,
    applied as at :150-155) on seeded logits and a token history: processed logits per parameter set."""
    import torch
    GP = pc.generation_engine.GenerationParameters
    rng = np.random.default_rng(77)
    V = 96
    logits = (3.0 * rng.standard_normal((len(SAMPLING_CASES), V))).astype(np.float32)
    history = rng.integers(0, V, size=(len(SAMPLING_CASES), 11))
    out = np.zeros_like(logits)
    for i, (t, rp, tp, tk) in enumerate(SAMPLING_CASES):
        params = GP(temperature=t, repetition_penalty=rp, top_p=tp, top_k=tk)
        chain = params.get_logits_processor()
        hist = torch.as_tensor([history[i].tolist()]) if rp > 1.0 else None
        out[i] = chain(hist, torch.from_numpy(logits[i:i + 1].copy()))[0].numpy()
    np.savez(os.path.join(GOLD, "sampling_chain.npz"), params=np.array(SAMPLING_CASES, dtype=np.float64), logits=logits,
             history=history, processed=out)
    print(f"[golden] sampling_chain: {len(SAMPLING_CASES)} parameter sets, V={V}")

def falcon_goldens(pc):
    """Falcon adapter fixtures (reference FalconForCausalLM, multi-query cache shape (L, 1, D))."""
    model_golden(pc, "falcon_tiny_trip", "falcon-tiny", seed=5, scale=4.0, schema_text=SYN_UNION, prompt_text=SYN_UNION_PROMPT,
                 max_ctx=256, family="falcon")
    model_golden(pc, "falcon_mid_doc", "falcon-mid", seed=6, scale=3.0, schema_text=SYN_FLAT, prompt_text=SYN_FLAT_PROMPT,
                 max_ctx=200, family="falcon")


def mpt_goldens(pc):
    """MPT adapter fixtures (reference MptForCausalLM: ALiBi bias gathered at the keys' position ids)."""
    model_golden(pc, "mpt_tiny_trip", "mpt-tiny", seed=7, scale=4.0, schema_text=SYN_UNION, prompt_text=SYN_UNION_PROMPT,
                 max_ctx=256, family="mpt")
    model_golden(pc, "mpt_mid_doc", "mpt-mid", seed=8, scale=3.0, schema_text=SYN_FLAT, prompt_text=SYN_FLAT_PROMPT,
                 max_ctx=200, family="mpt")


if __name__ == "__main__":
    if "--sampling-only" in sys.argv:
        sampling_goldens(ref_shim.import_reference())
        sys.exit(0)
    if "--mpt-only" in sys.argv:
        mpt_goldens(ref_shim.import_reference())
        sys.exit(0)
    if "--falcon-only" in sys.argv:      # add the Falcon fixtures without regenerating the others
        falcon_goldens(ref_shim.import_reference())
        sys.exit(0)
    if "--manifest-only" in sys.argv:
        write_manifest()
        sys.exit(0)
    main()
