"""The product's PML front-end (pml.py + pml_xml.py, no lxml) against the REFERENCE's schema / prompt
layout captured in tests/golden/pml_layout.json (integer work: bit-exact), and pml_xml's recovery against
libxml2's (tests/golden/pml_recover.json).

Reference schema files are read from /root/reference when it exists (build container); the synthetic
schemas (tests/golden/pml/*.xml) are checked everywhere.
"""
import json
import os
import zlib

import numpy as np
import pytest

from tests import helpers as H
from promptcache_amd import pml, pml_xml

REF = "/root/reference"
with open(os.path.join(H.GOLD, "pml_layout.json")) as f:
    LAYOUT = json.load(f)
with open(os.path.join(H.GOLD, "pml_recover.json")) as f:
    RECOVER = json.load(f)


def crc(xs):
    return zlib.crc32(np.asarray(list(xs), dtype=np.int64).tobytes())


def _schema_text(key):
    fn = key.split("@")[0].split("#")[0]
    fmt = H.llama_formatter()
    if fn.startswith("syn:"):
        with open(os.path.join(H.GOLD, "pml", {"syn:trip": "trip.xml", "syn:doc": "doc.xml"}.get(fn, fn[4:] + ".xml"))) as f:
            return fmt(f.read())
    path = os.path.join(REF, fn)
    if not os.path.exists(path):
        pytest.skip("reference checkout not present on this machine")
    return pml.read_file(path, [fmt])


SCHEMA_KEYS = [k for k in LAYOUT if k != "prompts"]


@pytest.mark.parametrize("key", SCHEMA_KEYS)
def test_schema_layout_matches_reference(key):
    rec = LAYOUT[key]
    text = _schema_text(key)
    lm = H.TokOnlyLM()
    if "error" in rec:
        # a schema the REFERENCE refuses (e.g. an XML comment among a module's children: lxml.etree.tostring hands the tokenizer
        # bytes, schema.py:362-363): the product must refuse it the same way -- same exception type, same message
        with pytest.raises(Exception) as ei:
            pml.Schema(text, lm, max_tokens=None)
        assert type(ei.value).__name__ == rec["error"] and str(ei.value) == rec["message"]
        return
    sc = pml.Schema(text, lm, max_tokens=rec["max_tokens"])
    assert sc.name == rec["name"]
    assert len(sc) == rec["length"]
    paths = sc.encode_paths()
    assert [str(p) for p in paths] == [p["path"] for p in rec["paths"]]
    for p, exp in zip(paths, rec["paths"]):
        sf = sc.get_scaffold(p)
        assert len(sf.token_ids()) == exp["n"]
        assert crc(sf.token_ids()) == exp["ids_crc"]
        assert crc(sf.position_ids()) == exp["pos_crc"]
        assert [[t.offset, len(t)] for t in sf.select(p).all_token_sequences()] == exp["targets"]


@pytest.mark.parametrize("key", sorted(LAYOUT["prompts"]))
def test_prompt_assembly_matches_reference(key):
    rec = LAYOUT["prompts"][key]
    text = _schema_text(key)
    lm = H.TokOnlyLM()
    mt = key.split("#")[0].partition("@")[2]
    sc = pml.Schema(text, lm, max_tokens=int(mt) if mt else None)
    prompt = pml.Prompt(rec["prompt"], [H.llama_formatter()])
    assert prompt.text == rec["text"]
    used, ids, pos = H.assemble(sc, prompt, lm)
    assert [[u.offset, len(u)] for u in used] == rec["used"]
    assert ids == rec["new_ids"] and pos == rec["new_pos"]
    # no_cache re-packing (cache_engine.py:476-493)
    pairs = sorted([pt for u in used for pt in zip(u.position_ids(), u.token_ids())] + list(zip(pos, ids)))
    assert len(pairs) == rec["nocache_n"]
    assert crc([t for _, t in pairs]) == rec["nocache_ids_crc"]
    assert crc(range(len(pairs))) == rec["nocache_pos_crc"]


def _dump(e):
    return {"tag": e.tag, "attrib": dict(e.attrib), "text": e.text, "tail": e.tail, "children": [_dump(c) for c in e]}


@pytest.mark.parametrize("i", range(len(RECOVER)))
def test_xml_recovery_matches_libxml2(i):
    rec = RECOVER[i]
    assert _dump(pml_xml.fromstring(rec["src"])) == rec["tree"]


def test_schema_and_prompt_errors():
    lm = H.TokOnlyLM()
    with pytest.raises(ValueError, match="Module name is missing"):
        pml.Schema("<schema>x</schema>", lm)
    with pytest.raises(ValueError, match="already defined"):
        pml.Schema('<schema name="s"><module name="a">x</module><module name="a">y</module></schema>', lm)
    with pytest.raises(ValueError, match="not allowed in schema"):
        pml.Schema('<schema name="s"><parameter name="p" length="3"/></schema>', lm)
    with pytest.raises(ValueError, match="too long"):
        pml.Schema('<schema name="s"><module name="a"><parameter name="p" length="1" scaffold="one two three"/></module></schema>', lm)
    with pytest.raises(ValueError, match="Only <module>"):
        pml.Schema('<schema name="s"><union><x/></union></schema>', lm)
    with pytest.raises(ValueError, match="cannot have text"):
        pml.Prompt("<prompt schema='s'><a>text</a></prompt>")
    with pytest.raises(ValueError, match="cannot be empty"):
        pml.Prompt("<prompt schema='s'/>")
    p = pml.Prompt("<prompt schema='s'>  just text  </prompt>")
    assert p.text == "just text" and p.modules == []


def test_max_tokens_quirk_duplicates_short_sequences():
    """schema.py:167-168 has no length guard: a 10-token run with max_tokens=12 becomes 6+6 tokens."""
    lm = H.TokOnlyLM()
    ts = pml.TokenSequence(0, "a b c d e f g h i j", lm, max_tokens=12)
    ids = lm.encode("a b c d e f g h i j")
    assert ts.token_ids() == ids[:6] + ids[-6:]
    odd = pml.TokenSequence(0, "a b c d e f g h i j", lm, max_tokens=5)
    assert odd.token_ids() == ids[:2] + ids[-3:]      # -5 // 2 == -3
