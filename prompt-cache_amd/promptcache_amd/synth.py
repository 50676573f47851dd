"""Synthetic PML workloads with the *structure* of the reference's schemas.

Neither the GPU box nor the benchmark harness can read the reference checkout, and there are no
datasets offline, so benchmarks and full-size tests build their schemas here.  Structures mirrored
(token counts from SURVEY.md section 8a/8d, measured on the reference files with a stand-in tokenizer):

  * ``persona_like``  -- ``examples/persona_generation.xml``: system text, a user preamble, then trait
    modules each holding one ``<union>`` of long members, whitespace between tags (=> 1-token
    segments); the prompt picks one member per trait and adds a short question.  Full size: 28 members
    in 6 traits (29 encode passes), 25 staged segments, S ~ 1725 cached tokens, q = 12.
  * ``flat_docs``     -- union-free schema whose prompt selects every module (``code_generation_game.xml``
    with ``max_tokens=800``: S ~ 4390; ``benchmark/squad_v2.py`` / ``longbench.py``: one context module).

Text is drawn from a fixed list of short words: with the stand-in tokenizer every word is exactly one
token, so segment lengths are controlled to the token.
"""
from __future__ import annotations

import random
from typing import List, Sequence, Tuple

# <= 4 letters: with its leading space a word is one piece (<= 5 characters) of the stand-in tokenizer
_WORDS = ("the sea wind rain city road dusk rock moss lamp hill lake pond wave tide foam reef cove bay "
          "tree bird warm cold calm soft fast slow east west walk read sing grow keep find give take "
          "make know time year day hand eye mind work home land ship farm fog dew ice snow sun moon "
          "star sky sand clay salt wine milk rice bean seed root leaf stem vine fig plum pear lime "
          "kale herb mint sage oak elm ash pine fern reed wool silk rope sail oar keel mast deck").split()


def words(n: int, seed: int) -> str:
    """``n`` space-separated short words (n tokens under the stand-in tokenizer once embedded in PML
    text, where the run's leading whitespace collapses into the first word's leading space)."""
    rnd = random.Random(seed)
    return " ".join(rnd.choice(_WORDS) for _ in range(n))


def persona_like(name: str = "persona", system_len: int = 292, intro_len: int = 84,
                 traits: Sequence[Tuple[str, Sequence[int]]] = (
                     ("age", (174, 169, 181, 160, 177)), ("residence", (256, 240, 249, 262, 251)),
                     ("education", (155, 149, 161, 158, 150)), ("occupation", (267, 259, 270, 255, 263)),
                     ("marital-status", (265, 250, 258, 271)), ("personality", (232, 240, 226, 229))),
                 question_len: int = 8, seed: int = 0, pick: Sequence[int] = ()) -> Tuple[str, str]:
    """Returns ``(schema_pml, prompt_pml)``.  Member text lengths are in tokens; a run of text inside
    ``<module>`` also gets one trailing whitespace token from the closing indentation."""
    s = seed * 1000
    out = [f'<schema name="{name}">', "    <system>", f"        {words(system_len - 8, s + 1)}", "    </system>", "    <user>",
           f"        {words(intro_len - 1, s + 2)}"]
    refs = []
    for ti, (trait, members) in enumerate(traits):
        out += [f'        <module name="{trait}">', "            <union>"]
        for mi, ln in enumerate(members):
            out += [f'                <module name="{trait}-{mi}">', f"                    {words(max(1, ln - 1), s + 10 * ti + mi + 3)}",
                    "                </module>"]
        out += ["            </union>", "        </module>"]
        choice = pick[ti] if ti < len(pick) else (ti + 1) % len(members)
        refs.append(f"    <{trait}><{trait}-{choice}/></{trait}>")
    out += ["    </user>", "</schema>"]
    prompt = "\n".join([f"<prompt schema='{name}'>"] + refs + [f"    <user>{words(question_len, s + 999)}</user>", "</prompt>"])
    return "\n".join(out) + "\n", prompt


def flat_docs(name: str = "docs", system_len: int = 30, module_lens: Sequence[int] = (306, 76, 800, 800, 800, 800, 800),
              question_len: int = 12, seed: int = 0) -> Tuple[str, str]:
    """Union-free schema; the prompt references every module (cached == no-cache up to fp16 KV rounding)."""
    s = seed * 1000 + 500
    out = [f'<schema name="{name}">', f"    <system>{words(system_len, s)}</system>", "    <user>"]
    refs = []
    for i, ln in enumerate(module_lens):
        out.append(f'        <module name="m{i}">{words(ln, s + i + 1)}</module>')
        refs.append(f"<m{i}/>")
    out += ["    </user>", "</schema>"]
    prompt = f"<prompt schema='{name}'>" + "".join(refs) + f"<user>{words(question_len, s + 999)}</user></prompt>"
    return "\n".join(out) + "\n", prompt
