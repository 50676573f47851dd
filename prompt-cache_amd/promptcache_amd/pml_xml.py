"""A small recovering XML reader for PML (no lxml in the MI355X image).

The reference parses PML with ``lxml.etree.XMLParser(recover=True)``
(``promptcache/schema.py:285-286``, ``promptcache/prompt.py:158-159``), i.e. libxml2 in recovery mode, and
some shipped schemas rely on it (``examples/code_generation_game.xml`` carries bare ``<`` / ``<=`` inside
Python code).  This reader produces the same element tree (tag / attrib / text / tail / children) for
well-formed input and reproduces libxml2's recovery for the malformed constructs that occur in PML
files (pinned against libxml2 itself by ``tests/golden/pml_recover.json``):

  * a ``<`` that does not open a tag (next char is not a name start, ``/``, ``!`` or ``?``) is dropped;
  * a ``&`` that does not start a known entity / character reference is dropped (an unknown ``&name;``
    is dropped whole);
  * an end tag that matches no open element is ignored; elements still open at their parent's end tag
    (or at EOF) are closed there;
  * once any such error has been seen, libxml2 (2.9.x) stops substituting the predefined entities:
    every later ``&lt; &gt; &amp; &quot; &apos;`` is dropped, while character references (``&#60;``) still
    resolve.  This is observable in the shipped game schema: after the first bare ``<`` in its Python code
    the formatter's escaped ``</s><s> [INST]`` arrives as ``/ss [INST]``.

Comments are kept as nodes whose ``tag`` is the ``Comment`` sentinel (lxml does the same; the schema
loader serialises them as text, ``schema.py:362-363``).
"""
from __future__ import annotations

import re
from typing import Dict, Iterator, List, Optional

_NAME_START = re.compile(r"[A-Za-z_:À-￿]")
_NAME = re.compile(r"[A-Za-z_:À-￿][-A-Za-z0-9_:.·À-￿]*")
_ATTR = re.compile(r"\s*([A-Za-z_:À-￿][-A-Za-z0-9_:.·À-￿]*)\s*=\s*(\"([^\"]*)\"|'([^']*)')")
_ENTITIES = {"lt": "<", "gt": ">", "amp": "&", "quot": '"', "apos": "'"}


def Comment():  # sentinel, compared by identity (node.tag is Comment)
    raise TypeError("Comment is a sentinel")


class Node:
    __slots__ = ("tag", "attrib", "text", "tail", "children")

    def __init__(self, tag, attrib: Optional[Dict[str, str]] = None):
        self.tag = tag
        self.attrib: Dict[str, str] = attrib or {}
        self.text: Optional[str] = None
        self.tail: Optional[str] = None
        self.children: List["Node"] = []

    def __iter__(self) -> Iterator["Node"]:
        return iter(self.children)

    def __len__(self) -> int:
        return len(self.children)

    def __repr__(self) -> str:
        return f"<Node {self.tag!r} attrib={self.attrib!r} children={len(self.children)}>"


_REF = re.compile(r"&(#x[0-9A-Fa-f]+|#[0-9]+|[A-Za-z_][-A-Za-z0-9_.]*);")


class _State:
    """Parser-wide recovery state: ``broken`` flips at the first well-formedness error."""
    __slots__ = ("broken",)

    def __init__(self):
        self.broken = False


def unescape(s: str, state: Optional[_State] = None) -> str:
    """Resolve entity / character references; drop what libxml2's recovery drops."""
    if "&" not in s:
        return s
    state = state or _State()
    out = []
    i, n = 0, len(s)
    while i < n:
        j = s.find("&", i)
        if j < 0:
            out.append(s[i:])
            break
        out.append(s[i:j])
        m = _REF.match(s, j)
        if not m:
            state.broken = True   # bare '&' -> dropped
            i = j + 1
            continue
        body = m.group(1)
        if body.startswith("#x"):
            out.append(chr(int(body[2:], 16)))
        elif body.startswith("#"):
            out.append(chr(int(body[1:])))
        elif body in _ENTITIES:
            if not state.broken:
                out.append(_ENTITIES[body])
        else:
            state.broken = True   # unknown named entity -> dropped whole
        i = m.end()
    return "".join(out)


def escape_text(s: str) -> str:
    return s.replace("&", "&amp;").replace("<", "&lt;").replace(">", "&gt;")


def tostring(node: Node, with_tail: bool = True) -> str:
    """Serialise like ``lxml.etree.tostring`` (tail included), as text."""
    if node.tag is Comment:
        s = f"<!--{node.text or ''}-->"
    else:
        attrs = "".join(f' {k}="{escape_text(v).replace(chr(34), "&quot;")}"' for k, v in node.attrib.items())
        if not node.children and not node.text:
            s = f"<{node.tag}{attrs}/>"
        else:
            inner = escape_text(node.text or "") + "".join(tostring(c) for c in node.children)
            s = f"<{node.tag}{attrs}>{inner}</{node.tag}>"
    if with_tail and node.tail:
        s += escape_text(node.tail)
    return s


def fromstring(src: str) -> Node:
    """Parse one document and return its root element."""
    root: Optional[Node] = None
    stack: List[Node] = []
    last: Optional[Node] = None  # last closed child of stack[-1] (its tail receives text)
    buf: List[str] = []
    state = _State()

    def flush():
        nonlocal buf
        if not buf:
            return
        text = "".join(buf)
        buf = []
        if not stack:
            return  # text outside the root element is ignored
        if last is not None:
            last.tail = (last.tail or "") + text
        else:
            stack[-1].text = (stack[-1].text or "") + text

    i, n = 0, len(src)
    while i < n:
        c = src[i]
        if c != "<":
            j = src.find("<", i)
            j = n if j < 0 else j
            buf.append(unescape(src[i:j], state))
            i = j
            continue
        nxt = src[i + 1] if i + 1 < n else ""
        if src.startswith("<!--", i):
            j = src.find("-->", i + 4)
            j = n - 3 if j < 0 else j
            flush()
            if stack:
                cm = Node(Comment)
                cm.text = src[i + 4:j]
                stack[-1].children.append(cm)
                last = cm
            i = j + 3
        elif src.startswith("<![CDATA[", i):
            j = src.find("]]>", i + 9)
            j = n if j < 0 else j
            buf.append(src[i + 9:j])
            i = j + 3
        elif nxt == "?" or nxt == "!":
            j = src.find(">", i)
            i = n if j < 0 else j + 1
        elif nxt == "/":
            m = _NAME.match(src, i + 2)
            j = src.find(">", i)
            j = n - 1 if j < 0 else j
            if m:
                name = m.group(0)
                depth = next((d for d in range(len(stack) - 1, -1, -1) if stack[d].tag == name), None)
                if depth is not None:
                    flush()
                    closed = stack[depth]
                    if depth != len(stack) - 1:
                        state.broken = True  # elements left open inside: closed here (mismatched nesting)
                    del stack[depth:]
                    last = closed if stack else None
                else:
                    state.broken = True  # stray end tag -> ignored
            i = j + 1
        elif _NAME_START.match(nxt or " "):
            m = _NAME.match(src, i + 1)
            name = m.group(0)
            k = m.end()
            attrib: Dict[str, str] = {}
            while True:
                am = _ATTR.match(src, k)
                if not am:
                    break
                attrib[am.group(1)] = unescape(am.group(3) if am.group(3) is not None else am.group(4), state)
                k = am.end()
            j = src.find(">", k)
            j = n - 1 if j < 0 else j
            selfclose = src[j - 1] == "/" if j > k - 1 else False
            flush()
            node = Node(name, attrib)
            if stack:
                stack[-1].children.append(node)
            elif root is None:
                root = node
            else:
                i = j + 1
                continue  # content after the root element is ignored
            if selfclose:
                last = node if stack else None
            else:
                stack.append(node)
                last = None
            i = j + 1
        else:
            state.broken = True
            i += 1  # stray '<' -> dropped (libxml2 recovery)
    flush()
    if root is None:
        raise ValueError("no root element found")
    return root
