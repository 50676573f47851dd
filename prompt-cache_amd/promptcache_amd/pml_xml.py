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


# XML 1.0 (5th edition) Name: what libxml2 reads behind an '&' before it looks for the ';'
_NS = ":A-Za-z_\u00C0-\u00D6\u00D8-\u00F6\u00F8-\u02FF\u0370-\u037D\u037F-\u1FFF\u200C\u200D\u2070-\u218F\u2C00-\u2FEF\u3001-\uD7FF" \
      "\uF900-\uFDCF\uFDF0-\uFFFD\U00010000-\U000EFFFF"
_ENTNAME = f"[{_NS}][-.0-9\u00B7\u0300-\u036F\u203F\u2040{_NS}]*"
_REF = re.compile(rf"&(#x[0-9A-Fa-f]+|#[0-9]+|{_ENTNAME});")
_UNTERMINATED = re.compile(rf"&(#x[0-9A-Fa-f]+|#[0-9]+|#x?;?|{_ENTNAME})")


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
            # An unterminated reference.  libxml2's recovering parser has already CONSUMED what it recognised of it when it
            # misses the ';': `&name` (an XML Name: "c&d e" -> "c e", "x &amp y" -> "x  y"), `&#` + decimal digits
            # ("x&#12y" -> "xy"), `&#x` + hex digits ("x&#x4g;" -> "xg;") are dropped whole; an '&' that starts none of
            # these ("p & q", "x&1y", "x&;y") is dropped alone.  (Pinned against libxml2: tests/golden/pml_recover.json.)
            state.broken = True
            mu = _UNTERMINATED.match(s, j)
            i = mu.end() if mu else j + 1
            continue
        body = m.group(1)
        if body.startswith("#"):
            # a character reference outside XML's Char production (&#0; &#12; &#xD800; &#xFFFE; &#x110000; ...) is an error
            # to libxml2: the recovering parser drops it and is "broken" from there on (tests/golden/pml_recover.json)
            cp = int(body[2:], 16) if body.startswith("#x") else int(body[1:])
            if cp in (0x9, 0xA, 0xD) or 0x20 <= cp <= 0xD7FF or 0xE000 <= cp <= 0xFFFD or 0x10000 <= cp <= 0x10FFFF:
                out.append(chr(cp))
            else:
                state.broken = True
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
        if not stack or not text:
            return  # text outside the root element is ignored; a run that recovery reduced to nothing is no text node
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
            # libxml2's recovering parser lets ANY end tag close the innermost open element (a mismatched name is an error it
            # reports and walks over: "<b>x</c>y</b>z" -> <b>x</b>, tail "y", and the second end tag closes the parent); what
            # follows the root element's end is ignored.  (tests/golden/pml_recover.json)
            # An end tag whose name cannot be read ("</#", "</<") closes the innermost element just the same, and the input goes
            # on right behind the "</"; behind a readable name blanks and the '>' are consumed when they are there.
            m = _NAME.match(src, i + 2)
            k = m.end() if m else i + 2
            while k < n and src[k] in " \t\r\n":
                k += 1
            if k < n and src[k] == ">":
                k += 1
            else:
                state.broken = True
            if stack:
                flush()
                closed = stack.pop()
                if m is None or closed.tag != m.group(0):
                    state.broken = True
                last = closed if stack else None
                if not stack:
                    break                                   # the root element is closed: the rest of the input is ignored
            else:
                state.broken = True
            i = k
        elif _NAME_START.match(nxt or " "):
            # Start tag, as xmlParseStartTag recovers: attributes need whitespace in front of them and a quoted value; a name
            # without '= value' is dropped and the tag goes on; anything else where an attribute, '>' or '/>' should stand ends
            # the tag THERE -- the element is created with the attributes read so far, closed at once, and what follows is
            # parsed as content ("<b x='1'y='2'>t" -> <b x="1"/> then the text "y='2'>t"; "<b/ >t" -> <b/> then "/ >t").
            m = _NAME.match(src, i + 1)
            name = m.group(0)
            k = m.end()
            attrib: Dict[str, str] = {}
            opened, selfclose = False, False
            while True:
                k0 = k
                while k < n and src[k] in " \t\r\n":
                    k += 1
                if k < n and src[k] == ">":
                    opened, k = True, k + 1
                    break
                if src.startswith("/>", k):
                    opened, selfclose, k = True, True, k + 2
                    break
                am = _NAME.match(src, k) if k > k0 else None
                if am is None:
                    break                                   # no whitespace / no attribute name: the tag ends here, unfinished
                k = am.end()
                while k < n and src[k] in " \t\r\n":
                    k += 1
                if k >= n or src[k] != "=":
                    state.broken = True                     # attribute without a value: dropped, the tag goes on
                    continue
                k += 1
                while k < n and src[k] in " \t\r\n":
                    k += 1
                if k < n and src[k] in "\"'":
                    q = src.find(src[k], k + 1)
                    lt = src.find("<", k + 1)
                    if lt >= 0 and (q < 0 or lt < q):
                        # '<' inside an attribute value: the value ends in front of it and so does the tag (unfinished)
                        if am.group(0) not in attrib:
                            attrib[am.group(0)] = unescape(src[k + 1:lt], state)
                        k = lt
                        break
                    if q < 0:
                        break
                    if am.group(0) not in attrib:
                        attrib[am.group(0)] = unescape(src[k + 1:q], state)
                    k = q + 1
                else:
                    break                                   # unquoted value: the tag ends behind the '='
            flush()
            node = Node(name, attrib)
            if stack:
                stack[-1].children.append(node)
            elif root is None:
                root = node
            else:
                break  # content after the root element is ignored
            if not opened:
                state.broken = True
                last = node if stack else None              # created and closed at once
                if not stack:
                    break
            elif selfclose:
                last = node if stack else None
                if not stack:
                    break
            else:
                stack.append(node)
                last = None
            i = k
        else:
            state.broken = True
            i += 1  # stray '<' -> dropped (libxml2 recovery)
    flush()
    if root is None:
        raise ValueError("no root element found")
    return root
