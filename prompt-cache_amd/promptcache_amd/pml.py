"""Prompt Markup Language front-end: schema layout (position-id assignment) and prompt parsing.

Integer/CPU work that feeds the hot path.  Mirrors the public surface of the reference's
``promptcache/schema.py`` (``Path`` :32-73, ``Parameter`` :100-153, ``TokenSequence`` :156-182,
``UnionModule`` :185-259, ``Module`` :262-435, ``Scaffold`` :439-523, ``Schema`` :527-533) and
``promptcache/prompt.py`` (``read_file`` :18-27, ``compact_surrounding_spaces`` :46-47, ``CompactSpaces``
:78-93, ``ModuleRef`` :96-131, ``Prompt`` :140-210) so PML files load unchanged, on top of the
lxml-free reader in ``pml_xml``.  Layout rules:

  * a module lays its children out consecutively from its own offset; every text run is
    surrounding-space-compacted and becomes a ``TokenSequence`` when non-empty (so the whitespace
    between two tags is a 1-token segment);
  * all members of a ``<union>`` share one offset and the union is as long as its longest member;
  * a ``<parameter length=n>`` reserves n positions (scaffold text, then ``unk`` padding);
  * ``max_tokens`` keeps ``ids[:max_tokens//2] + ids[-max_tokens//2:]`` with NO length guard
    (schema.py:167-168) -- shorter sequences are duplicated; reproduced because it changes token
    streams and therefore every downstream tensor.
"""
from __future__ import annotations

import os
import re
from typing import Callable, Iterable, List, Optional, Sequence, Union

from . import pml_xml

_NAME_OK = re.compile(r"^[a-zA-Z_][a-zA-Z0-9_\-.]*$")


# ------------------------------------------------------------------------------------------------
# text preprocessors (prompt.py:18-93)
# ------------------------------------------------------------------------------------------------

def compact_surrounding_spaces(text: str) -> str:
    """Leading and trailing whitespace runs each collapse to one space (prompt.py:46-47)."""
    return re.sub(r"^\s+|\s+$", " ", text)


def compact_spaces(text: str) -> str:
    return " ".join(text.split())


def escape_xml(data: str) -> str:
    """prompt.py:38-42 (``xml.sax.saxutils.escape`` with quote entities)."""
    data = data.replace("&", "&amp;").replace(">", "&gt;").replace("<", "&lt;")
    return data.replace("'", "&apos;").replace('"', "&quot;")


class Preprocessor:
    def __call__(self, prompt: str) -> str:  # pragma: no cover - interface
        raise NotImplementedError


class PreprocessorList(Preprocessor):
    def __init__(self, pre: Sequence[Callable[[str], str]]):
        self.pre = list(pre)

    def __call__(self, prompt: str) -> str:
        for p in self.pre:
            prompt = p(prompt)
        return prompt


class CompactSpaces(Preprocessor):
    def __init__(self, only_surrounding: bool = False):
        self.only_surrounding = only_surrounding

    def __call__(self, prompt: str) -> str:
        return compact_surrounding_spaces(prompt) if self.only_surrounding else compact_spaces(prompt)


def read_file(filename: str, preprocessors: Optional[Iterable[Callable[[str], str]]] = None) -> str:
    with open(filename, "r") as f:
        text = f.read()
    for p in preprocessors or ():
        text = p(text)
    return text


# ------------------------------------------------------------------------------------------------
# paths
# ------------------------------------------------------------------------------------------------

class Path:
    def __init__(self, path: Union[None, str, Sequence[str]] = None):
        if path is None:
            parts: List[str] = []
        elif isinstance(path, str):
            parts = [s.strip() for s in path.split("/")] if "/" in path else ([path] if path else [])
        else:
            parts = list(path)
        self.path = parts

    def __len__(self):
        return len(self.path)

    def __str__(self):
        return "/".join(self.path)

    __repr__ = __str__

    @property
    def is_root(self) -> bool:
        return not self.path

    is_empty = is_root

    @property
    def head(self) -> Optional[str]:
        return self.path[0] if self.path else None

    @property
    def next(self) -> "Path":
        return Path(self.path[1:])


# ------------------------------------------------------------------------------------------------
# layout elements
# ------------------------------------------------------------------------------------------------

class Element:
    name: Optional[str] = None
    offset: int = 0

    def __len__(self) -> int:  # pragma: no cover - interface
        raise NotImplementedError

    def span(self) -> str:
        return f"[{self.offset}:{self.offset + len(self)}]"


class TokenSequence(Element):
    """A cached text segment: ``len`` consecutive position ids starting at ``offset``."""

    def __init__(self, offset: int, text: str, lm, max_tokens: Optional[int] = None):
        self.offset = offset
        self.text = text
        ids = list(lm.encode(text))
        if max_tokens is not None:
            ids = ids[:max_tokens // 2] + ids[-max_tokens // 2:]   # schema.py:167-168, no length guard
        self._ids = ids
        self._pos = list(range(offset, offset + len(ids)))

    def __len__(self):
        return len(self._ids)

    def token_ids(self) -> List[int]:
        return self._ids

    def position_ids(self) -> List[int]:
        return self._pos

    def __repr__(self):
        return f"{self.span()} Text: {self.text!r}"


class Parameter(Element):
    def __init__(self, offset: int, spec: pml_xml.Node, lm):
        self.offset = offset
        a = spec.attrib
        if "name" not in a:
            raise ValueError("Parameter name is missing")
        if "length" not in a:
            raise ValueError("Parameter length (in tokens) is missing")
        if not _NAME_OK.fullmatch(a["name"]):
            raise ValueError(f'Parameter name {a["name"]} is not valid')
        self.name = a["name"]
        self.length = int(a["length"])
        self.placeholder_token = lm.unk_token_id
        ids: List[int] = []
        if "scaffold" in a:
            ids = list(lm.encode(a["scaffold"]))
            if len(ids) > self.length:
                raise ValueError(f"Scaffold for parameter {self.name} is too long")
        self._ids = ids + [self.placeholder_token] * (self.length - len(ids))
        self._pos = list(range(offset, offset + self.length))

    def __len__(self):
        return self.length

    def token_ids(self) -> List[int]:
        return self._ids

    def position_ids(self) -> List[int]:
        return self._pos

    def __repr__(self):
        return f"{self.span()} Parameter @{self.name}"


class UnionModule(Element):
    def __init__(self, offset: int, spec: pml_xml.Node, lm, max_tokens: Optional[int] = None):
        self.offset = offset
        self.modules: List[Module] = []
        self.scaffold_name: Optional[str] = None
        for e in spec:
            if e.tag != "module":
                raise ValueError("Only <module> tags are allowed in union")
            self.modules.append(Module(offset, e, lm, max_tokens=max_tokens))  # every member at the same offset
        self.length = max((len(m) for m in self.modules), default=0)
        if "scaffold" in spec.attrib:
            if self.select(spec.attrib["scaffold"]) is None:
                raise ValueError(f'Union scaffold {spec.attrib["scaffold"]} is not found in union')
            self.scaffold_name = spec.attrib["scaffold"]

    def __len__(self):
        return self.length

    def token_ids(self):
        raise ValueError("Cannot get token_ids() on union. Try again on its scaffold")

    def position_ids(self):
        raise ValueError("Cannot get position_ids() on union. Try again on its scaffold")

    def select(self, path: Union[None, str, Path]) -> Optional["Module"]:
        if path is None:
            return None
        path = Path(path) if isinstance(path, str) else path
        if path.is_root:
            raise ValueError("Cannot select root of union")
        for m in self.modules:
            if m.name == path.head:
                return m if len(path) == 1 else m.select(path.next)
        return None

    def __repr__(self):
        return f"{self.span()} Union(" + ", ".join(m.name for m in self.modules) + ")"


class Module(Element):
    def __init__(self, offset: int, spec: Union[str, pml_xml.Node], lm, is_root: bool = False,
                 max_tokens: Optional[int] = None):
        self.offset = offset
        self.children: List[Element] = []
        self.cache = True
        self._is_root = is_root
        self._contains_union = False
        if isinstance(spec, str):
            spec = pml_xml.fromstring(spec)
        self._build(spec, lm, max_tokens)

    # -- construction ---------------------------------------------------------------------------
    def _text(self, cursor: int, raw: Optional[str], lm, max_tokens) -> int:
        if raw is None:
            return cursor
        text = compact_surrounding_spaces(raw)
        if not text:
            return cursor
        seq = TokenSequence(cursor, text, lm, max_tokens=max_tokens)
        self.children.append(seq)
        return cursor + len(seq)

    def _build(self, root: pml_xml.Node, lm, max_tokens):
        want = "schema" if self._is_root else "module"
        if root.tag != want:
            raise ValueError(f"expected <{want}> but found <{root.tag}>")
        if "name" not in root.attrib:
            raise ValueError("Module name is missing")
        if not _NAME_OK.fullmatch(root.attrib["name"]):
            raise ValueError(f'Module name {root.attrib["name"]} is not valid')
        if not self._is_root and "cache" in root.attrib:
            self.cache = root.attrib["cache"] == "true"
        self.name = root.attrib["name"]

        cursor = self.offset
        if "src" in root.attrib:
            src = root.attrib["src"]
            if not os.path.exists(src):
                raise ValueError(f"Module source file {src} does not exist")
            with open(src) as f:
                cursor = self._text(cursor, f.read(), lm, max_tokens)
        cursor = self._text(cursor, root.text, lm, max_tokens)

        for e in root:
            if e.tag == "module":
                child: Element = Module(cursor, e, lm, max_tokens=max_tokens)
                self._contains_union = self._contains_union or child._contains_union
                if child.name in [c.name for c in self.modules()]:
                    raise ValueError(f"Module {child.name} is already defined")
            elif e.tag == "union":
                child = UnionModule(cursor, e, lm, max_tokens=max_tokens)
                self._contains_union = True
                taken = [c.name for c in self.modules()]
                for c in child.modules:
                    if c.name in taken:
                        raise ValueError(f"Module {c.name} is already defined")
            elif e.tag == "parameter":
                if self._is_root:
                    raise ValueError("Parameters are not allowed in schema")
                child = Parameter(cursor, e, lm)
                if child.name in [c.name for c in self.parameters()]:
                    raise ValueError(f"Parameter {child.name} is already defined")
            else:
                # any other tag (and XML comments: child nodes whose tag is not a string) is serialised and handed to the
                # tokenizer (schema.py:362-363).  The reference serialises with lxml.etree.tostring, which returns BYTES (ASCII with
                # character references, tail included), so what `lm.encode` receives is bytes here too: the HF tokenizers refuse
                # them with a ValueError, i.e. the reference cannot load a schema with a comment or foreign tag among a module's
                # children -- and neither does this loader (pinned by tests/golden/pml_layout.json: syn:comment_module,
                # syn:unknown_tag and the two benchmark/schema/test files that carry comments).
                # Raised HERE, with the text the reference's (slow, transformers 4.34) tokenizer gives, so that the outcome does not
                # depend on the tokenizer behind `lm.encode` (HF fast tokenizers raise TypeError for bytes, a custom LM might
                # silently tokenize them); the cause names what is wrong with the schema.
                data = pml_xml.tostring(e).encode("ascii", "xmlcharrefreplace")
                raise ValueError(f"Input {data!r} is not valid. Should be a string, a list/tuple of strings or a list/tuple of "
                                 f"integers.") from ValueError(
                    "XML comment or unknown tag among a module's children: the reference hands its serialised BYTES to the "
                    "tokenizer (schema.py:362-363) and cannot load such a schema; remove it or make it a <module>")
            self.children.append(child)
            cursor += len(child)
            cursor = self._text(cursor, e.tail, lm, max_tokens)
        self.length = cursor - self.offset

    # -- queries --------------------------------------------------------------------------------
    def __len__(self):
        return self.length

    def contains_union(self) -> bool:
        return self._contains_union

    def modules(self) -> List["Module"]:
        out: List[Module] = []
        for c in self.children:
            if type(c) is Module:
                out.append(c)
            elif type(c) is UnionModule:
                out.extend(c.modules)
        return out

    def parameters(self) -> List[Parameter]:
        return [c for c in self.children if type(c) is Parameter]

    def token_sequences(self) -> List[TokenSequence]:
        return [c for c in self.children if type(c) is TokenSequence]

    def token_ids(self) -> List[int]:
        if self._contains_union:
            raise ValueError("Cannot get token_ids() on module that contains union. Try again on its scaffold")
        return [t for c in self.children for t in c.token_ids()]

    def position_ids(self) -> List[int]:
        if self._contains_union:
            raise ValueError("Cannot get position_ids() on module that contains union. Try again on its scaffold")
        return [t for c in self.children for t in c.position_ids()]

    def select(self, path: Union[str, Path]) -> Optional["Module"]:
        if isinstance(path, str):
            # (request assembly selects by plain module name once per referenced module, on the TTFT path: remembered)
            memo = self.__dict__.setdefault("_select_memo", {})
            if path not in memo:
                memo[path] = self.select(Path(path))
            return memo[path]
        if path.is_root:
            return self
        for m in self.modules():
            if m.name == path.head:
                return m if len(path) == 1 else m.select(path.next)
        return None

    def get_scaffold(self, *paths: Path) -> "Scaffold":
        return Scaffold(self, *paths)

    def __repr__(self):
        kind = "Schema" if self._is_root else "Module"
        lines = [f"{self.span()} {kind} @{self.name}"]
        for c in self.children:
            lines += ["\t" + s for s in repr(c).split("\n")]
        return "\n".join(lines)


class Scaffold(Element):
    """A union-free projection of a module: each union is replaced by the member on a requested path,
    else by its declared default member, else dropped (schema.py:449-481)."""

    def __init__(self, module: Module, *paths: Path):
        self.module = module
        self.name = module.name
        self.offset = module.offset
        self.children: List[Element] = []
        for e in module.children:
            if type(e) is UnionModule:
                rel = [p for p in paths if e.select(p.head)]
                names = list({p.head for p in rel})
                if len(names) > 1:
                    raise ValueError("Union cannot have multiple names in scaffold")
                pick = names[0] if rel else e.scaffold_name
                if pick is None:
                    continue
                self.children.append(Scaffold(e.select(pick), *[p.next for p in rel]))
            elif type(e) is Module:
                self.children.append(Scaffold(e, *[p.next for p in paths if p.head == e.name]))
            else:
                self.children.append(e)

    def __len__(self):
        return self.module.length

    def token_ids(self) -> List[int]:
        return [t for c in self.children for t in c.token_ids()]

    def position_ids(self) -> List[int]:
        return [t for c in self.children for t in c.position_ids()]

    def select(self, path: Union[str, Path]) -> Optional["Scaffold"]:
        path = Path(path) if isinstance(path, str) else path
        if path.is_root:
            return self
        for c in self.children:
            if type(c) is Scaffold and c.name == path.head:
                return c if len(path) == 1 else c.select(path.next)
        return None

    def all_token_sequences(self) -> List[TokenSequence]:
        out: List[TokenSequence] = []
        for c in self.children:
            if type(c) is Scaffold:
                out += c.all_token_sequences()
            elif type(c) is TokenSequence:
                out.append(c)
        return out


class Schema(Module):
    def __init__(self, spec: Union[str, pml_xml.Node], lm, max_tokens: Optional[int] = None):
        super().__init__(0, spec, lm, is_root=True, max_tokens=max_tokens)
        self.lm = lm

    def encode_paths(self) -> List[Path]:
        """The scaffolds the cache engine encodes: the root scaffold plus one per union member that
        is not the default all the way up (cache_engine.py:188-210; LIFO traversal order kept because a
        segment reachable from several scaffolds keeps the KV of the LAST one encoded)."""
        paths = [Path()]
        stack = [([], True, self)] if self.contains_union() else []
        while stack:
            prefix, parent_default, u = stack.pop()
            here = prefix + [u.name]
            for e in u.children:
                if type(e) is Module and e.contains_union():
                    stack.append((here, parent_default, e))
                elif type(e) is UnionModule:
                    for m in e.modules:
                        is_default = (e.scaffold_name == m.name) and parent_default
                        if m.contains_union():
                            stack.append((here, is_default, m))
                        if not is_default:
                            paths.append(Path(here + [m.name]).next)
        return paths


# ------------------------------------------------------------------------------------------------
# prompts (prompt.py:96-210)
# ------------------------------------------------------------------------------------------------

class Argument:
    def __init__(self, name: str, value: str):
        self.name, self.value = name, value

    def __repr__(self):
        return f"{self.name}={self.value!r}"


class ModuleRef:
    def __init__(self, spec: Optional[pml_xml.Node] = None):
        self.name: str = ""
        self.args: List[Argument] = []
        self.modules: List[ModuleRef] = []
        if spec is not None:
            self.name = spec.tag
            self.args = [Argument(k, v) for k, v in spec.attrib.items()]
            if spec.text is not None and spec.text.strip():
                raise ValueError("Module reference cannot have text")
            for e in spec:
                self.modules.append(ModuleRef(e))
                if e.tail is not None and e.tail.strip():
                    raise ValueError("Module reference cannot have text")

    def __repr__(self):
        head = f"@{self.name}" + (f"({' '.join(map(repr, self.args))})" if self.args else "")
        return "\n".join([head] + ["\t" + s for m in self.modules for s in repr(m).split("\n")])


class Prompt(ModuleRef):
    def __init__(self, spec: Union[str, pml_xml.Node], preproc: Optional[Sequence[Callable[[str], str]]] = None):
        super().__init__()
        self.preproc = list(preproc) if preproc is not None else []
        self.text = ""
        if isinstance(spec, str):
            for p in self.preproc:
                spec = p(spec)
            spec = pml_xml.fromstring(spec)
        if spec.tag != "prompt":
            raise ValueError(f"expected <prompt> but found <{spec.tag}>")
        self.schema = spec.attrib.get("schema", "")
        self.name = self.schema
        if len(spec) == 0:                                  # text-only prompt
            if spec.text is None:
                raise ValueError("Prompt cannot be empty")
            self.text = compact_surrounding_spaces(spec.text)
        else:
            if spec.text is not None and spec.text.strip():
                raise ValueError("Prompt cannot have leading text")
            for e in spec:
                self.modules.append(ModuleRef(e))
                if e.tail is not None:                       # the LAST child's tail wins (prompt.py:194-195)
                    self.text = compact_surrounding_spaces(e.tail)
        self.text = self.text.strip()

    def add_text(self, text: str):
        for p in self.preproc:
            text = p(text)
        self.text += text

    def __repr__(self):
        lines = [f"Schema: @{self.name}"] + ["\t" + s for m in self.modules for s in repr(m).split("\n")]
        return "\n".join(lines + [f"Text: {self.text!r}"])
