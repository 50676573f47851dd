"""CPU-side check: the C-ABI library loads and exports every symbol include/*.h declares."""
import ctypes
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = []
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names += re.findall(r"\b(pc_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_header_declares_the_hot_path_entry_points():
    d = _declared()
    for must in ("pc_kv_gather", "pc_kv_slice_store", "pc_rope_table", "pc_rope_append", "pc_attn_fwd",
                 "pc_attn_workspace_bytes", "pc_version", "pc_last_error_string"):
        assert must in d


def test_library_loads_and_exports_every_declared_symbol():
    from promptcache_amd import _native
    assert os.path.exists(_native.lib_path()), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_native.lib_path())
    for name in _declared():
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    # and the ctypes signature table covers exactly the header
    assert sorted(_native.SIGNATURES) == _declared()
    assert _native.load().pc_version() == 1


def test_argument_errors_need_no_gpu():
    from promptcache_amd import _native
    lib = _native.load()
    assert lib.pc_kv_gather(None, None, None, 1, None, 1, 1, 128, 16, None) == -1001
    assert b"null pointer" in lib.pc_last_error_string()
    assert lib.pc_attn_workspace_bytes(1, 32, 128, 4390, 4390) == 0        # encode regime: no split
    assert lib.pc_attn_workspace_bytes(1, 32, 128, 12, 1737) > 0           # cached prefill: split-KV partials


def _declared_arity():
    """name -> number of parameters of every ``pc_*`` prototype in include/*.h."""
    out = {}
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        for m in re.finditer(r"\b(pc_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
            params = m.group(2).strip()
            out[m.group(1)] = 0 if params in ("", "void") else params.count(",") + 1
    return out


def test_ctypes_signatures_have_the_arity_of_the_header_prototypes():
    """A parameter added on one side only would shift every later argument silently (ctypes does not check)."""
    from promptcache_amd import _native
    arity = _declared_arity()
    assert sorted(arity) == _declared()
    for name, (_, argtypes) in _native.SIGNATURES.items():
        assert len(argtypes) == arity[name], (name, len(argtypes), arity[name])
