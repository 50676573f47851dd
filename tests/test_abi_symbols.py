"""CPU-side check: the C-ABI library loads and exports every symbol include/*.h declares."""
import ctypes
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = []
    for h in _abi_headers():
        src = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names += re.findall(r"\b(pc_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def _abi_headers():
    """The headers that DECLARE exported symbols."""
    return glob.glob(os.path.join(ROOT, "include", "*.h"))


def test_header_declares_the_hot_path_entry_points():
    d = _declared()
    for must in ("pc_kv_gather", "pc_kv_slice_store", "pc_rope_table", "pc_rope_append", "pc_attn", "pc_gemm",
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
    for h in _abi_headers():
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


def test_struct_layouts_match_the_header(tmp_path):
    """ctypes mirrors of the header's structs have the size and field offsets the C compiler gives them (a field added on one
    side only would shift every later one silently), and the header is valid C99 and C++17."""
    import subprocess
    from promptcache_amd import _native
    src = tmp_path / "layout.c"
    structs = (("pc_attn_args", _native.AttnArgs), ("pc_gemm_args", _native.GemmArgs), ("pc_dense_qkv_args", _native.DenseQkvArgs),
               ("pc_kv_seg", _native.KvSeg), ("pc_kv_row", _native.KvRow), ("pc_gemm_q8_args", _native.GemmQ8Args))
    body = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{ROOT}/include/promptcache_hip.h"', 'int main(void) {']
    for st, cls in structs:
        body.append(f'  printf("{st} %zu\\n", sizeof({st}));')
        for f, _ in cls._fields_:
            body.append(f'  printf("{st}.{f} %zu\\n", offsetof({st}, {f}));')
    body += [f'  printf("PC_KV_ROW_STAGED %u\\n", PC_KV_ROW_STAGED);', '  return 0;', '}']
    src.write_text("\n".join(body))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-Wno-unused-function", str(src), "-o", str(exe)])
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-x", "c++", str(src)])
    got = dict(line.rsplit(" ", 1) for line in subprocess.check_output([str(exe)], text=True).strip().splitlines())
    for st, cls in structs:
        assert int(got[st]) == ctypes.sizeof(cls), st
        for name, _ in cls._fields_:
            assert int(got[f"{st}.{name}"]) == getattr(cls, name).offset, (st, name)
    assert int(got["PC_KV_ROW_STAGED"]) == _native.KV_ROW_STAGED
