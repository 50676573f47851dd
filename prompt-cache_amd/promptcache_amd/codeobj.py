"""Kernel resource table of the built HIP library, read from the code objects embedded in the shared object (no GPU, no hipcc).

``kernels(path)`` -> list of dicts (name, demangled, vgpr, agpr, sgpr, scratch bytes per lane, lds bytes, max threads), one per gfx950 kernel:
the AMDGPU metadata note (msgpack, ``amdhsa.kernels``) of every device ELF inside the ``__CLANG_OFFLOAD_BUNDLE__`` images hipcc
links into ``.hip_fatbin``.  Used by tests/test_kernel_resources.py (no reachable kernel may spill) and tools/kernel_table.py."""
from __future__ import annotations

import struct
import subprocess
from typing import Dict, List

_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _device_images(blob: bytes, arch: str = "gfx950") -> List[bytes]:
    out, pos = [], 0
    while True:
        pos = blob.find(_MAGIC, pos)
        if pos < 0:
            return out
        n = struct.unpack_from("<Q", blob, pos + len(_MAGIC))[0]
        p = pos + len(_MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if arch in triple and size:
                out.append(blob[pos + off:pos + off + size])
        pos += len(_MAGIC)


def _notes(elf: bytes):
    assert elf[:4] == b"\x7fELF" and elf[4] == 2, "not a 64-bit ELF"
    shoff = struct.unpack_from("<Q", elf, 0x28)[0]
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for i in range(shnum):
        sh = elf[shoff + i * shentsize: shoff + (i + 1) * shentsize]
        sh_type = struct.unpack_from("<I", sh, 4)[0]
        off, size = struct.unpack_from("<QQ", sh, 0x18)
        if sh_type != 7:          # SHT_NOTE
            continue
        p, end = off, off + size
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            name = elf[p:p + namesz].rstrip(b"\0")
            p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            yield name, ntype, desc


def kernels(path: str) -> List[Dict]:
    import msgpack
    blob = open(path, "rb").read()
    rows = []
    for img in _device_images(blob):
        for name, ntype, desc in _notes(img):
            if name == b"AMDGPU" and ntype == 32:
                meta = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                for k in meta.get("amdhsa.kernels", []):
                    rows.append(dict(name=k[".name"], vgpr=k.get(".vgpr_count", 0), agpr=k.get(".agpr_count", 0),
                                     sgpr=k.get(".sgpr_count", 0), scratch=k.get(".private_segment_fixed_size", 0),
                                     lds=k.get(".group_segment_fixed_size", 0), max_threads=k.get(".max_flat_workgroup_size", 0)))
    names = "\n".join(r["name"] for r in rows)
    try:
        dem = subprocess.run(["c++filt"], input=names, capture_output=True, text=True, check=True).stdout.split("\n")
    except Exception:
        dem = [r["name"] for r in rows]
    for r, d in zip(rows, dem):
        r["demangled"] = d.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    return rows
