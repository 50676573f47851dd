"""Compressed opcode trace of one kernel out of a hipcc -S listing: one letter per instruction, one line per basic block
(M mfma, v VALU, X v_exp, L ds_read, l other ds, W s_waitcnt, B s_barrier, G global_load_lds, g other VMEM, n s_nop, s SALU,
J branch, P s_setprio).  usage: isa_trace.py <file.s> <substring of the mangled kernel name> [min mfma per block to print]"""
import re
import sys


def trace(path, key, min_m=16):
    lines = open(path).read().split("\n")
    start = [i for i, l in enumerate(lines) if key in l and l.split(":")[0].startswith("_Z") and ":" in l and not l.startswith("\t")][0]
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    out = []
    for l in lines[start + 1:end]:
        t = l.strip()
        if re.match(r"^\.LBB", t):
            out.append("\n" + t.split(":")[0] + ": ")
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        op = t.split()[0]
        c = ("M" if op.startswith("v_mfma") else "X" if op.startswith("v_exp") else "L" if op.startswith("ds_read") else
             "l" if op.startswith("ds_") else "W" if op.startswith("s_waitcnt") else "B" if op.startswith("s_barrier") else
             "G" if op.startswith("global_load_lds") else "g" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else
             "n" if op.startswith("s_nop") else "P" if op.startswith("s_setprio") else
             "J" if op.startswith(("s_cbranch", "s_branch")) else "s" if op.startswith("s_") else "v" if op.startswith("v_") else "?")
        out.append(c)
    return "".join(out)


if __name__ == "__main__":
    s = trace(sys.argv[1], sys.argv[2])
    min_m = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    for blk in s.split("\n"):
        if blk.count("M") >= min_m:
            print(blk)
            print("   ", {c: blk.count(c) for c in "MvXLlWBGgnsP" if blk.count(c)})
