#!/usr/bin/env python3
"""Disassembly summary of one kernel of libmarl_hip.so: where its scratch traffic, barriers and memory operations sit.

    python tools/kernel_isa.py <kernel name substring> [--dump] [path/to/lib.so]

Prints, per matching function: instruction counts by class, and the sequence of s_barrier / scratch / global / branch
instructions in program order (a spill inside the stage sequence of a fused integrator shows as scratch_* between barriers).
"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from check_stream_isa import DEFAULT_SO, disassemble_so, functions  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    dump = "--dump" in sys.argv
    pat = args[0]
    so = args[1] if len(args) > 1 else DEFAULT_SO
    fns = functions(disassemble_so(so), pat)
    for name, ins in fns.items():
        c = collections.Counter()
        for _a, op, _o, _t in ins:
            k = ("scratch" if op.startswith("scratch_") else "global" if op.startswith("global_") else "ds" if op.startswith("ds_") else
                 "barrier" if op == "s_barrier" else "branch" if op.startswith(("s_branch", "s_cbranch")) else
                 "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "other")
            c[k] += 1
        print(name)
        print("  ", dict(c), "total", len(ins))
        seq = []
        for a, op, o, t in ins:
            if op == "s_barrier" or op.startswith(("scratch_", "global_", "s_cbranch", "s_branch", "buffer_")):
                seq.append((a, op, o, t))
        if dump:
            for a, op, o, t in seq:
                print("   %6x %-28s %s%s" % (a, op, o[:60], (" -> %x" % t) if t else ""))
        else:
            line = []
            for a, op, o, t in seq:
                line.append({"s_barrier": "|"}.get(op, "S" if op.startswith("scratch_store") else "L" if op.startswith("scratch_load") else
                                                   "g" if op.startswith("global_load") else "w" if op.startswith("global_store") else
                                                   "a" if op.startswith("global_atomic") else "b"))
            print("  ", "".join(line))


if __name__ == "__main__":
    main()
