#!/usr/bin/env python
"""Instruction-mix statistics of one kernel from a hipcc -S --cuda-device-only listing.

    python benchmarks/isa_stats.py nerf.s _Z19field_kernel_mfma16ILi1ELb0EEv9FieldArgsj7FastDiv [--blocks]

Prints per basic block (label) the counts of VALU / MFMA / SALU / LDS / VMEM instructions, so the inner loops
(K-pass loop, tile loop) can be budgeted instruction by instruction."""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    path, kern = sys.argv[1], sys.argv[2]
    want_blocks = "--blocks" in sys.argv
    want_ops = "--ops" in sys.argv
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(kern + ":"))
    blocks, cur = collections.OrderedDict(), "entry"
    blocks[cur] = []
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end") or l.strip().startswith(".end_amdhsa_kernel"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            continue
        s = l.strip()
        if not s or s.startswith((";", ".", "//")):
            continue
        blocks[cur].append(s.split()[0])
    total = collections.Counter()
    for name, ops in blocks.items():
        c = collections.Counter(classify(o) for o in ops)
        total.update(c)
        if want_blocks and len(ops) >= 20:
            print(f"{name:12s} n={len(ops):5d} " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
            if want_ops:
                oc = collections.Counter(o for o in ops if classify(o) == "valu")
                print("     " + ", ".join(f"{k}:{v}" for k, v in oc.most_common(24)))
    print("TOTAL " + " ".join(f"{k}={v}" for k, v in sorted(total.items())))
    for l in lines[start:]:
        if any(t in l for t in ("; NumVgprs", "; NumAgprs", "; ScratchSize", "; Occupancy", "; LDSByteSize", ".vgpr_count", "; TotalNumVgprs")):
            print(l.strip())
        if l.startswith(".Lfunc_end"):
            pass
        if "; Occupancy" in l:
            break


if __name__ == "__main__":
    main()
