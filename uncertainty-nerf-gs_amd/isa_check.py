"""Build-time scan of the gfx950 code objects inside libunerf.so (lib.build_library runs it after every fresh compile).

Why: the matrix kernels carry hand-placed instructions the compiler cannot see into -- the split-f16 residuals
(v_fma_mixlo / mixhi_f16 in one inline-assembly statement), the Laplace heads' packed moment update -- and the hazard
recogniser does not look inside inline assembly.  Round 4 found MFMAs ONE instruction behind the v_fma_mixhi_f16 that
completes their B operand (benchmarks/hazard_probe.hip: with no instruction between a VALU write of a VGPR and the MFMA
that reads it, 97 % of the accumulators are wrong; with one wait state or more, none; the compiler keeps two for its own
instructions).  That class of defect is invisible to every parity test that happens to pass, so it is checked where it is
made: in the listing.

check_library(path) disassembles both code objects (llvm-objdump, < 1 s) and, per kernel,
  * FAILS (IsaHazard) if an MFMA reads a VGPR -- as A, B or C operand -- that a VALU instruction wrote fewer than
    MIN_WAIT_STATES wait states earlier in straight-line code (each instruction in between is one wait state, `s_nop n`
    is n + 1; a label or a branch in between ends the window: nothing is assumed across control flow);
  * reports how many packed-fp32 instructions (v_pk_fma / mul / add_f32) the kernels that contain MFMAs hold.  Those are
    legal, and measured (docs/experiments.md 6.3): one behind an MFMA of the SAME wave costs 18 cycles, but replacing them
    by scalar forms changed the field kernels by -1.8 ... +1.2 % -- the other wave of the SIMD covers the wait.  The count
    is kept because the one build that ever returned different values from run to run (the fused-blend experiment,
    experiments 4.5.62 / 4.5.73) was dense with them; it is information, not a gate.
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import tempfile
from typing import Dict, List, Tuple

LLVM_BIN = os.environ.get("UNERF_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
MIN_WAIT_STATES = 2


class IsaHazard(RuntimeError):
    pass


_REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def _regs(operand: str) -> List[int]:
    out: List[int] = []
    for m in _REG.finditer(operand):
        if m.group(1) is not None:
            out.append(int(m.group(1)))
        else:
            out.extend(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def _split_operands(rest: str) -> List[str]:
    ops, depth, cur = [], 0, ""
    for ch in rest:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            ops.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        ops.append(cur.strip())
    return ops


# VALU opcodes whose first operand is NOT a vector destination
_NO_VDST = ("v_cmp", "v_cmpx", "v_readlane", "v_readfirstlane", "v_nop")


def scan_listing(text: str) -> Tuple[List[str], Dict[str, Dict[str, int]]]:
    """-> (hazards, per-kernel {"mfma": n, "pk_f32": n}) of an llvm-objdump -d listing"""
    hazards: List[str] = []
    stats: Dict[str, Dict[str, int]] = {}
    kernel = None
    recent: List[Tuple[int, List[int], str]] = []   # (wait states since, vgprs written, text) of VALU writes still in the window
    for line in text.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            name = m.group(1)
            if not name.startswith("L") and not name.startswith(".L"):   # a function symbol
                kernel = name
                stats.setdefault(kernel, {"mfma": 0, "pk_f32": 0})
            recent = []          # any label: control flow may join here
            continue
        if not line.startswith("\t") or kernel is None:
            continue
        ins = line.split("//")[0].strip()
        if not ins:
            continue
        op, _, rest = ins.partition(" ")
        if op.startswith("s_cbranch") or op in ("s_branch", "s_setpc_b64", "s_swappc_b64", "s_endpgm", "s_barrier"):
            recent = []
            continue
        states = 1
        if op == "s_nop":
            try:
                states = int(rest.strip(), 0) + 1
            except ValueError:
                states = 1
        if op.startswith("v_mfma") or op.startswith("v_smfmac"):
            stats[kernel]["mfma"] += 1
            ops = _split_operands(rest)
            read = [r for o in ops[1:4] for r in _regs(o)]
            for dist, written, txt in recent:
                hit = sorted(set(read) & set(written))
                if hit and dist < MIN_WAIT_STATES:
                    hazards.append(f"{kernel}: `{ins}` reads v{hit[0]} {dist} wait state(s) behind `{txt}`")
        elif re.match(r"v_pk_(fma|mul|add)_f32", op):
            stats[kernel]["pk_f32"] += 1
        # age the window, then add this instruction's writes
        recent = [(d + states, w, t) for d, w, t in recent if d + states < MIN_WAIT_STATES + 1]
        if op.startswith("v_") and not op.startswith(_NO_VDST) and not op.startswith("v_mfma") and not op.startswith("v_smfmac"):
            ops = _split_operands(rest)
            if ops:
                w = _regs(ops[0])
                if w:
                    recent.append((0, w, ins))
    return hazards, stats


def disassemble(path: str) -> List[str]:
    """listings of the gfx950 code objects bundled in a HIP shared library"""
    objdump = os.path.join(LLVM_BIN, "llvm-objdump")
    if not os.path.exists(objdump):
        raise FileNotFoundError(objdump)
    tmp = tempfile.mkdtemp(prefix="unerf_isa_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(path, local)
        subprocess.run([objdump, "--offloading", local], cwd=tmp, capture_output=True, text=True, check=True)
        outs = []
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" in f:
                res = subprocess.run([objdump, "-d", os.path.join(tmp, f)], capture_output=True, text=True, check=True)
                outs.append(res.stdout)
        return outs
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def check_library(path: str, verbose: bool = False) -> Dict[str, Dict[str, int]]:
    hazards: List[str] = []
    stats: Dict[str, Dict[str, int]] = {}
    for text in disassemble(path):
        h, s = scan_listing(text)
        hazards += h
        stats.update(s)
    if verbose:
        mk = {k: v for k, v in stats.items() if v["mfma"]}
        print(f"[isa_check] {len(stats)} kernels, {len(mk)} with MFMAs ({sum(v['mfma'] for v in mk.values())} MFMAs), "
              f"{sum(v['pk_f32'] for v in mk.values())} packed-fp32 instructions in those; {len(hazards)} VALU->MFMA hazards")
    if hazards:
        raise IsaHazard(f"{len(hazards)} MFMA operand(s) read fewer than {MIN_WAIT_STATES} wait states behind the VALU instruction "
                        "that writes them (inline assembly the hazard recogniser cannot see?):\n  " + "\n  ".join(hazards[:20]))
    return stats
