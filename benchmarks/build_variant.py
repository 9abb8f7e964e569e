#!/usr/bin/env python
"""Build one A/B variant of libunerf with extra -D switches (same flags as lib.build_library otherwise):

    python benchmarks/build_variant.py <name> [-DUNERF_X=1 ...]   ->  benchmarks/build_probe/libunerf_<name>.so

The splat translation unit is compiled once (benchmarks/build_probe/_splat.o, keyed by a digest of its sources) and
linked into every variant; only unerf_nerf.hip is rebuilt.  Used with benchmarks/multi_ab.sh on the GPU box."""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from uncertainty_nerf_gs_amd import lib  # noqa: E402

B = os.path.join(ROOT, "benchmarks", "build_probe")


def main():
    name, defs = sys.argv[1], sys.argv[2:]
    os.makedirs(B, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cflags = [f for f in lib.HIPCC_FLAGS if f != "-shared"] + ["-I", lib.INCLUDE]
    h = hashlib.sha256(" ".join(cflags).encode())
    for f in ("unerf_splat.hip", "unerf_common.hpp"):
        h.update(open(os.path.join(lib.CSRC, f), "rb").read())
    h.update(open(os.path.join(lib.INCLUDE, "unerf.h"), "rb").read())
    splat_o = os.path.join(B, f"_splat_{h.hexdigest()[:12]}.o")
    if not os.path.exists(splat_o):
        subprocess.check_call([hipcc] + cflags + ["-c", "-o", splat_o, os.path.join(lib.CSRC, "unerf_splat.hip")])
    nerf_o = os.path.join(B, f"_nerf_{name}.o")
    subprocess.check_call([hipcc] + cflags + defs + ["-c", "-o", nerf_o, os.path.join(lib.CSRC, "unerf_nerf.hip")])
    out = os.path.join(B, f"libunerf_{name}.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, nerf_o, splat_o])
    os.remove(nerf_o)
    print(out)


if __name__ == "__main__":
    main()
