"""Builds libofarn.so (HIP kernels + C-ABI) in-tree with hipcc for gfx950.

    python -m hackathonopticalflow_amd.build [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off keeps every rounding of the arithmetic
contract (no FMA contraction), which is what makes the device results comparable bit for bit
with the CPU oracle.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libofarn.so")
SOURCES = ["kernels_generic.hip", "kernels_fast.hip", "kernels_gauss.hip", "kernels_tile.hip", "kernels_frontend.hip", "kernels_lk.hip", "ofarn_api.hip", "ofarn_api_extras.hip", "ofarn_api_lk.hip", "ofarn_api_stream.hip", "ofarn_api_multi.hip"]
HEADERS = [os.path.join(CSRC, "ofarn_internal.h"), os.path.join(CSRC, "farneback_device.h"), os.path.join(CSRC, "flow_iter_common.h"), os.path.join(CSRC, "ofarn_host.h"), os.path.join(ROOT, "include", "ofarn.h")]
# wrong-result / diagnostic experiment bodies, included only under their -D macro (never by the product build)
HEADERS += sorted(os.path.join(CSRC, "experiments", f) for f in os.listdir(os.path.join(CSRC, "experiments")) if f.endswith(".inc"))
# -fno-slp-vectorize: the SLP vectoriser turns pairs of f32 operations into v_pk_mul_f32 / v_pk_add_f32,
# which measured SLOWER than two scalar VALU ops in these VALU-bound kernels (polyexp 2.42 -> 2.00 ms).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fno-slp-vectorize", "-fvisibility=hidden", "-Wall", "-Wno-unused-result", "-Wno-pass-failed"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, extra_flags=(), out: str | None = None) -> str:
    """Compiles the sources and links libofarn.so.  `extra_flags` / `out` build an experimental
    variant (e.g. -DOFARN_STAMPS=1, the in-kernel s_memtime stamps of the diagnostic build) next to the product library; such a
    variant is selected at run time with the OFARN_LIB environment variable."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    lib = LIB if out is None else os.path.join(HERE, out)
    tag = "" if out is None else "." + os.path.splitext(out)[0]
    # a variant build recompiles only the sources that mention one of its -D macros (all of them if a header does, or if a flag
    # is not a -D); the other objects are the product build's, brought up to date first
    macros = [f[2:].split("=")[0] for f in extra_flags if f.startswith("-D")]
    everywhere = len(macros) != len(list(extra_flags)) or any(m in open(h).read() for m in macros for h in HEADERS)
    if extra_flags and not everywhere:
        build(force=False, verbose=verbose)
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        variant = bool(extra_flags) and (everywhere or any(m in open(s).read() for m in macros))
        o = os.path.join(CSRC, src.rsplit(".", 1)[0] + (tag if variant else "") + ".o")
        objs.append(o)
        if (force and (variant or not extra_flags)) or variant or _stale(o, [s] + HEADERS + [os.path.abspath(__file__)]):
            cmd = [hipcc] + FLAGS + (list(extra_flags) if variant else []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.run(cmd, check=True)
    if force or extra_flags or _stale(lib, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-o", lib] + objs + ["-ldl", "-lpthread"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    return lib


if __name__ == "__main__":
    args = sys.argv[1:]
    out = None
    if "--out" in args:
        i = args.index("--out")
        out = args[i + 1]
        del args[i:i + 2]
    print(build(force="--force" in args, verbose=True, extra_flags=[a for a in args if a.startswith("-") and a != "--force"], out=out))
