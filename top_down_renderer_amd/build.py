"""Builds libtdr_hip.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.  No GPU is needed to compile."""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
SRC = [os.path.join(PKG, "csrc", f) for f in
       ("tdr_core.hip", "tdr_map.hip", "tdr_raster.hip", "tdr_score.hip", "tdr_filter.hip", "tdr_prefix.hip", "tdr_geo.hip", "tdr_cmap.hip",
        "tdr_host.cpp", "tdr_gmm.cpp")]
HDR = [os.path.join(ROOT, "include", "tdr.h"), os.path.join(PKG, "csrc", "tdr_common.h"),
       os.path.join(PKG, "csrc", "tdr_sincosf.h"), os.path.join(PKG, "csrc", "tdr_atan2f.h")]
OUT = os.path.join(PKG, "libtdr_hip.so")

# -ffp-contract=off: index arithmetic must round like the reference's non-FMA x86-64 build (see csrc/tdr_common.h)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-I", os.path.join(ROOT, "include")]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built (there is no CPU fallback)")
    return exe


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(f) > t for f in SRC + HDR + [os.path.abspath(__file__)])


def build(force=False, verbose=False, extra_flags=()):
    if not force and not needs_build():
        return OUT
    cmd = [hipcc()] + FLAGS + list(extra_flags) + ["-o", OUT, "-x", "hip"] + SRC
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
