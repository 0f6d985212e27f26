"""Build libdeft_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m deft_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRCS = [os.path.join(HERE, "csrc", f) for f in ("igemm.hip", "igemm3.hip", "dcn.hip", "direct.hip", "ops.hip")]
DEPS = SRCS + [os.path.join(HERE, "csrc", "common.h"), os.path.join(HERE, "..", "include", "deft_hip.h")]
OUT = os.path.join(HERE, "lib", "libdeft_hip.so")
OBJ_DIR = os.path.join(HERE, "lib", "obj")
# dcn.hip: the SLP vectoriser turns the four-corner blend into packed fp32 math, which wants every corner weight duplicated into a
# register pair (+36 VGPRs: spills at two waves per SIMD)
EXTRA = {"dcn.hip": ["-fno-slp-vectorize"]}


def build(force=False, verbose=True):
    os.makedirs(OBJ_DIR, exist_ok=True)
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in DEPS):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
    objs = []
    for src in SRCS:                      # one object per source: per-file flags (EXTRA), and only the edited file recompiles
        obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
        objs.append(obj)
        deps = [src] + DEPS[len(SRCS):]
        if not force and os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in deps):
            continue
        cmd = base + EXTRA.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    cmd = base + ["-shared", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
