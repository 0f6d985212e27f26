"""Build libdeft_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m deft_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRCS = [os.path.join(HERE, "csrc", f) for f in ("igemm.hip", "igemm3.hip", "direct.hip", "ops.hip")]
DEPS = SRCS + [os.path.join(HERE, "csrc", "common.h"), os.path.join(HERE, "..", "include", "deft_hip.h")]
OUT = os.path.join(HERE, "lib", "libdeft_hip.so")


def build(force=False, verbose=True):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in DEPS):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", OUT] + SRCS
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
