"""Build libdeft_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m deft_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRCS = [os.path.join(HERE, "csrc", f) for f in ("igemm.hip", "igemm3.hip", "dcn.hip", "direct.hip", "ops.hip", "pairmlp.hip", "assoc.hip")]
DEPS = SRCS + [os.path.join(HERE, "csrc", "common.h"), os.path.join(HERE, "..", "include", "deft_hip.h"), os.path.abspath(__file__)]
OUT = os.path.join(HERE, "lib", "libdeft_hip.so")
OBJ_DIR = os.path.join(HERE, "lib", "obj")
# dcn.hip: the SLP vectoriser turns the four-corner blend into packed fp32 math, which wants every corner weight duplicated into a
# register pair (+36 VGPRs: spills at two waves per SIMD)
# assoc.hip (host-only C++: the association cascade, the Kalman filter): no fused multiply-add anywhere in the file -- its results are compared
# operation for operation with the reference's numpy arithmetic (the pragma inside the file covers only the functions below it)
EXTRA = {"dcn.hip": ["-fno-slp-vectorize"], "assoc.hip": ["-ffp-contract=off"]}
# NO packed-fp32 VALU instructions (v_pk_add/mul/fma_f32) in any kernel.  Round 4 found the sampling records of igemm.hip's MODE_DCN wrong in
# lanes 48-63 of a wave -- only while a DIFFERENT kernel (any matrix-core launch of the other sub-batch plan's stream) ran on the same
# compute unit; the same launch alone, or beside a copy kernel, is bit-exact.  The record code's scalar fp32 maths had been SLP-packed into
# v_pk_*_f32; the build without them is bit-exact under every co-runner (tools/probe/concurrency_bisect.py, profiles/r4_pkf32_hazard.md),
# and no slower (config B 933 -> 937 frames/s: MI355X_MICROARCH.md prices a packed fp32 op beside MFMAs above its two scalar halves).
# The feature switch removes the instructions from compiler-generated code altogether, explicit ext_vector arithmetic included.
NO_PK_F32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
_NOISE = "'-packed-fp32-ops' is not a recognized feature for this target"      # (the HOST pass of hipcc sees the switch too and says so)


# The split arithmetic is a compile-time property of a build (csrc/common.h DEFT_PIECES): the product library uses two fp16 pieces per
# operand (three matrix instructions per fp32 product); `libdeft_bf16x3.so` is the same sources with three bf16 pieces (six products, the
# arithmetic of rounds 1-4) -- kept buildable as the cross-check of the two-piece arithmetic (DEFT_HIP_LIB selects it; bench.py reports both).
VARIANTS = {"hip": [], "bf16x3": ["-DDEFT_PIECES=3"]}
# libdeft_hip.so carries BOTH arithmetics (round 6): next to every entry point `name` of the two-fp16-piece build sits `name_p3`, the same source
# compiled with three bf16 pieces (no range limit: the fallback a running process switches to when an activation leaves the fp16 range,
# deft_amd/detector.py).  The second set is the bf16x3 objects with every symbol they DEFINE renamed (llvm-objcopy --redefine-syms; references to
# the runtime and libc are untouched; kernels are registered per object file, so equal device names in two objects do not meet).  assoc.hip is
# host-only C++ without pieces: one copy.
TWIN_SUFFIX = "_p3"
OBJCOPY = "/opt/rocm/lib/llvm/bin/llvm-objcopy"


def _twin_objects(p3_objs, verbose):
    """Copies of the bf16x3 objects (all but assoc.hip) with every extern symbol they define renamed to <name>_p3."""
    twin_dir = os.path.join(HERE, "lib", "obj_twin")
    os.makedirs(twin_dir, exist_ok=True)
    objs = [o for o in p3_objs if not o.endswith("assoc.hip.o")]
    names = set()
    for o in objs:
        for ln in subprocess.check_output(["nm", "--defined-only", "--extern-only", o], text=True).splitlines():
            parts = ln.split()
            if len(parts) >= 3:
                names.add(parts[-1])
    mp = os.path.join(twin_dir, "redefine.map")
    with open(mp, "w") as f:
        for n in sorted(names):
            f.write("%s %s%s\n" % (n, n, TWIN_SUFFIX))
    out = []
    for o in objs:
        t = os.path.join(twin_dir, os.path.basename(o))
        cmd = [OBJCOPY, "--redefine-syms=" + mp, o, t]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        out.append(t)
    return out


def build(force=False, verbose=True, variant="hip", objects_only=False):
    flags = VARIANTS[variant]
    out = os.path.join(HERE, "lib", "libdeft_%s.so" % variant)
    obj_dir = OBJ_DIR if variant == "hip" else os.path.join(HERE, "lib", "obj_" + variant)
    os.makedirs(obj_dir, exist_ok=True)
    if not force and not objects_only and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in DEPS):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
    objs, jobs = [], []
    for src in SRCS:                      # one object per source: per-file flags (EXTRA), and only the edited file recompiles
        obj = os.path.join(obj_dir, os.path.basename(src) + ".o")
        objs.append(obj)
        deps = [src] + DEPS[len(SRCS):]
        if not force and os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in deps):
            continue
        cmd = base + NO_PK_F32 + flags + EXTRA.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        jobs.append((cmd, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))      # (the files compile side by side)
    failed = None
    for cmd, pr in jobs:
        _, stderr = pr.communicate()
        err = "\n".join(ln for ln in stderr.splitlines() if _NOISE not in ln)
        if err.strip():
            sys.stderr.write(err + "\n")
        if pr.returncode != 0 and failed is None:
            failed = subprocess.CalledProcessError(pr.returncode, cmd)
    if failed is not None:
        raise failed
    if objects_only:
        return objs
    if variant == "hip":
        objs = objs + _twin_objects(build(force=force, verbose=verbose, variant="bf16x3", objects_only=True), verbose)
    cmd = base + ["-shared", "-o", out] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    force = "--force" in sys.argv
    names = [a for a in sys.argv[1:] if a in VARIANTS] or (list(VARIANTS) if "--all" in sys.argv else ["hip"])
    for v in names:
        print(build(force=force, variant=v))
