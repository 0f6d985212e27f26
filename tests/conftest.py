import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_sessionfinish(session, exitstatus):
    """Leave nothing of the GPU tests for interpreter shutdown: plans, hipGraphs, side streams and pinned buffers are released, and the
    device is idle, while the HIP runtime is still fully alive (graph / stream destructors running during runtime teardown are not a
    thing to depend on)."""
    import gc
    if "torch" in sys.modules:
        import torch
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            torch.cuda.synchronize()
            gc.collect()
            torch.cuda.synchronize()
            torch.cuda.empty_cache()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU emulation test")


@pytest.fixture(scope="session")
def emu_lib():
    """libdeft_emu.so: the unmodified HIP sources compiled against the SIMT emulator
    (tests/hipemu).  Test infrastructure only -- the product never loads it."""
    import subprocess
    from deft_amd import hiplib
    so = os.path.join(ROOT, "tests", "hipemu", "_build", "libdeft_emu.so")
    srcs = [os.path.join(ROOT, "deft_amd", "csrc", f) for f in ("igemm.hip", "igemm3.hip", "dcn.hip", "direct.hip", "ops.hip", "pairmlp.hip", "assoc.hip", "common.h")]
    srcs += [os.path.join(ROOT, "tests", "hipemu", "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "deft_hip.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call([os.path.join(ROOT, "tests", "hipemu", "build_emu.sh")])
    return hiplib.HipLib(so)


@pytest.fixture(scope="session")
def gpu_lib():
    import torch
    from deft_amd import hiplib
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    return hiplib.get_lib()      # raises loudly if libdeft_hip.so is missing


@pytest.fixture(autouse=True)
def _gc_experiment():
    """tools/probe/r6_crash_bisect3.sh: DEFT_TEST_GC=each -> collect + synchronise after every test; =off -> no cyclic collection at all."""
    mode = os.environ.get("DEFT_TEST_GC")
    import gc
    if mode == "off":
        gc.disable()
    yield
    if mode == "each" and "torch" in sys.modules:
        import torch
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            torch.cuda.synchronize()
            gc.collect()
            torch.cuda.synchronize()
