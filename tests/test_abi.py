"""-m "not gpu": the C-ABI library loads and exports every symbol include/deft_hip.h declares
(no compute call is made), the product fails loudly without it, and the product never
reaches into oracle/."""
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "deft_hip.h")).read()
    return sorted(set(re.findall(r"^\s*(?:int|const char\*)\s+(deft_\w+)\s*\(", src, flags=re.M)))


def test_header_matches_binding():
    from deft_amd import hiplib
    assert _declared() == sorted(hiplib.EXPORTS)


def test_gfx950_library_exports_every_symbol():
    from deft_amd import build, hiplib
    so = build.build(force=False, verbose=False)           # hipcc cross-compiles without a GPU
    lib = hiplib.HipLib(so)                                # binds every symbol; raises if one is missing
    for name in _declared():
        assert hasattr(lib.cdll, name)
    assert lib.cdll.deft_version() == 2
    # the code object is gfx950-only (no other offload arch, no host fallback path)
    blob = open(so, "rb").read()
    assert b"gfx950" in blob and b"gfx942" not in blob and b"gfx90a" not in blob


def test_missing_library_fails_loudly():
    from deft_amd import hiplib
    with pytest.raises(hiplib.DeftHipError):
        hiplib.HipLib(os.path.join(ROOT, "deft_amd", "lib", "does_not_exist.so"))


def test_product_never_imports_oracle_or_emulator():
    for f in glob.glob(os.path.join(ROOT, "deft_amd", "**", "*.py"), recursive=True) + [os.path.join(ROOT, "dcn_v2.py")]:
        src = open(f).read()
        assert "deft_oracle" not in src and "hipemu" not in src and "import oracle" not in src, f


def test_gemm_desc_layout_matches_header():
    """ctypes mirror of DeftGemmDesc: 7 pointers then the ints, in header order."""
    from deft_amd.hiplib import GemmDesc
    src = open(os.path.join(ROOT, "include", "deft_hip.h")).read()
    body = src[src.index("typedef struct DeftGemmDesc {"):src.index("} DeftGemmDesc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in re.findall(r"(?:const float\*|float\*|const int\*|int)\s+([^;]+);", body):
        names += [n.strip() for n in decl.split(",")]
    assert names == [n for n, _ in GemmDesc._fields_]
