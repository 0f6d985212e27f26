"""Import shim: the reference does `from dcn_v2 import DCN` (src/lib/model/networks/dla.py:25-29,
also necks/dlaup.py:23, necks/msraup.py:20, resdcn.py:20).  With this repository on sys.path
ahead of any CUDA build of CharlesShang/DCNv2, that import resolves to the MI355X kernel."""
from deft_amd.integrate import DCN  # noqa: F401
