"""deft_amd -- MI355X (gfx950) native implementation of DEFT's per-frame hot path:
DLA-34/CenterNet forward incl. DCNv2, embedding head, pairwise affinity, LSTM motion
step, decode.  Compute lives in hand-written HIP kernels behind a C ABI
(include/deft_hip.h -> deft_amd/lib/libdeft_hip.so); this package is the Python host
side that mirrors the reference's Detector/Tracker seams.  No CPU fallback."""
__version__ = "0.1.0"
