"""`import lap` for the reference's utils/matching.py:1 where the `lap` package is not installed: put
`deft_amd/compat` on PYTHONPATH.  Only `lapjv` (the one function the reference calls, matching.py:48)."""
from deft_amd.association import lapjv  # noqa: F401
