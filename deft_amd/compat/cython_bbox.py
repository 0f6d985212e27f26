"""`from cython_bbox import bbox_overlaps` for the reference's utils/matching.py:4 where the `cython_bbox`
package is not installed: put `deft_amd/compat` on PYTHONPATH."""
from deft_amd.association import bbox_overlaps  # noqa: F401
