"""Import-path shim: `from utils.bbox import decode_outputs, post_process` (reference utils/bbox.py:13, :82)."""
from ccnet_amd.utils.bbox import decode_outputs, post_process  # noqa: F401
