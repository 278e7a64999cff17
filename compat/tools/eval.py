"""Import-path shim: `tools.eval.evaluate` / `evaluate_one_img` (reference tools/eval.py:48, :107)."""
from ccnet_amd.tools.eval import evaluate, evaluate_one_img, map_from_dirs  # noqa: F401
