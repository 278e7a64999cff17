"""Import-path shim: `from utils.util import ...` as the reference's callers write it (tools/demo.py:21)."""
from ccnet_amd.utils.util import (draw_rectangle, get_classes, get_root_path, get_test_lines, get_train_lines,  # noqa: F401
                                  get_val_lines, load_model_weights)
