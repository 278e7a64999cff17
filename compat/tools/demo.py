"""Import-path shim: `tools.demo.inference(model, image, input_shape, conf_thres, nms_thres, device)` (reference tools/demo.py:28).
`tools.demo` IS ccnet_amd.tools.demo (one module object), so the module globals the reference's inference() reads
(`num_classes`, `class_names`, tools/demo.py:72,80) can be set on either name."""
import sys

import ccnet_amd.tools.demo as _impl

sys.modules[__name__] = _impl
