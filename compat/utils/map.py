"""Import-path shim for the mAP arithmetic (reference utils/map.py:112-153, :496-531; plots and pycocotools are out of scope)."""
from ccnet_amd.utils.map import average_precision, detections_to_records, map_50_95, voc_ap  # noqa: F401
