"""Drop-in for the reference's network/net_old.py (legacy checkpoints, tools/socket2springboot.py:23): the `net.py`
architecture with plain stride-2 BaseConvs where `net.py` has SpaceToDepth, and 3*base_depth blocks in dark3.
`Net(depth=1.0, width=1.0, num_classes=80, act="silu")` with the reference's state_dict keys/shapes; forward runs on
libccnet_hip (see network/graph.py for the dataflow)."""
from ._base import NetBase


class Net(NetBase):
    KIND = "net_old"
