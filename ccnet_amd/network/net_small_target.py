"""Drop-in for the reference's network/net_small_target.py: `Net(depth=1.0, width=1.0, num_classes=80, act="silu")` with the
reference's state_dict keys/shapes; forward runs on libccnet_hip (see network/graph.py for the dataflow)."""
from ._base import NetBase


class Net(NetBase):
    KIND = "net_small_target"
