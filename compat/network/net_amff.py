"""Import-path shim: the reference's `from network.net_amff import Net` resolves to the MI355X implementation."""
from ccnet_amd.network.net_amff import Net  # noqa: F401
