"""Import-path shim: the reference's `from network.net import Net` resolves to the MI355X implementation."""
from ccnet_amd.network.net import Net  # noqa: F401
