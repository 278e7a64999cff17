"""Import-path shim: `from utils.initialization import device_initializer, model_initializer`."""
from ccnet_amd.utils.initialization import device_initializer, model_initializer  # noqa: F401
