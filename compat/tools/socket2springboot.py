"""Import-path shim: the service surface of the reference's tools/socket2springboot.py (:33 inference, :126 main, :158 ServerThreading)."""
from ccnet_amd.tools.socket2springboot import ServerThreading, handle_request, inference, main  # noqa: F401

if __name__ == "__main__":
    main()
