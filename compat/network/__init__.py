"""Import-path shim for the reference's `network` package (see compat/README.md)."""
