"""Import-path shim for the reference's `tools` package (see compat/README.md)."""
