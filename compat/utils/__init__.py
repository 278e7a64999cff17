"""Import-path shim for the reference's `utils` package (see compat/README.md)."""
