"""Empty shim: reference registration.py:53 imports timm at module level."""
