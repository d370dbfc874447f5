"""Stub sub-package (see oracle/stubs/pde/__init__.py)."""
