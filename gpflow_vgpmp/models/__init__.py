"""See gpflow_vgpmp/__init__.py."""
