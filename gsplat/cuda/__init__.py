"""gsplat.cuda namespace of the drop-in shim (see gsplat/__init__.py)."""
