from . import pose3d_resnet  # noqa: F401  (reference: lib/models/__init__.py:1)
