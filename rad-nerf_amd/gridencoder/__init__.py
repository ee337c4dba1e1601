"""Drop-in `gridencoder` package (reference: gridencoder/__init__.py:1) backed by libradnerf_hip.so."""
from .encoder import GridEncoder, grid_encode  # noqa: F401
