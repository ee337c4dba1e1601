"""Drop-in `shencoder` package (reference: shencoder/__init__.py:1) backed by libradnerf_hip.so."""
from .encoder import SHEncoder, sh_encode  # noqa: F401
