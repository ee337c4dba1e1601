"""Drop-in `freqencoder` package (reference: freqencoder/__init__.py:1) backed by libradnerf_hip.so."""
from .encoder import FreqEncoder, freq_encode  # noqa: F401
