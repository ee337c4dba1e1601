"""Host-side mirror of the reference's nerf/ package for the render hot path (network + renderer),
plus the synthetic scene the benchmarks and parity tests use."""
from .network import NeRFNetwork, MLP, AudioNet, AudioAttNet  # noqa: F401
from .renderer import NeRFRenderer  # noqa: F401
