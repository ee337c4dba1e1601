"""Drop-in `raymarching` package (reference: raymarching/__init__.py:1) backed by libradnerf_hip.so."""
from .ops import *  # noqa: F401,F403
from .ops import (near_far_from_aabb, sph_from_ray, morton3D, morton3D_invert, packbits, morton3D_dilation,
                  march_rays_train, composite_rays_train, march_rays, composite_rays, compact_rays,
                  padded_samples)
