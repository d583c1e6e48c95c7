"""voxelhashing_demo_amd -- MI355X-native voxel-hashing TSDF fusion path.

The product is libvoxelhash_hip.so (hand-written gfx950 HIP behind the C-ABI in
include/voxelhash.h) plus the host-side mirrors of the reference's
SDF_Hashtable class: C++ (include/SDF_Hashtable.h) and Python (hashtable.py).
"""
from ._lib import (SEM_PINHOLE, SEM_REFERENCE, HashTableParams, VoxelHashError, load)  # noqa: F401
from .hashtable import ENTRY_DTYPE, VOXEL_DTYPE, SDFHashtable, default_params, preprocess  # noqa: F401

RAYCAST_FIXED_STEP, RAYCAST_DDA = 0, 1      # vh_set_option(ctx, "raycast_mode", ...), include/voxelhash.h
BAND_RAY, BAND_NORMAL_DDA, BAND_RAY_DDA = 0, 1, 2      # vh_set_option(ctx, "band_mode", ...)
