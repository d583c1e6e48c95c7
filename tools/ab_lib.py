#!/usr/bin/env python3
"""tools/ab_kernels.py against whatever library VOXELHASH_LIB names, including one built from an older
commit: entry points that library does not export are dropped from the binding table first.
Used by tools/ab_commits.sh; same arguments as ab_kernels.py."""
import ctypes as C
import os
import sys

import torch  # noqa: F401  (first: one HIP runtime per process, see _lib.load)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from voxelhashing_demo_amd import _lib  # noqa: E402

L = C.CDLL(_lib.LIB_PATH)
for name in list(_lib.SIGNATURES):
    if not hasattr(L, name):
        _lib.SIGNATURES.pop(name)

import ab_kernels  # noqa: E402

ab_kernels.main()
