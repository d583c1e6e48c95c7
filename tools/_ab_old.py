import sys, os, ctypes as C
import torch  # first: one HIP runtime per process (see _lib.load)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from voxelhashing_demo_amd import _lib
L = C.CDLL(_lib.LIB_PATH)
for n in list(_lib.SIGNATURES):
    if not hasattr(L, n):
        _lib.SIGNATURES.pop(n)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ab_kernels
ab_kernels.main()
