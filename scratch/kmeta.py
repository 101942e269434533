"""Register / spill metadata of the kernels in an object file: python scratch/kmeta.py gsc_kernels.o [name filter]"""
import os, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from test_abi_and_host import _kernel_metadata
obj = sys.argv[1]
if not os.path.exists(obj):
    obj = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "prosper_amd", "csrc", "build", obj)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as t:
    for n, md in _kernel_metadata(obj, t):
        if flt in n:
            print(n[:110], md)
