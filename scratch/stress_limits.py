"""Stress tests/test_limits_gpu.py::test_bsc_off_config_shapes_match_the_oracle for rare failures: each shape R times in one
process (fresh model each time), every assertion reported with its detail.  PYTHONPATH=.:tests python scratch/stress_limits.py [R]"""
import sys, os, traceback
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import pytest
import test_limits_gpu as T
R = int(sys.argv[1]) if len(sys.argv) > 1 else 100
fails = 0
for shape in T._SWEEP:
    if shape[0] * shape[1] > 1024 * 512:
        continue
    n_skip = 0
    for r in range(R):
        try:
            T.test_bsc_off_config_shapes_match_the_oracle(*shape)
        except pytest.skip.Exception:
            n_skip += 1
        except Exception:
            fails += 1
            print("FAIL", shape, "run", r)
            traceback.print_exc(limit=2)
            if fails > 5:
                sys.exit(1)
    print(shape, "runs", R, "near-tie skips", n_skip, "failures so far", fails, flush=True)
