"""Exploratory: the full-size EM-step tests (BSC / MCA truncation step, GSC two steps, BSC every row) with every model in
deterministic mode (libprosper_hip_det.so) -- the same oracle comparisons."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from prosper_amd.em.camodels import _device
orig = _device.DeviceCAModel.__init__
def init(self, *a, **k):
    orig(self, *a, **k)
    self.deterministic = True
_device.DeviceCAModel.__init__ = init
import test_bsc_gpu, test_mca_gpu, test_gsc_gpu
dev = torch.device("cuda", 0)
for name, fn in (("bsc truncation step", lambda: test_bsc_gpu.test_config2_full_shard_truncation_step_against_oracle(dev)),
                 ("bsc every row + statistics", lambda: test_bsc_gpu.test_config2_full_shard_every_row_against_oracle(dev)),
                 ("mca truncation step", test_mca_gpu.test_config5_full_shard_truncation_step_against_oracle),
                 ("gsc full shard, two steps", test_gsc_gpu.test_config4_full_shard_against_oracle)):
    try:
        fn()
        print(name, ": ok (deterministic build)", flush=True)
    except AssertionError as e:
        print(name, ": FAILED", str(e)[:600], flush=True)
