"""The RCCL code path on the one-GPU box: a world_size-1 `nccl` group with collectives forced through it runs full
BSC / MCA / GSC steps (tests/nccl_world1_worker.py, a fresh process: the group must exist before anything else
touches the device), and the kernels of the distributed k-th-largest select are pinned against numpy's sort."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs the GPU box (MI355X)")
    return torch.device("cuda", 0)


@pytest.mark.timeout(600)
def test_steps_over_a_world_size_1_rccl_group(dev):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "nccl_world1_worker.py")], env=env,
                       capture_output=True, text=True, timeout=540)
    lines = r.stdout.strip().splitlines()
    assert r.returncode == 0 and "ok" in lines, r.stdout[-2000:] + "\n" + r.stderr[-4000:]


@pytest.mark.timeout(900)
def test_two_ranks_share_the_gpu_over_gloo(dev):
    """Two processes, one GPU, a world_size-2 gloo group: each rank runs the HIP kernels on its ragged `stride_data`
    shard (tests/gloo_world2_gpu_worker.py); results equal the reference golden / the oracle's single-process
    trajectory and are bitwise identical on both ranks after every EM step."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2",
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "gloo_world2_gpu_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=840))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for rank, (p, (out, err)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("ok %d" % rank) in out.split("\n"), "rank %d\n%s\n%s" % (rank, out[-2000:], err[-4000:])


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


@pytest.mark.parametrize("n", [1, 5, 1000, 70001])
def test_kth_largest_kernels_match_sort(dev, n):
    """pm_kth_hist_f64 / pm_kth_scan / pm_kth_value_f64 over two 'ranks' (two shards histogrammed into one table --
    what the all-reduce of the bins amounts to), incl. duplicates, +-0, -inf and denormals."""
    from prosper_amd import _lib
    from prosper_amd.em.camodels._device import DeviceCAModel
    rng = np.random.RandomState(n)
    x = np.concatenate([rng.normal(size=n) * 50, [-np.inf, 0.0, -0.0, 1e-310, -1e-310, 3.5, 3.5]])
    rng.shuffle(x)
    cut = len(x) // 3
    shards = [torch.from_numpy(x[:cut].copy()).to(dev), torch.from_numpy(x[cut:].copy()).to(dev)]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for k in sorted({1, 2, len(x) // 2, len(x) - 1, len(x)}):
        state = torch.zeros(2, dtype=torch.int64, device=dev)
        state[1] = k
        hist = torch.zeros(4096, dtype=torch.int64, device=dev)
        for shift, bits in DeviceCAModel.KTH_ROUNDS:
            for sh in shards:
                _lib.call("pm_kth_hist_f64", _p(sh) if sh.numel() else None, sh.numel(), _p(state), shift, bits, _p(hist), st)
            _lib.call("pm_kth_scan", _p(hist), _p(state), shift, bits, st)
        out = torch.empty(1, dtype=torch.float64, device=dev)
        _lib.call("pm_kth_value_f64", _p(state), _p(out), st)
        got, want = float(out.item()), float(np.sort(x)[-k])
        assert got == want and np.signbit(got) == np.signbit(want), (k, got, want)
        assert int(hist.abs().sum()) == 0          # cleared for the next use
