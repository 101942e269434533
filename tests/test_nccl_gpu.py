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


@pytest.mark.timeout(900)
def test_bench_multi_rank_branch_over_gloo(dev):
    """`bench.py --gpus 2`'s multi-rank branch end to end on the one-GPU box: the self-launch through
    torch.distributed.run, process-group init, the device / seed self-checks, per-rank records and their gather -- with
    `--backend gloo` (two ranks share the device; RCCL would refuse).  One JSON line, n_gpus = 2, two `per_rank`
    entries with distinct data seeds, a timed statistics all-reduce."""
    import json
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                        "--n-per-gpu", "33000", "--steps", "2", "--warmup", "1", "--em-steps", "2", "--prewarm-ms", "5",
                        "--no-cpu-baseline", "--no-other-models"], env=env, capture_output=True, text=True, timeout=840)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak" and out["unit"] == "datapoints/s"
    assert out["config"]["global_datapoints"] == 66000 and out["config"]["parallelism"] == "dp2"
    assert out["config"]["backend"].startswith("gloo")
    pr = out["per_rank"]
    assert [e["rank"] for e in pr] == [0, 1] and len({e["data_seed"] for e in pr}) == 2
    assert all(e["rows"] == 33000 and e["ms_per_step"] > 0 and e["em_iter_ms"] > 0 and e["allreduce_us"] > 0 for e in pr)
    assert out["value"] > 0 and out["roofline"]["frac"] > 0 and out["cpu_baseline"] is None
    # value = all ranks' datapoints / the slowest rank's time
    slow = max(e["ms_per_step"] for e in pr)
    assert abs(out["value"] - 66000 / (slow * 1e-3)) <= 1e-6 * out["value"]


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


@pytest.mark.parametrize("n", [1, 5, 1000, 70001, 200000])
def test_kth_select_with_folded_scans_matches_sort(dev, n):
    """pm_kth_round_f64 / pm_kth_final_f64 (round 6: every round's kernel repeats the previous round's scan in each
    workgroup -- 1 + 6 + 1 launches, the value stays on the device): two 'ranks' histogram into the same per-round table
    (what the all-reduce amounts to; the second shard's call repeats the scan of a table that is complete by then only for
    the FIRST shard's round -- so the shards take turns round by round through two state/histogram sets, summed by hand),
    incl. duplicates, +-0, -inf and denormals; and the host-level entry ``_kth_select_dev`` on one shard."""
    from prosper_amd import _lib
    from prosper_amd.em.camodels._device import DeviceCAModel
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    rng = np.random.RandomState(n)
    x = np.concatenate([rng.normal(size=n) * 50 - 600.0, [-np.inf, 0.0, -0.0, 1e-310, -1e-310, 3.5, 3.5]])
    rng.shuffle(x)
    cut = len(x) // 3
    shards = [torch.from_numpy(x[:cut].copy()).to(dev), torch.from_numpy(x[cut:].copy()).to(dev)]
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    R = DeviceCAModel.KTH_ROUNDS
    m = BSC_ET(8, 4, 2, 2)
    whole = torch.from_numpy(x).to(dev)
    for k in sorted({1, 2, len(x) // 2, len(x) - 1, len(x)}):
        want = float(np.sort(x)[-k])
        # two ranks: each has its own buffer; the "all-reduce" adds the round's histograms before the next round reads them
        bufs = [torch.zeros(16 + 6 * 4096, dtype=torch.int64, device=dev) for _ in shards]
        for b in bufs:
            b[1] = k
        prev = (0, 1)
        for r, (shift, bits) in enumerate(R):
            if r:
                tot = bufs[0][16:].view(6, 4096)[r - 1] + bufs[1][16:].view(6, 4096)[r - 1]
                for b in bufs:
                    b[16:].view(6, 4096)[r - 1] = tot
            for sh, b in zip(shards, bufs):
                _lib.call("pm_kth_round_f64", _p(sh) if sh.numel() else None, sh.numel(), _p(b[:14]), _p(b[16:]), r, prev[0],
                          prev[1], shift, bits, st)
            prev = (shift, bits)
        tot = bufs[0][16:].view(6, 4096)[5] + bufs[1][16:].view(6, 4096)[5]
        outs = []
        for b in bufs:
            b[16:].view(6, 4096)[5] = tot
            out = torch.empty(1, dtype=torch.float64, device=dev)
            _lib.call("pm_kth_final_f64", _p(b[:14]), _p(b[16:]), 6, prev[0], prev[1], _p(out), st)
            outs.append(float(out.item()))
        # (np.sort leaves +0.0 and -0.0, which compare equal, in either order: the sign is checked away from zero only)
        assert outs[0] == outs[1] == want and (want == 0 or np.signbit(outs[0]) == np.signbit(want)), (k, outs, want)
        got = float(m._kth_select_dev(whole, k).item())
        assert got == want and (want == 0 or np.signbit(got) == np.signbit(want)), (k, got, want)
        assert m._kth_largest_global(whole, k) == want
