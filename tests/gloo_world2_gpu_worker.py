"""Child process of tests/test_nccl_gpu.py::test_two_ranks_share_the_gpu_over_gloo: rank RANK of a world_size-2 `gloo`
group; both ranks drive the ONE GPU of the box (RCCL refuses two ranks per device; gloo stages device tensors through
the host), each running the real HIP kernels on its `stride_data` shard of the data.  Checks, per rank:
  * one step on the reference golden bsc_step_c1_anneal_cut (N = 333: ragged shards, annealed prior, data truncation --
    the distributed radix select) equals the reference's single-process output;
  * five EM steps at config-2 dimensions (D=1024 H=256 H'=8 gamma=4, N = 2001: the 8-wavefront fused kernel with
    M-step statistics, then truncation steps) follow the oracle's single-process trajectory, and after EVERY step the
    two ranks hold bitwise identical W / pi / sigma (everything that decides a code path is a
    function of all-reduced data evaluated in a fixed order).
Prints "ok <rank>" on success."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")

import numpy as np
import torch
import torch.distributed as dist


class An(dict):
    crit_params = []

    def __missing__(self, k):
        return 0.0

    def as_dict(self):
        return dict(self)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from conftest import golden
        from oracle import bsc_oracle as O
        from prosper_amd.utils import parallel
        from prosper_amd.utils.datalog import dlog, StoreInMemory
        from prosper_amd.em.camodels.bsc_et import BSC_ET

        comm = parallel.Comm()
        assert (comm.rank, comm.size) == (rank, world)

        def same_on_all_ranks(new, what):
            for k in ("W", "pi", "sigma"):
                parts = comm.allgather(np.ascontiguousarray(np.asarray(new[k], dtype=np.float64)))
                for p in parts[1:]:
                    assert np.array_equal(parts[0], p), "%s: %s differs between ranks (max %.3e)" % (
                        what, k, np.abs(parts[0] - p).max())

        # ---- the reference's golden, sharded
        g = golden("bsc_step_c1_anneal_cut.npz")
        N = g["y"].shape[0]
        lo, hi = parallel.stride_data(N, comm=comm)
        assert N % world != 0 and hi - lo == N // world + (1 if rank < N % world else 0)
        m = BSC_ET(int(g["D"]), int(g["H"]), int(g["Hprime"]), int(g["gamma"]), comm=comm)
        an = An(T=float(g["T"]), Ncut_factor=float(g["Ncut_factor"]), anneal_prior=bool(g["anneal_prior"]))
        params = {"W": g["W"].copy(), "pi": float(g["pi"]), "sigma": float(g["sigma"])}
        h = dlog.set_handler(("L", "N", "N_use"), StoreInMemory)
        try:
            new = m.step(an, params, {"y": g["y"][lo:hi].copy()})
        finally:
            dlog.remove_handler(h)
        np.testing.assert_allclose(new["W"], g["W_new"], rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose([new["pi"], new["sigma"]], [g["pi_new"], g["sigma_new"]], rtol=1e-10)
        if rank == 0:       # (dlog is rank 0's)
            assert int(h.tables["N_use"][0]) == int(g["N_use"])
            np.testing.assert_allclose(float(h.tables["L"][0]), float(g["L"]), rtol=1e-10)
        same_on_all_ranks(new, "golden step")

        # ---- config-2 dimensions: the fused kernels, 5 EM steps, against the oracle's single-process trajectory
        D, H, Hp, gamma, N = 1024, 256, 8, 4, 2001
        rng = np.random.RandomState(17)
        W_gt = rng.normal(size=(D, H))
        y, _ = O.generate_bsc_data(W_gt, 3.0 / H, 1.0, N, rng)
        p0 = {"W": W_gt + 0.1 * rng.normal(size=(D, H)), "pi": 3.0 / H, "sigma": 1.05}
        lo, hi = parallel.stride_data(N, comm=comm)
        m = BSC_ET(D, H, Hp, gamma, comm=comm)
        assert m._fused()
        model = O.make_model(D, H, Hp, gamma)
        plan = [(1.1, 0.0), (1.0, 0.0), (1.0, 0.0), (1.0, 0.8), (1.0, 0.8)]       # (T, Ncut_factor)
        p, ref = dict(p0), dict(p0)
        shard = {"y": y[lo:hi].copy()}
        for step, (T, ncut) in enumerate(plan):
            p = m.step(An(T=T, Ncut_factor=ncut), p, shard)
            same_on_all_ranks(p, "config-2 step %d" % step)
            if rank == 0:
                ref, _ = O.em_step(O.Anneal(T=T, Ncut_factor=ncut), model, ref, y, stats_fn=O.m_step_stats_vec, vec=True)
                np.testing.assert_allclose(p["W"], ref["W"], rtol=1e-6, atol=1e-8)
                np.testing.assert_allclose([p["pi"], p["sigma"]], [ref["pi"], ref["sigma"]], rtol=1e-8)
        comm.Barrier()
        print("ok %d" % rank)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
