"""Child process of tests/test_nccl_gpu.py::test_two_ranks_share_the_gpu_over_gloo: rank RANK of a world_size-2 `gloo`
group; both ranks drive the ONE GPU of the box (RCCL refuses two ranks per device; gloo stages device tensors through
the host), each running the real HIP kernels on its `stride_data` shard of the data.  Checks, per rank:
  * one step on the reference golden bsc_step_c1_anneal_cut (N = 333: ragged shards, annealed prior, data truncation --
    the distributed radix select) equals the reference's single-process output;
  * five EM steps at config-2 dimensions (D=1024 H=256 H'=8 gamma=4, N = 2001: the 8-wavefront fused kernel with
    M-step statistics, then truncation steps) follow the oracle's single-process trajectory, and after EVERY step the
    two ranks hold bitwise identical W / pi / sigma (everything that decides a code path is a
    function of all-reduced data evaluated in a fixed order);
  * MCA at config-5 dimensions (D=256 H=128 H'=8 gamma=3, N = 1001: the fused E-step + M-statistics pass and its f64
    atomics on both ranks, then truncation steps through the distributed radix select) and GSC at config-4 dimensions
    (D=256 H=128 H'=6 gamma=3, N = 1001: per-XCD statistics scratch, the batched warm inverse, the device-side M-step
    tail), three EM steps each against the oracle's single-process steps (MCA: its free-running trajectory; GSC: the oracle's
    step from the same parameters), bitwise rank identity after every step.
Last, BSC at config-2 dimensions in DETERMINISTIC mode (libprosper_hip_det.so): the same sharded loop twice, bitwise equal between
the runs and between the ranks.
Rank 0 holds the oracle; its verdict is shared after every step (`agree`), so a failed comparison ends BOTH ranks at
once with the assertion's text instead of leaving rank 1 in the next collective until the parent's timeout.
Prints "ok <rank>" on success."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")

import datetime
import traceback

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
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    try:
        from conftest import golden
        from oracle import bsc_oracle as O
        from prosper_amd.utils import parallel
        from prosper_amd.utils.datalog import dlog, StoreInMemory
        from prosper_amd.em.camodels.bsc_et import BSC_ET

        comm = parallel.Comm()
        assert (comm.rank, comm.size) == (rank, world)

        def agree(check, what):
            """Run rank 0's comparison, share the verdict: every rank raises if it failed."""
            err = None
            if rank == 0:
                try:
                    check()
                except Exception:
                    err = traceback.format_exc()
            err = comm.bcast(err)
            if err is not None:
                raise AssertionError("%s failed on rank 0:\n%s" % (what, err))

        def same_on_all_ranks(new, what, keys=("W", "pi", "sigma")):
            for k in keys:
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
        def check_golden():
            np.testing.assert_allclose(new["W"], g["W_new"], rtol=1e-8, atol=1e-10)
            np.testing.assert_allclose([new["pi"], new["sigma"]], [g["pi_new"], g["sigma_new"]], rtol=1e-10)
            assert int(h.tables["N_use"][0]) == int(g["N_use"])       # (dlog is rank 0's)
            np.testing.assert_allclose(float(h.tables["L"][0]), float(g["L"]), rtol=1e-10)
        agree(check_golden, "golden step")
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
        p, ref = dict(p0), [dict(p0)]
        shard = {"y": y[lo:hi].copy()}
        for step, (T, ncut) in enumerate(plan):
            p = m.step(An(T=T, Ncut_factor=ncut), p, shard)
            same_on_all_ranks(p, "config-2 step %d" % step)

            def check_bsc():
                ref[0], _ = O.em_step(O.Anneal(T=T, Ncut_factor=ncut), model, ref[0], y, stats_fn=O.m_step_stats_vec, vec=True)
                np.testing.assert_allclose(p["W"], ref[0]["W"], rtol=1e-6, atol=1e-8)
                np.testing.assert_allclose([p["pi"], p["sigma"]], [ref[0]["pi"], ref[0]["sigma"]], rtol=1e-8)
            agree(check_bsc, "config-2 step %d" % step)
        del m, shard

        # ---- MCA at config-5 dimensions: fused E-step + M-statistics pass, then truncation steps
        from oracle import mca_oracle as MO
        from prosper_amd.em.camodels.mca_et import MCA_ET
        D, H, Hp, gamma, N = 256, 128, 8, 3, 1001
        rng = np.random.RandomState(23)
        Wm = np.abs(rng.normal(size=(D, H))) * 2 + 0.1
        ym, _ = MO.generate_mca_data(Wm, 2.0 / H, 1.0, N, rng)
        p0 = {"W": Wm * (1 + 0.1 * rng.uniform(-1, 1, size=(D, H))), "pi": 2.0 / H, "sigma": 1.0}
        lo, hi = parallel.stride_data(N, comm=comm)
        assert N % world != 0
        mm = MCA_ET(D, H, Hp, gamma, comm=comm)
        mmodel = MO.make_model(D, H, Hp, gamma)
        p, ref = dict(p0), [dict(p0)]
        shard = {"y": ym[lo:hi].copy()}
        for step, (T, ncut) in enumerate([(1.2, 0.0), (1.0, 0.0), (1.0, 0.7)]):
            p = mm.step(An(T=T, Ncut_factor=ncut), p, shard)
            same_on_all_ranks(p, "config-5 step %d" % step, keys=("W", "pi", "sigma", "Q"))

            def check_mca():
                ref[0], _ = MO.em_step(MO.Anneal(T=T, Ncut_factor=ncut), mmodel, ref[0], ym, vec=True)
                np.testing.assert_allclose(p["W"], ref[0]["W"], rtol=1e-7, atol=1e-9)
                np.testing.assert_allclose([p["pi"], p["sigma"]], [ref[0]["pi"], ref[0]["sigma"]], rtol=1e-8)
                ref[0] = {k: ref[0][k] for k in ("W", "pi", "sigma")}
            agree(check_mca, "config-5 step %d" % step)
            p = {k: p[k] for k in ("W", "pi", "sigma")}
        del mm, shard

        # ---- GSC at config-4 dimensions
        from oracle import gsc_oracle as GO
        from prosper_amd.em.camodels.gsc_et import GSC
        D, H, Hp, gamma, N = 256, 128, 6, 3, 1001
        rng = np.random.RandomState(29)
        gt = {"W": rng.normal(size=(D, H)), "pi": np.full(H, 2.0 / H), "mu": np.full(H, 1.5), "psi_sq": np.eye(H),
              "sigma_sq": 1.0}
        yg, _, _ = GO.generate_gsc_data(gt, N, rng)
        Q = 0.05 * rng.normal(size=(H, H))
        p0 = {"W": gt["W"] + 0.1 * rng.normal(size=(D, H)),
              "pi": np.clip(gt["pi"] * rng.uniform(0.8, 1.3, size=H), 0.01, 0.9), "mu": gt["mu"] + 0.1 * rng.normal(size=H),
              "psi_sq": np.diag(rng.uniform(0.7, 1.4, size=H)) + Q @ Q.T, "sigma_sq": 1.2}
        lo, hi = parallel.stride_data(N, comm=comm)
        mg = GSC(D, H, Hp, gamma, "scalar", comm=comm)
        gmodel = GO.make_model(D, H, Hp, gamma)
        gkeys = ("W", "pi", "mu", "psi_sq", "sigma_sq")
        p = {k: np.array(v, copy=True) for k, v in p0.items()}
        shard = {"y": yg[lo:hi].copy()}
        for step, T in enumerate([1.1, 1.0, 1.0]):
            # (every step against the oracle's step FROM THE SAME PARAMETERS: with ~16 active datapoints per latent the W
            # solve is ill-conditioned enough that free-running trajectories drift apart by more than one step's tolerance)
            start = {k: np.array(v, copy=True) for k, v in p.items()}
            p = mg.step(An(T=T), p, shard)
            same_on_all_ranks(p, "config-4 step %d" % step, keys=gkeys)

            def check_gsc():
                ref, log = GO.em_step(GO.Anneal(T=T), gmodel, start, yg)
                cond = np.linalg.cond(log["suff"]["xpt_szsz"].sum(0))
                tol = max(1e-8, 50 * cond * np.finfo(float).eps)
                for k in gkeys:
                    np.testing.assert_allclose(p[k], ref[k], rtol=10 * tol, atol=tol * max(1.0, np.abs(ref[k]).max()),
                                               err_msg="%s (step %d, cond %.2e)" % (k, step, cond))
            agree(check_gsc, "config-4 step %d" % step)
            p = {k: np.array(p[k], copy=True) for k in gkeys}
        # ---- deterministic mode across ranks: the same sharded 4-step loop twice -- fused kernels, then a
        # truncation step through the distributed radix select -- bit for bit the same W / pi / sigma in both runs, on both ranks
        D, H, Hp, gamma, N = 1024, 256, 8, 4, 2001
        rng = np.random.RandomState(31)
        W_gt = rng.normal(size=(D, H))
        y, _ = O.generate_bsc_data(W_gt, 3.0 / H, 1.0, N, rng)
        p0 = {"W": W_gt + 0.1 * rng.normal(size=(D, H)), "pi": 3.0 / H, "sigma": 1.05}
        lo, hi = parallel.stride_data(N, comm=comm)
        runs = []
        for rep in range(2):
            md = BSC_ET(D, H, Hp, gamma, comm=comm)
            md.deterministic = True
            p, traj = dict(p0), []
            shard = {"y": y[lo:hi].copy()}
            for step, (T, ncut) in enumerate([(1.1, 0.0), (1.0, 0.0), (1.0, 0.8), (1.0, 0.8)]):
                p = md.step(An(T=T, Ncut_factor=ncut), p, shard)
                same_on_all_ranks(p, "deterministic run %d step %d" % (rep, step))
                traj.append({k: np.array(p[k], copy=True) for k in ("W", "pi", "sigma")})
            runs.append(traj)
            del md
        for step, (a, b) in enumerate(zip(*runs)):
            for k in ("W", "pi", "sigma"):
                assert np.array_equal(a[k], b[k]), "deterministic mode, rank %d: step %d %s differs between two runs" % (rank, step, k)
        # ---- ... and GSC (round 6: the mode keeps the speculative list pass -- quanta derived on the device, dense rows sorted):
        # a sharded 6-step loop at config-4 dimensions twice, bit for bit the same parameters in both runs, on both ranks
        D, H, Hp, gamma, N = 256, 128, 6, 3, 9001
        rng = np.random.RandomState(33)
        gt = {"W": rng.normal(size=(D, H)), "pi": np.full(H, 2.0 / H), "mu": np.full(H, 1.5), "psi_sq": np.eye(H), "sigma_sq": 1.0}
        yg, _, _ = GO.generate_gsc_data(gt, N, rng)
        pg0 = {"W": gt["W"] + 0.1 * rng.normal(size=(D, H)), "pi": gt["pi"] * 1.1, "mu": gt["mu"] + 0.1 * rng.normal(size=H),
               "psi_sq": np.diag(rng.uniform(0.7, 1.4, size=H)), "sigma_sq": 1.2}
        lo, hi = parallel.stride_data(N, comm=comm)
        runs, hits = [], []
        for rep in range(2):
            mg = GSC(D, H, Hp, gamma, "scalar", comm=comm)
            mg.deterministic = True
            p, traj = {k: np.array(v, copy=True) for k, v in pg0.items()}, []
            shard = {"y": yg[lo:hi].copy()}
            for step in range(6):
                p = mg.step(An(T=1.2 if step < 2 else 1.0), p, shard)
                same_on_all_ranks(p, "deterministic GSC run %d step %d" % (rep, step), keys=("W", "pi", "mu", "psi_sq", "sigma_sq"))
                traj.append({k: np.array(p[k], copy=True) for k in ("W", "pi", "mu", "psi_sq", "sigma_sq")})
            runs.append(traj)
            hits.append(mg.spec_hits)
            del mg
        assert hits[0] == hits[1] and hits[0] >= 2, hits
        for step, (a, b) in enumerate(zip(*runs)):
            for k in a:
                assert np.array_equal(a[k], b[k]), "deterministic GSC, rank %d: step %d %s differs between two runs" % (rank, step, k)
        comm.Barrier()
        print("ok %d" % rank)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
