"""Child process of tests/test_nccl_gpu.py: a world_size-1 `nccl` (= RCCL) process group on the one GPU, created
before anything else touches the device, with collectives FORCED through the group (PM_FORCE_COLLECTIVES=1).  Every
collective call site of the hot path then really runs over RCCL: Comm.allreduce (device round trip on an nccl-only
group), allgather / bcast (all_gather_object / broadcast_object_list), allreduce_device (packed statistics),
the histogram all-reduces of the truncation cut -- inside full BSC / MCA / GSC steps with data truncation, checked
against the oracle.  Prints "ok" on success."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["PM_FORCE_COLLECTIVES"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from oracle import bsc_oracle as O
        from prosper_amd.utils import parallel
        from prosper_amd.em.camodels.bsc_et import BSC_ET

        comm = parallel.Comm()
        assert comm.size == 1 and not comm._solo()
        # mpi4py-shaped calls over an nccl-only group
        assert comm.allreduce(3) == 3 and comm.allreduce(0.5) == 0.5
        np.testing.assert_array_equal(comm.allreduce(np.arange(4.)), np.arange(4.))
        assert comm.allgather({"rows": 5}) == [{"rows": 5}]
        assert comm.bcast("x") == "x"
        buf = np.arange(3.)
        comm.Bcast([buf, parallel.DOUBLE])
        comm.Barrier()
        t = torch.arange(8, dtype=torch.float64, device="cuda")
        comm.allreduce_device(t)
        assert t.cpu().tolist() == list(range(8))

        class An(dict):
            crit_params = []

            def __missing__(self, k):
                return 0.0

            def as_dict(self):
                return dict(self)

        # BSC step with data truncation (kth-largest select: 6 histogram all-reduces) + the statistics all-reduce
        D, H, Hp, gamma, N = 48, 24, 5, 3, 900
        rng = np.random.RandomState(3)
        W_gt = rng.normal(size=(D, H))
        y, _ = O.generate_bsc_data(W_gt, 2.0 / H, 1.0, N, rng)
        params = {"W": W_gt + 0.2 * rng.normal(size=(D, H)), "pi": 2.4 / H, "sigma": 1.1}
        ref, rlog = O.em_step(O.Anneal(T=1.2, Ncut_factor=0.6), O.make_model(D, H, Hp, gamma), dict(params), y,
                              stats_fn=O.m_step_stats_vec, vec=True)
        m = BSC_ET(D, H, Hp, gamma, comm=comm)
        new = m.step(An(T=1.2, Ncut_factor=0.6), dict(params), {"y": y})
        np.testing.assert_allclose(new["W"], ref["W"], rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose([new["pi"], new["sigma"]], [ref["pi"], ref["sigma"]], rtol=1e-9)
        init = m.standard_init({"y": y})                                    # two moment all-reduces
        assert np.isfinite(init["W"]).all()
        # ... and without truncation (statistics from the fused E-step pass), a second step on the seeded parameters
        new2 = m.step(An(T=1.0), new, {"y": y})
        assert np.isfinite(new2["W"]).all()

        # MCA and GSC steps through the same communicator
        from oracle import mca_oracle as MO
        from prosper_amd.em.camodels.mca_et import MCA_ET
        Wm = np.abs(rng.normal(size=(D, H))) * 2 + 0.1
        s = rng.random_sample((N, H)) < 2.0 / H
        ym = np.where(s.any(1)[:, None], np.max(np.where(s[:, None, :], Wm[None, :, :], 0.0), axis=2), 0.0) + rng.normal(size=(N, D))
        pm = {"W": Wm * (1 + 0.1 * rng.uniform(-1, 1, size=(D, H))), "pi": 2.0 / H, "sigma": 1.0}
        mm = MCA_ET(D, H, Hp, gamma, comm=comm)
        newm = mm.step(An(T=1.3, Ncut_factor=0.5), dict(pm), {"y": ym})
        refm, _ = MO.em_step(MO.Anneal(T=1.3, Ncut_factor=0.5), MO.make_model(D, H, Hp, gamma), dict(pm), ym, vec=True)
        np.testing.assert_allclose(newm["W"], refm["W"], rtol=1e-7, atol=1e-8)

        from prosper_amd.em.camodels.gsc_et import GSC
        z = np.where(s, 1.5 + rng.normal(size=(N, H)), 0.0)
        yg = z @ W_gt.T + rng.normal(size=(N, D))
        pg = {"W": W_gt + 0.2 * rng.normal(size=(D, H)), "pi": np.full(H, 2.0 / H), "mu": np.full(H, 1.4),
              "psi_sq": np.eye(H) * 1.1, "sigma_sq": 1.2}
        mg = GSC(D, H, Hp, gamma, 'scalar', comm=comm)
        newg = mg.step(An(T=1.0), dict(pg), {"y": yg})
        assert np.isfinite(newg["W"]).all() and np.isfinite(newg["sigma_sq"])
        print("ok")
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
