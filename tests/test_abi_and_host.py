"""CPU-side checks: the C-ABI library loads and exports every symbol include/prosper_hip.h
declares (no compute without a GPU), and the host-side mirror of the reference's driver
(LinearAnnealing, EM loop, state table, dlog, comm helpers) behaves like the reference."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT, golden


def test_library_exports_every_declared_symbol():
    from prosper_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "prosper_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(pm_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), "libprosper_hip.so does not export %s" % name
    assert declared == set(_lib.SIGNATURES), "ctypes table and header disagree: %s" % (
        declared ^ set(_lib.SIGNATURES))
    assert lib.pm_version() >= 1000
    assert lib.pm_error_string(0) == b"ok"
    assert b"invalid" in lib.pm_error_string(-1)


def test_deterministic_build_exports_the_same_abi():
    """libprosper_hip_det.so (the same sources with -DPM_DETERMINISTIC) exports every declared symbol too, says so, and the default
    build refuses quanta (its kernels contain no quantisation)."""
    from prosper_amd import _lib
    import ctypes
    det, lib = _lib.load(det=True), _lib.load()
    for name in _lib.SIGNATURES:
        assert hasattr(det, name), name
    assert det.pm_det_build() == 1 and lib.pm_det_build() == 0
    M = (ctypes.c_double * 8)()
    assert lib.pm_det_set_quanta(0, M, None) == -2
    assert det.pm_det_set_quanta(99, M, None) == -1


def test_stats_layout_helpers():
    from prosper_amd import _lib
    lib = _lib.load()
    H, D = 256, 1024
    assert lib.pm_bsc_stats_offset_wq(H, D) == H * D
    assert lib.pm_bsc_stats_offset_qdiag(H, D) == H * D + H * H
    assert lib.pm_bsc_stats_offset_mus(H, D) == H * D + H * H + H
    assert lib.pm_bsc_stats_len(H, D) == H * D + H * H + 2 * H + 4


def test_bad_arguments_are_rejected_without_a_gpu():
    from prosper_amd import _lib
    lib = _lib.load()
    assert lib.pm_gemm_nt_f64(None, 1, None, 1, None, 1, 1, 1, 1, None) == -1
    assert lib.pm_bsc_select_f64(None, 1, None, 1, None, 1, 1, 1, None, None) == -1
    with pytest.raises(_lib.HipError):
        _lib.call("pm_row_sqnorm_f64", None, 1, 1, 1, None, None)


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from prosper_amd import _lib
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    m = BSC_ET(25, 10, 5, 3)
    with pytest.raises(_lib.HipError):
        m.select_Hprimes({"W": np.ones((25, 10))}, {"y": np.ones((4, 25))})


def test_annealing_tracks_match_reference():
    from prosper_amd.em.annealing import LinearAnnealing
    g = golden("anneal_tracks.npz")
    a = LinearAnnealing(int(g["steps"]))
    a["T"] = [(0, 2.), (.7, 1.)]
    a["Ncut_factor"] = [(0, 0.), (2. / 3, 1.)]
    a["anneal_prior"] = False
    a["W_noise"] = [(0, 0.5), (-10, 0.0)]
    names = [str(n) for n in g["names"]]
    rows = []
    while not a.finished:
        rows.append([float(a[n]) for n in names])
        a.next()
    np.testing.assert_allclose(np.array(rows), g["values"], rtol=0, atol=0)
    with pytest.raises(RuntimeError):
        a.next()
    with pytest.raises(TypeError):
        a["bad"] = [1.0, 2.0]


def test_state_matrix_matches_reference_fixture():
    from prosper_amd.em.camodels import generate_state_matrix
    g = golden("bsc_step_h256.npz")
    sl, S, SM, sabs = generate_state_matrix(int(g["Hprime"]), int(g["gamma"]))
    assert S == 154 and np.array_equal(SM, g["state_matrix"]) and np.array_equal(sabs, g["state_abs"])
    assert SM.dtype == np.uint8 and all(len(s) == a for s, a in zip(sl, sabs))


class _CountingModel(object):
    """Minimal Model: records the annealing values every step sees."""

    def __init__(self):
        self.seen = []

    def step(self, anneal, params, data):
        self.seen.append((anneal["T"], anneal["step"]))
        return {"x": params["x"] + 1}

    def gain(self, old, new):
        return 0.


def test_em_run_loop_contract():
    from prosper_amd.em import EM
    from prosper_amd.em.annealing import LinearAnnealing
    a = LinearAnnealing(7)
    a["T"] = [(0, 3.), (-1, 1.)]
    em = EM(model=_CountingModel(), anneal=a, data={}, lparams={"x": 0})
    em.run()
    assert em.lparams == {"x": 7}
    assert len(em.model.seen) == 7 and em.model.seen[0] == (3.0, 0.0)
    assert a.finished


def test_dlog_policy_and_wildcard():
    from prosper_amd.utils.datalog import DataLog, StoreInMemory
    d = DataLog()
    h1 = d.set_handler(("L", "N"), StoreInMemory)
    h2 = d.set_handler("*", StoreInMemory)
    d.append("L", 1.5)
    d.append_all({"W": np.ones(3), "N": 7})
    assert [float(v) for v in h1.tables["L"]] == [1.5] and int(h1.tables["N"][0]) == 7
    assert set(h2.tables) == {"L", "W", "N"} and "W" not in h1.tables
    assert not d.ignored("anything") and DataLog().ignored("L")
    d.remove_handler(h2)
    d.append("W", np.zeros(3))
    assert len(h2.tables["W"]) == 1


def test_single_rank_comm_and_helpers():
    from prosper_amd.utils import parallel
    c = parallel.Comm()
    assert (c.rank, c.size) == (0, 1)
    assert c.allreduce(5) == 5
    a, b = np.arange(6.).reshape(2, 3), np.empty((2, 3))
    c.Allreduce([a, parallel.DOUBLE], [b, parallel.DOUBLE])
    assert np.array_equal(a, b)
    x = np.array([3., 1., 2.])
    assert np.array_equal(parallel.allsort(x), np.array([1., 2., 3.]))
    assert parallel.allmean(np.ones((4, 3)), axis=0).tolist() == [1., 1., 1.]
    assert parallel.allsum(np.ones((4, 3))) == 12
    assert parallel.stride_data(10) == (0, 10)


def test_standard_init_and_generate_data_rng_stream():
    """Host-side CAModel mirror reproduces the reference's RNG stream (golden from reference)."""
    from prosper_amd.em.camodels.bsc_et import BSC_ET
    g = golden("bsc_init_c1.npz")
    m = BSC_ET(int(g["D"]), int(g["H"]), 5, 3)
    np.random.seed(int(g["seed_data"]))
    data = m.generate_data({"W": g["W_gt"], "pi": float(g["pi_gt"]), "sigma": float(g["sigma_gt"])}, int(g["N"]))
    assert np.array_equal(data["s"], g["s"])
    np.testing.assert_allclose(data["y"], g["y"], rtol=1e-13, atol=1e-13)
    np.random.seed(int(g["seed_init"]))
    from prosper_amd.em.camodels import CAModel
    init = CAModel.standard_init(m, {"y": g["y"]})          # host mirror; the device version is a GPU test
    np.testing.assert_allclose(init["W"], g["W0"], rtol=1e-13)
    np.testing.assert_allclose(init["sigma"], g["sigma0"], rtol=1e-13)
    assert init["pi"] == float(g["pi0"])


def test_store_to_npz_and_resume(tmp_path):
    """StoreToNpz rows + resume_params: the npz stand-in for result.h5 (one array per logged name, one row
    per EM step) and reading the last row back as lparams."""
    from prosper_amd.utils.datalog import DataLog, StoreToNpz, resume_params
    log = DataLog()
    dest = str(tmp_path / "result")
    log.set_handler(("W", "pi", "sigma", "L"), StoreToNpz, dest)
    for step in range(3):
        log.append_all({"W": np.full((4, 2), float(step)), "pi": 0.1 * (step + 1), "sigma": 1.0 + step})
        log.append("L", -10.0 + step)
    log.close()
    f = np.load(dest + ".npz")
    assert f["W"].shape == (3, 4, 2) and f["L"].tolist() == [-10.0, -9.0, -8.0]
    p = resume_params(dest, names=("W", "pi", "sigma"))
    assert sorted(p) == ["W", "pi", "sigma"] and p["sigma"] == 3.0 and abs(p["pi"] - 0.3) < 1e-15
    assert np.array_equal(p["W"], np.full((4, 2), 2.0))


def test_model_classes_build_the_reference_state_tables_on_cpu():
    """Constructors of the drop-in classes need no GPU and reproduce the reference's state tables
    (dsc_et.py:167-191, tsc_et.py:23-80, camodels/__init__.py:21-47)."""
    from prosper_amd.em.camodels.dsc_et import DSC_ET
    from prosper_amd.em.camodels.tsc_et import TSC_ET
    from prosper_amd.em.camodels.mmca_et import MMCA_ET
    from prosper_amd.em.camodels.mca_et import MCA_ET
    g = golden("dsc_step_k4.npz")
    m = DSC_ET(int(g["D"]), int(g["H"]), int(g["Hprime"]), int(g["gamma"]), states=g["states"])
    assert np.array_equal(m.state_matrix, g["state_matrix"]) and np.array_equal(m.state_abs, g["state_abs"])
    assert np.array_equal(m.single_state_matrix, g["single_state_matrix"]) and m._K_0 == 0 and m.K == 4
    with pytest.raises(TypeError):
        DSC_ET(4, 3, 2, 2, states=[-1., 0., 1.])
    g = golden("tsc_step_small.npz")
    t = TSC_ET(int(g["D"]), int(g["H"]), int(g["Hprime"]), int(g["gamma"]))
    assert np.array_equal(t.state_matrix, g["state_matrix"]) and t.no_states == int(g["no_states"]) == 3 ** 4
    g = golden("mmca_step_small.npz")
    mm = MMCA_ET(int(g["D"]), int(g["H"]), int(g["Hprime"]), int(g["gamma"]))
    assert np.array_equal(mm.state_matrix, g["state_matrix"]) and isinstance(mm, MCA_ET) and mm.signed_w == 1.0
    p = mm.check_params({"W": g["W"].copy(), "pi": 0.2, "sigma": 1.0})
    assert np.abs(p["W"]).min() >= mm.tol
    np.random.seed(3)
    d = mm.generate_data({"W": g["W_gt"], "pi": 0.3, "sigma": 0.0}, 5)
    np.random.seed(3)
    s = np.random.random(size=(5, mm.H)) < 0.3
    assert np.array_equal(d["s"], s)
    np.random.seed(4)
    dt = t.generate_data({"W": np.ones((t.D, t.H)), "pi": 0.4, "sigma": 0.0}, 6)
    np.random.seed(4)
    pr = np.random.random(size=(6, t.H))
    assert np.array_equal(dt["s"], np.where(pr < 0.2, -1, np.where(pr < 0.4, 1, 0)))
    np.testing.assert_allclose(dt["y"], dt["s"].sum(axis=1, keepdims=True) * np.ones((6, t.D)))


def test_powtab_header_is_what_the_generator_writes():
    """prosper_amd/csrc/pm_powtab.h (tables and polynomial coefficients of pm_pow_tab / pm_exp_tab, the MCA kernels'
    power function) is exactly the output of gen_powtab.py."""
    import contextlib
    import importlib.util
    import io
    path = os.path.join(ROOT, "prosper_amd", "csrc", "gen_powtab.py")
    spec = importlib.util.spec_from_file_location("gen_powtab", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        mod.main()
    with open(os.path.join(ROOT, "prosper_amd", "csrc", "pm_powtab.h")) as f:
        assert f.read() == buf.getvalue()


def test_pow_tab_algorithm_is_accurate_to_one_ulp():
    """The arithmetic of pm_pow_tab / pm_exp_tab (pm_common.h) restated operation by operation -- every fused
    multiply-add evaluated exactly and rounded once -- against a 200-bit reference: x^c for the exponents the MCA
    kernels use (1/rho, 1/rho - 1) and e^x on the online softmax's range."""
    mp = pytest.importorskip("mpmath")
    import re
    mp.mp.prec = 200
    src = open(os.path.join(ROOT, "prosper_amd", "csrc", "pm_powtab.h")).read()
    body = src.split("#define PM_POWTAB_VALUES")[1].split("#define PM_POW_A1")[0]
    tab = np.array([float.fromhex(x) for x in re.findall(r"-?0x1\.[0-9a-f]+p[+-]\d+", body)])
    assert tab.size == 384
    A = [float.fromhex(re.search(r"PM_POW_A%d (\S+)" % k, src).group(1)) for k in range(1, 7)]
    B = [float.fromhex(re.search(r"PM_POW_B%d (\S+)" % k, src).group(1)) for k in range(1, 6)]
    MAGIC = 6755399441055744.0

    def fma(a, b, c):
        return float(mp.mpf(a) * mp.mpf(b) + mp.mpf(c))

    def low_word(x):
        k = int(np.float64(x).view(np.uint64) & np.uint64(0xFFFFFFFF))
        return k - (1 << 32) if k >= (1 << 31) else k

    def pow_tab(x, c):
        hi = int(np.float64(x).view(np.uint64) >> np.uint64(32))
        e = float((hi >> 20) - 1023)
        idx = (hi >> 13) & 127
        m = float(x) / 2.0 ** e
        r, L = tab[2 * idx], tab[2 * idx + 1]
        d = fma(m, r, -1.0)
        p = A[5]
        for k in (4, 3, 2, 1, 0):
            p = fma(p, d, A[k])
        s = e + L
        t = fma(d, p, (e - s) + L)
        yh = c * s
        yl = fma(c, t, fma(c, s, -yh))
        sh = fma(yh, 128.0, MAGIC)
        k = low_word(sh)
        f = fma(sh - MAGIC, -0.0078125, yh) + yl
        q = B[4]
        for kk in (3, 2, 1, 0):
            q = fma(q, f, B[kk])
        E = tab[256 + (k & 127)]
        return float(np.ldexp(fma(E, f * q, E), k >> 7))

    def exp_tab(x):
        x = max(x, -708.0)
        sh = fma(x, 184.6649652337873, MAGIC)
        k = low_word(sh)
        kf = sh - MAGIC
        r = fma(kf, -1.4907929134926466e-12, fma(kf, -0.00541521234663378, x))
        q = 1.0 / 120.0
        for c in (1.0 / 24.0, 1.0 / 6.0, 0.5, 1.0):
            q = fma(q, r, c)
        E = tab[256 + (k & 127)]
        return float(np.ldexp(fma(E, r * q, E), k >> 7))

    rng = np.random.RandomState(0)
    xs = np.concatenate([np.exp(rng.uniform(-30, 30, 250)), 1.0 + rng.uniform(-1e-3, 1e-3, 30),
                         2.0 ** rng.randint(-20, 20, 20) * (1 + rng.uniform(-1e-12, 1e-12, 20))])
    for c in (1 / 21.0, 1 / 21.0 - 1.0, 0.5 - 1.0):
        worst = max(abs(mp.mpf(pow_tab(x, c)) / mp.power(mp.mpf(x), mp.mpf(c)) - 1) for x in xs)
        assert worst < 2.3e-16, (c, float(worst))
    ys = np.concatenate([rng.uniform(-700, 50, 250), rng.uniform(-1, 1, 50), [0.0, -1e-300]])
    worst = max(abs(mp.mpf(exp_tab(y)) / mp.exp(mp.mpf(y)) - 1) for y in ys)
    assert worst < 2.3e-16, float(worst)
    assert exp_tab(0.0) == 1.0


def _h5_or_skip():
    from prosper_amd.utils import autotable
    try:
        autotable._Lib.get()
    except autotable.HDF5Unavailable as e:
        pytest.skip(str(e))
    return autotable


def test_autotable_writes_the_reference_result_h5_layout(tmp_path):
    """AutoTable / StoreToH5 (prosper/utils/autotable.py:87-127, 234-278, datalog.py:53-93) on the HDF5 C library:
    one extendable array per name, one row per append, shuffle + zlib level 1, PyTables' EArray attributes -- checked
    by reading the file back and, independently, with the HDF5 tools' h5dump."""
    import shutil
    import subprocess
    at = _h5_or_skip()
    from prosper_amd.utils.datalog import StoreToH5, resume_params, dlog
    fname = str(tmp_path / "result.h5")
    rs = np.random.RandomState(3)
    rows_W = [rs.normal(size=(6, 4)) for _ in range(5)]
    h = dlog.set_handler(('W', 'pi', 'L', 'N_use', 'note'), StoreToH5, fname)
    try:
        for t, W in enumerate(rows_W):
            dlog.append('W', W)
            dlog.append_all({'pi': 0.1 + 0.01 * t, 'L': -100.0 + t})
            dlog.append('N_use', 1000 - t)
            dlog.append('note', "step %d" % t)
    finally:
        dlog.remove_handler(h)
        h.close()
    assert sorted(at.table_names(fname)) == ['L', 'N_use', 'W', 'note', 'pi']
    np.testing.assert_array_equal(at.read_table(fname, 'W'), np.stack(rows_W))
    np.testing.assert_array_equal(at.read_table(fname, 'N_use'), 1000 - np.arange(5))
    assert at.read_table(fname, 'N_use').dtype == np.int64 and at.read_table(fname, 'pi').dtype == np.float64
    np.testing.assert_array_equal(at.read_table(fname, 'W', rows=(1, 3)), np.stack(rows_W[1:3]))
    assert list(at.read_table(fname, 'note')) == ["step %d" % t for t in range(5)]
    last = resume_params(fname, ('W', 'pi', 'L'))
    np.testing.assert_array_equal(last['W'], rows_W[-1])
    assert last['pi'] == 0.1 + 0.04 and last['L'] == -96.0
    with pytest.raises(TypeError):          # a row of another shape, as upstream
        with at.AutoTable(str(tmp_path / "bad.h5")) as tbl:
            tbl.append('x', np.zeros(3))
            tbl.append('x', np.zeros(4))
    h5dump = shutil.which("h5dump") or "/opt/conda/bin/h5dump"
    if os.path.exists(h5dump):
        head = subprocess.run([h5dump, "-H", "-p", fname], capture_output=True, text=True, check=True).stdout
        assert "( 5, 6, 4 ) / ( H5S_UNLIMITED, 6, 4 )" in head and "PREPROCESSING SHUFFLE" in head
        assert "COMPRESSION DEFLATE { LEVEL 1 }" in head and "H5T_IEEE_F64LE" in head and "H5T_STD_I64LE" in head
        attrs = subprocess.run([h5dump, "-a", "/W/CLASS", "-a", "/W/EXTDIM", "-a", "/PYTABLES_FORMAT_VERSION", fname],
                               capture_output=True, text=True, check=True).stdout
        assert '"EARRAY"' in attrs and '"2.1"' in attrs
        data = subprocess.run([h5dump, "-d", "/L", "-y", "-w", "200", fname], capture_output=True, text=True, check=True).stdout
        assert "-100, -99, -98, -97, -96" in data


def test_gsc_resume_init_from_result_h5(tmp_path):
    """GSC.resume_init (gsc_et.py:112-160): parameters of the last logged step, sigma_sq converted between noise types."""
    at = _h5_or_skip()
    from prosper_amd.em.camodels.gsc_et import GSC
    from prosper_amd.utils.parallel import Comm
    D, H = 6, 4
    rs = np.random.RandomState(1)
    steps = [{'W': rs.normal(size=(D, H)), 'pi': rs.uniform(0.1, 0.3, H), 'mu': rs.normal(size=H),
              'psi_sq': np.eye(H) * (1 + t), 'sigma_sq': rs.uniform(0.5, 1.5, D)} for t in range(3)]
    fname = str(tmp_path / "result.h5")
    with at.AutoTable(fname) as tbl:
        for p in steps:
            tbl.append_all(p)
    try:
        for kind, want in (('diagonal', steps[-1]['sigma_sq']), ('scalar', steps[-1]['sigma_sq'].mean()),
                           ('full', np.diag(steps[-1]['sigma_sq']))):
            m = GSC.__new__(GSC)                      # host-only: no device needed for reading parameters back
            m.D, m.H, m.sigma_sq_type, m.comm = D, H, kind, Comm()
            p = GSC.resume_init.__wrapped__(m, fname) if hasattr(GSC.resume_init, "__wrapped__") else m.resume_init(fname)
            np.testing.assert_array_equal(p['W'], steps[-1]['W'])
            np.testing.assert_array_equal(p['psi_sq'], steps[-1]['psi_sq'])
            np.testing.assert_allclose(p['sigma_sq'], want)
    finally:
        pass


# ------------------------------------------------------------------------- code-object hygiene
def _kernel_metadata(obj_path, tmp):
    """[(mangled kernel name, {key: int})] from the AMDGPU code-object metadata of a hipcc object file: the gfx950 code
    object is unbundled from its .hip_fatbin section and its notes are read with llvm-readelf."""
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
    subprocess.run([os.path.join(llvm, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj_path, fat], check=True)
    subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
    notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
    out, cur = [], None
    for line in notes.splitlines():
        t = line.strip()
        if t.startswith(".name:"):
            cur = (t.split(":", 1)[1].strip(), {})
            out.append(cur)
        elif cur is not None and ":" in t and t.split(":", 1)[0] in (".vgpr_spill_count", ".vgpr_count", ".sgpr_spill_count",
                                                                     ".private_segment_fixed_size"):
            cur[1][t.split(":", 1)[0]] = int(t.split(":", 1)[1])
    return out


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/clang-offload-bundler"), reason="needs the ROCm LLVM tools")
def test_shipped_hot_kernels_do_not_spill(tmp_path):
    """The hot kernels are tuned to their register budgets: a vector-register spill to scratch inside
    them is a performance bug that no numerical test sees.  Every instantiation of the fused BSC E-step (8- / 16-wavefront
    tiles, with and without M-step statistics, the TAIL kernel), of the f64 GEMMs and of the BSC row kernels must report
    .vgpr_spill_count <= 4 in its code-object metadata, and the 16-wavefront kernel must fit 4 wavefronts per SIMD."""
    build = os.path.join(ROOT, "prosper_amd", "csrc", "build")
    seen = {}
    for obj, patterns in (("bsc_fused8.o", ("bsc_estep_fused8s_kernel",)), ("gemm_f64.o", ("gemm_nt_f64_dma_kernel", "gemm_tn_f64_dma_kernel")),
                          ("bsc_rows16.o", ("bsc_select_estep16_kernel", "bsc_mstep_rows16_kernel")),
                          ("bsc_wp_sparse.o", ("bsc_wp_sparse_kernel",)),
                          ("dsc_kernels.o", ("dsc_estep16_kernel", "dsc_mstep_rows16_kernel", "dsc_estep16_ms_kernelILi8ELi8ELi4E")),
                          ("gsc_kernels.o", ("gsc_estep_kernelILi8ELi3ELb0ELb1E",)),
                          ("gemm_small.o", ("gemm_nt_small_kernel", "gemm_nn_small_kernel"))):
        path = os.path.join(build, obj)
        if not os.path.exists(path):
            pytest.skip("no object files (library built elsewhere)")
        for name, md in _kernel_metadata(path, str(tmp_path)):
            if any(p in name for p in patterns):
                seen[name] = md
    assert len(seen) >= 25, sorted(seen)
    # (bsc_mstep_rows16_kernel -- the M-step's own pass after a data-truncation step -- was deliberately capped at 128
    # registers for four wavefronts per SIMD at the price of 12 spilled registers: 0.48 -> 0.42 ms)
    # (dsc_mstep_rows16_kernel<8, 8> likewise: 34 spilled registers at four wavefronts per SIMD, 0.19 vs 0.215 ms at three)
    # (round 4: dsc_estep16_ms_kernel<8, 8, 4> -- E-step + M-step statistics in one pass, ternary latents -- at three wavefronts
    # per SIMD with 20 spilled registers: 0.293 vs 0.312 ms at two without spills; gsc_estep_kernel<8, 3, false, true> at three
    # wavefronts per SIMD with 3)
    limit = lambda n: (16 if "bsc_mstep_rows16_kernel" in n else 40 if "dsc_mstep_rows16_kernelILi8ELi8E" in n
                       else 24 if "dsc_estep16_ms_kernel" in n else 4)
    bad = {n: md for n, md in seen.items() if md.get(".vgpr_spill_count", 0) > limit(n)}
    assert not bad, bad
    # <STAGES, H', gamma, FULL, MSTATS, TAIL = false>: the 16-wavefront main launch, <= 128 registers (4 wavefronts per SIMD)
    main = {n: md for n, md in seen.items() if "bsc_estep_fused8s_kernel" in n and n.split("EEEv")[0].endswith("ELb0")}
    # (round 6: H' = 5 .. 8 -> four times the eight instantiations of <gamma, FULL, MSTATS>)
    assert len(main) == 32 and all(md[".vgpr_count"] <= 128 for md in main.values()), main
    gsc = [md for n, md in seen.items() if "gsc_estep_kernel" in n]
    assert gsc and all(md[".vgpr_count"] <= 168 for md in gsc), gsc          # three wavefronts per SIMD


def test_committed_bench_line_keeps_the_contract():
    """The newest committed bench line (profiles/r0N_v*_bench.json, written by `python bench.py` on the GPU box) carries
    every key of the driver's contract, the roofline and cpu_baseline objects, and figures that are consistent with each
    other (value = datapoints / time; frac = achieved / peak)."""
    import glob
    import json
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[3-9]_v*_bench.json")))
    assert files
    d = json.load(open(files[-1]))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["dtype"] == "f64" and d["scaling"] == "weak" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    n = d["config"]["global_datapoints"]
    assert abs(d["value"] - n / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    flops = 2.0 * r["datapoints_per_launch"] * 1024 * 256
    assert abs(r["achieved"] - flops / (r["avg_launch_ms"] * 1e-3) / 1e12) < 1e-6 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] > 0.9 * 2.2e9
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    # round 6: the EM-iteration wall-clock -- flat schedule and the reference's own (T and Ncut_factor ramps: the regime 49 of
    # a canonical run's 50 steps are in) -- also inside `roofline`, the sub-dict a driver record keeps
    if "em_iter_annealed_ms" in d:
        for k in ("em_iter_ms", "em_iter_steady_ms", "em_iter_annealed_ms", "em_iter_annealed_ramp_ms", "em_iter_annealed_plateau_ms"):
            assert r[k] > 0, k
        assert abs(r["em_iter_annealed_ms"] - d["em_iter_annealed_ms"]) < 1e-12
        a = d["em_iter_annealed"]
        assert a["steps"] == 50 and "Ncut_factor" in a["schedule"] and a["spec_hits"] >= 40
        # the schedule's mean stays within 1.15 of the flat steady-state figure (round 5: 1.49)
        assert d["em_iter_annealed_ms"] < 1.15 * d["em_iter_steady_ms"], (d["em_iter_annealed_ms"], d["em_iter_steady_ms"])
