"""Fused row-partitioned BPCG loop on the GPU: 2, 3 and 5 ranks (separate processes, all on the
one visible MI355X, gloo backend with host staging because RCCL refuses two ranks on one
device).  Exercises exactly what the multi-GPU bench runs -- nss_bpcg2_phase kernels on the
local CSR blocks, halo pack (nss_gather_f64) + all_to_all exchanges, all-reduced scalars,
device-side stop test -- and compares with the single-GPU fused solve."""

import contextlib
import io
import re

import numpy as np
import pytest

from staggered_grid import mac_stokes
from test_distributed_cpu import launch

pytestmark = pytest.mark.gpu


def single_gpu(dim, n, pre, tol, maxsteps):
    import hipla
    from solvers.bramblepasciak_new import BpcgSession
    s = mac_stokes(dim, n, 0.01)
    f, g = s.rhs(0)
    A, B = hipla.SparseMatrix.from_scipy(s.A), hipla.SparseMatrix.from_scipy(s.B)
    preA = hipla.BlockJacobi(A, s.line_blocks(3)) if pre == "bjac" else hipla.JacobiPreconditioner(A)
    preM = hipla.DiagonalMatrix(1.0 / s.mass)

    class Form:
        def __init__(self, mat):
            self.mat, self.condense = mat, False

    sol = hipla.BlockVector([hipla.Vector(s.n_u), hipla.Vector(s.n_p)])
    from hipla import eigen
    eigen.NATIVE = False       # the partitioned set-up estimates k with the protocol recurrence (all-reducing inner
    try:                       # product): the same recurrence here, so that the bit-for-bit comparisons below hold
        with contextlib.redirect_stdout(io.StringIO()):
            ses = BpcgSession(Form(A), Form(B), None, hipla.Vector.from_numpy(f), hipla.Vector.from_numpy(g), preA, preM,
                              sol=sol)
    finally:
        eigen.NATIVE = True
    ses.first_direction()
    it, hist, conv = ses.fused.run(ses.wdn, ses.err0, tol, True, maxsteps)
    assert conv
    return s, dict(it=it, hist=hist, k=ses.k, err0=ses.err0, u=sol[0].numpy(), p=sol[1].numpy())


def single_gpu_minres(s, pre, tol, maxsteps):
    import hipla
    from minres import MinRes
    f, g = s.rhs(0)
    A, B = hipla.SparseMatrix.from_scipy(s.A), hipla.SparseMatrix.from_scipy(s.B)
    preA = hipla.BlockJacobi(A, s.line_blocks(3)) if pre == "bjac" else hipla.JacobiPreconditioner(A)
    K = hipla.BlockMatrix([[A, B.T], [B, None]])
    Cm = hipla.BlockMatrix([[preA, None], [None, hipla.DiagonalMatrix(1.0 / s.mass)]])
    with contextlib.redirect_stdout(io.StringIO()):
        u, errors = MinRes(mat=K, pre=Cm, rhs=hipla.BlockVector([hipla.Vector.from_numpy(f), hipla.Vector.from_numpy(g)]),
                           maxsteps=maxsteps, tol=tol, printrates=False)
    return dict(errors=np.array(errors), u=u[0].numpy(), p=u[1].numpy())


def single_gpu_bpcg1(s, pre, tol, maxsteps):
    import hipla
    from bramble_pasciak_cg import bramble_pasciak_cg
    f, g = s.rhs(0)
    A, B = hipla.SparseMatrix.from_scipy(s.A), hipla.SparseMatrix.from_scipy(s.B)
    preA = hipla.BlockJacobi(A, s.line_blocks(3)) if pre == "bjac" else hipla.JacobiPreconditioner(A)
    from hipla import eigen
    eigen.NATIVE = False       # (as in single_gpu: the protocol recurrence for k, like the partitioned set-up)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            x, errors = bramble_pasciak_cg(A, B, None, preA, hipla.DiagonalMatrix(1.0 / s.mass), hipla.Vector.from_numpy(f),
                                           hipla.Vector.from_numpy(g), tolerance=tol, max_steps=maxsteps, print_rates=False)
    finally:
        eigen.NATIVE = True
    return dict(errors=np.array(errors), u=x[0].numpy(), p=x[1].numpy())


@pytest.mark.parametrize("world,dim,n,pre", [(2, 3, 10, "bjac"), (3, 2, 24, "jacobi"), (5, 3, 10, "jacobi")])
def test_fused_distributed_loop_matches_single_gpu(hip_engine, world, dim, n, pre):
    tol, maxsteps = 1e-8, 4000
    s, ref = single_gpu(dim, n, pre, tol, maxsteps)
    ranks = launch(world, "gpu", dim, n, pre, tol, maxsteps)
    for d in ranks:
        assert int(d["ghost_mode"]) == 2      # one halo exchange per iteration: t4's and s1's ghosts are kept locally
        assert int(d["compact"]) == 1         # ... on the compact plan: 6 launches + 3 collectives per iteration
        # compact plan == eight-phase plan, bit for bit (same launch plan of the owned rows of B in both)
        assert int(d["same_plan"]) == 1 and d["launch_forms"][0] == d["launch_forms"][1]
        assert int(d["it_classic"]) == int(d["it"])
        np.testing.assert_array_equal(d["hist_classic"], d["hist"])
        np.testing.assert_array_equal(d["u_classic"], d["u"])
        np.testing.assert_array_equal(d["p_classic"], d["p"])
        assert abs(d["k"] - ref["k"]) < 1e-9 * ref["k"]
        assert abs(d["err0"] - ref["err0"]) < 1e-10 * ref["err0"]
        assert d["err_AxBTp"] < 1e-12 and d["err_Bx"] < 1e-12
        np.testing.assert_array_equal(d["hist"], ranks[0]["hist"])        # same decision on every rank
        w = min(30, len(ref["hist"]), len(d["hist"]))
        np.testing.assert_allclose(d["hist"][:w], ref["hist"][:w], rtol=1e-8)
        assert abs(int(d["it"]) - ref["it"]) <= max(3, int(0.03 * ref["it"]))
    u = np.concatenate([d["u"] for d in ranks])
    p = np.concatenate([d["p"] for d in ranks])
    assert np.linalg.norm(u - ref["u"]) < 1e-5 * np.linalg.norm(ref["u"])
    p0, pr = p - p.mean(), ref["p"] - ref["p"].mean()
    assert np.linalg.norm(p0 - pr) < 1e-4 * np.linalg.norm(pr)
    # the mailbox transport (csrc/p2p.h) -- the `world` processes map each other's mailboxes / landing zones through
    # HIP IPC (here all on one GPU; over xGMI on a node): the all-reduce adds the ranks' values in rank order (the same
    # bits on every rank), the put / wait-copy exchange fills every ghost tail with the owners' entries, and the NATIVE
    # compact loop over it -- no collective library inside an iteration -- reproduces the gloo-driven loop
    for rep in range(6):
        want = 0.0
        for r in range(world):
            want += np.sin(1.0 + r + 10.0 * rep) * 1e3
        for d in ranks:
            assert d["mailbox_allreduce"][rep] == want
    for d in ranks:
        assert int(d["mailbox_timeout"]) == 0
        assert max(float(d["mailbox_halo_err%d" % rep]) for rep in range(3)) == 0.0
        np.testing.assert_array_equal(d["mailbox_hist"], ranks[0]["mailbox_hist"])       # same decision on every rank
        w = min(30, len(d["hist"]), len(d["mailbox_hist"]))
        np.testing.assert_allclose(d["mailbox_hist"][:w], d["hist"][:w], rtol=1e-9)
        assert abs(int(d["mailbox_it"]) - int(d["it"])) <= max(3, int(0.03 * int(d["it"])))
        if world == 2:                # two values: the sum does not depend on the order -- identical bits
            np.testing.assert_array_equal(d["mailbox_hist"], d["hist"])
            np.testing.assert_array_equal(d["mailbox_u"], d["u"])
    um = np.concatenate([d["mailbox_u"] for d in ranks])
    assert np.linalg.norm(um - ref["u"]) < 1e-5 * np.linalg.norm(ref["u"])
    # fused row-partitioned MINRES against the single-GPU fused MINRES
    mref = single_gpu_minres(s, pre, tol, maxsteps)
    for d in ranks:
        np.testing.assert_array_equal(d["minres_errors"], ranks[0]["minres_errors"])
        w = min(40, len(mref["errors"]), len(d["minres_errors"]))
        np.testing.assert_allclose(d["minres_errors"][:w], mref["errors"][:w], rtol=1e-8)
        assert abs(len(d["minres_errors"]) - len(mref["errors"])) <= max(3, int(0.03 * len(mref["errors"])))
        assert int(d["minres_rel"]) == 1
    um = np.concatenate([d["minres_u"] for d in ranks])
    assert np.linalg.norm(um - mref["u"]) < 1e-5 * np.linalg.norm(mref["u"])
    # ... and the native C loop over the mailbox transport (the first time nss_minres_iterate_dist runs with > 1 rank)
    for d in ranks:
        assert int(d["minres_mb_timeout"]) == 0 and int(d["minres_mb_rel"]) == 1
        np.testing.assert_array_equal(d["minres_mb_errors"], ranks[0]["minres_mb_errors"])
        w = min(40, len(d["minres_errors"]), len(d["minres_mb_errors"]))
        np.testing.assert_allclose(d["minres_mb_errors"][:w], d["minres_errors"][:w], rtol=1e-9)
        assert abs(len(d["minres_mb_errors"]) - len(d["minres_errors"])) <= max(3, int(0.03 * len(d["minres_errors"])))
        if world == 2:
            np.testing.assert_array_equal(d["minres_mb_errors"], d["minres_errors"])
    umm = np.concatenate([d["minres_mb_u"] for d in ranks])
    assert np.linalg.norm(umm - mref["u"]) < 1e-5 * np.linalg.norm(mref["u"])
    # fused row-partitioned BPCG v1 (behind bramble_pasciak_cg on distributed operands) against the single-GPU loop
    vref = single_gpu_bpcg1(s, pre, tol, maxsteps)
    for d in ranks:
        assert int(d["bpcg1_fused"]) == 1
        np.testing.assert_array_equal(d["bpcg1_errors"], ranks[0]["bpcg1_errors"])
        w = min(30, len(vref["errors"]), len(d["bpcg1_errors"]))
        np.testing.assert_allclose(d["bpcg1_errors"][:w], vref["errors"][:w], rtol=1e-8)
        assert abs(len(d["bpcg1_errors"]) - len(vref["errors"])) <= max(3, int(0.03 * len(vref["errors"])))
    u1 = np.concatenate([d["bpcg1_u"] for d in ranks])
    assert np.linalg.norm(u1 - vref["u"]) < 1e-5 * np.linalg.norm(vref["u"])
    for d in ranks:                       # nss_bpcg1_iterate_dist over the mailbox transport, > 1 rank
        assert int(d["bpcg1_mb_timeout"]) == 0
        np.testing.assert_array_equal(d["bpcg1_mb_errors"], ranks[0]["bpcg1_mb_errors"])
        w = min(30, len(d["bpcg1_errors"]), len(d["bpcg1_mb_errors"]))
        np.testing.assert_allclose(d["bpcg1_mb_errors"][:w], d["bpcg1_errors"][:w], rtol=1e-9)
        assert abs(len(d["bpcg1_mb_errors"]) - len(d["bpcg1_errors"])) <= max(3, int(0.03 * len(d["bpcg1_errors"])))
        if world == 2:
            np.testing.assert_array_equal(d["bpcg1_mb_errors"], d["bpcg1_errors"])
    u1m = np.concatenate([d["bpcg1_mb_u"] for d in ranks])
    assert np.linalg.norm(u1m - vref["u"]) < 1e-5 * np.linalg.norm(vref["u"])
    if pre == "bjac":
        # DistributedAMG (finest level on the slabs, coarse levels replicated on every rank) on the
        # product engine: same hierarchy, same V-cycle and same BPCG history as one GPU
        import hipla
        from solvers.bramblepasciak_new import BramblePasciakCG
        import re
        A, B = hipla.SparseMatrix.from_scipy(s.A), hipla.SparseMatrix.from_scipy(s.B)
        V = hipla.SmoothedAggregationAMG(A, coarse_size=60)
        assert list(ranks[0]["amg_levels"]) == V.level_sizes
        xa = np.random.default_rng(9).standard_normal(s.n_u)
        ya = hipla.Vector(s.n_u)
        ya.data = V * hipla.Vector.from_numpy(xa)
        got = np.concatenate([d["amg_apply"] for d in ranks])
        assert np.linalg.norm(got - ya.numpy()) < 1e-12 * np.linalg.norm(ya.numpy())

        class Form:
            def __init__(self, mat):
                self.mat, self.condense = mat, False

        f, g = s.rhs(0)
        sol = hipla.BlockVector([hipla.Vector(s.n_u), hipla.Vector(s.n_p)])
        out = io.StringIO()
        with contextlib.redirect_stdout(out):
            it_a, _ = BramblePasciakCG(Form(A), Form(B), None, hipla.Vector.from_numpy(f), hipla.Vector.from_numpy(g), V,
                                       hipla.DiagonalMatrix(1.0 / s.mass), sol, tol=tol, maxsteps=maxsteps)
        hist = np.array([float(m) for m in re.findall(r"it =\s+\d+\s+err =\s+(\S+)", out.getvalue())])
        w = min(25, len(hist), len(ranks[0]["amg_hist"]))
        np.testing.assert_allclose(ranks[0]["amg_hist"][:w], hist[:w], rtol=1e-8)
        assert abs(int(ranks[0]["amg_it"]) - it_a) <= max(3, int(0.03 * it_a))
        ua = np.concatenate([d["amg_u"] for d in ranks])
        assert np.linalg.norm(ua - sol[0].numpy()) < 1e-5 * np.linalg.norm(sol[0].numpy())


@pytest.mark.parametrize("world,dim,n", [(2, 3, 10), (3, 3, 9)])
def test_gauss_seidel_and_auxiliary_space_preconditioners_on_slabs(hip_engine, world, dim, n):
    """The reference's default velocity preconditioner on slabs (templates/NavierStokesSIMPLE_iterative.py:168,376-381):
    (i) the multicolour block Gauss-Seidel sweep inside the slab, additive across slabs, applied natively by the fused
    partitioned loop (no communication inside the preconditioner), and (ii) MypreA(GS=True) -- those sweeps around the
    auxiliary-space term on slabs (`DistributedAuxiliary`: transform and its transpose with one plane of halo, one
    V-cycle on the stacked nodal Laplacian with replicated coarse levels), BPCG v2 through the protocol -- against the
    SAME operators assembled in one process and run by the single-GPU fused loop."""
    import hipla
    from solvers.bramblepasciak_new import BramblePasciakCG
    from test_distributed_cpu import slab_twin_of_mypre_a
    tol, maxsteps = 1e-8, 3000
    ranks = launch(world, "gpu", dim, n, "bjac", tol, maxsteps)
    s = mac_stokes(dim, n, 0.01)
    f, g = s.rhs(0)
    twin, G, A, levels = slab_twin_of_mypre_a(s, world, s.line_blocks(3), coarse_size=40)
    B = hipla.SparseMatrix.from_scipy(s.B)

    class Form:
        def __init__(self, mat):
            self.mat, self.condense = mat, False

    def solve(preA):
        sol = hipla.BlockVector([hipla.Vector(s.n_u), hipla.Vector(s.n_p)])
        out = io.StringIO()
        with contextlib.redirect_stdout(out):
            it, _ = BramblePasciakCG(Form(A), Form(B), None, hipla.Vector.from_numpy(f), hipla.Vector.from_numpy(g), preA,
                                     hipla.DiagonalMatrix(1.0 / s.mass), sol, tol=tol, maxsteps=maxsteps)
        hist = np.array([float(m) for m in re.findall(r"it =\s+\d+\s+err =\s+(\S+)", out.getvalue())])
        return it, hist, sol[0].numpy()

    # (i) hybrid Gauss-Seidel in the fused partitioned loop == the symmetric sweep over the slab-block-diagonal part of A
    it_g, hist_g, u_g = solve(G)
    for d in ranks:
        assert int(d["bgs_colors"]) == G.ncolors
        np.testing.assert_array_equal(d["bgs_hist"], ranks[0]["bgs_hist"])
        w = min(25, len(hist_g), len(d["bgs_hist"]))
        np.testing.assert_allclose(d["bgs_hist"][:w], hist_g[:w], rtol=1e-8)
        assert abs(int(d["bgs_it"]) - it_g) <= max(3, int(0.03 * it_g))
    ug = np.concatenate([d["bgs_u"] for d in ranks])
    assert np.linalg.norm(ug - u_g) < 1e-5 * np.linalg.norm(u_g)
    # (ii) MypreA(GS=True) with the auxiliary-space term on slabs
    np.testing.assert_array_equal(ranks[0]["aux_levels"], levels)
    xa = np.random.default_rng(9).standard_normal(s.n_u)
    ya = hipla.Vector(s.n_u)
    twin.Mult(hipla.Vector.from_numpy(xa), ya)
    got = np.concatenate([d["mypre_apply"] for d in ranks])
    assert np.linalg.norm(got - ya.numpy()) < 1e-12 * np.linalg.norm(ya.numpy())
    it_m, hist_m, u_m = solve(twin)
    w = min(20, len(hist_m), len(ranks[0]["mypre_hist"]))
    np.testing.assert_allclose(ranks[0]["mypre_hist"][:w], hist_m[:w], rtol=1e-8)
    assert abs(int(ranks[0]["mypre_it"]) - it_m) <= max(3, int(0.05 * it_m))
    um = np.concatenate([d["mypre_u"] for d in ranks])
    assert np.linalg.norm(um - u_m) < 1e-5 * np.linalg.norm(u_m)


def test_native_partitioned_mypre_a_single_rank(hip_engine, tmp_path):
    """MypreA(GS=True) with the auxiliary-space term applied NATIVELY inside the partitioned BPCG loop (nss_dist_aux_*:
    the C loop issues the halo exchanges of the argument and of the nodal correction, those of the V-cycle and its
    coarse all-reduce, and the exchange of the iterate for the residual between the sweeps), on a 1-rank RCCL
    communicator: the native auxiliary apply equals the protocol `DistributedAuxiliary.Mult`, and the solve follows the
    single-GPU fused loop with the same operator (one slab: its Gauss-Seidel is the single-GPU sweep)."""
    import torch.distributed as dist
    import hipla
    from distributed import DistributedBpcg2
    from rccl_comm import RcclComm
    from solvers.bramblepasciak_new import BramblePasciakCG
    s = mac_stokes(3, 12, 0.01)
    f, g = s.rhs(0)
    tol, maxsteps = 1e-8, 2000

    class Form:
        def __init__(self, mat):
            self.mat, self.condense = mat, False

    dist.init_process_group("gloo", init_method="file://" + str(tmp_path / "rdv"), rank=0, world_size=1)
    try:
        comm = RcclComm(dist, hip_engine)
        run = DistributedBpcg2(s, f, g, s.line_blocks(3), dist, hip_engine, comm=comm, pre="mypre_a",
                               aux_options=dict(coarse_size=300))
        assert run.native is not None and run.compact and run.ops.aux is not None
        aux = run.ops.aux
        x = hipla.Vector.from_numpy(np.random.default_rng(3).standard_normal(s.n_u))
        y_host, y_native = hipla.Vector(s.n_u), hipla.Vector(s.n_u)
        y_host.data = aux * x
        aux.native_apply(1.0, x, y_native)
        assert np.linalg.norm(y_native.numpy() - y_host.numpy()) <= 1e-13 * np.linalg.norm(y_host.numpy())
        it, conv = run.solve(tol=tol, maxsteps=maxsteps, poll_every=16)
        assert conv
        # the same operator on one GPU: sweeps over A, one V-cycle on the stacked nodal Laplacian
        A, B = hipla.SparseMatrix.from_scipy(s.A), hipla.SparseMatrix.from_scipy(s.B)
        st = s.auxiliary_space_stacked()
        V = hipla.SmoothedAggregationAMG(hipla.SparseMatrix.from_scipy(st["laplacian"]), coarse_size=300)
        assert aux.level_sizes == V.level_sizes
        single = hipla.BlockGaussSeidel(A, s.line_blocks(3),
                                        middle=hipla.AuxiliarySpaceAMG(hipla.SparseMatrix.from_scipy(st["transform"]), [V]))
        sol = hipla.BlockVector([hipla.Vector(s.n_u), hipla.Vector(s.n_p)])
        out = io.StringIO()
        from hipla import fused
        runs, orig_run = [], fused.Bpcg2Loop.run
        fused.Bpcg2Loop.run = lambda self, *a, **k: (runs.append(1), orig_run(self, *a, **k))[1]
        try:
            with contextlib.redirect_stdout(out):
                it_s, _ = BramblePasciakCG(Form(A), Form(B), None, hipla.Vector.from_numpy(f), hipla.Vector.from_numpy(g),
                                           single, hipla.DiagonalMatrix(1.0 / s.mass), sol, tol=tol, maxsteps=maxsteps)
        finally:
            fused.Bpcg2Loop.run = orig_run
        assert runs == [1]                                                     # the single-GPU fused loop, natively
        hist = np.array([float(m) for m in re.findall(r"it =\s+\d+\s+err =\s+(\S+)", out.getvalue())])
        w = min(20, len(hist), it + 1)
        np.testing.assert_allclose(run.history(it)[:w], hist[:w], rtol=1e-8)
        assert abs(it - it_s) <= max(3, int(0.05 * it_s)) and it < 120
        assert np.linalg.norm(run.sol[0].numpy() - sol[0].numpy()) < 1e-5 * np.linalg.norm(sol[0].numpy())
        run.release()
        comm.close()
    finally:
        dist.destroy_process_group()


def test_rccl_ctypes_communicator_single_rank(hip_engine, tmp_path):
    """librccl binds through ctypes in the process that already holds torch's copy; a 1-rank
    communicator initialises, all-reduces in place and accepts an empty halo exchange.  (Two ranks
    on one device are refused by RCCL, so multi-rank traffic is covered by the gloo tests.)"""
    import torch
    import torch.distributed as dist
    from rccl_comm import RcclComm
    dist.init_process_group("gloo", init_method="file://" + str(tmp_path / "rdv"), rank=0, world_size=1)
    try:
        comm = RcclComm(dist, hip_engine)
        comm.size = 2                      # force the RCCL calls (1-rank communicator underneath)
        x = torch.arange(5, dtype=torch.float64, device=hip_engine.device)
        comm.allreduce_sum(x)
        torch.cuda.synchronize()
        assert x.tolist() == [0.0, 1.0, 2.0, 3.0, 4.0]
        comm.size = 1
        comm.self_test(torch)
        comm.close()
    finally:
        dist.destroy_process_group()


def test_native_partitioned_loop_split_and_streams_single_rank(hip_engine, tmp_path):
    """nss_bpcg2_iterate_dist on a 1-rank RCCL communicator: in-place all-reduces, and (mode 2) the
    interior / boundary split with the second stream and events.  Partial sums are per row block and
    added in a fixed order, so every variant must reproduce the single-GPU loop bit for bit."""
    import torch
    import torch.distributed as dist
    from distributed import DistributedBpcg2
    from rccl_comm import RcclComm
    dim, n, pre, tol, maxsteps = 3, 12, "bjac", 1e-8, 4000
    s, ref = single_gpu(dim, n, pre, tol, maxsteps)
    f, g = s.rhs(0)
    dist.init_process_group("gloo", init_method="file://" + str(tmp_path / "rdv"), rank=0, world_size=1)
    try:
        comm = RcclComm(dist, hip_engine)
        for overlap, interior in [("compact", None), (0, None), (1, None), (2, None), (2, "middle"), (2, "empty")]:
            plan = "compact" if overlap == "compact" else "classic"      # the overlap modes belong to the eight-phase plan
            overlap = 0 if overlap == "compact" else overlap
            run = DistributedBpcg2(s, f, g, s.line_blocks(3), dist, hip_engine, comm=comm, plan=plan)
            assert run.native is not None and run.compact == (plan == "compact")
            if interior is not None:
                nbs = {k: m.local.handle.info()["row_blocks"] for k, m in (("s1", run.ops.BT), ("t1", run.ops.A),
                                                                            ("t4", run.ops.B))}
                assert min(nbs.values()) >= 3
                if interior == "middle":      # prefix and suffix boundary blocks around an interior run
                    interior = {k: (max(1, nb // 4), nb - max(1, nb // 4)) for k, nb in nbs.items()}
                else:                         # everything is "boundary"
                    interior = {k: (nb // 2, nb // 2) for k, nb in nbs.items()}
                run.enable_native(comm.comm, interior)
            else:
                nb = run.ops.A.local.handle.info()["row_blocks"]
                assert run.native[1][1].int_begin == 0 and run.native[1][1].int_end == nb      # no ghosts: all interior
            run.overlap = overlap
            it, conv = run.solve(tol=tol, maxsteps=maxsteps, poll_every=16)
            assert conv and it == ref["it"]
            np.testing.assert_array_equal(run.history(it), ref["hist"])
            np.testing.assert_array_equal(run.sol[0].numpy(), ref["u"])
            np.testing.assert_array_equal(run.sol[1].numpy(), ref["p"])
        comm.close()
    finally:
        dist.destroy_process_group()


def test_native_loop_profile_and_frozen_scalars_single_rank(hip_engine, tmp_path):
    """(i) nss_dist_profile_*: the native partitioned loop reports the average device time of its eight
    phases (two of them the all-reduces, one the halo exchange) -- what tells which collective costs what
    on a real node.  (ii) Once the stop flag is set, further enqueued iterations leave the poll-visible
    scalars untouched: the local sums stay in their own slots and the all-reduce is out of place."""
    import torch
    import torch.distributed as dist
    from distributed import DistributedBpcg2
    from rccl_comm import RcclComm
    s = mac_stokes(3, 12, 0.01)
    f, g = s.rhs(0)
    dist.init_process_group("gloo", init_method="file://" + str(tmp_path / "rdv"), rank=0, world_size=1)
    try:
        comm = RcclComm(dist, hip_engine)
        for plan, native in (("compact", True), ("compact", False), ("classic", True), ("classic", False)):
            run = DistributedBpcg2(s, f, g, s.line_blocks(3), dist, hip_engine, comm=comm, native=native, plan=plan)
            assert (run.native is not None) == native
            run.start(tol=0.0, maxsteps=64)
            run.iterate(0, 8)
            phases, n = run.profile(8, 12)
            assert n == 12 and set(phases) == set(run.phase_names())
            busy = ("C23_A_B", "C4_sum") if plan == "compact" else ("K2_A", "K4_sum")
            assert all(v >= 0.0 for v in phases.values()) and all(phases[b] > 0.0 for b in busy)
            assert sum(phases.values()) < 50.0                                 # ms per iteration: sane
        for plan in ("compact", "classic"):
            run = DistributedBpcg2(s, f, g, s.line_blocks(3), dist, hip_engine, comm=comm, plan=plan)
            it, conv = run.solve(tol=1e-6, maxsteps=4000, poll_every=16)
            assert conv
            before = hip_engine.to_host(run.loop.scal).copy()
            hist_before = run.history(it).copy()
            run.iterate(it + 1, it + 20)                                       # all no-ops on the device
            done, it_final, _ = run.poll()
            assert done and it_final == it
            np.testing.assert_array_equal(hip_engine.to_host(run.loop.scal), before)
            np.testing.assert_array_equal(run.history(it), hist_before)
        comm.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pre", ["bjac", "jacobi"])
def test_native_partitioned_minres_single_rank(hip_engine, tmp_path, pre):
    """nss_minres_iterate_dist on a 1-rank RCCL communicator (grouped exchange of both operands, out-of-place
    all-reduces of delta and gamma_new^2 issued from C) and the host-driven schedule over nss_minres_phases:
    with the single-GPU loop held to the same launch form (sums in their stand-alone kernels, block Jacobi
    as its own launch) all three give identical bits; and the scalars stay frozen after the stop."""
    import torch
    import torch.distributed as dist
    from distributed import DistributedMinres
    from rccl_comm import RcclComm
    s = mac_stokes(3, 12, 0.01)
    f, g = s.rhs(0)
    tol, maxsteps = 1e-8, 4000
    lib = hip_engine.lib
    dist.init_process_group("gloo", init_method="file://" + str(tmp_path / "rdv"), rank=0, world_size=1)
    try:
        comm = RcclComm(dist, hip_engine)
        # fuse mode 0: rows of B^T in their own launch half; 2: inside the launch of A's rows (round 3; the slab's
        # fixed-width copy of B^T indexes z1's [owned | ghosts] layout) -- the block Jacobi apart in both
        for fuse, native in ((0, True), (0, False), (2, True)):
            assert lib.nss_minres_fold_mode(0) == 0 and lib.nss_minres_fuse_mode(fuse) == 0
            ref = single_gpu_minres(s, pre, tol, maxsteps)
            run = DistributedMinres(s, f, g, s.line_blocks(3) if pre == "bjac" else None, dist, hip_engine, comm=comm,
                                    native=native)
            assert (run.native is not None) == native
            u, errors, rel = run.solve(tol=tol, maxsteps=maxsteps, poll_every=16)
            assert rel
            np.testing.assert_array_equal(np.array(errors), ref["errors"])
            np.testing.assert_array_equal(u[0].numpy(), ref["u"])
            np.testing.assert_array_equal(u[1].numpy(), ref["p"])
            before = hip_engine.to_host(run.loop.scal).copy()
            run._iterate(len(errors), len(errors) + 9)                  # after the stop: no-ops on the device
            torch.cuda.synchronize()
            np.testing.assert_array_equal(hip_engine.to_host(run.loop.scal), before)
            np.testing.assert_array_equal(u[0].numpy(), ref["u"])
            run.close()
        comm.close()
    finally:
        lib.nss_minres_fold_mode(-1)
        lib.nss_minres_fuse_mode(-1)
        dist.destroy_process_group()


@pytest.mark.parametrize("pre", ["bjac", "jacobi"])
def test_native_partitioned_bpcg1_single_rank(hip_engine, tmp_path, pre):
    """nss_bpcg1_iterate_dist on a 1-rank RCCL communicator (grouped exchange of d, exchanges of t2_u and a_u,
    out-of-place all-reduces of <d, t1> and rho_new issued from C) and the host-driven schedule over
    nss_bpcg1_phases, both behind the reference's entry point called with distributed operands: identical bits
    to the single-GPU loop; the scalars stay frozen after the stop."""
    import torch
    import torch.distributed as dist
    import distributed
    from bramble_pasciak_cg import bramble_pasciak_cg
    from distributed import DistributedStokes
    from rccl_comm import RcclComm
    s = mac_stokes(3, 12, 0.01)
    f, g = s.rhs(0)
    tol, maxsteps = 1e-8, 4000
    dist.init_process_group("gloo", init_method="file://" + str(tmp_path / "rdv"), rank=0, world_size=1)
    loops = []
    orig = distributed.Bpcg1DistLoop.try_create.__func__

    def spy(cls, *a, **k):
        loops.append(orig(cls, *a, **k))
        return loops[-1]

    try:
        distributed.Bpcg1DistLoop.try_create = classmethod(spy)
        # (the single-GPU loop would plan A's row blocks around the Jacobi blocks and apply them in the epilogue of A's
        # rows: another partition of the partial sums of <d, t1>.  Held to the slab's launch plan for the bit comparison;
        # the merged rows of B^T stay on in both.)
        assert hip_engine.lib.nss_bpcg2_fuse_block_jacobi(0) == 0
        ref = single_gpu_bpcg1(s, pre, tol, maxsteps)
        comm = RcclComm(dist, hip_engine)
        ops = DistributedStokes(s, s.line_blocks(3) if pre == "bjac" else None, comm, hip_engine)
        for native in (True, False):
            distributed.Bpcg1DistLoop.NATIVE = native
            fv, gv = ops.vectors(f, g)
            with contextlib.redirect_stdout(io.StringIO()):
                x, errors = bramble_pasciak_cg(ops.A, ops.B, None, ops.preA, ops.preM, fv, gv, tolerance=tol,
                                               max_steps=maxsteps, print_rates=False)
            run = loops[-1]
            assert run is not None and (run.native is not None) == native
            np.testing.assert_array_equal(np.array(errors), ref["errors"])
            np.testing.assert_array_equal(x[0].numpy(), ref["u"])
            np.testing.assert_array_equal(x[1].numpy(), ref["p"])
            before = hip_engine.to_host(run.loop.scal).copy()
            run.enqueue(len(errors), len(errors) + 9)                   # after the stop: no-ops on the device
            torch.cuda.synchronize()
            np.testing.assert_array_equal(hip_engine.to_host(run.loop.scal), before)
            np.testing.assert_array_equal(x[0].numpy(), ref["u"])
            run.close()
        comm.close()
    finally:
        hip_engine.lib.nss_bpcg2_fuse_block_jacobi(-1)
        distributed.Bpcg1DistLoop.NATIVE = True
        distributed.Bpcg1DistLoop.try_create = classmethod(orig)
        dist.destroy_process_group()


def test_native_partitioned_amg_single_rank(hip_engine, tmp_path):
    """The V-cycle with replicated coarse levels inside the NATIVE partitioned BPCG loop (nss_dist_amg_*: the C
    loop issues the cycle's two halo exchanges and its coarse all-reduce itself), on a 1-rank RCCL communicator:
    the native apply equals the host-driven `DistributedAMG.Mult`, and the solve -- preA = AMG and the additive
    AMG + block Jacobi -- follows the single-GPU fused loop with the same hierarchy."""
    import re
    import torch.distributed as dist
    import hipla
    from distributed import DistributedBpcg2
    from rccl_comm import RcclComm
    from solvers.bramblepasciak_new import BramblePasciakCG
    s = mac_stokes(3, 12, 0.01)
    f, g = s.rhs(0)
    tol, maxsteps = 1e-8, 2000

    class Form:
        def __init__(self, mat):
            self.mat, self.condense = mat, False

    dist.init_process_group("gloo", init_method="file://" + str(tmp_path / "rdv"), rank=0, world_size=1)
    try:
        comm = RcclComm(dist, hip_engine)
        A, B = hipla.SparseMatrix.from_scipy(s.A), hipla.SparseMatrix.from_scipy(s.B)
        V = hipla.SmoothedAggregationAMG(A)
        for pre, single in (("amg", V), ("amg+bjac", V + hipla.BlockJacobi(A, s.line_blocks(3)))):
            run = DistributedBpcg2(s, f, g, s.line_blocks(3), dist, hip_engine, comm=comm, pre=pre)
            assert run.native is not None and run.dist_amg is not None
            assert run.dist_amg.level_sizes == V.level_sizes
            x = hipla.Vector.from_numpy(np.random.default_rng(3).standard_normal(s.n_u))
            y_host, y_native = hipla.Vector(s.n_u), hipla.Vector(s.n_u)
            y_host.data = run.dist_amg * x
            run.dist_amg.native_apply(1.0, x, y_native)
            assert np.linalg.norm(y_native.numpy() - y_host.numpy()) <= 1e-13 * np.linalg.norm(y_host.numpy())
            it, conv = run.solve(tol=tol, maxsteps=maxsteps, poll_every=16)
            assert conv
            sol = hipla.BlockVector([hipla.Vector(s.n_u), hipla.Vector(s.n_p)])
            out = io.StringIO()
            with contextlib.redirect_stdout(out):
                it_s, _ = BramblePasciakCG(Form(A), Form(B), None, hipla.Vector.from_numpy(f), hipla.Vector.from_numpy(g),
                                           single, hipla.DiagonalMatrix(1.0 / s.mass), sol, tol=tol, maxsteps=maxsteps)
            hist = np.array([float(m) for m in re.findall(r"it =\s+\d+\s+err =\s+(\S+)", out.getvalue())])
            w = min(25, len(hist), it + 1)
            np.testing.assert_allclose(run.history(it)[:w], hist[:w], rtol=1e-8)
            assert abs(it - it_s) <= max(3, int(0.03 * it_s))
            assert np.linalg.norm(run.sol[0].numpy() - sol[0].numpy()) < 1e-5 * np.linalg.norm(sol[0].numpy())
            run.close()
        comm.close()
    finally:
        dist.destroy_process_group()


def test_interior_row_blocks_of_a_partitioned_matrix(hip_engine):
    """Row blocks flagged interior must not reference ghost columns; boundary blocks sit at the
    slab ends (checked on rank 1 of 3 without any communication)."""
    from distributed import DistSparseMatrix
    s = mac_stokes(3, 20, 0.01)
    vel, prs = s.partition(3)

    class FakeComm:
        rank, size = 1, 3

        def gather_requests(self, mine, compute):
            return [compute(q) for q in range(3)]

    for mat, rows, cols in ((s.A, vel, vel), (s.B, prs, vel), (s.B.T.tocsr(), vel, prs)):
        dm = DistSparseMatrix(mat, rows, cols, FakeComm(), hip_engine)
        b0, b1 = dm.interior_row_blocks()
        rb = dm.local.handle.row_blocks()
        loc = dm.local_scipy
        assert 0 <= b0 < b1 <= rb.size - 1
        inner = loc[rb[b0]:rb[b1]]
        assert inner.indices.max() < dm.n_cols_owned                      # interior: owned columns only
        # the native halo descriptor sends one contiguous run per neighbour straight from the operand
        h = dm.native_halo(dm.operand())
        import ctypes
        assert h.direct == 1 and h.n_pack == 0 and 1 <= h.n_send <= 2 and 1 <= h.n_recv <= 2
        offs = np.ctypeslib.as_array(ctypes.cast(h.h_send_off, ctypes.POINTER(ctypes.c_int64)), shape=(h.n_send,))
        cnts = np.ctypeslib.as_array(ctypes.cast(h.h_send_cnt, ctypes.POINTER(ctypes.c_int64)), shape=(h.n_send,))
        assert np.all(offs >= 0) and np.all(offs + cnts <= dm.n_cols_owned)
        if h.n_send == 2:
            assert offs[0] == 0 and offs[1] + cnts[1] == dm.n_cols_owned  # first / last planes of the slab
        assert dm.plan.n_ghost > 0 and (b1 - b0) > 0.5 * (rb.size - 1)    # most of the slab is interior
        h = dm.native_halo(dm.operand())
        assert (h.int_begin, h.int_end) == (b0, b1) and h.n_recv >= 1 and h.n_send >= 1
