"""N > 1 path on CPU: world_size-2 (and 3) gloo process groups, numpy checker engine.
Covers the row partition, halo plans, all_to_all halo exchange, all-reduced inner products and
the protocol-path BPCG v2 / MINRES loops on distributed operands, against the single-rank run."""

import contextlib
import io
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from conftest import ROOT
from staggered_grid import mac_stokes


def launch(world, mode, dim, n, pre, tol, maxsteps):
    tmp = tempfile.mkdtemp(prefix="nssdist_")
    init = os.path.join(tmp, "rendezvous")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(r), str(world), mode,
                               init, tmp, str(dim), str(n), pre, repr(tol), str(maxsteps)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode(errors="replace"))
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [np.load(os.path.join(tmp, "rank%d.npz" % r)) for r in range(world)]


def slab_twin_of_mypre_a(s, world, blocks, **amg_options):
    """The single-process operator a `world`-way `DistributedStokes(pre="mypre_a")` applies: multicolour block
    Gauss-Seidel sweeps over the slab-block-diagonal part of A (inside every slab, additive across slabs), the residual
    between them with the full A, and `transform @ V(L) @ transform.T` with one V-cycle on the stacked nodal Laplacian
    (`StokesSystem.auxiliary_space_stacked`).  Returns (MypreA twin, sweep-only twin, A, level sizes)."""
    import hipla
    import scipy.sparse as sp
    vel, _ = s.partition(world)
    A = hipla.SparseMatrix.from_scipy(s.A)
    bd = sp.block_diag([s.A[vel[r]:vel[r + 1], vel[r]:vel[r + 1]] for r in range(world)], format="csr")
    G = hipla.BlockGaussSeidel(hipla.SparseMatrix.from_scipy(bd), blocks)
    st = s.auxiliary_space_stacked()
    V = hipla.SmoothedAggregationAMG(hipla.SparseMatrix.from_scipy(st["laplacian"]), **amg_options)
    aux = hipla.AuxiliarySpaceAMG(hipla.SparseMatrix.from_scipy(st["transform"]), [V])

    class Twin(hipla.BaseMatrix):
        def Height(self):
            return s.n_u

        def Width(self):
            return s.n_u

        def CreateColVector(self):
            return hipla.Vector(s.n_u)

        CreateRowVector = CreateColVector

        def Mult(self, x, y):                      # templates/NavierStokesSIMPLE_iterative.py:377-381
            y[:] = 0.0
            G.Smooth(y, x)
            res = x.CreateVector()
            res.data = x - A * y
            y.data += aux * res
            G.SmoothBack(y, x)

        MultTrans = Mult

        @property
        def T(self):
            return self

    return Twin(), G, A, V.level_sizes


def single_rank_reference(dim, n, pre, tol, maxsteps, numpy_engine):
    import hipla
    from minres import MinRes
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
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        ses = BpcgSession(Form(A), Form(B), None, hipla.Vector.from_numpy(f), hipla.Vector.from_numpy(g), preA, preM,
                          sol=sol)
        it, conv = ses.protocol_loop(tol, maxsteps, True, True)
    hist = np.array([float(m) for m in re.findall(r"it =\s+\d+\s+err =\s+(\S+)", out.getvalue())])
    K = hipla.BlockMatrix([[A, B.T], [B, None]])
    Cm = hipla.BlockMatrix([[preA, None], [None, preM]])
    with contextlib.redirect_stdout(io.StringIO()):
        um, errs = MinRes(mat=K, pre=Cm, rhs=hipla.BlockVector([hipla.Vector.from_numpy(f), hipla.Vector.from_numpy(g)]),
                          maxsteps=maxsteps, tol=tol, printrates=False)
    from bramble_pasciak_cg import bramble_pasciak_cg
    with contextlib.redirect_stdout(io.StringIO()):
        x1, errs1 = bramble_pasciak_cg(A, B, None, preA, preM, hipla.Vector.from_numpy(f), hipla.Vector.from_numpy(g),
                                       tolerance=tol, max_steps=maxsteps, print_rates=False)
    amg_ref = {}
    if pre == "bjac":                                  # single-process AMG solve: what DistributedAMG must reproduce
        from solvers.bramblepasciak_new import BramblePasciakCG
        V = hipla.SmoothedAggregationAMG(A, coarse_size=60)
        xa = np.random.default_rng(9).standard_normal(s.n_u)
        ya = hipla.Vector(s.n_u)
        ya.data = V * hipla.Vector.from_numpy(xa)
        sol_a = hipla.BlockVector([hipla.Vector(s.n_u), hipla.Vector(s.n_p)])
        out = io.StringIO()
        with contextlib.redirect_stdout(out):
            it_a, _ = BramblePasciakCG(Form(A), Form(B), None, hipla.Vector.from_numpy(f), hipla.Vector.from_numpy(g), V,
                                       preM, sol_a, tol=tol, maxsteps=maxsteps)
        amg_ref = dict(amg_levels=np.array(V.level_sizes), amg_apply=ya.numpy(), amg_it=it_a, amg_u=sol_a[0].numpy(),
                       amg_hist=np.array([float(m) for m in re.findall(r"it =\s+\d+\s+err =\s+(\S+)", out.getvalue())]))
    return s, dict(it=it, hist=hist, k=ses.k, err0=ses.err0, u=sol[0].numpy(), p=sol[1].numpy(),
                   minres_errors=np.array(errs), minres_u=um[0].numpy(), bpcg1_errors=np.array(errs1),
                   bpcg1_u=x1[0].numpy(), **amg_ref)


@pytest.mark.parametrize("world,dim,n,pre", [(2, 3, 8, "bjac"), (3, 2, 20, "jacobi")])
def test_gloo_row_partitioned_solve_matches_single_rank(numpy_engine, world, dim, n, pre):
    tol, maxsteps = 1e-8, 3000
    ranks = launch(world, "cpu", dim, n, pre, tol, maxsteps)
    s, ref = single_rank_reference(dim, n, pre, tol, maxsteps, numpy_engine)
    vel, prs = s.partition(world)
    for r, d in enumerate(ranks):
        assert list(d["slices"]) == [vel[r], vel[r + 1], prs[r], prs[r + 1]]
        assert d["err_AxBTp"] < 1e-12 and d["err_Bx"] < 1e-12
        assert abs(d["dot"] - d["dot_ref"]) < 1e-10 * d["dot_ref"]
        # neighbour halos only: a few grid planes (A's and B's operands also carry the velocity dofs
        # around the ghost pressure cells, for the redundant ghost updates of the fused loop)
        plane_u = s.velocity_slab_offsets[1] - s.velocity_slab_offsets[0]
        assert list(d["direct"]) == [1, 1, 1]           # every neighbour is served by one contiguous run
        assert 0 < d["halo"][0] <= 4 * plane_u
        assert 0 <= d["halo"][1] <= 4 * plane_u and 0 <= d["halo"][2] <= 2 * s.n ** (dim - 1)
        # scalars are identical on every rank (all-reduced), histories match the single-rank run
        assert abs(d["k"] - ref["k"]) < 1e-9 * ref["k"]
        assert abs(d["err0"] - ref["err0"]) < 1e-10 * ref["err0"]
        np.testing.assert_array_equal(d["hist"], ranks[0]["hist"])
        w = min(30, len(ref["hist"]), len(d["hist"]))   # inside the stable window
        np.testing.assert_allclose(d["hist"][:w], ref["hist"][:w], rtol=1e-8)
        assert abs(int(d["it"]) - ref["it"]) <= max(3, int(0.03 * ref["it"]))
        np.testing.assert_allclose(d["minres_errors"][:40], ref["minres_errors"][:40], rtol=1e-8)
        assert abs(len(d["minres_errors"]) - len(ref["minres_errors"])) <= max(3, int(0.03 * len(ref["minres_errors"])))
        # BPCG v1 through the protocol on slabs that know their communicator (no solver changes)
        np.testing.assert_allclose(d["bpcg1_errors"][:30], ref["bpcg1_errors"][:30], rtol=1e-8)
        assert abs(len(d["bpcg1_errors"]) - len(ref["bpcg1_errors"])) <= max(3, int(0.03 * len(ref["bpcg1_errors"])))
    assert sum(int(d["halo"][1]) for d in ranks) > 0 and sum(int(d["halo"][2]) for d in ranks) > 0
    if pre == "bjac":      # distributed AMG with replicated coarse levels == the single-process V-cycle
        np.testing.assert_array_equal(ranks[0]["amg_levels"], ref["amg_levels"])
        ya = np.concatenate([d["amg_apply"] for d in ranks])
        assert np.linalg.norm(ya - ref["amg_apply"]) < 1e-12 * np.linalg.norm(ref["amg_apply"])
        w = min(25, len(ref["amg_hist"]), len(ranks[0]["amg_hist"]))
        np.testing.assert_allclose(ranks[0]["amg_hist"][:w], ref["amg_hist"][:w], rtol=1e-8)
        assert abs(int(ranks[0]["amg_it"]) - ref["amg_it"]) <= max(3, int(0.03 * ref["amg_it"]))
        ua = np.concatenate([d["amg_u"] for d in ranks])
        assert np.linalg.norm(ua - ref["amg_u"]) < 1e-5 * np.linalg.norm(ref["amg_u"])
        assert ref["amg_it"] < ref["it"] / 2                # and it pays: far fewer iterations than block Jacobi
    u1 = np.concatenate([d["bpcg1_u"] for d in ranks])
    assert np.linalg.norm(u1 - ref["bpcg1_u"]) < 1e-5 * np.linalg.norm(ref["bpcg1_u"])
    u = np.concatenate([d["u"] for d in ranks])
    p = np.concatenate([d["p"] for d in ranks])
    assert np.linalg.norm(u - ref["u"]) < 1e-5 * np.linalg.norm(ref["u"])
    p0, pr = p - p.mean(), ref["p"] - ref["p"].mean()
    assert np.linalg.norm(p0 - pr) < 1e-4 * np.linalg.norm(pr)
    um = np.concatenate([d["minres_u"] for d in ranks])
    assert np.linalg.norm(um - ref["minres_u"]) < 1e-5 * np.linalg.norm(ref["minres_u"])
    if pre == "bjac":
        # the reference's default preA on slabs (MypreA with GS=True: sweeps inside the slab around the auxiliary-space
        # term on slabs) == the same operator assembled in one process; BPCG v2 with it needs far fewer iterations
        import hipla
        from solvers.bramblepasciak_new import BramblePasciakCG
        twin, _, A, levels = slab_twin_of_mypre_a(s, world, s.line_blocks(3), coarse_size=40)
        np.testing.assert_array_equal(ranks[0]["aux_levels"], levels)
        xa = np.random.default_rng(9).standard_normal(s.n_u)
        ya = hipla.Vector(s.n_u)
        twin.Mult(hipla.Vector.from_numpy(xa), ya)
        got = np.concatenate([d["mypre_apply"] for d in ranks])
        assert np.linalg.norm(got - ya.numpy()) < 1e-12 * np.linalg.norm(ya.numpy())

        class Form:
            def __init__(self, mat):
                self.mat, self.condense = mat, False

        f, g = s.rhs(0)
        B = hipla.SparseMatrix.from_scipy(s.B)
        sol = hipla.BlockVector([hipla.Vector(s.n_u), hipla.Vector(s.n_p)])
        out = io.StringIO()
        with contextlib.redirect_stdout(out):
            it_m, _ = BramblePasciakCG(Form(A), Form(B), None, hipla.Vector.from_numpy(f), hipla.Vector.from_numpy(g), twin,
                                       hipla.DiagonalMatrix(1.0 / s.mass), sol, tol=tol, maxsteps=maxsteps)
        hist = np.array([float(m) for m in re.findall(r"it =\s+\d+\s+err =\s+(\S+)", out.getvalue())])
        w = min(20, len(hist), len(ranks[0]["mypre_hist"]))
        np.testing.assert_allclose(ranks[0]["mypre_hist"][:w], hist[:w], rtol=1e-8)
        assert abs(int(ranks[0]["mypre_it"]) - it_m) <= max(3, int(0.05 * it_m))
        um = np.concatenate([d["mypre_u"] for d in ranks])
        assert np.linalg.norm(um - sol[0].numpy()) < 1e-5 * np.linalg.norm(sol[0].numpy())
        assert it_m < ref["it"] / 2


def test_halo_plan_bookkeeping():
    from distributed import HaloPlan, even_offsets, localize_rows
    s = mac_stokes(3, 6)
    vel, prs = s.partition(3)
    total_recv = total_send = 0
    plans = []
    for r in range(3):
        loc, ghosts = localize_rows(s.A, (vel[r], vel[r + 1]), vel, r)
        assert loc.shape == (vel[r + 1] - vel[r], vel[r + 1] - vel[r] + ghosts.size)
        assert np.all((ghosts < vel[r]) | (ghosts >= vel[r + 1]))
        plans.append(HaloPlan(r, 3, vel[r + 1] - vel[r], ghosts, vel))
    for r, pl in enumerate(plans):
        pl.finalize([plans[q].requests()[r] for q in range(3)])
        assert pl.recv_counts[r] == 0 and pl.send_counts[r] == 0
        total_recv += pl.recv_counts.sum()
        total_send += pl.send_counts.sum()
        assert pl.send_idx.size == pl.send_counts.sum()
        assert pl.send_idx.min(initial=0) >= 0 and pl.send_idx.max(initial=0) < pl.n_owned
    assert total_recv == total_send > 0
    assert plans[0].recv_counts[2] == 0 and plans[2].recv_counts[0] == 0      # slabs: neighbours only
    # the renumbered local product equals the global one
    x = np.random.default_rng(0).standard_normal(s.n_u)
    for r in range(3):
        loc, ghosts = localize_rows(s.A, (vel[r], vel[r + 1]), vel, r)
        ext = np.concatenate([x[vel[r]:vel[r + 1]], x[ghosts]])
        np.testing.assert_allclose(loc @ ext, (s.A @ x)[vel[r]:vel[r + 1]], rtol=1e-14, atol=1e-14)
    assert list(even_offsets(10, 3)) == [0, 3, 7, 10]


def test_halo_plans_of_slab_partitions_send_contiguous_runs(numpy_engine):
    """Slab neighbours want (most of) one or two grid planes: after `densify_ghosts` every
    destination is served by ONE contiguous run of the owned entries, so the native loop sends
    straight out of the operand (nss_halo_t.direct) and needs no pack kernel.  Checked for every
    rank of 2-, 3- and 8-way partitions without communication."""
    from distributed import DistSparseMatrix, densify_ghosts
    from staggered_grid import mac_stokes
    s = mac_stokes(3, 16, 0.01)
    for nranks in (2, 3, 8):
        vel, prs = s.partition(nranks)
        for rank in range(nranks):
            class FakeComm:
                size = nranks

                def gather_requests(self, mine, compute):
                    return [compute(q) for q in range(nranks)]

            FakeComm.rank = rank
            for mat, rows, cols in ((s.A, vel, vel), (s.B, prs, vel), (s.B.T.tocsr(), vel, prs)):
                dm = DistSparseMatrix(mat, rows, cols, FakeComm(), numpy_engine)
                plan = dm.plan
                assert plan.direct
                ext = np.arange(plan.n_owned + plan.n_ghost, dtype=np.float64)
                at = 0
                for q in range(nranks):
                    c = int(plan.send_counts[q])
                    if c:
                        run = plan.send_runs[q]
                        assert 0 <= run and run + c <= plan.n_owned
                        np.testing.assert_array_equal(ext[run:run + c], ext[plan.send_idx[at:at + c]])
                    at += c
                # the ghost tail is still sorted by (owner, id) and covers what the rows reference
                assert np.all(np.diff(plan.ghosts) > 0)
                assert dm.local_scipy.indices.max() < plan.n_owned + plan.n_ghost
    # a scattered request stays a list (no densification beyond 2x)
    g = densify_ghosts(np.array([100, 150, 199]), np.array([0, 100, 200]))
    np.testing.assert_array_equal(g, [100, 150, 199])
    g = densify_ghosts(np.array([100, 101, 103]), np.array([0, 100, 200]))
    np.testing.assert_array_equal(g, [100, 101, 102, 103])


def test_rccl_direct_exchange_issues_the_right_segments():
    """`RcclComm.exchange_direct` (multi-rank RCCL cannot run on the development box): with a recording
    stand-in for librccl, the sends start at the contiguous runs of the operand and the receives fill
    the ghost tail owner by owner."""
    import types
    from rccl_comm import RcclComm
    calls = []

    class FakeLib:
        def ncclGroupStart(self):
            calls.append(("start",))
            return 0

        def ncclGroupEnd(self):
            calls.append(("end",))
            return 0

        def ncclSend(self, ptr, cnt, dtype, peer, comm, st):
            calls.append(("send", ptr, cnt, peer))
            return 0

        def ncclRecv(self, ptr, cnt, dtype, peer, comm, st):
            calls.append(("recv", ptr, cnt, peer))
            return 0

    comm = object.__new__(RcclComm)
    comm.size, comm.rank, comm._plans, comm.lib, comm.comm = 3, 1, {}, FakeLib(), 1234
    comm.engine = types.SimpleNamespace(stream=0)
    plan = types.SimpleNamespace(n_owned=100, send_counts=np.array([7, 0, 5]), recv_counts=np.array([4, 0, 6]),
                                 send_runs={0: 0, 2: 95}, direct=True)
    ext = types.SimpleNamespace(data_ptr=lambda: 1 << 20)
    comm.exchange_direct(plan, ext)
    base = 1 << 20
    assert calls == [("start",), ("send", base, 7, 0), ("send", base + 8 * 95, 5, 2),
                     ("recv", base + 8 * 100, 4, 0), ("recv", base + 8 * 104, 6, 2), ("end",)]
    calls.clear()
    comm.exchange_direct(plan, ext)                      # cached plan: same calls
    assert len(calls) == 6
    # packed variant: segments of the send buffer in destination order
    calls.clear()
    sendbuf = types.SimpleNamespace(data_ptr=lambda: 1 << 24)
    comm.exchange(plan, sendbuf, ext)
    assert calls == [("start",), ("send", 1 << 24, 7, 0), ("send", (1 << 24) + 8 * 7, 5, 2),
                     ("recv", base + 8 * 100, 4, 0), ("recv", base + 8 * 104, 6, 2), ("end",)]


def _simulated_ranks(s, nranks, engine, blocks=None):
    """`DistributedStokes` of every rank of an `nranks`-way partition, built in this process.  Two passes: the
    first records what every rank would contribute to the set-up all-gather of halo requests (call by call: B,
    A, B^T -- including the extra ghosts A's and B's operands receive for the redundant ghost updates), the
    second hands every rank the gathered lists, as `TorchComm.gather_requests` would."""
    from distributed import DistributedStokes
    recorded = [[] for _ in range(nranks)]

    def build(rank, replay):
        class FakeComm:
            size = nranks

            def __init__(self):
                self.calls = 0

            def gather_requests(self, mine, compute):
                k, self.calls = self.calls, self.calls + 1
                if not replay:
                    recorded[rank].append(mine)
                    return [compute(q) for q in range(nranks)]
                return [recorded[q][k] for q in range(nranks)]

        FakeComm.rank = rank
        return DistributedStokes(s, blocks, FakeComm(), engine)

    for rank in range(nranks):
        build(rank, False)
    return [build(rank, True) for rank in range(nranks)]


def _replay_native_exchange(mats, ext, x_global):
    """Move data exactly as csrc/dist.hip::exchange would from the `nss_halo_t` descriptors of every rank
    (`mats[r]`: the rank's DistSparseMatrix, `ext[r]`: its operand buffer [owned | ghosts]): per rank the sends in
    descriptor order out of ext (direct) or out of the packed buffer, matched to the receiver's descriptor entry
    of that peer.  Returns nothing; fills the ghost tails of `ext`."""
    import ctypes as C

    def arr(ptr, n, ct):
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n,)).copy() if n else np.zeros(0, dtype=np.int64)

    desc = []
    for r, m in enumerate(mats):
        hv = m.operand()
        hv.ext[:] = ext[r]
        h = m.native_halo(hv, interior=(0, 0))
        desc.append(dict(direct=h.direct, n_pack=h.n_pack,
                         sp=arr(h.h_send_peer, h.n_send, C.c_int32), so=arr(h.h_send_off, h.n_send, C.c_int64),
                         sc=arr(h.h_send_cnt, h.n_send, C.c_int64), rp=arr(h.h_recv_peer, h.n_recv, C.c_int32),
                         ro=arr(h.h_recv_off, h.n_recv, C.c_int64), rc=arr(h.h_recv_cnt, h.n_recv, C.c_int64),
                         keep=h))
    wire = {}
    for r, (m, d) in enumerate(zip(mats, desc)):
        src = ext[r] if d["direct"] else ext[r][m.plan.send_idx]        # pack kernel: sendbuf[i] = ext[send_idx[i]]
        assert d["direct"] or d["n_pack"] == m.plan.send_idx.size
        for peer, off, cnt in zip(d["sp"], d["so"], d["sc"]):
            assert (r, int(peer)) not in wire                             # one message per ordered pair
            wire[(r, int(peer))] = src[int(off): int(off) + int(cnt)].copy()
    for r, (m, d) in enumerate(zip(mats, desc)):
        for peer, off, cnt in zip(d["rp"], d["ro"], d["rc"]):
            msg = wire.pop((int(peer), r))
            assert msg.size == int(cnt) and int(off) >= m.plan.n_owned
            ext[r][int(off): int(off) + int(cnt)] = msg
    assert not wire                                                       # every send has its receive


@pytest.mark.parametrize("nranks", [2, 3, 8])
def test_native_halo_descriptors_move_the_right_entries(numpy_engine, nranks):
    """What the C loops hand to RCCL (`nss_halo_t`: send_off / cnt / peer, recv_off into the ext layout) for
    simulated ranks 0..R-1, replayed in numpy: afterwards every ghost tail holds the owner's entries -- for A's,
    B's and B^T's operands -- so that a multi-GPU run only has to confirm transport, not indexing.  Also the
    renumbered matrices the native loops multiply those buffers with: B in the layout of A's operand (MINRES,
    BPCG v1) and B with the ghost pressure rows behind its own (compact BPCG v2 plan)."""
    from distributed import b_in_layout_of_a
    s = mac_stokes(3, 16, 0.01)
    ranks = _simulated_ranks(s, nranks, numpy_engine)
    rng = np.random.default_rng(11)
    xu, xp = rng.standard_normal(s.n_u), rng.standard_normal(s.n_p)
    vel, prs = ranks[0].vel, ranks[0].prs
    for name, xg, offs in (("A", xu, vel), ("B", xu, vel), ("BT", xp, prs)):
        mats = [getattr(o, name) for o in ranks]
        ext = []
        for r, m in enumerate(mats):
            e = np.full(m.plan.n_owned + m.plan.n_ghost, np.nan)
            e[: m.plan.n_owned] = xg[offs[r]: offs[r + 1]]
            ext.append(e)
        _replay_native_exchange(mats, ext, xg)
        for r, m in enumerate(mats):
            np.testing.assert_array_equal(ext[r][m.plan.n_owned:], xg[m.plan.ghosts])
            # ... and the local block times the filled buffer is the global product on the slab
            rows = slice(int(m.row_offsets[r]), int(m.row_offsets[r + 1]))
            glob = {"A": s.A, "B": s.B, "BT": s.B.T.tocsr()}[name]
            np.testing.assert_allclose(m.local_scipy @ ext[r], (glob @ xg)[rows], rtol=1e-13, atol=1e-13)
        if name == "A":
            ext_a = ext
    for r, o in enumerate(ranks):
        # B's slab with its columns renumbered into A's operand layout multiplies A's buffer
        b_on_a = b_in_layout_of_a(o.A, o.B, numpy_engine).to_scipy()
        np.testing.assert_allclose(b_on_a @ ext_a[r], (s.B @ xu)[prs[r]: prs[r + 1]], rtol=1e-13, atol=1e-13)
        # compact plan: [owned pressure rows | ghost pressure rows], same layout, same entry order per row
        assert o.compact_layout_ok()
        b_ext = o.b_extended_scipy()
        assert b_ext.shape == (o.n_p + o.BT.plan.n_ghost, o.A.plan.n_owned + o.A.plan.n_ghost)
        want = (s.B @ xu)[np.concatenate([np.arange(prs[r], prs[r + 1]), o.BT.plan.ghosts])]
        np.testing.assert_allclose(b_ext @ ext_a[r], want, rtol=1e-13, atol=1e-13)
        own = b_ext[: o.n_p]
        np.testing.assert_array_equal(own.indptr, o.B.local_scipy.indptr)
        np.testing.assert_array_equal(own.data, o.B.local_scipy.data)      # owned/ghost order is kept: same row sums bit for bit
        np.testing.assert_array_equal(own.indices, b_on_a.indices)
