"""Worker of the multi-process tests (world_size ranks, spawned by test_distributed*.py).

mode "cpu": gloo backend, numpy checker engine -> exercises partition / halo / all-reduce host
logic and the protocol-path BPCG / MINRES loops on distributed operands.
mode "gpu": gloo backend with host staging, HIP engine, all ranks on the one visible GPU ->
exercises the fused distributed loop (nss_bpcg2_phase + halo exchanges + all-reduces)."""

import contextlib
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "navier-stokes-solver_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def run(rank, world, mode, init_file, out_dir, dim, n, pre, tol, maxsteps):
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    import hipla
    if mode == "cpu":
        from oracle.numpy_engine import NumpyEngine
        hipla.set_engine(NumpyEngine())
    else:
        torch.cuda.set_device(0)
        hipla.set_engine(None)
    eng = hipla.get_engine()
    from distributed import DistributedBpcg2, DistributedStokes, Form, TorchComm
    from minres import MinRes
    from solvers.bramblepasciak_new import BpcgSession
    from staggered_grid import mac_stokes

    sysm = mac_stokes(dim, n, 0.01)
    f, g = sysm.rhs(0)
    blocks = sysm.line_blocks(3) if pre == "bjac" else None
    comm = TorchComm(dist, eng)
    res = {}

    ops = DistributedStokes(sysm, blocks, comm, eng)
    us, ps = ops.local_slices()
    res["slices"] = np.array([us.start, us.stop, ps.start, ps.stop])
    res["halo"] = np.array([ops.A.plan.n_ghost, ops.B.plan.n_ghost, ops.BT.plan.n_ghost])
    res["direct"] = np.array([int(ops.A.plan.direct), int(ops.B.plan.direct), int(ops.BT.plan.direct)])
    # distributed SpMV against the global product
    rng = np.random.default_rng(5)
    xu, xp = rng.standard_normal(sysm.n_u), rng.standard_normal(sysm.n_p)
    vu, vp = hipla.Vector.from_numpy(xu[us]), hipla.Vector.from_numpy(xp[ps])
    y = hipla.Vector(ops.n_u)
    y.data = ops.A * vu + ops.B.T * vp
    res["err_AxBTp"] = np.max(np.abs(y.numpy() - (sysm.A @ xu + sysm.B.T @ xp)[us]))
    q = hipla.Vector(ops.n_p)
    q.data = ops.B * vu
    res["err_Bx"] = np.max(np.abs(q.numpy() - (sysm.B @ xu)[ps]))
    res["dot"] = ops.inner(vu, vu)
    res["dot_ref"] = float(np.dot(xu, xu))

    # ---- BPCG v2 on distributed operands -------------------------------------------------------
    sink = io.StringIO()
    if mode == "cpu":
        sol = hipla.BlockVector([hipla.Vector(ops.n_u), hipla.Vector(ops.n_p)])
        with contextlib.redirect_stdout(sink):
            ses = BpcgSession(Form(ops.A), Form(ops.B), None, hipla.Vector.from_numpy(f[us]),
                              hipla.Vector.from_numpy(g[ps]), ops.preA, ops.preM, sol=sol, inner=ops.inner)
            it, conv = ses.protocol_loop(tol, maxsteps, True, True)
        import re
        res["hist"] = np.array([float(m) for m in re.findall(r"it =\s+\d+\s+err =\s+(\S+)", sink.getvalue())])
        res["it"], res["k"], res["err0"] = it, ses.k, ses.err0
        res["u"], res["p"] = sol[0].numpy(), sol[1].numpy()
        # ---- MINRES on distributed operands ---------------------------------------------------------
        K = hipla.BlockMatrix([[ops.A, ops.B.T], [ops.B, None]])
        Cm = hipla.BlockMatrix([[ops.preA, None], [None, ops.preM]])
        fv, gv = ops.vectors(f, g)                  # slabs that know the communicator: global inner products
        with contextlib.redirect_stdout(io.StringIO()):
            um, errs = MinRes(mat=K, pre=Cm, rhs=hipla.BlockVector([fv, gv]), maxsteps=maxsteps, tol=tol,
                              printrates=False)
        res["minres_errors"] = np.array(errs)
        res["minres_u"] = um[0].numpy()
        # ---- BPCG v1 on distributed operands (scale factor from the distributed Lanczos) --------------
        from bramble_pasciak_cg import bramble_pasciak_cg
        fv, gv = ops.vectors(f, g)
        with contextlib.redirect_stdout(io.StringIO()):
            x1, errs1 = bramble_pasciak_cg(ops.A, ops.B, None, ops.preA, ops.preM, fv, gv, tolerance=tol,
                                           max_steps=maxsteps, print_rates=False)
        res["bpcg1_errors"] = np.array(errs1)
        res["bpcg1_u"] = x1[0].numpy()
    else:
        run_ = DistributedBpcg2(sysm, f, g, blocks, dist, eng, comm=comm)          # compact plan (default)
        it, conv = run_.solve(tol=tol, maxsteps=maxsteps, poll_every=8)
        res["hist"] = run_.history(it)
        res["compact"] = int(bool(run_.compact))
        # the same solve on the eight-phase plan: per lane the same arithmetic, so -- where B's launch plan is the
        # same with and without the ghost pressure rows behind it -- the same bits
        classic = DistributedBpcg2(sysm, f, g, blocks, dist, eng, comm=comm, plan="classic")
        it_c, _ = classic.solve(tol=tol, maxsteps=maxsteps, poll_every=5)
        res["hist_classic"], res["it_classic"] = classic.history(it_c), it_c
        res["u_classic"], res["p_classic"] = classic.sol[0].numpy(), classic.sol[1].numpy()
        rb_ext = run_.ops.b_extended().handle.row_blocks()
        rb_loc = run_.ops.B.local.handle.row_blocks()
        res["same_plan"] = int(np.array_equal(rb_ext[: rb_loc.size], rb_loc))
        res["launch_forms"] = np.array([run_.ops.b_extended().handle.info()["lanes_per_row"],
                                        run_.ops.B.local.handle.info()["lanes_per_row"]])
        res["ghost_mode"] = int(bool(run_.ghost_mode)) + int(bool(getattr(run_, "ghost_p_mode", False)))
        res["it"], res["k"], res["err0"] = it, run_.k, run_.err0
        res["u"], res["p"] = run_.sol[0].numpy(), run_.sol[1].numpy()
        # ---- the mailbox transport (csrc/p2p.h): peer-mapped mailboxes / landing zones, no collective library --------
        #      stand-alone: an all-reduce of rank-dependent values and the halo exchange of A's operand ...
        from distributed import MailboxTransport
        import torch as _torch
        probe = ops.A.operand()
        mb = MailboxTransport(comm, eng, [(ops.A.native_halo(probe, (0, 0)), ops.n_u)])
        vals = []
        for rep in range(6):
            src = _torch.tensor([np.sin(1.0 + rank + 10.0 * rep) * 1e3], dtype=_torch.float64, device=eng.device)
            dst = _torch.zeros(1, dtype=_torch.float64, device=eng.device)
            mb.allreduce(src, dst)
            vals.append(float(dst.cpu()[0]))
        res["mailbox_allreduce"] = np.array(vals)
        xg = np.random.default_rng(21).standard_normal(sysm.n_u)
        for rep in range(3):
            probe.set_from((rep + 1.0) * xg[us])
            mb.exchange()
            _torch.cuda.synchronize()
            got = eng.to_host(probe.ext)[ops.n_u:]
            res["mailbox_halo_err%d" % rep] = float(np.max(np.abs(got - (rep + 1.0) * xg[ops.A.plan.ghosts]))) if got.size else 0.0
        res["mailbox_timeout"] = int(mb.timed_out())
        mb.close()
        #      ... and the native compact loop over it: no RCCL / gloo call inside an iteration
        mrun = DistributedBpcg2(sysm, f, g, blocks, dist, eng, comm=comm, transport="mailbox")
        assert mrun.mailbox is not None and mrun.native is not None
        it_mb, _ = mrun.solve(tol=tol, maxsteps=maxsteps, poll_every=16)
        res["mailbox_hist"], res["mailbox_it"] = mrun.history(it_mb), it_mb
        res["mailbox_u"], res["mailbox_p"] = mrun.sol[0].numpy(), mrun.sol[1].numpy()
        prof, nprof = mrun.profile(it_mb + 1, 8)          # (after the stop: the kernels return at once; the call path works)
        res["mailbox_profile_n"] = nprof
        mrun.release()
        # ---- fused row-partitioned MINRES (device phases + exchanges / all-reduces in between) ----------
        from distributed import DistributedMinres
        mr = DistributedMinres(sysm, f, g, blocks, dist, eng, comm=comm)
        um, errs, rel = mr.solve(tol=tol, maxsteps=maxsteps, poll_every=8)
        res["minres_errors"], res["minres_rel"] = np.array(errs), int(rel)
        res["minres_u"], res["minres_p"] = um[0].numpy(), um[1].numpy()
        #      ... the NATIVE C loop (nss_minres_iterate_dist) over the mailbox transport: grouped exchange of both operands
        #      (ring slots swapped per iteration), two one-double all-reduces -- with more than one rank
        mrm = DistributedMinres(sysm, f, g, blocks, dist, eng, comm=comm, transport="mailbox")
        assert mrm.native is not None and mrm.mailbox is not None
        umm, errsm, relm = mrm.solve(tol=tol, maxsteps=maxsteps, poll_every=16)
        res["minres_mb_errors"], res["minres_mb_rel"], res["minres_mb_u"] = np.array(errsm), int(relm), umm[0].numpy()
        res["minres_mb_timeout"] = int(mrm.mailbox.timed_out())
        mrm.close()
        # ---- fused row-partitioned BPCG v1 behind the reference's entry point (distributed.Bpcg1DistLoop) ----
        from bramble_pasciak_cg import bramble_pasciak_cg
        import distributed
        created = []
        orig = distributed.Bpcg1DistLoop.try_create.__func__

        def spy(cls, *a, **k):
            loop = orig(cls, *a, **k)
            created.append(loop is not None)
            return loop

        distributed.Bpcg1DistLoop.try_create = classmethod(spy)
        fv, gv = ops.vectors(f, g)
        with contextlib.redirect_stdout(io.StringIO()):
            x1, errs1 = bramble_pasciak_cg(ops.A, ops.B, None, ops.preA, ops.preM, fv, gv, tolerance=tol,
                                           max_steps=maxsteps, print_rates=False)
        res["bpcg1_fused"] = int(created == [True])
        res["bpcg1_errors"] = np.array(errs1)
        res["bpcg1_u"], res["bpcg1_p"] = x1[0].numpy(), x1[1].numpy()
        #      ... and its NATIVE C loop (nss_bpcg1_iterate_dist) over the mailbox transport with more than one rank
        distributed.Bpcg1DistLoop.TRANSPORT = "mailbox"
        loops = []

        def spy2(cls, *a, **k):
            loops.append(orig(cls, *a, **k))
            return loops[-1]

        distributed.Bpcg1DistLoop.try_create = classmethod(spy2)
        try:
            fv, gv = ops.vectors(f, g)
            with contextlib.redirect_stdout(io.StringIO()):
                x1m, errs1m = bramble_pasciak_cg(ops.A, ops.B, None, ops.preA, ops.preM, fv, gv, tolerance=tol,
                                                 max_steps=maxsteps, print_rates=False)
            assert loops and loops[-1] is not None and loops[-1].mailbox is not None and loops[-1].native is not None
            res["bpcg1_mb_errors"], res["bpcg1_mb_u"] = np.array(errs1m), x1m[0].numpy()
            res["bpcg1_mb_timeout"] = int(loops[-1].mailbox.timed_out())
            loops[-1].close()
        finally:
            distributed.Bpcg1DistLoop.TRANSPORT = None
            distributed.Bpcg1DistLoop.try_create = classmethod(orig)
    # ---- distributed AMG (replicated coarse levels) as preA, BPCG v2 through the protocol ------------
    if pre == "bjac":
        from distributed import DistributedAMG
        from solvers.bramblepasciak_new import BramblePasciakCG
        amg = DistributedAMG(sysm.A, ops.A, coarse_size=60)
        res["amg_levels"] = np.array(amg.level_sizes)
        xa = np.random.default_rng(9).standard_normal(sysm.n_u)
        va = hipla.Vector.from_numpy(xa[us])
        ya = hipla.Vector(ops.n_u)
        ya.data = amg * va
        res["amg_apply"] = ya.numpy()
        fv, gv = ops.vectors(f, g)
        sol = hipla.BlockVector([fv.CreateVector(), gv.CreateVector()])
        out = io.StringIO()
        with contextlib.redirect_stdout(out):
            it_a, _ = BramblePasciakCG(Form(ops.A), Form(ops.B), None, fv, gv, amg, ops.preM, sol, tol=tol,
                                       maxsteps=maxsteps)
        import re as _re
        res["amg_hist"] = np.array([float(m) for m in _re.findall(r"it =\s+\d+\s+err =\s+(\S+)", out.getvalue())])
        res["amg_it"] = it_a
        res["amg_u"] = sol[0].numpy()
    # ---- the reference's default preA on slabs: MypreA(GS=True) = Gauss-Seidel sweeps inside the slab around the
    #      auxiliary-space term on slabs, residual with the partitioned A; BPCG v2 through the protocol --------------
    if pre == "bjac":
        from solvers.bramblepasciak_new import BramblePasciakCG
        import re as _re2
        mops = DistributedStokes(sysm, blocks, comm, eng, pre="mypre_a", aux_options=dict(coarse_size=40))
        res["aux_levels"] = np.array(mops.aux.level_sizes)
        xa = np.random.default_rng(9).standard_normal(sysm.n_u)
        ya = hipla.Vector(mops.n_u)
        ya.data = mops.preA * hipla.Vector.from_numpy(xa[us])
        res["mypre_apply"] = ya.numpy()
        fv, gv = mops.vectors(f, g)
        sol = hipla.BlockVector([fv.CreateVector(), gv.CreateVector()])
        out = io.StringIO()
        with contextlib.redirect_stdout(out):
            it_m, _ = BramblePasciakCG(Form(mops.A), Form(mops.B), None, fv, gv, mops.preA, mops.preM, sol, tol=tol,
                                       maxsteps=maxsteps)
        res["mypre_hist"] = np.array([float(m) for m in _re2.findall(r"it =\s+\d+\s+err =\s+(\S+)", out.getvalue())])
        res["mypre_it"] = it_m
        res["mypre_u"] = sol[0].numpy()
        if mode == "gpu":
            # hybrid Gauss-Seidel (inside the slab, additive across slabs) natively in the fused partitioned loop
            grun = DistributedBpcg2(sysm, f, g, blocks, dist, eng, comm=comm, pre="bgs")
            it_g, _ = grun.solve(tol=tol, maxsteps=maxsteps, poll_every=8)
            res["bgs_hist"], res["bgs_it"], res["bgs_u"] = grun.history(it_g), it_g, grun.sol[0].numpy()
            res["bgs_colors"] = int(grun.ops.preA.ncolors)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    a = sys.argv[1:]
    run(int(a[0]), int(a[1]), a[2], a[3], a[4], int(a[5]), int(a[6]), a[7], float(a[8]), int(a[9]))
