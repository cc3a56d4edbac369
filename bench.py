#!/usr/bin/env python3
"""Headline benchmark: Krylov iterations / second of the Bramble-Pasciak CG (v2) hot loop on
the 3-D SIMPLE Stokes system (BASELINE.json configs[3]: MAC grid n=136, ~1e7 DoF, fp64),
plus the HBM roofline of its dominant kernel and the CPU oracle timed beside it.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one BPCG iteration (reference loop body: solvers/bramblepasciak_new.py:200-249):
3 CSR SpMVs (B^T, A, B), the block-Jacobi preconditioner apply, the lumped-mass apply, two
global inner products and the fused vector updates.  Inputs are resident in HBM before the
timed region.  For N > 1 (launched by torch.distributed.run, one rank per GPU) the same
global system is row-partitioned into slabs: strong scaling, value = iterations/s of the
whole job.  Rank 0 prints ONE JSON line.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "navier-stokes-solver_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s)
HBM_ACHIEVABLE_GBS = 6290.0  # measured float4 copy, same guide ("8 TB/s peak (spec); ~6.3 TB/s achievable")


def csrc_digest():
    """sha256 over the kernel sources (csrc/*.hip, *.h, include/nss_krylov.h): stored with every PMC
    traffic profile, so that a profile taken with other kernels is not replayed as this run's traffic."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "navier-stokes-solver_amd", "csrc", "*.hip"))
                   + glob.glob(os.path.join(ROOT, "navier-stokes-solver_amd", "csrc", "*.h"))
                   + [os.path.join(ROOT, "include", "nss_krylov.h")])
    for path in files:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--grid", dest="n", type=int, default=136, help="MAC grid cells per direction (136 -> 1.0e7 DoF)")
    ap.add_argument("--dim", type=int, default=3)
    ap.add_argument("--nu", type=float, default=0.01, help="1/Re")
    ap.add_argument("--pre", default="bjac3", choices=["bjac3", "jacobi", "bgs3", "bgs3p", "amg"],
                    help="preA: block Jacobi bs=3 (headline), point Jacobi, symmetric block Gauss-Seidel bs=3, "
                         "smoothed-aggregation AMG V(1,1)")
    ap.add_argument("--inflate", type=int, default=1,
                    help="HDG-like stress variant: Kronecker-inflate the operator with bs x bs blocks "
                         "(12 -> ~84 nnz per row as the reference's order-2 3-D spaces); preA = facet blocks")
    ap.add_argument("--cpu-iters", type=int, default=-1, help="oracle iterations to time (-1: auto, 0: skip)")
    ap.add_argument("--kernel-reps", type=int, default=30)
    ap.add_argument("--windows", type=int, default=5,
                    help="consecutive timed windows of --steps iterations each; `value` is their median "
                         "(a single ~0.1-s window cannot separate a 3 %% change from noise), all reported as window_values")
    ap.add_argument("--mypre-a", dest="mypre_a", type=int, default=1,
                    help="1: also run the reference's default preconditioner MypreA(GS=True) with the auxiliary-space term on "
                         "the headline system (`roofline_mypre_a_gs`); 0 skips it")
    ap.add_argument("--secondary", type=int, default=1,
                    help="1: also time the other BASELINE configurations (cfg2 MINRES, cfg3 BPCG v2) and MINRES / BPCG v1 "
                         "at the headline size through their entry points (`secondary_configs`); 0 skips them")
    ap.add_argument("--hdg", type=int, default=36,
                    help="grid of the secondary HDG-like measurement (facet blocks of 12 dofs, ~84 non-zeros per "
                         "row: the reference's row regime, SURVEY.md A7); 0 skips it")
    return ap.parse_args()


def hdg_like_roofline(torch, eng, grid, reps=40):
    """Secondary measurement in the reference's row regime (order-2 3-D HDG spaces: ~84 non-zeros per row
    in facet blocks of 12 dofs; templates/NavierStokesSIMPLE_iterative.py:24-26,360-362): the headline
    operator Kronecker-inflated with a 12 x 12 SPD block, facet-block Jacobi preA, same fused loop.  Returns
    the iteration rate and the roofline of its dominant launch (rows of A and B, C23), timed with HIP events
    inside the running loop."""
    import contextlib
    import io
    import hipla
    from solvers.bramblepasciak_new import BpcgSession
    from staggered_grid import mac_stokes
    sysm = mac_stokes(3, grid, 0.01).inflate(12)
    f, g = sysm.rhs(0)
    A, B = hipla.SparseMatrix.from_scipy(sysm.A), hipla.SparseMatrix.from_scipy(sysm.B)
    preA = hipla.BlockJacobi(A, sysm.line_blocks(1))              # the 12 dofs of one facet
    preM = hipla.DiagonalMatrix(1.0 / sysm.mass)
    sol = hipla.BlockVector([hipla.Vector(sysm.n_u), hipla.Vector(sysm.n_p)])
    with contextlib.redirect_stdout(io.StringIO()):
        ses = BpcgSession(Form(A), Form(B), None, hipla.Vector.from_numpy(f), hipla.Vector.from_numpy(g), preA, preM,
                          sol=sol)
    loop = ses.fused
    ses.first_direction()
    warm, timed_its = 10, 60
    loop.start(ses.wdn, ses.err0, 0.0, True, warm + timed_its + reps)
    loop.enqueue(0, warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop.enqueue(warm, warm + timed_its)
    torch.cuda.synchronize()
    per_it = (time.perf_counter() - t0) / timed_its
    marks = []
    for it in range(warm + timed_its, warm + timed_its + reps):
        loop.cphases("C1", "C1", it)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        loop.cphases("C23", "C23", it)
        e1.record()
        loop.cphases("SUMA", "SUMW", it)
        marks.append((e0, e1))
    torch.cuda.synchronize()
    ms = sum(a.elapsed_time(b) for a, b in marks[8:]) / (len(marks) - 8)
    a_info, b_info = A.handle.info(), B.handle.info()
    nbytes = (a_info["algorithmic_bytes"] + 16 * sysm.n_u + b_info["algorithmic_bytes"] + 8 * sysm.n_u + 24 * sysm.n_p)
    gbs = nbytes / (ms * 1e-3) / 1e9
    done, _, last = loop.poll()
    return {"workload": "3-D MAC Stokes n=%d inflated with 12 x 12 blocks: %d velocity dofs, %.1f non-zeros per row of A, "
                        "facet-block Jacobi (bs=12), BPCG v2" % (grid, sysm.n_u, a_info["nnz"] / sysm.n_u),
            "iters_per_s": 1.0 / per_it, "ms_per_iteration": 1e3 * per_it, "bound": "hbm",
            "kernel": "csr_stream_dual_kernel<EpiK2c, EpiK3c> (rows of A and B, grouped 16-bit column stream: one "
                      "index per %d entries)" % a_info["index_group"],
            "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "frac_of_achievable_6290": gbs / HBM_ACHIEVABLE_GBS, "algorithmic_bytes_per_launch": nbytes,
            "avg_launch_ms": ms, "bytes_per_nonzero": {"values": 8, "column_index": 2.0 / a_info["index_group"]},
            "valid": bool((not done) and last == warm + timed_its + reps - 1)}


class Form:
    def __init__(self, mat):
        self.mat, self.condense = mat, False


def entry_point_rates(torch, hipla, sysm, A, B, preA, solvers, k1, k2):
    """Iterations / second of the fused loops THROUGH THEIR DROP-IN ENTRY POINTS (set-up, polling and all): each
    solver runs k1 and k2 iterations with the stop test disabled (tolerance 0); the difference of the two wall
    times cancels the set-up (scale factor, initial residuals).  Returns {solver: {...}}."""
    import contextlib
    import io
    from bramble_pasciak_cg import bramble_pasciak_cg
    from minres import MinRes
    from solvers.bramblepasciak_new import BramblePasciakCG
    f, g = sysm.rhs(0)
    preS = hipla.DiagonalMatrix(1.0 / sysm.mass)
    fv, gv = hipla.Vector.from_numpy(f), hipla.Vector.from_numpy(g)

    def v2(k):
        sol = hipla.BlockVector([hipla.Vector(sysm.n_u), hipla.Vector(sysm.n_p)])
        BramblePasciakCG(Form(A), Form(B), None, fv, gv, preA, preS, sol, tol=0.0, maxsteps=k, printrates=False)

    def v1(k):
        bramble_pasciak_cg(A, B, None, preA, preS, fv, gv, tolerance=0.0, max_steps=k, print_rates=False)

    def mr(k):
        K = hipla.BlockMatrix([[A, B.T], [B, None]])
        Cm = hipla.BlockMatrix([[preA, None], [None, preS]])
        MinRes(mat=K, pre=Cm, rhs=hipla.BlockVector([fv, gv]), maxsteps=k, tol=0.0, printrates=False)

    def timed(fn, k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            fn(k)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    table = {"bpcg_v2": (v2, "solvers/bramblepasciak_new.BramblePasciakCG", 3),
             "bpcg_v1": (v1, "bramble_pasciak_cg.bramble_pasciak_cg", 6), "minres": (mr, "minres.MinRes", 3)}
    out = {}
    for name in solvers:
        fn, entry, spmv = table[name]
        timed(fn, 10)                                        # warm-up: transposes, workspaces
        t1 = min(timed(fn, k1) for _ in range(2))
        t2 = min(timed(fn, k2) for _ in range(2))
        per = (t2 - t1) / (k2 - k1)
        out[name] = {"entry_point": entry, "iters_per_s": 1.0 / per, "us_per_iteration": 1e6 * per,
                     "spmv_per_iteration": spmv, "iterations_timed": [k1, k2]}
    return out


def mypre_a_gs_roofline(torch, eng, hipla, sysm, A, B, its=60, warm=10):
    """The reference's DEFAULT velocity preconditioner on the headline system (what `SolveInitial(iterative=True)`
    runs: templates/NavierStokesSIMPLE_iterative.py:168,360-391,397): MypreA(GS=True) = forward multicolour block
    Gauss-Seidel sweep over the facet-like blocks, residual, auxiliary-space correction `transform @ preAh1 @
    transform.T` (one V-cycle per velocity component on the nodal Laplacians), backward sweep -- natively inside the
    fused BPCG v2 loop.  Iteration rate, and the device time / algorithmic bytes / fraction of the HBM peak of its two
    kernels groups: one sweep call (gather into the colour-major numbering, one launch per colour with the block solve
    in the epilogue, scatter) and the auxiliary-space term (two SpMVs with the transform around the component V-cycles),
    each timed with HIP events between cache-sweeping launches."""
    import contextlib
    import io
    from solvers.bramblepasciak_new import BpcgSession
    from templates.NavierStokesSIMPLE_iterative import MypreA, auxiliary_space_preconditioner
    t0 = time.perf_counter()
    space = sysm.auxiliary_space()                       # host assembly of the auxiliary operators (scipy; NGSolve's job
    jac_blocks = sysm.line_blocks(3)                     # in the reference) -- timed apart from the set-up proper
    t_space = time.perf_counter() - t0
    t0 = time.perf_counter()
    _, _, aux = auxiliary_space_preconditioner(sysm, space)
    preA = MypreA(None, Form(A), jac_blocks, GS=True, aux=aux)
    torch.cuda.synchronize()
    t_pre = time.perf_counter() - t0
    f, g = sysm.rhs(0)
    sol = hipla.BlockVector([hipla.Vector(sysm.n_u), hipla.Vector(sysm.n_p)])
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        ses = BpcgSession(Form(A), Form(B), None, hipla.Vector.from_numpy(f), hipla.Vector.from_numpy(g), preA,
                          hipla.DiagonalMatrix(1.0 / sysm.mass), sol=sol)
    loop = ses.fused
    if loop is None:
        raise RuntimeError("fused loop declined MypreA(GS=True) with the auxiliary-space term")
    ses.first_direction()
    loop.start(ses.wdn, ses.err0, 0.0, True, warm + 3 * its + 40)
    loop.enqueue(0, warm)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t0
    wins = []
    for w in range(3):
        t0 = time.perf_counter()
        loop.enqueue(warm + w * its, warm + (w + 1) * its)
        torch.cuda.synchronize()
        wins.append((time.perf_counter() - t0) / its)
    per_it = sorted(wins)[1]
    marks = []
    for it in range(warm + 3 * its, warm + 3 * its + 32):       # C1 + the whole preconditioner apply, inside the loop
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        loop.cphases("C1", "C1", it)
        e1.record()
        loop.cphases("C23", "SUMW", it)
        marks.append((e0, e1))
    torch.cuda.synchronize()
    c1_ms = sum(a.elapsed_time(b) for a, b in marks[8:]) / (len(marks) - 8)
    done, _, last = loop.poll()
    valid = bool((not done) and np_isfinite(loop.history(last)))
    # the two kernel groups on their own, between launches that sweep the caches
    n = sysm.n_u
    x, y = hipla.Vector.from_numpy(f), hipla.Vector(n)
    big = [eng.zeros(1 << 25) for _ in range(3)]

    def timed(fn, reps=12):
        ev = []
        for _ in range(reps + 2):
            eng.stream_triad(0.5, big[0], big[1], big[2])
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            ev.append((a, b))
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in ev[2:]) / reps

    sweep_ms = timed(lambda: preA.Smooth(y, x))
    aux_ms = timed(lambda: aux.Mult(x, y))
    p = preA.perm_handle.info()
    bs = preA.bs
    sweep_bytes = (8 * p["nnz"] + p["index_bytes"] * (p["nnz"] // p["index_group"]) + 4 * (p["rows"] + 1)   # P A P^T as stored
                   + 8 * bs * n + 8 * n + 16 * n                                                            # inverse blocks, x, y in / out
                   + 2 * 16 * n + 16 * n)                                                                   # gathers of x, y; scatter of y
    spmv_bytes = lambda i: i["algorithmic_bytes"]
    aux_bytes = spmv_bytes(aux.transform.handle.info()) + spmv_bytes(aux.transform_t.handle.info())
    # components that share one hierarchy are cycled together (csrc/amg.hip: csr_multi_kernel): every level operator
    # is read once (4-byte columns), the vectors K times
    together = len(set(id(c) for c in aux.components)) == 1 and len(aux.components) in (2, 3) \
        and os.environ.get("NSS_AMG_BATCH", "1") != "0" and len(aux.components[0].levels) >= 2
    ncomp = len(aux.components)
    multi_bytes = lambda i: 12 * i["nnz"] + 4 * (i["rows"] + 1) + 8 * ncomp * (i["rows"] + i["cols"])
    for comp in (aux.components[:1] if together else aux.components):
        one = multi_bytes if together else spmv_bytes
        for lv in comp.levels:
            aux_bytes += one(lv["inv"].handle.info()) if "inv" in lv else (
                2 * one(lv["A"].handle.info()) + one(lv["P"].handle.info()) + one(lv["R"].handle.info())
                + 8 * 6 * lv["n"] * (ncomp if together else 1))

    def roof(nbytes, ms):
        gbs = nbytes / (ms * 1e-3) / 1e9
        return {"avg_ms": ms, "algorithmic_bytes": int(nbytes), "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": gbs / HBM_PEAK_GBS}

    return {"workload": "headline system (%d DoF), BPCG v2, preA = MypreA(GS=True): multicolour block Gauss-Seidel sweeps "
                        "(bs=%d, %d colours, %s layout) around the auxiliary-space term (%d component V-cycles, levels %s)"
                        % (sysm.ndof, bs, preA.ncolors, preA.layout, len(aux.components), aux.level_sizes[0]),
            "reference": "templates/NavierStokesSIMPLE_iterative.py:168,376-381,397",
            "iters_per_s": 1.0 / per_it, "ms_per_iteration": 1e3 * per_it, "window_values": [1.0 / w for w in wins],
            "scale_factor_k": ses.k, "C1_with_whole_preA_ms_in_loop": c1_ms,
            "sweep_call": dict(roof(sweep_bytes, sweep_ms), kernel="gs_enter + csr_stream_kernel<EpiGsFused> x %d colours + "
                               "gs_leave (one Smooth / SmoothBack call)" % preA.ncolors, calls_per_iteration=2),
            "auxiliary_space_term": dict(roof(aux_bytes, aux_ms), kernel="T^T SpMV, %d smoothed-aggregation V(1,1)-cycles%s, "
                                         "T SpMV (nss_amg_create_auxiliary)"
                                         % (ncomp, " cycled together: every level operator read once (csr_multi_kernel)" if together else ""),
                                         calls_per_iteration=1),
            "setup_s": {"assemble_auxiliary_space_host": t_space, "amg_hierarchies_colouring_permutation": t_pre,
                        "scale_factor_initial_residual": t_setup},
            "valid": valid}


def np_isfinite(a):
    import numpy as np
    return bool(np.all(np.isfinite(a)))


def secondary_configs(torch, hipla, headline=None):
    """The other BASELINE.json configurations that fit one GPU in a second each -- cfg2 (stokes_hcurldiv.py restated:
    2-D n=183, ~1e5 DoF, MINRES) and cfg3 (NavierStokesSIMPLE_test.py restated: 2-D n=577, ~1e6 DoF, BPCG v2) --
    and, on the headline matrices (`headline` = (sysm, A, B, preA)), MINRES and BPCG v1: all three fused loops at
    all sizes, timed by the driver's own run (block-Jacobi bs=3 preA, lumped-mass preS)."""
    from staggered_grid import mac_stokes
    out = {}
    if headline is not None:
        sysm, A, B, preA = headline
        out["cfg4_other_solvers"] = {"workload": "headline matrices (%d DoF)" % sysm.ndof,
                                     **entry_point_rates(torch, hipla, sysm, A, B, preA, ("minres", "bpcg_v1"), 50, 350)}
    for name, dim, n, solvers, ref in (("cfg2", 2, 183, ("minres", "bpcg_v2", "bpcg_v1"), "stokes_hcurldiv.py 2D, MINRES, ~1e5 DoF"),
                                       ("cfg3", 2, 577, ("bpcg_v2", "minres"), "templates/NavierStokesSIMPLE_test.py 2D, BPCG, ~1e6 DoF")):
        sysm = mac_stokes(dim, n, 0.01)
        A, B = hipla.SparseMatrix.from_scipy(sysm.A), hipla.SparseMatrix.from_scipy(sysm.B)
        preA = hipla.BlockJacobi(A, sysm.line_blocks(3))
        out[name] = {"workload": "%s restated: %d-D MAC Stokes n=%d, %d DoF" % (ref, dim, n, sysm.ndof),
                     **entry_point_rates(torch, hipla, sysm, A, B, preA, solvers, 500, 4500)}
        del A, B, preA
        torch.cuda.empty_cache()
    return out


def pmc_traffic(kernel_substring, workload_args):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/*_traffic.json, written by tools/profile_bench.sh on this same command; FETCH_SIZE
    corrected x2 as MI355X_MICROARCH.md prescribes, calibrated on the triad kernel).  PMC counters
    cannot be read from inside an unprofiled run, so the figure comes from a profile -- but only from
    one taken with EXACTLY these kernel sources (`csrc_sha256` stored in the profile == csrc_digest()
    now).  Returns (record-or-None, note)."""
    import glob
    digest = csrc_digest()
    best, note = None, "no PMC profile of this workload under profiles/"
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json"))):
        try:
            with open(path) as fh:
                doc = json.load(fh)
        except (OSError, ValueError):
            continue
        if doc.get("workload") and doc["workload"] != workload_args:
            continue
        if doc.get("csrc_sha256") != digest:
            if best is None:
                note = ("%s was taken with other kernel sources (csrc hash differs): not replayed"
                        % os.path.relpath(path, ROOT))
            continue
        for name, rec in doc.get("kernels", {}).items():
            if kernel_substring in name:
                best = {"bytes": rec["hbm_bytes_per_launch"], "source": os.path.relpath(path, ROOT)}
                note = "rocprofv3 PMC passes of this command with these kernel sources (profile, not this run)"
    return best, note


def event_time_ms(torch, fn, reps):
    """Average duration of `fn` (one kernel launch on the current stream) from HIP events."""
    start = torch.cuda.Event(enable_timing=True)
    stop = torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    start.record()
    for _ in range(reps):
        fn()
    stop.record()
    stop.synchronize()
    return start.elapsed_time(stop) / reps


# data paths of a multi-rank run, slowest first: torch.distributed collectives from the Python schedule; RCCL through
# ctypes from the Python schedule; RCCL issued by the native C loop; the native C loop over the mailbox transport
# (csrc/p2p.h: all-reduces inside the sum kernels, halos by put kernels into peer-mapped landing zones -- no collective
# library inside an iteration)
TIERS = ("torch", "rccl-python", "native", "mailbox")
TIER_MARK = "NSS_TIER_OK "


def rehearse_native_path(args, rank):
    """Multi-rank runs only.  The RCCL-through-ctypes communicator and the native partitioned loop
    cannot be exercised on the single-GPU development box (RCCL refuses two ranks on one device),
    so before this process touches its GPU every rank starts a CHILD `bench.py` that runs the same
    code paths on a small system (own rendezvous port), with a time limit: first RCCL driven from
    the Python schedule, then the native C loop, each cross-checked against torch.distributed; the
    child prints a marker per path that held up.  Whatever hangs or crashes in the child is not
    used by the parent job, which still reports a number on the best proven path.  Returns the index
    into TIERS of the fastest proven path."""
    import subprocess
    forced = os.environ.get("NSS_PROBE_FORCE") == "1"     # exercise the child mechanics on any backend (tests)
    if not forced and (os.environ.get("NSS_COMM", "rccl") != "rccl"
                       or os.environ.get("NSS_DIST_BACKEND", "nccl") != "nccl"):
        return 0
    if os.environ.get("NSS_SKIP_REHEARSAL") == "1":
        return len(TIERS) - 1
    env = os.environ.copy()
    env["NSS_PROBE_CHILD"] = "1"
    env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 1 + int(os.environ.get("NSS_PROBE_PORT_OFFSET", "36")))
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)        # the child group brings up its own store
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(args.gpus), "--grid", "48", "--steps", "6",
           "--warmup", "2", "--cpu-iters", "0", "--pre", args.pre if args.pre in ("bjac3", "jacobi") else "bjac3"]
    limit = float(os.environ.get("NSS_PROBE_TIMEOUT", "240"))
    t0 = time.perf_counter()
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    try:
        _, err = child.communicate(timeout=limit)
        rc = child.returncode
    except subprocess.TimeoutExpired:
        child.kill()
        _, err = child.communicate()
        rc = -9
    text = err.decode(errors="replace")
    level = max([0] + [TIERS.index(t) for t in TIERS if TIER_MARK + t in text])
    tail = [ln for ln in text.strip().splitlines() if TIER_MARK not in ln][-4:]
    print("rank %d: rehearsal (%.0f s, exit %s): best proven data path = %s%s"
          % (rank, time.perf_counter() - t0, rc, TIERS[level],
             "" if level == len(TIERS) - 1 else "\n  " + "\n  ".join(tail)), file=sys.stderr)
    return level


def emit(fd, doc):
    """The ONE JSON line, written to the process's original stdout."""
    os.write(fd, (json.dumps(doc) + "\n").encode())


def main():
    args = parse_args()
    # Everything except the result line goes to stderr -- also what native libraries print
    # (RCCL writes a version banner to fd 1 when a communicator is created).
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import numpy as np
    import torch
    import torch.distributed as dist

    # NSS_FORCE_DIST=1 sends a 1-rank launch through the partitioned code path (rehearsal of the
    # RCCL bootstrap / native loop on a single GPU)
    partitioned = world > 1 or os.environ.get("NSS_FORCE_DIST") == "1"
    probe_child = os.environ.get("NSS_PROBE_CHILD") == "1"
    tier_level = len(TIERS) - 1
    if partitioned and world > 1 and not probe_child:
        tier_level = rehearse_native_path(args, rank)             # before this process initialises its GPU
    if partitioned:
        # NSS_DIST_BACKEND=gloo lets several ranks share one GPU for rehearsals (RCCL refuses
        # that); the driver's runs use the default: one rank per GPU over RCCL / xGMI.
        backend = os.environ.get("NSS_DIST_BACKEND", "nccl")
        device = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(device)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)
        if world > 1:                                    # every rank must take the same data path
            agree = torch.tensor([float(tier_level)], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(agree, op=dist.ReduceOp.MIN)
            tier_level = int(agree.item())
    if args.gpus != world and rank == 0:
        print("note: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)

    import hipla
    from solvers.bramblepasciak_new import BpcgSession
    from staggered_grid import mac_stokes

    import contextlib
    eng = hipla.get_engine()
    info = eng.device_info()
    quiet = contextlib.redirect_stdout(sys.stderr)      # stdout carries the ONE JSON line only
    K, W = args.steps, args.warmup
    NWIN = max(1, args.windows)          # consecutive timed windows of exactly K steps each; value = their median
    total_its = W + NWIN * K

    t_asm = time.perf_counter()
    sysm = mac_stokes(args.dim, args.n, args.nu)
    if args.inflate > 1:
        est = sysm.A.nnz * args.inflate ** 2
        if est > 6e8:                                    # host assembly needs ~100 B per non-zero
            raise SystemExit("--inflate %d of --grid %d would give %.1e non-zeros in A; use a smaller "
                             "--grid (e.g. --grid 44 --inflate 12, --grid 60 --inflate 5)" % (args.inflate, args.n, est))
        sysm = sysm.inflate(args.inflate)
    f, g = sysm.rhs(0)
    blocks = sysm.line_blocks(3) if args.pre in ("bjac3", "bgs3", "bgs3p") else None
    if args.inflate > 1 and blocks is not None:
        blocks = sysm.facet_blocks() if sysm.dim * args.inflate <= 16 else sysm.line_blocks(1)
    gs_colors = None
    if args.pre == "bgs3p":        # block Gauss-Seidel in the colour-permuted dof space
        from hipla import coloring
        perm, blocks, gs_colors = coloring.colour_permutation(sysm.A, blocks)
        sysm = sysm.permuted(perm)
        f = f[perm]
    t_asm = time.perf_counter() - t_asm

    if partitioned:
        from distributed import DistributedBpcg2, TorchComm
        comm_kind = "torch.distributed/" + backend
        crosscheck_ms = {}
        rccl = None
        if backend == "nccl" and os.environ.get("NSS_COMM", "rccl") == "rccl" and tier_level >= 1:
            try:                                   # RCCL straight through ctypes on the compute stream
                from rccl_comm import RcclComm
                rccl = RcclComm(dist, eng)
                rccl.self_test(torch)
            except Exception as exc:               # any doubt -> the torch.distributed data path
                print("rank %d: RCCL ctypes communicator unavailable (%s); using torch.distributed" % (rank, exc),
                      file=sys.stderr)
                rccl = None
        flag = torch.tensor([1.0 if rccl is not None else 0.0], dtype=torch.float64,
                            device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)      # all ranks must agree on the data path
        if flag.item() == 0.0:
            rccl = None
        torch_comm = TorchComm(dist, eng)
        comm, use_native = torch_comm, False

        def probe_path(c, native, transport=None, probe_its=6, timed_its=40):
            """History of a few iterations on one data path + its cost per iteration."""
            with quiet:
                probe = DistributedBpcg2(sysm, f, g, blocks, dist, eng, comm=c, native=native, transport=transport)
            probe.start(tol=0.0, maxsteps=probe_its + timed_its)
            probe.iterate(0, probe_its)
            torch.cuda.synchronize()
            h = probe.history(probe_its - 1)
            dist.barrier()
            t_probe = time.perf_counter()
            probe.iterate(probe_its, probe_its + timed_its)
            torch.cuda.synchronize()
            probe.release()
            return h, 1e3 * (time.perf_counter() - t_probe) / timed_its

        transport = None
        candidates = []
        if rccl is not None:
            candidates += [("rccl-python", rccl, False, None), ("native", rccl, True, None)]
        # (with the gloo rehearsal backend -- several ranks on one GPU -- only on request: NSS_MAILBOX=1)
        if ((backend == "nccl" and os.environ.get("NSS_MAILBOX", "1") != "0") or os.environ.get("NSS_MAILBOX") == "1") \
                and world <= 16:
            candidates += [("mailbox", torch_comm, True, "mailbox")]
        candidates = [c for c in candidates if TIERS.index(c[0]) <= tier_level]
        if candidates:
            # cross-check every candidate path against torch.distributed collectives from the same state;
            # the rehearsal child walks them slowest first (a hang then leaves the markers of the paths
            # that held up), the job itself fastest first
            ref_hist, crosscheck_ms["torch"] = probe_path(torch_comm, False)
            if not probe_child:
                candidates.reverse()
            proven = []
            for label, cand_comm, native, cand_transport in candidates:
                try:
                    h, crosscheck_ms[label] = probe_path(cand_comm, native, cand_transport)
                    same = bool(np.all(np.isfinite(h)) and np.allclose(h, ref_hist, rtol=1e-9, atol=0.0))
                except Exception as exc:
                    print("rank %d: data path %s raised %r" % (rank, label, exc), file=sys.stderr)
                    same = False
                    crosscheck_ms[label] = float("inf")
                flag = torch.tensor([1.0 if same else 0.0, -crosscheck_ms[label] if same else 0.0], dtype=torch.float64,
                                    device="cuda" if backend == "nccl" else "cpu")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)            # all ranks agree; the slowest rank's time counts
                if flag[0].item() == 1.0:
                    if probe_child:
                        print(TIER_MARK + label, file=sys.stderr, flush=True)
                        continue
                    proven.append((-flag[1].item(), label, cand_comm, native, cand_transport))
                    if not native:
                        break                                          # (the Python schedule is never faster than what follows it)
                    continue
                print("rank %d: data path %s disagrees with torch.distributed; not used" % (rank, label), file=sys.stderr)
            if proven:                                                 # the fastest of the paths that held up
                _, label, comm, use_native, transport = min(proven, key=lambda t: t[0])
                comm_kind = ("mailbox transport (peer-mapped memory over xGMI, csrc/p2p.h)" if transport else
                             "rccl-ctypes" + ("" if use_native else " issued from the Python schedule"))
        if probe_child:
            dist.barrier()
            dist.destroy_process_group()
            sys.stderr.flush()
            os._exit(0)
        with quiet:
            run = DistributedBpcg2(sysm, f, g, blocks, dist, eng, comm=comm, native=use_native, transport=transport)
        if run.native is not None:
            comm_kind += " + native loop (%s)" % ("exchange overlapped with the interior rows" if run.overlap
                                                  else "exchange, then SpMV, on one stream")
        prof_its = 30
        run.start(tol=0.0, maxsteps=total_its + prof_its)
        run.iterate(0, W)
        windows = []
        for w in range(NWIN):                               # each window: K steps, barrier + synchronize on both sides
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run.iterate(W + w * K, W + (w + 1) * K)
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            windows.append(time.perf_counter() - t0)
        tt = torch.tensor(windows, dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)           # per window: the slowest rank
        windows = [float(v) for v in tt.tolist()]
        elapsed = float(np.median(windows))
        done, _, last = run.poll()
        hist = run.history(total_its - 1)
        ok = (not done) and last == total_its - 1 and bool(np.all(np.isfinite(hist)))
        # roofline of the dominant kernel on this rank's slab, HIP events on its stream, each launch right after the
        # kernel that precedes it in the loop (the loop's cache state: a kernel repeated back to back finds its
        # operands in L2 / MALL and reads ~12 % low).  Compact plan: C23 = rows of A and of B in one launch;
        # eight-phase plan: K2 = rows of A.
        a_loc = run.ops.A.local.handle.info()
        if run.compact:
            b_loc = run.ops.b_extended().handle.info()
            k2_bytes = (a_loc["algorithmic_bytes"] + 16 * run.ops.n_u + b_loc["algorithmic_bytes"] + 8 * run.ops.n_u
                        + 24 * b_loc["rows"])
            k2_name = "csr_stream_dual_kernel<EpiK2c, EpiK3c> (rows of A and B, C23) on rank 0's slab"
            before, timed = (lambda: run.loop.cphases("C1", "C1", total_its - 1)), (lambda: run.loop.cphases("C23", "C23", total_its - 1))
        else:
            k2_bytes = a_loc["algorithmic_bytes"] + 24 * run.ops.n_u
            k2_name = "csr_stream_kernel<1, EpiK2> on rank 0's slab"
            before, timed = (lambda: run.loop.phase("K1", total_its - 1)), (lambda: run.loop.phase("K2", total_its - 1))
        marks = []
        for _ in range(args.kernel_reps + 4):
            before()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            timed()
            e1.record()
            marks.append((e0, e1))
        torch.cuda.synchronize()
        k2_ms = sum(a.elapsed_time(b) for a, b in marks[4:]) / (len(marks) - 4)
        k2_gbs = k2_bytes / (k2_ms * 1e-3) / 1e9
        dist.barrier()
        # which collective costs what: per-phase device times of further iterations of the SAME loop
        try:
            phase_ms, phase_n = run.profile(total_its, prof_its)
        except Exception as exc:                     # never lose the result line over the diagnostics
            print("rank %d: per-phase profile failed: %r" % (rank, exc), file=sys.stderr)
            phase_ms, phase_n = None, 0
        dist.barrier()
        if rank == 0:
            out = {
                "metric": "Krylov iters/sec, 3D SIMPLE Stokes solve (BPCG)", "value": K / elapsed, "unit": "iters/s",
                "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": 1e3 * elapsed / K,
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
                "data": "synthetic",
                "timing": "median of %d consecutive windows of %d steps (max over ranks per window)" % (NWIN, K),
                "window_values": [K / w for w in windows],
                "config": {"workload": "templates/NavierStokesSIMPLE_test_3D.py restated: 3-D MAC Stokes n=%d, "
                                       "%d DoF, BPCG v2, %s preA, row-partitioned over %d GPUs"
                                       % (args.n, sysm.ndof, args.pre, world),
                           "n_u": sysm.n_u, "n_p": sysm.n_p, "nnz_A": int(sysm.A.nnz), "nnz_B": int(sysm.B.nnz)},
                "roofline": {"bound": "hbm", "kernel": k2_name, "timing": "HIP events around the kernel, each launch right after its predecessor in the loop on the same slab",
                             "achieved": k2_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k2_gbs / HBM_PEAK_GBS,
                             "traffic": None, "algorithmic_bytes_per_launch": k2_bytes, "avg_launch_ms": k2_ms},
                "cpu_baseline": None,
                "valid": ok, "halo_doubles_per_rank": run.halo_summary(), "comm": comm_kind,
                "plan": ("compact: C1, preA, exchange(t1), C23, sum, all-reduce, C4, sum, all-reduce" if run.compact
                         else "eight-phase: K1, preA, exchange(t1), K2, K3, sum, all-reduce, K4, sum, all-reduce, K5"),
                "ms_per_iteration_by_path_rank0": crosscheck_ms or None, "rehearsed_best_path": TIERS[tier_level],
                "phase_ms_rank0": phase_ms, "phase_iterations": phase_n,
            }
            emit(result_fd, out)
        dist.destroy_process_group()
        return

    # ------------------------------- single GPU --------------------------------------------
    A = hipla.SparseMatrix.from_scipy(sysm.A)
    B = hipla.SparseMatrix.from_scipy(sysm.B)
    if args.pre in ("bgs3", "bgs3p"):
        preA = hipla.BlockGaussSeidel(A, blocks, colors=gs_colors)
    elif args.pre == "amg":
        preA = hipla.SmoothedAggregationAMG(A)
    else:
        preA = hipla.BlockJacobi(A, blocks) if blocks is not None else hipla.JacobiPreconditioner(A)
    preM = hipla.DiagonalMatrix(1.0 / sysm.mass)
    sol = hipla.BlockVector([hipla.Vector(sysm.n_u), hipla.Vector(sysm.n_p)])
    t_setup = time.perf_counter()
    with quiet:
        ses = BpcgSession(Form(A), Form(B), None, hipla.Vector.from_numpy(f), hipla.Vector.from_numpy(g),
                          preA, preM, sol=sol, initialize=True)
    loop = ses.fused
    if loop is None:
        raise RuntimeError("fused BPCG loop unavailable for native operands")
    ses.first_direction()
    probe_its = 48                                           # extra iterations for the in-loop kernel timings
    loop.start(ses.wdn, ses.err0, 0.0, True, total_its + 2 * probe_its)   # tol = 0: never stops inside the run
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup

    loop.enqueue(0, W)
    torch.cuda.synchronize()
    windows = []
    for w in range(NWIN):                                   # each window: exactly K steps between two synchronisations
        t0 = time.perf_counter()
        loop.enqueue(W + w * K, W + (w + 1) * K)
        torch.cuda.synchronize()
        windows.append(time.perf_counter() - t0)
    elapsed = float(np.median(windows))
    done, _, last = loop.poll()
    hist = loop.history(total_its - 1)
    valid = (not done) and last == total_its - 1 and bool(np.all(np.isfinite(hist)))

    # ---- roofline of the dominant kernel: C23 = rows of A and rows of B in one launch -------------
    a_info, b_info = A.handle.info(), B.handle.info()
    bt_info = ses.matBT.handle.info()
    dual = (a_info["lanes_per_row"], a_info["index_bytes"]) == (b_info["lanes_per_row"], b_info["index_bytes"])
    # A part: t2 = A t1 (x = t1 and y = t2 inside algorithmic_bytes) + s0, t0 read for <s0, t2 - t0>;
    # B part: t3 = B (t1 - s0) (x, y inside algorithmic_bytes; the operand is formed from TWO gathered
    # vectors: + 8 n_u) + s1, w1 read, s1 written (s1 = beta s1 + w1) for <s1, t3>
    k2_bytes = (a_info["algorithmic_bytes"] + 16 * sysm.n_u
                + b_info["algorithmic_bytes"] + 8 * sysm.n_u + 24 * sysm.n_p)
    reps = args.kernel_reps
    # Kernel durations *inside the iteration*: HIP events on the loop's stream around each phase of
    # `probe_its` further iterations.  (A kernel repeated back to back finds its operands in L2 / MALL
    # from the previous repetition -- the A SpMV then looks 12 % faster than it is in the loop -- so nothing
    # is timed in isolation; the per-dispatch durations of rocprofv3 --kernel-trace for the same launches
    # are 2 % below these numbers: an event pair also sees the dispatch of the kernel it brackets.)
    names = ("C1", "C23", "SUMA", "C4", "SUMW")
    phase_ms = dict.fromkeys(names, 0.0)
    marks = []
    xs, ys = eng.zeros(sysm.n_u), eng.zeros(sysm.n_u)
    xs.fill_(1.0)
    spmv_marks = []
    for it in range(total_its, total_its + probe_its):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
        ev[0].record()
        for k, name in enumerate(names):
            loop.cphases(name, name, it)
            ev[k + 1].record()
        marks.append(ev)
    # the plain y = A x launch in the loop's cache state: between two whole iterations (it finds the caches as the
    # A rows of C23 find them behind C1 + preA), never back to back with itself; a second stretch of iterations so
    # that the phase timings above are not disturbed by it
    for it in range(total_its + probe_its, total_its + 2 * probe_its):
        loop.cphases("C1", "SUMW", it)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.csr_spmv(A.handle, 1.0, xs, 0.0, ys)
        e1.record()
        spmv_marks.append((e0, e1))
    torch.cuda.synchronize()
    skip = 8                                                 # event creation, clocks
    for ev in marks[skip:]:
        for k, name in enumerate(names):
            phase_ms[name] += ev[k].elapsed_time(ev[k + 1]) / (len(marks) - skip)
    k1_ms, k2_ms, k4_ms = phase_ms["C1"], phase_ms["C23"], phase_ms["C4"]
    sums_ms = phase_ms["SUMA"] + phase_ms["SUMW"]
    spmv_ms = sum(a.elapsed_time(b) for a, b in spmv_marks[skip:]) / (len(spmv_marks) - skip)
    spmv_b2b_ms = event_time_ms(torch, lambda: eng.csr_spmv(A.handle, 1.0, xs, 0.0, ys), reps)
    ntri = 1 << 26
    ta, tb, tc = eng.zeros(ntri), eng.zeros(ntri), eng.zeros(ntri)
    triad_ms = event_time_ms(torch, lambda: eng.stream_triad(0.5, ta, tb, tc), reps)
    triad_gbs = 24.0 * ntri / (triad_ms * 1e-3) / 1e9
    del ta, tb, tc
    k2_gbs = k2_bytes / (k2_ms * 1e-3) / 1e9
    spmv_gbs = a_info["algorithmic_bytes"] / (spmv_ms * 1e-3) / 1e9

    # per-iteration algorithmic bytes (DESIGN.md section "bytes per iteration")
    n_u, n_p = sysm.n_u, sysm.n_p
    mat_bytes = sum(8 * i["nnz"] + i["index_bytes"] * (i["nnz"] // i["index_group"]) + 4 * (i["rows"] + 1)
                    for i in (a_info, b_info, bt_info))
    if args.pre == "amg":      # per level: two SpMVs with A_l (residual, post-smoothing), one each with P_l, R_l
        spmv_bytes = lambda i: 12 * i["nnz"] + 4 * (i["rows"] + 1) + 8 * (i["rows"] + i["cols"])
        pre_bytes = 0
        for lv in preA.levels:
            pre_bytes += spmv_bytes(lv["inv"].handle.info()) if "inv" in lv else (
                2 * spmv_bytes(lv["A"].handle.info()) + spmv_bytes(lv["P"].handle.info())
                + spmv_bytes(lv["R"].handle.info()) + 8 * 6 * lv["n"])
    else:
        pre_bytes = preA.handle.algorithmic_bytes() - 16 * n_u if blocks is not None else 8 * n_u
    if args.pre in ("bgs3", "bgs3p"):      # two sweeps, each walks the CSR rows of A once and applies the blocks once
        pre_bytes = 2 * (pre_bytes + 12 * a_info["nnz"] + 4 * n_u + 24 * n_u)
    # vector passes of one iteration (reads + writes, each n_u resp. n_p doubles):
    #   C1 6 + 5 (q, z0, t2, s0, w0, u0 | u0, z0, q, s0, t0) and the gathers of s1, w1;  preA 2 (t0 | t1);
    #   C23 rows of A: gather t1, s0, t0 | t2;  rows of B: gathers of t1, s0;  s1, w1 | s1, t3;
    #   C4 5 + 2 (t0, t1, t2, d0, w0 | d0, w0) and 6 + 3 (s1, t3, u1, d1, minv, w1 | u1, d1, w1)
    #   (block Jacobi applied in C1's epilogue: t0 is not read back -- one velocity pass fewer)
    pre_in_c1 = loop.c1_applies_preA()
    vec_bytes = 8 * ((25 if pre_in_c1 else 26) * n_u + 15 * n_p)
    iter_bytes = mat_bytes + pre_bytes + vec_bytes
    iter_gbs = iter_bytes / (elapsed / K) / 1e9

    # ---- CPU baseline: the oracle restatement on this box's host cores -------------------------
    cpu = None
    parity = None
    if args.cpu_iters != 0:
        from oracle import krylov_ref as kr
        cpu_iters = args.cpu_iters if args.cpu_iters > 0 else max(3, min(40, int(2.0e9 / max(sysm.A.nnz, 1))))
        if args.pre in ("bgs3", "bgs3p", "amg"):
            raise SystemExit("--pre %s: no timed CPU leg for this preconditioner; use --cpu-iters 0" % args.pre)
        pa = kr.block_jacobi(sysm.A, blocks) if blocks is not None else kr.jacobi(sysm.A)
        timing = {}
        it_c, _, _, hist_c, err0_c = kr.bpcg_v2(sysm.A, sysm.B, pa, kr.diag_inverse(sysm.mass), f, g, ses.k,
                                                tol=0.0, maxsteps=cpu_iters, timing=timing)
        cpu = {"value": cpu_iters / timing["loop_seconds"], "unit": "iters/s", "cores": 1, "kind": "port",
               "host_cpus": os.cpu_count(),
               "sample": "%d iterations of oracle/krylov_ref.bpcg_v2 (scipy CSR matvec + numpy, single thread "
                         "for SpMV) on the identical matrices" % cpu_iters}
        m = min(len(hist_c), len(hist))
        parity = {"iterations_compared": m,
                  "history_max_rel_diff": float(np.max(np.abs(hist[:m] - hist_c[:m]) / np.abs(hist_c[:m]))),
                  "err0_rel_diff": abs(ses.err0 - err0_c) / err0_c}

    hdg = None
    scale_k, folds = ses.k, loop.folds_sums()
    mypre = None
    if args.mypre_a and args.inflate == 1 and args.pre == "bjac3" and args.dim == 3:
        try:                                                # (never at the cost of the headline line)
            mypre = mypre_a_gs_roofline(torch, eng, hipla, sysm, A, B)
        except Exception as exc:
            mypre = {"error": repr(exc)}
        torch.cuda.empty_cache()
    secondary = None
    if args.secondary and args.inflate == 1 and args.pre == "bjac3":
        try:                                                # secondary measurements must never cost the headline line
            secondary = secondary_configs(torch, hipla, (sysm, A, B, preA))
        except Exception as exc:
            secondary = {"error": repr(exc)}
    if args.hdg > 0 and args.inflate == 1:
        del ses, loop, sol, A, B, preA                      # the headline system's device memory
        torch.cuda.empty_cache()
        try:                                                # a secondary measurement must never cost the headline line
            hdg = hdg_like_roofline(torch, eng, args.hdg)
        except Exception as exc:
            hdg = {"error": repr(exc)}
    traffic, traffic_note = pmc_traffic("EpiK2c", "grid=%d dim=%d pre=%s" % (args.n, args.dim, args.pre))
    k2_bytes_int32 = k2_bytes + sum(4 * i["nnz"] - 2 * (i["nnz"] // i["index_group"]) for i in (a_info, b_info)
                                    if i["index_bytes"] == 2)
    out = {
        "metric": "Krylov iters/sec, 3D SIMPLE Stokes solve (BPCG)", "value": K / elapsed, "unit": "iters/s",
        "n_gpus": 1, "steps": K, "warmup": W, "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "timing": "median of %d consecutive windows of %d steps" % (NWIN, K),
        "window_values": [K / w for w in windows],
        "config": {"workload": "templates/NavierStokesSIMPLE_test_3D.py restated: %d-D MAC Stokes n=%d, %d DoF, "
                               "BPCG v2 (solvers/bramblepasciak_new.py), %s preA, lumped-mass preM, Re=%g"
                               % (args.dim, args.n, sysm.ndof, args.pre, 1.0 / args.nu),
                   "n_u": n_u, "n_p": n_p, "nnz_A": a_info["nnz"], "nnz_B": b_info["nnz"],
                   "scale_factor_k": scale_k, "device": info["arch"], "cu_count": info["cu_count"],
                   "column_index_bytes": {"A": a_info["index_bytes"], "B": b_info["index_bytes"],
                                          "BT": bt_info["index_bytes"]},
                   "entries_per_column_index": {"A": a_info["index_group"], "B": b_info["index_group"],
                                                "BT": bt_info["index_group"]},
                   # how the kernels reach the SpMV operand (csrc/csr_stream.h): staged = LDS copy of the row
                   # block's column runs by LDS-DMA (kernels whose operand is one stored vector; the others
                   # gather through the 16-bit window form of the same matrix), rows = row-per-lane kernel
                   "operand_form": {"A": a_info["operand_form"], "B": b_info["operand_form"],
                                    "BT": bt_info["operand_form"]}},
        "roofline": {"bound": "hbm",
                     "kernel": ("csr_stream_dual_kernel<EpiK2c, EpiK3c>: t2 = A t1 with <s0, t2 - t0> and "
                                "t3 = B (t1 - s0) with s1 = beta s1 + w1, <s1, t3> in one launch" if dual else
                                "csr_stream_kernel<EpiK2c> + csr_stream_kernel<EpiK3c> (launch plans of A and B "
                                "differ: two launches, timed together)"),
                     "achieved": k2_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": k2_gbs / HBM_PEAK_GBS,
                     "traffic": traffic["bytes"] if traffic else None,
                     "traffic_source": traffic["source"] if traffic else None, "traffic_note": traffic_note,
                     "algorithmic_bytes_per_launch": k2_bytes, "avg_launch_ms": k2_ms,
                     "bytes_accounting": "A and B: value stream 8 B per non-zero + column stream %d B per %d non-zeros "
                                         "(as stored), row pointers, x and y once; + s0, t0 read (A rows); + second "
                                         "gathered operand, s1, w1 read, s1 written (B rows)"
                                         % (a_info["index_bytes"], a_info["index_group"]),
                     "achieved_if_priced_as_int32_csr": k2_bytes_int32 / (k2_ms * 1e-3) / 1e9,
                     "frac_of_achievable_6290": k2_gbs / HBM_ACHIEVABLE_GBS,
                     "frac_of_stream_triad": k2_gbs / triad_gbs,
                     "timing": "HIP events around the kernel inside %d iterations of the running loop" % (probe_its - 8)},
        "cpu_baseline": cpu,
        "roofline_hdg_like": hdg,
        "roofline_mypre_a_gs": mypre,
        "secondary_configs": secondary,
        "valid": valid,
        "hbm_GBs": {"whole_iteration_algorithmic": iter_gbs, "stream_triad": triad_gbs,
                    "spmv_A_plain_in_loop_cache_state": spmv_gbs,
                    "spmv_A_plain_back_to_back_cache_flattered": a_info["algorithmic_bytes"] / (spmv_b2b_ms * 1e-3) / 1e9,
                    "spmv_AB_fused_C23": k2_gbs},
        "kernel_ms": {"C1_BT_preA": k1_ms, "C23_A_B": k2_ms, "C4_update": k4_ms, "sum_kernels": sums_ms,
                      "spmv_A_plain_in_loop_cache_state": spmv_ms, "spmv_A_plain_back_to_back": spmv_b2b_ms,
                      "triad_1.6GB": triad_ms},
        "launches_per_iteration": {"sums_folded_into_consumers": folds, "preA_applied_in_C1_epilogue": pre_in_c1},
        "bytes_per_iteration": iter_bytes,
        "parity": parity,
        "setup_s": {"assemble_host": t_asm, "upload_lanczos_initial_residual": t_setup},
    }
    emit(result_fd, out)


if __name__ == "__main__":
    main()
