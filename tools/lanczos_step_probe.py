#!/usr/bin/env python3
"""Where a step of the device-resident Lanczos goes at the small BASELINE sizes: N steps enqueued in ONE call of
nss_lanczos_iterate -- host time to enqueue them, device time between two events around them -- for the two-launch
and the five-launch form.   python tools/lanczos_step_probe.py [cfg2 cfg3]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "navier-stokes-solver_amd")):
    sys.path.insert(0, p)
import torch

import hipla
from hipla import eigen, fused
from staggered_grid import mac_stokes

CASES = {"cfg2": (2, 183), "cfg3": (2, 577), "cfg4": (3, 136)}
eng = hipla.get_engine()
for name in (sys.argv[1:] or ["cfg2", "cfg3"]):
    dim, n = CASES[name]
    s = mac_stokes(dim, n, 0.01)
    A = hipla.SparseMatrix.from_scipy(s.A)
    pre = hipla.BlockJacobi(A, s.line_blocks(3))
    pa = fused.native_velocity_pre(pre)
    for mode, label in ((-1, "two-launch (by size)"), (0, "five-launch")):
        eng.lib.nss_lanczos_fold_mode(mode)
        st = eigen._LanczosState.get()()
        st.A, st.pre_bjac, st.pre_scale, st.n = A.handle.ptr, pa["bjac"].handle.ptr, float(pa["scale"]), s.n_u
        vecs = [eng.zeros(s.n_u) for _ in range(6)]
        vecs[0].copy_(torch.from_numpy(eigen.lanczos_start_values(0, s.n_u)))
        for i in range(3):
            st.v[i] = vecs[i].data_ptr()
        st.z[0], st.z[1], st.p = vecs[3].data_ptr(), vecs[4].data_ptr(), vecs[5].data_ptr()
        na, nb = C.c_int64(), C.c_int64()
        eng._check(eng.lib.nss_lanczos_workspace(C.byref(st), C.byref(na), C.byref(nb)))
        pa_, pb_ = eng.zeros(max(1, na.value)), eng.zeros(max(1, nb.value))
        scal, ctrl, hist = eng.zeros(8), torch.zeros(4, dtype=torch.int32, device=eng.device), eng.zeros(2 * 1000)
        st.partials_a, st.partials_b, st.scal, st.ctrl, st.hist = (pa_.data_ptr(), pb_.data_ptr(), scal.data_ptr(),
                                                                    ctrl.data_ptr(), hist.data_ptr())
        eng._check(eng.lib.nss_lanczos_start(C.byref(st), eng.stream))
        eng._check(eng.lib.nss_lanczos_iterate(C.byref(st), 0, 50, eng.stream))
        torch.cuda.synchronize()
        N = 400
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        t0 = time.perf_counter()
        eng._check(eng.lib.nss_lanczos_iterate(C.byref(st), 50, 50 + N, eng.stream))
        t_host = time.perf_counter() - t0
        e1.record()
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print("%s %-22s: host enqueue %.1f us/step, device %.1f us/step, wall %.1f us/step (stop flag %d)"
              % (name, label, 1e6 * t_host / N, 1e3 * e0.elapsed_time(e1) / N, 1e6 * t_all / N, int(ctrl[0])))
    eng.lib.nss_lanczos_fold_mode(-1)
