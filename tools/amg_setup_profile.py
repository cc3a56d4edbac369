"""Where the AMG set-up time goes (GPU box):  python tools/amg_setup_profile.py [grid]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "navier-stokes-solver_amd"))


import hipla
from hipla import amg
from staggered_grid import mac_stokes


def main():
    grid = int(sys.argv[1]) if len(sys.argv) > 1 else 136
    s = mac_stokes(3, grid, 0.01)
    eng = hipla.get_engine()
    A = hipla.SparseMatrix.from_scipy(s.A)
    eng.synchronize()
    stamps = []

    def timed(name, fn):
        def wrapper(*a, **k):
            eng.synchronize()
            t = time.perf_counter()
            out = fn(*a, **k)
            eng.synchronize()
            stamps.append((name, time.perf_counter() - t))
            return out
        return wrapper

    for name in ("amg_aggregate", "amg_prolongator", "csr_spgemm", "csr_transpose", "csr_inverse_diagonal"):
        setattr(eng, name, timed(name, getattr(eng, name)))
    amg._coarsest = timed("coarsest_inverse(host)", amg._coarsest)
    for rep in range(2):
        stamps.clear()
        t = time.perf_counter()
        levels = amg.build_hierarchy(A)
        eng.synchronize()
        total = time.perf_counter() - t
    print("grid %d: levels %s, total %.3f s" % (grid, [lv["n"] for lv in levels], total))
    agg = {}
    for name, dt in stamps:
        agg[name] = agg.get(name, 0.0) + dt
    for name, dt in sorted(agg.items(), key=lambda kv: -kv[1]):
        print("  %-28s %.3f s" % (name, dt))
    print("  %-28s %.3f s" % ("other (host rng, wrappers)", total - sum(agg.values())))
    print("  per call:", ", ".join("%s %.0f ms" % (n.replace("csr_", "").replace("amg_", ""), 1e3 * d) for n, d in stamps))


if __name__ == "__main__":
    main()
