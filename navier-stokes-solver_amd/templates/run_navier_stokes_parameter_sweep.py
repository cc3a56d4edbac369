"""Parameter sweep with the reference's CSV schema
(templates/run_navier_stokes_parameter_sweep.py:44-70): mesh size x order x gauss_seidel ->
``iterations``, ``time`` (= `BramblePasciakCG`'s return, i.e. the iteration loop only)."""
import sys
import os

import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from templates.NavierStokesSIMPLE_iterative import NavierStokes, SyntheticMesh   # noqa: E402


def create_mesh(mesh_size, dim=2):
    return SyntheticMesh(mesh_size, dim=dim)


def create_nav_stokes(mesh, order, nu=0.001):
    return NavierStokes(mesh, nu=nu, order=order, timestep=1e-3, inflow="inlet", outflow="outlet",
                        wall="cyl|wall", uin=None)


def sweep(mesh_sizes, orders, gauss_seidel_enabled=(True, False), out="data.csv", dim=2, tol=1e-10, maxsteps=100000):
    frames = []
    for mesh_size in mesh_sizes:
        mesh = create_mesh(mesh_size, dim)
        for order in orders:
            navstokes = create_nav_stokes(mesh, order)
            for gauss_seidel in gauss_seidel_enabled:
                navstokes.gfu[:] = 0.0
                navstokes.gfup[:] = 0.0
                print("solving h = %g, p = %d, GS = %s" % (mesh_size, order, gauss_seidel))
                navstokes.SolveInitial(iterative=True, GS=gauss_seidel, tol=tol, maxsteps=maxsteps)
                frames.append(pd.DataFrame({'mesh_size': mesh_size, 'order': order,
                                            'iterations': navstokes.stokes_bpcg_iterations,
                                            'time': navstokes.stokes_bpcg_time,
                                            'gauss_seidel_enabled': gauss_seidel}, index=[0]))
    data = pd.concat(frames, ignore_index=True)
    if out:
        data.to_csv(out)
    return data


def sweep_reynolds(reynolds=(100, 400, 1000), mesh_size=1.0 / 32, order=1, dim=3, gauss_seidel=False, out="data_re.csv",
                   tol=1e-10, maxsteps=100000):
    """BASELINE.json config 5 restates the sweep over Reynolds numbers (nu = 1/Re) on one 3-D
    mesh: in the Stokes operator nu only rescales the velocity block
    (templates/NavierStokesSIMPLE_iterative.py:66,72), so this measures how the iteration count of
    the Bramble-Pasciak solve moves with it.  Same CSV columns plus ``reynolds``."""
    frames = []
    mesh = create_mesh(mesh_size, dim)
    for re in reynolds:
        navstokes = create_nav_stokes(mesh, order, nu=1.0 / re)
        navstokes.SolveInitial(iterative=True, GS=gauss_seidel, tol=tol, maxsteps=maxsteps)
        frames.append(pd.DataFrame({'mesh_size': mesh_size, 'order': order, 'reynolds': re,
                                    'iterations': navstokes.stokes_bpcg_iterations,
                                    'time': navstokes.stokes_bpcg_time,
                                    'gauss_seidel_enabled': gauss_seidel}, index=[0]))
    data = pd.concat(frames, ignore_index=True)
    if out:
        data.to_csv(out)
    return data


if __name__ == "__main__":
    sweep([2 ** -i for i in range(5, 1, -1)], range(3, 1, -1), (True, False))
