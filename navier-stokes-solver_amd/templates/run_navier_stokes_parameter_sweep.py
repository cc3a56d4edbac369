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


if __name__ == "__main__":
    sweep([2 ** -i for i in range(5, 1, -1)], range(3, 1, -1), (True, False))
