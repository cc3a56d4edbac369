"""2-D driver, counterpart of templates/NavierStokesSIMPLE_test.py (nu = 1e-3, order 2, only the
Stokes initial solve runs, :23).  BASELINE.json config 3 scales it to ~1e6 DoF."""
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hipla.ngstd import SetHeapSize, TaskManager                      # noqa: E402
from templates.NavierStokesSIMPLE_iterative import NavierStokes, SyntheticMesh   # noqa: E402


def main(maxh=0.05, order=2, tol=1e-10):
    mesh = SyntheticMesh(maxh, dim=2)
    SetHeapSize(100 * 1000 * 1000)
    timestep = 0.001
    with TaskManager():
        navstokes = NavierStokes(mesh, nu=0.001, order=order, timestep=timestep, inflow="inlet", outflow="outlet",
                                 wall="cyl|wall", uin=None)
    navstokes.SolveInitial(iterative=True, tol=tol)
    print("iterations", navstokes.stokes_bpcg_iterations, "time", navstokes.stokes_bpcg_time)
    return navstokes


if __name__ == "__main__":
    main()
