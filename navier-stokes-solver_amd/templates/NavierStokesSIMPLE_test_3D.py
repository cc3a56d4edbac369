"""3-D driver, counterpart of templates/NavierStokesSIMPLE_test_3D.py (maxh = 0.1, order 2,
nu = 1e-3; the solve runs outside the TaskManager block, :22-28).  BASELINE.json config 4 scales
it to ~1e7 DoF, which `bench.py` times."""
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hipla.ngstd import SetHeapSize, TaskManager                      # noqa: E402
from templates.NavierStokesSIMPLE_iterative import NavierStokes, SyntheticMesh   # noqa: E402


def main(maxh=0.1, order=2, tol=1e-10):
    mesh = SyntheticMesh(maxh, dim=3)
    SetHeapSize(100 * 1000 * 1000)
    timestep = 0.002
    with TaskManager():
        navstokes = NavierStokes(mesh, nu=0.001, order=order, timestep=timestep, inflow="inlet", outflow="outlet",
                                 wall="wall|cyl", uin=None)
    navstokes.SolveInitial(iterative=True, tol=tol)
    print("iterations", navstokes.stokes_bpcg_iterations, "time", navstokes.stokes_bpcg_time)
    return navstokes


if __name__ == "__main__":
    main()
