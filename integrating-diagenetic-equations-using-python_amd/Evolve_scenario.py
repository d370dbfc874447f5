"""Driver - host-side mirror of ``marlpde/Evolve_scenario.py``.

``integrate_equations(solver_parms, tracker_parms, pde_parms)`` takes the same three dictionaries as the
reference (Evolve_scenario.py:19) and returns the same tuple
``(last (5, N) profile, covered_time [years], depths, Xstar, store_folder)`` (:183).

* ``solver_parms["backend"] == "hip"`` and ``method == "RK45"``: the whole adaptive loop
  (scipy-exact Dormand-Prince controller, the seven monitors with root finding, ``t_eval`` by dense
  output) runs on the GPU via ``marl_integrate_rk45``.
* ``backend == "hip"`` and ``method == "Radau"`` (the reference's default, parameters.py:213): scipy's Radau IIA step
  logic restated natively, with the RHS, the finite-difference Jacobian (the reference's 27-diagonal pattern), the
  block-tridiagonal LU factorisations and every vector operation on the GPU via ``marl_integrate_radau`` - no scipy in the loop.
* ``backend == "hip"`` and ``method == "BDF"``: scipy's BDF step logic restated natively on the same Jacobian /
  factorisation kernels via ``marl_integrate_bdf`` - no scipy in the loop either.
* ``backend == "hip"`` and any other scipy method (RK23, DOP853, LSODA with the reference's ``lband = uband = 1``; or any
  method with ``solver_parms["scipy_driver"] = True``): scipy's ``solve_ivp`` drives, exactly as in the reference
  (:104-109), with the HIP RHS and HIP monitors as callables
  (``test_every_other_method_of_the_reference_solver_runs_through_scipy_on_the_hip_rhs``).
* other backends: rejected - this package has no CPU implementation.

Results are stored like the reference's (:156-178) with the same dataset names, as ``.npz`` (always) and
HDF5 (when h5py is importable).
"""
import os
import time
from datetime import datetime

import numpy as np

from .LHeureux_model import DepthGrid, LMAHeureuxPorosityDiff

EVENT_TEXT = (
    "any field at any depth crossed zero", "CA at any depth crossed zero", "CC at any depth crossed zero",
    "CA + CC at any depth crossed one", "the porosity at any depth crossed one", "U at any depth crossed zero",
    "W at any depth crossed zero")


class _NoBar:
    n = 0

    def update(self, k):
        self.n += k


def integrate_equations(solver_parms, tracker_parms, pde_parms, results_root="../Results/", verbose=True, device=0):
    solver_parms = dict(solver_parms)
    Xstar, Tstar = pde_parms["Xstar"], pde_parms["Tstar"]
    N = int(pde_parms["N"])
    depths = DepthGrid(pde_parms["max_depth"] / Xstar, N)

    backend = solver_parms.pop("backend", "hip")
    if backend != "hip":
        raise ValueError(f"backend {backend!r} is not available here: only 'hip' (MI355X kernels) is implemented; "
                         "use the reference package for its 'numba'/'numpy' CPU backends")
    eq = LMAHeureuxPorosityDiff.from_scenario(pde_parms, device=device)
    y0 = eq.get_state(pde_parms["CAIni"], pde_parms["CCIni"], pde_parms["cCaIni"], pde_parms["cCO3Ini"],
                      pde_parms["PhiIni"]).ravel()

    method = solver_parms.pop("method", "RK45")
    t_span = tuple(solver_parms.pop("t_span", (0, 1)))
    t_eval = tracker_parms.get("t_eval")
    no_progress_updates = tracker_parms.get("no_progress_updates", 100_000)
    start = time.time()
    scipy_driver = bool(solver_parms.pop("scipy_driver", False))
    if method == "RK45" and not scipy_driver:
        res = eq.integrate_rk45(y0, t_span, solver_parms["first_step"], solver_parms["rtol"], solver_parms["atol"],
                                t_eval=t_eval)
        t_out, y_out, t_events = res.t, res.y, res.t_events
        nfev, njev, nlu, status, message = res.nfev, 0, 0, res.status, res.message
        if status not in (0, -1):
            status = -1
        covered = Tstar * (t_span[1] if status == 0 else res.t_reached)
    elif method in ("Radau", "BDF") and not scipy_driver:
        # the reference's default: the whole implicit loop on the GPU (marl_integrate_radau).  The Jacobian pattern is the
        # reference's (parameters.py:150-199); when the caller's jac_sparsity has the matching shape its scipy column grouping
        # is used, otherwise a structured colouring - the Jacobian entries are the same either way.
        groups = None
        sp = solver_parms.get("jac_sparsity")
        if sp is not None and getattr(sp, "shape", None) == (5 * N, 5 * N):
            from scipy.optimize._numdiff import group_columns
            from scipy.sparse import csc_matrix
            groups = group_columns(csc_matrix(sp))
        integrate = eq.integrate_radau if method == "Radau" else eq.integrate_bdf   # (marl_integrate_bdf: the same machinery, scipy's BDF step logic)
        res = integrate(y0, t_span, solver_parms["first_step"], solver_parms["rtol"], solver_parms["atol"], t_eval=t_eval, groups=groups)
        t_out, y_out, t_events = res.t, res.y, res.t_events
        nfev, njev, nlu, status, message = res.nfev, res.njev, res.nlu, res.status, res.message
        if status not in (0, -1):
            status = -1
        covered = Tstar * (t_span[1] if status == 0 else res.t_reached)
    else:
        from scipy.integrate import solve_ivp
        # the reference forwards every remaining Solver key to solve_ivp; keep only what the method takes
        drop = {"jac_sparsity"} if method == "LSODA" else {"lband", "uband"}
        if method not in ("Radau", "BDF", "LSODA"):
            drop |= {"jac_sparsity"}
        opts = {k: v for k, v in solver_parms.items() if k not in drop and v is not None}
        bar = _NoBar()
        args = [bar, (t_span[1] - t_span[0]) / no_progress_updates, t_span[0]]
        sol = solve_ivp(eq.fun, t_span, y0, method=method, t_eval=t_eval,
                        events=[eq.zeros, eq.zeros_CA, eq.zeros_CC, eq.ones_CA_plus_CC, eq.ones_Phi, eq.zeros_U,
                                eq.zeros_W], args=args, **opts)
        t_out, y_out, t_events = sol.t, sol.y, sol.t_events
        nfev, njev, nlu, status, message = sol.nfev, sol.njev, sol.nlu, sol.status, sol.message
        covered = Tstar * t_span[1] if status == 0 else bar.n * Tstar * t_span[1] / no_progress_updates
    wall = time.time() - start

    if verbose:
        print(f"rhs evaluations {nfev}, Jacobian evaluations {njev}, LU decompositions {nlu}, status {status}")
        for text, te in zip(EVENT_TEXT, t_events):
            print(f"times [years] at which {text}: " + ", ".join(f"{Tstar * x:.2f}" for x in te))
        print(f"solver message: {message}\nwall time {wall:.2e} s")

    field_solutions = np.asarray(y_out).reshape(5, N, -1)
    store_folder = None
    if results_root is not None:
        store_folder = os.path.join(results_root, datetime.now().strftime("%d_%m_%Y_%H_%M_%S") + "/")
        os.makedirs(store_folder, exist_ok=True)
        meta = {k: v for k, v in (solver_parms | {"method": method, "backend": backend, "t_span": t_span}
                                  | tracker_parms | pde_parms).items() if k != "jac_sparsity"}
        arrays = {"solutions": field_solutions, "times": np.asarray(t_out)}
        arrays |= {f"event_{i}": np.asarray(te) for i, te in enumerate(t_events)}
        np.savez(store_folder + "LMAHeureuxPorosityDiff.npz", **arrays,
                 **{"attr_" + k: np.asarray(v) for k, v in meta.items() if v is not None})
        try:
            import h5py
        except ImportError:
            h5py = None
        if h5py is not None:
            with h5py.File(store_folder + "LMAHeureuxPorosityDiff.hdf5", "w") as stored:
                for k, v in arrays.items():
                    stored.create_dataset(k, data=v)
                stored.attrs.update({k: v for k, v in meta.items() if v is not None})
    eq.close()
    return field_solutions[:, :, -1], covered, depths, Xstar, store_folder


def Plot_results(last_field_sol, covered_time, depths, Xstar, store_folder):
    """Final profiles against depth (reference :185-205); needs matplotlib."""
    import matplotlib
    matplotlib.use("AGG")
    import matplotlib.pyplot as plt
    fig, ax = plt.subplots()
    fig.suptitle(f"Distributions after {covered_time:.2e} years")
    x_cm = depths.axes_coords[0] * Xstar
    for row, marker, label in zip(last_field_sol, "v^><o", ("CA", "CC", "cCa", "cCO3", "Phi")):
        ax.plot(x_cm, row, marker, ms=5, label=label)
    ax.set_xlabel("Depth (cm)")
    ax.set_ylabel("Compositions and concentrations (dimensionless)")
    ax.legend(loc="upper right")
    fig.savefig((store_folder or "./") + "Final_distributions.pdf", bbox_inches="tight")


if __name__ == "__main__":
    from dataclasses import asdict, replace

    from .parameters import Map_Scenario, Solver, Tracker
    Plot_results(*integrate_equations(asdict(replace(Solver(), method="RK45")), asdict(Tracker()), asdict(Map_Scenario())))
