"""CPU oracle for the FEM-FCT forward/adjoint time-stepping path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU (NumPy/SciPy; no compiled part)
restatement of the reference algorithm (KarolinaBenkova/FEM-FCT-PDECO,
``helpers.py``).  It exists to *check* the HIP product path, never to be it:

  * only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
    ``bench.py`` may import, link or execute anything under ``oracle/``;
  * the product package (``fem-fct-pdeco_amd/``) never imports it and fails
    loudly when its HIP library is missing.

Parity status: PINNED, except the adjoint chemotaxis exp-forms (last item).
  * ``oracle.fct`` (FCT step, ChebSI, artificial diffusion, norms, cost
    functional) is checked against the reference's own functions imported in
    the build container (``tests/golden/make_golden.py`` generated the
    committed vectors ``tests/golden/fct_*.npz``).
  * ``oracle.mesh`` / ``oracle.assembly`` / ``oracle.traj.solve_chtxs_system``
    are checked against the real-FEniCS trajectory the reference ships
    (``Chtxs_data_dx0.025_dt0.001/chtxs_{m,f}_t0.01.csv`` re-saved as
    ``tests/golden/chtxs_fenics_traj.npz``).
  * NOT pinned (no reference output exists in the tree): the adjoint
    chemotaxis exp-forms (``helpers.py:1499-1500``) -- quadrature degree
    inferred from UFL's estimation rule.  See DESIGN.md.

Every function cites the reference ``file:line`` it restates.
"""

from . import mesh, assembly, fct, traj  # noqa: F401
