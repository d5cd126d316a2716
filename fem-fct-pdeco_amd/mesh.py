"""Host-side mesh descriptor: the stand-in for the dolfin ``mesh`` / ``FunctionSpace V``
arguments of the reference's solvers (the mesh stays on the host; only its size,
extent and DoF numbering matter to the device code).

Replaces, on the structured right-diagonal square mesh,
  df.RectangleMesh(Point(a1,a1), Point(a2,a2), n, n) + FunctionSpace(mesh,'CG',1)
      (advection_solidbody_FCT_PDECO_finaltime.py:63-64, Schnak_FCT_PDECO_refactored.py:92-93)
  vertex_to_dof_map(V)        (advection_solidbody_FCT_PDECO_finaltime.py:101)
  find_node_neighbours        (helpers.py:271-307)
"""
from __future__ import annotations

import numpy as np


class SquareMeshP1:
    """``V`` of the reference: CG1 on an n_cells x n_cells right-diagonal mesh of [a1,a2]^2."""

    def __init__(self, a1: float, a2: float, n_cells: int):
        if n_cells < 1 or not a2 > a1:
            raise ValueError("need n_cells >= 1 and a2 > a1")
        self.a1, self.a2, self.n_cells = float(a1), float(a2), int(n_cells)
        self.N = self.n_cells + 1
        self.nodes = self.N * self.N
        self.h = (self.a2 - self.a1) / self.n_cells
        self._v2d = None

    def dim(self) -> int:  # V.dim()
        return self.nodes

    @property
    def vertex_to_dof(self) -> np.ndarray:
        """vertex_to_dof_map(V): vertex iy*N+ix -> rank of (ix-iy, iy), lexicographic."""
        if self._v2d is None:
            N = self.N
            iy, ix = np.divmod(np.arange(self.nodes), N)
            order = np.lexsort((iy, ix - iy))
            v2d = np.empty(self.nodes, dtype=np.int64)
            v2d[order] = np.arange(self.nodes)
            self._v2d = v2d
        return self._v2d

    def coordinates(self):
        """vertex coordinates (vertex order), as dolfin builds them: a1 + i*h."""
        iy, ix = np.divmod(np.arange(self.nodes), self.N)
        return self.a1 + ix * self.h, self.a1 + iy * self.h

    def dof_neighbors(self):
        """find_node_neighbours(mesh, nodes, vertex_to_dof): per DoF, the DoFs sharing a
        mesh edge, own index last (helpers.py:297-298)."""
        N = self.N
        v2d = self.vertex_to_dof
        out = [None] * self.nodes
        offs = ((-1, 0), (1, 0), (0, -1), (0, 1), (1, 1), (-1, -1))
        for v in range(self.nodes):
            iy, ix = divmod(v, N)
            nb = [int(v2d[(iy + dy) * N + ix + dx]) for dx, dy in offs
                  if 0 <= ix + dx < N and 0 <= iy + dy < N]
            nb.append(int(v2d[v]))
            out[int(v2d[v])] = nb
        return out

    def key(self):
        return (self.a1, self.a2, self.n_cells)


def reorder_vector_to_dof(vec, num_steps, nodes, vertex_to_dof):
    """helpers.py:13-39 (vectorised): vec_dof[n*nodes + v2d[i]] = vec[n*nodes + i]."""
    vec = np.asarray(vec, dtype=np.float64)
    out = np.zeros(vec.shape)
    v2d = np.asarray(vertex_to_dof, dtype=np.int64)
    a = vec.reshape(-1)[:num_steps * nodes].reshape(num_steps, nodes)
    o = out.reshape(-1)[:num_steps * nodes].reshape(num_steps, nodes)
    o[:, v2d] = a
    return out


def reorder_vector_from_dof(vec_dof, num_steps, nodes, vertex_to_dof):
    """helpers.py:41-67 (vectorised): vec[n*nodes + i] = vec_dof[n*nodes + v2d[i]]."""
    vec_dof = np.asarray(vec_dof, dtype=np.float64)
    out = np.zeros(vec_dof.shape)
    v2d = np.asarray(vertex_to_dof, dtype=np.int64)
    a = vec_dof.reshape(-1)[:num_steps * nodes].reshape(num_steps, nodes)
    o = out.reshape(-1)[:num_steps * nodes].reshape(num_steps, nodes)
    o[:] = a[:, v2d]
    return out


# aliases used by the stale scripts (advection_solidbody_FCT.py:121,153)
reorder_vector_to_dof_time = reorder_vector_to_dof
reorder_vector_from_dof_time = reorder_vector_from_dof
