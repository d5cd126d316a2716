"""Structured right-diagonal P1 mesh of a square, and the FEniCS DoF numbering.

TEST INFRASTRUCTURE (see oracle/__init__.py).

Restates what the reference obtains from dolfin:
  * ``df.RectangleMesh(Point(a1,a1), Point(a2,a2), n, n)`` with the default
    diagonal "right" (e.g. advection_solidbody_FCT_PDECO_finaltime.py:63,
    Schnak_FCT_PDECO_refactored.py:92-93),
  * ``vertex_to_dof_map(V)`` for ``FunctionSpace(mesh,'CG',1)``
    (advection_solidbody_FCT_PDECO_finaltime.py:101),
  * ``find_node_neighbours`` (helpers.py:271-307).
"""
from __future__ import annotations

import numpy as np


class SquareMesh:
    """n x n cells, N = n+1 vertices per side, vertex (ix,iy) -> iy*N+ix."""

    def __init__(self, a1: float, a2: float, n_cells: int):
        self.a1, self.a2, self.n_cells = float(a1), float(a2), int(n_cells)
        N = self.N = n_cells + 1
        self.nodes = N * N
        self.h = (self.a2 - self.a1) / n_cells
        ix, iy = np.meshgrid(np.arange(N), np.arange(N))  # row-major: iy slow
        self.ix = ix.reshape(-1)
        self.iy = iy.reshape(-1)
        # dolfin builds vertex coordinates as a + i*h
        self.x = self.a1 + self.ix * self.h
        self.y = self.a1 + self.iy * self.h
        # cells: (ix,iy) -> triangles (v0,v1,v3), (v0,v2,v3)
        cx, cy = np.meshgrid(np.arange(n_cells), np.arange(n_cells))
        cx = cx.reshape(-1)
        cy = cy.reshape(-1)
        v0 = cy * N + cx
        v1 = v0 + 1
        v2 = v0 + N
        v3 = v0 + N + 1
        tri = np.empty((2 * n_cells * n_cells, 3), dtype=np.int64)
        tri[0::2] = np.stack([v0, v1, v3], axis=1)
        tri[1::2] = np.stack([v0, v2, v3], axis=1)
        self.cells = tri
        self.vertex_to_dof = fenics_vertex_to_dof(N)
        self.dof_to_vertex = np.empty_like(self.vertex_to_dof)
        self.dof_to_vertex[self.vertex_to_dof] = np.arange(self.nodes)

    # -- neighbour lists -------------------------------------------------
    def vertex_neighbors(self):
        """Vertices sharing an edge with each vertex, then the vertex itself
        (helpers.py:291-300: edge neighbours, own index appended last)."""
        N = self.N
        offs = [(-1, 0), (1, 0), (0, -1), (0, 1), (1, 1), (-1, -1)]
        out = []
        for v in range(self.nodes):
            ix, iy = v % N, v // N
            nb = []
            for dx, dy in offs:
                jx, jy = ix + dx, iy + dy
                if 0 <= jx < N and 0 <= jy < N:
                    nb.append(jy * N + jx)
            nb.append(v)
            out.append(nb)
        return out

    def dof_neighbors(self):
        """helpers.py:302-307: neighbour lists renumbered into DoF order."""
        v2d = self.vertex_to_dof
        vn = self.vertex_neighbors()
        out = [None] * self.nodes
        for v in range(self.nodes):
            out[int(v2d[v])] = [int(v2d[w]) for w in vn[v]]
        return out


def fenics_vertex_to_dof(N: int) -> np.ndarray:
    """vertex_to_dof_map of CG1 on the N x N right-diagonal RectangleMesh.

    Recovered from the reference's real-FEniCS data:
    ``vertex_to_dof[iy*N+ix] = rank of (ix-iy, iy)`` in lexicographic order.
    Pinned by tests/test_oracle_golden.py::test_fenics_trajectory (frame 0 of
    the shipped trajectory equals the seeded IC only under this permutation).
    """
    ix, iy = np.meshgrid(np.arange(N), np.arange(N))
    ix = ix.reshape(-1)
    iy = iy.reshape(-1)
    order = np.lexsort((iy, ix - iy))  # primary key ix-iy, secondary iy
    v2d = np.empty(N * N, dtype=np.int64)
    v2d[order] = np.arange(N * N)
    return v2d


def reorder_vector_to_dof(vec, num_steps, nodes, vertex_to_dof):
    """helpers.py:13-39: vec_dof[n*nodes + v2d[i]] = vec[n*nodes + i]."""
    vec = np.asarray(vec, dtype=np.float64)
    out = np.zeros(vec.shape)
    v2d = np.asarray(vertex_to_dof, dtype=np.int64)
    for n in range(num_steps):
        out[n * nodes + v2d] = vec[n * nodes:(n + 1) * nodes]
    return out


def reorder_vector_from_dof(vec_dof, num_steps, nodes, vertex_to_dof):
    """helpers.py:41-67: vec[n*nodes + i] = vec_dof[n*nodes + v2d[i]]."""
    vec_dof = np.asarray(vec_dof, dtype=np.float64)
    out = np.zeros(vec_dof.shape)
    v2d = np.asarray(vertex_to_dof, dtype=np.int64)
    for n in range(num_steps):
        out[n * nodes:(n + 1) * nodes] = vec_dof[n * nodes + v2d]
    return out
