"""CPU restatement of the marching-cubes convention of fgs-nerf_amd/csrc/mcubes.hip.

TEST INFRASTRUCTURE ONLY -- imported by tests/ alone, never by the product.

PARITY UNPINNED: the reference calls PyMCubes (``mcubes.marching_cubes``, model/extract_geometry.py:24), a third-party
package that is not installed in this image, and holds no mesh fixture.  What is restated here is therefore the *contract*
of that call (vertices in index coordinates, float64, shared between triangles; linear interpolation
``(x2-x1)*(iso-f1)/(f2-f1)+x1`` along a crossed edge) plus the documented convention of the device implementation, written
independently of its table generator: plain loops over cells, contour segments found per cell at run time from six
hand-listed face rings, no lookup table.  Small grids only (pure Python).

Convention (same words as fgs-nerf_amd/mc_tables.py): corner flagged when ``field < iso``; per face, counter-clockwise seen
from outside, every maximal run of flagged corners is cut off by a segment from the edge where the run ends to the edge
where it starts; segments chain into loops v0..v(n-1) starting on the loop's smallest edge id, loops in order of that id;
a loop is split by the triangle (v_i, v_k, v_j) on its chain v_i..v_j with the smallest k for which neither new diagonal
joins two edges of one cell face (and for which both remaining chains can be finished the same way), recursively.  Vertex ids: rank in (lattice point
z-fastest, axis x<y<z) order.  Triangles: cell order, z fastest.
"""
from __future__ import annotations

import numpy as np

# corner id = dx + 2 dy + 4 dz; rings are counter-clockwise seen from outside the cell
_RINGS = (
    (4, 6, 2, 0),  # x = 0
    (1, 3, 7, 5),  # x = 1
    (1, 5, 4, 0),  # y = 0
    (2, 6, 7, 3),  # y = 1
    (2, 3, 1, 0),  # z = 0
    (4, 5, 7, 6),  # z = 1
)


def _edge(c0, c1):
    """(axis, lower-end corner id) of the cell edge between two adjacent corners."""
    axis = {1: 0, 2: 1, 4: 2}[c0 ^ c1]
    return axis, min(c0, c1)


def _edge_key(axis, corner):
    """Edge id 4*axis + offsets of the other two axes (increasing axis order) of the lower-end corner."""
    d = (corner & 1, (corner >> 1) & 1, (corner >> 2) & 1)
    others = [a for a in range(3) if a != axis]
    return 4 * axis + d[others[0]] + 2 * d[others[1]]


def _coplanar(e0, e1):
    """Both edges (axis, lower-end corner) lie in one face of the cell: they agree on a coordinate that is fixed for both."""
    (a0, c0), (a1, c1) = e0, e1
    return any(ax != a0 and ax != a1 and ((c0 >> ax) & 1) == ((c1 >> ax) & 1) for ax in range(3))


def _split(loop, i, j):
    if j - i < 2:
        return []
    for k in range(i + 1, j):
        if k - i > 1 and _coplanar(loop[i], loop[k]):
            continue
        if j - k > 1 and _coplanar(loop[k], loop[j]):
            continue
        a = _split(loop, i, k)
        b = _split(loop, k, j) if a is not None else None
        if b is not None:
            return [(loop[i], loop[k], loop[j])] + a + b
    return None


def cell_triangles(case):
    """Triangles of one cell as triples of (axis, lower-end corner) edges."""
    nxt = {}
    for ring in _RINGS:
        b = [(case >> c) & 1 for c in ring]
        if sum(b) in (0, 4):
            continue
        for i in range(4):
            if b[i] and not b[i - 1]:
                j = i
                while b[(j + 1) % 4]:
                    j = (j + 1) % 4
                nxt[_edge(ring[j], ring[(j + 1) % 4])] = _edge(ring[i - 1], ring[i])
    tris, todo = [], set(nxt)
    while todo:
        start = min(todo, key=lambda e: _edge_key(*e))
        loop, e = [], start
        while True:
            loop.append(e)
            todo.discard(e)
            e = nxt[e]
            if e == start:
                break
        tris += _split(loop, 0, len(loop) - 1)
    return tris


def marching_cubes(field, iso):
    """(vertices float64 [V,3] in index coordinates, triangles int64 [T,3])."""
    f = np.asarray(field, dtype=np.float32)
    iso32 = np.float32(iso)
    X, Y, Z = f.shape
    flag = f < iso32
    lin = lambda i, j, k: (i * Y + j) * Z + k
    step = ((1, 0, 0), (0, 1, 0), (0, 0, 1))
    vid, verts = {}, []
    for i in range(X):
        for j in range(Y):
            for k in range(Z):
                for a in range(3):
                    n = (i + step[a][0], j + step[a][1], k + step[a][2])
                    if n[0] >= X or n[1] >= Y or n[2] >= Z or flag[n] == flag[i, j, k]:
                        continue
                    f1, f2, level = float(f[i, j, k]), float(f[n]), float(iso32)
                    p = [float(i), float(j), float(k)]
                    p[a] = (p[a] + (p[a] + 1.0)) / 2.0 if f2 == f1 else (level - f1) / (f2 - f1) + p[a]
                    vid[(lin(i, j, k), a)] = len(verts)
                    verts.append(p)
    tris = []
    cache = {}
    for i in range(X - 1):
        for j in range(Y - 1):
            for k in range(Z - 1):
                case = 0
                for c in range(8):
                    if flag[i + (c & 1), j + ((c >> 1) & 1), k + (c >> 2)]:
                        case |= 1 << c
                if case not in cache:
                    cache[case] = cell_triangles(case)
                for t in cache[case]:
                    tris.append([vid[(lin(i + (c & 1), j + ((c >> 1) & 1), k + (c >> 2)), a)] for a, c in t])
    return (np.asarray(verts, dtype=np.float64).reshape(-1, 3), np.asarray(tris, dtype=np.int64).reshape(-1, 3))


def mesh_report(vertices, triangles):
    """Topology / geometry summary used by the known-answer tests: every undirected edge must be used by exactly two
    triangles, once in each direction (closed, consistently oriented 2-manifold)."""
    t = np.asarray(triangles, dtype=np.int64)
    v = np.asarray(vertices, dtype=np.float64)
    d = np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]])
    V = int(v.shape[0])
    directed = d[:, 0] * V + d[:, 1]
    und = np.minimum(d[:, 0], d[:, 1]) * V + np.maximum(d[:, 0], d[:, 1])
    _, cnt = np.unique(und, return_counts=True)
    closed = bool((cnt == 2).all()) and len(np.unique(directed)) == len(directed) and \
        bool(np.isin(d[:, 1] * V + d[:, 0], directed).all())
    a, b, c = v[t[:, 0]], v[t[:, 1]], v[t[:, 2]]
    cr = np.cross(b - a, c - a)
    return dict(closed_oriented=closed, euler=V - len(cnt) + len(t), area=float(0.5 * np.linalg.norm(cr, axis=1).sum()),
                signed_volume=float((a * cr).sum() / 6.0), used_vertices=int(len(np.unique(t))))
