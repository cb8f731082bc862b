"""Marching-cubes case table, generated (not transcribed) from one face rule.

The reference hands the extracted SDF field to PyMCubes (``mcubes.marching_cubes``, model/extract_geometry.py:24), a
third-party CPU package that is not part of this image.  The device implementation (csrc/mcubes.hip) needs a 256-entry
triangle table; instead of transcribing the classic one, this module derives a table from a single rule applied to the
six faces of a cell, which makes the result watertight by construction (the two cells sharing a face see the same four
corner signs and therefore draw the same contour segments on it):

* a corner is *flagged* when ``field < iso``;
* on every face, walking its boundary counter-clockwise as seen from outside the cell, each maximal run of flagged
  corners is cut off by one directed segment from the edge where the run ends to the edge where it starts (on a face with
  two diagonal flagged corners this isolates each of them -- the ambiguous case is always resolved the same way);
* the directed segments of the six faces chain into closed loops v0..v(n-1), v0 on the loop's smallest edge id; a loop is
  triangulated without any diagonal that lies in a face of the cell (two loop vertices on edges of one face): such a
  diagonal could coincide with a diagonal drawn by the neighbouring cell and make the edge non-manifold.  Among the
  triangulations the first one in this order is taken: the chain v_i..v_j is closed by the triangle (v_i, v_k, v_j) with
  the smallest admissible k, then the chains v_i..v_k and v_k..v_j are treated the same way (triangles in that order).

Triangle normals (right-hand rule) point towards the flagged side, i.e. towards *decreasing* field values: outward for the
reference's query ``-sdf`` at threshold 0 (model/nerf.py:1157-1170).

Numbering: corner ``c = dx + 2 dy + 4 dz``; edge ``e = 4 a + o`` runs along axis ``a`` from the corner whose other two
offsets (in increasing axis order) are ``o & 1`` and ``o >> 1``; the vertex on it is owned by the lattice point at the
edge's lower end.
"""
from __future__ import annotations

import functools

import numpy as np


def _edge_id(q0, q1):
    """Edge between two corners (offset triples) that differ along exactly one axis."""
    diff = [i for i in range(3) if q0[i] != q1[i]]
    assert len(diff) == 1
    a = diff[0]
    others = [i for i in range(3) if i != a]
    return 4 * a + q0[others[0]] + 2 * q0[others[1]]


def _faces():
    """Six faces as 4 corner-offset triples in counter-clockwise order seen from outside the cell."""
    out = []
    for a in range(3):
        b1, b2 = (a + 1) % 3, (a + 2) % 3          # (a, b1, b2) is a cyclic (right-handed) permutation of (x, y, z)
        for s in (0, 1):
            ring = []
            for u, v in ((0, 0), (1, 0), (1, 1), (0, 1)):
                q = [0, 0, 0]
                q[a], q[b1], q[b2] = s, u, v
                ring.append(tuple(q))
            out.append(ring if s == 1 else ring[::-1])
    return out


def _share_face(e0, e1):
    """Do two cell edges lie in a common face?  Edge e = 4a + o occupies coordinates: a free, the other two fixed."""
    def fixed(e):
        a, o = e >> 2, e & 3
        others = [i for i in range(3) if i != a]
        return {others[0]: o & 1, others[1]: o >> 1}
    f0, f1 = fixed(e0), fixed(e1)
    return any(ax in f1 and f1[ax] == val for ax, val in f0.items())


def _triangulate(loop, i, j):
    """First admissible triangulation of the chain loop[i..j] (see the module docstring), or None."""
    if j - i < 2:
        return []
    for k in range(i + 1, j):
        if (k > i + 1 and _share_face(loop[i], loop[k])) or (j > k + 1 and _share_face(loop[k], loop[j])):
            continue
        left = _triangulate(loop, i, k)
        if left is None:
            continue
        right = _triangulate(loop, k, j)
        if right is None:
            continue
        return [(loop[i], loop[k], loop[j])] + left + right
    return None


@functools.lru_cache(maxsize=None)
def tables():
    """(tri_table int8 [256, 16] of edge ids, -1 padded; n_tri uint8 [256])."""
    faces = _faces()
    tri = -np.ones((256, 16), dtype=np.int8)
    ntri = np.zeros(256, dtype=np.uint8)
    for case in range(256):
        flagged = lambda q: (case >> (q[0] + 2 * q[1] + 4 * q[2])) & 1
        nxt = {}
        for ring in faces:
            b = [flagged(q) for q in ring]
            if sum(b) in (0, 4):
                continue
            for i in range(4):
                if b[i] and not b[(i - 1) % 4]:                     # a run of flagged corners starts at i
                    j = i
                    while b[(j + 1) % 4]:
                        j = (j + 1) % 4                              # ... and ends at j
                    e_entry = _edge_id(ring[(i - 1) % 4], ring[i])
                    e_exit = _edge_id(ring[j], ring[(j + 1) % 4])
                    assert e_exit not in nxt
                    nxt[e_exit] = e_entry
        assert sorted(nxt) == sorted(nxt.values())                   # one segment in, one out per crossed edge
        todo, tris = set(nxt), []
        while todo:
            start = min(todo)
            loop, e = [], start
            while True:
                loop.append(e)
                todo.remove(e)
                e = nxt[e]
                if e == start:
                    break
            assert len(loop) >= 3
            part = _triangulate(loop, 0, len(loop) - 1)
            assert part is not None, (case, loop)
            tris += part
        assert len(tris) <= 5
        ntri[case] = len(tris)
        for t, (e0, e1, e2) in enumerate(tris):
            tri[case, 3 * t:3 * t + 3] = (e0, e1, e2)
    return tri, ntri
