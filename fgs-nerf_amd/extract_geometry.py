"""SDF field extraction for mesh export (reference ``model/extract_geometry.py``).

``extract_fields`` evaluates ``query_func`` (for the SDF model: the trilinear lookup of ``-sdf``, model/nerf.py:1163) on a
``resolution``^3 lattice in 64^3 blocks, exactly as model/extract_geometry.py:5-19 does; the lookups run on the HIP
trilerp kernel.  ``extract_geometry`` hands the volume to PyMCubes' marching cubes (third-party, CPU) when that package
is installed; it is not part of this image, so only the field half is exercised by the tests.
"""
from __future__ import annotations

import numpy as np
import torch


def extract_fields(bound_min, bound_max, resolution, query_func, N=64):
    dev = bound_min.device if isinstance(bound_min, torch.Tensor) else 'cpu'
    X = torch.linspace(float(bound_min[0]), float(bound_max[0]), resolution, device=dev).split(N)
    Y = torch.linspace(float(bound_min[1]), float(bound_max[1]), resolution, device=dev).split(N)
    Z = torch.linspace(float(bound_min[2]), float(bound_max[2]), resolution, device=dev).split(N)
    u = np.zeros([resolution, resolution, resolution], dtype=np.float32)
    with torch.no_grad():
        for xi, xs in enumerate(X):
            for yi, ys in enumerate(Y):
                for zi, zs in enumerate(Z):
                    xx, yy, zz = torch.meshgrid(xs, ys, zs, indexing='ij')
                    pts = torch.cat([xx.reshape(-1, 1), yy.reshape(-1, 1), zz.reshape(-1, 1)], dim=-1)
                    val = query_func(pts).reshape(len(xs), len(ys), len(zs)).detach().cpu().numpy()
                    u[xi * N: xi * N + len(xs), yi * N: yi * N + len(ys), zi * N: zi * N + len(zs)] = val
    return u


def extract_geometry(bound_min, bound_max, resolution, threshold, query_func, N=64):
    """model/extract_geometry.py:21-28: marching cubes on the extracted field, vertices mapped back to world space."""
    try:
        import mcubes
    except ImportError as e:  # PyMCubes is a third-party CPU dependency of the reference, absent from this image
        raise ImportError("extract_geometry needs PyMCubes (mcubes); extract_fields works without it") from e
    u = extract_fields(bound_min, bound_max, resolution, query_func, N)
    vertices, triangles = mcubes.marching_cubes(u, threshold)
    b_max_np = np.asarray(bound_max.detach().cpu() if isinstance(bound_max, torch.Tensor) else bound_max)
    b_min_np = np.asarray(bound_min.detach().cpu() if isinstance(bound_min, torch.Tensor) else bound_min)
    vertices = vertices / (resolution - 1.0) * (b_max_np - b_min_np)[None, :] + b_min_np[None, :]
    return vertices, triangles
