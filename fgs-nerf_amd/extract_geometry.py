"""SDF field extraction for mesh export (reference ``model/extract_geometry.py``).

``extract_fields`` evaluates ``query_func`` (for the SDF model: the trilinear lookup of ``-sdf``, model/nerf.py:1163) on a
``resolution``^3 lattice spanning the bounding box.  Same values and the same query granularity as the reference (at most
``N``^3 points per ``query_func`` call, model/extract_geometry.py:5-19), organised for the device: the lattice lives on
the accelerator, every block's result lands in one resident volume, and a single device->host copy ends the function
(the reference copies each block back as it goes).  ``extract_geometry`` hands the volume to PyMCubes' marching cubes
(third-party, CPU) when that package is installed; it is not part of this image, so only the field half is exercised by
the tests.
"""
from __future__ import annotations

import itertools

import numpy as np
import torch


def _axis(lo, hi, resolution, device):
    return torch.linspace(float(lo), float(hi), resolution, device=device)


def extract_fields(bound_min, bound_max, resolution, query_func, N=64):
    """float32 numpy volume [resolution]^3 of ``query_func`` over the lattice; ``query_func`` sees [n,3] points, n <= N^3."""
    device = bound_min.device if isinstance(bound_min, torch.Tensor) else torch.device('cpu')
    axes = [_axis(bound_min[a], bound_max[a], resolution, device) for a in range(3)]
    starts = range(0, resolution, N)
    field = torch.empty(resolution, resolution, resolution, dtype=torch.float32, device=device)
    with torch.no_grad():
        for i0, j0, k0 in itertools.product(starts, starts, starts):
            sub = [axes[0][i0:i0 + N], axes[1][j0:j0 + N], axes[2][k0:k0 + N]]
            shape = tuple(len(s) for s in sub)
            pts = torch.stack(torch.meshgrid(*sub, indexing='ij'), dim=-1).reshape(-1, 3)
            block = query_func(pts).reshape(shape)
            field[i0:i0 + shape[0], j0:j0 + shape[1], k0:k0 + shape[2]] = block.to(device=device, dtype=torch.float32)
    return field.cpu().numpy()


def extract_geometry(bound_min, bound_max, resolution, threshold, query_func, N=64):
    """model/extract_geometry.py:21-28: marching cubes on the extracted field, vertices mapped back to world space."""
    try:
        import mcubes
    except ImportError as e:  # PyMCubes is a third-party CPU dependency of the reference, absent from this image
        raise ImportError("extract_geometry needs PyMCubes (mcubes); extract_fields works without it") from e
    field = extract_fields(bound_min, bound_max, resolution, query_func, N)
    vertices, triangles = mcubes.marching_cubes(field, threshold)
    lo = np.asarray(torch.as_tensor(bound_min).detach().cpu(), dtype=np.float64)
    hi = np.asarray(torch.as_tensor(bound_max).detach().cpu(), dtype=np.float64)
    world = lo[None, :] + vertices * ((hi - lo) / (resolution - 1.0))[None, :]
    return world, triangles
