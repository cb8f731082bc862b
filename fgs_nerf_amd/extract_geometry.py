"""SDF field extraction for mesh export (reference ``model/extract_geometry.py``).

``extract_fields`` evaluates ``query_func`` (for the SDF model: the trilinear lookup of ``-sdf``, model/nerf.py:1163) on a
``resolution``^3 lattice spanning the bounding box.  Same values and the same query granularity as the reference (at most
``N``^3 points per ``query_func`` call, model/extract_geometry.py:5-19), organised for the device: the lattice lives on
the accelerator, every block's result lands in one resident volume, and a single device->host copy ends the function
(the reference copies each block back as it goes).

``marching_cubes`` replaces the reference's host call ``mcubes.marching_cubes(u, threshold)`` (PyMCubes, third party, CPU;
not part of this image) with three HIP passes over the resident field (csrc/mcubes.hip): same contract -- shared vertices
in index coordinates as float64 [V,3], triangles [T,3] -- with a generated case table (mc_tables.py).  PyMCubes' vertex
numbering and its choice among the valid triangulations of a cell are not reproduced (parity unpinned: the package is
absent and the reference holds no mesh fixture); the tests pin the surface itself: bit-exact against the CPU restatement
of the same convention, closed oriented 2-manifold on random fields, Euler characteristic / area / vertex distance on
analytic shapes.
"""
from __future__ import annotations

import itertools

import numpy as np
import torch


def _axis(lo, hi, resolution, device):
    return torch.linspace(float(lo), float(hi), resolution, device=device)


def extract_fields(bound_min, bound_max, resolution, query_func, N=64):
    """float32 numpy volume [resolution]^3 of ``query_func`` over the lattice; ``query_func`` sees [n,3] points, n <= N^3."""
    return extract_fields_device(bound_min, bound_max, resolution, query_func, N).cpu().numpy()


def extract_fields_device(bound_min, bound_max, resolution, query_func, N=64):
    """extract_fields without the final copy: the volume stays on the device of ``bound_min`` (for marching_cubes)."""
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
    return field


_TABLES = {}


def _device_tables(device):
    key = (device.type, device.index)
    if key not in _TABLES:
        from .mc_tables import tables
        tri, ntri = tables()
        _TABLES[key] = (torch.from_numpy(tri).to(device).contiguous(), torch.from_numpy(ntri).to(device))
    return _TABLES[key]


def marching_cubes_device(field: torch.Tensor, threshold: float):
    """(vertices float64 [V,3] in index coordinates, triangles int64 [T,3]) as CUDA tensors; field float32 [X,Y,Z] on CUDA."""
    from ._lib import call, check_input, lib, ptr, stream
    check_input(field, "field", torch.float32)
    if field.dim() != 3:
        raise RuntimeError("field must be [X, Y, Z]")
    X, Y, Z = field.shape
    dev = field.device
    tri, ntri = _device_tables(dev)
    nblk = int(lib().fgs_mc_num_blocks(X, Y, Z))
    vflags = torch.empty(X * Y * Z, dtype=torch.uint8, device=dev)
    counts = torch.empty(2, nblk, dtype=torch.int64, device=dev)
    offs = torch.empty(2, nblk + 1, dtype=torch.int64, device=dev)
    st = stream()
    call("fgs_mc_count", ptr(field), X, Y, Z, float(threshold), ptr(ntri), ptr(vflags), ptr(counts[0]), ptr(counts[1]), st)
    call("fgs_exclusive_scan_i64", ptr(counts[0]), nblk, ptr(offs[0]), st)
    call("fgs_exclusive_scan_i64", ptr(counts[1]), nblk, ptr(offs[1]), st)
    n_v, n_t = (int(x) for x in offs[:, nblk].tolist())           # the one host read: output sizes
    vertices = torch.empty(n_v, 3, dtype=torch.float64, device=dev)
    triangles = torch.empty(n_t, 3, dtype=torch.int64, device=dev)
    vbase = torch.empty(X * Y * Z, dtype=torch.int32, device=dev)    # uint32 ids
    call("fgs_mc_emit", ptr(field), X, Y, Z, float(threshold), ptr(tri), ptr(ntri), ptr(vflags), ptr(offs[0]), ptr(offs[1]),
         ptr(vbase), n_v, n_t, ptr(vertices), ptr(triangles), st)
    return vertices, triangles


def marching_cubes(field, threshold):
    """Drop-in for ``mcubes.marching_cubes(u, isovalue)``: numpy (vertices float64 [V,3], triangles [T,3]).  ``field`` may
    be a numpy volume (copied to the current CUDA device) or a CUDA tensor.  No CPU path: raises without the HIP library."""
    if not torch.is_tensor(field):
        field = torch.from_numpy(np.ascontiguousarray(field, dtype=np.float32))
    if not field.is_cuda:
        field = field.cuda()
    v, t = marching_cubes_device(field.float().contiguous(), threshold)
    return v.cpu().numpy(), t.cpu().numpy()


def extract_geometry(bound_min, bound_max, resolution, threshold, query_func, N=64):
    """model/extract_geometry.py:21-28: marching cubes on the extracted field, vertices mapped back to world space
    (float64 numpy, as the reference returns them)."""
    field = extract_fields_device(bound_min, bound_max, resolution, query_func, N)
    vertices, triangles = marching_cubes(field, threshold)
    lo = np.asarray(torch.as_tensor(bound_min).detach().cpu(), dtype=np.float32)
    hi = np.asarray(torch.as_tensor(bound_max).detach().cpu(), dtype=np.float32)
    world = vertices / (resolution - 1.0) * (hi - lo)[None, :] + lo[None, :]
    return world, triangles
