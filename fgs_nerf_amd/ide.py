"""Integrated directional encoding (IDE) of Ref-NeRF, the reference's ``generate_ide_fn`` (model/utils.py:515-574).

The reference builds ``self.integrated_dir_enc = generate_ide_fn(sh_max_level)`` in ``nerf.__init__`` (model/nerf.py:179)
and never calls it; BASELINE config 3 (smart_car) names the encoding, so it exists here as an optional extra with the same
factory signature.  The coefficient tables follow model/utils.py:168-210 (``np.math`` is gone from numpy 2: ``math``);
evaluation is one HIP kernel forward and one backward (csrc/ide.hip) behind an autograd.Function.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from ._lib import call, check_input, ptr, stream


def generalized_binomial_coeff(a, k):
    return np.prod(a - np.arange(k)) / math.factorial(k)


def assoc_legendre_coeff(l, m, k):
    """Coefficient of cos^k(theta) sin^m(theta) in P_l^m(cos theta) (model/utils.py:173-189)."""
    return ((-1) ** m * 2 ** l * math.factorial(l) / math.factorial(k) / math.factorial(l - k - m) *
            generalized_binomial_coeff(0.5 * (l + k + m - 1.0), l))


def sph_harm_coeff(l, m, k):
    return (np.sqrt((2.0 * l + 1.0) * math.factorial(l - m) / (4.0 * np.pi * math.factorial(l + m))) *
            assoc_legendre_coeff(l, m, k))


def get_ml_array(deg_view):
    """All (m, l) pairs: l = 1, 2, 4, ..., 2^(deg_view-1), m = 0..l  ->  int array [2, n]."""
    return np.array([(m, 2 ** i) for i in range(deg_view) for m in range(2 ** i + 1)]).T


class _Ide(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, kappa_inv, mat, ml):
        M, n = xyz.shape[0], ml.shape[1]
        out = torch.empty(M, 2 * n, dtype=torch.float32, device=xyz.device)
        call("fgs_ide_fwd", ptr(xyz), ptr(kappa_inv), ptr(mat), ptr(ml), n, mat.shape[0], M, ptr(out), stream())
        ctx.save_for_backward(xyz, kappa_inv, mat, ml)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_out):
        xyz, kappa_inv, mat, ml = ctx.saved_tensors
        M, n = xyz.shape[0], ml.shape[1]
        g_xyz, g_k = torch.empty_like(xyz), torch.empty_like(kappa_inv)
        call("fgs_ide_bwd", ptr(xyz), ptr(kappa_inv), ptr(mat), ptr(ml), n, mat.shape[0], M, ptr(g_out.contiguous()),
             ptr(g_xyz), ptr(g_k), stream())
        return g_xyz, g_k, None, None


def generate_ide_fn(deg_view):
    """Returns ``integrated_dir_enc_fn(xyz[..., 3], kappa_inv[..., 1]) -> [..., 2n]`` (real parts, then imaginary parts)."""
    if deg_view > 5:
        raise ValueError('Only deg_view of at most 5 is numerically stable.')
    ml_array = get_ml_array(deg_view)
    l_max = 2 ** (deg_view - 1)
    mat_host = torch.zeros(l_max + 1, ml_array.shape[1])
    for i, (m, l) in enumerate(ml_array.T):
        for k in range(l - m + 1):
            mat_host[k, i] = sph_harm_coeff(l, m, k)
    tables = {}

    def integrated_dir_enc_fn(xyz, kappa_inv):
        dev = xyz.device
        if dev not in tables:
            tables[dev] = (mat_host.to(dev).contiguous(), torch.from_numpy(ml_array.astype(np.int32)).to(dev).contiguous())
        mat, ml = tables[dev]
        lead = xyz.shape[:-1]
        x = check_input(xyz.reshape(-1, 3).float().contiguous(), "xyz")
        k = check_input(kappa_inv.expand(*lead, 1).reshape(-1).float().contiguous(), "kappa_inv")
        return _Ide.apply(x, k, mat, ml).reshape(*lead, 2 * ml.shape[1])

    return integrated_dir_enc_fn
