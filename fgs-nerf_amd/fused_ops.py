"""Thin tensor-level wrappers over the fused-path entry points of libfgs_hip.so (GEMM, march, features, compositing).

Everything here takes pre-allocated, correctly laid-out CUDA tensors and only forwards pointers, sizes and the current
HIP stream; argument checking that needs the device happens in the C ABI (negative FGS_E_* codes -> FgsError).
"""
from __future__ import annotations

import torch

from ._lib import call, ptr, stream

GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2


def gemm(op: int, A: torch.Tensor, B: torch.Tensor, C: torch.Tensor, M: int, N: int, K: int, bias=None, relu=False,
         mask=None, colsum=None) -> torch.Tensor:
    """C = op(A, B) with the epilogues of include/fgs_hip.h fgs_gemm_f32.  A, B, C, mask are 2-D row-major views
    (stride(1) == 1); leading dimensions are taken from stride(0)."""
    for t in (A, B, C) + ((mask,) if mask is not None else ()):
        if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1):
            raise RuntimeError("gemm operands must be 2-D float32 CUDA tensors with unit column stride")
    call("fgs_gemm_f32", op, M, N, K, ptr(A), A.stride(0), ptr(B), B.stride(0), ptr(C), C.stride(0), ptr(bias),
         int(bool(relu)), ptr(mask), 0 if mask is None else mask.stride(0), ptr(colsum), stream())
    return C
