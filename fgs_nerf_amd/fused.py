"""Fused MI355X path for ``nerf.forward_fine`` / ``nerf.forward_coarse`` (model/nerf.py:776-941, 943-1075): one autograd
node per forward, ~20 HIP launches each way.

Forward:   march (1 wave / ray: sampling + SDF + 6-tap gradient + NeuS alpha + exact early-terminating scan; coarse: two
           trilinear lookups in the dense smoothed / gradient volumes and both Alphas2Weights passes)
           -> scan of per-ray survivor counts -> ONE device->host read (M_s, needed to size the result tensors the
           reference API returns) -> survivor compaction -> feature kernels (k0 trilerp, 24 SDF taps, encodings,
           reflection) writing straight into the MLP operand buffers -> the whole MLP chain in one persistent fp32-MFMA
           launch (k_mlp_fwd; per-layer k_gemm launches for widths other than 256) -> 3-wide head + sigmoid -> per-ray
           compositing.  The backward pass's big zero fills are issued here, at the end.
Backward:  compositing -> head -> per layer ONE launch with the data-gradient tiles and the split-K weight-gradient
           workgroups (k_linear_bwd; bias gradients in the epilogues) -> feature scatter (k0.grad) -> march backward
           (alpha2weight + NeuS alpha) -> all sdf.grad contributions combined per survivor in LDS bricks.
           With a dist.GradAverager attached, the MLP gradients and k0.grad are handed to the exchange from in here.

The reference touches the host ~10 times per forward (`.item()`, seven boolean-mask compactions, `unique`); this
path does it once.  Configurations outside ``supports`` / ``supports_coarse`` run the operator-at-a-time kernels.
"""
# The implementation lives in three modules; this one keeps the public surface (and the names tests and scripts reach for):
#   fused_common.py   geometry, run record, sync-free switch, MLP launch helpers, k0 gradient buffer, workspace, hooks
#   fused_fine.py     supports, _FusedFine, forward_fine
#   fused_coarse.py   supports_coarse, _FusedCoarse, forward_coarse
from . import fused_common, fused_coarse, fused_fine      # noqa: F401
from .fused_common import *                                  # noqa: F401,F403
from .fused_coarse import _FusedCoarse, forward_coarse, supports_coarse      # noqa: F401
from .fused_fine import _FusedFine, _backward_rc, _fine_grad_layout, _layout, forward_fine, supports      # noqa: F401
