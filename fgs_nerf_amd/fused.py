"""Public surface of the fused MI355X path for ``nerf.forward_fine`` / ``nerf.forward_coarse`` (model/nerf.py:776-941,
943-1075): one autograd node per forward.

Forward:   march (1 wave / ray: sampling + SDF + 6-tap gradient + NeuS alpha + exact early-terminating scan; coarse: two
           trilinear lookups in the dense smoothed / gradient volumes and both Alphas2Weights passes) -> scan of the per-ray
           survivor counts with the capacity guard in the same launch -> survivor compaction -> feature kernels (k0 trilerp,
           24 SDF taps, encodings, reflection) writing straight into the MLP operand buffers -> the MLP forward chain in one
           register-resident fp32-MFMA launch (``fgs_mlp_rc_chain``; ``FGS_MLP=lds`` keeps round 1's LDS-tiled forms) ->
           3-wide head + sigmoid -> per-ray compositing.
Backward:  loss / compositing -> head -> the data-gradient chain (one launch) -> two narrow first-layer products ->
           every weight and bias gradient in ONE launch (``fgs_mlp_wgrad``) on a side stream / graph branch, beside the
           feature scatter (k0.grad), encoding backward, march backward (alpha2weight + NeuS alpha) and the LDS-brick sdf
           scatter.  With a ``dist.GradAverager`` attached the gradients are handed to the exchange from in here.

Host reads per step: none in the sync-free / captured form (the survivor count stays in device memory, ``fgs_dyn_t.row_count``);
the eager form reads the 8-byte count once to size the result tensors the reference API returns.  The reference touches the
host ~10 times per forward (``.item()``, seven boolean-mask compactions, ``unique``).  Configurations outside ``supports`` /
``supports_coarse`` run the operator-at-a-time kernels.
"""
# The implementation lives in three modules; this one keeps the public surface (and the names tests and scripts reach for):
#   fused_common.py   geometry, run record, sync-free switch, MLP launch helpers, k0 gradient buffer, workspace, hooks
#   fused_fine.py     supports, _FusedFine, forward_fine
#   fused_coarse.py   supports_coarse, _FusedCoarse, forward_coarse
from . import fused_common, fused_coarse, fused_fine      # noqa: F401
from .fused_common import *                                  # noqa: F401,F403
from .fused_coarse import _FusedCoarse, forward_coarse, supports_coarse      # noqa: F401
from .fused_fine import _FusedFine, _backward_rc, _fine_grad_layout, _layout, forward_fine, supports      # noqa: F401
