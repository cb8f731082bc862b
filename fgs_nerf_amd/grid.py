"""Dense voxel grids behind the reference's ``model/grid.py`` surface.

``create_grid``, ``DenseGrid`` and ``MaskGrid`` keep the constructor signatures, attribute names
(``.grid`` Parameter of logical shape [1,C,X,Y,Z], ``.channels``, ``.world_size``, ``.xyz_min``,
``.xyz_max``; ``MaskGrid`` buffers ``mask``, ``xyz2ijk_scale``, ``xyz2ijk_shift``), methods and
``state_dict`` keys of model/grid.py:27-130,253-287.  Underneath, the trilinear lookup is the HIP
kernel pair of csrc/trilerp.hip instead of F.grid_sample, and multi-channel grids are STORED
channel-last (torch.channels_last_3d strides on the same logical [1,C,X,Y,Z] shape) so the C values
of a voxel corner are one contiguous run in HBM.  Logical indexing, ``.data =`` assignment,
``F.interpolate`` rescaling and checkpoints are unaffected by the physical layout; a tensor
assigned in channel-first layout is served as is through the stride-generic kernels.

``TensoRFGrid`` (model/grid.py:136-247) is never instantiated by the reference and is out of scope.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


def resolution_for(xyz_min, xyz_max, num_voxels):
    """(voxel_size, world_size) of a box cut into about `num_voxels` cubic voxels (model/nerf.py:298-307,
    model/dvgo.py:101-109): edge = cube root of (box volume / voxel budget); voxels per axis = floor(extent / edge), int64."""
    extent = xyz_max - xyz_min
    edge = (extent.prod() / num_voxels).pow(1 / 3)
    return edge, (extent / edge).long()


def create_grid(type, **kwargs):
    """model/grid.py:27-33."""
    if type == 'DenseGrid':
        return DenseGrid(**kwargs)
    raise NotImplementedError(type)


def to_grid_layout(t: torch.Tensor) -> torch.Tensor:
    """Physical layout used for grid storage: channel-last for C > 1 (C == 1 is the same memory either way)."""
    if t.dim() == 5 and t.shape[1] > 1:
        return t.contiguous(memory_format=torch.channels_last_3d)
    return t.contiguous()


class _Trilerp(torch.autograd.Function):
    """grid[1,C,X,Y,Z] sampled at world points [M,3] -> [M,C]; backward scatter-adds into the grid."""

    @staticmethod
    def forward(ctx, grid, pts, xyz_min, xyz_max):
        out = ops.trilerp_fwd(grid, pts, xyz_min, xyz_max)
        ctx.save_for_backward(pts, xyz_min, xyz_max)
        ctx.grid_meta = (grid.shape, grid.stride())
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        pts, xyz_min, xyz_max = ctx.saved_tensors
        shape, stride = ctx.grid_meta
        grad_grid = torch.empty_strided(shape, stride, dtype=torch.float32, device=grad_out.device).zero_()
        ops.trilerp_bwd(grad_grid, pts, xyz_min, xyz_max, grad_out.contiguous())
        return grad_grid, None, None, None


def trilerp(grid: torch.Tensor, pts: torch.Tensor, xyz_min: torch.Tensor, xyz_max: torch.Tensor) -> torch.Tensor:
    try:
        ops.grid_strides(grid)
    except RuntimeError:
        grid = to_grid_layout(grid)  # e.g. a non-dense view; copy once
    return _Trilerp.apply(grid, pts.contiguous(), xyz_min, xyz_max)


class DenseGrid(nn.Module):
    """model/grid.py:38-130."""

    def __init__(self, channels, world_size, xyz_min, xyz_max, **kwargs):
        super().__init__()
        self.channels = channels
        self.world_size = world_size
        dev = 'cuda' if torch.cuda.is_available() else 'cpu'  # the reference hard-codes .to('cuda') (grid.py:45-46)
        self.xyz_min = torch.as_tensor(xyz_min, dtype=torch.float32).to(dev)
        self.xyz_max = torch.as_tensor(xyz_max, dtype=torch.float32).to(dev)
        self.grid = nn.Parameter(to_grid_layout(torch.zeros([1, channels, *[int(w) for w in world_size]])))

    def _bounds(self, device):
        if self.xyz_min.device != device:
            self.xyz_min = self.xyz_min.to(device)
            self.xyz_max = self.xyz_max.to(device)
        return self.xyz_min.contiguous(), self.xyz_max.contiguous()

    def forward(self, xyz, importance=None):
        '''
        xyz: global coordinates to query
        '''
        shape = xyz.shape[:-1]
        pts = xyz.reshape(-1, 3)
        lo, hi = self._bounds(self.grid.device)
        out = trilerp(self.grid, pts, lo, hi).reshape(*shape, self.channels)
        if self.channels == 1:
            out = out.squeeze(-1)
        if importance is not None:
            # model/grid.py:61-66 -- auxiliary lookup with align_corners=False; not on the hot path, left to torch
            ind_norm = ((xyz.reshape(1, 1, 1, -1, 3) - lo) / (hi - lo)).flip((-1,)) * 2 - 1
            sampled = F.grid_sample(importance, ind_norm, mode='bilinear', align_corners=False)
            sampled = sampled.reshape(self.channels, -1).T.reshape(*shape, self.channels)
            if self.channels == 1:
                return out, sampled.squeeze(-1)
        return out

    @torch.no_grad()
    def set_alpha(self, xyz, alpha):
        """model/grid.py:70-99.  The reference writes `alpha` into a CLONE of the grid and then re-points
        ``grid.data`` at a view of the original storage (its last statement, :99), so its net effect is
        that the grid keeps its values.  No caller exists in the reference; the quirk is kept, not fixed:
        arguments are validated the same way and the grid is left unchanged."""
        ws = torch.as_tensor(self.world_size, device=xyz.device)
        lo, hi = self._bounds(xyz.device)
        ind = (((xyz - lo) / (hi - lo)) * (ws - 1)).round().long()
        assert ind.shape[-1] == 3 and alpha is not None

    def scale_volume_grid(self, new_world_size):
        size = tuple(int(w) for w in new_world_size)
        if self.channels == 0:
            self.grid = nn.Parameter(torch.zeros([1, self.channels, *size], device=self.grid.device))
        else:
            self.grid = nn.Parameter(to_grid_layout(
                F.interpolate(self.grid.data, size=size, mode='trilinear', align_corners=True)))

    def total_variation_add_grad(self, wx, wy, wz, dense_mode, mask=None):
        '''Add gradients by total variation loss in-place'''
        grad = self.grid.grad
        if grad.stride() != self.grid.stride():
            grad = torch.empty_strided(self.grid.shape, self.grid.stride(), dtype=grad.dtype, device=grad.device).copy_(grad)
            self.grid.grad = grad
        if dense_mode and getattr(self.grid, '_fgs_touched', None) is not None:
            # a dense TV term makes the gradient non-zero outside the bricks the rays touched: MaskedAdam must take the
            # dense update for this step (the sparse mode only adds where grad != 0 and keeps the record valid)
            self.grid._fgs_touched = None
        if mask is None:
            ops.total_variation_cuda.total_variation_add_grad(self.grid, grad, wx, wy, wz, dense_mode)
        else:
            mask = mask.detach()
            if self.grid.size(1) > 1 and mask.size() != self.grid.size():
                mask = mask.repeat(1, self.grid.size(1), 1, 1, 1)
            assert mask.size() == self.grid.size()
            m = torch.empty_strided(self.grid.shape, self.grid.stride(), dtype=torch.float32, device=grad.device)
            m.copy_(mask)
            ops.total_variation_cuda.total_variation_add_grad_new(self.grid, grad, m, wx, wy, wz, dense_mode)

    def get_dense_grid(self):
        return self.grid

    @torch.no_grad()
    def __isub__(self, val):
        self.grid.data -= val
        return self

    def extra_repr(self):
        ws = self.world_size.tolist() if hasattr(self.world_size, 'tolist') else list(self.world_size)
        return f'channels={self.channels}, world_size={ws}'


class MaskGrid(nn.Module):
    """model/grid.py:253-287: nearest-voxel occupancy lookup."""

    def __init__(self, path=None, mask_cache_thres=None, mask=None, xyz_min=None, xyz_max=None):
        super().__init__()
        if path is not None:
            # only tensors and plain containers are expected in a stage checkpoint
            st = torch.load(path, map_location='cpu', weights_only=True)
            self.mask_cache_thres = mask_cache_thres
            alpha = F.max_pool3d(st['model_state_dict']['alpha.grid'], kernel_size=3, padding=1, stride=1)
            mask = (alpha >= self.mask_cache_thres).squeeze(0).squeeze(0)
            xyz_min = torch.Tensor(st['model_kwargs']['xyz_min'])
            xyz_max = torch.Tensor(st['model_kwargs']['xyz_max'])
        else:
            mask = mask.bool()
            xyz_min = torch.as_tensor(xyz_min, dtype=torch.float32)
            xyz_max = torch.as_tensor(xyz_max, dtype=torch.float32)
        dev = mask.device
        xyz_min, xyz_max = xyz_min.to(dev), xyz_max.to(dev)
        self.register_buffer('mask', mask.contiguous())
        xyz_len = xyz_max - xyz_min
        self.register_buffer('xyz2ijk_scale', (torch.Tensor(list(mask.shape)).to(dev) - 1) / xyz_len)
        self.register_buffer('xyz2ijk_shift', -xyz_min * self.xyz2ijk_scale)

    @torch.no_grad()
    def forward(self, xyz):
        '''Skip know freespace
        @xyz:   [..., 3] the xyz in global coordinate.
        '''
        shape = xyz.shape[:-1]
        xyz = xyz.reshape(-1, 3).contiguous()
        mask = ops.render_utils_cuda.maskcache_lookup(self.mask, xyz, self.xyz2ijk_scale, self.xyz2ijk_shift)
        return mask.reshape(shape)

    def extra_repr(self):
        return f'mask.shape={list(self.mask.shape)}'
