"""The SDF / NeuS + reflection-MLP scene model behind the reference's ``model/nerf.py`` surface.

``nerf(**model_kwargs)`` accepts the reference constructor arguments (model/nerf.py:23-39), exposes
the attributes the training / eval loops touch (``sdf``, ``k0``, ``rgbnet``, ``refnet``,
``mask_cache``, ``nonempty_mask``, ``world_size``, ``voxel_size``, ``s_val`` ...), keeps the
``state_dict`` key names (``sdf.grid``, ``k0.grid``, ``refnet.*``, ``rgbnet.*``, ``xyz_min`` ...) and
returns the same ``ret_dict`` keys from ``forward`` (rebound to ``forward_coarse`` or ``forward_fine``
by stage, model/nerf.py:47-50).

Two execution paths produce those results:

* ``fused=True`` (default on a GPU when the configuration is covered): the wave-per-ray march kernel,
  survivor feature kernel, fp32-MFMA MLP and compositing kernels of fused.py, one autograd node.
* otherwise: the operator-at-a-time HIP kernels of render.py composed exactly in the reference order.

Host-side utilities of the reference class that never touch the hot path (mesh extraction via
PyMCubes, IDE construction) are not reproduced here; see DESIGN.md "Out of scope".
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import grid as grid_mod
from . import nerf_ray, ops
from .render import Alphas2Weights, grid_sampler, l2_normalize, posenc, sample_sdfs, segment_sum


def _mlp(dim_in, width, depth, dim_out):
    """model/nerf.py:125-142 layout: Linear+ReLU, (depth-2) x Sequential(Linear, ReLU), Linear -- the nesting
    fixes the state_dict key names (``0.weight``, ``2.0.weight`` ... ``<depth>.weight``)."""
    return nn.Sequential(
        nn.Linear(dim_in, width), nn.ReLU(inplace=True),
        *[nn.Sequential(nn.Linear(width, width), nn.ReLU(inplace=True)) for _ in range(depth - 2)],
        nn.Linear(width, dim_out))


def mlp_layers(seq: nn.Sequential):
    """The Linear modules of an `_mlp` stack in execution order (cached on the stack: walking the module tree costs ~15 us
    and the fused path asks several times per step; the cache is keyed on the stack's direct children)."""
    key = tuple(id(m) for m in seq.children())
    cached = seq.__dict__.get('_fgs_linear_layers')
    if cached is None or cached[0] != key:
        cached = (key, [m for m in seq.modules() if isinstance(m, nn.Linear)])
        seq.__dict__['_fgs_linear_layers'] = cached
    return list(cached[1])


def load_checkpoint_file(path):
    """Stage hand-off files (`*_last.tar`) hold tensors, numbers and numpy arrays (model/nerf_training.py:524-530);
    they are read with the restricted unpickler, allowing only numpy's array reconstruction."""
    import numpy.core.multiarray as _ma  # noqa: F401
    allowed = [np.ndarray, np.dtype, _ma._reconstruct, type(np.dtype(np.float32)), type(np.dtype(np.float64)),
               type(np.dtype(np.int64))]
    with torch.serialization.safe_globals(allowed):
        return torch.load(path, map_location='cpu', weights_only=True)


class MaskCache(nn.Module):
    """model/nerf.py:1192-1209: known-free-space test = trilinear sample of the 3^3-max-pooled ``sdf_mask`` >= thres."""

    def __init__(self, path=None, mask_cache_thres=None, stage='', ks=3, sdf_mask=None, xyz_min=None, xyz_max=None):
        super().__init__()
        if path is not None:
            st = load_checkpoint_file(path)
            sdf_mask = st['model_state_dict']['sdf_mask.grid']
            xyz_min = st['MaskCache_kwargs']['xyz_min']
            xyz_max = st['MaskCache_kwargs']['xyz_max']
        self.mask_cache_thres = mask_cache_thres
        self.register_buffer('xyz_min', torch.as_tensor(np.asarray(xyz_min), dtype=torch.float32))
        self.register_buffer('xyz_max', torch.as_tensor(np.asarray(xyz_max), dtype=torch.float32))
        self.register_buffer('sdf_mask', F.max_pool3d(sdf_mask.float(), kernel_size=ks, padding=ks // 2, stride=1).contiguous())

    @torch.no_grad()
    def forward(self, xyz):
        shape = xyz.shape[:-1]
        pts = xyz.reshape(-1, 3).to(self.xyz_max.device).contiguous()
        v = ops.trilerp_fwd(self.sdf_mask, pts, self.xyz_min, self.xyz_max)
        return v.reshape(*shape) >= self.mask_cache_thres


class nerf(torch.nn.Module):
    def __init__(self,
                 xyz_min, xyz_max,
                 num_voxels=0, num_voxels_base=0,
                 nearest=False,
                 mask_cache_path=None, mask_cache_thres=1e-5,
                 fast_color_thres=0,
                 k0_dim=12, rgbnet_depth=4, rgbnet_width=256,
                 ref=False, refnet_width=256, refnet_depth=4, sh_max_level=4,
                 posbase_pe=5, viewbase_pe=3, refbase_pe=8,
                 grad_feat=(), sdf_feat=(),
                 k_grad_feat=(1.0,), k_sdf_feat=(),
                 use_grad_norm=True, center_sdf=True,
                 grad_mode='interpolate',
                 s_ratio=2000, s_start=0.05, s_learn=False, step_start=0,
                 smooth_ksize=0, smooth_sigma=1, smooth_scale=True,
                 training=False, stage='', use_viewdir=True, fused=None,
                 **kwargs):
        super().__init__()
        if nearest:
            raise NotImplementedError("nearest-voxel lookup is not used by any reference config")
        self.training = training
        self.stage = stage
        self.ref = ref
        self.use_viewdir = use_viewdir
        self.forward = self.forward_coarse if stage in ('coarse', 'geometry_searching') else self.forward_fine

        self.register_buffer('xyz_min', torch.Tensor(np.asarray(xyz_min, dtype=np.float32)))
        self.register_buffer('xyz_max', torch.Tensor(np.asarray(xyz_max, dtype=np.float32)))
        self.fast_color_thres = fast_color_thres
        self.nearest = nearest
        self.smooth_scale = smooth_scale
        self.s_ratio, self.s_start, self.s_learn, self.step_start = s_ratio, s_start, s_learn, step_start
        self.s_val = nn.Parameter(torch.ones(1) * s_start, requires_grad=s_learn)
        self.sdf_init_mode = "ball_init"

        self.num_voxels_base = num_voxels_base
        self.voxel_size_base = ((self.xyz_max - self.xyz_min).prod() / self.num_voxels_base).pow(1 / 3)
        self._set_grid_resolution(num_voxels)

        # SDF grid, ball initialisation (model/nerf.py:73-82)
        self.sdf = grid_mod.create_grid('DenseGrid', channels=1, world_size=self.world_size,
                                        xyz_min=self.xyz_min, xyz_max=self.xyz_max)
        gx, gy, gz = (int(w) for w in self.world_size)
        lx, ly, lz = np.mgrid[-1.0:1.0:gx * 1j, -1.0:1.0:gy * 1j, -1.0:1.0:gz * 1j]
        radius = (lx ** 2 + ly ** 2 + lz ** 2) ** 0.5
        init = radius if stage == 'geometry_searching' else radius - 1
        self.sdf.grid.data = torch.from_numpy(init).float()[None, None, ...]
        self.init_smooth_conv(smooth_ksize, smooth_sigma)

        self.k0_dim = k0_dim
        self.k0 = grid_mod.create_grid('DenseGrid', channels=self.k0_dim, world_size=self.world_size,
                                       xyz_min=self.xyz_min, xyz_max=self.xyz_max)
        self.register_buffer('posfreq', torch.FloatTensor([(2 ** i) for i in range(posbase_pe)]))
        self.register_buffer('viewfreq', torch.FloatTensor([(2 ** i) for i in range(viewbase_pe)]))
        self.register_buffer('reffreq', torch.FloatTensor([(2 ** i) for i in range(refbase_pe)]))

        self.use_grad_norm, self.center_sdf = use_grad_norm, center_sdf
        self.grad_feat, self.sdf_feat = tuple(grad_feat), tuple(sdf_feat)
        self.k_grad_feat, self.k_sdf_feat = tuple(k_grad_feat), tuple(k_sdf_feat)
        pos_dim, view_dim, ref_dim = 3 + 6 * posbase_pe, 3 + 6 * viewbase_pe, 3 + 6 * refbase_pe
        rgbnet_dim = pos_dim + k0_dim + 3 + 3 * len(self.grad_feat) + 6 * len(self.sdf_feat)
        rgbnet_dim += 1 if center_sdf else 0
        rgbnet_dim += view_dim if use_viewdir else 0
        if stage == 'fine':
            refnet_dim = ref_dim + refnet_width
        else:
            refnet_dim = ref_dim + k0_dim + pos_dim + 3 + (view_dim if use_viewdir else 0)
        self.refnet_width, self.refnet_depth, self.refnet_dim = refnet_width, refnet_depth, refnet_dim
        self.refnet = _mlp(refnet_dim, refnet_width, refnet_depth, 3)
        self.rgbnet = _mlp(rgbnet_dim, rgbnet_width, rgbnet_depth, rgbnet_width) if stage == 'fine' else None
        self.mlp_kwargs = {'rgbnet_dim': rgbnet_dim, 'rgbnet_width': rgbnet_width, 'rgbnet_depth': rgbnet_depth,
                           'refnet_dim': refnet_dim, 'refnet_width': refnet_width, 'refnet_depth': refnet_depth}

        # known free space (model/nerf.py:157-172).  mask_cache_path=None (no geometry_searching checkpoint yet,
        # e.g. synthetic benchmarks) simply disables the skip.
        self.mask_cache_path, self.mask_cache_thres = mask_cache_path, mask_cache_thres
        self.mask_cache, self.inc_mask = None, None
        self.register_buffer('nonempty_mask', None)
        if stage != 'geometry_searching' and mask_cache_path is not None:
            self.mask_cache = MaskCache(path=mask_cache_path, mask_cache_thres=mask_cache_thres, stage=stage)
            self._set_nonempty_mask()

        self.grad_mode = grad_mode
        self.init_gradient_conv()
        self.get_rays_of_a_view = nerf_ray.get_rays_of_a_view
        from .ide import generate_ide_fn                          # model/nerf.py:179 (built, never evaluated there)
        self.integrated_dir_enc = generate_ide_fn(sh_max_level)
        self.fused = fused
        self._gradient, self._gradient_pending = None, False

    # `self.gradient` -- the dense central-difference volume the reference recomputes inside EVERY forward
    # (model/nerf.py:856 fine, :972 coarse).  The fine stage only reads it from density_total_variation (every
    # tv_every-th iteration), so forward_fine marks it pending and the volume (a HIP stencil over the current sdf.grid,
    # an autograd node) is built on first access.
    @property
    def gradient(self):
        if self._gradient_pending:
            self._gradient_pending = False
            self._gradient = self._gradient_volume()
        return self._gradient

    @gradient.setter
    def gradient(self, value):
        self._gradient, self._gradient_pending = value, False

    def _gradient_volume(self):
        g = self.sdf.grid
        if g.is_cuda and self.grad_mode in ('interpolate', 'raw', 'grad_conv') and g.is_contiguous():
            from . import dense
            return dense.sdf_gradient_volume(g, float(self.voxel_size), mode=self.grad_mode,
                                             grad_conv_weight=self.grad_conv.weight if self.grad_mode == 'grad_conv' else None)
        return self.neus_sdf_gradient()

    # ------------------------------------------------------------------ resolution / bookkeeping
    def _set_grid_resolution(self, num_voxels):
        """model/nerf.py:298-307: the voxel budget fixes the (cubic) voxel size, the box extent then the per-axis counts."""
        self.num_voxels = num_voxels
        self.voxel_size, self.world_size = grid_mod.resolution_for(self.xyz_min, self.xyz_max, num_voxels)
        self.voxel_size_ratio = self.voxel_size / self.voxel_size_base
        self._world_size_max = None         # host copy of world_size.max(), fetched once per resolution (TV weights)

    def get_kwargs(self):
        """model/nerf.py:309-328."""
        return {'xyz_min': self.xyz_min.cpu().numpy(), 'xyz_max': self.xyz_max.cpu().numpy(),
                'num_voxels': self.num_voxels, 'num_voxels_base': self.num_voxels_base,
                'voxel_size': self.voxel_size, 'nearest': self.nearest, 'k0_dim': self.k0_dim,
                'grad_feat': self.grad_feat, 'sdf_feat': self.sdf_feat, 'center_sdf': self.center_sdf,
                'fast_color_thres': self.fast_color_thres, 'stage': self.stage, 'ref': self.ref,
                'use_viewdir': self.use_viewdir, 's_ratio': self.s_ratio, 's_start': self.s_start,
                **self.mlp_kwargs}

    def get_MaskCache_kwargs(self):
        """model/nerf.py:330-336."""
        return {'xyz_min': self.xyz_min.cpu().numpy(), 'xyz_max': self.xyz_max.cpu().numpy(),
                'voxel_size_ratio': self.voxel_size_ratio, 'nearest': self.nearest}

    def _grid_points(self):
        return torch.stack(torch.meshgrid(
            torch.linspace(float(self.xyz_min[0]), float(self.xyz_max[0]), self.sdf.grid.shape[2]),
            torch.linspace(float(self.xyz_min[1]), float(self.xyz_max[1]), self.sdf.grid.shape[3]),
            torch.linspace(float(self.xyz_min[2]), float(self.xyz_max[2]), self.sdf.grid.shape[4]),
            indexing='ij'), -1).to(self.sdf.grid.device)

    @torch.no_grad()
    def _set_nonempty_mask(self):
        """model/nerf.py:338-353."""
        pts = self._grid_points()
        if not pts.is_cuda:
            # model built on the host and moved later (the reference builds under set_default_device('cuda')): the
            # lookup itself only exists as a HIP kernel, so it runs on the accelerator and the mask comes back
            if not torch.cuda.is_available():
                raise RuntimeError("the mask-cache lookup needs the GPU (no CPU fallback)")
            self.mask_cache.to('cuda')
            pts = pts.cuda()
        nonempty = self.mask_cache(pts).contiguous().reshape(*self.sdf.grid.shape).to(self.sdf.grid.device)
        self.nonempty_mask = nonempty
        if self.stage == 'coarse':
            self.sdf.grid[~nonempty] = 1

    @torch.no_grad()
    def maskout_near_cam_vox(self, cam_o, near):
        """model/nerf.py:355-366."""
        pts = self._grid_points()
        nearest = torch.stack([(pts.unsqueeze(-2) - co).pow(2).sum(-1).sqrt().amin(-1)
                               for co in cam_o.to(pts.device).split(100)]).amin(0)
        self.sdf.grid[nearest[None, None] <= near] = 5

    @torch.no_grad()
    def scale_volume_grid(self, num_voxels):
        """model/nerf.py:368-381 (trilinear resampling of both grids, a16)."""
        self._set_grid_resolution(num_voxels)
        self.sdf.scale_volume_grid(self.world_size)
        self.k0.scale_volume_grid(self.world_size)
        if self.mask_cache is not None:
            self._set_nonempty_mask()

    @torch.no_grad()
    def reset_voxel_and_mlp(self):
        """model/nerf.py:383-396."""
        dev = self.sdf.grid.device
        self.refnet = _mlp(self.refnet_dim, self.refnet_width, self.refnet_depth, 3).to(dev)

    def set_sdf_mask(self):
        """model/nerf.py:181-200: a one-channel ``sdf_mask`` grid saved with the stage checkpoint, 1e-3 wherever the
        (optionally smoothed) SDF is below 0.5 and 0 elsewhere."""
        field = self.smooth_conv(self.sdf.grid) if self.smooth_sdf else self.sdf.grid
        inside = (field.detach() < 0.5).to(torch.float32)
        self.sdf_mask = grid_mod.create_grid('DenseGrid', channels=1, world_size=self.world_size,
                                             xyz_min=self.xyz_min, xyz_max=self.xyz_max)
        self.sdf_mask.grid.data = (inside * 1e-3).reshape(1, 1, *field.shape[-3:]).contiguous()

    def init_sdf_from_sdf(self, sdf0=None, smooth=False, reduce=1., ksize=3, sigma=1., zero2neg=True):
        """model/nerf.py:280-296 (fine stage start: resample the coarse SDF, 5^3 sigma=1 smoothing)."""
        dev = self.sdf.grid.device
        if sdf0.shape != self.sdf.grid.shape:
            sdf0 = F.interpolate(sdf0.to(dev), size=tuple(int(w) for w in self.world_size), mode='trilinear',
                                 align_corners=True)
        if smooth:
            m = self._gaussian_3dconv(ksize, sigma)
            self.sdf.grid = nn.Parameter(m(sdf0.to(dev) / reduce) / reduce)
        else:
            self.sdf.grid.data = (sdf0.to(dev) / reduce).contiguous()
        if self.mask_cache is not None:
            self._set_nonempty_mask()
        if self.smooth_scale:
            m = self._gaussian_3dconv(ksize=5, sigma=1)
            with torch.no_grad():
                self.sdf.grid = nn.Parameter(m(self.sdf.grid.data).contiguous())
        self.gradient = self.neus_sdf_gradient()

    # ------------------------------------------------------------------ dense per-step volume ops (a6)
    def _gaussian_3dconv(self, ksize=3, sigma=1):
        """model/nerf.py:260-272: frozen Conv3d with normalised Gaussian taps, replicate padding."""
        ax = np.arange(-(ksize // 2), ksize // 2 + 1, 1)
        xx, yy, zz = np.meshgrid(ax, ax, ax)
        kernel = torch.from_numpy(np.exp(-(xx ** 2 + yy ** 2 + zz ** 2) / (2 * sigma ** 2))).to(self.sdf.grid)
        m = nn.Conv3d(1, 1, ksize, stride=1, padding=ksize // 2, padding_mode='replicate').to(self.sdf.grid.device)
        m.weight.data = kernel[None, None, ...] / kernel.sum()
        m.bias.data = torch.zeros(1, device=self.sdf.grid.device)
        for p in m.parameters():
            p.requires_grad = False
        return m

    def init_smooth_conv(self, ksize=3, sigma=1):
        """model/nerf.py:274-278."""
        self.smooth_sdf = ksize > 0
        if self.smooth_sdf:
            self.smooth_conv = self._gaussian_3dconv(ksize, sigma)

    def init_gradient_conv(self, sigma=0):
        """model/nerf.py:224-258: ``tv_smooth_conv`` (3^3 binomial, replicate pad; used by density_total_variation) and the
        Sobel-like ``grad_conv`` of grad_mode='grad_conv' (no shipped config selects it; pinned against the reference's own
        weights and output in tests/golden/ref_fns.npz)."""
        base = np.asarray([[[1, 2, 1], [2, 4, 2], [1, 2, 1]], [[2, 4, 2], [4, 8, 4], [2, 4, 2]],
                           [[1, 2, 1], [2, 4, 2], [1, 2, 1]]], dtype=np.float64)
        dist = np.fromfunction(lambda i, j, k: (i - 1) ** 2 + (j - 1) ** 2 + (k - 1) ** 2 - 1, (3, 3, 3))
        kernel0 = base * np.exp(-dist * sigma)
        kernel1 = kernel0 / (kernel0[0].sum() * 2 * float(self.voxel_size))                      # :238
        weight = torch.from_numpy(np.concatenate([kernel1[None] for _ in range(3)])).float()     # :239-245
        weight[0, 1, :, :] *= 0
        weight[0, 0, :, :] *= -1
        weight[1, :, 1, :] *= 0
        weight[1, :, 0, :] *= -1
        weight[2, :, :, 1] *= 0
        weight[2, :, :, 0] *= -1
        self.grad_conv = nn.Conv3d(1, 3, (3, 3, 3), stride=(1, 1, 1), padding=(1, 1, 1), padding_mode='replicate')
        self.grad_conv.weight.data = weight.unsqueeze(1).float()
        self.grad_conv.bias.data = torch.zeros(3)
        for p in self.grad_conv.parameters():
            p.requires_grad = False
        self.tv_smooth_conv = nn.Conv3d(1, 1, (3, 3, 3), stride=1, padding=1, padding_mode='replicate')
        self.tv_smooth_conv.weight.data = torch.from_numpy(kernel0 / kernel0.sum()).float()[None, None]
        self.tv_smooth_conv.bias.data = torch.zeros(1)
        for p in self.tv_smooth_conv.parameters():
            p.requires_grad = False

    def neus_sdf_gradient(self, mode=None, sdf=None):
        """model/nerf.py:485-508: 'interpolate' (central difference, zero faces), 'raw' (forward difference), 'grad_conv'."""
        sdf = self.sdf.grid if sdf is None else sdf
        mode = self.grad_mode if mode is None else mode
        g = torch.zeros([1, 3, *self.sdf.grid.shape[-3:]], device=sdf.device)
        if mode == 'interpolate':
            g[:, 0, 1:-1, :, :] = (sdf[:, 0, 2:, :, :] - sdf[:, 0, :-2, :, :]) / 2 / self.voxel_size
            g[:, 1, :, 1:-1, :] = (sdf[:, 0, :, 2:, :] - sdf[:, 0, :, :-2, :]) / 2 / self.voxel_size
            g[:, 2, :, :, 1:-1] = (sdf[:, 0, :, :, 2:] - sdf[:, 0, :, :, :-2]) / 2 / self.voxel_size
        elif mode == 'raw':
            g[:, 0, :-1, :, :] = (sdf[:, 0, 1:, :, :] - sdf[:, 0, :-1, :, :]) / self.voxel_size
            g[:, 1, :, :-1, :] = (sdf[:, 0, :, 1:, :] - sdf[:, 0, :, :-1, :]) / self.voxel_size
            g[:, 2, :, :, :-1] = (sdf[:, 0, :, :, 1:] - sdf[:, 0, :, :, :-1]) / self.voxel_size
        elif mode == 'grad_conv':
            if sdf.is_cuda and sdf.is_contiguous():
                from . import dense
                return dense.sdf_gradient_volume(sdf, float(self.voxel_size), mode='grad_conv', grad_conv_weight=self.grad_conv.weight)
            return self.grad_conv.to(sdf.device)(sdf)
        else:
            raise NotImplementedError(mode)
        return g

    # ------------------------------------------------------------------ regularisers
    def density_total_variation(self, sdf_tv=0, smooth_grad_tv=0, sdf_thrd=0.999, weight=1.0, add_to=None):
        """model/nerf.py:430-447.  `weight`, `add_to` (not in the reference): returns weight * tv (+ add_to) -- on CUDA grids the
        HIP launches apply the factor and add the loss so far themselves, and a training step's loss then takes two launches per
        term instead of two plus the scalar arithmetic around them (nerf_training passes weight_tv_density and the loss)."""
        fused = self.sdf.grid.is_cuda and not (weight == 1.0 and add_to is None)
        if fused:
            return self._density_tv_fused(sdf_tv, smooth_grad_tv, float(weight), add_to)
        tv = self._density_tv(sdf_tv, smooth_grad_tv)
        if weight != 1.0:
            tv = weight * tv
        return tv if add_to is None else add_to + tv

    def _smooth_tv_operands(self, grad):
        """Host copy of tv_smooth_conv's frozen taps and the (masked) mean's element count as a cached DEVICE scalar."""
        from . import dense
        w = self.tv_smooth_conv.weight
        taps = self.__dict__.get('_tv_taps_c')
        if taps is None or taps[0] is not w:                 # host copy of the frozen taps, made once
            taps = (w, dense._taps_c(w))
            self.__dict__['_tv_taps_c'] = taps
        m = self.nonempty_mask
        cnt = self.__dict__.get('_nonempty_count')
        if cnt is None or cnt[0] is not m or cnt[2] != tuple(grad.shape):
            n = (3.0 * m.sum().to(torch.float32)) if m is not None else torch.tensor(float(grad.numel()), device=grad.device)
            mask_u8 = None if m is None else m.reshape(m.shape[-3:]).contiguous().view(torch.uint8)
            cnt = (m, (1.0 / n).reshape(1).contiguous(), tuple(grad.shape), mask_u8)
            self.__dict__['_nonempty_count'] = cnt
        return taps[1], cnt[3], cnt[1]

    def _voxel_size_host(self) -> float:
        """float(self.voxel_size) read once per voxel_size tensor (a device scalar: reading it inside a stream capture is illegal;
        the warm-up pass in front of every capture fills the cache)."""
        vs = self.voxel_size
        c = self.__dict__.get('_voxel_size_f')
        if c is None or c[0] is not vs:
            c = (vs, float(vs))
            self.__dict__['_voxel_size_f'] = c
        return c[1]

    def _density_tv_fused(self, sdf_tv, smooth_grad_tv, weight, acc):
        from . import dense
        if sdf_tv > 0:
            acc = dense.grid_tv_loss(self.sdf.grid, self.nonempty_mask, per_axis_mean=False,
                                     scale=weight * sdf_tv / 2.0 / self._voxel_size_host(), add_in=acc)
        if smooth_grad_tv > 0:
            grad = self.gradient                                     # [1,3,X,Y,Z]
            if not grad.is_contiguous():
                grad = grad.contiguous()
            taps_c, mask_u8, inv_count = self._smooth_tv_operands(grad)
            acc = dense.smooth_tv_loss(grad, taps_c, mask_u8, inv_count, weight * smooth_grad_tv, add_in=acc)
        return acc if acc is not None else 0

    def _density_tv(self, sdf_tv, smooth_grad_tv):
        tv = 0
        if sdf_tv > 0:
            tv += total_variation(self.sdf.grid, self.nonempty_mask) / 2 / self.voxel_size * sdf_tv
        if smooth_grad_tv > 0:
            grad = self.gradient                                     # [1,3,X,Y,Z]
            if grad.is_cuda and grad.is_contiguous():
                # value and gradient of the term in one LDS-tiled HIP pass per component (csrc/dense.hip); the element
                # count of the (masked) mean is a cached DEVICE scalar, so nothing here reads back to the host
                from . import dense
                taps_c, mask_u8, inv_count = self._smooth_tv_operands(grad)
                tv += dense.smooth_tv_loss(grad, taps_c, mask_u8, inv_count, smooth_grad_tv)
            else:
                g = grad.permute(1, 0, 2, 3, 4)
                err = self.tv_smooth_conv(g).detach() - g
                if self.nonempty_mask is not None:
                    err = err[self.nonempty_mask.repeat(3, 1, 1, 1, 1)] ** 2
                else:
                    err = err ** 2
                tv += err.mean() * smooth_grad_tv
        return tv

    def k0_total_variation(self, k0_tv=1., k0_grad_tv=0.):
        """model/nerf.py:449-459."""
        if k0_grad_tv > 0:
            raise NotImplementedError
        v = self.k0.grid
        if k0_tv <= 0:
            return 0
        # (the reference repeats the mask over the channels, model/nerf.py:454: the denominator is then C * mask.sum();
        # an expanded view says the same without materialising C copies)
        mask = None if self.nonempty_mask is None else self.nonempty_mask.expand(1, v.shape[1], -1, -1, -1)
        return total_variation(v, mask)

    def k0_total_variation_add_grad(self, weight, dense_mode=True):
        """model/nerf.py:461-463."""
        w = self._tv_weight(weight)
        self.k0.total_variation_add_grad(w, w, w, dense_mode)

    def sdf_total_variation_add_grad(self, weight, dense_mode):
        """model/nerf.py:465-467."""
        w = self._tv_weight(weight)
        self.sdf.total_variation_add_grad(w, w, w, dense_mode)

    def _tv_weight(self, weight) -> float:
        """`weight * self.world_size.max() / 128` (model/nerf.py:462,466) with the reference's float32 tensor arithmetic
        (python scalar x int64 tensor -> float32 product, float32 division) done on the host."""
        return float(np.float32(weight) * np.float32(self._world_max()) / np.float32(128))

    def _world_max(self) -> int:
        """world_size.max() as a host integer: the reference multiplies a Python float by the (device) tensor and hands the
        0-d result to the TV kernel wrapper, one device->host read per call; the value only changes with the resolution."""
        if getattr(self, '_world_size_max', None) is None:
            self._world_size_max = int(self.world_size.max())
        return self._world_size_max

    def orientation_loss(self, render_result):
        """model/nerf.py:469-478 (Ref-NeRF orientation regulariser)."""
        zero = torch.tensor(0.0, dtype=torch.float32, device=render_result['normal'].device)
        w = render_result['weights'].detach()
        n_dot_v = (render_result['normal'] * (-render_result['viewdirs'])).sum(dim=-1)
        return torch.mean((w * torch.fmin(zero, n_dot_v) ** 2).sum(dim=-1))

    def l2_normalize(self, x, eps=torch.finfo(torch.float32).eps):
        return l2_normalize(x)

    # ------------------------------------------------------------------ voxel-increment mask (a4)
    @torch.no_grad()
    def set_inc_mask(self, lower, upper):
        """model/nerf.py:1077-1088."""
        ws = [int(w) for w in self.world_size]
        gx, gy, gz = torch.meshgrid(torch.linspace(0, 1, ws[0]), torch.linspace(0, 1, ws[1]),
                                    torch.linspace(0, 1, ws[2]), indexing='ij')
        mask = ((gx >= lower[0]) & (gx <= upper[0]) & (gy >= lower[1]) & (gy <= upper[1]) &
                (gz >= lower[2]) & (gz <= upper[2]))
        self.inc_mask = grid_mod.MaskGrid(path=None, mask=mask.to(self.sdf.grid.device), xyz_min=self.xyz_min,
                                          xyz_max=self.xyz_max)

    @torch.no_grad()
    def unset_inc_mask(self):
        self.inc_mask = None

    def inc_index_bounds(self, lower, upper):
        """The mask of `set_inc_mask(lower, upper)` as six closed index ranges (lo_x, hi_x, lo_y, hi_y, lo_z, hi_z; lo > hi:
        empty axis): the lattice is a per-axis linspace(0, 1, W) compared with the bounds (model/nerf.py:1081-1087), and a
        monotone lattice makes each comparison a contiguous index range -- found with the same torch expressions, so the
        device-side rebuild (fgs_box_mask_fill, a captured iteration) sets exactly the voxels `set_inc_mask` would."""
        out = []
        for a, w in enumerate(int(w) for w in self.world_size):
            g = torch.linspace(0, 1, w)
            inside = (g >= lower[a]) & (g <= upper[a])
            idx = torch.nonzero(inside).flatten()
            if idx.numel() == 0:
                out += [1, 0]
            else:
                lo, hi = int(idx[0]), int(idx[-1])
                assert int(inside.sum()) == hi - lo + 1
                out += [lo, hi]
        return tuple(out)

    # ------------------------------------------------------------------ sampling
    def _stepdist(self, stepsize):
        return float(stepsize * self.voxel_size)

    def sample_ray(self, rays_o, rays_d, near, far, stepsize, **render_kwargs):
        """model/nerf.py:674-698: packed in-bbox samples -> (ray_pts, ray_id, step_id, mask_outbbox, N_steps)."""
        far = 1e9
        rays_o, rays_d = rays_o.contiguous(), rays_d.contiguous()
        ray_pts, mask_outbbox, ray_id, step_id, N_steps, t_min, t_max = ops.render_utils_cuda.sample_pts_on_rays(
            rays_o, rays_d, self.xyz_min, self.xyz_max, near, far, self._stepdist(stepsize))
        N_steps = ray_id.unique(return_counts=True)[1]
        inb = ~mask_outbbox
        return ray_pts[inb], ray_id[inb], step_id[inb], mask_outbbox, N_steps

    def sample_ray_cuda(self, rays_o, rays_d, near, far, stepsize, maskout=True, use_bg=False, **render_kwargs):
        """model/nerf.py:700-732."""
        if use_bg:
            raise NotImplementedError("background grid (voxel_size_bg) does not exist in the reference model")
        ray_pts, ray_id, step_id, mask_outbbox, N_steps = self.sample_ray(rays_o, rays_d, near, far, stepsize)
        if not maskout:
            raise NotImplementedError("maskout=False is not used by the reference")
        return ray_pts, ray_id, step_id, mask_outbbox, N_steps

    def sample_ray_ori(self, rays_o, rays_d, near, far, stepsize, is_train=False, **render_kwargs):
        """model/nerf.py:734-758: the padded [N, n_samples] sampler of the mask-cache ray pre-filter (model/nerf_ray.py:230):
        every ray gets the same number of equidistant points from its box entry, points outside the box are flagged.
        Returns (points [N, n, 3], outside-the-box flags [N, n], step lengths)."""
        from .rays import box_interval
        diag_voxels = float(np.linalg.norm(np.array(self.sdf.grid.shape[2:]) + 1))
        n_samples = int(diag_voxels / stepsize) + 1
        t_in, t_out = box_interval(rays_o, rays_d, self.xyz_min, self.xyz_max, near, far)
        k = torch.arange(n_samples, device=rays_d.device, dtype=torch.float32)[None]
        if is_train:                         # one random offset per ray, shared by its samples
            k = k.repeat(rays_d.shape[-2], 1)
            k += torch.rand_like(k[:, [0]])
        step = stepsize * self.voxel_size * k
        t = t_in[..., None] + step / rays_d.norm(dim=-1, keepdim=True)
        pts = rays_o[..., None, :] + rays_d[..., None, :] * t[..., None]
        missed = (t_out <= t_in)[..., None]
        outside = ((self.xyz_min > pts) | (pts > self.xyz_max)).any(dim=-1)
        return pts, missed | outside, step

    # ------------------------------------------------------------------ NeuS alpha (a8)
    def _s_val_for(self, global_step, is_train):
        if not is_train:
            return 0
        if self.s_learn:
            return self.s_val.item()
        s_val = 1. / (global_step + self.s_ratio / self.s_start - self.step_start) * self.s_ratio
        sf = self.__dict__.get('_fused_cache', {}).get('sync_free')
        if sf is not None and sf.get('inv_s_dev') is not None:
            return s_val                  # a device-resident schedule owns s_val and 1/s (graph_step.CapturedFineStep)
        self.s_val.data.fill_(s_val)      # model/nerf.py:520 `torch.ones_like(self.s_val) * s_val`: same value, one launch
        return s_val

    def neus_alpha_from_sdf_scatter(self, viewdirs, ray_id, dist, sdf, gradients, global_step, is_train, use_mid=True):
        """model/nerf.py:510-544."""
        assert use_mid
        s_val = self._s_val_for(global_step, is_train)
        dirs = viewdirs[ray_id]
        inv_s = torch.ones(1, device=sdf.device) / self.s_val
        true_cos = (dirs * gradients).sum(-1, keepdim=True)
        iter_cos = -(F.relu(-true_cos * 0.5 + 0.5) * (1.0 - 1.0) + F.relu(-true_cos) * 1.0)
        sdf = sdf.unsqueeze(-1)
        half = iter_cos * dist.reshape(-1, 1) * 0.5
        prev_cdf = torch.sigmoid((sdf - half) * inv_s.reshape(-1, 1))
        next_cdf = torch.sigmoid((sdf + half) * inv_s.reshape(-1, 1))
        alpha = ((prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)).clip(0.0, 1.0).squeeze(-1)
        return s_val, alpha

    # ------------------------------------------------------------------ forward paths
    def _use_fused(self, rays_o):
        if self.fused is False or not rays_o.is_cuda:
            return False
        from . import fused
        return fused.supports(self)

    def forward_fine(self, rays_o, rays_d, viewdirs, global_step=20000, **render_kwargs):
        """model/nerf.py:776-941."""
        self._gradient, self._gradient_pending = None, True     # model/nerf.py:856, evaluated lazily
        if self._use_fused(rays_o):
            from . import fused
            return fused.forward_fine(self, rays_o, rays_d, viewdirs, global_step, **render_kwargs)
        return self._forward_fine_composed(rays_o, rays_d, viewdirs, global_step, **render_kwargs)

    def forward_coarse(self, rays_o, rays_d, viewdirs, global_step=20000, **render_kwargs):
        """model/nerf.py:943-1075."""
        if self.fused is not False and rays_o.is_cuda:
            from . import fused
            if fused.supports_coarse(self):
                return fused.forward_coarse(self, rays_o, rays_d, viewdirs, global_step, **render_kwargs)
        return self._forward_coarse_composed(rays_o, rays_d, viewdirs, global_step, **render_kwargs)

    def _composite(self, weights, rgb, ray_id, N, bg, normal, step_id, dist, render_grad, render_depth):
        sig = torch.sigmoid(rgb)
        w1 = weights.unsqueeze(-1)
        rgb_marched = segment_sum(w1 * rgb, ray_id, N)
        cum_weights = segment_sum(w1, ray_id, N)
        sigmoid_rgb = segment_sum(w1 * sig, ray_id, N)
        rgb_marched = (rgb_marched + (1 - cum_weights) * bg).clamp(0, 1)
        sigmoid_rgb = (sigmoid_rgb + (1 - cum_weights) * bg).clamp(0, 1)
        normal_marched = segment_sum(w1 * normal, ray_id, N) if render_grad else None
        depth = disp = None
        if render_depth:
            with torch.no_grad():
                depth = segment_sum(weights * step_id * dist, ray_id, N)
                disp = 1 / depth
        return rgb_marched, sigmoid_rgb, normal_marched, depth, disp

    def _forward_fine_composed(self, rays_o, rays_d, viewdirs, global_step=20000, **render_kwargs):
        N = len(rays_o)
        lo, hi = self.xyz_min, self.xyz_max
        ray_pts, ray_id, step_id, mask_outbbox, _ = self.sample_ray(rays_o=rays_o, rays_d=rays_d, **render_kwargs)
        if self.mask_cache is not None:
            m = self.mask_cache(ray_pts)
            ray_pts, ray_id, step_id = ray_pts[m], ray_id[m], step_id[m]
            mask_outbbox[~mask_outbbox] |= ~m
        sdf_grid = self.smooth_conv(self.sdf.grid) if self.smooth_sdf else self.sdf.grid
        sdf, gradient, _ = grid_sampler(ray_pts, sdf_grid, lo, hi, self.voxel_size, sample_ret=True, sample_grad=True)
        dist = render_kwargs['stepsize'] * self.voxel_size
        s_val, alpha = self.neus_alpha_from_sdf_scatter(viewdirs, ray_id, dist.to(sdf.device), sdf, gradient,
                                                        global_step=global_step, is_train=global_step is not None)
        mask = None
        viewdirs_pts = viewdirs[ray_id]
        if self.fast_color_thres > 0:
            mask = alpha > self.fast_color_thres
            alpha, ray_id, viewdirs_pts, ray_pts = alpha[mask], ray_id[mask], viewdirs_pts[mask], ray_pts[mask]
            step_id, gradient, sdf = step_id[mask], gradient[mask], sdf[mask]
        weights, alphainv_last = Alphas2Weights.apply(alpha, ray_id, N)
        if self.fast_color_thres > 0:
            mask = weights > self.fast_color_thres
            weights, alpha, ray_pts, viewdirs_pts = weights[mask], alpha[mask], ray_pts[mask], viewdirs_pts[mask]
            ray_id, step_id, gradient, sdf = ray_id[mask], step_id[mask], gradient[mask], sdf[mask]
        normal = l2_normalize(gradient / (gradient.norm(dim=-1, keepdim=True) + 1e-7))
        xyz_emb = posenc((ray_pts - lo) / (hi - lo), self.posfreq)
        k0 = self.k0(ray_pts)
        grad_inds = sorted(set(self.grad_feat + self.k_grad_feat))
        assert grad_inds == sorted(set(self.sdf_feat + self.k_sdf_feat))
        hier = [sdf[:, None]] if self.center_sdf else []
        if len(grad_inds) > 0:
            all_feat, all_grad = sample_sdfs(ray_pts, sdf_grid, lo, hi, self.voxel_size, grad_inds,
                                             use_grad_norm=self.use_grad_norm)
            hier += [all_feat, all_grad]
        assert len(self.k_grad_feat) == 1 and self.k_grad_feat[0] == 1.0 and len(self.k_sdf_feat) == 0
        feats = [k0, xyz_emb]
        if self.use_viewdir:
            feats.append(posenc(viewdirs, self.viewfreq).flatten(0, -2)[ray_id])
        rgb_feat = self.rgbnet(torch.cat([*feats, *hier, gradient], dim=-1))
        reflect_r = viewdirs_pts - 2. * torch.sum(viewdirs_pts * normal, dim=-1, keepdim=True) * normal
        rgb = torch.sigmoid(self.refnet(torch.cat([rgb_feat, posenc(reflect_r, self.reffreq)], dim=-1)))
        rgb_marched, sigmoid_rgb, normal_marched, depth, disp = self._composite(
            weights, rgb, ray_id, N, render_kwargs['bg'], normal, step_id, dist,
            render_kwargs.get('render_grad', False), render_kwargs.get('render_depth', False))
        return {'alphainv_cum': alphainv_last, 'weights': weights, 'ray_id': ray_id, 'viewdirs': viewdirs[ray_id],
                'rgb_marched': rgb_marched, 'sigmoid_rgb': sigmoid_rgb, 'normal_marched': normal_marched,
                'normal': normal, 'raw_alpha': alpha, 'raw_rgb': rgb, 'depth': depth, 'disp': disp, 'mask': mask,
                'mask_outbbox': mask_outbbox, 'gradient': gradient, 's_val': s_val}

    def _forward_coarse_composed(self, rays_o, rays_d, viewdirs, global_step=20000, **render_kwargs):
        N = len(rays_o)
        lo, hi = self.xyz_min, self.xyz_max
        ray_pts, ray_id, step_id, mask_outbbox, _ = self.sample_ray_cuda(rays_o=rays_o, rays_d=rays_d, **render_kwargs)
        viewdirs_pts = viewdirs[ray_id]
        if self.stage == 'coarse' and self.mask_cache is not None:
            m = self.mask_cache(ray_pts)
            ray_pts, ray_id, viewdirs_pts, step_id = ray_pts[m], ray_id[m], viewdirs_pts[m], step_id[m]
            mask_outbbox[~mask_outbbox] |= ~m
        if self.inc_mask is not None:
            m = self.inc_mask(ray_pts)
            ray_pts, ray_id, viewdirs_pts, step_id = ray_pts[m], ray_id[m], viewdirs_pts[m], step_id[m]
        sdf_grid = self.smooth_conv(self.sdf.grid) if self.smooth_sdf else self.sdf.grid
        sdf = grid_sampler(ray_pts, sdf_grid, lo, hi)
        self.gradient = self.neus_sdf_gradient(sdf=self.sdf.grid)
        gradient = grid_sampler(ray_pts, self.gradient, lo, hi)
        dist = render_kwargs['stepsize'] * self.voxel_size.to(ray_id.device)
        s_val, alpha = self.neus_alpha_from_sdf_scatter(viewdirs, ray_id, dist, sdf, gradient,
                                                        global_step=global_step, is_train=global_step is not None)
        weights, alphainv_last = Alphas2Weights.apply(alpha, ray_id, N)
        mask = None
        if self.fast_color_thres > 0:
            mask = weights > self.fast_color_thres
            ray_pts, ray_id, viewdirs_pts, step_id = ray_pts[mask], ray_id[mask], viewdirs_pts[mask], step_id[mask]
            alpha, gradient = alpha[mask], gradient[mask]
        weights, alphainv_last = Alphas2Weights.apply(alpha, ray_id, N)   # second pass on the compacted list (:990)
        normal = l2_normalize(gradient / (gradient.norm(dim=-1, keepdim=True) + 1e-7))
        xyz_emb = posenc((ray_pts - lo) / (hi - lo), self.posfreq)
        k0 = self.k0(ray_pts)
        reflect_r = viewdirs_pts - 2. * torch.sum(viewdirs_pts * normal, dim=-1, keepdim=True) * normal
        feats = [k0, xyz_emb, posenc(reflect_r, self.reffreq), normal]
        if self.use_viewdir:
            feats.append(posenc(viewdirs, self.viewfreq).flatten(0, -2)[ray_id])
        rgb = torch.sigmoid(self.refnet(torch.cat(feats, dim=-1)))
        rgb_marched, sigmoid_rgb, normal_marched, depth, disp = self._composite(
            weights, rgb, ray_id, N, render_kwargs['bg'], normal, step_id, dist,
            render_kwargs.get('render_grad', False), render_kwargs.get('render_depth', True))
        return {'alphainv_cum': alphainv_last, 'weights': weights, 'ray_id': ray_id, 'viewdirs': viewdirs[ray_id],
                'rgb_marched': rgb_marched, 'sigmoid_rgb': sigmoid_rgb, 'normal_marched': normal_marched,
                'normal': normal, 'raw_alpha': alpha, 'raw_rgb': rgb, 'depth': depth, 'disp': disp, 'mask': mask,
                'mask_outbbox': mask_outbbox, 'gradient': gradient, 's_val': s_val}


def _nerf_extract_geometry(self, bound_min, bound_max, resolution=128, threshold=0.0, **kwargs):
    """model/nerf.py:1157-1170: field = trilinear lookup of -sdf (smoothed in the coarse stages) on a dense lattice."""
    from .extract_geometry import extract_geometry
    sdf_grid = self.smooth_conv(self.sdf.grid) if self.smooth_sdf else self.sdf.grid
    neg = (-sdf_grid).detach().contiguous()
    if resolution is None:
        resolution = int(self.world_size[0])
    return extract_geometry(bound_min, bound_max, resolution=resolution, threshold=threshold,
                            query_func=lambda pts: grid_sampler(pts.to(neg.device), neg, self.xyz_min, self.xyz_max))


def _nerf_extract_fields(self, bound_min, bound_max, resolution=128):
    """The field half of extract_geometry (no PyMCubes needed): float32 ndarray [res,res,res] of -sdf."""
    from .extract_geometry import extract_fields
    sdf_grid = self.smooth_conv(self.sdf.grid) if self.smooth_sdf else self.sdf.grid
    neg = (-sdf_grid).detach().contiguous()
    return extract_fields(bound_min, bound_max, resolution,
                          lambda pts: grid_sampler(pts.to(neg.device), neg, self.xyz_min, self.xyz_max))


nerf.extract_geometry = _nerf_extract_geometry
nerf.extract_fields = _nerf_extract_fields


def total_variation(v, mask=None):
    """model/nerf.py:1212-1221: sum over the three grid axes of |v[i+1] - v[i]| over the pairs whose two voxels are inside
    `mask`, divided by 3 and by mask.sum() -- by v.sum() without a mask (sic; dvgo's variant takes per-axis means).
    CUDA grids: one HIP value pass + one HIP gradient pass (dense.grid_tv_loss, csrc/tvloss.hip)."""
    if v.is_cuda:
        from . import dense
        return dense.grid_tv_loss(v, mask, per_axis_mean=False)
    return _pair_tv(v, mask, per_axis_mean=False)


def _pair_tv(v, mask, per_axis_mean):
    """Host-tensor form of the two total_variation variants (the CPU tests' models; same value as the reference's
    diff / boolean-index expression up to summation order)."""
    per_axis = []
    for axis in (2, 3, 4):
        n = v.shape[axis] - 1
        d = (v.narrow(axis, 1, n) - v.narrow(axis, 0, n)).abs()
        if mask is None:
            per_axis.append((d.sum(), d.numel()))
        else:
            both = (mask.narrow(axis, 1, n) & mask.narrow(axis, 0, n)).expand_as(d)
            per_axis.append(((d * both).sum(), both.sum()))
    if per_axis_mean:
        return sum(s / c for s, c in per_axis) / 3
    return sum(s for s, _ in per_axis) / 3 / (v.sum() if mask is None else mask.sum())
