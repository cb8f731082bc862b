"""Stage hand-off files: what the reference writes at model/nerf_training.py:522-531 and reads back through
model/utils.py:26-98 (`load_grid_data`, `load_checkpoint`, `load_model`, `load_weight_by_name`) and
model/nerf_training.py:40-58 (`compute_bbox_by_coarse_geo`).  Same call signatures, same file layout

    {'global_step', 'model_kwargs', 'MaskCache_kwargs', 'model_state_dict', 'optimizer_state_dict'}

so that a checkpoint written by the reference loads here and vice versa.  Two deliberate differences: files are read with the
restricted unpickler (`nerf.load_checkpoint_file`: tensors, numbers, numpy arrays -- nothing in a file is executed), and values
are COPIED into the parameters (`copy_`), which keeps this build's channel-last storage of multi-channel grids whatever
layout the file's tensors have; the reference rebinds `.data`, which would silently switch the layout.
"""
from __future__ import annotations

import os

import torch

from .nerf import load_checkpoint_file

__all__ = ["save_checkpoint", "load_grid_data", "load_checkpoint", "load_model", "load_weight_by_name",
           "compute_bbox_by_coarse_geo"]


def save_checkpoint(path, model, optimizer, global_step) -> None:
    """model/nerf_training.py:522-531 (the caller runs `model.set_sdf_mask()` first, as the reference does)."""
    torch.save({'global_step': int(global_step), 'model_kwargs': model.get_kwargs(),
                'MaskCache_kwargs': model.get_MaskCache_kwargs(), 'model_state_dict': model.state_dict(),
                'optimizer_state_dict': optimizer.state_dict()}, path)


def _assign(param, value) -> None:
    """value -> param, shape checked, layout and device of `param` kept."""
    if tuple(param.shape) != tuple(value.shape):
        raise ValueError(f"shape mismatch: checkpoint {tuple(value.shape)} vs model {tuple(param.shape)}")
    with torch.no_grad():
        param.data.copy_(value.to(param.device))


def _grid_entry(state, name):
    """model/utils.py:30-31: a grid is stored under `name` or `name + '.grid'`."""
    if name in state:
        return state[name]
    return state[name + '.grid']


def load_grid_data(model, ckpt_path, deduce=1, name='density', return_raw=False):
    """model/utils.py:26-39: one grid (`sdf`, `k0`, `density`, ...) out of a stage file; `return_raw` hands back the stored
    tensor (the fine stage resamples the coarse SDF from it: model/nerf_training.py:125), otherwise it is loaded into
    `getattr(model, name)`.  A stored grid of another resolution is an error here (the reference rebinds `.data` and
    leaves the model inconsistent)."""
    value = _grid_entry(load_checkpoint_file(ckpt_path)['model_state_dict'], name)
    if return_raw:
        return value
    module = getattr(model, name)
    _assign(module.grid if hasattr(module, 'grid') else module, value)
    return model


def load_checkpoint(model, optimizer, ckpt_path, no_reload_optimizer, stage='coarse', num_voxels=0, strict=True):
    """model/utils.py:42-60: resume a stage.  For the fine stage the file's mask-cache volume is dropped (the model owns a
    fresh one) and the grids are rescaled to `num_voxels` after loading."""
    ckpt = load_checkpoint_file(ckpt_path)
    state = dict(ckpt['model_state_dict'])
    if stage == 'fine':
        state.pop('mask_cache.density', None)
    model.load_state_dict(state, strict=strict)
    if stage == 'fine':
        model.scale_volume_grid(num_voxels)
    if not no_reload_optimizer:
        try:
            optimizer.load_state_dict(ckpt['optimizer_state_dict'])
        except (ValueError, KeyError, RuntimeError) as err:
            if strict:
                raise ValueError(f"optimizer state of {ckpt_path} does not fit this optimizer") from err
    return model, optimizer, ckpt['global_step']


def load_model(model_class, ckpt_path, new_kwargs=None, strict=False):
    """model/utils.py:63-86: rebuild a model from a stage file (`model_kwargs`, optionally overridden), pointing its mask
    cache at the `geometry_searching_last.tar` that lies next to the file, and load the weights -- exactly if possible,
    otherwise (strict=False) whatever matches.  Returns (model, global_step)."""
    ckpt = load_checkpoint_file(ckpt_path)
    kwargs = dict(ckpt['model_kwargs'])
    if new_kwargs:
        kwargs.update(new_kwargs)
    kwargs['mask_cache_path'] = os.path.join(os.path.dirname(ckpt_path) or '.', 'geometry_searching_last.tar')
    if not os.path.exists(kwargs['mask_cache_path']):
        kwargs['mask_cache_path'] = None          # (the reference would fail inside MaskCache; a missing file means no cache)
    model = model_class(**kwargs)
    try:
        model.load_state_dict(ckpt['model_state_dict'], strict=True)
    except RuntimeError:
        if strict:
            raise
        model.load_state_dict(ckpt['model_state_dict'], strict=False)
    return model, ckpt['global_step']


def load_weight_by_name(model, ckpt_path, deduce=1, name='density', return_raw=False):
    """model/utils.py:89-97: every parameter whose qualified name contains `name` and exists in the file is overwritten
    with the file's value (e.g. name='rgbnet' warm-starts the colour MLP of the next stage)."""
    state = load_checkpoint_file(ckpt_path)['model_state_dict']
    for qualified, param in model.named_parameters():
        if name in qualified and qualified in state:
            _assign(param, state[qualified])
    return model


@torch.no_grad()
def compute_bbox_by_coarse_geo(model_class, model_path, thres):
    """model/nerf_training.py:40-58: the axis-aligned box around the voxels the coarse stage marked in `sdf_mask`
    (> 0), in world coordinates: voxel (i, j, k) of an [X, Y, Z] grid sits at xyz_min + (i/(X-1), j/(Y-1), k/(Z-1)) *
    (xyz_max - xyz_min).  Instead of materialising the X*Y*Z*3 coordinate lattice, the occupied index range per axis is
    read off the mask's projections; the corner coordinates then use the reference's own blend
    `xyz_min * (1 - t) + xyz_max * t` with t from `torch.linspace(0, 1, n)`, so they are bit-identical to its amin / amax."""
    st = load_checkpoint_file(model_path)
    lo = torch.tensor(st['model_kwargs']['xyz_min'])
    hi = torch.tensor(st['model_kwargs']['xyz_max'])
    occupied = (st['model_state_dict']['sdf_mask.grid'] > 0)[0, 0]
    if not bool(occupied.any()):
        raise ValueError(f"{model_path}: sdf_mask marks no voxel")
    out_min, out_max = torch.empty(3), torch.empty(3)
    for axis in range(3):
        others = tuple(a for a in range(3) if a != axis)
        idx = occupied.any(dim=others).nonzero().flatten()
        t = torch.linspace(0, 1, occupied.shape[axis])[idx]
        coords = lo[axis] * (1 - t) + hi[axis] * t        # (all occupied indices: the fp32 blend need not be monotonic)
        out_min[axis], out_max[axis] = coords.min(), coords.max()
    return out_min, out_max
