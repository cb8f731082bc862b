"""ctypes binding of libfgs_hip.so (the C ABI declared in include/fgs_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol is absent this
module raises, and every operator built on it raises with it.  Build it with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C fgs-nerf_amd/csrc``.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfgs_hip.so")

P, F32, I64, I32, F64 = c_void_p, c_float, c_int64, c_int, c_double

# The ABI this binding was written against (include/fgs_hip.h FGS_ABI_VERSION).  lib() refuses a library built from another
# header: a stale libfgs_hip.so whose symbol NAMES all exist would otherwise be called with this table's argument lists.
ABI_VERSION = 14

# name -> argtypes (all functions return int); mirrors include/fgs_hip.h one to one
_SIGNATURES = {
    "fgs_infer_t_minmax": [P, P, P, P, F32, F32, I64, P, P, P],
    "fgs_infer_n_samples": [P, P, P, F32, I64, P, P],
    "fgs_infer_ray_start_dir": [P, P, P, I64, P, P, P],
    "fgs_sample_count": [P, P, P, P, F32, F32, F32, I64, P, P, P, P, P],
    "fgs_sample_emit": [P, P, P, P, F32, I64, P, P, I64, P, P, P, P, P],
    "fgs_copy_f32": [P, P, I64, P],
    "fgs_gather_batch": [P, I64, I64, P, P, P, P, P, P],
    "fgs_sample_ndc_pts": [P, P, P, P, I64, I64, P, P, P],
    "fgs_sample_bg_pts": [P, P, P, F32, I64, I64, P, P],
    "fgs_maskcache_lookup": [P, P, P, P, I32, I32, I32, I64, P, P],
    "fgs_raw2alpha": [P, F32, F32, P, I64, P, P, P],
    "fgs_raw2alpha_bwd": [P, P, F32, P, I64, P, P],
    "fgs_alpha2weight_fwd": [P, P, I64, I64, P, P, P, P, P, P],
    "fgs_alpha2weight_bwd": [P, P, P, P, P, P, I64, I64, P, P, P, P],
    "fgs_tv_add_grad": [P, P, P, F32, F32, F32, I32, I64, I64, I64, I64, I64, I64, I64, I64, P],
    "fgs_adam_upd": [P, P, P, P, P, I64, I32, F32, F32, F32, F32, I32, P],
    "fgs_trilerp_fwd": [P, I64, I64, I64, I64, I64, I64, I64, I64, P, P, P, I64, P, P],
    "fgs_trilerp_bwd": [P, I64, I64, I64, I64, I64, I64, I64, I64, P, P, P, I64, P, P],
    "fgs_sdf_taps_fwd": [P, I64, I64, I64, P, P, P, I64, P, I32, P, P, P],
    "fgs_sdf_taps_bwd": [P, I64, I64, I64, P, P, P, I64, P, I32, P, P],
    "fgs_gemm_f32": [I32, I64, I64, I64, P, I64, P, I64, P, I64, P, I32, P, I64, P, P, I64, P, P],
    "fgs_linear_bwd_f32": [I64, I64, I64, P, I64, P, I64, P, I64, P, I64, P, I64, P, P, I64, P],
    "fgs_mlp_fwd_f32": [I64, I32, P, I64, I32, P, I64, I32, P, P, P, P, P, P, P, P],
    "fgs_mlp_chain_f32": [I64, I32, P, I64, I32, P, I64, I32, P, P, P, P, P, P, P, P, P, P, P, P, P],
    "fgs_transpose_multi": [I32, P, P, P, P, P, P, P],
    "fgs_pad_cols_multi": [I32, P, P, P, P, P, P, P],
    "fgs_copy_cols_multi": [I32, P, P, P, P, P, P, P, P],
    "fgs_debug_pad_cols_old_indexing": [P, I32, I32, I64, P, I64, P],
    "fgs_step_scalars_tick": [P, I32, I32, P, P, I32, P, P],
    "fgs_step_scalars_tick2": [P, I32, I32, P, P, I32, P, I32, P, P, P, P],
    "fgs_count_guard": [P, I64, I64, P, P, P],
    "fgs_box_mask_fill": [P, I32, I32, I32, P, P],
    "fgs_adam_upd_dev": [P, P, P, P, P, I64, P, I32, F32, F32, F32, F32, I32, P, P],
    "fgs_adam_upd_multi_dev": [I32, P, P, P, P, P, P, P, P, P, F32, F32, F32, P, P],
    "fgs_mlp_rc_chain": [I32, I64, I32, P, P, I64, I32, P, I64, P, P],
    "fgs_mlp_rc2_chain": [I32, I64, I32, P, P, I64, I32, P, I64, I32, P, P],
    "fgs_mlp_rc2_pack": [I32, I32, P, I32, P, I64, I32, I32, P, I32, P, I64, P],
    "fgs_mlp_wgrad_debug_stamps": [P],
    "fgs_mlp_rc_debug_stamps": [P],
    "fgs_mlp_wgrad": [I64, I32, P, P, P],
    "fgs_mlp_wgrad_ws": [I64, I32, P, P, I64, P, P],
    "fgs_exclusive_scan_i64": [P, I64, P, P],
    "fgs_exclusive_scan_guard_i64": [P, I64, P, I64, P, P, P],
    "fgs_march_fine_fwd": [P, P, P, I64, P, P, I32, I32, I32, F32, F32, F32, F32, P, F32, F32, F32,
                           P, P, P, I32, I32, I32, F32, I32, P, P, P, P, P, P, P, P, P, P, P, P, P, P],
    "fgs_march_count": [P, P, P, I64, P, P, I32, I32, I32, F32, F32, F32, F32, P, F32, F32, F32,
                        P, P, P, I32, I32, I32, F32, I32, P, P, P, P],
    "fgs_surv_compact": [I64, I64, P, I32, P, P, P, P, P, P, P, P, P, P, I32, I32, I32, F32, F32, F32,
                         P, P, P, P, P, P, P, P, P, P],
    "fgs_march_fine_bwd": [P, P, P, I64, P, P, I32, I32, I32, F32, F32, F32, F32, F32, F32, I32,
                           P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P],
    "fgs_sdf_scatter_surv": [I64, P, P, P, I32, I32, I32, F32, P, P, P, P, P, P, P, P, P],
    "fgs_brick_flags": [P, I32, I32, I32, I32, P, P],
    "fgs_brick_gather": [P, I32, I32, I32, I32, P, I64, P, P],
    "fgs_brick_scatter": [P, I32, I32, I32, I32, P, I64, P, F32, P],
    "fgs_brick_flags_pts": [P, I64, P, P, I32, I32, I32, P, P, P],
    "fgs_brick_compact": [P, I64, P, P, P],
    "fgs_brick_gather_dev": [P, I32, I32, I32, I32, P, P, I64, P, P],
    "fgs_brick_scatter_dev": [P, I32, I32, I32, I32, P, P, I64, P, F32, P],
    "fgs_brick_count_guard": [P, I64, P, P, P, P],
    "fgs_tv_loss_value": [P, P, I64, I64, I64, I64, I64, I64, I64, I64, P, I32, F64, P, P, I64, P, P, P, P],
    "fgs_tv_loss_grad": [P, P, I64, I64, I64, I64, I64, I64, I64, I64, P, P, P, I32, P],
    "fgs_brick_masks_pts": [P, I64, P, P, I32, I32, I32, P, P, P],
    "fgs_adam_upd_voxels": [P, P, P, P, I32, I32, I32, I32, P, I32, F32, F32, F32, F32, P, P, P],
    "fgs_adam_upd_bricks": [P, P, P, P, I32, I32, I32, I32, P, P, I64, P, I32, F32, F32, F32, F32, P, P, P],
    "fgs_adam_upd_multi": [I32, P, P, P, P, P, P, P, P, F32, F32, F32, P],
    "fgs_fine_loss_fwd": [I64, I64, P, P, P, P, P, P, P, P, P, P, P, P, I64, P, P],
    "fgs_fine_loss_bwd": [I64, I64, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P],
    "fgs_fine_render_loss": [I64, I64, P, P, P, P, P, F32, F32, P, P, P, P, P, P, P, P, P, P, P, P, P, I64, P, P, P, P, P, P, P],
    "fgs_feat_fine_fwd": [I64, P, P, P, P, P, P, P, I32, I32, I32, F32, P, P, P, P, I64, I64, I64, I64, P, P, P, P, P],
    "fgs_feat_fine_bwd": [I64, P, P, P, P, P, P, P, I32, I32, I32, F32, P, P, P, P, P, P, P, P, P, I64, I64, I64, I64,
                          P, P, P, P],
    "fgs_head_fwd": [P, I64, I32, I64, P, P, P, P, P],
    "fgs_head_bwd": [P, I64, I32, I64, P, P, P, P, P, P, P, P, P],
    "fgs_composite_fwd": [I64, P, P, P, P, P, F32, F32, P, P, P, P, P, P, P],
    "fgs_composite_bwd": [I64, P, P, P, P, P, P, P, P, P, F32, P, P, P, P],
    "fgs_smooth3d_fwd": [P, I32, I32, I32, I32, P, P, P],
    "fgs_smooth3d_bwd": [P, I64, I32, I32, I32, I32, P, P, P, P],
    "fgs_smooth_tv_loss": [P, I32, I32, I32, P, P, P, F32, P, P, I64, P, P, P],
    "fgs_sdf_gradvol_fwd": [P, I32, I32, I32, F32, I32, P, P, P, P],
    "fgs_sdf_gradvol_bwd": [P, I64, I64, I32, I32, I32, F32, I32, P, I32, P, P],
    "fgs_march_coarse_fwd": [P, P, P, I64, P, P, I32, I32, I32, F32, F32, F32, P, P, P, F32, F32, F32,
                             P, P, P, I32, I32, I32, F32, P, I32, I32, I32, P, P, I32,
                             P, P, P, P, P, P, P, P, P, P, P, P, P, P],
    "fgs_march_coarse_bwd": [P, P, P, I64, P, P, I32, I32, I32, F32, F32, F32, F32, F32, I32,
                             P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P],
    "fgs_feat_coarse_fwd": [I64, P, P, P, P, P, P, I32, I32, I32, P, P, I64, I64, I64, I64, P, P, P, P],
    "fgs_feat_coarse_bwd": [I64, P, P, P, P, P, P, I32, I32, I32, P, P, P, P, P, I64, I64, I64, I64, P, P, P],
    "fgs_ide_fwd": [P, P, P, P, I32, I32, I64, P, P],
    "fgs_ide_bwd": [P, P, P, P, I32, I32, I64, P, P, P, P],
    "fgs_mc_count": [P, I32, I32, I32, F32, P, P, P, P, P],
    "fgs_mc_emit": [P, I32, I32, I32, F32, P, P, P, P, P, P, I64, I64, P, P, P],
}

_lib = None


class RcLayer(ctypes.Structure):
    """fgs_rc_layer_t (include/fgs_hip.h): one Linear layer of a register-resident MLP chain."""
    _fields_ = [("W", c_void_p), ("ldw", c_int64), ("n_out", c_int), ("n_in", c_int),
                ("bias", c_void_p), ("relu", c_int),
                ("mask_bits", c_void_p),
                ("out", c_void_p), ("ldo", c_int64), ("n_store", c_int),
                ("ext", c_void_p), ("ld_ext", c_int64), ("ext_cols", c_int)]


class Rc2Layer(ctypes.Structure):
    """fgs_rc2_layer_t (include/fgs_hip.h): one layer of a feature-split MLP chain (csrc/mlp_rc2.hip); `side` marks a narrow
    product of the current carried input that only goes to `out`."""
    _fields_ = RcLayer._fields_ + [("side", c_int)]


class WgradItem(ctypes.Structure):
    """fgs_wgrad_item_t (include/fgs_hip.h): one weight-gradient product dW += dY^T X (+ bias gradient)."""
    _fields_ = [("dY", c_void_p), ("ld_dy", c_int64), ("n_out", c_int),
                ("X", c_void_p), ("ld_x", c_int64), ("n_in", c_int),
                ("dW", c_void_p), ("ld_dw", c_int64),
                ("dbias", c_void_p)]


class Dyn(ctypes.Structure):
    """fgs_dyn_t (include/fgs_hip.h): the device-resident values a launch may read instead of its host arguments."""
    _fields_ = [("row_count", c_void_p), ("inv_s", c_void_p), ("dx0_compact", c_int),
                ("stamps", c_void_p), ("stamp_step", c_void_p), ("stamp_slots", c_int64), ("stamp_stride", c_int64)]


def dyn(row_count=None, inv_s=None, compact: bool = False, stamps=None):
    """`const fgs_dyn_t *` argument for the entry points that take one: None (NULL) when nothing is device-resident, else a
    pointer to a struct holding the raw device addresses (ints) of the survivor count / NeuS 1/s and the compact-dX0 flag.
    `stamps`: (address of this launch's region in slot 0, address of the device step counter or None, slots, stride in uint64)
    -- the in-kernel wall-clock stamps of the matrix-core launches (measurement only).
    The C side copies the members into its kernel arguments before it returns: the struct need not outlive the call."""
    if row_count is None and inv_s is None and not compact and stamps is None:
        return None
    if stamps is None:
        return ctypes.byref(Dyn(row_count, inv_s, int(bool(compact)), None, None, 0, 0))
    return ctypes.byref(Dyn(row_count, inv_s, int(bool(compact)), stamps[0], stamps[1], int(stamps[2]), int(stamps[3])))


class FgsError(RuntimeError):
    """A libfgs_hip.so entry point returned a non-zero status."""


def exported_symbols():
    """Every symbol include/fgs_hip.h declares (used by the CPU-side ABI test)."""
    return sorted(list(_SIGNATURES) + ["fgs_last_error", "fgs_version", "fgs_device_info", "fgs_gemm_workspace_bytes",
                                          "fgs_mc_num_blocks", "fgs_head_bwd_scratch_floats", "fgs_mlp_rc_image_floats",
                                          "fgs_mlp_rc2_image_floats", "fgs_adam_step_size", "fgs_smooth_tv_scratch_floats",
                                          "fgs_tv_loss_scratch_doubles", "fgs_mlp_wgrad_ws_floats"])


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FgsError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run __graft_entry__.build() or `make -C fgs-nerf_amd/csrc`). There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        handle.fgs_version.restype = c_int
        handle.fgs_version.argtypes = []
        built = int(handle.fgs_version())
        if built != ABI_VERSION:
            raise FgsError(f"{LIB_PATH} was built for ABI version {built}, this binding is written against {ABI_VERSION}: "
                           "rebuild it (python -c 'import __graft_entry__ as g; g.build()')")
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the ABI and the header drifted apart
            fn.argtypes = argtypes
            fn.restype = c_int
        handle.fgs_last_error.restype = c_char_p
        handle.fgs_last_error.argtypes = []
        handle.fgs_gemm_workspace_bytes.restype = c_int64
        handle.fgs_gemm_workspace_bytes.argtypes = []
        handle.fgs_head_bwd_scratch_floats.restype = c_int64
        handle.fgs_head_bwd_scratch_floats.argtypes = [c_int]
        handle.fgs_adam_step_size.restype = c_float
        handle.fgs_adam_step_size.argtypes = [c_int, c_float, c_float, c_float]
        handle.fgs_mlp_rc_image_floats.restype = c_int64
        handle.fgs_mlp_rc_image_floats.argtypes = [c_int, c_int, c_void_p]
        handle.fgs_mlp_rc2_image_floats.restype = c_int64
        handle.fgs_mlp_rc2_image_floats.argtypes = [c_int, c_int, c_void_p]
        handle.fgs_smooth_tv_scratch_floats.restype = c_int64
        handle.fgs_smooth_tv_scratch_floats.argtypes = [c_int, c_int, c_int]
        handle.fgs_tv_loss_scratch_doubles.restype = c_int64
        handle.fgs_tv_loss_scratch_doubles.argtypes = []
        handle.fgs_mlp_wgrad_ws_floats.restype = c_int64
        handle.fgs_mlp_wgrad_ws_floats.argtypes = []
        handle.fgs_mc_num_blocks.restype = c_int64
        handle.fgs_mc_num_blocks.argtypes = [c_int, c_int, c_int]
        handle.fgs_device_info.argtypes = [c_int, c_char_p, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int),
                                           ctypes.POINTER(c_int64)]
        handle.fgs_device_info.restype = c_int
        _lib = handle
    return _lib


def call(name: str, *args) -> None:
    """Invoke an int-returning entry point; raise FgsError carrying fgs_last_error() on failure."""
    handle = lib()
    fn = getattr(handle, name)
    if len(args) != len(fn.argtypes):      # (ctypes itself accepts surplus arguments to a cdecl function without a word)
        raise FgsError(f"{name}: {len(args)} arguments given, the ABI takes {len(fn.argtypes)}")
    rc = fn(*args)
    if rc != 0:
        msg = handle.fgs_last_error()
        raise FgsError(f"{name} failed with code {rc}: {msg.decode() if msg else ''}")


def ptr(t):
    """Raw device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream() -> int:
    """PyTorch-ROCm's current HIP stream as an integer handle (the raw accessor: ~0.5 us instead of the ~13 us of
    torch.cuda.current_stream(), and the step asks a dozen times)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def check_input(x: torch.Tensor, name: str, dtype=None) -> torch.Tensor:
    """The reference's CHECK_INPUT (model/cuda/render_utils.cpp:46-48): device + contiguity, RuntimeError."""
    if not isinstance(x, torch.Tensor):
        raise RuntimeError(f"{name} must be a tensor")
    if not x.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not x.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    if dtype is not None and x.dtype != dtype:
        raise RuntimeError(f"{name} must have dtype {dtype}, got {x.dtype}")
    return x


def device_info(device: int = 0):
    name = ctypes.create_string_buffer(256)
    cu, wave, lds = c_int(0), c_int(0), c_int64(0)
    rc = lib().fgs_device_info(device, name, 256, ctypes.byref(cu), ctypes.byref(wave), ctypes.byref(lds))
    if rc != 0:
        raise FgsError(f"fgs_device_info failed: {lib().fgs_last_error().decode()}")
    return {"name": name.value.decode(), "cu_count": cu.value, "wave_size": wave.value, "lds_bytes": lds.value}
