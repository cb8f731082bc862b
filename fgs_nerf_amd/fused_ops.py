"""Thin tensor-level wrappers over the fused-path entry points of libfgs_hip.so (GEMM, march, features, compositing).

Everything here takes pre-allocated, correctly laid-out CUDA tensors and only forwards pointers, sizes and the current
HIP stream; argument checking that needs the device happens in the C ABI (negative FGS_E_* codes -> FgsError).
"""
from __future__ import annotations

import os

import torch

from ._lib import call, lib, ptr, stream, dyn

GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2

_WORKSPACE = {}   # (device index, stream handle) -> zero-initialised scratch of fgs_gemm_workspace_bytes()


def gemm_workspace(device: torch.device) -> torch.Tensor:
    """Scratch for the stream-K form of the NT / NN products: one buffer per (device, stream), zeroed once (the kernels
    leave the flag words zero)."""
    key = (device.index, stream())
    ws = _WORKSPACE.get(key)
    if ws is None:
        ws = torch.zeros(int(lib().fgs_gemm_workspace_bytes()), dtype=torch.uint8, device=device)
        _WORKSPACE[key] = ws
    return ws


def gemm(op: int, A: torch.Tensor, B: torch.Tensor, C: torch.Tensor, M: int, N: int, K: int, bias=None, relu=False,
         mask=None, colsum=None, stream_k: bool = False, rows_dev=None, stamp=None) -> torch.Tensor:
    """C = op(A, B) with the epilogues of include/fgs_hip.h fgs_gemm_f32.  A, B, C, mask are 2-D row-major views
    (stride(1) == 1); leading dimensions are taken from stride(0).  `stream_k` selects the opt-in stream-K grid for
    NT / NN (measured: no faster than one tile per workgroup at the MLP shapes, see csrc/gemm_f32.hip).  `rows_dev`: device address of the
    actual row count (NT / NN: M is then the capacity of A and C; fgs_dyn_t.row_count).  `stamp`: (label, flop) -- the launch
    takes part in the in-kernel timing of a step (STAMPS)."""
    for t in (A, B, C) + ((mask,) if mask is not None else ()):
        if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1):
            raise RuntimeError("gemm operands must be 2-D float32 CUDA tensors with unit column stride")
    ws = gemm_workspace(C.device) if (stream_k and op != GEMM_TN) else None
    call("fgs_gemm_f32", op, M, N, K, ptr(A), A.stride(0), ptr(B), B.stride(0), ptr(C), C.stride(0), ptr(bias),
         int(bool(relu)), ptr(mask), 0 if mask is None else mask.stride(0), ptr(colsum), ptr(ws),
         0 if ws is None else ws.numel(),
         dyn(row_count=rows_dev, stamps=None if stamp is None else _stamp_arg(stamp[0], stamp[1], "gemm")), stream())
    return C


def linear_bwd(dY: torch.Tensor, W: torch.Tensor, X: torch.Tensor, dX: torch.Tensor, dW: torch.Tensor, M: int, N_out: int,
               K_in: int, mask=None, colsum=None) -> None:
    """Backward of a Linear layer in one launch (include/fgs_hip.h fgs_linear_bwd_f32): dX = (dY W) * (mask > 0),
    colsum += column sums of dX, dW += dY^T X.  All 2-D row-major float32 views; dW must start at zero."""
    for t in (dY, W, X, dX, dW) + ((mask,) if mask is not None else ()):
        if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1):
            raise RuntimeError("linear_bwd operands must be 2-D float32 CUDA tensors with unit column stride")
    call("fgs_linear_bwd_f32", M, N_out, K_in, ptr(dY), dY.stride(0), ptr(W), W.stride(0), ptr(X), X.stride(0), ptr(dX),
         dX.stride(0), ptr(mask), 0 if mask is None else mask.stride(0), ptr(colsum), ptr(dW), dW.stride(0), stream())


def mlp_fwd(M: int, X0: torch.Tensor, k0: int, T, t_cols: int, layers) -> None:
    """Whole forward chain of width-256 Linear(+ReLU) layers in one persistent launch (include/fgs_hip.h fgs_mlp_fwd_f32).
    `layers`: list of (W [256, ldw], K, bias [256] or None, relu, out [M, >=256]); X0 [M, ldx0]; T [M, ldt] view or None."""
    import ctypes
    n = len(layers)
    PtrArr, I64Arr, IntArr = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int * n
    W = PtrArr(*[ptr(l[0]) for l in layers])
    ldw = I64Arr(*[l[0].stride(0) for l in layers])
    K = IntArr(*[int(l[1]) for l in layers])
    bias = PtrArr(*[ptr(l[2]) for l in layers])
    relu = IntArr(*[int(bool(l[3])) for l in layers])
    outs = PtrArr(*[ptr(l[4]) for l in layers])
    ldo = I64Arr(*[l[4].stride(0) for l in layers])
    call("fgs_mlp_fwd_f32", M, n, ptr(X0), X0.stride(0), k0, ptr(T), 0 if T is None else T.stride(0), t_cols, W, ldw, K,
         bias, relu, outs, ldo, stream())


def mlp_chain(M: int, X0: torch.Tensor, k0: int, layers) -> None:
    """General one-launch chain (include/fgs_hip.h fgs_mlp_chain_f32), used for the backward data gradients.
    `layers`: list of dicts with W [n_rows, ldw], K, out [M, >= n_store] and optional bias, relu, mask [M, ldm], colsum [256],
    n_rows (default 256), n_store (default 256)."""
    import ctypes
    n = len(layers)
    PtrArr, I64Arr, IntArr = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int * n
    g = lambda l, k, d=None: l.get(k, d)
    call("fgs_mlp_chain_f32", M, n, ptr(X0), X0.stride(0), k0, None, 0, 0,
         PtrArr(*[ptr(l['W']) for l in layers]), I64Arr(*[l['W'].stride(0) for l in layers]),
         IntArr(*[int(l['K']) for l in layers]), IntArr(*[int(g(l, 'n_rows', 256)) for l in layers]),
         PtrArr(*[ptr(g(l, 'bias')) for l in layers]), IntArr(*[int(bool(g(l, 'relu', False))) for l in layers]),
         PtrArr(*[ptr(g(l, 'mask')) for l in layers]),
         I64Arr(*[0 if g(l, 'mask') is None else l['mask'].stride(0) for l in layers]),
         PtrArr(*[ptr(g(l, 'colsum')) for l in layers]), PtrArr(*[ptr(l['out']) for l in layers]),
         I64Arr(*[l['out'].stride(0) for l in layers]), IntArr(*[int(g(l, 'n_store', 256)) for l in layers]), stream())


def transpose_multi(mats) -> list:
    """[W.t().contiguous() for W in mats] in one launch (W: 2-D float32 CUDA, unit column stride, at most 8 of them)."""
    import ctypes
    n = len(mats)
    outs = [torch.empty(w.shape[1], w.shape[0], dtype=torch.float32, device=w.device) for w in mats]
    PtrArr, I64Arr, IntArr = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int * n
    call("fgs_transpose_multi", n, PtrArr(*[ptr(w) for w in mats]), IntArr(*[w.shape[0] for w in mats]),
         IntArr(*[w.shape[1] for w in mats]), I64Arr(*[w.stride(0) for w in mats]), PtrArr(*[ptr(o) for o in outs]),
         I64Arr(*[o.stride(0) for o in outs]), stream())
    return outs


def pad_cols_multi(mats, widths, outs=None) -> list:
    """[F.pad(W, (0, width - W.shape[1])) for W, width in zip(mats, widths)] in ONE launch (include/fgs_hip.h
    fgs_copy_cols_multi; W: 2-D float32 CUDA with unit column stride, at most 8 of them).  `outs`: per matrix None (allocate) or
    a caller's [rows, width] view to write into -- column slices of one tensor: a gather of column ranges in the same launch."""
    import ctypes
    n = len(mats)
    res = []
    for i, (w, width) in enumerate(zip(mats, widths)):
        o = outs[i] if outs is not None else None
        if o is None:
            o = torch.empty(w.shape[0], int(width), dtype=torch.float32, device=w.device)
        elif not (o.is_cuda and o.dtype == torch.float32 and o.dim() == 2 and o.stride(1) == 1 and o.shape[0] == w.shape[0]
                  and o.shape[1] == int(width) and o.stride(0) >= o.shape[1] and int(width) >= w.shape[1]):
            # (checked on the host: the kernel writes rows x width elements at this pitch whatever the tensor behind it holds)
            raise RuntimeError(f"pad_cols_multi: output {i} must be a [rows, width] float32 view with unit column stride")
        res.append(o)
    PtrArr, I64Arr, IntArr = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int * n
    call("fgs_copy_cols_multi", n, PtrArr(*[ptr(w) for w in mats]), IntArr(*[w.shape[0] for w in mats]),
         IntArr(*[w.shape[1] for w in mats]), I64Arr(*[w.stride(0) for w in mats]), PtrArr(*[ptr(o) for o in res]),
         I64Arr(*[o.stride(0) for o in res]), IntArr(*[o.shape[1] for o in res]), stream())
    return res


_RC_IMAGES = {}   # (device index, stream handle, backward) -> packed-weight scratch of the register-resident chains

# bench.py's roofline timing: while enabled, HIP events are recorded on the launch stream IMMEDIATELY around the C call that
# launches a matrix-core kernel (no Python between the events and the launch), with the launch's algorithmic FLOP count.
TIMING = {"enabled": False, "events": []}


# The same figure from INSIDE a captured step, whose replays cannot carry events: while STAMPS["buf"] is set, every matrix-core
# launch is handed a region of it (fgs_dyn_t.stamps) and its workgroups write their 100 MHz wall-clock start / end readings
# there, into the slot the device-side step counter selects -- bench.py reads the slots of the timed replays afterwards.
#   buf      int64 tensor [slots, STAMP_LAUNCHES, STAMP_WORDS] (zeroed by the owner before the steps it wants to read)
#   counter  device int64 the step's tick kernel advances (graph_step.CapturedFineStep.counter), or None (slot 0)
#   launches [(label, flop, kind)] in issue order of ONE step, rebuilt by stamps_begin_step() ... the launches that follow
STAMP_LAUNCHES, STAMP_WORDS = 8, 2048
STAMPS = {"buf": None, "counter": None, "launches": [], "seq": 0}


def stamps_begin_step() -> None:
    STAMPS["seq"] = 0
    STAMPS["launches"] = []


def _stamp_arg(label: str, flop: float, kind: str):
    buf = STAMPS["buf"]
    if buf is None:
        return None
    seq = STAMPS["seq"]
    if seq >= STAMP_LAUNCHES:
        return None
    STAMPS["seq"] = seq + 1
    STAMPS["launches"].append((label, float(flop), kind))
    cnt = STAMPS["counter"]
    return (buf.data_ptr() + seq * STAMP_WORDS * 8, None if cnt is None else cnt.data_ptr(), buf.shape[0],
            STAMP_LAUNCHES * STAMP_WORDS)


def stamps_read(buf=None, launches=None):
    """Per launch of a step: (label, flop, [duration in seconds of that launch in every slot that holds one]) from the wall-clock
    readings (100 MHz: s_memrealtime): chain / weight-gradient kernels min(start) .. max(end) over their workgroups, the tiled
    product its two atomically reduced words."""
    buf = STAMPS["buf"] if buf is None else buf
    launches = STAMPS["launches"] if launches is None else launches
    host = buf.cpu().numpy().astype("uint64")
    import numpy as np
    out = []
    for seq, (label, flop, kind) in enumerate(launches):
        durs = []
        for slot in range(host.shape[0]):
            w = host[slot, seq]
            if kind == "gemm":
                if w[1] == 0:
                    continue
                t0, t1 = int(~w[0] & np.uint64(0xFFFFFFFFFFFFFFFF)), int(w[1])
            else:
                rec = w.reshape(-1, 8)
                end = rec[:, 3] if kind == "rc" else rec[:, 5]
                live = end != 0
                if not live.any():
                    continue
                t0, t1 = int(rec[live, 1].min()), int(end[live].max())
            durs.append((t1 - t0) / 100e6)
        out.append((label, flop, durs))
    return out


def _timed(label: str, flop: float, launch) -> None:
    if not TIMING["enabled"]:
        launch()
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch()
    e1.record()
    TIMING["events"].append((e0, e1, None, label, 1, float(flop)))


def rc_mask_bits(M: int, device) -> torch.Tensor:
    """Buffer for the ReLU sign bits of one layer of `rc_chain` (16 bytes per lane of every 32-sample group)."""
    return torch.empty(((M + 31) // 32 + 4) * 64 * 4, dtype=torch.int32, device=device)


# Bumped by MaskedAdam whenever it updates parameters: a weight image packed under an older epoch is stale (rc2_pack / rc_chain
# `prepacked`).  (Writes to the weights that bypass the optimizer between a forward pass and its backward pass are not seen.)
WEIGHTS_EPOCH = [0]


def _rc_layer_array(layers, form: int):
    from ._lib import Rc2Layer, RcLayer
    n = len(layers)
    arr = ((Rc2Layer if form == 2 else RcLayer) * n)()
    for i, l in enumerate(layers):
        W = l['W']
        arr[i].W, arr[i].ldw, arr[i].n_out = ptr(W), W.stride(0), W.shape[0]
        arr[i].n_in = int(l.get('n_in', W.shape[1]))
        arr[i].bias, arr[i].relu = ptr(l.get('bias')), int(bool(l.get('relu', False)))
        arr[i].mask_bits = ptr(l.get('mask_bits'))
        out = l.get('out')
        arr[i].out, arr[i].ldo = ptr(out), (0 if out is None else out.stride(0))
        arr[i].n_store = int(l.get('n_store', 0 if out is None else out.shape[1]))
        ext = l.get('ext')
        arr[i].ext, arr[i].ld_ext = ptr(ext), (0 if ext is None else ext.stride(0))
        arr[i].ext_cols = int(l.get('ext_cols', 0 if ext is None else ext.shape[1]))
        if form == 2:
            arr[i].side = int(l.get('side', 0))            # (True / 1: side layer; 2: the output head)
    return arr


def _rc_image_ws(arr, n, backward: bool, form: int, device):
    import ctypes
    image_floats = lib().fgs_mlp_rc2_image_floats if form == 2 else lib().fgs_mlp_rc_image_floats
    need = int(image_floats(int(backward), n, ctypes.cast(arr, ctypes.c_void_p)))
    if need < 0:
        raise RuntimeError("rc_chain: bad layer list")
    key = (device.index, stream(), bool(backward), form)
    ws = _RC_IMAGES.get(key)
    if ws is None or ws.numel() < need:
        ws = _RC_IMAGES[key] = torch.empty(need, dtype=torch.float32, device=device)
    return ws


def _rc_weight_sig(layers):
    """What a packed image depends on: the weight views (address, pitch, shape) and roles of the layers, in order."""
    return tuple((l['W'].data_ptr(), l['W'].stride(0), tuple(l['W'].shape), int(l.get('n_in', l['W'].shape[1])),
                  int(l.get('side', 0)), int(l.get('ext_cols', 0 if l.get('ext') is None else l['ext'].shape[1])))
                 for l in layers)


def rc2_pack(layers_a, backward_a: bool, in0_cols_a: int, layers_b=None, backward_b: bool = True, in0_cols_b: int = 256,
             device=None):
    """The weight images of one or two form-2 chains in ONE launch (fgs_mlp_rc2_pack); returns a token for rc_chain(prepacked=...):
    the chains then skip their own pack launches while the token is current (same workspaces, same weight views, no optimizer
    update since)."""
    import ctypes
    arr_a = _rc_layer_array(layers_a, 2)
    ws_a = _rc_image_ws(arr_a, len(layers_a), backward_a, 2, device)
    arr_b = ws_b = None
    if layers_b:
        arr_b = _rc_layer_array(layers_b, 2)
        ws_b = _rc_image_ws(arr_b, len(layers_b), backward_b, 2, device)
    call("fgs_mlp_rc2_pack", int(backward_a), len(layers_a), ctypes.cast(arr_a, ctypes.c_void_p), int(in0_cols_a), ptr(ws_a),
         ws_a.numel(), int(backward_b), 0 if not layers_b else len(layers_b),
         None if arr_b is None else ctypes.cast(arr_b, ctypes.c_void_p), int(in0_cols_b), ptr(ws_b),
         0 if ws_b is None else ws_b.numel(), stream())
    sig, ws = {bool(backward_a): _rc_weight_sig(layers_a)}, {bool(backward_a): ws_a}
    if layers_b:
        sig[bool(backward_b)] = _rc_weight_sig(layers_b)
        ws[bool(backward_b)] = ws_b
    return dict(epoch=WEIGHTS_EPOCH[0], ws=ws, sig=sig)


def rc_chain(backward: bool, M: int, in0: torch.Tensor, in0_cols: int, layers, flop: float = 0.0, label: str = None,
             rows_dev=None, form: int = 1, prepacked=None) -> None:
    """One-launch MLP chain.  form 1: register-resident activations (include/fgs_hip.h fgs_mlp_rc_chain); form 2: the waves split
    the features, the slab's activations in LDS (fgs_mlp_rc2_chain: widths 256 / 192 / 128, no round quantisation, `side` layers).
    `layers`: list of dicts with W (the nn.Linear weight [n_out, >= n_in], any leading dimension) and optional n_in (default
    W.shape[1]), bias, relu, mask_bits (int32 buffer from rc_mask_bits), out ([M, >= n_store] row-major) / n_store, ext
    ([M, >= ext_cols] view) / ext_cols, side (form 2).  `prepacked`: rc2_pack's token."""
    import ctypes
    n = len(layers)
    arr = _rc_layer_array(layers, form)
    ws = _rc_image_ws(arr, n, backward, form, in0.device)
    # valid while no optimizer update happened since, the workspace is the one the pack filled and the weight views are the same
    pre = int(form == 2 and prepacked is not None and prepacked['epoch'] == WEIGHTS_EPOCH[0]
              and prepacked['ws'].get(bool(backward)) is ws and prepacked['sig'].get(bool(backward)) == _rc_weight_sig(layers))
    kern, pack = ("k_mlp_rc2", "k_rc2_pack") if form == 2 else ("k_mlp_rc", "k_rc_pack")
    label = label or (f"{kern} backward chain (+ {pack})" if backward else f"{kern} forward chain (+ {pack})")
    d = dyn(row_count=rows_dev, stamps=_stamp_arg(label, flop, "rc"))
    if form == 2:
        fn = lambda: call("fgs_mlp_rc2_chain", int(backward), M, n, ctypes.cast(arr, ctypes.c_void_p), ptr(in0), in0.stride(0),      # noqa: E731
                          in0_cols, ptr(ws), ws.numel(), pre, d, stream())
    else:
        fn = lambda: call("fgs_mlp_rc_chain", int(backward), M, n, ctypes.cast(arr, ctypes.c_void_p), ptr(in0), in0.stride(0),       # noqa: E731
                          in0_cols, ptr(ws), ws.numel(), d, stream())
    _timed(label, flop, fn)


def mlp_wgrad(M: int, items, flop: float = 0.0, rows_dev=None) -> None:
    """All weight / bias gradients of the MLPs in one launch (include/fgs_hip.h fgs_mlp_wgrad).  `items`: list of
    (dY [M, >= n_out], X [M, >= n_in], dW [n_out, >= n_in] zero-initialised, dbias [n_out] or None, n_out, n_in)."""
    import ctypes
    from ._lib import WgradItem
    n = len(items)
    arr = (WgradItem * n)()
    for i, (dY, X, dW, db, n_out, n_in) in enumerate(items):
        arr[i].dY, arr[i].ld_dy, arr[i].n_out = ptr(dY), dY.stride(0), int(n_out)
        arr[i].X, arr[i].ld_x, arr[i].n_in = ptr(X), X.stride(0), int(n_in)
        arr[i].dW, arr[i].ld_dw, arr[i].dbias = ptr(dW), dW.stride(0), ptr(db)
    d = dyn(row_count=rows_dev, stamps=_stamp_arg("k_mlp_wgrad", flop, "wgrad"))
    if _WGRAD_STORE:
        dev = items[0][0].device
        # one workspace per device: a device runs ONE weight-gradient launch at a time (the launch pairs of a training step are
        # ordered on their stream; a caller that overlaps two of them on two streams must set FGS_WGRAD_STORE=0)
        ws = _WGRAD_WS.get(dev.index)
        if ws is None:
            ws = _WGRAD_WS[dev.index] = torch.empty(int(lib().fgs_mlp_wgrad_ws_floats()), dtype=torch.float32, device=dev)
        _timed("k_mlp_wgrad", flop, lambda: call("fgs_mlp_wgrad_ws", M, n, ctypes.cast(arr, ctypes.c_void_p), ptr(ws), ws.numel(), d,
                                                 stream()))
        return
    _timed("k_mlp_wgrad", flop, lambda: call("fgs_mlp_wgrad", M, n, ctypes.cast(arr, ctypes.c_void_p), d, stream()))


# FGS_WGRAD_STORE=1 (default): the weight-gradient launch writes its partial blocks with plain stores into a workspace (73 MB per
# device) and a second launch adds them in order (fgs_mlp_wgrad_ws): no float atomics for the weights -- 64 MB of them per launch
# queued at the memory-side atomic units in front of the scatter kernels' --, bit-reproducible weight gradients.  0: atomics.
_WGRAD_STORE = os.environ.get("FGS_WGRAD_STORE", "1") == "1"
_WGRAD_WS = {}
