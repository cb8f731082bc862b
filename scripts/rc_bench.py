"""Micro-benchmark of the MLP forward chains at the bench's survivor counts: register-resident (fgs_mlp_rc_chain) vs the
LDS-resident persistent kernel (fgs_mlp_fwd_f32).  Interleaved rounds in one process, HIP events."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fgs_nerf_amd import fused_ops as fo

dev = torch.device('cuda:0')
for M in (49920, 58430, 65536, 32768):
    torch.manual_seed(0)
    X0 = torch.randn(M, 108, device=dev); X0[:, 106:] = 0
    Z = torch.randn(M, 308, device=dev); Z[:, 307] = 0
    Ks = [106, 256, 256, 256, 307, 256, 256]
    Ws = [torch.randn(256, k, device=dev) * 0.06 for k in Ks]
    bs = [torch.randn(256, device=dev) * 0.1 for _ in Ks]
    relu = [1, 1, 1, 0, 1, 1, 1]
    outs = [torch.empty(M, 256, device=dev) for _ in Ks]; outs[3] = Z
    bits = [fo.rc_mask_bits(M, dev) for _ in Ks]
    rc_layers = []
    for i in range(7):
        L = dict(W=Ws[i], bias=bs[i], relu=relu[i], mask_bits=bits[i] if relu[i] else None, out=outs[i], n_store=256)
        if i == 4: L.update(ext=Z[:, 256:], ext_cols=52)
        rc_layers.append(L)
    W0p = torch.nn.functional.pad(Ws[0], (0, 2)); V0p = torch.nn.functional.pad(Ws[4], (0, 1))
    old_layers = [(W0p if i == 0 else V0p if i == 4 else Ws[i], 108 if i == 0 else 308 if i == 4 else 256, bs[i], relu[i], outs[i]) for i in range(7)]
    flop = 2.0 * M * 256 * sum(Ks)
    def run_rc(): fo.rc_chain(False, M, X0, 108, rc_layers)
    def run_old(): fo.mlp_fwd(M, X0, 108, Z[:, 256:], 52, old_layers)
    res = {}
    for name, fn in (("rc", run_rc), ("lds", run_old)):
        for _ in range(3): fn()
    for rnd in range(5):
        for name, fn in (("rc", run_rc), ("lds", run_old)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) / 5 * 1e3)
    for name, v in res.items():
        v = sorted(v)
        print(f"M={M} {name:4s} median {v[len(v)//2]:8.1f} us  min {v[0]:8.1f} us  -> {flop / (v[len(v)//2] * 1e-6) / 1e12:6.1f} TFLOP/s", flush=True)

# in-kernel clock of the register-resident chain (shader-clock / wall-clock stamps per workgroup)
from fgs_nerf_amd._lib import call
# (run_rc still refers to the buffers of the LAST M of the loop above: do not touch M here)
stamps = torch.zeros(8 * 1024, dtype=torch.int64, device=dev)
call("fgs_mlp_rc_debug_stamps", stamps.data_ptr())
for _ in range(20):
    run_rc()
torch.cuda.synchronize()
stamps.zero_()
run_rc()
torch.cuda.synchronize()
call("fgs_mlp_rc_debug_stamps", None)
st = stamps.view(-1, 8)[:256].cpu().double()
cyc, wall = st[:, 2] - st[:, 0], (st[:, 3] - st[:, 1]) * 10e-9       # 100 MHz ticks -> seconds
ghz = (cyc / wall / 1e9)
print(f"in-kernel: shader cycles per workgroup median {cyc.median():.0f}, wall {wall.median() * 1e6:.1f} us, clock median {ghz.median():.3f} GHz "
      f"(min {ghz.min():.3f}, max {ghz.max():.3f}); ideal MFMA cycles per pass {54 * 128 * 64} x passes {(M + 32767) // 32768}", flush=True)
print("phases (median shader cycles per workgroup): init %.0f, chunks %.0f, epilogue %.0f, input load %.0f" %
      tuple(float(st[:, k].median()) for k in (4, 5, 6, 7)), flush=True)

# the backward data-gradient chain on the same buffers (layers 6 .. 1, masks from the forward's sign bits)
dY = torch.randn(M, 256, device=dev)
douts = [torch.empty(M, 256, device=dev) for _ in range(6)]
bwd_layers = [dict(W=Ws[6], mask_bits=bits[5], out=douts[0], n_store=256), dict(W=Ws[5], mask_bits=bits[4], out=douts[1], n_store=256),
              dict(W=Ws[4][:, :256], out=douts[2], n_store=256), dict(W=Ws[3], mask_bits=bits[2], out=douts[3], n_store=256),
              dict(W=Ws[2], mask_bits=bits[1], out=douts[4], n_store=256), dict(W=Ws[1], mask_bits=bits[0], out=douts[5], n_store=256)]
def run_bwd(): fo.rc_chain(True, M, dY, 256, bwd_layers)
for _ in range(20):
    run_bwd()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run_bwd()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 5 * 1e3)
print(f"M={M} backward chain median {sorted(ts)[2]:.1f} us")
stamps.zero_()
call("fgs_mlp_rc_debug_stamps", stamps.data_ptr())
run_bwd()
torch.cuda.synchronize()
call("fgs_mlp_rc_debug_stamps", None)
st = stamps.view(-1, 8)[:256].cpu().double()
print("backward phases (median shader cycles per workgroup of %d): total %.0f, init %.0f, chunks %.0f, epilogue %.0f, input load %.0f" %
      ((M + 32767) // 32768, float((st[:, 2] - st[:, 0]).median()), *(float(st[:, k].median()) for k in (4, 5, 6, 7))), flush=True)
