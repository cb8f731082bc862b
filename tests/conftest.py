import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): builds oracle/liboracle.so on first use."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def rel_l2(a, b):
    import torch
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def rel_l2_finite(a, b):
    """rel_l2 over the entries that are finite in the expectation; non-finite patterns must coincide."""
    import torch
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    fin = torch.isfinite(b)
    assert torch.equal(torch.isfinite(a), fin), "non-finite pattern differs"
    return float((a[fin] - b[fin]).norm() / b[fin].norm().clamp_min(1e-30))


def match_survivors(res, ref, thres=1e-4, label=""):
    """The kept-sample lists (ray_id, step_id) of the HIP path and of the oracle must be IDENTICAL.  The only accepted
    deviation is a decision flipped by an ulp-level difference in expf/sigmoid: every sample that is in one list and not
    in the other must sit within 1e-7 of one of the path's three decision thresholds (alpha > thres, weights > thres,
    T < 1e-3 with T = weights / alpha), judged on the values of the side that kept it.  Such samples are printed.
    Returns (ia, ib, n_flips): index tensors selecting the common samples, in order, from `res` and `ref`."""
    import torch
    ka = res["ray_id"].cpu().long() * (1 << 24) + res["step_id"].cpu().long()
    kb = torch.as_tensor(ref["ray_id"]).long() * (1 << 24) + torch.as_tensor(ref["step_id"]).long()
    if ka.shape == kb.shape and torch.equal(ka, kb):
        idx = torch.arange(ka.numel())
        return idx, idx, 0
    in_b, in_a = torch.isin(ka, kb), torch.isin(kb, ka)
    flips = []
    # coarse stage (model/nerf.py:981-988): the kept set is `weights > thres` of a FIRST Alphas2Weights pass over all
    # samples; the oracle records that pass ('pass1'), and a differing sample is judged on its first-pass values there
    p1 = ref.get("pass1") if hasattr(ref, "get") else None
    k1 = None if p1 is None else p1["ray_id"].long() * (1 << 24) + p1["step_id"].long()
    for side, d, keep in (("hip", res, ~in_b), ("oracle", ref, ~in_a)):
        for i in torch.nonzero(keep).flatten().tolist():
            w = float(torch.as_tensor(d["weights"])[i])
            a = float(torch.as_tensor(d["raw_alpha"])[i])
            near = min(abs(a - thres), abs(w - thres), abs(w / max(a, 1e-30) - 1e-3))
            if k1 is not None:
                key = int(torch.as_tensor(d["ray_id"])[i]) * (1 << 24) + int(torch.as_tensor(d["step_id"])[i])
                j = torch.nonzero(k1 == key).flatten()
                if j.numel():
                    w1, a1 = float(p1["weights"][j[0]]), float(p1["raw_alpha"][j[0]])
                    # (a sample the first pass never reached has w1 == 0 there: it was cut by the T < 1e-3 stop, whose
                    # deciding T is that of the last sample the pass did reach on this ray)
                    near = min(near, abs(w1 - thres), abs(w1 / max(a1, 1e-30) - 1e-3))
            flips.append((side, int(torch.as_tensor(d["ray_id"])[i]), int(torch.as_tensor(d["step_id"])[i]), a, w, near))
    print(f"[match_survivors{' ' + label if label else ''}] {len(flips)} sample(s) differ between the HIP path and the oracle:")
    for f in flips:
        print("    only in %-6s ray %d step %d alpha %.9g weight %.9g  distance to nearest threshold %.3g" % f)
    assert all(f[5] < 1e-7 for f in flips), "a kept-sample difference that no threshold explains"
    return torch.nonzero(in_b).flatten(), torch.nonzero(in_a).flatten(), len(flips)


def adam_drift_report(name, p_a, p_b, lr, steps, tight=1e-4, max_frac_tight=1.0, max_frac_tenth=1.0, max_frac_lr=1.0, max_worst_lr=None):
    """Two runs of the same Adam-trained tensor (HIP vs oracle, captured vs eager) after `steps` updates, element by element,
    instead of one loose norm.  What can be promised: with the reference's loss scaling (means over rays x 3)
    nearly every gradient element of this model is smaller than 1000 e', e' = eps / sqrt(1 - beta2) = 1e-7
    (tests/test_bench_step_parity_gpu.py counts them): Adam works in its eps regime, where an ABSOLUTE gradient difference dg moves an
    element by up to lr dg / e' per step -- float32 summation noise of a voxel's hundreds of cancelling atomic contributions
    (~1e-10) is worth 1e-3..1e-2 lr, and a hidden unit whose pre-activation rounds to the other side of zero more.  A per-element
    bar tied to a gradient floor is therefore vacuous here (no element is above a floor where Adam's normalisation has saturated);
    the honest statement is the DISTRIBUTION of the element-wise difference d = |a - b|, each level bounded by the caller:
        fraction with d > tight * (|b| + lr)      <= max_frac_tight      (differs at all, beyond rounding)
        fraction with d > 0.1 lr                  <= max_frac_tenth      (a tenth of one update)
        fraction with d > lr                      <= max_frac_lr         (a whole update apart)
        every element:  d <= 2 lr steps           (the update direction of a noise-level gradient is not determined;
                                                   `max_worst_lr`, in units of lr, replaces the bound where a tighter one holds)
    The single-step test (tests/test_bench_step_parity_gpu.py) separates the two causes exactly: same gradient in norm, same
    update rule per element."""
    import numpy as np
    import torch
    a = torch.as_tensor(p_a).detach().double().cpu().reshape(-1).numpy()
    b = torch.as_tensor(p_b).detach().double().cpu().reshape(-1).numpy()
    d = np.abs(a - b)
    n = max(d.size, 1)
    f_t, f_10, f_lr = (float((d > bar).sum()) / n for bar in (tight * (np.abs(b) + lr), 0.1 * lr, lr))
    worst = float(d.max()) if d.size else 0.0
    print(f"    {name:<22} elements {a.size:>9}  beyond {tight:g} (|p| + lr): {f_t:.2e}  beyond 0.1 lr: {f_10:.2e}  beyond lr: {f_lr:.2e}  "
          f"worst {worst / lr:.3f} lr (bound {2 * steps} lr)")
    assert worst <= (2 * steps if max_worst_lr is None else max_worst_lr) * lr, (name, worst)
    assert f_t <= max_frac_tight and f_10 <= max_frac_tenth and f_lr <= max_frac_lr, (name, f_t, f_10, f_lr)
    return f_t, f_10, f_lr, worst
