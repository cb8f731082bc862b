"""Fused MI355X render path (wave-per-ray march, survivor features, fp32-MFMA MLP, compositing).

Placeholder until the fused kernels land: ``supports`` answers False so every model uses the
operator-at-a-time HIP path of render.py.
"""


def supports(model) -> bool:
    return False
