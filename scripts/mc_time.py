"""Times mesh extraction: field sampling of -sdf on a res^3 lattice + device marching cubes.
usage: python scripts/mc_time.py [res ...]"""
import sys
import time

import torch

sys.path.insert(0, '.')
from fgs_nerf_amd import synth  # noqa: E402
from fgs_nerf_amd.extract_geometry import extract_fields_device, marching_cubes_device  # noqa: E402
from fgs_nerf_amd.nerf import grid_sampler  # noqa: E402

dev = torch.device('cuda:0')
model = synth.build_model(160, synth.FINE_MODEL, device=dev)
neg = (-model.sdf.grid).detach().contiguous()
lo, hi = model.xyz_min.clone().float(), model.xyz_max.clone().float()
for res in [int(a) for a in sys.argv[1:]] or [256, 512]:
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        field = extract_fields_device(lo, hi, res, lambda p: grid_sampler(p, neg, model.xyz_min, model.xyz_max))
        torch.cuda.synchronize(); t1 = time.perf_counter()
        v, t = marching_cubes_device(field, 0.0)
        torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"res {res}: field {1e3 * (t1 - t0):.1f} ms, marching cubes {1e3 * (t2 - t1):.1f} ms, "
          f"{v.shape[0]} vertices, {t.shape[0]} triangles", flush=True)
