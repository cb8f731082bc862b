"""fgs-nerf_amd -- MI355X (gfx950) native voxel-NeRF render / training hot path.

Drop-in surfaces (same names, arguments and results as the reference's modules):

    ops.render_utils_cuda / ops.total_variation_cuda / ops.adam_upd_cuda   model/cuda/*.cpp pybind modules
    grid.create_grid / grid.DenseGrid / grid.MaskGrid                      model/grid.py
    adam.MaskedAdam                                                        model/adam.py
    dvgo_ray.* / nerf_ray.*                                                model/dvgo_ray.py, model/nerf_ray.py
    render.Alphas2Weights                                                  model/nerf.py:1173, model/dvgo.py:390
    nerf.nerf / dvgo.dvgo                                                  model/nerf.py, model/dvgo.py

The package directory is ``fgs_nerf_amd/`` (importable); ``fgs-nerf_amd`` at the repository root is a symbolic link to it,
kept for the paths documents and scripts were written with.
All device work goes through libfgs_hip.so (include/fgs_hip.h); there is no CPU fallback.
"""
__version__ = "0.1.0"
