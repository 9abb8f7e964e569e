"""uncertainty-nerf-gs_amd: MI355X-native uncertainty rendering hot path.

Layout
  csrc/       hand-written HIP kernels + the C ABI (include/unerf.h) -> csrc/libunerf.so
  lib.py      ctypes binding of the C ABI (fails loudly when the library is missing)
  ops.py      torch-tensor wrappers, one per C entry point
  render.py   frame-level pipelines (proposal sampling -> field -> composite -> moments)
  fields.py / models.py / ensemble.py
              host-side mirrors of the reference's Field / Model / EnsemblePipeline surface
  metrics.py  ause / auce / psnr / nll (parity metrics)
"""
__version__ = "0.1.0"
