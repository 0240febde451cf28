"""MI355X-native diffusion LoRA train-step hot path (gfx950 HIP kernels behind a C ABI).

Layout:
  csrc/      hand-written HIP kernels + extern "C" entry points (include/aitk_mi355.h)
  build.py   hipcc driver (in-tree libaitk_mi355.so)
  _capi.py   ctypes binding of the C ABI (fails loudly when the library is missing)
  ops.py     tensor-level wrappers (pointer/stride plumbing only)
"""
__version__ = "0.1.0"
