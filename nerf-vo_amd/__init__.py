"""nerf_vo_amd -- MI355X (gfx950) native implementation of NeRF-VO's mapping hot path.

Package layout (SURVEY.md section 8 is the scope contract):
  csrc/        hand-written HIP kernels + the C-ABI (include/nerfvo_hip.h) -> lib/libnerfvo_hip.so
  _lib.py      ctypes binding of that C-ABI (fails loudly when the library is missing)
  tinycudann/  tcnn-compatible Python surface (Encoding / Network / NetworkWithInputEncoding)
"""
__version__ = "0.1.0"
