"""misaki-render_amd — MI355X-native back end for misaki-render's sampling hot path.

The directory name carries the reference's hyphen, so import it with
``importlib.import_module("misaki-render_amd")`` (see tests/conftest.py).  Contents:

  csrc/        hand-written HIP (gfx950) wavefront path tracer + the C ABI (include/msk_gpu.h)
  host/        C++ host side mirroring the reference's plugin / Properties / XML interface
  abi.py       ctypes binding of the C ABI (fails loudly when the HIP library is missing)
  hostmirror.py, rgb2spec.py   scripting-side mirror of the reference's scene set-up
"""
__version__ = "0.1.0"
