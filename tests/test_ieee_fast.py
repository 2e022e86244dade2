"""The one IEEE operation pair the device code expands by hand instead of leaving it to the compiler — `1.f / sqrtf(x)` of
srgb_model_eval (render/srgb.h:16), csrc/msk_device.h: rsqrt_ieee — against the compiler's correctly rounded division and square
root on EVERY binary32 bit pattern, against the host's arithmetic on a sample, and srgb_model_eval (whose guard sends NaN / infinite
/ huge arguments to the compiler's form) against a host restatement: tools/micro/ieee_fast_check.hip, compiled here with the
library's flags and run on the GPU.  The film / per-sample parity tests cover the same code on the arguments scenes produce; this
covers the arguments they do not."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_hand_expanded_rsqrt_is_the_compilers_on_every_float(tmp_path):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    exe = str(tmp_path / "ieee_fast_check")
    flags = [f for f in ge.HIPCC_FLAGS if f not in ("-fPIC", "-shared")]
    subprocess.run([hipcc] + flags + ["-I", os.path.join(ROOT, "misaki-render_amd", "csrc"), "-o", exe,
                                      os.path.join(ROOT, "tools", "micro", "ieee_fast_check.hip")], check=True, capture_output=True)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=200)
    assert p.returncode == 0 and "all ok" in p.stdout and "WRONG" not in p.stdout, p.stdout[-2000:] + p.stderr[-500:]
    assert "838860800 through the fast form, 0 differ" in p.stdout          # every float in [1, 2^100) took the hand-written form
