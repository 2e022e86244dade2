"""The progress watchdog of the wavefront loop (misaki-render_amd/csrc/msk_watchdog.h): SamplingIntegrator::render returns and
main.cpp:55-57 catches what it throws — a caller of msk_gpu_render gets the same from a loop that waits on a device: a sync group
that does not finish within MSK_WATCHDOG_S, or counters that stand still for MSK_WATCHDOG_GROUPS groups, end the render with
MSK_ERR_HIP "no progress" and a lost context.  The decision logic runs on the CPU; the GPU test provokes the wall limit with
a limit no group can meet."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_decision_logic(tmp_path):
    exe = str(tmp_path / "watchdog_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "native", "watchdog_check.cpp")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr


@pytest.mark.gpu
def test_a_render_that_makes_no_progress_ends_with_an_error_and_a_lost_context(abi, hostmirror, golden_lookup, monkeypatch):
    flat = hostmirror.cbox_scene(256, 256, coeff_lookup=golden_lookup)
    ctx = abi.Context(0)
    sc = abi.Scene(ctx, flat)
    film, st = sc.render(abi.render_params(spp=4))                  # the watchdog at its defaults: an ordinary render
    assert st.samples == 256 * 256 * 4 and np.isfinite(film).all()
    monkeypatch.setenv("MSK_WATCHDOG_S", "1e-9")                    # no sync group finishes within a nanosecond
    with pytest.raises(abi.MskError) as e:
        sc.render(abi.render_params(spp=64))
    assert e.value.code == abi.MSK_ERR_HIP and "no progress" in str(e.value) and "MSK_WATCHDOG_S" in str(e.value)
    monkeypatch.delenv("MSK_WATCHDOG_S")
    # the context is lost: every later call on it is refused at once, closing it does not wait for the device
    with pytest.raises(abi.MskError) as e:
        sc.render(abi.render_params(spp=1))
    assert "context is lost" in str(e.value)
    with pytest.raises(abi.MskError) as e:
        abi.Scene(ctx, flat)
    assert "context is lost" in str(e.value)
    sc.close()
    ctx.close()
    # (the kernels of the abandoned render finish on their own here — nothing was hung) and a new context renders again
    with abi.Context(0) as fresh:
        s2 = abi.Scene(fresh, flat)
        film2, st2 = s2.render(abi.render_params(spp=4))
        s2.close()
    assert np.array_equal(film2, film)
