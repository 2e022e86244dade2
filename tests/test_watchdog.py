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


@pytest.mark.gpu
@pytest.mark.parametrize("wait,threads", [("callback", "1"), ("callback", "4"), ("sleep", "1"), ("poll", "4"), ("event", "1"), ("sync", "4")])
def test_every_wait_strategy_renders_the_same_film_and_keeps_the_wall_limit(abi, hostmirror, golden_lookup, monkeypatch, wait, threads):
    """MSK_WAIT x MSK_HOST_THREADS (msk_watchdog.h: wait_any; msk_gpu.hip: drive): who waits how for a sync group changes no bit
    of the film — 1024+ regions, so the four-part loop is what runs — and the strategies that can time a wait (callback, sleep,
    poll) end a render that makes no progress the same way."""
    flat = hostmirror.cbox_scene(192, 160, coeff_lookup=golden_lookup)
    prm = abi.render_params(spp=40, seed=7)
    with abi.Context(0) as ctx:
        sc = abi.Scene(ctx, flat)
        ref, st0 = sc.render(prm)
        sc.close()
    monkeypatch.setenv("MSK_WAIT", wait)
    monkeypatch.setenv("MSK_HOST_THREADS", threads)
    ctx = abi.Context(0)
    assert f"wait = {wait}" in ctx.describe() and f"{threads} loop thread" in ctx.describe()
    sc = abi.Scene(ctx, flat)
    film, st = sc.render(prm)
    assert st.samples == st0.samples == 192 * 160 * 40 and st.segments == st0.segments
    assert np.array_equal(film.view(np.uint32), ref.view(np.uint32))
    if wait in ("callback", "sleep", "poll"):
        monkeypatch.setenv("MSK_WATCHDOG_S", "1e-9")
        with pytest.raises(abi.MskError) as e:
            sc.render(abi.render_params(spp=64))
        assert "no progress" in str(e.value) and "MSK_WATCHDOG_S" in str(e.value)
        monkeypatch.delenv("MSK_WATCHDOG_S")
        with pytest.raises(abi.MskError) as e:
            sc.render(prm)
        assert "context is lost" in str(e.value)
    sc.close()
    ctx.close()
