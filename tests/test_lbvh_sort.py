"""The device-side BVH builder's own radix sort and exclusive scan (csrc/msk_lbvh.hip: k_rs_hist / k_rs_scatter / k_scan_tile /
k_scan_add — hand-written since round 4, hipCUB before) against std::sort and a host prefix sum: tools/micro/lbvh_sort_test.hip,
compiled here with the library's flags and run on the GPU (1 … 5 M keys; random Morton codes, a handful of distinct codes, all
codes equal — the stable order among equal codes is the triangle index)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_device_radix_sort_and_scan(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    exe = str(tmp_path / "lbvh_sort_test")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "misaki-render_amd", "csrc"),
                    "-o", exe, os.path.join(ROOT, "tools", "micro", "lbvh_sort_test.hip")], check=True, capture_output=True)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "all ok" in p.stdout and "WRONG" not in p.stdout, p.stdout[-2000:] + p.stderr[-500:]
