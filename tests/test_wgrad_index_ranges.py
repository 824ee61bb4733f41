"""CPU: the LDS index arithmetic of the weight-gradient kernels (3d-playground_amd/csrc/conv_wgrad_geom.h, the header the
kernels themselves compile) is enumerated on the host for every tile instance the launchers use -- pixel-table reads,
direct-to-LDS fill blocks, fragment reads, the bf16 kernel's transposing reads -- and every index must stay inside the
structure it addresses (DESIGN.md "Rules": the grouped-wgrad experiment of round 1 faulted the GPU exactly there)."""
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_wgrad_lds_indices_stay_in_range(tmp_path):
    exe = str(tmp_path / "wgrad_index_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(REPO, "3d-playground_amd", "csrc"),
                           os.path.join(REPO, "tests", "wgrad_index_check.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-3000:]
    assert r.stdout.count("ok ") == 8, r.stdout
