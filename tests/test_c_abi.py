"""A plain-C host of the shared library (tests/c_abi/check_abi.c): no Python, no
torch between the caller and the C ABI."""
import os
import subprocess

import pytest

from tests.conftest import ROOT


def _build(tmp_path):
    exe = str(tmp_path / "check_abi")
    libdir = os.path.join(ROOT, "disconet_amd")
    cmd = ["gcc", "-std=c11", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
           "-I", "/opt/rocm/include", os.path.join(ROOT, "tests", "c_abi", "check_abi.c"),
           "-L", libdir, "-ldisconet_hip", "-L", "/opt/rocm/lib", "-lamdhip64", "-lm",
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.check_call(cmd)
    return exe


def test_c_host_error_behaviour(tmp_path):
    out = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "C ABI error behaviour: ok" in out.stdout


@pytest.mark.gpu
def test_c_host_conv_and_voxelizer(tmp_path):
    out = subprocess.run([_build(tmp_path), "--gpu"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "bit-exact" in out.stdout and out.stdout.strip().endswith("OK")
