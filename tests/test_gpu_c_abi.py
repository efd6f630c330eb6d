"""libtspgnn.so driven from a plain C program (tests/c_abi/abi_smoke.c): the boundary is a C ABI, not a torch
extension -- no Python objects, no torch types, device pointers from hipMalloc."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_program_links_and_runs(cuda_device, tmp_path):
    gcc = shutil.which("gcc")
    if gcc is None or not os.path.isdir("/opt/rocm/include/hip"):
        pytest.skip("no C toolchain / ROCm headers on this box")
    libdir = os.path.join(ROOT, "tsp-gnn_amd", "tspgnn")
    exe = str(tmp_path / "abi_smoke")
    # link against the HIP runtime that torch bundles if there is one (a process must hold ONE HIP runtime), else ROCm's
    import torch
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    hiplib = tlib if any(f.startswith("libamdhip64") for f in os.listdir(tlib)) else "/opt/rocm/lib"
    cmd = [gcc, "-std=c99", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
           os.path.join(ROOT, "tests", "c_abi", "abi_smoke.c"), "-L", libdir, "-ltspgnn", "-L", hiplib, "-lamdhip64",
           "-Wl,-rpath," + libdir, "-Wl,-rpath," + hiplib, "-o", exe]
    build = subprocess.run(cmd, capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "C ABI OK" in run.stdout
