"""Locate (and, where hipcc is present, build) the TOOLS-side build of the library: the product's sources compiled with
-DEIG_TOOLS, which adds the experiment hooks declared in tools/eigsolve_tools.h (operand-mask gemm probes, the two-stage launch
skeleton).  Import this BEFORE eigensolver_gpu_amd.api: it points EIGSOLVE_GPU_LIB at tools/_lib/libeigsolve_gpu.so unless the
caller already chose a variant build."""
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS_LIB = os.path.join(ROOT, "tools", "_lib", "libeigsolve_gpu.so")


def ensure():
    if os.environ.get("EIGSOLVE_GPU_LIB"):
        return os.environ["EIGSOLVE_GPU_LIB"]
    # (built here, shipped to the GPU box with the snapshot; rebuilt only where it is missing -- run `make tools` after source edits)
    if not os.path.exists(TOOLS_LIB) and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "eigensolver_gpu_amd", "csrc"), "tools", "-j", "8"])
    if not os.path.exists(TOOLS_LIB):
        raise RuntimeError("tools build missing: make -C eigensolver_gpu_amd/csrc tools")
    os.environ["EIGSOLVE_GPU_LIB"] = TOOLS_LIB
    return TOOLS_LIB


ensure()
