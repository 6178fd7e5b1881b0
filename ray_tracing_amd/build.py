"""Builds librt_hip.so (HIP kernels + C++ host + C ABI) and rt_cli for gfx950, in-tree.

hipcc cross-compiles without a GPU, so this runs anywhere the ROCm toolchain is installed.
"""
import json
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librt_hip.so")
CLI = os.path.join(HERE, "rt_cli")
RESOURCES = os.path.join(HERE, "kernel_resources.json")   # per kernel: registers, scratch bytes, spills, LDS (written by every library build)
SOURCES = ["render.hip", "lbvh.hip", "lbvh_host.cpp", "refit.hip", "refit_host.cpp", "capi.cpp", "scene_builder.cpp", "flatten.cpp", "wide_bvh.cpp", "config.cpp"]
HEADERS = ["rt_types.h", "rt_device.h", "frame.h", "lbvh.h", "refit.h", "scene_builder.h", "capi_internal.h", "../../include/rt_hip.h"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-function", "-ffp-contract=off"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the ROCm toolchain is required to build librt_hip.so")
    return exe


def _split_remarks(stderr):
    """The compiler's kernel-resource-usage remarks as {kernel: {field: value}}, and whatever else it printed."""
    kernels, other, cur = {}, [], None
    for line in stderr.splitlines():
        m = re.search(r"remark: \s*(.*?)\s*\[-Rpass-analysis=kernel-resource-usage\]", line)
        if not m:
            if not line.startswith("In file included from"):   # the include trail of a remark about a library kernel
                other.append(line)
            continue
        key, _, val = m.group(1).partition(":")
        if key.strip() == "Function Name":
            name = val.strip()
            try:
                name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0] or name
            except OSError:
                pass
            cur = kernels.setdefault(name.replace("void ", ""), {})
        elif cur is not None and val.strip():
            v = val.strip()
            cur[key.strip()] = int(v) if v.lstrip("-").isdigit() else v
    return kernels, "\n".join(other)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = _hipcc()
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    if force or _stale(LIB, srcs + hdrs + [os.path.abspath(__file__)]):
        cmd = [hipcc] + FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-fno-caret-diagnostics", "-shared", "-o", LIB] + srcs
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, cwd=CSRC, stderr=subprocess.PIPE, text=True)
        remarks, other = _split_remarks(r.stderr)
        if other.strip():
            print(other, file=sys.stderr)
        if r.returncode != 0:
            raise subprocess.CalledProcessError(r.returncode, cmd)
        with open(RESOURCES, "w") as f:
            json.dump(remarks, f, indent=1, sort_keys=True)
    cli_src = os.path.join(CSRC, "rt_cli.cpp")
    if force or _stale(CLI, [cli_src, LIB]):
        cmd = [hipcc, "-O2", "-std=c++17", "-o", CLI, cli_src, "-L" + HERE, "-lrt_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB


def build_host_asan():
    """AddressSanitizer + UndefinedBehaviorSanitizer build of the host sources (no HIP file) and the oracle, linked into the test
    driver tests/asan/host_asan_driver (see oracle/Makefile, target `asan`)."""
    subprocess.run(["make", "asan"], check=True, cwd=os.path.join(HERE, "..", "oracle"))


if __name__ == "__main__":
    if "--host-asan" in sys.argv:
        build_host_asan()
    else:
        build(force="--force" in sys.argv, verbose=True)
