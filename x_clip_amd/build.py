"""Builds libxclip_hip.so (the gfx950 kernel library) in-tree with hipcc.

    python -m x_clip_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU; the .so is git-ignored but travels with the working tree.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libxclip_hip.so")


def _sources():
    out = [os.path.join(CSRC, "xclip_api.hip"), os.path.join(CSRC, "hw", "xc_device.h"),
           os.path.join(os.path.dirname(HERE), "include", "xclip.h")]
    kdir = os.path.join(CSRC, "kernels")
    out += [os.path.join(kdir, f) for f in sorted(os.listdir(kdir))]
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(s) for s in _sources()):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mcode-object-version=5",
           "-ffp-contract=fast", "-Wno-unused-value",
           "-I", os.path.join(CSRC, "hw"), "-I", CSRC, os.path.join(CSRC, "xclip_api.hip"), "-o", LIB]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
