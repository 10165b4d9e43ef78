"""TEST INFRASTRUCTURE ONLY: builds tests/emu/libxclip_emu.so = the product's C-ABI + kernel sources compiled
for the HOST against the wave64 emulator header (tests/emu/xc_device.h).  Used by the CPU test-suite."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "x_clip_amd", "csrc")
OUT = os.path.join(HERE, "libxclip_emu.so")
OUT_MEASURE = os.path.join(HERE, "libxclip_emu_measure.so")   # the same with -DXCLIP_MEASURE (experiments + switches, x_clip_amd/build.py)
UNITS = ["xclip_api.hip", "xclip_attn.hip"]                  # the product's translation units (x_clip_amd/build.py)


def sources():
    out = [os.path.join(CSRC, u) for u in UNITS] + [os.path.join(CSRC, "api_common.h"), os.path.join(HERE, "xc_device.h"),
                                                    os.path.join(ROOT, "include", "xclip.h")]
    kdir = os.path.join(CSRC, "kernels")
    out += [os.path.join(kdir, f) for f in sorted(os.listdir(kdir)) if f.endswith(".h")]
    mdir = os.path.join(kdir, "measure")
    out += [os.path.join(mdir, f) for f in sorted(os.listdir(mdir)) if f.endswith(".h")]
    adir = os.path.join(kdir, "asm")
    out += [os.path.join(adir, f) for f in sorted(os.listdir(adir)) if f.endswith(".inc")]
    return out


def build(force=False, measure=False):
    OUT = OUT_MEASURE if measure else globals()["OUT"]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(s) for s in sources()):
        return OUT
    cxx = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(cxx):
        cxx = "clang++"
    cmd = [cxx, "-x", "c++", "-std=c++17", "-O2", "-fPIC", "-shared", "-fno-strict-aliasing", "-Wno-unused-value", "-Wno-psabi",
           *(["-DXCLIP_MEASURE"] if measure else []), "-I", HERE, "-I", CSRC, *[os.path.join(CSRC, u) for u in UNITS]]
    # (into a private file, then renamed: the workers of a multi-rank test may find the library stale at the same moment -- two compilers
    #  writing one path left a rank loading a file that did not exist for an instant)
    tmp = f"{OUT}.{os.getpid()}.tmp"
    try:
        subprocess.run(cmd + ["-o", tmp], check=True)
        os.replace(tmp, OUT)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
