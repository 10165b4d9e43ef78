"""Builds libxclip_hip.so (the gfx950 kernel library) in-tree with hipcc.

    python -m x_clip_amd.build [--force] [--measure]

hipcc cross-compiles for gfx950 without a GPU; the .so is git-ignored but travels with the working tree.

`--measure` builds libxclip_hip_measure.so instead: the same sources with -DXCLIP_MEASURE, i.e. WITH the environment switches of
the A/B and ablation runs (kernel generations, ablation masks whose results are garbage, the experiments under
csrc/kernels/measure/).  Only tools/ load it (x_clip_amd._lib.use_measurement_build()); the product library has none of them.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libxclip_hip.so")
LIB_MEASURE = os.path.join(HERE, "libxclip_hip_measure.so")


UNITS = [  # (source, extra flags)
    ("xclip_api.hip", []),
    # MFMA results stay in ordinary VGPRs in this unit: the attention kernels consume every accumulator with VALU right away and
    # the default heuristic parked them in AGPRs (32 v_accvgpr_read per 12 MFMAs in the backward's inner loop)
    ("xclip_attn.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form"]),
]


def _sources():
    out = [os.path.join(CSRC, u) for u, _ in UNITS] + [os.path.join(CSRC, "api_common.h"), os.path.join(CSRC, "hw", "xc_device.h"),
                                                        os.path.join(os.path.dirname(HERE), "include", "xclip.h")]
    kdir = os.path.join(CSRC, "kernels")
    out += [os.path.join(kdir, f) for f in sorted(os.listdir(kdir)) if f.endswith(".h")]
    mdir = os.path.join(kdir, "measure")
    out += [os.path.join(mdir, f) for f in sorted(os.listdir(mdir)) if f.endswith(".h")]
    adir = os.path.join(kdir, "asm")
    out += [os.path.join(adir, f) for f in sorted(os.listdir(adir)) if f.endswith(".inc")]       # (the generated body is the dependency, not its generator)
    return out


def _generate():
    """the hand-scheduled kernel bodies (csrc/kernels/asm/*_gen.py -> *_body.inc): regenerated when the generator's CONTENT differs from
    the one that wrote the body (its SHA-256 is kept beside the objects).  File times say nothing here: the generator leaves an unchanged
    body untouched, so after a fresh checkout -- or an edit of the generator that does not change its output -- the body stays older than
    its generator for ever and a time comparison would run the subprocess on every build()."""
    import hashlib
    adir = os.path.join(CSRC, "kernels", "asm")
    stamps = os.path.join(HERE, "_obj")
    for gen in sorted(f for f in os.listdir(adir) if f.endswith("_gen.py")):
        inc = os.path.join(adir, gen.replace("_gen.py", "_body.inc"))
        with open(os.path.join(adir, gen), "rb") as f:
            digest = hashlib.sha256(f.read()).hexdigest()
        stamp = os.path.join(stamps, gen + ".sha256")
        if os.path.exists(inc) and os.path.exists(stamp) and open(stamp).read().strip() == digest:
            continue
        subprocess.run([sys.executable, os.path.join(adir, gen)], check=True, stdout=subprocess.DEVNULL)
        try:
            os.makedirs(stamps, exist_ok=True)
            with open(stamp, "w") as f:
                f.write(digest + "\n")
        except OSError:                                         # a read-only tree: the generator simply runs again next time
            pass


def build(force: bool = False, verbose: bool = False, measure: bool = False) -> str:
    LIB = LIB_MEASURE if measure else globals()["LIB"]
    _generate()
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(s) for s in _sources()):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # the toolchain goes into the library (xclip_build_info): the wait states around the asm buffer stores were validated against ROCm 7.2's
    # code generation (DESIGN_APPENDIX.md 6d) -- another compiler means: run the GPU gate tests before trusting the GEMM epilogues
    try:
        ver = subprocess.run([hipcc, "--version"], capture_output=True, text=True, check=True).stdout.splitlines()[0].strip()
    except Exception:                                           # noqa: BLE001
        ver = "unknown"
    if "7.2." not in ver:
        print(f"x_clip_amd.build: WARNING: built with '{ver}', validated with HIP 7.2 -- run tests/test_kernels_gpu.py (the full-size GEMM gate tests) on an MI355X", file=sys.stderr)
    common = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mcode-object-version=5", "-ffp-contract=fast",
              "-Wno-unused-value", "-I", os.path.join(CSRC, "hw"), "-I", CSRC]
    if verbose:
        common.insert(0, "-Rpass-analysis=kernel-resource-usage")
    common.append('-DXCLIP_BUILD_TOOLCHAIN="hipcc ' + ver.replace('"', "'") + '"')
    if measure:
        common.append("-DXCLIP_MEASURE")
    objdir = os.path.join(HERE, "_obj_measure" if measure else "_obj")
    os.makedirs(objdir, exist_ok=True)
    procs, objs = [], []
    for unit, extra in UNITS:                                   # the units compile side by side
        obj = os.path.join(objdir, unit.replace(".hip", ".o"))
        objs.append(obj)
        procs.append(subprocess.Popen([hipcc, *common, *extra, "-c", os.path.join(CSRC, unit), "-o", obj]))
    if any(p.wait() != 0 for p in procs):
        raise RuntimeError("hipcc failed")
    tmp = f"{LIB}.{os.getpid()}.tmp"                             # (linked beside the target, then renamed: a concurrent loader never sees half a file)
    try:
        subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", tmp], check=True)
        os.replace(tmp, LIB)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv, measure="--measure" in sys.argv))
