#!/usr/bin/env python
"""Static scan of the gfx950 ISA of every kernel in the product's two translation units for the patterns that cost round 3 its
three largest epilogue stalls (none of them visible in the source):

  * a vector-memory load followed within three instructions by `s_waitcnt vmcnt(0)` -- a serialized round trip (and a full drain
    of whatever else was in flight).  Found that way: the per-tile read of `*log_scale` in the contrastive-head forward (the
    compiler cannot use a scalar load behind the previous tile's stores), four branch-guarded token-mask loads at the head of every
    `filip5_kernel` tile;
  * atomics inside a persistent tile loop (the G kernel's per-tile `atomic_add(dtau)`: 16 k same-address atomics, each older than
    the next tile's first counted wait);
  * scratch (spill) traffic and the spill counts of the kernel descriptor (`hot_scratch`: the scratch instructions behind the
    kernel's first MFMA, i.e. in or after its tile loop -- a value parked during a persistent kernel's prologue costs nothing).

    python tools/isa_scan.py [--all]        # compiles x_clip_amd/csrc/xclip_api.hip and xclip_attn.hip to assembly under /tmp

tests/test_isa_guard.py holds the hot kernels to what this scan reports today (no spilled vector registers, no atomics in tile loops).

Streaming row kernels (LayerNorm family) legitimately wait for the row they just requested -- their latency is hidden by occupancy,
not inside the wave -- so a count is a pointer to read the code, not a verdict."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "x_clip_amd", "csrc")
OUT = "/tmp/xclip_isa"


def assemble(unit, extra=()):
    """the unit's device assembly with exactly the flags x_clip_amd/build.py compiles it with"""
    os.makedirs(OUT, exist_ok=True)
    dst = os.path.join(OUT, unit.replace(".hip", ".s"))
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mcode-object-version=5",
           "-ffp-contract=fast", "-Wno-unused-value", "-I", os.path.join(CSRC, "hw"), "-I", CSRC, *extra, "-S", "--cuda-device-only",
           os.path.join(CSRC, unit), "-o", dst]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return dst


def scan_product():
    """{demangled kernel name: counters} over both translation units"""
    sys.path.insert(0, ROOT)
    from x_clip_amd.build import UNITS
    out = {}
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(len(UNITS)) as ex:                  # the units compile side by side
        paths = list(ex.map(lambda ue: assemble(*ue), UNITS))
    for path in paths:
        st = scan(path)
        names = demangle(list(st))
        for n, v in st.items():
            out[re.sub(r"\(.*$", "", names[n]).replace("void ", "").replace("xc::", "").replace("unsigned short", "bf16")] = v
    return out


def scan(path):
    lines = open(path).read().split("\n")
    stats, name = {}, None
    for i, l in enumerate(lines):
        m = re.match(r"^(_ZN2xc\S+):\s", l)
        if m:
            name = m.group(1)
            stats[name] = dict(serial=0, loads=0, atomics=0, scratch=0, hot_scratch=0, vspill=0, sspill=0, _mfma=False)
            continue
        m = re.match(r"\s+\.name:\s+(_ZN2xc\S+)", l)
        if m and m.group(1) in stats:                      # kernel descriptor (metadata at the end of the file)
            for j in range(i, min(i + 24, len(lines))):
                v = re.match(r"\s+\.vgpr_spill_count:\s+(\d+)", lines[j])
                s = re.match(r"\s+\.sgpr_spill_count:\s+(\d+)", lines[j])
                if v:
                    stats[m.group(1)]["vspill"] = int(v.group(1))
                if s:
                    stats[m.group(1)]["sspill"] = int(s.group(1))
            continue
        if name is None:
            continue
        t = l.strip()
        if t.startswith(("global_load", "buffer_load", "flat_load")) and "lds" not in t:
            stats[name]["loads"] += 1
            if any("s_waitcnt vmcnt(0)" in lines[i + j] for j in range(1, 4) if i + j < len(lines)):
                stats[name]["serial"] += 1
        elif t.startswith(("global_atomic", "buffer_atomic", "flat_atomic")):
            stats[name]["atomics"] += 1
        elif t.startswith("scratch_"):
            stats[name]["scratch"] += 1
            if stats[name]["_mfma"]:                           # behind the kernel's first MFMA: inside (or after) its tile loop, not in its prologue
                stats[name]["hot_scratch"] += 1
        elif t.startswith("v_mfma"):
            stats[name]["_mfma"] = True
    return stats


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        return dict(zip(names, out.split("\n")))
    except Exception:
        return {n: n for n in names}


def main():
    show_all = "--all" in sys.argv
    print(f"{'serial':>6s} {'loads':>6s} {'atomics':>7s} {'scratch':>7s} {'vspill':>6s} {'sspill':>6s}  kernel")
    rows = sorted(scan_product().items(), key=lambda kv: -(kv[1]["serial"] + kv[1]["scratch"] + kv[1]["vspill"]))
    for n, s in rows:
        if not show_all and s["serial"] < 2 and s["scratch"] == 0 and s["vspill"] == 0 and s["atomics"] == 0:
            continue
        print(f"{s['serial']:6d} {s['loads']:6d} {s['atomics']:7d} {s['scratch']:7d} {s['vspill']:6d} {s['sspill']:6d}  {n[:90]}")


if __name__ == "__main__":
    main()
