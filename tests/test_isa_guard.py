"""The hot kernels' gfx950 assembly is held to what tools/isa_scan.py reports today.  Round 3's three largest epilogue stalls were
invisible in the source (an atomic per tile, scalar parameters re-read as vector loads behind a full drain, branch-guarded loads that
serialize), and innocuous edits moved `sim5_grad_fast_kernel` from 0 to 102 / 199 / 255 spilled registers three times in one
session -- this test compiles both translation units to assembly with the build's own flags (hipcc cross-compiles without a GPU) and
checks the kernels that carry a step."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC) or shutil.which("c++filt") is None, reason="needs hipcc and c++filt")

# kernel -> (max serialized load + drain pairs, max atomic instructions); every one of them: no spilled vector register, no scratch
HOT = {
    "gemm5_kernel<false, true, 0, 0>": (0, 0), "gemm5_kernel<false, false, 0, 0>": (0, 0),      # forward NT / dgrad NN, interior path only
    "sim5_lse_kernel": (0, 0),                                                                  # no load at all in a tile's epilogue
    "filip5_kernel": (0, 0),
    # (round 4: the two key-validity bytes of a thread are requested together and waited for once -- the scan counts both loads of that one
    #  round trip, plus the loop for shapes the kernels never get; the backward's 5 branch-guarded loads of round 3 are gone)
    "attn3_fwd_kernel<false>": (3, 0), "attn3_bwd_kernel<false>": (6, 0), "attn3_bwd_kernel<true>": (6, 0),
    # (last template argument: the non-temporal hint on the streamed rows, xclip_api.hip ROWS_NT)
    "ln_geglu_bwd_kernel<bf16, 2, 2, true>": (0, 0), "ln_fwd_kernel<bf16, 4, true, true>": (4, 0), "ln_fwd_kernel<bf16, 1, false, true>": (2, 0),
    "ln_bwd_kernel<bf16, 1, false, true, false>": (3, 0), "ln_chain_fwd_kernel<bf16, 1, false>": (2, 0), "ln_chain_bwd_kernel<bf16, 1, true>": (1, 0),
    # (round 6: the same kernel writing the lower block's feed-forward row constants -- one more pair: its W2 gamma vector, once per work-group)
    "ln_bwd_kernel<bf16, 1, false, true, true>": (4, 0),
    "splitk_reduce_kernel<bf16>": (0, 0),
}
# G of the contrastive head: round 4's exact two-exponential form for wave blocks whose lse values spread beyond one reference point
#  sits beside the fast form -- the compiler parks 9 loop-invariant values in scratch during the prologue; what is held is that the
# K loop has no scratch traffic and a tile's epilogue at most one reload (kernel: spilled registers, scratch instructions behind the first MFMA)
PROLOGUE_SPILLS_ONLY = {"sim5_grad_fast_kernel<true, 0>": (16, 2), "sim5_grad_fast_kernel<false, 0>": (16, 2)}
# wide-head attention (attention4.h, round 4): the backward lives in 340 of its wave's 512 registers without spilling; the forward (eight waves,
# 256 registers) parks ~20 values of its two step variants
WIDE_HEADS = {"attn4_bwd_kernel<false>": 0, "attn4_bwd_kernel<true>": 0, "attn4_fwd_kernel<false>": 24, "attn4_fwd_kernel<true>": 24}
# kernels whose ragged-tile path legitimately holds serialized loads (row gathers, residual rows): spills only
NO_SPILL = ["gemm8_kernel<false, 0>", "gemm8_kernel<false, 1>", "gemm8_kernel<true, 0>", "gemm8_kernel<true, 1>",   # (asm units: the compiler's part around them)
            "gemm4_kernel<true, true, 1>", "gemm5_kernel<false, false, 3, 0>", "gemm5_kernel<false, true, 3, 0>", "filip_route_kernel<bf16>",
            "attn_pool_fwd_kernel<bf16, 64>", "attn_pool_bwd_kernel<bf16, 64>", "scatter_add_sorted_kernel<bf16, 1>",
            # round 5: the fused feed-forward backward (its epilogue has no register to spare: compiler-visible stores with recomputed addresses),
            # the single-pass attention backward, the latency-built small-output GEMM
            "gemm9_geglu_bwd_kernel<0>", "attn5_bwd_kernel", "gemm_small_kernel<false, false, false>", "gemm_small_kernel<false, true, false>",
            "gemm_small_kernel<true, true, false>", "gemm_small_kernel<false, false, true>",
            # round 6: the radix sort in front of the embedding gradient
            "sort_hist_kernel<true>", "sort_hist_kernel<false>", "sort_scan_kernel", "sort_scatter_kernel<true, false>", "sort_scatter_kernel<false, true>",
            "sort_scatter_kernel<true, true>", "sort_scatter_kernel<false, false>"]


@pytest.fixture(scope="module")
def isa():
    import isa_scan
    return isa_scan.scan_product()


def test_hot_kernels_do_not_spill_or_stall(isa):
    bad = []
    for k, (serial, atomics) in HOT.items():
        assert k in isa, f"{k}: not in the library any more -- update tests/test_isa_guard.py"
        s = isa[k]
        if s["vspill"] or s["scratch"]:
            bad.append(f"{k}: {s['vspill']} spilled vector registers, {s['scratch']} scratch instructions")
        if s["serial"] > serial:
            bad.append(f"{k}: {s['serial']} load + full-drain pairs (was {serial})")
        if s["atomics"] > atomics:
            bad.append(f"{k}: {s['atomics']} atomic instructions (was {atomics})")
    for k, (vs, hot) in PROLOGUE_SPILLS_ONLY.items():
        assert k in isa, f"{k}: not in the library any more -- update tests/test_isa_guard.py"
        s = isa[k]
        if s["vspill"] > vs or s["hot_scratch"] > hot or s["serial"] > 1 or s["atomics"] > 1:
            bad.append(f"{k}: {s['vspill']} spilled vector registers (<= {vs}), {s['hot_scratch']} scratch instructions behind the first MFMA (<= {hot}), "
                       f"{s['serial']} load + drain pairs, {s['atomics']} atomics")
    for k, vs in WIDE_HEADS.items():
        assert k in isa, f"{k}: not in the library any more -- update tests/test_isa_guard.py"
        if isa[k]["vspill"] > vs or isa[k]["atomics"]:
            bad.append(f"{k}: {isa[k]['vspill']} spilled vector registers (<= {vs}), {isa[k]['atomics']} atomics")
    for k in NO_SPILL:
        assert k in isa, f"{k}: not in the library any more -- update tests/test_isa_guard.py"
        if isa[k]["vspill"] or isa[k]["scratch"]:
            bad.append(f"{k}: {isa[k]['vspill']} spilled vector registers, {isa[k]['scratch']} scratch instructions")
    assert not bad, "\n".join(bad)


def test_generated_asm_bodies_are_in_sync():
    """csrc/kernels/asm/*_body.inc is what its generator writes today (the .inc is committed so that the asm can be read in the repository)"""
    import subprocess
    adir = os.path.join(ROOT, "x_clip_amd", "csrc", "kernels", "asm")
    for gen in sorted(f for f in os.listdir(adir) if f.endswith("_gen.py")):
        inc = os.path.join(adir, gen.replace("_gen.py", "_body.inc"))
        before = open(inc).read()
        subprocess.run([sys.executable, os.path.join(adir, gen)], check=True, stdout=subprocess.DEVNULL)
        after = open(inc).read()
        assert before == after, f"{inc} was stale: regenerated -- commit it"


def test_asm_units_name_only_their_own_registers(isa):
    """a whole-kernel asm unit (gemm8.h) lives in the registers it names; the compiler's part must not spill around it and the kernel must fit two
    waves per SIMD (256 registers)"""
    for k in [n for n in isa if n.startswith("gemm8_kernel")]:
        assert isa[k]["vspill"] == 0 and isa[k]["scratch"] == 0, (k, isa[k])
