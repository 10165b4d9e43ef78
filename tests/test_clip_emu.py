"""CPU end-to-end run of the product `x_clip_amd.CLIP` (host mirror + the C-ABI kernels compiled against the wave64
emulator) on BASELINE config 0 (cfg1: dim 64, depth 2/2, image 64 patch 32, seq 32, batch 4), checked against the
reference-generated golden fixtures and against the fp64 oracle."""
import os
import sys

import pytest
import torch

from x_clip_amd import _lib

sys.path.insert(0, os.path.dirname(__file__))
import clip_cases as C  # noqa: E402
from emu.build_emu import build  # noqa: E402
from oracle import clip_oracle as O  # noqa: E402

DEV = torch.device("cpu")


@pytest.fixture(scope="module", autouse=True)
def emulator_library():
    _lib._use_library_for_tests(build())
    yield
    _lib._use_library_for_tests(None)


# the emulator runs every fixture in 15-20 s; the plain variants whose code paths are a subset of a combined fixture below
# (cfg1_dcl, cfg1_extra_dcl, cfg1_multiview, cfg1_multiview_m3n1, cfg1_filip, cfg1_simreg_extra, cfg1_rotary, cfg1_filip_downsample, cfg1_mlm,
# cfg1_simsiam, cfg1_simclr (4096-wide projector: minutes on the emulator), cfg1_causal) are exercised on the GPU only
# (cfg1_filip_dcl and cfg1_simsiam_mlm_dcl run in test_no_kernel_reads_unwritten_memory below, with poisoned allocations)
@pytest.mark.parametrize("name", ["cfg1_infonce", "cfg1_patchdrop",
                                  "cfg1_simreg_extra_dcl", "cfg1_rotary_dcl_multiview", "cfg1_filip_downsample_extra_dcl", "cfg1_mlm_dcl_multiview", "cfg1_causal_dcl_multiview"])
def test_clip_matches_reference_fixture(name):
    C.case_golden(DEV, name)


def test_clip_bf16_vs_oracle():
    C.case_vs_oracle(DEV, torch.bfloat16, O.CFG1, 4)


def test_state_dict_keys_and_shapes():
    from x_clip_amd import CLIP
    model = CLIP(**O.CFG1.ctor_kwargs())
    shapes = O.state_dict_shapes(O.CFG1)
    sd = model.state_dict()
    assert set(sd) == set(shapes)
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k


def test_no_cpu_fallback_without_library():
    """the product refuses CPU tensors when the real HIP library (not the emulator) is bound"""
    from x_clip_amd import ops
    _lib._use_library_for_tests(None)
    try:
        with pytest.raises(RuntimeError):
            ops.l2norm_fwd(torch.randn(4, 64))
    finally:
        _lib._use_library_for_tests(build())


def test_short_and_fully_padded_text():
    C.case_short_and_padded_text(DEV)


def test_freeze_and_early_returns():
    C.case_freeze_and_early_returns(DEV, O.CFG1, batch=4)


def test_pluggable_encoders_head_only():
    C.case_pluggable_encoders_head_only(DEV)


def test_no_kernel_reads_unwritten_memory():
    """reference fixtures again (FILIP head; SimSiam + MLM side losses) with every torch.empty the product makes poisoned with NaN"""
    with C.poisoned_empty():
        C.case_golden(DEV, "cfg1_filip_dcl")
        C.case_golden(DEV, "cfg1_simsiam_mlm_dcl")
