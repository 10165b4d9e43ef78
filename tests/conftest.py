import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def pytest_sessionfinish(session, exitstatus):
    """the parity tables: largest error each per-kernel check measured in this session next to the bound it was held to
    (tests/kernel_cases.py: REPORT) and the end-to-end loss / gradient errors against the fp64 oracle (tests/clip_cases.py: REPORT)
    -> gpurun_out/parity_report_{gpu,emu}.txt; the GPU run's copy is committed under profiles/"""
    lines = []
    where = "gpu" if _have_gpu() else "emu"
    K = sys.modules.get("kernel_cases")
    C = sys.modules.get("clip_cases")
    if K is not None and K.REPORT:
        lines.append(f"# per-kernel checks, maximum over the {where} suite: measured error | bound | unit")
        for k in sorted(K.REPORT):
            m, b, u = K.REPORT[k]
            lines.append(f"{k:58s} {m:10.3g} | {b:8.3g} | {u}")
    if C is not None and C.REPORT:
        lines.append(f"# end to end vs the fp64 oracle ({where}): |loss - oracle| / max(1, |oracle|); worst parameter-gradient relative error; lowest cosine")
        for k in sorted(C.REPORT):
            r = C.REPORT[k]
            lines.append(f"{k:78s} loss {r['loss_err']:9.3g} | rel {r['worst_rel'][0]:9.3g} ({r['worst_rel'][1]}) | cos {r['worst_cos'][0]:.5f} ({r['worst_cos'][1]})")
    if not lines:
        return
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, f"parity_report_{where}.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
