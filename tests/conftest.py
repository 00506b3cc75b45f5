import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "llava-reward_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN_DIR


def record_locked_form(golden_name, dtype, model, err):
    """Artefact of the golden tests (round-4 review: 'prints the form instead of recording it'): one JSON line per (golden, mode) with
    the operand form .to('cuda') locked, its probe distances and the error against the reference -> gpurun_out/locked_forms.jsonl
    (gpurun merges gpurun_out/ back; tools/show_forms.py tabulates it; profiles/r5_locked_forms.jsonl is a committed copy)."""
    import json
    d = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        fi = dict(model.form_info or {})
        with open(os.path.join(d, "locked_forms.jsonl"), "a") as f:
            f.write(json.dumps({"golden": golden_name, "mode": dtype, "backbone": model.model_type, "form": model.operand_form,
                                "distance_to_strict": fi.get("distance_to_strict"), "probe_rows": fi.get("rows"), "budget": fi.get("budget"), "strict_noise_floor": fi.get("strict_noise_floor"),
                                "probe_seconds": fi.get("seconds"), "max_batch": model._opts["max_batch"], "max_seq": model._opts["max_seq"],
                                "max_crops": model._opts["max_crops"], "abs_err_vs_reference": err}) + "\n")
    except OSError:
        pass
