"""GPU: the shipped library-GEMM solution table (hotrack_amd/tunableop_gfx950.csv) is checked for its EFFECT at start-up,
not only for its form (tests/test_capi.py): solution indices are valid only for the hipBLASLt / rocBLAS build they were
recorded on, and a silent fall-back to the default heuristic was 10x slower on some of this network's shapes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = "import json; from hotrack_amd import gemm_tuning; print('STATUS ' + json.dumps(gemm_tuning.status()))"


def _status(env_extra):
    env = dict(os.environ, PYTHONPATH=ROOT, **env_extra)
    out = subprocess.run([sys.executable, "-c", PROBE], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("STATUS ")][-1]
    return json.loads(line[len("STATUS "):]), out.stderr


def test_shipped_table_applies_on_this_box():
    st, _ = _status({})
    assert st["gemm_table"] == "applied", st
    assert set(st["detail"]) == {"65536x384->256", "1024x384->512"}
    for v in st["detail"].values():
        assert v["table_ms"] <= 1.25 * v["default_ms"]


def test_table_from_another_library_build_is_reported_stale(tmp_path):
    from hotrack_amd import gemm_tuning
    txt = open(gemm_tuning.RESULTS).read()
    assert "Validator,HIPBLASLT_VERSION," in txt
    bad = tmp_path / "stale.csv"
    bad.write_text(txt.replace("Validator,HIPBLASLT_VERSION,", "Validator,HIPBLASLT_VERSION,0-another-build-"))
    st, err = _status({"PN2_TUNED_GEMMS_FILE": str(bad)})
    assert st["gemm_table"] == "stale", st
    assert "does not apply to this installation" in err  # logged once, loudly


def test_switch_off():
    st, _ = _status({"PN2_TUNED_GEMMS": "0"})
    assert st["gemm_table"] == "off"


TUNE_PROBE = r"""
import json, os, torch, torch.nn.functional as F
import torch.cuda.tunable as tunable
from hotrack_amd import gemm_tuning
x, w = torch.randn(7777, 136, device="cuda"), torch.randn(72, 136, device="cuda")   # a shape the shipped table does not hold
gemm_tuning.enable()
n0 = len(tunable.get_results())
with gemm_tuning.scope():            # plain scope: nothing is tuned
    y0 = F.linear(x, w)
n1 = len(tunable.get_results())
with gemm_tuning.scope(tune=True):   # the trainer's eager warm-up: tuned only on request
    y1 = F.linear(x, w)
n2 = len(tunable.get_results())
with gemm_tuning.scope():
    y2 = F.linear(x, w)
ref = (x.double() @ w.double().t()).float()
print("TUNE " + json.dumps({"n": [n0, n1, n2], "tuning_left_on": bool(tunable.tuning_is_enabled()), "enabled_left_on": bool(tunable.is_enabled()),
                            "err": [float((y - ref).abs().max()) for y in (y0, y1, y2)],
                            "cache": os.path.exists(os.environ.get("HOTRACK_GEMM_CACHE", "/nonexistent"))}))
"""


def _tune(env_extra):
    env = dict(os.environ, PYTHONPATH=ROOT, **env_extra)
    out = subprocess.run([sys.executable, "-c", TUNE_PROBE], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("TUNE ")][-1][5:])


def test_unknown_shapes_are_tuned_only_on_request_and_remembered(tmp_path):
    """scope(tune=True) (the trainer's capture warm-up): with HOTRACK_TUNE_GEMMS=1 a GEMM shape the shipped table does not hold gets
    a timed solution that later scopes use, and HOTRACK_GEMM_CACHE carries it to the next process; without the request nothing is
    tuned.  Either way the product is the same to round-off and TunableOp is left switched off outside the scopes."""
    r = _tune({})
    assert r["n"][0] == r["n"][1] == r["n"][2] and not r["tuning_left_on"] and not r["enabled_left_on"] and max(r["err"]) < 1e-3
    cache = tmp_path / "gemms.csv"
    r = _tune({"HOTRACK_TUNE_GEMMS": "1", "HOTRACK_GEMM_CACHE": str(cache)})
    assert r["n"][0] == r["n"][1] and r["n"][2] == r["n"][1] + 1, r
    assert not r["tuning_left_on"] and not r["enabled_left_on"] and max(r["err"]) < 1e-3 and r["cache"]
    assert "7777" in cache.read_text()
    r = _tune({"HOTRACK_GEMM_CACHE": str(cache)})  # next process: read back, nothing to tune
    assert r["n"][0] == r["n"][2] and r["n"][0] >= 1
