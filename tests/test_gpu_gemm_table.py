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
