"""Library-GEMM algorithm selection for gfx950.

The inference path's dense layers are plain fp32 library GEMMs (hipBLASLt / rocBLAS through torch).  The libraries'
default heuristic is poor for several of this network's shapes on MI355X (e.g. the 1024x384x512 layer-1 GEMM of the
keypoint branches at batch 1: 79 us by default, 8 us with the best rocBLAS solution), so the solution per
(op, shape) was selected offline on an MI355X with PyTorch's TunableOp (scripts/tune_gemms.py) and is shipped as
`tunableop_gfx950.csv`.  `enable()` loads it with tuning OFF and `scope()` applies it around the inference forward: known shapes use the
recorded solution, unknown shapes the library default; a file recorded for another torch / ROCm / GPU is rejected by TunableOp's validators.
Set PN2_TUNED_GEMMS=0 to leave the library defaults in place.
"""
from __future__ import annotations

import os

RESULTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop_gfx950.csv")
_state = {"done": False, "on": False}


def enable() -> bool:
    """Idempotent.  Returns True when the tuned table is active."""
    if _state["done"]:
        return _state["on"]
    _state["done"] = True
    if os.environ.get("PN2_TUNED_GEMMS", "1") == "0" or not os.path.exists(RESULTS):
        return False
    import torch
    if not torch.cuda.is_available():
        return False
    import torch.cuda.tunable as tunable
    if tunable.is_enabled():  # the user runs their own TunableOp session: do not interfere
        return False
    tunable.enable(True)
    tunable.tuning_enable(False)
    # TunableOp rewrites its table to this path at process exit; keep it away from the shipped file and per process
    tunable.set_filename(os.path.join(os.environ.get("TMPDIR", "/tmp"), "pn2_tunableop_%d.csv" % os.getpid()))
    _state["on"] = bool(tunable.read_file(RESULTS))
    tunable.enable(False)  # the table stays loaded; `scope()` switches it on around the inference path only
    return _state["on"]


class scope:
    """`with gemm_tuning.scope():` -- the recorded solutions apply to the GEMMs issued inside (the fused inference
    forward, eager or while it is being captured into a HIP graph) and nothing else in the process (a training loop
    that validates with the fast path keeps the library defaults and pays no per-GEMM table lookup)."""

    def __enter__(self):
        self._mine = False
        if enable():
            import torch.cuda.tunable as tunable
            if not tunable.is_enabled():
                tunable.enable(True)
                self._mine = True
        return self

    def __exit__(self, *exc):
        if self._mine:
            import torch.cuda.tunable as tunable
            tunable.enable(False)
        return False
