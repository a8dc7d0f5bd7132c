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
import sys

RESULTS = os.environ.get("PN2_TUNED_GEMMS_FILE") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop_gfx950.csv")
_state = {"done": False, "on": False, "status": "off", "detail": None}

# Sentinel shapes of the start-up self-test (rows, in, out): the largest GEMM of a 64-cloud forward and the batch-1 layer-1
# GEMM of the keypoint branches, whose default-heuristic solution was 10x slower than the recorded one.
SENTINELS = ((65536, 384, 256), (1024, 384, 512))


def recorded_on() -> dict:
    """The library versions the shipped table was recorded on (its TunableOp validator rows)."""
    out = {}
    try:
        for line in open(RESULTS):
            f = line.strip().split(",")
            if len(f) >= 3 and f[0] == "Validator":
                out[f[1]] = ",".join(f[2:])
    except OSError:
        pass
    return out


def _time_linear(x, w, scoped: bool) -> float:
    import torch
    import torch.nn.functional as F
    import torch.cuda.tunable as tunable
    was = tunable.is_enabled()
    tunable.enable(bool(scoped))
    try:
        for _ in range(3):
            F.linear(x, w)
        best = float("inf")
        for _ in range(5):  # the minimum of five short runs: other work on the GPU (a test suite, another stream) only ever adds time
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                F.linear(x, w)
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10.0)
        return best
    finally:
        tunable.enable(was)


def _self_test(loaded: bool):
    """Does the table do what it was shipped for ON THIS INSTALLATION?  Solution indices are only valid for the hipBLASLt /
    rocBLAS build they were recorded on; TunableOp's validators reject a table from another build (`loaded` False), and a
    table that loads but loses on the sentinel shapes is switched off.  Logged once; bench.py reports `gemm_table`."""
    import sys
    import torch
    if not loaded:
        _state["status"], _state["detail"] = "stale", {"reason": "TunableOp rejected the table (validators: recorded on %s)" % recorded_on()}
    elif torch.cuda.is_current_stream_capturing():
        _state["status"], _state["detail"] = "applied", {"reason": "first use inside a stream capture: sentinel timing skipped"}
        return
    else:
        detail, lost = {}, False
        gen = torch.Generator(device="cuda")  # a private generator: the self-test must not advance the default CUDA RNG stream
        gen.manual_seed(0)                    # (whether / when the first enable() fires would change every later torch dropout mask)
        for rows, cin, cout in SENTINELS:
            x = torch.randn(rows, cin, device="cuda", generator=gen)
            w = torch.randn(cout, cin, device="cuda", generator=gen)
            # interleaved, minimum of three rounds each (and each round the minimum of five runs): the first measurement of a
            # process runs on clocks that are still ramping -- a 20 us GEMM once read 33 us that way and the table was declared stale
            t_tab = t_def = float("inf")
            for _ in range(3):
                t_def = min(t_def, _time_linear(x, w, False))
                t_tab = min(t_tab, _time_linear(x, w, True))
            detail["%dx%d->%d" % (rows, cin, cout)] = {"table_ms": round(t_tab, 4), "default_ms": round(t_def, 4)}
            lost |= t_tab > 1.5 * t_def  # (a stale table is 3-10x off on these shapes; timing noise is not)
            del x, w  # ~100 MB for the large sentinel: returned to the allocator before the next one is drawn
        _state["detail"] = detail
        _state["status"] = "stale" if lost else "applied"
        if lost:
            _state["on"] = False
    if _state["status"] == "stale":
        print("hotrack_amd.gemm_tuning: the shipped GEMM solution table does not apply to this installation (%s); the library's "
              "default heuristic is used -- re-record it with scripts/tune_gemms.py" % (_state["detail"],), file=sys.stderr, flush=True)


def status() -> dict:
    """{"gemm_table": "applied" | "stale" | "off", "detail": ...} after the first enable()."""
    enable()
    return {"gemm_table": _state["status"], "detail": _state["detail"]}


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
    _state["ours"] = True
    # TunableOp rewrites its table to this path at process exit; keep it away from the shipped file and per process
    tunable.set_filename(os.path.join(os.environ.get("TMPDIR", "/tmp"), "pn2_tunableop_%d.csv" % os.getpid()))
    _state["on"] = bool(tunable.read_file(RESULTS))
    cache = os.environ.get("HOTRACK_GEMM_CACHE")
    if cache and os.path.exists(cache):  # solutions a previous process tuned for its own shapes (scope(tune=True))
        _state["cache_loaded"] = bool(tunable.read_file(cache))
    tunable.enable(False)  # the table stays loaded; `scope()` switches it on around the inference path only
    _self_test(_state["on"])
    return _state["on"]


def _tune_requested() -> bool:
    return os.environ.get("HOTRACK_TUNE_GEMMS", "0") == "1"


class scope:
    """`with gemm_tuning.scope():` -- the recorded solutions apply to the GEMMs issued inside (the fused inference
    forward, eager or while it is being captured into a HIP graph) and nothing else in the process (a training loop
    that validates with the fast path keeps the library defaults and pays no per-GEMM table lookup).

    `scope(tune=True)` marks EAGER warm-up work whose GEMM shapes may be tuned on the spot when the user asks for it
    (HOTRACK_TUNE_GEMMS=1): TunableOp then times the library's candidates for every shape the loaded table does not hold and keeps
    the winner for the rest of the process.  The shipped table covers the benchmark configurations (per-GPU batch 32 / 64 x 1024
    points); another batch size or width issues other shapes, and the libraries' default heuristic is poor on some of them -- a
    (128 x 32768) (32768 x 384) weight-gradient product ran at 15 TFLOP/s untuned against ~90 tuned (profiles/r04_misc_measurements.md).
    HOTRACK_GEMM_CACHE=<file>: solutions found this way are also written there, and read back by the next process."""

    def __init__(self, tune: bool = False):
        self._tune = bool(tune)

    def __enter__(self):
        self._mine = self._tuning = False
        if enable() or (self._tune and _tune_requested() and _state["done"] and _tunable_usable()):
            import torch
            import torch.cuda.tunable as tunable
            if not tunable.is_enabled():
                tunable.enable(True)
                self._mine = True
            if self._tune and _tune_requested() and not torch.cuda.is_current_stream_capturing():
                tunable.set_max_tuning_duration(30)
                tunable.set_max_tuning_iterations(30)
                tunable.tuning_enable(True)
                self._tuning = True
        return self

    def __exit__(self, *exc):
        import torch.cuda.tunable as tunable
        if self._tuning:
            tunable.tuning_enable(False)
            cache = os.environ.get("HOTRACK_GEMM_CACHE")
            if cache:
                try:
                    _write_results(cache)
                except Exception as e:  # a read-only location must not take the training run down
                    print("hotrack_amd.gemm_tuning: could not write %s (%s)" % (cache, e), file=sys.stderr)
        if self._mine:
            tunable.enable(False)
        return False


def _write_results(path: str) -> None:
    """Every solution TunableOp holds in this process (the shipped table's and the ones tuned here) in its own file format:
    the validators of this installation, then (operator, problem, solution, time) rows."""
    import torch.cuda.tunable as tunable
    lines = ["Validator,%s,%s" % (k, v) for k, v in tunable.get_validators()]
    lines += [",".join(str(f) for f in row) for row in tunable.get_results()]
    tmp = "%s.%d.tmp" % (path, os.getpid())
    with open(tmp, "w") as f:
        f.write("\n".join(lines) + "\n")
    os.replace(tmp, path)


def _tunable_usable() -> bool:
    """TunableOp is ours to switch (nobody else enabled it before us) even when the shipped table did not load."""
    return _state.get("ours", False)
