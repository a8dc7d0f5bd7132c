"""hotrack_amd -- MI355X (gfx950) native PointNet++ operator stack behind HOTrack's HandTrackNet.

Layout (only what the hot path needs):
  csrc/              hand-written HIP kernels + the C-ABI (include/pn2_hip.h) -> libpn2_hip.so
  pointnet2_hip.py   drop-in for the reference's pybind module `pointnet2_cuda`
                     (network/models/pointnet_lib/src/pointnet2_api.cpp:11-24), bound over the C-ABI
  pointnet2_utils.py drop-in for network/models/pointnet_lib/pointnet2_utils.py (autograd Functions)
  fused.py           eval-time fused set-abstraction layers (gather -> MLP (MFMA) -> max)

There is NO CPU fallback anywhere in this package: CPU tensors raise, and importing
`pointnet2_hip` raises ImportError if libpn2_hip.so has not been built (see _build.build()).
"""
__version__ = "0.1.0"


# ---- several batches in flight need their own hardware queues ---------------------------------------------------------------
# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, shared with the
# framework's own streams); streams that share a queue run behind each other.  A serving loop that keeps N >= 2 captured
# forwards in flight on N streams (bench.py: 4) therefore needs GPU_MAX_HW_QUEUES >= N + 1 in the environment BEFORE the
# runtime initialises: measured on MI355X, four batches in flight give 0.82 ms / step with the default mapping and 0.73 ms with
# 8 queues.  The variable cannot be changed once the runtime is up, so all this module can do is say so.
_hw_queue_warned = False


def streams_in_flight(n: int) -> bool:
    """Call with the number of HIP streams a loop keeps busy at once (bench.py, serving code).  Returns True when the
    process's hardware-queue setting covers them; otherwise warns ONCE with the setting to use and returns False."""
    import os
    import warnings
    global _hw_queue_warned
    try:
        have = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    except ValueError:
        have = 4
    if n < 2 or have >= n + 1:
        return True
    if not _hw_queue_warned:
        _hw_queue_warned = True
        warnings.warn(
            f"hotrack_amd: {n} batches in flight on {n} HIP streams but GPU_MAX_HW_QUEUES={have}: streams that share a hardware "
            f"queue run behind each other (measured: 0.82 instead of 0.73 ms / step with four 64-cloud batches in flight). "
            f"Set GPU_MAX_HW_QUEUES={max(8, n + 1)} in the environment before the process touches the GPU.",
            RuntimeWarning, stacklevel=2)
    return False
