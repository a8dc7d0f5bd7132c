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
