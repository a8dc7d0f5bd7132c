"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol the headers declare
(no compute calls -- there is no GPU here), and the Python boundary validates its arguments."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pn2[xs]?_[a-z0-9_]+)\s*\(", src)))


def test_headers_declare_the_reference_surface():
    names = _declared("pn2_hip.h")
    for n in ("pn2_ball_query", "pn2_group_points", "pn2_group_points_grad", "pn2_gather_points", "pn2_gather_points_grad",
              "pn2_furthest_point_sampling", "pn2_knn", "pn2_three_nn", "pn2_three_interpolate", "pn2_three_interpolate_grad"):
        assert n in names  # one per m.def of pointnet2_api.cpp:11-24


def test_library_exports_every_declared_symbol(hip_lib_path):
    lib = ctypes.CDLL(hip_lib_path)
    for header in ("pn2_hip.h", "pn2_ext.h", "pn2_sdf.h"):
        for name in _declared(header):
            assert hasattr(lib, name), f"{name} declared in {header} but not exported"
    lib.pn2_abi_version.restype = ctypes.c_int
    assert lib.pn2_abi_version() >= 1
    lib.pn2_strerror.restype = ctypes.c_char_p
    assert b"NULL" in lib.pn2_strerror(-2)


def test_argument_validation_without_gpu(hip_lib_path):
    """Invalid arguments are rejected before anything touches the device."""
    lib = ctypes.CDLL(hip_lib_path)
    vp = ctypes.c_void_p
    lib.pn2_knn.argtypes = [ctypes.c_int] * 4 + [vp] * 5
    assert lib.pn2_knn(1, 4, 16, 0, None, None, None, None, None) == -1      # k < 1
    assert lib.pn2_knn(1, 4, 16, 201, None, None, None, None, None) == -3    # k > 200 (reference array bound)
    assert lib.pn2_knn(1, 4, 16, 8, None, None, None, None, None) == -2      # NULL pointers
    lib.pn2_furthest_point_sampling.argtypes = [ctypes.c_int] * 3 + [vp] * 4
    assert lib.pn2_furthest_point_sampling(1, 0, 4, None, None, None, None) == -1   # n < 1
    assert lib.pn2_furthest_point_sampling(0, 16, 4, None, None, None, None) == 0   # empty batch is a no-op
    lib.pn2_ball_query.argtypes = [ctypes.c_int] * 3 + [ctypes.c_float, ctypes.c_int] + [vp] * 4
    assert lib.pn2_ball_query(1, 16, 4, 0.1, 0, None, None, None, None) == -1       # nsample < 1
    lib.pn2_group_points.argtypes = [ctypes.c_int] * 5 + [vp] * 4
    assert lib.pn2_group_points(2, 0, 16, 4, 4, None, None, None, None) == 0        # C == 0 is legal (sa1)
    cf, ci = ctypes.c_float, ctypes.c_int
    lib.pn2s_nearest.argtypes = [ci, ci, vp, vp, vp, vp, ci, ci, cf, vp, vp, vp, vp]
    assert lib.pn2s_nearest(4, 8, None, None, None, None, 1, 150, 0.003, None, None, None, None) == -1   # even res
    assert lib.pn2s_nearest(4, 8, None, None, None, None, 1, 151, 0.003, None, None, None, None) == -2   # NULL pointers
    lib.pn2s_trilinear.argtypes = [ci, vp, vp, ci, ci, cf, cf, cf, cf, vp, vp]
    assert lib.pn2s_trilinear(8, None, None, 1, 201, -0.2, 0.0, -0.05, 0.05, None, None) == -1           # stride <= 0
    assert lib.pn2s_trilinear(0, None, None, 1, 201, -0.2, 0.002, -0.05, 0.05, None, None) == 0          # empty is a no-op
    cl = ctypes.c_long
    lib.pn2x_mlp2_rows.argtypes = [cl, ci, ci, ci, vp, ci, vp, vp, vp, vp, vp, vp, ci, vp]
    assert lib.pn2x_mlp2_rows(-1, 128, 128, 128, None, 128, None, None, None, None, None, None, 128, None) == -1  # rows < 0
    assert lib.pn2x_mlp2_rows(8, 128, 128, 128, None, 126, None, None, None, None, None, None, 128, None) == -1   # ldx < c1
    assert lib.pn2x_mlp2_rows(0, 128, 128, 128, None, 128, None, None, None, None, None, None, 128, None) == 0    # no rows: no-op
    assert lib.pn2x_mlp2_rows(8, 128, 128, 128, None, 128, None, None, None, None, None, None, 130, None) == -1   # ldo not a multiple of 4
    assert lib.pn2x_mlp2_rows(8, 128, 128, 128, None, 128, None, None, None, None, None, None, 128, None) == -2   # NULL pointers
    assert lib.pn2x_mlp2_rows_supported(128, 128, 128) == 1 and lib.pn2x_mlp2_rows_supported(96, 96, 96) == 0
    lib.pn2x_ln_linear_small.argtypes = [ci, ci, ci, vp, vp, vp, vp, vp, cf, vp, vp, cf, vp, vp, ci, vp, ci, vp, ci, vp]
    assert lib.pn2x_ln_linear_small(21, 385, 64, None, None, None, None, None, 1e-5, None, None, 0.0, None, None, 385, None, 0, None, 64, None) == -1  # k > 384
    assert lib.pn2x_ln_linear_small(21, 192, 64, None, None, None, None, None, 1e-5, None, None, 0.0, None, None, 192, None, 0, None, 64, None) == -2  # NULL
    assert lib.pn2x_ln_linear_small(0, 192, 64, None, None, None, None, None, 1e-5, None, None, 0.0, None, None, 192, None, 0, None, 64, None) == 0   # no rows
    lib.pn2x_kabsch_backward.argtypes = [ci, ci, ci] + [vp] * 7
    assert lib.pn2x_kabsch_backward(4, 2, 6, None, None, None, None, None, None, None) == -1   # x batch neither 1 nor b
    assert lib.pn2x_kabsch_backward(4, 1, 6, None, None, None, None, None, None, None) == -2   # NULL pointers
    assert lib.pn2x_kabsch_backward(0, 1, 6, None, None, None, None, None, None, None) == 0    # empty batch
    assert lib.pn2x_sa_set_compute_units(-3) == -1 and lib.pn2x_sa_set_compute_units(0) == 0
    lib.pn2x_sa_mlp_max.argtypes = [ci] * 7 + [vp, ci, vp, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, cl, ci, ci, vp]
    # a row stride whose byte offsets would not fit the 24-bit address multiplier of the gathers is refused, not mis-addressed
    one = ctypes.c_void_p(16)
    assert lib.pn2x_sa_mlp_max(1, 64, 4, 16, 128, 128, 192, one, 1 << 23, one, one, one, one, None, 0, one, one, one, one, one, one, 768, 192, 1, None) == -3


def test_python_boundary_exports_reference_names():
    from hotrack_amd import pointnet2_hip, pointnet2_utils
    for n in pointnet2_hip.EXPORTED:
        assert callable(getattr(pointnet2_hip, n))
    for n in ("furthest_point_sample", "gather_operation", "knn", "three_nn", "three_interpolate", "grouping_operation",
              "ball_query", "QueryAndGroup", "GroupAll", "KNNAndGroup", "FurthestPointSampling", "GatherOperation", "KNN",
              "ThreeNN", "ThreeInterpolate", "GroupingOperation", "BallQuery"):
        assert hasattr(pointnet2_utils, n)


def test_cpu_tensors_fail_loudly():
    from hotrack_amd import pointnet2_utils as ops
    xyz = torch.rand(1, 32, 3)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.furthest_point_sample(xyz, 8)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.ball_query(0.1, 4, xyz, xyz[:, :4].contiguous())
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.three_nn(xyz, xyz)
    with pytest.raises(TypeError):
        ops.furthest_point_sample(xyz.double(), 8)


def test_product_code_never_imports_the_oracle():
    bad = []
    for base in ("hotrack_amd", "network", "configs", "datasets"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp")):
                    txt = open(os.path.join(dp, f)).read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "pn2_oracle" in txt:
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_tuned_gemm_table_is_well_formed():
    from hotrack_amd import gemm_tuning
    rows = [l.strip().split(",") for l in open(gemm_tuning.RESULTS) if l.strip()]
    assert any(r[0] == "Validator" and r[1] == "GCN_ARCH_NAME" and r[2].startswith("gfx950") for r in rows)
    ops = [r for r in rows if r[0] != "Validator"]
    assert len(ops) > 50 and all(len(r) >= 3 and r[0].startswith("Gemm") for r in ops)
    rec = gemm_tuning.recorded_on()  # the library build the solution indices are valid for travels with the table
    assert rec.get("HIPBLASLT_VERSION") and rec.get("ROCBLAS_VERSION") and rec.get("PT_VERSION")
    if not torch.cuda.is_available():
        assert gemm_tuning.enable() is False  # no GPU: nothing is touched
        assert gemm_tuning.status()["gemm_table"] == "off"


def test_build_refuses_kernels_with_scratch_memory(tmp_path):
    """The build parses hipcc's kernel-resource-usage remarks and fails on any kernel that uses scratch (private) memory
    (hot tg_bwd / tg_dgrad instantiations kept a register array there unnoticed)."""
    import subprocess
    from hotrack_amd import _build
    sample = ("a.hip:1:1: remark: Function Name: _Z3foov [-Rpass-analysis=kernel-resource-usage]\n"
              "a.hip:1:1: remark:     ScratchSize [bytes/lane]: 0 [-Rpass-analysis=kernel-resource-usage]\n"
              "a.hip:9:1: remark: Function Name: _Z3barv [-Rpass-analysis=kernel-resource-usage]\n"
              "a.hip:9:1: remark:     ScratchSize [bytes/lane]: 48 [-Rpass-analysis=kernel-resource-usage]\n")
    assert _build.scratch_kernels(sample) == {"_Z3barv": 48}
    # a real compile: a dynamically indexed private array must live in scratch memory
    src = tmp_path / "scr.hip"
    src.write_text('#include <hip/hip_runtime.h>\n__global__ void k(float *o, const int *i, int n) {\n  float a[64];\n'
                   '  for (int j = 0; j < 64; ++j) a[j] = o[j];\n  for (int j = 0; j < n; ++j) a[i[j] & 63] += 1.f;\n'
                   '  o[threadIdx.x] = a[i[threadIdx.x] & 63];\n}\n')
    p = subprocess.run([_build._hipcc(), *_build.HIPCC_FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", str(src), "-o",
                        str(tmp_path / "scr.o")], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-500:]
    assert list(_build.scratch_kernels(p.stderr).values())[0] >= 256


def test_streams_in_flight_warns_once_about_hardware_queues(monkeypatch):
    """several captured forwards in flight need GPU_MAX_HW_QUEUES >= streams + 1 before the runtime starts;
    hotrack_amd.streams_in_flight says so once instead of silently running the streams behind each other."""
    import warnings
    import hotrack_amd
    monkeypatch.setattr(hotrack_amd, "_hw_queue_warned", False)
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert hotrack_amd.streams_in_flight(1) is True and not w
        assert hotrack_amd.streams_in_flight(4) is False and len(w) == 1 and "GPU_MAX_HW_QUEUES=8" in str(w[0].message)
        assert hotrack_amd.streams_in_flight(4) is False and len(w) == 1   # once
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "8")
    assert hotrack_amd.streams_in_flight(4) is True
