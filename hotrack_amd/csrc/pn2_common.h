// pn2_common.h -- shared device helpers for the gfx950 PointNet++ operator kernels.
// wave64 only; DPP-based cross-lane reductions (no LDS round trips inside a wave).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pn2_hip.h"

namespace pn2 {

constexpr int kWave = 64;
// fp64 BatchNorm accumulators exist kBnRep times (workgroup b adds into copy b % kBnRep): 8x less same-address contention
// (train_ops.hip, train_gemm.hip; pn2x_bn_sums_doubles reports the resulting size)
constexpr int kBnRep = 8;

// Squared distance in the one fixed contraction order used by oracle and kernels alike:
// fma(dz,dz, fma(dx,dx, dy*dy)).  (reference expression: sampling_gpu.cu:133,
// ball_query_gpu.cu:33, interpolate_gpu.cu:40,108 compiled with nvcc --fmad=true.)
// Two squared distances at once with the packed fp32 VALU ops of gfx90a+ (v_pk_add / v_pk_mul / v_pk_fma_f32: each
// lane-wise IEEE operation is the same as the scalar one, so results are bit-identical to sqdist()).
typedef float pn2_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pn2_f32x2 sqdist2(pn2_f32x2 ax, pn2_f32x2 ay, pn2_f32x2 az, float bx, float by, float bz) {
    const pn2_f32x2 dx = ax - bx, dy = ay - by, dz = az - bz;
    return __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dx, dx, dy * dy));
}

__device__ __forceinline__ float sqdist(float ax, float ay, float az, float bx, float by, float bz) {
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dx, dx, dy * dy));
}

// One fused DPP reduction step:  v = OP(v, lane_permute(v))  as a single VOP2-DPP instruction.
// hipcc does not fold update_dpp+max into one instruction (it emits mov / s_nop / mov_dpp /
// max), so the step is written in asm.  "s_nop 1" = the 2 wait states gfx9 requires between
// a VALU write of a VGPR and a DPP read of it; lanes disabled by row_mask keep v (vdst==src).
#define PN2_DPP_STEP(OPC, V, CTRL) \
    asm volatile("s_nop 1\n\t" OPC " %0, %0, %0 " CTRL : "+v"(V))

// Max over aligned groups of NL lanes (NL in {1,2,4,8,16}) -- butterfly inside a row of 16:
// afterwards every lane of the group holds the group's max.
template <int NL>
__device__ __forceinline__ int row_group_max_i32(int v) {
    if constexpr (NL >= 2) PN2_DPP_STEP("v_max_i32_dpp", v, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
    if constexpr (NL >= 4) PN2_DPP_STEP("v_max_i32_dpp", v, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
    if constexpr (NL >= 8) PN2_DPP_STEP("v_max_i32_dpp", v, "row_half_mirror row_mask:0xf bank_mask:0xf");
    if constexpr (NL >= 16) PN2_DPP_STEP("v_max_i32_dpp", v, "row_mirror row_mask:0xf bank_mask:0xf");
    return v;
}
template <int NL>
__device__ __forceinline__ unsigned row_group_min_u32(unsigned v) {
    if constexpr (NL >= 2) PN2_DPP_STEP("v_min_u32_dpp", v, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
    if constexpr (NL >= 4) PN2_DPP_STEP("v_min_u32_dpp", v, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
    if constexpr (NL >= 8) PN2_DPP_STEP("v_min_u32_dpp", v, "row_half_mirror row_mask:0xf bank_mask:0xf");
    if constexpr (NL >= 16) PN2_DPP_STEP("v_min_u32_dpp", v, "row_mirror row_mask:0xf bank_mask:0xf");
    return v;
}

// Full wave64 signed-int max; result is wave-uniform (read from lane 63 into an SGPR).
// row_bcast:15 feeds lane 15 of each row to the next row (rows 1,3 enabled), row_bcast:31
// feeds lane 31 to rows 2,3: lane 63 ends up with the max of all four rows.
__device__ __forceinline__ int wave_max_i32(int v) {
    v = row_group_max_i32<16>(v);
    PN2_DPP_STEP("v_max_i32_dpp", v, "row_bcast:15 row_mask:0xa bank_mask:0xf");
    PN2_DPP_STEP("v_max_i32_dpp", v, "row_bcast:31 row_mask:0xc bank_mask:0xf");
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    v = row_group_min_u32<16>(v);
    PN2_DPP_STEP("v_min_u32_dpp", v, "row_bcast:15 row_mask:0xa bank_mask:0xf");
    PN2_DPP_STEP("v_min_u32_dpp", v, "row_bcast:31 row_mask:0xc bank_mask:0xf");
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// Full wave64 fp32 sum, same DPP ladder (6 v_add_f32_dpp, no LDS crossbar); wave-uniform result.  The summation order
// is fixed (butterfly inside each row of 16, then rows 0+1, 2+3, then halves), so results are run-to-run identical.
__device__ __forceinline__ float wave_sum_f32(float v) {
    PN2_DPP_STEP("v_add_f32_dpp", v, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
    PN2_DPP_STEP("v_add_f32_dpp", v, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
    PN2_DPP_STEP("v_add_f32_dpp", v, "row_half_mirror row_mask:0xf bank_mask:0xf");
    PN2_DPP_STEP("v_add_f32_dpp", v, "row_mirror row_mask:0xf bank_mask:0xf");
    PN2_DPP_STEP("v_add_f32_dpp", v, "row_bcast:15 row_mask:0xa bank_mask:0xf");
    PN2_DPP_STEP("v_add_f32_dpp", v, "row_bcast:31 row_mask:0xc bank_mask:0xf");
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// fminf without the v_max canonicalisation hipcc inserts in IEEE mode (operands are never sNaN here).
// index into an LDS-staged array of n entries: a corrupt index (negative or >= n) must not read or update another array's
// LDS (the global-memory kernels of the reference do not check either, but there a bad index cannot corrupt a neighbour's sums)
__device__ __forceinline__ int lds_index(int i, int n) {
    const unsigned u = (unsigned)i, hi = (unsigned)(n - 1);
    return (int)(u < hi ? u : hi);
}

__device__ __forceinline__ float fmin_raw(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// fmaxf without the canonicalising v_max hipcc adds in IEEE mode: median(a, b, +inf) == max(a, b) for non-NaN inputs
// and v_med3_f32 is a real instruction to the compiler (NOT inline asm: the hazard recogniser must see a VALU read
// of MFMA accumulators to insert the required wait states -- an asm v_max here read half-written accumulators).
__device__ __forceinline__ float fmax_raw(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, __builtin_inff()); }
__device__ __forceinline__ float fmax3_raw(float a, float b, float c) { return fmax_raw(fmax_raw(a, b), c); }

// max over the four 16-lane rows of a wave64 (lanes l, l^16, l^32, l^48), result in every lane: two gfx950
// v_permlane{16,32}_swap + two v_max, no LDS crossbar round trip (ds_bpermute) and no wait.
__device__ __forceinline__ float rows_max4(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);  // a[0]: rows {0,0,2,2}, a[1]: rows {1,1,3,3}
    v = fmax_raw(__builtin_bit_cast(float, (unsigned)a[0]), __builtin_bit_cast(float, (unsigned)a[1]));
    const unsigned w = __builtin_bit_cast(unsigned, v);
    auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);  // b[0]: lower half twice, b[1]: upper half twice
    return fmax_raw(__builtin_bit_cast(float, (unsigned)b[0]), __builtin_bit_cast(float, (unsigned)b[1]));
}

__device__ __forceinline__ int f2i(float f) { return __builtin_bit_cast(int, f); }
__device__ __forceinline__ float i2f(int i) { return __builtin_bit_cast(float, i); }

__device__ __forceinline__ int lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ int prefix_popc(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// Host-side launch helpers ---------------------------------------------------------------
void set_last_hip_error(int e);
// Compute units of the current device (256 on MI355X), cached per device: persistent-grid sizing.
inline int num_compute_units() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE property of a kernel: a per-instantiation `static`
// of this type remembers which devices have had it raised (a process may drive several GPUs).
struct PerDeviceOnce {
    bool done[64] = {};
    // true exactly once per device (and always for device ids outside the table: the call is idempotent)
    bool first_use() {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
        if (done[dev]) return false;
        done[dev] = true;
        return true;
    }
};

// train_ops.hip: counting-sort inversion of an index list (offsets (b, n_dst+1), order (b, l)); PN2_ERANGE if n_dst does not fit LDS
int inverse_index_launch(int b, int n_dst, int l, const int *idx, int *offsets, int *order, hipStream_t st);
int inverse_index_chunked_launch(int b, int n_dst, int l, int mt, int nchunks, const int *idx, unsigned short *offsets, int off_stride,
                                 unsigned short *order, hipStream_t st);
// Device scratch for kernels whose C ABI (the reference's signatures) has no scratch argument: a STREAM-ORDERED temporary,
// allocated (hipMallocAsync) before the enqueue and released (hipFreeAsync) right after it, in stream order -- the library keeps
// nothing between calls (pn2_hip.h: "keeps no state"); the memory comes from and returns to the HIP runtime's own pool.
// p == nullptr whenever the stream is being captured into a graph (an allocation node would tie the graph to this call) or the
// allocation fails: callers then take their scratch-free path.
struct StreamScratch {
    int *p = nullptr;
    hipStream_t st = nullptr;
    StreamScratch() = default;
    int *acquire(size_t count, hipStream_t stream);  // at most once per object; returns p
    ~StreamScratch();
    StreamScratch(const StreamScratch &) = delete;
    StreamScratch &operator=(const StreamScratch &) = delete;
};
// Atomics-free backward of group / gather / three_interpolate on the channel-major operator layout (scatter_cm.hip):
// t = 1: grad_points[b,c,idx[b,e]] += grad_out[b,c,e];  t = 3: grad_points[b,c,idx[b,j,k]] += w[b,j,k] * grad_out[b,c,j].
// Returns PN2_ERANGE when the shape does not fit (caller falls back to its LDS-atomic kernel).
// scratch == nullptr: a stream-ordered temporary (PN2_ERANGE if it cannot be provided, e.g. during graph capture).
int scatter_cm_dispatch(int t, int b, int c, int n_dst, int m_src, const float *grad_out, const int *idx, const float *weight,
                        float *grad_points, hipStream_t st, int *scratch = nullptr, size_t scratch_ints = 0);

// ball_query_grid.hip: PN2_ERANGE when the shape is not covered (n < 2048, radius <= 0, no scratch): the caller scans
int ball_query_grid_dispatch(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz, int *idx,
                             unsigned *scratch, size_t scratch_words, hipStream_t st);

inline int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_hip_error((int)e);
        return PN2_ELAUNCH;
    }
    return PN2_OK;
}

}  // namespace pn2
