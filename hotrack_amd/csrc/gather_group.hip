// gather_group.hip -- gather_points / group_points forward and backward for gfx950.
//
// Replaces gather_points_kernel_fast / _grad_ (reference sampling_gpu.cu:8-83) and
// group_points_kernel_fast / _grad_ (group_points_gpu.cu:8-86).  gather is group with
// nsample = 1, so both share one pair of kernels.
//
// Forward is pure HBM streaming: the reference launches grid.y = C and re-reads the index
// tile for every channel.  Here a thread owns VEC consecutive output positions, loads their
// indices ONCE (one 16-byte load), then walks a channel range: VEC L2-resident gathers and
// one 16-byte coalesced store per channel, several channels in flight.
//
// Backward: the reference issues one global fp32 atomicAdd per gradient element.  Here a
// workgroup owns (cloud, CC channels), accumulates them in an LDS-private [CC][N] slab with
// ds_add_f32 and adds the slab to grad_points once with coalesced read-modify-writes -- no
// global atomics, no contention outside the CU.
#include "pn2_common.h"

namespace pn2 {

constexpr int kGgThreads = 256;

template <int VEC>
__global__ void __launch_bounds__(kGgThreads)
group_fwd_kernel(int c, int n, int ps, int c_per_block, const float *__restrict__ points_all,
                 const int *__restrict__ idx_all, float *__restrict__ out_all) {
    const int b = blockIdx.z;
    const int e0 = (blockIdx.x * kGgThreads + threadIdx.x) * VEC;
    if (e0 >= ps) return;
    const int c0 = blockIdx.y * c_per_block;
    const int c1 = (c0 + c_per_block) < c ? (c0 + c_per_block) : c;
    const int *__restrict__ idx = idx_all + (size_t)b * ps + e0;
    int id[VEC];
    if constexpr (VEC == 4) {
        const int4 v = *reinterpret_cast<const int4 *>(idx);
        id[0] = v.x; id[1] = v.y; id[2] = v.z; id[3] = v.w;
    } else {
        id[0] = idx[0];
    }
    const float *__restrict__ src = points_all + ((size_t)b * c + c0) * n;
    float *__restrict__ dst = out_all + ((size_t)b * c + c0) * ps + e0;
#pragma unroll 4
    for (int ch = c0; ch < c1; ++ch) {
        if constexpr (VEC == 4) {
            float4 v;
            v.x = src[id[0]]; v.y = src[id[1]]; v.z = src[id[2]]; v.w = src[id[3]];
            *reinterpret_cast<float4 *>(dst) = v;
        } else {
            dst[0] = src[id[0]];
        }
        src += n;
        dst += ps;
    }
}

// Large clouds: when the (cloud, channel) rows being gathered from no longer stay in L2 (a row is n floats, thousands of
// workgroups work on different rows at once), every 4-byte gather pulls a 128-byte line from HBM: 0.15 of the HBM roofline
// at n = 8192.  Here a workgroup stages CC channel rows of one cloud in LDS (coalesced, read once), then streams its share of
// the positions: 16-byte index loads, LDS gathers, 16-byte coalesced stores -- what is left is the output write.
constexpr int kGgBig = 1024;
__global__ void __launch_bounds__(kGgBig)
group_fwd_lds_kernel(int c, int n, int ps, int cc, int e_per_block, const float *__restrict__ points_all,
                     const int *__restrict__ idx_all, float *__restrict__ out_all) {
    extern __shared__ __attribute__((aligned(16))) float slab[];  // [cc][n]
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * cc;
    const int nc = (c - c0) < cc ? (c - c0) : cc;
    const float *__restrict__ src = points_all + ((size_t)b * c + c0) * n;
    const int total = nc * n;
    if ((((uintptr_t)src) % 16 == 0) && (total % 4 == 0)) {
        for (int i = threadIdx.x * 4; i < total; i += kGgBig * 4) *reinterpret_cast<float4 *>(slab + i) = *reinterpret_cast<const float4 *>(src + i);
    } else {
        for (int i = threadIdx.x; i < total; i += kGgBig) slab[i] = src[i];
    }
    __syncthreads();
    const int e_begin = blockIdx.x * e_per_block;
    const int e_end = (e_begin + e_per_block) < ps ? (e_begin + e_per_block) : ps;
    const int *__restrict__ idx = idx_all + (size_t)b * ps;
    float *__restrict__ dst0 = out_all + ((size_t)b * c + c0) * ps;
    for (int e0 = e_begin + threadIdx.x * 4; e0 < e_end; e0 += kGgBig * 4) {  // ps % 4 == 0 and e_per_block % 4 == 0
        int4 id = *reinterpret_cast<const int4 *>(idx + e0);
        id.x = lds_index(id.x, n); id.y = lds_index(id.y, n); id.z = lds_index(id.z, n); id.w = lds_index(id.w, n);
        const float *row = slab;
        float *dst = dst0 + e0;
#pragma unroll 4
        for (int ch = 0; ch < nc; ++ch) {
            *reinterpret_cast<float4 *>(dst) = make_float4(row[id.x], row[id.y], row[id.z], row[id.w]);
            row += n;
            dst += ps;
        }
    }
}

int group_fwd_dispatch(int b, int c, int n, int npoints, int nsample, const float *points, const int *idx,
                       float *out, hipStream_t st) {
    const long ps_l = (long)npoints * nsample;
    if (b == 0 || c == 0 || ps_l == 0) return PN2_OK;  // C == 0: HandTrackNet sa1 groups zero-channel features
    const int ps = (int)ps_l;
    const bool vec4 = (ps % 4 == 0) && (((uintptr_t)idx | (uintptr_t)out) % 16 == 0);
    // LDS-staged variant: rows too many / too long for L2 (see above) and every staged element gathered often enough
    const long lds_cap = 128 * 1024;
    if (vec4 && (long)n * 4 <= lds_cap && (long)b * c * n * 4 >= (8L << 20) && ps_l >= 8L * n) {
        int cc = (int)(lds_cap / ((long)n * 4));
        if (cc > c) cc = c;
        if (cc > 16) cc = 16;
        const int ych = (c + cc - 1) / cc;
        int xb = (int)((1024 + (long)ych * b - 1) / ((long)ych * b));  // >= ~1024 workgroups, but a workgroup's output stays
        const int xb_max = (int)(ps_l / (8L * n));                                       // >= 8x the elements it stages (>= 1 here)
        if (xb > xb_max) xb = xb_max;
        if (xb < 1) xb = 1;
        int e_per_block = (int)(((ps_l + xb - 1) / xb + 3) / 4 * 4);
        xb = (int)((ps_l + e_per_block - 1) / e_per_block);
        const size_t lds = (size_t)cc * n * sizeof(float);
        static PerDeviceOnce raised;
        if (lds > 64 * 1024 && raised.first_use())
            (void)hipFuncSetAttribute((const void *)group_fwd_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap);
        hipLaunchKernelGGL(group_fwd_lds_kernel, dim3(xb, ych, b), dim3(kGgBig), lds, st, c, n, ps, cc, e_per_block, points, idx, out);
        return check_launch();
    }
    const int per_block = kGgThreads * (vec4 ? 4 : 1);
    const int xb = (ps + per_block - 1) / per_block;
    // split channels so the launch reaches >= ~2048 workgroups when the problem allows it
    int ysplit = (int)((2048 + (long)xb * b - 1) / ((long)xb * b));
    if (ysplit < 1) ysplit = 1;
    if (ysplit > c) ysplit = c;
    const int c_per_block = (c + ysplit - 1) / ysplit;
    ysplit = (c + c_per_block - 1) / c_per_block;
    dim3 grid(xb, ysplit, b);
    if (vec4)
        hipLaunchKernelGGL(group_fwd_kernel<4>, grid, dim3(kGgThreads), 0, st, c, n, ps, c_per_block, points, idx, out);
    else
        hipLaunchKernelGGL(group_fwd_kernel<1>, grid, dim3(kGgThreads), 0, st, c, n, ps, c_per_block, points, idx, out);
    return check_launch();
}

// ---- backward ---------------------------------------------------------------------------
// A work item = (position e, chunk of kGgU channels): the kGgU gradient loads of an item are independent and issued
// together (a thread walking the channels one load at a time sat on one HBM round trip per element: 0.02-0.14 of the
// HBM roofline), then the LDS adds follow; consecutive threads take consecutive positions -> coalesced rows of grad_out.
constexpr int kGgU = 8;
__global__ void __launch_bounds__(kGgBig)
group_bwd_lds_kernel(int c, int n, int ps, int cc, const float *__restrict__ grad_out_all,
                     const int *__restrict__ idx_all, float *__restrict__ grad_points_all) {
    extern __shared__ __attribute__((aligned(16))) float acc[];  // [cc][n]
    const int b = blockIdx.y;
    const int c0 = blockIdx.x * cc;
    const int nc = (c - c0) < cc ? (c - c0) : cc;
    for (int i = threadIdx.x; i < nc * n; i += (int)blockDim.x) acc[i] = 0.f;
    __syncthreads();
    const int *__restrict__ idx = idx_all + (size_t)b * ps;
    const float *__restrict__ g = grad_out_all + ((size_t)b * c + c0) * ps;
    const int chunks = (nc + kGgU - 1) / kGgU;
    for (int item = threadIdx.x; item < ps * chunks; item += (int)blockDim.x) {
        const int chunk = item / ps, e = item - chunk * ps;
        const int ch0 = chunk * kGgU;
        const int id = lds_index(idx[e], n);
        float v[kGgU];
#pragma unroll
        for (int u = 0; u < kGgU; ++u) v[u] = (ch0 + u < nc) ? g[(size_t)(ch0 + u) * ps + e] : 0.f;
#pragma unroll
        for (int u = 0; u < kGgU; ++u)
            if (ch0 + u < nc) atomicAdd(&acc[(ch0 + u) * n + id], v[u]);
    }
    __syncthreads();
    float *__restrict__ dst = grad_points_all + ((size_t)b * c + c0) * n;
    const int total = nc * n;
    int i = threadIdx.x;
    for (; i + 3 * (int)blockDim.x < total; i += 4 * (int)blockDim.x) {  // four independent read-modify-writes in flight
        const float d0 = dst[i], d1 = dst[i + (int)blockDim.x], d2 = dst[i + 2 * (int)blockDim.x], d3 = dst[i + 3 * (int)blockDim.x];
        dst[i] = d0 + acc[i];
        dst[i + (int)blockDim.x] = d1 + acc[i + (int)blockDim.x];
        dst[i + 2 * (int)blockDim.x] = d2 + acc[i + 2 * (int)blockDim.x];
        dst[i + 3 * (int)blockDim.x] = d3 + acc[i + 3 * (int)blockDim.x];
    }
    for (; i < total; i += (int)blockDim.x) dst[i] += acc[i];
}

__global__ void __launch_bounds__(kGgThreads)
group_bwd_atomic_kernel(int c, int n, int ps, const float *__restrict__ grad_out_all,
                        const int *__restrict__ idx_all, float *__restrict__ grad_points_all) {
    const int b = blockIdx.z, ch = blockIdx.y;
    const int e = blockIdx.x * kGgThreads + threadIdx.x;
    if (e >= ps) return;
    const int id = idx_all[(size_t)b * ps + e];
    unsafeAtomicAdd(grad_points_all + ((size_t)b * c + ch) * n + id, grad_out_all[((size_t)b * c + ch) * ps + e]);
}

int group_bwd_dispatch(int b, int c, int n, int npoints, int nsample, const float *grad_out, const int *idx,
                       float *grad_points, hipStream_t st) {
    const long ps_l = (long)npoints * nsample;
    if (b == 0 || c == 0 || ps_l == 0) return PN2_OK;
    const int ps = (int)ps_l;
    // inverted index + LDS-staged segment sums (scatter_cm.hip); shapes it does not cover (very long position lists, a
    // scratch that would have to grow during graph capture) take the LDS-atomic slab kernel below
    if (const int rc = scatter_cm_dispatch(1, b, c, n, ps, grad_out, idx, nullptr, grad_points, st); rc != PN2_ERANGE) return rc;  // only "shape not covered" falls through
    const int lds_budget = 64 * 1024;
    int cc = lds_budget / (int)(sizeof(float) * (size_t)n);
    if (cc >= 1) {
        if (cc > 16) cc = 16;
        if (cc > c) cc = c;
        // keep enough workgroups in flight: shrink the channel slab while the grid is small
        while (cc > 1 && (long)b * ((c + cc - 1) / cc) < 1024) cc = (cc + 1) / 2;
        dim3 grid((c + cc - 1) / cc, b);
        // a slab of >= 16 KB leaves room for few workgroups per CU: 1024 threads each keep its gradient loads in flight
        const int threads = (size_t)cc * n * sizeof(float) >= 16 * 1024 ? kGgBig : kGgThreads;
        hipLaunchKernelGGL(group_bwd_lds_kernel, grid, dim3(threads), (size_t)cc * n * sizeof(float), st, c, n, ps, cc,
                           grad_out, idx, grad_points);
    } else {
        dim3 grid((ps + kGgThreads - 1) / kGgThreads, c, b);
        hipLaunchKernelGGL(group_bwd_atomic_kernel, grid, dim3(kGgThreads), 0, st, c, n, ps, grad_out, idx, grad_points);
    }
    return check_launch();
}

}  // namespace pn2
