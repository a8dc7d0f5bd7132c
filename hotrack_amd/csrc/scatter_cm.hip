// scatter_cm.hip -- atomics-free backward of group_points / gather_points / three_interpolate on the reference's
// channel-major layout (grad_out (B,C,M), grad_points (B,C,N)).
//
// Replaces the data path of group_points_grad_kernel_fast / gather_points_grad_kernel_fast / three_interpolate_grad_kernel_fast
// (reference group_points_gpu.cu:8-25, sampling_gpu.cu:46-63, interpolate_gpu.cu:192-214: one global atomicAdd per element).
// fp32 atomics -- global or LDS -- retire about one lane per clock per CU on MI355X, which held the LDS-slab kernels of
// round 1 at 0.02-0.14 of the HBM roofline.  Here the index list of each cloud is inverted once (counting sort by target:
// offsets + order, train_ops.hip) and a workgroup that owns (cloud, cc channels) stages its [cc][M] slice of grad_out in LDS
// with coalesced loads; every (target i, channel) then SUMS its contributions with plain LDS reads and adds the result to
// grad_points with one coalesced read-modify-write (the reference's accumulate-into-the-caller's-buffer semantics).
#include "pn2_common.h"
#include "../../include/pn2_ext.h"

namespace pn2 {

int *StreamScratch::acquire(size_t count, hipStream_t stream) {
    st = stream;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return nullptr;
    }
    if (hipMallocAsync((void **)&p, (count ? count : 1) * sizeof(int), st) != hipSuccess) {
        (void)hipGetLastError();
        p = nullptr;
    }
    return p;
}

StreamScratch::~StreamScratch() {
    if (p && hipFreeAsync(p, st) != hipSuccess) (void)hipGetLastError();  // after the kernels enqueued in between, in stream order
}

constexpr int kScT = 1024;  // 16 waves per workgroup: the LDS slab allows only ~2 workgroups per CU, the waves hide the update latency

// LDS: [cc][m_src] slice of grad_out | offsets (n_dst + 1) | order (L) | weights (L, T == 3): the inverted lists are walked
// from LDS too (per-item dependent global loads -- offsets, then order, then the value -- were a latency chain).
template <int T>
__global__ void __launch_bounds__(kScT)
cm_segment_sum_kernel(int c, int n_dst, int m_src, int cc, const float *__restrict__ grad_out_all, const int *__restrict__ offsets_all,
                      const int *__restrict__ order_all, const float *__restrict__ weight_all, float *__restrict__ grad_points_all) {
    extern __shared__ __attribute__((aligned(16))) float G[];
    const int L = m_src * T;
    int *loff = reinterpret_cast<int *>(G + (size_t)cc * m_src);
    int *lord = loff + n_dst + 1;
    float *lw = reinterpret_cast<float *>(lord + L);
    const int b = blockIdx.y, c0 = blockIdx.x * cc;
    const int nc = (c - c0) < cc ? (c - c0) : cc;
    const float *__restrict__ src = grad_out_all + ((size_t)b * c + c0) * m_src;
    const int tot = nc * m_src;
    if ((((uintptr_t)src) & 15) == 0 && (tot & 3) == 0) {
        for (int i = threadIdx.x * 4; i < tot; i += kScT * 4) *reinterpret_cast<float4 *>(G + i) = *reinterpret_cast<const float4 *>(src + i);
    } else {
        for (int i = threadIdx.x; i < tot; i += kScT) G[i] = src[i];
    }
    const int *__restrict__ offsets = offsets_all + (size_t)b * (n_dst + 1);
    const int *__restrict__ order = order_all + (size_t)b * L;
    for (int i = threadIdx.x; i <= n_dst; i += kScT) loff[i] = offsets[i];
    for (int i = threadIdx.x; i < L; i += kScT) lord[i] = order[i];
    if constexpr (T == 3) {
        const float *__restrict__ weight = weight_all + (size_t)b * L;
        for (int i = threadIdx.x; i < L; i += kScT) lw[i] = weight[i];
    }
    __syncthreads();
    float *__restrict__ dst = grad_points_all + ((size_t)b * c + c0) * n_dst;
    const int total = n_dst * nc;
    constexpr int U = 4;  // items per thread in flight
    for (int item0 = threadIdx.x; item0 < total; item0 += U * kScT) {
        int p0[U], p1[U], chn[U];
        float d[U], acc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {  // list bounds first (LDS): targets nobody contributed to cost no HBM traffic at all
            const int item = item0 + u * kScT;
            p0[u] = p1[u] = 0;
            chn[u] = 0;
            if (item < total) {
                const int ch = item / n_dst, i = item - ch * n_dst;  // consecutive threads -> consecutive targets: coalesced update
                chn[u] = ch;
                p0[u] = loff[i];
                p1[u] = loff[i + 1];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) d[u] = p1[u] > p0[u] ? dst[item0 + u * kScT] : 0.f;  // the U loads are issued together
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float *__restrict__ g = G + chn[u] * m_src;
            acc[u] = 0.f;
            for (int p = p0[u]; p < p1[u]; ++p) {
                const int e = lord[p];
                if constexpr (T == 1) acc[u] += g[e];
                else acc[u] += lw[e] * g[e / 3];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (p1[u] > p0[u]) dst[item0 + u * kScT] = d[u] + acc[u];
    }
}

// ---- long position lists (round 6) ---------------------------------------------------------------------------------------
// The kernel above keeps a cloud's whole [cc][m_src] slice and its lists in LDS: m_src + lists <= 16384 words.  The stress shape
// of BASELINE.json configs[4] (n = 8192 points, 2048 x 64 = 131072 grouped positions, C = 67) is far beyond that and fell through
// to the LDS-atomic slab kernel of round 1: 515 us = 0.59 TB/s, because ds_add_f32 retires ~0.33 lanes per clock per CU on
// this part whatever the address pattern (scripts/probes/lds_atomic: 794 adds / us / CU unique, 600 same-address).
// Both sides of a scatter want LDS (a 4-byte gather from HBM / L2 pulls a whole line per lane), so the list is cut into chunks
// of Mt consecutive SOURCE positions: chunk k of a cloud is inverted on its own (offsets over all targets, order within the
// chunk -- the same counting sort, run on b x nchunks "clouds"), a workgroup that owns (cloud, 2 channels) walks the chunks,
// stages the chunk's [2][Mt] slice of grad_out (contiguous, coalesced) and its order list in LDS, and every thread keeps the
// running sums of ITS targets (thread t owns targets t, t + NT, ...) in registers across all chunks; one coalesced
// read-modify-write of grad_points at the end.  No atomics, grad_out read exactly once, lists re-read from L2 by the
// workgroups of the other channels.
constexpr int kChCC = 2;     // channels per workgroup (parallelism: b * ceil(c / 2) workgroups)

template <int TPT, int NT>
__global__ void __launch_bounds__(NT)
cm_chunked_sum_kernel(int c, int n_dst, int m_src, int mt, int nchunks, const float *__restrict__ grad_out_all,
                      const int *__restrict__ offsets_all, const int *__restrict__ order_all, float *__restrict__ grad_points_all) {
    extern __shared__ __attribute__((aligned(16))) float G[];  // [kChCC][mt] | order [mt]
    int *lord = reinterpret_cast<int *>(G + (size_t)kChCC * mt);
    const int b = blockIdx.y, c0 = blockIdx.x * kChCC;
    const int nc = (c - c0) < kChCC ? (c - c0) : kChCC;
    const float *__restrict__ src = grad_out_all + ((size_t)b * c + c0) * m_src;
    const int i_base = blockIdx.z * (TPT * NT);  // clouds of more than TPT * NT points: one workgroup per target range
    float acc[TPT][kChCC];
#pragma unroll
    for (int j = 0; j < TPT; ++j)
#pragma unroll
        for (int u = 0; u < kChCC; ++u) acc[j][u] = 0.f;
    for (int k = 0; k < nchunks; ++k) {
        const int e0 = k * mt;
        const int len = (m_src - e0) < mt ? (m_src - e0) : mt;
        const int *__restrict__ off = offsets_all + ((size_t)b * nchunks + k) * (n_dst + 1);
        const int *__restrict__ ord = order_all + (size_t)b * m_src + e0;
        // this thread's list bounds (registers), requested together with the slab
        int p0[TPT], p1[TPT];
#pragma unroll
        for (int j = 0; j < TPT; ++j) {
            const int i = i_base + threadIdx.x + j * NT;
            p0[j] = i < n_dst ? off[i] : 0;
            p1[j] = i < n_dst ? off[i + 1] : 0;
        }
        __syncthreads();  // the previous chunk's readers are done with G / lord
        for (int u = 0; u < nc; ++u) {
            const float *__restrict__ row = src + (size_t)u * m_src + e0;
            float *__restrict__ dstl = G + (size_t)u * mt;
            if ((((uintptr_t)row) & 15) == 0) {
                int i = threadIdx.x * 4;
                for (; i + 3 < len; i += NT * 4) *reinterpret_cast<float4 *>(dstl + i) = *reinterpret_cast<const float4 *>(row + i);
                for (; i < len; ++i) dstl[i] = row[i];  // (the last, partial quad of the one thread that meets it)
            } else {
                for (int i = threadIdx.x; i < len; i += NT) dstl[i] = row[i];
            }
        }
        if ((((uintptr_t)ord) & 15) == 0) {
            int i = threadIdx.x * 4;
            for (; i + 3 < len; i += NT * 4) *reinterpret_cast<int4 *>(lord + i) = *reinterpret_cast<const int4 *>(ord + i);
            for (; i < len; ++i) lord[i] = ord[i];
        } else {
            for (int i = threadIdx.x; i < len; i += NT) lord[i] = ord[i];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TPT; ++j) {
            for (int p = p0[j]; p < p1[j]; ++p) {
                const int e = lord[p];
#pragma unroll
                for (int u = 0; u < kChCC; ++u) acc[j][u] += G[(size_t)u * mt + e];  // (row 1 of a one-channel tail holds stale data: never stored)
            }
        }
    }
    float *__restrict__ dst = grad_points_all + ((size_t)b * c + c0) * n_dst;
#pragma unroll
    for (int j = 0; j < TPT; ++j) {
        const int i = i_base + threadIdx.x + j * NT;
        if (i < n_dst) {
            for (int u = 0; u < nc; ++u) dst[(size_t)u * n_dst + i] += acc[j][u];
        }
    }
}

// chunk length of the long-list path (0: the shape takes the one-slab kernel or is not covered at all)
static int chunked_mt(int t, int n_dst, int m_src) {
    if (t != 1 || n_dst > 15 * 1024) return 0;  // (the chunk inversion keeps n_dst + 1 counters in 64 KiB of LDS)
    const size_t l = (size_t)m_src;
    const size_t meta = (size_t)n_dst + 1 + l;
    if (!((size_t)n_dst + 1 + 256 > 16384 || meta + m_src > 16384)) return 0;  // fits the one-slab kernel
    return m_src < 8192 ? ((m_src + 3) / 4 * 4) : 8192;
}

size_t scatter_cm_scratch_ints(int t, int b, int n_dst, int m_src) {
    if (const int mt = chunked_mt(t, n_dst, m_src)) {
        const size_t nchunks = ((size_t)m_src + mt - 1) / mt;
        return (size_t)b * (nchunks * ((size_t)n_dst + 1) + (size_t)m_src);
    }
    return (size_t)b * ((size_t)n_dst + 1 + (size_t)m_src * t);
}

template <int TPT, int NT>
static int launch_chunked(int b, int c, int n_dst, int m_src, int mt, int nchunks, const float *grad_out, const int *offsets,
                          const int *order, float *grad_points, hipStream_t st) {
    const size_t lds = ((size_t)kChCC * mt + mt) * sizeof(float);
    static bool attr_ok = false, attr_tried = false;  // (per kernel instantiation)
    if (!attr_tried) {
        attr_tried = true;
        attr_ok = hipFuncSetAttribute(reinterpret_cast<const void *>(&cm_chunked_sum_kernel<TPT, NT>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) == hipSuccess;
        if (!attr_ok) (void)hipGetLastError();
    }
    if (lds > 64 * 1024 && !attr_ok) return PN2_ERANGE;  // (the caller falls back to the slab / atomic kernels)
    hipLaunchKernelGGL((cm_chunked_sum_kernel<TPT, NT>), dim3((c + kChCC - 1) / kChCC, b, (n_dst + TPT * NT - 1) / (TPT * NT)), dim3(NT), lds, st,
                       c, n_dst, m_src, mt, nchunks,
                       grad_out, offsets, order, grad_points);
    return check_launch();
}

int scatter_cm_dispatch(int t, int b, int c, int n_dst, int m_src, const float *grad_out, const int *idx, const float *weight,
                        float *grad_points, hipStream_t st, int *scratch, size_t scratch_ints) {
    if (b == 0 || c == 0 || m_src == 0) return PN2_OK;
    const size_t need = scatter_cm_scratch_ints(t, b, n_dst, m_src);
    const int mt = chunked_mt(t, n_dst, m_src);
    const size_t l = (size_t)m_src * t;
    const size_t meta = (size_t)n_dst + 1 + l * (t == 3 ? 2 : 1);  // ints / floats next to the slab
    if (!mt && ((size_t)n_dst + 1 + 256 > 16384 || meta + m_src > 16384)) return PN2_ERANGE;  // 64 KiB of LDS per workgroup
    StreamScratch own;  // released in stream order when this call returns (after the launches below)
    if (!scratch) {
        scratch = own.acquire(need, st);
        if (!scratch) return PN2_ERANGE;
    } else if (scratch_ints < need) {
        return PN2_ESCRATCH;
    }
    if (mt) {
        const int nchunks = (m_src + mt - 1) / mt;
        int *offsets = scratch, *order = scratch + (size_t)b * nchunks * ((size_t)n_dst + 1);
        int rc = inverse_index_chunked_launch(b, n_dst, m_src, mt, nchunks, idx, offsets, order, st);
        if (rc != PN2_OK) return rc;
        constexpr int NT = 1024;
        if (n_dst <= 4 * NT) return launch_chunked<4, NT>(b, c, n_dst, m_src, mt, nchunks, grad_out, offsets, order, grad_points, st);
        return launch_chunked<8, NT>(b, c, n_dst, m_src, mt, nchunks, grad_out, offsets, order, grad_points, st);
    }
    int cc = (int)((16384 - meta) / m_src);
    if (cc > 16) cc = 16;
    if (cc > c) cc = c;
    while (cc > 1 && (long)b * ((c + cc - 1) / cc) < 512) cc = (cc + 1) / 2;  // >= 2 workgroups of 16 waves per CU
    int *offsets = scratch, *order = scratch + (size_t)b * (n_dst + 1);
    int rc = inverse_index_launch(b, n_dst, (int)l, idx, offsets, order, st);
    if (rc != PN2_OK) return rc;
    const dim3 grid((c + cc - 1) / cc, b);
    const size_t lds = ((size_t)cc * m_src + meta) * sizeof(float);
    if (t == 1)
        hipLaunchKernelGGL(cm_segment_sum_kernel<1>, grid, dim3(kScT), lds, st, c, n_dst, m_src, cc, grad_out, offsets, order, weight, grad_points);
    else
        hipLaunchKernelGGL(cm_segment_sum_kernel<3>, grid, dim3(kScT), lds, st, c, n_dst, m_src, cc, grad_out, offsets, order, weight, grad_points);
    return check_launch();
}

}  // namespace pn2

// Operator-API backward with caller-provided scratch (graph-capture safe: nothing is allocated inside):
//   t = 1  group_points_grad / gather_points_grad (m_src = npoints * nsample positions, idx (b, m_src))
//   t = 3  three_interpolate_grad                 (m_src = n query points, idx / weight (b, n, 3))
extern "C" long pn2x_scatter_cm_scratch_ints(int t, int b, int n_dst, int m_src) {
    if ((t != 1 && t != 3) || b < 0 || n_dst < 1 || m_src < 0) return -1;
    return (long)pn2::scatter_cm_scratch_ints(t, b, n_dst, m_src);
}

extern "C" int pn2x_scatter_cm(int t, int b, int c, int n_dst, int m_src, const float *grad_out, const int *idx, const float *weight,
                               float *grad_points, int *scratch, long scratch_ints, void *stream) {
    using namespace pn2;
    if ((t != 1 && t != 3) || b < 0 || c < 0 || n_dst < 1 || m_src < 0 || scratch_ints < 0) return PN2_EINVAL;
    if (b == 0 || c == 0 || m_src == 0) return PN2_OK;
    if (!grad_out || !idx || !grad_points || !scratch || (t == 3 && !weight)) return PN2_ENULL;
    return scatter_cm_dispatch(t, b, c, n_dst, m_src, grad_out, idx, weight, grad_points, (hipStream_t)stream, scratch, (size_t)scratch_ints);
}
