// nn_search.hip -- three_nn and knn for gfx950.
//
// three_nn  replaces three_nn_kernel_fast (reference interpolate_gpu.cu:81-146):
//   thread per query, but the known set is staged once per workgroup in LDS as 16-byte
//   xyz_ records and read with wave-uniform (broadcast) ds_read_b128; the 3-entry cascade is
//   branch-free (v_cmp + v_cndmask), so a wave never diverges.
//
// knn       replaces knn_kernel_fast (interpolate_gpu.cu:9-79), which keeps double[200]
//   per thread in local memory and leaves HandTrackNet's 21 queries/cloud on 21 threads.
//   Here one WAVE owns a query: each lane computes the distances of a contiguous chunk of
//   P candidates into registers as 64-bit keys (dist_bits << 32 | index), sorts them with a
//   compile-time odd-even merge network, and the k outputs are produced by k rounds of
//   "wave-min over the lane heads (DPP) -> first lane holding it pops".  The 64-bit key order
//   is exactly the reference's (distance ascending, index ascending) insertion order.
#include "pn2_common.h"
#include "knn_wave.h"

namespace pn2 {

// ------------------------------------------------------------------------------------------
// three_nn
// ------------------------------------------------------------------------------------------
constexpr int kNnTile = 2048;  // known points per LDS tile (32 KiB as float4)

// WEIGHTS == false: the reference operator (squared distances + indices).
// WEIGHTS == true : pn2x_three_nn_weights -- the same search, but the second output is the normalised
//   inverse-distance interpolation weight the caller of the reference computes in four torch kernels
//   (pointnet_utils.py:446-449: d = sqrt(d2); r = 1/(d + 1e-8); w = r / sum r).
template <int THREADS, bool WEIGHTS>
__global__ void __launch_bounds__(THREADS)
three_nn_kernel(int n, int m, const float *__restrict__ unknown_all, const float *__restrict__ known_all,
                float *__restrict__ dist2_all, int *__restrict__ idx_all) {
    extern __shared__ __attribute__((aligned(16))) float4 sk[];
    const int b = blockIdx.y;
    const float *__restrict__ known = known_all + (size_t)b * m * 3;
    const int q = blockIdx.x * THREADS + threadIdx.x;
    const bool active = q < n;
    const float *__restrict__ u = unknown_all + ((size_t)b * n + (active ? q : 0)) * 3;
    const float ux = u[0], uy = u[1], uz = u[2];

    // double 1e40 sentinels of the reference (interpolate_gpu.cu:102) == +inf for fp32 compares
    float b1 = __builtin_inff(), b2 = __builtin_inff(), b3 = __builtin_inff();
    int i1 = 0, i2 = 0, i3 = 0;

    for (int t0 = 0; t0 < m; t0 += kNnTile) {
        const int tn = (m - t0) < kNnTile ? (m - t0) : kNnTile;
        if (t0 > 0) __syncthreads();
        for (int p = threadIdx.x; p < tn; p += THREADS) {
            const float *src = known + (size_t)3 * (t0 + p);
            sk[p] = make_float4(src[0], src[1], src[2], 0.f);
        }
        __syncthreads();
#pragma unroll 4
        for (int p = 0; p < tn; ++p) {
            const float4 kp = sk[p];  // wave-uniform address -> LDS broadcast
            const float d = sqdist(ux, uy, uz, kp.x, kp.y, kp.z);
            const int k = t0 + p;
            const bool lt1 = d < b1, lt2 = d < b2, lt3 = d < b3;  // strict: ties keep the lower index
            b3 = lt2 ? b2 : (lt3 ? d : b3);
            i3 = lt2 ? i2 : (lt3 ? k : i3);
            b2 = lt1 ? b1 : (lt2 ? d : b2);
            i2 = lt1 ? i1 : (lt2 ? k : i2);
            b1 = lt1 ? d : b1;
            i1 = lt1 ? k : i1;
        }
    }
    if (active) {
        float *od = dist2_all + ((size_t)b * n + q) * 3;
        int *oi = idx_all + ((size_t)b * n + q) * 3;
        if constexpr (WEIGHTS) {
            const float r1 = 1.0f / (__builtin_sqrtf(b1) + 1e-8f), r2 = 1.0f / (__builtin_sqrtf(b2) + 1e-8f),
                        r3 = 1.0f / (__builtin_sqrtf(b3) + 1e-8f);
            const float norm = (r1 + r2) + r3;  // torch.sum over 3 elements adds left to right
            od[0] = r1 / norm; od[1] = r2 / norm; od[2] = r3 / norm;
        } else {
            od[0] = b1; od[1] = b2; od[2] = b3;
        }
        oi[0] = i1; oi[1] = i2; oi[2] = i3;
    }
}

// Small problems (the whole known set in one LDS tile, too few queries to fill the chip with one thread each): the scan
// is a per-thread latency chain of m steps.  Four threads per query -- wave c scans the c-th quarter of the known set
// (still wave-uniform broadcast reads) -- then wave 0 merges the four 3-entry lists in index order with the same
// strict-< cascade, which preserves the (distance, index) order: a later chunk's candidate never overtakes an equal
// distance from an earlier (lower-index) chunk.  7.5 us -> ~3 us for 1024 queries x 256 known per cloud.
template <bool WEIGHTS>
__global__ void __launch_bounds__(256)
three_nn_split_kernel(int n, int m, const float *__restrict__ unknown_all, const float *__restrict__ known_all,
                      float *__restrict__ dist2_all, int *__restrict__ idx_all) {
    extern __shared__ __attribute__((aligned(16))) float4 sk[];
    __shared__ float sd[3][3][64];  // chunks 1..3: their three best distances / indices per query
    __shared__ int si[3][3][64];
    const int b = blockIdx.y;
    const float *__restrict__ known = known_all + (size_t)b * m * 3;
    const int ql = threadIdx.x & 63, c = threadIdx.x >> 6;  // c is wave-uniform
    const int q = blockIdx.x * 64 + ql;
    const bool active = q < n;
    const float *__restrict__ u = unknown_all + ((size_t)b * n + (active ? q : 0)) * 3;
    const float ux = u[0], uy = u[1], uz = u[2];
    for (int p = threadIdx.x; p < m; p += 256) {
        const float *src = known + (size_t)3 * p;
        sk[p] = make_float4(src[0], src[1], src[2], 0.f);
    }
    __syncthreads();
    float b1 = __builtin_inff(), b2 = __builtin_inff(), b3 = __builtin_inff();
    int i1 = 0, i2 = 0, i3 = 0;
    auto insert = [&](float d, int k) {
        const bool lt1 = d < b1, lt2 = d < b2, lt3 = d < b3;  // strict: ties keep the lower index
        b3 = lt2 ? b2 : (lt3 ? d : b3);
        i3 = lt2 ? i2 : (lt3 ? k : i3);
        b2 = lt1 ? b1 : (lt2 ? d : b2);
        i2 = lt1 ? i1 : (lt2 ? k : i2);
        b1 = lt1 ? d : b1;
        i1 = lt1 ? k : i1;
    };
    const int len = (m + 3) / 4, p0 = c * len, p1 = min(m, p0 + len);
#pragma unroll 4
    for (int p = p0; p < p1; ++p) {
        const float4 kp = sk[p];  // wave-uniform address -> LDS broadcast
        insert(sqdist(ux, uy, uz, kp.x, kp.y, kp.z), p);
    }
    if (c > 0) {
        sd[c - 1][0][ql] = b1; sd[c - 1][1][ql] = b2; sd[c - 1][2][ql] = b3;
        si[c - 1][0][ql] = i1; si[c - 1][1][ql] = i2; si[c - 1][2][ql] = i3;
    }
    __syncthreads();
    if (c != 0 || !active) return;
#pragma unroll
    for (int cc = 0; cc < 3; ++cc)
#pragma unroll
        for (int e = 0; e < 3; ++e) insert(sd[cc][e][ql], si[cc][e][ql]);  // +inf placeholders never insert
    float *od = dist2_all + ((size_t)b * n + q) * 3;
    int *oi = idx_all + ((size_t)b * n + q) * 3;
    if constexpr (WEIGHTS) {
        const float r1 = 1.0f / (__builtin_sqrtf(b1) + 1e-8f), r2 = 1.0f / (__builtin_sqrtf(b2) + 1e-8f),
                    r3 = 1.0f / (__builtin_sqrtf(b3) + 1e-8f);
        const float norm = (r1 + r2) + r3;  // torch.sum over 3 elements adds left to right
        od[0] = r1 / norm; od[1] = r2 / norm; od[2] = r3 / norm;
    } else {
        od[0] = b1; od[1] = b2; od[2] = b3;
    }
    oi[0] = i1; oi[1] = i2; oi[2] = i3;
}

// pn2x_three_nn_interpolate_pm: the search above (weights form) AND the interpolation of the feature rows it selects, in one
// launch.  Feature propagation needs (weight, index) for nothing else (pointnet_utils.py:440-453), so they stay in LDS: wave 0
// finishes the 64 queries of the workgroup, then all four waves blend the three source rows of each query, 16 bytes per lane --
// the same fma order as interp_pm_kernel (interpolate.hip), hence the same floats as the two launches.
__global__ void __launch_bounds__(256)
three_nn_interp_kernel(int n, int m, int c4, const float *__restrict__ unknown_all, const float *__restrict__ known_all,
                       const float *__restrict__ points_all, int ldp, float *__restrict__ out_all, int ldo) {
    extern __shared__ __attribute__((aligned(16))) float4 sk[];
    __shared__ float sd[3][3][64];
    __shared__ int si[3][3][64];
    __shared__ float fw[3][64];
    __shared__ int fi[3][64];
    const int b = blockIdx.y;
    const float *__restrict__ known = known_all + (size_t)b * m * 3;
    const int ql = threadIdx.x & 63, c = threadIdx.x >> 6;  // c is wave-uniform
    const int q = blockIdx.x * 64 + ql;
    const bool active = q < n;
    const float *__restrict__ u = unknown_all + ((size_t)b * n + (active ? q : 0)) * 3;
    const float ux = u[0], uy = u[1], uz = u[2];
    for (int p = threadIdx.x; p < m; p += 256) {
        const float *src = known + (size_t)3 * p;
        sk[p] = make_float4(src[0], src[1], src[2], 0.f);
    }
    __syncthreads();
    float b1 = __builtin_inff(), b2 = __builtin_inff(), b3 = __builtin_inff();
    int i1 = 0, i2 = 0, i3 = 0;
    auto insert = [&](float d, int k) {
        const bool lt1 = d < b1, lt2 = d < b2, lt3 = d < b3;  // strict: ties keep the lower index
        b3 = lt2 ? b2 : (lt3 ? d : b3);
        i3 = lt2 ? i2 : (lt3 ? k : i3);
        b2 = lt1 ? b1 : (lt2 ? d : b2);
        i2 = lt1 ? i1 : (lt2 ? k : i2);
        b1 = lt1 ? d : b1;
        i1 = lt1 ? k : i1;
    };
    const int len = (m + 3) / 4, p0 = c * len, p1 = min(m, p0 + len);
#pragma unroll 4
    for (int p = p0; p < p1; ++p) {
        const float4 kp = sk[p];  // wave-uniform address -> LDS broadcast
        insert(sqdist(ux, uy, uz, kp.x, kp.y, kp.z), p);
    }
    if (c > 0) {
        sd[c - 1][0][ql] = b1; sd[c - 1][1][ql] = b2; sd[c - 1][2][ql] = b3;
        si[c - 1][0][ql] = i1; si[c - 1][1][ql] = i2; si[c - 1][2][ql] = i3;
    }
    __syncthreads();
    if (c == 0) {
#pragma unroll
        for (int cc = 0; cc < 3; ++cc)
#pragma unroll
            for (int e = 0; e < 3; ++e) insert(sd[cc][e][ql], si[cc][e][ql]);  // +inf placeholders never insert
        const float r1 = 1.0f / (__builtin_sqrtf(b1) + 1e-8f), r2 = 1.0f / (__builtin_sqrtf(b2) + 1e-8f),
                    r3 = 1.0f / (__builtin_sqrtf(b3) + 1e-8f);
        const float norm = (r1 + r2) + r3;  // torch.sum over 3 elements adds left to right
        fw[0][ql] = r1 / norm; fw[1][ql] = r2 / norm; fw[2][ql] = r3 / norm;
        fi[0][ql] = i1; fi[1][ql] = i2; fi[2][ql] = i3;
    }
    __syncthreads();
    const int rows = min(64, n - (int)blockIdx.x * 64);
    const float *__restrict__ src = points_all + (size_t)b * m * ldp;
    float *__restrict__ dst = out_all + ((size_t)b * n + (size_t)blockIdx.x * 64) * ldo;
    for (int e = threadIdx.x; e < rows * c4; e += 256) {
        const int r = e / c4, col = e - r * c4;
        const float w0 = fw[0][r], w1 = fw[1][r], w2 = fw[2][r];
        const float4 a0 = *reinterpret_cast<const float4 *>(src + (size_t)fi[0][r] * ldp + 4 * col);
        const float4 a1 = *reinterpret_cast<const float4 *>(src + (size_t)fi[1][r] * ldp + 4 * col);
        const float4 a2 = *reinterpret_cast<const float4 *>(src + (size_t)fi[2][r] * ldp + 4 * col);
        float4 o;
        o.x = __builtin_fmaf(w2, a2.x, __builtin_fmaf(w0, a0.x, w1 * a1.x));
        o.y = __builtin_fmaf(w2, a2.y, __builtin_fmaf(w0, a0.y, w1 * a1.y));
        o.z = __builtin_fmaf(w2, a2.z, __builtin_fmaf(w0, a0.z, w1 * a1.z));
        o.w = __builtin_fmaf(w2, a2.w, __builtin_fmaf(w0, a0.w, w1 * a1.w));
        *reinterpret_cast<float4 *>(dst + (size_t)r * ldo + 4 * col) = o;
    }
}

// From 16384 queries up: below that the blend of a workgroup's 64 rows by its own four waves is a longer chain than the
// many-workgroup interpolation launch it replaces (B = 1 tracking step, same box: 0.3965 ms with, 0.3931 ms without).
bool three_nn_interp_supported(long b, long n, long m, long c, long ldp, long ldo) {
    return b * n >= 16384 && b * n < 256L * 1024 && m >= 16 && m <= kNnTile && c >= 4 && c % 4 == 0 && ldp % 4 == 0 && ldo % 4 == 0;
}

int three_nn_interp_dispatch(int b, int n, int m, int c, const float *unknown, const float *known, const float *points, int ldp,
                             float *out, int ldo, hipStream_t st) {
    if (b == 0 || n == 0) return PN2_OK;
    if (!three_nn_interp_supported(b, n, m, c, ldp, ldo) || (((uintptr_t)points | (uintptr_t)out) % 16) != 0) return PN2_ERANGE;
    dim3 grid((n + 63) / 64, b);
    hipLaunchKernelGGL(three_nn_interp_kernel, grid, dim3(256), (size_t)m * sizeof(float4), st, n, m, c / 4, unknown, known, points, ldp, out, ldo);
    return check_launch();
}

int three_nn_dispatch(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                      int *idx, hipStream_t st, bool weights) {
    if (b == 0 || n == 0) return PN2_OK;
    if ((long)b * n < 256L * 1024 && m >= 16 && m <= kNnTile) {
        dim3 grid((n + 63) / 64, b);
        const size_t lds_split = (size_t)m * sizeof(float4);
        if (weights) hipLaunchKernelGGL((three_nn_split_kernel<true>), grid, dim3(256), lds_split, st, n, m, unknown, known, dist2, idx);
        else hipLaunchKernelGGL((three_nn_split_kernel<false>), grid, dim3(256), lds_split, st, n, m, unknown, known, dist2, idx);
        return check_launch();
    }
    const int tile_cap = m < kNnTile ? m : kNnTile;
    const size_t lds = (size_t)(tile_cap > 0 ? tile_cap : 1) * sizeof(float4);
    // few queries -> single-wave workgroups so the grid still spreads over the CUs
    if ((long)b * n < 256L * 1024) {
        dim3 grid((n + 63) / 64, b);
        if (weights) hipLaunchKernelGGL((three_nn_kernel<64, true>), grid, dim3(64), lds, st, n, m, unknown, known, dist2, idx);
        else hipLaunchKernelGGL((three_nn_kernel<64, false>), grid, dim3(64), lds, st, n, m, unknown, known, dist2, idx);
    } else {
        dim3 grid((n + 255) / 256, b);
        if (weights) hipLaunchKernelGGL((three_nn_kernel<256, true>), grid, dim3(256), lds, st, n, m, unknown, known, dist2, idx);
        else hipLaunchKernelGGL((three_nn_kernel<256, false>), grid, dim3(256), lds, st, n, m, unknown, known, dist2, idx);
    }
    return check_launch();
}

// ------------------------------------------------------------------------------------------
// knn
// ------------------------------------------------------------------------------------------
// kInfKey, sort_keys, knn_wave_body: knn_wave.h
template <int P>
__global__ void __launch_bounds__(256)
knn_wave_kernel(int n, int m, int k, const float *__restrict__ unknown_all, const float *__restrict__ known_all,
                float *__restrict__ dist2_all, int *__restrict__ idx_all, int k2, int *__restrict__ idx2_all) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= n) return;  // wave-uniform
    knn_wave_body<P>(blockIdx.y, q, n, m, k, unknown_all, known_all, dist2_all, idx_all, k2, idx2_all);
}

// m > 64*32 candidates (any m): still one WAVE per query, but the candidates stream past in steps of 64 and the k best keys
// live as a sorted list in LDS.  A step computes 64 keys, ballots those below the current k-th key (wave-uniform, in SGPRs)
// and inserts the few that pass one by one in lane (= index) order: every lane owns the list entries e = lane + 64 s, reads
// entry e and e-1, and the new entry e is old[e] (below the new key), the new key (old[e-1] below it, old[e] not) or
// old[e-1] -- a shift by one done with two LDS reads and one write per entry, no private-memory arrays (the reference keeps
// double[200] + int[200] per thread in local memory; so did this fallback until round 4: 1616 bytes of scratch per lane).
// After the first few steps almost nothing passes the ballot (expected k (1 + ln(m / k)) insertions per query in total), so the
// scan runs at the rate of its distance evaluations.  The result is the k smallest (distance, index) keys in order, i.e. the
// reference's insertion list (strict `<`: an equal distance at a higher index never displaces a lower one).
template <int NS>  // list slots per lane: k <= 64 NS
__global__ void __launch_bounds__(256)
knn_stream_kernel(int n, int m, int k, const float *__restrict__ unknown_all, const float *__restrict__ known_all,
                  float *__restrict__ dist2_all, int *__restrict__ idx_all, int k2, int *__restrict__ idx2_all) {
    __shared__ unsigned long long lists[4][64 * NS];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wv;
    if (q >= n) return;  // wave-uniform; the kernel has no workgroup barrier
    // volatile: the accesses below stay in program order (a wave executes its LDS instructions in order, so "all reads of an
    // insertion before its writes" in the program is what the other lanes of the wave observe)
    volatile unsigned long long *L = lists[wv];
    const float *__restrict__ known = known_all + (size_t)b * m * 3;
    const float *__restrict__ u = unknown_all + ((size_t)b * n + q) * 3;
    const float ux = u[0], uy = u[1], uz = u[2];
#pragma unroll
    for (int s = 0; s < NS; ++s) L[lane + 64 * s] = kInfKey;
    unsigned long long kth = kInfKey;  // wave-uniform: the current k-th best key
    for (int base = 0; base < m; base += 64) {
        const int c = base + lane;
        const bool in = c < m;
        const float *src = known + (size_t)3 * (in ? c : 0);
        const float d = sqdist(ux, uy, uz, src[0], src[1], src[2]);
        // `d < best[k-1]` against the 1e40 sentinel never admits inf / NaN (interpolate_gpu.cu:33,41)
        const unsigned long long key = (in && d < __builtin_inff()) ? (((unsigned long long)(unsigned)f2i(d) << 32) | (unsigned)c) : kInfKey;
        uint64_t pass = __ballot(key < kth);
        while (pass) {
            const int l = __builtin_ctzll(pass);
            pass &= pass - 1;
            const unsigned long long kb = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(key >> 32), l) << 32) |
                                          (unsigned)__builtin_amdgcn_readlane((int)(unsigned)key, l);
            if (!(kb < kth)) continue;  // (wave-uniform) the k-th key has dropped since the ballot
            unsigned long long cur[NS], prev[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int e = lane + 64 * s;
                cur[s] = L[e];
                prev[s] = e > 0 ? L[e - 1] : 0ull;
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) L[lane + 64 * s] = cur[s] < kb ? cur[s] : (prev[s] < kb ? kb : prev[s]);
            kth = L[k - 1];  // broadcast read of the entry just written
            kth = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(kth >> 32)) << 32) |
                  (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)kth);
        }
    }
    float *__restrict__ od = dist2_all ? dist2_all + ((size_t)b * n + q) * k : nullptr;
    int *__restrict__ oi = idx_all + ((size_t)b * n + q) * k;
    int *__restrict__ oi2 = idx2_all ? idx2_all + ((size_t)b * n + q) * k2 : nullptr;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int e = lane + 64 * s;
        if (e < k) {
            const unsigned long long v = L[e];
            if (od) od[e] = i2f((int)(unsigned)(v >> 32));
            oi[e] = (int)(unsigned)v;
            if (oi2 && e < k2) oi2[e] = (int)(unsigned)v;
        }
    }
}

int knn_dispatch(int b, int n, int m, int k, const float *unknown, const float *known, float *dist2,
                 int *idx, hipStream_t st, int k2, int *idx2) {
    if (b == 0 || n == 0) return PN2_OK;
    dim3 grid((n + 3) / 4, b);
#define PN2_KNN_CASE(PP)                                                                                    \
    if (m <= 64 * PP) {                                                                                     \
        hipLaunchKernelGGL(knn_wave_kernel<PP>, grid, dim3(256), 0, st, n, m, k, unknown, known, dist2, idx, k2, idx2); \
        return check_launch();                                                                              \
    }
    PN2_KNN_CASE(1) PN2_KNN_CASE(2) PN2_KNN_CASE(4) PN2_KNN_CASE(8) PN2_KNN_CASE(16) PN2_KNN_CASE(32)
#undef PN2_KNN_CASE
    if (k <= 64) hipLaunchKernelGGL(knn_stream_kernel<1>, grid, dim3(256), 0, st, n, m, k, unknown, known, dist2, idx, k2, idx2);
    else if (k <= 128) hipLaunchKernelGGL(knn_stream_kernel<2>, grid, dim3(256), 0, st, n, m, k, unknown, known, dist2, idx, k2, idx2);
    else hipLaunchKernelGGL(knn_stream_kernel<4>, grid, dim3(256), 0, st, n, m, k, unknown, known, dist2, idx, k2, idx2);
    static_assert(PN2_KNN_MAX_K <= 256, "knn_stream_kernel<4> holds 256 list entries");
    return check_launch();
}

}  // namespace pn2
