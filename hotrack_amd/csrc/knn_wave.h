// knn_wave.h -- the one-wave-per-query k-NN selection for m <= 64 P candidates (register-resident sorted keys), shared by
// nn_search.hip (knn_wave_kernel) and fps.hip (fps_knn_kernel: the keypoints' k-NN lists ride in the launch of the first sampling
// level, whose single workgroup per cloud leaves the other compute units idle).
#pragma once
#include "pn2_common.h"

namespace pn2 {

constexpr unsigned long long kInfKey = 0x7F80000000000000ull;  // (+inf, index 0)

template <int P>
__device__ __forceinline__ void sort_keys(unsigned long long (&key)[P]) {
    // Batcher odd-even merge sort network, fully unrolled (all indices compile-time).
#pragma unroll
    for (int p = 1; p < P; p <<= 1) {
#pragma unroll
        for (int k = p; k >= 1; k >>= 1) {
#pragma unroll
            for (int j = k % p; j + k < P; j += 2 * k) {
#pragma unroll
                for (int i = 0; i < k; ++i) {
                    if (i + j + k < P && (i + j) / (2 * p) == (i + j + k) / (2 * p)) {
                        const unsigned long long a = key[i + j], c = key[i + j + k];
                        const bool sw = c < a;
                        key[i + j] = sw ? c : a;
                        key[i + j + k] = sw ? a : c;
                    }
                }
            }
        }
    }
}

// query q of cloud b, run by one wave
template <int P>
__device__ __forceinline__ void knn_wave_body(const int b, const int q, int n, int m, int k, const float *__restrict__ unknown_all,
                                              const float *__restrict__ known_all, float *__restrict__ dist2_all,
                                              int *__restrict__ idx_all, int k2, int *__restrict__ idx2_all) {
    const int lane = threadIdx.x & 63;
    const float *__restrict__ known = known_all + (size_t)b * m * 3;
    const float *__restrict__ u = unknown_all + ((size_t)b * n + q) * 3;
    const float ux = u[0], uy = u[1], uz = u[2];

    unsigned long long key[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int c = lane * P + j;
        const bool in = c < m;
        const float *src = known + (size_t)3 * (in ? c : 0);
        const float d = sqdist(ux, uy, uz, src[0], src[1], src[2]);
        // `d < best[j]` against the 1e40 sentinel (interpolate_gpu.cu:33,41) never admits inf/NaN
        const bool ok = in && (d < __builtin_inff());
        key[j] = ok ? (((unsigned long long)(unsigned)f2i(d) << 32) | (unsigned)c) : kInfKey;
    }
    sort_keys<P>(key);

    // dist2_all may be NULL (indices only); idx2_all, if given, receives the first k2 <= k indices as a second,
    // contiguous (b, n, k2) list -- the k-NN list is sorted, so a smaller neighbourhood is its prefix
    float *__restrict__ od = dist2_all ? dist2_all + ((size_t)b * n + q) * k : nullptr;
    int *__restrict__ oi = idx_all + ((size_t)b * n + q) * k;
    int *__restrict__ oi2 = idx2_all ? idx2_all + ((size_t)b * n + q) * k2 : nullptr;
    for (int r = 0; r < k; ++r) {
        const unsigned hd = (unsigned)(key[0] >> 32);
        const unsigned mn = wave_min_u32(hd);
        const uint64_t tie = __ballot(hd == mn);
        const int wl = __builtin_ctzll(tie);  // lanes own ascending index ranges: lowest lane = lowest index
        if (lane == wl) {
            if (od) od[r] = i2f((int)mn);
            oi[r] = (int)(unsigned)key[0];
            if (oi2 && r < k2) oi2[r] = (int)(unsigned)key[0];
#pragma unroll
            for (int j = 0; j + 1 < P; ++j) key[j] = key[j + 1];
            key[P - 1] = kInfKey;
        }
    }
}

}  // namespace pn2
