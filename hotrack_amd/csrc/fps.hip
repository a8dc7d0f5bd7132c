// fps.hip -- furthest point sampling for gfx950 (MI355X).
//
// Replaces furthest_point_sampling_kernel / _launcher (reference sampling_gpu.cu:94-253).
// Design (not a translation of the CUDA kernel):
//   * one workgroup per cloud (the M-step dependency chain is inherent);
//   * every point and its running min-distance live in VGPRs for the whole kernel --
//     no `temp` traffic to HBM, xyz is read exactly once;
//   * points are laid out over (thread, slot) in the ORDER OF THE REFERENCE'S TIE KEY
//     (bitrev(k mod bs), k div bs), so "first maximum in my own order" is exactly the
//     reference's shared-memory tree winner (sampling_gpu.cu:86-91,143-203) and the
//     arg-max needs no index in the reduction: wave max via 6 DPP steps on the fp32 bit
//     pattern, a ballot + s_ff1 picks the first lane holding the max;
//   * the winning lane of each wave publishes (dist, k[, x, y, z]) to LDS; ONE barrier per
//     iteration (double-buffered by iteration parity; the reference needs 11);
//   * all waves redundantly combine the <=16 wave candidates with DPP inside one row and
//     pull the winner's coordinates into SGPRs with v_readlane.
#include <stdlib.h>
#include "pn2_common.h"
#include "knn_wave.h"
#include "fps_tie.h"

namespace pn2 {

enum { kCentEntry = 0, kCentLds = 1, kCentGlobal = 2 };

// skip_flags (pn2x_furthest_point_sampling_prefix, pn2_ext.h): `nflags` ints per cloud; if given and all of a cloud's
// are zero, the sample is known to be 0..m-1 and the workgroup writes that and returns.
// RAD: also record radii (pn2x_furthest_point_sampling_radii); a template flag so the plain kernel's loop is untouched.
// The sampling of cloud `cloud` by one workgroup of T threads (a whole fps_kernel workgroup, or one of the first b workgroups of
// fps_knn_kernel below).
template <int T, int P, int CENT, bool RAD>
__device__ __forceinline__ void fps_body(const int cloud, int n, int m, int bs, int lg, int Q, const float *__restrict__ xyz_all,
                                         int *__restrict__ idx_all, const int *__restrict__ skip_flags, int nflags,
                                         float *__restrict__ radii_all) {
    constexpr int W = T / kWave;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // entries: [parity][field][wave]   field: 0 dist, 1 k, 2 x, 3 y, 4 z
    float *ent = smem;
    float *lxyz = smem + 2 * 5 * 16;

    const float *__restrict__ xyz = xyz_all + (size_t)cloud * n * 3;
    int *__restrict__ idx = idx_all + (size_t)cloud * m;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    if (skip_flags) {  // wave-uniform
        int any = 0;
        for (int f = 0; f < nflags; ++f) any |= skip_flags[(size_t)cloud * nflags + f];
        if (!any) {
            for (int i = tid; i < m; i += T) idx[i] = i;
            return;
        }
    }

    float px[P], py[P], pz[P], pt[P];
    int pk[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int s = tid * P + j;
        const int bslot = s / Q;
        const int q = s - bslot * Q;
        const unsigned rev = lg ? (__builtin_bitreverse32((unsigned)bslot) >> (32 - lg)) : 0u;
        const int k = (int)rev + q * bs;
        const bool valid = (bslot < bs) && (k < n);
        pk[j] = valid ? k : 0;
        px[j] = valid ? xyz[3 * k + 0] : 0.f;
        py[j] = valid ? xyz[3 * k + 1] : 0.f;
        pz[j] = valid ? xyz[3 * k + 2] : 0.f;
        pt[j] = valid ? 1e10f : -1.0f;  // padded slots can never win (all real dists >= 0)
    }
    if constexpr (CENT == kCentLds) {
        for (int i = tid; i < 3 * n; i += T) lxyz[i] = xyz[i];
    }
    float cx = xyz[0], cy = xyz[1], cz = xyz[2];
    if (tid == 0) idx[0] = 0;
    if constexpr (CENT == kCentLds) __syncthreads();

    for (int it = 1; it < m; ++it) {
        float best;
        int bestk;
        if constexpr (P == 1) {
            const float d = sqdist(px[0], py[0], pz[0], cx, cy, cz);
            pt[0] = fmin_raw(d, pt[0]);
            best = pt[0];
            bestk = pk[0];
        } else {
            best = -1.0f;
            bestk = 0;
            if constexpr (P <= 8 && !(W == 1 && P == 4)) {  // (one wave x 4 points: 28.3 us scalar, 29.1 packed)
                // pairs through the packed-fp32 ops (half the VALU issue of the distance arithmetic; measured 90.7 -> 86.3 us
                // at N=1024; at P = 16 the register pairing costs more than it saves: 1345 -> 1474 us at N=8192)
#pragma unroll
                for (int j = 0; j < P; j += 2) {
                    const pn2_f32x2 d2 = sqdist2((pn2_f32x2){px[j], px[j + 1]}, (pn2_f32x2){py[j], py[j + 1]},
                                                 (pn2_f32x2){pz[j], pz[j + 1]}, cx, cy, cz);
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const float tt = fmin_raw(d2[u], pt[j + u]);
                        pt[j + u] = tt;
                        const bool gt = tt > best;  // strict: first (lowest tie-rank) maximum wins
                        bestk = gt ? pk[j + u] : bestk;
                        best = gt ? tt : best;
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < P; ++j) {
                    const float d = sqdist(px[j], py[j], pz[j], cx, cy, cz);
                    const float tt = fmin_raw(d, pt[j]);
                    pt[j] = tt;
                    const bool gt = tt > best;  // strict: first (lowest tie-rank) maximum wins
                    bestk = gt ? pk[j] : bestk;
                    best = gt ? tt : best;
                }
            }
        }
        // fp32 >= 0 (or exactly -1.0f) orders like its bit pattern as a signed int.
        const int bi = f2i(best);
#if defined(PN2_FPS_PROBE) && PN2_FPS_PROBE == 2   /* timing probe: no DPP reduction */
        const int wmax = __builtin_amdgcn_readfirstlane(bi);
#else
        const int wmax = wave_max_i32(bi);
#endif
#if defined(PN2_FPS_PROBE) && PN2_FPS_PROBE == 4   /* timing probe: no ballot / ctz */
        const int wl = it & 63;
#else
        const uint64_t tie = __ballot(bi == wmax);
        const int wl = __builtin_ctzll(tie);  // first lane == lowest tie rank in this wave
#endif

        int kstar;
        int gbits = wmax;  // bit pattern of the global maximum of this iteration
        if constexpr (W == 1) {
            kstar = __builtin_amdgcn_readlane(bestk, wl);
            if constexpr (CENT == kCentEntry) {
                cx = i2f(__builtin_amdgcn_readlane(f2i(px[0]), wl));
                cy = i2f(__builtin_amdgcn_readlane(f2i(py[0]), wl));
                cz = i2f(__builtin_amdgcn_readlane(f2i(pz[0]), wl));
            }
        } else {
            float *e = ent + (it & 1) * (5 * 16);
            if (lane == wl) {
                e[0 * 16 + w] = best;
                e[1 * 16 + w] = i2f(bestk);
                if constexpr (CENT == kCentEntry) {
                    e[2 * 16 + w] = px[0];
                    e[3 * 16 + w] = py[0];
                    e[4 * 16 + w] = pz[0];
                }
            }
            __syncthreads();
            const int sl = lane & (W - 1);
            const int ev = f2i(e[0 * 16 + sl]);
            const int ek = f2i(e[1 * 16 + sl]);
            int ex = 0, ey = 0, ez = 0;
            if constexpr (CENT == kCentEntry) {
                ex = f2i(e[2 * 16 + sl]);
                ey = f2i(e[3 * 16 + sl]);
                ez = f2i(e[4 * 16 + sl]);
            }
            const int gmax = __builtin_amdgcn_readfirstlane(row_group_max_i32<W>(ev));
            const uint64_t tie2 = __ballot(ev == gmax);
            const int wl2 = __builtin_ctzll(tie2);  // lanes 0..W-1 hold waves 0..W-1: lowest wave wins
            kstar = __builtin_amdgcn_readlane(ek, wl2);
            gbits = gmax;
            if constexpr (CENT == kCentEntry) {
                cx = i2f(__builtin_amdgcn_readlane(ex, wl2));
                cy = i2f(__builtin_amdgcn_readlane(ey, wl2));
                cz = i2f(__builtin_amdgcn_readlane(ez, wl2));
            }
        }
        if (tid == 0) {
            idx[it] = kstar;
            // the winning (maximum) running distance of this pick -- input of the post-hoc tie check
            if constexpr (RAD) radii_all[(size_t)cloud * m + it] = i2f(gbits);
        }
#if defined(PN2_FPS_PROBE) && PN2_FPS_PROBE == 1   /* timing probe: no dependent centroid read */
        if constexpr (CENT == kCentLds) {
            cx += 1e-7f * (float)(kstar & 1);
        } else
#endif
        if constexpr (CENT == kCentLds) {
            cx = lxyz[3 * kstar + 0];
            cy = lxyz[3 * kstar + 1];
            cz = lxyz[3 * kstar + 2];
        } else if constexpr (CENT == kCentGlobal) {
            cx = xyz[3 * kstar + 0];
            cy = xyz[3 * kstar + 1];
            cz = xyz[3 * kstar + 2];
        }
    }
}

template <int T, int P, int CENT, bool RAD>
__global__ void __launch_bounds__(T)
fps_kernel(int n, int m, int bs, int lg, int Q, const float *__restrict__ xyz_all, int *__restrict__ idx_all,
           const int *__restrict__ skip_flags, int nflags, float *__restrict__ radii_all) {
    fps_body<T, P, CENT, RAD>((int)blockIdx.x, n, m, bs, lg, Q, xyz_all, idx_all, skip_flags, nflags, radii_all);
}

// First sampling level of the inference path (1024 slots: T = 256, P = 4, centroids from LDS, radii recorded) AND the k-NN lists
// of `nq` query points per cloud (the keypoints) among the same n points, in one launch: workgroups [0, b) sample, the rest run
// one query per wave (knn_wave.h).  The sampling is a 0.35 us-per-pick dependency chain on ONE compute unit per cloud; the
// k-NN search (21 us as its own launch at B = 1, on the critical path of the tracking loop) only needs the same coordinates.
template <int KP>
__global__ void __launch_bounds__(256)
fps_knn_kernel(int b, int n, int m, int bs, int lg, int Q, const float *__restrict__ xyz_all, int *__restrict__ idx_all,
               float *__restrict__ radii_all, int nq, int k, const float *__restrict__ query_all, int *__restrict__ kidx_all, int k2,
               int *__restrict__ kidx2_all) {
    if ((int)blockIdx.x < b) {  // workgroup-uniform
        fps_body<256, 4, kCentLds, true>((int)blockIdx.x, n, m, bs, lg, Q, xyz_all, idx_all, nullptr, 0, radii_all);
        return;
    }
    const int e = (int)blockIdx.x - b, qb = (nq + 3) / 4;
    const int cloud = e / qb, q = (e - cloud * qb) * 4 + (int)(threadIdx.x >> 6);
    if (q >= nq) return;  // wave-uniform; this role has no workgroup barrier
    knn_wave_body<KP>(cloud, q, nq, n, k, query_all, xyz_all, nullptr, kidx_all, k2, kidx2_all);
}

// Large-cloud fallback (n > 65536): running distances in the caller's `temp` (HBM), the
// reference's thread structure (bs = 1024 threads, strided ownership) with the tie key
// carried explicitly through a 64-bit LDS tree.  Correct for any n; not tuned.
__global__ void __launch_bounds__(1024)
fps_large_kernel(int n, int m, const float *__restrict__ xyz_all, float *__restrict__ temp_all,
                 int *__restrict__ idx_all) {
    __shared__ unsigned long long keys[1024];
    const float *__restrict__ xyz = xyz_all + (size_t)blockIdx.x * n * 3;
    float *__restrict__ temp = temp_all + (size_t)blockIdx.x * n;
    int *__restrict__ idx = idx_all + (size_t)blockIdx.x * m;
    const int tid = threadIdx.x;
    // rank of this thread in the reference tree = bitrev10(tid); smaller wins ties.
    const unsigned trank = __builtin_bitreverse32((unsigned)tid) >> 22;
    int old = 0;
    if (tid == 0) idx[0] = 0;
    for (int it = 1; it < m; ++it) {
        const float cx = xyz[3 * old], cy = xyz[3 * old + 1], cz = xyz[3 * old + 2];
        float best = -1.f;
        int besti = 0;
        for (int k = tid; k < n; k += 1024) {
            const float d = sqdist(xyz[3 * k], xyz[3 * k + 1], xyz[3 * k + 2], cx, cy, cz);
            const float tt = fmin_raw(d, temp[k]);
            temp[k] = tt;
            besti = tt > best ? k : besti;
            best = tt > best ? tt : best;
        }
        // key: dist bits (>=0) high; inverted (thread rank, k div 1024) low -> max = winner
        const unsigned lo = ~((trank << 21) | (unsigned)(besti >> 10));
        keys[tid] = ((unsigned long long)(unsigned)f2i(best) << 32) | lo;
        __syncthreads();
        for (int s = 512; s >= 1; s >>= 1) {
            if (tid < s) {
                const unsigned long long a = keys[tid], b2 = keys[tid + s];
                keys[tid] = a > b2 ? a : b2;
            }
            __syncthreads();
        }
        const unsigned wlo = ~(unsigned)keys[0];
        const unsigned wt = __builtin_bitreverse32((wlo >> 21) & 1023u) >> 22;
        old = (int)(((wlo & 0x1FFFFFu) << 10) | wt);
        if (tid == 0) idx[it] = old;
        __syncthreads();
    }
}

// Clouds of 16385 .. 65536 points: one workgroup of 1024 threads per cloud, thread t owns the points t + 1024 q like the
// reference (sampling_gpu.cu:143-167).  Their RUNNING DISTANCES live in QMAX registers per lane for the whole kernel -- no `temp`
// traffic at all, against 1 MB read + 256 KB written per pick by fps_large_kernel below -- and the coordinates sit where there is
// room, nearest first: slots q < QR in registers (3 QR per lane), the next QL slots in LDS (structure of arrays, 12 KB per slot:
// conflict-free ds_read_b32), the rest is re-read every pick (L2-resident, coalesced 12-byte records through a buffer descriptor
// whose bounds check returns zeros for the slots beyond n).  At 32768 points 28 of 32 slots are resident, at 65536 points 13 of 64 (LDS only).
// The arg-max follows the reference's tie order without a tree: wave maximum of the distance bits (6 DPP steps), then the smallest
// (bitrev10(thread), q) among the lanes that hold it (6 more), one LDS entry per wave, ONE barrier per pick (entries
// double-buffered by parity), every wave combines the 16 entries on its own.
template <int QMAX, int QR, int QL>
__global__ void __launch_bounds__(1024)
fps_stream_kernel(int n, int m, const float *__restrict__ xyz_all, int *__restrict__ idx_all) {
    static_assert(QR + QL <= QMAX, "resident slots");
    __shared__ unsigned ent[2][2][16];  // [parity][distance bits | tie key][wave]
    extern __shared__ __attribute__((aligned(16))) float lco[];  // [QL][3][1024]
    const float *__restrict__ xyz = xyz_all + (size_t)blockIdx.x * n * 3;
    int *__restrict__ idx = idx_all + (size_t)blockIdx.x * m;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const unsigned trank = __builtin_bitreverse32((unsigned)tid) >> 22;  // rank of this thread in the reference's tree: smaller wins ties
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xyz), 0, 12 * n, 0x00020000);
    typedef float f32x3v __attribute__((ext_vector_type(3)));
    // (whole-vector cast: a bit cast of a vector ELEMENT reads element 0 with this hipcc)
    auto record = [&](int q) { return __builtin_bit_cast(f32x3v, __builtin_amdgcn_raw_buffer_load_b96(rx, 12 * tid + 12 * 1024 * q, 0, 0)); };
    float pt[QMAX];
#pragma unroll
    for (int q = 0; q < QMAX; ++q) pt[q] = tid + 1024 * q < n ? 1e10f : -1.0f;  // slots beyond n can never win (real distances >= 0)
    float rxs[QR > 0 ? QR : 1], rys[QR > 0 ? QR : 1], rzs[QR > 0 ? QR : 1];
#pragma unroll
    for (int q = 0; q < QR; ++q) {
        const f32x3v p = record(q);
        rxs[q] = p.x; rys[q] = p.y; rzs[q] = p.z;
    }
#pragma unroll
    for (int q = 0; q < QL; ++q) {  // (read back by the thread that wrote them: no barrier needed)
        const f32x3v p = record(QR + q);
        lco[(3 * q + 0) * 1024 + tid] = p.x;
        lco[(3 * q + 1) * 1024 + tid] = p.y;
        lco[(3 * q + 2) * 1024 + tid] = p.z;
    }
    float cx = xyz[0], cy = xyz[1], cz = xyz[2];
    if (tid == 0) idx[0] = 0;
    for (int it = 1; it < m; ++it) {
        float best = -1.0f;
        int bestq = 0;
        auto visit = [&](int q, float x, float y, float z) {
            const float d = sqdist(x, y, z, cx, cy, cz);
            const float tt = fmin_raw(d, pt[q]);
            pt[q] = tt;
            const bool gt = tt > best;  // strict: the first maximum in increasing k wins inside a thread
            bestq = gt ? q : bestq;
            best = gt ? tt : best;
        };
#pragma unroll
        for (int q = 0; q < QR; ++q) visit(q, rxs[q], rys[q], rzs[q]);
#pragma unroll
        for (int q = 0; q < QL; ++q) visit(QR + q, lco[(3 * q + 0) * 1024 + tid], lco[(3 * q + 1) * 1024 + tid], lco[(3 * q + 2) * 1024 + tid]);
#pragma unroll
        for (int q = QR + QL; q < QMAX; ++q) {
            const f32x3v p = record(q);
            visit(q, p.x, p.y, p.z);
        }
        const int bi = f2i(best);  // fp32 >= 0 (or exactly -1.0f) orders like its bit pattern as a signed int
        const int wmax = wave_max_i32(bi);
        const unsigned wkey = wave_min_u32(bi == wmax ? ((trank << 21) | (unsigned)bestq) : 0xffffffffu);
        if (lane == 0) {
            ent[it & 1][0][w] = (unsigned)wmax;
            ent[it & 1][1][w] = wkey;
        }
        __syncthreads();
        const int ev = (int)ent[it & 1][0][lane & 15];
        const unsigned ek = ent[it & 1][1][lane & 15];
        const int gmax = __builtin_amdgcn_readfirstlane(row_group_max_i32<16>(ev));
        const unsigned gkey = (unsigned)__builtin_amdgcn_readfirstlane((int)row_group_min_u32<16>(ev == gmax ? ek : 0xffffffffu));
        const unsigned wt = __builtin_bitreverse32(gkey >> 21) >> 22;
        const int old = (int)(((gkey & 0x1FFFFFu) << 10) | wt);
        if (tid == 0) idx[it] = old;
        cx = xyz[3 * old];
        cy = xyz[3 * old + 1];
        cz = xyz[3 * old + 2];
    }
}

template <int QMAX, int QR, int QL>
static int launch_fps_stream(int b, int n, int m, const float *xyz, int *idx, hipStream_t st) {
    const size_t lds = (size_t)QL * 3 * 1024 * sizeof(float);
    static PerDeviceOnce raised;
    static bool refused[64] = {};  // devices whose runtime would not raise the dynamic-LDS cap (the result was ignored)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (lds > 48 * 1024 && raised.first_use()) {
        if (hipFuncSetAttribute((const void *)fps_stream_kernel<QMAX, QR, QL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            refused[dev] = true;
        }
    }
    if (lds > 48 * 1024 && refused[dev]) return PN2_ERANGE;  // the caller takes the HBM-temp kernel (needs `temp`)
    hipLaunchKernelGGL((fps_stream_kernel<QMAX, QR, QL>), dim3(b), dim3(1024), lds, st, n, m, xyz, idx);
    return check_launch();
}

template <int T, int P, bool RAD>
static int launch_fps(int b, int n, int m, int bs, int lg, int Q, const float *xyz, int *idx, hipStream_t st, const int *skip_flags,
                      int nflags, float *radii) {
    const size_t ent_bytes = 2 * 5 * 16 * sizeof(float);
    if constexpr (P == 1) {
        hipLaunchKernelGGL((fps_kernel<T, 1, kCentEntry, RAD>), dim3(b), dim3(T), ent_bytes, st, n, m, bs, lg, Q, xyz, idx, skip_flags, nflags, radii);
    } else {
        const size_t need = ent_bytes + (size_t)n * 3 * sizeof(float);
        if (need <= 128 * 1024) {
            auto kfn = fps_kernel<T, P, kCentLds, RAD>;
            static PerDeviceOnce raised;  // raise the dynamic-LDS cap once per instantiation and device, not per launch
            if (need > 64 * 1024 && raised.first_use())
                (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            hipLaunchKernelGGL(kfn, dim3(b), dim3(T), need, st, n, m, bs, lg, Q, xyz, idx, skip_flags, nflags, radii);
        } else {
            hipLaunchKernelGGL((fps_kernel<T, P, kCentGlobal, RAD>), dim3(b), dim3(T), ent_bytes, st, n, m, bs, lg, Q, xyz, idx, skip_flags, nflags, radii);
        }
    }
    return check_launch();
}

// the post-hoc tie check (pn2x_fps_prefix_ties): fps_tie.h
__global__ void __launch_bounds__(256)
fps_tie_check_kernel(int n, int m, int m1, const float *__restrict__ xyz_all, const int *__restrict__ idx_all,
                     const float *__restrict__ radii_all, int *__restrict__ flags) {
    fps_tie_body((int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, n, m, m1, xyz_all, idx_all, radii_all, flags);
}

int fps_tie_check(int b, int n, int m, int m1, const float *xyz, const int *idx, const float *radii, int *flags, hipStream_t st) {
    if (m > kTieMaxM) return PN2_ERANGE;
    hipLaunchKernelGGL(fps_tie_check_kernel, dim3((n + kTiePts - 1) / kTiePts, b), dim3(256), 0, st, n, m, m1, xyz, idx, radii, flags);
    return check_launch();
}

static int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

int fps_dispatch(int b, int n, int m, const float *xyz, float *temp, int *idx, hipStream_t st, int force_threads,
                 const int *skip_flags, int nflags, float *radii) {
    // reference block size: cuda_utils.h:10-14
    int bs = 1;
    while (bs * 2 <= n && bs * 2 <= 1024) bs *= 2;
    int lg = 0;
    while ((1 << lg) < bs) ++lg;
    const int Q = (n + bs - 1) / bs;
    const int slots = bs * Q;
    if (slots > 1024 * 16) {
        if (skip_flags || radii) return PN2_ERANGE;  // the shortcut covers the register-resident kernels only
        static const bool no_stream = getenv("PN2_FPS_NO_STREAM") != nullptr;  // (A/B against the HBM-temp kernel: tests, probes)
        if (Q <= 64 && !no_stream) {  // running distances in registers, coordinates streamed from L2: no scratch buffer
            // (64 distance registers leave no room for coordinates: hipcc spills from QR = 4)
            const int rc = Q <= 32 ? launch_fps_stream<32, 16, 12>(b, n, m, xyz, idx, st) : launch_fps_stream<64, 0, 13>(b, n, m, xyz, idx, st);
            if (rc != PN2_ERANGE) return rc;  // PN2_ERANGE: no 144-156 KiB of dynamic LDS here -> the HBM-temp kernel below
        }
        if (!temp) return PN2_ESCRATCH;
        hipLaunchKernelGGL(fps_large_kernel, dim3(b), dim3(1024), 0, st, n, m, xyz, temp, idx);
        return check_launch();
    }
    // One lane per point (T = slots) minimises VALU work per iteration, but the iteration is a latency
    // chain (DPP reduce -> LDS exchange -> barrier), and 4 waves per SIMD serialise it: measured on MI355X
    // (B=64): N=1024 T=1024 107 us, T=256 89 us; N=256 T=256 40 us, T=64 29 us.  So: 4 points per lane.
    // Measured per (slots) on MI355X, any batch <= 256 clouds (scripts/probes/fps_threads.py):
    //   512 slots: T=64 39.5 us (one wave, no barrier at all), T=128 45.9;   1024: T=256 90.7, T=128 107, T=512 ~90;
    //   8192: T=512 1353 us, T=1024 1413.
    int T = slots <= 512 ? 64 : (slots <= 1024 ? 256 : 512);
    if (slots > 512 * 16) T = 1024;
    if (force_threads == 64 || force_threads == 128 || force_threads == 256 || force_threads == 512 || force_threads == 1024) {
        if (force_threads * 16 >= slots) T = force_threads;
    }
    const int P = next_pow2((slots + T - 1) / T);
#define PN2_FPS_CASE(TT, PP) \
    if (T == TT && P == PP) return radii ? launch_fps<TT, PP, true>(b, n, m, bs, lg, Q, xyz, idx, st, skip_flags, nflags, radii) \
                                         : launch_fps<TT, PP, false>(b, n, m, bs, lg, Q, xyz, idx, st, skip_flags, nflags, radii);
#define PN2_FPS_ROW(TT) PN2_FPS_CASE(TT, 1) PN2_FPS_CASE(TT, 2) PN2_FPS_CASE(TT, 4) PN2_FPS_CASE(TT, 8) PN2_FPS_CASE(TT, 16)
    PN2_FPS_ROW(64) PN2_FPS_ROW(128) PN2_FPS_ROW(256) PN2_FPS_ROW(512) PN2_FPS_ROW(1024)
#undef PN2_FPS_ROW
#undef PN2_FPS_CASE
    return PN2_ERANGE;
}

// pn2x_fps_radii_knn: the co-launch above, for the one geometry it is built for (everything else: PN2_ERANGE, the caller runs
// the two launches)
static int fps_block_size(int n) {  // reference block size: cuda_utils.h:10-14 (as fps_dispatch)
    int bs = 1;
    while (bs * 2 <= n && bs * 2 <= 1024) bs *= 2;
    return bs;
}

bool fps_knn_supported(int n, int nq, int k) {
    if (n < 1) return false;
    const int bs = fps_block_size(n);
    const int Q = (n + bs - 1) / bs;
    const int slots = bs * Q;
    return slots > 512 && slots <= 1024 && n <= 1024 && nq >= 1 && k >= 1 && k <= n && k <= PN2_KNN_MAX_K &&
           2 * 5 * 16 * sizeof(float) + (size_t)n * 12 <= 64 * 1024;
}

int fps_knn_dispatch(int b, int n, int m, const float *xyz, int *idx, float *radii, int nq, int k, int k2, const float *query,
                     int *kidx, int *kidx2, hipStream_t st) {
    if (!fps_knn_supported(n, nq, k)) return PN2_ERANGE;
    const int bs = fps_block_size(n);
    int lg = 0;
    while ((1 << lg) < bs) ++lg;
    const int Q = (n + bs - 1) / bs;
    const size_t need = 2 * 5 * 16 * sizeof(float) + (size_t)n * 3 * sizeof(float);
    const long grid = (long)b + (long)b * ((nq + 3) / 4);
    if (grid > 2147483647L) return PN2_ERANGE;
    hipLaunchKernelGGL(fps_knn_kernel<16>, dim3((unsigned)grid), dim3(256), need, st, b, n, m, bs, lg, Q, xyz, idx, radii, nq, k, query,
                       kidx, k2, kidx2);
    return check_launch();
}

}  // namespace pn2
