// sdf.hip -- the particle optimisers' signed-distance-volume lookups (include/pn2_sdf.h).
//
// Gather-bound work: every (particle, point) pair costs one 3x3 transform, a few dozen fp32 ops and 8 (trilinear)
// or 1 (nearest) reads of a 2-byte voxel.  The volume (201^3 fp16 = 16 MB) and the cloud (N x 12 B) are shared by
// all particles and stay in L2 / Infinity Cache; nothing of size P x N is ever written.  One workgroup per
// particle (P = 2048..5120 >> 256 CUs), 256 threads striding over the points, one block reduction per particle.
//
// Arithmetic follows the reference's torch expressions operation for operation (fp32, true division, same
// association; compiled with -ffp-contract=off), so Distance() is bit-identical to the reference.
#include <hip/hip_fp16.h>

#include "pn2_common.h"
#include "../../include/pn2_sdf.h"

namespace pn2 {

struct SdfVol {
    const void *p;
    int res;
    float bbox_min, stride, lo, hi;
};

template <int FMT>
__device__ __forceinline__ float vol_at(const void *v, int i) {
    if constexpr (FMT & 1) return __half2float(reinterpret_cast<const __half *>(v)[i]);
    else return reinterpret_cast<const float *>(v)[i];
}

__device__ __forceinline__ float clampf(float v, float lo, float hi) {  // torch.clamp: min(max(v, lo), hi)
    v = v < lo ? lo : v;
    return v > hi ? hi : v;
}

// One point of gf_optimize_obj.Distance (optimization_obj.py:184-228), in three steps so that callers can issue
// the gathers of several points before blending any of them (the kernel is gather-latency bound).
struct TriPoint {
    int i000;
    float x, y, z;  // fractional parts
};

__device__ __forceinline__ TriPoint tri_setup(const SdfVol &A, float vx, float vy, float vz) {
    const float top = (float)(A.res - 1);
    float x = clampf((vx - A.bbox_min) / A.stride, 0.0f, top);
    float y = clampf((vy - A.bbox_min) / A.stride, 0.0f, top);
    float z = clampf((vz - A.bbox_min) / A.stride, 0.0f, top);
    const int xi = (int)x, yi = (int)y, zi = (int)z;  // x >= 0: trunc == floor
    TriPoint t;
    t.x = x - (float)xi;
    t.y = y - (float)yi;
    t.z = z - (float)zi;
    t.i000 = (xi * A.res + yi) * A.res + zi;
    return t;
}

#ifndef SDF_PAIR
#define SDF_PAIR 1
#endif
#ifndef SDF_U
#define SDF_U 1  /* measured: 1, 2, 4 within 1% once SDF_PAIR is on (profiles/r01_sdf_sweep.txt) */
#endif

// Two z-adjacent voxels (i, i+1) with ONE load: the kernel is bound by the number of divergent gather lanes the
// texture-address path has to process, and z-neighbours are adjacent in memory (2-byte aligned only: memcpy lets
// the compiler pick an unaligned dword load, which gfx950 global memory supports).
template <int FMT>
__device__ __forceinline__ void vol_pair(const void *v, int i, float &a, float &b) {
    if constexpr (FMT & 1) {
        unsigned w;
        __builtin_memcpy(&w, reinterpret_cast<const __half *>(v) + i, 4);
        a = __half2float(__ushort_as_half((unsigned short)(w & 0xffffu)));
        b = __half2float(__ushort_as_half((unsigned short)(w >> 16)));
    } else {
        float2 w;
        __builtin_memcpy(&w, reinterpret_cast<const float *>(v) + i, 8);
        a = w.x;
        b = w.y;
    }
}

template <int FMT>
__device__ __forceinline__ void tri_gather(const SdfVol &A, const TriPoint &t, float *d) {
    const int R = A.res, RR = R * R, last = RR * R - 1, i = t.i000;
    if constexpr (FMT >= 2) {
        // corner layout (pn2s_build_corner_volume): cell i holds the eight values the reference would fetch for
        // i000 == i, already clamped its way -> ONE 16-byte (fp16) or two 16-byte (fp32) loads per point
        if constexpr (FMT == 3) {
            const uint4 w = reinterpret_cast<const uint4 *>(A.p)[i];
            const unsigned u[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                d[2 * k] = __half2float(__ushort_as_half((unsigned short)(u[k] & 0xffffu)));
                d[2 * k + 1] = __half2float(__ushort_as_half((unsigned short)(u[k] >> 16)));
            }
        } else {
            const float4 a = reinterpret_cast<const float4 *>(A.p)[2 * (size_t)i], b = reinterpret_cast<const float4 *>(A.p)[2 * (size_t)i + 1];
            d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w;
            d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
        }
        return;
    }
    if (SDF_PAIR && i + 1 + R + RR <= last) {  // everywhere except the volume's last corner
        vol_pair<FMT>(A.p, i, d[0], d[1]);
        vol_pair<FMT>(A.p, i + R, d[2], d[3]);
        vol_pair<FMT>(A.p, i + RR, d[4], d[5]);
        vol_pair<FMT>(A.p, i + R + RR, d[6], d[7]);
        return;
    }
    // the reference clamps each corner index to [0, res^3 - 1] (:215-222); only the upper clamp can bind
    d[0] = vol_at<FMT>(A.p, i);                          d[1] = vol_at<FMT>(A.p, min(i + 1, last));
    d[2] = vol_at<FMT>(A.p, min(i + R, last));           d[3] = vol_at<FMT>(A.p, min(i + 1 + R, last));
    d[4] = vol_at<FMT>(A.p, min(i + RR, last));          d[5] = vol_at<FMT>(A.p, min(i + 1 + RR, last));
    d[6] = vol_at<FMT>(A.p, min(i + R + RR, last));      d[7] = vol_at<FMT>(A.p, min(i + 1 + R + RR, last));
}

__device__ __forceinline__ float tri_blend(const SdfVol &A, const TriPoint &t, const float *d) {
    const float mx = 1.0f - t.x, my = 1.0f - t.y, mz = 1.0f - t.z;
    const float lo = ((d[0] * mz + d[1] * t.z) * my + (d[2] * mz + d[3] * t.z) * t.y) * mx;  // :223-226, same association
    const float hi = ((d[4] * mz + d[5] * t.z) * my + (d[6] * mz + d[7] * t.z) * t.y) * t.x;
    return clampf(lo + hi, A.lo, A.hi);
}

template <int FMT>
__device__ __forceinline__ float trilinear(const SdfVol &A, float vx, float vy, float vz) {
    const TriPoint t = tri_setup(A, vx, vy, vz);
    float d[8];
    tri_gather<FMT>(A, t, d);
    return tri_blend(A, t, d);
}

// (p - t) @ R with the fixed chain o_j = fma(q2, R2j, fma(q1, R1j, q0*R0j)) (same chain in oracle/sdf_oracle.c).
__device__ __forceinline__ void to_object_frame(float px, float py, float pz, const float *t, const float *R, float &ox,
                                                float &oy, float &oz) {
    const float q0 = px - t[0], q1 = py - t[1], q2 = pz - t[2];
    ox = fmaf(q2, R[6], fmaf(q1, R[3], q0 * R[0]));
    oy = fmaf(q2, R[7], fmaf(q1, R[4], q0 * R[1]));
    oz = fmaf(q2, R[8], fmaf(q1, R[5], q0 * R[2]));
}

__device__ __forceinline__ float block_sum_256(float v, float *sm) {  // sm: >= 4 floats of LDS; all threads get the sum
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    const float r = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    __syncthreads();
    return r;
}

template <int FMT>
__global__ void __launch_bounds__(256) sdf_trilinear_kernel(int m, const float *__restrict__ V, SdfVol A, float *__restrict__ out) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < m; i += gridDim.x * 256)
        out[i] = trilinear<FMT>(A, V[3 * i], V[3 * i + 1], V[3 * i + 2]);
}

// mean_j |Distance((pcld_j - t) @ R)| for the calling block's particle; every thread returns the value.
template <int FMT>
__device__ __forceinline__ float particle_sdf_energy(int n, const float *__restrict__ pcld, const float *R, const float *t,
                                                     const SdfVol &A, float *sm) {
    constexpr int U = SDF_U;  // points per thread whose gathers are all issued before any blend
    float acc = 0.0f;
    for (int j0 = threadIdx.x; j0 < n; j0 += 256 * U) {
        TriPoint tp[U];
        float d[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = min(j0 + 256 * u, n - 1);  // out-of-range slots re-read the last point and are masked below
            float ox, oy, oz;
            to_object_frame(pcld[3 * j], pcld[3 * j + 1], pcld[3 * j + 2], t, R, ox, oy, oz);
            tp[u] = tri_setup(A, ox, oy, oz);
            tri_gather<FMT>(A, tp[u], d[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float v = fabsf(tri_blend(A, tp[u], d[u]));
            acc += (j0 + 256 * u < n) ? v : 0.0f;
        }
    }
    return block_sum_256(acc, sm) / (float)n;
}

template <int FMT>
__global__ void __launch_bounds__(256)
sdf_particle_energy_kernel(int n, const float *__restrict__ pcld, const float *__restrict__ rot, const float *__restrict__ trans,
                           SdfVol A, float *__restrict__ sdf_energy) {
    __shared__ float sm[4];
    const int i = blockIdx.x;
    float R[9], t[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = rot[9 * i + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) t[k] = trans[3 * i + k];
    const float e = particle_sdf_energy<FMT>(n, pcld, R, t, A, sm);
    if (threadIdx.x == 0) sdf_energy[i] = e;
}

// ---- gf_optimize_obj.optimize's particle loop (optimization_obj.py:253-301) ----------------------------------------
// work layout (floats): [0..5] search size, [6..11] previous search size, [12] previous-success flag,
// [13] "previous search size is still the Python scalar c1" flag, [14] ticket counter (u32), [15] pad,
// [16 .. 16+p) per-particle sdf_energy.
constexpr int W_SEARCH = 0, W_PREV = 6, W_PREV_OK = 12, W_PREV_SCALAR = 13, W_TICKET = 14, W_ENERGY = 16;

__device__ __forceinline__ void quat_to_matrix(float w, float x, float y, float z, float *m) {  // rotations.py:105-113
    m[0] = 1.0f - 2.0f * y * y - 2.0f * z * z;  m[1] = 2.0f * x * y - 2.0f * z * w;         m[2] = 2.0f * x * z + 2.0f * y * w;
    m[3] = 2.0f * x * y + 2.0f * z * w;         m[4] = 1.0f - 2.0f * x * x - 2.0f * z * z;  m[5] = 2.0f * y * z - 2.0f * x * w;
    m[6] = 2.0f * x * z - 2.0f * y * w;         m[7] = 2.0f * y * z + 2.0f * x * w;         m[8] = 1.0f - 2.0f * x * x - 2.0f * y * y;
}

__device__ __forceinline__ void mat3_mul(const float *a, const float *b, float *o) {  // o = a @ b, fixed fma chain
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) o[3 * i + j] = fmaf(a[3 * i + 2], b[6 + j], fmaf(a[3 * i + 1], b[3 + j], a[3 * i] * b[j]));
}

// sample = [qw, pre*search] (:259-261)
__device__ __forceinline__ void particle_sample(const float *__restrict__ pre, int i, const float *search, float *s) {
#pragma unroll
    for (int k = 0; k < 6; ++k) s[1 + k] = pre[6 * i + k] * search[k];
    s[0] = sqrtf(1.0f - s[1] * s[1] - s[2] * s[2] - s[3] * s[3]);
}

__device__ __forceinline__ void normalize3(float *v) {  // rotations.py:328-340
    const float mag = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    if (mag > 1e-8f) {
        v[0] /= mag; v[1] /= mag; v[2] /= mag;
    } else {
        v[0] = 1.0f; v[1] = 0.0f; v[2] = 0.0f;
    }
}

struct OptArgs {
    int p, n;
    const float *pcld, *pre;
    SdfVol V;
    float c2, beta, one_minus_beta, carry0;
    float *pose, *work;
};

__global__ void opt_init_kernel(float *work, float c1) {
    const int t = threadIdx.x;
    if (t < 6) work[W_SEARCH + t] = c1;
    if (t >= 6 && t < 12) work[t] = c1;
    if (t == 12) work[W_PREV_OK] = 1.0f;
    if (t == 13) work[W_PREV_SCALAR] = 1.0f;
    if (t == 14) reinterpret_cast<unsigned *>(work)[W_TICKET] = 0u;
}

template <int FMT>
__global__ void __launch_bounds__(256) obj_optimize_kernel(const OptArgs A) {
    __shared__ float sm[4];
    __shared__ float red[9 * 4 + 4];
    __shared__ int is_last;
    const int i = blockIdx.x, tid = threadIdx.x;
    float *__restrict__ work = A.work;
    float *__restrict__ energy = work + W_ENERGY;

    float search[6], R0[9], t0[3];
#pragma unroll
    for (int k = 0; k < 6; ++k) search[k] = work[W_SEARCH + k];
#pragma unroll
    for (int k = 0; k < 9; ++k) R0[k] = A.pose[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) t0[k] = A.pose[9 + k];
    {
        float s[7], S[9], R[9], t[3];
        particle_sample(A.pre, i, search, s);
        quat_to_matrix(s[0], s[1], s[2], s[3], S);
        mat3_mul(R0, S, R);  // :263
#pragma unroll
        for (int k = 0; k < 3; ++k) t[k] = t0[k] + s[4 + k];  // :264
        const float e = particle_sdf_energy<FMT>(A.n, A.pcld, R, t, A.V, sm);
        // agent-scope (write-through) store: visible to every XCD without an L2 write-back
        if (tid == 0) __hip_atomic_store(&energy[i], e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- the last workgroup to finish performs the pose / search-size update of this iteration --------------------
    // No fences: a release/acquire pair at agent scope costs an L2 write-back / invalidate per workgroup on this
    // multi-XCD part (measured: 163 us per iteration instead of ~40, the shared volume keeps being evicted).
    // Instead energy[] is written and read with agent-scope atomics (coherent per location), and the store is
    // complete (vmcnt(0)) before this workgroup takes its ticket, so whoever draws the last ticket sees them all.
    if (tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned ticket = __hip_atomic_fetch_add(reinterpret_cast<unsigned *>(work) + W_TICKET, 1u, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT);
        is_last = (ticket == (unsigned)A.p - 1u);
    }
    __syncthreads();
    if (!is_last) return;

    const float origin_sdf = __hip_atomic_load(&energy[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float origin = origin_sdf * 500.0f;  // :236
    float acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.0f;
    int any = 0;
    // one workgroup walks all p energies: issue 8 independent loads per thread before using any (a dependent
    // load-use chain per particle made this tail ~15 us of a ~40 us iteration)
    constexpr int UQ = 8;
    for (int q0 = tid; q0 < A.p; q0 += 256 * UQ) {
        float se[UQ], pr[UQ][6];
#pragma unroll
        for (int u = 0; u < UQ; ++u) {
            const int q = min(q0 + 256 * u, A.p - 1);
            se[u] = __hip_atomic_load(&energy[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int k = 0; k < 6; ++k) pr[u][k] = A.pre[6 * q + k];
        }
#pragma unroll
        for (int u = 0; u < UQ; ++u) {
            if (q0 + 256 * u >= A.p) continue;
            const float en = se[u] * 500.0f;
            const bool better = en < origin;  // :271
            const float w = better ? origin - en : 0.0f;
            any |= better;
            float s[7];
#pragma unroll
            for (int k = 0; k < 6; ++k) s[1 + k] = pr[u][k] * search[k];  // == particle_sample
            s[0] = sqrtf(1.0f - s[1] * s[1] - s[2] * s[2] - s[3] * s[3]);
            acc[0] += w;
            acc[1] += se[u] * w;
#pragma unroll
            for (int k = 0; k < 7; ++k) acc[2 + k] += s[k] * w;
        }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        float v = acc[k];
#pragma unroll
        for (int o = 32; o; o >>= 1) v += __shfl_down(v, o);
        if ((tid & 63) == 0) red[k * 4 + (tid >> 6)] = v;
    }
    const unsigned long long any_b = __ballot(any);
    if ((tid & 63) == 0) red[36 + (tid >> 6)] = any_b ? 1.0f : 0.0f;
    __syncthreads();
    if (tid != 0) return;

    float sum[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) sum[k] = (red[k * 4] + red[k * 4 + 1]) + (red[k * 4 + 2] + red[k * 4 + 3]);
    const bool success = (red[36] + red[37] + red[38] + red[39]) > 0.0f;
    const float wsum = sum[0] + 1e-5f;  // :273
    float mean_sdf, mt[7];
    if (success) {
        mean_sdf = sum[1] / wsum;  // :275
#pragma unroll
        for (int k = 0; k < 7; ++k) mt[k] = sum[2 + k] / wsum;  // :283
        const float qn = sqrtf(mt[0] * mt[0] + mt[1] * mt[1] + mt[2] * mt[2] + mt[3] * mt[3]) + 1e-8f;
#pragma unroll
        for (int k = 0; k < 4; ++k) mt[k] /= qn;  // :284
        float S[9], Rn[9];
        quat_to_matrix(mt[0], mt[1], mt[2], mt[3], S);
        mat3_mul(R0, S, Rn);  // :285
        // SO(3) re-projection: Gram-Schmidt on the first two ROWS (:287, rotations.py:356-369 + transpose)
        float x[3] = {Rn[0], Rn[1], Rn[2]}, yr[3] = {Rn[3], Rn[4], Rn[5]}, z[3], y[3];
        normalize3(x);
        z[0] = x[1] * yr[2] - x[2] * yr[1]; z[1] = x[2] * yr[0] - x[0] * yr[2]; z[2] = x[0] * yr[1] - x[1] * yr[0];
        normalize3(z);
        y[0] = z[1] * x[2] - z[2] * x[1]; y[1] = z[2] * x[0] - z[0] * x[2]; y[2] = z[0] * x[1] - z[1] * x[0];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            A.pose[k] = x[k];
            A.pose[3 + k] = y[k];
            A.pose[6 + k] = z[k];
            A.pose[9 + k] = t0[k] + mt[4 + k];  // :288
        }
    } else {
        mean_sdf = origin_sdf;  // :278
#pragma unroll
        for (int k = 0; k < 7; ++k) mt[k] = 0.0f;
    }
    // update_seach_size (:239-242) and its smoothing (:293-299)
    float s[6], nrm = 0.0f;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        s[k] = fabsf(mt[1 + k]) + 1e-3f;
        nrm += s[k] * s[k];
    }
    nrm = sqrtf(nrm);
    const bool prev_ok = work[W_PREV_OK] != 0.0f, prev_scalar = work[W_PREV_SCALAR] != 0.0f;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        float ns = mean_sdf * A.c2 * s[k] / nrm + 1e-3f;
        if (prev_ok && success) {
            const float carry = prev_scalar ? A.carry0 : A.one_minus_beta * work[W_PREV + k];
            ns = A.beta * ns + carry;
        }
        work[W_SEARCH + k] = ns;
        if (success) work[W_PREV + k] = ns;
    }
    if (success) work[W_PREV_SCALAR] = 0.0f;
    work[W_PREV_OK] = success ? 1.0f : 0.0f;
    reinterpret_cast<unsigned *>(work)[W_TICKET] = 0u;
}

// ---- gf_optimize_hand_pose.query_sdf (+ get_penetration_loss) (optimization_hand.py:252-268) -----------------------
// torch's `tensor // scalar` on floats is c10::div_floor_floating: fmod, (a - mod) / b, sign fix-up, floor, and a
// +1 if that floor fell below the rounding error -- for b > 0 and |a/b| < 2^22 that is exactly the mathematical
// floor of the real quotient a/b (derivation in DESIGN.md section 8).  fmodf costs ~100 instructions and made this
// kernel ALU-bound (39 us); the same integer comes from one correctly rounded division and one exact-sign FMA
// remainder:  k = floor(RN(a/b));  r = fma(-k, b, a)  has the sign of the true remainder a - k*b (an FMA rounds
// once and never rounds a non-zero value to zero, so its SIGN is exact -- its magnitude is not: a tiny negative a
// gives r = b - tiny, which rounds to b, hence the second test looks at the sign of a - (k+1)*b instead of r >= b);
// k is off by at most one, fixed by the two sign tests.  Beyond 2^22 both versions are far outside the clamp
// range [-res/2, res/2] applied next, so the voxel index is identical for every finite input
// (tests: bit-exact indices vs the literal restatement in oracle/sdf_oracle.c on adversarial k*b +- ulp inputs).
__device__ __forceinline__ float div_floor(float a, float b) {
    float k = floorf(a / b);
    if (fmaf(-k, b, a) < 0.0f) k -= 1.0f;                  // a - k*b < 0: k is one too large
    else if (fmaf(-(k + 1.0f), b, a) >= 0.0f) k += 1.0f;   // a - (k+1)*b >= 0: k is one too small
    return k;
}

template <bool F16>
__global__ void __launch_bounds__(256)
sdf_nearest_kernel(int n, const float *__restrict__ hand, const float *__restrict__ obj_r, const float *__restrict__ obj_t,
                   const void *__restrict__ vol, int res, float voxel_scale, int *__restrict__ out_idx,
                   void *__restrict__ out_sdf, void *__restrict__ out_pen) {
    __shared__ float sm[4];
    const int b = blockIdx.x;
    float R[9], t[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = obj_r[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) t[k] = obj_t[k];
    const int half = res / 2;
    const float fh = (float)half;
    float pen = 0.0f;
    constexpr int U = 4;
    for (int j0 = threadIdx.x; j0 < n; j0 += 256 * U) {
        int flat[U];
        float v[U];
        __half h[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t e = (size_t)b * n + min(j0 + 256 * u, n - 1);
            float ox, oy, oz;
            to_object_frame(hand[3 * e], hand[3 * e + 1], hand[3 * e + 2], t, R, ox, oy, oz);
            const int ix = (int)clampf(div_floor(ox, voxel_scale), -fh, fh) + half;
            const int iy = (int)clampf(div_floor(oy, voxel_scale), -fh, fh) + half;
            const int iz = (int)clampf(div_floor(oz, voxel_scale), -fh, fh) + half;
            flat[u] = (ix * res + iy) * res + iz;
            if constexpr (F16) h[u] = reinterpret_cast<const __half *>(vol)[flat[u]];
            else v[u] = reinterpret_cast<const float *>(vol)[flat[u]];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (j0 + 256 * u >= n) continue;
            const size_t e = (size_t)b * n + j0 + 256 * u;
            if (out_idx) out_idx[e] = flat[u];
            if constexpr (F16) {
                if (out_sdf) reinterpret_cast<__half *>(out_sdf)[e] = h[u];
                v[u] = __half2float(h[u]);
            } else {
                if (out_sdf) reinterpret_cast<float *>(out_sdf)[e] = v[u];
            }
            if (v[u] < 0.0f) pen = fmaxf(pen, -v[u]);
        }
    }
    if (!out_pen) return;
#pragma unroll
    for (int o = 32; o; o >>= 1) pen = fmaxf(pen, __shfl_down(pen, o));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = pen;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
        if constexpr (F16) reinterpret_cast<__half *>(out_pen)[b] = __float2half(m);  // exact: m is a half value
        else reinterpret_cast<float *>(out_pen)[b] = m;
    }
}

// Corner layout: cell i = the eight corner values the reference's Distance() fetches when i000 == i, with ITS index
// arithmetic (flat +1 / +R / +R^2 offsets, each clamped to the last element, optimization_obj.py:206-222), so a
// lookup through this layout is bit-identical to one through the linear volume for every input.
template <bool F16>
__global__ void __launch_bounds__(256) corner_volume_kernel(const void *__restrict__ vol, int res, void *__restrict__ out) {
    const int R = res, RR = R * R, last = RR * R - 1;
    for (int i = blockIdx.x * 256 + threadIdx.x; i <= last; i += gridDim.x * 256) {
        const int id[8] = {i, min(i + 1, last), min(i + R, last), min(i + 1 + R, last),
                           min(i + RR, last), min(i + 1 + RR, last), min(i + R + RR, last), min(i + 1 + R + RR, last)};
        if constexpr (F16) {
            const unsigned short *v = reinterpret_cast<const unsigned short *>(vol);
            uint4 w;
            w.x = v[id[0]] | ((unsigned)v[id[1]] << 16);
            w.y = v[id[2]] | ((unsigned)v[id[3]] << 16);
            w.z = v[id[4]] | ((unsigned)v[id[5]] << 16);
            w.w = v[id[6]] | ((unsigned)v[id[7]] << 16);
            reinterpret_cast<uint4 *>(out)[i] = w;
        } else {
            const float *v = reinterpret_cast<const float *>(vol);
            reinterpret_cast<float4 *>(out)[2 * (size_t)i] = make_float4(v[id[0]], v[id[1]], v[id[2]], v[id[3]]);
            reinterpret_cast<float4 *>(out)[2 * (size_t)i + 1] = make_float4(v[id[4]], v[id[5]], v[id[6]], v[id[7]]);
        }
    }
}

inline bool vol_args_ok(int res, float stride) { return res >= 2 && res <= 1024 && stride > 0.0f; }

}  // namespace pn2

using namespace pn2;

#define SDF_FMT_DISPATCH(fmt, CALL)                   \
    switch (fmt) {                                    \
        case 0: { constexpr int FMT = 0; CALL; } break; \
        case 1: { constexpr int FMT = 1; CALL; } break; \
        case 2: { constexpr int FMT = 2; CALL; } break; \
        case 3: { constexpr int FMT = 3; CALL; } break; \
        default: return PN2_EINVAL;                   \
    }

extern "C" long pn2s_corner_volume_elems(int res) { return (res < 2 || res > 1024) ? (long)PN2_EINVAL : 8L * res * res * res; }

extern "C" int pn2s_build_corner_volume(const void *vol, int vol_f16, int res, void *out, void *stream) {
    if (res < 2 || res > 1024 || (vol_f16 != 0 && vol_f16 != 1)) return PN2_EINVAL;
    if (!vol || !out) return PN2_ENULL;
    const long cells = (long)res * res * res;
    const unsigned blocks = (unsigned)min((cells + 255) / 256, 65536L);
    hipStream_t st = (hipStream_t)stream;
    if (vol_f16) hipLaunchKernelGGL(corner_volume_kernel<true>, dim3(blocks), dim3(256), 0, st, vol, res, out);
    else hipLaunchKernelGGL(corner_volume_kernel<false>, dim3(blocks), dim3(256), 0, st, vol, res, out);
    return check_launch();
}

extern "C" int pn2s_trilinear(int m, const float *V, const void *vol, int vol_fmt, int res, float bbox_min, float stride,
                              float clamp_lo, float clamp_hi, float *out, void *stream) {
    if (m < 0 || !vol_args_ok(res, stride)) return PN2_EINVAL;
    if (m == 0) return (vol_fmt < 0 || vol_fmt > 3) ? PN2_EINVAL : PN2_OK;
    if (!V || !vol || !out) return PN2_ENULL;
    const SdfVol A{vol, res, bbox_min, stride, clamp_lo, clamp_hi};
    const unsigned blocks = (unsigned)min((m + 255) / 256, 16384);
    hipStream_t st = (hipStream_t)stream;
    SDF_FMT_DISPATCH(vol_fmt, hipLaunchKernelGGL(sdf_trilinear_kernel<FMT>, dim3(blocks), dim3(256), 0, st, m, V, A, out))
    return check_launch();
}

extern "C" int pn2s_particle_energy(int p, int n, const float *pcld, const float *rot, const float *trans, const void *vol,
                                    int vol_fmt, int res, float bbox_min, float stride, float clamp_lo, float clamp_hi,
                                    float *sdf_energy, void *stream) {
    if (p < 0 || n < 1 || !vol_args_ok(res, stride)) return PN2_EINVAL;
    if (p == 0) return (vol_fmt < 0 || vol_fmt > 3) ? PN2_EINVAL : PN2_OK;
    if (!pcld || !rot || !trans || !vol || !sdf_energy) return PN2_ENULL;
    const SdfVol A{vol, res, bbox_min, stride, clamp_lo, clamp_hi};
    hipStream_t st = (hipStream_t)stream;
    SDF_FMT_DISPATCH(vol_fmt, hipLaunchKernelGGL(sdf_particle_energy_kernel<FMT>, dim3(p), dim3(256), 0, st, n, pcld, rot, trans, A, sdf_energy))
    return check_launch();
}

extern "C" int pn2s_obj_optimize_work_floats(int p) { return p < 0 ? PN2_EINVAL : W_ENERGY + p; }

extern "C" int pn2s_obj_optimize(int p, int n, int iterations, const float *pcld, const float *pre_sampled, const void *vol,
                                 int vol_fmt, int res, float bbox_min, float stride, float clamp_lo, float clamp_hi, float c1,
                                 float c2, float beta, float *pose, float *work, void *stream) {
    if (p < 1 || n < 1 || iterations < 0 || !vol_args_ok(res, stride) || vol_fmt < 0 || vol_fmt > 3) return PN2_EINVAL;
    if (!pcld || !pre_sampled || !vol || !pose || !work) return PN2_ENULL;
    OptArgs A;
    A.p = p; A.n = n; A.pcld = pcld; A.pre = pre_sampled;
    A.V = SdfVol{vol, res, bbox_min, stride, clamp_lo, clamp_hi};
    A.c2 = c2; A.beta = beta;
    // the reference mixes a Python double scalar into fp32 tensors here (:295): (1-beta) is rounded to fp32 when it
    // multiplies a tensor, but (1-beta)*c1 is a double product rounded once while prev_search_size is still c1
    A.one_minus_beta = (float)(1.0 - (double)beta);
    A.carry0 = (float)((1.0 - (double)beta) * (double)c1);
    A.pose = pose; A.work = work;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(opt_init_kernel, dim3(1), dim3(64), 0, st, work, c1);
    for (int it = 0; it < iterations; ++it) {
        SDF_FMT_DISPATCH(vol_fmt, hipLaunchKernelGGL(obj_optimize_kernel<FMT>, dim3(p), dim3(256), 0, st, A))
    }
    return check_launch();
}

extern "C" int pn2s_nearest(int b, int n, const float *hand, const float *obj_r, const float *obj_t, const void *vol,
                            int vol_f16, int res, float voxel_scale, int *out_idx, void *out_sdf, void *out_pen, void *stream) {
    if (b < 0 || n < 1 || res < 1 || res > 1024 || (res & 1) == 0 || !(voxel_scale > 0.0f)) return PN2_EINVAL;
    if (b == 0) return PN2_OK;
    if (!hand || !obj_r || !obj_t || !vol) return PN2_ENULL;
    hipStream_t st = (hipStream_t)stream;
    if (vol_f16) hipLaunchKernelGGL(sdf_nearest_kernel<true>, dim3(b), dim3(256), 0, st, n, hand, obj_r, obj_t, vol, res, voxel_scale, out_idx, out_sdf, out_pen);
    else hipLaunchKernelGGL(sdf_nearest_kernel<false>, dim3(b), dim3(256), 0, st, n, hand, obj_r, obj_t, vol, res, voxel_scale, out_idx, out_sdf, out_pen);
    return check_launch();
}
