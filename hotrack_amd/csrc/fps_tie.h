// fps_tie.h -- the post-hoc tie check of a finished sampling run as a device function, shared by fps.hip (fps_tie_check_kernel) and
// ball_query.hip (ball_tie_kernel: the check rides in the launch of the level-1 ball query, which needs the same picks).
#pragma once
#include "pn2_common.h"

namespace pn2 {

// ---- were the first m picks of a finished FPS run unique arg-maxima?  (pn2x_fps_prefix_ties, pn2_ext.h) ----------------
// With the picks known there is no dependency chain left: every point replays its own running minimum against the
// picks in order (the same sqdist / min chain as the sampling kernel, hence the same floats) and compares it with
// the value the sampling kernel recorded for that pick (radii[i] = the maximum it selected at step i).  A point other
// than pick i that reaches radii[i] at step i means that arg-max was tied.  One workgroup per 256 points.
constexpr int kTieMaxM = 1024;
constexpr int kTieChunks = 4, kTiePts = 256 / kTieChunks;  // a workgroup = 64 points x 4 chunks of the pick sequence
// The running minimum is a prefix-min, and min is associative: chunk c of the picks is replayed by its own thread
// (wave c of the workgroup; pick reads are wave-uniform LDS broadcasts), first to get the chunk's total, then -- with
// the minimum of the earlier chunks' totals as the starting value -- again with the comparison.  Two passes over a
// quarter of the picks instead of one over all of them: the per-thread latency chain halves.
// workgroup (bx, by) of an (nbx, b) grid of 256 threads
__device__ __forceinline__ void fps_tie_body(const int bx, const int by, const int nbx, int n, int m, int m1,
                                             const float *__restrict__ xyz_all, const int *__restrict__ idx_all,
                                             const float *__restrict__ radii_all, int *__restrict__ flags) {
    __shared__ float4 pick[kTieMaxM];  // x, y, z of pick i, radius of pick i+1 (what a point is compared with after meeting pick i)
    __shared__ int ck[kTieMaxM];
    __shared__ float tot[kTieChunks][kTiePts];
    __shared__ int tie_any;
    const float *__restrict__ xyz = xyz_all + (size_t)by * n * 3;
    const int *__restrict__ idx = idx_all + (size_t)by * m1;
    const float *__restrict__ radii = radii_all + (size_t)by * m1;
    const int tid = threadIdx.x;
    if (tid == 0) tie_any = 0;
    for (int i = tid; i < m; i += 256) {
        const int k = idx[i];
        ck[i] = k;
        pick[i] = make_float4(xyz[3 * k], xyz[3 * k + 1], xyz[3 * k + 2], i + 1 < m ? radii[i + 1] : -1.0f);
    }
    __syncthreads();
    const int q = tid & (kTiePts - 1), c = tid / kTiePts;  // c is wave-uniform
    const int k = bx * kTiePts + q;
    const int steps = m - 1;                                 // picks 0 .. m-2 are met, pick i is compared with radius i+1
    const int len = (steps + kTieChunks - 1) / kTieChunks;
    const int i0 = c * len, i1 = min(steps, i0 + len);
    const bool live = k < n;
    const float px = live ? xyz[3 * k] : 0.f, py = live ? xyz[3 * k + 1] : 0.f, pz = live ? xyz[3 * k + 2] : 0.f;
    float d = 1e10f;
#pragma unroll 8
    for (int i = i0; i < i1; ++i) {
        const float4 p = pick[i];
        d = fmin_raw(sqdist(px, py, pz, p.x, p.y, p.z), d);
    }
    tot[c][q] = d;
    __syncthreads();
    d = 1e10f;
    for (int cc = 0; cc < c; ++cc) d = fmin_raw(tot[cc][q], d);  // min is exact: any association gives the same float
    bool tie = false;
#pragma unroll 8
    for (int i = i0; i < i1; ++i) {
        const float4 p = pick[i];
        d = fmin_raw(sqdist(px, py, pz, p.x, p.y, p.z), d);
        if (d == p.w) tie |= live && (k != ck[i + 1]);  // rarely true: the pick itself, or a tie
    }
    if (__ballot(tie) != 0 && (tid & 63) == 0) tie_any = 1;
    __syncthreads();
    if (tid == 0) flags[(size_t)by * nbx + bx] = tie_any;
}

}  // namespace pn2
