// ball_query.hip -- radius neighbour search for gfx950.
//
// Replaces ball_query_kernel_fast (reference ball_query_gpu.cu:9-66), which runs one thread
// per centroid scanning all N points from global memory.  Here:
//   * the cloud is staged ONCE per workgroup into LDS as SoA x[]/y[]/z[] tiles
//     (conflict-free ds_read_b32 across lanes), coalesced from the AoS HBM layout;
//   * one WAVE per centroid: the 64 lanes test 64 consecutive candidates, a ballot plus
//     mbcnt prefix-popcount appends the hits in ascending index order (the reference's
//     serial scan order), and the wave leaves the scan as soon as nsample hits are found;
//   * every element of the output row is written (hits, then first-hit padding, or zeros),
//     so the caller does not have to pre-zero idx (pointnet2_utils.py:262).
#include "pn2_common.h"
#include "fps_tie.h"

namespace pn2 {

constexpr int kBqThreads = 256;
constexpr int kBqWaves = kBqThreads / kWave;
constexpr int kBqTile = 4096;  // points per LDS tile: 48 KiB -> 3 workgroups / CU

// workgroup (bx, by) of a (ceil(m / (waves CPW)), b) grid
template <int CPW, bool JOINT>  // centroids per wave; JOINT: they walk the candidates together (shared LDS reads, packed distances)
__device__ __forceinline__ void ball_query_body(const int bx, const int by, int n, int m, float radius2, int nsample,
                                                const float *__restrict__ new_xyz_all, const float *__restrict__ xyz_all,
                                                int *__restrict__ idx_all, const int *__restrict__ picks_all,
                                                float *__restrict__ new_xyz_out_all, float *__restrict__ new_xyz_copy, int copy_ld) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tile_cap = n < kBqTile ? n : kBqTile;
    float *sx = smem, *sy = smem + tile_cap, *sz = smem + 2 * tile_cap;

    const int b = by;
    const float *__restrict__ xyz = xyz_all + (size_t)b * n * 3;
    // centroids either given by coordinates (the reference operator) or as indices into this cloud (`picks`, the
    // output of FPS -- pn2x_ball_query_picks), in which case their coordinates are also written out: the gather
    // launch that follows FPS in the reference (pointnet_utils.py:379) is folded into the query
    const float *__restrict__ new_xyz = picks_all ? nullptr : new_xyz_all + (size_t)b * m * 3;
    const int *__restrict__ picks = picks_all ? picks_all + (size_t)b * m : nullptr;
    int *__restrict__ idx = idx_all + (size_t)b * m * nsample;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int c_base = bx * (kBqWaves * CPW);

    float cx[CPW], cy[CPW], cz[CPW];
    int cnt[CPW], first[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int s = c_base + c * kBqWaves + w;
        const bool ok = s < m;
        const float *__restrict__ src = !ok ? nullptr : (picks ? xyz + 3 * (size_t)picks[s] : new_xyz + 3 * (size_t)s);
        cx[c] = ok ? src[0] : 0.f;
        cy[c] = ok ? src[1] : 0.f;
        cz[c] = ok ? src[2] : 0.f;
        if (ok && picks && lane == 0) {
            float *o = new_xyz_out_all + ((size_t)b * m + s) * 3;
            o[0] = cx[c]; o[1] = cy[c]; o[2] = cz[c];
            if (new_xyz_copy) {  // second copy into three columns of a consumer's wider row buffer
                float *o2 = new_xyz_copy + ((size_t)b * m + s) * copy_ld;
                o2[0] = cx[c]; o2[1] = cy[c]; o2[2] = cz[c];
            }
        }
        cnt[c] = ok ? 0 : nsample;  // out-of-range centroids are "done"
        first[c] = 0;
    }

    for (int t0 = 0; t0 < n; t0 += kBqTile) {
        const int tn = (n - t0) < kBqTile ? (n - t0) : kBqTile;
        if (t0 > 0) __syncthreads();
        // AoS (12 B/pt) -> SoA; 3*tn consecutive floats read coalesced.
        for (int i = tid; i < 3 * tn; i += kBqThreads) {
            const float v = xyz[(size_t)3 * t0 + i];
            const int p = i / 3, comp = i - 3 * p;
            (comp == 0 ? sx : comp == 1 ? sy : sz)[p] = v;
        }
        __syncthreads();
        if constexpr (JOINT) {  // long scans (n >= 4096): measured 1.2-1.6x; short ones keep their cheaper per-centroid early exit
            // All CPW centroids of the wave walk the tile TOGETHER: a candidate's coordinates are read from LDS once per step and,
            // for centroid pairs, the distance arithmetic runs on the packed fp32 ops (sqdist2: the same IEEE operations per
            // half, bit-identical to sqdist) -- the scan is VALU / LDS-issue bound, not memory bound.  A centroid that has its
            // nsample hits drops out (wave-uniform test); the walk ends when all have.
            int open = 0;
#pragma unroll
            for (int c = 0; c < CPW; ++c) open += cnt[c] < nsample ? 1 : 0;
            for (int base = 0; base < tn && open > 0; base += kWave) {
                const int p = base + lane;
                const bool in = p < tn;
                const float x = in ? sx[p] : 0.f, y = in ? sy[p] : 0.f, z = in ? sz[p] : 0.f;
                float d2[CPW];
                if constexpr (CPW >= 2) {
#pragma unroll
                    for (int c = 0; c < CPW; c += 2) {
                        const pn2_f32x2 d = sqdist2((pn2_f32x2){cx[c], cx[c + 1]}, (pn2_f32x2){cy[c], cy[c + 1]}, (pn2_f32x2){cz[c], cz[c + 1]}, x, y, z);
                        d2[c] = d.x;
                        d2[c + 1] = d.y;
                    }
                } else {
                    d2[0] = sqdist(cx[0], cy[0], cz[0], x, y, z);
                }
#pragma unroll
                for (int c = 0; c < CPW; ++c) {
                    if (cnt[c] >= nsample) continue;  // wave-uniform
                    const bool hit = in && (d2[c] < radius2);
                    const uint64_t mask = __ballot(hit);
                    if (mask) {
                        int *__restrict__ row = idx + (size_t)(c_base + c * kBqWaves + w) * nsample;
                        if (cnt[c] == 0) first[c] = t0 + base + __builtin_ctzll(mask);
                        const int pos = cnt[c] + prefix_popc(mask);
                        if (hit && pos < nsample) row[pos] = t0 + p;
                        cnt[c] += __builtin_popcountll(mask);
                        if (cnt[c] >= nsample) --open;
                    }
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < CPW; ++c) {
                const int s = c_base + c * kBqWaves + w;
                int *__restrict__ row = idx + (size_t)s * nsample;
                int mycnt = cnt[c];
                if (mycnt >= nsample) continue;  // wave-uniform
                for (int base = 0; base < tn; base += kWave) {
                    const int p = base + lane;
                    const bool in = p < tn;
                    const float x = in ? sx[p] : 0.f, y = in ? sy[p] : 0.f, z = in ? sz[p] : 0.f;
                    const float d2 = sqdist(cx[c], cy[c], cz[c], x, y, z);
                    const bool hit = in && (d2 < radius2);
                    const uint64_t mask = __ballot(hit);
                    if (mask) {
                        if (mycnt == 0) first[c] = t0 + base + __builtin_ctzll(mask);
                        const int pos = mycnt + prefix_popc(mask);
                        if (hit && pos < nsample) row[pos] = t0 + p;
                        mycnt += __builtin_popcountll(mask);
                        if (mycnt >= nsample) break;
                    }
                }
                cnt[c] = mycnt;
            }
        }
    }
    // padding (ball_query_gpu.cu:35-39: slots beyond the hits repeat the first hit; no hit -> 0)
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int s = c_base + c * kBqWaves + w;
        if (s >= m) continue;
        int *__restrict__ row = idx + (size_t)s * nsample;
        const int fill = cnt[c] > 0 ? first[c] : 0;
        for (int p = cnt[c] + lane; p < nsample; p += kWave) row[p] = fill;
    }
}

template <int CPW, bool JOINT>
__global__ void __launch_bounds__(kBqThreads)
ball_query_kernel(int n, int m, float radius2, int nsample, const float *__restrict__ new_xyz_all,
                  const float *__restrict__ xyz_all, int *__restrict__ idx_all, const int *__restrict__ picks_all,
                  float *__restrict__ new_xyz_out_all, float *__restrict__ new_xyz_copy, int copy_ld) {
    ball_query_body<CPW, JOINT>((int)blockIdx.x, (int)blockIdx.y, n, m, radius2, nsample, new_xyz_all, xyz_all, idx_all, picks_all,
                                new_xyz_out_all, new_xyz_copy, copy_ld);
}

// The level-1 ball query around the picks of a sampling run AND that run's tie check (fps_tie.h) in one launch: both need the
// picks and nothing of each other.  Workgroups [0, nb_ball) query, the rest check (pn2x_ball_query_picks_ties).
template <int CPW>
__global__ void __launch_bounds__(kBqThreads)
ball_tie_kernel(int nbx_ball, int nb_ball, int nbx_tie, int n, int m, float radius2, int nsample, const float *__restrict__ xyz_all,
                int *__restrict__ idx_all, const int *__restrict__ picks_all, float *__restrict__ new_xyz_out_all,
                float *__restrict__ new_xyz_copy, int copy_ld, int m2, const float *__restrict__ radii_all, int *__restrict__ flags) {
    const int L = (int)blockIdx.x;
    if (L < nb_ball) {  // workgroup-uniform
        const int by = L / nbx_ball;
        // (new_xyz_all is unused with picks; a literal nullptr here crashes the inliner of this hipcc)
        ball_query_body<CPW, false>(L - by * nbx_ball, by, n, m, radius2, nsample, xyz_all, xyz_all, idx_all, picks_all, new_xyz_out_all,
                                    new_xyz_copy, copy_ld);
    } else {
        const int T = L - nb_ball, by = T / nbx_tie;
        fps_tie_body(T - by * nbx_tie, by, nbx_tie, n, m2, m, xyz_all, picks_all, radii_all, flags);
    }
}

int ball_query_dispatch(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                        const float *xyz, int *idx, hipStream_t st, const int *picks, float *new_xyz_out, float *new_xyz_copy,
                        int copy_ld) {
    if (b == 0 || m == 0) return PN2_OK;
    const float radius2 = radius * radius;  // fp32 product, ball_query_gpu.cu:23
    const int tile_cap = n < kBqTile ? n : kBqTile;
    const size_t lds = (size_t)3 * tile_cap * sizeof(float);
    // pick centroids/wave so the grid still oversubscribes 256 CUs when it can
    const long total = (long)b * m;
    int cpw = 1;
    if (total >= 4L * kBqWaves * 4096) cpw = 4;
    else if (total >= 2L * kBqWaves * 2048) cpw = 2;
    const int per_block = kBqWaves * cpw;
    dim3 grid((m + per_block - 1) / per_block, b);
    const bool joint = cpw >= 2 && n >= 4096;
    if (cpw == 4 && joint)
        hipLaunchKernelGGL((ball_query_kernel<4, true>), grid, dim3(kBqThreads), lds, st, n, m, radius2, nsample, new_xyz, xyz, idx, picks, new_xyz_out, new_xyz_copy, copy_ld);
    else if (cpw == 4)
        hipLaunchKernelGGL((ball_query_kernel<4, false>), grid, dim3(kBqThreads), lds, st, n, m, radius2, nsample, new_xyz, xyz, idx, picks, new_xyz_out, new_xyz_copy, copy_ld);
    else if (cpw == 2 && joint)
        hipLaunchKernelGGL((ball_query_kernel<2, true>), grid, dim3(kBqThreads), lds, st, n, m, radius2, nsample, new_xyz, xyz, idx, picks, new_xyz_out, new_xyz_copy, copy_ld);
    else if (cpw == 2)
        hipLaunchKernelGGL((ball_query_kernel<2, false>), grid, dim3(kBqThreads), lds, st, n, m, radius2, nsample, new_xyz, xyz, idx, picks, new_xyz_out, new_xyz_copy, copy_ld);
    else
        hipLaunchKernelGGL((ball_query_kernel<1, false>), grid, dim3(kBqThreads), lds, st, n, m, radius2, nsample, new_xyz, xyz, idx, picks, new_xyz_out, new_xyz_copy, copy_ld);
    return check_launch();
}

// pn2x_ball_query_picks_ties: the co-launch above where it applies (the per-centroid scan variants, tie check within its pick limit)
// Small batches only (the tracking loop): there the two launches are two links of a dependent chain of ~4 us launches; with
// many clouds in flight the query workgroups do better without the check's 21 KB of static LDS (headline, same box: 88.5 k
// frames/s with, 88.8 k without).
// LDS: the tie check's static 21.5 KB (fps_tie_body: picks, check keys, chunk totals) ride on top of the query's dynamic cloud
// tile (12 B per point, up to kBqTile points): held to 64 KB per workgroup in total -- the limit a launch gets without opting
// in to more -- i.e. clouds of at most 3584 points take this path (larger clouds use the two separate launches).
constexpr long kBallTieMaxN = (64 * 1024 - 22 * 1024) / 12 / 64 * 64;  // 3584
bool ball_tie_supported(long b, long n, long m, long m2) {
    return b >= 1 && m >= 1 && n <= kBallTieMaxN && b * n <= 16384 && b * m < 2L * kBqWaves * 2048 && m2 >= 1 && m2 <= m &&
           m2 <= kTieMaxM;
}

int ball_tie_dispatch(int b, int n, int m, float radius, int nsample, const float *xyz, int *idx, const int *picks, float *new_xyz_out,
                      float *new_xyz_copy, int copy_ld, int m2, const float *radii, int *flags, hipStream_t st) {
    if (!ball_tie_supported(b, n, m, m2)) return PN2_ERANGE;
    const float radius2 = radius * radius;
    const int tile_cap = n < kBqTile ? n : kBqTile;
    const size_t lds = (size_t)3 * tile_cap * sizeof(float);
    const int per_block = kBqWaves;  // one centroid per wave (what ball_query_dispatch picks at these sizes)
    const int nbx_ball = (m + per_block - 1) / per_block, nbx_tie = (n + kTiePts - 1) / kTiePts;
    const long grid = (long)b * nbx_ball + (long)b * nbx_tie;
    hipLaunchKernelGGL(ball_tie_kernel<1>, dim3((unsigned)grid), dim3(kBqThreads), lds, st, nbx_ball, b * nbx_ball, nbx_tie, n, m, radius2, nsample,
                       xyz, idx, picks, new_xyz_out, new_xyz_copy, copy_ld, m2, radii, flags);
    return check_launch();
}

}  // namespace pn2
