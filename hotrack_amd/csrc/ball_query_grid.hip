// ball_query_grid.hip -- index-exact ball query through a cell grid, for large clouds (gfx950).
//
// Same result as ball_query_kernel_fast (reference ball_query_gpu.cu:9-66) and as the brute-force kernel of
// ball_query.hip, bit for bit: the first `nsample` indices k in ASCENDING order with d2(new_xyz[s], xyz[k]) < r*r (the same
// fp32 expression on the same pairs), padded with the first hit, all zeros when there is none.  What changes is which pairs
// are evaluated: at N = 8192, r = 0.1 the brute-force scan evaluates all 8192 candidates per centroid to find ~34 hits
// (0.13 of the VALU rate, 0.007 of the HBM roofline: it is neither memory- nor latency-bound, just wasted arithmetic).
//
//   build  (one workgroup per cloud)  bounding box -> cells of side >= r (1 + 1e-3) (so every point within r of a centroid
//          lies in the 3x3x3 cells around the centroid's cell, fp rounding of the cell coordinate included); the index range is
//          cut into BLOCKS of NB consecutive indices and the points are counting-sorted by (block, cell) into 16-byte records
//          (x, y, z, index);
//   query  (one wave per centroid)  blocks in index order; per block the 3x3 columns of cells around the centroid are 9
//          contiguous runs of records; every hit sets bit (index mod NB) of a per-wave LDS bitmap, which is then read out in
//          order -- ascending indices without sorting -- and the walk stops after the block in which the nsample-th hit
//          fell (the early exit of the reference's serial scan: for dense neighbourhoods, r = 0.2, only the first block or
//          two are visited).  NB is chosen at build time so that a block is expected to hold ~1.5 nsample hits.
#include "pn2_common.h"
#include "../../include/pn2_ext.h"

namespace pn2 {
namespace bqg {

constexpr int kBuildT = 1024;
constexpr int kQueryT = 256;
constexpr int kQueryWaves = kQueryT / 64;
constexpr int kMaxKeys = 12288;   // (block, cell) counters in LDS: 48 KiB
constexpr int kMaxAxis = 16;
constexpr int kHeader = 16;       // 4-byte words

struct Header {  // kHeader words at the start of a cloud's scratch
    float minx, miny, minz, invx, invy, invz;
    int gx, gy, gz, wpl, nblocks, ncell, pad0, pad1, pad2, pad3;
};
static_assert(sizeof(Header) == kHeader * 4, "header size");

__host__ __device__ inline size_t cloud_words(int n) { return (size_t)kHeader + (kMaxKeys + 4) + (size_t)4 * n; }

__device__ __forceinline__ float fmin_nan(float a, float b) { return fminf(a, b); }  // fminf / fmaxf ignore a NaN operand

__global__ void __launch_bounds__(kBuildT)
build_kernel(int n, float radius, int nsample, const float *__restrict__ xyz_all, unsigned *__restrict__ scratch_all) {
    __shared__ float red[6][kBuildT / 64];
    __shared__ Header hd;
    extern __shared__ int cnt[];  // [nkeys + 1] counts -> cursors, then [kBuildT] chunk totals
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *__restrict__ xyz = xyz_all + (size_t)b * n * 3;
    unsigned *__restrict__ scratch = scratch_all + (size_t)b * cloud_words(n);
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int i = tid; i < n; i += kBuildT) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = xyz[3 * (size_t)i + a];
            lo[a] = fmin_nan(lo[a], v);
            hi[a] = fmaxf(hi[a], v);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = fmin_nan(lo[a], __shfl_xor(lo[a], o));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o));
        }
        if (lane == 0) { red[a][wave] = lo[a]; red[3 + a][wave] = hi[a]; }
    }
    __syncthreads();
    if (tid == 0) {
        float mn[3], ext[3];
        int g[3];
        const float h = radius * 1.001f;  // cell side lower bound: > r by more than any rounding of d2 or of the cell coordinate
        for (int a = 0; a < 3; ++a) {
            float l = red[a][0], u = red[3 + a][0];
            for (int w = 1; w < kBuildT / 64; ++w) { l = fmin_nan(l, red[a][w]); u = fmaxf(u, red[3 + a][w]); }
            mn[a] = l;
            ext[a] = u - l;
            const bool ok = ext[a] > 0.f && ext[a] < 3.0e38f && h > 0.f && l > -3.0e38f;  // finite, non-degenerate
            if (!ok) { g[a] = 1; ext[a] = 0.f; if (!(l > -3.0e38f && l < 3.0e38f)) mn[a] = 0.f; }
            else {
                const float q = ext[a] / h;
                g[a] = q >= (float)kMaxAxis ? kMaxAxis : (int)q + 1;
            }
        }
        // blocks of NB = 2048 * wpl consecutive indices: a block should hold ~1.5 nsample hits of a typical centroid
        float vol = 1.f;
        bool vol_ok = true;
        for (int a = 0; a < 3; ++a) { if (ext[a] > 0.f) vol *= ext[a]; else vol_ok = false; }
        float frac = vol_ok ? 4.18879f * radius * radius * radius / vol : 1.f;
        if (!(frac < 1.f)) frac = 1.f;
        if (!(frac > 1e-9f)) frac = 1e-9f;
        const float want = 1.5f * (float)nsample / frac;
        int wpl = want <= 2048.f ? 1 : (want <= 4096.f ? 2 : 4);
        int nblocks = (n + 2048 * wpl - 1) / (2048 * wpl);
        while ((long)g[0] * g[1] * g[2] * nblocks > kMaxKeys) {  // coarser cells stay correct (side only grows)
            int a = g[0] >= g[1] ? (g[0] >= g[2] ? 0 : 2) : (g[1] >= g[2] ? 1 : 2);
            if (g[a] > 1) g[a] = (g[a] + 1) / 2;
            else if (wpl < 4) { wpl *= 2; nblocks = (n + 2048 * wpl - 1) / (2048 * wpl); }
            else break;
        }
        float inv[3];
        for (int a = 0; a < 3; ++a) {
            const float side = g[a] > 1 ? fmaxf(h, ext[a] * 1.001f / (float)g[a]) : 1.f;
            inv[a] = g[a] > 1 ? 1.0f / side : 0.f;
        }
        hd.minx = mn[0]; hd.miny = mn[1]; hd.minz = mn[2];
        hd.invx = inv[0]; hd.invy = inv[1]; hd.invz = inv[2];
        hd.gx = g[0]; hd.gy = g[1]; hd.gz = g[2];
        hd.wpl = wpl; hd.nblocks = nblocks; hd.ncell = g[0] * g[1] * g[2];
        hd.pad0 = hd.pad1 = hd.pad2 = hd.pad3 = 0;
        *reinterpret_cast<Header *>(scratch) = hd;
    }
    __syncthreads();
    const Header H = hd;
    const int nkeys = H.ncell * H.nblocks;
    if (nkeys > kMaxKeys) return;  // cannot happen for n <= 8192 * kMaxKeys; the launcher bounds n
    int *part = cnt + nkeys + 1;
    for (int i = tid; i <= nkeys; i += kBuildT) cnt[i] = 0;
    __syncthreads();
    const int shift = 11 + (H.wpl == 1 ? 0 : (H.wpl == 2 ? 1 : 2));
    auto key_of = [&](int i, float &x, float &y, float &z) {
        x = xyz[3 * (size_t)i]; y = xyz[3 * (size_t)i + 1]; z = xyz[3 * (size_t)i + 2];
        int cx = (int)((x - H.minx) * H.invx), cy = (int)((y - H.miny) * H.invy), cz = (int)((z - H.minz) * H.invz);  // NaN -> 0
        cx = cx < 0 ? 0 : (cx >= H.gx ? H.gx - 1 : cx);
        cy = cy < 0 ? 0 : (cy >= H.gy ? H.gy - 1 : cy);
        cz = cz < 0 ? 0 : (cz >= H.gz ? H.gz - 1 : cz);
        return (i >> shift) * H.ncell + (cx * H.gy + cy) * H.gz + cz;
    };
    for (int i = tid; i < n; i += kBuildT) {
        float x, y, z;
        atomicAdd(&cnt[key_of(i, x, y, z)], 1);
    }
    __syncthreads();
    const int chunk = (nkeys + kBuildT - 1) / kBuildT;
    const int i0 = tid * chunk, i1 = (i0 + chunk) < nkeys ? (i0 + chunk) : nkeys;
    int sum = 0;
    for (int i = i0; i < i1; ++i) sum += cnt[i];
    part[tid] = sum;
    __syncthreads();
    if (tid < 64) {  // scan of the 1024 chunk totals by one wave: 16 per lane
        int loc[kBuildT / 64], s = 0;
        for (int j = 0; j < kBuildT / 64; ++j) { loc[j] = s; s += part[tid * (kBuildT / 64) + j]; }
        int incl = s;
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        const int base = incl - s;
        for (int j = 0; j < kBuildT / 64; ++j) part[tid * (kBuildT / 64) + j] = base + loc[j];
    }
    __syncthreads();
    int *__restrict__ start = reinterpret_cast<int *>(scratch) + kHeader;
    int run = part[tid];
    for (int i = i0; i < i1; ++i) {
        const int v = cnt[i];
        start[i] = run;
        cnt[i] = run;
        run += v;
    }
    if (tid == 0) start[nkeys] = n;
    __syncthreads();
    float4 *__restrict__ rec = reinterpret_cast<float4 *>(scratch + kHeader + kMaxKeys + 4);
    for (int i = tid; i < n; i += kBuildT) {
        float x, y, z;
        const int pos = atomicAdd(&cnt[key_of(i, x, y, z)], 1);
        rec[pos] = make_float4(x, y, z, __builtin_bit_cast(float, i));
    }
}

__device__ __forceinline__ int wave_excl_scan(int v, int lane) {
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    return incl - v;
}

__global__ void __launch_bounds__(kQueryT)
query_kernel(int n, int m, float radius2, int nsample, const float *__restrict__ new_xyz_all, const unsigned *__restrict__ scratch_all,
             int *__restrict__ idx_all) {
    // LDS: per wave 256 bitmap words (a lane owns 4 consecutive words).  The cell starts are read where they lie (global
    // memory, wave-uniform addresses: scalar loads through the constant cache) -- a copy of the table in LDS (48 KiB reserved
    // for the largest grid) held the kernel at 3 workgroups per compute unit, and its loads wait on L2 round trips.
    __shared__ unsigned bitmaps[kQueryWaves * 256];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned *__restrict__ scratch = scratch_all + (size_t)b * cloud_words(n);
    const Header H = *reinterpret_cast<const Header *>(scratch);
    const int *__restrict__ tab = reinterpret_cast<const int *>(scratch) + kHeader;
    unsigned *bm = bitmaps + w * 256;
    const float4 *__restrict__ rec = reinterpret_cast<const float4 *>(scratch + kHeader + kMaxKeys + 4);
    const int NB = 2048 * H.wpl;  // indices per block (<= 8192 = the bitmap)
    const int per_wg = kQueryWaves * 4;  // 4 centroids per wave
    for (int c = 0; c < 4; ++c) {
        const int s = blockIdx.x * per_wg + c * kQueryWaves + w;
        if (s >= m) break;  // wave-uniform
        const float *__restrict__ q = new_xyz_all + ((size_t)b * m + s) * 3;
        const float qx = q[0], qy = q[1], qz = q[2];
        int *__restrict__ row = idx_all + ((size_t)b * m + s) * nsample;
        // the centroid's cell, NOT clamped to the grid (it may lie outside the cloud's box); float -> int saturates, NaN -> 0
        int cx = (int)floorf((qx - H.minx) * H.invx), cy = (int)floorf((qy - H.miny) * H.invy), cz = (int)floorf((qz - H.minz) * H.invz);
        cx = cx < -2 ? -2 : (cx > H.gx + 1 ? H.gx + 1 : cx);
        cy = cy < -2 ? -2 : (cy > H.gy + 1 ? H.gy + 1 : cy);
        cz = cz < -2 ? -2 : (cz > H.gz + 1 ? H.gz + 1 : cz);
        const int x0 = cx - 1 < 0 ? 0 : cx - 1, x1 = cx + 1 >= H.gx ? H.gx - 1 : cx + 1;
        const int y0 = cy - 1 < 0 ? 0 : cy - 1, y1 = cy + 1 >= H.gy ? H.gy - 1 : cy + 1;
        const int z0 = cz - 1 < 0 ? 0 : cz - 1, z1 = cz + 1 >= H.gz ? H.gz - 1 : cz + 1;
        int cnt = 0, first = 0;
        if (x0 <= x1 && y0 <= y1 && z0 <= z1) {
            for (int blk = 0; blk < H.nblocks && cnt < nsample; ++blk) {
                *reinterpret_cast<uint4 *>(bm + 4 * lane) = make_uint4(0u, 0u, 0u, 0u);
                const int kb = blk * H.ncell;
                // the (up to) nine runs of this block -- one per (x, y) column, its z-neighbours being contiguous records --
                // are walked as ONE flattened candidate list: every step issues 64 independent record loads (walking the runs
                // one after the other made a step a dependent L2 round trip for a handful of candidates each)
                int rp0[9], rlen[9], ncand = 0;
#pragma unroll
                for (int r = 0; r < 9; ++r) {
                    const int ix = x0 + r / 3, iy = y0 + r % 3;
                    const bool in = ix <= x1 && iy <= y1;
                    const int key = kb + ((in ? ix : x0) * H.gy + (in ? iy : y0)) * H.gz;
                    const int p0 = tab[key + z0], p1 = tab[key + z1 + 1];
                    rp0[r] = p0;
                    rlen[r] = in ? p1 - p0 : 0;
                    ncand += rlen[r];
                }
                if (ncand > 384) {  // long runs (dense neighbourhoods): every run fills whole waves by itself, no flattening arithmetic
#pragma unroll
                    for (int r = 0; r < 9; ++r)
                        for (int p = rp0[r] + lane; p < rp0[r] + rlen[r]; p += 64) {
                            const float4 rc = rec[p];
                            if (sqdist(qx, qy, qz, rc.x, rc.y, rc.z) < radius2) {
                                const int k = __builtin_bit_cast(int, rc.w) & (NB - 1);
                                atomicOr(&bm[k >> 5], 1u << (k & 31));
                            }
                        }
                    ncand = 0;
                }
                // flat candidate f -> record: f + (start of its run - candidates in front of the run).  The run boundaries are
                // wave-uniform (scalar registers), so a lane adds the step of every boundary it lies behind: three VALU
                // instructions per run, half of what the select / subtract chain over (start, length) pairs cost -- the kernel
                // is VALU-bound on exactly this arithmetic since the cell table left LDS
                int cum[9], step[9];
                {
                    int c = 0, dprev = 0;
#pragma unroll
                    for (int r = 0; r < 9; ++r) {
                        const int d = rp0[r] - c;  // record index = f + d inside run r
                        cum[r] = c;
                        step[r] = d - dprev;
                        dprev = d;
                        c += rlen[r];
                    }
                }
                for (int f0 = 0; f0 < ncand; f0 += 64) {
                    const int f = f0 + lane;
                    int p = f + step[0];
#pragma unroll
                    for (int r = 1; r < 9; ++r) p += f >= cum[r] ? step[r] : 0;
                    if (f < ncand) {
                        const float4 r = rec[p];
                        if (sqdist(qx, qy, qz, r.x, r.y, r.z) < radius2) {
                            const int k = __builtin_bit_cast(int, r.w) & (NB - 1);
                            atomicOr(&bm[k >> 5], 1u << (k & 31));
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const uint4 wv = *reinterpret_cast<const uint4 *>(bm + 4 * lane);
                const unsigned wd[4] = {wv.x, wv.y, wv.z, wv.w};
                const int pc = __builtin_popcount(wd[0]) + __builtin_popcount(wd[1]) + __builtin_popcount(wd[2]) + __builtin_popcount(wd[3]);
                const uint64_t any = __ballot(pc > 0);
                if (any == 0) continue;
                if (cnt == 0) {  // the overall first hit: lowest set bit of the first lane that has one
                    int lowest = 0;
#pragma unroll
                    for (int j = 3; j >= 0; --j)
                        if (wd[j]) lowest = 32 * (4 * lane + j) + __builtin_ctz(wd[j]);
                    first = blk * NB + __shfl(lowest, __builtin_ctzll(any));
                }
                const int excl = wave_excl_scan(pc, lane);
                const int total = __shfl(excl + pc, 63);
                int pos = cnt + excl;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned bits = wd[j];
                    while (bits && pos < nsample) {
                        const int t = __builtin_ctz(bits);
                        bits &= bits - 1;
                        row[pos++] = blk * NB + 32 * (4 * lane + j) + t;
                    }
                }
                cnt += total;
                __builtin_amdgcn_wave_barrier();
            }
        }
        const int have = cnt < nsample ? cnt : nsample;
        const int fill = cnt > 0 ? first : 0;  // ball_query_gpu.cu:35-39: slots beyond the hits repeat the first hit; no hit -> 0
        for (int p = have + lane; p < nsample; p += 64) row[p] = fill;
    }
}

}  // namespace bqg

// PN2_ERANGE: shape not covered (caller runs the brute-force kernel)
int ball_query_grid_dispatch(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz, int *idx,
                             unsigned *scratch, size_t scratch_words, hipStream_t st) {
    using namespace bqg;
    if (b == 0 || m == 0) return PN2_OK;
    if (n < 2048 || n > (1 << 20) || !(radius > 0.f) || !scratch) return PN2_ERANGE;
    if ((long)((n + 8191) / 8192) > kMaxKeys) return PN2_ERANGE;
    if (scratch_words < (size_t)b * cloud_words(n)) return PN2_ESCRATCH;
    const float radius2 = radius * radius;  // fp32 product, ball_query_gpu.cu:23
    hipLaunchKernelGGL(build_kernel, dim3(b), dim3(kBuildT), (size_t)(kMaxKeys + 1 + kBuildT) * sizeof(int), st, n, radius, nsample, xyz, scratch);
    if (int rc = check_launch()) return rc;
    const dim3 grid((m + kQueryWaves * 4 - 1) / (kQueryWaves * 4), b);
    hipLaunchKernelGGL(query_kernel, grid, dim3(kQueryT), 0, st, n, m, radius2, nsample, new_xyz, scratch, idx);
    return check_launch();
}

}  // namespace pn2

extern "C" long pn2x_ball_query_grid_scratch_words(int b, int n) {
    if (b < 0 || n < 1) return -1;
    return (long)((size_t)b * pn2::bqg::cloud_words(n));
}

extern "C" int pn2x_ball_query_grid(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz, int *idx,
                                    void *scratch, long scratch_words, void *stream) {
    if (b < 0 || n < 1 || m < 0 || nsample < 1 || scratch_words < 0) return PN2_EINVAL;
    if (b == 0 || m == 0) return PN2_OK;
    if (!new_xyz || !xyz || !idx || !scratch) return PN2_ENULL;
    if (((uintptr_t)scratch) % 16) return PN2_EINVAL;
    return pn2::ball_query_grid_dispatch(b, n, m, radius, nsample, new_xyz, xyz, idx, (unsigned *)scratch, (size_t)scratch_words,
                                         (hipStream_t)stream);
}
