// train_wgrad.hip -- the weight gradients of EVERY plain linear layer of a training step in one grouped launch (gfx950).
//
// The layers of the training path that are not inside a fused [Conv 1x1 + BatchNorm + ReLU] stack -- layer 1 of every stack
// over the un-grouped points (reference pointnet_utils.py:399-403, :460-462, :504-506, :577-581 compute it over the grouped
// tensor), the rearrange linears (blocks.py:226-239), the 21-token tail (transformer.py:72-82, hand_network.py:141-147) --
// are library GEMMs.  Their weight gradients dW_p (N_p x K_p) = G_p^T X_p reduce over R_p = 672 ... 32768 rows into a small
// output: 15 launches of a training step, most of them latency-bound (8 - 39 us for < 2 GFLOP on a fraction of the chip),
// the two large ones split by the library into 512 workgroups of 192 x 48.  Nothing reads a weight gradient before the
// optimiser, so the autograd pass only RECORDS (G_p, X_p, dW_p) and one launch at the end of the pass computes them all:
//
//   wgm_partial   The work of all problems is ONE sequence of (128 x 128 output tile, 32-row chunk) steps, tile-major, cut into
//                 equal shares for a persistent grid that fills every CU slot exactly once ("stream-K": no tail round, no
//                 per-problem rounding).  A workgroup walks its share segment by segment (a segment = consecutive chunks of
//                 one tile), accumulates in registers (v_mfma_f32_32x32x2_f32; a wave owns a 64 x 64 block) and leaves the
//                 segment as a partial tile -- or as the result itself when it covered the tile's whole row range.  Both
//                 operands are row-major with the reduction along the rows, i.e. already k-major for the matrix cores: a
//                 chunk of G and of X is staged in LDS as it lies in memory (conflict-free ds_read_b32 by construction).
//                 Loads are 16-byte buffer loads with constant per-lane offsets against a descriptor that is re-based per
//                 chunk in SGPRs: rows beyond the segment and the columns beyond the problem in its last row read as zero
//                 through the hardware bounds check, columns beyond the problem elsewhere read in-range garbage that only
//                 reaches output columns / rows nobody stores -- no clamps, no selects, no address arithmetic on the VALU
//                 (next to a wave that streams fp32 MFMAs every other instruction costs an issue slot of the matrix pipe).
//   wgm_reduce    sums the partial tiles of the tiles that were cut, in workgroup order (deterministic, no atomics), any row
//                 stride of dW (a column block of a wider first-layer weight [feature | xyz | centre] is written in place).
//
// Bound: MFMA (fp32 dense peak 157.3 TFLOP/s); algorithmic work 2 R_p N_p K_p flop per problem.
#include "pn2_common.h"
#include "../../include/pn2_ext.h"

#ifndef WGM_DMA
#define WGM_DMA 1   // 1: chunks go from memory straight into LDS (buffer_load_dwordx4 ... lds: no staging registers, no ds_write pass)
#endif
#ifndef WGM_NBUF
#define WGM_NBUF (WGM_DMA ? 2 : 1)  // LDS chunk buffers: 1 = two barriers per chunk; 2 = one barrier per chunk, 64 KB per workgroup
#endif
#if WGM_DMA && WGM_NBUF != 2
#error "WGM_DMA needs two LDS chunk buffers"
#endif

namespace pn2 {
namespace wgm {

#ifdef WGM_PROBE_NOBARRIER  // timing probe (wrong results)
#define WGM_SYNC() do {} while (0)
#else
#define WGM_SYNC() __syncthreads()
#endif

constexpr int kT = 256;
constexpr int TN = 128, TK = 128;  // output tile: rows (channels of G) x columns (channels of X)
#ifndef WGM_RC
#define WGM_RC 32
#endif
constexpr int RC = WGM_RC;         // rows per reduction chunk
constexpr int kMaxP = 40;          // problems per launch (kernel-argument table: 72 bytes each)
constexpr int NBUF = WGM_NBUF;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

struct Prob {
    const float *G, *X;
    float *dW;
    int ldg, ldx, lddw;
    int R, N, K;
    int tk, tiles;   // tiles along K, tiles in all
    int chunks;      // ceil(R / RC)
    int pos0;        // first step of this problem in the launch's (tile, chunk) sequence
    int tile0;       // global index of its first tile
    int red_start;   // first workgroup of this problem in the reduction launch
};
struct Args {
    Prob p[kMaxP];
    int n;
    int total;       // steps of the whole launch
    int per_wg;      // steps per workgroup
    float *partial;  // slots of TN x TK floats; workgroup w leaves a cut segment of global tile t in slot w + t
};

__device__ __forceinline__ rsrc_t make_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 ld_b128(rsrc_t r, unsigned voff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0));
}

__device__ __forceinline__ unsigned uniform(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ const char *uniform_ptr(const char *p) {
    const uint64_t a = (uint64_t)p;
    return reinterpret_cast<const char *>(((uint64_t)uniform((unsigned)(a >> 32)) << 32) | uniform((unsigned)a));
}

// chunks [c0, c1) of tile `tile` of problem p, by workgroup w.  FULL: every 32 x 32 block of the tile lies (at least partly) inside
// the problem -- the instruction stream then has no per-block conditions (two instantiations, no merged register state)
template <bool FULL>
__device__ __forceinline__ void segment(const Prob &p, float *__restrict__ partial, int w, int tile, int c0, int c1, float *lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = (int)uniform((unsigned)tid >> 6);
    const int n0 = (tile / p.tk) * TN, k0 = (tile % p.tk) * TK;
    const int r_begin = c0 * RC;
    const int r_end = c1 * RC < p.R ? c1 * RC : p.R;
    const int nc = c1 - c0;
    // the segment's operands as byte ranges [base, base + bytes): up to the last element of the problem in the segment's last row
    // (everything a descriptor is built from is pinned to SGPRs: a descriptor the compiler believes divergent costs a waterfall
    // loop around every load)
    const char *gb = uniform_ptr(reinterpret_cast<const char *>(p.G + (size_t)r_begin * p.ldg + n0));
    const char *xb = uniform_ptr(reinterpret_cast<const char *>(p.X + (size_t)r_begin * p.ldx + k0));
    const unsigned gbytes = uniform(4u * ((unsigned)(r_end - 1 - r_begin) * p.ldg + (unsigned)(p.N - n0)));
    const unsigned xbytes = uniform(4u * ((unsigned)(r_end - 1 - r_begin) * p.ldx + (unsigned)(p.K - k0)));
    const unsigned gstep = uniform(4u * RC * p.ldg), xstep = uniform(4u * RC * p.ldx);
    const int q = tid & 31, rr0 = tid >> 5;  // this thread's column quad and first row of a chunk (rows rr0 + 8 i)
    constexpr int LI = RC / 8;  // 16-byte loads per thread, operand and chunk
    unsigned vg[LI], vx[LI];
#pragma unroll
    for (int i = 0; i < LI; ++i) {
        // a quad wholly beyond the problem's columns is never fetched (offset beyond every descriptor: reads 0)
        vg[i] = n0 + 4 * q < p.N ? 4u * ((unsigned)(rr0 + 8 * i) * p.ldg + 4 * q) : 0x7ffffff0u;
        vx[i] = k0 + 4 * q < p.K ? 4u * ((unsigned)(rr0 + 8 * i) * p.ldx + 4 * q) : 0x7ffffff0u;
    }
#if !WGM_DMA
    f32x4 pg[LI], px[LI];
    auto prefetch = [&](int ci) {
#ifdef WGM_PROBE_NOLOAD  // timing probe (wrong results): no global loads, no LDS writes
        return;
#endif
#ifdef WGM_PROBE_SAMECHUNK  // timing probe (wrong results): every chunk re-reads the segment's first one -- the rate without memory traffic
        ci = 0;
#endif
        const unsigned go = uniform((unsigned)ci) * gstep, xo = uniform((unsigned)ci) * xstep;
        // (ci < nc: the chunk starts inside the segment, so bytes > offset -- no clamp: hipcc turns `a > b ? a - b : 0` into a
        // saturating VALU subtract, and a descriptor with a VGPR component is loaded through a waterfall loop)
        const rsrc_t rg = make_rsrc(gb + go, gbytes - go);
        const rsrc_t rx = make_rsrc(xb + xo, xbytes - xo);
#pragma unroll
        for (int i = 0; i < LI; ++i) pg[i] = ld_b128(rg, vg[i]);
#pragma unroll
        for (int i = 0; i < LI; ++i) px[i] = ld_b128(rx, vx[i]);
    };
    auto commit = [&](float *Gd, float *Xd) {
#ifdef WGM_PROBE_NOLOAD
        return;
#endif
#pragma unroll
        for (int i = 0; i < LI; ++i) *reinterpret_cast<f32x4 *>(Gd + (rr0 + 8 * i) * TN + 4 * q) = pg[i];
#pragma unroll
        for (int i = 0; i < LI; ++i) *reinterpret_cast<f32x4 *>(Xd + (rr0 + 8 * i) * TK + 4 * q) = px[i];
    };
#endif
    // wave (wn, wk) owns the 32 x 32 blocks {wn, wn + 2} x {wk, wk + 2} of the tile's 4 x 4 (its two operand fragments of a step are
    // 64 floats apart and consecutive steps 256: every LDS read of a chunk is one base register + an immediate, ds_read2st64_b32);
    // blocks that lie wholly outside the problem (edge tiles: K = 131 has a 3-column second tile) are skipped, wave-uniformly
    const int wn = wave & 1, wk = wave >> 1, l31 = lane & 31, kh = lane >> 5;
    const bool va0 = n0 + 32 * wn < p.N, va1 = n0 + 32 * wn + 64 < p.N;
    const bool vb0 = k0 + 32 * wk < p.K, vb1 = k0 + 32 * wk + 64 < p.K;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int goff = 32 * wn + l31 + kh * TN, xoff = 32 * wk + l31 + kh * TK;
    auto mma = [&](const float *Gd, const float *Xd) {
        const float *ga = Gd + goff, *xa = Xd + xoff;
        if constexpr (FULL) {
            // the operands of step s + 1 are requested BEFORE the four instructions of step s (sched_barrier: left alone, hipcc
            // sinks every ds_read in front of its consumers and the wave waits an LDS round trip per four instructions)
            float a0 = ga[0], a1 = ga[64], b0 = xa[0], b1 = xa[64];
#pragma unroll
            for (int s = 0; s < RC / 2; ++s) {
                float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
                if (s + 1 < RC / 2) {
                    na0 = ga[2 * (s + 1) * TN]; na1 = ga[2 * (s + 1) * TN + 64];
                    nb0 = xa[2 * (s + 1) * TK]; nb1 = xa[2 * (s + 1) * TK + 64];
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
            }
        } else if (va0 && vb0) {
#pragma unroll 2
            for (int s = 0; s < RC / 2; ++s) {
                const float a0 = ga[2 * s * TN], a1 = ga[2 * s * TN + 64];
                const float b0 = xa[2 * s * TK], b1 = xa[2 * s * TK + 64];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                if (vb1) acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                if (va1) acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                if (va1 && vb1) acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
    };
#if WGM_DMA
    // the chunk lands in LDS in lane order: lane l of wave w fetches quad (l & 31) of row 2 w + (l >> 5) + 8 i, i.e. the wave's 64 quads
    // are two consecutive 512-byte rows of the tile -- exactly base + 16 l, the only layout LDS-DMA can write.  Out-of-range lanes
    // write zeros.  Chunk ci + 1 is requested right behind the barrier that publishes chunk ci and has a whole chunk of matrix
    // instructions to arrive (hipcc waits vmcnt(0) in front of the next barrier).
    typedef __attribute__((address_space(3))) void *lds_ptr;
    auto issue = [&](int ci, float *Gd, float *Xd) {
#ifdef WGM_PROBE_NOLOAD
        return;
#endif
        const unsigned go = uniform((unsigned)ci) * gstep, xo = uniform((unsigned)ci) * xstep;
        const rsrc_t rg = make_rsrc(gb + go, gbytes - go);
        const rsrc_t rx = make_rsrc(xb + xo, xbytes - xo);
#pragma unroll
        for (int i = 0; i < LI; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (lds_ptr)(Gd + (2 * wave + 8 * i) * TN), 16, (int)vg[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < LI; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(Xd + (2 * wave + 8 * i) * TK), 16, (int)vx[i], 0, 0, 0);
    };
    issue(0, lds, lds + RC * TN);
    for (int ci = 0; ci < nc; ++ci) {
        float *cur = lds + (ci & 1) * (RC * (TN + TK)), *nxt = lds + ((ci + 1) & 1) * (RC * (TN + TK));
        WGM_SYNC();  // chunk ci is in LDS for every wave; nobody reads the other buffer any more
        if (ci + 1 < nc) issue(ci + 1, nxt, nxt + RC * TN);
        mma(cur, cur + RC * TN);
    }
    WGM_SYNC();  // (the next segment's first chunk goes into buffer 0)
#else
    if constexpr (NBUF == 1) {
        float *Gd = lds, *Xd = lds + RC * TN;
        prefetch(0);
        for (int ci = 0; ci < nc; ++ci) {
            commit(Gd, Xd);
            WGM_SYNC();
            if (ci + 1 < nc) prefetch(ci + 1);
            mma(Gd, Xd);
            WGM_SYNC();
        }
    } else {
        prefetch(0);
        commit(lds, lds + RC * TN);
        WGM_SYNC();
        for (int ci = 0; ci < nc; ++ci) {
            float *cur = lds + (ci & 1) * (RC * (TN + TK)), *nxt = lds + ((ci + 1) & 1) * (RC * (TN + TK));
            if (ci + 1 < nc) prefetch(ci + 1);
            mma(cur, cur + RC * TN);
            if (ci + 1 < nc) commit(nxt, nxt + RC * TN);
            WGM_SYNC();
        }
    }
#endif
    // accumulator element r of block (i, j): row 32 wn + 64 i + (r & 3) + 8 (r >> 2) + 4 kh, column 32 wk + 64 j + l31
    if (c0 == 0 && c1 == p.chunks) {  // the whole row range: this IS the result
        // 4-byte buffer stores against [first element of the tile, last element of the problem]: rows beyond N fall outside the
        // descriptor, lanes whose column is beyond K get an offset outside it -- one 32-bit offset per store, no 64-bit addresses
        const rsrc_t rd = make_rsrc(uniform_ptr(reinterpret_cast<const char *>(p.dW + (size_t)n0 * p.lddw + k0)),
                                    uniform(4u * ((unsigned)(p.N - n0 - 1) * p.lddw + (unsigned)(p.K - k0))));
        const unsigned ld4 = uniform(4u * p.lddw);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int kc = 32 * wk + 64 * j + l31;
            const unsigned vcol = k0 + kc < p.K ? 4u * kc + (unsigned)(32 * wn + 4 * kh) * ld4 : 0x7ffffff0u;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                // (whole-vector bit cast: __builtin_bit_cast on a vector ELEMENT reads element 0 whatever the subscript with this hipcc)
                const u32x16 u = __builtin_bit_cast(u32x16, acc[i][j]);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(u[r], rd, (int)(vcol + (unsigned)(64 * i + (r & 3) + 8 * (r >> 2)) * ld4), 0, 0);
            }
        }
    } else {
        const rsrc_t ro = make_rsrc(uniform_ptr(reinterpret_cast<const char *>(partial + (size_t)(w + p.tile0 + tile) * (size_t)(TN * TK))),
                                    4u * TN * TK);
        const unsigned vo = 4u * ((unsigned)(32 * wn + 4 * kh) * TK + 32 * wk + l31);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (!FULL && !((i ? va1 : va0) && (j ? vb1 : vb0))) continue;
                const u32x16 u = __builtin_bit_cast(u32x16, acc[i][j]);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(u[r], ro, (int)(vo + 4u * ((64 * i + (r & 3) + 8 * (r >> 2)) * TK + 64 * j)), 0, 0);
            }
    }
}

// (occupancy: three workgroups per CU with one LDS buffer, two with two -- the register budget is stated, hipcc otherwise spends
// 180 VGPRs on this kernel without needing them)
__global__ void __launch_bounds__(kT) __attribute__((amdgpu_waves_per_eu(NBUF == 1 ? 3 : 2)))
wgm_partial_kernel(Args a) {
    __shared__ __attribute__((aligned(16))) float lds[NBUF * RC * (TN + TK)];
    const int w = blockIdx.x;
    int pos = w * a.per_wg;
    const int end = (pos + a.per_wg) < a.total ? (pos + a.per_wg) : a.total;
    int t = 0;
    while (pos < end) {
        while (t + 1 < a.n && a.p[t + 1].pos0 <= pos) ++t;
        const Prob &p = a.p[t];
        const int rel = pos - p.pos0;
        const int tile = rel / p.chunks, c0 = rel - tile * p.chunks;
        int c1 = c0 + (end - pos);
        c1 = c1 < p.chunks ? c1 : p.chunks;
        // (a tile is FULL when its last 32 x 32 block row / column starts inside the problem)
        const int n0 = (tile / p.tk) * TN, k0 = (tile % p.tk) * TK;
        if (n0 + 96 < p.N && k0 + 96 < p.K) segment<true>(p, a.partial, w, tile, c0, c1, lds);
        else segment<false>(p, a.partial, w, tile, c0, c1, lds);
        pos += c1 - c0;
    }
}

// dW[n][k] = sum of the partial tiles of its tile, in workgroup order; a thread owns four consecutive columns.  Tiles that one
// workgroup covered alone were written by it.
__global__ void __launch_bounds__(kT)
wgm_reduce_kernel(Args a) {
    int t = 0;
    while (t + 1 < a.n && a.p[t + 1].red_start <= (int)blockIdx.x) ++t;
    const Prob &p = a.p[t];
    const int kq = (p.K + 3) / 4;
    const long e = (long)((int)blockIdx.x - p.red_start) * kT + threadIdx.x;
    if (e >= (long)p.N * kq) return;
    const int n = (int)(e / kq), k = 4 * (int)(e % kq);
    const int tile = (n / TN) * p.tk + k / TK;
    const int s0 = p.pos0 + tile * p.chunks, s1 = s0 + p.chunks;
    const int w0 = s0 / a.per_wg, w1 = (s1 - 1) / a.per_wg;
    if (w0 == w1) return;
    const float *src = a.partial + (size_t)(p.tile0 + tile) * (TN * TK) + (n % TN) * TK + (k % TK);
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, u = s;
    int w = w0;
    for (; w + 1 <= w1; w += 2) {
        const f32x4 v0 = *reinterpret_cast<const f32x4 *>(src + (size_t)w * (TN * TK));
        const f32x4 v1 = *reinterpret_cast<const f32x4 *>(src + (size_t)(w + 1) * (TN * TK));
        s += v0;
        u += v1;
    }
    if (w <= w1) s += *reinterpret_cast<const f32x4 *>(src + (size_t)w * (TN * TK));
    const f32x4 r = s + u;
    float *dst = p.dW + (size_t)n * p.lddw + k;
    dst[0] = r.x;
    if (k + 1 < p.K) dst[1] = r.y;
    if (k + 2 < p.K) dst[2] = r.z;
    if (k + 3 < p.K) dst[3] = r.w;
}

struct Plan {
    int order[kMaxP];
    int tk[kMaxP], tiles[kMaxP], chunks[kMaxP];
    int total, per_wg, n_wg, total_tiles;
    long floats;
};

static int resident_workgroups() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 2 * num_compute_units();
    if (cached[dev] == 0) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, wgm_partial_kernel, kT, 0) != hipSuccess || per_cu < 1) per_cu = 2;
        cached[dev] = per_cu * num_compute_units();
    }
    return cached[dev];
}

// the launch's step sequence: problems by rows (descending: the long row ranges first), tile-major inside a problem; an equal
// share of it per resident workgroup (at least 4 steps each)
static int plan(int count, const int *rows, const int *n, const int *k, const int *ldg, const int *ldx, Plan &pl) {
    long total = 0, tiles = 0;
    for (int i = 0; i < count; ++i) {
        if (rows[i] < 1 || n[i] < 1 || k[i] < 1) return PN2_EINVAL;
        pl.tk[i] = (k[i] + TK - 1) / TK;
        pl.tiles[i] = ((n[i] + TN - 1) / TN) * pl.tk[i];
        pl.chunks[i] = (rows[i] + RC - 1) / RC;
        total += (long)pl.tiles[i] * pl.chunks[i];
        tiles += pl.tiles[i];
        pl.order[i] = i;
    }
    if (total > 0x3fffffffL) return PN2_EINVAL;
    for (int i = 1; i < count; ++i)
        for (int j = i; j > 0 && rows[pl.order[j]] > rows[pl.order[j - 1]]; --j) {
            const int tmp = pl.order[j]; pl.order[j] = pl.order[j - 1]; pl.order[j - 1] = tmp;
        }
    long wgs = resident_workgroups();
    if (wgs > (total + 3) / 4) wgs = (total + 3) / 4;
    if (wgs < 1) wgs = 1;
    pl.per_wg = (int)((total + wgs - 1) / wgs);
    pl.n_wg = (int)((total + pl.per_wg - 1) / pl.per_wg);
    pl.total = (int)total;
    pl.total_tiles = (int)tiles;
    pl.floats = (long)(pl.n_wg + tiles) * (TN * TK);
    // a segment's operand range must fit a 32-bit buffer descriptor
    for (int i = 0; i < count; ++i) {
        const long span = (long)(pl.per_wg < pl.chunks[i] ? pl.per_wg : pl.chunks[i]) * RC;
        if (ldg && span * ldg[i] * 4L >= 0x7fff0000L) return PN2_EINVAL;
        if (ldx && span * ldx[i] * 4L >= 0x7fff0000L) return PN2_EINVAL;
    }
    return PN2_OK;
}

}  // namespace wgm
}  // namespace pn2

using namespace pn2;
using namespace pn2::wgm;

extern "C" int pn2x_wgrad_multi_max(void) { return kMaxP; }

extern "C" long pn2x_wgrad_multi_scratch_floats(int count, const int *rows, const int *n, const int *k) {
    if (count < 1 || count > kMaxP || !rows || !n || !k) return -1;
    Plan pl;
    if (plan(count, rows, n, k, nullptr, nullptr, pl) != PN2_OK) return -1;
    return pl.floats;
}

extern "C" int pn2x_wgrad_multi(int count, const float *const *g, const int *ldg, const float *const *x, const int *ldx, const int *rows,
                                const int *n, const int *k, float *const *dw, const int *lddw, float *scratch, long scratch_floats,
                                void *stream) {
    if (count == 0) return PN2_OK;
    if (count < 0 || count > kMaxP) return PN2_EINVAL;
    if (!g || !ldg || !x || !ldx || !rows || !n || !k || !dw || !lddw) return PN2_ENULL;
    Plan pl;
    if (int rc = plan(count, rows, n, k, ldg, ldx, pl)) return rc;
    if (!scratch || scratch_floats < pl.floats || (uintptr_t)scratch % 16) return PN2_ESCRATCH;
    Args a;
    a.n = count;
    a.partial = scratch;
    a.total = pl.total;
    a.per_wg = pl.per_wg;
    int pos = 0, tile0 = 0, rblocks = 0;
    for (int s = 0; s < count; ++s) {
        const int i = pl.order[s];
        if (!g[i] || !x[i] || !dw[i]) return PN2_ENULL;
        if (ldg[i] < n[i] || ldx[i] < k[i] || lddw[i] < k[i]) return PN2_EINVAL;
        if (((uintptr_t)g[i] | (uintptr_t)x[i] | (uintptr_t)dw[i]) % 4) return PN2_EINVAL;
        Prob &p = a.p[s];
        p.G = g[i]; p.X = x[i]; p.dW = dw[i];
        p.ldg = ldg[i]; p.ldx = ldx[i]; p.lddw = lddw[i];
        p.R = rows[i]; p.N = n[i]; p.K = k[i];
        p.tk = pl.tk[i]; p.tiles = pl.tiles[i]; p.chunks = pl.chunks[i];
        p.pos0 = pos; p.tile0 = tile0;
        pos += p.tiles * p.chunks;
        tile0 += p.tiles;
        p.red_start = rblocks;
        if (p.chunks > 1) rblocks += (int)(((long)p.N * ((p.K + 3) / 4) + kT - 1) / kT);  // (one chunk is never cut)
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(wgm_partial_kernel, dim3(pl.n_wg), dim3(kT), 0, st, a);
    if (int rc = check_launch()) return rc;
    if (rblocks > 0) {
        hipLaunchKernelGGL(wgm_reduce_kernel, dim3(rblocks), dim3(kT), 0, st, a);
        return check_launch();
    }
    return PN2_OK;
}
