// train_fwd.hip -- the forward of a fused [Conv 1x1 + BatchNorm + ReLU] layer for 64- / 128- / 256-channel inputs (gfx950).
//
// Same mathematics and interface as train_gemm.hip's tg_fwd (Y_i = relu(BN_{i-1}(Y_{i-1})) W_i^T, statistics of Y_i in the
// epilogue, the consumer finalises its producer's running statistics), different schedule.  tg_fwd walks the reduction in
// chunks of 32 through one LDS buffer (two barriers per chunk), rebuilds the normalised A operand once per 64/128-column
// block of the output and takes 128-row tiles -- 336 of them for the 43008-row keypoint-query layers, on 256 CUs.  Here:
//   * W_i (this workgroup's 64 / 128 / 192 output channels x all K input channels) sits in LDS for the whole kernel,
//     k-minor with an odd row stride (both MFMA operands are then conflict-free ds_read_b32);
//   * a persistent workgroup takes 64-row tiles: the normalised + ReLU'd tile H (64 x K) is built ONCE in LDS from operands
//     fetched a tile ahead, then every wave runs the whole reduction for its own 32 x 32 output block (2 row blocks x
//     2 / 4 / 6 column blocks = 4 / 8 / 12 waves) without a barrier in between: two barriers per TILE;
//   * outputs go from the accumulators to HBM (two 128-byte segments per store instruction), their per-channel
//     sum / sum of squares stay in two registers per lane for all tiles of the workgroup.
// v_mfma_f32_32x32x2_f32.  Layers with 32 input channels keep tg_fwd (HBM-bound there already); 256 input channels (the two small
// feature-propagation stacks) take 64 output columns per workgroup so that W_i's slice still fits next to the operand tile.
#include "pn2_common.h"
#include "../../include/pn2_ext.h"

namespace pn2 {
namespace tgf {

constexpr int BM = 64;
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float relu_nan(float h) { return !(h <= 0.f) ? h : 0.f; }  // propagates NaN like torch

struct FwdArgs {
    long R;
    int N;                     // all output channels (row stride of the statistics)
    const float *X; int ldx;   // pre-activations of the previous layer (R x K)
    const float *W; int ldw;   // (N x K)
    float *Y; int ldy;         // (R x N)
    const double *sums_in;     // previous layer's forward sums
    const float *gamma, *beta, *conv_bias;
    float eps, momentum;
    float *running_mean, *running_var;
    long long *nbt;
    float *save_mean, *save_invstd;  // written by workgroup (0, 0)
    double *sums_out;                // this layer's forward sums (may be null)
};

template <int KB, int NBLK>
struct Plan {
    static constexpr int K = 32 * KB, NW = 32 * NBLK, WAVES = 2 * NBLK, T = 64 * WAVES;
    static constexpr int LDK = K + 1;
    static constexpr int GROUPS = WAVES >= 8 ? 1 : 8 / WAVES;  // 8-row groups of the tile a wave commits
    static constexpr int lds_floats = 4 * K + BM * LDK + NW * LDK + WAVES * 64;
};

// The tile loop of one problem, run by workgroup `wg` of the `nwg` workgroups (per column group) assigned to it: a whole launch
// (tgf_kernel), or one share of a launch that serves two problems of the same layer shape at once (tgf_pair_kernel: the two
// neighbourhood sizes of a keypoint-query module -- same widths, different weights and row counts).
template <int KB, int NBLK>
__device__ __forceinline__ void tgf_body(const FwdArgs &a, const int wg, const int nwg) {
    using P = Plan<KB, NBLK>;
    constexpr int K = P::K, NW = P::NW, LDK = P::LDK, T = P::T;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *cst = lds;              // [4][K]: mean, invstd, gamma, beta of the input channels
    float *Hs = cst + 4 * K;       // [BM][LDK]
    float *Ws = Hs + BM * LDK;     // [NW][LDK]
    float *red = Ws + NW * LDK;    // [WAVES][2][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
    const int n0 = blockIdx.y * NW;
    const bool first = wg == 0 && blockIdx.y == 0;
    const long tiles = (a.R + BM - 1) / BM;
    // tile operands: a wave fetches 8 rows x 8 float4 (128 contiguous bytes per row) per step; the first 8 waves (all of
    // them where there are fewer) cover the tile's eight 8-row groups
    const int arow = lane >> 3, aq = lane & 7;
    const bool loader = P::WAVES <= 8 ? true : wave < 8;  // (a compile-time `true` where every wave loads: no branch around the prefetch)
    float4 px[P::GROUPS * KB];
    auto prefetch = [&](long tile) {
        if (!loader) return;
#pragma unroll
        for (int g = 0; g < P::GROUPS; ++g) {
            long row = tile * BM + 8 * (wave + g * P::WAVES) + arow;
            row = row < a.R ? row : a.R - 1;  // unconditional loads; the commit zeroes what lies beyond the problem
#pragma unroll
            for (int i = 0; i < KB; ++i) px[g * KB + i] = *reinterpret_cast<const float4 *>(a.X + row * a.ldx + 4 * (8 * i + aq));
        }
    };
    long tile = wg;
    // (the first tile's operands are requested after the constants' loads, below: loads return in order, and the fp64 sums the
    // constants are derived from would otherwise wait for every HBM row queued in front of them -- round 5, train_bwd.hip)
    // W_i's slice: requested now, stored to LDS after the constants (one memory round trip for both instead of two in a row)
    constexpr int WCNT = NW * (K / 4) / T;
    static_assert(NW * (K / 4) % T == 0, "W_i slice splits evenly over the workgroup");
    float4 wreg[WCNT];
#pragma unroll
    for (int i = 0; i < WCNT; ++i) {
        const int e = tid + i * T, n = e / (K / 4), q = e % (K / 4);
        wreg[i] = *reinterpret_cast<const float4 *>(a.W + (size_t)(n0 + n) * a.ldw + 4 * q);
    }
    for (int k = tid; k < K; k += T) {
        double s1 = 0.0, s2 = 0.0;
        for (int r = 0; r < kBnRep; ++r) {
            s1 += a.sums_in[(size_t)r * 2 * K + k];
            s2 += a.sums_in[(size_t)r * 2 * K + K + k];
        }
        const double m = s1 / (double)a.R;
        double v = s2 / (double)a.R - m * m;
        v = v > 0.0 ? v : 0.0;
        const float mean = (float)m, invstd = (float)(1.0 / sqrt(v + (double)a.eps));  // same arithmetic as train_ops.hip's bn_consts
        cst[k] = mean; cst[K + k] = invstd; cst[2 * K + k] = a.gamma[k]; cst[3 * K + k] = a.beta[k];
        if (first) {
            a.save_mean[k] = mean;
            a.save_invstd[k] = invstd;
            if (a.running_mean) {  // torch: running = (1 - m) running + m batch, the variance unbiased
                const float bm = mean + (a.conv_bias ? a.conv_bias[k] : 0.f);  // the GEMM output excludes the conv bias (cancels in BN)
                const float bv = (float)(a.R > 1 ? v * ((double)a.R / (double)(a.R - 1)) : v);
                a.running_mean[k] = (1.f - a.momentum) * a.running_mean[k] + a.momentum * bm;
                a.running_var[k] = (1.f - a.momentum) * a.running_var[k] + a.momentum * bv;
            }
        }
    }
    if (first && tid == 0 && a.nbt) *a.nbt += 1;
    if (tile < tiles) prefetch(tile);
#pragma unroll
    for (int i = 0; i < WCNT; ++i) {
        const int e = tid + i * T, n = e / (K / 4), q = e % (K / 4);
        float *dst = Ws + n * LDK + 4 * q;
        dst[0] = wreg[i].x; dst[1] = wreg[i].y; dst[2] = wreg[i].z; dst[3] = wreg[i].w;
    }
    __syncthreads();

    auto commit = [&](long tile) {
        if (!loader) return;
#pragma unroll
        for (int g = 0; g < P::GROUPS; ++g) {
            const int r = 8 * (wave + g * P::WAVES) + arow;
            const bool v = tile * BM + r < a.R;
#pragma unroll
            for (int i = 0; i < KB; ++i) {
                const int c = 4 * (8 * i + aq);
                const float4 mean = *reinterpret_cast<const float4 *>(cst + c), is = *reinterpret_cast<const float4 *>(cst + K + c);
                const float4 ga = *reinterpret_cast<const float4 *>(cst + 2 * K + c), be = *reinterpret_cast<const float4 *>(cst + 3 * K + c);
                const float4 x = px[g * KB + i];
                float *dst = Hs + r * LDK + c;
                const float h0 = relu_nan(((x.x - mean.x) * is.x) * ga.x + be.x);  // torch's evaluation order
                const float h1 = relu_nan(((x.y - mean.y) * is.y) * ga.y + be.y);
                const float h2 = relu_nan(((x.z - mean.z) * is.z) * ga.z + be.z);
                const float h3 = relu_nan(((x.w - mean.w) * is.w) * ga.w + be.w);
                dst[0] = v ? h0 : 0.f; dst[1] = v ? h1 : 0.f; dst[2] = v ? h2 : 0.f; dst[3] = v ? h3 : 0.f;
            }
        }
    };

    const int rblk = wave / NBLK, nb = wave % NBLK;
    const float *ap = Hs + (rblk * 32 + l31) * LDK + kh;
    const float *bp = Ws + (nb * 32 + l31) * LDK + kh;
    float cs = 0.f, cq = 0.f;
    // the outputs leave through a buffer descriptor over the rows of the problem: rows beyond R are dropped by the hardware, every
    // tile issues the same sixteen stores -- no full / ragged branch (see the note on s_waitcnt at the tile step below)
    typedef __amdgpu_buffer_rsrc_t rsrc_t;
    const size_t ybytes = ((size_t)(a.R - 1) * a.ldy + (size_t)a.N) * 4;
    const rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.Y, 0, ybytes > 0x7ffffff0u ? 0x7ffffff0 : (int)ybytes, 0x00020000);
    const unsigned yrow = (unsigned)__builtin_amdgcn_readfirstlane(4 * a.ldy);
    // One tile.  hipcc's s_waitcnt insertion merges the memory operations in flight over every path into a block: with a loop whose
    // first trip arrives with only the prefetched operands outstanding and whose later trips arrive with [operands, then 16 output
    // stores], it waited at the top of EVERY trip until all but one operation had completed -- i.e. for the previous tile's stores
    // to be acknowledged, with the matrix pipe idle.  Hence: no conditional loads or stores inside the step (the prefetch of a tile
    // beyond the last re-reads the last row, the stores are bounds-checked), and the first tile is peeled off the loop, so that both
    // ways into the loop carry the same operations in the same order and the wait in front of the commit counts the stores out.
    auto step = [&](long tile) __attribute__((always_inline)) -> long {
        commit(tile);
        __syncthreads();
        const long ntile = tile + nwg;
        prefetch(ntile);  // nothing else of this tile reads global memory (rows are clamped to the problem's last one)
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        {
            // operands of the next U matrix instructions are requested from LDS while the current U issue (left to itself the
            // compiler waits for each pair of ds_reads right in front of the two instructions that use them)
            constexpr int U = 8, NBATCH = (K / 2) / U;
            static_assert(NBATCH % 2 == 0, "batches come in pairs");
            float a0[U], b0[U], a1[U], b1[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { a0[u] = ap[2 * u]; b0[u] = bp[2 * u]; }
#pragma unroll
            for (int bt = 0; bt < NBATCH; bt += 2) {
#pragma unroll
                for (int u = 0; u < U; ++u) { a1[u] = ap[2 * ((bt + 1) * U + u)]; b1[u] = bp[2 * ((bt + 1) * U + u)]; }
                __builtin_amdgcn_sched_barrier(0);  // (the scheduler would sink the reads back in front of their uses)
#pragma unroll
                for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b0[u], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (bt + 2 < NBATCH) {
#pragma unroll
                    for (int u = 0; u < U; ++u) { a0[u] = ap[2 * ((bt + 2) * U + u)]; b0[u] = bp[2 * ((bt + 2) * U + u)]; }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], b1[u], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();  // every wave is done with Hs: the next commit may overwrite it while the stores below drain
        {
            const unsigned row0 = (unsigned)(tile * BM + rblk * 32 + 4 * kh);  // (rows < 2^31 / ldy: launcher)
            const unsigned base = row0 * yrow + 4u * (unsigned)(n0 + nb * 32 + l31);
            typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
            const u32x16 bits = __builtin_bit_cast(u32x16, acc);  // (whole-vector cast: a cast of a vector ELEMENT reads element 0)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                __builtin_amdgcn_raw_buffer_store_b32(bits[r], ry, (int)(base + (unsigned)((r & 3) + 8 * (r >> 2)) * yrow), 0, 0);
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {  // rows beyond R are exact zeros
                s += acc[r];
                q += acc[r] * acc[r];
            }
            cs += s;
            cq += q;
        }
        return ntile;
    };
    if (tile < tiles) {
        tile = step(tile);
        while (tile < tiles) tile = step(tile);
    }
    if (a.sums_out) {
        const float s = cs + __shfl_xor(cs, 32), q = cq + __shfl_xor(cq, 32);  // lanes l and l ^ 32 hold the same column
        if (lane < 32) {
            red[wave * 64 + lane] = s;
            red[wave * 64 + 32 + lane] = q;
        }
        __syncthreads();
        if (tid < NW) {
            const int b = tid >> 5, c = tid & 31;  // column block b: waves b (rows 0..31) and NBLK + b (rows 32..63)
            const double sd = (double)red[b * 64 + c] + (double)red[(NBLK + b) * 64 + c];
            const double qd = (double)red[b * 64 + 32 + c] + (double)red[(NBLK + b) * 64 + 32 + c];
            double *dst = a.sums_out + (size_t)((wg + blockIdx.y) % kBnRep) * 2 * a.N;
            unsafeAtomicAdd(dst + n0 + tid, sd);
            unsafeAtomicAdd(dst + a.N + n0 + tid, qd);
        }
    }
}

template <int KB, int NBLK>
__global__ void __launch_bounds__(128 * NBLK)
tgf_kernel(FwdArgs a) {
    tgf_body<KB, NBLK>(a, (int)blockIdx.x, (int)gridDim.x);
}

// workgroups [0, n0) of every column group run problem 0, the rest problem 1
template <int KB, int NBLK>
__global__ void __launch_bounds__(128 * NBLK)
tgf_pair_kernel(FwdArgs a0, FwdArgs a1, int n0) {
    if ((int)blockIdx.x < n0) tgf_body<KB, NBLK>(a0, (int)blockIdx.x, n0);
    else tgf_body<KB, NBLK>(a1, (int)blockIdx.x - n0, (int)gridDim.x - n0);
}

// column blocks per workgroup for n output channels: 6 (192 columns) where that divides n and 128 does not leave fewer
// workgroup columns, 4 (128), 2 (64); 0 = not covered
static int blocks_for(int n) {
    if (n % 192 == 0 && n % 128 != 0) return 6;
    if (n % 128 == 0) return 4;
    if (n % 192 == 0) return 6;
    if (n == 64) return 2;
    return 0;
}

}  // namespace tgf
}  // namespace pn2

using namespace pn2;
using namespace pn2::tgf;

// 256 input channels: 64 output columns per workgroup (W_i slice 66 KiB next to the 66 KiB operand tile), four waves
static int blocks_for_k(int k, int n) { return k == 256 ? (n % 64 == 0 ? 2 : 0) : blocks_for(n); }

extern "C" int pn2x_tg_fwd2_supported(int c_in, int c_out) {
    return ((c_in == 64 || c_in == 128 || c_in == 256) && blocks_for_k(c_in, c_out) != 0) ? 1 : 0;
}

extern "C" int pn2x_tg_fwd2(long rows, int k, int n, const float *x, int ldx, const float *w, int ldw, float *y, int ldy,
                            const double *sums_in, const float *gamma, const float *beta, const float *conv_bias, float eps,
                            float momentum, float *running_mean, float *running_var, long long *num_batches_tracked,
                            float *save_mean, float *save_invstd, double *sums_out, void *stream) {
    if (rows < 1 || !pn2x_tg_fwd2_supported(k, n) || ldx < k || ldx % 4 || ldw < k || ldw % 4 || ldy < n) return PN2_EINVAL;
    if (!x || !w || !y || !sums_in || !gamma || !beta || !save_mean || !save_invstd) return PN2_ENULL;
    if (((uintptr_t)x | (uintptr_t)w) % 16) return PN2_EINVAL;
    FwdArgs a{rows, n, x, ldx, w, ldw, y, ldy, sums_in, gamma, beta, conv_bias, eps, momentum, running_mean, running_var,
              num_batches_tracked, save_mean, save_invstd, sums_out};
    const int nblk = blocks_for_k(k, n), kb = k / 32;
    const int ny = n / (32 * nblk);
    const long tiles = (rows + BM - 1) / BM;
    hipStream_t st = (hipStream_t)stream;
#define PN2_TGF_LAUNCH(KB_, NBLK_)                                                                                      \
    do {                                                                                                              \
        using P = Plan<KB_, NBLK_>;                                                                                   \
        const size_t lds = (size_t)P::lds_floats * sizeof(float);                                                     \
        static PerDeviceOnce once;                                                                                    \
        if (once.first_use())                                                                                         \
            (void)hipFuncSetAttribute((const void *)tgf_kernel<KB_, NBLK_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        /* persistent workgroups: as many per CU as the LDS footprint admits (at most 2), shared among the column groups */ \
        long per_cu = (long)(160 * 1024 / lds);                                                                       \
        if (per_cu > 2) per_cu = 2;                                                                                   \
        if (per_cu < 1) per_cu = 1;                                                                                   \
        long gx = (long)num_compute_units() * per_cu / ny;  /* never more workgroups than slots: a straggler doubles the time */ \
        if (gx < 1) gx = 1;                                                                                           \
        if (gx > tiles) gx = tiles;                                                                                   \
        hipLaunchKernelGGL((tgf_kernel<KB_, NBLK_>), dim3((unsigned)gx, ny), dim3(P::T), lds, st, a);                 \
    } while (0)
    if (kb == 8) {
        PN2_TGF_LAUNCH(8, 2);
    } else if (kb == 4) {
        if (nblk == 6) PN2_TGF_LAUNCH(4, 6);
        else if (nblk == 4) PN2_TGF_LAUNCH(4, 4);
        else PN2_TGF_LAUNCH(4, 2);
    } else {
        if (nblk == 6) PN2_TGF_LAUNCH(2, 6);
        else if (nblk == 4) PN2_TGF_LAUNCH(2, 4);
        else PN2_TGF_LAUNCH(2, 2);
    }
#undef PN2_TGF_LAUNCH
    return check_launch();
}

// instantiated for the 128-channel layers of the keypoint-query modules (128 -> 128, 128 -> 192)
extern "C" int pn2x_tg_fwd2_pair_supported(int c_in, int c_out) {
    return (c_in == 128 && (c_out == 128 || c_out == 192) && blocks_for_k(c_in, c_out) != 0) ? 1 : 0;
}

// Two problems of the SAME layer shape (k -> n) in one launch: the persistent workgroups are split in proportion to the
// problems' tile counts, so both shares run the same number of rounds.  The two neighbourhood sizes of a keypoint-query module
// (10752 and 43008 rows at 32 clouds) as two launches were 168 one-tile workgroups + 3 ragged rounds, each launch paying its own
// constants / W_i prologue and tail.
extern "C" int pn2x_tg_fwd2_pair(long rows0, int k, int n, const float *x0, int ldx0, const float *w0, int ldw0, float *y0, int ldy0,
                                 const double *sums_in0, const float *gamma0, const float *beta0, const float *conv_bias0, float eps0,
                                 float momentum0, float *running_mean0, float *running_var0, long long *nbt0, float *save_mean0,
                                 float *save_invstd0, double *sums_out0,
                                 long rows1, const float *x1, int ldx1, const float *w1, int ldw1, float *y1, int ldy1,
                                 const double *sums_in1, const float *gamma1, const float *beta1, const float *conv_bias1, float eps1,
                                 float momentum1, float *running_mean1, float *running_var1, long long *nbt1, float *save_mean1,
                                 float *save_invstd1, double *sums_out1, void *stream) {
    if (rows0 < 1 || rows1 < 1) return PN2_EINVAL;
    if (!pn2x_tg_fwd2_pair_supported(k, n)) return PN2_ERANGE;
    if (ldx0 < k || ldx0 % 4 || ldw0 < k || ldw0 % 4 || ldy0 < n || ldx1 < k || ldx1 % 4 || ldw1 < k || ldw1 % 4 || ldy1 < n) return PN2_EINVAL;
    if (!x0 || !w0 || !y0 || !sums_in0 || !gamma0 || !beta0 || !save_mean0 || !save_invstd0) return PN2_ENULL;
    if (!x1 || !w1 || !y1 || !sums_in1 || !gamma1 || !beta1 || !save_mean1 || !save_invstd1) return PN2_ENULL;
    if (((uintptr_t)x0 | (uintptr_t)w0 | (uintptr_t)x1 | (uintptr_t)w1) % 16) return PN2_EINVAL;
    FwdArgs a0{rows0, n, x0, ldx0, w0, ldw0, y0, ldy0, sums_in0, gamma0, beta0, conv_bias0, eps0, momentum0, running_mean0, running_var0,
               nbt0, save_mean0, save_invstd0, sums_out0};
    FwdArgs a1{rows1, n, x1, ldx1, w1, ldw1, y1, ldy1, sums_in1, gamma1, beta1, conv_bias1, eps1, momentum1, running_mean1, running_var1,
               nbt1, save_mean1, save_invstd1, sums_out1};
    const int nblk = blocks_for_k(k, n);
    const int ny = n / (32 * nblk);
    const long t0 = (rows0 + BM - 1) / BM, t1 = (rows1 + BM - 1) / BM;
    hipStream_t st = (hipStream_t)stream;
#define PN2_TGF_PAIR(KB_, NBLK_)                                                                                        \
    do {                                                                                                              \
        using P = Plan<KB_, NBLK_>;                                                                                   \
        const size_t lds = (size_t)P::lds_floats * sizeof(float);                                                     \
        static PerDeviceOnce once;                                                                                    \
        if (once.first_use())                                                                                         \
            (void)hipFuncSetAttribute((const void *)tgf_pair_kernel<KB_, NBLK_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        long per_cu = (long)(160 * 1024 / lds);                                                                       \
        if (per_cu > 2) per_cu = 2;                                                                                   \
        if (per_cu < 1) per_cu = 1;                                                                                   \
        long cap = (long)num_compute_units() * per_cu / ny;                                                           \
        if (cap < 2) cap = 2;                                                                                         \
        long rounds = (t0 + t1 + cap - 1) / cap, g0, g1;                                                              \
        for (;; ++rounds) {  /* same number of rounds for both shares */                                              \
            g0 = (t0 + rounds - 1) / rounds;                                                                          \
            g1 = (t1 + rounds - 1) / rounds;                                                                          \
            if (g0 + g1 <= cap || rounds > t0 + t1) break;                                                            \
        }                                                                                                             \
        hipLaunchKernelGGL((tgf_pair_kernel<KB_, NBLK_>), dim3((unsigned)(g0 + g1), ny), dim3(P::T), lds, st, a0, a1, (int)g0); \
    } while (0)
    if (nblk == 6) PN2_TGF_PAIR(4, 6);
    else PN2_TGF_PAIR(4, 4);
#undef PN2_TGF_PAIR
    return check_launch();
}
