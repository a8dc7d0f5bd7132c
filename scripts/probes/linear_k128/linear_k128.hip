// linear_k128.hip -- y = act(x W^T + bias) for a 128-channel input over MANY rows, no LDS (gfx950).
//
// The backbone's last per-point layer (conv1 + bn1 folded, reference network/models/backbones.py:131-133: Conv1d(128, 384, 1) +
// BatchNorm1d + ReLU over all B * N points) is the one large dense layer of the inference path with a SHORT reduction: 65536 x
// 128 -> 384 at batch 64.  The library's best recorded solution runs it at 0.62 of the fp32 matrix peak (66 us); a reduction of 128
// is short enough for BOTH operands of the matrix instruction to live in registers:
//
//   * v_mfma_f32_16x16x4_f32.  A wave owns CB column blocks of 16 outputs; its slice of W (16 CB columns x 128) is loaded ONCE into
//     32 CB registers per lane and stays there for every row block of the persistent workgroup (the B operand);
//   * the A operand comes STRAIGHT from memory: a row block is 16 rows x 128 channels = 8 KB = eight 16-byte loads per lane,
//     requested one block ahead into a second register set -- no LDS staging, no ds_read / ds_write, no barrier anywhere in the
//     kernel.  The k index is permuted so that a lane's 16 bytes are consecutive channels (lane (row i, quarter q) holds channels
//     16 u + 4 q + c, u < 8, c < 4: the four quarter-lanes of a row read 64 contiguous bytes per load) -- the sum over k does not
//     care about the order as long as W is held in the same permutation;
//   * the eight waves of a workgroup take the eight 48-column slabs of the 384 outputs (CB = 3) and walk the same row blocks, so a
//     block is fetched from HBM once and found in the CU's vector cache by the other seven waves;
//   * bias = accumulator initialisation, ReLU on the accumulators, 64-byte store segments straight from them.
//
// Per row block and wave: 32 CB matrix instructions x 32 cycles against 8 loads + 4 CB stores: the matrix pipe is the bound.
// Bound: MFMA (fp32 dense peak 157.3 TFLOP/s); algorithmic work 2 R 128 N flop, 4 R (128 + N) bytes.
#include <cstdlib>
#include "pn2_common.h"
#include "../../include/pn2_ext.h"

namespace pn2 {
namespace lk {

constexpr int K = 128;
constexpr int kWaves = 8, kT = 64 * kWaves;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

struct Args {
    long R;
    const float *X; int ldx;
    const float *W; int ldw;
    const float *bias;
    float *Y; int ldy;
    int relu;
    int blocks_per_wg;  // 16-row blocks per workgroup (consecutive)
};

__device__ __forceinline__ rsrc_t make_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 ld_b128(rsrc_t r, unsigned voff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0));
}
__device__ __forceinline__ unsigned uniform(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }

template <int CB>
__global__ void __launch_bounds__(kT)
linear_k128_kernel(Args a) {
    const int lane = threadIdx.x & 63, wave = (int)uniform(threadIdx.x >> 6);
    const int i = lane & 15, q = lane >> 4;  // operand row (A) / column (B) of the 16 x 16 block, and the lane's quarter of k
    const int n0 = wave * 16 * CB;
    // ---- this wave's slice of W, in the k permutation: lane (j, q) register (u, c) = W[n0 + 16 cb + j][16 u + 4 q + c]
    f32x4 wreg[CB][8];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
        const float *wr = a.W + (size_t)(n0 + 16 * cb + i) * a.ldw + 4 * q;
#pragma unroll
        for (int u = 0; u < 8; ++u) wreg[cb][u] = *reinterpret_cast<const f32x4 *>(wr + 16 * u);
    }
    float binit[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) binit[cb] = a.bias ? a.bias[n0 + 16 * cb + i] : 0.f;

    // rows of this workgroup: blocks [b0, b1) of 16; loads against a descriptor over the whole operand (rows beyond R read zero),
    // stores against one over the output (rows beyond R are dropped)
    const long nblocks = (a.R + 15) / 16;
    const long b0 = (long)blockIdx.x * a.blocks_per_wg;
    const long b1 = (b0 + a.blocks_per_wg) < nblocks ? (b0 + a.blocks_per_wg) : nblocks;
    if (b0 >= b1) return;
    const char *xb = reinterpret_cast<const char *>(a.X) + (size_t)b0 * 16 * a.ldx * 4;
    const size_t xleft = ((size_t)(a.R - 1 - b0 * 16) * a.ldx + K) * 4;  // bytes from this workgroup's first row to the operand's end
    const rsrc_t rx = make_rsrc(xb, xleft > 0x7ffffff0u ? 0x7ffffff0u : (unsigned)xleft);
    char *yb = reinterpret_cast<char *>(a.Y) + (size_t)b0 * 16 * a.ldy * 4;
    const size_t yleft = ((size_t)(a.R - 1 - b0 * 16) * a.ldy + (size_t)(kWaves * 16 * CB)) * 4;
    const rsrc_t ry = make_rsrc(yb, yleft > 0x7ffffff0u ? 0x7ffffff0u : (unsigned)yleft);
    const unsigned xrow = uniform(4u * a.ldx), yrow = uniform(4u * a.ldy);
    const unsigned xlane = (unsigned)i * xrow + 16u * q;                  // + 64 u + block * 16 rows
    const unsigned ylane = (unsigned)(4 * q) * yrow + 4u * (n0 + i);      // accumulator element r: row 4 q + r, column i of the block

    const float floor_ = a.relu ? 0.f : -__builtin_inff();
    const int nb = (int)(b1 - b0);  // blocks of this workgroup, local indices below
    f32x4 a0[8], a1[8];
    auto fetch = [&](f32x4 (&dst)[8], int blk) {
        const unsigned base = (unsigned)blk * 16u * xrow + xlane;
#pragma unroll
        for (int u = 0; u < 8; ++u) dst[u] = ld_b128(rx, base + 64u * u);
    };
    // (`live` false: a block beyond this workgroup's range -- computed on re-fetched rows, every store pushed outside the descriptor.
    // The loop below has NO conditional loads or stores: hipcc's s_waitcnt counts are then exact, and a block's matrix instructions
    // wait for ITS loads only; with `if (next block exists) fetch` the merge of the two paths made them wait for the next block's)
    auto compute = [&](const f32x4 (&av)[8], int blk, bool live) {
        f32x4 acc[CB];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) acc[cb] = f32x4{binit[cb], binit[cb], binit[cb], binit[cb]};
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][c], wreg[cb][u][c], acc[cb], 0, 0, 0);
            }
        }
        const unsigned base = live ? (unsigned)blk * 16u * yrow + ylane : 0x7ffff000u;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            f32x4 v = acc[cb];
            // ReLU without a branch (floor = 0 or -inf; propagates NaN like torch.relu): no control flow inside the block loop
            v.x = !(v.x <= floor_) ? v.x : floor_; v.y = !(v.y <= floor_) ? v.y : floor_;
            v.z = !(v.z <= floor_) ? v.z : floor_; v.w = !(v.w <= floor_) ? v.w : floor_;
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 bits = __builtin_bit_cast(u32x4, v);  // (whole-vector cast: a bit cast of a vector ELEMENT reads element 0)
            __builtin_amdgcn_raw_buffer_store_b32(bits.x, ry, (int)(base + 64u * cb), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(bits.y, ry, (int)(base + 64u * cb + yrow), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(bits.z, ry, (int)(base + 64u * cb + 2u * yrow), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(bits.w, ry, (int)(base + 64u * cb + 3u * yrow), 0, 0);
        }
    };
    const int last = nb - 1;
    fetch(a0, 0);
    fetch(a1, 1 < nb ? 1 : last);
    for (int blk = 0; blk < nb; blk += 2) {  // (both register sets are requested before the loop: the same loads are in flight at
        compute(a0, blk, true);             // its head whichever way it is entered)
        __builtin_amdgcn_sched_barrier(0);  // (left alone the scheduler sinks each fetch to just in front of its first use)
        fetch(a0, blk + 2 < nb ? blk + 2 : last);
        __builtin_amdgcn_sched_barrier(0);
        compute(a1, blk + 1, blk + 1 < nb);
        __builtin_amdgcn_sched_barrier(0);
        fetch(a1, blk + 3 < nb ? blk + 3 : last);
        __builtin_amdgcn_sched_barrier(0);
    }
}

}  // namespace lk
}  // namespace pn2

using namespace pn2;

extern "C" int pn2x_linear_k128_supported(int k, int n) { return (k == 128 && (n == 128 || n == 256 || n == 384)) ? 1 : 0; }

extern "C" int pn2x_linear_k128(long rows, int n, const float *x, int ldx, const float *w, int ldw, const float *bias, int relu, float *y,
                                int ldy, void *stream) {
    using namespace pn2::lk;
    if (rows < 1 || !pn2x_linear_k128_supported(K, n) || ldx < K || ldw < K || ldy < n || ldx % 4 || ldw % 4) return PN2_EINVAL;
    if (!x || !w || !y) return PN2_ENULL;
    if (((uintptr_t)x | (uintptr_t)w) % 16 || (uintptr_t)y % 4) return PN2_EINVAL;
    if ((size_t)rows * (size_t)(ldx > ldy ? ldx : ldy) * 4 > 0x7fffffffffffULL) return PN2_EINVAL;
    const long nblocks = (rows + 15) / 16;
    // persistent grid: one workgroup (eight waves, two per SIMD) per compute unit; a workgroup's descriptor must span its rows
    static const int wg_cap = [] { const char *e = getenv("PN2_LK_WGS"); const int v = e ? atoi(e) : 0; return v; }();  // probes
    long wgs = wg_cap > 0 ? wg_cap : num_compute_units();
    if (wgs > nblocks) wgs = nblocks;
    long per = (nblocks + wgs - 1) / wgs;
    while ((size_t)per * 16 * (size_t)(ldx > ldy ? ldx : ldy) * 4 > 0x7ff00000ULL) {  // (32-bit descriptor offsets)
        per = (per + 1) / 2;
    }
    wgs = (nblocks + per - 1) / per;
    Args a{rows, x, ldx, w, ldw, bias, y, ldy, relu, (int)per};
    hipStream_t st = (hipStream_t)stream;
    if (n == 384) hipLaunchKernelGGL(lk::linear_k128_kernel<3>, dim3((unsigned)wgs), dim3(kT), 0, st, a);
    else if (n == 256) hipLaunchKernelGGL(lk::linear_k128_kernel<2>, dim3((unsigned)wgs), dim3(kT), 0, st, a);
    else hipLaunchKernelGGL(lk::linear_k128_kernel<1>, dim3((unsigned)wgs), dim3(kT), 0, st, a);
    return check_launch();
}
