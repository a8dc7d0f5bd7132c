// fps_cull.hip -- furthest point sampling of LARGE clouds (2048 < n <= 8192) with exact spatial culling (gfx950).
//
// Same operator, same picks as fps.hip (reference sampling_gpu.cu:94-253: arg-max of the running minimum distance, ties by the
// shared-memory tree's order = (bitrev(k mod bs), k div bs) ascending).  fps.hip keeps every point in registers and updates
// ALL of them every iteration: at n = 8192 that is 16 points per lane on one compute unit, 0.66 us per pick of pure VALU
// issue (1.36 ms for 2048 picks, whatever the batch: one workgroup per cloud, the chain is sequential).  But after the first
// few dozen picks a new sample only lowers the running minimum of points NEAR it.  Here:
//
//   * prologue: the cloud is counting-sorted by a 16^3 cell grid in Morton order (LDS histogram + scan + scatter); a wave's
//     register slot j (64 points, one per lane) then holds 64 spatially coherent points.  Per slot: an exact bounding box and
//     the maximum of its points' running minima (wave-uniform values, kept by lane j of the wave);
//   * per pick: lane j evaluates the squared distance from the new sample to box j with the SAME rounded operations as the
//     point distance (fp32 subtraction, multiplication and fma are monotone, so this is a lower bound of every point's computed
//     distance, not just of the exact one); if it is not below the slot's maximum no point of the slot changes and the slot is
//     skipped -- a wave-uniform decision, so only the touched slots (about 1.3 of 16 per wave and pick at n = 8192) run the
//     distance update and the 6-step DPP maximum that refreshes the slot's value;
//   * the arg-max needs no scan over the points either: the wave's maximum is the maximum of its 16 slot values (4 DPP steps),
//     the winner inside the slot that holds it is found by an equality ballot on that one slot.  The points are no longer in the
//     reference's tie order (they are sorted by cell), so every point carries its tie rank and equal maxima are resolved by the
//     smallest rank: within the slot (wave min), across slots, across waves (LDS exchange, one barrier per pick, as fps.hip).
//   * a register slot selected at run time: the slot number is wave-uniform, so the register arrays are native 16-element
//     vectors indexed through M0-relative register addressing -- no branch chain over code copies per slot.
//
// Index-exact with fps.hip / the oracle on every case of tests/test_gpu_ops.py (lattices, duplicates, planes, lines, all-equal
// clouds: whole slots tie there and every rank comparison is exercised).
#include <stdlib.h>

#include "pn2_common.h"

namespace pn2 {
namespace fpc {

constexpr int kG = 16, kCells = kG * kG * kG;      // cell grid of the prologue sort

// signed-int order == float order (any non-NaN floats, both signs)
__device__ __forceinline__ int ordered(float f) {
    const int i = f2i(f);
    return i ^ ((i >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float unordered(int i) { return i2f(i ^ ((i >> 31) & 0x7fffffff)); }


__device__ __forceinline__ unsigned spread4(unsigned v) {  // 4 bits -> every third bit
    return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6);
}

// kT threads = kW waves, kP register slots of 64 points per wave: kW kP 64 = 8192 points
template <int kT, int kP>
__global__ void __launch_bounds__(kT)
fps_cull_kernel(int n, int m, int bs, int lg, const float *__restrict__ xyz_all, int *__restrict__ idx_all) {
    constexpr int kW = kT / kWave;
    static_assert(kW * kP * 64 == 8192 && (kP == 16 || kP == 32), "register layout");
    typedef float fvec __attribute__((ext_vector_type(kP)));
    typedef int ivec __attribute__((ext_vector_type(kP)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int *ent = reinterpret_cast<int *>(smem);          // [parity][field][wave]: field 0 = value bits, 1 = tie rank
    float *lxyz = smem + 2 * 2 * 16;                   // (n, 3): the sample's coordinates by point index (main loop)
    // prologue scratch, aliased with lxyz (which is filled last)
    int *hist = reinterpret_cast<int *>(lxyz);         // [kCells]
    int *part = hist + kCells;                         // [2][kT] scan ping-pong
    int *perm = part + 2 * kT;                         // [n] sorted position -> point index
    int *red = perm + 8192;                            // [6][kW] bounding-box partials

    const float *__restrict__ xyz = xyz_all + (size_t)blockIdx.x * n * 3;
    int *__restrict__ idx = idx_all + (size_t)blockIdx.x * m;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;

    // ---- 1. bounding box of the cloud ---------------------------------------------------------------------------------------
    int bb[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) bb[a] = (int)0x80000000;
    for (int k = tid; k < n; k += kT) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = xyz[3 * k + a];
            const int hi = ordered(v), lo = ordered(-v);
            bb[a] = hi > bb[a] ? hi : bb[a];
            bb[3 + a] = lo > bb[3 + a] ? lo : bb[3 + a];
        }
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        const int v = wave_max_i32(bb[a]);
        if (lane == 0) red[a * kW + w] = v;
    }
    for (int i = tid; i < kCells; i += kT) hist[i] = 0;
    __syncthreads();
    float lo3[3], sc3[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        int hi = red[a * kW], lo = red[(3 + a) * kW];
        for (int ww = 1; ww < kW; ++ww) {
            hi = max(hi, red[a * kW + ww]);
            lo = max(lo, red[(3 + a) * kW + ww]);
        }
        const float fhi = unordered(hi), flo = -unordered(lo), ext = fhi - flo;
        lo3[a] = flo;
        sc3[a] = ext > 0.f ? (float)kG / ext : 0.f;  // (the sort only groups nearby points: any finite scale is correct)
    }
    auto cell_key = [&](int k) -> int {
        unsigned c[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float g = (xyz[3 * k + a] - lo3[a]) * sc3[a];
            int ci = (int)g;                       // NaN / Inf coordinates land in cell 0 or kG-1 after the clamp
            ci = ci < 0 ? 0 : (ci > kG - 1 ? kG - 1 : ci);
            c[a] = (unsigned)ci;
        }
        return (int)(spread4(c[0]) | (spread4(c[1]) << 1) | (spread4(c[2]) << 2));  // Morton order of the cells
    };
    // ---- 2. counting sort by cell -------------------------------------------------------------------------------------------
    for (int k = tid; k < n; k += kT) atomicAdd(&hist[cell_key(k)], 1);
    __syncthreads();
    constexpr int kPer = kCells / kT;  // 8 cells per thread
    int loc[kPer], s = 0;
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
        loc[i] = s;
        s += hist[tid * kPer + i];
    }
    part[tid] = s;
    __syncthreads();
    int src = 0;
    for (int d = 1; d < kT; d <<= 1) {  // inclusive scan of the per-thread totals
        const int v = part[src * kT + tid] + (tid >= d ? part[src * kT + tid - d] : 0);
        part[(1 - src) * kT + tid] = v;
        src = 1 - src;
        __syncthreads();
    }
    const int base = part[src * kT + tid] - s;
#pragma unroll
    for (int i = 0; i < kPer; ++i) hist[tid * kPer + i] = base + loc[i];
    __syncthreads();
    for (int k = tid; k < n; k += kT) perm[atomicAdd(&hist[cell_key(k)], 1)] = k;
    __syncthreads();

    // ---- 3. this thread's points: slot j of wave w, lane l = sorted position (j kW + w) 64 + l.  Blocks of 64 consecutive
    // sorted points are dealt to the waves round-robin: the slots one sample touches are neighbours in the sorted order, and
    // the waves meet at a barrier every pick -- dealt wave-major, one wave updated 4 - 8 slots while seven waited (4.3 ms)
    // Register arrays as native vectors: a slot is selected by a WAVE-UNIFORM run-time index, which the compiler turns into
    // M0-relative register addressing (s_set_gpr_idx / v_movrel: two or three instructions), not into a branch tree over 16
    // code copies (measured: 0.37 us per touched slot through a 4-level scalar branch tree -- instruction refetches -- and
    // 0.41 us for the winner search; the builder's 16-way chain of round 3 was worse still).
    // block of slot j of this wave: round-robin, skewed by the higher bits of j -- in Morton order a block's spatial neighbours
    // sit 1, 2, 4, 8, ... blocks away, and a plain j kW + w puts the ones 8 and 16 away into the same wave again
    auto blk = [&](int j) { return j * kW + ((w ^ j ^ (j >> 3)) & (kW - 1)); };
    fvec px, py, pz, pt;
    ivec pk;  // tie rank (bitrev(k mod bs) << 4 | k div bs): smaller wins among equal distances
#pragma unroll
    for (int j = 0; j < kP; ++j) {
        const int sp = blk(j) * 64 + lane;
        const bool valid = sp < n;
        const int k = valid ? perm[sp] : 0;
        px[j] = valid ? xyz[3 * k + 0] : 0.f;
        py[j] = valid ? xyz[3 * k + 1] : 0.f;
        pz[j] = valid ? xyz[3 * k + 2] : 0.f;
        pt[j] = valid ? 1e10f : -1.0f;  // padded lanes can never win (all real distances are >= 0)
        const unsigned rev = lg ? (__builtin_bitreverse32((unsigned)(k & (bs - 1))) >> (32 - lg)) : 0u;
        pk[j] = valid ? (int)((rev << 4) | (unsigned)(k >> lg)) : 0x7fffffff;
    }
    // slot boxes and slot maxima, kept by lane j (of every 16-lane row: only row 0 is read)
    float blx = 0.f, bly = 0.f, blz = 0.f, bhx = 0.f, bhy = 0.f, bhz = 0.f;
    int slotbits = (int)0x80000000;
#pragma unroll
    for (int j = 0; j < kP; ++j) {
        const bool valid = blk(j) * 64 + lane < n;
        const int none = (int)0x80000000;
        const int hx = wave_max_i32(valid ? ordered(px[j]) : none), lx = wave_max_i32(valid ? ordered(-px[j]) : none);
        const int hy = wave_max_i32(valid ? ordered(py[j]) : none), ly = wave_max_i32(valid ? ordered(-py[j]) : none);
        const int hz = wave_max_i32(valid ? ordered(pz[j]) : none), lz = wave_max_i32(valid ? ordered(-pz[j]) : none);
        const bool any = blk(j) * 64 < n;  // (wave-uniform) the slot holds at least one point
        if ((lane & (kP - 1)) == j) {
            bhx = unordered(hx); bhy = unordered(hy); bhz = unordered(hz);
            blx = -unordered(lx); bly = -unordered(ly); blz = -unordered(lz);
            slotbits = any ? f2i(1e10f) : f2i(-1.0f);
        }
    }
    __syncthreads();  // perm is dead: lxyz may overwrite it
    for (int i = tid; i < 3 * n; i += kT) lxyz[i] = xyz[i];
    float cx = xyz[0], cy = xyz[1], cz = xyz[2];
    if (tid == 0) idx[0] = 0;
    __syncthreads();

    // ---- 4. the picks -------------------------------------------------------------------------------------------------------
    for (int it = 1; it < m; ++it) {
        // which slots can change at all?  (lanes 0..15 decide for slots 0..15; NaN anywhere -> the slot is updated)
        const float ex = fmaxf(fmaxf(blx - cx, cx - bhx), 0.f), ey = fmaxf(fmaxf(bly - cy, cy - bhy), 0.f),
                    ez = fmaxf(fmaxf(blz - cz, cz - bhz), 0.f);
        const float bound = __builtin_fmaf(ez, ez, __builtin_fmaf(ex, ex, ey * ey));
        constexpr unsigned kSlots = kP == 32 ? 0xffffffffu : 0xffffu;
        unsigned touch = (unsigned)__ballot(!(bound >= i2f(slotbits))) & kSlots;
        while (touch) {
            const int j = __builtin_ctz(touch);  // wave-uniform
            touch &= touch - 1;
            const float tt = fmin_raw(sqdist(px[j], py[j], pz[j], cx, cy, cz), pt[j]);
            pt[j] = tt;
            const int mx = wave_max_i32(f2i(tt));  // >= 0 floats (or -1.0f) order like their bit patterns
            slotbits = (lane & (kP - 1)) == j ? mx : slotbits;
        }
        // the wave's maximum = the maximum of its slot values; its holder = the smallest tie rank among the points that reach it
        int rmax = row_group_max_i32<16>(slotbits);
        if constexpr (kP == 32) PN2_DPP_STEP("v_max_i32_dpp", rmax, "row_bcast:15 row_mask:0xa bank_mask:0xf");  // lane 31: rows 0 and 1
        const int wbest = __builtin_amdgcn_readlane(rmax, kP - 1);
        unsigned cand = (unsigned)__ballot(slotbits == wbest) & kSlots;
        unsigned ltk = 0x7fffffffu;
        while (cand) {
            const int j = __builtin_ctz(cand);
            cand &= cand - 1;
            const unsigned r = wave_min_u32(f2i(pt[j]) == wbest ? (unsigned)pk[j] : 0x7fffffffu);
            ltk = r < ltk ? r : ltk;
        }
        int *e = ent + (it & 1) * (2 * 16);
        if (lane == 0) {
            e[w] = wbest;
            e[16 + w] = (int)ltk;
        }
        __syncthreads();
        const int sl = lane & (kW - 1);
        const int ev = e[sl];
        const unsigned etk = (unsigned)e[16 + sl];
        const int gmax = __builtin_amdgcn_readfirstlane(row_group_max_i32<kW>(ev));
        const unsigned wtk = (unsigned)__builtin_amdgcn_readfirstlane((int)row_group_min_u32<kW>(ev == gmax ? etk : 0x7fffffffu));
        const unsigned rev = wtk >> 4;
        const int kstar = (int)((lg ? (__builtin_bitreverse32(rev) >> (32 - lg)) : 0u) + ((wtk & 15u) << lg));
        if (tid == 0) idx[it] = kstar;
        cx = lxyz[3 * kstar + 0];
        cy = lxyz[3 * kstar + 1];
        cz = lxyz[3 * kstar + 2];
    }
}

// n in (2048, 8192], bs = 1024 (lg = 10): the register layout holds 8192 points, the tie rank packs k div bs into 4 bits.
template <int kT, int kP>
static int launch_cull(int b, int n, int m, int bs, int lg, const float *xyz, int *idx, hipStream_t st) {
    constexpr int kW = kT / kWave;
    const size_t prologue = (size_t)(kCells + 2 * kT + 8192 + 6 * kW) * sizeof(int);
    const size_t loop = (size_t)3 * n * sizeof(float);
    const size_t lds = 2 * 2 * 16 * sizeof(int) + (prologue > loop ? prologue : loop);
    auto kfn = fps_cull_kernel<kT, kP>;
    static PerDeviceOnce raised;
    if (raised.first_use())
        (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipLaunchKernelGGL(kfn, dim3(b), dim3(kT), lds, st, n, m, bs, lg, xyz, idx);
    return check_launch();
}

}  // namespace fpc

int fps_cull_launch(int b, int n, int m, int bs, int lg, const float *xyz, int *idx, hipStream_t st) {
    if (n > 8192 || (n >> lg) > 15) return PN2_ERANGE;
    static const int waves = [] { const char *e = getenv("PN2_FPC_WAVES"); return e ? atoi(e) : 8; }();  // A/B: 8 waves x 16 slots | 4 x 32
    if (waves == 4) return fpc::launch_cull<256, 32>(b, n, m, bs, lg, xyz, idx, st);
    return fpc::launch_cull<512, 16>(b, n, m, bs, lg, xyz, idx, st);
}

}  // namespace pn2
