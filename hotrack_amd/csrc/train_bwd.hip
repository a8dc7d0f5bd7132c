// train_bwd.hip -- the whole backward of one fused [Conv 1x1 + BatchNorm + ReLU] layer in ONE kernel (gfx950).
//
// train_gemm.hip runs the backward of layer i as two kernels, tg_dgrad (G_{i-1} = (dY_i W_i) . mask) and tg_wgrad
// (dW_i = dY_i^T H_{i-1}); each of them reads G_i and Y_i and rebuilds the pre-activation gradient
//      dY_i = gamma_i invstd_i (g_i - mean(g_i) - xhat_i mean(g_i xhat_i))
// on the way into LDS (tg_dgrad once per 64-column block), and each reads Y_{i-1}.  On the layer shapes of the training step
// that traffic IS the cost: the wide-row / narrow-channel layers (262144 x 32..64, 131072 x 64..128) run at the HBM rate, the
// others at a quarter of the matrix rate.  Here a persistent 512-thread workgroup takes a 64-row tile and
//   1. builds dY_i (64 x C_i) and xhat_{i-1} (64 x C_{i-1}) ONCE, in LDS (operands fetched into registers one tile ahead);
//   2. data gradient:   acc_g (64 x C_{i-1}) = dY_i W_i            -- A from LDS, W_i straight from L2 (it is re-read by every
//                                                                     tile of every workgroup: 4..96 KiB, cache resident);
//   3. weight gradient: acc_w (C_i x C_{i-1}) += dY_i^T H_{i-1}    -- both operands from LDS, H = relu(xhat gamma + beta) on
//                       the way into the matrix instruction; the accumulators stay in registers for ALL tiles of the workgroup
//                       and are written once, as one partial tile per workgroup (tg_reduce_multi sums them);
//   4. epilogue: ReLU mask of layer i-1, its BatchNorm-backward sums, G_{i-1} stored as float4 rows.
// So G_i, Y_i and Y_{i-1} are read once and G_{i-1} written once: 201 MB instead of 368 MB for a 131072 x (64 -> 128) layer.
//
// v_mfma_f32_32x32x2_f32 throughout.  Work split over the 8 waves: the 64 x C_{i-1} data-gradient tile is 2 NB blocks of
// 32 x 32 (C_{i-1} = 32 NB); with fewer than 8 blocks the reduction over C_i is split across wave groups and summed in the
// epilogue.  The C_i x C_{i-1} weight gradient is NB KB blocks (C_i = 32 KB): a wave owns NB KB / 8 of them (sharing the H
// operand), or, with fewer than 8 blocks, a slice of the tile's 64 rows (one partial tile per slice).
// Gradient sources: GMODE 0, the dense pre-masked g_i this kernel itself writes for the layer above; GMODE 1, the dense gradient
// of a materialised top layer (ReLU mask recomputed from Y_i on load); GMODE 2, the top layer of
// a max-pooled stack, routed on load: g_i[row] = dout[row / K] where arg[row / K] == row % K, ReLU-masked from Y_i (dout / arg
// are K times smaller than the activation and stay in L2: the layer reads Y_i and Y_{i-1} only).
// A layer wider than the instantiated C_i (fp1's 128 -> 384 top, sa3's 128 -> 512) runs as 2 - 4 launches over column slices of
// layer i: every slice owns its rows of dW_i completely; the data gradient is a sum over the slices, so all but the last launch
// write their raw partial product (raw_out) and the next one adds it (Gadd, in place) before the mask and the sums.
#include <stdlib.h>
#include "pn2_common.h"
#include "../../include/pn2_ext.h"

namespace pn2 {
namespace tgb {

constexpr int kT = 512;
constexpr int BM = 64;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float relu_nan(float h) { return !(h <= 0.f) ? h : 0.f; }  // propagates NaN like torch

struct BwdArgs {
    long R;
    const float *G; int ldg;    // GMODE 0: g_i = dH_i . [H_i > 0] (R x C_i); GMODE 2: d(max over Kmax rows) (R / Kmax x C_i)
    const int *arg; int ldarg, Kmax, kshift;  // GMODE 2: arg-max row within the group (row stride ldarg); kshift >= 0: Kmax = 1 << kshift
    const float *Y; int ldy;    // pre-activations of layer i
    const float *mean, *invstd, *gamma, *beta;
    const double *sums_bwd;     // layer i: kBnRep copies of [sum(g) | sum(g xhat)], each half sums_ld wide (>= C_i: a column slice)
    int sums_ld;
    const float *Gadd; int ldga; // partial data gradient of the other column slices of layer i, added before the mask (or null)
    int raw_out;                 // 1: Gp receives the unmasked partial product, no sums (a later slice finishes the layer)
    const float *W; int ldw;    // (C_i x C_{i-1})
    const float *Yp; int ldyp;  // pre-activations of layer i-1 (R x C_{i-1})
    const float *mean_p, *invstd_p, *gamma_p, *beta_p;
    float *Gp; int ldgp;        // out: g_{i-1}
    double *sums_bwd_p;         // out (accumulated): layer i-1
    float *partial;             // out: [gridDim.x * KS_W][C_i][C_{i-1}]
    float *dW;                  // zeroed by workgroup 0 (tg_reduce_multi accumulates into it)
#ifdef PN2_TGB_PROFILE
    long long *prof;            // tuning builds: [gridDim.x][8 waves][8] cycle counts of each wave's lane 0 (scripts/probes/tgb_profile.py)
#endif
};

#ifdef PN2_TGB_PROFILE
#define TGB_T(var) const long long var = clock64()
#define TGB_ADD(slot, t1, t0) do { if (lane == 0) pacc[slot] += (t1) - (t0); } while (0)
#define TGB_ADD2(slot, t1, t0) do { if (lane == 0) pacc2[slot] += (t1) - (t0); } while (0)
#else
#define TGB_T(var)
#define TGB_ADD(slot, t1, t0)
#define TGB_ADD2(slot, t1, t0)
#endif

template <int NB, int KB, int GMODE = 0>
struct Plan {
    static constexpr int N = 32 * NB, Kd = 32 * KB;
    static constexpr int LDY = Kd + 1;   // odd: the 32 rows a data-gradient A read touches fall on 32 banks
    static constexpr int LDX = N + 4;    // float4 rows
    static constexpr int DBLK = 2 * NB;
    static constexpr int KS_D = DBLK >= 8 ? 1 : 8 / DBLK;
    static constexpr int WBLK = NB * KB;
    static constexpr int KS_W = WBLK >= 8 ? 1 : 8 / WBLK;
    static constexpr int WB = WBLK >= 8 ? WBLK / 8 : 1;
    static constexpr int QN = N / 4, RG = kT / QN;  // epilogue / Yp mapping: thread = (column quad, row group), NB rows each
    // W_i: resident in LDS for the narrow layers (<= 32 KiB; their kernels are HBM-bound and the next tile's operands can then
    // be requested a whole tile ahead), streamed from L2 through a register double buffer for the 128-channel ones
    static constexpr bool WLDS = NB <= 2;
    static constexpr int steps_d = (Kd / 2) / KS_D;      // matrix instructions of a wave's data-gradient block
    static constexpr int U = steps_d >= 32 ? 16 : steps_d / 2;  // W values per register buffer
    static constexpr int lds_floats = 7 * Kd + 4 * N + BM * LDY + BM * LDX + KS_D * BM * LDX + (WLDS ? Kd * N : 0);
    static_assert(WLDS || (steps_d % (2 * U) == 0 && U >= 1), "data-gradient batches come in pairs");
    static_assert(NB == 1 || NB == 2 || NB == 4, "C_{i-1} in {32, 64, 128}");
    static_assert(WBLK < 8 || WBLK % 8 == 0, "weight-gradient blocks must split evenly over 8 waves");
    static_assert(2 * RG * N <= KS_D * BM * LDX, "the final column-sum staging reuses the data-gradient staging");
};

// The tile loop of one problem, run by workgroup `wg` of the `nwg` workgroups assigned to it: a whole launch (tg_bwd_kernel) or
// one share of a launch serving two problems of the same layer shape (tg_bwd_pair_kernel: the two neighbourhood sizes of a
// keypoint-query module).
template <int NB, int KB, int GMODE>
__device__ __forceinline__ void tg_bwd_body(const BwdArgs &a, const int wg, const int nwg) {
    using P = Plan<NB, KB, GMODE>;
    constexpr int N = P::N, Kd = P::Kd, LDY = P::LDY, LDX = P::LDX;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *cA = lds;                        // [7][Kd]: mean, invstd, scale = gamma invstd, m1 = sum(g) / R, m2 = sum(g xhat) / R, gamma, beta
    float *cP = cA + 7 * Kd;                // [4][N]: mean, invstd, gamma, beta of layer i-1
    float *dYs = cP + 4 * N;                // [BM][LDY]
    float *Xs = dYs + BM * LDY;             // [BM][LDX]  xhat_{i-1}
    float *Gs = Xs + BM * LDX;              // [KS_D][BM][LDX]
    float *Ws = Gs + P::KS_D * BM * LDX;    // [Kd][N] (WLDS)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
#ifdef PN2_TGB_PROFILE
    long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    TGB_T(t_start);

    const long T = (a.R + BM - 1) / BM;
    // dY operands: a wave fetches 8 rows x 8 float4 (128 contiguous bytes per row) per step; KB steps cover its 8 rows
    const int arow = 8 * wave + (lane >> 3), aq = lane & 7;
    // Yp operands and the epilogue: thread = (column quad cq, row group rg), rows rg + RG i
    const int cq = tid % P::QN, rg = tid / P::QN;
    float4 pg[KB], py[KB], ph[NB];
    int4 par[GMODE == 2 ? KB : 1];
    // routed source on the 128-channel kernels: dout / arg are fetched by the commit itself (L2 hits: one row serves Kmax
    // rows of the tile) instead of travelling a tile ahead in 48 more registers
    constexpr bool kLateG = GMODE == 2 && NB == 4;
    auto fetch_g = [&](long tile) {
        long row = tile * BM + arow;
        row = row < a.R ? row : a.R - 1;
        const long grow = GMODE == 2 ? (a.kshift >= 0 ? ((int)row >> a.kshift) : ((int)row / a.Kmax)) : row;
#pragma unroll
        for (int i = 0; i < KB; ++i) {
            pg[i] = *reinterpret_cast<const float4 *>(a.G + grow * a.ldg + 4 * (8 * i + aq));
            if constexpr (GMODE == 2) par[i] = *reinterpret_cast<const int4 *>(a.arg + grow * a.ldarg + 4 * (8 * i + aq));
        }
    };
    auto prefetch = [&](long tile) {
        long row = tile * BM + arow;
        row = row < a.R ? row : a.R - 1;  // unconditional loads; the commit zeroes what lies beyond the problem
        if constexpr (!kLateG) fetch_g(tile);
#pragma unroll
        for (int i = 0; i < KB; ++i) py[i] = *reinterpret_cast<const float4 *>(a.Y + row * a.ldy + 4 * (8 * i + aq));
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            long r = tile * BM + rg + P::RG * i;
            r = r < a.R ? r : a.R - 1;
            ph[i] = *reinterpret_cast<const float4 *>(a.Yp + r * a.ldyp + 4 * cq);
        }
    };
    long tile = wg;
    // (the first tile's operands are requested after the constants' loads, further down: loads return in order, and the fp64
    // sums of the constants would otherwise wait for every HBM row queued in front of them -- round 5, see tg_bwd2_body)

    // W_i (where it is LDS-resident) is requested before the constants are derived and stored after them: one memory round trip
    constexpr int WQ = Kd * N / 4, WCNT = P::WLDS ? (WQ + kT - 1) / kT : 1;
    f32x4 wreg[WCNT];  // a native vector type: an array of float4 structs is copied with memcpy and stays in scratch memory
    if constexpr (P::WLDS) {
#pragma unroll
        for (int i = 0; i < WCNT; ++i) {
            const int e0 = tid + i * kT, e = e0 < WQ ? e0 : WQ - 1, kd = e / (N / 4), q = e % (N / 4);  // unconditional clamped loads
            wreg[i] = *reinterpret_cast<const f32x4 *>(a.W + (size_t)kd * a.ldw + 4 * q);
        }
    }
    {
        const float inv_r = (float)(1.0 / (double)a.R);
        for (int c = tid; c < Kd; c += kT) {
            double sa = 0.0, sb = 0.0;
            for (int r = 0; r < kBnRep; ++r) {
                sa += a.sums_bwd[(size_t)r * 2 * a.sums_ld + c];
                sb += a.sums_bwd[(size_t)r * 2 * a.sums_ld + a.sums_ld + c];
            }
            const float is = a.invstd[c];
            cA[c] = a.mean[c]; cA[Kd + c] = is; cA[2 * Kd + c] = a.gamma[c] * is;
            cA[3 * Kd + c] = (float)sa * inv_r; cA[4 * Kd + c] = (float)sb * inv_r;
            if constexpr (GMODE != 0) { cA[5 * Kd + c] = a.gamma[c]; cA[6 * Kd + c] = a.beta[c]; }
        }
        for (int c = tid; c < N; c += kT) {
            cP[c] = a.mean_p[c]; cP[N + c] = a.invstd_p[c]; cP[2 * N + c] = a.gamma_p[c]; cP[3 * N + c] = a.beta_p[c];
        }
        if (wg == 0)
            for (int e = tid; e < Kd * N; e += kT) a.dW[e] = 0.f;
        if (tile < T) prefetch(tile);
        if constexpr (P::WLDS) {
#pragma unroll
            for (int i = 0; i < WCNT; ++i) {
                const int e = tid + i * kT, kd = e / (N / 4), q = e % (N / 4);
                if (e < WQ) *reinterpret_cast<f32x4 *>(Ws + kd * N + 4 * q) = wreg[i];
            }
        }
    }
    __syncthreads();

    // (this thread's four channels of layer i-1 are the same for every tile; their constants are re-read from LDS where they
    // are used -- four ds_read_b128 per tile against sixteen registers held across both matrix phases)
    // xhat_{i-1} of this thread's elements: built by the commit, used again by the epilogue of the same tile (kept in registers
    // where there is room, re-read from LDS by the 128-channel kernels)
    constexpr bool kKeepX = NB < 4 && !(NB == 1 && GMODE == 2);
    float4 xh[kKeepX ? NB : 1];
    auto commit = [&](long tile) {
        const bool v = tile * BM + arow < a.R;
        if constexpr (kLateG) fetch_g(tile);
        int kk = 0;
        if constexpr (GMODE == 2) {
            const long row = tile * BM + arow;
            const int rr = (int)(row < a.R ? row : a.R - 1);
            kk = a.kshift >= 0 ? (rr & (a.Kmax - 1)) : (rr % a.Kmax);
        }
#pragma unroll
        for (int i = 0; i < KB; ++i) {
            const int c = 4 * (8 * i + aq);
            const float4 mean = *reinterpret_cast<const float4 *>(cA + c), is = *reinterpret_cast<const float4 *>(cA + Kd + c);
            const float4 sc = *reinterpret_cast<const float4 *>(cA + 2 * Kd + c), m1 = *reinterpret_cast<const float4 *>(cA + 3 * Kd + c);
            const float4 m2 = *reinterpret_cast<const float4 *>(cA + 4 * Kd + c);
            float *dst = dYs + arow * LDY + c;
            // same expressions as train_gemm.hip's dy_value
            const float x0 = (py[i].x - mean.x) * is.x, x1 = (py[i].y - mean.y) * is.y;
            const float x2 = (py[i].z - mean.z) * is.z, x3 = (py[i].w - mean.w) * is.w;
            float g0 = pg[i].x, g1 = pg[i].y, g2 = pg[i].z, g3 = pg[i].w;
            if constexpr (GMODE == 2) {
                const float4 ga = *reinterpret_cast<const float4 *>(cA + 5 * Kd + c), be = *reinterpret_cast<const float4 *>(cA + 6 * Kd + c);
                g0 = (par[i].x == kk && x0 * ga.x + be.x > 0.f) ? g0 : 0.f;  // routed to the arg-max row, [relu(BN(y)) > 0]
                g1 = (par[i].y == kk && x1 * ga.y + be.y > 0.f) ? g1 : 0.f;
                g2 = (par[i].z == kk && x2 * ga.z + be.z > 0.f) ? g2 : 0.f;
                g3 = (par[i].w == kk && x3 * ga.w + be.w > 0.f) ? g3 : 0.f;
            }
            if constexpr (GMODE == 1) {
                const float4 ga = *reinterpret_cast<const float4 *>(cA + 5 * Kd + c), be = *reinterpret_cast<const float4 *>(cA + 6 * Kd + c);
                g0 = (x0 * ga.x + be.x > 0.f) ? g0 : 0.f;  // [relu(BN(y)) > 0], torch's evaluation order
                g1 = (x1 * ga.y + be.y > 0.f) ? g1 : 0.f;
                g2 = (x2 * ga.z + be.z > 0.f) ? g2 : 0.f;
                g3 = (x3 * ga.w + be.w > 0.f) ? g3 : 0.f;
            }
            const float d0 = sc.x * (g0 - m1.x - x0 * m2.x);
            const float d1 = sc.y * (g1 - m1.y - x1 * m2.y);
            const float d2 = sc.z * (g2 - m1.z - x2 * m2.z);
            const float d3 = sc.w * (g3 - m1.w - x3 * m2.w);
            dst[0] = v ? d0 : 0.f; dst[1] = v ? d1 : 0.f; dst[2] = v ? d2 : 0.f; dst[3] = v ? d3 : 0.f;
        }
        const float4 pm = *reinterpret_cast<const float4 *>(cP + 4 * cq), pis = *reinterpret_cast<const float4 *>(cP + N + 4 * cq);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int r = rg + P::RG * i;
            const bool vr = tile * BM + r < a.R;
            float4 x;
            x.x = vr ? (ph[i].x - pm.x) * pis.x : 0.f;
            x.y = vr ? (ph[i].y - pm.y) * pis.y : 0.f;
            x.z = vr ? (ph[i].z - pm.z) * pis.z : 0.f;
            x.w = vr ? (ph[i].w - pm.w) * pis.w : 0.f;
            if constexpr (kKeepX) xh[i] = x;
            *reinterpret_cast<float4 *>(Xs + r * LDX + 4 * cq) = x;
        }
    };

    // data gradient: this wave's 32 x 32 block of the tile and its slice of the reduction
    const int db = wave % P::DBLK, ksd = wave / P::DBLK, rblk = db / NB, nbd = db % NB;
    constexpr int steps_d = P::steps_d, U = P::U;
    const float *wp = a.W + (size_t)(kh + 2 * ksd * steps_d) * a.ldw + nbd * 32 + l31;
    float wv0[P::WLDS ? 1 : U], wv1[P::WLDS ? 1 : U];
    auto load_w = [&](float *wv, int batch) {
#pragma unroll
        for (int u = 0; u < U; ++u) wv[u] = wp[(size_t)(2 * (batch * U + u)) * a.ldw];
    };
    if constexpr (!P::WLDS) load_w(wv0, 0);
    // weight gradient: this wave's blocks (sharing the column block nbw) and its slice of the tile's rows
    const int wb0 = P::WBLK >= 8 ? wave * P::WB : wave % P::WBLK;
    const int ksw = P::WBLK >= 8 ? 0 : wave / P::WBLK;
    const int nbw = wb0 / KB, kb0 = wb0 % KB;
    constexpr int steps_w = (BM / 2) / P::KS_W;
    const float wga = cP[2 * N + nbw * 32 + l31], wbe = cP[3 * N + nbw * 32 + l31];
    f32x16 accw[P::WB];
#pragma unroll
    for (int j = 0; j < P::WB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accw[j][r] = 0.f;
    float cs[4] = {0.f, 0.f, 0.f, 0.f}, cqs[4] = {0.f, 0.f, 0.f, 0.f};

    TGB_T(t_pro);
    TGB_ADD(0, t_pro, t_start);
    while (tile < T) {
        TGB_T(t0);
        commit(tile);
        TGB_T(t1);
        __syncthreads();
        TGB_T(t2);
        const long ntile = tile + nwg;
        if constexpr (P::WLDS) {  // nothing else of this tile touches global memory before the epilogue's stores
            prefetch(ntile);
        }
        {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float *ap = dYs + (rblk * 32 + l31) * LDY + kh + 2 * ksd * steps_d;
            if constexpr (P::WLDS) {
                const float *bp = Ws + (kh + 2 * ksd * steps_d) * N + nbd * 32 + l31;
#pragma unroll 8
                for (int s = 0; s < steps_d; ++s)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s], bp[2 * s * N], acc, 0, 0, 0);
            } else {
                // W_i is the same for every tile: the buffer for batch 0 is refilled by the last pair of this tile.
                // (Round 4, per-wave cycle counts -- scripts/probes/tgb_profile.py: the two waves of a SIMD leave this phase at
                // 39 k and 52 k cycles for 32 k of matrix instructions, the early ones then wait 13 k at the barrier: the phase
                // runs at ~66 % because W_i arrives from L2 one batch ahead only.  Requesting the LDS operand half a batch
                // ahead behind scheduling fences changed nothing -- the partner wave already covers that latency -- and cost
                // the narrow layers 4-10 %; a deeper W_i buffer does not fit the register file of the 128 -> 192 layer.)
                constexpr int NBATCH = steps_d / U;
#pragma unroll 1
                for (int b = 0; b < NBATCH; b += 2) {
                    load_w(wv1, b + 1);
#pragma unroll
                    for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * (b * U + u)], wv0[u], acc, 0, 0, 0);
                    load_w(wv0, (b + 2) % NBATCH);
#pragma unroll
                    for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * ((b + 1) * U + u)], wv1[u], acc, 0, 0, 0);
                }
            }
            float *gs = Gs + (ksd * BM + rblk * 32 + 4 * kh) * LDX + nbd * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) gs[((r & 3) + 8 * (r >> 2)) * LDX] = acc[r];
        }
        // the next tile's operands travel behind the weight-gradient instructions (which touch LDS only; the data gradient's
        // own global loads would otherwise queue behind these in the in-order load counter)
        TGB_T(t3);
        if constexpr (!P::WLDS) {
            prefetch(ntile);
        }
        {
            const float *xp = Xs + (kh + 2 * ksw * steps_w) * LDX + nbw * 32 + l31;
            const float *ap = dYs + (kh + 2 * ksw * steps_w) * LDY + kb0 * 32 + l31;
#pragma unroll 4
            for (int s = 0; s < steps_w; ++s) {
                const float bv = relu_nan(xp[2 * s * LDX] * wga + wbe);
#pragma unroll
                for (int j = 0; j < P::WB; ++j) {
                    const float av = ap[2 * s * LDY + 32 * j];
                    accw[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, accw[j], 0, 0, 0);
                }
            }
        }
        TGB_T(t4);
        __syncthreads();
        TGB_T(t5);
        {
            const bool full = tile * BM + BM <= a.R;
            const float4 pga = *reinterpret_cast<const float4 *>(cP + 2 * N + 4 * cq), pbe = *reinterpret_cast<const float4 *>(cP + 3 * N + 4 * cq);
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int r = rg + P::RG * i;
                float4 g = *reinterpret_cast<const float4 *>(Gs + r * LDX + 4 * cq);
#pragma unroll
                for (int k = 1; k < P::KS_D; ++k) {
                    const float4 t = *reinterpret_cast<const float4 *>(Gs + (k * BM + r) * LDX + 4 * cq);
                    g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w;
                }
                const long row = tile * BM + r;
                if (a.raw_out) {  // (workgroup-uniform) a column slice of a wider layer: the next slice finishes these rows
                    if (a.Gadd && (full || row < a.R)) {
                        const float4 t = *reinterpret_cast<const float4 *>(a.Gadd + row * a.ldga + 4 * cq);
                        g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w;
                    }
                    if (full || row < a.R) *reinterpret_cast<float4 *>(a.Gp + row * a.ldgp + 4 * cq) = g;
                    continue;
                }
                if (a.Gadd && (full || row < a.R)) {
                    const float4 t = *reinterpret_cast<const float4 *>(a.Gadd + row * a.ldga + 4 * cq);
                    g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w;
                }
                float4 x;
                if constexpr (kKeepX) x = xh[i];
                else x = *reinterpret_cast<const float4 *>(Xs + r * LDX + 4 * cq);
                g.x = (x.x * pga.x + pbe.x > 0.f) ? g.x : 0.f;  // [relu(BN(y)) > 0], torch's evaluation order
                g.y = (x.y * pga.y + pbe.y > 0.f) ? g.y : 0.f;
                g.z = (x.z * pga.z + pbe.z > 0.f) ? g.z : 0.f;
                g.w = (x.w * pga.w + pbe.w > 0.f) ? g.w : 0.f;
                cs[0] += g.x; cs[1] += g.y; cs[2] += g.z; cs[3] += g.w;   // rows beyond R: dY == 0 -> g == 0, xhat == 0
                cqs[0] += g.x * x.x; cqs[1] += g.y * x.y; cqs[2] += g.z * x.z; cqs[3] += g.w * x.w;
                if (full || row < a.R) *reinterpret_cast<float4 *>(a.Gp + row * a.ldgp + 4 * cq) = g;
            }
        }
        // no barrier here: the next commit overwrites dYs / Xs, whose readers all passed the barrier above; Gs is rewritten
        // only after the next commit's barrier, which every thread reaches after its epilogue reads
        TGB_T(t6);
        TGB_ADD(1, t1, t0); TGB_ADD(2, t2, t1); TGB_ADD(3, t3, t2); TGB_ADD(4, t4, t3); TGB_ADD(5, t5, t4); TGB_ADD(6, t6, t5);
        tile = ntile;
    }
    TGB_T(t_loop);

    // weight-gradient partial tile of this workgroup (and row slice)
    {
        float *out = a.partial + ((size_t)wg * P::KS_W + ksw) * (size_t)(Kd * N);
#pragma unroll
        for (int j = 0; j < P::WB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kd = (kb0 + j) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                out[(size_t)kd * N + nbw * 32 + l31] = accw[j][r];
            }
    }
    // BatchNorm-backward sums of layer i-1
    if (a.raw_out) return;  // (workgroup-uniform)
    __syncthreads();
    float *redS = Gs, *redQ = Gs + P::RG * N;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        redS[rg * N + 4 * cq + e] = cs[e];
        redQ[rg * N + 4 * cq + e] = cqs[e];
    }
    __syncthreads();
    if (tid < N) {
        double s = 0.0, q = 0.0;
        for (int r = 0; r < P::RG; ++r) {
            s += (double)redS[r * N + tid];
            q += (double)redQ[r * N + tid];
        }
        double *dst = a.sums_bwd_p + (size_t)(wg % kBnRep) * 2 * N;
        unsafeAtomicAdd(dst + tid, s);
        unsafeAtomicAdd(dst + N + tid, q);
    }
#ifdef PN2_TGB_PROFILE
    if (lane == 0 && a.prof) {  // lane 0 of EVERY wave: [workgroup][wave][8 phases]
        pacc[7] = clock64() - t_loop;
        for (int k = 0; k < 8; ++k) a.prof[((size_t)wg * 8 + wave) * 8 + k] = pacc[k];
    }
#endif
}

template <int NB, int KB, int GMODE>
__global__ void __launch_bounds__(kT, (NB == 1 && KB <= 2 ? 4 : 2))  // sa1's layers: two workgroups per CU (<= 128 registers)
tg_bwd_kernel(BwdArgs a) {
    tg_bwd_body<NB, KB, GMODE>(a, (int)blockIdx.x, (int)gridDim.x);
}

template <int NB, int KB, int GMODE>
__global__ void __launch_bounds__(kT, (NB == 1 && KB <= 2 ? 4 : 2))
tg_bwd_pair_kernel(BwdArgs a0, BwdArgs a1, int n0) {
    if ((int)blockIdx.x < n0) tg_bwd_body<NB, KB, GMODE>(a0, (int)blockIdx.x, n0);
    else tg_bwd_body<NB, KB, GMODE>(a1, (int)blockIdx.x - n0, (int)gridDim.x - n0);
}


// ---------------------------------------------------------------------------------------------------------------------------
// Round 5: the same layer backward with the data gradient's B operand (W_i) in REGISTERS (C_{i-1} in {64, 128}).
//
// Per-wave cycle counts of the kernel above (profiles/r04_tgb_phase_cycles_per_wave.json) show where its time goes on the
// 128-channel layers: the data-gradient phase runs at 66 % of the matrix rate (W_i streams from L2 one batch of 16 instructions
// ahead; a deeper buffer does not fit: a wave's 32 x 32 output block needs C_i / 2 = 96 registers of W_i for the 32x32x2
// instruction), and the first wave of a SIMD then waits 17 k cycles at the barrier for the second.  A first attempt this round
// gave the waves two ROLES (four hold W_i and own the data gradient, four own the weight gradient, both concurrently): the
// data-gradient waves then ran at the full matrix rate, but a wave that issues VALU / LDS / store instructions beside a
// co-resident wave streaming fp32 MFMAs gets about one issue slot per matrix instruction -- their accumulator-direct epilogue
// took 20 k cycles per tile -- and the sum was no faster (profiles/r05_misc_measurements.md).  What is kept from it: the
// register-resident W_i.  With v_mfma_f32_16x16x4_f32 a wave's output block is 16 columns wide, so its slice of W_i is
// C_i / 4 = 32 .. 48 registers (8 waves x 16 columns = the 128 columns of the tile; 64 columns: 4 slices x 2 row halves), the
// A fragments are 16-byte LDS reads (one per four instructions, k-permuted like sa_fused.hip's), and every wave still takes
// part in both matrix phases, the staging tile and the float4 epilogue exactly as above.
template <int NB, int KB>
struct Plan2 {
    static constexpr int N = 32 * NB, Kd = 32 * KB;
    static constexpr int LDY = Kd + 4;           // float4 rows; 16 lanes x 4 k-groups of a fragment read cover every bank 4 times
    static constexpr int LDX = N + 4;
    static constexpr int CS = N / 16;            // 16-column slices of the data-gradient tile (8 | 4)
    static constexpr int RH = 8 / CS;            // row groups (1 | 2): wave = (row group, column slice)
    static constexpr int RT = 4 / RH;            // 16-row tiles per wave (4 | 2)
    static constexpr int NQ = Kd / 16;           // k groups: one 16-byte A read and four matrix instructions per row tile each
    static constexpr int WBLK = NB * KB;
    static constexpr int KS_W = WBLK >= 8 ? 1 : 8 / WBLK;
    static constexpr int WB = WBLK >= 8 ? WBLK / 8 : 1;
    static constexpr int QN = N / 4, RG = kT / QN;
    static constexpr int lds_floats = 7 * Kd + 4 * N + BM * LDY + 2 * BM * LDX;
    static_assert(NB == 2 || NB == 4, "C_{i-1} in {64, 128}");
    static_assert(WBLK < 8 || WBLK % 8 == 0, "weight-gradient blocks must split evenly over 8 waves");
    static_assert(2 * RG * N <= BM * LDX, "the final column-sum staging reuses the data-gradient staging");
};

template <int NB, int KB, int GMODE>
__device__ __forceinline__ void tg_bwd2_body(const BwdArgs &a, const int wg, const int nwg) {
    using P = Plan2<NB, KB>;
    constexpr int N = P::N, Kd = P::Kd, LDY = P::LDY, LDX = P::LDX;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *cA = lds;                        // [7][Kd]
    float *cP = cA + 7 * Kd;                // [4][N]
    float *dYs = cP + 4 * N;                // [BM][LDY]
    float *Xs = dYs + BM * LDY;             // [BM][LDX]  xhat_{i-1}
    float *Gs = Xs + BM * LDX;              // [BM][LDX]  data gradient of the tile, for the float4 epilogue
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5, li = lane & 15, g4 = lane >> 4;
#ifdef PN2_TGB_PROFILE
    long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long pacc2[4] = {0, 0, 0, 0};  // prologue: loads issued | constants | W_i staged | barrier
#endif
    TGB_T(t_start);

    const long T = (a.R + BM - 1) / BM;
    const int arow = 8 * wave + (lane >> 3), aq = lane & 7;
    const int cq = tid % P::QN, rg = tid / P::QN;
    constexpr bool kLateG = GMODE == 2;  // dout / arg are L2 hits (one row serves Kmax rows of the tile): fetched by the commit
    // (routed source: dout / arg arrive in two halves -- half of their 2 x C_i / 32 registers live at a time)
    constexpr int GH = kLateG && KB >= 4 ? KB / 2 : KB;
    constexpr bool kEarlyG = kLateG && !(NB == 4 && KB == 6);  // (the 128 -> 192 kernel has no registers to carry it across the epilogue)
    float4 pg[GH], py[KB], ph[NB];
    int4 par[GMODE == 2 ? GH : 1];
    auto fetch_g = [&](long tile, int i0) {
        long row = tile * BM + arow;
        row = row < a.R ? row : a.R - 1;
        const long grow = GMODE == 2 ? (a.kshift >= 0 ? ((int)row >> a.kshift) : ((int)row / a.Kmax)) : row;
#pragma unroll
        for (int i = 0; i < GH; ++i) {
            pg[i] = *reinterpret_cast<const float4 *>(a.G + grow * a.ldg + 4 * (8 * (i0 + i) + aq));
            if constexpr (GMODE == 2) par[i] = *reinterpret_cast<const int4 *>(a.arg + grow * a.ldarg + 4 * (8 * (i0 + i) + aq));
        }
    };
    auto prefetch = [&](long tile) {  // G_i (dense sources) and Y_i: requested right behind the commit's barrier
        long row = tile * BM + arow;
        row = row < a.R ? row : a.R - 1;
        if constexpr (!kLateG) fetch_g(tile, 0);
#pragma unroll
        for (int i = 0; i < KB; ++i) py[i] = *reinterpret_cast<const float4 *>(a.Y + row * a.ldy + 4 * (8 * i + aq));
    };
    auto prefetch_p = [&](long tile) {  // Y_{i-1}: requested between the two matrix phases (not live during the first)
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            long r = tile * BM + rg + P::RG * i;
            r = r < a.R ? r : a.R - 1;
            ph[i] = *reinterpret_cast<const float4 *>(a.Yp + r * a.ldyp + 4 * cq);
        }
    };
    // data gradient: this wave's 16 columns and 16-row tiles rt0 .. rt0 + RT - 1
    const int dcs = wave % P::CS, rt0 = (wave / P::CS) * P::RT;
    // W_i, this wave's slice, as the B operand of instruction (q, j): lane (li, g4) holds W_i[16 q + 4 g4 + j][16 dcs + li] -- requested
    // first (one memory round trip with the first tile's operands and the constants), used after the first commit
    // It reaches the registers through LDS: 16-byte coalesced loads of the whole matrix by the whole workgroup (the region of
    // the tiles is free before the first commit), then every lane picks its C_i / 4 values.  (Fetched directly -- 4-byte
    // loads, 64-byte segments, the same addresses in the same order on every CU at the same moment -- the prologue of these
    // kernels took 16 k cycles instead of 7 k: profiles/r05_misc_measurements.md.)  Workgroups start at rotated offsets.
    float wreg[4 * P::NQ];
    constexpr int WQ = Kd * N / 4, WCNT = WQ / kT, LDW = N + 4;  // staging rows of N + 4 floats: the four k slots of a fragment hit disjoint banks
    static_assert(WQ % kT == 0 && Kd * LDW <= BM * LDY + 2 * BM * LDX, "W_i staging fits the tile region");
    f32x4 wst[WCNT];
    const int wrot = (int)(((unsigned)wg * 1103u) % (unsigned)WQ);
#pragma unroll
    for (int i = 0; i < WCNT; ++i) {
        int e = tid + i * kT + wrot;
        e = e >= WQ ? e - WQ : e;
        const int kd = e / (N / 4), q = e % (N / 4);
#ifdef TGB2_PROBE_NO_W   /* timing probe: what W_i costs the prologue (results are wrong) */
        wst[i] = (f32x4){1.f, 1.f, 1.f, 1.f};
#else
        wst[i] = *reinterpret_cast<const f32x4 *>(a.W + (size_t)kd * a.ldw + 4 * q);
#endif
    }
    long tile = wg;
    // The first tile's operands are requested AFTER the constants' own loads below, not before them: loads return in order, so
    // with 82 KB of HBM rows per workgroup queued in front, the fp64 sums the constants are derived from (and with them the
    // whole staging chain: constants -> barrier -> W_i to registers -> barrier) waited for the slowest load of the launch.
    // Measured on the 128 -> 192 keypoint-query layer: prologue 17.6 k -> 9.3 k cycles, the kernel 115.6 k -> 106.7 k, and the
    // first commit does not wait any longer for it (profiles/r05_misc_measurements.md; TGB2_PROBE_EARLY_TILE: the old order).
#ifdef TGB2_PROBE_EARLY_TILE
    if (tile < T) {
        prefetch(tile);
        prefetch_p(tile);
        if constexpr (kEarlyG) fetch_g(tile, 0);
    }
#endif
    TGB_T(t_p1);
    {
        const float inv_r = (float)(1.0 / (double)a.R);
        for (int c = tid; c < Kd; c += kT) {
            double sa = 0.0, sb = 0.0;
            for (int r = 0; r < kBnRep; ++r) {
                sa += a.sums_bwd[(size_t)r * 2 * a.sums_ld + c];
                sb += a.sums_bwd[(size_t)r * 2 * a.sums_ld + a.sums_ld + c];
            }
            const float is = a.invstd[c];
            cA[c] = a.mean[c]; cA[Kd + c] = is; cA[2 * Kd + c] = a.gamma[c] * is;
            cA[3 * Kd + c] = (float)sa * inv_r; cA[4 * Kd + c] = (float)sb * inv_r;
            if constexpr (GMODE != 0) { cA[5 * Kd + c] = a.gamma[c]; cA[6 * Kd + c] = a.beta[c]; }
        }
        for (int c = tid; c < N; c += kT) {
            cP[c] = a.mean_p[c]; cP[N + c] = a.invstd_p[c]; cP[2 * N + c] = a.gamma_p[c]; cP[3 * N + c] = a.beta_p[c];
        }
        if (wg == 0)
            for (int e = tid; e < Kd * N; e += kT) a.dW[e] = 0.f;
    }
#ifndef TGB2_PROBE_EARLY_TILE
    if (tile < T) {
        prefetch(tile);
        prefetch_p(tile);
        if constexpr (kEarlyG) fetch_g(tile, 0);
    }
#endif
    TGB_T(t_p2);
    TGB_ADD2(0, t_p1, t_start); TGB_ADD2(1, t_p2, t_p1);
    {
#pragma unroll
        for (int i = 0; i < WCNT; ++i) {
            int e = tid + i * kT + wrot;
            e = e >= WQ ? e - WQ : e;
            const int kd = e / (N / 4), q = e % (N / 4);
            *reinterpret_cast<f32x4 *>(dYs + kd * LDW + 4 * q) = wst[i];
        }
    }
    TGB_T(t_p3);
    __syncthreads();
    TGB_T(t_p4);
    TGB_ADD2(2, t_p3, t_p2); TGB_ADD2(3, t_p4, t_p3);
    {
        const float *wp = dYs + (4 * g4) * LDW + dcs * 16 + li;
#pragma unroll
        for (int q = 0; q < P::NQ; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) wreg[4 * q + j] = wp[(16 * q + j) * LDW];
    }
    __syncthreads();  // the staging region becomes the tiles

    auto commit = [&](long tile) {
        const bool v = tile * BM + arow < a.R;
        // (routed source: the first half of dout / arg was requested behind the previous tile's weight-gradient phase -- the
        // registers are free there -- and the second half follows once the first is consumed)
        if constexpr (kLateG && !kEarlyG) fetch_g(tile, 0);
        int kk = 0;
        if constexpr (GMODE == 2) {
            const long row = tile * BM + arow;
            const int rr = (int)(row < a.R ? row : a.R - 1);
            kk = a.kshift >= 0 ? (rr & (a.Kmax - 1)) : (rr % a.Kmax);
        }
#pragma unroll
        for (int i = 0; i < KB; ++i) {
            if constexpr (kLateG && GH < KB) {
                if (i == GH) fetch_g(tile, GH);
            }
            const int gi = i % GH;
            const int c = 4 * (8 * i + aq);
            const float4 mean = *reinterpret_cast<const float4 *>(cA + c), is = *reinterpret_cast<const float4 *>(cA + Kd + c);
            const float4 sc = *reinterpret_cast<const float4 *>(cA + 2 * Kd + c), m1 = *reinterpret_cast<const float4 *>(cA + 3 * Kd + c);
            const float4 m2 = *reinterpret_cast<const float4 *>(cA + 4 * Kd + c);
            // same expressions as train_gemm.hip's dy_value
            const float x0 = (py[i].x - mean.x) * is.x, x1 = (py[i].y - mean.y) * is.y;
            const float x2 = (py[i].z - mean.z) * is.z, x3 = (py[i].w - mean.w) * is.w;
            float g0 = pg[gi].x, g1 = pg[gi].y, g2 = pg[gi].z, g3 = pg[gi].w;
            if constexpr (GMODE == 2) {
                const float4 ga = *reinterpret_cast<const float4 *>(cA + 5 * Kd + c), be = *reinterpret_cast<const float4 *>(cA + 6 * Kd + c);
                g0 = (par[gi].x == kk && x0 * ga.x + be.x > 0.f) ? g0 : 0.f;  // routed to the arg-max row, [relu(BN(y)) > 0]
                g1 = (par[gi].y == kk && x1 * ga.y + be.y > 0.f) ? g1 : 0.f;
                g2 = (par[gi].z == kk && x2 * ga.z + be.z > 0.f) ? g2 : 0.f;
                g3 = (par[gi].w == kk && x3 * ga.w + be.w > 0.f) ? g3 : 0.f;
            }
            if constexpr (GMODE == 1) {
                const float4 ga = *reinterpret_cast<const float4 *>(cA + 5 * Kd + c), be = *reinterpret_cast<const float4 *>(cA + 6 * Kd + c);
                g0 = (x0 * ga.x + be.x > 0.f) ? g0 : 0.f;  // [relu(BN(y)) > 0], torch's evaluation order
                g1 = (x1 * ga.y + be.y > 0.f) ? g1 : 0.f;
                g2 = (x2 * ga.z + be.z > 0.f) ? g2 : 0.f;
                g3 = (x3 * ga.w + be.w > 0.f) ? g3 : 0.f;
            }
            float4 d;
            d.x = v ? sc.x * (g0 - m1.x - x0 * m2.x) : 0.f;
            d.y = v ? sc.y * (g1 - m1.y - x1 * m2.y) : 0.f;
            d.z = v ? sc.z * (g2 - m1.z - x2 * m2.z) : 0.f;
            d.w = v ? sc.w * (g3 - m1.w - x3 * m2.w) : 0.f;
            *reinterpret_cast<float4 *>(dYs + arow * LDY + c) = d;
        }
        const float4 pm = *reinterpret_cast<const float4 *>(cP + 4 * cq), pis = *reinterpret_cast<const float4 *>(cP + N + 4 * cq);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int r = rg + P::RG * i;
            const bool vr = tile * BM + r < a.R;
            float4 x;
            x.x = vr ? (ph[i].x - pm.x) * pis.x : 0.f;
            x.y = vr ? (ph[i].y - pm.y) * pis.y : 0.f;
            x.z = vr ? (ph[i].z - pm.z) * pis.z : 0.f;
            x.w = vr ? (ph[i].w - pm.w) * pis.w : 0.f;
            *reinterpret_cast<float4 *>(Xs + r * LDX + 4 * cq) = x;
        }
    };

    // weight gradient: this wave's blocks (sharing the column block nbw) and its slice of the tile's rows -- as above
    const int wb0 = P::WBLK >= 8 ? wave * P::WB : wave % P::WBLK;
    const int ksw = P::WBLK >= 8 ? 0 : wave / P::WBLK;
    const int nbw = wb0 / KB, kb0 = wb0 % KB;
    constexpr int steps_w = (BM / 2) / P::KS_W;
    const float wga = cP[2 * N + nbw * 32 + l31], wbe = cP[3 * N + nbw * 32 + l31];
    f32x16 accw[P::WB];
#pragma unroll
    for (int j = 0; j < P::WB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accw[j][r] = 0.f;
    float cs[4] = {0.f, 0.f, 0.f, 0.f}, cqs[4] = {0.f, 0.f, 0.f, 0.f};

    TGB_T(t_pro);
    TGB_ADD(0, t_pro, t_start);
    while (tile < T) {
        TGB_T(t0);
        commit(tile);
        TGB_T(t1);
        __syncthreads();
        TGB_T(t2);
        const long ntile = tile + nwg;
        prefetch(ntile);  // (unconditional: rows clamp to the last one) nothing else of this tile reads global memory before the epilogue
        {
            // data gradient: acc[t] (16 x 16: row 4 g4 + r, column li) of row tile rt0 + t; instruction (q, j) multiplies
            // dY[.][16 q + 4 g4 + j] by W_i[16 q + 4 g4 + j][.]: every k once, in the order the fragments are laid out
            f32x4 acc[P::RT];
#pragma unroll
            for (int t = 0; t < P::RT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float *ap = dYs + (rt0 * 16 + li) * LDY + 4 * g4;
            // (the 128 -> 192 kernel has no registers left for fragments a whole k group ahead: its reads are issued where the
            // compiler places them, the SIMD's other wave covers their latency)
            constexpr bool kAhead = !(NB == 4 && KB == 6);
            float4 an[kAhead ? P::RT : 1];
            if constexpr (kAhead) {
#pragma unroll
                for (int t = 0; t < P::RT; ++t) an[t] = *reinterpret_cast<const float4 *>(ap + t * 16 * LDY);
            }
#pragma unroll
            for (int q = 0; q < P::NQ; ++q) {
                float4 av[P::RT];
#pragma unroll
                for (int t = 0; t < P::RT; ++t) {
                    if constexpr (kAhead) av[t] = an[t];
                    else av[t] = *reinterpret_cast<const float4 *>(ap + t * 16 * LDY + 16 * q);
                }
                if (kAhead && q + 1 < P::NQ) {  // the next k group's fragments travel behind this group's instructions
#pragma unroll
                    for (int t = 0; t < P::RT; ++t) an[t] = *reinterpret_cast<const float4 *>(ap + t * 16 * LDY + 16 * (q + 1));
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int t = 0; t < P::RT; ++t) {
                        const float x = j == 0 ? av[t].x : j == 1 ? av[t].y : j == 2 ? av[t].z : av[t].w;
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, wreg[4 * q + j], acc[t], 0, 0, 0);
                    }
            }
            float *gs = Gs + ((rt0 * 16 + 4 * g4) * LDX + dcs * 16 + li);
#pragma unroll
            for (int t = 0; t < P::RT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) gs[(t * 16 + r) * LDX] = acc[t][r];
        }
        TGB_T(t3);
        prefetch_p(ntile);
        {
            const float *xp = Xs + (kh + 2 * ksw * steps_w) * LDX + nbw * 32 + l31;
            const float *ap = dYs + (kh + 2 * ksw * steps_w) * LDY + kb0 * 32 + l31;
#pragma unroll 4
            for (int s = 0; s < steps_w; ++s) {
                const float bv = relu_nan(xp[2 * s * LDX] * wga + wbe);
#pragma unroll
                for (int j = 0; j < P::WB; ++j) {
                    const float av = ap[2 * s * LDY + 32 * j];
                    accw[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, accw[j], 0, 0, 0);
                }
            }
        }
        TGB_T(t4);
        if constexpr (kEarlyG) {
            fetch_g(ntile, 0);
        }
        __syncthreads();
        TGB_T(t5);
        {
            const bool full = tile * BM + BM <= a.R;
            const float4 pga = *reinterpret_cast<const float4 *>(cP + 2 * N + 4 * cq), pbe = *reinterpret_cast<const float4 *>(cP + 3 * N + 4 * cq);
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int r = rg + P::RG * i;
                float4 g = *reinterpret_cast<const float4 *>(Gs + r * LDX + 4 * cq);
                const long row = tile * BM + r;
                if (a.Gadd && (full || row < a.R)) {
                    const float4 t = *reinterpret_cast<const float4 *>(a.Gadd + row * a.ldga + 4 * cq);
                    g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w;
                }
                if (a.raw_out) {  // (workgroup-uniform) a column slice of a wider layer: the next slice finishes these rows
                    if (full || row < a.R) *reinterpret_cast<float4 *>(a.Gp + row * a.ldgp + 4 * cq) = g;
                    continue;
                }
                const float4 x = *reinterpret_cast<const float4 *>(Xs + r * LDX + 4 * cq);
                g.x = (x.x * pga.x + pbe.x > 0.f) ? g.x : 0.f;  // [relu(BN(y)) > 0], torch's evaluation order
                g.y = (x.y * pga.y + pbe.y > 0.f) ? g.y : 0.f;
                g.z = (x.z * pga.z + pbe.z > 0.f) ? g.z : 0.f;
                g.w = (x.w * pga.w + pbe.w > 0.f) ? g.w : 0.f;
                cs[0] += g.x; cs[1] += g.y; cs[2] += g.z; cs[3] += g.w;   // rows beyond R: dY == 0 -> g == 0, xhat == 0
                cqs[0] += g.x * x.x; cqs[1] += g.y * x.y; cqs[2] += g.z * x.z; cqs[3] += g.w * x.w;
                if (full || row < a.R) *reinterpret_cast<float4 *>(a.Gp + row * a.ldgp + 4 * cq) = g;
            }
        }
        // no barrier here: the next commit overwrites dYs / Xs at the elements this thread itself read in the epilogue (same
        // (row, quad) mapping), other readers passed the barrier above; Gs is rewritten only after the next commit's barrier
        TGB_T(t6);
        TGB_ADD(1, t1, t0); TGB_ADD(2, t2, t1); TGB_ADD(3, t3, t2); TGB_ADD(4, t4, t3); TGB_ADD(5, t5, t4); TGB_ADD(6, t6, t5);
        tile = ntile;
    }
    TGB_T(t_loop);

    // (the indices of the code below are re-derived from an opaque copy of the thread index: left to itself hipcc forms the
    // output addresses BEFORE the tile loop and spills them across it on the 128 -> 192 kernel)
    int tid_e = tid;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63, wave_e = tid_e >> 6, l31_e = lane_e & 31, kh_e = lane_e >> 5;
    const int wb0_e = P::WBLK >= 8 ? wave_e * P::WB : wave_e % P::WBLK, ksw_e = P::WBLK >= 8 ? 0 : wave_e / P::WBLK;
    const int nbw_e = wb0_e / KB, kb0_e = wb0_e % KB, cq_e = tid_e % P::QN, rg_e = tid_e / P::QN;
    {  // weight-gradient partial tile of this workgroup (and row slice)
        float *out = a.partial + ((size_t)wg * P::KS_W + ksw_e) * (size_t)(Kd * N);
#pragma unroll
        for (int j = 0; j < P::WB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kd = (kb0_e + j) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh_e;
                out[(size_t)kd * N + nbw_e * 32 + l31_e] = accw[j][r];
            }
    }
    if (a.raw_out) return;  // (workgroup-uniform)
    __syncthreads();
    float *redS = Gs, *redQ = Gs + P::RG * N;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        redS[rg_e * N + 4 * cq_e + e] = cs[e];
        redQ[rg_e * N + 4 * cq_e + e] = cqs[e];
    }
    __syncthreads();
    if (tid_e < N) {
        double s = 0.0, q = 0.0;
        for (int r = 0; r < P::RG; ++r) {
            s += (double)redS[r * N + tid_e];
            q += (double)redQ[r * N + tid_e];
        }
        double *dst = a.sums_bwd_p + (size_t)(wg % kBnRep) * 2 * N;
        unsafeAtomicAdd(dst + tid_e, s);
        unsafeAtomicAdd(dst + N + tid_e, q);
    }
#ifdef PN2_TGB_PROFILE
    if (lane == 0 && a.prof) {
        pacc[7] = clock64() - t_loop;
        for (int k = 0; k < 8; ++k) a.prof[((size_t)wg * 8 + wave) * 8 + k] = pacc[k];
        for (int k = 0; k < 4; ++k) a.prof[(size_t)1024 * 64 + ((size_t)wg * 8 + wave) * 4 + k] = pacc2[k];
    }
#endif
}

template <int NB, int KB, int GMODE>
__global__ void __launch_bounds__(kT, 2) tg_bwd2_kernel(BwdArgs a) {
    tg_bwd2_body<NB, KB, GMODE>(a, (int)blockIdx.x, (int)gridDim.x);
}

template <int NB, int KB, int GMODE>
__global__ void __launch_bounds__(kT, 2) tg_bwd2_pair_kernel(BwdArgs a0, BwdArgs a1, int n0) {
    if ((int)blockIdx.x < n0) tg_bwd2_body<NB, KB, GMODE>(a0, (int)blockIdx.x, n0);
    else tg_bwd2_body<NB, KB, GMODE>(a1, (int)blockIdx.x - n0, (int)gridDim.x - n0);
}

// the (C_{i-1} / 32, C_i / 32) pairs the two-role kernel is instantiated for
#define PN2_TGB2_SHAPES(X) X(2, 2) X(2, 4) X(4, 4) X(4, 6)
static bool g_tgb2 = true;  // (pn2x_tg_bwd_set_variant(0): the round-4 kernel for every shape -- tests of both, A/B benches)
}  // namespace tgb
}  // namespace pn2
// 1: the register-resident-W kernel where it is instantiated (default), 0: the round-4 kernel everywhere.  Process-wide; the
// partial-tile counts pn2x_tg_bwd_partials reports follow it, so switch between whole backward passes only (tests, A/B benches).
extern "C" int pn2x_tg_bwd_set_variant(int v2) {
    pn2::tgb::g_tgb2 = v2 != 0;
    return PN2_OK;
}
namespace pn2 {
namespace tgb {
static bool v2_shape(int nb, int kb) {
    if (!g_tgb2) return false;
#define X(NB_, KB_) if (nb == NB_ && kb == KB_) return true;
    PN2_TGB2_SHAPES(X)
#undef X
    return false;
}
template <int NB, int KB>
static size_t lds2_of() { return (size_t)Plan2<NB, KB>::lds_floats * sizeof(float); }

struct Shape {
    int nb, kb, ks_w;
    size_t lds;
    bool v2;   // the register-resident-W kernel of round 5 (one workgroup per CU)
};
template <int NB, int KB>
static Shape shape_of() { return Shape{NB, KB, Plan<NB, KB>::KS_W, (size_t)Plan<NB, KB>::lds_floats * sizeof(float), false}; }

static int kshift_of(int k) {
    if (k < 1 || (k & (k - 1))) return -1;
    int s = 0;
    while ((1 << s) < k) ++s;
    return s;
}

// the instantiated (C_{i-1} / 32, C_i / 32) pairs
#define PN2_TGB_SHAPES(X) X(1, 1) X(1, 2) X(1, 4) X(2, 1) X(2, 2) X(2, 4) X(4, 2) X(4, 4) X(4, 6)

static bool find_shape(int c_in, int c_out, Shape &s) {
    if (c_in % 32 || c_out % 32) return false;
    const int nb = c_in / 32, kb = c_out / 32;
    if (v2_shape(nb, kb)) {
#define X(NB_, KB_) if (nb == NB_ && kb == KB_) { s = Shape{NB_, KB_, Plan2<NB_, KB_>::KS_W, lds2_of<NB_, KB_>(), true}; return true; }
        PN2_TGB2_SHAPES(X)
#undef X
    }
#define X(NB_, KB_) if (nb == NB_ && kb == KB_) { s = shape_of<NB_, KB_>(); return true; }
    PN2_TGB_SHAPES(X)
#undef X
    return false;
}

static int grid_of(long rows, const Shape &s) {
    const long tiles = (rows + BM - 1) / BM;
    // persistent workgroups: one per CU where the LDS footprint admits only one, two otherwise
    // (two workgroups of the register-resident-W kernel per CU were measured in round 5 and are slower)
    const long cap = (long)num_compute_units() * (!s.v2 && s.lds * 2 <= 160 * 1024 ? 2 : 1);
    return (int)(tiles < cap ? tiles : cap);
}

}  // namespace tgb
}  // namespace pn2

using namespace pn2;
using namespace pn2::tgb;

#ifdef PN2_TGB_PROFILE
static long long *g_prof = nullptr;
extern "C" void pn2x_tg_bwd_set_profile(long long *p) { g_prof = p; }
#endif

extern "C" int pn2x_tg_bwd_supported(int c_in, int c_out) {
    Shape s;
    return find_shape(c_in, c_out, s) ? 1 : 0;
}

extern "C" int pn2x_tg_bwd_partials(long rows, int c_out, int c_in) {
    Shape s;
    if (rows < 1 || !find_shape(c_in, c_out, s)) return -1;
    return grid_of(rows, s) * s.ks_w;
}

extern "C" int pn2x_tg_bwd(long rows, int n, int k, int gmode, const float *g, int ldg, const int *arg, int kmax, const float *yi,
                           int ldyi, const float *mean_i, const float *invstd_i, const float *gamma_i, const float *beta_i,
                           const double *sums_bwd_i, const float *w, int ldw,
                           const float *yp, int ldyp, const float *mean_p, const float *invstd_p, const float *gamma_p,
                           const float *beta_p, float *gp, int ldgp, double *sums_bwd_p, float *partial, long partial_floats,
                           float *dw, void *stream) {
    return pn2x_tg_bwd_slice(rows, n, k, gmode, g, ldg, arg, ldg, kmax, yi, ldyi, mean_i, invstd_i, gamma_i, beta_i, sums_bwd_i, n, w, ldw, yp,
                             ldyp, mean_p, invstd_p, gamma_p, beta_p, gp, ldgp, sums_bwd_p, partial, partial_floats, dw, nullptr, 0, 0,
                             stream);
}

// One column slice [c0, c0 + n) of a layer with sums_ld >= n channels: the caller passes g / yi / the per-channel vectors / sums_bwd_i /
// w / dw already offset to the slice; arg (gmode 2) has its own row stride ldarg (g may be a column block of a wider tensor).  g_add (row stride ldga, may alias gp): the data-gradient partial of earlier slices; raw_out: leave
// mask and sums to a later slice.
#define PN2_TGB_PARAMS(S)                                                                                                              \
    long rows##S, int n##S, int k##S, int gmode##S, const float *g##S, int ldg##S, const int *arg##S, int ldarg##S, int kmax##S,         \
        const float *yi##S, int ldyi##S, const float *mean_i##S, const float *invstd_i##S, const float *gamma_i##S,                    \
        const float *beta_i##S, const double *sums_bwd_i##S, int sums_ld##S, const float *w##S, int ldw##S, const float *yp##S,        \
        int ldyp##S, const float *mean_p##S, const float *invstd_p##S, const float *gamma_p##S, const float *beta_p##S, float *gp##S,  \
        int ldgp##S, double *sums_bwd_p##S, float *partial##S, long partial_floats##S, float *dw##S, const float *g_add##S, int ldga##S, \
        int raw_out##S
#define PN2_TGB_ARGS(S)                                                                                                                \
    rows##S, n##S, k##S, gmode##S, g##S, ldg##S, arg##S, ldarg##S, kmax##S, yi##S, ldyi##S, mean_i##S, invstd_i##S, gamma_i##S, beta_i##S, \
        sums_bwd_i##S, sums_ld##S, w##S, ldw##S, yp##S, ldyp##S, mean_p##S, invstd_p##S, gamma_p##S, beta_p##S, gp##S, ldgp##S,        \
        sums_bwd_p##S, partial##S, partial_floats##S, dw##S, g_add##S, ldga##S, raw_out##S

// validation + kernel arguments of one problem (grid-independent part)
static int fill_bwd(PN2_TGB_PARAMS(), BwdArgs &a, Shape &s) {
    if (rows < 1 || rows > 0x7fffffffL || !find_shape(k, n, s) || gmode < 0 || gmode > 2 || sums_ld < n) return PN2_EINVAL;
    if (gmode == 2 && (kmax < 1 || rows % kmax || ldarg < n || ldarg % 4)) return PN2_EINVAL;
    if ((gmode == 2 && !arg) || (gmode != 0 && !beta_i)) return PN2_ENULL;
    if (g_add && (ldga < k || ldga % 4 || (uintptr_t)g_add % 16)) return PN2_EINVAL;
    if ((uintptr_t)arg % 16) return PN2_EINVAL;
    if (ldg < n || ldg % 4 || ldyi < n || ldyi % 4 || ldw < k || ldyp < k || ldyp % 4 || ldgp < k || ldgp % 4) return PN2_EINVAL;
    if (!g || !yi || !mean_i || !invstd_i || !gamma_i || !sums_bwd_i || !w || !yp || !mean_p || !invstd_p || !gamma_p || !beta_p ||
        !gp || !sums_bwd_p || !partial || !dw)
        return PN2_ENULL;
    if (((uintptr_t)g | (uintptr_t)yi | (uintptr_t)yp | (uintptr_t)gp) % 16) return PN2_EINVAL;
    (void)partial_floats;
    a = BwdArgs{rows, g, ldg, arg, ldarg, kmax, kshift_of(kmax), yi, ldyi, mean_i, invstd_i, gamma_i, beta_i, sums_bwd_i, sums_ld, g_add, ldga, raw_out ? 1 : 0, w, ldw, yp, ldyp, mean_p, invstd_p, gamma_p, beta_p,
                gp, ldgp, sums_bwd_p, partial, dw
#ifdef PN2_TGB_PROFILE
                , g_prof
#endif
    };
    return PN2_OK;
}

extern "C" int pn2x_tg_bwd_slice(PN2_TGB_PARAMS(), void *stream) {
    Shape s;
    BwdArgs a;
    const int rc = fill_bwd(PN2_TGB_ARGS(), a, s);
    if (rc != PN2_OK) return rc;
    const int grid = grid_of(rows, s);
    if (partial_floats < (long)grid * s.ks_w * n * k) return PN2_ESCRATCH;
    hipStream_t st = (hipStream_t)stream;
    const int nb = k / 32, kb = n / 32;
#define PN2_TGB_LAUNCH(NB_, KB_, GM_)                                                                                  \
    do {                                                                                                              \
        static PerDeviceOnce once;                                                                                    \
        if (once.first_use())                                                                                         \
            (void)hipFuncSetAttribute((const void *)tg_bwd_kernel<NB_, KB_, GM_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)s.lds);                                                                    \
        hipLaunchKernelGGL((tg_bwd_kernel<NB_, KB_, GM_>), dim3(grid), dim3(kT), s.lds, st, a);                       \
    } while (0)
#define PN2_TGB2_LAUNCH(NB_, KB_, GM_)                                                                                 \
    do {                                                                                                              \
        static PerDeviceOnce once;                                                                                    \
        if (once.first_use())                                                                                         \
            (void)hipFuncSetAttribute((const void *)tg_bwd2_kernel<NB_, KB_, GM_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)s.lds);                                                                    \
        hipLaunchKernelGGL((tg_bwd2_kernel<NB_, KB_, GM_>), dim3(grid), dim3(kT), s.lds, st, a);                      \
    } while (0)
    if (s.v2) {
#define X(NB_, KB_)                                                                                                   \
    if (nb == NB_ && kb == KB_) {                                                                                     \
        if (gmode == 0) PN2_TGB2_LAUNCH(NB_, KB_, 0);                                                                 \
        else if (gmode == 1) PN2_TGB2_LAUNCH(NB_, KB_, 1);                                                            \
        else PN2_TGB2_LAUNCH(NB_, KB_, 2);                                                                            \
    }
        PN2_TGB2_SHAPES(X)
#undef X
        return check_launch();
    }
#undef PN2_TGB2_LAUNCH
#define X(NB_, KB_)                                                                                                   \
    if (nb == NB_ && kb == KB_) {                                                                                     \
        if (gmode == 0) PN2_TGB_LAUNCH(NB_, KB_, 0);                                                                  \
        else if (gmode == 1) PN2_TGB_LAUNCH(NB_, KB_, 1);                                                             \
        else PN2_TGB_LAUNCH(NB_, KB_, 2);                                                                             \
    }
    PN2_TGB_SHAPES(X)
#undef X
#undef PN2_TGB_LAUNCH
    return check_launch();
}

// the pair kernel is instantiated for the 128-channel layers of the keypoint-query modules only (128 -> 128, 128 -> 192)
extern "C" int pn2x_tg_bwd_pair_supported(int c_in, int c_out) { return (c_in == 128 && (c_out == 128 || c_out == 192)) ? 1 : 0; }

// Two problems of the same layer shape and gradient source in ONE launch (the two neighbourhood sizes of a keypoint-query
// module): the persistent workgroups are split in proportion to the tile counts.  n_partials[2] receives the number of partial
// weight-gradient tiles each problem wrote (its share of the grid x the row slices per workgroup): the caller's reduction must
// sum exactly those.  Each partial buffer must hold what the problem would need alone (pn2x_tg_bwd_partials).
extern "C" int pn2x_tg_bwd_slice_pair(PN2_TGB_PARAMS(0), PN2_TGB_PARAMS(1), int *n_partials, void *stream) {
    Shape s0, s1;
    BwdArgs a0, a1;
    if (!n_partials) return PN2_ENULL;
    int rc = fill_bwd(PN2_TGB_ARGS(0), a0, s0);
    if (rc != PN2_OK) return rc;
    rc = fill_bwd(PN2_TGB_ARGS(1), a1, s1);
    if (rc != PN2_OK) return rc;
    if (n0 != n1 || k0 != k1 || gmode0 != gmode1) return PN2_EINVAL;
    if (!pn2x_tg_bwd_pair_supported(k0, n0)) return PN2_ERANGE;
    if (s0.v2 && k0 == 128 && n0 == 192 && gmode0 != 2) {
        // (the pair form of the 128 -> 192 register-resident-W kernel with a dense source needs three registers more than a
        // wave has: that combination -- not one of this network's -- keeps the round-4 kernel; same partial-tile layout)
        s0 = s1 = shape_of<4, 6>();
    }
    const long t0 = (rows0 + BM - 1) / BM, t1 = (rows1 + BM - 1) / BM;
    long cap = (long)num_compute_units() * (!s0.v2 && s0.lds * 2 <= 160 * 1024 ? 2 : 1);
    if (cap < 2) cap = 2;
    long rounds = (t0 + t1 + cap - 1) / cap, wg0, wg1;
    for (;; ++rounds) {  // the same number of rounds for both shares
        wg0 = (t0 + rounds - 1) / rounds;
        wg1 = (t1 + rounds - 1) / rounds;
        if (wg0 + wg1 <= cap || rounds > t0 + t1) break;
    }
    if (partial_floats0 < wg0 * s0.ks_w * n0 * k0 || partial_floats1 < wg1 * s1.ks_w * n1 * k1) return PN2_ESCRATCH;
    n_partials[0] = (int)wg0 * s0.ks_w;
    n_partials[1] = (int)wg1 * s1.ks_w;
    hipStream_t st = (hipStream_t)stream;
    const int nb = k0 / 32, kb = n0 / 32, gmode = gmode0;
#define PN2_TGB_LAUNCH(NB_, KB_, GM_)                                                                                  \
    do {                                                                                                              \
        static PerDeviceOnce once;                                                                                    \
        if (once.first_use())                                                                                         \
            (void)hipFuncSetAttribute((const void *)tg_bwd_pair_kernel<NB_, KB_, GM_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)s0.lds);                                                                   \
        hipLaunchKernelGGL((tg_bwd_pair_kernel<NB_, KB_, GM_>), dim3((unsigned)(wg0 + wg1)), dim3(kT), s0.lds, st, a0, a1, (int)wg0); \
    } while (0)
#define X(NB_, KB_)                                                                                                   \
    if (nb == NB_ && kb == KB_) {                                                                                     \
        if (gmode == 0) PN2_TGB_LAUNCH(NB_, KB_, 0);                                                                  \
        else if (gmode == 1) PN2_TGB_LAUNCH(NB_, KB_, 1);                                                             \
        else PN2_TGB_LAUNCH(NB_, KB_, 2);                                                                             \
    }
#define PN2_TGB2_LAUNCH(NB_, KB_, GM_)                                                                                 \
    do {                                                                                                              \
        static PerDeviceOnce once;                                                                                    \
        if (once.first_use())                                                                                         \
            (void)hipFuncSetAttribute((const void *)tg_bwd2_pair_kernel<NB_, KB_, GM_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)s0.lds);                                                                   \
        hipLaunchKernelGGL((tg_bwd2_pair_kernel<NB_, KB_, GM_>), dim3((unsigned)(wg0 + wg1)), dim3(kT), s0.lds, st, a0, a1, (int)wg0); \
    } while (0)
#define Y(NB_, KB_)                                                                                                   \
    if (nb == NB_ && kb == KB_) {                                                                                     \
        if (gmode == 0) PN2_TGB2_LAUNCH(NB_, KB_, 0);                                                                 \
        else if (gmode == 1) PN2_TGB2_LAUNCH(NB_, KB_, 1);                                                            \
        else PN2_TGB2_LAUNCH(NB_, KB_, 2);                                                                            \
    }
    if (s0.v2) {
        if (nb == 4 && kb == 4) { Y(4, 4) }
        else PN2_TGB2_LAUNCH(4, 6, 2);
        return check_launch();
    }
#undef Y
#undef PN2_TGB2_LAUNCH
    X(4, 4) X(4, 6)
#undef X
#undef PN2_TGB_LAUNCH
    return check_launch();
}
