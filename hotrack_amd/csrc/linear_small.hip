// linear_small.hip -- y = act(x W^T + bias) for FEW rows (the B = 1 / B = 8 tracking loop, gfx950).
//
// A single 1024-point frame sends 21 ... 1024 rows through ~20 dense layers (network/models/fast_eval.py).  The BLAS library's
// kernels are built for thousands of rows: at these sizes its best solutions (hotrack_amd/tunableop_gfx950.csv) put one or two
// workgroups on the problem and walk the reduction serially -- 5 ... 12 us per layer where the launch floor is ~4.5 us
// (profiles/r03_latency_b1_replay.csv: 155 us of a 428 us frame).  Here a workgroup owns ONE 32 x 32 output block and its four
// waves split the reduction (then meet in LDS): (rows / 32) x (columns / 32) workgroups, e.g. 16 for 128 x 128 x 128 and 256
// for 1024 x 320 -> 256, each with 16 ... 80 v_mfma_f32_32x32x2_f32 per wave.  Any K (zero-padded in LDS to the chunk), any
// alignment (16-byte loads where the row strides allow).  Results differ from the library's in summation order only.
#include "pn2_common.h"
#include "../../include/pn2_ext.h"

namespace pn2 {
namespace ls {

constexpr int kT = 256;
constexpr int KC = 128;      // reduction chunk staged in LDS (2 x 16.5 KiB + 16.9 KiB of partial blocks: static LDS)
constexpr int LD = KC + 1;   // odd row stride: the 32 rows an operand read touches fall on 32 banks
typedef float f32x16 __attribute__((ext_vector_type(16)));

// rows [r0, r0 + 32) x columns [k0, k0 + kc) of a row-major operand into registers: thread t owns the four consecutive columns
// 4 (t % 32) of rows t / 32 + 8 i (i < 4) -- 16-byte loads where the operand's rows are 16-byte aligned, guarded scalars for
// the last partial quad and for unaligned operands; rows / columns beyond the problem read as zero
__device__ __forceinline__ void fetch(float4 (&v)[4], const float *__restrict__ P, int ld, int rows, int r0, int k0, int kc, bool vec) {
    const int q = threadIdx.x & 31, rr = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + rr + 8 * i, k = 4 * q;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < rows && k < kc) {
            const float *p = P + (size_t)r * ld + k0 + k;
            if (vec && k + 3 < kc) {
                t = *reinterpret_cast<const float4 *>(p);
            } else {
                t.x = p[0];
                if (k + 1 < kc) t.y = p[1];
                if (k + 2 < kc) t.z = p[2];
                if (k + 3 < kc) t.w = p[3];
            }
        }
        v[i] = t;
    }
}

__device__ __forceinline__ void stage(const float4 (&v)[4], float *S) {
    const int q = threadIdx.x & 31, rr = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float *d = S + (rr + 8 * i) * LD + 4 * q;
        d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
    }
}

// PRE3 (K <= 3 KC = 384: every layer the dispatcher sends here): all chunks of both operands and the bias are requested before the
// first barrier -- at these sizes the kernel is a chain of memory round trips (operands, next chunk, bias), ~1 us each on an idle
// chip, and the tracking frame has thirteen of these launches in a row.  Same instructions on the same values: same bits.
template <bool PRE3>
__global__ void __launch_bounds__(kT)
linear_small_kernel(int M, int K, int N, const float *__restrict__ X, int ldx, const float *__restrict__ W, int ldw,
                    const float *__restrict__ bias, int relu, float *__restrict__ Y, int ldy, int vec_x, int vec_w) {
    __shared__ float As[32 * LD];
    __shared__ float Bs[32 * LD];
    __shared__ float red[4][32 * 33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int NPRE = PRE3 ? 3 : 1;
    float4 pa[NPRE][4], pb[NPRE][4];
#pragma unroll
    for (int c = 0; c < NPRE; ++c) {
        const int kc0 = c * KC, kcn = kc0 < K ? ((K - kc0) < KC ? (K - kc0) : KC) : 0;  // (an absent chunk: no loads, zeros)
        fetch(pa[c], X, ldx, M, m0, kcn ? kc0 : 0, kcn, vec_x);
        fetch(pb[c], W, ldw, N, n0, kcn ? kc0 : 0, kcn, vec_w);
    }
    // (the epilogue's elements tid + 256 i all lie in column tid & 31)
    const float bias_r = (PRE3 && bias && n0 + (tid & 31) < N) ? bias[n0 + (tid & 31)] : 0.f;
    for (int k0 = 0; k0 < K; k0 += KC) {
        if (k0) __syncthreads();  // the previous chunk's readers are done
        if constexpr (PRE3) {
            if (k0 == 0) { stage(pa[0], As); stage(pb[0], Bs); }
            else if (k0 == KC) { stage(pa[1], As); stage(pb[1], Bs); }
            else { stage(pa[2], As); stage(pb[2], Bs); }
        } else {
            stage(pa[0], As);
            stage(pb[0], Bs);
        }
        __syncthreads();
        const int kn = k0 + KC;
        if (!PRE3 && kn < K) {  // the next chunk travels behind this chunk's matrix instructions
            fetch(pa[0], X, ldx, M, m0, kn, (K - kn) < KC ? (K - kn) : KC, vec_x);
            fetch(pb[0], W, ldw, N, n0, kn, (K - kn) < KC ? (K - kn) : KC, vec_w);
        }
        const int kc = (K - k0) < KC ? (K - k0) : KC;
        const int steps = ((kc + 7) & ~7) / 2;  // zero-padded to the four waves' two-wide steps (the staged tile is full width)
        const float *ap = As + l31 * LD + kh, *bp = Bs + l31 * LD + kh;
#pragma unroll 4
        for (int s = wave; s < steps; s += 4) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s], bp[2 * s], acc, 0, 0, 0);
    }
    // the four partial blocks meet in LDS: D row = (r & 3) + 8 (r >> 2) + 4 kh, column = l31
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][((r & 3) + 8 * (r >> 2) + 4 * kh) * 33 + l31] = acc[r];
    __syncthreads();
    for (int e = tid; e < 32 * 32; e += kT) {
        const int r = e >> 5, c = e & 31;
        if (m0 + r < M && n0 + c < N) {
            float v = (red[0][r * 33 + c] + red[1][r * 33 + c]) + (red[2][r * 33 + c] + red[3][r * 33 + c]);
            if (bias) v += PRE3 ? bias_r : bias[n0 + c];
            if (relu) v = !(v <= 0.f) ? v : 0.f;  // propagates NaN like torch
            Y[(size_t)(m0 + r) * ldy + n0 + c] = v;
        }
    }
}

// ---- the same block kernel with the A operand PRODUCED in the prologue: x = LN2(LN1(xa + ya + ybias)) -------------------------
// The 21-token tail at small batch is [LayerNorm(s) -> Linear + ReLU -> Linear -> residual + LayerNorm(s)] x 2 -> Linear + ReLU ->
// head, every launch at the ~5 us floor of a replayed graph node (profiles/r05_latency_b1_replay.csv).  Here the LayerNorm launch
// in front of a Linear disappears: every workgroup normalises its 32 rows itself (one wave per row, the arithmetic of tail.hip's
// add_layernorm_kernel instruction for instruction: same sums in the same order, so results are bit-equal to the two launches),
// keeps them in LDS as the A operand for the whole reduction, and the workgroups of the first column block also write them out
// (the next residual needs them).  C <= 384.
constexpr int LNK = 384, LDXN = LNK + 1, LN_EPL = LNK / 64;
static_assert(LNK <= 3 * KC, "the three prefetched chunks of W cover the reduction");
constexpr size_t ln_lds_bytes = (size_t)(32 * LDXN + 32 * LD + 4 * 32 * 33) * sizeof(float);

constexpr int kLnT = 1024;  // sixteen waves normalise two rows each (one wave per row is the LayerNorm's own parallelism: with four
                            // waves walking eight rows each the prologue was a 6 us dependent chain); waves 0 - 3 then run the product
__global__ void __launch_bounds__(kLnT)
ln_linear_small_kernel(int M, int K, int N, const float *__restrict__ xa, const float *__restrict__ ya, const float *__restrict__ ybias,
                       const float *__restrict__ g1, const float *__restrict__ b1, float eps1, const float *__restrict__ g2,
                       const float *__restrict__ b2, float eps2, float *__restrict__ xout, const float *__restrict__ W, int ldw,
                       const float *__restrict__ bias, int relu, float *__restrict__ Y, int ldy, int vec_w) {
    extern __shared__ __attribute__((aligned(16))) float lsm[];
    float *XN = lsm;                  // [32][LDXN]
    float *Bs = XN + 32 * LDXN;       // [32][LD]
    float *red = Bs + 32 * LD;        // [4][32 * 33]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const bool gemm = wave < 4;  // (wave-uniform; fetch / stage address by threadIdx.x < 256)
    // every global operand of the product is requested up front (all three 128-wide chunks of W, the bias): the kernel is a chain
    // of memory round trips at this size, and each one not taken is ~1 us
    float4 pb[3][4];
    float bias_r = 0.f;  // the epilogue handles element tid of the 32 x 32 block: column tid & 31
    if (gemm) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int kc0 = c * KC;
            fetch(pb[c], W, ldw, N, n0, kc0 < K ? kc0 : 0, kc0 < K ? ((K - kc0) < KC ? (K - kc0) : KC) : 0, vec_w);
        }
    }
    if (bias) bias_r = n0 + (tid & 31) < N ? bias[n0 + (tid & 31)] : 0.f;
    const int kpad = (K + 7) & ~7;
    // a wave owns rows wave and wave + 16.  Everything it needs from memory is requested FIRST, for both rows at once (clamped
    // addresses, no branches): one memory round trip in front of the arithmetic
    float gam1[LN_EPL], bet1[LN_EPL], gam2[LN_EPL], bet2[LN_EPL], yb[LN_EPL];
#pragma unroll
    for (int e = 0; e < LN_EPL; ++e) {
        const int ch = lane + 64 * e, cc = ch < K ? ch : K - 1;
        gam1[e] = g1[cc]; bet1[e] = b1[cc];
        gam2[e] = g2 ? g2[cc] : 0.f; bet2[e] = g2 ? b2[cc] : 0.f;
        yb[e] = ybias ? ybias[cc] : 0.f;
    }
    float v[2][LN_EPL];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        long row = m0 + wave + 16 * i;
        row = row < M ? row : M - 1;
#pragma unroll
        for (int e = 0; e < LN_EPL; ++e) {
            const int ch = lane + 64 * e, cc = ch < K ? ch : K - 1;
            float t = xa[row * K + cc];
            if (ya) t += ya[row * K + cc];      // (uniform)
            if (ybias) t += yb[e];
            v[i][e] = ch < K ? t : 0.f;
        }
    }
    const float inv_c = 1.0f / (float)K;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = wave + 16 * i;
        const long row = m0 + r;
        if (row < M) {  // (wave-uniform) the arithmetic of add_layernorm_kernel, in its order
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < LN_EPL; ++e) s += v[i][e];
            float mean = wave_sum_f32(s) * inv_c, q = 0.f;
#pragma unroll
            for (int e = 0; e < LN_EPL; ++e) {
                const float d = (lane + 64 * e < K) ? v[i][e] - mean : 0.f;
                q += d * d;
            }
            float rstd = rsqrtf(wave_sum_f32(q) * inv_c + eps1);
            s = 0.f;
#pragma unroll
            for (int e = 0; e < LN_EPL; ++e) {
                v[i][e] = (lane + 64 * e < K) ? (v[i][e] - mean) * rstd * gam1[e] + bet1[e] : 0.f;
                s += v[i][e];
            }
            if (g2) {
                mean = wave_sum_f32(s) * inv_c;
                q = 0.f;
#pragma unroll
                for (int e = 0; e < LN_EPL; ++e) {
                    const float d = (lane + 64 * e < K) ? v[i][e] - mean : 0.f;
                    q += d * d;
                }
                rstd = rsqrtf(wave_sum_f32(q) * inv_c + eps2);
#pragma unroll
                for (int e = 0; e < LN_EPL; ++e)
                    if (lane + 64 * e < K) v[i][e] = (v[i][e] - mean) * rstd * gam2[e] + bet2[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < LN_EPL; ++e) v[i][e] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < LN_EPL; ++e) {
            const int ch = lane + 64 * e;
            if (ch < kpad) XN[r * LDXN + ch] = v[i][e];  // (columns K .. kpad - 1: zeros, the matrix steps are two wide over four waves)
            if (xout && blockIdx.y == 0 && row < M && ch < K) xout[row * K + ch] = v[i][e];
        }
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += KC) {
        if (k0) __syncthreads();  // the previous chunk's readers are done
        if (gemm) {
            if (k0 == 0) stage(pb[0], Bs);
            else if (k0 == KC) stage(pb[1], Bs);
            else stage(pb[2], Bs);
        }
        __syncthreads();          // (also: the normalised rows are complete)
        if (!gemm) continue;
        const int kc = (K - k0) < KC ? (K - k0) : KC;
        const int steps = ((kc + 7) & ~7) / 2;
        const float *ap = XN + l31 * LDXN + k0 + kh, *bp = Bs + l31 * LD + kh;
#pragma unroll 4
        for (int s = wave; s < steps; s += 4) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s], bp[2 * s], acc, 0, 0, 0);
    }
    if (gemm) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave * (32 * 33) + ((r & 3) + 8 * (r >> 2) + 4 * kh) * 33 + l31] = acc[r];
    }
    __syncthreads();
    for (int e = tid; e < 32 * 32; e += kLnT) {
        const int r = e >> 5, c = e & 31;
        if (m0 + r < M && n0 + c < N) {
            float v = (red[r * 33 + c] + red[32 * 33 + r * 33 + c]) + (red[2 * 32 * 33 + r * 33 + c] + red[3 * 32 * 33 + r * 33 + c]);
            if (bias) v += bias_r;  // (c == tid & 31: 1024 threads, 1024 elements)
            if (relu) v = !(v <= 0.f) ? v : 0.f;
            Y[(size_t)(m0 + r) * ldy + n0 + c] = v;
        }
    }
}

}  // namespace ls
}  // namespace pn2

// y (m x n, row stride ldy) = act(x (m x k, row stride ldx) . w^T (w: n x k, row stride ldw) + bias (n | NULL)); relu != 0: ReLU.
// Meant for m <= ~2048 (one workgroup per 32 x 32 output block); larger problems belong to the BLAS library.
extern "C" int pn2x_linear_small(int m, int k, int n, const float *x, int ldx, const float *w, int ldw, const float *bias, int relu,
                                 float *y, int ldy, void *stream) {
    using namespace pn2;
    if (m < 0 || n < 0 || k < 1 || ldx < k || ldw < k || ldy < n) return PN2_EINVAL;
    if (m == 0 || n == 0) return PN2_OK;
    if (!x || !w || !y) return PN2_ENULL;
    const int vec_x = (ldx % 4 == 0 && (uintptr_t)x % 16 == 0) ? 1 : 0, vec_w = (ldw % 4 == 0 && (uintptr_t)w % 16 == 0) ? 1 : 0;
    const dim3 grid((m + 31) / 32, (n + 31) / 32);
    if (k <= 3 * ls::KC)
        hipLaunchKernelGGL(ls::linear_small_kernel<true>, grid, dim3(ls::kT), 0, (hipStream_t)stream, m, k, n, x, ldx, w, ldw, bias, relu, y,
                           ldy, vec_x, vec_w);
    else
        hipLaunchKernelGGL(ls::linear_small_kernel<false>, grid, dim3(ls::kT), 0, (hipStream_t)stream, m, k, n, x, ldx, w, ldw, bias, relu, y,
                           ldy, vec_x, vec_w);
    return check_launch();
}

// y (m x n) = act(LN2(LN1(xa + ya + ybias)) . w^T + bias) with the normalised rows also written to xout (m x k, or NULL): the
// element-wise launch in front of a small Linear folded into it (k <= 384 channels, contiguous rows of k floats; ya / ybias / the
// second LayerNorm (g2, b2) / xout optional).  The LayerNorm arithmetic is pn2x_add_layernorm's: the pair of launches it replaces
// gives the same bits.
extern "C" int pn2x_ln_linear_small(int m, int k, int n, const float *xa, const float *ya, const float *ybias, const float *g1,
                                    const float *b1, float eps1, const float *g2, const float *b2, float eps2, float *xout,
                                    const float *w, int ldw, const float *bias, int relu, float *y, int ldy, void *stream) {
    using namespace pn2;
    if (m < 0 || n < 0 || k < 1 || k > ls::LNK || ldw < k || ldy < n) return PN2_EINVAL;
    if (m == 0 || n == 0) return PN2_OK;
    if (!xa || !g1 || !b1 || !w || !y || (g2 && !b2)) return PN2_ENULL;
    const int vec_w = (ldw % 4 == 0 && (uintptr_t)w % 16 == 0) ? 1 : 0;
    static PerDeviceOnce once;
    if (once.first_use())
        (void)hipFuncSetAttribute((const void *)ls::ln_linear_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ls::ln_lds_bytes);
    const dim3 grid((m + 31) / 32, (n + 31) / 32);
    hipLaunchKernelGGL(ls::ln_linear_small_kernel, grid, dim3(ls::kLnT), ls::ln_lds_bytes, (hipStream_t)stream, m, k, n, xa, ya, ybias, g1, b1,
                       eps1, g2, b2, eps2, xout, w, ldw, bias, relu, y, ldy, vec_w);
    return check_launch();
}
