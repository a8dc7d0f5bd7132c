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
    float4 pa[4], pb[4];
    fetch(pa, X, ldx, M, m0, 0, K < KC ? K : KC, vec_x);
    fetch(pb, W, ldw, N, n0, 0, K < KC ? K : KC, vec_w);
    for (int k0 = 0; k0 < K; k0 += KC) {
        if (k0) __syncthreads();  // the previous chunk's readers are done
        stage(pa, As);
        stage(pb, Bs);
        __syncthreads();
        const int kn = k0 + KC;
        if (kn < K) {  // the next chunk travels behind this chunk's matrix instructions
            fetch(pa, X, ldx, M, m0, kn, (K - kn) < KC ? (K - kn) : KC, vec_x);
            fetch(pb, W, ldw, N, n0, kn, (K - kn) < KC ? (K - kn) : KC, vec_w);
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
            if (bias) v += bias[n0 + c];
            if (relu) v = !(v <= 0.f) ? v : 0.f;  // propagates NaN like torch
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
    hipLaunchKernelGGL(ls::linear_small_kernel, grid, dim3(ls::kT), 0, (hipStream_t)stream, m, k, n, x, ldx, w, ldw, bias, relu, y, ldy,
                       vec_x, vec_w);
    return check_launch();
}
