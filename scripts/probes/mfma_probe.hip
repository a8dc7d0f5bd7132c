// Micro-probe: what fp32 MFMA (16x16x4) rate does one wave per SIMD reach with (a) registers only,
// (b) A operand streamed from LDS like sa_mlp_max_kernel, (c) 8 waves/CU.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NACC>
__global__ void __launch_bounds__(256) probe(float *out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 132];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 64 * 132; i += 256) lds[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    float b[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) b[i] = 0.001f * (i + lane);
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    const float *arow = lds + (lane & 15) * 132 + 4 * (lane >> 4);
    float a0 = 1.0f + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int tq = 0; tq < 8; ++tq) {
            float4 a[NACC / 2];
            if (MODE == 1) {
#pragma unroll
                for (int r = 0; r < NACC / 2; ++r) a[r] = *reinterpret_cast<const float4 *>(arow + r * 16 * 132 + 16 * tq);
            } else {
#pragma unroll
                for (int r = 0; r < NACC / 2; ++r) a[r] = make_float4(a0, a0 + 1, a0 + 2, a0 + 3);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < NACC / 2; ++r) {
                    const float av = j == 0 ? a[r].x : j == 1 ? a[r].y : j == 2 ? a[r].z : a[r].w;
                    acc[2 * r] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[4 * tq + j], acc[2 * r], 0, 0, 0);
                    acc[2 * r + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[(4 * tq + j + 7) & 31], acc[2 * r + 1], 0, 0, 0);
                }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int NACC>
void run(const char *name, int grid, int iters) {
    float *d;
    hipMalloc(&d, grid * 256 * 4);
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    probe<MODE, NACC><<<grid, 256>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(s);
    probe<MODE, NACC><<<grid, 256>>>(d, iters);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms;
    hipEventElapsedTime(&ms, s, e);
    double mf = (double)grid * 4 * iters * 8 * 4 * NACC;  // MFMAs
    printf("%-28s grid %4d: %.1f us, %.1f TFLOP/s, %.1f cycles@2.1GHz per MFMA per wave\n", name, grid, ms * 1e3,
           mf * 2048 / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.1e9 / (iters * 8.0 * 4 * NACC));
    hipFree(d);
}

int main() {
    run<0, 4>("regs only, 4 acc", 256, 200);
    run<0, 8>("regs only, 8 acc", 256, 100);
    run<1, 4>("A from LDS b128, 4 acc", 256, 200);
    run<1, 8>("A from LDS b128, 8 acc", 256, 100);
    run<0, 4>("regs only, 4 acc, 2 WG/CU", 512, 200);
    run<1, 4>("A from LDS, 4 acc, 2 WG/CU", 512, 200);
    run<1, 8>("A from LDS, 8 acc, 2 WG/CU", 512, 100);
    return 0;
}
