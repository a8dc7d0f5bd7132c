// How fast does one CU retire ds_add_f32?  One 1024-thread workgroup per CU, 32768 floats of LDS, every lane issues `iters`
// adds to addresses of a given pattern; reports lane-adds per clock per CU.  Patterns: 0 = unique random bank-spread,
// 1 = all lanes of a wave the same address, 2 = consecutive addresses, 3 = random with pairs of lanes colliding.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_atomic_rate lds_atomic_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(1024) k(int pattern, int iters, float *out, long long *cyc) {
    __shared__ float acc[32768];
    for (int i = threadIdx.x; i < 32768; i += 1024) acc[i] = 0.f;
    __syncthreads();
    unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    const int lane = threadIdx.x & 63;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        int a;
        if (pattern == 0) a = (s >> 9) & 32767;
        else if (pattern == 1) a = (((s >> 9) & 32767) & ~0) * 0 + ((it * 977 + (threadIdx.x >> 6) * 131) & 32767);
        else if (pattern == 2) a = (it * 1024 + threadIdx.x) & 32767;
        else a = ((s >> 9) & 32767) & ~1 | 0, a = (lane & 1) ? __shfl_xor(a, 1) : a;
        __hip_atomic_fetch_add(&acc[a], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    float v = 0.f;
    for (int i = threadIdx.x; i < 32768; i += 1024) v += acc[i];
    out[blockIdx.x * 1024 + threadIdx.x] = v;
}

int main() {
    float *out;
    long long *cyc;
    hipMalloc(&out, 256 * 1024 * 4);
    hipMalloc(&cyc, 256 * 8);
    for (int pattern = 0; pattern < 4; ++pattern) {
        const int iters = 4096;
        hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, pattern, iters, out, cyc);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, pattern, iters, out, cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(256);
        hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
        double c = 0;
        for (auto x : h) c += x;
        c /= 256;
        printf("pattern %d: %.3f ms, %.0f clock64 ticks per WG (100 MHz ticks?) ; lane-adds per CU per us: %.1f\n", pattern, ms, c,
               1024.0 * iters / (ms * 1e3));
    }
    return 0;
}
