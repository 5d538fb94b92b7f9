// fp32 MFMA issue-rate microbenchmark: dependent chains per wave x waves per SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CH>
__global__ void k(float* out, long long* cyc, int iters) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = threadIdx.x * 0.001f, b = 1.0f + blockIdx.x * 0.0001f;
    long long t0 = clock64();
    long long w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    long long t1 = clock64();
    long long w1 = wall_clock64();
    float s = 0.f;
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}
template <int CH>
void run(int threads, int blocks, int iters) {
    float* out; long long* cyc;
    hipMalloc(&out, (size_t)threads * blocks * 4); hipMalloc(&cyc, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<CH>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    const double n = (double)iters * 8 * CH;            // MFMAs per wave
    const double waves_per_simd = threads / 256.0 * (blocks / 256.0);
    const double tf = n * (threads / 64.0) * blocks * 4096.0 / (ms * 1e-3) / 1e12;
    printf("chains %d  waves/SIMD %.0f  iters %d: %.1f us  clock64 %.1f cyc/MFMA/wave  wall %.1f ns/MFMA/wave  => %.1f TF  (clk %.2f GHz)\n",
           CH, waves_per_simd, iters, ms * 1e3, h[0] / n, h[1] * 10.0 / n, tf, (double)h[0] / (h[1] * 10.0));
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int iters : {20, 200, 2000}) {
        run<1>(256, 256, iters); run<2>(256, 256, iters); run<4>(256, 256, iters);
        run<1>(512, 256, iters); run<1>(1024, 256, iters); run<2>(1024, 256, iters);
    }
    return 0;
}
