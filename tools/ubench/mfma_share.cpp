// Micro-benchmark: does the f32 MFMA pipe (v_mfma_f32_16x16x4_f32) keep its rate when 1, 2 or 4 waves share a SIMD,
// each wave issuing groups of 8 MFMAs on 4 accumulators separated by a few VALU ops (like the scalar GEMM loop)?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_share.cpp -o gpurun_out/mfma_share && gpurun_out/mfma_share
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int GAP>
__global__ void __launch_bounds__(1024) k(float* out, int iters, float a0, float b0) {
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < GAP; ++g) { a = a * 1.0000001f + 1e-9f; asm volatile("" : "+v"(a)); }
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int GAP>
void run(int threads, int blocks_per_cu) {
    float* out;
    hipMalloc(&out, (size_t)256 * blocks_per_cu * threads * 4);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<GAP>, dim3(256 * blocks_per_cu), dim3(threads), 0, 0, out, 100, 1.0f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<GAP>, dim3(256 * blocks_per_cu), dim3(threads), 0, 0, out, iters, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = 256.0 * blocks_per_cu * threads / 64;
    const double flops = waves * iters * 8.0 * 2048.0;
    printf("gap %d  threads/block %4d blocks/CU %d  waves/SIMD %.1f  %.1f TFLOP/s  (%.2f ms)\n", GAP, threads, blocks_per_cu,
           threads / 64.0 * blocks_per_cu / 4, flops / ms / 1e9, ms);
    hipFree(out);
}

int main() {
    run<0>(256, 1); run<0>(512, 1); run<0>(1024, 1); run<0>(512, 2);
    run<4>(256, 1); run<4>(512, 1); run<4>(1024, 1); run<4>(512, 2);
    run<16>(256, 1); run<16>(512, 1); run<16>(1024, 1); run<16>(512, 2);
    return 0;
}
