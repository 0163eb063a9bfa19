// Micro-benchmark (round 2): what does ONE instruction of a given kind cost the f32 matrix pipe when it is issued between groups of
// v_mfma_f32_16x16x4_f32 (8 MFMAs on 4 accumulators, like the scalar-GEMM loop), 4 waves per SIMD?  Kinds: plain f32 VALU (v_fma),
// packed f32 VALU (v_pk_fma: two FMAs per lane), transcendental (v_exp, v_rcp), integer VALU (v_add_u32), SALU (s_add), LDS read.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_cost.cpp -o gpurun_out/valu_cost && gpurun_out/valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int GAP>
__global__ void __launch_bounds__(512) k(float* out, int iters, float a0, float b0) {
    __shared__ float lds[512];
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    f32x2 p = {a, a + 1.f};
    int u = threadIdx.x;
    int s = iters;
    lds[threadIdx.x] = a;
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < GAP; ++g) {
            if (KIND == 0) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b)); }
            if (KIND == 1) { asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p) : "v"(p)); }
            if (KIND == 2) { asm volatile("v_exp_f32 %0, %0" : "+v"(a)); }
            if (KIND == 3) { asm volatile("v_rcp_f32 %0, %0" : "+v"(a)); }
            if (KIND == 4) { asm volatile("v_add_u32 %0, %0, %0" : "+v"(u)); }
            if (KIND == 5) { asm volatile("s_add_u32 %0, %0, 1" : "+s"(s)); }
            if (KIND == 6) { float t; asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(u & 0x7fc)); a += t * 0.f; }
            if (KIND == 7) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(p)); }
            if (KIND == 8) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b)); }
        }
    }
    float r = a + p[0] + p[1] + (float)u + (float)s;
    for (int j = 0; j < 4; ++j) r += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND, int GAP>
double run() {
    float* out;
    hipMalloc(&out, (size_t)512 * 512 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, GAP>), dim3(512), dim3(512), 0, 0, out, 100, 1.0f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, GAP>), dim3(512), dim3(512), 0, 0, out, iters, 1.0f, 0.5f);     // 2 workgroups of 8 waves per CU = 4 waves / SIMD
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    return ms;
}

template <int KIND>
void kind(const char* name) {
    const double m0 = run<KIND, 0>(), m8 = run<KIND, 8>(), m16 = run<KIND, 16>();
    // per-iteration: 8 MFMAs = 256 matrix-pipe cycles per wave, 4 waves per SIMD -> 1024 cycles per SIMD per iteration at 100 %
    const double cyc_per_iter0 = 1024.0;      // by definition of the baseline
    const double c8 = (m8 / m0 - 1.0) * cyc_per_iter0 / (8 * 4), c16 = (m16 / m0 - 1.0) * cyc_per_iter0 / (16 * 4);
    printf("%-12s  gap0 %.2f ms  gap8 %.2f ms  gap16 %.2f ms   matrix-pipe cycles lost per instruction: %.2f (gap 8)  %.2f (gap 16)\n", name, m0, m8, m16, c8, c16);
}

int main() {
    kind<0>("v_fma_f32"); kind<8>("v_mul_f32"); kind<1>("v_pk_fma_f32"); kind<7>("v_pk_mul_f32"); kind<2>("v_exp_f32"); kind<3>("v_rcp_f32");
    kind<4>("v_add_u32"); kind<5>("s_add_u32"); kind<6>("ds_read_b32");
    return 0;
}
