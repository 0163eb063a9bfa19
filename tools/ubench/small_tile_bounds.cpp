// Micro-benchmark behind DESIGN.md section 3 "Small-batch latency": what a SINGLE CU can do for a node tile whose GVP chain is serial.
//   (1) L2 -> CU weight stream of one workgroup (8 waves, coalesced 512-B / 1-KB buffer loads, software-pipelined like the GEMM loops):
//       a scalar GEMM of a GVP streams 296 x 256 x 4 B = 303 KB of packed weights through the CU that owns the tile, whatever the tile's
//       row count -- the floor of a 4-row tile's GEMM time;
//   (2) the rate of v_mfma_f32_4x4x1_16B_f32 (the instruction a 4-row tile would use) against v_mfma_f32_16x16x4_f32 on one CU.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/small_tile_bounds.cpp -o build_ab/small_tile_bounds && build_ab/small_tile_bounds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VEC>   // 2: dwordx2 (512 B per wave instruction), 4: dwordx4 (1 KB)
__global__ void __launch_bounds__(512) stream_k(const float* __restrict__ w, size_t n_floats, int reps, float* out) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), (short)0, (int)(n_floats * 4), 0x00020000);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per_wave_instr = 64 * VEC * 4;                       // bytes one wave instruction moves
    const int n_instr = (int)(n_floats * 4 / per_wave_instr);      // instructions to cover the buffer once
    float acc = 0.f;
    for (int r = 0; r < reps; ++r) {
        // 8 loads in flight per wave, like the three-register-set GEMM loops (2 column tiles x prefetch distance 2 ... 4)
        for (int i = wave; i + 7 * 8 < n_instr; i += 8 * 8) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int off = (i + q * 8) * per_wave_instr + lane * VEC * 4;
                if (VEC == 2) { const auto t = __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0); v[q] = __builtin_bit_cast(float, (unsigned)t[0]) + __builtin_bit_cast(float, (unsigned)t[1]); }
                else { const auto t = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0); v[q] = __builtin_bit_cast(float, (unsigned)t[0]) + __builtin_bit_cast(float, (unsigned)t[3]); }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += v[q];
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}

template <int KIND>   // 0: 16x16x4 (2048 FLOP), 1: 4x4x1 16 blocks (512 FLOP)
__global__ void __launch_bounds__(512) mfma_k(float* out, int iters, float a0, float b0) {
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[j] = KIND == 0 ? __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0) : __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[j], 0, 0, 0);
    float s = 0;
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

static float timed(void (*launch)(int), int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(reps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

static float* g_w; static size_t g_n; static float* g_out; static int g_blocks;
template <int VEC> static void launch_stream(int reps) { hipLaunchKernelGGL(stream_k<VEC>, dim3(g_blocks), dim3(512), 0, 0, g_w, g_n, reps, g_out); }
template <int KIND> static void launch_mfma(int it) { hipLaunchKernelGGL(mfma_k<KIND>, dim3(g_blocks), dim3(512), 0, 0, g_out, it * 2000, 1.0f, 0.5f); }

int main() {
    hipMalloc(&g_out, 1024 * 512 * 4);
    const size_t sizes[] = {303104 / 4, 2 * 1024 * 1024 / 4 + 0, 2621440 / 4};      // one scalar GEMM; 2 MiB; all weights of a fused node tile (2.6 MB)
    for (size_t n : sizes) {
        n = n / (64 * 4 * 8 * 8) * (64 * 4 * 8 * 8);                                  // whole iterations of the unrolled loop
        g_n = n;
        std::vector<float> h(n, 1.0f);
        hipMalloc(&g_w, n * 4);
        hipMemcpy(g_w, h.data(), n * 4, hipMemcpyHostToDevice);
        for (int blocks : {1, 256}) {
            g_blocks = blocks;
            const int reps = 200;
            const float ms2 = timed(launch_stream<2>, reps), ms4 = timed(launch_stream<4>, reps);
            printf("L2->CU stream  buffer %7.1f KB  workgroups %3d (one per CU)  dwordx2: %6.1f GB/s per CU, %6.2f us per pass   dwordx4: %6.1f GB/s per CU, %6.2f us per pass\n",
                   n * 4 / 1024.0, blocks, n * 4.0 * reps / (ms2 * 1e-3) / 1e9, ms2 * 1e3 / reps, n * 4.0 * reps / (ms4 * 1e-3) / 1e9, ms4 * 1e3 / reps);
        }
        hipFree(g_w);
    }
    for (int blocks : {1, 256}) {
        g_blocks = blocks;
        const float m0 = timed(launch_mfma<0>, 10), m1 = timed(launch_mfma<1>, 10);
        const double n_mfma = 8.0 * 20000 * 16;      // per workgroup: 8 waves x iters x 16
        printf("MFMA rate, %3d workgroup(s) of 8 waves: 16x16x4_f32 %.1f GFLOP/s per CU (%.1f cycles per instruction per SIMD at 2.4 GHz)   4x4x1_16B_f32 %.1f GFLOP/s per CU (%.1f cycles)\n",
               blocks, n_mfma * 2048 / (m0 * 1e-3) / 1e9, m0 * 1e-3 * 2.4e9 / (n_mfma / 4), n_mfma * 512 / (m1 * 1e-3) / 1e9, m1 * 1e-3 * 2.4e9 / (n_mfma / 4));
    }
    return 0;
}
