// Does v_mfma_f32_16x16x32_f16 on gfx950 keep fp16 SUBNORMAL inputs, and is its operand layout the bf16 instruction's?  (Round 5: the question behind an
// f16 hi + lo split -- 22 mantissa bits in two 16-bit planes; the lo parts of small weights are fp16 subnormals.)
//   test 1: layout / exact products with ordinary values (as tools/ubench/mfma_bf16_layout.cpp);
//   test 2: A = 2^-20 (an fp16 subnormal: the smallest normal is 2^-14) at k = 0, B = 2^10 -> D must be 2^-10, 0 would mean the input was flushed;
//   test 3: the same with the subnormal on the B side;
//   test 4: 32 products of 2^-24 (the smallest subnormal) x 1 summed -> 2^-19.
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma_f16_denorm.cpp -o build_ab/mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* A, const unsigned short* B, float* D) {       // A [16][32], B [32][16] (fp16 bits), D [16][16]
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    s16x8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (short)A[i * 32 + 8 * g + q]; b[q] = (short)B[(8 * g + q) * 16 + i]; }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + i] = c[r];
}
static unsigned short h16(float f) {          // float -> fp16 bits, round to nearest even, subnormals kept
    _Float16 h = (_Float16)f; unsigned short u; memcpy(&u, &h, 2); return u;
}
static float f16(unsigned short u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }
int main() {
    unsigned short hA[512], hB[512]; float hD[256];
    unsigned short *dA, *dB; float* dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 1024);
    auto run = [&]() { hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
                       hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD); hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost); };
    int bad = 0;
    for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 32; ++kk) hA[i * 32 + kk] = h16((float)((i * 7 + kk * 3) % 11 - 5) * 0.25f);
    for (int kk = 0; kk < 32; ++kk) for (int j = 0; j < 16; ++j) hB[kk * 16 + j] = h16((float)((kk * 5 + j * j) % 13 - 6) * 0.5f);
    run();
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        float s = 0; for (int kk = 0; kk < 32; ++kk) s += f16(hA[i * 32 + kk]) * f16(hB[kk * 16 + j]);
        if (s != hD[i * 16 + j] && bad++ < 5) printf("test 1 mismatch D[%d][%d] = %g, expected %g\n", i, j, hD[i * 16 + j], s);
    }
    printf("test 1 (layout, exact products): %s\n", bad ? "WRONG" : "OK");
    memset(hA, 0, sizeof hA); memset(hB, 0, sizeof hB);
    for (int i = 0; i < 16; ++i) hA[i * 32] = h16(ldexpf(1.f, -20));
    for (int j = 0; j < 16; ++j) hB[j] = h16(1024.f);
    run();
    printf("test 2 (subnormal A = 2^-20, bits 0x%04x, x 2^10): D[0][0] = %g (2^-10 = %g): %s\n", hA[0], hD[0], ldexpf(1.f, -10), hD[0] == ldexpf(1.f, -10) ? "KEPT" : "FLUSHED");
    memset(hA, 0, sizeof hA); memset(hB, 0, sizeof hB);
    for (int i = 0; i < 16; ++i) hA[i * 32] = h16(1024.f);
    for (int j = 0; j < 16; ++j) hB[j] = h16(ldexpf(1.f, -20));
    run();
    printf("test 3 (subnormal B): D[0][0] = %g: %s\n", hD[0], hD[0] == ldexpf(1.f, -10) ? "KEPT" : "FLUSHED");
    for (int i = 0; i < 512; ++i) { hA[i] = 0x0001; hB[i] = h16(1.f); }
    run();
    printf("test 4 (32 x 2^-24 x 1): D[0][0] = %g (2^-19 = %g): %s\n", hD[0], ldexpf(1.f, -19), hD[0] == ldexpf(1.f, -19) ? "KEPT" : "FLUSHED / inexact");
    return 0;
}
