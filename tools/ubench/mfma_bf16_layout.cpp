// Operand-layout check of v_mfma_f32_16x16x32_bf16 on gfx950 with an ASYMMETRIC B (cdna_hip_programming.md): lane l holds A[i = l & 15][k = 8 (l >> 4) .. +7],
// B[k = 8 (l >> 4) .. +7][j = l & 15]; D[row = 4 (l >> 4) + r][col = l & 15].  Prints "layout OK" or the first mismatches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const short* A, const short* B, float* D) {       // A [16][32] row-major, B [32][16] row-major (bf16 bits), D [16][16]
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    s16x8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = A[i * 32 + 8 * g + q]; b[q] = B[(8 * g + q) * 16 + i]; }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + i] = c[r];
}
static short bf(float f) { unsigned u; memcpy(&u, &f, 4); return (short)(u >> 16); }
int main() {
    short hA[512], hB[512]; float fA[512], fB[512], hD[256];
    for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 32; ++kk) { fA[i * 32 + kk] = (float)((i * 7 + kk * 3) % 11 - 5); hA[i * 32 + kk] = bf(fA[i * 32 + kk]); }
    for (int kk = 0; kk < 32; ++kk) for (int j = 0; j < 16; ++j) { fB[kk * 16 + j] = (float)((kk * 5 + j * j) % 13 - 6); hB[kk * 16 + j] = bf(fB[kk * 16 + j]); }
    short *dA, *dB; float* dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 1024);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        float s = 0; for (int kk = 0; kk < 32; ++kk) s += fA[i * 32 + kk] * fB[kk * 16 + j];
        if (s != hD[i * 16 + j] && bad++ < 5) printf("mismatch D[%d][%d] = %g, expected %g\n", i, j, hD[i * 16 + j], s);
    }
    printf(bad ? "layout WRONG (%d mismatches)\n" : "layout OK\n", bad);
    return bad != 0;
}
