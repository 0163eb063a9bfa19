// Operand layout of v_mfma_f32_4x4x1_16B_f32 on gfx950, checked against the formula the 4-row GEMM path assumes:
//   block b = lane >> 2;  A operand of lane (b, i = lane & 3) = A_b[i];  B operand of lane (b, j = lane & 3) = B_b[j];
//   D register r of lane (b, j) = C + A_b[r] * B_b[j]          (rows in registers, columns across the block's four lanes)
// With B_b[j] = W[k][4b + j] and A_b[i] = X[i][k] for every block, one instruction adds X[0..3][k] (x) W[k][0..63] to a 4 x 64 output tile.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_4x4_layout.cpp -o build_ab/mfma_4x4_layout && build_ab/mfma_4x4_layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}
int main() {
    float ha[64], hb[64], hd[256], *a, *b, *d;
    for (int l = 0; l < 64; ++l) { ha[l] = 1.0f + l; hb[l] = 100.0f + 3 * l; }
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, d);
    hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int blk = l >> 2;
            const float want = ha[4 * blk + r] * hb[l];
            if (hd[l * 4 + r] != want) { if (bad < 8) printf("lane %d reg %d: got %g, assumed %g\n", l, r, hd[l * 4 + r], want); ++bad; }
        }
    printf("v_mfma_f32_4x4x1_16B_f32 layout: %s (%d mismatches of 256)\n", bad ? "DIFFERS from the assumed formula" : "matches the assumed formula", bad);
    if (bad) for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g   (a=%g b=%g)\n", l, hd[4 * l], hd[4 * l + 1], hd[4 * l + 2], hd[4 * l + 3], ha[l], hb[l]);
    return 0;
}
