// Hardware check of fm_group_sum (flowmol_amd/csrc/fm_device.h): the DPP / ds_swizzle group sums the LayerNorm statistics use must be BIT FOR BIT the xor
// butterfly `for (o = 1; o < LPR; o <<= 1) s += __shfl_xor(s, o)` they replaced, for every group width in use (8, 16, 32 lanes) and on values whose sums round
// (random magnitudes over 12 binades, mixed signs).  The header is the shipped one.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/dpp_group_sum.cpp -o build_ab/dpp_group_sum && build_ab/dpp_group_sum
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../flowmol_amd/csrc/fm_device.h"

template <int LPR>
__global__ void __launch_bounds__(512) k(const float* in, float* dpp, float* ref) {
    const int i = blockIdx.x * 512 + threadIdx.x;
    const float v = in[i];
    dpp[i] = fm_group_sum<LPR>(v);
    float s = v;
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) s += __shfl_xor(s, o);
    ref[i] = s;
}

template <int LPR>
static int run(const float* din, float* ddpp, float* dref, const std::vector<float>& hin, int n) {
    hipLaunchKernelGGL(k<LPR>, dim3(n / 512), dim3(512), 0, 0, din, ddpp, dref);
    std::vector<float> a(n), b(n);
    hipMemcpy(a.data(), ddpp, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), dref, n * 4, hipMemcpyDeviceToHost);
    int bad = 0, group_bad = 0;
    for (int i = 0; i < n; ++i) {
        if (memcmp(&a[i], &b[i], 4)) { if (bad < 4) printf("  LPR %d lane %d: dpp %.9g butterfly %.9g\n", LPR, i, a[i], b[i]); ++bad; }
        if (memcmp(&a[i], &a[i / LPR * LPR], 4)) ++group_bad;          // every lane of a group holds the same total
    }
    double exact = 0, got = 0;                                           // and it IS the group's sum (first group, in double)
    for (int j = 0; j < LPR; ++j) exact += hin[j];
    got = a[0];
    printf("fm_group_sum<%2d>: %d of %d lanes differ from the xor butterfly, %d lanes differ from their group's lane 0; first group: %.9g (exact %.9g)\n", LPR, bad, n,
           group_bad, got, exact);
    return bad + group_bad;
}

int main() {
    const int n = 512 * 4096;
    std::vector<float> h(n);
    srand(1);
    for (auto& v : h) v = ((rand() & 1) ? 1.f : -1.f) * (float)(rand() % 100000 + 1) * (1.0f / 8192.f) * (float)(1 << (rand() % 12));
    float *din, *ddpp, *dref;
    hipMalloc(&din, n * 4); hipMalloc(&ddpp, n * 4); hipMalloc(&dref, n * 4);
    hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice);
    const int bad = run<8>(din, ddpp, dref, h, n) + run<16>(din, ddpp, dref, h, n) + run<32>(din, ddpp, dref, h, n);
    printf("%s\n", bad ? "MISMATCH" : "fm_group_sum == xor butterfly bit for bit (2 M lanes per width)");
    return bad != 0;
}
