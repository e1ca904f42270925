// Effective matrix-pipe rate of the whole chip: every SIMD of every CU issues dependent v_mfma_f32_32x32x16_bf16 back to back
// (argv[1] waves per SIMD = 1, 2 or 4; N per wave).  Prints the wall time and the clock a 32-cycle (8-pass) MFMA implies.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CHAINS>
__global__ void k(float* out, int n) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)1.0f; }
    for (int i = 0; i < n; ++i)
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 2;
    const int n = 40000 / wps;
    float* out; (void)hipMalloc(&out, 256 * 1024 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<1>, dim3(256), dim3(256 * wps), 0, 0, out, n);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double mf = (double)n * wps;       // MFMAs per SIMD
        if (rep) printf("waves/SIMD %d: %.1f us, %.0f MFMA/SIMD -> %.3f GHz at 32 cycles each; %.0f TFLOP/s\n", wps, ms * 1e3, mf, mf * 32 / (ms * 1e-3) / 1e9,
                        mf * 1024 * 32768.0 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
