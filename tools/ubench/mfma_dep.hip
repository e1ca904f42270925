// Microbenchmark (gfx950): issue cost of DEPENDENT v_mfma_f32_32x32x16_bf16 chains (same accumulator) vs independent ones,
// for 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0)

template <int NACC, int PAT>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 0.001f + j); b[j] = (__bf16)(j * 0.5f); }
    f32x16 acc[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            // PAT 0: round-robin over NACC accumulators; PAT 1: three in a row on the same accumulator, then the next
            const int i = PAT == 0 ? u % NACC : (u / 3) % NACC;
            MFMA(acc[i], a, b);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, int PAT>
float run(float* d, int threads, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, PAT>), dim3(256), dim3(threads), 0, 0, d, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, PAT>), dim3(256), dim3(threads), 0, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3f * 2.4e9f / iters / 12.0f;   // cycles (2.4 GHz) per MFMA per wave-slot
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 1024 * 4);
    const int it = 20000;
    for (int waves = 1; waves <= 4; waves *= 2) {
        const int th = 256 * waves;
        printf("%d wave(s)/SIMD: cycles per MFMA issued by ONE wave (pipe floor = 32/waves... x waves)\n", waves);
        printf("  1 acc, fully dependent chain      : %6.1f\n", run<1, 0>(d, th, it));
        printf("  2 accs round-robin                : %6.1f\n", run<2, 0>(d, th, it));
        printf("  4 accs round-robin                : %6.1f\n", run<4, 0>(d, th, it));
        printf("  2 accs, 3 in a row on each (x3)   : %6.1f\n", run<2, 1>(d, th, it));
        printf("  4 accs, 3 in a row on each        : %6.1f\n", run<4, 1>(d, th, it));
    }
    return 0;
}
