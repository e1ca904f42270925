// LDS-DMA (global_load_lds_dwordx4) rate of a CU when every wave streams the SAME 768 KB of weight fragments over and over (sa_w_kernel's
// panels: twelve 64 KB panels per round, read by all 256 CUs at once -- L2 hits), by waves per CU.  One 1 KB wave instruction per fragment,
// `DEPTH` of them issued between waits (sa_w issues 16 per wave and panel, fp_s 8).
//   hipcc --offload-arch=gfx950 -O3 -o bin/dma_bw dma_bw.hip && bin/dma_bw
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int DEPTH>
__global__ void k(const char* __restrict__ w, int frags, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + wave * DEPTH * 1024;
    int f = wave;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            lds_dma16(w + (size_t)f * 1024 + lane * 16, __builtin_amdgcn_readfirstlane(base + d * 1024));
            f += nw; if (f >= frags) f -= frags;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (threadIdx.x == 0) sink[blockIdx.x] = lds[0];
}
template <int DEPTH>
static double run(const char* w, int wpc, float* sink) {
    const int iters = 8192 / DEPTH / wpc * 4;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    double best = 0;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<DEPTH>), dim3(256), dim3(64 * wpc), wpc * DEPTH * 1024, 0, w, 768, iters, sink);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double tbs = 256.0 * wpc * iters * DEPTH * 1024.0 / (ms * 1e-3) / 1e12;
        if (rep && tbs > best) best = tbs;
    }
    return best;
}
int main() {
    char* w; float* sink; (void)hipMalloc(&w, 768 * 1024); (void)hipMemset(w, 1, 768 * 1024); (void)hipMalloc(&sink, 1024);
    printf("LDS-DMA of a 768 KB fragment table shared by all CUs: TB/s (GB/s per CU) by waves per CU x copies issued between waits\n");
    printf("   waves/CU      4 deep              8                  16\n");
    for (int wpc : {4, 8, 12}) {
        const double a = run<4>(w, wpc, sink), b = run<8>(w, wpc, sink), c = wpc * 16 <= 128 ? run<16>(w, wpc, sink) : 0.0;
        printf("   %2d      %6.2f (%5.1f)    %6.2f (%5.1f)    %6.2f (%5.1f)\n", wpc, a, a * 1e3 / 256, b, b * 1e3 / 256, c, c * 1e3 / 256);
    }
    return 0;
}
