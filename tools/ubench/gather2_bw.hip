// Random-row gather ceiling of a CU with CHEAP addresses (round 5).  gather_bw.hip (round 4) formed every row number with a 32-bit modulo
// (~30 VALU instructions per load): at four waves per CU it measured its own address arithmetic (8, 16 and 24 loads in flight gave the same
// 11 TB/s), not the memory path.  Here the row numbers come from a precomputed index list -- one coalesced load of 64 indices per batch,
// requested a batch ahead, v_readlane per row: what the interpolating kernels do -- and two ways of landing a 1 KB row are compared:
//   VGPR   global_load_dwordx4 (64 lanes x 16 bytes) into DEPTH x 4 registers, then an add per register (stand-in for the interpolation);
//   DMA    global_load_lds_dwordx4 into a DEPTH KB slot ring of the wave in LDS (no destination registers), then one ds_read_b128 per row.
// Rows of a batch come from a window of 256 rows (one cloud of fp2's Y; the window moves every batch), table 2 MB / 37.7 MB / 604 MB.
//   hipcc --offload-arch=gfx950 -O3 -o bin/gather2_bw gather2_bw.hip && bin/gather2_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// idx: [wave][iter][64] row numbers (only the first DEPTH of each 64 are used)
template <int DEPTH, bool DMA, int ROWB>
__global__ void k(const char* __restrict__ rows, const int* __restrict__ idx, int iters, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const unsigned lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int* ip = idx + (size_t)wave * iters * 64 + lane;
    constexpr unsigned LPR = ROWB / 16, RPI = 64 / LPR;       // lanes per row, rows per wave instruction
    const unsigned sub = lane / LPR, lo = (lane % LPR) * 16;
    const unsigned slot = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + lw * DEPTH * 1024;
    f32x4 acc = {0, 0, 0, 0};
    int iv = ip[0];
    for (int it = 0; it < iters; ++it) {
        const int cur = iv;
        if (it + 1 < iters) iv = ip[(size_t)(it + 1) * 64];
        if constexpr (DMA) {
#pragma unroll
            for (int q = 0; q < DEPTH; ++q) {
                const char* src;
                if constexpr (RPI == 1) src = rows + (size_t)(unsigned)__builtin_amdgcn_readlane(cur, q) * ROWB + lo;
                else src = rows + (size_t)(unsigned)__shfl(cur, q * RPI + sub) * ROWB + lo;
                lds_dma16(src, __builtin_amdgcn_readfirstlane(slot + q * 1024));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < DEPTH; ++q) acc += *reinterpret_cast<const f32x4*>(lds + lw * DEPTH * 1024 + q * 1024 + lane * 16);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            f32x4 v[DEPTH];
#pragma unroll
            for (int q = 0; q < DEPTH; ++q) {
                const char* src;
                if constexpr (RPI == 1) src = rows + (size_t)(unsigned)__builtin_amdgcn_readlane(cur, q) * ROWB + lo;
                else src = rows + (size_t)(unsigned)__shfl(cur, q * RPI + sub) * ROWB + lo;
                v[q] = *reinterpret_cast<const f32x4*>(src);
            }
#pragma unroll
            for (int q = 0; q < DEPTH; ++q) acc += v[q];
        }
    }
    sink[(size_t)wave * 64 + lane] = acc[0] + acc[1] + acc[2] + acc[3];
}

static int* d_idx;
static std::vector<int> h_idx;
static void make_idx(unsigned nrows_of_rowb, int waves, int iters, unsigned window) {
    h_idx.resize((size_t)waves * iters * 64);
    unsigned s = 12345u;
    for (int w = 0; w < waves; ++w)
        for (int it = 0; it < iters; ++it) {
            s = s * 1664525u + 1013904223u;
            const unsigned base = window < nrows_of_rowb ? (s >> 8) % (nrows_of_rowb - window) : 0u;
            for (int l = 0; l < 64; ++l) {
                s = s * 1664525u + 1013904223u;
                h_idx[((size_t)w * iters + it) * 64 + l] = (int)(base + (s >> 8) % window);
            }
        }
    (void)hipMemcpy(d_idx, h_idx.data(), h_idx.size() * 4, hipMemcpyHostToDevice);
}

template <int DEPTH, bool DMA, int ROWB>
static double run(const char* rows, int wpc, int iters, float* sink) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    double best = 0;
    if (DMA) (void)hipFuncSetAttribute((const void*)k<DEPTH, DMA, ROWB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<DEPTH, DMA, ROWB>), dim3(256), dim3(64 * wpc), DMA ? wpc * DEPTH * 1024 : 0, 0, rows, d_idx, iters, sink);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double tbs = 256.0 * wpc * iters * DEPTH * 1024.0 / (ms * 1e-3) / 1e12;
        if (rep && tbs > best) best = tbs;
    }
    return best;
}

template <bool DMA, int ROWB>
static void table(const char* rows, float* sink, int iters) {
    printf("   %s, rows of %d B\n   waves/CU      8 in flight          16                 24\n", DMA ? "LDS-DMA (global_load_lds_dwordx4)" : "VGPR (global_load_dwordx4)", ROWB);
    for (int wpc : {4, 8, 12}) {
        const double a = run<8, DMA, ROWB>(rows, wpc, iters, sink), b = (!DMA || wpc * 16 <= 152) ? run<16, DMA, ROWB>(rows, wpc, iters, sink) : 0.0,
                     c = (!DMA || wpc * 24 <= 152) ? run<24, DMA, ROWB>(rows, wpc, iters, sink) : 0.0;
        printf("   %2d      %6.2f (%5.1f)    %6.2f (%5.1f)    %6.2f (%5.1f)\n", wpc, a, a * 1e3 / 256, b, b * 1e3 / 256, c, c * 1e3 / 256);
    }
}

int main() {
    const int iters = 160, maxw = 256 * 12;
    float* sink; (void)hipMalloc(&sink, (size_t)maxw * 64 * 4);
    (void)hipMalloc(&d_idx, (size_t)maxw * iters * 64 * 4);
    const unsigned sizes[3] = {2048u, 36864u, 589824u};       // KB: 2 MB (inside every XCD's 4 MB L2), 37.7 MB (fp2's Y at B = 64), 604 MB
    for (int si = 0; si < 3; ++si) {
        char* rows; (void)hipMalloc(&rows, (size_t)sizes[si] * 1024); (void)hipMemset(rows, 0, (size_t)sizes[si] * 1024);
        printf("table %.1f MB, rows of a batch from a window of 256 KB: TB/s of gathered rows (GB/s per CU) by waves per CU x KB in flight per wave\n", sizes[si] * 1024.0 / 1e6);
        make_idx(sizes[si], maxw, iters, 256);
        table<false, 1024>(rows, sink, iters);
        table<true, 1024>(rows, sink, iters);
        if (si == 1) {
            make_idx(sizes[si] * 2, maxw, iters, 512);
            table<false, 512>(rows, sink, iters);
            table<true, 512>(rows, sink, iters);
        }
        (void)hipFree(rows);
    }
    return 0;
}
