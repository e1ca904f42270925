// What a CU can draw when every wave gathers random 1 KB rows (one row per wave instruction: 64 lanes x 16 bytes), the access pattern of the
// interpolating stages (fp2 / fp3 / fp1: three rows per point out of a per-cloud table).  Sweeps the waves per CU and the loads each wave keeps in
// flight; the table is either 37.7 MB (fp2's Y at B = 64: L2 + Infinity-Cache resident after the first pass) or 604 MB (HBM).
//   hipcc --offload-arch=gfx950 -O3 -o bin/gather_bw gather_bw.hip && bin/gather_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// LOCAL: rows of a wave's batch come from a window of `window` consecutive rows (the rows of one cloud: 256 for fp2) whose base changes per batch
// ROWF floats per row (256 = 1 KB: one row per wave instruction; 128 / 64: two / four rows per instruction, lanes split between them)
template <int INFLIGHT, int ROWF = 256>
__global__ void k(const float* __restrict__ rows, unsigned nrows, unsigned window, int iters, float* __restrict__ sink) {
    const unsigned lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    unsigned s = wave * 2654435761u + 12345u;
    f32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        const unsigned base = (window > 1u && window < nrows) ? (s >> 8) % (nrows - window) : 0u;
        f32x4 v[INFLIGHT];
#pragma unroll
        for (int q = 0; q < INFLIGHT; ++q) {
            s = s * 1664525u + 1013904223u;
            if constexpr (ROWF == 256) {
                const unsigned r = window == 1u ? (unsigned)((wave * 977u + (unsigned)it * INFLIGHT + q) % nrows)       // (window = 1: consecutive rows -- streaming)
                                                : __builtin_amdgcn_readfirstlane(base + (s >> 8) % window);
                v[q] = *reinterpret_cast<const f32x4*>(rows + (size_t)r * 256 + 4 * lane);
            } else {
                constexpr unsigned LPR = ROWF / 4;                    // lanes per row
                const unsigned sub = lane / LPR;
                const unsigned r = base + ((s >> 8) * (2u * sub + 1u) + sub * 7919u) % window;
                v[q] = *reinterpret_cast<const f32x4*>(rows + (size_t)r * ROWF + 4 * (lane % LPR));
            }
        }
#pragma unroll
        for (int q = 0; q < INFLIGHT; ++q) acc += v[q];
    }
    sink[(size_t)wave * 64 + lane] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int INFLIGHT, int ROWF = 256>
static double run(const float* rows, unsigned nrows, unsigned window, int wpc, float* sink) {
    const int iters = 4096 / INFLIGHT;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    double best = 0;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<INFLIGHT, ROWF>), dim3(256), dim3(64 * wpc), 0, 0, rows, nrows, window, iters, sink);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double tbs = 256.0 * wpc * iters * INFLIGHT * 1024.0 / (ms * 1e-3) / 1e12;
        if (rep && tbs > best) best = tbs;
    }
    return best;
}

int main() {
    float* sink; (void)hipMalloc(&sink, 256 * 16 * 64 * 4);
    const unsigned sizes[3] = {2048u, 36864u, 589824u};       // rows of 1 KB: 2 MB (inside every XCD's 4 MB L2), 37.7 MB (fp2's Y at B = 64), 604 MB
    for (int si = 0; si < 3; ++si) {
        float* rows; (void)hipMalloc(&rows, (size_t)sizes[si] * 1024); (void)hipMemset(rows, 0, (size_t)sizes[si] * 1024);
        for (unsigned window : {256u, sizes[si]}) {
            printf("table %.1f MB, rows of a batch from %s: TB/s of gathered rows (GB/s per CU) by waves per CU x row loads in flight per wave\n", sizes[si] * 1024.0 / 1e6,
                   window == 256u ? "a window of 256 rows (one cloud)" : "the whole table");
            printf("   waves/CU      8 in flight          16                 24                 32\n");
            for (int wpc : {4, 8, 16}) {
                const double a = run<8>(rows, sizes[si], window, wpc, sink), b = run<16>(rows, sizes[si], window, wpc, sink), c = run<24>(rows, sizes[si], window, wpc, sink),
                             d = run<32>(rows, sizes[si], window, wpc, sink);
                printf("   %2d      %6.2f (%5.1f)    %6.2f (%5.1f)    %6.2f (%5.1f)    %6.2f (%5.1f)\n", wpc, a, a * 1e3 / 256, b, b * 1e3 / 256, c, c * 1e3 / 256, d, d * 1e3 / 256);
            }
        }
        printf("table %.1f MB, CONSECUTIVE rows (streaming), 24 in flight: TB/s by waves per CU:", sizes[si] * 1024.0 / 1e6);
        for (int wpc : {4, 8, 16}) printf("  %d: %.2f", wpc, run<24>(rows, sizes[si], 1u, wpc, sink));
        printf("\n");
        if (si == 1) {
            printf("rows of 512 B (fp1, sa3) / 256 B (sa2), 37.7 MB table, window 256 rows, 24 in flight: TB/s by waves per CU\n");
            for (int wpc : {4, 8, 16}) printf("   %2d      512 B %6.2f     256 B %6.2f\n", wpc, run<24, 128>(rows, sizes[si] * 2, 256, wpc, sink), run<24, 64>(rows, sizes[si] * 4, 256, wpc, sink));
        }
        (void)hipFree(rows);
    }
    return 0;
}
