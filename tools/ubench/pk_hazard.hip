// Reproducer attempt for the lost update of packed fp32 FMAs found in round 4 (DESIGN.md "A hardware hazard found on the way", common.hpp::interp3).
// One work-group per CU, eight waves: waves 0-3 issue dependent MFMAs (argv[1] = 1) or exit (0); waves 4-7 each gather three random 1 KB rows
// per point (16 bytes per lane), then evaluate  r = fma(d, w2, fma(b, w1, a * w0))  with PER-LANE weights held in a VGPR triple -- plain vector
// arithmetic, which hipcc packs into v_pk_mul_f32 / v_pk_fma_f32 with op_sel operand selection, the form that lost updates in fp_ws_kernel --
// (argv[2] = 1: the scalar form of common.hpp::interp3 instead) and store r.  The host evaluates the same expression and counts the
// lane-elements that differ.
//   hipcc --offload-arch=gfx950 -O3 -o bin/pk_hazard pk_hazard.hip && bin/pk_hazard 1 0
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int GB = 16;        // 192 registers of rows in flight per gather lane

template <bool MFMA, bool SCALAR>
__global__ __launch_bounds__(512) void k(const float* __restrict__ rows, const int* __restrict__ idx, const float* __restrict__ wts, int iters,
                                         f32x4* __restrict__ out, float* __restrict__ sink) {
    __shared__ float pad[36 * 1024];          // 144 KB: one work-group per CU
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    for (int e = tid; e < 8192; e += 512) pad[e] = 0.0f;
    __syncthreads();
    if (wave < 4) {
        if (!MFMA) { for (int it = 0; it < iters; ++it) { __syncthreads(); __syncthreads(); } return; }
        f32x16 accs[12];          // 192 registers: the matrix wave of fp_ws_kernel holds 253
        for (int c = 0; c < 12; ++c) for (int r = 0; r < 16; ++r) accs[c][r] = (float)(tid + c);
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(lane + j); b[j] = (__bf16)1.0f; }
        for (int it = 0; it < iters; ++it) {
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int c = 0; c < 12; ++c) accs[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, accs[c], 0, 0, 0);
            for (int r = 0; r < 16; ++r) accs[0][r] += pad[(lane * 16 + r + it) & 8191];          // (LDS reads next to the gather waves' LDS stores)
            __syncthreads(); __syncthreads();
        }
        float s = 0;
        for (int c = 0; c < 12; ++c) for (int r = 0; r < 16; ++r) s += accs[c][r];
        sink[blockIdx.x * 512 + tid] = s;
        return;
    }
    const int gw = wave - 4;
    for (int it = 0; it < iters; ++it) {
        // a "point" = (work-group, iteration, gather wave, q, quarter of the wave): four lanes per point as in the round-3 mapping, so that
        // the weights are per-lane values in VGPRs
        const size_t pbase = ((((size_t)blockIdx.x * iters + it) * 4 + gw) * GB) * 16;
        f32x4 ga[GB], gb[GB], gd[GB];
        float w0[GB], w1[GB], w2[GB];
#pragma unroll
        for (int q = 0; q < GB; ++q) {
            const size_t pt = pbase + (size_t)q * 16 + (lane >> 2);
            const int ia = idx[pt * 3], ib = idx[pt * 3 + 1], ic = idx[pt * 3 + 2];
            w0[q] = wts[pt * 3]; w1[q] = wts[pt * 3 + 1]; w2[q] = wts[pt * 3 + 2];
            ga[q] = *reinterpret_cast<const f32x4*>(rows + (size_t)ia * 256 + 4 * (lane & 3));
            gb[q] = *reinterpret_cast<const f32x4*>(rows + (size_t)ib * 256 + 4 * (lane & 3));
            gd[q] = *reinterpret_cast<const f32x4*>(rows + (size_t)ic * 256 + 4 * (lane & 3));
        }
        __syncthreads();          // (the loads are in flight across the barrier, as in fp_ws_kernel; the wait for them follows it)
#pragma unroll
        for (int q = 0; q < GB; ++q) {
            f32x4 r;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (SCALAR) {
                    float t = ga[q][j] * w0[q];
                    asm("" : "+v"(t));
                    t = __builtin_fmaf(gb[q][j], w1[q], t);
                    asm("" : "+v"(t));
                    t = __builtin_fmaf(gd[q][j], w2[q], t);
                    asm("" : "+v"(t));
                    r[j] = t;
                } else {
                    r[j] = __builtin_fmaf(gd[q][j], w2[q], __builtin_fmaf(gb[q][j], w1[q], ga[q][j] * w0[q]));
                }
            }
            ga[q] = r;
        }
#pragma unroll
        for (int q = 0; q < GB; ++q) *reinterpret_cast<f32x4*>(pad + 8192 + ((gw * GB + q) * 64 + lane) * 4) = ga[q];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < GB; ++q) out[(pbase + (size_t)q * 16) * 4 + lane] = *reinterpret_cast<const f32x4*>(pad + 8192 + ((gw * GB + q) * 64 + lane) * 4);
    }
}

int main(int argc, char** argv) {
    const int mfma = argc > 1 ? atoi(argv[1]) : 1, scalar = argc > 2 ? atoi(argv[2]) : 0, iters = argc > 3 ? atoi(argv[3]) : 16, reps = argc > 4 ? atoi(argv[4]) : 30;
    const int nrows = 147456, G = 256;
    const size_t npts = (size_t)G * iters * 4 * GB * 16;
    std::vector<float> hr((size_t)nrows * 256), hw(npts * 3), want(npts * 16), got(npts * 16);
    std::vector<int> hi(npts * 3);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
    for (auto& v : hr) v = (float)(rnd() >> 8) / 16777216.0f * 8.0f - 4.0f;
    for (auto& v : hw) v = (float)(rnd() >> 8) / 16777216.0f;
    for (auto& v : hi) v = (int)(rnd() % (unsigned)nrows);
    for (size_t p = 0; p < npts; ++p)
        for (int l4 = 0; l4 < 4; ++l4)
            for (int j = 0; j < 4; ++j) {
                const int ch = 4 * l4 + j;
                const float a = hr[(size_t)hi[p * 3] * 256 + ch], b = hr[(size_t)hi[p * 3 + 1] * 256 + ch], d = hr[(size_t)hi[p * 3 + 2] * 256 + ch];
                // out index: (pbase + q*16)*4 + lane, lane = 4 * (point within the 16) + l4  ->  (p - p % 16) * 4 + (p % 16) * 4 + l4 = p * 4 + l4
                want[(p * 4 + l4) * 4 + j] = __builtin_fmaf(d, hw[p * 3 + 2], __builtin_fmaf(b, hw[p * 3 + 1], a * hw[p * 3]));
            }
    float *dr, *dw, *sink; int* di; f32x4* dout;
    (void)hipMalloc(&dr, hr.size() * 4); (void)hipMalloc(&dw, hw.size() * 4); (void)hipMalloc(&di, hi.size() * 4); (void)hipMalloc(&sink, G * 512 * 4);
    (void)hipMalloc(&dout, got.size() * 4);
    (void)hipMemcpy(dr, hr.data(), hr.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(di, hi.data(), hi.size() * 4, hipMemcpyHostToDevice);
    size_t bad = 0, shown = 0;
    for (int r = 0; r < reps; ++r) {
        (void)hipMemset(dout, 0, got.size() * 4);
        if (mfma && scalar) hipLaunchKernelGGL((k<true, true>), dim3(G), dim3(512), 0, 0, dr, di, dw, iters, dout, sink);
        else if (mfma) hipLaunchKernelGGL((k<true, false>), dim3(G), dim3(512), 0, 0, dr, di, dw, iters, dout, sink);
        else if (scalar) hipLaunchKernelGGL((k<false, true>), dim3(G), dim3(512), 0, 0, dr, di, dw, iters, dout, sink);
        else hipLaunchKernelGGL((k<false, false>), dim3(G), dim3(512), 0, 0, dr, di, dw, iters, dout, sink);
        (void)hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost);
        for (size_t e = 0; e < got.size(); ++e)
            if (memcmp(&got[e], &want[e], 4)) {
                ++bad;
                if (shown++ < 10) {
                    const size_t p = e / 16; const int l4 = (int)(e / 4 % 4), j = (int)(e % 4), ch = 4 * l4 + j;
                    const float a = hr[(size_t)hi[p * 3] * 256 + ch], b = hr[(size_t)hi[p * 3 + 1] * 256 + ch], d = hr[(size_t)hi[p * 3 + 2] * 256 + ch];
                    printf("   rep %d point %zu (lane %d of its wave) element %d: got %.9g want %.9g; a w0 + d w2 = %.9g, a w0 + b w1 = %.9g\n", r, p, (int)(p % 16) * 4 + l4, j, got[e],
                           want[e], __builtin_fmaf(d, hw[p * 3 + 2], a * hw[p * 3]), __builtin_fmaf(b, hw[p * 3 + 1], a * hw[p * 3]));
                }
            }
    }
    printf("MFMA waves %s, %s arithmetic: %zu of %.3g lane-elements differ from the host's evaluation (%d launches)\n", mfma ? "busy" : "absent", scalar ? "scalar" : "packed",
           bad, (double)reps * got.size(), reps);
    return bad ? 1 : 0;
}
