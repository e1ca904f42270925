// Microbenchmark (gfx950): can VALU work hide behind MFMA work on one SIMD?
//   A: MFMA only   B: VALU only   C: one wave, fine interleave (1 MFMA : V VALU)   D: two waves per SIMD, one MFMA-only + one VALU-only
//   E: like C but the VALU block follows a block of 8 MFMAs (coarse interleave)
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu.hip -o gpurun_out/ubench_mfma_valu && gpurun_out/ubench_mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0)


#define VOP4(INS) asm volatile(INS " %0, %0, %4\n\t" INS " %1, %1, %4\n\t" INS " %2, %2, %4\n\t" INS " %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(mm), "v"(cc))
template <int OP>
__device__ __forceinline__ void vops(float& a0, float& a1, float& a2, float& a3, float mm, float cc) {
    if constexpr (OP == 0) asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(mm), "v"(cc));
    else if constexpr (OP == 1) VOP4("v_max_f32");
    else if constexpr (OP == 2) VOP4("v_and_b32");
    else if constexpr (OP == 3) VOP4("v_add_f32");
    else if constexpr (OP == 4) VOP4("v_cvt_pk_bf16_f32");
    else if constexpr (OP == 5) VOP4("v_lshlrev_b32");
    else if constexpr (OP == 6) VOP4("v_mul_f32");
    else if constexpr (OP == 7) VOP4("v_max_i32");
    else if constexpr (OP == 8) VOP4("v_pk_max_i16");
    else if constexpr (OP == 9) VOP4("v_sub_f32");
    else if constexpr (OP == 10) asm volatile("v_perm_b32 %0, %0, %4, %5\n\tv_perm_b32 %1, %1, %4, %5\n\tv_perm_b32 %2, %2, %4, %5\n\tv_perm_b32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(mm), "v"(cc));
    else if constexpr (OP == 11) asm volatile("v_and_or_b32 %0, %0, %4, %5\n\tv_and_or_b32 %1, %1, %4, %5\n\tv_and_or_b32 %2, %2, %4, %5\n\tv_and_or_b32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(mm), "v"(cc));
    else if constexpr (OP == 12) { typedef float f2 __attribute__((ext_vector_type(2))); f2 p = {a0, a1}, q = {a2, a3}, w = {mm, cc};
        asm volatile("v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %2\n\tv_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %2" : "+v"(p), "+v"(q) : "v"(w));
        a0 = p[0]; a1 = p[1]; a2 = q[0]; a3 = q[1]; }
    else if constexpr (OP == 13) VOP4("v_lshrrev_b32");
    else if constexpr (OP == 14) VOP4("v_cndmask_b32");
    else if constexpr (OP == 15) asm volatile("v_alignbit_b32 %0, %0, %4, %5\n\tv_alignbit_b32 %1, %1, %4, %5\n\tv_alignbit_b32 %2, %2, %4, %5\n\tv_alignbit_b32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(mm), "v"(cc));
}
#define VOPS(a,b,c_,d,m_,cc_) vops<OP>(a,b,c_,d,m_,cc_)
template <int MODE, int V, int OP>
__global__ __launch_bounds__(512) void k(float* out, int iters, int waves_mfma) {
    const int wave = threadIdx.x >> 6;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 0.001f + j); b[j] = (__bf16)(j * 0.5f); }
    f32x16 acc0 = {}, acc1 = {};
    float v0 = threadIdx.x, v1 = 1.0f, v2 = 2.0f, v3 = 3.0f;
    const float m = 1.0001f, c = 0.5f;
    const bool do_m = (MODE == 0) || (MODE == 2) || (MODE == 4) || (MODE == 3 && wave < waves_mfma);
    const bool do_v = (MODE == 1) || (MODE == 2) || (MODE == 4) || (MODE == 3 && wave >= waves_mfma);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (u & 1) MFMA(acc1, a, b); else MFMA(acc0, a, b);
#pragma unroll
                for (int q = 0; q < V / 4; ++q) {
                    VOPS(v0, v1, v2, v3, m, c);
                }
            }
        } else {
            if (do_m) {
#pragma unroll
                for (int u = 0; u < 8; ++u) { if (u & 1) MFMA(acc1, a, b); else MFMA(acc0, a, b); }
            }
            if (do_v) {
#pragma unroll
                for (int q = 0; q < 8 * V / 4; ++q) {
                    VOPS(v0, v1, v2, v3, m, c);
                }
            }
        }
    }
    float s = v0 + v1 + v2 + v3;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int V, int OP = 0>
float run(float* d, int threads, int iters, int wm) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, V, OP>), dim3(256), dim3(threads), 0, 0, d, 10, wm);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, V, OP>), dim3(256), dim3(threads), 0, 0, d, iters, wm);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    const int it = 20000;
    // cycles per iteration at 2.4 GHz (8 MFMAs + 8*V VALU per iteration)
    auto cyc = [&](float ms) { return ms * 1e-3 * 2.4e9 / it; };
    printf("one wave per SIMD (256 threads):\n");
    printf("  A mfma only            : %8.1f cyc/iter (8 MFMA 32x32x16 bf16)\n", cyc(run<0, 8>(d, 256, it, 0)));
    printf("  B valu only V=8        : %8.1f cyc/iter (64 v_fma)\n", cyc(run<1, 8>(d, 256, it, 0)));
    printf("  B valu only V=4        : %8.1f cyc/iter (32 v_fma)\n", cyc(run<1, 4>(d, 256, it, 0)));
    printf("  C interleave 1:8       : %8.1f\n", cyc(run<2, 8>(d, 256, it, 0)));
    printf("  C interleave 1:4       : %8.1f\n", cyc(run<2, 4>(d, 256, it, 0)));
    printf("  E coarse 8 MFMA + 64 V : %8.1f\n", cyc(run<4, 8>(d, 256, it, 0)));
    printf("  E coarse 8 MFMA + 32 V : %8.1f\n", cyc(run<4, 4>(d, 256, it, 0)));
    printf("two waves per SIMD (512 threads), waves 0-3 MFMA, 4-7 VALU:\n");
    printf("  D V=8                  : %8.1f\n", cyc(run<3, 8>(d, 512, it, 4)));
    printf("  D V=4                  : %8.1f\n", cyc(run<3, 4>(d, 512, it, 4)));
    printf("two waves per SIMD, both MFMA  : %8.1f\n", cyc(run<0, 8>(d, 512, it, 0)));
    printf("two waves per SIMD, both VALU 8: %8.1f\n", cyc(run<1, 8>(d, 512, it, 0)));
    printf("two waves per SIMD, both interleave 1:8: %8.1f\n", cyc(run<2, 8>(d, 512, it, 0)));
    printf("two waves per SIMD, both coarse 8+64   : %8.1f\n", cyc(run<4, 8>(d, 512, it, 0)));

    const char* names[16] = {"v_fma_f32", "v_max_f32", "v_and_b32", "v_add_f32", "v_cvt_pk_bf16_f32", "v_lshlrev_b32", "v_mul_f32", "v_max_i32", "v_pk_max_i16", "v_sub_f32", "v_perm_b32", "v_and_or_b32", "v_pk_add_f32(4 ops=64 instr)", "v_lshrrev_b32", "v_cndmask_b32", "v_alignbit_b32"};
    printf("per VALU op: B alone (1 wave/SIMD, 64 ops) | D (MFMA wave + VALU wave) | two waves both interleaved 1:8 | both VALU only\n");
#define ROW(OP) printf("  %-20s %8.1f %8.1f %8.1f %8.1f\n", names[OP], cyc(run<1, 8, OP>(d, 256, it, 0)), cyc(run<3, 8, OP>(d, 512, it, 4)), cyc(run<2, 8, OP>(d, 512, it, 0)), cyc(run<1, 8, OP>(d, 512, it, 0)))
    ROW(0); ROW(1); ROW(2); ROW(3); ROW(4); ROW(5); ROW(6); ROW(7); ROW(8); ROW(9); ROW(10); ROW(11); ROW(12); ROW(13); ROW(14); ROW(15);
    return 0;
}
