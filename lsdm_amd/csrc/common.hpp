// Shared device/host helpers for liblsdm_hip (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// resident FPS waves per SIMD (fps_chain_reg_kernel: one wave per cloud, everything in registers; 97 VGPRs at 4, 95 at 5, 80 at 6 -- all without spills; 7 spills)
#ifndef LSDM_FPS_WPE
#define LSDM_FPS_WPE 6
#endif

namespace lsdm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kPoints = 1024;      // points per cloud (model/sdm.py:21 pcd_points)
constexpr int kSlots = 9;          // human + 8 objects (posa/dataset.py:453)
constexpr int kNSample = 32;       // ball-query group size (pointnet2.py:46-49)

// exact-erf GELU (nn.GELU() default)
__device__ __forceinline__ float gelu(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float sigmoidf(float v) { return 1.0f / (1.0f + expf(-v)); }
__device__ __forceinline__ float siluf(float v) { return v / (1.0f + expf(-v)); }
// ReLU as ONE VALU instruction: a signed-integer max on the bit pattern (negative floats and -0 are negative integers).  The C
// forms `v > 0 ? v : 0` / fmaxf cost two (compare + select, or a canonicalising max before the max); an inline-asm v_max_f32
// is invisible to the compiler's MFMA -> VALU hazard handling and read stale accumulators (measured: nondeterministic results).
__device__ __forceinline__ float reluf(float v) {
    const int b = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}

// Activations of the split-bf16 dense stacks (bf16x3 / bf16 modes): the same functions evaluated with the hardware
// exp2 / reciprocal (1 ulp each) instead of libm's erff / expf / IEEE division -- the denoiser applies ~640 GELU / sigmoid
// evaluations per point and VALU time adds to MFMA time on gfx950.
//   GELU: Phi(v) through Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7), written so that the negative branch has no
//   cancellation: Phi(-|v|) = P(t) e^{-x^2} / 2,  Phi(|v|) = 1 - P(t) e^{-x^2} / 2,  x = |v| / sqrt(2), t = 1 / (1 + p x).
// Absolute error of gelu_fast <= |v| * 1.1e-7, of sigmoid_fast <= 3e-7; the exact-f32 mode keeps gelu / sigmoidf above.
__device__ __forceinline__ float gelu_fast(float v) {
    const float x = fabsf(v) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
    float pl = fmaf(t, 1.061405429f, -1.453152027f);
    pl = fmaf(t, pl, 1.421413741f);
    pl = fmaf(t, pl, -0.284496736f);
    pl = fmaf(t, pl, 0.254829592f);
    const float q = 0.5f * (pl * t) * __expf(-x * x);
    return v * (v < 0.0f ? q : 1.0f - q);
}
__device__ __forceinline__ float sigmoid_fast(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// Explicit GLOBAL load of a V from p.  A pointer that went through an opaque `asm("" : "+s"(p))` (used to keep loop-invariant
// loads inside a loop) has lost its address space: plain loads through it become FLAT loads, which count on vmcnt AND
// lgkmcnt and make the compiler wait with vmcnt(0) -- every younger prefetch is waited for as well.  (Neither a cast chain
// through address space 1 nor an assume(!is_shared && !is_private) brings the inference back with ROCm 7.2.)
template <class V, class T>
__device__ __forceinline__ V gload(const T* p) {
    return *(const __attribute__((address_space(1))) V*)(p);
}
// opaque copy of a pointer: same value, unknown to loop-invariant code motion
template <class T>
__device__ __forceinline__ const T* opaque_ptr(const T* p) {
    asm volatile("" : "+s"(p));
    return p;
}

// One wave-wide LDS-DMA copy: lane l's 16 bytes at gsrc -> LDS byte address lds_dst + 16 l (lds_dst wave-uniform).  Written
// as inline asm so that hipcc does not count it: with the builtin it waits vmcnt(0) before the next LDS read of ANY buffer,
// which makes the copy synchronous.  The issuer waits itself (lds_dma_wait) before the barrier that publishes the data.
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void lds_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <class T>
__device__ __forceinline__ unsigned lds_addr(const T* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) T*)p;
}

// Three-neighbour interpolation of one value (pointnet2_utils.py:299-303: torch.sum(points[idx] * w, dim=2)) with its rounding sequence
// SPELLED OUT -- a w0, then two fused multiply-adds -- and each step kept a SCALAR VALU instruction (the empty asm statements):
//  * written as plain arithmetic (or with __fmul_rn / __fadd_rn, which do not stop contraction) hipcc fuses one of the products into an
//    add, and WHICH one differs from call site to call site: two inlined copies of the same expression disagreed in the last bit (fp2's
//    three kernels on the second tile of a persistent work-group, round 4);
//  * left to itself hipcc packed the four values of a 16-byte piece into v_pk_mul_f32 / v_pk_fma_f32 with op_sel operand selection, and in
//    the gather waves of round 4's wave-specialised kernels (one gather and one MFMA wave per SIMD) those lost an update now and then
//    (the library is now built without compiler-packed fp32 altogether, csrc/Makefile).
__device__ __forceinline__ float interp3(float a, float b, float d, float w0, float w1, float w2) {
    float t = a * w0;
    asm("" : "+v"(t));
    t = __builtin_fmaf(b, w1, t);
    asm("" : "+v"(t));
    t = __builtin_fmaf(d, w2, t);
    asm("" : "+v"(t));
    return t;
}

// Work-group barrier of the persistent kernels that hand plain LDS stores from one wave to another: an explicit wait for the wave's LDS
// operations in front of it.  __syncthreads() alone was NOT enough in fp_ws_kernel: hipcc (ROCm 7.2, gfx950) emitted the barrier at the
// head of the tile loop without `s_waitcnt lgkmcnt(0)` although the loop's latch ends in ds_write instructions -- the other role's waves read
// the interpolated tile before the stores had landed (a few hundred wrong elements per launch under load; round 4, tools/race_hunt.py).
// Only lgkmcnt: the gather loads in flight across the barrier must not be waited for.
__device__ __forceinline__ void wg_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
}

// C/D map of the 32x32 MFMA shapes (dtype-independent on gfx950): col = lane & 31, row = acc_row(r, lane)
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// XCD-aware work-group remap.  MI355X dispatches work-group b to XCD b % 8 and every XCD has its own 4 MiB L2, so work-groups
// that share operands (the tiles of one cloud gather rows of the same level tensors) should sit on ONE XCD.  Returns a virtual
// id such that each XCD owns one contiguous range of ids; bijective for any grid size.  A pure speed choice: results do not
// depend on placement.
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}
// Round schedule of a PERSISTENT kernel (grid = a multiple of 8 work-groups, `rounds` units of work).  With one contiguous chunk of
// rounds per work-group, the work-groups of an XCD sit in 32 .. 64 different clouds at any time and the rows they gather (0.25 .. 0.5 MB
// per cloud) evict each other from the XCD's 4 MiB L2 (round 2 PMC: fp1 fetched 2.4x, sa2 2.2x its unique input from HBM).  Here the
// work-groups of one XCD walk ONE contiguous range of rounds together -- round i of work-group j of XCD x is x * rpx + i * wpx + j --
// so that at any time an XCD gathers from a handful of clouds.  Results do not depend on the schedule.
struct XcdRounds {
    int first, stride, end;
};
__device__ __forceinline__ XcdRounds xcd_rounds(int rounds) {
    const int G = (int)gridDim.x, v = xcd_remap((int)blockIdx.x, G);
    XcdRounds r;
    if (G & 7) {                     // not XCD-divisible: one contiguous chunk per work-group
        const int chunk = (rounds + G - 1) / G;
        r.first = v * chunk; r.stride = 1; r.end = r.first + chunk < rounds ? r.first + chunk : rounds;
        return r;
    }
    const int wpx = G >> 3, x = v / wpx, j = v - x * wpx, rpx = (rounds + 7) >> 3;
    r.first = x * rpx + j; r.stride = wpx; r.end = (x + 1) * rpx < rounds ? (x + 1) * rpx : rounds;
    return r;
}

// 1-D grid of tiles_per_cloud * num_clouds work-groups -> (cloud, tile)
__device__ __forceinline__ void xcd_cloud_tile(int tiles_per_cloud, int& cloud, int& tile) {
    const int v = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    cloud = v / tiles_per_cloud;
    tile = v - cloud * tiles_per_cloud;
}

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_SIGMOID = 3 };

template <int ACT>
__device__ __forceinline__ float apply_act_fast(float v) {
    if constexpr (ACT == ACT_RELU) return reluf(v);
    else if constexpr (ACT == ACT_GELU) return gelu_fast(v);
    else if constexpr (ACT == ACT_SIGMOID) return sigmoid_fast(v);
    else return v;
}

template <int ACT>
__device__ __forceinline__ float apply_act(float v) {
    if constexpr (ACT == ACT_RELU) return reluf(v);
    else if constexpr (ACT == ACT_GELU) return gelu(v);
    else if constexpr (ACT == ACT_SIGMOID) return sigmoidf(v);
    else return v;
}

// pointnet2_utils.py:19-38 square_distance, in the fp32 rounding sequence of the CPU reference for K=3
// (oracle/cpu_ref.py::square_distance, pinned by tests/test_oracle_golden.py::test_g3_square_distance_rounding):
//   dot = fma(a2,b2, fma(a1,b1, a0*b0));  d = -2*dot;  d += |a|^2;  d += |b|^2,  |v|^2 = (v0^2+v1^2)+v2^2
__device__ __forceinline__ float sumsq3(float x, float y, float z) {
    return __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
}
__device__ __forceinline__ float sqdist_ref(float ax, float ay, float az, float a2, float bx, float by, float bz, float b2) {
    float dot = __fmul_rn(ax, bx);
    dot = __fmaf_rn(ay, by, dot);
    dot = __fmaf_rn(az, bz, dot);
    float d = __fmul_rn(-2.0f, dot);
    d = __fadd_rn(d, a2);
    d = __fadd_rn(d, b2);
    return d;
}

}  // namespace lsdm
