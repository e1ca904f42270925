// Split-bf16 MFMA tile core (gfx950): the dense layers of the path on v_mfma_f32_32x32x16_bf16.
//
// fp32 operands are carried as one or two bf16 planes:
//   NP = 2 ("bf16x3"): a = a_hi + a_lo (a_hi = bf16(a), a_lo = bf16(a - a_hi)); a product a*w is evaluated as
//                      a_lo*w_hi + a_hi*w_lo + a_hi*w_hi with fp32 accumulation (the dropped a_lo*w_lo term is
//                      2^-18 relative) -- fp32-class results at 16/3 = 5.3x the rate of the f32-input MFMA;
//   NP = 1 ("bf16")  : a_hi*w_hi only, the full 16x rate.
// Layout conventions
//   * activations live in LDS as [row = point][k] bf16 with a leading dimension of 8*odd elements (16*odd bytes):
//     the ds_read_b128 of a fragment (32 rows x 2 k-halves) is bank-conflict free.  The lo plane follows the hi
//     plane at a fixed element offset (`plane`).
//   * weights are packed on the host as Wp[kb][n][NP][16] (k = 16*kb + j, K zero-padded to 16), so one wave's fragment
//     load of a 32-column block is a contiguous 1 KiB (NP = 1) / 2 KiB (NP = 2) segment served from L2.
//   * both operands index k as (lane>>5)*8 + j, j = 0..7 (one 16-byte load per lane per plane).
//   * a layer is evaluated TRANSPOSED by default (weights = MFMA A operand, activations = B operand), D[n][point]:
//     a lane then owns 4 consecutive output channels of one point per accumulator quad, so the epilogue
//     (bias + activation + bf16 split) packs 4 channels into one 8-byte LDS store of the next layer's operand.
//     The last layer of a stack may use the direct orientation D[point][n] (max-pool over points in-lane; 32
//     consecutive channels per half-wave for global stores).
#pragma once
#include "common.hpp"

namespace lsdm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct DenseH {  // packed bf16 planes Wp[kb][n][NP][16] + fp32 bias[n]
    const __bf16* w;
    const float* b;
};

// element type of the level features in HBM: fp32 for bf16x3 (same workspace as the f32 path), bf16 for plain bf16
template <int NP> struct FeatT { typedef float type; };
template <> struct FeatT<1> { typedef __bf16 type; };

__device__ __forceinline__ f32x4 load4f(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 load4f(const __bf16* p) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    f32x4 r = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    return r;
}
__device__ __forceinline__ void store1(float* p, float v) { *p = v; }
__device__ __forceinline__ void store1(__bf16* p, float v) { *p = (__bf16)v; }

__host__ __device__ constexpr int round16(int k) { return (k + 15) / 16 * 16; }
// LDS leading dimension (bf16 elements) for a logical width k (multiple of 8): (ld/8) odd
__host__ __device__ constexpr int ldh(int k) { return ((k / 8) % 2 == 1) ? k : k + 8; }
__host__ __device__ constexpr int imax(int a, int b) { return a > b ? a : b; }

__device__ __forceinline__ float bf_hi_as_float(__bf16 h) { return (float)h; }

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// hi / lo split of a PAIR of fp32 values: hi = bf16(v) (one v_cvt_pk_bf16_f32 for both), lo = bf16(v - hi) with v - hi formed by
// v_dot2c_f32_bf16 straight from the packed hi pair (hi x (-1, 0) + v: exact, the difference is representable) -- no unpacking
// of hi back to fp32: 3 VALU instructions per pair of values less than the convert / shift / subtract form.  (v_dot2c executes on
// the matrix pipe's side -- profiles/r03_ubench_mfma_fill.txt: 4 of them per MFMA gap cost 47 ticks against 26 for plain VALU --
// but the plain-VALU form measured no faster in any kernel, block-placed or software-pipelined: sa3 326 vs 323 us, sa2 305 vs 303.)
template <int NP>
__device__ __forceinline__ void split2(float v0, float v1, bf16x2& hi, bf16x2& lo) {
    hi[0] = (__bf16)v0; hi[1] = (__bf16)v1;
    if constexpr (NP == 2) {
        // the multipliers (-1, 0) / (0, -1) must sit in REGISTERS: written as literals hipcc (ROCm 7.2) folds them into the
        // inline constant -1.0, which the instruction does not read as a bf16 pair (tools/ubench/split_test.hip: half of
        // all lo words wrong); the empty asm makes the values opaque
        unsigned u0 = 0x0000bf80u, u1 = 0xbf800000u;
        asm("" : "+s"(u0), "+s"(u1));
        const bf16x2 m0 = __builtin_bit_cast(bf16x2, u0), m1 = __builtin_bit_cast(bf16x2, u1);
        const float l0 = __builtin_amdgcn_fdot2_f32_bf16(hi, m0, v0, false);
        const float l1 = __builtin_amdgcn_fdot2_f32_bf16(hi, m1, v1, false);
        lo[0] = (__bf16)l0; lo[1] = (__bf16)l1;
    }
}

// split 4 fp32 values into bf16 hi (and lo) quads
template <int NP>
__device__ __forceinline__ void split4(float v0, float v1, float v2, float v3, bf16x4& hi, bf16x4& lo) {
    bf16x2 h0, l0, h1, l1;
    split2<NP>(v0, v1, h0, l0);
    split2<NP>(v2, v3, h1, l1);
    hi[0] = h0[0]; hi[1] = h0[1]; hi[2] = h1[0]; hi[3] = h1[1];
    if constexpr (NP == 2) { lo[0] = l0[0]; lo[1] = l0[1]; lo[2] = l1[0]; lo[3] = l1[1]; }
}

// split 8 fp32 values into the two bf16x8 operand fragments of a k-step
template <int NP>
__device__ __forceinline__ void split8_h(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        bf16x2 h, l;
        split2<NP>(v[j], v[j + 1], h, l);
        hi[j] = h[0]; hi[j + 1] = h[1];
        if constexpr (NP == 2) { lo[j] = l[0]; lo[j + 1] = l[1]; }
    }
}

// store 4 consecutive fp32 values as bf16 planes at dst (8-byte aligned)
template <int NP>
__device__ __forceinline__ void store_split4(__bf16* dst, int plane, float v0, float v1, float v2, float v3) {
    bf16x4 hi, lo;
    split4<NP>(v0, v1, v2, v3, hi, lo);
    *reinterpret_cast<bf16x4*>(dst) = hi;
    if constexpr (NP == 2) *reinterpret_cast<bf16x4*>(dst + plane) = lo;
}

// value of an LDS activation (hi [+ lo])
template <int NP>
__device__ __forceinline__ float lds_val(const __bf16* p, int plane) {
    float v = (float)p[0];
    if constexpr (NP == 2) v += (float)p[plane];
    return v;
}

// acc[rb] += X[rb-th 32-point block, k-blocks kb0..kb1) (x) W(columns cb*32 .. +32)
//   X     : LDS hi plane, row 0 of the wave's first point block, k = x_kb_off*16 at column 0
//   WA    : true  -> D[n][point]  (weights are the A operand)
//           false -> D[point][n]  (activations are the A operand)
// k-blocks are processed in groups of U; the next group's weight fragments are requested before the current group's MFMAs
__host__ __device__ constexpr int kgroup(int nkb) { return nkb <= 5 ? nkb : (nkb % 4 == 0 ? 4 : (nkb % 3 == 0 ? 3 : 4)); }

template <int NP, int U>
struct WFrag {   // weight fragments of one k-group for one 32-channel block (hi [+ lo] planes)
    bf16x8 h[U];
    bf16x8 l[NP == 2 ? U : 1];
};

// request the FIRST k-group of (layer Wp with N channels, channel block cb) -- callable long before the K loop
// (ahead of a gather, or before the previous layer's epilogue) so that its L2 latency is off the critical path
template <int NP, int NKB>
__device__ __forceinline__ void wfrag_load(WFrag<NP, kgroup(NKB)>& w, const __bf16* __restrict__ Wp, int N, int cb, int kb0, int lane) {
    constexpr int U = kgroup(NKB);
    const size_t ws = (size_t)N * NP * 16;
    const __bf16* wp = Wp + ((size_t)(cb * 32 + (lane & 31)) * NP) * 16 + 8 * (lane >> 5) + (size_t)kb0 * ws;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        w.h[u] = gload<bf16x8>(wp + (size_t)u * ws);
        if constexpr (NP == 2) w.l[u] = gload<bf16x8>(wp + (size_t)u * ws + 16);
    }
}

// acc[rb] += X[rb-th 32-point block, NKB k-blocks from weight k-block kb0) (x) W(channels cb*32 .. +32)
//   X     : LDS hi plane, row 0 of the wave's first point block, at the column of the first k-block
//   w     : the first k-group, already requested with wfrag_load (clobbered)
//   WA    : true  -> D[n][point]  (weights are the MFMA A operand);  false -> D[point][n]
template <int RB, int NP, bool WA, int NKB>
__device__ __forceinline__ void mma_kloop_pre_h(f32x16 (&acc)[RB], const __bf16* X, int ldx, int plane, const __bf16* __restrict__ Wp,
                                                int N, int cb, int kb0, int lane, WFrag<NP, kgroup(NKB)>& w) {
    constexpr int U = kgroup(NKB);
    constexpr int NG = NKB / U, TAIL = NKB % U;
    const int i = lane & 31, h = lane >> 5;
    const __bf16* xp = X + i * ldx + 8 * h;
    const size_t ws = (size_t)N * NP * 16;
    const __bf16* wp = Wp + ((size_t)(cb * 32 + i) * NP) * 16 + 8 * h + (size_t)kb0 * ws;
    // Weight fragments are requested TWO k-groups ahead (the loaded L2 latency is a multiple of the ~0.8k cycles one group
    // of MFMAs lasts): `w` (group 0, requested by the caller) and `w2` (group 1) are in flight on entry; the buffer a group
    // is copied out of is refilled at once with the group after next.  The tail group is requested up front as well.
    WFrag<NP, U> w2;
    if constexpr (NG > 1) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            w2.h[u] = gload<bf16x8>(wp + (size_t)(U + u) * ws);
            if constexpr (NP == 2) w2.l[u] = gload<bf16x8>(wp + (size_t)(U + u) * ws + 16);
        }
    }
    [[maybe_unused]] WFrag<NP, kgroup(TAIL > 0 ? TAIL : 1)> wt;
    if constexpr (TAIL > 0) wfrag_load<NP, TAIL>(wt, Wp, N, cb, kb0 + NG * U, lane);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        WFrag<NP, U>& src = (g & 1) ? w2 : w;
        bf16x8 ch[U], cl[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { ch[u] = src.h[u]; if constexpr (NP == 2) cl[u] = src.l[u]; }
        if (g + 2 < NG) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                src.h[u] = gload<bf16x8>(wp + (size_t)((g + 2) * U + u) * ws);
                if constexpr (NP == 2) src.l[u] = gload<bf16x8>(wp + (size_t)((g + 2) * U + u) * ws + 16);
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of this group's MFMAs
#pragma unroll
        for (int u = 0; u < U; ++u) {
            bf16x8 xh[RB], xl[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const __bf16* p = xp + rb * 32 * ldx + (g * U + u) * 16;
                xh[rb] = *reinterpret_cast<const bf16x8*>(p);
                if constexpr (NP == 2) xl[rb] = *reinterpret_cast<const bf16x8*>(p + plane);
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                if constexpr (WA) {
                    if constexpr (NP == 2) {
                        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch[u], xl[rb], acc[rb], 0, 0, 0);
                        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cl[u], xh[rb], acc[rb], 0, 0, 0);
                    }
                    acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ch[u], xh[rb], acc[rb], 0, 0, 0);
                } else {
                    if constexpr (NP == 2) {
                        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl[rb], ch[u], acc[rb], 0, 0, 0);
                        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[rb], cl[u], acc[rb], 0, 0, 0);
                    }
                    acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[rb], ch[u], acc[rb], 0, 0, 0);
                }
            }
        }
    }
    if constexpr (TAIL > 0) mma_kloop_pre_h<RB, NP, WA, TAIL>(acc, X + NG * U * 16, ldx, plane, Wp, N, cb, kb0 + NG * U, lane, wt);
}

template <int RB, int NP, bool WA, int NKB>
__device__ __forceinline__ void mma_kloop_h(f32x16 (&acc)[RB], const __bf16* X, int ldx, int plane, const __bf16* __restrict__ Wp,
                                            int N, int cb, int kb0, int lane) {
    WFrag<NP, kgroup(NKB)> w;
    wfrag_load<NP, NKB>(w, Wp, N, cb, kb0, lane);
    mma_kloop_pre_h<RB, NP, WA, NKB>(acc, X, ldx, plane, Wp, N, cb, kb0, lane, w);
}

template <int RB>
__device__ __forceinline__ void acc_zero_h(f32x16 (&acc)[RB]) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][r] = 0.0f;
}

// Work split of a (TM points) x (N channels) layer over NW waves: channel blocks first, then point groups.
template <int NW, int TM, int N, int CS = 0>
struct SplitH {
    static constexpr int RBT = TM / 32, CB = N / 32;
    static constexpr int CSTEP = CS > 0 ? CS : ((CB >= NW) ? NW : CB);   // channel blocks processed concurrently
    static constexpr int WPC = NW / CSTEP;               // waves sharing one channel block (split points)
    static constexpr int RB = RBT / WPC;                 // point blocks per wave
    static_assert(TM % 32 == 0 && N % 32 == 0, "tile dims");
    static_assert(NW % CSTEP == 0 && RBT % WPC == 0 && RB >= 1, "tile split");
};

// bias quads of a 32-channel block in the lane order of the transposed epilogue (requested before the K loop so that
// their latency hides behind the MFMAs)
__device__ __forceinline__ void load_bias_h(f32x4 (&bv)[4], const float* __restrict__ bias, int cb, int lane) {
#pragma unroll
    for (int q = 0; q < 4; ++q) bv[q] = gload<f32x4>(bias + cb * 32 + 8 * q + 4 * (lane >> 5));
}

// accumulators of a transposed block start from the bias (row = channel 8q + 4h + j of accumulator 4q + j): the epilogue
// then needs no add
template <int RB>
__device__ __forceinline__ void acc_bias_h(f32x16 (&acc)[RB], const f32x4 (&bv)[4]) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[rb][4 * q + j] = bv[q][j];
}

// epilogue of a transposed (WA) layer whose accumulators started from the bias: activation + split -> LDS [point][channel]
template <int ACT, int RB, int NP>
__device__ __forceinline__ void store_lds_b_h(const f32x16 (&acc)[RB], int rb0, int cb, __bf16* out, int ldo, int plane, int lane) {
    const int p = lane & 31, h = lane >> 5;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = cb * 32 + 8 * q + 4 * h;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            store_split4<NP>(out + ((rb0 + rb) * 32 + p) * ldo + n, plane, apply_act_fast<ACT>(acc[rb][4 * q + 0]),
                             apply_act_fast<ACT>(acc[rb][4 * q + 1]), apply_act_fast<ACT>(acc[rb][4 * q + 2]),
                             apply_act_fast<ACT>(acc[rb][4 * q + 3]));
        }
    }
}

// epilogue of a transposed (WA) layer: bias + activation + split -> LDS [point][channel]
template <int ACT, int RB, int NP>
__device__ __forceinline__ void store_lds_h(const f32x16 (&acc)[RB], int rb0, int cb, const f32x4 (&bv)[4], __bf16* out, int ldo,
                                            int plane, int lane) {
    const int p = lane & 31, h = lane >> 5;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = cb * 32 + 8 * q + 4 * h;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            store_split4<NP>(out + ((rb0 + rb) * 32 + p) * ldo + n, plane, apply_act_fast<ACT>(acc[rb][4 * q + 0] + bv[q][0]),
                             apply_act_fast<ACT>(acc[rb][4 * q + 1] + bv[q][1]), apply_act_fast<ACT>(acc[rb][4 * q + 2] + bv[q][2]),
                             apply_act_fast<ACT>(acc[rb][4 * q + 3] + bv[q][3]));
        }
    }
}

// One dense layer, whole K resident in LDS: out(LDS) = act(W x X + bias), transposed orientation
// (waves >= NW of a larger workgroup take no part)
template <int NW, int TM, int K, int N, int ACT, int NP, int CS = 0>
__device__ __forceinline__ void layer_lds_h(const __bf16* X, int ldx, int xplane, DenseH L, __bf16* out, int ldo, int oplane) {
    using TS = SplitH<NW, TM, N, CS>;
    static_assert(K % 16 == 0, "K must be padded to 16");
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= NW) return;
    const int cb0 = wave % TS::CSTEP, rg = wave / TS::CSTEP;
#pragma unroll 1
    for (int cb = cb0; cb < TS::CB; cb += TS::CSTEP) {
        f32x16 acc[TS::RB];
        f32x4 bv[4];
        load_bias_h(bv, L.b, cb, lane);
        acc_bias_h(acc, bv);
        mma_kloop_h<TS::RB, NP, true, K / 16>(acc, X + rg * TS::RB * 32 * ldx, ldx, xplane, L.w, N, cb, 0, lane);
        store_lds_b_h<ACT, TS::RB, NP>(acc, rg * TS::RB, cb, out, ldo, oplane, lane);
    }
}

// Same layer with the first channel block's first k-group and bias already in flight (w0, bv0), and a hook that runs
// between the first block's K loop and its epilogue -- the place to request the NEXT layer's first k-group, so that the
// weight fetch of layer l+1 overlaps the epilogue + barrier of layer l.
template <int NW, int TM, int K, int N, int ACT, int NP, class Hook>
__device__ __forceinline__ void layer_lds_pre_h(const __bf16* X, int ldx, int xplane, DenseH L, __bf16* out, int ldo, int oplane,
                                                WFrag<NP, kgroup(K / 16)>& w0, f32x4 (&bv0)[4], Hook&& hook) {
    using TS = SplitH<NW, TM, N>;
    static_assert(K % 16 == 0, "K must be padded to 16");
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= NW) { hook(); return; }
    const int cb0 = wave % TS::CSTEP, rg = wave / TS::CSTEP;
    {
        f32x16 acc[TS::RB];
        acc_bias_h(acc, bv0);
        mma_kloop_pre_h<TS::RB, NP, true, K / 16>(acc, X + rg * TS::RB * 32 * ldx, ldx, xplane, L.w, N, cb0, 0, lane, w0);
        hook();
        store_lds_b_h<ACT, TS::RB, NP>(acc, rg * TS::RB, cb0, out, ldo, oplane, lane);
    }
#pragma unroll 1
    for (int cb = cb0 + TS::CSTEP; cb < TS::CB; cb += TS::CSTEP) {
        f32x16 acc[TS::RB];
        f32x4 bv[4];
        load_bias_h(bv, L.b, cb, lane);
        acc_bias_h(acc, bv);
        mma_kloop_h<TS::RB, NP, true, K / 16>(acc, X + rg * TS::RB * 32 * ldx, ldx, xplane, L.w, N, cb, 0, lane);
        store_lds_b_h<ACT, TS::RB, NP>(acc, rg * TS::RB, cb, out, ldo, oplane, lane);
    }
}

// ---------------------------------------------------------------------------------------------
// Pooling of a 32-row block that two ball-query groups share (the SA kernels' item lists; geometry.hip: sa_pair_kernel).
// two maxima at once: lanes 0-31 get max(a[l], a[l + 32]), lanes 32-63 get max(b[l - 32], b[l]) (one swap: afterwards a = [a.lo, b.lo], b = [a.hi, b.hi])
__device__ __forceinline__ float max_halves2(float a, float b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}

// the pooled row(s) of a split block (PAIR): lanes 0-31 get the maximum over A's rows, lanes 32-63 over B's (an unpaired item: the maximum
// over all 32 rows in both).  own[g]: register quartet g of this lane's half-wave belongs to A.
__device__ __forceinline__ float max_split(const f32x16& a, const bool (&own)[4], bool paired) {
    float ma = -INFINITY, mb = -INFINITY;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float m4 = fmaxf(fmaxf(a[4 * g], a[4 * g + 1]), fmaxf(a[4 * g + 2], a[4 * g + 3]));
        ma = fmaxf(ma, own[g] ? m4 : -INFINITY);
        mb = fmaxf(mb, own[g] ? -INFINITY : m4);
    }
    return max_halves2(ma, paired ? mb : ma);
}


}  // namespace lsdm
