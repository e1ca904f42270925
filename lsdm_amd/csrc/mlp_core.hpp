// fp32 MFMA tile core shared by every dense layer of the path (gfx950).
//
// A layer  Y = act(X * W^T + b)  (X: TM x K rows in LDS, W: N x K as in torch.nn.Linear / 1x1 conv)
// is evaluated with v_mfma_f32_32x32x2_f32 -- exact fp32 (bitwise an fmaf chain), 64 cycles/SIMD,
// 157 TF chip peak (MI355X_MICROARCH.md).  Operand feeding:
//   * K is processed in blocks of 8.  Inside a block the lane half h = lane>>5 owns k = 8*kb + 4*h + s
//     for MFMA step s = 0..3, so one float4 per lane feeds four MFMAs for both operands.
//   * A (activations) : LDS, row-major with a leading dimension whose /4 is odd -> the ds_read_b128 of
//     32 rows x 2 halves is bank-conflict free.
//   * B (weights)     : global memory, pre-packed by the host as Wp[kb][n][8] (pack_weight in api.hip),
//     so the wave's float4 load is one contiguous 1 KiB segment; weights stay L2 / Infinity-Cache
//     resident across workgroups.  The next block's B is fetched before the current block's MFMAs.
//   * C/D accumulators: f32x16 per 32x32 block: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
#pragma once
#include "common.hpp"

namespace lsdm {

// leading dimension for an LDS activation buffer of logical width k (multiple of 4): (ld/4) odd
__host__ __device__ constexpr int lds_ld(int k) { return ((k / 4) % 2 == 1) ? k : k + 4; }

// acc[rb] += A[rb*32 .. +32, kb0*8 .. kb1*8) * Wp(cols cb*32 .. +32)
//   A  : LDS pointer to (row 0 of the first row block, k = 0)
//   Wp : global pointer to packed weights of this layer; N = number of output columns of the layer
template <int RB>
__device__ __forceinline__ void mma_kloop(f32x16 (&acc)[RB], const float* A, int lda, const float* __restrict__ Wp, int N,
                                          int cb, int kb0, int kb1, int a_kb_off, int lane) {
    const int i = lane & 31, h = lane >> 5;
    const float* ap = A + i * lda + 4 * h;
    const float* wp = Wp + ((size_t)(cb * 32 + i) * 8 + 4 * h);
    f32x4 bnext = *reinterpret_cast<const f32x4*>(wp + (size_t)kb0 * N * 8);
    for (int kb = kb0; kb < kb1; ++kb) {
        const f32x4 b = bnext;
        if (kb + 1 < kb1) bnext = *reinterpret_cast<const f32x4*>(wp + (size_t)(kb + 1) * N * 8);
        f32x4 a[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) a[rb] = *reinterpret_cast<const f32x4*>(ap + rb * 32 * lda + (kb - a_kb_off) * 8);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rb][s], b[s], acc[rb], 0, 0, 0);
        }
    }
}

template <int RB>
__device__ __forceinline__ void acc_zero(f32x16 (&acc)[RB]) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rb][r] = 0.0f;
}

// Work split of a TM x N output tile over NW waves: column blocks first, then row groups.
template <int NW, int TM, int N>
struct TileSplit {
    static constexpr int RBT = TM / 32, CB = N / 32;
    static constexpr int CSTEP = (CB >= NW) ? NW : CB;   // column blocks processed concurrently
    static constexpr int WPC = NW / CSTEP;               // waves sharing one column block (split rows)
    static constexpr int RB = RBT / WPC;                 // row blocks per wave
    static_assert(TM % 32 == 0 && N % 32 == 0, "tile dims");
    static_assert(NW % CSTEP == 0 && RBT % WPC == 0, "tile split");
};

// epilogue: bias + activation -> LDS buffer (next layer's A operand)
template <int ACT, int RB>
__device__ __forceinline__ void store_lds(const f32x16 (&acc)[RB], int rb0, int cb, const float* __restrict__ bias, float* out,
                                          int ldo, int lane) {
    const int col = cb * 32 + (lane & 31);
    const float bv = bias[col];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[((rb0 + rb) * 32 + acc_row(r, lane)) * ldo + col] = apply_act<ACT>(acc[rb][r] + bv);
}

// One dense layer, whole K resident in LDS:  out(LDS) = act(A * W^T + bias)
template <int NW, int TM, int K, int N, int ACT>
__device__ __forceinline__ void layer_lds(const float* A, int lda, const float* __restrict__ Wp, const float* __restrict__ bias,
                                          float* out, int ldo) {
    using TS = TileSplit<NW, TM, N>;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int cb0 = wave % TS::CSTEP, rg = wave / TS::CSTEP;
    for (int cb = cb0; cb < TS::CB; cb += TS::CSTEP) {
        f32x16 acc[TS::RB];
        acc_zero(acc);
        mma_kloop<TS::RB>(acc, A + rg * TS::RB * 32 * lda, lda, Wp, N, cb, 0, K / 8, 0, lane);
        store_lds<ACT, TS::RB>(acc, rg * TS::RB, cb, bias, out, ldo, lane);
    }
}

}  // namespace lsdm
