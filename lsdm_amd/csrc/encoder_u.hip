// Set abstraction with the first shared-MLP layer SPLIT across the grouping (bf16x3 mode, levels sa3 / sa4).
//
// sample_and_group feeds the first 1x1 conv with [xyz_j - c_i || f_j] for the 32 neighbours j of centroid i
// (pointnet2_utils.py:119-135,189-193).  The conv (+ folded BatchNorm) is linear in that input, so with W = [Wx | Wf]
//     W x [x_j - c_i ; f_j] + b  =  (Wf x f_j + Wx x x_j + b)  -  Wx x c_i  =  U_j - V_i ,      c_i = x_{idx_i}  =>  V_i = V_{idx_i}
// with U, V evaluated ONCE per source point (N rows) instead of once per grouped row (32 S = 8 N rows): the grouped kernel
// starts from ReLU(U_j - V_i) and runs the two remaining layers.  sa3: 137.4 -> 104 MMAC per cloud, sa4: 135.8 -> 104; the
// weights a tile streams shrink by the same quarter.  U - V is formed in fp32 (|U|, |V| = O(10): 1e-6 absolute), the xyz
// part of U and V in plain fp32 FMAs; results differ from the concat form at fp32 rounding level only.
//
//   lin_u_kernel : U = X x Wf^T + xyz x Wx^T + b,  V = xyz x Wx^T     (rows = source points of all clouds)
//   sa_u_kernel  : register-chained two-layer stack on ReLU(U_j - V_i) + max over the 32 neighbours (sa3; cf. sa_s_kernel)
#include "bf16_core.hpp"
#include "kernels.hpp"

namespace lsdm {

namespace {

template <int NP>
__device__ __forceinline__ void split8u(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        hi[j] = (__bf16)v[j];
        if constexpr (NP == 2) lo[j] = (__bf16)(v[j] - (float)hi[j]);
    }
}
template <int NP>
__device__ __forceinline__ bf16x8 wfrag_lds_u(const __bf16* base, int NS, int nb, int s, int plane, int lane) {
    return *reinterpret_cast<const bf16x8*>(base + ((size_t)((nb * NS + s) * NP + plane) * 64 + lane) * 8);
}
template <int KS, int NP, bool WA>
__device__ __forceinline__ void mma_panel_u(f32x16& acc, const __bf16* W, const bf16x8 (&xh)[KS], const bf16x8 (&xl)[KS], int lane) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const bf16x8 wh = wfrag_lds_u<NP>(W, KS, 0, s, 0, lane);
        if constexpr (NP == 2) {
            const bf16x8 wlo = wfrag_lds_u<NP>(W, KS, 0, s, 1, lane);
            if constexpr (WA) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl[s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo, xh[s], acc, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl[s], wh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[s], wlo, acc, 0, 0, 0);
            }
        }
        if constexpr (WA) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh[s], acc, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[s], wh, acc, 0, 0, 0);
        if (s % 3 == 2) __builtin_amdgcn_sched_barrier(0);   // bound how many LDS weight fragments are in flight (VGPRs)
    }
}
template <int NP>
__device__ __forceinline__ void relu_split_pair_u(const f32x16& acc, const float* bias32, int h, bf16x8& h0, bf16x8& l0, bf16x8& h1, bf16x8& l1) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(bias32 + 16 * s2 + 4 * h);
        const f32x4 c1 = *reinterpret_cast<const f32x4*>(bias32 + 16 * s2 + 8 + 4 * h);
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = reluf(acc[8 * s2 + j] + c0[j]); v[4 + j] = reluf(acc[8 * s2 + 4 + j] + c1[j]); }
        if (s2 == 0) split8u<NP>(v, h0, l0); else split8u<NP>(v, h1, l1);
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// rows are independent (the source points of all clouds, flattened): X (R,K) fp32, xyz (R,3) -> U (R,N), V (R,N) fp32
template <int K, int N>
__global__ __launch_bounds__(256, 2) void lin_u_kernel(const float* __restrict__ X, const float* __restrict__ xyz, int R, DenseH W,
                                                       const float* __restrict__ wx, float* __restrict__ U, float* __restrict__ V) {
    constexpr int TM = 64, NW = 4, NP = 2, NT = NW * 64;
    constexpr int LDX = ldh(K), PX = TM * LDX;
    __shared__ __attribute__((aligned(16))) __bf16 bufX[NP * PX];
    __shared__ float sxyz[TM * 3];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r0 = blockIdx.x * TM;
    {
        // rows past R repeat the last row; their results are not stored
        constexpr int ITER = TM * (K / 4) / NT;
        static_assert((TM * (K / 4)) % NT == 0, "tile load");
        f32x4 gv[ITER];
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int e = tid + it * NT;
            const int row = e / (K / 4), c4 = e - row * (K / 4);
            const int rr = r0 + row < R ? r0 + row : R - 1;
            gv[it] = load4f(X + (size_t)rr * K + 4 * c4);
        }
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int e = tid + it * NT;
            const int row = e / (K / 4), c4 = e - row * (K / 4);
            store_split4<NP>(bufX + row * LDX + 4 * c4, PX, gv[it][0], gv[it][1], gv[it][2], gv[it][3]);
        }
        for (int e = tid; e < TM * 3; e += NT) {
            const int row = e / 3;
            const int rr = r0 + row < R ? r0 + row : R - 1;
            sxyz[e] = xyz[(size_t)rr * 3 + (e - row * 3)];
        }
    }
    __syncthreads();
    using TS = SplitH<NW, TM, N>;
    const int cb0 = wave % TS::CSTEP, rg = wave / TS::CSTEP;
#pragma unroll 1
    for (int cb = cb0; cb < TS::CB; cb += TS::CSTEP) {
        f32x16 acc[TS::RB];
        acc_zero_h(acc);
        const int col = cb * 32 + (lane & 31);
        const float bias = W.b[col];
        const float wx0 = wx[col * 3 + 0], wx1 = wx[col * 3 + 1], wx2 = wx[col * 3 + 2];
        mma_kloop_h<TS::RB, NP, false, K / 16>(acc, bufX + rg * TS::RB * 32 * LDX, LDX, PX, W.w, N, cb, 0, lane);
#pragma unroll
        for (int rb = 0; rb < TS::RB; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (rg * TS::RB + rb) * 32 + acc_row(r, lane);
                if (r0 + row < R) {
                    const float v = fmaf(wx2, sxyz[row * 3 + 2], fmaf(wx1, sxyz[row * 3 + 1], wx0 * sxyz[row * 3 + 0]));
                    V[(size_t)(r0 + row) * N + col] = v;
                    U[(size_t)(r0 + row) * N + col] = (acc[rb][r] + bias) + v;
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------
// Register-chained stack (cf. encoder_reg.hip::sa_s_kernel) on ReLU(U_j - V_i): layer 2 (C1 -> C2) and layer 3 (C2 -> C3, max
// over the 32 neighbours).  blob: [W2 frags: C2/32 x C1/16 x NP x 512][W3: C3/32 x C2/16 x NP x 512], biases [b2 | b3];
// the k-slot (s, h, j) of layer 2 is channel 16 s + 8 h + j of U (identity order), layer 3 follows the accumulator order.
template <int C1, int C2, int C3, int NW, int NP>
__global__ __launch_bounds__(NW * 64) void sa_u_kernel(const float* __restrict__ U, const float* __restrict__ V,
                                                        const int* __restrict__ cent_idx, const int* __restrict__ ball, int NSRC, int S,
                                                        long long total_centroids, const __bf16* __restrict__ blob_w,
                                                        const float* __restrict__ blob_b, float* __restrict__ out) {
    constexpr int NT = NW * 64;
    constexpr int KS2 = C1 / 16, KS3 = C2 / 16;
    constexpr int NB2 = C2 / 32, NB3 = C3 / 32;
    constexpr int F2 = KS2 * NP, F3 = KS3 * NP;
    constexpr int FMAX = F2 > F3 ? F2 : F3;
    constexpr int W2E = NB2 * F2 * 512;
    constexpr int NPANEL = NB2 + NB3;
    static_assert(NPANEL % 2 == 0, "panel count must be even (double-buffer parity is compile-time)");
    constexpr int PIECES = (FMAX * 64 + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) __bf16 wl[2][FMAX * 512];
    __shared__ __attribute__((aligned(16))) float bl[C2 + C3];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int e = tid; e < C2 + C3; e += NT) bl[e] = blob_b[e];
    const float* b2 = bl;
    const float* b3 = bl + C2;
    const int p = lane & 31, h = lane >> 5;

    auto panel_off = [&](int q) -> int { return q < NB2 ? q * F2 * 512 : W2E + (q - NB2) * F3 * 512; };
    auto panel_nf = [&](int q) -> int { return q < NB2 ? F2 : F3; };
    bf16x8 stage[PIECES];
    auto fetch = [&](int q) {
        const __bf16* src = blob_w + panel_off(q);
        asm volatile("" : "+s"(src));   // keeps the (round-invariant) panel fetches inside the persistent loop
        const int n16 = panel_nf(q) * 64;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int e = tid + i * NT;
            if (e < n16) stage[i] = *reinterpret_cast<const bf16x8*>(src + (size_t)e * 8);
        }
    };
    auto commit = [&](int q) {
        const int n16 = panel_nf(q) * 64;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int e = tid + i * NT;
            if (e < n16) *reinterpret_cast<bf16x8*>(&wl[q & 1][(size_t)e * 8]) = stage[i];
        }
    };
    fetch(0);
    commit(0);
    __syncthreads();

    const long long rounds = (total_centroids + NW - 1) / NW;
    const long long rchunk = (rounds + gridDim.x - 1) / gridDim.x;
    const long long rbeg = (long long)xcd_remap((int)blockIdx.x, (int)gridDim.x) * rchunk;
    const long long rend = rbeg + rchunk < rounds ? rbeg + rchunk : rounds;
    for (long long rd = rbeg; rd < rend; ++rd) {
        long long cent = rd * NW + wave;
        const bool live = cent < total_centroids;
        if (!live) cent = total_centroids - 1;
        const long long c = cent / S;
        const int idx = ball[cent * kNSample + p];
        // layer-2 operand fragments: ReLU(U_j - V_i), this lane's k-slots of step s = channels 16 s + 8 h .. + 7
        const float* urow = U + ((size_t)c * NSRC + idx) * C1 + 8 * h;
        const float* vrow = V + ((size_t)c * NSRC + cent_idx[cent]) * C1 + 8 * h;
        bf16x8 yh[KS2], yl[KS2];
        {
            f32x4 u0[KS2], u1[KS2], v0[KS2], v1[KS2];
#pragma unroll
            for (int s = 0; s < KS2; ++s) {
                u0[s] = load4f(urow + 16 * s); u1[s] = load4f(urow + 16 * s + 4);
                v0[s] = load4f(vrow + 16 * s); v1[s] = load4f(vrow + 16 * s + 4);
            }
#pragma unroll
            for (int s = 0; s < KS2; ++s) {
                const float t[8] = {reluf(u0[s][0] - v0[s][0]), reluf(u0[s][1] - v0[s][1]), reluf(u0[s][2] - v0[s][2]), reluf(u0[s][3] - v0[s][3]),
                                    reluf(u1[s][0] - v1[s][0]), reluf(u1[s][1] - v1[s][1]), reluf(u1[s][2] - v1[s][2]), reluf(u1[s][3] - v1[s][3])};
                split8u<NP>(t, yh[s], yl[s]);
            }
        }
        bf16x8 zh[KS3], zl[KS3];
#define PANEL_BEGIN(Q) { fetch(((Q) + 1) % NPANEL); const __bf16* W = &wl[(Q) & 1][0]; int lz = 0; asm volatile("" : "+v"(lz));
#define PANEL_END(Q) commit(((Q) + 1) % NPANEL); __syncthreads(); }
#pragma unroll
        for (int nb = 0; nb < NB2; ++nb) {
            PANEL_BEGIN(nb)
            f32x16 acc;
            mma_panel_u<KS2, NP, true>(acc, W, yh, yl, lane);
            relu_split_pair_u<NP>(acc, b2 + lz + 32 * nb, h, zh[2 * nb], zl[2 * nb], zh[2 * nb + 1], zl[2 * nb + 1]);
            PANEL_END(nb)
        }
#pragma unroll
        for (int nb = 0; nb < NB3; ++nb) {
            PANEL_BEGIN(NB2 + nb)
            f32x16 acc;
            mma_panel_u<KS3, NP, false>(acc, W, zh, zl, lane);
            float m = acc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[r]);
            m = fmaxf(m, __shfl_xor(m, 32));
            if (live && lane < 32) out[(size_t)cent * C3 + nb * 32 + lane] = reluf(m + b3[nb * 32 + lane]);
            PANEL_END(NB2 + nb)
        }
#undef PANEL_BEGIN
#undef PANEL_END
    }
}

// ---------------------------------------------------------------------------------------------
void launch_lin_u(int level, int R, const float* X, const float* xyz, DenseH W, const float* wx, float* U, float* V, hipStream_t st) {
    const dim3 grid((R + 63) / 64), block(256);
    if (level == 2) hipLaunchKernelGGL((lin_u_kernel<128, 128>), grid, block, 0, st, X, xyz, R, W, wx, U, V);
    else hipLaunchKernelGGL((lin_u_kernel<256, 256>), grid, block, 0, st, X, xyz, R, W, wx, U, V);
}

void launch_sa_u(int level, int nc, const float* U, const float* V, const int* cent_idx, const int* ball, const __bf16* blob_w,
                 const float* blob_b, float* out, hipStream_t st) {
    if (level == 2) {
        const long long tot = (long long)nc * 64;
        hipLaunchKernelGGL((sa_u_kernel<128, 128, 256, 4, 2>), dim3(512), dim3(256), 0, st, U, V, cent_idx, ball, 256, 64, tot, blob_w, blob_b, out);
    }
}

}  // namespace lsdm
