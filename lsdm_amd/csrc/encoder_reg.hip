// Register-chained set-abstraction kernel for the two NARROW levels (sa1, sa2) on the split-bf16 MFMA core, gfx950.
//
// sa_h_kernel (encoder_bf.hip) keeps the activations of a point tile in LDS and streams weights from L2; for the
// narrow levels that spends most of its time in LDS round trips, epilogue VALU work and four workgroup barriers per tile.
// Here the roles are swapped:
//   * all weights of the level (14 KiB for sa1, 68 KiB for sa2 as bf16 hi+lo fragments) are copied into LDS ONCE per
//     persistent workgroup and stay there;
//   * each WAVE owns one centroid (its 32 neighbours = one 32-point MFMA block) and carries the activations of the
//     whole 3-layer stack in REGISTERS: in the transposed product D[channel][point] a lane owns, for its point, the
//     channels {8q + 4h + j}; accumulator registers 8*s2 .. 8*s2+7 of a 32-channel block are therefore exactly the
//     8 k-slots that the same lane must supply as the B-operand fragment of the next layer's k-step 2*block + s2 --
//     bias + ReLU + bf16 split + pack turn accumulators into operands without any data movement.  The host packs the
//     next layer's weight fragments with the matching k-slot order (pack_weight_r in api.hip).
//   * the last layer uses the direct orientation D[point][channel] (the same registers serve as the A operand), so the
//     max over the 32 neighbours is an in-lane max over 16 accumulators + one cross-half shuffle.
// No workgroup barrier exists after the weight copy; waves stream over centroids independently.
// Restates pointnet2_utils.py:107-135,173-199 (sample_and_group + 3 x (conv1x1 + BN(eval, folded) + ReLU) + max).
#include <cstdlib>

#include "bf16_core.hpp"
#include "kernels.hpp"

namespace lsdm {

namespace {

template <int NP>
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) { split8_h<NP>(v, hi, lo); }

// weight fragment (nb, s, plane) of a layer whose fragments start at `base` (LDS): 1 KiB each, lane-linear
template <int NP>
__device__ __forceinline__ bf16x8 wfrag_lds(const __bf16* base, int NS, int nb, int s, int plane, int lane) {
    return *reinterpret_cast<const bf16x8*>(base + ((size_t)((nb * NS + s) * NP + plane) * 64 + lane) * 8);
}

// one transposed layer, registers -> registers:  Y = ReLU(W x X + b)
//   X: KS k-steps of B-operand fragments (hi, lo); Y: N/16 k-steps for the next layer
template <int KS, int N, int NP>
__device__ __forceinline__ void layer_reg(const bf16x8 (&xh)[KS], const bf16x8 (&xl)[KS], const __bf16* W, const float* bias,
                                          bf16x8 (&yh)[N / 16], bf16x8 (&yl)[N / 16], int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int nb = 0; nb < N / 32; ++nb) {
        // accumulators 8*s2 .. 8*s2+7 = channels 32*nb + 16*s2 + {0..3, 8..11} + 4h  ->  next layer's k-step 2*nb + s2.
        // They START from the bias (the LDS read lands in the accumulator registers: no zeroing, no add in the epilogue).
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 bq = *reinterpret_cast<const f32x4*>(bias + 32 * nb + 8 * q + 4 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[4 * q + j] = bq[j];
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bf16x8 wh = wfrag_lds<NP>(W, KS, nb, s, 0, lane);
            if constexpr (NP == 2) {
                const bf16x8 wl = wfrag_lds<NP>(W, KS, nb, s, 1, lane);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl[s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh[s], acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh[s], acc, 0, 0, 0);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = reluf(acc[8 * s2 + j]);
            split8<NP>(v, yh[2 * nb + s2], yl[2 * nb + s2]);
        }
        __builtin_amdgcn_sched_barrier(0);   // one channel block at a time: keeps the unrolled LDS weight reads from piling up in VGPRs
    }
}

// acc = W(panel: KS k-steps of one channel block, LDS) (x) X(registers); WA: weights are the MFMA A operand
template <int KS, int NP, bool WA>
__device__ __forceinline__ void mma_panel(f32x16& acc, const __bf16* W, const bf16x8 (&xh)[KS], const bf16x8 (&xl)[KS], int lane) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const bf16x8 wh = wfrag_lds<NP>(W, KS, 0, s, 0, lane);
        if constexpr (NP == 2) {
            const bf16x8 wlo = wfrag_lds<NP>(W, KS, 0, s, 1, lane);
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

// bias + ReLU + split of one transposed 32-channel block into the next layer's two k-step fragments
// (bias32 = bias of the block's first channel; accumulators 8*s2 .. 8*s2+7 <-> channels 16*s2 + {0..3, 8..11} + 4h)
template <int NP>
__device__ __forceinline__ void relu_split_pair(const f32x16& acc, const float* bias32, int h, bf16x8& h0, bf16x8& l0, bf16x8& h1, bf16x8& l1) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(bias32 + 16 * s2 + 4 * h);
        const f32x4 c1 = *reinterpret_cast<const f32x4*>(bias32 + 16 * s2 + 8 + 4 * h);
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = reluf(acc[8 * s2 + j] + c0[j]); v[4 + j] = reluf(acc[8 * s2 + 4 + j] + c1[j]); }
        if (s2 == 0) split8<NP>(v, h0, l0); else split8<NP>(v, h1, l1);
    }
}

}  // namespace

// blob layout (built by api.hip::build_sa_reg_blob), all offsets in bf16 elements / floats:
//   [W1 frags: C1/32 x KS1 x NP x 512][W2: C2/32 x C1/16 x NP x 512][W3: C3/32 x C2/16 x NP x 512] then fp32 [b1|b2|b3]
// PAIR (level 1 only, D == 3): the launch walks an item list of launch_sa_pair_plan (geometry.hip) -- two ball-query groups whose distinct
// neighbours fit into 32 rows together in one block (encoder_u.hip::sa_p_kernel): `ball` = the list's row table (32 source rows per item,
// numbered over the launch), `pmeta` {source row of centroid A, of centroid B, output row of A | k << 24, of B}, *pcount its length.  A
// centroid IS one of the source points, so its coordinates are read from src_xyz (new_xyz holds the same floats in centroid order).
template <int D, int C1, int C2, int C3, int NW, int NP, bool PAIR = false>
__global__ __launch_bounds__(NW * 64) void sa_r_kernel(const float* __restrict__ src_xyz,
                                                        const typename FeatT<NP>::type* __restrict__ src_feat,
                                                        const float* __restrict__ new_xyz, const int* __restrict__ ball, int NSRC,
                                                        int S, long long total_centroids, const __bf16* __restrict__ blob_w,
                                                        const float* __restrict__ blob_b, typename FeatT<NP>::type* __restrict__ out,
                                                        const int4* __restrict__ pmeta, const int* __restrict__ pcount) {
    static_assert(!PAIR || D == 3, "item lists: level 1");
    constexpr int KS1 = (D == 3) ? 1 : (D + 16) / 16;          // features (D) + one step holding [dx dy dz 0...]
    constexpr int KS2 = C1 / 16, KS3 = C2 / 16;
    constexpr int W1E = (C1 / 32) * KS1 * NP * 512, W2E = (C2 / 32) * KS2 * NP * 512, W3E = (C3 / 32) * KS3 * NP * 512;
    constexpr int NBIAS = C1 + C2 + C3;
    __shared__ __attribute__((aligned(16))) __bf16 wl[W1E + W2E + W3E];
    __shared__ __attribute__((aligned(16))) float bl[NBIAS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // ---- one-time copy of the level's weights into LDS ---------------------------------------------------------------
    for (int e = tid; e < (W1E + W2E + W3E) / 8; e += NW * 64)
        *reinterpret_cast<bf16x8*>(wl + (size_t)e * 8) = *reinterpret_cast<const bf16x8*>(blob_w + (size_t)e * 8);
    for (int e = tid; e < NBIAS; e += NW * 64) bl[e] = blob_b[e];
    __syncthreads();
    const __bf16* W1 = wl;
    const __bf16* W2 = wl + W1E;
    const __bf16* W3 = wl + W1E + W2E;
    const float* b1 = bl;
    const float* b2 = bl + C1;
    const float* b3 = bl + C1 + C2;
    const int p = lane & 31, h = lane >> 5;

    // each work-group owns one contiguous chunk of centroids, and the chunks of the work-groups of one XCD are adjacent
    // (common.hpp::xcd_remap): the rows a chunk gathers stay in that XCD's L2
    const long long total = PAIR ? (long long)*pcount : total_centroids;
    const long long chunk = ((total + gridDim.x - 1) / gridDim.x + NW - 1) / NW * NW;
    const long long cbeg = (long long)xcd_remap((int)blockIdx.x, (int)gridDim.x) * chunk;
    const long long cend = cbeg + chunk < total ? cbeg + chunk : total;
    for (long long cent = cbeg + wave; cent < cend; cent += NW) {
        const long long c = cent / S;                                  // cloud
        const int idx = ball[cent * kNSample + p];
        int4 mt = {0, 0, 0, 0};
        if constexpr (PAIR) mt = pmeta[cent];
        const int ka = PAIR ? mt.z >> 24 : 8;                          // A's row quartets
        const bool paired = ka < 8;
        const bool own[4] = {h < ka, 2 + h < ka, 4 + h < ka, 6 + h < ka};
        const size_t orow = PAIR ? (size_t)(h ? mt.w : mt.z & 0xffffff) : (size_t)cent;
        const float* xyz = src_xyz + (PAIR ? (size_t)idx : (size_t)c * NSRC + idx) * 3;
        const float* cen = PAIR ? src_xyz + (size_t)(p < 4 * ka ? mt.x : mt.y) * 3 : new_xyz + (size_t)cent * 3;
        const float dx = xyz[0] - cen[0], dy = xyz[1] - cen[1], dz = xyz[2] - cen[2];
        // ---- layer-1 operand fragments straight from global memory (k-step s, this lane's slots = k 16s + 8h .. +7) -------
        bf16x8 xh[KS1], xl[KS1];
        if constexpr (D == 3) {
            float v[8] = {xyz[0], xyz[1], xyz[2], dx, dy, dz, 0.0f, 0.0f};
            if (h == 1) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = 0.0f;
            }
            split8<NP>(v, xh[0], xl[0]);
        } else {
            const typename FeatT<NP>::type* row = src_feat + ((size_t)c * NSRC + idx) * D + 8 * h;
            f32x4 r0[D / 16], r1[D / 16];
#pragma unroll
            for (int s = 0; s < D / 16; ++s) {
                r0[s] = load4f(row + 16 * s);
                r1[s] = load4f(row + 16 * s + 4);
            }
#pragma unroll
            for (int s = 0; s < D / 16; ++s) {
                const float v[8] = {r0[s][0], r0[s][1], r0[s][2], r0[s][3], r1[s][0], r1[s][1], r1[s][2], r1[s][3]};
                split8<NP>(v, xh[s], xl[s]);
            }
            float t[8] = {dx, dy, dz, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            if (h == 1) { t[0] = 0.0f; t[1] = 0.0f; t[2] = 0.0f; }
            split8<NP>(t, xh[D / 16], xl[D / 16]);
        }
        bf16x8 yh[KS2], yl[KS2];
        layer_reg<KS1, C1, NP>(xh, xl, W1, b1, yh, yl, lane);
        bf16x8 zh[KS3], zl[KS3];
        layer_reg<KS2, C2, NP>(yh, yl, W2, b2, zh, zl, lane);
        // ---- layer 3, direct orientation D[point][channel], max over the 32 neighbours ---------------------------------------
#pragma unroll
        for (int nb = 0; nb < C3 / 32; ++nb) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
            for (int s = 0; s < KS3; ++s) {
                const bf16x8 wh = wfrag_lds<NP>(W3, KS3, nb, s, 0, lane);
                if constexpr (NP == 2) {
                    const bf16x8 wlo = wfrag_lds<NP>(W3, KS3, nb, s, 1, lane);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zl[s], wh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zh[s], wlo, acc, 0, 0, 0);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zh[s], wh, acc, 0, 0, 0);
            }
            // relu(x + b) is monotone in x, so pooling before bias/ReLU is exact
            if constexpr (PAIR) {
                const float m = max_split(acc, own, paired);            // half-wave 0 stores A's row, half-wave 1 B's (unpaired: the same row twice)
                store1(out + orow * C3 + nb * 32 + p, reluf(m + b3[nb * 32 + p]));
            } else {
                float m = acc[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[r]);
                m = fmaxf(m, __shfl_xor(m, 32));
                if (lane < 32) store1(out + (size_t)cent * C3 + nb * 32 + lane, reluf(m + b3[nb * 32 + lane]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Same register-chained stack for a level whose weights do NOT fit in LDS (sa3: 266 KB as hi+lo fragments): the eight
// waves of a workgroup (one centroid each) walk the (layer, channel-block) weight panels in lockstep; a panel (<= 18 KiB)
// is staged global -> registers -> LDS while the previous one feeds the MFMAs (double buffer, one barrier per panel).
// Every panel is fetched from L2 once per 256 points instead of once per 64 (sa_h_kernel), and no activation ever
// touches LDS.
// ---------------------------------------------------------------------------------------------
template <int D, int C1, int C2, int C3, int NW, int NP>
__global__ __launch_bounds__(NW * 64) void sa_s_kernel(const float* __restrict__ src_xyz,
                                                        const typename FeatT<NP>::type* __restrict__ src_feat,
                                                        const float* __restrict__ new_xyz, const int* __restrict__ ball, int NSRC,
                                                        int S, long long total_centroids, const __bf16* __restrict__ blob_w,
                                                        const float* __restrict__ blob_b, typename FeatT<NP>::type* __restrict__ out) {
    constexpr int NT = NW * 64;
    constexpr int KS1 = (D + 16) / 16, KS2 = C1 / 16, KS3 = C2 / 16;
    constexpr int NB1 = C1 / 32, NB2 = C2 / 32, NB3 = C3 / 32;
    constexpr int F1 = KS1 * NP, F2 = KS2 * NP, F3 = KS3 * NP;            // fragments (1 KiB each) per panel
    constexpr int FMAX = F1 > F2 ? (F1 > F3 ? F1 : F3) : (F2 > F3 ? F2 : F3);
    constexpr int W1E = NB1 * F1 * 512, W2E = NB2 * F2 * 512;
    constexpr int NPANEL = NB1 + NB2 + NB3;
    static_assert(NPANEL % 2 == 0, "panel count must be even (double-buffer parity is compile-time)");
    constexpr int PIECES = (FMAX * 64 + NT - 1) / NT;                       // 16-byte pieces per thread per panel
    __shared__ __attribute__((aligned(16))) __bf16 wl[2][FMAX * 512];
    __shared__ __attribute__((aligned(16))) float bl[C1 + C2 + C3];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int e = tid; e < C1 + C2 + C3; e += NT) bl[e] = blob_b[e];
    const float* b1 = bl;
    const float* b2 = bl + C1;
    const float* b3 = bl + C1 + C2;
    const int p = lane & 31, h = lane >> 5;

    // panel q (compile-time) -> offset into the blob and fragment count
    auto panel_off = [&](int q) -> int { return q < NB1 ? q * F1 * 512 : (q < NB1 + NB2 ? W1E + (q - NB1) * F2 * 512 : W1E + W2E + (q - NB1 - NB2) * F3 * 512); };
    auto panel_nf = [&](int q) -> int { return q < NB1 ? F1 : (q < NB1 + NB2 ? F2 : F3); };
    bf16x8 stage[PIECES];
    auto fetch = [&](int q) {       // global -> registers (asynchronous until first use)
        const __bf16* src = blob_w + panel_off(q);
        // the panels are the same every round: without this the compiler hoists all NPANEL fetches out of the persistent
        // loop and parks them in ~200 VGPRs
        src = opaque_ptr(src);   // keeps the (round-invariant) panel fetches inside the persistent loop, as global loads
        const int n16 = panel_nf(q) * 64;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int e = tid + i * NT;
            if (e < n16) stage[i] = gload<bf16x8>(src + (size_t)e * 8);
        }
    };
    auto commit = [&](int q) {      // registers -> LDS buffer q & 1
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
        if (!live) cent = total_centroids - 1;                              // keep the barriers convergent; results discarded
        const long long c = cent / S;
        const int idx = ball[cent * kNSample + p];
        const float* xyz = src_xyz + ((size_t)c * NSRC + idx) * 3;
        const float* cen = new_xyz + (size_t)cent * 3;
        const float dx = xyz[0] - cen[0], dy = xyz[1] - cen[1], dz = xyz[2] - cen[2];
        bf16x8 xh[KS1], xl[KS1];
        {
            const typename FeatT<NP>::type* row = src_feat + ((size_t)c * NSRC + idx) * D + 8 * h;
            f32x4 r0[D / 16], r1[D / 16];
#pragma unroll
            for (int s = 0; s < D / 16; ++s) {
                r0[s] = load4f(row + 16 * s);
                r1[s] = load4f(row + 16 * s + 4);
            }
#pragma unroll
            for (int s = 0; s < D / 16; ++s) {
                const float v[8] = {r0[s][0], r0[s][1], r0[s][2], r0[s][3], r1[s][0], r1[s][1], r1[s][2], r1[s][3]};
                split8<NP>(v, xh[s], xl[s]);
            }
            float t[8] = {dx, dy, dz, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            if (h == 1) { t[0] = 0.0f; t[1] = 0.0f; t[2] = 0.0f; }
            split8<NP>(t, xh[D / 16], xl[D / 16]);
        }
        bf16x8 yh[KS2], yl[KS2], zh[KS3], zl[KS3];
        // one panel step: request the next panel, run this panel's MFMAs + epilogue, publish the next panel
// lz: an opaque zero defined per panel -- the LDS reads of biases / conv2 weights are the same every round, and without
// it the compiler hoists all of them to the top of the persistent loop and parks them in ~200 VGPRs
#define PANEL_BEGIN(Q) { fetch(((Q) + 1) % NPANEL); const __bf16* W = &wl[(Q) & 1][0]; int lz = 0; asm volatile("" : "+v"(lz));
#define PANEL_END(Q) commit(((Q) + 1) % NPANEL); __syncthreads(); }
#pragma unroll
        for (int nb = 0; nb < NB1; ++nb) {
            PANEL_BEGIN(nb)
            f32x16 acc;
            mma_panel<KS1, NP, true>(acc, W, xh, xl, lane);
            relu_split_pair<NP>(acc, b1 + lz + 32 * nb, h, yh[2 * nb], yl[2 * nb], yh[2 * nb + 1], yl[2 * nb + 1]);
            PANEL_END(nb)
        }
#pragma unroll
        for (int nb = 0; nb < NB2; ++nb) {
            PANEL_BEGIN(NB1 + nb)
            f32x16 acc;
            mma_panel<KS2, NP, true>(acc, W, yh, yl, lane);
            relu_split_pair<NP>(acc, b2 + lz + 32 * nb, h, zh[2 * nb], zl[2 * nb], zh[2 * nb + 1], zl[2 * nb + 1]);
            PANEL_END(NB1 + nb)
        }
#pragma unroll
        for (int nb = 0; nb < NB3; ++nb) {
            PANEL_BEGIN(NB1 + NB2 + nb)
            f32x16 acc;
            mma_panel<KS3, NP, false>(acc, W, zh, zl, lane);
            float m = acc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[r]);
            m = fmaxf(m, __shfl_xor(m, 32));
            if (live && lane < 32) store1(out + (size_t)cent * C3 + nb * 32 + lane, reluf(m + b3[nb * 32 + lane]));
            PANEL_END(NB1 + NB2 + nb)
        }
#undef PANEL_BEGIN
#undef PANEL_END
    }
}

void launch_sa_r(int np, int level, int nc, const float* src_xyz, const void* src_feat, const float* new_xyz, const int* ball,
                 const __bf16* blob_w, const float* blob_b, void* out, hipStream_t st) {
    // persistent grids: sa1 (15 KB of LDS, 57 VGPRs) 4 workgroups of 8 waves per CU, sa2 (69 KB, 123 VGPRs) 2 per CU
    (void)np;      // (the register-chained full stacks exist for two planes; plain bf16 reaches sa1 only, with its two-plane blob: api.hip)
    const float* sf = static_cast<const float*>(src_feat);
    float* o = static_cast<float*>(out);
    if (level == 0)
        hipLaunchKernelGGL((sa_r_kernel<3, 32, 32, 64, 8, 2>), dim3(1024), dim3(512), 0, st, src_xyz, sf, new_xyz, ball, 1024, 1024, (long long)nc * 1024, blob_w, blob_b, o,
                           nullptr, nullptr);
    else if (level == 2)
        hipLaunchKernelGGL((sa_s_kernel<128, 128, 128, 256, 4, 2>), dim3(512), dim3(256), 0, st, src_xyz, sf, new_xyz, ball, 256, 64, (long long)nc * 64, blob_w, blob_b, o);
    else
        hipLaunchKernelGGL((sa_r_kernel<64, 64, 64, 128, 8, 2>), dim3(512), dim3(512), 0, st, src_xyz, sf, new_xyz, ball, 1024, 256, (long long)nc * 256, blob_w, blob_b, o,
                           nullptr, nullptr);
}

// sa1 (bf16x3) over the item list of launch_sa_pair_plan: rows = 32 source rows per item, meta = {centroid rows, output rows}, *count items
void launch_sa_r_items(const float* src_xyz, const int* rows, const int4* meta, const int* count, const __bf16* blob_w, const float* blob_b, float* out,
                       hipStream_t st) {
    // (70 VGPRs against the plain form's 64: three work-groups of 8 waves per CU instead of four)
    hipLaunchKernelGGL((sa_r_kernel<3, 32, 32, 64, 8, 2, true>), dim3(768), dim3(512), 0, st, src_xyz, static_cast<const float*>(nullptr), src_xyz, rows, 1024, 1024, 0LL,
                       blob_w, blob_b, out, meta, count);
}

}  // namespace lsdm
