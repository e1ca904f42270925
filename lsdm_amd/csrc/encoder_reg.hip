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

// Measurement hooks of tools/ubench/chain_bench.hip (never defined in the library build): drop one cost at a time from fp_s_kernel
//   LSDM_X_NOLDS  weight fragments come from a register instead of LDS      LSDM_X_NOMFMA  no matrix instructions
//   LSDM_X_NOEPI  no ReLU + hi/lo split between the layers                  LSDM_X_NOBAR   no barrier / panel wait per layer
#ifdef LSDM_X_NOLDS
#define XWFRAG(W, KS, NBI, s, pl, lane) xfrag_const
#else
#define XWFRAG(W, KS, NBI, s, pl, lane) wfrag_lds<NP>(W, KS, NBI, s, pl, lane)
#endif
#ifdef LSDM_X_NOMFMA
#define XMFMA(a, b, c) xfake_mfma(a, b, c)
__device__ __forceinline__ f32x16 xfake_mfma(const bf16x8& a, const bf16x8& b, f32x16 c) {
    c[0] += (float)a[0] * (float)b[0];
    return c;
}
#else
#define XMFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif
#ifdef LSDM_X_NOBAR
#define XBARRIER
#else
#define XBARRIER lds_dma_wait(); __syncthreads();
#endif
#ifdef LSDM_X_NOEPI
#define XSPLIT(v, oh, ol) { oh[0] = (__bf16)v[0]; ol[0] = (__bf16)v[7]; }
#else
#define XSPLIT(v, oh, ol) split8<NP>(v, oh, ol)
#endif

// ---------------------------------------------------------------------------------------------
// fp1 + head as a register-chained stack (bf16x3): interpolation of the three nearest level-1 rows (pointnet2_utils.py:286-297)
// -> 3 x (conv 128 -> 128 + BN + ReLU) -> conv1 + bn1 + ReLU -> conv2 (128 -> 3)   (pointnet2.py:72-80).
// fp_h_kernel runs the five layers on a 64-point LDS tile: every layer is an LDS round trip with a work-group barrier and
// every work-group streams the 320 KB of weight fragments for 64 points (MFMA pipe 25 % busy).  Here a wave owns 32 points and
// carries their activations through the whole stack in registers (layer_reg's accumulator -> operand identity).  The weights
// of one LAYER (64 KB as hi + lo fragments) are a panel: two panels in LDS, the next layer's is copied global -> LDS by the
// LDS-DMA path (no staging registers) while the waves work on the current one, ONE barrier per layer -- inside a layer the
// waves drift apart, so one wave's epilogue / gather overlaps another's MFMAs (measured with tools/ubench/chain_bench.hip:
// sixteen 16 KB panels with a barrier each cost 75 us of 272; the barriers keep all waves of a CU in the same phase and expose
// the gather latency of every round).  conv2's padded block stays resident.  Weight traffic per point: 1 KB instead of 5 KB.
// (Round 3: requesting the first half of the NEXT round's gather -- 24 x 16 bytes per lane -- in front of conv2 and the second half
// next to the first half's interpolation arithmetic was measured slower, 349 vs 290 us: the 96 staging registers do not fit next
// to the two operand sets at two waves per SIMD and part of them goes through scratch.  The two half-work-groups OUT OF PHASE -- waves
// 0-3 gathering while waves 4-7 run the layers and vice versa, so that each SIMD's matrix pipe always has exactly one wave feeding it and
// the other wave's loads and interpolation arithmetic issue in its gaps; panels streamed once per phase -- was measured slower as well,
// 294 vs 271 us, 318 with the round's 48 row requests issued at once: a SIMD's pipe is fed better by two waves that both issue MFMAs and
// drift apart inside a layer than by one wave with dependent accumulator chains and block-placed epilogues.)
// (The gather as four batches of two k-steps with two batches in flight -- the L2 round trips of consecutive batches overlapping --
// measured the same as two batches of four: 276 vs 275 us.)
// blob: four layers [nb][s][plane][lane][8] (layer 1 in channel order, later layers in accumulator order) + conv2 zero-padded
// to 32 channels; biases [4 x 128 | 32].
template <int NW, int NP, int PH = 1>
__global__ __launch_bounds__(NW * 64, NW / 4) void fp_s_kernel(const float* __restrict__ points2, const int* __restrict__ nn_idx,
                                                        const float* __restrict__ nn_w, int N, int S, long long total_points,
                                                        const __bf16* __restrict__ blob_w, const float* __restrict__ blob_b,
                                                        float* __restrict__ out, int dbg, const int* __restrict__ order, int order_mod) {
    constexpr int C = 128, NT = NW * 64, KS = C / 16, NB = C / 32, F = KS * NP;
    constexpr int NL = 4, FL = NB * F;                       // fragments (1 KiB) per layer
    constexpr int FPN = FL / PH, NBP = NB / PH;              // PH panels per layer: fragments / channel blocks per panel
    static_assert(NB % PH == 0 && (NL * PH) % 2 == 0, "panel split");
    constexpr int GK = NW > 8 ? 2 : 4;                       // k-steps per batch of gather requests (staging registers)
    __shared__ __attribute__((aligned(16))) __bf16 wl[2][FPN * 512];
    // LSDM_FP1_ROWS = 1 (A/B, round 4; default 0): the gather fetches WHOLE rows -- one wave instruction = two 512-byte rows, lanes 0-31 one
    // point's row, lanes 32-63 the next point's: 8 full lines per instruction -- interpolates and splits them in that layout and turns them
    // into the operand layout (lane (p, h) <- channels 16 s + 8 h .. of ITS point p) through a 2 KB LDS slot of the wave, four points at a
    // time; conv2's block moves out of LDS to make room (read through the L1, as with PH > 1).  The direct form has every lane fetch its own
    // operand slices: an instruction touches 32 rows and 32 bytes of each of their lines, every line is requested by four instructions.
    // Bit-identical results; measured the SAME time (271.5 / 273.1 against 270.5 / 270.2 us, same box): the repeated requests of a line are
    // L1 hits and cost nothing that shows -- what bounds the stage's gather is the round trip of its first touch, not the number of touches.
#ifndef LSDM_FP1_ROWS
#define LSDM_FP1_ROWS 0
#endif
    constexpr bool ROWS = LSDM_FP1_ROWS != 0 && NP == 2;
    constexpr bool W5LDS = PH == 1 && !ROWS;
    constexpr int TRP = 4 * C;                                  // plane stride of a wave's transposition slot [plane][4 points][C]
    __shared__ __attribute__((aligned(16))) __bf16 tr[ROWS ? NW * NP * TRP : 8];
    // conv2's padded block: resident in LDS, or (PH > 1: two work-groups share a CU's LDS; ROWS) read from global memory through the L1
    __shared__ __attribute__((aligned(16))) __bf16 w5[W5LDS ? F * 512 : 8];
    __shared__ __attribute__((aligned(16))) float bl[NL * C + 32];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    if constexpr (W5LDS)
        for (int e = tid; e < F * 64; e += NT) *reinterpret_cast<bf16x8*>(w5 + (size_t)e * 8) = gload<bf16x8>(blob_w + (size_t)NL * FL * 512 + (size_t)e * 8);
    const __bf16* w5g = blob_w + (size_t)NL * FL * 512;
    for (int e = tid; e < NL * C + 32; e += NT) bl[e] = blob_b[e];
    const int p = lane & 31, h = lane >> 5;
#ifdef LSDM_X_NOLDS
    bf16x8 xfrag_const = gload<bf16x8>(blob_w + lane * 8);
    asm volatile("" : "+v"(xfrag_const));
#endif

    // panel q (PH per layer, channel-block major) -> LDS buffer q & 1: each wave copies FPN / NW fragments, a fragment = one 1 KiB
    // wave-wide LDS-DMA
    auto fetch = [&](int q) {
        const __bf16* src = opaque_ptr(blob_w + (size_t)q * FPN * 512);   // (round-invariant: keep the copy inside the loop)
#pragma unroll
        for (int i = 0; i < (FPN + NW - 1) / NW; ++i) {
            const int fr = wave + i * NW;
            if (FPN % NW == 0 || fr < FPN)
                lds_dma16(src + (size_t)fr * 512 + lane * 8, __builtin_amdgcn_readfirstlane(lds_addr(&wl[q & 1][(size_t)fr * 512])));
        }
    };
    // the same copy piecewise, issued from inside the channel blocks of the panel before it (LSDM_FPS_SPREAD): the copies of a wave then
    // issue in the shadow of its own matrix instructions instead of as one block in front of them (sa_w_kernel, encoder_u.hip)
#ifndef LSDM_FPS_SPREAD
#define LSDM_FPS_SPREAD 1
#endif
    constexpr bool SPREAD = LSDM_FPS_SPREAD != 0 && FPN % (NW * NBP) == 0;
    constexpr int PPB = FPN / NW / NBP;                      // pieces per channel block and wave
    auto fetch_pieces = [&](int q, int blk) {
        const __bf16* src = opaque_ptr(blob_w + (size_t)q * FPN * 512);
#pragma unroll
        for (int i = blk * PPB; i < (blk + 1) * PPB; ++i) {
            const int fr = wave + i * NW;
            lds_dma16(src + (size_t)fr * 512 + lane * 8, __builtin_amdgcn_readfirstlane(lds_addr(&wl[q & 1][(size_t)fr * 512])));
        }
    };
    fetch(0);
    lds_dma_wait();
    __syncthreads();

    const long long rounds = total_points / (32 * NW);
    const XcdRounds sched = xcd_rounds((int)rounds);       // the work-groups of an XCD walk the rounds together (common.hpp)
    const long long rbeg = sched.first, rend = sched.end, rstep = sched.stride;
    // neighbour indices / weights of a round are requested during the previous one
    // order (optional; round 5): the points of a cloud are walked along its Z curve (encoder_d.hip::morton_order_kernel, once per call on the
    // source clouds) instead of in index order, so that the 32 points of a wave are neighbours in space and share their level-1 neighbours:
    // ~49 distinct rows per wave and round instead of ~93 of the 96 requested -- half the first-touch round trips of the gather.  Position k of
    // cloud c is point order[(c % order_mod) * N + k]; its neighbour list and its output row are addressed by the point.  Rows of the stack
    // are independent: results do not depend on the walking order.  The point number of a round is requested two rounds ahead, its neighbour
    // list one round ahead.
    int ni[3] = {0, 0, 0};
    float nw[3] = {0.0f, 0.0f, 0.0f};
    int pn_next = 0, pn_cur = 0;            // point (within its cloud) of this lane in the next round / in the round whose list is in ni, nw
    auto point_of = [&](long long rd) -> int {
        const long long k = rd * (32 * NW) + wave * 32 + p;
        const long long c = k / N;
        return order ? order[(size_t)(order_mod > 0 ? c % order_mod : c) * N + (k - c * N)] : (int)(k - c * N);
    };
    auto request = [&](long long rd, int pn) {
        const long long c = (rd * (32 * NW) + wave * 32 + p) / N;
        const size_t o = ((size_t)c * N + pn) * 3;
#pragma unroll
        for (int j = 0; j < 3; ++j) { ni[j] = nn_idx[o + j]; nw[j] = nn_w[o + j]; }
        if (dbg == 1) { ni[0] = ni[1] = ni[2] = (int)((o / 3) % S); }          // experiment: sequential rows
        if (dbg == 2) { ni[0] = ni[1] = ni[2] = (int)(wave * 4); }               // experiment: one row per wave
    };
    if (rbeg < rend) {
        pn_cur = point_of(rbeg);
        request(rbeg, pn_cur);
        if (rbeg + rstep < rend) pn_next = point_of(rbeg + rstep);
    }
    for (long long rd = rbeg; rd < rend; rd += rstep) {
        const long long pt = rd * (32 * NW) + wave * 32 + p;    // this lane's walking position (both halves of the wave: same point)
        const long long c = pt / N;
        const int pn_this = pn_cur;                             // the point at that position
        [[maybe_unused]] const float* r0 = points2 + ((size_t)c * S + ni[0]) * C + 8 * h;
        const float* r1 = points2 + ((size_t)c * S + ni[1]) * C + 8 * h;
        const float* r2 = points2 + ((size_t)c * S + ni[2]) * C + 8 * h;
        const float w0 = nw[0], w1 = nw[1], w2 = nw[2];
        bf16x8 yh[KS], yl[KS], zh[KS], zl[KS];
        if constexpr (ROWS) {
            // ---- row-wise gather: chunk ch = points 4 ch .. 4 ch + 3 of the wave as two instructions per neighbour (pair g: lanes 0-31 <- point
            // 4 ch + 2 g, lanes 32-63 <- the next); four chunks (24 loads, 96 registers) in flight
            __bf16* trw = tr + (size_t)wave * NP * TRP;
            const float* pb = points2 + (size_t)c * S * C + 4 * p;
            auto pick_i = [&](int v, int la) { const int a = __builtin_amdgcn_readlane(v, la), b = __builtin_amdgcn_readlane(v, la + 1); return h ? b : a; };
            auto pick_f = [&](float v, int la) { return __builtin_bit_cast(float, pick_i(__builtin_bit_cast(int, v), la)); };
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                f32x4 a[4][2], b[4][2], d[4][2];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const int la = 4 * (4 * half + q) + 2 * g;          // lane (= point of the wave) that holds pair g's first point
                        a[q][g] = load4f(pb + (size_t)pick_i(ni[0], la) * C);
                        b[q][g] = load4f(pb + (size_t)pick_i(ni[1], la) * C);
                        d[q][g] = load4f(pb + (size_t)pick_i(ni[2], la) * C);
                    }
                // (the weights of this round are still needed below: the next round's indices are requested after the last chunk)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ch = 4 * half + q;
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const int la = 4 * ch + 2 * g;
                        const float u0 = pick_f(nw[0], la), u1 = pick_f(nw[1], la), u2 = pick_f(nw[2], la);
                        float v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = interp3(a[q][g][j], b[q][g][j], d[q][g][j], u0, u1, u2);
                        store_split4<NP>(trw + (2 * g + h) * C + 4 * p, TRP, v[0], v[1], v[2], v[3]);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                    if ((p >> 2) == ch) {
#pragma unroll
                        for (int s = 0; s < KS; ++s) {
                            yh[s] = *reinterpret_cast<const bf16x8*>(trw + (p & 3) * C + 16 * s + 8 * h);
                            yl[s] = *reinterpret_cast<const bf16x8*>(trw + (p & 3) * C + 16 * s + 8 * h + TRP);
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                }
            }
            if (rd + rstep < rend) { request(rd + rstep, pn_next); pn_cur = pn_next; if (rd + 2 * rstep < rend) pn_next = point_of(rd + 2 * rstep); }
        } else {
        // ---- interpolated operand, k-step s = channels 16 s + 8 h .. + 7 of this lane's point (four k-steps per batch of loads)
#pragma unroll
        for (int s0 = 0; s0 < KS; s0 += GK) {
            f32x4 a[GK][2], b[GK][2], d[GK][2];
            // (row-major request order: the four 16-byte pieces a lane pair takes from one 128-byte line are consecutive
            // instructions, so the line is still in the L1 when its last piece is asked for)
#pragma unroll
            for (int s = 0; s < GK; ++s)
#pragma unroll
                for (int q = 0; q < 2; ++q) a[s][q] = load4f(r0 + 16 * (s0 + s) + 4 * q);
#pragma unroll
            for (int s = 0; s < GK; ++s)
#pragma unroll
                for (int q = 0; q < 2; ++q) b[s][q] = load4f(r1 + 16 * (s0 + s) + 4 * q);
#pragma unroll
            for (int s = 0; s < GK; ++s)
#pragma unroll
                for (int q = 0; q < 2; ++q) d[s][q] = load4f(r2 + 16 * (s0 + s) + 4 * q);
            if (s0 + GK >= KS && rd + rstep < rend) { request(rd + rstep, pn_next); pn_cur = pn_next; if (rd + 2 * rstep < rend) pn_next = point_of(rd + 2 * rstep); }
#pragma unroll
            for (int s = 0; s < GK; ++s) {
                float v[8];
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        v[4 * q + j] = interp3(a[s][q][j], b[s][q][j], d[s][q][j], w0, w1, w2);
                split8<NP>(v, yh[s0 + s], yl[s0 + s]);
            }
        }
        }
        // one channel block of a layer: accumulators start from the bias, MFMAs against the resident panel, ReLU + split into the
        // next layer's two k-step fragments
#define FP_BLOCK(L, NBI, XH, XL, OH, OL)   /* channel block NBI of layer L, read from the panel that holds it */                                                                                     \
    {                                                                                                                        \
        if constexpr (SPREAD) fetch_pieces(nextq, (NBI) % NBP);                                                              \
        f32x16 acc;                                                                                                          \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                      \
            const f32x4 bq = *reinterpret_cast<const f32x4*>(bl + lz + (L) * C + 32 * (NBI) + 8 * q + 4 * h);               \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[4 * q + j] = bq[j];                                            \
        }                                                                                                                    \
        _Pragma("unroll") for (int s = 0; s < KS; ++s) {                                                                     \
            const bf16x8 wh = XWFRAG(W, KS, (NBI) % NBP, s, 0, lane);                                                              \
            if constexpr (NP == 2) {                                                                                         \
                const bf16x8 wlo = XWFRAG(W, KS, (NBI) % NBP, s, 1, lane);                                                         \
                acc = XMFMA(wh, XL[s], acc);                                                                                 \
                acc = XMFMA(wlo, XH[s], acc);                                                                                \
            }                                                                                                                \
            acc = XMFMA(wh, XH[s], acc);                                                                                     \
            if (s % 3 == 2) __builtin_amdgcn_sched_barrier(0);                                                               \
        }                                                                                                                    \
        _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) {                                                                   \
            float v[8];                                                                                                      \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) v[j] = reluf(acc[8 * s2 + j]);                                     \
            XSPLIT(v, OH[2 * (NBI) + s2], OL[2 * (NBI) + s2]);                                                               \
        }                                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                                   \
    }
        // a layer: start the copy of the next layer's panel (its buffer was last read one barrier ago), four channel blocks from
        // this layer's panel, then wait for the own copies and the barrier that publishes the panel
#define FP_PANEL(L, H, XH, XL, OH, OL, BODY)                                                                                 \
    {                                                                                                                        \
        constexpr int nextq = ((L) * PH + (H) + 1) % (NL * PH);                                                              \
        if constexpr (!SPREAD) fetch(nextq);                                                                                 \
        const __bf16* W = &wl[((L) * PH + (H)) & 1][0];                                                                      \
        int lz = 0;                                                                                                          \
        asm volatile("" : "+v"(lz));                                                                                         \
        W += lz;                                                                                                             \
        BODY                                                                                                                 \
        XBARRIER                                                                                                             \
    }
#define FP_LAYER(L, XH, XL, OH, OL)                                                                                          \
    if constexpr (PH == 1) {                                                                                                 \
        FP_PANEL(L, 0, XH, XL, OH, OL, FP_BLOCK(L, 0, XH, XL, OH, OL) FP_BLOCK(L, 1, XH, XL, OH, OL) FP_BLOCK(L, 2, XH, XL, OH, OL) FP_BLOCK(L, 3, XH, XL, OH, OL)) \
    } else {                                                                                                                 \
        FP_PANEL(L, 0, XH, XL, OH, OL, FP_BLOCK(L, 0, XH, XL, OH, OL) FP_BLOCK(L, 1, XH, XL, OH, OL))                        \
        FP_PANEL(L, 1, XH, XL, OH, OL, FP_BLOCK(L, 2, XH, XL, OH, OL) FP_BLOCK(L, 3, XH, XL, OH, OL))                        \
    }
        FP_LAYER(0, yh, yl, zh, zl)
        FP_LAYER(1, zh, zl, yh, yl)
        FP_LAYER(2, yh, yl, zh, zl)
        FP_LAYER(3, zh, zl, yh, yl)
#undef FP_LAYER
#undef FP_PANEL
#undef FP_BLOCK
        // ---- conv2 (128 -> 3, rows zero-padded to one 32-channel block), direct orientation D[point][channel] -------------------
        {
            int lz = 0;
            asm volatile("" : "+v"(lz));
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 wh = W5LDS ? wfrag_lds<NP>(w5 + lz, KS, 0, s, 0, lane) : gload<bf16x8>(opaque_ptr(w5g) + ((size_t)(s * NP) * 64 + lane) * 8);
                if constexpr (NP == 2) {
                    const bf16x8 wlo = W5LDS ? wfrag_lds<NP>(w5 + lz, KS, 0, s, 1, lane) : gload<bf16x8>(opaque_ptr(w5g) + ((size_t)(s * NP + 1) * 64 + lane) * 8);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yl[s], wh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yh[s], wlo, acc, 0, 0, 0);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yh[s], wh, acc, 0, 0, 0);
            }
            {
                // accumulator r of lane (p, h) = channel p of the wave's point acc_row(r, lane): its output row is that point's -- read out of the
                // lane that owns it
                const float bv = bl[lz + NL * C + (p < 3 ? p : 0)];
                const size_t cb = (size_t)c * N;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int pr = __shfl(pn_this, acc_row(r, lane));
                    if (p < 3) out[(cb + pr) * 3 + p] = acc[r] + bv;
                }
            }
        }
    }
}

void launch_fp_s(int nc, const float* points2, const int* nn_idx, const float* nn_w, const __bf16* blob_w, const float* blob_b, float* out,
                 hipStream_t st, const int* order, int order_mod) {
    // timing experiment (compile-time only, `make TIMING=1`; results are wrong with it): 1 sequential rows, 2 one row per wave
#ifdef LSDM_TIMING_EXPERIMENTS
    static const int dbg = getenv("LSDM_DBG_GATHER") ? atoi(getenv("LSDM_DBG_GATHER")) : 0;
#else
    constexpr int dbg = 0;
#endif
    const long long tot = (long long)nc * 1024;
    // (12 waves -- three per SIMD, as sa3 runs -- do not fit: two 128-channel operand sets + accumulators + fragments need ~172 VGPRs against
    // the 168 of that occupancy and the build spills 95 of them)
    // (two work-groups of four waves per CU on half-layer panels -- fp_s_kernel<4, 2, 2>, grid 512: the two groups drift out of phase, one gathers
    // while the other multiplies, for twice the panel traffic -- measured 290 against 271 us, round 4)
    hipLaunchKernelGGL((fp_s_kernel<8, 2>), dim3(256), dim3(512), 0, st, points2, nn_idx, nn_w, 1024, 1024, tot, blob_w, blob_b, out, dbg, order, order_mod);
}

void launch_sa_r(int np, int level, int nc, const float* src_xyz, const void* src_feat, const float* new_xyz, const int* ball,
                 const __bf16* blob_w, const float* blob_b, void* out, hipStream_t st) {
    // persistent grids: sa1 (15 KB of LDS, 57 VGPRs) 4 workgroups of 8 waves per CU, sa2 (69 KB, 123 VGPRs) 2 per CU
    if (level == 0) {
        const long long tot = (long long)nc * 1024;
        if (np == 2) hipLaunchKernelGGL((sa_r_kernel<3, 32, 32, 64, 8, 2>), dim3(1024), dim3(512), 0, st, src_xyz, static_cast<const float*>(src_feat),
                                        new_xyz, ball, 1024, 1024, tot, blob_w, blob_b, static_cast<float*>(out), nullptr, nullptr);
        else hipLaunchKernelGGL((sa_r_kernel<3, 32, 32, 64, 8, 1>), dim3(1024), dim3(512), 0, st, src_xyz, static_cast<const __bf16*>(src_feat),
                                new_xyz, ball, 1024, 1024, tot, blob_w, blob_b, static_cast<__bf16*>(out), nullptr, nullptr);
    } else if (level == 2) {
        const long long tot = (long long)nc * 64;
        if (np == 2) hipLaunchKernelGGL((sa_s_kernel<128, 128, 128, 256, 4, 2>), dim3(512), dim3(256), 0, st, src_xyz, static_cast<const float*>(src_feat),
                                        new_xyz, ball, 256, 64, tot, blob_w, blob_b, static_cast<float*>(out));
        else hipLaunchKernelGGL((sa_s_kernel<128, 128, 128, 256, 4, 1>), dim3(512), dim3(256), 0, st, src_xyz, static_cast<const __bf16*>(src_feat),
                                new_xyz, ball, 256, 64, tot, blob_w, blob_b, static_cast<__bf16*>(out));
    } else {
        const long long tot = (long long)nc * 256;
        if (np == 2) hipLaunchKernelGGL((sa_r_kernel<64, 64, 64, 128, 8, 2>), dim3(512), dim3(512), 0, st, src_xyz, static_cast<const float*>(src_feat),
                                        new_xyz, ball, 1024, 256, tot, blob_w, blob_b, static_cast<float*>(out), nullptr, nullptr);
        else hipLaunchKernelGGL((sa_r_kernel<64, 64, 64, 128, 8, 1>), dim3(512), dim3(512), 0, st, src_xyz, static_cast<const __bf16*>(src_feat),
                                new_xyz, ball, 1024, 256, tot, blob_w, blob_b, static_cast<__bf16*>(out), nullptr, nullptr);
    }
}

// sa1 (bf16x3) over the item list of launch_sa_pair_plan: rows = 32 source rows per item, meta = {centroid rows, output rows}, *count items
void launch_sa_r_items(const float* src_xyz, const int* rows, const int4* meta, const int* count, const __bf16* blob_w, const float* blob_b, float* out,
                       hipStream_t st) {
    // (70 VGPRs against the plain form's 64: three work-groups of 8 waves per CU instead of four)
    hipLaunchKernelGGL((sa_r_kernel<3, 32, 32, 64, 8, 2, true>), dim3(768), dim3(512), 0, st, src_xyz, static_cast<const float*>(nullptr), src_xyz, rows, 1024, 1024, 0LL,
                       blob_w, blob_b, out, meta, count);
}

}  // namespace lsdm
