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
//   sa_p_kernel (sa3) / sa_w_kernel (sa4) / sa_ur_kernel (sa2) : register-chained two-layer stack on ReLU(U_j - V_i) + max over the 32 neighbours
#include <cstdlib>
#include <utility>

#include "bf16_core.hpp"
#include "kernels.hpp"

namespace lsdm {

namespace {

template <int NP>
__device__ __forceinline__ void split8u(const float (&v)[8], bf16x8& hi, bf16x8& lo) { split8_h<NP>(v, hi, lo); }
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
// (V only for the centroids, compact, from a launch of its own -- one row in four is ever read, 75 + 38 MB of writes per step saved -- was
// measured: sa3 317 vs 318 us, sa4 344 vs 340 incl. the U / V launches: the stage is not bound by that write; not kept.  A persistent form -- 461 work-groups
// walking five tiles each, the next tile's rows requested before this tile's matrix instructions -- measured slower as well, sa3 stage 323 vs 320 us, strict sa2
// 363 vs 356, round 4: two one-tile work-groups per CU already overlap each other's loads and stores)
template <int K, int N, bool XYZ = true, int NP = 2>
__global__ __launch_bounds__(256, 2) void lin_u_kernel(const float* __restrict__ X, const float* __restrict__ xyz, int R, DenseH W,
                                                       const float* __restrict__ wx, float* __restrict__ U, float* __restrict__ V) {
    constexpr int TM = 64, NW = 4, NT = NW * 64;
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
        if constexpr (XYZ) {
            for (int e = tid; e < TM * 3; e += NT) {
                const int row = e / 3;
                const int rr = r0 + row < R ? r0 + row : R - 1;
                sxyz[e] = xyz[(size_t)rr * 3 + (e - row * 3)];
            }
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
        const float bias = W.b ? W.b[col] : 0.0f;
        float wx0 = 0.0f, wx1 = 0.0f, wx2 = 0.0f;
        if constexpr (XYZ) { wx0 = wx[col * 3 + 0]; wx1 = wx[col * 3 + 1]; wx2 = wx[col * 3 + 2]; }
        mma_kloop_h<TS::RB, NP, false, K / 16>(acc, bufX + rg * TS::RB * 32 * LDX, LDX, PX, W.w, N, cb, 0, lane);
#pragma unroll
        for (int rb = 0; rb < TS::RB; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (rg * TS::RB + rb) * 32 + acc_row(r, lane);
                if (r0 + row < R) {
                    if constexpr (XYZ) {
                        const float v = fmaf(wx2, sxyz[row * 3 + 2], fmaf(wx1, sxyz[row * 3 + 1], wx0 * sxyz[row * 3 + 0]));
                        V[(size_t)(r0 + row) * N + col] = v;
                        U[(size_t)(r0 + row) * N + col] = (acc[rb][r] + bias) + v;
                    } else {
                        U[(size_t)(r0 + row) * N + col] = acc[rb][r] + bias;
                    }
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------
// sa_p_kernel (sa3): register-chained stack (cf. encoder_reg.hip::sa_r_kernel) on ReLU(U_j - V_i): layer 2 (C1 -> C2) RESIDENT in LDS
// (64 KB of fragments for the life of the persistent work-group), layer 3 (C2 -> C3, max over the 32 neighbours) streamed through two
// 32 KB panels by LDS-DMA (0.5 KB of weights per point; streaming both layers was L2-bound in round 2), with the epilogues
// SOFTWARE-PIPELINED against the next channel block's matrix instructions.
// blob: [W2 frags: C2/32 x C1/16 x NP x 512][W3: C3/32 x C2/16 x NP x 512], biases [b2 | b3]; the k-slot (s, h, j) of layer 2 is
// channel 16 s + 8 h + j of U (identity order), layer 3 follows the accumulator order.
//
// Round 2's form of this kernel (sa_u2_kernel: a block's MFMAs, then its epilogue) met a barrier four times per round, so the waves
// of a SIMD ran in the same phase: all want the matrix pipe, then all run their ReLU + hi/lo split (or max-pool) epilogue while the
// pipe idles -- nothing overlaps (tools/ubench/gen_mfma_fill.py, profiles/r03_ubench_mfma_fill.txt: 8 MFMAs followed by a block of
// 40 VALU instructions cost 32.0 ticks per MFMA at two waves per SIMD, the same instructions interleaved 5 per MFMA 25.9 against
// 24.8 for bare MFMAs).  Here a wave ping-pongs two accumulators: the epilogue of channel block n reads accumulator n & 1 while the 24 matrix instructions
// of block n + 1 fill the other, and __builtin_amdgcn_sched_group_barrier pins the interleave (1 MFMA : 2 VALU in layer 2, 1 : 1 in
// layer 3).  The cross-half max of the pooling uses v_permlane32_swap (VALU) instead of ds_bpermute, and both half-waves store
// (same value, same address), so the pooled epilogue has no branch and no LDS round trip.  Twelve waves (three per SIMD) share a
// panel pass.  Same-box A/B at B = 64 (gpurun_out/abv_*.log, round 3): sa_u2 355-360 us per launch incl. the 48 us U/V launch, this
// kernel with 8 waves 342-345, with 12 waves 320-324; results bit-identical to sa_u2 (same products, same accumulation order).
// max(m[l], m[l ^ 32]) in every lane: v_permlane32_swap exchanges the upper half of its first operand with the lower half of its
// second, so after swapping two copies of m one holds m[l & 31] and the other m[(l & 31) + 32] in all lanes.  Written as asm: with
// the builtin hipcc (ROCm 7.2) folds the max over the two results of one swap of a value with itself away.
__device__ __forceinline__ float max_halves(float m) {
    float a = m, b = m;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// scheduling pattern of one channel block of KS k-steps (3 MFMAs each) next to NV VALU instructions per MFMA: per k-step two LDS
// fragment reads, then MFMA / VALU alternating; the first LEAD MFMAs run without VALU (the epilogue they cover reads the previous
// block's accumulator, whose last MFMA is still in flight)
template <int KS, int NV, int LEAD, int NM = 3>
__device__ __forceinline__ void sgb_block() {
    static_for<KS>([&](auto s_) {
        constexpr int s = decltype(s_)::value;
        static_for<NM>([&](auto m_) {
            constexpr int m = decltype(m_)::value;
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if constexpr (NM * s + m >= LEAD) __builtin_amdgcn_sched_group_barrier(0x002, NV * 3 / NM, 0);
        });
    });
}

// PAIR (launch_sa_pair_plan, geometry.hip): the launch walks a list of ITEMS instead of the centroids.  A ball-query group is padded to 32 rows
// with repeats of its first neighbour; the repeats change nothing under the max, so two groups whose distinct neighbours fit into 32 rows
// together share ONE 32-row block: rows 0 .. 4k-1 belong to the first centroid ("A"), rows 4k .. 31 to the second ("B"), each with its own
// V row, its own maximum and its own output row.  Row r of the block is accumulator register (r & 3) + 4 (r >> 3) of half-wave (r >> 2) & 1:
// register quartet g of half-wave h is A's iff 2 g + h < k.  `ball` is then the item list's row table (32 U rows per item, numbered over
// the launch), `pmeta` {V row of A, of B, output row of A | k << 24, of B} (k = 8: an unpaired item) and *pcount the number of items.
template <int C1, int C2, int C3, int NW, int NP, bool PAIR = false>
__global__ __launch_bounds__(NW * 64) void sa_p_kernel(const float* __restrict__ U, const float* __restrict__ V,
                                                        const int* __restrict__ cent_idx, const int* __restrict__ ball, int NSRC, int S,
                                                        long long total_centroids, const __bf16* __restrict__ blob_w,
                                                        const float* __restrict__ blob_b, float* __restrict__ out,
                                                        const int4* __restrict__ pmeta, const int* __restrict__ pcount) {
    constexpr int NT = NW * 64;
    constexpr int KS2 = C1 / 16, KS3 = C2 / 16;
    constexpr int NM = NP == 2 ? 3 : 1;                      // matrix instructions per k-step
    constexpr int NB2 = C2 / 32, NB3 = C3 / 32;
    constexpr int F2 = KS2 * NP, F3 = KS3 * NP;
    constexpr int W2E = NB2 * F2 * 512;
    constexpr int PB = 2;
    constexpr int NPANEL = NB3 / PB, FP = PB * F3;
    static_assert(NB3 % PB == 0 && NPANEL % 2 == 0 && NB2 % 2 == 0, "ping-pong parity is compile-time");
    constexpr int NLD = NW > 8 ? 8 : NW;                      // waves that copy panels
    static_assert(FP % NLD == 0, "panel split over the waves");
    __shared__ __attribute__((aligned(16))) __bf16 w2[W2E];
    __shared__ __attribute__((aligned(16))) __bf16 wl[2][FP * 512];
    __shared__ __attribute__((aligned(16))) float bl[C2 + C3];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    for (int e = tid; e < W2E / 8; e += NT) *reinterpret_cast<bf16x8*>(w2 + (size_t)e * 8) = *reinterpret_cast<const bf16x8*>(blob_w + (size_t)e * 8);
    for (int e = tid; e < C2 + C3; e += NT) bl[e] = blob_b[e];
    const int p = lane & 31, h = lane >> 5;
    const __bf16* w3g = blob_w + W2E;
    auto fetch = [&](int q) {
        const __bf16* src = opaque_ptr(w3g + (size_t)q * FP * 512);
        if (wave < NLD) {
#pragma unroll
            for (int i = 0; i < FP / NLD; ++i) {
                const int fr = wave + i * NLD;
                lds_dma16(src + (size_t)fr * 512 + lane * 8, __builtin_amdgcn_readfirstlane(lds_addr(&wl[q & 1][(size_t)fr * 512])));
            }
        }
    };
    fetch(0);
    lds_dma_wait();
    __syncthreads();

    const int total = PAIR ? __builtin_amdgcn_readfirstlane(*pcount) : (int)total_centroids;
    // An even, contiguous share of the blocks for every work-group (consecutive shares on one XCD; the XCD-cooperative schedule of sa2 / fp1
    // was measured without effect here and in sa_w: their HBM fetch already equals the unique input).  The LAST round of a share is partial:
    // its idle waves skip everything but the panel copies and the barriers, so the waves that do have a block run it with a SIMD (nearly) to
    // themselves -- with whole rounds per work-group a launch of 10.3 rounds' worth of blocks took 11.
    const int G = (int)gridDim.x, bx = xcd_remap((int)blockIdx.x, G);
    const int per = total / G, extra = total - per * G;
    const int ibeg = bx * per + (bx < extra ? bx : extra), nit = per + (bx < extra ? 1 : 0);
    const int rbeg = 0, rend = (nit + NW - 1) / NW;
    constexpr int rstep = 1;
    auto cent_of = [&](int rd) { const int k = rd * NW + wave; return ibeg + (k < nit ? k : nit - 1); };
    int idx_n = 0, ci_n = 0;
    int4 mt_n = {0, 0, 0, 0};
    auto request_idx = [&](int ce) {
        idx_n = ball[(size_t)ce * kNSample + p];
        if constexpr (PAIR) mt_n = pmeta[ce];
        else ci_n = cent_idx[ce];
    };
    if (rbeg < rend) request_idx(cent_of(rbeg));
    for (int rd = rbeg; rd < rend; rd += rstep) {
        if (rd * NW + wave >= nit) {           // no block for this wave in the share's last round: its part of the panel stream only
#pragma unroll 1
            for (int q = 0; q < NPANEL; ++q) {
                fetch((q + 1) % NPANEL);
                lds_dma_wait();
                __syncthreads();
            }
            continue;
        }
        const int cent = cent_of(rd);
        const int c = cent / S;
        const int4 mt = mt_n;
        const int ka = PAIR ? mt.z >> 24 : 8;              // A's row quartets
        const bool paired = ka < 8;
        const bool own[4] = {h < ka, 2 + h < ka, 4 + h < ka, 6 + h < ka};
        const size_t orow = PAIR ? (size_t)(h ? mt.w : mt.z & 0xffffff) : (size_t)cent;
        const float* urow = U + (PAIR ? (size_t)idx_n : (size_t)c * NSRC + idx_n) * C1 + 8 * h;
        const float* vrow = V + (PAIR ? (size_t)(p < 4 * ka ? mt.x : mt.y) : (size_t)c * NSRC + ci_n) * C1 + 8 * h;
        bf16x8 yh[KS2], yl[KS2];
        {
            f32x4 u0[KS2], u1[KS2], v0[KS2], v1[KS2];
#pragma unroll
            for (int s = 0; s < KS2; ++s) {
                u0[s] = load4f(urow + 16 * s); u1[s] = load4f(urow + 16 * s + 4);
                v0[s] = load4f(vrow + 16 * s); v1[s] = load4f(vrow + 16 * s + 4);
            }
            if (rd + rstep < rend) request_idx(cent_of(rd + rstep));
#pragma unroll
            for (int s = 0; s < KS2; ++s) {
                const float t[8] = {reluf(u0[s][0] - v0[s][0]), reluf(u0[s][1] - v0[s][1]), reluf(u0[s][2] - v0[s][2]), reluf(u0[s][3] - v0[s][3]),
                                    reluf(u1[s][0] - v1[s][0]), reluf(u1[s][1] - v1[s][1]), reluf(u1[s][2] - v1[s][2]), reluf(u1[s][3] - v1[s][3])};
                split8u<NP>(t, yh[s], yl[s]);
            }
        }
        bf16x8 zh[KS3], zl[KS3];
        f32x16 acc[2];
        int lz = 0;
        asm volatile("" : "+v"(lz));          // opaque zero: the (round-invariant) LDS reads stay inside the loop
        // weight fragments travel through a three-slot register ring, requested two k-steps ahead of their MFMAs (the LDS latency
        // of a just-in-time read is exposed on every k-step otherwise: both waves of a SIMD run in lockstep between the barriers)
        bf16x8 fh[3], fl[3];
        auto frag = [&](const __bf16* W, int KS, int t, bf16x8& oh, bf16x8& ol) {      // flat step t of a stream of blocks of KS k-steps
            oh = wfrag_lds_u<NP>(W, KS, 0, t, 0, lane);
            if constexpr (NP == 2) ol = wfrag_lds_u<NP>(W, KS, 0, t, 1, lane);
        };
        auto l2_epi = [&](auto nb_, const f32x16& a) {
            constexpr int nb = decltype(nb_)::value;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = reluf(a[8 * s2 + j]);
                split8u<NP>(v, zh[2 * nb + s2], zl[2 * nb + s2]);
            }
        };
        // max over the 32 rows of block nb (ReLU(x + b) is monotone: pool first); both half-waves store the same values
        auto l3_epi = [&](int nb, const f32x16& a) {
            float m;
            if constexpr (PAIR) {
                // half-wave 0 stores A's row, half-wave 1 B's (an unpaired item: one maximum, the same row twice as below)
                m = max_split(a, own, paired);
            } else {
                m = a[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) m = fmaxf(m, a[r]);
                m = max_halves(m);
            }
            out[orow * C3 + nb * 32 + p] = reluf(m + bl[lz + C2 + nb * 32 + p]);
        };
        // ---- layer 2 (resident fragments): one stream of NB2 x KS2 k-steps; block nb accumulates in acc[nb & 1] while the
        // epilogue of block nb - 1 reads the other accumulator
        const __bf16* W2p = w2 + lz;
        frag(W2p, KS2, 0, fh[0], fl[0]);
        frag(W2p, KS2, 1, fh[1], fl[1]);
        static_for<NB2>([&](auto nb_) {
            constexpr int nb = decltype(nb_)::value;
            f32x16& a = acc[nb & 1];
            // (source order: the epilogue of the previous block first -- the pattern below moves this block's MFMAs up between its
            // instructions wherever the data allows)
            if constexpr (nb > 0) l2_epi(std::integral_constant<int, (nb > 0 ? nb - 1 : 0)>{}, acc[(nb - 1) & 1]);
            const float* bq = bl + lz + 32 * nb + 4 * h;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bq + 8 * q);
#pragma unroll
                for (int j = 0; j < 4; ++j) a[4 * q + j] = bv[j];
            }
            static_for<KS2>([&](auto s_) {
                constexpr int s = decltype(s_)::value, t = nb * KS2 + s;
                // (past the end of layer 2 the ring continues with the first k-steps of layer 3's panel 0, published a round ago)
                if constexpr (t + 2 < NB2 * KS2) frag(W2p, KS2, t + 2, fh[(t + 2) % 3], fl[(t + 2) % 3]);
                else frag(&wl[0][0] + lz, KS3, t + 2 - NB2 * KS2, fh[(t + 2) % 3], fl[(t + 2) % 3]);
                __builtin_amdgcn_sched_barrier(0x47f);      // the fragment reads keep their place two k-steps ahead; everything else may move
                if constexpr (NP == 2) {
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[t % 3], yl[s], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl[t % 3], yh[s], a, 0, 0, 0);
                }
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[t % 3], yh[s], a, 0, 0, 0);
            });
            if constexpr (nb > 0) sgb_block<KS2, 3, 2, NM>();
            __builtin_amdgcn_sched_barrier(0);
        });
        // ---- layer 3 (streamed panels of PB = 2 blocks): block j accumulates in acc[j & 1]; its pooled epilogue runs under block j + 1
        static_for<NPANEL>([&](auto q_) {
            constexpr int q = decltype(q_)::value;
            fetch((q + 1) % NPANEL);
            const __bf16* W = &wl[q & 1][0] + lz;
            if constexpr (q > 0) {                        // a new panel: its first two k-steps can only be requested after the barrier
                frag(W, KS3, 0, fh[0], fl[0]);
                frag(W, KS3, 1, fh[1], fl[1]);
            }
            static_for<PB>([&](auto u_) {
                constexpr int u = decltype(u_)::value, j = q * PB + u;
                f32x16& a = acc[j & 1];
                // the previous block's epilogue first (block 0 needs the last two k-steps of z that layer 2's last epilogue produces)
                if constexpr (j == 0) l2_epi(std::integral_constant<int, NB2 - 1>{}, acc[(NB2 - 1) & 1]);
                else l3_epi(j - 1, acc[(j - 1) & 1]);
#pragma unroll
                for (int r = 0; r < 16; ++r) a[r] = 0.0f;
                static_for<KS3>([&](auto s_) {
                    constexpr int s = decltype(s_)::value, t = u * KS3 + s;
                    constexpr int base = q == 0 ? (NB2 * KS2) % 3 : 0;   // ring slot of the panel's step 0 (panel 0 continues layer 2's ring)
                    if constexpr (t + 2 < PB * KS3) frag(W, KS3, t + 2, fh[(base + t + 2) % 3], fl[(base + t + 2) % 3]);
                    __builtin_amdgcn_sched_barrier(0x47f);
                    if constexpr (NP == 2) {
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zl[s], fh[(base + t) % 3], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zh[s], fl[(base + t) % 3], a, 0, 0, 0);
                    }
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zh[s], fh[(base + t) % 3], a, 0, 0, 0);
                });
                if constexpr (j == 0) sgb_block<KS3, 3, 2, NM>();
                else sgb_block<KS3, 1, 2, NM>();
                __builtin_amdgcn_sched_barrier(0);
            });
            lds_dma_wait();
            __syncthreads();
        });
        l3_epi(NB3 - 1, acc[(NB3 - 1) & 1]);
    }
}

// ---------------------------------------------------------------------------------------------
// sa_ur_kernel: the split first layer for a NARROW level (sa2: 64 -> 64 -> 128 on ReLU(U_j - V_i)) -- all fragments of both
// remaining layers (48 KB as hi + lo) stay in LDS for the life of a persistent work-group, each wave carries one centroid
// through the stack in registers (encoder_reg.hip::sa_r_kernel without its first layer: 72 instead of 102 MFMAs per centroid).
// `remap` (optional, per cloud NSRC entries): U / V are indexed by remap[position] instead of the position -- the loop-invariant
// schedule evaluates U, V once per call in SOURCE order and reads them through the step's level-1 FPS order.
// PAIR: the launch walks an item list of launch_sa_pair_plan (see sa_p_kernel: two groups in one 32-row block) -- `ball` = the list's row table,
// `pmeta` its V / output rows, *list_count its length; cent_idx, remap, list and src_mod are not read.
template <int C1, int C2, int C3, int NW, int NP, bool PAIR = false>
__global__ __launch_bounds__(NW * 64) void sa_ur_kernel(const float* __restrict__ U, const float* __restrict__ V,
                                                         const int* __restrict__ cent_idx, const int* __restrict__ ball,
                                                         const int* __restrict__ remap, int NSRC, int S, long long total_centroids,
                                                         const __bf16* __restrict__ blob_w, const float* __restrict__ blob_b,
                                                         float* __restrict__ out, int src_mod, const int* __restrict__ list,
                                                         const int* __restrict__ list_count, const int4* __restrict__ pmeta) {
    // list (optional): the launch covers the centroids list[0 .. *list_count) (global centroid numbers, any order) instead of
    // 0 .. total_centroids -- the loop-invariant schedule's memo of sa2 (api.hip: memo_ready)
    constexpr int KS2 = C1 / 16, KS3 = C2 / 16;
    constexpr int NB2 = C2 / 32, NB3 = C3 / 32;
    constexpr int W2E = NB2 * KS2 * NP * 512, W3E = NB3 * KS3 * NP * 512;
    __shared__ __attribute__((aligned(16))) __bf16 wl[W2E + W3E];
    __shared__ __attribute__((aligned(16))) float bl[C2 + C3];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    for (int e = tid; e < (W2E + W3E) / 8; e += NW * 64)
        *reinterpret_cast<bf16x8*>(wl + (size_t)e * 8) = *reinterpret_cast<const bf16x8*>(blob_w + (size_t)e * 8);
    for (int e = tid; e < C2 + C3; e += NW * 64) bl[e] = blob_b[e];
    __syncthreads();
    const __bf16* W2 = wl;
    const __bf16* W3 = wl + W2E;
    const float* b2 = bl;
    const float* b3 = bl + C2;
    const int p = lane & 31, h = lane >> 5;
    // 32-bit index arithmetic (a 64-bit division per centroid costs ~60 instructions); the neighbour / centroid indices of a
    // centroid -- two dependent loads when they go through `remap` -- are requested one iteration ahead
    const int total = (PAIR || list) ? *list_count : (int)total_centroids;
    const XcdRounds sched = xcd_rounds((total + NW - 1) / NW);       // a round = NW centroids, one per wave (common.hpp)
    const int cbeg = sched.first * NW, cstep = sched.stride * NW;
    const int cend = sched.end * NW < total ? sched.end * NW : total;
    int idx_n = 0, ci_n = 0, cent_n = 0;
    int4 mt_n = {0, 0, 0, 0};
    auto request = [&](int k) {
        if constexpr (PAIR) {
            idx_n = ball[(size_t)k * kNSample + p];
            mt_n = pmeta[k];
            return;
        }
        const int ce = list ? list[k] : k;
        const int cc = ce / S;
        cent_n = ce;
        idx_n = ball[(size_t)ce * kNSample + p];
        ci_n = cent_idx[ce];
        if (remap) { idx_n = remap[(size_t)cc * NSRC + idx_n]; ci_n = remap[(size_t)cc * NSRC + ci_n]; }
    };
    // LSDM_SAUR_PREFETCH (default): the 32 U rows of the NEXT centroid are requested between this centroid's layers 2 and 3 -- its indices,
    // requested a centroid ahead, have landed by then -- so that their L2 round trip runs under layer 3's matrix instructions instead of in
    // front of the next centroid's (the kernel has no barrier in its loop, but at four waves per SIMD a wave's own gather latency was a third of
    // its iteration).  32 more registers (126: still four waves per SIMD).
#ifndef LSDM_SAUR_PREFETCH
#define LSDM_SAUR_PREFETCH 1
#endif
    constexpr bool PF = LSDM_SAUR_PREFETCH != 0;
    [[maybe_unused]] f32x4 un0[KS2], un1[KS2];
    auto cloud_of = [&](int cent) { return src_mod > 0 ? (cent / S) % src_mod : cent / S; };
    auto request_u = [&]() {        // the rows of the centroid whose indices `request` fetched last
        const float* urow = U + (PAIR ? (size_t)idx_n : (size_t)cloud_of(cent_n) * NSRC + idx_n) * C1 + 8 * h;
#pragma unroll
        for (int s = 0; s < KS2; ++s) { un0[s] = load4f(urow + 16 * s); un1[s] = load4f(urow + 16 * s + 4); }
    };
    if (cbeg + wave < cend) {
        request(cbeg + wave);
        if constexpr (PF) request_u();
    }
    for (int kk = cbeg + wave; kk < cend; kk += cstep) {
        const int cent = cent_n;
        // (src_mod > 0: a launch over the clouds of several steps whose U / V are per-CALL tensors in source order)
        const int c = cloud_of(cent);
        const int idx = idx_n, ci = ci_n;
        const int4 mt = mt_n;
        const int ka = PAIR ? mt.z >> 24 : 8;              // A's row quartets (sa_p_kernel)
        const bool paired = ka < 8;
        const bool own[4] = {h < ka, 2 + h < ka, 4 + h < ka, 6 + h < ka};
        const size_t orow = PAIR ? (size_t)(h ? mt.w : mt.z & 0xffffff) : (size_t)cent;
        const float* urow = U + (PAIR ? (size_t)idx : (size_t)c * NSRC + idx) * C1 + 8 * h;
        const float* vrow = V + (PAIR ? (size_t)(p < 4 * ka ? mt.x : mt.y) : (size_t)c * NSRC + ci) * C1 + 8 * h;
        bf16x8 yh[KS2], yl[KS2];
        {
            f32x4 u0[KS2], u1[KS2], v0[KS2], v1[KS2];
#pragma unroll
            for (int s = 0; s < KS2; ++s) {
                if constexpr (PF) { u0[s] = un0[s]; u1[s] = un1[s]; }
                else { u0[s] = load4f(urow + 16 * s); u1[s] = load4f(urow + 16 * s + 4); }
                v0[s] = load4f(vrow + 16 * s); v1[s] = load4f(vrow + 16 * s + 4);
            }
            if (kk + cstep < cend) request(kk + cstep);
#pragma unroll
            for (int s = 0; s < KS2; ++s) {
                const float t[8] = {reluf(u0[s][0] - v0[s][0]), reluf(u0[s][1] - v0[s][1]), reluf(u0[s][2] - v0[s][2]), reluf(u0[s][3] - v0[s][3]),
                                    reluf(u1[s][0] - v1[s][0]), reluf(u1[s][1] - v1[s][1]), reluf(u1[s][2] - v1[s][2]), reluf(u1[s][3] - v1[s][3])};
                split8u<NP>(t, yh[s], yl[s]);
            }
        }
        // ---- layer 2: accumulators start from the bias; block nb feeds k-steps 2 nb, 2 nb + 1 of layer 3
        bf16x8 zh[KS3], zl[KS3];
#pragma unroll
        for (int nb = 0; nb < NB2; ++nb) {
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(b2 + 32 * nb + 8 * q + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[4 * q + j] = bv[j];
            }
#pragma unroll
            for (int s = 0; s < KS2; ++s) {
                const bf16x8 wh = wfrag_lds_u<NP>(W2, KS2, nb, s, 0, lane);
                if constexpr (NP == 2) {
                    const bf16x8 wlo = wfrag_lds_u<NP>(W2, KS2, nb, s, 1, lane);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, yl[s], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo, yh[s], acc, 0, 0, 0);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, yh[s], acc, 0, 0, 0);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = reluf(acc[8 * s2 + j]);
                split8u<NP>(v, zh[2 * nb + s2], zl[2 * nb + s2]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (PF) {
            if (kk + cstep < cend) request_u();
        }
        // ---- layer 3, direct orientation, max over the 32 neighbours (ReLU(x + b) is monotone: pool first)
#pragma unroll
        for (int nb = 0; nb < NB3; ++nb) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
            for (int s = 0; s < KS3; ++s) {
                const bf16x8 wh = wfrag_lds_u<NP>(W3, KS3, nb, s, 0, lane);
                if constexpr (NP == 2) {
                    const bf16x8 wlo = wfrag_lds_u<NP>(W3, KS3, nb, s, 1, lane);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zl[s], wh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zh[s], wlo, acc, 0, 0, 0);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zh[s], wh, acc, 0, 0, 0);
            }
            float m;
            if constexpr (PAIR) m = max_split(acc, own, paired);      // half-wave 0 stores A's row, half-wave 1 B's
            else {
                m = acc[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[r]);
                m = max_halves(m);      // both half-waves store the same values: no branch, no LDS round trip
            }
            out[orow * C3 + nb * 32 + p] = reluf(m + b3[nb * 32 + p]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// sa_w_kernel: the WIDE level (sa4) with ONE wave per SIMD and the whole register file behind it.
//
// A wave owns a centroid's 32 rows and carries BOTH 256-wide operands of the chained layers in registers (x: 128, z: 128 -- with one
// wave per SIMD the 512-entry register file of gfx950 belongs to it; the matrix instructions read their operands from either half).
// No exchange between waves, no K split: 4 centroids = 128 rows per pass over the weights, twelve panels of 64 KB (two
// channel blocks x all of K) by LDS-DMA, one barrier per panel.  The two channel blocks of a panel are two INDEPENDENT accumulator
// chains, issued alternately -- a single dependent chain leaves the matrix pipe idle a quarter of the time at one wave per SIMD
// (tools/ubench/mfma_clock: 1.59 vs 2.08 PFLOP/s).
// blob: build_chain_blob({conv1 256 x 256, conv2 512 x 256}, identity first map): [nb][s = 0..15][plane][lane][8]; biases [256 | 512].
// (Round 3: the epilogue of panel Q - 1 software-pipelined against panel Q's 96 matrix instructions on a second accumulator set, as in
// sa_p_kernel, measured no change -- 353 vs 353-355 us: at one wave per SIMD this kernel waits for its 64 KB weight panels, not for
// its epilogues; skipping the panel copies gives 317 us.)
#ifdef LSDM_PHASE_TIMING
__device__ unsigned long long g_saw_phase[64 * 8];   // 64 spread copies: 0 prologue, 1 layer-2 k loops, 2 layer-2 epilogues, 3 layer-3 k loops, 4 max + store, 5 panel wait + barrier, 7 rounds
#define SAW_T0() unsigned long long st_ = __builtin_amdgcn_s_memtime(); unsigned long long sd_[6] = {}
#define SAW_MARK(slot) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); sd_[slot] += n_ - st_; st_ = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define SAW_FLUSH(nr) do { if (lane == 0 && wave == 0) { unsigned long long* g_ = g_saw_phase + (blockIdx.x & 63) * 8; \
        for (int i_ = 0; i_ < 6; ++i_) atomicAdd(g_ + i_, sd_[i_]); atomicAdd(g_ + 7, (unsigned long long)(nr)); } } while (0)
extern "C" __attribute__((visibility("default"))) int lsdm_debug_saw_phase(unsigned long long* out8, int reset) {
    static unsigned long long h[64 * 8], z[64 * 8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_saw_phase), sizeof h) != hipSuccess) return -1;
    for (int s = 0; s < 8; ++s) { out8[s] = 0; for (int c = 0; c < 64; ++c) out8[s] += h[c * 8 + s]; }
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_saw_phase), z, sizeof z) != hipSuccess) return -1;
    return 0;
}
#else
#define SAW_T0() do {} while (0)
#define SAW_MARK(slot) do {} while (0)
#define SAW_FLUSH(nr) do {} while (0)
#endif
// PAIR: the launch walks the item list of sa_pair_plan (see sa_p_kernel): two groups of at most 16 distinct neighbours in one 32-row block.
template <int NP, bool PAIR = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void sa_w_kernel(
    const float* __restrict__ U, const float* __restrict__ V, const int* __restrict__ cent_idx, const int* __restrict__ ball,
    long long total_centroids, const __bf16* __restrict__ blob_w, const float* __restrict__ blob_b, float* __restrict__ out,
    const int4* __restrict__ pmeta, const int* __restrict__ pcount) {
    constexpr int C1 = 256, C2 = 256, C3 = 512, NSRC = 64, S = 16, NW = 4, NT = NW * 64;
    constexpr int KS = C1 / 16;
    constexpr int NB2 = C2 / 32, NB3 = C3 / 32, PB = 2, NPANEL = (NB2 + NB3) / PB;
    constexpr int FB = KS * NP, FP = PB * FB;                 // fragments (1 KiB) per channel block / per panel
    static_assert(C1 == C2 && NPANEL % 2 == 0 && FP % NW == 0, "panel split");
    __shared__ __attribute__((aligned(16))) __bf16 wl[2][FP * 512];
    __shared__ __attribute__((aligned(16))) float bl[C2 + C3];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid; e < C2 + C3; e += NT) bl[e] = blob_b[e];
    const int p = lane & 31, h = lane >> 5;
    auto fetch = [&](int q) {
        const __bf16* src = opaque_ptr(blob_w + (size_t)q * FP * 512);
#pragma unroll
        for (int i = 0; i < FP / NW; ++i) {
            const int fr = wave + i * NW;
            lds_dma16(src + (size_t)fr * 512 + lane * 8, __builtin_amdgcn_readfirstlane(lds_addr(&wl[q & 1][(size_t)fr * 512])));
        }
    };
    // the same copy in pieces: piece i of panel q (i < FP / NW), issued from INSIDE the k loop of the panel before it (LSDM_SAW_SPREAD).  Issued
    // as one block in front of a panel's 96 matrix instructions, the 16 copies of a wave (~60 issue cycles each) run with the matrix pipe
    // idle -- at one wave per SIMD nothing else covers them; spread two per k-step over the first half of the k loop they issue in the
    // shadow of the MFMAs and still have half a panel's time to land.
    auto fetch_piece = [&](int q, int i) {
        const __bf16* src = opaque_ptr(blob_w + (size_t)q * FP * 512);
        const int fr = wave + i * NW;
        lds_dma16(src + (size_t)fr * 512 + lane * 8, __builtin_amdgcn_readfirstlane(lds_addr(&wl[q & 1][(size_t)fr * 512])));
    };
#ifndef LSDM_SAW_SPREAD
#define LSDM_SAW_SPREAD 1
#endif
    constexpr bool SPREAD = LSDM_SAW_SPREAD != 0;
#ifndef LSDM_SAW_SPREAD_STEPS
#define LSDM_SAW_SPREAD_STEPS (KS / 2)
#endif
    constexpr int SST = (FP / NW) < (LSDM_SAW_SPREAD_STEPS) ? (FP / NW) : (LSDM_SAW_SPREAD_STEPS);       // k-steps of a panel's k loop that issue copies of the next panel (from the first one on)
    constexpr int PPS = (FP / NW) / SST;             // pieces per k-step
    static_assert(!SPREAD || (FP / NW) % SST == 0, "spread");
    fetch(0);
    lds_dma_wait();
    __syncthreads();

    const long long total = PAIR ? (long long)__builtin_amdgcn_readfirstlane(*pcount) : total_centroids;
    const long long rounds = (total + NW - 1) / NW;      // (a wave past the end repeats the last item: same values, same addresses)
    const long long rchunk = (rounds + gridDim.x - 1) / gridDim.x;
    const long long rbeg = (long long)xcd_remap((int)blockIdx.x, (int)gridDim.x) * rchunk;
    const long long rend = rbeg + rchunk < rounds ? rbeg + rchunk : rounds;
    constexpr long long rstep = 1;
    // The neighbour / centroid indices of a round are requested one round ahead and its 32 U rows (the gather: 128 registers per
    // lane) at the start of the previous round's layer 3, when x is dead -- with one wave per SIMD nothing else would hide that latency.
    // (The centroid's V row -- the same 1 KB for all 32 rows of the wave -- travels with them: one 16-byte piece per lane, parked in LDS at
    // the top of the round and read from there; fetched at the top of the round it cost two exposed L2 round trips per round.)
    constexpr int NV = PAIR ? 2 : 1;                 // V rows of a block: one per centroid in it
    __shared__ __attribute__((aligned(16))) float vl[NW][NV][C1];
    int idx_n = 0, ci_n = 0;
    int4 mt_n = {0, 0, 0, 0};
    f32x4 un0[KS], un1[KS], vn[NV];
    auto item_of = [&](long long rd) { const long long ce = rd * NW + wave; return ce < total ? ce : total - 1; };
    auto request_idx = [&](long long ce) {
        idx_n = ball[ce * kNSample + p];
        if constexpr (PAIR) mt_n = pmeta[ce];
        else ci_n = cent_idx[ce];
    };
    auto request_u = [&](long long cc) {
        const float* urow = U + (PAIR ? (size_t)idx_n : (size_t)cc * NSRC + idx_n) * C1 + 8 * h;
#pragma unroll
        for (int s = 0; s < KS; ++s) { un0[s] = load4f(urow + 16 * s); un1[s] = load4f(urow + 16 * s + 4); }
        if constexpr (PAIR) {
            vn[0] = load4f(V + (size_t)mt_n.x * C1 + 4 * lane);
            vn[NV - 1] = load4f(V + (size_t)mt_n.y * C1 + 4 * lane);
        } else
            vn[0] = load4f(V + ((size_t)cc * NSRC + ci_n) * C1 + 4 * lane);
    };
    if (rbeg < rend) {
        const long long ce = item_of(rbeg);
        request_idx(ce);
        request_u(ce / S);
    }
    SAW_T0();
    [[maybe_unused]] int nrounds = 0;
    for (long long rd = rbeg; rd < rend; rd += rstep) {
        const long long cent = item_of(rd);
        const int4 mt = mt_n;                            // (this round's item; mt_n moves on to the next round's below)
        const int ka = PAIR ? mt.z >> 24 : 8;            // A's row quartets (sa_p_kernel)
        const bool paired = ka < 8;
        const bool own[4] = {h < ka, 2 + h < ka, 4 + h < ka, 6 + h < ka};
        const size_t orow = PAIR ? (size_t)(h ? mt.w : mt.z & 0xffffff) : (size_t)cent;
        bf16x8 xh[KS], xl[KS], zh[KS], zl[KS];
        {
#pragma unroll
            for (int k = 0; k < NV; ++k) *reinterpret_cast<f32x4*>(&vl[wave][k][4 * lane]) = vn[k];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const float* vrow = &vl[wave][PAIR && p >= 4 * ka ? 1 : 0][8 * h];
#pragma unroll
            for (int s0 = 0; s0 < KS; s0 += 8) {
                f32x4 v0[8], v1[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) { v0[s] = *reinterpret_cast<const f32x4*>(vrow + 16 * (s0 + s)); v1[s] = *reinterpret_cast<const f32x4*>(vrow + 16 * (s0 + s) + 4); }
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const f32x4 u0 = un0[s0 + s], u1 = un1[s0 + s];
                    const float t[8] = {reluf(u0[0] - v0[s][0]), reluf(u0[1] - v0[s][1]), reluf(u0[2] - v0[s][2]), reluf(u0[3] - v0[s][3]),
                                        reluf(u1[0] - v1[s][0]), reluf(u1[1] - v1[s][1]), reluf(u1[2] - v1[s][2]), reluf(u1[3] - v1[s][3])};
                    split8u<NP>(t, xh[s0 + s], xl[s0 + s]);
                }
            }
            if (rd + rstep < rend) request_idx(item_of(rd + rstep));
        }
        SAW_MARK(0);
#define SAW_BEGIN(Q) { if (!SPREAD) fetch(((Q) + 1) % NPANEL); const __bf16* W = &wl[(Q) & 1][0]; int lz = 0; asm volatile("" : "+v"(lz)); W += lz;
#define SAW_END(Q) lds_dma_wait(); __syncthreads(); SAW_MARK(5); }
        // ---- layer 2 (transposed): two channel blocks per panel as two interleaved accumulator chains
#pragma unroll
        for (int q = 0; q < NB2 / PB; ++q) {
            SAW_BEGIN(q)
            f32x16 a0, a1;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(bl + lz + 32 * (2 * q) + 8 * qq + 4 * h);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(bl + lz + 32 * (2 * q + 1) + 8 * qq + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) { a0[4 * qq + j] = b0[j]; a1[4 * qq + j] = b1[j]; }
            }
            // (requesting the fragments of k-step s + 1 in front of the MFMAs of k-step s was measured: no gain, 359 vs 352 us)
            bf16x8 f0h = wfrag_lds_u<NP>(W, KS, 0, 0, 0, lane), f1h = wfrag_lds_u<NP>(W, KS, 1, 0, 0, lane);
            bf16x8 f0l = wfrag_lds_u<NP>(W, KS, 0, 0, NP - 1, lane), f1l = wfrag_lds_u<NP>(W, KS, 1, 0, NP - 1, lane);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 w0h = f0h, w1h = f1h, w0l = f0l, w1l = f1l;
                if (s + 1 < KS) {
                    f0h = wfrag_lds_u<NP>(W, KS, 0, s + 1, 0, lane); f1h = wfrag_lds_u<NP>(W, KS, 1, s + 1, 0, lane);
                    f0l = wfrag_lds_u<NP>(W, KS, 0, s + 1, NP - 1, lane); f1l = wfrag_lds_u<NP>(W, KS, 1, s + 1, NP - 1, lane);
                }
                if constexpr (SPREAD) {
                    if (s < SST) {
#pragma unroll
                        for (int i = 0; i < PPS; ++i) fetch_piece((q + 1) % NPANEL, s * PPS + i);
                    }
                }

                if constexpr (NP == 2) {
                    a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0h, xl[s], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1h, xl[s], a1, 0, 0, 0);
                    a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0l, xh[s], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1l, xh[s], a1, 0, 0, 0);
                }
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0h, xh[s], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1h, xh[s], a1, 0, 0, 0);
                if (s % 2 == 1) __builtin_amdgcn_sched_barrier(0);
            }
            SAW_MARK(1);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float v[8], w[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { v[j] = reluf(a0[8 * s2 + j]); w[j] = reluf(a1[8 * s2 + j]); }
                split8u<NP>(v, zh[4 * q + s2], zl[4 * q + s2]);
                split8u<NP>(w, zh[4 * q + 2 + s2], zl[4 * q + 2 + s2]);
            }
            SAW_MARK(2);
            SAW_END(q)
        }
        // ---- layer 3 (direct orientation): max over the 32 rows
        if (rd + rstep < rend) request_u(item_of(rd + rstep) / S);
#pragma unroll
        for (int q = 0; q < NB3 / PB; ++q) {
            SAW_BEGIN(NB2 / PB + q)
            f32x16 a0, a1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { a0[r] = 0.0f; a1[r] = 0.0f; }
            bf16x8 f0h = wfrag_lds_u<NP>(W, KS, 0, 0, 0, lane), f1h = wfrag_lds_u<NP>(W, KS, 1, 0, 0, lane);
            bf16x8 f0l = wfrag_lds_u<NP>(W, KS, 0, 0, NP - 1, lane), f1l = wfrag_lds_u<NP>(W, KS, 1, 0, NP - 1, lane);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 w0h = f0h, w1h = f1h, w0l = f0l, w1l = f1l;
                if (s + 1 < KS) {
                    f0h = wfrag_lds_u<NP>(W, KS, 0, s + 1, 0, lane); f1h = wfrag_lds_u<NP>(W, KS, 1, s + 1, 0, lane);
                    f0l = wfrag_lds_u<NP>(W, KS, 0, s + 1, NP - 1, lane); f1l = wfrag_lds_u<NP>(W, KS, 1, s + 1, NP - 1, lane);
                }
                if constexpr (SPREAD) {
                    if (s < SST) {
#pragma unroll
                        for (int i = 0; i < PPS; ++i) fetch_piece((NB2 / PB + q + 1) % NPANEL, s * PPS + i);
                    }
                }

                if constexpr (NP == 2) {
                    a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zl[s], w0h, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zl[s], w1h, a1, 0, 0, 0);
                    a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zh[s], w0l, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zh[s], w1l, a1, 0, 0, 0);
                }
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zh[s], w0h, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zh[s], w1h, a1, 0, 0, 0);
                if (s % 2 == 1) __builtin_amdgcn_sched_barrier(0);
            }
            SAW_MARK(3);
            if constexpr (PAIR) {
                // half-wave 0 stores A's row, half-wave 1 B's (unpaired: one maximum, both half-waves store the same row)
                const float m0 = max_split(a0, own, paired), m1 = max_split(a1, own, paired);
                out[orow * C3 + (2 * q) * 32 + p] = reluf(m0 + bl[lz + C2 + (2 * q) * 32 + p]);
                out[orow * C3 + (2 * q + 1) * 32 + p] = reluf(m1 + bl[lz + C2 + (2 * q + 1) * 32 + p]);
            } else {
                float m0 = a0[0], m1 = a1[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) { m0 = fmaxf(m0, a0[r]); m1 = fmaxf(m1, a1[r]); }
                m0 = fmaxf(m0, __shfl_xor(m0, 32));
                m1 = fmaxf(m1, __shfl_xor(m1, 32));
                if (lane < 32) {
                    out[(size_t)cent * C3 + (2 * q) * 32 + lane] = reluf(m0 + bl[lz + C2 + (2 * q) * 32 + lane]);
                    out[(size_t)cent * C3 + (2 * q + 1) * 32 + lane] = reluf(m1 + bl[lz + C2 + (2 * q + 1) * 32 + lane]);
                }
            }
            SAW_MARK(4);
            SAW_END(NB2 / PB + q)
        }
#undef SAW_BEGIN
#undef SAW_END
#ifdef LSDM_PHASE_TIMING
        ++nrounds;
#endif
    }
    SAW_FLUSH(nrounds);
}

// ---------------------------------------------------------------------------------------------
// FP stage with the interpolated operand PRE-MULTIPLIED at the coarse level (fp2, fp3).  The first FP layer is linear in
// [p1 || sum_j w_j p2_j] (pointnet2_utils.py:299-312), so with W = [Wa | Wb]:  W x [...] + b = Wa x p1 + sum_j w_j (Wb x p2_j) + b.
// Y = p2 x Wb^T is evaluated once per COARSE point (lin_u_kernel<.., XYZ = false>, S = N/4 rows); here the three Y rows of a
// fine point are gathered and interpolated cooperatively (coalesced 16-byte pieces, fp32, no bf16 split) into an LDS tile
// from which every wave initialises its accumulators; the matrix pipe then only contracts over the D1 columns of p1.
// One LDS region is reused in sequence: interpolated tile (fp32) -> p1 operand tile -> layer-1 activations.
// SL > 0 (fp3, fp4: a cloud has 64 / 16 coarse rows, 64 / 16 KB): no gather and no tile -- the cloud's Y rows land in LDS whole (round 6; phase
// table before / after in profiles/r06_fp_pre_phase.txt: requests + interpolation 8.7 -> 5.4 us of a work-group's 32.7 -> 30.9, fp3 169 -> 161 us).
// FUSE: the stage's output feeds ONLY the next finer stage's interpolation (fp3 -> fp2), whose first layer is pre-multiplied at THIS
// level: ynext = out x Wb_next^T is one more layer on the rows that are already on chip -- no separate row-wise GEMM launch, no
// re-read of `out` (which is then stored only when keep_out: the parity taps of the stage entry points).
// TM_ x NW_: 64 rows x 4 waves, two work-groups per CU; or 128 rows x 8 waves, one per CU (half the weight stream per row).
#ifdef LSDM_PHASE_TIMING
// per-phase s_memtime sums of wave 0 of every work-group (tools/fppre_phase.py), by D1 (64: fp2's reference form, 128: fp3, 256: fp4): 3 x 8 slots --
// 0 requests + interpolation -> tile, 1 accumulators <- tile, 2 p1 tile + layer-1 k loops, 3 layer-1 epilogue, 4 layer-2 k loops, 5 layer-2
// epilogue -> next tile, 6 extra layer (or the plain layer 2 of the unfused form) + stores, 7 = work-groups
__device__ unsigned long long g_fppre_phase[64 * 24];
#define FPP_T0() unsigned long long pt_ = __builtin_amdgcn_s_memtime(); unsigned long long pd_[7] = {}
#define FPP_MARK(slot) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); pd_[slot] += n_ - pt_; pt_ = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define FPP_FLUSH(D1V) do { if (threadIdx.x == 0) { unsigned long long* g_ = g_fppre_phase + (blockIdx.x & 63) * 24 + ((D1V) == 64 ? 0 : ((D1V) == 128 ? 8 : 16)); \
        for (int i_ = 0; i_ < 7; ++i_) atomicAdd(g_ + i_, pd_[i_]); atomicAdd(g_ + 7, 1ull); } } while (0)
extern "C" __attribute__((visibility("default"))) int lsdm_debug_fppre_phase(unsigned long long* out24, int reset) {
    static unsigned long long h[64 * 24], z[64 * 24];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fppre_phase), sizeof h) != hipSuccess) return -1;
    for (int s = 0; s < 24; ++s) { out24[s] = 0; for (int c = 0; c < 64; ++c) out24[s] += h[c * 24 + s]; }
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_fppre_phase), z, sizeof z) != hipSuccess) return -1;
    return 0;
}
#else
#define FPP_T0() do {} while (0)
#define FPP_MARK(slot) do {} while (0)
#define FPP_FLUSH(D1V) do {} while (0)
#endif
template <int D1, int C1, int C2, bool FUSE = false, int TM_ = 64, int NW_ = 4, int NP_ = 2, int SL = 0>
__global__ __launch_bounds__(NW_ * 64, TM_ == 64 ? NW_ / 2 : 1) void fp_pre_kernel(const float* __restrict__ points1, const int* __restrict__ p1_remap,
                                                        const float* __restrict__ y, const int* __restrict__ nn_idx,
                                                        const float* __restrict__ nn_w, int N, int S, DenseH l1, DenseH l2,
                                                        float* __restrict__ out, DenseH l3, float* __restrict__ ynext, int keep_out, int p1_mod) {
    constexpr int TM = TM_, NW = NW_, NP = NP_, NT = NW * 64;
    constexpr int TPP = NT / TM;                               // threads per fine point in the interpolation phase
    static_assert(TPP == 4 || TPP == 8, "four or eight threads per fine point in the interpolation phase");
    // fp32 tile row: C1 floats, no padding (a row is a whole number of bank rounds), and the quad index of an access is XORed with
    // tile_swz(row) = row bit 0 -> quad bit 0, row bit 1 -> quad bit 3, row bits 2-3 -> quad bits 1-2.  The 16-lane phases of the
    // interpolation stores (rows r .. r+3 x quads {0, 2, 4, 6} + const, see FP_CH) and of the accumulator loads (16 consecutive rows x one
    // quad) then both cover 16 distinct bank quads.
    constexpr int LDT = C1;
    static_assert(C1 % 64 == 0, "tile rows: whole bank rounds");
    auto tile_swz = [](int row) -> int { return (((row & 1) | ((row & 2) << 2) | (((row >> 2) & 3) << 1)) << 2); };
    constexpr int LDX = ldh(D1), PX = TM * LDX, LDB = ldh(C1), PB = TM * LDB, LDC = ldh(C2), PC = TM * LDC;
    // SL > 0 (fp3: 64, fp4: 16 coarse points per cloud): ALL the cloud's Y rows land in LDS once, by LDS-DMA (one 1 KB row per wave instruction, no
    // index in front of the request: they leave with the index loads), and the accumulators are interpolated straight out of them as in
    // fp_d_kernel -- no gathered 192 KB per tile, no interpolated tile.  Same three rows, same interp3, same order: bit-identical.
    constexpr int RS = 260;                                    // floats per landed row (1 KB + 16 bytes: consecutive rows start four banks apart)
    static_assert(SL == 0 || C1 == 256, "a landed row is one 1 KB wave copy");
    constexpr int BYTES = imax(imax(imax(imax(SL > 0 ? 0 : TM * LDT * 4, SL * RS * 4), NP * PX * 2), NP * PB * 2), FUSE ? NP * PC * 2 : 0);
    __shared__ __attribute__((aligned(16))) unsigned char region[BYTES];
    float* tile = reinterpret_cast<float*>(region);
    __bf16* bufX = reinterpret_cast<__bf16*>(region);
    __bf16* bufB = reinterpret_cast<__bf16*>(region);
    int c, tl;
    xcd_cloud_tile(N / TM, c, tl);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int p = lane & 31, h = lane >> 5;
    const int n0 = tl * TM;
    const float* p1 = points1 + ((size_t)c * N + n0) * D1;
    const float* yb = y + (size_t)c * S * C1;
    // the p1 tile is requested first (consumed in phase C); p1_remap (optional): row i of the cloud is points1[p1_remap[i]]
    // (loop-invariant schedule: the level-1 features stay in SOURCE order, the step's FPS order is the index list)
    constexpr int XIT = TM * (D1 / 4) / NT;
    f32x4 gx[XIT];
    // Every INDEX the tile needs is requested before any row that depends on one: the p1 row numbers (p1_remap) together with the
    // three neighbour indices and weights of phase A -- one L2 round trip for all of them, then the p1 rows and the Y rows go out
    // back to back.  (With the p1 rows fetched through their indices first, phase A's indices waited a whole round trip behind them:
    // fp2 333 us in the loop-invariant schedule against 303 in the strict one, whose p1 rows are contiguous.)
    FPP_T0();
    if constexpr (SL > 0) {
        const unsigned dst0 = lds_addr(tile);
        for (int r = wave; r < SL; r += NW) lds_dma16(yb + (size_t)r * C1 + 4 * lane, __builtin_amdgcn_readfirstlane(dst0 + (unsigned)r * (RS * 4)));
    }
    const int arow = tid / TPP, sub = tid % TPP;
    const size_t ao = ((size_t)c * N + n0 + (SL > 0 ? p : arow)) * 3;    // (SL: the lane's point of row block 0; block rb is 96 words further)
    int xrow[XIT];
    if (p1_remap) {
#pragma unroll
        for (int it = 0; it < XIT; ++it) xrow[it] = p1_remap[(size_t)c * N + n0 + (tid + it * NT) / (D1 / 4)];
    }
    constexpr int NRB = SL > 0 ? TM / 32 : 1;                 // (SL: the three rows and weights of the lane's point in every row block)
    int ni[NRB][3];
    float nw[NRB][3];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int j = 0; j < 3; ++j) { ni[rb][j] = nn_idx[ao + 96 * rb + j]; nw[rb][j] = nn_w[ao + 96 * rb + j]; }
    // SL: ONE round trip in front of the accumulators -- the landed rows and every index; the p1 rows (a second, dependent one through
    // p1_remap) are requested behind that wait and fly under phase B (`late` pins their address arithmetic behind the wait)
    int late = 0;
    if constexpr (SL > 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(late) : : "memory");
    if (p1_remap) {
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int e = tid + it * NT;
            const int row = e / (D1 / 4), c4 = e - row * (D1 / 4);
            // (p1_mod > 0: a launch over the clouds of several steps; points1 is the per-CALL tensor in source order, cloud c % p1_mod)
            gx[it] = load4f(points1 + ((size_t)(p1_mod > 0 ? c % p1_mod : c) * N + (xrow[it] + late)) * D1 + 4 * c4);
        }
    } else {
#pragma unroll
        for (int it = 0; it < XIT; ++it) gx[it] = load4f(p1 + (size_t)(tid + it * NT) * 4 + late);
    }
    // ---- phase A: interpolated Y rows -> fp32 tile.  Four threads per fine point, 16 pieces (of 4 channels) each ------------
    if constexpr (SL == 0) {
        const int row = arow;
        const int i0 = ni[0][0], i1 = ni[0][1], i2 = ni[0][2];
        const float w0 = nw[0][0], w1 = nw[0][1], w2 = nw[0][2];
        const float* r0 = yb + (size_t)i0 * C1;
        const float* r1 = yb + (size_t)i1 * C1;
        const float* r2 = yb + (size_t)i2 * C1;
        constexpr int PIECES = C1 / (4 * TPP);               // per thread
        // piece k of thread `sub` of a row: consecutive pieces of a thread are ADJACENT 16-byte quads in pairs, so that the TPP threads of
        // a row read one whole 128-byte line per pair of requests (with piece k at quad sub + TPP k every line was requested in two
        // separate iterations, far enough apart to be fetched from L2 twice)
#define FP_CH(sub, k) (4 * (2 * (sub) + ((k) & 1) + 2 * TPP * ((k) >> 1)))
        // GB pieces x 3 rows in flight per batch: the phase is a chain of L2 round trips, so fewer, larger batches (registers
        // permitting: 12 GB VGPRs) shorten it
        constexpr int GB = (PIECES >= 8 && TPP == 4) ? 8 : 4;
#pragma unroll 1
        for (int k0 = 0; k0 < PIECES; k0 += GB) {
            f32x4 a[GB], b[GB], d[GB];
#pragma unroll
            for (int k = 0; k < GB; ++k) {
                const int ch = FP_CH(sub, k0 + k);
                a[k] = load4f(r0 + ch); b[k] = load4f(r1 + ch); d[k] = load4f(r2 + ch);
            }
#pragma unroll
            for (int k = 0; k < GB; ++k) {
                const int ch = FP_CH(sub, k0 + k);
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = interp3(a[k][j], b[k][j], d[k][j], w0, w1, w2);
                *reinterpret_cast<f32x4*>(tile + row * LDT + (ch ^ tile_swz(row))) = v;
            }
        }
    }
    __syncthreads();
    FPP_MARK(0);
    // ---- phase B: accumulators start from the interpolated rows (transposed product: lane owns channels 8q + 4h + j of point p)
    using TS = SplitH<NW, TM, C1>;
    constexpr int NCB = TS::CB / TS::CSTEP;
    static_assert(TS::WPC == 1, "split");
    constexpr int RB1 = TS::RB;
    const int cb0 = wave % TS::CSTEP;
    f32x16 acc[NCB][RB1];
    if constexpr (SL > 0) {
        static_assert(SL == 0 || RB1 == NRB, "one set of neighbours per row block of the wave");
#pragma unroll
        for (int rb = 0; rb < RB1; ++rb) {
            const float *r0 = tile + ni[rb][0] * RS + 4 * h, *r1 = tile + ni[rb][1] * RS + 4 * h, *r2 = tile + ni[rb][2] * RS + 4 * h;
#pragma unroll
            for (int q = 0; q < NCB; ++q)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int ch = (cb0 + q * TS::CSTEP) * 32 + 8 * qd;
                    const f32x4 a = *reinterpret_cast<const f32x4*>(r0 + ch), b = *reinterpret_cast<const f32x4*>(r1 + ch), d = *reinterpret_cast<const f32x4*>(r2 + ch);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[q][rb][4 * qd + j] = interp3(a[j], b[j], d[j], nw[rb][0], nw[rb][1], nw[rb][2]);
                }
        }
    } else {
#pragma unroll
        for (int q = 0; q < NCB; ++q)
#pragma unroll
            for (int rb = 0; rb < RB1; ++rb)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(tile + (rb * 32 + p) * LDT + (((cb0 + q * TS::CSTEP) * 32 + 8 * qd + 4 * h) ^ tile_swz(p)));
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[q][rb][4 * qd + j] = t[j];
                }
    }
    f32x4 bv1[NCB][4];
#pragma unroll
    for (int q = 0; q < NCB; ++q) load_bias_h(bv1[q], l1.b, cb0 + q * TS::CSTEP, lane);
    __syncthreads();
    FPP_MARK(1);
    // ---- phase C: p1 operand tile, K = D1 -----------------------------------------------------------------------------------
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
        const int e = tid + it * NT;
        const int row = e / (D1 / 4), c4 = e - row * (D1 / 4);
        store_split4<NP>(bufX + row * LDX + 4 * c4, PX, gx[it][0], gx[it][1], gx[it][2], gx[it][3]);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NCB; ++q)
        mma_kloop_h<RB1, NP, true, D1 / 16>(acc[q], bufX, LDX, PX, l1.w, C1, cb0 + q * TS::CSTEP, 0, lane);
    __syncthreads();
    FPP_MARK(2);
    // ---- phase D: layer-1 epilogue -> bufB, layer 2 (direct orientation) -> global -----------------------------------------------
#pragma unroll
    for (int q = 0; q < NCB; ++q) store_lds_h<ACT_RELU, RB1, NP>(acc[q], 0, cb0 + q * TS::CSTEP, bv1[q], bufB, LDB, PB, lane);
    __syncthreads();
    FPP_MARK(3);
    using T2 = SplitH<NW, TM, C2>;
    const int cb2 = wave % T2::CSTEP, rg2 = wave / T2::CSTEP;
    if constexpr (!FUSE) {
#pragma unroll 1
        for (int cb = cb2; cb < T2::CB; cb += T2::CSTEP) {
            f32x16 a2[T2::RB];
            acc_zero_h(a2);
            const int col = cb * 32 + (lane & 31);
            const float bv = l2.b[col];
            mma_kloop_h<T2::RB, NP, false, C1 / 16>(a2, bufB + rg2 * T2::RB * 32 * LDB, LDB, PB, l2.w, C2, cb, 0, lane);
#pragma unroll
            for (int rb = 0; rb < T2::RB; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (rg2 * T2::RB + rb) * 32 + acc_row(r, lane);
                    out[((size_t)c * N + n0 + row) * C2 + col] = reluf(a2[rb][r] + bv);
                }
        }
    } else {
        // layer 2 transposed, every block of the wave kept in accumulators until all waves are done with bufB; its ReLU output
        // becomes the operand tile of the extra layer in the same region
        static_assert(!FUSE || T2::WPC == 1, "split");
        constexpr int NCB2 = T2::CB / T2::CSTEP, RB2 = T2::RB;
        __bf16* bufC = reinterpret_cast<__bf16*>(region);
        f32x16 a2[NCB2][RB2];
#pragma unroll
        for (int q = 0; q < NCB2; ++q) {
            f32x4 bv2[4];
            load_bias_h(bv2, l2.b, cb2 + q * T2::CSTEP, lane);
            acc_bias_h(a2[q], bv2);
            mma_kloop_h<RB2, NP, true, C1 / 16>(a2[q], bufB, LDB, PB, l2.w, C2, cb2 + q * T2::CSTEP, 0, lane);
        }
        __syncthreads();
        FPP_MARK(4);
#pragma unroll
        for (int q = 0; q < NCB2; ++q) {
            const int cb = cb2 + q * T2::CSTEP;
            if (keep_out) {
#pragma unroll
                for (int rb = 0; rb < RB2; ++rb)
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        const f32x4 v = {reluf(a2[q][rb][4 * qd]), reluf(a2[q][rb][4 * qd + 1]), reluf(a2[q][rb][4 * qd + 2]), reluf(a2[q][rb][4 * qd + 3])};
                        *reinterpret_cast<f32x4*>(out + ((size_t)c * N + n0 + rb * 32 + p) * C2 + cb * 32 + 8 * qd + 4 * h) = v;
                    }
            }
            store_lds_b_h<ACT_RELU, RB2, NP>(a2[q], 0, cb, bufC, LDC, PC, lane);
        }
        __syncthreads();
        FPP_MARK(5);
        using T3 = SplitH<NW, TM, 256>;
        const int cb3 = wave % T3::CSTEP, rg3 = wave / T3::CSTEP;
#pragma unroll 1
        for (int cb = cb3; cb < T3::CB; cb += T3::CSTEP) {
            f32x16 a3[T3::RB];
            acc_zero_h(a3);
            const int col = cb * 32 + (lane & 31);
            mma_kloop_h<T3::RB, NP, false, C2 / 16>(a3, bufC + rg3 * T3::RB * 32 * LDC, LDC, PC, l3.w, 256, cb, 0, lane);
#pragma unroll
            for (int rb = 0; rb < T3::RB; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (rg3 * T3::RB + rb) * 32 + acc_row(r, lane);
                    ynext[((size_t)c * N + n0 + row) * 256 + col] = a3[rb][r];
                }
        }
    }
    FPP_MARK(6);
    FPP_FLUSH(D1);
}

// Every launcher below takes np, the number of bf16 planes of the operands and of the weight blobs handed in: 2 = bf16x3 (three matrix
// instructions per product), 1 = plain bf16 (one).  Level features, U / V and Y are fp32 in HBM in both.
#define LSDM_NP(np, ...) do { if ((np) == 2) { constexpr int NPK = 2; __VA_ARGS__; } else { constexpr int NPK = 1; __VA_ARGS__; } } while (0)

// LSDM_FP2_WS: fp2 on 4 = the two-stage persistent kernel over Morton-ordered tiles with deduplicated rows (fp_d_kernel, encoder_d.hip; the default),
// 0 = the tile-per-work-group form (fp_pre_kernel: the reference form of the stage for the bit-equality tests).  Same results bit for bit.
int fp2_ws() {
    const char* e = getenv("LSDM_FP2_WS");      // (read per launch: the A/B tests flip it inside one process)
    return e ? (atoi(e) ? 4 : 0) : 4;
}

// which = 2: fp2 (1024 fine <- 256 coarse points per cloud, D1 = 64); 3: fp3 (256 <- 64, D1 = 128); 4: fp4 (64 <- 16, D1 = 256).
// ybuf: the stage's pre-multiplied operand Y = coarse output x Wb^T -- written by the coarser stage's launch (which = 2, 3: y_ready) or here from
// fpo_coarse (which = 4).  wb_next / ynext (which = 3, 4): the next finer stage's Y = out x wb_next^T as one more layer of this launch; `out` itself is
// then stored only when keep_out.
void launch_fp_pre(int np, int which, int nc, const float* points1, const int* p1_remap, const float* fpo_coarse, DenseH wb, float* ybuf,
                   const int* nn_idx, const float* nn_w, DenseH l1a, DenseH l2, float* out, bool y_ready, DenseH wb_next, float* ynext,
                   bool keep_out, hipStream_t st, int p1_mod, const int* plan) {
    const int S = which == 2 ? 256 : (which == 3 ? 64 : 16);
    const DenseH none{nullptr, nullptr};
    if (which == 4) {
        // fp4: 64 fine <- 16 coarse points per cloud, D1 = 256, the coarse features are 512 wide
        LSDM_NP(np, hipLaunchKernelGGL((lin_u_kernel<512, 256, false, NPK>), dim3((nc * 16 + 63) / 64), dim3(256), 0, st, fpo_coarse, nullptr, nc * 16, wb, nullptr, ybuf, nullptr));
        LSDM_NP(np, hipLaunchKernelGGL((fp_pre_kernel<256, 256, 256, true, 64, 4, NPK, 16>), dim3(64 / 64 * nc), dim3(256), 0, st, points1, p1_remap, ybuf, nn_idx, nn_w, 64, 16,
                                       l1a, l2, out, wb_next, ynext, keep_out ? 1 : 0, p1_mod));
        return;
    }
    (void)y_ready; (void)S;
    if (which == 2 && plan) {
        launch_fp_d(np, nc, points1, ybuf, nn_w, l1a, l2, out, plan, st, p1_mod);
    }
    else if (which == 2) {
        LSDM_NP(np, hipLaunchKernelGGL((fp_pre_kernel<64, 256, 128, false, 64, 4, NPK>), dim3(1024 / 64 * nc), dim3(256), 0, st, points1, p1_remap, ybuf, nn_idx, nn_w, 1024, 256,
                                       l1a, l2, out, none, nullptr, 1, p1_mod));
    }
    else {
        // (128-row tiles on one 8-wave work-group per CU were measured slower, 175 vs 156 us: two independent work-groups per CU cover
        // each other's gather and barrier phases; 64 rows on EIGHT waves, four per SIMD at 128 registers: 240 vs 169 us, fp4 98 vs 54 --
        // profiles/r06_fp_pre_phase.txt)
        LSDM_NP(np, hipLaunchKernelGGL((fp_pre_kernel<128, 256, 256, true, 64, 4, NPK, 64>), dim3(256 / 64 * nc), dim3(256), 0, st, points1, p1_remap, ybuf, nn_idx, nn_w, 256, 64,
                                       l1a, l2, out, wb_next, ynext, keep_out ? 1 : 0, p1_mod));
    }
}

// ---------------------------------------------------------------------------------------------
void launch_lin_u(int np, int level, int R, const float* X, const float* xyz, DenseH W, const float* wx, float* U, float* V, hipStream_t st) {
    const dim3 grid((R + 63) / 64), block(256);
    if (level == 1) LSDM_NP(np, hipLaunchKernelGGL((lin_u_kernel<64, 64, true, NPK>), grid, block, 0, st, X, xyz, R, W, wx, U, V));
    else if (level == 2) LSDM_NP(np, hipLaunchKernelGGL((lin_u_kernel<128, 128, true, NPK>), grid, block, 0, st, X, xyz, R, W, wx, U, V));
    else LSDM_NP(np, hipLaunchKernelGGL((lin_u_kernel<256, 256, true, NPK>), grid, block, 0, st, X, xyz, R, W, wx, U, V));
}

// sa2 (the level-1 kernel) over a LIST of centroids: S centroids per cloud in the numbering of the list, list_count on the device
void launch_sa_u_list(int np, int S, const float* U, const float* V, const int* cent_idx, const int* ball, const int* remap, const __bf16* blob_w,
                      const float* blob_b, float* out, hipStream_t st, int src_mod, const int* list, const int* list_count) {
    LSDM_NP(np, hipLaunchKernelGGL((sa_ur_kernel<64, 64, 128, 8, NPK>), dim3(512), dim3(512), 0, st, U, V, cent_idx, ball, remap, 1024, S, 0LL, blob_w, blob_b, out, src_mod,
                                   list, list_count, nullptr));
}

void launch_sa_u(int np, int level, int nc, const float* U, const float* V, const int* cent_idx, const int* ball, const int* remap,
                 const __bf16* blob_w, const float* blob_b, float* out, hipStream_t st, int src_mod) {
    if (level == 3) {
        LSDM_NP(np, hipLaunchKernelGGL((sa_w_kernel<NPK>), dim3(256), dim3(256), 0, st, U, V, cent_idx, ball, (long long)nc * 16, blob_w, blob_b, out, nullptr, nullptr));
    } else if (level == 1) {
        const long long tot = (long long)nc * 256;
        // (two work-groups per CU = four waves per SIMD; three per CU measured the same, 305 vs 307 us, one per CU 325: the kernel is
        // issue-bound -- MFMA and VALU time add -- not latency-bound)
        LSDM_NP(np, hipLaunchKernelGGL((sa_ur_kernel<64, 64, 128, 8, NPK>), dim3(512), dim3(512), 0, st, U, V, cent_idx, ball, remap, 1024, 256, tot, blob_w, blob_b, out, src_mod,
                                       nullptr, nullptr, nullptr));
    } else if (level == 2) {
        const long long tot = (long long)nc * 64;
        LSDM_NP(np, hipLaunchKernelGGL((sa_p_kernel<128, 128, 256, 12, NPK>), dim3(256), dim3(12 * 64), 0, st, U, V, cent_idx, ball, 256, 64, tot, blob_w, blob_b, out, nullptr,
                                       nullptr));
    }
}

// sa2 (level 1: the memo's pass over the source points) / sa3 / sa4 over the item list of launch_sa_pair_plan (geometry.hip): rows = 32 U rows per
// item, meta = {V rows, output rows}, *count items
void launch_sa_u_items(int np, int level, const float* U, const float* V, const int* rows, const int4* meta, const int* count, const __bf16* blob_w,
                       const float* blob_b, float* out, hipStream_t st) {
    if (level == 1)
        LSDM_NP(np, hipLaunchKernelGGL((sa_ur_kernel<64, 64, 128, 8, NPK, true>), dim3(512), dim3(512), 0, st, U, V, nullptr, rows, nullptr, 1024, 1024, 0LL, blob_w, blob_b, out, 0,
                                       nullptr, count, meta));
    else if (level == 3)
        LSDM_NP(np, hipLaunchKernelGGL((sa_w_kernel<NPK, true>), dim3(256), dim3(256), 0, st, U, V, nullptr, rows, 0LL, blob_w, blob_b, out, meta, count));
    else
        LSDM_NP(np, hipLaunchKernelGGL((sa_p_kernel<128, 128, 256, 12, NPK, true>), dim3(256), dim3(12 * 64), 0, st, U, V, nullptr, rows, 256, 64, 0LL, blob_w, blob_b, out, meta,
                                       count));
}

}  // namespace lsdm
