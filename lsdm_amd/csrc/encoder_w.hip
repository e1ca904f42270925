// fp3 / fp4 (bf16x3): feature propagation with the interpolated operand pre-multiplied at the coarse level, the stage's second layer and
// the NEXT finer stage's pre-multiplication (pointnet2_utils.py:273-312, pointnet2.py:51-52,71-72) as ONE register-chained stack:
//
//     acc  <- sum_j w_j Y[i_j]              Y = p2 x Wb^T, once per COARSE point (lin_u_kernel / the coarser stage's fused extra layer)
//     a1   =  ReLU(acc + Wa x p1 + b1)      K = D1 (128: fp3, 256: fp4), 256 channels
//     a2   =  ReLU(W2 x a1 + b2)            256 -> 256                               (= the stage's output, stored only for the parity taps)
//     ynext=  a2 x Wb_next^T                256 -> 256, the next stage's Y on THIS stage's rows (no bias, no activation)
//
// Design (the one sa_w_kernel uses for sa4; round 6, replaces fp_pre_kernel<D1, 256, 256, FUSE> on these two stages): a persistent work-group
// of four waves, ONE wave per SIMD with the whole 512-entry register file.  A wave carries 32 points through the three layers in registers --
// the accumulators of a transposed block ARE the next layer's B fragments (bf16_core.hpp; the host packs layers 2 / 3 in that k order,
// api.hip::build_chain_blob) -- and the weights of all layers stream through LDS as twelve panels of two channel blocks (32 / 64 KB) per
// round, copied by LDS-DMA from inside the previous panel's k loop and shared by the four waves: 640 / 768 KB of fragments per 128 rows, where
// fp_pre_kernel pulled the same fragments through the vector L1 per 64 rows and per work-group (43 B/clk of the 64 B/clk path with two
// work-groups per CU: PMC wait_inst 63 %, MFMA pipe 33.7 % busy, 5 barriers of dependent phases per tile).  The two channel blocks of a
// panel are two independent accumulator chains.  What a round needs from memory is requested a phase ahead: the interpolation rows of panel
// q + 1 (24 x 16 bytes per lane) before panel q's k loop, the next round's p1 rows and neighbour lists during this round's last layer.
// The three Y rows of a point are interpolated in the lane that owns the accumulator (interp3: the same rounding sequence as every other FP
// stage); the k-steps of layers 2 / 3 hold the same 16 channels as fp_pre_kernel's, in the accumulator order.
#include <utility>

#include "bf16_core.hpp"
#include "kernels.hpp"

namespace lsdm {

namespace {

template <class F, int... Is>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ bf16x8 wfrag(const __bf16* base, int KS, int nb, int s, int plane, int lane) {
    return *reinterpret_cast<const bf16x8*>(base + ((size_t)((nb * KS + s) * 2 + plane) * 64 + lane) * 8);
}

}  // namespace

#ifdef LSDM_PHASE_TIMING
__device__ unsigned long long g_fpw_phase[64 * 8];   // 64 spread copies: 0 prologue, 1 layer-1 init, 2 k loops, 3 epilogues, 4 stores, 5 panel wait + barrier, 7 rounds
#define FPW_T0() unsigned long long st_ = __builtin_amdgcn_s_memtime(); unsigned long long sd_[6] = {}
#define FPW_MARK(slot) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); sd_[slot] += n_ - st_; st_ = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define FPW_FLUSH(nr) do { if (lane == 0 && wave == 0) { unsigned long long* g_ = g_fpw_phase + (blockIdx.x & 63) * 8; \
        for (int i_ = 0; i_ < 6; ++i_) atomicAdd(g_ + i_, sd_[i_]); atomicAdd(g_ + 7, (unsigned long long)(nr)); } } while (0)
extern "C" __attribute__((visibility("default"))) int lsdm_debug_fpw_phase(unsigned long long* out8, int reset) {
    static unsigned long long h[64 * 8], z[64 * 8];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fpw_phase), sizeof h) != hipSuccess) return -1;
    for (int s = 0; s < 8; ++s) { out8[s] = 0; for (int c = 0; c < 64; ++c) out8[s] += h[c * 8 + s]; }
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_fpw_phase), z, sizeof z) != hipSuccess) return -1;
    return 0;
}
#else
#define FPW_T0() do {} while (0)
#define FPW_MARK(slot) do {} while (0)
#define FPW_FLUSH(nr) do {} while (0)
#endif

// points1 (rows, D1), y (clouds, S, 256), nn_idx / nn_w (rows, 3): rows = clouds x N fine points, 32-row blocks never straddle a cloud
// (N = 256 / 64).  blob: [layer 1: 8 x D1/16 | layer 2: 8 x 16 | layer 3: 8 x 16] channel block x k-step fragments of [plane][lane][8] bf16,
// biases [b1 (256) | b2 (256)].  out (optional): the stage's output rows; ynext: the next stage's pre-multiplied operand.
template <int D1, bool KEEP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void fp_w_kernel(
    const float* __restrict__ points1, const float* __restrict__ y, const int* __restrict__ nn_idx, const float* __restrict__ nn_w, int N, int S,
    long long total_blocks, const __bf16* __restrict__ blob_w, const float* __restrict__ blob_b, float* __restrict__ out,
    float* __restrict__ ynext) {
    constexpr int C = 256, NW = 4, NT = NW * 64;
    constexpr int KS1 = D1 / 16, KS = C / 16, NB = C / 32, PB = 2, NPL = NB / PB;       // panels per layer
    constexpr int FP1 = PB * KS1 * 2, FP = PB * KS * 2;                                 // fragments (1 KiB) per panel: layer 1 / layers 2, 3
    constexpr int NPANEL = 3 * NPL;
    constexpr size_t L1E = (size_t)NB * KS1 * 2 * 512, L2E = (size_t)NB * KS * 2 * 512;   // bf16 elements of layer 1 / of layer 2
    static_assert(NPANEL % 2 == 0 && FP1 % NW == 0 && FP % NW == 0, "panel split");
    __shared__ __attribute__((aligned(16))) __bf16 wl[2][FP * 512];
    __shared__ __attribute__((aligned(16))) float bl[2 * C];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid; e < 2 * C; e += NT) bl[e] = blob_b[e];
    const int p = lane & 31, h = lane >> 5;
    // panel q of a round: layer q / NPL, channel blocks 2 (q % NPL), 2 (q % NPL) + 1
    auto panel_src = [&](int q) -> const __bf16* {
        return blob_w + (q < NPL ? (size_t)q * FP1 * 512 : (q < 2 * NPL ? L1E + (size_t)(q - NPL) * FP * 512 : L1E + L2E + (size_t)(q - 2 * NPL) * FP * 512));
    };
    // piece i of panel q (i < pieces(q) = its fragments / NW), issued from inside the k loop of the panel before it
    auto fetch_piece = [&](int q, int i) {
        const __bf16* src = opaque_ptr(panel_src(q));
        const int fr = wave + i * NW;
        lds_dma16(src + (size_t)fr * 512 + lane * 8, __builtin_amdgcn_readfirstlane(lds_addr(&wl[q & 1][(size_t)fr * 512])));
    };
    for (int i = 0; i < FP1 / NW; ++i) fetch_piece(0, i);
    lds_dma_wait();
    __syncthreads();

    const long long rounds = (total_blocks + NW - 1) / NW;      // (a wave past the end repeats the last block: same values, same addresses)
    // an even, contiguous share of the rounds per work-group, consecutive shares on one XCD
    const long long G = gridDim.x, bx = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const long long per = rounds / G, extra = rounds - per * G;
    const long long rbeg = bx * per + (bx < extra ? bx : extra), rend = rbeg + per + (bx < extra ? 1 : 0);
    auto block_of = [&](long long rd) { const long long b = rd * NW + wave; return b < total_blocks ? b : total_blocks - 1; };

    // requested a phase ahead: neighbour lists (top of the previous round), p1 rows (the previous round's layer 3), interpolation rows of
    // a panel (before the k loop of the panel in front of it)
    int i_n[3] = {0, 0, 0};
    float w_n[3] = {0.0f, 0.0f, 0.0f};
    f32x4 un0[KS1], un1[KS1];
    f32x4 R[PB][4][3];
    auto request_nn = [&](long long b) {
        const size_t o = ((size_t)b * 32 + p) * 3;
#pragma unroll
        for (int k = 0; k < 3; ++k) { i_n[k] = nn_idx[o + k]; w_n[k] = nn_w[o + k]; }
    };
    auto request_p1 = [&](long long b) {
        const float* row = points1 + ((size_t)b * 32 + p) * D1 + 8 * h;
#pragma unroll
        for (int s = 0; s < KS1; ++s) { un0[s] = load4f(row + 16 * s); un1[s] = load4f(row + 16 * s + 4); }
    };
    // rows of the Y tensor of block b's cloud, for the two channel blocks of layer-1 panel q, in the lane order of a transposed accumulator
    auto request_rows = [&](long long b, const int (&ii)[3], int q) {
        const float* yc = y + (size_t)((b * 32) / N) * S * C + 32 * (PB * q) + 4 * h;
#pragma unroll
        for (int u = 0; u < PB; ++u)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
#pragma unroll
                for (int k = 0; k < 3; ++k) R[u][qq][k] = load4f(yc + (size_t)ii[k] * C + 32 * u + 8 * qq);
    };
    // the p1 operand of a round as bf16 planes: split from the raw rows while the PREVIOUS round's last layer runs (the planes are MFMA
    // operands only and may live in the accumulator half of the register file; the raw rows, VALU inputs, may not)
    bf16x8 xh[KS1], xl[KS1];
    auto split_p1 = [&]() {
#pragma unroll
        for (int s = 0; s < KS1; ++s) {
            const float t[8] = {un0[s][0], un0[s][1], un0[s][2], un0[s][3], un1[s][0], un1[s][1], un1[s][2], un1[s][3]};
            split8_h<2>(t, xh[s], xl[s]);
        }
    };
    // the accumulators layer-1 panel q + 1 starts from (the interpolated rows), formed under the LAST k-steps of panel q -- the rows have had
    // most of a panel's time to arrive, and the 96 interpolation instructions issue in the shadow of the matrix instructions
    f32x16 n0, n1;
    auto interp_rows = [&](const float (&wt)[3]) {
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                n0[4 * qq + j] = interp3(R[0][qq][0][j], R[0][qq][1][j], R[0][qq][2][j], wt[0], wt[1], wt[2]);
                n1[4 * qq + j] = interp3(R[1][qq][0][j], R[1][qq][1][j], R[1][qq][2][j], wt[0], wt[1], wt[2]);
            }
    };
    if (rbeg < rend) {
        const long long b = block_of(rbeg);
        request_nn(b);
        request_p1(b);
        request_rows(b, i_n, 0);
#ifndef LSDM_FPW_SIMPLE
        split_p1();
        interp_rows(w_n);
#endif
    }
    FPW_T0();
    [[maybe_unused]] int nrounds = 0;
    for (long long rd = rbeg; rd < rend; ++rd) {
        const long long blk = block_of(rd);
        const size_t row0 = (size_t)blk * 32;
        const int ii[3] = {i_n[0], i_n[1], i_n[2]};
        const float ww[3] = {w_n[0], w_n[1], w_n[2]};
        bf16x8 ah[KS], al[KS], zh[KS], zl[KS];
#ifdef LSDM_FPW_SIMPLE       // (A/B: everything of a round formed at its use -- p1 split at the round's top, a panel's rows interpolated at its top)
        split_p1();
#endif
        const bool more = rd + 1 < rend;
        const long long nblk = more ? block_of(rd + 1) : blk;
        if (more) request_nn(nblk);
        FPW_MARK(0);
        // one panel: Q = its index in the round; the copy of panel Q + 1 is issued from inside this panel's k loop
#define FPW_BEGIN(Q) { const __bf16* W = &wl[(Q) & 1][0]; int lz = 0; asm volatile("" : "+v"(lz)); W += lz;
#define FPW_END(Q) lds_dma_wait(); __syncthreads(); FPW_MARK(5); }
#define FPW_KLOOP(Q, KSC, HS, HOOK, MMA)                                                                                                  \
            {                                                                                                                     \
                constexpr int NXT = ((Q) + 1) % NPANEL, PCS = (NXT < NPL ? FP1 : FP) / NW, PPS = PCS / ((KSC) / 2);                   \
                static_assert(PCS % ((KSC) / 2) == 0, "pieces per k-step");                                                       \
                bf16x8 f0h = wfrag(W, KSC, 0, 0, 0, lane), f1h = wfrag(W, KSC, 1, 0, 0, lane);                                    \
                bf16x8 f0l = wfrag(W, KSC, 0, 0, 1, lane), f1l = wfrag(W, KSC, 1, 0, 1, lane);                                    \
                _Pragma("unroll") for (int s = 0; s < (KSC); ++s) {                                                               \
                    const bf16x8 w0h = f0h, w1h = f1h, w0l = f0l, w1l = f1l;                                                      \
                    if (s + 1 < (KSC)) {                                                                                          \
                        f0h = wfrag(W, KSC, 0, s + 1, 0, lane); f1h = wfrag(W, KSC, 1, s + 1, 0, lane);                           \
                        f0l = wfrag(W, KSC, 0, s + 1, 1, lane); f1l = wfrag(W, KSC, 1, s + 1, 1, lane);                           \
                    }                                                                                                             \
                    if (s < (KSC) / 2) {                                                                                          \
                        _Pragma("unroll") for (int i = 0; i < PPS; ++i) fetch_piece(NXT, s * PPS + i);                            \
                    }                                                                                                             \
                    if (s == (HS)) { HOOK }                                                                                       \
                    MMA                                                                                                           \
                    if (s % 2 == 1) __builtin_amdgcn_sched_barrier(0);                                                            \
                }                                                                                                                 \
            }
        // ---- layer 1 (transposed): accumulators start from the interpolated Y rows, K = D1 over the p1 operand
        sfor<NPL>([&](auto q_) {
            constexpr int q = decltype(q_)::value;
            FPW_BEGIN(q)
#ifdef LSDM_FPW_SIMPLE
            interp_rows(ww);
#endif
            f32x16 a0 = n0, a1 = n1;
            if constexpr (q + 1 < NPL) request_rows(blk, ii, q + 1);
            FPW_MARK(1);
#ifdef LSDM_FPW_SIMPLE
            FPW_KLOOP(q, KS1, -1, ,
#else
            FPW_KLOOP(q, KS1, KS1 - 2, if constexpr (q + 1 < NPL) interp_rows(ww);,
#endif
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0h, xl[s], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1h, xl[s], a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0l, xh[s], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1l, xh[s], a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0h, xh[s], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1h, xh[s], a1, 0, 0, 0);)
            FPW_MARK(2);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const f32x4 c00 = *reinterpret_cast<const f32x4*>(bl + lz + 32 * (2 * q) + 16 * s2 + 4 * h);
                const f32x4 c01 = *reinterpret_cast<const f32x4*>(bl + lz + 32 * (2 * q) + 16 * s2 + 8 + 4 * h);
                const f32x4 c10 = *reinterpret_cast<const f32x4*>(bl + lz + 32 * (2 * q + 1) + 16 * s2 + 4 * h);
                const f32x4 c11 = *reinterpret_cast<const f32x4*>(bl + lz + 32 * (2 * q + 1) + 16 * s2 + 8 + 4 * h);
                float v[8], w[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = reluf(a0[8 * s2 + j] + c00[j]); v[4 + j] = reluf(a0[8 * s2 + 4 + j] + c01[j]);
                    w[j] = reluf(a1[8 * s2 + j] + c10[j]); w[4 + j] = reluf(a1[8 * s2 + 4 + j] + c11[j]);
                }
                split8_h<2>(v, ah[4 * q + s2], al[4 * q + s2]);
                split8_h<2>(w, ah[4 * q + 2 + s2], al[4 * q + 2 + s2]);
            }
            FPW_MARK(3);
            FPW_END(q)
        });
        // ---- layer 2 (transposed): accumulators start from the bias
        sfor<NPL>([&](auto q_) {
            constexpr int q = decltype(q_)::value;
            FPW_BEGIN(NPL + q)
            f32x16 a0, a1;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(bl + lz + C + 32 * (2 * q) + 8 * qq + 4 * h);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(bl + lz + C + 32 * (2 * q + 1) + 8 * qq + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) { a0[4 * qq + j] = b0[j]; a1[4 * qq + j] = b1[j]; }
            }
            FPW_KLOOP(NPL + q, KS, -1, ,
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0h, al[s], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1h, al[s], a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0l, ah[s], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1l, ah[s], a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0h, ah[s], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1h, ah[s], a1, 0, 0, 0);)
            FPW_MARK(2);
            if constexpr (KEEP) {       // the parity taps of the stage entry points: the stage's output rows
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const f32x4 v0 = {reluf(a0[4 * qq]), reluf(a0[4 * qq + 1]), reluf(a0[4 * qq + 2]), reluf(a0[4 * qq + 3])};
                    const f32x4 v1 = {reluf(a1[4 * qq]), reluf(a1[4 * qq + 1]), reluf(a1[4 * qq + 2]), reluf(a1[4 * qq + 3])};
                    *reinterpret_cast<f32x4*>(out + (row0 + p) * C + 32 * (2 * q) + 8 * qq + 4 * h) = v0;
                    *reinterpret_cast<f32x4*>(out + (row0 + p) * C + 32 * (2 * q + 1) + 8 * qq + 4 * h) = v1;
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float v[8], w[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { v[j] = reluf(a0[8 * s2 + j]); w[j] = reluf(a1[8 * s2 + j]); }
                split8_h<2>(v, zh[4 * q + s2], zl[4 * q + s2]);
                split8_h<2>(w, zh[4 * q + 2 + s2], zl[4 * q + 2 + s2]);
            }
            FPW_MARK(3);
            FPW_END(NPL + q)
        });
        // ---- layer 3 (direct orientation): the next stage's Y on these rows; the next round's operands are requested under it
        if (more) request_p1(nblk);
        sfor<NPL>([&](auto q_) {
            constexpr int q = decltype(q_)::value;
            FPW_BEGIN(2 * NPL + q)
            f32x16 a0, a1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { a0[r] = 0.0f; a1[r] = 0.0f; }
#ifdef LSDM_FPW_SIMPLE
            if constexpr (q == NPL - 1) {
                if (more) request_rows(nblk, i_n, 0);
            }
            FPW_KLOOP(2 * NPL + q, KS, -1, ,
#else
            if constexpr (q == NPL - 2) {
                if (more) { split_p1(); request_rows(nblk, i_n, 0); }
            }
            FPW_KLOOP(2 * NPL + q, KS, KS - 4, if constexpr (q == NPL - 1) { if (more) interp_rows(w_n); },
#endif
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zl[s], w0h, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zl[s], w1h, a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zh[s], w0l, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zh[s], w1l, a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zh[s], w0h, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zh[s], w1h, a1, 0, 0, 0);)
            FPW_MARK(2);
            {
                // (uniform base + one 32-bit lane offset: the 32 stores of a panel differ by instruction offsets and three adds)
                float* const ob = ynext + row0 * C + 32 * (2 * q);
                const int vo = (4 * h) * C + p;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    ob[vo + ((r & 3) + 8 * (r >> 2)) * C] = a0[r];
                    ob[vo + ((r & 3) + 8 * (r >> 2)) * C + 32] = a1[r];
                }
            }
            FPW_MARK(4);
            FPW_END(2 * NPL + q)
        });
#undef FPW_BEGIN
#undef FPW_END
#undef FPW_KLOOP
#ifdef LSDM_PHASE_TIMING
        ++nrounds;
#endif
    }
    FPW_FLUSH(nrounds);
}

// which = 3: fp3 (256 fine <- 64 coarse points per cloud, D1 = 128); which = 4: fp4 (64 <- 16, D1 = 256).  y: the stage's pre-multiplied
// operand (clouds, S, 256); out: the stage's output rows or NULL; ynext: the next finer stage's pre-multiplied operand (rows, 256).
void launch_fp_w(int which, int nc, const float* points1, const float* y, const int* nn_idx, const float* nn_w, const __bf16* blob_w, const float* blob_b,
                 float* out, float* ynext, int num_cus, hipStream_t st) {
    const int N = which == 3 ? 256 : 64, S = N / 4;
    const long long blocks = (long long)nc * (N / 32), rounds = (blocks + 3) / 4;
    const int grid = (int)(rounds < num_cus ? rounds : num_cus);
    if (which == 3 && out)
        hipLaunchKernelGGL((fp_w_kernel<128, true>), dim3(grid), dim3(256), 0, st, points1, y, nn_idx, nn_w, N, S, blocks, blob_w, blob_b, out, ynext);
    else if (which == 3)
        hipLaunchKernelGGL((fp_w_kernel<128, false>), dim3(grid), dim3(256), 0, st, points1, y, nn_idx, nn_w, N, S, blocks, blob_w, blob_b, out, ynext);
    else if (out)
        hipLaunchKernelGGL((fp_w_kernel<256, true>), dim3(grid), dim3(256), 0, st, points1, y, nn_idx, nn_w, N, S, blocks, blob_w, blob_b, out, ynext);
    else
        hipLaunchKernelGGL((fp_w_kernel<256, false>), dim3(grid), dim3(256), 0, st, points1, y, nn_idx, nn_w, N, S, blocks, blob_w, blob_b, out, ynext);
}

}  // namespace lsdm
