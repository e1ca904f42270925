// PointNet++ scene encoder dense stages on the split-bf16 MFMA core (bf16_core.hpp), gfx950.
//
// Same stages and HBM tensors as encoder.hip (sa_kernel / fp_kernel restate pointnet2_utils.py:119-135,173-199 and
// :299-312 + pointnet2.py:76-79); only the arithmetic of the 1x1-conv stacks differs:
//   NP = 2  bf16x3 products with fp32 accumulation; level features stay fp32 in HBM (same workspace as the f32 path)
//   NP = 1  plain bf16 products; level features are bf16 in HBM
// Activations are split into bf16 planes once, when they are written to LDS (gather or layer epilogue), never per
// MFMA.  The grouped (B,C,32,S) tensor never exists; the K order of the first SA layer is [features || xyz - centroid]
// (the host permutes the weight columns) so that feature rows are copied as aligned 16-byte pieces.
#include "bf16_core.hpp"
#include "kernels.hpp"

namespace lsdm {

// Optional phase timing (make PHASE=1): lane 0 of every workgroup adds its s_memtime deltas per phase to a device array.
#ifdef LSDM_PHASE_TIMING
__device__ unsigned long long g_phase[64 * 64];   // 64 spread copies of 64 slots (no same-address contention)
#define PHASE_T0() unsigned long long pt_ = __builtin_amdgcn_s_memtime(); unsigned long long pd_[8] = {}
#define PHASE_MARK(slot)                                                                  \
    do {                                                                                  \
        const unsigned long long n_ = __builtin_amdgcn_s_memtime();                       \
        pd_[(slot) & 7] += n_ - pt_;                                                      \
        pt_ = n_;                                                                         \
    } while (0)
#else
#define PHASE_T0() do {} while (0)
#define PHASE_MARK(slot) do {} while (0)
#endif


// ---------------------------------------------------------------------------------------------
// PRE: the first layer has been split across the grouping (encoder_u.hip): src_feat = U (NSRC x C1 per cloud), pre_v = V,
// pre_cent = index of every centroid among the source points; the gather writes ReLU(U_j - V_i) straight into the layer-2
// operand buffer and layer 1 is skipped.
template <int D, int C1, int C2, int C3, int TM, int NW, int NP, int WPS = 1, bool PRE = false>
__global__ __launch_bounds__(NW * 64, WPS) void sa_h_kernel(const float* __restrict__ src_xyz, const typename FeatT<NP>::type* __restrict__ src_feat,
                                                        const float* __restrict__ new_xyz, const int* __restrict__ ball, int NSRC,
                                                        int S, DenseH l1, DenseH l2, DenseH l3,
                                                        typename FeatT<NP>::type* __restrict__ out,
                                                        const float* __restrict__ pre_v = nullptr, const int* __restrict__ pre_cent = nullptr) {
    constexpr int NT = NW * 64;
    constexpr int K0P = (D == 3) ? 16 : round16(D + 3);
    constexpr int LDA = ldh(imax(K0P, C2)), LDB = ldh(C1);
    constexpr int PA = TM * LDA, PB = TM * LDB;   // plane sizes (elements)
    __shared__ __attribute__((aligned(16))) __bf16 smem[NP * (PA + PB)];
    __bf16* bufA = smem;
    __bf16* bufB = smem + NP * PA;
    int c, tile;
    xcd_cloud_tile(S / (TM / 32), c, tile);
    const int tid = threadIdx.x;
    const int s0 = tile * (TM / 32);
    const float* xyz = src_xyz + (size_t)c * NSRC * 3;
    const int* bl = ball + ((size_t)c * S + s0) * kNSample;
    const float* cen = new_xyz + ((size_t)c * S + s0) * 3;
    // weight fetches are chained one layer ahead: layer 1's first k-group is requested before the gather, layer l+1's
    // before layer l's epilogue
    using T1 = SplitH<NW, TM, C1>;
    using T2 = SplitH<NW, TM, C2>;
    using T3 = SplitH<NW, TM, C3>;
    const int wave_ = tid >> 6, lane_ = tid & 63;
    WFrag<NP, kgroup(K0P / 16)> w1;
    f32x4 b1[4];
    WFrag<NP, kgroup(C1 / 16)> w2;
    f32x4 b2[4];
    if constexpr (!PRE) {
        wfrag_load<NP, K0P / 16>(w1, l1.w, C1, wave_ % T1::CSTEP, 0, lane_);
        load_bias_h(b1, l1.b, wave_ % T1::CSTEP, lane_);
    } else {
        wfrag_load<NP, C1 / 16>(w2, l2.w, C2, wave_ % T2::CSTEP, 0, lane_);
        load_bias_h(b2, l2.b, wave_ % T2::CSTEP, lane_);
    }
    [[maybe_unused]] constexpr int PS = (D == 3 ? 0 : D == 64 ? 8 : D == 128 ? 16 : 24);   // phase-timing slot base of this level
    PHASE_T0();
    if constexpr (PRE) {
        static_assert(NP == 2 && D == C1, "split first layer: fp32 U / V of width C1");
        const float* Ub = src_feat + (size_t)c * NSRC * C1;
        const float* Vb = pre_v + (size_t)c * NSRC * C1;
        const int* ci = pre_cent + (size_t)c * S + s0;
        constexpr int CH = C1 / 4;
        static_assert((TM * CH) % NT == 0, "gather tiling");
        constexpr int ITER = TM * CH / NT;
        int gidx[ITER], vidx[ITER];
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int row = (tid + it * NT) / CH;
            gidx[it] = bl[row];
            vidx[it] = ci[row >> 5];
        }
        f32x4 gu[ITER], gw[ITER];
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int e = tid + it * NT;
            gu[it] = load4f(Ub + (size_t)gidx[it] * C1 + 4 * (e % CH));
            gw[it] = load4f(Vb + (size_t)vidx[it] * C1 + 4 * (e % CH));
        }
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int e = tid + it * NT;
            const int row = e / CH, ch = e - row * CH;
            store_split4<NP>(bufB + row * LDB + 4 * ch, PB, reluf(gu[it][0] - gw[it][0]), reluf(gu[it][1] - gw[it][1]),
                             reluf(gu[it][2] - gw[it][2]), reluf(gu[it][3] - gw[it][3]));
        }
    } else if constexpr (D == 3) {
        // sa1: the point features are the raw coordinates.  Row = [x y z | dx dy dz | 0 x 10]
        for (int row = tid; row < TM; row += NT) {
            const int idx = bl[row];
            const float x = xyz[idx * 3 + 0], y = xyz[idx * 3 + 1], z = xyz[idx * 3 + 2];
            const float* cp = cen + (row >> 5) * 3;
            __bf16* dst = bufA + row * LDA;
            store_split4<NP>(dst, PA, x, y, z, x - cp[0]);
            store_split4<NP>(dst + 4, PA, y - cp[1], z - cp[2], 0.0f, 0.0f);
            store_split4<NP>(dst + 8, PA, 0.0f, 0.0f, 0.0f, 0.0f);
            store_split4<NP>(dst + 12, PA, 0.0f, 0.0f, 0.0f, 0.0f);
        }
    } else {
        const typename FeatT<NP>::type* feat = src_feat + (size_t)c * NSRC * D;
        // (1) feature columns: TM x D/4 aligned 4-channel pieces, branch-free; every index load, then every row
        //     load of the thread is in flight before the first LDS store
        constexpr int CH = D / 4;
        static_assert((TM * CH) % NT == 0, "gather tiling");
        constexpr int ITER = TM * CH / NT;
        int gidx[ITER];
#pragma unroll
        for (int it = 0; it < ITER; ++it) gidx[it] = bl[(tid + it * NT) / CH];
        f32x4 gv[ITER];
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int e = tid + it * NT;
            gv[it] = load4f(feat + (size_t)gidx[it] * D + 4 * (e % CH));
        }
        // (2) the 16-wide tail [dx dy dz 0 ...] of each row
        f32x4 tv = {0.0f, 0.0f, 0.0f, 0.0f};
        if (tid < TM) {
            const int idx = bl[tid];
            const float* cp = cen + (tid >> 5) * 3;
            tv[0] = xyz[idx * 3 + 0] - cp[0]; tv[1] = xyz[idx * 3 + 1] - cp[1]; tv[2] = xyz[idx * 3 + 2] - cp[2];
        }
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int e = tid + it * NT;
            const int row = e / CH, ch = e - row * CH;
            store_split4<NP>(bufA + row * LDA + 4 * ch, PA, gv[it][0], gv[it][1], gv[it][2], gv[it][3]);
        }
        if (tid < TM) {
            __bf16* dst = bufA + tid * LDA + D;
            store_split4<NP>(dst, PA, tv[0], tv[1], tv[2], 0.0f);
#pragma unroll
            for (int j = 1; j < (K0P - D) / 4; ++j) store_split4<NP>(dst + 4 * j, PA, 0.0f, 0.0f, 0.0f, 0.0f);
        }
    }
    PHASE_MARK(PS + 0);   // gather issued + written (this wave)
    __syncthreads();
    PHASE_MARK(PS + 1);   // barrier wait
    if constexpr (!PRE) {
        layer_lds_pre_h<NW, TM, K0P, C1, ACT_RELU, NP>(bufA, LDA, PA, l1, bufB, LDB, PB, w1, b1, [&]() {
            wfrag_load<NP, C1 / 16>(w2, l2.w, C2, wave_ % T2::CSTEP, 0, lane_);
            load_bias_h(b2, l2.b, wave_ % T2::CSTEP, lane_);
        });
        PHASE_MARK(PS + 2);
        __syncthreads();
        PHASE_MARK(PS + 3);
    }
    WFrag<NP, kgroup(C2 / 16)> w3;
    layer_lds_pre_h<NW, TM, C1, C2, ACT_RELU, NP>(bufB, LDB, PB, l2, bufA, LDA, PA, w2, b2,
                                                   [&]() { wfrag_load<NP, C2 / 16>(w3, l3.w, C3, wave_ % T3::CSTEP, 0, lane_); });
    PHASE_MARK(PS + 4);
    __syncthreads();
    PHASE_MARK(PS + 5);
    // layer 3 in the direct orientation D[point][channel] + max over the 32 neighbours of each centroid
    using TS = T3;
    const int wave = wave_, lane = lane_;
    const int cb0 = wave % TS::CSTEP, rg = wave / TS::CSTEP;
#pragma unroll 1
    for (int cb = cb0; cb < TS::CB; cb += TS::CSTEP) {
        f32x16 acc[TS::RB];
        acc_zero_h(acc);
        const int col = cb * 32 + (lane & 31);
        const float bv = l3.b[col];
        if (cb == cb0) mma_kloop_pre_h<TS::RB, NP, false, C2 / 16>(acc, bufA + rg * TS::RB * 32 * LDA, LDA, PA, l3.w, C3, cb, 0, lane, w3);
        else mma_kloop_h<TS::RB, NP, false, C2 / 16>(acc, bufA + rg * TS::RB * 32 * LDA, LDA, PA, l3.w, C3, cb, 0, lane);
#pragma unroll
        for (int rb = 0; rb < TS::RB; ++rb) {
            float m = acc[rb][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[rb][r]);
            m = fmaxf(m, __shfl_xor(m, 32));
            // relu(x + b) is monotone in x, so pooling before bias/ReLU is exact
            if (lane < 32) store1(out + ((size_t)c * S + s0 + rg * TS::RB + rb) * C3 + col, reluf(m + bv));
        }
    }
    PHASE_MARK(PS + 6);
#ifdef LSDM_PHASE_TIMING
    if (threadIdx.x == 0) {
        unsigned long long* g = g_phase + (blockIdx.x & 63) * 64 + PS;
        for (int i = 0; i < 7; ++i) atomicAdd(g + i, pd_[i]);
        atomicAdd(g + 7, 1ull);
    }
#endif
}

#ifdef LSDM_PHASE_TIMING
__global__ void calib_kernel(unsigned long long* out) {
    const unsigned long long w0 = wall_clock64(), m0 = __builtin_amdgcn_s_memtime(), c0 = clock64();
    while (wall_clock64() - w0 < 100000ull) {}   // 1 ms of the 100 MHz constant clock
    out[0] = __builtin_amdgcn_s_memtime() - m0;
    out[1] = clock64() - c0;
    out[2] = wall_clock64() - w0;
}
extern "C" __attribute__((visibility("default"))) int lsdm_debug_calibrate(unsigned long long* out3) {
    unsigned long long* d = nullptr;
    if (hipMalloc(&d, 24) != hipSuccess) return -1;
    hipLaunchKernelGGL(calib_kernel, dim3(1), dim3(64), 0, 0, d);
    const int rc = hipMemcpy(out3, d, 24, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
    (void)hipFree(d);
    return rc;
}
extern "C" __attribute__((visibility("default"))) int lsdm_debug_phase_counters(unsigned long long* out, int reset) {
    static unsigned long long h[64 * 64], z[64 * 64];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase), sizeof h) != hipSuccess) return -1;
    for (int s = 0; s < 64; ++s) { out[s] = 0; for (int c = 0; c < 64; ++c) out[s] += h[c * 64 + s]; }
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof z) != hipSuccess) return -1;
    return 0;
}
#endif

// ---------------------------------------------------------------------------------------------
// FP stage: [points1 || sum_j w_j * points2[idx_j]] -> conv+BN+ReLU stack; K streamed through LDS in chunks of KC.
// HEAD adds conv1+bn1+ReLU (128->128) and conv2 (128->3, fp32 VALU) and writes fp32.
// FUSEY (two-layer form): the stage's output feeds only the next finer stage's interpolation, whose first layer is pre-multiplied at
// this level (encoder_u.hip::fp_pre_kernel): ynext = out x l3^T (no bias, no activation) is one more layer on the rows already on
// chip; `out` itself is then stored only when keep_out (parity taps).
template <int D1, int D2, int C1, int C2, int C3, bool HEAD, int TM, int KC, int NW, int NP, int WPS = 1, bool FUSEY = false>
__global__ __launch_bounds__(NW * 64, WPS) void fp_h_kernel(const typename FeatT<NP>::type* __restrict__ points1,
                                                        const typename FeatT<NP>::type* __restrict__ points2,
                                                        const int* __restrict__ nn_idx, const float* __restrict__ nn_w, int N, int S,
                                                        DenseH l1, DenseH l2, DenseH l3, DenseH h1, Lin h2, DenseH h2m, void* __restrict__ out_v,
                                                        float* __restrict__ ynext = nullptr, int keep_out = 1) {
    typedef typename FeatT<NP>::type feat_t;
    constexpr int NT = NW * 64;
    constexpr int K0 = D1 + D2;
    static_assert(K0 % KC == 0 && KC % 16 == 0 && D1 % 4 == 0, "chunking");
    constexpr int LDX = ldh(KC), LDB = ldh(C1);
    // two-layer form: the chunk buffer is dead when the layer-1 result is written, so both share one region;
    // three-layer + head form: ping-pong between two 128-wide buffers
    constexpr int PX = TM * LDX, PB = TM * LDB;
    constexpr int SMEM = HEAD ? NP * (PX + PB) : NP * imax(PX, PB);
    __shared__ __attribute__((aligned(16))) __bf16 smem[SMEM];
    __bf16* bufX = smem;
    __bf16* bufB = HEAD ? smem + NP * PX : smem;
    int c, tile;
    xcd_cloud_tile(N / TM, c, tile);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n0 = tile * TM;
    const feat_t* p1 = D1 > 0 ? points1 + ((size_t)c * N + n0) * D1 : nullptr;
    const feat_t* p2 = points2 + (size_t)c * S * D2;
    const int* ni = nn_idx + ((size_t)c * N + n0) * 3;
    const float* nw = nn_w + ((size_t)c * N + n0) * 3;

    using TS = SplitH<NW, TM, C1>;
    constexpr int NCB = TS::CB / TS::CSTEP;
    const int cb0 = wave % TS::CSTEP, rg = wave / TS::CSTEP;
    f32x16 acc[NCB][TS::RB];
    f32x4 bv1[NCB][4];
#pragma unroll
    for (int q = 0; q < NCB; ++q) { load_bias_h(bv1[q], l1.b, cb0 + q * TS::CSTEP, lane); acc_bias_h(acc[q], bv1[q]); }
    for (int ch = 0; ch < K0 / KC; ++ch) {
        if (ch > 0) __syncthreads();
        static_assert((TM * (KC / 4)) % NT == 0 && (D1 % KC == 0), "gather tiling: a chunk is all points1 or all interpolation");
        constexpr int ITER = TM * (KC / 4) / NT;
        if (ch * KC < D1) {   // uniform: plain copy of points1 columns
            f32x4 gv[ITER];
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int e = tid + it * NT;
                const int row = e / (KC / 4), c4 = e - row * (KC / 4);
                gv[it] = load4f(p1 + (size_t)row * D1 + ch * KC + 4 * c4);
            }
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int e = tid + it * NT;
                const int row = e / (KC / 4), c4 = e - row * (KC / 4);
                store_split4<NP>(bufX + row * LDX + 4 * c4, PX, gv[it][0], gv[it][1], gv[it][2], gv[it][3]);
            }
        } else {              // inverse-distance interpolation of the three nearest coarse points
            const int cc0 = ch * KC - D1;
            constexpr int HALF = ITER > 2 ? 2 : ITER;   // bound the loads in flight (3 rows x 16 B each per iteration)
            static_assert(ITER % HALF == 0, "gather tiling");
#pragma unroll 1
            for (int h0 = 0; h0 < ITER; h0 += HALF) {
                int i0[HALF], i1[HALF], i2[HALF];
                float w0[HALF], w1[HALF], w2[HALF];
#pragma unroll
                for (int it = 0; it < HALF; ++it) {
                    const int row = (tid + (h0 + it) * NT) / (KC / 4);
                    i0[it] = ni[row * 3 + 0]; i1[it] = ni[row * 3 + 1]; i2[it] = ni[row * 3 + 2];
                    w0[it] = nw[row * 3 + 0]; w1[it] = nw[row * 3 + 1]; w2[it] = nw[row * 3 + 2];
                }
                f32x4 f0[HALF], f1[HALF], f2[HALF];
#pragma unroll
                for (int it = 0; it < HALF; ++it) {
                    const int e = tid + (h0 + it) * NT;
                    const int cc = cc0 + 4 * (e % (KC / 4));
                    f0[it] = load4f(p2 + (size_t)i0[it] * D2 + cc);
                    f1[it] = load4f(p2 + (size_t)i1[it] * D2 + cc);
                    f2[it] = load4f(p2 + (size_t)i2[it] * D2 + cc);
                }
#pragma unroll
                for (int it = 0; it < HALF; ++it) {
                    const int e = tid + (h0 + it) * NT;
                    const int row = e / (KC / 4), c4 = e - row * (KC / 4);
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        v[j] = interp3(f0[it][j], f1[it][j], f2[it][j], w0[it], w1[it], w2[it]);
                    store_split4<NP>(bufX + row * LDX + 4 * c4, PX, v[0], v[1], v[2], v[3]);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NCB; ++q)
            mma_kloop_h<TS::RB, NP, true, KC / 16>(acc[q], bufX + rg * TS::RB * 32 * LDX, LDX, PX, l1.w, C1, cb0 + q * TS::CSTEP,
                                                   ch * (KC / 16), lane);
    }
    if constexpr (!HEAD) __syncthreads();   // bufB aliases the chunk buffer
#pragma unroll
    for (int q = 0; q < NCB; ++q) store_lds_b_h<ACT_RELU, TS::RB, NP>(acc[q], rg * TS::RB, cb0 + q * TS::CSTEP, bufB, LDB, PB, lane);
    __syncthreads();

    if constexpr (!HEAD && FUSEY) {
        // layer 2 transposed, all blocks of the wave kept in accumulators until every wave is done with bufB; its ReLU output becomes
        // the operand tile of the extra layer in the same region
        static_assert(!FUSEY || (NP == 2 && C2 == 256), "fused Y: bf16x3, 256-wide stage output");
        float* out = static_cast<float*>(out_v);
        using T2 = SplitH<NW, TM, C2>;
        static_assert(!FUSEY || T2::WPC == 1, "split");
        constexpr int RB2 = T2::RB;
        constexpr int NCB2 = T2::CB / T2::CSTEP;
        constexpr int LDC = ldh(C2), PC = TM * LDC;
        static_assert(!FUSEY || NP * PC <= SMEM, "operand tile of the extra layer fits the region");
        const int cb2 = wave % T2::CSTEP, p = lane & 31, h = lane >> 5;
        f32x16 a2[NCB2][RB2];
#pragma unroll
        for (int q = 0; q < NCB2; ++q) {
            f32x4 bv2[4];
            load_bias_h(bv2, l2.b, cb2 + q * T2::CSTEP, lane);
            acc_bias_h(a2[q], bv2);
            mma_kloop_h<RB2, NP, true, C1 / 16>(a2[q], bufB, LDB, PB, l2.w, C2, cb2 + q * T2::CSTEP, 0, lane);
        }
        __syncthreads();
        __bf16* bufC = smem;
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
    } else if constexpr (!HEAD) {
        // last layer, direct orientation: 32 consecutive channels per half-wave -> global
        feat_t* out = static_cast<feat_t*>(out_v);
        using T2 = SplitH<NW, TM, C2>;
        const int cb2 = wave % T2::CSTEP, rg2 = wave / T2::CSTEP;
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
                    store1(out + ((size_t)c * N + n0 + row) * C2 + col, reluf(a2[rb][r] + bv));
                }
        }
    } else {
        static_assert(!HEAD || (C1 == 128 && C2 == 128 && C3 == 128 && KC == 128), "head expects the fp1 widths");
        float* out = static_cast<float*>(out_v);
        layer_lds_h<NW, TM, C1, C2, ACT_RELU, NP>(bufB, LDB, PB, l2, bufX, LDX, PX);
        __syncthreads();
        layer_lds_h<NW, TM, C2, C3, ACT_RELU, NP>(bufX, LDX, PX, l3, bufB, LDB, PB);   // fp1 output
        __syncthreads();
        layer_lds_h<NW, TM, 128, 128, ACT_RELU, NP>(bufB, LDB, PB, h1, bufX, LDX, PX);  // conv1 + bn1 + ReLU (dropout = identity in eval)
        __syncthreads();
        // conv2: 128 -> 3 on the matrix pipe (h2m = the three rows zero-padded to one 32-channel block): two waves, one
        // 32-point block each -- as VALU work the 64 x 3 x 128 FMAs with their hi + lo LDS reads cost every wave ~480 instructions
        if (wave < TM / 32) {
            f32x16 a3[1];
            acc_zero_h(a3);
            mma_kloop_h<1, NP, false, 128 / 16>(a3, bufX + wave * 32 * LDX, LDX, PX, h2m.w, 32, 0, 0, lane);
            const int col = lane & 31;
            if (col < 3) {
                const float bv = h2.b[col];
#pragma unroll
                for (int r = 0; r < 16; ++r) out[((size_t)c * N + n0 + wave * 32 + acc_row(r, lane)) * 3 + col] = a3[0][r] + bv;
            }
        }
    }
}

// acc_row is defined in mlp_core.hpp for the f32 core; same C/D map here
// (col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5))

// ---------------------------------------------------------------------------------------------
template <int NP>
static void launch_sa_h_t(int level, int nc, const float* src_xyz, const void* src_feat, const float* new_xyz, const int* ball,
                          const DenseH* l, void* out, hipStream_t st) {
    typedef typename FeatT<NP>::type F;
    const F* sf = static_cast<const F*>(src_feat);
    F* o = static_cast<F*>(out);
    switch (level) {      // (the reference form is kept for sa4 only: sa1 / sa2 / sa3 have register-chained ones, encoder_reg.hip)
        case 3: hipLaunchKernelGGL((sa_h_kernel<256, 256, 256, 512, 64, 8, NP>), dim3(16 / 2 * nc), dim3(512), 0, st, src_xyz, sf, new_xyz,
                                   ball, 64, 16, l[0], l[1], l[2], o); break;
    }
}

template <int NP>
static void launch_fp_h_t(int which, int nc, const void* points1, const void* points2, const int* nn_idx, const float* nn_w,
                          const DenseH* l, const DenseH* head1, Lin head2, void* out, hipStream_t st) {
    typedef typename FeatT<NP>::type F;
    const F* p1 = static_cast<const F*>(points1);
    const F* p2 = static_cast<const F*>(points2);
    const DenseH none{nullptr, nullptr};
    const Lin nol{nullptr, nullptr};
    switch (which) {
        case 4: hipLaunchKernelGGL((fp_h_kernel<256, 512, 256, 256, 0, false, 64, 128, 4, NP>), dim3(64 / 64 * nc), dim3(256), 0, st, p1, p2,
                                   nn_idx, nn_w, 64, 16, l[0], l[1], none, none, nol, none, out); break;
        case 3: hipLaunchKernelGGL((fp_h_kernel<128, 256, 256, 256, 0, false, 64, 128, 4, NP>), dim3(256 / 64 * nc), dim3(256), 0, st, p1, p2,
                                   nn_idx, nn_w, 256, 64, l[0], l[1], none, none, nol, none, out); break;
        case 2: hipLaunchKernelGGL((fp_h_kernel<64, 256, 256, 128, 0, false, 64, 64, 4, NP>), dim3(1024 / 64 * nc), dim3(256), 0, st, p1, p2,
                                   nn_idx, nn_w, 1024, 256, l[0], l[1], none, none, nol, none, out); break;
        case 1: hipLaunchKernelGGL((fp_h_kernel<0, 128, 128, 128, 128, true, 64, 128, 4, NP>), dim3(1024 / 64 * nc), dim3(256), 0, st, p1, p2,
                                   nn_idx, nn_w, 1024, 1024, l[0], l[1], l[2], head1[0], head2, head1[1], out); break;
    }
}

void launch_sa_h(int np, int level, int nc, const float* src_xyz, const void* src_feat, const float* new_xyz, const int* ball,
                 const DenseH* l, void* out, hipStream_t st) {
    (void)np;      // (the LDS-tiled reference forms exist for two planes: plain bf16 never reaches them, api.hip)
    launch_sa_h_t<2>(level, nc, src_xyz, src_feat, new_xyz, ball, l, out, st);
}

void launch_fp_h(int np, int which, int nc, const void* points1, const void* points2, const int* nn_idx, const float* nn_w,
                 const DenseH* l, const DenseH* head1, Lin head2, void* out, hipStream_t st) {
    (void)np;
    launch_fp_h_t<2>(which, nc, points1, points2, nn_idx, nn_w, l, head1, head2, out, st);
}

}  // namespace lsdm
