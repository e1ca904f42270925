// PointNet++ scene encoder dense stages for gfx950 (the dominant cost: 93 % of the path's FLOPs).
//
//   sa_kernel : PointNetSetAbstraction.forward minus the geometry  (pointnet2_utils.py:119-135,173-199)
//               gather [xyz_j - centroid || feat_j] -> 3 x (1x1 conv + BN(eval, folded) + ReLU) -> max over
//               the 32 neighbours.  The (B,C,32,S) grouped tensor never exists: a workgroup keeps its
//               TM = 32 x (centroids per tile) rows in LDS from the gather to the pooled output.
//   fp_kernel : PointNetFeaturePropagation.forward minus the 3-NN search  (pointnet2_utils.py:299-312)
//               [points1 || sum_j w_j * points2[idx_j]] -> conv+BN+ReLU stack; K is streamed through LDS
//               in chunks while the accumulators stay in registers.  The fp1 instance also runs the
//               backbone head conv1+bn1+ReLU, conv2 (pointnet2.py:76-79).
// Layout: every feature tensor is channel-last (clouds, points, C) so gathers read contiguous rows.
#include "mlp_core.hpp"
#include "kernels.hpp"

namespace lsdm {

__host__ __device__ constexpr int round8(int k) { return (k + 7) / 8 * 8; }
__host__ __device__ constexpr int cmax(int a, int b) { return a > b ? a : b; }

// ---------------------------------------------------------------------------------------------
template <int D, int C1, int C2, int C3, int TM>
__global__ __launch_bounds__(256) void sa_kernel(const float* __restrict__ src_xyz, const float* __restrict__ src_feat,
                                                 const float* __restrict__ new_xyz, const int* __restrict__ ball, int NSRC, int S,
                                                 Dense l1, Dense l2, Dense l3, float* __restrict__ out) {
    constexpr int K0 = 3 + D, K0P = round8(K0);
    constexpr int LDA = lds_ld(cmax(K0P, C2)), LDB = lds_ld(C1);
    __shared__ __attribute__((aligned(16))) float smem[TM * (LDA + LDB)];
    float* bufA = smem;
    float* bufB = smem + TM * LDA;
    const int c = blockIdx.y, tid = threadIdx.x;
    const int s0 = blockIdx.x * (TM / 32);
    const float* xyz = src_xyz + (size_t)c * NSRC * 3;
    const float* feat = src_feat + (size_t)c * NSRC * D;
    const int* bl = ball + ((size_t)c * S + s0) * kNSample;
    const float* cen = new_xyz + ((size_t)c * S + s0) * 3;
    for (int e = tid; e < TM * K0P; e += 256) {
        const int row = e / K0P, col = e - row * K0P;
        const int idx = bl[row];
        float v = 0.0f;
        if (col < 3) v = xyz[idx * 3 + col] - cen[(row >> 5) * 3 + col];
        else if (col < K0) v = feat[(size_t)idx * D + (col - 3)];
        bufA[row * LDA + col] = v;
    }
    __syncthreads();
    layer_lds<4, TM, K0P, C1, ACT_RELU>(bufA, LDA, l1.w, l1.b, bufB, LDB);
    __syncthreads();
    layer_lds<4, TM, C1, C2, ACT_RELU>(bufB, LDB, l2.w, l2.b, bufA, LDA);
    __syncthreads();
    // layer 3 + max over the 32 rows of each centroid, straight from the accumulators
    using TS = TileSplit<4, TM, C3>;
    const int wave = tid >> 6, lane = tid & 63;
    const int cb0 = wave % TS::CSTEP, rg = wave / TS::CSTEP;
    for (int cb = cb0; cb < TS::CB; cb += TS::CSTEP) {
        f32x16 acc[TS::RB];
        acc_zero(acc);
        mma_kloop<TS::RB>(acc, bufA + rg * TS::RB * 32 * LDA, LDA, l3.w, C3, cb, 0, C2 / 8, 0, lane);
        const int col = cb * 32 + (lane & 31);
        const float bv = l3.b[col];
#pragma unroll
        for (int rb = 0; rb < TS::RB; ++rb) {
            float m = acc[rb][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[rb][r]);
            m = fmaxf(m, __shfl_xor(m, 32));
            // relu(x + b) is monotone in x, so pooling before bias/ReLU is exact
            if (lane < 32) out[((size_t)c * S + s0 + rg * TS::RB + rb) * C3 + col] = reluf(m + bv);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// FP stage.  NL = number of conv layers (2 or 3); HEAD adds conv1+bn1+ReLU (128->128) and conv2 (128->3).
template <int D1, int D2, int C1, int C2, int C3, bool HEAD, int TM, int KC>
__global__ __launch_bounds__(256) void fp_kernel(const float* __restrict__ points1, const float* __restrict__ points2,
                                                 const int* __restrict__ nn_idx, const float* __restrict__ nn_w, int N, int S,
                                                 Dense l1, Dense l2, Dense l3, Dense h1, Dense h2, float* __restrict__ out) {
    constexpr int K0 = D1 + D2;
    static_assert(K0 % KC == 0 && KC % 8 == 0, "chunking");
    constexpr int LDA = lds_ld(cmax(KC, (C3 > 0 || HEAD) ? C2 : 0)), LDB = lds_ld(C1);
    __shared__ __attribute__((aligned(16))) float smem[TM * (LDA + LDB)];
    float* bufA = smem;
    float* bufB = smem + TM * LDA;
    const int c = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n0 = blockIdx.x * TM;
    const float* p1 = D1 > 0 ? points1 + ((size_t)c * N + n0) * D1 : nullptr;
    const float* p2 = points2 + (size_t)c * S * D2;
    const int* ni = nn_idx + ((size_t)c * N + n0) * 3;
    const float* nw = nn_w + ((size_t)c * N + n0) * 3;

    using TS = TileSplit<4, TM, C1>;
    constexpr int NCB = TS::CB / TS::CSTEP;
    const int cb0 = wave % TS::CSTEP, rg = wave / TS::CSTEP;
    f32x16 acc[NCB][TS::RB];
#pragma unroll
    for (int q = 0; q < NCB; ++q) acc_zero(acc[q]);
    for (int ch = 0; ch < K0 / KC; ++ch) {
        if (ch > 0) __syncthreads();
        for (int e = tid; e < TM * KC; e += 256) {
            const int row = e / KC, col = ch * KC + (e - row * KC);
            float v;
            if (col < D1) {
                v = p1[(size_t)row * D1 + col];
            } else {
                const int cc = col - D1;
                const float f0 = p2[(size_t)ni[row * 3 + 0] * D2 + cc] * nw[row * 3 + 0];
                const float f1 = p2[(size_t)ni[row * 3 + 1] * D2 + cc] * nw[row * 3 + 1];
                const float f2 = p2[(size_t)ni[row * 3 + 2] * D2 + cc] * nw[row * 3 + 2];
                v = __fadd_rn(__fadd_rn(f0, f1), f2);
            }
            bufA[row * LDA + (col - ch * KC)] = v;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NCB; ++q)
            mma_kloop<TS::RB>(acc[q], bufA + rg * TS::RB * 32 * LDA, LDA, l1.w, C1, cb0 + q * TS::CSTEP, ch * (KC / 8),
                              (ch + 1) * (KC / 8), ch * (KC / 8), lane);
    }
#pragma unroll
    for (int q = 0; q < NCB; ++q) store_lds<ACT_RELU, TS::RB>(acc[q], rg * TS::RB, cb0 + q * TS::CSTEP, l1.b, bufB, LDB, lane);
    __syncthreads();

    if constexpr (C3 == 0 && !HEAD) {
        // last layer: accumulators -> global (each half-wave writes 128 contiguous bytes per row)
        using T2 = TileSplit<4, TM, C2>;
        const int cb2 = wave % T2::CSTEP, rg2 = wave / T2::CSTEP;
        for (int cb = cb2; cb < T2::CB; cb += T2::CSTEP) {
            f32x16 a2[T2::RB];
            acc_zero(a2);
            mma_kloop<T2::RB>(a2, bufB + rg2 * T2::RB * 32 * LDB, LDB, l2.w, C2, cb, 0, C1 / 8, 0, lane);
            const int col = cb * 32 + (lane & 31);
            const float bv = l2.b[col];
#pragma unroll
            for (int rb = 0; rb < T2::RB; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (rg2 * T2::RB + rb) * 32 + acc_row(r, lane);
                    out[((size_t)c * N + n0 + row) * C2 + col] = reluf(a2[rb][r] + bv);
                }
        }
    } else {
        static_assert(!HEAD || (C1 == 128 && C2 == 128 && C3 == 128), "head expects the fp1 widths");
        layer_lds<4, TM, C1, C2, ACT_RELU>(bufB, LDB, l2.w, l2.b, bufA, LDA);
        __syncthreads();
        layer_lds<4, TM, C2, C3, ACT_RELU>(bufA, LDA, l3.w, l3.b, bufB, LDB);   // fp1 output
        __syncthreads();
        layer_lds<4, TM, 128, 128, ACT_RELU>(bufB, LDB, h1.w, h1.b, bufA, LDA);  // conv1 + bn1 + ReLU (dropout is identity in eval)
        __syncthreads();
        // conv2: 128 -> 3, plain weights (3,128) row-major, VALU
        for (int e = tid; e < TM * 3; e += 256) {
            const int row = e / 3, ch3 = e - row * 3;
            const float* a = bufA + row * LDA;
            const float* w = h2.w + ch3 * 128;
            float s = 0.0f;
#pragma unroll 8
            for (int k = 0; k < 128; ++k) s = fmaf(a[k], w[k], s);
            out[((size_t)c * N + n0 + row) * 3 + ch3] = s + h2.b[ch3];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// launchers (shapes are the fixed PointNet++ configuration of pointnet2.py:46-55)
// ---------------------------------------------------------------------------------------------
void launch_sa(int level, int num_clouds, const float* src_xyz, const float* src_feat, const float* new_xyz, const int* ball,
               const Dense* l, float* out, hipStream_t st) {
    switch (level) {
        case 0: hipLaunchKernelGGL((sa_kernel<3, 32, 32, 64, 256>), dim3(1024 / 8, num_clouds), dim3(256), 0, st, src_xyz, src_feat,
                                   new_xyz, ball, 1024, 1024, l[0], l[1], l[2], out); break;
        case 1: hipLaunchKernelGGL((sa_kernel<64, 64, 64, 128, 128>), dim3(256 / 4, num_clouds), dim3(256), 0, st, src_xyz, src_feat,
                                   new_xyz, ball, 1024, 256, l[0], l[1], l[2], out); break;
        case 2: hipLaunchKernelGGL((sa_kernel<128, 128, 128, 256, 64>), dim3(64 / 2, num_clouds), dim3(256), 0, st, src_xyz, src_feat,
                                   new_xyz, ball, 256, 64, l[0], l[1], l[2], out); break;
        case 3: hipLaunchKernelGGL((sa_kernel<256, 256, 256, 512, 64>), dim3(16 / 2, num_clouds), dim3(256), 0, st, src_xyz, src_feat,
                                   new_xyz, ball, 64, 16, l[0], l[1], l[2], out); break;
    }
}

void launch_fp(int which, int num_clouds, const float* points1, const float* points2, const int* nn_idx, const float* nn_w,
               const Dense* l, const Dense* head, float* out, hipStream_t st) {
    const Dense none{nullptr, nullptr};
    switch (which) {
        case 4: hipLaunchKernelGGL((fp_kernel<256, 512, 256, 256, 0, false, 64, 128>), dim3(64 / 64, num_clouds), dim3(256), 0, st,
                                   points1, points2, nn_idx, nn_w, 64, 16, l[0], l[1], none, none, none, out); break;
        case 3: hipLaunchKernelGGL((fp_kernel<128, 256, 256, 256, 0, false, 64, 128>), dim3(256 / 64, num_clouds), dim3(256), 0, st,
                                   points1, points2, nn_idx, nn_w, 256, 64, l[0], l[1], none, none, none, out); break;
        case 2: hipLaunchKernelGGL((fp_kernel<64, 256, 256, 128, 0, false, 64, 64>), dim3(1024 / 64, num_clouds), dim3(256), 0, st,
                                   points1, points2, nn_idx, nn_w, 1024, 256, l[0], l[1], none, none, none, out); break;
        case 1: hipLaunchKernelGGL((fp_kernel<0, 128, 128, 128, 128, true, 64, 128>), dim3(1024 / 64, num_clouds), dim3(256), 0, st,
                                   points1, points2, nn_idx, nn_w, 1024, 1024, l[0], l[1], l[2], head[0], head[1], out); break;
    }
}

}  // namespace lsdm
