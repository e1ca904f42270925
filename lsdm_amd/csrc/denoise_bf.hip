// The x/t-dependent tail of one denoising step on the split-bf16 MFMA core (bf16_core.hpp), gfx950.
//
// Same fusion and the same fp32 side arithmetic as denoise.hip (guidance g with the Q1/Q2 scrambles, in-place x += g,
// posterior / DDIM update; model/sdm.py:196-217, gaussian_diffusion.py:258-280,545-561,790-811); the dense layers
//   upsampling_layer.4 (transposed: rows = points, K = 512, per-POINT bias), combine_extraction, InputProcess and
//   OutputProcess (model/diffusion_utils.py:51-78,98-122)
// run as bf16x3 (NP = 2) or bf16 (NP = 1) products with fp32 accumulation.  One 256-thread workgroup per
// (sample, 32-point tile); all activations stay in LDS as bf16 planes:
//   R0 [32][264]  e tile (up-sampling output) -> later the 192-wide hidden state of combination_extraction
//   R1 [32][136]  emb = combine_extraction(e)  (shared by both denoiser passes; read as the second K half of cex0)
//   R2 [32][136]  W4 chunk staging -> pose embedding / 128-wide hidden states
//   R3 [32][72]   64-wide hidden states
#include "bf16_core.hpp"
#include "kernels.hpp"

namespace lsdm {

namespace {
// 32 points x 4 waves: 79 KB of LDS in bf16x3, i.e. two independent workgroups per CU whose phases interleave
// (64 points x 8 waves, one workgroup per CU, measured 7 % slower)
constexpr int TM = 32, NW = 4, NT = NW * 64;
constexpr int LD0 = ldh(256), LD1 = ldh(128), LDC = ldh(192), LD3 = ldh(64);
constexpr int P0 = TM * LD0, P1 = TM * LD1, P3 = TM * LD3;
constexpr int KC = 128;  // K chunk of the up-sampling GEMM (K = 512)
}  // namespace

__device__ __forceinline__ float scrambled_h(const float* __restrict__ pcd_b, const float* __restrict__ w_b, int op, int pp, int cp) {
    const int f = (op * kPoints + pp) * 3 + cp;
    const int j = f / kSlots, o = f - j * kSlots;
    return pcd_b[o * (kPoints * 3) + j] * w_b[o];
}

// per-point denoiser on the 64x3 tile zt (LDS fp32) -> res (64x3 fp32)
template <int NP>
__device__ __forceinline__ void denoiser_pass_h(const DenoiseHParams& P, const float* zt, __bf16* R0, const __bf16* R1, __bf16* R2,
                                                __bf16* R3, float* res) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // pose_embedding.0: 3 -> 64 sigmoid (K = 3: VALU), 4 consecutive channels per task
    for (int e = tid; e < TM * 16; e += NT) {
        const int row = e >> 4, j0 = (e & 15) * 4;
        const float z0 = zt[row * 3], z1 = zt[row * 3 + 1], z2 = zt[row * 3 + 2];
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* w = P.pose0.w + (j0 + j) * 3;
            v[j] = sigmoid_fast(fmaf(z2, w[2], fmaf(z1, w[1], z0 * w[0])) + P.pose0.b[j0 + j]);
        }
        store_split4<NP>(R3 + row * LD3 + j0, P3, v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
    layer_lds_h<NW, TM, 64, 128, ACT_SIGMOID, NP>(R3, LD3, P3, P.pose2, R2, LD1, P1);   // pose (128) -> R2
    __syncthreads();
    {   // combination_extraction.0: [pose (R2) || emb (R1)] 256 -> 192 sigmoid -> R0
        using TS = SplitH<NW, TM, 192, 4>;   // 6 channel blocks over 4 column groups (x point groups when NW > 4)
        const int cb0 = wave % TS::CSTEP, rg = wave / TS::CSTEP;
#pragma unroll 1
        for (int cb = cb0; cb < TS::CB; cb += TS::CSTEP) {
            f32x16 acc[TS::RB];
            acc_zero_h(acc);
            f32x4 bv[4];
            load_bias_h(bv, P.cex0.b, cb, lane);
            mma_kloop_h<TS::RB, NP, true, 8>(acc, R2 + rg * TS::RB * 32 * LD1, LD1, P1, P.cex0.w, 192, cb, 0, lane);
            mma_kloop_h<TS::RB, NP, true, 8>(acc, R1 + rg * TS::RB * 32 * LD1, LD1, P1, P.cex0.w, 192, cb, 8, lane);
            store_lds_h<ACT_SIGMOID, TS::RB, NP>(acc, rg * TS::RB, cb, bv, R0, LDC, P0, lane);
        }
    }
    __syncthreads();
    layer_lds_h<NW, TM, 192, 128, ACT_SIGMOID, NP>(R0, LDC, P0, P.cex2, R2, LD1, P1);
    __syncthreads();
    layer_lds_h<(TM / 32) * 2, TM, 128, 64, ACT_GELU, NP>(R2, LD1, P1, P.fin0, R3, LD3, P3);   // 2 x TM/32 block tasks: first waves only
    __syncthreads();
    // pose_final.2: 64 -> 3 GELU on the matrix pipe (fin2m = the three rows zero-padded to one 32-channel block), one wave per
    // 32-point block
    if (wave < TM / 32) {
        f32x16 a3[1];
        acc_zero_h(a3);
        mma_kloop_h<1, NP, false, 64 / 16>(a3, R3 + wave * 32 * LD3, LD3, P3, P.fin2m.w, 32, 0, 0, lane);
        const int c = lane & 31;
        if (c < 3) {
            const float bv = P.fin2.b[c];
#pragma unroll
            for (int r = 0; r < 16; ++r) res[(wave * 32 + acc_row(r, lane)) * 3 + c] = gelu_fast(a3[0][r] + bv);
        }
    }
    __syncthreads();
}

template <int NP>
__global__ __launch_bounds__(NT) void denoise_h_kernel(DenoiseHParams P, StepCoefs sc, int mode, float eta, int B, float* __restrict__ x,
                                                       const int64_t* __restrict__ t, const float* __restrict__ mask,
                                                       const float* __restrict__ pcd, const float* __restrict__ w,
                                                       const float* __restrict__ o12, const float* __restrict__ hm,
                                                       const __bf16* __restrict__ h2p, const float* __restrict__ noise,
                                                       float* __restrict__ out_x0, float* __restrict__ out_guiding,
                                                       float* __restrict__ out_sample) {
    __shared__ __attribute__((aligned(16))) __bf16 smem[NP * (P0 + 2 * P1 + P3)];
    __shared__ float fsm[3 * TM * 3 + 32];
    __bf16* R0 = smem;
    __bf16* R1 = R0 + NP * P0;
    __bf16* R2 = R1 + NP * P1;
    __bf16* R3 = R2 + NP * P1;
    float* xt = fsm;                  // mutated x tile (64x3)
    float* gt = xt + TM * 3;          // guiding-point tile
    float* res = gt + TM * 3;         // denoiser output tile
    float* oterm = res + TM * 3;      // W15[c, 3:15] . o[b,o',:]   (27 values)
    int b, tile;
    xcd_cloud_tile(kPoints / TM, b, tile);   // the tiles of one chain share its h2p / pcd rows: keep them on one XCD
    const int n0 = tile * TM;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float* pcd_b = pcd + (size_t)b * kSlots * kPoints * 3;
    const float* w_b = w + (size_t)b * kSlots;

    // ---- guidance g and the in-place update of x (fp32, as denoise.hip) ----------------------------------------
    if (tid < kSlots * 3) {
        const int op = tid / 3, c = tid - op * 3;
        float s = 0.0f;
        for (int k = 0; k < 12; ++k) s = fmaf(o12[((size_t)b * kSlots + op) * 12 + k], P.pwt.w[c * 15 + 3 + k], s);
        oterm[tid] = s;
    }
    __syncthreads();
    if (tid < TM * 3) {
        const int row = tid / 3, c = tid - row * 3, n = n0 + row;
        float acc = 0.0f;
        for (int op = 0; op < kSlots; ++op) {
            const float p0 = scrambled_h(pcd_b, w_b, op, n, 0), p1 = scrambled_h(pcd_b, w_b, op, n, 1), p2 = scrambled_h(pcd_b, w_b, op, n, 2);
            float s = fmaf(p2, P.pwt.w[c * 15 + 2], fmaf(p1, P.pwt.w[c * 15 + 1], p0 * P.pwt.w[c * 15]));
            s = gelu_fast(s + oterm[op * 3 + c] + P.pwt.b[c]);
            // (N,-1,B,O) view: the multiplier of flat element f is mask.flatten()[f % (B*O)]  (Q2)
            const long long f = (((long long)b * kSlots + op) * kPoints + n) * 3 + c;
            acc += s * mask[f % ((long long)B * kSlots)];
        }
        const size_t gi = ((size_t)b * kPoints + n) * 3 + c;
        const float g = (acc + hm[gi]) / 2.0f;
        const float xv = x[gi] + g;
        x[gi] = xv;  // Q5: the caller's tensor is mutated
        xt[tid] = xv;
        gt[tid] = g;
    }

    // ---- third up-sampling layer, transposed: e[p, c] = GELU(W4[n0+p, :] . h2[b, c, :] + b4[n0+p]) ------------------
    {
        using TS = SplitH<NW, TM, 256>;   // 8 channel blocks over the waves (one or two per wave), all point blocks each
        constexpr int NCB = TS::CB / TS::CSTEP;
        static_assert(TS::WPC == 1, "split");
        const __bf16* h2b = h2p + (size_t)b * 32 * 256 * NP * 16;
        f32x16 acc[NCB][TS::RB];
#pragma unroll
        for (int q = 0; q < NCB; ++q) acc_zero_h(acc[q]);
        for (int ch = 0; ch < 512 / KC; ++ch) {
            __syncthreads();
            // stage the chunk of the pre-split W4 rows: TM rows x 128 k per plane, 16-byte pieces
#pragma unroll
            for (int pl = 0; pl < NP; ++pl) {
                const __bf16* src = (pl == 0 ? P.up4_h : P.up4_l) + (size_t)n0 * 512 + ch * KC;
#pragma unroll
                for (int it = 0; it < TM * (KC / 8) / NT; ++it) {
                    const int e = tid + it * NT;
                    const int row = e / (KC / 8), c8 = e - row * (KC / 8);
                    *reinterpret_cast<bf16x8*>(R2 + pl * P1 + row * LD1 + c8 * 8) = *reinterpret_cast<const bf16x8*>(src + (size_t)row * 512 + c8 * 8);
                }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < NCB; ++q)
                mma_kloop_h<TS::RB, NP, true, KC / 16>(acc[q], R2, LD1, P1, h2b, 256, wave + q * TS::CSTEP, ch * (KC / 16), lane);
        }
        // epilogue: per-POINT bias, GELU, split -> R0[p][c]
        const int p = lane & 31, h = lane >> 5;
#pragma unroll
        for (int q = 0; q < NCB; ++q)
#pragma unroll
            for (int rb = 0; rb < TS::RB; ++rb) {
                const float bp = P.up4_b[n0 + rb * 32 + p];
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int n = (wave + q * TS::CSTEP) * 32 + 8 * qd + 4 * h;
                    store_split4<NP>(R0 + (rb * 32 + p) * LD0 + n, P0, gelu_fast(acc[q][rb][4 * qd + 0] + bp), gelu_fast(acc[q][rb][4 * qd + 1] + bp),
                                     gelu_fast(acc[q][rb][4 * qd + 2] + bp), gelu_fast(acc[q][rb][4 * qd + 3] + bp));
                }
            }
    }
    __syncthreads();
    // ---- combine_extraction 256 -> 128 GELU -> emb (R1) -----------------------------------------------------------
    layer_lds_h<NW, TM, 256, 128, ACT_GELU, NP>(R0, LD0, P0, P.comb, R1, LD1, P1);
    __syncthreads();

    // ---- denoiser on the mutated x ----------------------------------------------------------------------------------
    denoiser_pass_h<NP>(P, xt, R0, R1, R2, R3, res);
    if (tid < TM * 3) {
        const size_t gi = ((size_t)b * kPoints + n0) * 3 + tid;
        const float x0 = res[tid];
        out_x0[gi] = x0;
        if (mode != MODE_FORWARD) {
            const int tb = (int)t[b];
            const float xm = xt[tid], nz = tb != 0 ? 1.0f : 0.0f, nv = noise ? noise[gi] : 0.0f;
            float smp;
            if (mode == MODE_P_SAMPLE) {
                const float mean = __fadd_rn(__fmul_rn(sc.coef1[tb], x0), __fmul_rn(sc.coef2[tb], xm));
                const float sig = expf(__fmul_rn(0.5f, sc.logvar[tb]));
                smp = __fadd_rn(mean, __fmul_rn(__fmul_rn(nz, sig), nv));
            } else {
                const float ab = sc.acp[tb], abp = sc.acp_prev[tb];
                const float eps = __fdiv_rn(__fsub_rn(__fmul_rn(sc.sr_acp[tb], xm), x0), sc.srm1_acp[tb]);
                const float sigma = eta * sqrtf(__fdiv_rn(1.0f - abp, 1.0f - ab)) * sqrtf(1.0f - __fdiv_rn(ab, abp));
                const float mp = __fadd_rn(__fmul_rn(x0, sqrtf(abp)), __fmul_rn(sqrtf(1.0f - abp - sigma * sigma), eps));
                smp = __fadd_rn(mp, __fmul_rn(__fmul_rn(nz, sigma), nv));
            }
            out_sample[gi] = smp;
        }
    }
    __syncthreads();
    // ---- denoiser on the guiding points (saved_guiding_points, Q12) ------------------------------------------------
    denoiser_pass_h<NP>(P, gt, R0, R1, R2, R3, res);
    if (tid < TM * 3) out_guiding[((size_t)b * kPoints + n0) * 3 + tid] = res[tid];
}

void launch_denoise_h(int np, const DenoiseHParams& P, const StepCoefs& sc, int mode, float eta, int B, float* x, const int64_t* t,
                      const float* mask, const float* pcd, const float* w, const float* o12, const float* hm, const void* h2p,
                      const float* noise, float* out_x0, float* out_guiding, float* out_sample, hipStream_t st) {
    const __bf16* h2 = static_cast<const __bf16*>(h2p);
    if (np == 2)
        hipLaunchKernelGGL(denoise_h_kernel<2>, dim3(kPoints / TM * B), dim3(NT), 0, st, P, sc, mode, eta, B, x, t, mask, pcd, w, o12, hm, h2,
                           noise, out_x0, out_guiding, out_sample);
    else
        hipLaunchKernelGGL(denoise_h_kernel<1>, dim3(kPoints / TM * B), dim3(NT), 0, st, P, sc, mode, eta, B, x, t, mask, pcd, w, o12, hm, h2,
                           noise, out_x0, out_guiding, out_sample);
}

}  // namespace lsdm
