// The x/t-dependent tail of one denoising step, fused into one kernel per (sample, 64-point tile) on gfx950.
//
// Restates, for the tile's points n0..n0+63 of sample b:
//   model/sdm.py:196-204   point_wise_trans_layer on [P || o], mask through the (N,-1,B,O) view (Q2), sum over the
//                          objects, (sum + hm)/2 = g, and the IN-PLACE  x += g  (Q5)
//   model/sdm.py:164-167   third upsampling layer  e[n,c] = GELU(W4[n,:] . h2[c,:] + b4[n])  -- evaluated in the
//                          transposed orientation so the (B,256,N)->(B,N,256) permute never touches memory: the
//                          weight rows W4[n,:] are the A operand, the per-sample activations h2 (packed by
//                          upsample12_kernel) are the B operand
//   model/sdm.py:208       combine_extraction 256 -> 128 GELU
//   model/sdm.py:211-217   InputProcess + OutputProcess on x and again on the guiding points g
//                          (model/diffusion_utils.py:51-78, 98-122)
//   diffusion/gaussian_diffusion.py:258-280, 545-561   posterior mean on the MUTATED x and the ancestral sample,
//   diffusion/gaussian_diffusion.py:790-811            or the DDIM update (eta-parametrised)
// so the (B,N,256) conditioning tensor, the per-point hidden states and the posterior temporaries never reach HBM.
#include "mlp_core.hpp"
#include "kernels.hpp"

namespace lsdm {

namespace {
constexpr int TM = 64;
constexpr int LD0 = lds_ld(256);  // 260: e tile, later [pose(128) || emb(128)]
constexpr int LD1 = lds_ld(128);  // 132: W4 chunk staging, emb, 192->128 output
constexpr int LD2 = lds_ld(192);  // 196: 3->64, 256->192, 128->64 outputs
constexpr int KC = 128;           // K chunk of the up-sampling GEMM (K = 512)
}  // namespace

__device__ __forceinline__ float scrambled(const float* __restrict__ pcd_b, const float* __restrict__ w_b, int op, int pp, int cp) {
    const int f = (op * kPoints + pp) * 3 + cp;
    const int j = f / kSlots, o = f - j * kSlots;
    return pcd_b[o * (kPoints * 3) + j] * w_b[o];
}

// per-point denoiser on the 64x3 tile held in zt (LDS); result (64x3) written to res (LDS)
__device__ __forceinline__ void denoiser_pass(const DenoiseParams& P, const float* zt, float* buf0, float* buf1, float* buf2, float* res) {
    const int tid = threadIdx.x;
    // pose_embedding.0: 3 -> 64 sigmoid (K = 3: VALU)
    for (int e = tid; e < TM * 64; e += 256) {
        const int row = e >> 6, j = e & 63;
        const float* w = P.pose0.w + j * 3;
        const float s = fmaf(zt[row * 3 + 2], w[2], fmaf(zt[row * 3 + 1], w[1], zt[row * 3] * w[0])) + P.pose0.b[j];
        buf2[row * LD2 + j] = sigmoidf(s);
    }
    __syncthreads();
    layer_lds<4, TM, 64, 128, ACT_SIGMOID>(buf2, LD2, P.pose2.w, P.pose2.b, buf0, LD0);   // -> buf0[:, 0:128]
    __syncthreads();
    layer_lds<4, TM, 256, 192, ACT_SIGMOID>(buf0, LD0, P.cex0.w, P.cex0.b, buf2, LD2);    // [pose || emb] -> 192
    __syncthreads();
    layer_lds<4, TM, 192, 128, ACT_SIGMOID>(buf2, LD2, P.cex2.w, P.cex2.b, buf1, LD1);
    __syncthreads();
    layer_lds<4, TM, 128, 64, ACT_GELU>(buf1, LD1, P.fin0.w, P.fin0.b, buf2, LD2);
    __syncthreads();
    // pose_final.2: 64 -> 3 GELU (VALU)
    for (int e = tid; e < TM * 3; e += 256) {
        const int row = e / 3, c = e - row * 3;
        const float* a = buf2 + row * LD2;
        const float* w = P.fin2.w + c * 64;
        float s = 0.0f;
#pragma unroll 8
        for (int k = 0; k < 64; ++k) s = fmaf(a[k], w[k], s);
        res[e] = gelu(s + P.fin2.b[c]);
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void denoise_kernel(DenoiseParams P, StepCoefs sc, int mode, float eta, int B,
                                                      float* __restrict__ x, const int64_t* __restrict__ t,
                                                      const float* __restrict__ mask, const float* __restrict__ pcd,
                                                      const float* __restrict__ w, const float* __restrict__ o12,
                                                      const float* __restrict__ hm, const float* __restrict__ h2p,
                                                      const float* __restrict__ noise, float* __restrict__ out_x0,
                                                      float* __restrict__ out_guiding, float* __restrict__ out_sample) {
    __shared__ __attribute__((aligned(16))) float smem[TM * (LD0 + LD1 + LD2) + 3 * TM * 3 + 32];
    float* buf0 = smem;
    float* buf1 = buf0 + TM * LD0;
    float* buf2 = buf1 + TM * LD1;
    float* xt = buf2 + TM * LD2;      // mutated x tile (64x3)
    float* gt = xt + TM * 3;          // guiding-point tile
    float* res = gt + TM * 3;         // denoiser output tile
    float* oterm = res + TM * 3;      // W15[c, 3:15] . o[b,o',:]   (27 values)
    const int b = blockIdx.y, n0 = blockIdx.x * TM;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float* pcd_b = pcd + (size_t)b * kSlots * kPoints * 3;
    const float* w_b = w + (size_t)b * kSlots;

    // ---- guidance g and the in-place update of x --------------------------------------------------------------
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
            const float p0 = scrambled(pcd_b, w_b, op, n, 0), p1 = scrambled(pcd_b, w_b, op, n, 1), p2 = scrambled(pcd_b, w_b, op, n, 2);
            float s = fmaf(p2, P.pwt.w[c * 15 + 2], fmaf(p1, P.pwt.w[c * 15 + 1], p0 * P.pwt.w[c * 15]));
            s = gelu(s + oterm[op * 3 + c] + P.pwt.b[c]);
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

    // ---- third up-sampling layer, transposed: e[row, c] = GELU(W4[n0+row, :] . h2[b, c, :] + b4[n0+row]) --------
    {
        using TS = TileSplit<4, TM, 256>;  // CB = 8: two column blocks per wave, RB = 2
        constexpr int NCB = TS::CB / TS::CSTEP;
        const float* h2b = h2p + (size_t)b * 512 * 256;
        f32x16 acc[NCB][TS::RB];
#pragma unroll
        for (int q = 0; q < NCB; ++q) acc_zero(acc[q]);
        for (int ch = 0; ch < 512 / KC; ++ch) {
            __syncthreads();
            for (int e = tid; e < TM * (KC / 4); e += 256) {
                const int row = e / (KC / 4), c4 = e - row * (KC / 4);
                const f32x4 v = *reinterpret_cast<const f32x4*>(P.up4_w + (size_t)(n0 + row) * 512 + ch * KC + c4 * 4);
                *reinterpret_cast<f32x4*>(buf1 + row * LD1 + c4 * 4) = v;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < NCB; ++q)
                mma_kloop<TS::RB>(acc[q], buf1, LD1, h2b, 256, wave + q * TS::CSTEP, ch * (KC / 8), (ch + 1) * (KC / 8), ch * (KC / 8), lane);
        }
#pragma unroll
        for (int q = 0; q < NCB; ++q) {
            const int col = (wave + q * TS::CSTEP) * 32 + (lane & 31);
#pragma unroll
            for (int rb = 0; rb < TS::RB; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rb * 32 + acc_row(r, lane);
                    buf0[row * LD0 + col] = gelu(acc[q][rb][r] + P.up4_b[n0 + row]);
                }
        }
    }
    __syncthreads();
    // ---- combine_extraction 256 -> 128 GELU; park emb in buf0[:, 128:256] for both denoiser passes ---------------
    layer_lds<4, TM, 256, 128, ACT_GELU>(buf0, LD0, P.comb.w, P.comb.b, buf1, LD1);
    __syncthreads();
    for (int e = tid; e < TM * 128; e += 256) {
        const int row = e >> 7, j = e & 127;
        buf0[row * LD0 + 128 + j] = buf1[row * LD1 + j];
    }
    __syncthreads();

    // ---- denoiser on the mutated x ------------------------------------------------------------------------------
    denoiser_pass(P, xt, buf0, buf1, buf2, res);
    if (tid < TM * 3) {
        const size_t gi = ((size_t)b * kPoints + n0) * 3 + tid;
        const float x0 = res[tid];
        out_x0[gi] = x0;
        if (mode != MODE_FORWARD) {
            const int tb = (int)t[b];
            const float xm = xt[tid], nz = tb != 0 ? 1.0f : 0.0f, nv = noise ? noise[gi] : 0.0f;
            float smp;
            if (mode == MODE_P_SAMPLE) {
                // mean = c1*x0 + c2*x_t(mutated);  sample = mean + 1[t!=0]*exp(0.5*logvar)*noise
                const float mean = __fadd_rn(__fmul_rn(sc.coef1[tb], x0), __fmul_rn(sc.coef2[tb], xm));
                const float sig = expf(__fmul_rn(0.5f, sc.logvar[tb]));
                smp = __fadd_rn(mean, __fmul_rn(__fmul_rn(nz, sig), nv));
            } else {
                // eps = (sqrt(1/acp)*x - x0) / sqrt(1/acp - 1);  x' = x0*sqrt(acp_prev) + sqrt(1 - acp_prev - sigma^2)*eps + sigma*noise
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
    // ---- denoiser on the guiding points (saved_guiding_points, Q12) ----------------------------------------------
    denoiser_pass(P, gt, buf0, buf1, buf2, res);
    if (tid < TM * 3) out_guiding[((size_t)b * kPoints + n0) * 3 + tid] = res[tid];
}

void launch_denoise(const DenoiseParams& P, const StepCoefs& sc, int mode, float eta, int B, float* x, const int64_t* t,
                    const float* mask, const float* pcd, const float* w, const float* o12, const float* hm, const float* h2p,
                    const float* noise, float* out_x0, float* out_guiding, float* out_sample, hipStream_t st) {
    hipLaunchKernelGGL(denoise_kernel, dim3(kPoints / TM, B), dim3(256), 0, st, P, sc, mode, eta, B, x, t, mask, pcd, w, o12, hm,
                       h2p, noise, out_x0, out_guiding, out_sample);
}

}  // namespace lsdm
