// The x/t-dependent tail of a denoising step on the split-bf16 MFMA core (bf16_core.hpp), gfx950.
//
// Same fusion and the same fp32 side arithmetic as denoise.hip (guidance g with the Q1/Q2 scrambles, in-place x += g,
// posterior / DDIM update; model/sdm.py:196-217, gaussian_diffusion.py:258-280,545-561,790-811); the dense layers
//   upsampling_layer.4 (transposed: rows = points, K = 512, per-POINT bias), combine_extraction, InputProcess and
//   OutputProcess (model/diffusion_utils.py:51-78,98-122)
// run as bf16x3 (NP = 2) or bf16 (NP = 1) products with fp32 accumulation.
//
// What is evaluated where (model/sdm.py:164-167, 208-217):
//   * e = upsampling_layer([ets || enc]) has 256 channels per point: 0..127 come from the timestep embedding, 128..255 from the
//     text embedding only.  combine_extraction is linear in e before its GELU, so its text half
//         ctext[b, p, :] = W_comb[:, 128:256] . GELU(W4[p, :] . h2[b, 128 + c, :] + b4[p])
//     does not depend on t or x: ctext_kernel evaluates it ONCE per library call (33 MB at B = 64) and every step starts
//     the accumulators of combine_extraction from it -- the K = 512 up-sampling product and combine_extraction shrink by half.
//   * combination_extraction.0 reads [pose(z) || emb]; its emb half (K = 128 of 256) is the same for the two denoiser passes
//     of a step (z = the mutated x, z = the guiding points): it is accumulated once per step and kept in registers.
//   * denoise_h_kernel runs `nsteps` consecutive steps of its 32-point tile inside ONE launch: x[p] at step k+1 depends only on
//     x[p], the guidance g[p], the step's embedding rows and its noise, so the tile's x stays in LDS between steps.  The strict
//     schedules launch it with nsteps = 1 (the scene features change every step); the hoisted-conditioning mode runs a whole
//     group of steps per launch -- no launch boundary, no x round trip.  The denoiser pass on the guiding points produces
//     model.saved_guiding_points, which only the last forward of a call defines (Q12): with nsteps > 1 it runs on the last step.
// One 256-thread workgroup per (sample, 32-point tile); activations in LDS as bf16 planes:
//   R0 [32][200]  e_t tile (128 wide) -> later the 192-wide hidden state of combination_extraction
//   R1 [32][136]  emb = combine_extraction(e)  (shared by both denoiser passes)
//   R2 [32][136]  W4 chunk staging -> pose embedding / 128-wide hidden states
//   R3 [32][72]   64-wide hidden states
#include "bf16_core.hpp"
#include "kernels.hpp"

namespace lsdm {

namespace {
constexpr int TM = 32, NW = 4, NT = NW * 64;
constexpr int LD0 = ldh(192), LD1 = ldh(128), LD3 = ldh(64);
constexpr int P0 = TM * LD0, P1 = TM * LD1, P3 = TM * LD3;
constexpr int KC = 128;  // K chunk of the up-sampling product (K = 512)
constexpr int NE = 128;  // channels of each half of e
}  // namespace

__device__ __forceinline__ float scrambled_h(const float* __restrict__ pcd_b, const float* __restrict__ w_b, int op, int pp, int cp) {
    const int f = (op * kPoints + pp) * 3 + cp;
    const int j = f / kSlots, o = f - j * kSlots;
    return pcd_b[o * (kPoints * 3) + j] * w_b[o];
}

// epilogue of a transposed block whose bias is already in the accumulators: activation + split -> LDS [point][channel]
template <int ACT, int NP>
__device__ __forceinline__ void store_lds_nb_h(const f32x16& acc, int cb, __bf16* out, int ldo, int plane, int lane) {
    const int p = lane & 31, h = lane >> 5;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        store_split4<NP>(out + p * ldo + cb * 32 + 8 * q + 4 * h, plane, apply_act_fast<ACT>(acc[4 * q + 0]), apply_act_fast<ACT>(acc[4 * q + 1]),
                         apply_act_fast<ACT>(acc[4 * q + 2]), apply_act_fast<ACT>(acc[4 * q + 3]));
}

// One half of e for the tile: acc = W4[n0 + p, :] . h2half[c, :] (transposed: D[c][p]), one 32-channel block per wave;
// the W4 rows are staged through R2 in chunks of KC; epilogue per-POINT bias + GELU + split -> R0[p][c]
template <int NP>
__device__ __forceinline__ void upsample_half_h(const DenoiseHParams& P, int n0, const __bf16* __restrict__ h2half, __bf16* R0, __bf16* R2) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    f32x16 acc[1];
    acc_zero_h(acc);
    for (int ch = 0; ch < 512 / KC; ++ch) {
        __syncthreads();
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
            const __bf16* src = (pl == 0 ? P.up4_h : P.up4_l) + (size_t)n0 * 512 + ch * KC;
#pragma unroll
            for (int it = 0; it < TM * (KC / 8) / NT; ++it) {
                const int e = tid + it * NT;
                const int row = e / (KC / 8), c8 = e - row * (KC / 8);
                *reinterpret_cast<bf16x8*>(R2 + pl * P1 + row * LD1 + c8 * 8) = gload<bf16x8>(src + (size_t)row * 512 + c8 * 8);
            }
        }
        __syncthreads();
        mma_kloop_h<1, NP, true, KC / 16>(acc, R2, LD1, P1, h2half, NE, wave, ch * (KC / 16), lane);
    }
    const int p = lane & 31, h = lane >> 5;
    const float bp = gload<float>(P.up4_b + n0 + p);
#pragma unroll
    for (int qd = 0; qd < 4; ++qd)
        store_split4<NP>(R0 + p * LD0 + wave * 32 + 8 * qd + 4 * h, P0, gelu_fast(acc[0][4 * qd + 0] + bp), gelu_fast(acc[0][4 * qd + 1] + bp),
                         gelu_fast(acc[0][4 * qd + 2] + bp), gelu_fast(acc[0][4 * qd + 3] + bp));
}

// ---- once per call: the text half of combine_extraction's pre-activation, ctext (B, 1024, 128) fp32 -----------------------
// kb0 = NE / 16: the text half (columns 128..255 of combine_extraction), one row of h2x per chain, once per call;
// kb0 = 0: the TIMESTEP half (columns 0..127) for one row of h2 per STEP -- multi-step calls run every chain of the batch at the
// same timestep, so W_comb[:, :128] . GELU(W4 . h2t + b4) is the same (1024, 128) tile for all of them: evaluated once per step
// here instead of once per (chain, step) inside denoise_h_kernel (which runs the identical instruction sequence when the
// timesteps differ per sample, so the two paths agree bit for bit).
template <int NP>
__global__ __launch_bounds__(NT) void ctext_kernel(DenoiseHParams P, const __bf16* __restrict__ h2x, float* __restrict__ ctext, int kb0) {
    __shared__ __attribute__((aligned(16))) __bf16 smem[NP * (P0 + P1)];
    __bf16* R0 = smem;
    __bf16* R2 = R0 + NP * P0;
    int b, tile;
    xcd_cloud_tile(kPoints / TM, b, tile);
    const int n0 = tile * TM, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    upsample_half_h<NP>(P, n0, h2x + (size_t)b * 32 * NE * NP * 16, R0, R2);
    __syncthreads();
    f32x16 acc[1];
    acc_zero_h(acc);
    mma_kloop_h<1, NP, true, NE / 16>(acc, R0, LD0, P0, P.comb.w, 128, wave, kb0, lane);
    const int p = lane & 31, h = lane >> 5;
    float* dst = ctext + ((size_t)b * kPoints + n0 + p) * 128 + wave * 32 + 4 * h;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = {acc[0][4 * q + 0], acc[0][4 * q + 1], acc[0][4 * q + 2], acc[0][4 * q + 3]};
        *reinterpret_cast<f32x4*>(dst + 8 * q) = v;
    }
}

// per-point denoiser on the 32x3 tile zt (LDS fp32) -> res (32x3 fp32); S = the emb half of combination_extraction.0 (+ bias)
template <int NP>
__device__ __forceinline__ void denoiser_pass_h(const DenoiseHParams& P, const float* zt, const f32x16 (&S)[2], __bf16* R0, __bf16* R2,
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
            v[j] = sigmoid_fast(fmaf(z2, gload<float>(w + 2), fmaf(z1, gload<float>(w + 1), z0 * gload<float>(w))) + gload<float>(P.pose0.b + j0 + j));
        }
        store_split4<NP>(R3 + row * LD3 + j0, P3, v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
    layer_lds_h<NW, TM, 64, 128, ACT_SIGMOID, NP>(R3, LD3, P3, P.pose2, R2, LD1, P1);   // pose (128) -> R2
    __syncthreads();
    // combination_extraction.0: [pose (R2) || emb] 256 -> 192 sigmoid -> R0; the emb half is S
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int cb = wave + i * NW;
        if (cb < 6) {
            f32x16 acc[1];
            acc[0] = S[i];
            mma_kloop_h<1, NP, true, 8>(acc, R2, LD1, P1, P.cex0.w, 192, cb, 0, lane);
            store_lds_nb_h<ACT_SIGMOID, NP>(acc[0], cb, R0, LD0, P0, lane);
        }
    }
    __syncthreads();
    layer_lds_h<NW, TM, 192, 128, ACT_SIGMOID, NP>(R0, LD0, P0, P.cex2, R2, LD1, P1);
    __syncthreads();
    layer_lds_h<2, TM, 128, 64, ACT_GELU, NP>(R2, LD1, P1, P.fin0, R3, LD3, P3);   // two 32-channel blocks: the first two waves
    __syncthreads();
    // pose_final.2: 64 -> 3 GELU on the matrix pipe (fin2m = the three rows zero-padded to one 32-channel block)
    if (wave == 0) {
        f32x16 a3[1];
        acc_zero_h(a3);
        mma_kloop_h<1, NP, false, 64 / 16>(a3, R3, LD3, P3, P.fin2m.w, 32, 0, 0, lane);
        const int c = lane & 31;
        if (c < 3) {
            const float bv = gload<float>(P.fin2.b + c);
#pragma unroll
            for (int r = 0; r < 16; ++r) res[acc_row(r, lane) * 3 + c] = gelu_fast(a3[0][r] + bv);
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void launder_params(DenoiseHParams& P) {
    P.up4_h = opaque_ptr(P.up4_h); P.up4_l = opaque_ptr(P.up4_l); P.up4_b = opaque_ptr(P.up4_b);
    P.comb.w = opaque_ptr(P.comb.w); P.comb.b = opaque_ptr(P.comb.b);
    P.pose2.w = opaque_ptr(P.pose2.w); P.pose2.b = opaque_ptr(P.pose2.b);
    P.cex0.w = opaque_ptr(P.cex0.w); P.cex0.b = opaque_ptr(P.cex0.b);
    P.cex2.w = opaque_ptr(P.cex2.w); P.cex2.b = opaque_ptr(P.cex2.b);
    P.fin0.w = opaque_ptr(P.fin0.w); P.fin0.b = opaque_ptr(P.fin0.b);
    P.fin2m.w = opaque_ptr(P.fin2m.w);
    P.pose0.w = opaque_ptr(P.pose0.w); P.pose0.b = opaque_ptr(P.pose0.b);
    P.fin2.b = opaque_ptr(P.fin2.b);
}

// h2t: the timestep half of the second up-sampling activation: row (s * B + b) of step s, sample b when h2t_per_sample, row s
// otherwise (multi-step calls: every chain of the batch is at the same timestep), 32 * 128 * NP * 16 elements per row;
// t: (nsteps, B); noise: (nsteps, B, 1024, 3).  out_sample != NULL: the sample of the (single) step goes there and x keeps x + g (Q5);
// out_sample == NULL (sampling modes): a chain of nsteps steps whose final sample is written to x.
template <int NP>
__global__ __launch_bounds__(NT, 2) void denoise_h_kernel(DenoiseHParams P0_, StepCoefs sc, int mode, float eta, int B, int nsteps, int h2t_per_sample,
                                                       float* __restrict__ x, const int64_t* __restrict__ t, const float* __restrict__ mask,
                                                       const float* __restrict__ pcd, const float* __restrict__ w,
                                                       const float* __restrict__ o12, const float* __restrict__ hm,
                                                       const __bf16* __restrict__ h2t, const float* __restrict__ ctext,
                                                       const float* __restrict__ ctime,
                                                       const float* __restrict__ noise, float* __restrict__ out_x0,
                                                       float* __restrict__ out_guiding, float* __restrict__ out_sample) {
    __shared__ __attribute__((aligned(16))) __bf16 smem[NP * (P0 + 2 * P1 + P3)];
    __shared__ float fsm[4 * TM * 3 + 32];
    __bf16* R0 = smem;
    __bf16* R1 = R0 + NP * P0;
    __bf16* R2 = R1 + NP * P1;
    __bf16* R3 = R2 + NP * P1;
    float* xt = fsm;                  // mutated x tile (32x3)
    float* gt = xt + TM * 3;          // guiding-point tile
    float* res = gt + TM * 3;         // denoiser output tile
    float* xs = res + TM * 3;         // the tile's current sample between steps
    float* oterm = xs + TM * 3;       // W15[c, 3:15] . o[b,o',:]   (27 values)
    int b, tile;
    xcd_cloud_tile(kPoints / TM, b, tile);   // the tiles of one chain share its h2 / pcd rows: keep them on one XCD
    const int n0 = tile * TM;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float* pcd_b = pcd + (size_t)b * kSlots * kPoints * 3;
    const float* w_b = w + (size_t)b * kSlots;
    const size_t gi = ((size_t)b * kPoints + n0) * 3 + tid;   // this thread's element of the tile (tid < 96)

    // ---- guidance g (fp32, as denoise.hip): depends on the scene conditioning only ---------------------------------------
    if (tid < kSlots * 3) {
        const int op = tid / 3, c = tid - op * 3;
        float s = 0.0f;
        for (int k = 0; k < 12; ++k) s = fmaf(o12[((size_t)b * kSlots + op) * 12 + k], P0_.pwt.w[c * 15 + 3 + k], s);
        oterm[tid] = s;
    }
    __syncthreads();
    if (tid < TM * 3) {
        const int row = tid / 3, c = tid - row * 3, n = n0 + row;
        float acc = 0.0f;
        for (int op = 0; op < kSlots; ++op) {
            const float p0 = scrambled_h(pcd_b, w_b, op, n, 0), p1 = scrambled_h(pcd_b, w_b, op, n, 1), p2 = scrambled_h(pcd_b, w_b, op, n, 2);
            float s = fmaf(p2, P0_.pwt.w[c * 15 + 2], fmaf(p1, P0_.pwt.w[c * 15 + 1], p0 * P0_.pwt.w[c * 15]));
            s = gelu_fast(s + oterm[op * 3 + c] + P0_.pwt.b[c]);
            // (N,-1,B,O) view: the multiplier of flat element f is mask.flatten()[f % (B*O)]  (Q2)
            const long long f = (((long long)b * kSlots + op) * kPoints + n) * 3 + c;
            acc += s * mask[f % ((long long)B * kSlots)];
        }
        gt[tid] = (acc + hm[gi]) / 2.0f;
        xs[tid] = x[gi];
    }

    const bool chain = out_sample == nullptr && mode != MODE_FORWARD;   // hoisted groups: the sample stays in the tile, the final one lands in x
#pragma unroll 1
    for (int step = 0; step < nsteps; ++step) {
        const bool last = step == nsteps - 1;
        // the layer operands are the same every step: without an opaque copy the compiler hoists their (step-invariant)
        // fragment loads out of this loop and parks them in ~500 registers
        DenoiseHParams P = P0_;
        launder_params(P);
        // in-place update of x (Q5); the caller's tensor sees it when the launch is a single step
        if (tid < TM * 3) {
            const float xv = xs[tid] + gt[tid];
            xt[tid] = xv;
            if (!chain) x[gi] = xv;
        }
        // ---- emb = GELU(W_comb[:, :128] . e_t + (ctext + b)) -> R1.  ctime != NULL: the timestep product of this step comes from
        // ctext_kernel(kb0 = 0) (one (1024, 128) tile per step, shared by the chains); otherwise it is evaluated here per sample
        const int p = lane & 31, h = lane >> 5;
        {
            f32x16 acc[1];
            if (ctime) {
                const float* cm = ctime + ((size_t)step * kPoints + n0 + p) * 128 + wave * 32 + 4 * h;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(cm + 8 * q);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[0][4 * q + j] = v[j];
                }
            } else {
                upsample_half_h<NP>(P, n0, h2t + (size_t)(h2t_per_sample ? step * B + b : step) * 32 * NE * NP * 16, R0, R2);
                acc_zero_h(acc);
                __syncthreads();
                mma_kloop_h<1, NP, true, NE / 16>(acc, R0, LD0, P0, P.comb.w, 128, wave, 0, lane);
            }
            const float* ct = ctext + ((size_t)b * kPoints + n0 + p) * 128 + wave * 32 + 4 * h;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 cv = *reinterpret_cast<const f32x4*>(ct + 8 * q);
                const f32x4 bv = gload<f32x4>(P.comb.b + wave * 32 + 8 * q + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[0][4 * q + j] += cv[j] + bv[j];
            }
            store_lds_nb_h<ACT_GELU, NP>(acc[0], wave, R1, LD1, P1, lane);
        }
        __syncthreads();
        // ---- emb half of combination_extraction.0 (+ bias), shared by the two passes ---------------------------------------
        f32x16 S[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int cb = wave + i * NW;
            if (cb < 6) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 bv = gload<f32x4>(P.cex0.b + cb * 32 + 8 * q + 4 * h);
#pragma unroll
                    for (int j = 0; j < 4; ++j) S[i][4 * q + j] = bv[j];
                }
                f32x16 acc[1];
                acc[0] = S[i];
                mma_kloop_h<1, NP, true, 8>(acc, R1, LD1, P1, P.cex0.w, 192, cb, 8, lane);
                S[i] = acc[0];
            }
        }
        // ---- denoiser on the mutated x + sampler update ----------------------------------------------------------------------
        denoiser_pass_h<NP>(P, xt, S, R0, R2, R3, res);
        if (tid < TM * 3) {
            const float x0 = res[tid];
            if (last) out_x0[gi] = x0;
            if (mode != MODE_FORWARD) {
                const int tb = (int)t[(size_t)step * B + b];
                const float xm = xt[tid], nz = tb != 0 ? 1.0f : 0.0f;
                const float nv = noise ? noise[(size_t)step * B * kPoints * 3 + gi] : 0.0f;
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
                xs[tid] = smp;
                if (last) (chain ? x : out_sample)[gi] = smp;
            }
        }
        __syncthreads();
        // ---- denoiser on the guiding points (saved_guiding_points, Q12): defined by the last forward ----------------------------
        if (last) {
            denoiser_pass_h<NP>(P, gt, S, R0, R2, R3, res);
            if (tid < TM * 3) out_guiding[gi] = res[tid];
        }
    }
}

void launch_ctext(int np, const DenoiseHParams& P, int rows, const void* h2x, float* ctext, bool time_half, hipStream_t st) {
    const __bf16* h2 = static_cast<const __bf16*>(h2x);
    const int kb0 = time_half ? 0 : NE / 16;
    if (np == 2) hipLaunchKernelGGL(ctext_kernel<2>, dim3(kPoints / TM * rows), dim3(NT), 0, st, P, h2, ctext, kb0);
    else hipLaunchKernelGGL(ctext_kernel<1>, dim3(kPoints / TM * rows), dim3(NT), 0, st, P, h2, ctext, kb0);
}

void launch_denoise_h(int np, const DenoiseHParams& P, const StepCoefs& sc, int mode, float eta, int B, int nsteps, int h2t_per_sample, float* x, const int64_t* t,
                      const float* mask, const float* pcd, const float* w, const float* o12, const float* hm, const void* h2t,
                      const float* ctext, const float* ctime, const float* noise, float* out_x0, float* out_guiding, float* out_sample, hipStream_t st) {
    const __bf16* h2 = static_cast<const __bf16*>(h2t);
    if (np == 2)
        hipLaunchKernelGGL(denoise_h_kernel<2>, dim3(kPoints / TM * B), dim3(NT), 0, st, P, sc, mode, eta, B, nsteps, h2t_per_sample, x, t, mask, pcd, w, o12, hm,
                           h2, ctext, ctime, noise, out_x0, out_guiding, out_sample);
    else
        hipLaunchKernelGGL(denoise_h_kernel<1>, dim3(kPoints / TM * B), dim3(NT), 0, st, P, sc, mode, eta, B, nsteps, h2t_per_sample, x, t, mask, pcd, w, o12, hm,
                           h2, ctext, ctime, noise, out_x0, out_guiding, out_sample);
}

}  // namespace lsdm
