// Conditioning-side kernels of SceneDiffusionModel.forward (model/sdm.py:141-195), gfx950.
//
//   cond_kernel        steps 1-4, 8, 10 of SURVEY.md §3.2: timestep MLP, text MLP, category head, category
//                      embedding, object-attention weights (with the mask-order quirk Q3/Q4), translation tokens.
//   posa_kernel        POSA Decoder with seq_length = 1 (posa/posa_models.py:310-326): per-point MLP with
//                      GroupNorm(8,64) over the whole sample; one 1024-thread workgroup per sample, one point
//                      per thread, activations never leave registers.
//   upsample12_kernel  first two layers of upsampling_layer (model/sdm.py:104-111,164-167) on the (B*256) rows of
//                      [ets || enc]; the result is written directly in the packed B-operand layout consumed by
//                      the third layer inside denoise_kernel (denoise.hip).
//   scene_attn_kernel  pcd_attention collapsed to one query row per (b, o) (Q7): 12 heads of dim 1 over 1024 keys
//                      read through the raw-reshape scramble Q1.
#include "mlp_core.hpp"
#include "kernels.hpp"

namespace lsdm {

// out[j] = act(W[j,:] . in + b[j]) for j < N; `in` lives in LDS, one output per thread (strided)
template <int ACT>
__device__ __forceinline__ void dense_vec(const float* in, int K, Lin l, float* out, int N, float scale = 1.0f) {
    for (int j = threadIdx.x; j < N; j += blockDim.x) {
        const float* w = l.w + (size_t)j * K;
        float s = 0.0f;
        for (int k = 0; k < K; ++k) s = fmaf(in[k], w[k], s);
        out[j] = apply_act<ACT>(s + l.b[j]) * scale;
    }
}

__global__ __launch_bounds__(256) void cond_kernel(CondParams P, const int64_t* __restrict__ t, const float* __restrict__ text,
                                                   const float* __restrict__ given_cats, const float* __restrict__ mask, int B,
                                                   int C, float* __restrict__ out_cat, float* __restrict__ e0,
                                                   float* __restrict__ tr_out, float* __restrict__ w_out) {
    __shared__ float sa[512], sb[512], enc[128], ecat[kSlots * 32], trh[kSlots * 128];
    __shared__ float q[128], kk[kSlots * 128], logit[8 * kSlots];
    // A launch may cover G consecutive steps of the same batch (grid = G*B): block r = step*B + b reads the conditioning
    // of sample b and step `step`'s timestep, and writes row r of e0 / tr / w (the front sets are contiguous).
    const int r = blockIdx.x, b = r % B, tid = threadIdx.x;
    // 1. timestep embedding: pe[t] -> Linear -> SiLU -> Linear  (diffusion_utils.py:14-21)
    const float* pe = P.pe + (size_t)t[r] * 128;
    if (tid < 128) sa[tid] = pe[tid];
    __syncthreads();
    for (int j = tid; j < 128; j += 256) {
        const float* w = P.time0.w + j * 128;
        float s = 0.0f;
        for (int k = 0; k < 128; ++k) s = fmaf(sa[k], w[k], s);
        sb[j] = siluf(s + P.time0.b[j]);
    }
    __syncthreads();
    dense_vec<ACT_NONE>(sb, 128, P.time2, e0 + (size_t)r * 256, 128);
    // 2. text embedding 512 -> 256 -> 256 -> 128, GELU after every layer (Q9)
    __syncthreads();
    for (int k = tid; k < 512; k += 256) sa[k] = text[(size_t)b * 512 + k];
    __syncthreads();
    dense_vec<ACT_GELU>(sa, 512, P.text0, sb, 256);
    __syncthreads();
    dense_vec<ACT_GELU>(sb, 256, P.text2, sa, 256);
    __syncthreads();
    dense_vec<ACT_GELU>(sa, 256, P.text4, enc, 128);
    __syncthreads();
    if (tid < 128) e0[(size_t)r * 256 + 128 + tid] = enc[tid];
    // 3. category head 128 -> 64 -> 32 -> C, GELU on the logits, softmax (Q10)
    dense_vec<ACT_GELU>(enc, 128, P.pcat0, sa, 64);
    __syncthreads();
    dense_vec<ACT_GELU>(sa, 64, P.pcat2, sb, 32);
    __syncthreads();
    dense_vec<ACT_GELU>(sb, 32, P.pcat4, sa, C);
    __syncthreads();
    if (tid == 0) {
        float m = sa[0];
        for (int j = 1; j < C; ++j) m = fmaxf(m, sa[j]);
        float s = 0.0f;
        for (int j = 0; j < C; ++j) { sb[j] = expf(sa[j] - m); s += sb[j]; }
        for (int j = 0; j < C; ++j) out_cat[(size_t)b * C + j] = sb[j] / s;
    }
    // 4. category embedding (O x C) -> (O x 32)
    for (int e = tid; e < kSlots * 32; e += 256) {
        const int o = e >> 5, j = e & 31;
        const float* in = given_cats + ((size_t)b * kSlots + o) * C;
        const float* w = P.ecat.w + (size_t)j * C;
        float s = 0.0f;
        for (int k = 0; k < C; ++k) s = fmaf(in[k], w[k], s);
        ecat[e] = gelu(s + P.ecat.b[j]);
    }
    __syncthreads();
    // 10. translation tokens: [ecat || enc] (160) -> 128 -> 12, GELU each  (sdm.py:185-187)
    for (int e = tid; e < kSlots * 128; e += 256) {
        const int o = e >> 7, j = e & 127;
        const float* w = P.tr0.w + (size_t)j * 160;
        float s = 0.0f;
        for (int k = 0; k < 32; ++k) s = fmaf(ecat[o * 32 + k], w[k], s);
        for (int k = 0; k < 128; ++k) s = fmaf(enc[k], w[32 + k], s);
        trh[e] = gelu(s + P.tr0.b[j]);
    }
    // 8. object attention: q = (Wq enc + bq)/4 ; k_o = Wk ecat_o + bk   (8 heads x 16)
    for (int j = tid; j < 128; j += 256) {
        const float* w = P.attn_wq + (size_t)j * 128;
        float s = 0.0f;
        for (int k = 0; k < 128; ++k) s = fmaf(enc[k], w[k], s);
        q[j] = (s + P.attn_inb[j]) * 0.25f;
    }
    for (int e = tid; e < kSlots * 128; e += 256) {
        const int o = e >> 7, j = e & 127;
        const float* w = P.attn_wk + (size_t)j * 32;
        float s = 0.0f;
        for (int k = 0; k < 32; ++k) s = fmaf(ecat[o * 32 + k], w[k], s);
        kk[e] = s + P.attn_inb[128 + j];
    }
    __syncthreads();
    for (int e = tid; e < kSlots * 12; e += 256) {
        const int o = e / 12, j = e - o * 12;
        const float* w = P.tr2.w + (size_t)j * 128;
        float s = 0.0f;
        for (int k = 0; k < 128; ++k) s = fmaf(trh[o * 128 + k], w[k], s);
        tr_out[((size_t)r * kSlots + o) * 12 + j] = gelu(s + P.tr2.b[j]);
    }
    if (tid < 8 * kSlots) {
        const int h = tid / kSlots, o = tid - h * kSlots;
        float s = 0.0f;
        for (int d = 0; d < 16; ++d) s = fmaf(q[h * 16 + d], kk[o * 128 + h * 16 + d], s);
        // float mask ADDED to the logits (Q4); the (h,b)-ordered repeat makes row (b,h) use mask[(b*8+h) % B] (Q3)
        logit[tid] = s + mask[(size_t)((b * 8 + h) % B) * kSlots + o];
    }
    __syncthreads();
    if (tid < kSlots) {
        float acc = 0.0f;
        for (int h = 0; h < 8; ++h) {
            float m = logit[h * kSlots];
            for (int o = 1; o < kSlots; ++o) m = fmaxf(m, logit[h * kSlots + o]);
            float s = 0.0f;
            for (int o = 0; o < kSlots; ++o) s += expf(logit[h * kSlots + o] - m);
            acc += expf(logit[h * kSlots + tid] - m) / s;
        }
        w_out[(size_t)r * kSlots + tid] = acc / 8.0f;
    }
}

// ---------------------------------------------------------------------------------------------
// block-wide sum of 8 per-group partials; result broadcast to all threads through LDS
__device__ __forceinline__ void block_sum8(float (&v)[8], float (*red)[8], float* tot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        float s = v[g];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) red[wave][g] = s;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        float s = 0.0f;
        for (int w = 0; w < 16; ++w) s += red[w][threadIdx.x];
        tot[threadIdx.x] = s;
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 8; ++g) v[g] = tot[g];
    __syncthreads();
}

// GroupNorm(8, 64) over (8 channels x count points) + ReLU, in place on the thread's 64 channels
__device__ __forceinline__ void group_norm_relu(float (&h)[64], bool active, float count, const float* __restrict__ gw,
                                                const float* __restrict__ gb, float (*red)[8], float* tot) {
    float s[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        float a = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) a += h[g * 8 + k];
        s[g] = active ? a : 0.0f;
    }
    block_sum8(s, red, tot);
    float mean[8], v[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        mean[g] = s[g] / count;
        float a = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float d = h[g * 8 + k] - mean[g]; a = fmaf(d, d, a); }
        v[g] = active ? a : 0.0f;
    }
    block_sum8(v, red, tot);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const float rstd = 1.0f / sqrtf(v[g] / count + 1e-5f);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int ch = g * 8 + k;
            h[ch] = reluf((h[ch] - mean[g]) * rstd * gw[ch] + gb[ch]);
        }
    }
}

__global__ __launch_bounds__(1024) void posa_kernel(PosaParams P, const float* __restrict__ given_objs, float* __restrict__ hm) {
    __shared__ float red[16][8];
    __shared__ float tot[8];
    const int b = blockIdx.x, n = threadIdx.x;
    const float* v = given_objs + ((size_t)b * kSlots * kPoints + n) * 3;  // slot 0 = human
    const float x0 = v[0], x1 = v[1], x2 = v[2];
    float h[64], g[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        const float* w = P.lin[0].w + j * 3;
        h[j] = fmaf(x2, w[2], fmaf(x1, w[1], x0 * w[0])) + P.lin[0].b[j];
    }
    group_norm_relu(h, true, 8.0f * kPoints, P.gn_w[0], P.gn_b[0], red, tot);
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        const float* w = P.lin[1].w + j * 64;
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 64; ++k) s = fmaf(h[k], w[k], s);
        g[j] = s + P.lin[1].b[j];
    }
    group_norm_relu(g, true, 8.0f * kPoints, P.gn_w[1], P.gn_b[1], red, tot);
    // third block runs on the first 655 vertices only (index_select with the identity spiral)
    const bool act = n < 655;
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        const float* w = P.lin[2].w + j * 64;
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 64; ++k) s = fmaf(g[k], w[k], s);
        h[j] = s + P.lin[2].b[j];
    }
    group_norm_relu(h, act, 8.0f * 655.0f, P.gn_w[2], P.gn_b[2], red, tot);
    // Linear 64->3, nearest x2 along the vertex axis, cut to 1024 rows: vertex i < 512 feeds rows 2i, 2i+1
    if (n < kPoints / 2) {
        float o[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* w = P.last.w + c * 64;
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < 64; ++k) s = fmaf(h[k], w[k], s);
            o[c] = s + P.last.b[c];
        }
        float* dst = hm + ((size_t)b * kPoints + 2 * n) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) { dst[c] = o[c]; dst[3 + c] = o[c]; }
    }
}

// ---------------------------------------------------------------------------------------------
// rows r = b*256 + c carry the scalar e0[r]; h1 = GELU(w1*e + b1) (128), h2 = GELU(W2 h1 + b2) (512)
// output packed as the per-sample "weight" operand of the third layer inside the denoise kernel:
//   NP = 0 : fp32  h2p[b][kb = k/8][c][k%8]                  (denoise_kernel, mlp_core.hpp)
//   NP = 1,2: bf16 h2p[b][kb = k/16][c][plane][k%16]         (denoise_h_kernel, bf16_core.hpp; plane 1 = lo part)
template <int NP>
__global__ __launch_bounds__(256) void upsample12_kernel(const float* __restrict__ e0, Lin up0, Dense up2, void* __restrict__ h2p_v) {
    constexpr int TM = 64, LDA = lds_ld(128);
    __shared__ __attribute__((aligned(16))) float bufA[TM * LDA];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r0 = blockIdx.x * TM;           // 64 consecutive channels c of one sample
    const int b = r0 >> 8, c0 = r0 & 255;
    for (int e = tid; e < TM * 128; e += 256) {
        const int row = e >> 7, j = e & 127;
        bufA[row * LDA + j] = gelu(fmaf(e0[r0 + row], up0.w[j], up0.b[j]));
    }
    __syncthreads();
    using TS = TileSplit<4, TM, 512>;
    for (int cb = wave; cb < TS::CB; cb += 4) {
        f32x16 acc[TS::RB];
        acc_zero(acc);
        mma_kloop<TS::RB>(acc, bufA, LDA, up2.w, 512, cb, 0, 128 / 8, 0, lane);
        const int k = cb * 32 + (lane & 31);
        const float bv = up2.b[k];
#pragma unroll
        for (int rb = 0; rb < TS::RB; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = c0 + rb * 32 + acc_row(r, lane);
                const float v = gelu(acc[rb][r] + bv);
                if constexpr (NP == 0) {
                    static_cast<float*>(h2p_v)[(((size_t)b * 64 + (k >> 3)) * 256 + c) * 8 + (k & 7)] = v;
                } else {
                    __bf16* dst = static_cast<__bf16*>(h2p_v) + ((((size_t)b * 32 + (k >> 4)) * 256 + c) * NP) * 16 + (k & 15);
                    const __bf16 hi = (__bf16)v;
                    dst[0] = hi;
                    if constexpr (NP == 2) dst[16] = (__bf16)(v - (float)hi);
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------
// value of the scrambled tensor P[b, o', p', c'] (Q1): flat f' in (O,N,3) == flat in (3N,O)
__device__ __forceinline__ float scrambled_P(const float* __restrict__ pcd_b, const float* __restrict__ w_b, int op, int pp, int cp) {
    const int f = (op * kPoints + pp) * 3 + cp;
    const int j = f / kSlots, o = f - j * kSlots;
    return pcd_b[o * (kPoints * 3) + j] * w_b[o];
}

__global__ __launch_bounds__(256) void scene_attn_kernel(SceneAttnParams P, const float* __restrict__ pcd, const float* __restrict__ w,
                                                         const float* __restrict__ tr, float* __restrict__ o_out) {
    __shared__ float q2[12], redm[4][12], reds[4][24], fin[24], osum[12];
    const int bo = blockIdx.x, b = bo / kSlots, op = bo - b * kSlots;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* pcd_b = pcd + (size_t)b * kSlots * kPoints * 3;
    const float* w_b = w + (size_t)b * kSlots;
    if (tid < 12) {
        float s = 0.0f;
        for (int k = 0; k < 12; ++k) s = fmaf(tr[(size_t)bo * 12 + k], P.wq[tid * 12 + k], s);
        q2[tid] = s + P.inb[tid];   // head_dim 1 -> scale 1
    }
    __syncthreads();
    float lg[4][12], vv[4][12];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = tid + 256 * i;
        const float p0 = scrambled_P(pcd_b, w_b, op, p, 0), p1 = scrambled_P(pcd_b, w_b, op, p, 1), p2 = scrambled_P(pcd_b, w_b, op, p, 2);
#pragma unroll
        for (int h = 0; h < 12; ++h) {
            const float kh = fmaf(p2, P.wk[h * 3 + 2], fmaf(p1, P.wk[h * 3 + 1], p0 * P.wk[h * 3])) + P.inb[12 + h];
            vv[i][h] = fmaf(p2, P.wv[h * 3 + 2], fmaf(p1, P.wv[h * 3 + 1], p0 * P.wv[h * 3])) + P.inb[24 + h];
            lg[i][h] = q2[h] * kh;
        }
    }
    // per-head max over the 1024 keys
#pragma unroll
    for (int h = 0; h < 12; ++h) {
        float m = fmaxf(fmaxf(lg[0][h], lg[1][h]), fmaxf(lg[2][h], lg[3][h]));
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        if (lane == 0) redm[wave][h] = m;
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 12; ++h) {
        const float m = fmaxf(fmaxf(redm[0][h], redm[1][h]), fmaxf(redm[2][h], redm[3][h]));
        float den = 0.0f, num = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float e = expf(lg[i][h] - m);
            den += e;
            num = fmaf(e, vv[i][h], num);
        }
        for (int off = 32; off > 0; off >>= 1) { den += __shfl_xor(den, off); num += __shfl_xor(num, off); }
        if (lane == 0) { reds[wave][h] = den; reds[wave][12 + h] = num; }
    }
    __syncthreads();
    if (tid < 24) fin[tid] = reds[0][tid] + reds[1][tid] + reds[2][tid] + reds[3][tid];
    __syncthreads();
    if (tid < 12) osum[tid] = fin[12 + tid] / fin[tid];
    __syncthreads();
    if (tid < 12) {
        float s = 0.0f;
        for (int k = 0; k < 12; ++k) s = fmaf(osum[k], P.outp.w[tid * 12 + k], s);
        o_out[(size_t)bo * 12 + tid] = s + P.outp.b[tid];
    }
}

// ---------------------------------------------------------------------------------------------
void launch_cond(const CondParams& P, const int64_t* t, const float* text, const float* given_cats, const float* mask, int B, int G, int C,
                 float* out_cat, float* e0, float* tr, float* w, hipStream_t st) {
    hipLaunchKernelGGL(cond_kernel, dim3(B * G), dim3(256), 0, st, P, t, text, given_cats, mask, B, C, out_cat, e0, tr, w);
}
void launch_posa(const PosaParams& P, const float* given_objs, int B, float* hm, hipStream_t st) {
    hipLaunchKernelGGL(posa_kernel, dim3(B), dim3(1024), 0, st, P, given_objs, hm);
}
void launch_upsample12(int np, const float* e0, Lin up0, const float* up2_wp, const float* up2_b, int B, void* h2p, hipStream_t st) {
    const dim3 grid(B * 256 / 64), block(256);
    const Dense up2{up2_wp, up2_b};
    if (np == 0) hipLaunchKernelGGL(upsample12_kernel<0>, grid, block, 0, st, e0, up0, up2, h2p);
    else if (np == 1) hipLaunchKernelGGL(upsample12_kernel<1>, grid, block, 0, st, e0, up0, up2, h2p);
    else hipLaunchKernelGGL(upsample12_kernel<2>, grid, block, 0, st, e0, up0, up2, h2p);
}
void launch_scene_attn(const SceneAttnParams& P, const float* pcd, const float* w, const float* tr, int B, float* o, hipStream_t st) {
    hipLaunchKernelGGL(scene_attn_kernel, dim3(B * kSlots), dim3(256), 0, st, P, pcd, w, tr, o);
}

}  // namespace lsdm
