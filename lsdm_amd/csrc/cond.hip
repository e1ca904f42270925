// Conditioning-side kernels of SceneDiffusionModel.forward (model/sdm.py:141-195), gfx950.
//
//   cond_kernel        steps 1-4, 8, 10 of SURVEY.md §3.2: timestep MLP, text MLP, category head, category
//                      embedding, object-attention weights (with the mask-order quirk Q3/Q4), translation tokens.
//   posa_stage_kernel  POSA Decoder with seq_length = 1 (posa/posa_models.py:310-326): per-vertex MLP with
//                      GroupNorm(8,64) over the whole sample; three launches of 64-vertex tiles that re-evaluate
//                      the decoder up to their block and exchange only mergeable moment triples.
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
                                                   float* __restrict__ tr_out, float* __restrict__ w_out, int time_only, int tstride) {
    __shared__ float sa[1024], sb[512], enc[128], ecat[kSlots * 32], trh[kSlots * 128];
    __shared__ float q[128], kk[kSlots * 128], logit[8 * kSlots];
    // A launch may cover G consecutive steps of the same batch (grid = G*B): block r = step*B + b reads the conditioning
    // of sample b and step `step`'s timestep, and writes row r of e0 / tr / w (the front sets are contiguous).
    // time_only: row r = one STEP of a multi-step call whose chains all sit at the same timestep t[r * tstride] -- only the
    // timestep embedding (first half of row r of e0) is evaluated; everything else of the conditioning is per call.
    const int r = blockIdx.x, b = r % B, tid = threadIdx.x;
    // 1. timestep embedding: pe[t] -> Linear -> SiLU -> Linear  (diffusion_utils.py:14-21)
    const float* pe = P.pe + (size_t)t[(size_t)r * tstride] * 128;
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
    if (time_only) return;
    // 2. text embedding D -> D/2 -> 256 -> 128, GELU after every layer (Q9); D = clip_dim <= 1024
    __syncthreads();
    const int D = P.text_dim;
    for (int k = tid; k < D; k += 256) sa[k] = text[(size_t)b * D + k];
    __syncthreads();
    dense_vec<ACT_GELU>(sa, D, P.text0, sb, D / 2);
    __syncthreads();
    dense_vec<ACT_GELU>(sb, D / 2, P.text2, sa, 256);
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
// POSA decoder (posa/posa_models.py:310-326) in three launches of 64-vertex tiles.
//
// GroupNorm(8, 64) takes its statistics over (8 channels x all vertices) of a sample, so every block of the decoder needs a
// sample-wide reduction before it can normalise.  A sample is 1024 x 3 floats and a block is 4096 MAC per vertex: far
// cheaper to RE-EVALUATE than to synchronise or to spill.  Each launch recomputes the decoder from the vertices up to its own
// block (deterministic: every re-evaluation produces the same bits), reduces that block's pre-activations per tile with a
// mergeable (count, mean, M2) triple, and the NEXT launch merges the tiles of its sample in a fixed order (Chan et al.; as
// accurate as a two-pass evaluation).  Nothing but the triples (a few KB) goes through memory; the 64 -> 64 blocks run on the
// exact-fp32 matrix pipe (mlp_core.hpp) from LDS tiles.
//   stage A (16 tiles): block-1 statistics of the whole sample (each tile re-derives them from the 1024 vertices: 768 FMA per
//                       thread), blocks 1-2 on its tile -> block-2 triple of the tile
//   stage B (11 tiles): + block 3 on the first 655 vertices (index_select with the identity spiral) -> block-3 triple
//   stage C ( 8 tiles): + Linear 64 -> 3, nearest x2 along the vertex axis cut to 1024 rows: vertex i < 512 feeds rows 2i, 2i+1
struct Moments { float n, mean, m2; };
__device__ __forceinline__ Moments merge(Moments a, Moments b) {
    const float n = a.n + b.n;
    if (n == 0.0f) return Moments{0.0f, 0.0f, 0.0f};
    const float d = b.mean - a.mean, f = b.n / n;
    return Moments{n, fmaf(d, f, a.mean), a.m2 + b.m2 + d * d * a.n * f};
}
__device__ __forceinline__ Moments shfl_xor(Moments m, int off) {
    return Moments{__shfl_xor(m.n, off), __shfl_xor(m.mean, off), __shfl_xor(m.m2, off)};
}
// moments of v[0..cnt) (two-pass in registers)
template <int N>
__device__ __forceinline__ Moments local_moments(const float (&v)[N], bool live) {
    if (!live) return Moments{0.0f, 0.0f, 0.0f};
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < N; ++i) s += v[i];
    const float mean = s * (1.0f / N);
    float m2 = 0.0f;
#pragma unroll
    for (int i = 0; i < N; ++i) { const float d = v[i] - mean; m2 = fmaf(d, d, m2); }
    return Moments{(float)N, mean, m2};
}

constexpr int kPosaTile = 64, kPosaLD = 68;   // (LD / 4) odd: conflict-free ds_read_b128 of the MFMA A operand

template <int STAGE>
__global__ __launch_bounds__(256) void posa_stage_kernel(PosaParams P, const float* __restrict__ given_objs, float* __restrict__ stat1,
                                                         float* __restrict__ part2, float* __restrict__ part3, float* __restrict__ hm) {
    __shared__ __attribute__((aligned(16))) float bufA[kPosaTile * kPosaLD];
    __shared__ __attribute__((aligned(16))) float bufB[kPosaTile * kPosaLD];
    __shared__ float w1[64 * 3], b1[64], gnw[3][64], gnb[3][64], w4[3 * 64], mean_s[8], rstd_s[8];
    __shared__ Moments red[4][8];
    constexpr int TILES = STAGE == 0 ? 16 : (STAGE == 1 ? 11 : 8);
    const int b = blockIdx.x / TILES, tile = blockIdx.x - b * TILES, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* verts = given_objs + (size_t)b * kSlots * kPoints * 3;   // slot 0 = human
    for (int e = tid; e < 192; e += 256) { w1[e] = P.lin[0].w[e]; w4[e] = P.last.w[e]; }
    if (tid < 64) {
        b1[tid] = P.lin[0].b[tid];
#pragma unroll
        for (int i = 0; i < 3; ++i) { gnw[i][tid] = P.gn_w[i][tid]; gnb[i][tid] = P.gn_b[i][tid]; }
    }
    __syncthreads();
    // ---- block-1 statistics of the sample
    if constexpr (STAGE == 0) {
        // thread = (group g = tid & 7, vertex slot tid >> 3): 32 vertices x 8 channels of one group
        const int g = tid & 7, v0 = tid >> 3;
        Moments acc{0.0f, 0.0f, 0.0f};
        for (int it = 0; it < 4; ++it) {
            float vals[64];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float* x = verts + (size_t)(v0 + 32 * (q + 8 * it)) * 3;
                const float x0 = x[0], x1 = x[1], x2 = x[2];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int ch = g * 8 + k;
                    vals[q * 8 + k] = fmaf(x2, w1[ch * 3 + 2], fmaf(x1, w1[ch * 3 + 1], x0 * w1[ch * 3])) + b1[ch];
                }
            }
            acc = merge(acc, local_moments(vals, true));
        }
        // lanes with equal (tid & 7) hold the same group: xor over bits 3..5 inside the wave, then across the 4 waves
        for (int off = 8; off < 64; off <<= 1) acc = merge(acc, shfl_xor(acc, off));
        if (lane < 8) red[wave][lane] = acc;
        __syncthreads();
        if (tid < 8) {
            Moments m = merge(merge(red[0][tid], red[1][tid]), merge(red[2][tid], red[3][tid]));
            mean_s[tid] = m.mean;
            rstd_s[tid] = 1.0f / sqrtf(m.m2 / m.n + 1e-5f);
            if (tile == 0) { stat1[b * 16 + tid] = mean_s[tid]; stat1[b * 16 + 8 + tid] = rstd_s[tid]; }
        }
    } else {
        if (tid < 8) { mean_s[tid] = stat1[b * 16 + tid]; rstd_s[tid] = stat1[b * 16 + 8 + tid]; }
    }
    __syncthreads();
    // ---- block 1 on the tile: thread = (row = tid >> 2, 16 channels)
    {
        const int row = tid >> 2, c0 = (tid & 3) * 16;
        const float* x = verts + (size_t)(tile * kPosaTile + row) * 3;
        const float x0 = x[0], x1 = x[1], x2 = x[2];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int ch = c0 + k, g = ch >> 3;
            const float h = fmaf(x2, w1[ch * 3 + 2], fmaf(x1, w1[ch * 3 + 1], x0 * w1[ch * 3])) + b1[ch];
            bufA[row * kPosaLD + ch] = reluf((h - mean_s[g]) * rstd_s[g] * gnw[0][ch] + gnb[0][ch]);
        }
    }
    __syncthreads();
    // ---- blocks 2 (and 3): Linear 64 -> 64 on the matrix pipe, GroupNorm from the merged tile triples, ReLU
    // statistics layout: thread = (group g = tid >> 5, r = tid & 31) owns rows r and r + 32, 8 channels: 16 values
    const int sg = tid >> 5, sr = tid & 31;
    float* src = bufA;
    float* dst = bufB;
#pragma unroll
    for (int blk = 1; blk <= (STAGE == 0 ? 1 : 2); ++blk) {
        layer_lds<4, kPosaTile, 64, 64, ACT_NONE>(src, kPosaLD, blk == 1 ? P.d1.w : P.d2.w, blk == 1 ? P.d1.b : P.d2.b, dst, kPosaLD);
        __syncthreads();
        const int valid = blk == 1 ? kPosaTile : min(kPosaTile, 655 - tile * kPosaTile);   // block 3 sees 655 vertices
        const bool mine = blk == (STAGE == 0 ? 1 : 2) && STAGE < 2;                        // this launch produces this block's triple
        float* part = blk == 1 ? part2 : part3;
        const int ntile = blk == 1 ? 16 : 11;
        if (mine) {
            Moments m{0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = dst[(sr + 32 * hh) * kPosaLD + sg * 8 + k];
                m = merge(m, local_moments(v, sr + 32 * hh < valid));
            }
            for (int off = 1; off < 32; off <<= 1) m = merge(m, shfl_xor(m, off));
            if (sr == 0) {
                float* o = part + ((size_t)(b * ntile + tile) * 8 + sg) * 3;
                o[0] = m.n; o[1] = m.mean; o[2] = m.m2;
            }
        } else {
            // merge the tiles of the sample in tile order (every work-group of the sample computes the same bits)
            if (tid < 8) {
                Moments m{0.0f, 0.0f, 0.0f};
                for (int t = 0; t < ntile; ++t) {
                    const float* o = part + ((size_t)(b * ntile + t) * 8 + tid) * 3;
                    m = merge(m, Moments{o[0], o[1], o[2]});
                }
                mean_s[tid] = m.mean;
                rstd_s[tid] = 1.0f / sqrtf(m.m2 / m.n + 1e-5f);
            }
            __syncthreads();
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int ch = sg * 8 + k, at = (sr + 32 * hh) * kPosaLD + ch;
                    dst[at] = reluf((dst[at] - mean_s[sg]) * rstd_s[sg] * gnw[blk][ch] + gnb[blk][ch]);
                }
            __syncthreads();
            float* t2 = src; src = dst; dst = t2;
        }
    }
    if constexpr (STAGE == 2) {
        // src holds ReLU(GN3(block 3)) of the tile's 64 vertices (all < 512): Linear 64 -> 3, rows 2i and 2i + 1
        if (tid < kPosaTile * 3) {
            const int row = tid / 3, c = tid - row * 3;
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < 64; ++k) s = fmaf(src[row * kPosaLD + k], w4[c * 64 + k], s);
            s += P.last.b[c];
            float* o = hm + ((size_t)b * kPoints + 2 * (tile * kPosaTile + row)) * 3 + c;
            o[0] = s;
            o[3] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// rows r = b*256 + c carry the scalar e0[r]; h1 = GELU(w1*e + b1) (128), h2 = GELU(W2 h1 + b2) (512)
// output packed as the per-sample "weight" operand of the third layer inside the denoise kernel:
//   NP = 0 : fp32  h2p[b][kb = k/8][c][k%8]                  (denoise_kernel, mlp_core.hpp)
//   NP = 1,2: bf16 h2p[b][kb = k/16][c][plane][k%16]         (denoise_h_kernel, bf16_core.hpp; plane 1 = lo part)
//   NC / c_off: the launch covers channels c_off .. c_off + NC of e0's 256 ([ets (128) || enc (128)]); the bf16 modes keep the
//   timestep half (per step) and the text half (per call, ctext_kernel) apart: h2p[b][k/16][c - c_off][plane][k%16], NC = 128
template <int NP, int NC>
__global__ __launch_bounds__(256) void upsample12_kernel(const float* __restrict__ e0, int c_off, Lin up0, Dense up2, void* __restrict__ h2p_v) {
    constexpr int TM = 64, LDA = lds_ld(128);
    __shared__ __attribute__((aligned(16))) float bufA[TM * LDA];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = blockIdx.x / (NC / TM), c0 = (blockIdx.x % (NC / TM)) * TM;   // 64 consecutive channels c of one row of e0
    const int r0 = b * 256 + c_off + c0;
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
                    __bf16* dst = static_cast<__bf16*>(h2p_v) + ((((size_t)b * 32 + (k >> 4)) * NC + c) * NP) * 16 + (k & 15);
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
                                                         const float* __restrict__ tr, float* __restrict__ o_out, int B) {
    __shared__ float q2[12], redm[4][12], reds[4][24], fin[24], osum[12];
    // (a launch over several steps of a group: block bo = (step * B + b) * 9 + op; w / tr are per SAMPLE, pcd / o per (step, sample))
    const int bo = blockIdx.x, b = bo / kSlots, op = bo - b * kSlots, bs = b % B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* pcd_b = pcd + (size_t)b * kSlots * kPoints * 3;
    const float* w_b = w + (size_t)bs * kSlots;
    if (tid < 12) {
        float s = 0.0f;
        for (int k = 0; k < 12; ++k) s = fmaf(tr[((size_t)bs * kSlots + op) * 12 + k], P.wq[tid * 12 + k], s);
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
    hipLaunchKernelGGL(cond_kernel, dim3(B * G), dim3(256), 0, st, P, t, text, given_cats, mask, B, C, out_cat, e0, tr, w, 0, 1);
}
void launch_cond_time(const CondParams& P, const int64_t* t, int tstride, int rows, float* e0, hipStream_t st) {
    hipLaunchKernelGGL(cond_kernel, dim3(rows), dim3(256), 0, st, P, t, nullptr, nullptr, nullptr, 1, 0, nullptr, e0, nullptr, nullptr, 1, tstride);
}
void launch_posa(const PosaParams& P, const float* given_objs, int B, float* ws, float* hm, hipStream_t st) {
    // ws: B * (16 + 16*24 + 11*24) floats: block-1 (mean, rstd) per sample, per-tile triples of blocks 2 and 3
    float* stat1 = ws;
    float* part2 = stat1 + (size_t)B * 16;
    float* part3 = part2 + (size_t)B * 16 * 24;
    hipLaunchKernelGGL(posa_stage_kernel<0>, dim3(B * 16), dim3(256), 0, st, P, given_objs, stat1, part2, part3, hm);
    hipLaunchKernelGGL(posa_stage_kernel<1>, dim3(B * 11), dim3(256), 0, st, P, given_objs, stat1, part2, part3, hm);
    hipLaunchKernelGGL(posa_stage_kernel<2>, dim3(B * 8), dim3(256), 0, st, P, given_objs, stat1, part2, part3, hm);
}
void launch_upsample12(int np, const float* e0, Lin up0, const float* up2_wp, const float* up2_b, int rows, int c_off, void* h2p, hipStream_t st) {
    const dim3 block(256);
    const Dense up2{up2_wp, up2_b};
    if (np == 0) hipLaunchKernelGGL((upsample12_kernel<0, 256>), dim3(rows * 4), block, 0, st, e0, 0, up0, up2, h2p);
    else if (np == 1) hipLaunchKernelGGL((upsample12_kernel<1, 128>), dim3(rows * 2), block, 0, st, e0, c_off, up0, up2, h2p);
    else hipLaunchKernelGGL((upsample12_kernel<2, 128>), dim3(rows * 2), block, 0, st, e0, c_off, up0, up2, h2p);
}
void launch_scene_attn(const SceneAttnParams& P, const float* pcd, const float* w, const float* tr, int B, float* o, hipStream_t st, int steps) {
    hipLaunchKernelGGL(scene_attn_kernel, dim3(steps * B * kSlots), dim3(256), 0, st, P, pcd, w, tr, o, B);
}

}  // namespace lsdm
