// Training objective of the path on MI355X: GaussianDiffusion.training_losses (diffusion/gaussian_diffusion.py:1256-1342) with its
// gradient -- q_sample, SceneDiffusionModel.forward (model/sdm.py:131-218) layer by layer with every activation kept, chamfer + category
// loss, and the reverse sweep through all of it into d loss / d theta of every state_dict parameter.
//
// Different shape from the sampling path on purpose: sampling fuses a whole stage into one kernel and keeps nothing; a training step has
// to keep every pre-activation, so the network is executed here as a TAPE of plain operators (dense layer, BatchNorm, GroupNorm, grouping
// gather, neighbour max, 3-NN interpolation, the two attention reductions, the Q1/Q2 scrambles ...), each a pair of hand-written HIP
// kernels (forward / backward) on channel-last fp32 tensors, recorded in execution order and replayed backwards.  Arithmetic is exact
// fp32 (FMA GEMM tiles through LDS): gradients are compared with torch.autograd of the CPU oracle to 1e-3 relative.  The PointNet++
// geometry (FPS, ball query, 3-NN) comes from the SAME kernels as the sampling path (geometry.hip) -- index-exact against the reference.
//
// Parameters are not copied: lsdm_train_bind hands over the device pointers of the caller's fp32 parameter tensors and of the buffers
// that receive their gradients, so an optimizer step on the caller's side is seen by the next lsdm_train_step without any upload.
// BatchNorm: flags & 1 selects training mode (model.train() of run/train_sdm.py:46: batch statistics, running statistics updated in
// place with momentum 0.1, unbiased variance); otherwise the running statistics are used (model.eval()).  Dropout(0.5) of the encoder
// head (pointnet2.py:58,76) is applied when a keep-mask is passed (the caller draws it: the reference draws it on the host generator).
#include "../../include/lsdm_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "kernels.hpp"

using namespace lsdm;

namespace {

constexpr int kO = 9, kN = 1024, kPosa = 655;
constexpr float kBnEps = 1e-5f, kGnEps = 1e-5f;

// ------------------------------------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------------------------------------
// C(m, n) (+)= sum_k A(m, k) B(k, n) [+ bias(n)], every operand through explicit element strides (a transposed or column-sliced view
// costs nothing), batched over blockIdx.z / ksplit, the reduction optionally split over `ksplit` work-groups that add their partial
// tiles atomically (weight gradients: tiny M x N, hundreds of thousands of rows to reduce over).  64 x 64 x 16 tiles through LDS,
// one 32 x 32 quadrant per wave on the exact-fp32 matrix instruction, the next k-block requested while this one is multiplied.
constexpr int kStatCopies = 32;
struct Gemm {
    const float* A; const float* B; float* C; const float* bias;
    int M, N, K;
    long long sam, sak, sbk, sbn, scm, scn;      // element strides
    long long bA, bB, bC;                         // batch strides
    int ksplit, kchunk;                           // reduction split (kchunk = K range of one work-group, multiple of 16)
    int mode;                                     // 0 store, 1 add (read-modify-write, exclusive), 2 atomic add
    // optional by-products (NULL = none):
    double* csum; double* csq;                    // += per-column sum / sum of squares of the values stored to C (mode 0: BatchNorm statistics),
                                                  //    kStatCopies private copies [copy][N] (copy = row tile % kStatCopies: the atomics of a column
                                                  //    of a million-row layer would otherwise queue on one address)
    float* asum;                                  // += sum_k A(m, k) per row m of A (the bias gradient next to dW = dZ^T X)
    // weighted rows (compacted PointNet++ groups, see cg_* below; NULL = every row counts once):
    const float* rw;                              // csum / csq add rw[m] v and rw[m] v^2 for the values of row m of C
    const float* kw;                              // A(m, k) is multiplied by kw[k] when loaded (dW = sum_rows w dZ^T X; asum follows)
};

__global__ __launch_bounds__(256) void gemm_kernel(Gemm g) {
    // k-major tiles: the MFMA operands of k-pair (2 kk, 2 kk + 1) are rows 2 kk + (lane >> 5) at column (lane & 31) of the wave's quadrant
    __shared__ __attribute__((aligned(16))) float As[16][64 + 4], Bs[16][64 + 4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int batch = blockIdx.z / g.ksplit, ks = blockIdx.z - batch * g.ksplit;
    const float* A = g.A + (long long)batch * g.bA;
    const float* B = g.B + (long long)batch * g.bB;
    float* C = g.C + (long long)batch * g.bC;
    const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;      // (row tiles on x: up to millions of rows)
    const int k_lo = ks * g.kchunk, k_hi = min(g.K, k_lo + g.kchunk);
    const bool a_kfast = g.sak == 1, b_nfast = g.sbn == 1;
    // The four A and four B elements this thread moves per k-block are CONSECUTIVE along the operand's contiguous axis (k for a row-major
    // activation / gradient matrix, m or n for its transposed view): one 16-byte load when the four are in range and the address is aligned
    // (column slices of a concat buffer start at odd offsets: those fall back to four scalar loads), four scalars otherwise.
    const int am0 = a_kfast ? t >> 2 : (t & 15) * 4, ak0 = a_kfast ? (t & 3) * 4 : t >> 4;
    const int bn0 = b_nfast ? (t & 15) * 4 : t >> 2, bk0 = b_nfast ? t >> 4 : (t & 3) * 4;
    const long long aes = a_kfast ? g.sak : g.sam, bes = b_nfast ? g.sbn : g.sbk;       // element stride along the thread's four
    const float* pa = A + (long long)(m0 + am0) * g.sam + (long long)(k_lo + ak0) * g.sak;
    const float* pb = B + (long long)(k_lo + bk0) * g.sbk + (long long)(n0 + bn0) * g.sbn;
    const long long da = 16 * g.sak, db = 16 * g.sbk;
    // rows / columns of the tile that exist (the k range is checked per block)
    const int am_n = a_kfast ? (m0 + am0 < g.M ? 4 : 0) : max(0, min(4, g.M - (m0 + am0)));
    const int bn_n = b_nfast ? max(0, min(4, g.N - (n0 + bn0))) : (n0 + bn0 < g.N ? 4 : 0);
    float ra[4], rb[4];
    auto fetch = [&](int k0) {
        const int ka = a_kfast ? max(0, min(4, k_hi - (k0 + ak0))) : (k0 + ak0 < k_hi ? 4 : 0);
        const int na = a_kfast ? (am_n ? ka : 0) : (ka ? am_n : 0);          // valid elements of the thread's four (a prefix)
        if (na == 4 && aes == 1 && ((size_t)pa & 15) == 0) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(pa);
            ra[0] = v[0]; ra[1] = v[1]; ra[2] = v[2]; ra[3] = v[3];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = i < na ? pa[i * aes] : 0.0f;
        }
        if (g.kw) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = a_kfast ? k0 + ak0 + i : k0 + ak0;
                if (i < na) ra[i] *= g.kw[k];
            }
        }
        const int kb = b_nfast ? (k0 + bk0 < k_hi ? 4 : 0) : max(0, min(4, k_hi - (k0 + bk0)));
        const int nb = b_nfast ? (kb ? bn_n : 0) : (bn_n ? kb : 0);
        if (nb == 4 && bes == 1 && ((size_t)pb & 15) == 0) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(pb);
            rb[0] = v[0]; rb[1] = v[1]; rb[2] = v[2]; rb[3] = v[3];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) rb[i] = i < nb ? pb[i * bes] : 0.0f;
        }
        pa += da; pb += db;
    };
    // exact fp32 on the matrix pipe (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain over k): wave w owns the 32 x 32 quadrant
    // (w & 1, w >> 1) of the 64 x 64 tile; the next k-block's elements are in flight while this one is multiplied
    const int wm = (wave & 1) * 32, wn = (wave >> 1) * 32, p = lane & 31, h = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    float arow = 0.0f;
    if (k_lo < k_hi) fetch(k_lo);
    for (int k0 = k_lo; k0 < k_hi; k0 += 16) {
        if (a_kfast) {
#pragma unroll
            for (int i = 0; i < 4; ++i) As[ak0 + i][am0] = ra[i];
        } else {
            *reinterpret_cast<f32x4*>(&As[ak0][am0]) = f32x4{ra[0], ra[1], ra[2], ra[3]};
        }
        if (b_nfast) {
            *reinterpret_cast<f32x4*>(&Bs[bk0][bn0]) = f32x4{rb[0], rb[1], rb[2], rb[3]};
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) Bs[bk0 + i][bn0] = rb[i];
        }
        __syncthreads();
        if (k0 + 16 < k_hi) fetch(k0 + 16);
        if (g.asum && blockIdx.y == 0 && t < 64) {
#pragma unroll
            for (int k = 0; k < 16; ++k) arow += As[k][t];       // (rows / k beyond the operand are zero-filled)
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[2 * kk + h][wm + p], Bs[2 * kk + h][wn + p], acc, 0, 0, 0);
        __syncthreads();
    }
    if (g.asum && blockIdx.y == 0 && t < 64 && m0 + t < g.M) atomicAdd(g.asum + m0 + t, arow);
    const int gn = n0 + wn + p;
    const bool col_ok = gn < g.N;
    const float bv = (col_ok && g.bias && ks == 0) ? g.bias[gn] : 0.0f;
    double cs = 0.0, cq = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int gm = m0 + wm + acc_row(r, lane);
        if (gm >= g.M || !col_ok) continue;
        const float v = acc[r] + bv;
        float* dst = C + (long long)gm * g.scm + (long long)gn * g.scn;
        if (g.mode == 0) *dst = v;
        else if (g.mode == 1) *dst += v;
        else atomicAdd(dst, v);
        const double wv = g.rw ? (double)g.rw[gm] : 1.0;
        cs += wv * (double)v; cq += wv * (double)v * (double)v;
    }
    if (g.csum) {      // (wave-uniform; both half-waves hold 16 rows of the same 32 columns)
        cs += __shfl_xor(cs, 32); cq += __shfl_xor(cq, 32);
        const size_t cp = (size_t)(blockIdx.x % kStatCopies) * g.N;
        if (h == 0 && col_ok) { atomicAdd(g.csum + cp + gn, cs); atomicAdd(g.csq + cp + gn, cq); }
    }
}

enum TAct { T_NONE = 0, T_RELU = 1, T_GELU = 2, T_SIGMOID = 3, T_SILU = 4 };

__device__ __forceinline__ float act_f(int a, float z) {
    switch (a) {
        case T_RELU: return z > 0.0f ? z : 0.0f;
        case T_GELU: return gelu(z);
        case T_SIGMOID: return sigmoidf(z);
        case T_SILU: return siluf(z);
        default: return z;
    }
}
// d act / d z from the pre-activation z and the activation y = act(z)
__device__ __forceinline__ float act_d(int a, float z, float y) {
    switch (a) {
        case T_RELU: return z > 0.0f ? 1.0f : 0.0f;
        case T_GELU: return 0.5f * (1.0f + erff(z * 0.70710678118654752440f)) + z * 0.39894228040143267794f * expf(-0.5f * z * z);
        case T_SIGMOID: return y * (1.0f - y);
        case T_SILU: { const float s = sigmoidf(z); return s * (1.0f + z * (1.0f - s)); }
        default: return 1.0f;
    }
}

// rows x C views with a row stride (a column slice of a wider concat buffer)
__global__ void act_fwd_kernel(const float* __restrict__ Z, long long ldz, float* __restrict__ Y, long long ldy, long long rows, int C, int act) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * C) return;
    const long long r = i / C; const int c = (int)(i - r * C);
    Y[r * ldy + c] = act_f(act, Z[r * ldz + c]);
}
// dZ = dY * act'(Z)
__global__ void act_bwd_kernel(const float* __restrict__ dY, long long ldg, const float* __restrict__ Z, long long ldz, const float* __restrict__ Y,
                               long long ldy, float* __restrict__ dZ, long long rows, int C, int act) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * C) return;
    const long long r = i / C; const int c = (int)(i - r * C);
    dZ[i] = dY[r * ldg + c] * act_d(act, Z[r * ldz + c], Y[r * ldy + c]);
}
// out[c] += sum_r X[r, c]     (bias gradients); grid = (C / 64 rounded up, row chunks), block = 64 x 4
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, long long ld, long long rows, int C, float* __restrict__ out) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    const long long chunk = (rows + gridDim.y - 1) / gridDim.y, lo = blockIdx.y * chunk, hi = min(rows, lo + chunk);
    float s = 0.0f;
    if (c < C) for (long long r = lo + q; r < hi; r += 4) s += X[r * ld + c];
    part[q][threadIdx.x & 63] = s;
    __syncthreads();
    if (q == 0 && c < C) atomicAdd(out + c, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// ---- BatchNorm over the rows of Z (rows, C), fused with ReLU --------------------------------------------------------------------------
// (batch statistics -- per-channel sum and sum of squares in double -- are a by-product of the GEMM that produces Z: Gemm::csum / csq)
// mean / invstd of the normalisation; training mode also updates the running statistics (momentum 0.1, unbiased variance)
__global__ void bn_finish_kernel(const double* sum, const double* sq, long long rows, int C, int training, float* running_mean, float* running_var,
                                 float* mean, float* invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (training) {
        double sm = 0.0, sqq = 0.0;
        for (int k = 0; k < kStatCopies; ++k) { sm += sum[(size_t)k * C + c]; sqq += sq[(size_t)k * C + c]; }
        const double m = sm / (double)rows;
        double var = sqq / (double)rows - m * m;
        if (var < 0.0) var = 0.0;
        mean[c] = (float)m;
        invstd[c] = (float)(1.0 / sqrt(var + (double)kBnEps));
        const double unb = rows > 1 ? var * (double)rows / (double)(rows - 1) : var;
        running_mean[c] = (float)(0.9 * (double)running_mean[c] + 0.1 * m);
        running_var[c] = (float)(0.9 * (double)running_var[c] + 0.1 * unb);
    } else {
        mean[c] = running_mean[c];
        invstd[c] = 1.0f / sqrtf(running_var[c] + kBnEps);
    }
}
// the affine normalisation, with its rounding sequence spelled out: forward and both backward kernels evaluate the SAME value (the backward
// kernels decide Y > 0 from it instead of reading Y)
__device__ __forceinline__ float bn_act(float z, float mu, float is, float ga, float be) { return __fmaf_rn(__fmul_rn(__fsub_rn(z, mu), is), ga, be); }
// Y = ReLU((Z - mean) invstd gamma + beta) [* keep * 2: the Dropout(0.5) of the encoder head, mask (clouds, C, 1024) channel-first]
__global__ void bn_relu_fwd_kernel(const float* __restrict__ Z, const float* __restrict__ mean, const float* __restrict__ invstd,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ Y, long long ldy, long long rows,
                                   int C, const unsigned char* __restrict__ keep) {
    // four channels per thread (C and ldy are multiples of 4: conv_bn_relu checks)
    const long long i4 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int q = C >> 2;
    if (i4 >= rows * q) return;
    const long long r = i4 / q; const int c = (int)(i4 - r * q) * 4;
    const f32x4 z = *reinterpret_cast<const f32x4*>(Z + r * C + c);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), is = *reinterpret_cast<const f32x4*>(invstd + c);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c), be = *reinterpret_cast<const f32x4*>(beta + c);
    f32x4 y;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float v = bn_act(z[j], mu[j], is[j], ga[j], be[j]);
        v = v > 0.0f ? v : 0.0f;
        if (keep) v = keep[((r >> 10) * C + c + j) * 1024 + (r & 1023)] ? v * 2.0f : 0.0f;
        y[j] = v;
    }
    *reinterpret_cast<f32x4*>(Y + r * ldy + c) = y;
}
// first backward pass: dA = dY [* 2 keep] * (Y > 0)  ->  sum_r dA (d beta) and sum_r dA xhat (d gamma), in double.
// Y > 0 is decided from Z with the forward kernel's own expression (bn_act: the same instruction sequence, so the same sign), so the pass
// reads two tensors instead of three; the dropout layer (keep != NULL) also reads its mask.  Four channels per thread (16-byte loads: the
// tensors are dense (rows, C), C a multiple of 4), 256 / (C / 4) rows per pass of the work-group with four row groups in flight, the row
// slots of a channel quad combined through LDS, one atomic per channel and work-group.
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dY, const float* __restrict__ Z,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, long long rows, int C,
                                                            const unsigned char* __restrict__ keep, double* __restrict__ dbeta, double* __restrict__ dgamma,
                                                            const float* __restrict__ roww) {
    __shared__ double part[256][8];
    const int tpr = C >> 2, rpi = 256 / tpr, t = threadIdx.x;
    const int slot = t / tpr, c4 = (t - slot * tpr) * 4;
    const long long chunk = (rows + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * chunk, hi = min(rows, lo + chunk);
    double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
    if (slot < rpi) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c4), is = *reinterpret_cast<const f32x4*>(invstd + c4);
        const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c4), be = *reinterpret_cast<const f32x4*>(beta + c4);
        const float ks = keep ? 2.0f : 1.0f;
        constexpr int U = 4;
        for (long long r0 = lo + slot; r0 < hi; r0 += (long long)U * rpi) {
            f32x4 g[U], z[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long r = r0 + (long long)u * rpi;
                const bool ok = r < hi;
                g[u] = ok ? *reinterpret_cast<const f32x4*>(dY + r * C + c4) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                z[u] = ok ? *reinterpret_cast<const f32x4*>(Z + r * C + c4) : mu;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long r = r0 + (long long)u * rpi;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bool on = bn_act(z[u][j], mu[j], is[j], ga[j], be[j]) > 0.0f;
                    if (keep && r < hi) on = on && keep[((r >> 10) * C + c4 + j) * 1024 + (r & 1023)];
                    float d = on ? g[u][j] * ks : 0.0f;
                    if (roww && r < hi) d *= roww[r];          // (the row stands for roww[r] identical rows)
                    s[j] += d;
                    q[j] += (double)d * (double)((z[u][j] - mu[j]) * is[j]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { part[t][j] = s[j]; part[t][4 + j] = q[j]; }
    __syncthreads();
    if (t < tpr) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            double v = 0.0;
            for (int k = 0; k < rpi; ++k) v += part[k * tpr + t][j];
            atomicAdd((j < 4 ? dbeta : dgamma) + c4 + (j & 3), v);
        }
    }
}
// second pass: dZ.  training: gamma invstd (dA - (d beta + xhat d gamma) / rows);  eval: gamma invstd dA
__global__ void bn_bwd_apply_kernel(const float* __restrict__ dY, long long ldg, const float* __restrict__ Z,
                                    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, const double* __restrict__ dbeta, const double* __restrict__ dgamma, long long rows, int C,
                                    int training, const unsigned char* __restrict__ keep, float* __restrict__ dZ, long long norm_rows,
                                    float* __restrict__ gbeta, float* __restrict__ ggamma) {
    // (the two sums of the reduce pass are also the parameters' gradients: the first work-group adds them, one launch instead of three)
    if (blockIdx.x == 0) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            if (gbeta) gbeta[c] += (float)dbeta[c];
            if (ggamma) ggamma[c] += (float)dgamma[c];
        }
    }
    const long long i4 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int q = C >> 2;
    if (i4 >= rows * q) return;
    const long long r = i4 / q; const int c = (int)(i4 - r * q) * 4;
    const f32x4 g = *reinterpret_cast<const f32x4*>(dY + r * ldg + c), z = *reinterpret_cast<const f32x4*>(Z + r * C + c);
    const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + c), ga = *reinterpret_cast<const f32x4*>(gamma + c);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), be = *reinterpret_cast<const f32x4*>(beta + c);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        bool on = bn_act(z[j], mu[j], is[j], ga[j], be[j]) > 0.0f;
        if (keep) on = on && keep[((r >> 10) * C + c + j) * 1024 + (r & 1023)];
        float d = on ? g[j] : 0.0f;
        if (keep) d *= 2.0f;
        if (training) {
            const float xh = (z[j] - mu[j]) * is[j];
            d -= (float)((dbeta[c + j] + (double)xh * dgamma[c + j]) / (double)norm_rows);      // (norm_rows: the rows of the batch statistics)
        }
        o[j] = d * ga[j] * is[j];
    }
    *reinterpret_cast<f32x4*>(dZ + r * C + c) = o;
}

// ---- GroupNorm(8, 64) of the POSA decoder: per (sample, group) over 8 channels x P points, fused with ReLU ----------------------------
// stats: one work-group per (sample, group)
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ Z, int P, float* __restrict__ mean, float* __restrict__ invstd) {
    __shared__ double ps[256], pq[256];
    const int b = blockIdx.x >> 3, gidx = blockIdx.x & 7;
    double s = 0.0, s2 = 0.0;
    for (int e = threadIdx.x; e < P * 8; e += 256) {
        const double v = Z[((size_t)b * P + (e >> 3)) * 64 + gidx * 8 + (e & 7)];
        s += v; s2 += v * v;
    }
    ps[threadIdx.x] = s; pq[threadIdx.x] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) { ps[threadIdx.x] += ps[threadIdx.x + o]; pq[threadIdx.x] += pq[threadIdx.x + o]; } __syncthreads(); }
    if (threadIdx.x == 0) {
        const double n = (double)P * 8.0, m = ps[0] / n;
        double var = pq[0] / n - m * m;
        if (var < 0.0) var = 0.0;
        mean[blockIdx.x] = (float)m;
        invstd[blockIdx.x] = (float)(1.0 / sqrt(var + (double)kGnEps));
    }
}
__global__ void gn_relu_fwd_kernel(const float* __restrict__ Z, int P, const float* __restrict__ mean, const float* __restrict__ invstd,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ Y, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i & 63);
    const int b = (int)(i / ((long long)P * 64));
    const int sg = b * 8 + (c >> 3);
    const float v = (Z[i] - mean[sg]) * invstd[sg] * gamma[c] + beta[c];
    Y[i] = v > 0.0f ? v : 0.0f;
}
// per (sample, group): s1 = sum dxh, s2 = sum dxh xh with dxh = dA gamma; per channel (atomics): d gamma, d beta
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const float* __restrict__ dY, const float* __restrict__ Y, const float* __restrict__ Z, int P,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                            double* __restrict__ s1, double* __restrict__ s2, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta) {
    __shared__ double ps[256], pq[256];
    __shared__ float cg[8][32], cb[8][32];
    const int b = blockIdx.x >> 3, gidx = blockIdx.x & 7;
    const float mu = mean[blockIdx.x], is = invstd[blockIdx.x];
    double a = 0.0, a2 = 0.0;
    float dg = 0.0f, db = 0.0f;              // thread t always sees channel t & 7 (256 % 8 == 0)
    const int ch = gidx * 8 + (threadIdx.x & 7);
    for (int e = threadIdx.x; e < P * 8; e += 256) {
        const size_t o = ((size_t)b * P + (e >> 3)) * 64 + ch;
        const float d = Y[o] > 0.0f ? dY[o] : 0.0f;
        const float xh = (Z[o] - mu) * is;
        dg += d * xh; db += d;
        const double dx = (double)d * (double)gamma[ch];
        a += dx; a2 += dx * (double)xh;
    }
    ps[threadIdx.x] = a; pq[threadIdx.x] = a2;
    cg[threadIdx.x & 7][threadIdx.x >> 3] = dg; cb[threadIdx.x & 7][threadIdx.x >> 3] = db;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) { ps[threadIdx.x] += ps[threadIdx.x + o]; pq[threadIdx.x] += pq[threadIdx.x + o]; } __syncthreads(); }
    if (threadIdx.x == 0) { s1[blockIdx.x] = ps[0]; s2[blockIdx.x] = pq[0]; }
    if (threadIdx.x < 8) {
        float g = 0.0f, bb = 0.0f;
        for (int i = 0; i < 32; ++i) { g += cg[threadIdx.x][i]; bb += cb[threadIdx.x][i]; }
        atomicAdd(dgamma + gidx * 8 + threadIdx.x, g);
        atomicAdd(dbeta + gidx * 8 + threadIdx.x, bb);
    }
}
__global__ void gn_bwd_apply_kernel(const float* __restrict__ dY, const float* __restrict__ Y, const float* __restrict__ Z, int P,
                                    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                    const double* __restrict__ s1, const double* __restrict__ s2, float* __restrict__ dZ, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i & 63);
    const int b = (int)(i / ((long long)P * 64));
    const int sg = b * 8 + (c >> 3);
    const float d = Y[i] > 0.0f ? dY[i] : 0.0f;
    const float xh = (Z[i] - mean[sg]) * invstd[sg];
    const double n = (double)P * 8.0;
    dZ[i] = invstd[sg] * (float)((double)d * (double)gamma[c] - (s1[sg] + (double)xh * s2[sg]) / n);
}

// ---- PointNet++ plumbing ------------------------------------------------------------------------------------------------------------
// G[(c, s, k), :] = [xyz[c, j] - new_xyz[c, s]  ||  feat[c, j, :]],  j = ball[c, s, k]       (sample_and_group, pointnet2_utils.py:119-135)
__global__ void group_fwd_kernel(const float* __restrict__ xyz, const float* __restrict__ feat, const float* __restrict__ new_xyz, const int* __restrict__ ball,
                                 int NS, int S, int D, float* __restrict__ G, long long rows) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int W = 3 + D;
    if (i >= rows * W) return;
    const long long r = i / W; const int col = (int)(i - r * W);
    const long long cs = r >> 5;                 // (cloud, centroid)
    const long long c = cs / S;
    const int j = ball[r];
    G[i] = col < 3 ? xyz[(c * NS + j) * 3 + col] - new_xyz[cs * 3 + col] : feat[(c * NS + j) * D + (col - 3)];
}
// d feat[c, j, :] += dG[(c, s, k), 3:]
__global__ void group_bwd_kernel(const float* __restrict__ dG, const int* __restrict__ ball, int NS, int S, int D, float* __restrict__ dfeat, long long rows) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * D) return;
    const long long r = i / D; const int col = (int)(i - r * D);
    const long long c = (r >> 5) / S;
    atomicAdd(dfeat + (c * NS + ball[r]) * D + col, dG[r * (3 + D) + 3 + col]);
}
// out[(c, s), ch] = max_k Y[(c, s, k), ch]; arg = the first k attaining it
__global__ void maxpool_fwd_kernel(const float* __restrict__ Y, int C, float* __restrict__ out, unsigned char* __restrict__ arg, long long groups) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= groups * C) return;
    const long long gq = i / C; const int c = (int)(i - gq * C);
    float m = Y[(gq * 32) * C + c]; int a = 0;
    for (int k = 1; k < 32; ++k) { const float v = Y[(gq * 32 + k) * C + c]; if (v > m) { m = v; a = k; } }
    out[i] = m; arg[i] = (unsigned char)a;
}
__global__ void maxpool_bwd_kernel(const float* __restrict__ dout, const unsigned char* __restrict__ arg, int C, float* __restrict__ dY, long long rows) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * C) return;
    const long long r = i / C; const int c = (int)(i - r * C);
    const long long gq = r >> 5;
    dY[i] = arg[gq * C + c] == (unsigned char)(r & 31) ? dout[gq * C + c] : 0.0f;
}
// ---- compacted groups ---------------------------------------------------------------------------------------------------------------
// query_ball_point pads a group with fewer than 32 points in the ball with repeats of its FIRST entry (pointnet2_utils.py:101-104): rows
// d .. 31 of the group are copies of row 0 through the whole MLP stack -- same input, same pre-activations, and, since the max-pool's
// gradient goes to the FIRST maximum, none of them ever receives one.  The stack is therefore evaluated on d (+ 1) rows per group: the d
// distinct ones with weight 1 (row 0 included: it may carry the pool's gradient) and, when d < 32, ONE more copy of row 0 that stands for
// the 32 - d repeats.  Whatever sums over rows -- BatchNorm's batch statistics, its two backward sums, the weight / bias gradients, the
// scatter-add into the source features -- takes the weights; everything per row is unchanged.  At the finest level (r = 0.1: 7 distinct
// neighbours on the bench clouds) the 1.77 M x {32, 32, 64} tensors of a B = 6 step shrink to a quarter.
//   n[g] = d + (d < 32)
__global__ void cg_count_kernel(const int* __restrict__ ball, long long groups, int* __restrict__ n) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= groups) return;
    const int4* b = reinterpret_cast<const int4*>(ball + g * 32);
    const int first = ball[g * 32];
    int d = 1;
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int4 v = b[k]; d += (v.x != first) + (v.y != first) + (v.z != first) + (v.w != first); }
    n[g] = d + (d < 32 ? 1 : 0);
}
//   off[g] = exclusive prefix sum of n, off[groups] = *total = the compacted rows (one work-group walks the array: 10^3 .. 10^5 groups)
__global__ __launch_bounds__(1024) void cg_scan_kernel(const int* __restrict__ n, long long groups, int* __restrict__ off, int* __restrict__ total) {
    __shared__ int wsum[16], carry_s;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) carry_s = 0;
    __syncthreads();
    for (long long base = 0; base < groups; base += 1024) {
        const long long g = base + t;
        const int v = g < groups ? n[g] : 0;
        int x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o); if (lane >= o) x += y; }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int before = carry_s;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        if (g < groups) off[g] = before + x - v;
        __syncthreads();
        if (t == 1023) carry_s = before + x;
        __syncthreads();
    }
    if (t == 0) { off[groups] = carry_s; *total = carry_s; }
}
//   rows of group g: off[g] + k  <-  neighbour k (k < d), then the stand-in of the repeats
__global__ void cg_fill_kernel(const int* __restrict__ ball, const int* __restrict__ off, long long groups, int* __restrict__ rowsrc, int* __restrict__ rowgrp,
                               float* __restrict__ roww) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= groups) return;
    const int o = off[g], nrow = off[g + 1] - o;
    const int first = ball[g * 32];
    const int d = nrow == 32 && ball[g * 32 + 31] != first ? 32 : nrow - 1;      // (32 rows: either 32 distinct, or 31 + the stand-in)
    for (int k = 0; k < nrow; ++k) {
        rowsrc[o + k] = k < d ? ball[g * 32 + k] : first;
        rowgrp[o + k] = (int)g;
        roww[o + k] = k < d ? 1.0f : (float)(32 - d);
    }
}
// G[r, :] = [xyz[c, j] - new_xyz[g]  ||  feat[c, j, :]],  j = rowsrc[r], g = rowgrp[r], c = g / S
__global__ void group_fwd_c_kernel(const float* __restrict__ xyz, const float* __restrict__ feat, const float* __restrict__ new_xyz, const int* __restrict__ rowsrc,
                                   const int* __restrict__ rowgrp, int NS, int S, int D, float* __restrict__ G, long long rows) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int W = 3 + D;
    if (i >= rows * W) return;
    const long long r = i / W; const int col = (int)(i - r * W);
    const long long cs = rowgrp[r], c = cs / S;
    const int j = rowsrc[r];
    G[i] = col < 3 ? xyz[(c * NS + j) * 3 + col] - new_xyz[cs * 3 + col] : feat[(c * NS + j) * D + (col - 3)];
}
// d feat[c, j, :] += roww[r] dG[r, 3:]
__global__ void group_bwd_c_kernel(const float* __restrict__ dG, const int* __restrict__ rowsrc, const int* __restrict__ rowgrp, const float* __restrict__ roww, int NS,
                                   int S, int D, float* __restrict__ dfeat, long long rows) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * D) return;
    const long long r = i / D; const int col = (int)(i - r * D);
    const long long c = rowgrp[r] / S;
    atomicAdd(dfeat + (c * NS + rowsrc[r]) * D + col, roww[r] * dG[r * (3 + D) + 3 + col]);
}
// max over the group's rows (the stand-in equals row 0 and never wins the strict comparison); arg = position within the group = the k of the
// uncompacted group
__global__ void maxpool_fwd_c_kernel(const float* __restrict__ Y, int C, const int* __restrict__ off, float* __restrict__ out, unsigned char* __restrict__ arg,
                                     long long groups) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= groups * C) return;
    const long long gq = i / C; const int c = (int)(i - gq * C);
    const long long o = off[gq]; const int n = off[gq + 1] - (int)o;
    float m = Y[o * C + c]; int a = 0;
    for (int k = 1; k < n; ++k) { const float v = Y[(o + k) * C + c]; if (v > m) { m = v; a = k; } }
    out[i] = m; arg[i] = (unsigned char)a;
}
// The LAST layer of a level's stack feeds the max-pool only: its output gradient is dout[g, c] at the row the pool picked and zero elsewhere.
// Instead of materialising that (rows, C) tensor (maxpool_bwd_c_kernel) and reading it twice, BatchNorm's backward takes it from (dout, arg):
// the two sums run over the groups x C picked entries, the apply pass forms each row's gradient on the fly.
struct PoolGrad { const float* dout; const unsigned char* arg; const int* off; const int* rowgrp; long long groups; };
__global__ __launch_bounds__(256) void bn_bwd_reduce_pool_kernel(PoolGrad pg, const float* __restrict__ Z, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta, int C, double* __restrict__ dbeta,
                                                                 double* __restrict__ dgamma) {
    // thread = (group slot, channel): 256 / C' group slots per pass with C' = min(C, 256) channels side by side, grid.y = channel blocks of 256
    __shared__ double ps[256], pq[256];
    const int cw = C < 256 ? C : 256, slots = 256 / cw;
    const int c = blockIdx.y * 256 + (int)threadIdx.x % cw, slot = (int)threadIdx.x / cw;
    const long long per = (pg.groups + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, hi = min(pg.groups, lo + per);
    double s1 = 0.0, s2 = 0.0;
    const float mu = mean[c], is = invstd[c], ga = gamma[c], be = beta[c];
    for (long long g = lo + slot; g < hi; g += slots) {
        const long long r = (long long)pg.off[g] + pg.arg[g * C + c];
        const float z = Z[r * C + c];
        if (bn_act(z, mu, is, ga, be) > 0.0f) {                  // (the picked row is one of the group's distinct rows: weight 1)
            const float d = pg.dout[g * C + c];
            s1 += d; s2 += (double)d * (double)((z - mu) * is);
        }
    }
    ps[threadIdx.x] = s1; pq[threadIdx.x] = s2;
    __syncthreads();
    if ((int)threadIdx.x < cw) {
        double a = 0.0, b = 0.0;
        for (int k = 0; k < slots; ++k) { a += ps[k * cw + threadIdx.x]; b += pq[k * cw + threadIdx.x]; }
        atomicAdd(dbeta + c, a); atomicAdd(dgamma + c, b);
    }
}
__global__ void bn_bwd_apply_pool_kernel(PoolGrad pg, const float* __restrict__ Z, const float* __restrict__ mean, const float* __restrict__ invstd,
                                         const float* __restrict__ gamma, const float* __restrict__ beta, const double* __restrict__ dbeta,
                                         const double* __restrict__ dgamma, long long rows, int C, int training, float* __restrict__ dZ, long long norm_rows,
                                         float* __restrict__ gbeta, float* __restrict__ ggamma) {
    if (blockIdx.x == 0) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            if (gbeta) gbeta[c] += (float)dbeta[c];
            if (ggamma) ggamma[c] += (float)dgamma[c];
        }
    }
    const long long i4 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int q = C >> 2;
    if (i4 >= rows * q) return;
    const long long r = i4 / q; const int c = (int)(i4 - r * q) * 4;
    const long long g = pg.rowgrp[r];
    const unsigned char k = (unsigned char)(r - pg.off[g]);
    const uchar4 a4 = *reinterpret_cast<const uchar4*>(pg.arg + g * C + c);
    const unsigned char a[4] = {a4.x, a4.y, a4.z, a4.w};
    const f32x4 z = *reinterpret_cast<const f32x4*>(Z + r * C + c);
    const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + c), ga = *reinterpret_cast<const f32x4*>(gamma + c);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), be = *reinterpret_cast<const f32x4*>(beta + c);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool on = a[j] == k && bn_act(z[j], mu[j], is[j], ga[j], be[j]) > 0.0f;
        float d = on ? pg.dout[g * C + c + j] : 0.0f;
        if (training) {
            const float xh = (z[j] - mu[j]) * is[j];
            d -= (float)((dbeta[c + j] + (double)xh * dgamma[c + j]) / (double)norm_rows);
        }
        o[j] = d * ga[j] * is[j];
    }
    *reinterpret_cast<f32x4*>(dZ + r * C + c) = o;
}
// H[(c, n), off + ch] = sum_j w[(c, n), j] p2[c, idx[(c, n), j], ch]      (pointnet2_utils.py:299-303)
__global__ void interp_fwd_kernel(const float* __restrict__ p2, int S, int D2, const int* __restrict__ idx, const float* __restrict__ w, int N, float* __restrict__ H,
                                  int ldh, int off, long long rows) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * D2) return;
    const long long r = i / D2; const int ch = (int)(i - r * D2);
    const long long c = r / N;
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < 3; ++j) s += p2[(c * S + idx[r * 3 + j]) * D2 + ch] * w[r * 3 + j];
    H[r * ldh + off + ch] = s;
}
__global__ void interp_bwd_kernel(const float* __restrict__ dH, int ldh, int off, int S, int D2, const int* __restrict__ idx, const float* __restrict__ w, int N,
                                  float* __restrict__ dp2, long long rows) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * D2) return;
    const long long r = i / D2; const int ch = (int)(i - r * D2);
    const long long c = r / N;
    const float d = dH[r * ldh + off + ch];
#pragma unroll
    for (int j = 0; j < 3; ++j) atomicAdd(dp2 + (c * S + idx[r * 3 + j]) * D2 + ch, d * w[r * 3 + j]);
}
// dst[r, doff + c] (=|+=) src[r, soff + c]
__global__ void copy_cols_kernel(const float* __restrict__ src, long long lds_, int soff, float* __restrict__ dst, long long ldd, int doff, long long rows, int C,
                                 int add) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * C) return;
    const long long r = i / C; const int c = (int)(i - r * C);
    const float v = src[r * lds_ + soff + c];
    if (add) dst[r * ldd + doff + c] += v; else dst[r * ldd + doff + c] = v;
}

// ---- small conditioning pieces ----------------------------------------------------------------------------------------------------
__global__ void gather_rows_kernel(const float* __restrict__ table, const int64_t* __restrict__ t, int C, float* __restrict__ out, int rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows * C) out[i] = table[(size_t)t[i / C] * C + i % C];
}
// x_t = qa[b] x_start + qb[b] noise            (q_sample, gaussian_diffusion.py:233-256; qa / qb gathered from the f64 tables by the caller)
__global__ void qsample_kernel(const float* xs, const float* nz, const float* qa, const float* qb, float* xt, int per, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) { const int b = (int)(i / per); xt[i] = qa[b] * xs[i] + qb[b] * nz[i]; }
}
// Object attention weights (model/sdm.py:180-182): 8 heads of 16 over the 9 slots, float mask ADDED (Q4), mask row of (b, h) is
// mask[(b 8 + h) % B] (Q3), head-averaged.  One work-group of 64 threads per sample; q (B,128) already holds the bias, k (B*9,128) too.
__global__ __launch_bounds__(64) void objattn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ mask, int B,
                                                         float* __restrict__ prob, float* __restrict__ w) {
    __shared__ float lg[8][kO];
    const int b = blockIdx.x, t = threadIdx.x;
    for (int e = t; e < 8 * kO; e += 64) {
        const int h = e / kO, o = e - h * kO;
        float s = 0.0f;
        for (int d = 0; d < 16; ++d) s = fmaf(q[b * 128 + h * 16 + d] * 0.25f, k[((size_t)b * kO + o) * 128 + h * 16 + d], s);
        lg[h][o] = s + mask[(size_t)((b * 8 + h) % B) * kO + o];
    }
    __syncthreads();
    if (t < 8) {
        float m = lg[t][0];
        for (int o = 1; o < kO; ++o) m = fmaxf(m, lg[t][o]);
        float s = 0.0f;
        for (int o = 0; o < kO; ++o) { lg[t][o] = expf(lg[t][o] - m); s += lg[t][o]; }
        for (int o = 0; o < kO; ++o) { lg[t][o] /= s; prob[((size_t)b * 8 + t) * kO + o] = lg[t][o]; }
    }
    __syncthreads();
    if (t < kO) { float s = 0.0f; for (int h = 0; h < 8; ++h) s += lg[h][t]; w[b * kO + t] = s / 8.0f; }
}
__global__ __launch_bounds__(64) void objattn_bwd_kernel(const float* __restrict__ dw, const float* __restrict__ prob, const float* __restrict__ q,
                                                         const float* __restrict__ k, float* __restrict__ dq, float* __restrict__ dk) {
    __shared__ float dl[8][kO];
    const int b = blockIdx.x, t = threadIdx.x;
    if (t < 8) {
        float dot = 0.0f;
        for (int o = 0; o < kO; ++o) dot += prob[((size_t)b * 8 + t) * kO + o] * dw[b * kO + o] / 8.0f;
        for (int o = 0; o < kO; ++o) { const float p = prob[((size_t)b * 8 + t) * kO + o]; dl[t][o] = p * (dw[b * kO + o] / 8.0f - dot); }
    }
    __syncthreads();
    for (int e = t; e < 128; e += 64) {          // dq[b, h*16+d] += 0.25 sum_o dl[h][o] k[b,o,h,d]
        const int h = e >> 4;
        float s = 0.0f;
        for (int o = 0; o < kO; ++o) s = fmaf(dl[h][o], k[((size_t)b * kO + o) * 128 + e], s);
        dq[b * 128 + e] += 0.25f * s;
    }
    for (int e = t; e < kO * 128; e += 64) {     // dk[b, o, h*16+d] += dl[h][o] 0.25 q[b, h*16+d]
        const int o = e >> 7, hd = e & 127;
        dk[((size_t)b * kO + o) * 128 + hd] += dl[hd >> 4][o] * 0.25f * q[b * 128 + hd];
    }
}
// P (B,9,1024,3) = raw reshape of (pcd.permute(0,2,1) * w[:, None, :])  (Q1): flat f = j 9 + o of sample b holds pcd[b, o, j] w[b, o]
__global__ void pweight_fwd_kernel(const float* __restrict__ pcd, const float* __restrict__ w, float* __restrict__ P, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int per = kO * kN * 3;
    const int b = (int)(i / per), f = (int)(i - (long long)b * per);
    const int j = f / kO, o = f - j * kO;
    P[i] = pcd[((size_t)b * kO + o) * (kN * 3) + j] * w[b * kO + o];
}
__global__ void pweight_bwd_kernel(const float* __restrict__ dP, const float* __restrict__ pcd, const float* __restrict__ w, float* __restrict__ dpcd,
                                   float* __restrict__ dw, long long total) {
    // (a block of 256 never straddles two samples: kO kN 3 = 108 x 256 -- its nine partial sums of dw go through LDS, nine global atomics per
    // block instead of one per element onto 54 addresses: 347 -> 12 us at B = 6)
    static_assert((kO * kN * 3) % 256 == 0, "a block of 256 elements belongs to one sample");
    __shared__ float part[kO];
    if (threadIdx.x < kO) part[threadIdx.x] = 0.0f;
    __syncthreads();
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int per = kO * kN * 3;
    const int b = (int)(((long long)blockIdx.x * blockDim.x) / per);
    if (i < total) {
        const int f = (int)(i - (long long)b * per);
        const int j = f / kO, o = f - j * kO;
        const size_t src = ((size_t)b * kO + o) * (kN * 3) + j;
        dpcd[src] += dP[i] * w[b * kO + o];
        atomicAdd(&part[o], dP[i] * pcd[src]);
    }
    __syncthreads();
    if (threadIdx.x < kO) atomicAdd(dw + b * kO + threadIdx.x, part[threadIdx.x]);
}
// pcd_attention collapsed to ONE query row per (b, o) (Q7): 12 heads of dim 1 over the 1024 points of the slot, scale 1.
// q2 (BO,12), k2 / v2 (BO*1024,12) hold their biases.  One work-group per (b, o); a (BO*1024,12) is kept for the backward.
__global__ __launch_bounds__(256) void pcdattn_fwd_kernel(const float* __restrict__ q2, const float* __restrict__ k2, const float* __restrict__ v2,
                                                          float* __restrict__ a, float* __restrict__ opre) {
    __shared__ float red[256];
    __shared__ float hm_[12], hs[12];
    const int bo = blockIdx.x, t = threadIdx.x;
    const float* kb = k2 + (size_t)bo * kN * 12;
    const float* vb = v2 + (size_t)bo * kN * 12;
    float* ab = a + (size_t)bo * kN * 12;
    for (int h = 0; h < 12; ++h) {
        const float qh = q2[bo * 12 + h];
        float m = -3.4e38f;
        for (int n = t; n < kN; n += 256) m = fmaxf(m, qh * kb[n * 12 + h]);
        red[t] = m; __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (t < o) red[t] = fmaxf(red[t], red[t + o]); __syncthreads(); }
        m = red[0]; __syncthreads();
        float s = 0.0f;
        for (int n = t; n < kN; n += 256) { const float e = expf(qh * kb[n * 12 + h] - m); ab[n * 12 + h] = e; s += e; }
        red[t] = s; __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
        s = red[0]; __syncthreads();
        float acc = 0.0f;
        for (int n = t; n < kN; n += 256) { const float p = ab[n * 12 + h] / s; ab[n * 12 + h] = p; acc += p * vb[n * 12 + h]; }
        red[t] = acc; __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
        if (t == 0) opre[bo * 12 + h] = red[0];
        __syncthreads();
    }
    (void)hm_; (void)hs;
}
__global__ __launch_bounds__(256) void pcdattn_bwd_kernel(const float* __restrict__ dopre, const float* __restrict__ q2, const float* __restrict__ k2,
                                                          const float* __restrict__ v2, const float* __restrict__ a, float* __restrict__ dq2,
                                                          float* __restrict__ dk2, float* __restrict__ dv2) {
    __shared__ float red[256];
    const int bo = blockIdx.x, t = threadIdx.x;
    const size_t base = (size_t)bo * kN * 12;
    for (int h = 0; h < 12; ++h) {
        const float d = dopre[bo * 12 + h], qh = q2[bo * 12 + h];
        float dot = 0.0f;                          // sum_n a_n da_n, da_n = d v_n
        for (int n = t; n < kN; n += 256) dot += a[base + n * 12 + h] * d * v2[base + n * 12 + h];
        red[t] = dot; __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
        dot = red[0]; __syncthreads();
        float gq = 0.0f;
        for (int n = t; n < kN; n += 256) {
            const float p = a[base + n * 12 + h];
            dv2[base + n * 12 + h] += p * d;
            const float ds = p * (d * v2[base + n * 12 + h] - dot);
            dk2[base + n * 12 + h] += ds * qh;
            gq += ds * k2[base + n * 12 + h];
        }
        red[t] = gq; __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
        if (t == 0) dq2[bo * 12 + h] += red[0];
        __syncthreads();
    }
}
// X15[(bo, n), :] = [P[bo, n, :] (3) || o[bo, :] (12)]                     (model/sdm.py:196)
__global__ void cat15_fwd_kernel(const float* __restrict__ P, const float* __restrict__ o12, float* __restrict__ X, long long rows) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * 15) return;
    const long long r = i / 15; const int c = (int)(i - r * 15);
    X[i] = c < 3 ? P[r * 3 + c] : o12[(r >> 10) * 12 + (c - 3)];
}
__global__ __launch_bounds__(256) void cat15_bwd_kernel(const float* __restrict__ dX, float* __restrict__ dP, float* __restrict__ do12) {
    __shared__ float red[256];
    const int bo = blockIdx.x, t = threadIdx.x;
    for (int n = t; n < kN; n += 256)
        for (int c = 0; c < 3; ++c) dP[((size_t)bo * kN + n) * 3 + c] += dX[((size_t)bo * kN + n) * 15 + c];
    for (int c = 0; c < 12; ++c) {
        float s = 0.0f;
        for (int n = t; n < kN; n += 256) s += dX[((size_t)bo * kN + n) * 15 + 3 + c];
        red[t] = s; __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (t < o) red[t] += red[t + o]; __syncthreads(); }
        if (t == 0) do12[bo * 12 + c] += red[0];
        __syncthreads();
    }
}
// g = (sum_o pw[b, o, n, c] mask.flatten()[f % (B 9)] + hm[b, n, c]) / 2,  f = flat index in (B,9,1024,3) (Q2);  x = x_t + g
__global__ void guide_fwd_kernel(const float* __restrict__ pw, const float* __restrict__ mask, const float* __restrict__ hm, const float* __restrict__ xt, int B,
                                 float* __restrict__ g, float* __restrict__ x, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int per = kN * 3;
    const int b = (int)(i / per), e = (int)(i - (long long)b * per);
    float s = 0.0f;
    for (int o = 0; o < kO; ++o) {
        const long long f = ((long long)b * kO + o) * per + e;
        s += pw[f] * mask[f % ((long long)B * kO)];
    }
    const float gv = (s + hm[i]) / 2.0f;
    g[i] = gv; x[i] = xt[i] + gv;
}
__global__ void guide_bwd_kernel(const float* __restrict__ dg, const float* __restrict__ mask, int B, float* __restrict__ dpw, float* __restrict__ dhm,
                                 long long total_pw) {
    const long long f = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= total_pw) return;
    const int per = kN * 3;
    const long long bo = f / per; const int e = (int)(f - bo * per);
    const int b = (int)(bo / kO), o = (int)(bo - (long long)b * kO);
    const float d = dg[(long long)b * per + e] * 0.5f;
    dpw[f] += d * mask[f % ((long long)B * kO)];
    if (o == 0) dhm[(long long)b * per + e] += d;
}
// hm[b, n, :] = h[b, n / 2, :]  (nearest x2 up-sampling of the 655 decoder vertices cut to 1024 rows, posa_models.py:324-326)
__global__ void posa_up_fwd_kernel(const float* __restrict__ h, float* __restrict__ hm, int total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int b = i / (kN * 3), e = i - b * kN * 3, n = e / 3, c = e - n * 3;
    hm[i] = h[((size_t)b * kPosa + n / 2) * 3 + c];
}
__global__ void posa_up_bwd_kernel(const float* __restrict__ dhm, float* __restrict__ dh, int total) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // over (b, v < 512, c)
    if (i >= total) return;
    const int b = i / (512 * 3), e = i - b * 512 * 3, v = e / 3, c = e - v * 3;
    dh[((size_t)b * kPosa + v) * 3 + c] += dhm[((size_t)b * kN + 2 * v) * 3 + c] + dhm[((size_t)b * kN + 2 * v + 1) * 3 + c];
}
// rows [0, P2) of each sample's (P1, C) block -> (P2, C) blocks (h[:, :655]); backward adds
__global__ void slice_rows_kernel(const float* __restrict__ src, int P1, float* __restrict__ dst, int P2, int C, long long total, int backward) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long long per = (long long)P2 * C;
    const long long b = i / per, e = i - b * per;
    if (backward) dst[b * P1 * C + e] += src[i]; else dst[i] = src[b * P1 * C + e];
}

// ---- losses -----------------------------------------------------------------------------------------------------------------------
// cat_loss = lambda * mean_b CrossEntropy(p_b, target_b) with p = softmax(GELU(logits)) ALREADY a distribution (gaussian_diffusion.py:1296-1300:
// nn.CrossEntropyLoss applies log_softmax to it again).  Z (B,C) = pre-GELU logits.  One thread per sample; writes dZ * lambda / B.
__global__ void catloss_kernel(const float* __restrict__ Z, const int64_t* __restrict__ target, int B, int C, float lambda, float scale,
                               float* __restrict__ out_cat, float* __restrict__ dZ, float* __restrict__ loss) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float y[64], p[64];
    float m = -3.4e38f;
    for (int c = 0; c < C; ++c) { y[c] = gelu(Z[b * C + c]); m = fmaxf(m, y[c]); }
    float s = 0.0f;
    for (int c = 0; c < C; ++c) { p[c] = expf(y[c] - m); s += p[c]; }
    float m2 = -3.4e38f;
    for (int c = 0; c < C; ++c) { p[c] /= s; if (out_cat) out_cat[b * C + c] = p[c]; m2 = fmaxf(m2, p[c]); }
    float s2 = 0.0f;
    for (int c = 0; c < C; ++c) s2 += expf(p[c] - m2);
    const int tg = (int)target[b];
    const float lse = m2 + logf(s2);
    atomicAdd(loss, lambda * (lse - p[tg]) / (float)B);
    // d loss / d p_c = (softmax(p)_c - [c == tg]) lambda / B;  p = softmax(y): dy_c = p_c (dp_c - sum_j p_j dp_j);  dz = dy gelu'(z)
    float dp[64], dot = 0.0f;
    for (int c = 0; c < C; ++c) { dp[c] = (expf(p[c] - lse) - (c == tg ? 1.0f : 0.0f)) * lambda * scale / (float)B; dot += p[c] * dp[c]; }
    for (int c = 0; c < C; ++c) dZ[b * C + c] = p[c] * (dp[c] - dot) * act_d(T_GELU, Z[b * C + c], y[c]);
}
// chamfer (pytorch3d defaults): loss = mean_b [ mean_i d2x[b,i] + mean_j d2y[b,j] ];  d out[b,i] = 2 (out_i - y_{nx(i)}) / (N B) and, for
// every target j whose nearest prediction is i, += 2 (out_i - y_j) / (N B)
__global__ void chamfer_loss_kernel(const float* __restrict__ d2x, const float* __restrict__ d2y, long long total, float inv, float* __restrict__ loss) {
    __shared__ float red[256];
    float s = 0.0f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) s += d2x[i] + d2y[i];
    red[threadIdx.x] = s; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) atomicAdd(loss, red[0] * inv);
}
__global__ void chamfer_bwd_kernel(const float* __restrict__ out, const float* __restrict__ y, const int* __restrict__ nx, const int* __restrict__ ny, int N,
                                   float coef, float* __restrict__ dout, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;       // over (b, point)
    if (i >= total) return;
    const long long b = i / N;
    for (int c = 0; c < 3; ++c) {
        atomicAdd(dout + i * 3 + c, coef * (out[i * 3 + c] - y[(b * N + nx[i]) * 3 + c]));
        const long long p = b * N + ny[i];
        atomicAdd(dout + p * 3 + c, coef * (out[p * 3 + c] - y[i * 3 + c]));
    }
}
__global__ void add3_kernel(const float* a, const float* b, float* out) { out[0] = a[0] + b[0]; out[1] = a[0]; out[2] = b[0]; }
// zeroes a LIST of buffers in one launch (the ~160 parameter-gradient tensors of the caller: one memset each was 160 launches per step);
// block (x, y): chunk x of buffer y
__global__ void zero_list_kernel(float* const* __restrict__ ptrs, const long long* __restrict__ ns) {
    float* p = ptrs[blockIdx.y];
    const long long n = ns[blockIdx.y];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = 0.0f;
}
__global__ void axpy_kernel(const float* __restrict__ src, float* __restrict__ dst, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------------------
// host side: session, arena, tape
// ------------------------------------------------------------------------------------------------------------------------------
struct lsdm_train {
    int device = 0, C = 13;
    std::string err;
    struct Bound { float* p; float* g; int64_t n; };
    std::map<std::string, Bound> bound;
    // arena: chunks handed out front to back, everything released at the end of a step
    struct Chunk { char* base; size_t cap, used; };
    std::vector<Chunk> chunks;
    // buffers that must start from zero come from chunks of their own: the extent the PREVIOUS step used is cleared with one memset per
    // chunk at the start of a step (the allocation sequence of a step repeats), an allocation beyond it is cleared on its own
    struct ZChunk { char* base; size_t cap, used, zeroed; };
    std::vector<ZChunk> zchunks;
    // device-side list of the bound gradient buffers (zero_list_kernel), rebuilt when the bindings change
    float** zl_ptrs = nullptr; long long* zl_ns = nullptr; int zl_count = 0; long long zl_max = 0; bool zl_dirty = true;
    hipStream_t st = nullptr;
    std::vector<std::function<void()>> tape;
    std::map<std::string, std::pair<void*, int64_t>> named;     // intermediates of the last step kept for the tests (lsdm_train_tensor)
    bool failed = false;
    // compacted rows of the four PointNet++ levels of the step: counted on the device at the top of the step, read here just before the encoder
    int* h_rows = nullptr;          // pinned
    hipEvent_t ev_rows = nullptr;
    // the geometry of a step (FPS: 54 dependent chains of 1360 steps at B = 6, half a millisecond on an otherwise idle chip; ball queries; the
    // row layout) runs on a stream of its own next to the conditioning branches, which do not read it
    hipStream_t geo = nullptr;
    hipEvent_t ev_in = nullptr, ev_geo = nullptr;
    int compact = getenv("LSDM_TRAIN_COMPACT") ? atoi(getenv("LSDM_TRAIN_COMPACT")) : 1;    // A/B switch: 0 = every group on its 32 rows
};

namespace {

int tfail(lsdm_train* s, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (s) { s->err = buf; s->failed = true; }
    return code;
}

void* arena_alloc(lsdm_train* s, size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    for (auto& c : s->chunks)
        if (c.cap - c.used >= bytes) { void* p = c.base + c.used; c.used += bytes; return p; }
    const size_t cap = bytes > ((size_t)512 << 20) ? bytes : ((size_t)512 << 20);
    void* base = nullptr;
    if (hipMalloc(&base, cap) != hipSuccess) { tfail(s, LSDM_ERR_HIP, "training workspace: hipMalloc of %zu bytes failed", cap); return nullptr; }
    s->chunks.push_back({static_cast<char*>(base), cap, bytes});
    return base;
}
void* zarena_alloc(lsdm_train* s, size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    for (auto& c : s->zchunks)
        if (c.cap - c.used >= bytes) {
            char* p = c.base + c.used;
            if (c.used + bytes > c.zeroed) {        // (partly) beyond what the step's opening memset covered
                const size_t from = c.used > c.zeroed ? c.used : c.zeroed;
                (void)hipMemsetAsync(c.base + from, 0, c.used + bytes - from, s->st);
            }
            c.used += bytes;
            return p;
        }
    const size_t cap = bytes > ((size_t)256 << 20) ? bytes : ((size_t)256 << 20);
    void* base = nullptr;
    if (hipMalloc(&base, cap) != hipSuccess) { tfail(s, LSDM_ERR_HIP, "training workspace: hipMalloc of %zu bytes failed", cap); return nullptr; }
    (void)hipMemsetAsync(base, 0, bytes, s->st);
    s->zchunks.push_back({static_cast<char*>(base), cap, bytes, 0});
    return base;
}
template <class T>
T* talloc(lsdm_train* s, size_t n, bool zero = false) {
    return static_cast<T*>(zero ? zarena_alloc(s, n * sizeof(T)) : arena_alloc(s, n * sizeof(T)));
}

inline unsigned nblk(long long n, int b = 256) { return (unsigned)((n + b - 1) / b); }

// (rows, C) view, row stride ld.  gs (optional): state of a LAZY gradient buffer -- 0 = not written yet and NOT zeroed: its first writer must
// overwrite every element (a data-gradient GEMM in store mode, the max-pool's backward, which writes all rows); 1 = holds a value, later
// writers accumulate.  Only tensors whose consumers are all of that kind are created lazy (the hidden activations of the PointNet++
// MLP stacks: 1.77 M x 32 .. 64 at the finest level): no memset of the buffer and no read-modify-write in the first GEMM.
struct Ten { float* d = nullptr; float* g = nullptr; long long rows = 0; int C = 0; long long ld = 0; std::shared_ptr<int> gs; };

Ten make(lsdm_train* s, long long rows, int C, bool grad = true, bool lazy = false) {
    Ten t;
    t.rows = rows; t.C = C; t.ld = C;
    t.d = talloc<float>(s, (size_t)rows * C);
    if (grad) {
        t.g = talloc<float>(s, (size_t)rows * C, !lazy);
        if (lazy) t.gs = std::make_shared<int>(0);
    }
    return t;
}
Ten cols(const Ten& t, int off, int C) { Ten v = t; v.d += off; if (v.g) v.g += off; v.C = C; return v; }

struct Par { const float* w = nullptr; float* g = nullptr; int64_t n = 0; };
Par par(lsdm_train* s, const std::string& name, int64_t numel) {
    auto it = s->bound.find(name);
    if (it == s->bound.end()) { tfail(s, LSDM_ERR_PARAM, "training: parameter %s is not bound (lsdm_train_bind)", name.c_str()); return Par{}; }
    if (it->second.n != numel) { tfail(s, LSDM_ERR_PARAM, "training: %s has %lld elements, expected %lld", name.c_str(), (long long)it->second.n, (long long)numel); return Par{}; }
    return Par{it->second.p, it->second.g, numel};
}

void gemm(lsdm_train* s, const float* A, long long sam, long long sak, const float* B, long long sbk, long long sbn, float* C, long long scm, long long scn,
          int M, int N, int K, const float* bias, int mode, int batch = 1, long long bA = 0, long long bB = 0, long long bC = 0, int ksplit = 1,
          double* csum = nullptr, double* csq = nullptr, float* asum = nullptr, const float* rw = nullptr, const float* kw = nullptr) {
    if (s->failed || M <= 0 || N <= 0) return;
    Gemm g{A, B, C, bias, M, N, K, sam, sak, sbk, sbn, scm, scn, bA, bB, bC, ksplit, 0, mode, csum, csq, asum, rw, kw};
    g.kchunk = ((K + ksplit - 1) / ksplit + 15) / 16 * 16;
    g.ksplit = (K + g.kchunk - 1) / g.kchunk;
    if (g.ksplit < 1) g.ksplit = 1;
    hipLaunchKernelGGL(gemm_kernel, dim3((M + 63) / 64, (N + 63) / 64, batch * g.ksplit), dim3(256), 0, s->st, g);
}

// Y = X W^T + b on (rows, K) -> (rows, N) views
// (csum / csq: the per-channel sum and sum of squares of Z in double, accumulated by the same launch -- batch statistics)
// (roww: weights of the rows in the statistics -- compacted groups)
void linear_fwd(lsdm_train* s, const Ten& X, const Par& W, const Par& b, const Ten& Z, double* csum = nullptr, double* csq = nullptr, const float* roww = nullptr) {
    gemm(s, X.d, X.ld, 1, W.w, 1, X.C, Z.d, Z.ld, 1, (int)X.rows, Z.C, X.C, b.w, 0, 1, 0, 0, 0, 1, csum, csq, nullptr, roww);
}
// dX += dZ W ; dW += dZ^T X ; db += colsum(dZ)      (dZ dense (rows, N))
// (roww: a row stands for roww[r] identical rows -- the weight and bias gradients sum over all of them, the data gradient is per row)
void linear_bwd(lsdm_train* s, const Ten& X, const Par& W, const Par& b, const float* dZ, int N, bool need_dx = true, const float* roww = nullptr) {
    if (s->failed) return;
    const int rows = (int)X.rows, K = X.C;
    if (need_dx && X.g) {
        const bool first = X.gs && *X.gs == 0 && X.ld == K;       // lazy buffer, whole tensor: this GEMM stores it
        // (a column slice of a lazy buffer as the FIRST writer would accumulate into memory that was never zeroed: no call site does that
        // today -- the protocol is checked here instead of held by convention)
        if (X.gs && *X.gs == 0 && X.ld != K) { tfail(s, LSDM_ERR_STATE, "training: first writer of a lazy gradient buffer is a column slice (%d of %lld columns)", K, X.ld); return; }
        gemm(s, dZ, N, 1, W.w, K, 1, X.g, X.ld, 1, rows, K, N, nullptr, first ? 0 : 1);
        if (X.gs) *X.gs = 1;
    }
    if (W.g) {
        // (the bias gradient = row sums of the A operand dZ^T: a by-product of the same launch)
        // The reduction runs over the ROWS (10^4 .. 10^6) into an N x K tile grid of 1 .. 32 work-groups: split it until the launch fills the
        // chip (~3 work-groups per CU), down to 128 rows per work-group.  (One work-group per 4096 rows left the small layers on 2 .. 56
        // work-groups for ~200 us each: 5.5 of the step's 12.4 ms of GEMM time, profiles/r04_a_train_*.)
        const int tiles = ((N + 63) / 64) * ((K + 63) / 64);
        int ksplit = (768 + tiles - 1) / tiles;
        const int most = (rows + 127) / 128;
        if (ksplit > most) ksplit = most;
        if (ksplit < 1) ksplit = 1;
        gemm(s, dZ, 1, N, X.d, X.ld, 1, W.g, K, 1, N, K, rows, nullptr, 2, 1, 0, 0, 0, ksplit, nullptr, nullptr, b.g, nullptr, roww);
    }
    if (b.g && !W.g) {
        if (roww) { tfail(s, LSDM_ERR_STATE, "training: bias-only gradient of a layer on weighted rows"); return; }
        const int ry = rows > 2048 ? (rows + 2047) / 2048 : 1;
        hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64, ry), dim3(256), 0, s->st, dZ, (long long)N, (long long)rows, N, b.g);
    }
}

// dense layer with activation: Y = act(X W^T + b); keeps Z
Ten dense(lsdm_train* s, const Ten& X, const std::string& name, int N, int act, Ten* into = nullptr, bool need_dx = true) {
    const Par W = par(s, name + ".weight", (int64_t)N * X.C), b = par(s, name + ".bias", N);
    Ten Y = into ? *into : make(s, X.rows, N);
    if (s->failed) return Y;
    float* Z = act == T_NONE ? nullptr : talloc<float>(s, (size_t)X.rows * N);
    Ten Zt; Zt.d = act == T_NONE ? Y.d : Z; Zt.rows = X.rows; Zt.C = N; Zt.ld = act == T_NONE ? Y.ld : N;
    linear_fwd(s, X, W, b, Zt);
    if (act != T_NONE) hipLaunchKernelGGL(act_fwd_kernel, dim3(nblk(X.rows * N)), dim3(256), 0, s->st, Z, (long long)N, Y.d, Y.ld, X.rows, N, act);
    s->tape.push_back([=]() {
        float* dZ = talloc<float>(s, (size_t)X.rows * N);
        if (!dZ) return;
        if (act != T_NONE) hipLaunchKernelGGL(act_bwd_kernel, dim3(nblk(X.rows * N)), dim3(256), 0, s->st, Y.g, Y.ld, Z, (long long)N, Y.d, Y.ld, dZ, X.rows, N, act);
        else hipLaunchKernelGGL(copy_cols_kernel, dim3(nblk(X.rows * N)), dim3(256), 0, s->st, Y.g, Y.ld, 0, dZ, (long long)N, 0, X.rows, N, 0);
        linear_bwd(s, X, W, b, dZ, N, need_dx);
    });
    return Y;
}

// 1x1 conv + BatchNorm + ReLU (+ dropout keep-mask) on (rows, K) -> (rows, N)
// roww / norm_rows: the rows are COMPACTED groups (cg_* kernels): row r stands for roww[r] identical rows of the norm_rows the layer has in the
// reference -- batch statistics and every gradient that sums over rows weigh it so, everything per row is unchanged
Ten conv_bn_relu(lsdm_train* s, const Ten& X, const std::string& conv, const std::string& bn, int N, int training, const unsigned char* keep = nullptr,
                 bool need_dx = true, bool lazy_out = false, const float* roww = nullptr, long long norm_rows = 0, const PoolGrad* pool = nullptr) {
    const Par W = par(s, conv + ".weight", (int64_t)N * X.C), b = par(s, conv + ".bias", N);
    const Par gam = par(s, bn + ".weight", N), bet = par(s, bn + ".bias", N);
    const Par rm = par(s, bn + ".running_mean", N), rv = par(s, bn + ".running_var", N);
    Ten Y = make(s, X.rows, N, true, lazy_out);
    if (s->failed) return Y;
    if (N % 4 != 0 || N > 1024 || 256 % (N / 4) != 0) { tfail(s, LSDM_ERR_ARG, "training: BatchNorm width %d (a power of two in 32 .. 1024 expected)", N); return Y; }
    const long long rows = X.rows;
    const long long nrows = roww ? norm_rows : rows;
    float* Z = talloc<float>(s, (size_t)rows * N);
    float* mean = talloc<float>(s, N);
    float* invstd = talloc<float>(s, N);
    Ten Zt; Zt.d = Z; Zt.rows = rows; Zt.C = N; Zt.ld = N;
    // (work-groups of the backward reduction: 512 rows each for the large tensors -- bandwidth-bound, at most ~8 per CU.  Smaller tensors are
    // bound by the work-group's serial passes (256 / (N / 4) x 4 rows each: 46 work-groups of 64 passes left the coarsest level at 126 us per
    // launch) on one side and by the 2 N double atomics every work-group ends with on the other (864 work-groups on 55 k rows: 87 us against 52
    // with 108): 256 rows per work-group from 40 k rows, 64 below)
    const int per_wg = rows >= 200000 ? 512 : (rows >= 40000 ? 256 : 64);
    const int ry4 = (int)std::min<long long>(2048, std::max<long long>(1, (rows + per_wg - 1) / per_wg));
    double* st2 = talloc<double>(s, 2 * (size_t)N * kStatCopies, true);
    if (!st2) return Y;
    double* const stq = st2 + (size_t)N * kStatCopies;
    linear_fwd(s, X, W, b, Zt, training ? st2 : nullptr, training ? stq : nullptr, roww);      // batch statistics: by-product of the GEMM
    hipLaunchKernelGGL(bn_finish_kernel, dim3(nblk(N, 64)), dim3(64), 0, s->st, st2, stq, nrows, N, training, const_cast<float*>(rm.w), const_cast<float*>(rv.w),
                       mean, invstd);
    hipLaunchKernelGGL(bn_relu_fwd_kernel, dim3(nblk(rows * (N / 4))), dim3(256), 0, s->st, Z, mean, invstd, gam.w, bet.w, Y.d, Y.ld, rows, N, keep);
    const bool pooled = pool != nullptr;
    const PoolGrad pg = pool ? *pool : PoolGrad{};
    s->tape.push_back([=]() {
        float* dZ = talloc<float>(s, (size_t)rows * N);
        double* red = talloc<double>(s, 2 * (size_t)N, true);
        if (!dZ || !red) return;
        if (pooled) {
            // (the output feeds the level's max-pool only: its gradient is (dout, arg), never materialised)
            // (~8 passes of 256 / min(N, 256) groups per work-group: the gathers of a pass are independent, the passes a dependent chain)
            const int slots = 256 / (N < 256 ? N : 256);
            const int gx = (int)std::min<long long>(2048, std::max<long long>(1, pg.groups / (8 * slots)));
            hipLaunchKernelGGL(bn_bwd_reduce_pool_kernel, dim3(gx, (N + 255) / 256), dim3(256), 0, s->st, pg, Z, mean, invstd, gam.w, bet.w, N, red, red + N);
            hipLaunchKernelGGL(bn_bwd_apply_pool_kernel, dim3(nblk(rows * (N / 4))), dim3(256), 0, s->st, pg, Z, mean, invstd, gam.w, bet.w, red, red + N, rows, N, training,
                               dZ, nrows, bet.g, gam.g);
            linear_bwd(s, X, W, b, dZ, N, need_dx, roww);
            return;
        }
        hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(ry4), dim3(256), 0, s->st, Y.g, Z, mean, invstd, gam.w, bet.w, rows, N, keep, red, red + N, roww);
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(nblk(rows * (N / 4))), dim3(256), 0, s->st, Y.g, Y.ld, Z, mean, invstd, gam.w, bet.w, red, red + N, rows, N,
                           training, keep, dZ, nrows, bet.g, gam.g);
        linear_bwd(s, X, W, b, dZ, N, need_dx, roww);
    });
    return Y;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------------------------
extern "C" {

int lsdm_train_create(lsdm_train** out, int device, int max_cats) {
    if (!out || max_cats <= 0 || max_cats > 64) return LSDM_ERR_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return LSDM_ERR_HIP;
    lsdm_train* s = new lsdm_train();
    s->device = device;
    s->C = max_cats;
    *out = s;
    return LSDM_OK;
}

int lsdm_train_destroy(lsdm_train* s) {
    if (!s) return LSDM_ERR_ARG;
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(s->device);
    for (auto& c : s->chunks) (void)hipFree(c.base);
    for (auto& c : s->zchunks) (void)hipFree(c.base);
    if (s->zl_ptrs) (void)hipFree(s->zl_ptrs);
    if (s->zl_ns) (void)hipFree(s->zl_ns);
    if (s->h_rows) (void)hipHostFree(s->h_rows);
    if (s->ev_rows) (void)hipEventDestroy(s->ev_rows);
    if (s->ev_in) (void)hipEventDestroy(s->ev_in);
    if (s->ev_geo) (void)hipEventDestroy(s->ev_geo);
    if (s->geo) (void)hipStreamDestroy(s->geo);
    if (prev >= 0) (void)hipSetDevice(prev);
    delete s;
    return LSDM_OK;
}

const char* lsdm_train_last_error(const lsdm_train* s) { return s ? s->err.c_str() : "null training session"; }

int lsdm_train_bind(lsdm_train* s, const char* name, float* param, float* grad, int64_t numel) {
    if (!s || !name || !param || numel <= 0) return tfail(s, LSDM_ERR_ARG, "lsdm_train_bind: bad argument");
    // (re-binding the same tensors -- the Python side binds every parameter on every step -- leaves the device-side gradient list alone: it is
    // rebuilt, with two hipFree / hipMalloc / blocking copies, only when an entry is new or one of its pointers or its size changed)
    auto it = s->bound.find(name);
    if (it == s->bound.end() || it->second.p != param || it->second.g != grad || it->second.n != numel) {
        s->bound[name] = lsdm_train::Bound{param, grad, numel};
        s->zl_dirty = true;
    }
    return LSDM_OK;
}

int lsdm_train_read(lsdm_train* s, const char* name, void* host_dst, int64_t capacity_bytes, int64_t* nbytes) {
    if (!s || !name || !host_dst || !nbytes) return tfail(s, LSDM_ERR_ARG, "lsdm_train_read: bad argument");
    auto it = s->named.find(name);
    if (it == s->named.end()) return tfail(s, LSDM_ERR_ARG, "lsdm_train_read: no intermediate named %s in the last step", name);
    *nbytes = it->second.second;
    if (capacity_bytes < it->second.second) return tfail(s, LSDM_ERR_ARG, "lsdm_train_read: %s needs %lld bytes", name, (long long)it->second.second);
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(s->device);
    hipError_t e = hipStreamSynchronize(s->st);
    if (e == hipSuccess) e = hipMemcpy(host_dst, it->second.first, (size_t)it->second.second, hipMemcpyDeviceToHost);
    if (prev >= 0) (void)hipSetDevice(prev);
    if (e != hipSuccess) return tfail(s, LSDM_ERR_HIP, "lsdm_train_read: %s", hipGetErrorString(e));
    return LSDM_OK;
}

int lsdm_train_step(lsdm_train* s, int B, int flags, const float* x_start, const float* qa, const float* qb, const float* noise, const int64_t* t,
                    const float* mask, const float* given_objs, const float* given_cats, const int64_t* target, const float* text,
                    const int64_t* fps_starts, const unsigned char* dropout_keep, float lambda_cat, float grad_scale, float* losses,
                    float* model_output, float* out_cat, void* stream) {
    if (!s || B <= 0 || !x_start || !qa || !qb || !noise || !t || !mask || !given_objs || !given_cats || !target || !text || !fps_starts || !losses)
        return tfail(s, LSDM_ERR_ARG, "lsdm_train_step: bad argument");
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (prev != s->device && hipSetDevice(s->device) != hipSuccess) return tfail(s, LSDM_ERR_HIP, "hipSetDevice(%d) failed", s->device);
    s->failed = false;
    s->err.clear();
    s->st = static_cast<hipStream_t>(stream);
    s->tape.clear();
    s->named.clear();
    for (auto& c : s->chunks) c.used = 0;
    hipStream_t st = s->st;
    for (auto& c : s->zchunks) {
        if (c.used) (void)hipMemsetAsync(c.base, 0, c.used, st);
        c.zeroed = c.used;
        c.used = 0;
    }
    const int training = flags & 1, C = s->C;
    const int nc = B * kO;
    // gradient buffers start from zero (the caller owns them; this call defines their content)
    if (s->zl_dirty) {
        std::vector<float*> hp;
        std::vector<long long> hn;
        s->zl_max = 0;
        for (auto& kv : s->bound)
            if (kv.second.g) { hp.push_back(kv.second.g); hn.push_back((long long)kv.second.n); s->zl_max = std::max<long long>(s->zl_max, kv.second.n); }
        if (s->zl_ptrs) (void)hipFree(s->zl_ptrs);
        if (s->zl_ns) (void)hipFree(s->zl_ns);
        s->zl_ptrs = nullptr; s->zl_ns = nullptr; s->zl_count = (int)hp.size();
        if (s->zl_count) {
            if (hipMalloc(&s->zl_ptrs, hp.size() * sizeof(float*)) != hipSuccess || hipMalloc(&s->zl_ns, hn.size() * sizeof(long long)) != hipSuccess ||
                hipMemcpy(s->zl_ptrs, hp.data(), hp.size() * sizeof(float*), hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(s->zl_ns, hn.data(), hn.size() * sizeof(long long), hipMemcpyHostToDevice) != hipSuccess)
                return tfail(s, LSDM_ERR_HIP, "training: gradient list");
        }
        s->zl_dirty = false;
    }
    if (s->zl_count && !(flags & 2)) {
        const unsigned gx = (unsigned)std::min<long long>(64, (s->zl_max + 1023) / 1024);
        hipLaunchKernelGGL(zero_list_kernel, dim3(gx ? gx : 1, s->zl_count), dim3(256), 0, st, s->zl_ptrs, s->zl_ns);
    }
    (void)hipMemsetAsync(losses, 0, 3 * sizeof(float), st);

    auto body = [&]() {
    // ---- geometry of the scene encoder first (it depends on the inputs only): FPS, ball queries and -- with compacted groups -- the row
    // layout of the four levels, whose row counts travel to the host while the conditioning branches below are being enqueued
    static const int S[4] = {1024, 256, 64, 16}, NS[4] = {1024, 1024, 256, 64};
    static const double RAD[4] = {0.1, 0.2, 0.4, 0.8};
    int* idx[4]; float* xyz[4]; int* ball[4];
    int* cg_off[4] = {}; int* cg_src[4] = {}; int* cg_grp[4] = {}; float* cg_w[4] = {};
    const bool compact = s->compact != 0;
    {
        for (int l = 0; l < 4; ++l) {
            idx[l] = talloc<int>(s, (size_t)nc * S[l]);
            xyz[l] = talloc<float>(s, (size_t)nc * S[l] * 3);
            ball[l] = talloc<int>(s, (size_t)nc * S[l] * 32);
        }
        if (s->failed) return;
        if (!s->geo && (hipStreamCreateWithFlags(&s->geo, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&s->ev_in, hipEventDisableTiming) != hipSuccess ||
                        hipEventCreateWithFlags(&s->ev_geo, hipEventDisableTiming) != hipSuccess)) { tfail(s, LSDM_ERR_HIP, "training: geometry stream"); return; }
        // (the inputs and the arena's previous contents belong to the caller's stream: the side stream starts behind it)
        (void)hipEventRecord(s->ev_in, st);
        (void)hipStreamWaitEvent(s->geo, s->ev_in, 0);
        hipStream_t const sg = s->geo;
        launch_fps_chain(given_objs, fps_starts, nc, nc, idx, xyz, sg);
        const float* lx[5] = {given_objs, xyz[0], xyz[1], xyz[2], xyz[3]};
        for (int l = 0; l < 4; ++l) launch_ball_query(lx[l], lx[l + 1], nc, NS[l], S[l], (float)(RAD[l] * RAD[l]), ball[l], 0, sg);
        if (compact) {
            if (!s->h_rows && hipHostMalloc(reinterpret_cast<void**>(&s->h_rows), 4 * sizeof(int)) != hipSuccess) { tfail(s, LSDM_ERR_HIP, "training: pinned row counts"); return; }
            if (!s->ev_rows && hipEventCreateWithFlags(&s->ev_rows, hipEventDisableTiming) != hipSuccess) { tfail(s, LSDM_ERR_HIP, "training: event"); return; }
            int* totals = talloc<int>(s, 4);
            for (int l = 0; l < 4; ++l) {
                const long long groups = (long long)nc * S[l];
                int* n = talloc<int>(s, (size_t)groups);
                cg_off[l] = talloc<int>(s, (size_t)groups + 1);
                cg_src[l] = talloc<int>(s, (size_t)groups * 32);          // (upper bound: the counts are not on the host yet)
                cg_grp[l] = talloc<int>(s, (size_t)groups * 32);
                cg_w[l] = talloc<float>(s, (size_t)groups * 32);
                if (s->failed) return;
                hipLaunchKernelGGL(cg_count_kernel, dim3(nblk(groups)), dim3(256), 0, sg, ball[l], groups, n);
                hipLaunchKernelGGL(cg_scan_kernel, dim3(1), dim3(1024), 0, sg, n, groups, cg_off[l], totals + l);
                hipLaunchKernelGGL(cg_fill_kernel, dim3(nblk(groups)), dim3(256), 0, sg, ball[l], cg_off[l], groups, cg_src[l], cg_grp[l], cg_w[l]);
            }
            (void)hipMemcpyAsync(s->h_rows, totals, 4 * sizeof(int), hipMemcpyDeviceToHost, sg);
            (void)hipEventRecord(s->ev_rows, sg);
        }
        (void)hipEventRecord(s->ev_geo, sg);
    }
    // ---- forward --------------------------------------------------------------------------------------------------------------------
    // x_t = q_sample(x_start, t, noise)
    Ten xt = make(s, (long long)B * kN, 3, false);
    if (!s->failed) hipLaunchKernelGGL(qsample_kernel, dim3(nblk((long long)B * kN * 3)), dim3(256), 0, st, x_start, noise, qa, qb, xt.d, kN * 3, (long long)B * kN * 3);
    // 1. timestep embedding: pe[t] -> Linear -> SiLU -> Linear                                       (diffusion_utils.py:14-21)
    Ten e0 = make(s, B, 256);                       // [ets || enc]
    Ten pe_t = make(s, B, 128, false);
    {
        const Par pe = par(s, "sequence_pos_encoder.pe", 5000 * 128);
        if (!s->failed) hipLaunchKernelGGL(gather_rows_kernel, dim3(nblk(B * 128)), dim3(256), 0, st, pe.w, t, 128, pe_t.d, B);
    }
    Ten th = dense(s, pe_t, "embed_timestep.time_embed.0", 128, T_SILU, nullptr, false);
    Ten ets = cols(e0, 0, 128);
    dense(s, th, "embed_timestep.time_embed.2", 128, T_NONE, &ets);
    // 2. text embedding D -> D/2 -> 256 -> 128, GELU after every layer (Q9); D = clip_dim from the bound embed_text.0.weight (D/2, D)
    int D = 512;
    {
        auto it = s->bound.find("embed_text.0.weight");
        if (it != s->bound.end()) D = (int)std::llround(std::sqrt(2.0 * (double)it->second.n));
    }
    Ten tx; tx.d = const_cast<float*>(text); tx.rows = B; tx.C = D; tx.ld = D;
    Ten t1 = dense(s, tx, "embed_text.0", D / 2, T_GELU, nullptr, false);
    Ten t2 = dense(s, t1, "embed_text.2", 256, T_GELU);
    Ten enc = cols(e0, 128, 128);
    dense(s, t2, "embed_text.4", 128, T_GELU, &enc);
    // 3. category head on enc.detach(): 128 -> 64 -> 32 -> C, GELU on the logits, softmax (Q10); loss below
    Ten encd = make(s, B, 128, true);               // detached copy: its gradient is never propagated into enc (sdm.py:157)
    if (!s->failed) hipLaunchKernelGGL(copy_cols_kernel, dim3(nblk(B * 128)), dim3(256), 0, st, enc.d, enc.ld, 0, encd.d, 128LL, 0, (long long)B, 128, 0);
    Ten c1 = dense(s, encd, "predict_cat.0", 64, T_GELU, nullptr, false);
    Ten c2 = dense(s, c1, "predict_cat.2", 32, T_GELU);
    Ten zc = dense(s, c2, "predict_cat.4", C, T_NONE);
    // 4. category embedding C -> 32, GELU
    Ten gc; gc.d = const_cast<float*>(given_cats); gc.rows = nc; gc.C = C; gc.ld = C;
    Ten catx = make(s, nc, 160);                    // [ecat (32) || enc repeated (128)]: input of the translation layer
    Ten ecat = cols(catx, 0, 32);
    dense(s, gc, "embed_cat.0", 32, T_GELU, &ecat, false);
    // 5. up-sampled conditioning: rows (b, c) of [ets || enc] as scalars -> 128 -> 512 -> 1024 (GELU), read transposed below
    Ten e0s; e0s.d = e0.d; e0s.g = e0.g; e0s.rows = (long long)B * 256; e0s.C = 1; e0s.ld = 1;
    Ten u1 = dense(s, e0s, "upsampling_layer.0", 128, T_GELU);
    Ten u2 = dense(s, u1, "upsampling_layer.2", 512, T_GELU);
    Ten u3 = dense(s, u2, "upsampling_layer.4", 1024, T_GELU);       // (B*256, 1024): e[b, n, c] = u3[(b, c), n]
    // 16a. emb = GELU(combine_extraction.0(e)): (B*1024, 256) x (128, 256)^T with the transposed view of u3, per sample
    Ten embx = make(s, (long long)B * kN, 256);     // [pose_embedding.2 output (128) || emb (128)]: input of combination_extraction.0
    Ten emb = cols(embx, 128, 128);
    {
        const Par W = par(s, "combine_extraction.0.weight", 128 * 256), bb = par(s, "combine_extraction.0.bias", 128);
        float* Z = talloc<float>(s, (size_t)B * kN * 128);
        if (!s->failed) {
            gemm(s, u3.d, 1, 1024, W.w, 1, 256, Z, 128, 1, kN, 128, 256, bb.w, 0, B, 256LL * 1024, 0, (long long)kN * 128);
            hipLaunchKernelGGL(act_fwd_kernel, dim3(nblk((long long)B * kN * 128)), dim3(256), 0, st, Z, 128LL, emb.d, emb.ld, (long long)B * kN, 128, (int)T_GELU);
        }
        s->tape.push_back([=]() {
            float* dZ = talloc<float>(s, (size_t)B * kN * 128);
            if (!dZ) return;
            hipLaunchKernelGGL(act_bwd_kernel, dim3(nblk((long long)B * kN * 128)), dim3(256), 0, st, emb.g, emb.ld, Z, 128LL, emb.d, emb.ld, dZ, (long long)B * kN, 128,
                               (int)T_GELU);
            // d u3[(b, c), n] += sum_j dZ[(b, n), j] W[j, c]   ->  C(m = n, n = c) with strides (1, 1024)
            gemm(s, dZ, 128, 1, W.w, 256, 1, u3.g, 1, 1024, kN, 256, 128, nullptr, 1, B, (long long)kN * 128, 0, 256LL * 1024);
            // dW[j, c] += sum_{b, n} dZ[(b, n), j] u3[(b, c), n]
            // (one batched launch, the samples' partial sums added atomically: B launches of 8 work-groups took 49 us each)
            gemm(s, dZ, 1, 128, u3.d, 1, 1024, W.g, 256, 1, 128, 256, kN, nullptr, 2, B, (long long)kN * 128, 256LL * 1024, 0);
            hipLaunchKernelGGL(colsum_kernel, dim3(2, 8), dim3(256), 0, st, dZ, 128LL, (long long)B * kN, 128, bb.g);
        });
    }
    // 6. human branch: POSA decoder on given_objs[:, 0]
    Ten hm = make(s, (long long)B * kN, 3);
    {
        Ten hin = make(s, (long long)B * kN, 3, false);
        if (!s->failed) hipLaunchKernelGGL(slice_rows_kernel, dim3(nblk((long long)B * kN * 3)), dim3(256), 0, st, given_objs, kO * kN, hin.d, kN, 3, (long long)B * kN * 3, 0);
        Ten h = hin;
        for (int i = 0; i < 3; ++i) {
            const int P = i < 2 ? kN : kPosa;
            if (i == 2) {
                Ten cut = make(s, (long long)B * kPosa, 64);
                const Ten full = h;
                if (!s->failed) hipLaunchKernelGGL(slice_rows_kernel, dim3(nblk((long long)B * kPosa * 64)), dim3(256), 0, st, full.d, kN, cut.d, kPosa, 64, (long long)B * kPosa * 64, 0);
                s->tape.push_back([=]() {
                    hipLaunchKernelGGL(slice_rows_kernel, dim3(nblk((long long)B * kPosa * 64)), dim3(256), 0, st, cut.g, kN, full.g, kPosa, 64, (long long)B * kPosa * 64, 1);
                });
                h = cut;
            }
            const std::string pre = "human_backbone.de_spiral." + std::to_string(i);
            Ten z = dense(s, h, pre + ".conv.layer", 64, T_NONE, nullptr, i > 0);
            const Par gam = par(s, pre + ".norm.weight", 64), bet = par(s, pre + ".norm.bias", 64);
            Ten y = make(s, (long long)B * P, 64);
            float* mean = talloc<float>(s, (size_t)B * 8);
            float* invstd = talloc<float>(s, (size_t)B * 8);
            const long long tot = (long long)B * P * 64;
            if (!s->failed) {
                hipLaunchKernelGGL(gn_stats_kernel, dim3(B * 8), dim3(256), 0, st, z.d, P, mean, invstd);
                hipLaunchKernelGGL(gn_relu_fwd_kernel, dim3(nblk(tot)), dim3(256), 0, st, z.d, P, mean, invstd, gam.w, bet.w, y.d, tot);
            }
            s->tape.push_back([=]() {
                double* s12 = talloc<double>(s, (size_t)B * 16);
                if (!s12) return;
                hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3(B * 8), dim3(256), 0, st, y.g, y.d, z.d, P, mean, invstd, gam.w, s12, s12 + B * 8, gam.g, bet.g);
                hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(nblk(tot)), dim3(256), 0, st, y.g, y.d, z.d, P, mean, invstd, gam.w, s12, s12 + B * 8, z.g, tot);
            });
            h = y;
        }
        Ten h4 = dense(s, h, "human_backbone.de_spiral.3.layer", 3, T_NONE);
        if (!s->failed) hipLaunchKernelGGL(posa_up_fwd_kernel, dim3(nblk(B * kN * 3)), dim3(256), 0, st, h4.d, hm.d, B * kN * 3);
        s->tape.push_back([=]() { hipLaunchKernelGGL(posa_up_bwd_kernel, dim3(nblk(B * 512 * 3)), dim3(256), 0, st, hm.g, h4.g, B * 512 * 3); });
    }
    // 7. PointNet++ scene encoder on the B*9 clouds
    Ten pcd;                                         // (nc*1024, 3)
    {
        static const int CH[4][3] = {{32, 32, 64}, {64, 64, 128}, {128, 128, 256}, {256, 256, 512}};
        (void)hipStreamWaitEvent(st, s->ev_geo, 0);      // the geometry (side stream) from here on
        long long crows[4] = {0, 0, 0, 0};
        if (compact) {
            // (a HOST synchronisation per step -- the compacted row counts size the encoder's launches: lsdm_train_step with compacted groups
            // cannot be captured into a hipGraph; LSDM_TRAIN_COMPACT=0 has no host round trip)
            if (hipEventSynchronize(s->ev_rows) != hipSuccess) { tfail(s, LSDM_ERR_HIP, "training: row counts"); return; }
            for (int l = 0; l < 4; ++l) {
                crows[l] = s->h_rows[l];
                if (crows[l] < (long long)nc * S[l] || crows[l] > (long long)nc * S[l] * 32) { tfail(s, LSDM_ERR_STATE, "training: %lld compacted rows at level %d", crows[l], l + 1); return; }
            }
        }
        const float* lxyz[5] = {given_objs, xyz[0], xyz[1], xyz[2], xyz[3]};
        Ten feat[5];
        feat[0].d = const_cast<float*>(given_objs); feat[0].rows = (long long)nc * kN; feat[0].C = 3; feat[0].ld = 3;      // l0_points = xyz (pointnet2.py:63-64)
        for (int l = 0; l < 4; ++l) {
            const std::string pre = "pcd_backbone.sa" + std::to_string(l + 1);
            const int D = feat[l].C;
            const long long full_rows = (long long)nc * S[l] * 32;
            const long long rows = compact ? crows[l] : full_rows;
            const float* roww = compact ? cg_w[l] : nullptr;
            const int* rsrc = cg_src[l]; const int* rgrp = cg_grp[l]; const int* goff = cg_off[l];
            Ten G = make(s, rows, 3 + D, l > 0, /*lazy: written by the first conv's data gradient*/ true);
            const Ten src = feat[l];
            const int* bl = ball[l];
            const int NSl = NS[l], Sl = S[l];
            if (!s->failed) {
                if (compact) hipLaunchKernelGGL(group_fwd_c_kernel, dim3(nblk(rows * (3 + D))), dim3(256), 0, st, lxyz[l], src.d, lxyz[l + 1], rsrc, rgrp, NSl, Sl, D, G.d, rows);
                else hipLaunchKernelGGL(group_fwd_kernel, dim3(nblk(rows * (3 + D))), dim3(256), 0, st, lxyz[l], src.d, lxyz[l + 1], bl, NSl, Sl, D, G.d, rows);
            }
            if (l > 0) {
                s->tape.push_back([=]() {
                    if (compact) hipLaunchKernelGGL(group_bwd_c_kernel, dim3(nblk(rows * D)), dim3(256), 0, st, G.g, rsrc, rgrp, roww, NSl, Sl, D, src.g, rows);
                    else hipLaunchKernelGGL(group_bwd_kernel, dim3(nblk(rows * D)), dim3(256), 0, st, G.g, bl, NSl, Sl, D, src.g, rows);
                });
            }
            Ten out = make(s, (long long)nc * S[l], CH[l][2]);
            unsigned char* arg = talloc<unsigned char>(s, (size_t)nc * S[l] * CH[l][2]);
            const int Cc = CH[l][2];
            const long long groups = (long long)nc * S[l];
            const PoolGrad pg{out.g, arg, goff, rgrp, groups};
            Ten h = G;
            for (int i = 0; i < 3; ++i)
                h = conv_bn_relu(s, h, pre + ".mlp_convs." + std::to_string(i), pre + ".mlp_bns." + std::to_string(i), CH[l][i], training, nullptr, !(l == 0 && i == 0),
                                 /*lazy: consumed by the next conv / the max-pool, whose backward passes write every element*/ true, roww, full_rows,
                                 compact && i == 2 ? &pg : nullptr);
            if (!s->failed) {
                if (compact) hipLaunchKernelGGL(maxpool_fwd_c_kernel, dim3(nblk(groups * Cc)), dim3(256), 0, st, h.d, Cc, goff, out.d, arg, groups);
                else hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(nblk(groups * Cc)), dim3(256), 0, st, h.d, Cc, out.d, arg, groups);
            }
            s->named["sa" + std::to_string(l + 1) + ".arg"] = {arg, (int64_t)(groups * Cc)};
            const Ten hl = h;
            if (!compact)
                s->tape.push_back([=]() {
                    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(nblk(rows * Cc)), dim3(256), 0, st, out.g, arg, Cc, hl.g, rows);      // (stores every element)
                    if (hl.gs) *hl.gs = 1;
                });
            // (compacted groups: the last layer's BatchNorm backward reads the pool's gradient as (out.g, arg) -- conv_bn_relu above)
            feat[l + 1] = out;
        }
        // feature propagation: fp4 (l3 <- l4), fp3 (l2 <- l3), fp2 (l1 <- l2), fp1 (l0 <- l1, points1 = None)
        static const int FN[4] = {64, 256, 1024, 1024}, FS[4] = {16, 64, 256, 1024}, FC[4][3] = {{256, 256, 0}, {256, 256, 0}, {256, 128, 0}, {128, 128, 128}};
        const float* fine_xyz[4] = {xyz[2], xyz[1], xyz[0], given_objs};
        const float* coarse_xyz[4] = {xyz[3], xyz[2], xyz[1], xyz[0]};
        Ten p2 = feat[4];
        for (int f = 0; f < 4; ++f) {
            const int N = FN[f], Sc = FS[f];
            int* nidx = talloc<int>(s, (size_t)nc * N * 3);
            float* nw = talloc<float>(s, (size_t)nc * N * 3);
            if (s->failed) return;
            launch_three_nn(fine_xyz[f], coarse_xyz[f], nc, N, Sc, nidx, nw, 0, st);
            const Ten p1 = f < 3 ? feat[3 - f] : Ten{};
            const int D1 = f < 3 ? p1.C : 0, D2 = p2.C;
            const long long rows = (long long)nc * N;
            Ten H = make(s, rows, D1 + D2, true, /*lazy: written by the first conv's data gradient*/ true);
            const Ten p2c = p2;
            if (!s->failed) {
                if (D1) hipLaunchKernelGGL(copy_cols_kernel, dim3(nblk(rows * D1)), dim3(256), 0, st, p1.d, p1.ld, 0, H.d, H.ld, 0, rows, D1, 0);
                hipLaunchKernelGGL(interp_fwd_kernel, dim3(nblk(rows * D2)), dim3(256), 0, st, p2c.d, Sc, D2, nidx, nw, N, H.d, (int)H.ld, D1, rows);
            }
            s->tape.push_back([=]() {
                if (D1) hipLaunchKernelGGL(copy_cols_kernel, dim3(nblk(rows * D1)), dim3(256), 0, st, H.g, H.ld, 0, p1.g, p1.ld, 0, rows, D1, 1);
                hipLaunchKernelGGL(interp_bwd_kernel, dim3(nblk(rows * D2)), dim3(256), 0, st, H.g, (int)H.ld, D1, Sc, D2, nidx, nw, N, p2c.g, rows);
            });
            const std::string pre = "pcd_backbone.fp" + std::to_string(4 - f);
            Ten h = H;
            // (the last layer of fp4 / fp3 / fp2 feeds the next stage's interpolation, whose backward ACCUMULATES: not lazy)
            const int nl = f == 3 ? 3 : 2;
            for (int i = 0; i < nl; ++i)
                h = conv_bn_relu(s, h, pre + ".mlp_convs." + std::to_string(i), pre + ".mlp_bns." + std::to_string(i), FC[f][i], training, nullptr, true,
                                 i + 1 < nl || f == 3);
            p2 = h;
        }
        Ten hh = conv_bn_relu(s, p2, "pcd_backbone.conv1", "pcd_backbone.bn1", 128, training, dropout_keep, true, true);
        pcd = dense(s, hh, "pcd_backbone.conv2", 3, T_NONE);
    }
    {
        // 8. object attention weights
        Ten w = make(s, B, kO);
        {
            const Par wq = par(s, "attn_layer.q_proj_weight", 128 * 128), wk = par(s, "attn_layer.k_proj_weight", 128 * 32);
            const Par ib = par(s, "attn_layer.in_proj_bias", 384);
            Ten q = make(s, B, 128), k = make(s, nc, 128);
            float* prob = talloc<float>(s, (size_t)B * 8 * kO);
            if (s->failed) return;
            Par bq{ib.w, ib.g, 128}, bk{ib.w + 128, ib.g ? ib.g + 128 : nullptr, 128};
            linear_fwd(s, enc, wq, bq, q);
            linear_fwd(s, ecat, wk, bk, k);
            hipLaunchKernelGGL(objattn_fwd_kernel, dim3(B), dim3(64), 0, st, q.d, k.d, mask, B, prob, w.d);
            s->tape.push_back([=]() {
                hipLaunchKernelGGL(objattn_bwd_kernel, dim3(B), dim3(64), 0, st, w.g, prob, q.d, k.d, q.g, k.g);
                linear_bwd(s, enc, wq, bq, q.g, 128);
                linear_bwd(s, ecat, wk, bk, k.g, 128);
            });
        }
        // 9. P = raw reshape of the weighted encoder output (Q1)
        Ten P = make(s, (long long)nc * kN, 3);
        const long long ptot = (long long)nc * kN * 3;
        if (!s->failed) hipLaunchKernelGGL(pweight_fwd_kernel, dim3(nblk(ptot)), dim3(256), 0, st, pcd.d, w.d, P.d, ptot);
        s->tape.push_back([=]() { hipLaunchKernelGGL(pweight_bwd_kernel, dim3(nblk(ptot)), dim3(256), 0, st, P.g, pcd.d, w.d, pcd.g, w.g, ptot); });
        // 10. translation tokens: [ecat || enc] 160 -> 128 -> 12, GELU
        {
            const Ten encr = cols(catx, 32, 128);
            for (int o = 0; o < kO && !s->failed; ++o)      // enc.repeat(1, num_obj, 1): row (b, o) <- enc[b]
                hipLaunchKernelGGL(copy_cols_kernel, dim3(nblk(B * 128)), dim3(256), 0, st, enc.d, enc.ld, 0, encr.d + (size_t)o * catx.ld, catx.ld * kO, 0, (long long)B, 128, 0);
            s->tape.push_back([=]() {
                for (int o = 0; o < kO; ++o)
                    hipLaunchKernelGGL(copy_cols_kernel, dim3(nblk(B * 128)), dim3(256), 0, st, encr.g + (size_t)o * catx.ld, catx.ld * kO, 0, enc.g, enc.ld, 0, (long long)B, 128, 1);
            });
        }
        Ten tr1 = dense(s, catx, "translation_layer.0", 128, T_GELU);
        Ten tr = dense(s, tr1, "translation_layer.2", 12, T_GELU);
        // 11. pcd_attention, one query row per (b, o)
        Ten o12 = make(s, nc, 12);
        {
            const Par wq = par(s, "pcd_attention.q_proj_weight", 144), wk = par(s, "pcd_attention.k_proj_weight", 36), wv = par(s, "pcd_attention.v_proj_weight", 36);
            const Par ib = par(s, "pcd_attention.in_proj_bias", 36);
            Ten q2 = make(s, nc, 12), k2 = make(s, (long long)nc * kN, 12), v2 = make(s, (long long)nc * kN, 12), opre = make(s, nc, 12);
            float* a = talloc<float>(s, (size_t)nc * kN * 12);
            if (s->failed) return;
            Par bq{ib.w, ib.g, 12}, bk{ib.w + 12, ib.g ? ib.g + 12 : nullptr, 12}, bv{ib.w + 24, ib.g ? ib.g + 24 : nullptr, 12};
            linear_fwd(s, tr, wq, bq, q2);
            linear_fwd(s, P, wk, bk, k2);
            linear_fwd(s, P, wv, bv, v2);
            hipLaunchKernelGGL(pcdattn_fwd_kernel, dim3(nc), dim3(256), 0, st, q2.d, k2.d, v2.d, a, opre.d);
            s->tape.push_back([=]() {
                hipLaunchKernelGGL(pcdattn_bwd_kernel, dim3(nc), dim3(256), 0, st, opre.g, q2.d, k2.d, v2.d, a, q2.g, k2.g, v2.g);
                linear_bwd(s, tr, wq, bq, q2.g, 12);
                linear_bwd(s, P, wk, bk, k2.g, 12);
                linear_bwd(s, P, wv, bv, v2.g, 12);
            });
            dense(s, opre, "pcd_attention.out_proj", 12, T_NONE, &o12);
        }
        // 12. point-wise translation: [P || o] 15 -> 3, GELU
        Ten X15 = make(s, (long long)nc * kN, 15);
        if (!s->failed) hipLaunchKernelGGL(cat15_fwd_kernel, dim3(nblk((long long)nc * kN * 15)), dim3(256), 0, st, P.d, o12.d, X15.d, (long long)nc * kN);
        s->tape.push_back([=]() { hipLaunchKernelGGL(cat15_bwd_kernel, dim3(nc), dim3(256), 0, st, X15.g, P.g, o12.g); });
        Ten pw = dense(s, X15, "point_wise_trans_layer.0", 3, T_GELU);
        // 13.-15. mask through the (N,-1,B,O) view (Q2), guidance, in-place update of x
        Ten gd = make(s, (long long)B * kN, 3), x = make(s, (long long)B * kN, 3);
        if (!s->failed) hipLaunchKernelGGL(guide_fwd_kernel, dim3(nblk((long long)B * kN * 3)), dim3(256), 0, st, pw.d, mask, hm.d, xt.d, B, gd.d, x.d, (long long)B * kN * 3);
        s->tape.push_back([=]() {
            // x = x_t + g: dg = dx
            hipLaunchKernelGGL(guide_bwd_kernel, dim3(nblk(ptot)), dim3(256), 0, st, x.g, mask, B, pw.g, hm.g, ptot);
        });
        // 16./17. per-point denoiser on x (the guiding-point pass does not enter the loss)
        Ten p1 = dense(s, x, "input_process.pose_embedding.0", 64, T_SIGMOID);
        Ten pose = cols(embx, 0, 128);
        dense(s, p1, "input_process.pose_embedding.2", 128, T_SIGMOID, &pose);
        Ten c0 = dense(s, embx, "input_process.combination_extraction.0", 192, T_SIGMOID);
        Ten c2_ = dense(s, c0, "input_process.combination_extraction.2", 128, T_SIGMOID);
        Ten f0 = dense(s, c2_, "output_process.pose_final.0", 64, T_GELU);
        Ten out = dense(s, f0, "output_process.pose_final.2", 3, T_GELU);
        if (s->failed) return;
        if (model_output) (void)hipMemcpyAsync(model_output, out.d, (size_t)B * kN * 3 * sizeof(float), hipMemcpyDeviceToDevice, st);
        // ---- losses ---------------------------------------------------------------------------------------------------------------------
        float* lt = talloc<float>(s, 2, true);         // [mse, cat]
        float* d2x = talloc<float>(s, (size_t)B * kN);
        float* d2y = talloc<float>(s, (size_t)B * kN);
        int* nx = talloc<int>(s, (size_t)B * kN);
        int* ny = talloc<int>(s, (size_t)B * kN);
        if (s->failed) return;
        launch_nn_dist(out.d, kN, x_start, kN, B, d2x, nx, st);
        launch_nn_dist(x_start, kN, out.d, kN, B, d2y, ny, st);
        hipLaunchKernelGGL(chamfer_loss_kernel, dim3(64), dim3(256), 0, st, d2x, d2y, (long long)B * kN, 1.0f / ((float)kN * (float)B), lt);
        hipLaunchKernelGGL(chamfer_bwd_kernel, dim3(nblk((long long)B * kN)), dim3(256), 0, st, out.d, x_start, nx, ny, kN, 2.0f * grad_scale / ((float)kN * (float)B), out.g,
                           (long long)B * kN);
        hipLaunchKernelGGL(catloss_kernel, dim3(nblk(B, 64)), dim3(64), 0, st, zc.d, target, B, C, lambda_cat, grad_scale, out_cat, zc.g, lt + 1);
        hipLaunchKernelGGL(add3_kernel, dim3(1), dim3(1), 0, st, lt, lt + 1, losses);
        // ---- backward: the tape in reverse ----------------------------------------------------------------------------------------------
        // (flags & 2: the objective only -- no reverse sweep, the bound gradient buffers are left as they are)
        if (!(flags & 2))
            for (auto it = s->tape.rbegin(); it != s->tape.rend() && !s->failed; ++it) (*it)();
    }
    };
    body();
    // Every exit of body() after the fork, the failing ones included, leaves the caller's stream BEHIND the geometry stream: FPS / ball-query work
    // still in flight reads given_objs, fps_starts and arena memory that the next step (or the caller's allocator, on `st`) may reuse.  On the
    // normal path the encoder block has already joined (the second wait is a no-op); on an early return the event is recorded here first.
    if (s->geo && s->ev_geo) {
        if (s->failed) (void)hipEventRecord(s->ev_geo, s->geo);
        (void)hipStreamWaitEvent(st, s->ev_geo, 0);
    }
    s->tape.clear();
    if (prev >= 0 && prev != s->device) (void)hipSetDevice(prev);
    if (s->failed) return LSDM_ERR_STATE;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return tfail(s, LSDM_ERR_HIP, "lsdm_train_step: %s", hipGetErrorString(e));
    return LSDM_OK;
}

}  // extern "C"
