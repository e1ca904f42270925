// fp2 with the gathered rows DEDUPLICATED through LDS (round 5, bf16x3): fp_plan_kernel + fp_d_kernel.
//
// PointNetFeaturePropagation (pointnet2_utils.py:273-312) interpolates, for every fine point, the rows of its three nearest coarse points
// (:286-303).  fp_ws_kernel fetches those three 1 KB rows of the pre-multiplied operand Y per POINT: 192 KB per 64-point tile through the
// texture path of the CU although a cloud has only 256 coarse rows -- every row is fetched twelve times per cloud.  What bounded that kernel
// was exactly this path (round 4: TA 68 % busy, both roles waiting for it) -- not the L2, which delivers 25 TB/s of gathered rows with four
// waves per CU (tools/ubench/gather2_bw.hip, profiles/r05_gather2_bw.txt).
//
// Here the fine points of a cloud are walked in MORTON order (fp_plan_kernel: 15-bit codes over the cloud's bounding box, bitonic sort in
// LDS), so that the 32 points of a tile are neighbours in space and share their coarse neighbours: 26 distinct rows per tile on the bench
// clouds instead of 96 requests.  The planner lists a tile's distinct rows and gives every point the LDS slots of its three rows; the main
// kernel lands each distinct row ONCE, by LDS-DMA (global_load_lds_dwordx4: one 1 KB row per wave instruction, no destination registers),
// and the matrix waves interpolate their accumulators straight out of those rows -- the interpolated tile, its LDS round trip and the
// gather role's arithmetic are gone.  Which tile a point lands in does not enter its arithmetic (rows of a layer stack are independent,
// interp3 sees the same three rows in the same order): results are bit-identical to fp_ws_kernel / fp_pre_kernel.
//
// fp_d_kernel is persistent (one work-group per CU, 8 waves) and TWO-STAGE: every SIMD holds one wave of each role, both issue MFMAs.
//   waves 4..7 ("A"): tile i      accumulators <- interp3 of the tile's rows (LDS), layer 1 over the p1 operand (K = 64, bufX[i & 1]),
//                                 bias + ReLU + hi/lo split -> bufB[i & 1]
//   waves 0..3 ("B"): tile i - 1  layer 2 (256 -> 128) from bufB[(i - 1) & 1], bias + ReLU, rows stored to their points' positions;
//                     tile i + 1  its p1 rows (coalesced 32-byte pieces), hi/lo split -> bufX[(i + 1) & 1]
// ALL weights are register-resident (A: its two layer-1 channel blocks, 64 VGPRs; B: its layer-2 channel block over all of K, 128), the row
// landing zone, bufX and bufB are double-buffered, ONE barrier per tile.  The A waves issue the LDS-DMA of tile i + 1 at the top of iteration i
// (plan records are read two tiles ahead), so a whole iteration covers its latency; the B waves' p1 loads fly under their own MFMAs.
// (First cut, round 5: the A waves loading their p1 fragments straight from global memory -- 8 scattered 16-byte loads per lane, the same
// rows in all four waves, 32 lines per instruction -- spent 1.55 of 2.97 us per tile issuing memory instructions: tools/fpd_phase.py.)
// A tile with more than CAP distinct rows keeps the first CAP (by coarse index) in LDS; the points of the others read them from global
// memory in the accumulator initialisation (slot < 0: a rare, divergent path -- correctness does not depend on the cloud's geometry).
#include <cstdlib>
#include <type_traits>

#include "bf16_core.hpp"
#include "kernels.hpp"

namespace lsdm {

namespace fpd {
constexpr int TM = 32;       // points per tile (one MFMA point block)
constexpr int CAP = 36;      // distinct rows of a tile that land in LDS
constexpr int RS = 260;      // floats per landed row (1 KB + 16 bytes: consecutive slots start four banks apart)
constexpr int HDR = 10;      // ints per tile header = 40 BYTES: [0] rows in LDS, [1 ..] their coarse row numbers (S <= 256)
constexpr int PT = 2;        // ints per point record: { fine position | p1 row << 10,  slot0 | slot1 << 10 | slot2 << 20 }; slot field: bit 9 set = the row
                             // did not get a slot, low bits = its coarse number (read from global memory), else the slot
}  // namespace fpd

// per-phase s_memtime sums of the two roles (-DLSDM_PHASE_TIMING builds only; tools/fpd_phase.py)
#ifdef LSDM_PHASE_TIMING
__device__ unsigned long long g_fpd_phase[64 * 16];   // 64 spread copies of 16 slots: role A 0..5 (+7 = tiles), role B 8..13
#define FD_T0() unsigned long long wt_ = __builtin_amdgcn_s_memtime(); unsigned long long wd_[6] = {}
#define FD_MARK(slot) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); wd_[slot] += n_ - wt_; wt_ = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define FD_FLUSH(base, ntiles) do { if (lane == 0 && (wave & 3) == 0) { unsigned long long* g_ = g_fpd_phase + (blockIdx.x & 63) * 16 + (base); \
        for (int i_ = 0; i_ < 6; ++i_) atomicAdd(g_ + i_, wd_[i_]); atomicAdd(g_ + 7, (unsigned long long)(ntiles)); } } while (0)
extern "C" __attribute__((visibility("default"))) int lsdm_debug_fpd_phase(unsigned long long* out16, int reset) {
    static unsigned long long h[64 * 16], z[64 * 16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fpd_phase), sizeof h) != hipSuccess) return -1;
    for (int s = 0; s < 16; ++s) { out16[s] = 0; for (int c = 0; c < 64; ++c) out16[s] += h[c * 16 + s]; }
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_fpd_phase), z, sizeof z) != hipSuccess) return -1;
    return 0;
}
__device__ unsigned long long g_fpc_phase[64 * 32];   // fp_c_kernel: 64 spread copies of 4 roles x 8 slots (slot 7 = iterations)
#define FC_FLUSH(role, iters) do { if (lane == 0 && (wave & 1) == 0) { unsigned long long* g_ = g_fpc_phase + (blockIdx.x & 63) * 32 + (role) * 8; \
        for (int i_ = 0; i_ < 6; ++i_) atomicAdd(g_ + i_, wd_[i_]); atomicAdd(g_ + 7, (unsigned long long)(iters)); } } while (0)
extern "C" __attribute__((visibility("default"))) int lsdm_debug_fpc_phase(unsigned long long* out32, int reset) {
    static unsigned long long h[64 * 32], z[64 * 32];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fpc_phase), sizeof h) != hipSuccess) return -1;
    for (int s = 0; s < 32; ++s) { out32[s] = 0; for (int c = 0; c < 64; ++c) out32[s] += h[c * 32 + s]; }
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_fpc_phase), z, sizeof z) != hipSuccess) return -1;
    return 0;
}
#else
#define FD_T0() do {} while (0)
#define FD_MARK(slot) do {} while (0)
#define FD_FLUSH(base, ntiles) do {} while (0)
#define FC_FLUSH(role, iters) do {} while (0)
#endif

size_t fp_plan_ints(int N) { return (size_t)N * fpd::PT + (size_t)(N / fpd::TM) * fpd::HDR; }

// Morton order of the N points of a cloud (one work-group per cloud): order[c][k] = index of the k-th point along the Z curve of its
// 32^3 grid over the cloud's bounding box; ties (same cell) in increasing index, so the order is deterministic.
template <int N>
__global__ __launch_bounds__(256) void morton_order_kernel(const float* __restrict__ xyz_all, int* __restrict__ order) {
    static_assert(N == 1024, "four keys per thread");
    __shared__ unsigned keys[N];
    __shared__ float red[6 * 4];
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* q4 = xyz_all + (size_t)c * N * 3 + 12 * tid;                  // points 4 tid .. 4 tid + 3
    float pv[12];
#pragma unroll
    for (int e = 0; e < 3; ++e) { const f32x4 v = *reinterpret_cast<const f32x4*>(q4 + 4 * e); pv[4 * e] = v[0]; pv[4 * e + 1] = v[1]; pv[4 * e + 2] = v[2]; pv[4 * e + 3] = v[3]; }
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], pv[3 * r + a]); hi[a] = fmaxf(hi[a], pv[3 * r + a]); }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], m)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], m)); }
    if (lane == 0)
#pragma unroll
        for (int a = 0; a < 3; ++a) { red[a * 4 + wave] = lo[a]; red[12 + a * 4 + wave] = hi[a]; }
    __syncthreads();
    float sc[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = fminf(fminf(red[a * 4], red[a * 4 + 1]), fminf(red[a * 4 + 2], red[a * 4 + 3]));
        hi[a] = fmaxf(fmaxf(red[12 + a * 4], red[12 + a * 4 + 1]), fmaxf(red[12 + a * 4 + 2], red[12 + a * 4 + 3]));
        sc[a] = hi[a] > lo[a] ? 32.0f / (hi[a] - lo[a]) : 0.0f;
    }
    auto spread = [](unsigned v) { v &= 31u; v = (v | (v << 8)) & 0x100fu; v = (v | (v << 4)) & 0x10c3u; v = (v | (v << 2)) & 0x1249u; return v; };
    // keys (Morton code << 10 | point: unique); element e of the sort = register e & 3 of thread e >> 2
    unsigned key[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const unsigned qx = (unsigned)fminf((pv[3 * r] - lo[0]) * sc[0], 31.0f), qy = (unsigned)fminf((pv[3 * r + 1] - lo[1]) * sc[1], 31.0f),
                       qz = (unsigned)fminf((pv[3 * r + 2] - lo[2]) * sc[2], 31.0f);
        key[r] = ((spread(qx) | (spread(qy) << 1) | (spread(qz) << 2)) << 10) | (unsigned)(4 * tid + r);
    }
    // bitonic network over N = 1024: partner distance j < 4 inside the thread, 4 <= j < 256 inside the wave (xor-shuffle of lane j / 4),
    // j >= 256 through LDS (three stages of the 55)
    auto cas = [](unsigned& a, unsigned& b, bool asc) { const unsigned lo_ = a < b ? a : b, hi_ = a < b ? b : a; a = asc ? lo_ : hi_; b = asc ? hi_ : lo_; };
    for (int k = 2; k <= N; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            const bool asc = ((4 * tid) & k) == 0;                         // (k >= 4: the same for the thread's four elements; k = 2 handled below)
            if (j == 1) {
                if (k == 2) { cas(key[0], key[1], true); cas(key[2], key[3], false); }
                else { cas(key[0], key[1], asc); cas(key[2], key[3], asc); }
            } else if (j == 2) {
                cas(key[0], key[2], asc); cas(key[1], key[3], asc);
            } else if (j < 256) {
                const int m = j >> 2;
                const bool low = (tid & m) == 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned o = (unsigned)__shfl_xor((int)key[r], m);
                    const unsigned mn = key[r] < o ? key[r] : o, mx = key[r] < o ? o : key[r];
                    key[r] = (low == asc) ? mn : mx;
                }
            } else {
                const int m = j >> 2;
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 4; ++r) keys[4 * tid + r] = key[r];
                __syncthreads();
                const bool low = (tid & m) == 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned o = keys[4 * (tid ^ m) + r];
                    const unsigned mn = key[r] < o ? key[r] : o, mx = key[r] < o ? o : key[r];
                    key[r] = (low == asc) ? mn : mx;
                }
            }
        }
    *reinterpret_cast<int4*>(order + (size_t)c * N + 4 * tid) = make_int4((int)(key[0] & 1023u), (int)(key[1] & 1023u), (int)(key[2] & 1023u), (int)(key[3] & 1023u));
}

// One work-group per cloud.  plan (per cloud): [N records of PT ints, in the walking order][N / TM headers of HDR ints]
//   record  = { n (position in the fine level) | p1 row (p1_remap[n] or n) << 10,  three 10-bit slot fields }  (fpd::PT; the interpolation weights stay
//   in nn_w, which the main kernel reads by n: a 32-byte record with the weights made the plan 39 KB per cloud and the planner HBM-bound)
// order: the Morton order of cloud c % order_mod (order_mod = 0: of cloud c).  idx1 == NULL: the order's entries are positions of the fine level.
// idx1 != NULL: they are SOURCE points and the fine level is the source cloud in the order idx1 (the step's level-1 FPS order; loop-invariant
// schedule: the source order is sorted once per call) -- position n of source point s from the inverse of idx1; a cloud whose idx1 is not a
// permutation (duplicated points: FPS returns [start, 0, 0, ...]) is walked in index order instead.
template <int N, int S>
__global__ __launch_bounds__(256) void fp_plan_kernel(const int* __restrict__ order, int order_mod, const int* __restrict__ idx1, const int* __restrict__ nn_idx,
                                                      const int* __restrict__ p1_remap, int* __restrict__ plan) {
    using namespace fpd;
    constexpr int NT = N / TM, WPT = S / 32;
    static_assert(N == 1024 && S % 32 == 0 && NT * WPT == 256 && S <= 256, "planner shape");
    __shared__ int pos[N];
    __shared__ unsigned bits[NT * WPT];
    __shared__ unsigned short pre[NT * WPT];
    __shared__ __attribute__((aligned(16))) int nnl[N * 3];
    const int c = blockIdx.x, tid = threadIdx.x;
    // neighbour lists of the cloud, coalesced, into LDS
#pragma unroll
    for (int e = 0; e < 3; ++e) reinterpret_cast<int4*>(nnl)[tid + 256 * e] = reinterpret_cast<const int4*>(nn_idx + (size_t)c * N * 3)[tid + 256 * e];
    const int4 ov = *reinterpret_cast<const int4*>(order + (size_t)(order_mod > 0 ? c % order_mod : c) * N + 4 * tid);
    int nk[4] = {ov.x, ov.y, ov.z, ov.w};                       // walking position 4 tid + r -> position in the fine level
    bits[tid] = 0u;
    if (idx1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) pos[4 * tid + r] = -1;
        __syncthreads();
        const int4 iv = *reinterpret_cast<const int4*>(idx1 + (size_t)c * N + 4 * tid);
        pos[iv.x] = 4 * tid; pos[iv.y] = 4 * tid + 1; pos[iv.z] = 4 * tid + 2; pos[iv.w] = 4 * tid + 3;
        __syncthreads();
        bool ok = true;
#pragma unroll
        for (int r = 0; r < 4; ++r) ok = ok && pos[4 * tid + r] >= 0;
        const bool perm = __syncthreads_and(ok) != 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) nk[r] = perm ? pos[nk[r]] : 4 * tid + r;
    } else
        __syncthreads();
    // distinct coarse rows of every tile: one bit per (tile, coarse row)
    const int t = (4 * tid) / TM;                               // (TM = 32: the thread's four points share a tile)
    int ni[4][3];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            ni[r][a] = nnl[3 * nk[r] + a];
            atomicOr(&bits[t * WPT + (ni[r][a] >> 5)], 1u << (ni[r][a] & 31));
        }
    __syncthreads();
    int* pl = plan + (size_t)c * (N * PT + NT * HDR);
    {
        // thread (tile tt, word w): rows of the word in front of it, its own rows into the header
        const int tt = tid / WPT, w = tid - tt * WPT;
        int base = 0;
        for (int v = 0; v < w; ++v) base += __popc(bits[tt * WPT + v]);
        pre[tid] = (unsigned short)base;
        unsigned char* hd = reinterpret_cast<unsigned char*>(pl + N * PT + tt * HDR);
        unsigned m = bits[tid];
        while (m) {
            const int b = __ffs(m) - 1;
            m &= m - 1u;
            if (base < CAP) hd[1 + base] = (unsigned char)(w * 32 + b);
            ++base;
        }
        if (w == WPT - 1) hd[0] = (unsigned char)(base < CAP ? base : CAP);
    }
    __syncthreads();
    int rec[4][PT];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int prow = p1_remap ? p1_remap[(size_t)c * N + nk[r]] : nk[r];
        rec[r][0] = nk[r] | (prow << 10);
        rec[r][1] = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int w = ni[r][a] >> 5;
            const int slot = (int)pre[t * WPT + w] + __popc(bits[t * WPT + w] & ((1u << (ni[r][a] & 31)) - 1u));
            rec[r][1] |= (slot < CAP ? slot : (0x200 | ni[r][a])) << (10 * a);
        }
    }
    int4* dst = reinterpret_cast<int4*>(pl + (size_t)(4 * tid) * PT);
    dst[0] = make_int4(rec[0][0], rec[0][1], rec[1][0], rec[1][1]);
    dst[1] = make_int4(rec[2][0], rec[2][1], rec[3][0], rec[3][1]);
}

template <int N, int S, int D1, int C1, int C2, int NP>
__global__ __launch_bounds__(512) void fp_d_kernel(const float* __restrict__ points1, const float* __restrict__ y, const int* __restrict__ plan,
                                                   const float* __restrict__ nn_w, DenseH l1, DenseH l2, float* __restrict__ out, int p1_mod, int total_tiles) {
    using namespace fpd;
    constexpr int NWM = 4;
    constexpr int LDB = ldh(C1), PB = TM * LDB, LDX = ldh(D1), PX = TM * LDX;
    constexpr int KS1 = D1 / 16, KS2 = C1 / 16, NCB = C1 / 32 / NWM;
    constexpr int TPC = N / TM;                                   // tiles per cloud
    constexpr size_t PLAN = (size_t)N * PT + (size_t)TPC * HDR;   // plan ints per cloud
    static_assert(C1 == 4 * 64 && C1 % (32 * NWM) == 0 && C2 == 32 * NWM && D1 == 64 && TM == 32,
                  "a Y row is one wave-wide 16-byte copy; one layer-2 channel block per B wave; a p1 row is eight 32-byte pieces");
    __shared__ __attribute__((aligned(16))) float rows[2][CAP * RS];
    __shared__ __attribute__((aligned(16))) __bf16 bufB[2][NP * PB];
    __shared__ __attribute__((aligned(16))) __bf16 bufX[2][NP * PX];
    __shared__ __attribute__((aligned(16))) float b1s[C1];
    __shared__ __attribute__((aligned(16))) int nlist[2][TM];
    __shared__ __attribute__((aligned(16))) float wts[2][3][TM];      // interpolation weights of the tile's points (B wave 0 -> the A waves)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int p = lane & 31, h = lane >> 5;
    for (int e = tid; e < C1; e += 512) b1s[e] = l1.b[e];
    const XcdRounds R = xcd_rounds(total_tiles);
    // rows of tile t (hdr: header byte `lane` of the tile) -> landing zone `buf`: each distinct row once, one 1 KB LDS-DMA per row, the rows dealt
    // round-robin over ALL EIGHT waves (issuing one takes a wave ~80 ns under load: the four A waves alone spent 0.6 of their 2.25 us per tile on it)
    auto fetch_rows = [&](int t, int hdr, int buf) {
        const int nr = __builtin_amdgcn_readlane(hdr, 0);
        const float* yb = y + (size_t)(t / TPC) * S * C1 + 4 * lane;
        const unsigned dst0 = lds_addr(&rows[buf][0]);
        for (int r = wave; r < nr; r += 2 * NWM) {
            const int row = __builtin_amdgcn_readlane(hdr, 1 + r);
            lds_dma16(yb + (size_t)row * C1, __builtin_amdgcn_readfirstlane(dst0 + (unsigned)r * (RS * 4)));
        }
    };
    auto hdr_load = [&](int t) -> int {
        return reinterpret_cast<const unsigned char*>(plan + (size_t)(t / TPC) * PLAN + N * PT + (t % TPC) * HDR)[lane < 4 * HDR ? lane : 0];
    };

    if (wave >= NWM) {
        // ---------------- role A: rows -> accumulators, layer 1, ReLU + split -> bufB ----------------
        const int aw = wave - NWM;
        bf16x8 w1h[NCB][KS1], w1l[NCB][KS1];
        {
            const size_t ws = (size_t)C1 * NP * 16;
#pragma unroll
            for (int q = 0; q < NCB; ++q) {
                const __bf16* wp = l1.w + ((size_t)((aw + q * NWM) * 32 + p) * NP) * 16 + 8 * h;
#pragma unroll
                for (int u = 0; u < KS1; ++u) {
                    w1h[q][u] = gload<bf16x8>(wp + (size_t)u * ws);
                    if constexpr (NP == 2) w1l[q][u] = gload<bf16x8>(wp + (size_t)u * ws + 16); else w1l[q][u] = w1h[q][u];
                }
            }
#pragma unroll
            for (int q = 0; q < NCB; ++q)
#pragma unroll
                for (int u = 0; u < KS1; ++u) asm volatile("" : "+v"(w1h[q][u]), "+v"(w1l[q][u]));
        }
        // plan words of a tile: header byte `lane` (lanes < 4 HDR) and this lane's point record
        struct Meta { int hdr; int2 rec; };
        auto meta_load = [&](int t) -> Meta {
            const int c = t / TPC, j = t % TPC;
            const int* pl = plan + (size_t)c * PLAN;
            Meta m;
            m.hdr = hdr_load(t);
            m.rec = *reinterpret_cast<const int2*>(pl + (size_t)(j * TM + p) * PT);
            return m;
        };
        auto fetch = [&](int t, const Meta& m, int buf) { fetch_rows(t, m.hdr, buf); };
        Meta mc{}, mn{}, mnn{};
        if (R.first < R.end) {
            mc = meta_load(R.first);
            if (R.first + R.stride < R.end) mn = meta_load(R.first + R.stride);
            fetch(R.first, mc, 0);
        }
        lds_dma_wait();
        wg_sync();                                       // b1s, the first tile's rows and p1 operand
        int i = 0;
        FD_T0();
#pragma unroll 1
        for (int t = R.first; ; t += R.stride, ++i) {
            if (t >= R.end) { wg_sync(); break; }        // (the B waves' last tile)
            const bool has_next = t + R.stride < R.end, has_next2 = t + 2 * R.stride < R.end;
            int pv = p, hv = h, lv = lane;
            asm volatile("" : "+v"(pv), "+v"(hv), "+v"(lv));       // (per-lane addresses formed per tile, not hoisted and kept across the loop)
            // The plan words carried into this iteration are USED here, in front of the new requests: hipcc's wait for them (it does not see
            // the s_waitcnt of lds_dma_wait) then sits at the loop top, where everything has long landed -- left to the first real use, it
            // came AFTER this iteration's requests and waited for those as well (vmcnt(0): a full memory round trip per tile).
            asm volatile("" :: "v"(mc.rec.x), "v"(mc.rec.y), "v"(mn.hdr), "v"(mn.rec.x), "v"(mn.rec.y));
            // next tile's rows (a whole iteration to land); the plan words of the tile after it
            if (has_next) fetch(t + R.stride, mn, (i + 1) & 1);
            if (has_next2) mnn = meta_load(t + 2 * R.stride);
            FD_MARK(0);
            // accumulators <- the three rows of the lane's point, interpolated (pointnet2_utils.py:299-303)
            const int s0 = mc.rec.y & 1023, s1 = (mc.rec.y >> 10) & 1023, s2 = (mc.rec.y >> 20) & 1023;      // (bit 9: no slot, the row's coarse number)
            const float w0 = wts[i & 1][0][pv], w1 = wts[i & 1][1][pv], w2 = wts[i & 1][2][pv];
            const float* rb = &rows[i & 1][0];
            f32x16 acc[NCB][1];
            if (__builtin_expect(__ballot(((s0 | s1 | s2) & 0x200) != 0) == 0ull, 1)) {
                const float *r0 = rb + s0 * RS + 4 * hv, *r1 = rb + s1 * RS + 4 * hv, *r2 = rb + s2 * RS + 4 * hv;
#pragma unroll
                for (int q = 0; q < NCB; ++q)
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        const int ch = (aw + q * NWM) * 32 + 8 * qd;
                        const f32x4 a = *reinterpret_cast<const f32x4*>(r0 + ch), b = *reinterpret_cast<const f32x4*>(r1 + ch), d = *reinterpret_cast<const f32x4*>(r2 + ch);
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[q][0][4 * qd + j] = interp3(a[j], b[j], d[j], w0, w1, w2);
                    }
            } else {
                const float* yc = y + (size_t)(t / TPC) * S * C1 + 4 * hv;
                auto row4 = [&](int s, int ch) -> f32x4 {
                    return (s & 0x200) == 0 ? *reinterpret_cast<const f32x4*>(rb + s * RS + 4 * hv + ch) : gload<f32x4>(yc + (size_t)(s & 0x1ff) * C1 + ch);
                };
#pragma unroll
                for (int q = 0; q < NCB; ++q)
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        const int ch = (aw + q * NWM) * 32 + 8 * qd;
                        const f32x4 a = row4(s0, ch), b = row4(s1, ch), d = row4(s2, ch);
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[q][0][4 * qd + j] = interp3(a[j], b[j], d[j], w0, w1, w2);
                    }
            }
            FD_MARK(1);
            // layer 1 against the resident fragments (per k-step: hi x lo, lo x hi, hi x hi -- the order of mma_kloop_pre_h)
            {
                const __bf16* xp = &bufX[i & 1][0] + pv * LDX + 8 * hv;
                bf16x8 xh[KS1], xl[KS1];
#pragma unroll
                for (int u = 0; u < KS1; ++u) {
                    xh[u] = *reinterpret_cast<const bf16x8*>(xp + 16 * u);
                    if constexpr (NP == 2) xl[u] = *reinterpret_cast<const bf16x8*>(xp + 16 * u + PX);
                }
#pragma unroll
                for (int u = 0; u < KS1; ++u)
#pragma unroll
                    for (int q = 0; q < NCB; ++q) {
                        if constexpr (NP == 2) {
                            acc[q][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1h[q][u], xl[u], acc[q][0], 0, 0, 0);
                            acc[q][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1l[q][u], xh[u], acc[q][0], 0, 0, 0);
                        }
                        acc[q][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1h[q][u], xh[u], acc[q][0], 0, 0, 0);
                    }
            }
            FD_MARK(2);
#pragma unroll
            for (int q = 0; q < NCB; ++q) {
                f32x4 bv1[4];
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) bv1[qd] = *reinterpret_cast<const f32x4*>(b1s + (aw + q * NWM) * 32 + 8 * qd + 4 * hv);
                store_lds_h<ACT_RELU, 1, NP>(acc[q], 0, aw + q * NWM, bv1, &bufB[i & 1][0], LDB, PB, lv);
            }
            if (aw == 0 && lv < TM) nlist[i & 1][lv] = mc.rec.x & 1023;
            mc = mn; mn = mnn;
            FD_MARK(3);
            lds_dma_wait();                              // the next tile's rows (this wave's share) have landed
            FD_MARK(4);
            wg_sync();
            FD_MARK(5);
        }
        FD_FLUSH(0, i);
        return;
    }
    // ---------------- role B: layer 2 of the previous tile + output rows; the p1 operand of the next tile ----------------
    const int cb2 = wave;
    const float bv2 = l2.b[cb2 * 32 + p];
    bf16x8 w2h[KS2], w2l[KS2];
    {
        const size_t ws2 = (size_t)C2 * NP * 16;
        const __bf16* wp2 = l2.w + ((size_t)(cb2 * 32 + p) * NP) * 16 + 8 * h;
#pragma unroll
        for (int u = 0; u < KS2; ++u) {
            w2h[u] = gload<bf16x8>(wp2 + (size_t)u * ws2);
            if constexpr (NP == 2) w2l[u] = gload<bf16x8>(wp2 + (size_t)u * ws2 + 16); else w2l[u] = w2h[u];
        }
#pragma unroll
        for (int u = 0; u < KS2; ++u) asm volatile("" : "+v"(w2h[u]), "+v"(w2l[u]));
    }
    // p1 operand: this wave's eight rows of a tile, lane = (row 8 wave + lane / 8, floats 8 (lane % 8) .. + 7)
    const int xr = 8 * wave + (lane >> 3), xc = 8 * (lane & 7);
    // word 0 of the plan record of the lane's row of tile t: p1 row << 10 | n.  (The RAW word is carried to its use a tile later: arithmetic on it
    // here made hipcc wait for the load -- and, in order, for every other load of the iteration -- in front of the matrix instructions: +25 us.)
    auto p1row_load = [&](int t) -> int { return plan[(size_t)(t / TPC) * PLAN + (size_t)((t % TPC) * TM + xr) * PT]; };
    f32x4 xa, xb;
    auto p1_request = [&](int t, int prow) {
        const int c = t / TPC;
        const float* pr = points1 + ((size_t)(p1_mod > 0 ? c % p1_mod : c) * N + prow) * D1 + xc;
        xa = load4f(pr); xb = load4f(pr + 4);
    };
    auto p1_store = [&](int buf) {
        __bf16* d = &bufX[buf][0] + xr * LDX + xc;
        store_split4<NP>(d, PX, xa[0], xa[1], xa[2], xa[3]);
        store_split4<NP>(d + 4, PX, xb[0], xb[1], xb[2], xb[3]);
    };
    // interpolation weights of a tile's points (wave 0, lanes 0 .. 31): nn_w is addressed by the point's position n -- three scattered 4-byte
    // loads per point, done ONCE per tile and handed to the A waves through LDS (each A wave loading its own cost 22 us per launch)
    auto wrec_load = [&](int t) -> int { return plan[(size_t)(t / TPC) * PLAN + (size_t)((t % TPC) * TM + p) * PT]; };      // (raw word, as above)
    float wq[3] = {0.0f, 0.0f, 0.0f};
    auto w_request = [&](int t, int n) {
        const float* wp = nn_w + ((size_t)(t / TPC) * N + n) * 3;
        wq[0] = wp[0]; wq[1] = wp[1]; wq[2] = wp[2];
    };
    auto w_store = [&](int buf) {
        if (lane < TM) { wts[buf][0][lane] = wq[0]; wts[buf][1][lane] = wq[1]; wts[buf][2][lane] = wq[2]; }
    };
    int prn = 0, wn = 0, hdn = 0;                        // plan words of the NEXT tile (requested a tile ahead of their use): p1 row, position n, header byte
    if (R.first < R.end) {
        fetch_rows(R.first, hdr_load(R.first), 0);
        if (R.first + R.stride < R.end) hdn = hdr_load(R.first + R.stride);
        p1_request(R.first, p1row_load(R.first) >> 10);
        if (wave == 0) w_request(R.first, wrec_load(R.first) & 1023);
        if (R.first + R.stride < R.end) { prn = p1row_load(R.first + R.stride); if (wave == 0) wn = wrec_load(R.first + R.stride); }
        p1_store(0);
        if (wave == 0) w_store(0);
    }
    lds_dma_wait();
    wg_sync();
    int i = 0;
    FD_T0();
#pragma unroll 1
    for (int t = R.first; ; t += R.stride, ++i) {
        const bool has_next = t + R.stride < R.end;
        asm volatile("" :: "v"(prn), "v"(wn), "v"(hdn));      // (the carried plan words are waited for HERE, in front of the new requests: see role A)
        if (has_next) {
            fetch_rows(t + R.stride, hdn, (i + 1) & 1);
            if (t + 2 * R.stride < R.end) hdn = hdr_load(t + 2 * R.stride);
            p1_request(t + R.stride, prn >> 10);
            if (wave == 0) w_request(t + R.stride, wn & 1023);
            if (t + 2 * R.stride < R.end) { prn = p1row_load(t + 2 * R.stride); if (wave == 0) wn = wrec_load(t + 2 * R.stride); }
        }
        if (i >= 1) {
            int pv = p, hv = h;
            asm volatile("" : "+v"(pv), "+v"(hv));
            const int c = (t - R.stride) / TPC;
            const __bf16* bp = &bufB[(i - 1) & 1][0] + pv * LDB + 8 * hv;
            f32x16 a2;
#pragma unroll
            for (int r = 0; r < 16; ++r) a2[r] = 0.0f;
#pragma unroll
            for (int u = 0; u < KS2; ++u) {
                const bf16x8 xh = *reinterpret_cast<const bf16x8*>(bp + u * 16);
                if constexpr (NP == 2) {
                    const bf16x8 xl = *reinterpret_cast<const bf16x8*>(bp + u * 16 + PB);
                    a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, w2h[u], a2, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, w2l[u], a2, 0, 0, 0);
                }
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, w2h[u], a2, 0, 0, 0);
            }
            FD_MARK(0);
            float* ob = out + (size_t)c * N * C2 + cb2 * 32 + pv;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int4 nl = *reinterpret_cast<const int4*>(&nlist[(i - 1) & 1][8 * g + 4 * hv]);      // accumulator r = 4 g + j <-> point 8 g + 4 h + j of the tile
                ob[(size_t)nl.x * C2] = reluf(a2[4 * g + 0] + bv2);
                ob[(size_t)nl.y * C2] = reluf(a2[4 * g + 1] + bv2);
                ob[(size_t)nl.z * C2] = reluf(a2[4 * g + 2] + bv2);
                ob[(size_t)nl.w * C2] = reluf(a2[4 * g + 3] + bv2);
            }
        }
        FD_MARK(1);
        if (has_next) { p1_store((i + 1) & 1); if (wave == 0) w_store((i + 1) & 1); }
        lds_dma_wait();                                  // this wave's share of the next tile's rows has landed
        wg_sync();
        FD_MARK(2);
        if (t >= R.end) break;
    }
    FD_FLUSH(8, i);
}

// ---------------------------------------------------------------------------------------------
// fp1 + head as a PIPELINE OF STAGES (round 5): fp_c_kernel.
//
// Round 4's form (fp_s_kernel, removed in round 6) carried 32 points per wave through the five layers in registers and streamed every layer's
// 64 KB of weight fragments through LDS: each 1 KB fragment read feeds two or three matrix instructions, so eight waves read 2 MB of fragments
// per round -- the LDS port is ~90 % busy against the matrix time -- and one barrier per layer keeps all waves of a CU in the same phase (MFMA
// pipe 37 %).  Here the roles are swapped once more: the WEIGHTS stay in registers and the ACTIVATIONS move.  A persistent work-group of eight waves per CU;
// wave 2 s + f owns channel blocks 2 f, 2 f + 1 of layer s (128 registers of hi + lo fragments, loaded once) for s = 0 .. 3 (the three fp1
// layers and conv1); a 32-point tile moves from stage to stage through double-buffered LDS tiles (bf16 hi / lo planes, stored in the k-slot
// order the next layer's fragments expect: one 16-byte piece per lane, k-step and plane), ONE barrier per tile, every stage working on a
// different tile at the same time:
//   iteration i:  gather (waves 0..5: six / four points each): rows of tile i (three 512-byte rows per point, two points per wave instruction,
//                 requested at the top of the iteration), interp3 + split -> X0[i & 1] at its end;
//                 stage s (waves 2 s, 2 s + 1): tile i - 1 - s from X_s[(i - 1) & 1] -> X_{s+1}[i & 1];
//                 conv2 (128 -> 3, zero-padded block): its k-steps are the fragments conv1's two waves hold in registers -- wave 6 contracts
//                 k-steps 0..3 of tile i - 4 and parks the accumulators in LDS, wave 7 continues with k-steps 4..7 an iteration later (the same
//                 accumulation order as one wave doing all eight) and stores the tile's rows.
// Same arithmetic in the same order as that form on the same fragment blob (api.hip::build_chain_blob): it was bit-identical to it.
// The points of a cloud are walked along `order` (optional; cloud c % order_mod's when order_mod > 0: the source cloud's Z curve); position -> point -> neighbour list -> rows is a chain of three dependent
// loads, each link requested one iteration ahead of the next.
namespace fpc {
constexpr int TM = 32, C = 128, KS = C / 16, NB = C / 32, NL = 4;
constexpr int ROWE = C + 8;              // bf16 elements per point and plane in an LDS tile (272 bytes: consecutive points start 4 banks apart)
constexpr int PLANE = TM * ROWE;
}  // namespace fpc

template <int N, int S, int NP>
__global__ __launch_bounds__(512) void fp_c_kernel(const float* __restrict__ points2, const int* __restrict__ nn_idx, const float* __restrict__ nn_w,
                                                   const __bf16* __restrict__ blob_w, const float* __restrict__ blob_b, float* __restrict__ out,
                                                   const int* __restrict__ order, int order_mod, int total_tiles) {
    using namespace fpc;
    constexpr int TPC = N / TM;
    constexpr int FL = NB * KS * NP;         // 1 KiB fragments per layer in the blob
    __shared__ __attribute__((aligned(16))) __bf16 X[NL][2][NP * PLANE];          // X[s]: input tile of stage s
    __shared__ __attribute__((aligned(16))) float part[2][64 * 16];               // conv2's accumulators between its two halves
    __shared__ __attribute__((aligned(16))) float bl[NL * C + 32];
    __shared__ __attribute__((aligned(16))) int nring[8][TM];                     // point numbers of the tiles in flight
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int p = lane & 31, h = lane >> 5;
    for (int e = tid; e < NL * C + 32; e += 512) bl[e] = blob_b[e];
    const XcdRounds R = xcd_rounds(total_tiles);
    const int ntile = R.first < R.end ? (R.end - R.first + R.stride - 1) / R.stride : 0;       // tiles of this work-group: R.first + i * R.stride
    const int stage = wave >> 1, half = wave & 1;

    // ---- resident weights: channel blocks 2 half, 2 half + 1 of layer `stage`; waves 6 / 7 also conv2's k-steps 0..3 / 4..7
    bf16x8 wh[2][KS], wl[2][KS];
    {
        const __bf16* wb = blob_w + (size_t)stage * FL * 512;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                wh[q][s] = gload<bf16x8>(wb + ((size_t)(((2 * half + q) * KS + s) * NP + 0) * 64 + lane) * 8);
                if constexpr (NP == 2) wl[q][s] = gload<bf16x8>(wb + ((size_t)(((2 * half + q) * KS + s) * NP + 1) * 64 + lane) * 8);
                else wl[q][s] = wh[q][s];
            }
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(wh[q][s]), "+v"(wl[q][s]));
    }
    // one stage step: acc <- bias, eight k-steps against the input tile (two channel blocks as two interleaved chains), ReLU + split; the
    // resulting fragments (k-steps 4 half .. 4 half + 3 of the next layer) go to `xout` or, for conv1, stay in yh / yl
    auto stage_step = [&](const __bf16* xin, __bf16* xout, bf16x8 (&yh)[4], bf16x8 (&yl)[4], int pv, int hv) {
        f32x16 a0, a1;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bl + stage * C + 32 * (2 * half) + 8 * qq + 4 * hv);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(bl + stage * C + 32 * (2 * half + 1) + 8 * qq + 4 * hv);
#pragma unroll
            for (int j = 0; j < 4; ++j) { a0[4 * qq + j] = b0[j]; a1[4 * qq + j] = b1[j]; }
        }
        const __bf16* xp = xin + pv * ROWE + 8 * hv;
        bf16x8 xh = *reinterpret_cast<const bf16x8*>(xp), xl = xh;
        if constexpr (NP == 2) xl = *reinterpret_cast<const bf16x8*>(xp + PLANE);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bf16x8 ch = xh, cl = xl;
            if (s + 1 < KS) {
                xh = *reinterpret_cast<const bf16x8*>(xp + 16 * (s + 1));
                if constexpr (NP == 2) xl = *reinterpret_cast<const bf16x8*>(xp + 16 * (s + 1) + PLANE);
            }
            if constexpr (NP == 2) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[0][s], cl, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[1][s], cl, a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[0][s], ch, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[1][s], ch, a1, 0, 0, 0);
            }
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[0][s], ch, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[1][s], ch, a1, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = reluf(q == 0 ? a0[8 * s2 + j] : a1[8 * s2 + j]);
                split8_h<NP>(v, yh[2 * q + s2], yl[2 * q + s2]);
                if (xout) {      // k-step 2 (2 half + q) + s2 of the next layer, this lane's 8 k-slots: one 16-byte piece per plane
                    __bf16* d = xout + pv * ROWE + 16 * (2 * (2 * half + q) + s2) + 8 * hv;
                    *reinterpret_cast<bf16x8*>(d) = yh[2 * q + s2];
                    if constexpr (NP == 2) *reinterpret_cast<bf16x8*>(d + PLANE) = yl[2 * q + s2];
                }
            }
    };

    if (wave < 6) {
        // ================= waves 0..5: gather + their stage =================
        // The rows are dealt over SIX waves -- waves 0..3 three pairs of points each, waves 4, 5 two.  The two waves of a SIMD (w and w + 4) run
        // their phases in OPPOSITE order so that one issues its matrix instructions while the other issues memory instructions:
        //   waves 0..3 ("early"): requests of tile i -> stage step -> interpolation of tile i
        //   waves 4, 5 ("late") : stage step -> interpolation of tile i (rows requested an iteration ago) -> requests of tile i + 1
        // (waves 6, 7 have no gather and start with their stage step as well).  With all gathering waves requesting first, both waves of SIMD 0 / 1
        // sat in their memory phase together and then contended for the matrix pipe: 2.8 us per iteration, tools/fpc_phase.py.
      auto gather_role = [&](auto npair_c, auto late_c) {        // (compile-time pair count and phase order: the row registers are sized by it)
        constexpr bool late = decltype(late_c)::value;
        constexpr int npair = decltype(npair_c)::value;           // pairs of points this wave gathers
        const int gp0 = wave < 4 ? 6 * wave : 24 + 4 * (wave - 4);  // its first point of the tile
        const int gl = lane & 7;                                  // lanes 0 .. 2 npair - 1 of the wave hold the plan words of its points
        auto point_load = [&](int g) -> int {                     // point number of the lane's point of tile g
            const int t = R.first + g * R.stride, c = t / TPC, k = (t % TPC) * TM + gp0 + (gl < 2 * npair ? gl : 0);
            return order ? order[(size_t)(order_mod > 0 ? c % order_mod : c) * N + k] : k;
        };
        auto list_load = [&](int g, int n, int (&ix)[3], float (&w)[3]) {
            const int t = R.first + g * R.stride;
            const size_t o = ((size_t)(t / TPC) * N + n) * 3;
#pragma unroll
            for (int j = 0; j < 3; ++j) { ix[j] = nn_idx[o + j]; w[j] = nn_w[o + j]; }
        };
        // plan words: (nq, ixq, wq) of the tile to be requested next, nq1 / (ixq1, wq1) one / one tile further; (ni, wi) of the tile whose rows are
        // in the registers
        int nq = 0, nq1 = 0, nq2 = 0, ni = 0;
        int ixq[3] = {0, 0, 0}, ixq1[3] = {0, 0, 0};
        float wq[3] = {0.0f, 0.0f, 0.0f}, wq1[3] = {0.0f, 0.0f, 0.0f}, wi[3] = {0.0f, 0.0f, 0.0f};
        f32x4 ra[npair], rb[npair], rd[npair];
        auto request = [&](int g, int pv, int hv) {               // rows of tile g (its list in ixq / wq); then the plan words shift by one tile
            const int t = R.first + g * R.stride;
            const float* pb = points2 + (size_t)(t / TPC) * S * C + 4 * pv;
#pragma unroll
            for (int q = 0; q < npair; ++q) {
                const int src = 2 * q + hv;
                ra[q] = load4f(pb + (size_t)__shfl(ixq[0], src) * C);
                rb[q] = load4f(pb + (size_t)__shfl(ixq[1], src) * C);
                rd[q] = load4f(pb + (size_t)__shfl(ixq[2], src) * C);
            }
            ni = nq;
#pragma unroll
            for (int j = 0; j < 3; ++j) wi[j] = wq[j];
            if (g + 1 < ntile) list_load(g + 1, nq1, ixq1, wq1);
            if (g + 2 < ntile) nq2 = point_load(g + 2);
        };
        auto shift = [&]() {                                      // (after the loads of request() have been waited for: next iteration's top)
            nq = nq1; nq1 = nq2;
#pragma unroll
            for (int j = 0; j < 3; ++j) { ixq[j] = ixq1[j]; wq[j] = wq1[j]; }
        };
        auto interp = [&](int i, int pv, int hv) {                // rows in the registers = tile i -> X0[i & 1]
#pragma unroll
            for (int q = 0; q < npair; ++q) {
                const int src = 2 * q + hv;
                const float u0 = __shfl(wi[0], src), u1 = __shfl(wi[1], src), u2 = __shfl(wi[2], src);
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = interp3(ra[q][j], rb[q][j], rd[q][j], u0, u1, u2);
                store_split4<NP>(&X[0][i & 1][0] + (gp0 + src) * ROWE + 4 * pv, PLANE, v[0], v[1], v[2], v[3]);
            }
            if (lane < 2 * npair) nring[i & 7][gp0 + lane] = ni;
        };
        if (ntile > 0) { nq = point_load(0); list_load(0, nq, ixq, wq); }
        if (ntile > 1) nq1 = point_load(1);
        if (late && ntile > 0) { request(0, p, h); shift(); }     // (late waves enter the loop with tile 0's rows in flight)
        __syncthreads();
        bf16x8 yh[4], yl[4];
        FD_T0();
#pragma unroll 1
        for (int i = 0; i < ntile + 6; ++i) {
            int pv = p, hv = h;
            asm volatile("" : "+v"(pv), "+v"(hv));
            const int ti = i - 1 - stage;
            if (!late) {
                asm volatile("" :: "v"(nq), "v"(nq1), "v"(ixq[0]), "v"(ixq[1]), "v"(ixq[2]), "v"(wq[0]), "v"(wq[1]), "v"(wq[2]));   // (carried plan words: waited for here)
                if (i < ntile) request(i, pv, hv);
                FD_MARK(0);
                if (ti >= 0 && ti < ntile) stage_step(&X[stage][(i - 1) & 1][0], &X[stage + 1][i & 1][0], yh, yl, pv, hv);
                FD_MARK(1);
                if (i < ntile) { interp(i, pv, hv); shift(); }
                FD_MARK(2);
            } else {
                if (ti >= 0 && ti < ntile) stage_step(&X[stage][(i - 1) & 1][0], &X[stage + 1][i & 1][0], yh, yl, pv, hv);
                FD_MARK(1);
                if (i < ntile) interp(i, pv, hv);
                FD_MARK(2);
                if (i + 1 < ntile) { request(i + 1, pv, hv); shift(); }
                FD_MARK(0);
            }
            wg_sync();
            FD_MARK(3);
        }
        FC_FLUSH(stage, ntile + 6);
      };
        if (wave < 4) gather_role(std::integral_constant<int, 3>{}, std::false_type{});
        else gather_role(std::integral_constant<int, 2>{}, std::true_type{});
        return;
    }
    // ================= waves 6, 7: conv1 on tile i - 4 + conv2 (k-steps 4 half .. 4 half + 3 from the fragments in registers) =================
    bf16x8 w5h[4], w5l[4];
    {
        const __bf16* w5g = blob_w + (size_t)NL * FL * 512;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            w5h[s] = gload<bf16x8>(w5g + ((size_t)((4 * half + s) * NP + 0) * 64 + lane) * 8);
            if constexpr (NP == 2) w5l[s] = gload<bf16x8>(w5g + ((size_t)((4 * half + s) * NP + 1) * 64 + lane) * 8);
            else w5l[s] = w5h[s];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(w5h[s]), "+v"(w5l[s]));
    }
    const float bv = blob_b[NL * C + (p < 3 ? p : 0)];
    __syncthreads();
    bf16x8 yh[4] = {}, yl[4] = {};
    FD_T0();
#pragma unroll 1
    for (int i = 0; i < ntile + 6; ++i) {
        int pv = p, hv = h, lv = lane;
        asm volatile("" : "+v"(pv), "+v"(hv), "+v"(lv));
        if (half == 1) {
            // conv2, second half, of tile i - 5: accumulators from wave 6 (parked an iteration ago), k-steps 4..7 = the fragments this wave formed
            // an iteration ago, then the tile's rows
            const int tj = i - 5;
            if (tj >= 0 && tj < ntile) {
                f32x16 acc;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(&part[(i - 1) & 1][(size_t)lv * 16 + 4 * g]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[4 * g + j] = v[j];
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if constexpr (NP == 2) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yl[s], w5h[s], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yh[s], w5l[s], acc, 0, 0, 0);
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yh[s], w5h[s], acc, 0, 0, 0);
                }
                const int t = R.first + tj * R.stride;
                float* ob = out + (size_t)(t / TPC) * N * 3 + pv;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int4 nl = *reinterpret_cast<const int4*>(&nring[tj & 7][8 * g + 4 * hv]);      // accumulator 4 g + j <-> point 8 g + 4 h + j of the tile
                    if (pv < 3) {
                        ob[(size_t)nl.x * 3] = acc[4 * g + 0] + bv;
                        ob[(size_t)nl.y * 3] = acc[4 * g + 1] + bv;
                        ob[(size_t)nl.z * 3] = acc[4 * g + 2] + bv;
                        ob[(size_t)nl.w * 3] = acc[4 * g + 3] + bv;
                    }
                }
            }
        }
        const int ti = i - 4;
        if (ti >= 0 && ti < ntile) {
            stage_step(&X[3][(i - 1) & 1][0], nullptr, yh, yl, pv, hv);
            if (half == 0) {
                // conv2, first half: k-steps 0..3, direct orientation D[point][channel]
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if constexpr (NP == 2) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yl[s], w5h[s], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yh[s], w5l[s], acc, 0, 0, 0);
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yh[s], w5h[s], acc, 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(&part[i & 1][(size_t)lv * 16 + 4 * g]) = f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
            }
        }
        FD_MARK(1);
        wg_sync();
        FD_MARK(3);
    }
    FC_FLUSH(3, ntile + 6);
}

void launch_fp_c(int np, int nc, const float* points2, const int* nn_idx, const float* nn_w, const __bf16* blob_w, const float* blob_b, float* out, hipStream_t st,
                 const int* order, int order_mod) {
    const int total = 1024 / fpc::TM * nc;
    if (np == 2)
        hipLaunchKernelGGL((fp_c_kernel<1024, 1024, 2>), dim3(total < 256 ? total : 256), dim3(512), 0, st, points2, nn_idx, nn_w, blob_w, blob_b, out, order, order_mod, total);
    else
        hipLaunchKernelGGL((fp_c_kernel<1024, 1024, 1>), dim3(total < 256 ? total : 256), dim3(512), 0, st, points2, nn_idx, nn_w, blob_w, blob_b, out, order, order_mod, total);
}

void launch_morton_order(int nc, const float* xyz, int* order, hipStream_t st) {
    hipLaunchKernelGGL((morton_order_kernel<1024>), dim3(nc), dim3(256), 0, st, xyz, order);
}

void launch_fp_plan(int nc, const int* order, int order_mod, const int* idx1, const int* nn_idx, const int* p1_remap, int* plan, hipStream_t st) {
    hipLaunchKernelGGL((fp_plan_kernel<1024, 256>), dim3(nc), dim3(256), 0, st, order, order_mod, idx1, nn_idx, p1_remap, plan);
}

// fp2 (1024 fine <- 256 coarse points per cloud, D1 = 64) on the plan of its clouds
void launch_fp_d(int np, int nc, const float* points1, const float* ybuf, const float* nn_w, DenseH l1a, DenseH l2, float* out, const int* plan, hipStream_t st, int p1_mod) {
    const int total = 1024 / fpd::TM * nc;
    if (np == 2)
        hipLaunchKernelGGL((fp_d_kernel<1024, 256, 64, 256, 128, 2>), dim3(total < 256 ? total : 256), dim3(512), 0, st, points1, ybuf, plan, nn_w, l1a, l2, out, p1_mod, total);
    else
        hipLaunchKernelGGL((fp_d_kernel<1024, 256, 64, 256, 128, 1>), dim3(total < 256 ? total : 256), dim3(512), 0, st, points1, ybuf, plan, nn_w, l1a, l2, out, p1_mod, total);
}

}  // namespace lsdm
