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
#else
#define FD_T0() do {} while (0)
#define FD_MARK(slot) do {} while (0)
#define FD_FLUSH(base, ntiles) do {} while (0)
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

template <int N, int S, int D1, int C1, int C2>
__global__ __launch_bounds__(512) void fp_d_kernel(const float* __restrict__ points1, const float* __restrict__ y, const int* __restrict__ plan,
                                                   const float* __restrict__ nn_w, DenseH l1, DenseH l2, float* __restrict__ out, int p1_mod, int total_tiles) {
    using namespace fpd;
    constexpr int NP = 2, NWM = 4;
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
#ifdef LSDM_X_FPD_NODMA     // (timing experiment: results are wrong)
        if (t != R.first) return;
#endif
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
                for (int u = 0; u < KS1; ++u) { w1h[q][u] = gload<bf16x8>(wp + (size_t)u * ws); w1l[q][u] = gload<bf16x8>(wp + (size_t)u * ws + 16); }
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
                for (int u = 0; u < KS1; ++u) { xh[u] = *reinterpret_cast<const bf16x8*>(xp + 16 * u); xl[u] = *reinterpret_cast<const bf16x8*>(xp + 16 * u + PX); }
#pragma unroll
                for (int u = 0; u < KS1; ++u)
#pragma unroll
                    for (int q = 0; q < NCB; ++q) {
                        acc[q][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1h[q][u], xl[u], acc[q][0], 0, 0, 0);
                        acc[q][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1l[q][u], xh[u], acc[q][0], 0, 0, 0);
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
        for (int u = 0; u < KS2; ++u) { w2h[u] = gload<bf16x8>(wp2 + (size_t)u * ws2); w2l[u] = gload<bf16x8>(wp2 + (size_t)u * ws2 + 16); }
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
#ifdef LSDM_X_FPD_NOP1      // (timing experiment: results are wrong)
        if (t != R.first) return;
#endif
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
                const bf16x8 xh = *reinterpret_cast<const bf16x8*>(bp + u * 16), xl = *reinterpret_cast<const bf16x8*>(bp + u * 16 + PB);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, w2h[u], a2, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, w2l[u], a2, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, w2h[u], a2, 0, 0, 0);
            }
            FD_MARK(0);
            float* ob = out + (size_t)c * N * C2 + cb2 * 32 + pv;
#ifdef LSDM_X_FPD_NOSTORE   // (timing experiment: results are wrong)
            if (a2[0] == 12345.678f)
#endif
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

void launch_morton_order(int nc, const float* xyz, int* order, hipStream_t st) {
    hipLaunchKernelGGL((morton_order_kernel<1024>), dim3(nc), dim3(256), 0, st, xyz, order);
}

void launch_fp_plan(int nc, const int* order, int order_mod, const int* idx1, const int* nn_idx, const int* p1_remap, int* plan, hipStream_t st) {
    hipLaunchKernelGGL((fp_plan_kernel<1024, 256>), dim3(nc), dim3(256), 0, st, order, order_mod, idx1, nn_idx, p1_remap, plan);
}

// fp2 (1024 fine <- 256 coarse points per cloud, D1 = 64) on the plan of its clouds
void launch_fp_d(int nc, const float* points1, const float* ybuf, const float* nn_w, DenseH l1a, DenseH l2, float* out, const int* plan, hipStream_t st, int p1_mod) {
    const int total = 1024 / fpd::TM * nc;
    hipLaunchKernelGGL((fp_d_kernel<1024, 256, 64, 256, 128>), dim3(total < 256 ? total : 256), dim3(512), 0, st, points1, ybuf, plan, nn_w, l1a, l2, out, p1_mod, total);
}

}  // namespace lsdm
