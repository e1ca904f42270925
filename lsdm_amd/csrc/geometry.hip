// PointNet++ geometry for gfx950: farthest-point sampling, ball query, 3-NN.
//
// Restates (never materialising the reference's (S,N) distance / sort tensors):
//   farthest_point_sample   model/pcd_backbone/pointnet2_utils.py:60-81
//   query_ball_point        model/pcd_backbone/pointnet2_utils.py:84-104
//   3-NN + IDW weights      model/pcd_backbone/pointnet2_utils.py:283-298
// All arithmetic is fp32 in the reference's rounding sequence (common.hpp::sqdist_ref) so that the
// discrete decisions (arg-max, d > r^2, nearest-3) are index-exact against the CPU oracle.
// Built with -ffp-contract=off; explicit __f*_rn intrinsics are used where the order matters.
#include "common.hpp"
#include "kernels.hpp"
#include <cmath>

namespace lsdm {

// ---------------------------------------------------------------------------------------------
// FPS (fps_chain_reg_kernel): one WAVE per cloud runs the four chained levels (1024->1024->256->64->16) entirely in
// registers.  The lanes hold N/64 points each (point p = lane + 64*j) as register vectors with their running minima.
// Per iteration
//   * a lane keeps only the MAXIMUM of its running minima (v_min + half a v_max3 per point; no index bookkeeping);
//   * the wave maximum is a 32-bit DPP reduction (six v_max_f32 with a DPP source);
//   * the winner is located by comparing the registers against that maximum in increasing j with early exit: lowest j,
//     then lowest lane = lowest point index -- ties resolve like torch.max on CPU (pointnet2_utils.py:80);
//   * its coordinates are read straight out of the owning lane's registers (uniform dynamic register index + v_readlane):
//     no LDS access and no memory latency inside the dependent chain.
// The selections of 64 consecutive iterations are collected one per lane (v_writelane) and flushed as coalesced stores.
// No LDS: a level hands its selection to the next through its own global output (fps_load_prev), so 16 waves fit a CU
// (register-bound) and a batched launch of the multi-step entry point (~4 clouds per SIMD) is bound by VALU issue, not by
// the length of the dependent chain.
// Batched launches cover G consecutive steps of one chain batch: cloud c of the launch is cloud (c % nc_step) of step
// (c / nc_step) -- the same input cloud every step, with that step's start draws starts[(step*4 + level)*nc_step + cloud]
// (the (K,4,B*9) layout of lsdm_sample_steps).  A single-step launch has nc_step == number of clouds.
// ---------------------------------------------------------------------------------------------
template <int PPT> struct FpsVec { typedef float type __attribute__((ext_vector_type(PPT))); };
template <> struct FpsVec<1> { typedef float type; };
template <int PPT> __device__ __forceinline__ float fv_get(const typename FpsVec<PPT>::type& v, int j) { return v[j]; }
template <> __device__ __forceinline__ float fv_get<1>(const float& v, int) { return v; }
template <int PPT> __device__ __forceinline__ void fv_set(typename FpsVec<PPT>::type& v, int j, float x) { v[j] = x; }
template <> __device__ __forceinline__ void fv_set<1>(float& v, int, float x) { v = x; }
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Packed fp32 arithmetic (two points per instruction; each half rounds exactly like the scalar form) written as EXPLICIT instructions on
// plain register pairs.  This unit is compiled with the compiler's own packed-fp32 selection OFF (csrc/Makefile: NOPK): what hipcc packs by
// itself broadcasts scalars through operand selection (op_sel / op_sel_hi) -- the form DESIGN.md 5 identifies as losing updates in a wave
// that shares its SIMD with a matrix wave, and these kernels run next to MFMA kernels on the library's side streams.  The forms below carry no
// operand selection and no neg modifier: a broadcast operand is a real pair (two VGPRs, or two SGPRs holding the same word), a subtraction
// adds the sign-flipped word (x + (-c) == x - c bit for bit).  `make check-packed` asserts it on the generated code.
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x2 pk_add_s(f32x2 a, unsigned long long b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b)); return r; }
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { f32x2 r; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// {-c, -c} as an SGPR pair from the (wave-uniform) bit pattern of c
__device__ __forceinline__ unsigned long long neg_pair_s(int bits) {
    const unsigned n = (unsigned)bits ^ 0x80000000u;
    return ((unsigned long long)n << 32) | n;
}
// squared distances of two points to the centroid whose negated coordinates are the SGPR pairs: ((dx dx + dy dy) + dz dz), no contraction
__device__ __forceinline__ f32x2 pk_sqdist_s(f32x2 x2, f32x2 y2, f32x2 z2, unsigned long long ncx, unsigned long long ncy, unsigned long long ncz) {
    const f32x2 dx = pk_add_s(x2, ncx), dy = pk_add_s(y2, ncy), dz = pk_add_s(z2, ncz);
    return pk_add(pk_add(pk_mul(dx, dx), pk_mul(dy, dy)), pk_mul(dz, dz));
}

// single-instruction min / max3 (the values on this path are never NaN, so the canonicalising v_max the compiler puts in
// front of fminf / fmaxf under IEEE mode is dead weight)
__device__ __forceinline__ float vmin_f32(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax3_f32(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float vmax_f32(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// wave64 max, one v_max_f32 with a DPP source per step (s_nop 1 = the two wait states a DPP read needs after a VALU
// write of the same register); returned as a wave-uniform value
__device__ __forceinline__ float wave_max_f32(float v) {
    asm volatile(
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// lowest j with any lane holding run[j] == wmax (ascending search with early exit: ~PPT/2 compares on average), and the
// lane mask of that j
template <int PPT, int J>
__device__ __forceinline__ void fps_find(const typename FpsVec<PPT>::type& run, float wmax, int& selj, unsigned long long& selm) {
    const unsigned long long m = __ballot(fv_get<PPT>(run, J) == wmax);
    if (J == PPT - 1 || m != 0ull) { selj = J; selm = m; return; }
    if constexpr (J + 1 < PPT) fps_find<PPT, J + 1>(run, wmax, selj, selm);
}

// the same over the first PPT entries of a W-wide register vector
template <int W, int PPT, int J>
__device__ __forceinline__ void fps_find_w(const typename FpsVec<W>::type& run, float wmax, int& selj, unsigned long long& selm) {
    const unsigned long long m = __ballot(fv_get<W>(run, J) == wmax);
    if (J == PPT - 1 || m != 0ull) { selj = J; selm = m; return; }
    if constexpr (J + 1 < PPT) fps_find_w<W, PPT, J + 1>(run, wmax, selj, selm);
}

// src: N points in registers; the NP selected points go to out_xyz (global), from where the next level reads them back
template <int N, int NP>
__device__ __forceinline__ void fps_level_reg(const typename FpsVec<N / 64>::type& px, const typename FpsVec<N / 64>::type& py,
                                              const typename FpsVec<N / 64>::type& pz, int start,
                                              int* __restrict__ out_idx, float* __restrict__ out_xyz) {
    constexpr int PPT = N / 64;
    static_assert(N % 64 == 0 && (NP <= 64 || NP % 64 == 0), "fps_level_reg tiling");
    const int lane = threadIdx.x;
    typename FpsVec<PPT>::type run;
#pragma unroll
    for (int j = 0; j < PPT; ++j) fv_set<PPT>(run, j, 1e10f);
    int sj = __builtin_amdgcn_readfirstlane(start >> 6), sl = __builtin_amdgcn_readfirstlane(start & 63);
    constexpr int BLK = NP < 64 ? NP : 64;
    for (int ib = 0; ib < NP; ib += BLK) {
        int sel_i = 0, sel_x = 0, sel_y = 0, sel_z = 0;   // lane ii of these holds the ii-th selection of the block
#pragma unroll 1
        for (int ii = 0; ii < BLK; ++ii) {
            // centroid = point (sl + 64*sj): registers of lane sl (uniform dynamic register index + v_readlane)
            const int icx = __builtin_amdgcn_readlane(__float_as_int(fv_get<PPT>(px, sj)), sl);
            const int icy = __builtin_amdgcn_readlane(__float_as_int(fv_get<PPT>(py, sj)), sl);
            const int icz = __builtin_amdgcn_readlane(__float_as_int(fv_get<PPT>(pz, sj)), sl);
            asm volatile("s_mov_b32 m0, %4\n\tv_writelane_b32 %0, %5, m0\n\tv_writelane_b32 %1, %6, m0\n\t"
                         "v_writelane_b32 %2, %7, m0\n\tv_writelane_b32 %3, %8, m0"
                         : "+v"(sel_i), "+v"(sel_x), "+v"(sel_y), "+v"(sel_z)
                         : "s"(ii), "s"(sl + 64 * sj), "s"(icx), "s"(icy), "s"(icz)
                         : "m0");
            const float cx = __int_as_float(icx), cy = __int_as_float(icy), cz = __int_as_float(icz);
            float bv;
            if constexpr (PPT == 1) {
                const float dx = __fsub_rn(fv_get<PPT>(px, 0), cx), dy = __fsub_rn(fv_get<PPT>(py, 0), cy), dz = __fsub_rn(fv_get<PPT>(pz, 0), cz);
                bv = vmin_f32(fv_get<PPT>(run, 0), sumsq3(dx, dy, dz));
                fv_set<PPT>(run, 0, bv);
            } else {
                // two points per packed instruction (v_pk_add/mul_f32 round each half like the scalar forms)
                const unsigned long long ncx = neg_pair_s(icx), ncy = neg_pair_s(icy), ncz = neg_pair_s(icz);
                float r[PPT];
#pragma unroll
                for (int j = 0; j < PPT; j += 2) {
                    const f32x2 x2 = {fv_get<PPT>(px, j), fv_get<PPT>(px, j + 1)};
                    const f32x2 y2 = {fv_get<PPT>(py, j), fv_get<PPT>(py, j + 1)};
                    const f32x2 z2 = {fv_get<PPT>(pz, j), fv_get<PPT>(pz, j + 1)};
                    const f32x2 d = pk_sqdist_s(x2, y2, z2, ncx, ncy, ncz);
                    r[j] = vmin_f32(fv_get<PPT>(run, j), d[0]);       // dist < distance ? dist : distance
                    r[j + 1] = vmin_f32(fv_get<PPT>(run, j + 1), d[1]);
                    fv_set<PPT>(run, j, r[j]);
                    fv_set<PPT>(run, j + 1, r[j + 1]);
                }
                bv = vmax_f32(r[0], r[1]);
#pragma unroll
                for (int j = 2; j < PPT; j += 2) bv = vmax3_f32(bv, r[j], r[j + 1]);
            }
            const float wmax = wave_max_f32(bv);
            // lowest point index holding the maximum: lowest j first, then lowest lane (torch.max on CPU keeps the first)
            unsigned long long selm;
            int selj;
            fps_find<PPT, 0>(run, wmax, selj, selm);
            sl = __builtin_amdgcn_readfirstlane(__ffsll((long long)selm) - 1);
            sj = __builtin_amdgcn_readfirstlane(selj);
        }
        if (lane < BLK) {
            out_idx[ib + lane] = sel_i;
            float* o = out_xyz + (size_t)(ib + lane) * 3;
            o[0] = __int_as_float(sel_x); o[1] = __int_as_float(sel_y); o[2] = __int_as_float(sel_z);
        }
    }
}

// ---- level 1 (1024 of 1024: a complete ordering) with the dead points COMPACTED away ------------------------------------------------
// A selected point has running minimum 0 and can never win again while any point with a positive minimum is left, so after i selections
// at least i of the 1024 points only cost instructions.  The level therefore runs in four epochs of 256 selections on 16, 12, 8 and 4
// registers per lane: between epochs the points with a positive running minimum are packed (stable: in increasing ORIGINAL index, so
// "lowest register, then lowest lane" stays "lowest point index", the tie rule of torch.max) -- 62 % of the point updates of the level.
// The pack goes through memory the kernel owns anyway and that is dead at that moment: coordinates and indices into the still unwritten
// TAIL of the level's own outputs, the running minima into the next level's output; read back with agent-scope loads like the hand-over
// between levels.  If the wave maximum is exactly 0, every point coincides with a selected one: the reference then selects index 0 for
// ever (first index of an all-equal row; all-zero padding clouds, clouds of duplicates) -- the rest of the output is that point.
template <int PPT, bool IDENT>
struct FpsState {      // IDENT: the unpacked first epoch, point (lane, j) IS original index lane + 64 j (no index registers)
    // (register vectors of a power-of-two width: a dynamically indexed 12-wide vector is lowered through scratch memory)
    static constexpr int W = PPT > 8 ? 16 : (PPT > 4 ? 8 : 4);
    typename FpsVec<W>::type px, py, pz, run;
    typename FpsVec<W>::type oi;       // original indices, as the bit patterns of floats (a register vector like the others)
    __device__ __forceinline__ int index_of(int j) const {
        if constexpr (IDENT) return (int)threadIdx.x + 64 * j;
        else return __float_as_int(fv_get<W>(oi, j));
    }
    // original index of the point in (uniform) register sj of lane sl
    __device__ __forceinline__ int index_at(int sj, int sl) const {
        if constexpr (IDENT) return sl + 64 * sj;
        else return __builtin_amdgcn_readlane(__float_as_int(fv_get<W>(oi, sj)), sl);
    }
};

// `count` selections (a multiple of 64) into out[ob ..]; returns true when the all-zero state was reached (outputs from *filled on are
// still to be written by the caller)
template <int PPT, bool IDENT>
__device__ __forceinline__ bool fps_epoch(FpsState<PPT, IDENT>& S, int& sj, int& sl, int ob, int count, int* __restrict__ out_idx,
                                          float* __restrict__ out_xyz, int* filled) {
    const int lane = threadIdx.x;
    for (int ib = 0; ib < count; ib += 64) {
        int sel_i = 0, sel_x = 0, sel_y = 0, sel_z = 0;
        int ii = 0;
        bool zero = false;
#pragma unroll 1
        for (; ii < 64; ++ii) {
            const int icx = __builtin_amdgcn_readlane(__float_as_int(fv_get<FpsState<PPT, IDENT>::W>(S.px, sj)), sl);
            const int icy = __builtin_amdgcn_readlane(__float_as_int(fv_get<FpsState<PPT, IDENT>::W>(S.py, sj)), sl);
            const int icz = __builtin_amdgcn_readlane(__float_as_int(fv_get<FpsState<PPT, IDENT>::W>(S.pz, sj)), sl);
            const int isel = S.index_at(sj, sl);
            asm volatile("s_mov_b32 m0, %4\n\tv_writelane_b32 %0, %5, m0\n\tv_writelane_b32 %1, %6, m0\n\t"
                         "v_writelane_b32 %2, %7, m0\n\tv_writelane_b32 %3, %8, m0"
                         : "+v"(sel_i), "+v"(sel_x), "+v"(sel_y), "+v"(sel_z)
                         : "s"(ii), "s"(isel), "s"(icx), "s"(icy), "s"(icz)
                         : "m0");
            const unsigned long long ncx = neg_pair_s(icx), ncy = neg_pair_s(icy), ncz = neg_pair_s(icz);
            float r[PPT];
#pragma unroll
            for (int j = 0; j < PPT; j += 2) {
                const f32x2 x2 = {fv_get<FpsState<PPT, IDENT>::W>(S.px, j), fv_get<FpsState<PPT, IDENT>::W>(S.px, j + 1)};
                const f32x2 y2 = {fv_get<FpsState<PPT, IDENT>::W>(S.py, j), fv_get<FpsState<PPT, IDENT>::W>(S.py, j + 1)};
                const f32x2 z2 = {fv_get<FpsState<PPT, IDENT>::W>(S.pz, j), fv_get<FpsState<PPT, IDENT>::W>(S.pz, j + 1)};
                const f32x2 d = pk_sqdist_s(x2, y2, z2, ncx, ncy, ncz);
                r[j] = vmin_f32(fv_get<FpsState<PPT, IDENT>::W>(S.run, j), d[0]);
                r[j + 1] = vmin_f32(fv_get<FpsState<PPT, IDENT>::W>(S.run, j + 1), d[1]);
                fv_set<FpsState<PPT, IDENT>::W>(S.run, j, r[j]);
                fv_set<FpsState<PPT, IDENT>::W>(S.run, j + 1, r[j + 1]);
            }
            float bv = vmax_f32(r[0], r[1]);
#pragma unroll
            for (int j = 2; j < PPT; j += 2) bv = vmax3_f32(bv, r[j], r[j + 1]);
            const float wmax = wave_max_f32(bv);
            if (wmax == 0.0f) { zero = true; ++ii; break; }
            unsigned long long selm;
            int selj;
            fps_find_w<FpsState<PPT, IDENT>::W, PPT, 0>(S.run, wmax, selj, selm);
            sl = __builtin_amdgcn_readfirstlane(__ffsll((long long)selm) - 1);
            sj = __builtin_amdgcn_readfirstlane(selj);
        }
        if (lane < ii) {
            out_idx[ob + ib + lane] = sel_i;
            float* o = out_xyz + (size_t)(ob + ib + lane) * 3;
            o[0] = __int_as_float(sel_x); o[1] = __int_as_float(sel_y); o[2] = __int_as_float(sel_z);
        }
        if (zero) { *filled = ob + ib + ii; return true; }
    }
    return false;
}

// stores the points of S with a positive running minimum into the scratch, packed in increasing original index; returns their number
template <int PA, bool IDENT>
__device__ __forceinline__ int fps_pack_store(const FpsState<PA, IDENT>& S, int* __restrict__ s_idx, float* __restrict__ s_xyz, float* __restrict__ s_run) {
    int base = 0;
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        const bool alive = fv_get<FpsState<PA, IDENT>::W>(S.run, j) > 0.0f;
        const unsigned long long m = __ballot(alive);
        const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (alive) {
            s_idx[pos] = S.index_of(j);
            s_xyz[pos * 3 + 0] = fv_get<FpsState<PA, IDENT>::W>(S.px, j); s_xyz[pos * 3 + 1] = fv_get<FpsState<PA, IDENT>::W>(S.py, j); s_xyz[pos * 3 + 2] = fv_get<FpsState<PA, IDENT>::W>(S.pz, j);
            s_run[pos] = fv_get<FpsState<PA, IDENT>::W>(S.run, j);
        }
        base += __popcll(m);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return base;
}
// ... and loads them as a state of fewer registers per lane; `want` (original index of the next centroid) -> (sj, sl)
template <int PB>
__device__ __forceinline__ void fps_pack_load(FpsState<PB, false>& T, int count, const int* __restrict__ s_idx, const float* __restrict__ s_xyz,
                                              const float* __restrict__ s_run, int want, int& sj, int& sl) {
    const int lane = threadIdx.x;
    sj = 0; sl = 0;
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        const int q = lane + 64 * j;
        const bool live = q < count;         // (count <= 64 PB: every selection so far removed at least one point)
        const int qq = live ? q : 0;
        const float x = __hip_atomic_load(s_xyz + qq * 3 + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float y = __hip_atomic_load(s_xyz + qq * 3 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float z = __hip_atomic_load(s_xyz + qq * 3 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float rv = __hip_atomic_load(s_run + qq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int id = __hip_atomic_load(s_idx + qq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fv_set<FpsState<PB, false>::W>(T.px, j, live ? x : 0.0f); fv_set<FpsState<PB, false>::W>(T.py, j, live ? y : 0.0f); fv_set<FpsState<PB, false>::W>(T.pz, j, live ? z : 0.0f);
        fv_set<FpsState<PB, false>::W>(T.run, j, live ? rv : 0.0f);      // filler slots: running minimum 0 = never selected while a real point is left
        const int idv = live ? id : 0x7fffffff;
        fv_set<FpsState<PB, false>::W>(T.oi, j, __int_as_float(idv));
        const unsigned long long m = __ballot(idv == want);
        if (m) { sj = j; sl = __ffsll((long long)m) - 1; }
    }
    sj = __builtin_amdgcn_readfirstlane(sj); sl = __builtin_amdgcn_readfirstlane(sl);
    // the reads must have returned before the scratch (the tail of the outputs) is overwritten by the epoch's selections
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__device__ __forceinline__ void fps_level1_compact(const FpsVec<16>::type& px, const FpsVec<16>::type& py, const FpsVec<16>::type& pz, int start,
                                                   const float* __restrict__ src, int* __restrict__ out_idx, float* __restrict__ out_xyz,
                                                   float* __restrict__ run_scratch) {
    const int lane = threadIdx.x;
    int filled = 1024, count = 0, want = 0;
    bool zero;
    int sj = __builtin_amdgcn_readfirstlane(start >> 6), sl = __builtin_amdgcn_readfirstlane(start & 63);
    // The point selected LAST in an epoch (the centroid of the next iteration) has not been applied yet: its own running minimum is still
    // positive, so it survives the pack and is found again by its index (`want`).
    {
        FpsState<16, true> A;
        A.px = px; A.py = py; A.pz = pz;
        // (laundered: otherwise the pack after the epoch is matched to the (x, y, z) triples the loads delivered, which are then kept
        // alive next to the coordinate vectors -- 48 registers spilled in the prologue and reloaded in fps_pack_store)
        asm volatile("" : "+v"(A.px), "+v"(A.py), "+v"(A.pz));
#pragma unroll
        for (int j = 0; j < 16; ++j) A.run[j] = 1e10f;
        zero = fps_epoch<16, true>(A, sj, sl, 0, 256, out_idx, out_xyz, &filled);
        if (!zero) { want = A.index_at(sj, sl); count = fps_pack_store<16, true>(A, out_idx + 256, out_xyz + 256 * 3, run_scratch); }
    }
    if (!zero) {
        FpsState<12, false> B;
        fps_pack_load<12>(B, count, out_idx + 256, out_xyz + 256 * 3, run_scratch, want, sj, sl);
        zero = fps_epoch<12, false>(B, sj, sl, 256, 256, out_idx, out_xyz, &filled);
        if (!zero) { want = B.index_at(sj, sl); count = fps_pack_store<12, false>(B, out_idx + 512, out_xyz + 512 * 3, run_scratch); }
    }
    if (!zero) {
        FpsState<8, false> B;
        fps_pack_load<8>(B, count, out_idx + 512, out_xyz + 512 * 3, run_scratch, want, sj, sl);
        zero = fps_epoch<8, false>(B, sj, sl, 512, 256, out_idx, out_xyz, &filled);
        if (!zero) { want = B.index_at(sj, sl); count = fps_pack_store<8, false>(B, out_idx + 768, out_xyz + 768 * 3, run_scratch); }
    }
    if (!zero) {
        FpsState<4, false> B;
        fps_pack_load<4>(B, count, out_idx + 768, out_xyz + 768 * 3, run_scratch, want, sj, sl);
        zero = fps_epoch<4, false>(B, sj, sl, 768, 256, out_idx, out_xyz, &filled);
    }
    if (zero) {      // every remaining selection is point 0 (see above)
        const float x0 = src[0], y0 = src[1], z0 = src[2];
        for (int q = filled + lane; q < 1024; q += 64) {
            out_idx[q] = 0;
            out_xyz[(size_t)q * 3 + 0] = x0; out_xyz[(size_t)q * 3 + 1] = y0; out_xyz[(size_t)q * 3 + 2] = z0;
        }
    }
}

// Hand-over between levels through the level's own output in global memory: the wave waits for its stores (s_waitcnt vmcnt(0):
// they are complete in L2, the vector L1 is write-through) and reads the points back with agent-scope loads, which bypass
// the vector L1 -- no LDS at all, so the number of resident waves is bounded by registers only (4 per SIMD).
template <int PPT>
__device__ __forceinline__ void fps_load_prev(const float* prev_xyz, typename FpsVec<PPT>::type& px, typename FpsVec<PPT>::type& py,
                                              typename FpsVec<PPT>::type& pz) {
    const int lane = threadIdx.x;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const float* q = prev_xyz + (size_t)(lane + 64 * j) * 3;
        fv_set<PPT>(px, j, __hip_atomic_load(q + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        fv_set<PPT>(py, j, __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        fv_set<PPT>(pz, j, __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(LSDM_FPS_WPE, LSDM_FPS_WPE))) void fps_chain_reg_kernel(
    const float* __restrict__ clouds, const int64_t* __restrict__ starts, int nc_step,
    int* __restrict__ idx1, int* __restrict__ idx2, int* __restrict__ idx3, int* __restrict__ idx4,
    float* __restrict__ xyz1, float* __restrict__ xyz2, float* __restrict__ xyz3, float* __restrict__ xyz4) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const int step = c / nc_step, cl = c - step * nc_step;
    const float* src = clouds + (size_t)cl * kPoints * 3;
    const int64_t* st4 = starts + (size_t)step * 4 * nc_step + cl;
    const int s1 = (int)st4[0 * (size_t)nc_step], s2 = (int)st4[1 * (size_t)nc_step];
    const int s3 = (int)st4[2 * (size_t)nc_step], s4 = (int)st4[3 * (size_t)nc_step];
    {
        FpsVec<16>::type px, py, pz;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int p = lane + 64 * j;
            px[j] = src[p * 3 + 0]; py[j] = src[p * 3 + 1]; pz[j] = src[p * 3 + 2];
        }
        // (the plain form fps_level_reg<1024, 1024> measured 186 us per step against 157: same box, B = 64, indices identical)
        fps_level1_compact(px, py, pz, s1, src, idx1 + (size_t)c * 1024, xyz1 + (size_t)c * 1024 * 3, xyz2 + (size_t)c * 256 * 3);
        fps_load_prev<16>(xyz1 + (size_t)c * 1024 * 3, px, py, pz);
        fps_level_reg<1024, 256>(px, py, pz, s2, idx2 + (size_t)c * 256, xyz2 + (size_t)c * 256 * 3);
    }
    {
        FpsVec<4>::type px, py, pz;
        fps_load_prev<4>(xyz2 + (size_t)c * 256 * 3, px, py, pz);
        fps_level_reg<256, 64>(px, py, pz, s3, idx3 + (size_t)c * 64, xyz3 + (size_t)c * 64 * 3);
    }
    {
        float px, py, pz;
        fps_load_prev<1>(xyz3 + (size_t)c * 64 * 3, px, py, pz);
        fps_level_reg<64, 16>(px, py, pz, s4, idx4 + (size_t)c * 16, xyz4 + (size_t)c * 16 * 3);
    }
}

// ---------------------------------------------------------------------------------------------
// Ball query: one wave per centroid scans the N source points in index order, 64 at a time, and
// keeps the first 32 with NOT(d > r^2); unfilled slots repeat the first hit.  No sort.
// Workgroup = 4 waves x CPW centroids each; the source cloud is staged once in LDS as (x,y,z,|p|^2).
// ---------------------------------------------------------------------------------------------
// src_mod > 0: the source clouds repeat with period src_mod (level-0 clouds of a multi-step batch), else one per cloud
//
// N >= 128: a lane tests TWO points per step (p = 128k + lane and p + 64) with packed fp32 instructions -- v_pk_mul /
// v_pk_fma / v_pk_add round each half exactly like the scalar forms, so the d > r^2 decisions are unchanged -- and the
// slot bookkeeping (prefix count, LDS store) only runs when one of the 128 points is inside the ball, which is rare at
// the small radii of the fine levels.  The cloud is staged as pairs {x_a x_b y_a y_b | z_a z_b n_a n_b} so that a
// lane's two points are two conflict-free ds_read_b128.
template <int N>
__global__ __launch_bounds__(256) void ball_query_kernel(const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                                         int S, float r2, int* __restrict__ out, int src_mod,
                                                         const unsigned char* __restrict__ only) {
    // only (optional, one flag per centroid): centroids whose flag is 0 are skipped, their `out` rows left alone (sa2 memo)
    constexpr int CPW = 4;
    constexpr bool PAIRED = N >= 128;
    constexpr int NPAIR = PAIRED ? N / 128 : 1;
    __shared__ f32x4 pts[PAIRED ? NPAIR * 2 * 64 : N];
    __shared__ int slots[4][kNSample];
    int c, tile;
    xcd_cloud_tile((S + 15) / 16, c, tile);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* src = xyz + (size_t)(src_mod > 0 ? c % src_mod : c) * N * 3;
    for (int p = tid; p < N; p += 256) {
        const float x = src[p * 3 + 0], y = src[p * 3 + 1], z = src[p * 3 + 2];
        const float n2 = sumsq3(x, y, z);
        if constexpr (PAIRED) {
            float* q = reinterpret_cast<float*>(&pts[((p >> 7) * 2) * 64 + (p & 63)]);
            const int hb = (p >> 6) & 1;
            q[hb] = x; q[2 + hb] = y; q[64 * 4 + hb] = z; q[64 * 4 + 2 + hb] = n2;
        } else {
            pts[p] = f32x4{x, y, z, n2};
        }
    }
    __syncthreads();
    for (int q = 0; q < CPW; ++q) {
        const int s = (tile * 4 + wave) * CPW + q;
        if (s >= S) break;  // wave-uniform
        if (only && !only[(size_t)c * S + s]) continue;   // wave-uniform
        const float* cp = new_xyz + ((size_t)c * S + s) * 3;
        const float ax = cp[0], ay = cp[1], az = cp[2];
        const float a2 = sumsq3(ax, ay, az);
        int cnt = 0;
        if constexpr (PAIRED) {
            const f32x2 ax2 = {ax, ax}, ay2 = {ay, ay}, az2 = {az, az}, a22 = {a2, a2}, m2 = {-2.0f, -2.0f};
            for (int k = 0; k < NPAIR; ++k) {
                if (cnt >= kNSample) break;
                const f32x4 u = pts[(k * 2) * 64 + lane], v = pts[(k * 2 + 1) * 64 + lane];
                const f32x2 bx = {u[0], u[1]}, by = {u[2], u[3]}, bz = {v[0], v[1]}, b2 = {v[2], v[3]};
                // common.hpp::sqdist_ref on both halves
                f32x2 dot = pk_mul(ax2, bx);
                dot = pk_fma(ay2, by, dot);
                dot = pk_fma(az2, bz, dot);
                f32x2 d = pk_mul(m2, dot);
                d = pk_add(d, a22);
                d = pk_add(d, b2);
                const bool in_a = !(d[0] > r2), in_b = !(d[1] > r2);
                const unsigned long long ma = __ballot(in_a), mb = __ballot(in_b);
                if ((ma | mb) == 0ull) continue;
                const unsigned long long below = (1ull << lane) - 1ull;
                const int pos_a = cnt + __popcll(ma & below);
                if (in_a && pos_a < kNSample) slots[wave][pos_a] = k * 128 + lane;
                cnt += __popcll(ma);
                const int pos_b = cnt + __popcll(mb & below);
                if (in_b && pos_b < kNSample) slots[wave][pos_b] = k * 128 + 64 + lane;
                cnt += __popcll(mb);
            }
        } else {
            for (int base = 0; base < N && cnt < kNSample; base += 64) {
                const int p = base + lane;
                bool in = false;
                if (p < N) {
                    const f32x4 b = pts[p];
                    const float d = sqdist_ref(ax, ay, az, a2, b[0], b[1], b[2], b[3]);
                    in = !(d > r2);
                }
                const unsigned long long m = __ballot(in);
                const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
                if (in && pos < kNSample) slots[wave][pos] = p;
                cnt += __popcll(m);
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < kNSample) {
            // cnt == 0 cannot happen for a centroid drawn from the cloud (the reference would index out of range)
            const int first = slots[wave][0];
            const int v = lane < cnt ? slots[wave][lane] : first;
            out[((size_t)c * S + s) * kNSample + lane] = v;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------
// Ball query over a uniform grid (the finest level: 1024 source points, 1024 centroids).
//
// query_ball_point keeps the first 32 source indices, in index order, with NOT(d > r^2).  A point can only pass that test
// if it lies in one of the 27 cells around the centroid's cell when the cell edge is >= 1.02 r: two points whose cell
// coordinates differ by >= 2 on some axis are >= 1.02 r apart, i.e. d >= 1.04 r^2, far outside the rounding of the
// fp32 distance (~1e-6).  So each work-group
//   1. bins the cloud into a 16^3 grid over its bounding box (cell edge = max(1.02 r, extent / 16) per axis; counting
//      sort in LDS: histogram, scan, scatter),
//   2. lets ONE THREAD per centroid walk the nine x-rows of cells around it (each row is one contiguous range of the
//      sorted array), evaluate the reference's distance expression (common.hpp::sqdist_ref) on those candidates only, and
//      set bit `index` of a private 1024-bit mask,
//   3. reads the 32 lowest set bits of the mask -- the candidates are unordered, the mask restores index order.
// Index-exact against the scan over all points: the candidate set is a superset of the hits, and the decision on every
// candidate is the same expression.  A cloud that crowds > 256 points into one cell (degenerate clouds: the all-zero
// padding slots) takes the ordered scan instead (ball_scan_wave), whose early exit is what makes those cheap.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void ball_scan_wave(const f32x4* pts2, int* slots, int lane, float ax, float ay, float az, float r2,
                                               int* __restrict__ dst) {
    // pts2: the paired layout of ball_query_kernel<1024>; one wave, one centroid
    const float a2 = sumsq3(ax, ay, az);
    const f32x2 ax2 = {ax, ax}, ay2 = {ay, ay}, az2 = {az, az}, a22 = {a2, a2}, m2 = {-2.0f, -2.0f};
    int cnt = 0;
    for (int k = 0; k < 8 && cnt < kNSample; ++k) {
        const f32x4 u = pts2[(k * 2) * 64 + lane], v = pts2[(k * 2 + 1) * 64 + lane];
        const f32x2 bx = {u[0], u[1]}, by = {u[2], u[3]}, bz = {v[0], v[1]}, b2 = {v[2], v[3]};
        f32x2 dot = pk_mul(ax2, bx);
        dot = pk_fma(ay2, by, dot);
        dot = pk_fma(az2, bz, dot);
        f32x2 d = pk_mul(m2, dot);
        d = pk_add(d, a22);
        d = pk_add(d, b2);
        const bool in_a = !(d[0] > r2), in_b = !(d[1] > r2);
        const unsigned long long ma = __ballot(in_a), mb = __ballot(in_b);
        if ((ma | mb) == 0ull) continue;
        const unsigned long long below = (1ull << lane) - 1ull;
        const int pos_a = cnt + __popcll(ma & below);
        if (in_a && pos_a < kNSample) slots[pos_a] = k * 128 + lane;
        cnt += __popcll(ma);
        const int pos_b = cnt + __popcll(mb & below);
        if (in_b && pos_b < kNSample) slots[pos_b] = k * 128 + 64 + lane;
        cnt += __popcll(mb);
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < kNSample) dst[lane] = lane < cnt ? slots[lane] : slots[0];
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(256) void ball_query_grid_kernel(const float* __restrict__ xyz, const float* __restrict__ new_xyz, int S,
                                                              float r2, float cell_min, int* __restrict__ out, int src_mod) {
    constexpr int N = 1024, G = 16, NCELL = G * G * G, NT = 256, MS = NT + 1;   // MS: padded row of the per-thread mask / slot table
    __shared__ f32x4 spt[N];                     // grid path: points sorted by cell (x,y,z,|p|^2); scan path: the paired layout
    __shared__ unsigned short sidx[N];           // original index of a sorted point
    __shared__ unsigned int cend[NCELL];         // histogram -> running fill pointer -> end of the cell's range
    __shared__ unsigned short cbeg[NCELL];       // start of the cell's range
    __shared__ unsigned int tab[32 * MS];        // [word or slot][thread]: private 1024-bit masks, then the 32 result slots
    __shared__ float red[8][6];
    __shared__ unsigned int wsum[4];
    __shared__ int crowded;
    int c, tile;
    xcd_cloud_tile(S / NT, c, tile);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* src = xyz + (size_t)(src_mod > 0 ? c % src_mod : c) * N * 3;
    const float* qsrc = new_xyz + (size_t)c * S * 3;
    // ---- bounding box ---------------------------------------------------------------------------------------------------
    float px[4], py[4], pz[4];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = tid + NT * k;
        px[k] = src[p * 3 + 0]; py[k] = src[p * 3 + 1]; pz[k] = src[p * 3 + 2];
        lo[0] = fminf(lo[0], px[k]); hi[0] = fmaxf(hi[0], px[k]);
        lo[1] = fminf(lo[1], py[k]); hi[1] = fmaxf(hi[1], py[k]);
        lo[2] = fminf(lo[2], pz[k]); hi[2] = fmaxf(hi[2], pz[k]);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], o));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o));
        }
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { red[wave][a] = lo[a]; red[wave][3 + a] = hi[a]; }
    }
    for (int e = tid; e < NCELL; e += NT) cend[e] = 0u;
    if (tid == 0) crowded = 0;
    __syncthreads();
    float org[3], inv[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float l = fminf(fminf(red[0][a], red[1][a]), fminf(red[2][a], red[3][a]));
        const float h = fmaxf(fmaxf(red[0][3 + a], red[1][3 + a]), fmaxf(red[2][3 + a], red[3][3 + a]));
        org[a] = l;
        inv[a] = 1.0f / fmaxf(cell_min, (h - l) * (1.0f / G));
    }
    auto cell_of = [&](float x, float y, float z, int& ix, int& iy, int& iz) {   // the same expression for points and centroids
        ix = min(G - 1, max(0, (int)((x - org[0]) * inv[0])));
        iy = min(G - 1, max(0, (int)((y - org[1]) * inv[1])));
        iz = min(G - 1, max(0, (int)((z - org[2]) * inv[2])));
    };
    // exclusive scan of the 4096 counts in cend (16 consecutive cells per thread, wave scan of the thread sums, 4 wave sums);
    // leaves cbeg = cend = start of every cell; returns the largest count this thread saw
    auto scan_cells = [&]() -> unsigned int {
        unsigned int cnt[16], tsum = 0u, cmax = 0u;
#pragma unroll
        for (int j = 0; j < 16; ++j) { cnt[j] = cend[tid * 16 + j]; tsum += cnt[j]; cmax = cnt[j] > cmax ? cnt[j] : cmax; }
        unsigned int incl = tsum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        unsigned int base = incl - tsum;
        for (int w = 0; w < wave; ++w) base += wsum[w];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            cbeg[tid * 16 + j] = (unsigned short)base;
            cend[tid * 16 + j] = base;
            base += cnt[j];
        }
        __syncthreads();
        return cmax;
    };
    // ---- counting sort of the source points by cell id = (iz*G + iy)*G + ix ---------------------------------------------------
    int cid[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int ix, iy, iz;
        cell_of(px[k], py[k], pz[k], ix, iy, iz);
        cid[k] = (iz * G + iy) * G + ix;
        atomicAdd(&cend[cid[k]], 1u);
    }
    __syncthreads();
    if (scan_cells() > 256u) crowded = 1;
    __syncthreads();
    if (crowded) {
        // ordered scan with early exit, one wave per centroid (64 centroids per wave, one after the other)
        const int s = tile * NT + tid;
        const float ax = qsrc[s * 3 + 0], ay = qsrc[s * 3 + 1], az = qsrc[s * 3 + 2];
        f32x4* pts2 = spt;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = tid + NT * k;
            float* q = reinterpret_cast<float*>(&pts2[((p >> 7) * 2) * 64 + (p & 63)]);
            const int hb = (p >> 6) & 1;
            q[hb] = px[k]; q[2 + hb] = py[k]; q[64 * 4 + hb] = pz[k]; q[64 * 4 + 2 + hb] = sumsq3(px[k], py[k], pz[k]);
        }
        __syncthreads();
        int* slots = reinterpret_cast<int*>(tab) + wave * kNSample;
        for (int j = 0; j < 64; ++j) {
            const float cx = __shfl(ax, j), cy = __shfl(ay, j), cz = __shfl(az, j);
            ball_scan_wave(pts2, slots, lane, cx, cy, cz, r2, out + ((size_t)c * S + tile * NT + wave * 64 + j) * kNSample);
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned int pos = atomicAdd(&cend[cid[k]], 1u);
        spt[pos] = f32x4{px[k], py[k], pz[k], sumsq3(px[k], py[k], pz[k])};
        sidx[pos] = (unsigned short)(tid + NT * k);
    }
#pragma unroll
    for (int w = 0; w < 32; ++w) tab[w * MS + tid] = 0u;
    __syncthreads();
    // ---- one thread per centroid: candidates of the 3 x 3 x-rows around its cell ------------------------------------------------
    const int s = tile * NT + tid;
    {
        const float ax = qsrc[s * 3 + 0], ay = qsrc[s * 3 + 1], az = qsrc[s * 3 + 2];
        const float a2 = sumsq3(ax, ay, az);
        int ix, iy, iz;
        cell_of(ax, ay, az, ix, iy, iz);
        const int x0 = max(ix - 1, 0), x1 = min(ix + 1, G - 1);
        for (int dz = -1; dz <= 1; ++dz) {
            const int z = iz + dz;
            if (z < 0 || z >= G) continue;
            for (int dy = -1; dy <= 1; ++dy) {
                const int y = iy + dy;
                if (y < 0 || y >= G) continue;
                const int row = (z * G + y) * G;
                const int kb = cbeg[row + x0], ke = (int)cend[row + x1];
                // four candidates per trip: the loads are independent, so one LDS round trip serves four tests
                for (int k = kb; k < ke; k += 4) {
                    f32x4 b[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) b[j] = spt[min(k + j, ke - 1)];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float d = sqdist_ref(ax, ay, az, a2, b[j][0], b[j][1], b[j][2], b[j][3]);
                        if (k + j < ke && !(d > r2)) {
                            const unsigned int i = sidx[k + j];
                            tab[(i >> 5) * MS + tid] |= 1u << (i & 31u);
                        }
                    }
                }
            }
        }
    }
    // ---- the 32 lowest set bits, in place (all words are in registers before the first slot is written) -------------------------
    {
        unsigned int m[32];
#pragma unroll
        for (int w = 0; w < 32; ++w) m[w] = tab[w * MS + tid];
        int cnt = 0, first = 0;
#pragma unroll
        for (int w = 0; w < 32; ++w) {
            unsigned int mm = m[w];
            while (mm != 0u && cnt < kNSample) {
                const int i = 32 * w + (__ffs((int)mm) - 1);
                if (cnt == 0) first = i;
                tab[cnt * MS + tid] = (unsigned int)i;
                ++cnt;
                mm &= mm - 1u;
            }
        }
        // cnt == 0 cannot happen for a centroid drawn from the cloud (the reference would index out of range)
        for (; cnt < kNSample; ++cnt) tab[cnt * MS + tid] = (unsigned int)first;
    }
    __syncthreads();
    int* dst = out + ((size_t)c * S + tile * NT) * kNSample;
    for (int e = tid; e < NT * kNSample; e += NT) dst[e] = (int)tab[(e & 31) * MS + (e >> 5)];
}

// ---------------------------------------------------------------------------------------------
// 3-NN: one thread per fine point keeps the three smallest distances (ties -> lower index) to the S
// coarse points staged in LDS, then the inverse-distance weights of pointnet2_utils.py:297-299.
// ---------------------------------------------------------------------------------------------
// fine_mod > 0: the fine clouds repeat with period fine_mod (level-0 clouds of a multi-step batch)
template <int S>
__global__ __launch_bounds__(256) void three_nn_kernel(const float* __restrict__ fine, const float* __restrict__ coarse, int N,
                                                       int* __restrict__ out_idx, float* __restrict__ out_w, int fine_mod) {
    __shared__ float4 pts[S];
    int c, tile;
    xcd_cloud_tile((N + 255) / 256, c, tile);
    const int tid = threadIdx.x;
    const float* src = coarse + (size_t)c * S * 3;
    for (int p = tid; p < S; p += 256) {
        const float x = src[p * 3 + 0], y = src[p * 3 + 1], z = src[p * 3 + 2];
        pts[p] = make_float4(x, y, z, sumsq3(x, y, z));
    }
    __syncthreads();
    const int n = tile * 256 + tid;
    if (n >= N) return;
    const float* fp = fine + ((size_t)(fine_mod > 0 ? c % fine_mod : c) * N + n) * 3;
    const float ax = fp[0], ay = fp[1], az = fp[2];
    const float a2 = sumsq3(ax, ay, az);
    float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY;
    int i0 = 0, i1 = 0, i2 = 0;
    // UN coarse points per trip: independent (broadcast) LDS reads and distance evaluations, ONE test against the current
    // third-best; the insert (in index order, strict <: ties keep the lower index) only runs when one of them qualifies
    // (the distances two at a time in packed fp32 -- planes in LDS, 3 instead of 6 VALU instructions per candidate, bit-identical --
    // measured no faster, 47 vs 44 us per step at fp2: with 64 scattered fine points per wave some lane qualifies in nearly every trip,
    // and the 16 conditional inserts are what the kernel spends its time on)
    constexpr int UN = S >= 16 ? 16 : 4;
    for (int p = 0; p < S; p += UN) {
        float d[UN];
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const float4 b = pts[p + j];
            d[j] = sqdist_ref(ax, ay, az, a2, b.x, b.y, b.z, b.w);
        }
        float m = fminf(fminf(d[0], d[1]), fminf(d[2], d[3]));
#pragma unroll
        for (int j = 4; j < UN; j += 4) m = fminf(m, fminf(fminf(d[j], d[j + 1]), fminf(d[j + 2], d[j + 3])));
        if (m < d2) {
#pragma unroll
            for (int j = 0; j < UN; ++j) {
                if (d[j] < d2) {
                    if (d[j] < d1) {
                        d2 = d1; i2 = i1;
                        if (d[j] < d0) { d1 = d0; i1 = i0; d0 = d[j]; i0 = p + j; }
                        else { d1 = d[j]; i1 = p + j; }
                    } else { d2 = d[j]; i2 = p + j; }
                }
            }
        }
    }
    const float r0 = __fdiv_rn(1.0f, __fadd_rn(d0, 1e-8f));
    const float r1 = __fdiv_rn(1.0f, __fadd_rn(d1, 1e-8f));
    const float r2 = __fdiv_rn(1.0f, __fadd_rn(d2, 1e-8f));
    const float norm = __fadd_rn(__fadd_rn(r0, r1), r2);
    const size_t o = ((size_t)c * N + n) * 3;
    out_idx[o + 0] = i0; out_idx[o + 1] = i1; out_idx[o + 2] = i2;
    out_w[o + 0] = __fdiv_rn(r0, norm); out_w[o + 1] = __fdiv_rn(r1, norm); out_w[o + 2] = __fdiv_rn(r2, norm);
}

// ---------------------------------------------------------------------------------------------
// Loop-invariant level-1 work (lsdm_sample_steps_invariant).  The finest FPS level draws all 1024 points of a cloud, so
// the level-1 centroids are the cloud itself in a step-dependent ORDER; the ball of a centroid (first 32 source indices
// within r, in source order) and therefore its sa1 feature depend on WHICH point it is, not on where FPS put it, and the
// three nearest level-1 points of a source point are the same points every step, at step-dependent positions.  So once per
// call: F1[p] = sa1 feature of point p as a centroid, NN8[n] = the eight nearest points of n (distance, source index).
// Per step this kernel (one work-group per cloud) only
//   * copies rows:          feat1[i]   = F1[idx1[i]]
//   * translates positions: pos[idx1[i]] = i; the three smallest (distance, pos) pairs among the eight candidates -- the tie
//     rule of the per-step scan (lowest level-1 position among equal distances) -- and their inverse-distance weights
// Exact: the same arithmetic on the same operands, only evaluated once.  If idx1 is NOT a permutation (duplicated points
// make FPS re-draw an index: degenerate clouds) the multiset of level-1 points differs from the cloud and the 3-NN of that
// cloud is redone exhaustively here against the actual level-1 points, exactly as three_nn_kernel does.
// ---------------------------------------------------------------------------------------------
// once per call: the 8 nearest points of every point of a cloud, (distance, source index) ascending -- enough candidates
// to re-apply the reference's tie rule (lowest LEVEL-1 POSITION among equal distances) once the step's order is known
__global__ __launch_bounds__(256) void nn8_kernel(const float* __restrict__ clouds, float* __restrict__ out_d, int* __restrict__ out_i) {
    constexpr int N = kPoints;
    __shared__ float4 pts[N];
    const int c = blockIdx.y, tid = threadIdx.x;
    const float* src = clouds + (size_t)c * N * 3;
    for (int p = tid; p < N; p += 256) {
        const float x = src[p * 3 + 0], y = src[p * 3 + 1], z = src[p * 3 + 2];
        pts[p] = make_float4(x, y, z, sumsq3(x, y, z));
    }
    __syncthreads();
    const int n = blockIdx.x * 256 + tid;
    const float4 a = pts[n];
    float d[8];
    int id[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { d[j] = INFINITY; id[j] = 0; }
    // 16 candidates per trip: independent (broadcast) LDS reads and distance evaluations, then ONE test of their minimum against the current
    // eighth-best; candidates that qualify are taken out smallest first (the lowest index among equal distances first) and inserted with strict
    // comparisons, so equal distances stay in increasing source index -- the list a one-by-one scan would build.  (Round 5: the one-by-one scan
    // ran its 8-deep insertion under a branch per candidate, 33 k VALU instructions per wave for 7 k of distance arithmetic: 570 -> 250 us per call.)
    constexpr int UN = 16;
    for (int p = 0; p < N; p += UN) {
        float dc[UN];
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const float4 b = pts[p + j];
            dc[j] = sqdist_ref(a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w);   // fine point = a, coarse point = b, as in three_nn_kernel
        }
        auto tree_min = [&]() {
            float m = fminf(fminf(dc[0], dc[1]), fminf(dc[2], dc[3]));
#pragma unroll
            for (int j = 4; j < UN; j += 4) m = fminf(m, fminf(fminf(dc[j], dc[j + 1]), fminf(dc[j + 2], dc[j + 3])));
            return m;
        };
        float m = tree_min();
        while (m < d[7]) {
            int jm = 0;
#pragma unroll
            for (int j = UN - 1; j >= 0; --j) jm = dc[j] == m ? j : jm;      // the first (lowest-index) candidate at the minimum
            d[7] = m; id[7] = p + jm;
#pragma unroll
            for (int j = 7; j > 0; --j) {
                if (d[j] < d[j - 1]) {
                    const float td = d[j]; d[j] = d[j - 1]; d[j - 1] = td;
                    const int ti = id[j]; id[j] = id[j - 1]; id[j - 1] = ti;
                }
            }
#pragma unroll
            for (int j = 0; j < UN; ++j) dc[j] = j == jm ? INFINITY : dc[j];
            m = tree_min();
        }
    }
    const size_t o = ((size_t)c * N + n) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) { out_d[o + j] = d[j]; out_i[o + j] = id[j]; }
}

__global__ __launch_bounds__(256) void l1_remap_kernel(const unsigned char* __restrict__ F1, int row_bytes, const int* __restrict__ idx1,
                                                       const float* __restrict__ nn8_d, const int* __restrict__ nn8_i,
                                                       const float* __restrict__ l1_xyz, const float* __restrict__ clouds,
                                                       unsigned char* __restrict__ feat1, int* __restrict__ nn_idx, float* __restrict__ nn_w, int inv_mod) {
    constexpr int N = kPoints, NT = 256;
    __shared__ unsigned short pos[N];
    __shared__ float4 pts[N];
    __shared__ int missing;
    const int c = blockIdx.x, tid = threadIdx.x;
    const int ci = inv_mod > 0 ? c % inv_mod : c;       // cloud of the per-call inputs (a launch may cover the clouds of several steps)
    const int* id = idx1 + (size_t)c * N;
    for (int e = tid; e < N; e += NT) pos[e] = 0xFFFFu;
    if (tid == 0) missing = 0;
    __syncthreads();
    for (int i = tid; i < N; i += NT) pos[id[i]] = (unsigned short)i;   // duplicates: any of the positions (identical points)
    __syncthreads();
    for (int e = tid; e < N; e += NT)
        if (pos[e] == 0xFFFFu) missing = 1;
    // feature rows (16-byte pieces; row_bytes = 256 fp32 / 128 bf16)
    const int ppr = row_bytes / 16;
    const unsigned char* fsrc = F1 + (size_t)ci * N * row_bytes;
    unsigned char* fdst = feat1 + (size_t)c * N * row_bytes;
    if (feat1) {   // (NULL: the consumers read F1 through idx1 themselves)
        for (int e = tid; e < N * ppr; e += NT) {
            const int i = e / ppr, q = e - i * ppr;
            *reinterpret_cast<f32x4*>(fdst + (size_t)i * row_bytes + q * 16) = *reinterpret_cast<const f32x4*>(fsrc + (size_t)id[i] * row_bytes + q * 16);
        }
    }
    // 3-NN of the fine points from their 8 candidates: smallest (distance, level-1 position); certified when the 8th
    // candidate is strictly farther than the chosen third (otherwise more ties than candidates: exhaustive path below)
    if (!missing) {
        for (int n = tid; n < N; n += NT) {
            const size_t o8 = ((size_t)ci * N + n) * 8;
            float d[8];
            int ps[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { d[j] = nn8_d[o8 + j]; ps[j] = (int)pos[nn8_i[o8 + j]]; }
            float rd[3];
            int ri[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                int best = 0;
#pragma unroll
                for (int j = 1; j < 8; ++j)
                    if (d[j] < d[best] || (d[j] == d[best] && ps[j] < ps[best])) best = j;
                rd[r] = d[best]; ri[r] = ps[best];
                d[best] = INFINITY;
            }
            if (!(nn8_d[o8 + 7] > rd[2])) missing = 1;
            const float r0 = __fdiv_rn(1.0f, __fadd_rn(rd[0], 1e-8f));
            const float r1 = __fdiv_rn(1.0f, __fadd_rn(rd[1], 1e-8f));
            const float r2 = __fdiv_rn(1.0f, __fadd_rn(rd[2], 1e-8f));
            const float norm = __fadd_rn(__fadd_rn(r0, r1), r2);
            const size_t o = ((size_t)c * N + n) * 3;
            nn_idx[o + 0] = ri[0]; nn_idx[o + 1] = ri[1]; nn_idx[o + 2] = ri[2];
            nn_w[o + 0] = __fdiv_rn(r0, norm); nn_w[o + 1] = __fdiv_rn(r1, norm); nn_w[o + 2] = __fdiv_rn(r2, norm);
        }
    }
    __syncthreads();
    if (!missing) return;
    // degenerate cloud: exhaustive 3-NN against the actual level-1 points (three_nn_kernel's scan and tie rule)
    const float* l1 = l1_xyz + (size_t)c * N * 3;
    for (int p = tid; p < N; p += NT) {
        const float x = l1[p * 3 + 0], y = l1[p * 3 + 1], z = l1[p * 3 + 2];
        pts[p] = make_float4(x, y, z, sumsq3(x, y, z));
    }
    __syncthreads();
    const float* cl = clouds + (size_t)ci * N * 3;
    for (int n = tid; n < N; n += NT) {
        const float ax = cl[n * 3 + 0], ay = cl[n * 3 + 1], az = cl[n * 3 + 2];
        const float a2 = sumsq3(ax, ay, az);
        float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY;
        int i0 = 0, i1 = 0, i2 = 0;
        for (int p = 0; p < N; ++p) {
            const float4 b = pts[p];
            const float d = sqdist_ref(ax, ay, az, a2, b.x, b.y, b.z, b.w);
            if (d < d2) {
                if (d < d1) {
                    d2 = d1; i2 = i1;
                    if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = p; }
                    else { d1 = d; i1 = p; }
                } else { d2 = d; i2 = p; }
            }
        }
        const float r0 = __fdiv_rn(1.0f, __fadd_rn(d0, 1e-8f));
        const float r1 = __fdiv_rn(1.0f, __fadd_rn(d1, 1e-8f));
        const float r2 = __fdiv_rn(1.0f, __fadd_rn(d2, 1e-8f));
        const float norm = __fadd_rn(__fadd_rn(r0, r1), r2);
        const size_t o = ((size_t)c * N + n) * 3;
        nn_idx[o + 0] = i0; nn_idx[o + 1] = i1; nn_idx[o + 2] = i2;
        nn_w[o + 0] = __fdiv_rn(r0, norm); nn_w[o + 1] = __fdiv_rn(r1, norm); nn_w[o + 2] = __fdiv_rn(r2, norm);
    }
}

void launch_nn8(int nc, const float* clouds, float* out_d, int* out_i, hipStream_t st) {
    hipLaunchKernelGGL(nn8_kernel, dim3(kPoints / 256, nc), dim3(256), 0, st, clouds, out_d, out_i);
}

void launch_l1_remap(int nc, const void* F1, int row_bytes, const int* idx1, const float* nn8_d, const int* nn8_i, const float* l1_xyz,
                     const float* clouds, void* feat1, int* nn_idx, float* nn_w, hipStream_t st, int inv_mod) {
    hipLaunchKernelGGL(l1_remap_kernel, dim3(nc), dim3(256), 0, st, static_cast<const unsigned char*>(F1), row_bytes, idx1, nn8_d, nn8_i,
                       l1_xyz, clouds, static_cast<unsigned char*>(feat1), nn_idx, nn_w, inv_mod);
}

// ---------------------------------------------------------------------------------------------
// Memo of sa2 in the loop-invariant schedule (api.hip: memo_ready).  sa2's source set is the WHOLE cloud (level 1 = all 1024
// points in a step-dependent order) and its ball keeps the first 32 points, in level-1 order, with NOT(d > r^2).  When a point has
// FEWER than 32 points within r, its ball is all of them whatever the order (the padding repeats one of them) and the max over the
// ball -- the stage's output for that centroid -- depends only on WHICH point the centroid is: it is evaluated once per call for
// every such point (F2) and copied per step.  Only centroids with >= 32 points within r ("crowded": the order decides which 32) go
// through the stage each step.  Exact: per row the same arithmetic on the same operands; max is order-free.
// (Without the copies -- the consumers of the stage's output (sa3's first-layer launch, sa3's U / V gather, fp3's p1 rows) reading the
// per-call rows through per-step row maps, and that launch run for the crowded rows only -- was measured: sa3 344 vs 331 us per step, the
// step 1.93 vs 1.94 ms: the gather from a 300 MB per-call table loses what the skipped copies and rows save.  Not kept.)
//
// memo_src_kernel (per call): ball2 = the r-ball of every source point as a centroid, first 32 in SOURCE order, padded with its
// first entry (launch_ball_query).  Entries are strictly increasing, so a padded last slot == the first entry  <=>  count <= 31.
__global__ __launch_bounds__(256) void memo_src_kernel(const int* __restrict__ ball2, long long total, unsigned char* __restrict__ unc,
                                                       int* __restrict__ list, int* __restrict__ count) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool u = false;
    if (e < total) {
        u = ball2[e * kNSample + kNSample - 1] == ball2[e * kNSample];
        unc[e] = u ? 1 : 0;
    }
    const unsigned long long m = __ballot(u);
    int base = 0;
    if (lane == 0 && m) base = atomicAdd(count, __popcll(m));
    base = __shfl(base, 0);
    if (u) list[base + __popcll(m & ((1ull << lane) - 1ull))] = (int)e;
}

// memo_cent_kernel (per group; one work-group per (step, cloud), one thread per level-2 centroid): uncrowded centroids copy their F2
// row, crowded ones are appended to the work list of the stage.  A cloud whose level-1 index list is not a permutation (duplicated
// points made FPS re-draw an index: the level-1 multiset is then not the cloud) sends all its centroids to the list.
__global__ __launch_bounds__(256) void memo_cent_kernel(const int* __restrict__ idx0, const int* __restrict__ idx1, const unsigned char* __restrict__ unc,
                                                        const float* __restrict__ F2, int nc1, float* __restrict__ out, int* __restrict__ list,
                                                        int* __restrict__ count, unsigned char* __restrict__ crowd) {
    constexpr int N = kPoints, S = 256, C = 128;
    __shared__ unsigned char seen[N];
    __shared__ int missing;
    __shared__ int srow[S];
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int ci = c % nc1;
    const int* i0 = idx0 + (size_t)c * N;
    for (int e = tid; e < N; e += 256) seen[e] = 0;
    if (tid == 0) missing = 0;
    __syncthreads();
    for (int e = tid; e < N; e += 256) seen[i0[e]] = 1;
    __syncthreads();
    for (int e = tid; e < N; e += 256)
        if (!seen[e]) missing = 1;
    __syncthreads();
    const int src = i0[idx1[(size_t)c * S + tid]];
    const bool crowded = missing || !unc[(size_t)ci * N + src];
    srow[tid] = crowded ? -1 : src;
    crowd[(size_t)c * S + tid] = crowded ? 1 : 0;        // (the stage's ball query then runs for these centroids only)
    const unsigned long long m = __ballot(crowded);
    int base = 0;
    if (lane == 0 && m) base = atomicAdd(count, __popcll(m));
    base = __shfl(base, 0);
    if (crowded) list[base + __popcll(m & ((1ull << lane) - 1ull))] = c * S + tid;
    __syncthreads();
    // copies: 32 threads x 16 bytes per row, eight rows per pass
    const int sub = tid & 31;
    for (int r = tid >> 5; r < S; r += 8) {
        const int sr = srow[r];
        if (sr < 0) continue;
        const f32x4 v = *reinterpret_cast<const f32x4*>(F2 + ((size_t)ci * N + sr) * C + 4 * sub);
        *reinterpret_cast<f32x4*>(out + ((size_t)c * S + r) * C + 4 * sub) = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Item lists of sa3 / sa4 (sa_p_kernel / sa_w_kernel with PAIR, encoder_u.hip).  query_ball_point (pointnet2_utils.py:84-104; the padding :101-104) pads a group
// with fewer than 32 points in the ball with repeats of its FIRST entry; under the stage's max over the group the repeats change nothing.
// A group's d distinct entries come first (in increasing order), so its first 4 ceil(d / 4) rows already hold all of them: two groups whose
// row needs add up to at most 32 share ONE 32-row block of the stack -- the larger one ("A", k = ceil(d / 4) four-row groups) in rows
// 0 .. 4k-1, the other ("B") in rows 4k .. 31, each with its own V row, its own maximum and its own output row; every other group keeps a
// block to itself.  The split is a multiple of four rows because an accumulator register quartet of a half-wave holds four consecutive
// rows.  On the bench clouds (half the groups are full, the others spread over all sizes) 0.84-0.86 blocks per group, per row the same
// arithmetic: bit-identical results.
//
// A SEGMENT is one step's launch (T1 = clouds x S centroids, rows and output numbered within it); a launch covers the segments of a group.
// Work-group = chunk of 256 consecutive centroids of a segment, pairs are formed inside a chunk (neighbours in index order, usually of few
// clouds): size class = ceil(d / 4) (1 .. 8), then the two-pointer matching that maximises the number of pairs -- the largest class that
// still has a partner takes the smallest class left -- on the classes' COUNTS (plan_runs: at most 14 runs of (class, class, pairs)), every
// group finding its run from its rank within its class.  Two passes, no atomics, so that the list's ORDER is a function of the input too:
// pass 0 -> blocks per chunk; pass 1 -> every chunk adds up the items of the chunks before it and writes its items, [pairs][singles in
// index order].
//   rows  (T1 x 32)  U row of every block row, numbered over the segment (cloud x NSRC + ball entry)
//   meta  (T1) int4  {V row of A, V row of B, output row of A | k << 24, output row of B}; k = 8: an unpaired item (B = A)
//   count (1)        items of the segment
namespace sap {
struct Run { int big, small, m, off_big, off_small, slot0; };
constexpr int MAXRUN = 16;
// cnt[1 .. 7] -> runs; returns the number of pairs.  Ranks of a class are consumed from the front, whichever side of a pair they end up on.
__device__ inline int plan_runs(const int* cnt, Run* runs, int* nruns) {
    int rem[9], off[9];
    for (int c = 0; c < 9; ++c) { rem[c] = c >= 1 && c <= 7 ? cnt[c] : 0; off[c] = 0; }
    int np = 0, nr = 0;
    for (int big = 7; big >= 1; --big) {
        while (rem[big] > 0) {
            int small = 1;
            while (rem[small] == 0) ++small;                 // (the smallest class left: <= big)
            if (small + big > 8) break;
            const int m = small == big ? rem[big] / 2 : (rem[big] < rem[small] ? rem[big] : rem[small]);
            if (m == 0) break;
            if (runs) runs[nr] = Run{big, small, m, off[big], off[small], np};
            ++nr;
            if (small == big) { off[big] += 2 * m; rem[big] -= 2 * m; }
            else { off[big] += m; rem[big] -= m; off[small] += m; rem[small] -= m; }
            np += m;
        }
    }
    if (nruns) *nruns = nr;
    return np;
}
}  // namespace sap

// only (optional, one byte per centroid): the groups with a zero get no block (sa2's memo pass: the crowded source points are not evaluated);
// cent_idx == nullptr: a centroid is source point number (its index within the cloud), as in that pass.
// One launch plans up to four levels (a single step's four SA stages: two launches instead of eight).
template <bool WRITE>
__global__ __launch_bounds__(256) void sa_pair_kernel(PairLevels P) {
    __shared__ int wcnt[4][9], wsingle[4], wbefore[4], s_np, s_nruns;
    __shared__ sap::Run runs[sap::MAXRUN];
    int lv = 0;
    for (int i = 1; i < P.n; ++i)
        if ((int)blockIdx.x >= P.l[i].first_block) lv = i;
    const PairLevel& a = P.l[lv];
    const int* __restrict__ ball = a.ball; const int* __restrict__ cent_idx = a.cent_idx;
    const int T1 = a.T1, S = a.S, NSRC = a.NSRC, CPS = a.CPS;
    int* __restrict__ chunk_np = a.chunk_np; int* __restrict__ rows = a.rows; int4* __restrict__ meta = a.meta; int* __restrict__ count = a.count;
    const unsigned char* __restrict__ only = a.only;
    const int bid = (int)blockIdx.x - a.first_block;
    const int seg = bid / CPS, chunk = bid % CPS;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int e = chunk * 256 + tid;
    const bool valid = e < T1 && (!only || only[(size_t)seg * T1 + e]);
    const size_t ge = (size_t)seg * T1 + (e < T1 ? e : 0);
    const int4* src = reinterpret_cast<const int4*>(ball + ge * kNSample);
    int4 b[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) b[k] = src[k];
    int d = 1;
#pragma unroll
    for (int k = 0; k < 8; ++k) d += (b[k].x != b[0].x) + (b[k].y != b[0].x) + (b[k].z != b[0].x) + (b[k].w != b[0].x);
    const int cls = valid ? (d + 3) >> 2 : 0;                // 1 .. 8 (0: past the end of the segment)
    int rank = 0;                                            // groups of the same class before this one in the chunk
    for (int c = 1; c <= 7; ++c) {
        const unsigned long long m = __ballot(cls == c);
        if (lane == 0) wcnt[wave][c] = __popcll(m);
        if (cls == c) rank = __popcll(m & lt);
    }
    if (WRITE) {
        // blocks of the chunks before this one
        int before = 0;
        for (int k = tid; k < chunk; k += 256) before += chunk_np[seg * CPS + k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o);
        if (lane == 0) wbefore[wave] = before;
    }
    __syncthreads();
    if (tid == 0) {
        int cnt[9] = {};
        for (int c = 1; c <= 7; ++c) cnt[c] = wcnt[0][c] + wcnt[1][c] + wcnt[2][c] + wcnt[3][c];
        int nr = 0;
        s_np = sap::plan_runs(cnt, WRITE ? runs : nullptr, &nr);
        s_nruns = nr;
    }
    const int nvalid = __syncthreads_count(valid);           // (also publishes s_np and the runs)
    if (!WRITE) {
        if (tid == 0) chunk_np[bid] = nvalid - s_np;         // the chunk's blocks
        return;
    }
    for (int w = 0; w < wave; ++w) rank += cls >= 1 && cls <= 7 ? wcnt[w][cls] : 0;
    const int npairs = s_np, nruns = s_nruns;
    const int base = wbefore[0] + wbefore[1] + wbefore[2] + wbefore[3];
    // this group's run: side 0 = A (rows 0 .. 4k-1), side 1 = B
    int slot = -1, side = 0, kA = 8;
    for (int r = 0; r < nruns; ++r) {
        const sap::Run q = runs[r];
        if (q.big == q.small) {
            if (cls == q.big && rank >= q.off_big && rank < q.off_big + 2 * q.m) { slot = q.slot0 + ((rank - q.off_big) >> 1); side = (rank - q.off_big) & 1; kA = q.big; }
        } else if (cls == q.big && rank >= q.off_big && rank < q.off_big + q.m) { slot = q.slot0 + rank - q.off_big; side = 0; kA = q.big; }
        else if (cls == q.small && rank >= q.off_small && rank < q.off_small + q.m) { slot = q.slot0 + rank - q.off_small; side = 1; kA = q.big; }
    }
    const bool single = valid && slot < 0;
    const unsigned long long ms = __ballot(single);
    if (lane == 0) wsingle[wave] = __popcll(ms);
    __syncthreads();
    if (chunk == CPS - 1 && tid == 0) count[seg] = base + nvalid - npairs;
    if (!valid) return;
    if (single) {
        slot = npairs + __popcll(ms & lt);
        for (int w = 0; w < wave; ++w) slot += wsingle[w];
        kA = 8;
    }
    const size_t gi = (size_t)seg * T1 + base + slot;
    const int r0 = (e / S) * NSRC;                           // first U / V row of the centroid's cloud
    const int vrow = r0 + (cent_idx ? cent_idx[ge] : e % S);
    const int n4 = single ? 8 : (side ? 8 - kA : kA);        // quartets of rows this group fills
    int4* dst = reinterpret_cast<int4*>(rows + gi * kNSample + (side ? 4 * kA : 0));
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (k < n4) dst[k] = make_int4(b[k].x + r0, b[k].y + r0, b[k].z + r0, b[k].w + r0);
    }
    int* mi = reinterpret_cast<int*>(meta + gi);
    if (single) meta[gi] = make_int4(vrow, vrow, e | (8 << 24), e);
    else if (side == 0) { mi[0] = vrow; mi[2] = e | (kA << 24); }
    else { mi[1] = vrow; mi[3] = e; }
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
void launch_fps_chain(const float* clouds, const int64_t* starts, int num_clouds, int nc_step, int* const idx[4], float* const xyz[4],
                      hipStream_t st) {
    hipLaunchKernelGGL(fps_chain_reg_kernel, dim3(num_clouds), dim3(64), 0, st, clouds, starts, nc_step, idx[0], idx[1], idx[2], idx[3],
                       xyz[0], xyz[1], xyz[2], xyz[3]);
}

void launch_ball_query(const float* xyz, const float* new_xyz, int num_clouds, int N, int S, float r2, int* out, int src_mod,
                       hipStream_t st, const unsigned char* only, bool scan) {
    dim3 grid((S + 15) / 16 * num_clouds), block(256);
    if (N == 1024 && S == 1024 && !scan) {
        // finest level (r = 0.1, every point is a centroid): grid-pruned candidates, one thread per centroid.  Measured on
        // MI355X: 0.19 -> 0.10 ms per step; at the next level (r = 0.2, 8x the candidates) the ordered scan stays faster.
        hipLaunchKernelGGL(ball_query_grid_kernel, dim3(S / 256 * num_clouds), block, 0, st, xyz, new_xyz, S, r2, sqrtf(r2) * 1.02f, out, src_mod);
        return;
    }
    switch (N) {
        case 1024: hipLaunchKernelGGL(ball_query_kernel<1024>, grid, block, 0, st, xyz, new_xyz, S, r2, out, src_mod, only); break;
        case 256: hipLaunchKernelGGL(ball_query_kernel<256>, grid, block, 0, st, xyz, new_xyz, S, r2, out, src_mod, only); break;
        case 64: hipLaunchKernelGGL(ball_query_kernel<64>, grid, block, 0, st, xyz, new_xyz, S, r2, out, src_mod, only); break;
    }
}

void launch_memo_src(const int* ball2, long long total, unsigned char* unc, int* list, int* count, hipStream_t st) {
    hipLaunchKernelGGL(memo_src_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ball2, total, unc, list, count);
}
void launch_memo_cent(int clouds, const int* idx0, const int* idx1, const unsigned char* unc, const float* F2, int nc1, float* out, int* list, int* count,
                      unsigned char* crowd, hipStream_t st) {
    hipLaunchKernelGGL(memo_cent_kernel, dim3(clouds), dim3(256), 0, st, idx0, idx1, unc, F2, nc1, out, list, count, crowd);
}

int sa_pair_chunks(int T1) { return (T1 + 255) / 256; }
void launch_sa_pair_plans(const PairLevel* levels, int n, hipStream_t st) {
    PairLevels P;
    P.n = 0;
    int blocks = 0;
    for (int i = 0; i < n && P.n < 4; ++i) {
        PairLevel a = levels[i];
        a.CPS = sa_pair_chunks(a.T1);
        a.first_block = blocks;
        blocks += a.segs * a.CPS;
        P.l[P.n++] = a;
    }
    if (!blocks) return;
    hipLaunchKernelGGL(sa_pair_kernel<false>, dim3(blocks), dim3(256), 0, st, P);
    hipLaunchKernelGGL(sa_pair_kernel<true>, dim3(blocks), dim3(256), 0, st, P);
}
void launch_sa_pair_plan(int segs, int T1, int S, int NSRC, const int* ball, const int* cent_idx, int* chunk_np, int* rows, int4* meta, int* count,
                         hipStream_t st, const unsigned char* only) {
    const PairLevel a{ball, cent_idx, T1, S, NSRC, 0, segs, chunk_np, rows, meta, count, only, 0};
    launch_sa_pair_plans(&a, 1, st);
}

void launch_three_nn(const float* fine, const float* coarse, int num_clouds, int N, int S, int* out_idx, float* out_w, int fine_mod,
                     hipStream_t st) {
    dim3 grid((N + 255) / 256 * num_clouds), block(256);
    switch (S) {
        case 1024: hipLaunchKernelGGL(three_nn_kernel<1024>, grid, block, 0, st, fine, coarse, N, out_idx, out_w, fine_mod); break;
        case 256: hipLaunchKernelGGL(three_nn_kernel<256>, grid, block, 0, st, fine, coarse, N, out_idx, out_w, fine_mod); break;
        case 64: hipLaunchKernelGGL(three_nn_kernel<64>, grid, block, 0, st, fine, coarse, N, out_idx, out_w, fine_mod); break;
        case 16: hipLaunchKernelGGL(three_nn_kernel<16>, grid, block, 0, st, fine, coarse, N, out_idx, out_w, fine_mod); break;
    }
}

}  // namespace lsdm
