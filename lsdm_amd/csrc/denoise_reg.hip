// The x/t-dependent tail of a denoising step as a REGISTER-CHAINED stack on the split-bf16 MFMA core, gfx950.
//
// Same arithmetic contract as denoise_bf.hip (guidance g with the Q1/Q2 scrambles, in-place x += g, the per-point denoiser of
// model/sdm.py:196-217 through InputProcess / OutputProcess of model/diffusion_utils.py:51-78,98-122, posterior / DDIM update of
// gaussian_diffusion.py:258-280,545-561,790-811), different execution shape:
//
//   * denoise_h_kernel keeps a 32-point tile's activations in LDS and runs every layer as "fetch weights from L2 -> MFMAs -> epilogue
//     to LDS -> work-group barrier": ten dependent phases of a few hundred cycles of matrix work each, two work-groups per CU --
//     the kernel is latency-bound (MFMA pipe 18 % busy).
//   * Here a WAVE owns 32 points of one chain and carries their activations through the whole stack in registers (the
//     accumulator -> operand identity of encoder_reg.hip: in the transposed product D[channel][point] the accumulators 8 s2 .. 8 s2 + 7
//     of channel block nb ARE the lane's k-slots of k-step 2 nb + s2 of the next layer; the host packs the weight fragments in that
//     k order).  Weights go through LDS as PANELS of <= 32 KB shared by the 8 waves of the work-group (256 points per panel
//     pass) in a ring of four slots, copied global -> LDS by the LDS-DMA path THREE panels ahead of their use (a panel step is
//     only ~1 us of matrix work at one 32-point block per wave; measured against two 48 KB buffers filled one step ahead the
//     ring changes nothing -- 61 us per hoisted step either way: the stage is bound by its activations' VALU work, see
//     DESIGN.md); one barrier per panel.  Per step: [S x3 | pose2 | cex0 x3 | cex2 x4 | fin0];
//     pose_final.2 (3 rows zero-padded to one channel block) stays resident.
//   * The timestep half of combine_extraction's pre-activation comes from ctext_kernel(kb0 = 0) (`ctime`; one tile per step when
//     all chains share the timestep, one per (step, chain) otherwise), the text half from `ctext`: emb = GELU(ctime + ctext + b) is
//     formed straight into operand fragments.  S = W_cex0[:, 128:] . emb + b is shared by the two denoiser passes of a step's last
//     forward; the first pass consumes its registers in place, so on the last step it is parked in a global scratch tile
//     (24 KB per wave) and read back for the guiding-point pass.
//   * nsteps consecutive steps run in one launch (hoisted-conditioning groups): x never leaves the wave's LDS slot between steps.
#include "bf16_core.hpp"
#include "kernels.hpp"

namespace lsdm {

namespace {

constexpr int NW = 8, NT = NW * 64;

template <int NP>
struct RL {   // fragment (1 KiB = 512 bf16) offsets of the blob, see api.hip::build_denoise_reg_blob
    static constexpr int S_KS = 8, S_NB = 6;      // combination_extraction.0[:, 128:256] (emb half), first-layer k order
    static constexpr int P2_KS = 4, P2_NB = 4;    // pose_embedding.2 64 -> 128, first-layer k order
    static constexpr int C0_KS = 8, C0_NB = 6;    // combination_extraction.0[:, 0:128] (pose half), accumulator k order
    static constexpr int C2_KS = 12, C2_NB = 4;   // combination_extraction.2 192 -> 128
    static constexpr int F0_KS = 8, F0_NB = 2;    // pose_final.0 128 -> 64
    static constexpr int F2_KS = 4;               // pose_final.2 64 -> 3 (one zero-padded block)
    static constexpr int OFF_S = 0;
    static constexpr int OFF_P2 = OFF_S + S_NB * S_KS * NP;
    static constexpr int OFF_C0 = OFF_P2 + P2_NB * P2_KS * NP;
    static constexpr int OFF_C2 = OFF_C0 + C0_NB * C0_KS * NP;
    static constexpr int OFF_F0 = OFF_C2 + C2_NB * C2_KS * NP;
    static constexpr int OFF_F2 = OFF_F0 + F0_NB * F0_KS * NP;
    static constexpr int TOTAL = OFF_F2 + F2_KS * NP;
    // Panels (<= 32 KB) of a step, in the order they are consumed:
    //   0..2  S blocks {0,1} {2,3} {4,5} | 3 pose2 | 4..6 cex0 blocks {0,1} {2,3} {4,5} | 7..10 cex2 blocks 0..3 | 11 fin0
    // the guiding-point pass of a step's last forward repeats 3..11
    static constexpr int NPAN = 12, FIRST_PASS = 3;
    static constexpr int PANEL = 16 * NP;              // largest panel (fragments)
    static constexpr int MIN_PER_WAVE = 12 * NP / 8;   // LDS-DMA instructions EVERY wave issues for the smallest panel
    __device__ static int poff(int id) {
        return id < 3 ? OFF_S + id * 16 * NP : id == 3 ? OFF_P2 : id < 7 ? OFF_C0 + (id - 4) * 16 * NP : id < 11 ? OFF_C2 + (id - 7) * 12 * NP : OFF_F0;
    }
    __device__ static int pn(int id) { return (id >= 7 && id < 11) ? 12 * NP : 16 * NP; }
};
// bias block (floats), same order as the host builds it
constexpr int B_CEX0 = 0, B_POSE2 = 192, B_CEX2 = 320, B_FIN0 = 448, B_COMB = 512, B_POSE0 = 640, B_FIN2 = 896, B_TOTAL = 900;

__device__ __forceinline__ float scrambled_r(const float* __restrict__ pcd_b, const float* __restrict__ w_b, int op, int pp, int cp) {
    const int f = (op * kPoints + pp) * 3 + cp;
    const int j = f / kSlots, o = f - j * kSlots;
    return pcd_b[o * (kPoints * 3) + j] * w_b[o];
}

// LDS traffic of ONE wave with itself (write by some lanes, read by others): in-order inside the wave, the compiler only has to
// keep the program order
__device__ __forceinline__ void wave_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

template <int NP>
__device__ __forceinline__ bf16x8 wfrag(const __bf16* base, int KS, int nb, int s, int plane, int lane) {
    return *reinterpret_cast<const bf16x8*>(base + ((size_t)((nb * KS + s) * NP + plane) * 64 + lane) * 8);
}

// acc += W(block nb of a panel whose first block sits at W) (x) X(registers); transposed product D[channel][point]
template <int KS, int NP>
__device__ __forceinline__ void mma_t(f32x16& acc, const __bf16* W, int nb, const bf16x8 (&xh)[KS], const bf16x8 (&xl)[KS], int lane) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const bf16x8 wh = wfrag<NP>(W, KS, nb, s, 0, lane);
        if constexpr (NP == 2) {
            const bf16x8 wlo = wfrag<NP>(W, KS, nb, s, 1, lane);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl[s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo, xh[s], acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh[s], acc, 0, 0, 0);
        if (s % 3 == 2) __builtin_amdgcn_sched_barrier(0);   // bound the LDS weight fragments in flight (VGPRs)
    }
}

__device__ __forceinline__ void acc_from_bias(f32x16& acc, const float* bias32, int h) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 bq = *reinterpret_cast<const f32x4*>(bias32 + 8 * q + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[4 * q + j] = bq[j];
    }
}

// activation + split of a transposed block into the next layer's k-steps 2 nb, 2 nb + 1
template <int ACT, int NP>
__device__ __forceinline__ void act_split(const f32x16& acc, bf16x8& h0, bf16x8& l0, bf16x8& h1, bf16x8& l1) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = apply_act_fast<ACT>(acc[8 * s2 + j]);
        if (s2 == 0) split8_h<NP>(v, h0, l0); else split8_h<NP>(v, h1, l1);
    }
}

}  // namespace

template <int NP>
__global__ __launch_bounds__(NT) void denoise_r_kernel(const __bf16* __restrict__ blob_w, const float* __restrict__ blob_b, Lin pwt, StepCoefs sc,
                                                       int mode, float eta, int B, int nsteps, int ctime_per_sample, float* __restrict__ x,
                                                       const int64_t* __restrict__ t, const float* __restrict__ mask,
                                                       const float* __restrict__ pcd, const float* __restrict__ w,
                                                       const float* __restrict__ o12, const float* __restrict__ hm,
                                                       const float* __restrict__ ctext, const float* __restrict__ ctime,
                                                       const float* __restrict__ noise, float* __restrict__ stash,
                                                       float* __restrict__ out_x0, float* __restrict__ out_guiding,
                                                       float* __restrict__ out_sample, long long pcd_step, long long o12_step) {
    using L = RL<NP>;
    constexpr int NSLOT = 4;                                   // panel ring: the panel consumed now + three in flight
    __shared__ __attribute__((aligned(16))) __bf16 wl[NSLOT][L::PANEL * 512];
    __shared__ __attribute__((aligned(16))) __bf16 wf2[L::F2_KS * NP * 512];
    __shared__ __attribute__((aligned(16))) float bl[B_TOTAL + 4];
    __shared__ float fsm[NW][4 * 96 + 32];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int p = lane & 31, h = lane >> 5;
    for (int e = tid; e < L::F2_KS * NP * 64; e += NT)
        *reinterpret_cast<bf16x8*>(wf2 + (size_t)e * 8) = gload<bf16x8>(blob_w + (size_t)L::OFF_F2 * 512 + (size_t)e * 8);
    for (int e = tid; e < B_TOTAL; e += NT) bl[e] = blob_b[e];
    // (explicitly LDS-typed: with the guidance inside the step loop the compiler no longer proved these generic pointers to be LDS
    // and its flat-address fallback does not assemble -- "Illegal instruction detected: V_CMP_NE_U32 0, $src_shared_base")
    using LF = __attribute__((address_space(3))) float;
    LF* zt = (LF*)&fsm[wave][0];      // mutated x of the wave's 32 points (x 3)
    LF* gt = zt + 96;                 // guidance / guiding points
    LF* xs = gt + 96;                 // current sample between steps
    LF* res = xs + 96;                // denoiser output
    LF* oterm = res + 96;             // W15[c, 3:15] . o[b, o', :]   (27 values)

    const int total = B * (kPoints / 32);                       // 32-point blocks
    const int rounds = (total + NW - 1) / NW;
    const int rchunk = (rounds + (int)gridDim.x - 1) / (int)gridDim.x;
    const int rbeg = xcd_remap((int)blockIdx.x, (int)gridDim.x) * rchunk;
    const int rend = rbeg + rchunk < rounds ? rbeg + rchunk : rounds;
    const bool chain = out_sample == nullptr && mode != MODE_FORWARD;

    // ---- weight panels: the consumer counts panels (pc, slot = pc % NSLOT); the producer cursor walks the same sequence -- steps
    // 0 .. nsteps-1 of every round, panels 0 .. 11 of a step, 3 .. 11 again after the last step -- three panels ahead.  All of it
    // is wave-uniform scalar state.
    int pc = 0;                                                         // panels consumed
    int q_id = 0, q_step = 0, q_round = rbeg, q_pass2 = 0, q_cnt = 0;   // producer cursor; q_cnt = panels requested
    auto request_next = [&]() {
        if (q_round >= rend) return;
        const __bf16* src = opaque_ptr(blob_w) + (size_t)L::poff(q_id) * 512;
        const int n = L::pn(q_id);
        __bf16* dst = &wl[q_cnt % NSLOT][0];
        for (int i = wave; i < n; i += NW)
            lds_dma16(src + (size_t)i * 512 + lane * 8, __builtin_amdgcn_readfirstlane(lds_addr(dst + (size_t)i * 512)));
        ++q_cnt;
        if (q_id < L::NPAN - 1) { ++q_id; return; }
        if (!q_pass2 && q_step == nsteps - 1 && out_guiding != nullptr) { q_pass2 = 1; q_id = L::FIRST_PASS; return; }
        q_pass2 = 0; q_id = 0;
        if (++q_step >= nsteps) { q_step = 0; ++q_round; }
    };
    // the panel consumed now (the request for the one three ahead goes into the slot the previous barrier released)
    // (lzo: the opaque zero itself, for the bias reads of the same phase -- as an int, not as a difference of LDS pointers, which
    // the compiler takes through generic addresses)
    auto panel_begin = [&](int& lzo) -> const __bf16* {
        request_next();
        int lz = 0;
        asm volatile("" : "+v"(lz));      // opaque zero: the (step-invariant) LDS reads stay inside the loops
        lzo = lz;
        return &wl[pc % NSLOT][0] + lz;
    };
    // the NEXT panel must have landed: everything but the copies of the two panels behind it may still be in flight (loads return
    // in order; other loads issued since only make the wait stricter); at the tail of the sequence wait for everything
    auto panel_end = [&]() {
        if (q_cnt - pc >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * L::MIN_PER_WAVE) : "memory");
        else lds_dma_wait();
        __syncthreads();
        ++pc;
    };
    request_next();
    request_next();
    request_next();
    lds_dma_wait();
    __syncthreads();

    for (int rd = rbeg; rd < rend; ++rd) {
        const int blk_raw = rd * NW + wave;
        const bool live = blk_raw < total;
        const int blk = live ? blk_raw : total - 1;
        const int b = blk / (kPoints / 32), n0 = (blk - b * (kPoints / 32)) * 32;
        const float* pcd_b = pcd + (size_t)b * kSlots * kPoints * 3;
        const float* w_b = w + (size_t)b * kSlots;
        const size_t g0 = ((size_t)b * kPoints + n0) * 3;       // first element of the wave's (32, 3) tile
        int lr = lane;
        asm volatile("" : "+v"(lr));       // (see the step loop: per-lane element addresses are formed at their use)
#pragma unroll 1
        for (int i = lr; i < 96; i += 64) xs[i] = x[g0 + i];
        wave_sync();
        // the human branch's rows of the tile (per call): element lane and, for the lower half-wave, 64 + lane
        const float hm0 = hm[g0 + lr], hm1 = lr < 32 ? hm[g0 + 64 + lr] : 0.0f;

        // one denoiser pass on the (32, 3) tile zsrc (LDS) -> res; consumes S in place (panels 3 .. 11)
        auto pass = [&](const LF* zsrc, f32x16 (&S)[6]) {
            // pose_embedding.0: 3 -> 64 sigmoid (K = 3: VALU) straight into the K = 64 operand: k-slot (s, h, j) = channel 16 s + 8 h + j
            bf16x8 ah[4], al[4];
            {
                int lz = 0;
                asm volatile("" : "+v"(lz));
                const float z0 = zsrc[p * 3], z1 = zsrc[p * 3 + 1], z2 = zsrc[p * 3 + 2];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const f32x4 wb = *reinterpret_cast<const f32x4*>(bl + lz + B_POSE0 + 4 * (16 * s + 8 * h + j));
                        v[j] = sigmoid_fast(fmaf(z2, wb[2], fmaf(z1, wb[1], z0 * wb[0])) + wb[3]);
                    }
                    split8_h<NP>(v, ah[s], al[s]);
                }
            }
            // ---- panel 3: pose2 64 -> 128 sigmoid ------------------------------------------------------------------------------
            bf16x8 ph[8], pl[8];
            {
                int lzp;
                const __bf16* W = panel_begin(lzp);
                const float* bz = bl + lzp;
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    f32x16 acc;
                    acc_from_bias(acc, bz + B_POSE2 + 32 * nb, h);
                    mma_t<4, NP>(acc, W, nb, ah, al, lane);
                    act_split<ACT_SIGMOID, NP>(acc, ph[2 * nb], pl[2 * nb], ph[2 * nb + 1], pl[2 * nb + 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                panel_end();
            }
            // ---- panels 4..6: cex0 [pose || emb] 256 -> 192 sigmoid; the emb half is S ------------------------------------------------
            bf16x8 ch[12], cl[12];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                int lzp;
                const __bf16* W = panel_begin(lzp);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int nb = 2 * q + u;
                    mma_t<8, NP>(S[nb], W, u, ph, pl, lane);
                    act_split<ACT_SIGMOID, NP>(S[nb], ch[2 * nb], cl[2 * nb], ch[2 * nb + 1], cl[2 * nb + 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                panel_end();
            }
            // ---- panels 7..10: cex2 192 -> 128 sigmoid -----------------------------------------------------------------------------
            bf16x8 dh[8], dl[8];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                int lzp;
                const __bf16* W = panel_begin(lzp);
                const float* bz = bl + lzp;
                f32x16 acc;
                acc_from_bias(acc, bz + B_CEX2 + 32 * nb, h);
                mma_t<12, NP>(acc, W, 0, ch, cl, lane);
                act_split<ACT_SIGMOID, NP>(acc, dh[2 * nb], dl[2 * nb], dh[2 * nb + 1], dl[2 * nb + 1]);
                panel_end();
            }
            // ---- panel 11: fin0 128 -> 64 GELU; then pose_final.2 (resident): 64 -> 3 GELU, direct orientation ---------------------
            bf16x8 fh[4], fl[4];
            {
                int lzi;
                const __bf16* W = panel_begin(lzi);
                const float* bz = bl + lzi;
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    f32x16 acc;
                    acc_from_bias(acc, bz + B_FIN0 + 32 * nb, h);
                    mma_t<8, NP>(acc, W, nb, dh, dl, lane);
                    act_split<ACT_GELU, NP>(acc, fh[2 * nb], fl[2 * nb], fh[2 * nb + 1], fl[2 * nb + 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const bf16x8 wh = wfrag<NP>(wf2 + lzi, 4, 0, s, 0, lane);
                    if constexpr (NP == 2) {
                        const bf16x8 wlo = wfrag<NP>(wf2 + lzi, 4, 0, s, 1, lane);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl[s], wh, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[s], wlo, acc, 0, 0, 0);
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[s], wh, acc, 0, 0, 0);
                }
                int lf = lane;
                asm volatile("" : "+v"(lf));       // (laundered like the scalar sections: no per-lane addresses kept across the step loop)
                const int c = lf & 31;
                if (c < 3) {
                    const float bv = bz[B_FIN2 + c];
#pragma unroll
                    for (int r = 0; r < 16; ++r) res[acc_row(r, lf) * 3 + c] = gelu_fast(acc[r] + bv);
                }
                panel_end();
            }
        };

#pragma unroll 1
        for (int step = 0; step < nsteps; ++step) {
            const bool last = step == nsteps - 1;
            // ---- guidance g (fp32): depends on the scene conditioning only.  pcd_step / o12_step != 0: the launch covers the steps of a
            // loop-invariant group, every step with its own encoder output (step-major) -- g is formed at the top of each step; 0: one
            // conditioning for all steps of the launch (single step, hoisted groups).  (The group's guidance from a launch of its own, read
            // here as one coalesced row per step, was measured: 59 -> 57 us per step -- not kept.)
            if (step == 0 || pcd_step != 0) {
                const float* pcd_s = pcd_b + (size_t)step * pcd_step;
                const float* o12_s = o12 + (size_t)step * o12_step;
                // (the lane index of the scalar sections is laundered at each of them: formed from `lane` itself, the element addresses of
                // x / noise / out_* / the LDS slots are loop-invariant per-lane values that hipcc hoists out of the step loop and then
                // spills -- 41 VGPRs of scratch in round 3's build; recomputing them costs a handful of VALU instructions per section)
                int ln = lane;
                asm volatile("" : "+v"(ln));
                if (ln < kSlots * 3) {
                    const int op = ln / 3, c = ln - op * 3;
                    float s = 0.0f;
                    for (int k = 0; k < 12; ++k) s = fmaf(o12_s[((size_t)b * kSlots + op) * 12 + k], pwt.w[c * 15 + 3 + k], s);
                    oterm[ln] = s;
                }
                wave_sync();
#pragma unroll 1
                for (int i = ln; i < 96; i += 64) {
                    const int row = i / 3, c = i - row * 3, n = n0 + row;
                    float acc = 0.0f;
                    for (int op = 0; op < kSlots; ++op) {
                        const float p0 = scrambled_r(pcd_s, w_b, op, n, 0), p1 = scrambled_r(pcd_s, w_b, op, n, 1), p2 = scrambled_r(pcd_s, w_b, op, n, 2);
                        float s = fmaf(p2, pwt.w[c * 15 + 2], fmaf(p1, pwt.w[c * 15 + 1], p0 * pwt.w[c * 15]));
                        s = gelu_fast(s + oterm[op * 3 + c] + pwt.b[c]);
                        // (N,-1,B,O) view: the multiplier of flat element f is mask.flatten()[f % (B*O)]  (Q2)
                        const long long f = (((long long)b * kSlots + op) * kPoints + n) * 3 + c;
                        acc += s * mask[f % ((long long)B * kSlots)];
                    }
                    gt[i] = (acc + (i < 64 ? hm0 : hm1)) / 2.0f;
                }
                wave_sync();
            }
            // in-place update of x (Q5); the caller's tensor sees it when the launch is a single step
            int lu = lane;
            asm volatile("" : "+v"(lu));
#pragma unroll 1
            for (int i = lu; i < 96; i += 64) {
                const float xv = xs[i] + gt[i];
                zt[i] = xv;
                if (!chain && live) x[g0 + i] = xv;
            }
            wave_sync();
            // ---- emb = GELU(ctime + (ctext + b)) -> K = 128 operand: k-slot (s, h, j) = channel 16 s + 8 h + j ----------------------
            bf16x8 eh[8], el[8];
            {
                const float* cm = opaque_ptr(ctime) + ((size_t)(ctime_per_sample ? step * B + b : step) * kPoints + n0 + p) * 128 + 8 * h;
                const float* ct = opaque_ptr(ctext) + ((size_t)b * kPoints + n0 + p) * 128 + 8 * h;
                int lz = 0;
                asm volatile("" : "+v"(lz));
#pragma unroll
                for (int s0 = 0; s0 < 8; s0 += 4) {
                    f32x4 a[4][2], c[4][2];
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            a[s][q] = gload<f32x4>(cm + 16 * (s0 + s) + 4 * q);
                            c[s][q] = gload<f32x4>(ct + 16 * (s0 + s) + 4 * q);
                        }
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        float v[8];
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const f32x4 bq = *reinterpret_cast<const f32x4*>(bl + lz + B_COMB + 16 * (s0 + s) + 8 * h + 4 * q);
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[4 * q + j] = gelu_fast(a[s][q][j] + (c[s][q][j] + bq[j]));
                        }
                        split8_h<NP>(v, eh[s0 + s], el[s0 + s]);
                    }
                }
            }
            // ---- panels 0..2: S = W_cex0[:, 128:] . emb + b -----------------------------------------------------------------------
            f32x16 S[6];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                int lzp;
                const __bf16* W = panel_begin(lzp);
                const float* bz = bl + lzp;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int nb = 2 * q + u;
                    acc_from_bias(S[nb], bz + B_CEX0 + 32 * nb, h);
                    mma_t<8, NP>(S[nb], W, u, eh, el, lane);
                    __builtin_amdgcn_sched_barrier(0);
                }
                panel_end();
            }
            const bool guide = last && out_guiding != nullptr;   // out_guiding == NULL: a later forward of the call defines the guiding points
            // The 24 piece addresses of the scratch tile are formed where they are used, from a pointer the compiler cannot see
            // through: as loop invariants of the round it kept all of them in registers across the step loop and spilled them
            // (79 VGPRs of scratch in round 2's build).
            auto stash_tile = [&]() -> float* {
                float* q = stash;
                asm volatile("" : "+s"(q));
                return q + (size_t)blk * (6 * 16 * 64) + lane * 4;
            };
            if (guide) {
                // park S for the guiding-point pass (lane-linear 16-byte pieces: 1 KiB per instruction)
                float* sst = stash_tile();
#pragma unroll
                for (int nb = 0; nb < 6; ++nb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = {S[nb][4 * q], S[nb][4 * q + 1], S[nb][4 * q + 2], S[nb][4 * q + 3]};
                        *reinterpret_cast<f32x4*>(sst + (size_t)(nb * 4 + q) * 256) = v;
                    }
            }
            // ---- denoiser on the mutated x + sampler update -----------------------------------------------------------------------
            pass(zt, S);
            wave_sync();
            int ls = lane;
            asm volatile("" : "+v"(ls));
#pragma unroll 1
            for (int i = ls; i < 96; i += 64) {
                const float x0 = res[i];
                if (last && live) out_x0[g0 + i] = x0;
                if (mode != MODE_FORWARD) {
                    const int tb = (int)t[(size_t)step * B + b];
                    const float xm = zt[i], nz = tb != 0 ? 1.0f : 0.0f;
                    const float nv = noise ? noise[(size_t)step * B * kPoints * 3 + g0 + i] : 0.0f;
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
                    xs[i] = smp;
                    if (last && live) (chain ? x : out_sample)[g0 + i] = smp;
                }
            }
            wave_sync();
            // ---- denoiser on the guiding points (saved_guiding_points, Q12): defined by the last forward ----------------------------
            if (guide) {
                const float* sst = stash_tile();
#pragma unroll
                for (int nb = 0; nb < 6; ++nb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = gload<f32x4>(sst + (size_t)(nb * 4 + q) * 256);
#pragma unroll
                        for (int j = 0; j < 4; ++j) S[nb][4 * q + j] = v[j];
                    }
                pass(gt, S);
                wave_sync();
                int lg = lane;
                asm volatile("" : "+v"(lg));
#pragma unroll 1
                for (int i = lg; i < 96; i += 64)
                    if (live) out_guiding[g0 + i] = res[i];
                wave_sync();
            }
        }
    }
}

void launch_denoise_r(int np, const DenseH& blob, const Lin& pwt, const StepCoefs& sc, int mode, float eta, int B, int nsteps, int ctime_per_sample,
                      float* x, const int64_t* t, const float* mask, const float* pcd, const float* w, const float* o12, const float* hm,
                      const float* ctext, const float* ctime, const float* noise, float* stash, float* out_x0, float* out_guiding,
                      float* out_sample, int num_cus, hipStream_t st, long long pcd_step, long long o12_step) {
    const int rounds = (B * (kPoints / 32) + NW - 1) / NW;
    const int grid = rounds < num_cus ? rounds : num_cus;
    if (np == 2)
        hipLaunchKernelGGL(denoise_r_kernel<2>, dim3(grid), dim3(NT), 0, st, blob.w, blob.b, pwt, sc, mode, eta, B, nsteps, ctime_per_sample, x, t, mask, pcd,
                           w, o12, hm, ctext, ctime, noise, stash, out_x0, out_guiding, out_sample, pcd_step, o12_step);
    else
        hipLaunchKernelGGL(denoise_r_kernel<1>, dim3(grid), dim3(NT), 0, st, blob.w, blob.b, pwt, sc, mode, eta, B, nsteps, ctime_per_sample, x, t, mask, pcd,
                           w, o12, hm, ctext, ctime, noise, stash, out_x0, out_guiding, out_sample, pcd_step, o12_step);
}

}  // namespace lsdm
