// Parameter blocks and host-side launchers of liblsdm_hip's kernels (internal header).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bf16_core.hpp"

namespace lsdm {

struct Dense {  // folded (conv|linear)[+BN] layer for the MFMA core: packed weights Wp[kb][n][8], bias[n]
    const float* w;
    const float* b;
};

struct Lin {  // plain torch.nn.Linear parameters: weight (N,K) row-major, bias (N)
    const float* w;
    const float* b;
};

struct CondParams {
    const float* pe;                            // sequence_pos_encoder.pe (5000,1,128)
    Lin time0, time2;                           // embed_timestep.time_embed.{0,2}
    Lin text0, text2, text4;                    // embed_text.{0,2,4}: D -> D/2 -> 256 -> 128
    int text_dim;                               // D = clip_dim: 512 (CLIP ViT-B/32, util/model_util.py:26-72) or 768 (BERT pooler, sdm.py:261-264)
    Lin pcat0, pcat2, pcat4;                    // predict_cat.{0,2,4}
    Lin ecat;                                   // embed_cat.0
    const float *attn_wq, *attn_wk, *attn_inb;  // attn_layer.{q,k}_proj_weight, in_proj_bias
    Lin tr0, tr2;                               // translation_layer.{0,2}
};

struct PosaParams {
    Lin lin[3];            // human_backbone.de_spiral.{0,1,2}.conv.layer
    const float* gn_w[3];  // human_backbone.de_spiral.{0,1,2}.norm.{weight,bias}
    const float* gn_b[3];
    Lin last;              // human_backbone.de_spiral.3.layer (3,64)
    Dense d1, d2;          // lin[1], lin[2] packed for the exact-fp32 matrix pipe (mlp_core.hpp)
};

struct SceneAttnParams {
    const float *wq, *wk, *wv, *inb;  // pcd_attention.{q,k,v}_proj_weight (12,12),(12,3),(12,3); in_proj_bias (36)
    Lin outp;                         // pcd_attention.out_proj (12,12)
};

struct DenoiseParams {
    const float* up4_w;   // upsampling_layer.4.weight (1024,512) row-major: the A operand of the third up-sampling layer
    const float* up4_b;   // (1024) -- a per-ROW bias in the transposed product
    Dense comb;           // combine_extraction.0            256 -> 128  GELU
    Lin pose0;            // input_process.pose_embedding.0    3 -> 64   sigmoid (VALU)
    Dense pose2;          // input_process.pose_embedding.2   64 -> 128  sigmoid
    Dense cex0, cex2;     // input_process.combination_extraction.{0,2}  256 -> 192 -> 128 sigmoid
    Dense fin0;           // output_process.pose_final.0     128 -> 64   GELU
    Lin fin2;             // output_process.pose_final.2      64 -> 3    GELU (VALU)
    Lin pwt;              // point_wise_trans_layer.0         15 -> 3    GELU
};

struct DenoiseHParams {  // split-bf16 operands of the same layers (denoise_bf.hip)
    const __bf16* up4_h;  // upsampling_layer.4.weight (1024,512) as bf16 hi / lo planes, row-major (the "activation" operand)
    const __bf16* up4_l;
    const float* up4_b;
    DenseH comb, pose2, cex0, cex2, fin0;
    DenseH fin2m;         // output_process.pose_final.2 (3 x 64) zero-padded to one 32-channel block (bias: fin2.b)
    Lin pose0, fin2, pwt;
};

struct StepCoefs {  // per-timestep fp32 tables (gathered by t inside the kernel); see diffusion/schedule.py
    const float* coef1;     // posterior_mean_coef1
    const float* coef2;     // posterior_mean_coef2
    const float* logvar;    // posterior_log_variance_clipped
    const float* sr_acp;    // sqrt_recip_alphas_cumprod      (DDIM)
    const float* srm1_acp;  // sqrt_recipm1_alphas_cumprod    (DDIM)
    const float* acp;       // alphas_cumprod                 (DDIM)
    const float* acp_prev;  // alphas_cumprod_prev            (DDIM)
};

enum StepMode { MODE_FORWARD = 0, MODE_P_SAMPLE = 1, MODE_DDIM = 2 };

// geometry.hip
// num_clouds = clouds of the launch; nc_step = clouds per step (num_clouds = G * nc_step for a G-step geometry batch);
// src_mod / fine_mod = nc_step when the operand is the level-0 input cloud (shared by the steps), else 0
void launch_fps_chain(const float* clouds, const int64_t* starts, int num_clouds, int nc_step, int* const idx[4], float* const xyz[4],
                      hipStream_t st);
void launch_ball_query(const float* xyz, const float* new_xyz, int num_clouds, int N, int S, float r2, int* out, int src_mod, hipStream_t st,
                       const unsigned char* only = nullptr, bool scan = false);
// (only: one flag per centroid, 0 = skip it; scan: the ordered-scan kernel also where the grid kernel would be chosen)
void launch_three_nn(const float* fine, const float* coarse, int num_clouds, int N, int S, int* out_idx, float* out_w, int fine_mod,
                     hipStream_t st);
// level-1 loop-invariant work translated into the step's FPS order (lsdm_sample_steps_invariant): feature rows + 3-NN positions
void launch_nn8(int num_clouds, const float* clouds, float* out_d, int* out_i, hipStream_t st);
// inv_mod > 0: a launch over the clouds of several steps -- the per-call inputs (F1, nn8_*, clouds) are those of cloud c % inv_mod
void launch_l1_remap(int num_clouds, const void* F1, int row_bytes, const int* idx1, const float* nn8_d, const int* nn8_i,
                     const float* l1_xyz, const float* clouds, void* feat1, int* nn_idx, float* nn_w, hipStream_t st, int inv_mod = 0);
// encoder.hip
void launch_sa(int level, int num_clouds, const float* src_xyz, const float* src_feat, const float* new_xyz, const int* ball,
               const Dense* l, float* out, hipStream_t st);
void launch_fp(int which, int num_clouds, const float* points1, const float* points2, const int* nn_idx, const float* nn_w,
               const Dense* l, const Dense* head, float* out, hipStream_t st);
// encoder_bf.hip (np = 2: bf16x3 with fp32 features, np = 1: bf16 with bf16 features)
void launch_sa_h(int np, int level, int num_clouds, const float* src_xyz, const void* src_feat, const float* new_xyz, const int* ball,
                 const DenseH* l, void* out, hipStream_t st);
void launch_fp_h(int np, int which, int num_clouds, const void* points1, const void* points2, const int* nn_idx, const float* nn_w,
                 const DenseH* l, const DenseH* head1, Lin head2, void* out, hipStream_t st);
// encoder_reg.hip: register-chained SA kernel for levels 0 and 1 (weights resident in LDS)
void launch_sa_r(int np, int level, int num_clouds, const float* src_xyz, const void* src_feat, const float* new_xyz, const int* ball,
                 const __bf16* blob_w, const float* blob_b, void* out, hipStream_t st);
// y = coarse-level FP output x Wb^T (the pre-multiplied interpolation operand of the next finer FP stage), R rows of 256 channels
// fp1 + head as one register-chained stack (bf16x3), a pipeline of stages with register-resident weights (encoder_d.hip: fp_c_kernel):
// points2 (nc,1024,128) fp32, out (nc,1024,3)
// order (optional): walk each cloud's points in that order (cloud c % order_mod's when order_mod > 0) -- results do not depend on it
void launch_fp_c(int np, int num_clouds, const float* points2, const int* nn_idx, const float* nn_w, const __bf16* blob_w, const float* blob_b, float* out,
                 hipStream_t st, const int* order = nullptr, int order_mod = 0);
// encoder_u.hip (bf16x3): first SA layer split across the grouping -- U = X x Wf^T + xyz x Wx^T + b, V = xyz x Wx^T per source
// point (level 1: sa2, 64 -> 64; level 2: sa3, 128 -> 128; level 3: sa4, 256 -> 256), then the two remaining layers on
// ReLU(U_j - V_i).  remap (level 1 only, optional): U / V rows are in SOURCE order and position i of a cloud is row remap[i].
void launch_lin_u(int np, int level, int R, const float* X, const float* xyz, DenseH W, const float* wx, float* U, float* V, hipStream_t st);
// src_mod > 0 (level 1 with remap, a launch over the clouds of several steps): U / V are per-CALL tensors, those of cloud c % src_mod
// memo of sa2 (geometry.hip): per call the uncrowded source points (flags + list), per group the copies / the crowded centroids' list
void launch_memo_src(const int* ball2, long long total, unsigned char* unc, int* list, int* count, hipStream_t st);
void launch_memo_cent(int clouds, const int* idx0, const int* idx1, const unsigned char* unc, const float* F2, int nc1, float* out, int* list, int* count,
                      unsigned char* crowd, hipStream_t st);
void launch_sa_u_list(int np, int S, const float* U, const float* V, const int* cent_idx, const int* ball, const int* remap, const __bf16* blob_w,
                      const float* blob_b, float* out, hipStream_t st, int src_mod, const int* list, const int* list_count);
void launch_sa_u(int np, int level, int num_clouds, const float* U, const float* V, const int* cent_idx, const int* ball, const int* remap,
                 const __bf16* blob_w, const float* blob_b, float* out, hipStream_t st, int src_mod = 0);
// sa3 / sa4 over ITEM lists (two ball-query groups whose distinct neighbours fit into 32 rows together share a block; geometry.hip: sa_pair_kernel):
//   launch_sa_pair_plan: `segs` segments of T1 = clouds x S centroids each (ball (segs*T1, 32), cent_idx (segs*T1)) -> rows (segs*T1, 32),
//                        meta (segs*T1) int4, count (segs); chunk_np: segs * sa_pair_chunks(T1) ints of scratch
//   launch_sa_u_items  : one segment's stage (level 1 = sa2's memo pass over the source points, 2 = sa3, 3 = sa4) over its list
//   (only: one byte per centroid, zero = the group gets no block; cent_idx == nullptr: centroid e is point e % S of its cloud)
struct PairLevel { const int* ball; const int* cent_idx; int T1, S, NSRC, CPS /* filled in */, segs; int* chunk_np; int* rows; int4* meta; int* count;
                   const unsigned char* only; int first_block /* filled in */; };
struct PairLevels { PairLevel l[4]; int n; };
int sa_pair_chunks(int T1);
void launch_sa_pair_plans(const PairLevel* levels, int n, hipStream_t st);      // up to four levels in one pair of launches
void launch_sa_pair_plan(int segs, int T1, int S, int NSRC, const int* ball, const int* cent_idx, int* chunk_np, int* rows, int4* meta, int* count,
                         hipStream_t st, const unsigned char* only = nullptr);
void launch_sa_r_items(const float* src_xyz, const int* rows, const int4* meta, const int* count, const __bf16* blob_w, const float* blob_b, float* out,
                       hipStream_t st);      // sa1 (register-chained kernel, first layer included)
void launch_sa_u_items(int np, int level, const float* U, const float* V, const int* rows, const int4* meta, const int* count, const __bf16* blob_w,
                       const float* blob_b, float* out, hipStream_t st);
// fp2 / fp3 with the interpolated operand pre-multiplied at the coarse level (encoder_u.hip)
// p1_remap (optional): row i of a cloud's points1 is points1[p1_remap[i]]
// y_ready: ybuf was written by the coarser stage's fused extra layer; wb_next / ynext (which = 3, optional): this launch also writes
// fp2's pre-multiplied operand ynext = out x wb_next^T, and stores `out` itself only when keep_out (parity taps)
void launch_fp_pre(int np, int which, int num_clouds, const float* points1, const int* p1_remap, const float* fpo_coarse, DenseH wb, float* ybuf,
                   const int* nn_idx, const float* nn_w, DenseH l1a, DenseH l2, float* out, bool y_ready, DenseH wb_next, float* ynext,
                   bool keep_out, hipStream_t st, int p1_mod = 0,      // p1_mod > 0 (with p1_remap): points1 is a per-CALL tensor, cloud c % p1_mod
                   const int* plan = nullptr);
// (plan, which = 2: the tile plan of the launch's clouds, written by launch_fp_plan -- fp2 then runs on fp_d_kernel)
// encoder_d.hip: fp2 with the gathered rows deduplicated through LDS (Morton-ordered tiles, each distinct row landed once by LDS-DMA)
//   launch_morton_order: order (nc,1024) <- the Z-curve order of each cloud's points
//   launch_fp_plan     : plan (nc, fp_plan_ints(1024)) <- tiles of 32 points along `order` (of cloud c % order_mod when order_mod > 0), their distinct
//                        coarse rows and every point's slots / weights; idx1 != NULL: `order` sorts SOURCE points and the fine level is idx1's order
size_t fp_plan_ints(int N);
int fp2_ws();      // LSDM_FP2_WS (read per launch): 4 = fp_d_kernel
void launch_morton_order(int num_clouds, const float* xyz, int* order, hipStream_t st);
void launch_fp_plan(int num_clouds, const int* order, int order_mod, const int* idx1, const int* nn_idx, const int* p1_remap, int* plan, hipStream_t st);
void launch_fp_d(int np, int num_clouds, const float* points1, const float* ybuf, const float* nn_w, DenseH l1a, DenseH l2, float* out, const int* plan, hipStream_t st,
                 int p1_mod);
// cond.hip
// G = number of consecutive steps covered by the launch (t, e0, tr, w hold G*B rows; text / cats / mask are per sample)
void launch_cond(const CondParams& P, const int64_t* t, const float* text, const float* given_cats, const float* mask, int B, int G, int C,
                 float* out_cat, float* e0, float* tr, float* w, hipStream_t st);
// timestep embedding only: row r of e0 (256 floats per row, first 128 written) <- t[r * tstride]
void launch_cond_time(const CondParams& P, const int64_t* t, int tstride, int rows, float* e0, hipStream_t st);
// ws: B * 664 floats of scratch (moment triples of the GroupNorm reductions)
void launch_posa(const PosaParams& P, const float* given_objs, int B, float* ws, float* hm, hipStream_t st);
constexpr int kPosaWsPerSample = 16 + 16 * 24 + 11 * 24;
// np = 0: fp32 operand for denoise_kernel (all 256 channels of e0 per row); 1 / 2: bf16 planes for denoise_h_kernel /
// ctext_kernel, the 128 channels from c_off (0: timestep half, one row per (step, sample); 128: text half, one row per sample)
void launch_upsample12(int np, const float* e0, Lin up0, const float* up2_wp, const float* up2_b, int rows, int c_off, void* h2p, hipStream_t st);
void launch_scene_attn(const SceneAttnParams& P, const float* pcd, const float* w, const float* tr, int B, float* o, hipStream_t st, int steps = 1);
// (steps > 1: pcd / o hold that many steps of B samples, step-major; w / tr are the call's B sample rows)
// denoise.hip
void launch_denoise(const DenoiseParams& P, const StepCoefs& sc, int mode, float eta, int B, float* x, const int64_t* t,
                    const float* mask, const float* pcd, const float* w, const float* o12, const float* hm, const float* h2p,
                    const float* noise, float* out_x0, float* out_guiding, float* out_sample, hipStream_t st);

// metrics.hip: per point of A (batch,na,3) the squared distance to / index of its nearest point of B (batch,nb,3)
void launch_nn_dist(const float* A, int na, const float* B, int nb, int batch, float* out_d2, int* out_idx, hipStream_t st);

// denoise_bf.hip
// ctext (B,1024,128) fp32: the text half of combine_extraction's pre-activation, once per call (h2x: text half of the
// second up-sampling activation, one row per sample)
// time_half: the same product for the TIMESTEP half, one row of h2 per step (multi-step calls, all chains at the same timestep):
// (rows,1024,128) fp32 consumed by denoise_h_kernel through `ctime`
void launch_ctext(int np, const DenoiseHParams& P, int rows, const void* h2x, float* ctext, bool time_half, hipStream_t st);
// nsteps consecutive steps in one launch (t / noise hold nsteps rows; out_sample == NULL in a sampling mode: the final sample goes
// to x).  h2t rows: one per (step, sample) when h2t_per_sample, one per step otherwise (all chains at the same timestep)
void launch_denoise_h(int np, const DenoiseHParams& P, const StepCoefs& sc, int mode, float eta, int B, int nsteps, int h2t_per_sample, float* x, const int64_t* t,
                      const float* mask, const float* pcd, const float* w, const float* o12, const float* hm, const void* h2t,
                      const float* ctext, const float* ctime, const float* noise, float* out_x0, float* out_guiding, float* out_sample, hipStream_t st);

// denoise_reg.hip: the same step as launch_denoise_h with the per-point denoiser register-chained (blob: api.hip::build_denoise_reg_blob);
// ctime: (rows,1024,128) from launch_ctext(time_half), row = step (ctime_per_sample = 0) or step * B + sample; stash: B * 32 tiles of 24 KB
void launch_denoise_r(int np, const DenseH& blob, const Lin& pwt, const StepCoefs& sc, int mode, float eta, int B, int nsteps, int ctime_per_sample,
                      float* x, const int64_t* t, const float* mask, const float* pcd, const float* w, const float* o12, const float* hm,
                      const float* ctext, const float* ctime, const float* noise, float* stash, float* out_x0, float* out_guiding,
                      float* out_sample, int num_cus, hipStream_t st, long long pcd_step = 0, long long o12_step = 0);
// (pcd_step / o12_step != 0: step s of the launch reads pcd + s * pcd_step, o12 + s * o12_step -- the steps of a loop-invariant group)

}  // namespace lsdm
