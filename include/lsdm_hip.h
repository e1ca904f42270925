/* liblsdm_hip -- C ABI of the MI355X (gfx950) implementation of LSDM's denoising sampling path.
 *
 * The reference (andvg3/LSDM) is pure Python/PyTorch and has no FFI of its own; the drop-in
 * boundary at the Python level is SceneDiffusionModel.forward (model/sdm.py:131) and
 * GaussianDiffusion.p_sample (diffusion/gaussian_diffusion.py:501).  This header is the C boundary underneath the
 * Python mirror in lsdm_amd/: plain pointers and sizes, no C++ or torch types.  Each entry point names the
 * reference code it replaces.  INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success or a negative lsdm_status; lsdm_last_error() gives the message.
 *     Nothing throws across the boundary.
 *   - the caller allocates every input/output (raw device pointers, e.g. tensor.data_ptr()); the library owns only
 *     its parameter arena and a workspace inside the context (sized by lsdm_reserve).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls are asynchronous.  Everything a call does is ordered
 *     on `stream`: lsdm_sample_steps_invariant also runs the dense encoder chains of alternate steps on a side stream of the context
 *     (forked from `stream` by an event and joined back before the call's last launches; LSDM_STREAMS=1 keeps one stream) -- work the
 *     caller enqueues on `stream` after the call sees all of its results.
 *   - one context per (process, device); a context is not thread-safe; different contexts are independent.
 *   - all floating-point tensors are fp32, dense, row-major; timesteps / FPS starts are int64 (torch.long),
 *     internal index outputs are int32.
 *   - fixed by the model (model/sdm.py:19-22, pointnet2.py:46-55): 9 object slots, 1024 points per cloud, latent 128.
 *     The text feature width D is the model's clip_dim, read off embed_text.0.weight (D/2, D): 512 with the defaults of
 *     util/model_util.py (CLIP ViT-B/32), 768 for the BERT pooler output (model/sdm.py:261-264); every `text` pointer is (B, D).
 */
#ifndef LSDM_HIP_H
#define LSDM_HIP_H

#include <stdint.h>

/* The library is built with -fvisibility=hidden: the entry points below are its only dynamic symbols. */
#define LSDM_API __attribute__((visibility("default")))

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lsdm_ctx lsdm_ctx;

typedef enum {
    LSDM_OK = 0,
    LSDM_ERR_ARG = -1,      /* bad argument / shape */
    LSDM_ERR_STATE = -2,    /* call order (parameters not finalised, workspace too small, ...) */
    LSDM_ERR_HIP = -3,      /* HIP runtime error */
    LSDM_ERR_PARAM = -4     /* missing or mis-shaped parameter */
} lsdm_status;

typedef enum {
    LSDM_MODE_FORWARD = 0,  /* SceneDiffusionModel.forward only                      (model/sdm.py:131-218) */
    LSDM_MODE_P_SAMPLE = 1, /* forward + ancestral step                 (gaussian_diffusion.py:501-561) */
    LSDM_MODE_DDIM = 2      /* forward + DDIM step                      (gaussian_diffusion.py:790-811) */
} lsdm_mode;

typedef enum {
    LSDM_PREC_F32 = 0,      /* exact fp32 products: v_mfma_f32_32x32x2_f32 (bitwise an fmaf chain)            */
    LSDM_PREC_BF16X3 = 1,   /* fp32 operands split into bf16 hi+lo, 3 bf16 MFMAs per product, fp32 accumulate:
                             * fp32-class results (dropped term 2^-18 relative) at 5.3x the f32 MFMA rate       */
    LSDM_PREC_BF16 = 2      /* plain bf16 products (one MFMA per product), fp32 accumulate and fp32 level features     */
} lsdm_precision;

/* ---- lifetime ------------------------------------------------------------------------------------------- */
LSDM_API int lsdm_create(lsdm_ctx** out, int device);
LSDM_API int lsdm_destroy(lsdm_ctx* ctx);
/* message of the last failing call on ctx (ctx may be NULL for lsdm_create failures) */
LSDM_API const char* lsdm_last_error(const lsdm_ctx* ctx);

/* ---- parameters: replaces model.load_state_dict(ckpt['model_state_dict'])  (run/test_sdm.py:124) ---------- */
/* `name` is a state_dict key of SURVEY.md Appendix A; `data` is a HOST fp32 pointer (num_batches_tracked and the
 * duplicate pe alias are ignored).  Shapes are checked in lsdm_finalize_params. */
LSDM_API int lsdm_set_param(lsdm_ctx* ctx, const char* name, const float* data, const int64_t* shape, int ndim);
/* arithmetic of the dense (1x1-conv / Linear) stacks; geometry (FPS, ball query, 3-NN), normalisations, softmaxes and
 * the sampler update are always fp32.  Call before lsdm_finalize_params / lsdm_reserve (changing it invalidates both). */
LSDM_API int lsdm_set_precision(lsdm_ctx* ctx, int precision);
LSDM_API int lsdm_get_precision(const lsdm_ctx* ctx);
/* folds eval-mode BatchNorm into the 1x1 convolutions, packs the MFMA operands, uploads.  max_cats = C. */
LSDM_API int lsdm_finalize_params(lsdm_ctx* ctx, int max_cats);
/* per-timestep fp32 tables of the (possibly respaced) diffusion process, length T each:
 * posterior_mean_coef1/2, posterior_log_variance_clipped, sqrt_recip_alphas_cumprod, sqrt_recipm1_alphas_cumprod,
 * alphas_cumprod, alphas_cumprod_prev   (gaussian_diffusion.py:166-202, cast to fp32 as _extract_into_tensor :1594) */
LSDM_API int lsdm_set_schedule(lsdm_ctx* ctx, int T, const float* coef1, const float* coef2, const float* logvar,
                      const float* sqrt_recip_acp, const float* sqrt_recipm1_acp, const float* acp, const float* acp_prev);
/* allocates the workspace for batches up to max_batch samples (stage tensors + the geometry sets of 32 steps: ≈ 115 MB per chain; the
 * first multi-step call of the loop-invariant schedule adds the group workspace -- per-step slots of 32 steps' level features, ≈ 97 MB per
 * chain: 13.7 GB in all at B = 64, measured) */
LSDM_API int lsdm_reserve(lsdm_ctx* ctx, int max_batch);

/* ---- the hot path ---------------------------------------------------------------------------------------- */
/* One denoising-network evaluation, optionally fused with the sampler update.
 *   x            (B,1024,3)   in/out: mutated in place, x += g (model/sdm.py:204)
 *   mask         (B,9)        float 0/1
 *   t            (B)          int64 timesteps (index into the schedule tables and the pe buffer)
 *   given_objs   (B,9,1024,3)
 *   given_cats   (B,9,C)
 *   text_feat    (B,D)        text-encoder output, D = clip_dim (the path boundary, model/sdm.py:146-149)
 *   fps_start    (4,B*9)      int64 FPS start indices of sa1..sa4 (pointnet2_utils.py:72), ranges 1024,1024,256,64
 *   noise        (B,1024,3)   randn draw of p_sample (gaussian_diffusion.py:545); may be NULL for LSDM_MODE_FORWARD
 *   out_cat      (B,1,C)      softmax category prediction (= model.saved_cat)
 *   out_x0       (B,1024,3)   model output = pred_xstart
 *   out_guiding  (B,1024,3)   model.saved_guiding_points
 *   out_sample   (B,1024,3)   x_{t-1}; ignored for LSDM_MODE_FORWARD
 */
LSDM_API int lsdm_step(lsdm_ctx* ctx, int mode, float eta, int B, float* x, const float* mask, const int64_t* t,
              const float* given_objs, const float* given_cats, const float* text_feat, const int64_t* fps_start,
              const float* noise, float* out_cat, float* out_x0, float* out_guiding, float* out_sample, void* stream);

/* K consecutive ancestral steps t = t_first, t_first-1, ... with device-resident RNG draws; `x` ping-pongs with
 * an internal buffer and holds the final sample on return.  The x-independent PointNet++ geometry (FPS, ball queries,
 * 3-NN) and conditioning MLPs of up to 32 consecutive steps are evaluated in one launch per stage ahead of those steps;
 * results are identical to K single-step calls.  Replaces the loop body of
 * p_sample_loop_progressive (gaussian_diffusion.py:736-759) for a fixed batch.
 *   fps_start_steps (K,4,B*9) int64 ; noise_steps (K,B,1024,3) */
LSDM_API int lsdm_sample_steps(lsdm_ctx* ctx, int mode, float eta, int B, int t_first, int K, float* x, const float* mask,
                      const float* given_objs, const float* given_cats, const float* text_feat,
                      const int64_t* fps_start_steps, const float* noise_steps, float* out_cat, float* out_x0,
                      float* out_guiding, void* stream);

/* Same K steps, same results, with the LOOP-INVARIANT part of the scene encoder's finest level evaluated once per call.
 * farthest_point_sample at the finest level draws all 1024 points of a cloud (pointnet2.py:46: npoint = 1024 of 1024), so the
 * level-1 centroids are the cloud itself in a step-dependent order: the ball of a centroid (query_ball_point,
 * pointnet2_utils.py:84-104: first 32 source indices within r, in SOURCE order), its sa1 feature, and the three nearest
 * level-1 points of every source point (:283-298) depend on which point it is, not on its position in the FPS order.  They
 * are computed once; each step only re-orders them (feature rows, 3-NN positions with the per-step tie rule re-applied).
 * The arithmetic and its operands are those of lsdm_sample_steps -- results are bit-identical (tests) -- unlike the hoisted
 * variant below, which changes the random stream.  Clouds whose finest FPS is not a permutation (duplicated points) fall
 * back to the per-step 3-NN inside the kernel.
 * Degenerate clouds: a cloud whose 1024 points are bit-identical (the zero padding of a scene with fewer than eight objects,
 * posa/dataset.py:456) has an encoder output independent of the FPS draws (farthest_point_sample returns [start, 0, 0, ...], every
 * ball is indices 0..31: pointnet2_utils.py:60-104); the call classifies the clouds (ONE host synchronisation on `stream`, in front
 * of its other work), evaluates one representative per distinct such cloud once and runs the per-step encoder on the others only.
 * Results are bit-identical with and without it (LSDM_DEG_FAST=0). */
LSDM_API int lsdm_sample_steps_invariant(lsdm_ctx* ctx, int mode, float eta, int B, int t_first, int K, float* x, const float* mask,
                                const float* given_objs, const float* given_cats, const float* text_feat,
                                const int64_t* fps_start_steps, const float* noise_steps, float* out_cat, float* out_x0,
                                float* out_guiding, void* stream);

/* Hoisted-conditioning variant: the scene/text/category conditioning -- human branch, PointNet++
 * geometry AND dense stages, collapsed scene attention -- is evaluated ONCE with the pinned FPS starts `fps_start`
 * (4,B*9); each of the K steps then runs only the t-dependent embedding MLPs, the up-sampling layers and the fused
 * per-point denoiser + sampler update.  Equals the reference when its per-forward FPS re-draw (pointnet2_utils.py:72,
 * SURVEY.md Q11) is pinned to the same starts; NOT equal to the stock reference, whose scene features jitter per step. */
LSDM_API int lsdm_sample_steps_hoisted(lsdm_ctx* ctx, int mode, float eta, int B, int t_first, int K, float* x, const float* mask,
                              const float* given_objs, const float* given_cats, const float* text_feat, const int64_t* fps_start,
                              const float* noise_steps, float* out_cat, float* out_x0, float* out_guiding, void* stream);

/* ---- stage entry points (parity tests against the oracle's taps; same kernels the hot path launches) -------- */
/* farthest_point_sample x4 chained (pointnet2_utils.py:60-81): idx_l (num_clouds,S_l) int32, xyz_l (num_clouds,S_l,3),
 * S = 1024,256,64,16 */
LSDM_API int lsdm_fps(lsdm_ctx* ctx, const float* clouds, const int64_t* fps_start, int num_clouds, int32_t* idx1, int32_t* idx2,
             int32_t* idx3, int32_t* idx4, float* xyz1, float* xyz2, float* xyz3, float* xyz4, void* stream);
/* query_ball_point (pointnet2_utils.py:84-104): out (num_clouds,S,32) int32; N in {1024,256,64}.  `radius` is a
 * double like the reference's Python float; radius*radius is rounded to fp32 for the comparison */
LSDM_API int lsdm_ball_query(lsdm_ctx* ctx, const float* xyz, const float* new_xyz, int num_clouds, int N, int S, double radius,
                    int32_t* out, void* stream);
/* 3-NN + inverse-distance weights (pointnet2_utils.py:293-299): idx (num_clouds,N,3) int32, w (num_clouds,N,3) */
LSDM_API int lsdm_three_nn(lsdm_ctx* ctx, const float* fine, const float* coarse, int num_clouds, int N, int S, int32_t* idx,
                  float* w, void* stream);
/* get_backbone.forward (pointnet2.py:61-80): clouds (num_clouds,1024,3) -> out (num_clouds,1024,3) */
LSDM_API int lsdm_scene_encoder(lsdm_ctx* ctx, const float* clouds, const int64_t* fps_start, int num_clouds, float* out, void* stream);
/* POSA Decoder.forward (posa_models.py:320-326) on given_objs[:,0]: -> hm (B,1024,3) */
LSDM_API int lsdm_human_backbone(lsdm_ctx* ctx, const float* given_objs, int B, float* hm, void* stream);
/* device pointer + element count of a named workspace intermediate of the last call
 * ("feat1".."feat4","fp4","fp3","fp2","pcd","hm","w","tr","o12","e0","ball1".."ball4","nn_idx1".."nn_idx4") */
LSDM_API int lsdm_workspace_tensor(lsdm_ctx* ctx, const char* name, void** ptr, int64_t* numel);

/* ---- evaluation metrics of the CLI (run/test_sdm.py:185-196) -------------------------------------------------------- */
/* For every point of a (batch,na,3) the squared Euclidean distance to its nearest point of b (batch,nb,3), fp32
 * (ax-bx)^2+(ay-by)^2+(az-bz)^2, and optionally its index (lowest on ties; out_idx may be NULL).  One direction per
 * call.  The K = 1 search underneath pytorch3d.loss.chamfer_distance (run/test_sdm.py:186: mean over points of both
 * directions, summed) and open3d compute_point_cloud_distance (util/evaluation.py:37-38: sqrt(d2) < th for the F-score). */
LSDM_API int lsdm_nn_distance(lsdm_ctx* ctx, const float* a, int na, const float* b, int nb, int batch, float* out_d2,
                     int32_t* out_idx, void* stream);

/* ---- measurement: per-kernel durations from HIP events recorded on the launch stream ----------------------------- */
/* After lsdm_profile_begin every launch of the hot path is bracketed by events until lsdm_profile_end, which
 * synchronises and returns (number of kernels, or < 0): total_ms[k] = summed duration, launches[k] = launch count,
 * macs_per_unit[k] = algorithmic multiply-accumulates of one launch per cloud (PointNet++ kernels) or per sample
 * (the others) -- the figures of DESIGN.md / SURVEY.md 8(d).  Arrays must hold `capacity` >= 23 entries. */
LSDM_API int lsdm_profile_begin(lsdm_ctx* ctx);
LSDM_API int lsdm_profile_end(lsdm_ctx* ctx, int capacity, double* total_ms, int64_t* launches, double* macs_per_unit);
/* executed_macs[k] = the multiply-accumulates the launches of stage k ISSUED to the matrix pipe over the window just closed by
 * lsdm_profile_end, summed over all clouds / chains and launches, counted as products (MFMA MACs / 3 in the bf16x3 mode): K padded to
 * the instruction's k-step, 3-channel output layers padded to one channel block, linear first layers split across the grouping /
 * the concat boundary, work the loop-invariant schedule evaluates once per call counted once.  The reference has no
 * counterpart: its FLOPs are lsdm_profile_end's macs_per_unit. */
LSDM_API int lsdm_profile_executed(lsdm_ctx* ctx, int capacity, double* executed_macs);
LSDM_API const char* lsdm_profile_kernel_name(int k);

/* ---- training objective: replaces diffusion.training_losses(...) + loss.backward()  (gaussian_diffusion.py:1256-1342,
 *      run/train_sdm.py:62-80) -------------------------------------------------------------------------------------------------
 * A training session is its own object (it keeps every activation of a step; the sampling context keeps none).  Parameters are
 * NOT copied: lsdm_train_bind hands over, per state_dict key (BatchNorm running_mean / running_var and
 * sequence_pos_encoder.pe with grad = NULL), the device pointer of the caller's fp32 tensor and of the buffer that receives its
 * gradient -- an optimizer step on the caller's tensors is seen by the next lsdm_train_step.  attn_layer.{v_proj_weight,out_proj.*}
 * do not enter the loss and need not be bound. */
typedef struct lsdm_train lsdm_train;
LSDM_API int lsdm_train_create(lsdm_train** out, int device, int max_cats);
LSDM_API int lsdm_train_destroy(lsdm_train* s);
LSDM_API const char* lsdm_train_last_error(const lsdm_train* s);
LSDM_API int lsdm_train_bind(lsdm_train* s, const char* name, float* param, float* grad, int64_t numel);
/* One training batch: x_t = qa[b] x_start + qb[b] noise (q_sample, :233-256; qa = sqrt_alphas_cumprod[t], qb =
 * sqrt_one_minus_alphas_cumprod[t] gathered from the float64 tables and cast by the caller, as _extract_into_tensor does),
 * SceneDiffusionModel.forward(x_t, mask, t, given_objs, given_cats, text) with every activation kept,
 *   mse      = chamfer_distance(model_output, x_start)          (:1334; pytorch3d defaults: squared L2, point mean, both directions, batch mean)
 *   cat_loss = lambda_cat * CrossEntropy(out_cat, target)        (:1296-1300; out_cat already a softmax, target = class indices (B,))
 *   losses   = [mse + cat_loss, mse, cat_loss]                   (device, 3 floats)
 * and the reverse sweep: every bound gradient buffer is OVERWRITTEN with grad_scale * d (mse + cat_loss) / d parameter.
 * flags & 1: training-mode BatchNorm (batch statistics; the bound running statistics are updated in place, momentum 0.1) --
 * model.train() of run/train_sdm.py:46; otherwise the running statistics normalise (model.eval()).
 * flags & 2: the objective only (validation / torch.no_grad()): no reverse sweep, the bound gradient buffers are not touched.
 * dropout_keep (nullable): (B*9,128,1024) bytes, the keep-mask of the encoder head's Dropout(0.5) (pointnet2.py:58,76), drawn by
 * the caller (the reference draws it on the host generator right after the four FPS start draws); kept activations are doubled.
 * fps_starts: (4, B*9) int64 as in lsdm_step.  model_output (B,1024,3) and out_cat (B,max_cats) are optional outputs. */
LSDM_API int lsdm_train_step(lsdm_train* s, int B, int flags, const float* x_start, const float* qa, const float* qb, const float* noise,
                    const int64_t* t, const float* mask, const float* given_objs, const float* given_cats, const int64_t* target,
                    const float* text, const int64_t* fps_starts, const unsigned char* dropout_keep, float lambda_cat,
                    float grad_scale, float* losses, float* model_output, float* out_cat, void* stream);

/* Test hook: copies an intermediate of the LAST lsdm_train_step to host memory (synchronises the step's stream).
 * "sa1.arg" .. "sa4.arg": uint8 (B*9 * S, C) -- which of the 32 neighbours attains the max of every (centroid, channel), i.e. where the
 * max-pool routes the gradient.  (Two neighbours whose activations differ in the last bits make that choice depend on the summation
 * order of the layer before; the parity test injects these choices into the oracle and checks that every difference IS such a near-tie.) */
LSDM_API int lsdm_train_read(lsdm_train* s, const char* name, void* host_dst, int64_t capacity_bytes, int64_t* nbytes);

#ifdef __cplusplus
}
#endif
#endif /* LSDM_HIP_H */
