// liblsdm_hip: context, parameter folding/packing, workspace and the extern "C" surface (include/lsdm_hip.h).
#include "../../include/lsdm_hip.h"

#include <array>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <type_traits>
#include <vector>

#include "kernels.hpp"

using namespace lsdm;

namespace {

struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
    int64_t numel() const { int64_t n = 1; for (auto s : shape) n *= s; return n; }
};

thread_local std::string g_create_error;

constexpr int kO = 9, kN = 1024;
const int kS[4] = {1024, 256, 64, 16};          // points kept per SA level (pointnet2.py:46-49)
const int kSrcN[4] = {1024, 1024, 256, 64};     // points searched per level
const double kRadius[4] = {0.1, 0.2, 0.4, 0.8};    // python doubles in the reference (pointnet2.py:46-49)
const int kFeatC[4] = {64, 128, 256, 512};

}  // namespace

struct lsdm_ctx {
    int device = 0;
    std::string err;
    std::map<std::string, HostTensor> host;
    bool finalized = false;
    int max_cats = 0;
    std::vector<void*> allocs;  // parameter arena
    // device parameter blocks
    Dense sa[4][3], fp[4][3], head[2];  // fp index 0..3 = fp4, fp3, fp2, fp1
    int prec = LSDM_PREC_F32;           // arithmetic of the dense layers (lsdm_set_precision)
    DenseH sa_h[4][3], fp_h[4][3], head_h[2];   // head_h[1]: conv2 (3 x 128) zero-padded to one 32-channel block  // split-bf16 operands of the same layers (prec != F32)
    DenseH sa_r[3];                     // sa1..sa3 as fragment blobs of the register-chained kernels (encoder_reg.hip)
    // First SA layer split across the grouping (encoder_u.hip, bf16x3): per level i (0: sa3, 1: sa4) the feature columns of
    // the first layer (+ bias) for the row-wise kernel, its xyz columns (C1 x 3, fp32), the fragment blob of layers 2-3, and
    // the per-source-point tensors U = Wf f + Wx x + b, V = Wx x
    DenseH sa_uw[3], sa_ub[3];   // [0] sa3, [1] sa4, [2] sa2
    DenseH fp1_s{};              // fp1 + head as a register-chained blob (encoder_d.hip::fp_c_kernel)
    bool keep_taps = true;       // stage outputs that only feed a fused consumer (fp3) are still stored: off inside multi-step calls
    int fp1_reg = getenv("LSDM_FP1_REG") ? atoi(getenv("LSDM_FP1_REG")) : 1;   // A/B switch
    DenseH den_r{};              // the per-point denoiser as a register-chained blob (denoise_reg.hip)
    int den_reg = getenv("LSDM_DEN_REG") ? atoi(getenv("LSDM_DEN_REG")) : 1;   // A/B switch: 0 = the LDS-tiled denoise_h_kernel
    float* s_stash = nullptr;    // denoise_r_kernel: S of the last step, parked between its two denoiser passes
    // first FP layer split at the concat boundary (encoder_u.hip): columns of points1 (+ bias) / of the interpolated points2;
    // index 0: fp3, 1: fp2
    DenseH fp_a[3], fp_y[3];     // first FP layer split at the concat boundary: [0] fp3, [1] fp2, [2] fp4
    float* fp_ybuf[3] = {};
    int* fp2_plan = nullptr;     // fp2's tile plan (encoder_d.hip::fp_plan_kernel) of a single step, rewritten by every such step
    int* fp2_order = nullptr;    // Morton order of the step's level-1 points (single steps / the strict schedule)
    // item lists of the SA stages of a single step (geometry.hip: sa_pair_kernel; the groups of the loop-invariant schedule have their own in gws)
    // (step_*: by SA level 0 .. 3, one step's lists; the groups' sa3 / sa4 lists are in gws)
    int *step_rows[4] = {}, *step_cnt = nullptr, *step_chunk[4] = {}; int4* step_meta[4] = {};
    std::vector<int> step_host;          // (profiled passes: the step lists' block counts on the host, prof_read_counts)
    std::vector<int> pair_host[2];       // (... and the group's sa3 / sa4 lists')
    int* inv_order = nullptr;    // Morton order of the SOURCE clouds, once per call (loop-invariant schedule)
    int sa_pair = getenv("LSDM_SA_PAIR") ? atoi(getenv("LSDM_SA_PAIR")) : 1;         // A/B switch: 0 = every ball-query group gets a 32-row block of its own; 2 = item lists for single steps of any batch
    int fp1_order = getenv("LSDM_FP1_ORDER") ? atoi(getenv("LSDM_FP1_ORDER")) : 1;   // A/B switch: fp1 + head walks the source points along that order
    // A/B switch: 0 = fp4 / fp3 / fp2 on the LDS-tiled reference form (fp_h_kernel, the whole concatenated first layer per fine point); default: the
    // interpolated operand pre-multiplied at the coarse level, each stage writing the next finer stage's Y as one more layer on its rows
    bool fp_pre = !(getenv("LSDM_FP_PRE") && atoi(getenv("LSDM_FP_PRE")) == 0);
    const float* sa_wx[3] = {};
    float *sa_U[3] = {}, *sa_V[3] = {};
    // A/B switch: 0 = sa2 / sa3 / sa4 on the reference forms with the whole first layer per grouped row (sa_r_kernel, sa_s_kernel, sa_h_kernel); default:
    // the first layer split across the grouping (U / V per source point)
    bool sa_split = !(getenv("LSDM_SA_SPLIT") && atoi(getenv("LSDM_SA_SPLIT")) == 0);
    CondParams cond{};
    PosaParams posa{};
    SceneAttnParams sattn{};
    DenoiseParams den{};
    DenoiseHParams den_h{};
    Lin up0{};
    Dense up2{};
    StepCoefs sched{};
    int sched_T = 0;
    std::vector<void*> sched_allocs;    // the seven tables: owned apart from the parameter arena (lsdm_finalize_params frees that one)
    // workspace
    int cap_B = 0;
    std::vector<void*> ws_allocs;
    std::map<std::string, std::pair<void*, int64_t>> ws_named;
    // Everything a step computes WITHOUT x (conditioning vectors + PointNet++ geometry) lives in a FrontSet.  There are
    // kGroup of them, carved out of contiguous allocations [kGroup][...], so that lsdm_sample_steps can evaluate the
    // geometry of up to kGroup consecutive steps in ONE launch per stage (the fronts do not depend on x or on each other).
#ifndef LSDM_KGROUP
#define LSDM_KGROUP 32
#endif
    static constexpr int kGroup = LSDM_KGROUP;
    // loop-invariant level-1 work of lsdm_sample_steps_invariant: ball query + sa1 feature of every point as a centroid, the
    // 8 nearest points of every point (geometry.hip::l1_remap_kernel turns them into the step's level-1 order)
    int* inv_ball = nullptr;
    void* inv_F1 = nullptr;
    float* inv_nn8_d = nullptr;
    int* inv_nn8_i = nullptr;
    int fps_slots = 4096;               // resident FPS waves on the device (lsdm_create: 4 SIMDs x LSDM_FPS_WPE x CUs)
    int num_cus = 256;
    int group_cap = getenv("LSDM_GROUP") ? atoi(getenv("LSDM_GROUP")) : 0;   // A/B switch: steps per batched geometry launch
    struct FrontSet {
        int* idx[4] = {};
        float* xyz[4] = {};
        int* ball[4] = {};
        int* nn_idx[4] = {};
        float* nn_w[4] = {};
        float *e0 = nullptr, *tr = nullptr, *w = nullptr, *h2p = nullptr;
        float* ctime = nullptr;         // shared-timestep groups: this step's tile of launch_ctext(time_half), else NULL
        int64_t* tbuf = nullptr;
    } fs0;                              // set 0 = base of every [kGroup][...] allocation
    // Set j of a group of steps at batch B.  The batched launches of a group (fill_steps, cond, upsample12, FPS chain, ball
    // queries, 3-NN) write step j right behind step j-1 at the CURRENT batch size (row = step*B + b, cloud = step*B*9 + c),
    // so the per-step views are derived from B at call time -- not from the reserved capacity.
    FrontSet fset(int j, int B) const {
        FrontSet f = fs0;
        if (j == 0) return f;
        static const int S[4] = {1024, 256, 64, 16}, NN[4] = {64, 256, 1024, 1024};
        const size_t nc = (size_t)B * 9, o = (size_t)j;
        const size_t ne = enc_nc > 0 ? (size_t)enc_nc : nc;       // clouds per step of the encoder's batched launches (compacted: the live ones)
        for (int l = 0; l < 4; ++l) {
            f.idx[l] += o * ne * S[l];
            f.xyz[l] += o * ne * S[l] * 3;
            f.ball[l] += o * ne * S[l] * 32;
            f.nn_idx[l] += o * ne * NN[l] * 3;
            f.nn_w[l] += o * ne * NN[l] * 3;
        }
        f.e0 += o * B * 256;
        f.tr += o * nc * 12;
        f.w += o * nc;
        // upsample12_kernel packs one row as 256 x 512 fp32 (f32) or -- the timestep half only -- 128 x 512 bf16 hi + lo (bf16x3) /
        // bf16 hi (bf16); offsets in floats
        f.h2p += o * B * (prec == LSDM_PREC_F32 ? 256 * 512 : (prec == LSDM_PREC_BF16X3 ? 128 * 512 : 64 * 512));
        f.tbuf += o * B;
        return f;
    }
    // Multi-step calls in the bf16 modes: every chain of the batch is at the same timestep, so the conditioning vectors that do
    // not depend on t (text / category embeddings, attention weights, translation tokens: rows of set 0) are evaluated once
    // per call and the timestep half of the up-sampling has ONE row per step (row j of set 0's h2p) -- run_conditioning(shared).
    bool shared_t() const { return prec != LSDM_PREC_F32; }
    FrontSet fset_steps(int j, int B) const {
        FrontSet f = fset(j, B);
        if (!shared_t()) return f;
        f.e0 = fs0.e0; f.tr = fs0.tr; f.w = fs0.w;
        f.h2p = fs0.h2p + (size_t)j * (prec == LSDM_PREC_BF16X3 ? 128 * 512 : 64 * 512);
        f.ctime = ct_time + (size_t)j * 1024 * 128;
        return f;
    }
    float* feat[4] = {};
    float *fpo4 = nullptr, *fpo3 = nullptr, *fpo2 = nullptr, *pcd = nullptr;
    float *hm = nullptr, *o12 = nullptr, *xs = nullptr, *posa_ws = nullptr;
    float *h2x = nullptr, *ctext = nullptr;   // bf16 modes: text half of the up-sampling (per call), see denoise_bf.hip
    float* ct_time = nullptr;                 // bf16 modes: timestep half of combine_extraction's pre-activation, [kGroup][1024][128]
    // optional per-kernel timing with HIP events on the launch stream (bench.py roofline leg)
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;
    std::vector<int> prof_id;   // kernel id of the interval starting at event i (-1 = end-of-phase marker)
    std::vector<int> prof_w;    // steps the launch(es) of the interval cover (a dense stage batched over the steps of a group counts as that many)
    // level features of a whole GROUP of steps (group-batched encoder stages of the loop-invariant bf16x3 schedule): allocated on first use
    // (alt_*: a second set of the per-step scratch tensors of the dense chain -- sa3 / sa4's U and V, fp2's pre-multiplied operand, fp2's
    // output -- for the steps that run on the side stream)
    struct GroupWs { int cap_B = 0; float *feat[4] = {}, *ybuf[3] = {}, *pcd = nullptr, *o12 = nullptr;
                     float* pcd_c = nullptr;   // the encoder's output of the LIVE clouds of every step of a group (degenerate-cloud compaction)
                     struct Alt { float *U[2] = {}, *V[2] = {}, *ybuf1 = nullptr, *fpo2 = nullptr; } alt[3];
                     int* plan = nullptr;     // fp2's tile plans of every step of a group (one fp_plan_kernel launch per group)
                     // item lists of sa3 [0] / sa4 [1] of every step of a group (geometry.hip: sa_pair_kernel): U rows, {V rows, output rows},
                     // items per step, pairs per chunk (scratch)
                     int *pair_rows[2] = {}, *pair_cnt[2] = {}, *pair_chunk[2] = {}; int4* pair_meta[2] = {};
                     // memo of sa2 (geometry.hip: memo_src_kernel): r = 0.2 balls of the source points, flags, F2 rows, work lists, counters
                     int *memo_ball2 = nullptr, *memo_src_list = nullptr, *memo_list = nullptr, *memo_cnt = nullptr, *memo_ident = nullptr;
                     unsigned char *memo_unc = nullptr, *memo_crowd = nullptr; float* memo_F2 = nullptr;
                     int *memo_rows = nullptr, *memo_chunks = nullptr; int4* memo_meta = nullptr;   // item list of the memo's pass (sa_pair_kernel over the uncrowded source points)
                     std::vector<void*> allocs; } gws;
    // side stream of a loop-invariant group: the dense chains of its steps are independent of each other, odd steps run next to even ones
    int streams = getenv("LSDM_STREAMS") ? atoi(getenv("LSDM_STREAMS")) : 2;         // A/B switch: 1 = everything on the caller's stream
    static constexpr int kMaxSide = 3;
    hipStream_t side[kMaxSide] = {};
    hipEvent_t ev_fork = nullptr, ev_join[kMaxSide] = {};
    int nside() const { return streams - 1 < kMaxSide ? streams - 1 : kMaxSide; }
    int batch_enc = getenv("LSDM_BATCH_ENC") ? atoi(getenv("LSDM_BATCH_ENC")) : 1;   // A/B switch: 0 = the dense encoder stages one step at a time
    int sa2_memo = getenv("LSDM_SA2_MEMO") ? atoi(getenv("LSDM_SA2_MEMO")) : 1;      // A/B switch: 0 = every sa2 centroid through the stage every step
    bool memo_ready = false;                  // this call's F2 / flags are in place (set by the per-call block, read by the groups)
    int batch_den = getenv("LSDM_BATCH_DEN") ? atoi(getenv("LSDM_BATCH_DEN")) : 1;   // A/B switch: 0 = scene attention + denoiser one launch per step
    size_t prof_cur = 0;
    double prof_exec[64] = {0};                      // executed MACs per stage id over the profiled window (prof_exec())
    // Degenerate clouds: a cloud whose 1024 points are bit-identical -- the all-zero padding slots of a scene with fewer than 8
    // objects (posa/dataset.py:456) -- has an encoder output that does not depend on the FPS draws (farthest_point_sample returns
    // [start, 0, 0, ...] on it and every ball is the indices 0 .. 31, pointnet2_utils.py:60-104: every grouped row, hence every feature row of
    // every level, is the same vector).  lsdm_sample_steps_invariant evaluates the encoder for ONE representative of each distinct such cloud
    // once per call and runs the per-step encoder on the LIVE clouds only, compacted; a copy kernel spreads both over the (step, chain, slot)
    // layout the scene attention and the denoiser read.  enc_nc: clouds per step of the compacted encoder (0 = all B x 9).
    int deg_fast = getenv("LSDM_DEG_FAST") ? atoi(getenv("LSDM_DEG_FAST")) : 1;      // A/B switch: 0 = every cloud through the encoder every step
    int enc_nc = 0;
    unsigned char* cls_flag = nullptr;  float* cls_first = nullptr;      // per cloud: all points identical; its first point
    void* cls_host = nullptr;                                           // pinned: the flags and first points on the host
    int *enc_map = nullptr, *rep_map = nullptr, *exp_src = nullptr;      // live slot -> cloud; representative -> cloud; cloud -> live slot | -(representative + 1)
    float *enc_clouds = nullptr, *rep_clouds = nullptr, *rep_pcd = nullptr;
    int64_t *rep_starts = nullptr, *enc_starts = nullptr;
    size_t enc_starts_cap = 0;
    int deg_last[3] = {0, 0, 0};                                        // {live, degenerate, distinct degenerate} clouds of the last multi-step call
};

namespace {

int fail(lsdm_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

// Every entry point runs on the context's device and leaves the calling thread's current device as it found it
// (the host framework tracks its own current device).
struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(lsdm_ctx* c) : DeviceGuard(c, c->device) {}
    DeviceGuard(lsdm_ctx* c, int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev == device) { ok = true; prev = -1; return; }
        ok = hipSetDevice(device) == hipSuccess;
        if (!ok) fail(c, LSDM_ERR_HIP, "hipSetDevice(%d) failed", device);
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

#define HIP_TRY(c, expr)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return fail(c, LSDM_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

enum KernelId { K_COND = 0, K_UP12, K_POSA, K_FPS, K_BALL1, K_BALL2, K_BALL3, K_BALL4, K_SA1, K_SA2, K_SA3, K_SA4, K_NN4, K_NN3, K_NN2,
                K_NN1, K_FP4, K_FP3, K_FP2, K_FP1, K_SATTN, K_DENOISE, K_REMAP, K_COUNT };
const char* const kKernelNames[K_COUNT] = {"cond_kernel", "upsample12_kernel", "posa_kernel", "fps_chain_kernel", "ball_query<1024>/sa1",
    "ball_query<1024>/sa2", "ball_query<256>/sa3", "ball_query<64>/sa4", "sa_kernel<3,32,32,64>", "sa_kernel<64,64,64,128>",
    "sa_kernel<128,128,128,256>", "sa_kernel<256,256,256,512>", "three_nn<16>/fp4", "three_nn<64>/fp3", "three_nn<256>/fp2",
    "three_nn<1024>/fp1", "fp_kernel<fp4>", "fp_kernel<fp3>", "fp_kernel<fp2>", "fp_kernel<fp1+head>", "scene_attn_kernel",
    "denoise_kernel", "l1_remap_kernel"};
// algorithmic MACs of one launch per cloud (encoder kernels) or per sample (others); SURVEY.md §8(d), DESIGN.md
const double kKernelMacs[K_COUNT] = {0.24e6, 16.8e6, 7.2e6, 0, 0, 0, 0, 0,
    32768.0 * (6 * 32 + 32 * 32 + 32 * 64), 8192.0 * (67 * 64 + 64 * 64 + 64 * 128), 2048.0 * (131 * 128 + 128 * 128 + 128 * 256),
    512.0 * (259 * 256 + 256 * 256 + 256 * 512), 0, 0, 0, 0, 64.0 * (768 * 256 + 256 * 256), 256.0 * (384 * 256 + 256 * 256),
    1024.0 * (320 * 256 + 256 * 128), 1024.0 * (3 * 128 * 128 + 128 * 128 + 128 * 3), 0.0009e9 / 9,
    1024.0 * (512 * 256 + 256 * 128 + 2 * (3 * 64 + 64 * 128 + 256 * 192 + 192 * 128 + 128 * 64 + 64 * 3)), 0};

void prof_mark(lsdm_ctx* c, int id, hipStream_t st, int weight = 1) {
    if (!c->prof_on) return;
    if (c->prof_cur >= c->prof_ev.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) { c->prof_on = false; return; }
        c->prof_ev.push_back(e);
        c->prof_id.push_back(id);
        c->prof_w.push_back(weight);
    }
    c->prof_id[c->prof_cur] = id;
    c->prof_w[c->prof_cur] = weight;
    (void)hipEventRecord(c->prof_ev[c->prof_cur++], st);
}

// Executed-work accounting of the profiled window (bench.py "executed_gflop"): the MACs the launches of a stage ISSUE to the matrix pipe,
// counted as products (MFMA MACs / 3 in bf16x3) -- K padded to the 16-wide k-step of the bf16 cores, the 3-channel output layers padded to one
// 32-channel block, first SA / FP layers split (encoder_u.hip), work skipped by the loop-invariant schedule not counted.  Cross-checked
// against SQ_INSTS_MFMA x 16384 / 3 of the PMC pass (profiles/).  Stages without an explicit figure report their algorithmic MACs.
void prof_exec(lsdm_ctx* c, int id, double macs) {
    if (c->prof_on && id >= 0 && id < 64) c->prof_exec[id] += macs;
}
// ... with a count that lives on the device (the sa2 memo's work lists): macs_per_count x *counter, read back right away (the profiled
// pass keeps every stage on one stream and is not the timed region)
void prof_exec_dev(lsdm_ctx* c, int id, const int* counter, double macs_per_count, hipStream_t st) {
    if (!c->prof_on) return;
    int v = 0;
    if (hipMemcpyAsync(&v, counter, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return;
    prof_exec(c, id, macs_per_count * v);
}
// ... for the item lists of sa3 / sa4, whose counts are written by the planner BEFORE the stages run: read them once, inside an interval that
// belongs to no stage (the host round trip must not sit in a stage's event interval), and account per step from the host copy
void prof_read_counts(lsdm_ctx* c, const int* dev, int n, std::vector<int>& host, hipStream_t st) {
    host.assign((size_t)n, 0);
    if (!c->prof_on) return;
    prof_mark(c, -1, st);
    if (hipMemcpyAsync(host.data(), dev, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        host.assign((size_t)n, 0);
}
constexpr long long kItemRowLimit = 1ll << 24;      // output rows of an item list are 24-bit (sa_pair_kernel: out row A | k << 24)
namespace xm {     // per cloud (encoder) / per chain or row (others), split-bf16 kernels in their default configuration
constexpr double SA1 = 32768.0 * (16 * 32 + 32 * 32 + 32 * 64);
constexpr double SA2_UV = 1024.0 * 64 * 64, SA2_CENT = 32.0 * (64 * 64 + 64 * 128), SA2_FULL = 8192.0 * (80 * 64 + 64 * 64 + 64 * 128);
constexpr double SA1_ITEM = 32.0 * (16 * 32 + 32 * 32 + 32 * 64);
constexpr double SA3_UV = 256.0 * 128 * 128, SA3_ITEM = 32.0 * (128 * 128 + 128 * 256), SA4_UV = 64.0 * 256 * 256, SA4_ITEM = 32.0 * (256 * 256 + 256 * 512);
constexpr double SA3 = 256.0 * 128 * 128 + 2048.0 * (128 * 128 + 128 * 256), SA3_FULL = 2048.0 * (144 * 128 + 128 * 128 + 128 * 256);
constexpr double SA4 = 64.0 * 256 * 256 + 512.0 * (256 * 256 + 256 * 512), SA4_FULL = 512.0 * (272 * 256 + 256 * 256 + 256 * 512);
constexpr double FP4_PRE = 16.0 * 512 * 256 + 64.0 * 3 * 256 * 256, FP4_FULL = 64.0 * (768 * 256 + 256 * 256);
constexpr double FP3_PRE = 256.0 * (128 * 256 + 256 * 256), FP3_NEXT_Y = 256.0 * 256 * 256, FP3_FULL = 256.0 * (384 * 256 + 256 * 256);
constexpr double FP2_PRE = 1024.0 * (64 * 256 + 256 * 128), FP2_FULL = 1024.0 * (320 * 256 + 256 * 128);
constexpr double FP1 = 1024.0 * (4 * 128 * 128 + 128 * 32);
constexpr double DEN_PASS = 32.0 * 132 * 16384, DEN_S = 32.0 * 48 * 16384;   // denoise_reg.hip RL<>: products per 32-point block x 32 blocks per chain
constexpr double UP12_HALF = 128.0 * 128 * 512, CTEXT_HALF = 1024.0 * 128 * 512 + 1024.0 * 128 * 128;   // per row (sample or step), one half of the 256 rows
}  // namespace xm

int check_launch(lsdm_ctx* c, const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(c, LSDM_ERR_HIP, "launch %s: %s", what, hipGetErrorString(e));
    return LSDM_OK;
}

const HostTensor* find(lsdm_ctx* c, const std::string& name, std::initializer_list<int64_t> shape) {
    auto it = c->host.find(name);
    if (it == c->host.end()) { fail(c, LSDM_ERR_PARAM, "missing parameter %s", name.c_str()); return nullptr; }
    int64_t want = 1; for (auto s : shape) want *= s;
    if (it->second.numel() != want) {
        fail(c, LSDM_ERR_PARAM, "parameter %s has %lld elements, expected %lld", name.c_str(), (long long)it->second.numel(), (long long)want);
        return nullptr;
    }
    return &it->second;
}

float* upload(lsdm_ctx* c, const float* src, size_t n, std::vector<void*>* owner = nullptr) {
    void* d = nullptr;
    if (hipMalloc(&d, n * sizeof(float)) != hipSuccess) { fail(c, LSDM_ERR_HIP, "hipMalloc(%zu)", n * sizeof(float)); return nullptr; }
    (owner ? *owner : c->allocs).push_back(d);
    if (hipMemcpy(d, src, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { fail(c, LSDM_ERR_HIP, "hipMemcpy H2D"); return nullptr; }
    return static_cast<float*>(d);
}

// plain Linear: weight (N,K) as stored
bool load_lin(lsdm_ctx* c, const std::string& prefix, int N, int K, Lin* out) {
    const HostTensor* w = find(c, prefix + ".weight", {N, K});
    const HostTensor* b = find(c, prefix + ".bias", {N});
    if (!w || !b) return false;
    out->w = upload(c, w->data.data(), w->data.size());
    out->b = upload(c, b->data.data(), b->data.size());
    return out->w && out->b;
}

// MFMA B-operand layout: Wp[kb][n][j] = W[n][8*kb + j], K zero-padded to a multiple of 8 (mlp_core.hpp)
std::vector<float> pack_weight(const std::vector<float>& W, int N, int K) {
    const int Kp = (K + 7) / 8 * 8;
    std::vector<float> p((size_t)Kp * N, 0.0f);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) p[((size_t)(k / 8) * N + n) * 8 + (k % 8)] = W[(size_t)n * K + k];
    return p;
}

uint16_t f2bf(float f) {  // round-to-nearest-even fp32 -> bf16 (what v_cvt_pk_bf16_f32 does)
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
float bf2f(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// split-bf16 MFMA operand layout (bf16_core.hpp): Wp[kb][n][np][16], k = 16*kb + j; column k of the packed matrix is
// column perm[k] of W (perm empty = identity); K zero-padded to a multiple of 16.  np = 2: hi and lo planes.
std::vector<uint16_t> pack_weight_h(const std::vector<float>& W, int N, int K, int np, const std::vector<int>& perm) {
    const int Kp = (K + 15) / 16 * 16;
    std::vector<uint16_t> p((size_t)Kp * N * np, 0);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            const float v = W[(size_t)n * K + (perm.empty() ? k : perm[k])];
            const uint16_t hi = f2bf(v);
            const size_t at = ((size_t)(k / 16) * N + n) * np * 16 + (k % 16);
            p[at] = hi;
            if (np == 2) p[at + 16] = f2bf(v - bf2f(hi));
        }
    return p;
}

// (conv|linear) [+ eval BatchNorm folded: y = (Wx + b - mean) * gamma / sqrt(var + eps) + beta] -> W (N,K), bias (N)
bool fold_dense(lsdm_ctx* c, const std::string& wname, const std::string& bname, const std::string& bn, int N, int K,
                std::vector<float>& W, std::vector<float>& Bv) {
    const HostTensor* w = find(c, wname, {N, K});
    const HostTensor* b = find(c, bname, {N});
    if (!w || !b) return false;
    W = w->data;
    Bv = b->data;
    if (!bn.empty()) {
        const HostTensor* g = find(c, bn + ".weight", {N});
        const HostTensor* be = find(c, bn + ".bias", {N});
        const HostTensor* mu = find(c, bn + ".running_mean", {N});
        const HostTensor* var = find(c, bn + ".running_var", {N});
        if (!g || !be || !mu || !var) return false;
        for (int n = 0; n < N; ++n) {
            const double s = (double)g->data[n] / std::sqrt((double)var->data[n] + 1e-5);
            for (int k = 0; k < K; ++k) W[(size_t)n * K + k] = (float)((double)W[(size_t)n * K + k] * s);
            Bv[n] = (float)(((double)Bv[n] - (double)mu->data[n]) * s + (double)be->data[n]);
        }
    }
    return true;
}

// Fragment blob of a register-chained layer stack (encoder_reg.hip): every layer as
// [nb = n/32][k-step s][plane][lane = n%32 + 32*h][8] bf16 with k-slot (s, h, j) -> input channel
//   first layer : first_map[16*s + 8*h + j]  (column of W in the kernel's own K order; -1 / beyond the map = zero pad)
//   later layers: 32*(s/2) + 8*(2*(s%2) + j/4) + 4*h + j%4  -- the channels a lane owns in accumulator registers
//                 8*(s%2) .. 8*(s%2)+7 of block s/2 of the previous layer's transposed product
// followed (separately) by the fp32 biases of the layers back to back, BatchNorm folded.
struct ChainLayer { std::string w, b, bn; int N, K; int padN = 0; };   // padN: rows zero-padded to that many channels
bool build_chain_blob(lsdm_ctx* c, const std::vector<ChainLayer>& layers, const std::vector<int>& first_map, int np, DenseH* out) {
    std::vector<uint16_t> blob;
    std::vector<float> biases;
    for (size_t li = 0; li < layers.size(); ++li) {
        int N = layers[li].N;
        const int K = layers[li].K;
        std::vector<float> W, Bv;
        if (!fold_dense(c, layers[li].w, layers[li].b, layers[li].bn, N, K, W, Bv)) return false;
        if (layers[li].padN > N) {
            N = layers[li].padN;
            W.resize((size_t)N * K, 0.0f);
            Bv.resize(N, 0.0f);
        }
        const int KS = li == 0 ? ((int)first_map.size() + 15) / 16 : K / 16;
        for (int nb = 0; nb < N / 32; ++nb)
            for (int s = 0; s < KS; ++s)
                for (int pl = 0; pl < np; ++pl)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int n = nb * 32 + (lane & 31), h = lane >> 5;
                            float v = 0.0f;
                            if (li == 0) {
                                const int k = 16 * s + 8 * h + j;
                                if (k < (int)first_map.size() && first_map[k] >= 0) v = W[(size_t)n * K + first_map[k]];
                            } else {
                                v = W[(size_t)n * K + 32 * (s / 2) + 8 * (2 * (s % 2) + j / 4) + 4 * h + j % 4];
                            }
                            const uint16_t hi = f2bf(v);
                            blob.push_back(pl == 0 ? hi : f2bf(v - bf2f(hi)));
                        }
        biases.insert(biases.end(), Bv.begin(), Bv.end());
    }
    if (blob.size() % 2) blob.push_back(0);
    out->w = reinterpret_cast<const __bf16*>(upload(c, reinterpret_cast<const float*>(blob.data()), blob.size() / 2));
    out->b = upload(c, biases.data(), biases.size());
    return out->w && out->b;
}

// Fragment blob of denoise_reg.hip::denoise_r_kernel, fragments [nb][k-step s][plane][lane][8] per layer, layers back to back:
//   S  = combination_extraction.0[:, 128:256] (6 x 8), pose_embedding.2 (4 x 4)           -- k-slot (s, h, j) = column 16 s + 8 h + j
//   combination_extraction.0[:, 0:128] (6 x 8), combination_extraction.2 (4 x 12), pose_final.0 (2 x 8), pose_final.2 zero-padded
//   to 32 rows (1 x 4)                                                                       -- accumulator order of the previous layer
// biases (fp32): [cex0 192 | pose2 128 | cex2 128 | fin0 64 | combine_extraction.0 128 | pose_embedding.0 as (w0, w1, w2, b) x 64 | fin2 3 + pad]
bool build_denoise_reg_blob(lsdm_ctx* c, int np, DenseH* out) {
    std::vector<float> Wc0, Bc0, Wp2, Bp2, Wc2, Bc2, Wf0, Bf0, Wf2, Bf2, Wcomb, Bcomb, Wp0, Bp0;
    if (!fold_dense(c, "input_process.combination_extraction.0.weight", "input_process.combination_extraction.0.bias", "", 192, 256, Wc0, Bc0) ||
        !fold_dense(c, "input_process.pose_embedding.2.weight", "input_process.pose_embedding.2.bias", "", 128, 64, Wp2, Bp2) ||
        !fold_dense(c, "input_process.combination_extraction.2.weight", "input_process.combination_extraction.2.bias", "", 128, 192, Wc2, Bc2) ||
        !fold_dense(c, "output_process.pose_final.0.weight", "output_process.pose_final.0.bias", "", 64, 128, Wf0, Bf0) ||
        !fold_dense(c, "output_process.pose_final.2.weight", "output_process.pose_final.2.bias", "", 3, 64, Wf2, Bf2) ||
        !fold_dense(c, "combine_extraction.0.weight", "combine_extraction.0.bias", "", 128, 256, Wcomb, Bcomb) ||
        !fold_dense(c, "input_process.pose_embedding.0.weight", "input_process.pose_embedding.0.bias", "", 64, 3, Wp0, Bp0))
        return false;
    Wf2.resize((size_t)32 * 64, 0.0f);
    std::vector<uint16_t> blob;
    // one layer: NB channel blocks x KS k-steps; W is (N, ldw) row-major, the layer reads columns col0 .. col0 + 16 KS - 1
    auto layer = [&](const std::vector<float>& W, int ldw, int col0, int NB, int KS, bool first) {
        for (int nb = 0; nb < NB; ++nb)
            for (int s = 0; s < KS; ++s)
                for (int pl = 0; pl < np; ++pl)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int n = nb * 32 + (lane & 31), h = lane >> 5;
                            const int k = first ? 16 * s + 8 * h + j : 32 * (s / 2) + 8 * (2 * (s % 2) + j / 4) + 4 * h + j % 4;
                            const float v = W[(size_t)n * ldw + col0 + k];
                            const uint16_t hi = f2bf(v);
                            blob.push_back(pl == 0 ? hi : f2bf(v - bf2f(hi)));
                        }
    };
    layer(Wc0, 256, 128, 6, 8, true);
    layer(Wp2, 64, 0, 4, 4, true);
    layer(Wc0, 256, 0, 6, 8, false);
    layer(Wc2, 192, 0, 4, 12, false);
    layer(Wf0, 128, 0, 2, 8, false);
    layer(Wf2, 64, 0, 1, 4, false);
    std::vector<float> biases;
    biases.insert(biases.end(), Bc0.begin(), Bc0.end());
    biases.insert(biases.end(), Bp2.begin(), Bp2.end());
    biases.insert(biases.end(), Bc2.begin(), Bc2.end());
    biases.insert(biases.end(), Bf0.begin(), Bf0.end());
    biases.insert(biases.end(), Bcomb.begin(), Bcomb.end());
    for (int n = 0; n < 64; ++n) { biases.push_back(Wp0[n * 3]); biases.push_back(Wp0[n * 3 + 1]); biases.push_back(Wp0[n * 3 + 2]); biases.push_back(Bp0[n]); }
    for (int n = 0; n < 3; ++n) biases.push_back(Bf2[n]);
    biases.push_back(0.0f);
    if (blob.size() % 2) blob.push_back(0);
    out->w = reinterpret_cast<const __bf16*>(upload(c, reinterpret_cast<const float*>(blob.data()), blob.size() / 2));
    out->b = upload(c, biases.data(), biases.size());
    return out->w && out->b;
}

// SA level: three layers; the kernels order the first layer's K as [features (D) || xyz - centroid (3) || 0], the
// reference as [xyz - centroid (3) || features (D)]
bool build_sa_reg_blob(lsdm_ctx* c, const std::string& prefix, int D, const int (&mlp)[3], int np, DenseH* out) {
    std::vector<ChainLayer> layers;
    int K = D + 3;
    for (int li = 0; li < 3; ++li) {
        layers.push_back({prefix + ".mlp_convs." + std::to_string(li) + ".weight", prefix + ".mlp_convs." + std::to_string(li) + ".bias",
                          prefix + ".mlp_bns." + std::to_string(li), mlp[li], K});
        K = mlp[li];
    }
    std::vector<int> map;
    for (int k = 0; k < D; ++k) map.push_back(3 + k);
    for (int k = 0; k < 3; ++k) map.push_back(k);
    return build_chain_blob(c, layers, map, np, out);
}

bool load_dense_h(lsdm_ctx* c, const std::string& wname, const std::string& bname, const std::string& bn, int N, int K, int np,
                  const std::vector<int>& perm, DenseH* out) {
    std::vector<float> W, Bv;
    if (!fold_dense(c, wname, bname, bn, N, K, W, Bv)) return false;
    std::vector<uint16_t> p = pack_weight_h(W, N, K, np, perm);
    static_assert(sizeof(float) == 2 * sizeof(uint16_t), "upload granularity");
    if (p.size() % 2) p.push_back(0);
    out->w = reinterpret_cast<const __bf16*>(upload(c, reinterpret_cast<const float*>(p.data()), p.size() / 2));
    out->b = upload(c, Bv.data(), Bv.size());
    return out->w && out->b;
}

// (conv|linear) [+ eval BatchNorm] -> packed Dense of the f32 MFMA core
bool load_dense(lsdm_ctx* c, const std::string& wname, const std::string& bname, const std::string& bn, int N, int K, Dense* out) {
    const HostTensor* w = find(c, wname, {N, K});
    const HostTensor* b = find(c, bname, {N});
    if (!w || !b) return false;
    std::vector<float> W = w->data, Bv = b->data;
    if (!bn.empty()) {
        const HostTensor* g = find(c, bn + ".weight", {N});
        const HostTensor* be = find(c, bn + ".bias", {N});
        const HostTensor* mu = find(c, bn + ".running_mean", {N});
        const HostTensor* var = find(c, bn + ".running_var", {N});
        if (!g || !be || !mu || !var) return false;
        for (int n = 0; n < N; ++n) {
            const double s = (double)g->data[n] / std::sqrt((double)var->data[n] + 1e-5);
            for (int k = 0; k < K; ++k) W[(size_t)n * K + k] = (float)((double)W[(size_t)n * K + k] * s);
            Bv[n] = (float)(((double)Bv[n] - (double)mu->data[n]) * s + (double)be->data[n]);
        }
    }
    std::vector<float> p = pack_weight(W, N, K);
    out->w = upload(c, p.data(), p.size());
    out->b = upload(c, Bv.data(), Bv.size());
    return out->w && out->b;
}

template <typename T>
T* ws_alloc(lsdm_ctx* c, const char* name, size_t count) {
    void* d = nullptr;
    if (hipMalloc(&d, count * sizeof(T)) != hipSuccess) { fail(c, LSDM_ERR_HIP, "hipMalloc workspace %s (%zu bytes)", name, count * sizeof(T)); return nullptr; }
    c->ws_allocs.push_back(d);
    (void)hipMemset(d, 0, count * sizeof(T));        // (set-up time: every page touched once)
    c->ws_named[name] = {d, (int64_t)count};
    return static_cast<T*>(d);
}

// timesteps of g consecutive steps, step-major: p[j*B + b] = t0 - j
__global__ void fill_steps_kernel(int64_t* p, int64_t t0, int B, int g) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B * g) p[i] = t0 - i / B;
}

// ---- degenerate clouds (lsdm_ctx::deg_fast) -------------------------------------------------------------------------------------------
// one work-group per cloud: flag[c] = every point BIT-identical to point 0 (a +0 / -0 mix is not: sign bits reach the features); first[c] = point 0
__global__ __launch_bounds__(256) void cloud_class_kernel(const float* __restrict__ clouds, unsigned char* __restrict__ flag, float* __restrict__ first) {
    const unsigned* p = reinterpret_cast<const unsigned*>(clouds) + (size_t)blockIdx.x * kN * 3;
    const unsigned x0 = p[0], y0 = p[1], z0 = p[2];
    bool same = true;
    for (int i = threadIdx.x; i < kN; i += 256) same = same && p[i * 3] == x0 && p[i * 3 + 1] == y0 && p[i * 3 + 2] == z0;
    const int all = __syncthreads_and(same ? 1 : 0);
    if (threadIdx.x == 0) {
        flag[blockIdx.x] = (unsigned char)all;
        first[blockIdx.x * 3 + 0] = __uint_as_float(x0); first[blockIdx.x * 3 + 1] = __uint_as_float(y0); first[blockIdx.x * 3 + 2] = __uint_as_float(z0);
    }
}
// dst cloud i = src cloud map[i]
__global__ __launch_bounds__(256) void gather_clouds_kernel(const float* __restrict__ src, const int* __restrict__ map, float* __restrict__ dst) {
    const f32x4* s4 = reinterpret_cast<const f32x4*>(src + (size_t)map[blockIdx.x] * kN * 3);
    f32x4* d4 = reinterpret_cast<f32x4*>(dst + (size_t)blockIdx.x * kN * 3);
    for (int i = threadIdx.x; i < kN * 3 / 4; i += 256) d4[i] = s4[i];
}
// the start draws (rows, nc) of the clouds map[0 .. n): dst (rows, n)
__global__ void gather_starts_kernel(const int64_t* __restrict__ src, const int* __restrict__ map, int nc, int n, long long total, int64_t* __restrict__ dst) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) dst[i] = src[(i / n) * nc + map[i % n]];
}
// out (steps, nc, 1024, 3): cloud c of step j = the compacted encoder's slot exp_src[c] of that step, or representative -(exp_src[c] + 1) of the call
__global__ __launch_bounds__(256) void expand_pcd_kernel(const float* __restrict__ live, const float* __restrict__ rep, const int* __restrict__ exp_src, int nc,
                                                         int nlive, float* __restrict__ out) {
    const int c = blockIdx.x, j = blockIdx.y;
    const int e = exp_src[c];
    const f32x4* s4 = reinterpret_cast<const f32x4*>(e >= 0 ? live + ((size_t)j * nlive + e) * kN * 3 : rep + (size_t)(-e - 1) * kN * 3);
    f32x4* d4 = reinterpret_cast<f32x4*>(out + ((size_t)j * nc + c) * kN * 3);
    for (int i = threadIdx.x; i < kN * 3 / 4; i += 256) d4[i] = s4[i];
}

// geometry + dense stages of get_backbone.forward on `num_clouds` clouds
typedef lsdm_ctx::FrontSet FrontSet;

// PointNet++ geometry of `g` consecutive steps x `nc` clouds into front sets f, f+1, ..., f+g-1 (contiguous memory):
// FPS chain, 4 ball queries, 4 three-NN searches -- one launch each for the whole group (x-independent work).
// `starts` is the (g,4,nc) slice of the per-step start draws.
// skip_sa2_ball: the level-2 ball query is left to the caller (sa2 memo: it runs for the crowded centroids only, after their classification)
void run_geometry(lsdm_ctx* c, FrontSet& f, const float* clouds, const int64_t* starts, int nc, int g, hipStream_t st, bool skip_l1 = false,
                  bool skip_sa2_ball = false) {
    const int tot = nc * g;
    prof_mark(c, K_FPS, st);
    launch_fps_chain(clouds, starts, tot, nc, f.idx, f.xyz, st);
    const float* lvl_xyz[5] = {clouds, f.xyz[0], f.xyz[1], f.xyz[2], f.xyz[3]};
    for (int l = 0; l < 4; ++l) {
        // radius ** 2 is a double in the reference and is compared against fp32 distances (Q13)
        const float r2 = (float)(kRadius[l] * kRadius[l]);
        if (l == 0 && skip_l1) continue;   // level-1 balls are step-invariant (lsdm_sample_steps_invariant)
        if (l == 1 && skip_sa2_ball) continue;
        prof_mark(c, K_BALL1 + l, st);
        launch_ball_query(lvl_xyz[l], lvl_xyz[l + 1], tot, kSrcN[l], kS[l], r2, f.ball[l], l == 0 ? nc : 0, st);
    }
    // fp4: l3 <- l4 ; fp3: l2 <- l3 ; fp2: l1 <- l2 ; fp1: l0 <- l1
    prof_mark(c, K_NN4, st);
    launch_three_nn(f.xyz[2], f.xyz[3], tot, 64, 16, f.nn_idx[0], f.nn_w[0], 0, st);
    prof_mark(c, K_NN3, st);
    launch_three_nn(f.xyz[1], f.xyz[2], tot, 256, 64, f.nn_idx[1], f.nn_w[1], 0, st);
    prof_mark(c, K_NN2, st);
    launch_three_nn(f.xyz[0], f.xyz[1], tot, 1024, 256, f.nn_idx[2], f.nn_w[2], 0, st);
    if (!skip_l1) {
        prof_mark(c, K_NN1, st);
        launch_three_nn(clouds, f.xyz[0], tot, 1024, 1024, f.nn_idx[3], f.nn_w[3], nc, st);
    }
    prof_mark(c, -1, st);
}

// dense stages of get_backbone.forward (4 x SA, 4 x FP + head) on the geometry of a front set
// skip_sa1 (loop-invariant schedule): level 1 comes from the call's invariants.  In bf16x3 with the split sa2 / fp2 layers the
// invariants stay in SOURCE order (inv_F1, sa_U[2] / sa_V[2]) and are read through the step's FPS order f.idx[0]; otherwise
// l1_remap_kernel has re-ordered the feature rows into feat[0].
bool inv_in_source_order(const lsdm_ctx* c) { return c->prec != LSDM_PREC_F32 && c->sa_split && c->fp_pre; }

void run_encoder_dense(lsdm_ctx* c, const FrontSet& f, const float* clouds, int nc, float* out, hipStream_t st, bool skip_sa1 = false) {
    const float* lvl_xyz[5] = {clouds, f.xyz[0], f.xyz[1], f.xyz[2], f.xyz[3]};
    if (c->prec == LSDM_PREC_F32) {
        const float* lvl_feat[5] = {clouds, c->feat[0], c->feat[1], c->feat[2], c->feat[3]};
        for (int l = skip_sa1 ? 1 : 0; l < 4; ++l) {
            prof_mark(c, K_SA1 + l, st);
            launch_sa(l, nc, lvl_xyz[l], lvl_feat[l], lvl_xyz[l + 1], f.ball[l], c->sa[l], c->feat[l], st);
            prof_exec(c, K_SA1 + l, nc * kKernelMacs[K_SA1 + l]);
        }
        for (int k = K_FP4; k <= K_FP1; ++k) prof_exec(c, k, nc * kKernelMacs[k]);
        prof_mark(c, K_FP4, st);
        launch_fp(4, nc, c->feat[2], c->feat[3], f.nn_idx[0], f.nn_w[0], c->fp[0], nullptr, c->fpo4, st);
        prof_mark(c, K_FP3, st);
        launch_fp(3, nc, c->feat[1], c->fpo4, f.nn_idx[1], f.nn_w[1], c->fp[1], nullptr, c->fpo3, st);
        prof_mark(c, K_FP2, st);
        launch_fp(2, nc, c->feat[0], c->fpo3, f.nn_idx[2], f.nn_w[2], c->fp[2], nullptr, c->fpo2, st);
        prof_mark(c, K_FP1, st);
        launch_fp(1, nc, nullptr, c->fpo2, f.nn_idx[3], f.nn_w[3], c->fp[3], c->head, out, st);
        return;
    }
    // split-bf16 core (np = 2: bf16x3, np = 1: plain bf16): level features, U / V and Y are fp32 tensors in both
    const int np = c->prec == LSDM_PREC_BF16X3 ? 2 : 1;
    float* const* feat = c->feat;
    const void* lvl_feat[5] = {clouds, feat[0], feat[1], feat[2], feat[3]};
    // the step's item lists (geometry only): two ball-query groups whose distinct neighbours fit into 32 rows together share a block
    const bool src1 = skip_sa1 && inv_in_source_order(c);      // sa2 reads the per-call U / V through the step's level-1 order
    // (a list's output rows carry the split in their top byte: levels of 2^24 centroids and more -- sa1 from 16 384 clouds -- keep one block per group)
    const bool fits1 = (long long)nc * 1024 < kItemRowLimit;
    // (a single step pays two planner launches for its lists: measured break-even around 4 chains -- strict schedule 2262 vs 2269 steps/s at
    // B = 4, 1790 vs 1705 at B = 8, 1177 vs 1099 at B = 16, 400 vs 361 at B = 64; LSDM_SA_PAIR=2 forces them at any size)
    const bool on = c->sa_pair && (nc >= 54 || c->sa_pair > 1);
    const bool items[4] = {on && !skip_sa1 && fits1, on && c->sa_split && !src1, on && c->sa_split, on && c->sa_split};
    {
        const int S_[4] = {1024, 256, 64, 16}, N_[4] = {1024, 1024, 256, 64};
        PairLevel lv[4];
        int nlv = 0, firstl = -1;
        for (int l = 0; l < 4; ++l) {
            if (!items[l]) continue;
            if (firstl < 0) firstl = l;
            lv[nlv++] = PairLevel{f.ball[l], f.idx[l], nc * S_[l], S_[l], N_[l], 0, 1, c->step_chunk[l], c->step_rows[l], c->step_meta[l], c->step_cnt + l, nullptr, 0};
        }
        if (nlv) {
            prof_mark(c, K_SA1 + firstl, st, 0);      // (all levels in one pair of launches, timed with the first of them)
            launch_sa_pair_plans(lv, nlv, st);
        }
        if (items[0] || items[1] || items[2] || items[3]) prof_read_counts(c, c->step_cnt, 4, c->step_host, st);
        else c->step_host.assign(4, 0);
    }
    for (int l = skip_sa1 ? 1 : 0; l < 4; ++l) {
        prof_mark(c, K_SA1 + l, st);
        if (l == 0 && items[0]) {
            launch_sa_r_items(clouds, c->step_rows[0], c->step_meta[0], c->step_cnt, c->sa_r[0].w, c->sa_r[0].b, c->feat[0], st);
            prof_exec(c, K_SA1, c->step_host[0] * xm::SA1_ITEM);
        } else if (l == 1 && c->sa_split) {
            // sa2: U, V per level-1 point -- per step in FPS order, or (loop-invariant schedule) once per call in source order
            const bool src = src1;
            if (!src) launch_lin_u(np, 1, nc * 1024, c->feat[0], lvl_xyz[1], c->sa_uw[2], c->sa_wx[2], c->sa_U[2], c->sa_V[2], st);
            if (items[1]) {
                launch_sa_u_items(np, 1, c->sa_U[2], c->sa_V[2], c->step_rows[1], c->step_meta[1], c->step_cnt + 1, c->sa_ub[2].w, c->sa_ub[2].b, c->feat[1], st);
                prof_exec(c, K_SA1 + 1, nc * xm::SA2_UV + c->step_host[1] * xm::SA2_CENT);
            } else {
                launch_sa_u(np, 1, nc, c->sa_U[2], c->sa_V[2], f.idx[1], f.ball[1], src ? f.idx[0] : nullptr, c->sa_ub[2].w, c->sa_ub[2].b, c->feat[1], st);
                prof_exec(c, K_SA1 + 1, nc * ((src ? 0.0 : xm::SA2_UV) + 256 * xm::SA2_CENT));
            }
        } else if (l == 2 && c->sa_split) {
            // first layer split across the grouping: U, V per level-2 point, then the two remaining layers on ReLU(U_j - V_i)
            launch_lin_u(np, 2, nc * 256, c->feat[1], lvl_xyz[2], c->sa_uw[0], c->sa_wx[0], c->sa_U[0], c->sa_V[0], st);
            if (items[2]) {
                launch_sa_u_items(np, 2, c->sa_U[0], c->sa_V[0], c->step_rows[2], c->step_meta[2], c->step_cnt + 2, c->sa_ub[0].w, c->sa_ub[0].b, c->feat[2], st);
                prof_exec(c, K_SA1 + 2, nc * xm::SA3_UV + c->step_host[2] * xm::SA3_ITEM);
            } else {
                launch_sa_u(np, 2, nc, c->sa_U[0], c->sa_V[0], f.idx[2], f.ball[2], nullptr, c->sa_ub[0].w, c->sa_ub[0].b, c->feat[2], st);
                prof_exec(c, K_SA1 + 2, nc * xm::SA3);
            }
        } else if (l == 3 && c->sa_split) {
            launch_lin_u(np, 3, nc * 64, c->feat[2], lvl_xyz[3], c->sa_uw[1], c->sa_wx[1], c->sa_U[1], c->sa_V[1], st);
            if (items[3]) {
                launch_sa_u_items(np, 3, c->sa_U[1], c->sa_V[1], c->step_rows[3], c->step_meta[3], c->step_cnt + 3, c->sa_ub[1].w, c->sa_ub[1].b, c->feat[3], st);
                prof_exec(c, K_SA1 + 3, nc * xm::SA4_UV + c->step_host[3] * xm::SA4_ITEM);
            } else {
                launch_sa_u(np, 3, nc, c->sa_U[1], c->sa_V[1], f.idx[3], f.ball[3], nullptr, c->sa_ub[1].w, c->sa_ub[1].b, c->feat[3], st);
                prof_exec(c, K_SA1 + 3, nc * xm::SA4);
            }
        } else {
            // the reference forms of the bf16x3 build (LSDM_SA_SPLIT=0: the whole stack per grouped row -- register-chained sa1 / sa2, streamed sa3,
            // LDS-tiled sa4), and sa1 of a small batch (two planes in every mode)
            if (l < 3)
                launch_sa_r(2, l, nc, lvl_xyz[l], lvl_feat[l], lvl_xyz[l + 1], f.ball[l], c->sa_r[l].w, c->sa_r[l].b, feat[l], st);
            else
                launch_sa_h(2, l, nc, lvl_xyz[l], lvl_feat[l], lvl_xyz[l + 1], f.ball[l], c->sa_h[l], feat[l], st);
            const double full[4] = {xm::SA1, xm::SA2_FULL, xm::SA3_FULL, xm::SA4_FULL};
            prof_exec(c, K_SA1 + l, nc * full[l]);
        }
    }
    const Lin head2{c->head[1].w, c->head[1].b};
    prof_mark(c, K_FP4, st);
    // fp4 / fp3 also write the next finer stage's pre-multiplied interpolation operand (one more layer on their rows) when that stage
    // takes the pre-multiplied form
    if (c->fp_pre) {
        // fp4: Y4 = feat4 x Wb^T on the 16 coarse points of a cloud, the matrix pipe then contracts over the 256 columns of points1 only (K = 256
        // instead of 768: a third of the fragments every 64-row tile streams from L2); it writes fp3's Y, fp3 writes fp2's
        launch_fp_pre(np, 4, nc, c->feat[2], nullptr, c->feat[3], c->fp_y[2], c->fp_ybuf[2], f.nn_idx[0], f.nn_w[0], c->fp_a[2], c->fp_h[0][1], c->fpo4, false,
                      c->fp_y[0], c->fp_ybuf[0], c->keep_taps, st);
        prof_exec(c, K_FP4, nc * xm::FP4_PRE);
        prof_mark(c, K_FP3, st);
        launch_fp_pre(np, 3, nc, c->feat[1], nullptr, c->fpo4, c->fp_y[0], c->fp_ybuf[0], f.nn_idx[1], f.nn_w[1], c->fp_a[0], c->fp_h[1][1], c->fpo3, true,
                      c->fp_y[1], c->fp_ybuf[1], c->keep_taps, st);
        prof_exec(c, K_FP3, nc * (xm::FP3_PRE + xm::FP3_NEXT_Y));
        prof_mark(c, K_FP2, st);
        const bool src = skip_sa1 && inv_in_source_order(c);   // level-1 features in source order, read through f.idx[0]
        const float* p1 = src ? static_cast<const float*>(c->inv_F1) : c->feat[0];
        const int* remap = src ? f.idx[0] : nullptr;
        const bool planned = fp2_ws() == 4;
        if (planned) {
            // tiles along the Z curve of the step's level-1 points, each tile's distinct coarse rows (encoder_d.hip)
            launch_morton_order(nc, f.xyz[0], c->fp2_order, st);
            launch_fp_plan(nc, c->fp2_order, 0, nullptr, f.nn_idx[2], remap, c->fp2_plan, st);
        }
        launch_fp_pre(np, 2, nc, p1, remap, c->fpo3, c->fp_y[1], c->fp_ybuf[1], f.nn_idx[2], f.nn_w[2], c->fp_a[1], c->fp_h[2][1], c->fpo2, true,
                      DenseH{nullptr, nullptr}, nullptr, true, st, 0, planned ? c->fp2_plan : nullptr);
        prof_exec(c, K_FP2, nc * xm::FP2_PRE);
    } else {
        launch_fp_h(np, 4, nc, feat[2], feat[3], f.nn_idx[0], f.nn_w[0], c->fp_h[0], nullptr, head2, c->fpo4, st);
        prof_exec(c, K_FP4, nc * xm::FP4_FULL);
        prof_mark(c, K_FP3, st);
        launch_fp_h(np, 3, nc, feat[1], c->fpo4, f.nn_idx[1], f.nn_w[1], c->fp_h[1], nullptr, head2, c->fpo3, st);
        prof_exec(c, K_FP3, nc * xm::FP3_FULL);
        prof_mark(c, K_FP2, st);
        launch_fp_h(np, 2, nc, feat[0], c->fpo3, f.nn_idx[2], f.nn_w[2], c->fp_h[2], nullptr, head2, c->fpo2, st);
        prof_exec(c, K_FP2, nc * xm::FP2_FULL);
    }
    prof_mark(c, K_FP1, st);
    if (c->fp1_reg) launch_fp_c(np, nc, c->fpo2, f.nn_idx[3], f.nn_w[3], c->fp1_s.w, c->fp1_s.b, out, st);
    else launch_fp_h(np, 1, nc, nullptr, c->fpo2, f.nn_idx[3], f.nn_w[3], c->fp_h[3], c->head_h, head2, out, st);
    prof_exec(c, K_FP1, nc * xm::FP1);
}

// Group-batched encoder stages (loop-invariant schedule, bf16x3, default kernels).  Nothing in the scene encoder reads x or anything a
// step produces: like the geometry, the stages of the g steps of a group are independent of each other.  Measured per stage at B = 64
// (all seven stages batched over the group against one step at a time, us per step): level-1 re-ordering 27.7 -> 12.2, sa2 290 -> 284,
// fp4 85 -> 61 (576 tiles on 512 slots per step: a second, nearly empty round every step), fp1 272 -> 268 -- but sa3 309 -> 318, sa4
// 329 -> 352, fp3 160 -> 166, fp2 333 -> 339: a stage whose input was just written by its producer (U / V, Y: 75 .. 300 MB per step) finds it
// in the 256 MB Infinity Cache one step at a time and in HBM when seven steps' worth lies between them.  So the group runs
//   batched : level-1 re-ordering, sa2, fp4   (small or invariant inputs; tails and launches amortised over the group)
//   per step: sa3, sa4 (producer -> consumer through the cache), then fp3, fp2, fp1 + head
// with the per-call invariants (inv_F1, sa2's U / V in source order) read modulo the clouds of one step and the level features of the
// group in c->gws.  Same kernels, same per-cloud arithmetic: bit-identical to the step-by-step form (LSDM_BATCH_ENC=0).
bool group_batched(const lsdm_ctx* c) {
    return c->batch_enc && inv_in_source_order(c) && c->fp1_reg;
}

bool ensure_group_ws(lsdm_ctx* c, int B) {
    lsdm_ctx::GroupWs& g = c->gws;
    if (g.cap_B >= B) {
        if (g.memo_cnt) c->ws_named["memo_cnt"] = {g.memo_cnt, 3};     // (lsdm_reserve clears the names; the group workspace outlives it)
        if (g.pair_cnt[0]) { c->ws_named["sa3_items"] = {g.pair_cnt[0], lsdm_ctx::kGroup}; c->ws_named["sa4_items"] = {g.pair_cnt[1], lsdm_ctx::kGroup}; }
        return true;
    }
    for (void* p : g.allocs) (void)hipFree(p);
    g.allocs.clear();
    g.cap_B = 0;
    constexpr size_t G = lsdm_ctx::kGroup;
    const size_t nc = (size_t)B * kO * G;
    auto al = [&](size_t n) -> float* {
        void* p = nullptr;
        if (hipMalloc(&p, n * sizeof(float)) != hipSuccess) { fail(c, LSDM_ERR_HIP, "group workspace: hipMalloc of %zu bytes failed", n * sizeof(float)); return nullptr; }
        g.allocs.push_back(p);
        (void)hipMemset(p, 0, n * sizeof(float));      // touch every page once, at set-up: a group larger than any before it must not meet cold pages
        return static_cast<float*>(p);
    };
    bool ok = true;
    for (int l = 1; l < 4 && ok; ++l) ok = (g.feat[l] = al(nc * kS[l] * kFeatC[l])) != nullptr;
    ok = ok && (g.ybuf[0] = al(nc * 64 * 256)) && (g.ybuf[2] = al(nc * 16 * 256));
    ok = ok && (g.pcd = al(nc * kN * 3)) && (!c->deg_fast || (g.pcd_c = al(nc * kN * 3))) && (g.o12 = al(nc * 12)) && (g.plan = reinterpret_cast<int*>(al(nc * fp_plan_ints(1024))));
    if (ok && c->sa_pair) {
        for (int l = 0; l < 2 && ok; ++l) {
            const size_t S = l ? 16 : 64, T1 = (size_t)B * kO * S;
            ok = (g.pair_rows[l] = reinterpret_cast<int*>(al(nc * S * 32))) && (g.pair_meta[l] = reinterpret_cast<int4*>(al(nc * S * 4))) &&
                 (g.pair_cnt[l] = reinterpret_cast<int*>(al(G))) && (g.pair_chunk[l] = reinterpret_cast<int*>(al(G * sa_pair_chunks((int)T1))));
        }
    }
    if (ok && c->streams > 1) {
        const size_t n1 = (size_t)B * kO;
        for (int k = 0; k < c->nside() && ok; ++k) {
            lsdm_ctx::GroupWs::Alt& a = g.alt[k];
            ok = (a.U[0] = al(n1 * 256 * 128)) && (a.V[0] = al(n1 * 256 * 128)) && (a.U[1] = al(n1 * 64 * 256)) && (a.V[1] = al(n1 * 64 * 256)) &&
                 (a.ybuf1 = al(n1 * 256 * 256)) && (a.fpo2 = al(n1 * 1024 * 128));
            if (ok && !c->side[k])
                ok = hipStreamCreateWithFlags(&c->side[k], hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&c->ev_join[k], hipEventDisableTiming) == hipSuccess;
        }
        if (ok && !c->ev_fork) ok = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) == hipSuccess;
        if (!ok) fail(c, LSDM_ERR_HIP, "group workspace: side stream");
    }
    if (ok && c->sa2_memo) {
        const size_t n1 = (size_t)B * kO;
        g.memo_ball2 = reinterpret_cast<int*>(al(n1 * kN * 32));
        g.memo_src_list = reinterpret_cast<int*>(al(n1 * kN));
        g.memo_list = reinterpret_cast<int*>(al(nc * 256));
        g.memo_cnt = reinterpret_cast<int*>(al(4));
        g.memo_ident = reinterpret_cast<int*>(al(n1 * kN));
        g.memo_unc = reinterpret_cast<unsigned char*>(al((n1 * kN + 3) / 4));
        g.memo_F2 = al(n1 * kN * 128);
        g.memo_crowd = reinterpret_cast<unsigned char*>(al((nc * 256 + 3) / 4));
        ok = g.memo_crowd && g.memo_ball2 && g.memo_src_list && g.memo_list && g.memo_cnt && g.memo_ident && g.memo_unc && g.memo_F2;
        if (ok && c->sa_pair)
            ok = (g.memo_rows = reinterpret_cast<int*>(al(n1 * kN * 32))) && (g.memo_meta = reinterpret_cast<int4*>(al(n1 * kN * 4))) &&
                 (g.memo_chunks = reinterpret_cast<int*>(al((size_t)sa_pair_chunks((int)(n1 * kN)))));
        if (ok) c->ws_named["memo_cnt"] = {g.memo_cnt, 3};      // (lsdm_workspace_tensor: {uncrowded source points of the call, crowded centroids of its last group, blocks of the memo's pass})
        if (ok) {
            std::vector<int> ident(n1 * kN);
            for (size_t i = 0; i < ident.size(); ++i) ident[i] = (int)(i % kN);
            ok = hipMemcpy(g.memo_ident, ident.data(), ident.size() * sizeof(int), hipMemcpyHostToDevice) == hipSuccess;
        }
    }
    // (lsdm_workspace_tensor: the blocks each step of the call's last group ran in sa3 / sa4)
    if (ok && c->sa_pair) { c->ws_named["sa3_items"] = {g.pair_cnt[0], lsdm_ctx::kGroup}; c->ws_named["sa4_items"] = {g.pair_cnt[1], lsdm_ctx::kGroup}; }
    if (ok) g.cap_B = B;
    return ok;
}

// front half of the group: everything up to fp4 (which also writes fp3's pre-multiplied operand for every step of the group)
// Side streams (LSDM_STREAMS = 1 + their number): step j of the group runs on lane j % LSDM_STREAMS, lane 0 being the caller's stream and every other
// lane a stream of the context with its own scratch set (gws.alt), so that the chains of consecutive steps run side by side -- the
// persistent one-work-group-per-CU kernels of one chain fill the ramps, tails and barrier phases of the other (two whole calls side by
// side: 2.16 -> 2.10 ms per chain-step, tools/two_stream_probe.py).  Same kernels on the same data: bit-identical.  Per-stage event
// timing (lsdm_profile_*) needs the stages in sequence and keeps everything on one stream.
static int lanes(const lsdm_ctx* c) { return (c->streams > 1 && c->side[0] != nullptr && !c->prof_on) ? 1 + c->nside() : 1; }
static void stream_fork(lsdm_ctx* c, hipStream_t st) {
    (void)hipEventRecord(c->ev_fork, st);
    for (int k = 0; k < c->nside(); ++k) (void)hipStreamWaitEvent(c->side[k], c->ev_fork, 0);
}
static void stream_join(lsdm_ctx* c, hipStream_t st) {
    for (int k = 0; k < c->nside(); ++k) {
        (void)hipEventRecord(c->ev_join[k], c->side[k]);
        (void)hipStreamWaitEvent(st, c->ev_join[k], 0);
    }
}

void run_encoder_group_front(lsdm_ctx* c, const FrontSet& f, const float* clouds, int B, int steps, hipStream_t st0) {
    const lsdm_ctx::GroupWs& g = c->gws;
    const int nc1 = c->enc_nc > 0 ? c->enc_nc : B * kO, nc = nc1 * steps;
    const int np = c->prec == LSDM_PREC_BF16X3 ? 2 : 1;
    const int nl = lanes(c);
    const bool two = nl > 1;
    hipStream_t st = st0;
    prof_mark(c, K_REMAP, st, steps);
    launch_l1_remap(nc, c->inv_F1, 256, f.idx[0], c->inv_nn8_d, c->inv_nn8_i, f.xyz[0], clouds, nullptr, f.nn_idx[3], f.nn_w[3], st, nc1);
    prof_mark(c, K_SA1 + 1, st, steps);
    if (c->memo_ready) {
        // (centroids with fewer than 32 points within r copy the call's F2 row of their source point; the others are the launch's work list)
        (void)hipMemsetAsync(g.memo_cnt + 1, 0, sizeof(int), st);
        launch_memo_cent(nc, f.idx[0], f.idx[1], g.memo_unc, g.memo_F2, nc1, g.feat[1], g.memo_list, g.memo_cnt + 1, g.memo_crowd, st);
        prof_mark(c, K_BALL1 + 1, st, steps);
        launch_ball_query(f.xyz[0], f.xyz[1], nc, 1024, 256, (float)(kRadius[1] * kRadius[1]), f.ball[1], 0, st, g.memo_crowd);
        prof_mark(c, K_SA1 + 1, st, steps);
        launch_sa_u_list(np, 256, c->sa_U[2], c->sa_V[2], f.idx[1], f.ball[1], f.idx[0], c->sa_ub[2].w, c->sa_ub[2].b, g.feat[1], st, nc1, g.memo_list,
                         g.memo_cnt + 1);
        prof_exec_dev(c, K_SA1 + 1, g.memo_cnt + 1, xm::SA2_CENT, st);      // the crowded centroids of the group only
    } else {
        launch_sa_u(np, 1, nc, c->sa_U[2], c->sa_V[2], f.idx[1], f.ball[1], f.idx[0], c->sa_ub[2].w, c->sa_ub[2].b, g.feat[1], st, nc1);
        prof_exec(c, K_SA1 + 1, nc * 256 * xm::SA2_CENT);
    }
    // fp2's tile plans of every step of the group in one launch: the source clouds' Z-curve order (once per call) read through each step's
    // level-1 FPS order
    if (fp2_ws() == 4) {
        prof_mark(c, K_FP2, st, 0);      // (timed with the stage, not counted as launches of it: avg_ms stays per step)
        launch_fp_plan(nc, c->inv_order, nc1, f.idx[0], f.nn_idx[2], f.idx[0], g.plan, st);
    }
    // sa3 / sa4's item lists of every step of the group: two groups whose distinct neighbours fit into 32 rows together share a block
    if (c->sa_pair) {
        prof_mark(c, K_SA1 + 3, st, 0);      // (both levels in one pair of launches, timed with sa4 -- the stage the roofline leg follows)
        const PairLevel lv[2] = {{f.ball[2], f.idx[2], nc1 * 64, 64, 256, 0, steps, g.pair_chunk[0], g.pair_rows[0], g.pair_meta[0], g.pair_cnt[0], nullptr, 0},
                                 {f.ball[3], f.idx[3], nc1 * 16, 16, 64, 0, steps, g.pair_chunk[1], g.pair_rows[1], g.pair_meta[1], g.pair_cnt[1], nullptr, 0}};
        launch_sa_pair_plans(lv, 2, st);
        for (int l = 0; l < 2; ++l) prof_read_counts(c, g.pair_cnt[l], steps, c->pair_host[l], st);
    }
    // (the chains of a step in two or three cloud slices, so that each slice's U / V or Y fits the Infinity Cache with room to spare, were
    // measured: fp2 337 -> 323 us, but fp1 275 -> 295, fp3 165 -> 190, sa3 / sa4 slower as well -- the extra launch ramps and tails cost
    // more than the cache residency gains)
    constexpr int PS = 1;
    static_assert(PS == 1, "the item lists of sa3 / sa4 cover whole steps");
    const int ncp = nc1;
    if (two) stream_fork(c, st0);
    for (int j = 0; j < steps; ++j) {
        const FrontSet fj = c->fset(j, B);
        const int ln = j % nl;                 // lane of the step
        st = ln ? c->side[ln - 1] : st0;
        float* const U0 = ln ? g.alt[ln - 1].U[0] : c->sa_U[0];
        float* const V0 = ln ? g.alt[ln - 1].V[0] : c->sa_V[0];
        float* const U1 = ln ? g.alt[ln - 1].U[1] : c->sa_U[1];
        float* const V1 = ln ? g.alt[ln - 1].V[1] : c->sa_V[1];
        for (int h = 0; h < PS; ++h) {
            const size_t o = (size_t)h * ncp;      // first cloud of the slice
            float* const f1 = g.feat[1] + ((size_t)j * nc1 + o) * 256 * 128;
            float* const f2 = g.feat[2] + ((size_t)j * nc1 + o) * 64 * 256;
            float* const f3 = g.feat[3] + ((size_t)j * nc1 + o) * 16 * 512;
            prof_mark(c, K_SA1 + 2, st);
            launch_lin_u(np, 2, ncp * 256, f1, fj.xyz[1] + o * 256 * 3, c->sa_uw[0], c->sa_wx[0], U0, V0, st);
            if (c->sa_pair) {
                const size_t t1 = (size_t)j * nc1 * 64;
                launch_sa_u_items(np, 2, U0, V0, g.pair_rows[0] + t1 * 32, g.pair_meta[0] + t1, g.pair_cnt[0] + j, c->sa_ub[0].w, c->sa_ub[0].b, f2, st);
                prof_exec(c, K_SA1 + 2, ncp * xm::SA3_UV + c->pair_host[0][j] * xm::SA3_ITEM);     // the blocks that ran
            } else {
                launch_sa_u(np, 2, ncp, U0, V0, fj.idx[2] + o * 64, fj.ball[2] + o * 64 * 32, nullptr, c->sa_ub[0].w, c->sa_ub[0].b, f2, st);
                prof_exec(c, K_SA1 + 2, ncp * xm::SA3);
            }
            prof_mark(c, K_SA1 + 3, st);
            launch_lin_u(np, 3, ncp * 64, f2, fj.xyz[2] + o * 64 * 3, c->sa_uw[1], c->sa_wx[1], U1, V1, st);
            if (c->sa_pair) {
                const size_t t1 = (size_t)j * nc1 * 16;
                launch_sa_u_items(np, 3, U1, V1, g.pair_rows[1] + t1 * 32, g.pair_meta[1] + t1, g.pair_cnt[1] + j, c->sa_ub[1].w, c->sa_ub[1].b, f3, st);
                prof_exec(c, K_SA1 + 3, ncp * xm::SA4_UV + c->pair_host[1][j] * xm::SA4_ITEM);
            } else {
                launch_sa_u(np, 3, ncp, U1, V1, fj.idx[3] + o * 16, fj.ball[3] + o * 16 * 32, nullptr, c->sa_ub[1].w, c->sa_ub[1].b, f3, st);
                prof_exec(c, K_SA1 + 3, ncp * xm::SA4);
            }
        }
    }
    st = st0;
    if (two) stream_join(c, st0);
    prof_mark(c, K_FP4, st, steps);
    // (fp4 / fp3 store only the next finer stage's pre-multiplied operand inside a multi-step call: keep_out = false, `out` unused)
    launch_fp_pre(np, 4, nc, g.feat[2], nullptr, g.feat[3], c->fp_y[2], g.ybuf[2], f.nn_idx[0], f.nn_w[0], c->fp_a[2], c->fp_h[0][1], nullptr, false,
                  c->fp_y[0], g.ybuf[0], false, st);
    prof_exec(c, K_FP4, nc * xm::FP4_PRE);
    prof_mark(c, -1, st);
}

// back half, one step: fp3 (its pre-multiplied operand came from the group's fp4 launch), fp2, fp1 + head -> c->pcd
// (lane > 0: the step runs on side stream lane - 1 with that lane's scratch set)
void run_encoder_group_step(lsdm_ctx* c, const FrontSet& fj, int B, int j, float* pcd_out, hipStream_t st, int lane) {
    const lsdm_ctx::GroupWs& g = c->gws;
    const int nc1 = c->enc_nc > 0 ? c->enc_nc : B * kO;
    const int np = c->prec == LSDM_PREC_BF16X3 ? 2 : 1;
    float* const ybuf1 = lane ? g.alt[lane - 1].ybuf1 : c->fp_ybuf[1];
    float* const fpo2 = lane ? g.alt[lane - 1].fpo2 : c->fpo2;
    const int* const plan = fp2_ws() == 4 ? g.plan + (size_t)j * nc1 * fp_plan_ints(1024) : nullptr;
    // (fp3 / fp2 / fp1 in two or three cloud slices, so that fp2's 302 MB output stays in the Infinity Cache until fp1 has read it, were
    // measured: fp2 329 us either way, fp3 164 -> 186 -- not kept)
    {
        const size_t o = 0;
        const int ncp = nc1;
        prof_mark(c, K_FP3, st);
        launch_fp_pre(np, 3, ncp, g.feat[1] + ((size_t)j * nc1 + o) * 256 * 128, nullptr, nullptr, c->fp_y[0], g.ybuf[0] + ((size_t)j * nc1 + o) * 64 * 256,
                      fj.nn_idx[1] + o * 256 * 3, fj.nn_w[1] + o * 256 * 3, c->fp_a[0], c->fp_h[1][1], nullptr, true, c->fp_y[1], ybuf1, false, st);
        prof_exec(c, K_FP3, ncp * (xm::FP3_PRE + xm::FP3_NEXT_Y));
        prof_mark(c, K_FP2, st);
        launch_fp_pre(np, 2, ncp, static_cast<const float*>(c->inv_F1) + o * 1024 * 64, fj.idx[0] + o * 1024, nullptr, c->fp_y[1], ybuf1, fj.nn_idx[2] + o * 1024 * 3,
                      fj.nn_w[2] + o * 1024 * 3, c->fp_a[1], c->fp_h[2][1], fpo2, true, DenseH{nullptr, nullptr}, nullptr, true, st, 0, plan);
        prof_exec(c, K_FP2, ncp * xm::FP2_PRE);
        prof_mark(c, K_FP1, st);
        // (fp1's fine level is the source cloud: walked along the Z curve sorted once per call, LSDM_FP1_ORDER=0: in index order)
        launch_fp_c(np, ncp, fpo2, fj.nn_idx[3] + o * 1024 * 3, fj.nn_w[3] + o * 1024 * 3, c->fp1_s.w, c->fp1_s.b, pcd_out + o * kN * 3, st,
                                                  c->fp1_order ? c->inv_order + o * 1024 : nullptr, 0);
        prof_exec(c, K_FP1, ncp * xm::FP1);
    }
}

int run_scene_encoder(lsdm_ctx* c, const float* clouds, const int64_t* starts, int nc, float* out, hipStream_t st) {
    run_geometry(c, c->fs0, clouds, starts, nc, 1, st);
    run_encoder_dense(c, c->fs0, clouds, nc, out, st);
    return check_launch(c, "scene encoder");
}

// x-independent conditioning of `g` consecutive steps into front sets f, f+1, ... (contiguous): timestep / text /
// category MLPs + up-sampling layers 1-2, one launch each for the group; `t` holds g*B timesteps (step-major).
// The human branch depends on given_objs only and is evaluated once per multi-step call.
void run_conditioning(lsdm_ctx* c, FrontSet& f, int B, int g, const int64_t* t, const float* mask, const float* given_objs,
                      const float* given_cats, const float* text, float* out_cat, bool reuse_hm, hipStream_t st, bool shared = false) {
    const int np = c->prec == LSDM_PREC_F32 ? 0 : (c->prec == LSDM_PREC_BF16X3 ? 2 : 1);
    shared = shared && np != 0;
    prof_mark(c, K_COND, st);
    if (!shared) {
        launch_cond(c->cond, t, text, given_cats, mask, B, g, c->max_cats, out_cat, f.e0, f.tr, f.w, st);
        prof_mark(c, K_UP12, st);
        launch_upsample12(np, f.e0, c->up0, c->up2.w, c->up2.b, B * g, 0, f.h2p, st);
        prof_exec(c, K_UP12, np == 0 ? B * g * kKernelMacs[K_UP12] : B * g * xm::UP12_HALF);
        // per-sample timesteps: the timestep half of combine_extraction's pre-activation per (step, sample) row
        // (ctext_kernel evaluates the timestep / text halves of combine_extraction's pre-activation: work the reference counts in the denoiser
        // -- timed and accounted with that stage, weight 0: not a launch of it)
        if (np != 0 && c->den_reg && g == 1) {
            prof_mark(c, K_DENOISE, st, 0);
            launch_ctext(np, c->den_h, B, f.h2p, c->ct_time, true, st);
            prof_exec(c, K_DENOISE, B * xm::CTEXT_HALF);
        }
    } else if (!reuse_hm) {
        // per call: the sample rows (text / category embeddings, attention weights, translation tokens) of set 0
        launch_cond(c->cond, t, text, given_cats, mask, B, 1, c->max_cats, out_cat, f.e0, f.tr, f.w, st);
    }
    if (!reuse_hm) {
        // once per call: what depends on (given_objs, text) only
        if (np != 0) {
            prof_mark(c, K_UP12, st);
            launch_upsample12(np, f.e0, c->up0, c->up2.w, c->up2.b, B, 128, c->h2x, st);
            prof_exec(c, K_UP12, B * xm::UP12_HALF);
            prof_mark(c, K_DENOISE, st, 0);
            launch_ctext(np, c->den_h, B, c->h2x, c->ctext, false, st);
            prof_exec(c, K_DENOISE, B * xm::CTEXT_HALF);
        }
        prof_mark(c, K_POSA, st);
        launch_posa(c->posa, given_objs, B, c->posa_ws, c->hm, st);
        prof_exec(c, K_POSA, B * kKernelMacs[K_POSA]);
    }
    if (shared) {
        // per group: one timestep-embedding row per STEP (t of step j = t[j * B]) into the first halves of rows 0 .. g-1 of e0
        // (their text halves were consumed above), and its up-sampling rows
        prof_mark(c, K_COND, st);
        launch_cond_time(c->cond, t, B, g, f.e0, st);
        prof_mark(c, K_UP12, st);
        launch_upsample12(np, f.e0, c->up0, c->up2.w, c->up2.b, g, 0, f.h2p, st);
        prof_exec(c, K_UP12, g * xm::UP12_HALF);      // one row per STEP: every chain of the batch is at the same timestep
        prof_mark(c, K_DENOISE, st, 0);
        launch_ctext(np, c->den_h, g, f.h2p, c->ct_time, true, st);
        prof_exec(c, K_DENOISE, g * xm::CTEXT_HALF);
    }
    prof_mark(c, -1, st);
}

// FRONT of a single step: conditioning + PointNet++ geometry (nothing here reads x)
void run_front(lsdm_ctx* c, FrontSet& f, int B, const int64_t* t, const float* mask, const float* given_objs, const float* given_cats,
               const float* text, const int64_t* starts, float* out_cat, bool reuse_hm, hipStream_t st) {
    run_conditioning(c, f, B, 1, t, mask, given_objs, given_cats, text, out_cat, reuse_hm, st);
    run_geometry(c, f, given_objs, starts, B * kO, 1, st);
}

// conditioning that depends on the scene only: dense encoder stages + collapsed scene attention (needs f.w, f.tr)
void run_scene_conditioning(lsdm_ctx* c, const FrontSet& f, int B, const float* given_objs, hipStream_t st, bool skip_sa1 = false) {
    run_encoder_dense(c, f, given_objs, B * kO, c->pcd, st, skip_sa1);
    prof_mark(c, K_SATTN, st);
    launch_scene_attn(c->sattn, c->pcd, f.w, f.tr, B, c->o12, st);
}

// the fused per-point denoiser + sampler update (the only stage that reads x)
// nsteps > 1 (bf16 modes, hoisted conditioning): that many consecutive steps of the front sets f, f+1, ... in ONE launch; the
// final sample lands in x
void run_denoise(lsdm_ctx* c, const FrontSet& f, int mode, float eta, int B, float* x, const int64_t* t, const float* mask,
                 const float* noise, float* out_x0, float* out_guiding, float* out_sample, hipStream_t st, int nsteps = 1,
                 bool shared = false) {
    prof_mark(c, K_DENOISE, st);
    if (c->prec == LSDM_PREC_F32)
        launch_denoise(c->den, c->sched, mode, eta, B, x, t, mask, c->pcd, f.w, c->o12, c->hm, f.h2p, noise, out_x0, out_guiding,
                       out_sample, st);
    else if (c->den_reg)
        launch_denoise_r(c->prec == LSDM_PREC_BF16X3 ? 2 : 1, c->den_r, c->den_h.pwt, c->sched, mode, eta, B, nsteps, shared ? 0 : 1, x, t, mask, c->pcd, f.w,
                         c->o12, c->hm, c->ctext, shared && f.ctime ? f.ctime : c->ct_time, noise, c->s_stash, out_x0, out_guiding, out_sample,
                         c->num_cus, st);
    else
        launch_denoise_h(c->prec == LSDM_PREC_BF16X3 ? 2 : 1, c->den_h, c->sched, mode, eta, B, nsteps, shared ? 0 : 1, x, t, mask, c->pcd, f.w, c->o12, c->hm,
                         f.h2p, c->ctext, shared ? (f.ctime ? f.ctime : c->ct_time) : nullptr, noise, out_x0, out_guiding, out_sample, st);
    if (c->prec != LSDM_PREC_F32 && c->den_reg) prof_exec(c, K_DENOISE, (double)B * (nsteps * (xm::DEN_S + xm::DEN_PASS) + (out_guiding ? xm::DEN_PASS : 0.0)));
    else prof_exec(c, K_DENOISE, (double)B * nsteps * (kKernelMacs[K_DENOISE] - (c->prec != LSDM_PREC_F32 && shared ? 1024.0 * (512 * 256 + 256 * 128) : 0.0)));
    prof_mark(c, -1, st);
}

// BACK of a step: dense encoder stages, collapsed scene attention, the fused per-point denoiser + sampler update
void run_back(lsdm_ctx* c, const FrontSet& f, int mode, float eta, int B, float* x, const int64_t* t, const float* mask,
              const float* given_objs, const float* noise, float* out_x0, float* out_guiding, float* out_sample, hipStream_t st,
              bool inv = false, bool shared = false) {
    if (inv) {
        // level 1 of the encoder from the call's invariants: feature rows and 3-NN positions in this step's FPS order
        prof_mark(c, K_REMAP, st);
        launch_l1_remap(B * kO, c->inv_F1, 256, f.idx[0], c->inv_nn8_d, c->inv_nn8_i, f.xyz[0], given_objs,
                        inv_in_source_order(c) ? nullptr : (void*)c->feat[0], f.nn_idx[3], f.nn_w[3], st);
    }
    run_scene_conditioning(c, f, B, given_objs, st, inv);
    run_denoise(c, f, mode, eta, B, x, t, mask, noise, out_x0, out_guiding, out_sample, st, 1, shared);
}

int run_step(lsdm_ctx* c, int mode, float eta, int B, float* x, const float* mask, const int64_t* t, const float* given_objs,
             const float* given_cats, const float* text, const int64_t* starts, const float* noise, float* out_cat, float* out_x0,
             float* out_guiding, float* out_sample, hipStream_t st, bool reuse_hm = false) {
    run_front(c, c->fs0, B, t, mask, given_objs, given_cats, text, starts, out_cat, reuse_hm, st);
    run_back(c, c->fs0, mode, eta, B, x, t, mask, given_objs, noise, out_x0, out_guiding, out_sample, st);
    return check_launch(c, "step");
}

int check_ready(lsdm_ctx* c, int B, bool need_sched) {
    if (!c) return LSDM_ERR_ARG;
    if (!c->finalized) return fail(c, LSDM_ERR_STATE, "parameters not finalised (lsdm_finalize_params)");
    if (B <= 0 || B > c->cap_B) return fail(c, LSDM_ERR_STATE, "batch %d exceeds reserved workspace %d (lsdm_reserve)", B, c->cap_B);
    if (need_sched && c->sched_T == 0) return fail(c, LSDM_ERR_STATE, "schedule not set (lsdm_set_schedule)");
    return LSDM_OK;
}

}  // namespace

extern "C" {

int lsdm_create(lsdm_ctx** out, int device) {
    if (!out) return fail(nullptr, LSDM_ERR_ARG, "out is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return fail(nullptr, LSDM_ERR_HIP, "no HIP device available");
    if (device < 0 || device >= n) return fail(nullptr, LSDM_ERR_ARG, "device %d out of range (%d devices)", device, n);
    DeviceGuard guard(nullptr, device);
    if (!guard.ok) return LSDM_ERR_HIP;
    lsdm_ctx* c = new lsdm_ctx();
    c->device = device;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) { c->fps_slots = 4 * LSDM_FPS_WPE * cus; c->num_cus = cus; }
    *out = c;
    return LSDM_OK;
}

int lsdm_destroy(lsdm_ctx* c) {
    if (!c) return LSDM_OK;
    {
        DeviceGuard guard(c);
        (void)hipDeviceSynchronize();
        for (void* p : c->allocs) (void)hipFree(p);
        for (void* p : c->sched_allocs) (void)hipFree(p);
        for (void* p : c->ws_allocs) (void)hipFree(p);
        if (c->enc_starts) (void)hipFree(c->enc_starts);
        if (c->cls_host) (void)hipHostFree(c->cls_host);
        for (hipEvent_t e : c->prof_ev) (void)hipEventDestroy(e);
        for (void* p : c->gws.allocs) (void)hipFree(p);
        if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
        for (hipEvent_t e : c->ev_join) if (e) (void)hipEventDestroy(e);
        for (hipStream_t q : c->side) if (q) (void)hipStreamDestroy(q);
    }
    delete c;
    return LSDM_OK;
}

const char* lsdm_last_error(const lsdm_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int lsdm_set_param(lsdm_ctx* c, const char* name, const float* data, const int64_t* shape, int ndim) {
    if (!c || !name || !data || ndim < 0 || ndim > 8) return fail(c, LSDM_ERR_ARG, "lsdm_set_param: bad argument");
    HostTensor t;
    t.shape.assign(shape, shape + ndim);
    t.data.assign(data, data + t.numel());
    c->host[name] = std::move(t);
    c->finalized = false;
    return LSDM_OK;
}

int lsdm_set_precision(lsdm_ctx* c, int precision) {
    if (!c || precision < LSDM_PREC_F32 || precision > LSDM_PREC_BF16) return fail(c, LSDM_ERR_ARG, "lsdm_set_precision: bad precision %d", precision);
    if (precision != c->prec) {
        DeviceGuard guard(c);
        if (!guard.ok) return LSDM_ERR_HIP;
        HIP_TRY(c, hipDeviceSynchronize());
        // operand packing and the workspace element types depend on it
        c->prec = precision;
        c->finalized = false;
        for (void* p : c->ws_allocs) (void)hipFree(p);
        c->ws_allocs.clear();
        c->ws_named.clear();
        c->cap_B = 0;
    }
    return LSDM_OK;
}

int lsdm_get_precision(const lsdm_ctx* c) { return c ? c->prec : LSDM_ERR_ARG; }

int lsdm_finalize_params(lsdm_ctx* c, int C) {
    if (!c || C <= 0 || C > 512) return fail(c, LSDM_ERR_ARG, "lsdm_finalize_params: bad max_cats");
    DeviceGuard guard(c);
    if (!guard.ok) return LSDM_ERR_HIP;
    for (void* p : c->allocs) (void)hipFree(p);
    c->allocs.clear();
    c->max_cats = C;
    bool ok = true;
    const char* sa_names[4] = {"sa1", "sa2", "sa3", "sa4"};
    const int sa_in[4] = {6, 67, 131, 259};
    const int sa_mlp[4][3] = {{32, 32, 64}, {64, 64, 128}, {128, 128, 256}, {256, 256, 512}};
    for (int l = 0; l < 4 && ok; ++l) {
        int last = sa_in[l];
        for (int i = 0; i < 3 && ok; ++i) {
            const std::string p = std::string("pcd_backbone.") + sa_names[l];
            ok = load_dense(c, p + ".mlp_convs." + std::to_string(i) + ".weight", p + ".mlp_convs." + std::to_string(i) + ".bias",
                            p + ".mlp_bns." + std::to_string(i), sa_mlp[l][i], last, &c->sa[l][i]);
            last = sa_mlp[l][i];
        }
    }
    const char* fp_names[4] = {"fp4", "fp3", "fp2", "fp1"};
    const int fp_in[4] = {768, 384, 320, 128};
    const int fp_mlp[4][3] = {{256, 256, 0}, {256, 256, 0}, {256, 128, 0}, {128, 128, 128}};
    for (int l = 0; l < 4 && ok; ++l) {
        int last = fp_in[l];
        for (int i = 0; i < 3 && ok && fp_mlp[l][i]; ++i) {
            const std::string p = std::string("pcd_backbone.") + fp_names[l];
            ok = load_dense(c, p + ".mlp_convs." + std::to_string(i) + ".weight", p + ".mlp_convs." + std::to_string(i) + ".bias",
                            p + ".mlp_bns." + std::to_string(i), fp_mlp[l][i], last, &c->fp[l][i]);
            last = fp_mlp[l][i];
        }
    }
    ok = ok && load_dense(c, "pcd_backbone.conv1.weight", "pcd_backbone.conv1.bias", "pcd_backbone.bn1", 128, 128, &c->head[0]);
    if (c->prec != LSDM_PREC_F32) {
        const int np = c->prec == LSDM_PREC_BF16X3 ? 2 : 1;
        for (int l = 0; l < 4 && ok; ++l) {
            int last = sa_in[l];
            for (int i = 0; i < 3 && ok; ++i) {
                const std::string p = std::string("pcd_backbone.") + sa_names[l];
                // first layer: the kernels order K as [features (D) || xyz - centroid (3)], the reference as [xyz (3) || features]
                std::vector<int> perm;
                if (i == 0) {
                    const int D = last - 3;
                    for (int k = 0; k < D; ++k) perm.push_back(3 + k);
                    for (int k = 0; k < 3; ++k) perm.push_back(k);
                }
                ok = load_dense_h(c, p + ".mlp_convs." + std::to_string(i) + ".weight", p + ".mlp_convs." + std::to_string(i) + ".bias",
                                  p + ".mlp_bns." + std::to_string(i), sa_mlp[l][i], last, np, perm, &c->sa_h[l][i]);
                last = sa_mlp[l][i];
            }
        }
        for (int l = 0; l < 4 && ok; ++l) {
            int last = fp_in[l];
            for (int i = 0; i < 3 && ok && fp_mlp[l][i]; ++i) {
                const std::string p = std::string("pcd_backbone.") + fp_names[l];
                ok = load_dense_h(c, p + ".mlp_convs." + std::to_string(i) + ".weight", p + ".mlp_convs." + std::to_string(i) + ".bias",
                                  p + ".mlp_bns." + std::to_string(i), fp_mlp[l][i], last, np, {}, &c->fp_h[l][i]);
                last = fp_mlp[l][i];
            }
        }
        ok = ok && load_dense_h(c, "pcd_backbone.conv1.weight", "pcd_backbone.conv1.bias", "pcd_backbone.bn1", 128, 128, np, {}, &c->head_h[0]);
        if (ok) {
            const HostTensor* w2 = find(c, "pcd_backbone.conv2.weight", {3, 128});
            ok = w2 != nullptr;
            if (ok) {
                std::vector<float> wpad((size_t)32 * 128, 0.0f);
                std::copy(w2->data.begin(), w2->data.end(), wpad.begin());
                std::vector<uint16_t> pk = pack_weight_h(wpad, 32, 128, np, {});
                c->head_h[1].w = reinterpret_cast<const __bf16*>(upload(c, reinterpret_cast<const float*>(pk.data()), pk.size() / 2));
                c->head_h[1].b = nullptr;
                ok = c->head_h[1].w != nullptr;
            }
        }
        // Plain bf16 (np = 1) runs the same kernel family as bf16x3 with single-plane operands and ONE matrix instruction per product; only sa1
        // (once per call in the sampler loop) keeps two planes: its item-list kernel exists in that form only.
        if (np == 1) {      // the A/B switches select reference forms of the bf16x3 build; plain bf16 always takes the default family
            c->sa_split = true; c->fp_pre = true; c->fp1_reg = 1;
        }
        ok = ok && build_sa_reg_blob(c, "pcd_backbone.sa1", 3, sa_mlp[0], 2, &c->sa_r[0]) &&
             build_sa_reg_blob(c, "pcd_backbone.sa2", 64, sa_mlp[1], np, &c->sa_r[1]) &&
             build_sa_reg_blob(c, "pcd_backbone.sa3", 128, sa_mlp[2], np, &c->sa_r[2]);
        if (ok) {
            // sa3: first layer (128 x 131, reference column order [xyz (3) || features (128)]) split into Wx | Wf
            const std::string pre = "pcd_backbone.sa3";
            std::vector<float> W, Bv;
            ok = fold_dense(c, pre + ".mlp_convs.0.weight", pre + ".mlp_convs.0.bias", pre + ".mlp_bns.0", 128, 131, W, Bv);
            if (ok) {
                std::vector<float> wf((size_t)128 * 128), wx((size_t)128 * 3);
                for (int n = 0; n < 128; ++n) {
                    for (int k = 0; k < 3; ++k) wx[(size_t)n * 3 + k] = W[(size_t)n * 131 + k];
                    for (int k = 0; k < 128; ++k) wf[(size_t)n * 128 + k] = W[(size_t)n * 131 + 3 + k];
                }
                std::vector<uint16_t> pk = pack_weight_h(wf, 128, 128, np, {});
                c->sa_uw[0].w = reinterpret_cast<const __bf16*>(upload(c, reinterpret_cast<const float*>(pk.data()), pk.size() / 2));
                c->sa_uw[0].b = upload(c, Bv.data(), Bv.size());
                c->sa_wx[0] = upload(c, wx.data(), wx.size());
                std::vector<ChainLayer> layers = {{pre + ".mlp_convs.1.weight", pre + ".mlp_convs.1.bias", pre + ".mlp_bns.1", 128, 128},
                                                  {pre + ".mlp_convs.2.weight", pre + ".mlp_convs.2.bias", pre + ".mlp_bns.2", 256, 128}};
                std::vector<int> ident(128);
                for (int k = 0; k < 128; ++k) ident[k] = k;
                ok = c->sa_uw[0].w && c->sa_uw[0].b && c->sa_wx[0] && build_chain_blob(c, layers, ident, np, &c->sa_ub[0]);
            }
            // sa2: first layer (64 x 67) split the same way; layers 2-3 as a resident fragment blob (sa_ur_kernel)
            if (ok) {
                const std::string p2 = "pcd_backbone.sa2";
                std::vector<float> W2, B2;
                ok = fold_dense(c, p2 + ".mlp_convs.0.weight", p2 + ".mlp_convs.0.bias", p2 + ".mlp_bns.0", 64, 67, W2, B2);
                if (ok) {
                    std::vector<float> wf((size_t)64 * 64), wx((size_t)64 * 3);
                    for (int n = 0; n < 64; ++n) {
                        for (int k = 0; k < 3; ++k) wx[(size_t)n * 3 + k] = W2[(size_t)n * 67 + k];
                        for (int k = 0; k < 64; ++k) wf[(size_t)n * 64 + k] = W2[(size_t)n * 67 + 3 + k];
                    }
                    std::vector<uint16_t> pk = pack_weight_h(wf, 64, 64, np, {});
                    c->sa_uw[2].w = reinterpret_cast<const __bf16*>(upload(c, reinterpret_cast<const float*>(pk.data()), pk.size() / 2));
                    c->sa_uw[2].b = upload(c, B2.data(), B2.size());
                    c->sa_wx[2] = upload(c, wx.data(), wx.size());
                    std::vector<ChainLayer> layers = {{p2 + ".mlp_convs.1.weight", p2 + ".mlp_convs.1.bias", p2 + ".mlp_bns.1", 64, 64},
                                                      {p2 + ".mlp_convs.2.weight", p2 + ".mlp_convs.2.bias", p2 + ".mlp_bns.2", 128, 64}};
                    std::vector<int> ident(64);
                    for (int k = 0; k < 64; ++k) ident[k] = k;
                    ok = c->sa_uw[2].w && c->sa_uw[2].b && c->sa_wx[2] && build_chain_blob(c, layers, ident, np, &c->sa_ub[2]);
                }
            }
            // fp1 + head: the five layers as one chain (conv2's three rows zero-padded to a 32-channel block)
            if (ok) {
                const std::string f1 = "pcd_backbone.fp1", pb = "pcd_backbone";
                std::vector<ChainLayer> layers = {{f1 + ".mlp_convs.0.weight", f1 + ".mlp_convs.0.bias", f1 + ".mlp_bns.0", 128, 128},
                                                  {f1 + ".mlp_convs.1.weight", f1 + ".mlp_convs.1.bias", f1 + ".mlp_bns.1", 128, 128},
                                                  {f1 + ".mlp_convs.2.weight", f1 + ".mlp_convs.2.bias", f1 + ".mlp_bns.2", 128, 128},
                                                  {pb + ".conv1.weight", pb + ".conv1.bias", pb + ".bn1", 128, 128},
                                                  {pb + ".conv2.weight", pb + ".conv2.bias", "", 3, 128, 32}};
                std::vector<int> ident(128);
                for (int k = 0; k < 128; ++k) ident[k] = k;
                ok = build_chain_blob(c, layers, ident, np, &c->fp1_s);
            }
            // fp3 / fp2: first layer ([points1 (D1) || interpolated coarse output (256)] -> 256) split at the concat boundary
            {
                const char* pre[3] = {"pcd_backbone.fp3", "pcd_backbone.fp2", "pcd_backbone.fp4"};
                const int d1[3] = {128, 64, 256}, d2[3] = {256, 256, 512};
                for (int i = 0; i < 3 && ok; ++i) {
                    const int K = d1[i] + d2[i];
                    std::vector<float> Wf, Bf;
                    ok = fold_dense(c, std::string(pre[i]) + ".mlp_convs.0.weight", std::string(pre[i]) + ".mlp_convs.0.bias",
                                    std::string(pre[i]) + ".mlp_bns.0", 256, K, Wf, Bf);
                    if (!ok) break;
                    std::vector<float> wa((size_t)256 * d1[i]), wb((size_t)256 * d2[i]);
                    for (int n = 0; n < 256; ++n) {
                        for (int k = 0; k < d1[i]; ++k) wa[(size_t)n * d1[i] + k] = Wf[(size_t)n * K + k];
                        for (int k = 0; k < d2[i]; ++k) wb[(size_t)n * d2[i] + k] = Wf[(size_t)n * K + d1[i] + k];
                    }
                    std::vector<uint16_t> pa = pack_weight_h(wa, 256, d1[i], np, {}), pb = pack_weight_h(wb, 256, d2[i], np, {});
                    c->fp_a[i].w = reinterpret_cast<const __bf16*>(upload(c, reinterpret_cast<const float*>(pa.data()), pa.size() / 2));
                    c->fp_a[i].b = upload(c, Bf.data(), Bf.size());
                    c->fp_y[i].w = reinterpret_cast<const __bf16*>(upload(c, reinterpret_cast<const float*>(pb.data()), pb.size() / 2));
                    c->fp_y[i].b = nullptr;
                    ok = c->fp_a[i].w && c->fp_a[i].b && c->fp_y[i].w;
                }
            }
            // sa4: first layer (256 x 259) split the same way; layers 2-3 are the LDS-tiled operands sa_h[3][1..2]
            std::vector<float> W4, B4;
            ok = ok && fold_dense(c, "pcd_backbone.sa4.mlp_convs.0.weight", "pcd_backbone.sa4.mlp_convs.0.bias", "pcd_backbone.sa4.mlp_bns.0", 256, 259, W4, B4);
            if (ok) {
                std::vector<float> wf((size_t)256 * 256), wx((size_t)256 * 3);
                for (int n = 0; n < 256; ++n) {
                    for (int k = 0; k < 3; ++k) wx[(size_t)n * 3 + k] = W4[(size_t)n * 259 + k];
                    for (int k = 0; k < 256; ++k) wf[(size_t)n * 256 + k] = W4[(size_t)n * 259 + 3 + k];
                }
                std::vector<uint16_t> pk = pack_weight_h(wf, 256, 256, np, {});
                c->sa_uw[1].w = reinterpret_cast<const __bf16*>(upload(c, reinterpret_cast<const float*>(pk.data()), pk.size() / 2));
                c->sa_uw[1].b = upload(c, B4.data(), B4.size());
                c->sa_wx[1] = upload(c, wx.data(), wx.size());
                ok = c->sa_uw[1].w && c->sa_uw[1].b && c->sa_wx[1];
                // the same two layers as a fragment blob for the K-split wave-pair kernel (encoder_u.hip::sa_x_kernel, A/B)
                const std::string p4 = "pcd_backbone.sa4";
                std::vector<ChainLayer> layers = {{p4 + ".mlp_convs.1.weight", p4 + ".mlp_convs.1.bias", p4 + ".mlp_bns.1", 256, 256},
                                                  {p4 + ".mlp_convs.2.weight", p4 + ".mlp_convs.2.bias", p4 + ".mlp_bns.2", 512, 256}};
                std::vector<int> ident(256);
                for (int k = 0; k < 256; ++k) ident[k] = k;
                ok = ok && build_chain_blob(c, layers, ident, np, &c->sa_ub[1]);
            }
        }
    }
    if (ok) {
        Lin l{};
        ok = load_lin(c, "pcd_backbone.conv2", 3, 128, &l);
        c->head[1] = Dense{l.w, l.b};
    }
    // conditioning MLPs (plain)
    if (ok) {
        const HostTensor* pe = find(c, "sequence_pos_encoder.pe", {5000, 1, 128});
        ok = pe != nullptr;
        if (ok) c->cond.pe = upload(c, pe->data.data(), pe->data.size());
    }
    // the width of the text feature is the model's clip_dim (model/sdm.py:52-59): read off embed_text.0.weight (D/2, D)
    int D = 512;
    if (ok) {
        auto it = c->host.find("embed_text.0.weight");
        if (it != c->host.end() && it->second.shape.size() == 2) D = (int)it->second.shape[1];
        if (D < 2 || D > 1024 || D % 2) { fail(c, LSDM_ERR_PARAM, "embed_text.0.weight: text feature width %d is not an even number <= 1024", D); ok = false; }
        c->cond.text_dim = D;
    }
    ok = ok && load_lin(c, "embed_timestep.time_embed.0", 128, 128, &c->cond.time0) &&
         load_lin(c, "embed_timestep.time_embed.2", 128, 128, &c->cond.time2) && load_lin(c, "embed_text.0", D / 2, D, &c->cond.text0) &&
         load_lin(c, "embed_text.2", 256, D / 2, &c->cond.text2) && load_lin(c, "embed_text.4", 128, 256, &c->cond.text4) &&
         load_lin(c, "predict_cat.0", 64, 128, &c->cond.pcat0) && load_lin(c, "predict_cat.2", 32, 64, &c->cond.pcat2) &&
         load_lin(c, "predict_cat.4", C, 32, &c->cond.pcat4) && load_lin(c, "embed_cat.0", 32, C, &c->cond.ecat) &&
         load_lin(c, "translation_layer.0", 128, 160, &c->cond.tr0) && load_lin(c, "translation_layer.2", 12, 128, &c->cond.tr2);
    if (ok) {
        const HostTensor* wq = find(c, "attn_layer.q_proj_weight", {128, 128});
        const HostTensor* wk = find(c, "attn_layer.k_proj_weight", {128, 32});
        const HostTensor* ib = find(c, "attn_layer.in_proj_bias", {384});
        ok = wq && wk && ib;
        if (ok) {
            c->cond.attn_wq = upload(c, wq->data.data(), wq->data.size());
            c->cond.attn_wk = upload(c, wk->data.data(), wk->data.size());
            c->cond.attn_inb = upload(c, ib->data.data(), ib->data.size());
        }
    }
    // POSA decoder
    for (int i = 0; i < 3 && ok; ++i) {
        const std::string p = "human_backbone.de_spiral." + std::to_string(i);
        ok = load_lin(c, p + ".conv.layer", 64, i == 0 ? 3 : 64, &c->posa.lin[i]);
        const HostTensor* gw = ok ? find(c, p + ".norm.weight", {64}) : nullptr;
        const HostTensor* gb = ok ? find(c, p + ".norm.bias", {64}) : nullptr;
        ok = ok && gw && gb;
        if (ok) {
            c->posa.gn_w[i] = upload(c, gw->data.data(), 64);
            c->posa.gn_b[i] = upload(c, gb->data.data(), 64);
        }
    }
    ok = ok && load_lin(c, "human_backbone.de_spiral.3.layer", 3, 64, &c->posa.last) &&
         load_dense(c, "human_backbone.de_spiral.1.conv.layer.weight", "human_backbone.de_spiral.1.conv.layer.bias", "", 64, 64, &c->posa.d1) &&
         load_dense(c, "human_backbone.de_spiral.2.conv.layer.weight", "human_backbone.de_spiral.2.conv.layer.bias", "", 64, 64, &c->posa.d2);
    // pcd_attention
    if (ok) {
        const HostTensor* wq = find(c, "pcd_attention.q_proj_weight", {12, 12});
        const HostTensor* wk = find(c, "pcd_attention.k_proj_weight", {12, 3});
        const HostTensor* wv = find(c, "pcd_attention.v_proj_weight", {12, 3});
        const HostTensor* ib = find(c, "pcd_attention.in_proj_bias", {36});
        ok = wq && wk && wv && ib;
        if (ok) {
            c->sattn.wq = upload(c, wq->data.data(), 144);
            c->sattn.wk = upload(c, wk->data.data(), 36);
            c->sattn.wv = upload(c, wv->data.data(), 36);
            c->sattn.inb = upload(c, ib->data.data(), 36);
        }
    }
    ok = ok && load_lin(c, "pcd_attention.out_proj", 12, 12, &c->sattn.outp);
    // up-sampling + per-point denoiser
    ok = ok && load_lin(c, "upsampling_layer.0", 128, 1, &c->up0) &&
         load_dense(c, "upsampling_layer.2.weight", "upsampling_layer.2.bias", "", 512, 128, &c->up2);
    if (ok) {
        Lin l{};
        ok = load_lin(c, "upsampling_layer.4", 1024, 512, &l);
        c->den.up4_w = l.w;
        c->den.up4_b = l.b;
    }
    ok = ok && load_dense(c, "combine_extraction.0.weight", "combine_extraction.0.bias", "", 128, 256, &c->den.comb) &&
         load_lin(c, "input_process.pose_embedding.0", 64, 3, &c->den.pose0) &&
         load_dense(c, "input_process.pose_embedding.2.weight", "input_process.pose_embedding.2.bias", "", 128, 64, &c->den.pose2) &&
         load_dense(c, "input_process.combination_extraction.0.weight", "input_process.combination_extraction.0.bias", "", 192, 256, &c->den.cex0) &&
         load_dense(c, "input_process.combination_extraction.2.weight", "input_process.combination_extraction.2.bias", "", 128, 192, &c->den.cex2) &&
         load_dense(c, "output_process.pose_final.0.weight", "output_process.pose_final.0.bias", "", 64, 128, &c->den.fin0) &&
         load_lin(c, "output_process.pose_final.2", 3, 64, &c->den.fin2) && load_lin(c, "point_wise_trans_layer.0", 3, 15, &c->den.pwt);
    if (ok && c->prec != LSDM_PREC_F32) {
        const int np = c->prec == LSDM_PREC_BF16X3 ? 2 : 1;
        const HostTensor* w4 = find(c, "upsampling_layer.4.weight", {1024, 512});
        ok = w4 != nullptr;
        if (ok) {
            std::vector<uint16_t> hi(w4->data.size()), lo(w4->data.size());
            for (size_t i = 0; i < hi.size(); ++i) {
                hi[i] = f2bf(w4->data[i]);
                lo[i] = f2bf(w4->data[i] - bf2f(hi[i]));
            }
            c->den_h.up4_h = reinterpret_cast<const __bf16*>(upload(c, reinterpret_cast<const float*>(hi.data()), hi.size() / 2));
            c->den_h.up4_l = reinterpret_cast<const __bf16*>(upload(c, reinterpret_cast<const float*>(lo.data()), lo.size() / 2));
            c->den_h.up4_b = c->den.up4_b;
            c->den_h.pose0 = c->den.pose0;
            c->den_h.fin2 = c->den.fin2;
            c->den_h.pwt = c->den.pwt;
            ok = c->den_h.up4_h && c->den_h.up4_l;
        }
        ok = ok && load_dense_h(c, "combine_extraction.0.weight", "combine_extraction.0.bias", "", 128, 256, np, {}, &c->den_h.comb) &&
             load_dense_h(c, "input_process.pose_embedding.2.weight", "input_process.pose_embedding.2.bias", "", 128, 64, np, {}, &c->den_h.pose2) &&
             load_dense_h(c, "input_process.combination_extraction.0.weight", "input_process.combination_extraction.0.bias", "", 192, 256, np, {}, &c->den_h.cex0) &&
             load_dense_h(c, "input_process.combination_extraction.2.weight", "input_process.combination_extraction.2.bias", "", 128, 192, np, {}, &c->den_h.cex2) &&
             load_dense_h(c, "output_process.pose_final.0.weight", "output_process.pose_final.0.bias", "", 64, 128, np, {}, &c->den_h.fin0);
        if (ok) {
            const HostTensor* wf = find(c, "output_process.pose_final.2.weight", {3, 64});
            ok = wf != nullptr;
            if (ok) {
                std::vector<float> wpad((size_t)32 * 64, 0.0f);
                std::copy(wf->data.begin(), wf->data.end(), wpad.begin());
                std::vector<uint16_t> pk = pack_weight_h(wpad, 32, 64, np, {});
                c->den_h.fin2m.w = reinterpret_cast<const __bf16*>(upload(c, reinterpret_cast<const float*>(pk.data()), pk.size() / 2));
                c->den_h.fin2m.b = nullptr;
                ok = c->den_h.fin2m.w != nullptr;
            }
        }
        ok = ok && build_denoise_reg_blob(c, np, &c->den_r);
    }
    if (!ok) return c->err.empty() ? fail(c, LSDM_ERR_PARAM, "parameter upload failed") : LSDM_ERR_PARAM;
    c->finalized = true;
    return LSDM_OK;
}

int lsdm_set_schedule(lsdm_ctx* c, int T, const float* coef1, const float* coef2, const float* logvar, const float* sr,
                      const float* srm1, const float* acp, const float* acp_prev) {
    if (!c || T <= 0 || T > 5000 || !coef1 || !coef2 || !logvar || !sr || !srm1 || !acp || !acp_prev)
        return fail(c, LSDM_ERR_ARG, "lsdm_set_schedule: bad argument");
    DeviceGuard guard(c);
    if (!guard.ok) return LSDM_ERR_HIP;
    // kernels of earlier calls may still read the previous tables
    HIP_TRY(c, hipDeviceSynchronize());
    for (void* p : c->sched_allocs) (void)hipFree(p);
    c->sched_allocs.clear();
    c->sched_T = 0;
    c->sched.coef1 = upload(c, coef1, T, &c->sched_allocs);
    c->sched.coef2 = upload(c, coef2, T, &c->sched_allocs);
    c->sched.logvar = upload(c, logvar, T, &c->sched_allocs);
    c->sched.sr_acp = upload(c, sr, T, &c->sched_allocs);
    c->sched.srm1_acp = upload(c, srm1, T, &c->sched_allocs);
    c->sched.acp = upload(c, acp, T, &c->sched_allocs);
    c->sched.acp_prev = upload(c, acp_prev, T, &c->sched_allocs);
    if (!c->sched.coef1 || !c->sched.coef2 || !c->sched.logvar || !c->sched.sr_acp || !c->sched.srm1_acp || !c->sched.acp ||
        !c->sched.acp_prev)
        return LSDM_ERR_HIP;
    c->sched_T = T;
    return LSDM_OK;
}

int lsdm_reserve(lsdm_ctx* c, int B) {
    if (!c || B <= 0) return fail(c, LSDM_ERR_ARG, "lsdm_reserve: bad batch");
    if ((long long)B * kO > 65535) return fail(c, LSDM_ERR_ARG, "lsdm_reserve: batch %d exceeds 7281 chains per context (65535 clouds per launch)", B);
    DeviceGuard guard(c);
    if (!guard.ok) return LSDM_ERR_HIP;
    if (B <= c->cap_B) return LSDM_OK;
    HIP_TRY(c, hipDeviceSynchronize());
    for (void* p : c->ws_allocs) (void)hipFree(p);
    c->ws_allocs.clear();
    c->ws_named.clear();
    c->cap_B = 0;
    const size_t nc = (size_t)B * kO;
    bool ok = true;
    const char* lv[4] = {"1", "2", "3", "4"};
    for (int l = 0; l < 4 && ok; ++l)
        ok = (c->feat[l] = ws_alloc<float>(c, (std::string("feat") + lv[l]).c_str(), nc * kS[l] * kFeatC[l])) != nullptr;
    const int nn_n[4] = {64, 256, 1024, 1024};
    {
        // front sets: one allocation per tensor, [kGroup][...]; set 0 carries the plain workspace names (the parity
        // tests read them after single-step calls), the views of sets 1.. are derived per call (lsdm_ctx::fset)
        constexpr int G = lsdm_ctx::kGroup;
        auto grouped = [&](const std::string& name, size_t per_set, auto** slot0, size_t elem) -> bool {
            typedef typename std::remove_pointer<typename std::remove_pointer<decltype(slot0)>::type>::type T;
            T* base = ws_alloc<T>(c, name.c_str(), per_set * G);
            if (!base) return false;
            c->ws_named[name] = {base, (int64_t)per_set};
            *slot0 = base;
            (void)elem;
            return true;
        };
        lsdm_ctx::FrontSet& f0 = c->fs0;
        for (int l = 0; l < 4 && ok; ++l) {
            ok = grouped(std::string("fps_idx") + lv[l], nc * kS[l], &f0.idx[l], 4) && grouped(std::string("xyz") + lv[l], nc * kS[l] * 3, &f0.xyz[l], 4) &&
                 grouped(std::string("ball") + lv[l], nc * kS[l] * 32, &f0.ball[l], 4) &&
                 grouped(std::string("nn_idx") + lv[l], nc * nn_n[l] * 3, &f0.nn_idx[l], 4) &&
                 grouped(std::string("nn_w") + lv[l], nc * nn_n[l] * 3, &f0.nn_w[l], 4);
        }
        ok = ok && grouped("e0", (size_t)B * 256, &f0.e0, 4) && grouped("tr", nc * 12, &f0.tr, 4) && grouped("w", nc, &f0.w, 4) &&
             grouped("h2p", (size_t)B * 256 * 512, &f0.h2p, 4) && grouped("tbuf", (size_t)B, &f0.tbuf, 8);
    }
    ok = ok && (c->fpo4 = ws_alloc<float>(c, "fp4", nc * 64 * 256)) && (c->fpo3 = ws_alloc<float>(c, "fp3", nc * 256 * 256)) &&
         (c->fpo2 = ws_alloc<float>(c, "fp2", nc * 1024 * 128)) && (c->pcd = ws_alloc<float>(c, "pcd", nc * kN * 3)) &&
         (c->hm = ws_alloc<float>(c, "hm", (size_t)B * kN * 3)) && (c->o12 = ws_alloc<float>(c, "o12", nc * 12)) &&
         (c->xs = ws_alloc<float>(c, "xs", (size_t)B * kN * 3)) && (c->posa_ws = ws_alloc<float>(c, "posa_ws", (size_t)B * kPosaWsPerSample)) && (c->inv_ball = ws_alloc<int>(c, "inv_ball", nc * kN * 32)) &&
         (c->inv_F1 = ws_alloc<float>(c, "inv_f1", nc * kN * 64)) && (c->inv_nn8_d = ws_alloc<float>(c, "inv_nn8_d", nc * kN * 8)) &&
         (c->inv_nn8_i = ws_alloc<int>(c, "inv_nn8_i", nc * kN * 8)) && (c->sa_U[0] = ws_alloc<float>(c, "sa3_u", nc * 256 * 128)) &&
         (c->sa_V[0] = ws_alloc<float>(c, "sa3_v", nc * 256 * 128)) && (c->sa_U[1] = ws_alloc<float>(c, "sa4_u", nc * 64 * 256)) &&
         (c->sa_V[1] = ws_alloc<float>(c, "sa4_v", nc * 64 * 256)) && (c->sa_U[2] = ws_alloc<float>(c, "sa2_u", nc * 1024 * 64)) &&
         (c->sa_V[2] = ws_alloc<float>(c, "sa2_v", nc * 1024 * 64)) && (c->fp_ybuf[1] = ws_alloc<float>(c, "fp2_y", nc * 256 * 256)) &&
         (c->fp_ybuf[0] = ws_alloc<float>(c, "fp3_y", nc * 64 * 256)) && (c->fp_ybuf[2] = ws_alloc<float>(c, "fp4_y", nc * 16 * 256)) &&
         (c->h2x = ws_alloc<float>(c, "h2x", (size_t)B * 128 * 512)) && (c->ctext = ws_alloc<float>(c, "ctext", (size_t)B * kN * 128)) &&
         (c->ct_time = ws_alloc<float>(c, "ct_time", (size_t)(B > lsdm_ctx::kGroup ? B : lsdm_ctx::kGroup) * kN * 128)) &&
         (c->s_stash = ws_alloc<float>(c, "s_stash", (size_t)B * (kN / 32) * 6 * 16 * 64)) &&
         (c->fp2_plan = ws_alloc<int>(c, "fp2_plan", nc * fp_plan_ints(1024))) && (c->fp2_order = ws_alloc<int>(c, "fp2_order", nc * kN)) &&
         (c->inv_order = ws_alloc<int>(c, "inv_order", nc * kN));
    if (ok && c->deg_fast) {
        ok = (c->cls_flag = ws_alloc<unsigned char>(c, "cls_flag", nc)) && (c->cls_first = ws_alloc<float>(c, "cls_first", nc * 3)) &&
             (c->enc_map = ws_alloc<int>(c, "enc_map", nc)) && (c->rep_map = ws_alloc<int>(c, "rep_map", nc)) && (c->exp_src = ws_alloc<int>(c, "exp_src", nc)) &&
             (c->enc_clouds = ws_alloc<float>(c, "enc_clouds", nc * kN * 3)) && (c->rep_clouds = ws_alloc<float>(c, "rep_clouds", nc * kN * 3)) &&
             (c->rep_pcd = ws_alloc<float>(c, "rep_pcd", nc * kN * 3)) && (c->rep_starts = ws_alloc<int64_t>(c, "rep_starts", nc * 4));
        if (ok) {
            if (c->cls_host) (void)hipHostFree(c->cls_host);
            c->cls_host = nullptr;
            ok = hipHostMalloc(&c->cls_host, nc * 16) == hipSuccess;
        }
    }
    if (ok && c->sa_pair) {
        const char* nm[4] = {"sa1", "sa2", "sa3", "sa4"};
        for (int l = 0; l < 4 && ok; ++l)
            ok = (c->step_rows[l] = ws_alloc<int>(c, (std::string(nm[l]) + "_rows").c_str(), nc * kS[l] * 32)) &&
                 (c->step_meta[l] = reinterpret_cast<int4*>(ws_alloc<int>(c, (std::string(nm[l]) + "_meta").c_str(), nc * kS[l] * 4))) &&
                 (c->step_chunk[l] = ws_alloc<int>(c, (std::string(nm[l]) + "_chunks").c_str(), (size_t)sa_pair_chunks((int)(nc * kS[l]))));
        ok = ok && (c->step_cnt = ws_alloc<int>(c, "sa_step_items", 4));      // blocks of the last single step's lists, by SA level
    }
    if (!ok) return LSDM_ERR_HIP;
    c->cap_B = B;
    return LSDM_OK;
}

int lsdm_step(lsdm_ctx* c, int mode, float eta, int B, float* x, const float* mask, const int64_t* t, const float* given_objs,
              const float* given_cats, const float* text, const int64_t* starts, const float* noise, float* out_cat, float* out_x0,
              float* out_guiding, float* out_sample, void* stream) {
    int rc = check_ready(c, B, mode != LSDM_MODE_FORWARD);
    if (rc) return rc;
    if (mode < 0 || mode > 2) return fail(c, LSDM_ERR_ARG, "lsdm_step: bad mode %d", mode);
    if (!x || !mask || !t || !given_objs || !given_cats || !text || !starts || !out_cat || !out_x0 || !out_guiding ||
        (mode != LSDM_MODE_FORWARD && !out_sample))
        return fail(c, LSDM_ERR_ARG, "lsdm_step: NULL tensor");
    DeviceGuard guard(c);
    if (!guard.ok) return LSDM_ERR_HIP;
    return run_step(c, mode, eta, B, x, mask, t, given_objs, given_cats, text, starts, noise, out_cat, out_x0, out_guiding, out_sample,
                    static_cast<hipStream_t>(stream));
}

static int sample_steps_impl(lsdm_ctx* c, bool inv, int mode, float eta, int B, int t_first, int K, float* x, const float* mask,
                             const float* given_objs, const float* given_cats, const float* text, const int64_t* starts,
                             const float* noise, float* out_cat, float* out_x0, float* out_guiding, void* stream) {
    int rc = check_ready(c, B, true);
    if (rc) return rc;
    if (mode != LSDM_MODE_P_SAMPLE && mode != LSDM_MODE_DDIM) return fail(c, LSDM_ERR_ARG, "lsdm_sample_steps: bad mode %d", mode);
    if (K <= 0 || t_first >= c->sched_T || t_first - K + 1 < 0) return fail(c, LSDM_ERR_ARG, "lsdm_sample_steps: steps [%d..%d] outside schedule", t_first - K + 1, t_first);
    if (!x || !mask || !given_objs || !given_cats || !text || !starts || !noise || !out_cat || !out_x0 || !out_guiding)
        return fail(c, LSDM_ERR_ARG, "lsdm_sample_steps: NULL tensor");
    DeviceGuard guard(c);
    if (!guard.ok) return LSDM_ERR_HIP;
    // a multi-step call: stage outputs that feed only a fused consumer are not stored
    struct TapsOff { lsdm_ctx* c; explicit TapsOff(lsdm_ctx* c_) : c(c_) { c->keep_taps = false; } ~TapsOff() { c->keep_taps = true; } } taps_off(c);
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* cur = x;
    float* nxt = c->xs;
    const size_t xn = (size_t)B * kN * 3;
    const int nc_all = B * kO;
    // ---- degenerate clouds (lsdm_ctx::deg_fast): classified per call (one compare kernel + a 16-byte-per-cloud read-back: the only host round trip
    // of the call, in front of everything else); the encoder then runs on the live clouds only.  Applies to the schedule the sampler loop uses --
    // loop-invariant, group-batched, bf16 modes; every other path keeps all B x 9 clouds.
    struct EncReset { lsdm_ctx* c; ~EncReset() { c->enc_nc = 0; } } enc_reset{c};
    c->enc_nc = 0;
    const float* enc_clouds = given_objs;
    const int64_t* enc_starts = starts;
    int nu = 0;
    bool compacted = false;
    if (inv && c->deg_fast && c->cls_host && K > 1 && group_batched(c) && c->batch_den && c->den_reg && c->prec != LSDM_PREC_F32) {
        if (!ensure_group_ws(c, B)) return LSDM_ERR_HIP;
        hipLaunchKernelGGL(cloud_class_kernel, dim3(nc_all), dim3(256), 0, st, given_objs, c->cls_flag, c->cls_first);
        unsigned char* hflag = static_cast<unsigned char*>(c->cls_host);
        float* hfirst = reinterpret_cast<float*>(hflag + (((size_t)nc_all + 15) & ~(size_t)15));
        HIP_TRY(c, hipMemcpyAsync(hflag, c->cls_flag, (size_t)nc_all, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipMemcpyAsync(hfirst, c->cls_first, (size_t)nc_all * 3 * sizeof(float), hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        std::vector<int> live, reps, exp((size_t)nc_all);
        std::map<std::array<uint32_t, 3>, int> seen;      // distinct degenerate clouds by the bit pattern of their point
        for (int i = 0; i < nc_all; ++i) {
            if (!hflag[i]) { exp[i] = (int)live.size(); live.push_back(i); continue; }
            std::array<uint32_t, 3> key;
            memcpy(key.data(), hfirst + (size_t)i * 3, 12);
            auto it = seen.find(key);
            if (it == seen.end()) { it = seen.emplace(key, (int)reps.size()).first; reps.push_back(i); }
            exp[i] = -(it->second + 1);
        }
        c->deg_last[0] = (int)live.size(); c->deg_last[1] = nc_all - (int)live.size(); c->deg_last[2] = (int)reps.size();
        if (!reps.empty() && !live.empty()) {
            const int nl = (int)live.size();
            nu = (int)reps.size();
            // (pageable sources: the runtime stages them before the calls return)
            HIP_TRY(c, hipMemcpyAsync(c->enc_map, live.data(), live.size() * sizeof(int), hipMemcpyHostToDevice, st));
            HIP_TRY(c, hipMemcpyAsync(c->rep_map, reps.data(), reps.size() * sizeof(int), hipMemcpyHostToDevice, st));
            HIP_TRY(c, hipMemcpyAsync(c->exp_src, exp.data(), exp.size() * sizeof(int), hipMemcpyHostToDevice, st));
            const size_t need = (size_t)K * 4 * nl;
            if (need > c->enc_starts_cap) {
                HIP_TRY(c, hipStreamSynchronize(st));
                if (c->enc_starts) (void)hipFree(c->enc_starts);
                c->enc_starts = nullptr; c->enc_starts_cap = 0;
                HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->enc_starts), need * sizeof(int64_t)));
                c->enc_starts_cap = need;
            }
            hipLaunchKernelGGL(gather_clouds_kernel, dim3(nl), dim3(256), 0, st, given_objs, c->enc_map, c->enc_clouds);
            hipLaunchKernelGGL(gather_clouds_kernel, dim3(nu), dim3(256), 0, st, given_objs, c->rep_map, c->rep_clouds);
            hipLaunchKernelGGL(gather_starts_kernel, dim3((unsigned)((need + 255) / 256)), dim3(256), 0, st, starts, c->enc_map, nc_all, nl, (long long)need, c->enc_starts);
            hipLaunchKernelGGL(gather_starts_kernel, dim3((4 * nu + 255) / 256), dim3(256), 0, st, starts, c->rep_map, nc_all, nu, (long long)4 * nu, c->rep_starts);
            // the representatives through the encoder ONCE (the first step's draws; any would do), on the single-step workspace -- before the
            // per-call invariants below, which reuse parts of it (sa_U[2] / sa_V[2], the step lists) for the live clouds
            run_geometry(c, c->fs0, c->rep_clouds, c->rep_starts, nu, 1, st);
            run_encoder_dense(c, c->fs0, c->rep_clouds, nu, c->rep_pcd, st);
            enc_clouds = c->enc_clouds;
            enc_starts = c->enc_starts;
            c->enc_nc = nl;
            compacted = true;
        }
    }
    const int nc = compacted ? c->enc_nc : nc_all;        // clouds per step of the encoder
    // Steps are processed in groups of up to kGroup: the geometry of the whole group (it reads neither x nor anything a
    // step produces) is evaluated first, one launch per stage over group x B x 9 clouds, then the steps run in order.
    // (the cloud index of a batched geometry launch is gridDim.y: at most 65535 clouds per launch)
    int gmax = 65535 / nc < lsdm_ctx::kGroup ? (65535 / nc < 1 ? 1 : 65535 / nc) : lsdm_ctx::kGroup;
    // FPS runs one wave per cloud at 16 waves per CU (register-bound): a group whose clouds fit the chip in ONE round of waves
    // avoids a second, nearly empty round (B = 64: 7 steps = 4032 clouds on 4096 slots rather than 8 = 4608)
    if (c->fps_slots / nc >= 4 && c->fps_slots / nc < gmax) gmax = gmax / (c->fps_slots / nc) * (c->fps_slots / nc);
    if (c->group_cap > 0 && gmax > c->group_cap) gmax = c->group_cap;
    bool inv_pending = false;
    if (inv) {
        // once per call: everything of encoder level 1 that does not depend on the FPS order (geometry.hip::l1_remap_kernel).  With a side
        // stream it runs THERE, next to the conditioning and the geometry of the first group on the caller's stream (neither reads the
        // other's results); the caller's stream waits for it before the first stage that does (inv_pending).
        if (K > 1 && group_batched(c) && !ensure_group_ws(c, B)) return LSDM_ERR_HIP;
        hipStream_t si = st;
        if (K > 1 && group_batched(c) && lanes(c) > 1) {
            stream_fork(c, st);
            si = c->side[0];
        }
        const float r2 = (float)(kRadius[0] * kRadius[0]);
        prof_mark(c, K_BALL1, si);
        launch_ball_query(enc_clouds, enc_clouds, nc, 1024, 1024, r2, c->inv_ball, 0, si);
        prof_mark(c, K_SA1, si);
        if (c->prec == LSDM_PREC_F32) {
            launch_sa(0, nc, enc_clouds, enc_clouds, enc_clouds, c->inv_ball, c->sa[0], static_cast<float*>(c->inv_F1), si);
            prof_exec(c, K_SA1, nc * kKernelMacs[K_SA1]);
        } else {
            const bool items1 = c->sa_pair && (long long)nc * kN < kItemRowLimit;
            if (!items1) prof_exec(c, K_SA1, nc * xm::SA1);
            if (items1) {
                // (r = 0.1: seven distinct neighbours per group on the bench clouds -- half the blocks)
                launch_sa_pair_plan(1, nc * kN, kN, kN, c->inv_ball, nullptr, c->step_chunk[0], c->step_rows[0], c->step_meta[0], c->step_cnt, si);
                launch_sa_r_items(enc_clouds, c->step_rows[0], c->step_meta[0], c->step_cnt, c->sa_r[0].w, c->sa_r[0].b, static_cast<float*>(c->inv_F1), si);
                prof_exec_dev(c, K_SA1, c->step_cnt, xm::SA1_ITEM, si);
            } else launch_sa_r(2, 0, nc, enc_clouds, enc_clouds, enc_clouds, c->inv_ball, c->sa_r[0].w, c->sa_r[0].b, c->inv_F1, si);      // (sa1: two planes in every mode)
        }
        const int np = c->prec == LSDM_PREC_BF16X3 ? 2 : 1;      // (used by the split-bf16 branches below only)
        if (inv_in_source_order(c))   // sa2's split first layer per SOURCE point (rows of sa_U[2] / sa_V[2] are read through f.idx[0])
        {
            launch_lin_u(np, 1, nc * 1024, static_cast<const float*>(c->inv_F1), enc_clouds, c->sa_uw[2], c->sa_wx[2], c->sa_U[2], c->sa_V[2], si);
            prof_exec(c, K_SA1 + 1, nc * xm::SA2_UV);       // (timed inside the sa1 interval of the call's prelude)
        }
        // memo of sa2 (geometry.hip::memo_src_kernel): the stage's output of every source point with fewer than 32 points within r, once
        c->memo_ready = false;
        if (K > 1 && gmax > 1 && group_batched(c) && c->sa2_memo && c->gws.memo_F2) {
            const lsdm_ctx::GroupWs& gw = c->gws;
            const float r2b = (float)(kRadius[1] * kRadius[1]);
            prof_mark(c, K_BALL1 + 1, si);
            (void)hipMemsetAsync(gw.memo_cnt, 0, 2 * sizeof(int), si);
            launch_ball_query(enc_clouds, enc_clouds, nc, 1024, 1024, r2b, gw.memo_ball2, 0, si, nullptr, /*scan=*/true);
            launch_memo_src(gw.memo_ball2, (long long)nc * kN, gw.memo_unc, gw.memo_src_list, gw.memo_cnt, si);
            prof_mark(c, K_SA1 + 1, si);
            if (c->sa_pair && gw.memo_rows && (long long)nc * kN < kItemRowLimit) {
                // every uncrowded point's group has fewer than 32 distinct entries: two to a block where they fit (geometry.hip: sa_pair_kernel)
                launch_sa_pair_plan(1, (int)nc * kN, kN, kN, gw.memo_ball2, nullptr, gw.memo_chunks, gw.memo_rows, gw.memo_meta, gw.memo_cnt + 2, si, gw.memo_unc);
                launch_sa_u_items(np, 1, c->sa_U[2], c->sa_V[2], gw.memo_rows, gw.memo_meta, gw.memo_cnt + 2, c->sa_ub[2].w, c->sa_ub[2].b, gw.memo_F2, si);
                prof_exec_dev(c, K_SA1 + 1, gw.memo_cnt + 2, xm::SA2_CENT, si);     // the blocks that ran
            } else {
                launch_sa_u_list(np, 1024, c->sa_U[2], c->sa_V[2], gw.memo_ident, gw.memo_ball2, nullptr, c->sa_ub[2].w, c->sa_ub[2].b, gw.memo_F2, si, 0,
                                 gw.memo_src_list, gw.memo_cnt);
                prof_exec_dev(c, K_SA1 + 1, gw.memo_cnt, xm::SA2_CENT, si);     // sa2 once per call on the uncrowded source points
            }
            c->memo_ready = true;
        }
        prof_mark(c, K_NN1, si);
        launch_nn8(nc, enc_clouds, c->inv_nn8_d, c->inv_nn8_i, si);
        launch_morton_order(nc, enc_clouds, c->inv_order, si);      // fp2's walking order of the source points (encoder_d.hip)
        prof_mark(c, -1, si);
        if (si != st) {
            (void)hipEventRecord(c->ev_join[0], si);
            inv_pending = true;
        }
    }
    auto need_invariants = [&]() {
        if (inv_pending) { (void)hipStreamWaitEvent(st, c->ev_join[0], 0); inv_pending = false; }
    };
    for (int k0 = 0; k0 < K; k0 += gmax) {
        const int g = K - k0 < gmax ? K - k0 : gmax;
        hipLaunchKernelGGL(fill_steps_kernel, dim3((B * g + 63) / 64), dim3(64), 0, st, c->fs0.tbuf, (int64_t)(t_first - k0), B, g);
        run_conditioning(c, c->fs0, B, g, c->fs0.tbuf, mask, given_objs, given_cats, text, out_cat, /*reuse_hm=*/k0 > 0, st, /*shared=*/true);
        const bool grp = inv && (g > 1 || compacted) && group_batched(c);
        run_geometry(c, c->fs0, enc_clouds, enc_starts + (size_t)k0 * 4 * nc, nc, g, st, inv, grp && c->memo_ready);
        need_invariants();
        if (grp) {
            if (!ensure_group_ws(c, B)) return LSDM_ERR_HIP;
            run_encoder_group_front(c, c->fs0, enc_clouds, B, g, st);
        }
        if (grp && c->batch_den && c->den_reg && c->prec != LSDM_PREC_F32) {
            // Nothing of the encoder reads x: the rest of the encoder of every step of the group first (each step's output into its own
            // slot of gws.pcd), then the scene attention of the group in one launch and its denoiser steps in ONE launch -- a wave keeps
            // its 32 points' x on chip between the steps and forms each step's guidance from that step's slot (denoise_reg.hip); the
            // launch ramp, the weight-panel prologue and the tail are paid once per group instead of once per step.
            const lsdm_ctx::GroupWs& gw = c->gws;
            const size_t pstep = (size_t)nc_all * kN * 3, ostep = (size_t)nc_all * 12;
            const size_t estep = (size_t)nc * kN * 3;                      // the encoder's own output stride (compacted: the live clouds)
            float* const pcd_enc = compacted ? gw.pcd_c : gw.pcd;
            const int nl = lanes(c);
            if (nl > 1) stream_fork(c, st);
            for (int j = 0; j < g; ++j) {
                const int ln = j % nl;
                run_encoder_group_step(c, c->fset_steps(j, B), B, j, pcd_enc + (size_t)j * estep, ln ? c->side[ln - 1] : st, ln);
            }
            if (nl > 1) stream_join(c, st);
            prof_mark(c, K_SATTN, st, g);
            // (compacted: every (step, chain, slot) row block from its live slot or its call-constant representative)
            if (compacted) hipLaunchKernelGGL(expand_pcd_kernel, dim3(nc_all, g), dim3(256), 0, st, gw.pcd_c, c->rep_pcd, c->exp_src, nc_all, nc, gw.pcd);
            launch_scene_attn(c->sattn, gw.pcd, c->fs0.w, c->fs0.tr, B, gw.o12, st, g);
            prof_mark(c, K_DENOISE, st, g);
            launch_denoise_r(c->prec == LSDM_PREC_BF16X3 ? 2 : 1, c->den_r, c->den_h.pwt, c->sched, mode, eta, B, g, 0, cur, c->fs0.tbuf, mask, gw.pcd, c->fs0.w,
                             gw.o12, c->hm, c->ctext, c->ct_time, noise + (size_t)k0 * xn, c->s_stash, out_x0, k0 + g < K ? nullptr : out_guiding, nullptr,
                             c->num_cus, st, (long long)pstep, (long long)ostep);
            prof_exec(c, K_DENOISE, (double)B * (g * (xm::DEN_S + xm::DEN_PASS) + (k0 + g < K ? 0.0 : xm::DEN_PASS)));
            prof_mark(c, -1, st);
            rc = check_launch(c, "sample steps (group)");
            if (rc) return rc;
            continue;
        }
        for (int j = 0; j < g; ++j) {
            const int k = k0 + j;
            const lsdm_ctx::FrontSet f = c->fset_steps(j, B);
            if (grp) {
                // (LSDM_BATCH_DEN=0) the rest of the step's encoder, scene attention and denoiser, one step at a time
                const bool skip_guiding = c->den_reg && k != K - 1;
                run_encoder_group_step(c, f, B, j, c->pcd, st, 0);
                prof_mark(c, K_SATTN, st);
                launch_scene_attn(c->sattn, c->pcd, f.w, f.tr, B, c->o12, st);
                run_denoise(c, f, mode, eta, B, cur, f.tbuf, mask, noise + (size_t)k * xn, out_x0, skip_guiding ? nullptr : out_guiding, nxt, st, 1, c->shared_t());
                rc = check_launch(c, "sample step");
                if (rc) return rc;
                float* tmp = cur; cur = nxt; nxt = tmp;
                continue;
            }
            // saved_guiding_points is defined by the LAST forward of the call (Q12): the loop-invariant schedule runs the denoiser's
            // guiding-point pass only there (every other step's result would be overwritten unseen); the strict schedule keeps the
            // reference's literal work and runs it every step
            const bool skip_guiding = inv && c->den_reg && c->prec != LSDM_PREC_F32 && k != K - 1;
            run_back(c, f, mode, eta, B, cur, f.tbuf, mask, given_objs, noise + (size_t)k * xn, out_x0, skip_guiding ? nullptr : out_guiding, nxt, st, inv,
                     c->shared_t());
            rc = check_launch(c, "sample step");
            if (rc) return rc;
            float* tmp = cur; cur = nxt; nxt = tmp;
        }
    }
    if (cur != x) HIP_TRY(c, hipMemcpyAsync(x, cur, xn * sizeof(float), hipMemcpyDeviceToDevice, st));
    return LSDM_OK;
}

int lsdm_sample_steps(lsdm_ctx* c, int mode, float eta, int B, int t_first, int K, float* x, const float* mask, const float* given_objs,
                      const float* given_cats, const float* text, const int64_t* starts, const float* noise, float* out_cat,
                      float* out_x0, float* out_guiding, void* stream) {
    return sample_steps_impl(c, false, mode, eta, B, t_first, K, x, mask, given_objs, given_cats, text, starts, noise, out_cat, out_x0,
                             out_guiding, stream);
}

int lsdm_sample_steps_invariant(lsdm_ctx* c, int mode, float eta, int B, int t_first, int K, float* x, const float* mask,
                                const float* given_objs, const float* given_cats, const float* text, const int64_t* starts,
                                const float* noise, float* out_cat, float* out_x0, float* out_guiding, void* stream) {
    return sample_steps_impl(c, true, mode, eta, B, t_first, K, x, mask, given_objs, given_cats, text, starts, noise, out_cat, out_x0,
                             out_guiding, stream);
}

int lsdm_sample_steps_hoisted(lsdm_ctx* c, int mode, float eta, int B, int t_first, int K, float* x, const float* mask,
                              const float* given_objs, const float* given_cats, const float* text, const int64_t* starts,
                              const float* noise, float* out_cat, float* out_x0, float* out_guiding, void* stream) {
    int rc = check_ready(c, B, true);
    if (rc) return rc;
    if (mode != LSDM_MODE_P_SAMPLE && mode != LSDM_MODE_DDIM) return fail(c, LSDM_ERR_ARG, "lsdm_sample_steps_hoisted: bad mode %d", mode);
    if (K <= 0 || t_first >= c->sched_T || t_first - K + 1 < 0) return fail(c, LSDM_ERR_ARG, "lsdm_sample_steps_hoisted: steps [%d..%d] outside schedule", t_first - K + 1, t_first);
    if (!x || !mask || !given_objs || !given_cats || !text || !starts || !noise || !out_cat || !out_x0 || !out_guiding)
        return fail(c, LSDM_ERR_ARG, "lsdm_sample_steps_hoisted: NULL tensor");
    DeviceGuard guard(c);
    if (!guard.ok) return LSDM_ERR_HIP;
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* cur = x;
    float* nxt = c->xs;
    const size_t xn = (size_t)B * kN * 3;
    // once per call: everything that depends on (given_objs, text, cats, mask, FPS starts) only.  The scene attention needs
    // w / tr of the conditioning, which do not depend on t: the first group's conditioning is evaluated first.
    for (int k0 = 0; k0 < K; k0 += lsdm_ctx::kGroup) {
        const int g = K - k0 < lsdm_ctx::kGroup ? K - k0 : lsdm_ctx::kGroup;
        // per group: the t-dependent conditioning of its steps in one launch per stage
        hipLaunchKernelGGL(fill_steps_kernel, dim3((B * g + 63) / 64), dim3(64), 0, st, c->fs0.tbuf, (int64_t)(t_first - k0), B, g);
        run_conditioning(c, c->fs0, B, g, c->fs0.tbuf, mask, given_objs, given_cats, text, out_cat, /*reuse_hm=*/k0 > 0, st, /*shared=*/true);
        if (k0 == 0) {
            run_geometry(c, c->fs0, given_objs, starts, B * kO, 1, st);
            run_scene_conditioning(c, c->fs0, B, given_objs, st);
        }
        if (c->prec != LSDM_PREC_F32) {
            // the whole group in one launch: a tile's x stays on chip between its steps (denoise_bf.hip)
            run_denoise(c, c->fs0, mode, eta, B, cur, c->fs0.tbuf, mask, noise + (size_t)k0 * xn, out_x0,
                        (c->den_reg && k0 + g < K) ? nullptr : out_guiding, nullptr, st, g, true);
            rc = check_launch(c, "hoisted sample steps");
            if (rc) return rc;
            continue;
        }
        // per step: the fused denoiser + sampler update (the only stage that reads x)
        for (int j = 0; j < g; ++j) {
            const lsdm_ctx::FrontSet f = c->fset(j, B);
            run_denoise(c, f, mode, eta, B, cur, f.tbuf, mask, noise + (size_t)(k0 + j) * xn, out_x0, out_guiding, nxt, st);
            rc = check_launch(c, "hoisted sample step");
            if (rc) return rc;
            float* tmp = cur; cur = nxt; nxt = tmp;
        }
    }
    if (cur != x) HIP_TRY(c, hipMemcpyAsync(x, cur, xn * sizeof(float), hipMemcpyDeviceToDevice, st));
    return LSDM_OK;
}

int lsdm_fps(lsdm_ctx* c, const float* clouds, const int64_t* starts, int nc, int32_t* i1, int32_t* i2, int32_t* i3, int32_t* i4,
             float* x1, float* x2, float* x3, float* x4, void* stream) {
    if (!c || !clouds || !starts || nc <= 0) return fail(c, LSDM_ERR_ARG, "lsdm_fps: bad argument");
    DeviceGuard guard(c);
    if (!guard.ok) return LSDM_ERR_HIP;
    int* idx[4] = {i1, i2, i3, i4};
    float* xyz[4] = {x1, x2, x3, x4};
    launch_fps_chain(clouds, starts, nc, nc, idx, xyz, static_cast<hipStream_t>(stream));
    return check_launch(c, "fps");
}

int lsdm_ball_query(lsdm_ctx* c, const float* xyz, const float* new_xyz, int nc, int N, int S, double radius, int32_t* out, void* stream) {
    if (!c || !xyz || !new_xyz || !out || nc <= 0 || S <= 0) return fail(c, LSDM_ERR_ARG, "lsdm_ball_query: bad argument");
    if (N != 1024 && N != 256 && N != 64) return fail(c, LSDM_ERR_ARG, "lsdm_ball_query: N must be 1024, 256 or 64");
    DeviceGuard guard(c);
    if (!guard.ok) return LSDM_ERR_HIP;
    const float r2 = (float)(radius * radius);
    launch_ball_query(xyz, new_xyz, nc, N, S, r2, out, 0, static_cast<hipStream_t>(stream));
    return check_launch(c, "ball_query");
}

int lsdm_three_nn(lsdm_ctx* c, const float* fine, const float* coarse, int nc, int N, int S, int32_t* idx, float* w, void* stream) {
    if (!c || !fine || !coarse || !idx || !w || nc <= 0 || N <= 0) return fail(c, LSDM_ERR_ARG, "lsdm_three_nn: bad argument");
    if (S != 1024 && S != 256 && S != 64 && S != 16) return fail(c, LSDM_ERR_ARG, "lsdm_three_nn: S must be 1024, 256, 64 or 16");
    DeviceGuard guard(c);
    if (!guard.ok) return LSDM_ERR_HIP;
    launch_three_nn(fine, coarse, nc, N, S, idx, w, 0, static_cast<hipStream_t>(stream));
    return check_launch(c, "three_nn");
}

int lsdm_scene_encoder(lsdm_ctx* c, const float* clouds, const int64_t* starts, int nc, float* out, void* stream) {
    if (!c || !clouds || !starts || !out || nc <= 0) return fail(c, LSDM_ERR_ARG, "lsdm_scene_encoder: bad argument");
    if (!c->finalized) return fail(c, LSDM_ERR_STATE, "parameters not finalised");
    if ((nc + kO - 1) / kO > c->cap_B) return fail(c, LSDM_ERR_STATE, "%d clouds exceed the reserved workspace (%d samples)", nc, c->cap_B);
    DeviceGuard guard(c);
    if (!guard.ok) return LSDM_ERR_HIP;
    return run_scene_encoder(c, clouds, starts, nc, out, static_cast<hipStream_t>(stream));
}

int lsdm_human_backbone(lsdm_ctx* c, const float* given_objs, int B, float* hm, void* stream) {
    if (!c || !given_objs || !hm || B <= 0) return fail(c, LSDM_ERR_ARG, "lsdm_human_backbone: bad argument");
    if (!c->finalized) return fail(c, LSDM_ERR_STATE, "parameters not finalised");
    if (B > c->cap_B) return fail(c, LSDM_ERR_STATE, "batch %d exceeds reserved workspace %d (lsdm_reserve)", B, c->cap_B);
    DeviceGuard guard(c);
    if (!guard.ok) return LSDM_ERR_HIP;
    launch_posa(c->posa, given_objs, B, c->posa_ws, hm, static_cast<hipStream_t>(stream));
    return check_launch(c, "posa");
}

int lsdm_nn_distance(lsdm_ctx* c, const float* a, int na, const float* b, int nb, int batch, float* out_d2, int32_t* out_idx, void* stream) {
    if (!c || !a || !b || !out_d2 || na <= 0 || nb <= 0 || batch <= 0 || batch > 65535)
        return fail(c, LSDM_ERR_ARG, "lsdm_nn_distance: bad argument");
    DeviceGuard guard(c);
    if (!guard.ok) return LSDM_ERR_HIP;
    launch_nn_dist(a, na, b, nb, batch, out_d2, out_idx, static_cast<hipStream_t>(stream));
    return check_launch(c, "nn_distance");
}

int lsdm_profile_begin(lsdm_ctx* c) {
    if (!c) return LSDM_ERR_ARG;
    c->prof_cur = 0;
    for (double& v : c->prof_exec) v = 0.0;
    c->prof_on = true;
    return LSDM_OK;
}

int lsdm_profile_end(lsdm_ctx* c, int capacity, double* total_ms, int64_t* launches, double* macs_per_unit) {
    if (!c || !total_ms || !launches || capacity < K_COUNT) return fail(c, LSDM_ERR_ARG, "lsdm_profile_end: need capacity >= %d", (int)K_COUNT);
    c->prof_on = false;
    DeviceGuard guard(c);
    if (!guard.ok) return LSDM_ERR_HIP;
    HIP_TRY(c, hipDeviceSynchronize());
    for (int k = 0; k < K_COUNT; ++k) { total_ms[k] = 0.0; launches[k] = 0; if (macs_per_unit) macs_per_unit[k] = kKernelMacs[k]; }
    for (size_t i = 0; i + 1 < c->prof_cur; ++i) {
        const int id = c->prof_id[i];
        if (id < 0) continue;
        float ms = 0.0f;
        HIP_TRY(c, hipEventElapsedTime(&ms, c->prof_ev[i], c->prof_ev[i + 1]));
        total_ms[id] += ms;
        launches[id] += c->prof_w[i];
    }
    return K_COUNT;
}

int lsdm_profile_executed(lsdm_ctx* c, int capacity, double* executed_macs) {
    if (!c || !executed_macs || capacity < K_COUNT) return fail(c, LSDM_ERR_ARG, "lsdm_profile_executed: need capacity >= %d", (int)K_COUNT);
    for (int k = 0; k < K_COUNT; ++k) executed_macs[k] = c->prof_exec[k];
    return K_COUNT;
}

const char* lsdm_profile_kernel_name(int k) { return (k >= 0 && k < K_COUNT) ? kKernelNames[k] : ""; }

int lsdm_workspace_tensor(lsdm_ctx* c, const char* name, void** ptr, int64_t* numel) {
    if (!c || !name || !ptr || !numel) return fail(c, LSDM_ERR_ARG, "lsdm_workspace_tensor: bad argument");
    auto it = c->ws_named.find(name);
    if (it == c->ws_named.end()) return fail(c, LSDM_ERR_ARG, "no workspace tensor named %s", name);
    *ptr = it->second.first;
    *numel = it->second.second;
    return LSDM_OK;
}

}  // extern "C"
