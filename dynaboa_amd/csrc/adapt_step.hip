// Native frame stepper: ONE C call per adapted frame.
//
// The per-frame bilevel schedule of reference dynaboa_benchmark.py:126-193 (Adaptor.adaptation) - clone, inner_step x
// [lower-level loss -> learner.adapt -> inference], upper-level loss through the fast weights, zero_grad / backward /
// Adam.step, inference - issued from C++ over the engine (hmr_engine.hip), the SMPL kernels (smpl_lbs.hip), the loss
// head (losses.hip) and the flat-arena updates (optim.hip).  The Python Adaptor (dynaboa_amd/benchmark.py) stays the
// surface; in the configurations this file covers it forwards a frame here instead of walking the same ~1700 launches
// through torch.autograd + ctypes (9 ms of host time per frame there, ~2.5 us per launch here).  One stepper can also
// advance up to DYB_MAX_REPLICAS independent sequences in lockstep ("replicas", dyb_common.h): every launch of the chain
// then covers all of them.  dyb_stepper_* are re-entrant per stepper (own workspace, events, stream ordering).
//
// Same kernels, same order, same operands as the Python path => identical weights / Adam state (tests assert
// bit-equality); the metric reductions (MPJPE / PVE means) are this file's own kernels and agree to rounding.
//
// First-order MAML only (reference base_adaptor.py:119 first_order=True): theta_fast_{k+1} = theta_fast_k - fastlr*g_k,
// outer gradient = gradient of the upper-level loss AT the last fast weights (learn2learn clone() + adapt() with
// first_order=True: SURVEY Appendix B).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "dyb_common.h"

extern "C" {
size_t dyb_hmr_param_floats(const void*);
size_t dyb_hmr_act_floats(const void*);
size_t dyb_hmr_workspace_bytes(const void*);
int dyb_hmr_forward_train(void*, const float*, const float*, const float*, int, float*, void*, size_t, unsigned long long, unsigned long long, float,
                          hipStream_t);
long long dyb_hmr_act_offset_rotmat(const void*);
long long dyb_hmr_act_offset_state(const void*);
size_t dyb_lbs_saved_floats(int);
size_t dyb_lbs_bwd_workspace_bytes(int);
int dyb_lbs_fwd(const float* const*, const int* const*, const float*, int, const float*, float*, float*, float*, int, hipStream_t);
int dyb_lbs_bwd(const float* const*, const int* const*, const float*, const float*, const float*, const float*, float*, float*, int,
                int, void*, size_t, hipStream_t);
int dyb_regress_joints(const float*, const float*, float*, int, int, hipStream_t);
int dyb_rodrigues_fwd(const float*, float*, int, hipStream_t);
int dyb_frame_losses(const float*, const float*, int, const float*, int, const float*, const float*, const float*, const float*,
                     const float*, float, float, float, float*, float*, float*, int, float*, int, float*, int, void*, size_t,
                     hipStream_t);
int dyb_scale_add(const float*, const float*, const float*, float*, size_t, hipStream_t);
int dyb_head_grad_combine(const float*, const float*, const float*, const float*, const float*, const float*, const float*,
                          const float*, const float*, float*, float*, int, hipStream_t);
int dyb_fastweight_update(const float*, const float*, float*, float, size_t, hipStream_t);
int dyb_adam_step(float*, const float*, float*, float*, float, float, float, float, float, size_t, hipStream_t);
int dyb_ema_update(float*, const float*, float, size_t, hipStream_t);
int dyb_axpby(const float*, float*, float, float, size_t, hipStream_t);
int dyb_aux_loss_terms(int, int, int, float, const float*, const float*, int, const float*, int, const float*, const float*,
                       const float*, int, const float*, int, const float*, const float*, const float*, const float*, const float*,
                       const float*, float*, float*, float*, float*, float*, float*, float*, hipStream_t);
int dyb_hmr_feature_info(const void*, int, long long*, int*, int*);
}
int dyb_adam_step_rep(float*, const float*, float*, float*, float, float, const float*, const float*, float, size_t, hipStream_t);   // optim.hip
int dyb_adam_step_rep3(float*, const float*, const float*, const float*, float*, float*, float, float, const float*, const float*, float, size_t,
                       hipStream_t);
int dyb_fastweight_update3(const float*, const float*, const float*, const float*, float*, float, size_t, hipStream_t);
int dyb_fastweight_update_segs(const float*, const float*, float*, float, const DybFwSegs&, hipStream_t);       // optim.hip
int dyb_adam_step_rep3_ema(float*, const float*, const float*, const float*, float*, float*, float, float, const float*, const float*, float, size_t,
                           float*, float, hipStream_t);
int dyb_adam_step_segs(float*, const float*, float*, float*, float, float, const float*, const float*, float, const DybFwSegs&, hipStream_t);
int dyb_adam_write_scalars(float*, const float*, const float*, hipStream_t);

#define STATE_LD 160
#define NV 6890
#define NJ 49
#define RUN(x)                       \
  do {                               \
    int rc__ = (x);                  \
    if (rc__ != DYB_OK) return rc__; \
  } while (0)
#define HIPOK(x)                                  \
  do {                                            \
    if ((x) != hipSuccess) return DYB_ERR_LAUNCH; \
  } while (0)

// ---- metric record of one inference() (reference dynaboa_benchmark.py:204-262) ----------------------------------------
// pred17 / gt17m / gt17f: J_regressor_h36m @ vertices [B][17][3]; rec = pred14[B][14][3] | gt14[B][14][3] | mpjpe[B] | pve.
// pred14 = joints[H36M_TO_J14] - joints[0] (pelvis), gt from the male or the female mesh by gender (:221-233);
// mpjpe = mean_j |pred14 - gt14| (:236), pve = mean over batch and vertices of |gt_neutral - pred| (:244).
// PA-MPJPE needs an SVD per sample and is evaluated over all records at once (dyb_pa_mpjpe) when they are read.
__global__ __launch_bounds__(256) void metric_record_kernel(const float* __restrict__ pred17, const float* __restrict__ gt17m,
                                                            const float* __restrict__ gt17f, const long long* __restrict__ gender,
                                                            const int* __restrict__ j14, const float* __restrict__ pverts,
                                                            const float* __restrict__ gverts, float* __restrict__ rec, int B,
                                                            DybRep Rp) {
  DYB_REP_PROLOGUE(Rp);
  DYB_RB(Rp, pred17); DYB_RB(Rp, gt17m); DYB_RB(Rp, gt17f); DYB_RB(Rp, gender); DYB_RB(Rp, pverts); DYB_RB(Rp, gverts); DYB_RB(Rp, rec);
  __shared__ float s_err[64 * 14];
  __shared__ float s_red[4];
  const int t = threadIdx.x;
  float* pred14 = rec;
  float* gt14 = rec + (size_t)B * 42;
  float* mpjpe = rec + (size_t)B * 84;
  float* pve = mpjpe + B;
  for (int i = t; i < B * 14; i += 256) {
    const int b = i / 14, j = i - b * 14, src = j14[j];
    const float* g17 = (gender[b] == 1 ? gt17f : gt17m) + (size_t)b * 51;
    const float* p17 = pred17 + (size_t)b * 51;
    float e2 = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float p = p17[src * 3 + c] - p17[c], g = g17[src * 3 + c] - g17[c];
      pred14[i * 3 + c] = p;
      gt14[i * 3 + c] = g;
      e2 += (p - g) * (p - g);
    }
    s_err[i] = sqrtf(e2);
  }
  float acc = 0.f;
  const int nv = B * NV;
  for (int i = t; i < nv; i += 256) {
    const float dx = gverts[i * 3] - pverts[i * 3], dy = gverts[i * 3 + 1] - pverts[i * 3 + 1], dz = gverts[i * 3 + 2] - pverts[i * 3 + 2];
    acc += sqrtf(dx * dx + dy * dy + dz * dz);
  }
  acc = dyb_wave_sum(acc);
  if ((t & 63) == 0) s_red[t >> 6] = acc;
  __syncthreads();
  if (t < B) {
    float s = 0.f;
    for (int j = 0; j < 14; ++j) s += s_err[t * 14 + j];
    mpjpe[t] = s / 14.f;
  }
  if (t == 0) pve[0] = ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) / (float)nv;
}

// dst[0..3] = src[0..3] (a level's loss 4-vector into the frame's log row; replica-aware, unlike a device-to-device memcpy)
__global__ void copy4_kernel(const float* __restrict__ src, float* __restrict__ dst, DybRep Rp) {
  DYB_REP_PROLOGUE(Rp);
  DYB_RB(Rp, src); DYB_RB(Rp, dst);
  if (threadIdx.x < 4) dst[threadIdx.x] = src[threadIdx.x];
}
// the 16-float log row of one level of the full loss set: frame {s2d, shape prior, pose prior, total} | teacher {s2d, s3d, shape,
// pose, loss} | motion | labelled {s2d, s3d, shape, pose, loss} | level total = frame + wt*teacher + wm*motion + wl*labelled
__global__ void level_log_kernel(const float* __restrict__ frame4, const float* __restrict__ teach5, const float* __restrict__ motion5,
                                 const float* __restrict__ label5, float wt, float wm, float wl, float* __restrict__ row16, DybRep Rp) {
  DYB_REP_PROLOGUE(Rp);
  DYB_RB(Rp, frame4); DYB_RB(Rp, teach5); DYB_RB(Rp, motion5); DYB_RB(Rp, label5); DYB_RB(Rp, row16);
  if (threadIdx.x != 0) return;
  float total = frame4[3];
  for (int i = 0; i < 4; ++i) row16[i] = frame4[i];
  for (int i = 0; i < 5; ++i) row16[4 + i] = teach5 ? teach5[i] : 0.f;
  row16[9] = motion5 ? motion5[4] : 0.f;
  for (int i = 0; i < 5; ++i) row16[10 + i] = label5 ? label5[i] : 0.f;
  if (teach5) total += wt * teach5[4];
  if (motion5) total += wm * motion5[4];
  if (label5) total += wl * label5[4];
  row16[15] = total;
}
// cal_feature_diff (reference base_adaptor.py:211-219): cosine of each of the 15 feature pairs (flattened), one workgroup per
// feature; out[f] = cos_f, and - workgroup 12, the gate feature - the sequence number `seq` into out[15] AFTER its cosine, so a
// host that polls out[15] (device-visible pinned memory) reads a complete out[12].
struct FeatCosArgs {
  long long off[15];
  int rows[15], cols[15], ld[15];
};
__global__ __launch_bounds__(1024) void feat_cos_kernel(const float* __restrict__ A, const float* __restrict__ Bp, FeatCosArgs fa,
                                                        float eps, float* __restrict__ out_dev, volatile float* out_host, float seq,
                                                        DybRep Rp) {
  DYB_REP_PROLOGUE(Rp);
  DYB_RB(Rp, A); DYB_RB(Rp, Bp); DYB_RB(Rp, out_dev);
  if (out_host) out_host += 16 * dyb_rep;             // pinned host memory: 16 floats per physical replica
  // The three sums are accumulated in double (round 6): 1 - cos of feature 12 is ~1e-4 and the gate compares it with a threshold to a
  // few per cent, i.e. cos must be good to ~1e-6 - an fp32 sum over 2e5 ... 8e5 elements is not (torch's own fp32 evaluation returns
  // 1.000017 for feature 0: tests compare against the float64 evaluation of the reference's features, golden key gate_cos64)
  __shared__ double sm[16][3];
  const int f = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* a = A + fa.off[f];
  const float* b = Bp + fa.off[f];
  const long long n = (long long)fa.rows[f] * fa.cols[f];
  double ab = 0.0, aa = 0.0, bb = 0.0;
  for (long long i = t; i < n; i += 1024) {
    const long long r = i / fa.cols[f], c = i - r * fa.cols[f];
    const double x = (double)a[r * fa.ld[f] + c], y = (double)b[r * fa.ld[f] + c];
    ab += x * y; aa += x * x; bb += y * y;
  }
  for (int m = 32; m >= 1; m >>= 1) { ab += __shfl_xor(ab, m); aa += __shfl_xor(aa, m); bb += __shfl_xor(bb, m); }
  if (lane == 0) { sm[wave][0] = ab; sm[wave][1] = aa; sm[wave][2] = bb; }
  __syncthreads();
  if (t == 0) {
    double x = 0.0, y = 0.0, z = 0.0;
    for (int w = 0; w < 16; ++w) { x += sm[w][0]; y += sm[w][1]; z += sm[w][2]; }
    // torch cosine_similarity: x.y / (max(|x|, eps) max(|y|, eps)), rounded to fp32 once
    const float cs = (float)(x / (fmax(sqrt(y), (double)eps) * fmax(sqrt(z), (double)eps)));
    out_dev[f] = cs;
    if (out_host) {
      out_host[f] = cs;
      if (f == 12) {
        __threadfence_system();
        out_host[15] = seq;
      }
    }
  }
}

// replica r's frame inputs (separate caller tensors) into its staging area inside the workspace, one launch for all replicas
struct GatherArgs {
  const float* src[5][DYB_MAX_REPLICAS];     // up to five inputs per launch, per PHYSICAL replica (e.g. image, kp2d, gt_pose, gt_betas,
                                             // gender - int64 viewed as 2 floats); NULL: nothing to copy
  float* dst[5];                             // replica 0's staging buffers
  unsigned n[5];                             // floats per input
};
__global__ __launch_bounds__(256) void gather_inputs_kernel(GatherArgs a, DybRep Rp) {
  DYB_REP_PROLOGUE(Rp);
  const int k = blockIdx.y;
  const float* src = a.src[k][dyb_rep];
  if (!src) return;
  float* dst = a.dst[k];
  DYB_RB(Rp, dst);
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < a.n[k]; i += gridDim.x * 256) dst[i] = src[i];
}

// ---- one forward (+ backward) of HMR -> SMPL -> loss head ---------------------------------------------------------------
struct Pass {
  float* acts;
  char* ws;                 // engine workspace of the chain this pass runs on
  float *verts, *joints, *saved;
  float *losses, *drot_l, *dshape_l, *dcam_l, *djoints_l, *lws;      // dyb_frame_losses outputs
  float *djoints, *drot_s, *dbetas_s;                                // gradient assembly
  char* lbs_ws;
  float *d_rot, *d_state;
  float* pred17;
};

struct SideIssuer;
struct Stepper;
static void side_shutdown(Stepper& S);
struct Stepper {
  void* plan = nullptr;
  int B = 1, H = 224, W = 224;
  size_t n_params = 0, act_floats = 0, ws_bytes = 0, off_rot = 0, off_state = 0, lbs_saved = 0, lbs_wsb = 0;
  // options (doubles: the Python floats, so host-side scalars round exactly as in dynaboa_amd/optim.py / maml.py)
  int n_iter = 3, inner_step = 1, eval_lower = 1, use_side = 0, metrics = 1;
  // the full loss set (reference defaults): teacher / motion / labelled-exemplar terms, dynamic-BOA gate
  int full = 0, temporal_lower = 0, temporal_upper = 1, use_teacher = 1, use_motion = 1, interval = 5, mix_lower = 1, mix_upper = 1,
      dynamic = 1, optim_steps = 7;
  double teacher_w = 0.1, motion_w = 0.8, label_w = 0.1, alpha = 0.1, cos_thr = 3.1e-4;
  float* teacher = nullptr;           // teacher parameter arena (caller's)
  // train-mode teacher (the reference never calls teacher.eval(): base_adaptor.py:151-158 - its teacher forwards run with live
  // nn.Dropout(0.5) after fc1 / fc2, model/hmr.py:84,86): every teacher forward of a frame draws the masks of (drop_seed, drop_off + k),
  // k = teacher forwards issued since "drop_offset" was set - the same counter-based keys dynaboa_amd.hmr hands the autograd path
  int teacher_train = 0;
  unsigned long long drop_seed = 0, drop_off = 0;
  long long drop_used = 0;
  double drop_p = 0.5;
  volatile float* gate_host = nullptr;   // 16 floats of device-visible pinned host memory (caller's): 15 cosines + sequence number
  float* gate_log = nullptr;          // device: [loss_capacity][1 + optim_steps][16] cosines of every gate evaluation
  float* feat5_out = nullptr;         // [B][2048] the level's pooled feature for the retrieval callback
  int (*retrieve)(void*, int, const void**) = nullptr;
  int (*retrieve_rep)(void*, int, int, const void**) = nullptr;     // replica groups: (user, level, physical replica, out[5])
  void* retrieve_user = nullptr;
  FeatCosArgs fca{};
  float gate_seq = 0.f;
  Pass lvl0{}, ex{}, hist{}, teach{};
  float *grads2 = nullptr, *grads3 = nullptr, *zeros = nullptr, *ex_rot = nullptr;
  const float *lvl_g2 = nullptr, *lvl_g3 = nullptr;      // the current level's further gradient arenas (history pass, exemplar pass): summed by the update
  float *ext_rot = nullptr, *ext_shape = nullptr, *ext_cam = nullptr, *ext_joints = nullptr;     // aux-term gradients on the image pass
  float *exg_rot = nullptr, *exg_shape = nullptr, *exg_cam = nullptr, *exg_joints = nullptr;     // ... on the exemplar pass
  float *hg_cam = nullptr, *hg_joints = nullptr;                                                 // ... on the history pass
  float *vals_t = nullptr, *vals_m = nullptr, *vals_l = nullptr;
  int nrep = 1;                       // sequence replicas stepped in lockstep by every launch (dyb_common.h)
  int active[DYB_MAX_REPLICAS] = {};  // the replicas the next frame step covers (physical indices, ascending); default: all
  int nactive = 0;                    // 0 = all
  long long adam_t_rep[DYB_MAX_REPLICAS] = {};   // Adam steps taken, per replica (they differ once the dynamic loop or ragged streams do)
  size_t blob = 0;                    // workspace bytes of ONE replica
  float* in_stage[12] = {};           // staging of the frame inputs, in IN_* order (replica 0; nrep > 1 only)
  double lr = 3e-6, beta1 = 0.5, beta2 = 0.9, eps = 1e-8, fastlr = 8e-6, w2d = 10.0, wshape = 2e-6, wpose = 1e-4;
  long long adam_t = 0;
  // caller-owned state and tables (device pointers)
  float *theta = nullptr, *adam_m = nullptr, *adam_v = nullptr;
  const float* init_state = nullptr;
  const float* smpl_f[3][7] = {};         // neutral, male, female: the dyb_lbs_fwd table order
  const int* smpl_i[3][3] = {};
  const float *gmm_means = nullptr, *gmm_prec = nullptr, *gmm_logw = nullptr, *j_h36m = nullptr;
  const int* j14 = nullptr;
  float *records = nullptr, *loss_log = nullptr;
  const char* logs_base = nullptr;    // replica 0's block holding records | loss_log | gate_log | feat5_out (one replica arena for all four)
  size_t logs_bytes = 0;
  int record_capacity = 0, loss_capacity = 0;
  // workspace
  char* wsp = nullptr;
  size_t wsp_bytes = 0;
  bool bound = false, fresh = true;
  Pass main{}, fin{};
  float *theta_fast = nullptr, *grads = nullptr;
  // "fuse_fast" (round 6; frame-loss path): a lower level's unsplit throughput-form weight gradients write theta_next = theta_cur - fastlr * g
  // straight from their accumulators (DybWgradUpdateScope, igemm_conv.hip) - those spans never see a gradient in HBM nor the streaming
  // fast-weight pass, which covers what is left (fused_spans -> upd_spans).  theta_cur and theta_next must then be different buffers (a
  // layer's data gradient reads theta_cur on the chain while its weight gradient writes on the auxiliary stream): the levels ping-pong
  // between theta_fast and theta_fast2.
  float* theta_fast2 = nullptr;
  int fuse_fast = 1;
  std::vector<DybSpan> fused_spans, upd_spans;
  // "fuse_adam" (round 6; frame-loss path, replica groups, inner_step >= 1): the OUTER level's unsplit throughput-form weight gradients apply
  // Adam to theta / exp_avg / exp_avg_sq in place from their accumulators (igemm_tp.inc; the level is differentiated at the fast weights,
  // nobody reads theta meanwhile); the streaming Adam pass covers what is left.  The step's bias corrections are fixed BEFORE that
  // backward (adam_prepare) and live, per replica, in adam_sc (two floats of every replica's workspace copy).
  int fuse_adam = 1;
  int fuse_ema = 1;                // the teacher's EMA inside the Adam pass (default term set)
  float* adam_sc = nullptr;
  bool adam_prepared = false;
  float pre_ss[DYB_MAX_REPLICAS], pre_bc[DYB_MAX_REPLICAS];
  float *gt_rot = nullptr, *gt_verts[3] = {}, *gt_joints = nullptr, *gt_saved = nullptr, *gt17[2] = {};
  DybEvents* ev = nullptr;
  hipEvent_t e_theta = nullptr, e_side = nullptr, e_gt = nullptr;
  // weight updates by arena ranges beside the next forward ("upd_overlap", replica groups): the fast-weight steps and Adam are
  // streaming passes over [replicas] x 108 MB that nothing overlapped - now the range a forward needs first (stem .. layer2, 5 % of
  // the parameters) is updated on the main stream and the rest (layer3 | layer4 + regressor) on the auxiliary stream while the
  // forward's first layers run; the forward waits for each range right before its first reader (DybFwdGates)
  int upd_overlap = 1;
  // one sequence, full term set: the history pass and the exemplar pass of a level are independent of the frame pass until their gradients
  // are summed - they run on two streams of the stepper's own beside the chain ("par_passes"; batch-1 kernels fill a fraction of the chip)
  int par_passes = 1, par_max_replicas = DYB_MAX_REPLICAS;   // measured up to 32 per launch: 16: 290 -> 316, 32: 362 -> 377 frames/s (s30)
  hipStream_t par_stream[2] = {nullptr, nullptr};                                 // history pass | exemplar pass.  (A third stream for the teacher's
  // forward was measured and removed: 75.7 - 76.9 frames/s against 89.4 - 92.3 with two, profiles/r05_sessions.txt s25.)
  hipEvent_t par_ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // fork | hist fwd | hist head grads | hist bwd | label term | exemplar bwd | error-path tails of the two pass streams
  bool out_in_main = false;        // where the latest final inference lives (dyb_stepper_output): fin, or main after an odd number of shared steps
  int share_dyn_fwd = 1;           // dynamic loop: an extra step's upper level reuses the previous step's final inference as its forward
  int upd_blocks = 0;              // > 0: workgroup cap (all replicas together) of the ranged passes on the auxiliary stream (measured: no effect, 512 .. uncapped)
  hipEvent_t e_upd = nullptr;
  DybFwdGates gates{};
  bool gates_pending = false;
  int upd_late = 1;                // the last range (layer4 + regressor, 68 % of the parameters) is issued when the forward reaches layer3
  struct LateUpd {                 // what gates.late needs to issue it
    Stepper* S = nullptr;
    bool adam = false, ema = false;
    const float *g2 = nullptr, *g3 = nullptr;
    const float* p = nullptr;
    float* out = nullptr;
    hipStream_t aux = nullptr;
    DybRep scope{};
    float ss[DYB_MAX_REPLICAS], bc[DYB_MAX_REPLICAS];
  } late;
  size_t grp_bounds[2] = {0, 0};
  bool side_pending = false;
  // the final inference of a frame (side stream) is issued by the NEXT call, behind that frame's first level: the host cannot
  // run far ahead of the GPU (launches block when the queue is full), so issuing ~130 side-stream launches right after Adam
  // left the main queue empty for ~1 ms per frame (profiles/r02_s2_frame_timeline_S1.txt)
  struct Tail {
    bool on = false;
    const float* image = nullptr;
    const long long* gender = nullptr;
    int slot = 0;
    bool metrics = false;
  } tail;
  struct GtJob {
    bool on = false;
    const float *pose = nullptr, *betas = nullptr;
  } gtjob;
  hipStream_t tail_stream = nullptr;
  // "side_thread" (one sequence with a side stream): the side stream's launches - the previous frame's final forward, this frame's
  // ground-truth meshes, ~130 launches - are issued by a helper thread while the calling thread goes on with the frame's chain.  At one
  // sequence the frame is bound by the host's launch rate (~1 400 launches at ~8 us; profiles/r04_frame_timeline_S1.txt: the main queue
  // sat idle for 1.1 ms per frame while the calling thread issued them).  The calling thread waits for the helper before it first
  // touches anything the helper writes (e_gt, side_pending, the tail slot).
  int side_thread = 0;
  SideIssuer* issuer = nullptr;
  std::string err;
  // host-side issue time by section (always on: two clock reads per section), read back through dyb_stepper_get_f
  double h_fwd = 0, h_bwd = 0, h_head = 0, h_update = 0, h_tail = 0, h_total = 0;
  long long h_frames = 0;
};
struct HostTimer {
  double& acc;
  std::chrono::steady_clock::time_point t0;
  explicit HostTimer(double& a) : acc(a), t0(std::chrono::steady_clock::now()) {}
  ~HostTimer() { acc += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

static size_t a64(size_t v) { return (v + 63) & ~(size_t)63; }

// carve (or, with base == nullptr, size) the workspace
static size_t carve(Stepper& S, char* base) {
  size_t off = 0;
  auto take_f = [&](size_t nfloats) -> float* {
    float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
    off += a64(nfloats) * sizeof(float);
    return p;
  };
  auto take_b = [&](size_t nbytes) -> char* {
    char* p = base ? base + off : nullptr;
    off += a64((nbytes + 3) / 4) * sizeof(float);
    return p;
  };
  const size_t B = (size_t)S.B;
  auto pass = [&](Pass& P, bool with_grad) {
    P.acts = take_f(S.act_floats);
    P.ws = take_b(S.ws_bytes);
    P.verts = take_f(B * NV * 3);
    P.joints = take_f(B * NJ * 3);
    P.saved = take_f(S.lbs_saved);
    P.pred17 = take_f(B * 51);
    if (with_grad) {
      P.losses = take_f(4);
      P.drot_l = take_f(B * 216);
      P.dshape_l = take_f(B * 10);
      P.dcam_l = take_f(B * 3);
      P.djoints_l = take_f(B * NJ * 3);
      P.lws = take_f(B * 4);
      P.djoints = take_f(B * NJ * 3);
      P.drot_s = take_f(B * 216);
      P.dbetas_s = take_f(B * 10);
      P.lbs_ws = take_b(S.lbs_wsb);
      P.d_rot = take_f(B * 216);
      P.d_state = take_f(B * STATE_LD);
    }
  };
  pass(S.main, true);
  pass(S.fin, false);
  S.theta_fast = take_f(S.n_params);
  S.theta_fast2 = S.fuse_fast ? take_f(S.n_params) : nullptr;
  S.adam_sc = take_f(64);
  S.grads = take_f(S.n_params);
  S.gt_rot = take_f(B * 24 * 9);
  for (int i = 0; i < 3; ++i) S.gt_verts[i] = take_f(B * NV * 3);
  S.gt_joints = take_f(B * NJ * 3);
  S.gt_saved = take_f(S.lbs_saved);
  S.gt17[0] = take_f(B * 51);
  S.gt17[1] = take_f(B * 51);
  if (S.full) {
    S.lvl0 = S.main;                           // level 0 keeps its own activation arena: its features gate the dynamic loop
    S.lvl0.acts = take_f(S.act_floats);
    pass(S.ex, true);
    pass(S.hist, true);
    pass(S.teach, false);
    S.grads2 = take_f(S.n_params);
    S.grads3 = take_f(S.n_params);
    S.zeros = take_f(B * 216);
    S.ex_rot = take_f(B * 216);
    S.ext_rot = take_f(B * 216); S.ext_shape = take_f(B * 10); S.ext_cam = take_f(B * 3); S.ext_joints = take_f(B * NJ * 3);
    S.exg_rot = take_f(B * 216); S.exg_shape = take_f(B * 10); S.exg_cam = take_f(B * 3); S.exg_joints = take_f(B * NJ * 3);
    S.hg_cam = take_f(B * 3); S.hg_joints = take_f(B * NJ * 3);
    S.vals_t = take_f(8); S.vals_m = take_f(8); S.vals_l = take_f(8);
  }
  if (S.nrep > 1) {
    S.in_stage[0] = take_f(B * 3 * (size_t)S.H * S.W);
    S.in_stage[1] = take_f(B * NJ * 3);
    S.in_stage[2] = take_f(B * 72);
    S.in_stage[3] = take_f(B * 10);
    S.in_stage[4] = take_f(B * 2);
    if (S.full) {                                // history frame (image, kp2d) and the labelled exemplars (img, kp, pose, betas, pose_3d)
      S.in_stage[5] = take_f(B * 3 * (size_t)S.H * S.W);
      S.in_stage[6] = take_f(B * NJ * 3);
      S.in_stage[7] = take_f(B * 3 * (size_t)S.H * S.W);
      S.in_stage[8] = take_f(B * NJ * 3);
      S.in_stage[9] = take_f(B * 72);
      S.in_stage[10] = take_f(B * 10);
      S.in_stage[11] = take_f(B * 24 * 4);
    }
  }
  return off;
}

extern "C" int dyb_stepper_create(void* plan, int B, int H, int W, void** out) {
  DYB_REQUIRE(plan && out && B > 0 && B <= 64, DYB_ERR_ARG);
  Stepper* S = new Stepper();
  S->plan = plan; S->B = B; S->H = H; S->W = W;
  S->n_params = dyb_hmr_param_floats(plan);
  S->act_floats = dyb_hmr_act_floats(plan);
  S->ws_bytes = dyb_hmr_workspace_bytes(plan);
  S->off_rot = (size_t)dyb_hmr_act_offset_rotmat(plan);
  S->off_state = (size_t)dyb_hmr_act_offset_state(plan);
  S->lbs_saved = dyb_lbs_saved_floats(B);
  S->lbs_wsb = dyb_lbs_bwd_workspace_bytes(B);
  S->ev = dyb_hmr_events_create(plan);
  dyb_hmr_param_groups(plan, S->grp_bounds);
  if (const char* e = getenv("DYB_UPD_OVERLAP")) S->upd_overlap = atoi(e);       // (A/B runs; set_i "upd_overlap" afterwards wins)
  if (const char* e = getenv("DYB_UPD_BLOCKS")) S->upd_blocks = atoi(e);
  if (const char* e = getenv("DYB_SHARE_DYN_FWD")) S->share_dyn_fwd = atoi(e);
  if (const char* e = getenv("DYB_UPD_LATE")) S->upd_late = atoi(e);
  if (const char* e = getenv("DYB_FUSE_FAST")) S->fuse_fast = atoi(e);
  if (const char* e = getenv("DYB_FUSE_ADAM")) S->fuse_adam = atoi(e);
  if (const char* e = getenv("DYB_FUSE_EMA")) S->fuse_ema = atoi(e);
  if (const char* e = getenv("DYB_PAR_PASSES")) S->par_passes = atoi(e);
  if (const char* e = getenv("DYB_PAR_MAX_REPLICAS")) S->par_max_replicas = atoi(e);
  if (!S->ev || hipEventCreateWithFlags(&S->e_theta, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&S->e_side, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&S->e_gt, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&S->e_upd, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&S->gates.ev[0], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&S->gates.ev[1], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&S->gates.mid, hipEventDisableTiming) != hipSuccess) {
    delete S;
    return DYB_ERR_LAUNCH;
  }
  *out = S;
  return DYB_OK;
}
extern "C" void dyb_stepper_destroy(void* stepper) {
  Stepper* S = reinterpret_cast<Stepper*>(stepper);
  if (!S) return;
  side_shutdown(*S);
  dyb_hmr_events_destroy(S->ev);
  if (S->e_theta) (void)hipEventDestroy(S->e_theta);
  if (S->e_side) (void)hipEventDestroy(S->e_side);
  if (S->e_gt) (void)hipEventDestroy(S->e_gt);
  if (S->e_upd) (void)hipEventDestroy(S->e_upd);
  if (S->gates.ev[0]) (void)hipEventDestroy(S->gates.ev[0]);
  if (S->gates.ev[1]) (void)hipEventDestroy(S->gates.ev[1]);
  if (S->gates.mid) (void)hipEventDestroy(S->gates.mid);
  for (int i = 0; i < 8; ++i) if (S->par_ev[i]) (void)hipEventDestroy(S->par_ev[i]);
  for (int i = 0; i < 2; ++i) if (S->par_stream[i]) (void)hipStreamDestroy(S->par_stream[i]);
  delete S;
}

// ---- string-keyed setters (the whole configuration surface of the stepper; unknown key => DYB_ERR_ARG) -----------------
extern "C" int dyb_stepper_set_i(void* stepper, const char* key, long long v) {
  Stepper* S = reinterpret_cast<Stepper*>(stepper);
  DYB_REQUIRE(S && key, DYB_ERR_ARG);
  const std::string k(key);
  if (k == "n_iter") S->n_iter = (int)v;
  else if (k == "inner_step") S->inner_step = (int)v;
  else if (k == "eval_lower") S->eval_lower = (int)v;
  else if (k == "use_side") S->use_side = (int)v;
  else if (k == "upd_overlap") S->upd_overlap = (int)v;
  else if (k == "side_thread") S->side_thread = (int)v;
  else if (k == "upd_blocks") S->upd_blocks = (int)v;
  else if (k == "share_dyn_fwd") S->share_dyn_fwd = (int)v;
  else if (k == "par_passes") S->par_passes = (int)v;
  else if (k == "par_max_replicas") S->par_max_replicas = (int)v;
  else if (k == "upd_late") S->upd_late = (int)v;
  else if (k == "fuse_fast") { DYB_REQUIRE(!S->bound, DYB_ERR_ARG); S->fuse_fast = (int)v; }
  else if (k == "fuse_adam") S->fuse_adam = (int)v;
  else if (k == "fuse_ema") S->fuse_ema = (int)v;
  else if (k == "metrics") S->metrics = (int)v;
  else if (k == "adam_step") {
    S->adam_t = v;
    for (int r = 0; r < DYB_MAX_REPLICAS; ++r) S->adam_t_rep[r] = v;
  }
  else if (k.compare(0, 10, "adam_step_") == 0) {     // "adam_step_<replica>": replicas may join with different histories
    const int r = atoi(k.c_str() + 10);
    DYB_REQUIRE(r >= 0 && r < DYB_MAX_REPLICAS, DYB_ERR_ARG);
    S->adam_t_rep[r] = v;
    if (v > S->adam_t) S->adam_t = v;
  }
  else if (k == "full") S->full = (int)v;
  else if (k == "temporal_lower") S->temporal_lower = (int)v;
  else if (k == "temporal_upper") S->temporal_upper = (int)v;
  else if (k == "use_teacher") S->use_teacher = (int)v;
  else if (k == "teacher_train") S->teacher_train = (int)v;
  else if (k == "drop_seed") S->drop_seed = (unsigned long long)v;
  else if (k == "drop_offset") { S->drop_off = (unsigned long long)v; S->drop_used = 0; }
  else if (k == "use_motion") S->use_motion = (int)v;
  else if (k == "interval") S->interval = (int)v;
  else if (k == "mix_lower") S->mix_lower = (int)v;
  else if (k == "mix_upper") S->mix_upper = (int)v;
  else if (k == "dynamic") S->dynamic = (int)v;
  else if (k == "optim_steps") S->optim_steps = (int)v;
  else if (k == "replicas") {
    DYB_REQUIRE(v >= 1 && v <= DYB_MAX_REPLICAS && !S->bound, DYB_ERR_ARG);
    S->nrep = (int)v;
  }
  else if (k == "logs_bytes") S->logs_bytes = (size_t)v;
  else if (k == "record_capacity") S->record_capacity = (int)v;
  else if (k == "loss_capacity") S->loss_capacity = (int)v;
  else return DYB_ERR_ARG;
  return DYB_OK;
}
extern "C" int dyb_stepper_set_f(void* stepper, const char* key, double v) {
  Stepper* S = reinterpret_cast<Stepper*>(stepper);
  DYB_REQUIRE(S && key, DYB_ERR_ARG);
  const std::string k(key);
  if (k == "lr") S->lr = v;
  else if (k == "beta1") S->beta1 = v;
  else if (k == "beta2") S->beta2 = v;
  else if (k == "eps") S->eps = v;
  else if (k == "fastlr") S->fastlr = v;
  else if (k == "s2dloss_weight") S->w2d = v;
  else if (k == "shape_prior_weight") S->wshape = v;
  else if (k == "pose_prior_weight") S->wpose = v;
  else if (k == "teacherloss_weight") S->teacher_w = v;
  else if (k == "drop_p") { DYB_REQUIRE(v >= 0.0 && v < 1.0, DYB_ERR_ARG); S->drop_p = v; }
  else if (k == "motionloss_weight") S->motion_w = v;
  else if (k == "labelloss_weight") S->label_w = v;
  else if (k == "alpha") S->alpha = v;
  else if (k == "cos_sim_threshold") S->cos_thr = v;
  else return DYB_ERR_ARG;
  return DYB_OK;
}
// "smpl_<neutral|male|female>_<0..6>" = the seven float tables of dyb_lbs_fwd, "smpli_<...>_<0..2>" the three int tables
extern "C" int dyb_stepper_set_p(void* stepper, const char* key, const void* p) {
  Stepper* S = reinterpret_cast<Stepper*>(stepper);
  DYB_REQUIRE(S && key, DYB_ERR_ARG);
  const std::string k(key);
  static const char* const which[3] = {"neutral", "male", "female"};
  if (k == "theta") S->theta = (float*)p;
  else if (k == "adam_m") S->adam_m = (float*)p;
  else if (k == "adam_v") S->adam_v = (float*)p;
  else if (k == "init_state") S->init_state = (const float*)p;
  else if (k == "gmm_means") S->gmm_means = (const float*)p;
  else if (k == "gmm_precisions") S->gmm_prec = (const float*)p;
  else if (k == "gmm_log_weights") S->gmm_logw = (const float*)p;
  else if (k == "j_regressor_h36m") S->j_h36m = (const float*)p;
  else if (k == "j14") S->j14 = (const int*)p;
  else if (k == "logs_base") S->logs_base = (const char*)p;
  else if (k == "records") S->records = (float*)p;
  else if (k == "loss_log") S->loss_log = (float*)p;
  else if (k == "teacher") S->teacher = (float*)p;
  else if (k == "gate_host") S->gate_host = (volatile float*)p;
  else if (k == "gate_log") S->gate_log = (float*)p;
  else if (k == "feat5_out") S->feat5_out = (float*)p;
  else if (k == "retrieve_fn") S->retrieve = (int (*)(void*, int, const void**))p;
  else if (k == "retrieve_rep_fn") S->retrieve_rep = (int (*)(void*, int, int, const void**))p;
  else if (k == "retrieve_user") S->retrieve_user = (void*)p;
  else {
    for (int g = 0; g < 3; ++g) {
      const std::string pf = std::string("smpl_") + which[g] + "_", pi = std::string("smpli_") + which[g] + "_";
      if (k.compare(0, pf.size(), pf) == 0 && k.size() == pf.size() + 1 && k.back() >= '0' && k.back() <= '6') {
        S->smpl_f[g][k.back() - '0'] = (const float*)p;
        return DYB_OK;
      }
      if (k.compare(0, pi.size(), pi) == 0 && k.size() == pi.size() + 1 && k.back() >= '0' && k.back() <= '2') {
        S->smpl_i[g][k.back() - '0'] = (const int*)p;
        return DYB_OK;
      }
    }
    return DYB_ERR_ARG;
  }
  return DYB_OK;
}
extern "C" long long dyb_stepper_get_i(const void* stepper, const char* key) {
  const Stepper* S = reinterpret_cast<const Stepper*>(stepper);
  if (!S || !key) return -1;
  const std::string k(key);
  if (k == "adam_step") return S->adam_t;
  if (k.compare(0, 10, "adam_step_") == 0) {          // "adam_step_<replica>"
    const int r = atoi(k.c_str() + 10);
    return (r >= 0 && r < S->nrep) ? S->adam_t_rep[r] : -1;
  }
  if (k == "drop_used") return S->drop_used;
  if (k == "record_floats") return (long long)a64((size_t)S->B * 85 + 1);
  if (k == "loss_floats") return (S->full ? 16 : 4) * (long long)(S->inner_step + 1 + (S->full && S->dynamic ? S->optim_steps : 0));
  if (k == "slots_per_frame") return (S->eval_lower ? S->inner_step : 0) + 1 + (S->full && S->dynamic ? S->optim_steps : 0);
  return -1;
}
// host issue time (ms, accumulated over h_frames frame steps of the frame-loss path) by section:
// "host_ms_forward", "host_ms_backward", "host_ms_head", "host_ms_update", "host_ms_tail", "host_ms_total", "host_frames"
extern "C" double dyb_stepper_get_f(const void* stepper, const char* key) {
  const Stepper* S = reinterpret_cast<const Stepper*>(stepper);
  if (!S || !key) return -1.0;
  const std::string k(key);
  if (k == "host_ms_forward") return S->h_fwd;
  if (k == "host_ms_backward") return S->h_bwd;
  if (k == "host_ms_head") return S->h_head;
  if (k == "host_ms_update") return S->h_update;
  if (k == "host_ms_tail") return S->h_tail;
  if (k == "host_ms_total") return S->h_total;
  if (k == "host_frames") return (double)S->h_frames;
  return -1.0;
}
extern "C" size_t dyb_stepper_workspace_bytes(void* stepper) {
  Stepper* S = reinterpret_cast<Stepper*>(stepper);
  if (!S) return 0;
  Stepper tmp = *S;
  return a64(carve(tmp, nullptr) / 4) * 4 * (size_t)S->nrep;
}
// `ws` must stay valid (and untouched by others) while the stepper lives; its gradient staging is zeroed here, on `stream`.
extern "C" int dyb_stepper_bind_workspace(void* stepper, void* ws, size_t bytes, hipStream_t st) {
  Stepper* S = reinterpret_cast<Stepper*>(stepper);
  DYB_REQUIRE(S && ws, DYB_ERR_ARG);
  S->blob = a64(carve(*S, nullptr) / 4) * 4;
  DYB_REQUIRE(bytes >= S->blob * (size_t)S->nrep, DYB_ERR_WORKSPACE);
  S->wsp = reinterpret_cast<char*>(ws);
  S->wsp_bytes = bytes;
  carve(*S, S->wsp);                 // replica 0's pointers; replica r's = + r * blob (resolved inside the kernels)
  // the engine overwrites every tensor span of the gradient arena and only columns 144..156 of d_state: pads stay zero
  for (int r = 0; r < S->nrep; ++r) {
    HIPOK(hipMemsetAsync(reinterpret_cast<char*>(S->grads) + (size_t)r * S->blob, 0, S->n_params * sizeof(float), st));
    HIPOK(hipMemsetAsync(reinterpret_cast<char*>(S->main.d_state) + (size_t)r * S->blob, 0, (size_t)S->B * STATE_LD * sizeof(float), st));
  }
  if (S->full) {
    for (int r = 0; r < S->nrep; ++r) {
      auto at = [&](float* p0) { return reinterpret_cast<char*>(p0) + (size_t)r * S->blob; };
      HIPOK(hipMemsetAsync(at(S->zeros), 0, (size_t)S->B * 216 * sizeof(float), st));
      HIPOK(hipMemsetAsync(at(S->grads2), 0, S->n_params * sizeof(float), st));
      HIPOK(hipMemsetAsync(at(S->grads3), 0, S->n_params * sizeof(float), st));
      HIPOK(hipMemsetAsync(at(S->ex.d_state), 0, (size_t)S->B * STATE_LD * sizeof(float), st));
      HIPOK(hipMemsetAsync(at(S->hist.d_state), 0, (size_t)S->B * STATE_LD * sizeof(float), st));
    }
    for (int f = 0; f < 15; ++f) {
      long long off = 0;
      int dims[4] = {0, 0, 0, 0}, rs = 0;
      RUN(dyb_hmr_feature_info(S->plan, f, &off, dims, &rs));
      S->fca.off[f] = off;
      if (dims[2] > 0) {                                  // [B][H][W][C] dense
        S->fca.rows[f] = 1; S->fca.cols[f] = dims[0] * dims[1] * dims[2] * dims[3]; S->fca.ld[f] = S->fca.cols[f];
      } else {                                            // [B][cols] with a row stride
        S->fca.rows[f] = dims[0]; S->fca.cols[f] = dims[1]; S->fca.ld[f] = rs;
      }
    }
  }
  S->bound = true;
  return DYB_OK;
}

static int check_ready(const Stepper& S) {
  DYB_REQUIRE(S.bound && S.theta && S.adam_m && S.adam_v && S.init_state && S.gmm_means && S.gmm_prec && S.gmm_logw, DYB_ERR_ARG);
  for (int i = 0; i < 7; ++i) DYB_REQUIRE(S.smpl_f[0][i], DYB_ERR_ARG);
  for (int i = 0; i < 3; ++i) DYB_REQUIRE(S.smpl_i[0][i], DYB_ERR_ARG);
  if (S.metrics) {
    DYB_REQUIRE(S.j_h36m && S.j14 && S.records, DYB_ERR_ARG);
    for (int g = 1; g < 3; ++g) {
      for (int i = 0; i < 7; ++i) DYB_REQUIRE(S.smpl_f[g][i], DYB_ERR_ARG);
      for (int i = 0; i < 3; ++i) DYB_REQUIRE(S.smpl_i[g][i], DYB_ERR_ARG);
    }
  }
  DYB_REQUIRE(S.inner_step >= 0 && S.inner_step <= 16 && S.n_iter >= 1 && S.n_iter <= 3, DYB_ERR_UNSUPPORTED);
  return DYB_OK;
}

static int settle_update(Stepper& S, hipStream_t st);
// HMR forward at `theta` -> rotmat / shape / cam in the pass's arena -> SMPL (neutral) vertices + 49 joints
static int pass_forward(Stepper& S, Pass& P, const float* theta, const float* image, hipStream_t st, bool chain = true) {
  // (a ranged weight update may still be running on the auxiliary stream: this forward - the first reader of the new weights -
  // waits for each range where it first reads it).  chain = false: the side stream's forward (possibly issued by the helper thread)
  // - never a consumer of a ranged update (replica groups have no side stream) and it leaves the chain's gate state alone
  const DybFwdGates* gates = (chain && S.gates_pending) ? &S.gates : nullptr;
  {
    const int rc = dyb_hmr_forward_plain(S.plan, theta, image, S.init_state, S.n_iter, P.acts, P.ws, S.ws_bytes, st, gates);
    if (rc != DYB_OK) {
      // the forward may have stopped before it waited for / issued the later ranges: make the stream wait for the whole update so
      // that nothing stale is read by whatever the caller does next, then report the failure
      if (gates) (void)settle_update(S, st);
      return rc;
    }
    // consumed: every range's gate was waited for inside the forward (the deferred one issued there)
    if (gates) { DYB_REQUIRE(!S.gates.late, DYB_ERR_LAUNCH); S.gates_pending = false; }
  }
  const float* rot = P.acts + S.off_rot;
  const float* state = P.acts + S.off_state;
  return dyb_lbs_fwd(S.smpl_f[0], S.smpl_i[0], state + 144, STATE_LD, rot, P.verts, P.joints, P.saved, S.B, st);
}
// frame-loss head (reference base_adaptor.py:229-240 / :279-289): value + gradient in one launch
static int pass_frame_head(Stepper& S, Pass& P, const float* kp2d, hipStream_t st) {
  const float* rot = P.acts + S.off_rot;
  const float* state = P.acts + S.off_state;
  return dyb_frame_losses(rot, state + 144, STATE_LD, state + 154, STATE_LD, P.joints, kp2d, S.gmm_means, S.gmm_prec, S.gmm_logw,
                          (float)S.w2d, (float)S.wshape, (float)S.wpose, P.losses, P.drot_l, P.dshape_l, 10, P.dcam_l, 3, P.djoints_l,
                          S.B, P.lws, (size_t)S.B * 16, st);
}
// gradient of the pass's loss total w.r.t. `theta` into `grads` (the same five C calls as fused_level._LevelFunction.backward)
static int pass_backward(Stepper& S, Pass& P, const float* theta, float* grads, hipStream_t st, hipStream_t aux) {
  RUN(dyb_scale_add(nullptr, P.djoints_l, nullptr, P.djoints, (size_t)S.B * NJ * 3, st));
  RUN(dyb_lbs_bwd(S.smpl_f[0], S.smpl_i[0], P.acts + S.off_rot, P.saved, P.djoints, nullptr, P.drot_s, P.dbetas_s, 10, S.B, P.lbs_ws,
                  S.lbs_wsb, st));
  RUN(dyb_head_grad_combine(nullptr, P.drot_l, P.drot_s, nullptr, P.dshape_l, P.dbetas_s, nullptr, P.dcam_l, nullptr, P.d_rot,
                            P.d_state, S.B, st));
  return dyb_hmr_backward_ev(S.plan, theta, P.acts, P.d_rot, P.d_state, S.n_iter, grads, P.ws, S.ws_bytes, st, aux, S.ev);
}
// ground-truth side of the metrics (depends on the batch only): male / female / neutral meshes from the axis-angle pose
static int gt_meshes(Stepper& S, const float* gt_pose, const float* gt_betas, hipStream_t st) {
  RUN(dyb_rodrigues_fwd(gt_pose, S.gt_rot, S.B * 24, st));
  for (int g = 0; g < 3; ++g)
    RUN(dyb_lbs_fwd(S.smpl_f[g], S.smpl_i[g], gt_betas, 10, S.gt_rot, S.gt_verts[g], S.gt_joints, S.gt_saved, S.B, st));
  RUN(dyb_regress_joints(S.j_h36m, S.gt_verts[1], S.gt17[0], 17, S.B, st));
  return dyb_regress_joints(S.j_h36m, S.gt_verts[2], S.gt17[1], 17, S.B, st);
}
static int record_metrics(Stepper& S, Pass& P, const long long* gender, int slot, hipStream_t st) {
  DYB_REQUIRE(slot >= 0 && slot < S.record_capacity, DYB_ERR_ARG);
  RUN(dyb_regress_joints(S.j_h36m, P.verts, P.pred17, 17, S.B, st));
  float* rec = S.records + (size_t)slot * a64((size_t)S.B * 85 + 1);
  const DybRep& Rp = dyb_rep_current();
  hipLaunchKernelGGL(metric_record_kernel, dim3(1, 1, Rp.n), dim3(256), 0, st, P.pred17, S.gt17[0], S.gt17[1], gender, S.j14, P.verts,
                     S.gt_verts[0], rec, S.B, Rp);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}

// One adapted frame.  image [B][3][H][W], kp2d [B][49][3]; gt_pose [B][72] / gt_betas [B][10] / gender [B] (int64) feed the
// metric records only (NULL with metrics = 0).  Records: inner_step (eval_lower) + 1 slots starting at record_slot, in
// schedule order - after inner step 0, 1, ..., then the final one; losses: loss_slot*(inner_step+1) 4-vectors
// (s2d, shape prior, pose prior, weighted total) per level.  side (may be NULL): stream for the final no-grad forward and
// its metrics, overlapped with the next frame's first level; the call orders the weight hazards itself.
// issue what the side stream owes: the previous frame's final forward + record, then this frame's ground-truth meshes
static int issue_side_work(Stepper& S, hipStream_t side) {
  if (S.tail.on) {
    HIPOK(hipStreamWaitEvent(side, S.e_theta, 0));
    RUN(pass_forward(S, S.fin, S.theta, S.tail.image, side, false));
    if (S.tail.metrics) RUN(record_metrics(S, S.fin, S.tail.gender, S.tail.slot, side));
    HIPOK(hipEventRecord(S.e_side, side));
    S.side_pending = true;
    S.tail.on = false;
  }
  if (S.gtjob.on) {
    RUN(gt_meshes(S, S.gtjob.pose, S.gtjob.betas, side));
    HIPOK(hipEventRecord(S.e_gt, side));
    S.gtjob.on = false;
  }
  return DYB_OK;
}
struct SideIssuer {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  bool has_job = false, quit = false;
  std::atomic<int> busy{0};          // 1 from the post until every launch of the job has been issued
  int rc = DYB_OK;
  int device = 0;
  Stepper* S = nullptr;
  hipStream_t side = nullptr;
  void main() {
    (void)hipSetDevice(device);
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return has_job || quit; });
        if (quit) return;
        has_job = false;
      }
      rc = issue_side_work(*S, side);
      busy.store(0, std::memory_order_release);
    }
  }
};
// the helper is idle and what it wrote is visible; returns its last result
static int side_wait(Stepper& S) {
  SideIssuer* I = S.issuer;
  if (!I) return DYB_OK;
  while (I->busy.load(std::memory_order_acquire)) std::this_thread::yield();
  const int rc = I->rc;
  I->rc = DYB_OK;
  return rc;
}
// issue_side_work, by the helper thread when there is one
static int side_post(Stepper& S, hipStream_t side) {
  if (!S.side_thread || dyb_rep_current().n != 1) return issue_side_work(S, side);
  if (!S.issuer) {
    S.issuer = new SideIssuer();
    S.issuer->S = &S;
    if (hipGetDevice(&S.issuer->device) != hipSuccess) S.issuer->device = 0;
    S.issuer->th = std::thread([I = S.issuer] { I->main(); });
  }
  RUN(side_wait(S));
  SideIssuer* I = S.issuer;
  I->side = side;
  I->busy.store(1, std::memory_order_release);
  {
    std::lock_guard<std::mutex> lk(I->mu);
    I->has_job = true;
  }
  I->cv.notify_one();
  return DYB_OK;
}
static void side_shutdown(Stepper& S) {
  SideIssuer* I = S.issuer;
  if (!I) return;
  (void)side_wait(S);
  {
    std::lock_guard<std::mutex> lk(I->mu);
    I->quit = true;
  }
  I->cv.notify_one();
  if (I->th.joinable()) I->th.join();
  delete I;
  S.issuer = nullptr;
}
// out[lo, hi) = p - fastlr * (grads [+ g2 + g3]) minus the spans the level's weight gradients updated themselves (S.upd_spans, sorted)
static int fastweight_range(Stepper& S, const float* p, float* out, const float* g2, const float* g3, size_t lo, size_t hi, hipStream_t s) {
  bool any = false;
  for (const DybSpan& sp : S.upd_spans)
    if (sp.off < hi && sp.off + sp.n > lo) { any = true; break; }
  if (!any)
    return dyb_fastweight_update3(p + lo, S.grads + lo, g2 ? g2 + lo : nullptr, g3 ? g3 + lo : nullptr, out + lo, (float)S.fastlr, hi - lo, s);
  DYB_REQUIRE(!g2 && !g3 && lo % 4 == 0 && hi % 4 == 0, DYB_ERR_UNSUPPORTED);     // (the fused form exists for single-pass levels only)
  DybFwSegs t{};
  auto flush = [&]() -> int {
    if (t.n == 0) return DYB_OK;
    const int rc = dyb_fastweight_update_segs(p, S.grads, out, (float)S.fastlr, t, s);
    t.n = 0;
    return rc;
  };
  auto add = [&](size_t a, size_t b) -> int {
    if (a >= b) return DYB_OK;
    DYB_REQUIRE(a % 4 == 0 && b % 4 == 0 && (b - a) / 4 < 0xffffffffull && a / 4 < 0xffffffffull, DYB_ERR_UNSUPPORTED);
    if (t.n == DYB_FW_MAX_SEGS) RUN(flush());
    t.start4[t.n] = (unsigned)(a / 4); t.count4[t.n] = (unsigned)((b - a) / 4); ++t.n;
    return DYB_OK;
  };
  size_t at = lo;
  for (const DybSpan& sp : S.upd_spans) {
    if (sp.off + sp.n <= at) continue;
    if (sp.off >= hi) break;
    DYB_REQUIRE(sp.off >= lo && sp.off + sp.n <= hi, DYB_ERR_UNSUPPORTED);     // a tensor never straddles a range boundary
    RUN(add(at, sp.off));
    at = sp.off + sp.n;
  }
  RUN(add(at, hi));
  return flush();
}
// Adam over [lo, hi) minus the spans the outer level's weight gradients updated themselves (S.upd_spans, sorted)
static int adam_range(Stepper& S, const float* g2, const float* g3, const float* ss, const float* bc, size_t lo, size_t hi, hipStream_t s,
                      bool ema = false) {
  bool any = false;
  for (const DybSpan& sp : S.upd_spans)
    if (sp.off < hi && sp.off + sp.n > lo) { any = true; break; }
  if (!any)      // (ema: the teacher's EMA of the range in the same pass - "fuse_ema", default term set)
    return dyb_adam_step_rep3_ema(S.theta + lo, S.grads + lo, g2 ? g2 + lo : nullptr, g3 ? g3 + lo : nullptr, S.adam_m + lo, S.adam_v + lo,
                                  (float)S.beta1, (float)S.beta2, ss, bc, (float)S.eps, hi - lo, ema ? S.teacher + lo : nullptr, (float)S.alpha, s);
  DYB_REQUIRE(!ema, DYB_ERR_UNSUPPORTED);
  DYB_REQUIRE(!g2 && !g3 && lo % 4 == 0 && hi % 4 == 0, DYB_ERR_UNSUPPORTED);
  DybFwSegs t{};
  auto flush = [&]() -> int {
    if (t.n == 0) return DYB_OK;
    const int rc = dyb_adam_step_segs(S.theta, S.grads, S.adam_m, S.adam_v, (float)S.beta1, (float)S.beta2, ss, bc, (float)S.eps, t, s);
    t.n = 0;
    return rc;
  };
  auto add = [&](size_t a, size_t b) -> int {
    if (a >= b) return DYB_OK;
    DYB_REQUIRE(a % 4 == 0 && b % 4 == 0 && (b - a) / 4 < 0xffffffffull && a / 4 < 0xffffffffull, DYB_ERR_UNSUPPORTED);
    if (t.n == DYB_FW_MAX_SEGS) RUN(flush());
    t.start4[t.n] = (unsigned)(a / 4); t.count4[t.n] = (unsigned)((b - a) / 4); ++t.n;
    return DYB_OK;
  };
  size_t at = lo;
  for (const DybSpan& sp : S.upd_spans) {
    if (sp.off + sp.n <= at) continue;
    if (sp.off >= hi) break;
    DYB_REQUIRE(sp.off >= lo && sp.off + sp.n <= hi, DYB_ERR_UNSUPPORTED);
    RUN(add(at, sp.off));
    at = sp.off + sp.n;
  }
  RUN(add(at, hi));
  return flush();
}
// this step's Adam bias corrections for every replica of the current scope (step counts advance HERE), kept for weight_update(adam) and
// written into each replica's adam_sc on `st` for the weight-gradient epilogues that apply Adam themselves
static int adam_prepare(Stepper& S, hipStream_t st) {
  const DybRep& R = dyb_rep_current();
  for (int r = 0; r < DYB_MAX_REPLICAS; ++r) { S.pre_ss[r] = 0.f; S.pre_bc[r] = 1.f; }
  for (int i = 0; i < R.n; ++i) {
    const int r = dyb_rep_phys(R, i);
    const double t = (double)(++S.adam_t_rep[r]);
    S.pre_ss[r] = (float)(S.lr / (1.0 - pow(S.beta1, t)));
    S.pre_bc[r] = (float)sqrt(1.0 - pow(S.beta2, t));
    if (S.adam_t_rep[r] > S.adam_t) S.adam_t = S.adam_t_rep[r];
  }
  S.adam_prepared = true;
  return dyb_adam_write_scalars(S.adam_sc, S.pre_ss, S.pre_bc, st);
}
// Adam on every replica of the current launch scope, each with its own step count (bias corrections per physical replica)
// the deferred last range of a ranged weight update (see weight_update): issued from inside the consuming forward at layer3
static int late_update(void* user) {
  Stepper& S = *reinterpret_cast<Stepper*>(user);
  Stepper::LateUpd& L = S.late;
  DybRepScope scope(L.scope);
  const size_t lo = S.grp_bounds[1], n = S.n_params - lo;
  HIPOK(hipStreamWaitEvent(L.aux, S.gates.mid, 0));
  DybStreamCapScope cap(S.upd_blocks > 0 ? (S.upd_blocks / L.scope.n > 0 ? S.upd_blocks / L.scope.n : 1) : 0);
  const float *g2 = L.g2 ? L.g2 + lo : nullptr, *g3 = L.g3 ? L.g3 + lo : nullptr;
  if (L.adam) {
    const bool fe = L.ema && S.fuse_ema;
    RUN(adam_range(S, g2 ? L.g2 : nullptr, g3 ? L.g3 : nullptr, L.ss, L.bc, lo, lo + n, L.aux, fe));
    if (L.ema && !fe) RUN(dyb_ema_update(S.teacher + lo, S.theta + lo, (float)S.alpha, n, L.aux));
  } else {
    RUN(fastweight_range(S, L.p, L.out, g2 ? L.g2 : nullptr, g3 ? L.g3 : nullptr, lo, lo + n, L.aux));
  }
  HIPOK(hipEventRecord(S.gates.ev[1], L.aux));
  S.gates.late = nullptr;
  return DYB_OK;
}
// One weight update over the whole arena - the fast-weight step out = p - fastlr * grads (adam = false) or Adam in place on theta -
// as one launch on `st`, or (upd_overlap, replica groups with an auxiliary stream) by arena ranges: [0, layer3) on `st`, [layer3,
// layer4) and [layer4, end) on `aux` behind everything `st` has issued, each followed by its gate event; the next pass_forward waits
// for them where it first reads those weights.
static int weight_update(Stepper& S, bool adam, const float* p, float* out, hipStream_t st, hipStream_t aux, bool ema = false) {
  // a ranged update nobody consumed (no chain forward followed it) must not be overtaken: its tail range would stay un-issued and
  // its events would satisfy later waits with stale weights - settle it first
  if (S.gates_pending) RUN(settle_update(S, st));
  DYB_REQUIRE(!S.gates.late, DYB_ERR_LAUNCH);
  const DybRep& R = dyb_rep_current();
  const float *g2 = S.lvl_g2, *g3 = S.lvl_g3;        // further gradient arenas of the level just differentiated (full term set), consumed here
  S.lvl_g2 = S.lvl_g3 = nullptr;
  // the spans this level's weight gradients turned into fast weights themselves ("fuse_fast"): taken over, sorted, for this update only
  S.upd_spans.clear();
  S.upd_spans.swap(S.fused_spans);
  std::sort(S.upd_spans.begin(), S.upd_spans.end(), [](const DybSpan& a, const DybSpan& b) { return a.off < b.off; });
  S.fused_spans.clear();
  float ss[DYB_MAX_REPLICAS], bc[DYB_MAX_REPLICAS];
  if (adam && S.adam_prepared) {                       // fixed before the outer backward (adam_prepare): the epilogues used the same numbers
    for (int r = 0; r < DYB_MAX_REPLICAS; ++r) { ss[r] = S.pre_ss[r]; bc[r] = S.pre_bc[r]; }
    S.adam_prepared = false;
  } else if (adam) {
    for (int r = 0; r < DYB_MAX_REPLICAS; ++r) { ss[r] = 0.f; bc[r] = 1.f; }
    for (int i = 0; i < R.n; ++i) {
      const int r = dyb_rep_phys(R, i);
      const double t = (double)(++S.adam_t_rep[r]);
      ss[r] = (float)(S.lr / (1.0 - pow(S.beta1, t)));
      bc[r] = (float)sqrt(1.0 - pow(S.beta2, t));
      if (S.adam_t_rep[r] > S.adam_t) S.adam_t = S.adam_t_rep[r];
    }
  }
  auto range = [&](size_t lo, size_t hi, hipStream_t s) -> int {
    if (adam) {
      // update_teacher (base_adaptor.py:193-201) of the same range: inside the Adam pass ("fuse_ema"), or right behind it
      const bool fe = ema && S.fuse_ema;
      RUN(adam_range(S, g2, g3, ss, bc, lo, hi, s, fe));
      if (ema && !fe) RUN(dyb_ema_update(S.teacher + lo, S.theta + lo, (float)S.alpha, hi - lo, s));
      return DYB_OK;
    }
    return fastweight_range(S, p, out, g2, g3, lo, hi, s);
  };
  const bool ranged = S.upd_overlap && S.nrep > 1 && aux && aux != st && S.grp_bounds[0] > 0 && S.grp_bounds[1] > S.grp_bounds[0] &&
                      S.grp_bounds[1] < S.n_params;
  if (!ranged) return range(0, S.n_params, st);
  RUN(range(0, S.grp_bounds[0], st));
  HIPOK(hipEventRecord(S.e_upd, st));                 // the gradients' main-stream writers (and the first range) are behind this point
  HIPOK(hipStreamWaitEvent(aux, S.e_upd, 0));
  {
    DybStreamCapScope cap(S.upd_blocks > 0 ? (S.upd_blocks / R.n > 0 ? S.upd_blocks / R.n : 1) : 0);   // upd_blocks = workgroups over ALL replicas
    RUN(range(S.grp_bounds[0], S.grp_bounds[1], aux));
    HIPOK(hipEventRecord(S.gates.ev[0], aux));
    if (S.upd_late) {
      // the last range waits until the forward that consumes it reaches layer3 (gates.late, called from inside that forward)
      S.late.S = &S; S.late.adam = adam; S.late.ema = ema; S.late.p = p; S.late.out = out; S.late.aux = aux; S.late.scope = R;
      S.late.g2 = g2; S.late.g3 = g3;
      for (int r = 0; r < DYB_MAX_REPLICAS; ++r) { S.late.ss[r] = ss[r]; S.late.bc[r] = bc[r]; }
      S.gates.late = &late_update;
      S.gates.user = &S;
    } else {
      S.gates.late = nullptr;
      RUN(range(S.grp_bounds[1], S.n_params, aux));
      HIPOK(hipEventRecord(S.gates.ev[1], aux));
    }
  }
  S.gates_pending = true;
  return DYB_OK;
}
// make `st` wait for a ranged update nobody has consumed yet (a consumer other than a forward follows)
static int settle_update(Stepper& S, hipStream_t st) {
  if (!S.gates_pending) return DYB_OK;
  if (S.gates.late) {                               // nobody's forward will call it: issue the deferred range now
    HIPOK(hipEventRecord(S.gates.mid, st));
    RUN(late_update(&S));
  }
  HIPOK(hipStreamWaitEvent(st, S.gates.ev[0], 0));
  HIPOK(hipStreamWaitEvent(st, S.gates.ev[1], 0));
  S.gates_pending = false;
  return DYB_OK;
}
static int adam_scope(Stepper& S, hipStream_t st) { return weight_update(S, true, nullptr, nullptr, st, nullptr); }
static int adapt_frame_impl(Stepper& S, const float* image, const float* kp2d, const float* gt_pose, const float* gt_betas,
                            const long long* gender, int record_slot, int loss_slot, hipStream_t st, hipStream_t aux,
                            hipStream_t side) {
  const bool metrics = S.metrics != 0;
  DYB_REQUIRE(!metrics || (gt_pose && gt_betas && gender), DYB_ERR_ARG);
  S.out_in_main = false;                                  // this frame's final inference goes to `fin` (reset at every entry point)
  if (!S.use_side || side == st) side = nullptr;
  const int K = S.inner_step;
  const size_t n = S.n_params;
  float* losslog = (S.loss_log && loss_slot >= 0 && loss_slot < S.loss_capacity) ? S.loss_log + (size_t)loss_slot * 4 * (K + 1) : nullptr;
  int slot = record_slot;

  if (metrics) {
    // the ground-truth meshes only feed metric kernels: on the side stream when there is one, behind the previous frame's
    // tail (which still reads the buffers) - both are issued below, once this frame's first level is in the main queue
    if (side) {
      S.gtjob.on = true; S.gtjob.pose = gt_pose; S.gtjob.betas = gt_betas;
    } else {
      RUN(gt_meshes(S, gt_pose, gt_betas, st));
    }
  }
  // the main stream waits for the meshes only where it first reads them (the record after inner step 0): at the top of
  // the frame the side stream is still busy with the previous frame's final forward, and waiting there stalled the main
  // chain for that whole tail (1.1 ms per frame in the kernel trace)
  bool gt_waited = !(metrics && side);
  RUN(side_wait(S));                                  // (idle long since: the previous frame posted its job at its first level)
  HostTimer t_all(S.h_total);
  ++S.h_frames;
  const float* cur = S.theta;                    // clone(): the learner starts as an alias of theta
  for (int i = 0; i <= K; ++i) {                 // i < K: lower level + adapt; i == K: upper level
    {
      HostTimer t(S.h_fwd);
      RUN(pass_forward(S, S.main, cur, image, st));
    }
    {
      HostTimer th(S.h_head);
      RUN(pass_frame_head(S, S.main, kp2d, st));
      if (losslog) {
        hipLaunchKernelGGL(copy4_kernel, dim3(1, 1, dyb_rep_current().n), dim3(64), 0, st, (const float*)S.main.losses, losslog + 4 * i,
                           dyb_rep_current());
        DYB_CHECK_LAUNCH();
      }
      // inference() after inner step i-1 = this level's forward (same weights, same image: dynaboa_benchmark.py:142)
      if (metrics && S.eval_lower && i > 0) {
        if (!gt_waited) {
          RUN(side_wait(S));                          // e_gt has been recorded
          HIPOK(hipStreamWaitEvent(st, S.e_gt, 0));
          gt_waited = true;
        }
        RUN(record_metrics(S, S.main, gender, slot++, st));
      }
    }
    // the next level's weights: the other fast buffer when the weight gradients may write them while data gradients still read `cur`
    float* nxt = (S.fuse_fast && S.theta_fast2 && cur == S.theta_fast) ? S.theta_fast2 : S.theta_fast;
    {
      HostTimer t(S.h_bwd);
      S.fused_spans.clear();
      DybWgradUpdate upd{};
      if (i < K && S.fuse_fast && S.theta_fast2) {
        upd = DybWgradUpdate{S.grads, n * sizeof(float), cur, nxt, (float)S.fastlr, &S.fused_spans};
      } else if (i == K && S.fuse_adam && cur != S.theta && !side && S.nrep > 1) {
        // the outer level is differentiated at the fast weights: theta has no reader until the final inference - Adam may be applied
        // layer by layer as the weight gradients finish (replica groups: no side stream whose tail would still read theta)
        RUN(adam_prepare(S, st));
        upd = DybWgradUpdate{S.grads, n * sizeof(float), S.theta, S.theta, 0.f, &S.fused_spans};
        upd.adam_m = S.adam_m; upd.adam_v = S.adam_v; upd.adam_sc = S.adam_sc;
        upd.b1 = (float)S.beta1; upd.b2 = (float)S.beta2; upd.eps = (float)S.eps;
      }
      DybWgradUpdateScope fuse(upd);
      RUN(pass_backward(S, S.main, cur, S.grads, st, aux));
    }
    if (i == 0 && side) {
      HostTimer t(S.h_tail);
      RUN(side_post(S, side));
    }
    if (i < K) {
      HostTimer t(S.h_update);
      RUN(weight_update(S, false, cur, nxt, st, aux));
      cur = nxt;
    }
  }
  HostTimer t_tail(S.h_tail);
  // optimizer.step(): theta is about to change in place - the previous frame's tail on the side stream reads it
  RUN(side_wait(S));
  if (S.side_pending) {
    HIPOK(hipStreamWaitEvent(st, S.e_side, 0));
    S.side_pending = false;
  }
  RUN(weight_update(S, true, nullptr, nullptr, st, side ? nullptr : aux));
  // final inference() with the updated weights (dynaboa_benchmark.py:156): in line, or - with a side stream - owed to the
  // next call / dyb_stepper_join (the caller keeps this frame's inputs alive until then)
  if (side) {
    HIPOK(hipEventRecord(S.e_theta, st));
    S.tail.on = true; S.tail.image = image; S.tail.gender = gender; S.tail.slot = slot++; S.tail.metrics = metrics;
    S.tail_stream = side;
    return DYB_OK;
  }
  RUN(pass_forward(S, S.fin, S.theta, image, st));
  RUN(settle_update(S, st));                         // (consumed by the forward above; a no-op unless that changes)
  if (metrics) RUN(record_metrics(S, S.fin, gender, slot++, st));
  return DYB_OK;
}
// ---- the full loss set -------------------------------------------------------------------------------------------------
// gradient of (frame loss [+ external terms]) or of external terms alone w.r.t. theta through pass P
static int pass_backward_ext(Stepper& S, Pass& P, const float* theta, float* grads, bool frame_loss, const float* e_rot,
                             const float* e_shape, const float* e_cam, const float* e_joints, hipStream_t st, hipStream_t aux) {
  const size_t nj = (size_t)S.B * NJ * 3;
  if (frame_loss) RUN(dyb_scale_add(nullptr, P.djoints_l, e_joints, P.djoints, nj, st));
  else RUN(dyb_scale_add(nullptr, e_joints, nullptr, P.djoints, nj, st));
  RUN(dyb_lbs_bwd(S.smpl_f[0], S.smpl_i[0], P.acts + S.off_rot, P.saved, P.djoints, nullptr, P.drot_s, P.dbetas_s, 10, S.B, P.lbs_ws,
                  S.lbs_wsb, st));
  const float* z = S.zeros;
  RUN(dyb_head_grad_combine(nullptr, frame_loss ? P.drot_l : z, P.drot_s, e_rot, frame_loss ? P.dshape_l : z, P.dbetas_s, e_shape,
                            frame_loss ? P.dcam_l : z, e_cam, P.d_rot, P.d_state, S.B, st));
  return dyb_hmr_backward_ev(S.plan, theta, P.acts, P.d_rot, P.d_state, S.n_iter, grads, P.ws, S.ws_bytes, st, aux, S.ev);
}
enum { IN_IMAGE = 0, IN_KP, IN_GT_POSE, IN_GT_BETAS, IN_GENDER, IN_HIST_IMAGE, IN_HIST_KP, IN_EX_IMG, IN_EX_KP, IN_EX_POSE,
       IN_EX_BETAS, IN_EX_POSE3D, IN_COUNT };
// copy `nk` inputs (kinds k0 .. k0+nk-1 in IN_* order) of every replica of the current scope from the callers' separate tensors
// into the replicas' staging areas, one launch: src[(kind - k0) * ld + physical replica] (NULL: nothing to copy)
static int stage_inputs(Stepper& S, const void* const* src, int ld, int k0, int nk, hipStream_t st);
struct FullCtx {
  const void* in[IN_COUNT];
  float* losslog;       // this frame's rows
  int level_row;
};
// A level that forked onto the pass streams and fails before its joins must not return with work still queued there against the
// stepper's arenas: the caller's stream is ordered behind the CURRENT tails of both pass streams on every early return, so a
// retry or a free on that stream cannot race them.  Disarmed once the regular joins (par_ev[3] / par_ev[5]) have been issued.
struct ParTailGuard {
  hipStream_t st, sB, sC;
  hipEvent_t eB, eC;
  bool armed;
  ~ParTailGuard() {
    if (!armed) return;
    if (sB != st && hipEventRecord(eB, sB) == hipSuccess) (void)hipStreamWaitEvent(st, eB, 0);
    if (sC != st && hipEventRecord(eC, sC) == hipSuccess) (void)hipStreamWaitEvent(st, eC, 0);
  }
};
// one level (reference base_adaptor.py:222-317 lower / upper_level_adaptation) at weights `cur` through pass P: loss terms,
// log row, and the gradient of the level total w.r.t. `cur` in S.grads
// have_fwd: P already holds the forward of (cur, this frame's image) - the dynamic loop hands the previous step's final inference on
static int full_level(Stepper& S, FullCtx& C, Pass& P, const float* cur, bool upper, int level_index, hipStream_t st, hipStream_t aux,
                      bool have_fwd = false) {
  const int B = S.B;
  const float* image = (const float*)C.in[IN_IMAGE];
  const float* kp = (const float*)C.in[IN_KP];
  const bool temporal = upper ? S.temporal_upper != 0 : S.temporal_lower != 0;
  const bool teacher = temporal && S.use_teacher && S.teacher;
  const bool motion = temporal && S.use_motion && C.in[IN_HIST_IMAGE] && C.in[IN_HIST_KP];
  const bool label = (upper ? S.mix_upper : S.mix_lower) != 0;
  // One sequence: the history pass and the exemplar pass go to streams of their own.  Dependencies (events par_ev[]): both wait for
  // the weights (0: recorded on the chain where the level starts); the motion term (chain) needs the history forward (1) and leaves
  // the history pass's head gradients (2) for its backward; the chain's log row reads the label term's values (4); the weight update
  // that follows the level reads all three gradient arenas (3, 5).  Their backwards run without an auxiliary stream (weight gradients
  // in line: the engine's per-plan event set serves one two-stream backward at a time - the chain's).  Retrieval by callback
  // synchronises with the level's own forward on the host: sequential as before.
  // For replica groups too (up to par_max_replicas per launch, default: all): default term set, frames/s sequential -> parallel passes at
  // 1: 65 -> 78 (89 with 8 hardware queues), 2: 93 -> 107, 3: 116 -> 129, 4: 131 -> 145, 5: 154 -> 182, 8: 205 -> 232, 16: 290 -> 316,
  // 32: 362 -> 377 (profiles/r05_sessions.txt s29 / s30) - a level's three passes fill each other's ramps and tails.
  const bool par = S.par_passes && dyb_rep_current().n <= S.par_max_replicas && !S.retrieve && !S.retrieve_rep && (motion || label);
  // a ranged weight update may still be in flight (replica groups): the chain's forward waits for each range where it first reads it,
  // the pass streams read every weight - they wait for both range events, which exist once the chain's forward has been issued
  const bool ranged_before = par && S.gates_pending && !have_fwd;
  hipStream_t sB = st, sC = st;
  if (par) {
    for (int i = 0; i < 2; ++i)
      if (!S.par_stream[i]) HIPOK(hipStreamCreateWithFlags(&S.par_stream[i], hipStreamNonBlocking));
    for (int i = 0; i < 8; ++i)
      if (!S.par_ev[i]) HIPOK(hipEventCreateWithFlags(&S.par_ev[i], hipEventDisableTiming));
    sB = S.par_stream[0]; sC = S.par_stream[1];
    HIPOK(hipEventRecord(S.par_ev[0], st));
    if (motion) HIPOK(hipStreamWaitEvent(sB, S.par_ev[0], 0));
    if (label) HIPOK(hipStreamWaitEvent(sC, S.par_ev[0], 0));
  }
  ParTailGuard tails{st, sB, sC, S.par_ev[6], S.par_ev[7], par};
  if (!have_fwd) RUN(pass_forward(S, P, cur, image, st));
  RUN(pass_frame_head(S, P, kp, st));
  const float* rot = P.acts + S.off_rot;
  const float* state = P.acts + S.off_state;
  if (ranged_before) {
    DYB_REQUIRE(!S.gates_pending && !S.gates.late, DYB_ERR_LAUNCH);            // consumed by the chain's forward above
    for (int k = 0; k < 2; ++k) {
      if (motion) HIPOK(hipStreamWaitEvent(sB, S.gates.ev[k], 0));
      if (label) HIPOK(hipStreamWaitEvent(sC, S.gates.ev[k], 0));
    }
  }
  if (par && motion) {
    RUN(pass_forward(S, S.hist, cur, (const float*)C.in[IN_HIST_IMAGE], sB, false));
    HIPOK(hipEventRecord(S.par_ev[1], sB));
  }
  auto label_pass_forward_and_term = [&](hipStream_t s, bool chain) -> int {
    RUN(pass_forward(S, S.ex, cur, (const float*)C.in[IN_EX_IMG], s, chain));
    RUN(dyb_rodrigues_fwd((const float*)C.in[IN_EX_POSE], S.ex_rot, B * 24, s));          // utils/geometry.py:9-24 on the exemplar pose
    const float* es = S.ex.acts + S.off_state;
    return dyb_aux_loss_terms(2, B, 0, (float)S.label_w, S.ex.acts + S.off_rot, es + 144, STATE_LD, es + 154, STATE_LD, S.ex.joints, nullptr,
                              nullptr, 0, nullptr, 0, nullptr, (const float*)C.in[IN_EX_KP], nullptr, S.ex_rot,
                              (const float*)C.in[IN_EX_BETAS], (const float*)C.in[IN_EX_POSE3D], S.vals_l, S.exg_rot, S.exg_shape, S.exg_cam,
                              S.exg_joints, nullptr, nullptr, s);
  };
  if (par && label) {
    DYB_REQUIRE(C.in[IN_EX_IMG] && C.in[IN_EX_KP] && C.in[IN_EX_POSE] && C.in[IN_EX_BETAS] && C.in[IN_EX_POSE3D], DYB_ERR_ARG);
    RUN(label_pass_forward_and_term(sC, false));
    HIPOK(hipEventRecord(S.par_ev[4], sC));
    RUN(pass_backward_ext(S, S.ex, cur, S.grads3, false, S.exg_rot, S.exg_shape, S.exg_cam, S.exg_joints, sC, nullptr));
    HIPOK(hipEventRecord(S.par_ev[5], sC));
  }
  bool ext = false;
  if (teacher) {
    // teacher forward (no gradient): reference base_adaptor.py:324-329
    if (S.teacher_train) {
      // (by now the chain's own forward has consumed the gates of a ranged update: the teacher's ranges are complete on `st`)
      DYB_REQUIRE(!S.gates_pending, DYB_ERR_LAUNCH);
      RUN(dyb_hmr_forward_train(S.plan, S.teacher, image, S.init_state, S.n_iter, S.teach.acts, S.teach.ws, S.ws_bytes, S.drop_seed,
                                S.drop_off + (unsigned long long)S.drop_used, (float)S.drop_p, st));
      ++S.drop_used;
      const float* tstate = S.teach.acts + S.off_state;
      RUN(dyb_lbs_fwd(S.smpl_f[0], S.smpl_i[0], tstate + 144, STATE_LD, S.teach.acts + S.off_rot, S.teach.verts, S.teach.joints,
                      S.teach.saved, S.B, st));
    } else {
      RUN(pass_forward(S, S.teach, S.teacher, image, st));
    }
    const float* ts = S.teach.acts + S.off_state;
    RUN(dyb_aux_loss_terms(0, B, 0, (float)S.teacher_w, rot, state + 144, STATE_LD, state + 154, STATE_LD, P.joints,
                           S.teach.acts + S.off_rot, ts + 144, STATE_LD, ts + 154, STATE_LD, S.teach.joints, nullptr, nullptr, nullptr,
                           nullptr, nullptr, S.vals_t, S.ext_rot, S.ext_shape, S.ext_cam, S.ext_joints, nullptr, nullptr, st));
    ext = true;
  }
  if (motion) {
    // the history frame through the SAME weights, with gradient (base_adaptor.py:380-386)
    if (par) HIPOK(hipStreamWaitEvent(st, S.par_ev[1], 0));
    else RUN(pass_forward(S, S.hist, cur, (const float*)C.in[IN_HIST_IMAGE], st));
    const float* hs = S.hist.acts + S.off_state;
    RUN(dyb_aux_loss_terms(1, B, ext ? 1 : 0, (float)S.motion_w, rot, state + 144, STATE_LD, state + 154, STATE_LD, P.joints, nullptr,
                           nullptr, 0, hs + 154, STATE_LD, S.hist.joints, kp, (const float*)C.in[IN_HIST_KP], nullptr, nullptr, nullptr,
                           S.vals_m, S.ext_rot, S.ext_shape, S.ext_cam, S.ext_joints, S.hg_cam, S.hg_joints, st));
    ext = true;
    if (par) {
      HIPOK(hipEventRecord(S.par_ev[2], st));
      HIPOK(hipStreamWaitEvent(sB, S.par_ev[2], 0));
      RUN(pass_backward_ext(S, S.hist, cur, S.grads2, false, nullptr, nullptr, S.hg_cam, S.hg_joints, sB, nullptr));
      HIPOK(hipEventRecord(S.par_ev[3], sB));
    }
  }
  if (label && !par) {
    // retrieval (base_adaptor.py:82-96) happens on the host: hand it the pooled feature of this level's forward
    if (S.nrep > 1 && S.retrieve_rep) {
      DYB_REQUIRE(S.feat5_out, DYB_ERR_ARG);
      const DybRep& R = dyb_rep_current();
      RUN(dyb_scale_add(nullptr, P.acts + S.fca.off[5], nullptr, S.feat5_out, (size_t)B * 2048, st));     // every replica's feature, one launch
      const void* exin[5 * DYB_MAX_REPLICAS];
      for (int k = 0; k < 5 * DYB_MAX_REPLICAS; ++k) exin[k] = nullptr;
      for (int i = 0; i < R.n; ++i) {
        const int r = dyb_rep_phys(R, i);
        const void* one[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        if (S.retrieve_rep(S.retrieve_user, level_index, r, one) != 0) return DYB_ERR_ARG;
        for (int k = 0; k < 5; ++k) {
          DYB_REQUIRE(one[k], DYB_ERR_ARG);
          exin[k * DYB_MAX_REPLICAS + r] = one[k];
        }
      }
      RUN(stage_inputs(S, exin, DYB_MAX_REPLICAS, IN_EX_IMG, 5, st));
      for (int k = 0; k < 5; ++k) C.in[IN_EX_IMG + k] = S.in_stage[IN_EX_IMG + k];
    } else if (S.retrieve) {
      DYB_REQUIRE(S.feat5_out && S.nrep == 1, DYB_ERR_ARG);
      RUN(dyb_scale_add(nullptr, P.acts + S.fca.off[5], nullptr, S.feat5_out, (size_t)0 + (size_t)B * 2048, st));   // B == 1: contiguous
      const void* exin[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
      if (S.retrieve(S.retrieve_user, level_index, exin) != 0) return DYB_ERR_ARG;
      for (int k = 0; k < 5; ++k) C.in[IN_EX_IMG + k] = exin[k];
    }
    DYB_REQUIRE(C.in[IN_EX_IMG] && C.in[IN_EX_KP] && C.in[IN_EX_POSE] && C.in[IN_EX_BETAS] && C.in[IN_EX_POSE3D], DYB_ERR_ARG);
    RUN(label_pass_forward_and_term(st, true));
  }
  if (C.losslog) {
    if (par && label) HIPOK(hipStreamWaitEvent(st, S.par_ev[4], 0));          // the row reads the label term's values
    hipLaunchKernelGGL(level_log_kernel, dim3(1, 1, dyb_rep_current().n), dim3(64), 0, st, (const float*)P.losses,
                       teacher ? (const float*)S.vals_t : nullptr, motion ? (const float*)S.vals_m : nullptr,
                       label ? (const float*)S.vals_l : nullptr, (float)S.teacher_w, (float)S.motion_w, (float)S.label_w,
                       C.losslog + 16 * C.level_row, dyb_rep_current());
    DYB_CHECK_LAUNCH();
    ++C.level_row;
  }
  // gradient of the level total: image pass (+ its external terms), history pass, exemplar pass
  RUN(pass_backward_ext(S, P, cur, S.grads, true, ext ? S.ext_rot : nullptr, ext ? S.ext_shape : nullptr, ext ? S.ext_cam : nullptr,
                        ext ? S.ext_joints : nullptr, st, aux));
  // (the level's gradient = (image pass + history pass) + exemplar pass: the passes keep their own arenas and the weight update that
  // follows every level adds them while it reads them - no accumulation passes)
  S.lvl_g2 = S.lvl_g3 = nullptr;
  if (motion) {
    if (par) HIPOK(hipStreamWaitEvent(st, S.par_ev[3], 0));
    else RUN(pass_backward_ext(S, S.hist, cur, S.grads2, false, nullptr, nullptr, S.hg_cam, S.hg_joints, st, aux));
    S.lvl_g2 = S.grads2;
  }
  if (label) {
    if (par) HIPOK(hipStreamWaitEvent(st, S.par_ev[5], 0));
    else RUN(pass_backward_ext(S, S.ex, cur, S.grads3, false, S.exg_rot, S.exg_shape, S.exg_cam, S.exg_joints, st, aux));
    if (S.lvl_g2) S.lvl_g3 = S.grads3;
    else S.lvl_g2 = S.grads3;
  }
  tails.armed = false;                                    // both pass streams have been joined above
  return DYB_OK;
}
static int adam_and_teacher(Stepper& S, hipStream_t st, hipStream_t aux) {
  return weight_update(S, true, nullptr, nullptr, st, aux, S.use_teacher && S.teacher);      // Adam, then the teacher's EMA (base_adaptor.py:193-201)
}
// features of two forwards -> 15 cosines per replica of the current scope (device log row + host-visible copy); cos12[r] (indexed
// by PHYSICAL replica) = cos[12] read from the host copy
static int gate_cosine(Stepper& S, const float* actsA, const float* actsB, float* log_row, float* cos12, hipStream_t st) {
  DYB_REQUIRE(S.gate_host, DYB_ERR_ARG);
  const DybRep& R = dyb_rep_current();
  S.gate_seq += 1.f;
  hipLaunchKernelGGL(feat_cos_kernel, dim3(15, 1, R.n), dim3(1024), 0, st, actsA, actsB, S.fca, 1e-12f, log_row, S.gate_host, S.gate_seq, R);
  DYB_CHECK_LAUNCH();
  // the one host wait of the dynamic loop (the reference's `.item()`, dynaboa_benchmark.py:165): a poll of pinned memory the
  // kernel writes, no stream synchronise / device-to-host copy call
  for (int i = 0; i < R.n; ++i) {
    const int r = dyb_rep_phys(R, i);
    long spins = 0;
    while (S.gate_host[16 * r + 15] != S.gate_seq) {
      if (++spins > 2000000000L) return DYB_ERR_LAUNCH;
      __builtin_ia32_pause();
    }
    cos12[r] = S.gate_host[16 * r + 12];
  }
  return DYB_OK;
}
static int stage_inputs(Stepper& S, const void* const* src, int ld, int k0, int nk, hipStream_t st) {
  DYB_REQUIRE(nk >= 1 && nk <= 5 && S.nrep > 1, DYB_ERR_ARG);
  const DybRep& R = dyb_rep_current();
  const size_t B = (size_t)S.B, img = B * 3 * (size_t)S.H * S.W;
  const size_t cnt[IN_COUNT] = {img, B * NJ * 3, B * 72, B * 10, B * 2, img, B * NJ * 3, img, B * NJ * 3, B * 72, B * 10, B * 24 * 4};
  GatherArgs g{};
  for (int k = 0; k < nk; ++k) {
    DYB_REQUIRE(S.in_stage[k0 + k], DYB_ERR_ARG);
    g.dst[k] = S.in_stage[k0 + k];
    g.n[k] = (unsigned)cnt[k0 + k];
    for (int i = 0; i < R.n; ++i) g.src[k][dyb_rep_phys(R, i)] = reinterpret_cast<const float*>(src[(size_t)k * ld + dyb_rep_phys(R, i)]);
  }
  hipLaunchKernelGGL(gather_inputs_kernel, dim3(64, nk, R.n), dim3(256), 0, st, g, R);
  DYB_CHECK_LAUNCH();
  return DYB_OK;
}
// The launch scope of a set of physical replicas: every per-replica arena of the stepper, map = the set.  DYB_ERR_UNSUPPORTED when
// the stepper's arenas do not fit the scope's DYB_MAX_ARENAS slots: with the full term set and replicas the separate records /
// loss_log / gate_log / feat5_out pointers are one too many - pass them as sub-buffers of ONE per-replica block through
// "logs_base" / "logs_bytes" (what the Python driver does); an arena silently left out would alias every replica onto replica 0's.
static int make_scope(const Stepper& S, const int* idx, int n, DybRep* out) {
  DybRep R{};
  dyb_rep_identity(R);
  dyb_rep_set_map(R, idx, n);
  bool overflow = false;
  auto arena = [&](const void* lo, size_t bytes) {
    if (!lo || !bytes) return;
    if (R.narenas >= DYB_MAX_ARENAS) { overflow = true; return; }
    R.lo[R.narenas] = reinterpret_cast<const char*>(lo);
    R.span[R.narenas] = bytes;
    R.stride[R.narenas] = bytes;
    ++R.narenas;
  };
  const int rows = S.inner_step + 1 + (S.full && S.dynamic ? S.optim_steps : 0);
  arena(S.wsp, S.blob);
  arena(S.theta, S.n_params * sizeof(float));
  arena(S.adam_m, S.n_params * sizeof(float));
  arena(S.adam_v, S.n_params * sizeof(float));
  if (S.full) arena(S.teacher, S.n_params * sizeof(float));
  if (S.logs_base && S.logs_bytes) {
    arena(S.logs_base, S.logs_bytes);              // records, loss_log, gate_log, feat5_out: sub-buffers of one per-replica block
  } else {
    arena(S.records, (size_t)S.record_capacity * a64((size_t)S.B * 85 + 1) * sizeof(float));
    arena(S.loss_log, (size_t)S.loss_capacity * (S.full ? 16 : 4) * rows * sizeof(float));
    if (S.full) {
      arena(S.gate_log, (size_t)S.loss_capacity * (1 + S.optim_steps) * 16 * sizeof(float));
      arena(S.feat5_out, (size_t)S.B * 2048 * sizeof(float));
    }
  }
  DYB_REQUIRE(!overflow, DYB_ERR_UNSUPPORTED);
  *out = R;
  return DYB_OK;
}
// the replicas the next frame step covers (ascending physical indices; n = 0: all).  Sequences of different lengths: a replica
// whose stream has ended simply leaves the set - its weights, Adam state and records stay as they are.
extern "C" int dyb_stepper_set_active(void* stepper, const int* idx, int n) {
  Stepper* S = reinterpret_cast<Stepper*>(stepper);
  DYB_REQUIRE(S && n >= 0 && n <= S->nrep && (n == 0 || idx), DYB_ERR_ARG);
  for (int i = 0; i < n; ++i) DYB_REQUIRE(idx[i] >= 0 && idx[i] < S->nrep && (i == 0 || idx[i] > idx[i - 1]), DYB_ERR_ARG);
  S->nactive = n;
  for (int i = 0; i < n; ++i) S->active[i] = idx[i];
  return DYB_OK;
}
static int active_set(const Stepper& S, int* idx) {
  if (S.nactive > 0) {
    for (int i = 0; i < S.nactive; ++i) idx[i] = S.active[i];
    return S.nactive;
  }
  for (int i = 0; i < S.nrep; ++i) idx[i] = i;
  return S.nrep;
}

// Adaptor.adaptation with the reference's full term set (dynaboa_benchmark.py:126-193) for every replica of the current launch
// scope (one replica without a scope).  C.in: the frame's inputs - the callers' tensors (one replica) or the staging areas.
// extra[r] (physical replica) receives the number of dynamic-loop iterations replica r took: the gate is evaluated per replica,
// and a replica whose feature 12 has stopped moving LEAVES the launch set of the remaining iterations (the others are not held
// back by it, and it does not take Adam steps the reference would not have taken).
static int adapt_full_impl(Stepper& S, FullCtx& C, int record_slot, int loss_slot, int* extra, hipStream_t st, hipStream_t aux) {
  const bool metrics = S.metrics != 0;
  S.out_in_main = false;                                  // (set again below only when a shared-forward step of the dynamic loop ends in `main`)
  DYB_REQUIRE(C.in[IN_IMAGE] && C.in[IN_KP], DYB_ERR_ARG);
  DYB_REQUIRE(!metrics || (C.in[IN_GT_POSE] && C.in[IN_GT_BETAS] && C.in[IN_GENDER]), DYB_ERR_ARG);
  const long long* gender = (const long long*)C.in[IN_GENDER];
  const int K = S.inner_step;
  const int rows = K + 1 + (S.dynamic ? S.optim_steps : 0);
  C.losslog = (S.loss_log && loss_slot >= 0 && loss_slot < S.loss_capacity) ? S.loss_log + (size_t)loss_slot * 16 * rows : nullptr;
  C.level_row = 0;
  int slot = record_slot;
  const DybRep outer = dyb_rep_current();                 // (copy: nested scopes below replace the thread's current one)
  for (int i = 0; i < outer.n; ++i) extra[dyb_rep_phys(outer, i)] = 0;
  if (metrics) RUN(gt_meshes(S, (const float*)C.in[IN_GT_POSE], (const float*)C.in[IN_GT_BETAS], st));
  const float* cur = S.theta;
  for (int i = 0; i <= K; ++i) {
    Pass& P = (i == 0 && S.dynamic) ? S.lvl0 : S.main;
    RUN(full_level(S, C, P, cur, i == K, i, st, aux));
    // (the level's gradient is complete; the metric record of the previous inner step reads this level's forward)
    if (metrics && S.eval_lower && i > 0) RUN(record_metrics(S, P, gender, slot++, st));
    if (i < K) {
      RUN(weight_update(S, false, cur, S.theta_fast, st, aux));
      cur = S.theta_fast;
    }
  }
  RUN(adam_and_teacher(S, st, aux));
  const float* image = (const float*)C.in[IN_IMAGE];
  RUN(pass_forward(S, S.fin, S.theta, image, st));
  if (metrics) RUN(record_metrics(S, S.fin, gender, slot++, st));
  if (S.dynamic) {
    // dynaboa_benchmark.py:161-192: compare features of the un-adapted and the adapted forward; while feature 12 still
    // moves, repeat the upper level on the model itself (at most optim_steps times)
    float* glog = (S.gate_log && loss_slot >= 0 && loss_slot < S.loss_capacity) ? S.gate_log + (size_t)loss_slot * (1 + S.optim_steps) * 16 : nullptr;
    DYB_REQUIRE(glog, DYB_ERR_ARG);
    float cos12[DYB_MAX_REPLICAS];
    RUN(gate_cosine(S, S.lvl0.acts, S.fin.acts, glog, cos12, st));     // (inner_step 0: the upper level's own forward is lvl0)
    int cont[DYB_MAX_REPLICAS], ncont = 0;
    for (int i = 0; i < outer.n; ++i)
      if (1.f - cos12[dyb_rep_phys(outer, i)] > (float)S.cos_thr) cont[ncont++] = dyb_rep_phys(outer, i);
    int step = 0;
    Pass* post = &S.fin;                                   // where the latest final inference lives
    S.out_in_main = false;
    while (ncont > 0) {
      ++step;
      if (step > S.optim_steps) {
        for (int i = 0; i < ncont; ++i) extra[cont[i]] = S.optim_steps + 1;
        break;
      }
      {
        // the replicas still adapting get a scope of their own (one replica without replica machinery: the scope it came with)
        DybRep sub = outer;
        dyb_rep_set_map(sub, cont, ncont);
        DybRepScope scope(sub);
        C.level_row = K + step;
        // The upper level of this step starts with a forward at (theta, this image) - exactly the final inference the previous step
        // ended with (same weights, same image, deterministic kernels): it is not repeated.  `up` = that forward's products with the
        // chain's gradient buffers; the new final inference goes to the other activation arena (the gate compares the two), and the
        // two arenas swap roles every step ("share_dyn_fwd"; 0: the literal two forwards per step).
        if (S.share_dyn_fwd) {
          Pass up = S.main;
          up.acts = post->acts; up.verts = post->verts; up.joints = post->joints; up.saved = post->saved; up.pred17 = post->pred17;
          Pass* nxt = (post == &S.fin) ? &S.main : &S.fin;
          RUN(full_level(S, C, up, S.theta, true, K + step, st, aux, true));
          RUN(adam_and_teacher(S, st, aux));
          RUN(pass_forward(S, *nxt, S.theta, image, st));
          RUN(gate_cosine(S, up.acts, nxt->acts, glog + 16 * step, cos12, st));
          if (metrics) RUN(record_metrics(S, *nxt, gender, slot, st));
          post = nxt;
          S.out_in_main = (post == &S.main);
        } else {
          RUN(full_level(S, C, S.main, S.theta, true, K + step, st, aux));
          RUN(adam_and_teacher(S, st, aux));
          RUN(pass_forward(S, S.fin, S.theta, image, st));
          RUN(gate_cosine(S, S.main.acts, S.fin.acts, glog + 16 * step, cos12, st));
          if (metrics) RUN(record_metrics(S, S.fin, gender, slot, st));
        }
      }
      ++slot;
      int m = 0;
      for (int i = 0; i < ncont; ++i) {
        extra[cont[i]] = step;
        if (1.f - cos12[cont[i]] > (float)S.cos_thr) cont[m++] = cont[i];
      }
      ncont = m;
    }
  }
  return DYB_OK;
}
// One sequence: inputs = HOST array of IN_COUNT device pointers (image, kp2d, gt_pose, gt_betas, gender, hist_image, hist_kp,
// ex_img, ex_kp, ex_pose, ex_betas, ex_pose3d; the history pair NULL while there is no frame `interval` steps back, the exemplar
// five NULL when a retrieval callback is set).  *extra_steps receives the number of dynamic-loop iterations taken.
extern "C" int dyb_stepper_adapt_frame_full(void* stepper, const void* const* inputs, int record_slot, int loss_slot, int* extra_steps,
                                            hipStream_t st, hipStream_t aux) {
  Stepper* Sp = reinterpret_cast<Stepper*>(stepper);
  DYB_REQUIRE(Sp && inputs && extra_steps, DYB_ERR_ARG);
  Stepper& S = *Sp;
  RUN(check_ready(S));
  DYB_REQUIRE(S.full && S.nrep == 1 && S.B <= 16, DYB_ERR_UNSUPPORTED);
  FullCtx C{};
  for (int i = 0; i < IN_COUNT; ++i) C.in[i] = inputs[i];
  int extra[DYB_MAX_REPLICAS];
  RUN(adapt_full_impl(S, C, record_slot, loss_slot, extra, st, aux));
  *extra_steps = extra[0];
  return DYB_OK;
}
// The same for the stepper's replicas (the active set, dyb_stepper_set_active): every launch of the chain covers all of them,
// teacher / history / exemplar passes included.  inputs: HOST array of IN_COUNT x replicas device pointers, kind-major -
// inputs[kind * replicas + r], r the PHYSICAL replica (entries of inactive replicas are ignored); the history pair must be
// present for all active replicas or for none; the exemplar five may be NULL when a per-replica retrieval callback
// ("retrieve_rep_fn") is set.  theta / adam_m / adam_v / teacher are [replicas][param floats], records / loss_log / gate_log
// [replicas][...], gate_host 16 floats per replica, feat5_out [replicas][B][2048].  extra_steps: `replicas` ints.
extern "C" int dyb_stepper_adapt_frames_full(void* stepper, const void* const* inputs, int record_slot, int loss_slot, int* extra_steps,
                                             hipStream_t st, hipStream_t aux) {
  Stepper* Sp = reinterpret_cast<Stepper*>(stepper);
  DYB_REQUIRE(Sp && inputs && extra_steps, DYB_ERR_ARG);
  Stepper& S = *Sp;
  RUN(check_ready(S));
  DYB_REQUIRE(S.full && S.B <= 16, DYB_ERR_UNSUPPORTED);
  if (S.nrep == 1) return dyb_stepper_adapt_frame_full(stepper, inputs, record_slot, loss_slot, extra_steps, st, aux);
  int act[DYB_MAX_REPLICAS];
  const int na = active_set(S, act);
  const int n = S.nrep;
  bool have_hist = false, have_ex = false, have_gt = false;
  for (int i = 0; i < na; ++i) {
    const int r = act[i];
    DYB_REQUIRE(inputs[IN_IMAGE * n + r] && inputs[IN_KP * n + r], DYB_ERR_ARG);
    const bool h = inputs[IN_HIST_IMAGE * n + r] && inputs[IN_HIST_KP * n + r];
    const bool e = inputs[IN_EX_IMG * n + r] && inputs[IN_EX_KP * n + r] && inputs[IN_EX_POSE * n + r] && inputs[IN_EX_BETAS * n + r] &&
                   inputs[IN_EX_POSE3D * n + r];
    const bool gt = inputs[IN_GT_POSE * n + r] && inputs[IN_GT_BETAS * n + r] && inputs[IN_GENDER * n + r];
    if (i == 0) { have_hist = h; have_ex = e; have_gt = gt; }
    DYB_REQUIRE(h == have_hist && e == have_ex && gt == have_gt, DYB_ERR_UNSUPPORTED);     // lockstep: the same terms for every replica
  }
  DybRep R{};
  RUN(make_scope(S, act, na, &R));
  DybRepScope scope(R);
  RUN(stage_inputs(S, inputs, n, IN_IMAGE, 5, st));
  if (have_hist) RUN(stage_inputs(S, inputs + (size_t)IN_HIST_IMAGE * n, n, IN_HIST_IMAGE, 2, st));
  if (have_ex) RUN(stage_inputs(S, inputs + (size_t)IN_EX_IMG * n, n, IN_EX_IMG, 5, st));
  FullCtx C{};
  for (int k = 0; k < IN_COUNT; ++k) C.in[k] = nullptr;
  C.in[IN_IMAGE] = S.in_stage[IN_IMAGE]; C.in[IN_KP] = S.in_stage[IN_KP];
  if (have_gt) { C.in[IN_GT_POSE] = S.in_stage[IN_GT_POSE]; C.in[IN_GT_BETAS] = S.in_stage[IN_GT_BETAS]; C.in[IN_GENDER] = S.in_stage[IN_GENDER]; }
  if (have_hist) { C.in[IN_HIST_IMAGE] = S.in_stage[IN_HIST_IMAGE]; C.in[IN_HIST_KP] = S.in_stage[IN_HIST_KP]; }
  if (have_ex) for (int k = IN_EX_IMG; k <= IN_EX_POSE3D; ++k) C.in[k] = S.in_stage[k];
  for (int r = 0; r < n; ++r) extra_steps[r] = 0;
  return adapt_full_impl(S, C, record_slot, loss_slot, extra_steps, st, aux);
}

extern "C" int dyb_stepper_adapt_frame(void* stepper, const float* image, const float* kp2d, const float* gt_pose,
                                       const float* gt_betas, const long long* gender, int record_slot, int loss_slot,
                                       hipStream_t st, hipStream_t aux, hipStream_t side) {
  Stepper* Sp = reinterpret_cast<Stepper*>(stepper);
  DYB_REQUIRE(Sp && image && kp2d, DYB_ERR_ARG);
  RUN(check_ready(*Sp));
  DYB_REQUIRE(Sp->nrep == 1, DYB_ERR_ARG);
  return adapt_frame_impl(*Sp, image, kp2d, gt_pose, gt_betas, gender, record_slot, loss_slot, st, aux, side);
}
// The same frame step for `replicas` independent sequences at once: every launch of the chain covers all of them (replica
// = a grid dimension, dyb_common.h), each with its own weights / Adam moments / workspace / records - theta, adam_m, adam_v
// are [replicas][param floats], records [replicas][record_capacity][record_floats], loss_log [replicas][loss_capacity]
// [loss_floats], the workspace `replicas` blobs.  inputs: HOST array of 5 x replicas device pointers, kind-major:
// image[r], kp2d[r], gt_pose[r], gt_betas[r], gender[r] (the last three may be NULL with metrics = 0).
extern "C" int dyb_stepper_adapt_frames(void* stepper, const void* const* inputs, int record_slot, int loss_slot, hipStream_t st,
                                        hipStream_t aux, hipStream_t side) {
  Stepper* Sp = reinterpret_cast<Stepper*>(stepper);
  DYB_REQUIRE(Sp && inputs, DYB_ERR_ARG);
  Stepper& S = *Sp;
  RUN(check_ready(S));
  const int n = S.nrep;
  if (n == 1)
    return adapt_frame_impl(S, (const float*)inputs[0], (const float*)inputs[1], (const float*)inputs[2], (const float*)inputs[3],
                            (const long long*)inputs[4], record_slot, loss_slot, st, aux, side);
  int act[DYB_MAX_REPLICAS];
  const int na = active_set(S, act);
  // replicas + side stream: the owed final inference of frame f would read staging buffers the next call has already refilled,
  // under the next frame's launch scope - replica groups run their tail in line
  DYB_REQUIRE(!(S.use_side && side && side != st), DYB_ERR_UNSUPPORTED);
  DybRep R{};
  RUN(make_scope(S, act, na, &R));
  DybRepScope scope(R);
  for (int i = 0; i < na; ++i) DYB_REQUIRE(inputs[0 * n + act[i]] && inputs[1 * n + act[i]], DYB_ERR_ARG);
  // previous frame's tail on the side stream still reads the staged inputs
  hipStream_t gst = st;
  RUN(side_wait(S));
  if (S.side_pending) HIPOK(hipStreamWaitEvent(gst, S.e_side, 0));
  RUN(stage_inputs(S, inputs, n, IN_IMAGE, 5, gst));
  const bool have_gt = inputs[2 * n + act[0]] && inputs[3 * n + act[0]] && inputs[4 * n + act[0]];
  if (S.use_side && side && side != st) {
    // the ground-truth meshes are issued on the side stream: it has to see the staged inputs
    HIPOK(hipEventRecord(S.e_gt, st));
    HIPOK(hipStreamWaitEvent(side, S.e_gt, 0));
  }
  return adapt_frame_impl(S, S.in_stage[0], S.in_stage[1], have_gt ? S.in_stage[2] : nullptr, have_gt ? S.in_stage[3] : nullptr,
                          have_gt ? reinterpret_cast<const long long*>(S.in_stage[4]) : nullptr, record_slot, loss_slot, st, aux, side);
}
// make `st` wait for everything the stepper has in flight on its side stream (before reading records / the final prediction)
extern "C" int dyb_stepper_join(void* stepper, hipStream_t st) {
  Stepper* S = reinterpret_cast<Stepper*>(stepper);
  DYB_REQUIRE(S, DYB_ERR_ARG);
  RUN(side_wait(*S));
  if ((S->tail.on || S->gtjob.on) && S->tail_stream) RUN(issue_side_work(*S, S->tail_stream));      // the last frame's final inference
  if (S->side_pending) {
    HIPOK(hipStreamWaitEvent(st, S->e_side, 0));
    S->side_pending = false;
  }
  return DYB_OK;
}
// device locations of the last final-inference outputs: which = 0 rotmat [B][24][9], 1 state [B][160] (shape 144.., cam 154..),
// 2 vertices [B][6890][3], 3 joints [B][49][3]
extern "C" const float* dyb_stepper_output(const void* stepper, int which) {
  const Stepper* S = reinterpret_cast<const Stepper*>(stepper);
  if (!S || !S->bound) return nullptr;
  // (the dynamic loop's steps alternate between the two activation arenas, "share_dyn_fwd": the last launch's final inference - for a
  // replica that left the loop a step earlier than the last one, read its prediction from the records)
  const Pass& F = (S->out_in_main ? S->main : S->fin);
  switch (which) {
    case 0: return F.acts + S->off_rot;
    case 1: return F.acts + S->off_state;
    case 2: return F.verts;
    case 3: return F.joints;
    default: return nullptr;
  }
}
