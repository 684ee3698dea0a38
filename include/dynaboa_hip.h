/* libdynaboa_hip.so - C ABI of the MI355X (gfx950) kernels behind DynaBOA's per-frame adaptation
 * hot path.
 *
 * The reference (syguan96/DynaBOA) has no FFI: its boundary for this path is the Python object
 * surface (hmr()/HMR.forward, SMPL.forward, MAML.clone/adapt, torch.optim.Adam, BaseAdaptor.*;
 * SURVEY.md 8b).  dynaboa_amd/ reproduces that surface in Python and calls down into the entry
 * points below; each one names the reference code whose arithmetic it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 unless stated otherwise; the library never
 *     allocates, frees or retains caller memory; scratch is passed in (`ws`, size from the
 *     matching *_workspace_bytes);
 *   - everything is enqueued on `stream` in order; no call synchronises the host - with ONE exception, stated where it occurs:
 *     dyb_stepper_adapt_frame_full / _frames_full with the dynamic-BOA gate on poll device-written pinned memory once per gate
 *     check (the reference's `.item()`, dynaboa_benchmark.py:165) to decide which sequences take a further step;
 *   - return value: 0 = ok, -1 bad argument, -2 launch failure, -3 unsupported shape,
 *     -4 workspace too small;
 *   - one host thread per GPU process; re-entrant across different streams + workspaces.
 *   - `const T* const*` arguments are HOST arrays of device pointers;
 *   - activations are NHWC, conv weights [R][S][Cin][Cout] (Cin padded to >= 4), linear weights
 *     (out, in) row-major with the row stride padded to a multiple of 4 floats.
 */
#ifndef DYNABOA_HIP_H
#define DYNABOA_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* dyb_stream_t; /* == hipStream_t */

/* ---- convolution (implicit GEMM on fp32 MFMA) ------------------------------------------------
 * replaces nn.Conv2d(bias=False) forward and its autograd in the ResNet-50 backbone:
 * reference model/hmr.py:28-33,40-56 (Bottleneck), :70,110-114,138 (stem / downsample). */
size_t dyb_conv2d_workspace_bytes(int N, int H, int W, int C, int K, int R, int S, int stride, int pad);
int dyb_conv2d_nhwc_fwd(const float* x, const float* w, float* y, int N, int H, int W, int C, int K, int R, int S,
                        int stride, int pad, void* ws, size_t ws_bytes, dyb_stream_t stream);
int dyb_conv2d_nhwc_dgrad(const float* dy, const float* w, float* dx, const float* addend, int N, int H, int W, int C,
                          int K, int R, int S, int stride, int pad, void* ws, size_t ws_bytes, dyb_stream_t stream);
int dyb_conv2d_nhwc_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int K, int R, int S,
                          int stride, int pad, void* ws, size_t ws_bytes, dyb_stream_t stream);

/* ---- GroupNorm(4 groups, eps 1e-5) + optional residual add + optional ReLU --------------------
 * replaces nn.GroupNorm(32//8, C) (reference model/hmr.py:14-18), nn.ReLU and `out += residual`
 * (:43-58).  `slabs`/`nslabs` let the kernel fold split-K partial sums of the preceding conv.
 * stats = [N][4][2] (mean, rstd) saved for backward. */
size_t dyb_groupnorm_workspace_bytes(int N, int HW, int C);
int dyb_groupnorm_fwd(const float* slabs, int nslabs, float* y, const float* gamma, const float* beta,
                      const float* residual, float* out, float* stats, int N, int HW, int C, int relu, void* ws,
                      size_t ws_bytes, dyb_stream_t stream);
int dyb_groupnorm_bwd(const float* dout, const float* out, const float* y, const float* stats, const float* gamma,
                      float* dy, float* dres, float* dgamma, float* dbeta, int N, int HW, int C, int relu, void* ws,
                      size_t ws_bytes, dyb_stream_t stream);

/* as dyb_groupnorm_bwd, but the incoming gradient is sum_z dout_slabs[z*slab_stride + .] (+ addend),
 * i.e. the un-folded split-K slabs of the data-gradient convolution that produced it plus the
 * residual-edge gradient; the sum is formed once inside the reduce kernel and written to `folded`. */
int dyb_groupnorm_bwd_fold(const float* dout_slabs, int nslabs, size_t slab_stride, const float* addend, float* folded,
                           const float* out, const float* y, const float* stats, const float* gamma, float* dy,
                           float* dres, float* dgamma, float* dbeta, int N, int HW, int C, int relu, void* ws,
                           size_t ws_bytes, dyb_stream_t stream);

/* One-pass GroupNorm backward of ONE image (the throughput schedule's form: one image per sequence replica; replaces
 * nn.GroupNorm's autograd, reference model/hmr.py:14-18): every input is read once and every output written once - a workgroup
 * keeps its rows of an (image, group) slab in registers between the sums and the apply; slabs beyond `cap` float4 per workgroup
 * (0: the policy's, 2048) are cut into up to 32 row chunks whose workgroups meet on a counter inside `ws`.  Incoming gradient = sum_z
 * dout_slabs[z*slab_stride + .] (+ addend); out = the saved activation (ReLU mask) or NULL: the mask is recomputed from y
 * (beta required); dm (may be NULL) receives the masked gradient.  DYB_ERR_UNSUPPORTED when the shape does not qualify
 * (C a power of two in 64..2048, a slab of at most 32 chunks). */
size_t dyb_groupnorm_bwd_onepass_workspace_bytes(int HW, int C);
int dyb_groupnorm_bwd_onepass(const float* dout_slabs, int nslabs, size_t slab_stride, const float* addend, const float* out,
                              const float* y, const float* stats, const float* gamma, const float* beta, float* dm, float* dy,
                              float* dgamma, float* dbeta, int HW, int C, int relu, int cap, void* ws, size_t ws_bytes,
                              dyb_stream_t stream);

/* GroupNorm backward split for consumers that form dy on the fly: the reduce half alone.  dm = dout
 * masked by the ReLU (relu == 0: pass dm == dout, nothing is copied); `part`
 * (dyb_groupnorm_bwd_partial_floats floats) receives the per-channel / per-group partial sums.  The
 * two convolution gradients below consume (dm, y_gn, stats, part, gamma) in their operand loaders -
 * dy = rstd*(gamma*dm - c1 - xhat*c2) never exists in memory - and the weight-gradient launch also
 * writes dgamma / dbeta.  Same autograd as dyb_groupnorm_bwd + dyb_conv2d_nhwc_dgrad/_wgrad, one
 * dependent launch less per layer (the engine's critical chain is launch-latency-bound at batch 1). */
size_t dyb_groupnorm_bwd_partial_floats(int N, int HW, int C);
int dyb_groupnorm_bwd_reduce(const float* dout, const float* out, const float* y, const float* stats,
                             const float* gamma, const float* beta, float* dm, float* part, int N, int HW, int C,
                             int relu, dyb_stream_t stream);
int dyb_conv2d_nhwc_dgrad_gn(const float* dm, const float* y_gn, const float* stats, const float* part,
                             const float* gamma, const float* w, float* dx, const float* addend, int N, int H, int W,
                             int C, int K, int R, int S, int stride, int pad, void* ws, size_t ws_bytes,
                             dyb_stream_t stream);
int dyb_conv2d_nhwc_wgrad_gn(const float* x, const float* dm, const float* y_gn, const float* stats, const float* part,
                             const float* gamma, float* dw, float* dgamma, float* dbeta, int N, int H, int W, int C,
                             int K, int R, int S, int stride, int pad, void* ws, size_t ws_bytes, dyb_stream_t stream);

/* GroupNorm forward split the same way.  dyb_groupnorm_stats: the statistics half alone (partials =
 * dyb_groupnorm_workspace_bytes(N,HW,C) bytes).  dyb_groupnorm_apply: the apply half; its residual may be
 * NULL, plain, or (res_partials != NULL) the GroupNorm without ReLU of a raw conv output - the
 * shortcut branch's downsample.1 (model/hmr.py:66-70) - normalised on the fly.
 * dyb_conv2d_nhwc_fwd_gnin: y = conv(relu?(gn(y_prev))) with the producer's GroupNorm applied in the
 * operand loader from its partials (inside a bottleneck, model/hmr.py:43-52, bn1/bn2 outputs have a
 * single consumer and are never written); dyb_conv2d_nhwc_wgrad_gn_gnin is the matching weight gradient. */
int dyb_groupnorm_stats(const float* slabs, int nslabs, float* y, float* partials, int N, int HW, int C,
                        dyb_stream_t stream);
int dyb_groupnorm_apply(const float* y, const float* partials, const float* gamma, const float* beta,
                        const float* residual, const float* res_partials, const float* res_gamma, const float* res_beta,
                        float* res_stats, float* out, float* stats, int N, int HW, int C, int relu, dyb_stream_t stream);
int dyb_conv2d_nhwc_fwd_gnin(const float* y_prev, const float* part_prev, const float* gamma_prev, const float* beta_prev,
                             int relu_prev, float* stats_prev_out, const float* w, float* y, int N, int H, int W, int C,
                             int K, int R, int S, int stride, int pad, void* ws, size_t ws_bytes, dyb_stream_t stream);
int dyb_conv2d_nhwc_wgrad_gn_gnin(const float* y_prev, const float* stats_prev, const float* gamma_prev,
                                  const float* beta_prev, int relu_prev, const float* dm, const float* y_gn,
                                  const float* stats, const float* part, const float* gamma, float* dw, float* dgamma,
                                  float* dbeta, int N, int H, int W, int C, int K, int R, int S, int stride, int pad,
                                  void* ws, size_t ws_bytes, dyb_stream_t stream);

/* Measurement aid for bench.py's roofline leg: between _begin and _end every convolution kernel launched
 * from the calling thread is timed on its own dispatch (start/stop events on the launch, no extra stream
 * packets).  _end (after a device synchronise) returns the summed kernel time, the launch count and the
 * algorithmic flop / bytes those launches stand for. */
int dyb_conv_timing_begin(int max_launches);
int dyb_conv_timing_end(double* ms_total, long long* launches, double* flop, double* bytes);
/* Per-shape breakdown of the scope closed last, as CSV text (header line first; kind f/d/w (t/u/v: throughput form) = tiled forward / data gradient /
 * weight gradient kernel, F/D = the single-launch 1x1 kernels).  Copies at most cap-1 bytes + terminator into buf; returns
 * the size needed.  tools/conv_table.py prints it. */
size_t dyb_conv_timing_table(char* buf, size_t cap);
double dyb_conv_timing_union_ms(void); /* of the scope closed last: time during which at least one timed launch was executing */
/* Diagnostic: while set (buf != NULL), launches of the throughput-form conv kernel whose mode (0 forward, 1 data gradient,
 * 2 weight gradient) and layer (H, C, K, R) match write per-wave phase clocks into buf ([workgroup][4 waves][8] 64-bit words,
 * cap_wgs workgroups of room; layout in igemm_conv.hip).  bench.py --probe / tools/tp_probe.py read it.  buf = NULL clears. */
int dyb_conv_probe_set(void* buf, long cap_wgs, int mode, int H, int C, int K, int R);

/* ---- second-order building blocks (exact Hessian-vector products, hvp_kernels.hip / hvp_engine.inc) ----
 * Tangent ("t" prefix = directional derivative along a parameter direction) of GroupNorm(4, C)(+ReLU)(+residual) and of its
 * backward; NHWC [N][HW][C]; stats = the forward's [N][4][2] (mean, rstd); tstats [N][4][2] receives / supplies the tangent
 * statistics.  out may be NULL.  ty2 (may be NULL): a second half of the tangent, added into ty first (the two halves of a
 * convolution's tangent).  dyb_gn_jvp_bwd: dout / tdout = gradient w.r.t. the layer's output and its tangent, out_mask =
 * the primal output (ReLU mask, relu = 1); writes the masked gradients (dm, tdm; may be NULL), the gradient w.r.t. y and its
 * tangent (dy, tdy) and the tangents of dgamma / dbeta ([C]).  scratch: dyb_gn_jvp_scratch_floats(N, HW, C) floats, 8-byte aligned
 * (each (image, group) slab is cut into row chunks, one workgroup each; partial sums meet there).  Not replica-aware. */
size_t dyb_gn_jvp_scratch_floats(int N, int HW, int C);
int dyb_gn_jvp_fwd(const float* y, float* ty, const float* ty2, const float* stats, const float* gamma, const float* beta,
                   const float* tgamma, const float* tbeta, const float* res, const float* tres, float* out, float* tout, float* tstats,
                   float* scratch, int N, int HW, int C, int relu, dyb_stream_t stream);
int dyb_gn_jvp_bwd(const float* dout, const float* tdout, const float* out_mask, const float* y, const float* ty, const float* stats,
                   const float* tstats, const float* gamma, const float* tgamma, float* dm, float* tdm, float* dy, float* tdy,
                   float* scratch, float* tdgamma, float* tdbeta, int N, int HW, int C, int relu, dyb_stream_t stream);
/* The same two as ONE launch each (the chunks of a slab meet on an arrival counter inside the launch instead of between two launches):
 * sync = dyb_gn_jvp_sync_words(N) 32-bit words - an error word, then one counter per (image, group) slab - zero on entry and private to
 * the call until it has finished on the stream.  A wait that lasts 0.2 s raises sync[0] and goes on (results then wrong). */
size_t dyb_gn_jvp_sync_words(int N);
int dyb_gn_jvp_fwd_onepass(const float* y, float* ty, const float* ty2, const float* stats, const float* gamma, const float* beta,
                           const float* tgamma, const float* tbeta, const float* res, const float* tres, float* out, float* tout,
                           float* tstats, float* scratch, unsigned* sync, int N, int HW, int C, int relu, dyb_stream_t stream);
int dyb_gn_jvp_bwd_onepass(const float* dout, const float* tdout, const float* out_mask, const float* y, const float* ty,
                           const float* stats, const float* tstats, const float* gamma, const float* tgamma, float* dm, float* tdm,
                           float* dy, float* tdy, float* scratch, unsigned* sync, float* tdgamma, float* tdbeta, int N, int HW, int C,
                           int relu, dyb_stream_t stream);
/* Exact Hessian-vector product through HMR, forward-over-reverse (hvp_engine.inc).  acts = the arena dyb_hmr_forward filled at
 * (params, image); tparams = the direction v (parameter arena layout); dual = scratch of dyb_hmr_hvp_dual_floats floats shared by
 * the two passes.  _jvp_forward leaves the tangent of the regressor's final state [B][160] at dual + dyb_hmr_hvp_offset_tstate;
 * the caller differentiates the head (rot6d -> SMPL -> losses) along it; _jvp_backward takes the head's gradient w.r.t. that state
 * (d_state, rot6d folded in) and its tangent (td_state) and writes hv = H v in the parameter arena layout (tensor spans only: zero
 * hv first).  Eval mode, fp32, one sequence per call.  aux (may be NULL): a side stream; the halves of every tangent pair that
 * are off the dependency chain (conv(x, tw), the weight-gradient pairs, dgrad(dy, tw)) are issued there, ordered by events, and
 * everything has joined `stream` again when a call returns. */
size_t dyb_hmr_hvp_dual_floats(const void* plan);
long long dyb_hmr_hvp_offset_tstate(const void* plan);
int dyb_hmr_jvp_forward(void* plan, const float* params, const float* tparams, const float* acts, float* dual, int n_iter, void* ws,
                        size_t ws_bytes, dyb_stream_t stream, dyb_stream_t aux);
int dyb_hmr_jvp_backward(void* plan, const float* params, const float* tparams, const float* acts, float* dual, const float* d_state,
                         const float* td_state, int n_iter, float* hv, void* ws, size_t ws_bytes, dyb_stream_t stream, dyb_stream_t aux);
/* tangent of MaxPool2d(3,2,1): ty = tx gathered at the tap indices dyb_maxpool3x3s2_fwd stored */
int dyb_maxpool3x3s2_jvp(const float* tx, const uint32_t* idx, float* ty, int N, int H, int W, int C, dyb_stream_t stream);

/* One forward layer = conv + the GroupNorm statistics of its output (what the engine issues per layer): y and *nchunks
 * partial records [G][2] in `partials` (>= dyb_groupnorm_workspace_bytes(N, Ho*Wo, K) and >= 4*(Ho*Wo/32+1)*(K/32)*8
 * bytes).  part_prev != NULL: x is the producer's raw output, normalised in the loader from its nch_prev partials.
 * Small 1x1 layers at batch 1 run as ONE launch (statistics in the conv epilogue); dyb_groupnorm_apply_n is
 * dyb_groupnorm_apply with the partial counts given explicitly. */
int dyb_conv2d_nhwc_fwd_gnstats(const float* x, const float* part_prev, int nch_prev, const float* gamma_prev,
                                const float* beta_prev, int relu_prev, float* stats_prev_out, const float* w, float* y,
                                float* partials, int* nchunks, int N, int H, int W, int C, int K, int R, int S, int stride,
                                int pad, void* ws, size_t ws_bytes, dyb_stream_t stream);
int dyb_groupnorm_apply_n(const float* y, const float* partials, int nch, const float* gamma, const float* beta,
                          const float* residual, const float* res_partials, int res_nch, const float* res_gamma,
                          const float* res_beta, float* res_stats, float* out, float* stats, int N, int HW, int C, int relu,
                          dyb_stream_t stream);

/* Backward mirror of dyb_conv2d_nhwc_fwd_gnstats: the data gradient of a conv fused with the GroupNorm-backward reduce
 * of the PRODUCER of its input (GroupNorm over [N][H*W][C]): dm_p and the partial block part_p
 * (dyb_groupnorm_bwd_partial_floats(N, H*W, C) floats) come out, *nch_p / *ncolb_p give part_p's layout; dx itself only
 * exists in dx_scratch (or not at all).  With DYB_K4_BWD=1 small 1x1 layers at batch 1 are ONE launch; otherwise it is
 * dyb_conv2d_nhwc_dgrad_gn + dyb_groupnorm_bwd_reduce.  (nch, ncolb) describe `part` of THIS conv's GroupNorm (0, 0 = the
 * dyb_groupnorm_bwd_reduce layout).  The _n gradients take such an explicit layout; in dyb_conv2d_nhwc_wgrad_gn_n the
 * conv input is x, or - x == NULL - relu(gn(y_prev)) formed in the loader. */
int dyb_conv2d_nhwc_dgrad_gn_reduce(const float* dm, const float* y_gn, const float* stats, const float* part, int nch,
                                    int ncolb, const float* gamma, const float* w, const float* addend, const float* y_p,
                                    const float* out_p, const float* stats_p, const float* gamma_p, const float* beta_p,
                                    float* dm_p, float* part_p, int* nch_p, int* ncolb_p, float* dx_scratch, int N, int H,
                                    int W, int C, int K, int R, int S, int stride, int pad, void* ws, size_t ws_bytes,
                                    dyb_stream_t stream);
int dyb_conv2d_nhwc_dgrad_gn_n(const float* dm, const float* y_gn, const float* stats, const float* part, int nch, int ncolb,
                               const float* gamma, const float* w, float* dx, const float* addend, int N, int H, int W, int C,
                               int K, int R, int S, int stride, int pad, void* ws, size_t ws_bytes, dyb_stream_t stream);
int dyb_conv2d_nhwc_wgrad_gn_n(const float* x, const float* y_prev, const float* stats_prev, const float* gamma_prev,
                               const float* beta_prev, const float* dm, const float* y_gn, const float* stats,
                               const float* part, int nch, int ncolb, const float* gamma, float* dw, float* dgamma,
                               float* dbeta, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, void* ws,
                               size_t ws_bytes, dyb_stream_t stream);

/* ---- pooling / layout: nn.MaxPool2d(3,2,1), nn.AvgPool2d(7) (reference model/hmr.py:73,78,142,155)
 * and the NCHW(3) -> NHWC(4) repack of the dataloader image (boa_dataset/pw3d.py:115). */
int dyb_nchw3_to_nhwc4(const float* x, float* y, int N, int H, int W, dyb_stream_t stream);
int dyb_maxpool3x3s2_fwd(const float* x, float* y, uint32_t* idx, int N, int H, int W, int C, dyb_stream_t stream);
int dyb_maxpool3x3s2_bwd(const float* dy, const uint32_t* idx, float* dx, int N, int H, int W, int C,
                         dyb_stream_t stream);
int dyb_avgpool_fwd(const float* x, float* const* dsts, int ndst, int ld, int N, int HW, int C, dyb_stream_t stream);
int dyb_avgpool_bwd(const float* dxf, int ld, float* dx, int N, int HW, int C, dyb_stream_t stream);

/* ---- linear layers of the iterative regressor: nn.Linear fc1/fc2/decpose/decshape/deccam
 * (reference model/hmr.py:82-90,161-172) and their autograd. */
int dyb_linear_fwd(const float* x, int ldx, const float* w, int ldw, const float* bias, const float* res, int ldres,
                   float* y, int ldy, int B, int I, int O, dyb_stream_t stream);
size_t dyb_linear_bwd_workspace_bytes(int B, int I, int O);
int dyb_linear_bwd_dx(const float* dy, int lddy, const float* w, int ldw, int B, int I, int O, float* dstA, int ldA,
                      int accA, int split_col, float* dstB, int ldB, const float* addB, int ldaddB, void* ws,
                      size_t ws_bytes, dyb_stream_t stream);
int dyb_linear_bwd_dw(const float* const* dys, const int* lddys, const float* const* xs, const int* ldxs, int T, int B,
                      int I, int O, float* dw, int ldw, float* db, dyb_stream_t stream);

/* ---- rotation maps: utils/geometry.py:47-61 (rot6d_to_rotmat) and :184-306
 * (rotation_matrix_to_angle_axis), forward and backward. */
int dyb_rot6d_fwd(const float* x6, int ldx, float* rotmat, int B, dyb_stream_t stream);
int dyb_rot6d_bwd(const float* x6, int ldx, const float* drotmat, float* dx6, int lddx, int B, dyb_stream_t stream);
/* smplx.lbs.batch_rodrigues (pose2rot=True path of SMPL.forward; metric ground truth only) */
int dyb_rodrigues_fwd(const float* axis_angle, float* rotmat, int n, dyb_stream_t stream);
int dyb_rotmat_to_aa_fwd(const float* R, float* aa, int n, dyb_stream_t stream);
int dyb_rotmat_to_aa_bwd(const float* R, const float* daa, float* dR, int n, dyb_stream_t stream);

/* ---- SMPL forward/backward: smplx lbs + vertex joints + J_regressor_extra + 49-joint gather
 * (reference model/smpl.py:25-37).
 * tables_f = {v_template[6890*3], shapedirs[6890*3][10], posedirs[207][6890*3], weights_t[24][6890],
 *             j_template[24*3], j_shapedirs[72][10], j_extra[9][6890]}
 * tables_i = {parents[24], vertex_joint_ids[21], joint_map[49]}        (device int32) */
size_t dyb_lbs_saved_floats(int B);
size_t dyb_lbs_bwd_workspace_bytes(int B);
int dyb_lbs_fwd(const float* const* tables_f, const int* const* tables_i, const float* betas, int ldb,
                const float* rotmat, float* verts, float* joints49, float* saved, int B, dyb_stream_t stream);
int dyb_lbs_bwd(const float* const* tables_f, const int* const* tables_i, const float* rotmat, const float* saved,
                const float* djoints49, const float* dverts, float* drot, float* dbetas, int lddb, int B, void* ws,
                size_t ws_bytes, dyb_stream_t stream);
/* J_regressor_h36m @ vertices of the metric path (reference dynaboa_benchmark.py:220-233) */
int dyb_regress_joints(const float* reg, const float* verts, float* out, int nj, int B, dyb_stream_t stream);

/* ---- losses: BaseAdaptor.projection (base_adaptor.py:160-170) and the frame-loss block of
 * lower/upper_level_adaptation (:229-240 / :279-289 with cal_shape_prior :401, cal_pose_prior :405,
 * MaxMixturePrior.merged_log_likelihood utils/smplify/prior.py:181-196).  dyb_frame_losses returns
 * the three loss values + weighted total AND the gradient of the total in one launch. */
int dyb_projection_fwd(const float* cam, int ldc, const float* p3, float* p2, int B, int np, dyb_stream_t stream);
int dyb_projection_bwd(const float* cam, int ldc, const float* p3, const float* g2, float* dp3, float* dcam, int lddc,
                       int B, int np, dyb_stream_t stream);
/* utils/geometry.py:63-91 perspective_projection(points (B,N,3), rotation (B,3,3), translation (B,3), focal_length, camera_center
 * (B,2)) -> (B,N,2) and its gradient w.r.t. points and translation; focal: per sample with stride ldf floats (0: one value). */
int dyb_perspective_projection_fwd(const float* points, const float* rotation, const float* translation, const float* focal, int ldf,
                                   const float* center, float* out, int B, int np, dyb_stream_t stream);
int dyb_perspective_projection_bwd(const float* points, const float* rotation, const float* translation, const float* focal, int ldf,
                                   const float* g2, float* dpoints, float* dtranslation, int B, int np, dyb_stream_t stream);
/* MaxMixturePrior.forward / merged_log_likelihood (utils/smplify/prior.py:181-196) on an axis-angle body pose [B][69]: out[B] =
 * per-sample min over the 8 Gaussians, dpose69 (may be NULL) = its gradient. */
int dyb_gmm_prior(const float* pose69, const float* gmm_means, const float* gmm_prec, const float* gmm_logw, float* out,
                  float* dpose69, int B, dyb_stream_t stream);
int dyb_frame_losses(const float* rotmat, const float* shape, int lds, const float* cam, int ldc, const float* joints49,
                     const float* kp2d, const float* gmm_means, const float* gmm_prec, const float* gmm_logw, float w2d,
                     float wshape, float wpose, float* losses_out, float* drot, float* dshape, int ldds, float* dcam,
                     int lddc, float* djoints49, int B, void* ws, size_t ws_bytes, dyb_stream_t stream);

/* The other terms of the level losses, value + gradient in one launch each (B <= 16): mode 0 mean-teacher consistency
 * (reference base_adaptor.py:320-343: 5 mse(s2d) + 5 mse(s3d) + 0.001 mse(shape) + mse(rotmat) against the teacher's outputs
 * rot2 / shape2 / cam2 / joints2), mode 1 motion (:379-398: confidence-masked mse of the projected-keypoint motion between the
 * frame and the history frame - cam2 / joints2 are the history pass's outputs, kp / kp2 the two frames' keypoints), mode 2
 * labelled exemplar (:346-376 with the hip-centred 3-D loss of :412-422; kp / gt_rot / gt_betas / gt_s3d [B][24][4] are the
 * exemplar's annotations).  s2d is the normalised projection (:160-170) formed inside.  Gradients of weight * term w.r.t. the
 * student pass's rotmat [B][216] / shape [B][10] / cam [B][3] / joints49 [B][147] (accumulate != 0: added) and, mode 1, w.r.t.
 * the history pass's cam / joints49.  vals5 = {s2d, s3d, shape, pose, loss} un-weighted (mode 1: {motion, 0, 0, 0, motion}). */
int dyb_aux_loss_terms(int mode, int B, int accumulate, float weight, const float* rot, const float* shape, int lds,
                       const float* cam, int ldc, const float* joints49, const float* rot2, const float* shape2, int lds2,
                       const float* cam2, int ldc2, const float* joints2, const float* kp, const float* kp2,
                       const float* gt_rot, const float* gt_betas, const float* gt_s3d, float* vals5, float* d_rot,
                       float* d_shape, float* d_cam, float* d_joints49, float* d_cam2, float* d_joints2, dyb_stream_t stream);

/* Gradient assembly of the fused HMR + SMPL + frame-loss node (one autograd node per adaptation level instead
 * of three plus glue): out = g*a (+ ext) with g a device scalar (NULL = 1), and the d_rotmat / d_state inputs of
 * dyb_hmr_backward from the dyb_frame_losses pieces (scaled by g), the dyb_lbs_bwd pieces and optional external
 * gradients on rotmat / shape / cam (teacher, motion and label terms attach there). */
int dyb_scale_add(const float* g, const float* a, const float* ext, float* out, size_t n, dyb_stream_t stream);
int dyb_head_grad_combine(const float* g, const float* drot_loss, const float* drot_smpl, const float* drot_ext,
                          const float* dshape_loss, const float* dbetas_smpl, const float* dshape_ext,
                          const float* dcam_loss, const float* dcam_ext, float* d_rot, float* d_state, int B,
                          dyb_stream_t stream);

/* PA-MPJPE on the device: compute_similarity_transform(_batch) of reference utils/pose_utils.py:9-64 (centre, K = X1^T X2,
 * 3x3 SVD, reflection fix, scale, translation) + the mean per-joint error of dynaboa_benchmark.py:236-240, one sample per
 * thread, so that the metric path ships scalars instead of joint sets.  aligned (optional) receives the aligned prediction. */
int dyb_pa_mpjpe(const float* pred, const float* gt, float* out, float* aligned, int n, int J, dyb_stream_t stream);

/* ---- flat-arena updates: learn2learn MAML.adapt (call sites dynaboa_benchmark.py:136,140),
 * torch.optim.Adam (base_adaptor.py:126; dynaboa_benchmark.py:149-151), update_teacher
 * (base_adaptor.py:193-201), cal_feature_diff's cosine (:211-219). n = float count, multiple of 4. */
int dyb_fastweight_update(const float* p, const float* g, float* out, float lr, size_t n, dyb_stream_t stream);
int dyb_adam_step(float* p, const float* g, float* m, float* v, float beta1, float beta2, float step_size,
                  float bc2_sqrt, float eps, size_t n, dyb_stream_t stream);
/* Adam on the gradient g - alpha * h: the second-order path's last accumulation (v - lr * H v) fused into the optimiser step */
int dyb_adam_step_accum(float* p, const float* g, const float* h, float alpha, float* m, float* v, float beta1, float beta2,
                        float step_size, float bc2_sqrt, float eps, size_t n, dyb_stream_t stream);
int dyb_ema_update(float* teacher, const float* p, float alpha, size_t n, dyb_stream_t stream);
int dyb_axpby(const float* x, float* y, float a, float b, size_t n, dyb_stream_t stream);
int dyb_cosine_sim(const float* a, const float* b, size_t n, float eps, float* out, dyb_stream_t stream);

/* run-time switches of the dispatch policy (A/B runs, tests).  Read from the environment (DYB_K4, DYB_K4_BWD,
 * DYB_K4_BATCH, DYB_K4_MAXC) once at first use - never on the dispatch path - and changed here afterwards.
 * names: "k4" single-launch 1x1 forward conv + statistics; "k4_bwd" 1x1 data gradient carrying the producer's
 * GroupNorm-backward reduce; "k4_batch" both at batch > 1; "k4_maxc" their channel limit. */
/* Diagnostic: one convolution mode (0 forward, 1 data gradient, 2 weight gradient) for `nrep` sequence replicas in one launch -
 * x / w / dy / out are [nrep][...] stacks, every replica with its own weights (what the native stepper's launches look like); used by
 * tools/tp_lab.py to time a layer's kernels alone. */
int dyb_debug_conv_replicas(int mode, const float* x, const float* w, const float* dy, float* out, int nrep, int N, int H, int W, int C,
                            int K, int R, int S, int stride, int pad, void* ws, size_t ws_bytes, dyb_stream_t stream);
/* diagnostic (tools/gn_lab.py): dyb_groupnorm_bwd_onepass for nrep replicas in one launch; blob = [nrep] x { din | y | out | dm | dy |
 * stats(8) | dgamma | dbeta | workspace }, blob_floats per replica; mode bit 0: mask from the saved activation, bit 1: write dm */
int dyb_debug_gn_onepass_replicas(float* blob, size_t blob_floats, int nrep, const float* gamma, const float* beta, int HW, int C, int relu,
                                  int mode, dyb_stream_t stream);
/* diagnostic / tests: the operand-pair convolution the tangent passes of the exact Hessian-vector product are made of (latency form,
 * csrc/hvp_engine.inc): out = op(a1, b1) + op(a2, b2) (+ addend; data gradient only) as ONE launch with a K loop over both pairs.
 * mode 0 forward (a = x [N][H][W][C], b = w [R][S][C][K]), 1 data gradient (a = dy [N][Ho][Wo][K], b = w), 2 weight gradient (a = x,
 * b = dy).  DYB_ERR_UNSUPPORTED where the throughput schedule or the bf16 form is in force (option "conv_pair" = 0 turns pairs off). */
int dyb_debug_conv_pair(int mode, const float* a1, const float* b1, const float* a2, const float* b2, float* out, const float* addend,
                        int N, int H, int W, int C, int K, int R, int S, int stride, int pad, void* ws, size_t ws_bytes,
                        dyb_stream_t stream);
/* In-kernel hand-offs (one-pass GroupNorm backward: the workgroups of a slab meet on a counter) give up after ~0.2 s and go on with
 * incomplete sums rather than block the queue.  This returns how many did since the library was loaded; it waits for `stream`.
 * The reference has no counterpart (PyTorch kernels do not rendezvous); the drivers check it at their metric flush and raise. */
int dyb_sync_error_count(unsigned* count_host, dyb_stream_t stream);
/* tests / lab: counter region (nwords zeroed 32-bit words, device memory) for the in-kernel split-K fold of the calling thread's plain
 * conv calls, until reset with (NULL, 0).  With a region in scope a split launch's last-arriving workgroup per tile adds the slabs
 * itself (csrc/igemm_tp.inc); the engine passes its own regions per pass.  Option "stat_folds" counts such launches. */
int dyb_debug_set_conv_sync(unsigned* ctr, int nwords);
/* tests / lab ("fuse_fast", round 6): while set, a throughput-form weight gradient of the calling thread that runs UNSPLIT and whose
 * result would land inside [grads, grads + bytes) writes p_next[off] = p_cur[off] - lr * g from its accumulators instead of g - the MAML
 * fast-weight step (learn2learn MAML.adapt: p' = p - lr * dL/dp; reference dynaboa_benchmark.py:136,140) fused into the convolution's
 * epilogue.  Reset with grads = NULL.  dyb_debug_wgrad_update_spans: how many launches took that form since the scope was set.  The frame
 * stepper opens such a scope around every lower level's backward (csrc/adapt_step.hip). */
int dyb_debug_set_wgrad_update(const float* grads, size_t bytes, const float* p_cur, float* p_next, float lr);
int dyb_debug_wgrad_update_spans(void);
/* the same for Adam ("fuse_adam"): an unsplit throughput-form weight gradient whose result would land inside [grads, grads + bytes) applies
 * torch.optim.Adam's single-tensor step (reference base_adaptor.py:126, dynaboa_benchmark.py:149-151) to theta / m / v IN PLACE at the same
 * offset from its accumulators; sc = device pointer to (step_size = lr / (1 - beta1^t), bc2_sqrt = sqrt(1 - beta2^t)).  Reset with
 * dyb_debug_set_wgrad_update(NULL, ...).  The frame stepper opens such a scope around the outer level's backward of a replica group. */
int dyb_debug_set_wgrad_adam(const float* grads, size_t bytes, float* theta, float* m, float* v, const float* sc, float beta1, float beta2,
                             float eps);
int dyb_set_option(const char* name, int value);
int dyb_get_option(const char* name, int* value);

/* diagnostic: n dependent launches of a trivial kernel (per-launch floor of a kernel chain) */
int dyb_debug_launch_chain(float* scratch, int n, int blocks, dyb_stream_t stream);

/* ---- native HMR engine: HMR.forward (reference model/hmr.py:127-181) and its backward over a
 * static plan.  Parameter / activation arena layouts are queried from the plan. */
int dyb_hmr_plan_create(int B, int H, int W, void** plan);
void dyb_hmr_plan_destroy(void* plan);
size_t dyb_hmr_param_floats(const void* plan);
size_t dyb_hmr_act_floats(const void* plan);
size_t dyb_hmr_workspace_bytes(const void* plan);
int dyb_hmr_num_tensors(const void* plan);
/* kind: 0 conv weight, 1 norm weight, 2 norm bias, 3 fc weight, 4 fc bias, 5 fused decoder weight
 * [160][1024] (decpose 0..143 | decshape 144..153 | deccam 154..156), 6 fused decoder bias */
int dyb_hmr_tensor_info(const void* plan, int i, char* name, int name_cap, int* kind, long long* offset, int* dims4,
                        int* cin_pad);
int dyb_hmr_feature_info(const void* plan, int which, long long* offset, int* dims4, int* row_stride);
long long dyb_hmr_act_offset_rotmat(const void* plan); /* [B][24][9] */
long long dyb_hmr_act_offset_state(const void* plan);  /* [B][160] pose|shape|cam|pad */
/* graph mode: whole forward / backward calls are captured into hipGraphs keyed by their pointer
 * arguments (second sighting of a key) and replayed; results are those of the eager path. */
int dyb_hmr_set_graph_mode(void* plan, int on);
/* bf16-MFMA variant of the plan's convolutions (BASELINE configs[4]; never the parity default): fp32 master weights, fp32
 * activations / GroupNorm statistics / accumulators, operand tiles rounded to bf16 (nearest even) as they are staged for
 * v_mfma_f32_32x32x16_bf16.  The option "bf16" (dyb_set_option) does the same for direct calls of the conv entry points. */
int dyb_hmr_set_bf16(void* plan, int on);
int dyb_hmr_graph_stats(const void* plan, long long* stats10); /* replays, eager, captures, fwd keys, bwd keys, 5 failure counters */
int dyb_hmr_forward(void* plan, const float* params, const float* image_nchw, const float* init_state, int n_iter,
                    float* acts, void* ws, size_t ws_bytes, dyb_stream_t stream);
/* aux_stream (may be NULL): second stream for the weight-gradient convolutions, which are off the
 * critical path; `stream` waits for them before the call returns control of ordering. */
int dyb_hmr_backward(void* plan, const float* params, const float* acts, const float* d_rotmat, const float* d_state,
                     int n_iter, float* grads, void* ws, size_t ws_bytes, dyb_stream_t stream, dyb_stream_t aux_stream);

/* ---- frame preprocessing: bounding-box crop + anti-aliased bilinear resize + /255 + Normalize + HWC->CHW ----------------
 * reference utils/dataprocess.py:48-96 crop() (test-time path: rot = 0) as called by boa_dataset/pw3d.py:131-136 and
 * base_adaptor.py:529-533, with skimage.transform.resize 0.17.2 semantics (Gaussian anti-aliasing sigma = (in/out - 1)/2,
 * 'mirror' boundary, bilinear warp at (j + 0.5) * in/out - 0.5), then pw3d.py:121-123.  img: decoded frame [H][W][3] uint8
 * RGB (device); (ul, br): box corners in frame pixels as the reference's transform(..., invert=1) gives them (may lie
 * outside the frame: zero fill); out: [3][res][res] fp32.  ws: dyb_crop_workspace_bytes(br_y - ul_y, br_x - ul_x). */
size_t dyb_crop_workspace_bytes(int box_h, int box_w);
int dyb_crop_resize_normalize(const uint8_t* img, int H, int W, int ul_x, int ul_y, int br_x, int br_y, float* out, int res,
                              float mean0, float mean1, float mean2, float std0, float std1, float std2, void* ws,
                              size_t ws_bytes, dyb_stream_t stream);

/* ---- native frame stepper: Adaptor.adaptation (reference dynaboa_benchmark.py:126-193) as ONE call per frame --------
 * The first-order bilevel schedule with the frame-loss set - clone, inner_step x [lower-level loss
 * (base_adaptor.py:222-268) -> learner.adapt -> inference], upper-level loss (:270-317) through the fast weights,
 * zero_grad / backward / Adam.step, inference (:204-262) - issued from C++ over the entry points above: same kernels, same
 * order and operands as the Python composition (dynaboa_amd/benchmark.py), hence identical weights and Adam state, at
 * ~2.5 us of host time per launch instead of ~5.  A stepper owns no memory: theta / Adam moments / tables are the
 * caller's (dyb_stepper_set_p), scratch is one blob (dyb_stepper_workspace_bytes -> dyb_stepper_bind_workspace).
 * Steppers are independent: several may run on different streams from different host threads (sequence replicas
 * sharing one GPU).  Keys:
 *   set_i: n_iter, inner_step, eval_lower (metric record after every inner step), use_side, metrics, adam_step,
 *          record_capacity, loss_capacity, replicas
 *   set_f: lr, beta1, beta2, eps, fastlr, s2dloss_weight, shape_prior_weight, pose_prior_weight
 *   set_p: theta, adam_m, adam_v, init_state [B][160], gmm_means, gmm_precisions, gmm_log_weights, j_regressor_h36m
 *          [17][6890], j14 (device int32[14]), records, loss_log, smpl_{neutral,male,female}_{0..6} and
 *          smpli_{...}_{0..2} (the table order of dyb_lbs_fwd)
 *   get_i: adam_step, record_floats (floats per metric record: pred14 [B][14][3] | gt14 [B][14][3] | mpjpe [B] | pve),
 *          loss_floats (floats per frame in loss_log: (inner_step + 1) x {s2d, shape prior, pose prior, weighted total})
 * dyb_stepper_adapt_frame: gender is int64 [B]; gt_* / gender may be NULL with metrics = 0.  Records go to slots
 * record_slot.. (one per inner step when eval_lower, then the final one).  `aux`: weight-gradient stream (may be NULL);
 * `side`: stream for the final no-grad forward + its metrics (may be NULL), overlapped with the next frame.
 * dyb_stepper_join makes `stream` wait for the side stream's tail; dyb_stepper_output: 0 rotmat, 1 state, 2 vertices,
 * 3 joints49 of the last final inference. */
int dyb_stepper_create(void* plan, int B, int H, int W, void** stepper);
void dyb_stepper_destroy(void* stepper);
int dyb_stepper_set_i(void* stepper, const char* key, long long value);
int dyb_stepper_set_f(void* stepper, const char* key, double value);
int dyb_stepper_set_p(void* stepper, const char* key, const void* ptr);
long long dyb_stepper_get_i(const void* stepper, const char* key);
/* host issue time in ms accumulated over "host_frames" frame steps, by section: host_ms_forward / _backward / _head (loss head
 * + records) / _update (fast weights) / _tail (Adam + final inference) / _total */
double dyb_stepper_get_f(const void* stepper, const char* key);
size_t dyb_stepper_workspace_bytes(void* stepper);
int dyb_stepper_bind_workspace(void* stepper, void* ws, size_t bytes, dyb_stream_t stream);
int dyb_stepper_adapt_frame(void* stepper, const float* image, const float* kp2d, const float* gt_pose, const float* gt_betas,
                            const long long* gender, int record_slot, int loss_slot, dyb_stream_t stream, dyb_stream_t aux,
                            dyb_stream_t side);
/* The same step for `replicas` (set_i key, 1..64, before sizing the workspace) independent sequences in lockstep: every
 * launch of the chain covers all replicas (replica = a grid dimension; pointer arguments inside a replica's arenas are
 * rebased in the kernels), each replica with its own weights / Adam moments / workspace / records: theta, adam_m, adam_v are
 * [replicas][param floats], records [replicas][record_capacity][record_floats], loss_log [replicas][loss_capacity]
 * [loss_floats].  inputs: HOST array of 5 x replicas device pointers, kind-major: image[r], kp2d[r], gt_pose[r],
 * gt_betas[r], gender[r].  Results per replica are those of dyb_stepper_adapt_frame on that replica alone. */
int dyb_stepper_adapt_frames(void* stepper, const void* const* inputs, int record_slot, int loss_slot, dyb_stream_t stream,
                             dyb_stream_t aux, dyb_stream_t side);
/* The reference's FULL term set (its default flags) as one call per frame: per level the frame losses + (upper level) the
 * mean-teacher term with a no-grad teacher forward + the motion term with a second forward of the history frame + the
 * labelled-exemplar term with a forward of the retrieved exemplars (dyb_aux_loss_terms), gradients of the passes summed, Adam,
 * teacher EMA, final inference, and the dynamic-BOA loop (dynaboa_benchmark.py:161-192): the 15 feature cosines come from one
 * launch that also writes them to device-visible pinned host memory, which the call polls - its only host wait, no stream
 * synchronise - to decide on up to optim_steps further upper-level steps.  Enable with set_i "full" = 1 before sizing the
 * workspace; further keys: set_i temporal_lower/upper, use_teacher, use_motion, interval, mix_lower/upper, dynamic, optim_steps;
 * set_f teacherloss_weight, motionloss_weight, labelloss_weight, alpha, cos_sim_threshold; set_p teacher (parameter arena),
 * gate_host (16 floats, pinned), gate_log ([loss_capacity][1 + optim_steps][16]), feat5_out ([2048]), retrieve_fn / retrieve_user
 * (int fn(void* user, int level, const void** ex5): host retrieval of exemplars from feat5_out, batch 1).  inputs: HOST array of
 * 12 device pointers: image, kp2d, gt_pose, gt_betas, gender, hist_image, hist_kp2d (NULL: no motion term), ex_img, ex_kp,
 * ex_pose, ex_betas, ex_pose3d (NULL with a retrieval callback).  loss_log rows are 16 floats per level (adapt_step.hip). */
int dyb_stepper_adapt_frame_full(void* stepper, const void* const* inputs, int record_slot, int loss_slot, int* extra_steps,
                                 dyb_stream_t stream, dyb_stream_t aux);
/* The full term set for the stepper's sequence replicas in lockstep (round 3; reference dynaboa_benchmark.py:126-193 per sequence):
 * every launch - teacher forward, history-frame pass, exemplar pass, the feature cosines - covers all active replicas.  The
 * dynamic-BOA gate is decided PER replica: a replica whose feature 12 has stopped moving leaves the launch set of the remaining
 * iterations, so replicas take different numbers of Adam steps (per-replica step counts: get_i "adam_step_<r>").  inputs: HOST
 * array of 12 x replicas device pointers, kind-major - inputs[kind * replicas + r] in the order of dyb_stepper_adapt_frame_full,
 * r the physical replica (entries of inactive replicas ignored; the history pair present for all active replicas or none;
 * exemplars NULL with the per-replica callback set_p "retrieve_rep_fn": int fn(void* user, int level, int replica, const void**
 * ex5)).  teacher / gate_log / feat5_out are [replicas][...] like theta; gate_host 16 floats per replica.  extra_steps: `replicas`
 * ints.  A launch scope holds at most 8 per-replica arenas (workspace, theta, adam_m, adam_v, teacher + the logs): with replicas
 * the four log buffers (records, loss_log, gate_log, feat5_out) must be sub-buffers of ONE per-replica block registered with
 * set_p "logs_base" / set_i "logs_bytes" (replica r's block at logs_base + r * logs_bytes); with separate log pointers the call
 * returns DYB_ERR_UNSUPPORTED rather than alias replicas.  dyb_stepper_adapt_frames (replicas > 1) takes no side stream
 * (DYB_ERR_UNSUPPORTED: the owed final inference would read restaged inputs).  dyb_stepper_set_active: the replicas following
 * frame steps cover (ascending physical indices; n = 0: all) - sequences of
 * different lengths: a replica whose stream has ended leaves the set, its weights / Adam state / records stay as they are. */
int dyb_stepper_adapt_frames_full(void* stepper, const void* const* inputs, int record_slot, int loss_slot, int* extra_steps,
                                  dyb_stream_t stream, dyb_stream_t aux);
int dyb_stepper_set_active(void* stepper, const int* idx, int n);
int dyb_stepper_join(void* stepper, dyb_stream_t stream);
const float* dyb_stepper_output(const void* stepper, int which);

/* HMR in train() mode: nn.Dropout(p) after fc1 / fc2 of every regressor iteration (reference model/hmr.py:84,86,165,169 -
 * the reference's mean teacher runs like this, base_adaptor.py:151-158 never calls teacher.eval()).  Masks are
 * counter-based (Philox-4x32-10) functions of (seed, offset, iteration, element): a backward regenerates them from the same
 * pair.  dyb_hmr_feature_info_ex(train = 1) locates the post-dropout hidden vectors (features 7+3t). */
int dyb_hmr_forward_train(void* plan, const float* params, const float* image_nchw, const float* init_state, int n_iter,
                          float* acts, void* ws, size_t ws_bytes, unsigned long long seed, unsigned long long offset, float p,
                          dyb_stream_t stream);
int dyb_hmr_backward_train(void* plan, const float* params, const float* acts, const float* d_rotmat, const float* d_state,
                           int n_iter, float* grads, void* ws, size_t ws_bytes, unsigned long long seed,
                           unsigned long long offset, float p, dyb_stream_t stream, dyb_stream_t aux_stream);
int dyb_hmr_feature_info_ex(const void* plan, int which, int train, long long* offset, int* dims4, int* row_stride);

#ifdef __cplusplus
}
#endif
#endif
