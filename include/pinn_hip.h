/* C-ABI of the MI355X-native PINN-elastodynamics hot path (libpinn_hip.so).
 *
 * The reference (Raocp/PINN-elastodynamics) has no FFI: its "operator interface" for this path is
 * the method surface of the TF1 model class.  Each entry point below names the reference graph
 * piece it replaces (file:line relative to the reference repository, INF =
 * ElasticWaveInfinite/ElasticWave.py).  INTEGRATION.md shows the ctypes binding a maintainer of
 * the reference would add.
 *
 * Conventions
 *   - every pointer except `layers`, `lb`, `ub`, `term_weights`, `out_weights` is a DEVICE pointer
 *     owned by the caller; the library allocates nothing and keeps no state between calls;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); calls return immediately;
 *   - return value: 0 on success, a negative PINN_ERR_* code for bad arguments, a positive
 *     hipError_t value if a launch failed.  Nothing throws.
 *   - flat parameter vector: W0, b0, W1, b1, ... with each W row-major [in, out] and each b of
 *     length out -- the arrays of the reference checkpoint [W_list, b_list] (INF:159-165)
 *     concatenated layer by layer;
 *   - `layers` = {3, H, ..., H, n_out} exactly like the reference's uv_layers (INF:645).
 */
#ifndef PINN_HIP_H
#define PINN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* precision_mode: operand type of the matrix pipe (accumulation is always fp32) */
enum {
    PINN_PREC_BF16 = 0,   /* bf16 operands, one MFMA per product (BASELINE config "bf16-MFMA/fp32-accum") */
    PINN_PREC_F16X3 = 1,  /* fp16 hi + scaled-lo split, three MFMAs per product: fp32-class accuracy.  Limits of the class (DESIGN
                             section 7): (1) the hi + lo weights carry ~23 significant bits -- their split error is 1.75x the fp32
                             rounding of the same weights and, like it, the same for every point: where a residual is a > 1000-fold
                             cancellation of O(1) outputs (the plate's hole traction at trained weights) the gradient error does
                             not fall with the number of points as host fp32's does; (2) the weight gradient of padded widths <= 64
                             takes the layer states as fp16 high parts, and in the four- and five-stream collocation kernels the adjoints too (round 4: one
                             MFMA per product; forward and reverse chain keep hi + lo): a random rounding noise of 3.5e-4 / sqrt(points)
                             relative to the gradient (1e-5 at 4096 points, 2.5e-7 at 2 M) that stays inside host-fp32's own per-layer error at
                             the reference's trained weights -- the wider layouts use both parts of both;
                             (3) |w| <= 2047 in the
                             fused kernels' weight format (32 w must stay finite in fp16).  A larger weight is DETECTED when the weights
                             are packed: the call then returns NaN in every gradient entry and every loss sum (never a plausible wrong
                             number) -- pass PINN_FLAG_TWO_KERNEL to evaluate such weights (the two-kernel path holds |w| up to 65504);
                             pinn_fused_weight_limit() returns the bound;  (4) the collocation kernels of padded width 64 (wave and plate
                             heads, 8 or 4 hidden layers) hand the per-layer states to their activation reverse as fp16 high part + the TOP BYTE
                             of the fp16 low part (14 significant bits; round 4): measured at the reference's trained nets the gradient blocks
                             stay where full low parts put them (tools/studies/wgrad_operand_study.py, the per-layer fp32 bounds of the GPU
                             tests); since round 5 the lowest parked state of the four-stream kernel (S_2) travels without that byte -- which layers'
                             low parts the reverse needs was priced layer by layer (tools/studies/lo_policy_study.py: the upper layers carry it; at
                             the most sensitive reference net the worst per-layer multiple of fp32's own error moves 3.9 -> 4.6 of the tests' bound
                             of 6, elsewhere not at all); PINN_FLAG_STATE_FP16 drops the low parts altogether and is NOT parity-grade. */
    PINN_PREC_F16 = 2,    /* fp16 operands, one MFMA per product */
    PINN_PREC_BF16X3 = 3, /* bf16 hi/lo split, three MFMAs per product (16-bit significand) */
    PINN_PREC_FP32 = 4,   /* plain fp32 FMA arithmetic, no matrix pipe (the reference's own precision, INF:71-92): ~100x slower,
                             for parity checks; offered by every entry point (wave, data, fields, the plate family with its second
                             time derivative, the 4-input heads) */
    /* OR this into precision_mode when the PREVIOUS call on the same workspace used the same params_flat contents, layers and
     * mode: the packed MFMA weight fragments are still in the workspace and the repack launch is skipped (the reference feeds
     * one set of variables to every loss term of a step, INF:297-305; a step makes 3-4 calls). */
    PINN_FLAG_WEIGHTS_PACKED = 0x100,
    /* OR this into precision_mode to let the fused kernel of the 8-layer, width <= 64 collocation path park its per-layer states as fp16
     * high parts only (no low parts): 17 % faster, but the activation reverse then sees states rounded to 2^-12, which cancellation at
     * TRAINED weights amplifies (first-layer gradient blocks 5e-3 off at the reference's trained nets, fp32 itself 2e-4).  Fine far
     * from an optimum (early Adam steps); not parity-grade.  Ignored where it does not apply. */
    PINN_FLAG_STATE_FP16 = 0x200,
    /* OR this into precision_mode to keep this call off the fused persistent kernel: forward / reverse chain and weight gradient run as
     * the two-kernel path (state and adjoint panels through the workspace): slower, no |w| <= 2047 bound (see PINN_PREC_F16X3). */
    PINN_FLAG_TWO_KERNEL = 0x400
    /* PINN_ADJOINT_SHIFT(k), k = 0..24, may be OR'ed in as well: see below */
};

enum {
    PINN_OK = 0,
    PINN_ERR_NULL = -1,        /* a required pointer is NULL */
    PINN_ERR_LAYERS = -2,      /* unsupported layer list (see pinn_supported_width) */
    PINN_ERR_PRECISION = -3,   /* unknown precision_mode */
    PINN_ERR_WORKSPACE = -4,   /* workspace smaller than pinn_min_workspace_bytes() or misaligned */
    PINN_ERR_SIZE = -5,        /* n < 0 (n == 0 is a valid empty batch: zero sums, zero / untouched gradient) */
    PINN_ERR_COLLECTIVE = -6,  /* pinn_p2p_*: not connected; a coarse-grained buffer across devices (pinn_p2p_connect); or a rank did not arrive within the
                                * bounded wait of some call (pinn_p2p_set_timeout_ms, default 30 s) -- that call's buffer is then NaN on this rank */
    PINN_ERR_RANGE = -7        /* pinn_wave2d_loss_grad_checked: gradient non-finite even on the two-kernel path with the reverse pass scaled by 2^-24 */
};

/* Padded hidden width the kernels use for a real hidden width h (0 if unsupported). */
int pinn_supported_width(int h);
/* Largest |w| the fused kernels' weight format holds (2047); see PINN_PREC_F16X3 (3) and PINN_FLAG_TWO_KERNEL. */
float pinn_fused_weight_limit(void);

/* Which kernel path a loss + gradient call takes (round 5: no silent slow path).  `head` names the entry point family, `ws_bytes` is the
 * workspace the calls will be given (0: assume pinn_workspace_bytes() of a large set).  The answer is the rule the calls themselves apply
 * (same code), so a caller -- the model classes warn once -- can tell BEFORE training that a net lands on the two-kernel path, which is
 * 2.3-5x slower than the fused persistent kernel (profiles/r04_conf_time.txt): the fused kernel is compiled for the depths the reference
 * uses (padded width <= 64: 4 or 8 hidden layers; 96 / 128: 8; 160: 6; the 3-D net: 10 x 128). */
enum {
    PINN_HEAD_WAVE = 0,        /* pinn_wave2d_loss_grad: four streams */
    PINN_HEAD_DATA = 1,        /* pinn_data_loss_grad(_multi), pinn_plate2d_traction_loss_grad: one stream */
    PINN_HEAD_PLATE = 2,       /* pinn_plate2d_loss_grad: five streams (second time derivative) */
    PINN_HEAD_NC3D = 3,        /* pinn_nc3d_loss_grad: four inputs, five first-order streams */
    PINN_HEAD_NC3D_DATA = 4,   /* pinn_nc3d_data_loss_grad */
    PINN_HEAD_STREAMS = 5      /* pinn_stream_loss_grad (the plate's pre-training losses) */
};
enum {
    PINN_PATH_FUSED_REGISTERS = 1,   /* fused persistent kernel, tile state in registers (padded width <= 64) */
    PINN_PATH_FUSED_LDS = 2,         /* fused persistent kernel, LDS-operand layout (padded widths 96 / 128 / 160) */
    PINN_PATH_TWO_KERNEL = 3,        /* chain_kernel + wgrad_kernel with state / adjoint panels through the workspace */
    PINN_PATH_FP32 = 4               /* PINN_PREC_FP32: the checker mode, no matrix pipe */
};
/* returns a PINN_PATH_* value, or a negative PINN_ERR_* code (unsupported layer list / precision mode) */
int pinn_path_for(const int* layers, int n_layers, int precision_mode, int head, size_t ws_bytes);
/* Process-wide counters of the paths the loss + gradient calls actually took since the last reset: counts[PINN_PATH_*] (index 0 unused).
 * `reset` != 0 zeroes them after the read.  Tests assert with it that a case ran where it was meant to run. */
void pinn_debug_path_counts(int64_t counts[5], int reset);

/* Workspace sizing.  `recommended` holds all tiles of an n-point call in one pass; anything
 * >= `min` works (the call then walks the points in several chunks).  The fused persistent kernel (one launch per call, what the
 * published numbers are measured with) needs one scratch image per workgroup on top of the fixed part -- pinn_workspace_bytes() of a
 * few hundred thousand points covers its 256 workgroups for every net --; a workspace that holds fewer than 64 of them makes the call
 * take the two-kernel path instead (same results, slower), it never runs a persistent launch on a handful of CUs. */
size_t pinn_workspace_bytes(const int* layers, int n_layers, int64_t n, int precision_mode);
size_t pinn_min_workspace_bytes(const int* layers, int n_layers, int precision_mode);

/* Replaces net_f_sig + the seven mean-squares + their gradient: INF:221-265 (SEMI:228-272,
 * CONF:304-348), INF:104-110, and d/d(W,b) as built by optimizer_Adam.minimize INF:131-133.
 *   loss_terms_out[i] = sum_n f_i(n)^2, i in (f_u,f_v,f_ut,f_vt,f_s11,f_s22,f_s12)   (INF:265 order)
 *   grad_flat_out (+)= d/dparams sum_i term_weights[i] * loss_terms[i]
 * The caller folds the loss layout (INF:119 / SEMI:127 / CONF:156) and reduce_mean's 1/N into
 * term_weights.  normalize != 0 applies INF:191's input map with lb/ub. */
int pinn_wave2d_loss_grad(const float* params_flat, const int* layers, int n_layers,
                          const float* x, const float* y, const float* t, int64_t n,
                          const double lb[3], const double ub[3], int normalize,
                          double E, double mu, double rho, int plane_strain,
                          const float term_weights[7],
                          float* loss_terms_out, float* grad_flat_out, int accumulate,
                          int precision_mode, void* workspace, size_t ws_bytes, void* stream);

/* Same call, but brackets every kernel with HIP events on `stream` and returns, in the HOST array
 * kernel_ms, the summed durations of {weight repack, chain kernel (forward + reverse chain),
 * weight-gradient kernel, small reductions}.  Synchronous; used by bench.py's roofline leg only. */
int pinn_wave2d_loss_grad_profile(const float* params_flat, const int* layers, int n_layers,
                                  const float* x, const float* y, const float* t, int64_t n,
                                  const double lb[3], const double ub[3], int normalize,
                                  double E, double mu, double rho, int plane_strain,
                                  const float term_weights[7],
                                  float* loss_terms_out, float* grad_flat_out, int accumulate,
                                  int precision_mode, void* workspace, size_t ws_bytes, void* stream,
                                  float kernel_ms[4]);

/* Replaces the value-only terms on the small side sets and their gradient: loss_IC INF:111-114,
 * loss_SRC INF:115-116, loss_NB INF:117-118 (SEMI:125-126), loss_FIX CONF:145-146.
 *   loss_terms_out[o] = sum_n (Y_o(n) - targets[o][n])^2      (targets == NULL means 0)
 *   grad_flat_out (+)= d/dparams sum_o out_weights[o] * loss_terms[o]
 * targets is SoA [n_out][n]. */
int pinn_data_loss_grad(const float* params_flat, const int* layers, int n_layers,
                        const float* x, const float* y, const float* t, int64_t n,
                        const double lb[3], const double ub[3], int normalize,
                        const float* targets, const float* out_weights,
                        float* loss_terms_out, float* grad_flat_out, int accumulate,
                        int precision_mode, void* workspace, size_t ws_bytes, void* stream);

/* Replaces net_uv + the Jacobian pieces net_e needs (INF:201-219), i.e. what predict/probe
 * evaluate (INF:337-359):  fields_out is SoA [4*n_out][n] = Y, dY/dx, dY/dy, dY/dt. */
int pinn_wave2d_fields(const float* params_flat, const int* layers, int n_layers,
                       const float* x, const float* y, const float* t, int64_t n,
                       const double lb[3], const double ub[3], int normalize,
                       float* fields_out, int precision_mode, void* workspace, size_t ws_bytes, void* stream);

/* ---- plate-with-hole family (PLATE = PlateHoleQuarter/train/train.py); precision_mode must be a split mode ---------------
 * Streams of one net at raw or normalised (x,y,t): streams_out is SoA [5][n_out][n] = Y, dY/dx, dY/dy, dY/dt, d2Y/dt2.
 * Replaces the tf.gradients calls of net_dist_dt / net_part / net_vel (PLATE:330-356,398-402) and provides the frozen
 * distance / particular streams the composite needs. */
int pinn_net_streams(const float* params_flat, const int* layers, int n_layers,
                     const float* x, const float* y, const float* t, int64_t n,
                     const double lb[3], const double ub[3], int normalize,
                     float* streams_out, int precision_mode, void* workspace, size_t ws_bytes, void* stream);

/* Replaces net_f_sig of the plate (composite u = P + D*N, PLATE:358-388; nested u_tt, plane stress, PLATE:404-439), the
 * five mean-squares PLATE:187-191 and their gradient w.r.t. the uv net only (D, P frozen, PLATE:240-241,249-250).
 *   frozen_streams: SoA [2][5][5][n] = streams (value,x,y,t,tt) x fields (u,v,s11,s22,s12) of D, then of P (pinn_net_streams)
 *   loss_terms_out[i] = sum_n f_i(n)^2, i in (f_u, f_v, f_s11, f_s22, f_s12)   (PLATE:439 order) */
int pinn_plate2d_loss_grad(const float* params_flat, const int* layers, int n_layers,
                           const float* x, const float* y, const float* t, int64_t n,
                           const double lb[3], const double ub[3], int normalize,
                           const float* frozen_streams, double E, double mu, double rho,
                           const float term_weights[5],
                           float* loss_terms_out, float* grad_flat_out, int accumulate,
                           int precision_mode, void* workspace, size_t ws_bytes, void* stream);

/* Replaces net_t + loss_HOLE (PLATE:452-461,192-193) and its gradient w.r.t. the uv net.
 *   frozen_and_normals: SoA [12][n] = D values (5 fields), P values (5 fields), nx, ny at the hole points
 *   loss_terms_out = (sum tx^2, sum ty^2) */
int pinn_plate2d_traction_loss_grad(const float* params_flat, const int* layers, int n_layers,
                                    const float* x, const float* y, const float* t, int64_t n,
                                    const double lb[3], const double ub[3], int normalize,
                                    const float* frozen_and_normals, const float weights[2],
                                    float* loss_terms_out, float* grad_flat_out, int accumulate,
                                    int precision_mode, void* workspace, size_t ws_bytes, void* stream);

/* Replaces the pre-training losses loss_DIST / loss_PART (PLATE:194-215) and their gradients: a weighted sum of squares
 * over streams and outputs of ONE net,  sum_{s,o} weights[s*n_out+o] * sum_n (Y[s][o][n] - targets[s][o][n])^2
 * (targets SoA [5][n_out][n] or NULL = 0; weights is a host array of 5*n_out floats).
 *   loss_terms_out[o] = sum_s (weights[s][o]/max|weights|) * sum_n (...)^2 */
int pinn_stream_loss_grad(const float* params_flat, const int* layers, int n_layers,
                          const float* x, const float* y, const float* t, int64_t n,
                          const double lb[3], const double ub[3], int normalize,
                          const float* targets, const float* weights,
                          float* loss_terms_out, float* grad_flat_out, int accumulate,
                          int precision_mode, void* workspace, size_t ws_bytes, void* stream);

/* The 16-bit operand modes run the reverse pass on adjoints  2 w_i f_i / max|w|  (fp16: |x| < 65504).  A residual that is
 * orders of magnitude above its trained size -- the first trial points of an L-BFGS line search far from the optimum, a stiff
 * material -- can overflow that range; the sums of squares stay finite but the gradient comes back non-finite.  OR
 * PINN_ADJOINT_SHIFT(k) into precision_mode to run the reverse pass on adjoints scaled by 2^-k (the gradient is scaled back
 * by 2^k in the reduction, so the result is the same number); k > 0 costs accuracy only for adjoint elements that drop below
 * fp16's normal range.  The host classes raise k when a gradient comes back non-finite and lower it again as the loss falls.
 * (pinn_stream_loss_grad, whose reported sums carry the normalised weights, ignores the shift.) */
#define PINN_ADJOINT_SHIFT(k) (((k) & 0x1f) << 16)

/* ---- The finite-gradient ladder as library calls (round 6): what the host classes do around every evaluation whose result they look at
 * (pinn_elastodynamics_amd/elastic_wave.py: evaluate_with_finite_gradient / leave_fused_path_if_weights_out_of_range), for callers of this ABI
 * that are not Python.  Two ranges bound the 16-bit modes: |w| <= pinn_fused_weight_limit() in the fused kernels' weight format (beyond it a
 * call returns NaN throughout: sums AND gradient) and the fp16 range of the reverse pass's adjoints (beyond it the gradient alone is non-finite).
 *   pinn_probe_ranges              one small reduction + a stream synchronisation: is grad_flat[0..n_params) finite, and max |params_flat[i]|
 *                                  (either pointer may be NULL).  Uses 16 bytes at the end of `workspace` (dead data between calls).
 *   pinn_wave2d_loss_grad_checked  pinn_wave2d_loss_grad (overwrite form) + the ladder, SYNCHRONOUS: evaluates with the flags in `state`, probes, and
 *                                  on a non-finite gradient either sets state->two_kernel (a weight beyond the fused format: the calls leave the
 *                                  fused path for good) or raises state->adjoint_shift by 4 (up to 24) and repeats.  PINN_OK: loss sums and gradient
 *                                  are finite and final; PINN_ERR_RANGE: the ladder is exhausted.  The caller keeps `state` between calls (start
 *                                  from zeros) and may lower adjoint_shift again as its loss falls (the host classes do at 1/256 of the loss the
 *                                  shift was raised at).  The TWO_KERNEL flag and the shift bits of precision_mode are ignored: `state` owns them.
 * The other families take the same ladder around their own call: probe, then PINN_FLAG_TWO_KERNEL or PINN_ADJOINT_SHIFT(k + 4). */
typedef struct pinn_range_state {
    int adjoint_shift;   /* in/out: 0..24 */
    int two_kernel;      /* in/out: 0 / 1 */
    int attempts;        /* out: evaluations the last checked call made */
    int reserved;
} pinn_range_state;
int pinn_probe_ranges(const float* params_flat, const float* grad_flat, int64_t n_params, void* workspace, size_t ws_bytes, void* stream,
                      int* grad_finite_out, float* max_abs_weight_out);
int pinn_wave2d_loss_grad_checked(const float* params_flat, const int* layers, int n_layers,
                                  const float* x, const float* y, const float* t, int64_t n,
                                  const double lb[3], const double ub[3], int normalize,
                                  double E, double mu, double rho, int plane_strain,
                                  const float term_weights[7],
                                  float* loss_terms_out, float* grad_flat_out,
                                  int precision_mode, void* workspace, size_t ws_bytes, void* stream, pinn_range_state* state);

/* Several value-only sets in ONE call (the reference evaluates loss_IC, loss_SRC, loss_NB / loss_FIX of a step from one set of
 * variables, INF:111-119,297-305): same arithmetic as pinn_data_loss_grad per set, gradients summed, sums of set k written to
 * sets[k].loss_terms_out[0..n_out).  For nets the fused kernel covers this is one launch instead of n_sets; otherwise the sets
 * are processed one after the other.  n_sets <= PINN_MAX_SETS; empty sets (n == 0) are skipped and report zeros. */
#define PINN_MAX_SETS 4
typedef struct {
    const float* x;
    const float* y;
    const float* t;
    int64_t n;
    const float* targets;      /* SoA [n_out][n] or NULL (= 0) */
    float out_weights[8];
    float* loss_terms_out;     /* device, >= n_out floats */
} pinn_point_set;
int pinn_data_loss_grad_multi(const float* params_flat, const int* layers, int n_layers,
                              const pinn_point_set* sets, int n_sets,
                              const double lb[3], const double ub[3], int normalize,
                              float* grad_flat_out, int accumulate,
                              int precision_mode, void* workspace, size_t ws_bytes, void* stream);

/* One call per rank per training step (round 5): what the step of INF:282-319 evaluates -- net_f_sig on the collocation
 * batch (pinn_wave2d_loss_grad's arguments), the value-only side sets (pinn_data_loss_grad_multi's), the gradient of their weighted sum
 * and, with `adam`, the optimizer update of INF:131-133 -- as
 *     weight repack | ONE persistent launch for all point sets | ONE reduction (+ Adam)
 * for the nets whose calls take the register-state fused kernel (padded width <= 64, 4 or 8 hidden layers: pinn_path_for).  The side sets'
 * workgroups are dispatched behind the collocation set's and start on the compute units that run out of collocation steps first, so a set of
 * N points no longer leaves most of the chip idle during its last of ceil(N / 64 / 256) steps (the 250,000-point share of one of 8 GPUs: 67
 * workgroups have a 16th step, 189 do not).  Every other case -- other widths / depths, PINN_PREC_FP32, no side points, an empty batch, a
 * workspace without room for both parts' scratch images -- makes the three calls one after the other inside: the RESULTS are the same bits
 * either way (same partial sums, same order of the final additions, same Adam expression).
 *   adam == NULL: the gradient is left in grad_flat_out (data-parallel ranks all-reduce it, then call pinn_adam_step);
 *   adam != NULL: params_flat, adam->m, adam->v are updated in place behind the reduction (and grad_flat_out still holds the gradient).
 * n_sets may be 0.  Loss sums as in the two calls it replaces. */
typedef struct {
    float* m;                  /* first / second moment, length n_params, updated in place */
    float* v;
    double lr, beta1, beta2, eps;
    int64_t step;              /* 1-based */
} pinn_adam_state;
int pinn_wave2d_step(float* params_flat, const int* layers, int n_layers,
                     const float* x, const float* y, const float* t, int64_t n,
                     const double lb[3], const double ub[3], int normalize,
                     double E, double mu, double rho, int plane_strain,
                     const float term_weights[7], float* loss_terms_out,
                     const pinn_point_set* sets, int n_sets,
                     float* grad_flat_out, int accumulate, const pinn_adam_state* adam,
                     int precision_mode, void* workspace, size_t ws_bytes, void* stream);

/* The plate's step the same way (PLATE:187-217: loss = 10 (loss_f_uv + loss_f_s + loss_HOLE)): pinn_plate2d_loss_grad's collocation set and
 * pinn_plate2d_traction_loss_grad's hole set in one persistent launch (five-stream part, then the one-stream part), one reduction (+ Adam).
 * What it saves matters most where the reference spends its time: the L-BFGS stage on ~45 k points is latency-bound (a handful of steps per
 * workgroup), and every evaluation is two launches less.  Same fall-back rule and the same bits as the separate calls. */
int pinn_plate2d_step(float* params_flat, const int* layers, int n_layers,
                      const float* x, const float* y, const float* t, int64_t n,
                      const double lb[3], const double ub[3], int normalize,
                      const float* frozen_streams, double E, double mu, double rho,
                      const float term_weights[5], float* loss_terms_out,
                      const float* hole_x, const float* hole_y, const float* hole_t, int64_t hole_n,
                      const float* hole_frozen_and_normals, const float hole_weights[2], float* hole_loss_terms_out,
                      float* grad_flat_out, int accumulate, const pinn_adam_state* adam,
                      int precision_mode, void* workspace, size_t ws_bytes, void* stream);

/* ---- 3-D Navier-Cauchy extension (BASELINE.json configs[4]).  NOT in the reference: all four reference scripts are 2-D + time
 * (SURVEY.md section 0), so these entry points have no reference lines to replace; they state the 3-D form of net_f_sig
 * (INF:221-265) the way oracle/nc3d_oracle.py spells it out, and parity for them is unpinned by definition.
 *   layers = {4, H x k, 12}; inputs (x, y, z, t); outputs (u,v,w, ut,vt,wt, s11,s22,s33, s12,s13,s23);
 *   residuals (f_u,f_v,f_w, f_ut,f_vt,f_wt, f_s11,f_s22,f_s33, f_s12,f_s13,f_s23); isotropic Hooke law, E, mu, rho as INF:33-35.
 *   loss_terms_out[i] = sum_n f_i(n)^2 (12 values), grad_flat_out (+)= d/dparams sum_i term_weights[i] * loss_terms[i].
 * Split-precision modes only (PINN_PREC_F16X3 / PINN_PREC_BF16X3).  Workspace sizing: the same two functions (layers[0] == 4). */
int pinn_nc3d_loss_grad(const float* params_flat, const int* layers, int n_layers,
                        const float* x, const float* y, const float* z, const float* t, int64_t n,
                        const double lb[4], const double ub[4], int normalize,
                        double E, double mu, double rho, const float term_weights[12],
                        float* loss_terms_out, float* grad_flat_out, int accumulate,
                        int precision_mode, void* workspace, size_t ws_bytes, void* stream);

/* Value-only terms of the 4-input net (initial state, sources, the traction-free surface s33 = s13 = s23 = 0 of the half space):
 * pinn_data_loss_grad with four coordinates; targets SoA [n_out][n] or NULL, out_weights[n_out], loss_terms_out[n_out].
 * Round 6: the BASELINE configs[4] net (10 x 128, 12 outputs) takes the fused kernel's one-stream instantiation here as well
 * (pinn_path_for(.., PINN_HEAD_NC3D_DATA) = PINN_PATH_FUSED_LDS), every other layer list the two-kernel path. */
int pinn_nc3d_data_loss_grad(const float* params_flat, const int* layers, int n_layers,
                             const float* x, const float* y, const float* z, const float* t, int64_t n,
                             const double lb[4], const double ub[4], int normalize,
                             const float* targets, const float* out_weights,
                             float* loss_terms_out, float* grad_flat_out, int accumulate,
                             int precision_mode, void* workspace, size_t ws_bytes, void* stream);

/* Forward only: fields_out [5][n_out][n] = the outputs and their derivatives w.r.t. x, y, z, t (predict of the 3-D case). */
int pinn_nc3d_fields(const float* params_flat, const int* layers, int n_layers,
                     const float* x, const float* y, const float* z, const float* t, int64_t n,
                     const double lb[4], const double ub[4], int normalize,
                     float* fields_out, int precision_mode, void* workspace, size_t ws_bytes, void* stream);

/* Replaces tf.train.AdamOptimizer's update (INF:131-133; TF1 rule: epsilon outside the bias
 * correction).  step is 1-based.  All arrays are length n_params, updated in place. */
int pinn_adam_step(float* params_flat, float* m, float* v, const float* grad_flat, int64_t n_params,
                   double lr, double beta1, double beta2, double eps, int64_t step, void* stream);

/* ---- Latency-floor all-reduce of the step's buffer [gradient | loss sums] (round 5; SURVEY sections 5 and 8e).  The message is ~119 KB:
 * latency-bound, so a ring or tree buys nothing.  One-shot: every rank WRITES its buffer into a slot of every peer's IPC-mapped receive buffer
 * over its point-to-point xGMI link, flags it, and every rank sums the world's slots locally in rank order (the same bits on every rank) --
 * with the TF1 Adam update of INF:131-133 in the same kernel.  One launch replaces all-reduce + pinn_adam_step and does not depend on RCCL's
 * small-message protocol (RCCL stays the default collective of the model classes; this is `collective="p2p"`).
 *   pinn_p2p_create   allocates this rank's receive buffer (2 parities x world slots x max_floats, fine-grained device memory -- the ONE
 *                     allocation this library makes, owned by the comm object) and returns its IPC handle;
 *   the caller exchanges the world's handles (any out-of-band channel: the model classes use torch.distributed.all_gather_object);
 *   pinn_p2p_connect  opens the peers' buffers;   pinn_p2p_allreduce  buf[0..n) <- sum over ranks, enqueued on `stream`; with `adam` the
 *                     first n_params entries of the sum also update params_flat / adam->m / adam->v;   every rank must make the same calls.
 *   pinn_p2p_status   synchronises the device and returns 0, or PINN_ERR_COLLECTIVE if some call failed;   pinn_p2p_peek_status reads the same
 *                     word (pinned host memory, written by the kernel) WITHOUT synchronising: meaningful behind a synchronisation the caller
 *                     made anyway -- the model classes read it at every host sync point of train() / train_bfgs() and raise.
 * Failure semantics (round 6):
 *   - a call succeeds or fails AS A WHOLE on a rank (grid agreement of its 64 workgroups): never a half-applied sum or Adam update;
 *   - a failed call leaves buf[0..n) = NaN on that rank (a caller that only looks at its loss / gradient still notices), skips Adam, sets the
 *     status word, and writes its call number into every peer's abort word: a peer that arrives later -- or had completed the call already --
 *     fails its current or next call at once.  Failure is sticky: every later call on the comm fails immediately;
 *   - the wait is bounded by the device's constant-rate wall clock: pinn_p2p_set_timeout_ms (default 30 000 ms, or the environment variable
 *     PINN_P2P_TIMEOUT_MS at create time): longer than legitimate skew between ranks (first-step lazy initialisation, a rank writing a
 *     checkpoint), short enough that a dead rank ends the run instead of hanging the GPU;
 *   - pinn_p2p_connect returns PINN_ERR_COLLECTIVE, before mapping anything, if this rank's or a peer's buffer is coarse-grained
 *     (hipExtMallocWithFlags(hipDeviceMallocFinegrained) refused) AND the two live on different physical devices (PCI bus ids travel with the
 *     handles): coarse-grained memory is coherent at kernel boundaries only.  Ranks that share one GPU (the tests) may use it.
 *   - calls of one comm are made on ONE stream (the grid agreement counts workgroups per call number);
 *   - pushes are 16-byte stores when buf is 16-byte aligned and n % 4 == 0 (the model classes' buffers are), 4-byte stores otherwise.
 * Single-node use across physical GPUs is UNVERIFIED on hardware (the builder's boxes have one GPU: world 2 on one device is what the tests run;
 * tests/test_gpu_dp.py::test_p2p_across_two_devices runs where torch.cuda.device_count() >= 2). */
#define PINN_IPC_HANDLE_BYTES 128      /* hipIpcMemHandle_t (64) + memory kind + PCI bus id of the owning device */
typedef struct pinn_p2p_comm pinn_p2p_comm;
int pinn_p2p_create(int rank, int world, int64_t max_floats, pinn_p2p_comm** comm_out, unsigned char handle_out[PINN_IPC_HANDLE_BYTES]);
int pinn_p2p_connect(pinn_p2p_comm* comm, const unsigned char* all_handles /* world x PINN_IPC_HANDLE_BYTES, rank order */);
int pinn_p2p_set_timeout_ms(pinn_p2p_comm* comm, double timeout_ms);
int pinn_p2p_allreduce(pinn_p2p_comm* comm, float* buf, int64_t n, float* params_flat, const pinn_adam_state* adam, int64_t n_params, void* stream);
int pinn_p2p_status(pinn_p2p_comm* comm, int* fine_grained_out);
int pinn_p2p_peek_status(pinn_p2p_comm* comm);
int pinn_p2p_destroy(pinn_p2p_comm* comm);
/* Testing hook (process-wide): 1 makes pinn_p2p_create allocate plain (coarse-grained) device memory, as if the runtime had refused the
 * fine-grained flag; returns the previous setting.  For the refusal test of pinn_p2p_connect. */
int pinn_p2p_debug_force_coarse(int enable);

/* Testing hook (process-wide): 0 forces the two-kernel path (chain + wgrad kernels) even where the
 * fused kernel applies; 1 (default) prefers the fused kernel.  Returns the previous setting.  Both paths
 * compute the same numbers to the precision mode's accuracy (tests/test_gpu_parity.py). */
int pinn_debug_set_fused(int enable);
/* Tuning hook (process-wide): the XCD-aware step assignment of the fused collocation kernels.  Workgroups are dispatched round-robin over the 8
 * XCDs and the odd XCDs of an MI355X run these kernels 2-3 % slower than the even ones (profiles/r05_workgroup_lifetimes.txt), so the even-XCD
 * workgroups take `permille` / 1000 more steps (as the tail of the launch; default 16).  0 turns it off; returns the previous setting.  The
 * assignment is static: results stay a deterministic function of the inputs and the launch shape. */
int pinn_debug_set_xcd_bonus(int permille);
/* Testing hook (process-wide): at most `cap` workgroups in a fused per-family launch (0: the default, one per compute unit) -- several steps per
 * workgroup on small test inputs.  Returns the previous setting. */
int pinn_debug_set_fused_grid_cap(int cap);
/* Query (round 6): the per-workgroup memory of the fused collocation kernel a layer list takes (f16x3; head = PINN_HEAD_WAVE / _PLATE / _NC3D) and the
 * cache policy compiled into it.  A persistent workgroup owns `*images_bytes` of parked state images and `*sums_bytes` of running weight-gradient
 * sums; a full grid is 256 of them.  Where that exceeds the 256 MB Infinity Cache the kernel marks ONE of the two classes non-temporal so that the other
 * stays resident (profiles/r06_footprint_and_cache_policy.txt).  Returns 0 (no hint), 1 (the sums), 2 (the images), or
 * PINN_ERR_LAYERS for a layer list without fused kernel.  Host-only, no device call. */
int pinn_debug_cache_policy(const int* layers, int n_layers, int head, size_t* images_bytes, size_t* sums_bytes);
/* Profiling hook (process-wide): device buffer of 128 uint64 that the fused kernel fills with shader-clock
 * stamps of its phases (workgroup 0 only); NULL turns it off. */
void pinn_debug_set_stamp_buffer(void* device_u64x128);
/* Rate in kHz of the device wall clock the launch-long stamps use (slots 120 / 121 of the stamp buffer: wall-clock ticks, 124 / 125: shader cycles of
 * workgroup 0 around all of its steps -- a launch's duration and the clock it ran at, measured by the kernel itself); 0 if unknown. */
int pinn_debug_wall_clock_khz(void);
/* Profiling hook (process-wide): host array of 4 floats; while set, every loss+gradient call brackets its kernels with HIP
 * events on the call's stream, synchronises, and writes milliseconds {repack, chain (or the whole fused kernel), weight gradient,
 * reductions} of that call (what pinn_wave2d_loss_grad_profile does for one entry point, here for all of them: bench.py's
 * roofline of the plate / 3-D kernels).  NULL turns it off. */
void pinn_debug_set_profile_buffer(float* host_ms4);
/* Profiling hook (process-wide), asynchronous: _arm(k) starts recording the next k (<= 4096) launches of the fused kernel: each is
 * bracketed by HIP events on its call's stream and nothing synchronises, so a running step loop keeps its back-to-back stream order
 * (bench.py's roofline leg: the launch time under the same conditions as the timed steps).  _read() stops the recording, waits for the
 * recorded launches, writes their milliseconds and stream counts (4 / 5: a collocation set, 1: the value-only side sets) in launch
 * order and returns how many there were.  Either output may be NULL. */
int pinn_debug_profile_ring_arm(int max_launches);
/* ... of every `every`-th step only (a step = one collocation launch and the side-set launches behind it; default 1): the events of a
 * bracketed launch put barrier packets between back-to-back kernels, which slowed a fully bracketed block by 3.8 % (round 4); at a stride
 * of 8 the block is the timed loop again to 0.5 %.  Applies to the next _arm(). */
void pinn_debug_profile_ring_stride(int every);
int pinn_debug_profile_ring_read(float* ms_out, int* streams_out, int capacity);

const char* pinn_error_string(int code);
int pinn_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PINN_HIP_H */
