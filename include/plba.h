/*
 * plba.h — C ABI of the MI355X-native local bundle adjustment ("plba") hot path.
 *
 * This is the drop-in boundary between host C++ (the g2o-compatible facade used by
 * MapHandler::localBundleAdjustmentWithImu / ...WithImuAndMarg, reference
 * src/mapHandler.cpp:5086-5739 and :5741-6254) and the hand-written HIP kernels.
 * Plain C linkage, plain pointers and sizes, no C++ or torch types.
 *
 * Conventions
 *  - every function returns 0 (PLBA_OK) or a negative plba_status; text via plba_last_error().
 *  - the caller owns every host buffer it passes in (copied during the call) and every
 *    output buffer it supplies; the library owns all device memory.
 *  - thread-compatible: one thread per plba_problem.
 *  - all reals are IEEE double (the reference path is all-double); ids are int32.
 *  - matrices are row-major unless a name says colmajor.
 *  - keyframes are addressed by their index k in [0,K) in the arrays of plba_set_keyframes
 *    (ascending vertex id); points/lines by their index in plba_set_points / plba_set_lines.
 *  - a problem handle may be re-used for the next window (the reference builds a fresh optimizer per local BA,
 *    src/mapHandler.cpp:5799): every array stays until its plba_set_* is called again, so a window WITHOUT IMU edges or
 *    a prior clears the previous one's with M = 0 / n = 0; edges that still refer to keyframes or landmarks the current
 *    window does not have make plba_optimize fail with PLBA_ERR_INVALID (checked again at every structure build).
 *    Results do not depend on what the handle held before (tests/test_gpu_parity.py).
 *
 * What each entry point replaces in the reference is cited next to it.
 */
#ifndef PLBA_H
#define PLBA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct plba_problem plba_problem;

typedef enum {
    PLBA_OK = 0,
    PLBA_ERR_INVALID = -1,   /* bad argument / inconsistent sizes / unsorted input   */
    PLBA_ERR_STATE = -2,     /* call order violated (e.g. optimize before upload)    */
    PLBA_ERR_DEVICE = -3,    /* HIP runtime error, no device, kernel image missing    */
    PLBA_ERR_NUMERIC = -4,   /* non-finite input                                       */
    PLBA_ERR_EXCHANGE = -5   /* the multi-GPU exchange callback failed                 */
} plba_status;

/* Edge families of the path. */
typedef enum {
    PLBA_EDGE_POINT = 0,     /* EdgeNavStatePVRPointXYZ  IMU/g2otypes.h:220, .cpp:286  */
    PLBA_EDGE_LINE = 1,      /* EdgeNavStateLine         IMU/g2otypes.h:773, .cpp:1306 */
    PLBA_EDGE_IMU_PVR = 2,   /* EdgeNavStatePVR          IMU/g2otypes.cpp:27,94        */
    PLBA_EDGE_IMU_BIAS = 3,  /* EdgeNavStateBias         IMU/g2otypes.cpp:236,264      */
    PLBA_EDGE_PRIOR = 4      /* EdgeMarginalization      IMU/g2otypes.cpp:1423,1477    */
} plba_edge_kind;

/* Hard-coded constants of the reference exposed as options with identical defaults. */
typedef struct {
    double tau;                 /* g2o LM: lambda_init = tau * max|H_jj|             (1e-5)  */
    double good_step_lower;     /* g2o LM goodStepLowerScale                          (1/3)   */
    double good_step_upper;     /* g2o LM goodStepUpperScale                          (2/3)   */
    int    max_trials;          /* g2o LM maxTrialsAfterFailure                       (10)    */
    double user_lambda_init;    /* g2o LM userLambdaInit, 0 = automatic               (0)     */
    double marg_eps;            /* IMU/marginalization.h:99 pseudo-inverse threshold  (1e-8)  */
    int    fix_line_position_jacobian; /* 0 = reproduce IMU/g2otypes.cpp:1347 (SURVEY B-Q1)   */
                                /* (the marginalization factors are unweighted, as IMU/marginalization.cpp:67 has them — SURVEY B-Q4;
                                   there is no whitening option: round 3 declared one and rejected it at run time, round 4 removed it) */
    int    device;              /* HIP device ordinal, -1 = current device            (-1)    */
    int    use_mfma;            /* 1 = fp64 MFMA trailing update in the dense solve    (1)    */
    int    profile;             /* HIP-event timing into plba_stats.ms_phase: 1 = the dense factorisation launches only
                                   (two events on every 8th trial, scaled to all trials), 2 = every phase (0 = off) */
    int    factor_block;        /* block width of the dense factorisation: 32 or 64                (32)   */
    int    factor_flow;         /* 1 = the whole factorisation as ONE dataflow launch (factor_block 32 with use_mfma;
                                   experimental: measured slower at P = 735), 0 = one launch per block step (0) */
    int    chain_elim;          /* 1 = eliminate the velocity / bias variables (block-tridiagonal, no landmark coupling;
                                   segments between separator keyframes, one workgroup each) ahead of the dense
                                   factorisation (keyframes a prior edge touches keep theirs dense; needs use_mfma and
                                   factor_block 32, and IMU edges between neighbouring keyframes only: otherwise the
                                   dense path is taken); 0 = dense path on the full system                       (1) */
    int    wide_steps;          /* 1 = a launch of the dense factorisation retires 64 columns: the look-ahead workgroup factors
                                   the next 64 x 64 diagonal tile as two pipelined 32-column sweeps (factor_block 32 with
                                   use_mfma, factor_flow 0; experimental: measured slower);
                                   0 = one launch per 32 columns                                                    (0) */
    int    band_solve;          /* how the band of the compact dense system (after the chain elimination) is used when it has at most
                                   three sub-diagonal 32 x 32 tiles — tracks spanning a few consecutive keyframes, as in a sliding
                                   window.  1 = from 8 to 64 tiles the band is cut into two or four chains that are eliminated side by
                                   side in every launch of the dense factorisation (multi-chain form, plba_dense.hip: T - 1 dependent
                                   launches become about T / 2 or less); longer systems are factored and solved by two workgroups walking the band from
                                   both ends with the window resident in LDS (plba_band.hip).  2 = the in-LDS form from 8 tiles on
                                   (tests).  0 = neither.  Wider bands and smaller systems take the plain dense path      (1) */
    int    marg_exact;          /* pseudo-inverse of the dropped block Amm in plba_marginalize* (IMU/marginalization.cpp:351-353 thresholds
                                   the eigenvalues of the WHOLE block).  2 = always the dense eigen-decomposition of Amm (m <= 474 at
                                   the call site).  0 = always block by block (landmark blocks, then the keyframe block of the reduced
                                   system): identical whenever every discarded direction is a block-local null space.  1 = block by
                                   block when a device-side certificate proves that (no other eigenvalue of Amm at or below the
                                   threshold, discarded directions uncoupled), the dense path otherwise                       (1) */
    int    lm_fused;            /* the landmark side of an iteration as fused landmark-major passes (plba_lm_dev.h): observations are
                                   evaluated in registers where they are needed — the Schur complement as a rank-k update per group of
                                   landmarks on the matrix cores, back-substitution + trial residuals in one pass — instead of writing a
                                   record per observation and gathering it three times.  Possible (on one GPU or sharded: the ranks of a window
                                   vote and all take the same path) when the chain path is in effect and no landmark has more than 16
                                   observations (9 .. 16: wide groups, two 8-lane units per landmark block) or two in one keyframe.  2 = whenever possible;
                                   1 = when possible and the window holds at least 40 k observations (BASELINE configs[1] .. [4]; below,
                                   the record-based passes k_linearize / k_landmark_hll / k_schur_pairs / k_backsub run: DESIGN.md 4a);
                                   0 = never                                                                                (1) */
    int    lm_fused_min_obs;    /* lm_fused = 1: the observation count from which the fused passes are taken           (40000) */
    /* ---- measurement / diagnostic knobs (round 3 read these from the environment inside prepare(); they are options now, read
     * once, and nothing in the library calls getenv) ---- */
    int    lm_group_steps;      /* workgroup steps per landmark group of the fused passes, 1 .. 16; 0 = sized to whole rounds of
                                   workgroups on the device                                                  (0) */
    int    chain_seg;           /* keyframes per eliminated velocity / bias chain segment, 1 .. 8; 0 = chosen from the dense layout's
                                   launch count                                                              (0) */
    int    twin_max_tiles;      /* longest compact dense system (32-column tiles) the multi-chain factorisation takes; 0 = up to the
                                   in-LDS band solver's threshold                                                          (0) */
    int    diag;                /* bit 0: prepare() / plba_lba_visual print their lap timings to stderr; bit 1: plba_marginalize* keeps the
                                   stacked Jacobian and residual for plba_debug_get("marg_J") (the 40-digit fixture's input);
                                   bit 2 (fault injection, tests/test_lm_fused.py): the in-launch wait of k_lm_trial is given a count that
                                   never comes — the call must fail with PLBA_ERR_DEVICE, not hang; bit 3 (measurement / tests): the
                                   Jacobi of the marginalization's kept block starts cold, without the tridiagonal pre-rotation    (0) */
} plba_options;
#define PLBA_DIAG_TIMING 1
#define PLBA_DIAG_MARG_DUMP 2
#define PLBA_DIAG_LEAD_WAIT_FAIL 4
#define PLBA_DIAG_NO_MARG_PREROTATE 8

void plba_default_options(plba_options* o);

/* Result of one optimize() call = g2o SparseOptimizer::optimize(n). */
typedef struct {
    int    iterations;          /* outer LM iterations executed (g2o return value)             */
    int    trials;              /* total damped trial solves                                    */
    int    stop_reason;         /* 0 ran all iterations, 1 LM Terminate, 2 abort flag           */
    int    solver_failures;     /* trials whose reduced-camera Cholesky hit a pivot <= 0        */
    double chi2_initial;        /* activeRobustChi2 before the first iteration                  */
    double chi2_final;          /* activeRobustChi2 of the state left in the problem            */
    double lambda_final;
    double ms_total;            /* wall time of this call; on return every launch of the call has completed (its last trial's
                                   control block was read back; launches queued ahead for a next iteration that did not
                                   come — abort, LM Terminate — are waited for)                                          */
    double ms_phase[8];         /* device ms by HIP events when options.profile: [0] time inside the linearising launches (observations +
                                   IMU / prior edges, Jacobians) wherever they run — at the head of an iteration, or as the trial pass
                                   that linearises the trial state while it measures it; plba_debug_get "prof_lin_launches" counts them —, [1] the dense factorisation launches alone (first-block launch + one per block
                                   step; profile >= 1), [2] landmark inverse + assemble + Schur pairs, [3] dense solve (factorisation + back-substitution),
                                   [4] landmark back-substitution + state update, [5] errors-only trial passes, [6] exchange, [7] landmark Hll + reductions */
} plba_stats;

/* One row per LM trial, for golden traces (tests/golden). */
typedef struct {
    int    iteration, trial, accepted, solver_ok;
    double lambda, chi2_current, chi2_trial, scale, rho;
} plba_trace_row;

/* Output of plba_marginalize = the state MarginalizationInfo carries between BA calls
 * (IMU/marginalization.h:82-95).  J0 is n x n column-major like Eigen's linearized_jacobians. */
typedef struct {
    int      n;                 /* kept dimension                                              */
    int      m;                 /* dropped dimension                                           */
    int      nv;                /* kept vertices                                                */
    int32_t* vid;               /* keep_vertex_id   [nv]                                        */
    int32_t* size;              /* keep_vertex_size [nv] (9 PVR / 6 bias)                       */
    int32_t* idx;               /* keep_vertex_idx - m [nv]                                     */
    double*  x0;                /* keep_vertex_data packed: 10 doubles per PVR, 6 per bias      */
    double*  J0;                /* linearized_jacobians  n*n colmajor                           */
    double*  r0;                /* linearized_residuals  n                                      */
    double*  Ar;                /* reduced information A' (n*n, row-major == col-major, symmetric) */
    double*  br;                /* reduced b' (n)                                               */
} plba_prior;

/* Multi-GPU exchange hook.  Called on the problem's thread with a DEVICE buffer of n
 * doubles that must be all-reduced in place over every rank (op 0 = sum, 1 = max), ordered on
 * `stream` (a hipStream_t).  The host side supplies it (RCCL ncclAllReduce, or torch.distributed). */
typedef int (*plba_allreduce_fn)(void* user, double* device_buf, size_t n, int op, void* stream);

/* ---- lifetime ------------------------------------------------------------------------------ */
int  plba_create(const plba_options* opt, plba_problem** out);   /* g2o::SparseOptimizer ctor + solver chain, mapHandler.cpp:5787-5794 */
void plba_destroy(plba_problem* p);
const char* plba_last_error(const plba_problem* p);             /* p may be NULL: last create() error */
const char* plba_backend_name(void);                             /* "hip-gfx950" */

/* ---- problem upload (graph construction of mapHandler.cpp:5799-6034) ----------------------- */
int plba_set_camera(plba_problem* p, double fx, double fy, double cx, double cy,
                    const double Rbc[9], const double Pbc[3]);    /* SetParams, IMU/g2otypes.h:280-288 */
int plba_set_gravity(plba_problem* p, const double gw[3]);       /* EdgeNavStatePVR::SetParams h:116 */
/* vid_* ascending; vid_bias[k] = -1 when keyframe k has no bias vertex (configs 1-2). */
int plba_set_keyframes(plba_problem* p, int K, const int32_t* vid_pvr, const int32_t* vid_bias,
                       const double* P3, const double* V3, const double* q_xyzw4,
                       const double* bg3, const double* ba3, const double* dbg3, const double* dba3,
                       const uint8_t* fixed_pvr, const uint8_t* fixed_bias);
int plba_set_points(plba_problem* p, int Np, const double* xyz3, const uint8_t* fixed /*may be NULL*/);
int plba_set_lines(plba_problem* p, int Nl, const double* sPeP6, const uint8_t* fixed /*may be NULL*/);
/* observations must be landmark-major (pt[] non-decreasing), the reference's edge insertion order */
int plba_set_point_obs(plba_problem* p, int Ep, const int32_t* pt, const int32_t* kf,
                       const double* uv2, const double* inv_sigma2);
int plba_set_line_obs(plba_problem* p, int El, const int32_t* ln, const int32_t* kf,
                      const double* l3, const double* inv_sigma2);
/* preint142 = dP3 dV3 dR9 JPg9 JPa9 JVg9 JVa9 JRg9 cov81 dt (IMU/IMUPreintegrator.h:187-201);
 * info_pvr81 = cov^-1 (mapHandler.cpp:5269); info_bias36 = InvCovBgaRW/dt (:5287). */
int plba_set_imu_edges(plba_problem* p, int M, const int32_t* kf_i, const int32_t* kf_j,
                       const double* preint142, const double* info_pvr81, const double* info_bias36);
int plba_set_prior(plba_problem* p, int n, int nv, const int32_t* vid, const int32_t* size,
                   const int32_t* idx_minus_m, const double* x0_packed,
                   const double* J0_colmajor, const double* r0);  /* nv == 0 clears; mapHandler.cpp:6007-6034 */
int plba_set_robust(plba_problem* p, plba_edge_kind kind, int enabled, double huber_delta); /* setRobustKernel/setDelta */
/* setLevel, POINT/LINE only.  New edges are level 0: plba_set_point_obs resets the levels of BOTH kinds (the point count moves
 * the line range), plba_set_line_obs those of the line edges only. */
int plba_set_levels(plba_problem* p, plba_edge_kind kind, const uint8_t* level);
int plba_get_levels(plba_problem* p, plba_edge_kind kind, uint8_t* level);

/* ---- sliding window --------------------------------------------------------------------------------------------------
 * The mapping thread runs one local BA per new keyframe on a window that differs from the previous one by addKeyframeToSW /
 * deleteKeyframeInSW (src/mapHandler.cpp:1178-1221, 4815-4825): the oldest keyframe(s) leave, one arrives, and the local map is
 * "every landmark whose FIRST observation lies inside the window" (:5769-5783), so the landmarks first seen from a leaving
 * keyframe leave with it and every other landmark keeps all of its observations.  plba_slide_window edits the uploaded window
 * in place instead of setting every array again:
 *   - the n_drop oldest keyframes (indices 0 .. n_drop-1) leave, with every landmark that has an observation from one of them
 *     and every IMU edge that touches one of them; further landmarks / observations may be dropped by mask (the map's
 *     removeBadMapLandmarks and the culling of :5541-5620 between two BA calls);
 *   - the KEPT keyframes and landmarks keep the estimates the DEVICE holds — the previous plba_optimize's result, i.e. what
 *     the reference writes back to the map (:6202-6239) and reads again when it builds the next graph; nothing of them
 *     crosses PCIe;
 *   - K_add keyframes, Np_add / Nl_add landmarks, M_add IMU edges and Ep_add / El_add observations are appended.  Keyframe
 *     indices (po_kf, lo_kf, imu_kf_i / _j) are those AFTER the slide (old index - n_drop; the added keyframes follow).
 *     Landmark indices of the added observations are those BEFORE the slide for landmarks that stay (they must stay) and
 *     Np_before + i / Nl_before + i for the i-th added one; both lists sorted by that index.  A kept landmark's new
 *     observations are listed after its old ones, as the reference's kf_obs_list grows.
 *   - all edges are level 0 again and the prior is kept as set: call plba_set_prior for the new window's prior (or clear it).
 * point_map[Np_before] / line_map[Nl_before] (optional) receive each old landmark's new index or -1.  The structure the next
 * plba_optimize builds — and therefore every result, bit for bit — is that of a fresh handle given the same window through
 * plba_set_* (tests/test_slide_window.py).  One GPU only (a sharded problem takes a fresh upload). */
typedef struct {
    int n_drop;
    const uint8_t* drop_point;      /* [Np_before] 1 = leaves too; may be NULL */
    const uint8_t* drop_line;       /* [Nl_before] */
    const uint8_t* drop_point_obs;  /* [Ep_before] 1 = this observation leaves (its landmark may stay); may be NULL */
    const uint8_t* drop_line_obs;   /* [El_before] */
    int K_add;                      /* appended keyframes: arrays as in plba_set_keyframes */
    const int32_t* vid_pvr; const int32_t* vid_bias;
    const double *P3, *V3, *q_xyzw4, *bg3, *ba3, *dbg3, *dba3;
    const uint8_t* fixed_pvr;       /* [K_after] fixed flags of the WHOLE new window (the new oldest keyframe becomes fixed, */
    const uint8_t* fixed_bias;      /*           :5812-5825); NULL = kept keyframes keep theirs, added ones are free          */
    int M_add;                      /* appended IMU edges: arrays as in plba_set_imu_edges */
    const int32_t *imu_kf_i, *imu_kf_j;
    const double *preint142, *info_pvr81, *info_bias36;
    int Np_add; const double* xyz3; const uint8_t* point_fixed;
    int Nl_add; const double* sPeP6; const uint8_t* line_fixed;
    int Ep_add; const int32_t* po_pt; const int32_t* po_kf; const double* uv2; const double* po_inv_sigma2;
    int El_add; const int32_t* lo_ln; const int32_t* lo_kf; const double* l3; const double* lo_inv_sigma2;
} plba_slide;
int plba_slide_window(plba_problem* p, const plba_slide* s, int32_t* point_map, int32_t* line_map);
/* sizes of the uploaded window: out6 = [K, Np, Nl, Ep, El, M] (optimizer.vertices().size() / edges().size() by kind) */
int plba_get_sizes(const plba_problem* p, int32_t* out6);

/* ---- multi-GPU: this problem holds a landmark shard; pose-side edges are added by rank 0 only */
int plba_set_shard(plba_problem* p, int rank, int world, plba_allreduce_fn fn, void* user);
int plba_set_stream(plba_problem* p, void* hip_stream);           /* run on a caller stream (e.g. torch's) */

/* ---- solve ---------------------------------------------------------------------------------- */
/* = initializeOptimization(0); optimize(max_iters)  (mapHandler.cpp:6038-6039, 6068-6069). */
int plba_optimize(plba_problem* p, int max_iters, const volatile uint8_t* abort_flag, plba_stats* out);
/* chi2 > thresh || !isDepthPositive  =>  level 1, for POINT and LINE edges, then robust kernels
 * off on both kinds (mapHandler.cpp:6047-6066).  Returns the number of edges moved to level 1. */
int plba_gate_outliers(plba_problem* p, double chi2_thresh, int* n_point_out, int* n_line_out);
int plba_recompute_errors(plba_problem* p);                       /* computeActiveErrors on the current state */
/* per-edge chi2 (e^T Omega e, non-robustified) as of the last evaluation pass
 * and isDepthPositive on the current estimates; either pointer may be NULL. */
int plba_get_edge_chi2(plba_problem* p, plba_edge_kind kind, double* chi2, uint8_t* depth_positive);
int plba_get_trace(plba_problem* p, plba_trace_row* rows, int cap, int* n);
/* Observation culling decision of the call site after the final optimize() (mapHandler.cpp:5541-5556 points, :5611-5620
 * lines): an edge is bad iff  chi2() > thresh || !isDepthPositive()  on the final estimates, where a
 * level-1 (gated-out) edge first gets computeError() — its cached error is refreshed, as in the reference — and a
 * level-0 edge uses the error cached by the last evaluation pass.  bad_point[Ep] / bad_line[El] receive 0 / 1 (either
 * may be NULL); the counts are optional.  What the reference then does with a bad observation (erasing it from the map,
 * covisibility bookkeeping) is host map surgery and stays with the caller.  Returns the number of bad observations. */
int plba_cull_observations(plba_problem* p, double chi2_thresh, uint8_t* bad_point, uint8_t* bad_line, int* n_point_out, int* n_line_out);

/* ---- dense symmetric positive definite solve on the device (K7 stand-alone) ------------------------------------------
 * Solves A x = b for an n x n row-major A with the kernels plba_optimize uses for the reduced camera system (block
 * LL^T on the fp64 matrix cores + back-substitution): what g2o's LinearSolverEigen / LinearSolverCholmod do for the
 * pose graphs of loopClosureOptimizationEssGraphG2O / ...CovGraphG2O (src/mapHandler.cpp:4068-4297, 4299-4470), whose
 * host-evaluated graphs the facade sends here once they exceed 384 dims (include/plba_g2o/g2o_compat.h, solveHost).
 * *ok = 0 on a non-positive pivot (g2o: "Cholesky failure").  `p` supplies the device, the stream and the error text. */
int plba_dense_solve(plba_problem* p, int n, const double* A, const double* b, double* x, int* ok);

/* ---- results (write-back of mapHandler.cpp:6202-6239) --------------------------------------- */
int plba_get_keyframes(plba_problem* p, double* P3, double* V3, double* q_xyzw4, double* dbg3, double* dba3);
int plba_get_points(plba_problem* p, double* xyz3);
int plba_get_lines(plba_problem* p, double* sPeP6);
/* device-side snapshot/restore of all estimates (used by benches to replay a window) */
int plba_save_state(plba_problem* p);
int plba_restore_state(plba_problem* p);

/* ---- marginalization (mapHandler.cpp:6075-6199, IMU/marginalization.cpp:128-147,291-384) ----- */
int  plba_marginalize(plba_problem* p, int first_kf, int max_edges_per_kind /*NUM=50 admits 51*/,
                      plba_prior* out);
/* General form behind MarginalizationInfo::addResidualBlockInfo / preMarginalize / marginalizeWithoutThread
 * (IMU/marginalization.cpp:102-147,291-384): explicit factor lists.  Each IMU edge contributes its PVR edge and its
 * bias edge; point_edges / line_edges index the uploaded observation arrays; drop_vid lists the keyframe vertices to
 * marginalize out (the landmark of every listed observation is always dropped, drop_set {0} at the call site). */
int  plba_marginalize_factors(plba_problem* p, int n_imu, const int32_t* imu_edges, int n_pt, const int32_t* point_edges,
                              int n_ln, const int32_t* line_edges, int use_prior, int n_drop, const int32_t* drop_vid,
                              plba_prior* out);
void plba_prior_free(plba_prior* pr);
/* MarginalizationInfo::eps (IMU/marginalization.h:99 — a public, mutable member; the thresholds of
 * IMU/marginalization.cpp:353,365-366 read it): replaces options.marg_eps for the following plba_marginalize* calls. */
int  plba_set_marg_eps(plba_problem* p, double eps);

/* ---- IMU preintegration producer (upstream of plba_set_imu_edges) --------------------------
 * KeyFrame::ComputeIMUPreIntSinceLastFrame (src/keyFrame.cpp:139-172) for M keyframe intervals at once: per interval
 * IMUPreintegrator::reset (IMU/IMUPreintegrator.cpp:47-76), then one IMUPreintegrator::update (:80-139) per selected
 * sample.  Interval m owns the samples [sample_start[m], sample_start[m+1]) of t / gyr3 / acc3 (the keyframe's `imus`
 * vector); t_prev / t_curr are the two image times, bg3 / ba3 the previous keyframe's biases
 * (NavState::Get_BiasGyr / Get_BiasAcc).  Time stamps are `long double` as in the reference (IMU/imudata.h:58): every
 * dt is formed in long double on the host and rounded to double exactly where `double dt = ...` does there.  Sample
 * selection follows the reference literally, including the last partial step `dt = curr_t - t[i]` taken with the first
 * sample PAST curr_t (a negative dt; keyFrame.cpp:162-167).  gyr_meas_cov / acc_meas_cov: the diagonal value of
 * IMUData::getGyrMeasCov / getAccMeasCov (IMU/imudata.cpp:27-28).  out142: M x 142 payloads in the
 * plba_set_imu_edges layout.  `p` supplies the device, the stream and the error text; nothing needs to be uploaded. */
int plba_preintegrate(plba_problem* p, int M, const int32_t* sample_start, const long double* t, const double* gyr3,
                      const double* acc3, const long double* t_prev, const long double* t_curr, const double* bg3,
                      const double* ba3, double gyr_meas_cov, double acc_meas_cov, double* out142);

/* ---- pre-VIO-init visual-only local BA -------------------------------------------------------
 * MapHandler::levMarquardtOptimizationLBA (src/mapHandler.cpp:1441-2098), the optimiser localBundleAdjustment
 * (:1329-1439) runs while the IMU is not initialised: hand-rolled LM over [6 per local keyframe | 3 per point | 6 per
 * line], scalar residual = norm of the reprojection error, Cauchy weights (stvo-pl/src/auxiliar.cpp:556-559),
 * multiplicative damping, lambda schedule and termination tests exactly as coded there.  Deviations from the source text
 * (DESIGN.md §9): a line's end points are read from its own 6 entries (:1787-1788 read both from one, 3-strided, offset); a non-positive
 * pivot ends the run with stats->solver_failed = 1 (SimplicialLDLT has no such exit).  The stale map pose of the line
 * pass (:1790) IS reproduced unless use_iterate_poses != 0. */
typedef struct plba_lba_options {
    double lambda_lm;           /* SlamConfig::lambdaLbaLM      initial lambda, scaled by max |H_ii| (:1653-1659) */
    double lambda_k;            /* SlamConfig::lambdaLbaK       lambda /= k when the error grew, *= k otherwise (:1894-1900) */
    int    max_iters;           /* SlamConfig::maxItersLba */
    double homog_th;            /* SlamConfig::homogTh          floor of gz^2 and of the error norm in the Jacobians */
    double min_error;           /* Config::minError             (:1884) */
    double min_error_change;    /* Config::minErrorChange       (:1884, :1911) */
    int    use_iterate_poses;   /* 0 = the line pass linearises at the map poses, as the reference does (:1790) */
    int    variant;             /* 0 = levMarquardtOptimizationLBA; 1 = levMarquardtOptimizationGBA (:2210-2812): the same text
                                   with `int Hmax` (:2468: lambda scales with the truncated maximum), the error divided by the zero
                                   counters in EVERY pass (:2744: every step is taken) — and machine epsilon for both thresholds
                                   (:2746, :2776), which the caller passes as min_error / min_error_change */
} plba_lba_options;
typedef struct plba_lba_stats {
    int    iterations;          /* linear solves performed */
    int    updates;             /* of which applied to X */
    double err_first, err_last; /* robust error of the first pass / observations (the reference's own first value is x / 0 = inf, :1650,
                                   reproduced internally); of the last pass / landmarks (:1882) */
    double lambda;              /* lambda when the loop ended */
    int    solver_failed;
    int    reserved;
} plba_lba_stats;
void plba_lba_default_options(plba_lba_options* o);
/* K keyframes appear in the observations; kf_loc[k] = index 0..Nkf-1 of keyframe k among the optimised ones (kf_list
 * order) or -1 when it only anchors landmarks (:1516-1521).  T_kf_w16: K row-major 4x4 map poses (KeyFrame::T_kf_w;
 * x_kf_w is taken as logmap_se3 of it).  xyz3 (Np x 3) / pq6 (Nl x 6): in = map estimates, out = optimised.
 * Observations are landmark-major as localBundleAdjustment builds its lists.  T_out16: K x 16, the optimised poses
 * (unchanged for kf_loc = -1).  pt_moved / ln_moved (optional): 1 where the landmark moved more than 1 cm, for which the
 * reference clears `inlier` (:1944-1970).  `p` supplies the device, the stream and the error text. */
int plba_lba_visual(plba_problem* p, const plba_lba_options* opt, int K, const double* T_kf_w16, const int32_t* kf_loc,
                    int Np, double* xyz3, int Nl, double* pq6,
                    int Ep, const int32_t* po_pt, const int32_t* po_kf, const double* uv2,
                    int El, const int32_t* lo_ln, const int32_t* lo_kf, const double* l3,
                    double fx, double fy, double cx, double cy, double* T_out16, uint8_t* pt_moved, uint8_t* ln_moved,
                    plba_lba_stats* stats);

/* ---- diagnostics used by the parity tests (not needed by a drop-in caller) ------------------- */
/* Runs computeActiveErrors + buildSystem + setLambda(lambda) + Schur on the current state without
 * updating it, then exposes named internal buffers: "Hschur" (P*P row-major), "bschur" (P),
 * "bp" (P), "x" (P + 3Np + 6Nl after a solve), "hll_pt" (Np*9), "bl_pt" (Np*3), "hll_ln" (Nl*36),
 * "bl_ln" (Nl*6), "err_pvr" (M*9), "err_bias" (M*6), "err_prior" (n), "pose_dim" (1), "chi2" (1),
 * "maxdiag" (1); "marg_path" (5, after plba_marginalize*): [0] 0 = block-wise pseudo-inverse taken, 1 = dense
 * eigen-decomposition of Amm; [1..4] the certificate's w_max, smallest kept landmark eigenvalue, tau, smallest pivot. */
int plba_debug_build(plba_problem* p, double lambda, int do_solve);
int plba_debug_get(plba_problem* p, const char* what, double* out, size_t cap, size_t* n);
/* the same entry under its round-1 name (tests/test_gpu_parity.py); product code calls plba_dense_solve */
int plba_debug_dense_solve(plba_problem* p, int n, const double* A, const double* b, double* x, int* ok);

#ifdef __cplusplus
}
#endif
#endif /* PLBA_H */
