// plba_internal.h — device-resident problem layout shared by the HIP translation units.
//
// HBM layout (all fp64 unless noted; E = Ep + El observations, L = Np + Nl landmark "slots",
// points first; P = pose-side dimension, Ppad = P rounded up to the 64-wide dense tile):
//   kf[2]    K x 24      keyframe records, double-buffered (current / trial) + one saved copy
//   lm[2]    L x 6       landmark estimates (points use 3), double-buffered + saved
//   obs_*    E           landmark-major observation arrays: uv / l3, inv_sigma2, kf, slot, level
//   erec     E x 16      compact per-observation linearisation record = one 128-byte line (plba_math.h: EREC),
//                        stored KEYFRAME-major (ob_pos) so the Schur gathers of a keyframe pair are ascending streams
//   hll,bl   L x 12, L x 6   landmark blocks (points: 6 upper; lines: two 3x3 uppers)
//   dinv,tv  L x 12, L x 6   (Hll + lambda I)^-1 and (Hll + lambda I)^-1 bl
//   pairs    CSR of keyframe pairs sharing landmarks -> (edge_i, edge_j) entries for the Schur complement
//   Himu     Ppad x ld   IMU edges of this iteration (Himu_alt: zeroed by k_assemble, swapped in when a step is accepted);
//                        Hconst = prior J0^T J0 scattered (constant), added by k_assemble
//   sys      (Ppad+64) x ld  reduced camera system, augmented: row Ppad = bschur, row Ppad+1 = bp
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "plba.h"
#include "plba_math.h"

namespace plba {

constexpr int TILE = 64;          // dense tile edge (wavefront-wide)

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: raise it once per (device, kernel), under a
// lock, and report a failure instead of letting the launch that needs the LDS fail later.
inline hipError_t ensure_dyn_lds(const void* func, int bytes) {
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, int> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(mu);
    auto it = done.find({dev, func});
    if (it != done.end() && it->second >= bytes) return hipSuccess;
    e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done[{dev, func}] = bytes;
    return e;
}
constexpr int MAX_PART = 8192;    // per-block partial sums for deterministic reductions

// LM control block, device-resident; copied back once per trial.
struct Ctrl {
    double lambda, ni, current_chi, temp_chi, scale, rho, maxdiag, pad0;
    int accepted, solver_ok, iteration, trial, n_gate_pt, n_gate_ln, n_fail, sync_fail /* an in-launch wait ran into its bound */;
};
// Pinned, device-mapped host memory k_decide writes the control block into at the end of every LM trial: the host
// polls `seq` instead of paying for a device-to-host copy launch plus a stream synchronisation per trial.
struct Mailbox {
    Ctrl c;
    unsigned long long seq;
    plba_trace_row row;      // the trial's trace row (also appended to the device-side trace): the host keeps the trace from here, no read-back at the end of a call
};

// one chunk (<= 256 entries) of a keyframe pair's entry list: everything k_schur_pairs needs to know up front, 32 bytes
struct ChunkMeta { int32_t slot /* index in pair-major order: where its partial sums go */, start, end, ij /* i | j << 16 */, nch, ch0 /* the pair's chunks: slots [ch0, ch0 + nch) */, oi, oj /* kf_off_pvr of i, j */; };

struct ChainView;
struct DevBuf {  // trivially-copyable view of device pointers passed to kernels by value
    // sizes
    int K, Np, Nl, L, Ep, El, E, M, P, Ppad, ld, npairs, nent, nchunks;
    // camera / gravity
    Cam cam;
    V3 gw;
    int fix_q1;
    // state
    double* kf[2];
    double* lm[2];
    // observations
    const double* po_uv;   // Ep x 2
    const double* lo_l;    // El x 3
    const double* ob_w;    // E inv_sigma2 (float-rounded)
    const int32_t* ob_kf;  // E
    const int32_t* ob_slot;  // E
    uint8_t* ob_level;     // E
    double* ob_chi2;       // E   cached e^T Omega e of the last evaluation pass
    double* erec;          // E x 16, indexed by ob_pos[e]
    double* erec_alt;      // the idle record table: a trial that is linearised while it is measured writes here (swapped in on acceptance)
    const int32_t* ob_pos; // E  keyframe-major position of observation e's record
    // landmarks
    const int32_t* lm_start;  // L + 1 (unified edge index)
    const uint8_t* lm_fixed;  // L
    double *hll, *bl, *dinv, *tv, *xl;
    uint8_t* lm_active;
    // keyframes
    const int32_t *kf_off_pvr, *kf_off_bias;
    // pairs
    const int32_t *pair_i, *pair_j, *pair_start, *ent_pi, *ent_pj, *ent_slot;   // entries: record positions + landmark slot
    const ChunkMeta* ch_meta;                                                   // <= 256-entry chunks of the pair lists (k_schur_pairs)
    double* schur_part;    // nchunks x 48 partial sums
    const int32_t* alist;  // chain path: linear indices of the entries of sys the pose-side assembly has to rebuild (null: all of them)
    int nalist;
    const int32_t* xlist;  // sharded runs: linear indices (r * ld + c, r >= c) of the lower-triangle entries that can be non-zero before the factorisation
    int nxlist;
    int* pair_cnt;         // arrival counters (indexed by a pair's first chunk slot), zero between launches
    int* trial_cnt;        // arrival counter of the trial-error launch's workgroups (the last one decides), zero between launches
    unsigned* back_cnt;    // chain back-substitution segments finished, all k_lm_trial launches so far (monotonic; the launch's pose-side blocks wait for it)
    // IMU
    const int32_t *imu_i, *imu_j;
    const double *imu_pre, *imu_info_pvr, *imu_info_bias;
    double* imu_err;    // M x 16 (9 pvr + 6 bias + pad)
    double* imu_chi;    // M x 4  (raw pvr, raw bias, robust pvr, robust bias)
    // prior
    int pr_n, pr_nv;
    const int32_t *pr_kf, *pr_isbias, *pr_size, *pr_idx, *pr_x0off, *pr_off;
    const double *pr_x0, *pr_J0, *pr_r0;
    double *pr_err, *pr_dx, *pr_chi;
    // dense
    double *bprior, *bprior_alt;   // gradient of the prior edge (plain stores, swapped with the IMU accumulators)
    double *Hconst, *Himu, *bimu, *Himu_alt, *bimu_alt, *sys, *Lfac, *bpg, *x, *Linv32;   // Lfac: Cholesky factor (same shape as sys)
    double *Ninv, *Nwork;  // Ppad x ld each: N = L^-T, carried through the factorisation launches as identity rows of the augmented system
                           // (Nwork: the rows' unsolved trailing part); null: substitution instead
    double* Linv;          // T x 64 x 64 inverses of the diagonal tiles of Lfac
    double* LTblk;         // (Ppad/fb) x fb x fb transposed diagonal blocks of the factor, LT[j][t] = L[t][j]
    double* rdblk;         // Ppad reciprocals of the factor's diagonal
    // Linv32 (declared above; shares LTblk's storage): (Ppad/32) x 32 x 32 inverses of the diagonal blocks, row-major
    int fb;                // factorisation block width (32 or 64)
    int* flow_flags;       // T epoch-stamped flags of the back-substitution dataflow
    int* chol_flags;       // (T32 + 1) x T32 tile flags + T32 inverse flags of the single-launch factorisation (epoch-stamped)
    double* dbgbuf;        // 64 doubles for diagnostic builds (cycle stamps)
    int band;              // 1: the banded twisted solver handles this system (plba_band.hip): k_chain_schur leaves tile (0,0) unfactored
    // multi-chain ("twin") form of the multi-launch factorisation of a banded system (plba_dense.hip): the system is stored
    // PERMUTED [chain 0 | chain 1 | ... | separators], so that the chains are independent stretches of one Cholesky
    int twin_m0;           // first (permuted) tile of the separator region; 0: off
    const int32_t* cs_order;   // k_chain_schur: tile (ta << 16 | tb) of each workgroup — the tiles that are factored on the spot first, then the band, then the rest; null = natural order
    const int32_t* twin_fac;   // per NATURAL diagonal tile: -1, or the permuted tile it becomes as the first tile of a chain (| 1 << 16: turned
                               // around) — k_chain_schur factors those on the spot
    const int32_t* perm;   // Ppad: natural dense index -> permuted (k_chain_schur writes through it)
    const int32_t* xmap;   // Ppad: permuted -> natural (k_back_gemv writes x through it)
    double* wtw;           // (T (T + 1) / 2 + T) x 1024: the W^T W tile of every k_chain_schur workgroup, when the k_lm_gather launch forms them (round 4)
    double* alt2;          // ... and of the second stage's accumulating chain (nested plan)
    double* alt;           // same shape as sys: the bottom chain's updates of the middle block (folded in by the top chain's last step)
    int flow;              // 1: single-launch dataflow factorisation (k_chol_flow), 0: one launch per block step
    int wide;              // 1: a launch retires 64 columns (two pipelined 32-column sweeps in the look-ahead workgroup, k_chol64)
    // reductions / control
    double *chi_part, *scale_part, *maxd_part, *kfdiag, *posediag;
    Ctrl* ctrl;
    plba_trace_row* trace;
    int trace_cap;
    int* trace_n;
};

struct Robust { int on[5]; double delta[5]; };

// ---- fused landmark-major passes (plba_lm_dev.h): group structure built at upload ---------------------------------------------
constexpr int LMF_W = 8;                              // keyframes in a STANDARD group's window = observations per landmark it takes
constexpr int LMF_W2 = 16;                            // ... in a WIDE group's (round 4): landmarks seen from 9 .. 16 keyframes — the reference's own 12-keyframe
                                                      // sliding window (include/mapHandler.h:217) with tracks over most of it.  A wide landmark takes two 8-lane
                                                      // units (window slots 0 - 7 | 8 - 15), plba_lm_dev.h
// A group leaves for the gather pass: npair pose-pair blocks (p <= q, index q (q + 1) / 2 + p) x 36 entries, then per window slot bp (6) |
// bs = Hpl D bl (6).  npair / the stride are the problem's (LmView): 36 / 8 slots when every group is standard, 136 / 16 slots otherwise.
struct LmGroup { int32_t lm0, nlm, nw, kind /* bit 0: lines, bit 1: wide */; int32_t kf[LMF_W2]; int32_t off[LMF_W2]; };      // landmarks [lm0, lm0 + nlm) of the group
                                                      // order; window keyframes (ascending) and their kf_off_pvr (-1: fixed pose)
struct LmView {
    int ngrp;
    const LmGroup* grp;
    const int32_t* lm_slot;       // group order -> landmark slot
    const int32_t* lm_ob0;        // group order (+ 1): first observation of each landmark in the group-ordered observation arrays
    const int32_t* ob_orig;       // group order -> unified observation index (ob_level, ob_chi2)
    double* ob_chi_g;             // E, GROUP order: the cached per-observation chi2 the fused passes leave (coalesced; k_lm_chi_sync scatters it into ob_chi2 for the gating / culling / read-back that follow a call)
    int wmax, npair, part_stride; // 8 / 36 / 36 * 36 + 8 * 12 when every group is standard; 16 / 136 / 136 * 36 + 16 * 12 when some are wide
    const uint8_t* lm_ws8;        // group order, wmax per landmark: per window slot, the offset (in the landmark's observation range) of the observation made from that keyframe, 0xFF = none
    const uint8_t* lm_fixed_g;    // group order copy of lm_fixed
    uint8_t* ob_level_g;          // group order copy of ob_level (refreshed whenever the levels change: launch_lm_level_sync)
    const double* meas_pt;        // 2 per point observation (group order: points first, [0, Ep))
    const double* meas_ln;        // 3 per line observation ([Ep, E))
    const double* ob_wt;          // inv_sigma2
    double* part;                 // ngrp x part_stride
    int nblk;                     // pose-pair blocks some landmark couples (both keyframes free)
    const int32_t* blk_ij;        // i | j << 16  (i <= j)
    const int32_t* blk_start;     // nblk + 1
    const int32_t* blk_src;       // contributing (group * npair + pair index), ascending group
    int nrow;                     // free keyframes with observations
    const int32_t* row_kf;
    const int32_t* row_start;     // nrow + 1
    const int32_t* row_src;       // contributing (group * wmax + slot)
    const int32_t* alist2;        // d.alist without the entries of the gathered blocks
    int nalist2;
    const uint8_t* col_gather;    // ld: 1 = the gather pass writes this right-hand-side column
    double* ob_err;               // E x 2 residuals, written by the first-iteration pass for the parity tests (null: off)
    int dbg_out;                  // 1: also leave Hll, bl, lm_active, xl where the record-based path leaves them
    int lm_grouped;               // 1 (round 5): DevBuf::lm[] is stored in GROUP order — landmark gi of the group tables sits at lm + 6 gi — so a group streams
                                  // its landmarks' estimates instead of gathering 48-byte slots in random order (one 128-byte line each: 3.1 of the
                                  // 7.3 MB k_lm_schur fetched per launch at configs[2]); ob_slot then holds positions too, lm_slot the true slots (debug outputs)
};

}  // namespace plba

// ---- launchers implemented in plba_kernels.hip / plba_dense.hip / plba_marg.hip -----------------------
namespace plba {
struct LmParams { double tau, lower, upper, user_lambda; int max_trials; };

struct Mailbox;
struct DecideFusion { LmParams lp; double* red; Mailbox* mail; unsigned long long seq; };     // the LM decision taken by the last workgroup of the trial-error launch
void launch_linearize(const DevBuf& d, int state, bool jac, const Robust& rb, bool with_pose_edges, hipStream_t s, bool spec = false, const DecideFusion* df = nullptr);
                                                                        // spec: gated on the device-side LM decision; df: errors-only pass that also decides
void launch_pose_edges(const DevBuf& d, int state, bool jac, const Robust& rb, bool owns_pose_edges, hipStream_t s);
bool launch_landmark_hll(const DevBuf& d, int state, bool fuse_dinv_assemble, bool add_lambda, hipStream_t s, bool spec = false);   // spec: gated on the device-side LM decision
void launch_kfdiag(const DevBuf& d, int state, bool with_posediag, hipStream_t s);
void launch_landmark_dinv(const DevBuf& d, hipStream_t s);
void launch_assemble(const DevBuf& d, bool add_lambda, hipStream_t s);
void launch_schur_pairs(const DevBuf& d, int state, const ChainView* lead /* chain segments riding in front, or null */, hipStream_t s);
constexpr int SCHUR_XCD = 8;     // chunk order and lead padding assume workgroups are dealt round-robin over this many XCDs (speed only)
void launch_backsub(const DevBuf& d, int cur, int trial, const ChainView* lead /* chain back-substitution riding in front, or null */, const double* xd /* dense solution (dd.x) */, hipStream_t s);
void launch_update_kf(const DevBuf& d, int cur, int trial, hipStream_t s);
// red[0] = activeRobustChi2 (local), red[1] = landmark part of computeScale (local), red[2] = max |Hll_jj| (local)
void launch_reduce(const DevBuf& d, bool owns_pose_edges, double* red, hipStream_t s);
size_t tri_packed_size(const DevBuf& d);   // doubles in the packed lower block-triangle + the two rhs rows
size_t list_packed_size(const DevBuf& d);  // doubles in the structural exchange buffer (d.xlist entries + the two rhs rows)
void launch_list_pack(const DevBuf& d, double* buf, bool unpack, hipStream_t s);
void launch_tri_pack(const DevBuf& d, double* buf, bool unpack, hipStream_t s);
void launch_lambda_init2(const DevBuf& d, const LmParams& lp, double* red, bool first_iter, int iteration, bool fused, bool keep_chi, hipStream_t s);
void launch_decide(const DevBuf& d, const LmParams& lp, double* red, bool fused, Mailbox* mail, unsigned long long seq, hipStream_t s);
void launch_ctrl_reset(const DevBuf& d, hipStream_t s);      // control block and trace counter of a fresh plba_optimize call
// fused landmark-major passes (plba_lm_dev.h)
void launch_lm_schur(const DevBuf& d, const LmView& lv, int state, const Robust& rb, bool diag_pass, const ChainView* lead /* chain segments riding in front, or null */, bool spec, hipStream_t s, bool with_pose_edges = false);
void launch_lm_gather(const DevBuf& d, const LmView& lv, bool diag_pass, bool add_lambda, bool spec, hipStream_t s, const ChainView* wtw_cv = nullptr, const DevBuf* wtw_dd = nullptr);      // wtw_*: + the W^T W tiles of the chain Schur complement
void launch_lm_trial(const DevBuf& d, const LmView& lv, int cur, int trial, bool jac, const Robust& rb, const ChainView* lead, const double* xd, unsigned back_target, bool with_pose_edges, const DecideFusion* df, hipStream_t s);
void launch_reduce_n(const DevBuf& d, bool owns_pose_edges, double* red, int nred, hipStream_t s);
void launch_posediag(const DevBuf& d, hipStream_t s);      // A: chain back-substitution | landmark groups | the trial's pose-side edges (+ the LM decision)
void launch_lm_level_sync(const DevBuf& d, const LmView& lv, hipStream_t s);
void launch_lm_chi_sync(const DevBuf& d, const LmView& lv, hipStream_t s);
void launch_lambda_init_n(const DevBuf& d, const LmView& lv, const LmParams& lp, double* red, int iteration, int nred, hipStream_t s);
void launch_decide_n(const DevBuf& d, const LmParams& lp, double* red, int nred, Mailbox* mail, unsigned long long seq, hipStream_t s);
void launch_gate(const DevBuf& d, int state, double thresh, hipStream_t s);
void launch_cull(const DevBuf& d, int state, double thresh, uint8_t* bad, hipStream_t s);   // per-observation culling flags
void launch_depth(const DevBuf& d, int state, uint8_t* out, hipStream_t s);
int  edge_blocks(const DevBuf& d);

// dense
// chain-variable elimination ahead of the dense factorisation (plba_chain.hip)
constexpr int NINV_MAX_T = 32;   // explicit-inverse back-substitution up to this many 32-wide block steps (longer sums would outlast the pivot sweep)
constexpr int CHAIN_SEG = 8;      // at most this many chain blocks between two separators (one workgroup eliminates a segment)
constexpr int CHAIN_NSLOT = 15;   // dense columns a keyframe position can own: 6 pose + 9 separator chain dimensions
struct ChainView {
    int nel, nseg, npos, Pd, Pdpad, Wld;   // eliminated chain blocks, segments, chain-block positions; dense dims (+ padded); ld of W
    const int32_t* cidx;          // (nel + 1) x 9 system indices of the eliminated blocks' dims, -1 = padding
    const int32_t* epos;          // nel: position (keyframe order) of each eliminated block
    const int32_t* seg_start;     // nseg + 1: eliminated blocks of each segment
    const int32_t* seg_col;       // 2 x nseg: dense column window [lo, hi) that can couple to the segment
    const int32_t* pidx;          // Pd system indices of the dense dims (ordered by position: pose dims, then separator chain dims)
    const int32_t* ppos;          // Pd: position of the keyframe each dense dim belongs to
    const int32_t* pslot;         // Pd: its slot (0-5 pose, 6-14 separator chain dims)
    const int32_t* slotcol;       // npos x 15: dense column of each slot, -1 = none
    const int32_t* trow;          // 2 x Pdpad/32: rows [lo, hi) of W whose segments' column windows meet each 32-column block of the dense system
    const int32_t* kfpos;         // K: chain-block position of each keyframe, -1 = no free dims
    const int32_t* ekf;           // nel: keyframe of each eliminated block (its segment's workgroup applies that keyframe's step)
    const int32_t* ukf;           // nukf keyframes whose whole step sits in the dense solution (separators, fixed): workgroup 0 applies it
    int nukf;
    const int32_t* esrc;          // what chain_elim_segment stages, as ONE level of indices: a fixed region per segment, CHAIN_SEG x (162 entries of C | 3 x 15 x 9
                                  // of B | 9 of the right-hand side); code = offset into the pose-side system (bit 30: on its diagonal), -1 = 0, -2 = 1, -3 = skip
    const int32_t* esrc_off;      // (unused)
    const int32_t* bkf;           // chain_back_segment's keyframe descriptors, one level instead of four: (nseg x CHAIN_SEG + nukf) x 20 ints
                                  // [kf, off_pvr, off_bias, dense columns of the 6 pose dims, of the 9 chain dims (separators only), pad]
    const int32_t* imu_loc;       // M x 4: [segment of the edge's keyframes (-1: none eliminated), descriptor (bkf index) of keyframe i, of keyframe j, pad]; null: off
    double* W;                    // (nel * 9 + 4) x Wld:  L^-1 [B | b_c], column Pd = w_b; zero outside each segment's window
    double* Ldinv;                // nel x 81: L_ii^-1, row-major
    double* Lsub;                 // nel x 81: L_{i+1,i}
};
// banded twisted solve of the compact dense system (plba_band.hip)
constexpr int BAND_HB = 3;        // sub-diagonal 32 x 32 tiles of the band it supports
constexpr int TWIN_MAX_TILES = 64;   // longest system (32-column tiles) the two-ended multi-launch factorisation takes (its explicit inverse costs 2 Pd^2
                                     // doubles); longer ones go to the in-LDS sweep.  configs[4] (44 tiles): twin 0.80 ms / iteration, in-LDS sweep 0.95
constexpr int BAND_MIN_TILES = 24;   // shortest system (in 32-column tiles) the two-ended sweep is used for: below, one launch per tile is faster
struct BandView {
    int T, nA, nB;                // tiles; tiles eliminated by the top-down / bottom-up sweep (the middle block has BAND_HB tiles)
    double* Lband;                // 2 x T x 4 x 1024: per direction and column block: L(k,k)^-1 and the three panels below it
    double* y;                    // 2 x Pdpad: forward-substituted right-hand sides
    double* mid;                  // 2 x (9 x 1024 + 96): the two sweeps' middle windows and right-hand sides
};
size_t band_lds_bytes(int Pdpad);
void launch_band_solve(const DevBuf& dd, const BandView& bv, hipStream_t s);      // dd.sys (+ rhs row) -> dd.x
void launch_chain_elim(const DevBuf& d, const ChainView& cv, hipStream_t s);
void launch_chain_schur(const DevBuf& d, const ChainView& cv, const DevBuf& dd, hipStream_t s, bool products_done = false);   // writes dd.sys; products_done: W^T W is in dd.wtw (the gather launch formed it)
// one workgroup of a list-driven block step: block row / column, identity row (-1: a tile of the factorisation proper), flags
struct TwinTile { int16_t r, c, aj, flags, k, pad; };      // k: the pivot tile of the step this workgroup belongs to      // flags: 1 = writes d.alt instead of d.sys, 2 = no look-ahead on this tile, 4 = c is the step's first trailing column (stores the finished panel block),
                                                   // 8 = add d.alt's tile to the old value first, 16 = this (diagonal) tile is the next pivot: factor it here,
                                                   // 32 = with 1 / 8: d.alt2 instead of d.alt (second stage of a nested plan)
struct TwinView {
    int T, m0, nchains, nlaunch;   // tiles; first tile of the FINAL dense block (ordinary steps from there); chains; launches of the chain stages
    const TwinTile* list;          // all launches' tiles: launch t = [off[t], off[t + 1])
    std::vector<int> off;          // host side
};
void launch_twin_cholesky(const DevBuf& d, const TwinView& tv, hipStream_t s);      // sys (permuted; every chain's first tile factored by the producer) -> Lfac, Ninv
void launch_cholesky(const DevBuf& d, bool use_mfma, int epoch, hipStream_t s, bool tile0_done = false);   // sys -> Lfac (lower) incl. the augmented rows;
                                                                                     // tile0_done: the producer of sys already factored tile (0,0) (k_chain_schur)
bool chain_schur_factors_tile0(const DevBuf& dd);
void launch_trsv_back(const DevBuf& d, bool use_mfma, int epoch, hipStream_t s);      // x = L^-T y; epoch must differ from the previous call's
void launch_ata(const double* A_colmajor, int rows, int cols, double* out_rowmajor, int ldo, hipStream_t s);  // A^T A

// marginalization
int marginalize_device(struct plba_problem* p, int first_kf, int max_edges, plba_prior* out);
int marginalize_factors_device(struct plba_problem* p, const std::vector<int>& imu_edges, const std::vector<int>& pt_edges,
                               const std::vector<int>& ln_edges, bool use_prior, const std::vector<int>& drop_vid, plba_prior* out);
}  // namespace plba
