// plba_kernels.hip — hand-written gfx950 kernels of the sparse part of one LM iteration.
//
//   K1-K4  k_linearize      one lane per observation: residual + compact 128-byte linearisation record + Huber weight,
//                            keyframe camera blocks (Rcb*Rwb^T, Pwb) staged in LDS, coalesced SoA observation loads;
//                            extra blocks of the same launch evaluate the IMU PVR + bias edges (pose_edge_block) and
//                            the marginalization prior edge (prior_block).  k_pose_edges / k_prior: stand-alone forms.
//   K5     k_landmark_hll   per-landmark reduction of Jl^T w Jl / Jl^T w e over its contiguous edge range (fixed
//                            order, no atomics); when lambda is known it also forms (Hll + lambda I)^-1 and extra
//                            blocks assemble the pose-side system (assemble_part).  k_landmark_dinv / k_assemble:
//                            stand-alone forms for the first iteration and for retries.
//   K6     k_schur_pairs    one workgroup per <= 256-entry chunk of a co-observing keyframe pair: sum of g_i Q g_j^T,
//                            DPP wave reduction, last-arriver fold in chunk order, exclusive block writes
//   K8     k_backsub        landmark back-substitution + landmark / keyframe update into the trial buffers
//          k_lambda_init (first iteration only), k_decide   chi2 reductions, LM control block, mailbox to the host
//          k_reduce, k_tri_pack, k_posediag*               sharded runs: partial sums / packed exchange buffer
//
// g2o semantics reproduced: SURVEY.md Appendix A; reference formulas: see plba_math.h.
#include "plba_internal.h"
#include "plba_chain_dev.h"

namespace plba {

#define DEV __device__ __forceinline__

// -------------------------------------------------------------------------------------------------
// reductions (wave = 64 lanes)
// -------------------------------------------------------------------------------------------------
DEV double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
DEV double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    return v;
}
// deterministic block sum for 256-thread blocks; result valid on thread 0
DEV double block_sum_256(double v, double* s4) {
    v = wave_sum(v);
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) s4[w] = v;
    __syncthreads();
    double r = 0;
    if (threadIdx.x == 0) r = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    __syncthreads();
    return r;
}
DEV double block_max_256(double v, double* s4) {
    v = wave_max(v);
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) s4[w] = v;
    __syncthreads();
    double r = 0;
    if (threadIdx.x == 0) r = fmax(fmax(s4[0], s4[1]), fmax(s4[2], s4[3]));
    __syncthreads();
    return r;
}
// Landmark-parallel kernels (one lane per landmark walking its 2-8 observations: dependent gathers) run in one-wave
// workgroups: four times as many workgroups as 256-thread blocks, so all CUs carry some of the latency.
constexpr int LMB = 256;           // threads of a landmark-parallel workgroup (workgroup dispatch is not free: 3000 one-wave workgroups
                                   // arrive spread over ~2 us; the camera-block staging is shared by the four waves)
constexpr int LMG = 8;             // lanes that share one landmark's edge list (a landmark has ~4, at most a few dozen observations: the
                                   // kernels cost dependent gather rounds per edge, so a quad walks the list four edges at a time)
constexpr int LML = LMB / LMG;     // landmarks per workgroup
constexpr int LMW = LMB / 64;      // waves per workgroup
DEV int pmap(int r) { return r < 3 ? r : r + 3; }   // (dp, dphi) -> position inside the 9-dim PVR block

template <int CTRL, int ROW_MASK>
DEV double dpp_get(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
DEV double wave_sum_dpp(double v) {      // row_shr:1,2,4,8 (inclusive scan inside each 16-lane row), row_bcast:15, row_bcast:31
    v += dpp_get<0x111, 0xf>(v);
    v += dpp_get<0x112, 0xf>(v);
    v += dpp_get<0x114, 0xf>(v);
    v += dpp_get<0x118, 0xf>(v);
    v += dpp_get<0x142, 0xa>(v);
    v += dpp_get<0x143, 0xc>(v);
    return v;
}

// sum over the LMG lanes that share a landmark, result in all of them: quad_perm [1,0,3,2], [2,3,0,1], then (LMG == 8)
// row_half_mirror, which pairs every lane with one of the other quad of its 8-lane group
DEV double quad_sum(double v) {
    v += dpp_get<0xB1, 0xf>(v);
    v += dpp_get<0x4E, 0xf>(v);
    if (LMG == 8) v += dpp_get<0x141, 0xf>(v);
    return v;
}
DEV int quad_sum_i(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);
    if (LMG == 8) v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);
    return v;
}


// -------------------------------------------------------------------------------------------------
// K1/K2: per-observation residual / Jacobian / robust weight
// -------------------------------------------------------------------------------------------------
struct DecideArgs;
// arrive != null (the trial pass with Jacobians): the block counts itself in for the LM decision as soon as its chi2 is out, and goes on
struct LeadWait;
struct LocalStates;
template <bool JAC, int NT> DEV void pose_edge_block(const DevBuf& d, int state, const Robust& rb, int m, int lane, const DecideArgs* arrive = nullptr, int nblk_edges = 0, double* s4 = nullptr, const LeadWait* lw = nullptr, const LocalStates* loc = nullptr);
template <bool JAC> DEV void prior_block(const DevBuf& d, int state, const DecideArgs* arrive = nullptr, int nblk_edges = 0, double* s4a = nullptr, const LeadWait* lw = nullptr);
// End of a trial folded into the trial-error launch (one GPU): the workgroup that finishes last takes the LM decision
// (decide_body = what k_decide does), so the decision is out one launch earlier.  The trial-error launch is normally the
// JAC = true instance: steps are accepted far more often than not, so the trial state is linearised in the same pass that
// measures it — into the idle record table and accumulators, swapped in on acceptance — and the errors-only pass (13 us at
// configs[2]) leaves the iteration; a rejected trial costs nothing extra, its records are simply never swapped in.
struct DecideArgs { LmParams lp; double* red; Mailbox* mail; unsigned long long seq; int nblk_lm; int fuse; };
DEV void decide_body(const DevBuf& d, const LmParams& lp, double* red, int fused, int nblk_edges, int nblk_lm, Mailbox* mail, unsigned long long seq, double* s4, bool coherent);
DEV void publish(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }     // sc1: visible to a same-launch reader on another XCD
DEV double fetch(const double* p, bool coherent) { return coherent ? __hip_atomic_load(const_cast<double*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p; }
// In-launch hand-off (k_lm_trial): the pose-side blocks of the trial need the keyframe states the chain back-substitution segments of the
// SAME launch produce.  The segments have the lowest block indices — they are resident before any waiter can be — publish their
// keyframes (sc1 stores), drain, and count themselves in on a monotonic counter; a waiter polls it (bounded: a bound that is hit is
// reported through Ctrl::sync_fail and fails the call, it never hangs the queue) and then reads the states with sc1 loads.
struct LeadWait { const unsigned* cnt; unsigned target; int* fail; };
DEV void lead_done(unsigned* cnt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
DEV void lead_wait(const LeadWait& w) {
    if (threadIdx.x == 0 && w.cnt) {
        int n = 0;
        while ((int)(__hip_atomic_load(const_cast<unsigned*>(w.cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - w.target) < 0) {      // (wrap-safe)
            __builtin_amdgcn_s_sleep(16);
            if (++n > (1 << 20)) { __hip_atomic_store(w.fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
    }
    __syncthreads();
}

// Every workgroup of the trial-error launch ends here: its results have gone out with sc1 stores; once they are drained it
// counts itself, and the last one to arrive reads everybody's (sc1 loads) and decides.
// Ordering: the hand-off is TARGETED, not a fence.  The few words that cross workgroups are written with agent-scope atomic
// stores (write-through past the XCD's L2), `s_waitcnt vmcnt(0)` waits for their acknowledgement, the barrier orders thread 0's
// counter increment (an agent-scope RMW, performed at the same coherence point) after them, and the last arriver reads the
// words back with agent-scope atomic loads.  An acq_rel counter instead (the generic release -> flag -> acquire pattern) was
// measured in round 3: a release at agent scope is `buffer_wbl2 sc1` — it writes back EVERY dirty line of the XCD's L2, i.e.
// the records / blocks this very launch is producing — once per workgroup: k_linearize<true> 18.3 -> 26.2 us, k_schur_pairs
// 29.7 -> 49.7 us (profiles/r03_acqrel_kernel_stats.csv), 5.6 k -> 4.1 k iterations/s.  Relaxed + targeted it stays.
DEV void trial_arrive(const DevBuf& d, const DecideArgs& da, int nblk_edges, double* s4) {
    __shared__ int s_lastblk;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const int prev = __hip_atomic_fetch_add(d.trial_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_lastblk = (prev == (int)gridDim.x - 1) ? 1 : 0;
        if (s_lastblk) __hip_atomic_store(d.trial_cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
    }
    __syncthreads();
    if (!s_lastblk) return;
    decide_body(d, da.lp, da.red, 1, nblk_edges, da.nblk_lm, da.mail, da.seq, s4, true);
}

// Blocks [0, nblk_edges) handle observations.  Blocks beyond evaluate the pose-side edges inside the SAME launch
// (one IMU PVR+bias edge pair per block, then one block for the prior), so the serial per-edge IMU math overlaps the
// observation pass instead of following it.
template <bool JAC>
// spec != 0: launched BEFORE the host knows the LM decision of the trial that just ran (so that the host's reaction time
// hides behind this kernel): linearises the trial state iff the device-side decision was "accepted", else does nothing.
__global__ __launch_bounds__(256) void k_linearize(DevBuf d, int state, Robust rb, int nblk_edges, int spec, DecideArgs da) {
    if (spec && !d.ctrl->accepted) return;
    extern __shared__ double s_dyn[];
    double* s_kc = s_dyn;               // K x 12 staged camera blocks
    __shared__ double s4[4];
    if ((int)blockIdx.x >= nblk_edges) {
        const int m = blockIdx.x - nblk_edges;
#ifdef PLBA_STAMPS_LM
        const unsigned long long t0 = __builtin_readcyclecounter();
#endif
        const DecideArgs* early = nullptr;      // (arriving right after chi2 was measured slower: the deciding block is then an IMU block whose Jacobians wait for its own decision)
        if (m < d.M) pose_edge_block<JAC, 256>(d, state, rb, m, threadIdx.x, early, nblk_edges, s4);
        else prior_block<JAC>(d, state, early, nblk_edges, s4);
#ifdef PLBA_STAMPS_LM
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (JAC && threadIdx.x == 0 && (m == 0 || m == d.M / 2)) { d.dbgbuf[m == 0 ? 32 : 33] = (double)(__builtin_readcyclecounter() - t0); d.dbgbuf[m == 0 ? 34 : 35] = (double)(long long)__builtin_amdgcn_s_memrealtime(); }
#endif
        if (da.fuse) trial_arrive(d, da, nblk_edges, s4);
        return;
    }
#ifdef PLBA_STAMPS_LM
    const unsigned long long t0 = __builtin_readcyclecounter();
#endif
    const double* kf = d.kf[state];
    for (int k = threadIdx.x; k < d.K; k += 256) kfcam_make(d.cam, kf + (size_t)k * KF_STRIDE, s_kc + k * KFCAM_STRIDE);
    __syncthreads();
    const int e = blockIdx.x * 256 + threadIdx.x;
    double rho = 0.0;
    if (e < d.E) {
        if (d.ob_level[e] == 0) {
            const int k = d.ob_kf[e];
            const int slot = d.ob_slot[e];
            const double w0 = d.ob_w[e];
            const double* L = d.lm[state] + (size_t)slot * 6;
            const double* kc = s_kc + k * KFCAM_STRIDE;
            double e2[2], rec[12];
            bool dpos;
            int kind;
            V3 Pc = v3(0, 0, 0);
            if (e < d.Ep) {
                kind = PLBA_EDGE_POINT;
                const double2 uv = reinterpret_cast<const double2*>(d.po_uv)[e];
                point_edge_rec(d.cam, kc, v3(L[0], L[1], L[2]), uv.x, uv.y, e2, rec, dpos, false, &Pc);      // the record is Pc itself
            } else {
                kind = PLBA_EDGE_LINE;
                const double* l = d.lo_l + (size_t)(e - d.Ep) * 3;
                line_edge_rec(d.cam, kc, v3(L[0], L[1], L[2]), v3(L[3], L[4], L[5]), l[0], l[1], l[2], e2, rec, dpos, JAC);
            }
            const double chi = w0 * (e2[0] * e2[0] + e2[1] * e2[1]);
            double r0 = chi, r1 = 1.0;
            if (rb.on[kind]) huber(chi, rb.delta[kind], r0, r1);
            rho = r0;
            d.ob_chi2[e] = chi;
            if (JAC) {   // at the observation's keyframe-major position: half a 128-byte line for a point, a full one for a line
                double4* out = reinterpret_cast<double4*>(d.erec + (size_t)d.ob_pos[e] * EREC_UNIT);
                if (e < d.Ep) {
                    out[0] = make_double4(Pc.x, Pc.y, Pc.z, w0 * r1);
                    out[1] = make_double4(e2[0], e2[1], chi, 0.0);
                } else {
                    out[0] = make_double4(rec[0], rec[1], rec[2], rec[3]);
                    out[1] = make_double4(rec[4], rec[5], rec[6], rec[7]);
                    out[2] = make_double4(rec[8], rec[9], rec[10], rec[11]);
                    out[3] = make_double4(w0 * r1, e2[0], e2[1], chi);
                }
            }
        } else if (JAC) {
            double4* out = reinterpret_cast<double4*>(d.erec + (size_t)d.ob_pos[e] * EREC_UNIT);
            const double4 z = make_double4(0, 0, 0, 0);
            out[0] = z; out[1] = z;
            if (e >= d.Ep) { out[2] = z; out[3] = z; }
        }
    }
    double bs = block_sum_256(rho, s4);
    if (threadIdx.x == 0) { if (da.fuse) publish(&d.chi_part[blockIdx.x], bs); else d.chi_part[blockIdx.x] = bs; }
    if (da.fuse) { trial_arrive(d, da, nblk_edges, s4); return; }
#ifdef PLBA_STAMPS_LM
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (JAC && threadIdx.x == 0 && ((int)blockIdx.x == 0 || (int)blockIdx.x == nblk_edges / 2 || (int)blockIdx.x == nblk_edges - 1)) {
        const int o = blockIdx.x == 0 ? 40 : (int)blockIdx.x == nblk_edges / 2 ? 42 : 44;
        d.dbgbuf[o] = (double)(__builtin_readcyclecounter() - t0); d.dbgbuf[o + 1] = (double)(long long)__builtin_amdgcn_s_memrealtime();
    }
#endif
}

// -------------------------------------------------------------------------------------------------
// K5: landmark blocks.  Thread per landmark slot, fixed edge order (deterministic).
// -------------------------------------------------------------------------------------------------
// sys = Himu + Hconst (+ lambda I on the real diagonal, 1 on the padded diagonal); rows Ppad, Ppad+1 = pose-side
// gradient; clears the idle IMU accumulator.  Grid-stride over `nblocks` blocks of `nthreads` threads.
DEV void assemble_part(const DevBuf& d, int add_lambda, int bid, int nblocks, int tid, int nthreads) {
    const double lambda = d.ctrl->lambda;
    if (d.alist) {
        // structural version (d.alist, built at upload when the chain elimination is on: then nothing but this pass and
        // k_schur_pairs ever writes sys): only the entries that can be non-zero — IMU / prior blocks, the co-observing
        // keyframe pairs' blocks, the diagonal — are rebuilt; the rest of sys and of the idle accumulator stays zero.
        // A tenth of the 19 MB the full pass moves per iteration.
        const int n = d.nalist;
        const size_t stride = (size_t)nblocks * nthreads;
        for (size_t k = (size_t)bid * nthreads + tid; k < (size_t)n + 2 * (size_t)d.ld; k += stride) {
            if (k < (size_t)n) {
                const int idx = d.alist[k];
                const int r = idx / d.ld, c = idx - r * d.ld;
                double v = d.Himu[idx] + d.Hconst[idx];
                if (r == c && add_lambda) v += (r < d.P) ? lambda : 1.0;
                d.Himu_alt[idx] = 0.0;
                d.sys[idx] = v;
            } else {
                const int q = (int)(k - n), row = q / d.ld, c = q - row * d.ld;
                const double v = d.bimu[c] + d.bprior[c];
                if (row == 0) { d.bpg[c] = v; d.bimu_alt[c] = 0.0; }
                d.sys[(size_t)(d.Ppad + row) * d.ld + c] = v;
            }
        }
        return;
    }
    const size_t n = (size_t)(d.Ppad + TILE) * d.ld;
    for (size_t idx = (size_t)bid * nthreads + tid; idx < n; idx += (size_t)nblocks * nthreads) {
        const int r = (int)(idx / d.ld), c = (int)(idx % d.ld);
        double v = 0.0;
        if (r < d.Ppad) {
            v = d.Himu[idx] + d.Hconst[idx];      // IMU edges of this iteration + the constant prior J0^T J0
            if (r == c && add_lambda) v += (r < d.P) ? lambda : 1.0;
            d.Himu_alt[idx] = 0.0;                // the accumulator the NEXT outer iteration's pose-side edges add into
        } else if (r <= d.Ppad + 1) {
            v = d.bimu[c] + d.bprior[c];
            if (r == d.Ppad) { d.bpg[c] = v; d.bimu_alt[c] = 0.0; }
        }
        d.sys[idx] = v;
    }
}

// (Hll + lambda I)^-1 and D*bl of one landmark slot from its undamped blocks
DEV void landmark_dinv_one(const DevBuf& d, int slot, const double* h, const double* b, bool active, bool is_pt, double lambda) {
    double dd[12], tt[6];
#pragma unroll
    for (int i = 0; i < 12; ++i) dd[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) tt[i] = 0.0;
    if (active) {
        sym3_inv(h, lambda, dd);
        const V3 t0 = sym3_mul(dd, v3(b[0], b[1], b[2]));
        tt[0] = t0.x; tt[1] = t0.y; tt[2] = t0.z;
        if (!is_pt) {
            sym3_inv(h + 6, lambda, dd + 6);
            const V3 t1 = sym3_mul(dd + 6, v3(b[3], b[4], b[5]));
            tt[3] = t1.x; tt[4] = t1.y; tt[5] = t1.z;
        }
    }
    double* D = d.dinv + (size_t)slot * 12;
    double* t = d.tv + (size_t)slot * 6;
#pragma unroll
    for (int i = 0; i < 12; ++i) D[i] = dd[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) t[i] = tt[i];
}

// FUSE_DINV: lambda of this iteration is already known (every outer iteration but the first), so the damped inverse
// is formed right here from the registers instead of by a second pass over hll/bl.
// Blocks beyond nblk_lm (only launched together with FUSE_DINV, i.e. when lambda is already known) assemble the
// pose-side part of the reduced system in the shadow of the landmark pass instead of in a launch of their own.
template <bool FUSE_DINV>
__global__ __launch_bounds__(LMB) void k_landmark_hll(DevBuf d, int state, int nblk_lm, int add_lambda, int spec) {
    if (spec && !d.ctrl->accepted) return;      // enqueued behind the deciding launch: runs only for the state that was accepted
    extern __shared__ double s_dyn[];
    double* s_kc = s_dyn;
    if ((int)blockIdx.x >= nblk_lm) { assemble_part(d, add_lambda, blockIdx.x - nblk_lm, gridDim.x - nblk_lm, threadIdx.x, LMB); return; }
    // What this kernel costs is dependent memory round trips, so they are laid out explicitly: (1) the landmark's edge
    // range, its flags and lambda together with the keyframe states of the camera-block staging, (2) the edge's
    // indices, (3) its record; the indices of the group's next edge are fetched while the current one is processed.
#ifdef PLBA_STAMPS_LM
    unsigned long long ts[6] = {0, 0, 0, 0, 0, 0};
#define LMSTAMP(i) do { ts[i] = __builtin_readcyclecounter(); } while (0)
#else
#define LMSTAMP(i) do {} while (0)
#endif
    LMSTAMP(0);
    const int slot = blockIdx.x * LML + (threadIdx.x / LMG), sub = threadIdx.x % LMG;
    const bool valid = slot < d.L;          // uniform over the lanes of a landmark
    const int s = valid ? d.lm_start[slot] : 0, en = valid ? d.lm_start[slot + 1] : 0;
    const bool fixed = valid ? d.lm_fixed[slot] != 0 : true;
    const double lambda = FUSE_DINV ? d.ctrl->lambda : 0.0;
    for (int k = threadIdx.x; k < d.K; k += LMB) kfcam_make(d.cam, d.kf[state] + (size_t)k * KF_STRIDE, s_kc + k * KFCAM_STRIDE);
    int ed = s + sub;
    int lvl = 0, pos = 0, kfi = 0;
    if (ed < en) { lvl = d.ob_level[ed]; pos = d.ob_pos[ed]; kfi = d.ob_kf[ed]; }
    __syncthreads();
    LMSTAMP(1);
    double md = 0.0;
    if (valid) {
        double h[12], b[6];
#pragma unroll
        for (int i = 0; i < 12; ++i) h[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) b[i] = 0.0;
        int nact = 0;
        const bool is_pt = slot < d.Np;
        while (ed < en) {      // lane `sub` of the group takes every LMG-th edge
            const double4* r4 = reinterpret_cast<const double4*>(d.erec + (size_t)pos * EREC_UNIT);
            const double4 q0 = r4[0], q1 = r4[1];
            double4 q2 = make_double4(0, 0, 0, 0), q3 = q2;
            if (!is_pt) { q2 = r4[2]; q3 = r4[3]; }
            const int kf_now = kfi;
            if (lvl == 0) ++nact;
            ed += LMG;
            if (ed < en) { lvl = d.ob_level[ed]; pos = d.ob_pos[ed]; kfi = d.ob_kf[ed]; }
            const double w = is_pt ? q0.w : q3.x;
            if (w == 0.0) continue;
            const double* kc = s_kc + kf_now * KFCAM_STRIDE;
            M3 M;
#pragma unroll
            for (int i = 0; i < 9; ++i) M.a[i] = kc[i];
            V3 ua = v3(q0.x, q0.y, q0.z), ub = v3(q1.z, q1.w, q2.x);
            if (is_pt) { V3 Pp; point_rows_from_Pc(d.cam, v3(q0.x, q0.y, q0.z), ua, ub, Pp); }
            const V3 va = mulT(M, ua), vb = mulT(M, ub);   // M^T uA, M^T uB
            const double e0 = is_pt ? q1.x : q3.y, e1 = is_pt ? q1.y : q3.z;
            if (is_pt) {   // Jl rows = -va^T, -vb^T on the same 3 coordinates
                h[0] += w * (va.x * va.x + vb.x * vb.x); h[1] += w * (va.x * va.y + vb.x * vb.y); h[2] += w * (va.x * va.z + vb.x * vb.z);
                h[3] += w * (va.y * va.y + vb.y * vb.y); h[4] += w * (va.y * va.z + vb.y * vb.z); h[5] += w * (va.z * va.z + vb.z * vb.z);
                b[0] += w * (va.x * e0 + vb.x * e1); b[1] += w * (va.y * e0 + vb.y * e1); b[2] += w * (va.z * e0 + vb.z * e1);
            } else {       // row0 = +va^T on sP, row1 = +vb^T on eP: block-diagonal 6x6
                h[0] += w * va.x * va.x; h[1] += w * va.x * va.y; h[2] += w * va.x * va.z; h[3] += w * va.y * va.y; h[4] += w * va.y * va.z; h[5] += w * va.z * va.z;
                h[6] += w * vb.x * vb.x; h[7] += w * vb.x * vb.y; h[8] += w * vb.x * vb.z; h[9] += w * vb.y * vb.y; h[10] += w * vb.y * vb.z; h[11] += w * vb.z * vb.z;
                b[0] -= w * va.x * e0; b[1] -= w * va.y * e0; b[2] -= w * va.z * e0;
                b[3] -= w * vb.x * e1; b[4] -= w * vb.y * e1; b[5] -= w * vb.z * e1;
            }
        }
        LMSTAMP(2);
        // the quad's partial blocks, added in a fixed order (deterministic); every lane ends up with the sums
#pragma unroll
        for (int i = 0; i < 6; ++i) h[i] = quad_sum(h[i]);
#pragma unroll
        for (int i = 0; i < 3; ++i) b[i] = quad_sum(b[i]);
        if (!is_pt) {       // the second 3 x 3 block exists for lines only (uniform over the group, and over all but one wave)
#pragma unroll
            for (int i = 6; i < 12; ++i) h[i] = quad_sum(h[i]);
#pragma unroll
            for (int i = 3; i < 6; ++i) b[i] = quad_sum(b[i]);
        }
        nact = quad_sum_i(nact);
        const bool active = (nact > 0) && !fixed;
        if (sub == 0) {
            d.lm_active[slot] = active ? 1 : 0;
            double* ho = d.hll + (size_t)slot * 12;
            double* bo = d.bl + (size_t)slot * 6;
#pragma unroll
            for (int i = 0; i < 12; ++i) ho[i] = h[i];
#pragma unroll
            for (int i = 0; i < 6; ++i) bo[i] = b[i];
            if (active) {
                md = fmax(fmax(fabs(h[0]), fabs(h[3])), fabs(h[5]));
                if (!is_pt) md = fmax(md, fmax(fmax(fabs(h[6]), fabs(h[9])), fabs(h[11])));
            }
            if (FUSE_DINV) landmark_dinv_one(d, slot, h, b, active, is_pt, lambda);
        }
    }
    LMSTAMP(3);
    const double bm = wave_max(md);      // one partial per workgroup (the control kernel reads them all in one workgroup)
    __shared__ double s_w[LMW];
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = bm;
    __syncthreads();
    if (threadIdx.x == 0) { double v = s_w[0]; for (int q = 1; q < LMW; ++q) v = fmax(v, s_w[q]); d.maxd_part[blockIdx.x] = v; }
#ifdef PLBA_STAMPS_LM
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LMSTAMP(4);
    if (FUSE_DINV && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == nblk_lm / 2 || blockIdx.x == nblk_lm - 1)) {
        const int o = blockIdx.x == 0 ? 0 : blockIdx.x == nblk_lm / 2 ? 8 : 16;
        for (int q = 0; q < 5; ++q) d.dbgbuf[o + q] = (double)(ts[q] - ts[0]);
        d.dbgbuf[o + 5] = (double)(long long)__builtin_amdgcn_s_memrealtime();
        d.dbgbuf[o + 6] = (double)(en - s);
    }
#endif
}

__global__ __launch_bounds__(LMB) void k_landmark_dinv(DevBuf d) {
    const int slot = blockIdx.x * LMB + threadIdx.x;
    if (slot >= d.L) return;
    double h[12], b[6];
#pragma unroll
    for (int i = 0; i < 12; ++i) h[i] = d.hll[(size_t)slot * 12 + i];
#pragma unroll
    for (int i = 0; i < 6; ++i) b[i] = d.bl[(size_t)slot * 6 + i];
    landmark_dinv_one(d, slot, h, b, d.lm_active[slot] != 0, slot < d.Np, d.ctrl->lambda);
}

struct EdgeRows { V3 va, vb; double ga[6], gb[6], w, e0, e1; };
DEV EdgeRows load_rows(const DevBuf& d, const double* rec, const double* kc, bool is_pt) {
    const double4* r4 = reinterpret_cast<const double4*>(rec);
    const double4 q0 = r4[0], q1 = r4[1];
    EdgeRows o;
    if (is_pt) {      // 64-byte point record: camera-frame point, weight, errors
        V3 ua, ub, P;
        point_rows_from_Pc(d.cam, v3(q0.x, q0.y, q0.z), ua, ub, P);
        rec_row(true, d.fix_q1 != 0, kc, d.cam.Rcb, ua, P, o.va, o.ga);
        rec_row(true, d.fix_q1 != 0, kc, d.cam.Rcb, ub, P, o.vb, o.gb);
        o.w = q0.w; o.e0 = q1.x; o.e1 = q1.y;
        return o;
    }
    const double4 q2 = r4[2], q3 = r4[3];
    rec_row(false, d.fix_q1 != 0, kc, d.cam.Rcb, v3(q0.x, q0.y, q0.z), v3(q0.w, q1.x, q1.y), o.va, o.ga);
    rec_row(false, d.fix_q1 != 0, kc, d.cam.Rcb, v3(q1.z, q1.w, q2.x), v3(q2.y, q2.z, q2.w), o.vb, o.gb);
    o.w = q3.x; o.e0 = q3.y; o.e1 = q3.z;
    return o;
}

// per-keyframe diagonal of sum Jp^T w Jp (only needed for lambda_init at iteration 0)
// One workgroup per <= 256-entry chunk of a DIAGONAL pair (the other chunks leave at once), partial sums combined by the
// pair's last chunk in chunk order: the same descriptors, partial-sum slots and arrival counters as k_schur_pairs, which
// never runs at the same time.  (One workgroup per keyframe walked up to 2500 entries in ten dependent rounds: 21 us.)
__global__ __launch_bounds__(256) void k_kfdiag(DevBuf d, int state) {
    __shared__ double s4[4];
    __shared__ double s_kc[KFCAM_STRIDE];
    __shared__ double s_v[6];
    __shared__ int s_lastc;
    const ChunkMeta m = d.ch_meta[blockIdx.x];
    const int i = m.ij & 0xffff, j = (m.ij >> 16) & 0xffff;
    if (i != j) return;
    if (threadIdx.x == 0) kfcam_make(d.cam, d.kf[state] + (size_t)i * KF_STRIDE, s_kc);
    __syncthreads();
    double acc[6] = {0, 0, 0, 0, 0, 0};
    const int n = m.start + threadIdx.x;
    if (n < m.end) {
        const bool is_pt = d.ent_slot[n] < d.Np;
        const EdgeRows r = load_rows(d, d.erec + (size_t)d.ent_pi[n] * EREC_UNIT, s_kc, is_pt);
        double ja[6], jb[6];
        basis_apply(d.cam.Rcb, r.ga, ja);
        basis_apply(d.cam.Rcb, r.gb, jb);
#pragma unroll
        for (int c = 0; c < 6; ++c) acc[c] = (r.w != 0.0) ? r.w * (ja[c] * ja[c] + jb[c] * jb[c]) : 0.0;      // an inactive point's record is all zeros: its rows are not finite
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const double v = block_sum_256(acc[c], s4);
        if (threadIdx.x == 0) s_v[c] = v;
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (m.nch > 1) {
        if (t < 6) __hip_atomic_store(&d.schur_part[(size_t)m.slot * 48 + t], s_v[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) s_lastc = (__hip_atomic_fetch_add(&d.pair_cnt[m.ch0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == m.nch - 1) ? 1 : 0;
        __syncthreads();
        if (!s_lastc) return;
        if (t == 0) __hip_atomic_store(&d.pair_cnt[m.ch0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t < 6) {
            double sum = 0.0;
            for (int c = 0; c < m.nch; ++c) sum += __hip_atomic_load(&d.schur_part[(size_t)(m.ch0 + c) * 48 + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_v[t] = sum;
        }
    }
    if (t < 6) d.kfdiag[i * 6 + t] = s_v[t];
}

// -------------------------------------------------------------------------------------------------
// K6: Schur complement over keyframe pairs.
//   Hschur(i,j) = Hpp(i,j) - sum_l Hpl(i,l) D_l Hpl(j,l)^T  with Hpl(i,l) = Jp_i^T w_i Jl_i
//               = [pose-side edges + lambda] + C^T ( sum_entries sum_ab g_ia Q_ab g_jb^T ) C
//   Q = [ei==ej] w I2 - w_i w_j Jl_i D Jl_j^T   (2x2),   C = blkdiag(Rcb, Rcb)  (applied once per pair)
//   bschur_i   = [pose-side] + C^T sum_{e in kf i} -w (g_a (e_a + Jl_a t_l)),   t_l = D_l bl_l
// One 256-thread workgroup per pair; records are one 128-byte line each, stored keyframe-major, so both gather
// streams of a pair are ascending and dense.  Each lane accumulates 6x6 (+2x6 for diagonal pairs) in registers,
// DPP wave reduction, LDS across the 4 waves; the owning workgroup read-modify-writes its exclusive blocks of `sys`.
// -------------------------------------------------------------------------------------------------
// One workgroup per CHUNK of at most 256 entries of a keyframe pair (the heaviest pairs hold ~2500 entries: one
// workgroup per pair left the launch waiting for ten dependent gather rounds of a handful of workgroups).  A pair's
// chunks publish their 48 partial sums; the chunk that arrives last adds them up in chunk order (deterministic) and
// applies the result to the reduced system.
// Dependent memory round trips are what this kernel costs, so it is laid out as three of them: (1) one 32-byte chunk
// descriptor (scalar), (2) the entry's indices | the two keyframe states, (3) both 128-byte records, D_l, t_l and the
// old values of the output block — all issued before the staged camera blocks are needed; the arithmetic follows.
struct SchurLds {
    double s_red[4][48];
    double s_in[48], s_tmp[36];
    double s_kc[2 * KFCAM_STRIDE];
    double stA[4][32 * 12];       // per wave: [entry][ga(6) | gb(6)] of half a wave's entries (operand staging of the reduction)
    double stB[4][32 * 16];       // per wave: [entry][T0(6), -f0, -e0 | T1(6), -f1, -e1]
    int s_last;
};
// The first `nlead` workgroups eliminate one segment of the velocity / bias chain each instead (plba_chain_dev.h): that work
// only needs the assembled pose-side system, so on one GPU it runs in the shadow of the pair pass.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_schur_pairs(DevBuf d, int state, ChainView cv, int nlead) {      // 3 workgroups per CU: what the 49 KB of LDS allow
    static_assert(ELIM_THREADS == 256, "the chain segments ride in this launch");
    __shared__ __attribute__((aligned(16))) union { ChainElimLds ce; SchurLds sp; } lds;      // a workgroup is one or the other
    if ((int)blockIdx.x < nlead) { if ((int)blockIdx.x < cv.nseg) chain_elim_segment(d, cv, blockIdx.x, lds.ce); return; }     // nlead = nseg rounded up to the XCD count
    auto& s_red = lds.sp.s_red; auto& s_in = lds.sp.s_in; auto& s_tmp = lds.sp.s_tmp; auto& s_kc = lds.sp.s_kc; int& s_last = lds.sp.s_last;
    const int ch = blockIdx.x - nlead;
#ifdef PLBA_STAMPS_LM
    unsigned long long sts[6] = {0,0,0,0,0,0}; sts[0] = __builtin_readcyclecounter(); const long long rt0 = __builtin_amdgcn_s_memrealtime();
#define SSTAMP(i) sts[i] = __builtin_readcyclecounter()
#else
#define SSTAMP(i) do {} while (0)
#endif
    const ChunkMeta m = d.ch_meta[ch];
    const int p = m.ch0, i = m.ij & 0xffff, j = (m.ij >> 16) & 0xffff;      // p: the pair's arrival counter
    const bool diag = (i == j);
    const int t = threadIdx.x;
    const int n = m.start + t;
    const bool have = n < m.end;              // chunks hold at most 256 entries: one per lane
    const int ld = d.ld;
    // ---- round 2: entry indices (all lanes) | keyframe states (lanes 0, 1) -------------------------------------------------
    int pi = 0, pj = 0, slot = 0;
    if (have) { pi = d.ent_pi[n]; pj = d.ent_pj[n]; slot = d.ent_slot[n]; }
    double kfs[7];
    if (t < 2) {
        const double* s = d.kf[state] + (size_t)(t == 0 ? i : j) * KF_STRIDE;
        kfs[0] = s[0]; kfs[1] = s[1]; kfs[2] = s[2]; kfs[3] = s[6]; kfs[4] = s[7]; kfs[5] = s[8]; kfs[6] = s[9];
    }
    // old values of the output block (only the last chunk of a pair uses them; nobody else writes the block in this launch)
    double old0 = 0.0, old1 = 0.0;
    size_t a0 = 0, a1 = 0;
    if (t < 36) {
        const int r = t / 6, c = t % 6;
        a0 = (size_t)(m.oi + pmap(r)) * ld + m.oj + pmap(c);
        a1 = (size_t)(m.oj + pmap(c)) * ld + m.oi + pmap(r);
        old0 = d.sys[a0]; old1 = d.sys[a1];
    } else if (diag && t < 48) {
        const int q = t - 36;
        a0 = (size_t)((t < 42) ? d.Ppad : d.Ppad + 1) * ld + m.oi + pmap(q % 6);
        a1 = m.oi + pmap(q % 6);
        old0 = d.sys[a0];
        if (t >= 42) old1 = d.bpg[a1];
    }
    // ---- round 3: records, D_l, t_l ----------------------------------------------------------------------------------------
    const bool is_pt = slot < d.Np;
    double4 qi[4], qj[4];
    double D[12], tl[6];
    {
        const double4* ri4 = reinterpret_cast<const double4*>(d.erec + (size_t)pi * EREC_UNIT);
        const double4* rj4 = reinterpret_cast<const double4*>(d.erec + (size_t)pj * EREC_UNIT);
        const double4* D4 = reinterpret_cast<const double4*>(d.dinv + (size_t)slot * 12);
        const double2* t2 = reinterpret_cast<const double2*>(d.tv + (size_t)slot * 6);
        qi[0] = ri4[0]; qi[1] = ri4[1]; qj[0] = rj4[0]; qj[1] = rj4[1];
        const double4 z4 = make_double4(0.0, 0.0, 0.0, 0.0);
        qi[2] = z4; qi[3] = z4; qj[2] = z4; qj[3] = z4;
        if (!is_pt) { qi[2] = ri4[2]; qi[3] = ri4[3]; qj[2] = rj4[2]; qj[3] = rj4[3]; }      // point records are 64 bytes (plba_math.h)
        {   // a point's (Hll + lambda I)^-1 is one symmetric 3 x 3 (48 bytes); a line's two of them
            const double4 v0 = D4[0];
            const double2 v1 = reinterpret_cast<const double2*>(D4)[2];
            D[0] = v0.x; D[1] = v0.y; D[2] = v0.z; D[3] = v0.w; D[4] = v1.x; D[5] = v1.y;
#pragma unroll
            for (int q = 6; q < 12; ++q) D[q] = 0.0;
            if (!is_pt) {
                const double2 v2 = reinterpret_cast<const double2*>(D4)[3];
                const double4 v3 = D4[2];
                D[6] = v2.x; D[7] = v2.y; D[8] = v3.x; D[9] = v3.y; D[10] = v3.z; D[11] = v3.w;
            }
        }
        (void)t2;
    }
    if (t < 2) {
        double s[KF_STRIDE];
#pragma unroll
        for (int q = 0; q < KF_STRIDE; ++q) s[q] = 0.0;
        s[0] = kfs[0]; s[1] = kfs[1]; s[2] = kfs[2]; s[6] = kfs[3]; s[7] = kfs[4]; s[8] = kfs[5]; s[9] = kfs[6];
        kfcam_make(d.cam, s, s_kc + t * KFCAM_STRIDE);
    }
    __syncthreads();
    SSTAMP(1);
    // per-lane factors of the entry's contribution  g_i Q g_j^T = ga_i T0^T + gb_i T1^T  (zero for an idle lane), then each
    // of the 36 (+12) sums goes through the wave reduction as soon as it is formed: 6 live accumulators instead of 48
    const double wi = is_pt ? qi[0].w : qi[3].x, wj = is_pt ? qj[0].w : qj[3].x;
    const double ei0 = is_pt ? qi[1].x : qi[3].y, ei1 = is_pt ? qi[1].y : qi[3].z;
    const bool on = have && wi != 0.0 && wj != 0.0;
    double ga[6], gbv[6], T0[6], T1[6], f0 = 0.0, f1 = 0.0, e0 = 0.0, e1 = 0.0;
    {
        EdgeRows ri, rj;
        V3 uia = v3(qi[0].x, qi[0].y, qi[0].z), Pia = v3(qi[0].w, qi[1].x, qi[1].y), uib = v3(qi[1].z, qi[1].w, qi[2].x), Pib = v3(qi[2].y, qi[2].z, qi[2].w);
        V3 uja = v3(qj[0].x, qj[0].y, qj[0].z), Pja = v3(qj[0].w, qj[1].x, qj[1].y), ujb = v3(qj[1].z, qj[1].w, qj[2].x), Pjb = v3(qj[2].y, qj[2].z, qj[2].w);
        if (is_pt) {
            point_rows_from_Pc(d.cam, v3(qi[0].x, qi[0].y, qi[0].z), uia, uib, Pia); Pib = Pia;
            point_rows_from_Pc(d.cam, v3(qj[0].x, qj[0].y, qj[0].z), uja, ujb, Pja); Pjb = Pja;
        }
        rec_row(is_pt, d.fix_q1 != 0, s_kc, d.cam.Rcb, uia, Pia, ri.va, ri.ga);
        rec_row(is_pt, d.fix_q1 != 0, s_kc, d.cam.Rcb, uib, Pib, ri.vb, ri.gb);
        rec_row(is_pt, d.fix_q1 != 0, s_kc + KFCAM_STRIDE, d.cam.Rcb, uja, Pja, rj.va, rj.ga);
        rec_row(is_pt, d.fix_q1 != 0, s_kc + KFCAM_STRIDE, d.cam.Rcb, ujb, Pjb, rj.vb, rj.gb);
        double q00, q01, q10, q11;
        const double ww = wi * wj;
        if (is_pt) {
            const V3 Da = sym3_mul(D, rj.va), Db = sym3_mul(D, rj.vb);
            q00 = -ww * dot(ri.va, Da); q01 = -ww * dot(ri.va, Db);
            q10 = -ww * dot(ri.vb, Da); q11 = -ww * dot(ri.vb, Db);
        } else {
            q00 = -ww * dot(ri.va, sym3_mul(D, rj.va)); q11 = -ww * dot(ri.vb, sym3_mul(D + 6, rj.vb));
            q01 = 0.0; q10 = 0.0;
        }
        if (pi == pj) { q00 += wi; q11 += wi; }
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            T0[c] = on ? q00 * rj.ga[c] + q01 * rj.gb[c] : 0.0;
            T1[c] = on ? q10 * rj.ga[c] + q11 * rj.gb[c] : 0.0;
            ga[c] = on ? ri.ga[c] : 0.0;
            gbv[c] = on ? ri.gb[c] : 0.0;
        }
        if (diag && on) {      // diagonal pairs only (one in seven): t_l is fetched here, a round later, to keep the register
                               // footprint of every other chunk at four workgroups per CU
            const double2* t2 = reinterpret_cast<const double2*>(d.tv + (size_t)slot * 6);
#pragma unroll
            for (int q = 0; q < 3; ++q) { const double2 v = t2[q]; tl[2 * q] = v.x; tl[2 * q + 1] = v.y; }
            const double sl = is_pt ? -1.0 : 1.0;
            const V3 ta = v3(tl[0], tl[1], tl[2]);
            const V3 tb = is_pt ? ta : v3(tl[3], tl[4], tl[5]);
            e0 = wi * ei0; e1 = wi * ei1;
            f0 = wi * (ei0 + sl * dot(ri.va, ta)); f1 = wi * (ei1 + sl * dot(ri.vb, tb));
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // The 36 (+12) sums over the wave's entries are ONE small matrix product,  G (16 x 2n) * B (2n x 16)  with
    //   G[r][2e + s] = (s ? gb : ga)_e[r]  (r < 6),     B[2e + s][c] = (s ? T1 : T0)_e[c]  (c < 6),  -f_s (c = 6),  -e_s (c = 7),
    // so the matrix cores add them up (v_mfma_f64_16x16x4_f64, four k-columns = two entries per instruction) instead of
    // 48 six-step DPP reductions: the factors go through LDS once (half a wave at a time), ~200 instructions in place of
    // ~900, which is what this kernel was issuing when its gathers were not the bottleneck.  Fixed order: deterministic.
    typedef double double4v __attribute__((ext_vector_type(4)));
    double4v accE = (double4v){0.0, 0.0, 0.0, 0.0}, accO = accE;
    {
        const int li = lane & 15, lk = lane >> 4;
        double* stA = lds.sp.stA[wv];
        double* stB = lds.sp.stB[wv];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            if ((lane >> 5) == hb) {
                double2* pa = reinterpret_cast<double2*>(stA + (lane & 31) * 12);
                double2* pb = reinterpret_cast<double2*>(stB + (lane & 31) * 16);
#pragma unroll
                for (int q = 0; q < 3; ++q) { pa[q] = make_double2(ga[2 * q], ga[2 * q + 1]); pa[3 + q] = make_double2(gbv[2 * q], gbv[2 * q + 1]); }
#pragma unroll
                for (int q = 0; q < 3; ++q) { pb[q] = make_double2(T0[2 * q], T0[2 * q + 1]); pb[4 + q] = make_double2(T1[2 * q], T1[2 * q + 1]); }
                pb[3] = make_double2(-f0, -e0); pb[7] = make_double2(-f1, -e1);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 4
            for (int m = 0; m < 16; ++m) {      // k = 4 m + lk: entry 2 m + (lk >> 1) of this half, factor pair s = lk & 1
                const int eb = 2 * m + (lk >> 1), sp = lk & 1;
                const double av = stA[eb * 12 + sp * 6 + (li < 6 ? li : 0)];
                const double bv = stB[eb * 16 + sp * 8 + (li < 8 ? li : 0)];
                if (m & 1) accO = __builtin_amdgcn_mfma_f64_16x16x4f64(li < 6 ? av : 0.0, li < 8 ? bv : 0.0, accO, 0, 0, 0);
                else accE = __builtin_amdgcn_mfma_f64_16x16x4f64(li < 6 ? av : 0.0, li < 8 ? bv : 0.0, accE, 0, 0, 0);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        // C/D layout: column = lane & 15, row = (lane >> 4) + 4 v.  Rows 0-5 x columns 0-5: the block; column 6: bschur part; 7: bp part
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int r = lk + 4 * v;
            const double val = accE[v] + accO[v];
            if (r < 6) {
                if (li < 6) s_red[wv][r * 6 + li] = val;
                else if (li == 6) s_red[wv][36 + r] = val;
                else if (li == 7) s_red[wv][42 + r] = val;
            }
        }
    }
    __syncthreads();
    SSTAMP(2);
    const int nred = diag ? 48 : 36;
    if (t < nred) s_in[t] = (s_red[0][t] + s_red[1][t]) + (s_red[2][t] + s_red[3][t]);
    const int nch = m.nch;
    if (nch > 1) {
        // publish this chunk's partial sums (sc1, drained), count arrivals; the last chunk folds them in chunk order
        double* part = d.schur_part + (size_t)m.slot * 48;
        if (t < nred) __hip_atomic_store(&part[t], s_in[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) s_last = (__hip_atomic_fetch_add(&d.pair_cnt[p], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nch - 1) ? 1 : 0;      // targeted hand-off, see trial_arrive
        __syncthreads();
        SSTAMP(3);
#ifdef PLBA_STAMPS_LM
        if (!s_last && t == 0 && (ch == 0 || ch == d.nchunks / 2 || ch == d.nchunks - 1)) { const int o = ch == 0 ? 0 : ch == d.nchunks / 2 ? 8 : 16; for (int q = 0; q < 4; ++q) d.dbgbuf[o + q] = (double)(sts[q] - sts[0]); d.dbgbuf[o + 4] = -1; d.dbgbuf[o + 5] = (double)rt0; d.dbgbuf[o + 6] = (double)(long long)__builtin_amdgcn_s_memrealtime(); }
#endif
        if (!s_last) return;
        if (t == 0) __hip_atomic_store(&d.pair_cnt[p], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
        if (t < nred) {
            const double* base = d.schur_part + (size_t)m.ch0 * 48 + t;
            double sum = 0.0;
            for (int c = 0; c < nch; ++c) sum += __hip_atomic_load(const_cast<double*>(base + (size_t)c * 48), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_in[t] = sum;
        }
    }
    __syncthreads();
    // S = C^T Inner C with C = blkdiag(Rcb, Rcb):  tmp = Inner C, S = C^T tmp
    const double* Rcb = d.cam.Rcb.a;
    if (t < 36) {
        const int r = t / 6, c = t % 6, cb = (c / 3) * 3, cc = c % 3;
        s_tmp[t] = s_in[r * 6 + cb] * Rcb[cc] + s_in[r * 6 + cb + 1] * Rcb[3 + cc] + s_in[r * 6 + cb + 2] * Rcb[6 + cc];
    }
    __syncthreads();
    if (t < 36) {
        const int r = t / 6, c = t % 6, rb = (r / 3) * 3, rr = r % 3;
        const double v = Rcb[rr] * s_tmp[rb * 6 + c] + Rcb[3 + rr] * s_tmp[(rb + 1) * 6 + c] + Rcb[6 + rr] * s_tmp[(rb + 2) * 6 + c];
        if (diag) {      // (r,c) and (c,r) are two lanes' entries of the same block: each writes its own
            d.sys[a0] = old0 + v;
        } else {
            d.sys[a0] = old0 + v;
            d.sys[a1] = old1 + v;
        }
    } else if (diag && t < 48) {
        const int q = t - 36, r = q % 6, rb = (r / 3) * 3, rr = r % 3;
        const double* g = s_in + 36 + (q / 6) * 6;
        const double v = Rcb[rr] * g[rb] + Rcb[3 + rr] * g[rb + 1] + Rcb[6 + rr] * g[rb + 2];
        d.sys[a0] = old0 + v;                          // bschur row / bp row of the augmented system
        if (t >= 42) d.bpg[a1] = old1 + v;             // bp is consumed by the factorisation in sys; computeScale needs it afterwards
    }
#ifdef PLBA_STAMPS_LM
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SSTAMP(4);
    if (t == 0 && (ch == 0 || ch == d.nchunks / 2 || ch == d.nchunks - 1)) { const int o = ch == 0 ? 0 : ch == d.nchunks / 2 ? 8 : 16; for (int q = 0; q < 5; ++q) d.dbgbuf[o + q] = (double)(sts[q] - sts[0]); d.dbgbuf[o + 5] = (double)rt0; d.dbgbuf[o + 6] = (double)(long long)__builtin_amdgcn_s_memrealtime(); }
#endif
}

// sys = Himu + Hconst (+ lambda I on the real diagonal, 1 on the padded diagonal) ; row Ppad = row Ppad+1 = pose-side gradient
__global__ __launch_bounds__(256) void k_assemble(DevBuf d, int add_lambda) {
    assemble_part(d, add_lambda, blockIdx.x, gridDim.x, threadIdx.x, 256);
}

// SparseOptimizer::update for one keyframe (oplusImpl of VertexNavStatePVR / VertexNavStateBias)
DEV void update_kf_one(const DevBuf& d, int cur, int trial, int k) {
    const double* s = d.kf[cur] + (size_t)k * KF_STRIDE;
    double* o = d.kf[trial] + (size_t)k * KF_STRIDE;
    double tmp[KF_STRIDE];
#pragma unroll
    for (int i = 0; i < KF_STRIDE; ++i) tmp[i] = s[i];
    const bool ok = d.ctrl->solver_ok != 0;
    const int op = d.kf_off_pvr[k], ob = d.kf_off_bias[k];
    if (ok && op >= 0) {
        double u[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) u[i] = d.x[op + i];
        kf_oplus_pvr(s, u, tmp);
    }
    if (ok && ob >= 0) {   // IMU/NavState.cpp:100-121
#pragma unroll
        for (int i = 0; i < 6; ++i) tmp[16 + i] = s[16 + i] + d.x[ob + i];
    }
#pragma unroll
    for (int i = 0; i < KF_STRIDE; ++i) publish(o + i, tmp[i]);      // (sc1: the pose-side blocks of the SAME launch may read it, k_lm_trial)
}

// -------------------------------------------------------------------------------------------------
// Chain back-substitution of one segment (plba_chain.hip), riding in front of the landmark back-substitution launch:
//   x_c = L^-T (w_b - W_B x_d) over the segment's column window, x into system order, then the state update of the
//   segment's keyframes.  Workgroup 0 also scatters the whole dense solution and updates the keyframes whose step
//   lies entirely in it (separators, fixed keyframes).  The landmark workgroups of the same launch read the pose
//   part of the step from the dense solution `xd` itself, never from d.x.
// -------------------------------------------------------------------------------------------------
// state update of one keyframe from explicit step values (update_kf_one without the trip through d.x)
// the keyframe part of update(): trial state = state (+) step   (one function for the segments' global update and the IMU edge blocks' local copies)
DEV void kf_trial_state(const DevBuf& d, const double* s /*KF_STRIDE, current state*/, const double* u9, const double* ub6, bool has_pvr, bool has_bias, double* tmp) {
#pragma unroll
    for (int i = 0; i < KF_STRIDE; ++i) tmp[i] = s[i];
    const bool ok = d.ctrl->solver_ok != 0;
    if (ok && has_pvr) kf_oplus_pvr(s, u9, tmp);
    if (ok && has_bias) {   // IMU/NavState.cpp:100-121
#pragma unroll
        for (int i = 0; i < 6; ++i) tmp[16 + i] = s[16 + i] + ub6[i];
    }
}
DEV void update_kf_vals(const DevBuf& d, int trial, int k, const double* s /*KF_STRIDE, current state*/, const double* u9, const double* ub6, bool has_pvr, bool has_bias) {
    double tmp[KF_STRIDE];
    kf_trial_state(d, s, u9, ub6, has_pvr, has_bias, tmp);
    double* o = d.kf[trial] + (size_t)k * KF_STRIDE;
#pragma unroll
    for (int i = 0; i < KF_STRIDE; ++i) publish(o + i, tmp[i]);
}

// LOCAL (round 4): only the segment's solution, left in LDS (returned) — no keyframe is updated and nothing is published.  The trial
// launch's IMU edge blocks call it for the segment their two keyframes belong to and form the two trial states themselves, instead
// of waiting in-launch for the segment's own workgroup (an in-launch hand-off costs ~4 us on this machine; the
// redundant back-substitution runs on compute units that would otherwise idle).
template <bool LOCAL = false>
DEV const double* chain_back_segment(const DevBuf& d, const ChainView& cv, const double* xd, const int g, const int cur, const int trial) {
    constexpr int BACK_THREADS = LMB;
    constexpr int UFAST = 5;               // 16 lanes x 5 = 80 window columns on the all-in-flight path (6-slot interior positions)
    __shared__ double sv[SEGMAX * 9];
    __shared__ double sxc[SEGMAX * 9];     // the segment's solution, block-major
    __shared__ double sxw[192];            // dense solution over the segment's column window
    __shared__ double sM[SEGMAX][162];     // L_ii^-1 | L_{i+1,i}
#ifdef PLBA_STAMPS_LM
    unsigned long long cts[6]; cts[0] = __builtin_readcyclecounter();
#define CSTAMP(i) cts[i] = __builtin_readcyclecounter()
#else
#define CSTAMP(i) do {} while (0)
#endif
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, t = threadIdx.x;
    const int i0 = cv.seg_start[g], n = cv.seg_start[g + 1] - i0;
    const int wlo = cv.seg_col[2 * g], wn = cv.seg_col[2 * g + 1] - wlo;
    // ---- everything that does not depend on the dense solution is requested first ------------------------------------------
    // (a) W rows: 16 lanes per row, 16 rows per pass
    const int lg = t >> 4, ll = t & 15;
    constexpr int NPASS = (SEGMAX * 9 + 15) / 16;
    double wr[NPASS][UFAST], wbv[NPASS];
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
        const int r = q * 16 + lg;
        const bool in = r < n * 9;
        const double* Wr = cv.W + (size_t)(i0 * 9 + (in ? r : 0)) * cv.Wld + wlo;
#pragma unroll
        for (int u = 0; u < UFAST; ++u) { const int c = ll + 16 * u; wr[q][u] = Wr[c < wn ? c : 0]; }
        wbv[q] = Wr[cv.Pd - wlo];
    }
    // (b) the keyframe this thread will update (threads [0, n): the segment's; workgroup 0, threads [64, 64 + nukf): the rest): one
    // host-built descriptor, then its state — two levels of loads where the index maps took four
    int kf_k = -1, e_blk = -1;
    const int32_t* kd = nullptr;
    if (LOCAL) {}
    else if (t < n) { e_blk = t; kd = cv.bkf + (size_t)(g * CHAIN_SEG + t) * 20; }
    else if (g == 0 && t >= 64 && t - 64 < cv.nukf && t - 64 < BACK_THREADS - 64) kd = cv.bkf + (size_t)(cv.nseg * CHAIN_SEG + (t - 64)) * 20;
    double ks[KF_STRIDE];
    int op = -1, ob = -1, pc[6] = {-1, -1, -1, -1, -1, -1}, cc[9] = {-1, -1, -1, -1, -1, -1, -1, -1, -1};
    if (kd) {
        const int4* k4 = reinterpret_cast<const int4*>(kd);
        const int4 a0 = k4[0], a1 = k4[1], a2 = k4[2], a3 = k4[3], a4 = k4[4];
        kf_k = a0.x; op = a0.y; ob = a0.z;
        pc[0] = a0.w; pc[1] = a1.x; pc[2] = a1.y; pc[3] = a1.z; pc[4] = a1.w; pc[5] = a2.x;
        cc[0] = a2.y; cc[1] = a2.z; cc[2] = a2.w; cc[3] = a3.x; cc[4] = a3.y; cc[5] = a3.z; cc[6] = a3.w; cc[7] = a4.x; cc[8] = a4.y;
        const double* sp = d.kf[cur] + (size_t)kf_k * KF_STRIDE;
#pragma unroll
        for (int i = 0; i < KF_STRIDE; ++i) ks[i] = sp[i];
    }
    // (c) the factors
    for (int idx = t; idx < n * 81; idx += BACK_THREADS) {
        sM[idx / 81][idx % 81] = cv.Ldinv[(size_t)i0 * 81 + idx];
        sM[idx / 81][81 + idx % 81] = cv.Lsub[(size_t)i0 * 81 + idx];
    }
    // ---- the dense solution ---------------------------------------------------------------------------------------------------
    if (!LOCAL && g == 0) for (int c = t; c < cv.Pd; c += BACK_THREADS) publish(&d.x[cv.pidx[c]], xd[c]);      // (x is read by the deciding workgroup of the same launch)
    for (int c = t; c < wn; c += BACK_THREADS) sxw[c] = xd[wlo + c];
    double u9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, ub6[6] = {0, 0, 0, 0, 0, 0};
    if (kf_k >= 0) {
        const int ps[6] = {0, 1, 2, 6, 7, 8};
#pragma unroll
        for (int i = 0; i < 6; ++i) if (pc[i] >= 0) u9[ps[i]] = xd[pc[i]];
#pragma unroll
        for (int i = 0; i < 3; ++i) if (cc[i] >= 0) u9[3 + i] = xd[cc[i]];
#pragma unroll
        for (int i = 0; i < 6; ++i) if (cc[3 + i] >= 0) ub6[i] = xd[cc[3 + i]];
    }
    __syncthreads();
    CSTAMP(1);
    // ---- v = w_b - W_B x_d over the window ------------------------------------------------------------------------------------
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
        const int r = q * 16 + lg;
        double sacc = 0.0;
#pragma unroll
        for (int u = 0; u < UFAST; ++u) { const int c = ll + 16 * u; sacc = fma(wr[q][u], c < wn ? sxw[c] : 0.0, sacc); }
        if (wn > 16 * UFAST && r < n * 9) {      // wide windows (separators next to each other): the rest of the row, not pre-fetched
            const double* Wr = cv.W + (size_t)(i0 * 9 + r) * cv.Wld + wlo;
            for (int c = 16 * UFAST + ll; c < wn; c += 16) sacc = fma(Wr[c], sxw[c], sacc);
        }
        sacc += dpp_get<0x111, 0xf>(sacc);      // row_shr 1, 2, 4, 8: lane 15 of each 16-lane row holds the row's sum
        sacc += dpp_get<0x112, 0xf>(sacc);
        sacc += dpp_get<0x114, 0xf>(sacc);
        sacc += dpp_get<0x118, 0xf>(sacc);
        if (ll == 15 && r < n * 9) sv[r] = wbv[q] - sacc;
    }
    __syncthreads();
    CSTAMP(2);
    if (wv == 0) {
        const int r = lane < 9 ? lane : 8;     // lanes 0..8 = components; x_{i+1} is kept in lanes 0..8 of `xn`
        double xn = 0.0;
        for (int i = n - 1; i >= 0; --i) {
            const double* Li = sM[i];              // L_ii^-1
            const double* Ls = Li + 81;            // L_{i+1,i}
            double tt = sv[i * 9 + r];
            if (i + 1 < n) {
#pragma unroll
                for (int q = 0; q < 9; ++q) tt = fma(-Ls[q * 9 + r], lane_bcast(xn, q), tt);     // (L_{i+1,i}^T x_{i+1})_r
            }
            double xi = 0.0;
#pragma unroll
            for (int q = 0; q < 9; ++q) xi = fma(Li[q * 9 + r], lane_bcast(tt, q), xi);         // (L_ii^-T t)_r ; L^-1 is lower: zeros where q < r
            xn = xi;
            const int gi = cv.cidx[(i0 + i) * 9 + r];
            if (lane < 9) { sxc[i * 9 + r] = gi >= 0 ? xi : 0.0; if (!LOCAL && gi >= 0) publish(&d.x[gi], xi); }
        }
    }
    __syncthreads();
    CSTAMP(3);
    if (LOCAL) return sxc;
    // ---- keyframe part of update() -----------------------------------------------------------------------------------------
    if (kf_k >= 0) {
        if (e_blk >= 0) {
#pragma unroll
            for (int i = 0; i < 3; ++i) u9[3 + i] = sxc[e_blk * 9 + i];
#pragma unroll
            for (int i = 0; i < 6; ++i) ub6[i] = sxc[e_blk * 9 + 3 + i];
            // the pose part of this keyframe's step in system order (the same values workgroup 0 scatters)
            if (op >= 0) { publish(&d.x[op], u9[0]); publish(&d.x[op + 1], u9[1]); publish(&d.x[op + 2], u9[2]); publish(&d.x[op + 6], u9[6]); publish(&d.x[op + 7], u9[7]); publish(&d.x[op + 8], u9[8]); }
        }
        update_kf_vals(d, trial, kf_k, ks, u9, ub6, op >= 0, ob >= 0);
    }
    if (g == 0) for (int u = BACK_THREADS - 64 + t; u < cv.nukf; u += BACK_THREADS) update_kf_one(d, cur, trial, cv.ukf[u]);   // more than 192 of them: d.x is complete for these (scattered above, same workgroup)
#ifdef PLBA_STAMPS_LM
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CSTAMP(4);
    if (threadIdx.x == 0 && g < 2) { for (int q = 0; q < 5; ++q) d.dbgbuf[48 + 6 * g + q] = (double)(cts[q] - cts[0]); d.dbgbuf[48 + 6 * g + 5] = (double)(long long)__builtin_amdgcn_s_memrealtime(); }
#endif
    return sxc;
}
// what an IMU edge block needs to form its two keyframes' trial states locally
struct LocalStates { const ChainView* cv; const double* xd; int cur; };

// -------------------------------------------------------------------------------------------------
// landmark back-substitution + landmark update (+ landmark part of computeScale)
//   xl = D (bl - sum_e w Jl^T Jp x_kf)      trial_lm = cur_lm + xl
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LMB) void k_backsub(DevBuf d, int cur, int trial, ChainView cv, const double* xd, int nlead) {
    static_assert(LMB == 256, "the chain segments ride in this launch");
    if ((int)blockIdx.x < nlead) { chain_back_segment(d, cv, xd, blockIdx.x, cur, trial); return; }
    const int bid = blockIdx.x - nlead;
    extern __shared__ double s_dyn[];
    double* s_kc = s_dyn;                       // K x 12 camera blocks of the CURRENT (linearisation) state
    double* s_y = s_dyn + d.K * KFCAM_STRIDE;   // K x 6: blkdiag(Rcb,Rcb) * (dp, dphi) of every keyframe's step (0 for a fixed one)
    // round 1: everything that depends on the landmark slot alone, in flight together with the staging's loads
    const int slot = bid * LML + (threadIdx.x / LMG), sub = threadIdx.x % LMG;
    const bool valid = slot < d.L;          // uniform over the lanes of a landmark
    const bool lead = valid && sub == 0;
    const int s = valid ? d.lm_start[slot] : 0, en = valid ? d.lm_start[slot + 1] : 0;
    const bool on = valid && d.lm_active[slot] && d.ctrl->solver_ok;
    const double lambda = d.ctrl->lambda;
    double b[6], Dm[12], Lc[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) { b[t] = lead ? d.bl[(size_t)slot * 6 + t] : 0.0; Lc[t] = lead ? d.lm[cur][(size_t)slot * 6 + t] : 0.0; }
#pragma unroll
    for (int t = 0; t < 12; ++t) Dm[t] = lead ? d.dinv[(size_t)slot * 12 + t] : 0.0;
    for (int k = threadIdx.x; k < d.K; k += LMB) {
        kfcam_make(d.cam, d.kf[cur] + (size_t)k * KF_STRIDE, s_kc + k * KFCAM_STRIDE);
        const int o = d.kf_off_pvr[k];
        V3 yp = v3(0, 0, 0), yr = v3(0, 0, 0);
        if (o >= 0 && nlead) {      // the chain workgroups of this launch are still writing d.x: take the pose step from the dense solution
            const int32_t* sc = cv.slotcol + cv.kfpos[k] * NSLOT;
            yp = mul(d.cam.Rcb, v3(xd[sc[0]], xd[sc[1]], xd[sc[2]])); yr = mul(d.cam.Rcb, v3(xd[sc[3]], xd[sc[4]], xd[sc[5]]));
        } else if (o >= 0) { yp = mul(d.cam.Rcb, v3(d.x[o], d.x[o + 1], d.x[o + 2])); yr = mul(d.cam.Rcb, v3(d.x[o + 6], d.x[o + 7], d.x[o + 8])); }
        double* y = s_y + k * 6;
        y[0] = yp.x; y[1] = yp.y; y[2] = yp.z; y[3] = yr.x; y[4] = yr.y; y[5] = yr.z;
    }
    // round 2: the first edge's indices
    int ed = s + sub;
    int pos = 0, kfi = 0;
    if (on && ed < en) { pos = d.ob_pos[ed]; kfi = d.ob_kf[ed]; }
    __syncthreads();
    if (bid == 0 && nlead == 0) for (int k = threadIdx.x; k < d.K; k += LMB) update_kf_one(d, cur, trial, k);   // keyframe part of update()
    double sc = 0.0;
    if (valid) {
        double xl[6] = {0, 0, 0, 0, 0, 0};
        const bool is_pt = slot < d.Np;
        double c[6] = {0, 0, 0, 0, 0, 0};
        if (on) {
            const double sl = is_pt ? -1.0 : 1.0;
            while (ed < en) {      // lane `sub` takes every LMG-th edge; the next edge's indices travel while this one is processed
                const double* rec = d.erec + (size_t)pos * EREC_UNIT;
                const int k = kfi;
                const EdgeRows r = load_rows(d, rec, s_kc + k * KFCAM_STRIDE, is_pt);
                ed += LMG;
                if (ed < en) { pos = d.ob_pos[ed]; kfi = d.ob_kf[ed]; }
                if (r.w == 0.0) continue;
                const double* y = s_y + k * 6;      // zero for a fixed keyframe
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int q = 0; q < 6; ++q) { s0 += r.ga[q] * y[q]; s1 += r.gb[q] * y[q]; }
                const double a0 = r.w * sl * s0, a1 = r.w * sl * s1;    // w * Jl_a^T s_a with Jl_a = sl * v_a^T
                if (is_pt) { c[0] -= r.va.x * a0 + r.vb.x * a1; c[1] -= r.va.y * a0 + r.vb.y * a1; c[2] -= r.va.z * a0 + r.vb.z * a1; }
                else { c[0] -= r.va.x * a0; c[1] -= r.va.y * a0; c[2] -= r.va.z * a0; c[3] -= r.vb.x * a1; c[4] -= r.vb.y * a1; c[5] -= r.vb.z * a1; }
            }
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) c[t] = quad_sum(c[t]);      // uniform over the group (`on` is)
        if (!is_pt) {
#pragma unroll
            for (int t = 3; t < 6; ++t) c[t] = quad_sum(c[t]);
        }
        if (sub == 0) {
            if (on) {
#pragma unroll
                for (int t = 0; t < 6; ++t) c[t] += b[t];
                V3 a = sym3_mul(Dm, v3(c[0], c[1], c[2]));
                xl[0] = a.x; xl[1] = a.y; xl[2] = a.z;
                if (!is_pt) { V3 e = sym3_mul(Dm + 6, v3(c[3], c[4], c[5])); xl[3] = e.x; xl[4] = e.y; xl[5] = e.z; }
#pragma unroll
                for (int t = 0; t < 6; ++t) sc += xl[t] * (lambda * xl[t] + b[t]);
            }
            double* Lt = d.lm[trial] + (size_t)slot * 6;
#pragma unroll
            for (int t = 0; t < 6; ++t) { Lt[t] = Lc[t] + xl[t]; d.xl[(size_t)slot * 6 + t] = xl[t]; }
        }
    }
    const double bs = wave_sum(sc);      // one partial per workgroup, waves added in order
    __shared__ double s_w[LMW];
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = bs;
    __syncthreads();
    if (threadIdx.x == 0) { double v = s_w[0]; for (int q = 1; q < LMW; ++q) v += s_w[q]; d.scale_part[bid] = v; }
#ifdef PLBA_STAMPS_LM
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && (bid == 0 || bid == (int)gridDim.x - nlead - 1)) d.dbgbuf[60 + (bid == 0 ? 0 : 1)] = (double)(long long)__builtin_amdgcn_s_memrealtime();
#endif
}

__global__ void k_update_kf(DevBuf d, int cur, int trial) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < d.K) update_kf_one(d, cur, trial, k);
}

// -------------------------------------------------------------------------------------------------
// K3: IMU edges.  One wavefront per (PVR edge, bias edge) pair of consecutive keyframes.
// Lane 0 evaluates the residuals and the three Jacobian blocks into LDS; all 64 lanes then form
// Omega' J and J^T Omega' J (24 x 24 over [PVR_i | PVR_j | Bias_i]) and add them into the dense
// pose-side system with fp64 atomics (a handful of edges share a destination block).
// -------------------------------------------------------------------------------------------------
template <bool JAC, int NT>
DEV void pose_edge_block(const DevBuf& d, int state, const Robust& rb, int m, int lane, const DecideArgs* arrive, int nblk_edges, double* s4, const LeadWait* lw, const LocalStates* loc) {
    __shared__ double sKF[2 * KF_STRIDE];      // (k_lm_trial) the two keyframe states, read coherently once their producers have counted in
    __shared__ double sJ[9 * 24];     // [J0 | J1 | J2] row-major 9 x 24
    __shared__ double sOJ[9 * 24];
    __shared__ double sE[16];
    __shared__ double sT[16];         // Omega e (9) | Omega_b e_b (6)
    __shared__ double sW[2];
    __shared__ int sOff[24];          // system index of each of the 24 columns, -1 = fixed
    const int ki = d.imu_i[m], kj = d.imu_j[m];
    const double* si = d.kf[state] + (size_t)ki * KF_STRIDE;
    const double* sj = d.kf[state] + (size_t)kj * KF_STRIDE;
    if (loc && loc->cv->imu_loc) {
        // (k_lm_trial, round 4) the two trial states are formed HERE: the chain dims of the edge's keyframes from a local back-substitution of
        // their segment (or from the dense solution: separators), the pose dims from the dense solution, the same kf_trial_state the
        // segment's own workgroup publishes — bit-identical states, no in-launch wait
        const ChainView& cvl = *loc->cv;
        const int32_t* il = cvl.imu_loc + 4 * (size_t)m;
        const int seg = il[0];
        const double* sxc = nullptr;
        if (seg >= 0) sxc = chain_back_segment<true>(d, cvl, loc->xd, seg, loc->cur, state);      // (ends with a workgroup barrier)
        if (lane < 2) {
            const int k = lane ? kj : ki, dsc = il[1 + lane];
            const double* s = d.kf[loc->cur] + (size_t)k * KF_STRIDE;
            double ks[KF_STRIDE], u9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, ub6[6] = {0, 0, 0, 0, 0, 0}, tmp[KF_STRIDE];
#pragma unroll
            for (int i = 0; i < KF_STRIDE; ++i) ks[i] = s[i];
            bool has_pvr = false, has_bias = false;
            if (dsc >= 0) {
                const int32_t* kd = cvl.bkf + (size_t)dsc * 20;
                has_pvr = kd[1] >= 0; has_bias = kd[2] >= 0;
                const int ps[6] = {0, 1, 2, 6, 7, 8};
#pragma unroll
                for (int i = 0; i < 6; ++i) if (kd[3 + i] >= 0) u9[ps[i]] = loc->xd[kd[3 + i]];
                const int e = dsc - seg * CHAIN_SEG;
                if (seg >= 0 && e >= 0 && e < CHAIN_SEG && dsc < cvl.nseg * CHAIN_SEG) {      // an eliminated block of the segment: its chain dims come from the local back-substitution
#pragma unroll
                    for (int i = 0; i < 3; ++i) u9[3 + i] = sxc[e * 9 + i];
#pragma unroll
                    for (int i = 0; i < 6; ++i) ub6[i] = sxc[e * 9 + 3 + i];
                } else {
#pragma unroll
                    for (int i = 0; i < 3; ++i) if (kd[9 + i] >= 0) u9[3 + i] = loc->xd[kd[9 + i]];
#pragma unroll
                    for (int i = 0; i < 6; ++i) if (kd[12 + i] >= 0) ub6[i] = loc->xd[kd[12 + i]];
                }
            }
            kf_trial_state(d, ks, u9, ub6, has_pvr, has_bias, tmp);
#pragma unroll
            for (int i = 0; i < KF_STRIDE; ++i) sKF[lane * KF_STRIDE + i] = tmp[i];
        }
        __syncthreads();
        si = sKF; sj = sKF + KF_STRIDE;
    } else if (lw) {
        lead_wait(*lw);
        if (lane < 2 * KF_STRIDE) sKF[lane] = fetch((lane < KF_STRIDE ? si : sj - KF_STRIDE) + lane, true);
        __syncthreads();
        si = sKF; sj = sKF + KF_STRIDE;
    }
    const double* pre = d.imu_pre + (size_t)m * PRE_STRIDE;
    const double* Om = d.imu_info_pvr + (size_t)m * 81;
    const double* Ob = d.imu_info_bias + (size_t)m * 36;
#ifdef PLBA_STAMPS_LM
    unsigned long long pts[8] = {0,0,0,0,0,0,0,0}; pts[0] = __builtin_readcyclecounter();
#define PSTAMP(i) pts[i] = __builtin_readcyclecounter()
#else
#define PSTAMP(i) do {} while (0)
#endif
    if (JAC) {
        for (int t = lane; t < 9 * 24; t += NT) sJ[t] = 0.0;
        __syncthreads();      // LDS only: nothing global is waited for before the serial parts start
        if (lane >= 64 && lane < 64 + 24) {      // wave 1, in the shadow of the error evaluation
            const int c = lane - 64;
            const int o = c < 9 ? d.kf_off_pvr[ki] : c < 18 ? d.kf_off_pvr[kj] : d.kf_off_bias[ki];
            sOff[c] = o < 0 ? -1 : o + (c < 9 ? c : c < 18 ? c - 9 : c - 18);
        }
    }
    __shared__ double sS[18];         // Rj^T Ri | Jr(JRg dbg): what the residual-dependent Jacobian blocks need from the static part
    double e9[9];
    if (JAC && lane == 128) {         // wave 2: the Jacobian entries that do not depend on the rotation residual, next to the error
        M3 RjTRi, JrB;
        pvr_jac_static(si, sj, pre, d.gw, sJ, sJ + 9, sJ + 18, 24, 24, RjTRi, JrB);
#pragma unroll
        for (int q = 0; q < 9; ++q) { sS[q] = RjTRi.a[q]; sS[9 + q] = JrB.a[q]; }
    }
    if (lane == 0) {
        double e6[6];
        pvr_error(si, sj, pre, d.gw, e9);
        bias_error(si, sj, e6);
#pragma unroll
        for (int r = 0; r < 9; ++r) sE[r] = e9[r];
#pragma unroll
        for (int r = 0; r < 6; ++r) sE[9 + r] = e6[r];
        PSTAMP(1);
    }
    if (!JAC) {
        // error pass: chi = e^T Omega e with a lane per row of Omega (same row sums, same order as the serial form), so the
        // 117 dependent multiply-adds and their loads do not sit behind the serial error evaluation
        __syncthreads();
        if (lane < 9) {
            double t = 0.0;
            for (int c = 0; c < 9; ++c) t += Om[lane * 9 + c] * sE[c];
            sT[lane] = t;
        } else if (lane >= 64 && lane < 64 + 6) {
            const int r = lane - 64;
            double t = 0.0;
            for (int c = 0; c < 6; ++c) t += Ob[r * 6 + c] * sE[9 + c];
            sT[9 + r] = t;
        } else if (lane >= 128 && lane < 128 + 15) {
            d.imu_err[(size_t)m * 16 + (lane - 128)] = sE[lane - 128];
        }
        __syncthreads();
        if (lane == 0) {
            double chi = 0.0, chib = 0.0;
            for (int r = 0; r < 9; ++r) chi += sE[r] * sT[r];
            for (int r = 0; r < 6; ++r) chib += sE[9 + r] * sT[9 + r];
            double r0 = chi, r1 = 1.0, b0 = chib, b1 = 1.0;
            if (rb.on[PLBA_EDGE_IMU_PVR]) huber(chi, rb.delta[PLBA_EDGE_IMU_PVR], r0, r1);
            if (rb.on[PLBA_EDGE_IMU_BIAS]) huber(chib, rb.delta[PLBA_EDGE_IMU_BIAS], b0, b1);
            double* co = d.imu_chi + (size_t)m * 4;
            publish(co, chi); publish(co + 1, chib); publish(co + 2, r0); publish(co + 3, b0);      // read by the last-arriving workgroup of this launch
        }
        return;
    }
    __syncthreads();      // e in LDS
    PSTAMP(2);
    // wave 0, lane 0 goes on with the Jacobians (written straight into their LDS layout); the other waves form
    // chi = e^T Omega e, the robust weights and the cached error / chi2 records meanwhile
    if (lane == 0) {
        M3 RjTRi, JrB;
#pragma unroll
        for (int q = 0; q < 9; ++q) { RjTRi.a[q] = sS[q]; JrB.a[q] = sS[9 + q]; }
        pvr_jac_rphi(pre, e9, RjTRi, JrB, sJ, sJ + 9, sJ + 18, 24, 24);
        PSTAMP(3);
    } else if (lane >= 64 && lane < 64 + 9) {          // wave 1: (Omega e)_r, row sums in the serial order of the reference
        const int r = lane - 64;
        double t = 0.0;
        for (int c = 0; c < 9; ++c) t += Om[r * 9 + c] * sE[c];
        sT[r] = t;
    } else if (lane >= 128 && lane < 128 + 6) {        // wave 2: (Omega_b e_b)_r
        const int r = lane - 128;
        double t = 0.0;
        for (int c = 0; c < 6; ++c) t += Ob[r * 6 + c] * sE[9 + c];
        sT[9 + r] = t;
    } else if (lane >= 192 && lane < 192 + 15) {       // wave 3: cached errors
        d.imu_err[(size_t)m * 16 + (lane - 192)] = sE[lane - 192];
    }
    __syncthreads();
    if (lane == 64) {
        double chi = 0.0, chib = 0.0;
        for (int r = 0; r < 9; ++r) chi += sE[r] * sT[r];
        for (int r = 0; r < 6; ++r) chib += sE[9 + r] * sT[9 + r];
        double r0 = chi, r1 = 1.0, b0 = chib, b1 = 1.0;
        if (rb.on[PLBA_EDGE_IMU_PVR]) huber(chi, rb.delta[PLBA_EDGE_IMU_PVR], r0, r1);
        if (rb.on[PLBA_EDGE_IMU_BIAS]) huber(chib, rb.delta[PLBA_EDGE_IMU_BIAS], b0, b1);
        double* co = d.imu_chi + (size_t)m * 4;
        publish(co, chi); publish(co + 1, chib); publish(co + 2, r0); publish(co + 3, b0);      // the trial pass with Jacobians: read by the last-arriving workgroup of this launch
        sW[0] = r1; sW[1] = b1;
    }
    PSTAMP(4);
    __syncthreads();     // every thread of the NT-thread block reaches every barrier
    if (arrive) trial_arrive(d, *arrive, nblk_edges, s4);      // chi2 is out: the decision need not wait for this block's Jacobians
    const double w = sW[0], wb = sW[1];
    // OJ = w * Omega * J   (9 x 24)
    for (int t = lane; t < 9 * 24; t += NT) {
        const int r = t / 24, c = t % 24;
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 9; ++k) s += Om[r * 9 + k] * sJ[k * 24 + c];
        sOJ[t] = w * s;
    }
    __syncthreads();
    PSTAMP(5);
    const int ld = d.ld;
    // H += J^T OJ (24 x 24), g += -J^T (w Omega e) = -OJ^T e
    for (int t = lane; t < 24 * 24; t += NT) {
        const int a = t / 24, b = t % 24;
        const int oa = sOff[a], ob = sOff[b];
        if (oa < 0 || ob < 0) continue;
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 9; ++k) s += sJ[k * 24 + a] * sOJ[k * 24 + b];
        // the bias edge's own (Bias_i, Bias_i) term joins here, so that every destination receives ONE add per workgroup: with
        // the neighbouring edge's workgroup that makes two adders per address, and a two-term sum does not depend on the
        // order of the atomics (bit-reproducible pose-side system)
        if (a >= 18 && b >= 18) s += wb * Ob[(a - 18) * 6 + (b - 18)];
        if (s != 0.0) atomicAdd(&d.Himu[(size_t)oa * ld + ob], s);
    }
    if (lane < 24) {
        const int oa = sOff[lane];
        if (oa >= 0) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 9; ++k) s += sOJ[k * 24 + lane] * sE[k];
            if (lane >= 18) {      // + the bias edge's gradient on Bias_i
                double sb = 0.0;
#pragma unroll
                for (int c = 0; c < 6; ++c) sb += Ob[(lane - 18) * 6 + c] * sE[9 + c];
                s -= wb * sb;
            }
            atomicAdd(&d.bimu[oa], -s);
        }
    }
#ifdef PLBA_STAMPS_LM
    if (JAC && lane == 0 && m == 1) { PSTAMP(6); for (int q = 0; q < 7; ++q) d.dbgbuf[24 + q] = (double)(pts[q] - pts[0]); }
#endif
    // bias edge: J = [-I, +I] on (Bias_i, Bias_j); H_ii += W, H_jj += W, H_ij = H_ji -= W; g_i += W e, g_j -= W e
    {
        const int oi = d.kf_off_bias[ki], oj = d.kf_off_bias[kj];
        if (lane < 36) {
            const int r = lane / 6, c = lane % 6;
            const double v = wb * Ob[r * 6 + c];
            if (v != 0.0) {
                if (oj >= 0) atomicAdd(&d.Himu[(size_t)(oj + r) * ld + oj + c], v);      // (Bias_i, Bias_i): added with the PVR edge's block above
                if (oi >= 0 && oj >= 0) {
                    atomicAdd(&d.Himu[(size_t)(oi + r) * ld + oj + c], -v);
                    atomicAdd(&d.Himu[(size_t)(oj + r) * ld + oi + c], -v);
                }
            }
        } else if (lane < 42) {
            const int r = lane - 36;
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < 6; ++c) s += Ob[r * 6 + c] * sE[9 + c];
            s *= wb;
            if (oj >= 0) atomicAdd(&d.bimu[oj + r], -s);      // Bias_i's share went out with the PVR edge's gradient above
        }
    }
}
template <bool JAC>
__global__ __launch_bounds__(256) void k_pose_edges(DevBuf d, int state, Robust rb) {
    pose_edge_block<JAC, 256>(d, state, rb, blockIdx.x, threadIdx.x);      // the block's four waves have different jobs
}

// K4: marginalization prior edge (one workgroup): dx, e = r0 + J0 dx, chi2 = |e|^2, g += -J0^T e.
// Its Hessian J0^T J0 is constant and pre-scattered into Hconst at upload.
template <bool JAC>
DEV void prior_block(const DevBuf& d, int state, const DecideArgs* arrive, int nblk_edges, double* s4a, const LeadWait* lw) {
    __shared__ double s4[4];
    const int n = d.pr_n;
    if (lw) lead_wait(*lw);
    for (int v = threadIdx.x; v < d.pr_nv; v += 256) {
        const double* s = d.kf[state] + (size_t)d.pr_kf[v] * KF_STRIDE;
        double sc[KF_STRIDE];
        if (lw) {
#pragma unroll
            for (int i = 0; i < KF_STRIDE; ++i) sc[i] = fetch(s + i, true);
            s = sc;
        }
        const double* x0 = d.pr_x0 + d.pr_x0off[v];
        double* dx = d.pr_dx + d.pr_idx[v];
        if (d.pr_isbias[v]) prior_dx_bias(s, x0, dx);
        else prior_dx_pvr(s, x0, dx);
    }
    __syncthreads();
    double chi = 0.0;
    for (int r = threadIdx.x; r < n; r += 256) {
        double sacc = d.pr_r0[r];
        for (int c = 0; c < n; ++c) sacc += d.pr_J0[(size_t)c * n + r] * d.pr_dx[c];
        d.pr_err[r] = sacc;
        chi += sacc * sacc;
    }
    double tot = block_sum_256(chi, s4);
    if (threadIdx.x == 0) publish(&d.pr_chi[0], tot);
    if (!JAC) return;
    __syncthreads();
    if (arrive) trial_arrive(d, *arrive, nblk_edges, s4a);
    for (int v = 0; v < d.pr_nv; ++v) {
        const int o = d.pr_off[v];
        if (o < 0) continue;
        const int sz = d.pr_size[v], ix = d.pr_idx[v];
        for (int c = threadIdx.x; c < sz; c += 256) {
            const double* col = d.pr_J0 + (size_t)(ix + c) * n;
            double sacc = 0.0;
            for (int r = 0; r < n; ++r) sacc += col[r] * d.pr_err[r];
            d.bprior[o + c] = -sacc;      // a vector of its own (added by the assembly pass): the IMU edges' atomics stay two per address
        }
    }
}
template <bool JAC>
__global__ __launch_bounds__(256) void k_prior(DevBuf d, int state) { prior_block<JAC>(d, state); }

// -------------------------------------------------------------------------------------------------
// K8: reductions and LM control (single workgroup, fixed summation order)
// red[0] = activeRobustChi2 (local), red[1] = landmark part of computeScale (local), red[2] = max |Hll_jj| (local),
// red[3] = 1 when an in-launch wait of THIS rank ran into its bound (Ctrl::sync_fail; only rank 0 has waiters): summed over the ranks with
// the trial's [chi2, scale], so that every rank fails the call together instead of the others entering the next all-reduce alone
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_reduce(DevBuf d, int nblk_edges, int nblk_lm, int pose_edges, double* red) {
    __shared__ double s4[4];
    double c = 0.0, sc = 0.0, md = 0.0;
    for (int i = threadIdx.x; i < nblk_edges; i += 256) c += d.chi_part[i];
    if (pose_edges) {
        for (int i = threadIdx.x; i < d.M; i += 256) c += d.imu_chi[(size_t)i * 4 + 2] + d.imu_chi[(size_t)i * 4 + 3];
        if (threadIdx.x == 0 && d.pr_nv > 0) c += d.pr_chi[0];
    }
    for (int i = threadIdx.x; i < nblk_lm; i += 256) { sc += d.scale_part[i]; md = fmax(md, d.maxd_part[i]); }
    double C = block_sum_256(c, s4);
    double S = block_sum_256(sc, s4);
    double Mx = block_max_256(md, s4);
    if (threadIdx.x == 0) { red[0] = C; red[1] = S; red[2] = Mx; red[3] = d.ctrl->sync_fail ? 1.0 : 0.0; }
}

// diagonal of the pose-side Hessian held by this rank: pose-side edges (Himu) + sum_e Jp^T w Jp (kfdiag);
// in a sharded run the vector is all-reduced (sum) before the max, so every rank derives the same lambda
__global__ __launch_bounds__(256) void k_posediag(DevBuf d) {
    for (int r = blockIdx.x * 256 + threadIdx.x; r < d.ld; r += gridDim.x * 256) d.posediag[r] = (r < d.P) ? d.Himu[(size_t)r * d.ld + r] + d.Hconst[(size_t)r * d.ld + r] : 0.0;
}
__global__ __launch_bounds__(256) void k_posediag_kf(DevBuf d) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= d.K * 6) return;
    const int o = d.kf_off_pvr[t / 6];
    if (o >= 0) d.posediag[o + pmap(t % 6)] += d.kfdiag[t];
}
// start of an outer iteration: currentChi, and on the first one computeLambdaInit (tau * max |H_jj|)
// Sharded runs exchange only what the factorisation reads: row r of the reduced system up to the end of its 32-wide
// diagonal tile, plus the two right-hand-side rows in full.  Packed offset of row r = 32 a + b:
//   off(r) = 512 a (a + 1) + 32 b (a + 1);  the two rhs rows follow the triangle.
DEV size_t tri_off(int r) { const size_t a = (size_t)(r >> 5), b = (size_t)(r & 31); return 512 * a * (a + 1) + 32 * b * (a + 1); }
__global__ __launch_bounds__(256) void k_tri_pack(DevBuf d, double* buf, int unpack) {
    const int r = blockIdx.x;           // 0 .. Ppad + 1
    const bool rhs = r >= d.Ppad;
    const int n = rhs ? d.Ppad : (r | 31) + 1;
    double* row = d.sys + (size_t)r * d.ld;
    double* pk = buf + (rhs ? tri_off(d.Ppad) + (size_t)(r - d.Ppad) * d.Ppad : tri_off(r));
    if (unpack) for (int c = threadIdx.x; c < n; c += 256) row[c] = pk[c];
    else for (int c = threadIdx.x; c < n; c += 256) pk[c] = row[c];
}

// k_reduce inlined into the LM control kernels for the single-GPU path (no exchange between reduce and control)
// coherent: the partial sums come from other workgroups of the SAME launch (sc1 loads), not from an earlier kernel
DEV void reduce_inline(const DevBuf& d, int nblk_edges, int nblk_lm, double* red, double* s4, bool coherent = false) {
    double c = 0.0, sc = 0.0, md = 0.0;
    for (int i = threadIdx.x; i < nblk_edges; i += 256) c += fetch(&d.chi_part[i], coherent);
    for (int i = threadIdx.x; i < d.M; i += 256) c += fetch(&d.imu_chi[(size_t)i * 4 + 2], coherent) + fetch(&d.imu_chi[(size_t)i * 4 + 3], coherent);
    if (threadIdx.x == 0 && d.pr_nv > 0) c += fetch(&d.pr_chi[0], coherent);
    for (int i = threadIdx.x; i < nblk_lm; i += 256) { sc += d.scale_part[i]; md = fmax(md, d.maxd_part[i]); }
    const double C = block_sum_256(c, s4);
    const double S = block_sum_256(sc, s4);
    const double Mx = block_max_256(md, s4);
    if (threadIdx.x == 0) { red[0] = C; red[1] = S; red[2] = Mx; }
    __syncthreads();
}
__global__ __launch_bounds__(256) void k_lambda_init(DevBuf d, LmParams lp, double* red, int first_iter, int iteration, int fused, int keep_chi, int nblk_edges, int nblk_lm) {
    __shared__ double s4[4];
    if (fused) reduce_inline(d, nblk_edges, nblk_lm, red, s4);
    double md = 0.0;
    if (first_iter && fused) {      // one GPU: the pose diagonal is formed here (k_posediag + k_posediag_kf as two passes of this workgroup)
        for (int r = threadIdx.x; r < d.ld; r += 256) d.posediag[r] = (r < d.P) ? d.Himu[(size_t)r * d.ld + r] + d.Hconst[(size_t)r * d.ld + r] : 0.0;
        __syncthreads();
        for (int t = threadIdx.x; t < d.K * 6; t += 256) { const int o = d.kf_off_pvr[t / 6]; if (o >= 0) d.posediag[o + pmap(t % 6)] += d.kfdiag[t]; }
        __syncthreads();
    }
    if (first_iter) for (int r = threadIdx.x; r < d.P; r += 256) md = fmax(md, fabs(d.posediag[r]));
    double mx = block_max_256(md, s4);
    if (threadIdx.x == 0) {
        Ctrl* c = d.ctrl;
        if (!keep_chi) c->current_chi = red[0];
        c->iteration = iteration;
        c->trial = 0;
        if (first_iter) {
            mx = fmax(mx, red[2]);
            c->maxdiag = mx;
            c->lambda = lp.user_lambda > 0 ? lp.user_lambda : lp.tau * mx;
            c->ni = 2.0;
        }
    }
}

// end of a trial: rho test and lambda schedule of OptimizationAlgorithmLevenberg::solve
DEV void decide_body(const DevBuf& d, const LmParams& lp, double* red, int fused, int nblk_edges, int nblk_lm, Mailbox* mail, unsigned long long seq, double* s4, bool coherent) {
    // two memory rounds (the control block; then every partial sum and the pose step together) and ONE barrier: what is
    // left of a trial after its last error is known sits on the critical path of every LM iteration
    __shared__ double s_r[4][4];
    Ctrl* c = d.ctrl;
    const double lambda = c->lambda;
    const int sok = c->solver_ok;
    double cs = 0.0, sc = 0.0, md = 0.0, sp = 0.0;
    if (fused) {
        for (int i = threadIdx.x; i < nblk_edges; i += 256) cs += fetch(&d.chi_part[i], coherent);
        for (int i = threadIdx.x; i < d.M; i += 256) cs += fetch(&d.imu_chi[(size_t)i * 4 + 2], coherent) + fetch(&d.imu_chi[(size_t)i * 4 + 3], coherent);
        if (threadIdx.x == 0 && d.pr_nv > 0) cs += fetch(&d.pr_chi[0], coherent);
        for (int i = threadIdx.x; i < nblk_lm; i += 256) { sc += fetch(&d.scale_part[i], coherent); md = fmax(md, d.maxd_part[i]); }
    }
    if (sok) for (int j = threadIdx.x; j < d.P; j += 256) { const double xj = fetch(&d.x[j], coherent); sp += xj * (lambda * xj + d.bpg[j]); }
    cs = wave_sum(cs); sc = wave_sum(sc); md = wave_max(md); sp = wave_sum(sp);
    if ((threadIdx.x & 63) == 0) { double* r = s_r[threadIdx.x >> 6]; r[0] = cs; r[1] = sc; r[2] = md; r[3] = sp; }
    __syncthreads();
    double SP = 0.0;
    if (threadIdx.x == 0) {
        if (fused) {
            red[0] = (s_r[0][0] + s_r[1][0]) + (s_r[2][0] + s_r[3][0]);
            red[1] = (s_r[0][1] + s_r[1][1]) + (s_r[2][1] + s_r[3][1]);
            red[2] = fmax(fmax(s_r[0][2], s_r[1][2]), fmax(s_r[2][2], s_r[3][2]));
        }
        SP = (s_r[0][3] + s_r[1][3]) + (s_r[2][3] + s_r[3][3]);
    }
    (void)s4;
    if (threadIdx.x == 0) {
    if (!fused && red[3] != 0.0) c->sync_fail = 1;      // sharded runs: some rank's in-launch wait failed (k_reduce, all-reduced)
    double tempChi = red[0];
    if (!c->solver_ok) tempChi = 1.7976931348623157e308;
    double scale = SP + red[1];
    scale += 1e-3;
    const double rho = (c->current_chi - tempChi) / scale;
    const int n = *d.trace_n;
    plba_trace_row row;
    row.iteration = c->iteration; row.trial = c->trial; row.solver_ok = c->solver_ok;
    row.lambda = lambda; row.chi2_current = c->current_chi; row.chi2_trial = tempChi; row.scale = scale; row.rho = rho;
    row.accepted = (rho > 0 && isfinite(tempChi)) ? 1 : 0;
    if (n < d.trace_cap) d.trace[n] = row;
    *d.trace_n = n + 1;
    c->temp_chi = tempChi; c->scale = scale; c->rho = rho;
    if (rho > 0 && isfinite(tempChi)) {
        double alpha = 1. - pow((2 * rho - 1), 3);
        alpha = fmin(alpha, lp.upper);
        const double sf = fmax(lp.lower, alpha);
        c->lambda = lambda * sf;
        c->ni = 2.0;
        c->current_chi = tempChi;
        c->accepted = 1;
        c->iteration += 1;      // the next outer iteration starts from this state: its chi2 is tempChi, trial counter 0
        c->trial = -1;          // (+1 below)
    } else {
        c->lambda = lambda * c->ni;
        c->ni *= 2.0;
        c->accepted = 0;
    }
    c->trial += 1;
    if (!c->solver_ok) c->n_fail += 1;
    c->solver_ok = 1;
    if (mail) {     // hand the decision to the host: payload, system-scope fence, then the sequence number it polls
        mail->c = *c;
        mail->row = row;
        mail->c.sync_fail = __hip_atomic_load(&c->sync_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (possibly set by another workgroup of this launch)
        __threadfence_system();
        __hip_atomic_store(&mail->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    }
}
__global__ __launch_bounds__(256) void k_decide(DevBuf d, LmParams lp, double* red, int fused, int nblk_edges, int nblk_lm, Mailbox* mail, unsigned long long seq) {
    __shared__ double s4[4];
    decide_body(d, lp, red, fused, nblk_edges, nblk_lm, mail, seq, s4, false);
}

// chi2() > thresh || !isDepthPositive()  =>  setLevel(1)   (mapHandler.cpp:6047-6066)
__global__ __launch_bounds__(256) void k_gate(DevBuf d, int state, double thresh, uint8_t* depth_out, int do_gate) {
    extern __shared__ double s_dyn[];
    double* s_kc = s_dyn;
    const double* kf = d.kf[state];
    for (int k = threadIdx.x; k < d.K; k += 256) kfcam_make(d.cam, kf + (size_t)k * KF_STRIDE, s_kc + k * KFCAM_STRIDE);
    __syncthreads();
    const int e = blockIdx.x * 256 + threadIdx.x;
    bool newly_pt = false, newly_ln = false;
    if (e < d.E) {
        const double* L = d.lm[state] + (size_t)d.ob_slot[e] * 6;
        const double* kc = s_kc + d.ob_kf[e] * KFCAM_STRIDE;
        bool dpos = cam_Pc(d.cam, kc, v3(L[0], L[1], L[2])).z > 0.0;
        if (e >= d.Ep) dpos = dpos && (cam_Pc(d.cam, kc, v3(L[3], L[4], L[5])).z > 0.0);
        if (depth_out) depth_out[e] = dpos ? 1 : 0;
        if (do_gate && (d.ob_chi2[e] > thresh || !dpos)) {
            if (d.ob_level[e] == 0) { newly_pt = e < d.Ep; newly_ln = !newly_pt; }
            d.ob_level[e] = 1;
        }
    }
    if (do_gate) {      // one counter update per wave, not one per gated observation (they all hit the same two words)
        const int cp = __popcll(__ballot(newly_pt)), cl = __popcll(__ballot(newly_ln));
        if ((threadIdx.x & 63) == 0) {
            if (cp) atomicAdd(&d.ctrl->n_gate_pt, cp);
            if (cl) atomicAdd(&d.ctrl->n_gate_ln, cl);
        }
    }
}

// Culling decision after the final optimize (mapHandler.cpp:5541-5556, :5611-5620): level-1 edges get computeError() on the
// current estimates (cached chi2 refreshed), then  chi2 > thresh || !isDepthPositive  =>  bad.  Lane per observation.
__global__ __launch_bounds__(256) void k_cull(DevBuf d, int state, double thresh, uint8_t* bad) {
    extern __shared__ double s_dyn[];
    double* s_kc = s_dyn;
    const double* kf = d.kf[state];
    for (int k = threadIdx.x; k < d.K; k += 256) kfcam_make(d.cam, kf + (size_t)k * KF_STRIDE, s_kc + k * KFCAM_STRIDE);
    __syncthreads();
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= d.E) return;
    const double* L = d.lm[state] + (size_t)d.ob_slot[e] * 6;
    const double* kc = s_kc + d.ob_kf[e] * KFCAM_STRIDE;
    double e2[2], rec[12];
    bool dpos;
    if (e < d.Ep) {
        const double2 uv = reinterpret_cast<const double2*>(d.po_uv)[e];
        point_edge_rec(d.cam, kc, v3(L[0], L[1], L[2]), uv.x, uv.y, e2, rec, dpos, false);
    } else {
        const double* l = d.lo_l + (size_t)(e - d.Ep) * 3;
        line_edge_rec(d.cam, kc, v3(L[0], L[1], L[2]), v3(L[3], L[4], L[5]), l[0], l[1], l[2], e2, rec, dpos, false);
    }
    double chi = d.ob_chi2[e];
    if (d.ob_level[e] != 0) { chi = d.ob_w[e] * (e2[0] * e2[0] + e2[1] * e2[1]); d.ob_chi2[e] = chi; }
    bad[e] = (chi > thresh || !dpos) ? 1 : 0;
}

}  // namespace plba
#include "plba_lm_dev.h"
namespace plba {

// ---- fused landmark-major passes: kernels (bodies in plba_lm_dev.h) ------------------------------------------------------------------
// launch C of an iteration: [chain segments, reading the pose-side accumulators directly (they carry no landmark term) | groups]
// WIDE_OK: the instantiation that also carries the wide groups' code (landmarks over 9 .. 16 keyframes); a window without such landmarks
// (lv.wmax == 8: every BASELINE config) runs the instantiation without it — inlined next to the standard groups' code the wide path cost
// them registers (k_lm_schur<0> 24.7 -> 26.2 us at configs[2], profiles/r04_*), as a separate instantiation it costs nothing
template <int MODE, bool WIDE_OK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_lm_schur(DevBuf d, LmView lv, int state, Robust rb, ChainView cv, int nlead, int spec) {
    if (spec && !d.ctrl->accepted) return;      // enqueued behind the deciding launch: runs only for the state that was accepted
    __shared__ LmLds S;
    extern __shared__ __attribute__((aligned(16))) double s_dyn_lm[];      // LmAcc (36 KB) | a chain segment's staging (48 KB)
    int b = blockIdx.x;
    if (MODE == 0 && b < nlead) { chain_elim_segment<true>(d, cv, b, *reinterpret_cast<ChainElimLds*>(s_dyn_lm)); return; }
    b -= nlead;
    if (MODE == 1 && b >= lv.ngrp) {      // first iteration of a call: the pose-side edges (with Jacobians) ride behind the groups instead of in a launch of their own
        const int m = b - lv.ngrp;
        if (m < d.M) pose_edge_block<true, 256>(d, state, rb, m, threadIdx.x);
        else prior_block<true>(d, state);
        return;
    }
    LmAcc& A4 = *reinterpret_cast<LmAcc*>(s_dyn_lm);
    const int kind = lv.grp[b].kind;      // bit 0: lines, bit 1: wide (landmarks seen from 9 .. 16 keyframes)
    if (kind == 0) lm_schur_group<false, MODE>(d, lv, b, state, rb, S, A4);
    else if (kind == 1) lm_schur_group<true, MODE>(d, lv, b, state, rb, S, A4);
    else if (WIDE_OK && kind == 2) lm_schur_group_wide<false, MODE>(d, lv, b, state, rb, S, A4);
    else if (WIDE_OK) lm_schur_group_wide<true, MODE>(d, lv, b, state, rb, S, A4);
}
// launch D: [blocks assembling the part of the system no landmark touches | gather blocks: pose-pair blocks and right-hand-side rows =
// pose-side terms + the groups' parts]
// (+ since round 4, behind them: one workgroup per tile of the chain Schur complement's W^T W — it needs the chain elimination's W from the
// launch before, not this launch's output — left in dd.wtw for k_chain_schur<PRE>)
__global__ __launch_bounds__(256) void k_lm_gather(DevBuf d, LmView lv, int nasm, int add_lambda, int spec, int diag, ChainView cv, DevBuf dd, int ngather) {
    if (spec && !d.ctrl->accepted) return;
    __shared__ double red[256];
    int b = blockIdx.x;
    if (b >= ngather) {
        int ta, tb;
        chain_schur_tile_of(cv, dd, b - ngather, ta, tb);
        const double4v acc = chain_wtw_tile(cv, ta, tb);
        *reinterpret_cast<double4v*>(dd.wtw + (size_t)(b - ngather) * 1024 + 4 * threadIdx.x) = acc;
        return;
    }
    if (b < nasm) { lm_assemble_rest(d, lv, add_lambda, b, nasm, threadIdx.x, 256); return; }
    b -= nasm;
    if (diag) lm_gather_diag(d, lv, b, threadIdx.x, red);
    else lm_gather_part(d, lv, add_lambda, b, threadIdx.x, red);
}
// launch A: [chain back-substitution segments + the keyframes' update | groups: landmark back-substitution, update, trial residuals | the
// IMU / prior edges of the trial state — with Jacobians: into the idle accumulators]; the workgroup that finishes last takes the LM
// decision (da.fuse).  The pose-side blocks need the trial keyframes of the segments in front (lead_wait); they come LAST so that they
// fill the slots the second round of groups leaves free instead of delaying the first: their serial per-edge math (16 us) and the
// chain segments (17 us) then run in the shadow of the landmark pass — as launches of their own they cost 22 us / 17 us of an iteration.
template <bool JAC, bool WIDE_OK>
__global__ __launch_bounds__(LMB) void k_lm_trial(DevBuf d, LmView lv, int cur, int trial, Robust rb, ChainView cv, const double* xd, int nlead, int npose, unsigned back_target, DecideArgs da) {
    static_assert(LMB == 256, "chain_back_segment's thread layout");
    __shared__ double s4[4];
    int b = blockIdx.x;
    // With chain segments in front, the IMU edge blocks wait for nothing (they form their trial states themselves) and are the launch's longest
    // workgroups (a local back-substitution, then the serial per-edge math): they are dispatched right behind the segments, ahead of the groups.
    // The prior block still waits for the segments and stays LAST.  Without segments the old order holds (the edge blocks wait for group 0).
    // Only while the launch is one round of workgroups (configs[2]: 280; ab_opts 0.1445 -> 0.1373 ms per trial with both changes): at
    // configs[4] (1183 workgroups, two rounds of groups) 200 edge blocks at the head delay the first round of groups — 0.4257 -> 0.4310 —
    // and stay at the tail, where they fill the slots the second round leaves free.
    const int nimu_first = (nlead > 0 && npose > 0 && cv.imu_loc && gridDim.x <= 512) ? min(npose, d.M) : 0;
    if (b >= nlead && b < nlead + nimu_first) {
        const LocalStates loc{&cv, xd, cur};
        pose_edge_block<JAC, 256>(d, trial, rb, b - nlead, threadIdx.x, nullptr, 0, s4, nullptr, &loc);
        if (da.fuse) trial_arrive(d, da, lv.ngrp, s4);
        return;
    }
    if (b >= nlead + nimu_first) b -= nimu_first;
    if (b < nlead) {
        chain_back_segment(d, cv, xd, b, cur, trial);
        lead_done(d.back_cnt);
    } else if (b < nlead + lv.ngrp) {
        __shared__ LmLds S;
        const int g = b - nlead;
        if (g == 0 && nlead == 0) {      // no chain segments: this group stands in for them (keyframe part of update(), one count)
            for (int k = threadIdx.x; k < d.K; k += LMB) update_kf_one(d, cur, trial, k);
            lead_done(d.back_cnt);
        }
        const int kind = lv.grp[g].kind;
        if (kind == 0) lm_trial_group<false, false>(d, lv, g, cur, trial, rb, cv, xd, nlead != 0, S);
        else if (kind == 1) lm_trial_group<true, false>(d, lv, g, cur, trial, rb, cv, xd, nlead != 0, S);
        else if (WIDE_OK && kind == 2) lm_trial_group<false, true>(d, lv, g, cur, trial, rb, cv, xd, nlead != 0, S);
        else if (WIDE_OK) lm_trial_group<true, true>(d, lv, g, cur, trial, rb, cv, xd, nlead != 0, S);
    } else {
        const int m = b - nlead - lv.ngrp + nimu_first;      // (the edge blocks dispatched early are not in this range)
        const LeadWait lw{d.back_cnt, back_target, &d.ctrl->sync_fail};
        if (m < d.M) pose_edge_block<JAC, 256>(d, trial, rb, m, threadIdx.x, nullptr, 0, s4, &lw);
        else prior_block<JAC>(d, trial, nullptr, 0, s4, &lw);      // (the prior touches many keyframes: it waits for the segments as before)
    }
    if (da.fuse) trial_arrive(d, da, lv.ngrp, s4);
}

// -------------------------------------------------------------------------------------------------
// launchers
// -------------------------------------------------------------------------------------------------
void launch_lm_schur(const DevBuf& d, const LmView& lv, int state, const Robust& rb, bool diag_pass, const ChainView* lead, bool spec, hipStream_t s, bool with_pose_edges) {
    const size_t sh = sizeof(LmAcc) > sizeof(ChainElimLds) ? sizeof(LmAcc) : sizeof(ChainElimLds);      // a chain segment's staging shares the dynamic LDS
    const bool wide = lv.wmax > LMF_W;
    if (diag_pass) {
        const void* fn = wide ? reinterpret_cast<const void*>(k_lm_schur<1, true>) : reinterpret_cast<const void*>(k_lm_schur<1, false>);
        if (ensure_dyn_lds(fn, (int)sh) != hipSuccess) return;      // surfaces at the caller's hipGetLastError
        const int npose = with_pose_edges ? d.M + (d.pr_nv > 0 ? 1 : 0) : 0;
        if (wide) hipLaunchKernelGGL((k_lm_schur<1, true>), dim3(lv.ngrp + npose), dim3(256), sh, s, d, lv, state, rb, ChainView{}, 0, 0);
        else hipLaunchKernelGGL((k_lm_schur<1, false>), dim3(lv.ngrp + npose), dim3(256), sh, s, d, lv, state, rb, ChainView{}, 0, 0);
        return;
    }
    const void* fn = wide ? reinterpret_cast<const void*>(k_lm_schur<0, true>) : reinterpret_cast<const void*>(k_lm_schur<0, false>);
    if (ensure_dyn_lds(fn, (int)sh) != hipSuccess) return;
    const int nlead = lead ? lead->nseg : 0;
    if (wide) hipLaunchKernelGGL((k_lm_schur<0, true>), dim3(lv.ngrp + nlead), dim3(256), sh, s, d, lv, state, rb, lead ? *lead : ChainView{}, nlead, spec ? 1 : 0);
    else hipLaunchKernelGGL((k_lm_schur<0, false>), dim3(lv.ngrp + nlead), dim3(256), sh, s, d, lv, state, rb, lead ? *lead : ChainView{}, nlead, spec ? 1 : 0);
}
void launch_lm_gather(const DevBuf& d, const LmView& lv, bool diag_pass, bool add_lambda, bool spec, hipStream_t s, const ChainView* wtw_cv, const DevBuf* wtw_dd) {
    int nasm = 0;
    if (!diag_pass) {
        const size_t n = (size_t)lv.nalist2 + d.ld;
        nasm = (int)((n + 4 * 256 - 1) / (4 * 256));
        if (nasm > 1024) nasm = 1024;
    }
    const int nb = diag_pass ? lv.nrow : lm_gather_blocks(lv);
    int ntile = 0;
    if (wtw_cv && wtw_dd && wtw_dd->wtw) { const int T = wtw_cv->Pdpad / 32; ntile = T * (T + 1) / 2 + T; }
    if (nb + nasm + ntile == 0) return;
    hipLaunchKernelGGL(k_lm_gather, dim3(nasm + nb + ntile), dim3(256), 0, s, d, lv, nasm, add_lambda ? 1 : 0, spec ? 1 : 0, diag_pass ? 1 : 0,
                       ntile ? *wtw_cv : ChainView{}, ntile ? *wtw_dd : DevBuf{}, nasm + nb);
}
void launch_lm_trial(const DevBuf& d, const LmView& lv, int cur, int trial, bool jac, const Robust& rb, const ChainView* lead, const double* xd, unsigned back_target, bool with_pose_edges, const DecideFusion* df, hipStream_t s) {
    const int nlead = lead ? lead->nseg : 0, npose = with_pose_edges ? d.M + (d.pr_nv > 0 ? 1 : 0) : 0;      // (a sharded run: rank 0 owns the pose-side edges)
    DecideArgs da{};
    if (df) { da.lp = df->lp; da.red = df->red; da.mail = df->mail; da.seq = df->seq; da.nblk_lm = lv.ngrp; da.fuse = 1; }
    const dim3 grid(nlead + lv.ngrp + npose);
    const bool wide = lv.wmax > LMF_W;
    const ChainView cvl = lead ? *lead : ChainView{};
    if (jac && wide) hipLaunchKernelGGL((k_lm_trial<true, true>), grid, dim3(LMB), 0, s, d, lv, cur, trial, rb, cvl, xd, nlead, npose, back_target, da);
    else if (jac) hipLaunchKernelGGL((k_lm_trial<true, false>), grid, dim3(LMB), 0, s, d, lv, cur, trial, rb, cvl, xd, nlead, npose, back_target, da);
    else if (wide) hipLaunchKernelGGL((k_lm_trial<false, true>), grid, dim3(LMB), 0, s, d, lv, cur, trial, rb, cvl, xd, nlead, npose, back_target, da);
    else hipLaunchKernelGGL((k_lm_trial<false, false>), grid, dim3(LMB), 0, s, d, lv, cur, trial, rb, cvl, xd, nlead, npose, back_target, da);
}
// sharded runs of the fused passes: the local sums go out for the all-reduce instead of being consumed by the control kernel
void launch_reduce_n(const DevBuf& d, bool owns_pose_edges, double* red, int nred, hipStream_t s) {
    hipLaunchKernelGGL(k_reduce, dim3(1), dim3(256), 0, s, d, nred, nred, owns_pose_edges ? 1 : 0, red);
}
void launch_posediag(const DevBuf& d, hipStream_t s) {
    hipLaunchKernelGGL(k_posediag, dim3((d.ld + 255) / 256), dim3(256), 0, s, d);
    hipLaunchKernelGGL(k_posediag_kf, dim3((d.K * 6 + 255) / 256), dim3(256), 0, s, d);
}
// first iteration of a call on the fused path, one GPU: what k_lm_gather's diagonal pass + k_lambda_init did in two launches — the groups'
// diagonal parts summed per keyframe (fixed order), the pose diagonal, chi2, computeLambdaInit
__global__ __launch_bounds__(256) void k_lm_lambda_init(DevBuf d, LmView lv, LmParams lp, double* red, int iteration, int nred) {
    __shared__ double s4[4];
    __shared__ double s_part[256];
    reduce_inline(d, nred, nred, red, s4);
    for (int r = threadIdx.x; r < d.ld; r += 256) d.posediag[r] = (r < d.P) ? d.Himu[(size_t)r * d.ld + r] + d.Hconst[(size_t)r * d.ld + r] : 0.0;
    __syncthreads();
    // per keyframe and pose dimension: the groups' diagonal parts, the contributions dealt to nch lane groups (a small window has a
    // dozen keyframes and hundreds of groups: one lane per entry walked them all, 40 us) and added in a fixed order
    const int nent = lv.nrow * 6;
    const int nch = nent > 0 && nent <= 128 ? 256 / nent : 1;
    for (int base = 0; base < nent; base += 256) {
        const int idx = base + (int)threadIdx.x % (nch > 1 ? nent : 256), ch = nch > 1 ? (int)threadIdx.x / nent : 0;
        double v = 0.0;
        const bool live = idx < nent && ch < nch;
        int k = 0, c = 0;
        if (live) {
            k = idx / 6; c = idx - 6 * k;
            for (int q = lv.row_start[k] + ch; q < lv.row_start[k + 1]; q += nch) {
                const int src = lv.row_src[q];
                v += lv.part[(size_t)(src / lv.wmax) * lv.part_stride + lv.npair * 36 + (src % lv.wmax) * 12 + c];
            }
        }
        if (nch > 1) {
            s_part[threadIdx.x] = v;
            __syncthreads();
            if (live && ch == 0) { v = 0.0; for (int q = 0; q < nch; ++q) v += s_part[q * nent + idx]; }
        }
        if (live && ch == 0) {
            const int kf = lv.row_kf[k], o = d.kf_off_pvr[kf];
            d.kfdiag[kf * 6 + c] = v;
            if (o >= 0) d.posediag[o + pmap(c)] += v;
        }
    }
    __syncthreads();
    double md = 0.0;
    for (int r = threadIdx.x; r < d.P; r += 256) md = fmax(md, fabs(d.posediag[r]));
    double mx = block_max_256(md, s4);
    if (threadIdx.x == 0) {
        Ctrl* c = d.ctrl;
        c->current_chi = red[0];
        c->iteration = iteration;
        c->trial = 0;
        mx = fmax(mx, red[2]);
        c->maxdiag = mx;
        c->lambda = lp.user_lambda > 0 ? lp.user_lambda : lp.tau * mx;
        c->ni = 2.0;
    }
}
void launch_lambda_init_n(const DevBuf& d, const LmView& lv, const LmParams& lp, double* red, int iteration, int nred, hipStream_t s) {
    hipLaunchKernelGGL(k_lm_lambda_init, dim3(1), dim3(256), 0, s, d, lv, lp, red, iteration, nred);
}
__global__ __launch_bounds__(256) void k_lm_level_sync(DevBuf d, LmView lv) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < d.E) lv.ob_level_g[e] = d.ob_level[lv.ob_orig[e]];
}
__global__ __launch_bounds__(256) void k_lm_chi_sync(DevBuf d, LmView lv) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < d.E && lv.ob_level_g[e] == 0) d.ob_chi2[lv.ob_orig[e]] = lv.ob_chi_g[e];      // (a gated observation keeps the value it was gated on, as on the record-based path)
}
void launch_lm_chi_sync(const DevBuf& d, const LmView& lv, hipStream_t s) {
    if (d.E) hipLaunchKernelGGL(k_lm_chi_sync, dim3((d.E + 255) / 256), dim3(256), 0, s, d, lv);
}
void launch_lm_level_sync(const DevBuf& d, const LmView& lv, hipStream_t s) {
    if (d.E) hipLaunchKernelGGL(k_lm_level_sync, dim3((d.E + 255) / 256), dim3(256), 0, s, d, lv);
}
void launch_decide_n(const DevBuf& d, const LmParams& lp, double* red, int nred, Mailbox* mail, unsigned long long seq, hipStream_t s) {
    hipLaunchKernelGGL(k_decide, dim3(1), dim3(256), 0, s, d, lp, red, 1, nred, nred, mail, seq);
}
int edge_blocks(const DevBuf& d) { return (d.E + 255) / 256; }
static int lm_blocks(const DevBuf& d) { return (d.L + LML - 1) / LML; }

// with_pose_edges: this rank owns the IMU / prior edges; they are evaluated by extra blocks of the same launch
void launch_linearize(const DevBuf& d, int state, bool jac, const Robust& rb, bool with_pose_edges, hipStream_t s, bool spec, const DecideFusion* df) {
    const int nb = d.E ? edge_blocks(d) : 0;
    const int pose_blocks = with_pose_edges ? d.M + (d.pr_nv > 0 ? 1 : 0) : 0;
    if (nb + pose_blocks == 0) return;
    const size_t sh = (size_t)d.K * KFCAM_STRIDE * sizeof(double);
    DecideArgs da{};
    if (df) { da.lp = df->lp; da.red = df->red; da.mail = df->mail; da.seq = df->seq; da.nblk_lm = d.L ? lm_blocks(d) : 0; da.fuse = 1; }
    if (jac) hipLaunchKernelGGL(k_linearize<true>, dim3(nb + pose_blocks), dim3(256), sh, s, d, state, rb, nb, spec ? 1 : 0, da);
    else hipLaunchKernelGGL(k_linearize<false>, dim3(nb + pose_blocks), dim3(256), sh, s, d, state, rb, nb, 0, da);
}
void launch_pose_edges(const DevBuf& d, int state, bool jac, const Robust& rb, bool owns, hipStream_t s) {
    if (!owns) return;
    if (d.M > 0) {
        if (jac) hipLaunchKernelGGL(k_pose_edges<true>, dim3(d.M), dim3(256), 0, s, d, state, rb);
        else hipLaunchKernelGGL(k_pose_edges<false>, dim3(d.M), dim3(256), 0, s, d, state, rb);
    }
    if (d.pr_nv > 0) {
        if (jac) hipLaunchKernelGGL(k_prior<true>, dim3(1), dim3(256), 0, s, d, state);
        else hipLaunchKernelGGL(k_prior<false>, dim3(1), dim3(256), 0, s, d, state);
    }
}
// fuse_dinv_assemble: lambda is known (not the first iteration): the damped landmark inverses are formed in the same
// pass and extra blocks assemble the pose-side system (returns true if it did, so the caller skips k_assemble)
bool launch_landmark_hll(const DevBuf& d, int state, bool fuse_dinv_assemble, bool add_lambda, hipStream_t s, bool spec) {
    if (!d.L) return false;
    const size_t sh = (size_t)d.K * KFCAM_STRIDE * sizeof(double);
    const int nb = lm_blocks(d);
    if (fuse_dinv_assemble) {
        const size_t n = (size_t)(d.Ppad + TILE) * d.ld;
        int ab = (int)((n + 4 * LMB - 1) / (4 * LMB));
        if (ab > 4096) ab = 4096;
        hipLaunchKernelGGL(k_landmark_hll<true>, dim3(nb + ab), dim3(LMB), sh, s, d, state, nb, add_lambda ? 1 : 0, spec ? 1 : 0);
        return true;
    }
    hipLaunchKernelGGL(k_landmark_hll<false>, dim3(nb), dim3(LMB), sh, s, d, state, nb, 0, 0);
    return false;
}
void launch_kfdiag(const DevBuf& d, int state, bool with_posediag, hipStream_t s) {
    if (d.nchunks) hipLaunchKernelGGL(k_kfdiag, dim3(d.nchunks), dim3(256), 0, s, d, state);
    if (!with_posediag) return;      // one GPU: k_lambda_init forms the pose diagonal itself
    hipLaunchKernelGGL(k_posediag, dim3((d.ld + 255) / 256), dim3(256), 0, s, d);
    hipLaunchKernelGGL(k_posediag_kf, dim3((d.K * 6 + 255) / 256), dim3(256), 0, s, d);
}
void launch_landmark_dinv(const DevBuf& d, hipStream_t s) {
    if (d.L) hipLaunchKernelGGL(k_landmark_dinv, dim3((d.L + LMB - 1) / LMB), dim3(LMB), 0, s, d);
}
void launch_assemble(const DevBuf& d, bool add_lambda, hipStream_t s) {
    const size_t n = (size_t)(d.Ppad + TILE) * d.ld;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_assemble, dim3(blocks), dim3(256), 0, s, d, add_lambda ? 1 : 0);
}
void launch_schur_pairs(const DevBuf& d, int state, const ChainView* lead, hipStream_t s) {
    const int nlead = lead ? (lead->nseg + SCHUR_XCD - 1) / SCHUR_XCD * SCHUR_XCD : 0;      // keeps chunk index = block index (mod XCD count)
    if (d.nchunks + nlead) hipLaunchKernelGGL(k_schur_pairs, dim3(d.nchunks + nlead), dim3(256), 0, s, d, state, lead ? *lead : ChainView{}, nlead);
}
void launch_backsub(const DevBuf& d, int cur, int trial, const ChainView* lead, const double* xd, hipStream_t s) {
    const int nlead = lead ? lead->nseg : 0, nb = d.L ? lm_blocks(d) : 0;
    if (nb + nlead) hipLaunchKernelGGL(k_backsub, dim3(nb + nlead), dim3(LMB), (size_t)d.K * (KFCAM_STRIDE + 6) * sizeof(double), s, d, cur, trial, lead ? *lead : ChainView{}, xd, nlead);
}
void launch_update_kf(const DevBuf& d, int cur, int trial, hipStream_t s) {
    if (d.L) return;   // done by block 0 of k_backsub whenever there are landmarks (or by the chain workgroups riding in it)
    hipLaunchKernelGGL(k_update_kf, dim3((d.K + 63) / 64), dim3(64), 0, s, d, cur, trial);
}
size_t tri_packed_size(const DevBuf& d) { const size_t a = (size_t)(d.Ppad >> 5); return 512 * a * (a + 1) + 2 * (size_t)d.Ppad; }
// Sharded runs, structural version of the exchange: only the entries of the lower triangle that CAN be non-zero before the
// factorisation travel (pose x pose for the landmarks' Schur terms, the IMU / prior blocks, the diagonal: d.xlist, built at
// upload), plus the two right-hand-side rows.  buf = [entries | row Ppad | row Ppad + 1].  Unpacking mirrors every entry
// (the diagonal tiles are read whole by the factorisation).
__global__ __launch_bounds__(256) void k_list_pack(DevBuf d, double* buf, int unpack) {
    const int n = d.nxlist;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < n + 2 * d.Ppad; k += gridDim.x * 256) {
        if (k < n) {
            const int idx = d.xlist[k];
            if (unpack) { const double v = buf[k]; d.sys[idx] = v; const int r = idx / d.ld, c = idx % d.ld; d.sys[(size_t)c * d.ld + r] = v; }
            else buf[k] = d.sys[idx];
        } else {
            const int q = k - n, row = d.Ppad + q / d.Ppad, c = q % d.Ppad;
            if (unpack) d.sys[(size_t)row * d.ld + c] = buf[k];
            else buf[k] = d.sys[(size_t)row * d.ld + c];
        }
    }
}
size_t list_packed_size(const DevBuf& d) { return (size_t)d.nxlist + 2 * (size_t)d.Ppad; }
void launch_list_pack(const DevBuf& d, double* buf, bool unpack, hipStream_t s) {
    const size_t n = list_packed_size(d);
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_list_pack, dim3(blocks), dim3(256), 0, s, d, buf, unpack ? 1 : 0);
}
void launch_tri_pack(const DevBuf& d, double* buf, bool unpack, hipStream_t s) {
    hipLaunchKernelGGL(k_tri_pack, dim3(d.Ppad + 2), dim3(256), 0, s, d, buf, unpack ? 1 : 0);
}
void launch_reduce(const DevBuf& d, bool owns_pose_edges, double* red, hipStream_t s) {
    hipLaunchKernelGGL(k_reduce, dim3(1), dim3(256), 0, s, d, d.E ? edge_blocks(d) : 0, d.L ? lm_blocks(d) : 0, owns_pose_edges ? 1 : 0, red);
}
// fused: the per-block partials are summed inside the control kernel (single-GPU path, no exchange in between)
void launch_lambda_init2(const DevBuf& d, const LmParams& lp, double* red, bool first_iter, int iteration, bool fused, bool keep_chi, hipStream_t s) {
    hipLaunchKernelGGL(k_lambda_init, dim3(1), dim3(256), 0, s, d, lp, red, first_iter ? 1 : 0, iteration, fused ? 1 : 0, keep_chi ? 1 : 0, d.E ? edge_blocks(d) : 0, d.L ? lm_blocks(d) : 0);
}
__global__ void k_ctrl_reset(Ctrl* c, int* trace_n) {
    Ctrl z;
    memset(&z, 0, sizeof z);
    z.solver_ok = 1; z.ni = 2.0;
    *c = z; *trace_n = 0;
}
void launch_ctrl_reset(const DevBuf& d, hipStream_t s) { hipLaunchKernelGGL(k_ctrl_reset, dim3(1), dim3(1), 0, s, d.ctrl, d.trace_n); }
void launch_decide(const DevBuf& d, const LmParams& lp, double* red, bool fused, Mailbox* mail, unsigned long long seq, hipStream_t s) {
    hipLaunchKernelGGL(k_decide, dim3(1), dim3(256), 0, s, d, lp, red, fused ? 1 : 0, d.E ? edge_blocks(d) : 0, d.L ? lm_blocks(d) : 0, mail, seq);
}
void launch_gate(const DevBuf& d, int state, double thresh, hipStream_t s) {
    if (d.E == 0) return;
    const size_t sh = (size_t)d.K * KFCAM_STRIDE * sizeof(double);
    hipLaunchKernelGGL(k_gate, dim3(edge_blocks(d)), dim3(256), sh, s, d, state, thresh, (uint8_t*)nullptr, 1);
}
void launch_cull(const DevBuf& d, int state, double thresh, uint8_t* bad, hipStream_t s) {
    if (d.E == 0) return;
    hipLaunchKernelGGL(k_cull, dim3(edge_blocks(d)), dim3(256), (size_t)d.K * KFCAM_STRIDE * sizeof(double), s, d, state, thresh, bad);
}
void launch_depth(const DevBuf& d, int state, uint8_t* out, hipStream_t s) {
    if (d.E == 0) return;
    const size_t sh = (size_t)d.K * KFCAM_STRIDE * sizeof(double);
    hipLaunchKernelGGL(k_gate, dim3(edge_blocks(d)), dim3(256), sh, s, d, state, 0.0, out, 0);
}

}  // namespace plba
