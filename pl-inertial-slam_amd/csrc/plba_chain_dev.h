// plba_chain_dev.h — device body of the chain-variable elimination (see plba_chain.hip for the scheme).  A header because
// the segment workgroups also ride in front of the Schur-pair launch (plba_kernels.hip::k_schur_pairs).
#pragma once
#include "plba_internal.h"

namespace plba {

typedef double double4v __attribute__((ext_vector_type(4)));

namespace chain_detail {
__device__ __forceinline__ double lane_bcast(double v, int l) {   // lane l (compile time) -> uniform
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rsqrt_full(double d) {          // v_rsq_f64 + two Newton steps
    double y = __builtin_amdgcn_rsq(d);
    y = y * fma(-0.5 * d * y, y, 1.5);
    y = y * fma(-0.5 * d * y, y, 1.5);
    return y;
}
template <typename T> __device__ __forceinline__ void lds_store(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
template <typename T> __device__ __forceinline__ T lds_load(const T* p) { return __hip_atomic_load(const_cast<T*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ double sym_at(const double* sys, int ld, int i, int j) { return i >= j ? sys[(size_t)i * ld + j] : sys[(size_t)j * ld + i]; }
constexpr int SPIN_MAX = 1 << 20;
constexpr int SEGMAX = CHAIN_SEG;          // block steps per workgroup; also the depth of the LDS ring (no wrap, no back-pressure)
constexpr int NSLOT = CHAIN_NSLOT;         // dense columns a keyframe can own: 6 pose + 9 separator chain dimensions
constexpr int COLW = 3;                    // column waves: <= 192 coupled columns per segment (window of SEG + 2 keyframes + rhs)
constexpr int ELIM_THREADS = 64 * (1 + COLW);
}  // namespace chain_detail
using namespace chain_detail;

// one segment (workgroup of ELIM_THREADS threads); g = segment index.  Called from k_chain_elim and, on one GPU, from the
// leading workgroups of the Schur-pair launch (the two are independent: chain blocks carry no landmark coupling).
struct ChainElimLds {          // LDS of one segment's elimination (a struct so that a host kernel can overlay it with its own)
    double sLsub[SEGMAX][90], sLinv[SEGMAX][90];   // published chain factors; rows of 9 padded to 10: aligned pairs
    double sCg[SEGMAX][162];                       // C_ii (81) | C_{i+1,i} (81)
    double sBg[SEGMAX][3 * NSLOT * 9];             // B_i against the dense columns of positions p-1, p, p+1
    double sRhs[SEGMAX][9];
    double sA[81];
    int s_step, s_bad;
};
// ACC: read the pose-side system straight from what it is assembled from — Himu + Hconst (+ lambda on the diagonal), bimu + bprior —
// instead of from d.sys: chain blocks and their couplings carry no landmark term, so the segments can run in the SAME launch that
// assembles d.sys (the fused landmark passes, k_lm_gather) instead of in one behind it.
template <bool ACC>
__device__ __forceinline__ double chain_sys_at(const DevBuf& d, int ld, int i, int j, double lambda) {
    const size_t idx = i >= j ? (size_t)i * ld + j : (size_t)j * ld + i;
    if (!ACC) return d.sys[idx];
    return d.Himu[idx] + d.Hconst[idx] + ((i == j && i < d.P) ? lambda : 0.0);
}
template <bool ACC = false>
__device__ __forceinline__ void chain_elim_segment(const DevBuf& d, const ChainView& cv, const int g, ChainElimLds& LDS) {
    auto& sLsub = LDS.sLsub; auto& sLinv = LDS.sLinv; auto& sCg = LDS.sCg; auto& sBg = LDS.sBg; auto& sRhs = LDS.sRhs; auto& sA = LDS.sA;
    int& s_step = LDS.s_step; int& s_bad = LDS.s_bad;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ld = d.ld;
    const int i0 = cv.seg_start[g], i1 = cv.seg_start[g + 1], n = i1 - i0;      // eliminated blocks of this segment
    if (threadIdx.x == 0) { s_step = 0; s_bad = 0; }
    const double lambda_acc = ACC ? d.ctrl->lambda : 0.0;
#ifdef PLBA_STAMPS_LM
    unsigned long long ets[12] = {0,0,0,0,0,0,0,0,0,0,0,0}; ets[0] = __builtin_readcyclecounter();
#define ESTAMP(i) ets[i] = __builtin_readcyclecounter()
#else
#define ESTAMP(i) do {} while (0)
#endif
    // ---- stage everything the steps read: one level of host-built indices (cv.esrc: a fixed-size region per segment, so these loads wait
    // for nothing but the kernel arguments), every gather in flight at once ----------------------------------------------------------------
    {
        constexpr int NBB = 3 * NSLOT * 9, NCB = SEGMAX * (162 + NBB), REG = NCB + SEGMAX * 9;
        const int32_t* src = cv.esrc + (size_t)g * REG;
        double* flatC = &sCg[0][0];      // [SEGMAX][162]
        double* flatB = &sBg[0][0];      // [SEGMAX][3 * NSLOT * 9]
        constexpr int PER = (NCB + ELIM_THREADS - 1) / ELIM_THREADS;
        int32_t code[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) { const int idx = q * ELIM_THREADS + threadIdx.x; code[q] = idx < NCB ? src[idx] : -3; }
        double va[PER], vb[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {      // (every gather issued before the first one is used)
            const int32_t c = code[q];
            const size_t off = c >= 0 ? (size_t)(c & 0x3fffffff) : 0;
            va[q] = ACC ? d.Himu[off] : d.sys[off];
            vb[q] = ACC ? d.Hconst[off] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int idx = q * ELIM_THREADS + threadIdx.x;
            const int32_t c = code[q];
            if (c == -3) continue;
            const double v = c >= 0 ? va[q] + vb[q] + ((ACC && (c >> 30)) ? lambda_acc : 0.0) : (c == -2 ? 1.0 : 0.0);
            if (idx < SEGMAX * 162) flatC[idx] = v; else flatB[idx - SEGMAX * 162] = v;
        }
        for (int idx = threadIdx.x; idx < SEGMAX * 9; idx += ELIM_THREADS) {
            const int gi = src[NCB + idx];
            if (gi == -3) continue;
            sRhs[idx / 9][idx % 9] = gi < 0 ? 0.0 : (ACC ? d.bimu[gi] + d.bprior[gi] : d.sys[(size_t)d.Ppad * ld + gi]);
        }
    }
    __syncthreads();
    ESTAMP(1);
    if (wv == 0) {
        // ---- the chain wave ----------------------------------------------------------------------------------------------
        bool bad = false;
        const int e0 = lane, e1 = lane + 64;              // the (up to) two entries of a 9 x 9 block this lane owns
        const int r0 = e0 / 9, c0 = e0 % 9, r1 = (e1 < 81 ? e1 : 0) / 9, c1 = (e1 < 81 ? e1 : 0) % 9;
        for (int i = 0; i < n; ++i) {
            const bool has_next = i + 1 < n;
            const double* Cg = sCg[i];
            double v0 = Cg[e0], v1 = Cg[e1 < 81 ? e1 : 0];
            double cs0[9], cs1[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) { cs0[t] = Cg[81 + r0 * 9 + t]; cs1[t] = Cg[81 + r1 * 9 + t]; }
            if (i > 0) {       // (a) C_ii - L_{i,i-1} L_{i,i-1}^T
                const double* Lp = sLsub[i - 1];
#pragma unroll
                for (int t = 0; t < 9; ++t) { v0 = fma(-Lp[r0 * 10 + t], Lp[c0 * 10 + t], v0); v1 = fma(-Lp[r1 * 10 + t], Lp[c1 * 10 + t], v1); }
            }
            sA[e0] = v0;
            if (e1 < 81) sA[e1] = v1;
            // (b) 9 x 9 Cholesky, lane = row (lanes >= 9 shadow row 8; their values are never used)
            const int row = lane < 9 ? lane : 8;
            double a[9], rs[9];
#pragma unroll
            for (int c = 0; c < 9; ++c) a[c] = sA[row * 9 + c];
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const double pj = lane_bcast(a[j], j);
                const bool bj = !(pj > 0.0);
                bad = bad || bj;
                const double r = rsqrt_full(bj ? 1.0 : pj);
                rs[j] = r;
                const double lij = a[j] * r;
                a[j] = lij;
#pragma unroll
                for (int c = j + 1; c < 9; ++c) a[c] = fma(-lij, lane_bcast(lij, c), a[c]);
            }
            // (c) L_ii^-1, lane = column (forward substitution on the identity)
            double x[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) x[t] = (t == lane) ? 1.0 : 0.0;
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                x[j] *= rs[j];
#pragma unroll
                for (int t = j + 1; t < 9; ++t) x[t] = fma(-lane_bcast(a[j], t), x[j], x[t]);     // L[t][j] lives in lane t, register j
            }
            double* Li = sLinv[i];
            if (lane < 9) {
#pragma unroll
                for (int t = 0; t < 9; ++t) { Li[t * 10 + lane] = x[t]; cv.Ldinv[(size_t)(i0 + i) * 81 + t * 9 + lane] = x[t]; }
                Li[lane * 10 + 9] = 0.0;
            }
            // (d) L_{i+1,i} = C_{i+1,i} L_ii^-T
            if (has_next) {
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int t = 0; t < 9; ++t) { s0 = fma(cs0[t], Li[c0 * 10 + t], s0); s1 = fma(cs1[t], Li[c1 * 10 + t], s1); }
                double* Lo = sLsub[i];
                Lo[r0 * 10 + c0] = s0;
                cv.Lsub[(size_t)(i0 + i) * 81 + e0] = s0;
                if (e1 < 81) { Lo[r1 * 10 + c1] = s1; cv.Lsub[(size_t)(i0 + i) * 81 + e1] = s1; }
                if (lane < 9) Lo[lane * 10 + 9] = 0.0;
            }
            asm volatile("" ::: "memory");
            lds_store(&s_step, i + 1);        // a wave's LDS operations execute in order: the data above is visible first
#ifdef PLBA_STAMPS_LM
            if (i < 8) ets[2 + i] = __builtin_readcyclecounter();
#endif
        }
        if (bad && lane == 0) d.ctrl->solver_ok = 0;
#ifdef PLBA_STAMPS_LM
        if (ACC && lane == 0 && g == 1) { for (int q = 0; q < 10; ++q) d.dbgbuf[64 + q] = (double)(ets[q] - ets[0]); d.dbgbuf[75] = (double)n; }
#endif
        return;
    }
    // ---- column lanes: w_i = L_ii^-1 (B_i - L_{i,i-1} w_{i-1}) over the dense columns of the segment's window + the rhs -------
    const int wlo = cv.seg_col[2 * g], whi = cv.seg_col[2 * g + 1];      // dense columns [wlo, whi) can couple to this segment
    const int lc = (wv - 1) * 64 + lane;                                  // local column; the one behind the window is the rhs
    const bool rhs = (lc == whi - wlo);
    const bool act = lc <= whi - wlo;
    const int col = rhs ? cv.Pd : wlo + (act ? lc : 0);
    const int cpos = (act && !rhs) ? cv.ppos[col] : 0, cslot = (act && !rhs) ? cv.pslot[col] : 0;
    double wp[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) wp[r] = 0.0;
    for (int i = 0; i < n; ++i) {
        // B is sparse: chain block i only couples to the dense columns of the neighbouring keyframe positions
        const int dl = cpos - cv.epos[i0 + i] + 1;
        const bool near = act && !rhs && dl >= 0 && dl <= 2;
        const double* src = rhs ? sRhs[i] : sBg[i] + ((near ? dl : 0) * NSLOT + cslot) * 9;
        double t[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) { const double v = src[r]; t[r] = (near || (rhs && act)) ? v : 0.0; }
        int spins = 0;
        while (lds_load(&s_step) <= i) { __builtin_amdgcn_s_sleep(1); if (++spins > SPIN_MAX) { s_bad = 1; break; } }
        asm volatile("" ::: "memory");
        const double2* Li2 = reinterpret_cast<const double2*>(sLinv[i]);
        if (i > 0) {
            const double2* Lp2 = reinterpret_cast<const double2*>(sLsub[i - 1]);      // row stride 5 pairs, pad column 0
#pragma unroll
            for (int r = 0; r < 9; ++r)
#pragma unroll
                for (int q2 = 0; q2 < 5; ++q2) {
                    const double2 l = Lp2[r * 5 + q2];
                    t[r] = fma(-l.x, wp[2 * q2], t[r]);
                    if (2 * q2 + 1 < 9) t[r] = fma(-l.y, wp[2 * q2 + 1], t[r]);
                }
        }
        double wn[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            wn[r] = 0.0;
#pragma unroll
            for (int q2 = 0; q2 <= r / 2; ++q2) {
                const double2 l = Li2[r * 5 + q2];            // L^-1 is lower triangular: entries beyond the diagonal are 0
                wn[r] = fma(l.x, t[2 * q2], wn[r]);
                if (2 * q2 + 1 <= r) wn[r] = fma(l.y, t[2 * q2 + 1], wn[r]);
            }
        }
        if (act) {
#pragma unroll
            for (int r = 0; r < 9; ++r) { wp[r] = wn[r]; cv.W[(size_t)((i0 + i) * 9 + r) * cv.Wld + col] = wn[r]; }
        }
    }
    if (s_bad && threadIdx.x == 64) d.ctrl->solver_ok = 0;
#ifdef PLBA_STAMPS_LM
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (ACC && threadIdx.x == 64 && g == 1) { d.dbgbuf[76] = (double)(__builtin_readcyclecounter() - ets[0]); d.dbgbuf[77] = (double)(ets[1] - ets[0]); }
#endif
}


// ---- k_chain_schur's two halves (shared with k_lm_gather since round 4) ---------------------------------------------------------------
// tile (ta, tb) of workgroup b: lower tiles in dd.cs_order (or natural order), then the right-hand-side row (ta == T)
__device__ __forceinline__ void chain_schur_tile_of(const ChainView& cv, const DevBuf& dd, int b, int& ta, int& tb) {
    const int T = cv.Pdpad / 32, ntri = T * (T + 1) / 2;
    if (dd.cs_order) { const int o = dd.cs_order[b]; ta = o >> 16; tb = o & 0xffff; }      // (the workgroups with the longest path — a tile factored on the spot — are dispatched first)
    else if (b < ntri) {
        ta = (int)((sqrt(8.0 * b + 1.0) - 1.0) * 0.5);
        while ((ta + 1) * (ta + 2) / 2 <= b) ++ta;
        while (ta * (ta + 1) / 2 > b) --ta;
        tb = b - ta * (ta + 1) / 2;
    } else { ta = T; tb = b - ntri; }
}
// acc = (W^T W)(tile ta, tb): only the rows of W that belong to segments whose column window meets both tiles are read.  C/D layout of the
// 16 x 16 sub-tile of wave (tr, tc): col = lane & 15, row = (lane >> 4) + 4 v
__device__ __forceinline__ double4v chain_wtw_tile(const ChainView& cv, int ta, int tb) {
    const int T = cv.Pdpad / 32;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const int tr = wv >> 1, tc = wv & 1;
    const bool rhs_row = (ta == T);
    const int acol = rhs_row ? cv.Pd : ta * 32 + tr * 16 + li;     // rhs row: every output row uses w_b; only row 0 is kept
    const int bcol = tb * 32 + tc * 16 + li;
    const bool a_ok = rhs_row || acol < cv.Pd, b_ok = bcol < cv.Pd;
    const double* Wa = cv.W + (a_ok ? acol : cv.Wld - 1);          // column Wld - 1 is zero padding (Wld >= Pd + 2)
    const double* Wb = cv.W + (b_ok ? bcol : cv.Wld - 1);
    int rlo = cv.trow[2 * tb], rhi = cv.trow[2 * tb + 1];
    if (!rhs_row) { rlo = max(rlo, cv.trow[2 * ta]); rhi = min(rhi, cv.trow[2 * ta + 1]); }
    double4v acc = (double4v){0.0, 0.0, 0.0, 0.0};
    constexpr int NCH = 3;
    for (int s0 = rlo; s0 < rhi; s0 += 32 * NCH) {
        double av[NCH][8], bv[NCH][8];
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = s0 + 32 * ch + 4 * u + lk;
                const bool in = r < rhi;
                av[ch][u] = in ? Wa[(size_t)r * cv.Wld] : 0.0;
                bv[ch][u] = in ? Wb[(size_t)r * cv.Wld] : 0.0;
            }
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ch][u], bv[ch][u], acc, 0, 0, 0);
    }
    return acc;
}

}  // namespace plba
