// plba_dense_dev.h — the look-ahead pipeline of the 32-wide factorisation (see plba_dense.hip): one workgroup factors a
// 32 x 32 tile held in LDS and publishes L(k,k) and L(k,k)^-1.  A header because the workgroup of k_chain_schur that
// produces the first diagonal tile of the compact dense system factors it on the spot (plba_chain.hip) instead of
// leaving that to a launch of its own.
#pragma once
#include "plba_internal.h"

namespace plba {

typedef double double4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double bcast_lane(double v, int l) {   // lane l (compile-time) -> SGPR pair
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double fast_rcp(double d) {            // v_rcp_f64 (2^-23) + two Newton steps -> full fp64
    double r = __builtin_amdgcn_rcp(d);
    r = fma(r, fma(-d, r, 1.0), r);
    r = fma(r, fma(-d, r, 1.0), r);
    return r;
}

constexpr int LS = 34;   // LDS row stride of a 32 x 32 tile: conflict-free operand reads (68 dwords = 4 mod 64), rows 16-byte aligned

__device__ __forceinline__ double bcast_half(double v, int hsel) {   // lanes of half `hsel` (compile time) -> both halves
    const int lo = __double2loint(v), hi = __double2hiint(v);
    // v_permlane32_swap a, b: a[32..63] <-> b[0..31]; with a == b the first result is the low half everywhere, the second the high half
    const auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(hsel ? r1[1] : r1[0], hsel ? r0[1] : r0[0]);
}

// Look-ahead factorisation of a 32 x 32 tile by one workgroup, as a pipeline of wavefronts over LDS.
// With A = Lu D Lu^T (Lu unit lower, D = diag(pivots)) the Cholesky factor is L = Lu D^1/2 and L^-1 = D^-1/2 Lu^-1:
//   wave 0  the pivot chain.  Lane = (row i = lane & 31, parity h = lane >> 5) holds a[q] = A[i][2q + h].  Column j
//           lives in register j/2 of the lanes of parity j%2; by symmetry it is also row j, so one ds_write publishes
//           what every lane needs for its rank-1 update.  The LDS copy is de-interleaved (slot(c) = (c&1)*16 + c/2) so
//           each parity reads its own columns as aligned pairs, and every column keeps its own 32 slots: after the
//           sweep cols[j][slot(r)] = Lu[r][j] * pivot_j.  Pivot, A[j+1][j] and A[j+2][j] come from v_readlane and
//           the multiplier crosses the halves with v_permlane32_swap: the dependent chain never waits on LDS.  The
//           multipliers themselves are column j of Lu: they are stored (lu) and flagged for the other waves.
//   wave 1  Lu11^-1 by column-oriented substitution, a step per flagged column, then Mu = Lu21 Lu11^-1 (matrix cores)
//   wave 3  Lu22^-1 the same way (in lock-step with columns 16..31), Wu = -Lu22^-1 Mu, then the row scaling D^-1/2;
//           a few hundred cycles after the last pivot L^-1 is on its way to global memory for the next launch
//   wave 2  1/sqrt(pivot) for every pivot as it appears (a lane per pivot), L into Lfac, pivot check (<= 0 or NaN
//           clears solver_ok).
// Cross-wave hand-off is by relaxed atomic LDS stores/loads only: a wave's LDS operations execute in order, so a flag
// written after the data is seen after the data, and the compiler keeps atomics in program order.  No fences, no
// s_waitcnt on the pivot chain.
__device__ __forceinline__ int slot32(int c) { return ((c & 1) << 4) | (c >> 1); }
__device__ __forceinline__ double fast_rsqrt(double d) {          // v_rsq_f64 + two Newton steps
    double y = __builtin_amdgcn_rsq(d);
    y = y * fma(-0.5 * d * y, y, 1.5);
    y = y * fma(-0.5 * d * y, y, 1.5);
    return y;
}
// relaxed workgroup-scope atomics: emitted in program order (ordered memory references for the scheduler) but, unlike
// volatile accesses, without an s_waitcnt after each one
template <typename T> __device__ __forceinline__ void vstore(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
template <typename T> __device__ __forceinline__ T vload(const T* p) { return __hip_atomic_load(const_cast<T*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
constexpr int SPIN_LIMIT = 1 << 18;     // never hang on a logic error: give up, the result is flagged as a failed solve

__device__ __forceinline__ double piv_empty() { return __longlong_as_double(-1ll); }    // a NaN no computation produces
__device__ __forceinline__ bool published(double v) { return __double_as_longlong(v) != -1ll; }
struct Look32 {            // LDS of the look-ahead pipeline
    double cols[32 * 32];  // cols[j][slot(r)] = A(j)[r][j]
    double lu[32 * 32];    // lu[j][slot(r)] = Lu[r][j]
    double sInv[32 * LS];  // Lu^-1 diagonal blocks (operand staging)
    double sM[256];        // Mu in MFMA B-operand order
    double pivd[32];       // pivot j; written last, so it doubles as "column j of lu is valid"
    double rs[32];         // 1 / sqrt(pivot)
    int rsflag[32];
    int mflag;
    int fail;
};

__device__ __forceinline__ void potrf32_stream(double* a, int lane, Look32& S) {
    const int i = lane & 31, h = lane >> 5;
    const int pos = slot32(i);
    if (h == 0) S.cols[pos] = a[0];
    // The pivots run ahead of the matrix on a uniform (all lanes redundant) recurrence,
    //   pivot_{j+1} = A[j+1][j+1] - A[j+1][j]^2 / pivot_j,
    // whose inputs are read (v_readlane) BEFORE step j's multipliers exist: the dependent chain per column is
    // rcp + 2 fma, and the multiplier broadcast, the rank-1 update and all LDS traffic hang off it with slack.
    double sj = bcast_lane(a[0], 0);
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const int qj = j >> 1, hj = j & 1;
        const double* buf = S.cols + j * 32;
        double* nbuf = S.cols + (j + 1) * 32;
        const double rj = fast_rcp(sj);
        const double pj = sj;
        const double s1 = (j + 1 < 32) ? bcast_lane(a[qj], j + 1 + 32 * hj) : 0.0;                       // A[j+1][j]
        if (j + 1 < 32) sj = fma(-(s1 * rj), s1, bcast_lane(a[(j + 1) >> 1], j + 1 + 32 * ((j + 1) & 1)));   // next pivot
        const double f = bcast_half(a[qj] * rj, hj);      // Lu[i][j], valid in both halves
        if (j + 1 < 32) {
            const int q1 = (j + 1) >> 1;
            if (hj == 0) {          // column j+1 is register qj of parity 1
                if (h == 1) { a[q1] = fma(-f, s1, a[q1]); nbuf[pos] = a[q1]; vstore(&S.lu[j * 32 + pos], f); }
            } else {                // register qj+1: column j+1 (parity 0) and column j+2 (parity 1)
                const double s2 = (j + 2 < 32) ? bcast_lane(a[qj], j + 2 + 32 * hj) : 0.0;
                a[q1] = fma(-f, h == 0 ? s1 : s2, a[q1]);
                if (h == 0) { nbuf[pos] = a[q1]; vstore(&S.lu[j * 32 + pos], f); }
            }
        }
        vstore(&S.pivd[j], pj);      // also the "column j is published" flag (reset value: PIV_EMPTY bits)
        {   // bulk: registers qs..15 of both parities from the LDS copy of column j
            const int qs = hj == 0 ? qj + 1 : qj + 2;
            const double2* b2 = reinterpret_cast<const double2*>(buf + h * 16);
#pragma unroll
            for (int q2 = qs / 2; q2 < 8; ++q2) {
                const double2 v = b2[q2];
                if (2 * q2 >= qs) a[2 * q2] = fma(-f, v.x, a[2 * q2]);
                a[2 * q2 + 1] = fma(-f, v.y, a[2 * q2 + 1]);
            }
        }
    }
}

// One 16 x 16 diagonal block (block B) of Lu^-1: a substitution step per column as wave 0 flags it.  The flag and
// the column are read in one batch (flag first: LDS returns in order), so a ready column costs a single LDS latency.
// x[t] = (Lu^-1)[16 B + t][16 B + (lane & 15)].
template <int B>
__device__ __forceinline__ void inv16_follow(Look32& S, double* x, int lane) {
    const int cc = lane & 15;
#pragma unroll
    for (int t = 0; t < 16; ++t) x[t] = (t == cc) ? 1.0 : 0.0;
#pragma unroll
    for (int j = 0; j < 15; ++j) {
        const int J = 16 * B + j;
        const double* u = S.lu + J * 32 + 8 * B;
        double uv[16];
        int spins = 0;
        while (true) {
            const bool fl = published(vload(&S.pivd[J]));
#pragma unroll
            for (int t = j + 1; t < 16; ++t) uv[t] = vload(&u[((t & 1) << 4) + (t >> 1)]);
            if (fl) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) { vstore(&S.fail, 1); break; }
        }
        const double m = x[j];
#pragma unroll
        for (int t = j + 1; t < 16; ++t) x[t] = fma(-m, uv[t], x[t]);
    }
}
// every lane waits for rs[base + (lane & 15)]; afterwards all 16 values of the block are readable
__device__ __forceinline__ void wait_rs16(Look32& S, int base, int lane) {
    int spins = 0;
    while (!vload(&S.rsflag[base + (lane & 15)])) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SPIN_LIMIT) { vstore(&S.fail, 1); break; }
    }
    asm volatile("" ::: "memory");     // plain loads of S.rs below this point stay below it
}

#ifdef PLBA_STAMPS
static __device__ unsigned long long g_lstamp[32];
#define LSTAMP(i) do { if (lane == 0 && kb == 5) g_lstamp[i] = __builtin_readcyclecounter(); } while (0)
#define CSTAMP(i) do { if (threadIdx.x == 0 && c == 5) g_lstamp[i] = __builtin_readcyclecounter(); } while (0)
#else
#define LSTAMP(i) do {} while (0)
#define CSTAMP(i) do {} while (0)
#endif
// tile in sC (row stride LS, complete and visible: call after a barrier that also saw S.pivd / rsflag / mflag / fail
// reset by look32_reset); all four waves enter.
// sc1 (agent-coherent, write-through) store / load of data other workgroups of the SAME launch consume
__device__ __forceinline__ void gstore_sc1(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double gload_sc1(const double* p) { return __hip_atomic_load(const_cast<double*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <bool FLOW> __device__ __forceinline__ void gput(double* p, double v) { if (FLOW) gstore_sc1(p, v); else *p = v; }

// Side job of wave 2 in the dataflow factorisation: fetch the two tiles of the NEXT chain step into LDS as soon as
// their helper workgroups have published them (polled between pivots).
struct NextTiles {
    const int* fa; const int* fd; int want;      // flags of pre(c+1,c) and diagpre(c+1)
    const double* A; const double* D; int ld;    // their location in sys
    double* sA; double* sD;                      // LDS destinations (row stride LS)
    bool active, done, failed, stamp;
};
__device__ __forceinline__ bool flags_ready(const NextTiles& n) {
    return __hip_atomic_load(const_cast<int*>(n.fa), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n.want &&
           __hip_atomic_load(const_cast<int*>(n.fd), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n.want;
}
__device__ __forceinline__ void fetch_next_tiles(NextTiles& n, int lane);

// FLOW: called from the single-launch dataflow factorisation: global results go out with sc1 stores, L^-1 is also
// left in LDS (sLinv, row stride LS) for the same workgroup's next step, and wave 2 prefetches `next`.
// KEEP: L^-1 is also left in LDS (sLinv, row stride LS) for the same workgroup's next stage.
template <bool FLOW, bool KEEP = FLOW>
__device__ __forceinline__ void lookahead_factor32(const DevBuf& d, int kb, const double* sC, Look32& S, int wv, int lane, double* sLinv = nullptr,
                                                   NextTiles* next = nullptr) {
    const int li = lane & 15, lk = lane >> 4;
    double* Ig = d.Linv32 + (size_t)kb * 1024;
    if (wv == 0) {
        const int i = lane & 31, h = lane >> 5;
        double a[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = sC[i * LS + 2 * q + h];
        LSTAMP(0);
        potrf32_stream(a, lane, S);
        LSTAMP(1);
    } else if (wv == 2) {
        LSTAMP(8);
        const int j = lane & 31;
        bool done = false, bad = false;
        int spins = 0;
        while (!__all(done)) {
            const double pv = vload(&S.pivd[j]);
            if (!done && published(pv)) {
                bad = !(pv > 0.0);
                vstore(&S.rs[j], fast_rsqrt(bad ? 1.0 : pv));
                vstore(&S.rsflag[j], 1);
                done = true;
            }
            if (FLOW && next->active && !next->done && flags_ready(*next)) fetch_next_tiles(*next, lane);
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) { bad = true; break; }
        }
        LSTAMP(10);
        asm volatile("" ::: "memory");
        {   // columns 16-31 of L (rows 16-31; wave 1 wrote the left half)
            double* Lg = d.Lfac + (size_t)(kb * 32) * d.ld + kb * 32;
            const int cl = 16 + li;
            const double rsc = S.rs[cl];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int rw = 16 + 4 * e + lk;
                gput<FLOW>(&Lg[(size_t)rw * d.ld + cl], (cl <= rw) ? S.cols[cl * 32 + slot32(rw)] * rsc : 0.0);
            }
        }
        if ((__any(bad) || S.fail) && lane == 0) d.ctrl->solver_ok = 0;
        if (FLOW && next->active && !next->done) {      // the helpers were slower than the pivot sweep: wait for them now
            int spins2 = 0;
            while (!flags_ready(*next)) { __builtin_amdgcn_s_sleep(2); if (++spins2 > (1 << 18)) { next->failed = true; break; } }
            fetch_next_tiles(*next, lane);
        }
        LSTAMP(11);
    } else if (wv == 1) {
        LSTAMP(4);
        double x[16];
        inv16_follow<0>(S, x, lane);
#pragma unroll
        for (int t = 0; t < 16; ++t) S.sInv[t * LS + li] = x[t];
        LSTAMP(5);
        // Mu = Lu21 Lu11^-1 lands in the C/D layout (col = lane & 15, row = (lane >> 4) + 4 v), which is exactly a
        // B operand of wave 3's product when its k index is enumerated as (lane >> 4) + 4 v
        double4v m = (double4v){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int t = 4 * s + lk;
            m = __builtin_amdgcn_mfma_f64_16x16x4f64(vload(&S.lu[t * 32 + ((li & 1) << 4) + 8 + (li >> 1)]), S.sInv[t * LS + li], m, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) S.sM[v * 64 + lane] = m[v];
        asm volatile("" ::: "memory");
        vstore(&S.mflag, 1);
        LSTAMP(6);
        wait_rs16(S, 0, lane);
#pragma unroll
        for (int e = 0; e < 8; ++e) {      // rows 0-15 of L^-1: [D1^-1/2 Lu11^-1 | 0]
            const int idx = e * 64 + lane, rw = idx >> 5, cl = idx & 31;
            const double v = (cl < 16) ? S.sInv[rw * LS + cl] * S.rs[rw] : 0.0;
            gput<FLOW>(&Ig[idx], v);
            if (KEEP) sLinv[rw * LS + cl] = v;
        }
        {   // columns 0-15 of L (all rows) and the zero block above the diagonal
            double* Lg = d.Lfac + (size_t)(kb * 32) * d.ld + kb * 32;
            const double rsc = S.rs[li];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int rw = 4 * e + lk;
                gput<FLOW>(&Lg[(size_t)rw * d.ld + li], (li <= rw) ? S.cols[li * 32 + slot32(rw)] * rsc : 0.0);
                if (rw < 16) gput<FLOW>(&Lg[(size_t)rw * d.ld + 16 + li], 0.0);
            }
        }
        LSTAMP(7);
    } else {
        double x[16];
        inv16_follow<1>(S, x, lane);
        LSTAMP(12);
#pragma unroll
        for (int t = 0; t < 16; ++t) S.sInv[(16 + t) * LS + 16 + li] = x[t];
        int spins = 0;
        while (!vload(&S.mflag)) { __builtin_amdgcn_s_sleep(1); if (++spins > SPIN_LIMIT) { vstore(&S.fail, 1); break; } }
        asm volatile("" ::: "memory");
        double4v w = (double4v){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int v = 0; v < 4; ++v) w = __builtin_amdgcn_mfma_f64_16x16x4f64(S.sInv[(16 + li) * LS + 16 + lk + 4 * v], S.sM[v * 64 + lane], w, 0, 0, 0);
        LSTAMP(13);
        wait_rs16(S, 16, lane);
        // rows 16-31 of L^-1 = D2^-1/2 [ -Lu22^-1 Mu | Lu22^-1 ], straight from the registers
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const double val = -w[v] * S.rs[16 + lk + 4 * v];
            gput<FLOW>(&Ig[(16 + lk + 4 * v) * 32 + li], val);
            if (KEEP) sLinv[(16 + lk + 4 * v) * LS + li] = val;
        }
        if (lane < 16) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const double val = x[t] * S.rs[16 + t];
                gput<FLOW>(&Ig[(16 + t) * 32 + 16 + li], val);
                if (KEEP) sLinv[(16 + t) * LS + 16 + li] = val;
            }
        }
        LSTAMP(14);
    }
}
__device__ __forceinline__ void fetch_next_tiles(NextTiles& n, int lane) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int idx = e * 64 + lane, rw = idx >> 5, cl = idx & 31;
        n.sA[rw * LS + cl] = gload_sc1(&n.A[(size_t)rw * n.ld + cl]);
        n.sD[rw * LS + cl] = gload_sc1(&n.D[(size_t)rw * n.ld + cl]);
    }
    n.done = true;
#ifdef PLBA_STAMPS
    if (lane == 0 && n.want != 0 && n.stamp) g_lstamp[21] = __builtin_readcyclecounter();
#endif
}
__device__ __forceinline__ void look32_reset(Look32& S, int tid) {
    if (tid < 32) { S.pivd[tid] = piv_empty(); S.rsflag[tid] = 0; }
    if (tid == 32) { S.mflag = 0; S.fail = 0; }
}

// The 64 x 64 diagonal tile [D11 .; D21 D22] (three 32 x 32 tiles in LDS, row stride LS) factored by one workgroup as two
// pipelined 32-column sweeps, everything in between staying in LDS:
//     L11, I11 = L11^-1  (sweep 1)    L21 = D21 I11^T    D22 -= L21 L21^T    L22, I22  (sweep 2)    I21 = -I22 L21 I11
// L -> d.Lfac tiles (kb,kb), (kb+1,kb), (kb+1,kb+1);  I11, I22 -> d.Linv32[kb], [kb+1];  I21 -> its place in the 64 x 64
// inverse d.Linv[kb / 2].  sI11 / sI22 / sL21: 32 x LS scratch each; sD21 is overwritten (L21 I11).  All 256 threads
// enter; the tiles must be complete and visible (barrier) on entry.  This is what lets a block step of the factorisation
// cover 64 columns: the per-launch costs (panel products, trailing update, write-out, kernel boundary) are paid once
// per two pivot sweeps.
__device__ __forceinline__ void factor64_lds(const DevBuf& d, const int kb, double* sD11, double* sD21, double* sD22, Look32& S,
                                             double* sI11, double* sI22, double* sL21, const int wv, const int lane) {
    const int li = lane & 15, lk = lane >> 4, tr = wv >> 1, tc = wv & 1;
    // one copy of the sweep's code for both halves (it is long and fully unrolled: two inlined copies would not share
    // the instruction cache)
#pragma nounroll
    for (int half = 0; half < 2; ++half) {
        if (half == 1) {
            {   // L21 = D21 I11^T
                double4v m = (double4v){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) m = __builtin_amdgcn_mfma_f64_16x16x4f64(sD21[(tr * 16 + li) * LS + kk * 4 + lk], sI11[(tc * 16 + li) * LS + kk * 4 + lk], m, 0, 0, 0);
                double* Lg = d.Lfac + (size_t)((kb + 1) * 32) * d.ld + kb * 32;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int row = tr * 16 + lk + 4 * v, col = tc * 16 + li;
                    sL21[row * LS + col] = m[v];
                    Lg[(size_t)row * d.ld + col] = m[v];
                }
            }
            __syncthreads();
            {   // D22 -= L21 L21^T
                double4v m = (double4v){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) m = __builtin_amdgcn_mfma_f64_16x16x4f64(sL21[(tr * 16 + li) * LS + kk * 4 + lk], sL21[(tc * 16 + li) * LS + kk * 4 + lk], m, 0, 0, 0);
#pragma unroll
                for (int v = 0; v < 4; ++v) sD22[(tr * 16 + lk + 4 * v) * LS + tc * 16 + li] -= m[v];
            }
        }
        look32_reset(S, threadIdx.x);
        __syncthreads();
        lookahead_factor32<false, true>(d, kb + half, half ? sD22 : sD11, S, wv, lane, half ? sI22 : sI11);
        __syncthreads();
    }
    double* sT = sD21;
    {   // T = L21 I11
        double4v m = (double4v){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) m = __builtin_amdgcn_mfma_f64_16x16x4f64(sL21[(tr * 16 + li) * LS + kk * 4 + lk], sI11[(kk * 4 + lk) * LS + tc * 16 + li], m, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 4; ++v) sT[(tr * 16 + lk + 4 * v) * LS + tc * 16 + li] = m[v];
    }
    __syncthreads();
    {   // I21 = -I22 T
        double4v m = (double4v){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) m = __builtin_amdgcn_mfma_f64_16x16x4f64(sI22[(tr * 16 + li) * LS + kk * 4 + lk], sT[(kk * 4 + lk) * LS + tc * 16 + li], m, 0, 0, 0);
        double* Ig = d.Linv + (size_t)(kb >> 1) * TILE * TILE + 32 * TILE;
#pragma unroll
        for (int v = 0; v < 4; ++v) Ig[(tr * 16 + lk + 4 * v) * TILE + tc * 16 + li] = -m[v];
    }
}

}  // namespace plba
