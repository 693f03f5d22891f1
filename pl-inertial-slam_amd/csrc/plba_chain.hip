// plba_chain.hip — elimination of the velocity / bias ("chain") variables ahead of the dense factorisation.
//
// Of the 15 dimensions a keyframe contributes to the reduced camera system only 6 (position, rotation) are coupled to
// landmarks.  The other 9 (velocity, gyro / accel bias deltas) are touched by the IMU edges alone: in keyframe order
// their 9 x 9 blocks form a block-TRIDIAGONAL matrix, coupled to the pose dimensions of the neighbouring keyframes only.
// One level of nested dissection makes that structure parallel: every (SEG+1)-th chain block is kept as a SEPARATOR
// and joins the dense variables; the runs of SEG blocks in between ("segments") no longer see each other and are
// eliminated by one workgroup each, all at once:
//
//     [ C   B ] [x_c]   [b_c]     C = blockdiag over segments of block-tridiagonal C_g = L_g L_g^T
//     [ B^T A ] [x_d] = [b_d]     W = L^-1 [B | b_c];   (A - W_B^T W_B) x_d = b_d - W_B^T w_b;   x_c = L^-T (w_b - W_B x_d)
//
// with x_d = pose dimensions + separator chain dimensions (348 instead of 735 at the headline size: 12 block steps of
// the dense factorisation instead of 24) and at most SEG sequential 9 x 9 steps per workgroup instead of one per keyframe.
// B only has columns for the keyframes next to a segment, so each workgroup carries ~150 columns, and W^T W only
// touches the diagonal band of the dense system.
//
//   k_chain_elim   one workgroup per segment.  wave 0: per chain block  C_ii -= L_i,i-1 L_i,i-1^T, 9 x 9 Cholesky and
//                  inverse in registers (v_readlane broadcasts), L_i+1,i = C_i+1,i L_ii^-T, published through LDS;
//                  waves 1-3: a lane per coupled column:  w_i = L_ii^-1 (B_i - L_i,i-1 w_i-1).  Everything a step reads
//                  was staged in LDS up front.
//   k_chain_schur  tile (a,b) of  A - W^T W  (v_mfma_f64_16x16x4_f64) over the rows of the segments whose column window
//                  meets the tile; right-hand side row; identity padding
//   chain_back_segment (plba_kernels.hip, riding in front of the landmark back-substitution launch) one workgroup per
//                  segment:  v = w_b - W_B x_d  over its column window, backward block substitution, scatter of x into
//                  system order, and the state update of the segment's keyframes
//
// Applicable when no marginalization prior is attached (it couples chain variables of several keyframes) and every IMU
// edge joins neighbouring chain blocks; otherwise the solver falls back to the dense path on the full system.
// Everything is read from the LOWER triangle of `sys` (in a sharded run only that part is globally summed).
#include "plba_internal.h"
#include "plba_chain_dev.h"
#include "plba_factor32_dev.h"

namespace plba {

__global__ __launch_bounds__(ELIM_THREADS) void k_chain_elim(DevBuf d, ChainView cv) {
    __shared__ __attribute__((aligned(16))) ChainElimLds LDS;
    chain_elim_segment(d, cv, blockIdx.x, LDS);
}

// dd.sys tile (ta, tb), ta >= tb, of the dense system  A - W_B^T W_B  (32 x 32, matrix cores); the tiles of block row
// ta == Pdpad / 32 carry the right-hand side  b_d - W_B^T w_b  in their first row.  Only the rows of W that belong to
// segments whose column window meets both tiles are read (everything else in those columns is zero).
// PRE (round 4, fused landmark path on one GPU): the W^T W products were formed by extra workgroups of the k_lm_gather launch (they need W
// alone — the chain elimination's output — not the assembled system) and left in dd.wtw, workgroup b's 1024 values at b * 1024 + 4 t; this
// launch then only subtracts them and factors the chains' first tiles: one memory round and the matrix-core loop less on the path of the
// on-the-spot factorisation.
template <bool PRE>
__global__ __launch_bounds__(256) void k_chain_schur(DevBuf d, ChainView cv, DevBuf dd) {
    __shared__ __attribute__((aligned(16))) double sC0[32 * LS];      // workgroup 0: the first diagonal tile, factored on the spot
    __shared__ __attribute__((aligned(16))) Look32 S0;
    const int T = cv.Pdpad / 32;
    const int b = blockIdx.x;
    int ta, tb;
    chain_schur_tile_of(cv, dd, b, ta, tb);
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const int tr = wv >> 1, tc = wv & 1;
    const int ldd = dd.ld;
    const bool rhs_row = (ta == T);
    double4v acc;
    if (PRE) acc = *reinterpret_cast<const double4v*>(dd.wtw + (size_t)b * 1024 + 4 * threadIdx.x);
    double aold[4];
    {
        const int cb = tb * 32 + tc * 16 + li;
        const int pb = cb < cv.Pd ? cv.pidx[cb] : 0;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int lr = tr * 16 + lk + 4 * v;
            if (rhs_row) aold[v] = (lr == 0 && cb < cv.Pd) ? d.sys[(size_t)d.Ppad * d.ld + pb] : 0.0;
            else {
                const int ca = ta * 32 + lr;
                aold[v] = (ca < cv.Pd && cb < cv.Pd) ? sym_at(d.sys, d.ld, cv.pidx[ca], pb) : ((ca == cb) ? 1.0 : 0.0);
            }
        }
    }
    if (!PRE) acc = chain_wtw_tile(cv, ta, tb);
    // C/D layout: col = lane & 15, row = (lane >> 4) + 4 v
    // Multi-chain factorisation (dd.twin_m0 > 0): the system is written PERMUTED through dd.perm — [chains | separators] — every
    // chain's first tile is factored on the spot like tile (0,0) (turned around for the chain that is eliminated bottom-up), and
    // the separator region of dd.alt (the accumulator of every second chain) is cleared.
    const bool twin = dd.twin_m0 > 0;
    const int fsel = (!rhs_row && ta == tb) ? (twin ? dd.twin_fac[ta] : (b == 0 ? 0 : -1)) : -1;      // >= 0: this workgroup factors its tile
    const bool frev = fsel >= 0 && (fsel >> 16) != 0;
    const int m0c = dd.twin_m0 * 32;      // first permuted index of the separator region
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int lr = tr * 16 + lk + 4 * v, lc = tc * 16 + li;
        const int cb = tb * 32 + lc;
        const double out = (rhs_row && lr != 0) ? 0.0 : ((rhs_row || (ta * 32 + lr < cv.Pd && cb < cv.Pd)) ? aold[v] - acc[v] : aold[v]);
        const int pcb = twin ? dd.perm[cb] : cb;
        if (rhs_row) {
            dd.sys[(size_t)(cv.Pdpad + lr) * ldd + pcb] = (cb < cv.Pd) ? out : 0.0;
            if (twin && pcb >= m0c) { dd.alt[(size_t)(cv.Pdpad + lr) * ldd + pcb] = 0.0; dd.alt2[(size_t)(cv.Pdpad + lr) * ldd + pcb] = 0.0; }
        } else {
            const int ca = ta * 32 + lr;
            const int pca = twin ? dd.perm[ca] : ca;
            dd.sys[(size_t)pca * ldd + pcb] = out;
            if (ta != tb) dd.sys[(size_t)pcb * ldd + pca] = out;     // keep the matrix symmetric (debug readers; the permuted lower triangle may be the natural upper one)
            if (twin && pca >= m0c && pcb >= m0c) {
                dd.alt[(size_t)pca * ldd + pcb] = 0.0; dd.alt2[(size_t)pca * ldd + pcb] = 0.0;
                if (ta != tb) { dd.alt[(size_t)pcb * ldd + pca] = 0.0; dd.alt2[(size_t)pcb * ldd + pca] = 0.0; }
            }
            if (fsel >= 0) sC0[(frev ? 31 - lr : lr) * LS + (frev ? 31 - lc : lc)] = out;
        }
    }
    if (fsel < 0 || dd.flow || dd.wide || dd.band) return;
    // tile (0,0) is complete in LDS: run the look-ahead pipeline of the factorisation on it right here (L(0,0) -> dd.Lfac,
    // L(0,0)^-1 -> dd.Linv32[0]) instead of in a launch of its own (k_potrf0_32): the first block step follows directly
    factor32_reset(S0, threadIdx.x);
    __syncthreads();
    factor32_tile<false>(dd, fsel & 0xffff, sC0, S0, wv, lane);
}

void launch_chain_elim(const DevBuf& d, const ChainView& cv, hipStream_t s) {
    hipLaunchKernelGGL(k_chain_elim, dim3(cv.nseg), dim3(ELIM_THREADS), 0, s, d, cv);
}
bool chain_schur_factors_tile0(const DevBuf& dd) { return !dd.flow && !dd.wide && !dd.band; }
void launch_chain_schur(const DevBuf& d, const ChainView& cv, const DevBuf& dd, hipStream_t s, bool products_done) {
    const int T = cv.Pdpad / 32;
    if (products_done) hipLaunchKernelGGL(k_chain_schur<true>, dim3(T * (T + 1) / 2 + T), dim3(256), 0, s, d, cv, dd);
    else hipLaunchKernelGGL(k_chain_schur<false>, dim3(T * (T + 1) / 2 + T), dim3(256), 0, s, d, cv, dd);
}

}  // namespace plba
