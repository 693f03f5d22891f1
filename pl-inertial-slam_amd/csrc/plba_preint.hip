// plba_preint.hip — IMU preintegration producer.
//
// Replaces KeyFrame::ComputeIMUPreIntSinceLastFrame (src/keyFrame.cpp:139-172) + IMUPreintegrator::reset / update
// (IMU/IMUPreintegrator.cpp:47-139) for all keyframe intervals of a window at once: the reference re-runs it for
// every window keyframe whenever the bias estimate changes (src/mapHandler.cpp:4850,4951).
//
//   host   : sample selection and the dt of every step, in long double like the reference's time stamps
//            (pure index work, O(samples))
//   device : one lane per interval runs the update recurrence (plba_math.h::preint_update) over its schedule;
//            the 142-double payload lives in registers / scratch, samples are streamed from a shared table.
// Intervals are independent and there are only K-1 of them, so this is a latency-bound single-wave kernel; it is
// here so the measurement the path consumes is produced on the device it is consumed on.
#include <vector>

#include "plba_problem.h"

namespace plba {

__global__ __launch_bounds__(64) void k_preintegrate(int M, const int* __restrict__ sched_start, const int* __restrict__ sched_idx,
                                                     const double* __restrict__ sched_dt, const double* __restrict__ gyr,
                                                     const double* __restrict__ acc, const double* __restrict__ bg,
                                                     const double* __restrict__ ba, double gcov, double acov, double* __restrict__ out) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    double pre[PREINT_DOUBLES];
    preint_reset(pre);
    const V3 g0 = ld_v3(bg + 3 * m), a0 = ld_v3(ba + 3 * m);
    for (int s = sched_start[m]; s < sched_start[m + 1]; ++s) {
        const int i = sched_idx[s];
        preint_update(pre, ld_v3(gyr + 3 * (size_t)i) - g0, ld_v3(acc + 3 * (size_t)i) - a0, sched_dt[s], gcov, acov);
    }
    for (int q = 0; q < PREINT_DOUBLES; ++q) out[(size_t)m * PREINT_DOUBLES + q] = pre[q];
}

}  // namespace plba

using namespace plba;

extern "C" int plba_preintegrate(plba_problem* p, int M, const int32_t* sample_start, const long double* t, const double* gyr3,
                                 const double* acc3, const long double* t_prev, const long double* t_curr, const double* bg3,
                                 const double* ba3, double gyr_meas_cov, double acc_meas_cov, double* out142) {
    if (!p) return PLBA_ERR_INVALID;
    if (M < 0 || (M > 0 && (!sample_start || !t || !gyr3 || !acc3 || !t_prev || !t_curr || !bg3 || !ba3 || !out142)))
        PLBA_FAIL(p, PLBA_ERR_INVALID, "preintegrate: null argument");
    if (M == 0) return PLBA_OK;
    for (int m = 0; m < M; ++m)
        if (sample_start[m + 1] < sample_start[m] || sample_start[0] != 0) PLBA_FAIL(p, PLBA_ERR_INVALID, "preintegrate: sample_start must ascend from 0");
    const int S = sample_start[M];
    // the step schedule of src/keyFrame.cpp:147-170, literally
    std::vector<int> sstart(M + 1, 0), sidx;
    std::vector<double> sdt;
    for (int m = 0; m < M; ++m) {
        const int end = sample_start[m + 1];
        int i = sample_start[m];
        while (i < end && t[i] < t_prev[m]) ++i;                    // :147-149 (the reference has no bound check)
        if (i < end) {
            sidx.push_back(i); sdt.push_back((double)(t[i] - t_prev[m])); ++i;                                        // :150-154
            while (i < end && t[i] <= t_curr[m]) { sidx.push_back(i); sdt.push_back((double)(t[i] - t[i - 1])); ++i; }   // :155-161
            if (i < end) { sidx.push_back(i); sdt.push_back((double)(t_curr[m] - t[i])); }                            // :162-167
        }
        sstart[m + 1] = (int)sidx.size();
    }
    PLBA_HIPCK(p, hipSetDevice(p->device));
    hipStream_t s = p->stream;
    DArrStreamScope staged(s, p->have_ctx ? p->ctx.stage : nullptr);      // uploads queued on the stream through the pinned staging area
    DArr<int> d_start, d_idx;
    DArr<double> d_dt, d_g, d_a, d_bg, d_ba, d_out;
    PLBA_HIPCK(p, d_start.upload(sstart));
    if (sidx.empty()) { sidx.push_back(0); sdt.push_back(0.0); }
    PLBA_HIPCK(p, d_idx.upload(sidx)); PLBA_HIPCK(p, d_dt.upload(sdt));
    const std::vector<double> hg(gyr3, gyr3 + 3 * (size_t)(S > 0 ? S : 1)), ha(acc3, acc3 + 3 * (size_t)(S > 0 ? S : 1)), hbg(bg3, bg3 + 3 * (size_t)M), hba(ba3, ba3 + 3 * (size_t)M);
    PLBA_HIPCK(p, d_g.upload(hg)); PLBA_HIPCK(p, d_a.upload(ha)); PLBA_HIPCK(p, d_bg.upload(hbg)); PLBA_HIPCK(p, d_ba.upload(hba));      // alive until the final wait
    PLBA_HIPCK(p, d_out.alloc((size_t)M * PREINT_DOUBLES));
    hipLaunchKernelGGL(k_preintegrate, dim3((M + 63) / 64), dim3(64), 0, s, M, d_start.p, d_idx.p, d_dt.p, d_g.p, d_a.p, d_bg.p, d_ba.p,
                       gyr_meas_cov, acc_meas_cov, d_out.p);
    PLBA_HIPCK(p, hipGetLastError());
    PLBA_HIPCK(p, plba_d2h(p, out142, d_out.p, (size_t)M * PREINT_DOUBLES * 8));
    PLBA_HIPCK(p, plba_stream_wait(s));
    return PLBA_OK;
}
