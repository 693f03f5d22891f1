"""IMU preintegration producer: KeyFrame::ComputeIMUPreIntSinceLastFrame + IMUPreintegrator::update
(src/keyFrame.cpp:139-172, IMU/IMUPreintegrator.cpp:47-139) — oracle vs an independent schedule, HIP vs oracle."""
import numpy as np
import pytest

LD = np.longdouble


def _stream(pkg, M, rng, t0=LD("1403636579.763555527"), rate=200.0, kf_dt=0.25, jitter=True):
    """EuRoC-shaped stream: ns-resolution stamps around 1.4e9 s (a double cannot hold them: long double matters),
    every interval owning the samples from a little before its first image to a little after its second."""
    t_prev = t0 + LD(kf_dt) * np.arange(M, dtype=LD)
    t_curr = t_prev + LD(kf_dt)
    ts, starts = [], [0]
    for m in range(M):
        n_before = int(rng.integers(0, 3))            # samples older than prev_t (skipped by the reference loop)
        n_after = int(rng.integers(0, 3))             # samples past curr_t (the first one takes the last partial step)
        n_in = int(kf_dt * rate)
        k = np.arange(-n_before, n_in + n_after, dtype=LD)
        off = LD(0.0007) if jitter else LD(0.0)
        tt = t_prev[m] + off + k / LD(rate) + (LD(1e-6) * rng.normal(size=len(k)).astype(LD) if jitter else 0)
        ts.append(np.sort(tt))
        starts.append(starts[-1] + len(tt))
    t = np.concatenate(ts)
    S = len(t)
    gyr = rng.normal(size=(S, 3)) * 0.3
    acc = rng.normal(size=(S, 3)) * 2.0 + np.array([0, 0, 9.81])
    bg = rng.normal(size=(M, 3)) * 1e-3
    ba = rng.normal(size=(M, 3)) * 1e-2
    return dict(sample_start=np.array(starts, dtype=np.int32), t=t, gyr=gyr, acc=acc, t_prev=t_prev, t_curr=t_curr, bg=bg, ba=ba)


def _schedule(s, m):
    """src/keyFrame.cpp:147-170 restated on its own (index, dt) — independent of the C implementations."""
    lo, hi = int(s["sample_start"][m]), int(s["sample_start"][m + 1])
    t, prev, curr = s["t"], s["t_prev"][m], s["t_curr"][m]
    i = lo
    while i < hi and t[i] < prev:
        i += 1
    out = []
    if i >= hi:
        return out
    out.append((i, float(t[i] - prev))); i += 1
    while i < hi and t[i] <= curr:
        out.append((i, float(t[i] - t[i - 1]))); i += 1
    if i < hi:
        out.append((i, float(curr - t[i])))
    return out


def _call(prob, pkg, s):
    return prob.preintegrate(s["sample_start"], s["t"], s["gyr"], s["acc"], s["t_prev"], s["t_curr"], s["bg"], s["ba"],
                             pkg.window.GYR_MEAS_COV, pkg.window.ACC_MEAS_COV)


def test_oracle_driver_follows_the_reference_schedule(pkg, orc):
    rng = np.random.default_rng(11)
    s = _stream(pkg, 6, rng)
    p = orc.new_problem()
    got = _call(p, pkg, s)
    gc, ac = pkg.window.GYR_MEAS_COV, pkg.window.ACC_MEAS_COV
    n_neg = 0
    for m in range(6):
        pre = np.zeros(142); pre[[6, 10, 14]] = 1.0
        for i, dt in _schedule(s, m):
            n_neg += dt < 0
            pre = orc.preint_update(pre, s["gyr"][i] - s["bg"][m], s["acc"][i] - s["ba"][m], dt, gc, ac)
        assert np.array_equal(got[m], pre)
    assert n_neg > 0          # the literal `dt = curr_t - t[i]` step of keyFrame.cpp:162-167 was exercised
    p.close()


def test_oracle_driver_against_the_window_generator(pkg, orc):
    """uniform 200 Hz samples ending exactly on the second image: the numpy generator's recurrence is the same thing"""
    rng = np.random.default_rng(12)
    M, S, dt = 3, 50, 0.005
    omega = rng.normal(size=(M, S, 3)) * 0.3
    acc = rng.normal(size=(M, S, 3)) * 2.0 + np.array([0, 0, 9.81])
    ref = pkg.window.preintegrate(omega, acc, dt)
    t_prev = LD(10.0) + LD(0.25) * np.arange(M, dtype=LD)
    t = np.concatenate([t_prev[m] + LD(dt) * np.arange(1, S + 1, dtype=LD) for m in range(M)])
    s = dict(sample_start=np.arange(M + 1, dtype=np.int32) * S, t=t, gyr=omega.reshape(-1, 3), acc=acc.reshape(-1, 3),
             t_prev=t_prev, t_curr=t.reshape(M, S)[:, -1].copy(), bg=np.zeros((M, 3)), ba=np.zeros((M, 3)))
    p = orc.new_problem()
    got = _call(p, pkg, s)
    p.close()
    assert np.allclose(got, ref, rtol=1e-9, atol=1e-16)


def test_empty_and_degenerate_intervals(pkg, orc):
    p = orc.new_problem()
    s = dict(sample_start=np.array([0, 0, 2], dtype=np.int32), t=np.array([1.0, 1.005], dtype=LD), gyr=np.zeros((2, 3)), acc=np.zeros((2, 3)),
             t_prev=np.array([0.0, 2.0], dtype=LD), t_curr=np.array([0.5, 2.5], dtype=LD), bg=np.zeros((2, 3)), ba=np.zeros((2, 3)))
    got = _call(p, pkg, s)     # interval 0 owns no samples, interval 1 only samples older than its first image
    p.close()
    ident = np.zeros(142); ident[[6, 10, 14]] = 1.0
    assert np.array_equal(got[0], ident) and np.array_equal(got[1], ident)


@pytest.mark.gpu
def test_hip_preintegration_matches_the_oracle(pkg, orc, hip):
    rng = np.random.default_rng(13)
    for M in (1, 49, 130):
        s = _stream(pkg, M, rng)
        g, o = pkg.new_problem(), orc.new_problem()
        a, b = _call(g, pkg, s), _call(o, pkg, s)
        g.close(); o.close()
        # fp64 with fused multiply-adds against the oracle's unfused arithmetic: relative to each block's own scale
        for lo, hi in ((0, 3), (3, 6), (6, 15), (15, 24), (24, 33), (33, 42), (42, 51), (51, 60), (60, 141), (141, 142)):
            sc = np.abs(b[:, lo:hi]).max()
            assert np.abs(a[:, lo:hi] - b[:, lo:hi]).max() <= 1e-11 * sc, (M, lo)


@pytest.mark.gpu
def test_hip_preintegration_degenerate(pkg, orc, hip):
    g = pkg.new_problem()
    s = dict(sample_start=np.array([0, 0, 2], dtype=np.int32), t=np.array([1.0, 1.005], dtype=LD), gyr=np.zeros((2, 3)), acc=np.zeros((2, 3)),
             t_prev=np.array([0.0, 2.0], dtype=LD), t_curr=np.array([0.5, 2.5], dtype=LD), bg=np.zeros((2, 3)), ba=np.zeros((2, 3)))
    got = _call(g, pkg, s)
    ident = np.zeros(142); ident[[6, 10, 14]] = 1.0
    assert np.array_equal(got[0], ident) and np.array_equal(got[1], ident)
    assert _call(g, pkg, dict(s, sample_start=np.array([0], dtype=np.int32))).shape == (0, 142)
    g.close()
