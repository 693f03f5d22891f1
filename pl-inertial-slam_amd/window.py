"""Synthetic EuRoC-shaped sliding windows for the local-BA hot path.

Produces exactly the arrays the reference call site hands to g2o
(src/mapHandler.cpp:5741-6034): NavStates of the window keyframes, map points / map lines with
their per-keyframe observation lists (landmark-major, include/mapFeatures.h:40-100), the
IMUPreintegrator payload of every consecutive keyframe pair (IMU/IMUPreintegrator.h:187-201,
recurrences of IMU/IMUPreintegrator.cpp:80-139) and the information matrices built at
mapHandler.cpp:5249-5251,5269-5270,5287.

Pure numpy, deterministic: a counter-based splitmix64 generator + Box-Muller (not
numpy.random, whose streams are version-defined), seed = 0x5EED0000 + config index, so every
machine regenerates bit-identical windows.  This module is data generation for tests and the
bench; it contains no solver code and does not touch the oracle.
"""
import numpy as np

# camera: config/dataset_params/euroc_params.yaml:2,9,11 ; T_BS: dataset_params.yaml:16-22
FX, FY, CX, CY = 458.654, 457.296, 367.215, 248.375
IMG_W, IMG_H = 752, 480
T_BS = np.array([[0.0148655429818, -0.999880929698, 0.00414029679422, -0.0216401454975],
                 [0.999557249008, 0.0149672133247, 0.025715529948, -0.064676986768],
                 [-0.0257744366974, 0.00375618835797, 0.999660727178, 0.00981073058949],
                 [0.0, 0.0, 0.0, 1.0]])
GW = np.array([0.0, 0.0, -9.81])
# IMU/imudata.cpp:24-32
GYR_BIAS_RW2 = 2.0e-5 * 2.0e-5
ACC_BIAS_RW2 = 5.0e-3 * 5.0e-3
GYR_MEAS_COV = 1.7e-4 * 1.7e-4 / 0.005
ACC_MEAS_COV = 2.0e-3 * 2.0e-3 / 0.005 * 100
IMU_DT = 0.005
KF_DT = 0.25
BG_TRUE = np.array([2.0, -1.0, 3.0]) * 1e-3
BA_TRUE = np.array([2.0, 5.0, -3.0]) * 1e-2
# Huber deltas are `const float` at the call site (mapHandler.cpp:5247-5248,5307,5366)
HUBER = {0: float(np.float32(np.sqrt(5.991))), 1: float(np.float32(np.sqrt(5.991))),
         2: float(np.float32(np.sqrt(21.666))), 3: float(np.float32(np.sqrt(16.812)))}
CHI2_GATE = 5.991

CONFIGS = {  # BASELINE.json configs (1-based index)
    1: dict(K=10, Np=2000, Nl=500, imu=False),
    2: dict(K=30, Np=10000, Nl=2000, imu=False),
    3: dict(K=50, Np=20000, Nl=4000, imu=True),
    4: dict(K=50, Np=20000, Nl=4000, imu=True),   # + marginalization prior (built by a preceding window)
    5: dict(K=200, Np=200000, Nl=40000, imu=True),
}


class Rng:
    """Counter-based generator: value i of stream s = splitmix64(seed, s, i)."""
    _M = np.uint64(0xFFFFFFFFFFFFFFFF)

    def __init__(self, seed):
        self.seed = np.uint64(seed)
        self.ctr = 0

    @staticmethod
    def _mix(x):
        with np.errstate(over="ignore"):
            x = x + np.uint64(0x9E3779B97F4A7C15)
            z = x
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            return z ^ (z >> np.uint64(31))

    def bits(self, n):
        idx = np.arange(self.ctr, self.ctr + n, dtype=np.uint64)
        self.ctr += n
        with np.errstate(over="ignore"):
            return self._mix(self._mix(idx) ^ (self.seed * np.uint64(0xD1342543DE82EF95)))

    def uniform(self, n, lo=0.0, hi=1.0):
        u = (self.bits(n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        return lo + (hi - lo) * u

    def integers(self, n, lo, hi):
        """uniform integers in [lo, hi]"""
        return lo + np.floor(self.uniform(n) * (hi - lo + 1)).astype(np.int64)

    def normal(self, shape, sigma=1.0):
        n = int(np.prod(shape))
        m = (n + 1) // 2
        u1 = 1.0 - self.uniform(m)          # (0, 1]
        u2 = self.uniform(m)
        r = np.sqrt(-2.0 * np.log(u1))
        z = np.concatenate([r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)])[:n]
        return sigma * z.reshape(shape)


# ---------------------------------------------------------------------------------------------
# batched SO(3) helpers (numpy; independent of the oracle's C code)
# ---------------------------------------------------------------------------------------------
def hat(v):
    v = np.asarray(v, dtype=np.float64)
    O = np.zeros(v.shape[:-1] + (3, 3))
    O[..., 0, 1] = -v[..., 2]; O[..., 0, 2] = v[..., 1]
    O[..., 1, 0] = v[..., 2]; O[..., 1, 2] = -v[..., 0]
    O[..., 2, 0] = -v[..., 1]; O[..., 2, 1] = v[..., 0]
    return O


def exp_so3(w):
    """Rodrigues, batched: w (...,3) -> R (...,3,3)."""
    w = np.asarray(w, dtype=np.float64)
    th = np.linalg.norm(w, axis=-1)[..., None, None]
    W = hat(w)
    small = th < 1e-10
    ths = np.where(small, 1.0, th)
    a = np.where(small, 1.0, np.sin(ths) / ths)
    b = np.where(small, 0.5, (1.0 - np.cos(ths)) / (ths * ths))
    return np.eye(3) + a * W + b * (W @ W)


def log_so3(R):
    R = np.asarray(R, dtype=np.float64)
    c = np.clip((np.trace(R, axis1=-2, axis2=-1) - 1.0) * 0.5, -1.0, 1.0)
    th = np.arccos(c)
    v = np.stack([R[..., 2, 1] - R[..., 1, 2], R[..., 0, 2] - R[..., 2, 0], R[..., 1, 0] - R[..., 0, 1]], -1)
    s = np.sin(th)
    f = np.where(th < 1e-8, 0.5, th / np.where(np.abs(s) < 1e-300, 1.0, 2.0 * s))
    return f[..., None] * v


def jr_so3(w):
    """right Jacobian with the reference's threshold (IMU/so3.cpp:32-49): theta < 1e-5 -> I."""
    w = np.asarray(w, dtype=np.float64)
    th = np.linalg.norm(w, axis=-1)[..., None, None]
    small = th < 1e-5
    ths = np.where(small, 1.0, th)
    K = hat(w / ths[..., 0])
    J = np.eye(3) - (1 - np.cos(ths)) / ths * K + (1 - np.sin(ths) / ths) * (K @ K)
    return np.where(small, np.eye(3), J)


def quat_from_R(R):
    """Eigen's Quaterniond(Matrix3d), (x,y,z,w); single matrix."""
    m = np.asarray(R, dtype=np.float64)
    q = np.zeros(4)
    t = m[0, 0] + m[1, 1] + m[2, 2]
    if t > 0:
        t = np.sqrt(t + 1.0)
        q[3] = 0.5 * t
        t = 0.5 / t
        q[0] = (m[2, 1] - m[1, 2]) * t
        q[1] = (m[0, 2] - m[2, 0]) * t
        q[2] = (m[1, 0] - m[0, 1]) * t
    else:
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j = (i + 1) % 3
        k = (j + 1) % 3
        t = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0)
        q[i] = 0.5 * t
        t = 0.5 / t
        q[3] = (m[k, j] - m[j, k]) * t
        q[j] = (m[j, i] + m[i, j]) * t
        q[k] = (m[k, i] + m[i, k]) * t
    return q / np.linalg.norm(q)


def R_from_quat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


# ---------------------------------------------------------------------------------------------
# trajectory
# ---------------------------------------------------------------------------------------------
def traj_p(t):
    t = np.asarray(t, dtype=np.float64)
    return np.stack([2 * np.cos(0.4 * t), 2 * np.sin(0.4 * t), 1 + 0.5 * np.sin(0.8 * t)], -1)


def traj_v(t):
    t = np.asarray(t, dtype=np.float64)
    return np.stack([-0.8 * np.sin(0.4 * t), 0.8 * np.cos(0.4 * t), 0.4 * np.cos(0.8 * t)], -1)


def traj_a(t):
    t = np.asarray(t, dtype=np.float64)
    return np.stack([-0.32 * np.cos(0.4 * t), -0.32 * np.sin(0.4 * t), -0.32 * np.sin(0.8 * t)], -1)


def traj_R(t):
    t = np.asarray(t, dtype=np.float64)
    yaw, roll, pitch = 0.4 * t + np.pi / 2, 0.1 * np.sin(0.7 * t), 0.1 * np.cos(0.5 * t)
    z = np.zeros_like(t); o = np.ones_like(t)

    def rot(c, s, ax):
        rows = {0: [[o, z, z], [z, c, -s], [z, s, c]], 1: [[c, z, s], [z, o, z], [-s, z, c]],
                2: [[c, -s, z], [s, c, z], [z, z, o]]}[ax]
        return np.stack([np.stack(r, -1) for r in rows], -2)
    return rot(np.cos(yaw), np.sin(yaw), 2) @ rot(np.cos(pitch), np.sin(pitch), 1) @ rot(np.cos(roll), np.sin(roll), 0)


def traj_omega_body(t, h=1e-5):
    Rm, Rp = traj_R(np.asarray(t) - h), traj_R(np.asarray(t) + h)
    return log_so3(np.swapaxes(Rm, -1, -2) @ Rp) / (2 * h)


# ---------------------------------------------------------------------------------------------
# IMU preintegration, batched over the K-1 keyframe intervals
# ---------------------------------------------------------------------------------------------
def preintegrate(omega, acc, dt):
    """omega, acc: (M, S, 3) bias-corrected samples; returns the (M,142) payload
    [dP3 dV3 dR9 JPg9 JPa9 JVg9 JVa9 JRg9 cov81 dt] following IMU/IMUPreintegrator.cpp:80-139."""
    M, S, _ = omega.shape
    I3 = np.eye(3)
    dP = np.zeros((M, 3)); dV = np.zeros((M, 3)); dR = np.tile(I3, (M, 1, 1))
    JPg = np.zeros((M, 3, 3)); JPa = np.zeros((M, 3, 3)); JVg = np.zeros((M, 3, 3)); JVa = np.zeros((M, 3, 3))
    JRg = np.zeros((M, 3, 3)); cov = np.zeros((M, 9, 9)); T = np.zeros(M)
    dt2 = dt * dt
    for s in range(S):
        w, a = omega[:, s], acc[:, s]
        dRk = exp_so3(w * dt)
        Jr = jr_so3(w * dt)
        Sa = hat(a)
        RS = dR @ Sa
        A = np.tile(np.eye(9), (M, 1, 1))
        A[:, 6:9, 6:9] = np.swapaxes(dRk, 1, 2)
        A[:, 3:6, 6:9] = -RS * dt
        A[:, 0:3, 6:9] = -0.5 * RS * dt2
        A[:, 0:3, 3:6] = I3 * dt
        Bg = np.zeros((M, 9, 3)); Bg[:, 6:9] = Jr * dt
        Ca = np.zeros((M, 9, 3)); Ca[:, 3:6] = dR * dt; Ca[:, 0:3] = 0.5 * dR * dt2
        cov = A @ cov @ np.swapaxes(A, 1, 2) + GYR_MEAS_COV * (Bg @ np.swapaxes(Bg, 1, 2)) \
            + ACC_MEAS_COV * (Ca @ np.swapaxes(Ca, 1, 2))
        RSJ = RS @ JRg
        JPa = JPa + JVa * dt - 0.5 * dR * dt2
        JPg = JPg + JVg * dt - 0.5 * RSJ * dt2
        JVa = JVa - dR * dt
        JVg = JVg - RSJ * dt
        JRg = np.swapaxes(dRk, 1, 2) @ JRg - Jr * dt
        Ra = np.einsum("mij,mj->mi", dR, a)
        dP = dP + dV * dt + 0.5 * Ra * dt2
        dV = dV + Ra * dt
        dRn = dR @ dRk
        # normalizeRotationM: through a w>=0 unit quaternion (IMUPreintegrator.h:166-180)
        for m in range(M):
            q = quat_from_R(dRn[m])
            if q[3] < 0:
                q = -q
            dRn[m] = R_from_quat(q / np.linalg.norm(q))
        dR = dRn
        T = T + dt
    out = np.zeros((M, 142))
    out[:, 0:3] = dP; out[:, 3:6] = dV; out[:, 6:15] = dR.reshape(M, 9)
    out[:, 15:24] = JPg.reshape(M, 9); out[:, 24:33] = JPa.reshape(M, 9)
    out[:, 33:42] = JVg.reshape(M, 9); out[:, 42:51] = JVa.reshape(M, 9)
    out[:, 51:60] = JRg.reshape(M, 9); out[:, 60:141] = cov.reshape(M, 81); out[:, 141] = T
    return out


# ---------------------------------------------------------------------------------------------
# window
# ---------------------------------------------------------------------------------------------
def _project(Rwb, Pwb, Rbc, Pbc, Pw):
    """Pc = Rcb Rwb^T (Pw - Pwb) - Rcb Pbc (IMU/g2otypes.h:243-260); batched over leading dims."""
    Rcb = Rbc.T
    d = Pw - Pwb
    Pc = np.einsum("ij,...j->...i", Rcb, np.einsum("...ji,...j->...i", Rwb, d)) - Rcb @ Pbc
    z = Pc[..., 2]
    zs = np.where(np.abs(z) < 1e-12, 1e-12, z)
    uv = np.stack([FX * Pc[..., 0] / zs + CX, FY * Pc[..., 1] / zs + CY], -1)
    return uv, z


def _gen_tracks(rng, K, N, Rwb, Pwb, Rbc, Pbc, is_line, track=(2, 8), revisit=0.0, revisit_gap=(1, 3)):
    """Landmarks in front of an anchor keyframe, observed in keyframes a..a+L-1 (L ~ U{track}, default U{2..8}) where
    the projection stays in the image with z > 0.1; keeps generating until N have >= 2 observations.
    revisit > 0: with that probability a landmark is seen again in ONE later keyframe beyond a gap of revisit_gap (default 1..3)
    keyframes after its track (a non-consecutive re-observation, as after a short occlusion; a long gap — tens of keyframes — is a
    place revisited: it couples keyframes far apart and leaves the reduced camera system without a band)."""
    lms, obs_lm, obs_kf, obs_uv = [], [], [], []
    count = 0
    while count < N:
        B = max(256, int((N - count) * 1.6))
        a = rng.integers(B, 0, K - 1)
        u = rng.uniform(B, 0, IMG_W); v = rng.uniform(B, 0, IMG_H)
        depth = rng.uniform(B, 1.0, 8.0)
        L = rng.integers(B, track[0], track[1])
        if revisit > 0.0:
            rv = rng.uniform(B) < revisit
            gap = rng.integers(B, revisit_gap[0], revisit_gap[1])
        Pc = np.stack([(u - CX) / FX * depth, (v - CY) / FY * depth, depth], -1)
        Pw = np.einsum("bij,bj->bi", Rwb[a], Pc @ Rbc.T + Pbc) + Pwb[a]
        if is_line:
            dirv = rng.normal((B, 3))
            dirv /= np.linalg.norm(dirv, axis=1, keepdims=True)
            Pe = Pw + dirv * rng.uniform(B, 0.3, 1.5)[:, None]
        for b in range(B):
            if count >= N:
                break
            ks = np.arange(a[b], min(a[b] + L[b], K))
            if revisit > 0.0 and rv[b] and a[b] + L[b] + gap[b] < K:
                ks = np.append(ks, a[b] + L[b] + gap[b])
            uv_s, z_s = _project(Rwb[ks], Pwb[ks], Rbc, Pbc, Pw[b])
            ok = (z_s > 0.1) & (uv_s[:, 0] >= 0) & (uv_s[:, 0] < IMG_W) & (uv_s[:, 1] >= 0) & (uv_s[:, 1] < IMG_H)
            if is_line:
                uv_e, z_e = _project(Rwb[ks], Pwb[ks], Rbc, Pbc, Pe[b])
                ok &= (z_e > 0.1) & (uv_e[:, 0] >= 0) & (uv_e[:, 0] < IMG_W) & (uv_e[:, 1] >= 0) & (uv_e[:, 1] < IMG_H)
            if ok.sum() < 2:
                continue
            ks = ks[ok]
            lms.append(np.concatenate([Pw[b], Pe[b]]) if is_line else Pw[b])
            obs_lm.append(np.full(len(ks), count)); obs_kf.append(ks)
            obs_uv.append(np.concatenate([uv_s[ok], uv_e[ok]], 1) if is_line else uv_s[ok])
            count += 1
    return (np.array(lms), np.concatenate(obs_lm).astype(np.int32), np.concatenate(obs_kf).astype(np.int32),
            np.concatenate(obs_uv))


def make_window(K, Np, Nl, imu=True, seed=0x5EED0003, outlier_frac=0.05, t0=0.0, kf_id0=0, kf_dt=KF_DT, track=(2, 8), revisit=0.0, revisit_gap=(1, 3)):
    """Build one synthetic window.  Returns a dict (see Problem.upload_window) plus 'truth'.
    kf_dt: keyframe spacing in seconds (a multiple of the 5 ms IMU period); track: range of the track lengths;
    revisit: probability of a non-consecutive re-observation (see _gen_tracks).  The defaults are SURVEY 8d's."""
    rng = Rng(seed)
    Rbc, Pbc = T_BS[:3, :3].copy(), T_BS[:3, 3].copy()
    tk = t0 + kf_dt * np.arange(K)
    Rwb, Pwb, Vwb = traj_R(tk), traj_p(tk), traj_v(tk)

    # ---- landmarks and clean observations -------------------------------------------------
    pts, po_pt, po_kf, po_uv = _gen_tracks(rng, K, Np, Rwb, Pwb, Rbc, Pbc, False, track, revisit, revisit_gap) if Np else \
        (np.zeros((0, 3)), np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros((0, 2)))
    lns, lo_ln, lo_kf, lo_uv4 = _gen_tracks(rng, K, Nl, Rwb, Pwb, Rbc, Pbc, True, track, revisit, revisit_gap) if Nl else \
        (np.zeros((0, 6)), np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros((0, 4)))
    Ep, El = len(po_pt), len(lo_ln)

    # ---- measurement noise + outliers ------------------------------------------------------
    def outlier_offsets(E):
        flag = rng.uniform(E) < outlier_frac
        mag = rng.uniform(E, 20.0, 100.0)
        ang = rng.uniform(E, 0.0, 2 * np.pi)
        return flag, np.where(flag, mag, 0.0)[:, None] * np.stack([np.cos(ang), np.sin(ang)], -1)
    po_uv = po_uv + rng.normal((Ep, 2))
    pout, off = outlier_offsets(Ep)
    po_uv = po_uv + off
    lo_uv4 = lo_uv4 + rng.normal((El, 4))
    lout, off = outlier_offsets(El)
    lo_uv4[:, 0:2] += off
    sp = np.concatenate([lo_uv4[:, 0:2], np.ones((El, 1))], 1)
    ep = np.concatenate([lo_uv4[:, 2:4], np.ones((El, 1))], 1)
    lvec = np.cross(sp, ep)
    lo_l = lvec / np.sqrt(lvec[:, 0:1] ** 2 + lvec[:, 1:2] ** 2)   # include/mapFeatures.h:93

    # ---- IMU -------------------------------------------------------------------------------
    bg_st = BG_TRUE + rng.normal(3, 1e-3)
    ba_st = BA_TRUE + rng.normal(3, 1e-2)
    imu_blk = None
    if imu and K > 1:
        S = int(round(kf_dt / IMU_DT))
        M = K - 1
        ts = tk[:-1, None] + IMU_DT * (np.arange(S)[None, :] + 0.5)        # mid-interval samples
        Rs = traj_R(ts)
        w_true = traj_omega_body(ts)
        a_true = np.einsum("msji,msj->msi", Rs, traj_a(ts) - GW)
        w_meas = w_true + BG_TRUE + rng.normal((M, S, 3), np.sqrt(GYR_MEAS_COV))
        a_meas = a_true + BA_TRUE + rng.normal((M, S, 3), np.sqrt(ACC_MEAS_COV))
        pre = preintegrate(w_meas - bg_st, a_meas - ba_st, IMU_DT)
        cov = pre[:, 60:141].reshape(M, 9, 9)
        info_pvr = np.linalg.inv(cov)                                           # mapHandler.cpp:5269
        info_pvr = 0.5 * (info_pvr + np.swapaxes(info_pvr, 1, 2))
        ib = np.zeros((M, 6, 6))
        ib[:, 0:3, 0:3] = np.eye(3) / GYR_BIAS_RW2                              # mapHandler.cpp:5249-5251
        ib[:, 3:6, 3:6] = np.eye(3) / ACC_BIAS_RW2
        ib = ib / pre[:, 141][:, None, None]                                    # :5287
        imu_blk = dict(kf_i=np.arange(M, dtype=np.int32), kf_j=np.arange(1, K, dtype=np.int32),
                       preint=pre, info_pvr=info_pvr.reshape(M, 81), info_bias=ib.reshape(M, 36))

    # ---- initial estimate = truth (+) noise; first keyframe fixed (mapHandler.cpp:5812-5825) ----
    P0 = Pwb.copy(); V0 = Vwb.copy(); R0 = Rwb.copy()
    P0[1:] += rng.normal((K - 1, 3), 0.02)
    V0[1:] += rng.normal((K - 1, 3), 0.05)
    R0[1:] = R0[1:] @ exp_so3(rng.normal((K - 1, 3), 0.01))
    q0 = np.stack([quat_from_R(R0[k]) for k in range(K)])
    pts0 = pts + rng.normal((len(pts), 3), 0.05)
    lns0 = lns + rng.normal((len(lns), 6), 0.05)
    fixed = np.zeros(K, np.uint8); fixed[0] = 1
    vid_pvr = (2 * (kf_id0 + np.arange(K))).astype(np.int32)
    vid_bias = (vid_pvr + 1).astype(np.int32) if imu else np.full(K, -1, np.int32)
    kf = dict(vid_pvr=vid_pvr, vid_bias=vid_bias, P=P0, V=V0, q=q0,
              bg=np.tile(bg_st, (K, 1)), ba=np.tile(ba_st, (K, 1)),
              dbg=np.zeros((K, 3)), dba=np.zeros((K, 3)), fixed_pvr=fixed, fixed_bias=fixed.copy())
    huber = {0: HUBER[0], 1: HUBER[1]}
    if imu:
        huber.update({2: HUBER[2], 3: HUBER[3]})
    return dict(cam=dict(fx=FX, fy=FY, cx=CX, cy=CY, Rbc=Rbc, Pbc=Pbc), gw=GW.copy(), kf=kf,
                points=pts0, lines=lns0,
                po_pt=po_pt, po_kf=po_kf, po_uv=po_uv, po_w=np.ones(Ep),
                lo_ln=lo_ln, lo_kf=lo_kf, lo_l=lo_l, lo_w=np.ones(El),
                imu=imu_blk, prior=None, huber=huber,
                truth=dict(P=Pwb, V=Vwb, R=Rwb, points=pts, lines=lns, bg=BG_TRUE, ba=BA_TRUE,
                           point_outlier=pout, line_outlier=lout),
                meta=dict(K=K, Np=len(pts), Nl=len(lns), Ep=Ep, El=El, imu=bool(imu), seed=int(seed)))


def make_config(idx, scale=1.0, seed=None):
    """BASELINE.json config `idx` (1..5); `scale` < 1 shrinks landmarks for quick tests."""
    c = CONFIGS[idx]
    return make_window(c["K"], max(1, int(c["Np"] * scale)), max(1, int(c["Nl"] * scale)), imu=c["imu"],
                       seed=(0x5EED0000 + idx) if seed is None else seed)


def shard_window(w, rank, world, by="time"):
    """Landmark shard of a window for rank `rank` of `world`: points and lines are partitioned together with all their
    observations; keyframes, IMU edges and the prior are replicated (the library adds pose-side edges on rank 0 only).
    by="time" (round 5): the landmarks are ordered by their FIRST keyframe and cut into `world` equal stretches, so that a rank's
    landmarks lie in one stretch of the window — its Schur partials touch one stretch of the band of the reduced camera system instead
    of all of it (with the index-ordered blocks of rounds 1-4 every rank touched every pose-pair block: landmark indices are random in
    time).  by="index": equal blocks of the landmark index, as before.  Within a shard the landmarks keep their ascending index order.
    shard["pt_index"] / ["ln_index"]: the window indices of the shard's points / lines."""
    def pick(N, ob_lm, ob_kf):
        if by == "index":
            return np.arange((N * rank) // world, (N * (rank + 1)) // world)
        first = _first_kf(ob_lm.astype(np.int64), ob_kf.astype(np.int64), N)
        order = np.argsort(first, kind="stable")
        return np.sort(order[(N * rank) // world:(N * (rank + 1)) // world])
    out = dict(w)
    pidx = pick(len(w["points"]), w["po_pt"], w["po_kf"])
    pos = np.full(len(w["points"]), -1, np.int64); pos[pidx] = np.arange(len(pidx))
    sel = pos[w["po_pt"]] >= 0
    out["points"] = w["points"][pidx]
    out["po_pt"] = pos[w["po_pt"][sel]].astype(np.int32)
    out["po_kf"] = w["po_kf"][sel]; out["po_uv"] = w["po_uv"][sel]; out["po_w"] = w["po_w"][sel]
    lidx = pick(len(w["lines"]), w["lo_ln"], w["lo_kf"])
    lpos = np.full(len(w["lines"]), -1, np.int64); lpos[lidx] = np.arange(len(lidx))
    lsel = lpos[w["lo_ln"]] >= 0
    out["lines"] = w["lines"][lidx]
    out["lo_ln"] = lpos[w["lo_ln"][lsel]].astype(np.int32)
    out["lo_kf"] = w["lo_kf"][lsel]; out["lo_l"] = w["lo_l"][lsel]; out["lo_w"] = w["lo_w"][lsel]
    if w.get("point_fixed") is not None:
        out["point_fixed"] = np.asarray(w["point_fixed"])[pidx]
    if w.get("line_fixed") is not None:
        out["line_fixed"] = np.asarray(w["line_fixed"])[lidx]
    out["shard"] = dict(rank=rank, world=world, by=by, pt_index=pidx, ln_index=lidx)
    return out


def make_visual_window(K=8, Np=300, Nl=60, n_fixed=2, seed=0x1BA, noise_px=0.5, pose_noise=(0.02, 0.01), lm_noise=0.03, track=6):
    """Synthetic input of the pre-init visual-only local BA (MapHandler::localBundleAdjustment, src/mapHandler.cpp:1329-1439,
    feeding levMarquardtOptimizationLBA): K stereo-rig keyframes T_kf_w (camera-to-world 4x4) moving sideways in front of a
    landmark cloud, the first `n_fixed` of them only anchoring landmarks (kf_loc = -1), landmark-major observation lists
    (points: pixel (u, v); lines: normalised image-line coefficients), estimates perturbed away from the truth."""
    rng = np.random.default_rng(seed)
    cam = (FX, FY, CX, CY)
    fx, fy, cx, cy = cam
    T_true = np.zeros((K, 4, 4))
    for k in range(K):
        R = exp_so3(np.array([0.02 * np.sin(0.7 * k), 0.05 * np.sin(0.4 * k + 0.3), 0.01 * k]))
        T_true[k, :3, :3] = R
        T_true[k, :3, 3] = [0.25 * k, 0.03 * np.sin(k), 0.05 * k]
        T_true[k, 3, 3] = 1.0

    def proj(T, X):
        Xc = T[:3, :3].T @ (X - T[:3, 3])
        return np.array([cx + fx * Xc[0] / Xc[2], cy + fy * Xc[1] / Xc[2]]), Xc[2]

    def tracks(n):
        out = []
        for _ in range(n):
            k0 = int(rng.integers(0, max(K - 1, 1)))
            ln = int(rng.integers(2, track + 1))
            out.append(list(range(k0, min(K, k0 + ln))))
        return out

    xyz_true = np.zeros((Np, 3)); po_pt, po_kf, uv = [], [], []
    for i, ks in enumerate(tracks(Np)):
        c = T_true[ks[len(ks) // 2]]
        xyz_true[i] = c[:3, 3] + c[:3, :3] @ np.array([rng.uniform(-2.5, 2.5), rng.uniform(-1.5, 1.5), rng.uniform(4.0, 10.0)])
        for k in ks:
            z, _ = proj(T_true[k], xyz_true[i])
            po_pt.append(i); po_kf.append(k); uv.append(z + rng.normal(0.0, noise_px, 2))
    pq_true = np.zeros((Nl, 6)); lo_ln, lo_kf, l3 = [], [], []
    for i, ks in enumerate(tracks(Nl)):
        c = T_true[ks[len(ks) // 2]]
        P = c[:3, 3] + c[:3, :3] @ np.array([rng.uniform(-2.5, 2.5), rng.uniform(-1.5, 1.5), rng.uniform(4.0, 10.0)])
        Q = P + c[:3, :3] @ np.array([rng.uniform(-1.0, 1.0), rng.uniform(-1.0, 1.0), rng.uniform(-0.5, 0.5)])
        pq_true[i] = np.concatenate([P, Q])
        for k in ks:
            a, _ = proj(T_true[k], P); b, _ = proj(T_true[k], Q)
            a = a + rng.normal(0.0, noise_px, 2); b = b + rng.normal(0.0, noise_px, 2)
            l = np.cross(np.append(a, 1.0), np.append(b, 1.0))
            lo_ln.append(i); lo_kf.append(k); l3.append(l / np.hypot(l[0], l[1]))
    kf_loc = np.array([-1 if k < n_fixed else k - n_fixed for k in range(K)], np.int32)
    T = T_true.copy()
    for k in range(n_fixed, K):
        dR = exp_so3(rng.normal(0.0, pose_noise[1], 3))
        T[k, :3, :3] = T[k, :3, :3] @ dR
        T[k, :3, 3] += rng.normal(0.0, pose_noise[0], 3)
    return dict(cam=cam, T_kf_w=T, kf_loc=kf_loc, xyz=xyz_true + rng.normal(0.0, lm_noise, xyz_true.shape),
                pq=pq_true + rng.normal(0.0, lm_noise, pq_true.shape),
                po_pt=np.array(po_pt, np.int32), po_kf=np.array(po_kf, np.int32), uv=np.array(uv).reshape(-1, 2),
                lo_ln=np.array(lo_ln, np.int32), lo_kf=np.array(lo_kf, np.int32), l3=np.array(l3).reshape(-1, 3),
                truth=dict(T=T_true, xyz=xyz_true, pq=pq_true))


# ---------------------------------------------------------------------------------------------
# consecutive sliding windows of one sequence (plba_slide_window; src/mapHandler.cpp:1178-1221, 4815-4825, 5769-5783)
# ---------------------------------------------------------------------------------------------
def make_sequence(K, n_windows, Np, Nl, imu=True, seed=0x5EED0051, **kw):
    """A trajectory of K + n_windows - 1 keyframes with the landmark density of a K-keyframe window of Np points / Nl lines: the
    material consecutive sliding windows are cut from (window_at)."""
    Kt = K + n_windows - 1
    return make_window(Kt, max(1, int(round(Np * Kt / K))), max(1, int(round(Nl * Kt / K))), imu=imu, seed=seed, **kw)


def _first_kf(ob_lm, ob_kf, N):
    f = np.full(N, 1 << 30, np.int64)
    np.minimum.at(f, ob_lm, ob_kf)
    return f


def window_at(seq, k0, K, prev=None):
    """The reference's local map for the sliding window of keyframes [k0, k0 + K) of a sequence: every landmark whose FIRST observation is
    from a window keyframe (src/mapHandler.cpp:5769-5783) and that has been seen at least twice by the newest window keyframe (a map
    landmark is born from a match), with its observations up to that keyframe.  Landmark order = map order: the landmarks of the previous
    window `prev` that are still local keep their relative order, the ones that entered the map since follow (ascending sequence index),
    as MapPoint::idx grows in the reference.  Estimates are the sequence's initial ones; window_from_results() replaces what a previous
    BA optimised.  `ids` records which sequence keyframes / landmarks the window holds."""
    k1 = k0 + K
    out = dict(cam=seq["cam"], gw=seq["gw"], huber=dict(seq["huber"]), prior=None)
    ks = np.arange(k0, k1)
    kf = {k: np.ascontiguousarray(v[ks]) for k, v in seq["kf"].items()}
    fixed = np.zeros(K, np.uint8); fixed[0] = 1      # the oldest window keyframe is fixed (mapHandler.cpp:5812-5825)
    kf["fixed_pvr"] = fixed; kf["fixed_bias"] = fixed.copy()
    out["kf"] = kf
    ids = dict(k0=int(k0), K=int(K))
    for kind, (lm_key, ob_lm_key, ob_kf_key, meas_key, w_key) in dict(points=("points", "po_pt", "po_kf", "po_uv", "po_w"), lines=("lines", "lo_ln", "lo_kf", "lo_l", "lo_w")).items():
        ob_lm, ob_kf = seq[ob_lm_key].astype(np.int64), seq[ob_kf_key].astype(np.int64)
        N = len(seq[lm_key])
        first = _first_kf(ob_lm, ob_kf, N)
        inw = (ob_kf >= k0) & (ob_kf < k1)
        cnt = np.bincount(ob_lm[inw], minlength=N)
        ok = (first >= k0) & (first < k1) & (cnt >= 2)
        if prev is not None:
            pid = prev["ids"][kind]
            stay = pid[ok[pid]]
            new = np.setdiff1d(np.flatnonzero(ok), pid, assume_unique=True)
            sel = np.concatenate([stay, new]).astype(np.int64)
        else:
            sel = np.flatnonzero(ok).astype(np.int64)
        rank = np.full(N, -1, np.int64); rank[sel] = np.arange(len(sel))
        take = inw & (rank[ob_lm] >= 0)
        o = np.flatnonzero(take)
        o = o[np.lexsort((ob_kf[o], rank[ob_lm[o]]))]      # landmark-major in the window's landmark order, keyframes ascending (kf_obs_list order)
        out[lm_key] = np.ascontiguousarray(seq[lm_key][sel])
        out[ob_lm_key] = rank[ob_lm[o]].astype(np.int32)
        out[ob_kf_key] = (ob_kf[o] - k0).astype(np.int32)
        out[meas_key] = np.ascontiguousarray(seq[meas_key][o])
        out[w_key] = np.ascontiguousarray(seq[w_key][o])
        ids[kind] = sel
        ids[kind + "_obs"] = o
    if seq.get("imu") is not None:
        im = seq["imu"]
        m = np.arange(k0, k1 - 1)      # edge m joins sequence keyframes m, m + 1
        out["imu"] = dict(kf_i=(im["kf_i"][m] - k0).astype(np.int32), kf_j=(im["kf_j"][m] - k0).astype(np.int32), preint=im["preint"][m], info_pvr=im["info_pvr"][m], info_bias=im["info_bias"][m])
    else:
        out["imu"] = None
    out["ids"] = ids
    out["meta"] = dict(K=K, Np=len(out["points"]), Nl=len(out["lines"]), Ep=len(out["po_pt"]), El=len(out["lo_ln"]), imu=out["imu"] is not None, seed=seq["meta"]["seed"])
    return out


def window_from_results(w_next, w_prev, res_prev):
    """What the reference hands the next local BA: the keyframes and landmarks the previous window's BA optimised carry its results (the
    write-back of src/mapHandler.cpp:6202-6239 — P, V, R, dbg, dba of every window keyframe, every local landmark), everything new keeps
    its initial estimate.  res_prev: protocol.results() of the previous window's problem."""
    w = dict(w_next)
    w["kf"] = {k: np.array(v, copy=True) for k, v in w_next["kf"].items()}
    sh = w_next["ids"]["k0"] - w_prev["ids"]["k0"]
    Kp = w_prev["ids"]["K"]
    n = Kp - sh      # keyframes both windows hold
    for key in ("P", "V", "q", "dbg", "dba"):
        w["kf"][key][:n] = np.asarray(res_prev[key])[sh:sh + n]
    for kind, key in (("points", "points"), ("lines", "lines")):
        arr = np.array(w_next[key], copy=True)
        prev_ids = w_prev["ids"][kind]
        pos = {int(i): j for j, i in enumerate(prev_ids)}
        for j, i in enumerate(w_next["ids"][kind]):
            q = pos.get(int(i))
            if q is not None:
                arr[j] = np.asarray(res_prev[key])[q]
        w[key] = arr
    return w


def slide_delta(w_prev, w_next):
    """The arguments of plba_slide_window (abi.Problem.slide_window) that turn the resident window w_prev into w_next: both from
    window_at() on one sequence, w_next built with prev=w_prev."""
    sh = w_next["ids"]["k0"] - w_prev["ids"]["k0"]
    Kp, Kn = w_prev["ids"]["K"], w_next["ids"]["K"]
    n_keep = Kp - sh
    d = dict(n_drop=int(sh))
    d["kf"] = {k: np.ascontiguousarray(v[n_keep:]) for k, v in w_next["kf"].items() if k not in ("fixed_pvr", "fixed_bias")}
    d["fixed_pvr"], d["fixed_bias"] = w_next["kf"]["fixed_pvr"], w_next["kf"]["fixed_bias"]
    if w_next.get("imu") is not None:
        im = w_next["imu"]
        new = (im["kf_i"] >= n_keep) | (im["kf_j"] >= n_keep)
        d["imu"] = {k: np.ascontiguousarray(v[new]) for k, v in im.items()}
    for kind, (lm_key, ob_lm_key, ob_kf_key, meas_key, w_key, dropkey) in dict(points=("points", "po_pt", "po_kf", "po_uv", "po_w", "drop_point"),
                                                                                lines=("lines", "lo_ln", "lo_kf", "lo_l", "lo_w", "drop_line")).items():
        pid, nid = w_prev["ids"][kind], w_next["ids"][kind]
        pos_prev = {int(i): j for j, i in enumerate(pid)}
        nstay = sum(1 for i in nid if int(i) in pos_prev)
        assert all(int(i) in pos_prev for i in nid[:nstay]) and not any(int(i) in pos_prev for i in nid[nstay:]), "w_next must list the staying landmarks first (window_at(prev=...))"
        N0 = len(pid)
        # landmarks of the previous window that the next one does not hold and that the slide's own rule (an observation from a dropped
        # keyframe) would keep: dropped by mask
        nset = set(int(i) for i in nid)
        auto = np.zeros(N0, bool)
        auto[w_prev[ob_lm_key][w_prev[ob_kf_key] < sh]] = True
        mask = np.array([(int(i) not in nset) for i in pid], np.uint8)
        mask[auto] = 0
        d[dropkey] = mask if mask.any() else None
        d[lm_key] = np.ascontiguousarray(w_next[lm_key][nstay:])
        # observations the previous window did not hold: those from the added keyframes, and every observation of an added landmark
        old_obs = set(int(o) for o in w_prev["ids"][kind + "_obs"])
        isnew = np.array([int(o) not in old_obs for o in w_next["ids"][kind + "_obs"]], bool)
        lm_next = w_next[ob_lm_key][isnew].astype(np.int64)
        idx_before = np.where(lm_next < nstay, np.array([pos_prev[int(nid[j])] for j in np.minimum(lm_next, max(nstay - 1, 0))], np.int64) if nstay else 0, N0 + (lm_next - nstay))
        o = np.argsort(idx_before, kind="stable")
        d[ob_lm_key] = idx_before[o].astype(np.int32)
        d[ob_kf_key] = w_next[ob_kf_key][isnew][o].astype(np.int32)
        d[meas_key] = np.ascontiguousarray(w_next[meas_key][isnew][o])
        d[w_key] = np.ascontiguousarray(w_next[w_key][isnew][o])
    return d
