"""ORACLE (test infrastructure, parity unpinned): numpy restatement of the monocular two-view initialisation
([UPSTREAM] openvslam initialize::perspective / solve::homography_solver / solve::fundamental_solver / initialize::base, which follow
ORB-SLAM's Initializer; reached from lpslam through feed_monocular_frame, /root/reference/src/Trackers/OpenVSLAMTracker.cpp:120).
Independent of lpslam_amd/host/two_view.cpp in its linear algebra (numpy SVD instead of Jacobi eigen-decompositions); the 8-match
sampler (xorshift32, sampling without replacement) is the same so that both sides test the same hypotheses.
Only tests/ may import this module."""
import numpy as np


def _normalize(p):
    m = p.mean(axis=0)
    d = np.abs(p - m).mean(axis=0)
    s = 1.0 / d
    T = np.array([[s[0], 0, -m[0] * s[0]], [0, s[1], -m[1] * s[1]], [0, 0, 1.0]])
    return (p - m) * s, T


def homography(x1, x2):
    rows = []
    for (u1, v1), (u2, v2) in zip(x1, x2):
        rows.append([0, 0, 0, -u1, -v1, -1, v2 * u1, v2 * v1, v2])
        rows.append([u1, v1, 1, 0, 0, 0, -u2 * u1, -u2 * v1, -u2])
    _, _, vt = np.linalg.svd(np.array(rows, float))
    return vt[-1].reshape(3, 3)


def fundamental(x1, x2):
    a = np.array([[u2 * u1, u2 * v1, u2, v2 * u1, v2 * v1, v2, u1, v1, 1.0] for (u1, v1), (u2, v2) in zip(x1, x2)])
    _, _, vt = np.linalg.svd(a, full_matrices=True)
    f = vt[-1].reshape(3, 3)
    u, w, vt2 = np.linalg.svd(f)
    w[2] = 0
    return u @ np.diag(w) @ vt2


def check_h(h21, h12, p1, p2, sigma):
    th, inv = 5.991, 1.0 / sigma ** 2
    one = np.ones((len(p1), 1))
    a = (h12 @ np.hstack([p2, one]).T).T; a = a[:, :2] / a[:, 2:]
    b = (h21 @ np.hstack([p1, one]).T).T; b = b[:, :2] / b[:, 2:]
    c1 = ((p1 - a) ** 2).sum(axis=1) * inv; c2 = ((p2 - b) ** 2).sum(axis=1) * inv
    score = np.where(c1 <= th, th - c1, 0).sum() + np.where(c2 <= th, th - c2, 0).sum()
    return score, (c1 <= th) & (c2 <= th)


def check_f(f21, p1, p2, sigma):
    th, ths, inv = 3.841, 5.991, 1.0 / sigma ** 2
    one = np.ones((len(p1), 1))
    x1 = np.hstack([p1, one]); x2 = np.hstack([p2, one])
    l2 = (f21 @ x1.T).T; l1 = (f21.T @ x2.T).T
    c1 = (l2 * x2).sum(axis=1) ** 2 / (l2[:, 0] ** 2 + l2[:, 1] ** 2) * inv
    c2 = (l1 * x1).sum(axis=1) ** 2 / (l1[:, 0] ** 2 + l1[:, 1] ** 2) * inv
    score = np.where(c1 <= th, ths - c1, 0).sum() + np.where(c2 <= th, ths - c2, 0).sum()
    return score, (c1 <= th) & (c2 <= th)


def triangulate(p1m, p2m, x1, x2):
    a = np.array([x1[0] * p1m[2] - p1m[0], x1[1] * p1m[2] - p1m[1], x2[0] * p2m[2] - p2m[0], x2[1] * p2m[2] - p2m[1]])
    _, _, vt = np.linalg.svd(a)
    x = vt[-1]
    return x[:3] / x[3]


def check_pose(r, t, k, p1, p2, inl, th2):
    km = np.array([[k[0], 0, k[2]], [0, k[1], k[3]], [0, 0, 1.0]])
    p1m = km @ np.hstack([np.eye(3), np.zeros((3, 1))]); p2m = km @ np.hstack([r, t.reshape(3, 1)])
    o2 = -r.T @ t
    pts = np.full((len(p1), 3), np.nan); good = np.zeros(len(p1), bool); cosines = []
    n_good = 0
    for i in range(len(p1)):
        if not inl[i]:
            continue
        x = triangulate(p1m, p2m, p1[i], p2[i])
        if not np.all(np.isfinite(x)):
            continue
        d2 = x - o2
        cosp = x @ d2 / (np.linalg.norm(x) * np.linalg.norm(d2))
        if x[2] <= 0 and cosp < 0.99998:
            continue
        x2 = r @ x + t
        if x2[2] <= 0 and cosp < 0.99998:
            continue
        e1 = np.array([k[0] * x[0] / x[2] + k[2], k[1] * x[1] / x[2] + k[3]]) - p1[i]
        if e1 @ e1 > th2:
            continue
        e2 = np.array([k[0] * x2[0] / x2[2] + k[2], k[1] * x2[1] / x2[2] + k[3]]) - p2[i]
        if e2 @ e2 > th2:
            continue
        cosines.append(cosp); pts[i] = x; n_good += 1
        good[i] = cosp < 0.99998
    par = 0.0
    if cosines:
        cs = np.sort(cosines)
        par = float(np.degrees(np.arccos(np.clip(cs[min(50, len(cs) - 1)], -1, 1))))
    return n_good, pts, good, par


def hyps_from_f(f21, k):
    km = np.array([[k[0], 0, k[2]], [0, k[1], k[3]], [0, 0, 1.0]])
    e = km.T @ f21 @ km
    u, _, vt = np.linalg.svd(e)
    if np.linalg.det(u) < 0: u[:, 2] *= -1
    if np.linalg.det(vt) < 0: vt[2] *= -1
    w = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]])
    r1 = u @ w @ vt; r2 = u @ w.T @ vt
    if np.linalg.det(r1) < 0: r1 = -r1
    if np.linalg.det(r2) < 0: r2 = -r2
    t = u[:, 2] / np.linalg.norm(u[:, 2])
    return [(r1, t), (r2, t), (r1, -t), (r2, -t)]


def hyps_from_h(h21, k):
    km = np.array([[k[0], 0, k[2]], [0, k[1], k[3]], [0, 0, 1.0]])
    a = np.linalg.inv(km) @ h21 @ km
    u, w, vt = np.linalg.svd(a)
    v = vt.T
    s = np.linalg.det(u) * np.linalg.det(vt)
    d1, d2, d3 = w
    if d1 / d2 < 1.00001 or d2 / d3 < 1.00001:
        return None
    aux1 = np.sqrt((d1 * d1 - d2 * d2) / (d1 * d1 - d3 * d3)); aux3 = np.sqrt((d2 * d2 - d3 * d3) / (d1 * d1 - d3 * d3))
    x1 = [aux1, aux1, -aux1, -aux1]; x3 = [aux3, -aux3, aux3, -aux3]
    out = []
    aux_s = np.sqrt((d1 * d1 - d2 * d2) * (d2 * d2 - d3 * d3)) / ((d1 + d3) * d2)
    ct = (d2 * d2 + d1 * d3) / ((d1 + d3) * d2)
    for i, st in enumerate([aux_s, -aux_s, -aux_s, aux_s]):
        rp = np.array([[ct, 0, -st], [0, 1, 0], [st, 0, ct]])
        r = s * u @ rp @ v.T
        t = u @ np.array([x1[i] * (d1 - d3), 0, -x3[i] * (d1 - d3)])
        out.append((r, t / np.linalg.norm(t)))
    aux_s = np.sqrt((d1 * d1 - d2 * d2) * (d2 * d2 - d3 * d3)) / ((d1 - d3) * d2)
    cp = (d1 * d3 - d2 * d2) / ((d1 - d3) * d2)
    for i, sp in enumerate([aux_s, -aux_s, -aux_s, aux_s]):
        rp = np.array([[cp, 0, sp], [0, -1, 0], [sp, 0, -cp]])
        r = s * u @ rp @ v.T
        t = u @ np.array([x1[i] * (d1 + d3), 0, x3[i] * (d1 + d3)])
        out.append((r, t / np.linalg.norm(t)))
    return out


class _Rng:
    def __init__(self, seed):
        self.s = (seed or 1) & 0xFFFFFFFF

    def next(self):
        s = self.s
        s ^= (s << 13) & 0xFFFFFFFF; s ^= s >> 17; s ^= (s << 5) & 0xFFFFFFFF
        self.s = s
        return s


def initialize(k, kp_ref, kp_cur, matches, sigma=1.0, ransac_iters=100, seed=0x9E3779B9, min_triangulated=50, parallax_thr=1.0, reproj_thr=4.0):
    m = np.asarray(matches).reshape(-1, 2)
    n = len(m)
    res = dict(ok=False, model=-1)
    if n < 8:
        return res
    p1 = np.asarray(kp_ref, float).reshape(-1, 2)[m[:, 0]]; p2 = np.asarray(kp_cur, float).reshape(-1, 2)[m[:, 1]]
    n1, t1 = _normalize(p1); n2, t2 = _normalize(p2)
    rng = _Rng(seed)
    best_h = best_f = -1.0; bh = bf = None; ih = if_ = None
    for _ in range(ransac_iters):
        avail = list(range(n)); left = n; idx = []
        for _k in range(8):
            r = rng.next() % left
            idx.append(avail[r]); avail[r] = avail[left - 1]; left -= 1
        hn = homography(n1[idx], n2[idx]); fn = fundamental(n1[idx], n2[idx])
        h21 = np.linalg.inv(t2) @ hn @ t1; f21 = t2.T @ fn @ t1
        if abs(np.linalg.det(h21)) > 1e-300:
            sh, inl = check_h(h21, np.linalg.inv(h21), p1, p2, sigma)
            if sh > best_h: best_h, bh, ih = sh, h21, inl
        sf, inl = check_f(f21, p1, p2, sigma)
        if sf > best_f: best_f, bf, if_ = sf, f21, inl
    # recompute = true: each best model again from all its inliers
    if bh is not None and ih.sum() >= 8:
        h21 = np.linalg.inv(t2) @ homography(n1[ih], n2[ih]) @ t1
        if abs(np.linalg.det(h21)) > 1e-300:
            best_h, ih2 = check_h(h21, np.linalg.inv(h21), p1, p2, sigma); bh, ih = h21, ih2
    if bf is not None and if_.sum() >= 8:
        f21 = t2.T @ fundamental(n1[if_], n2[if_]) @ t1
        best_f, if_ = check_f(f21, p1, p2, sigma); bf = f21
    res.update(score_h=max(best_h, 0), score_f=max(best_f, 0), H=bh, F=bf)
    use_h = res["score_h"] / (res["score_h"] + res["score_f"]) > 0.40
    res["model"] = 0 if use_h else 1
    inl = ih if use_h else if_
    res["inlier"] = inl
    hyps = hyps_from_h(bh, k) if use_h else hyps_from_f(bf, k)
    if hyps is None:
        return res
    th2 = reproj_thr * sigma * sigma
    best = (0, None); second = 0
    for r, t in hyps:
        ng, pts, good, par = check_pose(r, t, k, p1, p2, inl, th2)
        if ng > best[0]:
            second = best[0]; best = (ng, (r, t, pts, good, par))
        elif ng > second:
            second = ng
    res["n_valid"] = best[0]
    if best[1] is None:
        return res
    r, t, pts, good, par = best[1]
    res.update(parallax_deg=par)
    if best[0] < max(min_triangulated, int(0.9 * inl.sum())) or second > 0.8 * best[0] or par < parallax_thr:
        return res
    res.update(ok=True, R=r, t=t, points=pts, triangulated=good)
    return res


# ---- [UPSTREAM] solve::sim3_solver: Horn's absolute orientation on 3-match samples, RANSAC over the reprojection error in both images
def horn(x1, x2, fix_scale):
    """x1 = s R x2 + t from n >= 3 matched 3D points (rows); returns (R, t, s) or None"""
    x1 = np.asarray(x1, float); x2 = np.asarray(x2, float)
    o1, o2 = x1.mean(axis=0), x2.mean(axis=0)
    a, b = x2 - o2, x1 - o1
    m = a.T @ b                                          # m[i, j] = sum a_i b_j
    n = np.array([[m[0, 0] + m[1, 1] + m[2, 2], m[1, 2] - m[2, 1], m[2, 0] - m[0, 2], m[0, 1] - m[1, 0]],
                  [m[1, 2] - m[2, 1], m[0, 0] - m[1, 1] - m[2, 2], m[0, 1] + m[1, 0], m[2, 0] + m[0, 2]],
                  [m[2, 0] - m[0, 2], m[0, 1] + m[1, 0], -m[0, 0] + m[1, 1] - m[2, 2], m[1, 2] + m[2, 1]],
                  [m[0, 1] - m[1, 0], m[2, 0] + m[0, 2], m[1, 2] + m[2, 1], -m[0, 0] - m[1, 1] + m[2, 2]]])
    _, vec = np.linalg.eigh(n)
    q = vec[:, -1]
    q = q / np.linalg.norm(q)
    w, x, y, z = q
    r = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    s = 1.0
    if not fix_scale:
        p3 = a @ r.T
        nom, den = float((b * p3).sum()), float((p3 * p3).sum())
        if not (den > 0) or not (nom > 0):
            return None
        s = nom / den
    return r, o1 - s * (r @ o2), s


def sim3_solve_ransac(p1c, p2c, obs1, obs2, is1, is2, cam1, cam2, fix_scale, iterations=200, seed=0x9E3779B9):
    """returns (n_inliers, s12 (qw qx qy qz tx ty tz s) or None, inlier flags)"""
    p1c = np.asarray(p1c, float); p2c = np.asarray(p2c, float); obs1 = np.asarray(obs1, float); obs2 = np.asarray(obs2, float)
    n = len(p1c)
    best, best_s12, best_inl = 0, None, np.zeros(n, bool)
    if n < 3:
        return 0, None, best_inl
    rng = _Rng(seed)
    for _ in range(iterations):
        avail = list(range(n)); left = n; idx = []
        for _k in range(3):
            r = rng.next() % left
            idx.append(avail[r]); avail[r] = avail[left - 1]; left -= 1
        h = horn(p1c[idx], p2c[idx], fix_scale)
        if h is None:
            continue
        r, t, s = h
        q1 = s * (p2c @ r.T) + t
        q2 = ((p1c - t) @ r) / s
        with np.errstate(divide="ignore", invalid="ignore"):
            e1 = np.stack([cam1[0] * q1[:, 0] / q1[:, 2] + cam1[2], cam1[1] * q1[:, 1] / q1[:, 2] + cam1[3]], 1) - obs1
            e2 = np.stack([cam2[0] * q2[:, 0] / q2[:, 2] + cam2[2], cam2[1] * q2[:, 1] / q2[:, 2] + cam2[3]], 1) - obs2
            inl = (q1[:, 2] > 0) & (q2[:, 2] > 0) & ((e1 * e1).sum(1) * is1 < 9.210) & ((e2 * e2).sum(1) * is2 < 9.210)
        c = int(inl.sum())
        if c > best:
            m = r.reshape(9)
            tr = m[0] + m[4] + m[8]
            if tr > 0:
                s4 = np.sqrt(tr + 1.0) * 2; q = [0.25 * s4, (m[7] - m[5]) / s4, (m[2] - m[6]) / s4, (m[3] - m[1]) / s4]
            elif m[0] > m[4] and m[0] > m[8]:
                s4 = np.sqrt(1.0 + m[0] - m[4] - m[8]) * 2; q = [(m[7] - m[5]) / s4, 0.25 * s4, (m[1] + m[3]) / s4, (m[2] + m[6]) / s4]
            elif m[4] > m[8]:
                s4 = np.sqrt(1.0 + m[4] - m[0] - m[8]) * 2; q = [(m[2] - m[6]) / s4, (m[1] + m[3]) / s4, 0.25 * s4, (m[5] + m[7]) / s4]
            else:
                s4 = np.sqrt(1.0 + m[8] - m[0] - m[4]) * 2; q = [(m[3] - m[1]) / s4, (m[2] + m[6]) / s4, (m[5] + m[7]) / s4, 0.25 * s4]
            best, best_s12, best_inl = c, np.array(q + [t[0], t[1], t[2], s], float), inl.copy()
    return best, best_s12, best_inl


# ---- [UPSTREAM] solve::pnp_solver: EPnP (Lepetit, Moreno-Noguer, Fua, IJCV 2009) on 4-match samples + RANSAC + a refit on the inliers.
# The steps are the published ones (control points, barycentric coordinates, M^T M, the four smallest eigenvectors, the betas from the
# control points' distances by three linearised guesses + Gauss-Newton, rigid motion, best reprojection error); the linear algebra is
# written operation by operation like lpslam_amd/host/two_view.cpp's (cyclic Jacobi with the same sweep order, normal equations with
# partial pivoting) because EPnP on FOUR matches has a four-dimensional null space: which basis an eigen-solver returns for it decides
# which local solution the betas land in, and both sides must test the same hypotheses for the closed-loop comparison to mean anything.
def _jacobi_eig(A):
    """eigenvalues ascending, eigenvectors as columns: cyclic Jacobi, 64 sweeps at most (sym_eigen_jacobi)"""
    a = np.array(A, float); n = a.shape[0]
    v = np.eye(n)
    for _ in range(64):
        off = 0.0; diag = 0.0
        for i in range(n):
            for j in range(n):
                if i == j:
                    diag += a[i, j] * a[i, j]
                else:
                    off += a[i, j] * a[i, j]
        if off <= 1e-30 * (diag + 1e-300):
            break
        for p in range(n - 1):
            for q in range(p + 1, n):
                apq = a[p, q]
                if apq == 0.0:
                    continue
                theta = (a[q, q] - a[p, p]) / (2.0 * apq)
                t = (1.0 if theta >= 0 else -1.0) / (abs(theta) + np.sqrt(theta * theta + 1.0))
                c = 1.0 / np.sqrt(t * t + 1.0); sn = t * c
                akp = a[:, p].copy(); akq = a[:, q].copy()
                a[:, p] = c * akp - sn * akq; a[:, q] = sn * akp + c * akq
                apk = a[p, :].copy(); aqk = a[q, :].copy()
                a[p, :] = c * apk - sn * aqk; a[q, :] = sn * apk + c * aqk
                vkp = v[:, p].copy(); vkq = v[:, q].copy()
                v[:, p] = c * vkp - sn * vkq; v[:, q] = sn * vkp + c * vkq
    order = sorted(range(n), key=lambda x: a[x, x])
    return np.array([a[o, o] for o in order]), v[:, order]


def _lsq_small(A, b):
    """least squares by the normal equations and Gaussian elimination with partial pivoting (lsq_small); None when singular"""
    A = np.asarray(A, float); b = np.asarray(b, float)
    m, k = A.shape
    N = np.zeros((k, k + 1))
    for i in range(k):
        for j in range(k):
            s = 0.0
            for r in range(m):
                s += A[r, i] * A[r, j]
            N[i, j] = s
        s = 0.0
        for r in range(m):
            s += A[r, i] * b[r]
        N[i, k] = s
    for c in range(k):
        piv = c
        for r in range(c + 1, k):
            if abs(N[r, c]) > abs(N[piv, c]):
                piv = r
        if not (abs(N[piv, c]) > 1e-300):
            return None
        if piv != c:
            N[[c, piv]] = N[[piv, c]]
        for r in range(c + 1, k):
            f = N[r, c] / N[c, c]
            for j in range(c, k + 1):
                N[r, j] -= f * N[c, j]
    x = np.zeros(k)
    for i in range(k - 1, -1, -1):
        s2 = N[i, k]
        for j in range(i + 1, k):
            s2 -= N[i, j] * x[j]
        x[i] = s2 / N[i, i]
    return x


def _horn_jacobi(x1, x2):
    """x1 = R x2 + t (rigid), Horn's quaternion form with the Jacobi eigen-solver (horn_absolute_orientation, scale fixed)"""
    n = len(x1)
    o1 = np.zeros(3); o2 = np.zeros(3)
    for i in range(n):
        o1 += x1[i]; o2 += x2[i]
    o1 /= n; o2 /= n
    M = np.zeros((3, 3))                                  # M[a][b] = sum (x2 - o2)[a] (x1 - o1)[b]
    for i in range(n):
        M += np.outer(x2[i] - o2, x1[i] - o1)
    M = M.reshape(9)
    N = np.array([[M[0] + M[4] + M[8], M[5] - M[7], M[6] - M[2], M[1] - M[3]],
                  [M[5] - M[7], M[0] - M[4] - M[8], M[1] + M[3], M[6] + M[2]],
                  [M[6] - M[2], M[1] + M[3], -M[0] + M[4] - M[8], M[5] + M[7]],
                  [M[1] - M[3], M[6] + M[2], M[5] + M[7], -M[0] - M[4] + M[8]]])
    _, vec = _jacobi_eig(N)
    q = vec[:, 3].copy()
    qn = np.sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3])
    if not (qn > 0):
        return None
    q /= qn
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    t = np.array([o1[r] - 1.0 * (R[r, 0] * o2[0] + R[r, 1] * o2[1] + R[r, 2] * o2[2]) for r in range(3)])
    return R, t


def epnp_solve(pw, uv, cam):
    """world -> camera (R, t) from n >= 4 landmark / pixel matches, or None (epnp_solve)"""
    pw = np.asarray(pw, float); uv = np.asarray(uv, float)
    n = len(pw)
    if n < 4:
        return None
    fu, fv, uc, vc = (float(c) for c in cam[:4])
    cws = np.zeros((4, 3))
    for i in range(n):
        cws[0] += pw[i]
    cws[0] /= n
    Cm = np.zeros((3, 3))
    for i in range(n):
        d = pw[i] - cws[0]
        Cm += np.outer(d, d)
    ev, evec = _jacobi_eig(Cm)
    for j in range(1, 4):
        col = 3 - j
        k = np.sqrt(max(ev[col], 0.0) / n)
        cws[j] = cws[0] + k * evec[:, col]
    CC = np.zeros(9)
    for a in range(3):
        for j in range(1, 4):
            CC[a * 3 + (j - 1)] = cws[j][a] - cws[0][a]
    det = CC[0] * (CC[4] * CC[8] - CC[5] * CC[7]) - CC[1] * (CC[3] * CC[8] - CC[5] * CC[6]) + CC[2] * (CC[3] * CC[7] - CC[4] * CC[6])
    if not (abs(det) > 1e-300):
        return None
    CI = np.array([(CC[4] * CC[8] - CC[5] * CC[7]) / det, (CC[2] * CC[7] - CC[1] * CC[8]) / det, (CC[1] * CC[5] - CC[2] * CC[4]) / det,
                   (CC[5] * CC[6] - CC[3] * CC[8]) / det, (CC[0] * CC[8] - CC[2] * CC[6]) / det, (CC[2] * CC[3] - CC[0] * CC[5]) / det,
                   (CC[3] * CC[7] - CC[4] * CC[6]) / det, (CC[1] * CC[6] - CC[0] * CC[7]) / det, (CC[0] * CC[4] - CC[1] * CC[3]) / det])
    al = np.zeros((n, 4))
    for i in range(n):
        d0, d1, d2 = pw[i] - cws[0]
        for j in range(3):
            al[i, 1 + j] = CI[j * 3] * d0 + CI[j * 3 + 1] * d1 + CI[j * 3 + 2] * d2
        al[i, 0] = 1.0 - al[i, 1] - al[i, 2] - al[i, 3]
    MtM = np.zeros((12, 12))
    for i in range(n):
        r1 = np.zeros(12); r2 = np.zeros(12)
        for j in range(4):
            a = al[i, j]
            r1[3 * j] = a * fu; r1[3 * j + 2] = a * (uc - uv[i, 0])
            r2[3 * j + 1] = a * fv; r2[3 * j + 2] = a * (vc - uv[i, 1])
        MtM += np.outer(r1, r1) + np.outer(r2, r2)
    _, mvec = _jacobi_eig(MtM)
    v = [mvec[:, k].copy() for k in range(4)]
    pa, pb = (0, 0, 0, 1, 1, 2), (1, 2, 3, 2, 3, 3)
    dot3 = lambda x, y: x[0] * y[0] + x[1] * y[1] + x[2] * y[2]
    L = np.zeros((6, 10)); rho = np.zeros(6)
    for p in range(6):
        dv = [v[k][3 * pa[p]:3 * pa[p] + 3] - v[k][3 * pb[p]:3 * pb[p] + 3] for k in range(4)]
        L[p] = [dot3(dv[0], dv[0]), 2.0 * dot3(dv[0], dv[1]), dot3(dv[1], dv[1]), 2.0 * dot3(dv[0], dv[2]), 2.0 * dot3(dv[1], dv[2]), dot3(dv[2], dv[2]),
                2.0 * dot3(dv[0], dv[3]), 2.0 * dot3(dv[1], dv[3]), 2.0 * dot3(dv[2], dv[3]), dot3(dv[3], dv[3])]
        d2 = 0.0
        for a in range(3):
            d = cws[pa[p]][a] - cws[pb[p]][a]
            d2 += d * d
        rho[p] = d2
    best_err, best = 1e300, None
    for guess in range(3):
        be = [0.0, 0.0, 0.0, 0.0]
        if guess == 0:
            x4 = _lsq_small(L[:, [0, 1, 3, 6]], rho)
            if x4 is None:
                continue
            if x4[0] < 0:
                be[0] = np.sqrt(-x4[0]); be[1] = -x4[1] / be[0]; be[2] = -x4[2] / be[0]; be[3] = -x4[3] / be[0]
            else:
                be[0] = np.sqrt(x4[0]); be[1] = x4[1] / be[0]; be[2] = x4[2] / be[0]; be[3] = x4[3] / be[0]
        elif guess == 1:
            x3 = _lsq_small(L[:, [0, 1, 2]], rho)
            if x3 is None:
                continue
            if x3[0] < 0:
                be[0] = np.sqrt(-x3[0]); be[1] = np.sqrt(-x3[2]) if x3[2] < 0 else 0.0
            else:
                be[0] = np.sqrt(x3[0]); be[1] = np.sqrt(x3[2]) if x3[2] > 0 else 0.0
            if x3[1] < 0:
                be[0] = -be[0]
        else:
            x5 = _lsq_small(L[:, [0, 1, 2, 3, 4]], rho)
            if x5 is None:
                continue
            if x5[0] < 0:
                be[0] = np.sqrt(-x5[0]); be[1] = np.sqrt(-x5[2]) if x5[2] < 0 else 0.0
            else:
                be[0] = np.sqrt(x5[0]); be[1] = np.sqrt(x5[2]) if x5[2] > 0 else 0.0
            if x5[1] < 0:
                be[0] = -be[0]
            with np.errstate(divide="ignore", invalid="ignore"):
                be[2] = x5[3] / be[0]
        if not all(np.isfinite(b) for b in be):
            continue
        gn_ok = True
        for _ in range(5):
            A = np.zeros((6, 4)); r6 = np.zeros(6)
            for p in range(6):
                l = L[p]
                A[p, 0] = 2 * l[0] * be[0] + l[1] * be[1] + l[3] * be[2] + l[6] * be[3]
                A[p, 1] = l[1] * be[0] + 2 * l[2] * be[1] + l[4] * be[2] + l[7] * be[3]
                A[p, 2] = l[3] * be[0] + l[4] * be[1] + 2 * l[5] * be[2] + l[8] * be[3]
                A[p, 3] = l[6] * be[0] + l[7] * be[1] + l[8] * be[2] + 2 * l[9] * be[3]
                r6[p] = rho[p] - (l[0] * be[0] * be[0] + l[1] * be[0] * be[1] + l[2] * be[1] * be[1] + l[3] * be[0] * be[2] + l[4] * be[1] * be[2] +
                                  l[5] * be[2] * be[2] + l[6] * be[0] * be[3] + l[7] * be[1] * be[3] + l[8] * be[2] * be[3] + l[9] * be[3] * be[3])
            dx = _lsq_small(A, r6)
            if dx is None:
                gn_ok = False
                break
            for k in range(4):
                be[k] += dx[k]
        if not gn_ok or not all(np.isfinite(b) for b in be):
            continue
        ccs = np.zeros((4, 3))
        for j in range(4):
            for a in range(3):
                ccs[j, a] = be[0] * v[0][3 * j + a] + be[1] * v[1][3 * j + a] + be[2] * v[2][3 * j + a] + be[3] * v[3][3 * j + a]
        pc = np.zeros((n, 3))
        for i in range(n):
            for a in range(3):
                pc[i, a] = al[i, 0] * ccs[0, a] + al[i, 1] * ccs[1, a] + al[i, 2] * ccs[2, a] + al[i, 3] * ccs[3, a]
        if pc[0, 2] < 0:
            pc = -pc
        h = _horn_jacobi(pc, pw)
        if h is None:
            continue
        Rg, tg = h
        err, finite = 0.0, True
        for i in range(n):
            X = pw[i]
            xc = Rg[0, 0] * X[0] + Rg[0, 1] * X[1] + Rg[0, 2] * X[2] + tg[0]; yc = Rg[1, 0] * X[0] + Rg[1, 1] * X[1] + Rg[1, 2] * X[2] + tg[1]
            zc = Rg[2, 0] * X[0] + Rg[2, 1] * X[1] + Rg[2, 2] * X[2] + tg[2]
            with np.errstate(divide="ignore", invalid="ignore"):
                du = uc + fu * xc / zc - uv[i, 0]; dv2 = vc + fv * yc / zc - uv[i, 1]
            e = np.sqrt(du * du + dv2 * dv2)
            if not np.isfinite(e):
                finite = False
                break
            err += e
        if not finite:
            continue
        err /= n
        if err < best_err:
            best_err, best = err, (Rg.copy(), tg.copy())
    return best


def _rot_to_quat(R):
    m = np.asarray(R, float).reshape(9)
    tr = m[0] + m[4] + m[8]
    if tr > 0:
        s4 = np.sqrt(tr + 1.0) * 2; return [0.25 * s4, (m[7] - m[5]) / s4, (m[2] - m[6]) / s4, (m[3] - m[1]) / s4]
    if m[0] > m[4] and m[0] > m[8]:
        s4 = np.sqrt(1.0 + m[0] - m[4] - m[8]) * 2; return [(m[7] - m[5]) / s4, 0.25 * s4, (m[1] + m[3]) / s4, (m[2] + m[6]) / s4]
    if m[4] > m[8]:
        s4 = np.sqrt(1.0 + m[4] - m[0] - m[8]) * 2; return [(m[2] - m[6]) / s4, (m[1] + m[3]) / s4, 0.25 * s4, (m[5] + m[7]) / s4]
    s4 = np.sqrt(1.0 + m[8] - m[0] - m[4]) * 2; return [(m[3] - m[1]) / s4, (m[2] + m[6]) / s4, (m[5] + m[7]) / s4, 0.25 * s4]


def pnp_solve_ransac(pw, obs, inv_sigma2, cam, iterations=100, seed=0x9E3779B9):
    """returns (n_inliers, pose7 (qw qx qy qz tx ty tz, world -> camera) or None, inlier flags)"""
    pw = np.asarray(pw, float); obs = np.asarray(obs, float); inv_sigma2 = np.asarray(inv_sigma2, float)
    n = len(pw)
    best, best_rt, best_inl = 0, None, np.zeros(n, bool)
    if n < 4:
        return 0, None, best_inl
    rng = _Rng(seed)

    def count_inliers(R, t):
        xc = R[0, 0] * pw[:, 0] + R[0, 1] * pw[:, 1] + R[0, 2] * pw[:, 2] + t[0]; yc = R[1, 0] * pw[:, 0] + R[1, 1] * pw[:, 1] + R[1, 2] * pw[:, 2] + t[1]
        z = R[2, 0] * pw[:, 0] + R[2, 1] * pw[:, 1] + R[2, 2] * pw[:, 2] + t[2]
        with np.errstate(divide="ignore", invalid="ignore"):
            du = cam[0] * xc / z + cam[2] - obs[:, 0]; dv = cam[1] * yc / z + cam[3] - obs[:, 1]
            inl = (z > 0) & ((du * du + dv * dv) * inv_sigma2 < 5.991)
        return int(inl.sum()), inl
    for _ in range(iterations):
        avail = list(range(n)); left = n; idx = []
        for _k in range(4):
            r = rng.next() % left
            idx.append(avail[r]); avail[r] = avail[left - 1]; left -= 1
        rt = epnp_solve(pw[idx], obs[idx], cam)
        if rt is None:
            continue
        c, inl = count_inliers(rt[0], rt[1])
        if c > best:
            best, best_rt, best_inl = c, rt, inl.copy()
    if best < 4:
        return 0, None, np.zeros(n, bool)
    rt = epnp_solve(pw[best_inl], obs[best_inl], cam)       # refit on the inliers of the best sample
    if rt is not None:
        c, inl = count_inliers(rt[0], rt[1])
        if c >= best:
            best, best_rt, best_inl = c, rt, inl.copy()
    return best, np.array(_rot_to_quat(best_rt[0]) + list(best_rt[1]), float), best_inl
