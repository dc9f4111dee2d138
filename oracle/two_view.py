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


# ---- [UPSTREAM] solve::pnp_solver, as host/two_view.cpp restates it: three-point solver (Grunert) + the fourth match of a sample + RANSAC
def _quartic_roots(c):
    """roots of c[0] + ... + c[4] x^4 by Durand-Kerner (80 sweeps from fixed starting points, Gauss-Seidel order)"""
    a3, a2, a1, a0 = c[3] / c[4], c[2] / c[4], c[1] / c[4], c[0] / c[4]
    rad = 1.0 + max(abs(a3), abs(a2), abs(a1), abs(a0))
    seed, w, r = complex(0.4, 0.9), complex(1.0, 0.0), []
    for _ in range(4):
        r.append(w * rad * 0.5); w *= seed
    for _ in range(80):
        for i in range(4):
            x = r[i]
            px = (((x + a3) * x + a2) * x + a1) * x + a0
            den = complex(1.0, 0.0)
            for k in range(4):
                if k != i:
                    den *= (x - r[k])
            if abs(den) > 0:
                r[i] = x - px / den
    return r


def p3p_grunert(pw, f):
    """up to four (R, t) world -> camera from three world points (rows of pw) and their unit bearings (rows of f)"""
    d2 = lambda i, k: float(((pw[i] - pw[k]) ** 2).sum())
    a2, b2, c2 = d2(1, 2), d2(0, 2), d2(0, 1)
    if not (a2 > 1e-12 and b2 > 1e-12 and c2 > 1e-12):
        return []
    ca, cb, cg = float(f[1] @ f[2]), float(f[0] @ f[2]), float(f[0] @ f[1])
    q1, kc = (a2 - c2) / b2, c2 / b2
    N = [1.0 + q1, -2.0 * q1 * cb, q1 - 1.0]; D = [2.0 * cg, -2.0 * ca]; K = [1.0, -2.0 * cb, 1.0]
    DD = [D[0] * D[0], 2.0 * D[0] * D[1], D[1] * D[1]]
    poly = [0.0] * 5
    for i in range(3):
        poly[i] += DD[i]
    for i in range(3):
        for k in range(3):
            poly[i + k] += N[i] * N[k]
    for i in range(3):
        for k in range(2):
            poly[i + k] -= 2.0 * cg * N[i] * D[k]
    for i in range(3):
        for k in range(3):
            poly[i + k] -= kc * K[i] * DD[k]
    big = max(abs(v) for v in poly)
    if not (abs(poly[4]) > 1e-12 * big):
        return []
    sols = []
    for root in _quartic_roots(poly):
        v = root.real
        if not (abs(root.imag) < 1e-6 * (1.0 + abs(v))) or not (v > 0):
            continue
        den = D[0] + D[1] * v
        if not (abs(den) > 1e-12):
            continue
        u = (N[0] + N[1] * v + N[2] * v * v) / den
        if not (u > 0):
            continue
        kk = 1.0 + v * v - 2.0 * v * cb
        if not (kk > 0):
            continue
        s1 = np.sqrt(b2 / kk)
        pc = np.array([s1 * f[0], u * s1 * f[1], v * s1 * f[2]])
        h = horn(pc, pw, True)
        if h is not None:
            sols.append((h[0], h[1]))
    return sols


def pnp_solve_ransac(pw, obs, inv_sigma2, cam, iterations=100, seed=0x9E3779B9):
    """returns (n_inliers, pose7 (qw qx qy qz tx ty tz, world -> camera) or None, inlier flags)"""
    pw = np.asarray(pw, float); obs = np.asarray(obs, float); inv_sigma2 = np.asarray(inv_sigma2, float)
    n = len(pw)
    best, best_pose, best_inl = 0, None, np.zeros(n, bool)
    if n < 4:
        return 0, None, best_inl
    x = (obs[:, 0] - cam[2]) / cam[0]; y = (obs[:, 1] - cam[3]) / cam[1]; nn = np.sqrt(x * x + y * y + 1.0)
    f = np.stack([x / nn, y / nn, 1.0 / nn], 1)
    rng = _Rng(seed)

    def reproj2(R, t, idx):
        pc = pw[idx] @ R.T + t
        with np.errstate(divide="ignore", invalid="ignore"):
            du = cam[0] * pc[..., 0] / pc[..., 2] + cam[2] - obs[idx, 0]; dv = cam[1] * pc[..., 1] / pc[..., 2] + cam[3] - obs[idx, 1]
        return du * du + dv * dv, pc[..., 2]
    every = np.arange(n)
    for _ in range(iterations):
        avail = list(range(n)); left = n; idx = []
        for _k in range(4):
            r = rng.next() % left
            idx.append(avail[r]); avail[r] = avail[left - 1]; left -= 1
        pick, pick_err = None, 0.0
        for R, t in p3p_grunert(pw[idx[:3]], f[idx[:3]]):
            e, z = reproj2(R, t, idx[3])
            if z > 0 and (pick is None or e < pick_err):
                pick, pick_err = (R, t), float(e)
        if pick is None:
            continue
        e, z = reproj2(pick[0], pick[1], every)
        inl = (z > 0) & (e * inv_sigma2 < 5.991)
        c = int(inl.sum())
        if c > best:
            m = pick[0].reshape(9)
            tr = m[0] + m[4] + m[8]
            if tr > 0:
                s4 = np.sqrt(tr + 1.0) * 2; q = [0.25 * s4, (m[7] - m[5]) / s4, (m[2] - m[6]) / s4, (m[3] - m[1]) / s4]
            elif m[0] > m[4] and m[0] > m[8]:
                s4 = np.sqrt(1.0 + m[0] - m[4] - m[8]) * 2; q = [(m[7] - m[5]) / s4, 0.25 * s4, (m[1] + m[3]) / s4, (m[2] + m[6]) / s4]
            elif m[4] > m[8]:
                s4 = np.sqrt(1.0 + m[4] - m[0] - m[8]) * 2; q = [(m[2] - m[6]) / s4, (m[1] + m[3]) / s4, 0.25 * s4, (m[5] + m[7]) / s4]
            else:
                s4 = np.sqrt(1.0 + m[8] - m[0] - m[4]) * 2; q = [(m[3] - m[1]) / s4, (m[2] + m[6]) / s4, (m[5] + m[7]) / s4, 0.25 * s4]
            best, best_pose, best_inl = c, np.array(q + list(pick[1]), float), inl.copy()
    return best, best_pose, best_inl
