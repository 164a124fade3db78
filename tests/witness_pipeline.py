"""Independent witness of the WHOLE hot path (test infrastructure; numpy / scipy / Python floats only).

A second restatement of the reference's per-frame path, written from SURVEY.md Appendix A-C and the
reference sources, NOT from oracle/mpe_oracle.cpp and not sharing a line with it: other data structures
(numpy arrays, Python complex), other algorithms wherever the result does not depend on them —
  * contours: connected components by scipy.ndimage.label + Moore-neighbour boundary tracing per component
    and a flood fill of the outer background for RETR_EXTERNAL (the oracle and the HIP kernel follow
    OpenCV's Suzuki-Abe raster scan with border marks);
  * 4x4 inverse by numpy.linalg.inv, Kabsch by numpy.linalg.svd, Gauss-Newton step by numpy.linalg.solve,
    covariance by numpy.linalg.inv (the oracle has hand-written adjugate / Hestenes-Jacobi / LDL^T / Gauss-Jordan).
Its outputs are committed as tests/golden/witness_*.npz (tests/golden/make_witness_golden.py); the CPU oracle
(tests/test_golden_cpu.py) and the HIP path (tests/test_gpu_parity.py, -m gpu) are both checked against them.
It cannot make parity "pinned" (the reference holds no vectors), but a misreading shared by oracle and kernels
would have to be made a third time, independently, to go unnoticed.

Reference: lib = monocular_pose_estimator_lib/src.
"""
import math

import numpy as np

# ------------------------------------------------------------------------------------------------
# Combinations (lib/combinations.cpp:52-244)
# ------------------------------------------------------------------------------------------------


def combinations3(n):
    """Lexicographic 3-subsets of 1..n, 1-based rows (combinationsNoReplacement)."""
    return [(a, b, c) for a in range(1, n + 1) for b in range(a + 1, n + 1) for c in range(b + 1, n + 1)]


def permutations3(n):
    """permutationsNoReplacement(n, 3): for every lexicographic combination (a<b<c) the block
    [c b a], [c a b], [b c a], [b a c], [a b c], [a c b]."""
    out = []
    for (a, b, c) in combinations3(n):
        out += [(c, b, a), (c, a, b), (b, c, a), (b, a, c), (a, b, c), (a, c, b)]
    return out


def num_combinations_u32(n, k):
    """numCombinations with the 32-bit factorial of combinations.cpp:34-45."""
    def fact(m):
        r = 1
        for i in range(2, m + 1):
            r = (r * i) & 0xFFFFFFFF
        return r
    den = (fact(k) * fact(n - k)) & 0xFFFFFFFF
    return fact(n) // den if den else 0


# ------------------------------------------------------------------------------------------------
# P3P (lib/p3p.cpp)
# ------------------------------------------------------------------------------------------------


def _cpow(z, y):
    """std::pow(std::complex<double>, double) of libstdc++: real pow for a positive real base, else
    polar(exp(y log|z|), y arg z)."""
    if z.imag == 0.0 and z.real > 0.0:
        return complex(math.pow(z.real, y), 0.0)
    if z == 0:
        return complex(0.0, 0.0)
    rho = math.exp(y * math.log(abs(z)))
    th = y * math.atan2(z.imag, z.real)
    return complex(rho * math.cos(th), rho * math.sin(th))


def _csqrt(z):
    import cmath
    return cmath.sqrt(z)


def solve_quartic(f):
    """P3P::solveQuartic, p3p.cpp:238-286: Ferrari in complex double, REAL PARTS of the four roots."""
    A, B, C, D, E = [float(x) for x in f]
    A2, B2 = A * A, B * B
    A3, B3 = A2 * A, B2 * B
    A4, B4 = A3 * A, B3 * B
    alpha = -3 * B2 / (8 * A2) + C / A
    beta = B3 / (8 * A3) - B * C / (2 * A2) + D / A
    gamma = -3 * B4 / (256 * A4) + B2 * C / (16 * A3) - B * D / (4 * A2) + E / A
    alpha2 = alpha * alpha
    alpha3 = alpha2 * alpha
    P = complex(-alpha2 / 12 - gamma, 0.0)
    Q = complex(-alpha3 / 108 + alpha * gamma / 3 - beta ** 2 / 8, 0.0)
    R = -Q / 2.0 + _csqrt(_cpow(Q, 2.0) / 4.0 + _cpow(P, 3.0) / 27.0)
    U = _cpow(R, 1.0 / 3.0)
    if U.real == 0:
        y = -5.0 * alpha / 6.0 - _cpow(Q, 1.0 / 3.0)
    else:
        y = -5.0 * alpha / 6.0 - P / (3.0 * U) + U
    w = _csqrt(alpha + 2.0 * y)
    off = -B / (4.0 * A)
    with np.errstate(all="ignore"):
        try:
            bw = 2.0 * beta / w
        except ZeroDivisionError:
            bw = complex(float("nan"), float("nan"))
    r0 = off + 0.5 * (w + _csqrt(-(3.0 * alpha + 2.0 * y + bw)))
    r1 = off + 0.5 * (w - _csqrt(-(3.0 * alpha + 2.0 * y + bw)))
    r2 = off + 0.5 * (-w + _csqrt(-(3.0 * alpha + 2.0 * y - bw)))
    r3 = off + 0.5 * (-w - _csqrt(-(3.0 * alpha + 2.0 * y - bw)))
    return [r0.real, r1.real, r2.real, r3.real]


def _sqrt(x):
    return math.sqrt(x) if x >= 0 else float("nan")


def p3p(fv, wp):
    """P3P::computePoses, p3p.cpp:65-236.  fv, wp: 3x3 with the three bearings / world points as ROWS here.
    Returns None for collinear world points, else four (R 3x3, C 3) pairs (entries may be NaN)."""
    fv = np.asarray(fv, float)
    wp = np.asarray(wp, float)
    P1, P2, P3 = wp[0].copy(), wp[1].copy(), wp[2].copy()
    if np.linalg.norm(np.cross(P2 - P1, P3 - P1)) == 0:
        return None
    f1, f2, f3 = fv[0].copy(), fv[1].copy(), fv[2].copy()

    def tau(f1, f2):
        e1 = f1
        e3 = np.cross(f1, f2)
        e3 = e3 / np.linalg.norm(e3)
        e2 = np.cross(e3, e1)
        return np.vstack([e1, e2, e3])

    T = tau(f1, f2)
    f3t = T @ f3
    if f3t[2] > 0:
        f1, f2 = fv[1].copy(), fv[0].copy()
        T = tau(f1, f2)
        f3t = T @ fv[2]
        P1, P2, P3 = wp[1].copy(), wp[0].copy(), wp[2].copy()
    n1 = P2 - P1
    n1 = n1 / np.linalg.norm(n1)
    n3 = np.cross(n1, P3 - P1)
    n3 = n3 / np.linalg.norm(n3)
    n2 = np.cross(n3, n1)
    N = np.vstack([n1, n2, n3])
    P3n = N @ (P3 - P1)
    d_12 = float(np.linalg.norm(P2 - P1))
    with np.errstate(all="ignore"):
        f_1 = np.float64(f3t[0]) / np.float64(f3t[2])
        f_2 = np.float64(f3t[1]) / np.float64(f3t[2])
        p_1, p_2 = np.float64(P3n[0]), np.float64(P3n[1])
        cos_beta = np.float64(np.dot(f1, f2))
        b = np.float64(1) / (1 - cos_beta ** 2) - 1
        b = -np.sqrt(b) if cos_beta < 0 else np.sqrt(b)
    f_1, f_2, p_1, p_2, b = float(f_1), float(f_2), float(p_1), float(p_2), float(b)
    f_1_pw2, f_2_pw2 = f_1 ** 2, f_2 ** 2
    p_1_pw2 = p_1 ** 2
    p_1_pw3 = p_1_pw2 * p_1
    p_1_pw4 = p_1_pw3 * p_1
    p_2_pw2 = p_2 ** 2
    p_2_pw3 = p_2_pw2 * p_2
    p_2_pw4 = p_2_pw3 * p_2
    d_12_pw2 = d_12 ** 2
    b_pw2 = b ** 2
    F = [0.0] * 5
    F[0] = -f_2_pw2 * p_2_pw4 - p_2_pw4 * f_1_pw2 - p_2_pw4
    F[1] = 2 * p_2_pw3 * d_12 * b + 2 * f_2_pw2 * p_2_pw3 * d_12 * b - 2 * f_2 * p_2_pw3 * f_1 * d_12
    F[2] = (-f_2_pw2 * p_2_pw2 * p_1_pw2 - f_2_pw2 * p_2_pw2 * d_12_pw2 * b_pw2 - f_2_pw2 * p_2_pw2 * d_12_pw2
            + f_2_pw2 * p_2_pw4 + p_2_pw4 * f_1_pw2 + 2 * p_1 * p_2_pw2 * d_12 + 2 * f_1 * f_2 * p_1 * p_2_pw2 * d_12 * b
            - p_2_pw2 * p_1_pw2 * f_1_pw2 + 2 * p_1 * p_2_pw2 * f_2_pw2 * d_12 - p_2_pw2 * d_12_pw2 * b_pw2
            - 2 * p_1_pw2 * p_2_pw2)
    F[3] = (2 * p_1_pw2 * p_2 * d_12 * b + 2 * f_2 * p_2_pw3 * f_1 * d_12 - 2 * f_2_pw2 * p_2_pw3 * d_12 * b
            - 2 * p_1 * p_2 * d_12_pw2 * b)
    F[4] = (-2 * f_2 * p_2_pw2 * f_1 * p_1 * d_12 * b + f_2_pw2 * p_2_pw2 * d_12_pw2 + 2 * p_1_pw3 * d_12
            - p_1_pw2 * d_12_pw2 + f_2_pw2 * p_2_pw2 * p_1_pw2 - p_1_pw4 - 2 * f_2_pw2 * p_2_pw2 * p_1 * d_12
            + p_2_pw2 * f_1_pw2 * p_1_pw2 + f_2_pw2 * p_2_pw2 * d_12_pw2 * b_pw2)
    roots = solve_quartic(F)
    sols = []
    for rt in roots:
        with np.errstate(all="ignore"):
            rt = np.float64(rt)
            f1_, f2_ = np.float64(f_1), np.float64(f_2)
            cot_alpha = (-f1_ * p_1 / f2_ - rt * p_2 + d_12 * b) / (-f1_ * rt * p_2 / f2_ + p_1 - d_12)
            cos_theta = rt
            sin_theta = np.sqrt(1 - rt ** 2)
            sin_alpha = np.sqrt(1 / (cot_alpha ** 2 + 1))
            cos_alpha = np.sqrt(1 - sin_alpha ** 2)
            if cot_alpha < 0:
                cos_alpha = -cos_alpha
            k = sin_alpha * b + cos_alpha
            C = np.array([d_12 * cos_alpha * k, cos_theta * d_12 * sin_alpha * k, sin_theta * d_12 * sin_alpha * k])
            C = P1 + N.T @ C
            R = np.array([[-cos_alpha, -sin_alpha * cos_theta, -sin_alpha * sin_theta],
                          [sin_alpha, -cos_alpha * cos_theta, -cos_alpha * sin_theta],
                          [0.0, -sin_theta, cos_theta]])
            R = N.T @ R.T @ T
        sols.append((R, C))
    return sols


# ------------------------------------------------------------------------------------------------
# PoseEstimator (lib/pose_estimator.cpp)
# ------------------------------------------------------------------------------------------------


class Estimator:
    """The uninitialised branch of estimateBodyPose on detections: setImagePoints + initialise +
    checkCorrespondences + optimisePose (pose_estimator.cpp:80-91, 288-301, 544-721, 394-542, 733-792)."""

    def __init__(self, markers, K, back_tol=5.0, certainty_thr=0.75, valid_thr=0.7, hist_thr=0):
        self.M = np.asarray(markers, float).reshape(-1, 3)
        self.K = np.asarray(K, float).reshape(3, 3)
        self.back_tol, self.certainty_thr, self.valid_thr = back_tol, certainty_thr, valid_thr
        self.hist_thr = hist_thr if hist_thr else num_combinations_u32(len(self.M), 3)
        self.P = np.hstack([self.K, np.zeros((3, 1))])  # K [I | 0]

    def image_vectors(self, det):
        K = self.K
        v = np.stack([(det[:, 0] - K[0, 2]) / K[0, 0], (det[:, 1] - K[1, 2]) / K[1, 1], np.ones(len(det))], 1)
        return v / np.linalg.norm(v, axis=1, keepdims=True)

    def project2d(self, m, T):
        """pose_estimator.cpp:251-268: K [I|0] T (m;1), divided by z."""
        t = self.P @ T @ np.append(m, 1.0)
        return t[:2] / t[2]

    @staticmethod
    def H_of(R, C):
        H = np.eye(4)
        H[:3, :3] = R
        H[:3, 3] = C
        return H

    def vote(self, det):
        """initialise(), voting part: pose_estimator.cpp:544-702."""
        det = np.asarray(det, float).reshape(-1, 2)
        n_d, n_m = len(det), len(self.M)
        iv = self.image_vectors(det)
        hist = np.zeros((n_d, n_m), np.uint32)
        perms = permutations3(n_m)
        for c in combinations3(n_d):
            ci = [x - 1 for x in c]
            Ui = [a for a in range(n_d) if a not in ci]
            for p in perms:
                pi = [x - 1 for x in p]
                sols = p3p(iv[ci], self.M[pi])
                if sols is None:
                    continue
                Uo = [j for j in range(n_m) if j not in pi]
                for (R, C) in sols:
                    H = self.H_of(R, C)
                    if not np.all(np.isfinite(H)):
                        continue
                    with np.errstate(all="ignore"):
                        Hi = np.linalg.inv(H)
                        q = np.array([self.project2d(self.M[j], Hi) for j in Uo])
                        dist = np.sqrt(((det[Ui][:, None, :] - q[None, :, :]) ** 2).sum(-1))  # (unused det) x (unused markers)
                    nn = np.argmin(dist, axis=1)  # first minimum
                    dmin = dist[np.arange(len(Ui)), nn]
                    ok = dmin < self.back_tol  # strict <
                    if ok.any():
                        for k in range(3):
                            hist[ci[k], pi[k]] += 1
                        for a in np.nonzero(ok)[0]:
                            hist[Ui[a], Uo[nn[a]]] += 1
        return hist

    def correspondences_from_histogram(self, hist):
        """pose_estimator.cpp:344-370: arg-max peeling, column-major first maximum, column zeroed."""
        h = hist.astype(np.int64).copy()
        rows = []
        for _ in range(h.shape[1]):
            flat = int(np.argmax(h.T.reshape(-1)))  # column-major scan, first maximum
            c, r = divmod(flat, h.shape[0])
            if h[r, c] < self.hist_thr:
                break
            rows.append((c + 1, r + 1))  # (marker, detection), 1-based
            h[:, c] = 0
        return np.array(rows, np.uint32).reshape(-1, 2)

    def greedy_error(self, im, bp):
        """calculateSquaredReprojectionErrorAndCertainty, pose_estimator.cpp:303-342."""
        with np.errstate(all="ignore"):
            D = np.sqrt(((im[:, None, :] - bp[None, :, :]) ** 2).sum(-1))
        sq, cnt = 0.0, 0
        for _ in range(min(D.shape)):
            flat = int(np.argmin(D.T.reshape(-1)))
            c, r = divmod(flat, D.shape[0])
            if D[r, c] <= self.back_tol:
                sq += D[r, c] ** 2
                cnt += 1
                D[r, :] = np.inf
                D[:, c] = np.inf
            else:
                break
        return sq, cnt / D.shape[1]

    def check_correspondences(self, det, corr):
        """pose_estimator.cpp:394-542 -> (valid, T 4x4 or None)."""
        det = np.asarray(det, float).reshape(-1, 2)
        n_c = len(corr)
        if n_c < 4:
            return False, None
        iv = self.image_vectors(det)
        n_m = len(self.M)
        mean = np.zeros((n_m, 4))
        Mh = np.hstack([self.M, np.ones((n_m, 1))])
        combos = combinations3(n_c)
        n_valid = 0
        for c in combos:
            rows = [x - 1 for x in c]
            wp = self.M[[int(corr[r, 0]) - 1 for r in rows]]
            fv = iv[[int(corr[r, 1]) - 1 for r in rows]]
            others = [l for l in range(n_c) if l not in rows]
            un_obj = self.M[[int(corr[l, 0]) - 1 for l in others]]
            un_im = det[[int(corr[l, 1]) - 1 for l in others]]
            sols = p3p(fv, wp)
            if sols is None:
                continue
            best, best_sq, found = None, np.inf, False
            for (R, C) in sols:
                H = self.H_of(R, C)
                if not np.all(np.isfinite(H)):
                    continue
                Hi = np.linalg.inv(H)
                bp = np.array([self.project2d(m, Hi) for m in un_obj])
                sq, certainty = self.greedy_error(un_im, bp)
                if certainty >= self.certainty_thr:
                    found = True
                    if sq < best_sq:
                        best_sq, best = sq, Hi
            if found:
                n_valid += 1
                mean += (best @ Mh.T).T
        if n_valid / len(combos) >= self.valid_thr:
            mean = mean / n_valid
            return True, compute_transformation(self.M, mean[:, :3])
        return False, None

    def optimise_pose(self, det, corr, T0):
        """optimisePose, pose_estimator.cpp:733-792 (+ computeJacobian :932-960, exponentialMap :962-994)."""
        det = np.asarray(det, float).reshape(-1, 2)
        fx, fy = self.K[0, 0], self.K[1, 1]
        T = np.array(T0, float)
        A = np.zeros((6, 6))
        it_used = 0
        for it in range(500):
            A = np.zeros((6, 6))
            b = np.zeros(6)
            for (mi, di) in corr:
                if di == 0:
                    continue
                m = self.M[int(mi) - 1]
                e = det[int(di) - 1] - self.project2d(m, T)
                x, y, z = (T @ np.append(m, 1.0))[:3]
                z2 = z * z
                J = np.array([[fx / z, 0, -x * fx / z2, -x * y * fx / z2, (1 + x * x / z2) * fx, -y * fx / z],
                              [0, fy / z, -y * fy / z2, -(1 + y * y / z2) * fy, x * y * fy / z2, x * fy / z]])
                A += J.T @ J
                b += J.T @ e
            dT = np.linalg.solve(A, b)
            T = exponential_map(dT) @ T
            it_used = it + 1
            if np.abs(dT).max() <= 1e-13:
                break
        return T, np.linalg.inv(A), it_used

    def solve_bruteforce(self, det):
        det = np.asarray(det, float).reshape(-1, 2)
        out = dict(status=1, hist=np.zeros((len(det), len(self.M)), np.uint32), corr=np.zeros((0, 2), np.uint32),
                   T=np.eye(4), cov=np.zeros((6, 6)), gn_iterations=0)
        if len(det) < 4 or len(self.M) < 4:  # min_num_leds_detected_, pose_estimator.h:78
            return out
        out["hist"] = self.vote(det)
        if not out["hist"].any():
            return out
        out["corr"] = self.correspondences_from_histogram(out["hist"])
        ok, T0 = self.check_correspondences(det, out["corr"])
        if not ok:
            return out
        T, cov, it = self.optimise_pose(det, out["corr"], T0)
        out.update(status=0, T=T, cov=cov, gn_iterations=it)
        return out


def compute_transformation(obj, rep):
    """computeTransformation, pose_estimator.cpp:908-930 (Kabsch, no reflection guard)."""
    mo, mr = obj.mean(0), rep.mean(0)
    H = (obj - mo).T @ (rep - mr)
    U, _, Vt = np.linalg.svd(H)
    R = Vt.T @ U.T
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = mr - R @ mo
    return T


def exponential_map(tw):
    """pose_estimator.cpp:962-994: twist (upsilon, omega) -> 4x4."""
    u, w = np.asarray(tw[:3], float), np.asarray(tw[3:], float)
    th = float(np.linalg.norm(w))
    O = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th == 0:
        R, V = np.eye(3), np.eye(3)
    else:
        R = np.eye(3) + O / th * math.sin(th) + O @ O / th ** 2 * (1 - math.cos(th))
        V = np.eye(3) + (1 - math.cos(th)) / th ** 2 * O + (th - math.sin(th)) / th ** 3 * (O @ O)
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = V @ u
    return T


# ------------------------------------------------------------------------------------------------
# LEDDetector::findLeds (lib/led_detector.cpp:35-112, OpenCV semantics of SURVEY A.1)
# ------------------------------------------------------------------------------------------------

# clockwise on the screen (y grows downwards): E, SE, S, SW, W, NW, N, NE
_DIRS = [(1, 0), (1, 1), (0, 1), (-1, 1), (-1, 0), (-1, -1), (0, -1), (1, -1)]


def gaussian_taps_q8(sigma):
    n = int(np.rint(sigma * 6 + 1)) | 1
    x = np.arange(n) - (n - 1) * 0.5
    k = np.exp(-0.5 * x * x / (sigma * sigma)).astype(np.float32)
    k = (k * np.float32(1.0 / k.astype(np.float64).sum())).astype(np.float32)
    return np.rint(k.astype(np.float64) * 256.0).astype(np.int64)


def blurred_mask(img, thr, sigma):
    """THRESH_TOZERO (strict >) + GaussianBlur(ksize 0, sigma) in 8-bit fixed point, BORDER_REFLECT_101 -> non-zero mask."""
    t = np.where(img.astype(np.int64) > thr, img.astype(np.int64), 0)
    k = gaussian_taps_q8(sigma)
    r = len(k) // 2
    p = np.pad(t, r, mode="reflect") if r else t
    h = sum(int(k[j]) * p[:, j:j + t.shape[1]] for j in range(len(k)))
    v = sum(int(k[i]) * h[i:i + t.shape[0], :] for i in range(len(k)))
    return ((v + (1 << 15)) >> 16) != 0


def moore_boundary(comp):
    """Closed 8-connected outer boundary walk of one connected component (bool array), every visit emitted
    (1-pixel-wide parts twice), starting at the top-most, then left-most pixel.  Moore-neighbour tracing with
    Jacob's stopping criterion."""
    H, W = comp.shape
    ys, xs = np.nonzero(comp)
    i0 = np.lexsort((xs, ys))[0]
    sx, sy = int(xs[i0]), int(ys[i0])

    def fg(x, y):
        return 0 <= x < W and 0 <= y < H and comp[y, x]

    # first move: search clockwise starting after the west neighbour (known background)
    def next_from(x, y, back_dir):
        for k in range(1, 9):
            d = (back_dir + k) % 8
            nx, ny = x + _DIRS[d][0], y + _DIRS[d][1]
            if fg(nx, ny):
                return d
        return None

    d0 = next_from(sx, sy, 4)  # backtrack = W (background: the start pixel is the left-most of the top row)
    if d0 is None:
        return [(sx, sy)]
    pts = []
    x, y, d = sx, sy, d0
    while True:
        pts.append((x, y))
        x, y = x + _DIRS[d][0], y + _DIRS[d][1]
        nd = next_from(x, y, (d + 4) % 8)  # sweep clockwise, starting just after the pixel we came from
        if (x, y) == (sx, sy) and nd == d0:  # back at the start AND about to repeat the first move: closed
            break
        d = nd
    return pts


def find_leds(img, thr, sigma, min_area, max_area, max_wh, max_circ, K, D, roi=None):
    """-> (undistorted (n,2) float64 of float32 values, distorted (n,2) float32), reference order."""
    from scipy import ndimage
    img = np.asarray(img, np.uint8)
    rx, ry = 0, 0
    if roi is not None:
        rx, ry, rw, rh = roi
        img = img[ry:ry + rh, rx:rx + rw]
    mask = blurred_mask(img, thr, sigma)
    lab, n = ndimage.label(mask, structure=np.ones((3, 3), int))
    # RETR_EXTERNAL: keep components adjacent (4-neighbourhood) to the outer background (4-connected, frame zero-padded)
    bg, _ = ndimage.label(np.pad(~mask, 1, constant_values=True), structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])
    outer = (bg == bg[0, 0])
    touch = np.zeros_like(mask)
    o = outer
    touch |= o[1:-1, :-2] | o[1:-1, 2:] | o[:-2, 1:-1] | o[2:, 1:-1]
    blobs = []
    objs = ndimage.find_objects(lab)
    for li in range(1, n + 1):
        sl = objs[li - 1]
        comp = lab[sl] == li
        if not (touch[sl] & comp).any():
            continue
        y0, x0 = sl[0].start, sl[1].start
        pts = [(x + x0, y + y0) for (x, y) in moore_boundary(comp)]
        xs = np.array([p[0] for p in pts], float)
        ys = np.array([p[1] for p in pts], float)
        xp, yp = np.roll(xs, 1), np.roll(ys, 1)
        dxy = xp * ys - xs * yp
        a00, a10, a01 = dxy.sum(), (dxy * (xp + xs)).sum(), (dxy * (yp + ys)).sum()
        area = abs(a00) / 2
        w = int(xs.max() - xs.min() + 1)
        h = int(ys.max() - ys.min() + 1)
        if abs(a00) > 1.1920928955078125e-07:
            sgn = 1.0 if a00 > 0 else -1.0
            m00, m10, m01 = a00 * (sgn * 0.5), a10 * (sgn / 6.0), a01 * (sgn / 6.0)
        else:
            m00 = m10 = m01 = 0.0
        with np.errstate(all="ignore"):
            mcx = np.float32(np.float32(np.float64(m10) / np.float64(m00)) + np.float32(rx))
            mcy = np.float32(np.float32(np.float64(m01) / np.float64(m00)) + np.float32(ry))
            hw, hh = float(w // 2), float(h // 2)  # INTEGER halves
            ok = (min_area <= area <= max_area and abs(1 - min(w / h, h / w)) <= max_wh and
                  abs(1 - np.float64(area) / (math.pi * hw * hw)) <= max_circ and
                  abs(1 - np.float64(area) / (math.pi * hh * hh)) <= max_circ)
        if ok:
            first = min(pts, key=lambda p: (p[1], p[0]))
            blobs.append((first[1], first[0], mcx, mcy))
    blobs.sort(key=lambda b: (b[0], b[1]), reverse=True)  # newest contour first = reverse raster order of the start pixel
    dist = np.array([[b[2], b[3]] for b in blobs], np.float32).reshape(-1, 2)
    und = undistort_points(dist, K, D)
    return und.astype(np.float64), dist


def undistort_points(pts, K, D):
    """cv::undistortPoints(src, dst, K, D, noArray(), K): 5 fixed-point iterations, float32 out."""
    K = np.asarray(K, float).reshape(3, 3)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    d = list(np.asarray(D, float).reshape(-1)) + [0.0] * 8
    k1, k2, p1, p2, k3 = d[:5]
    out = np.zeros((len(pts), 2), np.float32)
    for i, (u, v) in enumerate(np.asarray(pts, np.float64)):
        x0 = x = (u - cx) * (1.0 / fx)  # OpenCV multiplies by ifx = 1 / fx
        y0 = y = (v - cy) * (1.0 / fy)
        for _ in range(5 if len(np.asarray(D).reshape(-1)) else 0):
            r2 = x * x + y * y
            icd = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2)
            dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
            dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
            x = (x0 - dx) * icd
            y = (y0 - dy) * icd
        out[i] = (np.float32(fx * x + cx), np.float32(fy * y + cy))
    return out


def estimate_frame(img, markers, K, D, params):
    """estimateBodyPose on a fresh estimator (pose_estimator.cpp:62-96) -> dict like the oracle's."""
    und, dist = find_leds(img, params["threshold_value"], params["gaussian_sigma"], params["min_blob_area"],
                          params["max_blob_area"], params["max_width_height_distortion"], params["max_circular_distortion"],
                          K, D)
    est = Estimator(markers, K, params["back_projection_pixel_tolerance"], params["certainty_threshold"],
                    params["valid_correspondence_threshold"], params.get("histogram_threshold", 0))
    r = est.solve_bruteforce(und)
    r.update(und=und, dist=dist)
    return r


# ------------------------------------------------------------------------------------------------
# The tracking path (round 4): estimateBodyPose as a STATE MACHINE over a sequence of frames —
# pose_estimator.cpp:62-147 (both branches, the whole-image retry), predictWithROI :814-829, predictPose :232-244,
# logarithmMap :996-1064, predictMarkerPositionsInImage :270-276, findCorrespondences :372-392 (+ calculateMinDistances-
# AndPairs :862-906), findCorrespondencesAndPredictPose :831-848, optimiseAndUpdatePose / updatePose :794-812;
# LEDDetector::determineROI / distortPoints, led_detector.cpp:114-224.  Written from those sources (not from the oracle's
# Tracker): numpy for the 4x4 algebra (numpy.linalg.inv where the reference calls .inverse()), Python floats elsewhere.
# ------------------------------------------------------------------------------------------------


def skew(w):
    return np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])


def _is_approx(a, b, prec):
    """Eigen's isApprox for matrices: ||a - b||_F^2 <= prec^2 * min(||a||_F^2, ||b||_F^2)."""
    return float(((a - b) ** 2).sum()) <= prec * prec * min(float((a ** 2).sum()), float((b ** 2).sum()))


def logarithm_map(T):
    """pose_estimator.cpp:996-1064 -> twist (upsilon, w)."""
    T = np.asarray(T, float)
    R, t = T[:3, :3], T[:3, 3]
    w_hat = np.zeros((3, 3))
    if not _is_approx(R, np.eye(3), 1e-10):
        temp = (np.trace(R) - 1) / 2
        temp = 1.0 if temp > 1 else (-1.0 if temp < -1 else temp)
        phi = math.acos(temp)
        if phi != 0:
            w_hat = (R - R.T) / (2 * math.sin(phi)) * phi
    w = np.array([w_hat[2, 1], w_hat[0, 2], w_hat[1, 0]])
    wn = float(np.sqrt((w ** 2).sum()))
    # t.isApproxToConstant(0, 1e-10): ||t - 0||^2 <= prec^2 * min(||t||^2, ||0||^2) = 0, i.e. only an exactly zero t
    if not np.any(t != 0):
        A_inv = np.zeros((3, 3))
    elif wn == 0 or math.sin(wn) == 0:
        A_inv = np.eye(3)
    else:
        A_inv = (np.eye(3) - w_hat / 2 +
                 (2 * math.sin(wn) - wn * (1 + math.cos(wn))) / (2 * wn * wn * math.sin(wn)) * (w_hat @ w_hat))
    return np.concatenate([A_inv @ t, w])


def distort_points(pts, K, D):
    """LEDDetector::distortPoints, led_detector.cpp:181-224: float32 points in, computed in double, float32 out."""
    K = np.asarray(K, float).reshape(3, 3)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    k1, k2, p1, p2, k3 = [float(v) for v in np.asarray(D, float).reshape(-1)[:5]]
    out = []
    for (px, py) in pts:
        px, py = float(np.float32(px)), float(np.float32(py))
        x, y = (px - cx) / fx, (py - cy) / fy
        r2 = x * x + y * y
        xc = x * (1. + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2)
        yc = y * (1. + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2)
        xc = xc + (2. * p1 * x * y + p2 * (r2 + 2. * x * x))
        yc = yc + (p1 * (r2 + 2. * y * y) + 2. * p2 * x * y)
        out.append((float(np.float32(xc * fx + cx)), float(np.float32(yc * fy + cy))))
    return out


def determine_roi(pred_px, rows, cols, border, K, D):
    """LEDDetector::determineROI, led_detector.cpp:114-179 -> (x, y, w, h).  Quirks kept: x_max / y_max start from 0,
    the corner points pass through cv::Point2f (float32) before and after the distortion, the rectangle's integer
    members truncate the doubles."""
    x_min = y_min = math.inf
    x_max = y_max = 0.0
    for (px, py) in np.asarray(pred_px, float).reshape(-1, 2):
        if px < x_min:
            x_min = px
        if px > x_max:
            x_max = px
        if py < y_min:
            y_min = py
        if py > y_max:
            y_max = py
    with np.errstate(all="ignore"):
        d = distort_points([(np.float32(x_min), np.float32(y_min)), (np.float32(x_max), np.float32(y_max))], K, D)
    (x_min_d, y_min_d), (x_max_d, y_max_d) = d
    x0 = max(0.0, min(float(cols), x_min_d - border))
    x1 = max(0.0, min(float(cols), x_max_d + border))
    y0 = max(0.0, min(float(rows), y_min_d - border))
    y1 = max(0.0, min(float(rows), y_max_d + border))
    if not (x1 - x0 >= 1 and y1 - y0 >= 1):  # (also taken when a coordinate is NaN: every comparison is false)
        return (0, 0, cols, rows)
    return (int(x0), int(y0), int(x1 - x0), int(y1 - y0))


class Tracker:
    """One stateful PoseEstimator object (pose_estimator.h:52-803) fed frame after frame."""

    def __init__(self, markers, K, D, params):
        self.p = dict(params)
        self.K = np.asarray(K, float).reshape(3, 3)
        self.D = np.asarray(D, float).reshape(-1)
        self.est = Estimator(markers, K, params["back_projection_pixel_tolerance"], params["certainty_threshold"],
                             params["valid_correspondence_threshold"], params.get("histogram_threshold", 0))
        self.n_m = len(self.est.M)
        self.it = 0                                   # it_since_initialized_
        # (the reference leaves poses / times uninitialised; nothing reads them before they are first written except
        #  previous_pose_ in the third frame's predictPose, which by then holds the first frame's current_pose_)
        self.prev_pose = np.eye(4)
        self.cur_pose = np.eye(4)
        self.pred_pose = np.eye(4)
        self.prev_time = self.cur_time = self.pred_time = 0.0
        self.pred_px = np.zeros((self.n_m, 2))
        self.roi = (0, 0, 0, 0)
        self.det = np.zeros((0, 2))
        self.corr = np.zeros((0, 2), np.uint32)
        self.cov = np.zeros((6, 6))
        self.used_bruteforce = False
        self.updated = False

    # -- pieces ------------------------------------------------------------------------------------
    def _find_leds(self, img, roi):
        p = self.p
        und, _ = find_leds(img, p["threshold_value"], p["gaussian_sigma"], p["min_blob_area"], p["max_blob_area"],
                           p["max_width_height_distortion"], p["max_circular_distortion"], self.K, self.D, roi=roi)
        return und

    def _initialise(self):
        """initialise(), pose_estimator.cpp:544-721: 1 and predicted_pose_ set on success."""
        self.used_bruteforce = True
        hist = self.est.vote(self.det)
        if not hist.any():
            return False
        self.corr = self.est.correspondences_from_histogram(hist)
        ok, T0 = self.est.check_correspondences(self.det, self.corr)
        if ok:
            self.pred_pose = T0
        return ok

    def _optimise_and_update(self):
        T, cov, _ = self.est.optimise_pose(self.det, self.corr, self.pred_pose)
        self.pred_pose, self.cov = T, cov
        if self.it < 2:
            self.it += 1
        self.prev_pose, self.cur_pose = self.cur_pose, self.pred_pose
        self.prev_time, self.cur_time = self.cur_time, self.pred_time
        self.updated = True

    def _predict_with_roi(self, t, rows, cols):
        if self.it >= 2:
            self.pred_time = t
            delta = logarithm_map(np.linalg.inv(self.prev_pose) @ self.cur_pose)
            delta_hat = delta / (self.cur_time - self.prev_time) * (self.pred_time - self.cur_time)
            self.pred_pose = self.cur_pose @ exponential_map(delta_hat)
        else:
            self.pred_time = t
        self.pred_px = np.array([self.est.project2d(m, self.pred_pose) for m in self.est.M])
        self.roi = determine_roi(self.pred_px, rows, cols, int(self.p["roi_border_thickness"]), self.K, self.D)

    def _find_correspondences(self):
        """findCorrespondences: for every PREDICTED marker pixel its nearest detection (first minimum), kept when the
        distance is <= nearest_neighbour_pixel_tolerance_; rows (marker, detection), 1-based."""
        rows = []
        for i, pp in enumerate(self.pred_px):
            best, bj = math.inf, 0
            for j, d in enumerate(self.det):
                d2 = float((pp[0] - d[0]) ** 2 + (pp[1] - d[1]) ** 2)
                if d2 < best:
                    best, bj = d2, j
            if math.sqrt(best) <= self.p["nearest_neighbour_pixel_tolerance"]:
                rows.append((i + 1, bj + 1))
        self.corr = np.array(rows, np.uint32).reshape(-1, 2)

    # -- estimateBodyPose --------------------------------------------------------------------------
    def estimate(self, img, t):
        img = np.asarray(img, np.uint8)
        rows, cols = img.shape
        self.updated = False
        self.used_bruteforce = False
        if self.it < 1:
            self.pred_time = t
            self.roi = (0, 0, cols, rows)
            self.det = self._find_leds(img, self.roi)
            if len(self.det) >= 4:
                if self._initialise():
                    self._optimise_and_update()
        else:
            self._predict_with_roi(t, rows, cols)
            self.det = self._find_leds(img, self.roi)
            loops = 0
            while True:
                loops += 1
                if len(self.det) >= 4:
                    self._find_correspondences()
                    ok, T0 = self.est.check_correspondences(self.det, self.corr)
                    if ok:
                        self.pred_pose = T0
                        self._optimise_and_update()
                    elif self._initialise():
                        self._optimise_and_update()
                    break
                if loops < 2:
                    self.roi = (0, 0, cols, rows)
                    self.det = self._find_leds(img, self.roi)
                else:
                    break
        return dict(updated=self.updated, T=self.pred_pose.copy(), cov=self.cov.copy(), roi=tuple(self.roi),
                    it_since_initialized=self.it, n_det=len(self.det), n_corr=len(self.corr),
                    used_bruteforce=self.used_bruteforce)
