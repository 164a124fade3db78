"""Forensics for HIP-vs-oracle mismatches (test infrastructure; numpy only, independent of the device code and of the
oracle's C++).

The only place where the HIP path is allowed to disagree with the CPU oracle is the unstable corner of the reference's
own quartic solver: P3P::solveQuartic (p3p.cpp:238-286) is Ferrari's method in plain double, and when
`alpha + 2y ~ 0` the term `2 beta / w` (p3p.cpp:272-277) amplifies every rounding of the preceding steps by ~1/|w|^2 —
two builds of the reference (other libm, other compiler) then produce different roots, hence different votes.  This
module decides, for a frame on which the two paths disagree, whether that is what happened:

  * `min_w`: the worst cancellation (ferrari_cancellation: |result| / sum |operands| of R = -Q/2 + sqrt(disc), of
    alpha + 2y, of the discriminant and of the two outer radicands) over ALL C(n_d,3) x P(n_m,3) hypotheses of the
    frame (coefficients as p3p.cpp:82-185, numpy complex arithmetic);
  * `oracle_flips_under_1ulp`: the ORACLE's own vote histogram changes when one detection coordinate moves by one ulp —
    the reference disagrees with itself on this frame.

A mismatch is `unstable` (explained) if either holds, `unexplained` otherwise; soaks and bench fail on unexplained ones.

`attribute_votes` goes one level deeper on a GPU box: it asks BOTH paths for every hypothesis' own votes
(mpe_vote_items / orc_vote_items), lists the hypotheses whose votes differ and requires of each one that it sits in
the corner (its own w below the threshold) or that the oracle's own P3P answer for it moves under a 1-ulp change of
an input — i.e. the difference of the two histograms is traced to the hypotheses that cast it.
"""
import itertools

import numpy as np

# cancellation below which a quantity of the solver carries no digits worth the name: its rounding error (~1e-16 of its
# operands) is then > 1e-4 of its value and the roots move in the 4th digit.  (An ordinary 5-LED frame has a handful of
# hypotheses at 1e-8 .. 1e-11 — garbage roots that vote for nothing in any build; the frames on which builds disagree
# have one at 1e-13 .. 1e-15.)
W_UNSTABLE = 1e-12


def bearings(det, K):
    """calculateImageVectors, pose_estimator.cpp:288-301."""
    det = np.asarray(det, float).reshape(-1, 2)
    K = np.asarray(K, float).reshape(3, 3)
    v = np.stack([(det[:, 0] - K[0, 2]) / K[0, 0], (det[:, 1] - K[1, 2]) / K[1, 1], np.ones(len(det))], 1)
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def permutations3(n):
    """Combinations::permutationsNoReplacement order (combinations.cpp:131-203): per lexicographic combination
    a < b < c the rows [c b a], [c a b], [b c a], [b a c], [a b c], [a c b] (0-based here)."""
    out = []
    for a, b, c in itertools.combinations(range(n), 3):
        out += [(c, b, a), (c, a, b), (b, c, a), (b, a, c), (a, b, c), (a, c, b)]
    return np.array(out, int)


def hypothesis_quartics(det, markers, K):
    """Quartic coefficients (p3p.cpp:82-185) of every hypothesis of a frame -> (n_triples * n_perms, 5), plus a mask of
    the hypotheses whose world points are not collinear."""
    iv = bearings(det, K)
    M = np.asarray(markers, float).reshape(-1, 3)
    tri = np.array(list(itertools.combinations(range(len(iv)), 3)), int)
    perm = permutations3(len(M))
    T_, P_ = np.meshgrid(np.arange(len(tri)), np.arange(len(perm)), indexing="ij")
    T_, P_ = T_.ravel(), P_.ravel()
    fa, fb, fc = iv[tri[T_, 0]], iv[tri[T_, 1]], iv[tri[T_, 2]]
    P1, P2, P3 = M[perm[P_, 0]], M[perm[P_, 1]], M[perm[P_, 2]]
    ok = np.linalg.norm(np.cross(P2 - P1, P3 - P1), axis=1) != 0

    def tau(f1, f2):
        e3 = np.cross(f1, f2)
        e3 /= np.linalg.norm(e3, axis=1, keepdims=True)
        return f1, np.cross(e3, f1), e3

    def apply(Trows, v):
        return np.stack([(r * v).sum(1) for r in Trows], 1)

    with np.errstate(all="ignore"):
        f3 = apply(tau(fa, fb), fc)
        sw = f3[:, 2] > 0  # p3p.cpp:100-121
        f1 = np.where(sw[:, None], fb, fa)
        f2 = np.where(sw[:, None], fa, fb)
        f3 = np.where(sw[:, None], apply(tau(fb, fa), fc), f3)
        P1s = np.where(sw[:, None], P2, P1)
        P2s = np.where(sw[:, None], P1, P2)
        n1 = P2s - P1s
        n1 /= np.linalg.norm(n1, axis=1, keepdims=True)
        n3 = np.cross(n1, P3 - P1s)
        n3 /= np.linalg.norm(n3, axis=1, keepdims=True)
        n2 = np.cross(n3, n1)
        P3n = apply((n1, n2, n3), P3 - P1s)
        d_12 = np.linalg.norm(P2s - P1s, axis=1)
        f_1, f_2 = f3[:, 0] / f3[:, 2], f3[:, 1] / f3[:, 2]
        p_1, p_2 = P3n[:, 0], P3n[:, 1]
        cb = (f1 * f2).sum(1)
        b = 1 / (1 - cb ** 2) - 1
        b = np.where(cb < 0, -np.sqrt(b), np.sqrt(b))
        f12, f22, p12, p22, d2, b2 = f_1 ** 2, f_2 ** 2, p_1 ** 2, p_2 ** 2, d_12 ** 2, b ** 2
        p13, p14, p23, p24 = p12 * p_1, p12 * p12, p22 * p_2, p22 * p22
        F = np.empty((len(T_), 5))
        F[:, 0] = -f22 * p24 - p24 * f12 - p24
        F[:, 1] = 2 * p23 * d_12 * b + 2 * f22 * p23 * d_12 * b - 2 * f_2 * p23 * f_1 * d_12
        F[:, 2] = (-f22 * p22 * p12 - f22 * p22 * d2 * b2 - f22 * p22 * d2 + f22 * p24 + p24 * f12 + 2 * p_1 * p22 * d_12 +
                   2 * f_1 * f_2 * p_1 * p22 * d_12 * b - p22 * p12 * f12 + 2 * p_1 * p22 * f22 * d_12 - p22 * d2 * b2 -
                   2 * p12 * p22)
        F[:, 3] = 2 * p12 * p_2 * d_12 * b + 2 * f_2 * p23 * f_1 * d_12 - 2 * f22 * p23 * d_12 * b - 2 * p_1 * p_2 * d2 * b
        F[:, 4] = (-2 * f_2 * p22 * f_1 * p_1 * d_12 * b + f22 * p22 * d2 + 2 * p13 * d_12 - p12 * d2 + f22 * p22 * p12 - p14 -
                   2 * f22 * p22 * p_1 * d_12 + p22 * f12 * p12 + f22 * p22 * d2 * b2)
    return F, ok


def ferrari_cancellation(F):
    """How much of its operands the WORST subtraction of Ferrari's method (p3p.cpp:253-283) leaves, per row of quartic
    coefficients: min over  R = -Q/2 + sqrt(disc)  (p3p.cpp:262: R ~ 0 when P ~ 0, then P / (3U) is 0/0-like),
    w^2 = alpha + 2y  (p3p.cpp:270),  the discriminant  Q^2/4 + P^3/27  and the two radicands
    -(3 alpha + 2y +- 2 beta / w)  (p3p.cpp:272-275),  each as |result| / sum |operands| in numpy complex arithmetic.
    1e-13 means: that quantity is known to 3 digits, and everything downstream of it inherits that."""
    F = np.asarray(F, float)
    with np.errstate(all="ignore"):
        A, B, C_, D_, E = (F[:, i] for i in range(5))
        alpha = -3 * B ** 2 / (8 * A ** 2) + C_ / A
        beta = B ** 3 / (8 * A ** 3) - B * C_ / (2 * A ** 2) + D_ / A
        gamma = -3 * B ** 4 / (256 * A ** 4) + B ** 2 * C_ / (16 * A ** 3) - B * D_ / (4 * A ** 2) + E / A
        P = (-alpha ** 2 / 12 - gamma).astype(complex)
        Q = (-alpha ** 3 / 108 + alpha * gamma / 3 - beta ** 2 / 8).astype(complex)
        disc = Q * Q / 4 + P * P * P / 27
        sq = np.sqrt(disc)
        R = -Q / 2 + sq
        U = R ** (1.0 / 3.0)
        y = -5 * alpha / 6 - np.where(U.real == 0, Q ** (1.0 / 3.0), P / (3 * U) - U)
        w2 = alpha + 2 * y
        bw = 2 * beta / np.sqrt(w2)
        tiny = 1e-300
        c = np.stack([
            np.abs(R) / (np.abs(Q / 2) + np.abs(sq) + tiny),
            np.abs(w2) / (np.abs(alpha) + 2 * np.abs(y) + tiny),
            np.abs(disc) / (np.abs(Q * Q / 4) + np.abs(P * P * P / 27) + tiny),
            np.abs(3 * alpha + 2 * y + bw) / (3 * np.abs(alpha) + 2 * np.abs(y) + np.abs(bw) + tiny),
            np.abs(3 * alpha + 2 * y - bw) / (3 * np.abs(alpha) + 2 * np.abs(y) + np.abs(bw) + tiny)])
        c = np.where(np.isfinite(c), c, np.inf)  # a NaN / inf on the way: NaN roots in every build, no votes anywhere
    return c.min(0)


def ferrari_w_batch(F):
    """(kept for callers that want the one indicator of tests/util.py::ferrari_w)  relative |alpha + 2y|."""
    F = np.asarray(F, float)
    with np.errstate(all="ignore"):
        A, B, C_, D_, E = (F[:, i] for i in range(5))
        alpha = -3 * B ** 2 / (8 * A ** 2) + C_ / A
        beta = B ** 3 / (8 * A ** 3) - B * C_ / (2 * A ** 2) + D_ / A
        gamma = -3 * B ** 4 / (256 * A ** 4) + B ** 2 * C_ / (16 * A ** 3) - B * D_ / (4 * A ** 2) + E / A
        P = (-alpha ** 2 / 12 - gamma).astype(complex)
        Q = (-alpha ** 3 / 108 + alpha * gamma / 3 - beta ** 2 / 8).astype(complex)
        R = -Q / 2 + np.sqrt(Q * Q / 4 + P * P * P / 27)
        U = R ** (1.0 / 3.0)
        y = -5 * alpha / 6 - np.where(U.real == 0, Q ** (1.0 / 3.0), P / (3 * U) - U)
        w = np.abs(alpha + 2 * y) / (np.abs(alpha) + 2 * np.abs(y) + 1e-300)
    return np.where(np.isfinite(w), w, np.inf)


def oracle_flips_under_1ulp(det, markers, K, tol, orc):
    """Does the oracle's OWN histogram change when one detection coordinate moves by one ulp?"""
    det = np.asarray(det, float).reshape(-1, 2)
    base = orc.vote_histogram(det, markers, K, tol)
    for i in range(len(det)):
        for c in range(2):
            for towards in (-np.inf, np.inf):
                d2 = det.copy()
                d2[i, c] = np.nextafter(d2[i, c], towards)
                if not np.array_equal(orc.vote_histogram(d2, markers, K, tol), base):
                    return True
    return False


def classify_frame(det, markers, K, tol, orc=None):
    """-> dict(min_w, hypotheses_below_threshold, oracle_flips_under_1ulp, unstable)."""
    det = np.asarray(det, float).reshape(-1, 2)
    if len(det) < 4:
        return {"min_w": None, "hypotheses_below_threshold": 0, "oracle_flips_under_1ulp": None, "unstable": False}
    F, ok = hypothesis_quartics(det, markers, K)
    w = ferrari_cancellation(F[ok])
    min_w = float(w.min()) if len(w) else float("inf")
    flips = None
    if orc is not None:
        flips = bool(oracle_flips_under_1ulp(det, markers, K, tol, orc))
    return {"min_w": min_w, "hypotheses_below_threshold": int((w < W_UNSTABLE).sum()),
            "oracle_flips_under_1ulp": flips, "unstable": bool(min_w < W_UNSTABLE or flips)}


def n_hypotheses(n_det, n_markers):
    return n_det * (n_det - 1) * (n_det - 2) // 6 * n_markers * (n_markers - 1) * (n_markers - 2)


def _oracle_p3p_moves(det, markers, K, hyp, orc):
    """Does the oracle's own computePoses answer for hypothesis `hyp` move (> 1e-6, or change its NaN pattern) when one
    bearing component changes by one ulp?  (The criterion tests/util.py::check_p3p_solutions applies.)"""
    iv = bearings(det, K)
    M = np.asarray(markers, float).reshape(-1, 3)
    tri = list(itertools.combinations(range(len(iv)), 3))
    perm = permutations3(len(M))
    ti, pj = divmod(int(hyp), len(perm))
    fv = iv[list(tri[ti])]
    wp = M[perm[pj]]
    rc0, s0 = orc.p3p(fv, wp)
    if rc0 != 0:
        return False
    for r in range(3):
        for c in range(3):
            for towards in (-np.inf, np.inf):
                f2 = fv.copy()
                f2[r, c] = np.nextafter(f2[r, c], towards)
                rc, s1 = orc.p3p(f2, wp)
                if rc != rc0 or not np.array_equal(np.isfinite(s0), np.isfinite(s1)):
                    return True
                both = np.isfinite(s0) & np.isfinite(s1)
                if np.abs(np.where(both, s1 - s0, 0.0)).max() > 1e-6:
                    return True
    return False


def attribute_votes(hip, orc, det, markers, K, tol):
    """Trace a histogram difference to hypotheses.  -> dict(differing_hypotheses=[{hyp, w, oracle_p3p_moves, hip, oracle}],
    consistent, explained).  `consistent`: the per-hypothesis histograms of each path add up to its whole-frame one."""
    det = np.asarray(det, float).reshape(-1, 2)
    M = np.asarray(markers, float).reshape(-1, 3)
    n = n_hypotheses(len(det), len(M))
    lo = np.arange(n)
    got = hip.vote_items(det, M, K, tol, lo, lo + 1).astype(np.int64)
    ref = np.stack([orc.vote_items(det, M, K, tol, i, i + 1) for i in range(n)]).astype(np.int64)
    whole_hip = hip.vote_batch([det], M, K, tol)[0].astype(np.int64)
    whole_orc = orc.vote_histogram(det, M, K, tol).astype(np.int64)
    consistent = bool(np.array_equal(got.sum(0), whole_hip) and np.array_equal(ref.sum(0), whole_orc))
    bad = np.nonzero((got != ref).reshape(n, -1).any(1))[0]
    F, ok = hypothesis_quartics(det, M, K)
    w = ferrari_cancellation(F)
    out = []
    for hyp in bad:
        moves = _oracle_p3p_moves(det, M, K, hyp, orc)
        out.append({"hyp": int(hyp), "w": float(w[hyp]), "oracle_p3p_moves_under_1ulp": bool(moves),
                    "unstable": bool(w[hyp] < W_UNSTABLE or moves),
                    "hip_votes": np.argwhere(got[hyp]).tolist(), "oracle_votes": np.argwhere(ref[hyp]).tolist()})
    return {"hypotheses": int(n), "differing_hypotheses": out, "consistent": consistent,
            "explained": bool(consistent and all(h["unstable"] for h in out))}


def classify_mismatch(det, markers, K, tol, orc, hip=None):
    """One verdict for a frame on which the two paths disagree (histogram, correspondences, status or pose)."""
    c = classify_frame(det, markers, K, tol, orc)
    if hip is not None and len(np.asarray(det).reshape(-1, 2)) >= 4:
        a = attribute_votes(hip, orc, det, markers, K, tol)
        c["attribution"] = a
        if a["differing_hypotheses"]:  # the votes differ: the verdict is that of the hypotheses that cast them
            c["unstable"] = bool(a["explained"])
    return c


def attribute_validation(hip, orc, det, markers, K, corr):
    """The other place where the two paths run the quartic: checkCorrespondences' C(n_c,3) P3P solves
    (pose_estimator.cpp:394-542), on the device with IEEE operators + device libm, in the oracle with glibc.  Lists the
    solves whose four [R|C] differ by more than 1e-6 (or in their NaN pattern) and requires each to be one on which the
    oracle's own answer moves under a 1-ulp change of an input, or to sit in the corner (cancellation < 1e-12)."""
    det = np.asarray(det, float).reshape(-1, 2)
    M = np.asarray(markers, float).reshape(-1, 3)
    corr = np.asarray(corr, int).reshape(-1, 2)
    iv = bearings(det, K)
    out = []
    combos = list(itertools.combinations(range(len(corr)), 3))
    if not combos:
        return {"solves": 0, "differing_solves": [], "explained": True}
    fv = np.stack([iv[corr[list(c), 1] - 1] for c in combos])
    wp = np.stack([M[corr[list(c), 0] - 1] for c in combos])
    st, sol = hip.p3p_batch(fv, wp)
    for k in range(len(combos)):
        rc, so = orc.p3p(fv[k], wp[k])
        same_nan = rc == st[k] and (rc != 0 or np.array_equal(np.isfinite(so), np.isfinite(sol[k])))
        d = 0.0
        if rc == 0 and st[k] == 0:
            both = np.isfinite(so) & np.isfinite(sol[k])
            d = float(np.abs(np.where(both, so - sol[k], 0.0)).max())
        if same_nan and d <= 1e-6:
            continue
        moved = False
        for r in range(3):
            for c in range(3):
                f2 = fv[k].copy()
                f2[r, c] = np.nextafter(f2[r, c], np.inf)
                rc2, s2 = orc.p3p(f2, wp[k])
                if rc2 != rc or not np.array_equal(np.isfinite(s2), np.isfinite(so)):
                    moved = True
                elif rc == 0 and np.abs(np.where(np.isfinite(so) & np.isfinite(s2), s2 - so, 0.0)).max() > 1e-6:
                    moved = True
        out.append({"solve": k, "max_abs_diff": d, "nan_pattern_equal": bool(same_nan),
                    "oracle_p3p_moves_under_1ulp": bool(moved), "unstable": bool(moved)})
    return {"solves": len(combos), "differing_solves": out, "explained": bool(all(o["unstable"] for o in out))}


def classify_end_to_end(hip, orc, det, markers, K, hip_params, orc_params):
    """A frame whose STATUS or POSE differs between the two paths: first the votes (attribute_votes), and if the
    histograms agree, the validation's P3P solves on the (then equal) correspondences."""
    det = np.asarray(det, float).reshape(-1, 2)
    tol = float(orc_params.back_projection_pixel_tolerance)
    v = classify_mismatch(det, markers, K, tol, orc, hip)
    a = v.get("attribution")
    if a is not None and a["differing_hypotheses"]:
        v["stage"] = "voting"
        return v
    ref = orc.solve_bruteforce(det, markers, K, orc_params)
    val = attribute_validation(hip, orc, det, markers, K, ref["corr"])
    v["stage"] = "validation"
    v["validation"] = val
    v["unstable"] = bool(val["differing_solves"] and val["explained"])
    return v
