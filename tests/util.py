import numpy as np


def rot_angle(Ra, Rb):
    """Geodesic angle between two rotation matrices (rad)."""
    # ||Ra - Rb||_F = 2*sqrt(2)*sin(angle/2): accurate for tiny angles (arccos of the trace is not)
    return float(2.0 * np.arcsin(min(1.0, np.linalg.norm(Ra - Rb) / (2.0 * np.sqrt(2.0)))))


def pose_diff(Ta, Tb):
    Ta = np.asarray(Ta).reshape(4, 4)
    Tb = np.asarray(Tb).reshape(4, 4)
    return float(np.linalg.norm(Ta[:3, 3] - Tb[:3, 3])), rot_angle(Ta[:3, :3], Tb[:3, :3])


# north_star tolerance: poses match the reference CPU path to <= 1e-4 m / <= 1e-3 rad
POS_TOL_M = 1e-4
ROT_TOL_RAD = 1e-3


def _synth():
    from rpg_monocular_pose_estimator_amd import synth
    return synth


# ---- P3P / quartic checks shared by the GPU tests (device kernels) and the CPU-tier host build of the same source
def random_p3p_problems(rng, n):
    """Bearings of three random world points seen from a random pose (rows = points)."""
    fv, wp = np.zeros((n, 3, 3)), np.zeros((n, 3, 3))
    for i in range(n):
        W = rng.uniform(-0.2, 0.2, (3, 3))
        ax = rng.normal(size=3)
        ax /= np.linalg.norm(ax)
        R = _synth().rodrigues(ax, rng.uniform(0, 1.0))
        t = np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.2, 0.2), rng.uniform(0.6, 3.0)])
        pc = W @ R.T + t
        fv[i] = pc / np.linalg.norm(pc, axis=1, keepdims=True)
        wp[i] = W
    return fv, wp


def ferrari_w(f):
    """|w| = |sqrt(alpha + 2y)| of the reference's Ferrari solver (p3p.cpp:253-272) in Python complex arithmetic,
    relative to the size of its operands: the conditioning indicator of the 2 beta / w term.
    Independent of the device and of the oracle."""
    A, B, C_, D_, E = [float(x) for x in f]
    alpha = -3 * B ** 2 / (8 * A ** 2) + C_ / A
    beta = B ** 3 / (8 * A ** 3) - B * C_ / (2 * A ** 2) + D_ / A
    gamma = -3 * B ** 4 / (256 * A ** 4) + B ** 2 * C_ / (16 * A ** 3) - B * D_ / (4 * A ** 2) + E / A
    P = complex(-alpha ** 2 / 12 - gamma)
    Q = complex(-alpha ** 3 / 108 + alpha * gamma / 3 - beta ** 2 / 8)
    R = -Q / 2 + (Q * Q / 4 + P * P * P / 27) ** 0.5
    U = R ** (1.0 / 3.0)
    y = -5 * alpha / 6 - (Q ** (1.0 / 3.0) if U.real == 0 else P / (3 * U) - U)
    return abs(alpha + 2 * y) / (abs(alpha) + 2 * abs(y) + 1e-300)


def p3p_test_problems():
    rng = np.random.default_rng(11)
    fv, wp = random_p3p_problems(rng, 2000)
    wp[7] = np.array([[0, 0, 0], [0.1, 0, 0], [0.3, 0, 0]])  # collinear
    return fv, wp


def check_p3p_solutions(st, sol, fv, wp, orc):
    """All four [R|C] solutions of every problem against the oracle.  Problems whose solutions differ by more than 1e-6
    are COUNTED, bounded, and each one must be a witnessed instability of the reference algorithm itself: moving one
    input of the ORACLE by one ulp moves the oracle's own answer by more than the disagreement tolerance (the
    alpha + 2y ~ 0 corner of Ferrari)."""
    worst = 0.0
    n_cmp = 0
    unstable = []
    for i in range(len(fv)):
        rc, so = orc.p3p(fv[i], wp[i])
        assert st[i] == rc, i
        if rc != 0:
            assert np.all(sol[i] == 0)
            continue
        fin = np.isfinite(so)
        assert np.array_equal(fin, np.isfinite(sol[i])), i
        d = np.abs(np.where(fin, sol[i] - so, 0.0)).max()
        if d > 1e-6:
            unstable.append((i, d))
            continue
        worst = max(worst, d)
        n_cmp += 1
    assert len(unstable) <= 6 and worst < 1e-6, (n_cmp, worst, unstable)
    for i, d in unstable:
        moved = 0.0
        for comp in range(3):
            w2 = wp[i].copy()
            w2[2, comp] = np.nextafter(w2[2, comp], 1.0)
            rc2, so2 = orc.p3p(fv[i], w2)
            rc0, so0 = orc.p3p(fv[i], wp[i])
            both = np.isfinite(so2) & np.isfinite(so0)
            moved = max(moved, float(np.abs(np.where(both, so2 - so0, 0.0)).max()),
                        1.0 if not np.array_equal(np.isfinite(so2), np.isfinite(so0)) else 0.0)
        assert moved > 1e-7, ("disagreement on a problem the reference algorithm solves stably", i, d, moved)


def quartic_test_problems(variant):
    rng = np.random.default_rng(5 + variant)
    f = rng.normal(size=(4000, 5))
    f[:, 0] = np.where(np.abs(f[:, 0]) < 0.05, 1.0, f[:, 0])
    # quartics with four known real roots as well
    roots = rng.uniform(-1, 1, (1000, 4))
    for i in range(1000):
        f[i] = np.poly(roots[i]) * rng.uniform(0.5, 2.0)
    return f, roots


def check_quartic_roots(got, f, roots, orc):
    """Every quartic whose real parts differ from the oracle's by more than 1e-9 is counted, bounded, and must sit in
    the unstable corner: |alpha + 2y| small against its operands (error amplification of the 2 beta / w term
    ~ 1 / |w|^2), computed in plain Python complex arithmetic."""
    ref = np.array([orc.solve_quartic(f[i]) for i in range(len(f))])
    err = np.abs(got - ref).max(axis=1)
    scale = np.maximum(1.0, np.abs(ref).max(axis=1))
    bad = np.nonzero(~(err <= 1e-9 * scale))[0]
    assert len(bad) <= 12, (len(bad), err[bad])
    for i in bad:
        cond = ferrari_w(f[i])
        # rounding (1e-16) amplified by 1 / cond: a disagreement of err needs cond <~ 1e-16 / err ... generously:
        assert cond < 1e-3 or err[i] / scale[i] < 1e-13 / max(cond, 1e-300) ** 2, (i, err[i], cond)
    good = np.sort(got[:1000], axis=1) - np.sort(roots, axis=1)
    assert np.median(np.abs(good).max(axis=1)) < 1e-9
