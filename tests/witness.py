"""Independent witnesses for the CPU oracle (numpy / scipy / mpmath only — nothing from oracle/ or
the product).  The reference ships no tests, so these are the known-answer checks that pin the
oracle's semantics."""
import numpy as np


def quartic_roots_mp(factors):
    """Roots of A x^4 + B x^3 + C x^2 + D x + E with mpmath at 50 digits."""
    import mpmath as mp
    mp.mp.dps = 50
    return [complex(r) for r in mp.polyroots([mp.mpf(float(c)) for c in factors], maxsteps=200, extraprec=200)]


def gaussian_taps_q8(sigma):
    """cv::getGaussianKernel(CV_32F) -> 8 fractional bits, written independently with numpy."""
    n = int(np.rint(sigma * 6 + 1)) | 1
    x = np.arange(n) - (n - 1) * 0.5
    k = np.exp(-0.5 * x * x / (sigma * sigma)).astype(np.float32)
    k = (k * np.float32(1.0 / k.astype(np.float64).sum())).astype(np.float32)
    return np.rint(k.astype(np.float64) * 256.0).astype(np.int64)


def blur_fixed_point(img, thr, sigma):
    """threshold(TOZERO) + separable fixed-point Gaussian with BORDER_REFLECT_101."""
    t = np.where(img > thr, img, 0).astype(np.int64)
    k = gaussian_taps_q8(sigma)
    r = len(k) // 2
    if r == 0:
        return t.astype(np.uint8)
    p = np.pad(t, r, mode="reflect")
    h = sum(k[j] * p[:, j:j + t.shape[1]] for j in range(len(k)))
    v = sum(k[i] * h[i:i + t.shape[0], :] for i in range(len(k)))
    return np.clip((v + (1 << 15)) >> 16, 0, 255).astype(np.uint8)


def shoelace(pts):
    x, y = pts[:, 0].astype(np.float64), pts[:, 1].astype(np.float64)
    return 0.5 * abs(np.sum(np.roll(x, 1) * y - x * np.roll(y, 1)))


def polygon_centroid(pts):
    x, y = pts[:, 0].astype(np.float64), pts[:, 1].astype(np.float64)
    xp, yp = np.roll(x, 1), np.roll(y, 1)
    d = xp * y - x * yp
    a = d.sum() / 2.0
    return np.array([((xp + x) * d).sum() / (6 * a), ((yp + y) * d).sum() / (6 * a)])


def se3_exp(twist):
    from scipy.linalg import expm
    u, w = np.asarray(twist[:3], float), np.asarray(twist[3:], float)
    M = np.zeros((4, 4))
    M[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
    M[:3, 3] = u
    return expm(M)


def project(T, p3, K):
    pc = T[:3, :3] @ p3 + T[:3, 3]
    return np.array([K[0, 0] * pc[0] / pc[2] + K[0, 2], K[1, 1] * pc[1] / pc[2] + K[1, 2]])


def kabsch(A, B):
    """Rigid transform B ~ R A + t from numpy's SVD, no reflection guard (as the reference)."""
    a0, b0 = A.mean(0), B.mean(0)
    H = (A - a0).T @ (B - b0)
    U, _, Vt = np.linalg.svd(H)
    R = Vt.T @ U.T
    return R, b0 - R @ a0
