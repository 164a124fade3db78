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


# ---- the Gaussian blur of CV_8U images in the three generations of OpenCV's smooth.cpp ----------------------------
# (restated from the published sources of OpenCV's imgproc module; there is no OpenCV in this image to run.)
#   A  "8u32s"        OpenCV 2.4 .. 3.4.1: generic separable filter engine, getGaussianKernel(CV_32F), both kernels
#                     converted to CV_32S with 8 fractional bits (cvRound), row filter 8u -> 32s, column filter with
#                     FixedPtCastEx: (sum + 2^15) >> 16.  This is what the oracle, the witness pipeline and the HIP
#                     kernels restate (ROS Kinetic / Ubuntu 16.04, the platform README:9 names, ships OpenCV 3.3.1).
#   B  "ufixedpoint16" OpenCV 3.4.2 .. 4.1.1: GaussianBlurFixedPoint<uint8_t, ufixedpoint16>: getGaussianKernel(CV_64F),
#                     every tap rounded to 8 fractional bits, horizontal pass accumulated in SATURATING 8.8 fixed
#                     point (16 bit), vertical pass in 16.16, (sum + 2^15) >> 16.
#   C  "bit-exact"    OpenCV >= 4.1.2: the same data path as B with the kernel from getGaussianKernelBitExact +
#                     getGaussianKernelFixedPoint_ED: rounding with error diffusion from the outside in, the centre
#                     tap takes the remainder so that the taps sum to exactly 256.
def gaussian_ksize_8u(sigma):
    return int(np.rint(sigma * 6 + 1)) | 1


def _gaussian_kernel_f64(sigma):
    n = gaussian_ksize_8u(sigma)
    x = np.arange(n) - (n - 1) * 0.5
    k = np.exp(-0.5 * x * x / (sigma * sigma))
    return k * (1.0 / k.sum())


def gaussian_taps_ufixedpoint16(sigma):
    """generation B: every tap of the double-precision kernel rounded to 8 fractional bits."""
    return np.rint(_gaussian_kernel_f64(sigma) * 256.0).astype(np.int64)


def gaussian_taps_bitexact_ed(sigma):
    """generation C: error diffusion from the outermost tap inwards, centre tap = 256 - the rest."""
    k = _gaussian_kernel_f64(sigma)
    n = len(k)
    out = np.zeros(n, np.int64)
    err = 0.0
    for i in range(n // 2):
        adj = k[i] * 256.0 + err
        v = int(np.rint(adj))
        err = adj - v
        out[i] = out[n - 1 - i] = v
    out[n // 2] = 256 - out.sum()
    return out


def blur_mask_generation(img, thr, taps, saturating16):
    """Non-zero mask of threshold(TOZERO) + separable fixed-point blur with the given 8-bit taps;
    saturating16 = the 8.8 / 16.16 data path of generations B / C (the horizontal sums saturate at 65535)."""
    t = np.where(img > thr, img, 0).astype(np.int64)
    k = np.asarray(taps, np.int64)
    r = len(k) // 2
    p = np.pad(t, r, mode="reflect") if r else t
    h = sum(k[j] * p[:, j:j + t.shape[1]] for j in range(len(k)))
    if saturating16:
        h = np.minimum(h, 65535)
    v = sum(k[i] * h[i:i + t.shape[0], :] for i in range(len(k)))
    if saturating16:
        v = np.minimum(v, (1 << 32) - 1)
    return ((v + (1 << 15)) >> 16) != 0
