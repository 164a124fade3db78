"""Seeded synthetic workload for the hot path: camera, marker sets, poses, frames.

The reference ships one marker file (4 LEDs) and no images, so the BASELINE configs are defined
here: README camera (README.md:165-166), marker sets M4 (demo_marker_positions.yaml:4-15), M5, M8,
demo.launch:12-22 parameters, uint8 frames with U{0..30} background and Gaussian LED spots placed
at the DISTORTED projection (distortion model of LEDDetector::distortPoints, LED.cpp:200-221).

numpy only (host).  `render_frames_torch` renders the same scene description on a torch device for
the large bench batches.
"""
import numpy as np

README_K = np.array([[615.652408400557, 0.0, 362.655454167686],
                     [0.0, 616.760184718123, 256.67210750994],
                     [0.0, 0.0, 1.0]])
README_D = np.array([-0.358561237166698, 0.149312912580924, 0.000484551782515636,
                     -0.000200189442379448, 0.0])

M4 = np.array([[0.0714197, 0.0800214, 0.0622611],
               [0.0400755, -0.0912328, 0.0317064],
               [-0.0647293, -0.0879977, 0.0830852],
               [-0.0558663, -0.0165446, 0.053473]])
M5 = np.vstack([M4, [[0.0120, 0.0310, 0.1210]]])
M8 = np.vstack([M5, [[0.0850, -0.0200, 0.0100], [-0.0300, 0.0900, 0.0200], [0.0100, -0.0500, 0.0950]]])

# demo.launch:12-22
DEMO_PARAMS = dict(threshold_value=140, gaussian_sigma=0.6, min_blob_area=10.0, max_blob_area=200.0,
                   max_width_height_distortion=0.5, max_circular_distortion=0.5,
                   back_projection_pixel_tolerance=5.0, nearest_neighbour_pixel_tolerance=7.0,
                   certainty_threshold=0.75, valid_correspondence_threshold=0.7,
                   roi_border_thickness=20, histogram_threshold=0)

# BASELINE.json configs[0..3] (C5 = 8 x C2 streams)
CONFIGS = {
    "C1": dict(rows=480, cols=752, markers=M4, n_distractors=0, spot_sigma=1.5),
    "C2": dict(rows=480, cols=752, markers=M5, n_distractors=0, spot_sigma=1.5),
    "C3": dict(rows=480, cols=752, markers=M8, n_distractors=4, spot_sigma=1.5),
    "C4": dict(rows=1200, cols=1920, markers=M5, n_distractors=0, spot_sigma=2.0),
}


def camera_for(rows, cols):
    """README camera, scaled when the frame is not 752x480 (same D)."""
    K = README_K.copy()
    sx, sy = cols / 752.0, rows / 480.0
    K[0, 0] *= sx
    K[0, 2] *= sx
    K[1, 1] *= sy
    K[1, 2] *= sy
    return K, README_D.copy()


def distort_px(xy, K, D):
    """Pinhole pixel -> distorted pixel (plumb-bob, k1 k2 p1 p2 k3)."""
    xy = np.asarray(xy, np.float64)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    k1, k2, p1, p2, k3 = D[:5]
    x = (xy[..., 0] - cx) / fx
    y = (xy[..., 1] - cy) / fy
    r2 = x * x + y * y
    rad = 1.0 + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2
    xc = x * rad + (2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x))
    yc = y * rad + (p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y)
    return np.stack([xc * fx + cx, yc * fy + cy], axis=-1)


def rodrigues(axis, angle):
    a = np.asarray(axis, np.float64)
    a = a / np.linalg.norm(a)
    Kx = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(angle) * Kx + (1 - np.cos(angle)) * (Kx @ Kx)


def project(T, pts, K):
    pc = (T[:3, :3] @ np.asarray(pts).T).T + T[:3, 3]
    return np.stack([K[0, 0] * pc[:, 0] / pc[:, 2] + K[0, 2], K[1, 1] * pc[:, 1] / pc[:, 2] + K[1, 2]], -1)


def sample_scene(rng, markers, K, D, rows, cols, n_distractors=0, min_sep=12.0, margin=12.0):
    """One random pose (T_camera_object) whose LEDs (after distortion) are >= margin px inside the
    image and pairwise >= min_sep px apart, plus distractor spot positions."""
    while True:
        axis = rng.normal(size=3)
        ang = rng.uniform(0.0, 0.8)
        T = np.eye(4)
        T[:3, :3] = rodrigues(axis, ang)
        T[:3, 3] = [rng.uniform(-0.3, 0.3), rng.uniform(-0.2, 0.2), rng.uniform(0.8, 2.5)]
        px = distort_px(project(T, markers, K), K, D)
        if (px[:, 0].min() < margin or px[:, 0].max() > cols - 1 - margin or px[:, 1].min() < margin
                or px[:, 1].max() > rows - 1 - margin):
            continue
        d = np.linalg.norm(px[:, None, :] - px[None, :, :], axis=-1) + np.eye(len(px)) * 1e9
        if d.min() < min_sep:
            continue
        spots = px
        tries = 0
        while len(spots) < len(px) + n_distractors and tries < 1000:
            tries += 1
            c = np.array([rng.uniform(margin, cols - 1 - margin), rng.uniform(margin, rows - 1 - margin)])
            if np.linalg.norm(spots - c, axis=1).min() >= min_sep:
                spots = np.vstack([spots, c])
        if len(spots) < len(px) + n_distractors:
            continue
        return T, spots


def render_frame(rng, spots, rows, cols, spot_sigma=1.5, peak=400.0, bg_max=30):
    """uint8 frame: background i.i.d. U{0..bg_max}; every spot adds round(peak*exp(-d^2/2s^2)),
    clipped to 255."""
    img = rng.integers(0, bg_max + 1, size=(rows, cols)).astype(np.float64)
    rad = int(np.ceil(4 * spot_sigma)) + 1
    for (sx, sy) in spots:
        x0, x1 = max(0, int(np.floor(sx)) - rad), min(cols, int(np.floor(sx)) + rad + 2)
        y0, y1 = max(0, int(np.floor(sy)) - rad), min(rows, int(np.floor(sy)) + rad + 2)
        if x0 >= x1 or y0 >= y1:
            continue
        xs = np.arange(x0, x1)[None, :]
        ys = np.arange(y0, y1)[:, None]
        g = peak * np.exp(-((xs - sx) ** 2 + (ys - sy) ** 2) / (2.0 * spot_sigma ** 2))
        img[y0:y1, x0:x1] += np.floor(g + 0.5)
    return np.clip(img, 0, 255).astype(np.uint8)


def make_scenes(config, n, seed):
    """Scene descriptions only (poses + spot centres), cheap: -> T (n,4,4), spots (n,S,2)."""
    cfg = CONFIGS[config] if isinstance(config, str) else config
    K, D = camera_for(cfg["rows"], cfg["cols"])
    Ts, sp = [], []
    for i in range(n):
        rng = np.random.default_rng([seed, i])
        T, spots = sample_scene(rng, cfg["markers"], K, D, cfg["rows"], cfg["cols"], cfg["n_distractors"])
        Ts.append(T)
        sp.append(spots)
    return np.array(Ts), np.array(sp)


def make_scenes_batch(config, n, seed, min_sep=12.0, margin=12.0):
    """Vectorised scene sampler for large bench batches (same distribution and acceptance rule as
    sample_scene, different random stream).  -> T (n,4,4), spots (n,S,2)."""
    cfg = CONFIGS[config] if isinstance(config, str) else config
    K, D = camera_for(cfg["rows"], cfg["cols"])
    rows, cols, M, nd = cfg["rows"], cfg["cols"], np.asarray(cfg["markers"]), cfg["n_distractors"]
    rng = np.random.default_rng(seed)
    Ts, sp = [], []
    need = n
    while need > 0:
        m = max(256, int(need * 1.6))
        axis = rng.normal(size=(m, 3))
        axis /= np.linalg.norm(axis, axis=1, keepdims=True)
        ang = rng.uniform(0.0, 0.8, m)
        Kx = np.zeros((m, 3, 3))
        Kx[:, 0, 1], Kx[:, 0, 2] = -axis[:, 2], axis[:, 1]
        Kx[:, 1, 0], Kx[:, 1, 2] = axis[:, 2], -axis[:, 0]
        Kx[:, 2, 0], Kx[:, 2, 1] = -axis[:, 1], axis[:, 0]
        R = np.eye(3)[None] + np.sin(ang)[:, None, None] * Kx + (1 - np.cos(ang))[:, None, None] * (Kx @ Kx)
        t = np.stack([rng.uniform(-0.3, 0.3, m), rng.uniform(-0.2, 0.2, m), rng.uniform(0.8, 2.5, m)], 1)
        pc = np.einsum("mij,kj->mki", R, M) + t[:, None, :]
        px = np.stack([K[0, 0] * pc[..., 0] / pc[..., 2] + K[0, 2], K[1, 1] * pc[..., 1] / pc[..., 2] + K[1, 2]], -1)
        px = distort_px(px, K, D)
        ok = ((px[..., 0].min(1) >= margin) & (px[..., 0].max(1) <= cols - 1 - margin) &
              (px[..., 1].min(1) >= margin) & (px[..., 1].max(1) <= rows - 1 - margin))
        dd = np.linalg.norm(px[:, :, None, :] - px[:, None, :, :], axis=-1) + np.eye(len(M))[None] * 1e9
        ok &= dd.min((1, 2)) >= min_sep
        idx = np.nonzero(ok)[0]
        for i in idx:
            if need == 0:
                break
            spots = px[i]
            tries = 0
            while len(spots) < len(M) + nd and tries < 1000:
                tries += 1
                c = np.array([rng.uniform(margin, cols - 1 - margin), rng.uniform(margin, rows - 1 - margin)])
                if np.linalg.norm(spots - c, axis=1).min() >= min_sep:
                    spots = np.vstack([spots, c])
            if len(spots) < len(M) + nd:
                continue
            T = np.eye(4)
            T[:3, :3], T[:3, 3] = R[i], t[i]
            Ts.append(T)
            sp.append(spots)
            need -= 1
    return np.array(Ts), np.array(sp)


def make_frames(config, n, seed):
    """-> dict(frames (n,rows,cols) u8, T_true (n,4,4), spots (n,S,2), K, D, markers)."""
    cfg = CONFIGS[config] if isinstance(config, str) else config
    K, D = camera_for(cfg["rows"], cfg["cols"])
    Ts, spots = make_scenes(cfg, n, seed)
    frames = np.empty((n, cfg["rows"], cfg["cols"]), np.uint8)
    for i in range(n):
        rng = np.random.default_rng([seed, i, 1])
        frames[i] = render_frame(rng, spots[i], cfg["rows"], cfg["cols"], cfg["spot_sigma"])
    return dict(frames=frames, T_true=Ts, spots=spots, K=K, D=D, markers=np.asarray(cfg["markers"]),
                rows=cfg["rows"], cols=cfg["cols"])


def make_sequence(config, n, seed, dt=0.02, lin_speed=0.15, ang_speed=0.6, dropout=(), salt=0.0):
    """A smooth trajectory (constant body twist + small jitter) for the tracking path: frame k is at
    time k*dt.  Frames listed in `dropout` are rendered with only 2 LEDs (forces the retry /
    re-initialisation logic); `salt`: that fraction of every frame's pixels saturated (isolated bright pixels inside
    and outside the ROI).  -> dict(frames, T_true, times, K, D, markers)"""
    cfg = CONFIGS[config] if isinstance(config, str) else config
    K, D = camera_for(cfg["rows"], cfg["cols"])
    rows, cols, M = cfg["rows"], cfg["cols"], np.asarray(cfg["markers"])
    rng = np.random.default_rng([seed, 7])
    while True:
        T0, _ = sample_scene(rng, M, K, D, rows, cols, 0, margin=80.0)
        v = rng.normal(size=3)
        v *= lin_speed / np.linalg.norm(v)
        w = rng.normal(size=3)
        w *= ang_speed / np.linalg.norm(w)
        Ts, ok = [], True
        for k in range(n):
            t = k * dt
            Tk = np.eye(4)
            Tk[:3, :3] = rodrigues(w, np.linalg.norm(w) * t) @ T0[:3, :3]
            Tk[:3, 3] = T0[:3, 3] + v * t + 0.0005 * rng.normal(size=3)
            px = distort_px(project(Tk, M, K), K, D)
            d = np.linalg.norm(px[:, None, :] - px[None, :, :], axis=-1) + np.eye(len(M)) * 1e9
            if (px[:, 0].min() < 20 or px[:, 0].max() > cols - 21 or px[:, 1].min() < 20 or px[:, 1].max() > rows - 21
                    or Tk[2, 3] < 0.5 or d.min() < 12.0):
                ok = False
                break
            Ts.append(Tk)
        if ok:
            break
    frames = np.empty((n, rows, cols), np.uint8)
    for k in range(n):
        px = distort_px(project(Ts[k], M, K), K, D)
        if k in dropout:
            px = px[:2]
        frames[k] = render_frame(np.random.default_rng([seed, k, 3]), px, rows, cols, cfg["spot_sigma"])
        if salt > 0.0:
            frames[k][np.random.default_rng([seed, k, 13]).random((rows, cols)) < salt] = 255
    return dict(frames=frames, T_true=np.array(Ts), times=np.arange(n) * dt, K=K, D=D, markers=M, rows=rows, cols=cols)


CLUTTER_KINDS = ("salt", "salt_dense", "patch", "ring", "grid", "d4", "d16")


def make_clutter_frames(kind, n, seed, config="C2"):
    """C2 frames (5 LEDs; `config`: another BASELINE config, e.g. the 1920x1200 C4) with what a real camera adds
    (numpy, deterministic; the clutter curve of bench.py draws the same kinds on the device): `salt` 0.05 % isolated
    saturated pixels, `salt_dense` 0.3 %, `patch` one saturated 64x64 square, `ring` a bright ring around the image
    centre (RETR_EXTERNAL drops what it encloses), `grid` a dot grid (every 7th row, 5th column), `d4` / `d16`
    distractor spots.  -> dict like make_frames."""
    cfg = dict(CONFIGS[config])
    if kind in ("d4", "d16"):
        cfg["n_distractors"] = int(kind[1:])
    d = make_frames(cfg, n, seed)
    rows, cols = d["rows"], d["cols"]
    for i in range(n):
        rng = np.random.default_rng([seed, i, 11])
        f = d["frames"][i]
        if kind in ("salt", "salt_dense"):
            f[rng.random((rows, cols)) < (0.0005 if kind == "salt" else 0.003)] = 255
        elif kind == "patch":
            y0, x0 = int(rng.integers(0, rows - 64)), int(rng.integers(0, cols - 64))
            f[y0:y0 + 64, x0:x0 + 64] = 255
        elif kind == "ring":
            yy, xx = np.mgrid[0:rows, 0:cols]
            rad = float(rng.uniform(120, 230))
            f[np.abs(np.hypot(xx - cols / 2, yy - rows / 2) - rad) < 2.5] = 250
        elif kind == "grid":
            f[::7, ::5] = np.maximum(f[::7, ::5], 180)
        elif kind not in ("d4", "d16"):
            raise ValueError(kind)
    d["kind"] = kind
    return d


def render_frames_torch(spots, rows, cols, spot_sigma, device, seed=0, peak=400.0, bg_max=30, out=None):
    """Render scenes on a torch device (bench plumbing): same image model as render_frame, noise
    from torch's generator.  spots: (n,S,2) numpy.  -> uint8 tensor (n,rows,cols)."""
    import torch
    n, S, _ = spots.shape
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    if out is None:
        out = torch.empty((n, rows, cols), dtype=torch.uint8, device=device)
    rad = int(np.ceil(4 * spot_sigma)) + 1
    win = 2 * rad + 2
    sp = torch.as_tensor(spots, dtype=torch.float64, device=device)  # (n,S,2)
    chunk = 256
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        img = torch.randint(0, bg_max + 1, (b - a, rows, cols), generator=g, device=device, dtype=torch.int16)
        s = sp[a:b]
        x0 = torch.floor(s[..., 0]).long() - rad
        y0 = torch.floor(s[..., 1]).long() - rad
        off = torch.arange(win, device=device)
        xs = x0[..., None] + off  # (m,S,win)
        ys = y0[..., None] + off
        gx = (xs.double() - s[..., 0:1]) ** 2
        gy = (ys.double() - s[..., 1:2]) ** 2
        val = torch.floor(peak * torch.exp(-(gy[..., :, None] + gx[..., None, :]) / (2.0 * spot_sigma ** 2)) + 0.5)
        valid = ((ys >= 0) & (ys < rows))[..., :, None] & ((xs >= 0) & (xs < cols))[..., None, :]
        fi = torch.arange(b - a, device=device)[:, None, None, None].expand(-1, S, win, win)
        yy = ys.clamp(0, rows - 1)[..., :, None].expand(-1, -1, -1, win)
        xx = xs.clamp(0, cols - 1)[..., None, :].expand(-1, -1, win, -1)
        val = torch.where(valid, val, torch.zeros_like(val)).to(torch.int16)
        img.index_put_((fi.reshape(-1), yy.reshape(-1), xx.reshape(-1)), val.reshape(-1), accumulate=True)
        out[a:b] = img.clamp_(0, 255).to(torch.uint8)
    return out
