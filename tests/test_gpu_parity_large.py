"""GPU parity tests at the sizes VERDICT round 2 asked for: C3 (8 LEDs / 12 detections) and C4 (1920 x 1200) on 256
frames each against the oracle, a full-size C4 batch through size-independent properties, marker sets up to the
documented capacity (12 and 16 markers) in both voting arithmetics, and the frames on which the HIP path is KNOWN to
differ from this CPU build of the oracle — each of those must be traced, hypothesis by hypothesis, to an instability of
the reference algorithm itself (tests/forensics.py), or the test fails."""
import glob
import os

import numpy as np
import pytest

from rpg_monocular_pose_estimator_amd import synth
import rpg_monocular_pose_estimator_amd as mpe
from util import pose_diff, POS_TOL_M, ROT_TOL_RAD
import forensics

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _frames_on_device(config, n, seed):
    import torch
    cfg = synth.CONFIGS[config]
    T_true, spots = synth.make_scenes_batch(cfg, n, seed=seed)
    dev = torch.device("cuda", 0)
    frames = synth.render_frames_torch(spots, cfg["rows"], cfg["cols"], cfg["spot_sigma"], dev, seed=seed + 1)
    torch.cuda.synchronize()
    return cfg, T_true, frames


def _compare_with_oracle(hip, orc, config, n, seed, tol, min_pose_frac):
    """n frames end to end; every disagreement must be explained by the forensics; returns the number of poses."""
    cfg, _, frames = _frames_on_device(config, n, seed)
    K, D = synth.camera_for(cfg["rows"], cfg["cols"])
    markers = np.asarray(cfg["markers"])
    Ph = mpe.demo_params(back_projection_pixel_tolerance=tol)
    Po = orc.make_params(back_projection_pixel_tolerance=tol)
    host = frames.cpu().numpy()
    got = hip.estimate_batch(frames, markers, K, D, Ph)
    ref = orc.estimate_batch(host, markers, K, D, Po, n_threads=len(os.sched_getaffinity(0)))
    assert np.array_equal(got["n_det"], ref["n_det"])
    # detections bit-equal on a sub-sample (the oracle's findLeds alone, frame by frame)
    det = hip.detect_batch(frames[:32], K, D, Ph)
    for i in range(32):
        und, dist = orc.find_leds(host[i], Po, K, D)
        k = len(und)
        assert det["n"][i] == k and np.array_equal(det["undist_xy"][i][:2 * k].reshape(-1, 2), und), i
        assert np.array_equal(det["dist_xy"][i][:2 * k].reshape(-1, 2), dist), i
    n_pose = n_bad = 0
    for i in range(n):
        same = got["status"][i] == ref["status"][i] and got["n_corr"][i] == ref["n_corr"][i]
        if same and ref["status"][i] == 0:
            dp, dr = pose_diff(got["T"][i], ref["T"][i])
            same = dp <= POS_TOL_M and dr <= ROT_TOL_RAD
            n_pose += 1
        if not same:
            n_bad += 1
            und, _ = orc.find_leds(host[i], Po, K, D)
            v = forensics.classify_end_to_end(hip, orc, und, markers, K, Ph, Po)
            assert v["unstable"], ("unexplained HIP-vs-oracle mismatch", config, i, v)
    # rounds 4 - 5 tolerated max(1, n / 128) mismatches the forensics classified as "unstable" (Ferrari's corner, where the
    # device's exact complex powers and glibc's exp(y log|z|) picked different branches).  The default arithmetic now
    # evaluates those powers as the CPU build does (vote_arith 3, csrc/mpe_ddmath.h): the allowance is gone
    assert hip.get_option("vote_arith") == 3
    assert n_bad == 0, n_bad
    assert n_pose >= min_pose_frac * n, (n_pose, n)
    return n_pose


@pytest.mark.parametrize("tol,min_pose_frac", [(2.0, 0.5), (5.0, 0.0)])
def test_c3_256_frames_against_the_oracle(hip, orc, tol, min_pose_frac):
    """BASELINE configs[2]: 8 LEDs + 4 distractors, 73 920 P3P solves per frame, 56 validation solves.  With the demo
    tolerance (5 px) the reference algorithm itself mostly fails to initialise (12 detections x 8 markers pollute the
    vote table) — parity of the verdicts is what is checked there; at 2 px most frames yield a pose."""
    _compare_with_oracle(hip, orc, "C3", 256, 5150 + int(tol), tol, min_pose_frac)


def test_c4_256_frames_against_the_oracle(hip, orc):
    """BASELINE configs[3]: 1920 x 1200 frames, 5 LEDs."""
    _compare_with_oracle(hip, orc, "C4", 256, 6160, 5.0, 0.85)


def test_c4_full_size_batch_properties(orc):
    """16 384 device-resident 1920 x 1200 frames (37.7 GB; 2 sub-batches of 8192): every schedule bit-identical to the
    plain chain of kernels, frames independent (reversed batch -> reversed records), a random sample equal to the
    oracle, poses close to the ground truth of the synthetic scenes."""
    import torch
    B = 16384
    cfg, T_true, frames = _frames_on_device("C4", B, 7170)
    rows, cols = cfg["rows"], cfg["cols"]
    K, D = synth.camera_for(rows, cols)
    markers = np.asarray(cfg["markers"])
    dev = frames.device
    P = mpe.demo_params()
    h = mpe.Handle(0)
    stream = torch.cuda.Stream(device=dev)
    h.set_stream(stream.cuda_stream)
    nb = B * mpe.RESULT_DTYPE.itemsize

    def run(fr, pipeline, mode):
        with torch.cuda.stream(stream):
            out = torch.zeros(nb, dtype=torch.uint8, device=dev)
            h.set_option("pipeline", pipeline)
            h.set_option("pipeline_mode", mode)
            h.estimate_batch_device(fr.data_ptr(), B, rows, cols, markers, K, D, P, out.data_ptr())
        stream.synchronize()
        return out

    plain = run(frames, 1, -1)
    for mode in (3, 4, 6, -1):
        piped = run(frames, 16, mode)
        assert torch.equal(piped, plain), mode
    assert h.get_option("last_schedule") == 6
    rec = np.frombuffer(plain.cpu().numpy().tobytes(), mpe.RESULT_DTYPE)
    # (reversing 37.7 GB in place would need a second copy of the batch: reverse the first quarter instead)
    q = B // 4
    flipped = torch.flip(frames[:q], dims=[0]).contiguous()
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        out = torch.zeros(q * mpe.RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        h.estimate_batch_device(flipped.data_ptr(), q, rows, cols, markers, K, D, P, out.data_ptr())
    stream.synchronize()
    assert torch.equal(out.view(q, -1).flip(0), plain.view(B, -1)[:q])
    del flipped
    found = rec["status"] == 0
    assert 0.88 < found.mean() <= 1.0 and (rec["status"] >= 0).all(), found.mean()
    T = rec["T"].reshape(B, 4, 4)
    dpos = np.linalg.norm(T[found][:, :3, 3] - T_true[found][:, :3, 3], axis=1)
    assert np.median(dpos) < 5e-3 and np.mean(dpos < 0.05) > 0.97, (np.median(dpos), np.mean(dpos < 0.05))
    idx = np.random.default_rng(1).choice(B, 64, replace=False)
    sample = frames[torch.as_tensor(idx, device=dev)].cpu().numpy()
    ref = orc.estimate_batch(sample, markers, K, D, orc.make_params(), n_threads=len(os.sched_getaffinity(0)))
    for j, i in enumerate(idx):
        assert rec["status"][i] == ref["status"][j] and rec["n_det"][i] == ref["n_det"][j], i
        if ref["status"][j] == 0:
            dp, dr = pose_diff(T[i], ref["T"][j].reshape(4, 4))
            assert dp <= POS_TOL_M and dr <= ROT_TOL_RAD, (i, dp, dr)
    h.close()


@pytest.mark.parametrize("n_markers", [9, 12, 14, 16])
def test_many_markers_both_voting_kernels(orc, n_markers):
    """Marker sets up to MPE_MAX_MARKERS: P(16,3) = 3360 permutations, 13 unused markers per hypothesis — the plain
    voting kernel's per-thread LDS columns (24 bytes per unused marker) decide its block size there — in the fast and
    in the strict arithmetic, histograms integer-equal to the oracle, and the whole brute-force solve."""
    rng = np.random.default_rng(900 + n_markers)
    K, _ = synth.camera_for(480, 752)
    markers = rng.uniform(-0.12, 0.12, (n_markers, 3))
    T = np.eye(4)
    T[:3, :3] = synth.rodrigues(np.array([0.3, -0.5, 0.8]) / np.linalg.norm([0.3, -0.5, 0.8]), 0.4)
    T[:3, 3] = [0.03, -0.02, 1.1]
    px = synth.project(T, markers, K)
    sets = []
    for n_d in (5, 6):
        pick = rng.choice(n_markers, n_d, replace=False)
        sets.append(np.asarray(px[pick] + rng.normal(0, 0.05, (n_d, 2)), np.float32).astype(float))
    h = mpe.Handle()
    try:
        for arith in (1, 0):
            h.set_option("vote_arith", arith)
            got = h.vote_batch(sets, markers, K, 3.0)
            for i, det in enumerate(sets):
                ref = orc.vote_histogram(det, markers, K, 3.0)
                if not np.array_equal(got[i], ref):
                    v = forensics.classify_mismatch(det, markers, K, 3.0, orc, h)
                    assert v["unstable"], (n_markers, arith, i, v)
                assert ref.sum() > 0
        h.set_option("vote_arith", 1)
        Po, Ph = orc.make_params(back_projection_pixel_tolerance=3.0), mpe.demo_params(back_projection_pixel_tolerance=3.0)
        for det in sets:
            ro = orc.solve_bruteforce(det, markers, K, Po)
            rh = h.solve_bruteforce(det, markers, K, Ph)
            assert rh["status"] == ro["status"] and np.array_equal(rh["corr"], ro["corr"])
    finally:
        h.close()


def _known_sets():
    files = sorted(glob.glob(os.path.join(HERE, "data", "unstable_det_*.npy")))
    return [os.path.join(HERE, "data", "vote_regression_det_0.npy")] + files


@pytest.mark.parametrize("path", _known_sets(), ids=lambda p: os.path.basename(p))
def test_known_mismatching_frames_are_explained(hip, orc, path):
    """Detection sets on which a build of the HIP path differed from this CPU build of the oracle (found by the parity
    soaks, committed under tests/data).  Whatever the current build does on them — agree or differ — a difference must
    be traced to hypotheses on which the reference algorithm disagrees with itself, and the oracle's own histogram must
    move under a 1-ulp change of a detection coordinate (that is what made the frame a mismatch in the first place)."""
    det = np.load(path)
    K, _ = synth.camera_for(480, 752)
    for arith in (1, 0):
        h = mpe.Handle()
        try:
            h.set_option("vote_arith", arith)
            got = h.vote_batch([det], synth.M5, K, 5.0)[0].astype(int)
            ref = orc.vote_histogram(det, synth.M5, K, 5.0).astype(int)
            v = forensics.classify_mismatch(det, synth.M5, K, 5.0, orc, h)
            assert v["oracle_flips_under_1ulp"] or v["min_w"] < forensics.W_UNSTABLE, v
            a = v["attribution"]
            assert a["consistent"]
            if not np.array_equal(got, ref):
                assert a["differing_hypotheses"] and a["explained"], a
                assert np.abs(got - ref).max() <= 2
        finally:
            h.close()


def test_forensics_reject_an_injected_error(hip, orc):
    """The classifier must not explain everything: a histogram difference that does NOT come from an unstable hypothesis
    (here: the HIP histogram of a slightly different tolerance, compared with the oracle at the nominal one) is
    reported as unexplained."""
    d = synth.make_frames("C2", 12, seed=77)
    n_checked = 0
    for i in range(12):
        und, _ = orc.find_leds(d["frames"][i], orc.make_params(), d["K"], d["D"])
        if len(und) < 5:
            continue
        got = hip.vote_batch([und], d["markers"], d["K"], 9.0)[0]
        ref = orc.vote_histogram(und, d["markers"], d["K"], 5.0)
        if np.array_equal(got, ref):
            continue
        # per-hypothesis comparison of the HIP path at 9 px against the oracle at 5 px
        n = forensics.n_hypotheses(len(und), len(d["markers"]))
        lo = np.arange(n)
        g = hip.vote_items(und, d["markers"], d["K"], 9.0, lo, lo + 1).astype(int)
        r = np.stack([orc.vote_items(und, d["markers"], d["K"], 5.0, k, k + 1) for k in range(n)]).astype(int)
        bad = np.nonzero((g != r).reshape(n, -1).any(1))[0]
        assert len(bad) > 0
        F, ok = forensics.hypothesis_quartics(und, d["markers"], d["K"])
        w = forensics.ferrari_cancellation(F)
        stable_bad = [k for k in bad if w[k] >= forensics.W_UNSTABLE and
                      not forensics._oracle_p3p_moves(und, d["markers"], d["K"], k, orc)]
        assert stable_bad, "an injected tolerance error was explained away"
        n_checked += 1
        if n_checked >= 2:
            break
    assert n_checked >= 1


def test_streaming_submissions_are_bit_identical():
    """mpe_estimate_batch_device_submit / _collect: a stream of batches whose last voting launch scans the first
    sub-batch of the NEXT submission.  Records byte-identical to the joined one-call entry — with a hint that comes
    true and without one (hints that do not come true: test_streaming_false_hints_rewrites_and_changing_shapes) —,
    the announced batch's own stand-alone scan disappears, and the misuse paths are loud."""
    import torch
    B = 32768
    cfg, _, fa = _frames_on_device("C2", B, 8180)
    _, _, fb = _frames_on_device("C2", B, 8190)
    rows, cols = cfg["rows"], cfg["cols"]
    K, D = synth.camera_for(rows, cols)
    markers = np.asarray(cfg["markers"])
    dev = fa.device
    P = mpe.demo_params()
    h = mpe.Handle(0)
    stream = torch.cuda.Stream(device=dev)
    consumer = torch.cuda.Stream(device=dev)
    h.set_stream(stream.cuda_stream)
    nb = B * mpe.RESULT_DTYPE.itemsize

    def plain(fr):
        with torch.cuda.stream(stream):
            out = torch.zeros(nb, dtype=torch.uint8, device=dev)
            h.estimate_batch_device(fr.data_ptr(), B, rows, cols, markers, K, D, P, out.data_ptr())
        stream.synchronize()
        return out

    ref_a, ref_b = plain(fa), plain(fb)
    assert not torch.equal(ref_a, ref_b)
    h.set_profiling(True)
    outs = [torch.zeros(nb, dtype=torch.uint8, device=dev) for _ in range(4)]
    torch.cuda.synchronize()
    seq = [(fa, fb), (fb, fa), (fa, fa), (fa, None), (fb, fa)]   # (batch, announced next): 3rd hint is true, 4th has none
    refs = [ref_a, ref_b, ref_a, ref_a, ref_b]
    prefetched_expected = [False, True, True, True, False]
    for i, (fr, nxt) in enumerate(seq):
        o = outs[i % 4]
        with torch.cuda.stream(stream):
            h.estimate_batch_device_submit(fr.data_ptr(), B, rows, cols, markers, K, D, P, o.data_ptr(),
                                           nxt.data_ptr() if nxt is not None else 0, B)
            h.estimate_batch_device_collect(consumer.cuda_stream)
        consumer.synchronize()
        assert torch.equal(o, refs[i]), i
        stream.synchronize()
        scan0 = h.last_kernel_ms_sub(0)["scan"]
        if prefetched_expected[i]:
            assert scan0 < 0.05, (i, scan0)      # no stand-alone scan of the first sub-batch: it was prefetched
        else:
            assert scan0 > 0.3, (i, scan0)
    h.set_profiling(False)
    # two in flight, collected in order on the handle's own stream
    with torch.cuda.stream(stream):
        h.estimate_batch_device_submit(fa.data_ptr(), B, rows, cols, markers, K, D, P, outs[0].data_ptr(), fb.data_ptr(), B)
        h.estimate_batch_device_submit(fb.data_ptr(), B, rows, cols, markers, K, D, P, outs[1].data_ptr(), 0, 0)
        with pytest.raises(mpe.MpeError):
            h.estimate_batch_device_submit(fa.data_ptr(), B, rows, cols, markers, K, D, P, outs[2].data_ptr(), 0, 0)
        with pytest.raises(mpe.MpeError):
            h.estimate_batch_device(fa.data_ptr(), B, rows, cols, markers, K, D, P, outs[2].data_ptr())
        h.estimate_batch_device_collect(0)
        h.estimate_batch_device_collect(0)
        with pytest.raises(mpe.MpeError):
            h.estimate_batch_device_collect(0)
    stream.synchronize()
    assert torch.equal(outs[0], ref_a) and torch.equal(outs[1], ref_b)
    # other schedules stream as well (no side streams: the completion event is recorded on the caller's stream)
    for mode in (3, 4, 0):
        h.set_option("pipeline_mode", mode)
        with torch.cuda.stream(stream):
            h.estimate_batch_device_submit(fa.data_ptr(), B, rows, cols, markers, K, D, P, outs[0].data_ptr(), fb.data_ptr(), B)
            h.estimate_batch_device_collect(consumer.cuda_stream)
            h.estimate_batch_device_submit(fb.data_ptr(), B, rows, cols, markers, K, D, P, outs[1].data_ptr(), 0, 0)
            h.estimate_batch_device_collect(consumer.cuda_stream)
        consumer.synchronize()
        assert torch.equal(outs[0], ref_a) and torch.equal(outs[1], ref_b), mode
    assert h.get_option("streams_concurrent") in (0, 1)
    h.close()


def test_streaming_false_hints_rewrites_and_changing_shapes():
    """The streaming entry where round 3's test did not look:
      * a hint that does NOT come true — another buffer of the same size, then another size — leaves a scanned-ahead
        flag region behind that nobody must mistake for its own;
      * an announced buffer that is still being WRITTEN when the submission is made, with the event of that write
        handed over (mpe_stream_next_ready): the launches that read it wait;
      * an announced buffer that is REWRITTEN before its own submission, withdrawn with mpe_stream_drop_prefetch;
      * submissions of different sizes in flight (large - small - large): the regions of the detection / histogram
        buffers a submission overwrites are ordered behind the previous one's tails even though the two cut their
        batches differently.
    Records byte-identical to the joined one-call entry every time."""
    import torch
    B, Bs = 32768, 16384
    cfg, _, fa = _frames_on_device("C2", B, 9180)
    _, _, fb = _frames_on_device("C2", B, 9190)
    rows, cols = cfg["rows"], cfg["cols"]
    K, D = synth.camera_for(rows, cols)
    markers = np.asarray(cfg["markers"])
    dev = fa.device
    P = mpe.demo_params()
    h = mpe.Handle(0)
    stream = torch.cuda.Stream(device=dev)
    consumer = torch.cuda.Stream(device=dev)
    upload = torch.cuda.Stream(device=dev)
    h.set_stream(stream.cuda_stream)
    item = mpe.RESULT_DTYPE.itemsize

    def plain(fr, n):
        with torch.cuda.stream(stream):
            out = torch.zeros(n * item, dtype=torch.uint8, device=dev)
            h.estimate_batch_device(fr.data_ptr(), n, rows, cols, markers, K, D, P, out.data_ptr())
        stream.synchronize()
        return out

    ref_a, ref_b, ref_as, ref_bs = plain(fa, B), plain(fb, B), plain(fa, Bs), plain(fb, Bs)

    def run(seq):
        """seq of (frames, n, announced frames or None, announced n, reference)"""
        outs = []
        for fr, n, nxt, nn, ref in seq:
            with torch.cuda.stream(stream):
                # (the zero fill belongs on the submission's stream: filled on torch's current stream it races with the
                #  kernels that write the records — found when the library's side streams got queues of their own and
                #  the fill, queued behind the busy upload stream's kernels below, landed AFTER the tail had written)
                o = torch.zeros(n * item, dtype=torch.uint8, device=dev)
                h.estimate_batch_device_submit(fr.data_ptr(), n, rows, cols, markers, K, D, P, o.data_ptr(),
                                               nxt.data_ptr() if nxt is not None else 0, nn)
                h.estimate_batch_device_collect(consumer.cuda_stream)
            outs.append((o, ref))
        consumer.synchronize()
        stream.synchronize()
        for i, (o, ref) in enumerate(outs):
            assert torch.equal(o, ref), i

    # false hints: fb announced, fa comes (same size); fa announced at full size, the small fb comes; then back
    run([(fa, B, fb, B, ref_a), (fa, B, fa, B, ref_a), (fb, Bs, fa, B, ref_bs), (fb, B, None, 0, ref_b),
         (fa, Bs, fb, Bs, ref_as), (fb, Bs, None, 0, ref_bs)])
    # large - small - large, two in flight each time (no consumer wait in between)
    run([(fa, B, None, 0, ref_a), (fb, Bs, None, 0, ref_bs), (fb, B, None, 0, ref_b), (fa, Bs, None, 0, ref_as),
         (fa, B, None, 0, ref_a)])
    # the announced buffer is still being written: a copy into `nxt` on another stream, its event handed over
    nxt = torch.zeros_like(fb)
    spin = torch.zeros(64 << 20, dtype=torch.float32, device=dev)
    for rep in range(2):
        nxt.zero_()
        torch.cuda.synchronize()
        with torch.cuda.stream(upload):
            for _ in range(20):
                spin.mul_(1.0001)            # keeps the upload stream busy: the copy below lands late
            nxt.copy_(fb, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(upload)
        h.stream_next_ready(ev.cuda_event)
        run([(fa, B, nxt, B, ref_a), (nxt, B, None, 0, ref_b)])
    # the announced buffer is rewritten between the two submissions: withdrawn, scanned again
    buf = fb.clone()
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        o1 = torch.zeros(B * item, dtype=torch.uint8, device=dev)
        o2 = torch.zeros(B * item, dtype=torch.uint8, device=dev)
        h.estimate_batch_device_submit(fa.data_ptr(), B, rows, cols, markers, K, D, P, o1.data_ptr(), buf.data_ptr(), B)
        h.estimate_batch_device_collect(consumer.cuda_stream)
    consumer.synchronize()
    stream.synchronize()
    buf.copy_(fa)                        # (the prefetched flag words now describe pixels that are gone)
    torch.cuda.synchronize()
    h.stream_drop_prefetch()
    with torch.cuda.stream(stream):
        h.estimate_batch_device_submit(buf.data_ptr(), B, rows, cols, markers, K, D, P, o2.data_ptr(), 0, 0)
        h.estimate_batch_device_collect(consumer.cuda_stream)
    consumer.synchronize()
    stream.synchronize()
    assert torch.equal(o1, ref_a) and torch.equal(o2, ref_a)
    assert h.get_option("vote_fixup_overflow") == 0
    h.close()


def test_vote_events_time_the_fused_launches_inside_a_region():
    """Option "vote_events": one pair of timing events per voting launch that carries a scan, for the last N pipelined
    calls, read back as a mean (bench.py's roofline duration comes from it).  Joined calls of two sub-batches carry one
    scan each, streaming submissions with a hint that comes true carry two; records are unaffected."""
    import torch
    B = 32768
    cfg, _, fa = _frames_on_device("C2", B, 8380)
    rows, cols = cfg["rows"], cfg["cols"]
    K, D = synth.camera_for(rows, cols)
    markers = np.asarray(cfg["markers"])
    dev = fa.device
    P = mpe.demo_params()
    h = mpe.Handle(0)
    stream = torch.cuda.Stream(device=dev)
    h.set_stream(stream.cuda_stream)
    nb = B * mpe.RESULT_DTYPE.itemsize
    out = [torch.zeros(nb, dtype=torch.uint8, device=dev) for _ in range(2)]
    with torch.cuda.stream(stream):
        h.estimate_batch_device(fa.data_ptr(), B, rows, cols, markers, K, D, P, out[0].data_ptr())
    stream.synchronize()
    ref = out[0].clone()
    assert h.get_option("vote_launches") == 0              # (off by default)
    h.set_option("vote_events", 3)
    with torch.cuda.stream(stream):
        for _ in range(3):
            h.estimate_batch_device(fa.data_ptr(), B, rows, cols, markers, K, D, P, out[1].data_ptr())
    assert h.get_option("vote_launches") == 3              # one scan-carrying launch per joined call
    ns = h.get_option("vote_launch_ns_mean")
    assert 2e5 < ns < 2e7, ns                              # 16 384 frames voted + 70 % of 16 384 scanned: ~0.7 ms
    assert torch.equal(out[1], ref)
    h.set_option("vote_events", 2)                         # (resets)
    with torch.cuda.stream(stream):
        for i in range(3):
            h.estimate_batch_device_submit(fa.data_ptr(), B, rows, cols, markers, K, D, P, out[i & 1].data_ptr(),
                                           fa.data_ptr(), B)
            h.estimate_batch_device_collect(0)
    assert h.get_option("vote_launches") == 4              # the ring keeps the last two submissions, two launches each
    stream.synchronize()
    assert torch.equal(out[0], ref) and torch.equal(out[1], ref)
    h.set_option("vote_events", 0)
    assert h.get_option("vote_launches") == 0
    h.close()


def test_capacity_overrun_is_reported_alike_by_single_and_batch_replays(orc):
    """A frame with more blobs than MPE_MAX_DETECTIONS in the middle of a sequence: mpe_tracker_run_sequence and the
    lock-step batch replay both hand out a ZEROED record that carries the status code for that frame (and a zeroed
    info row), and both carry on with the sequence; a collect without a submission and a failed batch leave the
    handle usable."""
    seq = synth.make_sequence("C2", 8, seed=61)
    frames = seq["frames"].copy()
    rng = np.random.default_rng(8)
    spots = np.stack([rng.uniform(20, 730, 130), rng.uniform(20, 460, 130)], 1)
    # the FIRST frame (whole-image detection: the estimator is not initialised yet) shows ~120 blobs, more than
    # MPE_MAX_DETECTIONS; the ordinary sequence follows
    frames[0] = synth.render_frame(rng, spots, 480, 752)
    times = seq["times"]
    P = mpe.demo_params()
    h1, h2 = mpe.Handle(), mpe.Handle()
    try:
        t1 = mpe.Tracker(h1, seq["markers"], seq["K"], seq["D"], P)
        rec1, info1 = t1.run_sequence(frames, times)
        t2 = mpe.Tracker(h2, seq["markers"], seq["K"], seq["D"], P)
        rec2, info2 = mpe.tracker_run_sequences_batch([t2], [frames], times)
        rec2, info2 = rec2[0], info2[0]
        assert rec1["status"][0] == -10 and rec2["status"][0] == -10
        assert np.array_equal(rec1.view(np.uint8), rec2.view(np.uint8))
        assert np.array_equal(info1, info2)
        assert not rec1["T"][0].any() and not info1[0].any()
        assert (rec1["status"][1:] >= 0).all() and (rec1["status"][1:] == 0).sum() >= 5
        # collect with nothing in flight is an error, cancel is harmless, and the handle still works afterwards
        lib = mpe.load_library()
        dets = np.zeros(1, mpe.DETECTIONS_DTYPE)
        corr = np.zeros(2 * 16, np.uint32)
        res = np.zeros(1, mpe.RESULT_DTYPE)
        import ctypes as C
        assert lib.mpe_track_step_batch_collect(h2._h, C.c_void_p(dets.ctypes.data), C.c_void_p(corr.ctypes.data),
                                                C.c_void_p(res.ctypes.data)) < 0
        assert lib.mpe_track_step_batch_cancel(h2._h) == 0
        t2.reset()
        rec3, _ = mpe.tracker_run_sequences_batch([t2], [frames[:3]], times[:3])
        assert np.array_equal(rec3[0].view(np.uint8), rec1[:3].view(np.uint8))
        t1.close()
        t2.close()
    finally:
        h1.close()
        h2.close()


@pytest.mark.parametrize("encoding", ["bgr8", "rgb8", "bgra8", "rgba8", "mono16", "mono8"])
def test_convert_to_mono8_bit_exact(hip, orc, encoding):
    """mpe_convert_to_mono8 (= cv_bridge::toCvCopy(msg, MONO8), monocular_pose_estimator.cpp:147) against the oracle's
    restatement: every byte equal — host and device sources, odd widths (unaligned rows), both byte orders for
    mono16 — and the decoded frames give the same detections as the mono8 original."""
    import torch
    rng = np.random.default_rng(41)
    for rows, cols in ((48, 64), (37, 53), (5, 3)):
        n = 3
        if encoding == "mono16":
            src = rng.integers(0, 65536, (n, rows, cols)).astype(np.uint16)
        elif encoding == "mono8":
            src = rng.integers(0, 256, (n, rows, cols)).astype(np.uint8)
        else:
            src = rng.integers(0, 256, (n, rows, cols, 4 if encoding.endswith("a8") else 3)).astype(np.uint8)
        ref = np.stack([orc.convert_to_mono8(src[i], encoding) for i in range(n)])
        assert np.array_equal(hip.convert_to_mono8(src, encoding), ref), (encoding, rows, cols)
        raw = src.view(np.uint8).reshape(n, rows, cols, 2) if encoding == "mono16" else src   # bytes per pixel last
        dev = hip.convert_to_mono8(torch.from_numpy(raw).cuda(), encoding)
        assert np.array_equal(dev.cpu().numpy(), ref), (encoding, rows, cols, "device source")
        if encoding == "mono16":
            assert np.array_equal(hip.convert_to_mono8(src.byteswap(), encoding, big_endian=True), ref)
    # a colour / 16-bit rendering of a synthetic frame decodes to the frame the detector was tested on
    d = synth.make_frames("C2", 2, seed=7)
    f = d["frames"]
    if encoding == "mono16":
        enc_img = (f.astype(np.uint16) * 257)           # 8-bit value v as 16-bit v * 65535 / 255
    elif encoding == "mono8":
        enc_img = f
    else:
        ch = 4 if encoding.endswith("a8") else 3
        enc_img = np.repeat(f[..., None], ch, axis=3)   # gray in every channel: Y = v exactly (the weights sum to 2^14)
    got = hip.convert_to_mono8(np.ascontiguousarray(enc_img), encoding)
    assert np.array_equal(got, f)
    with pytest.raises(mpe.MpeError):
        lib = mpe.load_library()
        import ctypes as C
        rc = lib.mpe_convert_to_mono8(hip._h, C.c_void_p(f.ctypes.data), 0, 9, 0, 1, 4, 4, 4, 16, C.c_void_p(f.ctypes.data), 0)
        hip._check(rc, "mpe_convert_to_mono8")


def test_refinement_kernels_agree_bit_for_bit(orc):
    """k3b_refine (one lane per frame) and k3b_refine_group (16 lanes per frame, for small launches): every scalar of the
    Gauss-Newton iteration is computed by the same sequence of operations, so poses, covariances and iteration counts
    are IDENTICAL — on 512 C2 frames, 64 C3 frames at 2 px (8 correspondences) and the 4-LED demo rig —, and both
    agree with the oracle."""
    h = mpe.Handle()
    try:
        for config, n, tol in (("C2", 512, 5.0), ("C3", 64, 2.0), ("C1", 64, 5.0)):
            d = synth.make_frames(config, n, seed=515)
            P = mpe.demo_params(back_projection_pixel_tolerance=tol)
            out = {}
            for variant in (1, 2):
                h.set_option("refine_variant", variant)
                out[variant] = h.estimate_batch(d["frames"], d["markers"], d["K"], d["D"], P)
            a, b = out[1], out[2]
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), config
            assert (a["status"] == 0).sum() >= n // 2
            ref = orc.estimate_batch(d["frames"], d["markers"], d["K"], d["D"],
                                     orc.make_params(back_projection_pixel_tolerance=tol), n_threads=8)
            ok = (ref["status"] == 0) & (b["status"] == 0)
            assert np.array_equal(ref["status"], b["status"])
            assert np.abs(b["T"][ok] - ref["T"][ok]).max() < 1e-9
            assert np.abs(b["gn_iterations"][ok].astype(int) - ref["gn_iterations"][ok].astype(int)).max() <= 1
        h.set_option("refine_variant", 0)
        with pytest.raises(mpe.MpeError):
            h.set_option("refine_variant", 3)
    finally:
        h.close()


def test_multi_device_gather_on_the_device(hip):
    """mpe_estimate_batch_multi_device_gather: the shards' pose records end up in ONE device array on GPU 0.  With two
    or more GPUs they travel over RCCL (grouped ncclSend / ncclRecv); this box has one, so several handles share
    GPU 0 and the gather is a device-to-device copy — the RCCL leg is then SKIPPED, loudly.  Either way the records
    equal those of a single-handle call."""
    import torch
    d = synth.make_frames("C2", 41, seed=4321)
    P = mpe.demo_params()
    one = hip.estimate_batch(d["frames"], d["markers"], d["K"], d["D"], P)
    n_gpu = torch.cuda.device_count()
    for n_dev in (1, 2, 3):
        hs = [mpe.Handle(i % n_gpu) for i in range(n_dev)]
        try:
            shards = []
            for i in range(n_dev):
                lo, hi = mpe.shard_bounds(len(d["frames"]), i, n_dev)
                shards.append(torch.from_numpy(d["frames"][lo:hi].copy()).to("cuda:%d" % (i % n_gpu)))
            got, used_rccl = mpe.estimate_batch_multi_device_gather(hs, shards, d["markers"], d["K"], d["D"], P)
            assert got.tobytes() == one.tobytes(), n_dev
            assert used_rccl == (n_dev > 1 and n_gpu >= n_dev)
        finally:
            for h in hs:
                h.close()
    if n_gpu < 2:
        pytest.skip("ONE GPU visible: the multi-rank RCCL (xGMI) leg of mpe_estimate_batch_multi_device_gather was NOT "
                    "exercised here (its one-rank self-test is test_rccl_gather_self_test_on_one_device); it needs a box "
                    "with >= 2 GPUs")


def test_rccl_gather_self_test_on_one_device(hip):
    """Option "force_rccl_gather": the RCCL leg of mpe_estimate_batch_multi_device_gather with ONE handle — librccl is
    dlopen'ed and its symbols resolved, ncclCommInitAll makes a one-rank communicator, the handle's records go through
    a grouped ncclSend / ncclRecv to itself into the caller's device array — so that the code an 8-GPU node would run
    first has executed on hardware.  Records byte-identical to the plain call, twice (the communicator is re-used)."""
    import torch
    d = synth.make_frames("C2", 37, seed=808)
    P = mpe.demo_params()
    one = hip.estimate_batch(d["frames"], d["markers"], d["K"], d["D"], P)
    h = mpe.Handle(0)
    try:
        fr = [torch.from_numpy(d["frames"].copy()).to("cuda:0")]
        got, used = mpe.estimate_batch_multi_device_gather([h], fr, d["markers"], d["K"], d["D"], P)
        assert got.tobytes() == one.tobytes() and used is False          # one handle, not forced: no exchange at all
        h.set_option("force_rccl_gather", 1)
        assert h.get_option("force_rccl_gather") == 1
        for _ in range(2):
            got, used = mpe.estimate_batch_multi_device_gather([h], fr, d["markers"], d["K"], d["D"], P)
            assert used is True, "the RCCL path was not taken"
            assert got.tobytes() == one.tobytes()
    finally:
        h.close()
