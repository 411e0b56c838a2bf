#!/usr/bin/env python3
"""Headline benchmark: relocalization frames/s at 640x480 against a 10k-landmark database.

    python bench.py --gpus 1 --steps 100 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A step = one batch of `--frames-per-step` synthetic 640x480 frames (32 distinct ones, cycled), each taken through the whole
hot path ON THE DEVICE: gray -> ORB(500) -> whole-database mutual-match scan (10 000 records x 64
descriptors, the reference's global relocalisation search, G:329-344) -> top-25 candidates ->
mutual match lists -> PnP-RANSAC(200) -> gates -> anchor pose; every frame's 96-byte result record is
copied back to (pinned) host memory.  Nothing is skipped or cached between frames.

`value` is measured with the frames already resident in HBM when the timed region starts (the contract of
the judged line).  `host_ingest` repeats the same K steps with every frame crossing PCIe first -- uploaded
from pinned host memory on the stream that processes it, SURVEY.md 3.1's "once per tick: upload one RGB
frame (921.6 KB)" -- and reports that rate beside it.  Per-step times (median / p95) come from events on
the streams, recorded inside the timed region without synchronising.

N > 1: frames are the independent units (a 10k-record database is 20 MB and fits every GPU), so each
rank runs its own frame stream against its own replica and there is no data-path collective;
per-GPU work is fixed ("weak").  value = frames of all ranks / max-over-ranks time.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline      dominant kernel (k_db_scan): algorithmic bytes / average launch duration measured
                with HIP events on the kernel's stream, against the 8 TB/s HBM peak; the kernel is
                VALU-bound by construction, so the VALU issue roofline is reported
                beside it
                roofline.in_config: the same launches timed while the S-stream timed configuration runs
  roofline_ragged  the same kernel on a ragged database (--rows ragged: N(60,25) rows per record) and on 45-row records
  roofline_small_q the few-query scan (k_db_scan_rows) on 100 000 records, Q = 1 / 8 / 32: the HBM-bound match shape
  roofline_matrix  the HBM-write-bound all-pairs u16 matrix kernel (BASELINE.json config 5 shape)
  latency       synchronous single-stream ticks back to back, and at the reference's 2 Hz cadence (0.5 s idle before each)
  cpu_baseline  the CPU oracle (a port) timed on a bounded sample of the same workload, frames in
                parallel on the host's cores
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H = 640, 480          # --size 720p switches to the 1280x720 characterisation shape (BASELINE.json config 3)
SEED = 20260501 + 1          # seed = 20260501 + config index


def build_workload(engine0, n_records, rows, n_frames):
    """frames (host), database arrays with one planted (matchable, PnP-solvable) record per frame, poses."""
    from nclt_slam_project_amd import synth
    rng = np.random.default_rng(SEED)
    frames = [synth.textured_frame(rng, W, H) for _ in range(n_frames)]
    desc, pts, off, poses = synth.descriptor_db(rng, n_records, rows)
    base_poses = []
    for f, img in enumerate(frames):
        feat = engine0.orb_detect_compute(engine0.gray(img), 500)
        r = int(rng.integers(0, n_records))
        n = int(off[r + 1] - off[r])
        sel = rng.choice(feat["n"], min(n, feat["n"]), replace=False)
        k = len(sel)
        # planted geometry: the record's 3-D points reproject onto the frame's keypoints under (R, t)
        rvec = rng.normal(size=3); rvec *= np.deg2rad(rng.uniform(1, 8)) / np.linalg.norm(rvec)
        tvec = rng.uniform(-0.5, 0.5, 3)
        R = synth.rodrigues(rvec)
        z = rng.uniform(2.0, 12.0, k)
        uv = feat["xy"][sel].astype(np.float64)
        pc = np.stack([(uv[:, 0] - 320.0) / 320.0 * z, (uv[:, 1] - 240.0) / 320.0 * z, z], 1)
        obj = (pc - tvec) @ R            # = R^T (pc - t)
        desc[off[r]:off[r] + k] = synth.perturb_descriptors(rng, feat["desc"][sel], 0.04)
        pts[off[r]:off[r] + k] = obj.astype(np.float32)
        base_poses.append(synth.base_pose(float(poses[r, 0]) + 0.5, 0.2, 1.0))
    return frames, (desc, pts, off, poses), base_poses


def opencv_baseline(frames, db, n_timed=100, n_warm=10):
    """BASELINE.md 2.2's opportunistic leg: if (and only if) OpenCV happens to be importable on the bench host, the calls the
    reference makes (M:305-306 cvtColor + ORB_create(500).detectAndCompute, M:327 BFMatcher(HAMMING, crossCheck=True).match
    against every record as G:329-344 does, M:342-346 solvePnPRansac on the top 25) are timed on the same frames and
    database -- 10 warm-up + >= 100 timed iterations, median / p95, cv2.setNumThreads stated.  Never required: returns None
    without OpenCV (this image has none)."""
    try:
        import cv2
    except ImportError:
        return None
    desc, pts, off, poses = db
    n_rec = len(off) - 1
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    cv2.setNumThreads(threads)
    orb = cv2.ORB_create(nfeatures=500)
    bf = cv2.BFMatcher(cv2.NORM_HAMMING, crossCheck=True)
    K = np.array([[320.0, 0, W / 2], [0, 320.0, H / 2], [0, 0, 1]], np.float32)
    dist0 = np.zeros((4, 1), np.float32)

    def one(img):
        t0 = time.perf_counter()
        gray = cv2.cvtColor(img, cv2.COLOR_BGR2GRAY)
        kp, d = orb.detectAndCompute(gray, None)
        if d is None:
            return time.perf_counter() - t0
        xy = np.array([k.pt for k in kp], np.float32)
        scored = []
        for r in range(n_rec):
            ms = bf.match(desc[off[r]:off[r + 1]], d)
            if len(ms) >= 10:
                scored.append((len(ms), r, ms))
        scored.sort(key=lambda t: -t[0])
        for _, r, ms in scored[:25]:
            obj = pts[off[r]:off[r + 1]][[m.queryIdx for m in ms]]
            im = xy[[m.trainIdx for m in ms]]
            try:
                cv2.solvePnPRansac(obj, im, K, dist0, iterationsCount=200, reprojectionError=3.0, flags=cv2.SOLVEPNP_ITERATIVE)
            except cv2.error:
                pass
        return time.perf_counter() - t0

    for i in range(n_warm):
        one(frames[i % len(frames)])
    ts = np.array([one(frames[i % len(frames)]) for i in range(n_timed)])
    return dict(value=1.0 / float(np.median(ts)), unit="frames/s", cores=threads, kind="reference",
                sample=f"{n_timed} frames after {n_warm} warm-up, one frame at a time, cv2.setNumThreads({threads}), OpenCV {cv2.__version__}: "
                       f"cvtColor + ORB(500) + crossCheck match against all {n_rec} records + solvePnPRansac on the top 25",
                frame_latency_ms=dict(median=float(np.median(ts)) * 1e3, p95=float(np.percentile(ts, 95)) * 1e3))


def cpu_baseline(frames, db, n_frames=112, n_warm=16):
    """The CPU oracle (oracle/, a port of the same pipeline) on the host cores of this machine.
    Timed build: -O3 -march=native (hardware popcnt), the scan evaluating each distance once; its outputs are first
    held equal to the strict checker build (-O2, literal two-pass matcher) on one frame / 300 records.  Frames are the
    parallel unit (one frame per thread: gray + ORB + whole-database scan + top-25 PnP, nothing sampled); `value` =
    frames / wall time on `cores` threads; per-frame latency median / p95 and the one-thread rate are reported too."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    desc, pts, off, poses = db
    n_rec = len(off) - 1
    try:
        threads = max(1, min(16, len(os.sched_getaffinity(0))))
    except AttributeError:
        threads = max(1, min(16, os.cpu_count() or 1))

    def one_frame(img, records=None):
        t0 = time.perf_counter()
        feat = O.orb_detect_compute(O.gray_u8(img), 500)
        t1 = time.perf_counter()
        nr = n_rec if records is None else records
        counts = O.db_match_counts(desc[: off[nr]], off[: nr + 1], feat["desc"])
        t2 = time.perf_counter()
        solved = []
        for r in O.topk_records(counts, 10, 25):
            qi, ti, dd = O.match_mutual(desc[off[r]:off[r + 1]], feat["desc"])
            if len(qi) >= 10:
                solved.append(O.pnp_ransac(pts[off[r]:off[r + 1]][qi], feat["xy"][ti], seed=1)[:3])
        t3 = time.perf_counter()
        return dict(feat=feat, counts=counts, solved=solved, t=(t1 - t0, t2 - t1, t3 - t2, t3 - t0))

    # strict build == timed build on a bounded sample (the strict build's scan is ~6x slower)
    O.select(False); O.set_threads(1); O.set_single_pass(False)
    ref = one_frame(frames[0], records=min(300, n_rec))
    O.select(True); O.set_threads(1); O.set_single_pass(True)
    chk = one_frame(frames[0], records=min(300, n_rec))
    assert (ref["feat"]["desc"] == chk["feat"]["desc"]).all() and (ref["feat"]["xy"] == chk["feat"]["xy"]).all()
    assert (ref["counts"] == chk["counts"]).all() and len(ref["solved"]) == len(chk["solved"])
    for a, b in zip(ref["solved"], chk["solved"]):
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    one_frame(frames[1 % len(frames)], records=min(300, n_rec))        # warm-up of the timed build
    t0 = time.perf_counter()
    single = one_frame(frames[0])
    t_single = time.perf_counter() - t0
    work = [frames[i % len(frames)] for i in range(n_frames)]
    # SURVEY.md 8(d) protocol: >= 10 warm-up iterations, >= 100 timed, median + p95 (whole frames, all of the database)
    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(one_frame, work[:n_warm]))
        t0 = time.perf_counter()
        out = list(ex.map(one_frame, work))
        wall = time.perf_counter() - t0
    O.select(False); O.set_single_pass(False)
    lat = np.array([o["t"][3] for o in out])
    st = np.array([o["t"][:3] for o in out]).mean(axis=0) * 1e3
    return dict(value=n_frames / wall, unit="frames/s", cores=threads, kind="port",
                sample=f"{n_frames} timed frames after {n_warm} warm-up ({len(frames)} distinct), one frame per thread on {threads} threads, whole pipeline per frame "
                       f"(gray + ORB + scan of all {n_rec} records + top-25 PnP-RANSAC), nothing sampled or scaled; per frame on a busy "
                       f"host: ORB {st[0]:.0f} ms, scan {st[1]:.0f} ms, PnP {st[2]:.0f} ms",
                frame_latency_ms=dict(median=float(np.median(lat)) * 1e3, p95=float(np.percentile(lat, 95)) * 1e3),
                single_core=dict(value=1.0 / t_single, cores=1, sample="one frame, one thread, idle host"),
                build="oracle/ -O3 -march=native -ffp-contract=off, scan evaluates each distance once; outputs asserted equal to the "
                      "strict checker build (-O2, literal two-pass matcher) on one frame / 300 records",
                cpu=_cpu_model(), host_cores=os.cpu_count())


def pmc_traffic(kernel):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/pmc_r4/summary.json, else pmc_r3 / pmc_r2 / pmc_r1e: FETCH_SIZE and
    WRITE_SIZE collected in separate passes, handled as MI355X_MICROARCH.md's HBM section prescribes); None if absent."""
    d = None
    for rnd in ("pmc_r4", "pmc_r3", "pmc_r2", "pmc_r1e"):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", rnd, "summary.json")))[kernel]
            break
        except (OSError, KeyError, ValueError):
            continue
    if d is None:
        return None
    # streaming 16-B/lane pattern (matrix kernel): FETCH_SIZE is doubled, WRITE_SIZE exact; the scan kernel reads through
    # scalar loads, for which the counter is uncalibrated: its raw value is used (lower bound)
    if "hbm_bytes_upper" not in d:
        return None
    # 16-byte-per-lane streaming reads (matrix kernel, few-query scan): FETCH_SIZE doubled; scalar-load reads (k_db_scan): raw
    return d["hbm_bytes_lower"] if kernel == "k_db_scan" else d["hbm_bytes_upper"]


def scan_case(e, L, rows, Qs, seed, forms=(False, True), n=40, preroll=30):
    """The whole-database mutual-match scan alone on one stream (reloc_db_match_counts_dev), HIP events around the kernel:
    one roofline object per (Q, scheduling form).  Q <= 64 runs the lane-per-teach-row kernel k_db_scan_rows (HBM-bound, no
    scheduling forms); larger Q the column-per-lane kernel k_db_scan, timed with the context marked `shared` and `alone`
    (reloc_set_exclusive).  Since round 4 a single launch runs ONE resident generation of workgroups in both (rounds 2-3: three
    generations with budgets when shared), so the two figures differ by run-to-run noise only; both are kept for continuity."""
    from nclt_slam_project_amd import synth
    rng = np.random.default_rng(seed)
    desc, pts, off, poses = synth.descriptor_db(rng, L, rows)
    e.db_upload(desc, pts, off, poses)
    T = int(off[-1])
    cnt = e.dev_alloc(L * 4)
    out = []
    for Q in Qs:
        cur = e.to_device(synth.random_descriptors(rng, Q))
        for alone in ((None,) if Q <= 64 else forms):
            e.set_exclusive(alone)
            for _ in range(preroll):                                      # clock settling, DESIGN.md section 4
                e.db_match_counts_dev(cur, Q, cnt)
            e.sync()
            e.profile_enable(True)
            for _ in range(n):
                e.db_match_counts_dev(cur, Q, cnt)
            e.sync()
            ms, k = e.profile_get(0)
            e.profile_enable(False)
            sec = ms / max(k, 1) * 1e-3
            alg = 32 * T + 32 * Q + 4 * L
            out.append(dict(kernel="k_db_scan_rows" if Q <= 64 else "k_db_scan", records=L, rows=str(rows), descriptors=T, Q=Q,
                            scheduling=None if Q <= 64 else ("alone" if alone else "shared"), bound="hbm", achieved=alg / sec / 1e9,
                            peak=8000.0, unit="GB/s", frac=alg / sec / 1e9 / 8000.0, traffic=None, avg_launch_us=sec * 1e6, launches=k,
                            algorithmic_bytes=alg, valu=dict(pairs_per_s=T * Q / sec, peak_pairs_per_s=2.99e12, frac=T * Q / sec / 2.99e12)))
        e.set_exclusive(None)
        e.dev_free(cur)
    e.dev_free(cnt)
    return out


def tick_latency_2hz(e, frames_dev, base_poses, result_row, n_ticks, idle_s, mode):
    """One synchronous tick every `idle_s` seconds -- the reference's cadence (timer at 2 Hz, M:76): every tick starts on a
    chip that has been idle for half a second, clocks down.  No pre-roll, nothing hidden.  The two deployment forms -- the
    context marked exclusive (reloc_set_exclusive: one scan generation, latency shapes) and not -- are INTERLEAVED tick by
    tick in one loop with the order alternating pair by pair (E N, N E, ...), `n_ticks` of each, so that they see the same
    chip state (round 3 timed 40 exclusive ticks and then 10 others: the two figures were from different minutes of the
    box).  The first pair is dropped."""
    e.tick_result_to(result_row)
    ts = {True: [], False: []}
    order = [True, False]
    for k in range(n_ticks):
        order += [True, False] if k % 2 == 0 else [False, True]
    for i, excl in enumerate(order):
        e.set_exclusive(excl)
        time.sleep(idle_s)
        result_row[72:76] = 255
        t0 = time.perf_counter()
        e.tick_dev(frames_dev[i % len(frames_dev)], W, H, base_poses[i % len(frames_dev)], False, mode, i)
        e.tick_wait()
        dt = time.perf_counter() - t0
        assert result_row[72] != 255, "tick result record did not arrive"
        if i >= 2:
            ts[excl].append(dt)
    e.set_exclusive(False)

    def stats(x):
        x = np.array(x) * 1e6
        return dict(median=float(np.median(x)), p95=float(np.percentile(x, 95)), min=float(x.min()), ticks=len(x), idle_s=idle_s)
    return stats(ts[True]), stats(ts[False])


def bench_sharded(args, rank, world, local_rank, dist, torch):
    """Strong scaling of the whole-database search (BASELINE config 4): every rank sees every frame, owns 1/world of the
    records; frames go through in batches of 8 with two small all-gathers per batch."""
    from nclt_slam_project_amd.engine import Engine
    from nclt_slam_project_amd.sharded import DeviceShardedRelocalizer, HipShard, ShardedRelocalizer
    e = Engine(local_rank, W, H, 2048)
    frames, db, base_poses = build_workload(e, args.records, args.rows, 8)
    BATCH = 8
    DEPTH = int(os.environ.get("BENCH_SHARD_DEPTH", "16"))        # batches the host runs ahead of the device, on four streams
                                                                  # (a multiple of 4; 8: 5 500-6 360 frames/s from run to run,
                                                                  # 16 / 32: 6 370, profiles/r3_shard_repeat.log)
    dev = None if (dist is None or args.backend != "nccl") else torch.device("cuda", local_rank)
    device_path = world == 1 or args.backend == "nccl"
    shard = HipShard(e, *db, rank=rank, world=world, w=W, h=H, n_slots=1 if device_path else BATCH)
    frames_dev = [e.to_device(f) for f in frames]
    B = max(BATCH, args.frames_per_step // BATCH * BATCH)
    if device_path:
        # exchange resident in HBM: top-k lists, merge, candidate hand-over and result records never visit the host inside a
        # batch; DEPTH batches in flight, each on its own stream
        forced = bool(args.force_dist and dist is not None)
        if forced and args.backend != "nccl":
            raise SystemExit("bench.py: --force-dist routes device tensors through the collective: it needs --backend nccl")
        sr = DeviceShardedRelocalizer(shard, rank, world, torch.device("cuda", local_rank), batch=BATCH, depth=DEPTH,
                                      force_collective=forced)

        flight = []                                                       # batches in flight, ACROSS steps: a step that drained its
                                                                          # own batches would empty the device once per step (round 3,
                                                                          # first form: 5336 frames/s at 10k records; the frames of a
                                                                          # stream do not wait for a step boundary)
        last_out = [None]

        def step(s0):
            for i in range(0, B, BATCH):
                flight.append(sr.submit(frames_dev, base_poses, [s0 + i + j for j in range(BATCH)]))
                if len(flight) >= DEPTH:
                    last_out[0] = sr.result(flight.pop(0))[-1]            # every batch's results are collected, the oldest first
            return last_out[0]

        def drain():
            while flight:
                last_out[0] = sr.result(flight.pop(0))[-1]
            return last_out[0]
    else:
        hr = ShardedRelocalizer(shard, shard.base, rank, world, device=dev)

        def step(s0):
            out = None
            for i in range(0, B, BATCH):
                out = hr.tick_batch(frames_dev, base_poses, seeds=[s0 + i + j for j in range(BATCH)])[-1]
            return out

        def drain():
            return None

    for w_ in range(args.warmup + 3):
        step(0)
    drain()
    e.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    last = None
    for k in range(args.steps):
        last = step(k * B)
    last = drain() or last                                                # the timed region ends when every frame's result is on the host
    e.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if device_path:
        sr.close()
    shard.close()
    e.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        L = len(db[2]) - 1
        print(json.dumps({
            "metric": "relocalization frames/sec, database sharded by record", "value": B * args.steps / elapsed, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{W}x{H} frames, {L}-record DB split over {world} rank(s), batches of {BATCH} frames ({DEPTH} in flight): "
                                   f"ORB on every rank, ONE shard-scan launch per batch, one all-gather of the {BATCH} top-25 lists, PnP on "
                                   f"the owners, one all-gather of the {BATCH} results",
                       "frames_per_step": B, "records": L, "shard_records": shard.n_records,
                       "collective": ("none (single rank: the exchange is a device-to-device copy)" if dist is None else
                                      f"torch.distributed {args.backend} all_gather_into_tensor x 2 per batch, world {world}"
                                      + (" (forced at world 1)" if args.force_dist and world == 1 else ""))},
            "last_outcome": int(last["outcome"]), "last_inliers": int(last["n_inliers"])}))


MATRIX_PREROLL = 400      # untimed launches (~70 ms) before the matrix kernel is timed
PREROLL_STEPS = 20        # untimed steps (~0.2 s) ahead of the W warm-up steps of the frame benchmark: within the first ~100 ms
                          # after an idle period the chip changes its clock state once, and a step that meets that change takes
                          # 50 ms instead of 11 (seen in one of two runs with 3 warm-up steps, in none with 30)


def bench_matrix(args, rank, world, local_rank, dist, torch):
    """Strong scaling of config 5: rank r writes rows [r F / N, (r + 1) F / N) of the F x K matrix into its own HBM; the K
    keyframe descriptors (640 KB) are replicated, nothing is exchanged."""
    from nclt_slam_project_amd.engine import Engine
    e = Engine(local_rank, 640, 480, 2048)
    F = K = 20000
    rng = np.random.default_rng(SEED + 4)
    A = rng.integers(0, 256, (F, 32), dtype=np.uint8)
    Bm = rng.integers(0, 256, (K, 32), dtype=np.uint8)
    from nclt_slam_project_amd.sharded import matrix_row_block
    r0, r1 = matrix_row_block(F, rank, world)
    a = e.to_device(A[r0:r1]); b = e.to_device(Bm)
    out = e.dev_alloc((r1 - r0) * K * 2)
    # the chip's clock governor needs ~40 ms of continuous load to settle (the first launches after an idle period run at
    # boost clock, the next ~30 ms 10-40 % slower, then steady): pre-roll, then the W warm-up steps
    for _ in range(MATRIX_PREROLL):
        e.hamming_matrix_dev(a, r1 - r0, b, K, out)
    for _ in range(max(args.warmup, 1)):
        e.hamming_matrix_dev(a, r1 - r0, b, K, out)
    e.sync()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        e.hamming_matrix_dev(a, r1 - r0, b, K, out)
    e.sync()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    e.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        nbytes = 32 * (F + K) + 2 * F * K
        print(json.dumps({
            "metric": "Hamming distance matrix, HBM GB/s (algorithmic bytes of the whole matrix per second)",
            "value": nbytes * args.steps / elapsed / 1e9, "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{F} x {K} 256-bit descriptors -> u16 distances, row blocks of {F // world} per rank, no collective"},
            "roofline": {"kernel": "k_hamming_matrix", "bound": "hbm", "achieved": nbytes * args.steps / elapsed / 1e9 / world,
                         "peak": 8000.0, "unit": "GB/s", "frac": nbytes * args.steps / elapsed / 1e9 / world / 8000.0,
                         "traffic": None, "note": f"per GPU, wall clock over the timed region (launch gaps included), after a clock-settling "
                                                  f"pre-roll of {MATRIX_PREROLL} untimed launches"}}))


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def launch_plan(args, argv, env):
    """What `python bench.py --gpus N` has to do before anything touches the GPU (BASELINE.json north_star: frames/s at 1, 2, 4
    and 8 GPUs; the driver's N > 1 line is `python -m torch.distributed.run ... bench.py --gpus N`, a bare `python bench.py
    --gpus N` must measure N GPUs as well, not one):
        ("run",)          this process is a rank (WORLD_SIZE == --gpus) or the single-GPU run
        ("spawn", cmd)    --gpus N > 1 outside a launcher: start N ranks as a FRESH child process (never an exec: this
                          interpreter may already hold a GPU context in other callers), relay its output, exit with its code
        ("error", text)   WORLD_SIZE and --gpus disagree: refuse instead of printing a line for the wrong GPU count"""
    world = env.get("WORLD_SIZE")
    if world is None:
        if args.gpus <= 1:
            return ("run",)
        import socket
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
        return ("spawn", cmd)
    try:
        w = int(world)
    except ValueError:
        return ("error", f"WORLD_SIZE={world!r} is not a number")
    if w != args.gpus:
        return ("error", f"--gpus {args.gpus} but the launcher started WORLD_SIZE={w} rank(s): the line would carry the wrong "
                         f"GPU count; launch with --nproc-per-node {args.gpus} or pass --gpus {w}")
    return ("run",)


def init_dist(args, torch, rank, world, local_rank):
    """torch.distributed group of the run, or None.  backend "nccl" IS RCCL on ROCm.  --force-dist initialises the group at
    world size 1 too (one rank, one GPU), so that communicator creation, barrier, all_reduce and all_gather_into_tensor run over
    RCCL beside this library's HIP runtime on a one-GPU box."""
    if world == 1 and not args.force_dist:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        import socket
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
    if args.backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    return dist


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames-per-step", type=int, default=64)
    ap.add_argument("--records", type=int, default=10000)
    ap.add_argument("--rows", default="fixed64", help="fixed64 | ragged | <int>")
    ap.add_argument("--streams", type=int, default=4)
    ap.add_argument("--batch", type=int, default=1, help="frames per whole-database scan launch on each stream (1..8)")
    ap.add_argument("--size", default="480p", choices=["480p", "720p"])
    ap.add_argument("--shard-db", action="store_true",
                    help="BASELINE config 4 shape: ONE frame stream, the database split by record over the ranks, per frame an\n"
                         "all-gather of the per-shard top-25 (not the judged default; reports its own JSON line)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend; gloo + --rehearse runs N ranks on one GPU")
    ap.add_argument("--rehearse", action="store_true", help="developer rehearsal: every rank uses device 0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-matrix", action="store_true")
    ap.add_argument("--no-ingest", action="store_true", help="skip the second timed run with frames uploaded from host memory")
    ap.add_argument("--distinct-frames", type=int, default=32, help="distinct synthetic frames cycled through (each with its own planted record)")
    ap.add_argument("--no-2hz", action="store_true", help="skip the tick latency at the reference's 2 Hz cadence (~50 s of mostly idle time)")
    ap.add_argument("--ticks-2hz", type=int, default=24, help="ticks of EACH form (exclusive / not), interleaved, per mode")
    ap.add_argument("--no-extra-scans", action="store_true", help="skip roofline_ragged / roofline_small_q")
    ap.add_argument("--matrix-only", action="store_true",
                    help="BASELINE config 5 shape: the 20000 x 20000 u16 Hamming matrix, row blocks split over the ranks, no\n"
                         "reduction (not the judged default; reports its own JSON line)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the torch.distributed group at world size 1 too and route barrier / all_reduce / both\n"
                         "all-gathers of --shard-db through it (RCCL with --backend nccl) instead of the single-rank shortcut")
    argv = list(sys.argv[1:] if argv is None else argv)
    args = ap.parse_args(argv)
    plan = launch_plan(args, argv, os.environ)
    if plan[0] == "error":
        raise SystemExit("bench.py: " + plan[1])
    if plan[0] == "spawn":
        import subprocess
        r = subprocess.run(plan[1], cwd=ROOT)          # the ranks print on this process's stdout / stderr
        raise SystemExit(r.returncode)
    global W, H
    if args.size == "720p":
        W, H = 1280, 720

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP library has no CPU fallback")
    if args.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = init_dist(args, torch, rank, world, local_rank)

    from nclt_slam_project_amd.engine import Engine

    if args.shard_db:
        return bench_sharded(args, rank, world, local_rank, dist, torch)
    if args.matrix_only:
        return bench_matrix(args, rank, world, local_rank, dist, torch)

    # args.streams streams; on each, args.batch contexts (one per frame of a batch) that share the stream, so that the
    # whole-database scans of a batch are ONE launch (reloc_tick_batch_dev).  All contexts scan one resident database.
    S, NB = args.streams, max(1, min(8, args.batch))
    groups = [[Engine(local_rank, W, H, 2048) for _ in range(NB)] for _ in range(S)]
    engines = [e for g in groups for e in g]
    n_distinct = args.distinct_frames
    frames, db, base_poses = build_workload(engines[0], args.records, args.rows, n_distinct)
    engines[0].db_upload(*db)
    for e in engines[1:]:
        e.db_share(engines[0])                         # all contexts scan ONE resident copy of the database
    # every ctx enqueues on a torch-made stream so that torch events can time the steps without synchronising
    tstreams = [torch.cuda.Stream(device=local_rank) for _ in range(S)]
    for g, ts in zip(groups, tstreams):
        for e in g:
            e.set_stream(ts.cuda_stream)
    e0 = engines[0]
    frames_dev = [e0.to_device(f) for f in frames]                       # resident frames (judged mode)
    frames_pin = []
    for f in frames:                                                      # host frames in pinned memory (ingest mode)
        p = e0.pinned(f.shape, np.uint8)
        p[...] = f
        frames_pin.append(p)
    stage = {id(e): e.dev_alloc(W * H * 3) for e in engines}              # per-context upload target
    B = args.frames_per_step // NB * NB
    results = e0.pinned((B, 96), np.uint8)                                # one result record per frame of a step

    def step(seed0, ingest):
        for c in range(B // NB):                                          # one batch of NB frames per call
            g = groups[c % S]
            fs = [(c * NB + j) % n_distinct for j in range(NB)]
            if ingest:
                for j, e in enumerate(g):
                    e.h2d_async(stage[id(e)], frames_pin[fs[j]])          # 921.6 KB over PCIe, same stream as its tick
                imgs = [stage[id(e)] for e in g]
            else:
                imgs = [frames_dev[f] for f in fs]
            for j, e in enumerate(g):
                e.tick_result_to(results[c * NB + j])                     # every frame's result record lands in host memory:
                                                                          # the tick's last kernel writes it over PCIe itself
            if NB == 1:
                g[0].tick_dev(imgs[0], W, H, base_poses[fs[0]], order_rgb=False, global_reloc=True, seed=seed0 + c)
            else:
                Engine.tick_batch_dev(g, imgs, W, H, [base_poses[f] for f in fs], global_reloc=True,
                                      seeds=[seed0 + c * NB + j for j in range(NB)])

    def sync_all():
        for e in engines:
            e.sync()
        torch.cuda.synchronize()

    def timed(ingest):
        # the event objects are made BEFORE the warm-up: creating 4 x steps events takes ~40 ms, and a GPU left idle that long
        # between warm-up and the timed region starts the timed region with its clocks down (first step 5x longer)
        ev0 = torch.cuda.Event(enable_timing=True)
        marks = [[torch.cuda.Event(enable_timing=True) for _ in tstreams] for _ in range(args.steps)]
        for w in range(PREROLL_STEPS + args.warmup):                      # pre-roll (clock governor, see PREROLL_STEPS), then W warm-up steps
            step(w * B, ingest)
        sync_all()
        if dist is not None:
            dist.barrier()
        sync_all()
        ev0.record(tstreams[0])
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(k * B, ingest)
            for ts, ev in zip(tstreams, marks[k]):
                ev.record(ts)
        sync_all()
        if dist is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        ends = np.array([max(ev0.elapsed_time(ev) for ev in row) for row in marks])      # ms since ev0
        per_step = np.diff(np.concatenate([[0.0], ends]))
        rec = np.array(results).view(np.int32).reshape(B, 24)
        outcomes = rec[:, 18]                                             # TickResult.outcome of the last step's frames
        return elapsed, per_step, dict(published_in_last_step=int((outcomes == 0).sum()), frames_in_step=B,
                                       last_inliers=int(rec[B - 1, 16]))

    elapsed, per_step, outcomes = timed(False)
    total_frames = world * B * args.steps
    ingest = None
    if not args.no_ingest:
        el_i, ps_i, oc_i = timed(True)
        ingest = dict(value=total_frames / el_i, unit="frames/s", ms_per_step=el_i / args.steps * 1e3,
                      step_ms=dict(median=float(np.median(ps_i)), p95=float(np.percentile(ps_i, 95))),
                      published_in_last_step=oc_i["published_in_last_step"],
                      how="every frame uploaded from pinned host memory (hipMemcpyAsync on the stream of its tick, 921.6 KB "
                          "per frame) inside the timed region; result records written to host memory per frame as in the judged mode")

    # ---- the same kernels INSIDE the timed configuration: HIP events around every context's scan / ORB / PnP launches while
    # all S streams run the step loop (16 further steps, untimed); the single-stream figures below are their unloaded times
    in_config = None
    if rank == 0:
        for w in range(3):
            step(w * B, False)
        sync_all()
        for e in engines:
            e.profile_enable(True)
        n_cfg_steps = max(1, min(16, 250 * len(engines) // B))            # <= 250 launches per context: the event ring never waits
        t0 = time.perf_counter()
        for k in range(n_cfg_steps):
            step(k * B, False)
        sync_all()
        cfg_elapsed = time.perf_counter() - t0
        acc = {0: [0.0, 0], 2: [0.0, 0], 3: [0.0, 0]}
        for e in engines:
            for which in acc:
                ms, n = e.profile_get(which)
                acc[which][0] += ms
                acc[which][1] += n
            e.profile_enable(False)
        in_config = dict(streams=S, frames_per_scan_launch=NB, steps=n_cfg_steps,
                         frames_per_s_with_events=B * n_cfg_steps / cfg_elapsed,
                         scan_avg_launch_us=acc[0][0] / max(acc[0][1], 1) * 1e3, scan_launches=acc[0][1],
                         orb_us=acc[2][0] / max(acc[2][1], 1) * 1e3, pnp_us=acc[3][0] / max(acc[3][1], 1) * 1e3,
                         note="kernel durations while all streams run (events on each context's stream); a scan lasts longer here than "
                              "alone because it shares the VALU with the other streams' kernels and yields to them (wave priority); "
                              "several scans are resident at once, so frames/s x scan time exceeds 1")

    result = None
    if rank == 0:
        e = engines[0]
        # ---- dominant kernel: HIP events around k_db_scan on its own stream, live
        e.profile_enable(True)
        n_prof = 40
        for i in range(n_prof):
            if NB == 1:
                e.tick_dev(frames_dev[i % n_distinct], W, H, base_poses[i % n_distinct], False, True, i)
            else:      # the launch the timed region ran: one scan for NB frames
                fs = [(i * NB + j) % n_distinct for j in range(NB)]
                Engine.tick_batch_dev(groups[0], [frames_dev[f] for f in fs], W, H, [base_poses[f] for f in fs], global_reloc=True,
                                      seeds=[i * NB + j for j in range(NB)])
        e.sync()
        scan_ms, scan_n = e.profile_get(0)
        orb_ms, orb_n = e.profile_get(2)
        pnp_ms, pnp_n = e.profile_get(3)
        e.profile_enable(False)
        # single-stream synchronous tick latency (enqueue + kernels + result record in host memory, waited for with reloc_tick_wait), global and local candidate
        # search, with the context told that it is alone on the GPU (reloc_set_exclusive; the timed 4-stream runs above are not)
        lat = {}
        for mode, name in ((1, "tick_global"), (0, "tick_local")):
            ts_ = []
            e.tick_result_to(results[0])
            e.set_exclusive(True)                                         # a synchronous single-stream loop: the deployment this hint is for
            for i in range(120):
                results[0, 72:76] = 255                                    # outcome field: overwritten by the tick
                t0 = time.perf_counter()
                e.tick_dev(frames_dev[i % n_distinct], W, H, base_poses[i % n_distinct], False, mode, i)
                e.tick_wait()                                             # the result record is in host memory now (reloc_tick_wait)
                ts_.append(time.perf_counter() - t0)
                assert results[0, 72] != 255, "tick result record did not arrive"
            e.set_exclusive(False)
            ts_ = np.array(ts_[20:]) * 1e6
            lat[name + "_us"] = dict(median=float(np.median(ts_)), p95=float(np.percentile(ts_, 95)))
        desc, pts, off, poses = db
        T, L, Q = int(off[-1]), len(off) - 1, 500
        alg_bytes = NB * (32 * T + 32 * Q + 4 * L)        # per frame: database once, queries once, one count per record; NB frames per launch
        scan_s = scan_ms / max(scan_n, 1) * 1e-3
        pairs = NB * T * Q
        # VALU ceilings measured on MI355X (tools/ubench_valu2.hip, profiles/r2_ubench_valu2.log): the bare distance,
        # 8 x (v_xor_b32 s,v ; accumulating v_bcnt_u32_b32), reaches 2.99 T pairs/s at 8 waves/SIMD (2.91 T at the
        # kernel's 4); with the kernel's argmin bookkeeping in the stream (shl16 + or + 3 min16 per pair) 2.21 T at 4
        valu_peak_pairs = 2.99e12
        roofline = dict(kernel="k_db_scan" if NB == 1 else "k_db_scan_batch", frames_per_launch=NB, bound="hbm", achieved=alg_bytes / scan_s / 1e9, peak=8000.0, unit="GB/s",
                        frac=alg_bytes / scan_s / 1e9 / 8000.0, traffic=(pmc_traffic("k_db_scan") or 0) * NB or None,
                        avg_launch_us=scan_s * 1e6, launches=scan_n, algorithmic_bytes=alg_bytes,
                        note="VALU-bound by construction: 250 int-op/B at Q=500 (SURVEY.md 8d); HBM fraction cannot exceed ~2 %",
                        valu=dict(pairs_per_s=pairs / scan_s, peak_pairs_per_s=valu_peak_pairs,
                                  frac=pairs / scan_s / valu_peak_pairs,
                                  lane_ops_per_s=16.0 * pairs / scan_s, spec_lane_ops_per_s=78.6e12,
                                  frac_of_spec_issue=16.0 * pairs / scan_s / 78.6e12,
                                  basis="measured chip ceiling of the bare 256-bit distance, 8 v_xor_b32 (full rate) + 8 accumulating "
                                        "v_bcnt_u32_b32 (half rate) per pair (tools/ubench_valu2.hip); spec issue = 256 CU x 4 SIMD-32 x "
                                        "2.4 GHz counts every instruction at full rate"))
        stage_us = dict(orb=orb_ms / max(orb_n, 1) * 1e3, db_scan_per_frame=scan_s * 1e6 / NB, pnp=pnp_ms / max(pnp_n, 1) * 1e3)
        if in_config:
            ic_s = in_config["scan_avg_launch_us"] * 1e-6
            roofline["in_config"] = dict(in_config, achieved=alg_bytes / ic_s / 1e9, frac=alg_bytes / ic_s / 1e9 / 8000.0,
                                         pairs_per_s_per_scan=pairs / ic_s,
                                         scans_resident=in_config["frames_per_s_with_events"] / NB * ic_s,
                                         aggregate_pairs_per_s=total_frames / elapsed * T * Q,
                                         aggregate_valu_frac=total_frames / elapsed * T * Q / valu_peak_pairs)
            roofline["note_single_stream"] = ("avg_launch_us / stage_us: one idle stream after the timed region (unloaded kernel times); "
                                              "in_config: the same launches while the timed configuration runs")
        # ---- 2 Hz cadence: a tick every 0.5 s, as the reference's timer fires (M:76), each on an idle chip
        if not args.no_2hz:
            for mode, name in ((1, "tick_global"), (0, "tick_local")):
                lat[name + "_2hz_us"], lat[name + "_2hz_not_exclusive_us"] = tick_latency_2hz(
                    e, frames_dev, base_poses, results[0], args.ticks_2hz, 0.5, mode)
            lat["note_2hz"] = ("exclusive and non-exclusive ticks interleaved in one loop, alternating order, same sample size; a local tick "
                               "launches the same kernel shapes in both forms (reloc_tick.hip: latency shapes for LOCAL ticks)")
        # ---- the scan on the record sizes of real teach databases, and the few-query shape that IS HBM-bound
        roofline_ragged = roofline_small_q = None
        if not args.no_extra_scans:
            ex = Engine(local_rank, W, H, 2048)
            rag = scan_case(ex, args.records, "ragged", (500,), SEED + 10)
            r45 = scan_case(ex, args.records, 45, (500,), SEED + 11)
            pick = lambda rows, form: next(r for r in rows if r["scheduling"] == form)
            roofline_ragged = dict(pick(rag, "shared"), workload=f"--rows ragged: {args.records} records of clip(round(N(60,25)),30,500) rows, Q = 500",
                                   alone=pick(rag, "alone"), rows45=dict(shared=pick(r45, "shared"), alone=pick(r45, "alone")),
                                   note="teach databases of the reference are ragged, ~45-100 rows per record (routes/01_road/teach/README.md:57,75); "
                                        "VALU-bound like the headline kernel: compare valu.pairs_per_s")
            sq = scan_case(ex, 100000, "fixed64", (1, 8, 32), SEED + 12)
            for r_ in sq:
                r_["traffic"] = pmc_traffic({1: "k_db_scan_rows_q1", 8: "k_db_scan_rows_q8", 32: "k_db_scan_rows_q32"}.get(r_["Q"], "-"))
            roofline_small_q = dict(sq[0], workload="100000 records x 64 rows (205 MB), Q = 1: the variant-G scan shape with few current descriptors "
                                                    "(G:329-344), the HBM-bound match shape", by_Q={str(r["Q"]): r for r in sq})
            ex.close()
        roofline_matrix = None
        if not args.no_matrix:
            F = K = 20000
            rng = np.random.default_rng(SEED + 4)
            a = e.to_device(rng.integers(0, 256, (F, 32), dtype=np.uint8))
            b = e.to_device(rng.integers(0, 256, (K, 32), dtype=np.uint8))
            out = e.dev_alloc(F * K * 2)
            for _ in range(MATRIX_PREROLL):                               # clock governor settles after ~40 ms of load, see bench_matrix
                e.hamming_matrix_dev(a, F, b, K, out)
            e.sync()
            e.profile_enable(True)
            for _ in range(100):
                e.hamming_matrix_dev(a, F, b, K, out)
            e.sync()
            m_ms, m_n = e.profile_get(1)
            e.profile_enable(False)
            mb = 32 * (F + K) + 2 * F * K
            ms = m_ms / max(m_n, 1) * 1e-3
            roofline_matrix = dict(kernel="k_hamming_matrix", shape=[F, K], bound="hbm", achieved=mb / ms / 1e9, peak=8000.0,
                                   unit="GB/s", frac=mb / ms / 1e9 / 8000.0, traffic=pmc_traffic("k_hamming_matrix"), avg_launch_us=ms * 1e6,
                                   launches=m_n, algorithmic_bytes=mb,
                                   valu=dict(pairs_per_s=F * K / ms, peak_pairs_per_s=2.99e12, frac=F * K / ms / 2.99e12))
            for p in (a, b, out):
                e.dev_free(p)
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(frames, db)
            cpu["opencv"] = opencv_baseline(frames, db)       # None unless OpenCV is importable on this host
        result = {
            "metric": "relocalization frames/sec @640x480, 10k-landmark DB; Hamming-match HBM GB/s",
            "value": total_frames / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{W}x{H} BGR frames, {L}-record landmark DB ({T} descriptors, rows={args.rows}), "
                                   f"global relocalization tick per frame: ORB(500) + whole-DB mutual Hamming scan + "
                                   f"top-25 PnP-RANSAC(200); frames resident in HBM, every frame's 96-byte result record written to pinned host memory by the tick's last kernel",
                       "frames_per_step": B, "distinct_frames": n_distinct, "preroll_steps": PREROLL_STEPS, "streams": args.streams, "frames_per_scan_launch": NB, "records": L, "descriptors": T,
                       "parallelism": "frames sharded across ranks, database replicated, no collective",
                       "process_group": None if dist is None else f"{args.backend}, world {world}: barrier + all_reduce(MAX) of the elapsed time"},
            "step_ms": dict(median=float(np.median(per_step)), p95=float(np.percentile(per_step, 95)), max=float(per_step.max()),
                            first=[round(float(x), 2) for x in per_step[:6]]),
            "host_ingest": ingest, "latency": lat,
            "roofline": roofline, "roofline_ragged": roofline_ragged, "roofline_small_q": roofline_small_q,
            "roofline_matrix": roofline_matrix, "cpu_baseline": cpu,
            "stage_us": stage_us, "hamming_match_GBps": roofline["achieved"], **outcomes,
        }
    for e in engines:
        e.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
