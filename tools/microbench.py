"""Microbenchmarks of SURVEY.md 8(d)'s table on one MI355X (run on the GPU box):
  * whole-database mutual Hamming scan, Q in {1, 32, 500, 4000} current descriptors against L = 10000 x 64
    (HIP events around the kernel) -- pairs/s and algorithmic GB/s;
  * ragged / 45-row / 100-row databases at Q = 500;
  * one 500 x 500 mutual match and 2-NN match through the host-pointer entry points (includes copies + sync);
  * PnP-RANSAC, m in {10, 50, 200, 500} matches, 200 iterations, 40 % outliers, sigma 0.5 px (host call, sync
    included) and the reprojection scorer alone (200 hypotheses);
  * tick latency on one stream: median / p95 over >= 100 synchronous ticks after 10 warm-ups.
Prints one JSON object per line.  Not part of bench.py's contract; evidence for profiles/."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nclt_slam_project_amd.engine import Engine
from nclt_slam_project_amd import synth
import bench

PROF_DB_SCAN, PROF_PNP = 0, 3


def out(**kw):
    print(json.dumps(kw), flush=True)


def scan_cases(e):
    """Every case in both scheduling forms of the whole-database scan: `shared` = three generations of workgroups with a
    record quota (what a context runs when other contexts of the process work beside it: bench.py's 4 streams) and
    `alone` = one resident generation (reloc_set_exclusive, and the default of a process's only context)."""
    rng = np.random.default_rng(5)
    small = (1, 8, 32)
    for rows, L, Qs in (("fixed64", 10000, small + (500, 4000)), ("ragged", 10000, (500,)), (45, 10000, (500,)), (100, 10000, (500,)),
                        ("fixed64", 100000, small + (500,)), ("ragged", 100000, small + (500,)), (45, 100000, (500,))):
        desc, pts, off, poses = synth.descriptor_db(rng, L, rows)
        e.db_upload(desc, pts, off, poses)
        T = int(off[-1])
        cnt = e.dev_alloc(L * 4)
        for Q in Qs:
            cur = e.to_device(synth.random_descriptors(rng, Q))
            for alone in ((None,) if Q <= 64 else (False, True)):
                e.set_exclusive(alone)
                for _ in range(30):                   # also the clock-settling pre-roll
                    e.db_match_counts_dev(cur, Q, cnt)
                e.sync()
                e.profile_enable(True)
                n = 40
                for _ in range(n):
                    e.db_match_counts_dev(cur, Q, cnt)
                e.sync()
                ms, k = e.profile_get(PROF_DB_SCAN)
                e.profile_enable(False)
                us = ms / k * 1e3
                alg = 32 * T + 32 * Q + 4 * L
                out(case="db_scan", rows=str(rows), records=L, descriptors=T, Q=Q,
                    scheduling="-" if Q <= 64 else ("alone" if alone else "shared"), us=round(us, 1),
                    pairs_per_s=T * Q / (us * 1e-6), algorithmic_GBps=alg / (us * 1e-6) / 1e9, hbm_frac=alg / (us * 1e-6) / 8e12,
                    note="lane = teach row kernel (k_db_scan_rows)" if Q <= 64 else "")
            e.set_exclusive(None)
            e.dev_free(cur)
        e.dev_free(cnt)


def match_cases(e):
    rng = np.random.default_rng(6)
    q = synth.random_descriptors(rng, 500); t = synth.perturb_descriptors(rng, q, 0.08)[rng.permutation(500)]
    for name, fn in (("match_mutual_500x500", e.match_mutual), ("match_knn2_500x500", e.match_knn2)):
        for _ in range(5):
            fn(q, t)
        ts = []
        for _ in range(100):
            t0 = time.perf_counter(); fn(q, t); ts.append(time.perf_counter() - t0)
        out(case=name, median_us=round(float(np.median(ts)) * 1e6, 1), p95_us=round(float(np.percentile(ts, 95)) * 1e6, 1),
            note="host-pointer entry point: H2D + kernel + D2H + sync")


def pnp_cases(e):
    rng = np.random.default_rng(7)
    for m in (10, 50, 200, 500):
        obj, img, rvec, tvec, inl = synth.pnp_problem(rng, m, 0.4, 0.5)
        for _ in range(3):
            e.pnp_ransac(obj, img, seed=1)
        ts = []; ok = 0
        for i in range(50):
            t0 = time.perf_counter(); r = e.pnp_ransac(obj, img, seed=i); ts.append(time.perf_counter() - t0)
            ok += bool(r[0])
        Rt = np.tile(np.concatenate([synth.rodrigues(rvec).ravel(), tvec]), (200, 1))
        for _ in range(3):
            e.pnp_score(obj, img, Rt)
        ts2 = []
        for _ in range(50):
            t0 = time.perf_counter(); e.pnp_score(obj, img, Rt); ts2.append(time.perf_counter() - t0)
        out(case="pnp_ransac", m=m, iterations=200, outlier_ratio=0.4, noise_px=0.5, solved=ok, of=50,
            median_us=round(float(np.median(ts)) * 1e6, 1), p95_us=round(float(np.percentile(ts, 95)) * 1e6, 1),
            scorer_200hyp_median_us=round(float(np.median(ts2)) * 1e6, 1), note="host call: H2D + 3 kernels + D2H + sync")


def tick_latency(e):
    W, H = 640, 480
    frames, db, base_poses = bench.build_workload(e, 10000, "fixed64", 8)
    e.db_upload(*db)
    fd = [e.to_device(f) for f in frames]
    for i in range(10):
        e.tick_dev(fd[i % 8], W, H, base_poses[i % 8], False, True, i); e.tick_result()
    for mode, name in ((True, "tick_global_10k"), (False, "tick_local_10k")):
        ts = []
        for i in range(200):
            t0 = time.perf_counter()
            e.tick_dev(fd[i % 8], W, H, base_poses[i % 8], False, mode, i)
            e.tick_result()                      # D2H of the 96-byte result + sync
            ts.append(time.perf_counter() - t0)
        out(case=name, ticks=200, median_us=round(float(np.median(ts)) * 1e6, 1), p95_us=round(float(np.percentile(ts, 95)) * 1e6, 1),
            note="frame resident in HBM; one stream, synchronous: enqueue + kernels + result copy")
    ts = []
    for i in range(100):
        t0 = time.perf_counter(); e.tick(frames[i % 8], base_poses[i % 8], global_reloc=True, seed=i); ts.append(time.perf_counter() - t0)
    out(case="tick_global_10k_host_frame", ticks=100, median_us=round(float(np.median(ts)) * 1e6, 1),
        p95_us=round(float(np.percentile(ts, 95)) * 1e6, 1), note="frame in pageable host memory: PCIe copy of 0.92 MB included")


if __name__ == "__main__":
    e = Engine(0, 1280, 720, 4096)
    scan_cases(e)
    match_cases(e)
    pnp_cases(e)
    tick_latency(e)
