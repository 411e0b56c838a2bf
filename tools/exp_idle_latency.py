"""Developer experiment: tick latency as a function of the idle time in front of the tick (the reference ticks at 2 Hz,
M:76, so every tick starts on a chip that has been idle for ~0.5 s), with the device-side stage times beside the wall time,
and the effect of keep-warm launches during the idle period.
    python tools/exp_idle_latency.py > gpurun_out/<round>/idle_latency.log"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nclt_slam_project_amd.engine import Engine
from nclt_slam_project_amd import synth
import bench

W, H = 640, 480


def run(e, fd, base_poses, res, idle_s, mode, n, warm_every_s=0.0, warm_call=None, pre_warm=None):
    ts, dev = [], []
    for i in range(n + 1):
        t_end = time.perf_counter() + idle_s
        if warm_every_s > 0:
            while time.perf_counter() + warm_every_s < t_end:
                time.sleep(warm_every_s)
                warm_call(); e.sync()
        rest = t_end - time.perf_counter()
        if rest > 0:
            time.sleep(rest)
        e.profile_enable(True)
        res[72:76] = 255
        t0 = time.perf_counter()
        if pre_warm:
            pre_warm()
        e.tick_dev(fd[i % 8], W, H, base_poses[i % 8], False, mode, i)
        (e.sync if os.environ.get("EXP_WAIT") == "sync" else e.tick_wait)()
        ts.append(time.perf_counter() - t0)
        e.sync()
        assert res[72] != 255
        st = [e.profile_get(k) for k in (2, 0, 3)]
        dev.append([ms / max(k, 1) * 1e3 for ms, k in st])
        e.profile_enable(False)
    ts = np.array(ts[1:]) * 1e6
    dev = np.array(dev[1:])
    return dict(median_us=round(float(np.median(ts)), 1), p95_us=round(float(np.percentile(ts, 95)), 1), min_us=round(float(ts.min()), 1),
                orb_us=round(float(np.median(dev[:, 0])), 1), scan_us=round(float(np.median(dev[:, 1])), 1), pnp_us=round(float(np.median(dev[:, 2])), 1))


if __name__ == "__main__":
    e = Engine(0, W, H, 2048)
    frames, db, base_poses = bench.build_workload(e, 10000, "fixed64", 8)
    e.db_upload(*db)
    fd = [e.to_device(f) for f in frames]
    res = e.pinned((96,), np.uint8)
    e.tick_result_to(res)
    e.set_exclusive(True)
    cur1 = e.to_device(synth.random_descriptors(np.random.default_rng(1), 1))
    cnt = e.dev_alloc(10000 * 4)
    small = lambda: e.db_match_counts_dev(cur1, 1, cnt)                # a 7 us kernel
    for i in range(30):
        e.tick_dev(fd[i % 8], W, H, base_poses[i % 8], False, 1, i); e.sync()
    for mode, name in ((1, "global"), (0, "local")):
        for idle in (0.0, 0.0005, 0.002, 0.005, 0.02, 0.05, 0.1, 0.5):
            n = 30 if idle <= 0.05 else 16
            print(json.dumps(dict(tick=name, idle_s=idle, **run(e, fd, base_poses, res, idle, mode, n))), flush=True)
        for every in (0.002, 0.01, 0.033, 0.1):
            print(json.dumps(dict(tick=name, idle_s=0.5, keep_warm_every_s=every, warm="7 us kernel + sync",
                                  **run(e, fd, base_poses, res, 0.5, mode, 12, every, small))), flush=True)
        print(json.dumps(dict(tick=name, idle_s=0.5, pre_warm="one 7 us kernel enqueued right in front of the tick (inside the timed span)",
                              **run(e, fd, base_poses, res, 0.5, mode, 12, pre_warm=small))), flush=True)
