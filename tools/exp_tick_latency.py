"""Developer experiment: synchronous single-stream tick latency (global / local candidate search) and per-stage kernel
time for one or more builds of the library (RELOC_LIB), each in its own subprocess, interleaved rounds.
    python tools/exp_tick_latency.py [lib.so ...]"""
import glob, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(lib):
    import numpy as np
    import bench
    from nclt_slam_project_amd.engine import Engine
    e = Engine(0, 640, 480, 2048)
    frames, db, base_poses = bench.build_workload(e, 10000, "fixed64", 8)
    e.db_upload(*db)
    fd = [e.to_device(f) for f in frames]
    res = e.pinned((96,), np.uint8)
    e.tick_result_to(res)
    if os.environ.get("EXP_EXCLUSIVE", "1") != "0":
        e.set_exclusive(True)
    out = dict(lib=os.path.basename(lib))
    for i in range(20):
        e.tick_dev(fd[i % 8], 640, 480, base_poses[i % 8], False, 1, i); e.sync()
    for mode, name in ((1, "global"), (0, "local")):
        ts = []
        for i in range(300):
            t0 = time.perf_counter()
            e.tick_dev(fd[i % 8], 640, 480, base_poses[i % 8], False, mode, i)
            e.sync()
            ts.append(time.perf_counter() - t0)
        ts = np.array(ts[50:]) * 1e6
        out[name + "_median_us"] = round(float(np.median(ts)), 1)
        out[name + "_p95_us"] = round(float(np.percentile(ts, 95)), 1)
        e.profile_enable(True)
        for i in range(40):
            e.tick_dev(fd[i % 8], 640, 480, base_poses[i % 8], False, mode, i)
        e.sync()
        st = {}
        for k, nm in ((2, "orb"), (0, "scan"), (3, "pnp")):
            ms, n = e.profile_get(k)
            st[nm] = round(ms / max(n, 1) * 1e3, 1)
        e.profile_enable(False)
        out[name + "_stage_us"] = st
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--one":
        one(sys.argv[2])
    else:
        libs = sys.argv[1:] or [os.path.join(ROOT, "nclt-slam-project_amd", "csrc", "libreloc_hip.so")] + \
            sorted(glob.glob(os.path.join(ROOT, "build_variants", "*.so")))
        for rnd in range(2):
            for lib in libs:
                subprocess.run([sys.executable, os.path.abspath(__file__), "--one", lib], env=dict(os.environ, RELOC_DEV="1", RELOC_LIB=os.path.abspath(lib)), timeout=600)
