"""Dev experiment: per-stage HIP-event times (ORB / scan / PnP) of every stream while all streams run."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from nclt_slam_project_amd.engine import Engine
import bench
W, H = 640, 480
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
engines = [Engine(0, W, H, 2048) for _ in range(NS)]
frames, db, base_poses = bench.build_workload(engines[0], 10000, "fixed64", 8)
for e in engines:
    e.db_upload(*db)
fd = [[e.to_device(f) for f in frames] for e in engines]
def run(n):
    for i in range(n):
        s = i % NS
        engines[s].tick_dev(fd[s][i % 8], W, H, base_poses[i % 8], order_rgb=False, global_reloc=True, seed=i)
    for e in engines: e.sync()
run(128)
for e in engines: e.profile_enable(True)
t0 = time.perf_counter(); run(1024); dt = time.perf_counter() - t0
print(f"{NS} streams: {1024/dt:.0f} frames/s, period {dt/1024*1e6:.1f} us")
for k, e in enumerate(engines):
    r = {name: e.profile_get(i) for name, i in (("scan", 0), ("orb", 2), ("pnp", 3))}
    print(k, {n: round(ms / max(c, 1) * 1e3, 1) for n, (ms, c) in r.items()}, "events", r["scan"][1])
