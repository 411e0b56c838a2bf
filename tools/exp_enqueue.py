"""Dev experiment: is the 4-stream bench limited by the single enqueueing CPU thread?
(a) time of one tick_dev call (enqueue only), (b) throughput with one Python thread per stream
(ctypes releases the GIL during the C call)."""
import sys, time, threading
import numpy as np
sys.path.insert(0, ".")
import torch
from nclt_slam_project_amd.engine import Engine
import bench

W, H = 640, 480
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
engines = [Engine(0, W, H, 2048) for _ in range(NS)]
frames, db, base_poses = bench.build_workload(engines[0], 10000, "fixed64", 8)
for e in engines:
    e.db_upload(*db)
fd = [[e.to_device(f) for f in frames] for e in engines]

def run_single(n):
    t_enq = 0.0
    for i in range(n):
        s = i % NS
        t0 = time.perf_counter()
        engines[s].tick_dev(fd[s][i % 8], W, H, base_poses[i % 8], order_rgb=False, global_reloc=True, seed=i)
        t_enq += time.perf_counter() - t0
    for e in engines:
        e.sync()
    return t_enq

run_single(64)
t0 = time.perf_counter(); te = run_single(1280); dt = time.perf_counter() - t0
print(f"single thread: {1280/dt:.0f} frames/s; enqueue {te/1280*1e6:.1f} us per tick ({te/dt*100:.0f} % of wall)")

def worker(s, n):
    e = engines[s]
    for i in range(n):
        e.tick_dev(fd[s][i % 8], W, H, base_poses[i % 8], order_rgb=False, global_reloc=True, seed=i)
    e.sync()

for rep in range(2):
    th = [threading.Thread(target=worker, args=(s, 320)) for s in range(NS)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print(f"{NS} threads: {NS*320/dt:.0f} frames/s")
