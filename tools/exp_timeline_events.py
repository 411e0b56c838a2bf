"""Developer experiment (round 4): where the four streams of the timed configuration are at any moment, WITHOUT a profiler
(rocprofv3's per-dispatch tracing makes the run twice as slow and tears 200 us holes into it).  A whole-database tick composed of its
three public halves -- ORB (reloc_orb_frame_dev), scan (reloc_db_match_counts_dev on resident descriptors), matches + PnP +
finalisation of the last ranked candidates (reloc_tick_solve_dev) -- with a torch event between the halves on the context's stream.
Reports the composed tick's rate beside the fused tick's, the share of time in which 0 / 1 / 2 / ... streams are in their scan phase
(event after ORB .. event after the scan: queueing for slots included), and the phases' average lengths.
    python tools/exp_timeline_events.py [streams]"""
import ctypes as C, collections, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from nclt_slam_project_amd.engine import Engine

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4
W, H = 640, 480
engines = [Engine(0, W, H, 2048) for _ in range(S)]
frames, db, base_poses = bench.build_workload(engines[0], 10000, "fixed64", 8)
engines[0].db_upload(*db)
for e in engines[1:]:
    e.db_share(engines[0])
ts = [torch.cuda.Stream(device=0) for _ in range(S)]
for e, t in zip(engines, ts):
    e.set_stream(t.cuda_stream)
fd = [engines[0].to_device(f) for f in frames]
cnt = {id(e): e.dev_alloc(10000 * 4) for e in engines}
qd = {id(e): e.dev_alloc(500 * 32) for e in engines}
ids = {id(e): e.dev_alloc(64 * 4) for e in engines}
feat = engines[0].orb_detect_compute(engines[0].gray(frames[0]), 500)
for e in engines:
    e.h2d(qd[id(e)], feat["desc"][:500])
    e.tick_scan_enqueue(fd[0], W, H, None, 25)
    e.d2d(ids[id(e)], e._topk_dev, 25 * 4)
    e.sync()


def fused(n):
    for i in range(n):
        engines[i % S].tick_dev(fd[i % 8], W, H, base_poses[i % 8], False, 1, i)


def composed(n, marks=None):
    for i in range(n):
        e, t = engines[i % S], ts[i % S]
        if marks is not None:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record(t)
        e._lib.reloc_orb_frame_dev(e._ctx, C.c_void_p(fd[i % 8]), W, H, 3 * W, 0, 500)
        if marks is not None: ev[1].record(t)
        e.db_match_counts_dev(qd[id(e)], 500, cnt[id(e)])
        if marks is not None: ev[2].record(t)
        e.tick_solve_from(ids[id(e)], 25, base_poses[i % 8], False, i)
        if marks is not None:
            ev[3].record(t)
            marks.append((i % S, ev))


def rate(fn, n=2048):
    fn(256)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(n)
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)


out = dict(streams=S, fused_tick_fps=round(rate(fused), 1), composed_tick_fps=round(rate(composed), 1))
# the timeline: 768 composed ticks with events (created up front would not help: they are recorded inline), base = one event first
composed(256)
torch.cuda.synchronize()
base = torch.cuda.Event(enable_timing=True)
base.record(ts[0])
marks = []
t0 = time.perf_counter()
composed(768, marks)
torch.cuda.synchronize()
out["composed_with_events_fps"] = round(768 / (time.perf_counter() - t0), 1)
iv = [(s, [base.elapsed_time(x) * 1e3 for x in ev]) for s, ev in marks]          # us since base
lo = sorted(v[1] for _, v in iv)[len(iv) // 8]; hi = sorted(v[2] for _, v in iv)[-len(iv) // 8]
edges = []
for _, v in iv:
    a, b = max(v[1], lo), min(v[2], hi)
    if b > a:
        edges += [(a, 1), (b, -1)]
edges.sort()
cur, last, hist = 0, lo, collections.Counter()
for t, d in edges:
    hist[cur] += t - last; last = t; cur += d
tot = sum(hist.values())
out["streams_in_scan_phase_pct"] = {str(k): round(100 * v / tot, 1) for k, v in sorted(hist.items())}
out["phase_us"] = dict(orb=round(float(np.mean([v[1] - v[0] for _, v in iv])), 1), scan_incl_queueing=round(float(np.mean([v[2] - v[1] for _, v in iv])), 1),
                       solve=round(float(np.mean([v[3] - v[2] for _, v in iv])), 1))
per = collections.defaultdict(list)
for s, v in iv:
    per[s].append(v)
gaps = [b[0] - a[3] for s in per for a, b in zip(per[s], per[s][1:])]
out["gap_between_frames_on_a_stream_us"] = dict(median=round(float(np.median(gaps)), 1), p90=round(float(np.percentile(gaps, 90)), 1))
print(json.dumps(out))
# a stretch of the timeline per stream
mid = iv[len(iv) // 2][1][0]
for s in sorted(per):
    print("S%d" % s, " ".join("orb[%d-%d] scan[%d-%d] solve[%d-%d]" % (v[0] - mid, v[1] - mid, v[1] - mid, v[2] - mid, v[2] - mid, v[3] - mid)
                              for v in per[s] if mid <= v[0] < mid + 1500))
