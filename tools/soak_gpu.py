"""Developer soak test: device-memory accounting over engine create/destroy cycles and many ticks (needs a GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nclt_slam_project_amd import synth
from nclt_slam_project_amd.engine import Engine

def free_mb():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2**20

rng = np.random.default_rng(0)
img = synth.textured_frame(rng, 640, 480)
base = free_mb()
after_close = []
for cyc in range(5):
    e = Engine(0, 1280, 720, 8192)
    feat = e.orb_detect_compute(e.gray(img), 500)
    desc, pts, off, poses = synth.descriptor_db(rng, 2000, "ragged", feat["desc"], planted_records=(5, 100))
    e.db_upload(desc, pts, off, poses)
    e.db_upload(desc, pts, off, poses)          # re-upload frees the previous arena
    m0 = free_mb()
    for i in range(3000):
        e.tick(img, synth.base_pose(10.0, 0.0, 0.0), global_reloc=(i % 2 == 0), seed=i)
        if i % 500 == 0:
            e.match_mutual(desc[:400 + i % 97], feat["desc"]); e.hamming_matrix(desc[:300], feat["desc"])
            e.pnp_ransac(pts[:50], feat["xy"][:50]); e.record_frame(img, synth.ground_depth_mm(rng))
    m1 = free_mb()
    e.close()
    after_close.append(free_mb())
    print(f"cycle {cyc}: free before ticks {m0:.0f} MiB, after {m1:.0f} MiB, after close {free_mb():.0f} MiB (baseline {base:.0f})", flush=True)
# the first Engine pays the one-time HIP runtime / code-object footprint; compare cycle to cycle
assert abs(after_close[-1] - after_close[0]) < 16, "device memory grows across create/destroy cycles"
print("soak ok")
