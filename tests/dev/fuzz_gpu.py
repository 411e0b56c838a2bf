"""Randomized parity campaign: HIP library vs CPU oracle on random shapes (developer tool, needs a GPU).
    python tests/dev/fuzz_gpu.py --seconds 240 --seed 1 [--kinds db_small,db]
(lives under tests/: it checks against the CPU oracle, which only test code may import)
Exits non-zero on the first mismatch and prints the reproducing seed/case."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from nclt_slam_project_amd import synth
from nclt_slam_project_amd.engine import Engine
from oracle import oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--kinds", default="", help="comma-separated subset of the case kinds (default: all)")
args = ap.parse_args()
O.build()
e = Engine(0, 1280, 720, 8192)
# a second context in the generations form of the whole-database scan (row budgets + sweepers: the default of batched launches;
# single launches use one resident generation since round 4): developer switches are read at creation, under RELOC_DEV=1
os.environ["RELOC_DEV"] = "1"; os.environ["RELOC_SCAN_GENS"] = "3"
e_gens = Engine(0, 1280, 720, 8192)
del os.environ["RELOC_DEV"], os.environ["RELOC_SCAN_GENS"]
rng = np.random.default_rng(args.seed)
t_end = time.time() + args.seconds
n_case = {"match": 0, "knn": 0, "db": 0, "db_small": 0, "db_big": 0, "ratio": 0, "matrix": 0, "orb": 0, "orb_bgr": 0, "pnp": 0, "record": 0}


def descs(n, dup_p=0.1, low_entropy=False):
    d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    if low_entropy:                       # few distinct bits -> massive distance ties
        d &= rng.integers(0, 256, (1, 32), dtype=np.uint8) & 0x11
    for _ in range(int(n * dup_p)):
        d[rng.integers(0, n)] = d[rng.integers(0, n)]
    return d


def fail(kind, info):
    print("MISMATCH", kind, info, "seed", args.seed, "case", n_case)
    sys.exit(1)


t_report = time.time() + 60
while time.time() < t_end:
    if time.time() > t_report:                    # a progress line per minute (a silent GPU job is taken to be hung)
        print("progress", sum(n_case.values()), "cases", flush=True)
        t_report = time.time() + 60
    kind = rng.choice([k for k in n_case if not args.kinds or k in args.kinds.split(",")])
    n_case[kind] += 1
    if kind == "match":
        nq, nt = int(rng.integers(1, 700)), int(rng.integers(1, 1600))
        le = rng.random() < 0.3
        q, t = descs(nq, low_entropy=le), descs(nt, low_entropy=le)
        g, x = e.match_mutual(q, t), O.match_mutual(q, t)
        if any(not np.array_equal(a, b) for a, b in zip(g, x)): fail(kind, (nq, nt, le))
    elif kind == "knn":
        nq, nt = int(rng.integers(1, 900)), int(rng.integers(1, 5000))
        q, t = descs(nq), descs(nt, low_entropy=rng.random() < 0.3)
        gi, gd = e.match_knn2(q, t); xi, xd = O.match_knn2(q, t)
        if not (np.array_equal(gi, xi) and np.array_equal(gd, xd)): fail(kind, (nq, nt))
    elif kind in ("db", "ratio"):
        L = int(rng.integers(1, 200)); Q = int(rng.integers(1, 1100))
        n = rng.integers(0, 140, L); n[rng.integers(0, L)] = int(rng.integers(0, 600))
        off = np.zeros(L + 1, np.int64); off[1:] = np.cumsum(n)
        le = rng.random() < 0.3
        cur = descs(Q, low_entropy=le); db = descs(max(int(off[-1]), 1), low_entropy=le)[: int(off[-1])]
        if len(db) and rng.random() < 0.7:
            r = int(rng.integers(0, L)); k = int(min(n[r], Q))
            if k: db[off[r]:off[r] + k] = synth.perturb_descriptors(rng, cur[rng.choice(Q, k, replace=False)], 0.05)
        e.db_upload(db, np.zeros((len(db), 3), np.float32), off, np.tile([0, 0, 0, 0, 0, 0, 1.0], (L, 1)))
        if kind == "db":
            if not np.array_equal(e.db_match_counts(cur), O.db_match_counts(db, off, cur)): fail(kind, (L, Q, le))
        else:
            ratio = float(rng.choice([0.7, 0.75, 0.8, 0.9, 1.0]))
            if not np.array_equal(e.db_ratio_counts(cur, ratio), O.db_ratio_counts(db, off, cur, ratio)): fail(kind, (L, Q, ratio))
    elif kind == "db_small":      # the lane-per-teach-row kernel (<= 64 current descriptors): 4- / 8- / 16-query groups, records of
        L = int(rng.integers(1, 400)); Q = int(rng.integers(1, 65))          # 0 .. 1100 rows (> 1024 rows: the column kernel takes over)
        n = rng.integers(0, 140, L); n[rng.integers(0, L)] = int(rng.integers(0, 1100))
        if rng.random() < 0.3: n[:] = 64
        off = np.zeros(L + 1, np.int64); off[1:] = np.cumsum(n)
        le = rng.random() < 0.4
        cur = descs(Q, low_entropy=le); db = descs(max(int(off[-1]), 1), low_entropy=le)[: int(off[-1])]
        if len(db) and rng.random() < 0.7:
            r = int(rng.integers(0, L)); k = int(min(n[r], Q))
            if k: db[off[r]:off[r] + k] = synth.perturb_descriptors(rng, cur[rng.choice(Q, k, replace=False)], 0.05)
        e.db_upload(db, np.zeros((len(db), 3), np.float32), off, np.tile([0, 0, 0, 0, 0, 0, 1.0], (L, 1)))
        if not np.array_equal(e.db_match_counts(cur), O.db_match_counts(db, off, cur)): fail(kind, (L, Q, le))
    elif kind == "db_big":        # more records than resident workgroups: ticket counters, row budgets, sweepers; both scheduling forms
        L = int(rng.integers(1100, 3500)); Q = int(rng.integers(65, 400))
        n = rng.integers(0, 24, L); n[rng.choice(L, 5, replace=False)] = rng.integers(100, 900, 5)
        off = np.zeros(L + 1, np.int64); off[1:] = np.cumsum(n)
        cur = descs(Q); db = descs(max(int(off[-1]), 1))[: int(off[-1])]
        x = O.db_match_counts(db, off, cur)
        for form, eng in (("one", e), ("gens3", e_gens)):
            eng.db_upload(db, np.zeros((len(db), 3), np.float32), off, np.tile([0, 0, 0, 0, 0, 0, 1.0], (L, 1)))
            g = eng.db_match_counts(cur)
            if not np.array_equal(g, x): fail(kind, (L, Q, form, int((g != x).sum())))
    elif kind == "matrix":
        na, nb = int(rng.integers(1, 300)), int(rng.integers(1, 5000))
        a, b = descs(na), descs(nb)
        if not np.array_equal(e.hamming_matrix(a, b), O.hamming_matrix(a, b)): fail(kind, (na, nb))
    elif kind == "orb":
        w, h = int(rng.integers(63, 900)), int(rng.integers(63, 700))
        img = synth.textured_frame(rng, w, h, n_shapes=int(rng.integers(5, max(6, w * h // 600))), noise=float(rng.choice([0, 1, 4, 12])))
        if rng.random() < 0.2: img = (img // 64) * 64                  # posterised: many equal scores / responses
        gray = O.gray_u8(img)
        nf = int(rng.choice([100, 500, 500, 1000, 3000]))
        x = O.orb_detect_compute(gray, nf, max_out=e.max_feat); g = e.orb_detect_compute(gray, nf)
        if g["n"] != min(x["n"], e.max_feat): fail(kind, (w, h, nf, g["n"], x["n"]))
        k = g["n"]
        for f in ("xy", "angle", "response", "size"):
            if not np.array_equal(g[f].view(np.uint32), x[f][:k].view(np.uint32)): fail(kind, (w, h, nf, f))
        if not (np.array_equal(g["desc"], x["desc"][:k]) and np.array_equal(g["octave"], x["octave"][:k])): fail(kind, (w, h, nf, "desc"))
    elif kind == "orb_bgr":       # the tick's input form: interleaved 3-channel frame in device memory, any width / row stride
        w, h = int(rng.integers(64, 1281)), int(rng.integers(64, 721))
        img = synth.textured_frame(rng, w, h, n_shapes=int(rng.integers(5, max(6, w * h // 600))), noise=float(rng.choice([0, 1, 4, 12])))
        stride = 3 * w + int(rng.choice([0, 0, 1, 2, 3, 4, 13]))
        raw = np.zeros((h, stride), np.uint8); raw[:, :3 * w] = img.reshape(h, 3 * w)
        order = bool(rng.integers(0, 2))
        bits = int(rng.choice([15, 15, 14]))                              # reloc_params.gray_coeff_bits
        dev = e.dev_alloc(raw.nbytes)
        e.h2d(dev, raw)
        e.set_params(gray_coeff_bits=bits)
        n = e.orb_frame_dev(dev, w, h, stride, order_rgb=order)
        e.set_params(gray_coeff_bits=15)
        e.dev_free(dev)
        gray = O.gray_u8(img, order, bits)
        pyr = O.pyramid(gray)
        for l in range(8):
            if not np.array_equal(e.frame_debug_plane(0, l), pyr[l]): fail(kind, (w, h, stride, order, "level", l))
        x = O.orb_detect_compute(gray, 500, max_out=e.max_feat)
        if n != min(x["n"], e.max_feat): fail(kind, (w, h, stride, order, n, x["n"]))
    elif kind == "pnp":
        m = int(rng.integers(4, 800))
        obj, img, rv, tv, inl = synth.pnp_problem(rng, m=m, outlier_ratio=float(rng.uniform(0, 0.7)), noise_px=float(rng.choice([0, 0.3, 1.0])))
        seed = int(rng.integers(0, 1 << 30))
        g = e.pnp_ransac(obj, img, seed=seed); x = O.pnp_ransac(obj, img, seed=seed)
        if g[0] != x[0] or not np.array_equal(g[3], x[3]): fail(kind, (m, seed, "inliers"))
        if g[0] and (np.abs(g[2] - x[2]).max() > 1e-4 or np.abs(g[1] - x[1]).max() > 1e-4): fail(kind, (m, seed, "pose", np.abs(g[2] - x[2]).max()))
    elif kind == "record":
        img = synth.textured_frame(rng, 640, 480)
        dep = synth.ground_depth_mm(rng, zeros=float(rng.uniform(0, 0.3)))
        from nclt_slam_project_amd.recorder import LandmarkRecorderCore
        from nclt_slam_project_amd.cv2_shim import Cv2Shim
        a = LandmarkRecorderCore(cv2=Cv2Shim(e)).tick(img, dep, synth.base_pose(0, 0, 0), 0.0)
        b = LandmarkRecorderCore(engine=e).tick(img, dep, synth.base_pose(0, 0, 0), 0.0)
        if (a is None) != (b is None): fail(kind, "none")
        if a is not None and not (np.array_equal(a["descriptors"], b["descriptors"]) and
                                  np.array_equal(a["keypoints_3d_cam"].view(np.uint32), b["keypoints_3d_cam"].view(np.uint32))): fail(kind, "arrays")
print("fuzz ok", n_case, flush=True)
