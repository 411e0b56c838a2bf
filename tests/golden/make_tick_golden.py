#!/usr/bin/env python3
"""Generates tests/golden/tick_scene.json by running the REFERENCE's unmodified teach and repeat
nodes (VisualLandmarkRecorder._tick / VisualLandmarkMatcher._tick and the global-reloc variant)
on a synthetic scene.

Run in the build container only (needs /root/reference):
    python tests/golden/make_tick_golden.py

How the reference is driven.  Its modules import rclpy / sensor_msgs / geometry_msgs / cv2; none is
installed.  Stand-in modules are registered for the import: rclpy with a minimal Node (logger,
publisher, timer, clock), message classes that are plain attribute bags, and as `cv2` this
repository's cv2-shaped shim over the CPU oracle (tests/oracle_backend.py).  The reference code
itself -- candidate selection, gates, PnP call, pose composition, covariance, CSV writing,
accumulation -- runs unmodified; only `_read_pose` is replaced per tick because it reads a fixed
/tmp path.  The output is data: scene parameters, the poses fed in, and the CSV rows / published
poses / record summaries the reference produced.  tests/test_gpu_tick.py replays the same scene
through the HIP path and must reproduce it.
"""
import importlib.util
import json
import os
import sys
import tempfile
import types
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

REF_COMMON = "/root/reference/simulation/isaac/scripts/common"
REF_G = "/root/reference/simulation/isaac/experiments/63_global_reloc/scripts/visual_landmark_matcher.py"
REF_X = "/root/reference/simulation/isaac/experiments/69_repeat_road_split_landmarks_accel_noise/scripts/visual_landmark_matcher.py"
OUT = os.path.join(HERE, "tick_scene.json")

TEACH_X = [2.0, 4.5, 7.0, 9.5]
REPEAT = [(2.3, -0.2, -2.0), (4.6, 0.25, 3.0), (7.4, 0.1, 1.0), (9.0, -0.3, -1.5), (5.5, 2.5, 20.0), (40.0, 0.0, 0.0),
          (3.0, 0.0, 170.0), (4.0, 9.5, 0.0), (4.5, 0.0, 60.0), (8.0, -1.0, -35.0), (9.5, 0.0, 80.0), (2.0, 5.0, -70.0),
          (2.5, 0.5, 75.0), (12.0, 4.0, 50.0)]
GLOBAL = [(5.0, 9.0, 2.0), (6.0, -9.5, -3.0), (4.6, 0.25, 3.0)]   # first two: > 8 m from every record, only the whole-DB search can anchor
# A repeat session that exercises accumulation (M:435-500): (x, y, yaw_deg, ts).  Off-route poses more than 5 m from
# every record become new records once the matcher has been silent for 5 s; later ticks near them use those records.
SESSION = [(2.3, -0.2, -2.0, 100.0), (5.0, 9.0, 2.0, 100.5), (5.0, 9.0, 2.0, 107.0), (5.3, 9.2, 4.0, 107.5),
           (5.1, 8.8, 0.0, 108.0), (6.0, -9.5, -3.0, 108.5), (6.0, -9.5, -3.0, 120.0), (6.2, -9.3, -1.0, 120.5),
           (7.4, 0.1, 1.0, 121.0), (5.0, 14.5, 0.0, 140.0), (5.2, 14.2, 3.0, 140.5)]
# Split-landmark variant X: outbound = records 0-1, return = records 2-3; the flag file appears before tick X_SWAP_AT.
XRUN = [(2.3, -0.2, -2.0), (4.6, 0.25, 3.0), (7.4, 0.1, 1.0), (9.0, -0.3, -1.5), (7.4, 0.1, 1.0), (9.0, -0.3, -1.5),
        (2.3, -0.2, -2.0), (5.0, 9.0, 2.0), (5.2, 9.1, 1.0)]
X_SWAP_AT = 4


class _Published:
    def __init__(self):
        self.msgs = []

    def publish(self, msg):
        self.msgs.append(msg)


class _Bag:
    """attribute bag that creates nested bags on demand (stands in for ROS message classes)"""

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        v = _Bag()
        object.__setattr__(self, k, v)
        return v


def install_stubs(cv2_obj):
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class _Logger:
        def info(self, *a, **k): pass
        def warn(self, *a, **k): pass
        def error(self, *a, **k): pass

    class _Clock:
        def now(self):
            return types.SimpleNamespace(to_msg=lambda: 0)

    class Node:
        def __init__(self, *a, **k):
            self._pubs = []
        def get_logger(self): return _Logger()
        def create_subscription(self, *a, **k): return None
        def create_publisher(self, *a, **k):
            p = _Published(); self._pubs.append(p); return p
        def create_timer(self, *a, **k): return None
        def get_clock(self): return _Clock()
        def destroy_node(self): pass

    mod("rclpy", init=lambda *a, **k: None, shutdown=lambda *a, **k: None, spin=lambda *a, **k: None)
    mod("rclpy.node", Node=Node)
    mod("sensor_msgs")
    mod("sensor_msgs.msg", Image=_Bag)
    mod("geometry_msgs")
    mod("geometry_msgs.msg", PoseWithCovarianceStamped=_Bag)
    sys.modules["cv2"] = cv2_obj


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def crc(a):
    return int(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def record_summary(lm, index_xy=None):
    d = dict(n=int(lm["n_features"]), pose=[float(v) for v in lm["pose"]], desc_crc=crc(lm["descriptors"]),
             kp2d_crc=crc(lm["keypoints_2d"]), kp3d_crc=crc(lm["keypoints_3d_cam"]), ts=float(lm["ts"]),
             accumulated=bool(lm.get("accumulated", False)))
    if index_xy is not None:
        d["index_xy"] = [float(index_xy[0]), float(index_xy[1])]
    return d


def drive(mt, module, scene, ticks, fixed_ts, before_tick=None):
    """feeds (x, y, yaw, ts) ticks to an unmodified reference matcher node; returns (csv rows, published, accumulated)"""
    from nclt_slam_project_amd import synth
    module.time.time = lambda: fixed_ts[0]
    published = []
    n_init = len(mt.landmarks)
    for i, (x, y, yaw, ts) in enumerate(ticks):
        if before_tick:
            before_tick(i)
        bp = synth.base_pose(x, y, yaw)
        mt.last_rgb, mt.last_depth = scene.render(bp)
        mt._read_pose = lambda bp=bp: bp
        fixed_ts[0] = ts
        n0 = len(mt.anchor_pub.msgs)
        mt._tick()
        if len(mt.anchor_pub.msgs) > n0:
            msg = mt.anchor_pub.msgs[-1]
            p, o = msg.pose.pose.position, msg.pose.pose.orientation
            published.append(dict(tick=i, pose=[float(p.x), float(p.y), float(p.z), float(o.x), float(o.y), float(o.z), float(o.w)],
                                  cov=[float(c) for c in msg.pose.covariance]))
    rows = open(mt.log_csv).read().splitlines()
    acc = [record_summary(lm, mt.xy[k]) for k, lm in enumerate(mt.landmarks) if lm.get("accumulated")]
    return rows, published, acc, n_init


def run(cv2_obj):
    """returns the golden dict; also used by tests/test_reference_dropin.py"""
    from nclt_slam_project_amd import synth
    install_stubs(cv2_obj)
    R = load(os.path.join(REF_COMMON, "visual_landmark_recorder.py"), "ref_recorder")
    M = load(os.path.join(REF_COMMON, "visual_landmark_matcher.py"), "ref_matcher")
    G = load(REF_G, "ref_matcher_g")
    scene = synth.WallScene()
    tmp = tempfile.mkdtemp(prefix="reloc_golden_")
    pkl = os.path.join(tmp, "db", "landmarks.pkl")
    # ---- teach with the reference recorder
    rec = R.VisualLandmarkRecorder(pkl, 2.0)
    records = []
    for x in TEACH_X:
        bp = synth.base_pose(x, 0.0, 0.0)
        rec.last_rgb, rec.last_depth = scene.render(bp)
        rec.last_rgb_ts = x
        rec._read_pose = lambda bp=bp: bp
        n0 = len(rec.landmarks)
        rec._tick()
        if len(rec.landmarks) > n0:
            lm = rec.landmarks[-1]
            records.append(dict(x=x, n=int(lm["n_features"]), pose=[float(v) for v in lm["pose"]],
                                desc_crc=crc(lm["descriptors"]), kp2d_crc=crc(lm["keypoints_2d"]),
                                kp3d_crc=crc(lm["keypoints_3d_cam"])))
    rec._save()
    # ---- repeat with the reference matcher
    csv_path = os.path.join(tmp, "out", "anchor_matches.csv")
    mt = M.VisualLandmarkMatcher(pkl, csv_path)
    published = []
    fixed_ts = [1000.0]
    M.time.time = lambda: fixed_ts[0]
    for i, (x, y, yaw) in enumerate(REPEAT):
        bp = synth.base_pose(x, y, yaw)
        mt.last_rgb, mt.last_depth = scene.render(bp)
        mt._read_pose = lambda bp=bp: bp
        fixed_ts[0] = 1000.0 + 0.5 * i
        n0 = len(mt.anchor_pub.msgs)
        mt._tick()
        if len(mt.anchor_pub.msgs) > n0:
            msg = mt.anchor_pub.msgs[-1]
            p, o = msg.pose.pose.position, msg.pose.pose.orientation
            published.append(dict(tick=i, pose=[float(p.x), float(p.y), float(p.z), float(o.x), float(o.y), float(o.z), float(o.w)],
                                  cov=[float(c) for c in msg.pose.covariance]))
    rows = open(csv_path).read().splitlines()
    # ---- global relocalisation variant: drift file says 10 m, matcher silent for > 20 s
    csv_g = os.path.join(tmp, "out_g", "anchor_matches.csv")
    mg = G.VisualLandmarkMatcher(pkl, csv_g)
    G.time.time = lambda: fixed_ts[0]
    real_open = open

    def fake_open(path, *a, **k):
        if path == "/tmp/drift_est.txt":
            import io
            return io.StringIO("10.0\n")
        return real_open(path, *a, **k)

    G.open = fake_open
    published_g = []
    for i, (x, y, yaw) in enumerate(GLOBAL):
        bp = synth.base_pose(x, y, yaw)
        mg.last_rgb, mg.last_depth = scene.render(bp)
        mg._read_pose = lambda bp=bp: bp
        fixed_ts[0] = 5000.0 + 0.5 * i
        n0 = len(mg.anchor_pub.msgs)
        mg._tick()
        if len(mg.anchor_pub.msgs) > n0:
            msg = mg.anchor_pub.msgs[-1]
            p, o = msg.pose.pose.position, msg.pose.pose.orientation
            published_g.append(dict(tick=i, pose=[float(p.x), float(p.y), float(p.z), float(o.x), float(o.y), float(o.z), float(o.w)]))
    rows_g = real_open(csv_g).read().splitlines()
    accumulated_repeat = [record_summary(lm, mt.xy[k]) for k, lm in enumerate(mt.landmarks) if lm.get("accumulated")]
    # ---- accumulation session with the unmodified matcher M
    ms = M.VisualLandmarkMatcher(pkl, os.path.join(tmp, "out_s", "anchor_matches.csv"))
    rows_s, pub_s, acc_s, _ = drive(ms, M, scene, SESSION, fixed_ts)
    # ---- split-landmark variant X, unmodified: outbound / return files, swap flag appears mid-session
    import pickle
    X = load(REF_X, "ref_matcher_x")
    with real_open(pkl, "rb") as f:
        data = pickle.load(f)
    pkl_out, pkl_ret = os.path.join(tmp, "db", "out.pkl"), os.path.join(tmp, "db", "ret.pkl")
    for path, lms in ((pkl_out, data["landmarks"][:2]), (pkl_ret, data["landmarks"][2:])):
        with real_open(path, "wb") as f:
            pickle.dump({**data, "landmarks": lms}, f)
    flag = os.path.join(tmp, "swap_flag.txt")
    mx = X.VisualLandmarkMatcher(pkl_out, os.path.join(tmp, "out_x", "anchor_matches.csv"), return_pkl=pkl_ret, swap_flag=flag)

    def before(i):
        if i == X_SWAP_AT:
            with real_open(flag, "w") as f:
                f.write("1")

    rows_x, pub_x, acc_x, _ = drive(mx, X, scene, [(x, y, yaw, 3000.0 + 6.0 * i) for i, (x, y, yaw) in enumerate(XRUN)],
                                    fixed_ts, before)
    return dict(teach_x=TEACH_X, repeat=REPEAT, global_poses=GLOBAL, records=records, csv=rows, published=published,
                csv_global=rows_g, published_global=published_g, accumulated_repeat=accumulated_repeat,
                session=SESSION, csv_session=rows_s, published_session=pub_s, accumulated_session=acc_s,
                xrun=XRUN, x_swap_at=X_SWAP_AT, csv_x=rows_x, published_x=pub_x, accumulated_x=acc_x,
                x_landmarks_after=len(mx.landmarks))


def main():
    from oracle_backend import oracle_cv2
    gold = run(oracle_cv2())
    with open(OUT, "w") as f:
        json.dump(gold, f, indent=1)
    print("wrote", OUT)
    for r in gold["csv"]:
        print("  ", r)
    for r in gold["csv_global"]:
        print(" G", r)
    for r in gold["csv_session"]:
        print(" S", r)
    for r in gold["csv_x"]:
        print(" X", r)
    print("accumulated: repeat", len(gold["accumulated_repeat"]), "session", [(a["n"], a["index_xy"]) for a in gold["accumulated_session"]],
          "x", [(a["n"], a["index_xy"]) for a in gold["accumulated_x"]])
    print("records:", [(r["x"], r["n"]) for r in gold["records"]])


if __name__ == "__main__":
    main()
