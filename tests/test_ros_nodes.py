"""CPU: the rclpy wrappers (nclt-slam-project_amd/ros_nodes.py) under stand-in rclpy / message modules -- rclpy is not
installed anywhere this repository runs, so this is the only execution these wrappers get.  Features come from the
oracle-backed cv2 double; checked: node names, topics, constructor and CLI signatures of the reference nodes
(visual_landmark_matcher.py:175-231, 503-530; visual_landmark_recorder.py:154-179, 375-392; the split variant's
--landmarks-return / --swap-flag, X:513-519), message decoding (rgb8, 32FC1 with NaN/Inf) and that a teach + repeat
driven through the ROS callbacks writes the rows the reference's own nodes wrote (tests/golden/tick_scene.json)."""
import json
import os
import sys
import types

import numpy as np
import pytest

from nclt_slam_project_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


class _Pub:
    def __init__(self, topic):
        self.topic, self.msgs = topic, []

    def publish(self, m):
        self.msgs.append(m)


class _Bag:
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        v = _Bag()
        object.__setattr__(self, k, v)
        return v


@pytest.fixture()
def ros(monkeypatch):
    """minimal rclpy / sensor_msgs / geometry_msgs stand-ins; records subscriptions, publishers and timers"""
    log = dict(subs=[], pubs=[], timers=[], names=[], spun=[])

    class Node:
        def __init__(self, name, *a, **k):
            log["names"].append(name)

        def get_logger(self):
            return types.SimpleNamespace(info=lambda *a: None, warn=lambda *a: None, error=lambda *a: None)

        def create_subscription(self, typ, topic, cb, depth):
            log["subs"].append((topic, cb))

        def create_publisher(self, typ, topic, depth):
            p = _Pub(topic)
            log["pubs"].append(p)
            return p

        def create_timer(self, period, cb):
            log["timers"].append((period, cb))

        def get_clock(self):
            return types.SimpleNamespace(now=lambda: types.SimpleNamespace(to_msg=lambda: 0))

        def destroy_node(self):
            log["destroyed"] = True

    mods = {"rclpy": dict(init=lambda *a, **k: None, shutdown=lambda *a, **k: None, spin=lambda n: log["spun"].append(n)),
            "rclpy.node": dict(Node=Node), "sensor_msgs": {}, "sensor_msgs.msg": dict(Image=_Bag), "geometry_msgs": {},
            "geometry_msgs.msg": dict(PoseWithCovarianceStamped=_Bag)}
    for name, attrs in mods.items():
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        monkeypatch.setitem(sys.modules, name, m)
    return log


def _image(arr, encoding):
    return types.SimpleNamespace(data=np.ascontiguousarray(arr).tobytes(), height=arr.shape[0], width=arr.shape[1], encoding=encoding,
                                 header=types.SimpleNamespace(stamp=types.SimpleNamespace(sec=12, nanosec=500000000)))


def test_message_decoding():
    from nclt_slam_project_amd import ros_nodes as R
    bgr = np.arange(2 * 3 * 3, dtype=np.uint8).reshape(2, 3, 3)
    np.testing.assert_array_equal(R.img_msg_to_bgr(_image(bgr[:, :, ::-1], "rgb8")), bgr)      # M:27-33: rgb8 -> BGR
    np.testing.assert_array_equal(R.img_msg_to_bgr(_image(bgr, "bgr8")), bgr)
    d = np.array([[0.5, np.nan], [np.inf, 12.3456]], np.float32)                               # M:40-44: metres -> mm, NaN/Inf -> 0
    np.testing.assert_array_equal(R.img_msg_to_depth_mm(_image(d, "32FC1")), np.array([[500, 0], [0, 12345]], np.uint16))
    with pytest.raises(ValueError):
        R.img_msg_to_bgr(_image(bgr, "mono8"))


def test_teach_and_repeat_through_the_ros_callbacks(ros, oracle, tmp_path, monkeypatch):
    from oracle_backend import oracle_cv2
    from nclt_slam_project_amd import ros_nodes as R
    gold = json.load(open(os.path.join(GOLD, "tick_scene.json")))
    scene = synth.WallScene()
    cv2 = oracle_cv2()
    pose = [None]
    monkeypatch.setattr(R, "read_pose_file", lambda *a: pose[0])
    # ---- teach: VisualLandmarkRecorder(out_pkl, min_disp_m), 5 Hz timer, two camera topics (R:154-179)
    pkl = str(tmp_path / "db" / "landmarks.pkl")
    rec = R.make_recorder_node(pkl, 2.0, cv2=cv2)
    assert ros["names"] == ["visual_landmark_recorder"]
    assert [t for t, _ in ros["subs"]] == ["/camera/color/image_raw", "/camera/depth/image_rect_raw"]
    assert ros["timers"][0][0] == pytest.approx(0.2)
    for x in gold["teach_x"]:
        pose[0] = synth.base_pose(x, 0.0, 0.0)
        bgr, dep = scene.render(pose[0])
        rec._rgb_cb(_image(bgr[:, :, ::-1], "rgb8"))
        rec._depth_cb(_image(dep.astype(np.float32) / 1000.0, "32FC1"))
        ros["timers"][0][1]()
    assert [lm["n_features"] for lm in rec.core.landmarks] == [r["n"] for r in gold["records"]]
    rec.core.save()
    # ---- repeat: VisualLandmarkMatcher(pkl_path, log_csv), 2 Hz, publishes /anchor_correction (M:175-231)
    n_sub = len(ros["subs"])
    csv = str(tmp_path / "out" / "anchor_matches.csv")
    node = R.make_matcher_node(pkl, csv, cv2=cv2)
    assert ros["names"][-1] == "visual_landmark_matcher" and ros["pubs"][-1].topic == "/anchor_correction"
    assert [t for t, _ in ros["subs"][n_sub:]] == ["/camera/color/image_raw", "/camera/depth/image_rect_raw"]
    assert ros["timers"][-1][0] == pytest.approx(0.5)
    import time as _t
    ts = [1000.0]
    monkeypatch.setattr(_t, "time", lambda: ts[0])
    for i, (x, y, yaw) in enumerate(gold["repeat"]):
        pose[0] = synth.base_pose(x, y, yaw)
        bgr, dep = scene.render(pose[0])
        node._rgb_cb(_image(bgr[:, :, ::-1], "rgb8"))
        node._depth_cb(_image(dep, "16UC1"))
        ts[0] = 1000.0 + 0.5 * i
        ros["timers"][-1][1]()
    assert open(csv).read().splitlines() == gold["csv"]
    msgs = ros["pubs"][-1].msgs
    assert len(msgs) == len(gold["published"])
    for m, g in zip(msgs, gold["published"]):
        got = [m.pose.pose.position.x, m.pose.pose.position.y, m.pose.pose.position.z, m.pose.pose.orientation.x,
               m.pose.pose.orientation.y, m.pose.pose.orientation.z, m.pose.pose.orientation.w]
        np.testing.assert_allclose(got, g["pose"], atol=1e-9)
        np.testing.assert_allclose(m.pose.covariance, g["cov"], atol=1e-15)
        assert m.header.frame_id == "map"


def test_cli_flags_are_the_references(ros, monkeypatch, tmp_path):
    """--landmarks --out-csv (M:505-508), --landmarks-return --swap-flag (X:513-519), recorder --out --min-disp (R:376-379)"""
    from nclt_slam_project_amd import ros_nodes as R
    made = {}

    class _N:
        core = types.SimpleNamespace(save_augmented=lambda: made.setdefault("saved", True), save=lambda: made.setdefault("rsaved", True))

        def destroy_node(self):
            made["destroyed"] = True

    monkeypatch.setattr(R, "make_matcher_node", lambda *a: made.setdefault("matcher", a) and _N())
    monkeypatch.setattr(R, "make_recorder_node", lambda *a: made.setdefault("recorder", a) and _N())
    R.matcher_main(["--landmarks", "a.pkl", "--out-csv", "o.csv", "--landmarks-return", "b.pkl", "--swap-flag", "/tmp/f"])
    assert made["matcher"][:4] == ("a.pkl", "o.csv", "b.pkl", "/tmp/f") and made["saved"] and len(ros["spun"]) == 1
    made.pop("matcher")
    R.matcher_main(["--landmarks", "a.pkl", "--out-csv", "o.csv"])
    assert made["matcher"][2] is None and made["matcher"][3] == "/tmp/matcher_swap_return.txt"      # X:517 default
    R.recorder_main(["--out", "l.pkl", "--min-disp", "1.5"])
    assert made["recorder"] == ("l.pkl", 1.5) and made["rsaved"]
    with pytest.raises(SystemExit):
        R.matcher_main(["--out-csv", "o.csv"])                                                     # --landmarks is required
