#!/usr/bin/env python3
"""Generates tests/golden/anchor_fusion.json by driving the REFERENCE's unmodified pose relay
(simulation/isaac/scripts/common/tf_wall_clock_relay_v55.py): its /anchor_correction callback `_anchor_cb` (T:235-256)
and the regime switch / blend inside `_tick_slam_encoder` (T:533-591).

Run in the build container only (needs /root/reference):
    python tests/golden/make_anchor_fusion_golden.py

The relay imports rclpy, tf2_ros and ROS message packages; stand-ins are registered for the import (the same kind of
stubs tests/golden/make_tick_golden.py uses).  The relay's code runs unmodified; only its I/O is replaced per tick:
`_read_slam_pose_raw` (reads a /tmp file) and `_slam_se3_to_nav` (SLAM -> nav alignment, outside this path) return the
scripted SLAM position, the wall clock `pytime.time` is scripted, the compass / encoder noise draws are zero, and
`_publish_odom` records the blended nav pose instead of publishing it.  The fixture is data: the scripted inputs and the
regime / weights / nav position the reference produced.
"""
import json
import math
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF_T = "/root/reference/simulation/isaac/scripts/common/tf_wall_clock_relay_v55.py"
OUT = os.path.join(HERE, "anchor_fusion.json")

# (dt since start [s], ground-truth x, y, SLAM nav x, y, anchor or None); anchor = (x, y, cov0)
def script():
    ev = []
    t = 0.0
    x = 0.0
    std_seq = {  # tick -> anchor std
        12: 0.05, 22: 0.05, 32: 0.05, 40: 0.17, 52: 0.25, 70: 0.05, 80: 0.05, 90: 0.11, 100: 0.05, 440: 0.2, 445: 0.1,
    }
    for k in range(480):
        t = 0.05 * k
        x = 0.02 * k
        drift = 0.0 if k < 310 else min(12.0, 0.1 * (k - 310))       # SLAM drifts away later: exercises the alpha ladder
        slam = (x + 0.3 + drift, 0.1)
        anchor = None
        if k in std_seq:
            s = std_seq[k]
            anchor = (x - 0.2, 0.05, s * s)
        ev.append((t, x, 0.0, slam[0], slam[1], anchor))
    return ev


def main():
    import make_tick_golden as G
    G.install_stubs(types.ModuleType("cv2"))
    Bag = G._Bag

    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class _Br:
        def __init__(self, *a, **k): pass
        def sendTransform(self, *a, **k): pass

    class _Time(Bag):
        def __init__(self, **kw):
            for k, v in kw.items():
                object.__setattr__(self, k, v)

    mod("builtin_interfaces"); mod("builtin_interfaces.msg", Time=_Time)
    mod("geometry_msgs.msg", TransformStamped=Bag, Quaternion=Bag, Twist=Bag, PoseWithCovarianceStamped=Bag)
    mod("nav_msgs"); mod("nav_msgs.msg", Odometry=Bag)
    mod("sensor_msgs.msg", Image=Bag, CameraInfo=Bag, PointCloud2=Bag, PointField=Bag, Imu=Bag)
    mod("tf2_ros", TransformBroadcaster=_Br, StaticTransformBroadcaster=_Br)
    T = G.load(REF_T, "ref_relay")
    clock = [1000.0]
    T.pytime.time = lambda: clock[0]
    T.np.random.normal = lambda *a, **k: 0.0
    node = T.TFRelay(slam_encoder=True)
    slam_now = [None]
    node._read_slam_pose_raw = lambda: (0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, True)
    node._slam_se3_to_nav = lambda *a: (slam_now[0][0], slam_now[0][1], 0.0)
    published = []
    node._publish_odom = lambda now, x, y, *a: published.append((float(x), float(y)))
    rows = []
    for (t, gx, gy, sx, sy, anchor) in script():
        clock[0] = 1000.0 + t
        if anchor is not None:
            msg = Bag()
            msg.pose.pose.position.x, msg.pose.pose.position.y, msg.pose.pose.position.z = anchor[0], anchor[1], 0.0
            msg.pose.pose.orientation.x = msg.pose.pose.orientation.y = msg.pose.pose.orientation.z = 0.0
            msg.pose.pose.orientation.w = 1.0
            msg.pose.covariance = [anchor[2]] + [0.0] * 35
            node._anchor_cb(msg)
        slam_now[0] = (sx, sy)
        n0 = len(published)
        # move the SLAM camera a little every tick so the relay's "SLAM frozen" detector stays quiet
        node._read_slam_pose_raw = lambda k=len(rows): (0.02 * k, 0.0, 0.02 * k, 0.0, 0.0, 0.0, 1.0, True)
        node._tick_slam_encoder(0, gx, gy, 0.0, 0.0, 0.0, 0.0, 1.0)
        if len(published) == n0:
            rows.append(dict(t=t, anchor=anchor, slam=[sx, sy], enc=None, nav=None, regime=None))      # init tick
            continue
        rows.append(dict(t=t, anchor=anchor, slam=[sx, sy], enc=[float(node.enc_x), float(node.enc_y)], nav=list(published[-1]),
                         regime=node._last_regime, streak=int(node.anchor_strong_streak),
                         alpha=(float(node._alpha_noanchor) if node._last_regime == "no_anchor" else None),
                         staleness=float(node._last_anchor_staleness), std=float(node._last_anchor_std)))
    with open(OUT, "w") as f:
        json.dump(dict(rows=rows, thresholds=dict(stale=node.ANCHOR_STALE_S, strong=node.ANCHOR_STRONG_STD, ok=node.ANCHOR_OK_STD,
                                                   hysteresis=node.ANCHOR_HYSTERESIS_N)), f)
    regs = {}
    for r in rows:
        regs[r["regime"]] = regs.get(r["regime"], 0) + 1
    print("wrote", OUT, regs, "alphas", sorted({r.get("alpha") for r in rows if r.get("alpha") is not None}))


if __name__ == "__main__":
    main()
