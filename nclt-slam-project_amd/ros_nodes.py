"""rclpy wrappers with the reference's node names, constructor signatures, topics and CLI flags:

    VisualLandmarkMatcher(pkl_path, log_csv)            --landmarks --out-csv [--landmarks-return --swap-flag]
    VisualLandmarkRecorder(out_pkl, min_disp_m=2.0)     --out --min-disp

(reference simulation/isaac/scripts/common/visual_landmark_matcher.py:175-231,503-530,
 visual_landmark_recorder.py:154-179,375-392 and the exp-69 variant's extra flags).  They only move
data between ROS and the ROS-free cores in matcher.py / recorder.py; every feature operation runs in
the HIP library.  rclpy is imported lazily so the package imports on machines without ROS.
"""
from __future__ import annotations

import argparse
import signal
import sys

import numpy as np

from .matcher import FusedLandmarkMatcher, LandmarkMatcherCore, MatcherConfig
from .recorder import LandmarkRecorderCore

TICK_HZ = 2.0
POSE_FILE = "/tmp/isaac_pose.txt"
DRIFT_FILE = "/tmp/drift_est.txt"


def img_msg_to_bgr(msg):
    buf = np.frombuffer(msg.data, dtype=np.uint8).reshape(msg.height, msg.width, 3)
    if msg.encoding == "rgb8":
        return buf[:, :, ::-1].copy()
    if msg.encoding == "bgr8":
        return buf.copy()
    raise ValueError(f"unsupported rgb encoding {msg.encoding}")


def img_msg_to_depth_mm(msg):
    if msg.encoding in ("16UC1", "mono16"):
        return np.frombuffer(msg.data, dtype=np.uint16).reshape(msg.height, msg.width).copy()
    if msg.encoding == "32FC1":
        mm = np.frombuffer(msg.data, dtype=np.float32).reshape(msg.height, msg.width) * 1000.0
        return np.nan_to_num(mm, nan=0.0, posinf=0.0, neginf=0.0).astype(np.uint16)
    raise ValueError(f"unexpected depth encoding {msg.encoding}")


def read_pose_file(path=POSE_FILE):
    try:
        with open(path) as f:
            parts = f.readline().split()
        return tuple(float(p) for p in parts[:7]) if len(parts) >= 7 else None
    except Exception:
        return None


def read_drift(path=DRIFT_FILE):
    try:
        with open(path) as f:
            return float(f.readline().strip())
    except Exception:
        return 0.0


def _node_base():
    from rclpy.node import Node
    return Node


def make_matcher_node(pkl_path, log_csv, return_pkl=None, swap_flag=None, global_reloc=False, fused=False, cv2=None):
    """cv2: the cv2-shaped module the ROS-free core calls (default: the HIP shim); only the non-fused core uses it"""
    from geometry_msgs.msg import PoseWithCovarianceStamped
    from sensor_msgs.msg import Image
    Node = _node_base()

    class VisualLandmarkMatcher(Node):
        def __init__(self):
            super().__init__("visual_landmark_matcher")
            cfg = MatcherConfig(global_reloc=global_reloc)
            if fused:
                self.core = FusedLandmarkMatcher(pkl_path, log_csv, config=cfg, return_landmarks=return_pkl,
                                                 swap_flag=swap_flag, logger=lambda m: self.get_logger().info(m),
                                                 exclusive=True)          # one node, one camera: the only work on the GPU
            else:
                self.core = LandmarkMatcherCore(pkl_path, log_csv, cv2=cv2, config=cfg, return_landmarks=return_pkl,
                                                swap_flag=swap_flag, logger=lambda m: self.get_logger().info(m))
            self.last_rgb = self.last_depth = None
            self.create_subscription(Image, "/camera/color/image_raw", self._rgb_cb, 10)
            self.create_subscription(Image, "/camera/depth/image_rect_raw", self._depth_cb, 10)
            self.anchor_pub = self.create_publisher(PoseWithCovarianceStamped, "/anchor_correction", 10)
            self.timer = self.create_timer(1.0 / TICK_HZ, self._tick)
            signal.signal(signal.SIGTERM, self._sigterm)

        def _sigterm(self, *a):
            self.core.save_augmented()
            sys.exit(0)

        def _rgb_cb(self, msg):
            try:
                self.last_rgb = img_msg_to_bgr(msg)
            except Exception as e:
                self.get_logger().warn(f"rgb: {e}")

        def _depth_cb(self, msg):
            try:
                self.last_depth = img_msg_to_depth_mm(msg)
            except Exception as e:
                self.get_logger().warn(f"depth: {e}")

        def _tick(self):
            if self.last_rgb is None or self.last_depth is None:
                return
            pose = read_pose_file()
            if pose is None:
                return
            if fused:
                o = self.core.tick(self.last_rgb, pose, depth_mm=self.last_depth, drift_est=read_drift())
            else:
                o = self.core.tick(self.last_rgb, self.last_depth, pose, drift_est=read_drift())
            if o is None or not o.published:
                return
            msg = PoseWithCovarianceStamped()
            msg.header.frame_id = "map"
            msg.header.stamp = self.get_clock().now().to_msg()
            p, q = msg.pose.pose.position, msg.pose.pose.orientation
            p.x, p.y, p.z = o.anchor_pose[0], o.anchor_pose[1], o.anchor_pose[2]
            q.x, q.y, q.z, q.w = o.anchor_pose[3], o.anchor_pose[4], o.anchor_pose[5], o.anchor_pose[6]
            msg.pose.covariance = o.covariance
            self.anchor_pub.publish(msg)

    return VisualLandmarkMatcher()


def make_recorder_node(out_pkl, min_disp_m=2.0, cv2=None):
    from sensor_msgs.msg import Image
    Node = _node_base()

    class VisualLandmarkRecorder(Node):
        def __init__(self):
            super().__init__("visual_landmark_recorder")
            self.core = LandmarkRecorderCore(out_pkl, min_disp_m, cv2=cv2)
            self.last_rgb = self.last_depth = None
            self.last_rgb_ts = 0.0
            self.create_subscription(Image, "/camera/color/image_raw", self._rgb_cb, 10)
            self.create_subscription(Image, "/camera/depth/image_rect_raw", self._depth_cb, 10)
            self.timer = self.create_timer(0.2, self._tick)
            signal.signal(signal.SIGTERM, self._save_and_exit)
            signal.signal(signal.SIGINT, self._save_and_exit)

        def _rgb_cb(self, msg):
            try:
                self.last_rgb = img_msg_to_bgr(msg)
                self.last_rgb_ts = msg.header.stamp.sec + msg.header.stamp.nanosec * 1e-9
            except Exception as e:
                self.get_logger().warn(f"rgb cb: {e}")

        def _depth_cb(self, msg):
            try:
                self.last_depth = img_msg_to_depth_mm(msg)
            except Exception as e:
                self.get_logger().warn(f"depth cb: {e}")

        def _tick(self):
            self.core.tick(self.last_rgb, self.last_depth, read_pose_file(), self.last_rgb_ts)

        def _save_and_exit(self, *a):
            self.core.save()
            sys.exit(0)

    return VisualLandmarkRecorder()


def matcher_main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--landmarks", required=True)
    ap.add_argument("--out-csv", required=True)
    ap.add_argument("--landmarks-return", default=None)
    ap.add_argument("--swap-flag", default="/tmp/matcher_swap_return.txt")
    ap.add_argument("--global-reloc", action="store_true")
    ap.add_argument("--fused", action="store_true", help="run the whole tick in one device call")
    args = ap.parse_args(argv)
    import rclpy
    rclpy.init()
    node = make_matcher_node(args.landmarks, args.out_csv, args.landmarks_return, args.swap_flag, args.global_reloc, args.fused)
    try:
        rclpy.spin(node)
    except KeyboardInterrupt:
        pass
    finally:
        node.core.save_augmented()
        node.destroy_node()
        try:
            rclpy.shutdown()
        except Exception:
            pass


def recorder_main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--min-disp", type=float, default=2.0)
    args = ap.parse_args(argv)
    import rclpy
    rclpy.init()
    node = make_recorder_node(args.out, args.min_disp)
    try:
        rclpy.spin(node)
    except KeyboardInterrupt:
        pass
    finally:
        node.core.save()
        node.destroy_node()
        try:
            rclpy.shutdown()
        except Exception:
            pass
