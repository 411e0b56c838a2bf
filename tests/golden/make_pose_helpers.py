#!/usr/bin/env python3
"""Generates tests/golden/pose_helpers.npz by importing the REFERENCE's own pure-NumPy helpers.

Run in the build container only (needs /root/reference; the GPU box never has it):
    python tests/golden/make_pose_helpers.py

The reference modules import rclpy / sensor_msgs / geometry_msgs / cv2 at module scope; none of
those is installed here, so empty stand-in modules are registered in sys.modules for the import
only.  Nothing from cv2 or ROS is executed: the functions captured are plain NumPy/math
(visual_landmark_matcher.py:115-172, visual_landmark_recorder.py:137-151) plus the module-level
constants.  The output is data (inputs and expected outputs), not reference source.
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = "/root/reference/simulation/isaac/scripts/common"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pose_helpers.npz")


def _stub_modules():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class Node:  # minimal base class so `class X(Node)` parses
        def __init__(self, *a, **k):
            pass

    mod("rclpy")
    mod("rclpy.node", Node=Node)
    mod("sensor_msgs")
    mod("sensor_msgs.msg", Image=type("Image", (), {}))
    mod("geometry_msgs")
    mod("geometry_msgs.msg", PoseWithCovarianceStamped=type("PoseWithCovarianceStamped", (), {}))
    mod("cv2", NORM_HAMMING=6, COLOR_BGR2GRAY=6, SOLVEPNP_ITERATIVE=0, error=Exception)


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    _stub_modules()
    M = _load(os.path.join(REF, "visual_landmark_matcher.py"), "ref_matcher")
    R = _load(os.path.join(REF, "visual_landmark_recorder.py"), "ref_recorder")
    rng = np.random.default_rng(20260501)
    n = 64
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    # force every branch of rot_to_quat: near-180-degree rotations about x, y, z
    q[0] = [1, 0, 0, 1e-3]; q[1] = [0, 1, 0, 1e-3]; q[2] = [0, 0, 1, 1e-3]; q[3] = [0, 0, 0, 1]
    q[:4] /= np.linalg.norm(q[:4], axis=1, keepdims=True)
    t = rng.uniform(-50, 50, size=(n, 3))
    rot = np.stack([M.quat_to_rot(*qi) for qi in q])
    quat_back = np.array([M.rot_to_quat(Ri) for Ri in rot])
    cam_from_base = np.array([R.base_to_cam_world(*t[i], *q[i]) for i in range(n)])
    base_from_cam = np.array([
        M.cam_world_to_base_world(tuple(cam_from_base[i]), M.BASE_TO_CAM_TRANSLATION, M.BASE_TO_CAM_ROT)
        for i in range(n)])
    # std mapping at visual_landmark_matcher.py:400-405, restated as data: inliers -> std
    inl = np.arange(0, 60)
    std = np.array([0.05 if k >= 25 else (0.05 + 0.15 * (25 - k) / 10.0 if k >= 15 else 0.2) for k in inl])
    consts = dict(
        FX=M.FX, FY=M.FY, CX=M.CX, CY=M.CY, K=M.K, DIST=M.DIST,
        CANDIDATE_RADIUS_M=M.CANDIDATE_RADIUS_M, MAX_CANDIDATES=M.MAX_CANDIDATES,
        HEADING_TOL_DEG=M.HEADING_TOL_DEG, MIN_MATCHES=M.MIN_MATCHES, REPROJ_MAX_PX=M.REPROJ_MAX_PX,
        RANSAC_REPROJ_PX=M.RANSAC_REPROJ_PX, RANSAC_ITERATIONS=M.RANSAC_ITERATIONS,
        MIN_INLIERS=M.MIN_INLIERS, CONSISTENCY_M=M.CONSISTENCY_M, TICK_HZ=M.TICK_HZ,
        BASE_TO_CAM_TRANSLATION=M.BASE_TO_CAM_TRANSLATION, BASE_TO_CAM_ROT=M.BASE_TO_CAM_ROT,
        R_DEPTH_MIN_M=R.DEPTH_MIN_M, R_DEPTH_MAX_M=R.DEPTH_MAX_M, R_DEPTH_VAR_MAX_M=R.DEPTH_VAR_MAX_M,
        R_GROUND_Y_THRESHOLD=R.GROUND_Y_THRESHOLD, R_W=R.W, R_H=R.H,
    )
    np.savez(OUT, quat=q, trans=t, rot=rot, quat_back=quat_back, cam_from_base=cam_from_base,
             base_from_cam=base_from_cam, inliers=inl, std=std,
             **{"const_" + k: np.asarray(v) for k, v in consts.items()})
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
