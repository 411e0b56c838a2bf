"""Teach-time landmark recorder without ROS: the logic of the reference node
`VisualLandmarkRecorder` (simulation/isaac/scripts/common/visual_landmark_recorder.py:211-325).

`LandmarkRecorderCore.tick(bgr, depth_mm, base_pose, rgb_ts)` evaluates the displacement trigger,
extracts ORB features through a cv2-shaped module (the HIP shim by default), applies the
border / ground / depth-range / local-depth-variance gates, back-projects the survivors and appends
one record.  `save()` writes landmarks.pkl in the reference's schema.
"""
from __future__ import annotations

import math

import numpy as np

from . import pose as P
from .landmarks import new_database, save_landmarks

FX = FY = 320.0
CX, CY = 320.0, 240.0
W, H = 640, 480
DEPTH_MIN_M = 0.5
DEPTH_MAX_M = 15.0
DEPTH_VAR_MAX_M = 0.30
GROUND_Y_THRESHOLD = 180
MIN_RECORD_KPTS = 30


def local_depth_std(depth_mm, uu, vv):
    """std of the 3x3 depth patch around each (u, v) over pixels > 0.01 m, 999 when fewer than three
    are valid (R:262-266, vectorised: the reference loops per keypoint in Python)."""
    d = depth_mm.astype(np.float32) / 1000.0
    patches = np.stack([d[vv + dy, uu + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1)], axis=1)  # (n, 9)
    valid = patches > 0.01
    cnt = valid.sum(axis=1)
    out = np.full(len(uu), 999.0, np.float32)
    for i in np.nonzero(cnt >= 3)[0]:          # per-row std in the same float32 arithmetic as ndarray.std()
        out[i] = patches[i][valid[i]].std()
    return out


class LandmarkRecorderCore:
    def __init__(self, out_pkl=None, min_disp_m: float = 2.0, cv2=None, nfeatures: int = 500, logger=None, engine=None):
        """engine: when given, ORB + all per-keypoint gates + back-projection run in ONE device call
        (reloc_record_frame); otherwise the gates run in NumPy on the cv2-shaped module's features."""
        self.engine = engine
        self.nfeatures = nfeatures
        if cv2 is None and engine is None:
            from . import cv2_shim as cv2
        self.cv2 = cv2
        self.out_pkl = out_pkl
        self.min_disp_m = float(min_disp_m)
        self.orb = cv2.ORB_create(nfeatures=nfeatures) if cv2 is not None else None
        self.landmarks = []
        self.last_landmark_pose_world = None
        self.log = logger or (lambda msg: None)

    def tick(self, bgr, depth_mm, base_pose, rgb_ts=0.0):
        """Returns the appended record, or None when nothing was recorded this tick."""
        if bgr is None or depth_mm is None or base_pose is None:
            return None
        cam_pose = P.base_to_cam_world(*base_pose)
        if self.last_landmark_pose_world is not None:
            lx, ly = self.last_landmark_pose_world[0], self.last_landmark_pose_world[1]
            if math.hypot(cam_pose[0] - lx, cam_pose[1] - ly) < self.min_disp_m:
                return None
        if self.engine is not None:
            r = self.engine.record_frame(bgr, depth_mm, self.nfeatures)
            if r["n"] < MIN_RECORD_KPTS:
                return None
            rec = {"pose": cam_pose, "descriptors": r["desc"], "keypoints_2d": r["xy"], "keypoints_3d_cam": r["pts3d"],
                   "ts": rgb_ts, "n_features": int(r["n"])}
            self.landmarks.append(rec)
            self.last_landmark_pose_world = cam_pose
            return rec
        cv2 = self.cv2
        gray = cv2.cvtColor(bgr, cv2.COLOR_BGR2GRAY)
        kpts, desc = self.orb.detectAndCompute(gray, None)
        if desc is None or len(kpts) == 0:
            return None
        xy = np.array([k.pt for k in kpts], dtype=np.float32)
        uu = np.round(xy[:, 0]).astype(np.int32)
        vv = np.round(xy[:, 1]).astype(np.int32)
        hh, ww = depth_mm.shape
        keep = (uu >= 1) & (uu < ww - 1) & (vv >= 1) & (vv < hh - 1) & (vv > GROUND_Y_THRESHOLD)
        uu, vv, xy, desc = uu[keep], vv[keep], xy[keep], desc[keep]
        z = depth_mm[vv, uu].astype(np.float32) / 1000.0
        zstd = local_depth_std(depth_mm, uu, vv)
        ok = (z > DEPTH_MIN_M) & (z < DEPTH_MAX_M) & (zstd < DEPTH_VAR_MAX_M)
        if ok.sum() < MIN_RECORD_KPTS:
            return None
        uu, vv, z = uu[ok], vv[ok], z[ok]
        pts3 = np.stack([(uu - CX) * z / FX, (vv - CY) * z / FY, z], axis=-1).astype(np.float32)
        rec = {"pose": cam_pose, "descriptors": desc[ok], "keypoints_2d": xy[ok], "keypoints_3d_cam": pts3,
               "ts": rgb_ts, "n_features": int(len(pts3))}
        self.landmarks.append(rec)
        self.last_landmark_pose_world = cam_pose
        return rec

    def database(self) -> dict:
        return new_database(self.landmarks, W, H, FX, FY, CX, CY)

    def save(self, path=None):
        path = path or self.out_pkl
        if not self.landmarks or not path:
            return None
        save_landmarks(path, self.database())
        return path
