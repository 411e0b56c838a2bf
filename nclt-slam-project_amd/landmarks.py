"""landmarks.pkl <-> packed device arena.

On-disk schema written by the teach node (reference
simulation/isaac/scripts/common/visual_landmark_recorder.py:290-297, :319-325):
    {'intrinsics': {...}, 'base_to_cam_translation': [3], 'base_to_cam_rot': [[3x3]],
     'landmarks': [{'pose': (x,y,z,qx,qy,qz,qw), 'descriptors': u8 (n,32), 'keypoints_2d': f32 (n,2),
                    'keypoints_3d_cam': f32 (n,3), 'ts': float, 'n_features': int, ['accumulated': True]}]}
The packed form (what reloc_db_upload takes) is structure-of-arrays: descriptors (T,32) u8,
3-D points (T,3) f32, row offsets (L+1) i64, camera poses (L,7) f64.
"""
from __future__ import annotations

import io
import os
import pickle

import numpy as np

from .pose import BASE_TO_CAM_ROT, BASE_TO_CAM_TRANSLATION

_SAFE = {
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy", "ndarray"), ("numpy", "dtype"), ("numpy.core.multiarray", "scalar"),
    ("numpy._core.multiarray", "scalar"), ("numpy.core.numeric", "_frombuffer"),
    ("numpy._core.numeric", "_frombuffer"), ("builtins", "dict"), ("builtins", "list"), ("builtins", "tuple"),
    ("builtins", "float"), ("builtins", "int"), ("builtins", "bool"), ("builtins", "str"),
}


class _DataOnlyUnpickler(pickle.Unpickler):
    """landmarks.pkl holds only builtins and numpy arrays; anything else is refused, so loading a
    file from an untrusted source cannot execute code."""

    def find_class(self, module, name):
        if (module, name) in _SAFE:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"landmarks.pkl: refusing to load {module}.{name}")


def load_landmarks(path: str) -> dict:
    with open(path, "rb") as f:
        data = _DataOnlyUnpickler(io.BytesIO(f.read())).load()
    if not isinstance(data, dict) or "landmarks" not in data:
        raise ValueError(f"{path}: not a landmarks.pkl (no 'landmarks' key)")
    data.setdefault("base_to_cam_translation", BASE_TO_CAM_TRANSLATION.tolist())
    data.setdefault("base_to_cam_rot", BASE_TO_CAM_ROT.tolist())
    return data


def save_landmarks(path: str, data: dict) -> None:
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(path, "wb") as f:
        pickle.dump(data, f)


def new_database(landmarks, width=640, height=480, fx=320.0, fy=320.0, cx=320.0, cy=240.0) -> dict:
    return {
        "intrinsics": {"fx": fx, "fy": fy, "cx": cx, "cy": cy, "width": width, "height": height},
        "base_to_cam_translation": BASE_TO_CAM_TRANSLATION.tolist(),
        "base_to_cam_rot": BASE_TO_CAM_ROT.tolist(),
        "landmarks": list(landmarks),
    }


def pack_landmarks(landmarks):
    """list of record dicts -> (desc (T,32) u8, pts3d (T,3) f32, offsets (L+1) i64, poses (L,7) f64).
    Records whose descriptors are None contribute zero rows (the matcher skips them, M:321)."""
    L = len(landmarks)
    counts = np.zeros(L, np.int64)
    for i, lm in enumerate(landmarks):
        d = lm.get("descriptors")
        counts[i] = 0 if d is None else len(d)
    off = np.zeros(L + 1, np.int64)
    off[1:] = np.cumsum(counts)
    T = int(off[-1])
    desc = np.zeros((T, 32), np.uint8)
    pts = np.zeros((T, 3), np.float32)
    poses = np.zeros((L, 7), np.float64)
    for i, lm in enumerate(landmarks):
        poses[i] = np.asarray(lm["pose"], np.float64)
        if counts[i]:
            desc[off[i]:off[i + 1]] = np.asarray(lm["descriptors"], np.uint8)
            pts[off[i]:off[i + 1]] = np.asarray(lm["keypoints_3d_cam"], np.float32)
    return desc, pts, off, poses


def unpack_landmarks(desc, pts3d, offsets, poses, keypoints_2d=None):
    out = []
    for i in range(len(poses)):
        a, b = int(offsets[i]), int(offsets[i + 1])
        out.append({
            "pose": tuple(float(v) for v in poses[i]),
            "descriptors": np.ascontiguousarray(desc[a:b]),
            "keypoints_2d": (np.zeros((b - a, 2), np.float32) if keypoints_2d is None
                             else np.ascontiguousarray(keypoints_2d[a:b])),
            "keypoints_3d_cam": np.ascontiguousarray(pts3d[a:b]),
            "ts": 0.0,
            "n_features": b - a,
        })
    return out


def split_landmarks(data: dict):
    """Outbound / return split at the record with the largest x (reference
    simulation/isaac/experiments/69_repeat_road_split_landmarks_accel_noise/scripts/split_landmarks.py:18-37):
    outbound = records [0, i_peak], return = the rest; both keep every other top-level key."""
    lms = data["landmarks"]
    if not lms:
        raise ValueError("no landmarks to split")
    i_peak = max(range(len(lms)), key=lambda i: lms[i]["pose"][0])
    head = {k: v for k, v in data.items() if k != "landmarks"}
    return ({**head, "landmarks": lms[: i_peak + 1]}, {**head, "landmarks": lms[i_peak + 1:]}, i_peak)


def shard_by_rows(offsets, n_shards: int):
    """Contiguous record ranges with balanced descriptor counts: returns
    n_shards+1 record boundaries."""
    offsets = np.asarray(offsets, np.int64)
    L = len(offsets) - 1
    total = int(offsets[-1])
    bounds = [0]
    for s in range(1, n_shards):
        target = total * s / n_shards
        b = int(np.searchsorted(offsets, target, side="left"))
        bounds.append(min(max(b, bounds[-1]), L))
    bounds.append(L)
    return np.asarray(bounds, np.int64)
