"""Offline self-match harness: the reference's only test of the hot path
(simulation/isaac/experiments/55_visual_teach_repeat/scripts/checkpoint_a_selftest.py:39-113, 116-198),
driven by in-memory samples instead of a bag directory.

For every sampled teach record: ORB on the record's own frame -> candidates by distance (first 5 within
8 m, no heading test, S:54-57) -> knnMatch(desc_curr, desc_t, k=2) + Lowe 0.80 (query = CURRENT, train =
teach, the opposite of the live matcher, S:68-71) -> solvePnPRansac / reprojection gate / pose composition
(S:78-96) -> best by inliers.  Pass bar: >= 90 % of the samples self-localise within 0.3 m (S:171, 196-198).
"""
from __future__ import annotations

import math

import numpy as np

from . import pose as P
from .matcher import MatcherConfig

LOWE_RATIO = 0.80
PASS_FRACTION = 0.90
PASS_DIST_M = 0.30


def run_matcher_self(frame_bgr, vio_xy, landmarks, xy, cv2, cfg: MatcherConfig | None = None):
    """one self-match attempt; returns a dict with 'outcome' and, when ok, n_in / reproj / anchor_xy / teach_idx"""
    cfg = cfg or MatcherConfig()
    gray = cv2.cvtColor(frame_bgr, cv2.COLOR_BGR2GRAY)
    orb = cv2.ORB_create(nfeatures=cfg.nfeatures)
    bf = cv2.BFMatcher(cv2.NORM_HAMMING, crossCheck=False)
    kpts, desc = orb.detectAndCompute(gray, None)
    if desc is None or len(kpts) < cfg.min_matches:
        return {"outcome": "curr_no_features"}
    pts2d = np.array([k.pt for k in kpts], dtype=np.float32)
    d = np.linalg.norm(np.asarray(xy) - np.array(vio_xy), axis=1)
    order = np.lexsort((np.arange(len(d)), d))
    cand = [int(i) for i in order[: cfg.max_candidates] if d[i] < cfg.candidate_radius_m]
    if not cand:
        return {"outcome": "no_candidates"}
    dist0 = np.zeros((4, 1), dtype=np.float32)
    best = None
    for li in cand:
        lm = landmarks[li]
        desc_t = lm["descriptors"]
        if desc_t is None or len(desc_t) < cfg.min_matches:
            continue
        try:
            knn = bf.knnMatch(desc, desc_t, k=2)
        except cv2.error:
            continue
        good = [p[0] for p in knn if len(p) == 2 and p[0].distance < LOWE_RATIO * p[1].distance]
        if len(good) < cfg.min_matches:
            continue
        obj = np.asarray(lm["keypoints_3d_cam"], np.float32)[[m.trainIdx for m in good]]
        img = pts2d[[m.queryIdx for m in good]]
        ok, rvec, tvec, inl = cv2.solvePnPRansac(obj, img, cfg.K, dist0, iterationsCount=cfg.ransac_iterations,
                                                 reprojectionError=cfg.ransac_reproj_px, flags=cv2.SOLVEPNP_ITERATIVE)
        if not ok or inl is None or len(inl) < cfg.min_inliers:
            continue
        sel = inl[:, 0]
        proj, _ = cv2.projectPoints(obj[sel], rvec, tvec, cfg.K, dist0)
        err = float(np.linalg.norm(proj.reshape(-1, 2) - img[sel], axis=1).mean())
        if err > cfg.reproj_max_px:
            continue
        R_ct, _ = cv2.Rodrigues(rvec)
        t_tc = -R_ct.T @ np.asarray(tvec, np.float64).reshape(3)
        tp = lm["pose"]
        t_wc = np.array(tp[:3], np.float64) + P.quat_to_rot(*tp[3:7]) @ t_tc
        if best is None or len(inl) > best["n_in"]:
            best = {"n_in": len(inl), "reproj": err, "anchor_xy": (float(t_wc[0]), float(t_wc[1])), "teach_idx": li}
    if best is None:
        return {"outcome": "no_pnp_accept", "n_cand": len(cand)}
    return {"outcome": "ok", "n_cand": len(cand), **best}


def selftest(samples, database, cv2):
    """samples: iterable of (frame_bgr, record_index).  Returns (passed, summary dict)."""
    landmarks = database["landmarks"]
    xy = np.array([[lm["pose"][0], lm["pose"][1]] for lm in landmarks])
    n = n_ok = n_close = 0
    rows = []
    for frame, idx in samples:
        tp = landmarks[idx]["pose"]
        r = run_matcher_self(frame, (tp[0], tp[1]), landmarks, xy, cv2)
        n += 1
        if r["outcome"] == "ok":
            n_ok += 1
            err = math.hypot(r["anchor_xy"][0] - tp[0], r["anchor_xy"][1] - tp[1])
            n_close += err < PASS_DIST_M
            rows.append((idx, r["teach_idx"], r["n_in"], r["reproj"], err))
        else:
            rows.append((idx, -1, 0, 0.0, float("inf")))
    frac = n_close / n if n else 0.0
    return frac >= PASS_FRACTION, {"n": n, "n_ok": n_ok, "n_within": n_close, "fraction": frac, "rows": rows}
