"""Repeat-time landmark matcher without ROS: the logic of the reference node
`VisualLandmarkMatcher` (simulation/isaac/scripts/common/visual_landmark_matcher.py), its
global-relocalisation variant (experiments/63_global_reloc/scripts/visual_landmark_matcher.py) and
its split-database variant (experiments/69_.../scripts/visual_landmark_matcher.py), driven by
explicit inputs instead of topics and /tmp files.

`LandmarkMatcherCore.tick(bgr, depth_mm, base_pose, ts)` performs one attempt exactly as
`_tick` does (M:281-433): candidate selection, ORB, per-candidate mutual match, PnP-RANSAC,
reprojection / inlier gates, pose composition, best by inliers, consistency gate, covariance, CSV
row, optional accumulation.  All feature work goes through a cv2-shaped module (by default the HIP
shim); `FusedLandmarkMatcher` runs the same tick through the single fused device call.
The rclpy node wrappers live in ros_nodes.py.
"""
from __future__ import annotations

import math
import os
import time
from dataclasses import dataclass, field

import numpy as np

from . import pose as P
from .landmarks import load_landmarks, pack_landmarks, save_landmarks

CSV_HEADER = "ts,vio_x,vio_y,candidates_tried,best_n_inliers,best_reproj_err,anchor_x,anchor_y,outcome\n"


@dataclass
class MatcherConfig:
    fx: float = 320.0
    fy: float = 320.0
    cx: float = 320.0
    cy: float = 240.0
    candidate_radius_m: float = 8.0
    max_candidates: int = 5
    heading_tol_deg: float = 90.0
    min_matches: int = 10
    reproj_max_px: float = 2.0
    ransac_reproj_px: float = 3.0
    ransac_iterations: int = 200
    min_inliers: int = 10
    consistency_m: float = 5.0
    nfeatures: int = 500
    gray_coeff_bits: int = 15      # cvtColor fixed point: 15 = OpenCV 4.x (what M:305 computes on OpenCV >= 4.8), 14 = OpenCV <= 3.x / SURVEY.md A.1
                                   # (include/reloc_spec.h); fused path only --
                                   # the cv2-shaped path takes it from its backend (Engine.set_params)
    # global relocalisation (variant G)
    global_reloc: bool = False
    reloc_age_s: float = 20.0
    reloc_drift_m: float = 3.0
    reloc_max_candidates: int = 25
    reloc_min_inliers: int = 18
    reloc_reproj_max_px: float = 1.5
    # accumulation scaffolding (M:85-89)
    accum_enable: bool = True
    accum_silence_s: float = 5.0
    accum_min_dist_m: float = 5.0
    accum_min_kpts: int = 30

    @property
    def K(self):
        return np.array([[self.fx, 0, self.cx], [0, self.fy, self.cy], [0, 0, 1]], dtype=np.float32)


@dataclass
class TickOutcome:
    ts: float
    vio_xy: tuple
    n_candidates: int
    n_inliers: int
    reproj_err: float | None
    anchor_pose: tuple | None
    outcome: str
    std: float | None = None
    covariance: list | None = None
    lm_idx: int | None = None
    relocating: bool = False
    published: bool = False
    extra: dict = field(default_factory=dict)


class LandmarkMatcherCore:
    def __init__(self, landmarks, log_csv=None, cv2=None, config: MatcherConfig | None = None,
                 return_landmarks=None, swap_flag=None, logger=None):
        """landmarks / return_landmarks: a landmarks.pkl path or an already loaded dict."""
        self.cfg = config or MatcherConfig()
        if cv2 is None:
            from . import cv2_shim as cv2  # HIP-backed module-level shim
        self.cv2 = cv2
        self.log = logger or (lambda msg: None)
        self.pkl_path = landmarks if isinstance(landmarks, str) else None
        self._return_src = return_landmarks
        self.swap_flag = swap_flag
        self._swapped = False
        self._adopt(load_landmarks(landmarks) if isinstance(landmarks, str) else landmarks)
        self.orb = cv2.ORB_create(nfeatures=self.cfg.nfeatures)
        self.matcher = cv2.BFMatcher(cv2.NORM_HAMMING, crossCheck=True)
        self.dist = np.zeros((4, 1), dtype=np.float32)
        self.last_anchor_ts = 0.0
        self.n_attempts = 0
        self.n_published = 0
        self.log_csv = log_csv
        if log_csv:
            d = os.path.dirname(log_csv)
            if d:
                os.makedirs(d, exist_ok=True)
            with open(log_csv, "w") as f:
                f.write(CSV_HEADER)

    # ------------------------------------------------------------------ database
    def _adopt(self, data):
        self.pkl_data = data
        self.landmarks = data["landmarks"]
        self.base_to_cam_t = np.array(data.get("base_to_cam_translation", P.BASE_TO_CAM_TRANSLATION))
        self.base_to_cam_R = np.array(data.get("base_to_cam_rot", P.BASE_TO_CAM_ROT))
        self.xy = np.array([[lm["pose"][0], lm["pose"][1]] for lm in self.landmarks], dtype=np.float64).reshape(-1, 2)
        self.heading = np.array([P.heading_of_camera_pose(lm["pose"], self.base_to_cam_R) for lm in self.landmarks])
        self.n_initial_landmarks = len(self.landmarks)
        self.n_accumulated = 0

    def maybe_swap_to_return(self):
        """Variant X: once the flag file exists, replace the outbound set by the return-leg set."""
        if self._swapped or self._return_src is None or not self.swap_flag or not os.path.exists(self.swap_flag):
            return False
        data = load_landmarks(self._return_src) if isinstance(self._return_src, str) else self._return_src
        if isinstance(self._return_src, str):
            self.pkl_path = self._return_src
        self._adopt(data)
        self._swapped = True
        self.log(f"[SWAP] return-leg landmarks loaded ({len(self.landmarks)})")
        return True

    def save_augmented(self):
        if self.n_accumulated > 0 and self.pkl_path:
            out = self.pkl_path.replace(".pkl", "_augmented.pkl")
            self.pkl_data["landmarks"] = self.landmarks
            save_landmarks(out, self.pkl_data)
            return out
        return None

    # ------------------------------------------------------------------ candidates
    def heading_errors(self, base_pose):
        cur = P.heading_of_base_pose(base_pose)
        d = self.heading - cur
        return np.abs(np.arctan2(np.sin(d), np.cos(d)))

    def select_candidates(self, base_pose):
        """nearest 3*max by VIO distance, then radius and heading filters, first max kept (M:293-302)."""
        cfg = self.cfg
        if len(self.landmarks) == 0:
            return [], np.zeros(0), np.zeros(0)
        d = np.linalg.norm(self.xy - np.array([base_pose[0], base_pose[1]]), axis=1)
        herr = self.heading_errors(base_pose)
        order = np.lexsort((np.arange(len(d)), d))       # (distance, index): a total order
        tol = math.radians(cfg.heading_tol_deg)
        cand = [int(i) for i in order[: cfg.max_candidates * 3] if d[i] < cfg.candidate_radius_m and herr[i] < tol]
        return cand[: cfg.max_candidates], d, herr

    def global_candidates(self, desc_curr, herr):
        """Variant G: mutual-match count of every heading-compatible record, top-N (G:329-344)."""
        cfg = self.cfg
        scored = []
        for li in np.where(herr < math.radians(cfg.heading_tol_deg))[0]:
            desc_t = self.landmarks[li]["descriptors"]
            if desc_t is None or len(desc_t) < cfg.min_matches:
                continue
            try:
                n = len(self.matcher.match(desc_t, desc_curr))
            except self.cv2.error:
                continue
            if n >= cfg.min_matches:
                scored.append((n, int(li)))
        scored.sort(reverse=True)
        return [li for _, li in scored[: cfg.reloc_max_candidates]]

    # ------------------------------------------------------------------ one attempt
    def _csv(self, o: TickOutcome):
        if not self.log_csv:
            return
        err = "" if o.reproj_err is None else f"{o.reproj_err:.2f}"
        ax = o.anchor_pose[0] if o.anchor_pose else ""
        ay = o.anchor_pose[1] if o.anchor_pose else ""
        with open(self.log_csv, "a") as f:
            f.write(f"{o.ts:.3f},{o.vio_xy[0]:.3f},{o.vio_xy[1]:.3f},{o.n_candidates},{o.n_inliers},{err},{ax},{ay},"
                    f"{o.outcome}\n")

    def solve_candidate(self, li, desc_curr, pts_curr_2d, relocating=False):
        """match + PnP + gates + pose composition for one record; returns (n_inl, err, base_pose) or None."""
        cfg, cv2 = self.cfg, self.cv2
        lm = self.landmarks[li]
        desc_t = lm["descriptors"]
        if desc_t is None or len(desc_t) < cfg.min_matches:
            return None
        try:
            good = self.matcher.match(desc_t, desc_curr)        # queryIdx = teach, trainIdx = current
        except cv2.error:
            return None
        if len(good) < cfg.min_matches:
            return None
        qi = np.fromiter((m.queryIdx for m in good), dtype=np.int64, count=len(good))
        ti = np.fromiter((m.trainIdx for m in good), dtype=np.int64, count=len(good))
        obj_pts = np.asarray(lm["keypoints_3d_cam"], dtype=np.float32)[qi]
        img_pts = np.asarray(pts_curr_2d, dtype=np.float32)[ti]
        ok, rvec, tvec, inliers = cv2.solvePnPRansac(
            obj_pts, img_pts, cfg.K, self.dist, iterationsCount=cfg.ransac_iterations,
            reprojectionError=cfg.ransac_reproj_px, flags=cv2.SOLVEPNP_ITERATIVE)
        min_inl = cfg.reloc_min_inliers if relocating else cfg.min_inliers
        max_err = cfg.reloc_reproj_max_px if relocating else cfg.reproj_max_px
        if not ok or inliers is None or len(inliers) < min_inl:
            return None
        sel = inliers[:, 0]
        proj, _ = cv2.projectPoints(obj_pts[sel], rvec, tvec, cfg.K, self.dist)
        err = float(np.linalg.norm(proj.reshape(-1, 2) - img_pts[sel], axis=1).mean())
        if err > max_err:
            return None
        # PnP gives the teach camera in the current camera frame; invert and chain with the teach pose
        R_ct, _ = cv2.Rodrigues(rvec)
        R_tc = R_ct.T
        t_tc = -R_tc @ np.asarray(tvec, np.float64).reshape(3)
        tp = lm["pose"]
        R_wt = P.quat_to_rot(tp[3], tp[4], tp[5], tp[6])
        t_wc = np.array(tp[:3], dtype=np.float64) + R_wt @ t_tc
        q = P.rot_to_quat(R_wt @ R_tc)
        cam_world = (float(t_wc[0]), float(t_wc[1]), float(t_wc[2]), *q)
        return len(inliers), err, P.cam_world_to_base_world(cam_world, self.base_to_cam_t, self.base_to_cam_R)

    def tick(self, bgr, depth_mm, base_pose, ts=None, drift_est=0.0):
        """One repeat attempt.  bgr: (H,W,3) u8; depth_mm: (H,W) u16 or None; base_pose: 7-tuple."""
        cfg, cv2 = self.cfg, self.cv2
        self.maybe_swap_to_return()
        if bgr is None or base_pose is None:
            return None
        ts = time.time() if ts is None else ts
        self.n_attempts += 1
        vio_xy = (base_pose[0], base_pose[1])
        cand, d, herr = self.select_candidates(base_pose)
        gray = cv2.cvtColor(bgr, cv2.COLOR_BGR2GRAY)
        kpts, desc = self.orb.detectAndCompute(gray, None)
        if desc is None or len(kpts) < cfg.min_matches:
            o = TickOutcome(ts, vio_xy, len(cand), 0, None, None, "curr_no_features")
            self._csv(o)
            return o
        pts2d = np.array([k.pt for k in kpts], dtype=np.float32)
        relocating = False
        if (cfg.global_reloc and not cand and (ts - self.last_anchor_ts) > cfg.reloc_age_s
                and drift_est > cfg.reloc_drift_m):
            cand = self.global_candidates(desc, herr)
            relocating = True
        if not cand:
            o = TickOutcome(ts, vio_xy, 0, 0, None, None, "no_candidates", relocating=relocating)
            self._csv(o)
            self.maybe_accumulate(base_pose, desc, pts2d, depth_mm, ts)
            return o
        best = None
        for li in cand:
            r = self.solve_candidate(li, desc, pts2d, relocating)
            if r is not None and (best is None or r[0] > best[0]):
                best = (*r, li)
        if best is None:
            o = TickOutcome(ts, vio_xy, len(cand), 0, None, None, "no_pnp_accept", relocating=relocating)
            self._csv(o)
            self.maybe_accumulate(base_pose, desc, pts2d, depth_mm, ts)
            return o
        n_inl, err, anchor, lm_idx = best
        shift = math.hypot(anchor[0] - vio_xy[0], anchor[1] - vio_xy[1])
        if not relocating and shift > cfg.consistency_m:
            o = TickOutcome(ts, vio_xy, len(cand), n_inl, err, anchor, f"consistency_fail_{shift:.1f}m", lm_idx=lm_idx)
            self._csv(o)
            self.maybe_accumulate(base_pose, desc, pts2d, depth_mm, ts)
            return o
        std = P.anchor_std(n_inl)
        self.n_published += 1
        self.last_anchor_ts = ts
        o = TickOutcome(ts, vio_xy, len(cand), n_inl, err, anchor, f"published_std{std:.2f}_shift{shift:.1f}", std=std,
                        covariance=P.anchor_covariance(std), lm_idx=lm_idx, relocating=relocating, published=True)
        self._csv(o)
        return o

    # ------------------------------------------------------------------ accumulation (M:435-500)
    def maybe_accumulate(self, base_pose, desc_curr, pts2d, depth_mm, ts):
        cfg = self.cfg
        if not cfg.accum_enable or ts - self.last_anchor_ts < cfg.accum_silence_s:
            return False
        if len(self.xy) and np.linalg.norm(self.xy - np.array([base_pose[0], base_pose[1]]), axis=1).min() < cfg.accum_min_dist_m:
            return False
        if depth_mm is None or len(pts2d) == 0:
            return False
        H, W = depth_mm.shape
        uu = np.round(pts2d[:, 0]).astype(np.int32)
        vv = np.round(pts2d[:, 1]).astype(np.int32)
        inside = (uu >= 1) & (uu < W - 1) & (vv >= 1) & (vv < H - 1)
        uu, vv, p2, dsc = uu[inside], vv[inside], pts2d[inside], desc_curr[inside]
        if len(uu) == 0:
            return False
        z = depth_mm[vv, uu].astype(np.float32) / 1000.0
        ok = (z > 0.5) & (z < 15.0)
        if ok.sum() < cfg.accum_min_kpts:
            return False
        uu, vv, z = uu[ok], vv[ok], z[ok]
        pts3 = np.stack([(uu - cfg.cx) * z / cfg.fx, (vv - cfg.cy) * z / cfg.fy, z], axis=-1).astype(np.float32)
        R_wb = P.quat_to_rot(*base_pose[3:7])
        c = np.array(base_pose[:3], dtype=np.float64) + R_wb @ self.base_to_cam_t
        q = P.rot_to_quat_scipy(R_wb @ self.base_to_cam_R)          # the reference converts with scipy here (M:478-479)
        rec = {"pose": (float(c[0]), float(c[1]), float(c[2]), *(float(v) for v in q)), "descriptors": dsc[ok],
               "keypoints_2d": p2[ok], "keypoints_3d_cam": pts3, "ts": ts, "n_features": int(len(pts3)),
               "accumulated": True}
        self.landmarks.append(rec)
        self.xy = np.vstack([self.xy.reshape(-1, 2), [base_pose[0], base_pose[1]]])
        self.heading = np.append(self.heading, P.heading_of_camera_pose(rec["pose"], self.base_to_cam_R))
        self.n_accumulated += 1
        return True


class FusedLandmarkMatcher:
    """The whole repeat session through the fused device calls: both landmark databases live in HBM (outbound set and,
    for the split variant X, the return-leg set), a tick uploads one frame (and the depth image when accumulation may
    fire), the device runs candidate search -- local, and the whole-database search of variant G when that finds
    nothing and G's silence / drift conditions hold (G:324-326) -- matching, PnP, gates, pose composition and the
    accumulation of M:435-500, and the host reads back one result record.  Produces the same TickOutcome / CSV rows as
    LandmarkMatcherCore and the reference node."""

    def __init__(self, landmarks, log_csv=None, engine=None, config: MatcherConfig | None = None, seed: int = 0,
                 return_landmarks=None, swap_flag=None, logger=None, exclusive: bool = False):
        """exclusive: this matcher is the only stream of work on the GPU (the reference's deployment: one node, one camera) --
        Engine.set_exclusive, kernels sized for the latency of one tick; results do not depend on it."""
        from .engine import Engine
        self.cfg = cfg = config or MatcherConfig()
        self.engine = e = engine or Engine()
        if exclusive:
            e.set_exclusive(True)
        self.log = logger or (lambda msg: None)
        self.pkl_path = landmarks if isinstance(landmarks, str) else None
        data = load_landmarks(landmarks) if isinstance(landmarks, str) else landmarks
        e.set_params(nfeatures=cfg.nfeatures, max_candidates=cfg.max_candidates, min_matches=cfg.min_matches,
                     min_inliers=cfg.min_inliers, ransac_iterations=cfg.ransac_iterations,
                     global_max_candidates=cfg.reloc_max_candidates, global_min_inliers=cfg.reloc_min_inliers,
                     accum_min_kpts=cfg.accum_min_kpts, candidate_radius_m=cfg.candidate_radius_m,
                     heading_tol_deg=cfg.heading_tol_deg, reproj_max_px=cfg.reproj_max_px,
                     ransac_reproj_px=cfg.ransac_reproj_px, consistency_m=cfg.consistency_m,
                     global_reproj_max_px=cfg.reloc_reproj_max_px, accum_min_dist_m=cfg.accum_min_dist_m,
                     gray_coeff_bits=cfg.gray_coeff_bits)
        e.set_camera([cfg.fx, cfg.fy, cfg.cx, cfg.cy], data.get("base_to_cam_translation", P.BASE_TO_CAM_TRANSLATION),
                     data.get("base_to_cam_rot", P.BASE_TO_CAM_ROT))
        self._return_src = return_landmarks
        self.swap_flag = swap_flag
        self._swapped = False
        self._return_data = None
        e.db_select(0)
        if return_landmarks is not None:
            # the return-leg set is resident from the start (slot 1); the swap is a pointer flip
            self._return_data = load_landmarks(return_landmarks) if isinstance(return_landmarks, str) else return_landmarks
            e.db_select(1)
            self._upload(self._return_data["landmarks"])
            e.db_select(0)
        self._adopt(data)
        self._upload(self.landmarks)
        self.seed = seed
        self.last_anchor_ts = 0.0
        self.n_attempts = self.n_published = 0
        self._img_dev = self._depth_dev = 0
        self._img_cap = self._depth_cap = 0
        self.log_csv = log_csv
        if log_csv:
            d = os.path.dirname(log_csv)
            if d:
                os.makedirs(d, exist_ok=True)
            with open(log_csv, "w") as f:
                f.write(CSV_HEADER)

    # ------------------------------------------------------------------ database
    def _upload(self, landmarks):
        e = self.engine
        desc, pts, off, poses = pack_landmarks(landmarks)
        # reserve for the accumulation of a session up front: appends then never re-allocate
        e.db_reserve(len(poses) + 256, int(off[-1]) + 256 * self.cfg.nfeatures)
        e.db_upload(desc, pts, off, poses)

    def _adopt(self, data):
        self.pkl_data = data
        self.landmarks = data["landmarks"]
        self.n_initial_landmarks = len(self.landmarks)
        self.n_accumulated = 0

    def maybe_swap_to_return(self):
        """Variant X (X:274-294): once the flag file exists the return-leg set replaces the outbound one."""
        if self._swapped or self._return_data is None or not self.swap_flag or not os.path.exists(self.swap_flag):
            return False
        self.engine.db_select(1)
        if isinstance(self._return_src, str):
            self.pkl_path = self._return_src
        self._adopt(self._return_data)
        self._swapped = True
        self.log(f"[SWAP] return-leg landmarks selected ({len(self.landmarks)})")
        return True

    def save_augmented(self):
        if self.n_accumulated > 0 and self.pkl_path:
            out = self.pkl_path.replace(".pkl", "_augmented.pkl")
            self.pkl_data["landmarks"] = self.landmarks
            save_landmarks(out, self.pkl_data)
            return out
        return None

    # ------------------------------------------------------------------ one attempt
    def _stage(self, which, arr):
        """device staging buffer of the frame / depth image, grown on demand; returns the device pointer"""
        e = self.engine
        ptr, cap = (self._img_dev, self._img_cap) if which == "img" else (self._depth_dev, self._depth_cap)
        if cap < arr.nbytes:
            if ptr:
                e.dev_free(ptr)
            ptr, cap = e.dev_alloc(arr.nbytes), arr.nbytes
            if which == "img":
                self._img_dev, self._img_cap = ptr, cap
            else:
                self._depth_dev, self._depth_cap = ptr, cap
        e.h2d_async(ptr, arr)
        return ptr

    def tick(self, bgr, base_pose, ts=None, global_reloc=None, depth_mm=None, drift_est=0.0):
        """One repeat attempt.  global_reloc: None = decide as the reference does (local candidates; the whole-database
        search only under G's trigger, when cfg.global_reloc is set), True = whole-database search unconditionally
        (benchmark shape), False = local only.  depth_mm enables accumulation (cfg.accum_enable)."""
        cfg, e = self.cfg, self.engine
        self.maybe_swap_to_return()
        if bgr is None or base_pose is None:
            return None
        ts = time.time() if ts is None else ts
        self.n_attempts += 1
        if global_reloc is None:
            trigger = cfg.global_reloc and (ts - self.last_anchor_ts) > cfg.reloc_age_s and drift_est > cfg.reloc_drift_m
            mode = 2 if trigger else 0
        else:
            mode = 1 if global_reloc else 0
        bgr = np.ascontiguousarray(bgr, np.uint8)
        h, w, _ = bgr.shape
        e.tick_dev(self._stage("img", bgr), w, h, base_pose, order_rgb=False, global_reloc=mode, seed=self.seed)
        accumulate = cfg.accum_enable and depth_mm is not None
        if accumulate:
            depth_mm = np.ascontiguousarray(depth_mm, np.uint16)
            if depth_mm.shape != (h, w):
                raise ValueError("depth and colour sizes differ")
            e.tick_accumulate_dev(self._stage("depth", depth_mm), w, h, base_pose,
                                  silence_ok=not (ts - self.last_anchor_ts < cfg.accum_silence_s))
        r = e.tick_result()
        if accumulate:
            acc = e.accumulate_result()
            if acc["appended"]:
                rec = e.db_fetch(e.db_records - 1)
                rec.pop("index_xyh")
                rec.update(ts=ts, accumulated=True)
                self.landmarks.append(rec)
                self.n_accumulated += 1
                self.log(f"[ACCUM #{self.n_accumulated}] new landmark at ({base_pose[0]:.1f},{base_pose[1]:.1f})  "
                         f"n_kpts={rec['n_features']}  nearest_existing={acc['nearest_m']:.1f}m")
        vio_xy = (base_pose[0], base_pose[1])
        oc, reloc = r["outcome"], r["relocating"]
        if oc == 1:
            o = TickOutcome(ts, vio_xy, r["n_candidates"], 0, None, None, "curr_no_features")
        elif oc == 2:
            o = TickOutcome(ts, vio_xy, 0, 0, None, None, "no_candidates", relocating=reloc)
        elif oc == 3:
            o = TickOutcome(ts, vio_xy, r["n_candidates"], 0, None, None, "no_pnp_accept", relocating=reloc)
        else:
            anchor = tuple(float(v) for v in r["anchor_pose"])
            shift = math.hypot(anchor[0] - vio_xy[0], anchor[1] - vio_xy[1])
            if oc == 4:
                o = TickOutcome(ts, vio_xy, r["n_candidates"], r["n_inliers"], r["reproj"], anchor,
                                f"consistency_fail_{shift:.1f}m", lm_idx=r["lm_idx"])
            else:
                std = P.anchor_std(r["n_inliers"])
                self.n_published += 1
                self.last_anchor_ts = ts
                o = TickOutcome(ts, vio_xy, r["n_candidates"], r["n_inliers"], r["reproj"], anchor,
                                f"published_std{std:.2f}_shift{shift:.1f}", std=std, covariance=P.anchor_covariance(std),
                                lm_idx=r["lm_idx"], relocating=reloc, published=True)
        if self.log_csv:
            err = "" if o.reproj_err is None else f"{o.reproj_err:.2f}"
            ax = o.anchor_pose[0] if o.anchor_pose else ""
            ay = o.anchor_pose[1] if o.anchor_pose else ""
            with open(self.log_csv, "a") as f:
                f.write(f"{o.ts:.3f},{vio_xy[0]:.3f},{vio_xy[1]:.3f},{o.n_candidates},{o.n_inliers},{err},{ax},{ay},"
                        f"{o.outcome}\n")
        return o
