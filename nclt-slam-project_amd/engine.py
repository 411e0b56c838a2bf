"""Engine: one reloc_ctx (one HIP stream, one optional landmark database) with numpy in / numpy out
methods.  This is the object the cv2-shaped shim, the matcher/recorder nodes and bench.py share.

Every method calls the HIP library through the C-ABI of include/reloc.h; nothing falls back to
a CPU implementation.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N

K4_DEFAULT = np.array([320.0, 320.0, 320.0, 240.0], dtype=np.float64)  # fx fy cx cy (reference M:49-52)

OUTCOME_NAMES = {0: "published", 1: "curr_no_features", 2: "no_candidates", 3: "no_pnp_accept",
                 4: "consistency_fail"}


class _PinnedBlock:
    """one hipHostMalloc allocation, freed with its last reference"""

    def __init__(self, lib, ptr):
        self._lib, self._ptr = lib, ptr

    def __del__(self):
        if self._ptr:
            try:
                self._lib.reloc_host_free(C.c_void_p(self._ptr))
            except Exception:          # interpreter shutdown
                pass
            self._ptr = None


class Engine:
    def __init__(self, device: int = 0, max_w: int = 1280, max_h: int = 720, max_feat: int = 8192):
        self._lib = N.load()
        self._ctx = self._lib.reloc_create(device, max_w, max_h, max_feat)
        if not self._ctx:
            raise N.RelocError("reloc_create failed: " + N.last_error())
        self.device = device
        self.max_w, self.max_h, self.max_feat = max_w, max_h, max_feat

    # ------------------------------------------------------------------ lifetime / plumbing
    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.reloc_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def ctx(self):
        return self._ctx

    def set_stream(self, hip_stream: int | None):
        N.check(self._lib.reloc_set_stream(self._ctx, C.c_void_p(hip_stream or 0)), "reloc_set_stream")

    def sync(self):
        N.check(self._lib.reloc_sync(self._ctx), "reloc_sync")

    def tick_wait(self):
        """wait for the result record of the last tick enqueued (polls its sequence stamp in pinned memory: reloc_tick_wait)"""
        N.check(self._lib.reloc_tick_wait(self._ctx), "reloc_tick_wait")

    def dev_alloc(self, nbytes: int) -> int:
        p = self._lib.reloc_dev_alloc(self._ctx, int(nbytes))
        if not p:
            raise N.RelocError("reloc_dev_alloc failed: " + N.last_error())
        return int(p)

    def dev_free(self, p: int):
        N.check(self._lib.reloc_dev_free(self._ctx, C.c_void_p(p)), "reloc_dev_free")

    def h2d(self, dst_dev: int, src: np.ndarray):
        src = np.ascontiguousarray(src)
        N.check(self._lib.reloc_h2d(self._ctx, C.c_void_p(dst_dev), N.ptr(src), src.nbytes), "reloc_h2d")
        self.sync()

    def h2d_async(self, dst_dev: int, src: np.ndarray):
        """enqueue only; `src` must stay alive and unchanged until the stream has passed the copy"""
        assert src.flags["C_CONTIGUOUS"]
        N.check(self._lib.reloc_h2d(self._ctx, C.c_void_p(dst_dev), N.ptr(src), src.nbytes), "reloc_h2d")

    def d2d(self, dst_dev: int, src_dev: int, nbytes: int):
        N.check(self._lib.reloc_d2d(self._ctx, C.c_void_p(dst_dev), C.c_void_p(src_dev), int(nbytes)), "reloc_d2d")

    def d2h_async(self, dst: np.ndarray, src_dev: int):
        """enqueue only; read `dst` after sync()"""
        assert dst.flags["C_CONTIGUOUS"]
        N.check(self._lib.reloc_d2h(self._ctx, N.ptr(dst), C.c_void_p(src_dev), dst.nbytes), "reloc_d2h")

    def pinned(self, shape, dtype=np.uint8) -> np.ndarray:
        """numpy array over page-locked host memory (hipHostMalloc): H2D / D2H copies from it are asynchronous DMA.
        The memory belongs to the ARRAY, not to this engine: it is released when the last view of it is collected (closing
        an engine frees nothing that another engine's ticks may still write).  Keep the array alive until every
        context told to write into it (tick_result_to) has been synchronised or pointed elsewhere."""
        dt = np.dtype(dtype)
        n = int(np.prod(shape)) * dt.itemsize
        p = self._lib.reloc_host_alloc(n)
        if not p:
            raise N.RelocError("reloc_host_alloc failed: " + N.last_error())
        buf = (C.c_uint8 * max(n, 1)).from_address(p)
        buf._owner = _PinnedBlock(self._lib, p)               # the views' base chain ends in `buf`, which holds the block
        return np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)

    @property
    def stream_ptr(self) -> int:
        return int(self._lib.reloc_get_stream(self._ctx) or 0)

    @property
    def tick_result_dev(self) -> int:
        return int(self._lib.reloc_tick_result_dev(self._ctx))

    def set_exclusive(self, on: bool | None):
        """deployment hint (no effect on results): this context is the only stream of work on the GPU -- kernels sized for
        the latency of one synchronous tick instead of for sharing the chip with other streams (reloc_set_exclusive)"""
        N.check(self._lib.reloc_set_exclusive(self._ctx, -1 if on is None else int(bool(on))), "reloc_set_exclusive")

    def tick_result_to(self, record: np.ndarray | None):
        """the ticks enqueued from now on also write their 96-byte result record into `record` (a slice of a pinned() array;
        None stops it): no copy, complete once the stream has passed the tick (reloc_tick_result_to)"""
        if record is None:
            N.check(self._lib.reloc_tick_result_to(self._ctx, None), "reloc_tick_result_to")
            return
        if record.nbytes < 96 or not record.flags.c_contiguous:
            raise N.RelocError("tick_result_to: need a contiguous record of at least 96 bytes")
        N.check(self._lib.reloc_tick_result_to(self._ctx, C.c_void_p(record.ctypes.data)), "reloc_tick_result_to")

    def d2h(self, dst: np.ndarray, src_dev: int):
        assert dst.flags["C_CONTIGUOUS"]
        N.check(self._lib.reloc_d2h(self._ctx, N.ptr(dst), C.c_void_p(src_dev), dst.nbytes), "reloc_d2h")
        self.sync()

    def to_device(self, a: np.ndarray) -> int:
        a = np.ascontiguousarray(a)
        p = self.dev_alloc(max(a.nbytes, 16))
        self.h2d(p, a)
        return p

    def timer_begin(self):
        N.check(self._lib.reloc_timer_begin(self._ctx), "reloc_timer_begin")

    def timer_end(self) -> float:
        ms = C.c_float()
        N.check(self._lib.reloc_timer_end(self._ctx, C.byref(ms)), "reloc_timer_end")
        return ms.value

    def profile_enable(self, on: bool):
        N.check(self._lib.reloc_profile_enable(self._ctx, int(on)), "reloc_profile_enable")

    def profile_get(self, which: int):
        ms = C.c_float(); n = C.c_int32()
        N.check(self._lib.reloc_profile_get(self._ctx, which, C.byref(ms), C.byref(n)), "reloc_profile_get")
        return ms.value, n.value

    # ------------------------------------------------------------------ ORB front end
    def gray(self, img: np.ndarray, order_rgb: bool = False) -> np.ndarray:
        img = N.u8(img)
        if img.ndim != 3 or img.shape[2] != 3:
            raise N.RelocError("gray: expected an (H, W, 3) uint8 image")
        h, w, _ = img.shape
        out = np.empty((h, w), np.uint8)
        N.check(self._lib.reloc_gray_u8(self._ctx, N.ptr(img), w, h, w * 3, int(order_rgb), N.ptr(out)), "reloc_gray_u8")
        return out

    def orb_detect_compute(self, gray: np.ndarray, nfeatures: int = 500):
        gray = N.u8(gray)
        if gray.ndim != 2:
            raise N.RelocError("detectAndCompute: expected an (H, W) uint8 image")
        h, w = gray.shape
        mf = self.max_feat
        xy = np.empty((mf, 2), np.float32); size = np.empty(mf, np.float32); ang = np.empty(mf, np.float32)
        resp = np.empty(mf, np.float32); octv = np.empty(mf, np.int32); desc = np.empty((mf, 32), np.uint8)
        n = C.c_int32()
        N.check(self._lib.reloc_orb_detect_compute(self._ctx, N.ptr(gray), w, h, w, int(nfeatures), N.ptr(xy), N.ptr(size),
                                                   N.ptr(ang), N.ptr(resp), N.ptr(octv), N.ptr(desc), C.byref(n)),
                "reloc_orb_detect_compute")
        k = n.value
        return dict(xy=xy[:k].copy(), size=size[:k].copy(), angle=ang[:k].copy(), response=resp[:k].copy(),
                    octave=octv[:k].copy(), desc=desc[:k].copy(), n=k)

    def record_frame(self, bgr: np.ndarray, depth_mm: np.ndarray, nfeatures: int = 500, order_rgb: bool = False):
        """teach-side record arrays of one frame: dict(xy (n,2), desc (n,32), pts3d (n,3), kp_index (n,), n, n_kp)"""
        bgr = N.u8(bgr)
        depth_mm = np.ascontiguousarray(depth_mm, np.uint16)
        h, w, _ = bgr.shape
        if depth_mm.shape != (h, w):
            raise N.RelocError("record_frame: depth and colour sizes differ")
        mf = self.max_feat
        xy = np.empty((mf, 2), np.float32); desc = np.empty((mf, 32), np.uint8); pts = np.empty((mf, 3), np.float32)
        idx = np.empty(mf, np.int32); n = C.c_int32(); nk = C.c_int32()
        N.check(self._lib.reloc_record_frame(self._ctx, N.ptr(bgr), N.ptr(depth_mm), w, h, int(order_rgb), int(nfeatures),
                                             N.ptr(xy), N.ptr(desc), N.ptr(pts), N.ptr(idx), C.byref(n), C.byref(nk)),
                "reloc_record_frame")
        k = n.value
        return dict(xy=xy[:k].copy(), desc=desc[:k].copy(), pts3d=pts[:k].copy(), kp_index=idx[:k].copy(), n=k, n_kp=nk.value)

    def frame_debug_plane(self, what: int, level: int) -> np.ndarray:
        buf = np.empty(self.max_w * self.max_h, np.uint8)
        w = C.c_int32(); h = C.c_int32()
        N.check(self._lib.reloc_frame_debug_plane(self._ctx, what, level, N.ptr(buf), C.byref(w), C.byref(h)),
                "reloc_frame_debug_plane")
        return buf[: w.value * h.value].reshape(h.value, w.value).copy()

    # ------------------------------------------------------------------ matching
    @staticmethod
    def _desc(a, name):
        a = np.asarray(a)
        if a.dtype != np.uint8 or a.ndim != 2 or a.shape[1] != 32:
            raise N.RelocError(f"{name}: descriptors must be (N, 32) uint8, got {a.dtype} {a.shape}")
        return np.ascontiguousarray(a)

    def match_mutual(self, q, t):
        q = self._desc(q, "match"); t = self._desc(t, "match")
        cap = max(min(len(q), len(t)), 1)
        qi = np.empty(cap, np.int32); ti = np.empty(cap, np.int32); dd = np.empty(cap, np.int32)
        n = C.c_int32()
        N.check(self._lib.reloc_match_mutual(self._ctx, N.ptr(q), len(q), N.ptr(t), len(t), N.ptr(qi), N.ptr(ti),
                                             N.ptr(dd), C.byref(n)), "reloc_match_mutual")
        return qi[: n.value].copy(), ti[: n.value].copy(), dd[: n.value].copy()

    def match_knn2(self, q, t):
        q = self._desc(q, "knnMatch"); t = self._desc(t, "knnMatch")
        idx = np.empty((len(q), 2), np.int32); dist = np.empty((len(q), 2), np.int32)
        N.check(self._lib.reloc_match_knn2(self._ctx, N.ptr(q), len(q), N.ptr(t), len(t), N.ptr(idx), N.ptr(dist)),
                "reloc_match_knn2")
        return idx, dist

    def db_upload(self, desc, pts3d, offsets, poses):
        desc = self._desc(desc, "db_upload") if len(desc) else np.zeros((0, 32), np.uint8)
        pts3d = np.ascontiguousarray(pts3d, np.float32).reshape(-1, 3)
        offsets = np.ascontiguousarray(offsets, np.int64)
        poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 7)
        if len(offsets) != len(poses) + 1 or offsets[-1] != len(desc) or len(pts3d) != len(desc):
            raise N.RelocError("db_upload: inconsistent array sizes")
        N.check(self._lib.reloc_db_upload(self._ctx, N.ptr(desc), N.ptr(pts3d), N.ptr(offsets), N.ptr(poses),
                                          len(poses)), "reloc_db_upload")

    def db_reserve(self, cap_records: int, cap_rows: int):
        """room for that many records / descriptor rows in the selected database without re-allocation"""
        N.check(self._lib.reloc_db_reserve(self._ctx, int(cap_records), int(cap_rows)), "reloc_db_reserve")

    def db_append(self, desc, pts3d, pose, keypoints_2d=None, index_xy=None):
        """one more record behind the last (self.landmarks.append + index update, M:489-493)"""
        desc = self._desc(desc, "db_append") if len(desc) else np.zeros((0, 32), np.uint8)
        pts3d = np.ascontiguousarray(pts3d, np.float32).reshape(-1, 3)
        if len(pts3d) != len(desc):
            raise N.RelocError("db_append: descriptor / point counts differ")
        kp = None if keypoints_2d is None else np.ascontiguousarray(keypoints_2d, np.float32).reshape(-1, 2)
        if kp is not None and len(kp) != len(desc):
            raise N.RelocError("db_append: descriptor / keypoint counts differ")
        pose = np.ascontiguousarray(pose, np.float64).reshape(7)
        ixy = None if index_xy is None else np.ascontiguousarray(index_xy, np.float64).reshape(2)
        N.check(self._lib.reloc_db_append(self._ctx, N.ptr(desc), N.ptr(pts3d), N.ptr(kp), len(desc), N.ptr(pose), N.ptr(ixy)),
                "reloc_db_append")

    def db_share(self, src: "Engine"):
        """scan the database resident in `src` (same device) instead of holding a copy; read-only through this engine"""
        N.check(self._lib.reloc_db_share(self._ctx, src._ctx), "reloc_db_share")
        self._db_owner = src        # keeps the owner alive

    def db_select(self, slot: int):
        """switch between the two resident databases (outbound / return leg, X:274-294)"""
        N.check(self._lib.reloc_db_select(self._ctx, int(slot)), "reloc_db_select")

    def db_fetch(self, record: int):
        """one record back from the device: dict(pose, descriptors, keypoints_2d, keypoints_3d_cam, index_xyh)"""
        n = C.c_int32()
        N.check(self._lib.reloc_db_fetch(self._ctx, int(record), None, None, None, None, None, C.byref(n)), "reloc_db_fetch")
        k = n.value
        desc = np.empty((max(k, 1), 32), np.uint8); pts = np.empty((max(k, 1), 3), np.float32); kp = np.empty((max(k, 1), 2), np.float32)
        pose = np.empty(7); xyh = np.empty(4)
        N.check(self._lib.reloc_db_fetch(self._ctx, int(record), N.ptr(desc), N.ptr(pts), N.ptr(kp), N.ptr(pose), N.ptr(xyh),
                                         C.byref(n)), "reloc_db_fetch")
        return dict(pose=tuple(float(v) for v in pose), descriptors=desc[:k].copy(), keypoints_2d=kp[:k].copy(),
                    keypoints_3d_cam=pts[:k].copy(), index_xyh=xyh, n_features=k)

    @property
    def db_records(self) -> int:
        return int(self._lib.reloc_db_records(self._ctx))

    @property
    def db_rows(self) -> int:
        return int(self._lib.reloc_db_rows(self._ctx))

    def db_match_counts(self, cur):
        cur = self._desc(cur, "db_match_counts") if len(cur) else np.zeros((0, 32), np.uint8)
        counts = np.empty(self.db_records, np.int32)
        N.check(self._lib.reloc_db_match_counts(self._ctx, N.ptr(cur), len(cur), N.ptr(counts)), "reloc_db_match_counts")
        return counts

    def db_match_counts_dev(self, cur_dev: int, n_cur: int, counts_dev: int, n_cur_dev: int = 0):
        N.check(self._lib.reloc_db_match_counts_dev(self._ctx, C.c_void_p(cur_dev), C.c_void_p(n_cur_dev), int(n_cur),
                                                    C.c_void_p(counts_dev)), "reloc_db_match_counts_dev")

    def db_ratio_counts(self, cur, ratio: float = 0.75):
        """per record: current descriptors passing the k=2 Lowe ratio test against the record's rows"""
        cur = self._desc(cur, "db_ratio_counts") if len(cur) else np.zeros((0, 32), np.uint8)
        counts = np.empty(self.db_records, np.int32)
        N.check(self._lib.reloc_db_ratio_counts(self._ctx, N.ptr(cur), len(cur), float(ratio), N.ptr(counts)),
                "reloc_db_ratio_counts")
        return counts

    def depth_points(self, depth, step: int = 4, K4=K4_DEFAULT, zmin: float = 0.3, zmax: float = 10.0):
        """(n, 3) float32 obstacle points (z, -x, -y) of every step-th valid depth pixel, raster order"""
        depth = np.ascontiguousarray(depth)
        if depth.dtype not in (np.float32, np.uint16) or depth.ndim != 2:
            raise N.RelocError("depth_points: expected an (H, W) float32 (metres) or uint16 (millimetres) image")
        h, w = depth.shape
        K4 = np.ascontiguousarray(K4, np.float64)
        pts = np.empty((-(-w // step) * -(-h // step), 3), np.float32)
        n = C.c_int32()
        N.check(self._lib.reloc_depth_points(self._ctx, N.ptr(depth), int(depth.dtype == np.float32), w, h, int(step), N.ptr(K4),
                                             float(zmin), float(zmax), N.ptr(pts), C.byref(n)), "reloc_depth_points")
        return pts[: n.value].copy()

    def hamming_matrix(self, a, b):
        a = self._desc(a, "hamming_matrix"); b = self._desc(b, "hamming_matrix")
        out = np.empty((len(a), len(b)), np.uint16)
        N.check(self._lib.reloc_hamming_matrix(self._ctx, N.ptr(a), len(a), N.ptr(b), len(b), N.ptr(out)),
                "reloc_hamming_matrix")
        return out

    def hamming_matrix_dev(self, a_dev: int, na: int, b_dev: int, nb: int, out_dev: int):
        N.check(self._lib.reloc_hamming_matrix_dev(self._ctx, C.c_void_p(a_dev), na, C.c_void_p(b_dev), nb,
                                                   C.c_void_p(out_dev)), "reloc_hamming_matrix_dev")

    # ------------------------------------------------------------------ PnP
    def pnp_score(self, obj, img, Rt, K4=K4_DEFAULT, thr_px=3.0, want_mask=False):
        obj = np.ascontiguousarray(obj, np.float32).reshape(-1, 3)
        img = np.ascontiguousarray(img, np.float32).reshape(-1, 2)
        Rt = np.ascontiguousarray(Rt, np.float64).reshape(-1, 12)
        K4 = np.ascontiguousarray(K4, np.float64)
        cnt = np.empty(len(Rt), np.int32)
        mask = np.empty((len(Rt), len(obj)), np.uint8) if want_mask else None
        N.check(self._lib.reloc_pnp_score(self._ctx, N.ptr(obj), N.ptr(img), len(obj), N.ptr(Rt), len(Rt), N.ptr(K4),
                                          float(thr_px), N.ptr(cnt), N.ptr(mask)), "reloc_pnp_score")
        return (cnt, mask) if want_mask else cnt

    def pnp_ransac(self, obj, img, K4=K4_DEFAULT, iters=200, thr_px=3.0, conf=0.99, seed=0):
        obj = np.ascontiguousarray(obj, np.float32).reshape(-1, 3)
        img = np.ascontiguousarray(img, np.float32).reshape(-1, 2)
        if len(obj) != len(img):
            raise N.RelocError("solvePnPRansac: object/image point counts differ")
        K4 = np.ascontiguousarray(K4, np.float64)
        rvec = np.zeros(3); tvec = np.zeros(3)
        inl = np.empty(max(len(obj), 1), np.int32)
        n = C.c_int32(); ok = C.c_int32()
        N.check(self._lib.reloc_pnp_ransac(self._ctx, N.ptr(obj), N.ptr(img), len(obj), N.ptr(K4), int(iters),
                                           float(thr_px), float(conf), int(seed), N.ptr(rvec), N.ptr(tvec), N.ptr(inl),
                                           C.byref(n), C.byref(ok)), "reloc_pnp_ransac")
        return bool(ok.value), rvec, tvec, inl[: n.value].copy()

    # ------------------------------------------------------------------ fused tick
    def get_params(self) -> "N.RelocParams":
        p = N.RelocParams()
        N.check(self._lib.reloc_get_params(self._ctx, C.byref(p)), "reloc_get_params")
        return p

    def set_params_from(self, other: "Engine"):
        """copy another engine's matcher parameters (the contexts of a batch must carry equal ones)"""
        p = other.get_params()
        N.check(self._lib.reloc_set_params(self._ctx, C.byref(p)), "reloc_set_params")

    def set_params(self, **kw):
        """matcher parameters of the fused tick (reloc_params in include/reloc.h); unnamed fields keep their value"""
        p = self.get_params()
        for k, v in kw.items():
            if not hasattr(p, k):
                raise N.RelocError(f"set_params: unknown parameter {k}")
            setattr(p, k, v)
        N.check(self._lib.reloc_set_params(self._ctx, C.byref(p)), "reloc_set_params")

    def tick_accumulate_dev(self, depth_mm_dev: int, w: int, h: int, base_pose, silence_ok: bool):
        bp = np.ascontiguousarray(base_pose, np.float64).reshape(7)
        N.check(self._lib.reloc_tick_accumulate_dev(self._ctx, C.c_void_p(depth_mm_dev), w, h, N.ptr(bp), int(silence_ok)),
                "reloc_tick_accumulate_dev")

    def accumulate_result(self):
        ap = C.c_int32(); n = C.c_int32(); d = C.c_double()
        N.check(self._lib.reloc_accumulate_result(self._ctx, C.byref(ap), C.byref(n), C.byref(d)), "reloc_accumulate_result")
        return dict(appended=bool(ap.value), n_kpts=n.value, nearest_m=d.value)

    def set_camera(self, K4=None, base_to_cam_t=None, base_to_cam_R=None):
        a = None if K4 is None else np.ascontiguousarray(K4, np.float64).reshape(4)
        b = None if base_to_cam_t is None else np.ascontiguousarray(base_to_cam_t, np.float64).reshape(3)
        c = None if base_to_cam_R is None else np.ascontiguousarray(base_to_cam_R, np.float64).reshape(9)
        N.check(self._lib.reloc_set_camera(self._ctx, N.ptr(a), N.ptr(b), N.ptr(c)), "reloc_set_camera")

    def tick_debug(self):
        ids = np.zeros(32, np.int32); n = C.c_int32(); nm = np.zeros(32, np.int32); ni = np.zeros(32, np.int32)
        ok = np.zeros(32, np.int32); rep = np.zeros(32); Rt = np.zeros((32, 12))
        N.check(self._lib.reloc_tick_debug(self._ctx, N.ptr(ids), C.byref(n), N.ptr(nm), N.ptr(ni), N.ptr(ok), N.ptr(rep),
                                           N.ptr(Rt)), "reloc_tick_debug")
        k = n.value
        return dict(cand_ids=ids[:k].copy(), n_matches=nm[:k].copy(), n_inliers=ni[:k].copy(), ok=ok[:k].copy(),
                    reproj=rep[:k].copy(), Rt=Rt[:k].copy())

    def tick(self, img, base_pose, order_rgb=False, global_reloc=False, seed=0):
        img = N.u8(img)
        h, w, _ = img.shape
        bp = np.ascontiguousarray(base_pose, np.float64).reshape(7)
        anchor = np.zeros(7); n_inl = C.c_int32(); rep = C.c_float(); lm = C.c_int32(); oc = C.c_int32(); nc = C.c_int32()
        N.check(self._lib.reloc_tick(self._ctx, N.ptr(img), w, h, int(order_rgb), N.ptr(bp), int(global_reloc), int(seed),
                                     N.ptr(anchor), C.byref(n_inl), C.byref(rep), C.byref(lm), C.byref(oc), C.byref(nc)),
                "reloc_tick")
        return dict(anchor_pose=anchor, n_inliers=n_inl.value, reproj=rep.value, lm_idx=lm.value, outcome=oc.value,
                    n_candidates=nc.value)

    def tick_dev(self, img_dev: int, w: int, h: int, base_pose, order_rgb=False, global_reloc=False, seed=0):
        """global_reloc: False / True or a RELOC_TICK_* mode (2 = local first, whole-database search if that finds nothing)"""
        bp = np.ascontiguousarray(base_pose, np.float64).reshape(7)
        N.check(self._lib.reloc_tick_dev(self._ctx, C.c_void_p(img_dev), w, h, int(order_rgb), N.ptr(bp),
                                         int(global_reloc), int(seed)), "reloc_tick_dev")

    @staticmethod
    def tick_batch_dev(engines, imgs_dev, w: int, h: int, base_poses, order_rgb=False, global_reloc=True, seeds=None):
        """n <= 8 frames through ONE whole-database scan launch: engines[i] processes imgs_dev[i]; the engines share a
        stream (set_stream) and a database (db_share).  Enqueue only; read engines[i].tick_result() afterwards."""
        n = len(engines)
        ctxs = (C.c_void_p * n)(*[e._ctx for e in engines])
        imgs = (C.c_void_p * n)(*[int(p) for p in imgs_dev])
        bp = np.ascontiguousarray(base_poses, np.float64).reshape(n, 7)
        sd = None if seeds is None else np.ascontiguousarray(seeds, np.uint64).reshape(n)
        lib = engines[0]._lib
        N.check(lib.reloc_tick_batch_dev(ctxs, n, imgs, w, h, int(order_rgb), N.ptr(bp), int(global_reloc), N.ptr(sd)),
                "reloc_tick_batch_dev")

    # ---- sharded database, batched halves with the exchange in device memory (reloc_shard_*_dev) ----
    @staticmethod
    def shard_scan_batch_dev(engines, imgs_dev, w: int, h: int, base_poses, k: int, id_base: int, scan_out_dev: int, order_rgb=False):
        """ORB per frame + ONE scan launch + per-frame ranking for n <= 8 engines that share a stream and a shard; row f of
        scan_out_dev (2k + 2 int32) = k global ids, k counts, feature count, 0.  Enqueue only."""
        n = len(engines)
        ctxs = (C.c_void_p * n)(*[e._ctx for e in engines])
        imgs = (C.c_void_p * n)(*[int(p) for p in imgs_dev])
        bp = None if base_poses is None else np.ascontiguousarray(base_poses, np.float64).reshape(n, 7)
        lib = engines[0]._lib
        N.check(lib.reloc_shard_scan_batch_dev(ctxs, n, imgs, int(w), int(h), int(order_rgb), N.ptr(bp), int(k), int(id_base),
                                               C.c_void_p(scan_out_dev)), "reloc_shard_scan_batch_dev")

    def shard_merge_dev(self, all_scan_dev: int, world: int, stride_rank: int, n: int, k: int, id_base: int, n_local: int,
                        win_gid_dev: int, cand_local_dev: int, n_feat_dev: int):
        """the merge every rank performs on the gathered rows, on this engine's stream (enqueue only)"""
        N.check(self._lib.reloc_shard_merge_dev(self._ctx, C.c_void_p(all_scan_dev), int(world), int(stride_rank), int(n), int(k),
                                                int(id_base), int(n_local), C.c_void_p(win_gid_dev), C.c_void_p(cand_local_dev),
                                                C.c_void_p(n_feat_dev)), "reloc_shard_merge_dev")

    @staticmethod
    def shard_solve_batch_dev(engines, cand_local_dev: int, k: int, base_poses, seeds, res_out: int):
        """matches + PnP + gates for the owned winners of every frame; result record f stored at res_out + 96 f (device
        or pinned host memory).  Enqueue only."""
        n = len(engines)
        ctxs = (C.c_void_p * n)(*[e._ctx for e in engines])
        bp = np.ascontiguousarray(base_poses, np.float64).reshape(n, 7)
        sd = None if seeds is None else np.ascontiguousarray(seeds, np.uint64).reshape(n)
        N.check(engines[0]._lib.reloc_shard_solve_batch_dev(ctxs, n, C.c_void_p(cand_local_dev), int(k), N.ptr(bp), N.ptr(sd),
                                                            C.c_void_p(res_out)), "reloc_shard_solve_batch_dev")

    def tick_scan_enqueue(self, img_dev: int, w: int, h: int, base_pose=None, k: int = 25, order_rgb=False):
        """ORB + whole-shard scan + local top-k, enqueued on the ctx stream; read with tick_scan_fetch(k)."""
        if not hasattr(self, "_topk_dev"):
            self._topk_dev = self.dev_alloc(2 * 32 * 4)
        bp = None if base_pose is None else np.ascontiguousarray(base_pose, np.float64).reshape(7)
        N.check(self._lib.reloc_tick_scan_dev(self._ctx, C.c_void_p(img_dev), w, h, int(order_rgb), N.ptr(bp),
                                              C.c_void_p(self._topk_dev), C.c_void_p(self._topk_dev + 128), int(k)),
                "reloc_tick_scan_dev")

    def orb_frame_dev(self, img_dev: int, w: int, h: int, stride: int | None = None, order_rgb=False, nfeatures: int = 500) -> int:
        """gray + ORB of an interleaved 3-channel frame resident in device memory (reloc_orb_frame_dev); returns the
        number of keypoints (synchronises).  Descriptors / coordinates stay on the device; frame_debug_plane() reads planes."""
        N.check(self._lib.reloc_orb_frame_dev(self._ctx, C.c_void_p(img_dev), int(w), int(h), int(stride or 3 * w),
                                              int(order_rgb), int(nfeatures)), "reloc_orb_frame_dev")
        nf = np.empty(1, np.int32)
        N.check(self._lib.reloc_d2h(self._ctx, N.ptr(nf), C.c_void_p(self._lib.reloc_frame_count_dev(self._ctx)), 4), "reloc_d2h")
        self.sync()
        return int(nf[0])

    def orb_features(self):
        """descriptors and coordinates of the frame last extracted on the device (orb_frame_dev / a tick), copied to the host"""
        nf = np.empty(1, np.int32)
        self.d2h(nf, int(self._lib.reloc_frame_count_dev(self._ctx)))
        n = min(int(nf[0]), self.max_feat)
        desc = np.empty((n, 32), np.uint8)
        xy = np.empty((n, 2), np.float32)
        if n:
            self.d2h(desc, int(self._lib.reloc_frame_desc_dev(self._ctx)))
            self.d2h(xy, int(self._lib.reloc_frame_xy_dev(self._ctx)))
        return dict(n=n, desc=desc, xy=xy)

    def tick_scan_into(self, img_dev: int, w: int, h: int, base_pose, k: int, ids_dev: int, counts_dev: int, nfeat_dev: int,
                       order_rgb=False):
        """ORB + shard scan + local top-k, everything left in caller-owned device memory (k ids, k counts, 1 feature count);
        enqueue only"""
        bp = None if base_pose is None else np.ascontiguousarray(base_pose, np.float64).reshape(7)
        N.check(self._lib.reloc_tick_scan_dev(self._ctx, C.c_void_p(img_dev), w, h, int(order_rgb), N.ptr(bp),
                                              C.c_void_p(ids_dev), C.c_void_p(counts_dev), int(k)), "reloc_tick_scan_dev")
        self.d2d(nfeat_dev, int(self._lib.reloc_frame_count_dev(self._ctx)), 4)

    def tick_solve_from(self, cand_ids_dev: int, n: int, base_pose, check_consistency: bool, seed: int = 0):
        """matches + PnP + gates for the LOCAL record ids listed in device memory (-1 entries skipped); enqueue only"""
        bp = np.ascontiguousarray(base_pose, np.float64).reshape(7)
        N.check(self._lib.reloc_tick_solve_dev(self._ctx, C.c_void_p(cand_ids_dev), int(n), N.ptr(bp), int(check_consistency),
                                               int(seed)), "reloc_tick_solve_dev")

    def tick_scan_fetch(self, k: int = 25):
        """(local record ids (k,), counts (k,), n_features) of the last enqueued scan, -1 / 0 padded (synchronises the stream)."""
        buf = np.empty(64, np.int32); nf = np.empty(1, np.int32)
        N.check(self._lib.reloc_d2h(self._ctx, N.ptr(buf), C.c_void_p(self._topk_dev), buf.nbytes), "reloc_d2h")
        N.check(self._lib.reloc_d2h(self._ctx, N.ptr(nf), C.c_void_p(self._lib.reloc_frame_count_dev(self._ctx)), 4), "reloc_d2h")
        self.sync()
        return buf[:k].copy(), buf[32:32 + k].copy(), int(nf[0])

    def tick_scan(self, img_dev: int, w: int, h: int, base_pose=None, k: int = 25, order_rgb=False):
        self.tick_scan_enqueue(img_dev, w, h, base_pose, k, order_rgb)
        return self.tick_scan_fetch(k)

    def tick_solve_enqueue(self, local_ids, base_pose, check_consistency: bool, seed: int = 0):
        """matches + PnP + gates for the listed LOCAL record ids against the features of the last scan; read with
        tick_result()."""
        ids = np.full(32, -1, np.int32)
        ids[: len(local_ids)] = local_ids
        if not hasattr(self, "_cand_dev"):
            self._cand_dev = self.dev_alloc(32 * 4)
        self.h2d(self._cand_dev, ids)
        bp = np.ascontiguousarray(base_pose, np.float64).reshape(7)
        N.check(self._lib.reloc_tick_solve_dev(self._ctx, C.c_void_p(self._cand_dev), len(local_ids), N.ptr(bp),
                                               int(check_consistency), int(seed)), "reloc_tick_solve_dev")

    def tick_solve(self, local_ids, base_pose, check_consistency: bool, seed: int = 0):
        self.tick_solve_enqueue(local_ids, base_pose, check_consistency, seed)
        return self.tick_result()

    def tick_result(self):
        anchor = np.zeros(7); n_inl = C.c_int32(); rep = C.c_double(); lm = C.c_int32(); oc = C.c_int32(); nc = C.c_int32()
        nf = C.c_int32(); rl = C.c_int32()
        N.check(self._lib.reloc_tick_result_ex(self._ctx, N.ptr(anchor), C.byref(n_inl), C.byref(rep), C.byref(lm),
                                               C.byref(oc), C.byref(nc), C.byref(nf), C.byref(rl)), "reloc_tick_result_ex")
        return dict(anchor_pose=anchor, n_inliers=n_inl.value, reproj=float(np.float32(rep.value)), lm_idx=lm.value,
                    outcome=oc.value, n_candidates=nc.value, n_features=nf.value, relocating=bool(rl.value))
