"""cv2-shaped front end of the MI355X relocalization library.

Exposes exactly the OpenCV symbols the reference's teach/repeat nodes call:
    cvtColor, ORB_create(...).detectAndCompute / .detect, BFMatcher(...).match / .knnMatch,
    solvePnPRansac, projectPoints, Rodrigues, KeyPoint, DMatch, error and the constants,
with the same argument meaning, return shapes and error behaviour, so that
    import nclt_slam_project_amd.cv2_shim as cv2
drops into simulation/isaac/scripts/common/visual_landmark_matcher.py and
visual_landmark_recorder.py unchanged.  All arithmetic that is data-parallel runs in the HIP
library through `Engine`; there is no CPU fallback (a missing library or GPU raises `error`).

`Cv2Shim(backend)` takes any object with the Engine's method names; the module-level functions
bind to one lazily created HIP Engine.
"""
from __future__ import annotations

import math

import numpy as np

from ._native import RelocError

# ---- constants (values as in OpenCV 4.x) ----------------------------------------------------------
NORM_HAMMING = 6
NORM_HAMMING2 = 7
NORM_L2 = 4
COLOR_BGR2GRAY = 6
COLOR_RGB2GRAY = 7
SOLVEPNP_ITERATIVE = 0
SOLVEPNP_EPNP = 1
SOLVEPNP_P3P = 2
SOLVEPNP_AP3P = 5


class error(Exception):
    """Stands in for cv2.error: raised for malformed input (the reference catches it, M:328)."""


class KeyPoint:
    __slots__ = ("pt", "size", "angle", "response", "octave", "class_id")

    def __init__(self, x=0.0, y=0.0, size=0.0, angle=-1.0, response=0.0, octave=0, class_id=-1):
        self.pt = (float(x), float(y))
        self.size = float(size)
        self.angle = float(angle)
        self.response = float(response)
        self.octave = int(octave)
        self.class_id = int(class_id)

    def __repr__(self):
        return f"KeyPoint(pt={self.pt}, size={self.size:.1f}, angle={self.angle:.1f}, octave={self.octave})"


class DMatch:
    __slots__ = ("queryIdx", "trainIdx", "imgIdx", "distance")

    def __init__(self, queryIdx=-1, trainIdx=-1, distance=0.0, imgIdx=0):
        self.queryIdx = int(queryIdx)
        self.trainIdx = int(trainIdx)
        self.imgIdx = int(imgIdx)
        self.distance = float(distance)

    def __repr__(self):
        return f"DMatch({self.queryIdx}, {self.trainIdx}, {self.distance:.0f})"


def _rodrigues_matrix(rvec):
    r = np.asarray(rvec, np.float64).reshape(3)
    th = float(np.linalg.norm(r))
    if th < 1e-12:
        return np.eye(3)
    k = r / th
    K = np.array([[0.0, -k[2], k[1]], [k[2], 0.0, -k[0]], [-k[1], k[0], 0.0]])
    return np.eye(3) + math.sin(th) * K + (1.0 - math.cos(th)) * (K @ K)


class _ORB:
    def __init__(self, shim, nfeatures):
        self._shim = shim
        self._nfeatures = int(nfeatures)

    def detectAndCompute(self, image, mask=None):
        if mask is not None:
            raise error("detectAndCompute: masks are not supported (the reference passes None)")
        img = np.asarray(image)
        if img.dtype != np.uint8 or img.ndim != 2:
            raise error("detectAndCompute: expected a single-channel uint8 image")
        try:
            r = self._shim.backend.orb_detect_compute(img, self._nfeatures)
        except RelocError as e:
            raise error(str(e)) from e
        kps = tuple(KeyPoint(float(r["xy"][i, 0]), float(r["xy"][i, 1]), r["size"][i], r["angle"][i], r["response"][i],
                             r["octave"][i]) for i in range(r["n"]))
        return kps, (r["desc"] if r["n"] > 0 else None)

    def detect(self, image, mask=None):
        return self.detectAndCompute(image, mask)[0]

    def getMaxFeatures(self):
        return self._nfeatures


class _BFMatcher:
    def __init__(self, shim, normType, crossCheck):
        if normType not in (NORM_HAMMING,):
            raise error("BFMatcher: only NORM_HAMMING is implemented (the only norm the reference uses)")
        self._shim = shim
        self._cross = bool(crossCheck)

    @staticmethod
    def _check(q, t):
        q = np.asarray(q); t = np.asarray(t)
        if q.dtype != np.uint8 or t.dtype != np.uint8 or q.ndim != 2 or t.ndim != 2 or q.shape[1] != 32 or t.shape[1] != 32:
            raise error("BFMatcher: descriptors must be (N, 32) uint8 (ORB)")
        return q, t

    def match(self, queryDescriptors, trainDescriptors, mask=None):
        q, t = self._check(queryDescriptors, trainDescriptors)
        try:
            if self._cross:
                qi, ti, dd = self._shim.backend.match_mutual(q, t)
                return [DMatch(int(a), int(b), float(c)) for a, b, c in zip(qi, ti, dd)]
            idx, dist = self._shim.backend.match_knn2(q, t)
            return [DMatch(i, int(idx[i, 0]), float(dist[i, 0])) for i in range(len(q)) if idx[i, 0] >= 0]
        except RelocError as e:
            raise error(str(e)) from e

    def knnMatch(self, queryDescriptors, trainDescriptors, k=2, mask=None):
        if self._cross and k != 1:
            raise error("BFMatcher: crossCheck=True requires k == 1")
        if k not in (1, 2):
            raise error("knnMatch: k must be 1 or 2")
        q, t = self._check(queryDescriptors, trainDescriptors)
        try:
            idx, dist = self._shim.backend.match_knn2(q, t)
        except RelocError as e:
            raise error(str(e)) from e
        out = []
        for i in range(len(q)):
            row = [DMatch(i, int(idx[i, j]), float(dist[i, j])) for j in range(k) if idx[i, j] >= 0]
            out.append(row)
        return out


class Cv2Shim:
    """The cv2 surface over one backend (an Engine, or a test double with the same methods)."""

    NORM_HAMMING = NORM_HAMMING
    COLOR_BGR2GRAY = COLOR_BGR2GRAY
    COLOR_RGB2GRAY = COLOR_RGB2GRAY
    SOLVEPNP_ITERATIVE = SOLVEPNP_ITERATIVE
    SOLVEPNP_EPNP = SOLVEPNP_EPNP
    SOLVEPNP_P3P = SOLVEPNP_P3P
    SOLVEPNP_AP3P = SOLVEPNP_AP3P
    error = error
    KeyPoint = KeyPoint
    DMatch = DMatch

    def __init__(self, backend, ransac_seed: int = 0):
        self.backend = backend
        self.ransac_seed = int(ransac_seed)

    def cvtColor(self, src, code):
        img = np.asarray(src)
        if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 3:
            raise error("cvtColor: expected an (H, W, 3) uint8 image")
        if code not in (COLOR_BGR2GRAY, COLOR_RGB2GRAY):
            raise error("cvtColor: only COLOR_BGR2GRAY / COLOR_RGB2GRAY are implemented")
        try:
            return self.backend.gray(img, order_rgb=(code == COLOR_RGB2GRAY))
        except RelocError as e:
            raise error(str(e)) from e

    def ORB_create(self, nfeatures=500, **kwargs):
        defaults = dict(scaleFactor=1.2, nlevels=8, edgeThreshold=31, firstLevel=0, WTA_K=2, scoreType=0,
                        patchSize=31, fastThreshold=20)
        for k, v in kwargs.items():
            if k not in defaults or abs(float(v) - float(defaults[k])) > 1e-6:
                raise error(f"ORB_create: only OpenCV's default {k} is implemented (the reference passes nfeatures only)")
        return _ORB(self, nfeatures)

    def BFMatcher(self, normType=NORM_L2, crossCheck=False):
        return _BFMatcher(self, normType, crossCheck)

    def solvePnPRansac(self, objectPoints, imagePoints, cameraMatrix, distCoeffs, rvec=None, tvec=None,
                       useExtrinsicGuess=False, iterationsCount=100, reprojectionError=8.0, confidence=0.99,
                       inliers=None, flags=SOLVEPNP_ITERATIVE):
        """All accepted `flags` (ITERATIVE, EPNP, P3P, AP3P -- the reference's history switches between ITERATIVE and EPNP,
        M:342-348) select the SAME solver: P3P + 1 hypotheses scored by reprojection, Levenberg-Marquardt refinement on the
        inliers.  Anything this solver would silently ignore raises `error` instead: another flag, an
        extrinsic guess, lens distortion."""
        if flags not in (SOLVEPNP_ITERATIVE, SOLVEPNP_EPNP, SOLVEPNP_P3P, SOLVEPNP_AP3P):
            raise error(f"solvePnPRansac: flags={flags!r} is not implemented (ITERATIVE, EPNP, P3P and AP3P map to one solver)")
        if useExtrinsicGuess:
            raise error("solvePnPRansac: useExtrinsicGuess=True is not implemented (the reference never passes a guess)")
        obj = np.asarray(objectPoints, np.float32).reshape(-1, 3)
        img = np.asarray(imagePoints, np.float32).reshape(-1, 2)
        if len(obj) != len(img):
            raise error("solvePnPRansac: object/image point counts differ")
        if distCoeffs is not None and np.any(np.asarray(distCoeffs) != 0):
            raise error("solvePnPRansac: lens distortion is not implemented (the reference passes zeros)")
        K = np.asarray(cameraMatrix, np.float64).reshape(3, 3)
        K4 = np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]])
        if len(obj) < 4:
            return False, np.zeros((3, 1)), np.zeros((3, 1)), None
        try:
            ok, r, t, inl = self.backend.pnp_ransac(obj, img, K4=K4, iters=int(iterationsCount),
                                                    thr_px=float(reprojectionError), conf=float(confidence),
                                                    seed=self.ransac_seed)
        except RelocError as e:
            raise error(str(e)) from e
        if not ok:
            return False, np.zeros((3, 1)), np.zeros((3, 1)), None
        return True, r.reshape(3, 1).copy(), t.reshape(3, 1).copy(), inl.astype(np.int32).reshape(-1, 1)

    def projectPoints(self, objectPoints, rvec, tvec, cameraMatrix, distCoeffs=None, **_):
        obj = np.asarray(objectPoints, np.float64).reshape(-1, 3)
        K = np.asarray(cameraMatrix, np.float64).reshape(3, 3)
        R = _rodrigues_matrix(rvec)
        pc = obj @ R.T + np.asarray(tvec, np.float64).reshape(1, 3)
        uv = np.stack([K[0, 0] * pc[:, 0] / pc[:, 2] + K[0, 2], K[1, 1] * pc[:, 1] / pc[:, 2] + K[1, 2]], axis=1)
        return uv.reshape(-1, 1, 2), None

    def Rodrigues(self, src, **_):
        a = np.asarray(src, np.float64)
        if a.size == 3:
            return _rodrigues_matrix(a), None
        if a.shape == (3, 3):
            tr = max(-1.0, min(3.0, float(np.trace(a))))
            th = math.acos(max(-1.0, min(1.0, (tr - 1.0) / 2.0)))
            ax = np.array([a[2, 1] - a[1, 2], a[0, 2] - a[2, 0], a[1, 0] - a[0, 1]])
            n = float(np.linalg.norm(ax))
            r = np.zeros(3) if n < 1e-12 else ax / n * th
            return r.reshape(3, 1), None
        raise error("Rodrigues: expected a 3-vector or a 3x3 matrix")


# ---- module-level API bound to one lazily created HIP engine ---------------------------------------
_default = None


def default_shim() -> Cv2Shim:
    global _default
    if _default is None:
        from .engine import Engine
        try:
            _default = Cv2Shim(Engine())
        except RelocError as e:
            raise error(str(e)) from e
    return _default


def set_default_backend(backend, ransac_seed: int = 0):
    """Use an existing Engine (or compatible object) for the module-level cv2 functions."""
    global _default
    _default = Cv2Shim(backend, ransac_seed)


def cvtColor(src, code):
    return default_shim().cvtColor(src, code)


def ORB_create(nfeatures=500, **kw):
    return default_shim().ORB_create(nfeatures, **kw)


def BFMatcher(normType=NORM_L2, crossCheck=False):
    return default_shim().BFMatcher(normType, crossCheck)


def solvePnPRansac(*a, **kw):
    return default_shim().solvePnPRansac(*a, **kw)


def projectPoints(*a, **kw):
    return default_shim().projectPoints(*a, **kw)


def Rodrigues(*a, **kw):
    return default_shim().Rodrigues(*a, **kw)
