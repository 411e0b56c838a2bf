"""ctypes front end of the CPU oracle (oracle/_build/libreloc_oracle.so).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, by __graft_entry__.smoke() and by the
cpu_baseline leg of bench.py -- never by anything under nclt-slam-project_amd/ (the product
path), which must fail loudly when its HIP library is missing instead of falling back here.

The C files cite the reference call sites they restate; parity with OpenCV itself is UNPINNED
(see the header of orc_orb.c).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libreloc_oracle.so")
_FAST_PATH = os.path.join(_HERE, "_build", "libreloc_oracle_fast.so")
_lib = None
_libs = {}

NLEVELS = 8


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (seconds).  Safe to call repeatedly."""
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("orc_orb.c", "orc_match.c", "orc_pnp.c")
    ):
        subprocess.run(["make", "-C", _HERE] + (["-B"] if force else []), check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


def build_fast() -> str:
    """The timing-only -O3 -march=native build (bench.py cpu_baseline); compiled on the machine that runs it."""
    subprocess.run(["make", "-C", _HERE, "-B", "fast"], check=True, stdout=subprocess.DEVNULL)
    return _FAST_PATH


def select(fast: bool):
    """Route every call below to the strict build (default, what the tests check against) or the timing-only build."""
    global _lib
    key = "fast" if fast else "strict"
    if key not in _libs:
        if fast:
            build_fast()      # always, once per process: -march=native is for THIS machine, and a copy that travelled here may be stale
        _libs[key] = _load(_FAST_PATH if fast else build())
    _lib = _libs[key]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = _libs["strict"] = _load(_LIB_PATH)
    return _lib


def _load(path):
    so = C.CDLL(path)
    so.orc_harris_px.restype = C.c_float
    so.orc_ic_angle.restype = C.c_float
    so.orc_fast_atan2_deg.restype = C.c_float
    so.orc_fast_atan2_deg.argtypes = [C.c_float, C.c_float]
    so.orc_log_spec.restype = C.c_double
    so.orc_log_spec.argtypes = [C.c_double]
    so.orc_pnp_refine.restype = C.c_double
    so.orc_ransac_update_iters.argtypes = [C.c_double, C.c_double, C.c_int]
    so.orc_sincos_spec.argtypes = [C.c_float, C.c_void_p, C.c_void_p]
    return so


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


# ---------------------------------------------------------------- ORB front end
GRAY_DEFAULT_BITS = 15      # RELOC_GRAY_DEFAULT_BITS (include/reloc_spec.h): OpenCV 4.x's 8-bit coefficient set


def gray_u8(img, order_rgb=False, coeff_bits=GRAY_DEFAULT_BITS):
    img = _u8(img)
    h, w, _ = img.shape
    out = np.empty((h, w), np.uint8)
    rc = lib().orc_gray_u8_bits(_p(img), w, h, w * 3, int(order_rgb), _p(out), w, int(coeff_bits))
    assert rc == 0
    return out


def orb_layout(w, h, nfeatures=500):
    lw = np.zeros(NLEVELS, np.int32); lh = np.zeros(NLEVELS, np.int32)
    sc = np.zeros(NLEVELS, np.float32); q = np.zeros(NLEVELS, np.int32)
    lib().orc_orb_layout(w, h, nfeatures, _p(lw), _p(lh), _p(sc), _p(q))
    return lw, lh, sc, q


def resize_linear_exact(src, dw, dh):
    src = _u8(src)
    sh, sw = src.shape
    dst = np.empty((dh, dw), np.uint8)
    lib().orc_resize_linear_exact(_p(src), sw, sh, sw, _p(dst), dw, dh, dw)
    return dst


def resize_axis(src_n, dst_n):
    o = np.zeros(dst_n, np.int32); c = np.zeros(dst_n, np.int32)
    lib().orc_resize_axis(src_n, dst_n, _p(o), _p(c))
    return o, c


def pyramid(gray):
    gray = _u8(gray)
    h, w = gray.shape
    lw, lh, _, _ = orb_layout(w, h)
    off = np.zeros(NLEVELS + 1, np.int64)
    off[1:] = np.cumsum(lw.astype(np.int64) * lh)
    buf = np.zeros(int(off[-1]), np.uint8)
    lib().orc_orb_pyramid(_p(gray), w, h, w, _p(buf), _p(off))
    return [buf[off[l]:off[l + 1]].reshape(lh[l], lw[l]) for l in range(NLEVELS)]


def fast_score_map(img, thr=20):
    img = _u8(img)
    h, w = img.shape
    out = np.empty((h, w), np.uint8)
    lib().orc_fast_score_map(_p(img), w, h, w, thr, _p(out), w)
    return out


def fast_nms_map(score):
    score = _u8(score)
    h, w = score.shape
    out = np.empty((h, w), np.uint8)
    lib().orc_fast_nms_map(_p(score), w, h, w, _p(out), w)
    return out


def stage1_cut(hist, n_keep):
    hist = np.ascontiguousarray(hist, np.int32)
    return int(lib().orc_stage1_cut(_p(hist), int(n_keep)))


def harris_px(img, x, y):
    img = _u8(img)
    h, w = img.shape
    return float(lib().orc_harris_px(C.c_void_p(img.ctypes.data + y * w + x), w))


def ic_angle(img, x, y):
    img = _u8(img)
    h, w = img.shape
    return float(lib().orc_ic_angle(C.c_void_p(img.ctypes.data + y * w + x), w))


def ic_moments(img, x, y):
    img = _u8(img)
    h, w = img.shape
    m01 = C.c_int32(); m10 = C.c_int32()
    lib().orc_ic_moments(C.c_void_p(img.ctypes.data + y * w + x), w, C.byref(m01), C.byref(m10))
    return m01.value, m10.value


def fast_atan2_deg(y, x):
    return float(lib().orc_fast_atan2_deg(float(y), float(x)))


def sincos_spec(angle_deg):
    s = C.c_float(); c = C.c_float()
    lib().orc_sincos_spec(C.c_float(angle_deg), C.byref(s), C.byref(c))
    return s.value, c.value


def blur7(img):
    img = _u8(img)
    h, w = img.shape
    out = np.empty((h, w), np.uint8)
    rc = lib().orc_blur7(_p(img), w, h, w, _p(out), w)
    assert rc == 0
    return out


def brief(blurred, x, y, angle_deg):
    blurred = _u8(blurred)
    h, w = blurred.shape
    d = np.zeros(32, np.uint8)
    lib().orc_brief(C.c_void_p(blurred.ctypes.data + y * w + x), w, C.c_float(angle_deg), _p(d))
    return d


def orb_detect_compute(gray, nfeatures=500, max_out=20000, debug=False):
    """Returns dict(xy, size, angle, response, octave, xy_level, desc, n)."""
    gray = _u8(gray)
    h, w = gray.shape
    xy = np.zeros((max_out, 2), np.float32); size = np.zeros(max_out, np.float32)
    ang = np.zeros(max_out, np.float32); resp = np.zeros(max_out, np.float32)
    octv = np.zeros(max_out, np.int32); xyl = np.zeros((max_out, 2), np.int32)
    desc = np.zeros((max_out, 32), np.uint8)
    n = C.c_int32()
    s1 = np.zeros(NLEVELS, np.int32); cut = np.zeros(NLEVELS, np.int32)
    rc = lib().orc_orb_detect_compute(_p(gray), w, h, w, nfeatures, max_out, _p(xy), _p(size), _p(ang),
                                      _p(resp), _p(octv), _p(xyl), _p(desc), C.byref(n), _p(s1), _p(cut))
    assert rc == 0
    k = min(n.value, max_out)
    out = dict(xy=xy[:k], size=size[:k], angle=ang[:k], response=resp[:k], octave=octv[:k],
               xy_level=xyl[:k], desc=desc[:k], n=n.value)
    if debug:
        out["stage1_count"] = s1
        out["cut"] = cut
    return out


# ---------------------------------------------------------------- matching
def hamming_matrix(a, b):
    a = _u8(a); b = _u8(b)
    out = np.empty((a.shape[0], b.shape[0]), np.uint16)
    lib().orc_hamming_matrix(_p(a), C.c_int64(a.shape[0]), _p(b), C.c_int64(b.shape[0]), _p(out))
    return out


def match_mutual(q, t):
    q = _u8(q); t = _u8(t)
    nq, nt = q.shape[0], t.shape[0]
    cap = max(nq, 1)
    qi = np.zeros(cap, np.int32); ti = np.zeros(cap, np.int32); dd = np.zeros(cap, np.int32)
    n = C.c_int32()
    lib().orc_match_mutual(_p(q), nq, _p(t), nt, _p(qi), _p(ti), _p(dd), C.byref(n))
    return qi[:n.value], ti[:n.value], dd[:n.value]


def match_knn2(q, t):
    q = _u8(q); t = _u8(t)
    nq, nt = q.shape[0], t.shape[0]
    idx = np.zeros((nq, 2), np.int32); dist = np.zeros((nq, 2), np.int32)
    lib().orc_match_knn2(_p(q), nq, _p(t), nt, _p(idx), _p(dist))
    return idx, dist


def set_threads(n: int):
    """threads of the whole-database scan (the only multi-threaded oracle routine); default 1"""
    lib().orc_set_threads(int(n))


def set_single_pass(on: bool):
    """db_match_counts: evaluate each distance once (timing variant, bench.py cpu_baseline) instead of the literal two passes"""
    lib().orc_set_single_pass(int(on))


def db_match_counts(db, offsets, cur):
    db = _u8(db); cur = _u8(cur)
    offsets = np.ascontiguousarray(offsets, np.int64)
    n_rec = len(offsets) - 1
    counts = np.zeros(n_rec, np.int32)
    lib().orc_db_match_counts(_p(db), _p(offsets), C.c_int64(n_rec), _p(cur), cur.shape[0], _p(counts))
    return counts


def topk_records(counts, min_count, k):
    counts = np.ascontiguousarray(counts, np.int32)
    ids = np.zeros(k, np.int32)
    n = C.c_int32()
    lib().orc_topk_records(_p(counts), C.c_int64(len(counts)), int(min_count), int(k), _p(ids), C.byref(n))
    return ids[:n.value]


# ---------------------------------------------------------------- PnP
K4_DEFAULT = np.array([320.0, 320.0, 320.0, 240.0])


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def pnp_sample(seed, h, m):
    idx = np.zeros(4, np.int32)
    rc = lib().orc_pnp_sample(C.c_uint64(seed), h, m, _p(idx))
    return idx if rc == 0 else None


def p3p(P, xn):
    P = np.ascontiguousarray(P, np.float64); xn = np.ascontiguousarray(xn, np.float64)
    out = np.zeros((4, 12), np.float64)
    n = lib().orc_p3p(_p(P), _p(xn), _p(out))
    return out[:n]


def pnp_score(obj, img, Rt, K4=K4_DEFAULT, thr_px=3.0, want_mask=False):
    obj = _f32(obj); img = _f32(img)
    Rt = np.ascontiguousarray(Rt, np.float64).reshape(-1, 12)
    K4 = np.ascontiguousarray(K4, np.float64)
    m, H = obj.shape[0], Rt.shape[0]
    cnt = np.zeros(H, np.int32)
    mask = np.zeros((H, m), np.uint8) if want_mask else None
    lib().orc_pnp_score(_p(obj), _p(img), m, _p(Rt), H, _p(K4), C.c_float(thr_px), _p(cnt), _p(mask))
    return (cnt, mask) if want_mask else cnt


def pnp_hypothesis(obj, img, seed, h, K4=K4_DEFAULT):
    obj = _f32(obj); img = _f32(img)
    K4 = np.ascontiguousarray(K4, np.float64)
    Rt = np.zeros(12, np.float64)
    ok = lib().orc_pnp_hypothesis(_p(obj), _p(img), obj.shape[0], _p(K4), C.c_uint64(seed), h, _p(Rt))
    return Rt if ok else None


def ransac_select(count, m, conf=0.99):
    count = np.ascontiguousarray(count, np.int32)
    tried = C.c_int32()
    b = lib().orc_ransac_select(_p(count), len(count), m, C.c_double(conf), C.byref(tried))
    return int(b), tried.value


def pnp_refine(obj, img, sel, Rt, K4=K4_DEFAULT):
    obj = _f32(obj); img = _f32(img)
    sel = np.ascontiguousarray(sel, np.int32)
    Rt = np.ascontiguousarray(Rt, np.float64).copy()
    K4 = np.ascontiguousarray(K4, np.float64)
    cost = lib().orc_pnp_refine(_p(obj), _p(img), _p(sel), len(sel), _p(K4), _p(Rt))
    return Rt, float(cost)


def pnp_ransac(obj, img, K4=K4_DEFAULT, iters=200, thr_px=3.0, conf=0.99, seed=0):
    """(ok, rvec(3,), tvec(3,), inliers(k,), Rt(12,), best_h)"""
    obj = _f32(obj); img = _f32(img)
    K4 = np.ascontiguousarray(K4, np.float64)
    m = obj.shape[0]
    rvec = np.zeros(3); tvec = np.zeros(3); Rt = np.zeros(12)
    inl = np.zeros(max(m, 1), np.int32)
    n = C.c_int32(); ok = C.c_int32(); bh = C.c_int32()
    lib().orc_pnp_ransac(_p(obj), _p(img), m, _p(K4), iters, C.c_float(thr_px), C.c_double(conf),
                         C.c_uint64(seed), _p(rvec), _p(tvec), _p(inl), C.byref(n), C.byref(ok), _p(Rt),
                         C.byref(bh))
    return bool(ok.value), rvec, tvec, inl[:n.value].copy(), Rt, bh.value


def project_points(obj, rvec, tvec, K4=K4_DEFAULT):
    obj = _f32(obj)
    rvec = np.ascontiguousarray(rvec, np.float64).reshape(3)
    tvec = np.ascontiguousarray(tvec, np.float64).reshape(3)
    K4 = np.ascontiguousarray(K4, np.float64)
    uv = np.zeros((obj.shape[0], 2), np.float64)
    lib().orc_project_points(_p(obj), obj.shape[0], _p(rvec), _p(tvec), _p(K4), _p(uv))
    return uv


def rodrigues(rvec):
    rvec = np.ascontiguousarray(rvec, np.float64).reshape(3)
    R = np.zeros((3, 3), np.float64)
    lib().orc_rodrigues(_p(rvec), _p(R))
    return R


def rodrigues_log(R):
    R = np.ascontiguousarray(R, np.float64).reshape(3, 3)
    r = np.zeros(3, np.float64)
    lib().orc_rodrigues_log(_p(R), _p(r))
    return r


def db_ratio_counts(db, offsets, cur, ratio):
    db = _u8(db); cur = _u8(cur)
    offsets = np.ascontiguousarray(offsets, np.int64)
    n_rec = len(offsets) - 1
    counts = np.zeros(n_rec, np.int32)
    lib().orc_db_ratio_counts(_p(db), _p(offsets), C.c_int64(n_rec), _p(cur), cur.shape[0], C.c_double(ratio), _p(counts))
    return counts
