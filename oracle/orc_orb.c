/* orc_orb.c -- CPU ORACLE (test infrastructure only) for the ORB front end.
 *
 * THIS FILE IS A CHECKER, NOT A PRODUCT PATH.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may build, load or call it.  The shipped path is the HIP
 * library under nclt-slam-project_amd/csrc/ and never links or imports anything from
 * oracle/.
 *
 * What it restates.  The reference (vbronetskyi/nclt-slam-project) computes features with
 *     gray = cv2.cvtColor(bgr, cv2.COLOR_BGR2GRAY)
 *     kpts, desc = cv2.ORB_create(nfeatures=500).detectAndCompute(gray, None)
 * at simulation/isaac/scripts/common/visual_landmark_matcher.py:305-306 and
 * simulation/isaac/scripts/common/visual_landmark_recorder.py:240-241.  The arithmetic is in
 * OpenCV (pip `opencv-python`, version unpinned by the reference; the only pin anywhere in
 * the tree is `opencv-python>=4.8.0` in datasets/nclt/requirements.txt:3).  OpenCV is not in
 * /root/reference, not installed here and not fetchable, so this file restates OpenCV 4.x's
 * published ORB algorithm (modules/features2d/src/orb.cpp, fast.cpp, fast_score.cpp;
 * imgproc color/resize/smooth bit-exact 8U paths) as summarised in SURVEY.md Appendix A.
 *
 * PARITY UNPINNED: the reference holds no golden vectors, known-answer tests or fixtures for
 * keypoints/descriptors, and cv2 cannot be run here.  The known-answer
 * tests in tests/ (literal per-pixel FAST, numpy Harris/blur/resize mirrors, planted corners)
 * pin this file to the algorithm description, not to OpenCV binaries.  Where OpenCV's result
 * order is implementation-defined (std::nth_element in KeyPointsFilter::retainBest) this file
 * fixes a total order: level-major, raster (y, then x) inside a level; the kept SET follows
 * OpenCV's rule (best n by response plus every tie of the n-th).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/reloc_orb_pattern.h"
#include "../include/reloc_spec.h"

#define ORC_API __attribute__((visibility("default")))

/* ---- a2: cv2.cvtColor(.., COLOR_BGR2GRAY) (visual_landmark_matcher.py:305) ------------- */
/* coeff_bits: 15 = OpenCV 4.x's gray_shift set (RELOC_GRAY_DEFAULT_BITS, the default of reloc_params.gray_coeff_bits), 14 = SURVEY.md A.1 */
ORC_API int orc_gray_u8_bits(const uint8_t *img, int w, int h, int stride, int order_rgb,
                             uint8_t *gray, int gstride, int coeff_bits)
{
    if (!img || !gray || w <= 0 || h <= 0) return -1;
    if (coeff_bits != RELOC_GRAY_SHIFT && coeff_bits != RELOC_GRAY15_SHIFT) return -2;
    const int c15 = coeff_bits == RELOC_GRAY15_SHIFT;
    const int cb = c15 ? RELOC_GRAY15_CB : RELOC_GRAY_CB, cg = c15 ? RELOC_GRAY15_CG : RELOC_GRAY_CG,
              cr = c15 ? RELOC_GRAY15_CR : RELOC_GRAY_CR;
    for (int y = 0; y < h; ++y) {
        const uint8_t *s = img + (size_t)y * stride;
        uint8_t *d = gray + (size_t)y * gstride;
        for (int x = 0; x < w; ++x) {
            int c0 = s[3 * x], c1 = s[3 * x + 1], c2 = s[3 * x + 2];
            int b = order_rgb ? c2 : c0, r = order_rgb ? c0 : c2;
            d[x] = (uint8_t)((b * cb + c1 * cg + r * cr + (1 << (coeff_bits - 1))) >> coeff_bits);
        }
    }
    return 0;
}

ORC_API int orc_gray_u8(const uint8_t *img, int w, int h, int stride, int order_rgb,
                        uint8_t *gray, int gstride)
{
    return orc_gray_u8_bits(img, w, h, stride, order_rgb, gray, gstride, RELOC_GRAY_DEFAULT_BITS);
}

/* ---- a4: pyramid geometry and per-level feature quotas ---------------- */
ORC_API int orc_orb_layout(int w, int h, int nfeatures, int32_t *lw, int32_t *lh, float *lscale,
                           int32_t *quota)
{
    const int nl = RELOC_ORB_NLEVELS;
    for (int l = 0; l < nl; ++l) {
        float s = (float)pow(RELOC_ORB_SCALE_FACTOR, (double)l);
        lscale[l] = s;
        lw[l] = (int32_t)lrintf((float)w / s);
        lh[l] = (int32_t)lrintf((float)h / s);
    }
    float factor = (float)(1.0 / RELOC_ORB_SCALE_FACTOR);
    float nper = (float)(nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nl)));
    int sum = 0;
    for (int l = 0; l < nl - 1; ++l) {
        quota[l] = (int32_t)lrintf(nper);
        sum += quota[l];
        nper *= factor;
    }
    quota[nl - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
    return 0;
}

/* INTER_LINEAR_EXACT restatement: source coordinate (d + 0.5) * (src/dst) - 0.5, the fraction
 * quantised to 8 bits, horizontal pass in 8.8 fixed point, vertical pass in 16.16, rounded
 * half up.  Taps that fall outside the source clamp to the edge pixel with full weight. */
static void resize_axis(int src_n, int dst_n, int32_t *ofs, int32_t *coef)
{
    double scale = (double)src_n / (double)dst_n;
    for (int d = 0; d < dst_n; ++d) {
        double f = ((double)d + 0.5) * scale - 0.5;
        int s = (int)floor(f);
        double a = f - (double)s;
        if (s < 0) { s = 0; a = 0.0; }
        if (s >= src_n - 1) { s = src_n - 1; a = 0.0; }
        ofs[d] = s;
        coef[d] = (int32_t)lrint(a * (double)(1 << RELOC_RESIZE_COEF_BITS));
    }
}

ORC_API int orc_resize_axis(int src_n, int dst_n, int32_t *ofs, int32_t *coef)
{
    resize_axis(src_n, dst_n, ofs, coef);
    return 0;
}

ORC_API int orc_resize_linear_exact(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst,
                                    int dw, int dh, int dstride)
{
    if (sw <= 0 || sh <= 0 || dw <= 0 || dh <= 0) return -1;
    int32_t *xo = malloc(sizeof(int32_t) * (size_t)dw), *xc = malloc(sizeof(int32_t) * (size_t)dw);
    int32_t *yo = malloc(sizeof(int32_t) * (size_t)dh), *yc = malloc(sizeof(int32_t) * (size_t)dh);
    resize_axis(sw, dw, xo, xc);
    resize_axis(sh, dh, yo, yc);
    const int one = 1 << RELOC_RESIZE_COEF_BITS;
    for (int y = 0; y < dh; ++y) {
        int y0 = yo[y], y1 = y0 + 1 < sh ? y0 + 1 : sh - 1;
        const uint8_t *r0 = src + (size_t)y0 * sstride, *r1 = src + (size_t)y1 * sstride;
        uint32_t b = (uint32_t)yc[y];
        for (int x = 0; x < dw; ++x) {
            int x0 = xo[x], x1 = x0 + 1 < sw ? x0 + 1 : sw - 1;
            uint32_t a = (uint32_t)xc[x];
            uint32_t h0 = r0[x0] * (one - a) + r0[x1] * a; /* 8.8  */
            uint32_t h1 = r1[x0] * (one - a) + r1[x1] * a;
            uint32_t v = h0 * (one - b) + h1 * b;          /* 16.16 */
            dst[(size_t)y * dstride + x] = (uint8_t)((v + (1u << 15)) >> 16);
        }
    }
    free(xo); free(xc); free(yo); free(yc);
    return 0;
}

/* ---- a5: FAST-9/16 score ---------------------------------------------- */
static const int8_t RING_DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int8_t RING_DY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

/* Largest t' for which the pixel is still a FAST-9 corner (strict test), 0 if it is not a
 * corner at `thr`.  = max over the 16 arcs of 9 contiguous ring pixels of the smallest
 * same-signed difference, minus 1. */
static int fast_score_px(const uint8_t *p, int stride, int thr)
{
    int v[16];
    const int c = p[0];
    for (int k = 0; k < 16; ++k) v[k] = (int)p[RING_DY[k] * stride + RING_DX[k]] - c;
    int best = 0;
    for (int s = 0; s < 16; ++s) {
        int mb = 255, md = 255;
        for (int j = 0; j < RELOC_FAST_ARC; ++j) {
            int d = v[(s + j) & 15];
            if (d < mb) mb = d;
            if (-d < md) md = -d;
        }
        if (mb > best) best = mb;
        if (md > best) best = md;
    }
    return best > thr ? best - 1 : 0;
}

/* Score of every pixel of the scan region [3, w-4] x [3, h-4]; 0 elsewhere. */
ORC_API int orc_fast_score_map(const uint8_t *img, int w, int h, int stride, int thr,
                               uint8_t *score, int sstride)
{
    for (int y = 0; y < h; ++y) memset(score + (size_t)y * sstride, 0, (size_t)w);
    for (int y = 3; y < h - 3; ++y)
        for (int x = 3; x < w - 3; ++x)
            score[(size_t)y * sstride + x] =
                (uint8_t)fast_score_px(img + (size_t)y * stride + x, stride, thr);
    return 0;
}

/* 3x3 non-maximum suppression (strictly greater than all 8 neighbours) restricted to the ORB
 * edge margin 31 <= x < w-31, 31 <= y < h-31.  Output map: score if kept else 0. */
ORC_API int orc_fast_nms_map(const uint8_t *score, int w, int h, int sstride, uint8_t *kept,
                             int kstride)
{
    const int e = RELOC_ORB_EDGE;
    for (int y = 0; y < h; ++y) memset(kept + (size_t)y * kstride, 0, (size_t)w);
    for (int y = e; y < h - e; ++y)
        for (int x = e; x < w - e; ++x) {
            const uint8_t *s = score + (size_t)y * sstride + x;
            int c = s[0];
            if (!c) continue;
            if (c > s[-1] && c > s[1] && c > s[-sstride - 1] && c > s[-sstride] &&
                c > s[-sstride + 1] && c > s[sstride - 1] && c > s[sstride] && c > s[sstride + 1])
                kept[(size_t)y * kstride + x] = (uint8_t)c;
        }
    return 0;
}

/* Cut score of KeyPointsFilter::retainBest(2*quota) with ties kept, from the histogram of the
 * NMS survivors; raised while the kept set exceeds RELOC_ORB_STAGE1_CAP. */
ORC_API int orc_stage1_cut(const int32_t hist[256], int n_keep)
{
    int cut = RELOC_FAST_THRESHOLD;
    long total = 0;
    for (int s = 0; s < 256; ++s) total += hist[s];
    if (total > n_keep) {
        long c = 0;
        for (int s = 255; s >= 0; --s) {
            c += hist[s];
            if (c >= n_keep) { cut = s; break; }
        }
    }
    for (;;) {
        long c = 0;
        for (int s = cut; s < 256; ++s) c += hist[s];
        if (c <= RELOC_ORB_STAGE1_CAP || cut >= 255) break;
        ++cut;
    }
    return cut;
}

/* ---- a6: Harris response, 7x7 block, k = 0.04 ------------------------- */
ORC_API float orc_harris_px(const uint8_t *p, int step)
{
    const int r = RELOC_HARRIS_BLOCK / 2;
    int a = 0, b = 0, c = 0;
    for (int dy = -r; dy <= r; ++dy)
        for (int dx = -r; dx <= r; ++dx) {
            const uint8_t *q = p + dy * step + dx;
            int ix = (q[1] - q[-1]) * 2 + (q[-step + 1] - q[-step - 1]) + (q[step + 1] - q[step - 1]);
            int iy = (q[step] - q[-step]) * 2 + (q[step - 1] - q[-step - 1]) + (q[step + 1] - q[-step + 1]);
            a += ix * ix;
            b += iy * iy;
            c += ix * iy;
        }
    volatile float scale = 1.f / ((1 << 2) * RELOC_HARRIS_BLOCK * 255.f);
    volatile float s2 = scale * scale;
    volatile float s3 = s2 * scale;
    volatile float s4 = s3 * scale;
    volatile float fa = (float)a, fb = (float)b, fc = (float)c;
    volatile float t1 = fa * fb;
    volatile float t2 = fc * fc;
    volatile float t3 = fa + fb;
    volatile float t4 = RELOC_HARRIS_K * t3;
    volatile float t5 = t4 * t3;
    volatile float t6 = t1 - t2;
    volatile float t7 = t6 - t5;
    return t7 * s4;
}

/* ---- a7: intensity-centroid orientation ------------------------------- */
static const int UMAX[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

static float fast_atan2_deg(float y, float x)
{
    volatile float ax = fabsf(x), ay = fabsf(y);
    volatile float a, c, c2, t;
    if (ax >= ay) {
        t = ax + RELOC_ATAN2_EPS;
        c = ay / t;
        c2 = c * c;
        t = RELOC_ATAN2_P7 * c2; t = t + RELOC_ATAN2_P5;
        t = t * c2;              t = t + RELOC_ATAN2_P3;
        t = t * c2;              t = t + RELOC_ATAN2_P1;
        a = t * c;
    } else {
        t = ay + RELOC_ATAN2_EPS;
        c = ax / t;
        c2 = c * c;
        t = RELOC_ATAN2_P7 * c2; t = t + RELOC_ATAN2_P5;
        t = t * c2;              t = t + RELOC_ATAN2_P3;
        t = t * c2;              t = t + RELOC_ATAN2_P1;
        t = t * c;
        a = 90.f - t;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

ORC_API float orc_fast_atan2_deg(float y, float x) { return fast_atan2_deg(y, x); }

ORC_API int orc_ic_moments(const uint8_t *center, int step, int32_t *m01_out, int32_t *m10_out)
{
    int m01 = 0, m10 = 0;
    for (int u = -RELOC_ORB_HALF_PATCH; u <= RELOC_ORB_HALF_PATCH; ++u) m10 += u * center[u];
    for (int v = 1; v <= RELOC_ORB_HALF_PATCH; ++v) {
        int vsum = 0, d = UMAX[v];
        for (int u = -d; u <= d; ++u) {
            int vp = center[u + v * step], vm = center[u - v * step];
            vsum += vp - vm;
            m10 += u * (vp + vm);
        }
        m01 += v * vsum;
    }
    *m01_out = m01;
    *m10_out = m10;
    return 0;
}

ORC_API float orc_ic_angle(const uint8_t *center, int step)
{
    int32_t m01, m10;
    orc_ic_moments(center, step, &m01, &m10);
    return fast_atan2_deg((float)m01, (float)m10);
}

/* ---- a8: GaussianBlur 7x7 sigma 2, 8-bit fixed point, BORDER_REFLECT_101 -------------- */
static int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        if (p >= n) p = 2 * n - 2 - p;
    }
    return p;
}

ORC_API int orc_blur7(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride)
{
    static const int K[7] = {RELOC_BLUR_K0, RELOC_BLUR_K1, RELOC_BLUR_K2, RELOC_BLUR_K3,
                             RELOC_BLUR_K2, RELOC_BLUR_K1, RELOC_BLUR_K0};
    uint16_t *tmp = malloc(sizeof(uint16_t) * (size_t)w * (size_t)h);
    if (!tmp) return -2;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            uint32_t s = 0;
            for (int k = 0; k < 7; ++k) s += (uint32_t)K[k] * src[(size_t)y * sstride + reflect101(x + k - 3, w)];
            tmp[(size_t)y * w + x] = (uint16_t)s;          /* 8.8, <= 255*256 */
        }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            uint32_t s = 0;
            for (int k = 0; k < 7; ++k) s += (uint32_t)K[k] * tmp[(size_t)reflect101(y + k - 3, h) * w + x];
            dst[(size_t)y * dstride + x] = (uint8_t)((s + (1u << 15)) >> 16);
        }
    free(tmp);
    return 0;
}

/* ---- a9: steered BRIEF-256 --------------------------------------------- */
/* cos/sin of the steering angle: OpenCV evaluates (float)cos((double)theta).  libm and the GPU
 * math library are both within an ulp of the true double value but not bit-identical to each
 * other, so the specification evaluates sine and cosine itself: Cody-Waite reduction by pi/2
 * and Taylor polynomials in IEEE double (no fused multiply-add), then one rounding to float.
 * The result differs from (float)cos(theta) only if the true value lies within ~1e-16 relative
 * of a float rounding boundary. */
static void sincos_spec(double th, float *s_out, float *c_out)
{
    static const double PIO2_HI = 1.57079632679489655800e+00; /* 0x3FF921FB54442D18 */
    static const double PIO2_LO = 6.12323399573676603587e-17; /* pi/2 - PIO2_HI       */
    static const double TWO_OVER_PI = 6.36619772367581382433e-01;
    volatile double kd = floor(th * TWO_OVER_PI + 0.5);
    int k = (int)kd;
    volatile double r = th - kd * PIO2_HI;
    r = r - kd * PIO2_LO;
    volatile double r2 = r * r;
    /* sin r = r * (1 + r2*(S1 + r2*(S2 + ...))),  cos r = 1 + r2*(C1 + r2*(C2 + ...)) */
    static const double S[8] = {-1.0 / 6.0, 1.0 / 120.0, -1.0 / 5040.0, 1.0 / 362880.0,
                                -1.0 / 39916800.0, 1.0 / 6227020800.0, -1.0 / 1307674368000.0,
                                1.0 / 355687428096000.0};
    static const double C[8] = {-1.0 / 2.0, 1.0 / 24.0, -1.0 / 720.0, 1.0 / 40320.0,
                                -1.0 / 3628800.0, 1.0 / 479001600.0, -1.0 / 87178291200.0,
                                1.0 / 20922789888000.0};
    volatile double ps = S[7], pc = C[7];
    for (int i = 6; i >= 0; --i) {
        ps = ps * r2; ps = ps + S[i];
        pc = pc * r2; pc = pc + C[i];
    }
    ps = ps * r2; ps = ps + 1.0; ps = ps * r;
    pc = pc * r2; pc = pc + 1.0;
    double sn, cs;
    switch (k & 3) {
    case 0: sn = ps; cs = pc; break;
    case 1: sn = pc; cs = -ps; break;
    case 2: sn = -ps; cs = -pc; break;
    default: sn = -pc; cs = ps; break;
    }
    *s_out = (float)sn;
    *c_out = (float)cs;
}

ORC_API int orc_sincos_spec(float angle_deg, float *s_out, float *c_out)
{
    volatile float th = angle_deg * RELOC_DEG2RAD_F;
    sincos_spec((double)th, s_out, c_out);
    return 0;
}

ORC_API int orc_brief(const uint8_t *center, int step, float angle_deg, uint8_t desc[32])
{
    float a, b; /* a = cos, b = sin */
    volatile float th = angle_deg * RELOC_DEG2RAD_F;
    sincos_spec((double)th, &b, &a);
    for (int j = 0; j < 32; ++j) {
        unsigned byte = 0;
        for (int i = 0; i < 8; ++i) {
            const signed char *t = RELOC_ORB_PATTERN + 4 * (8 * j + i);
            int val[2];
            for (int e = 0; e < 2; ++e) {
                volatile float px = (float)t[2 * e], py = (float)t[2 * e + 1];
                volatile float xa = px * a, yb = py * b, xb = px * b, ya = py * a;
                volatile float rx = xa - yb, ry = xb + ya;
                int ix = (int)lrintf(rx), iy = (int)lrintf(ry);
                val[e] = center[iy * step + ix];
            }
            byte |= (unsigned)(val[0] < val[1]) << i;
        }
        desc[j] = (uint8_t)byte;
    }
    return 0;
}

/* ---- a3-a10: the whole detectAndCompute --------------------------------------------------
 * Outputs are level-major, raster order inside a level.  Returns the number of keypoints the
 * image produces in *n_out even when that exceeds max_out (only the first max_out are stored).
 * Optional debug outputs (may be NULL): per-level stage-1 counts and cut scores. */
ORC_API int orc_orb_detect_compute(const uint8_t *gray, int w, int h, int stride, int nfeatures,
                                   int max_out, float *xy, float *size, float *angle,
                                   float *response, int32_t *octave, int32_t *xy_level,
                                   uint8_t *desc, int32_t *n_out, int32_t *dbg_stage1_count,
                                   int32_t *dbg_cut)
{
    if (!gray || w < 1 || h < 1 || !n_out) return -1;
    const int nl = RELOC_ORB_NLEVELS;
    int32_t lw[RELOC_ORB_NLEVELS], lh[RELOC_ORB_NLEVELS], quota[RELOC_ORB_NLEVELS];
    float lscale[RELOC_ORB_NLEVELS];
    orc_orb_layout(w, h, nfeatures, lw, lh, lscale, quota);

    uint8_t *prev = NULL;
    int prev_w = 0, prev_h = 0;
    int n = 0;
    for (int l = 0; l < nl; ++l) {
        const int cw = lw[l], ch = lh[l];
        if (cw < 1 || ch < 1) break;
        uint8_t *img = malloc((size_t)cw * ch);
        if (l == 0)
            for (int y = 0; y < ch; ++y) memcpy(img + (size_t)y * cw, gray + (size_t)y * stride, (size_t)cw);
        else
            orc_resize_linear_exact(prev, prev_w, prev_h, prev_w, img, cw, ch, cw);
        if (dbg_stage1_count) dbg_stage1_count[l] = 0;
        if (dbg_cut) dbg_cut[l] = 0;

        if (cw > 2 * RELOC_ORB_EDGE && ch > 2 * RELOC_ORB_EDGE && quota[l] > 0) {
            uint8_t *score = malloc((size_t)cw * ch), *kept = malloc((size_t)cw * ch);
            uint8_t *blur = malloc((size_t)cw * ch);
            orc_fast_score_map(img, cw, ch, cw, RELOC_FAST_THRESHOLD, score, cw);
            orc_fast_nms_map(score, cw, ch, cw, kept, cw);
            int32_t hist[256] = {0};
            for (size_t i = 0; i < (size_t)cw * ch; ++i) if (kept[i]) hist[kept[i]]++;
            int cut = orc_stage1_cut(hist, 2 * quota[l]);
            if (dbg_cut) dbg_cut[l] = cut;
            /* stage-1 set, raster order, with Harris response */
            int m = 0;
            for (size_t i = 0; i < (size_t)cw * ch; ++i) if (kept[i] && kept[i] >= cut) ++m;
            if (dbg_stage1_count) dbg_stage1_count[l] = m;
            int32_t *px = malloc(sizeof(int32_t) * (size_t)(m + 1)), *py = malloc(sizeof(int32_t) * (size_t)(m + 1));
            float *resp = malloc(sizeof(float) * (size_t)(m + 1));
            int k = 0;
            for (int y = 0; y < ch; ++y)
                for (int x = 0; x < cw; ++x) {
                    int s = kept[(size_t)y * cw + x];
                    if (s && s >= cut) {
                        px[k] = x; py[k] = y;
                        resp[k] = orc_harris_px(img + (size_t)y * cw + x, cw);
                        ++k;
                    }
                }
            orc_blur7(img, cw, ch, cw, blur, cw);
            /* stage 2: keep i iff fewer than quota responses are strictly greater */
            for (int i = 0; i < m; ++i) {
                int greater = 0;
                for (int j = 0; j < m; ++j) greater += resp[j] > resp[i];
                if (greater >= quota[l]) continue;
                if (n < max_out) {
                    const uint8_t *c = img + (size_t)py[i] * cw + px[i];
                    float ang = orc_ic_angle(c, cw);
                    volatile float fx = (float)px[i] * lscale[l], fy = (float)py[i] * lscale[l];
                    volatile float sz = (float)RELOC_ORB_PATCH * lscale[l];
                    if (xy) { xy[2 * n] = fx; xy[2 * n + 1] = fy; }
                    if (size) size[n] = sz;
                    if (angle) angle[n] = ang;
                    if (response) response[n] = resp[i];
                    if (octave) octave[n] = l;
                    if (xy_level) { xy_level[2 * n] = px[i]; xy_level[2 * n + 1] = py[i]; }
                    if (desc) orc_brief(blur + (size_t)py[i] * cw + px[i], cw, ang, desc + 32 * (size_t)n);
                }
                ++n;
            }
            free(px); free(py); free(resp);
            free(score); free(kept); free(blur);
        }
        free(prev);
        prev = img; prev_w = cw; prev_h = ch;
    }
    free(prev);
    *n_out = n;
    return 0;
}

/* Build the pyramid only (level l written at out + off[l], row stride = level width). */
ORC_API int orc_orb_pyramid(const uint8_t *gray, int w, int h, int stride, uint8_t *out,
                            const int64_t *off)
{
    int32_t lw[RELOC_ORB_NLEVELS], lh[RELOC_ORB_NLEVELS], quota[RELOC_ORB_NLEVELS];
    float lscale[RELOC_ORB_NLEVELS];
    orc_orb_layout(w, h, 500, lw, lh, lscale, quota);
    for (int y = 0; y < h; ++y) memcpy(out + off[0] + (size_t)y * w, gray + (size_t)y * stride, (size_t)w);
    for (int l = 1; l < RELOC_ORB_NLEVELS; ++l)
        orc_resize_linear_exact(out + off[l - 1], lw[l - 1], lh[l - 1], lw[l - 1], out + off[l], lw[l],
                                lh[l], lw[l]);
    return 0;
}
