// reloc_pnp.hip -- batched PnP-RANSAC on gfx950: hypothesis generation, reprojection scoring,
// adaptive-cap selection and Levenberg-Marquardt refinement, for up to MAX_CAND candidate records
// per launch.
//
// Serves cv2.solvePnPRansac(obj, img, K, DIST, iterationsCount=200, reprojectionError=3.0,
//                           flags=SOLVEPNP_ITERATIVE)                    (reference M:342-346, S:78-82)
// and the mean inlier reprojection error of M:353-356.
//
// Three kernels per batch:
//   k_pnp_hyp     four lanes per (candidate, hypothesis): counter-based sampler, P3P on three points, one lane per
//                 root of the quartic, the fourth point picks the root.  IEEE double add/sub/mul/div/sqrt only, no FMA, so the
//                 pose list is bit-identical to the specification.
//   k_pnp_score   one wave per (candidate, hypothesis): reprojection error of every correspondence,
//                 inlier count by ballot.  This is the data-parallel bulk (H x m projections).
//   k_pnp_finish  one wave per candidate: sequential adaptive-cap scan over the counts (lane 0),
//                 inlier list by ballot compaction, LM refinement with the 6x6 normal equations
//                 summed across the wave in fp64, Rodrigues log, mean inlier error.
#include "reloc_internal.h"

// Developer build (-DRELOC_PNP_TIMING, tools/exp_pnp_phases.py): the first wave of each PnP kernel stamps the 100 MHz
// wall clock at its phase boundaries.  Not compiled into the product library.
#ifdef RELOC_PNP_TIMING
__device__ unsigned long long g_pnp_phase[32];
#define PNP_T(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) g_pnp_phase[i] = wall_clock64(); } while (0)
#define PNP_V(i, v) do { if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) g_pnp_phase[i] = (unsigned long long)(v); } while (0)
RELOC_API int reloc_debug_pnp_phases(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pnp_phase), sizeof(g_pnp_phase)) == hipSuccess ? 0 : -1;
}
#else
#define PNP_T(i) do { } while (0)
#define PNP_V(i, v) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// deterministic helpers (mirror the arithmetic order of the specification exactly)
__device__ __forceinline__ uint64_t sm64_next(uint64_t &s)
{
    s += RELOC_RNG_GOLDEN;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * RELOC_RNG_MUL1;
    z = (z ^ (z >> 27)) * RELOC_RNG_MUL2;
    return z ^ (z >> 31);
}

__device__ bool pnp_sample(uint64_t seed, int h, int m, int idx[4])
{
    if (m < RELOC_PNP_SAMPLE) return false;
    uint64_t s = seed ^ ((uint64_t)(h + 1) * RELOC_RNG_MUL1);
    for (int k = 0; k < RELOC_PNP_SAMPLE; ++k) {
        for (;;) {
            const uint64_t z = sm64_next(s);
            const int c = (int)(((z >> 32) * (uint64_t)m) >> 32);
            bool dup = false;
            for (int j = 0; j < k; ++j) dup |= idx[j] == c;
            if (!dup) { idx[k] = c; break; }
        }
    }
    return true;
}

__device__ double log_spec(double x)
{
    int e;
    double m = frexp(x, &e);
    if (m < 0.70710678118654752440) { m = m * 2.0; e -= 1; }
    const double z = (m - 1.0) / (m + 1.0), z2 = z * z;
    double p = 1.0 / 23.0;
    for (int k = 21; k >= 1; k -= 2) { p = p * z2; p = p + 1.0 / (double)k; }
    p = p * z; p = p * 2.0;
    return (double)e * 0.69314718055994530942 + p;
}

// the adaptive iteration cap; log_num = ransac_log_num(conf) does not change inside a RANSAC run and is taken once
__device__ double ransac_log_num(double conf)
{
    const double DBL_MIN_ = 2.2250738585072014e-308;
    const double p = conf < 0 ? 0 : (conf > 1 ? 1 : conf);
    double num = 1.0 - p; if (num < DBL_MIN_) num = DBL_MIN_;
    return log_spec(num);
}
__device__ int ransac_update_iters(double log_num, double outlier_ratio, int max_iters)
{
    const double DBL_MIN_ = 2.2250738585072014e-308;
    const double ep = outlier_ratio < 0 ? 0 : (outlier_ratio > 1 ? 1 : outlier_ratio);
    const double w = 1.0 - ep, w2 = w * w, w4 = w2 * w2;
    double den = 1.0 - w4;
    if (den < DBL_MIN_) return 0;
    const double num = log_num;
    den = log_spec(den);
    if (den >= 0 || -num >= (double)max_iters * (-den)) return max_iters;
    return (int)rint(num / den);
}

// safeguarded Halley iteration from a Fujiwara-type bound, RELOC_P3P_CUBIC_ITERS steps at most (include/reloc_spec.h)
__device__ double cubic_pos_root(double c2, double c1, double c0)
{
    double lo = 0.0, hi = 1.0 + fabs(c2);
    if (fabs(c1) + 1.0 > hi) hi = fabs(c1) + 1.0;
    if (fabs(c0) + 1.0 > hi) hi = fabs(c0) + 1.0;
    /* start: every real root of t^3 + P t + Q (t = x + c2 / 3) lies below 2 max(|P|^(1/2), |Q / 2|^(1/3)) (Fujiwara);
     * the cube root is replaced by the next power of two above it (exponent arithmetic only: exact everywhere) */
    const double P = c1 - c2 * c2 / 3.0;
    const double Q = (2.0 * c2 * c2 * c2 - 9.0 * c2 * c1) / 27.0 + c0;
    double b = sqrt(fabs(P));
    if (Q != 0.0) {
        int e;
        (void)frexp(fabs(Q) * 0.5, &e);                      /* |Q| / 2 = m 2^e, 0.5 <= m < 1 */
        const int k = e >= 0 ? (e + 2) / 3 : -((-e) / 3);     /* ceil(e / 3) */
        const double cb = ldexp(1.0, k);
        if (cb > b) b = cb;
    }
    double x = 2.0 * b - c2 / 3.0;
    if (!(x > lo)) x = lo;
    if (!(x < hi)) x = hi;
    for (int it = 0; it < RELOC_P3P_CUBIC_ITERS; ++it) {
        const double g = ((x + c2) * x + c1) * x + c0;
        const double dg = (3.0 * x + 2.0 * c2) * x + c1;
        const double ddg = 6.0 * x + 2.0 * c2;
        if (g > 0) hi = x; else lo = x;
        const double den = 2.0 * dg * dg - g * ddg;
        double xn = x - 2.0 * g * dg / den;                   /* Halley */
        if (!(den != 0.0) || !(xn > lo) || !(xn < hi)) xn = 0.5 * (lo + hi);
        if (xn == x) break;
        x = xn;
    }
    return x;
}

// Ferrari: the (up to 4) real roots of A4 v^4 + ... + A0 before their Newton polish, in the order the specification
// lists them; returns the count.  shift = a / 4 (root of the quartic = y - shift).
__device__ int quartic_root_starts(const double A[5], double y[4], double &shift)
{
    double amax = 0;
    for (int i = 0; i < 5; ++i) if (fabs(A[i]) > amax) amax = fabs(A[i]);
    if (!(fabs(A[4]) > 1e-12 * amax) || !(amax > 0)) return 0;
    const double a = A[3] / A[4], b = A[2] / A[4], c = A[1] / A[4], d = A[0] / A[4];
    const double a2 = a * a;
    const double p = b - 0.375 * a2;
    const double q = c - 0.5 * a * b + 0.125 * a2 * a;
    const double r = d - 0.25 * a * c + 0.0625 * a2 * b - (3.0 / 256.0) * a2 * a2;
    shift = 0.25 * a;
    int n = 0;
    const double scale = fabs(p) + sqrt(fabs(r)) + 1e-300;
    if (fabs(q) <= 1e-14 * scale * sqrt(scale)) {
        const double disc = p * p - 4.0 * r;
        if (disc >= 0) {
            const double sd = sqrt(disc);
            const double z1 = 0.5 * (-p + sd), z2 = 0.5 * (-p - sd);
            if (z1 >= 0) { const double s = sqrt(z1); y[n++] = s; y[n++] = -s; }
            if (z2 >= 0) { const double s = sqrt(z2); y[n++] = s; y[n++] = -s; }
        }
    } else {
        const double m = cubic_pos_root(p, 0.25 * p * p - r, -0.125 * q * q);
        PNP_T(4);
        if (!(m > 0)) return 0;
        const double s = sqrt(2.0 * m);
        const double t = q / (2.0 * s);
        const double h = 0.5 * p + m;
        const double d1 = s * s - 4.0 * (h - t);
        const double d2 = s * s - 4.0 * (h + t);
        if (d1 >= 0) { const double sd = sqrt(d1); y[n++] = 0.5 * (-s + sd); y[n++] = 0.5 * (-s - sd); }
        if (d2 >= 0) { const double sd = sqrt(d2); y[n++] = 0.5 * (s + sd); y[n++] = 0.5 * (s - sd); }
    }
    return n;
}
// three Newton steps on the quartic itself
__device__ __forceinline__ double quartic_polish(const double A[5], double v)
{
    for (int it = 0; it < 3; ++it) {
        const double f = (((A[4] * v + A[3]) * v + A[2]) * v + A[1]) * v + A[0];
        const double df = ((4.0 * A[4] * v + 3.0 * A[3]) * v + 2.0 * A[2]) * v + A[1];
        if (df != 0.0) { const double vn = v - f / df; if (vn == vn) v = vn; }
    }
    return v;
}

__device__ __forceinline__ double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ void cross3(const double *a, const double *b, double *o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ bool unit3(double *a)
{
    const double n = sqrt(dot3(a, a));
    if (!(n > 1e-300)) return false;
    a[0] /= n; a[1] /= n; a[2] /= n;
    return true;
}
__device__ bool frame3(const double *p0, const double *p1, const double *p2, double F[9])
{
    double e1[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
    double w[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
    double e3[3], e2[3];
    if (!unit3(e1)) return false;
    cross3(e1, w, e3);
    if (!unit3(e3)) return false;
    cross3(e3, e1, e2);
    for (int i = 0; i < 3; ++i) { F[3 * i] = e1[i]; F[3 * i + 1] = e2[i]; F[3 * i + 2] = e3[i]; }
    return true;
}

// P3P in two halves, so that the <= 4 roots of a hypothesis can be worked on by 4 lanes:
//   p3p_setup    bearing vectors, quartic coefficients, the unpolished roots, the object frame        (per hypothesis)
//   p3p_solution polish of root i, depths, camera frame, R | t                                         (per root)
// The arithmetic of each half is the specification's, operation by operation.
struct P3pSetup {
    double f[3][3], N[3], D[2], A[5], cb, b2, Fp[9], y[4], shift;
    int nr;
};
__device__ bool p3p_setup(const double P[9], const double xn[6], P3pSetup &S)
{
    S.nr = 0;
    for (int i = 0; i < 3; ++i) {
        S.f[i][0] = xn[2 * i]; S.f[i][1] = xn[2 * i + 1]; S.f[i][2] = 1.0;
        if (!unit3(S.f[i])) return false;
    }
    double d12[3], d02[3], d01[3];
    for (int k = 0; k < 3; ++k) {
        d12[k] = P[3 + k] - P[6 + k];
        d02[k] = P[k] - P[6 + k];
        d01[k] = P[k] - P[3 + k];
    }
    const double a2 = dot3(d12, d12), b2 = dot3(d02, d02), c2 = dot3(d01, d01);
    if (!(a2 > 0) || !(b2 > 0) || !(c2 > 0)) return false;
    const double ca = dot3(S.f[1], S.f[2]), cb = dot3(S.f[0], S.f[2]), cg = dot3(S.f[0], S.f[1]);
    const double K = (a2 - c2) / b2, q = c2 / b2;
    const double N[3] = {1.0 + K, -2.0 * K * cb, K - 1.0};
    const double D[2] = {2.0 * cg, -2.0 * ca};
    const double NN[5] = {N[0] * N[0], 2.0 * N[0] * N[1], 2.0 * N[0] * N[2] + N[1] * N[1], 2.0 * N[1] * N[2], N[2] * N[2]};
    const double ND[4] = {N[0] * D[0], N[0] * D[1] + N[1] * D[0], N[1] * D[1] + N[2] * D[0], N[2] * D[1]};
    const double DD[3] = {D[0] * D[0], 2.0 * D[0] * D[1], D[1] * D[1]};
    const double W[3] = {1.0 - q, 2.0 * q * cb, -q};
    const double DW[5] = {DD[0] * W[0], DD[0] * W[1] + DD[1] * W[0], DD[0] * W[2] + DD[1] * W[1] + DD[2] * W[0],
                          DD[1] * W[2] + DD[2] * W[1], DD[2] * W[2]};
    for (int i = 0; i < 5; ++i) S.A[i] = NN[i] + DW[i];
    for (int i = 0; i < 4; ++i) S.A[i] -= 2.0 * cg * ND[i];
    for (int i = 0; i < 3; ++i) S.N[i] = N[i];
    S.D[0] = D[0]; S.D[1] = D[1]; S.cb = cb; S.b2 = b2;
    PNP_T(3);
    S.nr = quartic_root_starts(S.A, S.y, S.shift);
    PNP_T(5);
    if (!frame3(P, P + 3, P + 6, S.Fp)) { S.nr = 0; return false; }
    return true;
}
// root i -> R | t (12 doubles); false: the root gives no pose
__device__ bool p3p_solution(const P3pSetup &S, const double P[9], int i, double *Rt)
{
    if (i >= S.nr) return false;
    const double v = quartic_polish(S.A, S.y[i] - S.shift);
    if (!(v > 0)) return false;
    const double Dv = S.D[1] * v + S.D[0];
    if (!(fabs(Dv) > 1e-12)) return false;
    const double u = ((S.N[2] * v + S.N[1]) * v + S.N[0]) / Dv;
    if (!(u > 0)) return false;
    const double den = 1.0 + v * v - 2.0 * v * S.cb;
    if (!(den > 1e-300)) return false;
    const double s0 = sqrt(S.b2 / den), s1 = u * s0, s2 = v * s0;
    double C[3][3];
    for (int k = 0; k < 3; ++k) { C[0][k] = s0 * S.f[0][k]; C[1][k] = s1 * S.f[1][k]; C[2][k] = s2 * S.f[2][k]; }
    double Fc[9];
    if (!frame3(C[0], C[1], C[2], Fc)) return false;
    double *R = Rt, *t = R + 9;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            R[3 * r + c] = Fc[3 * r] * S.Fp[3 * c] + Fc[3 * r + 1] * S.Fp[3 * c + 1] + Fc[3 * r + 2] * S.Fp[3 * c + 2];
    for (int r = 0; r < 3; ++r) t[r] = C[0][r] - (R[3 * r] * P[0] + R[3 * r + 1] * P[1] + R[3 * r + 2] * P[2]);
    bool ok = true;
    for (int k = 0; k < 12; ++k) ok &= R[k] == R[k];
    return ok;
}

__device__ __forceinline__ double reproj_err2(const double *Rt, const double K4[4], const float *obj, const float *img)
{
    const double X = obj[0], Y = obj[1], Z = obj[2];
    const double x = ((Rt[0] * X + Rt[1] * Y) + Rt[2] * Z) + Rt[9];
    const double y = ((Rt[3] * X + Rt[4] * Y) + Rt[5] * Z) + Rt[10];
    const double z = ((Rt[6] * X + Rt[7] * Y) + Rt[8] * Z) + Rt[11];
    const double u = K4[0] * (x / z) + K4[2];
    const double v = K4[1] * (y / z) + K4[3];
    const double du = u - (double)img[0], dv = v - (double)img[1];
    return du * du + dv * dv;
}

struct PnpParams {
    double K4[4];
    double conf;
    double thr2;
    uint64_t seed;
    int iters;
    int stride;      // rows per candidate in obj/img/inlier arrays
    int min_m;       // fewer correspondences than this => no model (matcher gate M:330, or 4)
    // the tick's inlier gate (M:349 MIN_INLIERS / G:381 RELOC_MIN_INLIERS), known before the refinement: a candidate whose
    // consensus set is smaller cannot be accepted whatever its refined pose, so k_pnp_finish does not refine it.  0: no gate
    // (cv2.solvePnPRansac through the shim refines everything).
    int gate_local, gate_global;
};

// Frame-batched launches (reloc_tick_batch_dev, the sharded halves): the candidates of up to 8 contexts in one launch per
// kernel; the frame is the last grid dimension.
struct PnpFrame {
    const float *obj, *img; const int32_t *m_arr, *n_cand_p; double *Rt; int32_t *cnt, *inl; PnpOut *out; uint64_t seed;
    const int32_t *relocating;
};
struct PnpBatch { PnpFrame f[RELOC_BATCH_MAX]; };

// grid (ceil(iters/64), n_cand_max), block 256: FOUR lanes per hypothesis.  The sampler, the quartic and its resolvent are
// per hypothesis (the four lanes of a quad compute them in lockstep, same values); each of the <= 4 real roots is then
// polished, turned into a pose and scored against the fourth sample point by its own lane, and the quad keeps the pose with
// the smallest error (lowest root on ties: the specification's first-minimum rule).  Rounds 1-2 walked the roots one after
// the other in one lane: 4 us of an 18 us dependent chain.
constexpr int HYP_BLOCK = 256;
// the tick's inlier gate (0 outside a tick), see PnpParams
__device__ __forceinline__ int pnp_gate(const PnpParams &prm, const int32_t *relocating_p)
{
    return relocating_p ? (*relocating_p ? prm.gate_global : prm.gate_local) : 0;
}

__device__ __forceinline__ void pnp_hyp_body(const float *__restrict__ obj, const float *__restrict__ img,
                                             const int32_t *__restrict__ m_arr, const int32_t *__restrict__ n_cand_p,
                                             const PnpParams &prm, double *__restrict__ Rt_out, int32_t *__restrict__ cnt,
                                             const int32_t *__restrict__ relocating_p)
{
    const int c = blockIdx.y;
    if (n_cand_p && c >= *n_cand_p) return;
    const int h = blockIdx.x * (HYP_BLOCK / 4) + (threadIdx.x >> 2), sol = threadIdx.x & 3;
    if (h >= prm.iters) return;                                           // quad-uniform, like every exit below
    const int m = m_arr[c];
    const float *o = obj + (size_t)c * prm.stride * 3;
    const float *im = img + (size_t)c * prm.stride * 2;
    double *out = Rt_out + ((size_t)c * MAX_HYP + h) * 12;
    int32_t *cn = cnt + (size_t)c * MAX_HYP + h;
    int idx[4];
    PNP_T(0);
    // fewer correspondences than the inlier gate: no consensus set of this candidate can be accepted -- no hypotheses at all
    if (m < prm.min_m || m < pnp_gate(prm, relocating_p) || !pnp_sample(prm.seed, h, m, idx)) { if (sol == 0) *cn = -1; return; }
    PNP_T(1);
    double P[9], xn[6], Rt[12];
    for (int k = 0; k < 3; ++k) {
        for (int e = 0; e < 3; ++e) P[3 * k + e] = o[3 * idx[k] + e];
        xn[2 * k] = ((double)im[2 * idx[k]] - prm.K4[2]) / prm.K4[0];
        xn[2 * k + 1] = ((double)im[2 * idx[k] + 1] - prm.K4[3]) / prm.K4[1];
    }
    PNP_T(2);
    P3pSetup S;
    p3p_setup(P, xn, S);
    double e = 0;
    bool valid = p3p_solution(S, P, sol, Rt);
    if (valid) {
        e = reproj_err2(Rt, prm.K4, o + 3 * idx[3], im + 2 * idx[3]);
        valid = e == e;
    }
    PNP_T(6);
    // the quad's winner: smallest error, lowest root index on ties
    int win = valid ? sol : 4;
    double we = e;
#pragma unroll
    for (int d = 1; d <= 2; d <<= 1) {
        const int ow = __shfl_xor(win, d);
        const double oe = __shfl_xor(we, d);
        const bool take = ow < 4 && (win >= 4 || oe < we || (oe == we && ow < win));
        if (take) { win = ow; we = oe; }
    }
    if (win >= 4) { if (sol == 0) *cn = -1; return; }
    if (sol == win) {
        for (int k = 0; k < 12; ++k) out[k] = Rt[k];
        *cn = 0;
    }
    PNP_T(7);
}
__global__ __launch_bounds__(HYP_BLOCK) void k_pnp_hyp(const float *__restrict__ obj, const float *__restrict__ img,
                                                const int32_t *__restrict__ m_arr, const int32_t *__restrict__ n_cand_p,
                                                PnpParams prm, double *__restrict__ Rt_out, int32_t *__restrict__ cnt,
                                                const int32_t *__restrict__ relocating_p)
{
    RELOC_SMALL_KERNEL_PRIO();
    pnp_hyp_body(obj, img, m_arr, n_cand_p, prm, Rt_out, cnt, relocating_p);
}
// grid (ceil(iters/64), n_cand_max, frames)
__global__ __launch_bounds__(HYP_BLOCK) void k_pnp_hyp_batch(PnpBatch b, PnpParams prm)
{
    RELOC_SMALL_KERNEL_PRIO();
    const PnpFrame &F = b.f[blockIdx.z];
    prm.seed = F.seed;
    pnp_hyp_body(F.obj, F.img, F.m_arr, F.n_cand_p, prm, F.Rt, F.cnt, F.relocating);
}


// grid (iters, n_cand_max), block 64: one wave scores one hypothesis
__device__ __forceinline__ void pnp_score_body(const float *__restrict__ obj, const float *__restrict__ img,
                                               const int32_t *__restrict__ m_arr, const int32_t *__restrict__ n_cand_p,
                                               const PnpParams &prm, const double *__restrict__ Rt_in,
                                               int32_t *__restrict__ cnt, uint8_t *__restrict__ mask, int hyp_stride)
{
    const int c = blockIdx.y;
    if (n_cand_p && c >= *n_cand_p) return;
    const int h = blockIdx.x;
    int32_t *cn = cnt + (size_t)c * hyp_stride + h;
    if (*cn < 0) return;
    const int m = m_arr[c];
    const float *o = obj + (size_t)c * prm.stride * 3;
    const float *im = img + (size_t)c * prm.stride * 2;
    double Rt[12];
    for (int k = 0; k < 12; ++k) Rt[k] = Rt_in[((size_t)c * hyp_stride + h) * 12 + k];
    int count = 0;
    for (int i0 = 0; i0 < m; i0 += 64) {
        const int i = i0 + threadIdx.x;
        bool in = false;
        if (i < m) in = reproj_err2(Rt, prm.K4, o + 3 * i, im + 2 * i) <= prm.thr2;
        if (mask && i < m) mask[((size_t)c * hyp_stride + h) * m + i] = (uint8_t)in;
        count += __popcll(__ballot(in));
    }
    if (threadIdx.x == 0) *cn = count;
}
__global__ __launch_bounds__(64) void k_pnp_score(const float *__restrict__ obj, const float *__restrict__ img,
                                                  const int32_t *__restrict__ m_arr, const int32_t *__restrict__ n_cand_p,
                                                  PnpParams prm, const double *__restrict__ Rt_in,
                                                  int32_t *__restrict__ cnt, uint8_t *__restrict__ mask, int hyp_stride)
{
    RELOC_SMALL_KERNEL_PRIO();
    pnp_score_body(obj, img, m_arr, n_cand_p, prm, Rt_in, cnt, mask, hyp_stride);
}
// grid (iters, n_cand_max, frames)
__global__ __launch_bounds__(64) void k_pnp_score_batch(PnpBatch b, PnpParams prm, int hyp_stride)
{
    RELOC_SMALL_KERNEL_PRIO();
    const PnpFrame &F = b.f[blockIdx.z];
    pnp_score_body(F.obj, F.img, F.m_arr, F.n_cand_p, prm, F.Rt, F.cnt, nullptr, hyp_stride);
}


// ------------------------------------------------------------------------------------------------
__device__ void rodrigues_exp(const double w[3], double R[9])
{
    const double th2 = dot3(w, w), th = sqrt(th2);
    double A, B;
    if (th < 1e-6) { A = 1.0 - th2 / 6.0; B = 0.5 - th2 / 24.0; }
    else { A = sin(th) / th; B = (1.0 - cos(th)) / th2; }
    const double x = w[0], y = w[1], z = w[2];
    R[0] = 1.0 - B * (y * y + z * z); R[1] = -A * z + B * x * y;      R[2] = A * y + B * x * z;
    R[3] = A * z + B * x * y;        R[4] = 1.0 - B * (x * x + z * z); R[5] = -A * x + B * y * z;
    R[6] = -A * y + B * x * z;       R[7] = A * x + B * y * z;        R[8] = 1.0 - B * (x * x + y * y);
}

__device__ void rodrigues_log(const double R[9], double rvec[3])
{
    const double tr = R[0] + R[4] + R[8];
    double c = 0.5 * (tr - 1.0);
    if (c > 1) c = 1;
    if (c < -1) c = -1;
    const double ax[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
    const double s = 0.5 * sqrt(dot3(ax, ax));
    const double th = atan2(s, c);
    if (s < 1e-9) {
        if (c > 0) { rvec[0] = 0.5 * ax[0]; rvec[1] = 0.5 * ax[1]; rvec[2] = 0.5 * ax[2]; return; }
        double xx = sqrt(fmax((R[0] + 1.0) * 0.5, 0.0)), yy = sqrt(fmax((R[4] + 1.0) * 0.5, 0.0)), zz = sqrt(fmax((R[8] + 1.0) * 0.5, 0.0));
        if (xx >= yy && xx >= zz) { if (R[1] < 0) yy = -yy; if (R[2] < 0) zz = -zz; }
        else if (yy >= zz) { if (R[1] < 0) xx = -xx; if (R[5] < 0) zz = -zz; }
        else { if (R[2] < 0) xx = -xx; if (R[5] < 0) yy = -yy; }
        const double n = sqrt(xx * xx + yy * yy + zz * zz);
        rvec[0] = th * xx / n; rvec[1] = th * yy / n; rvec[2] = th * zz / n;
        return;
    }
    const double k = th / (2.0 * s);
    rvec[0] = k * ax[0]; rvec[1] = k * ax[1]; rvec[2] = k * ax[2];
}

// Wave-wide sum of doubles, every lane gets the total.  The summation tree is the xor butterfly
// v += v[lane ^ 32], ^16, ^8, ^4, ^2, ^1 (the order fixes the rounding, so it is kept), but the exchanges go
// through the VALU (v_permlane32_swap / v_permlane16_swap, DPP row_ror / row_shl / row_shr / quad_perm)
// instead of six dependent ds_bpermute round trips per value: the LM step reduces 28 values per cost
// evaluation (PnP stage 83 -> 74 us, outputs bit-identical).
__device__ __forceinline__ double wave_sum(double v)
{
    unsigned lo = (unsigned)__double_as_longlong(v), hi = (unsigned)((unsigned long long)__double_as_longlong(v) >> 32);
    auto mk = [](unsigned l, unsigned h) { return __longlong_as_double((long long)(((unsigned long long)h << 32) | l)); };
    {   // ^32: after the swap the two results are this lane's value and its partner's (in either order)
        const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = mk(a[0], b[0]) + mk(a[1], b[1]);
        lo = (unsigned)__double_as_longlong(v); hi = (unsigned)((unsigned long long)__double_as_longlong(v) >> 32);
    }
    {   // ^16
        const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = mk(a[0], b[0]) + mk(a[1], b[1]);
        lo = (unsigned)__double_as_longlong(v); hi = (unsigned)((unsigned long long)__double_as_longlong(v) >> 32);
    }
#define RELOC_DPP_MOV(x, ctrl, bank, old) (unsigned)__builtin_amdgcn_update_dpp((int)(old), (int)(x), ctrl, 0xF, bank, false)
    {   // ^8: rotate by 8 inside each row of 16
        v += mk(RELOC_DPP_MOV(lo, 0x128, 0xF, lo), RELOC_DPP_MOV(hi, 0x128, 0xF, hi));
        lo = (unsigned)__double_as_longlong(v); hi = (unsigned)((unsigned long long)__double_as_longlong(v) >> 32);
    }
    {   // ^4: banks 0 and 2 read lane + 4 (row_shl:4), banks 1 and 3 read lane - 4 (row_shr:4)
        const unsigned pl = RELOC_DPP_MOV(lo, 0x114, 0xA, RELOC_DPP_MOV(lo, 0x104, 0x5, lo));
        const unsigned ph = RELOC_DPP_MOV(hi, 0x114, 0xA, RELOC_DPP_MOV(hi, 0x104, 0x5, hi));
        v += mk(pl, ph);
        lo = (unsigned)__double_as_longlong(v); hi = (unsigned)((unsigned long long)__double_as_longlong(v) >> 32);
    }
    {   // ^2, ^1: quad permutes
        v += mk(RELOC_DPP_MOV(lo, 0x4E, 0xF, lo), RELOC_DPP_MOV(hi, 0x4E, 0xF, hi));
        lo = (unsigned)__double_as_longlong(v); hi = (unsigned)((unsigned long long)__double_as_longlong(v) >> 32);
        v += mk(RELOC_DPP_MOV(lo, 0xB1, 0xF, lo), RELOC_DPP_MOV(hi, 0xB1, 0xF, hi));
    }
#undef RELOC_DPP_MOV
    return v;
}

// 28 wave-wide sums with the summation tree of wave_sum (v[l] + v[l ^ 32], then ^ 16, ^ 8, ^ 4, ^ 2, ^ 1: identical
// rounding, IEEE addition is commutative) but as ONE transposing butterfly over all values: at every step half of the
// lanes keep one half of the registers and the other lanes the other half, so step s costs 32 / 2^s additions instead
// of 28.  Step ^32 / ^16 exchange through v_permlane32_swap / v_permlane16_swap (the swap hands each side exactly
// the halves it needs), the in-row steps through DPP.  Afterwards lane 2 k (and 2 k + 1) holds the total of value k;
// v_readlane broadcasts the 28 totals.  v[28..31] must be zero.  (k_pnp_finish is one wave per candidate, so the
// length of this dependent chain IS the kernel's run time: 43 -> see DESIGN.md.)
__device__ __forceinline__ void wave_sum28(double (&v)[32], double (&tot)[28])
{
    auto lo32 = [](double x) { return (unsigned)__double_as_longlong(x); };
    auto hi32 = [](double x) { return (unsigned)((unsigned long long)__double_as_longlong(x) >> 32); };
    auto mk = [](unsigned l, unsigned h) { return __longlong_as_double((long long)(((unsigned long long)h << 32) | l)); };
    const int lane = threadIdx.x & 63;
    double r[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {       // ^32: lanes 0-31 keep value k, lanes 32-63 value k + 16
        const auto a = __builtin_amdgcn_permlane32_swap(lo32(v[k]), lo32(v[k + 16]), false, false);
        const auto b = __builtin_amdgcn_permlane32_swap(hi32(v[k]), hi32(v[k + 16]), false, false);
        r[k] = mk(a[0], b[0]) + mk(a[1], b[1]);
    }
    double t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {        // ^16: even rows keep r[k], odd rows r[k + 8]
        const auto a = __builtin_amdgcn_permlane16_swap(lo32(r[k]), lo32(r[k + 8]), false, false);
        const auto b = __builtin_amdgcn_permlane16_swap(hi32(r[k]), hi32(r[k + 8]), false, false);
        t[k] = mk(a[0], b[0]) + mk(a[1], b[1]);
    }
    // in-row steps: lanes with the step's bit clear keep the first register of a pair, the others the second
    auto step = [&](double a, double b, int bit, auto xchg) {
        const bool hi = lane & bit;
        const double mine = hi ? b : a, theirs = hi ? a : b;
        return mine + mk(xchg(lo32(theirs)), xchg(hi32(theirs)));
    };
    double u[4], w[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) u[k] = step(t[k], t[k + 4], 8, [](unsigned x) { return dpp_xor<8>(x); });
#pragma unroll
    for (int k = 0; k < 2; ++k) w[k] = step(u[k], u[k + 2], 4, [](unsigned x) { return dpp_xor<4>(x); });
    double x = step(w[0], w[1], 2, [](unsigned y) { return dpp_xor<2>(y); });
    x = x + mk(dpp_xor<1>(lo32(x)), dpp_xor<1>(hi32(x)));
    const unsigned xl = lo32(x), xh = hi32(x);
#pragma unroll
    for (int k = 0; k < 28; ++k)
        tot[k] = mk((unsigned)__builtin_amdgcn_readlane((int)xl, 2 * k), (unsigned)__builtin_amdgcn_readlane((int)xh, 2 * k));
}

// Normal equations of the reprojection cost over the selected points, summed across the wave.
// H: upper triangle (21 values, row-major a<=b), g: 6, returns the cost; every lane gets the sums.
__device__ double lm_normal_wave(const float *obj, const float *img, const int32_t *sel, int n, const double *Rt,
                                 const double K4[4], double H[21], double g[6])
{
    double cost = 0;
    for (int k = 0; k < 21; ++k) H[k] = 0;
    for (int k = 0; k < 6; ++k) g[k] = 0;
    for (int kk = threadIdx.x; kk < n; kk += 64) {
        const int i = sel[kk];
        const double X = obj[3 * i], Y = obj[3 * i + 1], Z = obj[3 * i + 2];
        const double xr = Rt[0] * X + Rt[1] * Y + Rt[2] * Z;
        const double yr = Rt[3] * X + Rt[4] * Y + Rt[5] * Z;
        const double zr = Rt[6] * X + Rt[7] * Y + Rt[8] * Z;
        const double x = xr + Rt[9], y = yr + Rt[10], z = zr + Rt[11];
        const double iz = 1.0 / z;
        const double ru = K4[0] * x * iz + K4[2] - (double)img[2 * i];
        const double rv = K4[1] * y * iz + K4[3] - (double)img[2 * i + 1];
        const double ux = K4[0] * iz, uz = -K4[0] * x * iz * iz;
        const double vy = K4[1] * iz, vz = -K4[1] * y * iz * iz;
        double Ju[6], Jv[6];
        Ju[0] = uz * yr;            Ju[1] = ux * zr - uz * xr; Ju[2] = -ux * yr;
        Jv[0] = -vy * zr + vz * yr; Jv[1] = -vz * xr;          Jv[2] = vy * xr;
        Ju[3] = ux; Ju[4] = 0;  Ju[5] = uz;
        Jv[3] = 0;  Jv[4] = vy; Jv[5] = vz;
        int o = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            g[a] += Ju[a] * ru + Jv[a] * rv;
#pragma unroll
            for (int b = a; b < 6; ++b) H[o++] += Ju[a] * Ju[b] + Jv[a] * Jv[b];
        }
        cost += ru * ru + rv * rv;
    }
#if defined(RELOC_PNP_SUM28) && RELOC_PNP_SUM28 == 0          // developer switch: the r1 form, one butterfly per value
    for (int k = 0; k < 21; ++k) H[k] = wave_sum(H[k]);
    for (int k = 0; k < 6; ++k) g[k] = wave_sum(g[k]);
    return wave_sum(cost);
#endif
    // 28 wave-wide sums at once (see wave_sum28): ~130 VALU instructions instead of 28 x 30
    double v[32], tot[28];
    for (int k = 0; k < 21; ++k) v[k] = H[k];
    for (int k = 0; k < 6; ++k) v[21 + k] = g[k];
    v[27] = cost;
    v[28] = v[29] = v[30] = v[31] = 0.0;
    wave_sum28(v, tot);
    for (int k = 0; k < 21; ++k) H[k] = tot[k];
    for (int k = 0; k < 6; ++k) g[k] = tot[21 + k];
    return tot[27];
}

__device__ bool chol_solve6(const double Ain[36], const double b[6], double x[6])
{
    double L[36];
    for (int i = 0; i < 36; ++i) L[i] = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = Ain[6 * i + j];
            for (int k = 0; k < j; ++k) s -= L[6 * i + k] * L[6 * j + k];
            if (i == j) { if (!(s > 0)) return false; L[6 * i + i] = sqrt(s); }
            else L[6 * i + j] = s / L[6 * j + j];
        }
    double y[6];
    for (int i = 0; i < 6; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[6 * i + k] * y[k];
        y[i] = s / L[6 * i + i];
    }
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
        for (int k = i + 1; k < 6; ++k) s -= L[6 * k + i] * x[k];
        x[i] = s / L[6 * i + i];
    }
    return true;
}

// grid n_cand_max, block 64.  Register budget: left alone the kernel takes 237 VGPRs, and a wave of that size only starts on a
// SIMD from which TWO scan waves have retired; held to 128 (WAVES = 4 per SIMD, 144 bytes of scratch) it fits when one has:
// 4-stream run 5940 -> 6020 frames/s, synchronous tick +1.7 us.  80 registers: no further gain, tick +16 us.
// (The same limit on k_pnp_hyp (108 -> 80) and k_pyramid (81 -> 64) changes nothing.)  Ticks that run no whole-database scan
// and the single-call entry point use the unconstrained instantiation (ctx->latency_shapes).
__device__ __forceinline__ void pnp_finish_body(const float *__restrict__ obj, const float *__restrict__ img,
                                                const int32_t *__restrict__ m_arr, const int32_t *__restrict__ n_cand_p,
                                                const PnpParams &prm, const double *__restrict__ Rt_all,
                                                const int32_t *__restrict__ cnt, int32_t *__restrict__ inl_out,
                                                PnpOut *__restrict__ out, const int32_t *__restrict__ relocating_p)
{
    const int c = blockIdx.x;
    if (n_cand_p && c >= *n_cand_p) return;
    const int lane = threadIdx.x;
    const int m = m_arr[c];
    const float *o = obj + (size_t)c * prm.stride * 3;
    const float *im = img + (size_t)c * prm.stride * 2;
    int32_t *inl = inl_out + (size_t)c * prm.stride;
    PnpOut &po = out[c];
    // adaptive-cap RANSAC decision.  Sequential semantics ("first hypothesis that beats the best so far,
    // then shrink the cap"), evaluated by the wave: lane l holds the counts of hypotheses l, l+64, ...;
    // each step finds the earliest improving hypothesis below the current cap with ballots.
    int cl[MAX_HYP / 64];
    PNP_T(8);
    const int nj = (prm.iters + 63) >> 6;          // registers that hold hypotheses at all (200 iterations: 4 of 16): wave-uniform
#pragma unroll
    for (int j = 0; j < MAX_HYP / 64; ++j) {
        const int h = j * 64 + lane;
        cl[j] = -1;
        if (j < nj && h < prm.iters) cl[j] = cnt[(size_t)c * MAX_HYP + h];
    }
    int s_best_v = -1, s_best_count = 0;
    if (m >= prm.min_m) {
        int niters = prm.iters, best_count = RELOC_PNP_SAMPLE - 1, pos = 0;
        const double log_num = ransac_log_num(prm.conf);
        for (;;) {
            int found = -1;
#pragma unroll
            for (int j = 0; j < MAX_HYP / 64; ++j) {
                if (j < nj && found < 0) {                               // wave-uniform: the walk stops at the first hit
                    const int h = j * 64 + lane;
                    const unsigned long long bal = __ballot(h >= pos && h < niters && cl[j] > best_count);
                    if (bal) found = j * 64 + __ffsll((long long)bal) - 1;
                }
            }
            if (found < 0) break;
            // `found` is wave-uniform: read the winner's count with v_readlane instead of four ds_bpermute
            const int fl = __builtin_amdgcn_readfirstlane(found) & 63, fj = __builtin_amdgcn_readfirstlane(found) >> 6;
            int ch = 0;
#pragma unroll
            for (int j = 0; j < MAX_HYP / 64; ++j) {
                if (j == fj) ch = __builtin_amdgcn_readlane(cl[j], fl);    // wave-uniform branch
            }
            s_best_v = found;
            s_best_count = ch;
            best_count = ch;
            niters = ransac_update_iters(log_num, (double)(m - ch) / (double)m, niters);
            pos = found + 1;
        }
    }
    const int best = s_best_v;
    PNP_T(9);
    if (best < 0) {
        if (lane == 0) { po.ok = 0; po.n_inl = 0; po.best_h = -1; po.n_matches = m; po.reproj_mean = 0; }
        return;
    }
    double Rt[12];
    for (int k = 0; k < 12; ++k) Rt[k] = Rt_all[((size_t)c * MAX_HYP + best) * 12 + k];
    {
        // The tick's inlier gate is known here: a consensus set below it is rejected by the finalisation whatever the refined
        // pose (M:349, G:381) -- and most candidates of a whole-database search are wrong records with a handful of chance
        // inliers, on which Levenberg-Marquardt takes its longest (ill-conditioned, many rejected steps).  They keep the
        // RANSAC pose and skip the inlier list, the refinement and the error: the kernel lasts as long as its slowest wave,
        // which is now a candidate that can win (k_pnp_finish 24.5 -> see DESIGN.md us in the benchmark's tick).
        const int gate = pnp_gate(prm, relocating_p);
        if (s_best_count < gate) {
            if (lane == 0) {
                for (int k = 0; k < 12; ++k) po.Rt[k] = Rt[k];
                po.rvec[0] = po.rvec[1] = po.rvec[2] = 0.0;
                po.reproj_mean = 0.0;
                po.ok = 1;
                po.n_inl = s_best_count;
                po.best_h = best;
                po.n_matches = m;
            }
            return;
        }
    }
    // inlier list (ascending) by ballot compaction
    int n = 0;
    for (int i0 = 0; i0 < m; i0 += 64) {
        const int i = i0 + lane;
        bool in = false;
        if (i < m) in = reproj_err2(Rt, prm.K4, o + 3 * i, im + 2 * i) <= prm.thr2;
        const unsigned long long bal = __ballot(in);
        if (in) inl[n + __popcll(bal & ((1ull << lane) - 1ull))] = i;
        n += __popcll(bal);
    }
    __syncthreads();
    PNP_T(10);
    // Levenberg-Marquardt on the inliers, left-multiplied rotation increment
    double H[21], g[6], Hn[21], gn[6];
    double cost = lm_normal_wave(o, im, inl, n, Rt, prm.K4, H, g);
    double lambda = RELOC_LM_LAMBDA0;
    PNP_T(11);
    for (int trial = 0; trial < RELOC_LM_MAX_TRIALS; ++trial) {
        PNP_V(16, trial + 1);
        double A[36], rhs[6], dx[6];
        {
            int oo = 0;
            for (int a = 0; a < 6; ++a)
                for (int b = a; b < 6; ++b) { A[6 * a + b] = H[oo]; A[6 * b + a] = H[oo]; ++oo; }
        }
        for (int a = 0; a < 6; ++a) {
            const double d = A[6 * a + a];
            A[6 * a + a] += lambda * (d > 1e-300 ? d : 1e-300);
            rhs[a] = -g[a];
        }
        if (trial == 0) PNP_T(17);
        if (!chol_solve6(A, rhs, dx)) { lambda *= 10.0; continue; }
        if (trial == 0) PNP_T(18);
        double dR[9], Rn[12];
        rodrigues_exp(dx, dR);
        for (int r = 0; r < 3; ++r)
            for (int cc = 0; cc < 3; ++cc)
                Rn[3 * r + cc] = dR[3 * r] * Rt[cc] + dR[3 * r + 1] * Rt[3 + cc] + dR[3 * r + 2] * Rt[6 + cc];
        for (int k = 0; k < 3; ++k) Rn[9 + k] = Rt[9 + k] + dx[3 + k];
        if (trial == 0) PNP_T(19);
        const double cn = lm_normal_wave(o, im, inl, n, Rn, prm.K4, Hn, gn);
        if (trial == 0) PNP_T(20);
        double step = 0;
        for (int a = 0; a < 6; ++a) if (fabs(dx[a]) > step) step = fabs(dx[a]);
        const bool tiny = cn == cn && fabs(cn - cost) <= RELOC_LM_COST_EPS * (cost > 1e-300 ? cost : 1e-300);
        if (cn == cn && cn <= cost) {
            for (int k = 0; k < 12; ++k) Rt[k] = Rn[k];
            for (int k = 0; k < 21; ++k) H[k] = Hn[k];
            for (int k = 0; k < 6; ++k) g[k] = gn[k];
            cost = cn;
            lambda *= 0.1; if (lambda < 1e-12) lambda = 1e-12;
            if (step < RELOC_LM_STEP_EPS || tiny) break;
        } else {
            lambda *= 10.0;
            if (step < RELOC_LM_STEP_EPS || tiny) break;
        }
    }
    PNP_T(12);
    // mean inlier reprojection error under the refined pose (reference M:353-356)
    double esum = 0;
    for (int kk = lane; kk < n; kk += 64) {
        const int i = inl[kk];
        esum += sqrt(reproj_err2(Rt, prm.K4, o + 3 * i, im + 2 * i));
    }
    esum = wave_sum(esum);
    if (lane == 0) {
        for (int k = 0; k < 12; ++k) po.Rt[k] = Rt[k];
        rodrigues_log(Rt, po.rvec);
        po.reproj_mean = n > 0 ? esum / (double)n : 0.0;
        po.ok = 1;
        po.n_inl = n;
        po.best_h = best;
        po.n_matches = m;
    }
    PNP_T(13);
    PNP_V(14, n);
}
template <int WAVES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, 8))) void k_pnp_finish(const float *__restrict__ obj, const float *__restrict__ img,
                                                   const int32_t *__restrict__ m_arr, const int32_t *__restrict__ n_cand_p,
                                                   PnpParams prm, const double *__restrict__ Rt_all,
                                                   const int32_t *__restrict__ cnt, int32_t *__restrict__ inl_out,
                                                   PnpOut *__restrict__ out, const int32_t *__restrict__ relocating_p)
{
    RELOC_SMALL_KERNEL_PRIO();
    pnp_finish_body(obj, img, m_arr, n_cand_p, prm, Rt_all, cnt, inl_out, out, relocating_p);
}
// grid (n_cand_max, frames)
template <int WAVES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, 8))) void k_pnp_finish_batch(PnpBatch b, PnpParams prm)
{
    RELOC_SMALL_KERNEL_PRIO();
    const PnpFrame &F = b.f[blockIdx.y];
    prm.seed = F.seed;
    pnp_finish_body(F.obj, F.img, F.m_arr, F.n_cand_p, prm, F.Rt, F.cnt, F.inl, F.out, F.relocating);
}


static PnpParams make_params(const double K4[4], int iters, float thr_px, double conf, uint64_t seed, int stride,
                             int min_m)
{
    PnpParams p;
    for (int k = 0; k < 4; ++k) p.K4[k] = K4[k];
    p.conf = conf;
    p.thr2 = (double)thr_px * (double)thr_px;
    p.seed = seed;
    p.iters = iters;
    p.stride = stride;
    p.min_m = min_m < RELOC_PNP_SAMPLE ? RELOC_PNP_SAMPLE : min_m;
    p.gate_local = p.gate_global = 0;
    return p;
}

int pnp_run_candidates(reloc_ctx *ctx, int n_cand_max, const int32_t *n_cand_dev, const double K4[4], int iters,
                       float thr_px, double conf, uint64_t seed, int min_m, const int32_t *relocating_dev, int gate_local, int gate_global)
{
    if (n_cand_max <= 0) return RELOC_OK;
    if (n_cand_max > MAX_CAND || iters < 1 || iters > MAX_HYP) {
        reloc_set_error("pnp: n_cand %d (max %d) iters %d (max %d)", n_cand_max, MAX_CAND, iters, MAX_HYP);
        return RELOC_E_CAPACITY;
    }
    PnpParams prm = make_params(K4, iters, thr_px, conf, seed, MAX_REC_ROWS, min_m);
    prm.gate_local = gate_local; prm.gate_global = gate_global;
    reloc_prof_begin(ctx, RELOC_PROF_PNP);
    hipLaunchKernelGGL(k_pnp_hyp, dim3((iters + 63) / 64, n_cand_max), dim3(HYP_BLOCK), 0, ctx->stream, ctx->p_obj, ctx->p_img,
                       ctx->m_n, n_cand_dev, prm, ctx->p_Rt, ctx->p_cnt, relocating_dev);
    hipLaunchKernelGGL(k_pnp_score, dim3(iters, n_cand_max), dim3(64), 0, ctx->stream, ctx->p_obj, ctx->p_img, ctx->m_n,
                       n_cand_dev, prm, ctx->p_Rt, ctx->p_cnt, (uint8_t *)nullptr, MAX_HYP);
    if (ctx->latency_shapes)
        hipLaunchKernelGGL(k_pnp_finish<1>, dim3(n_cand_max), dim3(64), 0, ctx->stream, ctx->p_obj, ctx->p_img, ctx->m_n,
                           n_cand_dev, prm, ctx->p_Rt, ctx->p_cnt, ctx->p_inl, ctx->p_out, relocating_dev);
    else
        hipLaunchKernelGGL(k_pnp_finish<4>, dim3(n_cand_max), dim3(64), 0, ctx->stream, ctx->p_obj, ctx->p_img, ctx->m_n,
                           n_cand_dev, prm, ctx->p_Rt, ctx->p_cnt, ctx->p_inl, ctx->p_out, relocating_dev);
    reloc_prof_end(ctx, RELOC_PROF_PNP);
    HIP_TRY(hipGetLastError());
    return RELOC_OK;
}

// the candidates of n contexts (one shared stream, equal parameters) in three launches
int pnp_run_candidates_batch(reloc_ctx *const *ctxs, int n, int n_cand_max, const uint64_t *seeds)
{
    reloc_ctx *c0 = ctxs[0];
    const int iters = c0->prm.ransac_iterations;
    if (n < 1 || n > RELOC_BATCH_MAX || n_cand_max <= 0 || n_cand_max > MAX_CAND || iters < 1 || iters > MAX_HYP) {
        reloc_set_error("pnp batch: n %d n_cand %d (max %d) iters %d (max %d)", n, n_cand_max, MAX_CAND, iters, MAX_HYP);
        return RELOC_E_CAPACITY;
    }
    PnpParams prm = make_params(c0->K4, iters, (float)c0->prm.ransac_reproj_px, c0->prm.ransac_confidence, 0, MAX_REC_ROWS,
                                c0->prm.min_matches);
    prm.gate_local = c0->prm.min_inliers; prm.gate_global = c0->prm.global_min_inliers;
    PnpBatch b;
    for (int f = 0; f < RELOC_BATCH_MAX; ++f) {
        reloc_ctx *c = ctxs[f < n ? f : 0];
        PnpFrame &F = b.f[f];
        F.obj = c->p_obj; F.img = c->p_img; F.m_arr = c->m_n; F.n_cand_p = c->cand_n; F.Rt = c->p_Rt; F.cnt = c->p_cnt; F.inl = c->p_inl;
        F.out = c->p_out; F.seed = seeds ? seeds[f < n ? f : 0] : 0; F.relocating = c->tick_flags;
    }
    reloc_prof_begin(c0, RELOC_PROF_PNP);
    hipLaunchKernelGGL(k_pnp_hyp_batch, dim3((iters + 63) / 64, n_cand_max, n), dim3(HYP_BLOCK), 0, c0->stream, b, prm);
    hipLaunchKernelGGL(k_pnp_score_batch, dim3(iters, n_cand_max, n), dim3(64), 0, c0->stream, b, prm, MAX_HYP);
    hipLaunchKernelGGL(k_pnp_finish_batch<4>, dim3(n_cand_max, n), dim3(64), 0, c0->stream, b, prm);
    reloc_prof_end(c0, RELOC_PROF_PNP);
    HIP_TRY(hipGetLastError());
    return RELOC_OK;
}


// ------------------------------------------------------------------------------------------------
RELOC_API int reloc_pnp_score(reloc_ctx *ctx, const float *obj, const float *img, int m, const double *Rt, int H,
                              const double K4[4], float thr_px, int32_t *inlier_count, uint8_t *mask)
{
    ARG_CHECK_CTX(ctx, m >= 0 && H >= 0 && K4 && (H == 0 || (Rt && inlier_count)) && (m == 0 || (obj && img)),
              "reloc_pnp_score");
    if (H == 0) return RELOC_OK;
    if (m == 0) { for (int h = 0; h < H; ++h) inlier_count[h] = 0; return RELOC_OK; }
    if (H > 65535) { reloc_set_error("pnp_score: more than 65535 hypotheses"); return RELOC_E_CAPACITY; }
    void *dobj, *dimg, *drt, *dcnt, *dmask = nullptr, *dm;
    int rc;
    if ((rc = reloc_scratch(ctx, 0, (int64_t)m * 12, &dobj))) return rc;
    if ((rc = reloc_scratch(ctx, 1, (int64_t)m * 8, &dimg))) return rc;
    if ((rc = reloc_scratch(ctx, 2, (int64_t)H * 96, &drt))) return rc;
    if ((rc = reloc_scratch(ctx, 3, (int64_t)H * 4 + 16, &dcnt))) return rc;
    if (mask && (rc = reloc_scratch(ctx, 4, (int64_t)H * m, &dmask))) return rc;
    dm = (char *)dcnt + (int64_t)H * 4;
    HIP_TRY(hipMemcpyAsync(dobj, obj, (size_t)m * 12, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dimg, img, (size_t)m * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(drt, Rt, (size_t)H * 96, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemsetAsync(dcnt, 0, (size_t)H * 4, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dm, &m, 4, hipMemcpyHostToDevice, ctx->stream));
    const PnpParams prm = make_params(K4, H, thr_px, 0.99, 0, m, 0);
    hipLaunchKernelGGL(k_pnp_score, dim3(H, 1), dim3(64), 0, ctx->stream, (const float *)dobj, (const float *)dimg,
                       (const int32_t *)dm, (const int32_t *)nullptr, prm, (const double *)drt, (int32_t *)dcnt,
                       (uint8_t *)dmask, H);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(inlier_count, dcnt, (size_t)H * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (mask) HIP_TRY(hipMemcpyAsync(mask, dmask, (size_t)H * m, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RELOC_OK;
}

RELOC_API int reloc_pnp_ransac(reloc_ctx *ctx, const float *obj, const float *img, int m, const double K4[4], int iters,
                               float thr_px, double conf, uint64_t seed, double rvec[3], double tvec[3],
                               int32_t *inliers, int32_t *n_inl, int32_t *ok)
{
    ARG_CHECK_CTX(ctx, m >= 0 && K4 && rvec && tvec && n_inl && ok && (m == 0 || (obj && img && inliers)),
              "reloc_pnp_ransac");
    ARG_CHECK(iters >= 1 && iters <= MAX_HYP, "iterationsCount must be in [1, 256]");
    *ok = 0;
    *n_inl = 0;
    if (m < RELOC_PNP_SAMPLE) return RELOC_OK;
    if (m > MAX_REC_ROWS) { reloc_set_error("solvePnPRansac: more than %d correspondences", MAX_REC_ROWS); return RELOC_E_CAPACITY; }
    HIP_TRY(hipMemcpyAsync(ctx->p_obj, obj, (size_t)m * 12, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->p_img, img, (size_t)m * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->m_n, &m, 4, hipMemcpyHostToDevice, ctx->stream));
    // the single-call path has no MIN_MATCHES gate (that gate belongs to the matcher, M:330)
    int rc;
    ctx->latency_shapes = true;
    rc = pnp_run_candidates(ctx, 1, nullptr, K4, iters, thr_px, conf, seed, RELOC_PNP_SAMPLE, nullptr, 0, 0);
    ctx->latency_shapes = false;
    if (rc) return rc;
    PnpOut po;
    HIP_TRY(hipMemcpyAsync(&po, ctx->p_out, sizeof(po), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (po.ok) {
        HIP_TRY(hipMemcpyAsync(inliers, ctx->p_inl, (size_t)po.n_inl * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        for (int k = 0; k < 3; ++k) { rvec[k] = po.rvec[k]; tvec[k] = po.Rt[9 + k]; }
        *n_inl = po.n_inl;
        *ok = 1;
    }
    return RELOC_OK;
}
