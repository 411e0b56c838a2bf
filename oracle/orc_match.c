/* orc_match.c -- CPU ORACLE (test infrastructure only) for 256-bit Hamming matching.
 *
 * THIS FILE IS A CHECKER, NOT A PRODUCT PATH (see orc_orb.c header for who may use it).
 *
 * Restates cv2.BFMatcher(cv2.NORM_HAMMING, crossCheck=True).match(desc_t, desc_curr)
 *   (simulation/isaac/scripts/common/visual_landmark_matcher.py:211, :327;
 *    simulation/isaac/experiments/63_global_reloc/scripts/visual_landmark_matcher.py:337)
 * and cv2.BFMatcher(cv2.NORM_HAMMING, crossCheck=False).knnMatch(q, t, k=2)
 *   (simulation/isaac/experiments/55_visual_teach_repeat/scripts/checkpoint_a_selftest.py:46,68).
 * OpenCV's batchDistance/normHamming (modules/core, absent from the reference tree) per
 * distance = popcount(a xor b) over 32 bytes as int32; nearest = smallest
 * distance, LOWEST index on ties; crossCheck keeps (i, j) only when i is also the
 * lowest-index nearest query of train j; result sorted by queryIdx.
 *
 * PARITY: the Hamming arithmetic is pinned by tests against numpy.unpackbits (a closed-form
 * known answer).  The tie rule is this repository's stated rule (A.7 hazard 4): OpenCV cannot
 * be run here, the reference holds no match fixtures, so tie behaviour vs OpenCV is UNPINNED.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/reloc_spec.h"

#define ORC_API __attribute__((visibility("default")))

static inline int ham256(const uint8_t *a, const uint8_t *b)
{
    uint64_t x[4], y[4];
    memcpy(x, a, 32);
    memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) +
           __builtin_popcountll(x[2] ^ y[2]) + __builtin_popcountll(x[3] ^ y[3]);
}

ORC_API int orc_hamming_matrix(const uint8_t *a, int64_t na, const uint8_t *b, int64_t nb, uint16_t *out)
{
    for (int64_t i = 0; i < na; ++i)
        for (int64_t j = 0; j < nb; ++j) out[i * nb + j] = (uint16_t)ham256(a + 32 * i, b + 32 * j);
    return 0;
}

/* forward nearest of every q row in t (lowest index on ties) */
static void nearest(const uint8_t *q, int nq, const uint8_t *t, int nt, int32_t *idx, int32_t *dist)
{
    for (int i = 0; i < nq; ++i) {
        int bd = 1 << 30, bi = -1;
        for (int j = 0; j < nt; ++j) {
            int d = ham256(q + 32 * (size_t)i, t + 32 * (size_t)j);
            if (d < bd) { bd = d; bi = j; }
        }
        idx[i] = bi;
        dist[i] = bd;
    }
}

/* crossCheck=True .match(q, t): mutual nearest neighbours, sorted by queryIdx. */
ORC_API int orc_match_mutual(const uint8_t *q, int nq, const uint8_t *t, int nt, int32_t *qidx,
                             int32_t *tidx, int32_t *dist, int32_t *n_out)
{
    *n_out = 0;
    if (nq <= 0 || nt <= 0) return 0;
    int32_t *fi = malloc(sizeof(int32_t) * (size_t)nq), *fd = malloc(sizeof(int32_t) * (size_t)nq);
    int32_t *ri = malloc(sizeof(int32_t) * (size_t)nt), *rd = malloc(sizeof(int32_t) * (size_t)nt);
    nearest(q, nq, t, nt, fi, fd);
    nearest(t, nt, q, nq, ri, rd);
    int n = 0;
    for (int i = 0; i < nq; ++i)
        if (ri[fi[i]] == i) {
            qidx[n] = i;
            tidx[n] = fi[i];
            dist[n] = fd[i];
            ++n;
        }
    free(fi); free(fd); free(ri); free(rd);
    *n_out = n;
    return 0;
}

/* knnMatch(q, t, k=2): two smallest distances per query, stable by index.  Missing second
 * neighbour (nt == 1) is reported as idx -1, dist -1. */
ORC_API int orc_match_knn2(const uint8_t *q, int nq, const uint8_t *t, int nt, int32_t *idx, int32_t *dist)
{
    for (int i = 0; i < nq; ++i) {
        int d0 = 1 << 30, d1 = 1 << 30, i0 = -1, i1 = -1;
        for (int j = 0; j < nt; ++j) {
            int d = ham256(q + 32 * (size_t)i, t + 32 * (size_t)j);
            if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = j; }
            else if (d < d1) { d1 = d; i1 = j; }
        }
        idx[2 * i] = i0; dist[2 * i] = i0 < 0 ? -1 : d0;
        idx[2 * i + 1] = i1; dist[2 * i + 1] = i1 < 0 ? -1 : d1;
    }
    return 0;
}

/* Variant-G whole-database scan (experiments/63_global_reloc/scripts/visual_landmark_matcher.py
 * :329-344): number of mutual matches of every record against the current frame's descriptors.
 * Record r owns teach descriptors [offsets[r], offsets[r+1]).  query = teach record rows,
 * train = current-frame rows, exactly as matcher.match(desc_t, desc_curr). */
/* Records are independent, so the scan is also the one place the oracle uses more than one core
 * (bench.py's cpu_baseline states the thread count; orc_set_threads(1) gives the scalar port). */
static int g_threads = 1;
ORC_API void orc_set_threads(int n) { g_threads = n > 0 ? n : 1; }

/* Same count with every distance evaluated ONCE (row and column minima tracked together, lowest index on ties both
 * ways): what a CPU implementation would do; used only for bench.py's cpu_baseline timing and held equal to the
 * literal two-pass restatement above by tests/test_oracle_known_answers.py. */
static int g_single_pass = 0;
ORC_API void orc_set_single_pass(int on) { g_single_pass = on != 0; }

static int mutual_count_single_pass(const uint8_t *q, int nq, const uint8_t *t, int nt, int32_t *colkey /* nt */)
{
    if (nq <= 0 || nt <= 0) return 0;
    /* key = distance << 16 | index of the other side: min(key) = smallest distance, lowest index on ties */
    for (int j = 0; j < nt; ++j) colkey[j] = 0x7fffffff;
    int n = 0;
    int32_t *rowkey = malloc(sizeof(int32_t) * (size_t)nq);
    for (int i = 0; i < nq; ++i) {
        int32_t best = 0x7fffffff;
        for (int j = 0; j < nt; ++j) {
            const int d = ham256(q + 32 * (size_t)i, t + 32 * (size_t)j);
            const int32_t kr = (d << 16) | j, kc = (d << 16) | i;
            if (kr < best) best = kr;
            if (kc < colkey[j]) colkey[j] = kc;
        }
        rowkey[i] = best;
    }
    for (int i = 0; i < nq; ++i) n += (colkey[rowkey[i] & 0xffff] & 0xffff) == i;
    free(rowkey);
    return n;
}

ORC_API int orc_db_match_counts(const uint8_t *db, const int64_t *offsets, int64_t n_rec,
                                const uint8_t *cur, int n_cur, int32_t *counts)
{
    int cap = 1;
    for (int64_t r = 0; r < n_rec; ++r) {
        int64_t n = offsets[r + 1] - offsets[r];
        if (n > cap) cap = (int)n;
    }
#pragma omp parallel num_threads(g_threads)
    {
        int32_t *qi = malloc(sizeof(int32_t) * (size_t)cap), *ti = malloc(sizeof(int32_t) * (size_t)cap);
        int32_t *dd = malloc(sizeof(int32_t) * (size_t)cap);
        int32_t *ck = malloc(sizeof(int32_t) * (size_t)(n_cur > 0 ? n_cur : 1));
#pragma omp for schedule(dynamic, 16)
        for (int64_t r = 0; r < n_rec; ++r) {
            int32_t n = 0;
            if (g_single_pass && n_cur <= 0xffff && offsets[r + 1] - offsets[r] <= 0xffff)
                n = mutual_count_single_pass(db + 32 * offsets[r], (int)(offsets[r + 1] - offsets[r]), cur, n_cur, ck);
            else
                orc_match_mutual(db + 32 * offsets[r], (int)(offsets[r + 1] - offsets[r]), cur, n_cur, qi, ti, dd, &n);
            counts[r] = n;
        }
        free(qi); free(ti); free(dd); free(ck);
    }
    return 0;
}

/* Top-k records by (count desc, record id desc) among records with count >= min_count:
 * `scored.sort(reverse=True)` on (len(good), li) tuples at .../63_global_reloc/...:342-343. */
ORC_API int orc_topk_records(const int32_t *counts, int64_t n_rec, int min_count, int k,
                             int32_t *ids, int32_t *n_out)
{
    int n = 0;
    for (int sel = 0; sel < k; ++sel) {
        int64_t best = -1;
        for (int64_t r = 0; r < n_rec; ++r) {
            if (counts[r] < min_count) continue;
            int taken = 0;
            for (int j = 0; j < n; ++j) taken |= ids[j] == r;
            if (taken) continue;
            if (best < 0 || counts[r] > counts[best] || (counts[r] == counts[best] && r > best)) best = r;
        }
        if (best < 0) break;
        ids[n++] = (int32_t)best;
    }
    *n_out = n;
    return 0;
}

/* Ratio-test score of every record: the archived anchor localizers and the
 * self-test count, per teach record, the current descriptors whose two nearest rows of the record
 * satisfy d1 < ratio * d2 -- `knnMatch(desc_cur, descriptors[aid], k=2)`, `len(pair) == 2 and
 * pair[0].distance < ratio * pair[1].distance`
 * (simulation/isaac/scripts/_archive/anchor_localizer.py:82-90,
 *  experiments/55_visual_teach_repeat/scripts/checkpoint_a_selftest.py:68-71).
 * Distances are ints cast to float by OpenCV; the product is evaluated in double like Python does. */
ORC_API int orc_db_ratio_counts(const uint8_t *db, const int64_t *offsets, int64_t n_rec, const uint8_t *cur,
                                int n_cur, double ratio, int32_t *counts)
{
    for (int64_t r = 0; r < n_rec; ++r) {
        const uint8_t *t = db + 32 * offsets[r];
        const int nt = (int)(offsets[r + 1] - offsets[r]);
        int good = 0;
        if (nt >= 2)
            for (int i = 0; i < n_cur; ++i) {
                int d0 = 1 << 30, d1 = 1 << 30;
                for (int j = 0; j < nt; ++j) {
                    int d = ham256(cur + 32 * (size_t)i, t + 32 * (size_t)j);
                    if (d < d0) { d1 = d0; d0 = d; }
                    else if (d < d1) d1 = d;
                }
                good += (double)d0 < ratio * (double)d1;
            }
        counts[r] = good;
    }
    return 0;
}
