// reloc_ctx.hip -- context lifetime, error reporting, stream/event plumbing of libreloc_hip.so.
#include <stdarg.h>
#include <stdlib.h>

#include "reloc_internal.h"

static thread_local char g_err[512] = "";

void reloc_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

RELOC_API const char *reloc_last_error(void) { return g_err; }

RELOC_API int reloc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

template <typename T>
static int dalloc(T **p, int64_t count)
{
    HIP_TRY(hipMalloc((void **)p, (size_t)(count > 0 ? count : 1) * sizeof(T)));
    return 0;
}

static int ctx_alloc(reloc_ctx *c)
{
    int rc = 0;
    const int64_t mf = c->max_feat;
    // pyramid geometry upper bound: sum over levels of stride*h with stride <= w+64 rounded
    int64_t pyr = 0;
    for (int l = 0; l < NLEV; ++l) {
        double s = pow(RELOC_ORB_SCALE_FACTOR, (double)l);
        int64_t w = (int64_t)(c->max_w / s) + 2, h = (int64_t)(c->max_h / s) + 2;
        pyr += ((w + 63) / 64 * 64) * h + 256;
    }
    c->pyr_bytes = pyr;
    rc |= dalloc(&c->pyr, pyr);
    rc |= dalloc(&c->blur, pyr);
    rc |= dalloc(&c->nms, pyr);
    rc |= dalloc(&c->rz_tab, (int64_t)NLEV * 2 * 2 * (c->max_w > c->max_h ? c->max_w : c->max_h));
    rc |= dalloc((char **)&c->pyr_tiles, (int64_t)((c->max_w + 15) / 16) * ((c->max_h + 15) / 16) * 128);   // >= any k_pyramid tiling
    rc |= dalloc(&c->hist, NLEV * 256);
    rc |= dalloc(&c->cand_cnt, NLEV);
    rc |= dalloc(&c->cand_key, (int64_t)NLEV * RELOC_ORB_STAGE1_CAP);
    rc |= dalloc(&c->cand_resp, (int64_t)NLEV * RELOC_ORB_STAGE1_CAP);
    rc |= dalloc(&c->kp_cnt, NLEV);
    rc |= dalloc(&c->kp_key, (int64_t)NLEV * RELOC_ORB_STAGE1_CAP);
    rc |= dalloc(&c->kp_resp, (int64_t)NLEV * RELOC_ORB_STAGE1_CAP);
    rc |= dalloc(&c->f_xy, mf * 2);
    rc |= dalloc(&c->f_size, mf);
    rc |= dalloc(&c->f_angle, mf);
    rc |= dalloc(&c->f_resp, mf);
    rc |= dalloc(&c->f_oct, mf);
    rc |= dalloc(&c->f_desc, mf * 32);
    rc |= dalloc(&c->f_count, 1);
    rc |= dalloc(&c->frame_img, (int64_t)c->max_w * c->max_h * 3);
    rc |= dalloc((char **)&c->orb_const, 1024);
    rc |= dalloc(&c->dbg_cut, NLEV);
    rc |= dalloc(&c->cand_ids, MAX_CAND);
    rc |= dalloc(&c->cand_n, 1);
    rc |= dalloc(&c->m_qidx, (int64_t)MAX_CAND * MAX_REC_ROWS);
    rc |= dalloc(&c->m_tidx, (int64_t)MAX_CAND * MAX_REC_ROWS);
    rc |= dalloc(&c->m_dist, (int64_t)MAX_CAND * MAX_REC_ROWS);
    rc |= dalloc(&c->m_n, MAX_CAND);
    rc |= dalloc(&c->p_obj, (int64_t)MAX_CAND * MAX_REC_ROWS * 3);
    rc |= dalloc(&c->p_img, (int64_t)MAX_CAND * MAX_REC_ROWS * 2);
    rc |= dalloc(&c->p_Rt, (int64_t)MAX_CAND * MAX_HYP * 12);
    rc |= dalloc(&c->p_cnt, (int64_t)MAX_CAND * MAX_HYP);
    rc |= dalloc(&c->p_inl, (int64_t)MAX_CAND * MAX_REC_ROWS);
    rc |= dalloc(&c->p_out, MAX_CAND);
    rc |= dalloc(&c->tick_res, 1);
    if (hipHostMalloc((void **)&c->tick_res_host, sizeof(TickResult), hipHostMallocDefault) != hipSuccess) {
        reloc_set_error("hipHostMalloc(result record) failed");
        c->tick_res_host = nullptr;
        rc |= RELOC_E_HIP;
    } else memset(c->tick_res_host, 0, sizeof(TickResult));
    rc |= dalloc(&c->accum_res, 1);
    rc |= dalloc(&c->tick_flags, 4);
    rc |= dalloc(&c->scan_ticket, (8 * 8 + 1) * 32);
    if (rc == 0 && hipMemset(c->scan_ticket, 0, (8 * 8 + 1) * 32 * 4) != hipSuccess) rc = RELOC_E_HIP;
    return rc;
}

int g_reloc_live_contexts = 0;

RELOC_API reloc_ctx *reloc_create(int device, int max_w, int max_h, int max_feat)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        reloc_set_error("no HIP device available (libreloc_hip has no CPU fallback)");
        return nullptr;
    }
    if (device < 0 || device >= n || max_w < 64 || max_h < 64 || max_feat < 64) {
        reloc_set_error("reloc_create: bad arguments (device %d of %d, %dx%d, max_feat %d)", device, n,
                        max_w, max_h, max_feat);
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) {
        reloc_set_error("hipSetDevice(%d) failed", device);
        return nullptr;
    }
    reloc_ctx *c = new reloc_ctx();
    __atomic_add_fetch(&g_reloc_live_contexts, 1, __ATOMIC_RELAXED);     // reloc_destroy (also the failure paths below) takes it back
    c->device = device;
    c->max_w = max_w;
    c->max_h = max_h;
    c->max_feat = max_feat;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->t0) != hipSuccess || hipEventCreate(&c->t1) != hipSuccess) {
        reloc_set_error("stream/event creation failed");
        __atomic_sub_fetch(&g_reloc_live_contexts, 1, __ATOMIC_RELAXED);
        delete c;
        return nullptr;
    }
    c->stream = c->own_stream;
    // reference constants (include/reloc_spec.h); reloc_set_params overrides them
    c->prm.nfeatures = 500;
    c->prm.max_candidates = RELOC_MAX_CANDIDATES;
    c->prm.min_matches = RELOC_MIN_MATCHES;
    c->prm.min_inliers = RELOC_MIN_INLIERS;
    c->prm.ransac_iterations = RELOC_RANSAC_ITERATIONS;
    c->prm.global_max_candidates = RELOC_GLOBAL_MAX_CANDIDATES;
    c->prm.global_min_inliers = RELOC_GLOBAL_MIN_INLIERS;
    c->prm.accum_min_kpts = RELOC_ACCUM_MIN_KPTS;
    c->prm.candidate_radius_m = RELOC_CANDIDATE_RADIUS_M;
    c->prm.heading_tol_deg = RELOC_HEADING_TOL_DEG;
    c->prm.reproj_max_px = RELOC_REPROJ_MAX_PX;
    c->prm.ransac_reproj_px = RELOC_RANSAC_REPROJ_PX;
    c->prm.ransac_confidence = RELOC_RANSAC_CONFIDENCE;
    c->prm.consistency_m = RELOC_CONSISTENCY_M;
    c->prm.global_reproj_max_px = RELOC_GLOBAL_REPROJ_MAX_PX;
    c->prm.accum_min_dist_m = RELOC_ACCUM_MIN_DIST_M;
    c->prm.accum_depth_min_m = RELOC_ACCUM_DEPTH_MIN_M;
    c->prm.accum_depth_max_m = RELOC_ACCUM_DEPTH_MAX_M;
    c->prm.gray_coeff_bits = RELOC_GRAY_DEFAULT_BITS;     // OpenCV 4.x set (reloc_spec.h)
    c->prm.reserved0 = 0;
    // Developer switches (tools/exp_*, one test): read ONLY when RELOC_DEV=1 is set, once, here.  A deployment's environment
    // cannot change the library's behaviour by accident; none of them changes a result.
    if (const char *dev = getenv("RELOC_DEV"); dev && dev[0] == '1' && dev[1] == 0) {
        if (const char *e = getenv("RELOC_SCAN_GRID")) c->scan_grid = atoi(e);
        if (const char *e = getenv("RELOC_SCAN_GENS")) c->scan_gens = atoi(e);
        if (const char *e = getenv("RELOC_SCAN_NW")) c->scan_nw = atoi(e);
        if (const char *e = getenv("RELOC_SCAN_BATCH_GENS")) c->scan_batch_gens = atoi(e);
        if (const char *e = getenv("RELOC_SCAN_QUOTA_ROWS")) c->scan_quota_rows = atoi(e);
        if (const char *e = getenv("RELOC_LOCAL_TWO_STAGE")) c->local_two_stage = atoi(e) != 0;
    }
    if (ctx_alloc(c) != 0) {
        reloc_destroy(c);
        return nullptr;
    }
    return c;
}

RELOC_API void reloc_destroy(reloc_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    // database arrays: one reference each (owner or adopter alike); counts / topk_part are this context's own scratch
    db_arrays_drop(c->db_share, c->db_desc, c->db_pts3d, c->db_kp2d, c->db_off, c->db_pose, c->db_xy_heading);
    void *ptrs[] = {c->pyr, c->blur, c->nms, c->rz_tab, c->pyr_tiles, c->hist, c->cand_cnt, c->cand_key, c->cand_resp,
                    c->kp_cnt, c->kp_key, c->kp_resp, c->f_xy, c->f_size, c->f_angle, c->f_resp, c->f_oct,
                    c->f_desc, c->f_count, c->frame_img, c->orb_const, c->dbg_cut, c->db_counts, c->topk_part, c->cand_ids, c->cand_n, c->m_qidx,
                    c->m_tidx, c->m_dist, c->m_n, c->p_obj, c->p_img, c->p_Rt, c->p_cnt, c->p_inl,
                    c->p_out, c->tick_res, c->accum_res, c->tick_flags, c->scan_ticket};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (c->tick_res_host) (void)hipHostFree(c->tick_res_host);
    {   // the database that is not selected
        DbArena &a = c->db_slot[1 - c->db_sel];
        db_arrays_drop(a.share, a.desc, a.pts3d, a.kp2d, a.off, a.pose, a.xy_heading);
        if (a.counts) (void)hipFree(a.counts);
        if (a.topk_part) (void)hipFree(a.topk_part);
    }
    for (int i = 0; i < 8; ++i)
        if (c->scratch[i]) (void)hipFree(c->scratch[i]);
    for (int k = 0; k < RELOC_PROF_N; ++k)
        if (c->prof[k].init)
            for (int i = 0; i < RELOC_PROF_RING; ++i) {
                (void)hipEventDestroy(c->prof[k].a[i]);
                (void)hipEventDestroy(c->prof[k].b[i]);
            }
    if (c->t0) (void)hipEventDestroy(c->t0);
    if (c->t1) (void)hipEventDestroy(c->t1);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    __atomic_sub_fetch(&g_reloc_live_contexts, 1, __ATOMIC_RELAXED);
    delete c;
}

RELOC_API int reloc_get_params(reloc_ctx *c, reloc_params *out)
{
    ARG_CHECK_CTX(c, out, "reloc_get_params");
    *out = c->prm;
    return RELOC_OK;
}

RELOC_API int reloc_set_params(reloc_ctx *c, const reloc_params *p)
{
    ARG_CHECK_CTX(c, p, "reloc_set_params");
    ARG_CHECK(p->nfeatures >= 1 && p->nfeatures <= c->max_feat, "nfeatures must be in [1, max_feat]");
    ARG_CHECK(p->max_candidates >= 1 && p->max_candidates * 3 <= 32, "max_candidates must be in [1, 10]");
    ARG_CHECK(p->global_max_candidates >= 1 && p->global_max_candidates <= MAX_CAND, "global_max_candidates must be in [1, 32]");
    ARG_CHECK(p->ransac_iterations >= 1 && p->ransac_iterations <= MAX_HYP, "ransac_iterations must be in [1, 1024]");
    ARG_CHECK(p->min_matches >= RELOC_PNP_SAMPLE && p->min_inliers >= 0 && p->global_min_inliers >= 0 && p->accum_min_kpts >= 1,
              "min_matches must be >= 4, inlier / keypoint gates non-negative");
    ARG_CHECK(p->candidate_radius_m >= 0 && p->heading_tol_deg >= 0 && p->heading_tol_deg <= 180 && p->ransac_reproj_px > 0 &&
                  p->ransac_confidence > 0 && p->ransac_confidence < 1 && p->consistency_m >= 0 && p->accum_min_dist_m >= 0,
              "a gate is out of range");
    ARG_CHECK((p->gray_coeff_bits == RELOC_GRAY_SHIFT || p->gray_coeff_bits == RELOC_GRAY15_SHIFT) && p->reserved0 == 0,
              "gray_coeff_bits must be 14 or 15, reserved0 must be 0");
    c->prm = *p;
    return RELOC_OK;
}

RELOC_API int reloc_set_stream(reloc_ctx *c, void *hip_stream)
{
    ARG_CHECK_CTX(c, true, "ctx is NULL");
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return RELOC_OK;
}

RELOC_API int reloc_sync(reloc_ctx *c)
{
    ARG_CHECK_CTX(c, true, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RELOC_OK;
}

RELOC_API void *reloc_dev_alloc(reloc_ctx *c, int64_t bytes)
{
    if (!c || bytes < 0) { reloc_set_error("reloc_dev_alloc: bad arguments"); return nullptr; }
    void *p = nullptr;
    (void)hipSetDevice(c->device);
    if (hipMalloc(&p, (size_t)(bytes > 0 ? bytes : 1)) != hipSuccess) {
        reloc_set_error("hipMalloc(%lld) failed", (long long)bytes);
        return nullptr;
    }
    return p;
}

RELOC_API int reloc_dev_free(reloc_ctx *c, void *p)
{
    ARG_CHECK_CTX(c, true, "ctx is NULL");
    if (p) HIP_TRY(hipFree(p));
    return RELOC_OK;
}

RELOC_API int reloc_h2d(reloc_ctx *c, void *dst, const void *src, int64_t bytes)
{
    ARG_CHECK_CTX(c, dst && src && bytes >= 0, "reloc_h2d");
    HIP_TRY(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
    return RELOC_OK;
}

RELOC_API int reloc_d2d(reloc_ctx *c, void *dst, const void *src, int64_t bytes)
{
    ARG_CHECK_CTX(c, dst && src && bytes >= 0, "reloc_d2d");
    HIP_TRY(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, c->stream));
    return RELOC_OK;
}

RELOC_API int reloc_set_exclusive(reloc_ctx *c, int on)
{
    ARG_CHECK_CTX(c, true, "ctx is NULL");
    c->exclusive_hint = on < 0 ? -1 : (on != 0);
    return RELOC_OK;
}

RELOC_API void *reloc_host_alloc(int64_t bytes)
{
    void *p = nullptr;
    if (bytes < 0 || hipHostMalloc(&p, (size_t)(bytes > 0 ? bytes : 1), hipHostMallocDefault) != hipSuccess) {
        reloc_set_error("hipHostMalloc(%lld) failed", (long long)bytes);
        return nullptr;
    }
    return p;
}

RELOC_API int reloc_host_free(void *p)
{
    if (p) HIP_TRY(hipHostFree(p));
    return RELOC_OK;
}

RELOC_API void *reloc_get_stream(reloc_ctx *c) { return c ? (void *)c->stream : nullptr; }

RELOC_API int reloc_d2h(reloc_ctx *c, void *dst, const void *src, int64_t bytes)
{
    ARG_CHECK_CTX(c, dst && src && bytes >= 0, "reloc_d2h");
    HIP_TRY(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
    return RELOC_OK;
}

RELOC_API int reloc_timer_begin(reloc_ctx *c)
{
    ARG_CHECK_CTX(c, true, "ctx is NULL");
    HIP_TRY(hipEventRecord(c->t0, c->stream));
    return RELOC_OK;
}

RELOC_API int reloc_timer_end(reloc_ctx *c, float *ms)
{
    ARG_CHECK_CTX(c, ms, "reloc_timer_end");
    HIP_TRY(hipEventRecord(c->t1, c->stream));
    HIP_TRY(hipEventSynchronize(c->t1));
    HIP_TRY(hipEventElapsedTime(ms, c->t0, c->t1));
    return RELOC_OK;
}

int reloc_scratch(reloc_ctx *c, int slot, int64_t bytes, void **out)
{
    if (c->scratch_bytes[slot] < bytes) {
        if (c->scratch[slot]) {
            HIP_TRY(hipStreamSynchronize(c->stream));
            HIP_TRY(hipFree(c->scratch[slot]));
            c->scratch[slot] = nullptr;
            c->scratch_bytes[slot] = 0;
        }
        int64_t cap = bytes + bytes / 4 + 4096;
        HIP_TRY(hipMalloc(&c->scratch[slot], (size_t)cap));
        c->scratch_bytes[slot] = cap;
    }
    *out = c->scratch[slot];
    return RELOC_OK;
}

// ---- per-kernel HIP-event stopwatch ----------------------------------------------------------
static void prof_flush(reloc_ctx *c, int which)
{
    auto &p = c->prof[which];
    for (int i = 0; i < p.n; ++i) {
        float ms = 0.f;
        if (hipEventSynchronize(p.b[i]) == hipSuccess && hipEventElapsedTime(&ms, p.a[i], p.b[i]) == hipSuccess) {
            p.total += ms;
            p.launches += 1;
        }
    }
    p.n = 0;
}

void reloc_prof_begin(reloc_ctx *c, int which)
{
    if (!c->prof_on) return;
    auto &p = c->prof[which];
    if (!p.init) {
        for (int i = 0; i < RELOC_PROF_RING; ++i) { (void)hipEventCreate(&p.a[i]); (void)hipEventCreate(&p.b[i]); }
        p.init = true;
    }
    if (p.n == RELOC_PROF_RING) prof_flush(c, which);
    (void)hipEventRecord(p.a[p.n], c->stream);
}

void reloc_prof_end(reloc_ctx *c, int which)
{
    if (!c->prof_on) return;
    auto &p = c->prof[which];
    (void)hipEventRecord(p.b[p.n], c->stream);
    p.n += 1;
}

RELOC_API int reloc_profile_enable(reloc_ctx *c, int on)
{
    ARG_CHECK_CTX(c, true, "ctx is NULL");
    for (int k = 0; k < RELOC_PROF_N; ++k) {
        prof_flush(c, k);
        c->prof[k].total = 0.f;
        c->prof[k].launches = 0;
    }
    c->prof_on = on;
    return RELOC_OK;
}

RELOC_API int reloc_profile_get(reloc_ctx *c, int which, float *total_ms, int32_t *launches)
{
    ARG_CHECK_CTX(c, which >= 0 && which < RELOC_PROF_N && total_ms && launches, "reloc_profile_get");
    prof_flush(c, which);
    *total_ms = c->prof[which].total;
    *launches = c->prof[which].launches;
    return RELOC_OK;
}
