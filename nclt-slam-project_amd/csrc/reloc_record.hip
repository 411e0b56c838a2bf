// reloc_record.hip -- teach-side record builder on gfx950.
//
// Serves VisualLandmarkRecorder._tick after ORB (reference R:247-288): keypoint rounding, border and
// ground masks, depth lookup (mm -> m), 3x3 non-zero depth standard deviation, range / variance gates,
// pin-hole back-projection.  The reference does this in NumPy with a Python loop per keypoint
// (R:262-266); NumPy's arithmetic is the specification here, so every expression below mirrors the
// NumPy dtype rules of the reference's lines:
//   np.round(float32)            -> round half to even in float32
//   depth.astype(float32)/1000.0 -> float32 division
//   patch[patch > 0.01].std()    -> float32 pairwise sum (numpy's 8-accumulator tree for n >= 8,
//                                   running sum from 0 for n < 8), mean, squared deviations, mean, sqrt
//   (uu - CX) * d_c / FX         -> int32 - python float = float64, * float32 = float64, / float64,
//                                   stacked with float32 z and cast to float32
// One lane per keypoint; survivors are compacted in keypoint order by a block-wide scan.
#include "reloc_internal.h"

struct RecordParams {
    double fx, fy, cx, cy;
    float depth_min, depth_max, var_max;
    int ground_y;
    int w, h;
};

// float32 sum with numpy's pairwise order for n <= 9
__device__ float np_sum9(const float *a, int n)
{
    if (n < 8) {
        float r = 0.f;
        for (int i = 0; i < n; ++i) r = __fadd_rn(r, a[i]);
        return r;
    }
    float r = __fadd_rn(__fadd_rn(__fadd_rn(a[0], a[1]), __fadd_rn(a[2], a[3])),
                        __fadd_rn(__fadd_rn(a[4], a[5]), __fadd_rn(a[6], a[7])));
    for (int i = 8; i < n; ++i) r = __fadd_rn(r, a[i]);
    return r;
}

__global__ __launch_bounds__(1024) void k_record(const float *__restrict__ f_xy, const uint8_t *__restrict__ f_desc,
                                                 const int32_t *__restrict__ f_count, int max_feat,
                                                 const uint16_t *__restrict__ depth, int dstride, RecordParams p,
                                                 float *__restrict__ o_xy, uint8_t *__restrict__ o_desc,
                                                 float *__restrict__ o_pts, int32_t *__restrict__ o_idx,
                                                 int32_t *__restrict__ o_n)
{
    RELOC_SMALL_KERNEL_PRIO();
    __shared__ int s_wsum[16];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = min(*f_count, max_feat);
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + tid;
        bool keep = false;
        float x = 0, y = 0, dz = 0;
        int u = 0, v = 0;
        if (i < n) {
            x = f_xy[2 * i]; y = f_xy[2 * i + 1];
            u = (int)rintf(x); v = (int)rintf(y);
            if (u >= 1 && u < p.w - 1 && v >= 1 && v < p.h - 1 && v > p.ground_y) {
                dz = __fdiv_rn((float)depth[(size_t)v * dstride + u], 1000.0f);
                float vals[9];
                int cnt = 0;
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) {
                        const float m = __fdiv_rn((float)depth[(size_t)(v + dy) * dstride + (u + dx)], 1000.0f);
                        if (m > 0.01f) vals[cnt++] = m;
                    }
                float sd = 999.0f;
                if (cnt >= 3) {
                    const float mean = __fdiv_rn(np_sum9(vals, cnt), (float)cnt);
                    float sq[9];
                    for (int k = 0; k < cnt; ++k) { const float d = __fsub_rn(vals[k], mean); sq[k] = __fmul_rn(d, d); }
                    sd = __fsqrt_rn(__fdiv_rn(np_sum9(sq, cnt), (float)cnt));
                }
                keep = dz > p.depth_min && dz < p.depth_max && sd < p.var_max;
            }
        }
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) s_wsum[wave] = __popcll(bal);
        __syncthreads();
        int before = s_base;
        for (int w = 0; w < wave; ++w) before += s_wsum[w];
        int total = 0;
        for (int w = 0; w < 16; ++w) total += s_wsum[w];
        if (keep) {
            const int pos = before + __popcll(bal & ((1ull << lane) - 1ull));
            o_xy[2 * pos] = x; o_xy[2 * pos + 1] = y;
            o_pts[3 * pos] = (float)__ddiv_rn(__dmul_rn(__dsub_rn((double)u, p.cx), (double)dz), p.fx);
            o_pts[3 * pos + 1] = (float)__ddiv_rn(__dmul_rn(__dsub_rn((double)v, p.cy), (double)dz), p.fy);
            o_pts[3 * pos + 2] = dz;
            o_idx[pos] = i;
            const uint4 *s = reinterpret_cast<const uint4 *>(f_desc + (size_t)i * 32);
            uint4 *d = reinterpret_cast<uint4 *>(o_desc + (size_t)pos * 32);
            d[0] = s[0]; d[1] = s[1];
        }
        __syncthreads();
        if (tid == 0) s_base += total;
        __syncthreads();
    }
    if (tid == 0) *o_n = s_base;
}

// Host-pointer entry point: ORB on the frame, then the gates.  Outputs hold up to the ctx's max_feat
// rows: xy (n,2) f32 keypoints_2d, desc (n,32) u8, pts3d (n,3) f32 keypoints_3d_cam, kp_index (n) i32
// (row of the surviving keypoint in the frame's ORB output); *n_out rows written, *n_kp = ORB keypoints.
RELOC_API int reloc_record_frame(reloc_ctx *ctx, const uint8_t *img, const uint16_t *depth_mm, int w, int h, int order,
                                 int nfeatures, float *xy, uint8_t *desc, float *pts3d, int32_t *kp_index, int32_t *n_out,
                                 int32_t *n_kp)
{
    ARG_CHECK_CTX(ctx, img && depth_mm && n_out && w >= 64 && h >= 64 && nfeatures > 0, "reloc_record_frame");
    if (w > ctx->max_w || h > ctx->max_h) { reloc_set_error("frame exceeds ctx capacity"); return RELOC_E_CAPACITY; }
    *n_out = 0;
    if (n_kp) *n_kp = 0;
    int rc;
    void *ddepth, *dout;
    const int64_t mf = ctx->max_feat;
    if ((rc = reloc_scratch(ctx, 5, (int64_t)w * h * 2, &ddepth))) return rc;
    if ((rc = reloc_scratch(ctx, 6, mf * (8 + 32 + 12 + 4) + 64, &dout))) return rc;
    uint8_t *o_desc = (uint8_t *)dout;
    float *o_xy = (float *)(o_desc + mf * 32), *o_pts = o_xy + mf * 2;
    int32_t *o_idx = (int32_t *)(o_pts + mf * 3), *o_n = o_idx + mf;
    HIP_TRY(hipMemcpyAsync(ctx->frame_img, img, (size_t)w * h * 3, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ddepth, depth_mm, (size_t)w * h * 2, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = orb_run_dev(ctx, ctx->frame_img, w, h, w * 3, 3, order, nfeatures))) return rc;
    RecordParams p;
    p.fx = ctx->K4[0]; p.fy = ctx->K4[1]; p.cx = ctx->K4[2]; p.cy = ctx->K4[3];
    p.depth_min = RELOC_DEPTH_MIN_M; p.depth_max = RELOC_DEPTH_MAX_M; p.var_max = RELOC_DEPTH_VAR_MAX_M;
    p.ground_y = RELOC_GROUND_Y_THRESHOLD; p.w = w; p.h = h;
    hipLaunchKernelGGL(k_record, dim3(1), dim3(1024), 0, ctx->stream, ctx->f_xy, ctx->f_desc, ctx->f_count, ctx->max_feat,
                       (const uint16_t *)ddepth, w, p, o_xy, o_desc, o_pts, o_idx, o_n);
    HIP_TRY(hipGetLastError());
    int32_t n = 0, nk = 0;
    HIP_TRY(hipMemcpyAsync(&n, o_n, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(&nk, ctx->f_count, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (n > 0) {
        if (xy) HIP_TRY(hipMemcpyAsync(xy, o_xy, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (desc) HIP_TRY(hipMemcpyAsync(desc, o_desc, (size_t)n * 32, hipMemcpyDeviceToHost, ctx->stream));
        if (pts3d) HIP_TRY(hipMemcpyAsync(pts3d, o_pts, (size_t)n * 12, hipMemcpyDeviceToHost, ctx->stream));
        if (kp_index) HIP_TRY(hipMemcpyAsync(kp_index, o_idx, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    *n_out = n;
    if (n_kp) *n_kp = nk;
    return RELOC_OK;
}

// ---------------------------------------------------------------------------------------------------
// Accumulation (reference M:435-500): after a tick that published nothing, the current frame becomes a new record when
// the matcher has been silent long enough (host-side fact, silence_ok), no filed record lies within min_dist of the
// robot and enough keypoints carry a usable depth.  The record is written straight behind the arena's last row; the
// host adopts it (record / row counts) when it reads AccumResult.  NumPy dtype rules of the reference lines, as above:
//   np.round(pts[:, 0]).astype(int32); (uu >= 1) & (uu < W - 1) & (vv >= 1) & (vv < H - 1)            M:449-453
//   d_c = depth[vv, uu].astype(float32) / 1000.0; ok = (d_c > 0.5) & (d_c < 15.0)                     M:460-461
//   (uu - cx) * d_c / fx in float64, stacked with z and cast to float32                                M:469-473
//   camera pose = base pose (+) static mount, quaternion by scipy's Rotation.from_matrix().as_quat()   M:475-480
struct AccumParams {
    double fx, fy, cx, cy;
    double base_pose[7], b2c_t[3], b2c_R[9];
    double min_dist;
    float zmin, zmax;
    int w, h, min_kpts, silence_ok;
    int64_t L, T;            // records / rows of the database before the append
};

// scipy.spatial.transform.Rotation.from_matrix (Markley's method) followed by as_quat(): x y z w, not canonicalised
__device__ void rot_to_quat_scipy(const double R[9], double q[4])
{
    const double d0 = R[0], d1 = R[4], d2 = R[8], tr = R[0] + R[4] + R[8];
    int choice = 0;
    double best = d0;
    if (d1 > best) { best = d1; choice = 1; }
    if (d2 > best) { best = d2; choice = 2; }
    if (tr > best) { best = tr; choice = 3; }
    if (choice != 3) {
        const int i = choice, j = (i + 1) % 3, k = (j + 1) % 3;
        q[i] = 1 - tr + 2 * R[3 * i + i];
        q[j] = R[3 * j + i] + R[3 * i + j];
        q[k] = R[3 * k + i] + R[3 * i + k];
        q[3] = R[3 * k + j] - R[3 * j + k];
    } else {
        q[0] = R[7] - R[5];
        q[1] = R[2] - R[6];
        q[2] = R[3] - R[1];
        q[3] = 1 + tr;
    }
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int c = 0; c < 4; ++c) q[c] = q[c] / n;
}

__global__ __launch_bounds__(1024) void k_accumulate(const float *__restrict__ f_xy, const uint8_t *__restrict__ f_desc,
                                                     const int32_t *__restrict__ f_count, int max_feat,
                                                     const uint16_t *__restrict__ depth, int dstride, AccumParams p,
                                                     const TickResult *__restrict__ tick, double *__restrict__ xyh,
                                                     uint8_t *__restrict__ db_desc, float *__restrict__ db_pts3d,
                                                     float *__restrict__ db_kp2d, int64_t *__restrict__ db_off,
                                                     double *__restrict__ db_pose, AccumResult *__restrict__ res)
{
    RELOC_SMALL_KERNEL_PRIO();
    __shared__ int s_wsum[16];
    __shared__ double s_wmin[16];
    __shared__ int s_base;
    __shared__ double s_near;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int oc = tick->outcome;
    const bool wanted = p.silence_ok && (oc == RELOC_OUT_NO_CANDIDATES || oc == RELOC_OUT_NO_PNP_ACCEPT || oc == RELOC_OUT_CONSISTENCY_FAIL);
    if (!wanted) {                                               // block-uniform
        if (tid == 0) { res->appended = 0; res->n_kpts = 0; res->nearest_m = -1.0; }
        return;
    }
    // nearest filed record (M:444-446)
    double dmin = 1e300;
    for (int64_t i = tid; i < p.L; i += 1024) {
        const double dx = xyh[4 * i] - p.base_pose[0], dy = xyh[4 * i + 1] - p.base_pose[1];
        dmin = fmin(dmin, sqrt(dx * dx + dy * dy));
    }
    for (int d = 32; d >= 1; d >>= 1) dmin = fmin(dmin, __shfl_xor(dmin, d));
    if (lane == 0) s_wmin[wave] = dmin;
    if (tid == 0) s_base = 0;
    __syncthreads();
    if (tid == 0) {
        double m = s_wmin[0];
        for (int w = 1; w < 16; ++w) m = fmin(m, s_wmin[w]);
        s_near = m;
    }
    __syncthreads();
    const double nearest = s_near;
    if (nearest < p.min_dist) {
        if (tid == 0) { res->appended = 0; res->n_kpts = 0; res->nearest_m = nearest; }
        return;
    }
    const int n = min(*f_count, max_feat);
    uint8_t *o_desc = db_desc + p.T * 32;
    float *o_pts = db_pts3d + p.T * 3, *o_xy = db_kp2d + p.T * 2;
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + tid;
        bool keep = false;
        float x = 0, y = 0, dz = 0;
        int u = 0, v = 0;
        if (i < n) {
            x = f_xy[2 * i]; y = f_xy[2 * i + 1];
            u = (int)rintf(x); v = (int)rintf(y);
            if (u >= 1 && u < p.w - 1 && v >= 1 && v < p.h - 1) {
                dz = __fdiv_rn((float)depth[(size_t)v * dstride + u], 1000.0f);
                keep = dz > p.zmin && dz < p.zmax;
            }
        }
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) s_wsum[wave] = __popcll(bal);
        __syncthreads();
        int before = s_base;
        for (int w = 0; w < wave; ++w) before += s_wsum[w];
        int total = 0;
        for (int w = 0; w < 16; ++w) total += s_wsum[w];
        if (keep) {
            const int pos = before + __popcll(bal & ((1ull << lane) - 1ull));
            o_xy[2 * pos] = x; o_xy[2 * pos + 1] = y;
            o_pts[3 * pos] = (float)__ddiv_rn(__dmul_rn(__dsub_rn((double)u, p.cx), (double)dz), p.fx);
            o_pts[3 * pos + 1] = (float)__ddiv_rn(__dmul_rn(__dsub_rn((double)v, p.cy), (double)dz), p.fy);
            o_pts[3 * pos + 2] = dz;
            const uint4 *sp = reinterpret_cast<const uint4 *>(f_desc + (size_t)i * 32);
            uint4 *dp = reinterpret_cast<uint4 *>(o_desc + (size_t)pos * 32);
            dp[0] = sp[0]; dp[1] = sp[1];
        }
        __syncthreads();
        if (tid == 0) s_base += total;
        __syncthreads();
    }
    if (tid == 0) {
        const int cnt = s_base;
        res->n_kpts = cnt;
        res->nearest_m = nearest;
        if (cnt < p.min_kpts) { res->appended = 0; return; }
        // camera pose from the base pose through the static mount (M:475-480)
        double Rwb[9], Rwc[9], q[4];
        quat_to_rot(p.base_pose[3], p.base_pose[4], p.base_pose[5], p.base_pose[6], Rwb);
        double *pose = db_pose + 7 * p.L;
        for (int r = 0; r < 3; ++r)
            pose[r] = p.base_pose[r] + ((Rwb[3 * r] * p.b2c_t[0] + Rwb[3 * r + 1] * p.b2c_t[1]) + Rwb[3 * r + 2] * p.b2c_t[2]);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c)
                Rwc[3 * r + c] = (Rwb[3 * r] * p.b2c_R[c] + Rwb[3 * r + 1] * p.b2c_R[3 + c]) + Rwb[3 * r + 2] * p.b2c_R[6 + c];
        rot_to_quat_scipy(Rwc, q);
        for (int c = 0; c < 4; ++c) pose[3 + c] = q[c];
        // filed under the VIO position (M:491); heading as for every record (M:233-245, k_db_index)
        double Rq[9];
        quat_to_rot(q[0], q[1], q[2], q[3], Rq);
        const double fx = Rq[0] * p.b2c_R[0] + Rq[1] * p.b2c_R[1] + Rq[2] * p.b2c_R[2];
        const double fy = Rq[3] * p.b2c_R[0] + Rq[4] * p.b2c_R[1] + Rq[5] * p.b2c_R[2];
        const double fn = sqrt(fx * fx + fy * fy);
        xyh[4 * p.L] = p.base_pose[0];
        xyh[4 * p.L + 1] = p.base_pose[1];
        xyh[4 * p.L + 2] = fn > 0 ? fx / fn : 1.0;
        xyh[4 * p.L + 3] = fn > 0 ? fy / fn : 0.0;
        db_off[p.L + 1] = p.T + cnt;
        res->appended = 1;
    }
}

RELOC_API int reloc_tick_accumulate_dev(reloc_ctx *ctx, const uint16_t *depth_mm_dev, int w, int h, const double base_pose[7],
                                        int silence_ok)
{
    ARG_CHECK_CTX(ctx, depth_mm_dev && base_pose && w >= 64 && h >= 64, "reloc_tick_accumulate_dev");
    if (!db_ready(ctx)) { reloc_set_error("no database uploaded"); return RELOC_E_STATE; }
    if (ctx->db_shared) { reloc_set_error("the selected database is shared from another context (read-only here)"); return RELOC_E_STATE; }
    if (ctx->db_records + 1 > ctx->db_cap_records || ctx->db_rows + ctx->max_feat > ctx->db_cap_rows) {
        // grow first (drains the stream): the kernel writes behind the last row without asking
        int rc = db_reserve(ctx, ctx->db_cap_records + ctx->db_cap_records / 2 + 64,
                            ctx->db_cap_rows + ctx->db_cap_rows / 2 + 64 * (int64_t)ctx->max_feat);
        if (rc) return rc;
    }
    AccumParams p;
    p.fx = ctx->K4[0]; p.fy = ctx->K4[1]; p.cx = ctx->K4[2]; p.cy = ctx->K4[3];
    for (int k = 0; k < 7; ++k) p.base_pose[k] = base_pose[k];
    for (int k = 0; k < 3; ++k) p.b2c_t[k] = ctx->b2c_t[k];
    for (int k = 0; k < 9; ++k) p.b2c_R[k] = ctx->b2c_R[k];
    p.min_dist = ctx->prm.accum_min_dist_m;
    p.zmin = (float)ctx->prm.accum_depth_min_m; p.zmax = (float)ctx->prm.accum_depth_max_m;
    p.w = w; p.h = h; p.min_kpts = ctx->prm.accum_min_kpts; p.silence_ok = silence_ok;
    p.L = ctx->db_records; p.T = ctx->db_rows;
    hipLaunchKernelGGL(k_accumulate, dim3(1), dim3(1024), 0, ctx->stream, ctx->f_xy, ctx->f_desc, ctx->f_count, ctx->max_feat,
                       depth_mm_dev, w, p, ctx->tick_res, ctx->db_xy_heading, ctx->db_desc, ctx->db_pts3d, ctx->db_kp2d, ctx->db_off,
                       ctx->db_pose, ctx->accum_res);
    HIP_TRY(hipGetLastError());
    return RELOC_OK;
}

RELOC_API int reloc_accumulate_result(reloc_ctx *ctx, int32_t *appended, int32_t *n_kpts, double *nearest_m)
{
    ARG_CHECK_CTX(ctx, true, "ctx is NULL");
    AccumResult r;
    HIP_TRY(hipMemcpyAsync(&r, ctx->accum_res, sizeof(r), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (r.appended == 1) {
        // adopt the record the kernel wrote behind the last row; exactly once per accumulate call
        ctx->db_records += 1;
        ctx->db_rows += r.n_kpts;
        if (r.n_kpts > ctx->db_max_rows) ctx->db_max_rows = r.n_kpts;
        const int32_t zero = 0;
        HIP_TRY(hipMemcpyAsync(&ctx->accum_res->appended, &zero, 4, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    if (appended) *appended = r.appended;
    if (n_kpts) *n_kpts = r.n_kpts;
    if (nearest_m) *nearest_m = r.nearest_m;
    return RELOC_OK;
}

// ---------------------------------------------------------------------------------------------------
// Depth image -> obstacle point cloud (reference relay depth_cb,
// tf_wall_clock_relay_v55.py:1020-1038): every `step`-th pixel, keep 0.3 < z < 10 and finite,
// point = (z, -(u - cx) / fx * z, -(v - cy) / fy * z) in float32, raster order.
__global__ __launch_bounds__(1024) void k_depth_points(const void *__restrict__ depth, int is_f32, int w, int h, int dstride_px,
                                                       int step, float cx, float cy, float fx, float fy, float zmin, float zmax,
                                                       float *__restrict__ out, int32_t *__restrict__ o_n)
{
    __shared__ int s_wsum[16];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gw = (w + step - 1) / step, gh = (h + step - 1) / step;
    const int n = gw * gh;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + tid;
        bool keep = false;
        float z = 0;
        int u = 0, v = 0;
        if (i < n) {
            v = (i / gw) * step; u = (i % gw) * step;
            if (is_f32) z = reinterpret_cast<const float *>(depth)[(size_t)v * dstride_px + u];
            else z = __fdiv_rn((float)reinterpret_cast<const uint16_t *>(depth)[(size_t)v * dstride_px + u], 1000.0f);
            keep = z > zmin && z < zmax && isfinite(z);
        }
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) s_wsum[wave] = __popcll(bal);
        __syncthreads();
        int before = s_base;
        for (int k = 0; k < wave; ++k) before += s_wsum[k];
        int total = 0;
        for (int k = 0; k < 16; ++k) total += s_wsum[k];
        if (keep) {
            const int pos = before + __popcll(bal & ((1ull << lane) - 1ull));
            const float px = __fmul_rn(__fdiv_rn(__fsub_rn((float)u, cx), fx), z);
            const float py = __fmul_rn(__fdiv_rn(__fsub_rn((float)v, cy), fy), z);
            out[3 * pos] = z; out[3 * pos + 1] = -px; out[3 * pos + 2] = -py;
        }
        __syncthreads();
        if (tid == 0) s_base += total;
        __syncthreads();
    }
    if (tid == 0) *o_n = s_base;
}

// depth: (h, w) float32 metres (is_f32 = 1) or uint16 millimetres (is_f32 = 0), dense rows.
// points must hold ceil(w/step) * ceil(h/step) rows of 3 floats.
RELOC_API int reloc_depth_points(reloc_ctx *ctx, const void *depth, int is_f32, int w, int h, int step, const double K4[4],
                                 float zmin, float zmax, float *points, int32_t *n_out)
{
    ARG_CHECK_CTX(ctx, depth && points && n_out && w > 0 && h > 0 && step > 0 && K4, "reloc_depth_points");
    if ((int64_t)w * h > (int64_t)ctx->max_w * ctx->max_h) { reloc_set_error("depth image exceeds ctx capacity"); return RELOC_E_CAPACITY; }
    const int64_t npt = (int64_t)((w + step - 1) / step) * ((h + step - 1) / step);
    void *ddepth, *dout;
    int rc;
    const int esz = is_f32 ? 4 : 2;
    if ((rc = reloc_scratch(ctx, 5, (int64_t)w * h * esz, &ddepth))) return rc;
    if ((rc = reloc_scratch(ctx, 6, npt * 12 + 16, &dout))) return rc;
    int32_t *o_n = (int32_t *)((char *)dout + npt * 12);
    HIP_TRY(hipMemcpyAsync(ddepth, depth, (size_t)w * h * esz, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_depth_points, dim3(1), dim3(1024), 0, ctx->stream, (const void *)ddepth, is_f32, w, h, w, step,
                       (float)K4[2], (float)K4[3], (float)K4[0], (float)K4[1], zmin, zmax, (float *)dout, o_n);
    HIP_TRY(hipGetLastError());
    int32_t n = 0;
    HIP_TRY(hipMemcpyAsync(&n, o_n, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (n > 0) {
        HIP_TRY(hipMemcpyAsync(points, dout, (size_t)n * 12, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    *n_out = n;
    return RELOC_OK;
}
