/* reloc.h -- C-ABI of libreloc_hip.so, the MI355X (gfx950) visual-relocalization library.
 *
 * The reference (vbronetskyi/nclt-slam-project) has no FFI for this path: its boundary is the
 * Python `cv2` API at the call sites listed below, with OpenCV's CPU code underneath.  This
 * header DEFINES the C-ABI that replaces that native layer; every entry point cites the
 * reference call it serves (paths relative to simulation/isaac/ in the reference tree):
 *   M = scripts/common/visual_landmark_matcher.py     R = scripts/common/visual_landmark_recorder.py
 *   G = experiments/63_global_reloc/scripts/visual_landmark_matcher.py
 *   S = experiments/55_visual_teach_repeat/scripts/checkpoint_a_selftest.py
 * The Python binding a maintainer adds is shown in INTEGRATION.md; the shipped one is
 * nclt-slam-project_amd/_native.py.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no framework types.
 *   - every function returns 0 on success or a negative RELOC_E_* code; reloc_last_error()
 *     gives the message of the calling thread's last failure.
 *   - a reloc_ctx owns one HIP stream, its scratch buffers and (optionally) one uploaded
 *     landmark database.  A ctx is not thread-safe; any number of ctxs may coexist.
 *   - functions without a suffix take HOST pointers, run synchronously and return results in
 *     caller-allocated host memory (what the cv2-shaped Python layer needs).
 *   - functions ending in _dev take DEVICE pointers (hipMalloc / reloc_dev_alloc / a torch
 *     tensor's data_ptr()), enqueue on the ctx stream and return without synchronising.
 *   - there is NO CPU fallback: every compute entry point fails with RELOC_E_NODEVICE when no
 *     gfx950 device is usable.
 */
#ifndef RELOC_H
#define RELOC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RELOC_OK            0
#define RELOC_E_ARG        (-1)   /* bad argument (cv2.error in the Python layer)          */
#define RELOC_E_NODEVICE   (-2)   /* no usable HIP device                                  */
#define RELOC_E_HIP        (-3)   /* HIP runtime error, see reloc_last_error()             */
#define RELOC_E_CAPACITY   (-4)   /* input exceeds a ctx capacity                          */
#define RELOC_E_STATE      (-5)   /* call needs state that is missing (e.g. no database)   */

#define RELOC_ORDER_BGR 0
#define RELOC_ORDER_RGB 1

/* outcome codes of reloc_tick, one per CSV outcome string of M:308,313,383,395,428 */
#define RELOC_OUT_PUBLISHED        0
#define RELOC_OUT_NO_FEATURES      1   /* curr_no_features  */
#define RELOC_OUT_NO_CANDIDATES    2   /* no_candidates     */
#define RELOC_OUT_NO_PNP_ACCEPT    3   /* no_pnp_accept     */
#define RELOC_OUT_CONSISTENCY_FAIL 4   /* consistency_fail_ */

typedef struct reloc_ctx reloc_ctx;

/* Matcher parameters of the fused tick: the module-level constants of the reference matcher (M:56-75, accumulation
 * M:85-89, global relocalisation G:80-86).  reloc_create() installs the reference's values; a host that runs with
 * other values (MatcherConfig in the Python layer) sets them here, so the fused device tick and the cv2-shaped path
 * gate identically. */
typedef struct reloc_params {
    int32_t nfeatures;               /* ORB_create(nfeatures)                 M:207   500   */
    int32_t max_candidates;          /* MAX_CANDIDATES (<= 10)                M:57    5     */
    int32_t min_matches;             /* MIN_MATCHES                           M:65    10    */
    int32_t min_inliers;             /* MIN_INLIERS                           M:70    10    */
    int32_t ransac_iterations;       /* RANSAC_ITERATIONS (<= 1024)           M:69    200   */
    int32_t global_max_candidates;   /* RELOC_MAX_CANDIDATES (<= 32)          G:84    25    */
    int32_t global_min_inliers;      /* RELOC_MIN_INLIERS                     G:85    18    */
    int32_t accum_min_kpts;          /* ACCUM_MIN_KPTS                        M:88    30    */
    double candidate_radius_m;       /* CANDIDATE_RADIUS_M                    M:56    8.0   */
    double heading_tol_deg;          /* HEADING_TOL_DEG                       M:58    90.0  */
    double reproj_max_px;            /* REPROJ_ERR_MAX_PX                     M:67    2.0   */
    double ransac_reproj_px;         /* RANSAC_REPROJ_PX                      M:68    3.0   */
    double ransac_confidence;        /* cv2.solvePnPRansac default                    0.99  */
    double consistency_m;            /* CONSISTENCY_M                         M:75    5.0   */
    double global_reproj_max_px;     /* RELOC_REPROJ_MAX_PX                   G:86    1.5   */
    double accum_min_dist_m;         /* ACCUM_MIN_DIST_M                      M:87    5.0   */
    double accum_depth_min_m;        /* d_c > 0.5                             M:461   0.5   */
    double accum_depth_max_m;        /* d_c < 15.0                            M:461   15.0  */
    int32_t gray_coeff_bits;         /* BGR2GRAY fixed point: 15 or 14 (reloc_spec.h) M:305  15    */
    int32_t reserved0;               /* must be 0                                                  */
} reloc_params;

/* ---- lifetime, errors, plumbing ----------------------------------------------------------- */
const char *reloc_last_error(void);
int  reloc_device_count(void);
/* max_w/max_h: largest frame; max_feat: largest keypoint / descriptor count per frame. */
reloc_ctx *reloc_create(int device, int max_w, int max_h, int max_feat);
void reloc_destroy(reloc_ctx *ctx);
/* Use an externally owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL
 * restores the ctx's own stream. */
int  reloc_set_stream(reloc_ctx *ctx, void *hip_stream);
int  reloc_sync(reloc_ctx *ctx);
void *reloc_dev_alloc(reloc_ctx *ctx, int64_t bytes);
int  reloc_dev_free(reloc_ctx *ctx, void *p);
int  reloc_h2d(reloc_ctx *ctx, void *dst_dev, const void *src_host, int64_t bytes);   /* async */
int  reloc_d2d(reloc_ctx *ctx, void *dst_dev, const void *src_dev, int64_t bytes);    /* async */
/* Pinned (page-locked) host memory for frames and results: copies from / to it are true DMA and overlap with kernels. */
void *reloc_host_alloc(int64_t bytes);
int  reloc_host_free(void *p);
/* The ctx's current hipStream_t (to order work of other libraries against it). */
void *reloc_get_stream(reloc_ctx *ctx);
int  reloc_d2h(reloc_ctx *ctx, void *dst_host, const void *src_dev, int64_t bytes);   /* async */
/* HIP-event stopwatch on the ctx stream: begin(); ...launches...; end() -> ms (synchronises). */
int  reloc_timer_begin(reloc_ctx *ctx);
int  reloc_timer_end(reloc_ctx *ctx, float *ms);
/* Per-kernel stopwatch: when enabled, the dominant kernels are bracketed by HIP events on the
 * ctx stream; get() synchronises and returns total ms and launch count of kernel `which`. */
#define RELOC_PROF_DB_SCAN  0
#define RELOC_PROF_MATRIX   1
#define RELOC_PROF_ORB      2
#define RELOC_PROF_PNP      3
#define RELOC_PROF_N        4
int  reloc_profile_enable(reloc_ctx *ctx, int on);
int  reloc_profile_get(reloc_ctx *ctx, int which, float *total_ms, int32_t *launches);

int  reloc_get_params(reloc_ctx *ctx, reloc_params *out);
int  reloc_set_params(reloc_ctx *ctx, const reloc_params *p);

/* ---- ORB front end ------------------------------------------------------------------------ */
/* cv2.cvtColor(img, COLOR_BGR2GRAY)                                         M:305  R:240  S:44 */
int reloc_gray_u8(reloc_ctx *ctx, const uint8_t *img, int w, int h, int stride, int order,
                  uint8_t *gray);
/* cv2.ORB_create(nfeatures).detectAndCompute(gray, None)                    M:306  R:241  S:48
 * Outputs hold up to the ctx's max_feat rows; *n_out = rows written.  Keypoints are level-major,
 * raster order inside a level.  xy = KeyPoint.pt (level-0 pixels), octave = pyramid level. */
int reloc_orb_detect_compute(reloc_ctx *ctx, const uint8_t *gray, int w, int h, int stride,
                             int nfeatures, float *xy, float *size, float *angle, float *response,
                             int32_t *octave, uint8_t *desc, int32_t *n_out);
/* Same from a 3-channel frame already in device memory (gray conversion fused in front);
 * results stay in ctx-owned device buffers (see reloc_frame_*).  Enqueue only. */
int reloc_orb_frame_dev(reloc_ctx *ctx, const uint8_t *img_dev, int w, int h, int stride, int order,
                        int nfeatures);
/* Device views of the last frame's features (valid until the next frame call). */
const uint8_t *reloc_frame_desc_dev(reloc_ctx *ctx);      /* max_feat x 32 u8                  */
const float   *reloc_frame_xy_dev(reloc_ctx *ctx);        /* max_feat x 2 f32                  */
const int32_t *reloc_frame_count_dev(reloc_ctx *ctx);     /* 1 x i32: number of keypoints      */
/* Debug/parity taps: copy intermediate planes of the last frame to host.  level in [0, 8).
 * what: 0 = pyramid level, 1 = blurred level, 2 = NMS-kept FAST score map. */
int reloc_frame_debug_plane(reloc_ctx *ctx, int what, int level, uint8_t *out, int32_t *w, int32_t *h);

/* Teach-side record builder (R:240-288): ORB on the frame, then per keypoint the border / ground masks,
 * depth lookup (uint16 millimetres), 3x3 non-zero depth std, range and variance gates and pin-hole
 * back-projection.  Outputs (up to max_feat rows, keypoint order kept): xy = keypoints_2d, desc =
 * descriptors, pts3d = keypoints_3d_cam, kp_index = row in the frame's ORB output.  The caller applies
 * the ">= 30 survivors" rule (R:270) and the displacement trigger. */
int reloc_record_frame(reloc_ctx *ctx, const uint8_t *img, const uint16_t *depth_mm, int w, int h, int order,
                       int nfeatures, float *xy, uint8_t *desc, float *pts3d, int32_t *kp_index,
                       int32_t *n_out, int32_t *n_kp);

/* Depth image -> obstacle points of the relay's depth_cb (scripts/common/tf_wall_clock_relay_v55.py:1020-1038):
 * every `step`-th pixel with zmin < z < zmax, point = (z, -(u-cx)/fx*z, -(v-cy)/fy*z) f32, raster order.
 * depth: float32 metres (is_f32) or uint16 millimetres; points holds ceil(w/step)*ceil(h/step) x 3. */
int reloc_depth_points(reloc_ctx *ctx, const void *depth, int is_f32, int w, int h, int step, const double K4[4],
                       float zmin, float zmax, float *points, int32_t *n_out);

/* ---- 256-bit Hamming matching --------------------------------------------------------------- */
/* cv2.BFMatcher(NORM_HAMMING, crossCheck=True).match(q, t)                  M:211,327  G:337
 * Mutual nearest neighbours, lowest index on ties, sorted by queryIdx.  Outputs sized min(nq,nt). */
int reloc_match_mutual(reloc_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt,
                       int32_t *qidx, int32_t *tidx, int32_t *dist, int32_t *n_out);
/* cv2.BFMatcher(NORM_HAMMING, crossCheck=False).knnMatch(q, t, k=2)         S:46,68
 * idx/dist are nq x 2; a missing second neighbour is (-1, -1). */
int reloc_match_knn2(reloc_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt,
                     int32_t *idx, int32_t *dist);
/* landmarks.pkl -> packed device arena (R:290-297, M:179-187).  desc: T x 32; pts3d: T x 3
 * (keypoints_3d_cam); offsets: L+1 row offsets; poses: L x 7 camera pose (x y z qx qy qz qw).
 * Replaces any previously uploaded database. */
int reloc_db_upload(reloc_ctx *ctx, const uint8_t *desc, const float *pts3d, const int64_t *offsets,
                    const double *poses, int64_t n_records);
int64_t reloc_db_records(reloc_ctx *ctx);
int64_t reloc_db_rows(reloc_ctx *ctx);
/* The database lives in a capacity-reserved arena: appending a record (accumulation, M:435-500) copies its rows behind
 * the last one and never re-allocates while the reserve lasts; reserve() grows the arena (device-to-device copy of
 * the contents) and is also what upload/append call when the reserve is exhausted (geometric growth). */
int reloc_db_reserve(reloc_ctx *ctx, int64_t cap_records, int64_t cap_rows);
/* self.landmarks.append(new_lm); self.xy = vstack(...); self.heading = append(...)        M:489-493
 * desc n x 32, pts3d n x 3 (keypoints_3d_cam), kp2d n x 2 or NULL (keypoints_2d, kept for reloc_db_fetch), pose =
 * camera pose x y z qx qy qz qw, index_xy = the (x, y) the candidate search files the record under (the reference
 * files an accumulated record under the VIO position, M:491; NULL = pose x, y as for taught records M:213). */
int reloc_db_append(reloc_ctx *ctx, const uint8_t *desc, const float *pts3d, const float *kp2d, int n,
                    const double pose[7], const double index_xy[2]);
/* Two resident databases (slot 0 / 1): the split-landmark variant keeps the outbound set in one and the return-leg
 * set in the other and flips at the turnaround (X:274-294).  upload / append / reserve / tick act on the selected
 * slot; selecting costs nothing on the device. */
int reloc_db_select(reloc_ctx *ctx, int slot);
/* Several contexts (streams) on one device scanning ONE resident database: dst adopts the selected database of src
 * (descriptors, points, offsets, poses, index) without copying and keeps only its own per-tick scratch.  The shared
 * database is read-only through dst (append / reserve on dst fail with RELOC_E_STATE).  dst sees the database as it was
 * when it was adopted: records that src appends or accumulates later, and a new upload through src, need a new
 * reloc_db_share to become visible.  The arrays are reference-counted: src may grow past its reserve, upload again or be
 * destroyed at any time -- dst keeps scanning the arrays it adopted, which are freed when their last holder lets go. */
int reloc_db_share(reloc_ctx *dst, reloc_ctx *src);
/* Read one record back (save of the augmented database M:502-514, parity taps).  Any output may be NULL; desc / pts3d /
 * kp2d must hold the record's rows (query *n first with all arrays NULL). */
int reloc_db_fetch(reloc_ctx *ctx, int64_t record, uint8_t *desc, float *pts3d, float *kp2d, double pose[7],
                   double index_xyh[4], int32_t *n);
/* Whole-database scan of G:329-344: per record, the number of mutual matches between the
 * record's descriptors (query) and the current frame's descriptors (train).  cur: n_cur x 32. */
int reloc_db_match_counts(reloc_ctx *ctx, const uint8_t *cur, int n_cur, int32_t *counts);
int reloc_db_match_counts_dev(reloc_ctx *ctx, const uint8_t *cur_dev, const int32_t *n_cur_dev,
                              int n_cur_max, int32_t *counts_dev);
/* Ratio-test score of every record: number of current descriptors whose two nearest rows of the record
 * satisfy d1 < ratio * d2 -- knnMatch(desc_cur, desc_record, k=2) + Lowe test of the archived anchor
 * localizers (_archive/anchor_localizer.py:82-90, ratio 0.75) and the self-test (S:68-71, 0.80). */
int reloc_db_ratio_counts(reloc_ctx *ctx, const uint8_t *cur, int n_cur, double ratio, int32_t *counts);
/* All-pairs u16 distance matrix (loop-closure sweep shape; no reference counterpart,
 * BASELINE.json config 5): out[i * nb + j] = hamming(a_i, b_j). */
int reloc_hamming_matrix(reloc_ctx *ctx, const uint8_t *a, int64_t na, const uint8_t *b, int64_t nb,
                         uint16_t *out);
int reloc_hamming_matrix_dev(reloc_ctx *ctx, const uint8_t *a_dev, int64_t na, const uint8_t *b_dev,
                             int64_t nb, uint16_t *out_dev);

/* ---- PnP ------------------------------------------------------------------------------------ */
/* Hypothesis scorer inside cv2.solvePnPRansac (M:342-346): inlier count of H poses
 * (R row-major 9 + t 3 doubles each) over m correspondences; K4 = fx fy cx cy. */
int reloc_pnp_score(reloc_ctx *ctx, const float *obj, const float *img, int m, const double *Rt, int H,
                    const double K4[4], float thr_px, int32_t *inlier_count, uint8_t *mask);
/* cv2.solvePnPRansac(obj, img, K, DIST, iterationsCount=, reprojectionError=, flags=ITERATIVE)
 *                                                                            M:342-346  S:78-82
 * inliers must hold m entries. */
int reloc_pnp_ransac(reloc_ctx *ctx, const float *obj, const float *img, int m, const double K4[4],
                     int iters, float thr_px, double conf, uint64_t seed, double rvec[3],
                     double tvec[3], int32_t *inliers, int32_t *n_inl, int32_t *ok);

/* Camera model used by the fused tick: K4 = fx fy cx cy (M:49-52) and the static base_link ->
 * camera transform stored in landmarks.pkl (M:183-184; R:81-88).  NULL keeps the current value;
 * the defaults are the reference's constants. */
int reloc_set_camera(reloc_ctx *ctx, const double K4[4], const double base_to_cam_t[3],
                     const double base_to_cam_R[9]);

/* ---- fused tick ----------------------------------------------------------------------------- */
#define RELOC_TICK_LOCAL   0   /* candidates by VIO distance / heading only                          M:293-302 */
#define RELOC_TICK_GLOBAL  1   /* whole-database search unconditionally (the benchmarked shape)       G:329-344 */
#define RELOC_TICK_AUTO    2   /* local first; whole-database search only if it finds no candidate -- the host passes
                                  this mode when G's silence and drift conditions hold                G:324-326 */
/* One repeat tick against the selected database (M:281-433 with G:315-344's whole-database
 * candidate search according to global_reloc = RELOC_TICK_*): gray -> ORB -> candidates -> mutual match ->
 * PnP-RANSAC -> reprojection gate -> pose compose -> best by inliers -> consistency gate.
 * base_pose: x y z qx qy qz qw of base_link (the /tmp/isaac_pose.txt line).
 * Outputs: anchor_pose[7] (base_link in the teach map), n_inl, reproj (px), lm_idx, outcome,
 * n_candidates.  Host pointers; synchronous. */
int reloc_tick(reloc_ctx *ctx, const uint8_t *img, int w, int h, int order, const double base_pose[7],
               int global_reloc, uint64_t seed, double anchor_pose[7], int32_t *n_inl, float *reproj,
               int32_t *lm_idx, int32_t *outcome, int32_t *n_candidates);
/* Same with the frame already resident in device memory; enqueues everything and leaves the
 * result in a ctx-owned device record readable with reloc_tick_result(). */
int reloc_tick_dev(reloc_ctx *ctx, const uint8_t *img_dev, int w, int h, int order,
                   const double base_pose[7], int global_reloc, uint64_t seed);
/* Batched relocalization (BASELINE.json config 4): n <= 8 frames, one context per frame; the contexts share ONE stream
 * (reloc_set_stream), one device and one database (reloc_db_share).  ORB of every frame, then ONE launch that scans the
 * database for all n frames (the launch-fixed cost of the scan is paid once per batch), then ranking / matches / PnP /
 * gates per frame.  Results per context as after reloc_tick_dev.  base_poses: n x 7, seeds: n or NULL. */
int reloc_tick_batch_dev(reloc_ctx *const *ctxs, int n, const uint8_t *const *imgs_dev, int w, int h, int order,
                         const double *base_poses, int global_reloc, const uint64_t *seeds);
/* The result of the last tick enqueued on ctx.  WAITS FOR THAT TICK'S RECORD ONLY (reloc_tick_wait below), not for the stream:
 * since round 3 this call is no stream barrier -- work enqueued behind the tick (reloc_tick_accumulate_dev, copies, the scan half
 * of a next frame) may still be running when it returns; call reloc_sync() before touching anything such work writes.  If the
 * last tick entry point on ctx returned an error before its result record was enqueued, this returns RELOC_E_STATE (it never
 * hands out the previous tick's record in its place). */
int reloc_tick_result(reloc_ctx *ctx, double anchor_pose[7], int32_t *n_inl, float *reproj,
                      int32_t *lm_idx, int32_t *outcome, int32_t *n_candidates);
/* Waits for the result record of the last tick enqueued on ctx -- and only for that: the tick's last kernel stores the
 * record into pinned host memory and, behind it, a sequence stamp; this call polls the stamp instead of going through the
 * stream-completion path of the runtime (10-50 us cheaper per synchronous tick, most after an idle period).  Falls back to
 * a stream synchronisation after 20 ms.  reloc_tick_result() / reloc_tick_result_ex() wait this way themselves. */
int reloc_tick_wait(reloc_ctx *ctx);
/* Device address of the 96-byte result record of the last tick (layout of reloc_tick_result_ex's outputs: double
 * anchor_pose[7], double reproj, int32 n_inl, lm_idx, outcome, n_candidates, n_features, relocating): lets a pipelined
 * host copy results with reloc_d2h into pinned memory without synchronising per frame. */
const void *reloc_tick_result_dev(reloc_ctx *ctx);
/* Streaming without a copy: the ticks enqueued after this call ALSO write their 96-byte result record (same layout) to
 * `pinned_record`, memory from reloc_host_alloc() -- the last kernel of the tick stores it over PCIe itself, so the record
 * is complete once the ctx stream has passed that tick (event / reloc_sync) -- or as soon as its int32 word 22 (byte 88)
 * holds the tick's sequence stamp, which the kernel stores last, system-scope release.  Point it at a different record before each
 * tick to keep one per frame; NULL stops it.  (reloc_tick_result() reads an internal record of the same kind.) */
int reloc_tick_result_to(reloc_ctx *ctx, void *pinned_record);
/* Deployment hint, no effect on results: `on` > 0 says this context is the only stream of work on the GPU (one robot,
 * one camera).  The whole-database scan then runs as ONE resident generation of workgroups (no early hand-over of CU
 * slots to other streams' kernels) and the tick's small kernels take the shapes that are fastest alone (8 waves per
 * record in the emit pass, full register set in the PnP refinement): ~12 us less per synchronous global tick, at the
 * price of ~10 % throughput if several such contexts do run side by side.  `on` == 0: it is not alone.  `on` < 0 (the
 * default): decided per call -- alone while it is the only live context this process has created with reloc_create. */
int reloc_set_exclusive(reloc_ctx *ctx, int on);
/* Same plus n_features and the `relocating` flag (1 when the whole-database search produced the candidates, G:344); waits like
 * reloc_tick_result (the tick's record only; RELOC_E_STATE after a tick that failed to enqueue). */
int reloc_tick_result_ex(reloc_ctx *ctx, double anchor_pose[7], int32_t *n_inl, double *reproj, int32_t *lm_idx,
                         int32_t *outcome, int32_t *n_candidates, int32_t *n_features, int32_t *relocating);
/* Accumulation (M:435-500), enqueued behind a tick on the same ctx: when the tick's outcome is no_candidates /
 * no_pnp_accept / consistency_fail (M:314,384,396), silence_ok is set (the host's `ts - last_anchor_ts >=
 * ACCUM_SILENCE_S`, M:441) and no record is filed within accum_min_dist_m of the robot, the current frame's keypoints
 * with valid depth become a new record at the tail of the arena (>= accum_min_kpts of them).  depth_mm_dev: (h, w)
 * uint16 millimetres in device memory.  The arena must have room for one record of max_feat rows (reloc_db_reserve);
 * reloc_accumulate_result() synchronises, reports what happened and makes the new record visible to later ticks. */
int reloc_tick_accumulate_dev(reloc_ctx *ctx, const uint16_t *depth_mm_dev, int w, int h, const double base_pose[7],
                              int silence_ok);
int reloc_accumulate_result(reloc_ctx *ctx, int32_t *appended, int32_t *n_kpts, double *nearest_m);
/* Parity tap: per-candidate records of the last tick (arrays of 32 entries; Rt 32 x 12).  Three shapes, by what the tick's inlier
 * gate (min_inliers; global_min_inliers in a whole-database search) lets a candidate become:
 *   consensus set >= gate         refined: ok 1, n_inl, the refined pose, reproj = mean reprojection error of the inliers
 *   matches >= gate > consensus   cannot be accepted, not refined: ok as RANSAC left it, n_inl = the consensus size, the RANSAC pose, reproj 0
 *   matches < gate                no hypothesis is drawn at all (its consensus set cannot reach the gate): ok 0, n_inl 0, reproj 0
 * With the gate at 0 every candidate takes the first shape (tests/test_gpu_tick.py::test_inlier_gate_shapes_of_the_parity_tap). */
int reloc_tick_debug(reloc_ctx *ctx, int32_t *cand_ids, int32_t *n_cand, int32_t *n_matches,
                     int32_t *n_inl, int32_t *ok, double *reproj, double *Rt);
/* Sharded database (one rank per GPU): per-record mutual-match counts are local; the caller
 * exchanges the per-shard top-k (count, global id) lists and tells each rank which of ITS
 * records made the global top-k.  These two calls split reloc_tick_dev at that exchange. */
int reloc_tick_scan_dev(reloc_ctx *ctx, const uint8_t *img_dev, int w, int h, int order,
                        const double base_pose[7] /* NULL: no heading mask */,
                        int32_t *topk_ids_dev, int32_t *topk_counts_dev, int k);
int reloc_tick_solve_dev(reloc_ctx *ctx, const int32_t *cand_ids_dev, int n_cand,
                         const double base_pose[7], int check_consistency, uint64_t seed);
/* The same two halves for a BATCH of n <= 8 frames (BASELINE.json config 4) with the exchange kept in device memory: one
 * call per half, everything enqueued on the ONE stream the n contexts share (as for reloc_tick_batch_dev: one device, one
 * shard through reloc_db_share, equal capacity and parameters), so the caller's collective (RCCL all-gather of the rows
 * below) can be enqueued on that same stream between them and the host never waits inside a batch.
 *   scan half:  ORB per frame, ONE scan launch for the n frames, per-frame ranking.  scan_out_dev: n rows of 2k + 2
 *               int32 = k GLOBAL record ids (id_base + local id, -1 padded), k mutual-match counts, the frame's feature
 *               count, 0.  base_poses: n x 7 (heading mask, G:329-330) or NULL.
 *   merge:      what every rank computes from the gathered rows (all_scan_dev: `world` blocks `stride_rank` int32 apart,
 *               each n rows as above): per frame the k best (count desc, global id desc; G:342-343) -> win_gid (n x k,
 *               -1 padded), cand_local (n x k: local ids of the winners THIS rank owns, -1 elsewhere), n_feat (n: the
 *               largest feature count any rank reported, -1 from ranks without records).
 *   solve half: match lists + PnP + gates (relocation gates, no consistency check, G:381-382,424) for the owned winners of
 *               every frame; the 96-byte result record of frame f is stored at res_out + 96 f by the tick's last kernel
 *               (device or pinned host memory). */
int reloc_shard_scan_batch_dev(reloc_ctx *const *ctxs, int n, const uint8_t *const *imgs_dev, int w, int h, int order,
                               const double *base_poses, int k, int64_t id_base, int32_t *scan_out_dev);
int reloc_shard_merge_dev(reloc_ctx *ctx, const int32_t *all_scan_dev, int world, int64_t stride_rank, int n, int k,
                          int64_t id_base, int64_t n_local, int32_t *win_gid_dev, int32_t *cand_local_dev,
                          int32_t *n_feat_dev);
int reloc_shard_solve_batch_dev(reloc_ctx *const *ctxs, int n, const int32_t *cand_local_dev, int k,
                                const double *base_poses, const uint64_t *seeds, void *res_out);

#ifdef __cplusplus
}
#endif
#endif /* RELOC_H */
