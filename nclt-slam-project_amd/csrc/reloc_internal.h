// reloc_internal.h -- shared declarations of libreloc_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/reloc.h"
#include "../../include/reloc_spec.h"

#define RELOC_API extern "C" __attribute__((visibility("default")))
#define RELOC_PROF_RING 256      /* event pairs per stopwatch before reloc_prof_begin has to wait for the device */

void reloc_set_error(const char *fmt, ...);

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            reloc_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                            __LINE__);                                                     \
            return RELOC_E_HIP;                                                            \
        }                                                                                  \
    } while (0)

#define ARG_CHECK(cond, msg)                                 \
    do {                                                     \
        if (!(cond)) {                                       \
            reloc_set_error("bad argument: %s", msg);        \
            return RELOC_E_ARG;                              \
        }                                                    \
    } while (0)

// argument check of an entry point that takes a ctx: also makes the ctx's device current, so one process
// may drive several GPUs through several ctxs
#define ARG_CHECK_CTX(c, cond, msg)                           \
    do {                                                      \
        if (!(c) || !(cond)) {                                \
            reloc_set_error("bad argument: %s", msg);         \
            return RELOC_E_ARG;                               \
        }                                                     \
        (void)hipSetDevice((c)->device);                      \
    } while (0)

constexpr int NLEV = RELOC_ORB_NLEVELS;
constexpr int MAX_REC_ROWS = 4096;   // largest record (teach rows) the fused scan accepts
constexpr int MAX_CAND = 32;         // PnP candidates per tick (5 local / 25 global)
constexpr int MAX_HYP = 1024;        // RANSAC hypotheses per candidate (iterationsCount)
constexpr int64_t MAX_DB_RECORDS = 0xFFFFF;   // the local-candidate key keeps the record index in 20 bits

// Wave issue priority of the tick's small kernels (s_setprio 3; the default, and the scan's, is 0).  With several streams
// on the chip a SIMD holds four scan waves and a wave or two of some other stream's ORB / ranking / PnP kernel; those
// kernels are short dependent chains, and at equal priority they get one issue slot in five, so every stream's non-scan
// phase stretches 2-3x (k_pyramid 49 vs 17 us, k_fast_blur 46 vs 16) and the scans of the streams overlap less.  With
// priority the same instructions are issued, only sooner: 4-stream run 6060 -> 6250 frames/s, single-stream ticks unchanged.
// -DRELOC_SMALL_PRIO=0 switches it off (A/B builds).
// Register budget of the scan kernels (k_db_scan, k_db_scan_batch), capped with amdgpu_num_vgpr (0 = the compiler's choice: 112).
// Four scan waves per SIMD at 112 registers leave 64 of the 512 per lane to whatever else wants to run beside them -- and
// k_pyramid (94), k_pnp_hyp (94), k_tick_finalize (80) do not fit into 64: they wait for a scan workgroup to retire.  At 104
// (28 bytes of scratch) four scan waves leave 96 and a fifth still does not fit (5 x 104 > 512); at 96 a fifth does, and
// nothing else any more.  4-stream run, interleaved on one box (profiles/r4_scan_vgpr.log): 112 / 104 / 96 registers = 6 648 /
// 6 716 / 6 560 frames/s, synchronous whole-database tick 273 / 269 / 280 us.  (Cutting k_pnp_finish and the emit pass to 96 registers so
// that they fit beside the scans as well: nothing with three scan generations, 6 747-6 760 vs 6 753 frames/s; with the one generation of
// today 4-7 % WORSE, 6 320-6 550 vs 6 822 -- they are better off in the gap between two scans than competing with one; same log.)
#ifndef RELOC_SCAN_NUM_VGPR
#define RELOC_SCAN_NUM_VGPR 104
#endif
#if RELOC_SCAN_NUM_VGPR > 0
// (the backend doubles the attribute's value on targets with the unified VGPR / AGPR file: half of the wanted count is passed)
#define RELOC_SCAN_VGPR_ATTR __attribute__((amdgpu_num_vgpr(RELOC_SCAN_NUM_VGPR / 2)))
#else
#define RELOC_SCAN_VGPR_ATTR
#endif
// Scheduling form of the whole-database scan in a context that shares the chip: n > 0 = n generations of workgroups, each with
// a row budget, sweepers behind them; 0 = ONE resident generation that draws records until the counters are dry (what a lone
// context has always run).  Rounds 2-3: three generations, so that other streams' small kernels found slots when a generation
// retired.  Since the scan leaves them 96 registers they run BESIDE resident scans: 4 streams, interleaved on one box
// (profiles/r4_scan_generations.log), 1 / 2 / 3 / 4 generations 6 853 / 6 759 / 6 752 / 6 604 frames/s, one generation without
// budgets 6 826 -- and that form scans in 152 us on an idle chip where one generation WITH budgets takes 177 (static shares
// leave a tail).  A batched launch (k_db_scan_batch) keeps one generation of budgets per frame.
#ifndef RELOC_SCAN_GENS_SHARED
#define RELOC_SCAN_GENS_SHARED 0
#endif
#ifndef RELOC_SMALL_PRIO
#define RELOC_SMALL_PRIO 3
#endif
#define RELOC_SMALL_KERNEL_PRIO() __builtin_amdgcn_s_setprio(RELOC_SMALL_PRIO)

struct OrbLevel {
    int w, h;          // level size
    int stride;        // row stride in bytes (multiple of 64)
    int64_t off;       // byte offset of the level inside a pyramid buffer
    float scale;
    int quota;
};

// Result record of one tick (device resident, copied out by reloc_tick_result)
struct TickResult {
    double anchor_pose[7];
    double reproj;
    int32_t n_inl;
    int32_t lm_idx;
    int32_t outcome;
    int32_t n_candidates;
    int32_t n_features;
    int32_t relocating;   // candidates came from the whole-database search (G:344)
    int32_t pad[2];       // 96 bytes; pad[0] of a record in HOST memory: the tick's sequence stamp, stored last (reloc_tick_wait)
};
static_assert(sizeof(TickResult) == 96, "TickResult is the 96-byte record documented in include/reloc.h");

// what reloc_tick_accumulate_dev left behind (device resident)
struct AccumResult {
    double nearest_m;     // distance to the nearest filed record (-1: not evaluated)
    int32_t appended;     // 1: a record was written behind the arena's last one
    int32_t n_kpts;       // keypoints with valid depth
};

// One resident landmark database: structure of arrays with reserved capacity (rows: cap_rows, records: cap_records).
// The six database arrays of one arena (desc, pts3d, kp2d, off, pose, xy_heading) are held by reference count: the
// context that uploaded them and every context that adopted them with reloc_db_share hold one reference each, and the
// arrays are freed by whoever lets go last.  An owner that re-allocates (growth past the reserve, a new upload) or is
// destroyed therefore never frees memory an adopter still scans: the adopter keeps the arrays -- and the record count --
// it adopted until it calls reloc_db_share / reloc_db_upload again.
struct DbShare { int refs = 1; };

struct DbArena {
    DbShare *share = nullptr;
    int64_t cap_records = 0, cap_rows = 0;
    int64_t records = 0, rows = 0;
    int max_rows = 0;
    uint8_t *desc = nullptr;          // cap_rows x 32
    float *pts3d = nullptr;           // cap_rows x 3
    float *kp2d = nullptr;            // cap_rows x 2 (keypoints_2d; only kept for reloc_db_fetch)
    int64_t *off = nullptr;           // cap_records + 1
    double *pose = nullptr;           // cap_records x 7
    double *xy_heading = nullptr;     // cap_records x 4
    int32_t *counts = nullptr;        // cap_records
    unsigned long long *topk_part = nullptr;
    int topk_blocks = 0;
};

// Per-candidate PnP output
struct PnpOut {
    double Rt[12];       // refined pose (teach cam -> current cam)
    double rvec[3];
    double reproj_mean;  // mean inlier reprojection error (px) under the refined pose
    int32_t ok;
    int32_t n_inl;
    int32_t best_h;
    int32_t n_matches;
};

// ---- wave-wide reductions in the VALU's DPP network -------------------------------------------------
// quad_perm / mirror steps leave a row's result in all of its 16 lanes, row_bcast15 / row_bcast31 carry it
// across the four rows, lane 63 ends with the total.  A chain of such reductions costs a few VALU
// instructions each instead of six dependent ds_bpermute round trips.  All 64 lanes must be active.
__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
    int x = (int)v;
#define RELOC_DPP_STEP(ctrl, rmask)                                                                     \
    {                                                                                                   \
        const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(x, x, ctrl, rmask, 0xF, false);         \
        x = (int)((unsigned)x > o ? (unsigned)x : o);                                                    \
    }
    RELOC_DPP_STEP(0xB1, 0xF)    // quad_perm [1,0,3,2]
    RELOC_DPP_STEP(0x4E, 0xF)    // quad_perm [2,3,0,1]
    RELOC_DPP_STEP(0x141, 0xF)   // row_half_mirror
    RELOC_DPP_STEP(0x140, 0xF)   // row_mirror
    RELOC_DPP_STEP(0x142, 0xA)   // row_bcast15 -> rows 1, 3
    RELOC_DPP_STEP(0x143, 0xC)   // row_bcast31 -> rows 2, 3
#undef RELOC_DPP_STEP
    return (unsigned)__builtin_amdgcn_readlane(x, 63);
}

// value of lane (l ^ K) for K = 1, 2, 4, 8 through DPP (quad permutes, row shifts with bank masks, row rotate)
template <int K>
__device__ __forceinline__ unsigned dpp_xor(unsigned v)
{
    static_assert(K == 1 || K == 2 || K == 4 || K == 8, "in-row exchanges only");
    const int x = (int)v;
    if constexpr (K == 1) return (unsigned)__builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false);
    else if constexpr (K == 2) return (unsigned)__builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false);
    else if constexpr (K == 8) return (unsigned)__builtin_amdgcn_update_dpp(x, x, 0x128, 0xF, 0xF, false);
    else {
        const int t = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xF, 0x5, false);      // banks 0, 2 <- lane + 4
        return (unsigned)__builtin_amdgcn_update_dpp(t, x, 0x114, 0xF, 0xA, false);    // banks 1, 3 <- lane - 4
    }
}

// integer sum (exact in any order); every lane gets the total
__device__ __forceinline__ int wave_sum_i32(int x)
{
#define RELOC_DPP_STEP(ctrl, rmask) x += __builtin_amdgcn_update_dpp(0, x, ctrl, rmask, 0xF, false);
    RELOC_DPP_STEP(0xB1, 0xF)
    RELOC_DPP_STEP(0x4E, 0xF)
    RELOC_DPP_STEP(0x141, 0xF)
    RELOC_DPP_STEP(0x140, 0xF)
    RELOC_DPP_STEP(0x142, 0xA)
    RELOC_DPP_STEP(0x143, 0xC)
#undef RELOC_DPP_STEP
    return __builtin_amdgcn_readlane(x, 63);
}

// ---- heading test shared by the scan and the candidate kernels (M:296-301, G:329-330) --------------
__device__ __forceinline__ void quat_to_rot(double qx, double qy, double qz, double qw, double R[9])
{
    R[0] = 1 - 2 * (qy * qy + qz * qz); R[1] = 2 * (qx * qy - qz * qw);     R[2] = 2 * (qx * qz + qy * qw);
    R[3] = 2 * (qx * qy + qz * qw);     R[4] = 1 - 2 * (qx * qx + qz * qz); R[5] = 2 * (qy * qz - qx * qw);
    R[6] = 2 * (qx * qz - qy * qw);     R[7] = 2 * (qy * qz + qx * qw);     R[8] = 1 - 2 * (qx * qx + qy * qy);
}

// |wrap(teach_hdg - cur_hdg)| < tol  <=>  cos(teach_hdg - cur_hdg) > cos(tol); the database index keeps
// (cos, sin) of every record's heading, so the test is one dot product.
__device__ __forceinline__ bool heading_ok(const double *__restrict__ rec4, double cc, double sc, double cos_tol)
{
    return rec4[2] * cc + rec4[3] * sc > cos_tol;
}

// (cos, sin) of the robot's heading from the base_link quaternion (x, y, z, w)
__device__ __forceinline__ void cur_heading_q(const double q[4], double &cc, double &sc)
{
    double Rb[9];
    quat_to_rot(q[0], q[1], q[2], q[3], Rb);
    const double n = sqrt(Rb[0] * Rb[0] + Rb[3] * Rb[3]);       // fwd = (R00, R10); only the direction matters
    cc = n > 0 ? Rb[0] / n : 1.0;
    sc = n > 0 ? Rb[3] / n : 0.0;
}

// Optional heading mask of the whole-database scan: records whose teach heading is incompatible with the
// robot's are not scored (count 0), exactly the records the reference skips at G:329-330.  xyh == NULL: no mask.
struct ScanMask {
    const double *xyh;
    double q[4];
    double cos_tol = 6.123233995736766e-17;      // cos(HEADING_TOL_DEG = 90 degrees) in double
    const int32_t *skip_if = nullptr;            // RELOC_TICK_AUTO: the whole launch stands down when *skip_if != 0
    // emit mode only (M:333-336): when g_obj is set, every mutual match also leaves its 3-D / 2-D pair
    // (keypoints_3d_cam[queryIdx], pts_curr_2d[trainIdx]) next to its index triplet, so no gather launch follows
    const float *g_pts3d = nullptr, *g_xy = nullptr;
    float *g_obj = nullptr, *g_img = nullptr;
};

struct reloc_ctx {
    int device = 0;
    int max_w = 0, max_h = 0, max_feat = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    int num_cu = 256;

    // profiling
    int prof_on = 0;
    // ring of event pairs per stopwatch: large enough that a measurement of a few hundred launches never has to wait for
    // the device in the middle (prof_flush synchronises)
    struct ProfSlot { hipEvent_t a[RELOC_PROF_RING], b[RELOC_PROF_RING]; int n = 0; float total = 0.f; int launches = 0; bool init = false; } prof[RELOC_PROF_N];

    // generic scratch (grown on demand by host-pointer entry points)
    void *scratch[8] = {};
    int64_t scratch_bytes[8] = {};

    // ---- ORB state ----
    int orb_w = 0, orb_h = 0, orb_nfeat = 0;     // geometry the tables below were built for
    OrbLevel lev[NLEV];
    int64_t pyr_bytes = 0;
    uint8_t *pyr = nullptr;        // pyramid levels
    uint8_t *blur = nullptr;       // blurred levels
    uint8_t *nms = nullptr;        // NMS-kept FAST score maps (stage-1 source, parity tap)
    int32_t *rz_tab = nullptr;     // resize tables (device)
    void *pyr_tiles = nullptr;     // PyrTile per workgroup of k_pyramid (device)
    int pyr_ntiles = 0, pyr_lds[NLEV + 1] = {}, pyr_lds_bytes = 0;   // LDS offsets of the level buffers, then the tables
    int32_t *hist = nullptr;       // NLEV x 256 score histograms
    int32_t *cand_cnt = nullptr;   // NLEV counters (stage-1 list sizes)
    uint32_t *cand_key = nullptr;  // NLEV x STAGE1_CAP packed (y<<16|x)
    float *cand_resp = nullptr;    // NLEV x STAGE1_CAP Harris responses
    int32_t *kp_cnt = nullptr;     // NLEV kept counts
    uint32_t *kp_key = nullptr;    // NLEV x STAGE1_CAP kept keypoints (raster order per level)
    float *kp_resp = nullptr;
    // frame feature outputs (max_feat rows)
    float *f_xy = nullptr, *f_size = nullptr, *f_angle = nullptr, *f_resp = nullptr;
    int32_t *f_oct = nullptr;
    uint8_t *f_desc = nullptr;
    int32_t *f_count = nullptr;
    uint8_t *frame_img = nullptr;  // staging for host frames (max_w*max_h*3)
    void *orb_const = nullptr;     // device copy of the OrbTable (reloc_orb.hip)
    char orb_tab_host[1024];       // host copy of the same table
    int32_t *dbg_cut = nullptr;    // NLEV stage-1 cut scores of the last frame

    // ---- camera (reference M:49-52, M:107-112) ----
    double K4[4] = {RELOC_FX, RELOC_FY, RELOC_CX, RELOC_CY};
    double b2c_t[3] = {0.35, 0.0, 0.18};
    double b2c_R[9] = {0, -1, 0, 0, 0, -1, 1, 0, 0};

    // ---- matcher parameters (reloc_set_params) ----
    reloc_params prm;
    int scan_grid = 0;               // RELOC_SCAN_GRID (developer switch), read once at creation: > 0 static grid of that
                                     // many workgroups, < 0 static default grid, 0 ticket scheduling
    int scan_gens = 0;               // RELOC_SCAN_GENS (developer switch): generations of the ticket grid, < 0 = one, no quota
    int scan_batch_gens = 0;         // RELOC_SCAN_BATCH_GENS (developer switch): generations of a batched scan launch
    int scan_quota_rows = 1;         // RELOC_SCAN_QUOTA_ROWS (developer switch): 1 = workgroup budgets in rows + sweepers, 0 = a quota of records (rounds 2-3a)
    int scan_nw = 0;                 // RELOC_SCAN_NW (developer switch): waves per record of the whole-database scan (1, 2, 4); 0 = by shape
    uint32_t *scan_ticket = nullptr; // per frame of a batch (<= 8) 8 per-XCD record counters, then 1 exit counter, 128 bytes apart

    // ---- database: two arenas, the fields below are the SELECTED one's (reloc_db_select copies them) ----
    DbArena db_slot[2];
    int db_sel = 0;
    bool db_shared = false;          // the selected database's arrays were adopted from another ctx (reloc_db_share): read-only here
    DbShare *db_share = nullptr;     // reference count of the selected arena's six shared arrays
    int64_t db_records = 0, db_rows = 0;
    int64_t db_cap_records = 0, db_cap_rows = 0;
    int db_max_rows = 0;
    uint8_t *db_desc = nullptr;
    float *db_pts3d = nullptr;
    float *db_kp2d = nullptr;
    int64_t *db_off = nullptr;
    double *db_pose = nullptr;
    double *db_xy_heading = nullptr; // L x 4 (x, y, cos heading, sin heading) for candidate selection
    int32_t *db_counts = nullptr;    // L per-record mutual counts
    unsigned long long *topk_part = nullptr;   // per-block winners of the two-stage top-k (topk_blocks x 32)
    int topk_blocks = 0;
    AccumResult *accum_res = nullptr;   // 1
    int32_t *tick_flags = nullptr;      // [0] relocating flag of the current tick

    // ---- tick state ----
    int32_t *cand_ids = nullptr;     // MAX_CAND
    int32_t *cand_n = nullptr;       // 1
    int32_t *m_qidx = nullptr, *m_tidx = nullptr, *m_dist = nullptr, *m_n = nullptr; // MAX_CAND x MAX_REC_ROWS
    float *p_obj = nullptr, *p_img = nullptr;     // MAX_CAND x MAX_REC_ROWS x {3,2}
    double *p_Rt = nullptr;          // MAX_CAND x MAX_HYP x 12
    int32_t *p_cnt = nullptr;        // MAX_CAND x MAX_HYP
    int32_t *p_inl = nullptr;        // MAX_CAND x MAX_REC_ROWS
    PnpOut *p_out = nullptr;         // MAX_CAND
    int exclusive_hint = -1;         // reloc_set_exclusive: 1 = this ctx is the only stream of work on the GPU, 0 = it is not,
                                     // -1 (default) = it is while it is the only live context of this process (ctx_alone())
    bool orb_latency_shape = true;   // k_pyramid with 512-thread workgroups; cleared by ticks that share the chip with scans
    bool latency_shapes = false;     // set around the emit pass / PnP of a tick that runs NO whole-database scan (and by the
                                     // single-call entry points): kernels sized for latency instead of for fitting beside a scan
    bool local_two_stage = false;    // developer switch RELOC_LOCAL_TWO_STAGE=1: local candidates by k_topk_part + k_candidates_local
    TickResult *tick_res = nullptr;  // 1
    TickResult *tick_res_host = nullptr;   // the same record in pinned host memory, written by k_tick_finalize
    bool tick_failed = false;              // the last tick entry point on this context returned an error before its result record was
                                           // enqueued: reloc_tick_wait / reloc_tick_result* report RELOC_E_STATE instead of the previous tick's record
    int32_t tick_seq = 0;                  // stamp of the last tick enqueued (TickResult.pad[0] of its host records); 0: none yet
    TickResult *tick_res_ext = nullptr;    // caller's pinned record for the next ticks (reloc_tick_result_to), or NULL
};

int reloc_scratch(reloc_ctx *ctx, int slot, int64_t bytes, void **out);
void reloc_prof_begin(reloc_ctx *ctx, int which);
void reloc_prof_end(reloc_ctx *ctx, int which);

// launchers shared between translation units
// rec_ids == NULL: scan records 0..n_ids_max-1 and write counts[record]; otherwise scan the listed
// records (count read from n_ids_dev when non-NULL) and write slot-indexed outputs.
// lets go of one reference to an arena's six arrays; frees them when it was the last (reloc_match.hip)
void db_arrays_drop(DbShare *&share, uint8_t *&desc, float *&pts3d, float *&kp2d, int64_t *&off, double *&pose, double *&xyh);
extern int g_reloc_live_contexts;           // contexts created and not yet destroyed in this process (reloc_ctx.hip)
static inline bool ctx_alone(const reloc_ctx *c)
{
    return c->exclusive_hint > 0 || (c->exclusive_hint < 0 && __atomic_load_n(&g_reloc_live_contexts, __ATOMIC_RELAXED) == 1);
}

int launch_db_scan(reloc_ctx *ctx, const uint8_t *db_desc, const int64_t *db_off, int64_t n_rec,
                   const int32_t *rec_ids, const int32_t *n_ids_dev, int n_ids_max, const uint8_t *cur,
                   const int32_t *n_cur_dev, int n_cur_max, int max_rows, int32_t *counts, int32_t *m_qidx,
                   int32_t *m_tidx, int32_t *m_dist, int32_t *m_n, int emit_stride, const ScanMask *mask = nullptr);
int launch_db_emit_batch(reloc_ctx *const *ctxs, int n);
int launch_db_scan_batch(reloc_ctx *const *ctxs, int n, const double *q, double cos_tol, bool auto_mode, bool heading_mask = true);
int db_reindex(reloc_ctx *ctx);
int db_reserve(reloc_ctx *ctx, int64_t cap_records, int64_t cap_rows);
inline bool db_ready(const reloc_ctx *ctx) { return ctx->db_desc && ctx->db_off && ctx->db_pose && ctx->db_xy_heading && ctx->db_counts && ctx->db_records > 0; }
int orb_prepare(reloc_ctx *ctx, int w, int h, int nfeatures);
constexpr int RELOC_BATCH_MAX = 8;          // frames per batched launch (reloc_tick_batch_dev, reloc_shard_*_batch_dev)
int orb_run_batch_dev(reloc_ctx *const *ctxs, int n, const uint8_t *const *srcs_dev, int w, int h, int stride, int order, int nfeatures);
int orb_run_dev(reloc_ctx *ctx, const uint8_t *src_dev, int w, int h, int stride, int channels, int order,
                int nfeatures);
int pnp_run_candidates_batch(reloc_ctx *const *ctxs, int n, int n_cand_max, const uint64_t *seeds);
int pnp_run_candidates(reloc_ctx *ctx, int n_cand_max, const int32_t *n_cand_dev, const double K4[4], int iters,
                       float thr_px, double conf, uint64_t seed, int min_m, const int32_t *relocating_dev, int gate_local, int gate_global);
