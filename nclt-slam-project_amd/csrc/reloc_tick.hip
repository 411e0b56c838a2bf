// reloc_tick.hip -- the fused repeat tick on gfx950: every step between "frame in HBM" and
// "anchor pose" runs on the device, the host only enqueues kernels.
//
// Mirrors VisualLandmarkMatcher._tick (reference M:281-433) and the whole-database candidate search
// of the global-relocalisation variant (reference G:315-344, gates G:381-382, no consistency gate
// G:424):
//   ORB (reloc_orb.hip) -> candidates (local: nearest 15 by VIO distance, radius 8 m, heading 90 deg,
//   first 5;  global: mutual-match count of every heading-compatible record, top 25) ->
//   mutual matches of each candidate (reloc_match.hip, emit mode) -> gather 3-D/2-D pairs ->
//   PnP-RANSAC batch (reloc_pnp.hip) -> gates, pose composition, best by inliers, consistency.
#include <time.h>

#include "reloc_internal.h"

struct TickParams {
    double base_pose[7];
    double b2c_t[3];
    double b2c_R[9];
    int mode;                 // RELOC_TICK_LOCAL / GLOBAL / AUTO; which one produced the candidates is the device flag
    int check_consistency;    // 1 / 0, or -1: only when the candidates are local (M:391, G:424)
    int n_records;
    // matcher parameters (reloc_params)
    int max_candidates, min_matches, min_inliers, global_min_inliers;
    double radius_m, cos_tol, reproj_max_px, global_reproj_max_px, consistency_m;
    int seq;                  // sequence stamp of this tick (ctx->tick_seq), stored into the host records after their body
};

__device__ void rot_to_quat(const double R[9], double q[4])
{
    const double tr = R[0] + R[4] + R[8];
    double qx, qy, qz, qw;
    if (tr > 0) {
        const double s = 0.5 / sqrt(tr + 1.0);
        qw = 0.25 / s;
        qx = (R[7] - R[5]) * s; qy = (R[2] - R[6]) * s; qz = (R[3] - R[1]) * s;
    } else if (R[0] > R[4] && R[0] > R[8]) {
        const double s = 2.0 * sqrt(1.0 + R[0] - R[4] - R[8]);
        qw = (R[7] - R[5]) / s; qx = 0.25 * s; qy = (R[1] + R[3]) / s; qz = (R[2] + R[6]) / s;
    } else if (R[4] > R[8]) {
        const double s = 2.0 * sqrt(1.0 + R[4] - R[0] - R[8]);
        qw = (R[2] - R[6]) / s; qx = (R[1] + R[3]) / s; qy = 0.25 * s; qz = (R[5] + R[7]) / s;
    } else {
        const double s = 2.0 * sqrt(1.0 + R[8] - R[0] - R[4]);
        qw = (R[3] - R[1]) / s; qx = (R[2] + R[6]) / s; qy = (R[5] + R[7]) / s; qz = 0.25 * s;
    }
    q[0] = qx; q[1] = qy; q[2] = qz; q[3] = qw;
}

__device__ __forceinline__ void cur_heading(const TickParams &prm, double &cc, double &sc)
{
    cur_heading_q(prm.base_pose + 3, cc, sc);
}

// Block-wide "k largest keys, descending" (keys unique, 0 = not eligible) without barriers in the
// selection loops: every wave first extracts the k largest keys of ITS records (record i belongs to
// thread (i - begin) mod TICK_BLOCK) -- per round one wave max-reduction (DPP), and only the lane that
// owned the maximum moves on to its next key -- then wave 0 merges the per-wave winners the
// same way.  One __syncthreads in total.  s_part: TICK_WAVES * TOPK_MAX entries.
// The single-block tick kernels use 256 threads (one wave per SIMD): a block of that shape fits into the
// slot one retiring k_db_scan workgroup frees, so these kernels of one stream start while another
// stream's scan still fills the chip; a 1024-thread block would wait for a whole CU to drain.
constexpr int TOPK_MAX = 32;
constexpr int TICK_BLOCK = 256, TICK_WAVES = TICK_BLOCK / 64;
static_assert(MAX_CAND <= TOPK_MAX, "candidate slots");

// The ranking loops below are chains of ~50 wave reductions, so the latency of one reduction is the
// kernel's run time: wave_max_u32 (DPP) instead of ds_bpermute butterflies took a ranking kernel from 23 to ~6 us.
// keys are unique: the maximum is the largest low word among the lanes that hold the largest high word
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v)
{
    const unsigned hi = wave_max_u32((unsigned)(v >> 32));
    const unsigned lo = wave_max_u32((unsigned)(v >> 32) == hi ? (unsigned)v : 0u);
    return ((unsigned long long)hi << 32) | lo;
}

template <typename KeyFn>
__device__ int block_topk(int begin, int end, int k, KeyFn keyfn, unsigned long long *s_part, unsigned long long *out_keys)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // final pick, by wave 0: a lane holds up to four keys sorted in registers; taking a winner is a register move
    unsigned long long a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    auto hold = [&](unsigned long long key) {
        unsigned long long t;
        if (key > a0) { t = a0; a0 = key; key = t; }
        if (key > a1) { t = a1; a1 = key; key = t; }
        if (key > a2) { t = a2; a2 = key; key = t; }
        if (key > a3) a3 = key;
    };
    auto pick = [&]() {
        int n = 0;
        for (int r = 0; r < k; ++r) {
            const unsigned long long top = wave_max_u64(a0);
            if (top == 0) break;
            if (lane == 0) out_keys[n] = top;
            ++n;
            if (a0 == top) { a0 = a1; a1 = a2; a2 = a3; a3 = 0; }
        }
        if (lane == 0) out_keys[TOPK_MAX] = (unsigned long long)n;
    };
    if (end - begin <= 256) {
        // few enough keys for one wave (the merge of the per-block lists of a 10k-record database is 250 keys)
        if (wave == 0) {
            for (int i = begin + lane; i < end; i += 64) hold(keyfn(i));
            pick();
        }
        __syncthreads();
        return (int)out_keys[TOPK_MAX];
    }
    // every thread keeps the 4 largest of its own keys sorted in registers and rescans its records only
    // when all four have been taken (a thread rarely owns more than a few of the block's top k)
    unsigned long long m0, m1, m2, m3;
    bool more;
    auto refill = [&](unsigned long long bound) {
        m0 = m1 = m2 = m3 = 0;
        for (int i = begin + tid; i < end; i += TICK_BLOCK) {
            unsigned long long key = keyfn(i);
            if (key >= bound) continue;
            unsigned long long t;
            if (key > m0) { t = m0; m0 = key; key = t; }
            if (key > m1) { t = m1; m1 = key; key = t; }
            if (key > m2) { t = m2; m2 = key; key = t; }
            if (key > m3) m3 = key;
        }
        more = m3 != 0;
    };
    refill(~0ull);
    for (int r = 0; r < k; ++r) {
        const unsigned long long top = wave_max_u64(m0);
        if (lane == 0) s_part[wave * TOPK_MAX + r] = top;
        if (top != 0 && m0 == top) {                                  // keys are unique: exactly one owner
            m0 = m1; m1 = m2; m2 = m3; m3 = 0;
            if (m0 == 0 && more) refill(top);
        }
    }
    __syncthreads();
    if (wave == 0) {
        // lane l owns entries l and l + 64 of the TICK_WAVES * k <= 128 per-wave winners
        static_assert(TICK_WAVES * TOPK_MAX <= 256, "four entries per lane");
        for (int e = lane; e < TICK_WAVES * k; e += 64) hold(s_part[(e / k) * TOPK_MAX + (e % k)]);
        pick();
    }
    __syncthreads();
    return (int)out_keys[TOPK_MAX];
}

// ---- candidate selection ---------------------------------------------------------------------------
// Two stages, so that a 10k-record database is not walked by one workgroup: k_topk_part gives every
// TOPK_SLICE records to one block, which leaves its k best keys in topk_part; the mode's final kernel
// (k_candidates_local / k_topk_counts) merges the blocks' lists (or, for a database of one slice, ranks the records directly) and applies the mode's
// epilogue.  Keys are unique and 0 means "not eligible".
//   local mode (M:293-302): nearest 15 by (distance, index), then the radius / heading filter in that
//     order, first 5 kept.  Order key: the bit pattern of a non-negative double is monotone in its value;
//     its top 44 bits order the candidates (ties at that resolution, < 1e-9 relative, fall back to the
//     lower index), the radius test itself uses the exact distance.
//   global mode (G:329-344): top-k of (count, id) descending among the records with count >= MIN_MATCHES
//     (the scan leaves count 0 on heading-incompatible records, see ScanMask): k_topk_counts, a histogram selection.
constexpr int TOPK_SLICE = 1024;

struct LocalKey {
    const double *xyh;
    double vx, vy;
    __device__ unsigned long long operator()(int i) const
    {
        const double dx = xyh[4 * i] - vx, dy = xyh[4 * i + 1] - vy;
        const double d = sqrt(dx * dx + dy * dy);
        const unsigned long long q = (unsigned long long)__double_as_longlong(d) >> 20;
        return ((0xFFFFFFFFFFFull - q) << 20) | (unsigned long long)(0xFFFFF - (i & 0xFFFFF));   // nearer, then lower index
    }
};

struct PartKey {
    const unsigned long long *part;
    int k;
    __device__ unsigned long long operator()(int i) const { return part[(i / k) * TOPK_MAX + (i % k)]; }
};

template <typename KeyFn>
__global__ __launch_bounds__(TICK_BLOCK) void k_topk_part(KeyFn keyfn, int L, int k, unsigned long long *__restrict__ part)
{
    RELOC_SMALL_KERNEL_PRIO();
    __shared__ unsigned long long s_red[TICK_WAVES * TOPK_MAX];
    __shared__ unsigned long long s_keys[TOPK_MAX + 1];
    const int begin = blockIdx.x * TOPK_SLICE, end = min(L, begin + TOPK_SLICE);
    const int n = block_topk(begin, end, k, keyfn, s_red, s_keys);
    __syncthreads();
    if ((int)threadIdx.x < k) part[blockIdx.x * TOPK_MAX + threadIdx.x] = (int)threadIdx.x < n ? s_keys[threadIdx.x] : 0ull;
}

// n_blocks == 0: rank the L records directly; otherwise merge n_blocks lists of k keys
template <typename KeyFn>
__device__ int topk_final(KeyFn keyfn, int L, int k, const unsigned long long *part, int n_blocks, unsigned long long *s_red,
                          unsigned long long *s_keys)
{
    if (n_blocks == 0) return block_topk(0, L, k, keyfn, s_red, s_keys);
    return block_topk(0, n_blocks * k, k, PartKey{part, k}, s_red, s_keys);
}

__global__ __launch_bounds__(TICK_BLOCK) void k_candidates_local(const double *__restrict__ xyh, TickParams prm,
                                                                 const unsigned long long *__restrict__ part, int n_blocks,
                                                                 int32_t *__restrict__ cand_ids, int32_t *__restrict__ cand_n,
                                                                 int32_t *__restrict__ relocating)
{
    RELOC_SMALL_KERNEL_PRIO();
    __shared__ unsigned long long s_red[TICK_WAVES * TOPK_MAX];
    __shared__ unsigned long long s_keys[TOPK_MAX + 1];
    const int L = prm.n_records;
    const double vx = prm.base_pose[0], vy = prm.base_pose[1];
    const int n = topk_final(LocalKey{xyh, vx, vy}, L, min(prm.max_candidates * 3, L), part, n_blocks, s_red, s_keys);
    __syncthreads();
    if (threadIdx.x == 0) {
        double cc, sc;
        cur_heading(prm, cc, sc);
        int m = 0;
        for (int r = 0; r < n && m < prm.max_candidates; ++r) {
            const int i = 0xFFFFF - (int)(s_keys[r] & 0xFFFFF);
            const double dx = xyh[4 * i] - vx, dy = xyh[4 * i + 1] - vy;
            if (sqrt(dx * dx + dy * dy) < prm.radius_m && heading_ok(xyh + 4 * i, cc, sc, prm.cos_tol)) cand_ids[m++] = i;
        }
        *cand_n = m;
        *relocating = 0;
    }
}

// Candidate ranking of the whole-database search (G:342-343: `scored.sort(reverse=True)[:25]` on (count, id) tuples):
// the k largest (count, id) among the records with count >= min_matches, in descending order.
// Counts take few distinct values (0 .. rows of the largest record) and cluster -- cross-checked matching of a 64-row
// record against 500 descriptors leaves ~57 +- 3 mutual pairs even on unrelated data, so nearly every record passes
// MIN_MATCHES and the winners are separated by ties -- which makes this a histogram selection, not a sort: one block
//   1. histograms the counts in LDS,
//   2. walks the bins from the top to the count c* that holds the k-th winner (records above c* all win),
//   3. takes the records above c* and, among those AT c*, the ones with the largest ids: every thread owns a contiguous
//      id range, a block-wide suffix sum of the per-thread tie counts tells it how many ties rank before its own,
//   4. orders the <= k winners with one wave.
// Two passes over L counts and no dependent chain of k reductions per slice (r1: 10 slice blocks x 50 chained wave
// reductions + a merge kernel = 23 us at L = 10 000; measured r2 in profiles/).
// skip_if: RELOC_TICK_AUTO -- the local search found candidates (*skip_if != 0): they stand, nothing is ranked (the scan
// before this kernel has skipped itself the same way).  relocating: set when this ranking produced the list.
constexpr int TOPK_HIST_THREADS = 1024;
__device__ __forceinline__ void topk_counts_body(const int32_t *__restrict__ counts, int L, int k, int id_base,
                                                 int min_matches, int max_count, int32_t *__restrict__ out_ids,
                                                 int32_t *__restrict__ out_counts, int32_t *out_n,
                                                 const int32_t *skip_if, int32_t *__restrict__ relocating,
                                                 const int32_t *__restrict__ f_count, int32_t *__restrict__ out_nfeat){
    extern __shared__ int s_hist[];                     // max_count + 2 bins
    if (out_nfeat && threadIdx.x == 0) *out_nfeat = *f_count;          // sharded scan: the frame's feature count rides along
    __shared__ int s_wave[16];
    __shared__ int s_cstar, s_above, s_nlist;
    __shared__ unsigned long long s_list[TOPK_MAX];
    if (skip_if && *skip_if != 0) return;               // block-uniform; out_n may BE skip_if (AUTO mode): neither is __restrict__
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i <= max_count + 1; i += TOPK_HIST_THREADS) s_hist[i] = 0;
    if (tid == 0) s_nlist = 0;
    __syncthreads();
    const int chunk = (L + TOPK_HIST_THREADS - 1) / TOPK_HIST_THREADS;
    const int lo = min(tid * chunk, L), hi = min(lo + chunk, L);          // this thread's ids, ascending
    for (int i = lo; i < hi; ++i) {
        const int c = min(counts[i], max_count + 1);
        if (c >= min_matches) atomicAdd(&s_hist[c], 1);
    }
    __syncthreads();
    if (wave == 0) {
        // bins from the top, 64 at a time: first bin (descending) at which the running total reaches k
        int above = 0, cstar = -1;
        for (int top = max_count + 1; top >= min_matches && cstar < 0; top -= 64) {
            const int bin = top - lane;                                    // lane 0 = highest bin of this group
            const int h = bin >= min_matches ? s_hist[bin] : 0;
            int incl = h;                                                  // inclusive prefix over lanes 0..lane
            for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
            const unsigned long long hit = __ballot(above + incl >= k && bin >= min_matches);
            if (hit) {
                const int l0 = __ffsll((long long)hit) - 1;
                cstar = top - l0;
                above += __shfl(incl, l0) - __shfl(h, l0);
            } else {
                above += __shfl(incl, 63);
            }
        }
        if (lane == 0) { s_cstar = cstar; s_above = above; }              // cstar < 0: fewer than k eligible records, all win
    }
    __syncthreads();
    const int cstar = s_cstar, need = cstar < 0 ? 0 : k - s_above;         // ties at c* still wanted
    int ties = 0;
    for (int i = lo; i < hi; ++i) {
        const int c = min(counts[i], max_count + 1);
        if (c < min_matches) continue;
        if (c > cstar || cstar < 0) {
            const int p = atomicAdd(&s_nlist, 1);
            if (p < TOPK_MAX) s_list[p] = ((unsigned long long)(unsigned)c << 32) | (unsigned)(i + 1);
        } else if (c == cstar) {
            ++ties;
        }
    }
    // ties ranking before this thread's = ties owned by threads with larger ids
    int incl = ties;
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_down(incl, d); if (lane + d < 64) incl += o; }   // suffix within the wave
    if (lane == 0) s_wave[wave] = incl;
    __syncthreads();
    int after = incl - ties;
    for (int w = wave + 1; w < TOPK_HIST_THREADS / 64; ++w) after += s_wave[w];
    if (ties && after < need) {
        int take = min(ties, need - after);
        for (int i = hi - 1; i >= lo && take > 0; --i) {
            if (min(counts[i], max_count + 1) == cstar) {
                const int p = atomicAdd(&s_nlist, 1);
                if (p < TOPK_MAX) s_list[p] = ((unsigned long long)(unsigned)cstar << 32) | (unsigned)(i + 1);
                --take;
            }
        }
    }
    __syncthreads();
    const int n = min(s_nlist, k);
    if (wave == 0) {
        // descending order by rank counting: keys are unique
        const unsigned long long mine = lane < n ? s_list[lane] : 0ull;
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += s_list[j] > mine;
        if (lane < n) {
            out_ids[rank] = (int32_t)((unsigned)(mine & 0xFFFFFFFFu) - 1) + id_base;
            if (out_counts) out_counts[rank] = (int32_t)(mine >> 32);
        }
        if (lane >= n && lane < k) {
            out_ids[lane] = -1;
            if (out_counts) out_counts[lane] = 0;
        }
        if (lane == 0) {
            *out_n = n;
            if (relocating) *relocating = 1;
        }
    }
}
__global__ __launch_bounds__(TOPK_HIST_THREADS) void k_topk_counts(const int32_t *__restrict__ counts, int L, int k, int id_base,
                                                                   int min_matches, int max_count, int32_t *__restrict__ out_ids,
                                                                   int32_t *__restrict__ out_counts, int32_t *out_n,
                                                                   const int32_t *skip_if, int32_t *__restrict__ relocating,
                                                                   const int32_t *__restrict__ f_count = nullptr,
                                                                   int32_t *__restrict__ out_nfeat = nullptr)
{
    RELOC_SMALL_KERNEL_PRIO();
    topk_counts_body(counts, L, k, id_base, min_matches, max_count, out_ids, out_counts, out_n, skip_if, relocating, f_count, out_nfeat);
}
// the rankings of up to 8 frames in one launch: block = frame
struct TopkFrame {
    const int32_t *counts; int32_t *out_ids, *out_counts, *out_n; const int32_t *skip_if; int32_t *relocating; const int32_t *f_count;
    int32_t *out_nfeat;
};
struct TopkBatch { TopkFrame f[RELOC_BATCH_MAX]; };
__global__ __launch_bounds__(TOPK_HIST_THREADS) void k_topk_counts_batch(TopkBatch b, int L, int k, int id_base, int min_matches, int max_count)
{
    RELOC_SMALL_KERNEL_PRIO();
    const TopkFrame &F = b.f[blockIdx.x];
    topk_counts_body(F.counts, L, k, id_base, min_matches, max_count, F.out_ids, F.out_counts, F.out_n, F.skip_if, F.relocating, F.f_count,
                     F.out_nfeat);
}


// Local candidates in ONE launch (databases up to NEAR_MAX_RECORDS).  Only a record within radius_m can become a
// candidate, and the nearest-15 cut (M:296) is taken in LocalKey order, which is monotone in the distance at 44-bit
// resolution: so the nearest 15 of ALL records, as far as they can pass the radius test, are the nearest 15 of
//   E' = { i : trunc44(d_i) <= trunc44(radius_m) }      (a superset of the records inside the radius),
// and everything behind E' in that order fails the radius test anyway.  The block collects E' in LDS (a handful of
// records on a teach path), orders it by rank counting and applies the radius / heading filter in that order exactly as
// k_candidates_local does.  More than NEAR_CAP records in E': the block ranks all L records itself (block_topk), slower
// but the same answer.  Replaces k_topk_part + k_candidates_local (9.5 + 7.7 us at 10 000 records) by one ~6 us kernel.
constexpr int NEAR_CAP = 256, NEAR_MAX_RECORDS = 1 << 17;
__global__ __launch_bounds__(TICK_BLOCK) void k_candidates_near(const double *__restrict__ xyh, TickParams prm,
                                                                int32_t *__restrict__ cand_ids, int32_t *__restrict__ cand_n,
                                                                int32_t *__restrict__ relocating)
{
    RELOC_SMALL_KERNEL_PRIO();
    __shared__ unsigned long long s_red[TICK_WAVES * TOPK_MAX];
    __shared__ unsigned long long s_keys[TOPK_MAX + 1];
    __shared__ unsigned long long s_near[NEAR_CAP];
    __shared__ int s_cnt;
    const int L = prm.n_records, tid = threadIdx.x;
    const double vx = prm.base_pose[0], vy = prm.base_pose[1];
    const LocalKey keyfn{xyh, vx, vy};
    const int k = min(prm.max_candidates * 3, L);
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    // LocalKey's distance field is 0xFFFFFFFFFFF - trunc44(d): "trunc44(d) <= trunc44(r)" is "key field >= that of r"
    const unsigned long long rfield = 0xFFFFFFFFFFFull - ((unsigned long long)__double_as_longlong(prm.radius_m) >> 20);
    // 8 records per thread and turn, their loads issued together: the walk is a chain of memory round trips otherwise
    constexpr int NEAR_UNROLL = 8;
    for (int i0 = tid; i0 < L; i0 += NEAR_UNROLL * TICK_BLOCK) {
        double rx[NEAR_UNROLL], ry[NEAR_UNROLL];
#pragma unroll
        for (int u = 0; u < NEAR_UNROLL; ++u) {
            const int i = min(i0 + u * TICK_BLOCK, L - 1);
            rx[u] = xyh[4 * i];
            ry[u] = xyh[4 * i + 1];
        }
#pragma unroll
        for (int u = 0; u < NEAR_UNROLL; ++u) {
            const int i = i0 + u * TICK_BLOCK;
            if (i >= L) break;
            // = LocalKey(i), from the values already loaded
            const double dx = rx[u] - vx, dy = ry[u] - vy;
            const double d = sqrt(dx * dx + dy * dy);
            const unsigned long long q = (unsigned long long)__double_as_longlong(d) >> 20;
            const unsigned long long key = ((0xFFFFFFFFFFFull - q) << 20) | (unsigned long long)(0xFFFFF - (i & 0xFFFFF));
            if ((key >> 20) >= rfield) {
                const int at = atomicAdd(&s_cnt, 1);
                if (at < NEAR_CAP) s_near[at] = key;
            }
        }
    }
    __syncthreads();
    const int ne = s_cnt;
    int n;
    if (ne <= NEAR_CAP) {
        // keys are unique: the rank of a key is the number of larger ones
        if (tid < ne) {
            const unsigned long long mine = s_near[tid];
            int rank = 0;
            for (int j = 0; j < ne; ++j) rank += s_near[j] > mine;
            if (rank < k) s_keys[rank] = mine;
        }
        n = min(ne, k);
        __syncthreads();
    } else {
        n = block_topk(0, L, k, keyfn, s_red, s_keys);
        __syncthreads();
    }
    if (tid == 0) {
        double cc, sc;
        cur_heading(prm, cc, sc);
        int m = 0;
        for (int r = 0; r < n && m < prm.max_candidates; ++r) {
            const int i = 0xFFFFF - (int)(s_keys[r] & 0xFFFFF);
            const double dx = xyh[4 * i] - vx, dy = xyh[4 * i + 1] - vy;
            if (sqrt(dx * dx + dy * dy) < prm.radius_m && heading_ok(xyh + 4 * i, cc, sc, prm.cos_tol)) cand_ids[m++] = i;
        }
        *cand_n = m;
        *relocating = 0;
    }
}

// host side: one launch (k_candidates_near), or the two stages for very large databases / RELOC_LOCAL_TWO_STAGE=1
static void launch_candidates_local(reloc_ctx *ctx, const TickParams &prm)
{
    if (prm.n_records <= NEAR_MAX_RECORDS && !ctx->local_two_stage) {
        hipLaunchKernelGGL(k_candidates_near, dim3(1), dim3(TICK_BLOCK), 0, ctx->stream, ctx->db_xy_heading, prm, ctx->cand_ids,
                           ctx->cand_n, ctx->tick_flags);
        return;
    }
    const int L = prm.n_records, k = L < prm.max_candidates * 3 ? L : prm.max_candidates * 3;
    const int nb = L > TOPK_SLICE ? (L + TOPK_SLICE - 1) / TOPK_SLICE : 0;
    if (nb)
        hipLaunchKernelGGL(k_topk_part<LocalKey>, dim3(nb), dim3(TICK_BLOCK), 0, ctx->stream,
                           LocalKey{ctx->db_xy_heading, prm.base_pose[0], prm.base_pose[1]}, L, k, ctx->topk_part);
    hipLaunchKernelGGL(k_candidates_local, dim3(1), dim3(TICK_BLOCK), 0, ctx->stream, ctx->db_xy_heading, prm, ctx->topk_part, nb,
                       ctx->cand_ids, ctx->cand_n, ctx->tick_flags);
}

// auto_mode: the ranking only takes effect when the local search left no candidate (see k_topk_counts)
static void launch_topk_counts(reloc_ctx *ctx, int k, int32_t *out_ids, int32_t *out_counts, bool auto_mode, int id_base = 0,
                               int32_t *out_nfeat = nullptr)
{
    const int L = (int)ctx->db_records;
    // a count cannot exceed the rows of the largest record nor the features of a frame
    const int max_count = ctx->db_max_rows < ctx->max_feat ? ctx->db_max_rows : ctx->max_feat;
    hipLaunchKernelGGL(k_topk_counts, dim3(1), dim3(TOPK_HIST_THREADS), (size_t)(max_count + 2) * sizeof(int), ctx->stream, ctx->db_counts,
                       L, k, id_base, ctx->prm.min_matches, max_count, out_ids, out_counts, ctx->cand_n,
                       auto_mode ? (const int32_t *)ctx->cand_n : (const int32_t *)nullptr, ctx->tick_flags,
                       (const int32_t *)ctx->f_count, out_nfeat);
}

__device__ __forceinline__ void tick_stamp(TickResult *res_host, TickResult *res_ext, int seq)
{
    if (!seq) return;
    __hip_atomic_store(&res_host->pad[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (res_ext) __hip_atomic_store(&res_ext->pad[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- gates, pose composition, best candidate (M:349-410; G:381-382,424) ---------------------------
// one lane per candidate composes its pose; the best (most inliers, first on ties) is picked by a wave
// reduction.
__device__ __forceinline__ void tick_finalize_body(const int32_t *__restrict__ cand_ids, const int32_t *__restrict__ cand_n,
                                                   const PnpOut *__restrict__ pnp, const double *__restrict__ db_pose,
                                                   const int32_t *__restrict__ f_count, const TickParams &prm,
                                                   const int32_t *__restrict__ relocating_p, TickResult *__restrict__ res,
                                                   TickResult *__restrict__ res_host, TickResult *__restrict__ res_ext, int seq){
    // seq: after the record's 96 bytes, its last-but-one word (pad[0]) receives the tick's sequence stamp with a system-scope
    // release store: a host that polls that word (reloc_tick_wait) has the complete record the moment it sees the stamp,
    // without going through the runtime's stream-completion path.
    // res: the device record (read by the accumulation and by device-side consumers); res_host: the ctx's own record in
    // pinned host memory, what reloc_tick_result() reads after the stream has drained -- the kernel writes it over PCIe
    // itself, which takes a 5 us copy kernel (and its launch) out of every synchronous tick; res_ext: a caller-named pinned
    // record (reloc_tick_result_to), the streaming form of the same.
    const int s = threadIdx.x;
    const int nc = min(*cand_n, MAX_CAND);
    const int nfeat = *f_count;
    const int relocating = *relocating_p;
    const int min_inl = relocating ? prm.global_min_inliers : prm.min_inliers;          // G:381
    const double max_err = relocating ? prm.global_reproj_max_px : prm.reproj_max_px;   // G:382
    const bool check = prm.check_consistency < 0 ? !relocating : prm.check_consistency != 0;
    bool okc = false;
    int inl = 0;
    double err = 0, pose[7] = {0, 0, 0, 0, 0, 0, 0};
    if (s < nc && nfeat >= prm.min_matches) {
        const PnpOut &p = pnp[s];
        if (p.ok && p.n_inl >= min_inl && !(p.reproj_mean > max_err)) {
            okc = true;
            inl = p.n_inl;
            err = p.reproj_mean;
            const double *R = p.Rt, *t = p.Rt + 9;
            double Rtc[9], ttc[3];                       // current camera in the teach camera frame
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) Rtc[3 * r + c] = R[3 * c + r];
            for (int r = 0; r < 3; ++r) ttc[r] = -(Rtc[3 * r] * t[0] + Rtc[3 * r + 1] * t[1] + Rtc[3 * r + 2] * t[2]);
            const double *tp = db_pose + 7 * (size_t)cand_ids[s];
            double Rwt[9], Rwc[9], twc[3];
            quat_to_rot(tp[3], tp[4], tp[5], tp[6], Rwt);
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c)
                    Rwc[3 * r + c] = Rwt[3 * r] * Rtc[c] + Rwt[3 * r + 1] * Rtc[3 + c] + Rwt[3 * r + 2] * Rtc[6 + c];
                twc[r] = tp[r] + (Rwt[3 * r] * ttc[0] + Rwt[3 * r + 1] * ttc[1] + Rwt[3 * r + 2] * ttc[2]);
            }
            double q[4], Rq[9], Rwb[9], qb[4];
            rot_to_quat(Rwc, q);
            // camera world pose -> base_link world pose (M:160-172), through the quaternion as the reference does
            quat_to_rot(q[0], q[1], q[2], q[3], Rq);
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c)
                    Rwb[3 * r + c] = Rq[3 * r] * prm.b2c_R[3 * c] + Rq[3 * r + 1] * prm.b2c_R[3 * c + 1] + Rq[3 * r + 2] * prm.b2c_R[3 * c + 2];
            for (int r = 0; r < 3; ++r)
                pose[r] = twc[r] - (Rwb[3 * r] * prm.b2c_t[0] + Rwb[3 * r + 1] * prm.b2c_t[1] + Rwb[3 * r + 2] * prm.b2c_t[2]);
            rot_to_quat(Rwb, qb);
            pose[3] = qb[0]; pose[4] = qb[1]; pose[5] = qb[2]; pose[6] = qb[3];
        }
    }
    // best = max inliers, lowest slot on ties (`len(inliers) > best[0]`, M:379)
    const int key = okc ? (inl << 6) | (63 - s) : -1;
    int best = key;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) best = max(best, __shfl_xor(best, d));
    if (best < 0) {
        if (s == 0) {
            TickResult out;
            out.pad[0] = out.pad[1] = 0;
            for (int k = 0; k < 7; ++k) out.anchor_pose[k] = 0;
            out.reproj = 0; out.n_inl = 0; out.lm_idx = -1; out.relocating = relocating; out.n_features = nfeat; out.n_candidates = nc;
            out.outcome = nfeat < prm.min_matches ? RELOC_OUT_NO_FEATURES : (nc == 0 ? RELOC_OUT_NO_CANDIDATES : RELOC_OUT_NO_PNP_ACCEPT);
            *res = out;
            *res_host = out;
            if (res_ext) *res_ext = out;
            tick_stamp(res_host, res_ext, seq);
        }
        return;
    }
    if (key == best) {
        TickResult out;
        out.pad[0] = out.pad[1] = 0;
        for (int k = 0; k < 7; ++k) out.anchor_pose[k] = pose[k];
        out.n_inl = inl; out.reproj = err; out.lm_idx = cand_ids[s]; out.relocating = relocating; out.n_features = nfeat; out.n_candidates = nc;
        const double dx = pose[0] - prm.base_pose[0], dy = pose[1] - prm.base_pose[1];
        const double shift = sqrt(dx * dx + dy * dy);
        out.outcome = (check && shift > prm.consistency_m) ? RELOC_OUT_CONSISTENCY_FAIL : RELOC_OUT_PUBLISHED;
        *res = out;
        *res_host = out;
        if (res_ext) *res_ext = out;
        tick_stamp(res_host, res_ext, seq);
    }
}
__global__ __launch_bounds__(64) void k_tick_finalize(const int32_t *__restrict__ cand_ids, const int32_t *__restrict__ cand_n,
                                                      const PnpOut *__restrict__ pnp, const double *__restrict__ db_pose,
                                                      const int32_t *__restrict__ f_count, TickParams prm,
                                                      const int32_t *__restrict__ relocating_p, TickResult *__restrict__ res,
                                                      TickResult *__restrict__ res_host, TickResult *__restrict__ res_ext)
{
    RELOC_SMALL_KERNEL_PRIO();
    tick_finalize_body(cand_ids, cand_n, pnp, db_pose, f_count, prm, relocating_p, res, res_host, res_ext, prm.seq);
}
// the finalisation of up to 8 frames in one launch: block = frame; prm carries what the frames share, F.base_pose the rest
struct FinalFrame {
    const int32_t *cand_ids, *cand_n; const PnpOut *pnp; const int32_t *f_count, *relocating; TickResult *res, *res_host, *res_ext;
    double base_pose[7];
    int seq;
};
struct FinalBatch { FinalFrame f[RELOC_BATCH_MAX]; };
__global__ __launch_bounds__(64) void k_tick_finalize_batch(FinalBatch b, const double *__restrict__ db_pose, TickParams prm)
{
    RELOC_SMALL_KERNEL_PRIO();
    const FinalFrame &F = b.f[blockIdx.x];
    for (int k = 0; k < 7; ++k) prm.base_pose[k] = F.base_pose[k];
    tick_finalize_body(F.cand_ids, F.cand_n, F.pnp, db_pose, F.f_count, prm, F.relocating, F.res, F.res_host, F.res_ext, F.seq);
}


// ------------------------------------------------------------------------------------------------
static double heading_cos_tol_host(const reloc_ctx *ctx) { return cos(ctx->prm.heading_tol_deg * 3.14159265358979323846 / 180.0); }

static TickParams make_tick_params(reloc_ctx *ctx, const double base_pose[7], int mode, int check_consistency)
{
    TickParams p;
    for (int k = 0; k < 7; ++k) p.base_pose[k] = base_pose[k];
    for (int k = 0; k < 3; ++k) p.b2c_t[k] = ctx->b2c_t[k];
    for (int k = 0; k < 9; ++k) p.b2c_R[k] = ctx->b2c_R[k];
    p.mode = mode;
    p.check_consistency = check_consistency;
    p.n_records = (int)ctx->db_records;
    const reloc_params &q = ctx->prm;
    p.max_candidates = q.max_candidates; p.min_matches = q.min_matches; p.min_inliers = q.min_inliers;
    p.global_min_inliers = q.global_min_inliers;
    p.radius_m = q.candidate_radius_m; p.cos_tol = heading_cos_tol_host(ctx); p.reproj_max_px = q.reproj_max_px;
    p.global_reproj_max_px = q.global_reproj_max_px; p.consistency_m = q.consistency_m;
    p.seq = 0;
    return p;
}

__global__ void k_set_flag(int32_t *flag, int v) { *flag = v; }

// the stamp of the tick being enqueued: 1, 2, ... (never 0: 0 means "no stamp")
static int tick_next_seq(reloc_ctx *ctx)
{
    ctx->tick_seq = ctx->tick_seq >= 0x7fffffff ? 1 : ctx->tick_seq + 1;
    ctx->tick_failed = false;                 // called where the finalisation (which stores the record and this stamp) is launched
    return ctx->tick_seq;
}

static int tick_solve(reloc_ctx *ctx, const TickParams &prm, uint64_t seed)
{
    int rc;
    hipStream_t st = ctx->stream;
    // mutual matches of every candidate, in queryIdx order, with their 3-D / 2-D pairs (M:333-336)
    ScanMask emit;
    emit.xyh = nullptr;
    emit.q[0] = emit.q[1] = emit.q[2] = 0; emit.q[3] = 1;
    emit.g_pts3d = ctx->db_pts3d; emit.g_xy = ctx->f_xy; emit.g_obj = ctx->p_obj; emit.g_img = ctx->p_img;
    // a tick of local candidates runs no whole-database scan: its emit pass and refinement are sized for latency
    ctx->latency_shapes = prm.mode == RELOC_TICK_LOCAL || ctx_alone(ctx);
    rc = launch_db_scan(ctx, ctx->db_desc, ctx->db_off, ctx->db_records, ctx->cand_ids, ctx->cand_n, MAX_CAND,
                        ctx->f_desc, ctx->f_count, ctx->max_feat, ctx->db_max_rows, nullptr, ctx->m_qidx, ctx->m_tidx,
                        ctx->m_dist, ctx->m_n, MAX_REC_ROWS, &emit);
    if (!rc)
        rc = pnp_run_candidates(ctx, MAX_CAND, ctx->cand_n, ctx->K4, ctx->prm.ransac_iterations, (float)ctx->prm.ransac_reproj_px,
                                ctx->prm.ransac_confidence, seed, ctx->prm.min_matches, ctx->tick_flags, ctx->prm.min_inliers,
                                ctx->prm.global_min_inliers);
    ctx->latency_shapes = false;
    if (rc) return rc;
    TickParams fin = prm;
    fin.seq = tick_next_seq(ctx);
    hipLaunchKernelGGL(k_tick_finalize, dim3(1), dim3(64), 0, st, ctx->cand_ids, ctx->cand_n, ctx->p_out, ctx->db_pose,
                       ctx->f_count, fin, ctx->tick_flags, ctx->tick_res, ctx->tick_res_host, ctx->tick_res_ext);
    HIP_TRY(hipGetLastError());
    return RELOC_OK;
}

RELOC_API int reloc_set_camera(reloc_ctx *ctx, const double K4[4], const double base_to_cam_t[3], const double base_to_cam_R[9])
{
    ARG_CHECK_CTX(ctx, true, "ctx is NULL");
    if (K4) for (int k = 0; k < 4; ++k) ctx->K4[k] = K4[k];
    if (base_to_cam_t) for (int k = 0; k < 3; ++k) ctx->b2c_t[k] = base_to_cam_t[k];
    if (base_to_cam_R) for (int k = 0; k < 9; ++k) ctx->b2c_R[k] = base_to_cam_R[k];
    return db_reindex(ctx);
}

// The tick in three parts, so that several contexts on one stream can share ONE scan launch (reloc_tick_batch_dev):
//   begin: ORB, local candidates;  scan: whole-database scan (single or batched);  end: ranking, matches, PnP, gates.
static int tick_begin(reloc_ctx *ctx, const uint8_t *img_dev, int w, int h, int order, const TickParams &prm)
{
    ctx->orb_latency_shape = prm.mode == RELOC_TICK_LOCAL || ctx_alone(ctx);
    const int rc = orb_run_dev(ctx, img_dev, w, h, w * 3, 3, order, ctx->prm.nfeatures);
    ctx->orb_latency_shape = true;
    if (rc) return rc;
    if (prm.mode != RELOC_TICK_GLOBAL) launch_candidates_local(ctx, prm);
    return RELOC_OK;
}

static int tick_scan_single(reloc_ctx *ctx, const TickParams &prm)
{
    // G:329-344: only heading-compatible records are scored (the scan leaves count 0 on the others).  In AUTO
    // mode the scan and the ranking stand down on the device when the local search has found candidates.
    ScanMask mask;
    mask.xyh = ctx->db_xy_heading;
    for (int k = 0; k < 4; ++k) mask.q[k] = prm.base_pose[3 + k];
    mask.cos_tol = prm.cos_tol;
    mask.skip_if = prm.mode == RELOC_TICK_AUTO ? ctx->cand_n : nullptr;
    reloc_prof_begin(ctx, RELOC_PROF_DB_SCAN);
    const int rc = launch_db_scan(ctx, ctx->db_desc, ctx->db_off, ctx->db_records, nullptr, nullptr, (int)ctx->db_records, ctx->f_desc,
                                  ctx->f_count, ctx->max_feat, ctx->db_max_rows, ctx->db_counts, nullptr, nullptr, nullptr, nullptr, 0,
                                  &mask);
    reloc_prof_end(ctx, RELOC_PROF_DB_SCAN);
    return rc;
}

static int tick_end(reloc_ctx *ctx, const TickParams &prm, uint64_t seed)
{
    if (prm.mode != RELOC_TICK_LOCAL) launch_topk_counts(ctx, ctx->prm.global_max_candidates, ctx->cand_ids, nullptr, prm.mode == RELOC_TICK_AUTO);
    return tick_solve(ctx, prm, seed);
}

// ---- frame-batched forms: n contexts on one stream, every stage ONE launch (blockIdx = frame) -----------------
// scan_rows != NULL (sharded scan half): frame f's list goes to row f of scan_rows (2k + 2 int32: k ids, k counts, feature
// count, 0) with id_base added; otherwise it becomes the frame's candidate list.
static void launch_topk_counts_batch(reloc_ctx *const *ctxs, int n, int k, bool auto_mode, int id_base, int32_t *scan_rows)
{
    reloc_ctx *c0 = ctxs[0];
    const int max_count = c0->db_max_rows < c0->max_feat ? c0->db_max_rows : c0->max_feat;
    TopkBatch b;
    for (int f = 0; f < RELOC_BATCH_MAX; ++f) {
        reloc_ctx *c = ctxs[f < n ? f : 0];
        TopkFrame &F = b.f[f];
        int32_t *row = scan_rows ? scan_rows + (size_t)(f < n ? f : 0) * (2 * k + 2) : nullptr;
        F.counts = c->db_counts;
        F.out_ids = row ? row : c->cand_ids; F.out_counts = row ? row + k : nullptr; F.out_n = c->cand_n;
        F.skip_if = auto_mode ? (const int32_t *)c->cand_n : (const int32_t *)nullptr;
        F.relocating = c->tick_flags; F.f_count = c->f_count; F.out_nfeat = row ? row + 2 * k : nullptr;
    }
    hipLaunchKernelGGL(k_topk_counts_batch, dim3(n), dim3(TOPK_HIST_THREADS), (size_t)(max_count + 2) * sizeof(int), c0->stream, b,
                       (int)c0->db_records, k, id_base, c0->prm.min_matches, max_count);
}

// emit pass + PnP + finalisation of n frames: 1 + 3 + 1 launches.  res_ext_base != NULL: frame f's record also goes to
// res_ext_base + f (device or pinned memory) instead of the context's reloc_tick_result_to target.
static int tick_solve_batch(reloc_ctx *const *ctxs, int n, const double *base_poses, int mode, int check_consistency,
                            const uint64_t *seeds, TickResult *res_ext_base)
{
    int rc;
    if ((rc = launch_db_emit_batch(ctxs, n))) return rc;
    if ((rc = pnp_run_candidates_batch(ctxs, n, MAX_CAND, seeds))) return rc;
    reloc_ctx *c0 = ctxs[0];
    const TickParams prm = make_tick_params(c0, base_poses, mode, check_consistency);
    FinalBatch b;
    for (int f = 0; f < RELOC_BATCH_MAX; ++f) {
        const int g = f < n ? f : 0;
        reloc_ctx *c = ctxs[g];
        FinalFrame &F = b.f[f];
        F.cand_ids = c->cand_ids; F.cand_n = c->cand_n; F.pnp = c->p_out; F.f_count = c->f_count; F.relocating = c->tick_flags;
        F.res = c->tick_res; F.res_host = c->tick_res_host; F.res_ext = res_ext_base ? res_ext_base + g : c->tick_res_ext;
        for (int k = 0; k < 7; ++k) F.base_pose[k] = base_poses[7 * g + k];
        F.seq = f < n ? tick_next_seq(c) : 0;
    }
    hipLaunchKernelGGL(k_tick_finalize_batch, dim3(n), dim3(64), 0, c0->stream, b, c0->db_pose, prm);
    HIP_TRY(hipGetLastError());
    return RELOC_OK;
}

RELOC_API int reloc_tick_dev(reloc_ctx *ctx, const uint8_t *img_dev, int w, int h, int order, const double base_pose[7],
                             int global_reloc, uint64_t seed)
{
    if (ctx) ctx->tick_failed = true;         // until the finalisation has been enqueued (tick_next_seq)
    ARG_CHECK_CTX(ctx, img_dev && base_pose && w >= 64 && h >= 64 && global_reloc >= 0 && global_reloc <= 2, "reloc_tick_dev");
    if (!db_ready(ctx)) { reloc_set_error("no database uploaded"); return RELOC_E_STATE; }
    if (ctx->max_feat > 65535) { reloc_set_error("tick: max_feat must be <= 65535"); return RELOC_E_CAPACITY; }
    int rc;
    const TickParams prm = make_tick_params(ctx, base_pose, global_reloc, -1);
    if ((rc = tick_begin(ctx, img_dev, w, h, order, prm))) return rc;
    if (prm.mode != RELOC_TICK_LOCAL && (rc = tick_scan_single(ctx, prm))) return rc;
    return tick_end(ctx, prm, seed);
}

RELOC_API int reloc_tick_batch_dev(reloc_ctx *const *ctxs, int n, const uint8_t *const *imgs_dev, int w, int h, int order,
                                   const double *base_poses, int global_reloc, const uint64_t *seeds)
{
    if (ctxs && n >= 1 && n <= 8)
        for (int f = 0; f < n; ++f) if (ctxs[f]) ctxs[f]->tick_failed = true;
    ARG_CHECK(ctxs && imgs_dev && base_poses && n >= 1 && n <= 8 && w >= 64 && h >= 64 && global_reloc >= 0 && global_reloc <= 2,
              "reloc_tick_batch_dev");
    for (int f = 0; f < n; ++f) {
        reloc_ctx *c = ctxs[f];
        ARG_CHECK(c && imgs_dev[f], "reloc_tick_batch_dev: NULL context or frame");
        if (!db_ready(c)) { reloc_set_error("no database uploaded"); return RELOC_E_STATE; }
        if (c->stream != ctxs[0]->stream || c->device != ctxs[0]->device || c->db_desc != ctxs[0]->db_desc ||
            c->db_records != ctxs[0]->db_records || c->max_feat != ctxs[0]->max_feat ||
            memcmp(&c->prm, &ctxs[0]->prm, sizeof(reloc_params)) != 0 || memcmp(c->K4, ctxs[0]->K4, sizeof(c->K4)) != 0 ||
            memcmp(c->b2c_t, ctxs[0]->b2c_t, sizeof(c->b2c_t)) != 0 || memcmp(c->b2c_R, ctxs[0]->b2c_R, sizeof(c->b2c_R)) != 0) {
            reloc_set_error("tick batch: the contexts must share one stream (reloc_set_stream), one device and one database "
                            "(reloc_db_share) and have equal feature capacity, matcher parameters (reloc_set_params) and camera (reloc_set_camera)");
            return RELOC_E_STATE;
        }
        for (int g = 0; g < f; ++g) ARG_CHECK(ctxs[g] != c, "reloc_tick_batch_dev: a context appears twice");
    }
    (void)hipSetDevice(ctxs[0]->device);
    int rc;
    double q[8 * 4];
    for (int f = 0; f < n; ++f)
        for (int k = 0; k < 4; ++k) q[4 * f + k] = base_poses[7 * f + 3 + k];
    // every stage of the batch is ONE launch with the frame as a grid dimension: ORB 5, local candidates n (LOCAL / AUTO
    // only), scan 1, ranking 1, emit 1, PnP 3, finalisation 1 -- 12 launches for 8 frames in whole-database mode instead
    // of the 100 of eight per-frame ticks (their serial chain of small kernels was what a batch spent its time on)
    if ((rc = orb_run_batch_dev(ctxs, n, imgs_dev, w, h, w * 3, order, ctxs[0]->prm.nfeatures))) return rc;
    const TickParams prm0 = make_tick_params(ctxs[0], base_poses, global_reloc, -1);
    if (global_reloc != RELOC_TICK_GLOBAL)
        for (int f = 0; f < n; ++f) launch_candidates_local(ctxs[f], make_tick_params(ctxs[f], base_poses + 7 * f, global_reloc, -1));
    if (global_reloc != RELOC_TICK_LOCAL) {
        reloc_prof_begin(ctxs[0], RELOC_PROF_DB_SCAN);
        rc = launch_db_scan_batch(ctxs, n, q, prm0.cos_tol, global_reloc == RELOC_TICK_AUTO);
        reloc_prof_end(ctxs[0], RELOC_PROF_DB_SCAN);
        if (rc) return rc;
        launch_topk_counts_batch(ctxs, n, ctxs[0]->prm.global_max_candidates, global_reloc == RELOC_TICK_AUTO, 0, nullptr);
    }
    return tick_solve_batch(ctxs, n, base_poses, global_reloc, -1, seeds, nullptr);
}

// Waits for the result record of the LAST tick enqueued on this context by polling its sequence stamp in pinned host memory
// (written by the tick's last kernel behind the record's body).  The runtime's own completion path (hipStreamSynchronize:
// completion signal, and after an idle period an interrupt) costs tens of microseconds more than the store is late; after
// 20 ms without the stamp this falls back to it.  Does NOT wait for work enqueued behind the tick.
RELOC_API int reloc_tick_wait(reloc_ctx *ctx)
{
    ARG_CHECK_CTX(ctx, true, "ctx is NULL");
    if (ctx->tick_failed) {
        reloc_set_error("reloc_tick_wait: the last tick on this context failed before its result record was enqueued");
        return RELOC_E_STATE;
    }
    const int want = ctx->tick_seq;
    if (want == 0 || !ctx->tick_res_host) { HIP_TRY(hipStreamSynchronize(ctx->stream)); return RELOC_OK; }
    volatile int32_t *stamp = (volatile int32_t *)&ctx->tick_res_host->pad[0];
    struct timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (unsigned spin = 1;; ++spin) {
        if (*stamp == want) { __atomic_thread_fence(__ATOMIC_ACQUIRE); return RELOC_OK; }
        __builtin_ia32_pause();
        if ((spin & 0x3FF) == 0) {
            struct timespec t1;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec) > 20000000ll) break;
        }
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (*stamp != want) { reloc_set_error("reloc_tick_wait: the stream drained without the tick's result record (a tick that failed to enqueue?)"); return RELOC_E_STATE; }
    return RELOC_OK;
}

RELOC_API int reloc_tick_result(reloc_ctx *ctx, double anchor_pose[7], int32_t *n_inl, float *reproj, int32_t *lm_idx,
                                int32_t *outcome, int32_t *n_candidates)
{
    ARG_CHECK_CTX(ctx, true, "ctx is NULL");
    { const int rc = reloc_tick_wait(ctx); if (rc) return rc; }
    const TickResult r = *ctx->tick_res_host;          // written by k_tick_finalize itself (pinned, device-visible)
    if (anchor_pose) for (int k = 0; k < 7; ++k) anchor_pose[k] = r.anchor_pose[k];
    if (n_inl) *n_inl = r.n_inl;
    if (reproj) *reproj = (float)r.reproj;
    if (lm_idx) *lm_idx = r.lm_idx;
    if (outcome) *outcome = r.outcome;
    if (n_candidates) *n_candidates = r.n_candidates;
    return RELOC_OK;
}

RELOC_API const void *reloc_tick_result_dev(reloc_ctx *ctx) { return ctx ? ctx->tick_res : nullptr; }

RELOC_API int reloc_tick_result_to(reloc_ctx *ctx, void *pinned_record)
{
    ARG_CHECK_CTX(ctx, true, "ctx is NULL");
    ctx->tick_res_ext = (TickResult *)pinned_record;
    return RELOC_OK;
}

RELOC_API int reloc_tick_result_ex(reloc_ctx *ctx, double anchor_pose[7], int32_t *n_inl, double *reproj, int32_t *lm_idx,
                                   int32_t *outcome, int32_t *n_candidates, int32_t *n_features, int32_t *relocating)
{
    ARG_CHECK_CTX(ctx, true, "ctx is NULL");
    { const int rc = reloc_tick_wait(ctx); if (rc) return rc; }
    const TickResult r = *ctx->tick_res_host;          // written by k_tick_finalize itself (pinned, device-visible)
    if (anchor_pose) for (int k = 0; k < 7; ++k) anchor_pose[k] = r.anchor_pose[k];
    if (n_inl) *n_inl = r.n_inl;
    if (reproj) *reproj = r.reproj;
    if (lm_idx) *lm_idx = r.lm_idx;
    if (outcome) *outcome = r.outcome;
    if (n_candidates) *n_candidates = r.n_candidates;
    if (n_features) *n_features = r.n_features;
    if (relocating) *relocating = r.relocating;
    return RELOC_OK;
}

RELOC_API int reloc_tick(reloc_ctx *ctx, const uint8_t *img, int w, int h, int order, const double base_pose[7],
                         int global_reloc, uint64_t seed, double anchor_pose[7], int32_t *n_inl, float *reproj,
                         int32_t *lm_idx, int32_t *outcome, int32_t *n_candidates)
{
    ARG_CHECK_CTX(ctx, img && base_pose && w >= 64 && h >= 64, "reloc_tick");
    if (w > ctx->max_w || h > ctx->max_h) { reloc_set_error("frame exceeds ctx capacity"); return RELOC_E_CAPACITY; }
    HIP_TRY(hipMemcpyAsync(ctx->frame_img, img, (size_t)w * h * 3, hipMemcpyHostToDevice, ctx->stream));
    int rc = reloc_tick_dev(ctx, ctx->frame_img, w, h, order, base_pose, global_reloc, seed);
    if (rc) return rc;
    return reloc_tick_result(ctx, anchor_pose, n_inl, reproj, lm_idx, outcome, n_candidates);
}

// ---- sharded database: scan and solve halves ------------------------------------------------------
RELOC_API int reloc_tick_scan_dev(reloc_ctx *ctx, const uint8_t *img_dev, int w, int h, int order, const double base_pose[7],
                                  int32_t *topk_ids_dev, int32_t *topk_counts_dev, int k)
{
    ARG_CHECK_CTX(ctx, img_dev && topk_ids_dev && topk_counts_dev && k > 0 && k <= MAX_CAND && w >= 64 && h >= 64,
              "reloc_tick_scan_dev");
    if (!db_ready(ctx)) { reloc_set_error("no database uploaded"); return RELOC_E_STATE; }
    int rc;
    ctx->orb_latency_shape = ctx_alone(ctx);                // the sharded tick scans: a neighbour of scans unless the ctx is alone
    rc = orb_run_dev(ctx, img_dev, w, h, w * 3, 3, order, ctx->prm.nfeatures);
    ctx->orb_latency_shape = true;
    if (rc) return rc;
    ScanMask mask;
    mask.xyh = base_pose ? ctx->db_xy_heading : nullptr;
    for (int k = 0; k < 4; ++k) mask.q[k] = base_pose ? base_pose[3 + k] : (k == 3 ? 1.0 : 0.0);
    mask.cos_tol = heading_cos_tol_host(ctx);
    reloc_prof_begin(ctx, RELOC_PROF_DB_SCAN);
    rc = launch_db_scan(ctx, ctx->db_desc, ctx->db_off, ctx->db_records, nullptr, nullptr, (int)ctx->db_records, ctx->f_desc,
                        ctx->f_count, ctx->max_feat, ctx->db_max_rows, ctx->db_counts, nullptr, nullptr, nullptr, nullptr, 0, &mask);
    reloc_prof_end(ctx, RELOC_PROF_DB_SCAN);
    if (rc) return rc;
    launch_topk_counts(ctx, k, topk_ids_dev, topk_counts_dev, false);
    HIP_TRY(hipGetLastError());
    return RELOC_OK;
}

__global__ void k_set_candidates(const int32_t *__restrict__ ids, int n, int32_t *__restrict__ cand_ids, int32_t *__restrict__ cand_n)
{
    if (threadIdx.x == 0) {
        int m = 0;
        for (int i = 0; i < n; ++i)
            if (ids[i] >= 0) cand_ids[m++] = ids[i];
        *cand_n = m;
    }
}

struct SetCandBatch { int32_t *cand_ids[RELOC_BATCH_MAX], *cand_n[RELOC_BATCH_MAX], *flags[RELOC_BATCH_MAX]; };
__global__ void k_set_candidates_batch(const int32_t *__restrict__ ids, int k, SetCandBatch b, int flag)
{
    const int f = blockIdx.x;
    if (threadIdx.x == 0) {
        int m = 0;
        for (int i = 0; i < k; ++i)
            if (ids[f * k + i] >= 0) b.cand_ids[f][m++] = ids[f * k + i];
        *b.cand_n[f] = m;
        *b.flags[f] = flag;
    }
}

RELOC_API int reloc_tick_solve_dev(reloc_ctx *ctx, const int32_t *cand_ids_dev, int n_cand, const double base_pose[7],
                                   int check_consistency, uint64_t seed)
{
    if (ctx) ctx->tick_failed = true;
    ARG_CHECK_CTX(ctx, cand_ids_dev && base_pose && n_cand >= 0 && n_cand <= MAX_CAND, "reloc_tick_solve_dev");
    if (!db_ready(ctx)) { reloc_set_error("no database uploaded"); return RELOC_E_STATE; }
    hipLaunchKernelGGL(k_set_candidates, dim3(1), dim3(64), 0, ctx->stream, cand_ids_dev, n_cand, ctx->cand_ids, ctx->cand_n);
    // candidates of a sharded whole-database search carry the relocation gates unless the caller asks for the
    // consistency check (= local candidates)
    hipLaunchKernelGGL(k_set_flag, dim3(1), dim3(1), 0, ctx->stream, ctx->tick_flags, check_consistency ? 0 : 1);
    const TickParams prm = make_tick_params(ctx, base_pose, check_consistency ? RELOC_TICK_LOCAL : RELOC_TICK_GLOBAL, check_consistency);
    return tick_solve(ctx, prm, seed);
}

// ---- sharded database, batched and device-resident (BASELINE.json config 4) ---------------------------------------
// One call per half and batch, so a rank's host enqueues a batch of 8 frames with three calls instead of ~40: the
// contexts of a batch share ONE stream and one shard (as for reloc_tick_batch_dev), and the exchange between the halves
// (RCCL all-gather of the rows written here) is enqueued on that same stream by the caller.
static int shard_batch_check(reloc_ctx *const *ctxs, int n, const char *what)
{
    ARG_CHECK(ctxs && n >= 1 && n <= 8, what);
    for (int f = 0; f < n; ++f) {
        reloc_ctx *c = ctxs[f];
        ARG_CHECK(c, "shard batch: NULL context");
        if (!db_ready(c)) { reloc_set_error("no database uploaded"); return RELOC_E_STATE; }
        if (c->stream != ctxs[0]->stream || c->device != ctxs[0]->device || c->db_desc != ctxs[0]->db_desc ||
            c->db_records != ctxs[0]->db_records || c->max_feat != ctxs[0]->max_feat ||
            memcmp(&c->prm, &ctxs[0]->prm, sizeof(reloc_params)) != 0 || memcmp(c->K4, ctxs[0]->K4, sizeof(c->K4)) != 0 ||
            memcmp(c->b2c_t, ctxs[0]->b2c_t, sizeof(c->b2c_t)) != 0 || memcmp(c->b2c_R, ctxs[0]->b2c_R, sizeof(c->b2c_R)) != 0) {
            reloc_set_error("shard batch: the contexts must share one stream (reloc_set_stream), one device and one database "
                            "(reloc_db_share) and have equal feature capacity, matcher parameters and camera");
            return RELOC_E_STATE;
        }
        for (int g = 0; g < f; ++g) ARG_CHECK(ctxs[g] != c, "shard batch: a context appears twice");
    }
    return RELOC_OK;
}

RELOC_API int reloc_shard_scan_batch_dev(reloc_ctx *const *ctxs, int n, const uint8_t *const *imgs_dev, int w, int h, int order,
                                         const double *base_poses, int k, int64_t id_base, int32_t *scan_out_dev)
{
    int rc = shard_batch_check(ctxs, n, "reloc_shard_scan_batch_dev");
    if (rc) return rc;
    ARG_CHECK(imgs_dev && scan_out_dev && k > 0 && k <= MAX_CAND && w >= 64 && h >= 64 && id_base >= 0 &&
              id_base + ctxs[0]->db_records <= 0x7fffffff, "reloc_shard_scan_batch_dev");
    (void)hipSetDevice(ctxs[0]->device);
    double q[8 * 4];
    for (int f = 0; f < n; ++f) {
        ARG_CHECK(imgs_dev[f], "reloc_shard_scan_batch_dev: NULL frame");
        for (int j = 0; j < 4; ++j) q[4 * f + j] = base_poses ? base_poses[7 * f + 3 + j] : (j == 3 ? 1.0 : 0.0);
    }
    if ((rc = orb_run_batch_dev(ctxs, n, imgs_dev, w, h, w * 3, order, ctxs[0]->prm.nfeatures))) return rc;
    reloc_prof_begin(ctxs[0], RELOC_PROF_DB_SCAN);
    rc = launch_db_scan_batch(ctxs, n, q, heading_cos_tol_host(ctxs[0]), false, base_poses != nullptr);
    reloc_prof_end(ctxs[0], RELOC_PROF_DB_SCAN);
    if (rc) return rc;
    launch_topk_counts_batch(ctxs, n, k, false, (int)id_base, scan_out_dev);
    HIP_TRY(hipGetLastError());
    return RELOC_OK;
}

// The merge every rank performs on the gathered lists (G:342-343 over the whole database): per frame the k best
// (count desc, global id desc) of the W x k entries, by rank counting -- keys are unique.  One block per frame.
__global__ __launch_bounds__(256) void k_shard_merge(const int32_t *__restrict__ all_scan, int world, int stride_w, int k, int id_base,
                                                     int n_local, int32_t *__restrict__ win_gid, int32_t *__restrict__ cand_local,
                                                     int32_t *__restrict__ n_feat)
{
    RELOC_SMALL_KERNEL_PRIO();
    extern __shared__ unsigned long long s_key[];               // world * k
    const int f = blockIdx.x, row = 2 * k + 2, E = world * k;
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        const int32_t *r = all_scan + (size_t)(e / k) * stride_w + (size_t)f * row;
        const int gid = r[e % k], cnt = r[k + e % k];
        s_key[e] = gid >= 0 ? ((unsigned long long)(unsigned)cnt << 32) | (unsigned)(gid + 1) : 0ull;
    }
    if ((int)threadIdx.x < k) { win_gid[f * k + threadIdx.x] = -1; cand_local[f * k + threadIdx.x] = -1; }
    if (threadIdx.x == 0) {
        int m = -1;
        for (int wv = 0; wv < world; ++wv) m = max(m, all_scan[(size_t)wv * stride_w + (size_t)f * row + 2 * k]);
        n_feat[f] = m;                                          // ranks without records report -1
    }
    __syncthreads();
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
        const unsigned long long mine = s_key[e];
        if (!mine) continue;
        int rank = 0;
        for (int j = 0; j < E; ++j) rank += s_key[j] > mine;
        if (rank < k) {
            const int gid = (int)(unsigned)(mine & 0xFFFFFFFFu) - 1;
            win_gid[f * k + rank] = gid;
            cand_local[f * k + rank] = gid >= id_base && gid < id_base + n_local ? gid - id_base : -1;
        }
    }
}

RELOC_API int reloc_shard_merge_dev(reloc_ctx *ctx, const int32_t *all_scan_dev, int world, int64_t stride_rank, int n, int k,
                                    int64_t id_base, int64_t n_local, int32_t *win_gid_dev, int32_t *cand_local_dev,
                                    int32_t *n_feat_dev)
{
    ARG_CHECK_CTX(ctx, all_scan_dev && win_gid_dev && cand_local_dev && n_feat_dev && world >= 1 && n >= 1 && n <= 8 && k > 0 &&
                  k <= MAX_CAND && stride_rank >= (int64_t)n * (2 * k + 2) && id_base >= 0 && n_local >= 0 &&
                  id_base + n_local <= 0x7fffffff && (int64_t)world * k <= 4096, "reloc_shard_merge_dev");
    hipLaunchKernelGGL(k_shard_merge, dim3(n), dim3(256), (size_t)world * k * sizeof(unsigned long long), ctx->stream, all_scan_dev, world,
                       (int)stride_rank, k, (int)id_base, (int)n_local, win_gid_dev, cand_local_dev, n_feat_dev);
    HIP_TRY(hipGetLastError());
    return RELOC_OK;
}

RELOC_API int reloc_shard_solve_batch_dev(reloc_ctx *const *ctxs, int n, const int32_t *cand_local_dev, int k,
                                          const double *base_poses, const uint64_t *seeds, void *res_out)
{
    if (ctxs && n >= 1 && n <= 8)
        for (int f = 0; f < n; ++f) if (ctxs[f]) ctxs[f]->tick_failed = true;
    int rc = shard_batch_check(ctxs, n, "reloc_shard_solve_batch_dev");
    if (rc) return rc;
    ARG_CHECK(cand_local_dev && base_poses && res_out && k > 0 && k <= MAX_CAND, "reloc_shard_solve_batch_dev");
    (void)hipSetDevice(ctxs[0]->device);
    SetCandBatch sb;
    for (int f = 0; f < RELOC_BATCH_MAX; ++f) {
        reloc_ctx *c = ctxs[f < n ? f : 0];
        sb.cand_ids[f] = c->cand_ids; sb.cand_n[f] = c->cand_n; sb.flags[f] = c->tick_flags;
    }
    // candidates of a whole-database search: relocation gates (flag 1), no consistency check
    hipLaunchKernelGGL(k_set_candidates_batch, dim3(n), dim3(64), 0, ctxs[0]->stream, cand_local_dev, k, sb, 1);
    return tick_solve_batch(ctxs, n, base_poses, RELOC_TICK_GLOBAL, 0, seeds, (TickResult *)res_out);
}

// read back the per-candidate PnP records of the last tick (parity taps for tests)
RELOC_API int reloc_tick_debug(reloc_ctx *ctx, int32_t *cand_ids, int32_t *n_cand, int32_t *n_matches, int32_t *n_inl,
                               int32_t *ok, double *reproj, double *Rt)
{
    ARG_CHECK_CTX(ctx, cand_ids && n_cand, "reloc_tick_debug");
    PnpOut po[MAX_CAND];
    HIP_TRY(hipMemcpyAsync(n_cand, ctx->cand_n, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(cand_ids, ctx->cand_ids, sizeof(int32_t) * MAX_CAND, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(po, ctx->p_out, sizeof(po), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int s = 0; s < *n_cand && s < MAX_CAND; ++s) {
        if (n_matches) n_matches[s] = po[s].n_matches;
        if (n_inl) n_inl[s] = po[s].n_inl;
        if (ok) ok[s] = po[s].ok;
        if (reproj) reproj[s] = po[s].reproj_mean;
        if (Rt) memcpy(Rt + 12 * s, po[s].Rt, sizeof(double) * 12);
    }
    return RELOC_OK;
}
