// reloc_match.hip -- 256-bit Hamming matching on gfx950 (CDNA4), XOR + popcount on the VALU.
//
// Serves cv2.BFMatcher(NORM_HAMMING, crossCheck=True).match        (reference M:211,327; G:337)
//        cv2.BFMatcher(NORM_HAMMING, crossCheck=False).knnMatch k=2 (reference S:46,68)
//        the whole-database candidate scoring of variant G          (reference G:329-344)
// and the all-pairs u16 distance matrix of BASELINE.json config 5.
//
// No MFMA: this is bitwise work.  Instruction costs measured on MI355X (tools/ubench_valu.hip,
// profiles/ubench_valu_r1.log): v_xor/v_or/v_and/v_add_u32 and the 16-bit v_min_u16 /
// v_lshlrev_b16 issue in ~2.5 cycles per wave64, while v_bcnt_u32_b32, 32-bit min/shift and every
// three-operand VOP3 (v_lshl_or, v_min3, v_add3...) take ~4.2.  The kernels are built around that:
//   - the accumulate form  v_bcnt_u32_b32 D, S0, S1 (D = popcount(S0) + S1)  keeps a 256-bit pair
//     at 8 xor + 8 bcnt (about 54 cycles per 64 pairs per SIMD: the floor of this problem);
//   - the argmin bookkeeping uses ONE 16-bit key per pair for both directions
//     (distance << 7 | row-in-chunk << 3 | column slot) with v_lshlrev_b16, v_or_b32 and two
//     v_min_u16, all in the cheap class.
// Hot kernel (k_db_scan): a wave keeps all 512 current-frame descriptors (8 per lane, 64 VGPRs);
// a database record's teach rows are wave-uniform, arrive through the scalar cache
// (s_load_dwordx8, the fetch of row t+1 issued as soon as row t has landed) and feed the VALU as
// SGPR operands, so one 32-byte scalar fetch pays for 512 pairs and no LDS traffic is in the
// inner loop.  A record's rows are dealt to the 4 waves of a workgroup as balanced contiguous
// ranges, walked in 16-row chunks and finished with ONE flexible chunk of 1-15 rows.  Per chunk a wave produces (a) per lane and column the best row
// (running 16-bit minimum), merged into LDS with ds_min_u32, and (b) per row the best column:
// an in-lane 8-way 16-bit minimum, then a register-tile butterfly across lanes
// (v_permlane16_swap / ds_bpermute).  Mutual nearest neighbours are resolved in LDS; the kernel
// emits the per-record count and optionally the (queryIdx, trainIdx, distance) list in queryIdx
// order.  Keys are (distance << k | index): the minimum of packed keys is the smallest distance
// with the LOWEST index on ties, which is the tie rule of the specification.
#include <stdlib.h>
#include <new>
#include <type_traits>
#include <utility>

#include "reloc_internal.h"

typedef uint32_t u32;

__device__ __forceinline__ u32 bcnt_acc(u32 x, u32 acc)
{
    u32 r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}
__device__ __forceinline__ u32 min_u16(u32 a, u32 b)
{
    u32 r;
    asm("v_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ u32 shl7_u16(u32 a)
{
    u32 r;
    asm("v_lshlrev_b16 %0, 7, %1" : "=v"(r) : "v"(a));
    return r;
}
__device__ __forceinline__ u32 ham8(const u32 q[8], const uint4 a, const uint4 b, u32 init)
{
    u32 acc = init;
    acc = bcnt_acc(q[0] ^ a.x, acc);
    acc = bcnt_acc(q[1] ^ a.y, acc);
    acc = bcnt_acc(q[2] ^ a.z, acc);
    acc = bcnt_acc(q[3] ^ a.w, acc);
    acc = bcnt_acc(q[4] ^ b.x, acc);
    acc = bcnt_acc(q[5] ^ b.y, acc);
    acc = bcnt_acc(q[6] ^ b.z, acc);
    acc = bcnt_acc(q[7] ^ b.w, acc);
    return acc;
}

// two independent 256-bit distances with their instruction chains interleaved: an in-order wave
// then always has an independent instruction behind a v_bcnt (SQ_WAIT_INST_ANY was 40 % of wave time
// with one serial xor -> bcnt chain per pair at 4 waves per SIMD)
__device__ __forceinline__ void ham8x2(const u32 q0[8], const u32 q1[8], const uint4 a, const uint4 b, u32 init0, u32 init1,
                                       u32 &d0, u32 &d1)
{
    const u32 w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    u32 acc0 = init0, acc1 = init1;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const u32 x0 = q0[k] ^ w[k], x1 = q1[k] ^ w[k];
        acc0 = bcnt_acc(x0, acc0);
        acc1 = bcnt_acc(x1, acc1);
    }
    d0 = acc0;
    d1 = acc1;
}

// NC 256-bit distances of one row (8 wave-uniform words w, SGPRs) against NC descriptors held in registers, as NC
// accumulator chains visited round-robin, word by word: xor, its v_bcnt, next column.  The order is pinned with
// asm volatile because it is the whole point -- measured on MI355X (tools/exp_matrix2.hip, 20000 x 20000, same run):
//   2 chains interleaved (accumulator reused after 4 instructions)      190 us  2.10 T pairs/s   (compiler-scheduled: same)
//   4 chains                                                          187 us
//   8 chains, xor directly followed by its own v_bcnt                   160 us  2.49 T pairs/s
//   8 chains, xor issued one step ahead of its v_bcnt                   189 us
//   16 chains                                                           same as 8
// i.e. a v_bcnt must not read the accumulator a v_bcnt wrote fewer than ~16 instructions earlier, while the xor -> bcnt
// pair itself wants to stay adjacent.  The compiler's scheduler, left free with the same 8 accumulators, produces 187 us.
// One statement per INSTRUCTION on purpose: the compiler then puts an `s_nop 0` between each xor and its bcnt (61 per
// row), and that spacing is part of the result -- the same order written as one asm block per word, same run, whole
// matrix kernel: xor; bcnt back to back 202 us, xor; s_nop 0; bcnt 188, xor; bcnt; s_nop 0 186, this form 177
// (profiles/r2_exp_matrix2_spacing.log); in k_db_scan the block form without the s_nop costs 170.7 vs 161.0 us.
template <int NC>
__device__ __forceinline__ void ham8_cols(const u32 (&q)[NC][8], const u32 (&w)[8], u32 (&h)[NC])
{
    // the first word starts each accumulator from the inline constant 0 (no v_mov per chain and row: 158.7 -> 156.3 us)
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            u32 x;
            asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "s"(w[k]), "v"(q[j][k]));
            if (k == 0) asm volatile("v_bcnt_u32_b32 %0, %1, 0" : "=v"(h[j]) : "v"(x));
            else asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(h[j]) : "v"(x));
        }
}

// ---- scalar-cache row prefetch ------------------------------------------------------------------
// A teach row (32 B, wave-uniform address) arrives through the scalar cache.  Left alone, the compiler
// places each scalar load directly in front of its first use, which stalls the wave for the whole
// scalar-cache latency once per row.  scan_chunk instead issues the fetch of row t+1 between "row t has
// landed" and "row t is used": srow_landed() is an empty asm that reads one dword of row t, so the
// compiler's own waitcnt bookkeeping puts the wait for row t there (scalar loads return out of order:
// any wait is a wait for all of them, hence exactly one fetch in flight), and sched barriers keep the
// next fetch above the ~180 VALU instructions that hide its latency.
__device__ __forceinline__ void srow_landed(u32 first_dword) { asm volatile("" ::"s"(first_dword)); }

__device__ __forceinline__ u32 umin(u32 a, u32 b) { return a < b ? a : b; }
__device__ __forceinline__ double uniform_f64(double v)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ u32 umax(u32 a, u32 b) { return a > b ? a : b; }

// One butterfly exchange: on entry a = vector i, b = vector i+K on every lane.  On exit the
// return value on lane l is min over lanes {l, l^K} of (bit K of l ? vector i+K : vector i).
template <int K>
__device__ __forceinline__ u32 bfly(u32 a, u32 b, int lane)
{
    if constexpr (K == 32) {
        auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
        return umin(r[0], r[1]);
    } else if constexpr (K == 16) {
        auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
        return umin(r[0], r[1]);
    } else {
        const bool hi = lane & K;
        const u32 mine = hi ? b : a;
        const u32 theirs = hi ? a : b;
        return umin(mine, dpp_xor<K>(theirs));
    }
}

// The same exchange with the lane movement IN the minimum (v_min_u32 with a DPP operand) for the lane bits where a bank mask
// can tell the two halves apart: bit 2 (K = 4: row_shl:4 on banks 0 / 2 serves the lanes that keep vector a, row_shr:4 on banks
// 1 / 3 those that keep b -- a lane needs only ITS vector's partner) and bit 3 (K = 8: row_ror:8, banks 0-1 / 2-3).  Two
// instructions instead of two selects, a DPP move and a minimum.  Inline asm: the s_nop covers the two wait states a DPP read
// needs after a VALU write of its source, which the compiler does not count for an asm statement.
template <int K>
__device__ __forceinline__ u32 bfly_dpp(u32 a, u32 b)
{
    static_assert(K == 4 || K == 8, "bank-maskable lane bits only");
    u32 r;
    if constexpr (K == 4)
        asm("s_nop 1\n\tv_min_u32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_min_u32_dpp %0, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa" : "=&v"(r) : "v"(a), "v"(b));
    else
        asm("s_nop 1\n\tv_min_u32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_min_u32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc" : "=&v"(r) : "v"(a), "v"(b));
    return r;
}
// all-reduce over lane bit 0 / bit 1 (quad_perm exchange in the minimum)
__device__ __forceinline__ u32 allmin_quad(u32 x)
{
    asm("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(x));
    return x;
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>), in order
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// One R-row chunk (R = 16 or 4) of a record against the wave's 64*NJ columns.
//   q[j]    : descriptor of column colbase + j*64 + lane (padding columns repeat the last real column:
//             a duplicate offers the same distance with a larger index, so it never wins a minimum)
//   FLEX    : tail chunk of nr < R rows: the rows it does not have are skipped (wave-uniform branch) and enter the
//             butterfly as "infinity"; their fetch addresses are clamped to the record's last row
// One 16-bit key per pair serves both directions: distance << 7 | row-in-chunk << 3 | column slot.
// Among the rows of one column the slot bits are equal, so the minimum is (distance, row); among the
// columns of one row the row bits are equal, so the minimum is (distance, slot).  Per pair that is
// shift + or + 2 min on top of the 16 instructions of the distance.
template <int NJ, int R, bool FLEX>
__device__ __forceinline__ void scan_chunk(const uint4 *__restrict__ rec, int n, int tc, int nr, const u32 (&q)[NJ][8],
                                           u32 colbase, u32 *rowkey, u32 *colbest, bool single_cb, int lane)
{
    // (column minima in LDS instead of registers: measured and dropped, profiles/README.md "Dropped experiments" #1)
    u32 cb16[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) cb16[j] = 0xFFFFu;
    // The R row keys are reduced by a register-tile butterfly, but AS THEY COME: the rows of the chunk
    // are visited in bit-reversed order (0, R/2, R/4, 3R/4, ...), so the two operands of every butterfly node are
    // finished right after each other and at most log2(R) + 1 partial results are alive instead of R keys -- 11
    // VGPRs less at R = 16, which takes the kernel from 115 to <= 104 registers: four resident workgroups then leave
    // 96 registers per SIMD lane free, enough for the small kernels of other streams to run BESIDE the scan.
    constexpr int LOG_R = R == 16 ? 4 : (R == 8 ? 3 : 2);
    u32 stk[LOG_R + 1];
    // Teach rows come through the scalar cache, one fetch in flight (see srow_landed).
    auto bitrev = [](int i) { int r = 0; for (int b = 0; b < LOG_R; ++b) r |= ((i >> b) & 1) << (LOG_R - 1 - b); return r; };
    auto row_of = [&](int i) { const int t = bitrev(i); return FLEX ? min(tc + t, n - 1) : tc + t; };   // wave-uniform
    uint4 a = rec[2 * row_of(0)], b = rec[2 * row_of(0) + 1];
    // compile-time row and column indices: the key constants are immediates.
    // (bookkeeping software-pipelined into the next row's chains: measured and dropped, profiles/README.md "Dropped experiments" #2)
    static_for<R>([&](auto ic_) {
        constexpr int i = decltype(ic_)::value;
        constexpr int t = [](int v) { int r = 0; for (int bb = 0; bb < LOG_R; ++bb) r |= ((v >> bb) & 1) << (LOG_R - 1 - bb); return r; }(i);
        srow_landed(a.x);                                                // this row is here ...
        uint4 na, nb;
        if (i + 1 < R) { na = rec[2 * row_of(i + 1)]; nb = rec[2 * row_of(i + 1) + 1]; }   // ... the next one on its way
        __builtin_amdgcn_sched_barrier(0);
        u32 best = 0x7FFFFFu;                                         // a row the chunk does not have: loses every minimum
        if (!FLEX || t < nr) {                                           // wave-uniform
            const u32 w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            u32 h[NJ];
            ham8_cols<NJ>(q, w, h);                                    // NJ accumulator chains, order pinned
#pragma unroll
            for (int j = 0; j < NJ; j += 2) {
                const u32 k0 = shl7_u16(h[j]) | (u32)(t * 8 + j), k1 = shl7_u16(h[j + 1]) | (u32)(t * 8 + j + 1);
                cb16[j] = min_u16(cb16[j], k0);                        // best row of column j
                cb16[j + 1] = min_u16(cb16[j + 1], k1);
                best = j == 0 ? min_u16(k0, k1) : min_u16(best, min_u16(k0, k1));   // best column of this row
            }
        }
        u32 v = (best << 9) | (u32)lane;
        // Node of level k joins rows t and t + R / 2^(k+1).  The four levels split on lane bits 2, 3 (DPP minima, two instructions
        // a node: the levels with 8 and 4 nodes), 4 and 5 (permlane swap + minimum), and the two lane bits left over are reduced
        // on the one value at the end: 32 instead of ~66 instructions per chunk and no select masks to keep (rounds 1-3: levels on
        // bits 3, 2, 1, 0 with select nodes -- two v_cndmask, a DPP move and a minimum each).  4-stream run +1.8 % (6 836 -> 6 958
        // frames/s, interleaved), and the kernel fits its 104 registers without scratch.  Lane l ends with row 8 b2 + 4 b3 + 2 b4 + b5.
        static_assert(R == 16, "the level -> lane bit assignment below is for 16-row chunks");
        auto node = [&](auto level, u32 lo, u32 hi_) -> u32 {
            constexpr int k = decltype(level)::value;
            if constexpr (k == 0) return bfly_dpp<4>(lo, hi_);
            else if constexpr (k == 1) return bfly_dpp<8>(lo, hi_);
            else if constexpr (k == 2) return bfly<16>(lo, hi_, lane);
            else return bfly<32>(lo, hi_, lane);
        };
        if constexpr ((i & 1) == 0) stk[0] = v;
        else {
            v = node(std::integral_constant<int, 0>{}, stk[0], v);
            if constexpr ((i & 2) == 0 || LOG_R < 2) stk[1] = v;
            else {
                v = node(std::integral_constant<int, 1>{}, stk[1], v);
                if constexpr (LOG_R >= 3) {
                    if constexpr ((i & 4) == 0) stk[2] = v;
                    else {
                        v = node(std::integral_constant<int, 2>{}, stk[2], v);
                        if constexpr (LOG_R >= 4) {
                            if constexpr ((i & 8) == 0) stk[3] = v;
                            else stk[4] = node(std::integral_constant<int, 3>{}, stk[3], v);
                        } else stk[3] = v;
                    }
                } else stk[2] = v;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (i + 1 < R) { a = na; b = nb; }
    });
    // lane bits 0, 1: every lane now holds the minimum of ITS row; one lane per row writes
    const u32 m = allmin_quad(stk[LOG_R]);
    const int row_in_chunk = ((lane >> 2) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 4) & 1) * 2 + (lane >> 5);
    const bool writer = (lane & 3) == 0;
    const int row = tc + row_in_chunk;
    if (writer && row_in_chunk < (FLEX ? nr : R)) {
        const u32 key = (m & 0xFFFF0000u) | (colbase + ((m >> 9) & 7u) * 64u + (m & 63u));   // distance << 16 | column
        if (single_cb) rowkey[row] = key;
        else atomicMin(&rowkey[row], key);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const u32 k16 = cb16[j];
        const u32 key = ((k16 >> 7) << 16) | (u32)(tc + (int)((k16 >> 3) & 15u));
        atomicMin(&colbest[colbase + j * 64 + lane], key);
    }
}

// grid: any; block: 256 (4 waves).  Dynamic LDS: (ncb*64*NJ + max_rows + 16) * 4 bytes.
// NJ = columns per lane.  A wave covers a "column block" of 64*NJ current descriptors.  When the
// number of column blocks ncb divides 4, wave w is bound to block w % ncb for the whole launch (its
// descriptors stay in registers) and shares a record's rows with the other waves bound to that block;
// otherwise every wave walks all column blocks and reloads its registers per block.
// mask.xyh != NULL: heading-incompatible records are not scored (count 0), see ScanMask.
// NJ = 8 is the working point (500 descriptors are one block, 64 VGPRs of descriptors, 4 waves per SIMD);
// NJ = 4 / 2 serve calls with at most 256 / 128 current descriptors, see k_db_scan.  (8 waves per SIMD -- NJ = 4 with two blocks,
// or the descriptors in LDS -- measured slower: profiles/README.md "Dropped experiments" #3.)
template <int NJ, bool EMIT, int NW = 4>
__device__ __forceinline__ void db_scan_body(
    u32 *lds, int C, const uint4 *__restrict__ db, const int64_t *__restrict__ off, const int32_t *__restrict__ rec_ids,
    const int32_t *__restrict__ n_ids_p, int n_ids_max, const uint4 *__restrict__ cur, int max_rows,
    int32_t *__restrict__ counts, int32_t *__restrict__ m_qidx, int32_t *__restrict__ m_tidx, int32_t *__restrict__ m_dist,
    int32_t *__restrict__ m_n, int emit_stride, const ScanMask &mask, u32 *ticket, int quota, u32 *ticket_pool = nullptr,
    int pool_frames = 1, int block = -1, int n_blocks = -1)
{
    // ticket: the 8 per-XCD record counters of THIS scan; ticket_pool / pool_frames: all counters of the launch (a batched
    // launch scans pool_frames frames, frame f owning ticket_pool + f * 8 * TICKET_STRIDE; the word behind them counts
    // the workgroups that have left).  block / n_blocks: this workgroup's index and the number of workgroups of ITS scan
    // (static deal; default = the launch's).
    if (!ticket_pool) ticket_pool = ticket;
    if (block < 0) { block = blockIdx.x; n_blocks = gridDim.x; }
    constexpr int CB = 64 * NJ;           // columns per block
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: row fetches stay scalar
    const int n_ids = n_ids_p ? min(*n_ids_p, n_ids_max) : n_ids_max;
    const int ncb = max((C + CB - 1) / CB, 1);
    u32 *colbest = lds;                   // ncb * CB : best (distance << 16 | row) per column
    u32 *rowkey = colbest + ncb * CB;     // max_rows : best (distance << 16 | column) per row
    u32 *wsum = rowkey + max_rows;        // 16

    u32 q[NJ][8];
    auto load_q = [&](int colbase) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int col = min(colbase + j * 64 + lane, max(C - 1, 0));   // padding repeats the last column
            const uint4 a = cur[2 * col], b = cur[2 * col + 1];
            q[j][0] = a.x; q[j][1] = a.y; q[j][2] = a.z; q[j][3] = a.w;
            q[j][4] = b.x; q[j][5] = b.y; q[j][6] = b.z; q[j][7] = b.w;
        }
    };
    static_assert(NW == 1 || NW == 2 || NW == 4 || NW == 8, "wsum[8] is the ticket slot");
    const bool bound = (NW % ncb) == 0;           // ncb divides the wave count: static wave -> column block binding
    const int my_cb = bound ? wave % ncb : 0;
    const int chunk0 = bound ? wave / ncb : wave, chunk_step = bound ? NW / ncb : NW;
    // RELOC_TICK_AUTO: this scan stands down (scan-uniform).  Its workgroups still check out at the end, so that the last
    // workgroup of the launch can put the counters back
    const bool stand_down = mask.skip_if && *mask.skip_if != 0;
    if (bound && !stand_down) load_q(my_cb * CB);
    double hc = 1.0, hs = 0.0, cos_tol = 0.0;
    if (mask.xyh) {
        cur_heading_q(mask.q, hc, hs);
        cos_tol = mask.cos_tol;
    }

    // Work distribution.  ticket == NULL: record it = blockIdx.x, + gridDim.x, ... (static).  Otherwise the grid is one
    // resident generation and every workgroup draws records from counters, so that all CUs stay full until the last
    // record (with a static deal the workgroups of a CU finish one after the other and the CU's tail runs at 3, 2, 1
    // waves per SIMD).  One counter serves ~88 draws per microsecond (measured: 10 000 draws on one word = 143 us), so
    // there are 8, one per XCD (HW_REG_XCC_ID) on its own 128-byte line: counter x deals records x, x + 8, x + 16, ...;
    // a workgroup whose counter has run dry moves on to the next one.  The draw for the next record is in flight
    // while the current one is processed.  The last workgroup to leave (ticket[TICKET_DONE]) zeroes all words.
    constexpr int TICKET_STRIDE = 32;
    int shard = 0, dry = 0;
    if (ticket) {
        u32 x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
        shard = (int)(x & 7u);
    }
    auto draw = [&]() -> u32 { return atomicAdd(&ticket[shard * TICKET_STRIDE], 1u); };
    // thread 0: turn a drawn ticket into a record index, moving to the next counter while the current one is dry
    auto settle = [&](u32 t) -> int {
        for (;;) {
            const long long rec_i = (long long)shard + 8ll * (long long)t;
            if (rec_i < n_ids) return (int)rec_i;
            if (++dry >= 8) return n_ids;
            shard = (shard + 1) & 7;
            t = draw();
        }
    };
    int it = block;
    if (stand_down) it = n_ids;
    else if (ticket) {
        if (tid == 0) wsum[8] = (u32)settle(draw());
        __syncthreads();
        it = __builtin_amdgcn_readfirstlane((int)wsum[8]);
    }
    // quota: a workgroup leaves after that many records although tickets remain (the grid then holds several
    // generations): its slot goes to whatever waits -- with several contexts sharing the chip that is another stream's
    // ORB / PnP kernel, which otherwise would not get a CU until this whole scan has drained
    int left = quota > 0 ? quota : 0x7fffffff;
    for (; it < n_ids;) {
        --left;
        u32 next_ticket = 0;
        if (ticket && tid == 0 && left > 0) next_ticket = draw();        // no draw that this workgroup would not serve
        auto advance = [&]() {
            if (ticket) {
                __syncthreads();
                if (tid == 0) wsum[8] = left > 0 ? (u32)settle(next_ticket) : (u32)n_ids;
                __syncthreads();
                it = __builtin_amdgcn_readfirstlane((int)wsum[8]);
            } else {
                it += n_blocks;
            }
        };
        const int r = rec_ids ? rec_ids[it] : it;
        if (mask.xyh && !heading_ok(mask.xyh + 4 * (int64_t)r, hc, hs, cos_tol)) {      // workgroup-uniform
            if (tid == 0 && counts) counts[EMIT ? it : r] = 0;
            if (EMIT && tid == 0 && m_n) m_n[it] = 0;
            advance();
            continue;
        }
        const int64_t row0 = off[r];
        const int n = (int)(off[r + 1] - row0);
        const uint4 *rec = db + 2 * row0;
        for (int i = tid; i < ncb * CB; i += 64 * NW) colbest[i] = 0xFFFFFFFFu;
        for (int i = tid; i < n; i += 64 * NW) rowkey[i] = 0xFFFFFFFFu;
        __syncthreads();
        if (n > 0 && C > 0) {
            for (int cb = bound ? my_cb : 0; cb < (bound ? my_cb + 1 : ncb); ++cb) {
                if (!bound) load_q(cb * CB);
                // The record's rows are dealt to the waves bound to this column block as contiguous ranges, balanced to
                // the single row; a wave walks its range in 16-row chunks and finishes it with ONE flexible chunk of
                // 1-15 rows (n = 64, 4 waves: one 16-row chunk each; n = 45: 12/11/11/11 rows = one flexible chunk each
                // -- round 2 walked tails in 4-row chunks and paid the chunk epilogue, butterfly + 9 LDS minima, every 4
                // rows: 1.56 T pairs/s on 45-row records against 2.0 T on 64-row ones).
                const int per = n / chunk_step, extra = n % chunk_step;
                int tc = chunk0 * per + min(chunk0, extra);
                const int tend = tc + per + (chunk0 < extra ? 1 : 0);
                // ONE instantiation serves full and partial chunks: the unrolled 16-row body is 45 KB of code, two of them
                // would not share the 64 KB instruction cache
#pragma nounroll
                for (; tc < tend; tc += 16)
                    scan_chunk<NJ, 16, true>(rec, n, tc, min(16, tend - tc), q, (u32)(cb * CB), rowkey, colbest, ncb == 1, lane);
            }
        }
        __syncthreads();
        // mutual resolution, in teach-row (queryIdx) order
        u32 base = 0;
        for (int rb = 0; rb < n; rb += 64 * NW) {
            const int row = rb + tid;
            bool mutual = false;
            u32 key = 0;
            if (row < n && C > 0) {
                key = rowkey[row];
                const u32 col = key & 0xFFFFu;
                mutual = (colbest[col] & 0xFFFFu) == (u32)row;
            }
            const unsigned long long bal = __ballot(mutual);
            if (lane == 0) wsum[wave] = (u32)__popcll(bal);
            __syncthreads();
            u32 before = base;
            for (int w = 0; w < wave; ++w) before += wsum[w];
            u32 total = 0;
            for (int w = 0; w < NW; ++w) total += wsum[w];
            if (EMIT && mutual) {
                const u32 pos = before + (u32)__popcll(bal & ((1ull << lane) - 1ull));
                const int64_t o = (int64_t)it * emit_stride + pos;
                m_qidx[o] = row;
                m_tidx[o] = (int32_t)(key & 0xFFFFu);
                m_dist[o] = (int32_t)(key >> 16);
                if (mask.g_obj) {
                    const float *p3 = mask.g_pts3d + 3 * (row0 + row), *p2 = mask.g_xy + 2 * (size_t)(key & 0xFFFFu);
                    float *po = mask.g_obj + 3 * o, *pi = mask.g_img + 2 * o;
                    po[0] = p3[0]; po[1] = p3[1]; po[2] = p3[2];
                    pi[0] = p2[0]; pi[1] = p2[1];
                }
            }
            base += total;
            __syncthreads();
        }
        if (tid == 0) {
            if (counts) counts[EMIT ? it : r] = (int32_t)base;
            if (EMIT && m_n) m_n[it] = (int32_t)base;
        }
        advance();
    }
    if (ticket && tid == 0) {
        const int words = pool_frames * 8;
        if (atomicAdd(&ticket_pool[words * TICKET_STRIDE], 1u) == gridDim.x - 1) {      // every other workgroup has made its last draw
            for (int x = 0; x <= words; ++x)
                __hip_atomic_store(&ticket_pool[x * TICKET_STRIDE], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// The counting scan (no match lists): the per-record flow of db_scan_body reduced to TWO barriers per record.
//   - colbest is double-buffered: the buffer of the NEXT record is cleared while this record's chunks run, so no
//     clear / barrier pair stands in front of a record;
//   - with one column block (<= 64 * NJ current descriptors: every fused tick) each row key is written by exactly one
//     wave with a plain store and never needs clearing;
//   - the mutual pairs are only counted: per wave one ballot and one LDS add, no prefix sums;
//   - the next record's ticket (drawn a record ahead) is settled between the same two barriers.
// A workgroup's FIRST record is dealt statically (record = workgroup index, no counter round trip in front of the first
// row); the counters deal the records behind the grid.  Round 2's flow had six barriers per record and waited for an
// atomic before the first row: 10 000 x 64 in three generations 162 -> [see DESIGN.md] us.
template <int NJ, int NW>
__device__ __forceinline__ void db_count_body(
    u32 *lds, int C, const uint4 *__restrict__ db, const int64_t *__restrict__ off, const int32_t *__restrict__ rec_ids,
    const int32_t *__restrict__ n_ids_p, int n_ids_max, const uint4 *__restrict__ cur, int max_rows,
    int32_t *__restrict__ counts, const ScanMask &mask, u32 *ticket, int quota, int n_bounded, int col_words,
    u32 *ticket_pool = nullptr, int pool_frames = 1, int block = -1, int n_blocks = -1)
{
    // quota / n_bounded: the first n_bounded workgroups of a scan leave once they have served `quota` ROWS (their slots go to
    // whatever waits: with several contexts on the chip another stream's ORB / PnP kernel); their budgets add up to the whole
    // database, so they normally serve every record.  The eight workgroups behind them (one per XCD, the last to start) draw
    // until the counters run dry: whatever the budgets left over is served (nothing, when the host knew the row total).
    // (n_bounded < 0: the record quota of rounds 2-3a, RELOC_SCAN_QUOTA_ROWS=0; numbers and the variants dropped on the way:
    // profiles/README.md "Dropped experiments" #4)
    if (!ticket_pool) ticket_pool = ticket;
    if (block < 0) { block = blockIdx.x; n_blocks = gridDim.x; }
    constexpr int CB = 64 * NJ;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_ids = n_ids_p ? min(*n_ids_p, n_ids_max) : n_ids_max;
    const int ncb = max((C + CB - 1) / CB, 1);
    // col_words words of column minima (best distance << 16 | row per column), sized by the host from the CAPACITY of the
    // current-descriptor buffer (one buffer's worth, at least two column blocks); the kernel knows the COUNT: when two buffers
    // of ncb blocks fit they are double-buffered (the next record's buffer is cleared while this record's chunks run: two
    // barriers per record), otherwise one buffer is cleared between the records (three barriers).  A tick's ~500 features in
    // a context made for 8192 are one block: double-buffered in 33 KB, where sizing two buffers for the capacity took 66 KB
    // and left room for two workgroups per CU instead of four.
    const bool dbl = 2 * ncb * CB <= col_words;                                // workgroup-uniform
    u32 *colbuf = lds;
    u32 *rowkey = colbuf + col_words;         // max_rows     : best (distance << 16 | column) per row
    u32 *wsum = rowkey + max_rows;            // [0], [1]: mutual-pair counters of the two buffers; [8]: next record

    u32 q[NJ][8];
    auto load_q = [&](int colbase) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int col = min(colbase + j * 64 + lane, max(C - 1, 0));   // padding repeats the last column
            const uint4 a = cur[2 * col], b = cur[2 * col + 1];
            q[j][0] = a.x; q[j][1] = a.y; q[j][2] = a.z; q[j][3] = a.w;
            q[j][4] = b.x; q[j][5] = b.y; q[j][6] = b.z; q[j][7] = b.w;
        }
    };
    const bool bound = (NW % ncb) == 0;
    const int my_cb = bound ? wave % ncb : 0;
    const int chunk0 = bound ? wave / ncb : wave, chunk_step = bound ? NW / ncb : NW;
    const bool stand_down = mask.skip_if && *mask.skip_if != 0;
    if (bound && !stand_down) load_q(my_cb * CB);
    double hc = 1.0, hs = 0.0, cos_tol = 0.0;
    if (mask.xyh) {
        cur_heading_q(mask.q, hc, hs);
        cos_tol = mask.cos_tol;
        // wave-uniform values computed on the vector unit: moved to scalar registers (they live across the whole launch, and
        // the kernel is capped at 104 vector registers)
        hc = uniform_f64(hc); hs = uniform_f64(hs); cos_tol = uniform_f64(cos_tol);
    }
    constexpr int TICKET_STRIDE = 32;
    int shard = 0, dry = 0;
    if (ticket) {
        u32 x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
        shard = (int)(x & 7u);
    }
    auto draw = [&]() -> u32 { return atomicAdd(&ticket[shard * TICKET_STRIDE], 1u); };
    // thread 0: ticket -> record index; counter x deals records n_blocks + x, n_blocks + x + 8, ... (the first n_blocks
    // records are the workgroups' static first records)
    auto settle = [&](u32 t) -> int {
        for (;;) {
            const long long rec_i = (long long)n_blocks + (long long)shard + 8ll * (long long)t;
            if (rec_i < n_ids) return (int)rec_i;
            if (++dry >= 8) return n_ids;
            shard = (shard + 1) & 7;
            t = draw();
        }
    };
    int it = stand_down ? n_ids : block;
    const bool by_rows = n_bounded >= 0;                                      // n_bounded < 0: a quota of RECORDS for every workgroup
    int left = (quota > 0 && (!by_rows || block < n_bounded)) ? quota : 0x7fffffff;   // rows (records) this workgroup may still take on
    for (int i = tid; i < ncb * CB; i += 64 * NW) colbuf[i] = 0xFFFFFFFFu;
    if (tid == 0) wsum[0] = wsum[1] = 0;
    __syncthreads();
    int p = 0;
    while (it < n_ids) {
        u32 *colbest = colbuf + (dbl ? p * ncb * CB : 0);
        const int r = rec_ids ? rec_ids[it] : it;
        const bool scored = !(mask.xyh && !heading_ok(mask.xyh + 4 * (int64_t)r, hc, hs, cos_tol));      // workgroup-uniform
        const int64_t row0 = off[r];
        const int n = scored ? (int)(off[r + 1] - row0) : 0;
        const uint4 *rec = db + 2 * row0;
        left -= by_rows ? max(n, 1) : 1;
        u32 next_ticket = 0;
        if (ticket && tid == 0 && left > 0) next_ticket = draw();        // no draw that this workgroup would not serve
        if (n > 0 && C > 0) {
            if (ncb > 1) {                                               // several waves write one row's key: atomic minima
                for (int i = tid; i < n; i += 64 * NW) rowkey[i] = 0xFFFFFFFFu;
                __syncthreads();
            }
            for (int cb = bound ? my_cb : 0; cb < (bound ? my_cb + 1 : ncb); ++cb) {
                if (!bound) load_q(cb * CB);
                // rows dealt to the waves of this column block as contiguous ranges balanced to the single row; 16-row
                // chunks, the last one partial (see db_scan_body)
                const int per = n / chunk_step, extra = n % chunk_step;
                int tc = chunk0 * per + min(chunk0, extra);
                const int tend = tc + per + (chunk0 < extra ? 1 : 0);
#pragma nounroll
                for (; tc < tend; tc += 16)
                    scan_chunk<NJ, 16, true>(rec, n, tc, min(16, tend - tc), q, (u32)(cb * CB), rowkey, colbest, ncb == 1, lane);
            }
        }
        if (dbl) {   // the other buffer, for the next record
            u32 *other = colbuf + (p ^ 1) * ncb * CB;
            for (int i = tid; i < ncb * CB; i += 64 * NW) other[i] = 0xFFFFFFFFu;
        }
        __syncthreads();                                                  // every minimum of this record is in LDS
        if (C > 0) {
            for (int rb = 0; rb < n; rb += 64 * NW) {                     // wave-uniform trip count
                const int row = rb + tid;
                bool mutual = false;
                if (row < n) {
                    const u32 col = rowkey[row] & 0xFFFFu;
                    mutual = (colbest[col] & 0xFFFFu) == (u32)row;
                }
                const u32 c = (u32)__popcll(__ballot(mutual));
                if (lane == 0 && c) atomicAdd(&wsum[p], c);
            }
        }
        if (tid == 0) wsum[8] = ticket ? (left > 0 ? (u32)settle(next_ticket) : (u32)n_ids) : (u32)(it + n_blocks);
        __syncthreads();
        if (tid == 0) {
            counts[r] = (int32_t)wsum[p];
            wsum[p] = 0;
        }
        it = __builtin_amdgcn_readfirstlane((int)wsum[8]);
        if (dbl) p ^= 1;
        else {       // one buffer: cleared between the records
            for (int i = tid; i < ncb * CB; i += 64 * NW) colbuf[i] = 0xFFFFFFFFu;
            __syncthreads();
        }
    }
    if (ticket && tid == 0) {
        const int words = pool_frames * 8;
        if (atomicAdd(&ticket_pool[words * TICKET_STRIDE], 1u) == gridDim.x - 1) {      // every other workgroup has made its last draw
            for (int x = 0; x <= words; ++x)
                __hip_atomic_store(&ticket_pool[x * TICKET_STRIDE], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// The kernel proper.  NJ is chosen by the host from the CAPACITY of the current-descriptor buffer: entry points
// that know the query count (reloc_db_match_counts*, reloc_match_mutual) scan 128 or 256 columns per wave when
// that is enough, so their cost follows the query count instead of being flat below 512; the fused tick passes
// its feature capacity and always runs NJ = 8.  (One kernel branching on the device-side count was measured:
// it costs the NJ = 8 path 3 %.)
template <int NJ, bool EMIT, int NW>
__global__ __launch_bounds__(64 * NW, NW >= 4 ? 16 / NW : 4) RELOC_SCAN_VGPR_ATTR void k_db_scan(
    const uint4 *__restrict__ db, const int64_t *__restrict__ off, const int32_t *__restrict__ rec_ids,
    const int32_t *__restrict__ n_ids_p, int n_ids_max, const uint4 *__restrict__ cur,
    const int32_t *__restrict__ n_cur_p, int n_cur_max, int max_rows, int32_t *__restrict__ counts,
    int32_t *__restrict__ m_qidx, int32_t *__restrict__ m_tidx, int32_t *__restrict__ m_dist,
    int32_t *__restrict__ m_n, int emit_stride, ScanMask mask, u32 *ticket, int quota, int n_bounded, int col_words)
{
    extern __shared__ u32 lds[];
    if constexpr (EMIT) RELOC_SMALL_KERNEL_PRIO();          // the emit pass of a few candidates is one of the tick's small kernels
    const int C = n_cur_p ? min(*n_cur_p, n_cur_max) : n_cur_max;
    if constexpr (EMIT)
        db_scan_body<NJ, true, NW>(lds, C, db, off, rec_ids, n_ids_p, n_ids_max, cur, max_rows, counts, m_qidx, m_tidx, m_dist, m_n,
                                   emit_stride, mask, ticket, quota);
    else
        db_count_body<NJ, NW>(lds, C, db, off, rec_ids, n_ids_p, n_ids_max, cur, max_rows, counts, mask, ticket, quota, n_bounded, col_words);
}

// The emit pass (match lists of the candidate records, M:327-336) of up to 8 frames in one launch: blockIdx.y = frame.
struct EmitFrame {
    const int32_t *rec_ids, *n_ids_p; const uint4 *cur; const int32_t *n_cur_p;
    int32_t *m_qidx, *m_tidx, *m_dist, *m_n; const float *g_xy; float *g_obj, *g_img;
};
struct EmitBatch { EmitFrame f[RELOC_BATCH_MAX]; };

__global__ __launch_bounds__(256, 4) RELOC_SCAN_VGPR_ATTR void k_db_emit_batch(const uint4 *__restrict__ db, const int64_t *__restrict__ off, int n_ids_max,
                                                          int n_cur_max, int max_rows, int emit_stride, const float *__restrict__ g_pts3d,
                                                          EmitBatch bt)
{
    extern __shared__ u32 lds[];
    RELOC_SMALL_KERNEL_PRIO();
    const EmitFrame &F = bt.f[blockIdx.y];
    ScanMask mask;
    mask.xyh = nullptr;
    mask.q[0] = mask.q[1] = mask.q[2] = 0; mask.q[3] = 1;
    mask.g_pts3d = g_pts3d; mask.g_xy = F.g_xy; mask.g_obj = F.g_obj; mask.g_img = F.g_img;
    const int C = F.n_cur_p ? min(*F.n_cur_p, n_cur_max) : n_cur_max;
    db_scan_body<8, true, 4>(lds, C, db, off, F.rec_ids, F.n_ids_p, n_ids_max, F.cur, max_rows, nullptr, F.m_qidx, F.m_tidx, F.m_dist, F.m_n,
                             emit_stride, mask, nullptr, 0);
}

int launch_db_emit_batch(reloc_ctx *const *ctxs, int n)
{
    reloc_ctx *c0 = ctxs[0];
    if (n < 1 || n > RELOC_BATCH_MAX) { reloc_set_error("emit batch: 1..%d frames", RELOC_BATCH_MAX); return RELOC_E_ARG; }
    if (c0->max_feat > 65535 || c0->db_max_rows > MAX_REC_ROWS) { reloc_set_error("emit batch: capacity"); return RELOC_E_CAPACITY; }
    EmitBatch bt;
    for (int f = 0; f < RELOC_BATCH_MAX; ++f) {
        reloc_ctx *c = ctxs[f < n ? f : 0];
        EmitFrame &F = bt.f[f];
        F.rec_ids = c->cand_ids; F.n_ids_p = c->cand_n; F.cur = (const uint4 *)c->f_desc; F.n_cur_p = c->f_count;
        F.m_qidx = c->m_qidx; F.m_tidx = c->m_tidx; F.m_dist = c->m_dist; F.m_n = c->m_n; F.g_xy = c->f_xy; F.g_obj = c->p_obj; F.g_img = c->p_img;
    }
    const int max_rows = c0->db_max_rows < 1 ? 1 : c0->db_max_rows;
    const int ncb = (c0->max_feat + 511) / 512;
    const size_t lds = (size_t)(ncb * 512 + max_rows + 16) * 4;
    if (lds > 160 * 1024) { reloc_set_error("emit batch: LDS demand %zu bytes", lds); return RELOC_E_CAPACITY; }
    hipLaunchKernelGGL(k_db_emit_batch, dim3(MAX_CAND, n), dim3(256), lds, c0->stream, (const uint4 *)c0->db_desc, c0->db_off, MAX_CAND,
                       c0->max_feat, max_rows, MAX_REC_ROWS, c0->db_pts3d, bt);
    HIP_TRY(hipGetLastError());
    return RELOC_OK;
}

// Several frames in ONE launch (BASELINE.json config 4: batched relocalization): workgroup b scans frame b % B -- its
// current descriptors, feature count, counts array, heading and ticket counters -- so B whole-database scans share one
// launch: the launch-fixed cost (descriptor prologue per workgroup, last-record tail, kernel boundary) is paid once per
// batch, and the deal stays dynamic per frame.  Frames whose local search found candidates (AUTO mode) stand down alone.
constexpr int SCAN_BATCH_MAX = 8;
struct ScanBatch {
    int n;
    const uint4 *cur[SCAN_BATCH_MAX];
    const int32_t *n_cur[SCAN_BATCH_MAX];
    int32_t *counts[SCAN_BATCH_MAX];
    const int32_t *skip_if[SCAN_BATCH_MAX];
    double q[SCAN_BATCH_MAX][4];
    const double *xyh;
    double cos_tol;
};

__global__ __launch_bounds__(256, 4) RELOC_SCAN_VGPR_ATTR void k_db_scan_batch(const uint4 *__restrict__ db, const int64_t *__restrict__ off, int n_ids,
                                                          int n_cur_max, int max_rows, ScanBatch bt, u32 *ticket_pool, int quota, int n_bounded, int col_words)
{
    extern __shared__ u32 lds[];
    const int f = blockIdx.x % bt.n;                       // workgroup-uniform
    ScanMask mask;
    mask.xyh = bt.xyh;
    for (int k = 0; k < 4; ++k) mask.q[k] = bt.q[f][k];
    mask.cos_tol = bt.cos_tol;
    mask.skip_if = bt.skip_if[f];
    const int C = bt.n_cur[f] ? min(*bt.n_cur[f], n_cur_max) : n_cur_max;
    db_count_body<8, 4>(lds, C, db, off, nullptr, nullptr, n_ids, bt.cur[f], max_rows, bt.counts[f], mask, ticket_pool + f * 8 * 32, quota,
                        n_bounded, col_words, ticket_pool, bt.n, (int)(blockIdx.x / bt.n), (int)((gridDim.x + bt.n - 1 - f) / bt.n));
}

// ---------------------------------------------------------------------------------------------
// Ratio-test score of every record (reference _archive/anchor_localizer.py:82-90,
// checkpoint_a_selftest.py:68-71): per record, the number of current descriptors whose two nearest rows
// of the record satisfy d1 < ratio * d2.  Same data flow as k_db_scan (512 current descriptors per wave
// in registers, teach rows through the scalar cache) but the epilogue only tracks the two smallest
// distances per column: v_max_u16 + 2 v_min_u16 per pair, no indices, no cross-lane work in the loop.
__device__ __forceinline__ u32 max_u16(u32 a, u32 b)
{
    u32 r;
    asm("v_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__global__ __launch_bounds__(256, 4) void k_db_ratio(const uint4 *__restrict__ db, const int64_t *__restrict__ off, int n_rec,
                                                     const uint4 *__restrict__ cur, const int32_t *__restrict__ n_cur_p,
                                                     int n_cur_max, double ratio, int32_t *__restrict__ counts)
{
    __shared__ u32 s_part[4][512];
    __shared__ int s_cnt[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = n_cur_p ? min(*n_cur_p, n_cur_max) : n_cur_max;
    const int ncb = max((C + 511) >> 9, 1);
    u32 q[8][8];
    auto load_q = [&](int colbase) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int col = min(colbase + j * 64 + lane, max(C - 1, 0));
            const uint4 a = cur[2 * col], b = cur[2 * col + 1];
            q[j][0] = a.x; q[j][1] = a.y; q[j][2] = a.z; q[j][3] = a.w;
            q[j][4] = b.x; q[j][5] = b.y; q[j][6] = b.z; q[j][7] = b.w;
        }
    };
    if (ncb == 1 && C > 0) load_q(0);
    for (int r = blockIdx.x; r < n_rec; r += gridDim.x) {
        const int64_t row0 = off[r];
        const int n = (int)(off[r + 1] - row0);
        const uint4 *rec = db + 2 * row0;
        int good = 0;
        if (n >= 2 && C > 0) {
            for (int cb = 0; cb < ncb; ++cb) {
                if (ncb > 1) load_q(cb * 512);
                u32 k0[8], k1[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { k0[j] = 0xFFFFu; k1[j] = 0xFFFFu; }
                for (int t0 = wave * 8; t0 < n; t0 += 32) {
                    // rows through the scalar cache, one fetch in flight (see srow_landed); rows past the end
                    // re-read the last row and are skipped below, so a duplicate never counts twice
                    auto row_of = [&](int t) { return min(t0 + t, n - 1); };
                    uint4 a = rec[2 * row_of(0)], b = rec[2 * row_of(0) + 1];
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        srow_landed(a.x);
                        uint4 na, nb;
                        if (t + 1 < 8) { na = rec[2 * row_of(t + 1)]; nb = rec[2 * row_of(t + 1) + 1]; }
                        __builtin_amdgcn_sched_barrier(0);
                        if (t0 + t < n) {                                // wave-uniform
#pragma unroll
                            for (int j = 0; j < 8; j += 2) {
                                u32 h0, h1;
                                ham8x2(q[j], q[j + 1], a, b, 0, 0, h0, h1);
                                const u32 m0 = max_u16(k0[j], h0), m1 = max_u16(k0[j + 1], h1);
                                k0[j] = min_u16(k0[j], h0); k0[j + 1] = min_u16(k0[j + 1], h1);
                                k1[j] = min_u16(k1[j], m0); k1[j + 1] = min_u16(k1[j + 1], m1);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (t + 1 < 8) { a = na; b = nb; }
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) s_part[wave][j * 64 + lane] = (k1[j] << 16) | k0[j];
                __syncthreads();
                for (int c = tid; c < 512; c += 256) {
                    u32 a0 = 0xFFFFu, a1 = 0xFFFFu;
                    for (int w = 0; w < 4; ++w) {
                        const u32 v = s_part[w][c];
                        const u32 lo = v & 0xFFFFu, hi = v >> 16;
                        // merge two sorted pairs: smallest two of {a0, a1, lo, hi}
                        const u32 n0 = a0 < lo ? a0 : lo;
                        const u32 n1 = a0 < lo ? (a1 < lo ? a1 : lo) : (hi < a0 ? hi : a0);
                        a0 = n0; a1 = n1;
                    }
                    const bool ok = cb * 512 + c < C && a1 != 0xFFFFu && (double)a0 < ratio * (double)a1;
                    good += __popcll(__ballot(ok));
                }
                __syncthreads();
            }
        }
        // every lane of a wave holds that wave's total; sum the four waves
        if (lane == 0) s_cnt[wave] = good;
        __syncthreads();
        if (tid == 0) counts[r] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Few current descriptors (C <= 64): the roles swap.  With one or a few dozen queries the column-per-lane kernel
// above wastes its 128 lanes x columns (Q = 1 and Q = 32 both cost the 128-column price, 55 us at 10 000 x 64); here a
// LANE is a teach row -- the wave reads 64 rows of the record straight from HBM (2 KB, coalesced), the queries are
// wave-uniform and arrive through the scalar cache -- so the work follows C and the kernel runs at the pace of the
// database read up to Q ~ 8.  One wave owns a record (no barriers); the SIMD's eight waves hide each other's HBM latency
// (one record per wave turn: taking two or four per turn lets the SIMD's waves fall into step, profiles/r3_small_q_records_per_turn.log).
// Records of more than SQ_MAX_ROWS rows, emit mode and the heading mask stay with k_db_scan.
constexpr int SQ_MAX_ROWS = 1024;
constexpr int SQ_WAVES = 4;

// ---- few-query scan, round 4 -------------------------------------------------------------------------------------------
// Round 4 rebuilt the kernel for fewer instructions per record (round 3's form: profiles/r3_*, git history):
//   (1) the distances of four queries run as EIGHT accumulator chains (query x descriptor half) in the pinned xor -> bcnt
//       order of ham8_cols (see there: a v_bcnt must not meet an accumulator written fewer than ~16 instructions earlier);
//       the two halves are merged and shifted by one v_add_lshl_u32;
//   (2) the butterfly that takes each query's best row over the 64 lanes is built from v_min_u16 WITH the DPP lane exchange
//       in the instruction (round 3: v_mov_dpp + two v_cndmask + v_min per node): a node that splits on lane bit 2 or 3 is two
//       bank-masked DPP minima, on bit 4 or 5 a v_permlane swap and a minimum, and the bits that are left are reduced on ONE
//       value -- 19 instead of ~36 instructions for 8 queries, 32 instead of ~70 for 16 (sq_bfly);
//   (3) the best row of every query lives in ONE register (lane sq_lane_of(q) holds query q: the butterfly leaves the
//       group's minima on every lane whose bits 2.. spell the query, so that lane takes its own), and a row's mutual test
//       fetches its query's entry with one ds_bpermute: no LDS arrays for records of up to 64 rows, one 16-bit LDS word
//       per row beyond that.
// (Requesting the next record's rows one record ahead: measured and dropped, profiles/README.md "Dropped experiments" #5.)
__device__ __forceinline__ u32 add_lshl6(u32 a, u32 b)
{
    u32 r;
    asm("v_add_lshl_u32 %0, %1, %2, 6" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// hs[j] = distance(query j, row) << 6 for four queries whose words are wave-uniform (SGPRs)
__device__ __forceinline__ void sq_dist4(const u32 (&qw)[4][8], const u32 (&w)[8], u32 (&hs)[4])
{
    u32 acc[4][2];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                u32 x;
                asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "s"(qw[j][4 * h + k]), "v"(w[4 * h + k]));
                if (k == 0) asm volatile("v_bcnt_u32_b32 %0, %1, 0" : "=v"(acc[j][h]) : "v"(x));
                else asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[j][h]) : "v"(x));
            }
#pragma unroll
    for (int j = 0; j < 4; ++j) hs[j] = add_lshl6(acc[j][0], acc[j][1]);
}

// four queries' words (32 SGPRs) by scalar loads at 32-bit byte offsets from the wave-uniform base; indices past the last
// query repeat it (a duplicate offers the same distance with a larger index: it never wins a minimum)
__device__ __forceinline__ void sq_load4(const uint4 *__restrict__ cur, int q0, int C, u32 (&qw)[4][8])
{
    const char *base = (const char *)cur;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const u32 o = (u32)min(q0 + jj, C - 1) << 5;
        const uint4 qa = *(const uint4 *)(base + o), qb = *(const uint4 *)(base + o + 16);
        qw[jj][0] = qa.x; qw[jj][1] = qa.y; qw[jj][2] = qa.z; qw[jj][3] = qa.w;
        qw[jj][4] = qb.x; qw[jj][5] = qb.y; qw[jj][6] = qb.z; qw[jj][7] = qb.w;
    }
}

// Butterfly minimum of G 16-bit keys per lane over the 64 lanes: afterwards EVERY lane l holds the minimum over all lanes of
// vector j(l) = (l >> 2) & (G - 1).  One asm statement: the order is fixed, so the wait states the hardware does not
// interlock are counted by hand (a VALU write of a register -> a DPP or v_permlane read of it needs 2 wait states: the
// producer is never closer than two instructions, or an s_nop 1 stands between; the opening s_nop covers the compiler's
// instruction in front of the statement).  Nodes:  lane bit 2: bank-masked row_shl:4 / row_shr:4;  bit 3: row_ror:8;
// bit 4 / 5: v_permlane16/32_swap (a half exchange: one operand's odd halves against the other's even halves) + minimum;
// bits left over are reduced on the one remaining value (quad_perm for bits 0 and 1; a swap with a copy for bits 4, 5).
#define SQ_N4(a, b)  "v_min_u16_dpp " a ", " a ", " a " row_shl:4 row_mask:0xf bank_mask:0x5\n\t" \
                     "v_min_u16_dpp " a ", " b ", " b " row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
#define SQ_N8(a, b)  "v_min_u16_dpp " a ", " a ", " a " row_ror:8 row_mask:0xf bank_mask:0x3\n\t" \
                     "v_min_u16_dpp " a ", " b ", " b " row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
#define SQ_Q1(a)     "v_min_u16_dpp " a ", " a ", " a " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define SQ_Q2(a)     "v_min_u16_dpp " a ", " a ", " a " quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
#define SQ_NOP       "s_nop 1\n\t"
template <int G>
__device__ __forceinline__ u32 sq_bfly(u32 (&d)[G])
{
    if constexpr (G == 16) {
        asm volatile(SQ_NOP
                     SQ_N4("%0", "%1") SQ_N4("%2", "%3") SQ_N4("%4", "%5") SQ_N4("%6", "%7")
                     SQ_N4("%8", "%9") SQ_N4("%10", "%11") SQ_N4("%12", "%13") SQ_N4("%14", "%15")
                     SQ_N8("%0", "%2") SQ_N8("%4", "%6") SQ_N8("%8", "%10") SQ_N8("%12", "%14")
                     "v_permlane16_swap_b32 %0, %4\n\t"              // %4 was written 5 instructions ago
                     "v_permlane16_swap_b32 %8, %12\n\t"             // %12: the instruction before last + one swap: pad
                     "v_min_u16 %0, %0, %4\n\t"
                     "v_min_u16 %8, %8, %12\n\t"
                     SQ_NOP
                     "v_permlane32_swap_b32 %0, %8\n\t"
                     "v_min_u16 %0, %0, %8\n\t"
                     SQ_NOP SQ_Q1("%0") SQ_NOP SQ_Q2("%0")
                     : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]), "+v"(d[8]),
                       "+v"(d[9]), "+v"(d[10]), "+v"(d[11]), "+v"(d[12]), "+v"(d[13]), "+v"(d[14]), "+v"(d[15]));
    } else if constexpr (G == 8) {
        asm volatile(SQ_NOP
                     SQ_N4("%0", "%1") SQ_N4("%2", "%3") SQ_N4("%4", "%5") SQ_N4("%6", "%7")
                     SQ_N8("%0", "%2") SQ_N8("%4", "%6")
                     SQ_NOP
                     "v_permlane16_swap_b32 %0, %4\n\t"
                     "v_min_u16 %0, %0, %4\n\t"
                     SQ_NOP SQ_Q1("%0") SQ_NOP SQ_Q2("%0")
                     "v_mov_b32 %4, %0\n\t"
                     SQ_NOP
                     "v_permlane32_swap_b32 %0, %4\n\t"
                     "v_min_u16 %0, %0, %4\n\t"
                     : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]));
    } else {
        static_assert(G == 4, "groups of 4, 8 or 16 queries");
        asm volatile(SQ_NOP
                     SQ_N4("%0", "%1") SQ_N4("%2", "%3")
                     SQ_NOP
                     SQ_N8("%0", "%2")
                     SQ_NOP SQ_Q1("%0") SQ_NOP SQ_Q2("%0")
                     "v_mov_b32 %2, %0\n\t"
                     SQ_NOP
                     "v_permlane16_swap_b32 %0, %2\n\t"
                     "v_min_u16 %0, %0, %2\n\t"
                     "v_mov_b32 %2, %0\n\t"
                     SQ_NOP
                     "v_permlane32_swap_b32 %0, %2\n\t"
                     "v_min_u16 %0, %0, %2\n\t"
                     : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));
    }
    return d[0];
}
#undef SQ_N4
#undef SQ_N8
#undef SQ_Q1
#undef SQ_Q2
#undef SQ_NOP
// the lane whose register holds query q's entry: bits 2.. = q's place in its group (what sq_bfly leaves there), the group
// number in the lane bits the butterfly reduced over
template <int G>
__device__ __forceinline__ u32 sq_lane_of(u32 q)
{
    if constexpr (G == 16) return ((q & 15u) << 2) | (q >> 4);                          // 4 groups: bits 0, 1
    else if constexpr (G == 8) return ((q & 7u) << 2) | ((q >> 3) & 3u) | (q & 32u);     // 8 groups: bits 0, 1, 5
    else return (q & 3u) << 2;
}

template <int G, bool HOIST>
__global__ __launch_bounds__(64 * SQ_WAVES) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_db_scan_rows(
    const uint4 *__restrict__ db, const int64_t *__restrict__ off, int n_rec, const uint4 *__restrict__ cur,
    const int32_t *__restrict__ n_cur_p, int n_cur_max, int32_t *__restrict__ counts)
{
    __shared__ unsigned short s_row[SQ_WAVES][SQ_MAX_ROWS];   // per row: distance << 6 | query of the best query (records > 64 rows)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int C = n_cur_p ? min(*n_cur_p, n_cur_max) : n_cur_max;
    unsigned short *rowk = s_row[wave];
    const int gw = blockIdx.x * SQ_WAVES + wave, nw = gridDim.x * SQ_WAVES;
    // the group whose entries this lane keeps (see sq_lane_of): the lane bits the butterfly reduces over
    const int my_group = G == 16 ? (lane & 3) : (G == 8 ? ((lane & 3) | ((lane >> 5) << 2)) : 0);
    constexpr int NH = HOIST ? G : 1;
    u32 qs[NH][8];
    if constexpr (HOIST) {
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int q = max(min(j, C - 1), 0);        // wave-uniform; repeats of the last query lose every tie (larger index)
            const uint4 qa = cur[2 * q], qb = cur[2 * q + 1];                // scalar loads, once
            qs[j][0] = qa.x; qs[j][1] = qa.y; qs[j][2] = qa.z; qs[j][3] = qa.w;
            qs[j][4] = qb.x; qs[j][5] = qb.y; qs[j][6] = qb.z; qs[j][7] = qb.w;
        }
    }
    // one 64-row chunk (rows tc .. tc + 63 in the lanes) against all queries: updates colk, returns the rows' best queries
    auto chunk = [&](const uint4 a, const uint4 b, int tc, u32 &colk) -> u32 {
        const u32 w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        u32 rbest = 0xFFFFu;
        for (int q0 = 0; q0 < (HOIST ? 1 : C); q0 += G) {
            u32 ck[G];
#pragma unroll
            for (int j4 = 0; j4 < G; j4 += 4) {
                u32 hs[4];
                if constexpr (HOIST) {
                    const u32 (&qv)[4][8] = *reinterpret_cast<const u32 (*)[4][8]>(&qs[j4 < NH ? j4 : 0]);
                    sq_dist4(qv, w, hs);
                } else {
                    u32 qw[4][8];
                    sq_load4(cur, q0 + j4, C, qw);
                    sq_dist4(qw, w, hs);
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    rbest = min_u16(rbest, hs[jj] | (u32)(q0 + j4 + jj));     // distance <= 256: (256 << 6 | 63) < 2^15
                    ck[j4 + jj] = hs[jj] | (u32)lane;
                }
            }
            const u32 m = sq_bfly<G>(ck);               // EVERY lane l: best (distance, lane) of query q0 + ((l >> 2) & (G - 1)) over the 64 rows
            const u32 key = ((m >> 6) << 16) | (u32)(tc + (int)(m & 63u));
            const u32 better = umin(colk, key);
            colk = my_group * G == q0 ? better : colk;                     // the lane of this group keeps the entry of its query
        }
        return rbest;
    };
    for (int r = gw; r < n_rec; r += nw) {
        const int64_t row0 = off[r];
        const int n = (int)(off[r + 1] - row0);
        if (n <= 0 || C <= 0) {
            if (lane == 0) counts[r] = 0;
            continue;
        }
        const uint4 *rec = db + 2 * row0;
        u32 colk = 0xFFFFFFFFu;                          // lane sq_lane_of(q): distance << 16 | row of query q's best row so far
        // mutual pairs: row t's best query must name t as its best row (lowest index on ties both ways)
        int total;
        {
            const int row = min(lane, n - 1);            // lanes past the end repeat the last row (larger lane index: loses every tie)
            const u32 rb = chunk(rec[2 * row], rec[2 * row + 1], 0, colk);
            if (n <= 64) {
                const u32 peer = (u32)__builtin_amdgcn_ds_bpermute((int)(sq_lane_of<G>(rb & 63u) << 2), (int)colk);
                total = __popcll(__ballot(lane < n && (peer & 0xFFFFu) == (u32)lane));
                if (lane == 0) counts[r] = total;
                continue;
            }
            rowk[lane] = (unsigned short)rb;
        }
        for (int tc = 64; tc < n; tc += 64) {                              // records of more than 64 rows: further chunks
            const int row = min(tc + lane, n - 1);
            const u32 rb = chunk(rec[2 * row], rec[2 * row + 1], tc, colk);
            if (tc + lane < n) rowk[tc + lane] = (unsigned short)rb;
        }
        total = 0;
        for (int tc = 0; tc < n; tc += 64) {
            const u32 k = tc + lane < n ? rowk[tc + lane] : 0u;
            const u32 peer = (u32)__builtin_amdgcn_ds_bpermute((int)(sq_lane_of<G>(k & 63u) << 2), (int)colk);
            total += __popcll(__ballot(tc + lane < n && (peer & 0xFFFFu) == (u32)(tc + lane)));
        }
        if (lane == 0) counts[r] = total;
    }
}

int launch_db_scan(reloc_ctx *ctx, const uint8_t *db_desc, const int64_t *db_off, int64_t n_rec,
                   const int32_t *rec_ids, const int32_t *n_ids_dev, int n_ids_max, const uint8_t *cur,
                   const int32_t *n_cur_dev, int n_cur_max, int max_rows, int32_t *counts, int32_t *m_qidx,
                   int32_t *m_tidx, int32_t *m_dist, int32_t *m_n, int emit_stride, const ScanMask *mask_p)
{
    if (n_ids_max <= 0) return RELOC_OK;
    ScanMask mask;
    mask.xyh = nullptr;
    mask.q[0] = mask.q[1] = mask.q[2] = 0; mask.q[3] = 1;
    if (mask_p) mask = *mask_p;
    if (n_cur_max > 65535) { reloc_set_error("db scan: more than 65535 current descriptors"); return RELOC_E_CAPACITY; }
    if (max_rows > MAX_REC_ROWS) { reloc_set_error("db scan: record larger than %d rows", MAX_REC_ROWS); return RELOC_E_CAPACITY; }
    if (max_rows < 1) max_rows = 1;
    if (!m_qidx && !rec_ids && !n_ids_dev && !mask.xyh && counts && n_cur_max <= 64 && max_rows <= SQ_MAX_ROWS) {
        // few queries: lane = teach row (k_db_scan_rows)
        if (n_cur_max == 0) {                              // a capacity of zero descriptors: nothing may be read from `cur`
            HIP_TRY(hipMemsetAsync(counts, 0, (size_t)n_ids_max * sizeof(int32_t), ctx->stream));
            return RELOC_OK;
        }
        int grid = ctx->num_cu * 8;                        // 8 workgroups of 4 waves per CU
        const int need = (n_ids_max + SQ_WAVES - 1) / SQ_WAVES;
        if (grid > need) grid = need;
#define RELOC_LAUNCH_ROWS(G, HOIST)                                                                                           \
    hipLaunchKernelGGL((k_db_scan_rows<G, HOIST>), dim3(grid), dim3(64 * SQ_WAVES), 0, ctx->stream, (const uint4 *)db_desc, db_off, n_ids_max, \
                       (const uint4 *)cur, n_cur_dev, n_cur_max, counts)
        // G = queries per butterfly group: a call with 1-4 queries evaluates 4 distances per row, not 16; its query words stay in SGPRs
        if (n_cur_max <= 4) RELOC_LAUNCH_ROWS(4, true); else if (n_cur_max <= 8) RELOC_LAUNCH_ROWS(8, false); else RELOC_LAUNCH_ROWS(16, false);
#undef RELOC_LAUNCH_ROWS
        HIP_TRY(hipGetLastError());
        return RELOC_OK;
    }
    const int nj = n_cur_max <= 128 ? 2 : (n_cur_max <= 256 ? 4 : 8);      // columns per lane, see k_db_scan
    const int cb = 64 * nj;
    const int ncb = (n_cur_max + cb - 1) / cb > 0 ? (n_cur_max + cb - 1) / cb : 1;
    // column minima: one buffer for the capacity; the counting scan double-buffers inside it when the run-time count leaves room
    // (db_count_body), so at least two column blocks
    const int col_words = (m_qidx || ncb >= 2 ? ncb : 2) * cb;
    const size_t lds = (size_t)(col_words + max_rows + 16) * 4;
    if (lds > 160 * 1024) { reloc_set_error("db scan: LDS demand %zu bytes", lds); return RELOC_E_CAPACITY; }
    // Whole-database scans (host-known record count, more records than resident workgroups): workgroups DRAW their records
    // from per-XCD ticket counters instead of a static round-robin deal, so the work stays balanced to the last record.  One
    // resident generation that lives as long as the launch is the fastest form alone but starves the other streams of a
    // multi-context run, so beside other streams a workgroup serves a budget and leaves: the grid holds THREE generations (the
    // small kernels run at wave priority 3, RELOC_SMALL_KERNEL_PRIO, and need few free slots); reloc_set_exclusive / a process's
    // only context selects the one-generation form.  Candidate lists, single records and the 128-column kernel (its short
    // records do not cover the draw latency) are dealt statically.  Measurements behind each choice: profiles/README.md
    // "Dropped experiments" #6.
    // Waves per record of the counting scan (NW): 4 waves share a record's rows (finest grain: shortest tail of the launch),
    // or 2, or ONE wave owns a record (no row left for a second chunk epilogue, no barrier that waits for anybody).
    int nw = 4;
    if (!m_qidx && nj == 8) {
        nw = ctx->scan_nw == 1 || ctx->scan_nw == 2 || ctx->scan_nw == 4 ? ctx->scan_nw : 4;      // (8: 180 vs 152 us, profiles/r4_scan_intercept.log)
        if (nw == 1 && lds * 16 > 150 * 1024) nw = 2;          // 16 one-wave workgroups per CU have to fit their LDS
    }
    int wg_per_cu = 16 / nw;                       // 16 waves per CU (128-VGPR kernel) ...
    if ((size_t)wg_per_cu * lds > 160 * 1024) wg_per_cu = (int)(160 * 1024 / lds);     // ... unless their LDS does not fit
    const int resident = ctx->num_cu * wg_per_cu;
    int grid = ctx->scan_grid > 0 ? ctx->scan_grid : ctx->num_cu * 16;    // RELOC_SCAN_GRID: developer switch, read at creation
    u32 *ticket = nullptr;
    int quota = 0, n_bounded = 0;
    if (!rec_ids && !n_ids_dev && n_ids_max > resident && ctx->scan_ticket && ctx->scan_grid >= 0 && nj == 8) {
        ticket = ctx->scan_ticket;
        const int gens = ctx->scan_gens > 0 ? ctx->scan_gens : (RELOC_SCAN_GENS_SHARED > 0 ? RELOC_SCAN_GENS_SHARED : 1);
        // `gens` generations of workgroups: all but the last resident one leave after their share of the database's ROWS, the
        // last generation draws until the counters are dry (see db_count_body); the row total is the host's when the
        // context's own database is scanned, else 64 per record
        const int q_rec = (n_ids_max + resident * gens - 1) / (resident * gens);          // records per workgroup at the average size
        n_bounded = (n_ids_max + q_rec - 1) / q_rec;                                      // workgroups x q_rec >= records
        grid = n_bounded + 8 < n_ids_max ? n_bounded + 8 : n_ids_max;                     // + one sweeper per XCD
        const int64_t rows = db_desc == ctx->db_desc && ctx->db_rows > 0 ? ctx->db_rows : (int64_t)n_ids_max * 64;
        quota = (int)((rows * q_rec + n_ids_max - 1) / n_ids_max);                        // that many records' worth of ROWS
        if (quota < 1) quota = 1;
        if (!ctx->scan_quota_rows) { quota = q_rec; grid = n_bounded; n_bounded = -1; }   // developer switch: the record quota of rounds 2-3a
        // ONE resident generation that draws until the counters are dry: always for a context that is alone (reloc_set_exclusive:
        // nobody to hand slots to), and since round 4 beside other streams as well (RELOC_SCAN_GENS_SHARED == 0, reloc_internal.h)
        if ((ctx_alone(ctx) || RELOC_SCAN_GENS_SHARED == 0) && ctx->scan_gens == 0) { quota = 0; grid = resident; }
        if (ctx->scan_gens < 0) { quota = 0; grid = ctx->scan_gens <= -2 ? ctx->num_cu * (-ctx->scan_gens - 1) : resident; }   // developer switch: one generation, no quota; -2 / -3 / -4: 1 / 2 / 3 workgroups per CU
    }
    if (grid > n_ids_max) grid = n_ids_max;
    // Match lists of a few candidate records (the tick's emit pass, reloc_match_mutual): one workgroup per record is
    // alone on its CU and one wave per SIMD issues an instruction only every ~7 cycles, so the record's rows are the
    // kernel's run time.  8 waves per record take the synchronous tick from 322 / 152 to 318 / 149 us (global / local) --
    // and the 4-stream whole-database run from 5940 to 5360 frames/s: a 512-thread workgroup of this register size needs
    // TWO scan workgroups of one CU to retire before it fits.  So: 8 waves where no scan runs beside it (local-candidate
    // ticks, single calls: ctx->latency_shapes), 4 waves in ticks that scan the database.
#define RELOC_LAUNCH_SCAN(NJ, EMIT, NW)                                                                                      \
    hipLaunchKernelGGL((k_db_scan<NJ, EMIT, NW>), dim3(grid), dim3(64 * NW), lds, ctx->stream, (const uint4 *)db_desc, db_off, rec_ids, \
                       n_ids_dev, n_ids_max, (const uint4 *)cur, n_cur_dev, n_cur_max, max_rows, counts, m_qidx, m_tidx, m_dist, \
                       m_n, emit_stride, mask, ticket, quota, n_bounded, col_words)
    if (m_qidx && ctx->latency_shapes) {
        if (nj == 2) RELOC_LAUNCH_SCAN(2, true, 8); else if (nj == 4) RELOC_LAUNCH_SCAN(4, true, 8); else RELOC_LAUNCH_SCAN(8, true, 8);
    } else if (m_qidx) {
        if (nj == 2) RELOC_LAUNCH_SCAN(2, true, 4); else if (nj == 4) RELOC_LAUNCH_SCAN(4, true, 4); else RELOC_LAUNCH_SCAN(8, true, 4);
    } else if (nj == 8) {
        if (nw == 1) RELOC_LAUNCH_SCAN(8, false, 1); else if (nw == 2) RELOC_LAUNCH_SCAN(8, false, 2); else RELOC_LAUNCH_SCAN(8, false, 4);
    } else {
        if (nj == 2) RELOC_LAUNCH_SCAN(2, false, 4); else RELOC_LAUNCH_SCAN(4, false, 4);
    }
#undef RELOC_LAUNCH_SCAN
    HIP_TRY(hipGetLastError());
    return RELOC_OK;
}

// One launch for the whole-database scans of n contexts that share a stream and a database (ctxs[0]'s is scanned):
// frame f = ctxs[f]'s current features, counts into ctxs[f]->db_counts.  q: n x 4 base_link quaternions (heading mask),
// auto_mode: frames whose local search found candidates stand down on the device.
int launch_db_scan_batch(reloc_ctx *const *ctxs, int n, const double *q, double cos_tol, bool auto_mode, bool heading_mask)
{
    reloc_ctx *c0 = ctxs[0];
    if (n < 1 || n > SCAN_BATCH_MAX) { reloc_set_error("scan batch: 1..%d frames", SCAN_BATCH_MAX); return RELOC_E_ARG; }
    if (c0->max_feat > 65535 || c0->db_max_rows > MAX_REC_ROWS) { reloc_set_error("scan batch: capacity"); return RELOC_E_CAPACITY; }
    ScanBatch bt;
    bt.n = n;
    bt.xyh = heading_mask ? c0->db_xy_heading : nullptr;
    bt.cos_tol = cos_tol;
    for (int f = 0; f < SCAN_BATCH_MAX; ++f) {
        reloc_ctx *c = ctxs[f < n ? f : 0];
        bt.cur[f] = (const uint4 *)c->f_desc; bt.n_cur[f] = c->f_count; bt.counts[f] = c->db_counts;
        bt.skip_if[f] = auto_mode ? c->cand_n : nullptr;
        for (int k = 0; k < 4; ++k) bt.q[f][k] = q[4 * (f < n ? f : 0) + k];
    }
    const int n_ids = (int)c0->db_records, max_rows = c0->db_max_rows < 1 ? 1 : c0->db_max_rows;
    // n_cur_max = the feature capacity; the 8-column kernel walks column blocks of 512 (one block for nfeatures <= 512)
    const int ncb = (c0->max_feat + 511) / 512;
    const int col_words = (ncb >= 2 ? ncb : 2) * 512;         // see launch_db_scan
    const size_t lds_all = (size_t)(col_words + max_rows + 16) * 4;
    int gens = c0->scan_gens > 0 ? c0->scan_gens : (RELOC_SCAN_GENS_SHARED > 0 ? RELOC_SCAN_GENS_SHARED : 1);
    if (c0->scan_batch_gens > 0) gens = c0->scan_batch_gens;
    const int resident = c0->num_cu * 4;
    // the grid holds `gens` generations in all (not per frame): a workgroup's quota grows with the batch, and with it
    // the share of the launch that is not prologue
    int per_frame = (resident * gens + n - 1) / n;
    if (per_frame > n_ids) per_frame = n_ids;
    if (per_frame < 1) per_frame = 1;
    // per frame: per_frame workgroups with a row budget (together: the whole database) + one sweeper behind them that draws
    // until the counters are dry (db_count_body)
    const int q_rec = (n_ids + per_frame - 1) / per_frame;                           // records per workgroup at the average size
    const int n_bounded = (n_ids + q_rec - 1) / q_rec;
    int quota = (int)(((c0->db_rows > 0 ? c0->db_rows : (int64_t)n_ids * 64) * q_rec + n_ids - 1) / n_ids);   // in ROWS
    if (quota < 1) quota = 1;
    per_frame = n_bounded + 1;                                                       // + one sweeper per frame
    if (per_frame > n_ids) per_frame = n_ids;
    int nb_arg = n_bounded;
    if (!c0->scan_quota_rows) { quota = q_rec; per_frame = n_bounded; nb_arg = -1; }
    hipLaunchKernelGGL(k_db_scan_batch, dim3(per_frame * n), dim3(256), lds_all, c0->stream, (const uint4 *)c0->db_desc, c0->db_off,
                       n_ids, c0->max_feat, max_rows, bt, c0->scan_ticket, quota, nb_arg, col_words);
    HIP_TRY(hipGetLastError());
    return RELOC_OK;
}

// ---------------------------------------------------------------------------------------------
// Generic two-nearest-neighbour search: lane = one row of A, B rows stream through the scalar
// cache.  key = distance << 22 | index (nb < 2^22).  grid.x tiles A rows, grid.y splits B.
// Partial results (2 keys per A row per split) are merged by k_knn2_merge.
__global__ __launch_bounds__(256) void k_knn2(const uint4 *__restrict__ A, int na, const uint4 *__restrict__ B,
                                              int nb, int rows_per_split, u32 *__restrict__ part)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int j0 = blockIdx.y * rows_per_split;
    const int j1 = min(nb, j0 + rows_per_split);
    u32 q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (i < na) {
        const uint4 a = A[2 * i], b = A[2 * i + 1];
        q[0] = a.x; q[1] = a.y; q[2] = a.z; q[3] = a.w;
        q[4] = b.x; q[5] = b.y; q[6] = b.z; q[7] = b.w;
    }
    u32 k0 = 0xFFFFFFFFu, k1 = 0xFFFFFFFFu;
#pragma unroll 4
    for (int j = j0; j < j1; ++j) {
        const u32 h = ham8(q, B[2 * j], B[2 * j + 1], 0);
        const u32 key = (h << 22) | (u32)j;
        const u32 m = umax(k0, key);
        k0 = umin(k0, key);
        k1 = umin(k1, m);
    }
    if (i < na) {
        part[((size_t)blockIdx.y * na + i) * 2] = k0;
        part[((size_t)blockIdx.y * na + i) * 2 + 1] = k1;
    }
}

__global__ void k_knn2_merge(const u32 *__restrict__ part, int na, int nsplit, int32_t *__restrict__ idx,
                             int32_t *__restrict__ dist)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= na) return;
    u32 k0 = 0xFFFFFFFFu, k1 = 0xFFFFFFFFu;
    for (int s = 0; s < nsplit; ++s)
        for (int e = 0; e < 2; ++e) {
            const u32 key = part[((size_t)s * na + i) * 2 + e];
            const u32 m = umax(k0, key);
            k0 = umin(k0, key);
            k1 = umin(k1, m);
        }
    idx[2 * i] = k0 == 0xFFFFFFFFu ? -1 : (int32_t)(k0 & 0x3FFFFFu);
    dist[2 * i] = k0 == 0xFFFFFFFFu ? -1 : (int32_t)(k0 >> 22);
    idx[2 * i + 1] = k1 == 0xFFFFFFFFu ? -1 : (int32_t)(k1 & 0x3FFFFFu);
    dist[2 * i + 1] = k1 == 0xFFFFFFFFu ? -1 : (int32_t)(k1 >> 22);
}

// ---------------------------------------------------------------------------------------------
// All-pairs u16 distance matrix.  Each lane keeps 8 consecutive B rows (64 VGPRs); A rows stream
// through the scalar cache; per A row a lane produces 8 distances packed into one 16-byte store,
// so a wave writes 1 KiB of one output row per instruction (HBM-write-bound shape).
// grid.x = column tiles of 2048 (4 waves x 64 lanes x 8), grid.y = row tiles of MAT_ROWS.
constexpr int MAT_ROWS = 128;      // row tile of the unaligned fallback kernel
constexpr int MAT_UNIT_ROWS = 8;   // rows per work unit of the persistent kernel

// Persistent: the grid is sized to what is resident at once (a grid a few percent larger than that
// runs a second, almost empty round and loses ~30 %).  A workgroup is bound to one column tile
// (blockIdx % n_col_tiles) so its 8 x 256 B columns stay in registers, and takes the 8-row units
// k, k + K, k + 2K, ... of that tile (K = workgroups per column tile): no atomics, no barriers.
// FULL = every lane's 8 columns exist (all column tiles except possibly the last).
template <bool FULL>
__device__ __forceinline__ void matrix_body(const uint4 *__restrict__ A, int64_t na, const uint4 *__restrict__ B, int64_t nb,
                                            uint16_t *__restrict__ out, int ct, int k0, int kstep, int n_units)
{
    const int64_t j0 = ((int64_t)ct * 256 + threadIdx.x) * 8;
    u32 b[8][8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int64_t j = FULL || j0 + c < nb ? j0 + c : nb - 1;
        const uint4 lo = B[2 * j], hi = B[2 * j + 1];
        b[c][0] = lo.x; b[c][1] = lo.y; b[c][2] = lo.z; b[c][3] = lo.w;
        b[c][4] = hi.x; b[c][5] = hi.y; b[c][6] = hi.z; b[c][7] = hi.w;
    }
    if (!FULL && j0 >= nb) return;
    // A rows come through the scalar cache, one fetch in flight: the fetch of the next row (of this unit,
    // or the first row of this workgroup's next unit) is issued as soon as the current row has landed and
    // hides behind the current row's ~135 VALU instructions (see srow_landed).
    // Measured r2 (tools/exp_matrix2.hip, interleaved rounds on one MI355X, 20000 x 20000): the r1 form of this loop
    // 218 us; non-temporal stores (the 800 MB output is written once and never read here: nt keeps it from evicting the
    // B rows' lines and its store stream alone runs 6.3 instead of 5.7 TB/s) 202 us; the 8 distances of a lane as 8
    // accumulator chains in pinned order (ham8_cols) and 6 resident workgroups per CU 168 us = 0.60 of 8 TB/s, the same
    // loop without its store 160 us: the kernel is bound by the VALU's ~2.5 T pairs/s on this formulation, not by HBM.
    // (After an idle period the chip needs ~40 ms of load before it runs at this rate: bench.py pre-rolls.)
    // row indices are 32-bit (launch_matrix checks na < 2^31): clamps and bound tests stay on the scalar unit
    const int last = (int)na - 1, n_rows = (int)na;
    auto row_at = [&](int i) { return min(i, last); };
    const int i_first = row_at(k0 * MAT_UNIT_ROWS);
    uint4 ra = A[2 * (int64_t)i_first], rb = A[2 * (int64_t)i_first + 1];
    typedef u32 v4u __attribute__((ext_vector_type(4)));
    const u32 lane_off = (u32)(j0 * 2);                                   // byte offset inside an output row: the row base is wave-uniform
    for (int unit = k0; unit < n_units; unit += kstep) {
        const int i0 = unit * MAT_UNIT_ROWS;
        const bool whole = i0 + MAT_UNIT_ROWS <= n_rows;                  // wave-uniform: no per-row bound checks in whole units
#pragma unroll
        for (int e = 0; e < MAT_UNIT_ROWS; ++e) {
            srow_landed(ra.x);
            const int inext = row_at(e + 1 < MAT_UNIT_ROWS ? i0 + e + 1 : i0 + kstep * MAT_UNIT_ROWS);
            const uint4 na_ = A[2 * (int64_t)inext], nb_ = A[2 * (int64_t)inext + 1];
            __builtin_amdgcn_sched_barrier(0);
            const u32 rw[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
            u32 o[8];
            ham8_cols<8>(b, rw, o);                                       // 8 accumulator chains, order pinned (see ham8_cols)
            u32 w[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) w[p] = o[2 * p] | (o[2 * p + 1] << 16);
            if (whole || i0 + e < n_rows) {
                char *row = reinterpret_cast<char *>(out + (int64_t)(i0 + e) * nb);            // scalar registers
                if (FULL || j0 + 8 <= nb) {
                    // SGPR row base + 32-bit lane offset: no vector address arithmetic per row; non-temporal
                    const v4u v = {w[0], w[1], w[2], w[3]};
                    asm volatile("global_store_dwordx4 %0, %1, %2 nt" ::"v"(lane_off), "v"(v), "s"(row) : "memory");
                } else {
                    uint16_t *o = reinterpret_cast<uint16_t *>(row + lane_off);
                    for (int c = 0; c < 8 && j0 + c < nb; ++c) o[c] = (uint16_t)(w[c >> 1] >> ((c & 1) * 16));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            ra = na_;
            rb = nb_;
        }
    }
}

__global__ __launch_bounds__(256) void k_hamming_matrix(const uint4 *__restrict__ A, int64_t na,
                                                        const uint4 *__restrict__ B, int64_t nb,
                                                        uint16_t *__restrict__ out, int n_col_tiles, int n_units)
{
    const int ct = blockIdx.x % n_col_tiles;
    const int k0 = blockIdx.x / n_col_tiles, kstep = gridDim.x / n_col_tiles;
    if ((int64_t)(ct + 1) * 2048 <= nb) matrix_body<true>(A, na, B, nb, out, ct, k0, kstep, n_units);
    else matrix_body<false>(A, na, B, nb, out, ct, k0, kstep, n_units);
}

// slow path for outputs whose rows are not 16-byte aligned (nb % 8 != 0)
__global__ void k_hamming_matrix_any(const uint4 *__restrict__ A, int64_t na, const uint4 *__restrict__ B,
                                     int64_t nb, uint16_t *__restrict__ out)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.y * MAT_ROWS;
    const int64_t i1 = i0 + MAT_ROWS < na ? i0 + MAT_ROWS : na;
    if (j >= nb) return;
    u32 q[8];
    const uint4 lo = B[2 * j], hi = B[2 * j + 1];
    q[0] = lo.x; q[1] = lo.y; q[2] = lo.z; q[3] = lo.w; q[4] = hi.x; q[5] = hi.y; q[6] = hi.z; q[7] = hi.w;
    for (int64_t i = i0; i < i1; ++i) out[i * nb + j] = (uint16_t)ham8(q, A[2 * i], A[2 * i + 1], 0);
}

static int launch_matrix(reloc_ctx *ctx, const uint8_t *a, int64_t na, const uint8_t *b, int64_t nb, uint16_t *out)
{
    if (na <= 0 || nb <= 0) return RELOC_OK;
    if (nb % 8 == 0 && ((uintptr_t)out & 15) == 0) {
        const int n_col_tiles = (int)((nb + 2047) / 2048);
        const int64_t n_units = (na + MAT_UNIT_ROWS - 1) / MAT_UNIT_ROWS;
        if (n_col_tiles > 4096 || na > 0x7ffffff0 || nb * 2 > 0xffffffffll) { reloc_set_error("hamming matrix: shape too large"); return RELOC_E_CAPACITY; }
        static int per_cu = 0;
        if (!per_cu) {
            // blocks of 4 waves = one wave per SIMD each: residency = waves per SIMD the register
            // allocation admits, capped by the occupancy query (which can over-report by one)
            hipFuncAttributes fa;
            int api = 0;
            per_cu = 4;
            if (hipFuncGetAttributes(&fa, (const void *)k_hamming_matrix) == hipSuccess && fa.numRegs > 0) {
                const int alloc = (fa.numRegs + 7) / 8 * 8;
                per_cu = 512 / alloc < 8 ? 512 / alloc : 8;
            }
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, k_hamming_matrix, 256, 0) == hipSuccess && api > 0 && api < per_cu)
                per_cu = api;
            if (per_cu > 6) per_cu = 6;           // measured (8 chains): 5 / 6 / 7 resident workgroups per CU 174 / 168 / 189 us
            if (per_cu < 1) per_cu = 1;
        }
        int grid = ctx->num_cu * per_cu;
        grid = grid / n_col_tiles * n_col_tiles;
        if (grid < n_col_tiles) grid = n_col_tiles;
        const int64_t cap = n_units * n_col_tiles;
        if (grid > cap) grid = (int)cap;
        reloc_prof_begin(ctx, RELOC_PROF_MATRIX);
        hipLaunchKernelGGL(k_hamming_matrix, dim3(grid), dim3(256), 0, ctx->stream, (const uint4 *)a, na, (const uint4 *)b, nb,
                           out, n_col_tiles, (int)n_units);
        reloc_prof_end(ctx, RELOC_PROF_MATRIX);
    } else {
        const int64_t gy = (na + MAT_ROWS - 1) / MAT_ROWS;
        if (gy > 65535) { reloc_set_error("hamming matrix: too many rows for the unaligned path"); return RELOC_E_CAPACITY; }
        reloc_prof_begin(ctx, RELOC_PROF_MATRIX);
        dim3 grid((unsigned)((nb + 255) / 256), (unsigned)gy);
        hipLaunchKernelGGL(k_hamming_matrix_any, grid, dim3(256), 0, ctx->stream, (const uint4 *)a, na, (const uint4 *)b, nb, out);
        reloc_prof_end(ctx, RELOC_PROF_MATRIX);
    }
    HIP_TRY(hipGetLastError());
    return RELOC_OK;
}

// ---------------------------------------------------------------------------------------------
// C-ABI entry points
RELOC_API int reloc_hamming_matrix_dev(reloc_ctx *ctx, const uint8_t *a, int64_t na, const uint8_t *b, int64_t nb,
                                       uint16_t *out)
{
    ARG_CHECK_CTX(ctx, a && b && out && na >= 0 && nb >= 0, "reloc_hamming_matrix_dev");
    ARG_CHECK((((uintptr_t)a | (uintptr_t)b) & 15) == 0, "descriptor arrays must be 16-byte aligned");
    return launch_matrix(ctx, a, na, b, nb, out);
}

RELOC_API int reloc_hamming_matrix(reloc_ctx *ctx, const uint8_t *a, int64_t na, const uint8_t *b, int64_t nb,
                                   uint16_t *out)
{
    ARG_CHECK_CTX(ctx, a && b && out && na >= 0 && nb >= 0, "reloc_hamming_matrix");
    if (na == 0 || nb == 0) return RELOC_OK;
    void *da, *db, *dout;
    int rc;
    if ((rc = reloc_scratch(ctx, 0, na * 32, &da))) return rc;
    if ((rc = reloc_scratch(ctx, 1, nb * 32, &db))) return rc;
    if ((rc = reloc_scratch(ctx, 2, na * nb * 2, &dout))) return rc;
    HIP_TRY(hipMemcpyAsync(da, a, (size_t)na * 32, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(db, b, (size_t)nb * 32, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = launch_matrix(ctx, (const uint8_t *)da, na, (const uint8_t *)db, nb, (uint16_t *)dout))) return rc;
    HIP_TRY(hipMemcpyAsync(out, dout, (size_t)na * nb * 2, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RELOC_OK;
}

RELOC_API int reloc_match_knn2(reloc_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt, int32_t *idx,
                               int32_t *dist)
{
    ARG_CHECK_CTX(ctx, nq >= 0 && nt >= 0 && (nq == 0 || (q && idx && dist)) && (nt == 0 || t), "reloc_match_knn2");
    if (nq == 0) return RELOC_OK;
    if (nt == 0) {
        for (int i = 0; i < 2 * nq; ++i) { idx[i] = -1; dist[i] = -1; }
        return RELOC_OK;
    }
    if (nt >= (1 << 22)) { reloc_set_error("knn2: train set too large"); return RELOC_E_CAPACITY; }
    int nsplit = (int)((int64_t)ctx->num_cu * 4 / ((nq + 255) / 256));
    if (nsplit < 1) nsplit = 1;
    if (nsplit > (nt + 63) / 64) nsplit = (nt + 63) / 64;
    const int rows_per_split = (nt + nsplit - 1) / nsplit;
    nsplit = (nt + rows_per_split - 1) / rows_per_split;
    void *dq, *dt, *dpart, *dout;
    int rc;
    if ((rc = reloc_scratch(ctx, 0, (int64_t)nq * 32, &dq))) return rc;
    if ((rc = reloc_scratch(ctx, 1, (int64_t)nt * 32, &dt))) return rc;
    if ((rc = reloc_scratch(ctx, 2, (int64_t)nsplit * nq * 8, &dpart))) return rc;
    if ((rc = reloc_scratch(ctx, 3, (int64_t)nq * 16, &dout))) return rc;
    HIP_TRY(hipMemcpyAsync(dq, q, (size_t)nq * 32, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dt, t, (size_t)nt * 32, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_knn2, dim3((nq + 255) / 256, nsplit), dim3(256), 0, ctx->stream, (const uint4 *)dq, nq,
                       (const uint4 *)dt, nt, rows_per_split, (u32 *)dpart);
    int32_t *didx = (int32_t *)dout, *ddist = didx + 2 * (size_t)nq;
    hipLaunchKernelGGL(k_knn2_merge, dim3((nq + 255) / 256), dim3(256), 0, ctx->stream, (const u32 *)dpart, nq, nsplit,
                       didx, ddist);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(idx, didx, (size_t)nq * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dist, ddist, (size_t)nq * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RELOC_OK;
}

RELOC_API int reloc_match_mutual(reloc_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt, int32_t *qidx,
                                 int32_t *tidx, int32_t *dist, int32_t *n_out)
{
    ARG_CHECK_CTX(ctx, n_out && nq >= 0 && nt >= 0, "reloc_match_mutual");
    *n_out = 0;
    if (nq == 0 || nt == 0) return RELOC_OK;
    ARG_CHECK(q && t && qidx && tidx && dist, "reloc_match_mutual: NULL array");
    if (nq > MAX_REC_ROWS) { reloc_set_error("match: query set larger than %d rows", MAX_REC_ROWS); return RELOC_E_CAPACITY; }
    void *dq, *dt, *dm;
    int rc;
    if ((rc = reloc_scratch(ctx, 0, (int64_t)nq * 32 + 64, &dq))) return rc;
    if ((rc = reloc_scratch(ctx, 1, (int64_t)nt * 32, &dt))) return rc;
    if ((rc = reloc_scratch(ctx, 2, (int64_t)nq * 12 + 64, &dm))) return rc;
    // offsets {0, nq} live in front of the match arrays
    int64_t offs[2] = {0, nq};
    int64_t *doff = (int64_t *)dm;
    int32_t *dn = (int32_t *)(doff + 2);
    int32_t *dqi = dn + 4, *dti = dqi + nq, *ddi = dti + nq;
    HIP_TRY(hipMemcpyAsync(doff, offs, sizeof(offs), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dq, q, (size_t)nq * 32, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(dt, t, (size_t)nt * 32, hipMemcpyHostToDevice, ctx->stream));
    ctx->latency_shapes = true;                               // a single record, nothing runs beside it: 8 waves
    rc = launch_db_scan(ctx, (const uint8_t *)dq, doff, 1, nullptr, nullptr, 1, (const uint8_t *)dt, nullptr, nt,
                        nq, nullptr, dqi, dti, ddi, dn, nq);
    ctx->latency_shapes = false;
    if (rc) return rc;
    int32_t n = 0;
    HIP_TRY(hipMemcpyAsync(&n, dn, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (n > 0) {
        HIP_TRY(hipMemcpyAsync(qidx, dqi, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipMemcpyAsync(tidx, dti, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipMemcpyAsync(dist, ddi, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    *n_out = n;
    return RELOC_OK;
}

// ---- database ---------------------------------------------------------------------------------
// The selected database is a capacity-reserved arena (DbArena in reloc_internal.h): upload fills it, append copies one
// record behind the last row, reserve grows it.  Nothing is published in the ctx before every allocation and copy of
// an operation has succeeded: a failed upload leaves "no database" (db_records == 0), a failed reserve / append leaves
// the database as it was.
__global__ void k_db_index(const double *__restrict__ pose, int64_t first, int64_t n, double b0, double b1, double b2,
                           double *__restrict__ xyh)
{
    // heading of base_link +X in the world from the stored CAMERA pose, exactly as the reference
    // composes it (M:233-245): R_wb = R_wc @ B.T, fwd = R_wb @ [1,0,0] = R_wc @ B[0,:]
    const int64_t i = first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= first + n) return;
    const double qx = pose[7 * i + 3], qy = pose[7 * i + 4], qz = pose[7 * i + 5], qw = pose[7 * i + 6];
    const double r00 = 1 - 2 * (qy * qy + qz * qz), r01 = 2 * (qx * qy - qz * qw), r02 = 2 * (qx * qz + qy * qw);
    const double r10 = 2 * (qx * qy + qz * qw), r11 = 1 - 2 * (qx * qx + qz * qz), r12 = 2 * (qy * qz - qx * qw);
    const double fx = r00 * b0 + r01 * b1 + r02 * b2, fy = r10 * b0 + r11 * b1 + r12 * b2;
    const double fn = sqrt(fx * fx + fy * fy);
    xyh[4 * i] = pose[7 * i];
    xyh[4 * i + 1] = pose[7 * i + 1];
    xyh[4 * i + 2] = fn > 0 ? fx / fn : 1.0;      // cos(heading)
    xyh[4 * i + 3] = fn > 0 ? fy / fn : 0.0;      // sin(heading)
}

// headings follow the camera mounting (reloc_set_camera); the (x, y) a record is filed under is kept
__global__ void k_db_reheading(const double *__restrict__ pose, int64_t n, double b0, double b1, double b2, double *__restrict__ xyh)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double qx = pose[7 * i + 3], qy = pose[7 * i + 4], qz = pose[7 * i + 5], qw = pose[7 * i + 6];
    const double r00 = 1 - 2 * (qy * qy + qz * qz), r01 = 2 * (qx * qy - qz * qw), r02 = 2 * (qx * qz + qy * qw);
    const double r10 = 2 * (qx * qy + qz * qw), r11 = 1 - 2 * (qx * qx + qz * qz), r12 = 2 * (qy * qz - qx * qw);
    const double fx = r00 * b0 + r01 * b1 + r02 * b2, fy = r10 * b0 + r11 * b1 + r12 * b2;
    const double fn = sqrt(fx * fx + fy * fy);
    xyh[4 * i + 2] = fn > 0 ? fx / fn : 1.0;
    xyh[4 * i + 3] = fn > 0 ? fy / fn : 0.0;
}

int db_reindex(reloc_ctx *ctx)
{
    // both resident databases follow a change of the camera mounting
    for (int slot = 0; slot < 2; ++slot) {
        const bool sel = slot == ctx->db_sel;
        const double *pose = sel ? ctx->db_pose : ctx->db_slot[slot].pose;
        double *xyh = sel ? ctx->db_xy_heading : ctx->db_slot[slot].xy_heading;
        const int64_t n = sel ? ctx->db_records : ctx->db_slot[slot].records;
        if (!pose || !xyh || n <= 0) continue;
        hipLaunchKernelGGL(k_db_reheading, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, pose, n, ctx->b2c_R[0],
                           ctx->b2c_R[1], ctx->b2c_R[2], xyh);
    }
    HIP_TRY(hipGetLastError());
    return RELOC_OK;
}

namespace {
struct DbBuffers {
    uint8_t *desc = nullptr; float *pts3d = nullptr, *kp2d = nullptr; int64_t *off = nullptr; double *pose = nullptr, *xyh = nullptr;
    int32_t *counts = nullptr; unsigned long long *topk = nullptr;
    void release()
    {
        void *p[] = {desc, pts3d, kp2d, off, pose, xyh, counts, topk};
        for (void *q : p) if (q) (void)hipFree(q);
        *this = DbBuffers();
    }
};
}   // namespace

void db_arrays_drop(DbShare *&share, uint8_t *&desc, float *&pts3d, float *&kp2d, int64_t *&off, double *&pose, double *&xyh)
{
    if (share && __atomic_sub_fetch(&share->refs, 1, __ATOMIC_ACQ_REL) == 0) {
        void *p[] = {desc, pts3d, kp2d, off, pose, xyh};
        for (void *q : p) if (q) (void)hipFree(q);
        delete share;
    }
    share = nullptr;
    desc = nullptr; pts3d = nullptr; kp2d = nullptr; off = nullptr; pose = nullptr; xyh = nullptr;
}

// an adopted database is let go (not freed while anybody else holds it) before this ctx gets one of its own again
static void db_unshare(reloc_ctx *ctx)
{
    if (!ctx->db_shared) return;
    if (ctx->db_counts) (void)hipFree(ctx->db_counts);
    if (ctx->topk_part) (void)hipFree(ctx->topk_part);
    db_arrays_drop(ctx->db_share, ctx->db_desc, ctx->db_pts3d, ctx->db_kp2d, ctx->db_off, ctx->db_pose, ctx->db_xy_heading);
    ctx->db_counts = nullptr; ctx->topk_part = nullptr;
    ctx->db_records = ctx->db_rows = ctx->db_cap_records = ctx->db_cap_rows = 0;
    ctx->db_max_rows = 0; ctx->topk_blocks = 0;
    ctx->db_shared = false;
}

// Grow the selected arena to at least (cap_records, cap_rows); contents are kept.  All-or-nothing.  The old arrays are
// let go of, not necessarily freed: contexts that adopted them (reloc_db_share) keep scanning them.
int db_reserve(reloc_ctx *ctx, int64_t cap_records, int64_t cap_rows)
{
    if (ctx->db_shared) { reloc_set_error("the selected database is shared from another context (read-only here)"); return RELOC_E_STATE; }
    if (cap_records < 1) cap_records = 1;
    if (cap_rows < 1) cap_rows = 1;
    if (cap_records <= ctx->db_cap_records && cap_rows <= ctx->db_cap_rows && ctx->db_desc) return RELOC_OK;
    if (cap_records < ctx->db_cap_records) cap_records = ctx->db_cap_records;
    if (cap_rows < ctx->db_cap_rows) cap_rows = ctx->db_cap_rows;
    if (cap_records > MAX_DB_RECORDS) { reloc_set_error("database: more than %lld records", (long long)MAX_DB_RECORDS); return RELOC_E_CAPACITY; }
    DbBuffers nb;
    const int blocks = (int)((cap_records + 1023) / 1024);
    hipError_t e = hipSuccess;
    auto grab = [&](void **p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes); };
    grab((void **)&nb.desc, (size_t)cap_rows * 32);
    grab((void **)&nb.pts3d, (size_t)cap_rows * 12);
    grab((void **)&nb.kp2d, (size_t)cap_rows * 8);
    grab((void **)&nb.off, (size_t)(cap_records + 1) * 8);
    grab((void **)&nb.pose, (size_t)cap_records * 56);
    grab((void **)&nb.xyh, (size_t)cap_records * 32);
    grab((void **)&nb.counts, (size_t)cap_records * 4);
    grab((void **)&nb.topk, (size_t)blocks * 32 * sizeof(unsigned long long));
    const int64_t L = ctx->db_desc ? ctx->db_records : 0, T = ctx->db_desc ? ctx->db_rows : 0;
    auto copy = [&](void *d, const void *s_, size_t bytes) {
        if (e == hipSuccess && bytes) e = hipMemcpyAsync(d, s_, bytes, hipMemcpyDeviceToDevice, ctx->stream);
    };
    if (L > 0) {
        copy(nb.desc, ctx->db_desc, (size_t)T * 32);
        copy(nb.pts3d, ctx->db_pts3d, (size_t)T * 12);
        copy(nb.kp2d, ctx->db_kp2d, (size_t)T * 8);
        copy(nb.off, ctx->db_off, (size_t)(L + 1) * 8);
        copy(nb.pose, ctx->db_pose, (size_t)L * 56);
        copy(nb.xyh, ctx->db_xy_heading, (size_t)L * 32);
    } else if (e == hipSuccess) {
        e = hipMemsetAsync(nb.off, 0, 8, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        nb.release();
        (void)hipGetLastError();          // a failed hipMalloc leaves a sticky error that the next launch check would report
        reloc_set_error("database reserve (%lld records, %lld rows) failed: %s", (long long)cap_records, (long long)cap_rows,
                        hipGetErrorString(e));
        return RELOC_E_HIP;
    }
    DbShare *fresh = new (std::nothrow) DbShare();
    if (!fresh) { nb.release(); reloc_set_error("database reserve: out of host memory"); return RELOC_E_HIP; }
    if (ctx->db_counts) (void)hipFree(ctx->db_counts);
    if (ctx->topk_part) (void)hipFree(ctx->topk_part);
    db_arrays_drop(ctx->db_share, ctx->db_desc, ctx->db_pts3d, ctx->db_kp2d, ctx->db_off, ctx->db_pose, ctx->db_xy_heading);
    ctx->db_share = fresh;
    ctx->db_desc = nb.desc; ctx->db_pts3d = nb.pts3d; ctx->db_kp2d = nb.kp2d; ctx->db_off = nb.off; ctx->db_pose = nb.pose;
    ctx->db_xy_heading = nb.xyh; ctx->db_counts = nb.counts; ctx->topk_part = nb.topk;
    ctx->topk_blocks = blocks;
    ctx->db_cap_records = cap_records;
    ctx->db_cap_rows = cap_rows;
    ctx->db_records = L;
    ctx->db_rows = T;
    return RELOC_OK;
}

RELOC_API int reloc_db_reserve(reloc_ctx *ctx, int64_t cap_records, int64_t cap_rows)
{
    ARG_CHECK_CTX(ctx, cap_records >= 0 && cap_rows >= 0, "reloc_db_reserve");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return db_reserve(ctx, cap_records, cap_rows);
}

RELOC_API int reloc_db_upload(reloc_ctx *ctx, const uint8_t *desc, const float *pts3d, const int64_t *offsets,
                              const double *poses, int64_t n_records)
{
    ARG_CHECK_CTX(ctx, offsets && n_records >= 0, "reloc_db_upload");
    const int64_t T = offsets[n_records];
    ARG_CHECK(offsets[0] == 0 && T >= 0, "offsets must start at 0 and be non-decreasing");
    int maxrows = 0;
    for (int64_t r = 0; r < n_records; ++r) {
        const int64_t n = offsets[r + 1] - offsets[r];
        ARG_CHECK(n >= 0, "offsets must be non-decreasing");
        if (n > MAX_REC_ROWS) { reloc_set_error("record %lld has %lld rows (max %d)", (long long)r, (long long)n, MAX_REC_ROWS); return RELOC_E_CAPACITY; }
        if (n > maxrows) maxrows = (int)n;
    }
    ARG_CHECK(T == 0 || (desc && pts3d), "desc / pts3d missing");
    ARG_CHECK(n_records == 0 || poses, "poses missing");
    if (n_records > MAX_DB_RECORDS) { reloc_set_error("database: %lld records (max %lld)", (long long)n_records, (long long)MAX_DB_RECORDS); return RELOC_E_CAPACITY; }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    db_unshare(ctx);
    if (ctx->db_share && __atomic_load_n(&ctx->db_share->refs, __ATOMIC_ACQUIRE) > 1) {
        // other contexts adopted these arrays: they keep them as they are, this upload goes into fresh ones
        db_arrays_drop(ctx->db_share, ctx->db_desc, ctx->db_pts3d, ctx->db_kp2d, ctx->db_off, ctx->db_pose, ctx->db_xy_heading);
        ctx->db_cap_records = ctx->db_cap_rows = 0;
    }
    // from here on the ctx holds no database until everything below has succeeded
    ctx->db_records = 0;
    ctx->db_rows = 0;
    ctx->db_max_rows = 0;
    int rc = db_reserve(ctx, n_records, T);
    if (rc) return rc;
    if (T > 0) {
        HIP_TRY(hipMemcpyAsync(ctx->db_desc, desc, (size_t)T * 32, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(ctx->db_pts3d, pts3d, (size_t)T * 12, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemsetAsync(ctx->db_kp2d, 0, (size_t)T * 8, ctx->stream));
    }
    HIP_TRY(hipMemcpyAsync(ctx->db_off, offsets, (size_t)(n_records + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    if (n_records > 0) {
        HIP_TRY(hipMemcpyAsync(ctx->db_pose, poses, (size_t)n_records * 56, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_db_index, dim3((unsigned)((n_records + 255) / 256)), dim3(256), 0, ctx->stream, ctx->db_pose, (int64_t)0,
                           n_records, ctx->b2c_R[0], ctx->b2c_R[1], ctx->b2c_R[2], ctx->db_xy_heading);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->db_records = n_records;
    ctx->db_rows = T;
    ctx->db_max_rows = maxrows;
    return RELOC_OK;
}

// room for one more record of n rows; geometric growth when the reserve is exhausted
static int db_make_room(reloc_ctx *ctx, int64_t n)
{
    if (ctx->db_desc && ctx->db_records + 1 <= ctx->db_cap_records && ctx->db_rows + n <= ctx->db_cap_rows) return RELOC_OK;
    const int64_t need_r = ctx->db_records + 1, need_t = ctx->db_rows + n;
    int64_t cr = ctx->db_cap_records + ctx->db_cap_records / 2 + 64, ct = ctx->db_cap_rows + ctx->db_cap_rows / 2 + 64 * 512;
    if (cr < need_r) cr = need_r;
    if (ct < need_t) ct = need_t;
    if (cr > MAX_DB_RECORDS) cr = MAX_DB_RECORDS;
    if (need_r > MAX_DB_RECORDS) { reloc_set_error("database: more than %lld records", (long long)MAX_DB_RECORDS); return RELOC_E_CAPACITY; }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return db_reserve(ctx, cr, ct);
}

RELOC_API int reloc_db_append(reloc_ctx *ctx, const uint8_t *desc, const float *pts3d, const float *kp2d, int n,
                              const double pose[7], const double index_xy[2])
{
    ARG_CHECK_CTX(ctx, n >= 0 && pose && (n == 0 || (desc && pts3d)), "reloc_db_append");
    if (ctx->db_shared) { reloc_set_error("the selected database is shared from another context (read-only here)"); return RELOC_E_STATE; }
    if (n > MAX_REC_ROWS) { reloc_set_error("record has %d rows (max %d)", n, MAX_REC_ROWS); return RELOC_E_CAPACITY; }
    int rc = db_make_room(ctx, n);
    if (rc) return rc;
    const int64_t L = ctx->db_records, T = ctx->db_rows;
    if (n > 0) {
        HIP_TRY(hipMemcpyAsync(ctx->db_desc + T * 32, desc, (size_t)n * 32, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(ctx->db_pts3d + T * 3, pts3d, (size_t)n * 12, hipMemcpyHostToDevice, ctx->stream));
        if (kp2d) HIP_TRY(hipMemcpyAsync(ctx->db_kp2d + T * 2, kp2d, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
        else HIP_TRY(hipMemsetAsync(ctx->db_kp2d + T * 2, 0, (size_t)n * 8, ctx->stream));
    }
    const int64_t end = T + n;
    HIP_TRY(hipMemcpyAsync(ctx->db_off + L + 1, &end, 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->db_pose + 7 * L, pose, 56, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_db_index, dim3(1), dim3(64), 0, ctx->stream, ctx->db_pose, L, (int64_t)1, ctx->b2c_R[0], ctx->b2c_R[1],
                       ctx->b2c_R[2], ctx->db_xy_heading);
    HIP_TRY(hipGetLastError());
    if (index_xy) HIP_TRY(hipMemcpyAsync(ctx->db_xy_heading + 4 * L, index_xy, 16, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));     // the host sources may go away; the record is visible from here on
    ctx->db_records = L + 1;
    ctx->db_rows = end;
    if (n > ctx->db_max_rows) ctx->db_max_rows = n;
    return RELOC_OK;
}

static void db_store_slot(reloc_ctx *ctx)
{
    DbArena &a = ctx->db_slot[ctx->db_sel];
    a.cap_records = ctx->db_cap_records; a.cap_rows = ctx->db_cap_rows; a.records = ctx->db_records; a.rows = ctx->db_rows;
    a.max_rows = ctx->db_max_rows; a.desc = ctx->db_desc; a.pts3d = ctx->db_pts3d; a.kp2d = ctx->db_kp2d; a.off = ctx->db_off;
    a.pose = ctx->db_pose; a.xy_heading = ctx->db_xy_heading; a.counts = ctx->db_counts; a.topk_part = ctx->topk_part;
    a.topk_blocks = ctx->topk_blocks;
    a.share = ctx->db_share;
}

RELOC_API int reloc_db_share(reloc_ctx *dst, reloc_ctx *src)
{
    ARG_CHECK_CTX(dst, src && src != dst, "reloc_db_share");
    if (src->device != dst->device) { reloc_set_error("reloc_db_share: contexts live on different devices"); return RELOC_E_ARG; }
    if (!db_ready(src)) { reloc_set_error("reloc_db_share: the source context has no database"); return RELOC_E_STATE; }
    HIP_TRY(hipStreamSynchronize(dst->stream));
    HIP_TRY(hipStreamSynchronize(src->stream));
    int32_t *counts = nullptr;
    unsigned long long *topk = nullptr;
    const int blocks = (int)((src->db_cap_records + 1023) / 1024);
    if (hipMalloc((void **)&counts, (size_t)src->db_cap_records * 4) != hipSuccess ||
        hipMalloc((void **)&topk, (size_t)blocks * 32 * sizeof(unsigned long long)) != hipSuccess) {
        if (counts) (void)hipFree(counts);
        reloc_set_error("reloc_db_share: scratch allocation failed");
        return RELOC_E_HIP;
    }
    if (dst->db_shared) db_unshare(dst);
    else {
        if (dst->db_counts) (void)hipFree(dst->db_counts);
        if (dst->topk_part) (void)hipFree(dst->topk_part);
        db_arrays_drop(dst->db_share, dst->db_desc, dst->db_pts3d, dst->db_kp2d, dst->db_off, dst->db_pose, dst->db_xy_heading);
    }
    __atomic_add_fetch(&src->db_share->refs, 1, __ATOMIC_ACQ_REL);
    dst->db_share = src->db_share;
    dst->db_desc = src->db_desc; dst->db_pts3d = src->db_pts3d; dst->db_kp2d = src->db_kp2d; dst->db_off = src->db_off;
    dst->db_pose = src->db_pose; dst->db_xy_heading = src->db_xy_heading;
    dst->db_counts = counts; dst->topk_part = topk; dst->topk_blocks = blocks;
    dst->db_records = src->db_records; dst->db_rows = src->db_rows; dst->db_max_rows = src->db_max_rows;
    dst->db_cap_records = src->db_cap_records; dst->db_cap_rows = src->db_cap_rows;
    dst->db_shared = true;
    return RELOC_OK;
}

RELOC_API int reloc_db_select(reloc_ctx *ctx, int slot)
{
    ARG_CHECK_CTX(ctx, slot == 0 || slot == 1, "reloc_db_select: slot must be 0 or 1");
    if (slot == ctx->db_sel) return RELOC_OK;
    if (ctx->db_shared) { reloc_set_error("reloc_db_select: the selected database is shared; upload or share per slot instead"); return RELOC_E_STATE; }
    db_store_slot(ctx);
    const DbArena &a = ctx->db_slot[slot];
    ctx->db_sel = slot;
    ctx->db_cap_records = a.cap_records; ctx->db_cap_rows = a.cap_rows; ctx->db_records = a.records; ctx->db_rows = a.rows;
    ctx->db_max_rows = a.max_rows; ctx->db_desc = a.desc; ctx->db_pts3d = a.pts3d; ctx->db_kp2d = a.kp2d; ctx->db_off = a.off;
    ctx->db_pose = a.pose; ctx->db_xy_heading = a.xy_heading; ctx->db_counts = a.counts; ctx->topk_part = a.topk_part;
    ctx->topk_blocks = a.topk_blocks;
    ctx->db_share = a.share;
    return RELOC_OK;
}

RELOC_API int reloc_db_fetch(reloc_ctx *ctx, int64_t record, uint8_t *desc, float *pts3d, float *kp2d, double pose[7],
                             double index_xyh[4], int32_t *n)
{
    ARG_CHECK_CTX(ctx, record >= 0, "reloc_db_fetch");
    if (!db_ready(ctx) || record >= ctx->db_records) { reloc_set_error("db fetch: record %lld of %lld", (long long)record, (long long)ctx->db_records); return RELOC_E_STATE; }
    int64_t o[2];
    HIP_TRY(hipMemcpyAsync(o, ctx->db_off + record, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const int64_t cnt = o[1] - o[0];
    if (n) *n = (int32_t)cnt;
    if (cnt > 0) {
        if (desc) HIP_TRY(hipMemcpyAsync(desc, ctx->db_desc + o[0] * 32, (size_t)cnt * 32, hipMemcpyDeviceToHost, ctx->stream));
        if (pts3d) HIP_TRY(hipMemcpyAsync(pts3d, ctx->db_pts3d + o[0] * 3, (size_t)cnt * 12, hipMemcpyDeviceToHost, ctx->stream));
        if (kp2d) HIP_TRY(hipMemcpyAsync(kp2d, ctx->db_kp2d + o[0] * 2, (size_t)cnt * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (pose) HIP_TRY(hipMemcpyAsync(pose, ctx->db_pose + 7 * record, 56, hipMemcpyDeviceToHost, ctx->stream));
    if (index_xyh) HIP_TRY(hipMemcpyAsync(index_xyh, ctx->db_xy_heading + 4 * record, 32, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RELOC_OK;
}

RELOC_API int64_t reloc_db_records(reloc_ctx *ctx) { return ctx ? ctx->db_records : -1; }
RELOC_API int64_t reloc_db_rows(reloc_ctx *ctx) { return ctx ? ctx->db_rows : -1; }

RELOC_API int reloc_db_match_counts_dev(reloc_ctx *ctx, const uint8_t *cur_dev, const int32_t *n_cur_dev, int n_cur_max,
                                        int32_t *counts_dev)
{
    ARG_CHECK_CTX(ctx, cur_dev && counts_dev && n_cur_max >= 0, "reloc_db_match_counts_dev");
    if (!db_ready(ctx)) { reloc_set_error("no database uploaded"); return RELOC_E_STATE; }
    reloc_prof_begin(ctx, RELOC_PROF_DB_SCAN);
    int rc = launch_db_scan(ctx, ctx->db_desc, ctx->db_off, ctx->db_records, nullptr, nullptr, (int)ctx->db_records, cur_dev,
                            n_cur_dev, n_cur_max, ctx->db_max_rows, counts_dev, nullptr, nullptr, nullptr, nullptr, 0);
    reloc_prof_end(ctx, RELOC_PROF_DB_SCAN);

    return rc;
}

RELOC_API int reloc_db_ratio_counts(reloc_ctx *ctx, const uint8_t *cur, int n_cur, double ratio, int32_t *counts)
{
    ARG_CHECK_CTX(ctx, counts && n_cur >= 0 && (n_cur == 0 || cur) && ratio > 0, "reloc_db_ratio_counts");
    if (!db_ready(ctx)) { reloc_set_error("no database uploaded"); return RELOC_E_STATE; }
    if (n_cur == 0) { memset(counts, 0, (size_t)ctx->db_records * 4); return RELOC_OK; }
    if (n_cur > 65535) { reloc_set_error("ratio scan: more than 65535 current descriptors"); return RELOC_E_CAPACITY; }
    void *dc;
    int rc;
    if ((rc = reloc_scratch(ctx, 0, (int64_t)n_cur * 32, &dc))) return rc;
    HIP_TRY(hipMemcpyAsync(dc, cur, (size_t)n_cur * 32, hipMemcpyHostToDevice, ctx->stream));
    int grid = ctx->num_cu * 4;
    if (grid > ctx->db_records) grid = (int)ctx->db_records;
    hipLaunchKernelGGL(k_db_ratio, dim3(grid), dim3(256), 0, ctx->stream, (const uint4 *)ctx->db_desc, ctx->db_off,
                       (int)ctx->db_records, (const uint4 *)dc, (const int32_t *)nullptr, n_cur, ratio, ctx->db_counts);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(counts, ctx->db_counts, (size_t)ctx->db_records * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RELOC_OK;
}

RELOC_API int reloc_db_match_counts(reloc_ctx *ctx, const uint8_t *cur, int n_cur, int32_t *counts)
{
    ARG_CHECK_CTX(ctx, counts && n_cur >= 0 && (n_cur == 0 || cur), "reloc_db_match_counts");
    if (!db_ready(ctx)) { reloc_set_error("no database uploaded"); return RELOC_E_STATE; }
    if (n_cur == 0) { memset(counts, 0, (size_t)ctx->db_records * 4); return RELOC_OK; }
    void *dc;
    int rc;
    if ((rc = reloc_scratch(ctx, 0, (int64_t)n_cur * 32, &dc))) return rc;
    HIP_TRY(hipMemcpyAsync(dc, cur, (size_t)n_cur * 32, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = reloc_db_match_counts_dev(ctx, (const uint8_t *)dc, nullptr, n_cur, ctx->db_counts))) return rc;
    HIP_TRY(hipMemcpyAsync(counts, ctx->db_counts, (size_t)ctx->db_records * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RELOC_OK;
}
