// tools/exp_hamming_mfma.hip -- OFF-CONTRACT micro-benchmark, never linked into libreloc_hip.so.
//
// BASELINE.json's north star rules MFMA out for the Hamming stage ("bitwise, not a dense contraction"), and the product's
// scan (k_db_scan) computes 256-bit distances as 8 v_xor + 8 v_bcnt on the VALU, where it sits at the issue-rate wall of that
// formulation.  This file answers what that clause costs: the SAME per-record mutual-match count, with
// the distances taken from the matrix cores.  For 0/1 vectors  Hamming(a, b) = |a| + |b| - 2 a.b  is exact in integers, so
//   A = teach rows, bits expanded to i8 0/1 on the fly (v_bfe + v_mul_u32_u24 + v_and per 4 bits), plus one augmented k-step
//       that carries |a| (three parts <= 127) and three ones;
//   B = the 512 current descriptors, expanded ONCE to i8 0/-2 plus the augmented step (three ones, |b| in three parts),
//       resident in LDS in fragment order (144 KB);
//   D = A.B over K = 256 + 32 is the distance matrix itself: 9 x v_mfma_i32_32x32x32_i8 per 32 x 32 tile, accumulator
//       started from 0, no VALU work to form a distance.
// One wave owns a 64-row record (two row tiles) and walks the 16 column tiles; the argmin bookkeeping is the product's
// (one key per pair and direction: distance << 9 | column for a row's best column, distance << 6 | row for a column's best
// row, lowest index on ties), the 32 row keys of a lane are reduced over the 32 column lanes by a DPP / permlane butterfly
// once per record, mutual pairs are counted through a 1 KB LDS array per wave.  tools/exp_hamming_mfma.py checks the counts
// against reloc_db_match_counts (k_db_scan) record by record and reports pairs/s of both.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32;
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int NQ = 512;            // current descriptors (columns), 16 tiles of 32
constexpr int NCT = NQ / 32;
constexpr int NS = 9;              // k-steps of 32: 8 of descriptor bits + 1 augmented
constexpr int B_BYTES = NCT * NS * 64 * 16;      // 147 456: B in fragment order [ct][s][lane][16]
constexpr int WAVES = 8;

// 4 bits (bit positions p .. p + 3 of w) -> 4 bytes of 0 / 1:  n * 0x204081 puts bit i of n at bits i, i + 7, i + 14, i + 21
// (all 16 positions distinct: no carries); bit 8j of the product is bit j of n
__device__ __forceinline__ u32 spread4(u32 w, int p)
{
    const u32 n = (w >> p) & 0xFu;
    return __umul24(n, 0x00204081u) & 0x01010101u;
}
// element j (0..15) of lane half h at k-step s = bit (16 h + j) of descriptor word s
__device__ __forceinline__ v4i expand16(u32 w, int h)
{
    const u32 x = h ? (w >> 16) : (w & 0xFFFFu);
    v4i r;
    r.x = (int)spread4(x, 0); r.y = (int)spread4(x, 4); r.z = (int)spread4(x, 8); r.w = (int)spread4(x, 12);
    return r;
}
__device__ __forceinline__ u32 part127(int v, int i) { const int p = v - 127 * i; return (u32)(p < 0 ? 0 : (p > 127 ? 127 : p)); }

// B image: [ct][s][lane][16 bytes]; lane = (c = lane & 31, h = lane >> 5): element j = -2 * bit (16 h + j) of word s of
// descriptor 32 ct + c;  s = 8, h = 0: elements 0..2 = 1, elements 3..5 = |b| in three parts;  h = 1: zeros
extern "C" __global__ void k_expand_b(const u32 *__restrict__ cur, uint8_t *__restrict__ img)
{
    const int ct = blockIdx.x, lane = threadIdx.x & 63, s = threadIdx.x >> 6;       // 9 waves
    const int c = lane & 31, h = lane >> 5;
    const u32 *d = cur + (size_t)(32 * ct + c) * 8;
    v4i v;
    if (s < 8) {
        v = expand16(d[s], h);
        v.x = (int)((u32)v.x * 0xFEu); v.y = (int)((u32)v.y * 0xFEu); v.z = (int)((u32)v.z * 0xFEu); v.w = (int)((u32)v.w * 0xFEu);   // 1 -> 0xFE = -2 per byte (bytes are 0 / 1: no carries)
    } else {
        int pc = 0;
        for (int k = 0; k < 8; ++k) pc += __popc(d[k]);
        v = v4i{0, 0, 0, 0};
        if (h == 0) {
            v.x = (int)(0x00010101u | (part127(pc, 0) << 24));
            v.y = (int)(part127(pc, 1) | (part127(pc, 2) << 8));
        }
    }
    *reinterpret_cast<v4i *>(img + ((size_t)(ct * NS + s) * 64 + lane) * 16) = v;
}

__device__ __forceinline__ u32 min16(u32 a, u32 b)
{
    u32 r;
    asm("v_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
#define DPP_MIN(dst, src, ctrl, bank) asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %1, %1 " ctrl " row_mask:0xf bank_mask:" bank : "+v"(dst) : "v"(src))

// grid: any; block: 512 (8 waves, one workgroup per CU: 144 KB of LDS hold B).  counts[r] = mutual nearest-neighbour pairs
// between the 64 rows of record r and the 512 columns.
extern "C" __global__ __launch_bounds__(512) void k_mfma_scan(const uint4 *__restrict__ db, int n_rec, const uint8_t *__restrict__ bimg,
                                                               int32_t *__restrict__ counts)
{
    extern __shared__ __align__(16) uint8_t lds[];
    uint8_t *sB = lds;                                               // B_BYTES
    unsigned short *sCol = (unsigned short *)(lds + B_BYTES);        // [wave][512]: distance << 6 | row of each column's best row
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < B_BYTES / 16; i += 512) reinterpret_cast<uint4 *>(sB)[i] = reinterpret_cast<const uint4 *>(bimg)[i];
    __syncthreads();
    const int c = lane & 31, h = lane >> 5;
    unsigned short *col = sCol + wave * NQ;
    const int gw = blockIdx.x * WAVES + wave, nw = gridDim.x * WAVES;
    for (int r = gw; r < n_rec; r += nw) {
        // A fragments of the record's two row tiles: lane (c, h) expands bits of rows c and 32 + c
        v4i A[2][NS];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const uint4 lo = db[2 * ((size_t)r * 64 + 32 * t + c)], hi = db[2 * ((size_t)r * 64 + 32 * t + c) + 1];
            const u32 w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            int pc = 0;
#pragma unroll
            for (int s = 0; s < 8; ++s) { A[t][s] = expand16(w[s], h); pc += __popc(w[s]); }
            A[t][8] = v4i{0, 0, 0, 0};
            if (h == 0) {
                A[t][8].x = (int)(part127(pc, 0) | (part127(pc, 1) << 8) | (part127(pc, 2) << 16) | 0x01000000u);
                A[t][8].y = 0x00000101;
            }
        }
        // per row of this lane: best (distance << 4 | column tile) so far -- the lane's 16 columns differ in the tile only, so
        // the key fits 16 bits (v_min_u16: full rate); it is widened to distance << 9 | column once per record
        u32 rk[2][16];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 16; ++g) rk[t][g] = 0xFFFFu;
        // the 18 MFMAs of column tile ct into (p0, p1)
        auto mfmas = [&](int ct, v16i &p0, v16i &p1) {
            const uint8_t *bp = sB + ((size_t)ct * NS * 64 + lane) * 16;
            p0 = v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            p1 = p0;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const v4i b = *reinterpret_cast<const v4i *>(bp + s * 64 * 16);
                p0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[0][s], b, p0, 0, 0, 0);
                p1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[1][s], b, p1, 0, 0, 0);
            }
        };
        // bookkeeping of column tile ct: p_t[g] = distance(row 32 t + (g & 3) + 8 (g >> 2) + 4 h, column 32 ct + c)
        auto epilogue = [&](int ct, const v16i &p0, const v16i &p1) {
            u32 cb = 0xFFFFu;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const u32 d0 = (u32)p0[g], d1 = (u32)p1[g];
                rk[0][g] = min16(rk[0][g], (d0 << 4) | (u32)ct);
                rk[1][g] = min16(rk[1][g], (d1 << 4) | (u32)ct);
                cb = min16(cb, (d0 << 6) | (u32)((g & 3) + 8 * (g >> 2)));
                cb = min16(cb, (d1 << 6) | (u32)(32 + (g & 3) + 8 * (g >> 2)));
            }
            cb |= (u32)(4 * h);                                      // the row constants have bit 2 clear
            {
                const auto sw = __builtin_amdgcn_permlane32_swap(cb, cb, false, false);
                cb = sw[0] < sw[1] ? sw[0] : sw[1];
            }
            if (h == 0) col[32 * ct + c] = (unsigned short)cb;
        };
        // Software pipeline: the MFMAs of tile ct + 1 are in program order IN FRONT of the bookkeeping of tile ct and independent
        // of it; the scheduling hints interleave them (1 LDS read, 2 MFMA, 14 VALU, nine times): an in-order wave that issued its
        // 18 MFMAs back to back would sit out 18 x 32 cycles of matrix pipe before its first vector instruction.
#define INTERLEAVE()                                                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < NS; ++i_) {                              \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                           \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                           \
        __builtin_amdgcn_sched_group_barrier(0x002, 14, 0);                          \
    }
        v16i P0, P1, Q0, Q1;
        mfmas(0, P0, P1);
#pragma unroll 1
        for (int ct = 0; ct < NCT; ct += 2) {
            mfmas(ct + 1, Q0, Q1);
            epilogue(ct, P0, P1);
            INTERLEAVE()
            mfmas(ct + 2 < NCT ? ct + 2 : 0, P0, P1);                  // (the last one is a dummy: keeps the loop body uniform)
            epilogue(ct + 1, Q0, Q1);
            INTERLEAVE()
        }
#undef INTERLEAVE
        // widen the row keys: distance << 9 | column
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 16; ++g) rk[t][g] = ((rk[t][g] >> 4) << 9) | ((rk[t][g] & 15u) << 5) | (u32)c;
        // this lane's 32 row keys, each a minimum over the lane's 16 columns: reduce over the 32 column lanes (bits 0..4).
        // Rows: (t, g) <-> vector index v = 16 t + g.  Nodes on lane bit 2 (row_shl / row_shr 4) pair v with v + 16, on bit 3
        // (row_ror 8) v with v + 8, on bit 4 (permlane16 swap) v with v + 4; the four vectors left are reduced over lane
        // bits 0, 1 with two plain butterfly nodes each.  Afterwards lane l holds the key of vector
        //   v(l) = 16 b2 + 8 b3 + 4 b4 + 2 b1' ... (see below), i.e. of ONE row; both halves hold different rows (their own h).
        u32 x[16];
#pragma unroll
        for (int g = 0; g < 16; ++g) {                               // bit 2: vector g (lanes with bit 2 clear) vs 16 + g
            x[g] = rk[0][g];
            DPP_MIN(x[g], rk[0][g], "row_shl:4", "0x5");
            u32 y = rk[1][g];
            asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xa" : "+v"(x[g]) : "v"(y));
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {                                // bit 3: g vs g + 8
            u32 a = x[g], b = x[g + 8];
            asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                         "v_min_u32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc" : "+v"(a) : "v"(b));
            x[g] = a;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {                                // bit 4: g vs g + 4
            const auto sw = __builtin_amdgcn_permlane16_swap(x[g], x[g + 4], false, false);
            x[g] = sw[0] < sw[1] ? sw[0] : sw[1];
        }
        // bits 1 and 0: plain nodes (select + quad_perm exchange)
        auto node = [&](u32 a, u32 b, int bit) -> u32 {
            const bool hi = lane & bit;
            const u32 mine = hi ? b : a, theirs = hi ? a : b;
            const u32 o = bit == 1 ? (u32)__builtin_amdgcn_update_dpp(0, (int)theirs, 0xB1, 0xF, 0xF, false)
                                   : (u32)__builtin_amdgcn_update_dpp(0, (int)theirs, 0x4E, 0xF, 0xF, false);
            return mine < o ? mine : o;
        };
        const u32 y0 = node(x[0], x[2], 2), y1 = node(x[1], x[3], 2);   // bit 1: g vs g + 2
        const u32 key = node(y0, y1, 1);                               // bit 0: g vs g + 1
        // the row this lane ended up with: t = bit 2, g = 8 b3 + 4 b4 + 2 b1 + b0
        const int t_ = (lane >> 2) & 1, g_ = 8 * ((lane >> 3) & 1) + 4 * ((lane >> 4) & 1) + 2 * ((lane >> 1) & 1) + (lane & 1);
        const u32 my_row = (u32)(32 * t_ + (g_ & 3) + 8 * (g_ >> 2) + 4 * h);
        const u32 q = key & 511u;
        __builtin_amdgcn_s_waitcnt(0xC07F);                          // lgkmcnt(0): the wave's own LDS stores of col[] have landed
        const bool mutual = ((u32)col[q] & 63u) == my_row;
        const int total = __popcll(__ballot(mutual));
        if (lane == 0) counts[r] = total;
    }
}

extern "C" int mfma_scan_lds_bytes() { return B_BYTES + WAVES * NQ * 2; }

// host entry points for tools/exp_hamming_mfma.py (device pointers in, milliseconds out)
extern "C" int mfma_expand_b(const void *cur_dev, void *img_dev)
{
    hipLaunchKernelGGL(k_expand_b, dim3(NCT), dim3(64 * NS), 0, 0, (const u32 *)cur_dev, (uint8_t *)img_dev);
    return (int)hipDeviceSynchronize();
}
extern "C" float mfma_scan(const void *db_dev, int n_rec, const void *img_dev, void *counts_dev, int grid, int reps)
{
    const int lds = mfma_scan_lds_bytes();
    if (hipFuncSetAttribute((const void *)k_mfma_scan, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -1.f;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL(k_mfma_scan, dim3(grid), dim3(512), lds, 0, (const uint4 *)db_dev, n_rec, (const uint8_t *)img_dev, (int32_t *)counts_dev);
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL(k_mfma_scan, dim3(grid), dim3(512), lds, 0, (const uint4 *)db_dev, n_rec, (const uint8_t *)img_dev, (int32_t *)counts_dev);
    hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) return -2.f;
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms / reps;
}
