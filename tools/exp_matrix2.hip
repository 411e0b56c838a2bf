// Developer experiment (round 2): the production u16 distance-matrix kernel (persistent grid, A rows prefetched through
// the scalar cache, 8 B rows per lane, one 16-byte store per lane and A row) with its two halves separated and a few
// variants, to see what the VALU part and the store stream cost on their own and how well they overlap.
//   hipcc --offload-arch=gfx950 -O3 -o tools/exp_matrix2 tools/exp_matrix2.hip && ./tools/exp_matrix2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef uint32_t u32;
__device__ __forceinline__ u32 bcnt_acc(u32 x, u32 acc) { u32 r; asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc)); return r; }
__device__ __forceinline__ void srow_landed(u32 first_dword) { asm volatile("" ::"s"(first_dword)); }
__device__ __forceinline__ u32 ham8(const u32 q[8], const uint4 a, const uint4 b, u32 init)
{
    u32 acc = init;
    acc = bcnt_acc(q[0] ^ a.x, acc); acc = bcnt_acc(q[1] ^ a.y, acc); acc = bcnt_acc(q[2] ^ a.z, acc); acc = bcnt_acc(q[3] ^ a.w, acc);
    acc = bcnt_acc(q[4] ^ b.x, acc); acc = bcnt_acc(q[5] ^ b.y, acc); acc = bcnt_acc(q[6] ^ b.z, acc); acc = bcnt_acc(q[7] ^ b.w, acc);
    return acc;
}

// one word of the 8 chains as ONE asm block (no compiler-inserted s_nop): PRE = text between an xor and its bcnt,
// POST = text after the bcnt
#define WORD_BLOCK(PRE, POST, ACC_IN)                                                                                      \
    asm volatile("v_xor_b32 %8, %16, %17\n\t" PRE "v_bcnt_u32_b32 %0, %8, " ACC_IN("%0") "\n\t" POST                        \
                 "v_xor_b32 %9, %16, %18\n\t" PRE "v_bcnt_u32_b32 %1, %9, " ACC_IN("%1") "\n\t" POST                        \
                 "v_xor_b32 %10, %16, %19\n\t" PRE "v_bcnt_u32_b32 %2, %10, " ACC_IN("%2") "\n\t" POST                      \
                 "v_xor_b32 %11, %16, %20\n\t" PRE "v_bcnt_u32_b32 %3, %11, " ACC_IN("%3") "\n\t" POST                      \
                 "v_xor_b32 %12, %16, %21\n\t" PRE "v_bcnt_u32_b32 %4, %12, " ACC_IN("%4") "\n\t" POST                      \
                 "v_xor_b32 %13, %16, %22\n\t" PRE "v_bcnt_u32_b32 %5, %13, " ACC_IN("%5") "\n\t" POST                      \
                 "v_xor_b32 %14, %16, %23\n\t" PRE "v_bcnt_u32_b32 %6, %14, " ACC_IN("%6") "\n\t" POST                      \
                 "v_xor_b32 %15, %16, %24\n\t" PRE "v_bcnt_u32_b32 %7, %15, " ACC_IN("%7") "\n\t" POST                      \
                 : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(o[7]),          \
                   "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7)                  \
                 : "s"(rw[q]), "v"(b[0][q]), "v"(b[1][q]), "v"(b[2][q]), "v"(b[3][q]), "v"(b[4][q]), "v"(b[5][q]), "v"(b[6][q]), \
                   "v"(b[7][q]))
#define ACC_SELF(r) r

// the same with ONE temporary for all eight xor results (what the compiler does with a statement per instruction)
#define WORD_BLOCK1(PRE, POST)                                                                                             \
    asm volatile("v_xor_b32 %8, %9, %10\n\t" PRE "v_bcnt_u32_b32 %0, %8, %0\n\t" POST                                      \
                 "v_xor_b32 %8, %9, %11\n\t" PRE "v_bcnt_u32_b32 %1, %8, %1\n\t" POST                                      \
                 "v_xor_b32 %8, %9, %12\n\t" PRE "v_bcnt_u32_b32 %2, %8, %2\n\t" POST                                      \
                 "v_xor_b32 %8, %9, %13\n\t" PRE "v_bcnt_u32_b32 %3, %8, %3\n\t" POST                                      \
                 "v_xor_b32 %8, %9, %14\n\t" PRE "v_bcnt_u32_b32 %4, %8, %4\n\t" POST                                      \
                 "v_xor_b32 %8, %9, %15\n\t" PRE "v_bcnt_u32_b32 %5, %8, %5\n\t" POST                                      \
                 "v_xor_b32 %8, %9, %16\n\t" PRE "v_bcnt_u32_b32 %6, %8, %6\n\t" POST                                      \
                 "v_xor_b32 %8, %9, %17\n\t" PRE "v_bcnt_u32_b32 %7, %8, %7\n\t" POST                                      \
                 : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(o[7]), "=&v"(x0) \
                 : "s"(rw[q]), "v"(b[0][q]), "v"(b[1][q]), "v"(b[2][q]), "v"(b[3][q]), "v"(b[4][q]), "v"(b[5][q]), "v"(b[6][q]), \
                   "v"(b[7][q]))
// MODE 0 full, 1 compute only (store behind a never-true test), 2 store only (no distances)
// ST    0 plain store, 1 nontemporal, 2 sc1 (write-through) via inline asm
// IL    1 serial chains, 2 two packed registers interleaved
template <int MODE, int ST, int IL, int UNIT, int WPS>
__global__ __launch_bounds__(256, WPS) void k(const uint4 *__restrict__ A, int64_t na, const uint4 *__restrict__ B, int64_t nb,
                                              uint16_t *__restrict__ out, int n_col_tiles, int n_units)
{
    const int ct = blockIdx.x % n_col_tiles;
    const int k0 = blockIdx.x / n_col_tiles, kstep = gridDim.x / n_col_tiles;
    const int64_t j0 = ((int64_t)ct * 256 + threadIdx.x) * 8;
    u32 b[8][8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int64_t j = j0 + c < nb ? j0 + c : nb - 1;
        const uint4 lo = B[2 * j], hi = B[2 * j + 1];
        b[c][0] = lo.x; b[c][1] = lo.y; b[c][2] = lo.z; b[c][3] = lo.w; b[c][4] = hi.x; b[c][5] = hi.y; b[c][6] = hi.z; b[c][7] = hi.w;
    }
    if (j0 >= nb) return;
    const int64_t last = na - 1;
    auto row_at = [&](int64_t i) { return i < last ? i : last; };
    int64_t i_first = row_at((int64_t)k0 * UNIT);
    uint4 ra = A[2 * i_first], rb = A[2 * i_first + 1];
    u32 wp[4] = {0, 0, 0, 0};                 // IL == 31: the previous row's packed distances, stored from inside the next row's chains
    char *prow = nullptr;
    const u32 lane_off31 = (u32)(j0 * 2);
    typedef u32 v4u31 __attribute__((ext_vector_type(4)));
    for (int unit = k0; unit < n_units; unit += kstep) {
        const int64_t i0 = (int64_t)unit * UNIT;
#pragma unroll
        for (int e = 0; e < UNIT; ++e) {
            srow_landed(ra.x);
            const int64_t inext = row_at(e + 1 < UNIT ? i0 + e + 1 : i0 + (int64_t)kstep * UNIT);
            uint4 na_, nb_;
            if (MODE == 3) {      // no scalar load in the loop: the row registers are only declared modified
                na_ = ra; nb_ = rb;
                asm volatile("" : "+s"(na_.x), "+s"(na_.y), "+s"(na_.z), "+s"(na_.w), "+s"(nb_.x), "+s"(nb_.y), "+s"(nb_.z), "+s"(nb_.w));
            } else { na_ = A[2 * inext]; nb_ = A[2 * inext + 1]; }
            __builtin_amdgcn_sched_barrier(0);
            u32 w[4];
            if (MODE == 2) {
                w[0] = ra.x ^ b[0][0]; w[1] = ra.y ^ b[1][0]; w[2] = ra.z ^ b[2][0]; w[3] = ra.w ^ b[3][0];
            } else if (IL == 1) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const u32 odd = ham8(b[2 * p + 1], ra, rb, 0);
                    w[p] = ham8(b[2 * p], ra, rb, odd << 16);
                }
            } else if (IL == 31) {
                const u32 rw[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
                u32 o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (q == 2 && prow) {
                        v4u31 vv = {wp[0], wp[1], wp[2], wp[3]};
                        asm volatile("global_store_dwordx4 %0, %1, %2 nt" ::"v"(lane_off31), "v"(vv), "s"(prow) : "memory");
                    }
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        u32 x;
                        asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "s"(rw[q]), "v"(b[c][q]));
                        asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(o[c]) : "v"(x));
                    }
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) { w[p] = o[2 * p] | (o[2 * p + 1] << 16); wp[p] = w[p]; }
                prow = (i0 + e < na) ? reinterpret_cast<char *>(out + (i0 + e) * nb) : nullptr;
            } else if (IL == 8 || IL == 9) {
                // ubench-style: 8 accumulators (one per column), q-major, xor immediately followed by its bcnt, order pinned
                const u32 rw[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
                u32 o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        u32 x;
                        if (IL == 8) {
                            asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "s"(rw[q]), "v"(b[c][q]));
                            asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(o[c]) : "v"(x));
                        } else {
                            x = b[c][q] ^ rw[q];
                            o[c] = bcnt_acc(x, o[c]);
                        }
                    }
#pragma unroll
                for (int p = 0; p < 4; ++p) w[p] = o[2 * p] | (o[2 * p + 1] << 16);
            } else if (IL >= 20 && IL <= 30) {
                const u32 rw[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
                u32 o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                u32 x0, x1, x2, x3, x4, x5, x6, x7;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (IL == 20) WORD_BLOCK("", "", ACC_SELF);
                    if (IL == 21) WORD_BLOCK("s_nop 0\n\t", "", ACC_SELF);
                    if (IL == 22) WORD_BLOCK("s_nop 1\n\t", "", ACC_SELF);
                    if (IL == 23) WORD_BLOCK("", "s_nop 0\n\t", ACC_SELF);
                    if (IL == 24) WORD_BLOCK("s_nop 0\n\t", "s_nop 0\n\t", ACC_SELF);
                    if (IL == 25) WORD_BLOCK("s_nop 2\n\t", "", ACC_SELF);
                    if (IL == 26) WORD_BLOCK("v_nop\n\t", "", ACC_SELF);
                    if (IL == 27) WORD_BLOCK1("s_nop 0\n\t", "");
                    if (IL == 28) WORD_BLOCK1("", "s_nop 0\n\t");
                    if (IL == 29) WORD_BLOCK1("", "");
                    if (IL == 30) WORD_BLOCK1("s_nop 0\n\t", "s_nop 0\n\t");
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) w[p] = o[2 * p] | (o[2 * p + 1] << 16);
            } else if (IL == 16) {
                // 16 accumulators: this row and a copy of it with the words rotated (stands in for a second row's SGPRs):
                // measures whether a longer accumulator reuse distance than 16 instructions helps further
                const u32 rw[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
                u32 o[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        u32 x;
                        asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "s"(rw[(q + (c >> 3)) & 7]), "v"(b[c & 7][q]));
                        asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(o[c]) : "v"(x));
                    }
#pragma unroll
                for (int p = 0; p < 4; ++p) w[p] = (o[2 * p] + o[8 + 2 * p]) | ((o[2 * p + 1] + o[8 + 2 * p + 1]) << 16);
            } else if (IL == 10) {
                // 8 accumulators, xor of the NEXT column issued before the bcnt of the current one (no adjacent dependency)
                const u32 rw[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
                u32 o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                u32 x, xn;
                asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "s"(rw[0]), "v"(b[0][0]));
#pragma unroll
                for (int i = 0; i < 64; ++i) {
                    const int q = i >> 3, c = i & 7, qn = (i + 1) >> 3, cn = (i + 1) & 7;
                    if (i + 1 < 64) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(xn) : "s"(rw[qn]), "v"(b[cn][qn]));
                    asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(o[c]) : "v"(x));
                    x = xn;
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) w[p] = o[2 * p] | (o[2 * p + 1] << 16);
            } else if (IL == 11) {
                // 8 accumulators, column-major inside groups of 4 dwords: c-major over q pairs (acc reuse distance 8)
                const u32 rw[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
                u32 o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int c = 0; c < 8; c += 2) {
                        u32 x0, x1;
                        asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x0) : "s"(rw[q]), "v"(b[c][q]));
                        asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x1) : "s"(rw[q]), "v"(b[c + 1][q]));
                        asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(o[c]) : "v"(x0));
                        asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(o[c + 1]) : "v"(x1));
                    }
#pragma unroll
                for (int p = 0; p < 4; ++p) w[p] = o[2 * p] | (o[2 * p + 1] << 16);
            } else if (IL == 12) {
                // 8 accumulators pinned, odd columns first into the high half: no final or/shift per pair (4 chains x 2 phases)
                const u32 rw[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
                u32 o[4] = {0, 0, 0, 0}, o2[4] = {0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        u32 x;
                        asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "s"(rw[q]), "v"(b[c][q]));
                        if (c & 1) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(o2[c >> 1]) : "v"(x));
                        else asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(o[c >> 1]) : "v"(x));
                    }
#pragma unroll
                for (int p = 0; p < 4; ++p) asm volatile("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(w[p]) : "v"(o2[p]), "v"(o[p]));
            } else if (IL == 4) {
                const u32 rw[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
                u32 o[4] = {0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int p = 0; p < 4; ++p) o[p] = bcnt_acc(b[2 * p + 1][q] ^ rw[q], o[p]);
#pragma unroll
                for (int p = 0; p < 4; ++p) o[p] <<= 16;
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int p = 0; p < 4; ++p) o[p] = bcnt_acc(b[2 * p][q] ^ rw[q], o[p]);
#pragma unroll
                for (int p = 0; p < 4; ++p) w[p] = o[p];
            } else {
                const u32 rw[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
#pragma unroll
                for (int p = 0; p < 4; p += 2) {
                    u32 o0 = 0, o1 = 0;
#pragma unroll
                    for (int q = 0; q < 8; ++q) { o0 = bcnt_acc(b[2 * p + 1][q] ^ rw[q], o0); o1 = bcnt_acc(b[2 * p + 3][q] ^ rw[q], o1); }
                    o0 <<= 16; o1 <<= 16;
#pragma unroll
                    for (int q = 0; q < 8; ++q) { o0 = bcnt_acc(b[2 * p][q] ^ rw[q], o0); o1 = bcnt_acc(b[2 * p + 2][q] ^ rw[q], o1); }
                    w[p] = o0; w[p + 1] = o1;
                }
            }
            const bool doit = IL == 31 ? false : ((MODE == 1 || MODE == 3) ? (w[0] == 0xdeadbeefu && w[1] == 0x12345678u) : (i0 + e < na));
            if (doit) {
                uint16_t *o = out + (i0 + e) * nb + j0;
                const uint4 v = make_uint4(w[0], w[1], w[2], w[3]);
                if (ST == 3) {
                    typedef u32 v4u __attribute__((ext_vector_type(4)));
                    v4u vv = {w[0], w[1], w[2], w[3]};
                    char *row = reinterpret_cast<char *>(out + (i0 + e) * nb);
                    const u32 lane_off = (u32)(j0 * 2);
                    asm volatile("global_store_dwordx4 %0, %1, %2 nt" ::"v"(lane_off), "v"(vv), "s"(row) : "memory");
                }
                else if (ST == 0) *reinterpret_cast<uint4 *>(o) = v;
                else if (ST == 1) {
                    typedef u32 v4u __attribute__((ext_vector_type(4)));
                    v4u vv = {w[0], w[1], w[2], w[3]};
                    __builtin_nontemporal_store(vv, reinterpret_cast<v4u *>(o));
                }
                else {
                    typedef u32 v4u __attribute__((ext_vector_type(4)));
                    v4u vv = {w[0], w[1], w[2], w[3]};
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(o), "v"(vv) : "memory");
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            ra = na_;
            rb = nb_;
        }
    }
    if (IL == 31 && prow) {
        v4u31 vv = {wp[0], wp[1], wp[2], wp[3]};
        asm volatile("global_store_dwordx4 %0, %1, %2 nt" ::"v"(lane_off31), "v"(vv), "s"(prow) : "memory");
    }
}
#include <algorithm>
#include <functional>
#include <string>
struct Variant { std::string name; std::function<float()> once; std::vector<float> t; };
template <int MODE, int ST, int IL, int UNIT, int WPS>
Variant make(const char *name, const uint4 *A, const uint4 *B, uint16_t *out, int64_t F, int64_t K, int per_cu)
{
    const int n_col_tiles = (int)((K + 2047) / 2048);
    const int n_units = (int)((F + UNIT - 1) / UNIT);
    int api = 0; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, k<MODE, ST, IL, UNIT, WPS>, 256, 0);
    if (per_cu <= 0) per_cu = api;
    int grid = 256 * per_cu / n_col_tiles * n_col_tiles; if (grid < n_col_tiles) grid = n_col_tiles;
    char buf[160];
    snprintf(buf, sizeof buf, "%-40s per_cu=%d grid=%5d", name, per_cu, grid);
    Variant v;
    v.name = buf;
    v.once = [=]() {
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        (void)hipEventRecord(a);
        for (int r = 0; r < 4; ++r) hipLaunchKernelGGL((k<MODE, ST, IL, UNIT, WPS>), dim3(grid), dim3(256), 0, 0, A, F, B, K, out, n_col_tiles, n_units);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        (void)hipEventDestroy(a); (void)hipEventDestroy(b);
        return ms / 4;
    };
    return v;
}
int main()
{
    const int64_t F = 20000, K = 20000;
    std::vector<uint32_t> h((size_t)F * 8);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < h.size(); ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)(s >> 16); }
    uint4 *A, *B; uint16_t *out;
    (void)hipMalloc(&A, F * 32); (void)hipMalloc(&B, K * 32); (void)hipMalloc(&out, F * K * 2 + 4096);
    (void)hipMemcpy(A, h.data(), F * 32, hipMemcpyHostToDevice); (void)hipMemcpy(B, h.data(), K * 32, hipMemcpyHostToDevice);
    std::vector<Variant> vs;
    if (getenv("EXP_DEFER")) {
        vs.push_back(make<0, 3, 8, 8, 1>("full nt (sgpr base) 8 chains, store after the row", A, B, out, F, K, 6));
        vs.push_back(make<0, 3, 31, 8, 1>("full nt (sgpr base) 8 chains, store inside the next row", A, B, out, F, K, 6));
        vs.push_back(make<0, 1, 8, 8, 1>("full nt 8 chains, store after the row (flat address)", A, B, out, F, K, 6));
        vs.push_back(make<0, 3, 31, 8, 1>("full nt (sgpr base) 8 chains, store inside the next row", A, B, out, F, K, 7));
        vs.push_back(make<1, 0, 8, 8, 1>("compute-only 8 chains", A, B, out, F, K, 6));
    } else if (getenv("EXP_SPACING")) {
        // what stands between an xor and its dependent bcnt (whole word as one asm block: nothing inserted by the compiler)
        vs.push_back(make<0, 1, 8, 8, 1>("full nt 8 chains, statement per instr (compiler s_nop)", A, B, out, F, K, 6));
        vs.push_back(make<0, 1, 20, 8, 1>("full nt block 8 temps: xor; bcnt", A, B, out, F, K, 6));
        vs.push_back(make<0, 1, 21, 8, 1>("full nt block 8 temps: xor; s_nop 0; bcnt", A, B, out, F, K, 6));
        vs.push_back(make<0, 1, 23, 8, 1>("full nt block 8 temps: xor; bcnt; s_nop 0", A, B, out, F, K, 6));
        vs.push_back(make<0, 1, 29, 8, 1>("full nt block 1 temp: xor; bcnt", A, B, out, F, K, 6));
        vs.push_back(make<0, 1, 27, 8, 1>("full nt block 1 temp: xor; s_nop 0; bcnt", A, B, out, F, K, 6));
        vs.push_back(make<0, 1, 28, 8, 1>("full nt block 1 temp: xor; bcnt; s_nop 0", A, B, out, F, K, 6));
        vs.push_back(make<0, 1, 30, 8, 1>("full nt block 1 temp: xor; s_nop 0; bcnt; s_nop 0", A, B, out, F, K, 6));
        vs.push_back(make<1, 0, 8, 8, 1>("compute-only 8 chains, statement per instr", A, B, out, F, K, 6));
        vs.push_back(make<1, 0, 27, 8, 1>("compute-only block 1 temp: xor; s_nop 0; bcnt", A, B, out, F, K, 6));
        vs.push_back(make<1, 0, 28, 8, 1>("compute-only block 1 temp: xor; bcnt; s_nop 0", A, B, out, F, K, 6));
    } else if (getenv("EXP_COMPUTE")) {
        // order of the 8 distance chains of a lane (the round's main finding), without and with the store
        vs.push_back(make<1, 0, 2, 8, 1>("compute-only 2 chains (il2)", A, B, out, F, K, 5));
        vs.push_back(make<1, 0, 4, 8, 1>("compute-only 4 chains (il4)", A, B, out, F, K, 5));
        vs.push_back(make<1, 0, 8, 8, 1>("compute-only 8 chains pinned", A, B, out, F, K, 5));
        vs.push_back(make<1, 0, 8, 8, 1>("compute-only 8 chains pinned", A, B, out, F, K, 6));
        vs.push_back(make<1, 0, 8, 8, 1>("compute-only 8 chains pinned", A, B, out, F, K, 7));
        vs.push_back(make<1, 0, 9, 8, 1>("compute-only 8 chains, compiler's order", A, B, out, F, K, 6));
        vs.push_back(make<1, 0, 10, 8, 1>("compute-only 8 chains, xor one step ahead", A, B, out, F, K, 6));
        vs.push_back(make<1, 0, 11, 8, 1>("compute-only 8 chains, 2 xor then 2 bcnt", A, B, out, F, K, 6));
        vs.push_back(make<1, 0, 16, 8, 1>("compute-only 16 chains pinned (2x work)", A, B, out, F, K, 5));
        vs.push_back(make<0, 1, 2, 8, 1>("full nt 2 chains (il2)", A, B, out, F, K, 5));
        vs.push_back(make<0, 1, 8, 8, 1>("full nt 8 chains pinned", A, B, out, F, K, 5));
        vs.push_back(make<0, 1, 8, 8, 1>("full nt 8 chains pinned", A, B, out, F, K, 6));
        vs.push_back(make<0, 1, 8, 8, 1>("full nt 8 chains pinned", A, B, out, F, K, 7));
        vs.push_back(make<0, 3, 8, 8, 1>("full nt (sgpr row base) 8 chains pinned", A, B, out, F, K, 6));
    } else {
    vs.push_back(make<0, 0, 1, 8, 1>("full plain serial (r1 kernel)", A, B, out, F, K, 6));
    vs.push_back(make<0, 1, 1, 8, 1>("full nt serial", A, B, out, F, K, 6));
    vs.push_back(make<0, 1, 1, 8, 1>("full nt serial", A, B, out, F, K, 5));
    vs.push_back(make<0, 1, 1, 8, 1>("full nt serial", A, B, out, F, K, 4));
    vs.push_back(make<0, 1, 2, 8, 1>("full nt il2", A, B, out, F, K, 6));
    vs.push_back(make<0, 1, 2, 8, 1>("full nt il2", A, B, out, F, K, 5));
    vs.push_back(make<0, 1, 4, 8, 1>("full nt il4", A, B, out, F, K, 5));
    vs.push_back(make<0, 1, 1, 16, 1>("full nt serial u16", A, B, out, F, K, 5));
    vs.push_back(make<0, 0, 1, 8, 1>("full plain serial", A, B, out, F, K, 5));
    vs.push_back(make<1, 0, 1, 8, 1>("compute-only serial", A, B, out, F, K, 5));
    vs.push_back(make<1, 0, 2, 8, 1>("compute-only il2", A, B, out, F, K, 5));
    vs.push_back(make<2, 1, 1, 8, 1>("store-only nt", A, B, out, F, K, 5));
    vs.push_back(make<2, 0, 1, 8, 1>("store-only plain", A, B, out, F, K, 5));
    }
    for (int i = 0; i < 100; ++i) vs[0].once();                // pre-roll: ~80 ms of load, the clocks settle
    for (auto &v : vs) v.once();                               // warm-up
    for (int round = 0; round < 15; ++round)                    // interleaved rounds: drift of the clock hits all variants alike
        for (auto &v : vs) v.t.push_back(v.once());
    for (auto &v : vs) {
        std::sort(v.t.begin(), v.t.end());
        const float med = v.t[v.t.size() / 2], mn = v.t.front();
        printf("%s  median %7.1f us  min %7.1f us  %5.1f%% of 8 TB/s (median)  %.2f T pairs/s\n", v.name.c_str(), med * 1e3, mn * 1e3,
               2.0 * F * K / med / 1e6 / 80.0, (double)F * K / med / 1e9);
    }
    return 0;
}
