// Developer experiment: producer/consumer u16 distance-matrix kernel.  4 compute waves write packed rows to a
// double-buffered LDS tile; 1 store wave streams finished tiles to HBM, so vmcnt waits only ever stall that wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef uint32_t u32;
__device__ __forceinline__ u32 bcnt_acc(u32 x, u32 acc) { u32 r; asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc)); return r; }
__device__ __forceinline__ u32 ham8(const u32 q[8], const uint4 a, const uint4 b, u32 init)
{
    u32 acc = init;
    acc = bcnt_acc(q[0] ^ a.x, acc); acc = bcnt_acc(q[1] ^ a.y, acc); acc = bcnt_acc(q[2] ^ a.z, acc); acc = bcnt_acc(q[3] ^ a.w, acc);
    acc = bcnt_acc(q[4] ^ b.x, acc); acc = bcnt_acc(q[5] ^ b.y, acc); acc = bcnt_acc(q[6] ^ b.z, acc); acc = bcnt_acc(q[7] ^ b.w, acc);
    return acc;
}
// block = (4 + NS) waves: waves 0..3 compute (cover 2048 columns), waves 4.. store.  UR rows per unit.
template <int UR, int NS>
__global__ __launch_bounds__(64 * (4 + NS)) void k(const uint4 *__restrict__ A, int64_t na, const uint4 *__restrict__ B, int64_t nb,
                                                    uint16_t *__restrict__ out, int n_col_tiles, int n_units)
{
    __shared__ uint4 tile[2][UR][256];            // [buffer][row][compute lane] 16 bytes each: UR * 4 KB per buffer
    const int ct = blockIdx.x % n_col_tiles;
    const int k0 = blockIdx.x / n_col_tiles, kstep = gridDim.x / n_col_tiles;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    int nu = 0;
    for (int u = k0; u < n_units; u += kstep) ++nu;
    if (wave < 4) {
        const int64_t j0 = ((int64_t)ct * 256 + threadIdx.x) * 8;
        u32 b[8][8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int64_t j = j0 + c < nb ? j0 + c : nb - 1;
            const uint4 lo = B[2 * j], hi = B[2 * j + 1];
            b[c][0] = lo.x; b[c][1] = lo.y; b[c][2] = lo.z; b[c][3] = lo.w; b[c][4] = hi.x; b[c][5] = hi.y; b[c][6] = hi.z; b[c][7] = hi.w;
        }
        int it = 0;
        for (int unit = k0; unit < n_units; unit += kstep, ++it) {
            const int64_t i0 = (int64_t)unit * UR;
            const int buf = it & 1;
#pragma unroll
            for (int e = 0; e < UR; ++e) {
                const int64_t ii = i0 + e < na ? i0 + e : na - 1;
                const uint4 ra = A[2 * ii], rb = A[2 * ii + 1];
                u32 w[4];
#pragma unroll
                for (int p = 0; p < 4; ++p) { const u32 odd = ham8(b[2 * p + 1], ra, rb, 0); w[p] = ham8(b[2 * p], ra, rb, odd << 16); }
                tile[buf][e][threadIdx.x] = make_uint4(w[0], w[1], w[2], w[3]);
            }
            __syncthreads();     // tile[buf] complete -> store waves may read it; also: store waves finished reading tile[buf^1]'s predecessor
        }
        if (nu & 0) __syncthreads();
    } else {
        const int sw = wave - 4;
        int it = 0;
        for (int unit = k0; unit < n_units; unit += kstep, ++it) {
            __syncthreads();     // wait for tile[it & 1]
            const int64_t i0 = (int64_t)unit * UR;
            const int buf = it & 1;
            // UR rows x 4 KB; this store wave handles rows sw, sw + NS, ...; 4 x 1 KB pieces per row
            for (int e = sw; e < UR; e += NS) {
                if (i0 + e >= na) break;
#pragma unroll
                for (int piece = 0; piece < 4; ++piece) {
                    const uint4 v = tile[buf][e][piece * 64 + lane];
                    const int64_t j0 = ((int64_t)ct * 256 + piece * 64 + lane) * 8;
                    if (j0 + 8 <= nb) *reinterpret_cast<uint4 *>(out + (i0 + e) * nb + j0) = v;
                }
            }
        }
    }
}
template <int UR, int NS> void run(const char *name, const uint4 *A, const uint4 *B, uint16_t *out, int64_t F, int64_t K, int per_cu)
{
    const int n_col_tiles = (int)((K + 2047) / 2048);
    const int n_units = (int)((F + UR - 1) / UR);
    int api = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, k<UR, NS>, 64 * (4 + NS), 0);
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void *)k<UR, NS>);
    if (per_cu <= 0) per_cu = api;
    int grid = 256 * per_cu / n_col_tiles * n_col_tiles;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<UR, NS>), dim3(grid), dim3(64 * (4 + NS)), 0, 0, A, F, B, K, out, n_col_tiles, n_units);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k<UR, NS>), dim3(grid), dim3(64 * (4 + NS)), 0, 0, A, F, B, K, out, n_col_tiles, n_units);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
    printf("%-26s regs=%3d lds=%6zu api=%d per_cu=%d grid=%5d %8.1f us  %7.1f GB/s  %5.1f%%\n", name, fa.numRegs, (size_t)fa.sharedSizeBytes, api, per_cu, grid,
           ms * 1e3, 2.0 * F * K / ms / 1e6, 2.0 * F * K / ms / 1e6 / 80.0);
}
int main()
{
    const int64_t F = 20000, K = 20000;
    std::vector<uint32_t> h((size_t)F * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u) ^ (uint32_t)(i >> 3);
    uint4 *A, *B; uint16_t *out;
    hipMalloc(&A, F * 32); hipMalloc(&B, K * 32); hipMalloc(&out, F * K * 2 + 4096);
    hipMemcpy(A, h.data(), F * 32, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), K * 32, hipMemcpyHostToDevice);
    hipMemset(out, 0xff, F * K * 2);
    run<4, 1>("pc ur4 ns1", A, B, out, F, K, 0);
    run<2, 1>("pc ur2 ns1", A, B, out, F, K, 0);
    run<4, 2>("pc ur4 ns2", A, B, out, F, K, 0);
    run<8, 1>("pc ur8 ns1", A, B, out, F, K, 0);
    run<8, 2>("pc ur8 ns2", A, B, out, F, K, 0);
    run<4, 4>("pc ur4 ns4", A, B, out, F, K, 0);
    // correctness spot check of the last run against a host popcount
    std::vector<uint16_t> row(K);
    hipMemcpy(row.data(), out + 12345 * K, K * 2, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int j = 0; j < 2000 * 8; j += 7) { int d = 0; for (int w = 0; w < 8; ++w) d += __builtin_popcount(h[12345 * 8 + w] ^ h[(size_t)j * 8 + w]); bad += d != row[j]; }
    printf("spot check mismatches: %d\n", bad);
    return 0;
}
