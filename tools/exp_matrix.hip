// Developer experiment: persistent u16 distance-matrix kernel variants (columns per lane, rows buffered before storing).
// hipcc --offload-arch=gfx950 -O3 -o tools/exp_matrix tools/exp_matrix.hip
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
// NJ columns per lane (4 or 8), G rows computed before their stores are issued (1, 2, 4, 8), UNIT rows per unit
template <int NJ, int G, int UNIT, int MODE, int BS>
__global__ __launch_bounds__(BS) void k(const uint4 *__restrict__ A, int64_t na, const uint4 *__restrict__ B, int64_t nb,
                                         uint16_t *__restrict__ out, int n_col_tiles, int n_units)
{
    const int ct = blockIdx.x % n_col_tiles;
    const int k0 = blockIdx.x / n_col_tiles, kstep = gridDim.x / n_col_tiles;
    const int64_t j0 = ((int64_t)ct * BS + threadIdx.x) * NJ;
    u32 b[NJ][8];
#pragma unroll
    for (int c = 0; c < NJ; ++c) {
        const int64_t j = j0 + c < nb ? j0 + c : nb - 1;
        const uint4 lo = B[2 * j], hi = B[2 * j + 1];
        b[c][0] = lo.x; b[c][1] = lo.y; b[c][2] = lo.z; b[c][3] = lo.w; b[c][4] = hi.x; b[c][5] = hi.y; b[c][6] = hi.z; b[c][7] = hi.w;
    }
    if (j0 >= nb) return;
    u32 keep = 0;
    for (int unit = k0; unit < n_units; unit += kstep) {
        const int64_t i0 = (int64_t)unit * UNIT;
#pragma unroll
        for (int g = 0; g < UNIT; g += G) {
            uint4 ra[G], rb[G];
#pragma unroll
            for (int e = 0; e < G; ++e) { const int64_t ii = i0 + g + e < na ? i0 + g + e : na - 1; ra[e] = A[2 * ii]; rb[e] = A[2 * ii + 1]; }
            u32 w[G][NJ / 2];
#pragma unroll
            for (int e = 0; e < G; ++e)
#pragma unroll
                for (int p = 0; p < NJ / 2; ++p) {
                    if (MODE == 2) w[e][p] = ra[e].x ^ b[2 * p][0];
                    else { const u32 odd = ham8(b[2 * p + 1], ra[e], rb[e], 0); w[e][p] = ham8(b[2 * p], ra[e], rb[e], odd << 16); }
                }
#pragma unroll
            for (int e = 0; e < G; ++e) {
                if (MODE == 1) { for (int p = 0; p < NJ / 2; ++p) keep ^= w[e][p]; }
                else if (i0 + g + e < na) {
                    uint16_t *o = out + (i0 + g + e) * nb + j0;
                    if (NJ == 8) *reinterpret_cast<uint4 *>(o) = make_uint4(w[e][0], w[e][1], w[e][2], w[e][3]);
                    else *reinterpret_cast<uint2 *>(o) = make_uint2(w[e][0], w[e][1]);
                }
            }
        }
    }
    if (MODE == 1) out[j0] = (uint16_t)keep;
}
template <int NJ, int G, int UNIT, int MODE, int BS = 256> void run(const char *name, const uint4 *A, const uint4 *B, uint16_t *out, int64_t F, int64_t K, int per_cu)
{
    const int n_col_tiles = (int)((K + BS * NJ - 1) / (BS * NJ));
    const int n_units = (int)((F + UNIT - 1) / UNIT);
    int api = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, k<NJ, G, UNIT, MODE, BS>, BS, 0);
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void *)k<NJ, G, UNIT, MODE, BS>);
    if (per_cu <= 0) per_cu = api;
    int grid = 256 * per_cu / n_col_tiles * n_col_tiles; if (grid < n_col_tiles) grid = n_col_tiles;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<NJ, G, UNIT, MODE, BS>), dim3(grid), dim3(BS), 0, 0, A, F, B, K, out, n_col_tiles, n_units);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k<NJ, G, UNIT, MODE, BS>), dim3(grid), dim3(BS), 0, 0, A, F, B, K, out, n_col_tiles, n_units);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
    printf("%-30s regs=%3d api=%d per_cu=%d grid=%5d %8.1f us  %7.1f GB/s  %5.1f%%\n", name, fa.numRegs, api, per_cu, grid, ms * 1e3,
           2.0 * F * K / ms / 1e6, 2.0 * F * K / ms / 1e6 / 80.0);
}
int main()
{
    const int64_t F = 20000, K = 20000;
    std::vector<uint32_t> h((size_t)F * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u) ^ (uint32_t)(i >> 3);
    uint4 *A, *B; uint16_t *out;
    hipMalloc(&A, F * 32); hipMalloc(&B, K * 32); hipMalloc(&out, F * K * 2 + 4096);
    hipMemcpy(A, h.data(), F * 32, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), K * 32, hipMemcpyHostToDevice);
    run<8, 4, 16, 0, 256>("nj8 g4 u16 bs256", A, B, out, F, K, 0);
    run<8, 4, 16, 0, 512>("nj8 g4 u16 bs512", A, B, out, F, K, 0);
    run<8, 4, 16, 0, 1024>("nj8 g4 u16 bs1024", A, B, out, F, K, 0);
    run<8, 2, 16, 0, 512>("nj8 g2 u16 bs512", A, B, out, F, K, 0);
    run<8, 2, 16, 0, 1024>("nj8 g2 u16 bs1024", A, B, out, F, K, 0);
    run<8, 4, 16, 2, 256>("store-only bs256", A, B, out, F, K, 0);
    run<8, 4, 16, 2, 512>("store-only bs512", A, B, out, F, K, 0);
    run<8, 4, 16, 2, 1024>("store-only bs1024", A, B, out, F, K, 0);
    run<8, 4, 16, 2, 64>("store-only bs64", A, B, out, F, K, 0);
    run<8, 4, 16, 0, 64>("nj8 g4 u16 bs64", A, B, out, F, K, 0);
    run<8, 4, 16, 0, 128>("nj8 g4 u16 bs128", A, B, out, F, K, 0);
    return 0;
}
