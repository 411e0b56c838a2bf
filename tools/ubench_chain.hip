// Developer microbenchmark: issue rate of the 256-bit distance as the kernels emit it:
// per column 8 x (v_xor_b32 v, s, v ; v_bcnt_u32_b32 acc, v, acc), IL columns interleaved.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32;
#define ITERS 512
template <int IL>
__global__ __launch_bounds__(256) void k(u32 *out, const u32 *rows)
{
    u32 q[8][8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int w = 0; w < 8; ++w) q[c][w] = threadIdx.x * 2654435761u + c * 977 + w;
    u32 acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < ITERS; ++i) {
        u32 s[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) s[w] = __builtin_amdgcn_readfirstlane(rows[(i * 8 + w) & 1023]);   // wave-uniform -> SGPR
#pragma unroll
        for (int c0 = 0; c0 < 8; c0 += IL) {
#pragma unroll
            for (int w = 0; w < 8; ++w) {
#pragma unroll
                for (int c = c0; c < c0 + IL; ++c) {
                    u32 x;
                    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "s"(s[w]), "v"(q[c][w]));
                    asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[c]) : "v"(x));
                }
            }
        }
    }
    u32 r = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) r ^= acc[c];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int IL> void run(int per_cu, u32 *out, u32 *rows)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    int grid = 256 * per_cu;
    hipLaunchKernelGGL(k<IL>, dim3(grid), dim3(256), 0, 0, out, rows);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<IL>, dim3(grid), dim3(256), 0, 0, out, rows);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    double winstr = (double)grid * 4 * ITERS * 128;      // wave-instructions (xor + bcnt)
    printf("interleave=%d waves/SIMD=%d : %.2f cyc per VALU instr per SIMD @2.4GHz  (%.2f T pairs/s)\n", IL, per_cu,
           2.4e9 * (ms * 1e-3) * 1024 / winstr, (double)grid * 256 * ITERS * 8 / ms / 1e9);
}
int main()
{
    u32 *out, *rows; hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&rows, 4096); hipMemset(rows, 0x5a, 4096);
    for (int w : {2, 4, 5, 8}) { run<1>(w, out, rows); run<2>(w, out, rows); run<4>(w, out, rows); run<8>(w, out, rows); }
    return 0;
}
