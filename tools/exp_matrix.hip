// Developer experiment: what binds the u16 distance-matrix kernel -- VALU or the store stream?
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
struct MatRows { uint4 a[4], b[4]; };
__device__ __forceinline__ MatRows mat_load(const uint4 *__restrict__ A, int64_t i, int64_t last)
{
    MatRows r;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const int64_t ii = i + e < last ? i + e : last; r.a[e] = A[2 * ii]; r.b[e] = A[2 * ii + 1]; }
    return r;
}
// MODE 0: full; 1: compute only (one store per block of rows, keeps values live); 2: store only (no distance work)
template <int MODE, int ROWS, int NT>
__global__ __launch_bounds__(256) void k(const uint4 *__restrict__ A, int64_t na, const uint4 *__restrict__ B, int64_t nb, uint16_t *__restrict__ out)
{
    const int64_t j0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    const int64_t i0 = (int64_t)blockIdx.y * ROWS;
    const int64_t i1 = i0 + ROWS < na ? i0 + ROWS : na;
    u32 b[8][8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int64_t j = j0 + c < nb ? j0 + c : nb - 1;
        const uint4 lo = B[2 * j], hi = B[2 * j + 1];
        b[c][0] = lo.x; b[c][1] = lo.y; b[c][2] = lo.z; b[c][3] = lo.w; b[c][4] = hi.x; b[c][5] = hi.y; b[c][6] = hi.z; b[c][7] = hi.w;
    }
    if (j0 >= nb) return;
    MatRows nxt = mat_load(A, i0, na - 1);
    u32 keep = 0;
    for (int64_t i = i0; i < i1; i += 4) {
        const MatRows cur = nxt;
        nxt = mat_load(A, i + 4 < na ? i + 4 : na - 1, na - 1);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            u32 w[4];
            if (MODE == 2) { w[0] = cur.a[e].x ^ b[0][0]; w[1] = cur.a[e].y ^ b[1][0]; w[2] = cur.a[e].z; w[3] = cur.a[e].w; }
            else {
#pragma unroll
                for (int p = 0; p < 4; ++p) { const u32 odd = ham8(b[2 * p + 1], cur.a[e], cur.b[e], 0); w[p] = ham8(b[2 * p], cur.a[e], cur.b[e], odd << 16); }
            }
            if (MODE == 1) { keep ^= w[0] ^ w[1] ^ w[2] ^ w[3]; }
            else if (i + e < i1) {
                uint16_t *o = out + (i + e) * nb + j0;
                if (NT) { typedef u32 v4 __attribute__((ext_vector_type(4))); v4 vv = {w[0], w[1], w[2], w[3]}; __builtin_nontemporal_store(vv, reinterpret_cast<v4 *>(o)); }
                else *reinterpret_cast<uint4 *>(o) = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
    }
    if (MODE == 1) out[(i0 * nb + j0)] = (uint16_t)keep;
}
template <int MODE, int ROWS, int NT> void run(const char *name, const uint4 *A, const uint4 *B, uint16_t *out, int64_t F, int64_t K)
{
    dim3 grid((unsigned)((K + 2047) / 2048), (unsigned)((F + ROWS - 1) / ROWS));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<MODE, ROWS, NT>), grid, dim3(256), 0, 0, A, F, B, K, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k<MODE, ROWS, NT>), grid, dim3(256), 0, 0, A, F, B, K, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
    printf("%-34s %8.1f us  %7.1f GB/s-equivalent  %6.2f T pairs/s\n", name, ms * 1e3, 2.0 * F * K / ms / 1e6, (double)F * K / ms / 1e9);
}
int main()
{
    const int64_t F = 20000, K = 20000;
    std::vector<uint32_t> h((size_t)F * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u) ^ (uint32_t)(i >> 3);
    uint4 *A, *B; uint16_t *out;
    hipMalloc(&A, F * 32); hipMalloc(&B, K * 32); hipMalloc(&out, F * K * 2 + 4096);
    hipMemcpy(A, h.data(), F * 32, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), K * 32, hipMemcpyHostToDevice);
    run<0, 128, 0>("full rows=128", A, B, out, F, K);
    run<0, 128, 1>("full rows=128 nontemporal", A, B, out, F, K);
    run<0, 64, 0>("full rows=64", A, B, out, F, K);
    run<0, 256, 0>("full rows=256", A, B, out, F, K);
    run<0, 512, 1>("full rows=512 nontemporal", A, B, out, F, K);
    run<1, 128, 0>("compute only rows=128", A, B, out, F, K);
    run<2, 128, 0>("store only rows=128", A, B, out, F, K);
    run<2, 128, 1>("store only rows=128 nontemporal", A, B, out, F, K);
    return 0;
}
