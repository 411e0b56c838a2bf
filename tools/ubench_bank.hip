// Developer microbenchmark: does the VGPR bank of the operands matter for the distance chain?
// One asm block with explicit registers.  v_bcnt_u32_b32 acc, x, acc reads two VGPRs: variant A puts x and acc in the
// same bank (register numbers equal mod 4), variant B in different banks; the xor reads one SGPR and one VGPR.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define ITERS 2048
// 8 columns: q in v[32..39] (or spread), acc in v[48..55], x temps in v[40..47]
template <int VAR>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed)
{
    const uint32_t s = __builtin_amdgcn_readfirstlane(seed * 2654435761u);
    uint32_t r = threadIdx.x;
    for (int i = 0; i < ITERS; ++i) {
        if (VAR == 0) {          // x = v40+c, acc = v48+c: same bank (40 % 4 == 48 % 4)
            asm volatile(
                "v_xor_b32 v40, %1, v32\n v_bcnt_u32_b32 v48, v40, v48\n v_xor_b32 v41, %1, v33\n v_bcnt_u32_b32 v49, v41, v49\n"
                "v_xor_b32 v42, %1, v34\n v_bcnt_u32_b32 v50, v42, v50\n v_xor_b32 v43, %1, v35\n v_bcnt_u32_b32 v51, v43, v51\n"
                "v_xor_b32 v44, %1, v36\n v_bcnt_u32_b32 v52, v44, v52\n v_xor_b32 v45, %1, v37\n v_bcnt_u32_b32 v53, v45, v53\n"
                "v_xor_b32 v46, %1, v38\n v_bcnt_u32_b32 v54, v46, v54\n v_xor_b32 v47, %1, v39\n v_bcnt_u32_b32 v55, v47, v55\n"
                : "+v"(r) : "s"(s) : "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55");
        } else if (VAR == 1) {   // x = v41+c (bank +1), acc = v48+c
            asm volatile(
                "v_xor_b32 v41, %1, v32\n v_bcnt_u32_b32 v48, v41, v48\n v_xor_b32 v42, %1, v33\n v_bcnt_u32_b32 v49, v42, v49\n"
                "v_xor_b32 v43, %1, v34\n v_bcnt_u32_b32 v50, v43, v50\n v_xor_b32 v44, %1, v35\n v_bcnt_u32_b32 v51, v44, v51\n"
                "v_xor_b32 v45, %1, v36\n v_bcnt_u32_b32 v52, v45, v52\n v_xor_b32 v46, %1, v37\n v_bcnt_u32_b32 v53, v46, v53\n"
                "v_xor_b32 v47, %1, v38\n v_bcnt_u32_b32 v54, v47, v54\n v_xor_b32 v40, %1, v39\n v_bcnt_u32_b32 v55, v40, v55\n"
                : "+v"(r) : "s"(s) : "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55");
        } else if (VAR == 2) {   // ONE temp register reused by every xor (what the compiler emits in the matrix kernel), same bank as acc
            asm volatile(
                "v_xor_b32 v40, %1, v32\n v_bcnt_u32_b32 v48, v40, v48\n v_xor_b32 v40, %1, v33\n v_bcnt_u32_b32 v48, v40, v48\n"
                "v_xor_b32 v40, %1, v34\n v_bcnt_u32_b32 v48, v40, v48\n v_xor_b32 v40, %1, v35\n v_bcnt_u32_b32 v48, v40, v48\n"
                "v_xor_b32 v40, %1, v36\n v_bcnt_u32_b32 v48, v40, v48\n v_xor_b32 v40, %1, v37\n v_bcnt_u32_b32 v48, v40, v48\n"
                "v_xor_b32 v40, %1, v38\n v_bcnt_u32_b32 v48, v40, v48\n v_xor_b32 v40, %1, v39\n v_bcnt_u32_b32 v48, v40, v48\n"
                : "+v"(r) : "s"(s) : "v32","v33","v34","v35","v36","v37","v38","v39","v40","v48");
        } else if (VAR == 3) {   // same serial chain, temp in another bank than acc
            asm volatile(
                "v_xor_b32 v41, %1, v32\n v_bcnt_u32_b32 v48, v41, v48\n v_xor_b32 v41, %1, v33\n v_bcnt_u32_b32 v48, v41, v48\n"
                "v_xor_b32 v41, %1, v34\n v_bcnt_u32_b32 v48, v41, v48\n v_xor_b32 v41, %1, v35\n v_bcnt_u32_b32 v48, v41, v48\n"
                "v_xor_b32 v41, %1, v36\n v_bcnt_u32_b32 v48, v41, v48\n v_xor_b32 v41, %1, v37\n v_bcnt_u32_b32 v48, v41, v48\n"
                "v_xor_b32 v41, %1, v38\n v_bcnt_u32_b32 v48, v41, v48\n v_xor_b32 v41, %1, v39\n v_bcnt_u32_b32 v48, v41, v48\n"
                : "+v"(r) : "s"(s) : "v32","v33","v34","v35","v36","v37","v38","v39","v41","v48");
        } else if (VAR == 4) {   // serial chain, 8 DIFFERENT row SGPRs like a real row (s[%1..]) -- uses s plus its neighbours via s_mov copies
            asm volatile(
                "s_mov_b32 s40, %1\n s_not_b32 s41, %1\n s_brev_b32 s42, %1\n s_mov_b32 s43, %1\n s_not_b32 s44, %1\n s_brev_b32 s45, %1\n s_mov_b32 s46, %1\n s_not_b32 s47, %1\n"
                "v_xor_b32 v41, s40, v32\n v_bcnt_u32_b32 v48, v41, v48\n v_xor_b32 v41, s41, v33\n v_bcnt_u32_b32 v48, v41, v48\n"
                "v_xor_b32 v41, s42, v34\n v_bcnt_u32_b32 v48, v41, v48\n v_xor_b32 v41, s43, v35\n v_bcnt_u32_b32 v48, v41, v48\n"
                "v_xor_b32 v41, s44, v36\n v_bcnt_u32_b32 v48, v41, v48\n v_xor_b32 v41, s45, v37\n v_bcnt_u32_b32 v48, v41, v48\n"
                "v_xor_b32 v41, s46, v38\n v_bcnt_u32_b32 v48, v41, v48\n v_xor_b32 v41, s47, v39\n v_bcnt_u32_b32 v48, v41, v48\n"
                : "+v"(r) : "s"(s) : "v32","v33","v34","v35","v36","v37","v38","v39","v41","v48","s40","s41","s42","s43","s44","s45","s46","s47");
        }
    }
    uint32_t acc;
    asm volatile("v_mov_b32 %0, v48" : "=v"(acc));
    out[blockIdx.x * 256 + threadIdx.x] = r ^ acc;
}
template <int VAR> void run(const char *name, int per_cu, uint32_t *out)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int grid = 256 * per_cu;
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<VAR>, dim3(grid), dim3(256), 0, 0, out, 1u);
    (void)hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(a);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<VAR>, dim3(grid), dim3(256), 0, 0, out, 1u);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= 5;
        if (ms < best) best = ms;
    }
    const double winstr = (double)grid * 4 * ITERS * 16;
    printf("%-56s waves/SIMD=%d  %.2f cyc/wave-instr/SIMD @2.4GHz  %.2f T pairs/s\n", name, per_cu, 2.4e9 * (best * 1e-3) * 1024 / winstr,
           (double)grid * 256 * ITERS / (best * 1e-3) / 1e12);
}
int main()
{
    uint32_t *out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int rnd = 0; rnd < 2; ++rnd)
        for (int w : {6, 4}) {
            run<0>("8 chains, x and acc same bank", w, out);
            run<1>("8 chains, x and acc different banks", w, out);
            run<2>("serial chain, one temp, same bank as acc", w, out);
            run<3>("serial chain, one temp, other bank", w, out);
            run<4>("serial chain, other bank, 8 different row SGPRs", w, out);
        }
    return 0;
}
