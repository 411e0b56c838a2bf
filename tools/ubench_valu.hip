// Developer microbenchmark: issue rate of the VALU instructions the Hamming kernels use.
// hipcc --offload-arch=gfx950 -O3 -o tools/ubench_valu tools/ubench_valu.hip && ./tools/ubench_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITERS 4096
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed)
{
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint32_t s = seed * 2654435761u;
    for (int i = 0; i < ITERS; ++i) {
#define OP8(INS) \
        asm volatile(INS " %0, %0, %1" : "+v"(a0) : "v"(a1)); asm volatile(INS " %0, %0, %1" : "+v"(a1) : "v"(a2)); \
        asm volatile(INS " %0, %0, %1" : "+v"(a2) : "v"(a3)); asm volatile(INS " %0, %0, %1" : "+v"(a3) : "v"(a4)); \
        asm volatile(INS " %0, %0, %1" : "+v"(a4) : "v"(a5)); asm volatile(INS " %0, %0, %1" : "+v"(a5) : "v"(a6)); \
        asm volatile(INS " %0, %0, %1" : "+v"(a6) : "v"(a7)); asm volatile(INS " %0, %0, %1" : "+v"(a7) : "v"(a0));
        if (MODE == 0) { OP8("v_xor_b32") OP8("v_xor_b32") }
        if (MODE == 1) { OP8("v_bcnt_u32_b32") OP8("v_bcnt_u32_b32") }
        if (MODE == 2) { OP8("v_xor_b32") OP8("v_bcnt_u32_b32") }
        if (MODE == 3) { OP8("v_add_u32") OP8("v_add_u32") }
        if (MODE == 4) { OP8("v_min_u32") OP8("v_min_u32") }
        if (MODE == 6) { OP8("v_max_u32") OP8("v_max_u32") }
        if (MODE == 7) { OP8("v_and_b32") OP8("v_and_b32") }
#define OP8C(INS, C) \
        asm volatile(INS " %0, %0, " C ", %1" : "+v"(a0) : "v"(a1)); asm volatile(INS " %0, %0, " C ", %1" : "+v"(a1) : "v"(a2)); \
        asm volatile(INS " %0, %0, " C ", %1" : "+v"(a2) : "v"(a3)); asm volatile(INS " %0, %0, " C ", %1" : "+v"(a3) : "v"(a4)); \
        asm volatile(INS " %0, %0, " C ", %1" : "+v"(a4) : "v"(a5)); asm volatile(INS " %0, %0, " C ", %1" : "+v"(a5) : "v"(a6)); \
        asm volatile(INS " %0, %0, " C ", %1" : "+v"(a6) : "v"(a7)); asm volatile(INS " %0, %0, " C ", %1" : "+v"(a7) : "v"(a0));
#define OP83(INS) \
        asm volatile(INS " %0, %0, %1, %2" : "+v"(a0) : "v"(a1), "v"(a2)); asm volatile(INS " %0, %0, %1, %2" : "+v"(a1) : "v"(a2), "v"(a3)); \
        asm volatile(INS " %0, %0, %1, %2" : "+v"(a2) : "v"(a3), "v"(a4)); asm volatile(INS " %0, %0, %1, %2" : "+v"(a3) : "v"(a4), "v"(a5)); \
        asm volatile(INS " %0, %0, %1, %2" : "+v"(a4) : "v"(a5), "v"(a6)); asm volatile(INS " %0, %0, %1, %2" : "+v"(a5) : "v"(a6), "v"(a7)); \
        asm volatile(INS " %0, %0, %1, %2" : "+v"(a6) : "v"(a7), "v"(a0)); asm volatile(INS " %0, %0, %1, %2" : "+v"(a7) : "v"(a0), "v"(a1));
        if (MODE == 8) { OP8C("v_lshl_or_b32", "16") OP8C("v_lshl_or_b32", "16") }
        if (MODE == 9) { OP83("v_min3_u32") OP83("v_min3_u32") }
        if (MODE == 10) { OP83("v_add3_u32") OP83("v_add3_u32") }
        if (MODE == 11) { OP83("v_xad_u32") OP83("v_xad_u32") }
        if (MODE == 12) { OP8("v_min_i32") OP8("v_min_i32") }
        if (MODE == 13) { OP8("v_min_u16") OP8("v_min_u16") }
        if (MODE == 14) { OP8("v_pk_min_u16") OP8("v_pk_min_u16") }
        if (MODE == 15) { OP8("v_min_f32") OP8("v_min_f32") }
        if (MODE == 16) { OP83("v_bfi_b32") OP83("v_bfi_b32") }
        if (MODE == 17) { OP83("v_perm_b32") OP83("v_perm_b32") }
        if (MODE == 18) { OP83("v_sad_u8") OP83("v_sad_u8") }
        if (MODE == 19) { OP8("v_or_b32") OP8("v_or_b32") }
        if (MODE == 20) { OP8("v_lshlrev_b32") OP8("v_lshlrev_b32") }
        if (MODE == 21) { OP8("v_sub_u32") OP8("v_sub_u32") }
        if (MODE == 22) { OP8("v_cndmask_b32") OP8("v_cndmask_b32") }
        if (MODE == 23) { OP8("v_add_u16") OP8("v_add_u16") }
        if (MODE == 24) { OP8("v_max_u16") OP8("v_max_u16") }
        if (MODE == 25) { OP8("v_lshlrev_b16") OP8("v_lshlrev_b16") }
        if (MODE == 26) { OP8("v_mul_u32_u24") OP8("v_mul_u32_u24") }
        if (MODE == 27) { OP83("v_mad_u32_u24") OP83("v_mad_u32_u24") }
        if (MODE == 28) { OP83("v_and_or_b32") OP83("v_and_or_b32") }
        if (MODE == 29) { OP83("v_lshl_add_u32") OP83("v_lshl_add_u32") }
        if (MODE == 30) { OP83("v_mad_u16") OP83("v_mad_u16") }
        if (MODE == 31) {
            for (int r = 0; r < 2; ++r) {
            asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a0) : "v"(a1) : "vcc");
            asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a2) : "v"(a3) : "vcc");
            asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a4) : "v"(a5) : "vcc");
            asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a6) : "v"(a7) : "vcc");
            }
        }
        if (MODE == 32) { OP8("v_xor_b32") OP8("v_min_u16") }
        if (MODE == 5) {  // xor with SGPR operand + bcnt
            asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a0) : "s"(s)); asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a1) : "v"(a0));
            asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a2) : "s"(s)); asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a3) : "v"(a2));
            asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a4) : "s"(s)); asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a5) : "v"(a4));
            asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a6) : "s"(s)); asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a7) : "v"(a6));
            asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a0) : "s"(s)); asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a1) : "v"(a0));
            asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a2) : "s"(s)); asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a3) : "v"(a2));
            asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a4) : "s"(s)); asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a5) : "v"(a4));
            asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a6) : "s"(s)); asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a7) : "v"(a6));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
template <int MODE> void run(const char *name, int blocks_per_cu)
{
    uint32_t *d; hipMalloc(&d, 256 * 64 * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, 1u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, 1u);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    double ops = (double)grid * 256 * ITERS * 16;
    printf("%-28s waves/SIMD=%d  %.2f T lane-op/s  (%.2f cyc/wave-instr/SIMD @2.4GHz)\n", name, blocks_per_cu, ops / ms / 1e9,
           2.4e9 * 1024 / (ops / 64 / (ms * 1e-3)));
    hipFree(d);
}
int main()
{
    for (int w : {8}) {
        run<19>("v_or_b32", w); run<20>("v_lshlrev_b32", w); run<21>("v_sub_u32", w); run<22>("v_cndmask_b32(vcc)", w); run<23>("v_add_u16", w);
        run<24>("v_max_u16", w); run<25>("v_lshlrev_b16", w); run<26>("v_mul_u32_u24", w); run<27>("v_mad_u32_u24", w); run<28>("v_and_or_b32", w);
        run<29>("v_lshl_add_u32", w); run<30>("v_mad_u16", w); run<31>("cmp+cndmask (8 pairs=16 ops)", w); run<32>("xor+min_u16", w);
    }
    for (int w : {999}) { if (w == 999) break;
        run<0>("v_xor_b32", w); run<1>("v_bcnt_u32_b32", w); run<4>("v_min_u32", w); run<6>("v_max_u32", w); run<7>("v_and_b32", w);
        run<8>("v_lshl_or_b32", w); run<9>("v_min3_u32", w); run<10>("v_add3_u32", w); run<11>("v_xad_u32", w); run<12>("v_min_i32", w);
        run<13>("v_min_u16", w); run<14>("v_pk_min_u16", w); run<15>("v_min_f32", w); run<16>("v_bfi_b32", w); run<17>("v_perm_b32", w);
        run<18>("v_sad_u8", w);
    }
    return 0;
}
