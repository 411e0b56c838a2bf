// Developer microbenchmark (round 2): VALU issue rates on gfx950 in REAL shader cycles.
// Every kernel stamps s_memtime (shader clock) and s_memrealtime (100 MHz) around its loop, so the clock the chip
// actually held is known and "cycles per wave-instruction per SIMD" no longer assumes 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_valu2 tools/ubench_valu2.hip && ./tools/ubench_valu2
// Reconciles tools/ubench_valu.hip (v_xor ~2.5 cyc, v_bcnt ~4.2 cyc at an assumed 2.4 GHz) with the
// microarchitecture guide's "v_fma_f32 wave64 = 2 cyc" row: v_fma_f32 / v_fmac_f32 / v_pk_fma_f32 are measured in
// the same harness.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
#define ITERS 2048

#define R8(X) X(a0, a1) X(a1, a2) X(a2, a3) X(a3, a4) X(a4, a5) X(a5, a6) X(a6, a7) X(a7, a0)
#define OP2(INS) { _Pragma("unroll") for (int u = 0; u < 2; ++u) { R8(OP2_##INS) } }

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint64_t *stamps, uint32_t seed)
{
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint32_t b0 = a0 ^ 0x55, b1 = a1 ^ 0x66, b2 = a2 ^ 77, b3 = a3 ^ 88, b4 = a4 ^ 99, b5 = a5 ^ 11, b6 = a6 ^ 22, b7 = a7 ^ 33;
    const uint32_t s = __builtin_amdgcn_readfirstlane(seed * 2654435761u);
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < ITERS; ++i) {
#define V2(INS, d, x) asm volatile(INS " %0, %0, %1" : "+v"(d) : "v"(x));
#define V2S(INS, d, x) asm volatile(INS " %0, %1, %0" : "+v"(d) : "s"(s));
#define V3(INS, d, x) asm volatile(INS " %0, %0, %1, %0" : "+v"(d) : "v"(x));
#define V3C(INS, C, d, x) asm volatile(INS " %0, %0, " C ", %1" : "+v"(d) : "v"(x));
#define REP16(M, INS) M(INS, a0, a1) M(INS, a1, a2) M(INS, a2, a3) M(INS, a3, a4) M(INS, a4, a5) M(INS, a5, a6) M(INS, a6, a7) M(INS, a7, a0) \
                      M(INS, a0, a1) M(INS, a1, a2) M(INS, a2, a3) M(INS, a3, a4) M(INS, a4, a5) M(INS, a5, a6) M(INS, a6, a7) M(INS, a7, a0)
        if (MODE == 0) { REP16(V2, "v_xor_b32") }
        if (MODE == 1) { REP16(V2, "v_bcnt_u32_b32") }
        if (MODE == 2) { REP16(V2S, "v_xor_b32") }
        if (MODE == 3) { REP16(V3, "v_fma_f32") }
        if (MODE == 4) { REP16(V2, "v_fmac_f32") }
        if (MODE == 5) { REP16(V2, "v_add_f32") }
        if (MODE == 6) { REP16(V2, "v_mul_f32") }
        if (MODE == 7) { REP16(V2, "v_min_u16") }
        if (MODE == 8) { REP16(V2, "v_pk_min_u16") }
        if (MODE == 9) { REP16(V2, "v_pk_lshlrev_b16") }
        if (MODE == 10) { REP16(V3, "v_pk_mad_u16") }
        if (MODE == 11) { REP16(V2, "v_pk_add_u16") }
        if (MODE == 12) { REP16(V2, "v_add_u32") }
        if (MODE == 13) { REP16(V2, "v_and_b32") }
        if (MODE == 14) { REP16(V2, "v_lshlrev_b16") }
        if (MODE == 15) {   // v_lshl_or_b32 d, d, 7, x
#define V3L(INS, d, x) asm volatile("v_lshl_or_b32 %0, %0, 7, %1" : "+v"(d) : "v"(x));
            REP16(V3L, "")
        }
        if (MODE == 16) {   // 64-bit packed f32 fma: v_pk_fma_f32 on register pairs
            uint64_t p0 = ((uint64_t)a1 << 32) | a0, p1 = ((uint64_t)a3 << 32) | a2, p2 = ((uint64_t)a5 << 32) | a4, p3 = ((uint64_t)a7 << 32) | a6;
#define PK(d, x) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(d) : "v"(x));
            for (int u = 0; u < 4; ++u) { PK(p0, p1) PK(p1, p2) PK(p2, p3) PK(p3, p0) }
            a0 = (uint32_t)p0; a1 = (uint32_t)(p0 >> 32); a2 = (uint32_t)p1; a3 = (uint32_t)(p1 >> 32);
            a4 = (uint32_t)p2; a5 = (uint32_t)(p2 >> 32); a6 = (uint32_t)p3; a7 = (uint32_t)(p3 >> 32);
        }
        if (MODE == 17) {   // alternate xor (SGPR operand) / accumulating bcnt: the scan's distance chain, 8 columns
#define XB(q, acc) { uint32_t x; asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "s"(s), "v"(q)); asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(x)); }
            XB(a0, b0) XB(a1, b1) XB(a2, b2) XB(a3, b3) XB(a4, b4) XB(a5, b5) XB(a6, b6) XB(a7, b7)
        }
        if (MODE == 18) {   // same work, grouped: 8 xor then 8 bcnt
            uint32_t x0, x1, x2, x3, x4, x5, x6, x7;
#define XO(x, q) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "s"(s), "v"(q));
#define BC(acc, x) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(x));
            XO(x0, a0) XO(x1, a1) XO(x2, a2) XO(x3, a3) XO(x4, a4) XO(x5, a5) XO(x6, a6) XO(x7, a7)
            BC(b0, x0) BC(b1, x1) BC(b2, x2) BC(b3, x3) BC(b4, x4) BC(b5, x5) BC(b6, x6) BC(b7, x7)
        }
        if (MODE == 19) {   // xor with VGPR operand instead of SGPR, alternating
#define XBV(q, acc, w) { uint32_t x; asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "v"(w), "v"(q)); asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(x)); }
            XBV(a0, b0, a7) XBV(a1, b1, a6) XBV(a2, b2, a5) XBV(a3, b3, a4) XBV(a4, b4, a3) XBV(a5, b5, a2) XBV(a6, b6, a1) XBV(a7, b7, a0)
        }
        if (MODE == 20) {   // 2 xor + 1 bcnt (what a packed / halved popcount would look like)
            uint32_t x0, x1, x2, x3, x4, x5, x6, x7;
            XO(x0, a0) XO(x1, a1) XO(x2, a2) XO(x3, a3) XO(x4, a4) XO(x5, a5) XO(x6, a6) XO(x7, a7)
            BC(b0, x0) BC(b1, x2) BC(b2, x4) BC(b3, x6)
            b4 ^= x1; b5 ^= x3; b6 ^= x5; b7 ^= x7;
        }
        if (MODE == 21) { REP16(V2, "v_max_u16") }
        if (MODE == 22) { REP16(V2, "v_sub_u16") }
        if (MODE == 23) {   // v_cmp + v_cndmask pairs
            asm volatile("v_cmp_lt_u16 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a0) : "v"(a1) : "vcc");
            asm volatile("v_cmp_lt_u16 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a2) : "v"(a3) : "vcc");
            asm volatile("v_cmp_lt_u16 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a4) : "v"(a5) : "vcc");
            asm volatile("v_cmp_lt_u16 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a6) : "v"(a7) : "vcc");
            asm volatile("v_cmp_lt_u16 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a1) : "v"(a2) : "vcc");
            asm volatile("v_cmp_lt_u16 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a3) : "v"(a4) : "vcc");
            asm volatile("v_cmp_lt_u16 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a5) : "v"(a6) : "vcc");
            asm volatile("v_cmp_lt_u16 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a7) : "v"(a0) : "vcc");
        }
        if (MODE == 24) {   // SDWA: v_or_b32 writing word 1 of the destination, word 0 preserved
#define SD(d, x) asm volatile("v_or_b32_sdwa %0, %1, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0" : "+v"(d) : "v"(x));
            SD(a0, a1) SD(a1, a2) SD(a2, a3) SD(a3, a4) SD(a4, a5) SD(a5, a6) SD(a6, a7) SD(a7, a0)
            SD(a0, a1) SD(a1, a2) SD(a2, a3) SD(a3, a4) SD(a4, a5) SD(a5, a6) SD(a6, a7) SD(a7, a0)
        }
        if (MODE == 25) {   // SDWA min on the high words
#define SM(d, x) asm volatile("v_min_u16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(d) : "v"(x));
            SM(a0, a1) SM(a1, a2) SM(a2, a3) SM(a3, a4) SM(a4, a5) SM(a5, a6) SM(a6, a7) SM(a7, a0)
            SM(a0, a1) SM(a1, a2) SM(a2, a3) SM(a3, a4) SM(a4, a5) SM(a5, a6) SM(a6, a7) SM(a7, a0)
        }
        if (MODE == 26) {   // DPP move (cross-lane step of the butterflies)
#define DP(d, x) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(x));
            DP(a0, a1) DP(a1, a2) DP(a2, a3) DP(a3, a4) DP(a4, a5) DP(a5, a6) DP(a6, a7) DP(a7, a0)
            DP(a0, a1) DP(a1, a2) DP(a2, a3) DP(a3, a4) DP(a4, a5) DP(a5, a6) DP(a6, a7) DP(a7, a0)
        }
        if (MODE == 27) {   // DPP fused into v_min_u16
#define DM(d, x) asm volatile("v_min_u16_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(x));
            DM(a0, a1) DM(a1, a2) DM(a2, a3) DM(a3, a4) DM(a4, a5) DM(a5, a6) DM(a6, a7) DM(a7, a0)
            DM(a0, a1) DM(a1, a2) DM(a2, a3) DM(a3, a4) DM(a4, a5) DM(a5, a6) DM(a6, a7) DM(a7, a0)
        }
        if (MODE == 28) { REP16(V2, "v_lshlrev_b32") }
        if (MODE == 29) { REP16(V3, "v_and_or_b32") }
        if (MODE == 30) {   // v_mov_b32 (plain)
#define MV(d, x) asm volatile("v_mov_b32 %0, %1" : "+v"(d) : "v"(x));
            MV(a0, a1) MV(a1, a2) MV(a2, a3) MV(a3, a4) MV(a4, a5) MV(a5, a6) MV(a6, a7) MV(a7, a0)
            MV(a0, a1) MV(a1, a2) MV(a2, a3) MV(a3, a4) MV(a4, a5) MV(a5, a6) MV(a6, a7) MV(a7, a0)
        }
        if (MODE == 31) {   // whole pair as the scan emits it: 8 x (xor s,v + bcnt) then shl16 + or + 2 min16, 2 columns interleaved
#define PAIR2(qa, qb, ca, cb, best) { uint32_t ha = 0, hb = 0, x, y; \
            _Pragma("unroll") for (int w = 0; w < 8; ++w) { \
                asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "s"(s), "v"(qa)); asm volatile("v_xor_b32 %0, %1, %2" : "=v"(y) : "s"(s), "v"(qb)); \
                asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(ha) : "v"(x)); asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(hb) : "v"(y)); } \
            asm volatile("v_lshlrev_b16 %0, 7, %0" : "+v"(ha)); asm volatile("v_lshlrev_b16 %0, 7, %0" : "+v"(hb)); \
            asm volatile("v_or_b32 %0, 8, %0" : "+v"(ha)); asm volatile("v_or_b32 %0, 9, %0" : "+v"(hb)); \
            asm volatile("v_min_u16 %0, %0, %1" : "+v"(ca) : "v"(ha)); asm volatile("v_min_u16 %0, %0, %1" : "+v"(cb) : "v"(hb)); \
            asm volatile("v_min_u16 %0, %0, %1" : "+v"(ha) : "v"(hb)); asm volatile("v_min_u16 %0, %0, %1" : "+v"(best) : "v"(ha)); }
            uint32_t best = 0xffff;
            PAIR2(a0, a1, b0, b1, best) PAIR2(a2, a3, b2, b3, best) PAIR2(a4, a5, b4, b5, best) PAIR2(a6, a7, b6, b7, best)
            b0 ^= best >> 15;
        }
        if (MODE == 32) {   // same with the key made by ONE v_lshl_or_b32 (half-rate) instead of shl16 + or
#define PAIR2L(qa, qb, ca, cb, best) { uint32_t ha = 0, hb = 0, x, y; \
            _Pragma("unroll") for (int w = 0; w < 8; ++w) { \
                asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "s"(s), "v"(qa)); asm volatile("v_xor_b32 %0, %1, %2" : "=v"(y) : "s"(s), "v"(qb)); \
                asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(ha) : "v"(x)); asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(hb) : "v"(y)); } \
            asm volatile("v_lshl_or_b32 %0, %0, 7, 8" : "+v"(ha)); asm volatile("v_lshl_or_b32 %0, %0, 7, 9" : "+v"(hb)); \
            asm volatile("v_min_u16 %0, %0, %1" : "+v"(ca) : "v"(ha)); asm volatile("v_min_u16 %0, %0, %1" : "+v"(cb) : "v"(hb)); \
            asm volatile("v_min_u16 %0, %0, %1" : "+v"(ha) : "v"(hb)); asm volatile("v_min_u16 %0, %0, %1" : "+v"(best) : "v"(ha)); }
            uint32_t best = 0xffff;
            PAIR2L(a0, a1, b0, b1, best) PAIR2L(a2, a3, b2, b3, best) PAIR2L(a4, a5, b4, b5, best) PAIR2L(a6, a7, b6, b7, best)
            b0 ^= best >> 15;
        }
        if (MODE == 33) {   // packed: two columns share one 32-bit accumulator (odd << 16 seeds the even chain), pk ops after
#define PAIRP(qa, qb, cab, best) { uint32_t hb = 0, x, y; \
            _Pragma("unroll") for (int w = 0; w < 8; ++w) { asm volatile("v_xor_b32 %0, %1, %2" : "=v"(y) : "s"(s), "v"(qb)); asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(hb) : "v"(y)); } \
            asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(hb)); \
            _Pragma("unroll") for (int w = 0; w < 8; ++w) { asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "s"(s), "v"(qa)); asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(hb) : "v"(x)); } \
            asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(hb) : "v"(kmul), "v"(kadd)); \
            asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(cab) : "v"(hb)); asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(best) : "v"(hb)); }
            uint32_t best = 0xffffffffu, kmul = 0x00800080u, kadd = 0x00090008u;
            asm volatile("" : "+v"(kmul), "+v"(kadd));
            PAIRP(a0, a1, b0, best) PAIRP(a2, a3, b2, best) PAIRP(a4, a5, b4, best) PAIRP(a6, a7, b6, best)
            b1 ^= best >> 15;
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

static uint32_t *d_out; static uint64_t *d_st;

template <int MODE> void run(const char *name, int per_cu, double instr_per_iter, double pairs_per_iter = 0)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int grid = 256 * per_cu;
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d_out, d_st, 1u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d_out, d_st, 1u);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
    std::vector<uint64_t> st(2 * grid);
    hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> clk, cyc;
    for (int i = 0; i < grid; ++i) { clk.push_back((double)st[2 * i] / (double)st[2 * i + 1] * 100e6); cyc.push_back((double)st[2 * i]); }
    std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
    const double f = clk[grid / 2], wave_cycles = cyc[grid / 2];
    const double winstr = (double)ITERS * instr_per_iter;                  // per wave
    // a SIMD hosts per_cu waves that each spend wave_cycles in the loop: cycles per wave-instruction per SIMD
    const double cyc_per_instr = wave_cycles / (winstr * per_cu);
    printf("%-34s waves/SIMD=%d  clock %.2f GHz  %.2f real cyc/wave-instr/SIMD  (%.2f at assumed 2.4 GHz from events)", name, per_cu,
           f / 1e9, cyc_per_instr, 2.4e9 * (ms * 1e-3) * 1024 / ((double)grid * 4 * winstr));
    if (pairs_per_iter > 0) printf("  %.2f T pairs/s", (double)grid * 256 * ITERS * pairs_per_iter / (ms * 1e-3) / 1e12);
    printf("\n");
    fflush(stdout);
}

int main()
{
    hipMalloc(&d_out, 256 * 8 * 256 * 4); hipMalloc(&d_st, 256 * 8 * 16);
    for (int w : {8, 4, 2, 1}) {
        run<0>("v_xor_b32 v,v", w, 16); run<2>("v_xor_b32 s,v", w, 16); run<1>("v_bcnt_u32_b32", w, 16);
        run<3>("v_fma_f32", w, 16); run<4>("v_fmac_f32", w, 16); run<16>("v_pk_fma_f32", w, 16);
        run<17>("xor(s)+bcnt alternating", w, 16, 1); run<18>("xor(s)x8 then bcnt x8", w, 16, 1); run<19>("xor(v)+bcnt alternating", w, 16, 1);
    }
    for (int w : {8, 4}) {
        run<5>("v_add_f32", w, 16); run<6>("v_mul_f32", w, 16); run<7>("v_min_u16", w, 16); run<21>("v_max_u16", w, 16); run<22>("v_sub_u16", w, 16);
        run<8>("v_pk_min_u16", w, 16); run<9>("v_pk_lshlrev_b16", w, 16); run<10>("v_pk_mad_u16", w, 16); run<11>("v_pk_add_u16", w, 16);
        run<12>("v_add_u32", w, 16); run<13>("v_and_b32", w, 16); run<14>("v_lshlrev_b16", w, 16); run<28>("v_lshlrev_b32", w, 16);
        run<15>("v_lshl_or_b32", w, 16); run<29>("v_and_or_b32", w, 16); run<30>("v_mov_b32", w, 16);
        run<23>("v_cmp_lt_u16+v_cndmask", w, 16); run<24>("v_or_b32_sdwa (word1)", w, 16); run<25>("v_min_u16_sdwa (word1)", w, 16);
        run<26>("v_mov_b32_dpp", w, 16); run<27>("v_min_u16_dpp", w, 16);
        run<20>("2 xor + 1 bcnt", w, 12);
        run<31>("scan pair: shl16+or+3 min16", w, 4 * (32 + 8), 8); run<32>("scan pair: lshl_or+3 min16", w, 4 * (32 + 6), 8);
        run<33>("scan pair: packed pk_mad+2 pk_min", w, 4 * (32 + 4), 8);
    }
    return 0;
}
