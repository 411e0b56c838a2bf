// reloc_orb.hip -- ORB front end on gfx950: gray -> 8-level pyramid -> FAST-9/16 + NMS ->
// best-2n by FAST score -> Harris -> best-n -> intensity-centroid angle -> 7x7 blur -> steered
// BRIEF-256.  Serves cv2.cvtColor(.., COLOR_BGR2GRAY) and
// cv2.ORB_create(nfeatures).detectAndCompute(gray, None)        (reference M:305-306, R:240-241).
//
// Everything is integer or strictly-ordered IEEE float arithmetic (no fused multiply-add, own
// sin/cos) so results are bit-identical to the specification; the algorithm constants live in
// include/reloc_spec.h.  Data layout in HBM: every pyramid level is a plane with a 64-byte-aligned
// row stride inside one arena (ctx->pyr); the blurred pyramid (ctx->blur) and the NMS score maps
// (ctx->nms) use the same geometry, so a level is addressed by one offset in all three.
//
// Launches per frame (all on the ctx stream):
//   k_pyramid                            gray conversion (or a gray plane) and all 8 levels in one launch: a tile
//                                        owns a rectangle of every level and derives level l from level l-1 in LDS
//                                        (fixed-point INTER_LINEAR_EXACT), 4 px per lane, dword stores
//   k_fast_blur                          one launch, two kinds of tiles over all levels:
//                                        FAST 32x32 tiles + halo in LDS: segment test, scores of the compacted
//                                        corners, 3x3 NMS, per-level score histogram (LDS atomics, then global);
//                                        blur 64x16 tiles staged in LDS (8.8 / 16.16 passes)
//   k_harris                             histogram -> cut score; survivors >= cut get a Harris
//                                        response and enter the per-level candidate list
//   k_select                             one workgroup per level: keep "fewer than quota strictly
//                                        greater", order raster by rank counting
//   k_describe                           one wave per keypoint: moments by wave reduction, angle,
//                                        256 steered tests -> 4 ballots = 32 descriptor bytes
// The per-frame HBM traffic is about 4 MB at 640x480; the stage is launch/latency-bound, not
// bandwidth-bound.
#include <math.h>

#include "../../include/reloc_orb_pattern.h"
#include "reloc_internal.h"

typedef uint32_t u32;

constexpr int HARRIS_CHUNK = 1024;   // bytes of the NMS map per k_harris block (4 per thread)
static_assert(HARRIS_CHUNK == 1024, "k_harris reads one dword per thread");

struct OrbTable {
    OrbLevel lev[NLEV];
    int fast_tile_base[NLEV + 1];   // 32x32 tiles over (stride x h)
    int blur_tile_base[NLEV + 1];   // 64x16 tiles over (w x h)
    int flat_base[NLEV + 1];        // HARRIS_CHUNK-byte chunks over stride*h
    int rz_off[NLEV][4];            // offsets into the resize table: xofs, xcoef, yofs, ycoef
};

__constant__ signed char c_pattern[RELOC_ORB_NTESTS * 4];
__constant__ int c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
__constant__ signed char c_ring_dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
__constant__ signed char c_ring_dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

__device__ __forceinline__ int find_level(const int *base, int id)
{
    int l = 0;
#pragma unroll
    for (int k = 1; k < NLEV; ++k) l += id >= base[k];
    return l;
}

__device__ __forceinline__ int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        if (p >= n) p = 2 * n - 2 - p;
    }
    return p;
}

// ---- gray ---------------------------------------------------------------------------------------
// `order_rgb` of the gray stages carries two flags: bit 0 = RELOC_ORDER_RGB, bit 1 = RELOC_GRAY_FLAG_15BIT (the 15-bit
// coefficient set of reloc_params.gray_coeff_bits == 15).  Both are launch-uniform: the selects run on the scalar unit.
__device__ __forceinline__ int gray_fixed(int b, int g, int r, int flags)
{
    const bool c15 = flags & RELOC_GRAY_FLAG_15BIT;
    const int cb = c15 ? RELOC_GRAY15_CB : RELOC_GRAY_CB, cg = c15 ? RELOC_GRAY15_CG : RELOC_GRAY_CG, cr = c15 ? RELOC_GRAY15_CR : RELOC_GRAY_CR;
    const int sh = c15 ? RELOC_GRAY15_SHIFT : RELOC_GRAY_SHIFT;
    return (b * cb + g * cg + r * cr + (1 << (sh - 1))) >> sh;
}
static inline int gray_flags(const reloc_ctx *ctx, int order)
{
    return (order & 1) | (ctx->prm.gray_coeff_bits == RELOC_GRAY15_SHIFT ? RELOC_GRAY_FLAG_15BIT : 0);
}
// plain gray output for reloc_gray_u8 (dense rows)
__global__ __launch_bounds__(256) void k_gray_plain(const uint8_t *__restrict__ src, int w, int h, int sstride, int order_rgb,
                                                    uint8_t *__restrict__ dst)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const uint8_t *s = src + (size_t)y * sstride + 3 * x;
    const int c0 = s[0], c1 = s[1], c2 = s[2];
    const int b = (order_rgb & 1) ? c2 : c0, r = (order_rgb & 1) ? c0 : c2;
    dst[(size_t)y * w + x] = (uint8_t)gray_fixed(b, c1, r, order_rgb);
}

// ---- fused pyramid ------------------------------------------------------------------------------
// One launch builds level 0 (gray conversion or a copy of a gray plane) AND levels 1..7.  Level l is a
// resize of level l-1; as seven launches (round 1) that was a dependent chain of tiny kernels, 4.3 us
// each, 30 us of a 75 us front end.  Here a workgroup owns one rectangle of EVERY level (`o`, 4-pixel
// aligned in x so it is stored as dwords) and computes, in LDS, the slightly larger rectangle `n` of
// each level that its rectangles of the levels above need as bilinear taps -- at most one extra
// row/column per level, so about 2x recomputation at level 0 and less above.  Per pixel the arithmetic
// is that of a plain per-level resize (same tables, same order), so the planes are bit-identical.  The rectangles and the
// table slices are worked out by the host once per frame size (orb_prepare).
#ifndef PYR_TW
#define PYR_TW 64
#define PYR_TH 32
#endif
constexpr int PT_W = PYR_TW, PT_H = PYR_TH;     // level-0 footprint of a tile
struct PyrTile {
    uint16_t o[NLEV][4];    // stored rectangle x0, x1, y0, y1 (x0 multiple of 4; x1 may reach into the row padding)
    uint16_t n[NLEV][4];    // computed rectangle (x0 multiple of 4, x1 <= level width)
};
static_assert(sizeof(PyrTile) == 128, "PyrTile is read as 8 dwordx4");

struct PyrLds { int lev[NLEV]; int tabs; };    // byte offsets of the level buffers and of the table slices in LDS

// Frame-batched launches (reloc_tick_batch_dev, the sharded halves): the five ORB kernels of up to 8 contexts as FIVE
// launches, blockIdx.y = frame.  Everything a kernel needs of one context travels in the kernel arguments.
struct OrbFrame {
    const OrbTable *tab; const PyrTile *tiles; const int32_t *rz; const uint8_t *src;
    uint8_t *pyr, *nms, *blur; int32_t *hist, *cand_cnt; u32 *cand_key; float *cand_resp; int32_t *dbg_cut;
    int32_t *kp_cnt; u32 *kp_key; float *kp_resp; float *f_xy, *f_size, *f_angle, *f_resp; int32_t *f_oct; uint8_t *f_desc;
    int32_t *f_count;
};
struct OrbBatch { OrbFrame f[RELOC_BATCH_MAX]; };

// gray value of 4 pixels from 12 interleaved bytes / a gray dword
template <int CH, bool ALIGNED>
__device__ __forceinline__ void pyr_fetch(const uint8_t *sp, int x4, int w, u32 (&d)[3])
{
    if (ALIGNED) {
        const u32 *s4 = reinterpret_cast<const u32 *>(sp);
        d[0] = s4[0];
        if (CH == 3) { d[1] = s4[1]; d[2] = s4[2]; }
    } else {
        d[0] = d[1] = d[2] = 0;
#pragma unroll
        for (int k = 0; k < 4 * CH; ++k)
            if (x4 + k / CH < w) d[k >> 2] |= (u32)sp[k] << (8 * (k & 3));
    }
}

template <int CH>
__device__ __forceinline__ u32 pyr_gray4(const u32 (&d)[3], int x4, int w, int order_rgb)
{
    if (CH == 1) return d[0];     // bytes beyond w are 0 (unaligned fetch) or do not exist (aligned: w % 4 == 0)
    u32 out = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int c[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) { const int bi = 3 * k + j; c[j] = (d[bi >> 2] >> (8 * (bi & 3))) & 0xFF; }
        const int b = (order_rgb & 1) ? c[2] : c[0], r = (order_rgb & 1) ? c[0] : c[2];
        const int g = gray_fixed(b, c[1], r, order_rgb);
        if (x4 + k < w) out |= (u32)g << (8 * k);
    }
    return out;
}

// flat table index i -> positions of the (offset, coefficient) entries in the resize tables; the level is found
// by selects over uniform values so that no load depends on it
__device__ __forceinline__ void pyr_tab_index(const OrbTable *__restrict__ tab, const PyrTile &T, const int (&tbase)[NLEV + 1], int i,
                                              int &po, int &pc)
{
    int j = i, nw = 0, x0 = 0, y0 = 0, ox = 0, cx = 0, oy = 0, cy = 0;
#pragma unroll
    for (int k = 1; k < NLEV; ++k) {
        const bool m = i >= tbase[k];
        j = m ? i - tbase[k] : j;
        nw = m ? T.n[k][1] - T.n[k][0] : nw;
        x0 = m ? (int)T.n[k][0] : x0;
        y0 = m ? (int)T.n[k][2] : y0;
        ox = m ? tab->rz_off[k][0] : ox;
        cx = m ? tab->rz_off[k][1] : cx;
        oy = m ? tab->rz_off[k][2] : oy;
        cy = m ? tab->rz_off[k][3] : cy;
    }
    const bool isx = j < nw;
    const int p = isx ? x0 + j : y0 + j - nw;
    po = (isx ? ox : oy) + p;
    pc = (isx ? cx : cy) + p;
}

// q / d for the small quad counts of a tile (q < 2^16, d <= 2^8): exact through one float multiply
__device__ __forceinline__ int pyr_div(int q, float inv) { return (int)(((float)q + 0.5f) * inv); }

// PYR_BS = threads per workgroup: 512 is the fastest alone (256 / 512 / 1024: ORB stage 73 / 68 / 66 us), 256 the best
// neighbour of a scan (a 256-thread workgroup fits into the slot one retiring scan workgroup frees: 4-stream run 6550 ->
// 6685 frames/s, synchronous tick +6 us), so both exist: see orb_run_dev.
template <int CH, bool ALIGNED, int PYR_BS>
__device__ __forceinline__ void pyramid_body(const OrbTable *__restrict__ tab, const PyrTile *__restrict__ tiles,
                                                 const int32_t *__restrict__ rz, const uint8_t *__restrict__ src, int w, int h,
                                                 int sstride, int order_rgb, uint8_t *__restrict__ pyr, PyrLds lds,
                                                 int32_t *__restrict__ hist, int32_t *__restrict__ cand_cnt)
{
    extern __shared__ u32 s_pyr[];
    const int tid = threadIdx.x;
    if (blockIdx.x == 0) {
        for (int i = tid; i < NLEV * 256; i += PYR_BS) hist[i] = 0;
        if (tid < NLEV) cand_cnt[tid] = 0;
    }
    const PyrTile &T = tiles[blockIdx.x];
    uint8_t *const base = reinterpret_cast<uint8_t *>(s_pyr);
    u32 *const tabs = s_pyr + (lds.tabs >> 2);
    // ---- phase A: every global read of the tile, issued before anything waits ---------------------
    // table slices of the levels this tile touches, one flat index over (level, x | y): offset | coefficient << 16
    int tbase[NLEV + 1];
    tbase[1] = 0;
#pragma unroll
    for (int l = 1; l < NLEV; ++l) tbase[l + 1] = tbase[l] + (T.n[l][1] - T.n[l][0]) + (T.n[l][3] - T.n[l][2]);
    constexpr int TAB_IT = 1024 / PYR_BS, L0_IT = 1536 / PYR_BS;
    u32 tv[TAB_IT][2];
#pragma unroll
    for (int it = 0; it < TAB_IT; ++it) {
        const int i = tid + it * PYR_BS;
        tv[it][0] = tv[it][1] = 0;
        if (i < tbase[NLEV]) {
            int po, pc;
            pyr_tab_index(tab, T, tbase, i, po, pc);
            tv[it][0] = (u32)rz[po];
            tv[it][1] = (u32)rz[pc];
        }
    }
    const int x00 = T.n[0][0], y00 = T.n[0][2], qpr0 = (T.n[0][1] - x00 + 3) >> 2, nq0 = qpr0 * (T.n[0][3] - y00);
    const float inv0 = 1.0f / (float)(qpr0 > 0 ? qpr0 : 1);
    u32 fv[L0_IT][3];
#pragma unroll
    for (int it = 0; it < L0_IT; ++it) {
        const int q = tid + it * PYR_BS;
        if (q < nq0) {
            const int ry = pyr_div(q, inv0), x4 = x00 + (q - ry * qpr0) * 4;
            pyr_fetch<CH, ALIGNED>(src + (size_t)(y00 + ry) * sstride + CH * x4, x4, w, fv[it]);
        }
    }
#pragma unroll
    for (int it = 0; it < TAB_IT; ++it) {
        const int i = tid + it * PYR_BS;
        if (i < tbase[NLEV]) tabs[i] = tv[it][0] | tv[it][1] << 16;
    }
#pragma unroll
    for (int it = 0; it < L0_IT; ++it) {
        const int q = tid + it * PYR_BS;
        if (q < nq0) {
            const int ry = pyr_div(q, inv0), x4 = x00 + (q - ry * qpr0) * 4;
            reinterpret_cast<u32 *>(base + lds.lev[0])[q] = pyr_gray4<CH>(fv[it], x4, w, order_rgb);      // pitch = 4 * qpr0
        }
    }
    for (int q = tid + L0_IT * PYR_BS; q < nq0; q += PYR_BS) {        // larger tiles than planned for: plain loop
        const int ry = pyr_div(q, inv0), x4 = x00 + (q - ry * qpr0) * 4;
        u32 d[3];
        pyr_fetch<CH, ALIGNED>(src + (size_t)(y00 + ry) * sstride + CH * x4, x4, w, d);
        reinterpret_cast<u32 *>(base + lds.lev[0])[q] = pyr_gray4<CH>(d, x4, w, order_rgb);
    }
    for (int i = tid + TAB_IT * PYR_BS; i < tbase[NLEV]; i += PYR_BS) {
        int po, pc;
        pyr_tab_index(tab, T, tbase, i, po, pc);
        tabs[i] = (u32)rz[po] | (u32)rz[pc] << 16;
    }
    __syncthreads();
    // ---- phase B: levels 1..7 in LDS, each in its own buffer -----------------------------------------
#pragma unroll
    for (int l = 0; l + 1 < NLEV; ++l) {
        const int sw = tab->lev[l].w, sh = tab->lev[l].h;
        const uint8_t *cur = base + lds.lev[l];
        u32 *nxt = reinterpret_cast<u32 *>(base + lds.lev[l + 1]);
        const int nx0 = T.n[l][0], ny0 = T.n[l][2], pitch = ((T.n[l][1] - nx0 + 3) >> 2) << 2;
        const int dx0 = T.n[l + 1][0], dx1 = T.n[l + 1][1], dh = T.n[l + 1][3] - T.n[l + 1][2];
        const int qpr = (dx1 - dx0 + 3) >> 2;
        const u32 *xt = tabs + tbase[l + 1], *yt = xt + (dx1 - dx0);
        const u32 one = 1u << RELOC_RESIZE_COEF_BITS;
        const float inv = 1.0f / (float)(qpr > 0 ? qpr : 1);
        for (int q = tid; q < qpr * dh; q += PYR_BS) {
            const int ry = pyr_div(q, inv), rx4 = (q - ry * qpr) * 4;
            const u32 ye = yt[ry];
            const int sy0 = (int)(ye & 0xFFFF), sy1 = sy0 + 1 < sh ? sy0 + 1 : sh - 1;
            const u32 b = ye >> 16;
            const uint8_t *r0 = cur + (sy0 - ny0) * pitch - nx0, *r1 = cur + (sy1 - ny0) * pitch - nx0;
            u32 out = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (dx0 + rx4 + k < dx1) {
                    const u32 xe = xt[rx4 + k];
                    const int k0 = (int)(xe & 0xFFFF), k1 = k0 + 1 < sw ? k0 + 1 : sw - 1;
                    const u32 a = xe >> 16;
                    const u32 h0 = (u32)r0[k0] * (one - a) + (u32)r0[k1] * a;
                    const u32 h1 = (u32)r1[k0] * (one - a) + (u32)r1[k1] * a;
                    const u32 v = h0 * (one - b) + h1 * b;
                    out |= ((v + (1u << 15)) >> 16) << (8 * k);
                }
            }
            nxt[q] = out;      // pitch = 4 * qpr
        }
        __syncthreads();
    }
    // ---- phase C: every level's own rectangle to HBM, nothing waits on these stores -------------------
#pragma unroll
    for (int l = 0; l < NLEV; ++l) {
        const OrbLevel L = tab->lev[l];
        const uint8_t *cur = base + lds.lev[l];
        const int nx0 = T.n[l][0], ny0 = T.n[l][2], pitch = ((T.n[l][1] - nx0 + 3) >> 2) << 2;
        const int x0 = T.o[l][0], x1 = T.o[l][1], y0 = T.o[l][2], oh = T.o[l][3] - y0;
        const int qpr = (x1 - x0) >> 2;
        uint8_t *dst = pyr + L.off;
        const float inv = 1.0f / (float)(qpr > 0 ? qpr : 1);
        for (int q = tid; q < qpr * oh; q += PYR_BS) {
            const int ry = pyr_div(q, inv), x4 = x0 + (q - ry * qpr) * 4;
            u32 v = 0;
            if (x4 < L.w) {
                v = *reinterpret_cast<const u32 *>(cur + (y0 + ry - ny0) * pitch + (x4 - nx0));
                if (x4 + 4 > L.w) v &= 0xFFFFFFFFu >> (8 * (x4 + 4 - L.w));
            }
            *reinterpret_cast<u32 *>(dst + (size_t)(y0 + ry) * L.stride + x4) = v;
        }
    }
}
template <int CH, bool ALIGNED, int PYR_BS>
__global__ __launch_bounds__(PYR_BS) void k_pyramid(const OrbTable *__restrict__ tab, const PyrTile *__restrict__ tiles,
                                                 const int32_t *__restrict__ rz, const uint8_t *__restrict__ src, int w, int h,
                                                 int sstride, int order_rgb, uint8_t *__restrict__ pyr, PyrLds lds,
                                                 int32_t *__restrict__ hist, int32_t *__restrict__ cand_cnt)
{
    RELOC_SMALL_KERNEL_PRIO();
    pyramid_body<CH, ALIGNED, PYR_BS>(tab, tiles, rz, src, w, h, sstride, order_rgb, pyr, lds, hist, cand_cnt);
}
template <int CH, bool ALIGNED, int PYR_BS>
__global__ __launch_bounds__(PYR_BS) void k_pyramid_batch(OrbBatch b, int w, int h, int sstride, int order_rgb, PyrLds lds)
{
    RELOC_SMALL_KERNEL_PRIO();
    const OrbFrame &F = b.f[blockIdx.y];
    pyramid_body<CH, ALIGNED, PYR_BS>(F.tab, F.tiles, F.rz, F.src, w, h, sstride, order_rgb, F.pyr, lds, F.hist, F.cand_cnt);
}


// ---- blur ---------------------------------------------------------------------------------------
constexpr int BT_W = 64, BT_H = 16;
__device__ __forceinline__ void blur7_tile(const OrbTable *__restrict__ tab, const uint8_t *__restrict__ pyr,
                                           uint8_t *__restrict__ blur, int bid)
{
    __shared__ uint8_t s_in[(BT_H + 6) * (BT_W + 8)];
    __shared__ uint16_t s_h[(BT_H + 6) * BT_W];
    const int l = find_level(tab->blur_tile_base, bid);
    const OrbLevel L = tab->lev[l];
    const int tile = bid - tab->blur_tile_base[l];
    const int tx = (L.w + BT_W - 1) / BT_W;
    const int x0 = (tile % tx) * BT_W, y0 = (tile / tx) * BT_H;
    const uint8_t *src = pyr + L.off;
    const int tid = threadIdx.x;
    for (int i = tid; i < (BT_H + 6) * (BT_W + 6); i += 256) {
        const int ry = i / (BT_W + 6), rx = i % (BT_W + 6);
        const int gy = reflect101(y0 + ry - 3, L.h), gx = reflect101(x0 + rx - 3, L.w);
        s_in[ry * (BT_W + 8) + rx] = src[(size_t)gy * L.stride + gx];
    }
    __syncthreads();
    for (int i = tid; i < (BT_H + 6) * BT_W; i += 256) {
        const int ry = i / BT_W, rx = i % BT_W;
        const uint8_t *p = s_in + ry * (BT_W + 8) + rx;
        const u32 s = RELOC_BLUR_K0 * (p[0] + p[6]) + RELOC_BLUR_K1 * (p[1] + p[5]) + RELOC_BLUR_K2 * (p[2] + p[4]) + RELOC_BLUR_K3 * p[3];
        s_h[i] = (uint16_t)s;
    }
    __syncthreads();
    // 4 output pixels per lane
    {
        const int ry = tid / 16, rx4 = (tid % 16) * 4;
        u32 out = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint16_t *p = s_h + ry * BT_W + rx4 + k;
            const u32 s = RELOC_BLUR_K0 * ((u32)p[0] + p[6 * BT_W]) + RELOC_BLUR_K1 * ((u32)p[BT_W] + p[5 * BT_W]) +
                          RELOC_BLUR_K2 * ((u32)p[2 * BT_W] + p[4 * BT_W]) + RELOC_BLUR_K3 * (u32)p[3 * BT_W];
            out |= ((s + (1u << 15)) >> 16) << (8 * k);
        }
        const int gy = y0 + ry, gx = x0 + rx4;
        if (gy < L.h && gx < L.stride) *reinterpret_cast<u32 *>(blur + L.off + (size_t)gy * L.stride + gx) = out;
    }
}

// ---- FAST + NMS ---------------------------------------------------------------------------------
// segment test: 0 = not a corner, +1 = 9 contiguous brighter, -1 = 9 contiguous darker ring pixels
__device__ int fast_is_corner(const uint8_t *p, int stride, int thr)
{
    const int c = p[0];
    u32 bright = 0, dark = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int v = (int)p[c_ring_dy[k] * stride + c_ring_dx[k]] - c;
        bright |= (u32)(v > thr) << k;
        dark |= (u32)(v < -thr) << k;
    }
    // 9 contiguous set bits on the 16-bit circle
    auto run9 = [](u32 m) {
        u32 x = m | (m << 16);
        u32 y = x & (x >> 1);
        y &= y >> 2;
        y &= y >> 4;          // runs of 8
        y &= x >> 8;          // runs of 9
        return (y & 0xFFFFu) != 0;
    };
    return run9(bright) ? 1 : (run9(dark) ? -1 : 0);
}

// corner score of a pixel that passed the segment test with polarity `sign`: the largest threshold it still
// passes, i.e. max over the 16 arcs of 9 of the arc's minimum |difference|, minus 1
__device__ int fast_corner_score(const uint8_t *p, int stride, int thr, int sign)
{
    const int c = p[0];
    int v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = sign * ((int)p[c_ring_dy[k] * stride + c_ring_dx[k]] - c);
    int m2[16], m4[16], best = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) m2[k] = min(v[k], v[(k + 1) & 15]);
#pragma unroll
    for (int k = 0; k < 16; ++k) m4[k] = min(m2[k], m2[(k + 2) & 15]);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int m8 = min(m4[k], m4[(k + 4) & 15]);
        const int m9 = min(m8, v[(k + 8) & 15]);
        best = max(best, m9);
    }
    return best > thr ? best - 1 : 0;
}

constexpr int FT = 32;
__device__ __forceinline__ void fast_nms_tile(const OrbTable *__restrict__ tab, const uint8_t *__restrict__ pyr,
                                              uint8_t *__restrict__ nms, int32_t *__restrict__ hist, int bid)
{
    __shared__ uint8_t s_img[(FT + 8) * (FT + 12)];
    __shared__ uint8_t s_sc[(FT + 2) * (FT + 4)];
    __shared__ unsigned short s_corner[(FT + 2) * (FT + 2)];
    __shared__ int s_nc;
    __shared__ int s_hist[256];
    const int l = find_level(tab->fast_tile_base, bid);
    const OrbLevel L = tab->lev[l];
    const int tile = bid - tab->fast_tile_base[l];
    const int tx = L.stride / FT;
    const int x0 = (tile % tx) * FT, y0 = (tile / tx) * FT;
    const int tid = threadIdx.x;
    const int e = RELOC_ORB_EDGE;
    uint8_t *out = nms + L.off;
    // tiles that cannot hold a kept corner only clear their part of the map
    const bool live = L.quota > 0 && x0 + FT > e && x0 < L.w - e && y0 + FT > e && y0 < L.h - e;
    if (!live) {
        const int y = y0 + tid / 8, x = x0 + (tid % 8) * 4;
        if (y < L.h) *reinterpret_cast<u32 *>(out + (size_t)y * L.stride + x) = 0;
        return;
    }
    s_hist[tid] = 0;
    const uint8_t *src = pyr + L.off;
    const int IS = FT + 12;
    for (int i = tid; i < (FT + 8) * (FT + 8); i += 256) {
        const int ry = i / (FT + 8), rx = i % (FT + 8);
        const int gy = min(max(y0 + ry - 4, 0), L.h - 1), gx = min(max(x0 + rx - 4, 0), L.w - 1);
        s_img[ry * IS + rx] = src[(size_t)gy * L.stride + gx];
    }
    __syncthreads();
    // segment test for every pixel of the tile + 1-px ring; the few corners are compacted into a list so that
    // the score (as long as the test itself) runs on full waves of corners instead of on every wave that
    // happens to contain one.  (Compacting the pretest survivors as well, so that the segment test too runs on full waves:
    // 21 % fewer instructions and no faster -- profiles/README.md "Dropped experiments" #7.)
    const int SS = FT + 4;
    if (tid == 0) s_nc = 0;
    __syncthreads();
    for (int i = tid; i < (FT + 2) * (FT + 2); i += 256) {
        const int ry = i / (FT + 2), rx = i % (FT + 2);
        const int gy = y0 + ry - 1, gx = x0 + rx - 1;
        int pol = 0;
        // The usual compass pretest, as a WAVE decision: an arc of 9 of the 16 ring pixels contains at least two of the four
        // compass pixels (ring positions 0, 4, 8, 12), so a pixel with fewer than two brighter and fewer than two darker
        // compass pixels is no corner.  Lanes cannot skip work on their own, but a wave whose 64 pixels all fail (flat
        // ground, sky, the inside of uniform shapes) skips the 16-pixel segment test altogether; the outcome is the same.
        bool cand = false;
        const uint8_t *pc = s_img + (ry + 3) * IS + (rx + 3);
        const bool inside = i < (FT + 2) * (FT + 2) && gx >= 3 && gx < L.w - 3 && gy >= 3 && gy < L.h - 3;
        if (inside) {
            const int c = pc[0];
            const int v0 = (int)pc[3 * IS] - c, v4 = (int)pc[3] - c, v8 = (int)pc[-3 * IS] - c, v12 = (int)pc[-3] - c;
            const int nb = (v0 > RELOC_FAST_THRESHOLD) + (v4 > RELOC_FAST_THRESHOLD) + (v8 > RELOC_FAST_THRESHOLD) + (v12 > RELOC_FAST_THRESHOLD);
            const int nd = (v0 < -RELOC_FAST_THRESHOLD) + (v4 < -RELOC_FAST_THRESHOLD) + (v8 < -RELOC_FAST_THRESHOLD) + (v12 < -RELOC_FAST_THRESHOLD);
            cand = nb >= 2 || nd >= 2;
        }
        if (__any(cand)) {
            if (cand) pol = fast_is_corner(pc, IS, RELOC_FAST_THRESHOLD);
        }
        s_sc[ry * SS + rx] = 0;
        if (pol) s_corner[atomicAdd(&s_nc, 1)] = (unsigned short)((ry << 6) | rx | (pol < 0 ? 0x8000 : 0));
    }
    __syncthreads();
    for (int i = tid; i < s_nc; i += 256) {
        const int e16 = s_corner[i], ry = (e16 >> 6) & 63, rx = e16 & 63;
        s_sc[ry * SS + rx] = (uint8_t)fast_corner_score(s_img + (ry + 3) * IS + (rx + 3), IS, RELOC_FAST_THRESHOLD, (e16 & 0x8000) ? -1 : 1);
    }
    __syncthreads();
    {
        const int ry = tid / 8, rx4 = (tid % 8) * 4;
        u32 word = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int gx = x0 + rx4 + k, gy = y0 + ry;
            const uint8_t *s = s_sc + (ry + 1) * SS + (rx4 + k + 1);
            const int c = s[0];
            if (c && gx >= e && gx < L.w - e && gy >= e && gy < L.h - e && c > s[-1] && c > s[1] && c > s[-SS - 1] &&
                c > s[-SS] && c > s[-SS + 1] && c > s[SS - 1] && c > s[SS] && c > s[SS + 1]) {
                word |= (u32)c << (8 * k);
                atomicAdd(&s_hist[c], 1);
            }
        }
        const int gy = y0 + ry;
        if (gy < L.h) *reinterpret_cast<u32 *>(out + (size_t)gy * L.stride + x0 + rx4) = word;
    }
    __syncthreads();
    if (s_hist[tid]) atomicAdd(&hist[l * 256 + tid], s_hist[tid]);
}

// ---- stage 1 cut + Harris -----------------------------------------------------------------------
__device__ float harris_px(const uint8_t *p, int step)
{
    int a = 0, b = 0, c = 0;
#pragma unroll
    for (int dy = -3; dy <= 3; ++dy)
#pragma unroll
        for (int dx = -3; dx <= 3; ++dx) {
            const uint8_t *q = p + dy * step + dx;
            const int ix = (q[1] - q[-1]) * 2 + (q[-step + 1] - q[-step - 1]) + (q[step + 1] - q[step - 1]);
            const int iy = (q[step] - q[-step]) * 2 + (q[step - 1] - q[-step - 1]) + (q[step + 1] - q[-step + 1]);
            a += ix * ix;
            b += iy * iy;
            c += ix * iy;
        }
    const float scale = __fdiv_rn(1.f, (float)((1 << 2) * RELOC_HARRIS_BLOCK) * 255.f);
    const float s2 = __fmul_rn(scale, scale), s3 = __fmul_rn(s2, scale), s4 = __fmul_rn(s3, scale);
    const float fa = (float)a, fb = (float)b, fc = (float)c;
    const float t1 = __fmul_rn(fa, fb), t2 = __fmul_rn(fc, fc), t3 = __fadd_rn(fa, fb);
    const float t4 = __fmul_rn(RELOC_HARRIS_K, t3), t5 = __fmul_rn(t4, t3);
    const float t6 = __fsub_rn(t1, t2), t7 = __fsub_rn(t6, t5);
    return __fmul_rn(t7, s4);
}

// FAST + NMS tiles and 7x7 blur tiles of all levels in ONE launch, one workgroup per tile: both only read the pyramid, FAST
// feeds Harris and the blur feeds the descriptors, so they need not run one after the other.  (A smaller grid whose workgroups
// walk the tiles paid beside 112-register scans and pays nothing beside 104-register ones: profiles/README.md "Dropped
// experiments" #8.)
__global__ __launch_bounds__(256) void k_fast_blur(const OrbTable *__restrict__ tab, const uint8_t *__restrict__ pyr,
                                                   uint8_t *__restrict__ nms, int32_t *__restrict__ hist,
                                                   uint8_t *__restrict__ blur, int n_fast)
{
    RELOC_SMALL_KERNEL_PRIO();
    if ((int)blockIdx.x < n_fast) fast_nms_tile(tab, pyr, nms, hist, (int)blockIdx.x);
    else blur7_tile(tab, pyr, blur, (int)blockIdx.x - n_fast);
}
__global__ __launch_bounds__(256) void k_fast_blur_batch(OrbBatch b, int n_fast)
{
    RELOC_SMALL_KERNEL_PRIO();
    const OrbFrame &F = b.f[blockIdx.y];
    if ((int)blockIdx.x < n_fast) fast_nms_tile(F.tab, F.pyr, F.nms, F.hist, (int)blockIdx.x);
    else blur7_tile(F.tab, F.pyr, F.blur, (int)blockIdx.x - n_fast);
}


// cut score from the level's histogram (KeyPointsFilter::retainBest(2*quota) with ties kept, raised
// while the kept set exceeds RELOC_ORB_STAGE1_CAP), by ONE wave without block barriers: lane i owns the
// bins 4i .. 4i+3.  c(s) = sum_{k >= s} hist[k] is non-increasing in s;
//   cut0 = max{s : c(s) >= n_keep} if c(0) > n_keep, else the FAST threshold (everything is kept);
//   cut  = min{s >= cut0 : c(s) <= CAP or s == 255}.
// Every lane returns the cut.
__device__ int stage1_cut_wave(const int32_t *__restrict__ hist_l, int n_keep, int lane)
{
    const int4 h = *reinterpret_cast<const int4 *>(hist_l + 4 * lane);
    // exclusive suffix sum of the lane totals
    const int mine = h.x + h.y + h.z + h.w;
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_down(incl, d);
        if (lane + d < 64) incl += v;
    }
    const int above = incl - mine;
    const int c3 = h.w + above, c2 = h.z + c3, c1 = h.y + c2, c0 = h.x + c1;     // c(4i+3) .. c(4i)
    const int total = __builtin_amdgcn_readlane(c0, 0);
    int cut0 = RELOC_FAST_THRESHOLD;
    if (total > n_keep) {
        const int s = c3 >= n_keep ? 3 : (c2 >= n_keep ? 2 : (c1 >= n_keep ? 1 : (c0 >= n_keep ? 0 : -1)));
        cut0 = (int)wave_max_u32(s < 0 ? 0u : (unsigned)(4 * lane + s + 1)) - 1;
    }
    // smallest s >= cut0 with c(s) <= CAP (or 255): largest of 256 - s over the admissible bins
    unsigned best = 0;
    const int cs[4] = {c0, c1, c2, c3};
#pragma unroll
    for (int k = 3; k >= 0; --k) {
        const int s = 4 * lane + k;
        if (s >= cut0 && (cs[k] <= RELOC_ORB_STAGE1_CAP || s == 255)) best = (unsigned)(256 - s);
    }
    return 256 - (int)wave_max_u32(best);
}

// Harris response of one pixel by a whole wave: lanes 0..48 own one pixel of the 7x7 block each.
// harris_terms: the lane's three products (its byte loads); harris_finish: wave sums + the float formula.
__device__ __forceinline__ void harris_terms(const uint8_t *p, int step, int lane, int &a, int &b, int &c)
{
    a = b = c = 0;
    if (lane < 49) {
        const int dy = lane / 7 - 3, dx = lane % 7 - 3;
        const uint8_t *q = p + dy * step + dx;
        const int ix = (q[1] - q[-1]) * 2 + (q[-step + 1] - q[-step - 1]) + (q[step + 1] - q[step - 1]);
        const int iy = (q[step] - q[-step]) * 2 + (q[step - 1] - q[-step - 1]) + (q[step + 1] - q[-step + 1]);
        a = ix * ix; b = iy * iy; c = ix * iy;
    }
}

__device__ __forceinline__ float harris_finish(int a, int b, int c)
{
    a = wave_sum_i32(a);                  // integer sums: exact in any order
    b = wave_sum_i32(b);
    c = wave_sum_i32(c);
    const float scale = __fdiv_rn(1.f, (float)((1 << 2) * RELOC_HARRIS_BLOCK) * 255.f);
    const float s2 = __fmul_rn(scale, scale), s3 = __fmul_rn(s2, scale), s4 = __fmul_rn(s3, scale);
    const float fa = (float)a, fb = (float)b, fc = (float)c;
    const float t1 = __fmul_rn(fa, fb), t2 = __fmul_rn(fc, fc), t3 = __fadd_rn(fa, fb);
    const float t4 = __fmul_rn(RELOC_HARRIS_K, t3), t5 = __fmul_rn(t4, t3);
    const float t6 = __fsub_rn(t1, t2), t7 = __fsub_rn(t6, t5);
    return __fmul_rn(t7, s4);
}

// Each block scans HARRIS_CHUNK bytes of the NMS map, collects the survivors >= cut in LDS, then its four waves
// compute their Harris responses (one wave per survivor) and append them to the level's candidate list.
// Corners cluster, and a block works through its survivors four at a time: small chunks keep the longest
// block short (ORB stage 77.5 us with 4096-byte chunks, 75.7 us with 1024-byte chunks).
__device__ __forceinline__ void harris_body(const OrbTable *__restrict__ tab, const uint8_t *__restrict__ pyr,
                                                const uint8_t *__restrict__ nms, const int32_t *__restrict__ hist,
                                                int32_t *__restrict__ cand_cnt, u32 *__restrict__ cand_key,
                                                float *__restrict__ cand_resp, int32_t *__restrict__ dbg_cut)
{
    __shared__ int s_cut;
    __shared__ int s_n, s_base;
    __shared__ u32 s_list[HARRIS_CHUNK];
    const int l = find_level(tab->flat_base, blockIdx.x);
    const OrbLevel L = tab->lev[l];
    if (L.quota <= 0 || L.w <= 2 * RELOC_ORB_EDGE || L.h <= 2 * RELOC_ORB_EDGE) return;
    const int64_t idx0 = ((int64_t)(blockIdx.x - tab->flat_base[l]) * 256 + threadIdx.x) * (HARRIS_CHUNK / 256);
    const int64_t total = (int64_t)L.stride * L.h;
    u32 v = 0;
    if (idx0 < total) v = *reinterpret_cast<const u32 *>(nms + L.off + idx0);
    const bool any = v != 0;
    if (threadIdx.x == 0) s_n = 0;
    if (!__syncthreads_or(any)) return;                      // nothing kept in this chunk: skip the cut computation
    if (threadIdx.x < 64) {
        const int c = stage1_cut_wave(hist + l * 256, 2 * L.quota, threadIdx.x);
        if (threadIdx.x == 0) {
            s_cut = c;
            if (dbg_cut) dbg_cut[l] = c;
        }
    }
    __syncthreads();
    const int cut = s_cut;
    if (any) {
#pragma unroll
        for (int k = 0; k < HARRIS_CHUNK / 256; ++k) {
            const int sc = (v >> (8 * k)) & 0xFF;
            if (sc && sc >= cut) {
                const int64_t idx = idx0 + k;
                const int y = (int)(idx / L.stride), x = (int)(idx % L.stride);
                s_list[atomicAdd(&s_n, 1)] = ((u32)y << 16) | (u32)x;
            }
        }
    }
    __syncthreads();
    const int n = s_n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint8_t *img = pyr + L.off;
    // ONE slot reservation per block, its round trip hidden behind the first responses (the per-survivor atomicAdd it
    // replaces was a second memory round trip in every turn); each wave takes two survivors per turn so that the pixel
    // loads of the second are in flight while the first is reduced.
    int base_ret = 0;
    if (threadIdx.x == 0 && n > 0) base_ret = atomicAdd(&cand_cnt[l], n);
    auto terms = [&](u32 key, int &a, int &b, int &c) {
        harris_terms(img + (size_t)(key >> 16) * L.stride + (key & 0xFFFF), L.stride, lane, a, b, c);
    };
    int i = wave * 2;
    const bool first = i < n;                                  // wave-uniform
    u32 k0 = 0, k1 = 0;
    int a0 = 0, b0 = 0, c0 = 0, a1 = 0, b1 = 0, c1 = 0;
    float h0 = 0.f, h1 = 0.f;
    if (first) {
        k0 = s_list[i]; k1 = s_list[i + 1 < n ? i + 1 : i];
        terms(k0, a0, b0, c0);
        terms(k1, a1, b1, c1);
    }
    if (threadIdx.x == 0) s_base = base_ret;
    if (first) { h0 = harris_finish(a0, b0, c0); h1 = harris_finish(a1, b1, c1); }
    __syncthreads();
    const int base = s_base;
    auto put = [&](int at, u32 key, float r) {
        const int pos = base + at;
        if (pos < RELOC_ORB_STAGE1_CAP) {
            cand_key[(size_t)l * RELOC_ORB_STAGE1_CAP + pos] = key;
            cand_resp[(size_t)l * RELOC_ORB_STAGE1_CAP + pos] = r;
        }
    };
    if (first && lane == 0) {
        put(i, k0, h0);
        if (i + 1 < n) put(i + 1, k1, h1);
    }
    for (i += 8; i < n; i += 8) {
        k0 = s_list[i]; k1 = s_list[i + 1 < n ? i + 1 : i];
        terms(k0, a0, b0, c0);
        terms(k1, a1, b1, c1);
        h0 = harris_finish(a0, b0, c0);
        h1 = harris_finish(a1, b1, c1);
        if (lane == 0) {
            put(i, k0, h0);
            if (i + 1 < n) put(i + 1, k1, h1);
        }
    }
}
__global__ __launch_bounds__(256) void k_harris(const OrbTable *__restrict__ tab, const uint8_t *__restrict__ pyr,
                                                const uint8_t *__restrict__ nms, const int32_t *__restrict__ hist,
                                                int32_t *__restrict__ cand_cnt, u32 *__restrict__ cand_key,
                                                float *__restrict__ cand_resp, int32_t *__restrict__ dbg_cut)
{
    RELOC_SMALL_KERNEL_PRIO();
    harris_body(tab, pyr, nms, hist, cand_cnt, cand_key, cand_resp, dbg_cut);
}
__global__ __launch_bounds__(256) void k_harris_batch(OrbBatch b)
{
    RELOC_SMALL_KERNEL_PRIO();
    const OrbFrame &F = b.f[blockIdx.y];
    harris_body(F.tab, F.pyr, F.nms, F.hist, F.cand_cnt, F.cand_key, F.cand_resp, F.dbg_cut);
}


// ---- stage 2: best quota by Harris (ties kept), raster order -----------------------------------
// One workgroup per level.  Element i is kept iff fewer than `quota` responses are strictly greater
// (= best quota plus every tie of the quota-th); the kept ones are then placed in raster order by
// counting smaller keys.  Quadratic in the list length, which is ~2*quota (a few hundred).
__device__ __forceinline__ void select_body(const OrbTable *__restrict__ tab, const int32_t *__restrict__ cand_cnt,
                                                 const u32 *__restrict__ cand_key, const float *__restrict__ cand_resp,
                                                 int32_t *__restrict__ kp_cnt, u32 *__restrict__ kp_key,
                                                 float *__restrict__ kp_resp)
{
    __shared__ u32 s_key[RELOC_ORB_STAGE1_CAP];
    __shared__ float s_resp[RELOC_ORB_STAGE1_CAP];
    __shared__ u32 s_kidx[RELOC_ORB_STAGE1_CAP];
    __shared__ int s_nk;
    const int l = blockIdx.x;
    const int quota = tab->lev[l].quota;
    const int M = min(cand_cnt[l], RELOC_ORB_STAGE1_CAP);
    const int tid = threadIdx.x;
    if (tid == 0) s_nk = 0;
    for (int i = tid; i < M; i += 1024) {
        s_key[i] = cand_key[(size_t)l * RELOC_ORB_STAGE1_CAP + i];
        s_resp[i] = cand_resp[(size_t)l * RELOC_ORB_STAGE1_CAP + i];
    }
    __syncthreads();
    for (int i = tid; i < M; i += 1024) {
        const float r = s_resp[i];
        int greater = 0;
        for (int j = 0; j < M; ++j) greater += s_resp[j] > r;
        if (greater < quota) s_kidx[atomicAdd(&s_nk, 1)] = (u32)i;
    }
    __syncthreads();
    const int K = s_nk;
    for (int i = tid; i < K; i += 1024) {
        const u32 src = s_kidx[i];
        const u32 key = s_key[src];
        int pos = 0;
        for (int j = 0; j < K; ++j) pos += s_key[s_kidx[j]] < key;
        kp_key[(size_t)l * RELOC_ORB_STAGE1_CAP + pos] = key;
        kp_resp[(size_t)l * RELOC_ORB_STAGE1_CAP + pos] = s_resp[src];
    }
    if (tid == 0) kp_cnt[l] = K;
}
__global__ __launch_bounds__(1024) void k_select(const OrbTable *__restrict__ tab, const int32_t *__restrict__ cand_cnt,
                                                 const u32 *__restrict__ cand_key, const float *__restrict__ cand_resp,
                                                 int32_t *__restrict__ kp_cnt, u32 *__restrict__ kp_key,
                                                 float *__restrict__ kp_resp)
{
    RELOC_SMALL_KERNEL_PRIO();
    select_body(tab, cand_cnt, cand_key, cand_resp, kp_cnt, kp_key, kp_resp);
}
__global__ __launch_bounds__(1024) void k_select_batch(OrbBatch b)
{
    RELOC_SMALL_KERNEL_PRIO();
    const OrbFrame &F = b.f[blockIdx.y];
    select_body(F.tab, F.cand_cnt, F.cand_key, F.cand_resp, F.kp_cnt, F.kp_key, F.kp_resp);
}


// ---- orientation + descriptor -------------------------------------------------------------------
__device__ float fast_atan2_deg(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    float a;
    if (ax >= ay) {
        const float c = __fdiv_rn(ay, __fadd_rn(ax, RELOC_ATAN2_EPS));
        const float c2 = __fmul_rn(c, c);
        float t = __fadd_rn(__fmul_rn(RELOC_ATAN2_P7, c2), RELOC_ATAN2_P5);
        t = __fadd_rn(__fmul_rn(t, c2), RELOC_ATAN2_P3);
        t = __fadd_rn(__fmul_rn(t, c2), RELOC_ATAN2_P1);
        a = __fmul_rn(t, c);
    } else {
        const float c = __fdiv_rn(ax, __fadd_rn(ay, RELOC_ATAN2_EPS));
        const float c2 = __fmul_rn(c, c);
        float t = __fadd_rn(__fmul_rn(RELOC_ATAN2_P7, c2), RELOC_ATAN2_P5);
        t = __fadd_rn(__fmul_rn(t, c2), RELOC_ATAN2_P3);
        t = __fadd_rn(__fmul_rn(t, c2), RELOC_ATAN2_P1);
        t = __fmul_rn(t, c);
        a = __fsub_rn(90.f, t);
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

__device__ void sincos_spec(double th, float *s_out, float *c_out)
{
    const double PIO2_HI = 1.57079632679489655800e+00;
    const double PIO2_LO = 6.12323399573676603587e-17;
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double kd = floor(__dadd_rn(__dmul_rn(th, TWO_OVER_PI), 0.5));
    const int k = (int)kd;
    double r = __dsub_rn(th, __dmul_rn(kd, PIO2_HI));
    r = __dsub_rn(r, __dmul_rn(kd, PIO2_LO));
    const double r2 = __dmul_rn(r, r);
    const double S[8] = {-1.0 / 6.0, 1.0 / 120.0, -1.0 / 5040.0, 1.0 / 362880.0, -1.0 / 39916800.0,
                         1.0 / 6227020800.0, -1.0 / 1307674368000.0, 1.0 / 355687428096000.0};
    const double C[8] = {-1.0 / 2.0, 1.0 / 24.0, -1.0 / 720.0, 1.0 / 40320.0, -1.0 / 3628800.0,
                         1.0 / 479001600.0, -1.0 / 87178291200.0, 1.0 / 20922789888000.0};
    double ps = S[7], pc = C[7];
#pragma unroll
    for (int i = 6; i >= 0; --i) {
        ps = __dadd_rn(__dmul_rn(ps, r2), S[i]);
        pc = __dadd_rn(__dmul_rn(pc, r2), C[i]);
    }
    ps = __dmul_rn(__dadd_rn(__dmul_rn(ps, r2), 1.0), r);
    pc = __dadd_rn(__dmul_rn(pc, r2), 1.0);
    double sn, cs;
    switch (k & 3) {
    case 0: sn = ps; cs = pc; break;
    case 1: sn = pc; cs = -ps; break;
    case 2: sn = -ps; cs = -pc; break;
    default: sn = -pc; cs = ps; break;
    }
    *s_out = (float)sn;
    *c_out = (float)cs;
}

// one wave per keypoint; block = 4 waves
__device__ __forceinline__ void describe_body(const OrbTable *__restrict__ tab, const uint8_t *__restrict__ pyr,
                                                  const uint8_t *__restrict__ blur, const int32_t *__restrict__ kp_cnt,
                                                  const u32 *__restrict__ kp_key, const float *__restrict__ kp_resp,
                                                  int max_feat, float *__restrict__ f_xy, float *__restrict__ f_size,
                                                  float *__restrict__ f_angle, float *__restrict__ f_resp,
                                                  int32_t *__restrict__ f_oct, uint8_t *__restrict__ f_desc,
                                                  int32_t *__restrict__ f_count)
{
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    int base[NLEV + 1];
    base[0] = 0;
#pragma unroll
    for (int l = 0; l < NLEV; ++l) base[l + 1] = base[l] + kp_cnt[l];
    const int total = base[NLEV];
    if (g == 0 && lane == 0) *f_count = min(total, max_feat);
    if (g >= total || g >= max_feat) return;
    int l = 0;
#pragma unroll
    for (int k = 1; k < NLEV; ++k) l += g >= base[k];
    const OrbLevel L = tab->lev[l];
    const int i = g - base[l];
    const u32 key = kp_key[(size_t)l * RELOC_ORB_STAGE1_CAP + i];
    const int x = key & 0xFFFF, y = key >> 16;
    const uint8_t *center = pyr + L.off + (size_t)y * L.stride + x;
    // intensity-centroid moments: 31 rows, lane = column offset (-15..15 -> lanes 0..30), two rows at a time
    int m10 = 0, m01 = 0;
    {
        const int u = (lane & 31) - 15;            // -15..16 (16 unused)
        const int half = lane >> 5;                // rows split between the two half-waves
        for (int vv = half; vv <= 30; vv += 2) {
            const int v = vv - 15;
            const int av = v < 0 ? -v : v;
            if (u <= 15 && (u < 0 ? -u : u) <= c_umax[av]) {
                const int I = center[v * L.stride + u];
                m10 += u * I;
                m01 += v * I;
            }
        }
        m10 = wave_sum_i32(m10);
        m01 = wave_sum_i32(m01);
    }
    const float angle = fast_atan2_deg((float)m01, (float)m10);
    float sn, cs;
    sincos_spec((double)__fmul_rn(angle, RELOC_DEG2RAD_F), &sn, &cs);
    const uint8_t *bc = blur + L.off + (size_t)y * L.stride + x;
    unsigned long long bits[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const signed char *t = c_pattern + 4 * (64 * it + lane);
        int val[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float px = (float)t[2 * e], py = (float)t[2 * e + 1];
            const float rx = __fsub_rn(__fmul_rn(px, cs), __fmul_rn(py, sn));
            const float ry = __fadd_rn(__fmul_rn(px, sn), __fmul_rn(py, cs));
            const int ix = (int)rintf(rx), iy = (int)rintf(ry);
            val[e] = bc[iy * L.stride + ix];
        }
        bits[it] = __ballot(val[0] < val[1]);
    }
    if (lane < 4) {
        const unsigned long long b = lane == 0 ? bits[0] : lane == 1 ? bits[1] : lane == 2 ? bits[2] : bits[3];
        reinterpret_cast<unsigned long long *>(f_desc + (size_t)g * 32)[lane] = b;
    }
    if (lane == 0) {
        f_xy[2 * g] = __fmul_rn((float)x, L.scale);
        f_xy[2 * g + 1] = __fmul_rn((float)y, L.scale);
        f_size[g] = __fmul_rn((float)RELOC_ORB_PATCH, L.scale);
        f_angle[g] = angle;
        f_resp[g] = kp_resp[(size_t)l * RELOC_ORB_STAGE1_CAP + i];
        f_oct[g] = l;
    }
}
__global__ __launch_bounds__(256) void k_describe(const OrbTable *__restrict__ tab, const uint8_t *__restrict__ pyr,
                                                  const uint8_t *__restrict__ blur, const int32_t *__restrict__ kp_cnt,
                                                  const u32 *__restrict__ kp_key, const float *__restrict__ kp_resp,
                                                  int max_feat, float *__restrict__ f_xy, float *__restrict__ f_size,
                                                  float *__restrict__ f_angle, float *__restrict__ f_resp,
                                                  int32_t *__restrict__ f_oct, uint8_t *__restrict__ f_desc,
                                                  int32_t *__restrict__ f_count)
{
    RELOC_SMALL_KERNEL_PRIO();
    describe_body(tab, pyr, blur, kp_cnt, kp_key, kp_resp, max_feat, f_xy, f_size, f_angle, f_resp, f_oct, f_desc, f_count);
}
__global__ __launch_bounds__(256) void k_describe_batch(OrbBatch b, int max_feat)
{
    RELOC_SMALL_KERNEL_PRIO();
    const OrbFrame &F = b.f[blockIdx.y];
    describe_body(F.tab, F.pyr, F.blur, F.kp_cnt, F.kp_key, F.kp_resp, max_feat, F.f_xy, F.f_size, F.f_angle, F.f_resp, F.f_oct, F.f_desc,
                  F.f_count);
}


// ------------------------------------------------------------------------------------------------
static void resize_axis(int src_n, int dst_n, int32_t *ofs, int32_t *coef)
{
    const double scale = (double)src_n / (double)dst_n;
    for (int d = 0; d < dst_n; ++d) {
        double f = ((double)d + 0.5) * scale - 0.5;
        int s = (int)floor(f);
        double a = f - (double)s;
        if (s < 0) { s = 0; a = 0.0; }
        if (s >= src_n - 1) { s = src_n - 1; a = 0.0; }
        ofs[d] = s;
        coef[d] = (int32_t)lrint(a * (double)(1 << RELOC_RESIZE_COEF_BITS));
    }
}

static bool g_pattern_uploaded[64] = {};

int orb_prepare(reloc_ctx *ctx, int w, int h, int nfeatures)
{
    if (w > ctx->max_w || h > ctx->max_h) {
        reloc_set_error("frame %dx%d exceeds the ctx capacity %dx%d", w, h, ctx->max_w, ctx->max_h);
        return RELOC_E_CAPACITY;
    }
    if (ctx->orb_w == w && ctx->orb_h == h && ctx->orb_nfeat == nfeatures) return RELOC_OK;
    if (!g_pattern_uploaded[ctx->device & 63]) {
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(c_pattern), RELOC_ORB_PATTERN, sizeof(RELOC_ORB_PATTERN)));
        g_pattern_uploaded[ctx->device & 63] = true;
    }
    OrbTable tab;
    memset(&tab, 0, sizeof(tab));
    int64_t off = 0;
    for (int l = 0; l < NLEV; ++l) {
        const float s = (float)pow(RELOC_ORB_SCALE_FACTOR, (double)l);
        OrbLevel &L = tab.lev[l];
        L.scale = s;
        L.w = (int)lrintf((float)w / s);
        L.h = (int)lrintf((float)h / s);
        L.stride = (L.w + 63) / 64 * 64;
        L.off = off;
        off += ((int64_t)L.stride * L.h + 255) / 256 * 256;
    }
    if (off > ctx->pyr_bytes) { reloc_set_error("pyramid arena too small"); return RELOC_E_CAPACITY; }
    {
        const float factor = (float)(1.0 / RELOC_ORB_SCALE_FACTOR);
        float nper = (float)(nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)NLEV)));
        int sum = 0;
        for (int l = 0; l < NLEV - 1; ++l) {
            tab.lev[l].quota = (int)lrintf(nper);
            sum += tab.lev[l].quota;
            nper *= factor;
        }
        tab.lev[NLEV - 1].quota = nfeatures - sum > 0 ? nfeatures - sum : 0;
    }
    for (int l = 0; l < NLEV; ++l) {
        const OrbLevel &L = tab.lev[l];
        tab.fast_tile_base[l + 1] = tab.fast_tile_base[l] + (L.stride / FT) * ((L.h + FT - 1) / FT);
        tab.blur_tile_base[l + 1] = tab.blur_tile_base[l] + ((L.w + BT_W - 1) / BT_W) * ((L.h + BT_H - 1) / BT_H);
        tab.flat_base[l + 1] = tab.flat_base[l] + (int)(((int64_t)L.stride * L.h + HARRIS_CHUNK - 1) / HARRIS_CHUNK);
    }
    // resize tables
    const int maxdim = ctx->max_w > ctx->max_h ? ctx->max_w : ctx->max_h;
    int32_t *host = (int32_t *)malloc(sizeof(int32_t) * (size_t)NLEV * 4 * maxdim);
    int pos = 0;
    for (int l = 1; l < NLEV; ++l) {
        const OrbLevel &S = tab.lev[l - 1], &D = tab.lev[l];
        tab.rz_off[l][0] = pos; tab.rz_off[l][1] = pos + D.w;
        resize_axis(S.w, D.w, host + pos, host + pos + D.w);
        pos += 2 * D.w;
        tab.rz_off[l][2] = pos; tab.rz_off[l][3] = pos + D.h;
        resize_axis(S.h, D.h, host + pos, host + pos + D.h);
        pos += 2 * D.h;
    }
    // fused-pyramid tiles: every tile owns a rectangle of every level (proportional split, x on 4-pixel
    // boundaries, the last column of tiles takes the row padding of levels >= 1, which is stored as 0)
    // and computes what the levels above need from it (k_pyramid).
    const int ntx = (w + PT_W - 1) / PT_W, nty = (h + PT_H - 1) / PT_H;
    PyrTile *tiles = (PyrTile *)calloc((size_t)ntx * nty, sizeof(PyrTile));
    int lds_lev[NLEV] = {}, lds_t = 0;
    for (int t = 0; t < ntx * nty; ++t) {
        const int tx = t % ntx, ty = t / ntx;
        PyrTile &T = tiles[t];
        int nx0 = 0, nx1 = 0, ny0 = 0, ny1 = 0;    // needed rectangle of the level above (empty)
        int tsum = 0;
        for (int l = NLEV - 1; l >= 0; --l) {
            const OrbLevel &L = tab.lev[l];
            const int quads = (l == 0 ? (L.w + 3) / 4 : L.stride / 4);
            const int ox0 = 4 * (int)((int64_t)tx * quads / ntx), ox1 = 4 * (int)((int64_t)(tx + 1) * quads / ntx);
            const int oy0 = (int)((int64_t)ty * L.h / nty), oy1 = (int)((int64_t)(ty + 1) * L.h / nty);
            T.o[l][0] = (uint16_t)ox0; T.o[l][1] = (uint16_t)ox1; T.o[l][2] = (uint16_t)oy0; T.o[l][3] = (uint16_t)oy1;
            // computed rectangle = own pixels (inside the image) united with the taps of the level above
            int cx0 = ox0, cx1 = ox1 < L.w ? ox1 : L.w, cy0 = oy0, cy1 = oy1;
            const bool stores = ox0 < ox1 && oy0 < oy1;                     // may be row padding only
            const bool own = cx0 < cx1 && cy0 < cy1, need = nx0 < nx1 && ny0 < ny1;
            if (need) {
                const int32_t *xo = host + tab.rz_off[l + 1][0], *yo = host + tab.rz_off[l + 1][2];
                int sx0 = xo[nx0], sx1 = xo[nx1 - 1] + 2, sy0 = yo[ny0], sy1 = yo[ny1 - 1] + 2;
                if (sx1 > L.w) sx1 = L.w;
                if (sy1 > L.h) sy1 = L.h;
                if (own) {
                    cx0 = cx0 < sx0 ? cx0 : sx0; cx1 = cx1 > sx1 ? cx1 : sx1;
                    cy0 = cy0 < sy0 ? cy0 : sy0; cy1 = cy1 > sy1 ? cy1 : sy1;
                } else {
                    cx0 = sx0; cx1 = sx1; cy0 = sy0; cy1 = sy1;
                }
            } else if (!own) {
                cx0 = cx1 = cy0 = cy1 = 0;
            }
            cx0 &= ~3;
            T.n[l][0] = (uint16_t)cx0; T.n[l][1] = (uint16_t)cx1; T.n[l][2] = (uint16_t)cy0; T.n[l][3] = (uint16_t)cy1;
            if (!stores) T.o[l][0] = T.o[l][1] = T.o[l][2] = T.o[l][3] = 0;
            nx0 = cx0; nx1 = cx1; ny0 = cy0; ny1 = cy1;
            const int bytes = ((cx1 - cx0 + 3) / 4 * 4) * (cy1 - cy0);
            lds_lev[l] = bytes > lds_lev[l] ? bytes : lds_lev[l];
            if (l >= 1) tsum += (cx1 - cx0) + (cy1 - cy0);
        }
        lds_t = tsum > lds_t ? tsum : lds_t;
    }
    ctx->pyr_ntiles = ntx * nty;
    {
        int o = 0;
        for (int l = 0; l < NLEV; ++l) { ctx->pyr_lds[l] = o; o += (lds_lev[l] + 15) / 16 * 16; }
        ctx->pyr_lds[NLEV] = o;
        ctx->pyr_lds_bytes = o + 4 * lds_t;
    }
    if (ctx->pyr_lds_bytes > 64 * 1024) { free(host); free(tiles); reloc_set_error("pyramid tile exceeds LDS"); return RELOC_E_CAPACITY; }
    hipError_t e0 = hipMemcpyAsync(ctx->pyr_tiles, tiles, sizeof(PyrTile) * (size_t)ntx * nty, hipMemcpyHostToDevice, ctx->stream);
    hipError_t e1 = hipMemcpyAsync(ctx->rz_tab, host, sizeof(int32_t) * (size_t)pos, hipMemcpyHostToDevice, ctx->stream);
    hipError_t e2 = hipMemcpyAsync(ctx->orb_const, &tab, sizeof(tab), hipMemcpyHostToDevice, ctx->stream);
    hipError_t e3 = hipStreamSynchronize(ctx->stream);
    free(host);
    free(tiles);
    HIP_TRY(e0); HIP_TRY(e1); HIP_TRY(e2); HIP_TRY(e3);
    memcpy(ctx->lev, tab.lev, sizeof(tab.lev));
    memcpy(ctx->orb_tab_host, &tab, sizeof(tab));
    ctx->orb_w = w; ctx->orb_h = h; ctx->orb_nfeat = nfeatures;
    return RELOC_OK;
}

// src_dev: channels == 3 -> interleaved frame (gray fused), channels == 1 -> gray plane.
int orb_run_dev(reloc_ctx *ctx, const uint8_t *src_dev, int w, int h, int stride, int channels, int order, int nfeatures)
{
    int rc = orb_prepare(ctx, w, h, nfeatures);
    if (rc) return rc;
    const OrbTable *tab_h = (const OrbTable *)ctx->orb_tab_host;
    const OrbTable *tab_d = (const OrbTable *)ctx->orb_const;
    hipStream_t st = ctx->stream;
    reloc_prof_begin(ctx, RELOC_PROF_ORB);
    {
        const bool aligned = (w % 4 == 0) && (stride % 4 == 0) && (((uintptr_t)src_dev) % 4 == 0);
        // 512-thread workgroups where nothing scans beside the tick (local-candidate ticks, exclusive contexts, single calls),
        // 256 in ticks that share the chip with whole-database scans
        const bool wide = ctx->orb_latency_shape;
        auto kern512 = channels == 3 ? (aligned ? k_pyramid<3, true, 512> : k_pyramid<3, false, 512>) : (aligned ? k_pyramid<1, true, 512> : k_pyramid<1, false, 512>);
        auto kern256 = channels == 3 ? (aligned ? k_pyramid<3, true, 256> : k_pyramid<3, false, 256>) : (aligned ? k_pyramid<1, true, 256> : k_pyramid<1, false, 256>);
        PyrLds lds;
        for (int l = 0; l < NLEV; ++l) lds.lev[l] = ctx->pyr_lds[l];
        lds.tabs = ctx->pyr_lds[NLEV];
        hipLaunchKernelGGL(wide ? kern512 : kern256, dim3(ctx->pyr_ntiles), dim3(wide ? 512 : 256), ctx->pyr_lds_bytes, st, tab_d,
                           (const PyrTile *)ctx->pyr_tiles, ctx->rz_tab, src_dev, w, h, stride, gray_flags(ctx, order), ctx->pyr, lds, ctx->hist, ctx->cand_cnt);
    }
    hipLaunchKernelGGL(k_fast_blur, dim3(tab_h->fast_tile_base[NLEV] + tab_h->blur_tile_base[NLEV]), dim3(256), 0, st, tab_d, ctx->pyr,
                       ctx->nms, ctx->hist, ctx->blur, tab_h->fast_tile_base[NLEV]);
    hipLaunchKernelGGL(k_harris, dim3(tab_h->flat_base[NLEV]), dim3(256), 0, st, tab_d, ctx->pyr, ctx->nms, ctx->hist,
                       ctx->cand_cnt, ctx->cand_key, ctx->cand_resp, ctx->dbg_cut);
    hipLaunchKernelGGL(k_select, dim3(NLEV), dim3(1024), 0, st, tab_d, ctx->cand_cnt,
                       ctx->cand_key, ctx->cand_resp, ctx->kp_cnt, ctx->kp_key, ctx->kp_resp);
    hipLaunchKernelGGL(k_describe, dim3((ctx->max_feat + 3) / 4), dim3(256), 0, st, tab_d, ctx->pyr, ctx->blur, ctx->kp_cnt,
                       ctx->kp_key, ctx->kp_resp, ctx->max_feat, ctx->f_xy, ctx->f_size, ctx->f_angle, ctx->f_resp,
                       ctx->f_oct, ctx->f_desc, ctx->f_count);
    reloc_prof_end(ctx, RELOC_PROF_ORB);
    HIP_TRY(hipGetLastError());
    return RELOC_OK;
}

// The same five kernels for n contexts (one frame each, equal geometry) that share a stream: five launches, blockIdx.y = frame.
int orb_run_batch_dev(reloc_ctx *const *ctxs, int n, const uint8_t *const *srcs_dev, int w, int h, int stride, int order, int nfeatures)
{
    if (n < 1 || n > RELOC_BATCH_MAX) { reloc_set_error("orb batch: 1..%d frames", RELOC_BATCH_MAX); return RELOC_E_ARG; }
    OrbBatch b;
    bool aligned = (w % 4 == 0) && (stride % 4 == 0);
    for (int f = 0; f < RELOC_BATCH_MAX; ++f) {
        reloc_ctx *c = ctxs[f < n ? f : 0];
        if (f < n) {
            const int rc = orb_prepare(c, w, h, nfeatures);
            if (rc) return rc;
            if (c->pyr_ntiles != ctxs[0]->pyr_ntiles || c->pyr_lds_bytes != ctxs[0]->pyr_lds_bytes || c->max_feat != ctxs[0]->max_feat ||
                c->prm.gray_coeff_bits != ctxs[0]->prm.gray_coeff_bits) {
                reloc_set_error("orb batch: contexts of unequal geometry");
                return RELOC_E_STATE;
            }
            aligned = aligned && (((uintptr_t)srcs_dev[f]) % 4 == 0);
        }
        OrbFrame &F = b.f[f];
        F.tab = (const OrbTable *)c->orb_const; F.tiles = (const PyrTile *)c->pyr_tiles; F.rz = c->rz_tab; F.src = srcs_dev[f < n ? f : 0];
        F.pyr = c->pyr; F.nms = c->nms; F.blur = c->blur; F.hist = c->hist; F.cand_cnt = c->cand_cnt; F.cand_key = c->cand_key;
        F.cand_resp = c->cand_resp; F.dbg_cut = c->dbg_cut; F.kp_cnt = c->kp_cnt; F.kp_key = c->kp_key; F.kp_resp = c->kp_resp;
        F.f_xy = c->f_xy; F.f_size = c->f_size; F.f_angle = c->f_angle; F.f_resp = c->f_resp; F.f_oct = c->f_oct; F.f_desc = c->f_desc;
        F.f_count = c->f_count;
    }
    reloc_ctx *c0 = ctxs[0];
    const OrbTable *tab_h = (const OrbTable *)c0->orb_tab_host;
    hipStream_t st = c0->stream;
    reloc_prof_begin(c0, RELOC_PROF_ORB);
    PyrLds lds;
    for (int l = 0; l < NLEV; ++l) lds.lev[l] = c0->pyr_lds[l];
    lds.tabs = c0->pyr_lds[NLEV];
    // 256-thread pyramid: a batch runs beside other streams' scans (see orb_run_dev)
    if (aligned)
        hipLaunchKernelGGL((k_pyramid_batch<3, true, 256>), dim3(c0->pyr_ntiles, n), dim3(256), c0->pyr_lds_bytes, st, b, w, h, stride, gray_flags(c0, order), lds);
    else
        hipLaunchKernelGGL((k_pyramid_batch<3, false, 256>), dim3(c0->pyr_ntiles, n), dim3(256), c0->pyr_lds_bytes, st, b, w, h, stride, gray_flags(c0, order), lds);
    hipLaunchKernelGGL(k_fast_blur_batch, dim3(tab_h->fast_tile_base[NLEV] + tab_h->blur_tile_base[NLEV], n), dim3(256), 0, st, b,
                       tab_h->fast_tile_base[NLEV]);
    hipLaunchKernelGGL(k_harris_batch, dim3(tab_h->flat_base[NLEV], n), dim3(256), 0, st, b);
    hipLaunchKernelGGL(k_select_batch, dim3(NLEV, n), dim3(1024), 0, st, b);
    hipLaunchKernelGGL(k_describe_batch, dim3((c0->max_feat + 3) / 4, n), dim3(256), 0, st, b, c0->max_feat);
    reloc_prof_end(c0, RELOC_PROF_ORB);
    HIP_TRY(hipGetLastError());
    return RELOC_OK;
}


// ------------------------------------------------------------------------------------------------
RELOC_API int reloc_gray_u8(reloc_ctx *ctx, const uint8_t *img, int w, int h, int stride, int order, uint8_t *gray)
{
    ARG_CHECK_CTX(ctx, img && gray && w > 0 && h > 0 && stride >= 3 * w, "reloc_gray_u8");
    if (w > ctx->max_w || h > ctx->max_h) { reloc_set_error("frame exceeds ctx capacity"); return RELOC_E_CAPACITY; }
    void *dout;
    int rc;
    if ((rc = reloc_scratch(ctx, 0, (int64_t)w * h, &dout))) return rc;
    HIP_TRY(hipMemcpy2DAsync(ctx->frame_img, (size_t)w * 3, img, stride, (size_t)w * 3, h, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_gray_plain, dim3((w + 255) / 256, h), dim3(256), 0, ctx->stream, ctx->frame_img, w, h, w * 3, gray_flags(ctx, order),
                       (uint8_t *)dout);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(gray, dout, (size_t)w * h, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RELOC_OK;
}

RELOC_API int reloc_orb_frame_dev(reloc_ctx *ctx, const uint8_t *img_dev, int w, int h, int stride, int order, int nfeatures)
{
    ARG_CHECK_CTX(ctx, img_dev && w >= 64 && h >= 64 && stride >= 3 * w && nfeatures > 0, "reloc_orb_frame_dev");
    return orb_run_dev(ctx, img_dev, w, h, stride, 3, order, nfeatures);
}

RELOC_API const uint8_t *reloc_frame_desc_dev(reloc_ctx *ctx) { return ctx ? ctx->f_desc : nullptr; }
RELOC_API const float *reloc_frame_xy_dev(reloc_ctx *ctx) { return ctx ? ctx->f_xy : nullptr; }
RELOC_API const int32_t *reloc_frame_count_dev(reloc_ctx *ctx) { return ctx ? ctx->f_count : nullptr; }

RELOC_API int reloc_orb_detect_compute(reloc_ctx *ctx, const uint8_t *gray, int w, int h, int stride, int nfeatures,
                                       float *xy, float *size, float *angle, float *response, int32_t *octave,
                                       uint8_t *desc, int32_t *n_out)
{
    ARG_CHECK_CTX(ctx, gray && n_out && w > 0 && h > 0 && stride >= w && nfeatures > 0, "reloc_orb_detect_compute");
    *n_out = 0;
    if (w < 63 || h < 63) return RELOC_OK;   // no level is wider than the 31-pixel edge margin on both sides
    if (w > ctx->max_w || h > ctx->max_h) { reloc_set_error("frame exceeds ctx capacity"); return RELOC_E_CAPACITY; }
    HIP_TRY(hipMemcpy2DAsync(ctx->frame_img, w, gray, stride, w, h, hipMemcpyHostToDevice, ctx->stream));
    int rc = orb_run_dev(ctx, ctx->frame_img, w, h, w, 1, 0, nfeatures);
    if (rc) return rc;
    int32_t n = 0;
    HIP_TRY(hipMemcpyAsync(&n, ctx->f_count, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (n > 0) {
        if (xy) HIP_TRY(hipMemcpyAsync(xy, ctx->f_xy, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
        if (size) HIP_TRY(hipMemcpyAsync(size, ctx->f_size, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (angle) HIP_TRY(hipMemcpyAsync(angle, ctx->f_angle, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (response) HIP_TRY(hipMemcpyAsync(response, ctx->f_resp, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (octave) HIP_TRY(hipMemcpyAsync(octave, ctx->f_oct, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (desc) HIP_TRY(hipMemcpyAsync(desc, ctx->f_desc, (size_t)n * 32, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    *n_out = n;
    return RELOC_OK;
}

RELOC_API int reloc_frame_debug_plane(reloc_ctx *ctx, int what, int level, uint8_t *out, int32_t *w, int32_t *h)
{
    ARG_CHECK_CTX(ctx, out && w && h && what >= 0 && what <= 2 && level >= 0 && level < NLEV, "reloc_frame_debug_plane");
    if (!ctx->orb_w) { reloc_set_error("no frame processed yet"); return RELOC_E_STATE; }
    const OrbLevel &L = ctx->lev[level];
    const uint8_t *src = (what == 0 ? ctx->pyr : what == 1 ? ctx->blur : ctx->nms) + L.off;
    HIP_TRY(hipMemcpy2DAsync(out, L.w, src, L.stride, L.w, L.h, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *w = L.w;
    *h = L.h;
    return RELOC_OK;
}
