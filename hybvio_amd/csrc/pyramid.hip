// Image pyramid for the LK tracker: 5x5 binomial down-sample and Scharr-gradient stencils.
//
// Replaces cv::buildOpticalFlowPyramid as called by CpuImagePyramidFactory::compute
// (reference: src/tracker/image_pyramid.cpp:40-48). Three kernels (r02 layout):
//   * pyr_down_l0_kernel  levels without a stored gradient plane (0 and 1 by default: the LK kernel forms those gradients itself):
//                          a pure 5x5 down-sample straight from 16-byte row loads, no LDS, no barrier;
//   * pyr_tail_kernel     levels >= 2 of one image in ONE workgroup, LDS resident: gradients, the physical 32-pixel border of the
//                          padded levels, the next gray level (used from 64 images up);
//   * pyr_level_kernel    the general tile kernel (128x32 source tile + halo staged in LDS, Scharr gradients as int16 dx|dy per pixel
//                          + the next level's gray tile): stored-gradient levels of the experiment layouts, few images, unaligned
//                          caller buffers, and the on-demand gradient read-back of hv_pyramid_download.
// A source level is read from HBM once per launch. HBM traffic equals each launch's OWN bytes (PMC: 1.00 - 1.03x); it is NOT the
// "algorithmic bytes" of SURVEY.md 8(d) any more -- those count gradient planes of levels 0 and 1 that are never written (bench.py
// reports the stage both ways). Integer arithmetic throughout: results are bit-identical to the OpenCV algorithm restated in
// oracle/pyrlk_oracle.c.
#include "hv_internal.hpp"

namespace hv {

namespace {

constexpr int TW = 128;            // tile width  (source pixels)
constexpr int TH = 32;             // tile height (source pixels); TW * TH = 4096 = 256 threads x 4 x 4 pixels.
                                   // measured alternatives (L0, 752x480, B = 256; this shape: 0.272 ms): 256 x 16 tiles 0.309 ms,
                                   // 256 x 32 tiles with 512 threads 0.308 ms (fewer interior tiles), persistent workgroups
                                   // with a grid-stride tile loop 0.355 ms, every tile on the per-dword border path 0.284 ms
constexpr int CG = TW / 4, OCG = TW / 8;   // lanes per gradient row group / per down-sampled row
constexpr int LWD = (TW + 8) / 4;  // LDS row in dwords: source columns -4 .. TW+3
constexpr int LH = TH + 4;         // LDS rows: source rows -2 .. TH+1

struct PyrLevelArgs {
    const uint8_t *src_base;   // level gray image of batch element / slot 0
    long long src_step;        // bytes between consecutive images (or slots)
    int src_stride;            // bytes per source row
    int src_by_slot;           // index src by slot (internal levels) or by batch position (external L0)
    const int *slots;          // [n_images] pyramid slot of every batch element
    uint8_t *slab;
    long long slot_bytes;
    long long doff;            // gradient level offset inside the slot
    uint32_t *grad_out;        // non-null: gradients go to this plain buffer (read-back of a level that is not stored)
    int write_grad;            // 0: skip the gradient half (level 0 by default: the LK kernel forms them itself)
    int dstride;               // dwords per gradient row
    long long goff_next;       // next gray level offset inside the slot
    int gstride_next;
    int w, h, wn, hn;
    int tiles_x, tiles_y;
    const uint8_t **l0_ptr;    // level 0 only: per-slot pointer table filled here
    int *l0_stride;
};

// ---- packed 16-bit helpers: every intermediate of both stencils fits 16 bits, so two pixels share
// one VALU instruction (v_pk_add_u16 / v_pk_mad_u16 / v_pk_lshrrev_b16); bytes are widened to
// 16-bit pairs with one v_perm_b32 each. Results are bit-identical to the scalar formulation. ----
typedef unsigned short us2 __attribute__((ext_vector_type(2)));

// bytes I and J of the 8-byte string {lo, hi} as the 16-bit pair (b_I, b_J)
template <int I, int J>
__device__ __forceinline__ us2 bytes2(uint32_t lo, uint32_t hi)
{
    constexpr uint32_t sel = 0x0C000C00u | (uint32_t)I | ((uint32_t)J << 16);
    return __builtin_bit_cast(us2, __builtin_amdgcn_perm(hi, lo, sel));
}
__device__ __forceinline__ us2 splat(unsigned v) { return us2{(unsigned short)v, (unsigned short)v}; }
// (a.hi, b.lo)
__device__ __forceinline__ us2 cross(us2 a, us2 b)
{
    return __builtin_bit_cast(us2, __builtin_amdgcn_alignbit(__builtin_bit_cast(uint32_t, b), __builtin_bit_cast(uint32_t, a), 16));
}
__device__ __forceinline__ uint32_t lo_pair(us2 a, us2 b) { return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, b), __builtin_bit_cast(uint32_t, a), 0x05040100u); }
__device__ __forceinline__ uint32_t hi_pair(us2 a, us2 b) { return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, b), __builtin_bit_cast(uint32_t, a), 0x07060302u); }

template <bool DOWN>
__global__ __launch_bounds__(256) void pyr_level_kernel(PyrLevelArgs a)
{
    __shared__ uint32_t tile[LH * LWD];

    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int img = lb / tiles_per_img;
    const int t_in = lb - img * tiles_per_img;
    const int ty = t_in / a.tiles_x, tx = t_in - ty * a.tiles_x;
    const int slot = a.slots[img];
    const uint8_t *src = a.src_base + (long long)(a.src_by_slot ? slot : img) * a.src_step;
    const int t = threadIdx.x;
    const int x0 = tx * TW, y0 = ty * TH;

    if (a.l0_ptr != nullptr && t_in == 0 && t == 0) {
        a.l0_ptr[slot] = src;
        a.l0_stride[slot] = a.src_stride;
    }

    // ---- stage the source tile + halo in LDS (coalesced 136-byte rows) ----
    // BORDER_REFLECT_101 rows only change the row address; only the (at most two) dwords per row
    // that straddle the left / right image edge need per-byte reflection, so the decision is made
    // per dword, not per tile (42 % of the level-0 tiles touch an edge).
    const bool aligned = ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)a.src_stride) & 3u) == 0;
    if (aligned && x0 >= 4 && x0 + TW + 4 <= a.w && y0 >= 2 && y0 + TH + 2 <= a.h) {
        // interior tile (58 % of level 0): no border arithmetic at all
        int r = t / LWD, c = t - r * LWD;                    // 256 = 7 * 34 + 18: step the 2-D index
        const uint8_t *base = src + (long long)(y0 - 2) * a.src_stride + (x0 - 4);
        for (int i = t; i < LH * LWD; i += 256) {
            tile[i] = *reinterpret_cast<const uint32_t *>(base + (unsigned)r * (unsigned)a.src_stride + 4u * (unsigned)c);
            r += 256 / LWD; c += 256 % LWD;
            if (c >= LWD) { c -= LWD; r += 1; }
        }
    } else {
        int r = t / LWD, c = t - r * LWD;
        for (int i = t; i < LH * LWD; i += 256) {
            const uint8_t *row = src + (long long)reflect101(y0 - 2 + r, a.h) * a.src_stride;
            const int x = x0 - 4 + 4 * c;
            uint32_t v;
            if (aligned && x >= 0 && x + 4 <= a.w) {
                v = *reinterpret_cast<const uint32_t *>(row + x);
            } else {
                v = (uint32_t)row[reflect101(x, a.w)] | ((uint32_t)row[reflect101(x + 1, a.w)] << 8) |
                    ((uint32_t)row[reflect101(x + 2, a.w)] << 16) | ((uint32_t)row[reflect101(x + 3, a.w)] << 24);
            }
            tile[i] = v;
            r += 256 / LWD; c += 256 % LWD;
            if (c >= LWD) { c -= LWD; r += 1; }
        }
    }
    __syncthreads();

    uint8_t *slot_base = a.slab + (long long)slot * a.slot_bytes;

    // ---- Scharr gradients: lane = 4 consecutive pixels x 4 consecutive rows ----
    if (a.write_grad) {
        uint32_t *dbase = a.grad_out ? a.grad_out : reinterpret_cast<uint32_t *>(slot_base + a.doff);
        const int cg = t % CG, rg = t / CG;
        const int x = x0 + 4 * cg, yb = y0 + 4 * rg;
        if (x < a.w && yb < a.h) {
            // source rows yb-1 .. yb+4 = LDS rows 4rg+1 .. 4rg+6; columns x-1 .. x+4 are bytes 3 .. 8 of the
            // 12 staged bytes, widened to the pairs (x-1,x) (x+1,x+2) (x+3,x+4)
            us2 R[6][3];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const uint32_t *p = tile + (4 * rg + 1 + k) * LWD + cg;
                const uint32_t a0 = p[0], a1 = p[1], a2 = p[2];
                R[k][0] = bytes2<3, 4>(a0, a1);
                R[k][1] = bytes2<1, 2>(a1, a1);
                R[k][2] = bytes2<3, 4>(a1, a2);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int y = yb + j;
                us2 T0[3], T1[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    T0[q] = (R[j][q] + R[j + 2][q]) * splat(3) + R[j + 1][q] * splat(10);   // 3*(top+bot) + 10*mid <= 4080
                    T1[q] = R[j + 2][q] - R[j][q];                                            // bot - top (int16 wrap)
                }
                // dx = t0[x+1] - t0[x-1]; dy = 3*(t1[x-1] + t1[x+1]) + 10*t1[x]; stored as 4*d + 2 (klt.hip)
                const us2 dxA = (T0[1] - T0[0]) * splat(4) + splat(2);
                const us2 dxB = (T0[2] - T0[1]) * splat(4) + splat(2);
                const us2 dyA = (T1[0] + T1[1]) * splat(12) + (cross(T1[0], T1[1]) * splat(40) + splat(2));
                const us2 dyB = (T1[1] + T1[2]) * splat(12) + (cross(T1[1], T1[2]) * splat(40) + splat(2));
                const uint32_t o0 = lo_pair(dxA, dyA), o1 = hi_pair(dxA, dyA), o2 = lo_pair(dxB, dyB), o3 = hi_pair(dxB, dyB);
                if (y < a.h) {
                    uint32_t *d = dbase + (long long)y * a.dstride + x;
                    if (x + 3 < a.w) {
                        // streamed once, consumed later by sparse LK windows: keep it out of the caches
                        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                        const u32x4 pk = {o0, o1, o2, o3};
                        __builtin_nontemporal_store(pk, reinterpret_cast<u32x4 *>(d));
                    } else {
                        const uint32_t o[4] = {o0, o1, o2, o3};
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (x + i < a.w) d[i] = o[i];
                    }
                }
            }
        }
    }

    // ---- next gray level: lane = 4 consecutive outputs of one output row ----
    if (DOWN) {
        const int ocg = t % OCG, orow = t / OCG;
        const int oy = (y0 >> 1) + orow, ox = (x0 >> 1) + 4 * ocg;
        if (oy < a.hn && ox < a.wn) {
            // source columns 2*ox-2 .. 2*ox+8 are bytes 2 .. 12 of the 16 staged bytes (c_k = byte 2+k);
            // horizontal [1 4 6 4 1] for outputs (0,1) and (2,3) as packed pairs, <= 4080
            us2 H01[5], H23[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const uint32_t *p = tile + (2 * orow + j) * LWD + 2 * ocg;
                const uint32_t q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
                const us2 e24 = bytes2<2, 4>(q0, q1), e35 = bytes2<3, 5>(q0, q1), e46 = bytes2<0, 2>(q1, q1), e57 = bytes2<1, 3>(q1, q1);
                const us2 e68 = bytes2<2, 4>(q1, q2), e79 = bytes2<3, 5>(q1, q2), e8a = bytes2<0, 2>(q2, q2), e9b = bytes2<1, 3>(q2, q2);
                const us2 eac = bytes2<2, 4>(q2, q3);
                H01[j] = (e24 + e68) + (e35 + e57) * splat(4) + e46 * splat(6);
                H23[j] = (e68 + eac) + (e79 + e9b) * splat(4) + e8a * splat(6);
            }
            // vertical [1 4 6 4 1] + 128, >> 8: sums <= 255*256 + 128 = 65408 still fit 16 bits
            const us2 V01 = ((H01[0] + H01[4]) + (H01[1] + H01[3]) * splat(4) + (H01[2] * splat(6) + splat(128))) >> splat(8);
            const us2 V23 = ((H23[0] + H23[4]) + (H23[1] + H23[3]) * splat(4) + (H23[2] * splat(6) + splat(128))) >> splat(8);
            const uint32_t packed = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, V23), __builtin_bit_cast(uint32_t, V01), 0x06040200u);
            uint8_t *g = slot_base + a.goff_next + (long long)oy * a.gstride_next + ox;
            if (ox + 3 < a.wn) {
                *reinterpret_cast<uint32_t *>(g) = packed;
            } else {
#pragma unroll
                for (int o = 0; o < 4; ++o)
                    if (ox + o < a.wn) g[o] = (uint8_t)(packed >> (8 * o));
            }
        }
    }
}


// ---- level 0 when its gradient plane is not stored: the launch is a pure 5x5 binomial down-sample (read L0 once, write the
// L1 gray level). No LDS, no barrier: a work item produces 4 columns x 2 rows of L1 from seven 16-byte row loads (source columns
// 8g-4 .. 8g+11 of rows 4p-2 .. 4p+4; vertically adjacent items share 3 of the 7 rows through L2), so the latency of the HBM
// stream is hidden by ~8 independent wavefronts per SIMD instead of by workgroups that load, synchronise and then compute.
// Items whose 16-byte loads would leave the row (the first and the last column group) come AFTER the interior items of an
// image in the index space: they gather their bytes with BORDER_REFLECT_101 columns, and wavefronts stay homogeneous.
// Rows reflect through the row index on both paths. Measured at 752x480, 2048 images: DESIGN.md 3.1.
struct DownL0Args {
    const uint8_t *src_base;
    long long src_step;
    int src_stride, src_by_slot;
    const int *slots;
    uint8_t *slab;
    long long slot_bytes, goff_next;
    int gstride_next;
    int w, h, wn, hn;
    int groups, g_lo, n_gi;    // 4-column output groups per row; interior groups g_lo .. g_lo + n_gi - 1
    int nrp;                   // output row pairs
    int edge_shift;            // w % 8 == 0: every group takes the vector-load path (g_lo = 0, n_gi = groups)
    int wgs_per_img;
    const uint8_t **l0_ptr;
    int *l0_stride;
};

// horizontal [1 4 6 4 1] at the 4 even columns 8g, 8g+2, 8g+4, 8g+6 from the 16 bytes of columns 8g-4 .. 8g+11
__device__ __forceinline__ void hpass(uint32_t q0, uint32_t q1, uint32_t q2, uint32_t q3, us2 &h01, us2 &h23)
{
    const us2 e24 = bytes2<2, 4>(q0, q1), e35 = bytes2<3, 5>(q0, q1), e46 = bytes2<0, 2>(q1, q1), e57 = bytes2<1, 3>(q1, q1);
    const us2 e68 = bytes2<2, 4>(q1, q2), e79 = bytes2<3, 5>(q1, q2), e8a = bytes2<0, 2>(q2, q2), e9b = bytes2<1, 3>(q2, q2);
    const us2 eac = bytes2<2, 4>(q2, q3);
    h01 = (e24 + e68) + (e35 + e57) * splat(4) + e46 * splat(6);
    h23 = (e68 + eac) + (e79 + e9b) * splat(4) + e8a * splat(6);
}

__global__ __launch_bounds__(256) void pyr_down_l0_kernel(DownL0Args a)
{
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int img = lb / a.wgs_per_img;
    const int item = (int)(lb - (unsigned)img * a.wgs_per_img) * 256 + (int)threadIdx.x;
    const int slot = a.slots[img];
    const uint8_t *src = a.src_base + (long long)(a.src_by_slot ? slot : img) * a.src_step;
    if (a.l0_ptr != nullptr && item == 0) {
        a.l0_ptr[slot] = src;
        a.l0_stride[slot] = a.src_stride;
    }
    const int n_int = a.n_gi * a.nrp;
    if (item >= a.groups * a.nrp) return;
    int g, rp;
    const bool interior = item < n_int;
    if (interior) {
        rp = item / a.n_gi; g = a.g_lo + (item - rp * a.n_gi);
    } else {
        const int nb = a.groups - a.n_gi, e = item - n_int;
        rp = e / nb;
        const int k = e - rp * nb;                       // border groups: 0 .. g_lo - 1, then g_lo + n_gi .. groups - 1
        g = k < a.g_lo ? k : k + a.n_gi;
    }
    const int oy = 2 * rp, ox = 4 * g;
    us2 H01[7], H23[7];
    if (interior) {
        // a.edge_shift (w % 8 == 0): the first and the last column group ride along -- their 16 bytes are loaded one dword to the
        // right / left of where they belong (inside the row) and the two reflected columns they need are bytes of the same load:
        //   first group: columns -2, -1 = columns 2, 1          last group: column w = column w - 2
        const int sh = a.edge_shift ? (g == 0 ? 1 : g == a.groups - 1 ? -1 : 0) : 0;
        uint4 q[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const uint8_t *row = src + (long long)reflect101(2 * oy - 2 + j, a.h) * a.src_stride + (8 * g - 4 + 4 * sh);
            __builtin_memcpy(&q[j], row, 16);
        }
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            uint32_t q0 = q[j].x, q1 = q[j].y, q2 = q[j].z, q3 = q[j].w;
            if (a.edge_shift) {
                const uint32_t refl = __builtin_amdgcn_perm(q[j].y, q[j].x, 0x01020304u);      // (c4, c3, c2, c1) of a load at column 0
                q0 = sh > 0 ? refl : sh < 0 ? q[j].y : q[j].x;
                q1 = sh > 0 ? q[j].x : sh < 0 ? q[j].z : q[j].y;
                q2 = sh > 0 ? q[j].y : sh < 0 ? q[j].w : q[j].z;
                q3 = sh > 0 ? q[j].z : sh < 0 ? (q[j].w >> 16) : q[j].w;
            }
            hpass(q0, q1, q2, q3, H01[j], H23[j]);
        }
    } else {
        // rare (first / last column group of an image whose width is not a multiple of 8): bytes gathered with reflected columns,
        // one row at a time (kept rolled: the unrolled form raised the kernel to 125 VGPRs, half the occupancy of the main path)
#pragma unroll
        for (int j = 0; j < 7; ++j) { H01[j] = splat(0); H23[j] = splat(0); }
#pragma unroll 1
        for (int j = 0; j < 7; ++j) {
            const uint8_t *row = src + (long long)reflect101(2 * oy - 2 + j, a.h) * a.src_stride;
            uint32_t q[4] = {0, 0, 0, 0};
#pragma unroll 1
            for (int k = 2; k < 13; ++k)           // columns 8g-2 .. 8g+8: the bytes hpass reads
                q[k >> 2] |= (uint32_t)row[reflect101(8 * g - 4 + k, a.w)] << (8 * (k & 3));
            us2 h01, h23;
            hpass(q[0], q[1], q[2], q[3], h01, h23);
#pragma unroll
            for (int jj = 0; jj < 7; ++jj)
                if (jj == j) { H01[jj] = h01; H23[jj] = h23; }
        }
    }
    uint8_t *gbase = a.slab + (long long)slot * a.slot_bytes + a.goff_next;
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        if (oy + o >= a.hn) break;
        const int j0 = 2 * o;
        // vertical [1 4 6 4 1] + 128, >> 8: sums <= 255*256 + 128 = 65408 still fit 16 bits
        const us2 V01 = ((H01[j0] + H01[j0 + 4]) + (H01[j0 + 1] + H01[j0 + 3]) * splat(4) + (H01[j0 + 2] * splat(6) + splat(128))) >> splat(8);
        const us2 V23 = ((H23[j0] + H23[j0 + 4]) + (H23[j0 + 1] + H23[j0 + 3]) * splat(4) + (H23[j0 + 2] * splat(6) + splat(128))) >> splat(8);
        const uint32_t packed = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, V23), __builtin_bit_cast(uint32_t, V01), 0x06040200u);
        uint8_t *gp = gbase + (long long)(oy + o) * a.gstride_next + ox;
        if (ox + 3 < a.wn) {
            *reinterpret_cast<uint32_t *>(gp) = packed;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (ox + k < a.wn) gp[k] = (uint8_t)(packed >> (8 * k));
        }
    }
}

// ---- the coarse tail (levels >= 2) of one image in ONE workgroup: the level fits LDS whole (188x120 = 22 KB, 320x180 = 58 KB),
// so its gradients, its physical REFLECT_101 border, the next gray level and that level's gradients and border all come from
// LDS copies -- one launch instead of three (two level launches + pyr_border_kernel, 0.18 ms per 2048 images for 0.4 GB),
// and the only HBM read is the level-2 interior the level-1 launch just wrote. Same packed 16-bit arithmetic as
// pyr_level_kernel (the LDS image has that kernel's tile layout: columns -4 .. w+3, rows -2 .. h+1, halo by REFLECT_101).
struct TailArgs {
    PyrLayout L;
    uint8_t *slab;
    const int *slots;
    int first;                 // first level handled here
    int buf1;                  // dword offset of the second level buffer
};
#ifndef HV_TAIL_THREADS
#define HV_TAIL_THREADS 512
#endif
constexpr int TAIL_THREADS = HV_TAIL_THREADS;

// i / d for 0 <= i < 2^22, d > 0 given rcp = 1.f / d: the float quotient is off by at most one (a hardware integer division is ~30
// instructions and every work item of the tail kernel starts with one or two)
__device__ __forceinline__ int fast_div(int i, int d, float rcp)
{
    int q = (int)((float)i * rcp);
    if (q * d > i) --q;
    else if ((q + 1) * d <= i) ++q;
    return q;
}
__device__ __forceinline__ int tail_lwd(int w) { return (w + 8 + 3) >> 2; }
__device__ __forceinline__ int tail_rows(int h) { return 4 * ((h + 3) >> 2) + 4; }

// halo (and the ragged right end) of a level image in LDS from its interior
__device__ __forceinline__ void tail_fill_halo(uint32_t *img, int w, int h, int lwd, int t)
{
    const uint8_t *b = reinterpret_cast<const uint8_t *>(img);
    const float rcp = 1.f / (float)lwd;
    for (int i = t; i < (h + 4) * lwd; i += TAIL_THREADS) {
        const int r = fast_div(i, lwd, rcp), k = i - r * lwd;
        const int y = r - 2, x = 4 * k - 4;
        if (y >= 0 && y < h && x >= 0 && x + 4 <= w) continue;              // interior dword
        const int rb = (reflect101(y, h) + 2) * lwd * 4 + 4;
        uint32_t v = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) v |= (uint32_t)b[rb + reflect101(x + q, w)] << (8 * q);
        img[i] = v;
    }
}

__global__ __launch_bounds__(TAIL_THREADS) void pyr_tail_kernel(TailArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t tl[];
    const PyrLayout &L = a.L;
    const int t = threadIdx.x;
    uint8_t *slot_base = a.slab + (long long)a.slots[blockIdx.x] * L.slot_bytes;
    uint32_t *cur = tl, *nxt = tl + a.buf1;

    // ---- the first level from HBM: interior dwords by aligned loads, the rest by reflection ----
    {
        const int l = a.first, w = L.w[l], h = L.h[l], lwd = tail_lwd(w), gs = L.gstride[l];
        const uint8_t *g = slot_base + L.goff[l];
        const float rcp = 1.f / (float)lwd;
        // four loads in flight per thread before the first LDS store (a load -> store loop pays the HBM latency per iteration)
        constexpr int U = 4;
        for (int base = t; base < (h + 4) * lwd; base += U * TAIL_THREADS) {
            uint32_t v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = min(base + u * TAIL_THREADS, (h + 4) * lwd - 1);
                const int r = fast_div(i, lwd, rcp), k = i - r * lwd;
                const int x = 4 * k - 4;
                const uint8_t *row = g + (long long)reflect101(r - 2, h) * gs;
                if (x >= 0 && x + 4 <= w) {
                    v[u] = *reinterpret_cast<const uint32_t *>(row + x);
                } else {
                    v[u] = 0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[u] |= (uint32_t)row[reflect101(x + q, w)] << (8 * q);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = base + u * TAIL_THREADS;
                if (i < (h + 4) * lwd) cur[i] = v[u];
            }
        }
    }
    __syncthreads();

    for (int l = a.first; l < L.levels; ++l) {
        const int w = L.w[l], h = L.h[l], lwd = tail_lwd(w);
        const bool down = l + 1 < L.levels;
        const int wn = down ? L.w[l + 1] : 0, hn = down ? L.h[l + 1] : 0, lwdn = tail_lwd(wn);

        // ---- Scharr gradients: item = 4 pixels x 4 rows ----
        {
            uint32_t *dbase = reinterpret_cast<uint32_t *>(slot_base + L.doff[l]);
            const int ncg = (w + 3) >> 2, nrg = (h + 3) >> 2, ds = L.dstride[l];
            const float rcp = 1.f / (float)ncg;
            for (int it = t; it < ncg * nrg; it += TAIL_THREADS) {
                const int rg = fast_div(it, ncg, rcp), cg = it - rg * ncg;
                const int x = 4 * cg, yb = 4 * rg;
                us2 R[6][3];
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const uint32_t *p = cur + (yb + 1 + k) * lwd + cg;
                    const uint32_t a0 = p[0], a1 = p[1], a2 = p[2];
                    R[k][0] = bytes2<3, 4>(a0, a1);
                    R[k][1] = bytes2<1, 2>(a1, a1);
                    R[k][2] = bytes2<3, 4>(a1, a2);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int y = yb + j;
                    us2 T0[3], T1[3];
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        T0[q] = (R[j][q] + R[j + 2][q]) * splat(3) + R[j + 1][q] * splat(10);
                        T1[q] = R[j + 2][q] - R[j][q];
                    }
                    const us2 dxA = (T0[1] - T0[0]) * splat(4) + splat(2);
                    const us2 dxB = (T0[2] - T0[1]) * splat(4) + splat(2);
                    const us2 dyA = (T1[0] + T1[1]) * splat(12) + (cross(T1[0], T1[1]) * splat(40) + splat(2));
                    const us2 dyB = (T1[1] + T1[2]) * splat(12) + (cross(T1[1], T1[2]) * splat(40) + splat(2));
                    const uint32_t o[4] = {lo_pair(dxA, dyA), hi_pair(dxA, dyA), lo_pair(dxB, dyB), hi_pair(dxB, dyB)};
                    if (y < h) {
                        uint32_t *d = dbase + (long long)y * ds + x;
                        if (x + 3 < w) {
                            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                            const u32x4 pk = {o[0], o[1], o[2], o[3]};
                            __builtin_nontemporal_store(pk, reinterpret_cast<u32x4 *>(d));
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (x + i < w) d[i] = o[i];
                        }
                    }
                }
            }
        }

        // ---- physical REFLECT_101 border of this gray level (what pyr_border_kernel writes) ----
        if (L.pad[l]) {
            const int pd = L.pad[l], gs = L.gstride[l];
            uint8_t *img = slot_base + L.goff[l];
            const uint8_t *b = reinterpret_cast<const uint8_t *>(cur);
            const int gw = (w + 2 * pd + 3) >> 2, n_tb = 2 * pd * gw;
            const int wq = w & ~3;
            const int gr = (w + pd - wq + 3) >> 2, gl = pd >> 2, n_lr = h * (gl + gr);
            const float rcp_gw = 1.f / (float)gw, rcp_lr = 1.f / (float)(gl + gr);
            for (int item = t; item < n_tb + n_lr; item += TAIL_THREADS) {
                int x, y;
                if (item < n_tb) {
                    const int r = fast_div(item, gw, rcp_gw);
                    x = 4 * (item - r * gw) - pd;
                    y = r < pd ? r - pd : h + (r - pd);
                } else {
                    const int e = item - n_tb, r = fast_div(e, gl + gr, rcp_lr), g = e - r * (gl + gr);
                    y = r;
                    x = g < gl ? 4 * g - pd : wq + 4 * (g - gl);
                }
                const int rr = reflect101(y, h) + 2;
                uint32_t v;
                if (x >= 0 && x + 4 <= w) {                          // above / below the image, inside its columns: one aligned LDS dword
                    v = cur[rr * lwd + 1 + (x >> 2)];
                } else {
                    const int rb = rr * lwd * 4 + 4;
                    v = 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) v |= (uint32_t)b[rb + reflect101(x + i, w)] << (8 * i);
                }
                uint8_t *dst = img + (long long)y * gs + x;
                if (x + 4 <= w + pd) {
                    *reinterpret_cast<uint32_t *>(dst) = v;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (x + i < w + pd) dst[i] = (uint8_t)(v >> (8 * i));
                }
            }
        }

        // ---- next gray level: to HBM and into the other LDS buffer ----
        if (down) {
            const int nocg = (wn + 3) >> 2, gsn = L.gstride[l + 1];
            uint8_t *gn = slot_base + L.goff[l + 1];
            const float rcp = 1.f / (float)nocg;
            for (int it = t; it < nocg * hn; it += TAIL_THREADS) {
                const int oy = fast_div(it, nocg, rcp), ocg = it - oy * nocg;
                const int ox = 4 * ocg;
                us2 H01[5], H23[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const uint32_t *p = cur + (2 * oy + j) * lwd + 2 * ocg;
                    hpass(p[0], p[1], p[2], p[3], H01[j], H23[j]);
                }
                const us2 V01 = ((H01[0] + H01[4]) + (H01[1] + H01[3]) * splat(4) + (H01[2] * splat(6) + splat(128))) >> splat(8);
                const us2 V23 = ((H23[0] + H23[4]) + (H23[1] + H23[3]) * splat(4) + (H23[2] * splat(6) + splat(128))) >> splat(8);
                const uint32_t packed = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, V23), __builtin_bit_cast(uint32_t, V01), 0x06040200u);
                nxt[(oy + 2) * lwdn + 1 + ocg] = packed;            // bytes beyond wn are rewritten by the halo fill
                uint8_t *gp = gn + (long long)oy * gsn + ox;
                if (ox + 3 < wn) {
                    *reinterpret_cast<uint32_t *>(gp) = packed;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (ox + k < wn) gp[k] = (uint8_t)(packed >> (8 * k));
                }
            }
            __syncthreads();
            tail_fill_halo(nxt, wn, hn, lwdn, t);
            __syncthreads();
            uint32_t *sw = cur; cur = nxt; nxt = sw;
        }
    }
}

// ---- physical borders of the padded (coarse) levels, see PyrLayout ----
struct BorderArgs {
    PyrLayout L;
    uint8_t *slab;
    const int *slots;          // [n] (gray borders) or NULL: slot = first_slot + blockIdx.y
    int first_slot;
    int first_level;           // levels first_level .. L.levels-1 are padded
};

// BORDER_REFLECT_101 frame of every padded gray level of one image, from the interior the level kernels just wrote
// (what cv::buildOpticalFlowPyramid's copyMakeBorder does). One dword = 4 border pixels per thread step; dwords that lie
// entirely inside the image are skipped. ~66 KB per 752x480 image (levels 2 and 3).
__global__ __launch_bounds__(256) void pyr_border_kernel(BorderArgs a)
{
    const PyrLayout &L = a.L;
    uint8_t *slot = a.slab + (long long)a.slots[blockIdx.y] * L.slot_bytes;
    int item = blockIdx.x * 256 + threadIdx.x;
    for (int l = a.first_level; l < L.levels; ++l) {
        const int pd = L.pad[l], w = L.w[l], h = L.h[l], gs = L.gstride[l];
        // items of a level: the pd rows above and below the image at full padded width, then the left and right pd columns
        // of the image rows -- all in dwords (the padded width and pd are multiples of 4 up to the ragged right end)
        const int gw = (w + 2 * pd + 3) >> 2, n_tb = 2 * pd * gw;
        const int wq = w & ~3;                              // columns [wq, w + pd) form the right strip (dword aligned start)
        const int gr = (w + pd - wq + 3) >> 2, gl = pd >> 2, n_lr = h * (gl + gr);
        if (item < n_tb + n_lr) {
            uint8_t *img = slot + L.goff[l];
            int x, y;
            if (item < n_tb) {
                const int r = item / gw;
                x = 4 * (item - r * gw) - pd;
                y = r < pd ? r - pd : h + (r - pd);
            } else {
                const int e = item - n_tb, r = e / (gl + gr), g = e - r * (gl + gr);
                y = r;
                x = g < gl ? 4 * g - pd : wq + 4 * (g - gl);
            }
            const uint8_t *src = img + (long long)reflect101(y, h) * gs;
            uint8_t *dst = img + (long long)y * gs + x;
            if (x >= 0 && x + 4 <= w) {                      // above / below the image, inside its columns: a dword copy
                *reinterpret_cast<uint32_t *>(dst) = *reinterpret_cast<const uint32_t *>(src + x);
            } else {
                uint32_t v = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) v |= (uint32_t)src[reflect101(x + i, w)] << (8 * i);
                // the right strip of an image row may start inside the image (w % 4 != 0): those bytes are rewritten with
                // their own values (reflect101 of an inside column is the column)
                if (x + 4 <= w + pd) {
                    *reinterpret_cast<uint32_t *>(dst) = v;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (x + i < w + pd) dst[i] = (uint8_t)(v >> (8 * i));
                }
            }
            return;
        }
        item -= n_tb + n_lr;
    }
}

// Gradient planes of the padded levels, whole padded rectangle = "0" (stored 4*0+2 in both halves). The level kernels only
// ever write the interior, so this runs once per slot.
__global__ __launch_bounds__(256) void grad_border_fill_kernel(BorderArgs a)
{
    const PyrLayout &L = a.L;
    uint8_t *slot = a.slab + (long long)(a.first_slot + (int)blockIdx.y) * L.slot_bytes;
    for (int l = a.first_level; l < L.levels; ++l) {
        const int pd = L.pad[l];
        uint32_t *base = reinterpret_cast<uint32_t *>(slot + L.doff[l]) - ((long long)pd * L.dstride[l] + pd);
        const int n = L.dstride[l] * (L.h[l] + 2 * pd);
        for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) base[i] = 0x00020002u;
    }
}

int first_padded_level(const PyrLayout &L)
{
    for (int l = 0; l < L.levels; ++l) if (L.pad[l]) return l;
    return L.levels;
}

}  // namespace

int fill_gradient_borders(Ctx *c, int first_slot, int n_slots)
{
    const int fl = first_padded_level(c->L);
    if (fl >= c->L.levels || n_slots <= 0) return HV_OK;
    BorderArgs a{};
    a.L = c->L; a.slab = c->slab; a.slots = nullptr; a.first_slot = first_slot; a.first_level = fl;
    hipLaunchKernelGGL(grad_border_fill_kernel, dim3(16, (unsigned)n_slots), dim3(256), 0, c->stream, a);
    HV_HIP(c, hipGetLastError());
    return HV_OK;
}

// hv_pyramid_download of the gradients of a level whose plane is not stored: the same stencil, written to a scratch buffer
int download_unstored_gradient(Ctx *c, int slot, int lv, int16_t *grad)
{
    const PyrLayout &L = c->L;
    const uint8_t *src = c->slab + (long long)slot * L.slot_bytes + L.goff[lv];
    int stride = L.gstride[lv];
    if (lv == 0) {
        HV_HIP(c, hipMemcpy(&src, c->d_l0_ptr + slot, sizeof(void *), hipMemcpyDeviceToHost));
        HV_HIP(c, hipMemcpy(&stride, c->d_l0_stride + slot, sizeof(int), hipMemcpyDeviceToHost));
    }
    if (!src) return HV_ERR_INVALID;
    uint32_t *tmp = nullptr;
    int *d_slot = nullptr;
    if (hipMalloc(&tmp, (size_t)L.dstride[lv] * L.h[lv] * 4) != hipSuccess) return HV_ERR_NOMEM;
    if (hipMalloc(&d_slot, sizeof(int)) != hipSuccess) { (void)hipFree(tmp); return HV_ERR_NOMEM; }
    int rc = HV_OK;
    PyrLevelArgs a{};
    a.src_base = src; a.src_step = 0; a.src_stride = stride; a.src_by_slot = 0;
    a.slots = d_slot; a.slab = c->slab; a.slot_bytes = L.slot_bytes;
    a.grad_out = tmp; a.write_grad = 1; a.dstride = L.dstride[lv];
    a.w = L.w[lv]; a.h = L.h[lv];
    a.tiles_x = (a.w + TW - 1) / TW; a.tiles_y = (a.h + TH - 1) / TH;
    if (hipMemcpy(d_slot, &slot, sizeof(int), hipMemcpyHostToDevice) != hipSuccess) rc = HV_ERR_HIP;
    if (rc == HV_OK) {
        hipLaunchKernelGGL(pyr_level_kernel<false>, dim3((unsigned)(a.tiles_x * a.tiles_y)), dim3(256), 0, c->stream, a);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) rc = HV_ERR_HIP;
    }
    if (rc == HV_OK && hipMemcpy2D(grad, (size_t)L.w[lv] * 4, tmp, (size_t)L.dstride[lv] * 4, (size_t)L.w[lv] * 4, L.h[lv],
                                   hipMemcpyDeviceToHost) != hipSuccess) rc = HV_ERR_HIP;
    (void)hipFree(tmp); (void)hipFree(d_slot);
    return rc;
}

// levels >= 2 in one launch (pyr_tail_kernel) when they fit LDS and no finer level is padded. One workgroup walks a whole
// image through its phases (~60 us): a throughput design -- with few images (one sequence: 2) the per-level launches finish
// sooner (bench latency leg: 11 us per frame), so it is used from 64 images up. Knob pyr_tail = 0 / 1 forces it off / on.
static int tail_first_level(const Ctx *c, const PyrLayout &L, int n_images, size_t *shmem, int *buf1)
{
    const int force = c->knob.pyr_tail;                                // tests / experiments switch it through hv_debug_set_knob
    const int first = 2;
    if (force == 0 || (force < 0 && n_images < 64) || L.levels <= first || first_padded_level(L) < first) return L.levels;
    auto dwords = [&](int l) { return (size_t)((L.w[l] + 8 + 3) >> 2) * (4 * ((L.h[l] + 3) >> 2) + 4) + 8; };
    const size_t b0 = dwords(first), b1 = first + 1 < L.levels ? dwords(first + 1) : 0;
    if ((b0 + b1) * 4 > 150 * 1024) return L.levels;
    *shmem = (b0 + b1) * 4; *buf1 = (int)b0;
    return first;
}

int launch_pyramid_levels(Ctx *c, int n, const int *slots_dev, const uint8_t *src_base,
                          long long src_step, int src_stride, bool src_indexed_by_slot)
{
    const PyrLayout &L = c->L;
    size_t tail_shmem = 0;
    int tail_buf1 = 0;
    const int tail_first = tail_first_level(c, L, n, &tail_shmem, &tail_buf1);
    for (int l = 0; l < tail_first && l < L.levels; ++l) {
        PyrLevelArgs a{};
        if (l == 0) {
            a.src_base = src_base; a.src_step = src_step; a.src_stride = src_stride;
            a.src_by_slot = src_indexed_by_slot ? 1 : 0;
            a.l0_ptr = c->d_l0_ptr; a.l0_stride = c->d_l0_stride;
        } else {
            a.src_base = c->slab + L.goff[l]; a.src_step = L.slot_bytes; a.src_stride = L.gstride[l];
            a.src_by_slot = 1;
        }
        a.slots = slots_dev;
        a.slab = c->slab; a.slot_bytes = L.slot_bytes;
        a.doff = L.doff[l]; a.dstride = L.dstride[l];
        a.write_grad = l >= L.grad_from ? 1 : 0;
        a.w = L.w[l]; a.h = L.h[l];
        const bool down = l + 1 < L.levels;
        if (down) { a.goff_next = L.goff[l + 1]; a.gstride_next = L.gstride[l + 1]; a.wn = L.w[l + 1]; a.hn = L.h[l + 1]; }
        a.tiles_x = (a.w + TW - 1) / TW; a.tiles_y = (a.h + TH - 1) / TH;
        const unsigned grid = (unsigned)(a.tiles_x * a.tiles_y * n);
        ScopedKernelTime tm(c, l == 0 ? HV_K_PYR_L0 : HV_K_PYR_LN);
        // a level without a stored gradient plane is a pure down-sample: the direct kernel, when the 16-byte row loads are legal
        const bool no_direct = c->knob.pyr_l0_tiled != 0;
        if (down && !a.write_grad && !no_direct && a.w >= 24 &&
            ((reinterpret_cast<uintptr_t>(a.src_base) | (uintptr_t)a.src_stride | (uintptr_t)(a.src_step & 3)) & 3u) == 0) {
            DownL0Args d{};
            d.src_base = a.src_base; d.src_step = a.src_step; d.src_stride = a.src_stride; d.src_by_slot = a.src_by_slot;
            d.slots = slots_dev; d.slab = c->slab; d.slot_bytes = L.slot_bytes; d.goff_next = a.goff_next; d.gstride_next = a.gstride_next;
            d.w = a.w; d.h = a.h; d.wn = a.wn; d.hn = a.hn;
            d.groups = (a.wn + 3) / 4; d.g_lo = 1;
            d.n_gi = (a.w - 12) / 8;                       // interior: 8g - 4 >= 0 and 8g + 12 <= w  <=>  1 <= g <= (w - 12) / 8
            if (d.n_gi > d.groups - 1) d.n_gi = d.groups - 1;
            if (a.w % 8 == 0 && a.w >= 32) { d.edge_shift = 1; d.g_lo = 0; d.n_gi = d.groups; }
            d.nrp = (a.hn + 1) / 2;
            d.wgs_per_img = (d.groups * d.nrp + 255) / 256;
            d.l0_ptr = a.l0_ptr; d.l0_stride = a.l0_stride;
            hipLaunchKernelGGL(pyr_down_l0_kernel, dim3((unsigned)(d.wgs_per_img * n)), dim3(256), 0, c->stream, d);
            HV_HIP(c, hipGetLastError());
            continue;
        }
        if (down) hipLaunchKernelGGL(pyr_level_kernel<true>, dim3(grid), dim3(256), 0, c->stream, a);
        else      hipLaunchKernelGGL(pyr_level_kernel<false>, dim3(grid), dim3(256), 0, c->stream, a);
        HV_HIP(c, hipGetLastError());
    }
    if (tail_first < L.levels) {
        TailArgs a{};
        a.L = L; a.slab = c->slab; a.slots = slots_dev; a.first = tail_first; a.buf1 = tail_buf1;
        static bool attr_set_dev[64] = {};
        bool &attr_set = attr_set_dev[c->p.device & 63];
        if (!attr_set) {
            HV_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(pyr_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
            attr_set = true;
        }
        ScopedKernelTime tm(c, HV_K_PYR_LN);
        hipLaunchKernelGGL(pyr_tail_kernel, dim3((unsigned)n), dim3(TAIL_THREADS), tail_shmem, c->stream, a);
        HV_HIP(c, hipGetLastError());
        return HV_OK;
    }
    const int fl = first_padded_level(L);
    if (fl < L.levels) {
        BorderArgs a{};
        a.L = L; a.slab = c->slab; a.slots = slots_dev; a.first_level = fl;
        int items = 0;
        for (int l = fl; l < L.levels; ++l) {
            const int pd = L.pad[l], w = L.w[l], h = L.h[l];
            items += 2 * pd * ((w + 2 * pd + 3) >> 2) + h * ((pd >> 2) + ((w + pd - (w & ~3) + 3) >> 2));
        }
        ScopedKernelTime tm(c, HV_K_PYR_LN);
        hipLaunchKernelGGL(pyr_border_kernel, dim3((unsigned)((items + 255) / 256), (unsigned)n), dim3(256), 0, c->stream, a);
        HV_HIP(c, hipGetLastError());
    }
    return HV_OK;
}

}  // namespace hv
