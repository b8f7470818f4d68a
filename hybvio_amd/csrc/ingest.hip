// Image ingest: colour -> gray and the undistort / rectify remap, written straight
// into level 0 of the pyramid slots, followed by the pyramid level kernels.
//
// Reference: src/tracker/image.cpp:272-306 (buildPrivate: colorToGrayOp, then Undistorter::undistort),
// image.cpp:351-367 (gray coefficients), src/tracker/undistorter.cpp:71-110 (remap arithmetic).
//
// The reference evaluates both camera models per pixel per frame (double precision, Newton iterations).
// The cameras are constant for a session (the reference's own GPU branch says so: undistorter.cpp:113-121), so
// here the host evaluates them once (through the reference's Camera objects, hv_ingest_set_undistort_map) and the
// device keeps a table per camera: the integer tap position and the two float weights exactly as
// undistorter.cpp:93-94 forms them. The per-frame kernel is then a gather + 4 multiply-adds per pixel, HBM/L2
// bound: per pixel 12 B of table (shared by every sequence of a batch, so it lives in L2 / Infinity Cache),
// the source bytes once, one gray byte out.
#include "hv_internal.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace hv {
namespace {

constexpr unsigned MAP_INVALID = 0xFFFFFFFFu;

struct IngestArgs {
    const uint8_t *src;        // image i at src + i * image_step, rows src_stride bytes apart, CH bytes per pixel
    long long image_step;
    int src_stride;
    const int *slots;          // NULL: every image goes to slot0 (single-image host entry)
    int slot0;
    uint8_t *slab;             // destination: slab + slot * slot_bytes + goff0, rows gstride0 bytes apart
    long long slot_bytes, goff0;
    int gstride0;
    int w, h, wq;              // wq = groups of 4 pixels per row
    unsigned groups;           // wq * h
    unsigned blocks_per_image;
    int tiles_x;               // remap_tile_kernel: 64 x 16 output tiles per row of tiles
    const int4 *tile_box;      // per tile: cx0, ymin, pitch, nrows | has_wrap << 16 (set with the table)
    unsigned n, frames_per_block;   // frames per launch; consecutive frames handled by one block (remap only)
    const uint32_t *map_xy;    // x0 | y0 << 16, MAP_INVALID where the output is 0; rows map_stride entries apart
    const float *map_xf, *map_yf;
    int map_stride;
};

// image.cpp:357: { 0.299, 0.587, 0.114 [, 0] } on channels 0, 1, 2; see oracle/ingest_oracle.c for the rounding
__device__ __forceinline__ uint32_t gray_of(uint32_t c0, uint32_t c1, uint32_t c2)
{
    const float g = ((0.299f * (float)c0 + 0.587f * (float)c1) + 0.114f * (float)c2) + 0.5f;
    return (uint32_t)(int)g;
}

// One gray tap of the remap: 32-bit byte offset from the (wave-uniform) frame base, always inside the frame.
template <int CH>
__device__ __forceinline__ float tap(const uint8_t *img, unsigned off)
{
    if (CH == 1) return (float)img[off];
    return (float)gray_of(img[off], img[off + 1], img[off + 2]);
}

template <int CH, bool REMAP>
__global__ __launch_bounds__(256) void ingest_kernel(IngestArgs a)
{
    // block -> (chunk of frames_per_block consecutive frames, group tile); a thread owns 4 horizontally adjacent
    // output pixels of every frame of its chunk
    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned chunk = bid / a.blocks_per_image;
    const unsigned g = (bid - chunk * a.blocks_per_image) * 256u + threadIdx.x;
    if (g >= a.groups) return;
    const int y = (int)(g / (unsigned)a.wq), x = 4 * (int)(g - (unsigned)y * (unsigned)a.wq);
    const unsigned img0 = chunk * a.frames_per_block;
    const long long dst_off = a.goff0 + (long long)y * a.gstride0 + x;   // level-0 rows are padded to a multiple of 16 bytes
    if (!REMAP) {
        const uint8_t *row = a.src + (long long)img0 * a.image_step + (long long)y * a.src_stride + x * CH;
        uint32_t packed = 0;
        if (x + 4 <= a.w) {
            if (CH == 1) {
                packed = *(const uint32_t *)row;
            } else if (CH == 3) {
                const uint32_t d0 = ((const uint32_t *)row)[0], d1 = ((const uint32_t *)row)[1], d2 = ((const uint32_t *)row)[2];
                packed = gray_of(d0 & 255u, (d0 >> 8) & 255u, (d0 >> 16) & 255u)
                       | gray_of(d0 >> 24, d1 & 255u, (d1 >> 8) & 255u) << 8
                       | gray_of((d1 >> 16) & 255u, d1 >> 24, d2 & 255u) << 16
                       | gray_of((d2 >> 8) & 255u, (d2 >> 16) & 255u, d2 >> 24) << 24;
            } else {
                const uint4 d = *(const uint4 *)row;
                packed = gray_of(d.x & 255u, (d.x >> 8) & 255u, (d.x >> 16) & 255u)
                       | gray_of(d.y & 255u, (d.y >> 8) & 255u, (d.y >> 16) & 255u) << 8
                       | gray_of(d.z & 255u, (d.z >> 8) & 255u, (d.z >> 16) & 255u) << 16
                       | gray_of(d.w & 255u, (d.w >> 8) & 255u, (d.w >> 16) & 255u) << 24;
            }
        } else {
            for (int o = 0; o < 4 && x + o < a.w; ++o) {
                const uint8_t *p = row + o * CH;
                packed |= (CH == 1 ? (uint32_t)p[0] : gray_of(p[0], p[1], p[2])) << (8 * o);
            }
        }
        *(uint32_t *)(a.slab + (long long)(a.slots ? a.slots[img0] : a.slot0) * a.slot_bytes + dst_off) = packed;
        return;
    }
    // ---- remap: the table entries of the 4 pixels are decoded once into tap offsets and weights and stay in
    // registers for every frame of the chunk, so the 12 B/pixel table is read once per chunk instead of once per
    // frame (at one frame per block the table traffic, not the frames, set the kernel time: 0.31 ms per 256 frames).
    const long long m = (long long)y * a.map_stride + x;     // map rows are padded to a multiple of 4 entries
    const uint4 xy = *(const uint4 *)(a.map_xy + m);
    const float4 xf = *(const float4 *)(a.map_xf + m), yf = *(const float4 *)(a.map_yf + m);
    const uint32_t xys[4] = {xy.x, xy.y, xy.z, xy.w};
    const float xfs[4] = {xf.x, xf.y, xf.z, xf.w}, yfs[4] = {yf.x, yf.y, yf.z, yf.w};
    const int hm = a.h - 1;
    const unsigned stride = (unsigned)a.src_stride;
    unsigned off[4][4];
    unsigned zero = 0;       // bit 4*o + k: tap k of pixel o reads as 0; bit 16 + o: pixel o is 0
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const bool ok = xys[o] != MAP_INVALID;
        const unsigned x0 = ok ? (xys[o] & 0xFFFFu) : 0u, y0 = ok ? (xys[o] >> 16) : 0u;
        // undistorter.cpp:99 indexes a continuous image without a bounds check: column w is column 0 of the next
        // row, anything past the last row reads as 0 (the definition shared with the oracle). Offsets are 32-bit
        // (a frame is far below 4 GB) with 24-bit multiplies: full-rate VALU instead of 64-bit address arithmetic.
        // Masked taps load from the (always legal) first tap so that no lane needs a branch.
        const unsigned rowoff = __umul24(y0, stride), o00 = rowoff + x0 * CH;
        const bool wrap = (int)x0 + 1 == a.w;
        const unsigned o01 = wrap ? rowoff + stride : o00 + CH;
        const bool z01 = wrap && (int)y0 >= hm, z10 = (int)y0 >= hm, z11 = (int)y0 + (wrap ? 1 : 0) >= hm;
        off[o][0] = o00; off[o][1] = z01 ? o00 : o01; off[o][2] = z10 ? o00 : o00 + stride; off[o][3] = z11 ? o00 : o01 + stride;
        zero |= ((z01 ? 2u : 0u) | (z10 ? 4u : 0u) | (z11 ? 8u : 0u)) << (4 * o) | (ok ? 0u : 1u << (16 + o));
    }
    for (unsigned f = 0; f < a.frames_per_block && img0 + f < a.n; ++f) {
        const uint8_t *src = a.src + (long long)(img0 + f) * a.image_step;     // wave-uniform base, 32-bit lane offsets
        float t[4][4];
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int k = 0; k < 4; ++k) t[o][k] = tap<CH>(src, off[o][k]);       // all 16 taps in flight before the first use
        uint32_t packed = 0;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const float fx = xfs[o], fy = yfs[o];
#pragma unroll
            for (int k = 1; k < 4; ++k) if (zero >> (4 * o + k) & 1u) t[o][k] = 0.0f;
            // undistorter.cpp:95-101: out += in * wx * wy, rows outer, float accumulation from 0
            float out = 0.0f;
            out += t[o][0] * (1 - fx) * (1 - fy);
            out += t[o][1] * fx * (1 - fy);
            out += t[o][2] * (1 - fx) * fy;
            out += t[o][3] * fx * fy;
            // :107 int(out + 0.5) with the sum formed in double, i.e. floor(out + 0.5) exactly (0 <= out <= 255 has 24
            // significant bits, so the double sum is exact). Same value without f64: the fraction out - trunc(out)
            // is exact in binary32, and the result is trunc(out) + (fraction >= 0.5).
            const int tr = (int)out;
            uint32_t v = (uint32_t)(tr + ((out - (float)tr) >= 0.5f ? 1 : 0)) & 255u;
            if (zero >> (16 + o) & 1u) v = 0;
            packed |= v << (8 * o);
        }
        *(uint32_t *)(a.slab + (long long)(a.slots ? a.slots[img0 + f] : a.slot0) * a.slot_bytes + dst_off) = packed;
    }
}

// ---- remap through an LDS-staged source footprint -------------------------------------------------------------
// Measured with the gather kernel above (256 frames 752x480): time scales with the number of per-lane byte loads
// (16 per thread gray: 0.29 ms, 48 per thread RGB: 0.93 ms, ~28 clocks per wave-level load per CU) and with
// nothing else -- not with the table traffic (one frame per block vs 16: 0.31 vs 0.29 ms), not with the VALU count
// (64-bit vs 24-bit address arithmetic: no change). So the taps must not be global loads. A 64 x 16 output tile
// maps to a compact source rectangle for any smooth camera pair; hv_ingest_set_undistort_map finds that rectangle
// per tile (min / max of the tap positions) when the table is installed, the block stages it into LDS with
// coalesced dword loads -- converting colour to gray once per SOURCE pixel instead of once per tap -- and takes the
// 16 taps per thread from LDS. The rectangle of frame f+1 is fetched into registers while the taps of frame f are
// computed, and two LDS buffers alternate, so a frame costs one barrier. A camera with a tile whose rectangle does
// not fit (a wild map) uses the gather kernel for the whole image.
constexpr int RT_W = 64, RT_H = 16;            // output tile: 16 lanes x 4 pixels wide, 16 rows
constexpr int RT_K = 4;                        // staged dwords (groups of 4 source pixels) per thread and frame
constexpr int RT_BUF = 256 * RT_K * 4 + 256;   // one LDS buffer: the rectangle + the wrap column

template <int CH> struct Stage { uint32_t d[CH == 1 ? 1 : CH]; };   // the dwords of 4 consecutive source pixels

template <int CH>
__device__ __forceinline__ uint32_t gray4(const Stage<CH> &s)   // 4 consecutive source pixels -> 4 packed gray bytes
{
    if (CH == 1) return s.d[0];
    if (CH == 3) {
        const uint32_t d0 = s.d[0], d1 = s.d[1], d2 = s.d[2];
        return gray_of(d0 & 255u, (d0 >> 8) & 255u, (d0 >> 16) & 255u) | gray_of(d0 >> 24, d1 & 255u, (d1 >> 8) & 255u) << 8
             | gray_of((d1 >> 16) & 255u, d1 >> 24, d2 & 255u) << 16 | gray_of((d2 >> 8) & 255u, (d2 >> 16) & 255u, d2 >> 24) << 24;
    }
    uint32_t r = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t d = s.d[CH == 4 ? q : 0];
        r |= gray_of(d & 255u, (d >> 8) & 255u, (d >> 16) & 255u) << (8 * q);
    }
    return r;
}

template <int CH>
__global__ __launch_bounds__(256) void remap_tile_kernel(IngestArgs a)
{
    constexpr int ND = CH == 1 ? 1 : CH;
    __shared__ __attribute__((aligned(16))) uint8_t tile[2][RT_BUF];
    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned chunk = bid / a.blocks_per_image, tl = bid - chunk * a.blocks_per_image;
    const int ty = (int)(tl / (unsigned)a.tiles_x), tx = (int)(tl - (unsigned)ty * (unsigned)a.tiles_x);
    const int x = tx * RT_W + 4 * (int)(threadIdx.x & 15u), y = ty * RT_H + (int)(threadIdx.x >> 4);
    const bool inside = x < a.w && y < a.h;
    const unsigned img0 = chunk * a.frames_per_block;
    const unsigned nf = min(a.frames_per_block, a.n - img0);
    // staged rectangle of this tile: rows ymin .. ymin+nrows-1 (row h, if reached, is staged as zeros), columns
    // cx0 .. cx0+pitch-1; the tap "column w" of undistorter.cpp:99 (= column 0 of the next row on a continuous image)
    // is kept beside it as one byte per row
    const int4 box = a.tile_box[tl];
    const int cx0 = box.x, ymin = box.y, pitch = box.z, nrows = box.w & 0xFFFF;
    const bool has_wrap = (box.w >> 16) != 0;
    const int wrap_base = nrows * pitch;
    uint32_t xys[4] = {MAP_INVALID, MAP_INVALID, MAP_INVALID, MAP_INVALID};
    float xfs[4] = {0, 0, 0, 0}, yfs[4] = {0, 0, 0, 0};
    if (inside) {
        const long long m = (long long)y * a.map_stride + x;            // map rows are padded to a multiple of 4 entries
        const uint4 xy = *(const uint4 *)(a.map_xy + m);
        const float4 xf = *(const float4 *)(a.map_xf + m), yf = *(const float4 *)(a.map_yf + m);
        xys[0] = xy.x; xys[1] = xy.y; xys[2] = xy.z; xys[3] = xy.w;
        xfs[0] = xf.x; xfs[1] = xf.y; xfs[2] = xf.z; xfs[3] = xf.w;
        yfs[0] = yf.x; yfs[1] = yf.y; yfs[2] = yf.z; yfs[3] = yf.w;
    }
    unsigned off[4][4];
    unsigned dead = 0;                                                   // bit o: pixel o is 0 (camera call failed / outside)
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const bool ok = xys[o] != MAP_INVALID;
        if (!ok) dead |= 1u << o;
        const unsigned x0 = ok ? (xys[o] & 0xFFFFu) : (unsigned)cx0, y0 = ok ? (xys[o] >> 16) : (unsigned)ymin;
        const bool wrap = (int)x0 + 1 == a.w;
        const unsigned r = y0 - (unsigned)ymin, o00 = __umul24(r, (unsigned)pitch) + (x0 - (unsigned)cx0);
        off[o][0] = o00; off[o][2] = o00 + (unsigned)pitch;
        off[o][1] = wrap ? (unsigned)wrap_base + r : o00 + 1u;
        off[o][3] = wrap ? (unsigned)wrap_base + r + 1u : o00 + (unsigned)pitch + 1u;
    }
    // staging work of this thread: dword group i = tid + 256 k of the rectangle (row-major, pitch/4 groups per row)
    const int pd = pitch >> 2, ndw = nrows * pd;
    unsigned sg[RT_K], sl[RT_K];      // byte offset in the frame, byte offset in the LDS buffer
    int sn[RT_K];                     // loadable dwords (0: zero fill, i.e. row h; < ND: the row's stride ends inside the group)
#pragma unroll
    for (int k = 0; k < RT_K; ++k) {
        const int i = (int)threadIdx.x + 256 * k;
        sn[k] = -1; sg[k] = 0; sl[k] = 0;
        if (i < ndw) {
            const int r = i / pd, cg = i - r * pd, row = ymin + r, c = cx0 + 4 * cg;
            sg[k] = (unsigned)row * (unsigned)a.src_stride + (unsigned)(c * CH);
            sl[k] = (unsigned)(r * pitch + 4 * cg);
            sn[k] = row < a.h ? min(ND, (a.src_stride - c * CH) >> 2) : 0;
        }
    }
    const long long dst_off = a.goff0 + (long long)y * a.gstride0 + x;
    Stage<CH> st[RT_K];
    auto fetch = [&](unsigned f) {
        const uint8_t *src = a.src + (long long)(img0 + f) * a.image_step;     // wave-uniform base, 32-bit lane offsets
#pragma unroll
        for (int k = 0; k < RT_K; ++k)
#pragma unroll
            for (int j = 0; j < ND; ++j) {
                st[k].d[j] = 0;
                if (j < sn[k]) st[k].d[j] = *(const uint32_t *)(src + sg[k] + 4 * j);
            }
    };
    if (nf) fetch(0);
    for (unsigned f = 0; f < nf; ++f) {
        uint8_t *buf = tile[f & 1u];
#pragma unroll
        for (int k = 0; k < RT_K; ++k)
            if (sn[k] >= 0) *(uint32_t *)(buf + sl[k]) = gray4<CH>(st[k]);
        if (has_wrap && (int)threadIdx.x < nrows) {
            const int row = ymin + (int)threadIdx.x + 1;
            const uint8_t *p = a.src + (long long)(img0 + f) * a.image_step + (long long)row * a.src_stride;
            buf[wrap_base + threadIdx.x] = row < a.h ? (uint8_t)(CH == 1 ? (uint32_t)p[0] : gray_of(p[0], p[1], p[2])) : (uint8_t)0;
        }
        __syncthreads();               // the only barrier of a frame: buffer f&1 is complete; the reads of the other buffer
                                       // (frame f-1) finished before every thread reached this point
        if (f + 1 < nf) fetch(f + 1);  // in flight while the taps below are computed
        float t[4][4];
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int k = 0; k < 4; ++k) t[o][k] = (float)buf[off[o][k]];
        uint32_t packed = 0;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const float fx = xfs[o], fy = yfs[o];
            // undistorter.cpp:95-101: out += in * wx * wy, rows outer, float accumulation from 0
            float out = 0.0f;
            out += t[o][0] * (1 - fx) * (1 - fy);
            out += t[o][1] * fx * (1 - fy);
            out += t[o][2] * (1 - fx) * fy;
            out += t[o][3] * fx * fy;
            // :107 int(out + 0.5) with the sum formed in double = floor(out + 0.5) exactly; see ingest_kernel
            const int tr = (int)out;
            uint32_t v = (uint32_t)(tr + ((out - (float)tr) >= 0.5f ? 1 : 0)) & 255u;
            if (dead >> o & 1u) v = 0;
            packed |= v << (8 * o);
        }
        if (inside) *(uint32_t *)(a.slab + (long long)(a.slots ? a.slots[img0 + f] : a.slot0) * a.slot_bytes + dst_off) = packed;
    }
}

// knob ingest_gather = 1 (HV_INGEST_GATHER at hv_create) selects the plain gather kernel for the remap (kept for A/B measurements)
bool gather_remap(const Ctx *c) { return c->knob.ingest_gather == 1; }

template <int CH>
void launch_ch(Ctx *c, const IngestArgs &a, bool remap, unsigned grid)
{
    if (remap && !a.tile_box) hipLaunchKernelGGL((ingest_kernel<CH, true>), dim3(grid), dim3(256), 0, c->stream, a);
    else if (remap) hipLaunchKernelGGL((remap_tile_kernel<CH>), dim3(grid), dim3(256), 0, c->stream, a);
    else            hipLaunchKernelGGL((ingest_kernel<CH, false>), dim3(grid), dim3(256), 0, c->stream, a);
}

}  // namespace

int launch_ingest(Ctx *c, int n, const int *slots_dev, int slot0, const uint8_t *src, long long image_step, int src_stride,
                  int channels, int camera)
{
    const PyrLayout &L = c->L;
    IngestArgs a{};
    a.src = src; a.image_step = image_step; a.src_stride = src_stride;
    a.slots = slots_dev; a.slot0 = slot0; a.slab = c->slab; a.slot_bytes = L.slot_bytes; a.goff0 = L.goff[0]; a.gstride0 = L.gstride[0];
    a.w = L.w[0]; a.h = L.h[0]; a.wq = (a.w + 3) / 4;
    a.groups = (unsigned)a.wq * (unsigned)a.h;
    a.blocks_per_image = (a.groups + 255u) / 256u;
    const bool tiled = camera >= 0 && !gather_remap(c) && c->map_tiled[camera];
    if (tiled) {
        a.tile_box = reinterpret_cast<const int4 *>(c->d_tile_box[camera]);
        a.tiles_x = (a.w + RT_W - 1) / RT_W;
        a.blocks_per_image = (unsigned)a.tiles_x * (unsigned)((a.h + RT_H - 1) / RT_H);
    }
    const bool remap = camera >= 0;
    if (remap) {
        a.map_xy = c->d_map_xy[camera]; a.map_xf = c->d_map_xf[camera]; a.map_yf = c->d_map_yf[camera];
        a.map_stride = c->map_stride;
    }
    // remap: as many frames per block as leaves >= ~4096 blocks (16 workgroups per CU), at most 16
    a.n = (unsigned)n;
    a.frames_per_block = 1;
    if (remap) a.frames_per_block = (unsigned)std::max(1ll, std::min(16ll, (long long)n * a.blocks_per_image / 4096));
    const unsigned grid = a.blocks_per_image * (((unsigned)n + a.frames_per_block - 1) / a.frames_per_block);
    {
        ScopedKernelTime tm(c, HV_K_INGEST);
        if (channels == 1) launch_ch<1>(c, a, remap, grid);
        else if (channels == 3) launch_ch<3>(c, a, remap, grid);
        else launch_ch<4>(c, a, remap, grid);
        HV_HIP(c, hipGetLastError());
    }
    return HV_OK;
}

}  // namespace hv

using hv::Ctx;

extern "C" {

int hv_ingest_set_undistort_map(hv_ctx *h, int camera, const double *pix_orig_xy, const uint8_t *valid)
{
    Ctx *c = hv::ctx_of(h);
    if (!c || camera < 0 || camera >= HV_INGEST_CAMERAS) return HV_ERR_INVALID;
    const int w = c->L.w[0], hgt = c->L.h[0];
    if (w >= 65535 || hgt >= 65535) return HV_ERR_INVALID;
    if (c->d_map_xy[camera]) {
        HV_HIP(c, hipStreamSynchronize(c->stream));
        (void)hipFree(c->d_map_xy[camera]); (void)hipFree(c->d_map_xf[camera]); (void)hipFree(c->d_map_yf[camera]);
        c->d_map_xy[camera] = nullptr; c->d_map_xf[camera] = c->d_map_yf[camera] = nullptr;
        if (c->d_tile_box[camera]) { (void)hipFree(c->d_tile_box[camera]); c->d_tile_box[camera] = nullptr; }
        c->map_tiled[camera] = false;
    }
    if (!pix_orig_xy) return HV_OK;                       // rectification off for this camera
    const int ms = (w + 3) / 4 * 4;
    c->map_stride = ms;
    std::vector<uint32_t> xy((size_t)ms * hgt, hv::MAP_INVALID);
    std::vector<float> xf((size_t)ms * hgt, 0.0f), yf((size_t)ms * hgt, 0.0f);
    for (int y = 0; y < hgt; ++y)
        for (int x = 0; x < w; ++x) {
            const size_t i = (size_t)y * w + x, o = (size_t)y * ms + x;
            const double px = pix_orig_xy[2 * i], py = pix_orig_xy[2 * i + 1];
            if (valid && !valid[i]) continue;                                       // undistorter.cpp:88 (camera calls failed)
            if (!(px >= 0 && px < w && py >= 0 && py < hgt)) continue;               // :89
            const int x0 = (int)std::floor(px), y0 = (int)std::floor(py);           // :91
            xy[o] = (uint32_t)x0 | (uint32_t)y0 << 16;
            xf[o] = (float)(px - x0); yf[o] = (float)(py - y0);                     // :92 (double difference, then float)
        }
    // source rectangle of every 64 x 16 output tile (see remap_tile_kernel); one tile that does not fit sends the
    // whole camera to the gather kernel
    const int tiles_x = (w + hv::RT_W - 1) / hv::RT_W, tiles_y = (hgt + hv::RT_H - 1) / hv::RT_H;
    std::vector<int> boxes((size_t)tiles_x * tiles_y * 4, 0);
    bool tiled = true;
    for (int ty = 0; ty < tiles_y; ++ty)
        for (int tx = 0; tx < tiles_x; ++tx) {
            int xmin = 1 << 30, xmax = -1, ymin = 1 << 30, ymax = -1;
            for (int y = ty * hv::RT_H; y < std::min(hgt, (ty + 1) * hv::RT_H); ++y)
                for (int x = tx * hv::RT_W; x < std::min(w, (tx + 1) * hv::RT_W); ++x) {
                    const uint32_t e = xy[(size_t)y * ms + x];
                    if (e == hv::MAP_INVALID) continue;
                    const int x0 = (int)(e & 0xFFFFu), y0 = (int)(e >> 16);
                    xmin = std::min(xmin, x0); xmax = std::max(xmax, x0); ymin = std::min(ymin, y0); ymax = std::max(ymax, y0);
                }
            int *b = &boxes[4 * ((size_t)ty * tiles_x + tx)];
            if (xmax < 0) continue;                                                   // nothing valid: all zeros, nothing staged
            const int cx0 = xmin & ~3, cx1 = (std::min(xmax + 1, w - 1) + 4) & ~3;
            const int pitch = cx1 - cx0, nrows = ymax + 2 - ymin;
            b[0] = cx0; b[1] = ymin; b[2] = pitch; b[3] = nrows | (xmax == w - 1 ? 1 << 16 : 0);
            if (nrows > 256 || nrows * (pitch / 4) > 256 * hv::RT_K) tiled = false;
        }
    c->map_tiled[camera] = tiled;
    if (c->d_tile_box[camera]) { (void)hipFree(c->d_tile_box[camera]); c->d_tile_box[camera] = nullptr; }
    HV_HIP(c, hipMalloc((void **)&c->d_tile_box[camera], boxes.size() * sizeof(int)));
    HV_HIP(c, hipMemcpy(c->d_tile_box[camera], boxes.data(), boxes.size() * sizeof(int), hipMemcpyHostToDevice));
    const size_t n = (size_t)ms * hgt;
    HV_HIP(c, hipMalloc((void **)&c->d_map_xy[camera], n * 4));
    HV_HIP(c, hipMalloc((void **)&c->d_map_xf[camera], n * 4));
    HV_HIP(c, hipMalloc((void **)&c->d_map_yf[camera], n * 4));
    HV_HIP(c, hipMemcpy(c->d_map_xy[camera], xy.data(), n * 4, hipMemcpyHostToDevice));
    HV_HIP(c, hipMemcpy(c->d_map_xf[camera], xf.data(), n * 4, hipMemcpyHostToDevice));
    HV_HIP(c, hipMemcpy(c->d_map_yf[camera], yf.data(), n * 4, hipMemcpyHostToDevice));
    return HV_OK;
}

static int ingest_check(Ctx *c, int channels, int camera)
{
    if (!c || (channels != 1 && channels != 3 && channels != 4)) return HV_ERR_INVALID;
    if (camera >= HV_INGEST_CAMERAS) return HV_ERR_INVALID;
    if (camera >= 0 && !c->d_map_xy[camera]) return HV_ERR_INVALID;                 // no map set for that camera
    return HV_OK;
}

int hv_ingest_build_batch_dev(hv_ctx *h, int n, const int *slots_dev, const uint8_t *src_dev,
                              long long image_stride_bytes, int row_stride_bytes, int channels, int camera)
{
    Ctx *c = hv::ctx_of(h);
    int rc = ingest_check(c, channels, camera);
    if (rc != HV_OK) return rc;
    if (n < 0 || (n > 0 && (!slots_dev || !src_dev)) || row_stride_bytes < c->L.w[0] * channels) return HV_ERR_INVALID;
    if (((uintptr_t)src_dev | (uintptr_t)image_stride_bytes | (uintptr_t)row_stride_bytes) & 3u) return HV_ERR_INVALID;
    if (n == 0) return HV_OK;
    rc = hv::launch_ingest(c, n, slots_dev, 0, src_dev, image_stride_bytes, row_stride_bytes, channels, camera);
    if (rc != HV_OK) return rc;
    const hv::PyrLayout &L = c->L;
    return hv::launch_pyramid_levels(c, n, slots_dev, c->slab + L.goff[0], L.slot_bytes, L.gstride[0], true);
}

int hv_ingest_build(hv_ctx *h, int slot, const uint8_t *image_host, int stride_bytes, int channels, int camera)
{
    Ctx *c = hv::ctx_of(h);
    int rc = ingest_check(c, channels, camera);
    if (rc != HV_OK) return rc;
    if (!image_host || stride_bytes < c->L.w[0] * channels) return HV_ERR_INVALID;
    if (slot < 0 || slot >= c->p.pool_size || !c->slot_used[slot]) return HV_ERR_POOL;
    const int w = c->L.w[0], hgt = c->L.h[0];
    const int st = (w * channels + 15) / 16 * 16;
    const size_t need = (size_t)st * hgt;
    if (c->ingest_stage_bytes < need) {
        HV_HIP(c, hipStreamSynchronize(c->stream));
        if (c->d_ingest_stage) (void)hipFree(c->d_ingest_stage);
        c->d_ingest_stage = nullptr; c->ingest_stage_bytes = 0;
        HV_HIP(c, hipMalloc((void **)&c->d_ingest_stage, need));
        c->ingest_stage_bytes = need;
    }
    HV_HIP(c, hipMemcpy2DAsync(c->d_ingest_stage, st, image_host, stride_bytes, (size_t)w * channels, hgt,
                               hipMemcpyHostToDevice, c->stream));
    rc = hv::launch_ingest(c, 1, nullptr, slot, c->d_ingest_stage, 0, st, channels, camera);
    if (rc != HV_OK) return rc;
    return hv::build_levels_of_slot(c, slot);
}

}  // extern "C"
