// Image pyramid for the LK tracker: fused Scharr-gradient + 5x5 binomial down-sample stencils.
//
// Replaces cv::buildOpticalFlowPyramid as called by CpuImagePyramidFactory::compute
// (reference: src/tracker/image_pyramid.cpp:40-48). One launch per pyramid level over a whole
// batch of images: every workgroup stages one 128x32 source tile (+halo) in LDS with coalesced
// row loads, then writes (a) the level's Scharr gradients (int16 dx|dy per pixel, 16 B per lane,
// 512 B contiguous per half-wave) and (b) the next level's gray tile (u8, (s+128)>>8).
// The source level is read from HBM exactly once; nothing else is written, so HBM traffic equals
// the algorithmic bytes of SURVEY.md section 8(d). Integer arithmetic throughout: results are
// bit-identical to the OpenCV algorithm restated in oracle/pyrlk_oracle.c.
#include "hv_internal.hpp"

namespace hv {

namespace {

constexpr int TW = 128;            // tile width  (source pixels)
constexpr int TH = 32;             // tile height (source pixels)
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
    int dstride;               // dwords per gradient row
    long long goff_next;       // next gray level offset inside the slot
    int gstride_next;
    int w, h, wn, hn;
    int tiles_x, tiles_y;
    const uint8_t **l0_ptr;    // level 0 only: per-slot pointer table filled here
    int *l0_stride;
};

__device__ __forceinline__ int byte_of(uint32_t v, int i) { return (v >> (8 * i)) & 0xFF; }

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
    const bool aligned = ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)a.src_stride) & 3u) == 0;
    const bool interior = aligned && x0 >= 4 && x0 + TW + 4 <= a.w && y0 >= 2 && y0 + TH + 2 <= a.h;
    if (interior) {
        for (int i = t; i < LH * LWD; i += 256) {
            const int r = i / LWD, c = i - r * LWD;
            tile[i] = *reinterpret_cast<const uint32_t *>(
                src + (long long)(y0 - 2 + r) * a.src_stride + (x0 - 4 + 4 * c));
        }
    } else {
        for (int i = t; i < LH * LWD; i += 256) {
            const int r = i / LWD, c = i - r * LWD;
            const uint8_t *row = src + (long long)reflect101(y0 - 2 + r, a.h) * a.src_stride;
            const int x = x0 - 4 + 4 * c;
            tile[i] = (uint32_t)row[reflect101(x, a.w)] | ((uint32_t)row[reflect101(x + 1, a.w)] << 8) |
                      ((uint32_t)row[reflect101(x + 2, a.w)] << 16) |
                      ((uint32_t)row[reflect101(x + 3, a.w)] << 24);
        }
    }
    __syncthreads();

    uint8_t *slot_base = a.slab + (long long)slot * a.slot_bytes;

    // ---- Scharr gradients of the source level: lane = 4 consecutive pixels, 8 rows per pass ----
    {
        uint32_t *dbase = reinterpret_cast<uint32_t *>(slot_base + a.doff);
        const int cg = t & 31, r0 = t >> 5;
#pragma unroll
        for (int pass = 0; pass < TH / 8; ++pass) {
            const int r = r0 + pass * 8;
            const int y = y0 + r, x = x0 + 4 * cg;
            if (y < a.h && x < a.w) {
                const uint32_t *p = tile + (r + 1) * LWD + cg;   // LDS rows r+1..r+3 = source rows y-1..y+1
                int t0[6], t1[6];
                {
                    const uint32_t a0 = p[0], a1 = p[1], a2 = p[2];
                    const uint32_t b0 = p[LWD], b1 = p[LWD + 1], b2 = p[LWD + 2];
                    const uint32_t c0 = p[2 * LWD], c1 = p[2 * LWD + 1], c2 = p[2 * LWD + 2];
                    // columns x-1 .. x+4 are bytes 3 .. 8 of the 12 staged bytes
                    const int top[6] = {byte_of(a0, 3), byte_of(a1, 0), byte_of(a1, 1), byte_of(a1, 2), byte_of(a1, 3), byte_of(a2, 0)};
                    const int mid[6] = {byte_of(b0, 3), byte_of(b1, 0), byte_of(b1, 1), byte_of(b1, 2), byte_of(b1, 3), byte_of(b2, 0)};
                    const int bot[6] = {byte_of(c0, 3), byte_of(c1, 0), byte_of(c1, 1), byte_of(c1, 2), byte_of(c1, 3), byte_of(c2, 0)};
#pragma unroll
                    for (int j = 0; j < 6; ++j) {
                        t0[j] = (top[j] + bot[j]) * 3 + mid[j] * 10;
                        t1[j] = bot[j] - top[j];
                    }
                }
                uint32_t o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int dx = t0[i + 2] - t0[i];
                    const int dy = (t1[i] + t1[i + 2]) * 3 + t1[i + 1] * 10;
                    // stored as 4*d + 2 (|.| <= 16322 fits int16): the LK kernel then takes the high
                    // half of a v_dot2 chain instead of add-and-shift (klt.hip); hv_pyramid_download
                    // undoes it with an arithmetic >> 2
                    o[i] = ((uint32_t)((dx << GRAD_SHIFT) + 2) & 0xFFFFu) | ((uint32_t)((dy << GRAD_SHIFT) + 2) << 16);
                }
                uint32_t *d = dbase + (long long)y * a.dstride + x;
                if (x + 3 < a.w) {
                    *reinterpret_cast<uint4 *>(d) = make_uint4(o[0], o[1], o[2], o[3]);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (x + i < a.w) d[i] = o[i];
                }
            }
        }
    }

    // ---- next gray level: lane = 4 consecutive outputs of one output row ----
    if (DOWN) {
        const int ocg = t & 15, orow = t >> 4;
        const int oy = (y0 >> 1) + orow, ox = (x0 >> 1) + 4 * ocg;
        if (oy < a.hn && ox < a.wn) {
            int acc[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int kv = (j == 0 || j == 4) ? 1 : (j == 2 ? 6 : 4);
                const uint32_t *p = tile + (2 * orow + j) * LWD + 2 * ocg;
                const uint32_t q[4] = {p[0], p[1], p[2], p[3]};
                int c[11];   // source columns 2*ox-2 .. 2*ox+8 are bytes 2 .. 12 of the 16 staged bytes
#pragma unroll
                for (int k = 0; k < 11; ++k) c[k] = byte_of(q[(k + 2) >> 2], (k + 2) & 3);
#pragma unroll
                for (int o = 0; o < 4; ++o)
                    acc[o] += kv * (c[2 * o] + c[2 * o + 4] + 4 * (c[2 * o + 1] + c[2 * o + 3]) + 6 * c[2 * o + 2]);
            }
            uint8_t *g = slot_base + a.goff_next + (long long)oy * a.gstride_next + ox;
            uint32_t packed = 0;
#pragma unroll
            for (int o = 0; o < 4; ++o) packed |= (uint32_t)((acc[o] + 128) >> 8) << (8 * o);
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

}  // namespace

int launch_pyramid_levels(Ctx *c, int n, const int *slots_dev, const uint8_t *src_base,
                          long long src_step, int src_stride, bool src_indexed_by_slot)
{
    const PyrLayout &L = c->L;
    for (int l = 0; l < L.levels; ++l) {
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
        a.w = L.w[l]; a.h = L.h[l];
        const bool down = l + 1 < L.levels;
        if (down) { a.goff_next = L.goff[l + 1]; a.gstride_next = L.gstride[l + 1]; a.wn = L.w[l + 1]; a.hn = L.h[l + 1]; }
        a.tiles_x = (a.w + TW - 1) / TW; a.tiles_y = (a.h + TH - 1) / TH;
        const unsigned grid = (unsigned)(a.tiles_x * a.tiles_y * n);
        ScopedKernelTime tm(c, l == 0 ? HV_K_PYR_L0 : HV_K_PYR_LN);
        if (down) hipLaunchKernelGGL(pyr_level_kernel<true>, dim3(grid), dim3(256), 0, c->stream, a);
        else      hipLaunchKernelGGL(pyr_level_kernel<false>, dim3(grid), dim3(256), 0, c->stream, a);
        HV_HIP(c, hipGetLastError());
    }
    return HV_OK;
}

}  // namespace hv
