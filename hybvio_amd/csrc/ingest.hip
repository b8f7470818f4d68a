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

#include <cmath>

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

template <int CH>
__device__ __forceinline__ float tap(const uint8_t *img, int stride, int w, int h, int yy, int xx)
{
    // undistorter.cpp:99 indexes a continuous image without a bounds check: column w is column 0 of the next
    // row, anything past the last row reads as 0 (the definition shared with the oracle)
    if (xx == w) { xx = 0; ++yy; }
    if (yy >= h) return 0.0f;
    const uint8_t *p = img + (long long)yy * stride + xx * CH;
    if (CH == 1) return (float)p[0];
    return (float)gray_of(p[0], p[1], p[2]);
}

template <int CH, bool REMAP>
__global__ __launch_bounds__(256) void ingest_kernel(IngestArgs a)
{
    const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned img = bid / a.blocks_per_image;
    const unsigned g = (bid - img * a.blocks_per_image) * 256u + threadIdx.x;
    if (g >= a.groups) return;
    const int y = (int)(g / (unsigned)a.wq), x = 4 * (int)(g - (unsigned)y * (unsigned)a.wq);
    const uint8_t *src = a.src + (long long)img * a.image_step;
    uint32_t packed = 0;
    if (!REMAP) {
        const uint8_t *row = src + (long long)y * a.src_stride + x * CH;
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
    } else {
        const long long m = (long long)y * a.map_stride + x;     // map rows are padded to a multiple of 4 entries
        const uint4 xy = *(const uint4 *)(a.map_xy + m);
        const float4 xf = *(const float4 *)(a.map_xf + m), yf = *(const float4 *)(a.map_yf + m);
        const uint32_t xys[4] = {xy.x, xy.y, xy.z, xy.w};
        const float xfs[4] = {xf.x, xf.y, xf.z, xf.w}, yfs[4] = {yf.x, yf.y, yf.z, yf.w};
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            uint32_t v = 0;
            if (xys[o] != MAP_INVALID) {
                const int x0 = (int)(xys[o] & 0xFFFFu), y0 = (int)(xys[o] >> 16);
                const float fx = xfs[o], fy = yfs[o];
                const float t00 = tap<CH>(src, a.src_stride, a.w, a.h, y0, x0);
                const float t01 = tap<CH>(src, a.src_stride, a.w, a.h, y0, x0 + 1);
                const float t10 = tap<CH>(src, a.src_stride, a.w, a.h, y0 + 1, x0);
                const float t11 = tap<CH>(src, a.src_stride, a.w, a.h, y0 + 1, x0 + 1);
                // undistorter.cpp:95-101: out += in * wx * wy, rows outer, float accumulation from 0
                float out = 0.0f;
                out += t00 * (1 - fx) * (1 - fy);
                out += t01 * fx * (1 - fy);
                out += t10 * (1 - fx) * fy;
                out += t11 * fx * fy;
                v = (uint32_t)(int)((double)out + 0.5) & 255u;   // :107 int(out + 0.5): the sum is formed in double
            }
            packed |= v << (8 * o);
        }
    }
    uint8_t *dst = a.slab + (long long)(a.slots ? a.slots[img] : a.slot0) * a.slot_bytes + a.goff0 + (long long)y * a.gstride0 + x;
    *(uint32_t *)dst = packed;       // level-0 rows are padded to a multiple of 16 bytes
}

template <int CH>
void launch_ch(Ctx *c, const IngestArgs &a, bool remap, unsigned grid)
{
    if (remap) hipLaunchKernelGGL((ingest_kernel<CH, true>), dim3(grid), dim3(256), 0, c->stream, a);
    else       hipLaunchKernelGGL((ingest_kernel<CH, false>), dim3(grid), dim3(256), 0, c->stream, a);
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
    const bool remap = camera >= 0;
    if (remap) {
        a.map_xy = c->d_map_xy[camera]; a.map_xf = c->d_map_xf[camera]; a.map_yf = c->d_map_yf[camera];
        a.map_stride = c->map_stride;
    }
    const unsigned grid = a.blocks_per_image * (unsigned)n;
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
