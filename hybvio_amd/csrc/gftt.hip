// GFTT feature detector: min-eigenvalue corner response + per-block arg-max.
//
// Replaces the detector HybVIO runs on the CPU (featureDetector "GPU-GFTT" without a GPU image
// factory): CpuCornerResponse = cv::cornerMinEigenVal(gray, response, gfttBlockSize, 3)
// (src/tracker/feature_detector.cpp:279-315), CollectMax::cpuImplementation (:393-417) and the
// sort / zero-prefix / applyMinDistance tail of detect() (:610-634, feature_detector_legacy.cpp:
// 177-213). Algorithm and float evaluation order: oracle/gftt_oracle.c.
//
// Design for CDNA4: the reference materialises five full-size float images (Dx, Dy, three products)
// plus the response before it scans for block maxima. Here one workgroup owns one bs x bs block of
// one image: it stages the (bs+4)^2 gray neighbourhood in LDS (BORDER_REFLECT_101), forms the
// derivative products and their 3x3 box sums there, reduces 16*response to the block's arg-max
// (first maximum in raster order, as the reference's scan) and writes ONE 12-byte key point. The
// level-0 image already sits in HBM for the pyramid: the detector's HBM traffic is that image read
// once (+27 % halo); the response map never exists. Sorting the <= (w/bs)(h/bs) key points and the
// greedy min-distance filter are the reference's host code and stay on the host.
#include <algorithm>
#include <vector>

#include "hv_internal.hpp"

namespace hv {

namespace {

struct GfttArgs {
    const uint8_t *const *l0_ptr;     // per-slot level-0 image table (filled by the pyramid build)
    const int *l0_stride;
    const int *slots;                 // [n_images], or null: the single image slot0
    int slot0;
    int w, h, nbx, nby;
    float k0, k1, min_response;
    float *kp;                        // [n_images][nby * nbx][3] = x, y, response
};

// total order of the reference's raster scan with strict '>': higher response wins, ties -> lower index
__device__ __forceinline__ bool better(float ra, int ia, float rb, int ib) { return ra > rb || (ra == rb && ia < ib); }

template <int BS>
__global__ __launch_bounds__(256) void gftt_block_kernel(GfttArgs a)
{
    constexpr int GW = BS + 4, CW = BS + 2;          // gray tile / product tile edge
    __shared__ float gray[GW * GW];
    __shared__ float cov[3][CW * CW];                // products at tile positions -1 .. BS
    __shared__ float rsum[3][CW * BS];               // row sums at columns 0 .. BS-1, rows -1 .. BS
    __shared__ float red_r[4];
    __shared__ int red_i[4];

    const int t = threadIdx.x;
    const int blocks = a.nbx * a.nby;
    const int img = blockIdx.x / blocks, bi = blockIdx.x - img * blocks;
    const int yb = bi / a.nbx, xb = bi - yb * a.nbx;
    const int x0 = xb * BS, y0 = yb * BS, w = a.w, h = a.h;
    const int slot = a.slots ? a.slots[img] : a.slot0;
    const uint8_t *src = a.l0_ptr[slot];
    const int stride = a.l0_stride[slot];

    for (int i = t; i < GW * GW; i += 256) {
        const int ty = i / GW, tx = i - ty * GW;
        gray[i] = (float)src[(size_t)reflect101(y0 - 2 + ty, h) * stride + reflect101(x0 - 2 + tx, w)];
    }
    __syncthreads();

    // derivative products at tile positions (px, py) in [-1, BS]; positions outside the image take the
    // value of their BORDER_REFLECT_101 mirror, which is what filtering the product images does
    const float k0 = a.k0, k1 = a.k1;
    for (int i = t; i < CW * CW; i += 256) {
        const int py = i / CW - 1, px = i - (py + 1) * CW - 1;
        const int gx = reflect101(x0 + px, w) - x0 + 2, gy = reflect101(y0 + py, h) - y0 + 2;     // gray tile index of the centre
        const float *g = gray + gy * GW + gx;
        const float dt = g[-GW + 1] - g[-GW - 1], dm = g[1] - g[-1], db = g[GW + 1] - g[GW - 1];
        const float vx = k0 * dm + k1 * (dt + db);
        const float st = k0 * g[-GW] + k1 * (g[-GW - 1] + g[-GW + 1]);
        const float sb = k0 * g[GW] + k1 * (g[GW - 1] + g[GW + 1]);
        const float vy = sb - st;
        cov[0][i] = vx * vx; cov[1][i] = vx * vy; cov[2][i] = vy * vy;
    }
    __syncthreads();
    for (int i = t; i < CW * BS; i += 256) {
        const int ry = i / BS, x = i - ry * BS;                  // row ry - 1, column x
        const float *c = &cov[0][ry * CW + x];                   // tile position (x - 1, ry - 1)
#pragma unroll
        for (int ch = 0; ch < 3; ch++) rsum[ch][i] = (c[ch * CW * CW] + c[ch * CW * CW + 1]) + c[ch * CW * CW + 2];
    }
    __syncthreads();

    float best_r = -1e10f;
    int best_i = 0;
    for (int p = t; p < BS * BS; p += 256) {                     // increasing raster order per thread
        const int y = p / BS, x = p - y * BS;
        const float *r = &rsum[0][y * BS + x];                   // rows y-1, y, y+1 are rsum rows y, y+1, y+2
        const float s0 = (r[0] + r[BS]) + r[2 * BS];
        const float s1 = (r[CW * BS] + r[CW * BS + BS]) + r[CW * BS + 2 * BS];
        const float s2 = (r[2 * CW * BS] + r[2 * CW * BS + BS]) + r[2 * CW * BS + 2 * BS];
        const float aa = s0 * 0.5f, bb = s1, cc = s2 * 0.5f, amc = aa - cc;
        const float resp = (aa + cc) - sqrtf(amc * amc + bb * bb);
        const float r16 = resp * 16.0f;                          // CpuCornerResponse::GAIN
        if (r16 > best_r && r16 > a.min_response) { best_r = r16; best_i = p; }
    }
    // threads without a candidate keep (-1e10, 0): index 0 / response -1e10 is also the reference's "no corner"
    for (int o = 32; o > 0; o >>= 1) {
        const float ro = __shfl_down(best_r, o);
        const int io = __shfl_down(best_i, o);
        if (better(ro, io, best_r, best_i)) { best_r = ro; best_i = io; }
    }
    if ((t & 63) == 0) { red_r[t >> 6] = best_r; red_i[t >> 6] = best_i; }
    __syncthreads();
    if (t == 0) {
        for (int k = 1; k < 4; k++) if (better(red_r[k], red_i[k], best_r, best_i)) { best_r = red_r[k]; best_i = red_i[k]; }
        const bool found = best_r > -1e10f;
        float *o = a.kp + ((size_t)img * blocks + bi) * 3;
        o[0] = found ? (float)(x0 + best_i % BS) : 0.f;
        o[1] = found ? (float)(y0 + best_i / BS) : 0.f;
        o[2] = best_r;
    }
}

int launch(Ctx *c, int n_images, const int *slots_dev, int slot0, int bs, float min_response, int block_size, float *kp_dev)
{
    if (block_size != 3) return HV_ERR_UNSUPPORTED;             // the reference default (gfttBlockSize 3); other box sizes: oracle only
    GfttArgs a{};
    a.l0_ptr = c->d_l0_ptr; a.l0_stride = c->d_l0_stride; a.slots = slots_dev; a.slot0 = slot0;
    a.w = c->L.w[0]; a.h = c->L.h[0];
    a.nbx = a.w / bs; a.nby = a.h / bs;                          // std::ceil(int / int): the ragged edge is skipped
    const double scale = 1.0 / (4.0 * block_size * 255.0);
    a.k1 = (float)scale; a.k0 = (float)(2.0 * scale);
    a.min_response = min_response; a.kp = kp_dev;
    const unsigned grid = (unsigned)(a.nbx * a.nby * n_images);
    if (grid == 0) return HV_OK;
    ScopedKernelTime tm(c, HV_K_GFTT);
    if (bs == 32)      hipLaunchKernelGGL(gftt_block_kernel<32>, dim3(grid), dim3(256), 0, c->stream, a);
    else if (bs == 16) hipLaunchKernelGGL(gftt_block_kernel<16>, dim3(grid), dim3(256), 0, c->stream, a);
    else if (bs == 8)  hipLaunchKernelGGL(gftt_block_kernel<8>, dim3(grid), dim3(256), 0, c->stream, a);
    else return HV_ERR_INVALID;
    HV_HIP(c, hipGetLastError());
    return HV_OK;
}

}  // namespace

}  // namespace hv

using hv::Ctx;

extern "C" {

void hv_gftt_default_params(hv_gftt_params *p)
{
    if (!p) return;
    p->gfttBlockSize = 3; p->gfttMinDistance = 50.0; p->gfttMinResponse = 1e-3f; p->maxTracks = 200;
}

int hv_gftt_block_size(const hv_gftt_params *p)
{
    if (!p) return HV_ERR_INVALID;
    const int target = (int)p->gfttMinDistance;                  // feature_detector.cpp:428-436
    return target >= 32 ? 32 : target >= 16 ? 16 : 8;
}

int hv_gftt_keypoint_count(hv_ctx *ctx, const hv_gftt_params *p)
{
    if (!ctx || !p) return HV_ERR_INVALID;
    Ctx *c = hv::ctx_of(ctx);
    const int bs = hv_gftt_block_size(p);
    return (c->L.w[0] / bs) * (c->L.h[0] / bs);
}

int hv_gftt_keypoints_batch_dev(hv_ctx *ctx, const hv_gftt_params *p, int n_images, const int *slots_dev, float *kp_dev)
{
    if (!ctx || !p || n_images < 0 || (n_images > 0 && (!slots_dev || !kp_dev))) return HV_ERR_INVALID;
    Ctx *c = hv::ctx_of(ctx);
    return hv::launch(c, n_images, slots_dev, 0, hv_gftt_block_size(p), p->gfttMinResponse, p->gfttBlockSize, kp_dev);
}

void hv_apply_min_distance(float *corners_xy, int *n_inout, const float *prev_xy, int n_prev, int r, int max_tracks)
{
    // FeatureDetector::applyMinDistance (feature_detector_legacy.cpp:177-213): in place, in order
    if (!corners_xy || !n_inout) return;
    int n_out = 0;
    const float r2 = static_cast<float>(r * r);
    for (int k = 0; k < *n_inout; ++k) {
        const float cx = corners_xy[2 * k], cy = corners_xy[2 * k + 1];
        bool near_other = false;
        if (r > 0) {
            for (int i = 0; i < n_prev && !near_other; ++i) {
                const float dx = prev_xy[2 * i] - cx, dy = prev_xy[2 * i + 1] - cy;
                near_other = dx * dx + dy * dy < r2;
            }
            for (int i = 0; i < n_out && !near_other; ++i) {
                const float dx = corners_xy[2 * i] - cx, dy = corners_xy[2 * i + 1] - cy;
                near_other = dx * dx + dy * dy < r2;
            }
        }
        if (!near_other) { corners_xy[2 * n_out] = cx; corners_xy[2 * n_out + 1] = cy; ++n_out; }
        if (n_out >= max_tracks) break;
    }
    *n_inout = n_out;
}

int hv_gftt_detect(hv_ctx *ctx, const hv_gftt_params *p, int slot, const float *prev_xy, int n_prev, int mask_radius,
                   float *corners_xy, int capacity, int *n_out)
{
    if (!ctx || !p || !corners_xy || !n_out || n_prev < 0 || (n_prev > 0 && !prev_xy)) return HV_ERR_INVALID;
    Ctx *c = hv::ctx_of(ctx);
    if (slot < 0 || slot >= c->p.pool_size || !c->slot_used[slot]) return HV_ERR_POOL;
    const int nk = hv_gftt_keypoint_count(ctx, p);
    if (capacity < 2 * nk) return HV_ERR_INVALID;
    *n_out = 0;
    if (nk == 0) return HV_OK;
    if (c->gftt_cap < nk) {
        if (c->d_gftt_kp) { (void)hipFree(c->d_gftt_kp); c->d_gftt_kp = nullptr; c->gftt_cap = 0; }
        HV_HIP(c, hipMalloc(reinterpret_cast<void **>(&c->d_gftt_kp), sizeof(float) * 3 * nk));
        c->gftt_cap = nk;
    }
    int rc = hv::launch(c, 1, nullptr, slot, hv_gftt_block_size(p), p->gfttMinResponse, p->gfttBlockSize, c->d_gftt_kp);
    if (rc != HV_OK) return rc;
    std::vector<float> kp(3 * (size_t)nk);
    HV_HIP(c, hipMemcpyAsync(kp.data(), c->d_gftt_kp, sizeof(float) * 3 * nk, hipMemcpyDeviceToHost, c->stream));
    HV_HIP(c, hipStreamSynchronize(c->stream));
    // detect(): stable sort by descending response, then corners.resize(n) + push_back (n zero points first)
    std::vector<int> order(nk);
    for (int i = 0; i < nk; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return kp[3 * a + 2] > kp[3 * b + 2]; });
    for (int i = 0; i < nk; ++i) { corners_xy[2 * i] = 0.f; corners_xy[2 * i + 1] = 0.f; }
    for (int i = 0; i < nk; ++i) { corners_xy[2 * (nk + i)] = kp[3 * order[i]]; corners_xy[2 * (nk + i) + 1] = kp[3 * order[i] + 1]; }
    int n = 2 * nk;
    if (mask_radius > 0) hv_apply_min_distance(corners_xy, &n, prev_xy, n_prev, mask_radius, p->maxTracks);
    *n_out = n;
    return HV_OK;
}

}  // extern "C"
