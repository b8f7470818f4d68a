// extern "C" entry points of libhybvio_hip.so (see include/hybvio_hip.h): context, pyramid
// pool, host-pointer convenience paths, per-kernel timers.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>

#include "hv_internal.hpp"

namespace hv {

namespace {
struct KnobEntry { const char *name; int Knobs::*field; };
const KnobEntry KNOB_TABLE[] = {
    {"pyr_tail", &Knobs::pyr_tail}, {"pyr_l0_tiled", &Knobs::pyr_l0_tiled}, {"gftt_tiled", &Knobs::gftt_tiled},
    {"klt_tile", &Knobs::klt_tile}, {"ekf_defer_jacobian", &Knobs::ekf_defer_jacobian}, {"vu_threads", &Knobs::vu_threads}, {"ekf_spec_split", &Knobs::ekf_spec_split},
    {"ekf_no_speculation", &Knobs::ekf_no_speculation}, {"ekf_stream_gate", &Knobs::ekf_stream_gate},
    {"ekf_gate_kmode", &Knobs::ekf_gate_kmode}, {"ingest_gather", &Knobs::ingest_gather},
    {"ekf_fused_gate", &Knobs::ekf_fused_gate}, {"ekf_spec_mode", &Knobs::ekf_spec_mode},
    {"rot_ransac_threads", &Knobs::rot_ransac_threads}, {"ekf_side_stream", &Knobs::ekf_side_stream}, {"ekf_dual_update", &Knobs::ekf_dual_update}, {"ekf_long_fused", &Knobs::ekf_long_fused}, {"ekf_long_first", &Knobs::ekf_long_first}, {"ekf_short_np", &Knobs::ekf_short_np}, {"ekf_predict_chain", &Knobs::ekf_predict_chain}, {"ekf_visit_order", &Knobs::ekf_visit_order},
    {"ekf_split_tri", &Knobs::ekf_split_tri}, {"vu_tri_threads", &Knobs::vu_tri_threads},
};
}  // namespace

int knob_set(Knobs &k, const char *name, int value)
{
    if (!name) return HV_ERR_INVALID;
    for (const KnobEntry &e : KNOB_TABLE)
        if (strcmp(e.name, name) == 0) { k.*(e.field) = value; return HV_OK; }
    return HV_ERR_INVALID;
}

int knob_get(const Knobs &k, const char *name, int *value)
{
    if (!name || !value) return HV_ERR_INVALID;
    for (const KnobEntry &e : KNOB_TABLE)
        if (strcmp(e.name, name) == 0) { *value = k.*(e.field); return HV_OK; }
    return HV_ERR_INVALID;
}

// hv_create only: the environment spelling of knob "abc_def" is HV_ABC_DEF
void knobs_from_env(Knobs &k)
{
    for (const KnobEntry &e : KNOB_TABLE) {
        char var[64] = "HV_";
        size_t i = 3;
        for (const char *q = e.name; *q && i + 1 < sizeof var; ++q, ++i) var[i] = (*q >= 'a' && *q <= 'z') ? (char)(*q - 32) : *q;
        var[i] = 0;
        if (const char *v = getenv(var)) k.*(e.field) = atoi(v);
    }
}

int hip_fail(Ctx *c, hipError_t e, const char *what)
{
    if (c) {
        char buf[512];
        snprintf(buf, sizeof buf, "%s -> %s", what, hipGetErrorString(e));
        c->last_error = buf;
    }
    return HV_ERR_HIP;
}

ScopedKernelTime::ScopedKernelTime(Ctx *c_, int id_, hipStream_t stream) : c(c_), id(id_), s(stream ? stream : c_->stream)
{
    if (!c->profiling) return;
    KernelTimer &t = c->timers[id];
    if (!t.free_list.empty()) { a = t.free_list.back().first; b = t.free_list.back().second; t.free_list.pop_back(); }
    else { (void)hipEventCreate(&a); (void)hipEventCreate(&b); }
    (void)hipEventRecord(a, s);
}

ScopedKernelTime::~ScopedKernelTime()
{
    if (!a) return;
    (void)hipEventRecord(b, s);
    c->timers[id].pending.emplace_back(a, b);
}

namespace {

// LK tile loads of a padded level may run a few bytes past the last row of the last slot (unused tile cells)
constexpr size_t SLAB_SLACK = 4096;

inline int align_up(int v, int a) { return (v + a - 1) / a * a; }
inline long long align_up_ll(long long v, long long a) { return (v + a - 1) / a * a; }

// A handful of ints travel to the device as kernel arguments (captured at launch, so the host
// copy may be reused immediately and no pinned staging is needed).
struct IntPack { int v[4]; };
__global__ void set_ints_kernel(int *dst, IntPack p, int n)
{
    if ((int)threadIdx.x < n) dst[threadIdx.x] = p.v[threadIdx.x];
}

int compute_layout(const hv_params &p, PyrLayout &L)
{
    L = PyrLayout{};
    L.win = p.win;
    int w = p.width, h = p.height, n = 0;
    // cv::buildOpticalFlowPyramid: a further level is built only while it stays larger than the window
    for (int l = 0; l < p.levels; ++l) {
        L.w[l] = w; L.h[l] = h; n = l + 1;
        w = (w + 1) / 2; h = (h + 1) / 2;
        if (w <= p.win || h <= p.win) break;
    }
    L.levels = n;
    // HV_PAD_FROM_LEVEL (environment, experiments only): first level with a physical border; >= levels turns it off
    int pad_from = 2;
    if (const char *e = getenv("HV_PAD_FROM_LEVEL")) pad_from = atoi(e);
    if (pad_from < 1) pad_from = 1;                     // level 0 may be the caller's image: never padded
    // HV_GRAD_FROM_LEVEL (environment, experiments only): first level whose gradient plane is stored; 0 = all (the r01 design).
    // HV_L0_GRADIENTS=1 is the older spelling of 0.
    L.grad_from = 2;
    if (const char *e = getenv("HV_GRAD_FROM_LEVEL")) L.grad_from = atoi(e);
    if (const char *e = getenv("HV_L0_GRADIENTS")) if (atoi(e) != 0) L.grad_from = 0;
    if (L.grad_from < 0) L.grad_from = 0;
    if (L.grad_from > pad_from) L.grad_from = pad_from;       // a padded level keeps its plane (constant-0 border)
    if (L.grad_from > n) L.grad_from = n;
    long long off = 0;
    for (int l = 0; l < n; ++l) {
        const int pd = L.pad[l] = l >= pad_from ? PYR_PAD : 0;
        L.gstride[l] = align_up(L.w[l] + 2 * pd, 16);
        L.goff[l] = off + (long long)pd * L.gstride[l] + pd;
        off = align_up_ll(off + (long long)L.gstride[l] * (L.h[l] + 2 * pd), 256);
    }
    for (int l = 0; l < n; ++l) {
        const int pd = L.pad[l];
        L.dstride[l] = align_up(L.w[l] + 2 * pd, 4);
        if (l < L.grad_from) { L.doff[l] = -1; continue; }           // formed inside the LK kernel, never stored
        L.doff[l] = off + ((long long)pd * L.dstride[l] + pd) * 4;
        off = align_up_ll(off + (long long)L.dstride[l] * 4 * (L.h[l] + 2 * pd), 256);
    }
    L.slot_bytes = off;
    return HV_OK;
}

int ensure_point_staging(Ctx *c, int n)
{
    if (n <= c->stage_points) return HV_OK;
    int cap = c->stage_points ? c->stage_points : 256;
    while (cap < n) cap *= 2;
    if (c->d_prev_xy) { (void)hipFree(c->d_prev_xy); (void)hipFree(c->d_next_xy); (void)hipFree(c->d_err); (void)hipFree(c->d_status); }
    c->d_prev_xy = c->d_next_xy = c->d_err = nullptr; c->d_status = nullptr; c->stage_points = 0;
    HV_HIP(c, hipMalloc(&c->d_prev_xy, sizeof(float) * 2 * cap));
    HV_HIP(c, hipMalloc(&c->d_next_xy, sizeof(float) * 2 * cap));
    HV_HIP(c, hipMalloc(&c->d_err, sizeof(float) * cap));
    HV_HIP(c, hipMalloc(&c->d_status, cap));
    c->stage_points = cap;
    return HV_OK;
}

bool slot_ok(Ctx *c, int s) { return s >= 0 && s < c->p.pool_size && c->slot_used[s]; }

// level-0 pointers that refer to the slot's own copy move with the slab; pointers into caller buffers stay
__global__ void rebase_l0_kernel(const uint8_t **dst, const uint8_t *const *src, int *dst_stride, const int *src_stride, int n_old,
                                 int n_new, const uint8_t *old_slab, long long old_bytes, const uint8_t *new_slab)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_new) return;
    const uint8_t *p = nullptr;
    int st = 0;
    if (s < n_old) {
        p = src[s]; st = src_stride[s];
        if (p >= old_slab && p < old_slab + old_bytes) p = new_slab + (p - old_slab);
    }
    dst[s] = p; dst_stride[s] = st;
}

// util::Allocator (src/util/allocator.hpp:55-67) creates a new pyramid whenever every pooled one is still referenced;
// the device pool does the same by doubling the slab (contents and slot numbers are kept). Synchronises the stream;
// HIP graphs captured before the growth hold the old addresses and must be re-captured.
int grow_pool(Ctx *c)
{
    const int n_old = c->p.pool_size, n_new = n_old * 2;
    const long long old_bytes = c->L.slot_bytes * n_old;
    HV_HIP(c, hipStreamSynchronize(c->stream));
    uint8_t *slab = nullptr; const uint8_t **l0p = nullptr; int *l0s = nullptr;
    if (hipMalloc(&slab, (size_t)c->L.slot_bytes * n_new + SLAB_SLACK) != hipSuccess) return HV_ERR_NOMEM;
    if (hipMalloc(&l0p, sizeof(void *) * n_new) != hipSuccess) { (void)hipFree(slab); return HV_ERR_NOMEM; }
    if (hipMalloc(&l0s, sizeof(int) * n_new) != hipSuccess) { (void)hipFree(slab); (void)hipFree(l0p); return HV_ERR_NOMEM; }
    hipError_t e = hipMemcpyAsync(slab, c->slab, (size_t)old_bytes, hipMemcpyDeviceToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(rebase_l0_kernel, dim3((n_new + 255) / 256), dim3(256), 0, c->stream, l0p, c->d_l0_ptr, l0s, c->d_l0_stride,
                           n_old, n_new, c->slab, old_bytes, slab);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { (void)hipFree(slab); (void)hipFree(l0p); (void)hipFree(l0s); return hip_fail(c, e, "grow_pool"); }
    (void)hipFree(c->slab); (void)hipFree(c->d_l0_ptr); (void)hipFree(c->d_l0_stride);
    c->slab = slab; c->d_l0_ptr = l0p; c->d_l0_stride = l0s;
    c->slot_used.resize(n_new, 0);
    for (int s = n_new - 1; s >= n_old; --s) c->free_slots.push_back(s);
    c->p.pool_size = n_new;
    return fill_gradient_borders(c, n_old, n_new - n_old);
}

}  // namespace
}  // namespace hv

using hv::Ctx;

extern "C" {
struct hv_ctx { Ctx c; };
}
namespace hv {
Ctx *ctx_of(hv_ctx *h) { return h ? &h->c : nullptr; }
// all level kernels over the level-0 image already sitting in the slot
int build_levels_of_slot(Ctx *c, int slot)
{
    const PyrLayout &L = c->L;
    IntPack pk{{slot, 0, 0, 0}};
    hipLaunchKernelGGL(set_ints_kernel, dim3(1), dim3(64), 0, c->stream, c->d_slots, pk, 1);
    HV_HIP(c, hipGetLastError());
    return launch_pyramid_levels(c, 1, c->d_slots, c->slab + L.goff[0], L.slot_bytes, L.gstride[0], true);
}
}

extern "C" {

void hv_default_params(hv_params *p)
{
    if (!p) return;
    memset(p, 0, sizeof *p);
    p->device = 0; p->width = 752; p->height = 480;
    p->levels = 4; p->win = 31; p->max_iter = 20; p->eps = 0.03; p->min_eig = 1e-3;
    p->max_tracks = 200; p->pool_size = 16; p->max_pairs = 1;
}

int hv_abi_version(void) { return HV_ABI_VERSION; }

int hv_debug_set_knob(hv_ctx *h, const char *name, int value)
{
    Ctx *c = hv::ctx_of(h);
    return c ? hv::knob_set(c->knob, name, value) : HV_ERR_INVALID;
}

int hv_debug_get_knob(hv_ctx *h, const char *name, int *value)
{
    Ctx *c = hv::ctx_of(h);
    return c ? hv::knob_get(c->knob, name, value) : HV_ERR_INVALID;
}

const char *hv_status_string(int s)
{
    switch (s) {
        case HV_OK: return "ok";
        case HV_ERR_INVALID: return "invalid argument";
        case HV_ERR_UNSUPPORTED: return "unsupported parameter";
        case HV_ERR_NO_DEVICE: return "no HIP device";
        case HV_ERR_HIP: return "HIP runtime error";
        case HV_ERR_POOL: return "pyramid pool exhausted or bad slot";
        case HV_ERR_NOMEM: return "out of memory";
        case HV_ERR_TIMEOUT: return "device hand-shake timed out (result discarded)";
        default: return "unknown status";
    }
}

// Both streams of a context come from ONE queue pool of the runtime. ROCclr keeps a pool of hardware (HSA) queues per stream priority
// (GPU_MAX_HW_QUEUES = 4 each) and hands a new stream the queue of its priority with the fewest users. With the default priority that
// pool is shared with everything else the process created before -- the null stream, the 32 pooled streams of a torch process, the
// streams a HIP graph instantiates for its branches -- so whether two contexts' busy streams ended up on one hardware queue (kernels
// of one queue run one after the other) depended on the process history (r03: 15.8 / 17.3 / 18.6 ms per step for the same two engines
// after different warm-ups). hv_lanes_create therefore takes its streams from the HIGH-priority pool, which nothing else uses
// unless asked to: up to four streams (two lanes) get a hardware queue each, independent of what ran before.
// which: 1 = the context stream, 2 = the second stream, 3 = both. A stream is bound to a hardware queue of its priority's pool (four
// queues) when it first runs a command -- to the queue with the fewest users: hv_lanes_create therefore creates AND uses the context
// streams of all its lanes before any second stream (r06: created pairwise, lanes 0 and 2 -- and 1 and 3 -- shared a queue with each
// other while the second streams, idle under graph replay, held the other two: never more than two lanes' kernels in flight,
// profiles/r06/concurrency_probe.txt)
static int create_streams(Ctx *c, int high_priority, int which)
{
    int least = 0, greatest = 0;
    if (high_priority && hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) high_priority = 0;
    if (high_priority && greatest == least) high_priority = 0;                 // the device has one priority level only
    c->stream_priority = high_priority ? 1 : 0;
    hipStream_t *both[2] = { &c->stream, &c->aux_stream };
    for (int i = 0; i < 2; ++i) {
        if (!(which & (1 << i))) continue;
        const hipError_t e = high_priority ? hipStreamCreateWithPriority(both[i], hipStreamNonBlocking, greatest)
                                           : hipStreamCreateWithFlags(both[i], hipStreamNonBlocking);
        if (e != hipSuccess) return hip_fail(c, e, "hipStreamCreate");
    }
    c->own_stream = true;
    return HV_OK;
}

// the second stream's first command (binds its hardware queue now, not inside the first frame)
static int prime_aux_stream(Ctx *c)
{
    if (hipMemsetAsync(c->d_slots, 0, sizeof(int) * 4, c->aux_stream) != hipSuccess) return HV_ERR_HIP;
    if (hipStreamSynchronize(c->aux_stream) != hipSuccess) return HV_ERR_HIP;
    return HV_OK;
}

static int create_ctx(const hv_params *params, int high_priority, hv_ctx **out)
{
    if (!params || !out) return HV_ERR_INVALID;
    *out = nullptr;
    const hv_params &p = *params;
    if (p.width < 1 || p.height < 1 || p.levels < 1 || p.levels > HV_MAX_LEVELS || p.pool_size < 1 ||
        p.max_iter < 0 || p.max_tracks < 1)
        return HV_ERR_INVALID;
    if (p.win != 31) return HV_ERR_UNSUPPORTED;   // lane layout of the LK kernel is built for 31x31
    if (p.width <= p.win || p.height <= p.win) return HV_ERR_UNSUPPORTED;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || p.device < 0 || p.device >= ndev)
        return HV_ERR_NO_DEVICE;
    hv_ctx *h = new (std::nothrow) hv_ctx();
    if (!h) return HV_ERR_NOMEM;
    Ctx *c = &h->c;
    c->p = p;
    if (c->p.max_pairs < 1) c->p.max_pairs = 1;
    if (high_priority) c->knob.ekf_side_stream = 5;       // lanes: the visit forks onto the second stream only outside a stream capture (ekf.hip)
    hv::knobs_from_env(c->knob);
    hv::compute_layout(p, c->L);
    int rc = HV_OK;
    do {
        if (hipSetDevice(p.device) != hipSuccess) { rc = HV_ERR_NO_DEVICE; break; }
        { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, p.device) == hipSuccess && prop.multiProcessorCount > 0) c->num_cus = prop.multiProcessorCount; }
        rc = create_streams(c, high_priority, high_priority ? 1 : 3);       // (lanes: the second streams follow once every lane's context stream is bound)
        if (rc != HV_OK) break;
        const size_t slab_bytes = (size_t)c->L.slot_bytes * p.pool_size + hv::SLAB_SLACK;
        if (hipMalloc(&c->slab, slab_bytes) != hipSuccess) { rc = HV_ERR_NOMEM; break; }
        if (hipMalloc(&c->d_l0_ptr, sizeof(void *) * p.pool_size) != hipSuccess) { rc = HV_ERR_NOMEM; break; }
        if (hipMalloc(&c->d_l0_stride, sizeof(int) * p.pool_size) != hipSuccess) { rc = HV_ERR_NOMEM; break; }
        if (hipMalloc(&c->d_slots, sizeof(int) * 4) != hipSuccess) { rc = HV_ERR_NOMEM; break; }
        if (hipMemsetAsync(c->d_l0_ptr, 0, sizeof(void *) * p.pool_size, c->stream) != hipSuccess) { rc = HV_ERR_HIP; break; }
        // (the second stream runs its first command here, not inside the first frame: a stream's hardware queue is bound when it is used)
        if (c->aux_stream && prime_aux_stream(c) != HV_OK) { rc = HV_ERR_HIP; break; }
        if (hipStreamSynchronize(c->stream) != hipSuccess) { rc = HV_ERR_HIP; break; }      // the context stream has run its first command
        c->slot_used.assign(p.pool_size, 0);
        for (int s = p.pool_size - 1; s >= 0; --s) c->free_slots.push_back(s);
        rc = hv::ensure_point_staging(c, p.max_tracks);
        if (rc == HV_OK) rc = hv::fill_gradient_borders(c, 0, p.pool_size);
        if (rc == HV_OK) rc = hv::rot_ransac_alloc_split(c);     // (r05 advisor: once, here -- never inside a launch that may be under capture)
    } while (0);
    if (rc != HV_OK) { hv_destroy(h); return rc; }
    *out = h;
    return HV_OK;
}

int hv_create(const hv_params *params, hv_ctx **out) { return create_ctx(params, 0, out); }

/* ---- lanes ---- */
struct hv_lanes { std::vector<hv_ctx *> ctx; };

int hv_lanes_create(const hv_params *params, int n_lanes, hv_lanes **out)
{
    if (!params || !out || n_lanes < 1 || n_lanes > HV_MAX_LANES) return HV_ERR_INVALID;
    *out = nullptr;
    hv_lanes *g = new (std::nothrow) hv_lanes();
    if (!g) return HV_ERR_NOMEM;
    for (int i = 0; i < n_lanes; ++i) {
        hv_ctx *h = nullptr;
        const int rc = create_ctx(params, 1, &h);
        if (rc != HV_OK) { hv_lanes_destroy(g); return rc; }
        g->ctx.push_back(h);
    }
    for (hv_ctx *h : g->ctx) {                               // second streams: behind every context stream (create_streams)
        int rc = create_streams(&h->c, 1, 2);
        if (rc == HV_OK) rc = prime_aux_stream(&h->c);
        if (rc != HV_OK) { hv_lanes_destroy(g); return rc; }
    }
    *out = g;
    return HV_OK;
}

int hv_lanes_count(const hv_lanes *g) { return g ? (int)g->ctx.size() : HV_ERR_INVALID; }
hv_ctx *hv_lanes_ctx(hv_lanes *g, int lane) { return (g && lane >= 0 && lane < (int)g->ctx.size()) ? g->ctx[lane] : nullptr; }

void hv_lanes_destroy(hv_lanes *g)
{
    if (!g) return;
    for (hv_ctx *h : g->ctx) hv_destroy(h);
    delete g;
}

void *hv_get_stream(hv_ctx *h) { return h ? reinterpret_cast<void *>(h->c.stream) : nullptr; }

void hv_destroy(hv_ctx *h)
{
    if (!h) return;
    Ctx *c = &h->c;
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto &t : c->timers) {
        for (auto &e : t.pending) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
        for (auto &e : t.free_list) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    }
    if (c->slab) (void)hipFree(c->slab);
    if (c->d_l0_ptr) (void)hipFree(c->d_l0_ptr);
    if (c->d_l0_stride) (void)hipFree(c->d_l0_stride);
    if (c->d_slots) (void)hipFree(c->d_slots);
    if (c->d_prev_xy) (void)hipFree(c->d_prev_xy);
    if (c->d_next_xy) (void)hipFree(c->d_next_xy);
    if (c->d_err) (void)hipFree(c->d_err);
    if (c->d_status) (void)hipFree(c->d_status);
    if (c->d_gftt_kp) (void)hipFree(c->d_gftt_kp);
    if (c->d_ingest_stage) (void)hipFree(c->d_ingest_stage);
    if (c->d_ransac_stage) (void)hipFree(c->d_ransac_stage);
    if (c->d_ransac_split) (void)hipFree(c->d_ransac_split);
    for (int k = 0; k < HV_INGEST_CAMERAS; ++k)
        if (c->d_tile_box[k]) (void)hipFree(c->d_tile_box[k]);
    for (int k = 0; k < HV_INGEST_CAMERAS; ++k)
        if (c->d_map_xy[k]) { (void)hipFree(c->d_map_xy[k]); (void)hipFree(c->d_map_xf[k]); (void)hipFree(c->d_map_yf[k]); }
    if (c->aux_stream) { (void)hipStreamSynchronize(c->aux_stream); (void)hipStreamDestroy(c->aux_stream); }
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete h;
}

const char *hv_last_error(hv_ctx *h) { return h ? h->c.last_error.c_str() : "null context"; }

int hv_set_stream(hv_ctx *h, void *hip_stream)
{
    if (!h) return HV_ERR_INVALID;
    Ctx *c = &h->c;
    if (c->stream_priority != 0) return HV_ERR_INVALID;   // a lane's streams are the point of the lane set: they stay the library's
    HV_HIP(c, hipStreamSynchronize(c->stream));
    if (c->own_stream) { (void)hipStreamDestroy(c->stream); c->own_stream = false; }
    c->stream = reinterpret_cast<hipStream_t>(hip_stream);
    return HV_OK;
}

int hv_synchronize(hv_ctx *h)
{
    if (!h) return HV_ERR_INVALID;
    HV_HIP(&h->c, hipStreamSynchronize(h->c.stream));
    return HV_OK;
}

/* ---- pyramid ---- */

int hv_pyramid_acquire(hv_ctx *h, int *slot_out)
{
    if (!h || !slot_out) return HV_ERR_INVALID;
    Ctx *c = &h->c;
    if (c->free_slots.empty()) {
        const int rc = hv::grow_pool(c);
        if (rc != HV_OK) return rc == HV_ERR_NOMEM ? HV_ERR_POOL : rc;
    }
    const int s = c->free_slots.back();
    c->free_slots.pop_back();
    c->slot_used[s] = 1;
    *slot_out = s;
    return HV_OK;
}

int hv_pyramid_release(hv_ctx *h, int slot)
{
    if (!h) return HV_ERR_INVALID;
    Ctx *c = &h->c;
    if (!hv::slot_ok(c, slot)) return HV_ERR_POOL;
    c->slot_used[slot] = 0;
    c->free_slots.push_back(slot);
    return HV_OK;
}

int hv_pyramid_level_size(hv_ctx *h, int level, int *width, int *height)
{
    if (!h || level < 0 || level >= h->c.L.levels) return HV_ERR_INVALID;
    if (width) *width = h->c.L.w[level];
    if (height) *height = h->c.L.h[level];
    return HV_OK;
}

int hv_pyramid_build(hv_ctx *h, int slot, const uint8_t *gray_host, int stride_bytes)
{
    if (!h || !gray_host || stride_bytes < h->c.p.width) return HV_ERR_INVALID;
    Ctx *c = &h->c;
    if (!hv::slot_ok(c, slot)) return HV_ERR_POOL;
    const hv::PyrLayout &L = c->L;
    uint8_t *l0 = c->slab + (long long)slot * L.slot_bytes + L.goff[0];
    HV_HIP(c, hipMemcpy2DAsync(l0, L.gstride[0], gray_host, stride_bytes, L.w[0], L.h[0],
                               hipMemcpyHostToDevice, c->stream));
    hv::IntPack pk{{slot, 0, 0, 0}};
    hipLaunchKernelGGL(hv::set_ints_kernel, dim3(1), dim3(64), 0, c->stream, c->d_slots, pk, 1);
    HV_HIP(c, hipGetLastError());
    return hv::launch_pyramid_levels(c, 1, c->d_slots, c->slab + L.goff[0], L.slot_bytes, L.gstride[0], true);
}

int hv_pyramid_build_batch_dev(hv_ctx *h, int n, const int *slots_dev, const uint8_t *gray_dev,
                               long long image_stride_bytes, int row_stride_bytes)
{
    if (!h || n < 0 || (n > 0 && (!slots_dev || !gray_dev)) || row_stride_bytes < h->c.p.width)
        return HV_ERR_INVALID;
    if (n == 0) return HV_OK;
    return hv::launch_pyramid_levels(&h->c, n, slots_dev, gray_dev, image_stride_bytes, row_stride_bytes, false);
}

int hv_pyramid_download(hv_ctx *h, int slot, int level, uint8_t *gray, int16_t *grad)
{
    if (!h) return HV_ERR_INVALID;
    Ctx *c = &h->c;
    if (!hv::slot_ok(c, slot)) return HV_ERR_POOL;
    const hv::PyrLayout &L = c->L;
    if (level < 0 || level >= L.levels) return HV_ERR_INVALID;
    HV_HIP(c, hipStreamSynchronize(c->stream));
    const uint8_t *base = c->slab + (long long)slot * L.slot_bytes;
    if (gray) {
        const uint8_t *src = base + L.goff[level];
        int stride = L.gstride[level];
        if (level == 0) {   // level 0 may live in the caller's buffer
            HV_HIP(c, hipMemcpy(&src, c->d_l0_ptr + slot, sizeof(void *), hipMemcpyDeviceToHost));
            HV_HIP(c, hipMemcpy(&stride, c->d_l0_stride + slot, sizeof(int), hipMemcpyDeviceToHost));
            if (!src) return HV_ERR_INVALID;
        }
        HV_HIP(c, hipMemcpy2D(gray, L.w[level], src, stride, L.w[level], L.h[level], hipMemcpyDeviceToHost));
    }
    if (grad && level < L.grad_from) {
        const int rc = hv::download_unstored_gradient(c, slot, level, grad);
        if (rc != HV_OK) return rc;
    } else if (grad) {
        HV_HIP(c, hipMemcpy2D(grad, (size_t)L.w[level] * 4, base + L.doff[level], (size_t)L.dstride[level] * 4,
                              (size_t)L.w[level] * 4, L.h[level], hipMemcpyDeviceToHost));
    }
    if (grad) {
        const size_t n = (size_t)L.w[level] * L.h[level] * 2;      // device stores 4*d (exact): undo
        for (size_t i = 0; i < n; ++i) grad[i] = (int16_t)(grad[i] >> hv::GRAD_SHIFT);
    }
    return HV_OK;
}

/* ---- Lucas-Kanade ---- */

int hv_klt_track(hv_ctx *h, int prev_slot, int next_slot, int n, const float *prev_xy, float *next_xy,
                 uint8_t *status, float *err, int use_initial_flow, int max_iter_override)
{
    if (!h || n < 0) return HV_ERR_INVALID;
    if (n == 0) return HV_OK;
    if (!prev_xy || !next_xy || !status) return HV_ERR_INVALID;
    Ctx *c = &h->c;
    if (!hv::slot_ok(c, prev_slot) || !hv::slot_ok(c, next_slot)) return HV_ERR_POOL;
    int rc = hv::ensure_point_staging(c, n);
    if (rc != HV_OK) return rc;
    HV_HIP(c, hipMemcpyAsync(c->d_prev_xy, prev_xy, sizeof(float) * 2 * n, hipMemcpyHostToDevice, c->stream));
    if (use_initial_flow)
        HV_HIP(c, hipMemcpyAsync(c->d_next_xy, next_xy, sizeof(float) * 2 * n, hipMemcpyHostToDevice, c->stream));
    hv::IntPack pk{{prev_slot, next_slot, 0, 0}};
    hipLaunchKernelGGL(hv::set_ints_kernel, dim3(1), dim3(64), 0, c->stream, c->d_slots + 2, pk, 2);
    HV_HIP(c, hipGetLastError());
    const int iters = max_iter_override > 0 ? max_iter_override : c->p.max_iter;
    rc = hv::launch_klt(c, 1, c->d_slots + 2, c->d_slots + 3, n, n, c->d_prev_xy, c->d_next_xy,
                        c->d_status, err ? c->d_err : nullptr, use_initial_flow, iters);
    if (rc != HV_OK) return rc;
    HV_HIP(c, hipMemcpyAsync(next_xy, c->d_next_xy, sizeof(float) * 2 * n, hipMemcpyDeviceToHost, c->stream));
    HV_HIP(c, hipMemcpyAsync(status, c->d_status, n, hipMemcpyDeviceToHost, c->stream));
    if (err) HV_HIP(c, hipMemcpyAsync(err, c->d_err, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
    HV_HIP(c, hipStreamSynchronize(c->stream));
    return HV_OK;
}

int hv_optical_flow_compute(hv_ctx *h, int prev_slot, int cur_slot, int n, const float *prev_corners,
                            float *corners, int32_t *track_status, int use_initial_corners,
                            int override_max_iterations)
{
    if (!h || n < 0) return HV_ERR_INVALID;
    if (n == 0) return HV_OK;   // optical_flow.cpp:37-40
    if (!prev_corners || !corners || !track_status) return HV_ERR_INVALID;
    std::vector<uint8_t> st((size_t)n);
    const int rc = hv_klt_track(h, prev_slot, cur_slot, n, prev_corners, corners, st.data(), nullptr,
                                use_initial_corners, override_max_iterations);
    if (rc != HV_OK) return rc;
    const float width = (float)h->c.L.w[0], height = (float)h->c.L.h[0];
    for (int i = 0; i < n; ++i) {   // optical_flow.cpp:52-58
        const float x = corners[2 * i], y = corners[2 * i + 1];
        track_status[i] = st[i] == 0 ? 2 /*FAILED_FLOW*/ : 0 /*TRACKED*/;
        if (x < 0.0f || x >= width || y < 0.0f || y >= height) track_status[i] = 4 /*FLOW_OUT_OF_RANGE*/;
    }
    return HV_OK;
}

int hv_klt_track_batch_dev(hv_ctx *h, int n_pairs, const int *prev_slots_dev, const int *next_slots_dev,
                           int pts_per_pair, const float *prev_xy_dev, float *next_xy_dev,
                           uint8_t *status_dev, float *err_dev, int use_initial_flow, int max_iter_override)
{
    if (!h || n_pairs < 0 || pts_per_pair < 0) return HV_ERR_INVALID;
    if (n_pairs == 0 || pts_per_pair == 0) return HV_OK;
    if (!prev_slots_dev || !next_slots_dev || !prev_xy_dev || !next_xy_dev || !status_dev)   /* err_dev may be NULL */
        return HV_ERR_INVALID;
    Ctx *c = &h->c;
    const int iters = max_iter_override > 0 ? max_iter_override : c->p.max_iter;
    return hv::launch_klt(c, n_pairs, prev_slots_dev, next_slots_dev, pts_per_pair, n_pairs * pts_per_pair,
                          prev_xy_dev, next_xy_dev, status_dev, err_dev, use_initial_flow, iters);
}

int hv_klt_track_batch_ragged_dev(hv_ctx *h, int n_pairs, const int *prev_slots_dev, const int *next_slots_dev,
                                  int pts_per_pair, const int *pts_in_pair_dev, const float *prev_xy_dev, float *next_xy_dev,
                                  uint8_t *status_dev, float *err_dev, int use_initial_flow, int max_iter_override)
{
    if (!h || n_pairs < 0 || pts_per_pair < 0) return HV_ERR_INVALID;
    if (n_pairs == 0 || pts_per_pair == 0) return HV_OK;
    if (!prev_slots_dev || !next_slots_dev || !prev_xy_dev || !next_xy_dev || !status_dev || !pts_in_pair_dev) return HV_ERR_INVALID;
    Ctx *c = &h->c;
    const int iters = max_iter_override > 0 ? max_iter_override : c->p.max_iter;
    return hv::launch_klt(c, n_pairs, prev_slots_dev, next_slots_dev, pts_per_pair, n_pairs * pts_per_pair,
                          prev_xy_dev, next_xy_dev, status_dev, err_dev, use_initial_flow, iters, pts_in_pair_dev);
}

/* ---- timers ---- */

int hv_profile_enable(hv_ctx *h, int on)
{
    if (!h) return HV_ERR_INVALID;
    h->c.profiling = on != 0;
    return HV_OK;
}

static int drain_timers(Ctx *c)
{
    HV_HIP(c, hipStreamSynchronize(c->stream));
    for (auto &t : c->timers) {
        for (auto &e : t.pending) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) { t.total_ms += ms; t.launches++; }
            t.free_list.push_back(e);
        }
        t.pending.clear();
    }
    return HV_OK;
}

int hv_profile_reset(hv_ctx *h)
{
    if (!h) return HV_ERR_INVALID;
    const int rc = drain_timers(&h->c);
    for (auto &t : h->c.timers) { t.total_ms = 0.0; t.launches = 0; }
    return rc;
}

int hv_profile_read(hv_ctx *h, int kernel_id, double *total_ms, long long *launches)
{
    if (!h || kernel_id < 0 || kernel_id >= HV_K_COUNT) return HV_ERR_INVALID;
    const int rc = drain_timers(&h->c);
    if (total_ms) *total_ms = h->c.timers[kernel_id].total_ms;
    if (launches) *launches = h->c.timers[kernel_id].launches;
    return rc;
}

}  // extern "C"
