// Internal declarations shared by the HIP translation units of libhybvio_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/hybvio_hip.h"

namespace hv {

// Device layout of one pyramid slot (passed by value to kernels).
//   gray level l >= 1 : u8, row stride gstride[l] bytes (multiple of 16), at goff[l]
//   gray level 0      : either the slot's own copy at goff[0] (host build path) or the
//                       caller's image used in place (batch_dev path); the per-slot pointer
//                       table (l0_ptr / l0_stride) says which.
//   gradient level l  : one dword per pixel = int16 4*dx+2 | int16 (4*dy+2) << 16, row stride
//                       dstride[l] dwords (multiple of 4), at doff[l]
//   levels l >= pad_from (default 2; never level 0) carry a PHYSICAL border of pad[l] = 32 pixels on every side, the one
//   OpenCV's padded pyramid holds: BORDER_REFLECT_101 for gray (pyr_border_kernel, per build), the constant 0 -- stored as
//   4*0+2 -- for gradients (written once per slot). goff / doff still address pixel (0, 0); strides include the border.
//   At the coarse levels most LK windows cross the image edge (68 % at level 3 of 752x480); with the border in memory the
//   tracker takes its border-free paths there. Levels 0 and 1 (94 % of the pyramid bytes) keep virtual borders.
//   The gradient planes of the fine levels l < grad_from (default 2: levels 0 and 1, 88 % of a slot's bytes at 752x480) are
//   NOT materialised: the LK kernel forms the Scharr gradients of its template window from the gray rows it loads anyway
//   (klt.hip), so the pyramid build never writes them (doff[l] == -1). HV_GRAD_FROM_LEVEL=0 restores every plane, =1 only
//   level 0 is formed in the LK kernel (experiments / A-B measurements). Padded levels always store their plane (its border
//   is the constant 0, which the gray border cannot reproduce).
struct PyrLayout {
    int levels;
    int win;
    int grad_from;                         // first level whose gradient plane exists in the slot
    int pad[HV_MAX_LEVELS];
    int w[HV_MAX_LEVELS], h[HV_MAX_LEVELS];
    int gstride[HV_MAX_LEVELS];
    int dstride[HV_MAX_LEVELS];
    long long goff[HV_MAX_LEVELS];
    long long doff[HV_MAX_LEVELS];
    long long slot_bytes;
};

// Gradients are stored as int16 ((d << GRAD_SHIFT) + 2) for dx and dy (see pyramid.hip / klt.hip).
constexpr int GRAD_SHIFT = 2;

struct KernelTimer {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> free_list;
    double total_ms = 0.0;
    long long launches = 0;
};

// Kernel-variant selectors ("knobs"). Production code never needs them: every default picks the variant by shape / batch size.
// They exist so that tests and A/B measurements can FORCE each variant at any batch size. Read from the environment ONCE, in
// hv_create (variable = "HV_" + upper-case name), and changed afterwards only through hv_debug_set_knob: the hot path never calls
// getenv (r02 advisor), and a HIP graph embeds whatever was selected when it was captured.
struct Knobs {
    int pyr_tail = -1;            // HV_PYR_TAIL: -1 auto (levels >= 2 in one launch from 64 images up), 0 off, 1 on
    int pyr_l0_tiled = 0;         // HV_PYR_L0_TILED: 1 = pure down-sample levels through the LDS-tile kernel
    int gftt_tiled = -1;          // HV_GFTT_TILED: -1 auto (marching kernel from 128 images up), 1 tiled, 0 marching
    int klt_tile = 5;             // HV_KLT_TILE: staged J tile / occupancy of the LK kernel (klt.hip)
    int vu_threads = 0;           // HV_VU_THREADS: 0 auto, 384 / 768 force the two-per-CU / the latency build of vu_prepare
    int ekf_spec_split = 0;       // HV_EKF_SPEC_SPLIT: legacy two-launch form of a speculative pass on the DENSE kernels
    int ekf_no_speculation = 0;   // HV_EKF_NO_SPECULATION: 1 = always the sequential visit loop
    int ekf_stream_gate = -1;     // HV_EKF_STREAM_GATE: -1 auto, 0 never, 1 also inside the visit loop
    int ekf_gate_kmode = -1;      // HV_EKF_GATE_KMODE: 1 = gate-only launches on the H-from-L2 update kernel
    int ingest_gather = 0;        // HV_INGEST_GATHER: 1 = plain gather kernel for the remap
    int ekf_fused_gate = -1;      // HV_EKF_FUSED_GATE: column-sparse chi2 gate: -1 auto = 1 inside the prepare kernel, 2 own launch (ekf_sparse_gate_kernel), 0 off (dense kernels)
    int ekf_spec_mode = -1;       // HV_EKF_SPEC_MODE: speculative pass form: -1 auto, 2 = gate launch + apply launch, 3 = one launch with hand-shake
    int ekf_side_stream = 6;      // HV_EKF_SIDE_STREAM, ragged visits with two length classes inside a frame loop over more filters than CUs (needs the per-frame sort of launch_visit_order): 0 = one stream; 6 (default of hv_create) = the visit forks: long class's prepare + gate launch on the context stream, short class's fused launch on the context's second stream, joined in front of the update launches (+6 % on the realistic C3 step against 0); 3 = r03's arrangement of the fork (long class on the second stream, enqueued first: 3 % slower than 6 with eager launches, equal under graph replay); 5 (default of the contexts of an hv_lanes set) = 6, but never inside a stream capture (a captured fork is replayed on a default-priority stream of the graph instance, not on the lane's own: ekf.hip). r03's other forms (1, 2, 4: whole long chain on the second stream / enqueued behind) measured slower and were removed in r04
    int ekf_long_fused = 1;       // HV_EKF_LONG_FUSED: 1 (r04 default) = prepare + column-sparse gate of the long class (49 .. 84 rows) in ONE launch (vu_gate_long_kernel); 0 = r03's vu_compact_kernel + ekf_sparse_gate_big_kernel
    int ekf_predict_chain = 1;    // HV_EKF_PREDICT_CHAIN: 1 (late r06) = launches of >= 3 IMU samples run ekf_predict_chain_kernel (the samples' mean recursion on one wavefront without workgroup barriers, dR / F / L of up to five samples per pass, then the 20 x 20 recursions); 2 = every launch; 0 = ekf_predict_kernel (nine barrier-separated stages per sample). Bit-identical
    int ekf_short_np = 12;        // HV_EKF_SHORT_NP: longest stereo track of the short class where the split form serves the visit: 12 (48 rows, r06 default) or 11 (the fused builds' boundary, r03 .. r05)
    int ekf_long_first = 0;       // HV_EKF_LONG_FIRST: launch order of a ONE-STREAM sorted ragged visit (captured lanes): 0 (r06 default) = short class (triangulation, gate) then long class; 1 = long class first (r04 / r05 default: the fused long launch needed whole CUs); 2 .. 4 = both triangulations in front of both gates (T long, T short, G long, G short / T short, T long, G short, G long / T long, T short, G short, G long). Four lanes: 27.37 ms per step with 0 against 27.57 with 1, the others in between (profiles/r06/visit_launch_order_sweep.txt)
    int ekf_dual_update = 1;      // HV_EKF_DUAL_UPDATE: 1 = ragged visits issue the short class's update and the long class's first block update as one grid (ekf_update_dual_kernel); 0 = one after the other
    int ekf_visit_order = 1;      // HV_EKF_VISIT_ORDER: 1 = ragged frame loops over more filters than the GPU has CUs hand the fused kernel its records longest track first (one sort launch per frame; the presorted long-class lists also enable ekf_side_stream 2 .. 4); 2 = at every batch size (tests); 0 = in filter order
    int ekf_defer_jacobian = 1;   // HV_EKF_DEFER_JACOBIAN: 1 (r05) = the long class's one-launch build (vu_gate_long_kernel) forms and stores the compact Jacobian Hc = Dp + O4 F4 BEHIND its gate, for inliers only (the update is its only reader); 0 = for every prepared track, in front of the gate (r04)
    int ekf_split_tri = 1;        // HV_EKF_SPLIT_TRI: 1 (r06) = ragged two-class visits run the triangulation front as its own launch (vu_tri_kernel: a wavefront per short track, four per long track) and the gates from its factor records (three short-class gates per CU instead of two fused workgroups); 0 = r03 .. r05's fused prepare + gate kernels
    int vu_tri_threads = 0;       // HV_VU_TRI_THREADS: threads per track of vu_tri_kernel: 0 auto (128 for the short class, 256 for the long class), 64 / 128 / 256 force
    int rot_ransac_threads = 0;   // HV_ROT_RANSAC_THREADS: 0 auto (25 workgroups per set while they all get a CU, 1024 threads up to 64 sets, 256 beyond), 25 / 256 / 1024 force
};
int knob_set(Knobs &k, const char *name, int value);   // HV_ERR_INVALID for an unknown name
int knob_get(const Knobs &k, const char *name, int *value);
void knobs_from_env(Knobs &k);

struct Ctx {
    hv_params p{};
    Knobs knob{};
    PyrLayout L{};
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // second stream of the context, library-owned, created WITH the context (r04; r03 created it lazily inside the first ragged visit,
    // so which hardware queue it landed on depended on what the process had done by then): the long-track class of a ragged visit
    // runs its prepare + gate launches on it beside the short class's launch on `stream` (ekf.hip), joined again inside the visit
    hipStream_t aux_stream = nullptr;
    int stream_priority = 0;              // 0: default priority; 1: both streams from the device's HIGH-priority queue pool (hv_lanes_create)
    int num_cus = 256;                    // multiProcessorCount of the device
    uint8_t *slab = nullptr;              // pool_size * slot_bytes
    const uint8_t **d_l0_ptr = nullptr;   // [pool_size]
    int *d_l0_stride = nullptr;           // [pool_size]
    std::vector<int> free_slots;
    std::vector<uint8_t> slot_used;
    // small staging buffers for the synchronous host-pointer entry points
    int *d_slots = nullptr;               // [2 * max_pairs]
    float *d_prev_xy = nullptr, *d_next_xy = nullptr, *d_err = nullptr;
    uint8_t *d_status = nullptr;
    int stage_points = 0;
    float *d_gftt_kp = nullptr;           // key points of the single-image detector entry point
    int gftt_cap = 0;
    // ingest (f2): per-camera remap tables and the staging buffer of the host entry point
    uint32_t *d_map_xy[HV_INGEST_CAMERAS] = {};
    float *d_map_xf[HV_INGEST_CAMERAS] = {}, *d_map_yf[HV_INGEST_CAMERAS] = {};
    int map_stride = 0;
    int *d_tile_box[HV_INGEST_CAMERAS] = {};   // per 64 x 16 output tile: source rectangle (remap_tile_kernel)
    bool map_tiled[HV_INGEST_CAMERAS] = {};
    uint8_t *d_ingest_stage = nullptr;
    size_t ingest_stage_bytes = 0;
    unsigned char *d_ransac_stage = nullptr;   // staging of the host-pointer rotation-RANSAC entry (f4)
    size_t ransac_stage_bytes = 0;
    unsigned char *d_ransac_split = nullptr;   // records of the split rotation-RANSAC launches (rot_ransac.hip), ransac_split_sets of them
    int ransac_split_sets = 0;
    std::string last_error;
    bool profiling = false;
    KernelTimer timers[HV_K_COUNT];
};

int hip_fail(Ctx *c, hipError_t e, const char *what);
Ctx *ctx_of(hv_ctx *h);
#define HV_HIP(c, call)                                                  \
    do {                                                                 \
        hipError_t e__ = (call);                                         \
        if (e__ != hipSuccess) return hv::hip_fail((c), e__, #call);     \
    } while (0)

// RAII-ish per-launch timing helper (no-op unless profiling is on).
struct ScopedKernelTime {
    Ctx *c; int id; hipEvent_t a = nullptr, b = nullptr; hipStream_t s = nullptr;
    ScopedKernelTime(Ctx *c, int id, hipStream_t stream = nullptr);   // stream: the one the kernel is launched on (null = the context's)
    ~ScopedKernelTime();
};

// pyramid.hip
constexpr int PYR_PAD = 32;            // physical border width of the padded levels (window 31 + the pair column)
int fill_gradient_borders(Ctx *c, int first_slot, int n_slots);   // once per slot (create / grow)
int download_unstored_gradient(Ctx *c, int slot, int level, int16_t *grad);   // computed on demand (test / debug read-back)
int launch_pyramid_levels(Ctx *c, int n, const int *slots_dev, const uint8_t *src_base,
                          long long src_step, int src_stride, bool src_indexed_by_slot);
// vu_prepare.hip: triangulation + prepareVisualUpdate of one track per filter
struct VuPrepareArgs {
    int batch, n, np, stereo;          // filters, state dimension, poses of the track, two cameras per pose
    const double *m;                   // [batch][n] means
    const int *pose_index;             // [batch][np] poseTrailIndex (0 = current pose, i = trail slot i - 1)
    const double *features, *velocities;   // [batch][ncam * np][2]: first camera's poses, then the second's
    const double *y;                   // [batch][2 * ncam * np] measured pixels (NULL: v = -f)
    double imu_to_cam[2][12];          // per camera the top 3 x 4 of imuToCamera, row-major
    double conv_threshold, conv_r, rcond_threshold, min_dist, max_dist;
    int gn_iters, est_shift;
    int defer_h;                  // Knobs::ekf_defer_jacobian (STRUCT builds only)
    int linear;                        // useLinearTriangulation: the closed-form branch instead of two-camera + Gauss-Newton
    // ragged batches: per-record pose counts (np above is then the record stride = the longest track; < 2: no track for this
    // record) and, as an output for the gate / update launch, the per-record row counts 2 * cameras * poses (0: none)
    const int *np_rec;
    int *rows_out;
    // hybrid-map tracks (mapPointUpdate, backend.cpp:1016,1075-1082; r04): map_index [records] = the map point of the track (>= 0) or -1
    // for a pose-trail track; the point is then the state m[map_base + 3 idx ..] instead of a triangulation (status HV_TRI_HYBRID), H gets
    // dip R in its three columns and no point-derivative terms (triangulation.cpp:968-985). Dense H only (fused = 0).
    const int *map_index; int map_base;
    // length classes of a ragged launch (r03): only records with np_lo <= poses <= np_hi are processed (0, 0: all of them); the others
    // return at once and leave every output alone -- except `active`, cleared when class_inactive is set (a second launch sequence with
    // other kernels serves them in the same visit: short tracks on the two-per-CU fused kernels, long ones on the dense kernels)
    int np_lo, np_hi, class_inactive;
    // compaction lists of a visit (r03): a masked launch of big-LDS workgroups loses half its time to the inactive ones queueing for CU
    // slots, so the kernels that find out which records need the NEXT launch append them to a list; that launch maps workgroup i to
    // list[i] and the rest exit at once. counts are zeroed per visit by the host entry (one memset node).
    int *inl_count, *inl_list;         // appended by the fused gate: records whose gate said INLIER (-> update launch)
    int *long_count, *long_list;       // appended by the short class's launch: records of the long class (-> its prepare / gate launches)
    const int *rec_count, *rec_list;   // this launch's own records (long-class prepare): workgroup i handles rec_list[i], i < *rec_count
    const int *order;                  // fused two-per-CU launches: workgroup i serves filter order[i] (a permutation of the batch, longest tracks first: launch_visit_order), or null
    double *H, *v, *f, *pf;            // [batch][rows * n] column-major, [batch][rows], optional [batch][rows], [batch][3]
    int *status;                       // [batch][2]: TriangulatorStatus, PrepareVuStatus
    unsigned char *active;             // optional [batch]: 1 where both are OK
    int *gate_status;                  // optional [batch]: preset to VuOutlierStatus::NOT_COMPUTED (1)
    const int *success_counter;        // optional [batch]: filters that already applied max_successful updates this frame are skipped
    int max_successful;
    // speculative frame loop (ekf.hip, hv_ekf_visual_frame_dev): grid (batch, spec_tracks), every array a [track][filter] record.
    // A record is prepared when its track is pending (>= cursor[filter]) and was last prepared at another update count (epoch).
    int spec_tracks;
    const int *cursor;                 // [batch]
    int *epoch;                        // [spec_tracks][batch]
    // fused column-sparse chi2 gate (vu_gate kernels, r03): H is non-zero only in the 7 columns of each pose of the track and in the
    // time-shift column (triangulation.cpp:908-921,958-980), so the kernel keeps the COMPACT Jacobian Hc (rows x na, na = 7 poses + 1)
    // in LDS, gathers P_aa = P(acol, acol) and evaluates visualTrackOutlierCheck (ekf.cpp:787-819) right there: S = Hc P_aa Hc' + R,
    // blocked Cholesky, chi2. The dense H is not written; Hc and acol go to HBM for the update of an inlier.
    int fused;
    double *Hc;                        // [records][rows_max * na_max], column u of a record = state column acol[u], leading dimension = the record's rows
    int *acol;                         // [records][na_max]
    int na_max;                        // 7 * np + 1
    const double *P;                   // [batch][n][n]
    double rd_gate, noise_scale;       // R = rd_gate I (already scaled by noiseScale), chi2 = noise_scale z'z
    // adaptive thresholds of the frame loop (backend.cpp:994-996,1192-1193; fused gates only): gate_scale[filter] multiplies
    // trackChiTestOutlierR and trackRmseThreshold (null: 1); a filter whose track fails the RMSE or the chi2 test multiplies its entry
    // by `growth`. rmse_thr < 0: no RMSE test (ekf.cpp:797-801)
    double *gate_scale; double growth, rmse_thr;
    double *chi2;                      // optional [records]
    // split form (r06, knob ekf_split_tri): vu_tri_kernel -- one wavefront (short class) or four (long class) per track -- runs the
    // triangulation and the per-pose part of prepareVisualUpdate and leaves a FACTOR RECORD per track in HBM / L2 (tri_rec, tri_stride
    // doubles per record: [nt_max][17] per-pose values, [np][21] summed point derivatives, the time-shift column, prep status); the
    // gate kernels launched with from_rec = 1 start from that record instead of running the front themselves, in a third of the LDS
    double *tri_rec; int tri_stride; int from_rec;
};
// doubles per factor record of vu_tri_kernel for tracks of up to np poses on ncam cameras
inline int vu_tri_rec_stride(int np, int ncam) { return 17 * np * ncam + 21 * np + 4; }
int rot_ransac_alloc_split(Ctx *c);     // rot_ransac.hip: the split form's record buffer, allocated and zeroed once per context
int launch_vu_tri(Ctx *c, const VuPrepareArgs &a, hipStream_t stream = nullptr);       // the triangulation front of the split form
bool vu_split_supported(const Ctx *c, const VuPrepareArgs &a, int fused);              // shapes the record-fed gate builds serve
bool vu_split_short_ok(const Ctx *c, int n_state, bool stereo, int batch, bool linear); // ... as far as a visit knows before its launches
int vu_split_short_np(const Ctx *c);                                                            // longest stereo track of the record-fed short-class gate (12)
int launch_vu_prepare(Ctx *c, const VuPrepareArgs &a, hipStream_t stream = nullptr);   // stream: null = the context's
// order[v][0 .. batch): the filters of visit v sorted by descending pose count among those with np_lo <= np_rec <= np_hi (the others last);
// long_list[v][0 .. long_count[v]) (optional): those with np_hi < np_rec <= np_max, longest first
int launch_visit_order(Ctx *c, int visits, int batch, const int *np_rec_dev, int np_lo, int np_hi, int np_max, int *order_dev,
                       int *long_list_dev, int *long_count_dev);
bool vu_fused_supported(const Ctx *c, int n_state, int np, int stereo, int batch);   // shapes the fused gate serves (else: dense path)
// capi.hip
int build_levels_of_slot(Ctx *c, int slot);
// klt.hip
int launch_klt(Ctx *c, int n_pairs, const int *prev_slots_dev, const int *next_slots_dev,
               int pts_per_pair, int n_points, const float *prev_xy, float *next_xy,
               uint8_t *status, float *err, int use_initial_flow, int max_iter, const int *pts_in_pair_dev = nullptr);

// XCD-aware block remap (MI355X: 8 XCDs, block b is dispatched to XCD b % 8): gives every XCD a
// contiguous range of logical tiles so neighbouring tiles share that XCD's L2. Bijective for
// any grid size (cdna_hip_programming.md T1).
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg)
{
    const unsigned xcd = bid & 7u, idx = bid >> 3;
    const unsigned q = nwg >> 3, r = nwg & 7u;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

__device__ __forceinline__ int reflect101(int p, int len)
{
    // OpenCV borderInterpolate(BORDER_REFLECT_101). Exact for -len < p < 2*len-1, which covers
    // the 31-px virtual border of every level (levels are > 31 px). The clamp only affects
    // staging margins that no LK window can touch; it keeps those reads in bounds.
    if (p < 0) p = -p;
    if (p >= len) p = 2 * len - 2 - p;
    return min(max(p, 0), len - 1);
}

}  // namespace hv
