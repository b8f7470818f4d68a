/*
 * hybvio_hip.h -- C ABI of libhybvio_hip.so: the MI355X (gfx950) implementation of HybVIO's
 * per-frame hot path (image pyramid + pyramidal Lucas-Kanade tracker + EKF covariance algebra) and of
 * its GFTT feature detector.
 *
 * Plain pointers and sizes only; no C++/torch types cross this boundary. Every entry point
 * returns HV_OK (0) or a negative hv_status; nothing throws. All device work of one hv_ctx is
 * issued on one HIP stream (the reference runs Tracker::add and every EKF method on a single
 * thread: src/api/api.cpp:82,425; src/util/bounded_processing_queue.hpp:18).
 *
 * Each group cites the reference interface (file:line under the HybVIO tree) that it replaces;
 * INTEGRATION.md shows the C++ adapters a maintainer adds on the reference side.
 *
 * Names ending in _dev take DEVICE pointers and are asynchronous on the context stream
 * (throughput / multi-session mode: many independent sequences per GPU in one launch).
 * All other entry points take HOST pointers and are synchronous on return unless stated.
 */
#ifndef HYBVIO_HIP_H_
#define HYBVIO_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HV_ABI_VERSION 4   /* 4 (r05): host-pointer forms of the r04 frame entries (hv_ekf_visual_frame_batch, hv_ekf_visual_track_hybrid, hv_ekf_symmetrize_augment), hv_ekf_insert_map_point / hv_ekf_get_map_point; 3 (r04): hv_lanes_*, hv_get_stream; 2 (r03): hv_debug_*_knob, hv_ekf_frame_error, HV_ERR_TIMEOUT; maxSuccessfulVisualUpdates <= 0 = no limit */
#define HV_MAX_LEVELS 6

typedef enum hv_status {
    HV_OK = 0,
    HV_ERR_INVALID = -1,      /* bad argument                                  */
    HV_ERR_UNSUPPORTED = -2,  /* parameter combination not implemented          */
    HV_ERR_NO_DEVICE = -3,    /* no HIP device / HIP runtime failed to start    */
    HV_ERR_HIP = -4,          /* a HIP call failed; see hv_last_error()         */
    HV_ERR_POOL = -5,         /* pyramid pool exhausted / bad slot              */
    HV_ERR_NOMEM = -6,
    HV_ERR_TIMEOUT = -7       /* a device-side wait gave up; the call's result must not be used */
} hv_status;

typedef struct hv_ctx hv_ctx;

/* Mirror of the few odometry::ParametersTracker fields the path reads
 * (codegen/parameter_definitions.c:262,336-344). */
typedef struct hv_params {
    int device;            /* HIP device ordinal                                         */
    int width, height;     /* level-0 image size (all images of a context share it)      */
    int levels;            /* pyrLKMaxLevel + 1                        (default 3 + 1)   */
    int win;               /* pyrLKWindowSize                          (default 31)      */
    int max_iter;          /* pyrLKMaxIter                             (default 20)      */
    double eps;            /* pyrLKEpsilon                             (default 0.03)    */
    double min_eig;        /* pyrLKMinEigThreshold                     (default 1e-3)    */
    int max_tracks;        /* tracker.maxTracks: points per LK call    (default 200)     */
    int pool_size;         /* pyramid slots kept on the device (util::Allocator analogue)*/
    int max_pairs;         /* max (prev,next) pyramid pairs per batched LK launch        */
} hv_params;

void hv_default_params(hv_params *p);                 /* fills the defaults listed above */
int hv_abi_version(void);
const char *hv_status_string(int status);
/* Test / measurement hooks: force a kernel variant that production code selects by shape or batch size (names: the fields of
 * hv::Knobs in csrc/hv_internal.hpp, e.g. "vu_threads" 384 / 768, "pyr_tail" 0 / 1, "rot_ransac_threads" 256 / 1024, "ekf_fused_gate"
 * 0). The same names in upper case with an HV_ prefix are read from the environment ONCE, by hv_create; nothing on the hot path
 * calls getenv. HV_ERR_INVALID for an unknown name. Not needed by an integration. */
int hv_debug_set_knob(hv_ctx *ctx, const char *name, int value);
int hv_debug_get_knob(hv_ctx *ctx, const char *name, int *value);

/* ---- context ---------------------------------------------------------------------------- */
int hv_create(const hv_params *params, hv_ctx **out);
void hv_destroy(hv_ctx *ctx);
const char *hv_last_error(hv_ctx *ctx);               /* text of the last HV_ERR_HIP           */
int hv_set_stream(hv_ctx *ctx, void *hip_stream);     /* run on a caller-owned hipStream_t     */
void *hv_get_stream(hv_ctx *ctx);                     /* the hipStream_t the context issues on  */
int hv_synchronize(hv_ctx *ctx);

/* ---- lanes: several batched contexts of one process on one GPU ------------------------------
 * The reference runs one session per process (main.cpp); a GPU serves many. One batched context advances its B resident
 * sequences through a frame as a chain of dependent launches -- a VALU-bound tracker half, then a latency-bound EKF half whose
 * launches fill a fraction of the chip -- so TWO contexts whose chains run beside each other deliver more frames per second
 * than one context alone (107 k -> 130 k frames/s at 2 x 1024 sequences, 138 k at 4 x 1024). Whether their
 * launches really overlap is decided by the HARDWARE QUEUE each busy stream is bound to, and for default-priority streams that
 * depends on everything the process created before (r03 needed a particular creation order in the caller). A lane set owns that
 * placement: hv_lanes_create creates n_lanes contexts (same parameters) whose two streams each -- the context stream and the
 * library's second stream of the ragged visit loop -- come from the device's HIGH-priority queue pool, which holds a queue per
 * stream for up to two lanes regardless of process history. Drive every lane from its own host thread or round-robin from one;
 * the lanes share nothing but the device. hv_lanes_ctx returns a lane's context (owned by the set: do not hv_destroy it, do not
 * hv_set_stream it); hv_get_stream gives the stream to enqueue caller-side work (or a HIP-graph capture) on. */
#define HV_MAX_LANES 8
typedef struct hv_lanes hv_lanes;
int hv_lanes_create(const hv_params *params, int n_lanes, hv_lanes **out);
int hv_lanes_count(const hv_lanes *lanes);
hv_ctx *hv_lanes_ctx(hv_lanes *lanes, int lane);
void hv_lanes_destroy(hv_lanes *lanes);

/* ---- image pyramid ----------------------------------------------------------------------
 * Replaces tracker::ImagePyramid::Factory::compute (src/tracker/image_pyramid.hpp:35-41,
 * image_pyramid.cpp:40-48 -> cv::buildOpticalFlowPyramid(img, pyr, Size(win,win), maxLevel)).
 * A slot is one image's pyramid: gray levels 1..L-1 (u8, 5x5 binomial, (s+128)>>8) and Scharr
 * gradients (int16 [dx,dy] interleaved, scale 1/32: image_pyramid.hpp:19-26). Only the coarse levels (>= 2) keep
 * their gradient plane in memory; those of levels 0 and 1 are formed inside the tracker kernel from the gray
 * rows -- hv_pyramid_download still returns every level's gradients (computed on demand).
 * Levels 0 and 1 are stored without a border (applied virtually by the tracker: BORDER_REFLECT_101 for gray,
 * 0 for gradients, exactly as OpenCV pads); levels >= 2 carry OpenCV's border physically.
 * Slots are pooled like util::Allocator (src/util/allocator.hpp:55-67): acquire when an Image
 * first needs its pyramid (image.cpp:209-214), release when the Image dies. Like the reference's
 * allocator the pool grows on demand: when every slot is in use hv_pyramid_acquire doubles the slab
 * (synchronises the stream; slot numbers and contents are kept; HIP graphs captured earlier hold stale
 * addresses) and only reports HV_ERR_POOL when the device is out of memory. pool_size is the initial size. */
int hv_pyramid_acquire(hv_ctx *ctx, int *slot_out);
int hv_pyramid_release(hv_ctx *ctx, int slot);
int hv_pyramid_level_size(hv_ctx *ctx, int level, int *width, int *height);
/* H2D copy of a host gray image (stride_bytes per row) + all level kernels; asynchronous. */
int hv_pyramid_build(hv_ctx *ctx, int slot, const uint8_t *gray_host, int stride_bytes);
/* n images already resident in HBM: image i starts at gray_dev + i*image_stride_bytes and
 * is used in place as level 0 (it must stay valid until the slot is released/rebuilt).
 * slots_dev: n slot indices in device memory. Asynchronous. */
int hv_pyramid_build_batch_dev(hv_ctx *ctx, int n, const int *slots_dev, const uint8_t *gray_dev,
                               long long image_stride_bytes, int row_stride_bytes);
/* Test/debug read-back of one level (interior only): gray w*h u8 and/or grad w*h*2 int16;
 * either pointer may be NULL. Synchronous. (ImagePyramid::getGrayLevel/getGradientLevel/getOpenCv) */
int hv_pyramid_download(hv_ctx *ctx, int slot, int level, uint8_t *gray, int16_t *grad);

/* ---- pyramidal Lucas-Kanade --------------------------------------------------------------
 * hv_klt_track replaces cv::calcOpticalFlowPyrLK as called at src/tracker/optical_flow.cpp:46-49
 * (window win x win, levels, TermCriteria(COUNT|EPS, max_iter, eps), minEigThreshold, err != NULL).
 * next_xy is in/out: initial guess when use_initial_flow != 0 (OPTFLOW_USE_INITIAL_FLOW).
 * status: 1 tracked / 0 lost. max_iter_override <= 0 keeps the context value. err may be NULL (the
 * reference discards it: optical_flow.cpp:26); the level-0 range check that can clear status runs
 * either way. Synchronous. */
int hv_klt_track(hv_ctx *ctx, int prev_slot, int next_slot, int n, const float *prev_xy,
                 float *next_xy, uint8_t *status, float *err, int use_initial_flow,
                 int max_iter_override);
/* Replaces tracker::OpticalFlow::compute (src/tracker/optical_flow.hpp:31-38,
 * optical_flow.cpp:10-59,78-102): runs LK, maps to Feature::Status (track.hpp:9-21:
 * TRACKED=0, FAILED_FLOW=2) and overrides to FLOW_OUT_OF_RANGE=4 outside the level-0 image.
 * corners is in/out (input when use_initial_corners). n == 0 is a no-op. Synchronous. */
int hv_optical_flow_compute(hv_ctx *ctx, int prev_slot, int cur_slot, int n,
                            const float *prev_corners, float *corners, int32_t *track_status,
                            int use_initial_corners, int override_max_iterations);
/* Batched: n_pairs independent (prev,next) pyramid pairs with pts_per_pair points each, one
 * launch; point j of pair p is element p*pts_per_pair + j of every array. All arrays are in
 * device memory (err_dev may be NULL); asynchronous. */
int hv_klt_track_batch_dev(hv_ctx *ctx, int n_pairs, const int *prev_slots_dev,
                           const int *next_slots_dev, int pts_per_pair, const float *prev_xy_dev,
                           float *next_xy_dev, uint8_t *status_dev, float *err_dev,
                           int use_initial_flow, int max_iter_override);
/* The same for pairs with different point counts (sequences of a batch do not track the same number of features): pair p
 * holds pts_in_pair_dev[p] <= pts_per_pair points at the start of its record; the padding behind them falls through at once
 * (status 0, next_xy unspecified). */
int hv_klt_track_batch_ragged_dev(hv_ctx *ctx, int n_pairs, const int *prev_slots_dev, const int *next_slots_dev,
                                  int pts_per_pair, const int *pts_in_pair_dev, const float *prev_xy_dev, float *next_xy_dev,
                                  uint8_t *status_dev, float *err_dev, int use_initial_flow, int max_iter_override);


/* ---- EKF covariance algebra --------------------------------------------------------------
 * Replaces the dense work of odometry::EKF (src/odometry/ekf.hpp:62-174, ekf.cpp) for a BATCH of
 * independent filters that share one parameter set (batch = 1 is the reference's single session).
 * State m (n = 20 + 7*cameraTrailLength + 3*hybridMapSize) and covariance P (n x n, f64,
 * column-major like Eigen) stay on the device; the scalar bookkeeping of the reference class
 * (sample clock, ZUPT rate limits, augmentTimes) lives in the host adapter (hybvio_amd/host).
 * Per-filter inputs are arrays indexed [filter][...]. Calls are asynchronous on the context
 * stream unless they return data to the host. */
typedef struct hv_ekf hv_ekf;

/* The odometry parameters EKFImplementation reads (codegen/parameter_definitions.c:68-160). */
typedef struct hv_ekf_params {
    int cameraTrailLength, hybridMapSize;
    double noiseScale, gravity, augmentR, initZuptR, rotationZuptR;
    double noiseInitialPos, noiseInitialOri, noiseInitialVel, noiseInitialPosTrail, noiseInitialOriTrail;
    double noiseInitialBGA, noiseInitialBAA, noiseInitialBAT, noiseInitialSFT;
    double noiseProcessAcc, noiseProcessGyro, noiseProcessBAA, noiseProcessBGA;
    double noiseProcessBAARev, noiseProcessBGARev;
} hv_ekf_params;

void hv_ekf_default_params(hv_ekf_params *p);
/* EKF::build (ekf.cpp:153-296, 1087-1092): initial m, P, Q as the reference constructor.
 * An hv_ekf must be destroyed before the hv_ctx it was created on. */
int hv_ekf_create(hv_ctx *ctx, const hv_ekf_params *params, int batch, hv_ekf **out);
void hv_ekf_destroy(hv_ekf *ekf);
int hv_ekf_state_dim(const hv_ekf *ekf);                         /* getStateDim            */
int hv_ekf_batch(const hv_ekf *ekf);
/* setState / setStateCovariance / getState / getStateCovariance (ekf.cpp:950-979); either
 * pointer may be NULL. Synchronous. Also the route for the rare mean-side edits the adapter does
 * on the host (lockBiases, conditionOnLastPose, insertMapPoint, setInertialState, translateTo). */
int hv_ekf_set_state(hv_ekf *ekf, int filter, const double *m, const double *P_colmajor);
int hv_ekf_get_state(hv_ekf *ekf, int filter, double *m, double *P_colmajor);
int hv_ekf_get_means(hv_ekf *ekf, double *m_all /* batch * n */);
int hv_ekf_set_process_noise(hv_ekf *ekf, int filter, const double *Q12x12);   /* setProcessNoise */
int hv_ekf_get_process_noise(hv_ekf *ekf, int filter, double *Q12x12);
int hv_ekf_get_dydx(hv_ekf *ekf, int filter, double *dydx20x20);               /* getDydx block   */
/* device addresses of the means (batch x n) and covariances (batch x n x n). hv_ekf_augment and
 * hv_ekf_undo_augment ping-pong between two covariance buffers: fetch P_dev again after either. */
int hv_ekf_device_pointers(hv_ekf *ekf, double **m_dev, double **P_dev);
/* predict (ekf.cpp:320-514): mean, Jacobians and P <- F P F' + L Q L' on the device. dt[f] <= 0
 * skips filter f. gyro/acc: [batch][3]. */
int hv_ekf_predict(hv_ekf *ekf, const double *dt, const double *gyro, const double *acc);
int hv_ekf_predict_dev(hv_ekf *ekf, const double *dt_dev, const double *gyro_dev, const double *acc_dev);
/* n_samples consecutive predicts (the IMU samples between two camera frames) in ONE launch: arrays are
 * [n_samples][batch] (dt) and [n_samples][batch][3]. The mean / Jacobian chain and the 20 x 20 block
 * run per sample on chip; the off-diagonal covariance blocks see the product of the samples' F once.
 * Equal to n_samples calls of hv_ekf_predict_dev up to rounding (1e-15 relative). */
int hv_ekf_predict_n_dev(hv_ekf *ekf, int n_samples, const double *dt_dev, const double *gyro_dev,
                         const double *acc_dev);
#define HV_EKF_MAX_PREDICT_SAMPLES 32
int hv_ekf_predict_n(hv_ekf *ekf, int n_samples, const double *dt, const double *gyro, const double *acc);   /* host arrays */
/* update(m,P,y,H,R,...) (ekf.cpp:57-82) with truncated H (n_rows x l, column-major, per filter),
 * R = r_diag[f] * I: used by ZUPT / ZRUPT / position / height / orientation updates. */
int hv_ekf_update(hv_ekf *ekf, int n_rows, int l, const double *H, const double *y, const double *r_diag,
                  const unsigned char *active /* optional [batch] */, int normalize_all_quaternions);
/* visualTrackOutlierCheck (ekf.cpp:787-819) given v = y - f: chi2[f] = noiseScale * v' S^-1 v and
 * status[f] = 0 (INLIER) / 3 (CHI2, chi2 > chi2inv95[n_rows]). Does not modify the filter. Synchronous.
 * n_rows must be < 201 (the size of the reference's chi2inv95 table, odometry/util.hpp:23; the reference asserts it):
 * HV_ERR_INVALID otherwise. Only the reference's defaults of the two optional branches are implemented: r > 0 (the
 * r < 0 "visualR-scaled" branch is not) and trackRmseThreshold = -1 (no RMSE test). S must be positive definite
 * (r > 0 guarantees it): where a pivot of the blocked Cholesky is not positive (the reference's pivoted LDLT would carry on) the
 * filter is reported as CHI2 and left untouched, by the gate and by every update entry point. */
int hv_ekf_visual_gate(hv_ekf *ekf, int n_rows, int l, const double *H, const double *v, double r,
                       double *chi2, int *status);
/* updateVisualTrack (ekf.cpp:829-844) given v = y - f; R = r^2 * noiseScale * I. */
int hv_ekf_visual_update(hv_ekf *ekf, int n_rows, int l, const double *H, const double *v, double r,
                         const unsigned char *active);
/* Device-resident variant: mode 0 = gate only, 1 = update, 2 = update only where the gate passes. */
int hv_ekf_visual_dev(hv_ekf *ekf, int n_rows, int l, const double *H_dev, const double *v_dev, double r,
                      int mode, double *chi2_dev, int *status_dev);
/* ---- per-track triangulation + prepareVisualUpdate on the device ----
 * Replaces, for one pose-trail track per filter, the host code between two EKF calls of the visual update loop
 * (src/odometry/backend.cpp:1063-1148): extractCameraPoseTrail (triangulation.cpp:65-103) from the DEVICE mean,
 * Triangulator::triangulate with derivatives (triangulation.cpp:120-407, the iterative default), the depth window,
 * the stereo derivative sum, prepareVisualUpdate (triangulation.cpp:897-987, full-width H). Parameter names are the
 * reference's (codegen/parameter_definitions.c:37-44,163,178-187). */
typedef struct hv_vu_params {
    double triangulationConvergenceThreshold, triangulationConvergenceR, triangulationRcondThreshold;
    unsigned triangulationGaussNewtonIterations;
    double triangulationMinDist, triangulationMaxDist;
    int estimateImuCameraTimeShift;
    int useStereo;
    double imuToCamera[16], secondImuToCamera[16];     /* 4 x 4 homogeneous, row-major (Parameters::imuToCamera) */
    int useLinearTriangulation;                         /* parameter_definitions.c:31 (default false): triangulateLinear instead of the
                                                          iterative PIVO method (triangulation.cpp:146-152, 820-895) */
    /* ABI 3 (r04): the adaptive outlier thresholds of Session::trackerVisualUpdate (backend.cpp:994-996,1159,1192-1193), applied by the
     * FRAME entry points (hv_ekf_visual_frame*): every track the chi2 / RMSE test rejects multiplies both thresholds of ITS filter by
     * the growth factor for the rest of the frame. trackRmseThreshold is in the units of r_gate (the caller divides by the focal length
     * as backend.cpp:995 does); < 0 = no RMSE test (ekf.cpp:797-801), the reference's default -1. Growth factor default 1
     * (parameter_definitions.c:21,27). The per-visit entry points (hv_ekf_visual_track*) apply the RMSE test with the threshold as
     * given and leave the growth to their caller. */
    double trackRmseThreshold, trackOutlierThresholdGrowthFactor;
} hv_vu_params;
void hv_vu_default_params(hv_vu_params *p);
/* odometry::TriangulatorStatus (output.hpp:21-29) and PrepareVuStatus (output.hpp:15-19) */
enum { HV_TRI_OK = 0, HV_TRI_HYBRID = 1, HV_TRI_BEHIND = 2, HV_TRI_BAD_COND = 3, HV_TRI_NO_CONVERGENCE = 4,
       HV_TRI_BAD_DEPTH = 5, HV_TRI_UNKNOWN_PROBLEM = 6,
       HV_TRI_NOT_VISITED = -1 /* not a reference value: the track was not processed (see hv_ekf_visual_track_limited_dev) */ };
enum { HV_PREPARE_VU_OK = 0, HV_PREPARE_VU_ZERO_DEPTH = 1, HV_PREPARE_VU_BEHIND = 2 };
/* All arrays in device memory, one track of n_poses poses per filter:
 *   pose_index_dev [batch][n_poses]            poseTrailIndex of the track (ekf_state_index: 0 = current pose)
 *   features_dev / velocities_dev [batch][ncam * n_poses][2]   normalized image points / their velocities, the first
 *                                              camera's poses then the second's (buildTrackVectors order)
 *   y_dev [batch][2 * ncam * n_poses]          measured points (the y of buildTrackVectors); may be NULL
 * Outputs: H_dev [batch][(2 * ncam * n_poses) x stateDim] column-major, v_dev = y - f, f_dev (may be NULL), pf_dev
 * [batch][3] world point, status_dev [batch][2] = {TriangulatorStatus, PrepareVuStatus}, active_dev (may be NULL)
 * [batch] = 1 where both are OK -- the inputs hv_ekf_visual_dev takes. Asynchronous. */
int hv_ekf_visual_prepare_dev(hv_ekf *ekf, const hv_vu_params *p, int n_poses, const int *pose_index_dev,
                              const double *features_dev, const double *velocities_dev, const double *y_dev,
                              double *H_dev, double *v_dev, double *f_dev, double *pf_dev, int *status_dev,
                              unsigned char *active_dev);
/* The whole per-track step in one call: prepare as above into internal buffers, then the chi2 gate and -- where the
 * triangulation, prepareVisualUpdate and the gate all pass -- updateVisualTrack (backend.cpp:1150-1185 with
 * batchVisualUpdate false). r_gate = trackChiTestOutlierR (the R of visualTrackOutlierCheck), r_update = visualR.
 * status_dev [batch][2] = {TriangulatorStatus, PrepareVuStatus}; gate_status_dev [batch] = VuOutlierStatus (0 INLIER,
 * 3 CHI2, 1 NOT_COMPUTED where the track never reached the gate); chi2_dev / pf_dev may be NULL. Asynchronous. */
int hv_ekf_visual_track_dev(hv_ekf *ekf, const hv_vu_params *p, int n_poses, const int *pose_index_dev,
                            const double *features_dev, const double *velocities_dev, const double *y_dev,
                            double r_gate, double r_update, int *status_dev, int *gate_status_dev, double *chi2_dev,
                            double *pf_dev);
/* The same with the reference's per-frame stopping rule for a BATCH of sequences: success_counter_dev [batch] counts the
 * visual updates applied to each filter (the caller zeroes it at the start of a frame); a filter whose count has reached
 * max_successful (odometry.maxSuccessfulVisualUpdates = 5, parameter_definitions.c:10) is not visited any more
 * (backend.cpp:1233-1238: status {HV_TRI_NOT_VISITED, HV_TRI_NOT_VISITED}, gate NOT_COMPUTED, filter untouched). */
int hv_ekf_visual_track_limited_dev(hv_ekf *ekf, const hv_vu_params *p, int n_poses, const int *pose_index_dev,
                                    const double *features_dev, const double *velocities_dev, const double *y_dev,
                                    double r_gate, double r_update, int *status_dev, int *gate_status_dev, double *chi2_dev,
                                    double *pf_dev, int *success_counter_dev, int max_successful);
/* One track visit of a session with a HYBRID MAP (odometry.hybridMapSize > 0: backend.cpp:1016, 1075-1082, 1146, 1160-1168; ABI 3, r04).
 * map_update_dev [batch]: the map point a mapPointUpdate track belongs to (>= 0), -1 for a pose-trail track. Such a track is not
 *   triangulated: the point is the state m[stateDim - 3 hybridMapSize + 3 idx ..], status {HV_TRI_HYBRID, prepare status}, H carries
 *   dip R in the point's three columns and no point-derivative terms (triangulation.cpp:968-985); gate and update as usual.
 * map_offer_dev [batch]: the slot ekfStateIndex.offerMapPoint hands to this (pose-trail) track if the gate accepts it, -1: none --
 *   the offer depends on the track and the index, not on the filter, so the adapter evaluates it before the call. Such an inlier is
 *   inserted as a map point (insertMapPoint, ekf.cpp:911-921) INSTEAD of being applied; its gate status reads INLIER.
 * Either array may be NULL. Dense kernels (state wider than 160 columns); trackRmseThreshold < 0 only. Asynchronous. */
int hv_ekf_visual_track_hybrid_dev(hv_ekf *ekf, const hv_vu_params *p, int n_poses, const int *pose_index_dev, const double *features_dev,
                                   const double *velocities_dev, const double *y_dev, const int *map_update_dev, const int *map_offer_dev,
                                   double r_gate, double r_update, int *status_dev, int *gate_status_dev, double *chi2_dev, double *pf_dev);
/* The whole visual-update loop of one frame (Session::trackerVisualUpdate, backend.cpp:1012-1252, with batchVisualUpdate false)
 * in ONE call: n_tracks track visits in the caller's order (the reference's score order), each seeing the mean the previous one
 * left, a filter leaving the loop after max_successful applied updates. Arrays are TRACK-major so that a visit is one contiguous
 * batch: pose_index_dev [n_tracks][batch][n_poses], features_dev / velocities_dev [n_tracks][batch][ncam * n_poses][2], y_dev
 * [n_tracks][batch][2 * ncam * n_poses]; outputs status_dev [n_tracks][batch][2], gate_status_dev [n_tracks][batch], chi2_dev
 * (may be NULL) [n_tracks][batch], pf_dev (may be NULL) the triangulated world points. success_counter_dev [batch] is zeroed here and
 * holds the applied updates on return.
 * Nothing crosses the host between the visits; the call is asynchronous and HIP-graph capturable once it has run once
 * (its work buffers are allocated on first use). The adaptive thresholds of backend.cpp:1192-1193 (hv_vu_params::
 * trackOutlierThresholdGrowthFactor != 1, trackRmseThreshold >= 0; gate status 2 = RMSE) are kept per filter on the device; with a
 * growth factor != 1 the loop always runs sequentially (a rejection changes the threshold of the NEXT track).
 * max_successful <= 0 means "no limit" (the reference's maxSuccessfulVisualUpdates <= 0, backend.cpp:1233).
 * With few sequences the loop runs speculatively: each pass prepares and gates every pending track of a filter in parallel against the
 * current (m, P), applies the first inlier in visit order and re-examines only the tracks behind it -- at most min(max_successful,
 * n_tracks) + 1 passes, the same statuses and the same filter as the sequential loop. Which frames take that form:
 *   - longest track <= 48 rows (12 stereo poses): batch * n_tracks <= 256, n_tracks >= 2, trackOutlierThresholdGrowthFactor == 1;
 *   - longest track 49 .. 84 rows (21 stereo poses, the reference's default stereo configuration; r04): batch * n_tracks <= the device's
 *     CU count (256 on MI355X), the same conditions, and the library's default kernel selection (knobs ekf_long_fused != 0,
 *     ekf_spec_split == 0, ekf_spec_mode != 3, ekf_no_speculation == 0);
 *   - every other frame (more filters, adaptive thresholds) runs the sequential visit loop, with the same results.
 * chi2_dev / pf_dev entries of tracks the loop never visits (status HV_TRI_NOT_VISITED) are 0.
 * hv_vu_params: always start from hv_vu_default_params() -- fields added in later ABI versions then hold their defaults. */
int hv_ekf_visual_frame_dev(hv_ekf *ekf, const hv_vu_params *p, int n_tracks, int n_poses, const int *pose_index_dev,
                            const double *features_dev, const double *velocities_dev, const double *y_dev, double r_gate,
                            double r_update, int *status_dev, int *gate_status_dev, double *chi2_dev, double *pf_dev /* [n_tracks][batch][3] or NULL */,
                            int *success_counter_dev, int max_successful);
/* Host-pointer form (arrays track-major as above): what a single session calls once per frame instead of n_tracks round trips
 * through hv_ekf_visual_track. chi2, pf and success_count may be NULL. Synchronous. */
/* The same loop over RAGGED tracks: the sequences of a batch do not share track lengths (nor the number of candidate tracks).
 * n_poses_dev [n_tracks][batch]: poses of every (visit, filter) track, 2 .. n_poses_max; anything else (0) = this filter has no
 * track at this visit (its statuses read HV_TRI_NOT_VISITED, nothing is gated). Every other array keeps the layout of
 * hv_ekf_visual_frame_dev with n_poses_max as the record size: a track's poses (first camera's, then the second's) sit at the
 * start of its record, the rest is padding. */
int hv_ekf_visual_frame_ragged_dev(hv_ekf *ekf, const hv_vu_params *p, int n_tracks, int n_poses_max, const int *n_poses_dev,
                                   const int *pose_index_dev, const double *features_dev, const double *velocities_dev,
                                   const double *y_dev, double r_gate, double r_update, int *status_dev, int *gate_status_dev,
                                   double *chi2_dev, double *pf_dev, int *success_counter_dev, int max_successful);
/* Session::trackerVisualUpdate with batchVisualUpdate -- or on a frame that is not a "full visual update" -- (backend.cpp:1001-1010,
 * 1169-1183, 1255-1262; ABI 3, r04): the inliers' blocks [H; f; y] are collected and applied as ONE updateVisualTrack per batch; a block
 * that would take the batch beyond max_update_rows flushes it first and opens the next one. Every track between two flushes is prepared
 * and gated against the same state, which the device does in one launch per batch (<= quota + 1 passes per frame). Arguments as
 * hv_ekf_visual_frame_ragged_dev (n_poses_dev may be NULL: every track has n_poses_max poses); max_update_rows = int(stateDim *
 * batchVisualUpdateMaxSizeMultiplier + 0.5), <= 0: stateDim. Valid for trackOutlierThresholdGrowthFactor 1 (HV_ERR_UNSUPPORTED
 * otherwise), n_tracks <= 64, n_tracks * batch <= 8192, max_update_rows >= the rows of the longest track and <= stateDim. */
int hv_ekf_visual_frame_batch_dev(hv_ekf *ekf, const hv_vu_params *p, int n_tracks, int n_poses_max, const int *n_poses_dev,
                                  const int *pose_index_dev, const double *features_dev, const double *velocities_dev, const double *y_dev,
                                  double r_gate, double r_update, int *status_dev, int *gate_status_dev, double *chi2_dev, double *pf_dev,
                                  int *success_counter_dev, int max_successful, int max_update_rows);
/* Reads (and clears) the error word of the filter batch: non-zero when a device-side wait of an asynchronous frame call gave up
 * (only the experimental hand-shake form of the speculative pass, knob ekf_spec_mode = 3, can raise it); the host-pointer entry
 * points check it themselves and return HV_ERR_TIMEOUT. Synchronous. */
int hv_ekf_frame_error(hv_ekf *ekf, int *flags);
int hv_ekf_visual_frame(hv_ekf *ekf, const hv_vu_params *p, int n_tracks, int n_poses, const int *pose_index, const double *features,
                        const double *velocities, const double *y, double r_gate, double r_update, int *status, int *gate_status,
                        double *chi2, double *pf, int *success_count, int max_successful);
/* host-pointer form of hv_ekf_visual_frame_ragged_dev (n_poses [n_tracks][batch] in host memory); synchronous */
int hv_ekf_visual_frame_ragged(hv_ekf *ekf, const hv_vu_params *p, int n_tracks, int n_poses_max, const int *n_poses, const int *pose_index,
                               const double *features, const double *velocities, const double *y, double r_gate, double r_update,
                               int *status, int *gate_status, double *chi2, double *pf, int *success_count, int max_successful);
/* host-pointer form of hv_ekf_visual_frame_batch_dev (ABI 4): what a single session with batchVisualUpdate calls once per frame
 * (backend.cpp:1001-1010,1169-1183,1255-1262). n_poses may be NULL (every track has n_poses_max poses). Synchronous. */
int hv_ekf_visual_frame_batch(hv_ekf *ekf, const hv_vu_params *p, int n_tracks, int n_poses_max, const int *n_poses, const int *pose_index,
                              const double *features, const double *velocities, const double *y, double r_gate, double r_update,
                              int *status, int *gate_status, double *chi2, double *pf, int *success_count, int max_successful,
                              int max_update_rows);
/* host-pointer form of hv_ekf_visual_track_hybrid_dev (ABI 4): one track visit per filter with the hybrid-map branches
 * (backend.cpp:1075-1082 mapPointUpdate tracks, :1146-1168 insertMapPoint for an offered slot). map_update / map_offer [batch] or NULL.
 * chi2 / pf may be NULL. Synchronous. */
int hv_ekf_visual_track_hybrid(hv_ekf *ekf, const hv_vu_params *p, int n_poses, const int *pose_index, const double *features,
                               const double *velocities, const double *y, const int *map_update, const int *map_offer, double r_gate,
                               double r_update, int *status, int *gate_status, double *chi2, double *pf);
/* EKF::insertMapPoint / getMapPoint (ekf.hpp:129-131, ekf.cpp:905-921) on the resident state of filter `filter` (ABI 4): the slot's rows
 * and columns of P are zeroed, 1e6 goes on its diagonal, pf into the mean -- 24 bytes cross the bus instead of the covariance twice.
 * insert is asynchronous (pf is read before the call returns), get synchronous. map_index in [0, hybridMapSize). */
int hv_ekf_insert_map_point(hv_ekf *ekf, int filter, int map_index, const double *pf);
int hv_ekf_get_map_point(hv_ekf *ekf, int filter, int map_index, double *pf);
/* Host-pointer form of hv_ekf_visual_track_dev (arrays [batch][...] as above): about 1 KB per track goes to the device
 * and 40 bytes come back, instead of the mean coming back and a (2 * ncam * n_poses) x stateDim Jacobian going up.
 * chi2 / pf may be NULL. Synchronous. */
int hv_ekf_visual_track(hv_ekf *ekf, const hv_vu_params *p, int n_poses, const int *pose_index, const double *features,
                        const double *velocities, const double *y, double r_gate, double r_update, int *status,
                        int *gate_status, double *chi2, double *pf);
/* updateVisualPoseAugmentation(discarded[f]) (ekf.cpp:848-885; -1 = last pose) incl. the Joseph form,
 * maintainPositiveSemiDefinite and normalizeQuaternions; updateUndoAugmentation (ekf.cpp:888-903). */
int hv_ekf_augment(hv_ekf *ekf, const int *discarded /* [batch] or NULL */, const unsigned char *active);
/* The same with device arrays (either may be NULL): no host copy, so a batch of sequences stays HIP-graph capturable. */
int hv_ekf_augment_dev(hv_ekf *ekf, const int *discarded_dev, const unsigned char *active_dev);
/* hv_ekf_symmetrize followed by hv_ekf_augment_dev in ONE pass over the covariances (ABI 3, r04): the end of a frame's visual
 * updates (maintainPositiveSemiDefinite, backend.cpp:1267) and the pose augmentation that follows it. The augmentation reads
 * every covariance element together with its mirror, so (P + P') / 2 is formed on the fly; results are bit-identical to the
 * two calls in sequence (tests/test_gpu_ekf.py), inactive filters come out symmetrised and otherwise untouched. */
int hv_ekf_symmetrize_augment_dev(hv_ekf *ekf, const int *discarded_dev, const unsigned char *active_dev);
/* host-array form (ABI 4): maintainPositiveSemiDefinite + updateVisualPoseAugmentation as the backend issues them at the end of a frame
 * (backend.cpp:1267, 804-805). States with map points fall back to the two separate calls (same results). */
int hv_ekf_symmetrize_augment(hv_ekf *ekf, const int *discarded /* [batch] or NULL */, const unsigned char *active);
int hv_ekf_undo_augment(hv_ekf *ekf, const unsigned char *active);
int hv_ekf_symmetrize(hv_ekf *ekf);                                 /* maintainPositiveSemiDefinite */
int hv_ekf_normalize_quaternions(hv_ekf *ekf, int only_current);    /* ekf.cpp:1024-1032            */
/* transformTo (ekf.cpp:704-758): the adapter computes pChangeMat (3x3), qChangeMat (4x4) (both
 * row-major) and the translation from the host mirror; the device applies m = A m, P = A P A'. */
int hv_ekf_transform(hv_ekf *ekf, int filter, const double *pChange3x3, const double *qChange4x4,
                     const double *translation3);

/* ---- GFTT feature detector ----------------------------------------
 * Replaces tracker::FeatureDetector (featureDetector "GPU-GFTT" on its CPU path:
 * src/tracker/feature_detector.cpp:279-315 cv::cornerMinEigenVal, :393-417 CollectMax, :610-634
 * detect(); src/tracker/feature_detector_legacy.cpp:177-213 applyMinDistance). Field names are the
 * reference's parameters (codegen/parameter_definitions.c:262,317-324). */
typedef struct hv_gftt_params {
    int    gfttBlockSize;      /* box filter of the structure matrix: 3 (default; tuned kernels), 5 or 7 (plain kernel) */
    double gfttMinDistance;    /* selects the arg-max block edge: >= 32 -> 32, >= 16 -> 16, else 8          */
    float  gfttMinResponse;    /* key points need 16 * minEigenVal > this                                   */
    int    maxTracks;          /* applyMinDistance stops after this many corners                            */
} hv_gftt_params;
void hv_gftt_default_params(hv_gftt_params *p);
int hv_gftt_block_size(const hv_gftt_params *p);                    /* CollectMax::blockSize()            */
int hv_gftt_keypoint_count(hv_ctx *ctx, const hv_gftt_params *p);   /* floor(w/bs) * floor(h/bs)          */
/* FeatureDetector::detect(image, corners, prevCorners, maskRadius) on the level-0 image of a pyramid
 * slot that has been built (or is being built on the context stream): response + block arg-max on
 * the device, then the reference's host tail (stable sort by response, one (0,0) point prepended per
 * key point -- feature_detector.cpp:629-631 --, applyMinDistance when mask_radius > 0). corners_xy
 * must hold 2 * hv_gftt_keypoint_count() points. Synchronous. */
int hv_gftt_detect(hv_ctx *ctx, const hv_gftt_params *p, int slot, const float *prev_xy, int n_prev,
                   int mask_radius, float *corners_xy, int capacity, int *n_out);
/* Device half only, batched: kp_dev[n_images][hv_gftt_keypoint_count()][3] = (x, y, 16 * response) per
 * block in block raster order ((0, 0, -1e10) where no pixel exceeds gfttMinResponse). Asynchronous. */
int hv_gftt_keypoints_batch_dev(hv_ctx *ctx, const hv_gftt_params *p, int n_images, const int *slots_dev,
                                float *kp_dev);
/* FeatureDetector::applyMinDistance (host only, no device work): in place, *n_inout updated. */
void hv_apply_min_distance(float *corners_xy, int *n_inout, const float *prev_xy, int n_prev, int r,
                           int max_tracks);

/* ---- image ingest --------------------------------------------------
 * Replaces the per-frame work of tracker::Image::Factory::build / buildStereo (src/tracker/image.cpp:272-306):
 * the colour -> gray copy (image.cpp:351-367, coefficients 0.299 / 0.587 / 0.114 on channels 0, 1, 2) and
 * tracker::Undistorter::undistort (src/tracker/undistorter.cpp:71-110, the CPU branch: bilinear remap through
 * rectifiedCamera.pixelToRay -> originalCamera.rayToPixel), followed by the pyramid level kernels. The result is
 * level 0 of the slot, so the tracker never sees the raw frame.
 * channels: 1 (gray), 3 (RGB / BGR) or 4 (RGBA / BGRA), 8 bits each, interleaved.
 * camera: -1 = no remap (tracker.useRectification false, the default: parameter_definitions.c:356), otherwise the
 * index (0 first / 1 second camera: image.cpp:322-328) of a table installed with hv_ingest_set_undistort_map. */
#define HV_INGEST_CAMERAS 2
/* Installs (pix_orig_xy != NULL) or removes (NULL) the remap table of one camera. pix_orig_xy[(y*w + x)*2 + {0,1}]
 * = the position in the original image of rectified pixel (x, y), i.e. what undistortCpu (undistorter.cpp:56-60)
 * returns for it, evaluated by the caller with the reference's own Camera objects; valid[y*w + x] = its boolean
 * result (NULL = all true). The cameras are constant for a session, so this runs once (the reference's GPU branch
 * makes the same assumption: undistorter.cpp:113-121). Synchronous. */
int hv_ingest_set_undistort_map(hv_ctx *ctx, int camera, const double *pix_orig_xy, const uint8_t *valid);
/* H2D copy of one host frame + ingest + all pyramid level kernels into `slot`; asynchronous like hv_pyramid_build. */
int hv_ingest_build(hv_ctx *ctx, int slot, const uint8_t *image_host, int stride_bytes, int channels, int camera);
/* n frames resident in HBM (frame i at src_dev + i*image_stride_bytes; base and strides multiples of 4 bytes).
 * Unlike hv_pyramid_build_batch_dev the result is written into the slots, so the frames may be reused at once.
 * Asynchronous. */
int hv_ingest_build_batch_dev(hv_ctx *ctx, int n, const int *slots_dev, const uint8_t *src_dev,
                              long long image_stride_bytes, int row_stride_bytes, int channels, int camera);

/* ---- 2-point rotation RANSAC -----------------------------------------
 * Replaces tracker::rot_ransac::RotRansac::fit (src/tracker/rot_ransac.cpp:41-120) as doRansac2 calls it on the
 * TRACKED features of every frame (src/tracker/ransac_pipeline.cpp:197-216). The camera models are the reference's
 * (src/tracker/camera.cpp): fill the first block of fields and call hv_camera_model_init. */
typedef struct hv_camera_model {
    int kind;                         /* 0 pinhole (radial k1 k2 k3, camera.cpp:93-221), 1 fisheye (k1..k4, :264-398) */
    double fx, fy, ppx, ppy;          /* api::CameraParameters */
    int n_coeffs; double coeffs[4];   /* distortionCoeffs: 0 or 3 (pinhole), 0 or 4 (fisheye) */
    int rotation_enabled; double rotation[9];   /* pinhole only (rectified cameras), row-major */
    double max_valid_fov_deg;         /* fisheye: validCameraFov */
    /* derived by hv_camera_model_init (the constructors of camera.cpp) */
    int distortion_enabled;
    double kinv[9], max_theta, max_r;
    int n_table; double table[50];
} hv_camera_model;
int hv_camera_model_init(hv_camera_model *m);
/* One point set = the n TRACKED features of one frame: c1 / c2 = pixel coordinates in the previous / current frame
 * (cameras[0][0] / cameras[0][1]). pairs[k] = {rng() % n, rng() % n} for k = 0..99 drawn by the caller from its
 * std::mt19937 (rot_ransac.cpp:82-83); afterwards the caller discards nothing and keeps 2 * hypotheses_visited draws
 * consumed (the reference loop stops at the first hypothesis that makes every point an inlier). threshold_pow2 =
 * RotRansac::threshold_pow2 (ransac_pipeline.cpp:91-93). status[i] = 0 TRACKED / 3 RANSAC_OUTLIER (track.hpp:9-21),
 * R = the returned cv::Matx33f (row-major), best_inlier_count = RotRansac::bestInlierCount. Synchronous. */
int hv_rot_ransac(hv_ctx *ctx, int n, const float *c1_xy, const float *c2_xy, const hv_camera_model *camera1,
                  const hv_camera_model *camera2, const int *pairs, float threshold_pow2, int *status, float *R,
                  int *best_inlier_count, int *hypotheses_visited);
/* n_sets point sets in device memory, set s holding n_points_dev[s] <= max_points (<= 1024) points at
 * c*_dev + s * max_points * 2, pairs_dev [n_sets][100][2], status_dev [n_sets][max_points], R_dev [n_sets][9],
 * summary_dev [n_sets][2] = {bestInlierCount, hypotheses_visited}. Sets with fewer than 2 points are skipped
 * (ransac_pipeline.cpp:209). Asynchronous. */
int hv_rot_ransac_batch_dev(hv_ctx *ctx, int n_sets, int max_points, const int *n_points_dev, const float *c1_dev,
                            const float *c2_dev, const hv_camera_model *camera1, const hv_camera_model *camera2,
                            const int *pairs_dev, float threshold_pow2, int *status_dev, float *R_dev, int *summary_dev);

/* The same fed straight from the LK outputs of hv_klt_track_batch_dev, nothing passing through the host: set s = the
 * n_points_dev[s] features of pair s (prev_xy / next_xy / status of the LK call, max_points = pts_per_pair); only features
 * whose lk_status equals lk_tracked_value (1 for hv_klt_track_batch_dev) take part, in feature order -- the c1 / c2 of
 * ransac_pipeline.cpp:106-112. draws_dev [n_sets][200] = the next 200 raw outputs of each set's std::mt19937; the kernel
 * forms rng() % n itself because n is only known on the device. status_dev [n_sets][max_points] is written at the ORIGINAL
 * feature numbers (0 / 3), other entries stay; summary_dev as above. Asynchronous. */
int hv_rot_ransac_lk_batch_dev(hv_ctx *ctx, int n_sets, int max_points, const int *n_points_dev, const float *c1_dev,
                               const float *c2_dev, const uint8_t *lk_status_dev, int lk_tracked_value,
                               const hv_camera_model *camera1, const hv_camera_model *camera2, const uint32_t *draws_dev,
                               float threshold_pow2, int *status_dev, float *R_dev, int *summary_dev);

/* ---- per-kernel timing (hipEvents on the context stream) ---------------------------------- */
enum { HV_K_PYR_L0 = 0, HV_K_PYR_LN = 1, HV_K_KLT = 2, HV_K_EKF_PREDICT = 3, HV_K_EKF_UPDATE = 4,
       HV_K_EKF_AUGMENT = 5, HV_K_GFTT = 6, HV_K_INGEST = 7, HV_K_VU_PREPARE = 8, HV_K_ROT_RANSAC = 9, HV_K_EKF_GATE = 10,
       HV_K_VU_TRI = 11 /* r06: the triangulation front of the split form (vu_tri_kernel); HV_K_VU_PREPARE then times the record-fed gates */,
       HV_K_COUNT = 12 };
int hv_profile_enable(hv_ctx *ctx, int on);
int hv_profile_reset(hv_ctx *ctx);
/* Synchronizes, then returns accumulated device milliseconds and launch count of a kernel class. */
int hv_profile_read(hv_ctx *ctx, int kernel_id, double *total_ms, long long *launches);

#ifdef __cplusplus
}
#endif
#endif /* HYBVIO_HIP_H_ */
