// hybvio_host.hpp -- C++ host adapters that present libhybvio_hip.so (include/hybvio_hip.h) behind
// HybVIO's own interfaces for the hot path:
//
//   tracker::ImagePyramid / ImagePyramid::Factory   src/tracker/image_pyramid.hpp:18-43
//   tracker::OpticalFlow                            src/tracker/optical_flow.hpp:20-40
//   tracker::Feature::{Point,Status}                src/tracker/track.hpp:8-32
//   odometry::EKF                                   src/odometry/ekf.hpp:62-174
//
// Same class and method names, argument order and meaning, ownership (factories return unique_ptr,
// pyramids are shared_ptr recycled from a pool) and error behaviour (contract violations assert;
// algorithmic failure is data: Feature::Status / VuOutlierStatus) as the reference. The only
// difference is vocabulary that the reference takes from libraries that are not vendored here:
// accelerated::Image becomes GrayImage (pointer + size + stride) and Eigen vectors / matrices
// become the plain aliases below (column-major storage, exactly what Eigen::Map wraps).
// INTEGRATION.md shows the three-line glue that binds these to the reference's real types and the
// two call sites (image.cpp:55-56, backend.cpp:187) where the HIP factories are selected.
//
// This layer is a pure client of the C ABI: it includes no HIP header and links only
// libhybvio_hip.so.
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <random>
#include <string>
#include <stdexcept>
#include <vector>

#include "../../include/hybvio_hip.h"

namespace hybvio {

// ---- stand-ins for Eigen types (same memory layout as Eigen's defaults) ----
using Vector3d = std::array<double, 3>;
using Vector4d = std::array<double, 4>;
using VectorXd = std::vector<double>;
struct MatrixXd {                       // column-major, like Eigen::MatrixXd
    int rows = 0, cols = 0;
    std::vector<double> data;
    MatrixXd() = default;
    MatrixXd(int r, int c) : rows(r), cols(c), data((size_t)r * c, 0.0) {}
    double &operator()(int i, int j) { return data[(size_t)j * rows + i]; }
    double operator()(int i, int j) const { return data[(size_t)j * rows + i]; }
};

struct DeviceError : std::runtime_error {
    int status;
    DeviceError(int rc, const std::string &msg) : std::runtime_error(msg), status(rc) {}
};

// One device session: the hv_ctx (stream + pyramid pool) shared by the tracker adapters and the EKF.
class Session {
public:
    explicit Session(const hv_params &params);
    ~Session();
    Session(const Session &) = delete;
    Session &operator=(const Session &) = delete;
    // a session on a context somebody else owns: lane i of a Lanes set (below)
    Session(hv_ctx *borrowed, const hv_params &params) : ctx_(borrowed), params_(params), owned_(false) {}
    hv_ctx *ctx() const { return ctx_; }
    const hv_params &params() const { return params_; }
    // Every C-ABI return code of the adapters passes through here: a device failure throws DeviceError (also under
    // NDEBUG) instead of leaving the caller with stale outputs. Contract violations stay asserts like the reference's.
    void check(int rc, const char *what) const;
private:
    hv_ctx *ctx_ = nullptr;
    hv_params params_{};
    bool owned_ = true;
};

// Several sessions of one process on one GPU whose launch chains are meant to run beside each other (a server multiplexing VIO
// sessions onto a GPU; the reference runs one session per process, main.cpp): hv_lanes_create gives every lane streams from the
// device's high-priority hardware-queue pool, so the lanes' work overlaps whatever else the process created (include/hybvio_hip.h).
// session(i) is a Session on lane i's context; the adapters' buildHip factories take it like any other Session.
class Lanes {
public:
    Lanes(const hv_params &params, int n);
    ~Lanes();
    Lanes(const Lanes &) = delete;
    Lanes &operator=(const Lanes &) = delete;
    int size() const { return (int)sessions_.size(); }
    Session &session(int i) { return *sessions_.at((size_t)i); }
    void *stream(int i) { return hv_get_stream(sessions_.at((size_t)i)->ctx()); }   // hipStream_t of lane i
private:
    hv_lanes *lanes_ = nullptr;
    std::vector<std::unique_ptr<Session>> sessions_;
};

namespace tracker {

struct Feature {   // src/tracker/track.hpp:8-32
    enum class Status { TRACKED, NEW, FAILED_FLOW, RANSAC_OUTLIER, FLOW_OUT_OF_RANGE, OUT_OF_RANGE,
                        FAILED_EPIPOLAR_CHECK, CULLED, BLACKLISTED };
    struct Point { float x, y; };
    int id = -1;
    Status status = Status::NEW;
    std::array<Point, 2> points = {{{-1, -1}, {-1, -1}}};
    float depth = -1;
};

struct GrayImage {           // what accelerated::opencv::ref(cv::Mat) carries at image.cpp:211
    const std::uint8_t *data;
    int width, height, strideBytes;
};

struct InputImage {          // the CPU accelerated::Image handed to Image::Factory::build (image.cpp:226-243)
    const std::uint8_t *data;
    int width, height, channels, strideBytes;      // channels 1, 3 or 4, 8 bits each
};

struct ImagePyramid {        // src/tracker/image_pyramid.hpp:18-43
    typedef std::uint8_t GrayType;
    typedef std::int16_t GradientType;
    static constexpr std::size_t GRADIENT_CHANNELS = 2;
    static constexpr float GRADIENT_SCALE_0_255 = 1.0f / 32;
    static constexpr float GRADIENT_SCALE_01 = GRADIENT_SCALE_0_255 / 255;

    // The reference's getGrayLevel/getGradientLevel are `assert(false && "TODO")` GPU hooks
    // (image_pyramid.cpp:17-25); here they read a level back from the device.
    virtual std::vector<GrayType> getGrayLevel(std::size_t i, int &width, int &height) = 0;
    virtual std::vector<GradientType> getGradientLevel(std::size_t i, int &width, int &height) = 0;
    virtual int deviceSlot() const = 0;
    virtual ~ImagePyramid();

    struct Factory {
        virtual std::shared_ptr<ImagePyramid> compute(const GrayImage &image) = 0;
        // colour (or gray) frame without rectification: the colorToGrayOp / copy of image.cpp:351-367 on the device
        virtual std::shared_ptr<ImagePyramid> computeFromFrame(const InputImage &image) = 0;
        virtual ~Factory();
        static std::unique_ptr<Factory> buildHip(Session &session);   // sibling of buildOpenCv
    };
};

// tracker::Camera (src/tracker/camera.hpp:12-90): the two methods the undistorter calls. In the reference tree the
// adapter takes the reference's own Camera (Eigen::Vector2d / Vector3d are layout-compatible with these arrays).
class Camera {
public:
    virtual ~Camera() = default;
    virtual bool pixelToRay(const double pixel[2], double ray[3]) const = 0;
    virtual bool rayToPixel(const double ray[3], double pixel[2]) const = 0;
    virtual double getFocalLength() const = 0;
};

// tracker::Undistorter (src/tracker/undistorter.hpp:14-41), SURVEY.md 8(f) row f2. The reference returns the
// rectified gray image; here it never leaves the device: the result is the pyramid whose level 0 it is, built by the
// same call (colour -> gray, remap, pyramid levels: Image::Factory::buildPrivate, image.cpp:272-306).
// cameraIndex 0 / 1 = first / second camera (image.cpp:322-328).
struct Undistorter {
    struct Result {
        std::shared_ptr<const Camera> camera;
        std::shared_ptr<ImagePyramid> image;
    };
    virtual ~Undistorter();
    static std::unique_ptr<Undistorter> buildRectifiedHip(Session &session, int cameraIndex,
                                                          std::shared_ptr<const Camera> rectifiedCamera);
    virtual Result undistort(const InputImage &image, std::shared_ptr<const Camera> camera) = 0;
};

struct OpticalFlow {         // src/tracker/optical_flow.hpp:20-40
    static std::unique_ptr<OpticalFlow> buildHip(Session &session);   // sibling of buildOpenCv
    virtual ~OpticalFlow();
    virtual void compute(
        ImagePyramid &prevImagePyramid,
        ImagePyramid &imagePyramid,
        const std::vector<Feature::Point> &prevCorners,
        std::vector<Feature::Point> &corners,
        std::vector<Feature::Status> &trackStatus,
        bool useInitialCorners,
        int overrideMaxIterations = -1) = 0;
};

// tracker::FeatureDetector (src/tracker/feature_detector.hpp:20-78), SURVEY.md 8(f) row f1. The reference
// passes the tracker::Image whose pixels it detects on; the HIP detector reads the level-0 image of
// that frame's device-resident pyramid, so the adapter takes the ImagePyramid the Image already owns
// (image.cpp:209-214 ensureImagePyramid) -- see INTEGRATION.md for the two-line glue in
// ImageImplementation::findKeypoints (image.cpp:69-85).
class FeatureDetector {
public:
    static std::unique_ptr<FeatureDetector> buildHip(Session &session, const hv_gftt_params &parameters);   // sibling of build()
    virtual ~FeatureDetector();
    virtual void detect(
        ImagePyramid &imagePyramid,
        std::vector<Feature::Point> &corners,
        const std::vector<Feature::Point> &prevCorners,
        int maskRadius) = 0;
    virtual bool supportsAsync() const { return false; }
    // feature_detector_legacy.cpp:177-213 (pure host code, same for every detector)
    void applyMinDistance(
        std::vector<Feature::Point> &corners,
        const std::vector<Feature::Point> &prevCorners,
        int minDistance) const;

protected:
    explicit FeatureDetector(const hv_gftt_params &parameters) : parameters(parameters) {}
    const hv_gftt_params parameters;
};

// tracker::rot_ransac::RotRansac (src/tracker/rot_ransac.hpp:14-41), SURVEY.md 8(f) row f4. Same members and call:
// fit() consumes rng exactly as the reference loop does (two draws per visited hypothesis, rot_ransac.cpp:82-83,104), so
// every later user of the pipeline's generator sees the same stream. The cameras are described by hv_camera_model
// (the reference's tracker::Camera parameters); the returned rotation is the cv::Matx33f, row-major.
namespace rot_ransac {
class RotRansac {
public:
    static std::unique_ptr<RotRansac> buildHip(Session &session);
    virtual ~RotRansac();
    virtual std::array<float, 9> fit(
        const std::vector<Feature::Point> &c1,
        const std::vector<Feature::Point> &c2,
        const hv_camera_model &camera1,
        const hv_camera_model &camera2,
        std::vector<Feature::Status> &bestInliers,
        std::mt19937 &rng) = 0;
    std::size_t bestInlierCount = 0;
    float threshold_pow2 = 2.0f * 2.0f;      // "Likely set later with information of the frame size." (rot_ransac.cpp:164)
};
}  // namespace rot_ransac

}  // namespace tracker

namespace odometry {

// state layout: src/odometry/ekf.hpp:26-50
constexpr int POS = 0, VEL = 3, ORI = 6, BGA = 10, BAA = 13, BAT = 16, SFT = 19, CAM = 20;
constexpr int INER_DIM = CAM, POSE_DIM = 7, MAP_POINT_DIM = 3;
constexpr int Q_ACC = 0, Q_GYRO = 3, Q_BGA_DRIFT = 6, Q_BAA_DRIFT = 9, Q_DIM = 12;

enum class VuOutlierStatus { INLIER, NOT_COMPUTED, RMSE, CHI2 };

using MatrixInertialCov = MatrixXd;     // INER_DIM x INER_DIM
using VectorInertialMean = VectorXd;    // INER_DIM

// Every pure virtual of odometry::EKF (ekf.hpp:62-174), same order.
class EKF {
public:
    static std::unique_ptr<EKF> buildHip(Session &session, const hv_ekf_params &parameters);
    virtual std::unique_ptr<EKF> clone() const = 0;
    virtual ~EKF();

    virtual void initializeOrientation(const Vector3d &xa) = 0;
    virtual void predict(double t, const Vector3d &xg, const Vector3d &xa) = 0;
    virtual Vector3d position() const = 0;
    virtual Vector3d velocity() const = 0;
    virtual Vector4d orientation() const = 0;
    virtual Vector3d biasGyroscopeAdditive() const = 0;
    virtual Vector3d biasAccelerometerAdditive() const = 0;
    virtual Vector3d biasAccelerometerTransform() const = 0;
    virtual int camTrailSize() const = 0;
    virtual Vector3d historyPosition(int i) const = 0;
    virtual Vector4d historyOrientation(int i) const = 0;
    virtual double historyTime(int i) const = 0;
    virtual double speed() const = 0;
    virtual double horizontalSpeed() const = 0;
    virtual void updateZupt(double r) = 0;
    virtual void updateZuptInitialization() = 0;
    virtual void updateZrupt(const Vector3d &xg) = 0;
    virtual void updatePseudoVelocity(double defaultSpeed, double r) = 0;
    virtual void updatePosition(const Vector3d &pos, double r) = 0;
    virtual void updateZeroHeight(double r) = 0;
    virtual void updateOrientation(const Vector4d &q, double r) = 0;
    virtual void getInertialState(VectorInertialMean &mean, MatrixInertialCov &cov) const = 0;
    virtual void setInertialState(const VectorInertialMean &mean, const MatrixInertialCov &cov) = 0;
    virtual double getImuToCameraTimeShift() const = 0;
    virtual void translateTo(const Vector3d &pos) = 0;
    virtual void transformTo(const Vector3d &pos, const Vector4d &q, int i = -1) = 0;
    virtual VuOutlierStatus visualTrackOutlierCheck(const MatrixXd &visH, const VectorXd &f, const VectorXd &y,
                                                    double r, double trackRmseThreshold) = 0;
    virtual void updateVisualTrack(const MatrixXd &visH, const VectorXd &f, const VectorXd &y, double r) = 0;
    virtual void updateVisualPoseAugmentation(int discardedPoseIndex = -1) = 0;
    virtual void updateUndoAugmentation() = 0;
    virtual Vector3d getMapPoint(int idx) const = 0;
    virtual void insertMapPoint(int idx, const Vector3d &pf) = 0;
    virtual int getMapPointStateIndex(int idx) const = 0;
    virtual void conditionOnLastPose() = 0;
    virtual void lockBiases() = 0;
    virtual void normalizeQuaternions(bool onlyCurrent) = 0;
    virtual void setFirstSampleTime(double t) = 0;
    virtual bool isPositiveSemiDefinite() = 0;
    virtual void maintainPositiveSemiDefinite() = 0;
    virtual void setState(const VectorXd &m) = 0;
    virtual void setStateCovariance(const MatrixXd &P) = 0;
    virtual void setProcessNoise(const MatrixXd &Q) = 0;
    virtual double getPlatformTime() const = 0;
    virtual int getPoseCount() const = 0;
    virtual const VectorXd &getState() const = 0;
    virtual MatrixXd getStateCovariance() const = 0;
    virtual const MatrixXd &getStateCovarianceRef() const = 0;
    virtual MatrixXd getVisAugH() const = 0;
    virtual MatrixXd getVisAugA() const = 0;
    virtual MatrixXd getVisAugQ() const = 0;
    virtual MatrixXd getDydx() const = 0;
    virtual std::string stateAsString() const = 0;
    virtual int getStateDim() const = 0;
    virtual bool getWasStationary() const = 0;

    // SURVEY.md 8(f) row f3 -- not part of the reference's EKF interface: the body of the per-track loop of
    // Session::trackerVisualUpdate (backend.cpp:1063-1185: extractCameraPoseTrail, Triangulator::triangulate with
    // derivatives, depth window, prepareVisualUpdate, visualTrackOutlierCheck with chiOutlierR, updateVisualTrack with
    // visualR) as ONE device pass on the resident mean. poseTrailIndex, imageFeatures, featureVelocities and y are what
    // EKFStateIndex::buildTrackVectors produces (imageFeatures / featureVelocities: x0 y0 x1 y1 ..., first camera's poses
    // then the second's). Map-point (hybrid) tracks keep the reference's host path.
    struct VisualTrackResult {
        int triangulateStatus;          // odometry::TriangulatorStatus (output.hpp:21-29)
        int prepareVuStatus;            // odometry::PrepareVuStatus (output.hpp:15-19)
        VuOutlierStatus outlierStatus;  // NOT_COMPUTED unless both of the above are OK
        Vector3d pf;                    // triangulated point (world)
    };
    virtual VisualTrackResult visualTrack(const hv_vu_params &parameters, const std::vector<int> &poseTrailIndex,
                                          const std::vector<double> &imageFeatures, const std::vector<double> &featureVelocities,
                                          const VectorXd &y, double chiOutlierR, double visualR) = 0;
    // The whole loop `for (trackIndex : tmp.trackOrder)` of Session::trackerVisualUpdate (backend.cpp:1012-1252) for pose-trail tracks
    // of ONE length in ONE device call (hv_ekf_visual_frame: a single round trip instead of one per track; with one session the
    // device runs it speculatively, see include/hybvio_hip.h). tracks[k] are in the backend's visit order; the loop stops applying
    // updates after maxSuccessfulVisualUpdates (results of unvisited tracks: triangulateStatus = -1). The backend replays its
    // per-track side effects (blacklist, statistics, point cloud) from the returned statuses. The adaptive thresholds of
    // backend.cpp:994-996,1192-1193 travel in `parameters` (hv_vu_params::trackRmseThreshold -- already divided by the focal length, like
    // chiOutlierR -- and trackOutlierThresholdGrowthFactor; ABI 3): outlierStatus RMSE / CHI2 come back per track, the growth is applied
    // per frame on the device.
    struct VisualFrameTrack {
        std::vector<int> poseTrailIndex;
        std::vector<double> imageFeatures, featureVelocities;
        VectorXd y;
    };
    virtual std::vector<VisualTrackResult> visualFrame(const hv_vu_params &parameters, const std::vector<VisualFrameTrack> &tracks,
                                                       double chiOutlierR, double visualR, int maxSuccessfulVisualUpdates,
                                                       int *updateSuccessCount = nullptr) = 0;
    // The same loop with parameters.batchVisualUpdate -- or on a frame that is not a "full visual update" -- (backend.cpp:1001-1010,
    // 1169-1183, 1255-1262): inliers' blocks are collected and applied as ONE updateVisualTrack per batch of at most maxUpdateRows rows
    // (= int(stateDim * batchVisualUpdateMaxSizeMultiplier + 0.5); <= 0: stateDim); every track between two flushes is gated against the
    // same state. hv_ekf_visual_frame_batch (ABI 4).
    virtual std::vector<VisualTrackResult> visualFrameBatch(const hv_vu_params &parameters, const std::vector<VisualFrameTrack> &tracks,
                                                            double chiOutlierR, double visualR, int maxSuccessfulVisualUpdates,
                                                            int maxUpdateRows, int *updateSuccessCount = nullptr) = 0;
    // One track visit with the hybrid-map branches of the loop (backend.cpp:1075-1082, 1146-1168): mapPointIndex >= 0 = the track's
    // point is map point mapPointIndex of the state (no triangulation: triangulateStatus HV_TRI_HYBRID); offeredMapPointIndex >= 0 = the
    // slot ekfStateIndex.offerMapPoint hands to this pose-trail track -- if the gate accepts it the point is INSERTED as that map point
    // (insertMapPoint) instead of being applied. Both -1: visualTrack. hv_ekf_visual_track_hybrid (ABI 4).
    virtual VisualTrackResult visualTrackHybrid(const hv_vu_params &parameters, const std::vector<int> &poseTrailIndex,
                                                const std::vector<double> &imageFeatures, const std::vector<double> &featureVelocities,
                                                const VectorXd &y, int mapPointIndex, int offeredMapPointIndex, double chiOutlierR,
                                                double visualR) = 0;
    // maintainPositiveSemiDefinite() followed by updateVisualPoseAugmentation(discardedPoseIndex), the end of every frame
    // (backend.cpp:1267, 804-805), as one pass over the covariance (hv_ekf_symmetrize_augment, ABI 4); bit-identical to the two calls.
    virtual void symmetrizeAugment(int discardedPoseIndex = -1) = 0;
};

}  // namespace odometry
}  // namespace hybvio
