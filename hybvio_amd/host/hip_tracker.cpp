// HIP-backed tracker::ImagePyramid::Factory and tracker::OpticalFlow (see hybvio_host.hpp).
// Counterparts of CpuImagePyramidFactory (src/tracker/image_pyramid.cpp:27-49) and
// OpenCvOpticalFlow (src/tracker/optical_flow.cpp:61-103).
#include <cassert>
#include <cstdio>
#include <stdexcept>

#include "hybvio_host.hpp"

namespace hybvio {

Session::Session(const hv_params &params) : params_(params)
{
    const int rc = hv_create(&params_, &ctx_);
    if (rc != HV_OK) throw std::runtime_error(std::string("hv_create: ") + hv_status_string(rc));
}

Session::~Session() { if (owned_) hv_destroy(ctx_); }

Lanes::Lanes(const hv_params &params, int n)
{
    const int rc = hv_lanes_create(&params, n, &lanes_);
    if (rc != HV_OK) throw std::runtime_error(std::string("hv_lanes_create: ") + hv_status_string(rc));
    for (int i = 0; i < hv_lanes_count(lanes_); ++i) sessions_.emplace_back(new Session(hv_lanes_ctx(lanes_, i), params));
}

Lanes::~Lanes()
{
    sessions_.clear();                     // (the borrowed contexts die with the set)
    hv_lanes_destroy(lanes_);
}

void Session::check(int rc, const char *what) const
{
    if (rc == HV_OK) return;
    std::string msg = std::string(what) + ": " + hv_status_string(rc);
    if (rc == HV_ERR_HIP) msg += std::string(" (") + hv_last_error(ctx_) + ")";
    throw DeviceError(rc, msg);
}

namespace tracker {

ImagePyramid::~ImagePyramid() = default;
ImagePyramid::Factory::~Factory() = default;
OpticalFlow::~OpticalFlow() = default;
FeatureDetector::~FeatureDetector() = default;
Undistorter::~Undistorter() = default;
rot_ransac::RotRansac::~RotRansac() = default;

void FeatureDetector::applyMinDistance(std::vector<Feature::Point> &corners, const std::vector<Feature::Point> &prevCorners,
                                       int minDistance) const
{
    static_assert(sizeof(Feature::Point) == 2 * sizeof(float), "Point must be two packed floats");
    int n = (int)corners.size();
    hv_apply_min_distance(reinterpret_cast<float *>(corners.data()), &n,
                          reinterpret_cast<const float *>(prevCorners.data()), (int)prevCorners.size(),
                          minDistance, parameters.maxTracks);
    corners.resize((size_t)n);
}

namespace {

// One pooled device slot. util::Allocator (src/util/allocator.hpp:55-67) recycles a pyramid when
// only the pool still references it; here the shared_ptr's last owner returns the slot to the
// device pool, which has the same effect: a pyramid lives exactly as long as an Image refers to it.
struct HipImagePyramid : ImagePyramid {
    Session &session;
    hv_ctx *ctx;
    int slot;
    HipImagePyramid(Session &s, int sl) : session(s), ctx(s.ctx()), slot(sl) {}
    ~HipImagePyramid() override { hv_pyramid_release(ctx, slot); }
    int deviceSlot() const final { return slot; }

    std::vector<GrayType> getGrayLevel(std::size_t i, int &w, int &h) final {
        session.check(hv_pyramid_level_size(ctx, (int)i, &w, &h), "hv_pyramid_level_size");
        std::vector<GrayType> out((size_t)w * h);
        session.check(hv_pyramid_download(ctx, slot, (int)i, out.data(), nullptr), "hv_pyramid_download");
        return out;
    }
    std::vector<GradientType> getGradientLevel(std::size_t i, int &w, int &h) final {
        session.check(hv_pyramid_level_size(ctx, (int)i, &w, &h), "hv_pyramid_level_size");
        std::vector<GradientType> out((size_t)w * h * GRADIENT_CHANNELS);
        session.check(hv_pyramid_download(ctx, slot, (int)i, nullptr, out.data()), "hv_pyramid_download");
        return out;
    }
};

class HipImagePyramidFactory : public ImagePyramid::Factory {
    Session &session;
public:
    explicit HipImagePyramidFactory(Session &s) : session(s) {}
    std::shared_ptr<ImagePyramid> compute(const GrayImage &img) final {
        assert(img.width == session.params().width && img.height == session.params().height);
        int slot = -1;
        session.check(hv_pyramid_acquire(session.ctx(), &slot), "hv_pyramid_acquire");   // the pool grows on demand
        auto pyramid = std::make_shared<HipImagePyramid>(session, slot);
        session.check(hv_pyramid_build(session.ctx(), slot, img.data, img.strideBytes), "hv_pyramid_build");   // H2D + kernels, asynchronous
        // the caller's pixels were handed to an asynchronous copy: fence before they may be reused
        session.check(hv_synchronize(session.ctx()), "hv_synchronize");
        return pyramid;
    }
    std::shared_ptr<ImagePyramid> computeFromFrame(const InputImage &img) final {
        assert(img.width == session.params().width && img.height == session.params().height);
        int slot = -1;
        session.check(hv_pyramid_acquire(session.ctx(), &slot), "hv_pyramid_acquire");   // the pool grows on demand
        auto pyramid = std::make_shared<HipImagePyramid>(session, slot);
        session.check(hv_ingest_build(session.ctx(), slot, img.data, img.strideBytes, img.channels, -1), "hv_ingest_build");
        session.check(hv_synchronize(session.ctx()), "hv_synchronize");
        return pyramid;
    }
};

// Counterpart of UndistorterImplementation (src/tracker/undistorter.cpp:43-135). Like the reference's GPU branch
// (:113-121) the remap is built once, from the first camera seen; later per-frame intrinsics are ignored with a
// warning. The table is the reference's own per-pixel evaluation (undistortCpu, :56-60) done once on the host.
class HipUndistorter : public Undistorter {
    Session &session;
    const int cameraIndex;
    std::shared_ptr<const Camera> originalCamera, undistortedCamera;
public:
    HipUndistorter(Session &s, int idx, std::shared_ptr<const Camera> rectified)
        : session(s), cameraIndex(idx), undistortedCamera(std::move(rectified)) {}
    ~HipUndistorter() override { hv_ingest_set_undistort_map(session.ctx(), cameraIndex, nullptr, nullptr); }

    Result undistort(const InputImage &image, std::shared_ptr<const Camera> camera) final {
        const int w = session.params().width, h = session.params().height;
        assert(image.width == w && image.height == h);
        if (originalCamera) {
            if (originalCamera->getFocalLength() - camera->getFocalLength() > 1e-6)          // undistorter.cpp:114-117
                std::fprintf(stderr, "Per-frame camera parameters ignored in HIP undistortion\n");
        } else {
            originalCamera = camera;
            std::vector<double> pixOrig((size_t)w * h * 2, 0.0);
            std::vector<std::uint8_t> valid((size_t)w * h, 0);
            for (int y = 0; y < h; ++y)
                for (int x = 0; x < w; ++x) {
                    const double pixRect[2] = {(double)x, (double)y};
                    double ray[3], *out = &pixOrig[2 * ((size_t)y * w + x)];
                    valid[(size_t)y * w + x] = undistortedCamera->pixelToRay(pixRect, ray) && originalCamera->rayToPixel(ray, out);
                }
            session.check(hv_ingest_set_undistort_map(session.ctx(), cameraIndex, pixOrig.data(), valid.data()), "hv_ingest_set_undistort_map");
        }
        int slot = -1;
        session.check(hv_pyramid_acquire(session.ctx(), &slot), "hv_pyramid_acquire");   // the pool grows on demand
        auto pyramid = std::make_shared<HipImagePyramid>(session, slot);
        session.check(hv_ingest_build(session.ctx(), slot, image.data, image.strideBytes, image.channels, cameraIndex), "hv_ingest_build");
        session.check(hv_synchronize(session.ctx()), "hv_synchronize");
        return {undistortedCamera, pyramid};
    }
};

class HipOpticalFlow : public OpticalFlow {
    Session &session;
    std::vector<std::int32_t> workStatus;
public:
    explicit HipOpticalFlow(Session &s) : session(s) {}
    void compute(ImagePyramid &prevImagePyramid, ImagePyramid &imagePyramid,
                 const std::vector<Feature::Point> &prevCorners, std::vector<Feature::Point> &corners,
                 std::vector<Feature::Status> &trackStatus, bool useInitialCorners,
                 int overrideMaxIterations) final {
        // optical_flow.cpp:26-40: outputs are cleared and sized by the callee; empty input returns early
        const size_t n = prevCorners.size();
        trackStatus.clear();
        trackStatus.resize(n, Feature::Status::FAILED_FLOW);
        if (n == 0) { corners.clear(); return; }
        if (!useInitialCorners) corners.assign(n, Feature::Point{0.f, 0.f});
        assert(corners.size() == n);
        workStatus.assign(n, 2);
        static_assert(sizeof(Feature::Point) == 2 * sizeof(float), "Point must be two packed floats (optical_flow.cpp:21-22)");
        session.check(hv_optical_flow_compute(session.ctx(), prevImagePyramid.deviceSlot(), imagePyramid.deviceSlot(),
                                               (int)n, reinterpret_cast<const float *>(prevCorners.data()),
                                               reinterpret_cast<float *>(corners.data()), workStatus.data(),
                                               useInitialCorners ? 1 : 0, overrideMaxIterations), "hv_optical_flow_compute");
        for (size_t i = 0; i < n; ++i) trackStatus[i] = static_cast<Feature::Status>(workStatus[i]);
    }
};

// Counterpart of FeatureDetectorImplementation (src/tracker/feature_detector.cpp:566-652) on its
// CPU-fallback path: corner response + block arg-max on the device, the reference's sort /
// zero-prefix / applyMinDistance tail inside hv_gftt_detect.
class HipFeatureDetector : public FeatureDetector {
    Session &session;
public:
    HipFeatureDetector(Session &s, const hv_gftt_params &p) : FeatureDetector(p), session(s) {}
    void detect(ImagePyramid &imagePyramid, std::vector<Feature::Point> &corners,
                const std::vector<Feature::Point> &prevCorners, int maskRadius) final {
        const int nk = hv_gftt_keypoint_count(session.ctx(), &parameters);
        assert(nk >= 0);
        corners.assign((size_t)(2 * nk), Feature::Point{0.f, 0.f});
        int n = 0;
        session.check(hv_gftt_detect(session.ctx(), &parameters, imagePyramid.deviceSlot(),
                                      reinterpret_cast<const float *>(prevCorners.data()), (int)prevCorners.size(),
                                      maskRadius, reinterpret_cast<float *>(corners.data()), 2 * nk, &n), "hv_gftt_detect");
        corners.resize((size_t)n);
    }
};

}  // namespace

namespace {
class HipRotRansac : public rot_ransac::RotRansac {
    Session &session;
    std::vector<int> pairs, status;
public:
    explicit HipRotRansac(Session &s) : session(s), pairs(200) {}
    std::array<float, 9> fit(const std::vector<Feature::Point> &c1, const std::vector<Feature::Point> &c2,
                             const hv_camera_model &camera1, const hv_camera_model &camera2,
                             std::vector<Feature::Status> &bestInliers, std::mt19937 &rng) final {
        assert(c1.size() == c2.size());
        const std::size_t n = c1.size();
        assert(n >= 2 && bestInliers.size() >= n);
        // draw the 100 index pairs from a COPY of the generator, then advance the real one by what the reference loop
        // would have consumed (it stops at the first hypothesis that makes every point an inlier)
        std::mt19937 ahead = rng;
        for (int k = 0; k < 100; ++k) {
            pairs[2 * k] = static_cast<int>(ahead() % n);
            pairs[2 * k + 1] = static_cast<int>(ahead() % n);
        }
        status.assign(n, 3);
        std::array<float, 9> R{};
        int best = 0, visited = 0;
        static_assert(sizeof(Feature::Point) == 2 * sizeof(float), "Point must be two packed floats");
        session.check(hv_rot_ransac(session.ctx(), static_cast<int>(n), reinterpret_cast<const float *>(c1.data()),
                                     reinterpret_cast<const float *>(c2.data()), &camera1, &camera2, pairs.data(), threshold_pow2,
                                     status.data(), R.data(), &best, &visited), "hv_rot_ransac");
        rng.discard(2ull * static_cast<unsigned long long>(visited));
        bestInlierCount = static_cast<std::size_t>(best);
        for (std::size_t i = 0; i < n; ++i) bestInliers.at(i) = static_cast<Feature::Status>(status[i]);
        return R;
    }
};
}  // namespace

std::unique_ptr<rot_ransac::RotRansac> rot_ransac::RotRansac::buildHip(Session &s)
{
    return std::unique_ptr<rot_ransac::RotRansac>(new HipRotRansac(s));
}

std::unique_ptr<Undistorter> Undistorter::buildRectifiedHip(Session &s, int cameraIndex, std::shared_ptr<const Camera> rectified)
{
    assert(cameraIndex >= 0 && cameraIndex < HV_INGEST_CAMERAS);
    return std::unique_ptr<Undistorter>(new HipUndistorter(s, cameraIndex, std::move(rectified)));
}

std::unique_ptr<FeatureDetector> FeatureDetector::buildHip(Session &s, const hv_gftt_params &p)
{
    return std::unique_ptr<FeatureDetector>(new HipFeatureDetector(s, p));
}

std::unique_ptr<ImagePyramid::Factory> ImagePyramid::Factory::buildHip(Session &s)
{
    return std::unique_ptr<ImagePyramid::Factory>(new HipImagePyramidFactory(s));
}

std::unique_ptr<OpticalFlow> OpticalFlow::buildHip(Session &s)
{
    return std::unique_ptr<OpticalFlow>(new HipOpticalFlow(s));
}

}  // namespace tracker
}  // namespace hybvio
