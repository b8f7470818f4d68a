// HIP-backed odometry::EKF (see hybvio_host.hpp). Counterpart of EKFImplementation
// (src/odometry/ekf.cpp:86-1080): the dense covariance work runs on the device through hv_ekf_*;
// this class keeps what the reference keeps in scalars (sample clock, rate limits, augmentTimes)
// and a host mirror of the mean, refreshed lazily (getState() is read by the backend between
// every two updates: triangulation.cpp:86-87, backend.cpp:579-609).
#include <cassert>
#include <cmath>
#include <iomanip>
#include <sstream>
#include <stdexcept>

#include "hybvio_host.hpp"

namespace hybvio {
namespace odometry {

EKF::~EKF() = default;

namespace {

inline double pow2(double x) { return x * x; }

struct HipEKF : public EKF {
    Session &session;
    hv_ekf_params par;
    hv_ekf *ekf = nullptr;
    const double noiseScale;
    const int camPoseCount, hybridMapDim, stateDim;

    // host mirrors (mutable: refreshed inside const getters, like a cache)
    mutable VectorXd m;
    mutable MatrixXd P;
    mutable bool meanFresh = false, covFresh = false;

    int augmentCount = 0;
    std::vector<double> augmentTimes;
    double time = 0.0, ZUPTtime = -1.0, ZRUPTtime = -1.0, initZUPTtime = -1.0;
    bool wasStationary = false;
    double prevSampleT = -1.0, firstSampleT = -1.0;
    bool firstSample = true;

    HipEKF(Session &s, const hv_ekf_params &p)
        : session(s), par(p), noiseScale(p.noiseScale * p.noiseScale), camPoseCount(p.cameraTrailLength),
          hybridMapDim(p.hybridMapSize * MAP_POINT_DIM),
          stateDim(INER_DIM + p.cameraTrailLength * POSE_DIM + p.hybridMapSize * MAP_POINT_DIM)
    {
        const int rc = hv_ekf_create(session.ctx(), &par, 1, &ekf);
        if (rc != HV_OK) throw std::runtime_error(std::string("hv_ekf_create: ") + hv_status_string(rc));
        m.assign(stateDim, 0.0);
        P = MatrixXd(stateDim, stateDim);
    }
    ~HipEKF() override { hv_ekf_destroy(ekf); }

    void check(int rc) const { session.check(rc, "hv_ekf"); }   // throws DeviceError, also under NDEBUG
    mutable std::vector<double> qdt, qg, qa;            // queued IMU samples (dt, gyro, acc)
    void flushPredicts() const {
        if (qdt.empty()) return;
        check(hv_ekf_predict_n(ekf, (int)qdt.size(), qdt.data(), qg.data(), qa.data()));
        qdt.clear(); qg.clear(); qa.clear();
    }
    hv_ekf *dev() const { flushPredicts(); return ekf; }  // every device call goes through here
    void dirty() { meanFresh = covFresh = false; }
    void syncMean() const { if (!meanFresh) { check(hv_ekf_get_state(dev(), 0, m.data(), nullptr)); meanFresh = true; } }
    void syncAll() const {
        if (!meanFresh || !covFresh) { check(hv_ekf_get_state(dev(), 0, m.data(), P.data.data())); meanFresh = covFresh = true; }
    }
    void pushAll() { check(hv_ekf_set_state(dev(), 0, m.data(), P.data.data())); meanFresh = covFresh = true; }
    void pushMean() { check(hv_ekf_set_state(dev(), 0, m.data(), nullptr)); meanFresh = true; }

    std::unique_ptr<EKF> clone() const final {
        std::unique_ptr<HipEKF> c(new HipEKF(session, par));
        syncAll();
        c->m = m; c->P = P; c->pushAll();
        double Q[Q_DIM * Q_DIM];    // process noise follows the filter (drift entries change in predict)
        check(hv_ekf_get_process_noise(dev(), 0, Q));
        check(hv_ekf_set_process_noise(c->ekf, 0, Q));
        c->augmentCount = augmentCount; c->augmentTimes = augmentTimes;
        c->time = time; c->ZUPTtime = ZUPTtime; c->ZRUPTtime = ZRUPTtime; c->initZUPTtime = initZUPTtime;
        c->wasStationary = wasStationary; c->prevSampleT = prevSampleT; c->firstSampleT = firstSampleT;
        c->firstSample = firstSample;
        return std::unique_ptr<EKF>(c.release());
    }

    // ekf.cpp:298-317 (Eigen::Quaterniond::FromTwoVectors(-gravity, xa))
    void initializeOrientation(const Vector3d &xa) final {
        syncAll();
        double a[3] = {0, 0, par.gravity}, b[3] = {xa[0], xa[1], xa[2]};
        const double na = std::sqrt(a[0]*a[0] + a[1]*a[1] + a[2]*a[2]), nb = std::sqrt(b[0]*b[0] + b[1]*b[1] + b[2]*b[2]);
        for (int i = 0; i < 3; i++) { a[i] /= na; b[i] /= nb; }
        const double c = a[0]*b[0] + a[1]*b[1] + a[2]*b[2];
        double q[4];
        if (c < -1.0 + 1e-12) { q[0] = 0; q[1] = 1; q[2] = 0; q[3] = 0; }   // antiparallel: any half turn about an orthogonal axis
        else {
            const double ax[3] = {a[1]*b[2] - a[2]*b[1], a[2]*b[0] - a[0]*b[2], a[0]*b[1] - a[1]*b[0]};
            const double s = std::sqrt((1.0 + c) * 2.0), invs = 1.0 / s;
            q[0] = s * 0.5; q[1] = ax[0] * invs; q[2] = ax[1] * invs; q[3] = ax[2] * invs;
        }
        for (int i = 0; i < 4; i++) m[ORI + i] = q[i];
        assert(q[3] == 0);
        const double v = pow2(par.noiseInitialOri) * noiseScale;
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) P(ORI + i, ORI + j) = 0.0;
        for (int i = 0; i < 3; i++) P(ORI + i, ORI + i) = v;
        pushAll();
    }

    // ekf.cpp:320-514: the clock stays here, mean + Jacobians + covariance run on the device
    // The device call is deferred: consecutive IMU samples (10 between two EuRoC frames) are queued and
    // go out as ONE hv_ekf_predict_n launch when anything else touches the filter (dev()).
    void predict(double t, const Vector3d &xg, const Vector3d &xa) final {
        double dt = 0.0;
        if (!firstSample) { dt = t - prevSampleT; time = t - firstSampleT; }
        else { firstSampleT = t; firstSample = false; }
        prevSampleT = t;
        if (dt <= 0.0) return;
        qdt.push_back(dt);
        for (int i = 0; i < 3; i++) { qg.push_back(xg[i]); qa.push_back(xa[i]); }
        dirty();
        if ((int)qdt.size() >= HV_EKF_MAX_PREDICT_SAMPLES) flushPredicts();
    }

    Vector3d seg3(int o) const { syncMean(); return {m[o], m[o + 1], m[o + 2]}; }
    Vector4d seg4(int o) const { syncMean(); return {m[o], m[o + 1], m[o + 2], m[o + 3]}; }
    Vector3d position() const final { return seg3(POS); }
    Vector3d velocity() const final { return seg3(VEL); }
    Vector4d orientation() const final { return seg4(ORI); }
    Vector3d biasGyroscopeAdditive() const final { return seg3(BGA); }
    Vector3d biasAccelerometerAdditive() const final { return seg3(BAA); }
    Vector3d biasAccelerometerTransform() const final { return seg3(BAT); }
    int camTrailSize() const final { return camPoseCount; }
    Vector3d historyPosition(int i) const final {
        if (i == -1) return position();
        assert(i >= 0 && i < camPoseCount);
        return seg3(CAM + POSE_DIM * i);
    }
    Vector4d historyOrientation(int i) const final {
        if (i == -1) return orientation();
        assert(i >= 0 && i < camPoseCount);
        return seg4(CAM + POSE_DIM * i + 3);
    }
    double historyTime(int i) const final {
        if (i == -1) return getPlatformTime();
        assert(i >= 0 && i < camPoseCount);
        const size_t n = augmentTimes.size();
        assert(i < static_cast<int>(n));
        return augmentTimes[n - i - 1];
    }
    double speed() const final { const Vector3d v = velocity(); return std::sqrt(v[0]*v[0] + v[1]*v[1] + v[2]*v[2]); }
    double horizontalSpeed() const final { const Vector3d v = velocity(); return std::sqrt(v[0]*v[0] + v[1]*v[1]); }

    // update(m,P,y,H,R,...) with an identity block H (ekf.cpp:573-677)
    void blockUpdate(int rows, int col0, const double *y, double rdiag, bool normalizeAll) {
        const int l = col0 + rows;
        std::vector<double> H((size_t)rows * l, 0.0);
        for (int i = 0; i < rows; i++) H[(size_t)(col0 + i) * rows + i] = 1.0;
        check(hv_ekf_update(dev(), rows, l, H.data(), y, &rdiag, nullptr, normalizeAll ? 1 : 0));
        dirty();
    }
    void updateZupt(double r) final {
        if (time - ZUPTtime < 0.25) return;
        ZUPTtime = time; wasStationary = true;
        const double y[3] = {0, 0, 0};
        blockUpdate(3, VEL, y, r * noiseScale, false);
    }
    void updateZuptInitialization() final {
        if (wasStationary || time > 60 || time - initZUPTtime < 0.1) return;
        initZUPTtime = time;
        const double y[3] = {0, 0, 0};
        blockUpdate(3, VEL, y, par.initZuptR * noiseScale * std::exp(0.5 * time), false);
    }
    void updateZrupt(const Vector3d &xg) final {
        if (time - ZRUPTtime < 0.25) return;
        ZRUPTtime = time;
        blockUpdate(3, BGA, xg.data(), par.rotationZuptR * noiseScale, false);
    }
    // ekf.cpp:628-649: scalar update on the horizontal speed; y - H m = defaultSpeed - |v_xy|
    void updatePseudoVelocity(double defaultSpeed, double r) final {
        syncMean();
        const double h = std::sqrt(m[VEL]*m[VEL] + m[VEL + 1]*m[VEL + 1]);
        if (h <= 1e-7) return;
        double H[VEL + 2] = {0};
        for (int i = 0; i < 2; i++) H[VEL + i] = m[VEL + i] / h;
        const double rd = r * noiseScale;
        check(hv_ekf_update(dev(), 1, VEL + 2, H, &defaultSpeed, &rd, nullptr, 0));
        dirty();
    }
    void updatePosition(const Vector3d &pos, double r) final {
        blockUpdate(3, POS, pos.data(), r * noiseScale, false);
        maintainPositiveSemiDefinite();
    }
    void updateZeroHeight(double r) final {
        const double H[POS + 3] = {0, 0, 1}, y = 0, rd = r * noiseScale;
        check(hv_ekf_update(dev(), 1, POS + 3, H, &y, &rd, nullptr, 0));
        dirty();
        maintainPositiveSemiDefinite();
    }
    void updateOrientation(const Vector4d &q, double r) final {
        blockUpdate(4, ORI, q.data(), r * noiseScale, true);   // + normalizeQuaternions()
        maintainPositiveSemiDefinite();
    }

    void getInertialState(VectorInertialMean &mean, MatrixInertialCov &cov) const final {
        syncAll();
        mean.assign(m.begin(), m.begin() + INER_DIM);
        cov = MatrixXd(INER_DIM, INER_DIM);
        for (int j = 0; j < INER_DIM; j++) for (int i = 0; i < INER_DIM; i++) cov(i, j) = P(i, j);
    }
    void setInertialState(const VectorInertialMean &mean, const MatrixInertialCov &cov) final {
        syncAll();
        for (int i = 0; i < INER_DIM; i++) m[i] = mean[i];
        for (int j = 0; j < INER_DIM; j++) for (int i = 0; i < INER_DIM; i++) P(i, j) = cov(i, j);
        pushAll();
        augmentCount = 0; augmentTimes.clear();
    }
    double getImuToCameraTimeShift() const final { syncMean(); return m[SFT]; }

    void translateTo(const Vector3d &pos) final {
        syncMean();
        const double d[3] = {pos[0] - m[POS], pos[1] - m[POS + 1], pos[2] - m[POS + 2]};
        for (int i = 0; i < 3; i++) m[POS + i] += d[i];
        for (int c = 0; c < camPoseCount; c++) for (int i = 0; i < 3; i++) m[CAM + POSE_DIM * c + i] += d[i];
        pushMean();
    }

    // ekf.cpp:704-758: the 3x3 / 4x4 change matrices and the translation come from the host mirror;
    // the block-diagonal similarity P = A P A' runs on the device
    void transformTo(const Vector3d &pos, const Vector4d &q, int i) final {
        const Vector4d q0 = i < 0 ? orientation() : historyOrientation(i);
        const double a[4] = {q0[0], -q0[1], -q0[2], -q0[3]};                 // conj(quat0) * quat1
        const double p1 = a[0]*q[0] - a[1]*q[1] - a[2]*q[2] - a[3]*q[3];
        const double p2 = a[0]*q[1] + a[1]*q[0] + a[2]*q[3] - a[3]*q[2];
        const double p3 = a[0]*q[2] - a[1]*q[3] + a[2]*q[0] + a[3]*q[1];
        const double p4 = a[0]*q[3] + a[1]*q[2] - a[2]*q[1] + a[3]*q[0];
        const double qC[16] = {p1, -p2, -p3, -p4,  p2, p1, p4, -p3,  p3, -p4, p1, p2,  p4, p3, -p2, p1};
        const double w = p1, x = p2, y = p3, z = p4;
        const double R[9] = {1 - 2*(y*y + z*z), 2*(x*y - z*w), 2*(x*z + y*w),
                             2*(x*y + z*w), 1 - 2*(x*x + z*z), 2*(y*z - x*w),
                             2*(x*z - y*w), 2*(y*z + x*w), 1 - 2*(x*x + y*y)};
        double pC[9];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) pC[3 * r + c] = R[3 * c + r];   // transpose
        const Vector3d refPos = i < 0 ? position() : historyPosition(i);
        double tr[3];
        for (int r = 0; r < 3; r++) tr[r] = pos[r] - (pC[3*r]*refPos[0] + pC[3*r + 1]*refPos[1] + pC[3*r + 2]*refPos[2]);
        check(hv_ekf_transform(dev(), 0, pC, qC, tr));
        dirty();
    }

    // gate status of the device entries (include/hybvio_hip.h: 0 inlier, 1 not computed, 2 RMSE, 3 chi2) -> the reference's enum; ONE
    // mapping for visualTrack, visualTrackHybrid and frameLoop (r05 advisor: two of the three reported an RMSE rejection as NOT_COMPUTED)
    static VuOutlierStatus outlierStatusOfGate(int gate) {
        return gate == 0 ? VuOutlierStatus::INLIER : gate == 3 ? VuOutlierStatus::CHI2 : gate == 2 ? VuOutlierStatus::RMSE : VuOutlierStatus::NOT_COMPUTED;
    }

    VuOutlierStatus visualTrackOutlierCheck(const MatrixXd &visH, const VectorXd &f, const VectorXd &y, double r,
                                            double trackRmseThreshold) final {
        const int n = visH.rows;
        assert(static_cast<int>(y.size()) == n && static_cast<int>(f.size()) == n && n > 0 && visH.cols > 0);
        VectorXd v(n);
        for (int i = 0; i < n; i++) v[i] = y[i] - f[i];
        if (trackRmseThreshold >= 0.0) {
            double s = 0; for (int i = 0; i < n; i++) s += v[i] * v[i];
            if (std::sqrt(s / n) > trackRmseThreshold) return VuOutlierStatus::RMSE;
        }
        if (r < 0.0) return VuOutlierStatus::INLIER;
        double chi2 = 0; int status = 0;
        check(hv_ekf_visual_gate(dev(), n, visH.cols, visH.data.data(), v.data(), r, &chi2, &status));
        return status == 3 ? VuOutlierStatus::CHI2 : VuOutlierStatus::INLIER;
    }

    void updateVisualTrack(const MatrixXd &visH, const VectorXd &f, const VectorXd &y, double r) final {
        const int n = visH.rows;
        assert(static_cast<int>(y.size()) == n && static_cast<int>(f.size()) == n);
        VectorXd v(n);
        for (int i = 0; i < n; i++) v[i] = y[i] - f[i];
        check(hv_ekf_visual_update(dev(), n, visH.cols, visH.data.data(), v.data(), r, nullptr));
        dirty();
    }

    VisualTrackResult visualTrack(const hv_vu_params &parameters, const std::vector<int> &poseTrailIndex,
                                  const std::vector<double> &imageFeatures, const std::vector<double> &featureVelocities,
                                  const VectorXd &y, double chiOutlierR, double visualR) final {
        const size_t n = poseTrailIndex.size(), nt = n * (parameters.useStereo ? 2 : 1);
        assert(imageFeatures.size() == 2 * nt && featureVelocities.size() == 2 * nt && y.size() == 2 * nt);
        int status[2] = {0, 0}, gate = 1;
        VisualTrackResult res{};
        check(hv_ekf_visual_track(dev(), &parameters, (int)n, poseTrailIndex.data(), imageFeatures.data(), featureVelocities.data(),
                                  y.data(), chiOutlierR, visualR, status, &gate, nullptr, res.pf.data()));
        res.triangulateStatus = status[0]; res.prepareVuStatus = status[1];
        res.outlierStatus = outlierStatusOfGate(gate);
        if (gate == 0) dirty();                       // the filter was updated
        return res;
    }

    // the frame's tracks as the C ABI takes them: records padded to the longest track, the device gets every track's own pose count
    std::vector<VisualTrackResult> frameLoop(const hv_vu_params &parameters, const std::vector<VisualFrameTrack> &tracks, double chiOutlierR,
                                             double visualR, int maxSuccessfulVisualUpdates, int *updateSuccessCount, bool batched,
                                             int maxUpdateRows) {
        std::vector<VisualTrackResult> out(tracks.size());
        if (tracks.empty()) return out;
        const size_t K = tracks.size(), ncam = parameters.useStereo ? 2 : 1;
        size_t n = 0;
        for (const VisualFrameTrack &t : tracks) n = std::max(n, t.poseTrailIndex.size());
        const size_t nt = n * ncam;
        std::vector<int> idx(K * n, 0), status(2 * K), gate(K), lens(K);
        std::vector<double> feat(K * 2 * nt, 0.0), vel(K * 2 * nt, 0.0), y(K * 2 * nt, 0.0), pf(3 * K);
        for (size_t k = 0; k < K; ++k) {
            const VisualFrameTrack &t = tracks[k];
            const size_t nk = t.poseTrailIndex.size(), ntk = nk * ncam;
            assert(t.imageFeatures.size() == 2 * ntk && t.featureVelocities.size() == 2 * ntk && t.y.size() == 2 * ntk);
            lens[k] = (int)nk;
            std::copy(t.poseTrailIndex.begin(), t.poseTrailIndex.end(), idx.begin() + k * n);
            std::copy(t.imageFeatures.begin(), t.imageFeatures.end(), feat.begin() + k * 2 * nt);
            std::copy(t.featureVelocities.begin(), t.featureVelocities.end(), vel.begin() + k * 2 * nt);
            std::copy(t.y.begin(), t.y.end(), y.begin() + k * 2 * nt);
        }
        int applied = 0;
        if (batched)
            check(hv_ekf_visual_frame_batch(dev(), &parameters, (int)K, (int)n, lens.data(), idx.data(), feat.data(), vel.data(), y.data(),
                                            chiOutlierR, visualR, status.data(), gate.data(), nullptr, pf.data(), &applied,
                                            maxSuccessfulVisualUpdates, maxUpdateRows));
        else
            check(hv_ekf_visual_frame_ragged(dev(), &parameters, (int)K, (int)n, lens.data(), idx.data(), feat.data(), vel.data(), y.data(),
                                             chiOutlierR, visualR, status.data(), gate.data(), nullptr, pf.data(), &applied,
                                             maxSuccessfulVisualUpdates));
        for (size_t k = 0; k < K; ++k) {
            out[k].triangulateStatus = status[2 * k]; out[k].prepareVuStatus = status[2 * k + 1];
            out[k].outlierStatus = outlierStatusOfGate(gate[k]);
            for (int q = 0; q < 3; ++q) out[k].pf[q] = pf[3 * k + q];
        }
        if (updateSuccessCount) *updateSuccessCount = applied;
        if (applied > 0) dirty();
        return out;
    }
    std::vector<VisualTrackResult> visualFrame(const hv_vu_params &parameters, const std::vector<VisualFrameTrack> &tracks,
                                               double chiOutlierR, double visualR, int maxSuccessfulVisualUpdates,
                                               int *updateSuccessCount) final {
        return frameLoop(parameters, tracks, chiOutlierR, visualR, maxSuccessfulVisualUpdates, updateSuccessCount, false, 0);
    }
    std::vector<VisualTrackResult> visualFrameBatch(const hv_vu_params &parameters, const std::vector<VisualFrameTrack> &tracks,
                                                    double chiOutlierR, double visualR, int maxSuccessfulVisualUpdates, int maxUpdateRows,
                                                    int *updateSuccessCount) final {
        return frameLoop(parameters, tracks, chiOutlierR, visualR, maxSuccessfulVisualUpdates, updateSuccessCount, true, maxUpdateRows);
    }

    VisualTrackResult visualTrackHybrid(const hv_vu_params &parameters, const std::vector<int> &poseTrailIndex,
                                        const std::vector<double> &imageFeatures, const std::vector<double> &featureVelocities,
                                        const VectorXd &y, int mapPointIndex, int offeredMapPointIndex, double chiOutlierR,
                                        double visualR) final {
        const size_t n = poseTrailIndex.size(), nt = n * (parameters.useStereo ? 2 : 1);
        assert(imageFeatures.size() == 2 * nt && featureVelocities.size() == 2 * nt && y.size() == 2 * nt);
        assert(mapPointIndex < par.hybridMapSize && offeredMapPointIndex < par.hybridMapSize);
        int status[2] = {0, 0}, gate = 1;
        VisualTrackResult res{};
        check(hv_ekf_visual_track_hybrid(dev(), &parameters, (int)n, poseTrailIndex.data(), imageFeatures.data(), featureVelocities.data(),
                                         y.data(), par.hybridMapSize > 0 ? &mapPointIndex : nullptr,
                                         par.hybridMapSize > 0 ? &offeredMapPointIndex : nullptr, chiOutlierR, visualR, status, &gate,
                                         nullptr, res.pf.data()));
        res.triangulateStatus = status[0]; res.prepareVuStatus = status[1];
        res.outlierStatus = outlierStatusOfGate(gate);
        if (gate == 0) dirty();                       // applied, or inserted as a map point
        return res;
    }

    void symmetrizeAugment(int discardedPoseIndex) final {
        check(hv_ekf_symmetrize_augment(dev(), &discardedPoseIndex, nullptr));
        dirty();
        augmentTimes.push_back(getPlatformTime());
        if (augmentCount < camPoseCount) augmentCount++;
        else augmentTimes.erase(augmentTimes.begin());
        assert(static_cast<int>(augmentTimes.size()) == augmentCount);
    }

    void updateVisualPoseAugmentation(int discardedPoseIndex) final {
        check(hv_ekf_augment(dev(), &discardedPoseIndex, nullptr));
        dirty();
        augmentTimes.push_back(getPlatformTime());
        if (augmentCount < camPoseCount) augmentCount++;
        else augmentTimes.erase(augmentTimes.begin());
        assert(static_cast<int>(augmentTimes.size()) == augmentCount);
    }
    void updateUndoAugmentation() final {
        check(hv_ekf_undo_augment(dev(), nullptr));
        dirty();
        assert(augmentCount > 0);
        augmentTimes.pop_back();
        augmentCount--;
    }

    Vector3d getMapPoint(int idx) const final {               // ekf.cpp:905-909: read from the resident mean
        assert(idx >= 0 && getMapPointStateIndex(idx) + MAP_POINT_DIM <= stateDim);
        Vector3d pf{};
        check(hv_ekf_get_map_point(dev(), 0, idx, pf.data()));
        return pf;
    }
    void insertMapPoint(int idx, const Vector3d &pf) final {   // ekf.cpp:911-921, on the resident state (ABI 4)
        assert(idx >= 0 && getMapPointStateIndex(idx) + MAP_POINT_DIM <= stateDim);
        check(hv_ekf_insert_map_point(dev(), 0, idx, pf.data()));
        dirty();
    }
    int getMapPointStateIndex(int idx) const final {
        if (idx == -1) return -1;
        return stateDim - hybridMapDim + idx * MAP_POINT_DIM;
    }

    void conditionOnLastPose() final {   // ekf.cpp:928-942: Schur complement on the last pose (rare: done on the host)
        assert(hybridMapDim == 0 && augmentCount > 0);
        syncAll();
        const int mm = stateDim - POSE_DIM;
        double B[POSE_DIM][POSE_DIM], Binv[POSE_DIM][POSE_DIM];
        for (int i = 0; i < POSE_DIM; i++) for (int j = 0; j < POSE_DIM; j++) { B[i][j] = P(mm + i, mm + j); Binv[i][j] = i == j; }
        for (int c = 0; c < POSE_DIM; c++) {       // Gauss-Jordan with partial pivoting
            int p = c; for (int r = c + 1; r < POSE_DIM; r++) if (std::fabs(B[r][c]) > std::fabs(B[p][c])) p = r;
            for (int j = 0; j < POSE_DIM; j++) { std::swap(B[c][j], B[p][j]); std::swap(Binv[c][j], Binv[p][j]); }
            const double d = B[c][c];
            for (int j = 0; j < POSE_DIM; j++) { B[c][j] /= d; Binv[c][j] /= d; }
            for (int r = 0; r < POSE_DIM; r++) if (r != c) {
                const double f = B[r][c];
                for (int j = 0; j < POSE_DIM; j++) { B[r][j] -= f * B[c][j]; Binv[r][j] -= f * Binv[c][j]; }
            }
        }
        MatrixXd X(POSE_DIM, mm);                   // B^-1 * P(m.., 0..m)
        for (int j = 0; j < mm; j++) for (int i = 0; i < POSE_DIM; i++) { double s = 0; for (int k = 0; k < POSE_DIM; k++) s += Binv[i][k] * P(mm + k, j); X(i, j) = s; }
        MatrixXd N(mm, mm);
        for (int j = 0; j < mm; j++) for (int i = 0; i < mm; i++) { double s = 0; for (int k = 0; k < POSE_DIM; k++) s += P(i, mm + k) * X(k, j); N(i, j) = P(i, j) - s; }
        for (int j = 0; j < mm; j++) for (int i = 0; i < mm; i++) P(i, j) = N(i, j);
        for (int k = 0; k < POSE_DIM; k++) for (int i = 0; i < mm; i++) { P(i, mm + k) = 0; P(mm + k, i) = 0; }
        for (int i = 0; i < POSE_DIM; i++) for (int j = 0; j < POSE_DIM; j++) P(mm + i, mm + j) = i == j ? 1e6 : 0.0;
        pushAll();
    }
    void lockBiases() final {   // ekf.cpp:944-947
        syncAll();
        for (int k = BGA; k < BGA + 9; k++) for (int i = 0; i < stateDim; i++) { P(k, i) = 0; P(i, k) = 0; }
        pushAll();
    }
    void normalizeQuaternions(bool onlyCurrent) final { check(hv_ekf_normalize_quaternions(dev(), onlyCurrent ? 1 : 0)); meanFresh = false; }
    void setFirstSampleTime(double t) final { assert(t > 0.0); firstSample = false; firstSampleT = t; prevSampleT = t; time = t; }
    bool isPositiveSemiDefinite() final {   // debug only (ekf.cpp:1043-1057): sign of the pivots of a pivoted LDL'
        syncAll();
        MatrixXd A = P;
        const int n = stateDim;
        for (int j = 0; j < n; j++) for (int i = 0; i < j; i++) { const double s = 0.5 * (A(i, j) + A(j, i)); A(i, j) = A(j, i) = s; }
        double scale = 0; for (int i = 0; i < n; i++) scale = std::max(scale, std::fabs(A(i, i)));
        for (int k = 0; k < n; k++) {
            int p = k; for (int i = k + 1; i < n; i++) if (std::fabs(A(i, i)) > std::fabs(A(p, p))) p = i;
            if (p != k) { for (int j = 0; j < n; j++) std::swap(A(k, j), A(p, j)); for (int i = 0; i < n; i++) std::swap(A(i, k), A(i, p)); }
            const double d = A(k, k);
            if (d < -1e-12 * scale) return false;
            if (std::fabs(d) <= 1e-300) continue;
            for (int i = k + 1; i < n; i++) { const double f = A(i, k) / d; for (int j = k + 1; j < n; j++) A(i, j) -= f * A(k, j); }
        }
        return true;
    }
    void maintainPositiveSemiDefinite() final { check(hv_ekf_symmetrize(dev())); covFresh = false; }
    void setState(const VectorXd &_m) final { assert(static_cast<int>(_m.size()) == stateDim); m = _m; pushMean(); }
    void setStateCovariance(const MatrixXd &_P) final {
        assert(_P.rows == stateDim && _P.cols == stateDim);
        P = _P; check(hv_ekf_set_state(dev(), 0, nullptr, P.data.data())); covFresh = true;
    }
    void setProcessNoise(const MatrixXd &_Q) final { assert(_Q.rows == Q_DIM && _Q.cols == Q_DIM); check(hv_ekf_set_process_noise(dev(), 0, _Q.data.data())); }
    double getPlatformTime() const final { return firstSampleT + time; }
    int getPoseCount() const final { return augmentCount + 1; }
    const VectorXd &getState() const final { syncMean(); return m; }
    MatrixXd getStateCovariance() const final { syncAll(); return P; }
    const MatrixXd &getStateCovarianceRef() const final { syncAll(); return P; }
    MatrixXd getVisAugH() const final {   // ekf.cpp:267-278
        MatrixXd H(POSE_DIM, stateDim);
        for (int i = 0; i < 3; i++) { H(i, POS + i) = 1; H(i, CAM + i) = -1; }
        for (int i = 0; i < 4; i++) { H(3 + i, ORI + i) = 1; H(3 + i, CAM + 3 + i) = -1; }
        return H;
    }
    MatrixXd getVisAugA() const final {   // visAugA.back(): drops the last pose (ekf.cpp:230-248)
        MatrixXd A(stateDim, stateDim);
        const int d = camPoseCount - 1;
        for (int i = 0; i < CAM; i++) A(i, i) = 1;
        for (int i = CAM; i < CAM + d * POSE_DIM; ++i) A(i + POSE_DIM, i) = 1;
        for (int i = CAM + (d + 1) * POSE_DIM; i < stateDim; i++) A(i, i) = 1;
        return A;
    }
    MatrixXd getVisAugQ() const final {   // ekf.cpp:280-288
        MatrixXd Q(stateDim, stateDim);
        for (int i = CAM; i < CAM + 3; i++) Q(i, i) = pow2(par.noiseInitialPosTrail) * noiseScale;
        for (int i = CAM + 3; i < CAM + POSE_DIM; i++) Q(i, i) = pow2(par.noiseInitialOriTrail) * noiseScale;
        return Q;
    }
    MatrixXd getDydx() const final {      // ekf.cpp:991-995
        MatrixXd full(stateDim, stateDim);
        for (int i = 0; i < stateDim; i++) full(i, i) = 1.0;
        double F[INER_DIM * INER_DIM];
        check(hv_ekf_get_dydx(dev(), 0, F));
        for (int j = 0; j < INER_DIM; j++) for (int i = 0; i < INER_DIM; i++) full(i, j) = F[j * INER_DIM + i];
        return full;
    }
    std::string stateAsString() const final {   // ekf.cpp:998-1022
        syncAll();
        static const int parts[7] = {POS, VEL, ORI, BGA, BAA, BAT, SFT}, sizes[7] = {3, 3, 4, 3, 3, 3, 1};
        static const char *names[7] = {"POS", "VEL", "ORI", "BGA", "BAA", "BAT", "SFT"};
        std::stringstream ss;
        for (int i = 0; i < 7; i++) {
            ss << names[i] << " ";
            double vmax = 0;
            for (int j = 0; j < sizes[i]; j++) { ss << std::setprecision(3) << m[parts[i] + j] << " "; vmax = std::max(vmax, P(parts[i] + j, parts[i] + j)); }
            ss << std::setprecision(2) << " [" << std::sqrt(vmax) << "], ";
            if (i == 2) ss << std::endl << " ";
        }
        ss << std::fixed << std::setprecision(3) << "t " << time;
        return ss.str();
    }
    int getStateDim() const final { return stateDim; }
    bool getWasStationary() const final { return wasStationary; }
};

}  // namespace

std::unique_ptr<EKF> EKF::buildHip(Session &session, const hv_ekf_params &parameters)
{
    return std::unique_ptr<EKF>(new HipEKF(session, parameters));
}

}  // namespace odometry
}  // namespace hybvio
