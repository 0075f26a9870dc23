// PointCloud.h -- host-side value types of the C++ shell: the subset of PM::DataPoints /
// PM::TransformationParameters that norlab_icp_mapper touches (usage sites in the reference:
// norlab_icp_mapper/Map.cpp:160-175,213-225; MapperModules/*.cpp; Trajectory.cpp:17-51).
//
// Layout is the one the C ABI consumes directly: `features` is (dim+1) x N column-major float, i.e. N
// consecutive (x, y, z, 1) quadruples; each descriptor is span x N column-major.  Value semantics
// throughout (copies are deep), like PM::DataPoints.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace nim {

struct InvalidField : std::runtime_error { using std::runtime_error::runtime_error; };       // PM::DataPoints::InvalidField
struct InvalidParameter : std::runtime_error { using std::runtime_error::runtime_error; };   // PM::Parametrizable::InvalidParameter
struct ConvergenceError : std::runtime_error { using std::runtime_error::runtime_error; };   // PM::ConvergenceError
struct TransformationError : std::runtime_error { using std::runtime_error::runtime_error; };

// 4x4 column-major float matrix == Eigen::Matrix4f storage == float[16] of the C ABI
struct Mat4 {
    std::array<float, 16> m{};
    static Mat4 identity() { Mat4 r; r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.f; return r; }
    float& operator()(int row, int col) { return m[4 * col + row]; }
    float operator()(int row, int col) const { return m[4 * col + row]; }
    const float* data() const { return m.data(); }
    float* data() { return m.data(); }
    Mat4 operator*(const Mat4& b) const {
        Mat4 r;
        for (int j = 0; j < 4; ++j)
            for (int i = 0; i < 4; ++i) {
                float s = 0.f;
                for (int k = 0; k < 4; ++k) s += (*this)(i, k) * b(k, j);
                r(i, j) = s;
            }
        return r;
    }
    // inverse of an affine transform [A t; 0 1] through the adjugate of A (what pose.inverse() is used for)
    Mat4 inverse() const {
        const Mat4& a = *this;
        const double c00 = (double)a(1, 1) * a(2, 2) - (double)a(1, 2) * a(2, 1);
        const double c01 = (double)a(1, 2) * a(2, 0) - (double)a(1, 0) * a(2, 2);
        const double c02 = (double)a(1, 0) * a(2, 1) - (double)a(1, 1) * a(2, 0);
        const double det = a(0, 0) * c00 + a(0, 1) * c01 + a(0, 2) * c02;
        if (det == 0.0) throw TransformationError("singular transformation");
        const double id = 1.0 / det;
        double inv[3][3];
        inv[0][0] = c00 * id; inv[1][0] = c01 * id; inv[2][0] = c02 * id;
        inv[0][1] = ((double)a(0, 2) * a(2, 1) - (double)a(0, 1) * a(2, 2)) * id;
        inv[1][1] = ((double)a(0, 0) * a(2, 2) - (double)a(0, 2) * a(2, 0)) * id;
        inv[2][1] = ((double)a(0, 1) * a(2, 0) - (double)a(0, 0) * a(2, 1)) * id;
        inv[0][2] = ((double)a(0, 1) * a(1, 2) - (double)a(0, 2) * a(1, 1)) * id;
        inv[1][2] = ((double)a(0, 2) * a(1, 0) - (double)a(0, 0) * a(1, 2)) * id;
        inv[2][2] = ((double)a(0, 0) * a(1, 1) - (double)a(0, 1) * a(1, 0)) * id;
        Mat4 r = Mat4::identity();
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) r(i, j) = (float)inv[i][j];
            r(i, 3) = (float)-(inv[i][0] * a(0, 3) + inv[i][1] * a(1, 3) + inv[i][2] * a(2, 3));
        }
        return r;
    }
};

struct Descriptor {
    std::string name;
    int span = 1;
    std::vector<float> data; // span x N column-major
};

// a row group of PM::DataPoints::times: Eigen::Matrix<std::int64_t, Dynamic, Dynamic> rows + a Label (Trajectory.cpp:35-47 stores
// `time_since_epoch().count()`; a float descriptor of seconds would resolve 128 s at epoch scale -- VERDICT r5)
struct TimeField {
    std::string name;
    int span = 1;
    std::vector<std::int64_t> data; // span x N column-major
};

class DataPoints {
public:
    std::vector<float> features;          // 4 x N column-major (x, y, z, 1)
    std::vector<Descriptor> descriptors;
    std::vector<TimeField> times;         // int64 rows; concatenate / keepOnly / load / save treat them like descriptors

    DataPoints() = default;
    explicit DataPoints(size_t n) : features(4 * n, 0.f) { for (size_t i = 0; i < n; ++i) features[4 * i + 3] = 1.f; }

    size_t getNbPoints() const { return features.size() / 4; }
    int getEuclideanDim() const { return 3; }
    const float* col(size_t i) const { return &features[4 * i]; }
    float* col(size_t i) { return &features[4 * i]; }

    bool descriptorExists(const std::string& name) const { return findDescriptor(name) >= 0; }
    int findDescriptor(const std::string& name) const {
        for (size_t d = 0; d < descriptors.size(); ++d) if (descriptors[d].name == name) return (int)d;
        return -1;
    }
    const Descriptor& getDescriptorByName(const std::string& name) const {
        const int d = findDescriptor(name);
        if (d < 0) throw InvalidField("Cannot find descriptor " + name);
        return descriptors[d];
    }
    Descriptor& getDescriptorByName(const std::string& name) {
        const int d = findDescriptor(name);
        if (d < 0) throw InvalidField("Cannot find descriptor " + name);
        return descriptors[d];
    }
    // add or overwrite (PM::DataPoints::addDescriptor); data must be span x N
    void addDescriptor(const std::string& name, int span, std::vector<float> data) {
        if (data.size() != (size_t)span * getNbPoints()) throw InvalidField("descriptor " + name + " has the wrong size");
        const int d = findDescriptor(name);
        if (d >= 0) { descriptors[d].span = span; descriptors[d].data = std::move(data); }
        else descriptors.push_back(Descriptor{name, span, std::move(data)});
    }
    void removeDescriptor(const std::string& name) {
        const int d = findDescriptor(name);
        if (d >= 0) descriptors.erase(descriptors.begin() + d);
    }
    // PM::DataPoints::timeExists / getTimeViewByName / addTime
    int findTime(const std::string& name) const {
        for (size_t d = 0; d < times.size(); ++d) if (times[d].name == name) return (int)d;
        return -1;
    }
    bool timeExists(const std::string& name) const { return findTime(name) >= 0; }
    const TimeField& getTimeByName(const std::string& name) const {
        const int d = findTime(name);
        if (d < 0) throw InvalidField("Cannot find time " + name);
        return times[d];
    }
    void addTime(const std::string& name, int span, std::vector<std::int64_t> data) {
        if (data.size() != (size_t)span * getNbPoints()) throw InvalidField("time " + name + " has the wrong size");
        const int d = findTime(name);
        if (d >= 0) { times[d].span = span; times[d].data = std::move(data); }
        else times.push_back(TimeField{name, span, std::move(data)});
    }

    // same descriptor set, zero points (createSimilarEmpty)
    DataPoints createSimilarEmpty(size_t reserve = 0) const {
        DataPoints r;
        r.features.reserve(4 * reserve);
        for (const auto& d : descriptors) { r.descriptors.push_back(Descriptor{d.name, d.span, {}}); r.descriptors.back().data.reserve(d.span * reserve); }
        for (const auto& t : times) { r.times.push_back(TimeField{t.name, t.span, {}}); r.times.back().data.reserve(t.span * reserve); }
        return r;
    }
    // append column i of src (src must carry every descriptor of *this; missing ones are an InvalidField)
    void appendColFrom(const DataPoints& src, size_t i) {
        features.insert(features.end(), src.col(i), src.col(i) + 4);
        for (auto& d : descriptors) {
            const Descriptor& s = src.getDescriptorByName(d.name);
            if (s.span != d.span) throw InvalidField("descriptor " + d.name + " span mismatch");
            d.data.insert(d.data.end(), s.data.begin() + (size_t)s.span * i, s.data.begin() + (size_t)s.span * (i + 1));
        }
        for (auto& t : times) {
            const TimeField& s = src.getTimeByName(t.name);
            if (s.span != t.span) throw InvalidField("time " + t.name + " span mismatch");
            t.data.insert(t.data.end(), s.data.begin() + (size_t)s.span * i, s.data.begin() + (size_t)s.span * (i + 1));
        }
    }
    // PM::DataPoints::concatenate: descriptors present on both sides are kept, others dropped; an
    // empty left-hand side adopts the right-hand side
    void concatenate(const DataPoints& other) {
        if (other.getNbPoints() == 0) return;
        if (getNbPoints() == 0 && descriptors.empty()) { *this = other; return; }
        std::vector<Descriptor> kept;
        for (auto& d : descriptors) {
            const int o = other.findDescriptor(d.name);
            if (o < 0 || other.descriptors[o].span != d.span) continue;
            d.data.insert(d.data.end(), other.descriptors[o].data.begin(), other.descriptors[o].data.end());
            kept.push_back(std::move(d));
        }
        descriptors = std::move(kept);
        std::vector<TimeField> keptT;
        for (auto& t : times) {
            const int o = other.findTime(t.name);
            if (o < 0 || other.times[o].span != t.span) continue;
            t.data.insert(t.data.end(), other.times[o].data.begin(), other.times[o].data.end());
            keptT.push_back(std::move(t));
        }
        times = std::move(keptT);
        features.insert(features.end(), other.features.begin(), other.features.end());
    }
    // keep the points whose mask entry is non-zero, preserving order
    template <typename Mask>
    void keepOnly(const Mask& keep) {
        const size_t n = getNbPoints();
        size_t w = 0;
        for (size_t i = 0; i < n; ++i) {
            if (!keep[i]) continue;
            if (w != i) {
                for (int r = 0; r < 4; ++r) features[4 * w + r] = features[4 * i + r];
                for (auto& d : descriptors) for (int r = 0; r < d.span; ++r) d.data[(size_t)d.span * w + r] = d.data[(size_t)d.span * i + r];
                for (auto& t : times) for (int r = 0; r < t.span; ++r) t.data[(size_t)t.span * w + r] = t.data[(size_t)t.span * i + r];
            }
            ++w;
        }
        features.resize(4 * w);
        for (auto& d : descriptors) d.data.resize((size_t)d.span * w);
        for (auto& t : times) t.data.resize((size_t)t.span * w);
    }

    // ASCII VTK POLYDATA in libpointmatcher's dialect: POINTS / VERTICES / POINT_DATA
    // with SCALARS, VECTORS and NORMALS blocks mapped to descriptors by name; a time row group `t` travels as the two
    // unsigned_int scalars `t_splitTime_high32` / `t_splitTime_low32` (upstream's IO.cpp convention, as recalled: VTK legacy
    // has no portable 64-bit integer type) and comes back as `times`
    static DataPoints load(const std::string& path);
    void save(const std::string& path, bool binary = false) const; // VTK legacy, ASCII (default, as the reference's examples) or BINARY
};

} // namespace nim
