// IcpSequence.cpp -- see IcpSequence.h.
#include "IcpSequence.h"

#include <cstdlib>

#include <algorithm>
#include <random>
#include <cmath>
#include <limits>
#include <cstring>
#include <unordered_map>

namespace nim {

void GpuICPSequence::check(icpmi_handle h, icpmi_status s)
{
    if (s == ICPMI_OK) return;
    const std::string msg = icpmi_last_error(h);
    switch (s) {
        case ICPMI_ERR_NO_POINT_TO_MINIMIZE:
        case ICPMI_ERR_NO_OUTLIER_TO_FILTER:
        case ICPMI_ERR_BOUND:
        case ICPMI_ERR_NAN: throw ConvergenceError(msg);
        case ICPMI_ERR_MISSING_NORMALS: throw InvalidField(msg);
        case ICPMI_ERR_INVALID_ARG: throw InvalidParameter(msg);
        default: throw std::runtime_error(msg);
    }
}

GpuICPSequence::GpuICPSequence(int device)
{
    icpmi_config_default(&cfg);
    cfg.device = device;
    recreate();
}

GpuICPSequence::~GpuICPSequence() { icpmi_destroy(h); }

// the mapper's is3D == false (Mapper.h:53): every cloud is planar (z == 0); kept across loadFromYamlNode / setDefault
void GpuICPSequence::setPlanar(bool on)
{
    planar = on;
    cfg.is_2d = on ? 1 : 0;
    recreate();
}

void GpuICPSequence::recreate()
{
    // the handle is created once and re-configured afterwards: filters, modules and transformations
    // created from it keep a valid GPU context across loadFromYamlNode / setDefault
    if (h) check(h, icpmi_set_config(h, &cfg));
    else check(nullptr, icpmi_create(&cfg, &h));
}

void GpuICPSequence::setDefault()
{
    // libpointmatcher's default chain (PM::ICPChainBase::setDefault, called at Mapper.cpp:77 when the configuration has no
    // `icp:` key): RandomSampling(0.75) on the reading, SamplingSurfaceNormal on the reference,
    // KDTreeMatcher knn 1, TrimmedDist 0.85, PointToPlane, Counter 40 + Differential(1e-3, 1e-3, 3).
    const int dev = cfg.device;
    icpmi_config_default(&cfg);
    cfg.device = dev;
    cfg.is_2d = planar ? 1 : 0;
    genericDescName.clear(); genericReadDescName.clear();
    cfg.n_outlier = 1;
    cfg.outlier[0].type = ICPMI_OUT_TRIMMEDDIST;
    cfg.outlier[0].param = 0.85f;
    cfg.minimizer = ICPMI_MIN_POINT_TO_PLANE;
    cfg.use_differential = 1;
    recreate();
    readingDataPointsFilters = std::make_shared<DataPointsFilters>();
    readingDataPointsFilters->ctx = h;
    readingDataPointsFilters->filters.push_back(createDataPointsFilter("RandomSamplingDataPointsFilter", yaml::Node(), h));
    referenceDataPointsFilters = std::make_shared<DataPointsFilters>();
    referenceDataPointsFilters->ctx = h;
    referenceDataPointsFilters->filters.push_back(createDataPointsFilter("SamplingSurfaceNormalDataPointsFilter", yaml::Node(), h));
    readingStepDataPointsFilters.reset();
}

static void requireKnown(const yaml::Node& params, std::initializer_list<const char*> known, const std::string& who)
{
    if (!params.IsMap()) return;
    for (const auto& kv : params.map) {
        bool ok = false;
        for (const char* k : known) ok |= kv.first == k;
        if (!ok) throw InvalidParameter(who + ": unknown parameter " + kv.first);
    }
}

static std::pair<std::string, yaml::Node> singleEntry(const yaml::Node& n, const std::string& what)
{
    if (n.IsScalar()) return {n.scalar, yaml::Node()};
    if (n.IsMap() && n.map.size() == 1) return {n.map[0].first, n.map[0].second};
    throw InvalidParameter("malformed " + what + " entry");
}

void GpuICPSequence::loadFromYamlNode(const yaml::Node& icp)
{
    const int dev = cfg.device;
    icpmi_config_default(&cfg);
    cfg.device = dev;
    cfg.is_2d = planar ? 1 : 0;
    if (icp.IsMap())
        for (const auto& kv : icp.map) {
            static const char* valid[] = {"matcher", "outlierFilters", "errorMinimizer", "transformationCheckers", "inspector", "logger",
                                          "readingDataPointsFilters", "referenceDataPointsFilters", "readingStepDataPointsFilters"};
            bool ok = false;
            for (const char* v : valid) ok |= kv.first == v;
            if (!ok) throw InvalidParameter("unknown ICP chain key: " + kv.first);
        }

    if (icp["matcher"]) {
        auto e = singleEntry(icp["matcher"], "matcher");
        if (e.first != "KDTreeMatcher") throw InvalidParameter("unknown matcher " + e.first);
        requireKnown(e.second, {"knn", "epsilon", "searchType", "maxDist", "maxDistField"}, "KDTreeMatcher");
        if (e.second["knn"]) cfg.knn = e.second["knn"].as<int>();
        if (e.second["epsilon"]) cfg.epsilon = e.second["epsilon"].as<float>();
        // NIM_EPSILON_APPROX=1 (deployment knob): `epsilon` prunes the search as libnabo's maxError2 does (icpmi_config::
        // epsilon_approx); default: the exact search, which is a valid answer for every epsilon
        static const bool approx = [] { const char* v = std::getenv("NIM_EPSILON_APPROX"); return v && std::atoi(v) != 0; }();
        cfg.epsilon_approx = approx ? 1 : 0;
        // NIM_KNN_WG_FROM=n (deployment knob, icpmi_config::knn_wg_from): the first iteration (n - 1) of a k > 1 loop the workgroup-cooperative matcher serves;
        // default 0 = from iteration 2.  Same results either way; which is faster depends on how far the first solve moves the reading
        static const int wgFrom = [] { const char* v = std::getenv("NIM_KNN_WG_FROM"); return v ? std::atoi(v) : 0; }();
        cfg.knn_wg_from = wgFrom;
        if (e.second["maxDist"]) cfg.max_dist = e.second["maxDist"].as<float>();
    }
    cfg.n_outlier = 0;
    genericDescName.clear(); genericReadDescName.clear();
    if (icp["outlierFilters"].IsSequence())
        for (const auto& item : icp["outlierFilters"].seq) {
            auto e = singleEntry(item, "outlier filter");
            icpmi_outlier o{};
            auto param = [&](const char* key, float def) {
                requireKnown(e.second, {key}, e.first);
                return e.second[key] ? e.second[key].as<float>() : def;
            };
            if (e.first == "TrimmedDistOutlierFilter") { o.type = ICPMI_OUT_TRIMMEDDIST; o.param = param("ratio", 0.85f); }
            else if (e.first == "MaxDistOutlierFilter") { o.type = ICPMI_OUT_MAXDIST; o.param = param("maxDist", 1.f); }
            else if (e.first == "MinDistOutlierFilter") { o.type = ICPMI_OUT_MINDIST; o.param = param("minDist", 1.f); }
            else if (e.first == "MedianDistOutlierFilter") { o.type = ICPMI_OUT_MEDIANDIST; o.param = param("factor", 3.f); }
            else if (e.first == "SurfaceNormalOutlierFilter") { o.type = ICPMI_OUT_SURFACENORMAL; o.param = param("maxAngle", 1.57f); }
            else if (e.first == "VarTrimmedDistOutlierFilter") {
                // defaults of upstream's registrar: minRatio 0.05, maxRatio 0.99, lambda 0.95
                const yaml::Node& p = e.second;
                requireKnown(p, {"minRatio", "maxRatio", "lambda"}, e.first);
                o.type = ICPMI_OUT_VARTRIMMEDDIST;
                o.param = p["minRatio"] ? p["minRatio"].as<float>() : 0.05f;
                o.param2 = p["maxRatio"] ? p["maxRatio"].as<float>() : 0.99f;
                o.param3 = p["lambda"] ? p["lambda"].as<float>() : 0.95f;
            }
            else if (e.first == "GenericDescriptorOutlierFilter") {
                // defaults of upstream's registrar: source reference, descName none, useSoftThreshold 0, useLargerThan 1, threshold 0.1
                const yaml::Node& p = e.second;
                requireKnown(p, {"source", "descName", "useSoftThreshold", "useLargerThan", "threshold"}, e.first);
                const std::string source = p["source"] ? p["source"].str() : "reference";
                if (source != "reference" && source != "reading") throw InvalidParameter("GenericDescriptorOutlierFilter: source must be reference or reading");
                const std::string name = p["descName"] ? p["descName"].str() : "none";
                if (source == "reading") {
                    // (r4) the descriptor of the READING point decides: its row rides to the device ahead of every registration (operator())
                    if (!genericReadDescName.empty() && genericReadDescName != name) throw InvalidParameter("GenericDescriptorOutlierFilter: one reading descriptor per chain");
                    genericReadDescName = name;
                } else {
                    if (!genericDescName.empty() && genericDescName != name) throw InvalidParameter("GenericDescriptorOutlierFilter: the device tracks one scalar descriptor of the map");
                    genericDescName = name;
                }
                o.type = ICPMI_OUT_GENERICDESCRIPTOR;
                o.param = p["threshold"] ? p["threshold"].as<float>() : 0.1f;
                o.iparam = ((p["useSoftThreshold"] && p["useSoftThreshold"].as<int>()) ? ICPMI_GEN_SOFT : 0) |
                           ((!p["useLargerThan"] || p["useLargerThan"].as<int>()) ? ICPMI_GEN_LARGER : 0) | (source == "reading" ? ICPMI_GEN_SOURCE_READING : 0);
            } else if (e.first == "RobustOutlierFilter") {
                // defaults: robustFct cauchy, tuning 1, scaleEstimator mad, nbIterationForScale 0, distanceType point2point, approximation inf
                const yaml::Node& p = e.second;
                requireKnown(p, {"robustFct", "tuning", "scaleEstimator", "nbIterationForScale", "distanceType", "approximation"}, e.first);
                static const char* fcts[] = {"cauchy", "welsch", "sc", "gm", "tukey", "huber", "L1", "student"};
                const std::string fct = p["robustFct"] ? p["robustFct"].str() : "cauchy";
                int fid = -1;
                for (int i = 0; i < 8; ++i) if (fct == fcts[i]) fid = i;
                if (fid < 0) throw InvalidParameter("RobustOutlierFilter: unknown robustFct " + fct);
                const std::string sc = p["scaleEstimator"] ? p["scaleEstimator"].str() : "mad";
                int sid;
                if (sc == "none") sid = ICPMI_SCALE_NONE; else if (sc == "mad") sid = ICPMI_SCALE_MAD;
                else if (sc == "berg") sid = ICPMI_SCALE_BERG; else if (sc == "std") sid = ICPMI_SCALE_STD;
                else throw InvalidParameter("RobustOutlierFilter: unknown scaleEstimator " + sc);
                const std::string dt = p["distanceType"] ? p["distanceType"].str() : "point2point";
                int did;
                if (dt == "point2point") did = ICPMI_DIST_POINT2POINT; else if (dt == "point2plane") did = ICPMI_DIST_POINT2PLANE;
                else throw InvalidParameter("RobustOutlierFilter: unknown distanceType " + dt);
                const float apx = p["approximation"] ? p["approximation"].as<float>() : std::numeric_limits<float>::infinity();
                if (!(apx > 0.f)) throw InvalidParameter("RobustOutlierFilter: approximation must be > 0");   // (upstream's range: min 0, max inf; 0 would drop every match)
                o.type = ICPMI_OUT_ROBUST;
                o.param = p["tuning"] ? p["tuning"].as<float>() : 1.f;
                o.param2 = p["nbIterationForScale"] ? (float)p["nbIterationForScale"].as<int>() : 0.f;
                o.param3 = apx;
                o.iparam = fid | (sid << 4) | (did << 8);
            }
            else throw InvalidParameter("unknown outlier filter " + e.first);
            if (cfg.n_outlier >= 8) throw InvalidParameter("at most 8 outlier filters");
            cfg.outlier[cfg.n_outlier++] = o;
        }
    if (icp["errorMinimizer"]) {
        auto e = singleEntry(icp["errorMinimizer"], "errorMinimizer");
        if (e.first == "IdentityErrorMinimizer") cfg.minimizer = ICPMI_MIN_IDENTITY;
        else if (e.first == "PointToPointErrorMinimizer") cfg.minimizer = ICPMI_MIN_POINT_TO_POINT;
        else if (e.first == "PointToPlaneErrorMinimizer") {
            cfg.minimizer = ICPMI_MIN_POINT_TO_PLANE;
            cfg.force_4dof = (e.second["force4DOF"] && e.second["force4DOF"].as<int>() != 0) ? 1 : 0;
            cfg.force_2d = (e.second["force2D"] && e.second["force2D"].as<int>() != 0) ? 1 : 0;
            if (cfg.force_4dof && cfg.force_2d) throw InvalidParameter("PointToPlaneErrorMinimizer: force2D and force4DOF exclude each other");
        } else throw InvalidParameter("unknown error minimizer " + e.first);
    }
    if (icp["transformationCheckers"].IsSequence())
        for (const auto& item : icp["transformationCheckers"].seq) {
            auto e = singleEntry(item, "transformation checker");
            if (e.first == "CounterTransformationChecker") {
                requireKnown(e.second, {"maxIterationCount"}, e.first);
                if (e.second["maxIterationCount"]) cfg.max_iterations = e.second["maxIterationCount"].as<int>();
            } else if (e.first == "DifferentialTransformationChecker") {
                requireKnown(e.second, {"minDiffRotErr", "minDiffTransErr", "smoothLength"}, e.first);
                cfg.use_differential = 1;
                if (e.second["minDiffRotErr"]) cfg.min_diff_rot = e.second["minDiffRotErr"].as<float>();
                if (e.second["minDiffTransErr"]) cfg.min_diff_trans = e.second["minDiffTransErr"].as<float>();
                if (e.second["smoothLength"]) cfg.smooth_length = e.second["smoothLength"].as<int>();
            } else if (e.first == "BoundTransformationChecker") {
                requireKnown(e.second, {"maxRotationNorm", "maxTranslationNorm"}, e.first);
                cfg.use_bound = 1;
                if (e.second["maxRotationNorm"]) cfg.max_rot_norm = e.second["maxRotationNorm"].as<float>();
                if (e.second["maxTranslationNorm"]) cfg.max_trans_norm = e.second["maxTranslationNorm"].as<float>();
            } else throw InvalidParameter("unknown transformation checker " + e.first);
        }
    recreate();
    auto chain = [&](const char* key) -> std::shared_ptr<DataPointsFilters> {
        if (!icp[key] || !icp[key].IsSequence() || icp[key].seq.empty()) return nullptr;
        return std::make_shared<DataPointsFilters>(icp[key], h);
    };
    readingDataPointsFilters = chain("readingDataPointsFilters");
    referenceDataPointsFilters = chain("referenceDataPointsFilters");
    readingStepDataPointsFilters = chain("readingStepDataPointsFilters");
    if (readingStepDataPointsFilters)
        for (const auto& f : readingStepDataPointsFilters->filters)
            if (!f->repeatable())
                throw InvalidParameter("readingStepDataPointsFilters: a filter whose result changes from call to call (RandomSampling with seed -1) "
                                       "cannot run inside the device-resident loop; give it a seed, or apply it as a readingDataPointsFilter");
}

bool GpuICPSequence::hasMap() const { return h && icpmi_has_map(h); }

bool GpuICPSequence::hasReadingFilters() const
{
    return (readingDataPointsFilters && readingDataPointsFilters->size()) || (readingStepDataPointsFilters && readingStepDataPointsFilters->size());
}

DataPoints GpuICPSequence::filteredReading(const DataPoints& reading) const
{
    DataPoints r = reading;
    if (readingDataPointsFilters) readingDataPointsFilters->apply(r);
    if (readingStepDataPointsFilters) readingStepDataPointsFilters->apply(r);
    return r;
}

bool GpuICPSequence::setMap(const DataPoints& mapIn)
{
    int32_t accepted = 0;
    if (mapIn.getNbPoints() == 0) return false; // "Ignoring attempt to setMap with an empty map"
    DataPoints filtered;
    const DataPoints* mapp = &mapIn;
    if (referenceDataPointsFilters && referenceDataPointsFilters->size()) {
        // upstream filters the CENTRED copy of the map (mean subtracted first): filters that look at coordinates
        // (BoundingBox, DistanceLimit) mean them relative to the centroid.  The original coordinates ride along as a descriptor
        // and are put back afterwards, so that the core centres the very points the caller handed in.
        filtered = mapIn;
        const size_t n = filtered.getNbPoints();
        double mean[3] = {0, 0, 0};
        for (size_t i = 0; i < n; ++i) for (int r = 0; r < 3; ++r) mean[r] += filtered.col(i)[r];
        std::vector<float> orig(3 * n);
        for (size_t i = 0; i < n; ++i)
            for (int r = 0; r < 3; ++r) { orig[3 * i + r] = filtered.col(i)[r]; filtered.col(i)[r] = (float)(filtered.col(i)[r] - mean[r] / (double)n); }
        filtered.addDescriptor("__icpmi_original_xyz", 3, std::move(orig));
        referenceDataPointsFilters->apply(filtered);
        const size_t m = filtered.getNbPoints();
        if (filtered.descriptorExists("__icpmi_original_xyz")) {
            const Descriptor& o = filtered.getDescriptorByName("__icpmi_original_xyz");
            for (size_t i = 0; i < m; ++i) for (int r = 0; r < 3; ++r) filtered.col(i)[r] = o.data[3 * i + r];
            filtered.removeDescriptor("__icpmi_original_xyz");
        } else
            for (size_t i = 0; i < m; ++i) for (int r = 0; r < 3; ++r) filtered.col(i)[r] = (float)(filtered.col(i)[r] + mean[r] / (double)n);
        if (m == 0) return false;
        mapp = &filtered;
    }
    const DataPoints& map = *mapp;
    const float* normals = nullptr;
    if (map.descriptorExists("normals")) {
        const Descriptor& d = map.getDescriptorByName("normals");
        if (d.span != 3) throw InvalidField("descriptor normals must have 3 rows");
        normals = d.data.data();
    }
    // a point-to-plane chain against a map without `normals`: the map is accepted (upstream's setMap does not look) and the
    // registration raises InvalidField("normals"), as upstream's minimiser does
    check(h, icpmi_set_map(h, map.features.data(), (int64_t)map.getNbPoints(), normals, &accepted));
    if (accepted && !genericDescName.empty()) {
        // GenericDescriptorOutlierFilter{source: reference}: the descriptor it reads travels as the map's tracked scalar channel
        if (!map.descriptorExists(genericDescName) || map.getDescriptorByName(genericDescName).span != 1)
            throw InvalidField("GenericDescriptorOutlierFilter: the reference has no 1-row descriptor " + genericDescName);
        uploadMapScalar(map.getDescriptorByName(genericDescName).data);
    }
    return accepted != 0;
}

void GpuICPSequence::mapUpdatePointDistance(const DataPoints& input, float minDist, int normalsKnn, std::vector<uint8_t>& keep, int64_t& appended,
                                            int64_t& mapSize)
{
    const float* normals = nullptr;
    if (normalsKnn <= 0 && input.descriptorExists("normals") && input.getDescriptorByName("normals").span == 3)
        normals = input.getDescriptorByName("normals").data.data();
    keep.assign(input.getNbPoints(), 0);
    check(h, icpmi_map_update_point_distance(h, input.features.data(), (int64_t)input.getNbPoints(), normals, minDist, normalsKnn, keep.data(),
                                             &appended, &mapSize));
}

Mat4 GpuICPSequence::registerWithPrior(const DataPoints& scan, const Mat4& prior)
{
    Mat4 T = Mat4::identity();
    stagedPoints = scan.getNbPoints();
    check(h, icpmi_register_prior(h, scan.features.data(), (int64_t)scan.getNbPoints(), prior.data(), T.data(), &lastStats));
    return T;
}

void GpuICPSequence::mapUpdateStaged(const Mat4& correction, float minDist, int normalsKnn, std::vector<uint8_t>& keep, int64_t& appended,
                                     int64_t& mapSize)
{
    keep.assign(stagedPoints, 0);
    check(h, icpmi_map_update_staged(h, correction.data(), minDist, normalsKnn, keep.data(), &appended, &mapSize));
}

static std::vector<float> scalarRow(const DataPoints& cloud, const std::string& name)
{
    const Descriptor& d = cloud.getDescriptorByName(name);
    const size_t n = cloud.getNbPoints();
    std::vector<float> out(n);
    for (size_t i = 0; i < n; ++i) out[i] = d.data[(size_t)d.span * i];
    return out;
}

void GpuICPSequence::mapUpdateChain(const DataPoints* input, const Mat4& correction, const std::string& scalarName, const DataPoints& scanDescriptors,
                                    const Mat4& pose, const std::vector<icpmi_map_op>& ops, int nModules, std::vector<int32_t>& src,
                                    int64_t& prefix, int64_t& mapSize, bool wantSrc)
{
    const Mat4 toSensor = pose.inverse();
    const size_t n = input ? input->getNbPoints() : stagedPoints;
    std::vector<float> scalar;
    if (!scalarName.empty()) scalar = scalarRow(scanDescriptors, scalarName);
    const float* sp = scalarName.empty() ? nullptr : scalar.data();
    src.resize(wantSrc ? (size_t)residentMapSize() + (size_t)(nModules > 0 ? nModules : 1) * n + 1 : 0);
    int32_t* srcp = wantSrc ? src.data() : nullptr;
    int64_t* prefp = wantSrc ? &prefix : nullptr;
    prefix = 0;
    if (input) {
        const float* normals = nullptr;
        if (input->descriptorExists("normals") && input->getDescriptorByName("normals").span == 3) normals = input->getDescriptorByName("normals").data.data();
        check(h, icpmi_map_update_chain(h, input->features.data(), (int64_t)n, normals, sp, toSensor.data(), pose.data(), ops.data(), (int32_t)ops.size(), nModules,
                                        srcp, (int64_t)src.size(), prefp, &mapSize));
    } else
        check(h, icpmi_map_update_chain_staged(h, correction.data(), sp, toSensor.data(), pose.data(), ops.data(), (int32_t)ops.size(), nModules, srcp,
                                               (int64_t)src.size(), prefp, &mapSize));
    if (wantSrc) src.resize((size_t)mapSize);
}

void GpuICPSequence::uploadMapScalar(const std::vector<float>& scalar) { check(h, icpmi_set_map_scalar(h, scalar.data(), (int64_t)scalar.size())); }

std::vector<float> GpuICPSequence::downloadMapScalar() const
{
    std::vector<float> out((size_t)residentMapSize());
    if (!out.empty()) check(h, icpmi_get_map_scalar(h, out.data(), (int64_t)out.size()));
    return out;
}

int64_t GpuICPSequence::residentMapSize() const
{
    int64_t m = 0;
    check(h, icpmi_get_map(h, nullptr, nullptr, 0, &m));
    return m;
}

bool GpuICPSequence::chainNeedsReadingNormals() const
{
    for (int f = 0; f < cfg.n_outlier; ++f)
        if (cfg.outlier[f].type == ICPMI_OUT_SURFACENORMAL) return true;
    return false;
}

DataPoints GpuICPSequence::downloadMap() const
{
    int64_t m = 0;
    check(h, icpmi_get_map(h, nullptr, nullptr, 0, &m));
    DataPoints out((size_t)m);
    if (m == 0) return out;
    std::vector<float> nrm(3 * (size_t)m);
    const icpmi_status s = icpmi_get_map(h, out.features.data(), nrm.data(), m, &m);
    if (s == ICPMI_ERR_MISSING_NORMALS) check(h, icpmi_get_map(h, out.features.data(), nullptr, m, &m));
    else { check(h, s); out.addDescriptor("normals", 3, std::move(nrm)); }
    return out;
}

Mat4 GpuICPSequence::operator()(const DataPoints& readingIn)
{
    Mat4 T = Mat4::identity();
    DataPoints owned;
    if (hasReadingFilters()) owned = filteredReading(readingIn);
    const DataPoints& reading = hasReadingFilters() ? owned : readingIn;
    const float* normals = nullptr;
    if (reading.descriptorExists("normals") && reading.getDescriptorByName("normals").span == 3)
        normals = reading.getDescriptorByName("normals").data.data();
    // ErrorMinimizer::getOverlap() (Mapper.cpp:219): a reading that carries `simpleSensorNoise` (and normals) gets upstream's
    // sensor-noise count instead of the weighted ratio; the row rides to the device ahead of the registration (one shot)
    if (!genericReadDescName.empty()) { // GenericDescriptorOutlierFilter{source: reading}
        if (!reading.descriptorExists(genericReadDescName) || reading.getDescriptorByName(genericReadDescName).span != 1)
            throw InvalidField("GenericDescriptorOutlierFilter: the reading has no 1-row descriptor " + genericReadDescName);
        check(h, icpmi_set_reading_scalar(h, reading.getDescriptorByName(genericReadDescName).data.data(), (int64_t)reading.getNbPoints()));
    }
    // (PointToPointErrorMinimizer::getOverlap() needs the noise row alone, only PointToPlane also the reading's normals)
    if ((normals || cfg.minimizer == ICPMI_MIN_POINT_TO_POINT) && reading.descriptorExists("simpleSensorNoise") &&
        reading.getDescriptorByName("simpleSensorNoise").span == 1)
        check(h, icpmi_set_reading_sensor_noise(h, reading.getDescriptorByName("simpleSensorNoise").data.data(), (int64_t)reading.getNbPoints()));
    check(h, icpmi_register(h, reading.features.data(), (int64_t)reading.getNbPoints(), normals, T.data(), &lastStats));
    return T;
}

// ------------------------------------------------------------------------------------------------
DataPoints RigidTransformation::compute(const DataPoints& cloud, const Mat4& T) const
{
    DataPoints out = cloud;
    const int64_t n = (int64_t)cloud.getNbPoints();
    const int dn = cloud.findDescriptor("normals");
    const float* nin = dn >= 0 && cloud.descriptors[dn].span == 3 ? cloud.descriptors[dn].data.data() : nullptr;
    float* nout = nin ? out.descriptors[dn].data.data() : nullptr;
    icpmi_status s = icpmi_transform(h, T.data(), cloud.features.data(), n, out.features.data(), nin, nout);
    if (s == ICPMI_ERR_INVALID_ARG) throw TransformationError(icpmi_last_error(h));
    GpuICPSequence::check(h, s);
    const int dobs = cloud.findDescriptor("observationDirections");
    if (dobs >= 0 && cloud.descriptors[dobs].span == 3 && n > 0) {
        // rotate the second direction field with the same operator (features are recomputed, cheap)
        std::vector<float> scratch(cloud.features.size());
        GpuICPSequence::check(h, icpmi_transform(h, T.data(), cloud.features.data(), n, scratch.data(),
                                                 cloud.descriptors[dobs].data.data(), out.descriptors[dobs].data.data()));
    }
    return out;
}

// ------------------------------------------------------------------------------------------------
// DataPointsFilters (semantics: SURVEY.md B.9)
// ------------------------------------------------------------------------------------------------
namespace {

struct DistanceLimitFilter : DataPointsFilter {
    int dim = -1; float dist = 1.f; bool removeInside = true;
    bool pointFilter(icpmi_point_filter& f) const override {
        f = icpmi_point_filter{}; f.type = ICPMI_FILT_DISTANCE_LIMIT; f.i = dim; f.f[0] = dist; f.f[1] = removeInside ? 1.f : 0.f;
        return dim >= -1 && dim <= 2;
    }
    void inPlaceFilter(DataPoints& c) const override {
        const size_t n = c.getNbPoints();
        std::vector<uint8_t> keep(n);
        const float ad = std::fabs(dist);
        for (size_t i = 0; i < n; ++i) {
            const float* p = c.col(i);
            const float v = dim < 0 ? std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) : std::fabs(p[dim]);
            keep[i] = removeInside ? v > ad : v < ad;
        }
        c.keepOnly(keep);
    }
};

struct BoundingBoxFilter : DataPointsFilter {
    float lo[3] = {-1, -1, -1}, hi[3] = {1, 1, 1}; bool removeInside = true;
    bool pointFilter(icpmi_point_filter& f) const override {
        f = icpmi_point_filter{}; f.type = ICPMI_FILT_BOUNDING_BOX; f.i = removeInside ? 1 : 0;
        for (int r = 0; r < 3; ++r) { f.f[r] = lo[r]; f.f[3 + r] = hi[r]; }
        return true;
    }
    void inPlaceFilter(DataPoints& c) const override {
        const size_t n = c.getNbPoints();
        std::vector<uint8_t> keep(n);
        for (size_t i = 0; i < n; ++i) {
            const float* p = c.col(i);
            bool inside = true;
            for (int r = 0; r < 3; ++r) inside &= p[r] > lo[r] && p[r] < hi[r];
            keep[i] = removeInside ? !inside : inside;
        }
        c.keepOnly(keep);
    }
};

struct AddDescriptorFilter : DataPointsFilter {
    std::string name; int dimension = 1; std::vector<float> values;
    void inPlaceFilter(DataPoints& c) const override {
        const size_t n = c.getNbPoints();
        std::vector<float> data((size_t)dimension * n);
        for (size_t i = 0; i < n; ++i) for (int r = 0; r < dimension; ++r) data[(size_t)dimension * i + r] = values[r];
        c.addDescriptor(name, dimension, std::move(data));
    }
};

struct CutAtDescriptorThresholdFilter : DataPointsFilter {
    std::string name; bool useLargerThan = true; float threshold = 0.f;
    bool residentOp(icpmi_map_op& op, std::string& scalarName) const override {
        op = icpmi_map_op{}; op.type = ICPMI_MOP_CUT_SCALAR; op.i = useLargerThan ? 1 : 0; op.f[0] = threshold;
        scalarName = name;
        return true;
    }
    void inPlaceFilter(DataPoints& c) const override {
        const Descriptor& d = c.getDescriptorByName(name);
        const size_t n = c.getNbPoints();
        std::vector<uint8_t> keep(n);
        for (size_t i = 0; i < n; ++i) {
            const float v = d.data[(size_t)d.span * i];
            keep[i] = useLargerThan ? !(v > threshold) : !(v < threshold);
        }
        c.keepOnly(keep);
    }
};

struct SurfaceNormalFilter : DataPointsFilter {
    icpmi_handle h; int knn = 5; bool keepDensities = false, keepMatchedIds = false, keepMeanDist = false, keepEigenValues = false, keepEigenVectors = false;
    int surfaceNormalKnn() const override { return knn; }
    bool residentOp(icpmi_map_op& op, std::string&) const override {
        op = icpmi_map_op{}; op.type = ICPMI_MOP_SURFACE_NORMALS; op.i = knn;
        // the resident map does not track `densities`, `matchedIds` or `meanDist`
        return knn >= 1 && knn <= 32 && !keepDensities && !keepMatchedIds && !keepMeanDist && !keepEigenValues && !keepEigenVectors;
    }
    void inPlaceFilter(DataPoints& c) const override {
        const size_t n = c.getNbPoints();
        std::vector<float> normals(3 * n), dens(keepDensities ? n : 0), md(keepMeanDist ? n : 0);
        std::vector<int32_t> ids(keepMatchedIds ? n * (size_t)knn : 0);
        std::vector<float> eva(keepEigenValues ? 3 * n : 0), eve(keepEigenVectors ? 9 * n : 0);
        GpuICPSequence::check(h, icpmi_surface_normals_ex3(h, c.features.data(), (int64_t)n, knn, normals.data(), keepDensities ? dens.data() : nullptr,
                                                           keepMatchedIds ? ids.data() : nullptr, keepMeanDist ? md.data() : nullptr,
                                                           keepEigenValues ? eva.data() : nullptr, keepEigenVectors ? eve.data() : nullptr));
        c.addDescriptor("normals", 3, std::move(normals));
        if (keepEigenValues) c.addDescriptor("eigValues", 3, std::move(eva));     // (upstream's descriptor names)
        if (keepEigenVectors) c.addDescriptor("eigVectors", 9, std::move(eve));
        if (keepDensities) c.addDescriptor("densities", 1, std::move(dens));
        if (keepMatchedIds) { // upstream stores the ids as descriptor rows of the cloud's scalar type
            std::vector<float> f(ids.size());
            for (size_t i = 0; i < ids.size(); ++i) f[i] = (float)ids[i];
            c.addDescriptor("matchedIds", knn, std::move(f));
        }
        if (keepMeanDist) c.addDescriptor("meanDist", 1, std::move(md));
    }
};

// std::minstd_rand (x <- 48271 x mod 2^31 - 1: fully specified by the C++ standard, so the oracle restates it) and the two ways
// upstream turns it into [0, 1): randomSamplingMethod 0 "direct" = x / float(max - min), 1 "uniform" =
// std::uniform_real_distribution<float> = (x - min) / float(max - min + 1) capped below 1 (libstdc++'s generate_canonical).
struct MinStd {
    uint32_t x;
    explicit MinStd(uint32_t seed) : x(seed % 2147483647u) { if (x == 0) x = 1; }
    uint32_t next() { x = (uint32_t)(((uint64_t)x * 48271ull) % 2147483647ull); return x; }
    float unit(int method) {
        const uint32_t v = next();
        if (method == 1) { const float r = (float)(v - 1u) / 2147483646.0f; return r < 1.0f ? r : std::nextafter(1.0f, 0.0f); }
        return (float)v / 2147483645.0f; // max() - min() = 2147483646 - 1
    }
};

} // namespace
// test seam (TestHooks.cpp: nim_test_minstd_nth): the raw n-th value of the host filters' generator
uint32_t minstdNth(uint32_t seed, uint32_t n) { MinStd g(seed); uint32_t v = g.x; for (uint32_t i = 0; i < n; ++i) v = g.next(); return v; }
namespace {

// RandomSamplingDataPointsFilter{prob 0.75, randomSamplingMethod 0, seed -1} [UPSTREAM 1.4.x, as recalled]: a fresh
// std::minstd_rand per call (seed -1: std::random_device), one number per point, point kept iff number < prob, never more
// than floor(n * prob) + 1 points.
struct RandomSamplingFilter : DataPointsFilter {
    float prob = 0.75f; int method = 0; int seed = -1;
    bool repeatable() const override { return seed != -1; }
    void inPlaceFilter(DataPoints& c) const override {
        const size_t n = c.getNbPoints();
        const size_t nOut = (size_t)((float)n * prob);
        MinStd rng(seed == -1 ? (uint32_t)std::random_device()() : (uint32_t)seed);
        std::vector<uint8_t> keep(n, 0);
        size_t j = 0;
        for (size_t i = 0; i < n && j <= nOut; ++i)
            if (rng.unit(method) < prob) { keep[i] = 1; ++j; }
        c.keepOnly(keep);
    }
};

// MaxDensityDataPointsFilter{maxDensity 10} [UPSTREAM]: needs `densities` (SurfaceNormalDataPointsFilter{keepDensities: 1});
// a point in a region denser than maxDensity survives with probability maxDensity / density.
struct MaxDensityFilter : DataPointsFilter {
    float maxDensity = 10.f; int seed = 1;
    void inPlaceFilter(DataPoints& c) const override {
        if (!c.descriptorExists("densities")) throw InvalidField("MaxDensityDataPointsFilter: Error, no densities found in descriptors.");
        const Descriptor& d = c.getDescriptorByName("densities");
        const size_t n = c.getNbPoints();
        MinStd rng((uint32_t)seed);
        std::vector<uint8_t> keep(n, 1);
        for (size_t i = 0; i < n; ++i) {
            const float density = d.data[(size_t)d.span * i];
            if (density > maxDensity) keep[i] = rng.unit(0) < maxDensity / density;
        }
        c.keepOnly(keep);
    }
};

struct IdentityFilter : DataPointsFilter { void inPlaceFilter(DataPoints&) const override {} };

// ObservationDirectionDataPointsFilter{x 0, y 0, z 0} [UPSTREAM]: descriptor `observationDirections` = sensor position - point
// (3 rows; it rotates with the cloud like `normals`, RigidTransformation::compute)
struct ObservationDirectionFilter : DataPointsFilter {
    float c[3] = {0, 0, 0};
    void inPlaceFilter(DataPoints& cl) const override {
        const size_t n = cl.getNbPoints();
        std::vector<float> d(3 * n);
        for (size_t i = 0; i < n; ++i) { const float* p = cl.col(i); for (int r = 0; r < 3; ++r) d[3 * i + r] = c[r] - p[r]; }
        if (cl.descriptorExists("observationDirections")) cl.removeDescriptor("observationDirections");
        cl.addDescriptor("observationDirections", 3, std::move(d));
    }
};

// OrientNormalsDataPointsFilter{towardCenter 1} [UPSTREAM]: a normal whose scalar product with the observation direction is
// negative (towardCenter) / positive (away) is flipped; needs `normals` and `observationDirections`
struct OrientNormalsFilter : DataPointsFilter {
    bool towardCenter = true;
    void inPlaceFilter(DataPoints& cl) const override {
        if (!cl.descriptorExists("normals")) throw InvalidField("OrientNormalsDataPointsFilter: Error, cannot find normals in descriptors.");
        if (!cl.descriptorExists("observationDirections")) throw InvalidField("OrientNormalsDataPointsFilter: Error, cannot find observation directions in descriptors.");
        Descriptor nrm = cl.getDescriptorByName("normals");
        const Descriptor& od = cl.getDescriptorByName("observationDirections");
        if (nrm.span != 3 || od.span != 3) throw InvalidField("OrientNormalsDataPointsFilter: normals and observationDirections must have 3 rows");
        const size_t n = cl.getNbPoints();
        for (size_t i = 0; i < n; ++i) {
            const float dot = nrm.data[3 * i] * od.data[3 * i] + nrm.data[3 * i + 1] * od.data[3 * i + 1] + nrm.data[3 * i + 2] * od.data[3 * i + 2];
            if (towardCenter ? dot < 0.f : dot > 0.f) for (int r = 0; r < 3; ++r) nrm.data[3 * i + r] = -nrm.data[3 * i + r];
        }
        cl.removeDescriptor("normals");
        cl.addDescriptor("normals", 3, std::move(nrm.data));
    }
};

struct RemoveNaNFilter : DataPointsFilter {
    void inPlaceFilter(DataPoints& c) const override {
        const size_t n = c.getNbPoints();
        std::vector<uint8_t> keep(n);
        for (size_t i = 0; i < n; ++i) { const float* p = c.col(i); keep[i] = !(std::isnan(p[0]) || std::isnan(p[1]) || std::isnan(p[2])); }
        c.keepOnly(keep);
    }
};

// symmetric 3x3 eigen-decomposition (cyclic Jacobi, double): eigenvalues w, eigenvectors in the columns of Q
static void jacobi3(const double C[9], double w[3], double Q[9])
{
    double A[3][3] = {{C[0], C[3], C[6]}, {C[1], C[4], C[7]}, {C[2], C[5], C[8]}};
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 60; ++sweep) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        const double dg = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
        if (off <= 1e-32 * dg || off < 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (A[p][q] == 0.0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < 3; ++k) { const double a = A[k][p], b = A[k][q]; A[k][p] = cs * a - sn * b; A[k][q] = sn * a + cs * b; }
                for (int k = 0; k < 3; ++k) { const double a = A[p][k], b = A[q][k]; A[p][k] = cs * a - sn * b; A[q][k] = sn * a + cs * b; }
                for (int k = 0; k < 3; ++k) { const double a = V[k][p], b = V[k][q]; V[k][p] = cs * a - sn * b; V[k][q] = sn * a + cs * b; }
            }
    }
    for (int e = 0; e < 3; ++e) { w[e] = A[e][e]; for (int r = 0; r < 3; ++r) Q[3 * e + r] = V[r][e]; }
}

// SamplingSurfaceNormalDataPointsFilter{ratio 0.5, knn 7, samplingMethod 0, maxBoxDim inf, averageExistingDescriptors 1,
// keepNormals 1} [UPSTREAM, as recalled] -- the reference filter of PM::ICPSequence::setDefault (Mapper.cpp:77): the cloud is
// split at the median of its widest dimension until a box holds at most knn points; every box gets ONE normal (smallest
// eigenvector of the covariance of its points; boxes of rank < 2 or wider than maxBoxDim are dropped), and its points are kept
// with probability `ratio` (samplingMethod 0) or replaced by their mean (1).  Output in box order.  Deterministic here: the
// median split orders by (coordinate, index), points inside a box by index, the random numbers are MinStd(seed) (upstream:
// std::nth_element's permutation and std::rand).  Host code: a recursive median split is what the reference runs on the CPU
// too; it runs once per setMap.
struct SamplingSurfaceNormalFilter : DataPointsFilter {
    icpmi_handle h = nullptr; // GPU context (createDataPointsFilter): the device path of inPlaceFilter
    float ratio = 0.5f; int knn = 7; int method = 0; float maxBoxDim = INFINITY; bool averageDescriptors = true; bool keepNormals = true; int seed = 1;
    struct Work {
        const DataPoints* in; DataPoints out; std::vector<float> normals; MinStd rng; const SamplingSurfaceNormalFilter* f;
        Work(const DataPoints* c, const SamplingSurfaceNormalFilter* ff) : in(c), out(c->createSimilarEmpty()), rng((uint32_t)ff->seed), f(ff) {}
    };
    void fuse(Work& w, std::vector<int32_t>& idx, size_t first, size_t last) const {
        const size_t cnt = last - first;
        std::sort(idx.begin() + first, idx.begin() + last);
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        double mean[3] = {0, 0, 0};
        for (size_t k = first; k < last; ++k) {
            const float* p = w.in->col((size_t)idx[k]);
            for (int r = 0; r < 3; ++r) { lo[r] = std::min(lo[r], p[r]); hi[r] = std::max(hi[r], p[r]); mean[r] += p[r]; }
        }
        if (std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2])) > maxBoxDim) return;
        for (int r = 0; r < 3; ++r) mean[r] /= (double)cnt;
        double C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (size_t k = first; k < last; ++k) {
            const float* p = w.in->col((size_t)idx[k]);
            const double v[3] = {p[0] - mean[0], p[1] - mean[1], p[2] - mean[2]};
            for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) C[3 * c + r] += v[r] * v[c];
        }
        double ev[3], Q[9];
        jacobi3(C, ev, Q);
        const double wmax = std::max(std::fabs(ev[0]), std::max(std::fabs(ev[1]), std::fabs(ev[2])));
        int rank = 0;
        for (int e = 0; e < 3; ++e) if (wmax > 0 && std::fabs(ev[e]) > 3.0 * 1.1920928955078125e-07 * wmax) ++rank;
        if (rank < 2) return; // the points of the box are "unfit"
        int e = 0;
        if (ev[1] < ev[e]) e = 1;
        if (ev[2] < ev[e]) e = 2;
        const float nrm[3] = {(float)Q[3 * e], (float)Q[3 * e + 1], (float)Q[3 * e + 2]};
        if (method == 1) {
            w.out.appendColFrom(*w.in, (size_t)idx[first]);
            const size_t j = w.out.getNbPoints() - 1;
            for (int r = 0; r < 3; ++r) w.out.col(j)[r] = (float)mean[r];
            if (averageDescriptors)
                for (size_t d = 0; d < w.in->descriptors.size(); ++d) {
                    const Descriptor& src = w.in->descriptors[d];
                    for (int r = 0; r < src.span; ++r) {
                        double s2 = 0;
                        for (size_t k = first; k < last; ++k) s2 += src.data[(size_t)src.span * idx[k] + r];
                        w.out.descriptors[d].data[(size_t)src.span * j + r] = (float)(s2 / (double)cnt);
                    }
                }
            w.normals.insert(w.normals.end(), nrm, nrm + 3);
            return;
        }
        for (size_t k = first; k < last; ++k)
            if (w.rng.unit(0) < ratio) { w.out.appendColFrom(*w.in, (size_t)idx[k]); w.normals.insert(w.normals.end(), nrm, nrm + 3); }
    }
    void build(Work& w, std::vector<int32_t>& idx, size_t first, size_t last, const float lo[3], const float hi[3]) const {
        const size_t cnt = last - first;
        if (cnt == 0) return;
        if (cnt <= (size_t)knn) { fuse(w, idx, first, last); return; }
        int dim = 0;
        for (int r = 1; r < 3; ++r) if (hi[r] - lo[r] > hi[dim] - lo[dim]) dim = r;
        const size_t right = cnt / 2, left = cnt - right;
        const DataPoints* in = w.in;
        auto less = [in, dim](int32_t a, int32_t b) { const float x = in->col((size_t)a)[dim], y = in->col((size_t)b)[dim]; return x < y || (x == y && a < b); };
        std::nth_element(idx.begin() + first, idx.begin() + first + left, idx.begin() + last, less);
        const float cut = in->col((size_t)idx[first + left])[dim];
        float lhi[3] = {hi[0], hi[1], hi[2]}, rlo[3] = {lo[0], lo[1], lo[2]};
        lhi[dim] = cut; rlo[dim] = cut;
        build(w, idx, first, first + left, lo, lhi);
        build(w, idx, first + left, last, rlo, hi);
    }
    void inPlaceFilter(DataPoints& c) const override {
        const size_t n = c.getNbPoints();
        if (n == 0) return;
        if (h && method == 0 && knn >= 3 && seed >= 0) {
            // r3: the partition, the box normals and the sampling run on the device (csrc/ssn.hip: one radix sort per tree level) -- the
            // filter sits on the REFERENCE of the default chain, i.e. on the whole map at every icp.setMap; the host recursion below
            // (one thread) is what a shell WITHOUT a GPU context runs (the CPU-only unit tests of the host classes)
            std::vector<int32_t> order(n);
            std::vector<float> nrm(3 * n);
            int64_t kept = 0;
            GpuICPSequence::check(h, icpmi_sampling_surface_normal(h, c.features.data(), (int64_t)n, ratio, knn, maxBoxDim, seed, order.data(), nrm.data(), &kept));
            DataPoints out = c.createSimilarEmpty();
            for (int64_t k = 0; k < kept; ++k) out.appendColFrom(c, (size_t)order[(size_t)k]);
            nrm.resize(3 * (size_t)kept);
            if (keepNormals) out.addDescriptor("normals", 3, std::move(nrm));
            c = std::move(out);
            return;
        }
        if (h && method == 1 && knn >= 3) {
            // r5: samplingMethod 1 on the device too -- partition, box normals and box means by csrc/ssn.hip; what stays here is the
            // bookkeeping of the container: the kept column per box and, with averageExistingDescriptors, the mean of every descriptor
            // row over the members the device lists (descriptor rows live in this container, not on the device)
            std::vector<int32_t> order(n), ms(n), mc(n), mem(n);
            std::vector<float> nrm(3 * n), mean(3 * n);
            int64_t boxes = 0;
            GpuICPSequence::check(h, icpmi_sampling_surface_normal_ex(h, c.features.data(), (int64_t)n, 1.0f, knn, maxBoxDim, seed < 0 ? 1 : seed, 1, order.data(),
                                                                      nrm.data(), &boxes, mean.data(), ms.data(), mc.data(), mem.data()));
            DataPoints out = c.createSimilarEmpty();
            for (int64_t b = 0; b < boxes; ++b) {
                out.appendColFrom(c, (size_t)order[(size_t)b]);
                const size_t j = out.getNbPoints() - 1;
                for (int r = 0; r < 3; ++r) out.col(j)[r] = mean[3 * (size_t)b + r];
                if (averageDescriptors)
                    for (size_t d = 0; d < c.descriptors.size(); ++d) {
                        const Descriptor& src = c.descriptors[d];
                        for (int r = 0; r < src.span; ++r) {
                            double s2 = 0;
                            for (int32_t k = 0; k < mc[(size_t)b]; ++k) s2 += src.data[(size_t)src.span * mem[(size_t)ms[(size_t)b] + k] + r];
                            out.descriptors[d].data[(size_t)src.span * j + r] = (float)(s2 / (double)mc[(size_t)b]);
                        }
                    }
            }
            nrm.resize(3 * (size_t)boxes);
            if (keepNormals) out.addDescriptor("normals", 3, std::move(nrm));
            c = std::move(out);
            return;
        }
        std::vector<int32_t> idx(n);
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (size_t i = 0; i < n; ++i) { idx[i] = (int32_t)i; for (int r = 0; r < 3; ++r) { lo[r] = std::min(lo[r], c.col(i)[r]); hi[r] = std::max(hi[r], c.col(i)[r]); } }
        Work w(&c, this);
        build(w, idx, 0, n, lo, hi);
        if (keepNormals) w.out.addDescriptor("normals", 3, std::move(w.normals));
        c = std::move(w.out);
    }
};

// OctreeGridDataPointsFilter (created at OctreeMapperModule.cpp:12, applied at :38): octree over the bounding
// cube of the cloud, split until the node edge is <= maxSizeByNode or the node holds <= maxPointByNode points, one point per
// leaf, the cloud left in leaf-visiting order.  The tree lives on the device (icpmi_octree_sample / ICPMI_MOP_OCTREE):
// samplingMethod 0 (first point of the leaf) and 1 (random point, made reproducible) pick there; 2 (centroid: features and
// descriptors averaged over the leaf) and 3 (medoid: the point with the smallest summed distance to the others of its leaf)
// are formed here from the device's leaf assignment.
struct OctreeGridFilter : DataPointsFilter {
    float maxSize = 0.f; int method = 0; int maxPointByNode = 1;
    icpmi_handle h = nullptr;
    bool residentOp(icpmi_map_op& op, std::string&) const override {
        op = icpmi_map_op{}; op.type = ICPMI_MOP_OCTREE; op.i = method; op.f[0] = maxSize; op.f[1] = (float)maxPointByNode;
        return (method == 0 || method == 1) && maxSize >= 0.f && maxPointByNode <= 64;
    }
    static DataPoints gather(const DataPoints& c, const std::vector<int32_t>& order) {
        DataPoints out = c.createSimilarEmpty(order.size());
        out.features.resize(4 * order.size());
        for (auto& d : out.descriptors) d.data.resize((size_t)d.span * order.size());
        for (size_t j = 0; j < order.size(); ++j) {
            const size_t i = (size_t)order[j];
            std::copy(c.col(i), c.col(i) + 4, out.features.begin() + 4 * j);
            for (size_t k = 0; k < c.descriptors.size(); ++k) {
                const Descriptor& s = c.descriptors[k];
                std::copy(s.data.begin() + (size_t)s.span * i, s.data.begin() + (size_t)s.span * (i + 1), out.descriptors[k].data.begin() + (size_t)s.span * j);
            }
        }
        return out;
    }
    void inPlaceFilter(DataPoints& c) const override {
        const size_t n = c.getNbPoints();
        if (n == 0) return;
        if (!h) throw std::logic_error("OctreeGridDataPointsFilter needs a GPU context");
        std::vector<int32_t> order(n), leaf;
        if (method >= 2) leaf.resize(n);
        int64_t m = 0;
        GpuICPSequence::check(h, icpmi_octree_sample(h, c.features.data(), (int64_t)n, maxSize, maxPointByNode, method == 1 ? 1 : 0, order.data(),
                                                     method >= 2 ? leaf.data() : nullptr, &m));
        order.resize((size_t)m);
        if (method <= 1) { c = gather(c, order); return; }
        // members of every leaf, in list order
        std::vector<std::vector<int32_t>> members((size_t)m);
        for (size_t i = 0; i < n; ++i) members[(size_t)leaf[i]].push_back((int32_t)i);
        if (method == 3) { // medoid
            for (size_t l = 0; l < (size_t)m; ++l) {
                const auto& mem = members[l];
                double best = INFINITY; int32_t pick = mem[0];
                for (int32_t a : mem) {
                    double sum = 0;
                    for (int32_t b : mem) {
                        double d2 = 0;
                        for (int r = 0; r < 3; ++r) { const double e = (double)c.col((size_t)a)[r] - c.col((size_t)b)[r]; d2 += e * e; }
                        sum += std::sqrt(d2);
                    }
                    if (sum < best) { best = sum; pick = a; }
                }
                order[l] = pick;
            }
            c = gather(c, order);
            return;
        }
        DataPoints out = gather(c, order); // centroid: the representative's slot receives the leaf's averages
        for (size_t l = 0; l < (size_t)m; ++l) {
            const auto& mem = members[l];
            const double inv = 1.0 / (double)mem.size();
            for (int r = 0; r < 3; ++r) { double s2 = 0; for (int32_t i : mem) s2 += c.col((size_t)i)[r]; out.col(l)[r] = (float)(s2 * inv); }
            for (size_t k = 0; k < c.descriptors.size(); ++k) {
                const Descriptor& d = c.descriptors[k];
                for (int r = 0; r < d.span; ++r) { double s2 = 0; for (int32_t i : mem) s2 += d.data[(size_t)d.span * i + r]; out.descriptors[k].data[(size_t)d.span * l + r] = (float)(s2 * inv); }
            }
        }
        c = std::move(out);
    }
};

float getf(const yaml::Node& p, const char* k, float def) { return p[k] ? p[k].as<float>() : def; }
int geti(const yaml::Node& p, const char* k, int def) { return p[k] ? p[k].as<int>() : def; }

} // namespace

std::shared_ptr<DataPointsFilter> createDataPointsFilter(const std::string& name, const yaml::Node& p, icpmi_handle ctx)
{
    if (name == "DistanceLimitDataPointsFilter") {
        requireKnown(p, {"dim", "dist", "removeInside"}, name);
        auto f = std::make_shared<DistanceLimitFilter>();
        f->dim = geti(p, "dim", -1); f->dist = getf(p, "dist", 1.f); f->removeInside = geti(p, "removeInside", 1) != 0;
        if (f->dim > 2) throw InvalidParameter(name + ": dim out of range");
        return f;
    }
    if (name == "BoundingBoxDataPointsFilter") {
        requireKnown(p, {"xMin", "xMax", "yMin", "yMax", "zMin", "zMax", "removeInside"}, name);
        auto f = std::make_shared<BoundingBoxFilter>();
        f->lo[0] = getf(p, "xMin", -1); f->hi[0] = getf(p, "xMax", 1);
        f->lo[1] = getf(p, "yMin", -1); f->hi[1] = getf(p, "yMax", 1);
        f->lo[2] = getf(p, "zMin", -1); f->hi[2] = getf(p, "zMax", 1);
        f->removeInside = geti(p, "removeInside", 1) != 0;
        return f;
    }
    if (name == "AddDescriptorDataPointsFilter") {
        requireKnown(p, {"descriptorName", "descriptorDimension", "descriptorValues"}, name);
        auto f = std::make_shared<AddDescriptorFilter>();
        if (!p["descriptorName"]) throw InvalidParameter(name + ": descriptorName is required");
        f->name = p["descriptorName"].as<std::string>();
        f->dimension = geti(p, "descriptorDimension", 1);
        if (p["descriptorValues"].IsSequence()) for (const auto& v : p["descriptorValues"].seq) f->values.push_back(v.as<float>());
        else if (p["descriptorValues"].IsScalar()) f->values.push_back(p["descriptorValues"].as<float>());
        if ((int)f->values.size() != f->dimension) throw InvalidParameter(name + ": descriptorValues must have descriptorDimension entries");
        return f;
    }
    if (name == "CutAtDescriptorThresholdDataPointsFilter") {
        requireKnown(p, {"descName", "useLargerThan", "threshold"}, name);
        auto f = std::make_shared<CutAtDescriptorThresholdFilter>();
        f->name = p["descName"] ? p["descName"].as<std::string>() : "none";
        f->useLargerThan = geti(p, "useLargerThan", 1) != 0; f->threshold = getf(p, "threshold", 0.f);
        return f;
    }
    if (name == "SurfaceNormalDataPointsFilter") {
        requireKnown(p, {"knn", "maxDist", "epsilon", "keepNormals", "keepDensities", "keepEigenValues", "keepEigenVectors",
                         "keepMatchedIds", "keepMeanDist", "sortEigen", "smoothNormals"}, name);
        if (geti(p, "smoothNormals", 0) != 0) throw InvalidParameter(name + ": smoothNormals is not on the accelerated path");
        // r5: keepEigenValues / keepEigenVectors are served in ASCENDING eigenvalue order, i.e. together with sortEigen: 1; upstream's unsorted
        // order is whatever Eigen::EigenSolver returns for the matrix at hand and is not reproduced
        if ((geti(p, "keepEigenValues", 0) != 0 || geti(p, "keepEigenVectors", 0) != 0) && geti(p, "sortEigen", 0) == 0)
            throw InvalidParameter(name + ": keepEigenValues / keepEigenVectors are served with sortEigen: 1 only (the unsorted order is the eigen-solver's)");
        auto f = std::make_shared<SurfaceNormalFilter>();
        f->h = ctx; f->knn = geti(p, "knn", 5); f->keepDensities = geti(p, "keepDensities", 0) != 0;
        f->keepEigenValues = geti(p, "keepEigenValues", 0) != 0; f->keepEigenVectors = geti(p, "keepEigenVectors", 0) != 0;
        f->keepMatchedIds = geti(p, "keepMatchedIds", 0) != 0; f->keepMeanDist = geti(p, "keepMeanDist", 0) != 0;
        return f;
    }
    if (name == "RandomSamplingDataPointsFilter") {
        requireKnown(p, {"prob", "randomSamplingMethod", "seed"}, name);
        auto f = std::make_shared<RandomSamplingFilter>();
        f->prob = getf(p, "prob", 0.75f); f->method = geti(p, "randomSamplingMethod", 0); f->seed = geti(p, "seed", -1);
        if (!(f->prob >= 0.f && f->prob <= 1.f) || f->method < 0 || f->method > 1 || f->seed < -1) throw InvalidParameter(name + ": parameter out of range");
        return f;
    }
    if (name == "SamplingSurfaceNormalDataPointsFilter") {
        requireKnown(p, {"ratio", "knn", "samplingMethod", "maxBoxDim", "averageExistingDescriptors", "keepNormals", "keepDensities", "keepEigenValues",
                         "keepEigenVectors", "seed"}, name);
        for (const char* k : {"keepDensities", "keepEigenValues", "keepEigenVectors"})
            if (geti(p, k, 0) != 0) throw InvalidParameter(name + ": " + k + " is not supported");
        auto f = std::make_shared<SamplingSurfaceNormalFilter>();
        f->h = ctx;
        f->ratio = getf(p, "ratio", 0.5f); f->knn = geti(p, "knn", 7); f->method = geti(p, "samplingMethod", 0);
        f->maxBoxDim = p["maxBoxDim"] ? p["maxBoxDim"].as<float>() : INFINITY;
        f->averageDescriptors = geti(p, "averageExistingDescriptors", 1) != 0; f->keepNormals = geti(p, "keepNormals", 1) != 0;
        f->seed = geti(p, "seed", 1);
        if (!(f->ratio > 0.f && f->ratio <= 1.f) || f->knn < 3 || f->method < 0 || f->method > 1) throw InvalidParameter(name + ": parameter out of range");
        return f;
    }
    if (name == "MaxDensityDataPointsFilter") {
        requireKnown(p, {"maxDensity", "seed"}, name);
        auto f = std::make_shared<MaxDensityFilter>();
        f->maxDensity = getf(p, "maxDensity", 10.f); f->seed = geti(p, "seed", 1);
        if (!(f->maxDensity > 0.f)) throw InvalidParameter(name + ": maxDensity must be > 0");
        return f;
    }
    if (name == "IdentityDataPointsFilter") return std::make_shared<IdentityFilter>();
    if (name == "ObservationDirectionDataPointsFilter") {
        requireKnown(p, {"x", "y", "z"}, name);
        auto f = std::make_shared<ObservationDirectionFilter>();
        f->c[0] = getf(p, "x", 0.f); f->c[1] = getf(p, "y", 0.f); f->c[2] = getf(p, "z", 0.f);
        return f;
    }
    if (name == "OrientNormalsDataPointsFilter") {
        requireKnown(p, {"towardCenter"}, name);
        auto f = std::make_shared<OrientNormalsFilter>();
        f->towardCenter = geti(p, "towardCenter", 1) != 0;
        return f;
    }
    if (name == "MinDistDataPointsFilter" || name == "MaxDistDataPointsFilter") {
        // the older names of DistanceLimitDataPointsFilter: MinDist{dim -1, minDist 1} keeps what lies beyond, MaxDist{dim -1, maxDist 1} within
        const bool isMin = name == "MinDistDataPointsFilter";
        requireKnown(p, {"dim", isMin ? "minDist" : "maxDist"}, name);
        auto f = std::make_shared<DistanceLimitFilter>();
        f->dim = geti(p, "dim", -1); f->dist = getf(p, isMin ? "minDist" : "maxDist", 1.f); f->removeInside = isMin;
        if (f->dim < -1 || f->dim > 2) throw InvalidParameter(name + ": dim must be in [-1, 2]");
        return f;
    }
    if (name == "RemoveNaNDataPointsFilter") return std::make_shared<RemoveNaNFilter>();
    if (name == "OctreeGridDataPointsFilter") {
        requireKnown(p, {"buildParallel", "maxPointByNode", "maxSizeByNode", "samplingMethod"}, name);
        auto f = std::make_shared<OctreeGridFilter>();
        f->maxSize = getf(p, "maxSizeByNode", 0.f); f->method = geti(p, "samplingMethod", 0);
        f->maxPointByNode = geti(p, "maxPointByNode", 1);
        if (f->maxSize < 0.f || f->maxPointByNode < 1 || f->method < 0 || f->method > 3) throw InvalidParameter(name + ": parameter out of range");
        f->h = ctx;
        return f;
    }
    throw InvalidParameter("unknown DataPointsFilter " + name);
}

void DataPointsFilters::apply(DataPoints& cloud, const DataPointsFilter* leading) const
{
    static const bool fuse = [] { const char* e = std::getenv("NIM_FUSED_INPUT_FILTERS"); return !e || std::atoi(e) != 0; }();
    std::vector<const DataPointsFilter*> chain;
    if (leading) chain.push_back(leading);
    for (const auto& f : filters) chain.push_back(f.get());
    size_t i = 0;
    while (i < chain.size()) {
        std::vector<icpmi_point_filter> run;
        icpmi_point_filter pf;
        while (fuse && ctx && i + run.size() < chain.size() && run.size() < 16 && chain[i + run.size()]->pointFilter(pf)) run.push_back(pf);
        if (run.size() >= 2 && cloud.getNbPoints() > 0) {
            std::vector<uint8_t> keep(cloud.getNbPoints());
            GpuICPSequence::check(ctx, icpmi_filter_points(ctx, cloud.features.data(), (int64_t)cloud.getNbPoints(), run.data(), (int32_t)run.size(), keep.data()));
            cloud.keepOnly(keep);
            i += run.size();
        } else {
            chain[i]->inPlaceFilter(cloud);
            ++i;
        }
    }
}

DataPointsFilters::DataPointsFilters(const yaml::Node& seq, icpmi_handle ctx_) : ctx(ctx_)
{
    icpmi_handle ctx = ctx_;
    if (!seq) return;
    if (!seq.IsSequence()) throw yaml::Exception("expected a sequence of filters");
    for (const auto& item : seq.seq) {
        auto e = singleEntry(item, "DataPointsFilter");
        filters.push_back(createDataPointsFilter(e.first, e.second, ctx));
    }
}

} // namespace nim
