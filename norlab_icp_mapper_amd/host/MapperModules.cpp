// MapperModules.cpp -- the three built-in map-update operators and their registrar.
#include "MapperModule.h"

#include <algorithm>
#include <cmath>
#include <limits>

namespace nim {

static void requireKeys(const yaml::Node& params, std::initializer_list<const char*> known, const std::string& who)
{
    if (!params.IsMap()) return;
    for (const auto& kv : params.map) {
        bool ok = false;
        for (const char* k : known) ok |= kv.first == k;
        if (!ok) throw InvalidParameter(who + ": unknown parameter " + kv.first); // Parametrizable::InvalidParameter
    }
}

// ---- PointDistanceMapperModule ------------------------------------------------------------------
PointDistanceMapperModule::PointDistanceMapperModule(const yaml::Node& params, icpmi_handle ctx) : h(ctx)
{
    requireKeys(params, {"minDistNewPoint"}, "PointDistanceMapperModule");
    if (params["minDistNewPoint"]) minDistNewPoint = params["minDistNewPoint"].as<float>();
    if (minDistNewPoint < 0.f) throw InvalidParameter("PointDistanceMapperModule: minDistNewPoint must be >= 0");
}

void PointDistanceMapperModule::inPlaceUpdateMap(const DataPoints& input, DataPoints& map, const Mat4&)
{
    const size_t n = input.getNbPoints();
    if (n == 0) return;
    std::vector<uint8_t> keep(n, 1);
    GpuICPSequence::check(h, icpmi_point_distance_keep(h, map.features.data(), (int64_t)map.getNbPoints(), input.features.data(),
                                                       (int64_t)n, minDistNewPoint, keep.data()));
    DataPoints kept = input;
    kept.keepOnly(keep);
    map.concatenate(kept);
}

// ---- OctreeMapperModule -------------------------------------------------------------------------
OctreeMapperModule::OctreeMapperModule(const yaml::Node& params, icpmi_handle ctx)
{
    requireKeys(params, {"buildParallel", "maxPointByNode", "maxSizeByNode", "samplingMethod"}, "OctreeMapperModule");
    octreeFilter = createDataPointsFilter("OctreeGridDataPointsFilter", params, ctx);
}

void OctreeMapperModule::inPlaceUpdateMap(const DataPoints& input, DataPoints& map, const Mat4&)
{
    map.concatenate(input);
    octreeFilter->inPlaceFilter(map);
}

// ---- DynamicPointsMapperModule ------------------------------------------------------------------
DynamicPointsMapperModule::DynamicPointsMapperModule(const yaml::Node& params, icpmi_handle ctx) : transformation(ctx)
{
    requireKeys(params, {"thresholdDynamic", "alpha", "beta", "beamHalfAngle", "epsilonA", "epsilonD", "sensorMaxRange"},
                "DynamicPointsMapperModule");
    auto get = [&](const char* k, float& v) { if (params[k]) v = params[k].as<float>(); };
    get("thresholdDynamic", thresholdDynamic); get("alpha", alpha); get("beta", beta); get("beamHalfAngle", beamHalfAngle);
    get("epsilonA", epsilonA); get("epsilonD", epsilonD); get("sensorMaxRange", sensorMaxRange);
    if (!(beamHalfAngle > 0.f)) throw InvalidParameter("DynamicPointsMapperModule: beamHalfAngle must be > 0");
}

namespace {

struct Spherical { float radius, elevation, azimuth; };

inline Spherical toSpherical(const float* p)
{
    Spherical s;
    s.radius = std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    s.elevation = std::asin(p[2] / s.radius);
    s.azimuth = std::atan2(p[1], p[0]);
    return s;
}

// exact nearest neighbour in the (elevation, azimuth) plane within `radius`, no wrap-around at +-pi
// (DynamicPointsMapperModule.cpp:75-78 does a plain 2-D kd-tree search): bucket grid of cell = radius
class AngularGrid {
public:
    AngularGrid(const std::vector<Spherical>& pts, float radius) : pts(pts), cell(radius)
    {
        ne = (int)std::floor(3.14159265358979f / cell) + 2;
        na = (int)std::floor(6.28318530717959f / cell) + 2;
        start.assign((size_t)ne * na + 1, 0);
        for (const auto& s : pts) ++start[key(s) + 1];
        for (size_t i = 1; i < start.size(); ++i) start[i] += start[i - 1];
        order.resize(pts.size());
        std::vector<uint32_t> fill(start.begin(), start.end() - 1);
        for (uint32_t i = 0; i < pts.size(); ++i) order[fill[key(pts[i])]++] = i;
    }
    // returns squared angular distance (float, as libnabo) or +inf; id through `best`
    float nearest(const Spherical& q, int& best) const
    {
        const int ce = ecell(q.elevation), ca = acell(q.azimuth);
        const float r2 = cell * cell;
        float bd = std::numeric_limits<float>::infinity();
        best = -1;
        for (int de = -1; de <= 1; ++de)
            for (int da = -1; da <= 1; ++da) {
                const int e = ce + de, a = ca + da;
                if (e < 0 || e >= ne || a < 0 || a >= na) continue;
                const size_t k = (size_t)e * na + a;
                for (uint32_t j = start[k]; j < start[k + 1]; ++j) {
                    const uint32_t i = order[j];
                    const float d0 = q.elevation - pts[i].elevation, d1 = q.azimuth - pts[i].azimuth;
                    const float d = d0 * d0 + d1 * d1;
                    if (d <= r2 && (d < bd || (d == bd && (int)i < best))) { bd = d; best = (int)i; }
                }
            }
        return bd;
    }
private:
    int ecell(float e) const { return std::min(ne - 1, std::max(0, (int)std::floor((e + 1.5707963267949f) / cell))); }
    int acell(float a) const { return std::min(na - 1, std::max(0, (int)std::floor((a + 3.14159265358979f) / cell))); }
    size_t key(const Spherical& s) const { return (size_t)ecell(s.elevation) * na + acell(s.azimuth); }
    const std::vector<Spherical>& pts;
    float cell;
    int ne, na;
    std::vector<uint32_t> start, order;
};

} // namespace

void DynamicPointsMapperModule::inPlaceUpdateMap(const DataPoints& input, DataPoints& map, const Mat4& pose)
{
    if (!input.descriptorExists("probabilityDynamic"))
        throw InvalidField("Missing field 'probabilityDynamic' in input point cloud. You can add it with the AddDescriptorDataPointsFilter in your input filters.");
    if (!map.descriptorExists("normals"))
        throw InvalidField("Missing field 'normals' in map point cloud. You can add it with the SurfaceNormalDataPointsFilter in your post filters.");
    if (!map.descriptorExists("probabilityDynamic")) throw InvalidField("Missing field 'probabilityDynamic' in map point cloud.");
    const float eps = 0.0001f;

    const Mat4 toSensor = pose.inverse();
    const DataPoints inputS = transformation.compute(input, toSensor);
    std::vector<Spherical> beams(inputS.getNbPoints());
    for (size_t i = 0; i < beams.size(); ++i) beams[i] = toSpherical(inputS.col(i));
    if (beams.empty()) return;

    const DataPoints mapS = transformation.compute(map, toSensor); // rotates the normals too
    const Descriptor& normals = mapS.getDescriptorByName("normals");
    Descriptor& probDyn = map.getDescriptorByName("probabilityDynamic");

    const float searchRadius = 2 * beamHalfAngle;
    AngularGrid grid(beams, searchRadius);
    const size_t m = map.getNbPoints();
    for (size_t i = 0; i < m; ++i) {
        const float* mp = mapS.col(i);
        const float mapNorm = std::sqrt(mp[0] * mp[0] + mp[1] * mp[1] + mp[2] * mp[2]);
        if (!(mapNorm < sensorMaxRange)) continue; // range cull (lines 60-69)
        int beam = -1;
        const float angDist2 = grid.nearest(toSpherical(mp), beam);
        if (beam < 0) continue; // no beam within 2 * beamHalfAngle

        const float* ip = inputS.col((size_t)beam);
        const float inputNorm = std::sqrt(ip[0] * ip[0] + ip[1] * ip[1] + ip[2] * ip[2]);
        const float dx = ip[0] - mp[0], dy = ip[1] - mp[1], dz = ip[2] - mp[2];
        const float delta = std::sqrt(dx * dx + dy * dy + dz * dz);
        const float d_max = epsilonA * inputNorm;
        const float* nrm = &normals.data[3 * i];
        const float ndot = (nrm[0] * mp[0] + nrm[1] * mp[1] + nrm[2] * mp[2]) / mapNorm;

        // the weights of Pomerleau et al. 2014 as the reference evaluates them (double where it writes `1.`)
        const float w_v = (float)(eps + (1. - eps) * std::fabs(ndot));
        const float w_d1 = (float)(eps + (1. - eps) * (1. - std::sqrt(angDist2) / (2 * beamHalfAngle)));
        const float offset = delta - epsilonD;
        float w_d2 = 1.f;
        if (delta < epsilonD || mapNorm > inputNorm) w_d2 = eps;
        else if (offset < d_max) w_d2 = eps + (1 - eps) * offset / d_max;
        float w_p2 = eps;
        if (delta < epsilonD) w_p2 = 1.f;
        else if (offset < d_max) w_p2 = (float)(eps + (1. - eps) * (1. - offset / d_max));

        if ((inputNorm + epsilonD + d_max) >= mapNorm) {
            const float lastDyn = probDyn.data[(size_t)probDyn.span * i];
            const float c1 = 1 - (w_v * w_d1);
            const float c2 = w_v * w_d1;
            float probDynamic, probStatic;
            if (lastDyn < thresholdDynamic) {
                probDynamic = c1 * lastDyn + c2 * w_d2 * ((1 - alpha) * (1 - lastDyn) + beta * lastDyn);
                probStatic = c1 * (1 - lastDyn) + c2 * w_p2 * (alpha * (1 - lastDyn) + (1 - beta) * lastDyn);
            } else { // latched: once dynamic, always dynamic
                probDynamic = 1 - eps;
                probStatic = eps;
            }
            probDyn.data[(size_t)probDyn.span * i] = probDynamic / (probDynamic + probStatic);
        }
    }
}

// ---- registrar ----------------------------------------------------------------------------------
std::shared_ptr<MapperModule> MapperModuleRegistrar::create(const std::string& name, const yaml::Node& params, icpmi_handle ctx) const
{
    auto it = factories.find(name);
    if (it == factories.end()) throw InvalidParameter("Trying to instanciate unknown element " + name + " from MapperModule registrar");
    return it->second(params, ctx);
}

std::shared_ptr<MapperModule> MapperModuleRegistrar::createFromYAML(const yaml::Node& item, icpmi_handle ctx) const
{
    if (item.IsScalar()) return create(item.scalar, yaml::Node(), ctx);
    if (!item.IsMap() || item.map.size() != 1) throw yaml::Exception("mapperModule entries must be single-key maps");
    return create(item.map[0].first, item.map[0].second, ctx);
}

} // namespace nim
