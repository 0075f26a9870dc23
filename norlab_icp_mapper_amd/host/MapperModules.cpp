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
DynamicPointsMapperModule::DynamicPointsMapperModule(const yaml::Node& params, icpmi_handle ctx) : h(ctx)
{
    requireKeys(params, {"thresholdDynamic", "alpha", "beta", "beamHalfAngle", "epsilonA", "epsilonD", "sensorMaxRange"},
                "DynamicPointsMapperModule");
    auto get = [&](const char* k, float& v) { if (params[k]) v = params[k].as<float>(); };
    get("thresholdDynamic", thresholdDynamic); get("alpha", alpha); get("beta", beta); get("beamHalfAngle", beamHalfAngle);
    get("epsilonA", epsilonA); get("epsilonD", epsilonD); get("sensorMaxRange", sensorMaxRange);
    if (!(beamHalfAngle > 0.f)) throw InvalidParameter("DynamicPointsMapperModule: beamHalfAngle must be > 0");
}

void DynamicPointsMapperModule::inPlaceUpdateMap(const DataPoints& input, DataPoints& map, const Mat4& pose)
{
    if (!input.descriptorExists("probabilityDynamic"))
        throw InvalidField("Missing field 'probabilityDynamic' in input point cloud. You can add it with the AddDescriptorDataPointsFilter in your input filters.");
    if (!map.descriptorExists("normals"))
        throw InvalidField("Missing field 'normals' in map point cloud. You can add it with the SurfaceNormalDataPointsFilter in your post filters.");
    if (!map.descriptorExists("probabilityDynamic")) throw InvalidField("Missing field 'probabilityDynamic' in map point cloud.");
    if (input.getNbPoints() == 0 || map.getNbPoints() == 0) return;

    // the whole update runs on the GPU (csrc/ops.hip: angular bucket grid over the beams, one lane per map point)
    const Mat4 toSensor = pose.inverse();
    const Descriptor& normals = map.getDescriptorByName("normals");
    Descriptor& probDyn = map.getDescriptorByName("probabilityDynamic");
    const size_t m = map.getNbPoints();
    icpmi_dynpts_params prm{thresholdDynamic, alpha, beta, beamHalfAngle, epsilonA, epsilonD, sensorMaxRange};
    const float* nptr = normals.data.data();
    if (normals.span != 3) throw InvalidField("descriptor 'normals' must have 3 rows");
    std::vector<float> prob(m);
    for (size_t i = 0; i < m; ++i) prob[i] = probDyn.data[(size_t)probDyn.span * i];
    GpuICPSequence::check(h, icpmi_dynamic_points_update(h, &prm, toSensor.data(), input.features.data(), (int64_t)input.getNbPoints(),
                                                         map.features.data(), nptr, (int64_t)m, prob.data()));
    for (size_t i = 0; i < m; ++i) probDyn.data[(size_t)probDyn.span * i] = prob[i];
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
