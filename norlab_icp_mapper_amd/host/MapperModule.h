// MapperModule.h -- the plugin interface of the map-update chain and its three built-in operators
// (reference: norlab_icp_mapper/MapperModules/MapperModule.h:20-29, PointDistanceMapperModule.{h,cpp},
// OctreeMapperModule.{h,cpp}, DynamicPointsMapperModule.{h,cpp}; registration Mapper.cpp:9-13).
// The update methods receive the input already in the map frame.
#pragma once
#include <functional>
#include <map>
#include <memory>
#include <string>

#include "IcpSequence.h"
#include "PointCloud.h"
#include "Yaml.h"

namespace nim {

class MapperModule {
public:
    virtual ~MapperModule() = default;
    // non-destructive / in-place pairs, exactly the four virtuals of the reference
    virtual DataPoints createMap(const DataPoints& input, const Mat4& pose) { DataPoints out(input); inPlaceCreateMap(out, pose); return out; }
    virtual void inPlaceCreateMap(DataPoints& input, const Mat4& pose) = 0;
    virtual DataPoints updateMap(const DataPoints& input, const DataPoints& map, const Mat4& pose) { DataPoints out(map); inPlaceUpdateMap(input, out, pose); return out; }
    virtual void inPlaceUpdateMap(const DataPoints& input, DataPoints& map, const Mat4& pose) = 0;
    // true when the module can run as a step of icpmi_map_update_chain on the resident map (Map::residentPlan): fills
    // `op` and names the scalar descriptor it reads / writes (empty: none).  Plugins that keep the default stay on the host path.
    virtual bool residentOp(icpmi_map_op& op, std::string& scalarName) const { (void)op; (void)scalarName; return false; }
};

// Keeps the input points whose exact nearest map neighbour (self-match excluded, no radius) is at
// least minDistNewPoint away, and appends them (PointDistanceMapperModule.cpp:28-50).  The NN runs on
// the GPU through icpmi_point_distance_keep.
class PointDistanceMapperModule : public MapperModule {
public:
    PointDistanceMapperModule(const yaml::Node& params, icpmi_handle ctx);
    static std::string description() { return "Create a new map from the first scan. Update the map by adding the scan points that are farther than minDistNewPoint from every map point."; }
    void inPlaceCreateMap(DataPoints&, const Mat4&) override {} // the first scan is the map
    void inPlaceUpdateMap(const DataPoints& input, DataPoints& map, const Mat4& pose) override;
    bool residentOp(icpmi_map_op& op, std::string&) const override { op = icpmi_map_op{}; op.type = ICPMI_MOP_POINT_DISTANCE; op.f[0] = minDistNewPoint; return true; }
    float minDistNewPoint = 0.03f; // availableParameters default (PointDistanceMapperModule.h:24)
private:
    icpmi_handle h;
};

// map.concatenate(input) followed by the octree decimation filter (OctreeMapperModule.cpp:35-39).
class OctreeMapperModule : public MapperModule {
public:
    OctreeMapperModule(const yaml::Node& params, icpmi_handle ctx);
    static std::string description() { return "Concatenate the scan to the map, then keep one point per octree leaf."; }
    void inPlaceCreateMap(DataPoints& input, const Mat4& pose) override { DataPoints empty = input.createSimilarEmpty(); inPlaceUpdateMap(empty, input, pose); }
    void inPlaceUpdateMap(const DataPoints& input, DataPoints& map, const Mat4& pose) override;
    bool residentOp(icpmi_map_op& op, std::string& scalarName) const override { return octreeFilter->residentOp(op, scalarName); }
private:
    std::shared_ptr<DataPointsFilter> octreeFilter;
};

// Bayesian dynamic-point probability update of every map point seen again by the scan
// (DynamicPointsMapperModule.cpp:34-172).  Needs `probabilityDynamic` on the input and `normals` on
// the map.  Host implementation: the spherical-coordinate nearest beam is found on an angular grid.
class DynamicPointsMapperModule : public MapperModule {
public:
    DynamicPointsMapperModule(const yaml::Node& params, icpmi_handle ctx);
    static std::string description() { return "Update the probability of map points to be dynamic from the beams of the scan that see them."; }
    void inPlaceCreateMap(DataPoints&, const Mat4&) override {}
    void inPlaceUpdateMap(const DataPoints& input, DataPoints& map, const Mat4& pose) override;
    bool residentOp(icpmi_map_op& op, std::string& scalarName) const override {
        op = icpmi_map_op{}; op.type = ICPMI_MOP_DYNAMIC_POINTS;
        const float prm[7] = {thresholdDynamic, alpha, beta, beamHalfAngle, epsilonA, epsilonD, sensorMaxRange};
        for (int r = 0; r < 7; ++r) op.f[r] = prm[r];
        scalarName = "probabilityDynamic";
        return true;
    }
    float thresholdDynamic = 0.6f, alpha = 0.8f, beta = 0.99f, beamHalfAngle = 0.01f, epsilonA = 0.01f, epsilonD = 0.01f, sensorMaxRange = 200.f;
private:
    icpmi_handle h;
};

// DEF_REGISTRAR(MapperModule) / ADD_TO_REGISTRAR / createFromYAML (Mapper.h:69, Mapper.cpp:9-13,169)
class MapperModuleRegistrar {
public:
    using Factory = std::function<std::shared_ptr<MapperModule>(const yaml::Node& params, icpmi_handle ctx)>;
    void add(const std::string& name, Factory f) { factories[name] = std::move(f); }
    std::shared_ptr<MapperModule> createFromYAML(const yaml::Node& singleKeyMap, icpmi_handle ctx) const;
    std::shared_ptr<MapperModule> create(const std::string& name, const yaml::Node& params, icpmi_handle ctx) const;
private:
    std::map<std::string, Factory> factories;
};

} // namespace nim
