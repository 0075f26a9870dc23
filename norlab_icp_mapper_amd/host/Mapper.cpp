// Mapper.cpp -- see Mapper.h.
#include "Mapper.h"

#include <algorithm>
#include <fstream>
#include <iostream>
#include <sstream>
#include <unordered_set>

namespace nim {

void Trajectory::save(const std::string& filename) const
{
    // positions as features, rotation columns as orientationX/Y/Z, stamps as "t" (Trajectory.cpp:17-51;
    // upstream stores the stamps in DataPoints::times, this cloud type carries them as a descriptor)
    const size_t n = poses.size();
    DataPoints cloud(n);
    std::vector<float> ox(3 * n), oy(3 * n), oz(3 * n), t(n);
    for (size_t i = 0; i < n; ++i) {
        for (int r = 0; r < 3; ++r) {
            cloud.col(i)[r] = poses[i](r, 3);
            ox[3 * i + r] = poses[i](r, 0); oy[3 * i + r] = poses[i](r, 1); oz[3 * i + r] = poses[i](r, 2);
        }
        t[i] = (float)std::chrono::duration<double>(stamps[i].time_since_epoch()).count();
    }
    cloud.addDescriptor("orientationX", 3, std::move(ox));
    cloud.addDescriptor("orientationY", 3, std::move(oy));
    if (dimension == 3) cloud.addDescriptor("orientationZ", 3, std::move(oz));
    cloud.addDescriptor("t", 1, std::move(t));
    cloud.save(filename);
}

void Mapper::fillRegistrar()
{
    registrar.add("PointDistanceMapperModule", [](const yaml::Node& p, icpmi_handle h) { return std::make_shared<PointDistanceMapperModule>(p, h); });
    registrar.add("OctreeMapperModule", [](const yaml::Node& p, icpmi_handle h) { return std::make_shared<OctreeMapperModule>(p, h); });
    registrar.add("DynamicPointsMapperModule", [](const yaml::Node& p, icpmi_handle h) { return std::make_shared<DynamicPointsMapperModule>(p, h); });
}

Mapper::Mapper(const std::string& configFilePath, bool is3D_, bool isOnline_, bool isMapping_, bool saveMapCellsOnHardDrive, int device)
    : icp(device), is3D(is3D_), isOnline(isOnline_), isMapping(isMapping_),
      map(is3D_, isOnline_, saveMapCellsOnHardDrive, icp, icpMapLock), trajectory(is3D_ ? 3 : 2), transformation(icp.handle())
{
    fillRegistrar();
    loadYamlConfig(configFilePath);
    rebuildRadiusFilter();
}

Mapper::~Mapper()
{
    if (mapUpdateFuture.valid()) mapUpdateFuture.wait();
}

void Mapper::rebuildRadiusFilter()
{
    // DistanceLimitDataPointsFilter{dim: -1, dist: sensorMaxRange, removeInside: 0} (Mapper.cpp:27-31)
    std::ostringstream y;
    y << "{dim: -1, dist: " << map.getSensorMaxRange() << ", removeInside: 0}";
    radiusFilter = createDataPointsFilter("DistanceLimitDataPointsFilter", yaml::Load(y.str()), icp.handle());
}

void Mapper::validateYamlKeys(const yaml::Node& node, const std::vector<std::string>& validKeys) const
{
    if (!node.IsMap()) throw yaml::Exception("Expected a YAML Map node.");
    std::unordered_set<std::string> seen;
    for (const auto& kv : node.map) {
        if (seen.count(kv.first)) throw yaml::Exception("Duplicated key: " + kv.first);
        if (std::find(validKeys.begin(), validKeys.end(), kv.first) == validKeys.end()) throw yaml::Exception("Invalid key: " + kv.first);
        seen.insert(kv.first);
    }
}

void Mapper::loadYamlConfig(const std::string& configFilePath)
{
    std::ifstream ifs(configFilePath.c_str());
    if (ifs.fail()) throw std::runtime_error("The input config file " + configFilePath + " does not exist");
    std::stringstream ss;
    ss << ifs.rdbuf();
    loadYamlConfigFromString(ss.str());
}

void Mapper::loadYamlConfigFromString(const std::string& text)
{
    const yaml::Node node = yaml::Load(text);
    validateYamlKeys(node, {"icp", "input", "post", "mapper"});

    {
        // the ICP object is recreated by loadFromYamlNode / setDefault: nothing may hold the old context
        std::lock_guard<std::mutex> g(icpMapLock);
        if (node["icp"]) icp.loadFromYamlNode(node["icp"]);
        else { std::cout << "icp config not found, using default" << std::endl; icp.setDefault(); }
    }
    // filters, modules and transformations share the (new) GPU context
    transformation = RigidTransformation(icp.handle());
    if (node["input"]) inputFilters = DataPointsFilters(node["input"], icp.handle());
    else std::cout << "Input config not found, using empty configuration." << std::endl;
    if (node["post"]) mapPostFilters = DataPointsFilters(node["post"], icp.handle());
    else std::cout << "Post config not found, using empty configuration." << std::endl;

    if (node["mapper"]) {
        const yaml::Node& mapperNode = node["mapper"];
        if (mapperNode["updateCondition"]) {
            const yaml::Node& uc = mapperNode["updateCondition"];
            validateYamlKeys(uc, {"type", "value"});
            if (!uc["type"]) throw yaml::Exception("Missing key: type");
            if (!uc["value"]) throw yaml::Exception("Missing key: value");
            mapUpdateCondition = uc["type"].as<std::string>();
            const float value = uc["value"].as<float>();
            if (mapUpdateCondition == "distance") {
                if (value < 0) throw yaml::Exception("Invalid map update distance: " + std::to_string(value));
                mapUpdateDistance = value;
            } else if (mapUpdateCondition == "overlap") {
                if (value < 0 || value > 1) throw yaml::Exception("Invalid map update overlap: " + std::to_string(value));
                mapUpdateOverlap = value;
            } else if (mapUpdateCondition == "delay") {
                if (value < 0) throw yaml::Exception("Invalid map update delay: " + std::to_string(value));
                mapUpdateDelay = value;
            } else throw yaml::Exception("Invalid map update condition: " + mapUpdateCondition);
        } else {
            std::cout << "Mapper update condition not found, using default configuration." << std::endl;
            setDefaultMapUpdateConfig();
        }
        if (mapperNode["sensorMaxRange"]) {
            const float r = mapperNode["sensorMaxRange"].as<float>();
            if (r < 0) throw yaml::Exception("Invalid sensor max range: " + std::to_string(r));
            map.setSensorMaxRange(r);
        }
        if (mapperNode["mapperModule"]) {
            if (!mapperNode["mapperModule"].IsSequence()) throw yaml::Exception("mapperModule must be a sequence");
            for (const auto& item : mapperNode["mapperModule"].seq) map.addMapperModule(registrar.createFromYAML(item, icp.handle()));
        } else {
            std::cout << "mapper module not found, using default" << std::endl;
            setDefaultMapperModule();
        }
    } else {
        std::cout << "mapper config not found, using default" << std::endl;
        setDefaultMapperConfig();
    }
}

void Mapper::setDefaultMapperModule()
{
    // PointDistanceMapperModule{minDistNewPoint: 0.15} (Mapper.cpp:330-336)
    map.addMapperModule(registrar.create("PointDistanceMapperModule", yaml::Load("{minDistNewPoint: 0.15}"), icp.handle()));
}

void Mapper::applyInputFilters(DataPoints& inputInSensorFrame)
{
    radiusFilter->inPlaceFilter(inputInSensorFrame);
    inputFilters.apply(inputInSensorFrame);
}

void Mapper::processInput(const DataPoints& filteredInputInSensorFrame, const Mat4& estimatedPose, const TimePoint& timeStamp)
{
    const DataPoints input = transformation.compute(filteredInputInSensorFrame, estimatedPose);
    Mat4 correctedPose;
    if (map.isLocalPointCloudEmpty()) {
        correctedPose = estimatedPose;
        map.updatePose(correctedPose);
        updateMap(input, correctedPose, timeStamp);
    } else {
        Mat4 correction;
        {
            std::lock_guard<std::mutex> g(icpMapLock);
            correction = icp(input);
        }
        correctedPose = correction * estimatedPose;
        map.updatePose(correctedPose);
        if (shouldUpdateMap(timeStamp, correctedPose, icp.errorMinimizer->getOverlap()))
            updateMap(transformation.compute(input, correction), correctedPose, timeStamp);
    }
    if (mapUpdateFuture.valid() && mapUpdateFuture.wait_for(std::chrono::milliseconds(1)) == std::future_status::ready) mapUpdateFuture.get();
    {
        std::lock_guard<std::mutex> g(poseLock);
        pose = correctedPose;
    }
    {
        std::lock_guard<std::mutex> g(trajectoryLock);
        trajectory.addPose(correctedPose, timeStamp);
    }
}

bool Mapper::shouldUpdateMap(const TimePoint& currentTime, const Mat4& currentPose, float currentOverlap) const
{
    if (!isMapping.load()) return false;
    if (isOnline && mapUpdateFuture.valid() && mapUpdateFuture.wait_for(std::chrono::milliseconds(0)) != std::future_status::ready)
        return false; // the previous update is still running
    if (mapUpdateCondition == "overlap") return currentOverlap < mapUpdateOverlap;
    if (mapUpdateCondition == "delay") return (currentTime - lastTimeMapWasUpdated) > std::chrono::duration<float>(mapUpdateDelay);
    float d2 = 0.f;
    for (int r = 0; r < 3; ++r) { const float d = currentPose(r, 3) - lastPoseWhereMapWasUpdated(r, 3); d2 += d * d; }
    return std::sqrt(d2) > mapUpdateDistance;
}

void Mapper::updateMap(const DataPoints& currentInput, const Mat4& currentPose, const TimePoint& currentTimeStamp)
{
    lastTimeMapWasUpdated = currentTimeStamp;
    lastPoseWhereMapWasUpdated = currentPose;
    if (isOnline && !map.isLocalPointCloudEmpty())
        mapUpdateFuture = std::async(std::launch::async, &Map::updateLocalPointCloud, &map, currentInput, currentPose, mapPostFilters);
    else
        map.updateLocalPointCloud(currentInput, currentPose, mapPostFilters);
}

void Mapper::setMap(const DataPoints& newMap)
{
    map.setGlobalPointCloud(newMap);
    std::lock_guard<std::mutex> g(trajectoryLock);
    trajectory.clear();
}

Mat4 Mapper::getPose()
{
    std::lock_guard<std::mutex> g(poseLock);
    return pose;
}

Trajectory Mapper::getTrajectory()
{
    std::lock_guard<std::mutex> g(trajectoryLock);
    return trajectory;
}

} // namespace nim
