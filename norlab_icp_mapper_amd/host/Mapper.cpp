// Mapper.cpp -- see Mapper.h.
#include "Mapper.h"
#include <thread>

#include <algorithm>
#include <cmath>
#include <iterator>
#include <fstream>
#include <iostream>
#include <sstream>
#include <unordered_set>

namespace nim {

void Trajectory::save(const std::string& filename) const
{
    // positions as features, rotation columns as orientationX/Y/Z, stamps as the int64 time row "t" holding
    // `time_since_epoch().count()` (Trajectory.cpp:15-53: DataPoints(features, labels, descriptors, labels, times, timeLabels))
    const size_t n = poses.size();
    DataPoints cloud(n);
    std::vector<float> ox(3 * n), oy(3 * n), oz(3 * n);
    std::vector<std::int64_t> t(n);
    for (size_t i = 0; i < n; ++i) {
        for (int r = 0; r < 3; ++r) {
            cloud.col(i)[r] = poses[i](r, 3);
            ox[3 * i + r] = poses[i](r, 0); oy[3 * i + r] = poses[i](r, 1); oz[3 * i + r] = poses[i](r, 2);
        }
        t[i] = (std::int64_t)stamps[i].time_since_epoch().count();
    }
    cloud.addDescriptor("orientationX", 3, std::move(ox));
    cloud.addDescriptor("orientationY", 3, std::move(oy));
    if (dimension == 3) cloud.addDescriptor("orientationZ", 3, std::move(oz));
    cloud.addTime("t", 1, std::move(t));
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

namespace {

// every key must be one of `allowed`, at most once (the reference's validateYamlKeys messages)
void requireOnlyKeys(const yaml::Node& node, std::initializer_list<const char*> allowed)
{
    if (!node.IsMap()) throw yaml::Exception("Expected a YAML Map node.");
    std::unordered_set<std::string> seen;
    for (const auto& entry : node.map) {
        const std::string& key = entry.first;
        if (!seen.insert(key).second) throw yaml::Exception("Duplicated key: " + key);
        const bool known = std::any_of(allowed.begin(), allowed.end(), [&](const char* a) { return key == a; });
        if (!known) throw yaml::Exception("Invalid key: " + key);
    }
}

std::string slurp(const std::string& path)
{
    std::ifstream in(path.c_str());
    if (!in) throw std::runtime_error("The input config file " + path + " does not exist");
    return std::string(std::istreambuf_iterator<char>(in), std::istreambuf_iterator<char>());
}

} // namespace

UpdatePolicy UpdatePolicy::fromYaml(const yaml::Node& uc)
{
    requireOnlyKeys(uc, {"type", "value"});
    for (const char* k : {"type", "value"})
        if (!uc[k]) throw yaml::Exception(std::string("Missing key: ") + k);
    // one row per condition: name, tag, admissible range of the value
    struct Row { const char* name; Kind kind; float lo, hi; };
    static const Row rows[] = {
        {"distance", Distance, 0.f, INFINITY},
        {"overlap", Overlap, 0.f, 1.f},
        {"delay", Delay, 0.f, INFINITY},
    };
    const std::string type = uc["type"].as<std::string>();
    const float value = uc["value"].as<float>();
    for (const Row& r : rows) {
        if (type != r.name) continue;
        if (!(value >= r.lo && value <= r.hi)) throw yaml::Exception("Invalid map update " + type + ": " + std::to_string(value));
        UpdatePolicy p; p.kind = r.kind; p.threshold = value;
        return p;
    }
    throw yaml::Exception("Invalid map update condition: " + type);
}

void Mapper::loadYamlConfig(const std::string& configFilePath) { loadYamlConfigFromString(slurp(configFilePath)); }

void Mapper::loadYamlConfigFromString(const std::string& text)
{
    const yaml::Node root = yaml::Load(text);
    requireOnlyKeys(root, {"icp", "input", "post", "mapper"});
    const auto announceMissing = [](const char* what) { std::cout << what << std::endl; };

    // 1. the registration chain; loadFromYamlNode / setDefault re-create the GPU-side chain, so nobody may be using it
    {
        std::lock_guard<std::mutex> g(icpMapLock);
        if (const yaml::Node& chain = root["icp"]) icp.loadFromYamlNode(chain);
        else { announceMissing("icp config not found, using default"); icp.setDefault(); }
    }
    // 2. everything that shares the GPU context of that chain
    transformation = RigidTransformation(icp.handle());
    struct FilterSection { const char* key; DataPointsFilters* target; const char* missing; };
    const FilterSection sections[] = {
        {"input", &inputFilters, "Input config not found, using empty configuration."},
        {"post", &mapPostFilters, "Post config not found, using empty configuration."},
    };
    for (const FilterSection& sec : sections) {
        if (const yaml::Node& n = root[sec.key]) *sec.target = DataPointsFilters(n, icp.handle());
        else announceMissing(sec.missing);
    }
    // 3. map growth
    if (!root["mapper"]) announceMissing("mapper config not found, using default");
    configureMapperSection(root["mapper"]);
}

void Mapper::configureMapperSection(const yaml::Node& mapperNode)
{
    const bool present = (bool)mapperNode;
    if (present && mapperNode["updateCondition"]) updatePolicy = UpdatePolicy::fromYaml(mapperNode["updateCondition"]);
    else {
        if (present) std::cout << "Mapper update condition not found, using default configuration." << std::endl;
        updatePolicy = UpdatePolicy(); // distance, 1 m
    }
    if (present && mapperNode["sensorMaxRange"]) {
        const float range = mapperNode["sensorMaxRange"].as<float>();
        if (range < 0) throw yaml::Exception("Invalid sensor max range: " + std::to_string(range));
        map.setSensorMaxRange(range);
    }
    if (present && mapperNode["mapperModule"]) {
        const yaml::Node& modules = mapperNode["mapperModule"];
        if (!modules.IsSequence()) throw yaml::Exception("mapperModule must be a sequence");
        for (const auto& item : modules.seq) map.addMapperModule(registrar.createFromYAML(item, icp.handle()));
    } else {
        if (present) std::cout << "mapper module not found, using default" << std::endl;
        setDefaultMapperModule();
    }
}

void Mapper::setDefaultMapperModule()
{
    // PointDistanceMapperModule{minDistNewPoint: 0.15} (Mapper.cpp:330-336)
    map.addMapperModule(registrar.create("PointDistanceMapperModule", yaml::Load("{minDistNewPoint: 0.15}"), icp.handle()));
}

void Mapper::applyInputFilters(DataPoints& inputInSensorFrame)
{
    inputFilters.apply(inputInSensorFrame, radiusFilter.get()); // radius filter first, then the `input:` chain
}

void Mapper::processInput(const DataPoints& filteredInputInSensorFrame, const Mat4& estimatedPose, const TimePoint& timeStamp)
{
    // One upload per scan when the map update can run on the resident map: the scan is moved by the prior, registered and --
    // if the policy asks for it -- moved by the correction and merged, all on the GPU (icpmi_register_prior /
    // icpmi_map_update_staged).  Offline only: an asynchronous update would race the next scan for the staged buffer.
    // (a reading with `simpleSensorNoise` takes the host-pointer path below: icpmi_register_prior does not carry the noise row that
    // ErrorMinimizer::getOverlap() then needs, Mapper.cpp:219)
    if (!isOnline && !icp.chainNeedsReadingNormals() && !icp.hasReadingFilters() && !icp.readsReadingDescriptor() &&
        !filteredInputInSensorFrame.descriptorExists("simpleSensorNoise") &&
        map.canStageScan(filteredInputInSensorFrame, mapPostFilters, &estimatedPose)) {
        const bool bootstrap = map.isLocalPointCloudEmpty();
        Mat4 correction;
        lastScanGrewMap = false;
        lastUpdMs = 0.0;
        const auto tReg = std::chrono::steady_clock::now();
        {
            std::lock_guard<std::mutex> g(icpMapLock);
            lastSeenMapVersion = map.icpMapVersion();
            correction = icp.registerWithPrior(filteredInputInSensorFrame, estimatedPose); // identity while there is no map
        }
        lastRegMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tReg).count();
        const Mat4 correctedPose = bootstrap ? estimatedPose : correction * estimatedPose;
        map.updatePose(correctedPose);
        if (bootstrap || mapUpdateIsDue(timeStamp, correctedPose, icp.errorMinimizer->getOverlap())) {
            const auto tUpd = std::chrono::steady_clock::now();
            struct UpdClock { const std::chrono::steady_clock::time_point t0; double& out; ~UpdClock() { out = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } } updClock{tUpd, lastUpdMs};
            lastTimeMapWasUpdated = timeStamp;
            lastPoseWhereMapWasUpdated = correctedPose;
            lastScanGrewMap = true;
            if (map.canStageScan(filteredInputInSensorFrame, mapPostFilters, &correctedPose))
                map.updateLocalPointCloudStaged(filteredInputInSensorFrame, bootstrap ? Mat4::identity() : correction, correctedPose, mapPostFilters);
            else { // paging in updatePose changed the picture (e.g. the local cloud was emptied): the host path
                DataPoints inMap = transformation.compute(filteredInputInSensorFrame, estimatedPose);
                if (!bootstrap) inMap = transformation.compute(inMap, correction);
                map.updateLocalPointCloud(inMap, correctedPose, mapPostFilters);
            }
        }
        { std::lock_guard<std::mutex> g(poseLock); pose = correctedPose; }
        { std::lock_guard<std::mutex> g(trajectoryLock); trajectory.addPose(correctedPose, timeStamp); }
        return;
    }

    // scan into the map frame by the prior; ICP then returns a correction expressed in the map frame
    DataPoints scanInMap = transformation.compute(filteredInputInSensorFrame, estimatedPose);
    const bool bootstrap = map.isLocalPointCloudEmpty(); // nothing to register against: the prior is the pose
    Mat4 correction = Mat4::identity();
    lastScanGrewMap = false;
    lastUpdMs = 0.0;
    const auto tReg = std::chrono::steady_clock::now();
    if (!bootstrap) {
        std::lock_guard<std::mutex> g(icpMapLock);
        lastSeenMapVersion = map.icpMapVersion();
        correction = icp(scanInMap);
    }
    lastRegMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tReg).count();
    const Mat4 correctedPose = bootstrap ? estimatedPose : correction * estimatedPose;
    map.updatePose(correctedPose);
    const auto tUpd = std::chrono::steady_clock::now();
    if (bootstrap) { growMap(scanInMap, correctedPose, timeStamp); lastScanGrewMap = true; }
    else if (mapUpdateIsDue(timeStamp, correctedPose, icp.errorMinimizer->getOverlap())) {
        growMap(transformation.compute(scanInMap, correction), correctedPose, timeStamp);
        lastScanGrewMap = true;
    }
    if (lastScanGrewMap) lastUpdMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tUpd).count(); // (online: the hand-off only)

    // surface an exception of a finished asynchronous update here, like the reference's future.get()
    if (mapUpdateFuture.valid() && mapUpdateFuture.wait_for(std::chrono::milliseconds(1)) == std::future_status::ready) mapUpdateFuture.get();
    { std::lock_guard<std::mutex> g(poseLock); pose = correctedPose; }
    { std::lock_guard<std::mutex> g(trajectoryLock); trajectory.addPose(correctedPose, timeStamp); }
}

bool Mapper::mapUpdateIsDue(const TimePoint& now, const Mat4& poseNow, float overlap) const
{
    if (!isMapping.load()) return false;
    const bool updateInFlight = isOnline && mapUpdateFuture.valid() && mapUpdateFuture.wait_for(std::chrono::milliseconds(0)) != std::future_status::ready;
    if (updateInFlight) return false;
    float travelled2 = 0.f;
    const int euclideanDim = is3D ? 3 : 2;                                   // Mapper.cpp:266-269: topRightCorner(euclideanDim, 1)
    for (int axis = 0; axis < euclideanDim; ++axis) {
        const float step = poseNow(axis, 3) - lastPoseWhereMapWasUpdated(axis, 3);
        travelled2 += step * step;
    }
    const float seconds = std::chrono::duration<float>(now - lastTimeMapWasUpdated).count();
    return updatePolicy.due(std::sqrt(travelled2), overlap, seconds);
}

void Mapper::growMap(const DataPoints& inputInMapFrame, const Mat4& poseNow, const TimePoint& now)
{
    lastTimeMapWasUpdated = now;
    lastPoseWhereMapWasUpdated = poseNow;
    const bool inBackground = isOnline && !map.isLocalPointCloudEmpty(); // the very first map is built synchronously
    if (!inBackground) { map.updateLocalPointCloud(inputInMapFrame, poseNow, mapPostFilters); return; }
    // NIM_TEST_UPDATE_DELAY_MS (tests): the background update starts late, so that the next scans find it in flight -- they are
    // registered against the older map and start no update of their own (Mapper.cpp:257-260)
    static const int delayMs = [] { const char* e = std::getenv("NIM_TEST_UPDATE_DELAY_MS"); return e ? std::atoi(e) : 0; }();
    mapUpdateFuture = std::async(std::launch::async, [this, inputInMapFrame, poseNow] {
        if (delayMs > 0) std::this_thread::sleep_for(std::chrono::milliseconds(delayMs));
        map.updateLocalPointCloud(inputInMapFrame, poseNow, mapPostFilters);
    });
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
