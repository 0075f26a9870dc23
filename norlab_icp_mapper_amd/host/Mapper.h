// Mapper.h -- the orchestrator (reference: norlab_icp_mapper/Mapper.{h,cpp}): YAML configuration,
// input filters, the ICP call, map-update policy, pose / trajectory bookkeeping, optional asynchronous
// map update.  Same constructor arguments and public methods as the reference class.
#pragma once
#include <atomic>
#include <chrono>
#include <future>
#include <mutex>
#include <string>
#include <vector>

#include "IcpSequence.h"
#include "Map.h"
#include "MapperModule.h"

namespace nim {

using TimePoint = std::chrono::time_point<std::chrono::steady_clock>;

// Trajectory.{h,cpp}: poses + steady_clock stamps; save() writes the positions as features and the
// rotation columns as descriptors orientationX/Y/Z, the stamps as the int64 time row "t" (Trajectory.cpp:35-47)
class Trajectory {
public:
    explicit Trajectory(int dimension = 3) : dimension(dimension) {}
    void addPose(const Mat4& pose, TimePoint stamp) { poses.push_back(pose); stamps.push_back(stamp); }
    void clear() { poses.clear(); stamps.clear(); }
    size_t size() const { return poses.size(); }
    const Mat4& pose(size_t i) const { return poses[i]; }
    TimePoint stamp(size_t i) const { return stamps[i]; }
    void save(const std::string& filename) const;
private:
    int dimension;
    std::vector<Mat4> poses;
    std::vector<TimePoint> stamps;
};

// When the map is grown (`mapper: updateCondition:` of the YAML; reference defaults Mapper.h:19-20): parsed once into a
// tagged value instead of being re-interpreted from its string on every scan.
struct UpdatePolicy {
    enum Kind { Distance, Overlap, Delay } kind = Distance;
    float threshold = 1.0f; // metres travelled / overlap ratio / seconds, by kind
    static UpdatePolicy fromYaml(const yaml::Node& updateCondition); // throws yaml::Exception with the reference's messages
    bool due(float metresSinceUpdate, float overlap, float secondsSinceUpdate) const {
        return kind == Overlap ? overlap < threshold : (kind == Delay ? secondsSinceUpdate > threshold : metresSinceUpdate > threshold);
    }
};

class Mapper {
public:
    Mapper(const std::string& configFilePath, bool is3D, bool isOnline, bool isMapping, bool saveMapCellsOnHardDrive, int device = 0);
    ~Mapper();

    void applyInputFilters(DataPoints& inputInSensorFrame);                                   // Mapper.cpp:187-191
    void processInput(const DataPoints& inputInSensorFrame, const Mat4& estimatedPose, const TimePoint& timeStamp); // :194-238
    DataPoints getMap() { return map.getGlobalPointCloud(); }
    long residentMapUpdates() const { return map.residentUpdateCount(); }
    size_t localMapSize() { return map.localSize(); }
    // online mode: blocks until the asynchronous map update in flight (Mapper.cpp:282) and every scheduled cell load / unload
    // (Map.cpp:35-57) are done.  No reference analogue; a replay that calls it after every processInput makes the online
    // pipeline reproduce the offline trajectory (the reference's online results depend on thread timing).
    void waitForPendingWork() { if (mapUpdateFuture.valid()) mapUpdateFuture.wait(); map.waitForPaging(); }
    void setMap(const DataPoints& newMap);
    bool getNewLocalMap(DataPoints& mapOut) { return map.getNewLocalPointCloud(mapOut); }
    Mat4 getPose();
    bool getIsMapping() const { return isMapping.load(); }
    void setIsMapping(bool v) { isMapping.store(v); }
    Trajectory getTrajectory();
    void setDefaultMapperModule();
    void loadYamlConfig(const std::string& configFilePath);
    void loadYamlConfigFromString(const std::string& text);
    const icpmi_stats& lastIcpStats() const { return icp.stats(); }
    // the last processInput: version of the registration map it ran against (Map::icpMapVersion, read under the ICP lock) and
    // whether it started a map update -- what a replay needs to reproduce a free-running online run scan by scan
    long lastRegistrationMapVersion() const { return lastSeenMapVersion; }
    bool lastScanStartedMapUpdate() const { return lastScanGrewMap; }
    // wall time of the last processInput's two halves (offline mode; host clock around the calls, the GPU work is waited for inside them):
    // the ICP call (Mapper.cpp:213) and the map update it started (Map::updateLocalPointCloud, 0 when none was due).  Measurement only.
    double lastRegisterMs() const { return lastRegMs; }
    double lastMapUpdateMs() const { return lastUpdMs; }

private:
    void fillRegistrar();
    void rebuildRadiusFilter();
    void configureMapperSection(const yaml::Node& mapperNode);                        // the `mapper:` block, or an undefined node
    bool mapUpdateIsDue(const TimePoint& now, const Mat4& poseNow, float overlap) const; // Mapper.cpp:240-272
    void growMap(const DataPoints& inputInMapFrame, const Mat4& poseNow, const TimePoint& now); // Mapper.cpp:274-288

    GpuICPSequence icp;                 // first: the filters and modules share its GPU context
    std::mutex poseLock, trajectoryLock, icpMapLock;
    DataPointsFilters inputFilters, mapPostFilters;
    UpdatePolicy updatePolicy;
    bool is3D, isOnline;
    long lastSeenMapVersion = 0;
    bool lastScanGrewMap = false;
    double lastRegMs = 0.0, lastUpdMs = 0.0;
    std::atomic_bool isMapping;
    Map map;
    Mat4 pose = Mat4::identity();
    Trajectory trajectory;
    RigidTransformation transformation;
    std::shared_ptr<DataPointsFilter> radiusFilter;
    TimePoint lastTimeMapWasUpdated;
    Mat4 lastPoseWhereMapWasUpdated = Mat4::identity();
    mutable std::future<void> mapUpdateFuture;
    MapperModuleRegistrar registrar;
};

} // namespace nim
