// Map.h -- local-map state and 20 m cell paging (reference: norlab_icp_mapper/Map.{h,cpp},
// CellManager.h, RAMCellManager.{h,cpp}).  Same public surface and constants; the ICP object behind
// it is the GPU-backed GpuICPSequence.
#pragma once
#include <atomic>
#include <climits>
#include <list>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "IcpSequence.h"
#include "MapperModule.h"
#include "PointCloud.h"

namespace nim {

// CellManager.h:15-18
class CellManager {
public:
    virtual ~CellManager() = default;
    virtual std::vector<std::string> getAllCellIds() const = 0;
    virtual void saveCell(const std::string& cellId, const DataPoints& cell) = 0;
    virtual DataPoints retrieveCell(const std::string& cellId) const = 0;
    virtual void clearAllCells() = 0;
};

// RAMCellManager.cpp:3-31: id -> cloud, saveCell overwrites, unknown id -> empty cloud
class RAMCellManager : public CellManager {
public:
    std::vector<std::string> getAllCellIds() const override {
        std::vector<std::string> ids;
        ids.reserve(cells.size());
        for (const auto& kv : cells) ids.push_back(kv.first);
        return ids;
    }
    void saveCell(const std::string& cellId, const DataPoints& cell) override { cells[cellId] = cell; }
    DataPoints retrieveCell(const std::string& cellId) const override {
        auto it = cells.find(cellId);
        return it == cells.end() ? DataPoints() : it->second;
    }
    void clearAllCells() override { cells.clear(); }
private:
    std::unordered_map<std::string, DataPoints> cells;
};

class Map {
public:
    static constexpr int BUFFER_SIZE = 2;                             // Map.h:30
    static constexpr float CELL_SIZE = 20.0f;                         // Map.h:31
    static constexpr float DEFAULT_SENSOR_MAX_RANGE = 200.f;          // Map.h:33

    Map(bool is3D, bool isOnline, bool saveCellsOnHardDrive, GpuICPSequence& icp, std::mutex& icpMapLock);
    ~Map();

    void updatePose(const Mat4& pose);                                                    // Map.cpp:246-460
    DataPoints getLocalPointCloud();
    void updateLocalPointCloud(DataPoints input, Mat4 pose, DataPointsFilters postFilters); // Map.cpp:502-534
    bool getNewLocalPointCloud(DataPoints& out);
    // processInput with the scan staged on the GPU: can this input / post-filter combination be updated on the resident map
    // (see tryResidentUpdate), and the update itself for the scan kept by GpuICPSequence::registerWithPrior
    // (pose: a planar map takes the resident chain only under an exactly planar pose, see planarInputs in Map.cpp)
    bool canStageScan(const DataPoints& inputInSensorFrame, const DataPointsFilters& postFilters, const Mat4* pose = nullptr);
    void updateLocalPointCloudStaged(const DataPoints& inputDescriptors, const Mat4& correction, const Mat4& pose, const DataPointsFilters& postFilters);
    DataPoints getGlobalPointCloud();                                                     // Map.cpp:552-573
    void setGlobalPointCloud(const DataPoints& cloud);                                    // Map.cpp:575-588
    bool isLocalPointCloudEmpty() const { return localPointCloudEmpty.load(); }
    void addMapperModule(std::shared_ptr<MapperModule> module) { mapperModuleVec.push_back(std::move(module)); }
    void setSensorMaxRange(float r) { sensorMaxRange = r; }
    float getSensorMaxRange() const { return sensorMaxRange; }
    long residentUpdateCount() const { return residentUpdates.load(); } // map updates that ran on the resident map
    // how many map UPDATES the registration map has taken (bumped under icpMapLock by updateLocalPointCloud{,Staged}; cell paging
    // and setGlobalPointCloud hand the map over too but are not counted): a registration that reads it under the same lock knows
    // which map it ran against (free-running online mode, tests)
    long icpMapVersion() const { return icpMapVersions.load(); }

    // grid arithmetic, public for the unit tests (Map.cpp:130-138,232-235,462-480)
    static int toGridCoordinate(float world) { return (int)std::floor(world / CELL_SIZE); }
    static int toInferiorGridCoordinate(float world, float range) { return (int)std::ceil(((world - range) / CELL_SIZE) - 1.0); }
    static int toSuperiorGridCoordinate(float world, float range) { return (int)std::floor((world + range) / CELL_SIZE); }
    // the points of `cloud` grouped by 20 m cell (Map.cpp:206-229), every cell handed to `sink` once
    static void binIntoCells(const DataPoints& cloud, const std::function<void(const std::string&, DataPoints&&)>& sink);
    static std::string cellId(int row, int column, int aisle) { return std::to_string(row) + "_" + std::to_string(column) + "_" + std::to_string(aisle); }

private:
    struct Box { int lo[3]; int hi[3]; };           // inclusive cell ranges: rows, columns, aisles
    struct Update { Box box; bool load; };

    void updateThreadFunction();
    void applyUpdate(const Update& u) { if (u.load) loadCells(u.box); else unloadCells(u.box); }
    void scheduleUpdate(const Update& u);
public:
    // points of the local cloud (the resident copy when the device runs ahead of the host copy)
    size_t localSize() { std::lock_guard<std::mutex> g(localPointCloudLock); return deviceAhead ? (size_t)residentCount : localPointCloud.getNbPoints(); }
    // online mode: blocks until the paging thread has applied every scheduled load / unload (no reference analogue: used by
    // replays that want the asynchronous pipeline drained between scans, so that an online run reproduces the offline one)
    void waitForPaging();
private:
    std::atomic_int updatesInFlight{0};
    void loadCells(Box box);
    void unloadCells(Box box);
    // Resident path (icpmi_map_update_chain): every mapper module and every post filter describes itself as a device
    // step (MapperModule::residentOp / DataPointsFilter::residentOp) -- the three built-in modules, SurfaceNormal and
    // CutAtDescriptorThreshold do; a plugin that does not sends the update down the host path.  The device copy then
    // runs ahead of localPointCloud, which is refreshed on the next host-side access.
    struct ResidentProgram { std::vector<icpmi_map_op> ops; int nModules = 0; std::string scalarName; bool computesNormals = false; };
    bool tryResidentUpdate(const DataPoints& input, const Mat4& pose, const DataPointsFilters& postFilters);
    bool residentPlan(const DataPoints& input, const DataPointsFilters& postFilters, ResidentProgram& prog) const; // lock held
    bool hostDescriptorsFollow(const DataPoints& input, const ResidentProgram& prog, bool first) const;          // lock held
    void dropLocalCloudAfterFailedUpdate();                                                                      // lock held
    void prepareResidentScalar(const ResidentProgram& prog, bool first);                                         // lock held
    void adoptResidentResult(const DataPoints& input, const ResidentProgram& prog, const std::vector<int32_t>& src, int64_t prefix,
                             int64_t mapSize, bool first);                                                        // lock held
    void syncLocalFromDevice(); // localPointCloudLock held
    bool deviceAhead = false;
    bool residentNormals = false;      // while deviceAhead: the device map carries `normals`
    std::string residentScalar;        // ... and this scalar descriptor (empty: none)
    int64_t residentCount = 0;         // ... and this many points
    std::atomic<long> residentUpdates{0};
    std::atomic<long> icpMapVersions{0};

    float sensorMaxRange = DEFAULT_SENSOR_MAX_RANGE;
    bool is3D, isOnline;
    GpuICPSequence& icp;
    std::mutex& icpMapLock;
    RigidTransformation transformation;
    DataPoints localPointCloud;
    std::mutex localPointCloudLock;
    std::unique_ptr<CellManager> cellManager;
    std::mutex cellManagerLock;
    std::unordered_set<std::string> loadedCellIds;
    int inferiorLast[3] = {0, 0, 0}, superiorLast[3] = {0, 0, 0}; // window edges at the last update, per axis
    bool newLocalPointCloudAvailable = false;
    std::atomic_bool localPointCloudEmpty{true};
    std::atomic_bool firstPoseUpdate{true};
    std::atomic_bool updateThreadLooping{true};
    std::thread updateThread;
    std::list<Update> updateList;
    std::mutex updateListLock;
    std::vector<std::shared_ptr<MapperModule>> mapperModuleVec;
};

} // namespace nim
