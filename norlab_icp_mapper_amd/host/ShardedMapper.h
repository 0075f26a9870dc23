// ShardedMapper.h -- one rank of the scan-sharded mapping mode (BASELINE config 5): one process and one GPU per
// rank, the map replicated in HBM, every rank registering its own scan stream (`Mapper::processInput`, Mapper.cpp:194-238, with
// the PointDistance update of MapperModules/PointDistanceMapperModule.cpp:28-50), the accepted points of all ranks all-gathered
// over RCCL inside the library (icpmi_staged_merge_allgather) and merged into every replica, then binned into 20 m cells for
// this rank's CellManager (Map.cpp:206-229, RAMCellManager.cpp:13-16).  No reference analogue: the reference is one process.
#pragma once
#include <vector>
#include <memory>
#include <string>
#include <unordered_map>

#include "IcpSequence.h"
#include "Map.h"

namespace nim {

// The CellManager of a replica (r6): a cell is a list of RUNS {offset, count} of the handle's device-resident cell log
// (icpmi_staged_bin_cells bins every epoch's merged set on the device; only the {ijk, offset, count} table reaches the host), fetched
// when somebody asks for the cell.  saveCell / retrieveCell / clearAllCells keep RAMCellManager's meaning (RAMCellManager.cpp:3-31):
// a cell saved from the host replaces whatever the log held for it.
class ResidentCellManager : public CellManager {
public:
    explicit ResidentCellManager(icpmi_handle handle) : h(handle) {}
    std::vector<std::string> getAllCellIds() const override;
    void saveCell(const std::string& cellId, const DataPoints& cell) override { hostCells[cellId] = cell; runs.erase(cellId); }
    DataPoints retrieveCell(const std::string& cellId) const override;
    void clearAllCells() override;
    void addRun(const std::string& cellId, int64_t offset, int64_t count) { runs[cellId].push_back({offset, count}); }
    int64_t pointsInLog() const;
private:
    struct Run { int64_t offset, count; };
    icpmi_handle h;
    std::unordered_map<std::string, std::vector<Run>> runs;
    std::unordered_map<std::string, DataPoints> hostCells;   // cells saved from the host (+ epochs that touched more cells than the device path takes)
};

class ShardedMapper {
public:
    // icpNode: the `icp:` sub-tree of a mapper configuration (empty: PM::ICPSequence::setDefault's chain without its filters)
    ShardedMapper(const yaml::Node& icpNode, float minDistNewPoint, int surfaceNormalKnn, int device = 0);

    static icpmi_comm_id createCommunicatorId();                          // rank 0; hand the 128 bytes to the other ranks
    void initCommunicator(const icpmi_comm_id& id, int nRanks, int rank); // collective

    bool setMap(const DataPoints& map);                                   // the same cloud on every rank
    // one epoch (collective when a communicator is set: every rank calls it once per epoch, with an EMPTY cloud when it has no
    // scan left): returns the corrected pose of THIS rank's scan.  A registration error is thrown after the exchange.
    Mat4 processScan(const DataPoints& scanInSensorFrame, const Mat4& estimatedPose);

    int64_t mapSize() const { return residentSize; }
    int64_t lastAcceptedLocal() const { return acceptedLocal; }
    int64_t lastAppended() const { return appended; }
    const icpmi_stats& lastIcpStats() const { return icp.stats(); }
    DataPoints getMap() const { return icp.downloadMap(); }
    CellManager& cells() { return *cellManager; }                         // every point the epochs appended, by 20 m cell
    int64_t lastCellsTouched() const { return cellsTouched; }             // cells the last epoch's merged set fell into

private:
    GpuICPSequence icp;
    std::unique_ptr<ResidentCellManager> cellManager;
    float minDist;
    int normalsKnn;
    int64_t residentSize = 0, acceptedLocal = 0, appended = 0, cellsTouched = 0;
    int ranks = 1;                 // size of the communicator (1 until initCommunicator)
    std::vector<float> merged;     // host copy of a merged set: only when an epoch touches more cells than the device path takes
    std::vector<int32_t> cellIjk; std::vector<int64_t> cellOff, cellCnt;   // the table of the last epoch (kept between epochs)
};

} // namespace nim
