// ShardedMapper.h -- one rank of the scan-sharded mapping mode (BASELINE config 5): one process and one GPU per
// rank, the map replicated in HBM, every rank registering its own scan stream (`Mapper::processInput`, Mapper.cpp:194-238, with
// the PointDistance update of MapperModules/PointDistanceMapperModule.cpp:28-50), the accepted points of all ranks all-gathered
// over RCCL inside the library (icpmi_staged_merge_allgather) and merged into every replica, then binned into 20 m cells for
// this rank's CellManager (Map.cpp:206-229, RAMCellManager.cpp:13-16).  No reference analogue: the reference is one process.
#pragma once
#include <vector>
#include <memory>
#include <string>

#include "IcpSequence.h"
#include "Map.h"

namespace nim {

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

private:
    GpuICPSequence icp;
    std::unique_ptr<CellManager> cellManager;
    float minDist;
    int normalsKnn;
    int64_t residentSize = 0, acceptedLocal = 0, appended = 0;
    int ranks = 1;                 // size of the communicator (1 until initCommunicator)
    std::vector<float> merged;     // what all ranks accepted in the last epoch, for the cell manager (kept between epochs)
};

} // namespace nim
