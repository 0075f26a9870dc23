// ShardedMapper.cpp -- see ShardedMapper.h.
#include "ShardedMapper.h"

#include <exception>
#include <vector>

namespace nim {

ShardedMapper::ShardedMapper(const yaml::Node& icpNode, float minDistNewPoint, int surfaceNormalKnn, int device)
    : icp(device), cellManager(new RAMCellManager()), minDist(minDistNewPoint), normalsKnn(surfaceNormalKnn)
{
    if (minDistNewPoint < 0.f) throw InvalidParameter("ShardedMapper: minDistNewPoint must be >= 0");
    if (icpNode) icp.loadFromYamlNode(icpNode);
    if (icp.hasReadingFilters() || icp.hasReferenceFilters())
        throw InvalidParameter("ShardedMapper: the staged epoch does not run DataPointsFilters inside the ICP chain; filter the scans first");
}

icpmi_comm_id ShardedMapper::createCommunicatorId()
{
    icpmi_comm_id id;
    GpuICPSequence::check(nullptr, icpmi_comm_get_unique_id(&id));
    return id;
}

void ShardedMapper::initCommunicator(const icpmi_comm_id& id, int nRanks, int rank)
{
    GpuICPSequence::check(icp.handle(), icpmi_comm_init(icp.handle(), &id, nRanks, rank));
    ranks = nRanks;
}

bool ShardedMapper::setMap(const DataPoints& map)
{
    DataPoints m = map;
    if (normalsKnn > 0 && m.getNbPoints() > 0 && !m.descriptorExists("normals")) {
        std::vector<float> n3(3 * m.getNbPoints());
        GpuICPSequence::check(icp.handle(), icpmi_surface_normals(icp.handle(), m.features.data(), (int64_t)m.getNbPoints(), normalsKnn, n3.data()));
        m.addDescriptor("normals", 3, std::move(n3));
    }
    const bool ok = icp.setMap(m);
    if (ok) residentSize = (int64_t)m.getNbPoints();
    return ok;
}

Mat4 ShardedMapper::processScan(const DataPoints& scan, const Mat4& estimatedPose)
{
    // The epoch below is a collective: EVERY rank must enter it once, also a rank that has run out of scans (an empty cloud)
    // and a rank whose registration throws (an ordinary PM::ConvergenceError) -- otherwise its peers wait in the all-gather
    // forever.  Such a rank contributes nothing, still appends what the others accepted, and reports its own
    // error once the epoch is through.
    Mat4 correction = Mat4::identity();
    std::exception_ptr failure;
    bool contribute = scan.getNbPoints() > 0;
    if (contribute) {
        try {
            correction = icp.registerWithPrior(scan, estimatedPose);        // Mapper.cpp:197,213: identity while there is no map
        } catch (...) {
            failure = std::current_exception();
            contribute = false;
        }
    }
    if (!contribute) (void)icpmi_stage_discard(icp.handle());               // nothing staged: count 0 in the exchange
    const Mat4 corrected = correction * estimatedPose;                       // :215
    int64_t mergedN = 0;
    // the merged set stays on the device; its size is only known after the exchange, so it is fetched in a second call (a
    // host buffer sized from this rank's scan alone was too small next to larger ranks and let replicas diverge)
    icpmi_status s = icpmi_staged_merge_allgather(icp.handle(), contribute ? correction.data() : nullptr, minDist, normalsKnn, &acceptedLocal,
                                                  &appended, &residentSize, nullptr, 0, &mergedN);
    GpuICPSequence::check(icp.handle(), s);
    if (mergedN > 0) {
        if (merged.size() < 4 * (size_t)mergedN) merged.resize(4 * (size_t)mergedN);
        GpuICPSequence::check(icp.handle(), icpmi_staged_merged_points(icp.handle(), merged.data(), mergedN, &mergedN));
        DataPoints grown((size_t)mergedN);
        std::copy(merged.begin(), merged.begin() + 4 * (size_t)mergedN, grown.features.begin());
        Map::binIntoCells(grown, [&](const std::string& id, DataPoints&& cell) {
            DataPoints old = cellManager->retrieveCell(id);
            if (old.getNbPoints() == 0) cellManager->saveCell(id, cell);
            else { old.concatenate(cell); cellManager->saveCell(id, old); }
        });
    }
    if (failure) std::rethrow_exception(failure);
    return corrected;
}

} // namespace nim
