// ShardedMapper.cpp -- see ShardedMapper.h.
#include "ShardedMapper.h"

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
    const Mat4 correction = icp.registerWithPrior(scan, estimatedPose);     // Mapper.cpp:197,213: identity while there is no map
    const Mat4 corrected = correction * estimatedPose;                       // :215
    // the merged set comes back once, for the cell manager: at most what all ranks can contribute (every rank its whole scan;
    // scans of the ranks are taken to be about this size -- a larger merge reports ICPMI_ERR_INVALID_ARG, nothing is lost)
    const size_t need = 4 * (size_t)(2 * (size_t)ranks * (scan.getNbPoints() + 1));
    if (merged.size() < need) merged.resize(need);
    int64_t mergedN = 0;
    icpmi_status s = icpmi_staged_merge_allgather(icp.handle(), correction.data(), minDist, normalsKnn, &acceptedLocal, &appended, &residentSize,
                                                  merged.data(), (int64_t)(merged.size() / 4), &mergedN);
    GpuICPSequence::check(icp.handle(), s);
    DataPoints grown((size_t)mergedN);
    std::copy(merged.begin(), merged.begin() + 4 * (size_t)mergedN, grown.features.begin());
    Map::binIntoCells(grown, [&](const std::string& id, DataPoints&& cell) {
        DataPoints old = cellManager->retrieveCell(id);
        if (old.getNbPoints() == 0) cellManager->saveCell(id, cell);
        else { old.concatenate(cell); cellManager->saveCell(id, old); }
    });
    return corrected;
}

} // namespace nim
