// ShardedMapper.cpp -- see ShardedMapper.h.
#include "ShardedMapper.h"

#include <exception>
#include <vector>

namespace nim {

std::vector<std::string> ResidentCellManager::getAllCellIds() const
{
    std::vector<std::string> ids;
    ids.reserve(runs.size() + hostCells.size());
    for (const auto& kv : runs) ids.push_back(kv.first);
    for (const auto& kv : hostCells) if (!runs.count(kv.first)) ids.push_back(kv.first);
    return ids;
}

DataPoints ResidentCellManager::retrieveCell(const std::string& cellId) const
{
    DataPoints out;
    auto hc = hostCells.find(cellId);
    if (hc != hostCells.end()) out = hc->second;                           // (saved from the host first, then what later epochs added)
    auto it = runs.find(cellId);
    if (it == runs.end()) return out;
    int64_t total = 0;
    for (const Run& r : it->second) total += r.count;
    DataPoints fromLog((size_t)total);
    size_t at = 0;
    for (const Run& r : it->second) {
        GpuICPSequence::check(h, icpmi_cell_log_read(h, r.offset, r.count, fromLog.features.data() + 4 * at, nullptr));
        at += (size_t)r.count;
    }
    if (out.getNbPoints() == 0) return fromLog;
    out.concatenate(fromLog);
    return out;
}

void ResidentCellManager::clearAllCells()
{
    runs.clear();
    hostCells.clear();
    GpuICPSequence::check(h, icpmi_cell_log_clear(h));
}

int64_t ResidentCellManager::pointsInLog() const
{
    int64_t n = 0;
    GpuICPSequence::check(h, icpmi_cell_log_read(h, 0, 0, nullptr, &n));
    return n;
}

ShardedMapper::ShardedMapper(const yaml::Node& icpNode, float minDistNewPoint, int surfaceNormalKnn, int device)
    : icp(device), cellManager(new ResidentCellManager(icp.handle())), minDist(minDistNewPoint), normalsKnn(surfaceNormalKnn)
{
    if (minDistNewPoint < 0.f) throw InvalidParameter("ShardedMapper: minDistNewPoint must be >= 0");
    if (icpNode) icp.loadFromYamlNode(icpNode);
    GpuICPSequence::check(icp.handle(), icpmi_cell_log_configure(icp.handle(), Map::CELL_SIZE)); // every epoch bins its merged set on the device, behind the merge
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
    cellsTouched = 0;
    if (mergedN > 0) {
        // Map.cpp:206-229 + RAMCellManager.cpp:13-16 on the device (r6): the merged set is binned into 20 m cells and appended to the
        // handle's cell log there; the host learns {ijk, offset, count} per touched cell and nothing else
        if (cellIjk.size() < 3 * 4096) { cellIjk.resize(3 * 4096); cellOff.resize(4096); cellCnt.resize(4096); }
        int64_t nCells = 0;
        s = icpmi_staged_bin_cells(icp.handle(), Map::CELL_SIZE, cellIjk.data(), cellOff.data(), cellCnt.data(), 4096, &nCells);
        if (s == ICPMI_ERR_UNSUPPORTED) {
            // more cells in one epoch than the device table takes (a 20 m grid: > 4096 cells of NEW points): the reference's loop on the host
            if (merged.size() < 4 * (size_t)mergedN) merged.resize(4 * (size_t)mergedN);
            GpuICPSequence::check(icp.handle(), icpmi_staged_merged_points(icp.handle(), merged.data(), mergedN, &mergedN));
            DataPoints grown((size_t)mergedN);
            std::copy(merged.begin(), merged.begin() + 4 * (size_t)mergedN, grown.features.begin());
            Map::binIntoCells(grown, [&](const std::string& id, DataPoints&& cell) {
                DataPoints old = cellManager->retrieveCell(id);
                ++cellsTouched;
                if (old.getNbPoints() == 0) cellManager->saveCell(id, cell);
                else { old.concatenate(cell); cellManager->saveCell(id, old); }
            });
        } else {
            GpuICPSequence::check(icp.handle(), s);
            for (int64_t r = 0; r < nCells; ++r)
                cellManager->addRun(Map::cellId(cellIjk[3 * r], cellIjk[3 * r + 1], cellIjk[3 * r + 2]), cellOff[(size_t)r], cellCnt[(size_t)r]);
            cellsTouched = nCells;
        }
    }
    if (failure) std::rethrow_exception(failure);
    return corrected;
}

} // namespace nim
