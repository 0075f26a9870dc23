// Map.cpp -- see Map.h.  Behavioural restatement of norlab_icp_mapper/Map.cpp over the GPU-backed
// ICP object; cell paging is host logic over value-type clouds exactly like the reference.
#include "Map.h"

#include <functional>

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <stdexcept>
#include <cstdio>

namespace nim {

Map::Map(bool is3D_, bool isOnline_, bool saveCellsOnHardDrive, GpuICPSequence& icp_, std::mutex& icpMapLock_)
    : is3D(is3D_), isOnline(isOnline_), icp(icp_), icpMapLock(icpMapLock_), transformation(icp_.handle())
{
    // is3D == false: clouds are planar (z == 0 in the 4 x N layout this host keeps for both cases); cells are binned over
    // two axes (below), the registration core switches to its planar minimisers and normals (icpmi_config::is_2d)
    if (!is3D) icp.setPlanar(true);
    if (saveCellsOnHardDrive) throw InvalidParameter("HardDriveCellManager is out of scope (DESIGN.md section 8); use RAM cells");
    cellManager.reset(new RAMCellManager());
    if (isOnline) updateThread = std::thread(&Map::updateThreadFunction, this);
}

Map::~Map()
{
    if (isOnline) {
        updateThreadLooping.store(false);
        if (updateThread.joinable()) updateThread.join();
    }
}

void Map::updateThreadFunction()
{
    while (updateThreadLooping.load()) {
        bool have = false;
        Update u{};
        {
            std::lock_guard<std::mutex> g(updateListLock);
            if (!updateList.empty()) { u = updateList.front(); updateList.pop_front(); have = true; }
        }
        if (have) { applyUpdate(u); updatesInFlight.fetch_sub(1); }
        else std::this_thread::sleep_for(std::chrono::milliseconds(10));
    }
}

void Map::waitForPaging()
{
    while (isOnline && updatesInFlight.load() > 0) std::this_thread::sleep_for(std::chrono::milliseconds(1));
}

void Map::scheduleUpdate(const Update& u)
{
    if (isOnline) {
        std::lock_guard<std::mutex> g(updateListLock);
        updatesInFlight.fetch_add(1);
        updateList.push_back(u);
    } else applyUpdate(u);
}

void Map::loadCells(Box box)
{
    if (!is3D) box.lo[2] = box.hi[2] = 0;
    DataPoints chunk;
    for (int i = box.lo[0]; i <= box.hi[0]; ++i)
        for (int j = box.lo[1]; j <= box.hi[1]; ++j)
            for (int k = box.lo[2]; k <= box.hi[2]; ++k) {
                DataPoints cell;
                {
                    std::lock_guard<std::mutex> g(cellManagerLock);
                    cell = cellManager->retrieveCell(cellId(i, j, k));
                }
                if (cell.getNbPoints() > 0) chunk.concatenate(cell);
            }
    std::lock_guard<std::mutex> g(localPointCloudLock);
    if (chunk.getNbPoints() > 0) {
        syncLocalFromDevice();
        localPointCloud.concatenate(chunk);
        {
            std::lock_guard<std::mutex> gi(icpMapLock);
            icp.setMap(localPointCloud);
        }
        localPointCloudEmpty.store(false);
        newLocalPointCloudAvailable = true;
    }
    for (int i = box.lo[0]; i <= box.hi[0]; ++i)
        for (int j = box.lo[1]; j <= box.hi[1]; ++j)
            for (int k = box.lo[2]; k <= box.hi[2]; ++k) loadedCellIds.insert(cellId(i, j, k));
}

void Map::unloadCells(Box box)
{
    if (!is3D) box.lo[2] = box.hi[2] = 0;
    float start[3], end[3];
    for (int a = 0; a < 3; ++a) {
        start[a] = box.lo[a] * CELL_SIZE;                 // toInferiorWorldCoordinate
        end[a] = ((float)box.hi[a] + 1.0f) * CELL_SIZE;   // toSuperiorWorldCoordinate (no int overflow at INT_MAX - 1)
    }
    DataPoints oldChunk;
    {
        std::lock_guard<std::mutex> g(localPointCloudLock);
        syncLocalFromDevice();
        const size_t n = localPointCloud.getNbPoints();
        std::vector<uint8_t> leaves(n), stays(n);
        for (size_t i = 0; i < n; ++i) {
            const float* p = localPointCloud.col(i);
            bool in = true;
            for (int a = 0; a < 3; ++a) in &= p[a] >= start[a] && p[a] < end[a];
            leaves[i] = in; stays[i] = !in;
        }
        oldChunk = localPointCloud;
        oldChunk.keepOnly(leaves);
        localPointCloud.keepOnly(stays);
        {
            std::lock_guard<std::mutex> gi(icpMapLock);
            icp.setMap(localPointCloud); // an empty cloud is rejected and leaves the previous map in place
        }
        // forget the ids inside the box (iterate what is loaded: the first update's box is all of space)
        for (auto it = loadedCellIds.begin(); it != loadedCellIds.end();) {
            int c[3] = {0, 0, 0};
            std::sscanf(it->c_str(), "%d_%d_%d", &c[0], &c[1], &c[2]);
            bool in = true;
            for (int a = 0; a < 3; ++a) in &= c[a] >= box.lo[a] && c[a] <= box.hi[a];
            it = in ? loadedCellIds.erase(it) : std::next(it);
        }
        localPointCloudEmpty.store(localPointCloud.getNbPoints() == 0);
        newLocalPointCloudAvailable = true;
    }
    // bin what left into 20 m cells; saveCell overwrites (RAMCellManager.cpp:13-16).  One pass: the rows of every cell are
    // collected first, then each cell is built from its own rows only (O(points), like the reference's per-point appends).
    binIntoCells(oldChunk, [&](const std::string& id, DataPoints&& cell) {
        std::lock_guard<std::mutex> g(cellManagerLock);
        cellManager->saveCell(id, cell);
    });
}

void Map::binIntoCells(const DataPoints& cloud, const std::function<void(const std::string&, DataPoints&&)>& sink)
{
    std::unordered_map<std::string, std::vector<size_t>> rows;
    const size_t n = cloud.getNbPoints();
    for (size_t i = 0; i < n; ++i) {
        const float* p = cloud.col(i);
        rows[cellId(toGridCoordinate(p[0]), toGridCoordinate(p[1]), toGridCoordinate(p[2]))].push_back(i);
    }
    for (auto& kv : rows) {
        DataPoints cell = cloud.createSimilarEmpty(kv.second.size());
        for (size_t i : kv.second) cell.appendColFrom(cloud, i);
        sink(kv.first, std::move(cell));
    }
}

void Map::updatePose(const Mat4& pose)
{
    const int axes = is3D ? 3 : 2;
    if (firstPoseUpdate.load()) {
        for (int a = 0; a < axes; ++a) {
            inferiorLast[a] = toInferiorGridCoordinate(pose(a, 3), sensorMaxRange);
            superiorLast[a] = toSuperiorGridCoordinate(pose(a, 3), sensorMaxRange);
        }
        {
            std::lock_guard<std::mutex> g(cellManagerLock);
            cellManager->clearAllCells();
        }
        {
            std::lock_guard<std::mutex> g(localPointCloudLock);
            loadedCellIds.clear();
        }
        // page the whole local cloud out (used by setGlobalPointCloud to re-page a loaded map) ...
        Box all;
        for (int a = 0; a < 3; ++a) { all.lo[a] = INT_MIN; all.hi[a] = INT_MAX - 1; }
        unloadCells(all);
        // ... and the window (+ buffer) back in
        Box win;
        for (int a = 0; a < 3; ++a) { win.lo[a] = inferiorLast[a] - BUFFER_SIZE; win.hi[a] = superiorLast[a] + BUFFER_SIZE; }
        loadCells(win);
        firstPoseUpdate.store(false);
        return;
    }
    // sliding window with a hysteresis of two cells per edge; a slab spans the current window (+ buffer)
    // on the other axes
    auto slab = [&](int axis, int lo, int hi, bool load) {
        Update u;
        for (int a = 0; a < 3; ++a) { u.box.lo[a] = inferiorLast[a] - BUFFER_SIZE; u.box.hi[a] = superiorLast[a] + BUFFER_SIZE; }
        u.box.lo[axis] = lo; u.box.hi[axis] = hi; u.load = load;
        scheduleUpdate(u);
    };
    for (int a = 0; a < axes; ++a) {
        const int inf = toInferiorGridCoordinate(pose(a, 3), sensorMaxRange);
        if (std::abs(inf - inferiorLast[a]) >= 2) {
            if (inf < inferiorLast[a]) {             // window grows at its low edge: bring cells in
                const int nb = inferiorLast[a] - inf;
                slab(a, inf - BUFFER_SIZE, inf - BUFFER_SIZE + nb - 1, true);
            } else {                                  // low edge retreats: page cells out
                const int nb = inf - inferiorLast[a];
                slab(a, inferiorLast[a] - BUFFER_SIZE, inferiorLast[a] - BUFFER_SIZE + nb - 1, false);
            }
            inferiorLast[a] = inf;
        }
        const int sup = toSuperiorGridCoordinate(pose(a, 3), sensorMaxRange);
        if (std::abs(sup - superiorLast[a]) >= 2) {
            if (sup < superiorLast[a]) {             // high edge retreats
                const int nb = superiorLast[a] - sup;
                slab(a, superiorLast[a] + BUFFER_SIZE - nb + 1, superiorLast[a] + BUFFER_SIZE, false);
            } else {                                  // window grows at its high edge
                const int nb = sup - superiorLast[a];
                slab(a, sup + BUFFER_SIZE - nb + 1, sup + BUFFER_SIZE, true);
            }
            superiorLast[a] = sup;
        }
    }
}

DataPoints Map::getLocalPointCloud()
{
    std::lock_guard<std::mutex> g(localPointCloudLock);
    syncLocalFromDevice();
    return localPointCloud;
}

void Map::syncLocalFromDevice()
{
    if (!deviceAhead) return;
    // features, `normals` and the tracked scalar descriptor live on the device; every other descriptor was kept current on the host
    DataPoints dev;
    std::vector<float> scalar;
    {
        std::lock_guard<std::mutex> gi(icpMapLock);
        dev = icp.downloadMap();
        if (!residentScalar.empty()) scalar = icp.downloadMapScalar();
    }
    localPointCloud.features = std::move(dev.features);
    localPointCloud.removeDescriptor("normals");
    if (residentNormals && dev.descriptorExists("normals")) localPointCloud.addDescriptor("normals", 3, std::move(dev.getDescriptorByName("normals").data));
    if (!residentScalar.empty()) localPointCloud.addDescriptor(residentScalar, 1, std::move(scalar));
    deviceAhead = false;
}

// The resident chain serves planar maps (is3D == false) because a planar pose maps z = 0 to z = 0 exactly and the kernels then see what the
// reference's 2-D branch computes.  That argument needs every input point at z == 0 and a pose whose row / column 2 is exactly identity:
// checked here, anything else takes the host-pointer path.
static bool planarPose(const Mat4& T)
{
    return T(2, 0) == 0.f && T(2, 1) == 0.f && T(0, 2) == 0.f && T(1, 2) == 0.f && T(2, 2) == 1.f && T(2, 3) == 0.f;
}
static bool planarCloud(const DataPoints& c)
{
    const size_t n = c.getNbPoints();
    for (size_t i = 0; i < n; ++i) if (c.features[4 * i + 2] != 0.f) return false;
    return true;
}

bool Map::residentPlan(const DataPoints& input, const DataPointsFilters& postFilters, ResidentProgram& prog) const
{
    static const bool enabled = [] { const char* e = std::getenv("NIM_RESIDENT_MAP_UPDATE"); return !e || std::atoi(e) != 0; }();
    // (a GenericDescriptorOutlierFilter reads a descriptor of the host cloud handed to setMap: host path)
    // (planar maps too, r3: the clouds keep z == 0 in the 4 x N layout and a planar pose maps z = 0 to z = 0 exactly, so the chain's
    //  kernels see what the host-pointer operators see -- spherical coordinates with elevation asin(0 / r) = 0, the reference's
    //  is3D == false branch of DynamicPointsMapperModule.cpp:156-172; normals through the planar eigen-solve, icpmi_config::is_2d)
    if (!enabled || mapperModuleVec.empty() || icp.hasReferenceFilters() || !icp.genericDescriptorName().empty()) return false;
    if (!is3D && !planarCloud(input)) return false;
    prog = ResidentProgram{};
    auto adopt = [&](bool ok, const icpmi_map_op& op, const std::string& name) {
        if (!ok) return false;
        if (!name.empty()) {
            if (!prog.scalarName.empty() && prog.scalarName != name) return false; // the device tracks one scalar descriptor
            prog.scalarName = name;
        }
        prog.ops.push_back(op);
        return true;
    };
    int dynamicAt = -1;
    for (const auto& module : mapperModuleVec) {
        icpmi_map_op op{}; std::string name;
        if (!adopt(module->residentOp(op, name), op, name)) return false;
        if (op.type == ICPMI_MOP_DYNAMIC_POINTS && dynamicAt < 0) dynamicAt = (int)prog.ops.size() - 1;
    }
    prog.nModules = (int)prog.ops.size();
    for (const auto& filter : postFilters.filters) {
        icpmi_map_op op{}; std::string name;
        if (!adopt(filter->residentOp(op, name), op, name)) return false;
        prog.computesNormals |= op.type == ICPMI_MOP_SURFACE_NORMALS;
    }
    // a point-to-plane chain needs normals on every map point: only with the SurfaceNormal post filter
    if (icp.config().minimizer == ICPMI_MIN_POINT_TO_PLANE && !prog.computesNormals) return false;
    const bool first = isLocalPointCloudEmpty();
    if (first && icp.hasMap()) return false; // the ICP map is not the local cloud (cells just unloaded)
    const bool inputNormals = input.descriptorExists("normals") && input.getDescriptorByName("normals").span == 3;
    const bool mapNormals = !first && (deviceAhead ? residentNormals : localPointCloud.descriptorExists("normals"));
    // without the post filter the map's normals must come with the input (or not exist at all): DataPoints::concatenate's rule
    if (!prog.computesNormals && !first && mapNormals != inputNormals) return false;
    // DynamicPointsMapperModule.cpp:38-41 throws on a map without normals: let the host path raise the reference's error
    if (dynamicAt >= 0 && ((!first && !mapNormals) || (first && dynamicAt > 0 && !inputNormals))) return false;
    if (!prog.scalarName.empty()) {
        if (!input.descriptorExists(prog.scalarName) || input.getDescriptorByName(prog.scalarName).span != 1) return false;
        if (!first) {
            if (deviceAhead ? residentScalar != prog.scalarName
                            : (!localPointCloud.descriptorExists(prog.scalarName) || localPointCloud.getDescriptorByName(prog.scalarName).span != 1))
                return false;
        }
    }
    return true;
}

bool Map::hostDescriptorsFollow(const DataPoints& input, const ResidentProgram& prog, bool first) const
{
    const std::vector<Descriptor>& fields = first ? input.descriptors : localPointCloud.descriptors;
    for (const Descriptor& d : fields) {
        if (d.name == "normals" || d.name == prog.scalarName) continue;
        if (input.descriptorExists(d.name) && input.getDescriptorByName(d.name).span == d.span) return true;
    }
    return false;
}

void Map::prepareResidentScalar(const ResidentProgram& prog, bool first)
{
    // the host copy is the authority whenever the device does not run ahead: hand it the tracked scalar (icp.setMap carries
    // features and normals only)
    if (first || prog.scalarName.empty() || deviceAhead) return;
    const Descriptor& d = localPointCloud.getDescriptorByName(prog.scalarName);
    std::vector<float> row(localPointCloud.getNbPoints());
    for (size_t i = 0; i < row.size(); ++i) row[i] = d.data[(size_t)d.span * i];
    std::lock_guard<std::mutex> gi(icpMapLock);
    icp.uploadMapScalar(row);
}

void Map::adoptResidentResult(const DataPoints& input, const ResidentProgram& prog, const std::vector<int32_t>& src, int64_t prefix,
                              int64_t mapSize, bool first)
{
    // `normals` and the tracked scalar live on the device; every other descriptor follows the provenance vector with
    // DataPoints::concatenate's rule (only fields both clouds have survive; the first scan brings its own set).  Features,
    // normals and the scalar of localPointCloud are stale until syncLocalFromDevice().
    const size_t m0 = first ? 0 : (size_t)residentCount;
    const bool hadNormals = !first && (deviceAhead ? residentNormals : localPointCloud.descriptorExists("normals"));
    const bool inputNormals = input.descriptorExists("normals") && input.getDescriptorByName("normals").span == 3;
    std::vector<Descriptor> next;
    const std::vector<Descriptor>& fields = first ? input.descriptors : localPointCloud.descriptors;
    for (const Descriptor& old : fields) {
        if (old.name == "normals" || old.name == prog.scalarName) continue;
        if (!input.descriptorExists(old.name) || input.getDescriptorByName(old.name).span != old.span) continue;
        const Descriptor& in = input.getDescriptorByName(old.name);
        const size_t span = (size_t)old.span;
        Descriptor d; d.name = old.name; d.span = old.span;
        d.data.resize(span * (size_t)mapSize);
        if (!first && prefix > 0) std::copy(old.data.begin(), old.data.begin() + span * (size_t)prefix, d.data.begin());
        for (size_t j = first ? 0 : (size_t)prefix; j < (size_t)mapSize; ++j) {
            const size_t from = (size_t)src[j];
            const float* row = from < m0 ? &old.data[span * from] : &in.data[span * (from - m0)];
            std::copy(row, row + span, d.data.begin() + span * j);
        }
        next.push_back(std::move(d));
    }
    localPointCloud.descriptors = std::move(next);
    residentNormals = prog.computesNormals || (first ? inputNormals : (hadNormals && inputNormals));
    residentScalar = prog.scalarName;
    residentCount = mapSize;
    deviceAhead = true;
    ++residentUpdates;
    localPointCloudEmpty.store(mapSize == 0);
    newLocalPointCloudAvailable = true;
}

// A chain that fails half way (out of device memory, a HIP error) has dropped the resident arrays (ops_map_update_chain): the host
// copy is stale by every update since the last sync and cannot be refreshed any more.  Leave a CONSISTENT empty local cloud -- the
// next scan creates the map anew, the registration index of the old map stays in the ICP object -- and let the error travel on.
void Map::dropLocalCloudAfterFailedUpdate()
{
    if (icp.residentMapSize() > 0) return; // validation errors leave the device state alone
    localPointCloud = DataPoints();
    deviceAhead = false; residentCount = 0; residentNormals = false; residentScalar.clear();
    localPointCloudEmpty.store(true);
    newLocalPointCloudAvailable = true;
}

bool Map::tryResidentUpdate(const DataPoints& input, const Mat4& pose, const DataPointsFilters& postFilters)
{
    ResidentProgram prog;
    if (!is3D && !planarPose(pose)) return false;
    if (!residentPlan(input, postFilters, prog)) return false;
    const bool first = isLocalPointCloudEmpty();
    if (!deviceAhead) residentCount = (int64_t)localPointCloud.getNbPoints();
    prepareResidentScalar(prog, first);
    std::vector<int32_t> src;
    int64_t prefix = 0, m = 0;
    {
        std::lock_guard<std::mutex> gi(icpMapLock);
        try {
            icp.mapUpdateChain(&input, Mat4::identity(), prog.scalarName, input, pose, prog.ops, prog.nModules, src, prefix, m,
                               hostDescriptorsFollow(input, prog, first));
        ++icpMapVersions;
        } catch (...) { dropLocalCloudAfterFailedUpdate(); throw; }
    }
    adoptResidentResult(input, prog, src, prefix, m, first);
    return true;
}

bool Map::canStageScan(const DataPoints& input, const DataPointsFilters& postFilters, const Mat4* pose)
{
    if (!is3D && pose && !planarPose(*pose)) return false;
    // descriptors that rotate with the cloud would have to be transformed along: host path
    if (input.descriptorExists("normals") || input.descriptorExists("observationDirections")) return false;
    std::lock_guard<std::mutex> g(localPointCloudLock);
    ResidentProgram prog;
    return residentPlan(input, postFilters, prog);
}

void Map::updateLocalPointCloudStaged(const DataPoints& inputDescriptors, const Mat4& correction, const Mat4& pose, const DataPointsFilters& postFilters)
{
    std::lock_guard<std::mutex> g(localPointCloudLock);
    ResidentProgram prog;
    if (!residentPlan(inputDescriptors, postFilters, prog)) throw std::logic_error("staged map update is not available for this configuration");
    const bool first = isLocalPointCloudEmpty();
    if (!deviceAhead) residentCount = (int64_t)localPointCloud.getNbPoints();
    prepareResidentScalar(prog, first);
    std::vector<int32_t> src;
    int64_t prefix = 0, m = 0;
    {
        std::lock_guard<std::mutex> gi(icpMapLock);
        try {
            icp.mapUpdateChain(nullptr, correction, prog.scalarName, inputDescriptors, pose, prog.ops, prog.nModules, src, prefix, m,
                               hostDescriptorsFollow(inputDescriptors, prog, first));
        ++icpMapVersions;
        } catch (...) { dropLocalCloudAfterFailedUpdate(); throw; }
    }
    adoptResidentResult(inputDescriptors, prog, src, prefix, m, first);
}

void Map::updateLocalPointCloud(DataPoints input, Mat4 pose, DataPointsFilters postFilters)
{
    std::lock_guard<std::mutex> g(localPointCloudLock);
    if (mapperModuleVec.empty()) throw InvalidParameter("no mapper module configured");
    if (tryResidentUpdate(input, pose, postFilters)) return;
    syncLocalFromDevice();
    if (isLocalPointCloudEmpty()) {
        // the first module creates the map, the others update it with the same scan
        auto it = mapperModuleVec.begin();
        localPointCloud = (*it)->createMap(input, pose);
        for (++it; it != mapperModuleVec.end(); ++it) (*it)->inPlaceUpdateMap(input, localPointCloud, pose);
    } else {
        for (const auto& module : mapperModuleVec) module->inPlaceUpdateMap(input, localPointCloud, pose);
    }
    // post filters run in the sensor frame
    DataPoints inSensorFrame = transformation.compute(localPointCloud, pose.inverse());
    postFilters.apply(inSensorFrame);
    localPointCloud = transformation.compute(inSensorFrame, pose);
    {
        std::lock_guard<std::mutex> gi(icpMapLock);
        icp.setMap(localPointCloud);
        ++icpMapVersions;
    }
    localPointCloudEmpty.store(localPointCloud.getNbPoints() == 0);
    newLocalPointCloudAvailable = true;
}

bool Map::getNewLocalPointCloud(DataPoints& out)
{
    std::lock_guard<std::mutex> g(localPointCloudLock);
    if (!newLocalPointCloudAvailable) return false;
    syncLocalFromDevice();
    out = localPointCloud;
    newLocalPointCloudAvailable = false;
    return true;
}

DataPoints Map::getGlobalPointCloud()
{
    DataPoints global;
    std::unordered_set<std::string> loaded;
    {
        std::lock_guard<std::mutex> g(localPointCloudLock);
        syncLocalFromDevice();
        global = localPointCloud;
        loaded = loadedCellIds;
    }
    std::vector<std::string> saved;
    {
        std::lock_guard<std::mutex> g(cellManagerLock);
        saved = cellManager->getAllCellIds();
    }
    for (const auto& id : saved) {
        if (loaded.count(id)) continue;
        DataPoints cell;
        {
            std::lock_guard<std::mutex> g(cellManagerLock);
            cell = cellManager->retrieveCell(id);
        }
        global.concatenate(cell);
    }
    return global;
}

void Map::setGlobalPointCloud(const DataPoints& cloud)
{
    std::lock_guard<std::mutex> g(localPointCloudLock);
    localPointCloud = cloud;
    deviceAhead = false;
    {
        std::lock_guard<std::mutex> gi(icpMapLock);
        icp.setMap(localPointCloud);
    }
    localPointCloudEmpty.store(localPointCloud.getNbPoints() == 0);
    firstPoseUpdate.store(true); // the next updatePose re-pages the cloud into cells
}

} // namespace nim
