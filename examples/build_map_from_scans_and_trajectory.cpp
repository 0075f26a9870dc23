// build_map_from_scans_and_trajectory -- counterpart of the reference's example harness
// (examples/build_map_from_scans_and_trajectory.cpp:196-239) over the GPU-backed host shell:
//   build_map_from_scans_and_trajectory <data dir> <config.yaml> [<trajectory_out.vtk>]
// <data dir> holds scans/*.vtk and trajectory.csv (a ROS odometry dump: stamp.sec, stamp.nanosec,
// frame ids, position xyz, orientation xyzw, ...).  Scans are paired with trajectory rows in
// LEXICOGRAPHIC file order, as the reference does (its line 191) -- including the resulting
// mis-pairing of cloud_1690309710_85582848.vtk in the bundled data.
#include <dirent.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "../norlab_icp_mapper_amd/host/Mapper.h"

using namespace nim;

struct StampedPose { Mat4 pose; long long ns; };

static std::vector<StampedPose> readTrajectory(const std::string& path)
{
    std::ifstream in(path);
    if (!in) throw std::runtime_error("cannot open " + path);
    std::vector<StampedPose> out;
    std::string line;
    // the columns are found by NAME in the header row (reference examples/build_map_from_scans_and_trajectory.cpp:38-90: a ROS odometry dump --
    // other columns, any order); a header without one of the nine is the reference's error
    if (!std::getline(in, line)) throw std::runtime_error("empty trajectory file " + path);
    const char* wanted[9] = {"header.stamp.sec", "header.stamp.nanosec", "pose.pose.position.x", "pose.pose.position.y", "pose.pose.position.z",
                             "pose.pose.orientation.x", "pose.pose.orientation.y", "pose.pose.orientation.z", "pose.pose.orientation.w"};
    int col[9] = {-1, -1, -1, -1, -1, -1, -1, -1, -1};
    size_t need = 0;
    {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::stringstream hs(line);
        std::string name;
        for (int i = 0; std::getline(hs, name, ','); ++i)
            for (int k = 0; k < 9; ++k) if (name == wanted[k]) { col[k] = i; need = std::max(need, (size_t)i + 1); }
        for (int k = 0; k < 9; ++k) if (col[k] < 0) throw std::runtime_error("Error: Required columns not found in the header.");
    }
    while (std::getline(in, line)) {
        if (line.empty()) continue;
        std::vector<std::string> f;
        std::stringstream ss(line);
        std::string tok;
        while (std::getline(ss, tok, ',')) f.push_back(tok);
        if (f.size() < need) throw std::runtime_error("malformed trajectory row: " + line);
        const long long sec = std::stoll(f[col[0]]), nsec = std::stoll(f[col[1]]);
        const double x = std::stod(f[col[2]]), y = std::stod(f[col[3]]), z = std::stod(f[col[4]]);
        const double qx = std::stod(f[col[5]]), qy = std::stod(f[col[6]]), qz = std::stod(f[col[7]]), qw = std::stod(f[col[8]]);
        Mat4 T = Mat4::identity();
        T(0, 0) = (float)(1 - 2 * (qy * qy + qz * qz)); T(0, 1) = (float)(2 * (qx * qy - qz * qw)); T(0, 2) = (float)(2 * (qx * qz + qy * qw));
        T(1, 0) = (float)(2 * (qx * qy + qz * qw)); T(1, 1) = (float)(1 - 2 * (qx * qx + qz * qz)); T(1, 2) = (float)(2 * (qy * qz - qx * qw));
        T(2, 0) = (float)(2 * (qx * qz - qy * qw)); T(2, 1) = (float)(2 * (qy * qz + qx * qw)); T(2, 2) = (float)(1 - 2 * (qx * qx + qy * qy));
        T(0, 3) = (float)x; T(1, 3) = (float)y; T(2, 3) = (float)z;
        out.push_back(StampedPose{T, sec * 1000000000ll + nsec});
    }
    return out;
}

static std::vector<std::string> listScans(const std::string& dir)
{
    std::vector<std::string> files;
    DIR* d = opendir(dir.c_str());
    if (!d) throw std::runtime_error("cannot open " + dir);
    while (dirent* e = readdir(d)) {
        const std::string name = e->d_name;
        if (name.size() > 4 && name.substr(name.size() - 4) == ".vtk") files.push_back(dir + "/" + name);
    }
    closedir(d);
    std::sort(files.begin(), files.end());
    return files;
}

int main(int argc, char** argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <data dir> <config.yaml> [<trajectory_out.vtk>]\n", argv[0]);
        return 2;
    }
    try {
        const std::string dataDir = argv[1], config = argv[2];
        const auto trajectory = readTrajectory(dataDir + "/trajectory.csv");
        const auto scans = listScans(dataDir + "/scans");
        if (trajectory.size() != scans.size()) throw std::runtime_error("trajectory rows and scan files differ in number");

        // NIM_ONLINE=1: online mode (asynchronous map updates and cell paging on their own threads, Mapper.cpp:274-288,
        // Map.cpp:35-57) -- the reference's example runs offline; the switch exists to exercise those threads
        const char* onlineEnv = std::getenv("NIM_ONLINE");
        const bool online = onlineEnv && std::atoi(onlineEnv) != 0;
        const char* drainEnv = std::getenv("NIM_ONLINE_DRAIN"); // with NIM_ONLINE: wait for the update / paging threads after every scan
        const bool drain = drainEnv && std::atoi(drainEnv) != 0;
        // NIM_2D=1: planar clouds (the constructor's is3D == false, Mapper.h:53): every scan has z == 0, poses move in the plane
        const char* planarEnv = std::getenv("NIM_2D");
        const bool is3D = !(planarEnv && std::atoi(planarEnv) != 0);
        // NIM_TIMING=k (k >= 1): measurement mode (bench.py: chains.config4_replay).  Every scan is loaded and input-filtered BEFORE the clock
        // starts (the ASCII VTK parser is not the path under test); the whole replay then runs k times on a fresh Mapper each, per scan the
        // wall time of processInput and of its two halves is printed, per pass one `replay:` line.  The last pass is the one whose map and
        // trajectory are saved, so the artefacts are those of a plain run.
        const char* timingEnv = std::getenv("NIM_TIMING");
        const int passes = timingEnv ? std::max(1, std::atoi(timingEnv)) : 1;
        const bool timing = timingEnv != nullptr;
        std::vector<DataPoints> preloaded;
        if (timing) {
            Mapper filt(config, is3D, false, true, false);
            const auto tl = std::chrono::steady_clock::now();
            for (const auto& f : scans) preloaded.push_back(DataPoints::load(f));
            const double loadMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tl).count();
            const auto tf = std::chrono::steady_clock::now();
            for (auto& c : preloaded) filt.applyInputFilters(c);
            const double filtMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tf).count();
            std::printf("preload: %zu scans, load %.1f ms (ASCII VTK), input filters %.2f ms\n", scans.size(), loadMs, filtMs);
        }
        std::unique_ptr<Mapper> mapperPtr;
        double secs = 0.0;
        for (int pass = 0; pass < passes; ++pass) {
        mapperPtr.reset(); // (one GPU context at a time)
        mapperPtr.reset(new Mapper(config, is3D, /*isOnline*/ online, /*isMapping*/ true, /*saveMapCellsOnHardDrive*/ false));
        Mapper& mapper = *mapperPtr;
        // NIM_SETMAP_AT=i: after scan i (0-based) the whole map is taken out with getMap() and handed back with setMap() -- the
        // reference's checkpoint / resume path (Mapper.cpp:295-301 -> Map::setGlobalPointCloud, Map.cpp:575-588: the next updatePose
        // pages the cloud into cells again; the trajectory restarts)
        const char* setMapEnv = std::getenv("NIM_SETMAP_AT");
        const long setMapAt = setMapEnv ? std::atol(setMapEnv) : -1;
        const auto t0 = std::chrono::steady_clock::now();
        double procMs = 0.0, regMs = 0.0, updMs = 0.0;
        int totalIters = 0;
        for (size_t i = 0; i < scans.size(); ++i) {
            const TimePoint stamp{std::chrono::nanoseconds(trajectory[i].ns)};
            DataPoints loaded;
            if (!timing) { loaded = DataPoints::load(scans[i]); mapper.applyInputFilters(loaded); }
            const DataPoints& cloud = timing ? preloaded[i] : loaded;
            const auto tp = std::chrono::steady_clock::now();
            mapper.processInput(cloud, trajectory[i].pose, stamp);
            const double pm = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp).count();
            procMs += pm; regMs += mapper.lastRegisterMs(); updMs += mapper.lastMapUpdateMs(); totalIters += mapper.lastIcpStats().iterations;
            if (drain) mapper.waitForPendingWork();
            if ((long)i == setMapAt) {
                mapper.waitForPendingWork();
                const DataPoints whole = mapper.getMap();
                mapper.setMap(whole);
                std::printf("setMap: %zu points handed back after scan %zu\n", whole.getNbPoints(), i + 1);
            }
            const Mat4 p = mapper.getPose();
            std::printf("scan %zu/%zu  %zu pts  pose %.4f %.4f %.4f  iterations %d  overlap %.3f  local map %zu  map_version %ld  update %d\n", i + 1,
                        scans.size(), cloud.getNbPoints(), p(0, 3), p(1, 3), p(2, 3), mapper.lastIcpStats().iterations,
                        mapper.lastIcpStats().weighted_point_used_ratio, mapper.localMapSize(), mapper.lastRegistrationMapVersion(),
                        mapper.lastScanStartedMapUpdate() ? 1 : 0);
            if (timing)
                std::printf("timing: pass %d scan %zu points %zu process_ms %.3f register_ms %.3f update_ms %.3f iterations %d map %zu\n", pass, i + 1,
                            cloud.getNbPoints(), pm, mapper.lastRegisterMs(), mapper.lastMapUpdateMs(), mapper.lastIcpStats().iterations, mapper.localMapSize());
        }
        secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (timing)
            std::printf("replay: pass %d scans %zu process_ms %.3f register_ms %.3f update_ms %.3f iterations %d scans_per_s %.2f\n", pass, scans.size(), procMs,
                        regMs, updMs, totalIters, (double)scans.size() / (procMs * 1e-3));
        }
        Mapper& mapper = *mapperPtr;
        DataPoints map = mapper.getMap();
        const char* binaryEnv = std::getenv("NIM_BINARY_VTK"); // BINARY legacy VTK (big-endian), as libpointmatcher's `binary` save option
        map.save(dataDir + "/map.vtk", binaryEnv && std::atoi(binaryEnv) != 0);
        if (argc > 3) mapper.getTrajectory().save(argv[3]);
        std::printf("map: %zu points, %zu scans in %.3f s -> %s/map.vtk\n", map.getNbPoints(), scans.size(), secs, dataDir.c_str());
        std::printf("resident map updates: %ld\n", mapper.residentMapUpdates());
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
