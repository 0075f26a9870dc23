// sharded_mapping -- BASELINE config 5 in C++: one process per GPU, independent scans of the examples/ trajectory sharded over
// the ranks, the accepted points all-gathered over RCCL inside libicpmi.so and merged into every rank's map replica and
// CellManager (norlab_icp_mapper_amd/host/ShardedMapper.h).
//
//   RANK=r WORLD_SIZE=N LOCAL_RANK=g ICPMI_COMM_FILE=/tmp/id  sharded_mapping <data dir> <config.yaml> [minDistNewPoint] [normalsKnn]
//
// <data dir> as the reference's example: scans/*.vtk + trajectory.csv, paired in lexicographic order (examples/
// build_map_from_scans_and_trajectory.cpp:191).  The first scan seeds the map on every rank; afterwards epoch e hands scan
// 1 + e N + r to rank r.  Rank 0 writes the communicator id to ICPMI_COMM_FILE, the others wait for it (any launcher works:
// the id is the only thing the ranks share besides the file system).  With WORLD_SIZE unset it is one rank and no RCCL.
#include <dirent.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../norlab_icp_mapper_amd/host/Mapper.h"
#include "../norlab_icp_mapper_amd/host/ShardedMapper.h"

using namespace nim;

static std::vector<Mat4> readPoses(const std::string& path)
{
    std::ifstream in(path);
    if (!in) throw std::runtime_error("cannot open " + path);
    std::vector<Mat4> out;
    std::string line;
    // columns by NAME, as the reference's harness finds them (examples/build_map_from_scans_and_trajectory.cpp:38-90)
    if (!std::getline(in, line)) throw std::runtime_error("empty trajectory file " + path);
    const char* wanted[7] = {"pose.pose.position.x", "pose.pose.position.y", "pose.pose.position.z",
                             "pose.pose.orientation.x", "pose.pose.orientation.y", "pose.pose.orientation.z", "pose.pose.orientation.w"};
    int col[7] = {-1, -1, -1, -1, -1, -1, -1};
    size_t need = 0;
    {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::stringstream hs(line);
        std::string name;
        for (int i = 0; std::getline(hs, name, ','); ++i)
            for (int k = 0; k < 7; ++k) if (name == wanted[k]) { col[k] = i; if ((size_t)i + 1 > need) need = (size_t)i + 1; }
        for (int k = 0; k < 7; ++k) if (col[k] < 0) throw std::runtime_error("Error: Required columns not found in the header.");
    }
    while (std::getline(in, line)) {
        if (line.empty()) continue;
        std::vector<std::string> f;
        std::stringstream ss(line);
        std::string tok;
        while (std::getline(ss, tok, ',')) f.push_back(tok);
        if (f.size() < need) throw std::runtime_error("malformed trajectory row");
        const double x = std::stod(f[col[0]]), y = std::stod(f[col[1]]), z = std::stod(f[col[2]]);
        const double qx = std::stod(f[col[3]]), qy = std::stod(f[col[4]]), qz = std::stod(f[col[5]]), qw = std::stod(f[col[6]]);
        Mat4 T = Mat4::identity();
        T(0, 0) = (float)(1 - 2 * (qy * qy + qz * qz)); T(0, 1) = (float)(2 * (qx * qy - qz * qw)); T(0, 2) = (float)(2 * (qx * qz + qy * qw));
        T(1, 0) = (float)(2 * (qx * qy + qz * qw)); T(1, 1) = (float)(1 - 2 * (qx * qx + qz * qz)); T(1, 2) = (float)(2 * (qy * qz - qx * qw));
        T(2, 0) = (float)(2 * (qx * qz - qy * qw)); T(2, 1) = (float)(2 * (qy * qz + qx * qw)); T(2, 2) = (float)(1 - 2 * (qx * qx + qy * qy));
        T(0, 3) = (float)x; T(1, 3) = (float)y; T(2, 3) = (float)z;
        out.push_back(T);
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

static int envInt(const char* k, int def) { const char* v = std::getenv(k); return v ? std::atoi(v) : def; }

int main(int argc, char** argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: %s <data dir> <config.yaml> [minDistNewPoint] [normalsKnn]\n", argv[0]); return 2; }
    try {
        const int rank = envInt("RANK", 0), world = envInt("WORLD_SIZE", 1), local = envInt("LOCAL_RANK", rank);
        const std::string dataDir = argv[1];
        const float minDist = argc > 3 ? (float)std::atof(argv[3]) : 0.15f;
        const int knn = argc > 4 ? std::atoi(argv[4]) : 10;
        std::ifstream cfgFile(argv[2]);
        if (!cfgFile) throw std::runtime_error(std::string("cannot open ") + argv[2]);
        std::stringstream cfgText; cfgText << cfgFile.rdbuf();
        const yaml::Node cfg = yaml::Load(cfgText.str());
        ShardedMapper mapper(cfg["icp"], minDist, knn, local);
        if (world > 1) {
            const char* file = std::getenv("ICPMI_COMM_FILE");
            if (!file) throw std::runtime_error("ICPMI_COMM_FILE must name a file all ranks can reach");
            icpmi_comm_id id;
            if (rank == 0) {
                id = ShardedMapper::createCommunicatorId();
                const std::string tmp = std::string(file) + ".tmp";
                { std::ofstream o(tmp, std::ios::binary); o.write(id.bytes, sizeof id.bytes); }
                std::rename(tmp.c_str(), file);
            } else {
                for (int tries = 0;; ++tries) {
                    std::ifstream i(file, std::ios::binary);
                    if (i && i.read(id.bytes, sizeof id.bytes)) break;
                    if (tries > 6000) throw std::runtime_error("no communicator id in ICPMI_COMM_FILE after 60 s");
                    usleep(10000);
                }
            }
            mapper.initCommunicator(id, world, rank);
        } else if (std::getenv("ICPMI_COMM_LOOPBACK")) {
            // test hook (csrc/comm.hip): R simulated ranks on this one GPU exercise the rank-ordered merge without RCCL
            icpmi_comm_id id{};
            mapper.initCommunicator(id, 1, 0);
        }
        const auto poses = readPoses(dataDir + "/trajectory.csv");
        const auto scans = listScans(dataDir + "/scans");
        if (poses.size() != scans.size() || scans.empty()) throw std::runtime_error("trajectory rows and scan files differ in number");
        // the mapper's own input filters (radius + the `input:` chain) through a Mapper-less path: DataPointsFilters of the configuration
        DataPointsFilters inputFilters(cfg["input"], nullptr);
        auto load = [&](size_t i) { DataPoints c = DataPoints::load(scans[i]); inputFilters.apply(c); return c; };
        {   // the first scan seeds the map, identically on every rank
            DataPoints first = load(0);
            DataPoints inMap = first;
            for (size_t i = 0; i < first.getNbPoints(); ++i) {
                const float* p = first.col(i); float* q = inMap.col(i);
                for (int r = 0; r < 3; ++r) q[r] = poses[0](r, 0) * p[0] + poses[0](r, 1) * p[1] + poses[0](r, 2) * p[2] + poses[0](r, 3);
            }
            mapper.setMap(inMap);
        }
        const auto t0 = std::chrono::steady_clock::now();
        size_t done = 0, failed = 0;
        // every rank runs the SAME number of epochs: the epoch is a collective.  A rank whose share of the scans is one short hands
        // in an empty cloud in the last epoch; a rank whose registration throws still went through the exchange (ShardedMapper::
        // processScan) and carries on with its next scan.
        const size_t epochs = (scans.size() - 1 + (size_t)world - 1) / (size_t)world;
        for (size_t e = 0; e < epochs; ++e) {
            const size_t i = 1 + e * (size_t)world + (size_t)rank;
            const bool have = i < scans.size();
            const DataPoints cloud = have ? load(i) : DataPoints(0);
            try {
                const Mat4 pose = mapper.processScan(cloud, have ? poses[i] : Mat4::identity());
                if (!have) continue;
                ++done;
                std::printf("rank %d epoch %zu scan %zu: %zu pts, pose %.4f %.4f %.4f, iterations %d, %ld accepted here, %ld appended by all ranks, map %ld\n",
                            rank, e, i, cloud.getNbPoints(), pose(0, 3), pose(1, 3), pose(2, 3), mapper.lastIcpStats().iterations,
                            (long)mapper.lastAcceptedLocal(), (long)mapper.lastAppended(), (long)mapper.mapSize());
            } catch (const ConvergenceError& ex) {
                ++failed;
                std::printf("rank %d epoch %zu scan %zu: registration failed (%s); the epoch went through, map %ld\n", rank, e, i, ex.what(), (long)mapper.mapSize());
            }
        }
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        size_t cellPts = 0;
        for (const auto& id : mapper.cells().getAllCellIds()) cellPts += mapper.cells().retrieveCell(id).getNbPoints();
        std::printf("rank %d: %zu scans in %.3f s, map %ld points, %zu cells holding %zu merged points\n", rank, done, secs, (long)mapper.mapSize(),
                    mapper.cells().getAllCellIds().size(), cellPts);
        {   // the cells (binned on the device, epoch after epoch) against the reference's loop over what the epochs appended: the
            // tail of the resident map in append order (Map.cpp:206-229 on the host)
            const DataPoints whole = mapper.getMap();
            const size_t tail = cellPts <= whole.getNbPoints() ? cellPts : whole.getNbPoints();
            DataPoints appendedPts(tail);
            std::copy(whole.features.begin() + 4 * (whole.getNbPoints() - tail), whole.features.end(), appendedPts.features.begin());
            size_t cellsSeen = 0; bool same = true;
            Map::binIntoCells(appendedPts, [&](const std::string& id, DataPoints&& cell) {
                ++cellsSeen;
                const DataPoints got = mapper.cells().retrieveCell(id);
                same = same && got.getNbPoints() == cell.getNbPoints() && got.features == cell.features;
            });
            same = same && cellsSeen == mapper.cells().getAllCellIds().size();
            std::printf("rank %d: cells equal to a host binning of the appended points: %s\n", rank, same ? "yes" : "NO");
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
