// PointCloudIO.cpp -- ASCII VTK reader / writer for the dialect libpointmatcher writes and the
// reference's example data uses (examples/data/scans/*.vtk):
//   # vtk DataFile Version 3.0 / <title> / ASCII / DATASET POLYDATA
//   POINTS n float, VERTICES n 2n, POINT_DATA n, then SCALARS <name> float [+ LOOKUP_TABLE default],
//   VECTORS <name> float, NORMALS <name> float.
#include "PointCloud.h"

#include <cstdio>
#include <fstream>
#include <sstream>

namespace nim {

DataPoints DataPoints::load(const std::string& path)
{
    std::ifstream in(path);
    if (!in) throw std::runtime_error("Cannot open file " + path);
    std::string line, word;
    DataPoints cloud;
    size_t n = 0;
    bool seenPoints = false;
    while (in >> word) {
        if (word == "#") { std::getline(in, line); continue; }
        if (word == "POINTS") {
            std::string type;
            in >> n >> type;
            cloud = DataPoints(n);
            for (size_t i = 0; i < n; ++i) in >> cloud.features[4 * i] >> cloud.features[4 * i + 1] >> cloud.features[4 * i + 2];
            seenPoints = true;
        } else if (word == "VERTICES" || word == "LINES" || word == "POLYGONS") {
            size_t cnt, total;
            in >> cnt >> total;
            long long skip;
            for (size_t i = 0; i < total; ++i) in >> skip;
        } else if (word == "POINT_DATA") {
            size_t cnt;
            in >> cnt;
            if (cnt != n) throw std::runtime_error("POINT_DATA size mismatch in " + path);
        } else if (word == "SCALARS") {
            std::string name, type;
            in >> name >> type;
            std::getline(in, line); // optional numComp on the same line
            int span = 1;
            { std::istringstream ls(line); int c; if (ls >> c) span = c; }
            std::streampos pos = in.tellg();
            in >> word;
            if (word == "LOOKUP_TABLE") in >> word; else in.seekg(pos);
            std::vector<float> data((size_t)span * n);
            for (auto& v : data) in >> v;
            cloud.addDescriptor(name, span, std::move(data));
        } else if (word == "VECTORS" || word == "NORMALS") {
            std::string name, type;
            in >> name >> type;
            std::vector<float> data(3 * n);
            for (auto& v : data) in >> v;
            cloud.addDescriptor(name, 3, std::move(data));
        } else if (word == "vtk" || word == "ASCII" || word == "DATASET" || word == "POLYDATA") {
            if (word == "vtk") std::getline(in, line);
        } else {
            // title line and anything else: skip to end of line
            std::getline(in, line);
        }
    }
    if (!seenPoints) throw std::runtime_error("No POINTS section in " + path);
    return cloud;
}

void DataPoints::save(const std::string& path) const
{
    FILE* f = std::fopen(path.c_str(), "w");
    if (!f) throw std::runtime_error("Cannot open file " + path + " for writing");
    const size_t n = getNbPoints();
    std::fprintf(f, "# vtk DataFile Version 3.0\nFile created by norlab_icp_mapper_amd\nASCII\nDATASET POLYDATA\n");
    std::fprintf(f, "POINTS %zu float\n", n);
    for (size_t i = 0; i < n; ++i) std::fprintf(f, "%.9g %.9g %.9g\n", features[4 * i], features[4 * i + 1], features[4 * i + 2]);
    std::fprintf(f, "VERTICES %zu %zu\n", n, 2 * n);
    for (size_t i = 0; i < n; ++i) std::fprintf(f, "1 %zu\n", i);
    std::fprintf(f, "POINT_DATA %zu\n", n);
    for (const auto& d : descriptors) {
        if (d.name == "normals" && d.span == 3) std::fprintf(f, "NORMALS %s float\n", d.name.c_str());
        else if (d.span == 3) std::fprintf(f, "VECTORS %s float\n", d.name.c_str());
        else if (d.span == 1) std::fprintf(f, "SCALARS %s float\nLOOKUP_TABLE default\n", d.name.c_str());
        else std::fprintf(f, "SCALARS %s float %d\nLOOKUP_TABLE default\n", d.name.c_str(), d.span);
        for (size_t i = 0; i < n; ++i) {
            for (int r = 0; r < d.span; ++r) std::fprintf(f, r ? " %.9g" : "%.9g", d.data[(size_t)d.span * i + r]);
            std::fprintf(f, "\n");
        }
    }
    std::fclose(f);
}

} // namespace nim
