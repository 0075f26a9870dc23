// oracle_pm.cpp -- OPTIONAL checker: the real libpointmatcher PM::ICPSequence behind a C entry point.
//
// TEST INFRASTRUCTURE ONLY (same rule as icp_oracle.c): only tests/, __graft_entry__.smoke() and the cpu_baseline leg
// of bench.py may load what this builds.  It is compiled only where <pointmatcher/PointMatcher.h> exists (`make oracle_pm`
// probes for it; the development container and, so far, the GPU box have neither libpointmatcher nor libnabo -- then the
// target reports "libpointmatcher: absent" and builds nothing).  Where it does build, `orc_pm_register` runs the very
// object norlab_icp_mapper holds (`PM::ICPSequence icp`, Mapper.h:23; `icp.loadFromYaml`, `icp.setMap`, `icp(reading)`,
// Mapper.cpp:72,213; Map.cpp:528) on the same clouds and YAML chain as the HIP path, which is the only way to turn
// "parity unpinned"  into a pinned statement.
#include <pointmatcher/PointMatcher.h>

#include <cstdint>
#include <cstring>
#include <sstream>
#include <string>

typedef PointMatcher<float> PM;

static PM::DataPoints make_cloud(const float* pts4, int64_t n, const float* normals3)
{
    PM::DataPoints::Labels fl;
    fl.push_back(PM::DataPoints::Label("x", 1));
    fl.push_back(PM::DataPoints::Label("y", 1));
    fl.push_back(PM::DataPoints::Label("z", 1));
    fl.push_back(PM::DataPoints::Label("pad", 1));
    PM::Matrix features = Eigen::Map<const PM::Matrix>(pts4, 4, n);
    if (!normals3) return PM::DataPoints(features, fl);
    PM::DataPoints::Labels dl;
    dl.push_back(PM::DataPoints::Label("normals", 3));
    PM::Matrix descriptors = Eigen::Map<const PM::Matrix>(normals3, 3, n);
    return PM::DataPoints(features, fl, descriptors, dl);
}

extern "C" {

// 1 when the library this file was built against is usable
int orc_pm_available(void) { return 1; }

// yaml: the `icp:` sub-tree as text (empty / NULL: icp.setDefault(), Mapper.cpp:77).  T_out: column-major 4x4, the
// correction in the map frame (what Mapper.cpp:213 receives).  Returns 0, or 1 with the exception text in err.
int orc_pm_register(const char* yaml, const float* map4, int64_t m, const float* map_normals3, const float* scan4, int64_t n,
                    const float* scan_normals3, float T_out[16], float* overlap_out, char* err, int err_cap)
{
    try {
        PM::ICPSequence icp;
        if (yaml && *yaml) { std::istringstream in{std::string(yaml)}; icp.loadFromYaml(in); }
        else icp.setDefault();
        icp.setMap(make_cloud(map4, m, map_normals3));
        const PM::TransformationParameters T = icp(make_cloud(scan4, n, scan_normals3));
        for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) T_out[4 * c + r] = T(r, c);
        if (overlap_out) *overlap_out = icp.errorMinimizer->getOverlap();
        return 0;
    } catch (const std::exception& e) {
        if (err && err_cap > 0) { std::strncpy(err, e.what(), (size_t)err_cap - 1); err[err_cap - 1] = 0; }
        return 1;
    }
}

} // extern "C"
