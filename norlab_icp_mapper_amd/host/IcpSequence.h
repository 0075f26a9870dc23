// IcpSequence.h -- `GpuICPSequence`: the four calls norlab_icp_mapper makes on its
// `PM::ICPSequence icp` member (Mapper.h:23), over the C ABI of include/icpmi.h:
//   loadFromYamlNode / setDefault   Mapper.cpp:72,77
//   setMap                          Map.cpp:111,178,528,581
//   operator()                      Mapper.cpp:213
//   errorMinimizer->getOverlap()    Mapper.cpp:219
// plus the registrar-created helpers the mapper uses next to it: RigidTransformation (Mapper.cpp:22,
// Map.cpp:14) and the DataPointsFilters of the shipped configuration (Mapper.cpp:27-31,82,92).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "../../include/icpmi.h"
#include "PointCloud.h"
#include "Yaml.h"

namespace nim {

class DataPointsFilters;

class GpuICPSequence {
public:
    struct ErrorMinimizerView {
        const GpuICPSequence* owner;
        float getOverlap() const { return owner->lastStats.sensor_noise_overlap >= 0.f ? owner->lastStats.sensor_noise_overlap : owner->lastStats.weighted_point_used_ratio; }
        float getPointUsedRatio() const { return owner->lastStats.point_used_ratio; }
    };

    explicit GpuICPSequence(int device = 0);
    ~GpuICPSequence();
    GpuICPSequence(const GpuICPSequence&) = delete;
    GpuICPSequence& operator=(const GpuICPSequence&) = delete;

    void setDefault();                                 // PM::ICPSequence::setDefault
    void loadFromYamlNode(const yaml::Node& icpNode);  // the `icp:` sub-tree
    bool setMap(const DataPoints& map);                // false on an empty cloud, state unchanged
    bool hasMap() const;
    Mat4 operator()(const DataPoints& reading);        // correction in the map frame
    const ErrorMinimizerView* errorMinimizer = &minimizerView;

    icpmi_handle handle() const { return h; }          // for the operators that share the GPU context
    // Map::updateLocalPointCloud for the PointDistance chain on the resident map (icpmi_map_update_point_distance)
    void mapUpdatePointDistance(const DataPoints& inputInMapFrame, float minDist, int normalsKnn, std::vector<uint8_t>& keep, int64_t& appended,
                                int64_t& mapSize);
    DataPoints downloadMap() const;                    // the resident map (features + `normals` if it has them)
    // Mapper::processInput with the scan staged once on the GPU (icpmi_register_prior / icpmi_map_update_staged)
    Mat4 registerWithPrior(const DataPoints& scanInSensorFrame, const Mat4& prior);
    void mapUpdateStaged(const Mat4& correction, float minDist, int normalsKnn, std::vector<uint8_t>& keep, int64_t& appended, int64_t& mapSize);
    // Map::updateLocalPointCloud for a whole module chain + post filters on the resident map (icpmi_map_update_chain /
    // icpmi_map_update_chain_staged when `input` is null: the scan kept by registerWithPrior, moved by `correction`).
    // src[j] for j >= prefix: provenance of new map point j in [old map ; scan]; the first `prefix` points are untouched.
    // pose: the post filters run in the sensor frame (the map is moved by pose^-1 and back by pose, Map.cpp:523-525), on the device.
    void mapUpdateChain(const DataPoints* inputInMapFrame, const Mat4& correction, const std::string& scalarName, const DataPoints& scanDescriptors,
                        const Mat4& pose, const std::vector<icpmi_map_op>& ops, int nModules, std::vector<int32_t>& src, int64_t& prefix,
                        int64_t& mapSize, bool wantSrc = true); // wantSrc = false: no host-side descriptor needs the provenance vector
    void uploadMapScalar(const std::vector<float>& scalar);   // the tracked scalar descriptor of the resident map
    std::vector<float> downloadMapScalar() const;
    int64_t residentMapSize() const;
    void setPlanar(bool on);                           // 2-D clouds (z == 0): planar minimisers and normals
    bool isPlanar() const { return planar; }
    const std::string& genericDescriptorName() const { return genericDescName; } // GenericDescriptorOutlierFilter.descName, or empty
    bool chainNeedsReadingNormals() const;             // SurfaceNormalOutlierFilter in the chain
    // true when the chain filters the reading inside operator() (readingDataPointsFilters / readingStepDataPointsFilters):
    // the staged-scan path of Mapper::processInput hands the unfiltered scan to the GPU and must not be taken then
    bool hasReadingFilters() const;
    // referenceDataPointsFilters present: every map must pass through setMap (the resident map update rebuilds the index
    // from the device copy and would skip them)
    bool hasReferenceFilters() const { return referenceDataPointsFilters != nullptr; }
    const icpmi_stats& stats() const { return lastStats; }
    const icpmi_config& config() const { return cfg; }
    static void check(icpmi_handle h, icpmi_status s); // status -> exception mapping

    // The DataPointsFilters chains INSIDE the ICP object (PM::ICPChainBase: `readingDataPointsFilters`, applied to a copy of
    // the reading in its incoming frame at the top of operator(); `readingStepDataPointsFilters`, applied to a copy of the
    // reading at the top of every iteration -- every filter here gives the same result on the same input, so once;
    // `referenceDataPointsFilters`, applied to the centred copy of the map inside setMap).
    std::shared_ptr<DataPointsFilters> readingDataPointsFilters, readingStepDataPointsFilters, referenceDataPointsFilters;

private:
    void recreate();
    DataPoints filteredReading(const DataPoints& reading) const;
    icpmi_handle h = nullptr;
    icpmi_config cfg;
    icpmi_stats lastStats{};
    size_t stagedPoints = 0;                           // size of the scan kept on the GPU by registerWithPrior
    std::string genericDescName;                       // GenericDescriptorOutlierFilter.descName (empty: no such filter)
    std::string genericReadDescName;                   // ... of a filter with source: reading (the row goes to the device with every reading)
  public:
    bool readsReadingDescriptor() const { return !genericReadDescName.empty(); }
  private:
    bool planar = false;                               // 2-D mapping (is3D == false): icpmi_config::is_2d
    ErrorMinimizerView minimizerView{this};
};

// PM::Transformation created with "RigidTransformation": features' = T * features, descriptors named
// "normals" / "observationDirections" rotated, everything else copied.
class RigidTransformation {
public:
    explicit RigidTransformation(icpmi_handle ctx) : h(ctx) {}
    DataPoints compute(const DataPoints& cloud, const Mat4& T) const;
private:
    icpmi_handle h;
};

// ---- DataPointsFilters used by the mapper and the shipped configuration -----------------------
class DataPointsFilter {
public:
    virtual ~DataPointsFilter() = default;
    virtual void inPlaceFilter(DataPoints& cloud) const = 0;
    // > 0 only for SurfaceNormalDataPointsFilter: lets Map run that post filter on the resident map
    virtual int surfaceNormalKnn() const { return 0; }
    // true when the filter can run as a step of icpmi_map_update_chain on the resident map: fills `op` and names the
    // scalar descriptor it reads (empty: none)
    virtual bool residentOp(icpmi_map_op& op, std::string& scalarName) const { (void)op; (void)scalarName; return false; }
    // true for per-point predicates (DistanceLimit, BoundingBox): a run of them is one icpmi_filter_points pass
    virtual bool pointFilter(icpmi_point_filter& f) const { (void)f; return false; }
    // false when two calls on the same cloud may differ (RandomSampling seeded from std::random_device): such a filter cannot
    // serve as a readingStepDataPointsFilter, which the accelerated loop applies once instead of once per iteration
    virtual bool repeatable() const { return true; }
};

class DataPointsFilters {
public:
    DataPointsFilters() = default;
    // a YAML sequence of single-key maps, e.g. the `input:` and `post:` sections (Mapper.cpp:82,92)
    DataPointsFilters(const yaml::Node& seq, icpmi_handle ctx);
    // the chain in order; runs of per-point predicates (optionally starting with `leading`, the mapper's radius filter,
    // Mapper.cpp:187-191) go through one fused GPU pass and one compaction
    void apply(DataPoints& cloud, const DataPointsFilter* leading = nullptr) const;
    size_t size() const { return filters.size(); }
    std::vector<std::shared_ptr<DataPointsFilter>> filters;
    icpmi_handle ctx = nullptr;
};

std::shared_ptr<DataPointsFilter> createDataPointsFilter(const std::string& name, const yaml::Node& params, icpmi_handle ctx);

} // namespace nim
