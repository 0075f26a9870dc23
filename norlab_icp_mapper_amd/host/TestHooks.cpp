// TestHooks.cpp -- a C entry into the host shell for the Python parity tests (tests/test_host_filters.py): runs a YAML sequence
// of DataPointsFilters (what `input:` / `post:` / the ICP chains' filter lists hold) on one cloud.  Not part of the reference
// surface; the product never calls it.
#include <cstring>

#include "IcpSequence.h"

namespace nim { uint32_t minstdNth(uint32_t seed, uint32_t n); } // IcpSequence.cpp: the generator RandomSampling / MaxDensity / SamplingSurfaceNormal draw from

extern "C" {

// the raw n-th value of the host filters' std::minstd_rand restatement ([rand.predef]: seed 1, n = 10 000 -> 399268537)
uint32_t nim_test_minstd_nth(uint32_t seed, uint32_t n) { return nim::minstdNth(seed, n); }

// yaml_seq: e.g. "- RandomSamplingDataPointsFilter: {prob: 0.5, seed: 3}".  in: 4 x n features + optional descriptor `desc_name`
// (span x n).  out4 (capacity 4 n), out_normals3 (3 n, may be NULL: receives `normals` if the result has them), out_desc (span x n,
// may be NULL: the input descriptor after filtering).  Returns 0, or 1 with the exception text in err.
int nim_test_filter_chain(icpmi_handle h, const char* yaml_seq, const float* in4, int64_t n, const char* desc_name, int desc_span,
                          const float* desc, float* out4, float* out_normals3, float* out_desc, int64_t* n_out, int* has_normals,
                          char* err, int err_cap)
{
    try {
        nim::DataPoints c((size_t)n);
        std::memcpy(c.features.data(), in4, sizeof(float) * 4 * (size_t)n);
        if (desc_name && desc) c.addDescriptor(desc_name, desc_span, std::vector<float>(desc, desc + (size_t)desc_span * n));
        nim::DataPointsFilters chain(nim::yaml::Load(yaml_seq), h);
        chain.apply(c);
        const size_t m = c.getNbPoints();
        std::memcpy(out4, c.features.data(), sizeof(float) * 4 * m);
        const bool hn = c.descriptorExists("normals");
        if (has_normals) *has_normals = hn ? 1 : 0;
        if (hn && out_normals3) std::memcpy(out_normals3, c.getDescriptorByName("normals").data.data(), sizeof(float) * 3 * m);
        if (desc_name && out_desc && c.descriptorExists(desc_name)) std::memcpy(out_desc, c.getDescriptorByName(desc_name).data.data(), sizeof(float) * (size_t)desc_span * m);
        *n_out = (int64_t)m;
        return 0;
    } catch (const std::exception& e) {
        if (err && err_cap > 0) { std::strncpy(err, e.what(), (size_t)err_cap - 1); err[err_cap - 1] = 0; }
        return 1;
    }
}

} // extern "C"
