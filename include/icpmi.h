/*
 * include/icpmi.h -- C ABI of the MI355X-native ICP registration core ("icpmi").
 *
 * This is the drop-in boundary for the hot path of norlab_icp_mapper (reference tree
 * /root/reference, v2.1.0).  The reference has no FFI of its own: the path is entered through the
 * in-process C++ object `PM::ICPSequence icp` (norlab_icp_mapper/Mapper.h:23) and the MapperModule
 * virtuals (norlab_icp_mapper/MapperModules/MapperModule.h:20-29).  Each entry point below names the
 * reference interface it replaces; INTEGRATION.md shows the C++ shim (`GpuICPSequence`,
 * `Gpu*MapperModule`) a maintainer would add on the reference side.
 *
 * Conventions
 *   - plain C, no torch / Eigen / HIP types in any signature;
 *   - clouds are `features` blocks of PM::DataPoints: 4 x N column-major float (x,y,z,1), i.e. N
 *     consecutive float4; normals are 3 x N column-major float (descriptor "normals");
 *   - 4x4 transforms are column-major float[16] (Eigen's default, PM::TransformationParameters);
 *   - pointers are HOST pointers unless the function name ends in `_dev`, in which case cloud
 *     pointers are device (HBM) pointers on the handle's GPU; the callee never retains caller
 *     memory (DataPoints value semantics, SURVEY.md 8b "Ownership");
 *   - every function returns an icpmi_status; icpmi_last_error(h) gives the message.  The reference
 *     reports the same conditions as C++ exceptions (PM::ConvergenceError, ...); the mapping is
 *     given next to each status code;
 *   - one HIP stream per handle; every entry point locks its handle for the duration of the call, so
 *     calls on one handle from several threads are safe and run one after the other (the reference
 *     serialises icp() and icp.setMap() with icpMapLock, Mapper.cpp:212, Map.cpp:527-529, but its
 *     modules and filters -- each with a private kd-tree -- run outside that lock: online mode reaches
 *     one handle from the caller's thread, the std::async update thread and the paging thread);
 *     pairs of calls that hand state to each other (icpmi_register_prior -> icpmi_map_update_*_staged)
 *     still belong to one logical owner; different handles are independent (one per GPU).
 */
#ifndef ICPMI_H
#define ICPMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ICPMI_VERSION 4

typedef struct icpmi_ctx* icpmi_handle;

typedef enum {
    ICPMI_OK = 0,
    ICPMI_ERR_INVALID_ARG = 1,          /* PM::Parametrizable::InvalidParameter / std::invalid_argument */
    ICPMI_ERR_HIP = 2,                  /* device runtime failure (no reference analogue)               */
    ICPMI_ERR_NO_POINT_TO_MINIMIZE = 3, /* PM::ConvergenceError("ErrorMnimizer: no point to minimize")  */
    ICPMI_ERR_NO_OUTLIER_TO_FILTER = 4, /* PM::ConvergenceError("no outlier to filter")                 */
    ICPMI_ERR_BOUND = 5,                /* PM::ConvergenceError from BoundTransformationChecker         */
    ICPMI_ERR_NAN = 6,                  /* PM::ConvergenceError("abs rotation norm not a number")       */
    ICPMI_ERR_MISSING_NORMALS = 7,      /* PM::DataPoints::InvalidField("normals")                      */
    ICPMI_ERR_UNSUPPORTED = 8
} icpmi_status;

/* ErrorMinimizer selection (icp.errorMinimizer in the YAML chain, Mapper.cpp:72) */
typedef enum {
    ICPMI_MIN_IDENTITY = 0,       /* IdentityErrorMinimizer (examples/config.yaml:62-63) */
    ICPMI_MIN_POINT_TO_POINT = 1, /* PointToPointErrorMinimizer                          */
    ICPMI_MIN_POINT_TO_PLANE = 2  /* PointToPlaneErrorMinimizer (docs/MapperConfiguration.md:174-189) */
} icpmi_minimizer;

/* OutlierFilters (icp.outlierFilters) */
typedef enum {
    ICPMI_OUT_MAXDIST = 1,      /* MaxDistOutlierFilter{maxDist}        w = d2 <= maxDist^2          */
    ICPMI_OUT_MINDIST = 2,      /* MinDistOutlierFilter{minDist}        w = d2 >= minDist^2          */
    ICPMI_OUT_MEDIANDIST = 3,   /* MedianDistOutlierFilter{factor}      w = d2 <= factor*median(d2)  */
    ICPMI_OUT_TRIMMEDDIST = 4,  /* TrimmedDistOutlierFilter{ratio}      w = d2 <= quantile(d2,ratio) */
    ICPMI_OUT_SURFACENORMAL = 5,/* SurfaceNormalOutlierFilter{maxAngle} w = n_read.n_ref > cos(maxAngle) */
    /* GenericDescriptorOutlierFilter{source: reference, descName, useSoftThreshold, useLargerThan, threshold}: the 1-row descriptor
     * of the matched map point -- the tracked scalar channel, icpmi_set_map_scalar -- decides.  param = threshold, iparam = flags.
     * hard: w = desc > threshold (useLargerThan) or desc < threshold; soft: w = desc.  source: reading (iparam | ICPMI_GEN_SOURCE_READING,
     * v4): the descriptor of the READING point decides -- its row is handed over with icpmi_set_reading_scalar before every registration. */
    ICPMI_OUT_GENERICDESCRIPTOR = 6,
    /* RobustOutlierFilter{robustFct, tuning, scaleEstimator: none | mad | berg | std, nbIterationForScale, distanceType,
     * approximation}: M-estimator weight of e2 = residual / scale^2.  param = tuning, param2 = nbIterationForScale, param3 =
     * approximation (0 or +inf: none; e2 >= approximation^2 forces the weight to 0), iparam = robustFct | scaleEstimator << 4 |
     * distanceType << 8.  berg (r5): scale 1.9 sqrt(median d2) at iteration 1, then 0.85 (scale - tuning) + tuning, with Bergstrom's
     * constants as the M-estimator's tuning (cauchy 4.3040, tukey 7.0589, huber 2.0138).  std (r5): sqrt of the standard deviation of
     * EVERY entry of the distance matrix (an infinite entry makes the registration end with ICPMI_ERR_NAN, as it poisons upstream's). */
    ICPMI_OUT_ROBUST = 7,
    /* VarTrimmedDistOutlierFilter{minRatio, maxRatio, lambda} (Phillips et al. 2007): TrimmedDist at the ratio that minimises
     * FRMS(i) = (sum of the i + 1 smallest d2) / ((i + 1) ((i + 1) / N)^(2 lambda)) over floor(minRatio N) <= i < floor(maxRatio N).
     * param = minRatio, param2 = maxRatio, param3 = lambda. */
    ICPMI_OUT_VARTRIMMEDDIST = 8
} icpmi_outlier_type;
enum { ICPMI_GEN_SOURCE_READING = 1, ICPMI_GEN_SOFT = 2, ICPMI_GEN_LARGER = 4 };
enum { ICPMI_ROB_CAUCHY = 0, ICPMI_ROB_WELSCH = 1, ICPMI_ROB_SC = 2, ICPMI_ROB_GM = 3, ICPMI_ROB_TUKEY = 4, ICPMI_ROB_HUBER = 5,
       ICPMI_ROB_L1 = 6, ICPMI_ROB_STUDENT = 7 };
enum { ICPMI_SCALE_NONE = 0, ICPMI_SCALE_MAD = 1, ICPMI_SCALE_BERG = 2, ICPMI_SCALE_STD = 3 };
enum { ICPMI_DIST_POINT2POINT = 0, ICPMI_DIST_POINT2PLANE = 1 };

typedef struct {
    int32_t type; /* icpmi_outlier_type */
    float   param;
    int32_t iparam; /* flags / enums of GenericDescriptor and Robust, 0 otherwise */
    float   param2; /* Robust: nbIterationForScale; VarTrimmedDist: maxRatio        */
    float   param3; /* VarTrimmedDist: lambda; Robust: approximation (0 / +inf: none)  */
} icpmi_outlier;

typedef enum { ICPMI_STOP_NONE = 0, ICPMI_STOP_COUNTER = 1, ICPMI_STOP_DIFFERENTIAL = 2 } icpmi_stop_reason;

/* The ICP chain of PM::ICPSequence::loadFromYamlNode (Mapper.cpp:72) restricted to the modules on
 * the hot path.  icpmi_config_default() fills libpointmatcher's defaults for those modules. */
typedef struct {
    int32_t device;            /* HIP device ordinal                                              */
    /* matcher: KDTreeMatcher */
    int32_t knn;               /* default 1                                                       */
    float   max_dist;          /* default +inf                                                    */
    float   epsilon;           /* default 0; > 0 is served by the EXACT search (a valid epsilon-    */
                               /* answer, but not libnabo's pick -- SURVEY.md 0.4) unless           */
                               /* epsilon_approx below asks for libnabo's pruning                   */
    /* outlier filters, applied in order, weights multiply */
    int32_t n_outlier;
    icpmi_outlier outlier[8];
    /* error minimiser */
    int32_t minimizer;         /* icpmi_minimizer                                                 */
    int32_t force_4dof;        /* PointToPlaneErrorMinimizer.force4DOF: yaw + translation only     */
    int32_t force_2d;          /* PointToPlaneErrorMinimizer.force2D on 3-D clouds: yaw + (tx, ty), */
                               /* residual = the 2-D dot with the top two rows of the normals       */
    int32_t is_2d;             /* planar clouds (the mapper's is3D == false, Mapper.h:53): every    */
                               /* point has z == 0; point-to-point solves the 2-D rotation,         */
                               /* point-to-plane the force2D system, SurfaceNormal the 2 x 2 problem */
    /* transformation checkers */
    int32_t max_iterations;    /* CounterTransformationChecker.maxIterationCount, default 40      */
    int32_t use_differential;  /* DifferentialTransformationChecker present                       */
    float   min_diff_rot;      /* default 0.001                                                   */
    float   min_diff_trans;    /* default 0.001                                                   */
    int32_t smooth_length;     /* default 3 (<= 16)                                               */
    int32_t use_bound;         /* BoundTransformationChecker present                              */
    float   max_rot_norm;      /* default 1                                                       */
    float   max_trans_norm;    /* default 1                                                       */
    /* engine knobs (no reference analogue) */
    float   grid_cell;         /* NN grid cell edge in metres; 0 = choose from the map density     */
    int32_t use_graph;         /* 1 = fixed-iteration registrations replay one hipGraph (default) */
    int32_t profile;           /* 1 = eager launches with HIP events around every NN launch        */
    int32_t fuse_solve;        /* (v4) asked for the solve inside the next NN launch (three launches per iteration).  The path lost  */
                               /* on two of three shapes and was REMOVED in r5: the field keeps the struct's layout    */
                               /* and is ignored (every value runs the four-launch iteration, bit-identical results as before).     */
    /* r6: the two switches of the k > 1 matcher that tests still need were environment variables through r5 (read once per process); they are
       per-handle fields now.  Zero = the default, so a zeroed tail keeps every old caller's behaviour.                                        */
    int32_t knn_wg_from;       /* which iterations of a k > 1 loop nnk_wg_kernel serves: 0 = from iteration 2 (default), n > 0 = from          */
                               /* iteration n - 1, < 0 = none (nnk_ml_kernel everywhere).  Same bits either way (tests/test_gpu_knn_wg.py).    */
    int32_t sel_window_off;    /* 1 = the quantile selection of a k > 1 loop always builds its full level-0 histogram (no speculative window, */
                               /* DESIGN_history.md 13.7b).  Same bits either way (tests/test_gpu_sel_window.py).                              */
    int32_t epsilon_approx;    /* (r6) 1 = `epsilon` prunes the matcher's search as libnabo's maxError2 does (`new_rd * (1 + epsilon)^2 <         */
                               /* heap.headValue()`): cells farther than (k-th distance so far) / (1 + epsilon) are not visited, a query is decided  */
                               /* once its k-th distance is within (1 + epsilon) x the margin of its block.  Every returned distance d_j <=           */
                               /* (1 + epsilon) x the exact j-th distance (tests/test_gpu_epsilon.py); WHICH valid answer comes back differs from     */
                               /* libnabo's (its pick depends on the kd-tree's traversal order).  0 (default): the exact search.  k <= 16; filters'   */
                               /* own searches (PointDistance, SurfaceNormal) stay exact.                                                              */
    int32_t reserved[4];
} icpmi_config;

/* What PM::ICPSequence exposes after a call: errorMinimizer->getOverlap() (Mapper.cpp:219) is
 * weighted_point_used_ratio; the inspector statistics of SURVEY.md section 5 are the rest. */
typedef struct {
    int32_t iterations;
    int32_t stop_reason;               /* icpmi_stop_reason                                       */
    int64_t pairs;                     /* P of the last iteration                                 */
    float   point_used_ratio;          /* P / (knn N)                                             */
    float   weighted_point_used_ratio; /* sum w / (knn N) == getOverlap()                         */
    float   trimmed_limit;             /* last quantile limit on d^2 (Trimmed / Median), else -1  */
    float   loop_ms;                   /* device time of the iteration loop (HIP events)          */
    float   nn_ms_avg;                 /* mean NN launch time: profile mode = HIP events (event-pair gap removed); otherwise device  */
                                       /* clocks from the NN kernel's first workgroup to the first workgroup of the next kernel       */
    int32_t nn_launches;               /* number of NN launches averaged                          */
    int64_t hard_queries;              /* NN queries the grid pyramid could not decide (brute pass) */
    int32_t reserved[4];
    float   sensor_noise_overlap;      /* getOverlap() of a reading that carries `simpleSensorNoise` and `normals`
                                          (icpmi_set_reading_sensor_noise): the share of the last iteration's pairs within the
                                          sensor noise; -1 when not computed (then getOverlap() == weighted_point_used_ratio) */
    int32_t reserved2;
} icpmi_stats;

void icpmi_config_default(icpmi_config* cfg);

/* Replaces constructing / configuring `PM::ICPSequence icp` (Mapper.h:23, Mapper.cpp:72,77). */
icpmi_status icpmi_create(const icpmi_config* cfg, icpmi_handle* out);
/* Re-configure the chain of an existing handle (`icp.loadFromYamlNode` / `icp.setDefault()` called
 * again on the same object): the handle, its device and its map stay. cfg->device must not change. */
icpmi_status icpmi_set_config(icpmi_handle h, const icpmi_config* cfg);
void         icpmi_destroy(icpmi_handle h);
const char*  icpmi_last_error(icpmi_handle h); /* h may be NULL: error of the last failed create */

/* Replaces `bool PM::ICPSequence::setMap(const DataPoints&)` (Map.cpp:111,178,528,581): copies the
 * cloud, centres it on its centroid, builds the NN index on the device.  *accepted = 0 and state
 * unchanged when m == 0 (upstream returns false and warns).  normals3 may be NULL. */
icpmi_status icpmi_set_map(icpmi_handle h, const float* map4, int64_t m, const float* normals3, int32_t* accepted);
icpmi_status icpmi_set_map_dev(icpmi_handle h, const float* d_map4, int64_t m, const float* d_normals3, int32_t* accepted);
/* `icp.hasMap()` */
int32_t      icpmi_has_map(icpmi_handle h);
/* centroid used for centring (T_refIn_refMean translation) */
icpmi_status icpmi_get_map_mean(icpmi_handle h, float mean3[3]);

/* `ErrorMinimizer::getOverlap()` (read at Mapper.cpp:219, used by the `overlap` update condition, Mapper.cpp:257-260): when the reading
 * carries the descriptors `simpleSensorNoise` (1 x N) and `normals`, upstream's minimisers do not return the weighted ratio but the share of
 * the last iteration's pairs that lie within the sensor noise -- PointToPoint: |p - q| < mean|p - q| + noise_i; PointToPlane:
 * |(p - q) . n_i / |n_i|| < noise_i with n_i the READING's normal.  This call hands over the noise row of the NEXT registration's reading
 * (host pointer, n floats, one shot: it is consumed by that registration, whose scan_normals3 must be given and whose n must match; any
 * other case leaves stats.sensor_noise_overlap at -1).  noise == NULL or n == 0 clears it. */
icpmi_status icpmi_set_reading_sensor_noise(icpmi_handle h, const float* noise, int64_t n);

/* (v4) GenericDescriptorOutlierFilter{source: reading}: the 1-row descriptor `descName` of the NEXT registration's reading (host pointer, n
 * floats, one shot: consumed by that registration, whose n must match -- else it fails with InvalidField like upstream's missing descriptor).
 * scalar == NULL or n == 0 clears it.  Registrations of a batch with such a chain are not served (every reading needs its own row). */
icpmi_status icpmi_set_reading_scalar(icpmi_handle h, const float* scalar, int64_t n);

/* Replaces `TransformationParameters PM::ICPSequence::operator()(const DataPoints&)`
 * (Mapper.cpp:213): scan4 is the reading already moved by the prior (Mapper.cpp:197); T_out is the
 * correction in the map frame (so that correctedPose = T_out * estimatedPose, Mapper.cpp:215).
 * Without a map returns identity and ICPMI_OK like upstream. scan_normals3 may be NULL. */
icpmi_status icpmi_register(icpmi_handle h, const float* scan4, int64_t n, const float* scan_normals3,
                            float T_out[16], icpmi_stats* stats);
icpmi_status icpmi_register_dev(icpmi_handle h, const float* d_scan4, int64_t n, const float* d_scan_normals3,
                                float T_out[16], icpmi_stats* stats);
/* Throughput mode: run exactly `iterations` loop passes (Counter only) -- what the bench times. */
icpmi_status icpmi_register_fixed_dev(icpmi_handle h, const float* d_scan4, int64_t n, const float* d_scan_normals3,
                                      int32_t iterations, float T_out[16], icpmi_stats* stats);

/* `icp(input)` for `batch` independent readings against the same map in ONE launch sequence (no reference analogue: the
 * reference registers one scan per call, Mapper.cpp:213; this is what a multi-stream front end -- BASELINE config 5: several
 * scan streams against a shared map -- calls instead of `batch` registrations): every kernel of the loop runs once per
 * iteration for all readings (blockIdx.y = reading), each reading keeps its own loop state and stops on its own checkers.
 * Results are bit-identical to registering each reading alone.  d_scans4: host array of `batch` device pointers (readings
 * already moved by their priors), n: their sizes; fixed_iterations > 0 = throughput mode (Counter only), 0 = the handle's
 * checkers.  T_out: 16 floats per reading; stats / status: one entry per reading (status may be NULL: the first error is
 * then the return value).  1 <= batch <= 16.  Chains the batched kernels do not serve (knn > 1, several quantile filters,
 * SurfaceNormalOutlierFilter, unbounded maxDist) are registered one reading after the other -- same results. */
icpmi_status icpmi_register_batch_dev(icpmi_handle h, int32_t batch, const float* const* d_scans4, const int64_t* n,
                                      int32_t fixed_iterations, float* T_out, icpmi_stats* stats, icpmi_status* status);

/* ---- stage-level entry points (the per-stage virtuals of SURVEY.md 8b B2; used by parity tests) ---- */

/* `Transformation::compute(cloud, T)` (Mapper.cpp:197,221; Map.cpp:523,525): out4 = T * in4;
 * normals (3 x n, may be NULL) are rotated by the top-left 3x3. Rejects |1 - det R| > 1e-3
 * (TransformationError) with ICPMI_ERR_INVALID_ARG. */
icpmi_status icpmi_transform(icpmi_handle h, const float T[16], const float* in4, int64_t n, float* out4,
                             const float* in_normals3, float* out_normals3);

/* `Matcher::findClosests` == `Nabo::NNS::knn` against the CENTRED map of the handle
 * (queries are given in the centred frame). ids: k x n int32 (-1 unfilled, ORIGINAL map indices),
 * d2: k x n float (+inf unfilled), ascending by (d2, id). allow_self = 0 reproduces optionFlags = 0
 * of PointDistanceMapperModule.cpp:36. max_dist may be +inf. */
icpmi_status icpmi_knn(icpmi_handle h, const float* q4, int64_t n, int32_t k, float max_dist, int32_t allow_self,
                       int32_t* ids, float* d2);

/* `OutlierFilters::compute` for the handle's chain on given matches (host arrays, k x n; ids = ORIGINAL map indices, needed by
 * SurfaceNormal / GenericDescriptor / Robust; read_normals3 = the reading's `normals`, 3 x n, needed by SurfaceNormalOutlierFilter,
 * whose map side comes from the normals handed to icpmi_set_map). */
icpmi_status icpmi_outlier_weights(icpmi_handle h, const float* d2, const int32_t* ids, int32_t k, int64_t n,
                                   const float* read_normals3, float* weights, float* limit_out);

/* `ErrorMinimizer::compute`: one minimisation step for a reading given in the centred map frame,
 * with T_iter applied on the device first (T_iter may be NULL = identity). Runs NN + outlier
 * filters + minimiser of the handle's chain once. Outputs: T_step (4x4), and for inspection
 * sums[32] (double): point-to-plane -> A (upper 21, row-major packed) then b (6); point-to-point ->
 * sum w, sum w p (3), sum w q (3), sum w q p^T (9). Any output pointer may be NULL. */
icpmi_status icpmi_minimize_step(icpmi_handle h, const float* reading4, int64_t n, const float* T_iter,
                                 float T_step[16], double sums[32], icpmi_stats* stats);

/* ---- map-side operators on the path ---- */

/* `SurfaceNormalDataPointsFilter{knn}` (Map.cpp:524 via examples/config.yaml:26-27). */
icpmi_status icpmi_surface_normals(icpmi_handle h, const float* pts4, int64_t m, int32_t knn, float* normals3);
/* ... with `keepDensities: 1`: densities (m floats, may be NULL) = knn / (4/3 pi r^3), r = the largest distance of a neighbour
 * from the centroid of the neighbourhood (the descriptor MaxDensityDataPointsFilter reads). */
icpmi_status icpmi_surface_normals_ex(icpmi_handle h, const float* pts4, int64_t m, int32_t knn, float* normals3, float* densities);
/* (v4) ... with `keepMatchedIds: 1` / `keepMeanDist: 1`: matched_ids (knn x m int32, column i = the neighbours of point i by ascending
 * (d2, index), the point itself first; may be NULL) and mean_dist (m floats: distance from the point to the mean of its neighbours; may be NULL). */
icpmi_status icpmi_surface_normals_ex2(icpmi_handle h, const float* pts4, int64_t m, int32_t knn, float* normals3, float* densities,
                                       int32_t* matched_ids, float* mean_dist);
/* (r5) ... with `keepEigenValues: 1` / `keepEigenVectors: 1` and `sortEigen: 1`: eig_values3 (3 x m, may be NULL) = the eigenvalues of the
 * neighbourhood's scatter matrix NN NN^T in ascending order; eig_vectors9 (9 x m, may be NULL) = upstream's serializeEigVec of the
 * eigenvector matrix in that column order (entry 3 k + j of a point = component k of eigenvector j).  Rank < 2: zeros / identity, as upstream.
 * The sign of an eigenvector is the solver's.  Upstream's UNSORTED order (sortEigen: 0) is whatever Eigen::EigenSolver returns and is not
 * reproduced: the host filter serves the two descriptors only together with sortEigen: 1. */
icpmi_status icpmi_surface_normals_ex3(icpmi_handle h, const float* pts4, int64_t m, int32_t knn, float* normals3, float* densities,
                                       int32_t* matched_ids, float* mean_dist, float* eig_values3, float* eig_vectors9);

/* `PointDistanceMapperModule::inPlaceUpdateMap` keep mask (PointDistanceMapperModule.cpp:28-50):
 * keep[i] = 1 iff the exact NN of input i in map (self match excluded) has d2 >= min_dist^2. */
icpmi_status icpmi_point_distance_keep(icpmi_handle h, const float* map4, int64_t m, const float* in4, int64_t n,
                                       float min_dist, uint8_t* keep);

/* `Map::updateLocalPointCloud` for the PointDistance chain, on the resident map (Map.cpp:502-534 with
 * PointDistanceMapperModule.cpp:28-50 as the only module and, when normals_knn > 0, SurfaceNormalDataPointsFilter{knn}
 * as the post filter, examples/config.yaml:26-27): scan4 (n x 4, MAP frame) -> keep mask against the handle's current
 * map (self match excluded, keep iff d2 >= min_dist^2) -> kept points appended in input order -> normals of the whole
 * grown map recomputed (normals_knn > 0; else the appended points take scan_normals3 or zeros) -> index rebuilt
 * (`icp.setMap`, Map.cpp:528).  Only the scan crosses PCIe; keep_out (n bytes, may be NULL) receives the keep mask so
 * that a host that owns further descriptors of the map can append the kept rows itself.  On a handle without a map the scan becomes the map
 * (`PointDistanceMapperModule::createMap`).  The post filter runs in the map frame (the reference rotates the cloud
 * into the sensor frame and back, Map.cpp:523-525; PCA normals are rotation-equivariant up to rounding). */
icpmi_status icpmi_map_update_point_distance(icpmi_handle h, const float* scan4, int64_t n, const float* scan_normals3,
                                             float min_dist, int32_t normals_knn, uint8_t* keep_out, int64_t* appended, int64_t* new_m);

/* `Mapper::processInput` with the scan staged once (Mapper.cpp:194-238): scan4 arrives in the SENSOR frame;
 * icpmi_register_prior moves it into the map frame by the prior on the device (`transformation->compute(input,
 * estimatedPose)`, :197), registers it (`icp(input)`, :213) and keeps that cloud in HBM; T_out is the correction.
 * If the update policy then asks for a map update, icpmi_map_update_staged applies the correction to the kept
 * cloud (`transformation->compute(input, correction)`, :221) and runs icpmi_map_update_point_distance on it --
 * the scan crosses PCIe once per processInput instead of five times. */
icpmi_status icpmi_register_prior(icpmi_handle h, const float* scan4, int64_t n, const float prior[16], float T_out[16],
                                  icpmi_stats* stats);
/* (v4) the same with the sensor-frame scan already in HBM on the handle's GPU: no PCIe crossing at all */
icpmi_status icpmi_register_prior_dev(icpmi_handle h, const float* d_scan4, int64_t n, const float prior[16], float T_out[16],
                                      icpmi_stats* stats);
icpmi_status icpmi_map_update_staged(icpmi_handle h, const float correction[16], float min_dist, int32_t normals_knn,
                                     uint8_t* keep_out, int64_t* appended, int64_t* new_m);

/* `PointDistanceMapperModule` keep mask of the scan staged by icpmi_register_prior against the RESIDENT map, without
 * touching the map (scan-sharded mapping, SURVEY.md 8e: every rank decides which of its points are new, the accepted
 * points of all ranks are exchanged, and only then does every replica append the same set): the staged cloud is moved by
 * `correction` (Mapper.cpp:221); keep_out[i] = 1 iff its exact nearest map neighbour is at least min_dist away
 * (PointDistanceMapperModule.cpp:33-42); placed_out4 (n x 4, may be NULL) receives the moved cloud. */
icpmi_status icpmi_staged_point_distance_keep(icpmi_handle h, const float correction[16], float min_dist, uint8_t* keep_out,
                                              float* placed_out4);

/* Download of the resident map in the caller's order (what `Map::getLocalPointCloud` returns, Map.cpp:536-540);
 * out4 / normals3 may be NULL to query *m only. */
icpmi_status icpmi_get_map(icpmi_handle h, float* out4, float* normals3, int64_t capacity, int64_t* m);

/* `OctreeMapperModule::inPlaceUpdateMap` decimation (OctreeMapperModule.cpp:35-39 -> OctreeGridDataPointsFilter
 * {maxSizeByNode: edge, samplingMethod: 0}), as a lattice stand-in: voxel index floor((p - lo) / edge) per axis
 * with lo the bounding-box minimum; keep[i] = 1 iff i is the smallest index of its voxel. */
icpmi_status icpmi_voxel_keep_first(icpmi_handle h, const float* in4, int64_t n, float edge, uint8_t* keep);

/* `DynamicPointsMapperModule::inPlaceUpdateMap` (DynamicPointsMapperModule.cpp:34-172; parameters :6-13).
 * `to_sensor` = pose^-1 (col-major 4x4, :51,57); input4 / map4 in the map frame; map_normals3 = descriptor `normals`
 * of the map; prob_dynamic (m floats) = descriptor `probabilityDynamic` of the map, updated in place for every
 * map point within sensor_max_range that has an input beam within 2 * beam_half_angle in (elevation, azimuth). */
typedef struct icpmi_dynpts_params {
    float threshold_dynamic; /* 0.6  */
    float alpha;             /* 0.8  */
    float beta;              /* 0.99 */
    float beam_half_angle;   /* 0.01 rad */
    float epsilon_a;         /* 0.01 */
    float epsilon_d;         /* 0.01 m */
    float sensor_max_range;  /* 200 m */
} icpmi_dynpts_params;
icpmi_status icpmi_dynamic_points_update(icpmi_handle h, const icpmi_dynpts_params* prm, const float to_sensor[16],
                                         const float* input4, int64_t n, const float* map4, const float* map_normals3,
                                         int64_t m, float* prob_dynamic);

/* Same decimation with a choice of representative: method 0 = smallest index of the voxel (icpmi_voxel_keep_first),
 * method 1 = `samplingMethod: 1` (a random point of the voxel, examples/config.yaml:49) made reproducible: the point
 * whose index has the smallest 32-bit finaliser hash (fmix32 of MurmurHash3, a bijection) represents its voxel. */
icpmi_status icpmi_voxel_keep(icpmi_handle h, const float* in4, int64_t n, float edge, int32_t method, uint8_t* keep);

/* `SamplingSurfaceNormalDataPointsFilter{ratio, knn, samplingMethod: 0, maxBoxDim, seed}` -- the REFERENCE filter of
 * `PM::ICPSequence::setDefault()` (a configuration without an `icp:` key, Mapper.cpp:74-78), run at every `icp.setMap` (Map.cpp:111,178,528,581):
 * median splits of the widest box dimension (ties by index) until a box holds <= knn points, one PCA normal per box (boxes of rank < 2 or
 * wider than maxBoxDim are dropped), the points of a box -- in index order -- kept with probability ratio (std::minstd_rand seeded with
 * `seed`, one number per point of a surviving box, boxes in depth-first order).  order_out (capacity n) receives the kept indices in box
 * order, normals3_out (capacity 3 n) their normals; *n_out the number kept.  Entirely on the device (csrc/ssn.hip). */
icpmi_status icpmi_sampling_surface_normal(icpmi_handle h, const float* in4, int64_t n, float ratio, int32_t knn, float max_box_dim, int32_t seed,
                                           int32_t* order_out, float* normals3_out, int64_t* n_out);
/* ... with `samplingMethod` (upstream's fuseRange, as recalled): 0 = the call above (the extra outputs are not touched); 1 = every surviving box
 * is replaced by ONE point: order_out[j] = the smallest index of box j (the column the caller keeps), mean3_out[3 j ..] its new position -- the
 * mean of the box, accumulated in double in index order --, normals3_out[3 j ..] the box normal, and members_out[member_start_out[j] ..
 * + member_count_out[j]) the members of the box in index order: what `averageExistingDescriptors` averages over (descriptor rows live with
 * the caller).  No random number is drawn; boxes in depth-first order; *n_out = number of boxes.  Capacities n (3 n for mean3_out); any
 * of the four extra outputs may be NULL.  Entirely on the device. */
icpmi_status icpmi_sampling_surface_normal_ex(icpmi_handle h, const float* in4, int64_t n, float ratio, int32_t knn, float max_box_dim, int32_t seed,
                                              int32_t method, int32_t* order_out, float* normals3_out, int64_t* n_out, float* mean3_out,
                                              int32_t* member_start_out, int32_t* member_count_out, int32_t* members_out);

/* `OctreeGridDataPointsFilter{maxSizeByNode, maxPointByNode, samplingMethod}` (created at OctreeMapperModule.cpp:12, applied at :38):
 * octree over the bounding cube of the cloud, split until the node edge is <= max_size or the node holds <= max_points points
 * (depth capped at 21), one representative per leaf: method 0 = the first point of the leaf (smallest index), 1 = a random
 * point made reproducible (smallest fmix32(index)).  order_out (capacity n) = indices of the kept points in the order upstream
 * leaves the cloud in (depth-first leaf order); leaf_of_out (n entries, may be NULL) = leaf ordinal of every input point (for
 * callers that build centroids / medoids, samplingMethod 2 / 3, themselves). */
icpmi_status icpmi_octree_sample(icpmi_handle h, const float* in4, int64_t n, float max_size, int32_t max_points, int32_t method,
                                 int32_t* order_out, int32_t* leaf_of_out, int64_t* n_out);

/* `Map::updateLocalPointCloud` (Map.cpp:502-534) for a whole module chain on the RESIDENT map: the mapper modules
 * (`mapperModuleVec`, Map.cpp:506-521) and then the post filters (Map.cpp:523-525) run as one program on the device copy
 * of the map; only the scan crosses PCIe.  The device tracks the features, the `normals` and ONE scalar descriptor of
 * the map (the shipped chain's `probabilityDynamic`); for every other descriptor the host applies `src_out`:
 * new map point j was point src_out[j] of [old map (m_old points) ; scan (n points)] -- src_out[j] < m_old: an old map
 * point, else scan point src_out[j] - m_old.
 *   ICPMI_MOP_POINT_DISTANCE   f[0] = minDistNewPoint                   (PointDistanceMapperModule.cpp:28-50)
 *   ICPMI_MOP_DYNAMIC_POINTS   f[0..6] = icpmi_dynpts_params in order   (DynamicPointsMapperModule.cpp:34-172; updates the scalar)
 *   ICPMI_MOP_VOXEL            f[0] = maxSizeByNode, i = samplingMethod 0 | 1 (OctreeMapperModule.cpp:35-39: concatenate, then decimate)
 *   ICPMI_MOP_OCTREE           f[0] = maxSizeByNode, f[1] = maxPointByNode (0 = 1), i = samplingMethod 0 | 1: the octree itself
 *                              (OctreeMapperModule.cpp:8-12,35-39 -> OctreeGridDataPointsFilter: bounding-cube root, split until the
 *                              edge is <= maxSizeByNode or the node holds <= maxPointByNode points, one point per leaf, the map left
 *                              in leaf-visiting order); ICPMI_MOP_VOXEL is the fixed-lattice decimation kept for callers that want it
 *   ICPMI_MOP_SURFACE_NORMALS  i = knn                                  (post filter, examples/config.yaml:26-27)
 *   ICPMI_MOP_CUT_SCALAR       f[0] = threshold, i = useLargerThan      (CutAtDescriptorThresholdDataPointsFilter, config.yaml:29-32)
 * The first n_modules entries are mapper modules: on a handle without a map the first one creates the map from the scan
 * (`createMap`: PointDistance / DynamicPoints take the scan as it is, Voxel decimates it) and the others update it with
 * the same scan (Map.cpp:508-516).  scan_scalar (n floats) is the scan's value of the tracked scalar, NULL if the chain
 * has none; scan_normals3 may be NULL (appended points then carry zero normals until a SURFACE_NORMALS step).
 * to_sensor = pose^-1 (sensor <- map; `pose` is the modules' argument, DynamicPointsMapperModule.cpp:51,57), from_sensor = pose.
 * With from_sensor given the post filters run exactly where the reference runs them: the whole working map is moved into the
 * sensor frame by to_sensor, filtered, and moved back by from_sensor (Map.cpp:523-525) -- every map point picks up the
 * rounding of that round trip on every update, as in the reference (two transform kernels over the map: microseconds).
 * from_sensor == NULL: post filters in the map frame, coordinates of old map points untouched (not what the reference
 * does: a replay of the bundled trajectory then leaves the reference's poses by up to 0.4 mm); to_sensor may be NULL when
 * from_sensor is and the chain has no DYNAMIC_POINTS.
 * src_capacity must be >= m_old + n_modules * n (every module appends at most the whole scan).  identity_prefix (may be NULL): when given, receives the length of the head of
 * the new map that is the untouched head of the old one (src[j] == j for j < *identity_prefix) and src_out is written
 * from that position on only -- a host that owns further descriptors leaves those rows alone and gathers the rest, and
 * an append-only chain downloads a few kilobytes instead of the whole vector. */
typedef enum {
    ICPMI_MOP_POINT_DISTANCE = 0, ICPMI_MOP_DYNAMIC_POINTS = 1, ICPMI_MOP_VOXEL = 2, ICPMI_MOP_SURFACE_NORMALS = 3, ICPMI_MOP_CUT_SCALAR = 4,
    ICPMI_MOP_OCTREE = 5
} icpmi_map_op_type;
typedef struct icpmi_map_op { int32_t type; int32_t i; float f[7]; } icpmi_map_op;
icpmi_status icpmi_map_update_chain(icpmi_handle h, const float* scan4, int64_t n, const float* scan_normals3, const float* scan_scalar,
                                    const float to_sensor[16], const float from_sensor[16], const icpmi_map_op* ops, int32_t n_ops,
                                    int32_t n_modules, int32_t* src_out, int64_t src_capacity, int64_t* identity_prefix, int64_t* new_m);
/* The same on the scan staged by icpmi_register_prior, moved by `correction` first (Mapper.cpp:221). */
icpmi_status icpmi_map_update_chain_staged(icpmi_handle h, const float correction[16], const float* scan_scalar, const float to_sensor[16],
                                           const float from_sensor[16], const icpmi_map_op* ops, int32_t n_ops, int32_t n_modules,
                                           int32_t* src_out, int64_t src_capacity, int64_t* identity_prefix, int64_t* new_m);
/* The tracked scalar descriptor of the resident map: upload after a icpmi_set_map (m must equal the map size),
 * download next to icpmi_get_map (no map, or an empty resident map: ICPMI_OK and nothing written, like icpmi_get_map's 0 points). */
icpmi_status icpmi_set_map_scalar(icpmi_handle h, const float* scalar, int64_t m);
icpmi_status icpmi_get_map_scalar(icpmi_handle h, float* scalar_out, int64_t capacity);

/* Input-side filters as ONE pass (`Mapper::applyInputFilters`, Mapper.cpp:187-191: the DistanceLimit radius filter of
 * Mapper.cpp:27-31 followed by the `input:` chain, e.g. the two BoundingBox filters of examples/config.yaml:2-18): every
 * filter of these two kinds is a per-point predicate, so a run of them is one fused kernel and one compaction instead
 * of one pass + one copy of every descriptor per filter.  keep[i] = 1 iff point i passes ALL filters.
 *   ICPMI_FILT_DISTANCE_LIMIT  i = dim (-1: radial distance, 0..2: |coordinate|), f[0] = dist, f[1] = removeInside (0 | 1):
 *                              v = dim < 0 ? sqrt(x^2 + y^2 + z^2) : |p[dim]|; kept iff removeInside ? v > |dist| : v < |dist|
 *   ICPMI_FILT_BOUNDING_BOX    f[0..2] = xMin, yMin, zMin, f[3..5] = xMax, yMax, zMax, i = removeInside:
 *                              inside iff min < p < max on all three axes; kept iff removeInside ? !inside : inside */
typedef enum { ICPMI_FILT_DISTANCE_LIMIT = 0, ICPMI_FILT_BOUNDING_BOX = 1 } icpmi_point_filter_type;
typedef struct icpmi_point_filter { int32_t type; int32_t i; float f[6]; } icpmi_point_filter;
icpmi_status icpmi_filter_points(icpmi_handle h, const float* in4, int64_t n, const icpmi_point_filter* filters, int32_t n_filters,
                                 uint8_t* keep);

/* `Map::unloadCells` binning (Map.cpp:206-209,232-235): ijk3[3 i + r] = floor(p_r / cell_size). */
icpmi_status icpmi_bin_cells(icpmi_handle h, const float* pts4, int64_t n, float cell_size, int32_t* ijk3);

/* ---- multi-GPU: scan-sharded mapping (BASELINE config 5) ----
 * One process and one handle per GPU, the map replicated, every rank registering its own scan stream; the one exchange is map
 * growth: the points every rank accepted are all-gathered over RCCL (xGMI) so that all replicas append the same set -- what a
 * host then bins into 20 m cells for its CellManager (Map.cpp:206-229).  No reference analogue (the reference is one process).
 *   icpmi_comm_get_unique_id  rank 0 creates the id (an ncclUniqueId); the application hands it to the other ranks
 *   icpmi_comm_init           collective over all ranks: ncclCommInitRank on the handle's device
 *   icpmi_staged_merge_allgather  one map-growth epoch, entirely on the device: the scan staged by icpmi_register_prior is moved
 *       by `correction` (Mapper.cpp:221); the points at least min_dist from the resident map are compacted
 *       (PointDistanceMapperModule.cpp:33-42); counts, then the padded point blocks, are all-gathered on the handle's stream;
 *       the blocks are merged in rank order, block r keeping only the points that are at least min_dist from the points
 *       accepted from ranks < r (exactly what one mapper would have appended had it processed the scans in rank order);
 *       the merged set is appended to the resident map, its normals recomputed (normals_knn > 0) and the index rebuilt.
 *       No accepted point crosses PCIe unless merged_out4 (capacity merged_capacity points) asks for the merged set.
 *       Without a communicator the handle is its own single rank.
 *       Collective discipline (v3): the call is a collective -- EVERY rank of the communicator must make it once per epoch, also a
 *       rank that has nothing to add: `correction == NULL` (or nothing staged, see icpmi_stage_discard) contributes zero points and
 *       still receives and appends what the others accepted.  A rank whose local part fails reports count -1 and ALL ranks return an
 *       error together without appending; buffers are grown to the gathered sizes behind a second one-word exchange, so no rank can
 *       leave between two collectives.  The merged set is always appended: when merged_capacity is smaller than the merged set,
 *       merged_out4 receives the first merged_capacity points, *merged_n the full count, the status stays ICPMI_OK and
 *       icpmi_staged_merged_points returns the whole set -- replicas never diverge over a host buffer.
 *   icpmi_staged_merged_points  the merged set of the last epoch (out4 == NULL: only *n), kept on the device until the next epoch
 *   icpmi_staged_bin_cells (r6)  the merged set of the last epoch binned into cubic cells of edge cell_size ON THE DEVICE -- Map.cpp:206-229's
 *       per-point loop (cell = floor(coordinate / cell_size) per axis, Map.cpp:472-480) feeding RAMCellManager.cpp:13-16 -- and appended,
 *       cell after cell (cells in the order of their first point, the points of a cell in merged order: what Map::binIntoCells yields on
 *       the host), to the handle's device-resident CELL LOG.  Returned per cell r < *n_cells: ijk3[3 r ..], counts[r] and offsets[r], the
 *       position of the cell's run in the log.  Only this table crosses PCIe.  Once per epoch (a second call for the same epoch is an
 *       ICPMI_ERR_INVALID_ARG); capacity < *n_cells: ICPMI_ERR_INVALID_ARG with *n_cells set and nothing appended; more than 4096 cells
 *       touched by one epoch: ICPMI_ERR_UNSUPPORTED (fetch icpmi_staged_merged_points and bin on the host).
 *   icpmi_cell_log_configure     cell_size > 0: from now on every epoch enqueues that binning itself, between its merge and its append (the kernels
 *                                run in the shadow of the index insert); icpmi_staged_bin_cells with the same cell_size then only collects the
 *                                table.  0: off (the default)
 *   icpmi_cell_log_read          `count` points of the log from `offset` (out4 == NULL: only *log_size, the points in the log)
 *   icpmi_cell_log_clear         forgets the log (CellManager::clearAllCells)
 *   icpmi_stage_discard          drops the scan staged by icpmi_register_prior (e.g. after its registration failed: the rank then
 *                                takes part in the epoch empty-handed and reports its own error afterwards) */
typedef struct icpmi_comm_id { char bytes[128]; } icpmi_comm_id;
icpmi_status icpmi_comm_get_unique_id(icpmi_comm_id* id);
icpmi_status icpmi_comm_init(icpmi_handle h, const icpmi_comm_id* id, int32_t n_ranks, int32_t rank);
icpmi_status icpmi_comm_destroy(icpmi_handle h);
/* What the handle's communicator reports about itself (RCCL: ncclCommCount / ncclCommUserRank); kind 0 = none (the handle is its own
 * single rank), 1 = RCCL, 2 = the loopback test communicator.  Any pointer may be NULL. */
icpmi_status icpmi_comm_info(icpmi_handle h, int32_t* n_ranks, int32_t* rank, int32_t* kind);
icpmi_status icpmi_staged_merge_allgather(icpmi_handle h, const float correction[16], float min_dist, int32_t normals_knn,
                                          int64_t* accepted_local, int64_t* appended_total, int64_t* new_m, float* merged_out4,
                                          int64_t merged_capacity, int64_t* merged_n);
icpmi_status icpmi_staged_merged_points(icpmi_handle h, float* out4, int64_t capacity, int64_t* n);
icpmi_status icpmi_staged_bin_cells(icpmi_handle h, float cell_size, int32_t* ijk3, int64_t* offsets, int64_t* counts, int64_t capacity,
                                    int64_t* n_cells);
icpmi_status icpmi_cell_log_configure(icpmi_handle h, float cell_size);
icpmi_status icpmi_cell_log_read(icpmi_handle h, int64_t offset, int64_t count, float* out4, int64_t* log_size);
icpmi_status icpmi_cell_log_clear(icpmi_handle h);
icpmi_status icpmi_stage_discard(icpmi_handle h);

/* ---- plumbing ---- */
/* Use an externally owned hipStream_t (e.g. torch's current stream) instead of the handle's own. */
icpmi_status icpmi_set_stream(icpmi_handle h, void* hip_stream);
/* Grid parameters chosen by the last set_map: cell edge, dims[3], number of cells (for DESIGN/bench). */
icpmi_status icpmi_get_grid_info(icpmi_handle h, float* cell, int32_t dims[3], int64_t* n_cells, int64_t* n_occupied);
int32_t      icpmi_version(void);
/* "icpmi <version> src:<stamp>": <stamp> = the first 16 hex digits of the SHA-256 over the library's sources (csrc Makefile: sorted *.hip, common.h,
 * solve.h, include/icpmi.h) as they were when this binary was linked -- tests/conftest.py recomputes it from the tree and refuses to run GPU
 * tests against a binary built from other sources. */
const char*  icpmi_build_info(void);
/* The library caches freed device blocks per device (at most ICPMI_ALLOC_CACHE_MB, default 1024) so that a mapper that is dropped and rebuilt
 * finds its ~300 arrays again.  A host that shares the GPU with another allocator calls this when it is done with a mapper: every cached
 * block goes back to the runtime now; live handles stay valid. */
icpmi_status icpmi_trim_cache(void);
/* Diagnostics of the last registration (engine internals, not part of the reference surface).  Slots 12 / 13 (r5): iterations of a k > 1 loop whose
   quantile selection took its level 0 from the NN kernel's window / from the full histogram behind a window that missed. */
icpmi_status icpmi_debug_counters(icpmi_handle h, uint64_t out[24]);
/* Test seam: the n-th value (n >= 1) of the std::minstd_rand stream as the DEVICE computes it by skip-ahead (csrc/ssn.hip, behind
 * SamplingSurfaceNormalDataPointsFilter -- PM::ICPSequence::setDefault(), Mapper.cpp:74-78).  [rand.predef]: seed 1, n = 10 000 -> 399268537. */
icpmi_status icpmi_debug_minstd_nth(icpmi_handle h, uint32_t seed, uint32_t n, uint32_t* out);

#ifdef __cplusplus
}
#endif
#endif
