"""CPU tests (no GPU): host-side logic, the synthetic workload, and that the C-ABI library loads and
exports every symbol include/icpmi.h declares (no compute calls without a GPU)."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "icpmi.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(icpmi_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from norlab_icp_mapper_amd import _capi
    if not os.path.exists(_capi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _capi.load()
    declared = _header_symbols()
    assert len(declared) >= 20
    bound = {name for name, _, _ in _capi.SYMBOLS}
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in icpmi.h but not exported"
        assert name in bound, f"{name} declared in icpmi.h but missing from the ctypes table"
    assert lib.icpmi_version() == 4


def test_config_defaults_and_struct_layout():
    from norlab_icp_mapper_amd import _capi
    lib = _capi.load()
    cfg = _capi.Config()
    lib.icpmi_config_default(C.byref(cfg))
    assert cfg.knn == 1 and math.isinf(cfg.max_dist) and cfg.epsilon == 0
    assert cfg.minimizer == _capi.MIN_POINT_TO_PLANE and cfg.max_iterations == 40
    assert cfg.min_diff_rot == pytest.approx(1e-3) and cfg.smooth_length == 3 and cfg.use_graph == 1
    # the ctypes mirrors must have the C layout: 8-byte aligned int64 members, trailing reserved block
    assert C.sizeof(_capi.Stats) == 72
    assert C.sizeof(_capi.Config) == 5 * 4 + 8 * 20 + 15 * 4 + 8 * 4
    assert cfg.force_4dof == 0


def test_operator_structs_match_the_header(tmp_path):
    """icpmi_map_op / icpmi_point_filter / icpmi_dynpts_params as gcc lays them out from include/icpmi.h
    against the ctypes mirrors the tests and the bench bind."""
    import subprocess
    from norlab_icp_mapper_amd import _capi
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "icpmi.h"\nint main(void) { printf("%zu %zu %zu %zu %zu %zu\\n", '
                   'sizeof(icpmi_map_op), offsetof(icpmi_map_op, f), sizeof(icpmi_point_filter), offsetof(icpmi_point_filter, f), '
                   'sizeof(icpmi_dynpts_params), sizeof(icpmi_stats)); return 0; }\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert got[0] == C.sizeof(_capi.MapOp) and got[1] == _capi.MapOp.f.offset
    assert got[2] == C.sizeof(_capi.PointFilter) and got[3] == _capi.PointFilter.f.offset
    assert got[4] == 7 * 4 and got[5] == C.sizeof(_capi.Stats)
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "icpmi.h"\nint main(void) { printf("%zu %zu %zu %zu %zu\\n", '
                   'sizeof(icpmi_config), sizeof(icpmi_outlier), offsetof(icpmi_config, outlier), offsetof(icpmi_config, force_4dof), '
                   'offsetof(icpmi_config, reserved)); return 0; }\n')
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert got == [C.sizeof(_capi.Config), C.sizeof(_capi.Outlier), _capi.Config.outlier.offset, _capi.Config.force_4dof.offset,
                   _capi.Config.reserved.offset]


def test_create_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import norlab_icp_mapper_amd as pkg
    with pytest.raises(pkg.icp.HipError) as e:
        pkg.ICPSequence(minimizer=1)
    assert "no HIP device" in str(e.value)
    with pytest.raises(pkg.InvalidParameter):
        pkg.ICPSequence(knn=0)


def test_roctx_ranges_are_optional_and_harmless():
    """ICPMI_ROCTX=1 looks the marker library up at run time; entry points behave the same with and without it"""
    import subprocess, sys
    code = ("import ctypes as C; lib = C.CDLL(%r); T = (C.c_float * 16)(); "
            "print(lib.icpmi_register(None, None, 0, None, T, None), lib.icpmi_has_map(None))" % os.path.join(ROOT, "norlab_icp_mapper_amd", "libicpmi.so"))
    outs = [subprocess.check_output([sys.executable, "-c", code], env=dict(os.environ, ICPMI_ROCTX=v)).split() for v in ("0", "1")]
    assert outs[0] == outs[1] and int(outs[0][0]) != 0


def test_yaml_chain_translation():
    import yaml
    import norlab_icp_mapper_amd as pkg
    from norlab_icp_mapper_amd import _capi
    chain = yaml.safe_load("""
matcher:
  KDTreeMatcher:
    knn: 6
    maxDist: 2.0
    epsilon: 1
outlierFilters:
  - TrimmedDistOutlierFilter:
      ratio: 0.9
  - MaxDistOutlierFilter:
      maxDist: 1.5
errorMinimizer:
  PointToPlaneErrorMinimizer:
transformationCheckers:
  - CounterTransformationChecker:
      maxIterationCount: 25
  - DifferentialTransformationChecker:
      minDiffRotErr: 0.002
      minDiffTransErr: 0.01
      smoothLength: 4
inspector: NullInspector
""")
    cfg = pkg.config_from_yaml_chain(chain)
    assert (cfg.knn, cfg.max_dist, cfg.epsilon) == (6, 2.0, 1.0)
    assert cfg.n_outlier == 2 and cfg.outlier[0].type == _capi.OUT_TRIMMEDDIST and cfg.outlier[0].param == pytest.approx(0.9)
    assert cfg.outlier[1].type == _capi.OUT_MAXDIST and cfg.outlier[1].param == 1.5
    assert cfg.minimizer == _capi.MIN_POINT_TO_PLANE and cfg.max_iterations == 25
    assert cfg.use_differential == 1 and cfg.smooth_length == 4 and cfg.min_diff_trans == pytest.approx(0.01)
    # the bundled example's chain (examples/config.yaml:55-68)
    bundled = yaml.safe_load("""
matcher:
  KDTreeMatcher: {knn: 6, maxDist: 2.0, epsilon: 1}
errorMinimizer:
  IdentityErrorMinimizer:
transformationCheckers:
  - CounterTransformationChecker: {maxIterationCount: 10}
inspector: NullInspector
""")
    cfg = pkg.config_from_yaml_chain(bundled)
    assert cfg.minimizer == _capi.MIN_IDENTITY and cfg.max_iterations == 10 and cfg.n_outlier == 0
    # defaults when sections are missing
    cfg = pkg.config_from_yaml_chain({})
    assert cfg.knn == 1 and math.isinf(cfg.max_dist) and cfg.minimizer == _capi.MIN_POINT_TO_PLANE and cfg.max_iterations == 40
    for bad in ({"matcher": {"OctreeMatcher": {}}}, {"matcher": {"KDTreeMatcher": {"nope": 1}}},
                {"outlierFilters": [{"FooFilter": {}}]}, {"errorMinimizer": "FooMinimizer"},
                {"transformationCheckers": [{"FooChecker": {}}]}, {"icp": {}}):
        with pytest.raises(pkg.InvalidParameter):
            pkg.config_from_yaml_chain(bad)
    assert pkg.config_from_yaml_chain({"errorMinimizer": {"PointToPlaneErrorMinimizer": {"force2D": 1}}}).force_2d == 1
    with pytest.raises(pkg.InvalidParameter):
        pkg.config_from_yaml_chain({"errorMinimizer": {"PointToPlaneErrorMinimizer": {"force2D": 1, "force4DOF": 1}}})
    # the chain elements beyond the bundled configurations
    cfg = pkg.config_from_yaml_chain({
        "outlierFilters": [{"RobustOutlierFilter": {"robustFct": "huber", "tuning": 1.5, "scaleEstimator": "mad", "nbIterationForScale": 4,
                                                    "distanceType": "point2plane"}},
                           {"GenericDescriptorOutlierFilter": {"descName": "probabilityDynamic", "useLargerThan": 0, "threshold": 0.6}},
                           "RobustOutlierFilter", {"VarTrimmedDistOutlierFilter": {"minRatio": 0.1, "lambda": 2.0}}],
        "errorMinimizer": {"PointToPlaneErrorMinimizer": {"force4DOF": 1}}})
    assert cfg.force_4dof == 1 and cfg.n_outlier == 4
    assert (cfg.outlier[3].type, cfg.outlier[3].param3) == (_capi.OUT_VARTRIMMEDDIST, 2.0)
    assert cfg.outlier[3].param == pytest.approx(0.1) and cfg.outlier[3].param2 == pytest.approx(0.99)
    o = cfg.outlier
    assert (o[0].type, o[0].param, o[0].iparam, o[0].param2) == (_capi.OUT_ROBUST, 1.5, 5 | (1 << 4) | (1 << 8), 4.0)
    assert (o[1].type, o[1].iparam) == (_capi.OUT_GENERICDESCRIPTOR, 0) and o[1].param == pytest.approx(0.6)
    assert (o[2].type, o[2].param, o[2].iparam, o[2].param2) == (_capi.OUT_ROBUST, 1.0, 0 | (1 << 4), 0.0)  # upstream's defaults
    assert pkg.config_from_yaml_chain({"errorMinimizer": {"PointToPointErrorMinimizer": {}}}).force_4dof == 0
    for bad in ({"RobustOutlierFilter": {"robustFct": "foo"}}, {"RobustOutlierFilter": {"nope": 1}}, {"GenericDescriptorOutlierFilter": {"nope": 1}}):
        with pytest.raises(pkg.InvalidParameter):
            pkg.config_from_yaml_chain({"outlierFilters": [bad]})
    rd = pkg.config_from_yaml_chain({"outlierFilters": [{"GenericDescriptorOutlierFilter": {"source": "reading", "descName": "intensity", "threshold": 0.3}}]})
    assert rd.outlier[0].iparam == _capi.GEN_SOURCE_READING | _capi.GEN_LARGER      # r4: served (icpmi_set_reading_scalar)
    # r5: berg / std scale estimators and `approximation` are served
    rb = pkg.config_from_yaml_chain({"outlierFilters": [{"RobustOutlierFilter": {"scaleEstimator": "berg", "tuning": 0.05, "approximation": 2.0}},
                                                        {"RobustOutlierFilter": {"scaleEstimator": "std", "robustFct": "tukey"}}]})
    assert (rb.outlier[0].iparam, rb.outlier[0].param3, rb.outlier[0].param) == (0 | (2 << 4), 2.0, pytest.approx(0.05))
    assert rb.outlier[1].iparam == 4 | (3 << 4) and math.isinf(rb.outlier[1].param3)
    for bad in ({"RobustOutlierFilter": {"scaleEstimator": "foo"}}, {"RobustOutlierFilter": {"approximation": 0.0}}, {"RobustOutlierFilter": {"approximation": -1.0}}):
        with pytest.raises(pkg.InvalidParameter):
            pkg.config_from_yaml_chain({"outlierFilters": [bad]})


def test_synthetic_scene_is_deterministic_and_well_formed():
    from norlab_icp_mapper_amd import synth
    a = synth.make_scene(m=5000, n=700)
    b = synth.make_scene(m=5000, n=700)
    for k in ("map", "normals", "scan", "scan_normals"):
        assert np.array_equal(a[k], b[k])
    assert a["map"].shape == (5000, 4) and a["map"].dtype == np.float32 and (a["map"][:, 3] == 1).all()
    assert a["scan"].shape == (700, 4)
    np.testing.assert_allclose(np.linalg.norm(a["normals"], axis=1), 1.0, atol=1e-6)
    # the room: x, y in [-50, 50], z in [0, 10] (+ noise)
    assert np.abs(a["map"][:, :2]).max() < 50.1 and a["map"][:, 2].min() > -0.1 and a["map"][:, 2].max() < 10.1
    # scan = surfaces within 60 m of the sensor, moved by T_gt^-1
    T = a["T_gt"]
    back = a["scan"][:, :3].astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    assert np.linalg.norm(back - np.array([3.0, -2.0, 1.5]), axis=1).max() < 60.1
    # counter-based generator: known first outputs of splitmix64
    assert int(synth.splitmix64(np.uint64(0))) == 0xE220A8397B1DCDAF
    u = synth.uniform(42, 0, 4)
    assert ((0 <= u) & (u < 1)).all() and len(set(u.tolist())) == 4
    dt, dr = synth.pose_error(synth.make_T((0.01, 0, 0), (0.1, 0, 0)), np.eye(4))
    assert dt == pytest.approx(0.1) and dr == pytest.approx(0.01, rel=1e-6)


def test_bundled_fixture_known_answer():
    """examples/data as shipped: 14 trajectory rows; lexicographic scan order puts
    cloud_1690309710_85582848.vtk last, so rows and scans are mis-paired by design."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "bundled_scans.npz"))
    names = g["scan_names"].tolist()
    assert len(names) == 14 and names == sorted(names) and names[-1] == "cloud_1690309710_85582848.vtk"
    assert g["scan0_xyz"].shape == (41400, 3) and g["trajectory"].shape == (14, 9)
    t = g["trajectory"]
    stamps = t[:, 0] + 1e-9 * t[:, 1]
    assert (np.diff(stamps) > 0.09).all() and (np.diff(stamps) < 0.11).all()  # 10 Hz
    np.testing.assert_allclose(np.linalg.norm(t[:, 5:9], axis=1), 1.0, atol=1e-9)
