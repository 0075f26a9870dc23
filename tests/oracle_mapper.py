"""Oracle-side restatement of `Mapper::processInput` / `Map::updateLocalPointCloud` (TEST INFRASTRUCTURE).

Composes the CPU oracle's single operators (oracle/icp_oracle.c through tests/oracle_bindings.py -- never a kernel of
libicpmi.so) exactly as the reference composes libpointmatcher's, so that the C++ host shell + HIP path can be checked
scan by scan against an independent replay of a whole trajectory:

    applyInputFilters      norlab_icp_mapper/Mapper.cpp:187-191   radius filter (Mapper.cpp:27-31), then the `input:` chain
    processInput           Mapper.cpp:194-238                    prior -> icp -> correction * prior -> policy -> update
    shouldUpdateMap        Mapper.cpp:240-272                    overlap / delay / distance
    updateLocalPointCloud  Map.cpp:502-534                       first scan: module 0 creates, the rest update (:505-515);
                                                                 to the sensor frame, post filters, back, icp.setMap (:523-528)
    module bodies          MapperModules/{PointDistance,Octree,DynamicPoints}MapperModule.cpp

    updatePose             Map.cpp:246-460                       the sliding window of loaded 20 m cells (`paging=True`): window
                                                                 edges ceil((p - R)/20 - 1) / floor((p + R)/20) (:472-480), moved when
                                                                 an edge is 2 cells off (:278...), BUFFER_SIZE 2; loadCells :71-128,
                                                                 unloadCells :140-230, RAMCellManager.cpp:3-31; getMap :552-573

Clouds are dicts {"xyz1": (n, 4) float32, <descriptor name>: (n, span) float32}; `concatenate` keeps the descriptors
both clouds have (PM::DataPoints::concatenate).
"""
import numpy as np

import oracle_bindings as ob


def _h(xyz):
    out = np.ones((xyz.shape[0], 4), dtype=np.float32)
    out[:, :3] = xyz[:, :3]
    return out


def keep_only(cloud, mask):
    return {k: np.ascontiguousarray(v[mask]) for k, v in cloud.items()}


def gather(cloud, order):
    """the cloud re-ordered / decimated by an index list (OctreeGridDataPointsFilter leaves it in leaf-visiting order)"""
    return {k: np.ascontiguousarray(v[order]) for k, v in cloud.items()}


def concatenate(a, b):
    """PM::DataPoints::concatenate: features stacked, descriptors present in both kept."""
    out = {"xyz1": np.concatenate([a["xyz1"], b["xyz1"]], 0)}
    for k in a:
        if k != "xyz1" and k in b:
            out[k] = np.concatenate([a[k], b[k]], 0)
    return out


def transform_cloud(T, cloud):
    """RigidTransformation::compute: features by T, `normals` / `observationDirections` by R, the rest copied."""
    out = dict(cloud)
    out["xyz1"] = ob.transform(T, cloud["xyz1"])
    for name in ("normals", "observationDirections"):
        if name in cloud:
            out[name] = ob.rotate3(T, cloud[name])
    return out


def mat4_mul_f32(A, B):
    """4x4 float product as the host shell forms it (PointCloud.h Mat4::operator*: s = 0; s += a(i,k) * b(k,j), k = 0..3,
    plain float multiply and add) -- the corrected pose feeds pose^-1 and the sensor-frame round trip of the whole map,
    where the last bit matters."""
    A = np.asarray(A, dtype=np.float32); B = np.asarray(B, dtype=np.float32)
    R = np.zeros((4, 4), dtype=np.float32)
    for i in range(4):
        for j in range(4):
            s = np.float32(0.0)
            for k in range(4):
                s = np.float32(s + np.float32(A[i, k] * B[k, j]))
            R[i, j] = s
    return R


class OracleMapper:
    def __init__(self, icp_kw, modules, post=(), update=("distance", 1.0), sensor_max_range=80.0, input_filters=(),
                 add_descriptors=(), nthreads=8, post_in_map_frame=False, paging=False):
        """modules: [("point_distance", minDist) | ("dynamic_points", {params}) | ("octree", maxSize, samplingMethod[, maxPointByNode])]
        post: [("surface_normals", knn) | ("cut", descName, useLargerThan, threshold)]
        input_filters: oracle_bindings.filter_points rows; add_descriptors: [(name, value)] (AddDescriptorDataPointsFilter)"""
        self.icp = ob.OracleICP(ob.make_config(nthreads=nthreads, **icp_kw))
        self.modules, self.post, self.update = list(modules), list(post), update
        self.sensor_max_range, self.input_filters, self.add_descriptors = sensor_max_range, list(input_filters), list(add_descriptors)
        self.nthreads = nthreads
        # True: the post filters see the cloud in the map frame (what the resident device path does: no rotation of the whole
        # map into the sensor frame and back on every update); False: the reference's Map.cpp:523-525
        self.post_in_map_frame = post_in_map_frame
        self.map = None
        self.paging = paging
        self.cells, self.loaded, self.first_pose_update = {}, set(), True      # RAMCellManager, loadedCellIds, firstPoseUpdate
        self.win = None                                                        # [inferior, superior] LastUpdateIndex per axis
        self.page_events, self.reloaded_points = [], 0
        self.pose = np.eye(4, dtype=np.float32)
        self.trajectory, self.iterations, self.updated = [], [], []
        self.last_time, self.last_pose = None, None

    # ---- Mapper::applyInputFilters ----
    def apply_input_filters(self, xyz, extra=None):
        cloud = {"xyz1": _h(xyz)}
        if extra:
            cloud.update({k: np.asarray(v, dtype=np.float32).reshape(xyz.shape[0], -1) for k, v in extra.items()})
        filters = [("distance_limit", -1, float(self.sensor_max_range), 0)] + self.input_filters
        cloud = keep_only(cloud, ob.filter_points(cloud["xyz1"], filters))
        for name, value in self.add_descriptors:
            cloud[name] = np.full((cloud["xyz1"].shape[0], 1), value, dtype=np.float32)
        return cloud

    # ---- the module chain ----
    def _module_update(self, mod, inp, mp, pose):
        kind = mod[0]
        if kind == "point_distance":                       # PointDistanceMapperModule.cpp:28-50
            keep = ob.point_distance_keep(mp["xyz1"], inp["xyz1"], mod[1], nthreads=self.nthreads)
            return concatenate(mp, keep_only(inp, keep))
        if kind == "octree":                               # OctreeMapperModule.cpp:35-39: concatenate, then decimate
            both = concatenate(mp, inp)
            return gather(both, ob.octree_sample(both["xyz1"], mod[1], mod[3] if len(mod) > 3 else 1, mod[2]))
        if kind == "dynamic_points":                       # DynamicPointsMapperModule.cpp:34-151
            to_sensor = np.linalg.inv(pose.astype(np.float64)).astype(np.float32)
            out = dict(mp)
            out["probabilityDynamic"] = ob.dynamic_points_update(to_sensor, inp["xyz1"], mp["xyz1"], mp["normals"], mp["probabilityDynamic"][:, 0],
                                                                 nthreads=self.nthreads, **mod[1])[:, None]
            return out
        raise ValueError(kind)

    def _module_create(self, mod, inp, pose):
        if mod[0] in ("point_distance", "dynamic_points"):  # createMap keeps the input untouched (.cpp:8-21 / :14-27)
            return dict(inp)
        if mod[0] == "octree":                              # OctreeMapperModule.cpp:15-27: update of an empty map
            return gather(inp, ob.octree_sample(inp["xyz1"], mod[1], mod[3] if len(mod) > 3 else 1, mod[2]))
        raise ValueError(mod[0])

    def update_local_point_cloud(self, inp, pose):
        if self.map is None or self.map["xyz1"].shape[0] == 0:
            mp = self._module_create(self.modules[0], inp, pose)
            for mod in self.modules[1:]:
                mp = self._module_update(mod, inp, mp, pose)
        else:
            mp = self.map
            for mod in self.modules:
                mp = self._module_update(mod, inp, mp, pose)
        inv = np.linalg.inv(pose.astype(np.float64)).astype(np.float32)
        sensor = dict(mp) if self.post_in_map_frame else transform_cloud(inv, mp)
        for f in self.post:
            if f[0] == "surface_normals":
                sensor["normals"] = ob.surface_normals(sensor["xyz1"], knn=f[1], nthreads=self.nthreads)
            elif f[0] == "cut":
                v = sensor[f[1]][:, 0]
                sensor = keep_only(sensor, ~(v > f[3]) if f[2] else ~(v < f[3]))
            else:
                raise ValueError(f[0])
        self.map = sensor if self.post_in_map_frame else transform_cloud(pose, sensor)
        self.icp.setMap(self.map["xyz1"], self.map.get("normals"))

    # ---- Map::updatePose and the cell manager ----
    CELL, BUF = np.float32(20.0), 2

    def _inf(self, w):
        return int(np.ceil(np.float32(np.float32(np.float32(w) - np.float32(self.sensor_max_range)) / self.CELL) - 1.0))

    def _sup(self, w):
        return int(np.floor(np.float32(np.float32(w) + np.float32(self.sensor_max_range)) / self.CELL))

    def _set_icp_map(self):
        if self.map is not None and self.map["xyz1"].shape[0]:
            self.icp.setMap(self.map["xyz1"], self.map.get("normals"))            # an empty cloud is ignored (setMap returns false)

    def _unload(self, lo, hi):
        """Map.cpp:140-230: points inside the half-open box of the cell range leave the local cloud and are saved per cell"""
        if self.map is not None and self.map["xyz1"].shape[0]:
            p = self.map["xyz1"][:, :3]
            inside = np.ones(p.shape[0], bool)
            for a in range(3):
                start = -np.inf if lo[a] is None else np.float32(lo[a]) * self.CELL
                end = np.inf if hi[a] is None else np.float32(hi[a] + 1) * self.CELL
                inside &= (p[:, a] >= start) & (p[:, a] < end)
            old = keep_only(self.map, inside)
            self.map = keep_only(self.map, ~inside)
            self._set_icp_map()
            ijk = np.floor(old["xyz1"][:, :3] / self.CELL).astype(np.int64)
            ids = [f"{i}_{j}_{k}" for i, j, k in ijk]
            order = {}
            for row, cid in enumerate(ids):
                order.setdefault(cid, []).append(row)
            for cid, rows in order.items():
                self.cells[cid] = gather(old, np.array(rows))                  # saveCell overwrites
        if lo[0] is None:
            self.loaded.clear()
        else:
            for i in range(lo[0], hi[0] + 1):
                for j in range(lo[1], hi[1] + 1):
                    for k in range(lo[2], hi[2] + 1):
                        self.loaded.discard(f"{i}_{j}_{k}")

    def _load(self, lo, hi):
        """Map.cpp:71-128"""
        chunk = None
        for i in range(lo[0], hi[0] + 1):
            for j in range(lo[1], hi[1] + 1):
                for k in range(lo[2], hi[2] + 1):
                    cell = self.cells.get(f"{i}_{j}_{k}")
                    if cell is not None and cell["xyz1"].shape[0]:
                        chunk = cell if chunk is None else concatenate(chunk, cell)
                    self.loaded.add(f"{i}_{j}_{k}")
        if chunk is not None:
            self.reloaded_points += int(chunk["xyz1"].shape[0])
            self.map = chunk if self.map is None or self.map["xyz1"].shape[0] == 0 and not self.map_has_fields() else concatenate(self.map, chunk)
            self._set_icp_map()

    def map_has_fields(self):
        return self.map is not None and len(self.map) > 1

    def update_pose(self, pose):
        if not self.paging:
            return
        pos = [float(pose[a, 3]) for a in range(3)]
        B = self.BUF
        if self.first_pose_update:
            self.win = [[self._inf(pos[a]), self._sup(pos[a])] for a in range(3)]
            self.cells.clear(); self.loaded.clear()
            self._unload([None] * 3, [None] * 3)
            self._load([self.win[a][0] - B for a in range(3)], [self.win[a][1] + B for a in range(3)])
            self.first_pose_update = False
            return
        for a in range(3):                                                      # rows, columns, aisles -- in that order (Map.cpp:276-457)
            def box(start, end):
                lo = [self.win[b][0] - B for b in range(3)]; hi = [self.win[b][1] + B for b in range(3)]
                lo[a], hi[a] = start, end
                return lo, hi
            inf_now, sup_now = self._inf(pos[a]), self._sup(pos[a])
            inf_last, sup_last = self.win[a]
            if abs(inf_now - inf_last) >= 2:                                    # the trailing edge
                if inf_now < inf_last:
                    n = inf_last - inf_now
                    lo, hi = box(inf_now - B, inf_now - B + n - 1); self._load(lo, hi); self.page_events.append(("load", a, lo[a], hi[a], len(self.trajectory)))
                if inf_now > inf_last:
                    n = inf_now - inf_last
                    lo, hi = box(inf_last - B, inf_last - B + n - 1); self._unload(lo, hi); self.page_events.append(("unload", a, lo[a], hi[a], len(self.trajectory)))
                self.win[a][0] = inf_now
            if abs(sup_now - sup_last) >= 2:                                    # the leading edge
                if sup_now < sup_last:
                    n = sup_last - sup_now
                    lo, hi = box(sup_last + B - n + 1, sup_last + B); self._unload(lo, hi); self.page_events.append(("unload", a, lo[a], hi[a], len(self.trajectory)))
                if sup_now > sup_last:
                    n = sup_now - sup_last
                    lo, hi = box(sup_now + B - n + 1, sup_now + B); self._load(lo, hi); self.page_events.append(("load", a, lo[a], hi[a], len(self.trajectory)))
                self.win[a][1] = sup_now

    def set_map(self, cloud):
        """Mapper::setMap (Mapper.cpp:295-301) -> Map::setGlobalPointCloud (Map.cpp:575-588): the cloud replaces the local map, the
        matcher is rebuilt, the next updatePose pages it into cells from scratch, the trajectory restarts"""
        self.map = {k: v.copy() for k, v in cloud.items()}
        self._set_icp_map()
        self.first_pose_update = True
        self.trajectory = []

    def get_map(self):
        """Mapper::getMap (Map.cpp:552-573): the local cloud plus every saved cell that is not loaded"""
        out = self.map
        for cid, cell in self.cells.items():
            if cid not in self.loaded:
                out = cell if out is None else concatenate(out, cell)
        return out

    # ---- Mapper::shouldUpdateMap ----
    def _should_update(self, stamp, pose, overlap):
        kind, value = self.update
        if kind == "overlap":
            return overlap < value
        if kind == "delay":
            return (stamp - self.last_time) > np.float32(value)
        d = pose[:3, 3].astype(np.float32) - self.last_pose[:3, 3].astype(np.float32)
        return float(np.sqrt(np.float32(np.dot(d, d)))) > value

    # ---- Mapper::processInput ----
    def process_input(self, filtered, prior, stamp_s):
        prior = np.asarray(prior, dtype=np.float32)
        inp = transform_cloud(prior, filtered)
        if self.map is None or self.map["xyz1"].shape[0] == 0:
            corrected = prior
            self.update_pose(corrected)
            self.last_time, self.last_pose = stamp_s, corrected
            self.update_local_point_cloud(inp, corrected)
            self.iterations.append(0); self.updated.append(True)
        else:
            err, corr = self.icp(inp["xyz1"], inp.get("normals"))
            if err:
                raise RuntimeError(f"oracle ICP error {err}")
            corrected = mat4_mul_f32(corr, prior)
            self.iterations.append(self.icp.stats.iterations)
            self.update_pose(corrected)
            upd = self._should_update(stamp_s, corrected, self.icp.stats.weighted_point_used_ratio)
            if upd:
                self.last_time, self.last_pose = stamp_s, corrected
                self.update_local_point_cloud(transform_cloud(corr, inp), corrected)
            self.updated.append(bool(upd))
        self.pose = corrected
        self.trajectory.append(corrected)
        return corrected
