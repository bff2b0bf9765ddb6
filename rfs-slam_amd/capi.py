"""ctypes binding of the C ABI in include/rfsgpu.h.

The binding is prefix-parametrised (`rfsgpu_` for the product library) so that test infrastructure can
drive a second library exposing the same call shapes through identical Python calls.  Nothing here loads
anything but the library it is handed.

Mirrors the public surface of rfs::RBPHDFilter (reference include/RBPHDFilter.hpp:72-251): method
names follow the reference (predict / update / getGMSize / getLandmark / ...), error behaviour too
(getGMSize -> -1, getLandmark -> None on bad indices, update/predict raise only on engine failure).
"""
import ctypes as C
import numpy as np

OK = 0
ERR_INVALID, ERR_HIP, ERR_CAPACITY, ERR_NO_DEVICE, ERR_UNSUPPORTED = 1, 2, 3, 4, 5
INHERIT_REFERENCE, INHERIT_EAGER, INHERIT_EXTERNAL = 0, 1, 2
MODEL_RNGBRG_2D = 0
MODEL_VICTORIAPARK_3D = 1
VP_MAX_PD = 16
MAX_CANDIDATES = 256
MAX_Z = 64
MAX_EVAL = 64


class FilterConfig(C.Structure):
    """RBPHDFilter::Config (include/RBPHDFilter.hpp:90-146)."""
    _fields_ = [
        ("birthGaussianWeight", C.c_double),
        ("birthGaussianMeasurementCountThreshold", C.c_uint),
        ("birthGaussianMeasurementCheckThreshold", C.c_uint),
        ("birthGaussianMeasurementSupportDist", C.c_double),
        ("birthGaussianCurrentMeasurementCountThreshold", C.c_uint),
        ("newGaussianCreateInnovMDThreshold", C.c_double),
        ("importanceWeightingEvalPointCount", C.c_int),
        ("importanceWeightingEvalPointGuassianWeight", C.c_double),
        ("importanceWeightingMeasurementLikelihoodMDThreshold", C.c_double),
        ("gaussianMergingThreshold", C.c_double),
        ("gaussianMergingCovarianceInflationFactor", C.c_double),
        ("gaussianPruningThreshold", C.c_double),
        ("minUpdatesBeforeResample", C.c_int),
        ("minMeasurementsBeforeResample", C.c_int),
        ("useClusterProcess", C.c_int),
    ]


class RngBrgConfig(C.Structure):
    """MeasurementModel_RngBrg::Config + R (include/MeasurementModel_RngBrg.hpp:65-71)."""
    _fields_ = [
        ("R", C.c_double * 4),
        ("probabilityOfDetection", C.c_double),
        ("uniformClutterIntensity", C.c_double),
        ("rangeLimMax", C.c_double),
        ("rangeLimMin", C.c_double),
        ("rangeLimBuffer", C.c_double),
    ]


class VPConfig(C.Structure):
    """MeasurementModel_VictoriaPark::Config + R, Slb (include/MeasurementModel_VictoriaPark.hpp:150-158)."""
    _fields_ = [
        ("R", C.c_double * 9),
        ("Slb", C.c_double),
        ("PdTable", C.c_double * VP_MAX_PD),
        ("nPd", C.c_int),
        ("expectedClutterNumber", C.c_double),
        ("rangeLimMax", C.c_double),
        ("rangeLimMin", C.c_double),
        ("bearingLimitMax", C.c_double),
        ("bearingLimitMin", C.c_double),
        ("bufferZonePd", C.c_double),
    ]


class KFConfig(C.Structure):
    """KalmanFilter_RngBrg::Config (include/KalmanFilter_RngBrg.hpp:55-60)."""
    _fields_ = [("rangeInnovationThreshold", C.c_double), ("bearingInnovationThreshold", C.c_double)]


class FastSlamConfig(C.Structure):
    """FastSLAM::Config (include/FastSLAM.hpp:106-132)."""
    _fields_ = [
        ("minUpdatesBeforeResample", C.c_int),
        ("minMeasurementsBeforeResample", C.c_int),
        ("landmarkExistencePrior", C.c_double),
        ("mapExistencePruneThreshold", C.c_double),
        ("minLogMeasurementLikelihood", C.c_double),
        ("nParticlesMax", C.c_int),
        ("maxNDataAssocHypotheses", C.c_uint),
        ("maxDataAssocLogLikelihoodDiff", C.c_double),
        ("landmarkCandidateMeasurementSupportDist", C.c_double),
        ("landmarkCandidateMeasurementCountThreshold", C.c_uint),
        ("landmarkCandidateCurrentMeasurementCountThreshold", C.c_uint),
        ("landmarkCandidateMeasurementCheckThreshold", C.c_uint),
        ("landmarkLockWeight", C.c_double),
        ("pruningMeasurementsThreshold", C.c_uint),
    ]


class Timing(C.Structure):
    """RBPHDFilter::TimingInfo (include/RBPHDFilter.hpp:152-167), ns."""
    _fields_ = [(n + s, C.c_longlong) for n in
                ("predict", "mapUpdate", "mapUpdate_kf", "particleWeighting", "mapMerge", "mapPrune", "particleResample")
                for s in ("_wall", "_cpu")]


# every symbol include/rfsgpu.h declares, without prefix (tests check the product .so exports them all)
ABI_SYMBOLS = [
    "abi_version", "create", "destroy", "last_error", "default_filter_config", "set_filter_config",
    "get_filter_config", "set_model_rngbrg", "set_kf_config", "set_lmk_process_noise", "set_poses", "get_poses",
    "set_weights", "get_weights", "gm_size", "get_landmark", "import_gm", "export_gm", "gm_sizes", "predict_map",
    "update", "update_map", "importance_weighting", "merge", "prune", "get_unused", "landmarks_in_fov",
    "weight_sums", "weight_sums_async", "weight_sums_device_ptr", "normalize_weights", "resample_apply",
    "get_timing", "reset_timing", "synchronize", "stream", "last_kernel_ns", "last_step_variant", "mat_perm", "mat_perm_last_kernel_ms",
    "set_stream", "bind_weight_sums_buffer", "save_state", "restore_state", "state_ring_create", "state_ring_seed", "state_ring_next", "import_aux",
    "set_model_victoriapark", "set_laser_scan", "export_birth_candidates", "import_birth_candidates",
    "update_async", "kernel_time_stats", "post_kernel_avg_ns", "set_step_timing_stride", "step_launch_order",
    "default_fastslam_config", "set_fastslam_config", "get_fastslam_config", "fastslam_update",
    "normalize_weights_parts", "create_ex", "n_particles", "max_particles", "resample_apply_n",
    "fastslam_set_resample_occured", "particle_parents", "vp_probe_pd",
    "slab_row_bytes", "export_slab_rows", "import_slab_rows", "weights_device_ptr", "step_async", "set_step_inputs_async", "predict_map_async", "set_phase_timing", "static_steps_async", "propagate_ackerman_async", "propagate_ackerman_run_async", "set_partition_mode", "get_partition_mode",
    "set_birth_inheritance", "get_birth_inheritance", "get_particle_ids", "set_particle_ids", "resample_occured", "get_unused_masks", "set_unused_masks", "has_birth_candidates", "predict_map_level", "murty_partition_sums", "cycle_async", "update_io", "step_async_deferred", "step_async_trailing", "collective_gate", "collective_publish", "collective_probe",
    "group_create", "group_destroy", "group_last_error", "group_n_shards", "group_n_particles", "group_shard", "group_locate",
    "group_set_filter_config", "group_set_model_rngbrg", "group_set_kf_config", "group_set_lmk_process_noise", "group_set_poses",
    "group_get_poses", "group_set_weights", "group_get_weights", "group_predict_map", "group_update", "group_normalize",
    "group_resample", "group_apply_plan", "group_migration_stats", "group_gm_size", "group_get_landmark", "group_synchronize", "group_set_birth_inheritance", "group_get_particle_ids",
    "group_update_io", "group_update_deferred", "group_set_model_victoriapark", "group_set_laser_scan", "group_set_phase_timing", "group_get_timing", "group_collective",
]

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


class EngineError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"rfsgpu status {status}: {msg}")
        self.status = status


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


class CFilter:
    """One filter handle behind the C ABI (one GPU / one shard of particles)."""

    def __init__(self, lib, prefix, n_particles, model=MODEL_RNGBRG_2D, device_id=0, gm_capacity=512, max_particles=None, borrowed=None):
        self._lib, self._p = lib, prefix
        self.model = model
        self.device_id = device_id
        self.dm = self.dz = 3 if model == MODEL_VICTORIAPARK_3D else 2
        self._borrowed = borrowed is not None
        if self._borrowed:          # a shard handle owned by an rfsgpu_group
            self._h = C.c_void_p(borrowed)
            return
        self._h = C.c_void_p()
        fn = self._fn("create_ex")
        fn.restype = C.c_int
        rc = fn(C.byref(self._h), C.c_int(model), C.c_int(int(n_particles)), C.c_int(device_id), C.c_int(gm_capacity),
                C.c_int(int(max_particles) if max_particles is not None else int(n_particles)))
        if rc != OK:
            self._h = C.c_void_p()
            raise EngineError(rc, "create failed (no gfx950 device / bad arguments)")

    @property
    def n(self):
        """nParticles_: the particle count NOW (the multi-hypothesis FastSLAM update and resample_apply_n change it)."""
        fn = self._fn("n_particles")
        fn.restype = C.c_int
        return fn(self._h)

    @property
    def max_particles(self):
        """The particle capacity the handle was created with (rfsgpu_create_ex)."""
        fn = self._fn("max_particles")
        fn.restype = C.c_int
        return fn(self._h)

    # -- plumbing ------------------------------------------------------------------------------
    def _fn(self, name):
        return getattr(self._lib, self._p + name)

    def _call(self, name, *args):
        fn = self._fn(name)
        fn.restype = C.c_int
        rc = fn(self._h, *args)
        if rc != OK:
            le = self._fn("last_error")
            le.restype = C.c_char_p
            raise EngineError(rc, (le(self._h) or b"").decode())
        return rc

    def close(self):
        if getattr(self, "_borrowed", False):
            self._h = C.c_void_p()
            return
        if getattr(self, "_h", None) and self._h.value:
            d = self._fn("destroy")
            d.restype = None
            d(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _ptr(a):
        return a.ctypes.data_as(C.c_void_p)

    # -- configuration -------------------------------------------------------------------------
    def default_filter_config(self):
        cfg = FilterConfig()
        fn = self._fn("default_filter_config")
        fn.restype = None
        fn(C.byref(cfg))
        return cfg

    def set_filter_config(self, cfg):
        self._call("set_filter_config", C.byref(cfg))

    def get_filter_config(self):
        cfg = FilterConfig()
        self._call("get_filter_config", C.byref(cfg))
        return cfg

    # -- FastSLAM on the same handle (include/FastSLAM.hpp) -------------------------------------
    def default_fastslam_config(self):
        cfg = FastSlamConfig()
        fn = self._fn("default_fastslam_config")
        fn.restype = None
        fn(C.byref(cfg))
        return cfg

    def set_fastslam_config(self, cfg):
        self._call("set_fastslam_config", C.byref(cfg))

    def get_fastslam_config(self):
        cfg = FastSlamConfig()
        self._call("get_fastslam_config", C.byref(cfg))
        return cfg

    def vp_probe_pd(self, slot):
        """Probe: Pd and the near-limit flag of every Gaussian of particle `slot` (Victoria Park model), as the kernels see them."""
        n = int(self.gm_sizes()[slot])
        pd = np.empty(max(n, 1), dtype=np.float64)
        cl = np.empty(max(n, 1), dtype=np.int32)
        self._call("vp_probe_pd", C.c_int(int(slot)), self._ptr(pd), self._ptr(cl), C.c_int(n))
        return pd[:n], cl[:n].astype(bool)

    def fastslam_set_resample_occured(self, flag):
        self._call("fastslam_set_resample_occured", C.c_int(1 if flag else 0))

    def particle_parents(self):
        """Which slot each particle was copied from by the last (multi-hypothesis) fastslam_update."""
        n = self.n
        p = np.empty(n, dtype=np.int32)
        self._call("particle_parents", self._ptr(p), C.c_int(n))
        return p

    def fastslam_update(self, Z):
        """FastSLAM::updateMap for every particle (:387-418); resampleWithMapCopy stays with the caller."""
        Z = _f64(Z).reshape(-1, self.dz)
        self._call("fastslam_update", self._ptr(Z), C.c_int(Z.shape[0]))

    def set_model_rngbrg(self, R, Pd, c, rmax, rmin, rbuf):
        m = RngBrgConfig()
        R = _f64(R, (4,))
        for k in range(4):
            m.R[k] = R[k]
        m.probabilityOfDetection, m.uniformClutterIntensity = Pd, c
        m.rangeLimMax, m.rangeLimMin, m.rangeLimBuffer = rmax, rmin, rbuf
        self._call("set_model_rngbrg", C.byref(m))

    def set_model_victoriapark(self, R, Slb, pd_table, expected_clutter, rmax, rmin, bmax, bmin, buffer_pd):
        m = VPConfig()
        R = _f64(R, (9,))
        for k in range(9):
            m.R[k] = R[k]
        m.Slb = Slb
        pd_table = list(pd_table)
        assert len(pd_table) <= VP_MAX_PD
        for k, v in enumerate(pd_table):
            m.PdTable[k] = v
        m.nPd = len(pd_table)
        m.expectedClutterNumber = expected_clutter
        m.rangeLimMax, m.rangeLimMin, m.bearingLimitMax, m.bearingLimitMin, m.bufferZonePd = rmax, rmin, bmax, bmin, buffer_pd
        self._call("set_model_victoriapark", C.byref(m))

    def set_laser_scan(self, scan):
        s = _f64(scan).reshape(-1)
        self._call("set_laser_scan", self._ptr(s), C.c_int(s.size))

    def export_birth_candidates(self, i):
        n = MAX_CANDIDATES
        mean, cov = np.empty((n, self.dm)), np.empty((n, self.dm, self.dm))
        sup, chk = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.int32)
        nout = C.c_int()
        self._call("export_birth_candidates", C.c_int(i), C.c_int(n), C.byref(nout), self._ptr(mean), self._ptr(cov), self._ptr(sup), self._ptr(chk))
        k = min(nout.value, n)
        return mean[:k], cov[:k], sup[:k], chk[:k]

    def import_birth_candidates(self, i, mean, cov, sup, chk):
        sup = np.ascontiguousarray(sup, dtype=np.int32)
        k = sup.size
        mean = _f64(mean, (k, self.dm)) if k else np.zeros((0, self.dm))
        cov = _f64(cov, (k, self.dm, self.dm)) if k else np.zeros((0, self.dm, self.dm))
        chk = np.ascontiguousarray(chk, dtype=np.int32)
        self._call("import_birth_candidates", C.c_int(i), C.c_int(k), self._ptr(mean), self._ptr(cov), self._ptr(sup), self._ptr(chk))

    def set_kf_config(self, range_thr, bearing_thr):
        k = KFConfig(range_thr, bearing_thr)
        self._call("set_kf_config", C.byref(k))

    def set_lmk_process_noise(self, Q):
        self._call("set_lmk_process_noise", self._ptr(_f64(Q, (self.dm * self.dm,))))

    # -- particle state ------------------------------------------------------------------------
    def set_poses(self, x, cov=None):
        x = _f64(x, (self.n, 3))
        if cov is None:
            self._call("set_poses", self._ptr(x), C.c_void_p(), C.c_int(0))
        else:
            cov = _f64(cov)
            stride = 0 if cov.size == 9 else 9
            assert cov.size in (9, 9 * self.n)
            self._call("set_poses", self._ptr(x), self._ptr(cov), C.c_int(stride))

    def get_poses(self):
        x = np.empty((self.n, 3))
        self._call("get_poses", self._ptr(x))
        return x

    def set_weights(self, w):
        self._call("set_weights", self._ptr(_f64(w, (self.n,))))

    def get_weights(self):
        w = np.empty(self.n)
        self._call("get_weights", self._ptr(w))
        return w

    # -- map access ----------------------------------------------------------------------------
    def getGMSize(self, i):
        fn = self._fn("gm_size")
        fn.restype = C.c_int
        return fn(self._h, C.c_int(i))

    def gm_sizes(self):
        s = np.empty(self.n, dtype=np.int32)
        self._call("gm_sizes", self._ptr(s))
        return s

    def getLandmark(self, i, m):
        mean = np.empty(self.dm)
        cov = np.empty((self.dm, self.dm))
        w = C.c_double()
        fn = self._fn("get_landmark")
        fn.restype = C.c_int
        rc = fn(self._h, C.c_int(i), C.c_int(m), self._ptr(mean), self._ptr(cov), C.byref(w))
        if rc != OK:
            return None
        return mean, cov, w.value

    def import_gm(self, i, w, mean, cov):
        w = _f64(w).reshape(-1)
        n = w.size
        mean = _f64(mean, (n, self.dm))
        cov = _f64(cov, (n, self.dm, self.dm))
        self._call("import_gm", C.c_int(i), C.c_int(n), self._ptr(w), self._ptr(mean), self._ptr(cov))

    def export_gm(self, i):
        n = max(self.getGMSize(i), 0)
        w, wp = np.empty(n), np.empty(n)
        mean, cov = np.empty((n, self.dm)), np.empty((n, self.dm, self.dm))
        nout = C.c_int()
        self._call("export_gm", C.c_int(i), C.c_int(n), C.byref(nout), self._ptr(w), self._ptr(wp), self._ptr(mean), self._ptr(cov))
        k = min(nout.value, n)
        return w[:k], wp[:k], mean[:k], cov[:k]

    def import_aux(self, i, unused_idx, n_in_fov):
        u = np.ascontiguousarray(unused_idx, dtype=np.int32)
        self._call("import_aux", C.c_int(i), self._ptr(u), C.c_int(u.size), C.c_int(int(n_in_fov)))

    # -- hot path ------------------------------------------------------------------------------
    def predict_map(self, add_birth=True):
        self._call("predict_map", C.c_int(1 if add_birth else 0))

    def _z(self, Z):
        Z = _f64(Z).reshape(-1, self.dz) if np.size(Z) else np.zeros((0, self.dz))
        return Z, C.c_int(Z.shape[0])

    def update(self, Z):
        Z, n = self._z(Z)
        self._call("update", self._ptr(Z), n)

    def update_map(self, Z):
        Z, n = self._z(Z)
        self._call("update_map", self._ptr(Z), n)

    def importance_weighting(self):
        self._call("importance_weighting")

    def merge(self):
        self._call("merge")

    def prune(self):
        self._call("prune")

    def get_unused(self, i):
        idx = np.empty(MAX_Z, dtype=np.int32)
        n = C.c_int()
        self._call("get_unused", C.c_int(i), self._ptr(idx), C.c_int(MAX_Z), C.byref(n))
        return idx[: n.value].copy()

    def landmarks_in_fov(self, i):
        n = C.c_int()
        self._call("landmarks_in_fov", C.c_int(i), C.byref(n))
        return n.value

    # -- weights / resampling ------------------------------------------------------------------
    def weight_sums(self):
        out = np.empty(2)
        self._call("weight_sums", self._ptr(out))
        return out

    def normalize_weights(self, total, sum_dev_ptr=None, n_parts=1):
        if n_parts == 1:
            self._call("normalize_weights", C.c_double(total), C.c_void_p(sum_dev_ptr))
        else:
            self._call("normalize_weights_parts", C.c_double(total), C.c_void_p(sum_dev_ptr), C.c_int(n_parts))

    def resample_apply(self, src_slot, n_out=None):
        """ParticleFilter::resample's copies; n_out < n shrinks the particle set (resample(n), ParticleFilter.hpp:417-483)."""
        s = np.ascontiguousarray(src_slot, dtype=np.int32)
        if n_out is None:
            assert s.size == self.n
            self._call("resample_apply", self._ptr(s))
        else:
            assert s.size == n_out
            self._call("resample_apply_n", self._ptr(s), C.c_int(int(n_out)))

    # birth-state inheritance after a resampling (rfsgpu.h: RFSGPU_INHERIT_*; include/RBPHDFilter.hpp:1005-1011)
    def set_birth_inheritance(self, mode):
        self._call("set_birth_inheritance", C.c_int(int(mode)))

    def get_birth_inheritance(self):
        fn = self._fn("get_birth_inheritance")
        fn.restype = C.c_int
        return int(fn(self._h))

    def get_particle_ids(self):
        """(Particle::getId, Particle::getParentId) of the particle in every slot."""
        ids = np.zeros(self.n, dtype=np.int32)
        par = np.zeros(self.n, dtype=np.int32)
        self._call("get_particle_ids", self._ptr(ids), self._ptr(par))
        return ids, par

    def set_particle_ids(self, ids=None, parent_ids=None):
        a = None if ids is None else np.ascontiguousarray(ids, dtype=np.int32)
        b = None if parent_ids is None else np.ascontiguousarray(parent_ids, dtype=np.int32)
        self._call("set_particle_ids", C.c_void_p(None) if a is None else self._ptr(a), C.c_void_p(None) if b is None else self._ptr(b))

    def resample_occured(self):
        fn = self._fn("resample_occured")
        fn.restype = C.c_int
        return bool(fn(self._h))

    def predict_map_level(self, add_birth, level_of_slot, level, do_static):
        lv = np.ascontiguousarray(level_of_slot, dtype=np.int32)
        assert lv.size == self.n
        self._call("predict_map_level", C.c_int(1 if add_birth else 0), self._ptr(lv), C.c_int(int(level)), C.c_int(1 if do_static else 0))

    def has_birth_candidates(self):
        """Has this handle ever held birth-candidate lists (rfsgpu_has_birth_candidates)?"""
        fn = self._fn("has_birth_candidates")
        fn.restype = C.c_int
        rc = fn(self._h)
        if rc < 0:
            raise EngineError(rc, "has_birth_candidates: null handle")
        return bool(rc)

    def murty_partition_sums(self, mats, nR, nC):
        """[test] Murty-200 sums of the given extended tables by the step's post kernel (rfsgpu_murty_partition_sums)."""
        flat = np.ascontiguousarray(np.concatenate([np.asarray(m, dtype=np.float64).ravel() for m in mats]))
        r = np.ascontiguousarray(nR, dtype=np.int32)
        c = np.ascontiguousarray(nC, dtype=np.int32)
        out = np.zeros(len(mats), dtype=np.float64)
        self._call("murty_partition_sums", self._ptr(flat), self._ptr(r), self._ptr(c), C.c_int(len(mats)), self._ptr(out))
        return out

    def get_unused_masks(self):
        m = np.zeros(self.n, dtype=np.uint64)
        self._call("get_unused_masks", self._ptr(m))
        return m

    def set_unused_masks(self, masks):
        m = np.ascontiguousarray(masks, dtype=np.uint64)
        assert m.size == self.n
        self._call("set_unused_masks", self._ptr(m))

    def set_partition_mode(self, exact):
        """Partitions with nR + nC > 8: Murty-200 as the reference (False, default) or the exact subset recurrence (True)."""
        self._call("set_partition_mode", C.c_int(1 if exact else 0))

    # -- asynchronous host loop -----------------------------------------------------------------------------------
    def set_step_inputs_async(self, x=None, cov=None, scan=None):
        xp = cp = sp = C.c_void_p()
        stride, ns = 0, 0
        if x is not None:
            x = _f64(x, (self.n, 3))
            xp = self._ptr(x)
            if cov is not None:
                cov = _f64(cov)
                stride = 0 if cov.size == 9 else 9
                cp = self._ptr(cov)
        if scan is not None:
            scan = _f64(scan).reshape(-1)
            sp, ns = self._ptr(scan), scan.size
        self._call("set_step_inputs_async", xp, cp, C.c_int(stride), sp, C.c_int(ns))

    def set_phase_timing(self, on=True):
        """update() as three launches with per-phase TimingInfo / last_kernel_ns (True) or as one fused launch (False, default)."""
        self._call("set_phase_timing", C.c_int(1 if on else 0))

    def predict_map_async(self, add_birth=True):
        self._call("predict_map_async", C.c_int(1 if add_birth else 0))

    def static_steps_async(self, n, noises=None):
        """A run of n birth-less predicts (Sigma += Q_k each) in one launch; noises [n][D][D] or None = the current noise n times."""
        q = None
        if noises is not None:
            q = _f64(noises, (int(n), self.dm, self.dm))
        self._call("static_steps_async", C.c_int(int(n)), self._ptr(q) if q is not None else None)

    def propagate_ackerman_async(self, u, var, dt, geom, seed, call):
        """ParticleFilter::propagate with MotionModel_Ackerman2d on the device (csrc/motion.h); var None: noise-free."""
        up = (C.c_double * 2)(float(u[0]), float(u[1]))
        vp = (C.c_double * 2)(float(var[0]), float(var[1])) if var is not None else None
        gp = (C.c_double * 4)(*[float(g) for g in geom])
        self._call("propagate_ackerman_async", up, vp, C.c_double(float(dt)), gp, C.c_ulonglong(int(seed)), C.c_ulonglong(int(call)))

    def propagate_ackerman_run_async(self, u, var, dt, geom, seed, call0):
        """n propagations in one launch: u [n][2], var [n][2] or None, dt [n]."""
        u = _f64(u).reshape(-1, 2)
        n = u.shape[0]
        v = _f64(var, (n, 2)) if var is not None else None
        d = _f64(dt, (n,))
        gp = (C.c_double * 4)(*[float(g) for g in geom])
        self._call("propagate_ackerman_run_async", C.c_int(n), self._ptr(u), self._ptr(v) if v is not None else None, self._ptr(d), gp,
                   C.c_ulonglong(int(seed)), C.c_ulonglong(int(call0)))

    # -- cross-shard migration (packed rows in the backend's own memory space: device memory for the engine) ----------
    def slab_row_bytes(self):
        fn = self._fn("slab_row_bytes")
        fn.restype = C.c_size_t
        return int(fn(self._h))

    def export_slab_rows(self, slots, rows_ptr):
        s = np.ascontiguousarray(slots, dtype=np.int32)
        self._call("export_slab_rows", self._ptr(s), C.c_int(s.size), C.c_void_p(rows_ptr))

    def import_slab_rows(self, slots, rows_ptr):
        s = np.ascontiguousarray(slots, dtype=np.int32)
        self._call("import_slab_rows", self._ptr(s), C.c_int(s.size), C.c_void_p(rows_ptr))

    def weights_device_ptr(self):
        fn = self._fn("weights_device_ptr")
        fn.restype = C.c_void_p
        return fn(self._h)

    # -- timing --------------------------------------------------------------------------------
    def getTimingInfo(self):
        t = Timing()
        self._call("get_timing", C.byref(t))
        return t

    def reset_timing(self):
        self._call("reset_timing")

    def synchronize(self):
        self._call("synchronize")


def mat_perm(lib, prefix, A, device_id=0):
    """MatPerm::calc (src/MatrixPermanent.cpp:41-112), batched: A is (batch, n, n)."""
    A = _f64(A)
    if A.ndim == 2:
        A = A[None]
    b, n, _ = A.shape
    out = np.empty(b)
    fn = getattr(lib, prefix + "mat_perm")
    fn.restype = C.c_int
    rc = fn(A.ctypes.data_as(C.c_void_p), C.c_int(n), C.c_int(b), out.ctypes.data_as(C.c_void_p), C.c_int(device_id))
    if rc != OK:
        raise EngineError(rc, "mat_perm failed")
    return out


class Group:
    """rfsgpu_group_*: one filter over several GPUs from one host thread (contiguous particle blocks, one shard per device)."""

    def __init__(self, lib, n_particles, device_ids, gm_capacity=512, model=MODEL_RNGBRG_2D):
        self._lib = lib
        self._g = C.c_void_p()
        ids = np.ascontiguousarray(device_ids, dtype=np.int32)
        lib.rfsgpu_group_create.restype = C.c_int
        rc = lib.rfsgpu_group_create(C.byref(self._g), C.c_int(model), C.c_int(int(n_particles)), ids.ctypes.data_as(C.c_void_p), C.c_int(ids.size),
                                     C.c_int(gm_capacity))
        if rc != OK:
            self._g = C.c_void_p()
            raise EngineError(rc, "group_create failed")
        self.n = int(n_particles)
        self.dm = self.dz = 3 if model == MODEL_VICTORIAPARK_3D else 2
        lib.rfsgpu_group_shard.restype = C.c_void_p
        self.shards = [CFilter(lib, "rfsgpu_", 0, model=model, device_id=int(ids[k]), gm_capacity=gm_capacity,
                               borrowed=lib.rfsgpu_group_shard(self._g, C.c_int(k))) for k in range(ids.size)]

    def _call(self, name, *args):
        fn = getattr(self._lib, "rfsgpu_group_" + name)
        fn.restype = C.c_int
        rc = fn(self._g, *args)
        if rc != OK:
            le = self._lib.rfsgpu_group_last_error
            le.restype = C.c_char_p
            raise EngineError(rc, (le(self._g) or b"").decode())

    def close(self):
        if self._g and self._g.value:
            for s in self.shards:
                s.close()
            self._lib.rfsgpu_group_destroy.restype = None
            self._lib.rfsgpu_group_destroy(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def locate(self, particle):
        k, s = C.c_int(), C.c_int()
        self._call("locate", C.c_int(int(particle)), C.byref(k), C.byref(s))
        return k.value, s.value

    def set_filter_config(self, cfg):
        self._call("set_filter_config", C.byref(cfg))

    def set_model_rngbrg(self, R, Pd, c, rmax, rmin, rbuf):
        m = RngBrgConfig()
        R = _f64(R, (4,))
        for k in range(4):
            m.R[k] = R[k]
        m.probabilityOfDetection, m.uniformClutterIntensity = Pd, c
        m.rangeLimMax, m.rangeLimMin, m.rangeLimBuffer = rmax, rmin, rbuf
        self._call("set_model_rngbrg", C.byref(m))

    def set_kf_config(self, range_thr, bearing_thr):
        self._call("set_kf_config", C.byref(KFConfig(range_thr, bearing_thr)))

    def set_lmk_process_noise(self, Q):
        self._call("set_lmk_process_noise", _f64(Q, (self.dm * self.dm,)).ctypes.data_as(C.c_void_p))

    def default_filter_config(self):
        return self.shards[0].default_filter_config()

    def set_poses(self, x, cov=None):
        x = _f64(x, (self.n, 3))
        if cov is None:
            self._call("set_poses", x.ctypes.data_as(C.c_void_p), C.c_void_p(), C.c_int(0))
        else:
            cov = _f64(cov)
            self._call("set_poses", x.ctypes.data_as(C.c_void_p), cov.ctypes.data_as(C.c_void_p), C.c_int(0 if cov.size == 9 else 9))

    def get_poses(self):
        x = np.empty((self.n, 3))
        self._call("get_poses", x.ctypes.data_as(C.c_void_p))
        return x

    def set_weights(self, w):
        self._call("set_weights", _f64(w, (self.n,)).ctypes.data_as(C.c_void_p))

    def get_weights(self):
        w = np.empty(self.n)
        self._call("get_weights", w.ctypes.data_as(C.c_void_p))
        return w

    def import_gm(self, i, w, mean, cov):
        k, s = self.locate(i)
        self.shards[k].import_gm(s, w, mean, cov)

    def export_gm(self, i):
        k, s = self.locate(i)
        return self.shards[k].export_gm(s)

    def get_unused(self, i):
        k, s = self.locate(i)
        return self.shards[k].get_unused(s)

    def gm_sizes(self):
        return np.concatenate([s.gm_sizes() for s in self.shards])

    def predict_map(self, add_birth=True):
        self._call("predict_map", C.c_int(1 if add_birth else 0))

    def set_birth_inheritance(self, mode):
        self._call("set_birth_inheritance", C.c_int(int(mode)))

    def get_particle_ids(self):
        ids = np.zeros(self.n, dtype=np.int32)
        par = np.zeros(self.n, dtype=np.int32)
        self._call("get_particle_ids", ids.ctypes.data_as(C.c_void_p), par.ctypes.data_as(C.c_void_p))
        return ids, par

    def update(self, Z):
        Z = _f64(Z).reshape(-1, self.dz)
        sums = np.empty(2)
        self._call("update", Z.ctypes.data_as(C.c_void_p), C.c_int(Z.shape[0]), sums.ctypes.data_as(C.c_void_p))
        return sums

    def normalize(self):
        sums = np.empty(2)
        self._call("normalize", sums.ctypes.data_as(C.c_void_p))
        return sums

    def resample(self, eff_n_threshold, u01):
        fired = C.c_int()
        plan = np.empty(self.n, dtype=np.int32)
        self._call("resample", C.c_double(eff_n_threshold), C.c_double(u01), C.byref(fired), plan.ctypes.data_as(C.c_void_p))
        return bool(fired.value), (plan if fired.value else None)

    def migration_stats(self):
        r, b = C.c_longlong(), C.c_longlong()
        self._call("migration_stats", C.byref(r), C.byref(b))
        return r.value, b.value

    def collective(self):
        """"rccl" (the weight sums are all-reduced over RCCL on the shards' streams) or "host: <why not>"."""
        fn = self._lib.rfsgpu_group_collective
        fn.restype = C.c_char_p
        return (fn(self._g) or b"").decode()

    def set_model_victoriapark(self, R, Slb, pd_table, expected_clutter, rmax, rmin, bmax, bmin, buffer_pd):
        m = VPConfig()
        R = _f64(R, (9,))
        for k in range(9):
            m.R[k] = R[k]
        m.Slb = Slb
        pd_table = list(pd_table)
        for k, v in enumerate(pd_table):
            m.PdTable[k] = v
        m.nPd = len(pd_table)
        m.expectedClutterNumber = expected_clutter
        m.rangeLimMax, m.rangeLimMin, m.bearingLimitMax, m.bearingLimitMin, m.bufferZonePd = rmax, rmin, bmax, bmin, buffer_pd
        self._call("set_model_victoriapark", C.byref(m))

    def set_laser_scan(self, scan):
        s = _f64(scan).reshape(-1)
        self._call("set_laser_scan", s.ctypes.data_as(C.c_void_p), C.c_int(s.size))

    def set_phase_timing(self, on=True):
        self._call("set_phase_timing", C.c_int(1 if on else 0))

    def getTimingInfo(self):
        t = Timing()
        self._call("get_timing", C.byref(t))
        return t

    def get_filter_config(self):
        return self.shards[0].get_filter_config()

    def update_deferred(self, Z):
        """rfsgpu_group_update_deferred: the normalisation trails by one step, the collective runs beside the next step's kernel."""
        Z = _f64(Z).reshape(-1, self.dz)
        self._call("update_deferred", Z.ctypes.data_as(C.c_void_p), C.c_int(Z.shape[0]))

    def update_nosums(self, Z):
        """rfsgpu_group_update without the host copy of the sums: with the RCCL collective nothing waits for the GPUs."""
        Z = _f64(Z).reshape(-1, self.dz)
        self._call("update", Z.ctypes.data_as(C.c_void_p), C.c_int(Z.shape[0]), C.c_void_p())

    def synchronize(self):
        self._call("synchronize")
