"""ctypes binding of include/otmb.h (libotmb_hip.so).  No fallback: a missing library raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OTMB_LIB_OVERRIDE") or os.path.join(_HERE, "lib", "libotmb_hip.so")  # (override: A/B of builds in fresh processes, tools/fresh_ab.sh)

OK = 0
STATUS_NAMES = {
    1: "RHO_NAN", 2: "TADV_NAN", 3: "TKH_NAN", 4: "TKVML_NAN", 5: "TKVDEEP_NAN", 6: "FLUX_INTO_LAND",
    7: "UNKNOWN_TOPOLOGY", 8: "ALL_MISSING", 9: "ALLOC", 10: "HIP", 11: "INVALID_ARG", 12: "NO_PLAN",
    13: "NONCANONICAL_INDICES", 14: "CAPACITY", 15: "PUSH_MASK", 16: "ASYMMETRIC_PATTERN", 17: "GIVEN_FOREIGN",
}
GIVEN_FOREIGN = 17
CAPACITY = 14
PHI_ORDER = ("east", "west", "north", "south", "top", "bottom")  # OTMB_EAST..OTMB_BOTTOM
HDIRS = ("west", "east", "south", "north")  # OTMB_DIR_*
MATS = ("T", "Tadv", "TκH", "TκVML", "TκVdeep")  # OTMB_T..OTMB_TKVDEEP

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)


class OtmbError(RuntimeError):
    """Raised with the reference's own message for the reference's own failures
    (ErrorException / AssertionError texts of src/matrixbuilding.jl:39,61,90,114,233,
    src/gridtopology.jl:111-116, src/velocities.jl:199-200)."""

    def __init__(self, status, message, step=None):
        super().__init__(message)
        self.status = status
        self.name = STATUS_NAMES.get(status, str(status))
        self.step = step  # asynchronous pipelines: 0-based index of the first step that failed


class Csc(C.Structure):
    """otmb_csc: a SparseMatrixCSC{Float64,Int64} by its three arrays (1-based)."""
    _fields_ = [("colptr", C.c_void_p), ("rowval", C.c_void_p), ("nzval", C.c_void_p), ("nnz", C.c_int64)]


class TmArgs(C.Structure):
    _fields_ = [
        ("nx", C.c_int64), ("ny", C.c_int64), ("nz", C.c_int64),
        ("topology", C.c_int32), ("upwind", C.c_int32), ("n_wet", C.c_int64),
        ("phi", C.c_void_p * 6), ("v3d", C.c_void_p), ("thkcello", C.c_void_p),
        ("rho", C.c_void_p), ("rho_scalar", C.c_double), ("lwet3d", C.c_void_p), ("lwet", C.c_void_p),
        ("edge_length", C.c_void_p * 4), ("dist_nbr", C.c_void_p * 4),
        ("area2d", C.c_void_p), ("zt", C.c_void_p), ("mlotst", C.c_void_p),
        ("kappa_h", C.c_double), ("kappa_vml", C.c_double), ("kappa_vdeep", C.c_double),
        ("push_mask", C.c_void_p),
        ("only_t", C.c_int32),
        ("ignore_ops", C.c_int32),
        ("skip_ops", C.c_int32),  # bit m: matrix m is not wanted (neither counted, written nor copied home)
        ("given", Csc * 5),  # operators the caller passes (transportmatrix's Tadv = / TκH = / TκVML = / TκVdeep = keywords)
    ]


class FfCounts(C.Structure):
    _fields_ = [("tables", C.c_void_p), ("lwet3d", C.c_void_p), ("mlotst", C.c_void_p), ("zt", C.c_void_p),
                ("n_wet", C.c_int64), ("upwind", C.c_int32), ("only_t", C.c_int32)]


class FfSlab(C.Structure):
    _fields_ = [("k_own0", C.c_int64), ("nz_ext", C.c_int64), ("wet_base", C.c_int64)]


# every symbol include/otmb.h declares: (restype, argtypes)
_vp = C.c_void_p
SYMBOLS = {
    "otmb_ctx_create": (C.c_int32, [C.c_int32, C.POINTER(_vp)]),
    "otmb_ctx_destroy": (None, [_vp]),
    "otmb_ctx_set_stream": (C.c_int32, [_vp, _vp]),
    "otmb_ctx_use_default_stream": (C.c_int32, [_vp]),
    "otmb_ctx_synchronize": (C.c_int32, [_vp]),
    "otmb_ctx_set_reuse_grid": (C.c_int32, [_vp, C.c_int32]),
    "otmb_ctx_set_reuse_fluxes": (C.c_int32, [_vp, C.c_int32]),
    "otmb_ctx_uploaded_bytes": (C.c_int64, [_vp]),
    "otmb_host_alloc": (C.c_int32, [_vp, C.c_int64, C.POINTER(_vp)]),
    "otmb_host_free": (C.c_int32, [_vp, _vp]),
    "otmb_host_pool_stats": (C.c_int32, [_ip, _ip, _ip]),
    "otmb_mgpu_create": (C.c_int32, [C.c_int32, C.POINTER(C.c_int32), C.POINTER(_vp)]),
    "otmb_mgpu_destroy": (None, [_vp]),
    "otmb_mgpu_last_error": (C.c_char_p, [_vp]),
    "otmb_mgpu_ndev": (C.c_int32, [_vp]),
    "otmb_mgpu_transport": (C.c_int32, [_vp]),
    "otmb_mgpu_partition": (C.c_int32, [_vp, _ip]),
    "otmb_mgpu_set_reuse": (C.c_int32, [_vp, C.c_int32, C.c_int32]),
    "otmb_mgpu_uploaded_bytes": (C.c_int64, [_vp]),
    "otmb_mgpu_set_chain_pieces": (C.c_int32, [_vp, C.c_int32]),
    "otmb_balanced_partition": (C.c_int32, [_ip, C.c_int64, C.c_int32, _ip]),
    "otmb_mgpu_facefluxes": (C.c_int32, [_vp, _vp, _vp, C.c_int32, _vp, C.c_double, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.POINTER(_vp * 6)]),
    "otmb_mgpu_transportmatrix_plan": (C.c_int32, [_vp, C.POINTER(TmArgs), C.POINTER(C.c_int64 * 5)]),
    "otmb_mgpu_transportmatrix_fetch": (C.c_int32, [_vp, C.POINTER(_vp * 5), C.POINTER(_vp * 5), C.POINTER(_vp * 5), C.POINTER(C.c_int64 * 5)]),
    "otmb_mgpu_transportmatrix_onepass": (C.c_int32, [_vp, C.POINTER(TmArgs), C.POINTER(_vp * 5), C.POINTER(_vp * 5), C.POINTER(_vp * 5),
                                                      C.POINTER(C.c_int64 * 5), C.POINTER(C.c_int64 * 5)]),
    "otmb_static_capacity": (C.c_int32, [_vp, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.POINTER(C.c_int64 * 5)]),
    "otmb_ctx_set_tile_order": (C.c_int32, [_vp, C.c_int32]),
    "otmb_ctx_forget_given": (C.c_int32, [_vp]),
    "otmb_ctx_given_state": (C.c_int32, [_vp, C.c_int32]),
    "otmb_ctx_given_checks": (C.c_int64, [_vp]),
    "otmb_last_error": (C.c_char_p, [_vp]),
    "otmb_status_string": (C.c_char_p, [C.c_int32]),
    "otmb_version": (C.c_char_p, []),
    "otmb_ctx_timing_enable": (C.c_int32, [_vp, C.c_int32]),
    "otmb_ctx_timing_collect": (C.c_int32, [_vp, _dp, _ip, C.c_int32]),
    "otmb_kernel_name": (C.c_char_p, [C.c_int32]),
    "otmb_ctx_box_ceilings": (C.c_int32, [_vp, _dp, _dp]),
    "otmb_ctx_stream_mix": (C.c_int32, [_vp, C.c_int32, _vp, _ip, C.c_int32, _vp, _ip, C.c_int64, _dp]),
    "otmb_makeindices_dev": (C.c_int32, [_vp, _vp, C.c_int64, C.c_int64, C.c_int64, _vp, _vp, _vp, _ip]),
    "otmb_makeindices": (C.c_int32, [_vp, _vp, C.c_int64, C.c_int64, C.c_int64, _vp, _vp, _vp, _ip]),
    "otmb_facefluxes_dev": (C.c_int32, [_vp, _vp, _vp, C.c_int32, _vp, C.c_double, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.POINTER(_vp * 6)]),
    "otmb_facefluxes": (C.c_int32, [_vp, _vp, _vp, C.c_int32, _vp, C.c_double, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.POINTER(_vp * 6)]),
    "otmb_bgrid_to_cgrid_dev": (C.c_int32, [_vp, _vp, _vp, C.c_int32, C.c_double, C.c_int64, C.c_int64, C.c_int64, _vp, _vp]),
    "otmb_bgrid_to_cgrid": (C.c_int32, [_vp, _vp, _vp, C.c_int32, C.c_double, C.c_int64, C.c_int64, C.c_int64, _vp, _vp]),
    "otmb_velocity2fluxes_dev": (C.c_int32, [_vp, _vp, _vp, C.c_int32, _vp, C.c_double, _vp, _vp, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int32, _vp, _vp]),
    "otmb_fluxes2velocity_dev": (C.c_int32, [_vp, _vp, _vp, C.c_int32, _vp, C.c_double, _vp, _vp, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int32, _vp, _vp]),
    "otmb_velocity2fluxes": (C.c_int32, [_vp, _vp, _vp, C.c_int32, _vp, C.c_double, _vp, _vp, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int32, _vp, _vp]),
    "otmb_fluxes2velocity": (C.c_int32, [_vp, _vp, _vp, C.c_int32, _vp, C.c_double, _vp, _vp, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int32, _vp, _vp]),
    "otmb_facefluxes_slab_dev": (C.c_int32, [_vp, _vp, _vp, C.c_int32, _vp, C.c_double, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.POINTER(_vp * 6), _vp, _vp]),
    "otmb_facefluxes_rows_dev": (C.c_int32, [_vp, _vp, _vp, C.c_int32, _vp, C.c_double, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.POINTER(_vp * 6), _vp, _vp,
                                              C.c_int64, C.c_int64, C.c_int32]),
    "otmb_wetflags_dev": (C.c_int32, [_vp, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int32, _vp]),
    "otmb_facefluxes_flags_dev": (C.c_int32, [_vp, _vp, _vp, C.c_int32, _vp, C.c_double, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.POINTER(_vp * 6), _vp, _vp]),
    "otmb_count_tables_bytes": (C.c_int64, [_vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64]),
    "otmb_count_tables_dev": (C.c_int32, [_vp, _vp, _vp, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int32, _vp]),
    "otmb_facefluxes_counts_dev": (C.c_int32, [_vp, _vp, _vp, C.c_int32, _vp, C.c_double, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.POINTER(_vp * 6), _vp, C.POINTER(FfCounts)]),
    "otmb_count_tables_slab_dev": (C.c_int32, [_vp, _vp, _vp, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.POINTER(FfSlab), _vp]),
    "otmb_facefluxes_slab_counts_dev": (C.c_int32, [_vp, _vp, _vp, C.c_int32, _vp, C.c_double, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.POINTER(_vp * 6), _vp,
                                                    _vp, C.POINTER(FfCounts), C.POINTER(FfSlab), C.c_int64, C.c_int64, C.c_int32]),
    "otmb_facefluxes_counts_pending": (C.c_int32, [_vp]),
    "otmb_push_mask_dev": (C.c_int32, [_vp, C.POINTER(_vp * 6), _vp, C.c_int64, C.c_int64, _vp]),
    "otmb_lump_and_spray_plan_dev": (C.c_int32, [_vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _vp, _vp,
                                                  C.c_int64, C.c_int64, C.c_int64, _ip]),
    "otmb_lump_and_spray_fill_dev": (C.c_int32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "otmb_lump_and_spray": (C.c_int32, [_vp, _vp, _vp, C.c_int64, C.c_int64, C.c_int64, _vp, C.c_int64, _vp, _vp, C.c_int64,
                                         C.c_int64, C.c_int64, _vp, _vp, _vp, _vp, _vp, _ip]),
    "otmb_facefluxes_slab_flags": (C.c_int32, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "otmb_facefluxes_pending_flags": (C.c_int32, [_vp, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "otmb_transportmatrix_failed_step": (C.c_int32, [_vp, _ip]),
    "otmb_transportmatrix_result_step": (C.c_int32, [_vp, C.c_int64, C.POINTER(C.c_int64 * 5)]),
    "otmb_transportmatrix_set_slab": (C.c_int32, [_vp, C.c_int64]),
    "otmb_transportmatrix_dev": (C.c_int32, [_vp, C.POINTER(TmArgs), C.POINTER(_vp * 5), C.POINTER(_vp * 5), C.POINTER(_vp * 5), C.POINTER(C.c_int64 * 5)]),
    "otmb_transportmatrix_result": (C.c_int32, [_vp, C.POINTER(C.c_int64 * 5)]),
    "otmb_step_dev": (C.c_int32, [_vp, _vp, _vp, C.c_int32, C.c_double, _vp, _vp, _vp, C.POINTER(TmArgs), C.POINTER(_vp * 5), C.POINTER(_vp * 5), C.POINTER(_vp * 5),
                                   C.POINTER(C.c_int64 * 5)]),
    "otmb_transportmatrix_set_nnz_base": (C.c_int32, [_vp, C.POINTER(C.c_int64 * 5)]),
    "otmb_makegridmetrics_dev": (C.c_int32, [_vp, _vp, _vp, C.c_double, C.c_double, _vp, _vp, _vp, _vp, C.POINTER(C.c_int32 * 4),
                                              C.c_int64, C.c_int64, C.c_int64, C.c_int32, _vp, _vp, _vp, _vp,
                                              C.POINTER(_vp * 4), C.POINTER(_vp * 4), C.POINTER(_vp * 4)]),
    "otmb_makegridmetrics": (C.c_int32, [_vp, _vp, _vp, C.c_double, C.c_double, _vp, _vp, _vp, _vp, C.POINTER(C.c_int32 * 4),
                                          C.c_int64, C.c_int64, C.c_int64, C.c_int32, _vp, _vp, _vp, _vp,
                                          C.POINTER(_vp * 4), C.POINTER(_vp * 4), C.POINTER(_vp * 4)]),
    "otmb_bolus_gm_velocity_dev": (C.c_int32, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_double, C.c_double, _vp, _vp]),
    "otmb_bolus_gm_velocity": (C.c_int32, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_double, C.c_double, _vp, _vp]),
    "otmb_sparse_entries_plan_dev": (C.c_int32, [_vp, C.c_int32, C.POINTER(TmArgs), _ip]),
    "otmb_sparse_entries_fill_dev": (C.c_int32, [_vp, _vp, _vp, _vp]),
    "otmb_sparse_plan_dev": (C.c_int32, [_vp, _vp, _vp, _vp, C.c_int64, C.c_int64, C.c_int64, _ip]),
    "otmb_sparse_fill_dev": (C.c_int32, [_vp, _vp, _vp, _vp]),
    "otmb_spadd_plan_dev": (C.c_int32, [_vp, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, _ip]),
    "otmb_spadd_fill_dev": (C.c_int32, [_vp, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "otmb_spadd": (C.c_int32, [_vp, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ip]),
    "otmb_transportmatrix_plan_dev": (C.c_int32, [_vp, C.POINTER(TmArgs), C.POINTER(C.c_int64 * 5)]),
    "otmb_transportmatrix_fill_dev": (C.c_int32, [_vp, C.POINTER(_vp * 5), C.POINTER(_vp * 5), C.POINTER(_vp * 5)]),
    "otmb_transportmatrix_plan": (C.c_int32, [_vp, C.POINTER(TmArgs), C.POINTER(C.c_int64 * 5)]),
    "otmb_transportmatrix_fetch": (C.c_int32, [_vp, C.POINTER(_vp * 5), C.POINTER(_vp * 5), C.POINTER(_vp * 5), C.POINTER(C.c_int64 * 5)]),
    "otmb_transportmatrix_nnz": (C.c_int32, [_vp, C.POINTER(C.c_int64 * 5)]),
}

_lib = None


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (same
    SONAME as /opt/rocm's); if libotmb_hip.so pulled in /opt/rocm's copy first and torch was
    imported afterwards, two runtimes would initialise and the second fails.  Loading torch's copy
    (when torch is installed) before anything else makes every later NEEDED entry resolve to it."""
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


_lenient = False


def use_library(path, lenient=False):
    """Switch the process to another build of the library (perf A/B of variants; contexts are per library).
    lenient: tolerate symbols the build lacks (timing an older commit with tools/ab_variants.py only)."""
    global _lib, LIB_PATH, _lenient
    LIB_PATH = path
    _lenient = lenient
    _lib = None


def lib():
    """Load libotmb_hip.so (built by build.py / __graft_entry__.build()).  Fails loudly if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError(f"{LIB_PATH} is missing: build it with `python {os.path.join(_HERE, 'build.py')}` "
                          "(there is no CPU fallback for the product path)")
        _preload_hip_runtime()
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            if _lenient and not hasattr(l, name):
                continue
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


class _PinnedOwner:
    def __init__(self, ctx, ptr):
        self.ctx, self.ptr = ctx, ptr

    def __del__(self):
        # otmb_host_free ignores its context argument: the block goes back to the process-wide pool whether or not the context
        # that allocated it is still alive, from whichever thread the garbage collector runs on
        try:
            self.ctx._lib.otmb_host_free(None, _vp(self.ptr))
        except Exception:
            pass


class Context:
    """One otmb_ctx bound to a GPU."""

    def __init__(self, device=0):
        self._h = _vp()
        self._lib = lib()  # a context belongs to the build of the library that created it (tools/ab_variants.py mixes builds)
        rc = self._lib.otmb_ctx_create(int(device), C.byref(self._h))
        if rc != OK:
            raise OtmbError(rc, f"otmb_ctx_create(device={device}) failed: " + self._lib.otmb_status_string(rc).decode())
        self.device = device

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.otmb_ctx_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def check(self, rc):
        if rc != OK:
            raise OtmbError(rc, self._lib.otmb_last_error(self._h).decode("utf-8"))

    def set_stream(self, stream_ptr):
        """Borrow a HIP stream for every _dev call.  Handle 0 is the device's default (null) stream -- torch's default
        stream -- so that the library's kernels are ordered with the caller's torch operations."""
        if not stream_ptr:
            self.check(self._lib.otmb_ctx_use_default_stream(self._h))
        else:
            self.check(self._lib.otmb_ctx_set_stream(self._h, _vp(stream_ptr)))

    def set_reuse_grid(self, on=True):
        self.check(self._lib.otmb_ctx_set_reuse_grid(self._h, int(bool(on))))

    def set_reuse_fluxes(self, on=True):
        self.check(self._lib.otmb_ctx_set_reuse_fluxes(self._h, int(bool(on))))

    def pinned_empty(self, shape, dtype, order="F"):
        """numpy array in pinned host memory of this context (otmb_host_alloc): the DMA's own target -- no staging copy and
        no page faults; the block returns to the context's pool when the array (and every view of it) is gone."""
        import numpy as np

        dt = np.dtype(dtype)
        n = int(np.prod(shape, dtype=np.int64)) if np.ndim(shape) else int(shape)
        nbytes = max(n * dt.itemsize, 1)
        ptr = _vp()
        self.check(self._lib.otmb_host_alloc(self._h, nbytes, C.byref(ptr)))
        buf = (C.c_char * nbytes).from_address(ptr.value)
        buf._otmb_owner = _PinnedOwner(self, ptr.value)  # freed when buf dies, i.e. when no array refers to it any more
        a = np.frombuffer(buf, dtype=dt, count=n)
        return a.reshape(shape, order=order) if np.ndim(shape) else a

    def set_tile_order(self, rows_per_band):
        """Speed only: 0 = tiles in wet-rank order, R > 0 = march order in bands of R rows, -1 = the library default (bands of 8 rows)."""
        self.check(self._lib.otmb_ctx_set_tile_order(self._h, int(rows_per_band)))

    def forget_given(self):
        """The verdicts on given operators (otmb_tm_args.given) are keyed to array addresses: call after rewriting such an array in place."""
        self.check(self._lib.otmb_ctx_forget_given(self._h))

    def given_state(self, m):
        """How the last plan / _dev call treated operator m (index into MATS): 0 not given, 1 given and derived, 2 given and foreign, 3 given with the derived rows and other values (another κ: read by the fill pass)."""
        return int(self._lib.otmb_ctx_given_state(self._h, int(m)))

    def use_own_stream(self):
        self.check(self._lib.otmb_ctx_set_stream(self._h, _vp(0)))

    def synchronize(self):
        self.check(self._lib.otmb_ctx_synchronize(self._h))

    def timing_enable(self, on=True):
        self.check(self._lib.otmb_ctx_timing_enable(self._h, int(on)))

    def box_ceilings(self):
        """(read GB/s, write GB/s) of one plain HBM read / non-temporal write stream on this box (otmb_ctx_box_ceilings; diagnostic)."""
        r, w = C.c_double(0), C.c_double(0)
        self.check(self._lib.otmb_ctx_box_ceilings(self._h, C.byref(r), C.byref(w)))
        return float(r.value), float(w.value)

    def stream_mix(self, inputs, outputs, tiles):
        """GB/s of an ideal streaming kernel over these arrays (otmb_ctx_stream_mix): inputs / outputs = [(device pointer, bytes)].
        DESTROYS the outputs' contents."""
        ni, no = len(inputs), len(outputs)
        ip, ib = (_vp * max(ni, 1))(*[p for p, _ in inputs]), (C.c_int64 * max(ni, 1))(*[int(b) for _, b in inputs])
        op, ob = (_vp * max(no, 1))(*[p for p, _ in outputs]), (C.c_int64 * max(no, 1))(*[int(b) for _, b in outputs])
        g = C.c_double(0)
        self.check(self._lib.otmb_ctx_stream_mix(self._h, ni, ip, ib, no, op, ob, int(tiles), C.byref(g)))
        return float(g.value)

    def timing_collect(self, n=32):
        """{kernel name: (sum_ms, launches)} since the previous collect (HIP events on the launch stream)."""
        ms = (C.c_double * n)()
        cnt = (C.c_int64 * n)()
        self.check(self._lib.otmb_ctx_timing_collect(self._h, ms, cnt, n))
        return {self._lib.otmb_kernel_name(k).decode(): (ms[k], int(cnt[k])) for k in range(n) if cnt[k]}


class Mgpu:
    """One otmb_mgpu: depth slabs over several GPUs of this process (include/otmb.h).  device_ids all different (RCCL
    hand-offs between them) or all equal (same-device copies: tests on a one-GPU box)."""

    def __init__(self, device_ids):
        self._h = _vp()
        self._lib = lib()
        ids = (C.c_int32 * len(device_ids))(*[int(d) for d in device_ids])
        rc = self._lib.otmb_mgpu_create(len(device_ids), ids, C.byref(self._h))
        if rc != OK:
            raise OtmbError(rc, f"otmb_mgpu_create({list(device_ids)}) failed: " + self._lib.otmb_status_string(rc).decode())
        self.device_ids = tuple(int(d) for d in device_ids)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.otmb_mgpu_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    @property
    def transport(self):
        return {0: "same-device copy", 1: "rccl", 2: "peer copy"}[self._lib.otmb_mgpu_transport(self._h)]

    def check(self, rc):
        if rc != OK:
            raise OtmbError(rc, self._lib.otmb_mgpu_last_error(self._h).decode("utf-8"))

    def set_reuse(self, grid=False, fluxes=False):
        self.check(self._lib.otmb_mgpu_set_reuse(self._h, int(bool(grid)), int(bool(fluxes))))

    def uploaded_bytes(self):
        return int(self._lib.otmb_mgpu_uploaded_bytes(self._h))

    def set_chain_pieces(self, pieces):
        """Speed only: row bands the facefluxes chain is handed over in (0 = the library's rule)."""
        self.check(self._lib.otmb_mgpu_set_chain_pieces(self._h, int(pieces)))

    def partition(self):
        n = self._lib.otmb_mgpu_ndev(self._h)
        b = (C.c_int64 * (n + 1))()
        self.check(self._lib.otmb_mgpu_partition(self._h, b))
        return [int(x) for x in b]


def balanced_partition(level_counts, nslabs):
    """otmb_balanced_partition: [(k0, k1)] * nslabs (pure host arithmetic inside the library: needs no GPU)."""
    counts = (C.c_int64 * len(level_counts))(*[int(x) for x in level_counts])
    bounds = (C.c_int64 * (nslabs + 1))()
    rc = lib().otmb_balanced_partition(counts, len(level_counts), int(nslabs), bounds)
    if rc != OK:
        raise ValueError(f"{nslabs} slabs for {len(level_counts)} levels")
    return [(int(bounds[r]), int(bounds[r + 1])) for r in range(nslabs)]


def ptr_array(n, ptrs):
    a = (_vp * n)()
    for k, p in enumerate(ptrs):
        a[k] = p
    return a
