"""ctypes binding of libnbk.so (include/nbk.h).

The product path has NO CPU fallback: if the library is missing, or no GPU is visible when a compute
entry point is called, this module raises.  (Loading the library and reading its symbols works without
a GPU, which is what the CPU-side tests check.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libnbk.so")

NBK_OK = 0
STATUS = {0: "NBK_OK", -1: "NBK_ERR_INVALID", -2: "NBK_ERR_NO_DEVICE", -3: "NBK_ERR_HIP",
          -4: "NBK_ERR_UNSUPPORTED", -5: "NBK_ERR_ALLOC"}

# every symbol include/nbk.h declares
SYMBOLS = [
    "nbk_abi_version", "nbk_status_string", "nbk_last_error", "nbk_device_count",
    "nbk_model_create", "nbk_model_destroy", "nbk_model_num_pairs",
    "nbk_fk_batch", "nbk_frameset_create", "nbk_frameset_destroy", "nbk_fk_frames_batch", "nbk_jacobian_batch", "nbk_ik_batch", "nbk_validity_batch", "nbk_validity_workspace_bytes",
    "nbk_validity_batch_ws", "nbk_closest_batch",
    "nbk_pair_distances_batch", "nbk_proximity_jacobian_batch", "nbk_edge_validity_batch", "nbk_selftest_math",
    "nbk_fk_batch_host", "nbk_validity_batch_host", "nbk_knn_prefix",
    "nbk_validity_scalar_host", "nbk_edge_validity_scalar_host",
]


class ModelDesc(C.Structure):
    _fields_ = [
        ("n_q", C.c_int32), ("n_joints", C.c_int32),
        ("joint_parent", C.c_void_p), ("joint_type", C.c_void_p), ("joint_qidx", C.c_void_p),
        ("joint_rot", C.c_void_p), ("joint_trans", C.c_void_p), ("joint_slide", C.c_void_p),
        ("joint_axis", C.c_void_p), ("base_pose", C.c_void_p),
        ("n_rshapes", C.c_int32),
        ("rshape_frame", C.c_void_p), ("rshape_type", C.c_void_p), ("rshape_local", C.c_void_p),
        ("rshape_param", C.c_void_p),
        ("n_wshapes", C.c_int32),
        ("wshape_type", C.c_void_p), ("wshape_pose", C.c_void_p), ("wshape_param", C.c_void_p),
        ("n_pairs", C.c_int32),
        ("pair_a", C.c_void_p), ("pair_b", C.c_void_p),
        ("n_hulls", C.c_int32),
        ("hull_vert_begin", C.c_void_p), ("hull_verts", C.c_void_p),
        ("hull_face_begin", C.c_void_p), ("hull_planes", C.c_void_p),
    ]


class NbkError(RuntimeError):
    pass


_lib = None


def load():
    """Load libnbk.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NbkError(
            f"{LIB_PATH} is missing: build it with `python -m numbotics_amd.csrc.build` "
            "(or __graft_entry__.build()).  There is no CPU fallback for the device path.")
    # PyTorch-ROCm ships its own HIP runtime: load it first so that libnbk.so binds to the runtime torch uses (loaded the other
    # way round the process ends up with two runtimes and torch sees no device)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    lib.nbk_status_string.restype = C.c_char_p
    lib.nbk_status_string.argtypes = [C.c_int32]
    lib.nbk_last_error.restype = C.c_char_p
    lib.nbk_model_destroy.restype = None
    lib.nbk_model_destroy.argtypes = [C.c_void_p]
    lib.nbk_model_create.argtypes = [C.POINTER(ModelDesc), C.POINTER(C.c_void_p)]
    vp, i32, i64, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    lib.nbk_fk_batch.argtypes = [vp, vp, i64, vp, i32, vp, vp, vp, vp]
    lib.nbk_jacobian_batch.argtypes = [vp, vp, i64, vp, i32, vp, i32, vp, vp, vp]
    lib.nbk_frameset_create.argtypes = [vp, i32, vp, vp, vp]
    lib.nbk_frameset_destroy.argtypes = [vp]
    lib.nbk_frameset_destroy.restype = None
    lib.nbk_fk_frames_batch.argtypes = [vp, vp, vp, i64, vp, vp]
    lib.nbk_ik_batch.argtypes = [vp, vp, vp, i64, vp, i32, vp, vp, f64, i32, i32, vp, vp, vp, vp, vp]
    lib.nbk_validity_batch.argtypes = [vp, vp, i64, f64, vp, vp, vp]
    lib.nbk_validity_workspace_bytes.argtypes = [vp, i64]
    lib.nbk_validity_workspace_bytes.restype = i64
    lib.nbk_validity_batch_ws.argtypes = [vp, vp, i64, f64, vp, vp, vp, i64, vp]
    lib.nbk_closest_batch.argtypes = [vp, vp, i64, vp, vp, vp]
    lib.nbk_pair_distances_batch.argtypes = [vp, vp, i64, vp, vp, vp]
    lib.nbk_proximity_jacobian_batch.argtypes = [vp, vp, i64, vp, vp, vp, vp]
    lib.nbk_edge_validity_batch.argtypes = [vp, vp, vp, vp, i64, f64, f64, i32, f64, vp, vp, vp, vp]
    lib.nbk_selftest_math.argtypes = [vp, vp, i64, vp, vp, vp, vp, vp]
    lib.nbk_knn_prefix.argtypes = [vp, i32, i32, i32, vp, vp]
    lib.nbk_fk_batch_host.argtypes = [vp, vp, i64, vp, i32, vp, vp]
    lib.nbk_validity_batch_host.argtypes = [vp, vp, i64, f64, vp]
    lib.nbk_model_num_pairs.argtypes = [vp]
    lib.nbk_validity_scalar_host.argtypes = [vp, vp, f64, vp]
    lib.nbk_edge_validity_scalar_host.argtypes = [vp, vp, vp, f64, f64, f64, i32, f64, vp, vp, vp]
    lib.nbk_debug_set_option.argtypes = [C.c_char_p, i64]
    lib.nbk_debug_narrow_variant.argtypes = [vp, f64]
    _lib = lib
    return lib


# tuning / diagnostic switches of the library (process-wide; seeded once from the NBK_* environment variables when the library
# is loaded, never read again): name -> default.  None of them changes a result.
DEBUG_OPTIONS = {"two_kernel_min_b": 1, "edge_batch_min_e": 1, "no_reg_broad": 0, "f64_broad": 0, "jac_two_sweep": 0,
                 "closest_brute": 0, "narrow_parts_max": 16, "queue_budget": 1 << 30, "pipeline_tiles": 1, "pipe_tile": 1 << 20, "fk_lds_q": 0}


def set_debug_option(name: str, value: int):
    check(load().nbk_debug_set_option(name.encode(), int(value)), f"nbk_debug_set_option({name})")


class debug_option:
    """``with debug_option("two_kernel_min_b", 10**9): ...`` -- route calls through another kernel path (tests, tools)."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        set_debug_option(self.name, self.value)

    def __exit__(self, *exc):
        set_debug_option(self.name, DEBUG_OPTIONS[self.name])
        return False


def check(status: int, what: str):
    if status != NBK_OK:
        lib = load()
        msg = lib.nbk_status_string(status).decode()
        detail = lib.nbk_last_error().decode() if status == -3 else ""
        raise NbkError(f"{what} failed: {STATUS.get(status, status)} ({msg}) {detail}".strip())
