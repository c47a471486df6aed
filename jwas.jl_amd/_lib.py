"""ctypes binding of libjwas_hip.so (the C ABI declared in include/jwas_hip.h).

The library is built in-tree by jwas.jl_amd/csrc/build.sh (hipcc --offload-arch=gfx950).  There is
no CPU fallback: if the shared object is missing or no gfx950 device is present, loading /
context creation raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libjwas_hip.so")

MAX_TRAITS = 4
MAX_STATES = 16

BAYESC, BAYESB, BAYESR, MTBAYESC1, MTBAYESC2, MEGABAYESC, MTBAYESB1, MTBAYESB2, MEGABAYESB = 0, 1, 2, 3, 4, 5, 6, 7, 8
GRAM_F64, GRAM_MFMA = 0, 1

# every symbol include/jwas_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "jwas_hip_create", "jwas_hip_destroy", "jwas_hip_last_error", "jwas_hip_set_stream",
    "jwas_hip_device_info", "jwas_hip_load_dense_f32", "jwas_hip_alloc_dense_f32",
    "jwas_hip_dense_layout", "jwas_hip_get_columns", "jwas_hip_estimate_bytes",
    "jwas_hip_synth_genotypes", "jwas_hip_setup_blocks", "jwas_hip_get_xpx", "jwas_hip_get_gram",
    "jwas_hip_set_gram", "jwas_hip_num_blocks", "jwas_hip_init_state", "jwas_hip_set_state",
    "jwas_hip_get_state", "jwas_hip_set_residual", "jwas_hip_get_residual", "jwas_hip_residual_dev",
    "jwas_hip_residual_to_dev", "jwas_hip_residual_from_dev",
    "jwas_hip_residual_sub_xalpha", "jwas_hip_mul_alpha", "jwas_hip_load_output_dense_f32", "jwas_hip_mul_alpha_output", "jwas_hip_window_sums", "jwas_hip_window_sums2", "jwas_hip_set_kernel_timing", "jwas_hip_sweep", "jwas_hip_last_sweep_counters",
    "jwas_hip_accumulate", "jwas_hip_get_posterior",
    "jwas_hip_load_jgb2", "jwas_hip_load_packed2bit", "jwas_hip_alloc_packed2bit", "jwas_hip_storage_info",
    "jwas_hip_set_xpx", "jwas_hip_estimate_bytes_storage", "jwas_hip_add_block_size", "jwas_hip_select_block_size",
    "jwas_hip_set_weights", "jwas_hip_set_weights_f64", "jwas_hip_synth_single_step", "jwas_hip_setup_blocks_explicit",
    "jwas_hip_comm_row_shards", "jwas_hip_comm_init_loopback", "jwas_hip_update_geometry", "jwas_hip_set_cross_gram",
    "jwas_hip_set_columns", "jwas_hip_get_alpha_sparse", "jwas_hip_comm_unique_id", "jwas_hip_comm_init", "jwas_hip_comm_destroy", "jwas_hip_sweep_sharded",
    "jwas_hip_residual_add_scalar", "jwas_hip_comm_info", "jwas_hip_sample_marker_covariances", "jwas_hip_get_marker_covariances",
    "jwas_hip_set_precision", "jwas_hip_load_dense_f64", "jwas_hip_get_xpx_f64", "jwas_hip_set_state_f64", "jwas_hip_get_state_f64",
    "jwas_hip_set_residual_f64", "jwas_hip_get_residual_f64", "jwas_hip_mul_alpha_f64", "jwas_hip_get_posterior_f64",
    "jwas_hip_setup_groups",
]
STORAGE_DENSE_F32, STORAGE_PACKED2BIT = 0, 1


class SweepParams(C.Structure):
    _fields_ = [
        ("method", C.c_int32), ("ntraits", C.c_int32), ("nreps", C.c_int32), ("iteration", C.c_uint32),
        ("seed", C.c_uint64), ("marker_offset", C.c_uint32), ("independent_blocks", C.c_uint32),
        ("vare", C.c_float * (MAX_TRAITS * MAX_TRAITS)),
        ("var_effect", C.c_float * (MAX_TRAITS * MAX_TRAITS)),
        ("pi", C.c_double), ("pi_classes", C.c_double * 4), ("gamma", C.c_double * 4),
        ("log_prior_states", C.c_double * MAX_STATES),
        ("var_effect_vec", C.POINTER(C.c_float)),
        ("pi_vec", C.POINTER(C.c_double)),
        ("pi_matrix", C.POINTER(C.c_double)),
        ("log_prior_states_matrix", C.POINTER(C.c_double)),
        ("var_effect_matrix", C.POINTER(C.c_float)),
        ("vare_f64", C.c_double * (MAX_TRAITS * MAX_TRAITS)),
        ("var_effect_f64", C.c_double * (MAX_TRAITS * MAX_TRAITS)),
        ("var_effect_vec_f64", C.POINTER(C.c_double)),
        ("section_solve", C.c_int32),
        ("group_launch", C.c_int32),
    ]


class SweepStats(C.Structure):
    _fields_ = [
        ("sum_delta", C.c_double * MAX_TRAITS),
        ("alpha_ss", C.c_double * (MAX_TRAITS * MAX_TRAITS)),
        ("beta_ss", C.c_double * (MAX_TRAITS * MAX_TRAITS)),
        ("resid_ss", C.c_double * (MAX_TRAITS * MAX_TRAITS)),
        ("resid_sum", C.c_double * MAX_TRAITS),
        ("class_counts", C.c_double * 4),
        ("bayesr_ssq", C.c_double), ("bayesr_nnz", C.c_double),
        ("state_counts", C.c_double * MAX_STATES),
        ("n_events", C.c_double), ("sweep_ms", C.c_double),
        ("update_kernel_ms", C.c_double), ("update_kernel_samples", C.c_double), ("update_kernel_bytes", C.c_double),
        ("event_overhead_ms", C.c_double),
    ]


class JwasHipError(RuntimeError):
    """Raised for any non-zero status of the C ABI (the analogue of the reference's error(...))."""

    def __init__(self, code, message):
        super().__init__(f"libjwas_hip error {code}: {message}")
        self.code = code
        self.message = message


_lib = None


def load():
    """Load libjwas_hip.so and declare prototypes.  Fails loudly if the HIP library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with jwas.jl_amd/csrc/build.sh (hipcc, gfx950). "
            "The MI355X path has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u64, f32p, f64p = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.POINTER(C.c_float), C.POINTER(C.c_double)
    L.jwas_hip_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.jwas_hip_destroy.argtypes = [vp]
    L.jwas_hip_destroy.restype = None
    L.jwas_hip_last_error.argtypes = [vp]
    L.jwas_hip_last_error.restype = C.c_char_p
    L.jwas_hip_set_stream.argtypes = [vp, vp]
    L.jwas_hip_device_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(i64), C.POINTER(i64)]
    L.jwas_hip_load_dense_f32.argtypes = [vp, vp, i64, i64, i64]
    L.jwas_hip_alloc_dense_f32.argtypes = [vp, i64, i64]
    L.jwas_hip_dense_layout.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(vp)]
    L.jwas_hip_get_columns.argtypes = [vp, i64, i64, vp]
    L.jwas_hip_set_columns.argtypes = [vp, i64, i64, vp, i64]
    L.jwas_hip_get_alpha_sparse.argtypes = [vp, i32, i64, vp, vp, C.POINTER(i64)]
    L.jwas_hip_estimate_bytes.argtypes = [i64, i64, i32, i32]
    L.jwas_hip_estimate_bytes.restype = i64
    L.jwas_hip_synth_genotypes.argtypes = [vp, u64, i32, i32, i64]
    L.jwas_hip_synth_single_step.argtypes = [vp, u64, i64, i32, i64]
    L.jwas_hip_setup_blocks.argtypes = [vp, i32, i32]
    L.jwas_hip_get_xpx.argtypes = [vp, vp]
    L.jwas_hip_get_gram.argtypes = [vp, i64, vp]
    L.jwas_hip_set_gram.argtypes = [vp, i64, vp]
    L.jwas_hip_num_blocks.argtypes = [vp, C.POINTER(i64), C.POINTER(i32)]
    L.jwas_hip_init_state.argtypes = [vp, i32, i32]
    L.jwas_hip_set_state.argtypes = [vp, i32, vp, vp, vp]
    L.jwas_hip_get_state.argtypes = [vp, i32, vp, vp, vp]
    L.jwas_hip_set_residual.argtypes = [vp, i32, vp]
    L.jwas_hip_get_residual.argtypes = [vp, i32, vp]
    L.jwas_hip_residual_dev.argtypes = [vp, C.POINTER(vp), C.POINTER(i64)]
    L.jwas_hip_residual_to_dev.argtypes = [vp, i32, vp]
    L.jwas_hip_residual_from_dev.argtypes = [vp, i32, vp]
    L.jwas_hip_set_kernel_timing.argtypes = [vp, i32]
    L.jwas_hip_last_sweep_counters.argtypes = [vp, C.POINTER(C.c_uint64), i32]
    L.jwas_hip_residual_sub_xalpha.argtypes = [vp, i32]
    L.jwas_hip_mul_alpha.argtypes = [vp, i32, vp]
    L.jwas_hip_load_output_dense_f32.argtypes = [vp, vp, i64, i64, i64]
    L.jwas_hip_mul_alpha_output.argtypes = [vp, i32, vp]
    L.jwas_hip_window_sums.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp]
    L.jwas_hip_window_sums2.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.jwas_hip_sweep.argtypes = [vp, C.POINTER(SweepParams), C.POINTER(SweepStats)]
    L.jwas_hip_sweep_sharded.argtypes = [vp, C.POINTER(SweepParams), C.POINTER(SweepStats)]
    L.jwas_hip_comm_unique_id.argtypes = [vp]
    L.jwas_hip_comm_init.argtypes = [vp, vp, i32, i32]
    L.jwas_hip_comm_destroy.argtypes = [vp]
    L.jwas_hip_comm_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.jwas_hip_residual_add_scalar.argtypes = [vp, i32, C.c_double]
    L.jwas_hip_sample_marker_covariances.argtypes = [vp, C.c_double, f64p, u64, C.c_uint32, C.c_uint32]
    L.jwas_hip_get_marker_covariances.argtypes = [vp, vp]
    L.jwas_hip_set_precision.argtypes = [vp, i32]
    L.jwas_hip_load_dense_f64.argtypes = [vp, vp, i64, i64, i64]
    L.jwas_hip_get_xpx_f64.argtypes = [vp, vp]
    L.jwas_hip_set_state_f64.argtypes = [vp, i32, vp, vp, vp]
    L.jwas_hip_get_state_f64.argtypes = [vp, i32, vp, vp, vp]
    L.jwas_hip_set_residual_f64.argtypes = [vp, i32, vp]
    L.jwas_hip_get_residual_f64.argtypes = [vp, i32, vp]
    L.jwas_hip_mul_alpha_f64.argtypes = [vp, i32, vp]
    L.jwas_hip_get_posterior_f64.argtypes = [vp, i32, vp, vp, vp]
    L.jwas_hip_accumulate.argtypes = [vp, C.c_double]
    L.jwas_hip_get_posterior.argtypes = [vp, i32, vp, vp, vp]
    L.jwas_hip_load_jgb2.argtypes = [vp, C.c_char_p]
    L.jwas_hip_load_packed2bit.argtypes = [vp, vp, i64, i64, i64, vp, i32]
    L.jwas_hip_alloc_packed2bit.argtypes = [vp, i64, i64, i32]
    L.jwas_hip_storage_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]
    L.jwas_hip_set_xpx.argtypes = [vp, vp]
    L.jwas_hip_set_weights.argtypes = [vp, vp]
    L.jwas_hip_set_weights_f64.argtypes = [vp, vp]
    L.jwas_hip_add_block_size.argtypes = [vp, i32, i32]
    L.jwas_hip_setup_blocks_explicit.argtypes = [vp, C.POINTER(C.c_int64), i64, i32]
    L.jwas_hip_comm_row_shards.argtypes = [vp, i32]
    L.jwas_hip_update_geometry.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.jwas_hip_set_cross_gram.argtypes = [vp, i64, vp]
    L.jwas_hip_comm_init_loopback.argtypes = [vp, i32, i32, i32]
    L.jwas_hip_select_block_size.argtypes = [vp, i32]
    L.jwas_hip_setup_groups.argtypes = [vp, i32, i32]
    L.jwas_hip_estimate_bytes_storage.argtypes = [i64, i64, i32, i32, i32]
    L.jwas_hip_estimate_bytes_storage.restype = i64
    for name in SYMBOLS:
        fn = getattr(L, name)
        if name not in ("jwas_hip_destroy", "jwas_hip_last_error", "jwas_hip_estimate_bytes", "jwas_hip_estimate_bytes_storage"):
            fn.restype = C.c_int
    _lib = L
    return L
