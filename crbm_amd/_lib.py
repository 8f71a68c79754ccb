"""ctypes binding of the C-ABI in include/crbm_amd.h.

The shared library is built in-tree (crbm_amd/csrc/libcrbm_hip.so, see
crbm_amd/csrc/build.py).  There is no CPU fallback: if the library is missing
or no GPU is visible, using it fails loudly.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libcrbm_hip.so")

UNIQUE_ID_BYTES = 128
IPC_HANDLE_BYTES = 64
ABI_VERSION = 3          # CRBM_AMD_ABI_VERSION of include/crbm_amd.h

CRBM_OK = 0
ERR_INVALID, ERR_HIP, ERR_NOT_ONEHOT, ERR_NOT_BINARY, ERR_RCCL, ERR_NO_GPU, ERR_IPC_TIMEOUT = -1, -2, -3, -4, -5, -6, -7


class CrbmConfig(ctypes.Structure):
    """struct crbm_config (include/crbm_amd.h)."""
    _fields_ = [
        ("num_motifs", ctypes.c_int32),
        ("motif_length", ctypes.c_int32),
        ("input_dims", ctypes.c_int32),
        ("doublestranded", ctypes.c_int32),
        ("batchsize", ctypes.c_int32),
        ("cd_k", ctypes.c_int32),
        ("pooling", ctypes.c_int32),
        ("fantasy_hidden_len", ctypes.c_int32),
        ("learning_rate", ctypes.c_float),
        ("momentum", ctypes.c_float),
        ("rho", ctypes.c_float),
        ("lambda_rate", ctypes.c_float),
        ("seed", ctypes.c_uint64),
        ("device", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
    ]


class CrbmLaunchInfo(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "nq", "group", "gibbs_grid", "gibbs_block", "gibbs_seqs_per_tile", "gibbs_lds_bytes",
        "stats_grid_x", "stats_grid_y", "stats_block", "stats_lds_bytes", "gibbs_sparse", "activity_ppm", "stats_fused", "chain_parts")]


_H = ctypes.c_void_p
_F = ctypes.POINTER(ctypes.c_float)
_I32 = ctypes.c_int32
_U32 = ctypes.c_uint32
_U64 = ctypes.c_uint64
_U8P = ctypes.POINTER(ctypes.c_uint8)

# name -> (restype, argtypes); every symbol declared in include/crbm_amd.h
SIGNATURES = {
    "crbm_create": (_I32, [ctypes.POINTER(CrbmConfig), ctypes.POINTER(_H)]),
    "crbm_precompile": (_I32, [ctypes.POINTER(CrbmConfig)]),
    "crbm_destroy": (_I32, [_H]),
    "crbm_last_error": (ctypes.c_char_p, [_H]),
    "crbm_abi_version": (_I32, []),
    "crbm_device_count": (_I32, []),
    "crbm_set_params": (_I32, [_H, _F, _F, _F]),
    "crbm_get_params": (_I32, [_H, _F, _F, _F]),
    "crbm_set_velocities": (_I32, [_H, _F, _F, _F]),
    "crbm_get_velocities": (_I32, [_H, _F, _F, _F]),
    "crbm_set_fantasy": (_I32, [_H, _F, _F]),
    "crbm_get_fantasy": (_I32, [_H, _F, _F]),
    "crbm_get_fantasy_visible": (_I32, [_H, _F]),
    "crbm_set_rng": (_I32, [_H, _U64, _U32, _U32]),
    "crbm_get_rng": (_I32, [_H, ctypes.POINTER(_U64), ctypes.POINTER(_U32), ctypes.POINTER(_U32)]),
    "crbm_set_shard": (_I32, [_H, _U32]),
    "crbm_train_step": (_I32, [_H, _F, _I32, _I32]),
    "crbm_dataset_upload": (_I32, [_H, _F, _I32, _I32]),
    "crbm_dataset_upload_codes": (_I32, [_H, _U8P, _I32, _I32]),
    "crbm_dataset_select": (_I32, [_H, _I32]),
    "crbm_train_step_resident": (_I32, [_H, _I32, _I32]),
    "crbm_train_epoch_resident": (_I32, [_H, _I32]),
    "crbm_train_epoch_sharded": (_I32, [_H, _I32, _I32, _I32]),
    "crbm_gibbs_steps": (_I32, [_H, _I32]),
    "crbm_gibbs_steps_async": (_I32, [_H, _I32]),
    "crbm_sync": (_I32, [_H]),
    "crbm_wait_idle": (_I32, [_H]),
    "crbm_time_gibbs": (_I32, [_H, _I32, _I32, _F]),
    "crbm_time_train": (_I32, [_H, _I32, _I32, _I32, _F]),
    "crbm_h_given_v": (_I32, [_H, _F, _I32, _I32, _I32, _U32, _F, _F, _F]),
    "crbm_v_given_h": (_I32, [_H, _F, _F, _I32, _I32, _U32, _F, _F, _F]),
    "crbm_hit_probs": (_I32, [_H, _F, _I32, _I32, _F]),
    "crbm_free_energy": (_I32, [_H, _F, _I32, _I32, _F]),
    "crbm_free_energy_per_motif": (_I32, [_H, _F, _I32, _I32, _F]),
    "crbm_eval_data": (_I32, [_H, _F, _I32, _I32, _F, _F]),
    "crbm_eval_params": (_I32, [_H, _F, _F, _F]),
    "crbm_hit_probs_codes": (_I32, [_H, _U8P, _I32, _I32, _F]),
    "crbm_hit_probs_resident": (_I32, [_H, _I32, _I32, _F]),
    "crbm_free_energy_codes": (_I32, [_H, _U8P, _I32, _I32, _F, _F]),
    "crbm_free_energy_resident": (_I32, [_H, _I32, _I32, _F, _F]),
    "crbm_eval_data_resident": (_I32, [_H, _I32, _I32, _F, _F]),
    "crbm_eval_epoch_resident": (_I32, [_H, _I32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "crbm_hit_summary": (_I32, [_H, _F, _I32, _I32, _F, _F, _F]),
    "crbm_hit_summary_codes": (_I32, [_H, _U8P, _I32, _I32, _F, _F, _F]),
    "crbm_hit_summary_resident": (_I32, [_H, _I32, _I32, _F, _F, _F]),
    "crbm_comm_unique_id": (_I32, [_U8P]),
    "crbm_comm_init": (_I32, [_H, _U8P, _I32, _I32]),
    "crbm_comm_destroy": (_I32, [_H]),
    "crbm_comm_broadcast_state": (_I32, [_H, _I32]),
    "crbm_ipc_export": (_I32, [_H, _U8P]),
    "crbm_ipc_attach": (_I32, [_H, _U8P, _I32, _I32]),
    "crbm_ipc_detach": (_I32, [_H]),
    "crbm_ipc_status": (_I32, [_H, ctypes.POINTER(_I32)]),
    "crbm_sums_count": (_I32, [_H]),
    "crbm_train_local": (_I32, [_H, _F, _I32, _I32, _F]),
    "crbm_train_apply": (_I32, [_H, _F, _I32]),
    "crbm_time_allreduce": (_I32, [_H, _I32, _F]),
    "crbm_get_launch_info": (_I32, [_H, ctypes.POINTER(CrbmLaunchInfo)]),
    "crbm_copy_bandwidth": (_I32, [_H, ctypes.c_int64, _I32, _F]),
    "crbm_last_shader_clock": (_I32, [_H, _F]),
    "crbm_gibbs_state_bytes": (ctypes.c_int64, [_H]),
}

_lib = None


class CrbmLibraryError(RuntimeError):
    pass


def load():
    """Load libcrbm_hip.so and bind every declared symbol (no GPU needed)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CrbmLibraryError(
            "HIP extension not built: %s is missing. Run `python -m crbm_amd.csrc.build` "
            "(or __graft_entry__.build()). There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.crbm_abi_version() != ABI_VERSION:
        raise CrbmLibraryError("ABI version mismatch")
    _lib = lib
    return lib


def fptr(arr):
    """float32 C-contiguous ndarray -> float* (None -> NULL)."""
    if arr is None:
        return None
    assert arr.dtype == np.float32 and arr.flags["C_CONTIGUOUS"]
    return arr.ctypes.data_as(_F)


def as_f32(x):
    """Accept float64/float32/ints like the reference's floatX cast; C-contiguous float32 out."""
    return np.ascontiguousarray(x, dtype=np.float32)
