"""ctypes binding of include/hehub_amd.h.

The library is the product; this module only loads it and declares signatures.
It raises loudly when the shared library is missing -- there is no CPU or
PyTorch fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# HEHUB_AMD_LIB selects another build of the SAME engine (kernel-tuning experiments); never a fallback
LIB_PATH = os.environ.get("HEHUB_AMD_LIB") or os.path.join(HERE, "lib", "libhehub_amd.so")

u64 = C.c_uint64
szt = C.c_size_t
P = C.c_void_p
INT = C.c_int

HP_OK, HP_EINVAL, HP_EUNSUPPORTED, HP_EHIP, HP_ENOMEM, HP_ELOGIC, HP_ERANGE = range(7)

# name -> (restype, argtypes); mirrors include/hehub_amd.h one to one
SIGNATURES = {
    "hp_ctx_create": (INT, [INT, C.POINTER(P)]),
    "hp_ctx_destroy": (None, [P]),
    "hp_last_error": (C.c_char_p, [P]),
    "hp_version": (C.c_char_p, []),
    "hp_ctx_fork": (INT, [P, C.POINTER(P)]),
    "hp_ctx_wait_for": (INT, [P, P]),
    "hp_ctx_set_stream": (INT, [P, P]),
    "hp_ctx_reset_stream": (INT, [P]),
    "hp_ctx_get_stream": (P, [P]),
    "hp_sync": (INT, [P]),
    "hp_ctx_workspace_bytes": (szt, [P]),
    "hp_ctx_workspace_generation": (C.c_ulong, [P]),
    "hp_ctx_release_workspace": (INT, [P]),
    "hp_dev_alloc": (INT, [P, szt, C.POINTER(P)]),
    "hp_dev_free": (INT, [P, P]),
    "hp_host_alloc": (INT, [P, szt, C.POINTER(P)]),
    "hp_host_free": (INT, [P, P]),
    "hp_memcpy_h2d": (INT, [P, P, P, szt]),
    "hp_memcpy_d2h": (INT, [P, P, P, szt]),
    "hp_host_register": (INT, [P, P, szt]),
    "hp_host_unregister": (INT, [P, P]),
    "hp_memcpy_h2d_async": (INT, [P, P, P, szt]),
    "hp_memcpy_d2h_async": (INT, [P, P, P, szt]),
    "hp_memcpy_peer_async": (INT, [P, P, P, P, szt]),
    "hp_ctx_device": (INT, [P]),
    "hp_dev_store_host_rows": (INT, [P, szt, szt, P, P]),
    "hp_dev_load_host_rows": (INT, [P, szt, szt, P, P]),
    "hp_dev_gather_rows": (INT, [P, szt, szt, P, P]),
    "hp_dev_scatter_rows": (INT, [P, szt, szt, P, P]),
    "hp_dev_poly_fold_rows": (INT, [P, szt, szt, P, szt, szt, P, P, P]),
    "hp_ctx_set_force_generic": (INT, [P, INT]),
    "hp_ctx_set_parity_level": (INT, [P, INT]),
    "hp_ctx_get_parity_level": (INT, [P]),
    "hp_ntt_negacyclic_inplace_lazy": (INT, [P, szt, u64, P]),
    "hp_intt_negacyclic_inplace_lazy": (INT, [P, szt, u64, P]),
    "hp_cache_ntt_factors_strict": (INT, [P, szt, P, szt]),
    "hp_check_chain": (INT, [P, szt, P, szt, INT]),
    "hp_batched_barrett_lazy": (INT, [P, u64, szt, P]),
    "hp_batched_barrett": (INT, [P, u64, szt, P]),
    "hp_batched_reduce_strict": (INT, [P, u64, szt, P]),
    "hp_batched_mul_mod_hybrid_lazy": (INT, [P, u64, szt, P, P, P]),
    "hp_batched_mul_mod_barrett_lazy": (INT, [P, u64, szt, P, P, P]),
    "hp_batched_montgomery_128_lazy": (INT, [P, u64, szt, P, P]),
    "hp_dev_ntt": (INT, [P, szt, szt, P, szt, P]),
    "hp_dev_intt": (INT, [P, szt, szt, P, szt, P, INT]),
    "hp_dev_ntt_residues": (INT, [P, szt, szt, P, szt, P]),
    "hp_dev_intt_residues": (INT, [P, szt, szt, P, szt, P]),
    "hp_dev_poly_add": (INT, [P, szt, szt, P, szt, P, P, P]),
    "hp_dev_poly_sub": (INT, [P, szt, szt, P, szt, P, P, P]),
    "hp_dev_poly_mul": (INT, [P, szt, szt, P, szt, P, P, P]),
    "hp_dev_poly_scalar_mul": (INT, [P, szt, szt, P, szt, P, P, P]),
    "hp_dev_poly_reduce_strict": (INT, [P, szt, szt, P, szt, P]),
    "hp_dev_copy": (INT, [P, szt, P, P]),
    "hp_dev_poly_involution": (INT, [P, szt, szt, szt, P, P]),
    "hp_dev_poly_cycle": (INT, [P, szt, szt, szt, szt, P, P]),
    "hp_dev_mult_low_level": (INT, [P, szt, szt, P, szt, P, P, P]),
    "hp_dev_ext_prod_montgomery": (INT, [P, szt, szt, P, szt, P, P, P]),
    "hp_dev_ckks_rescale": (INT, [P, szt, szt, P, szt, P, P]),
    "hp_dev_bgv_mod_switch": (INT, [P, szt, szt, P, u64, szt, P, P]),
    "hp_dev_ckks_relinearize": (INT, [P, szt, szt, P, szt, P, P, P]),
    "hp_dev_bgv_relinearize": (INT, [P, szt, szt, P, u64, szt, P, P, P]),
    "hp_dev_ckks_rotate": (INT, [P, szt, szt, P, szt, szt, P, P, P]),
    "hp_dev_ckks_conjugate": (INT, [P, szt, szt, P, szt, P, P, P]),
    "hp_dev_rlwe_encrypt_core": (INT, [P, szt, szt, P, szt, P, P, P, P, P]),
    "hp_dev_rlwe_decrypt_core": (INT, [P, szt, szt, P, szt, P, P, P]),
    "hp_dev_rns_base_from_single": (INT, [P, szt, u64, szt, P, szt, P, P]),
    "hp_dev_rns_base_to_single_small": (INT, [P, szt, szt, P, u64, szt, P, P, P]),
    "hp_dev_rns_base_to_single": (INT, [P, szt, szt, P, u64, szt, P, P]),
    "hp_dev_mult_low_level_range": (INT, [P, szt, szt, P, szt, szt, szt, P, P, P]),
    "hp_dev_ks_coef_range": (INT, [P, szt, szt, P, szt, szt, szt, P, szt, P]),
    "hp_dev_ks_inner_range": (INT, [P, szt, szt, P, szt, szt, szt, P, P, szt, P, P]),
    "hp_dev_ks_inner_range_strict": (INT, [P, szt, szt, P, szt, szt, szt, P, P, szt, P, P]),
    "hp_dev_drop_coeffs": (INT, [P, szt, szt, P, u64, szt, P, P]),
    "hp_dev_drop_apply_range": (INT, [P, szt, szt, P, u64, szt, szt, szt, P, P, P, szt, szt, C.c_uint, P]),
    "hp_dev_drop_apply_range_strict": (INT, [P, szt, szt, P, u64, szt, szt, szt, P, P, P, szt, szt, C.c_uint, P]),
    "hp_node_create": (INT, [P, szt, C.POINTER(P)]),
    "hp_node_destroy": (None, [P]),
    "hp_node_size": (szt, [P]),
    "hp_node_ctx": (P, [P, szt]),
    "hp_node_set_parity_level": (INT, [P, INT]),
    "hp_node_last_error": (C.c_char_p, [P]),
    "hp_node_placement": (INT, [P, szt, P, P]),
    "hp_device_numa": (INT, [INT, P, P, szt]),
    "hp_node_peer_matrix": (INT, [P, P]),
    "hp_node_set_transport": (INT, [P, INT]),
    "hp_node_get_transport": (INT, [P]),
    "hp_node_slice": (INT, [P, szt, szt, C.POINTER(szt), C.POINTER(szt)]),
    "hp_node_sync": (INT, [P]),
    "hp_node_replicate": (INT, [P, P, szt, P]),
    "hp_node_free_replicas": (INT, [P, P]),
    "hp_node_ckks_mult_relin_rescale": (INT, [P, szt, szt, P, szt, P, P, P, P]),
    "hp_node_bgv_mult_relin_modswitch": (INT, [P, szt, szt, P, u64, szt, P, P, P, P]),
    "hp_node_ntt": (INT, [P, szt, szt, P, szt, P, INT, INT]),
    "hp_node_dev_ckks_mult_relin_rescale": (INT, [P, szt, szt, P, P, P, P, P, P]),
    "hp_node_dev_bgv_mult_relin_modswitch": (INT, [P, szt, szt, P, u64, P, P, P, P, P]),
    "hp_node_sharded_create": (INT, [P, szt, szt, P, u64, szt, C.POINTER(P)]),
    "hp_node_sharded_destroy": (None, [P]),
    "hp_node_sharded_range": (INT, [P, szt, C.POINTER(szt), C.POINTER(szt)]),
    "hp_node_sharded_mult": (INT, [P, P, P, P, P]),
    "hp_node_sharded_mult_dev": (INT, [P, P, P, P, P]),
    "hp_dev_ckks_mult_relin_rescale": (INT, [P, szt, szt, P, szt, P, P, P, P]),
    "hp_dev_bgv_mult_relin_modswitch": (INT, [P, szt, szt, P, u64, szt, P, P, P, P]),
    "hp_dev_bgv_mult_relin_modswitch_t": (INT, [P, szt, szt, P, u64, szt, P, P, P, P]),
    "hp_dev_ext_prod_montgomery_at": (INT, [P, szt, szt, szt, P, szt, P, P, P]),
    "hp_dev_ckks_relinearize_at": (INT, [P, szt, szt, szt, P, szt, P, P, P]),
    "hp_dev_ckks_rotate_at": (INT, [P, szt, szt, szt, P, szt, szt, P, P, P]),
    "hp_dev_ckks_rotate_many": (INT, [P, szt, szt, szt, P, szt, P, P, P, P, P]),
    "hp_dev_ckks_rotate_many_rows": (INT, [P, szt, szt, szt, P, szt, P, P, P, P, P]),
    "hp_dev_ckks_mult_relin_rescale_rows": (INT, [P, szt, szt, szt, P, szt, P, P, P]),
    "hp_dev_bgv_mult_relin_modswitch_rows": (INT, [P, szt, szt, P, u64, szt, P, P, P]),
    "hp_dev_ckks_conjugate_at": (INT, [P, szt, szt, szt, P, szt, P, P, P]),
    "hp_dev_ckks_mult_relin_rescale_at": (INT, [P, szt, szt, szt, P, szt, P, P, P, P]),
    "hp_dev_rns_base_many_to_many": (INT, [P, szt, szt, P, szt, P, szt, P, P]),
    "hp_dev_hks_switch": (INT, [P, szt, szt, szt, szt, P, szt, P, P, P]),
    "hp_dev_ckks_rotate_hks": (INT, [P, szt, szt, szt, szt, P, szt, szt, P, P, P]),
    "hp_dev_ckks_conjugate_hks": (INT, [P, szt, szt, szt, szt, P, szt, P, P, P]),
    "hp_dev_ckks_mult_relin_rescale_hks": (INT, [P, szt, szt, szt, szt, P, szt, P, P, P, P]),
    "hp_dev_ckks_rescale_n": (INT, [P, szt, szt, P, szt, szt, P, P, P]),
    "hp_wire_fnv1a64": (u64, [P, szt]),
    "hp_wire_payload_words": (szt, [P]),
    "hp_wire_bytes": (szt, [P]),
    "hp_wire_pack": (INT, [P, P, P, P, szt]),
    "hp_wire_unpack": (INT, [P, szt, P, P, szt, P]),
    "hp_dev_wire_load": (INT, [P, P, szt, P]),
    "hp_dev_wire_store": (INT, [P, P, P, P, P, szt]),
    "hp_prof_begin": (INT, [P, C.c_char_p]),
    "hp_prof_end": (INT, [P, C.POINTER(szt), C.POINTER(C.c_double)]),
    "hp_prof_end_families": (INT, [P, szt, P, P, P, P]),
}

_lib = None


class EngineMissing(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libhehub_amd.so (build it first with `python -m hehub_amd.build`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineMissing(
            f"{LIB_PATH} is missing: the HIP engine has not been built "
            "(python -m hehub_amd.build). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = ABI drift; let it propagate
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
