"""ctypes binding of libflowmol_hip.so (C ABI: include/flowmol_hip.h).

The product path loads ONLY the HIP library built by ``flowmol_amd.build`` and raises if it is
missing -- there is no CPU or PyTorch fallback.  (``load(path)`` with an explicit path exists so the
test-suite can point the same host code at the host-emulation build of the same sources,
tests/emu/libflowmol_emu.so; nothing in the package ever does that on its own.)
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

FM_ABI_VERSION = 7
FM_DFM_CAMPBELL, FM_DFM_GAT = 0, 1
FM_NOISE_TENSORS, FM_NOISE_PHILOX = 0, 1
FM_PREC_F32, FM_PREC_BF16X3, FM_PREC_BF16X6, FM_PREC_F16X3 = 0, 1, 2, 3
FM_MAX_CONVS = 16

LIB_NAME = 'libflowmol_hip.so'
PKG_DIR = Path(__file__).resolve().parent


class fm_config(C.Structure):
    _fields_ = [
        ('abi_version', C.c_int32), ('n_atom_types', C.c_int32), ('n_charges', C.c_int32), ('n_bond_types', C.c_int32),
        ('n_vec_channels', C.c_int32), ('n_hidden_scalars', C.c_int32), ('n_hidden_edge_feats', C.c_int32),
        ('rbf_dim', C.c_int32), ('n_convs', C.c_int32), ('n_updaters', C.c_int32),
        ('update_after', C.c_int32 * FM_MAX_CONVS), ('self_conditioning', C.c_int32),
        ('time_embedding_dim', C.c_int32), ('a_token_dim', C.c_int32), ('c_token_dim', C.c_int32),
        ('e_token_dim', C.c_int32), ('rbf_dmax', C.c_float), ('msg_z', C.c_float),
        ('s_dst_feats', C.c_int32), ('v_dst_feats', C.c_int32), ('has_mask', C.c_int32), ('precision', C.c_int32),
        ('n_recycles', C.c_int32), ('edge_update_no_distance', C.c_int32),
        # ABI 5: launch-tuning overrides, 0 = automatic (see include/flowmol_hip.h)
        ('tile_edge', C.c_int32), ('tile_node', C.c_int32), ('tile_edge_update', C.c_int32), ('xcd_swizzle', C.c_int32),
        ('fuse_node', C.c_int32), ('pair_mlps', C.c_int32), ('mlp_small_tiles', C.c_int32), ('pair_slab', C.c_int32),
        # ABI 7: 0 / 1 = canonical arithmetic (a molecule's bits do not depend on its batch), -1 = the pair slab follows the batch size (results then equal to f32 summation order)
        ('canonical', C.c_int32),
    ]


TUNING_FIELDS = ('tile_edge', 'tile_node', 'tile_edge_update', 'xcd_swizzle', 'fuse_node', 'pair_mlps', 'mlp_small_tiles', 'pair_slab', 'canonical')


class fm_tensor_desc(C.Structure):
    _fields_ = [('name', C.c_char_p), ('offset', C.c_int64), ('ndim', C.c_int32), ('shape', C.c_int64 * 2)]


class fm_dst(C.Structure):
    _fields_ = [('x', C.c_void_p), ('a', C.c_void_p), ('c', C.c_void_p), ('e', C.c_void_p)]


class fm_state(C.Structure):
    _fields_ = [('x_t', C.c_void_p), ('a_t', C.c_void_p), ('c_t', C.c_void_p), ('e_t', C.c_void_p)]


class fm_dense_state(C.Structure):
    _fields_ = [('x_t', C.c_void_p), ('a_t', C.c_void_p), ('c_t', C.c_void_p), ('e_t', C.c_void_p)]


class fm_endpoint_scalars(C.Structure):
    _fields_ = [('dt', C.c_float), ('coef', C.c_float * 4), ('scale', C.c_float)]


class fm_step_noise(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('q_a', 'u1_a', 'u2_a', 'q_c', 'u1_c', 'u2_c', 'q_e', 'u1_e', 'u2_e')]


class fm_step_scalars(C.Structure):
    _fields_ = [('t', C.c_float), ('dt', C.c_float), ('x_coef', C.c_float), ('unmask_prob', C.c_float * 3),
                ('mask_prob', C.c_float * 3), ('hc_thresh', C.c_float), ('cat_temperature', C.c_float),
                ('last_step', C.c_int32), ('x_scale', C.c_float), ('dfm_type', C.c_int32), ('gat_cf', C.c_float * 3),
                ('gat_cb', C.c_float * 3), ('gat_fw', C.c_float), ('gat_bw', C.c_float),
                ('noise_mode', C.c_int32), ('step_index', C.c_int32), ('philox_seed_lo', C.c_uint32), ('philox_seed_hi', C.c_uint32)]


class fm_sampled(C.Structure):
    _fields_ = [('a1', C.c_void_p), ('c1', C.c_void_p), ('e1', C.c_void_p)]


class fm_traj_sink(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('x', 'a', 'c', 'e', 'x1', 'a1', 'c1', 'e1')]


class FlowMolHipError(RuntimeError):
    pass


_EXPORTS = {
    # name: (restype, argtypes)
    'fm_last_error': (C.c_char_p, [C.c_void_p]),
    'fm_abi_version': (C.c_int, []),
    'fm_create': (C.c_int, [C.POINTER(fm_config), C.POINTER(fm_tensor_desc), C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    'fm_destroy': (C.c_int, [C.c_void_p]),
    'fm_workspace_bytes': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]),
    'fm_batch_bind': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    'fm_remove_com': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    'fm_set_molecule_ids': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    'fm_prior_philox': (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    'fm_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(fm_state), C.c_void_p, C.POINTER(fm_dst), C.c_int, C.c_int,
                             C.POINTER(fm_dst)]),
    'fm_forward_dense': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(fm_dense_state), C.c_void_p, C.c_int, C.POINTER(fm_dst)]),
    'fm_endpoint_step': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(fm_dense_state), C.POINTER(fm_dst), C.POINTER(fm_endpoint_scalars)]),
    'fm_ctmc_step': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(fm_state), C.POINTER(fm_dst), C.POINTER(fm_step_noise),
                               C.POINTER(fm_step_scalars), C.POINTER(fm_sampled)]),
    'fm_integrate': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(fm_state), C.c_int, C.POINTER(fm_step_scalars), C.c_void_p,
                               C.POINTER(fm_step_noise), C.POINTER(fm_dst), C.POINTER(fm_dst), C.POINTER(fm_dst), C.POINTER(fm_traj_sink),
                               C.POINTER(C.c_int)]),
    'fm_set_tap': (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p]),
    'fm_clear_taps': (C.c_int, [C.c_void_p]),
    'fm_batch_query': (C.c_int, [C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p]),
    'fm_stability': (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(fm_state), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'fm_profile_enable': (C.c_int, [C.c_void_p, C.c_int]),
    'fm_profile_get': (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
}

EXPORTED_SYMBOLS = tuple(_EXPORTS)


def default_lib_path() -> Path:
    return PKG_DIR / LIB_NAME


def load(path=None) -> C.CDLL:
    """Open the shared library and type its entry points.  Raises FlowMolHipError if it is missing."""
    p = Path(path) if path is not None else default_lib_path()
    if not p.exists():
        raise FlowMolHipError(
            f"{p} not found: the HIP extension has not been built.  Run `python -m flowmol_amd.build` "
            f"(needs hipcc; cross-compiles for gfx950 without a GPU).  There is no CPU fallback.")
    if path is None:
        # the in-tree library must be the build of the in-tree sources: a stale .so (sources edited, library not rebuilt) would
        # silently test and benchmark old kernels.  The build stamp is the digest of csrc/ + the header + the flags.
        from . import build as _build
        if (_build.SRC / 'fm_engine.cpp').exists():          # a source tree (an installed copy without csrc/ has nothing to be stale against)
            try:
                digest = _build._digest()
            except OSError as e:      # e.g. csrc/ present but include/flowmol_hip.h missing: cannot tell -- say so instead of a raw FileNotFoundError
                raise FlowMolHipError(f"cannot verify that {p} matches the sources ({e}); restore include/flowmol_hip.h or rebuild with `python -m flowmol_amd.build`") from e
            if not _build.STAMP.exists():
                import warnings
                warnings.warn(f"{p} has no build stamp ({_build.STAMP.name}): it cannot be checked against flowmol_amd/csrc; rebuild with `python -m flowmol_amd.build`", RuntimeWarning)
            elif _build.STAMP.read_text().strip() != digest:
                raise FlowMolHipError(f"{p} is stale: flowmol_amd/csrc or include/flowmol_hip.h changed since it was built.  "
                                      f"Run `python -m flowmol_amd.build`.")
    lib = C.CDLL(str(p))
    for name, (res, args) in _EXPORTS.items():
        fn = getattr(lib, name)        # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.fm_abi_version() != FM_ABI_VERSION:
        raise FlowMolHipError(f"{p}: ABI version {lib.fm_abi_version()} != {FM_ABI_VERSION}")
    return lib


_lib_cache = {}


def get_lib(path=None) -> C.CDLL:
    key = str(path) if path is not None else None
    if key not in _lib_cache:
        _lib_cache[key] = load(path)
    return _lib_cache[key]
