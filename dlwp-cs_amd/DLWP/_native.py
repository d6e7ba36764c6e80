"""
ctypes binding of libdlwpcs.so (the C ABI declared in include/dlwpcs.h).

The product path has NO CPU fallback: if the HIP library is missing or a tensor is not on a HIP device, the ops raise.
torch is used for device memory, streams and autograd glue only; no torch type crosses the ABI (raw device pointers,
sizes and a hipStream_t handle do).
"""
import ctypes
import os
import threading

import numpy as np
import torch   # must be imported before the library so that one HIP runtime (torch's) serves both

_HERE = os.path.dirname(os.path.abspath(__file__))
# development only: DLWPCS_LIB_TAG=<tag> loads lib/libdlwpcs_<tag>.so (an instrumented build made by build.py with the same
# variable set, e.g. the -DDLWPCS_TIMELINE s_memtime build of tools/timeline_*.py); the product library has no tag
_TAG = os.environ.get('DLWPCS_LIB_TAG', '')
LIB_PATH = os.path.join(os.path.dirname(_HERE), 'lib', 'libdlwpcs%s.so' % ('_' + _TAG if _TAG else ''))

F32 = 0
BF16 = 1
MSE_TARGET_F32 = 0x100
MSE_OVERWRITE = 0x200
ADAM_ZERO_GRAD = 1
HEAD_DEFER_STAGE2 = 2
ACT_NONE = 0
ACT_LEAKY_CLIP = 1
CONV_ACCUMULATE_WGRAD = 1
CONV_PREPACKED = 2
CONV_REUSE_DZ = 4
CONV_DEFER_REDUCE = 8
CONV_DEFER_RING0 = 16
CONV_OUT_PADDED = 32
CONV_DGRAD_GATHER = 64
CONSTRAINT_MAX_NORM, CONSTRAINT_NON_NEG, CONSTRAINT_UNIT_NORM, CONSTRAINT_MIN_MAX_NORM = 1, 2, 3, 4
PACK_FWD, PACK_BWD, PACK_BIAS = 0, 1, 2
WGRAD_BATCH_MAX = 24

c_void_p, c_int, c_float, c_size_t = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
c_i32p = ctypes.POINTER(ctypes.c_int32)


class ConvDesc(ctypes.Structure):
    """struct dlwpcs_conv_desc (include/dlwpcs.h)"""
    _fields_ = [('B', ctypes.c_int32), ('N', ctypes.c_int32), ('C0', ctypes.c_int32), ('C1', ctypes.c_int32),
                ('Cout', ctypes.c_int32), ('ksize', ctypes.c_int32), ('halo', ctypes.c_int32),
                ('up0', ctypes.c_int32), ('flip_north_pole', ctypes.c_int32), ('act', ctypes.c_int32),
                ('alpha', ctypes.c_float), ('vmax', ctypes.c_float), ('dtype', ctypes.c_int32),
                ('flags', ctypes.c_int32), ('c0_valid', ctypes.c_int32)]


class PackItem(ctypes.Structure):
    """struct dlwpcs_pack_item (include/dlwpcs.h)"""
    _fields_ = [(n, ctypes.c_void_p) for n in ('w_eq', 'w_pol', 'w_np', 'b_eq', 'b_pol', 'b_np', 'wpk_fwd', 'wpk_bwd',
                                                'bias_pk')] + \
               [(n, ctypes.c_int32) for n in ('ksize', 'Cin', 'Cout', 'flip_north_pole', 'dtype', 'reserved')]


class ReduceItem(ctypes.Structure):
    """struct dlwpcs_reduce_item (include/dlwpcs.h)"""
    _fields_ = [(n, ctypes.c_void_p) for n in ('partial', 'bpartial', 'dw_eq', 'dw_pol', 'dw_np', 'db_eq', 'db_pol',
                                                'db_np')] + \
               [(n, ctypes.c_int32) for n in ('ksize', 'Cin', 'Cout', 'CinP', 'CoutP', 'n_eq', 'n_4', 'n_5',
                                              'flip_north_pole', 'accumulate', 'vec', 'nblocks')] + \
               [('reserved', ctypes.c_int32 * 4)]


class WgradItem(ctypes.Structure):
    """struct dlwpcs_wgrad_item (include/dlwpcs.h)"""
    _fields_ = [('d', ConvDesc)] + [(n, ctypes.c_void_p) for n in ('src0', 'src1', 'dz', 'y', 'table_dev', 'dw_eq', 'dw_pol',
                                                                    'dw_np', 'db_eq', 'db_pol', 'db_np')]


class LossTail(ctypes.Structure):
    """struct dlwpcs_loss_tail (include/dlwpcs.h)"""
    _fields_ = [('partial', ctypes.c_void_p), ('loss_out', ctypes.c_void_p), ('nblocks', ctypes.c_int32), ('inv_n', ctypes.c_float),
                ('weight', ctypes.c_float), ('overwrite', ctypes.c_int32)]


class GConvDesc(ctypes.Structure):
    """struct dlwpcs_gconv_desc (include/dlwpcs.h)"""
    _fields_ = [(n, ctypes.c_int32) for n in
                ('B', 'H', 'W', 'Cin', 'Cout', 'kh', 'kw', 'sh', 'sw', 'dh', 'dw', 'pad_t', 'pad_l', 'Ho', 'Wo',
                 'flip_north_pole', 'dtype')]


# name -> (restype, argtypes); must list EVERY symbol of include/dlwpcs.h (tests/test_abi.py checks this)
PROTOTYPES = {
    'dlwpcs_version': (c_int, []),
    'dlwpcs_last_error': (ctypes.c_char_p, []),
    'dlwpcs_halo_table': (c_int, [c_int, c_int, c_void_p]),
    'dlwpcs_halo_inverse_table': (c_int, [c_int, c_int, c_void_p]),
    'dlwpcs_dgrad_gather_plan_ints': (c_size_t, [c_int]),
    'dlwpcs_dgrad_gather_plan': (c_int, [c_int, c_void_p]),
    'dlwpcs_pad_fwd': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'dlwpcs_pad_bwd': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'dlwpcs_conv_workspace_bytes': (c_size_t, [ctypes.POINTER(ConvDesc)]),
    'dlwpcs_conv_packed_bytes': (c_size_t, [ctypes.POINTER(ConvDesc), c_int]),
    'dlwpcs_pack_batch': (c_int, [c_void_p, c_int, c_void_p]),
    'dlwpcs_conv_fwd': (c_int, [ctypes.POINTER(ConvDesc)] + [c_void_p] * 8 + [c_void_p, c_void_p, c_void_p, c_size_t,
                                                                                 c_void_p]),
    'dlwpcs_conv_fwd_pool': (c_int, [ctypes.POINTER(ConvDesc)] + [c_void_p] * 8 + [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                                                                      c_void_p]),
    'dlwpcs_conv_fwd_head': (c_int, [ctypes.POINTER(ConvDesc)] + [c_void_p] * 4 + [ctypes.POINTER(ConvDesc)] + [c_void_p] * 6 +
                             [c_size_t, ctypes.POINTER(c_int), c_void_p]),
    'dlwpcs_conv_bwd_data': (c_int, [ctypes.POINTER(ConvDesc)] + [c_void_p] * 5 + [c_void_p, c_void_p, c_void_p,
                                                                                      c_void_p, c_size_t, c_void_p]),
    'dlwpcs_conv_bwd_data_masked': (c_int, [ctypes.POINTER(ConvDesc)] + [c_void_p] * 4 + [c_void_p] * 4 + [c_float, c_float] +
                                    [c_void_p, c_void_p, c_size_t, c_void_p]),
    'dlwpcs_conv_bwd_weights': (c_int, [ctypes.POINTER(ConvDesc)] + [c_void_p] * 4 + [c_void_p] * 6 +
                                [c_void_p, c_void_p, c_size_t, c_void_p]),
    'dlwpcs_conv_wgrad_reduce_item': (c_int, [ctypes.POINTER(ConvDesc)] + [c_void_p] * 6 +
                                      [c_void_p, c_size_t, ctypes.POINTER(ReduceItem)]),
    'dlwpcs_wgrad_reduce_batch': (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    'dlwpcs_wgrad_batch_supported': (c_int, [ctypes.POINTER(ConvDesc)]),
    'dlwpcs_wgrad_batch_sizes': (c_int, [c_void_p, c_int, ctypes.POINTER(c_size_t), ctypes.POINTER(c_size_t)]),
    'dlwpcs_wgrad_batch_plan': (c_int, [c_void_p, c_int, c_void_p, c_size_t]),
    'dlwpcs_wgrad_batch': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'dlwpcs_wgrad_batch_adam': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    'dlwpcs_wgrad_batch_adam_tail': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p,
                                             c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'dlwpcs_wgrad_batch_apply': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'dlwpcs_gconv_fwd': (c_int, [ctypes.POINTER(GConvDesc)] + [c_void_p] * 9),
    'dlwpcs_gconv_bwd_data': (c_int, [ctypes.POINTER(GConvDesc)] + [c_void_p] * 6),
    'dlwpcs_gconv_bwd_weights': (c_int, [ctypes.POINTER(GConvDesc)] + [c_void_p] * 9),
    'dlwpcs_act_fwd': (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_float, c_float, c_int, c_void_p]),
    'dlwpcs_act_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_float, c_float, c_int, c_void_p]),
    'dlwpcs_avgpool2_fwd': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'dlwpcs_avgpool2_bwd': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'dlwpcs_avgpool2_bwd_add': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'dlwpcs_avgpool2_bwd_masked': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_int,
                                           c_void_p]),
    'dlwpcs_avgpool2_bwd_ring': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_int,
                                         c_void_p, c_void_p, c_int, c_int, c_void_p]),
    'dlwpcs_conv_ring_info': (c_int, [ctypes.POINTER(ConvDesc), ctypes.POINTER(c_size_t), ctypes.POINTER(c_int)]),
    'dlwpcs_upsample2_fwd': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'dlwpcs_upsample2_bwd': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'dlwpcs_concat2': (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_void_p]),
    'dlwpcs_split2': (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_void_p]),
    'dlwpcs_cf_to_cl': (c_int, [c_void_p, c_void_p, c_int, c_int, c_size_t, c_int, c_void_p]),
    'dlwpcs_cl_to_cf': (c_int, [c_void_p, c_void_p, c_int, c_int, c_size_t, c_int, c_void_p]),
    'dlwpcs_add': (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    'dlwpcs_l1l2_regularize': (c_int, [c_void_p, c_void_p, c_size_t, c_float, c_float, c_float, c_void_p, c_void_p]),
    'dlwpcs_weight_constraint': (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_float, c_float, c_void_p]),
    'dlwpcs_mse_scratch_bytes': (c_size_t, []),
    'dlwpcs_mse_fwd_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_int, c_void_p,
                                   c_void_p]),
    'dlwpcs_adam_step': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_float, c_float,
                                 c_float, c_float, c_float, c_void_p]),
    'dlwpcs_adam_step_fused': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_float, c_float,
                                       c_float, c_float, c_float, c_int, c_void_p]),
    'dlwpcs_pad_channels': (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_void_p]),
    'dlwpcs_slice_channels': (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, c_void_p]),
    'dlwpcs_state_repack': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_size_t, c_int, c_int, c_int, c_int, c_void_p]),
    'dlwpcs_head_mse_scratch_bytes': (c_size_t, []),
    'dlwpcs_head_mse_tail': (c_int, [c_void_p, c_float, c_int, c_void_p, c_void_p, c_void_p]),
    'dlwpcs_loss_tail_run': (c_int, [c_void_p, c_void_p]),
    'dlwpcs_head_mse_step': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                     c_void_p, c_int, c_void_p, c_void_p]),
    'dlwpcs_head_mse_step_masked': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p,
                                            c_void_p, c_void_p, c_int, c_void_p, c_float, c_float, c_void_p]),
    'dlwpcs_adam_step_dev': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_int,
                                     c_void_p]),
    'dlwpcs_batch_gather': (c_int, [c_void_p, ctypes.c_int64, c_int, ctypes.c_int64, c_void_p, c_int, c_void_p, c_int,
                                    c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'dlwpcs_comm_load': (c_int, [ctypes.c_char_p]),
    'dlwpcs_comm_unique_id': (c_int, [c_void_p]),
    'dlwpcs_comm_init': (c_int, [ctypes.POINTER(c_void_p), c_void_p, c_int, c_int]),
    'dlwpcs_comm_destroy': (c_int, [c_void_p]),
    'dlwpcs_allreduce_f32': (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    'dlwpcs_lds_oob_probe': (c_int, [c_void_p, c_void_p]),
    'dlwpcs_prof_enable': (c_int, [c_int]),
    'dlwpcs_prof_reset': (c_int, []),
    'dlwpcs_prof_count': (c_int, []),
    'dlwpcs_prof_get': (c_int, [c_int, ctypes.c_char_p, c_int, ctypes.POINTER(ctypes.c_double),
                                ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    'dlwpcs_prof_known_tags': (c_int, []),
    'dlwpcs_prof_known_tag': (c_int, [c_int, ctypes.c_char_p, c_int]),
}

_lib = None
_lock = threading.Lock()


class NativeError(RuntimeError):
    pass


def lib():
    """Load libdlwpcs.so (once).  Raises loudly when it has not been built -- there is no fallback path."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise NativeError(
                        'libdlwpcs.so not found at %s: build it with `python dlwp-cs_amd/build.py` '
                        '(or __graft_entry__.build()).  The DLWP-CS MI355X engine has no CPU fallback.' % LIB_PATH)
                handle = ctypes.CDLL(LIB_PATH)
                for name, (res, args) in PROTOTYPES.items():
                    fn = getattr(handle, name)
                    fn.restype = res
                    fn.argtypes = args
                _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().dlwpcs_last_error()
        msg = msg.decode() if msg else ''
        if rc == -1:
            raise ValueError('%s: %s' % (what, msg))
        if rc == -2:
            raise NotImplementedError('%s: %s' % (what, msg))
        raise NativeError('%s failed (code %d): %s' % (what, rc, msg))


def require_device(t, what):
    if not isinstance(t, torch.Tensor):
        raise TypeError('%s: expected a torch.Tensor, got %r' % (what, type(t)))
    if not t.is_cuda:
        raise NativeError('%s: tensor is on %s; the DLWP-CS MI355X engine only runs on a HIP device '
                          '(no CPU fallback).' % (what, t.device))
    if t.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError('%s: dtype %s not supported (float32 or bfloat16)' % (what, t.dtype))


def dtype_tag(t):
    """DLWPCS_* dtype tag of an activation tensor."""
    return BF16 if t.dtype == torch.bfloat16 else F32


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return 0 if t is None else t.data_ptr()


# ------------------------------------------------------------------------------------------------------------------ #
# Halo tables: host computation through the C ABI, cached per (N, p, device)
# ------------------------------------------------------------------------------------------------------------------ #

_table_cache = {}


def halo_table_host(N, p):
    M = N + 2 * p
    out = np.empty((6, M, M), dtype=np.int32)
    check(lib().dlwpcs_halo_table(int(N), int(p), out.ctypes.data), 'dlwpcs_halo_table')
    return out


def halo_inverse_table_host(N, p):
    out = np.empty((6 * N * N, 4), dtype=np.int32)
    check(lib().dlwpcs_halo_inverse_table(int(N), int(p), out.ctypes.data), 'dlwpcs_halo_inverse_table')
    return out


def dgrad_gather_plan_host(N):
    """dlwpcs_dgrad_gather_plan(N): the inverse table followed by the gather-form data-gradient plan, or None (N < 8)."""
    n = int(lib().dlwpcs_dgrad_gather_plan_ints(int(N)))
    if n == 0:
        return None
    out = np.empty((n,), dtype=np.int32)
    check(lib().dlwpcs_dgrad_gather_plan(int(N), out.ctypes.data), 'dlwpcs_dgrad_gather_plan')
    return out


_gather_ok = set()
_lds_probe = {}


def _run_lds_oob_probe(device):
    """One dlwpcs_lds_oob_probe launch on `device`: the number of non-zero words the device returned for LDS reads beyond a
    workgroup's allocation (0 on gfx950)."""
    dev = torch.device(device)
    with torch.cuda.device(dev):
        nz = torch.full((1,), -1, dtype=torch.int32, device=dev)
        check(lib().dlwpcs_lds_oob_probe(nz.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), 'dlwpcs_lds_oob_probe')
        return int(nz.item())


def _capturing():
    return torch.cuda.is_current_stream_capturing()


def lds_oob_reads_zero(device):
    """Does this device return zeros for LDS reads beyond the workgroup's allocation?  The gather-form data gradient (conv_ws.h,
    EDGE) and the batched weight gradient's slab tails (wgrad_batch.hip) mask operands by ADDRESS on that property; one probe
    launch per device at first use decides it (never inside a graph capture: the answer is read back), a failing device gets
    a warning and the padded-grid data gradient / per-layer weight gradients."""
    key = str(torch.device(device))
    hit = _lds_probe.get(key)
    if hit is None:
        if _capturing():
            raise NativeError('lds_oob_reads_zero(%s): first use inside a graph capture (run one eager step first)' % key)
        bad = _run_lds_oob_probe(device)
        hit = bad == 0
        if not hit:
            import warnings
            warnings.warn('%s: LDS reads beyond the allocation returned %d non-zero words (dlwpcs_lds_oob_probe): the gather-form '
                          'data gradient and the batched weight gradient are switched off on this device (padded-grid / '
                          'per-layer kernels serve)' % (key, bad))
        _lds_probe[key] = hit
    return hit


def dgrad_gather_ready(N, p, device):
    """True when halo_tables(N, p, device)[1] is a dlwpcs_dgrad_gather_plan buffer AND the device passed the LDS out-of-range
    probe (CONV_DGRAD_GATHER may be set)."""
    halo_tables(N, p, device)
    return (int(N), int(p), str(device)) in _gather_ok and lds_oob_reads_zero(device)


def halo_tables(N, p, device):
    """(table, inverse_table) int32 device tensors, immutable, cached.  p == 1, N >= 8: the inverse table is the head of a
    dlwpcs_dgrad_gather_plan buffer (same pointer serves every call that takes inv_table_dev)."""
    key = (int(N), int(p), str(device))
    hit = _table_cache.get(key)
    if hit is None:
        t = torch.from_numpy(halo_table_host(N, p)).to(device)
        try:        # (a face size the plan builder refuses keeps the plain inverse table: the padded-grid data gradient serves it)
            plan = dgrad_gather_plan_host(N) if int(p) == 1 else None
        except NativeError:
            plan = None
        if plan is not None:
            inv = torch.from_numpy(plan).to(device)
            _gather_ok.add(key)
        else:
            inv = torch.from_numpy(halo_inverse_table_host(N, p)).to(device)
        hit = (t, inv)
        _table_cache[key] = hit
    return hit
