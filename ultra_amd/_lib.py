"""ctypes binding of libultr_hip.so (the C ABI declared in include/ultr_hip.h).

The library is the product: there is no CPU fallback.  ``lib()`` raises if the shared object
is missing, and every wrapper raises ``UltrError`` with the library's own message when an
entry point returns a negative code.
"""
import ctypes
import os
import weakref

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libultr_hip.so')

ULTR_MAX_LAYERS = 8
ULTR_MAX_LIST = 64
ULTR_FEED_INFO_INTS = 1088

ACT = {'elu': 0, 'relu': 1, 'selu': 2, 'tanh': 3, 'sigmoid': 4}
NORM_NONE, NORM_LAYER, NORM_BATCH, NORM_BATCH_EVAL = 0, 1, 2, 3
DTYPE_F32, DTYPE_BF16 = 0, 1
CLICK_PBM, CLICK_CASCADE, CLICK_ORACLE, CLICK_TABLE, CLICK_UBM = 0, 1, 2, 3, 4
LOSS_SOFTMAX, LOSS_SIGMOID = 0, 1
PROB_SOFTMAX, PROB_SIGMOID = 0, 1
OPT_ADAGRAD, OPT_SGD = 0, 1
METRIC_BITS = {'mrr': 1, 'ndcg': 2, 'err': 4}
METRIC_ORDER = ('mrr', 'ndcg', 'err')


class UltrError(RuntimeError):
    pass


class UltrDnnDesc(ctypes.Structure):
    _fields_ = [
        ('n_layers', ctypes.c_int),
        ('in_features', ctypes.c_int),
        ('sizes', ctypes.c_int * ULTR_MAX_LAYERS),
        ('activation', ctypes.c_int),
        ('norm', ctypes.c_int),
        ('ln_eps', ctypes.c_float),
        ('off_W', ctypes.c_longlong * ULTR_MAX_LAYERS),
        ('off_b', ctypes.c_longlong * ULTR_MAX_LAYERS),
        ('off_gamma', ctypes.c_longlong * ULTR_MAX_LAYERS),
        ('off_beta', ctypes.c_longlong * ULTR_MAX_LAYERS),
        ('n_params', ctypes.c_longlong),
        ('compute_dtype', ctypes.c_int),
    ]


_P = ctypes.c_void_p
_I = ctypes.c_int
_U = ctypes.c_uint
_LL = ctypes.c_longlong
_ULL = ctypes.c_ulonglong
_F = ctypes.c_float
_DESC = ctypes.POINTER(UltrDnnDesc)

# name -> (restype, argtypes); mirrors include/ultr_hip.h one to one
SIGNATURES = {
    'ultr_version': (_I, []),
    'ultr_last_error': (ctypes.c_char_p, []),
    'ultr_pending_create': (_P, []),
    'ultr_pending_bind': (_P, [_P]),
    'ultr_pending_count': (_I, [_P]),
    'ultr_pending_destroy': (None, [_P]),
    'ultr_profile_enable': (_I, [_I]),
    'ultr_profile_report': (_I, [ctypes.c_char_p, _I]),
    'ultr_ln_row_stats': (_I, [_P, _LL, _I, _I, _F, _P, _P]),
    'ultr_click_pbm': (_I, [_P, _LL, _I, _P, _I, _P, _I, _P, _ULL, _U, _P, _P]),
    'ultr_click_sample': (_I, [_I, _P, _LL, _I, _P, _I, _P, _I, _P, _ULL, _U, _P, _P]),
    'ultr_click_cascade': (_I, [_P, _LL, _I, _P, _I, _P, _I, _P, _ULL, _U, _P, _P]),
    'ultr_batch_assemble': (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _I, _P, _I, _I, _ULL, _U, _P, _U, _P, _I, _I,
                                 _I, _I, _I, _P, _P, _P, _P, _P]),
    'ultr_batch_assemble_deferred': (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _I, _P, _I, _ULL, _U, _P, _U, _I, _I, _I, _P, _P,
                                          _P, _P, _P]),
    'ultr_batch_assemble_flush': (_I, [_P]),
    'ultr_batch_assemble_cancel': (_I, []),
    'ultr_online_rerank_click': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _ULL, _U, _P, _U, _I, _I, _P, _P, _P,
                                      _P]),
    'ultr_online_sample_click': (_I, [_F, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _ULL, _U, _P, _U, _I, _I, _P,
                                      _P, _P, _P]),
    'ultr_dnn_workspace_bytes': (_LL, [_DESC, _LL]),
    'ultr_dnn_forward': (_I, [_DESC, _P, _P, _P, _I, _I, _P, _LL, _P, _P, _P]),
    'ultr_dnn_backward': (_I, [_DESC, _P, _P, _P, _I, _I, _P, _LL, _P, _P, _P, _P]),
    'ultr_dnn_backward_sumsq': (_I, [_DESC, _P, _P, _P, _I, _I, _P, _LL, _P, _P, _P, _P, _P]),
    'ultr_dnn_sumsq_count': (_I, [_DESC]),
    'ultr_dnn_fused_workspace_bytes': (_LL, [_DESC, _I, _I]),
    'ultr_dnn_fused_rows': (_I, [_DESC, _I, _I]),
    'ultr_dnn_fused_tail_lists': (_I, [_DESC, _I, _I]),
    'ultr_dnn_train_fused': (_I, [_DESC, _P, _P, _P, _I, _I, _P, _I, _I, _P, _I, _P, _P, _P, _P]),
    'ultr_clip_adagrad_fused': (_I, [_DESC, _I, _I, _P, _P, _P, _P, _LL, _P, _F, _F, _F, _F, _I, _P, _P, _P, _P, _P, _I, _P,
                                     _F, _F, _F, _F, _P, _P, _I, _LL, _P, _P, _I, _P]),
    'ultr_loss_workspace_floats': (_LL, [_I, _I]),
    'ultr_loss_ipw_fwdbwd': (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P]),
    'ultr_ipw_weights': (_I, [_P, _I, _I, _P, _I, _P, _P]),
    'ultr_loss_dla_fwdbwd': (_I, [_P, _P, _I, _I, _P, _I, _I, _F, _P, _P, _P, _P, _P, _P]),
    'ultr_loss_dla_deferred': (_I, [_P, _P, _I, _I, _P, _I, _I, _F, _P, _P, _P, _P, _P, _P]),
    'ultr_loss_flush': (_I, [_P]),
    'ultr_loss_cancel': (_I, []),
    'ultr_loss_ipw_deferred': (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P]),
    'ultr_loss_ipw_table_deferred': (_I, [_P, _P, _P, _I, _P, _I, _I, _I, _P, _P, _P, _P, _P]),
    'ultr_loss_pd_deferred': (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _P]),
    'ultr_loss_rem_deferred': (_I, [_P, _P, _I, _I, _P, _P, _P, _ULL, _U, _P, _U, _P, _P, _P, _P, _P, _P, _P]),
    'ultr_loss_rem_fwdbwd': (_I, [_P, _P, _I, _I, _P, _P, _P, _ULL, _U, _P, _U, _P, _P, _P, _P, _P, _P, _P]),
    'ultr_rem_update_propensity': (_I, [_P, _I, _P, _F, _P, _P]),
    'ultr_loss_pd_fwdbwd': (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _P]),
    'ultr_pd_update_ratios': (_I, [_P, _P, _I, _P, _F, _F, _P, _P]),
    'ultr_rem_update_propensity_deferred': (_I, [_P, _I, _P, _F, _P, _P]),
    'ultr_pd_update_ratios_deferred': (_I, [_P, _P, _I, _P, _F, _F, _P, _P]),
    'ultr_em_update_flush': (_I, [_P]),
    'ultr_metrics_tail_flush': (_I, [_P]),
    'ultr_metrics_workspace_floats': (_LL, [_I, _I]),
    'ultr_metrics': (_I, [_P, _P, _P, _I, _P, _P, _I, _I, ctypes.POINTER(ctypes.c_int), _I, _I, _F, _P, _P, _P]),
    'ultr_metrics_static_max_floats': (_LL, []),
    'ultr_metrics_static_prepare': (_I, []),
    'ultr_metrics_static': (_I, [_P, _P, _P, _I, _P, _P, _I, _I, ctypes.POINTER(ctypes.c_int), _I, _I, _F, _P, _P]),
    'ultr_metrics_deferred': (_I, [_P, _P, _P, _I, _P, _P, _I, _I, ctypes.POINTER(ctypes.c_int), _I, _I, _F, _P, _P, _P]),
    'ultr_metrics_ties': (_I, [_P, _P, _P, _I, _P, _I, _I, ctypes.POINTER(ctypes.c_int), _I, _I, _F,
                               ctypes.c_ulonglong, ctypes.c_uint, _P, ctypes.c_uint, _P, _P, _P]),
    'ultr_metrics_ties_deferred': (_I, [_P, _P, _P, _I, _P, _I, _I, ctypes.POINTER(ctypes.c_int), _I, _I, _F,
                                        ctypes.c_ulonglong, ctypes.c_uint, _P, ctypes.c_uint, _P, _P, _P]),
    'ultr_metrics_flush': (_I, [_P]),
    'ultr_metrics_cancel': (_I, []),
    'ultr_host_device_pointer': (_I, [_P, ctypes.POINTER(ctypes.c_void_p)]),
    'ultr_sort_by_scores': (_I, [_P, _P, _I, _I, _I, _P, _P]),
    'ultr_clip_adagrad': (_I, [_P, _P, _P, _LL, _P, _F, _F, _F, _F, _I, _P, _P, _P]),
    'ultr_clip_adagrad2': (_I, [_P, _P, _P, _LL, _P, _F, _F, _F, _F, _I, _P, _P, _P, _P, _P, _I, _P, _F, _F, _F, _F, _P, _P]),
    'ultr_clip_adagrad3': (_I, [_P, _P, _P, _LL, _P, _F, _F, _F, _F, _I, _P, _P, _P, _P, _P, _I, _P, _F, _F, _F, _F, _P,
                                _P, _I, _LL, _P, _P, _I, _P]),
}

_lib = None


def lib():
    """Load libultr_hip.so (once).  Fails loudly: the HIP library IS the implementation."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise UltrError(
                'libultr_hip.so not found at %s -- build it with `make -C ultra_amd/csrc` '
                '(or __graft_entry__.build()); ultra_amd has no CPU fallback' % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here = header and library out of sync
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what=''):
    if rc < 0:
        raise UltrError('%s failed (%d): %s' % (what, rc, lib().ultr_last_error().decode()))
    return rc


def _destroy_context(handle):
    if _lib is not None:
        _lib.ultr_pending_destroy(handle)


class PendingContext(object):
    """A caller-owned context for the library's parked (*_deferred) requests: ultr_pending_create / _bind / _destroy
    (include/ultr_hip.h, Conventions).  Every learning algorithm owns one and binds it before the calls it makes for
    itself, and so do the feeds built for it, so models stepped alternately on one host thread never see each other's
    requests.  ``PendingContext(None)`` stands for the calling thread's own default context."""

    def __init__(self, handle='new'):
        self.assemble_owner = None  # id() of the feed whose next-batch request was parked here last
        if handle == 'new':
            h = lib().ultr_pending_create()
            if not h:
                raise UltrError('ultr_pending_create: out of host memory')
            self.handle = ctypes.c_void_p(h)
            self._fin = weakref.finalize(self, _destroy_context, self.handle)
            self._fin.atexit = False  # at interpreter shutdown the feeds may still want to cancel into it
        else:
            self.handle = handle

    def bind(self):
        """Make this the calling thread's current context (it stays bound until another one is)."""
        lib().ultr_pending_bind(self.handle)

    def count(self):
        return lib().ultr_pending_count(self.handle)


_thread_default = PendingContext(None)


def context_of(model):
    """The request context of a learning algorithm (the thread's default for anything that has none)."""
    return getattr(model, 'pending_ctx', None) or _thread_default


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_raw_stream = None


def host_device_ptr(t):
    """Device-side address of a pinned host tensor (None if the memory is not mapped for the GPU)."""
    out = ctypes.c_void_p()
    if lib().ultr_host_device_pointer(ctypes.c_void_p(t.data_ptr()), ctypes.byref(out)) != 0 or not out.value:
        return None
    return out


def current_stream():
    """hipStream_t of torch's current stream on the current device (what the launches are ordered on)."""
    global _raw_stream
    import torch
    if _raw_stream is None:
        # the raw accessor skips building a torch.cuda.Stream object: 6 of these per step were a quarter of the
        # host-side cost of a training step
        _raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', False)
    if _raw_stream:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
