"""The DNN (multi-layer perceptron) list scorer on MI355X.

Reference: ultra/ranking_model/DNN.py -- hparams :25-34, graph :63-96 (per layer: LayerNorm on
the layer INPUT, matmul, bias_add, activation on hidden layers).  Same hparam string, same
variables (dnn_W_j, dnn_b_j, layer_norm_j gamma/beta, DNN.py:70-88); the arithmetic runs in the
fp32-MFMA kernels of csrc/dnn.hip through ``ultr_dnn_forward`` / ``ultr_dnn_backward``.

Variables live in ONE flat fp32 device buffer (each tensor 64-float aligned; W_j rows padded to a
multiple of 4 columns) so clip + Adagrad and the data-parallel all-reduce are single passes.
"""
import ctypes
import math

import numpy as np
import torch

from .. import _lib
from ..utils import hparams as hp
from .base_ranking_model import BaseRankingModel

LN_EPS = 1e-3  # tf.keras.layers.LayerNormalization default epsilon


def _round_up(x, m):
    return (x + m - 1) // m * m


class DNN(BaseRankingModel):
    def __init__(self, hparams_str):
        self.hparams = hp.HParams(
            hidden_layer_sizes=[512, 256, 128],  # neurons per hidden layer
            activation_func='elu',               # elu, relu, selu, sigmoid or tanh
            initializer='None',                  # 'constant' -> 0.001
            norm='layer',                        # 'layer' | 'batch' | anything else: none
            output_size=1,
            compute_dtype='f32')                 # (new) 'bf16': opt-in fast path of the fused training step
        self.hparams.parse(hparams_str)
        self.act_func = BaseRankingModel.ACT_FUNC_DIC.get(self.hparams.activation_func)
        self.initializer = BaseRankingModel.INITIALIZER_DIC.get(self.hparams.initializer)
        self.model_parameters = {}
        self.desc = None
        self.params = None
        self._ws = {}
        self._last = None
        self._last_fused = None
        self._wt_state = None

    # ---- variables -------------------------------------------------------------------------
    def layer_shapes(self, feature_size):
        sizes = list(self.hparams.hidden_layer_sizes) + [self.hparams.output_size]
        shapes, cur = [], feature_size
        for out in sizes:
            shapes.append((cur, out))
            cur = out
        return shapes

    def create_variables(self, feature_size, device, seed=None):
        """Allocate and initialise the flat parameter buffer; idempotent (AUTO_REUSE, DNN.py:63-64)."""
        if self.params is not None:
            if feature_size != self.desc.in_features:
                raise ValueError('DNN variables were created for feature_size %d' % self.desc.in_features)
            return self.params
        if self.hparams.output_size != 1:
            raise NotImplementedError('output_size must be 1: scores are concatenated to [B, list] '
                                      '(base_algorithm.py:102)')
        if len(self.hparams.hidden_layer_sizes) + 1 > _lib.ULTR_MAX_LAYERS:
            raise ValueError('at most %d layers' % _lib.ULTR_MAX_LAYERS)
        if self.act_func is None:
            raise ValueError("activation_func '%s' unknown (elu, relu, selu, tanh, sigmoid)"
                             % self.hparams.activation_func)
        shapes = self.layer_shapes(feature_size)
        d = _lib.UltrDnnDesc()
        d.n_layers = len(shapes)
        d.in_features = feature_size
        d.activation = self.act_func
        # 'batch': keras BatchNormalization (DNN.py:70-84).  Training passes use the batch statistics (NORM_BATCH); the
        # evaluation pass (is_training=False) the moving statistics, which the reference never updates (desc_eval)
        d.norm = {'layer': _lib.NORM_LAYER, 'batch': _lib.NORM_BATCH}.get(self.hparams.norm, _lib.NORM_NONE)
        d.ln_eps = LN_EPS
        dtype = getattr(self.hparams, 'compute_dtype', 'f32')  # subclasses with their own hparams (Linear) have none
        if dtype not in ('f32', 'bf16'):
            raise ValueError("compute_dtype '%s' unknown (f32, bf16)" % dtype)
        d.compute_dtype = _lib.DTYPE_BF16 if dtype == 'bf16' else _lib.DTYPE_F32
        off = 0
        self.slices = []
        for j, (k, n) in enumerate(shapes):
            last = j == len(shapes) - 1
            ldw = 1 if last else _round_up(n, 4)  # the output layer's W is a plain [K] vector
            d.sizes[j] = n
            ent = {'K': k, 'N': n, 'ldw': ldw}
            for name, length in (('W', k * ldw), ('b', n), ('gamma', k), ('beta', k)):
                ent[name] = (off, length)
                getattr(d, 'off_' + name)[j] = off
                off += _round_up(length, 64)
            self.slices.append(ent)
        d.n_params = off
        # 64 spare floats behind the kernel-visible parameters: algorithms keep extra trainables there
        # (regression-EM's sigmoid_prob_b) so one clip + Adagrad pass covers every trainable variable
        self.n_total = off + 64
        self.desc = d
        self.desc_eval = d
        if d.norm == _lib.NORM_BATCH:
            self.desc_eval = _lib.UltrDnnDesc()
            ctypes.memmove(ctypes.byref(self.desc_eval), ctypes.byref(d), ctypes.sizeof(d))
            self.desc_eval.norm = _lib.NORM_BATCH_EVAL
        self.device = torch.device(device)
        gen = torch.Generator().manual_seed(int(seed) if seed is not None else torch.seed() % (1 << 31))
        host = np.zeros(self.n_total, dtype=np.float32)
        for j, ent in enumerate(self.slices):
            k, n, ldw = ent['K'], ent['N'], ent['ldw']
            if self.initializer is not None:
                W = np.full((k, n), self.initializer, dtype=np.float32)
                b = np.full((n,), self.initializer, dtype=np.float32)
            else:
                # tf.get_variable default = glorot_uniform; a rank-1 [n] variable has fan_in = fan_out = n
                lim_w = math.sqrt(6.0 / (k + n))
                lim_b = math.sqrt(6.0 / (n + n))
                W = ((torch.rand((k, n), generator=gen, dtype=torch.float64) * 2 - 1) * lim_w).numpy().astype(np.float32)
                b = ((torch.rand((n,), generator=gen, dtype=torch.float64) * 2 - 1) * lim_b).numpy().astype(np.float32)
            self._put(host, j, W, b, np.ones(k, np.float32), np.zeros(k, np.float32))
        self.params = torch.from_numpy(host).to(self.device)
        self._name_variables()
        return self.params

    def _put(self, host, j, W, b, gamma, beta):
        ent = self.slices[j]
        k, n, ldw = ent['K'], ent['N'], ent['ldw']
        o, ln = ent['W']
        Wp = np.zeros((k, ldw), dtype=np.float32)
        Wp[:, :n] = W
        host[o:o + ln] = Wp.reshape(-1)
        for name, arr in (('b', b), ('gamma', gamma), ('beta', beta)):
            o, ln = ent[name]
            host[o:o + ln] = arr

    def _name_variables(self):
        self.model_parameters = {}
        for j, ent in enumerate(self.slices):
            o, ln = ent['W']
            self.model_parameters['ranking_model/dnn_W_%d:0' % j] = self.params[o:o + ln].view(ent['K'], ent['ldw'])[:, :ent['N']]
            o, ln = ent['b']
            self.model_parameters['ranking_model/dnn_b_%d:0' % j] = self.params[o:o + ln]
            if self.desc.norm in (_lib.NORM_LAYER, _lib.NORM_BATCH):  # both layers are named layer_norm_%d (DNN.py:73-74)
                o, ln = ent['gamma']
                self.model_parameters['ranking_model/layer_norm_%d/gamma:0' % j] = self.params[o:o + ln]
                o, ln = ent['beta']
                self.model_parameters['ranking_model/layer_norm_%d/beta:0' % j] = self.params[o:o + ln]

    def load_numpy_params(self, layers):
        """layers: list of {'W','b','gamma','beta'} arrays in TF shapes (tests / checkpoints)."""
        host = self.params.detach().cpu().numpy().copy()
        for j, p in enumerate(layers):
            self._put(host, j, np.asarray(p['W'], np.float32), np.asarray(p['b'], np.float32),
                      np.asarray(p['gamma'], np.float32), np.asarray(p['beta'], np.float32))
        self.params.copy_(torch.from_numpy(host))

    def unflatten(self, flat):
        """Flat buffer (params or grads) -> list of {'W','b','gamma','beta'} numpy arrays in TF shapes."""
        host = flat.detach().cpu().numpy()
        out = []
        for ent in self.slices:
            o, ln = ent['W']
            item = {'W': host[o:o + ln].reshape(ent['K'], ent['ldw'])[:, :ent['N']].copy()}
            for name in ('b', 'gamma', 'beta'):
                o, ln = ent[name]
                item[name] = host[o:o + ln].copy()
            out.append(item)
        return out

    # ---- device passes -----------------------------------------------------------------------
    def _workspace(self, M):
        ws = self._ws.get(M)
        if ws is None:
            nbytes = _lib.lib().ultr_dnn_workspace_bytes(self.desc, M)
            if nbytes < 0:
                raise _lib.UltrError(_lib.lib().ultr_last_error().decode())
            ws = torch.zeros(nbytes // 4, dtype=torch.float32, device=self.device)
            self._ws[M] = ws
        return ws

    def forward_lists(self, table, table_stats, n_rows, docids_lb, out=None, is_training=True):
        """Scores [L, B] (position-major) of the docs ``table[docids_lb]``; id == n_rows is the pad row.
        ``is_training`` only matters under norm='batch' (batch statistics against the moving ones, DNN.py:82-84)."""
        L, B = docids_lb.shape
        M = L * B
        scores = out if out is not None else torch.empty((L, B), dtype=torch.float32, device=self.device)
        ws = self._workspace(M)
        _lib.check(_lib.lib().ultr_dnn_forward(
            self.desc if is_training else self.desc_eval, _lib.ptr(self.params), _lib.ptr(table), _lib.ptr(table_stats), int(n_rows), int(table.stride(0)),
            _lib.ptr(docids_lb), M, _lib.ptr(ws), _lib.ptr(scores), _lib.current_stream()), 'ultr_dnn_forward')
        self._last = (table, table_stats, int(n_rows), docids_lb, M, ws)
        return scores

    def backward(self, d_scores_lb, grads, sumsq=None):
        """grads[:n_params] = d(sum d_scores*scores)/d params for the LAST forward_lists call.  sumsq (optional,
        ``sumsq_count()`` floats) receives partial sums of squares of that gradient (``ultr_dnn_backward_sumsq``)."""
        table, table_stats, n_rows, docids_lb, M, ws = self._last
        if sumsq is not None:
            _lib.check(_lib.lib().ultr_dnn_backward_sumsq(
                self.desc, _lib.ptr(self.params), _lib.ptr(table), _lib.ptr(table_stats), n_rows, int(table.stride(0)),
                _lib.ptr(docids_lb), M, _lib.ptr(ws), _lib.ptr(d_scores_lb), _lib.ptr(grads), _lib.ptr(sumsq),
                _lib.current_stream()), 'ultr_dnn_backward_sumsq')
            return grads
        _lib.check(_lib.lib().ultr_dnn_backward(
            self.desc, _lib.ptr(self.params), _lib.ptr(table), _lib.ptr(table_stats), n_rows, int(table.stride(0)),
            _lib.ptr(docids_lb), M, _lib.ptr(ws), _lib.ptr(d_scores_lb), _lib.ptr(grads), _lib.current_stream()),
            'ultr_dnn_backward')
        return grads

    def fused_supported(self, B, L):
        """True when (B lists of L) can run as the fused training step (``ultr_dnn_train_fused``)."""
        return self._fused_workspace(B, L) is not None

    def fused_rows(self, B, L):
        """Rows per workgroup of the fused step for (B, L): 16, 32 (large batches, chain32.hpp) or 0 (not supported)."""
        return int(_lib.lib().ultr_dnn_fused_rows(self.desc, int(B), int(L)))

    def fused_tail_lists(self, B, L):
        """Lists of (B, L) that the fused step hands to a second launch of 16-row workgroups (0: one launch)."""
        return int(_lib.lib().ultr_dnn_fused_tail_lists(self.desc, int(B), int(L)))

    def _fused_workspace(self, B, L):
        key = ('fused', B, L)
        if key not in self._ws:
            nbytes = _lib.lib().ultr_dnn_fused_workspace_bytes(self.desc, int(B), int(L))
            if nbytes < 0:
                raise _lib.UltrError(_lib.lib().ultr_last_error().decode())
            self._ws[key] = torch.zeros(nbytes // 4, dtype=torch.float32, device=self.device) if nbytes > 0 else None
        return self._ws[key]

    def train_fused(self, table, table_stats, n_rows, docids_lb, scores, grads, sumsq=None):
        """Forward, the parked loss request's gradient and the backward pass in one call: scores [L, B] and
        grads[:n_params] are written (``ultr_dnn_train_fused``)."""
        L, B = docids_lb.shape
        ws = self._fused_workspace(B, L)
        # the workspace's transposed weight copies are current iff the last writer of the parameters was the update
        # call that refreshed them (torch bumps the version counter on every in-place write of its own)
        fresh = self._wt_state == (ws.data_ptr(), self.params._version)
        _lib.check(_lib.lib().ultr_dnn_train_fused(
            self.desc, _lib.ptr(self.params), _lib.ptr(table), _lib.ptr(table_stats), int(n_rows), int(table.stride(0)),
            _lib.ptr(docids_lb), int(B), int(L), _lib.ptr(ws), 1 if fresh else 0, _lib.ptr(scores), _lib.ptr(grads),
            _lib.ptr(sumsq), _lib.current_stream()), 'ultr_dnn_train_fused')
        self._last_fused = (int(B), int(L), ws)
        return scores

    def fused_update_target(self):
        """(B, L, workspace) of the last fused step: the update call refreshes the transposed weights there."""
        return self._last_fused

    def mark_wt_fresh(self, ws):
        self._wt_state = (ws.data_ptr(), self.params._version)

    def sumsq_count(self):
        return int(_lib.lib().ultr_dnn_sumsq_count(self.desc))

    def build(self, input_list, noisy_params=None, noise_rate=0.05, is_training=False, **kwargs):
        """Reference API (DNN.py:48-96): list of [B, F] feature tensors -> list of [B, 1] scores."""
        if noisy_params is not None:
            raise NotImplementedError('parameter noise is only used by the online algorithms (out of scope)')
        L, (B, F) = len(input_list), input_list[0].shape
        self.create_variables(F, input_list[0].device)
        ldx = _round_up(F, 4)
        table = torch.zeros((L * B, ldx), dtype=torch.float32, device=self.device)
        table[:, :F] = torch.cat([x.to(torch.float32) for x in input_list], dim=0)
        stats = feature_row_stats(table, F)
        docids = torch.arange(L * B, dtype=torch.int32, device=self.device).view(L, B)
        scores = self.forward_lists(table, stats, L * B, docids, is_training=bool(is_training))
        return [scores[l].view(B, 1) for l in range(L)]


def feature_row_stats(table, feature_size, eps=LN_EPS):
    """(mean, rstd) per table row plus the pad row's entry: the layer-0 LayerNorm statistics."""
    n_rows = table.shape[0]
    stats = torch.empty((n_rows + 1, 2), dtype=torch.float32, device=table.device)
    _lib.check(_lib.lib().ultr_ln_row_stats(_lib.ptr(table), n_rows, int(feature_size), int(table.stride(0)),
                                            float(eps), _lib.ptr(stats), _lib.current_stream()), 'ultr_ln_row_stats')
    return stats
