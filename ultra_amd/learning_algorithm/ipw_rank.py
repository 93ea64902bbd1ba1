"""Inverse propensity weighting (Wang et al. SIGIR'16, Joachims et al. WSDM'17) on MI355X.

Reference: ultra/learning_algorithm/ipw_rank.py -- hparams :56-66, graph :81-195, step :197-242
(host loop over lists calling getPropensityForOneList, :212-223).  Here the per-list weight lookup
is the ``ultr_ipw_weights`` kernel and the weighted loss + its gradient ``ultr_loss_ipw_fwdbwd``.
"""
import torch

from .. import _lib
from ..utils import find_class, hparams as hp
from .base_algorithm import BaseAlgorithm


class IPWrank(BaseAlgorithm):
    def __init__(self, data_set, exp_settings, forward_only=False):
        self._setup(data_set, exp_settings, forward_only, hp.HParams(
            propensity_estimator_type='ultra.utils.propensity_estimator.RandomizedPropensityEstimator',
            propensity_estimator_json='./example/PropensityEstimator/randomized_pbm_0.1_1.0_4_1.0.json',
            learning_rate=0.05,
            max_gradient_norm=5.0,
            loss_func='softmax_loss',
            l2_loss=0.0,
            grad_strategy='ada'))
        h = self.hparams
        name = h.propensity_estimator_type
        if name.startswith('ultra.'):
            name = 'ultra_amd.' + name[len('ultra.'):]
        self.propensity_estimator = find_class(name)(h.propensity_estimator_json)
        if h.loss_func == 'pairwise_loss':
            raise NameError("name 'dtypes' is not defined (loss_func=pairwise_loss fails in the reference too)")
        self.loss_kind = _lib.LOSS_SIGMOID if h.loss_func == 'sigmoid_loss' else _lib.LOSS_SOFTMAX
        if not forward_only:
            L = self.rank_list_size
            self.ipw_table = torch.tensor(self.propensity_estimator.ipw_table(L), dtype=torch.float32,
                                          device=self.device)
            self._ipw_host = self.ipw_table.cpu().numpy()
            self._pw = None

    def step(self, session, input_feed, forward_only):
        if forward_only:
            return self._eval_step(input_feed)
        return self._train_step(input_feed)

    def _loss(self, tb, batch, scores):
        L = self.rank_list_size
        if self._pw is None or self._pw.shape[1] != batch.B:
            self._pw = torch.empty((L, batch.B), dtype=torch.float32, device=self.device)
        # propensity_weights placeholders filled per list (ipw_rank.py:214-223)
        if self._loss_rides():  # weights and loss in one parked request, carried by the backward pass's first launch
            _lib.check(_lib.lib().ultr_loss_ipw_table_deferred(
                _lib.ptr(scores), _lib.ptr(batch.labels), _lib.ptr(self.ipw_table), L, _lib.ptr(self._pw), batch.B, L,
                self.loss_kind, _lib.ptr(tb['d_scores']), _lib.ptr(tb['stats']), _lib.ptr(tb['list_weights']),
                _lib.ptr(tb['loss_ws']), _lib.current_stream()), 'ultr_loss_ipw_table_deferred')
            return
        _lib.check(_lib.lib().ultr_ipw_weights(_lib.ptr(batch.labels), batch.B, L, _lib.ptr(self.ipw_table), L,
                                               _lib.ptr(self._pw), _lib.current_stream()), 'ultr_ipw_weights')
        _lib.check(_lib.lib().ultr_loss_ipw_fwdbwd(
            _lib.ptr(scores), _lib.ptr(batch.labels), _lib.ptr(self._pw), batch.B, L, self.loss_kind,
            _lib.ptr(tb['d_scores']), _lib.ptr(tb['stats']), _lib.ptr(tb['list_weights']), _lib.ptr(tb['loss_ws']),
            _lib.current_stream()), 'ultr_loss_ipw_fwdbwd')

    def _update(self, tb, batch):
        h = self.hparams
        if h.l2_loss > 0:
            tb['local'][2:3].copy_((self.model.params[:self.n_params] ** 2).sum().reshape(1) * 0.5)
        # l2 over every trainable (ipw_rank.py:145-148); single clip + optimiser (:155-165)
        self._clip_adagrad(self.model.params, self.acc, tb['grads'], self.n_total, tb['stats'][1:2], 1.0,
                           h.l2_loss if h.l2_loss > 0 else 0.0, h.max_gradient_norm if h.max_gradient_norm > 0 else 0.0,
                           h.learning_rate, tb['local'][0:1])

    def _loss_value(self, stats, local):
        loss = stats[0] / stats[1]
        if self.hparams.l2_loss > 0:
            loss = loss + self.hparams.l2_loss * local[2]
        return loss

    def _summary(self, stats, local, n_lists):
        out = {}
        for i in range(self.rank_list_size):
            # reduce_max over the batch (ipw_rank.py:120-125).  A maximum is not additive, but the weights are
            # (propensity_estimator.py:33-42) IPW[l] where clicked and 0 elsewhere: the summed per-rank maxima are
            # positive exactly when some rank saw a click at l, and then the global maximum is the table entry.
            if self.world > 1:
                out['Propensity weights %d' % i] = self._ipw_host[i] if stats[2 + i] > 0 else 0.0
            else:
                out['Propensity weights %d' % i] = stats[2 + i]
        out['Learning Rate'] = self.learning_rate.value
        out['Loss'] = self._loss_value(stats, local)
        return out
