"""Pairwise debiasing / Unbiased LambdaMART (Hu et al., WWW'19) on MI355X.

Reference: ultra/learning_algorithm/pairwise_debias.py -- hparams :62-71, pair loop :139-159,
t+/t- ratio updates :162-171, optimiser :174-199, step :225-260; pair cross entropy
base_algorithm.py:136-161 (= softplus(o_j - o_i)).  ``ultr_loss_pd_fwdbwd`` evaluates all L^2-L
ordered pairs per list from an LDS tile; the L x L pair-loss matrix rides in the all-reduced bucket
and ``ultr_pd_update_ratios`` turns it into the new ratios (pre-update ratios feed loss and
gradient, q16).  The reference returns the loss as a [1,1] array; here it is a scalar.
"""
import torch

from .. import _lib
from ..utils import hparams as hp
from .base_algorithm import BaseAlgorithm

PD_MAX_LIST = 32


class PairDebias(BaseAlgorithm):
    WEIGHTED_METRICS = False  # pairwise_debias.py:213-219 logs plain metrics only

    def __init__(self, data_set, exp_settings, forward_only=False):
        print('Build Pairwise Debiasing algorithm.')
        self._L_hint = exp_settings.get('selection_bias_cutoff', 10)
        self._setup(data_set, exp_settings, forward_only, hp.HParams(
            EM_step_size=0.05,
            learning_rate=0.005,
            max_gradient_norm=5.0,
            regulation_p=1,
            l2_loss=0.0,
            grad_strategy='ada'))
        if not forward_only:
            L = self.rank_list_size
            if L > PD_MAX_LIST:
                raise ValueError('selection_bias_cutoff %d > %d for PairDebias' % (L, PD_MAX_LIST))
            self.t_plus = torch.ones(L, dtype=torch.float32, device=self.device)   # :116-119
            self.t_minus = torch.ones(L, dtype=torch.float32, device=self.device)

    def _n_stats(self):
        L = min(int(self._L_hint), PD_MAX_LIST)
        return (L * L + 1 + 3) // 4 * 4 + 64

    def _extra_state(self):
        return {'t_plus': self.t_plus, 't_minus': self.t_minus} if not self.forward_only else {}

    def step(self, session, input_feed, forward_only):
        if forward_only:
            return self._eval_step(input_feed)
        return self._train_step(input_feed)

    def _loss(self, tb, batch, scores):
        fn = _lib.lib().ultr_loss_pd_deferred if self._loss_rides() else _lib.lib().ultr_loss_pd_fwdbwd
        _lib.check(fn(
            _lib.ptr(scores), _lib.ptr(batch.labels), batch.B, self.rank_list_size, _lib.ptr(self.t_plus),
            _lib.ptr(self.t_minus), _lib.ptr(tb['d_scores']), _lib.ptr(tb['stats']), _lib.ptr(tb['loss_ws']),
            _lib.current_stream()), 'ultr_loss_pd_fwdbwd')

    def _update(self, tb, batch):
        h, L = self.hparams, self.rank_list_size
        if h.l2_loss > 0:
            tb['local'][2:3].copy_((self.model.params[:self.n_params] ** 2).sum().reshape(1) * 0.5)
        # parked: the optimiser's update launch below applies it (one launch fewer per step)
        _lib.check(_lib.lib().ultr_pd_update_ratios_deferred(
            _lib.ptr(self.t_plus), _lib.ptr(self.t_minus), L, _lib.ptr(tb['stats']), float(h.EM_step_size),
            float(h.regulation_p), _lib.ptr(tb['local'][8:]), _lib.current_stream()), 'ultr_pd_update_ratios_deferred')
        # the loss is a batch SUM (:151-159): no normaliser
        self._clip_adagrad(self.model.params, self.acc, tb['grads'], self.n_total, None, 1.0,
                           h.l2_loss if h.l2_loss > 0 else 0.0, h.max_gradient_norm if h.max_gradient_norm > 0 else 0.0,
                           h.learning_rate, tb['local'][0:1])

    def _loss_value(self, stats, local):
        L = self.rank_list_size
        loss = stats[L * L]
        if self.hparams.l2_loss > 0:
            loss = loss + self.hparams.l2_loss * local[2]
        return loss

    def _summary(self, stats, local, n_lists):
        L = self.rank_list_size
        out = {}
        for i in range(L):
            out['t_plus Probability %d' % i] = local[8 + i]
            out['t_minus Probability %d' % i] = local[8 + L + i]
        if self.hparams.max_gradient_norm > 0:
            out['Gradient Norm'] = float(local[0])
        out['Learning Rate'] = self.learning_rate.value
        out['Loss'] = self._loss_value(stats, local)
        return out
