"""The Dual Learning Algorithm (Ai et al., SIGIR'18) on MI355X.

Reference: ultra/learning_algorithm/dla.py -- hparams :53-68, graph :84-222, two-group clip +
Adagrad :224-255, propensity tower :257-306, step :308-341, normalised weights :343-364.
Same constructor, hparam string and step() contract; the graph's arithmetic runs in
``ultr_loss_dla_fwdbwd`` (rank loss, exam loss, tower forward/backward, both weightings, in one
kernel) plus the shared DNN / optimiser / metric kernels.
"""
import math

import torch

from .. import _lib
from ..utils import hparams as hp
from .base_algorithm import BaseAlgorithm


class DLA(BaseAlgorithm):
    def __init__(self, data_set, exp_settings, forward_only=False):
        print('Build DLA')
        self._setup(data_set, exp_settings, forward_only, hp.HParams(
            learning_rate=0.05,                 # Learning rate.
            max_gradient_norm=5.0,              # Clip gradients to this norm.
            loss_func='softmax_loss',           # softmax_loss | sigmoid_loss
            logits_to_prob='softmax',           # softmax | sigmoid
            propensity_learning_rate=-1.0,      # -1: same as learning_rate
            ranker_loss_weight=1.0,
            l2_loss=0.0,
            max_propensity_weight=-1,
            constant_propensity_initialization=False,
            grad_strategy='ada'))
        h = self.hparams
        self.propensity_learning_rate = HostLR(h.learning_rate if h.propensity_learning_rate < 0
                                               else h.propensity_learning_rate)
        if h.loss_func == 'pairwise_loss':
            # base_algorithm.py:226 references an undefined name: the reference cannot build this graph
            raise NameError("name 'dtypes' is not defined (loss_func=pairwise_loss fails in the reference too)")
        self.loss_kind = _lib.LOSS_SIGMOID if h.loss_func == 'sigmoid_loss' else _lib.LOSS_SOFTMAX
        self.prob_kind = _lib.PROB_SIGMOID if h.logits_to_prob == 'sigmoid' else _lib.PROB_SOFTMAX
        if not forward_only:
            L = self.rank_list_size
            if L > _lib.ULTR_MAX_LIST:
                raise ValueError('selection_bias_cutoff %d > %d' % (L, _lib.ULTR_MAX_LIST))
            # denoising_model/propensity_network/W_0 [L,1], b_0 [1] (dla.py:278-281): Glorot unless constant
            if h.constant_propensity_initialization:
                init = torch.full((L + 1,), 0.001, dtype=torch.float32)
            else:
                gen = torch.Generator().manual_seed(int(exp_settings.get('seed') or 0) + 7919)
                lim_w, lim_b = math.sqrt(6.0 / (L + 1)), math.sqrt(6.0 / 2)
                u = torch.rand(L + 1, generator=gen, dtype=torch.float64) * 2 - 1
                init = torch.cat([u[:L] * lim_w, u[L:] * lim_b]).float()
            self.prop_params = init.to(self.device)
            self.prop_acc = torch.full((L + 1,), 0.1, dtype=torch.float32, device=self.device)

    def _n_stats(self):
        return 8 + 4 * _lib.ULTR_MAX_LIST + 72

    def _extra_state(self):
        return {'prop_params': self.prop_params, 'prop_acc': self.prop_acc} if not self.forward_only else {}

    def step(self, session, input_feed, forward_only):
        if forward_only:
            return self._eval_step(input_feed)
        return self._train_step(input_feed)

    # stats: kernel writes [0, 4+3L); un-normalised tower gradient lives right behind it
    def _pg_offset(self):
        return (4 + 3 * self.rank_list_size + 3) // 4 * 4

    def _loss(self, tb, batch, scores):
        L, h = self.rank_list_size, self.hparams
        pg = tb['stats'][self._pg_offset():]
        # The loss rides with the first launch of the backward pass (ultr_loss_dla_deferred) when nothing needs its
        # outputs before that: the in-step metrics, which read the list weights, ride with a later launch of the
        # same pass.  ULTR_LOSS_RIDE=0: a launch of its own.
        fn = _lib.lib().ultr_loss_dla_deferred if self._loss_rides() else _lib.lib().ultr_loss_dla_fwdbwd
        _lib.check(fn(
            _lib.ptr(scores), _lib.ptr(batch.labels), batch.B, L, _lib.ptr(self.prop_params), self.loss_kind,
            self.prob_kind, float(h.max_propensity_weight), _lib.ptr(tb['d_scores']), _lib.ptr(pg),
            _lib.ptr(tb['list_weights']), _lib.ptr(tb['stats']), _lib.ptr(tb['loss_ws']), _lib.current_stream()),
            'ultr_loss_dla_fwdbwd')

    def _update(self, tb, batch):
        L, h = self.rank_list_size, self.hparams
        stats, local = tb['stats'], tb['local']
        pg = stats[self._pg_offset():]
        clip = h.max_gradient_norm if h.max_gradient_norm > 0 else 0.0
        if h.l2_loss > 0:
            local[2:3].copy_((self.model.params[:self.n_params] ** 2).sum().reshape(1) * 0.5)
        # exam loss -> propensity tower (dla.py:237,241-242,247,250-251); rank loss (+ l2 on the ranker only, :230-234)
        # -> ranking model, clip norm scaled by the loss weight (:243-244).  Two independent clip + Adagrad groups.
        self._clip_adagrad2(
            (self.model.params, self.acc, tb['grads'], self.n_total, stats[1:2], 1.0,
             h.l2_loss if h.l2_loss > 0 else 0.0, clip * h.ranker_loss_weight, self.learning_rate.value, local[0:1]),
            (self.prop_params, self.prop_acc, pg, L + 1, stats[3:4], 1.0, 0.0, clip,
             self.propensity_learning_rate.value, local[1:2]))

    def _losses(self, stats, local):
        rank = stats[0] / stats[1]
        if self.hparams.l2_loss > 0:
            rank = rank + self.hparams.l2_loss * local[2]
        return rank, stats[2] / stats[3]

    def _loss_value(self, stats, local):
        rank, exam = self._losses(stats, local)
        return exam + self.hparams.ranker_loss_weight * rank  # dla.py:180

    def _summary(self, stats, local, n_lists):
        L, h = self.rank_list_size, self.hparams
        rank, exam = self._losses(stats, local)
        out = {}
        for i in range(L):
            out['Inverse Propensity weights %d' % i] = stats[4 + i] / self.world
        out['Rank Loss'] = rank
        for i in range(L):
            out['Relevance weights %d' % i] = stats[4 + L + i] / n_lists
        out['Exam Loss'] = exam
        nr, nd = float(local[0]), float(local[1])
        if h.max_gradient_norm > 0:  # norm of the CLIPPED gradients (dla.py:245)
            nr, nd = min(nr, h.max_gradient_norm * h.ranker_loss_weight), min(nd, h.max_gradient_norm)
        out['Gradient Norm'] = math.sqrt(nr * nr + nd * nd)
        out['Learning Rate'] = self.learning_rate.value
        out['Final Loss'] = exam + h.ranker_loss_weight * rank
        return out


class HostLR(object):
    def __init__(self, value):
        self.value = float(value)

    def eval(self, session=None):
        return self.value
