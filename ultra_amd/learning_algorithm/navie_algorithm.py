"""The naive algorithm: train on the (simulated) clicks as if they were relevance labels, no debiasing.

Reference: ultra/learning_algorithm/navie_algorithm.py (the file name is the reference's spelling) -- hparams
:35-42 (`loss_func` is 'softmax_cross_entropy' | 'sigmoid_cross_entropy' | 'pairwise_loss'), graph :43-172, step
:174-214.  The loss is base_algorithm.softmax_loss / sigmoid_loss_on_list with no propensity weights
(base_algorithm.py:163-186, 228-251), i.e. the IPW kernels with unit weights; l2 over every trainable (:137-145);
one clip + optimiser (:153-165).  Train summaries (:93-119, 166-172): Max/Min_output_score (with and without the
pad rows), [L2 Loss], Gradient Norm, Learning Rate, Loss and the plain metrics.
"""
import torch

from .. import _lib
from ..utils import hparams as hp
from .base_algorithm import BaseAlgorithm


class NavieAlgorithm(BaseAlgorithm):
    WEIGHTED_METRICS = False  # metric weights are None (navie_algorithm.py:166-172)

    def __init__(self, data_set, exp_settings, forward_only=False):
        self._setup(data_set, exp_settings, forward_only, hp.HParams(
            learning_rate=0.05,
            max_gradient_norm=5.0,
            loss_func='softmax_cross_entropy',
            l2_loss=0.0,
            grad_strategy='ada'))
        h = self.hparams
        if h.loss_func == 'pairwise_loss':
            raise NameError("name 'dtypes' is not defined (loss_func=pairwise_loss fails in the reference too)")
        self.loss_kind = _lib.LOSS_SIGMOID if h.loss_func == 'sigmoid_cross_entropy' else _lib.LOSS_SOFTMAX
        self._ones = None

    def step(self, session, input_feed, forward_only):
        if forward_only:
            return self._eval_step(input_feed)
        return self._train_step(input_feed)

    def _loss(self, tb, batch, scores):
        L = self.rank_list_size
        if self._ones is None or self._ones.shape[1] != batch.B:
            self._ones = torch.ones((L, batch.B), dtype=torch.float32, device=self.device)
        fn = _lib.lib().ultr_loss_ipw_deferred if self._loss_rides() else _lib.lib().ultr_loss_ipw_fwdbwd
        _lib.check(fn(
            _lib.ptr(scores), _lib.ptr(batch.labels), _lib.ptr(self._ones), batch.B, L, self.loss_kind,
            _lib.ptr(tb['d_scores']), _lib.ptr(tb['stats']), _lib.ptr(tb['list_weights']), _lib.ptr(tb['loss_ws']),
            _lib.current_stream()), 'ultr_loss_ipw_fwdbwd')

    def _after_scores(self, tb, batch, scores):
        # Max/Min_output_score summaries (:93-119): rank-local values, with and without the pad rows (-100000)
        local = tb['local']
        no_pad = torch.where(batch.docids == batch.n_rows, torch.full_like(scores, float(self.PADDING_SCORE)), scores)
        local[3:4].copy_(scores.max().reshape(1))
        local[4:5].copy_(scores.min().reshape(1))
        local[5:6].copy_(no_pad.max().reshape(1))
        local[6:7].copy_(no_pad.min().reshape(1))

    def _update(self, tb, batch):
        h = self.hparams
        if h.l2_loss > 0:
            tb['local'][2:3].copy_((self.model.params[:self.n_params] ** 2).sum().reshape(1) * 0.5)
        self._clip_adagrad(self.model.params, self.acc, tb['grads'], self.n_total, tb['stats'][1:2], 1.0,
                           h.l2_loss if h.l2_loss > 0 else 0.0, h.max_gradient_norm if h.max_gradient_norm > 0 else 0.0,
                           h.learning_rate, tb['local'][0:1])

    def _loss_value(self, stats, local):
        loss = stats[0] / stats[1]
        if self.hparams.l2_loss > 0:
            loss = loss + self.hparams.l2_loss * local[2]
        return loss

    def _summary(self, stats, local, n_lists):
        out = {'Max_output_score': float(local[3]), 'Min_output_score': float(local[4]),
               'Max_output_score_without_pad': float(local[5]), 'Min_output_score_without_pad': float(local[6])}
        if self.hparams.l2_loss > 0:
            out['L2 Loss'] = float(local[2])
        if self.hparams.max_gradient_norm > 0:
            out['Gradient Norm'] = float(local[0])
        out['Learning Rate'] = self.learning_rate.value
        out['Loss'] = self._loss_value(stats, local)
        return out


NaiveAlgorithm = NavieAlgorithm  # the reference exports the misspelt class name (learning_algorithm/__init__.py)
