"""Regression-based EM (Wang et al., WSDM'18) on MI355X.

Reference: ultra/learning_algorithm/regression_EM.py -- hparams :66-73, E-step :134-148, M-step
:151-155, Bernoulli pseudo labels :29-39,158, sigmoid CE :159-165, optimiser :177-202, step :233-268.
``ultr_loss_rem_fwdbwd`` does E-step, sampling, loss and gradient in one pass over the lists; the
M-step statistics ride in the all-reduced bucket and ``ultr_rem_update_propensity`` applies them.
The loss and the gradient use the PRE-update propensity (the reference fetches the assign and the
loss in one session.run without an ordering edge, q16).
"""
import torch

from .. import _lib
from ..utils import hparams as hp
from .base_algorithm import BaseAlgorithm


class RegressionEM(BaseAlgorithm):
    def __init__(self, data_set, exp_settings, forward_only=False):
        print('Build Regression-based EM algorithm.')
        self._setup(data_set, exp_settings, forward_only, hp.HParams(
            EM_step_size=0.05,
            learning_rate=0.05,
            max_gradient_norm=5.0,
            l2_loss=0.0,
            grad_strategy='ada'))
        if not forward_only:
            L = self.rank_list_size
            if L > _lib.ULTR_MAX_LIST:
                raise ValueError('selection_bias_cutoff %d > %d' % (L, _lib.ULTR_MAX_LIST))
            # non-trainable state, init 0.9 (:121-122); sigmoid_prob_b (trainable, init 0, :116) lives
            # in the spare tail of the flat parameter buffer so it shares the clip + Adagrad pass
            self.propensity = torch.full((L,), 0.9, dtype=torch.float32, device=self.device)
            self.sigmoid_prob_b = self.model.params[self.n_params:self.n_params + 1]
            self.seed = int(exp_settings.get('seed') or 0) * 2654435761 % (1 << 63) + 17
            self.step_ctr = None      # device step counter shared with the feed (hipGraph replay)
            self.uniforms = None      # tests inject [L, B] uniforms here

    def _extra_state(self):
        return {'propensity': self.propensity} if not self.forward_only else {}

    def step(self, session, input_feed, forward_only):
        if forward_only:
            return self._eval_step(input_feed)
        return self._train_step(input_feed)

    def _loss(self, tb, batch, scores):
        L = self.rank_list_size
        # parked for the backward pass's first launch (ultr_loss_rem_deferred) unless something reads its outputs earlier
        fn = _lib.lib().ultr_loss_rem_deferred if self._loss_rides() else _lib.lib().ultr_loss_rem_fwdbwd
        _lib.check(fn(
            _lib.ptr(scores), _lib.ptr(batch.labels), batch.B, L, _lib.ptr(self.propensity),
            # the Bernoulli draws are keyed by the device step counter the feed advances (fresh draws on every replay
            # of a captured step, the same ones an eager step makes); the host step count only without such a feed
            _lib.ptr(self.sigmoid_prob_b), _lib.ptr(self.uniforms), self.seed,
            0 if self.step_ctr is not None else self.global_step.value & 0xFFFFFFFF,
            _lib.ptr(self.step_ctr), self.rank * batch.B, _lib.ptr(tb['d_scores']), None,
            _lib.ptr(tb['list_weights']), _lib.ptr(tb['stats']), _lib.ptr(tb['grads'][self.n_params:]),
            _lib.ptr(tb['loss_ws']), _lib.current_stream()), 'ultr_loss_rem_fwdbwd')

    def _update(self, tb, batch):
        h, L = self.hparams, self.rank_list_size
        if h.l2_loss > 0:
            tb['local'][2:3].copy_((self.model.params ** 2).sum().reshape(1) * 0.5)
        # parked: the optimiser's update launch below applies it (one launch fewer per step)
        _lib.check(_lib.lib().ultr_rem_update_propensity_deferred(
            _lib.ptr(self.propensity), L, _lib.ptr(tb['stats']), float(h.EM_step_size), _lib.ptr(tb['local'][8:]),
            _lib.current_stream()), 'ultr_rem_update_propensity_deferred')
        # mean over the (global) batch: divide by stats[1] = B (regression_EM.py:159); all trainables incl. the bias
        self._clip_adagrad(self.model.params, self.acc, tb['grads'], self.n_total, tb['stats'][1:2], 1.0,
                           h.l2_loss if h.l2_loss > 0 else 0.0, h.max_gradient_norm if h.max_gradient_norm > 0 else 0.0,
                           h.learning_rate, tb['local'][0:1])

    def _loss_value(self, stats, local):
        loss = stats[0] / stats[1]
        if self.hparams.l2_loss > 0:
            loss = loss + self.hparams.l2_loss * local[2]
        return loss

    def _summary(self, stats, local, n_lists):
        L = self.rank_list_size
        out = {}
        for i in range(L):
            out['Examination Probability %d' % i] = local[8 + i]
        for i in range(L):  # (sum pseudo - sum clicks) / (B - sum clicks)  (:170-171)
            den = stats[1] - stats[2 + 2 * L + i]
            out['Additional pseudo clicks %d' % i] = (stats[2 + L + i] - stats[2 + 2 * L + i]) / den if den != 0 \
                else float('nan')
        if self.hparams.max_gradient_norm > 0:
            out['Gradient Norm'] = float(local[0])
        out['Learning Rate'] = self.learning_rate.value
        out['Loss'] = self._loss_value(stats, local)
        return out
