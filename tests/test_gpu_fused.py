"""GPU parity tests (-m gpu) of the fused training step (ultr_dnn_train_fused: forward + d_score + dgrad chain in one
launch, every weight gradient in the next) against the per-launch plan (ultr_dnn_forward / loss / ultr_dnn_backward)
on the same batches and the same initial variables.  The per-launch plan is itself pinned to the oracle by
tests/test_gpu_step.py and tests/test_gpu_kernels.py; the step tests there run through the fused path by default,
so the oracle comparison covers it as well.  Bounds: clicks / doc ids bit-exact, scores 2e-6 relative (different
summation order of exact fp32 arithmetic), parameters after 3 steps 2e-5."""
import os

import numpy as np
import pytest
import torch

from tests.util import GOLDEN, SyntheticData, relerr, relerr_floor
from tests.test_gpu_step import settings, make, CASCADE, IPW_JSON, PBM

pytestmark = pytest.mark.gpu

CASES = [
    # algo, algo hparams, model hparams, F, L (cutoff), B
    ('DLA', '', 'hidden_layer_sizes=[512, 256, 128]', 136, 10, 256),
    ('DLA', 'loss_func=sigmoid_loss,logits_to_prob=sigmoid,max_propensity_weight=4', 'hidden_layer_sizes=[64, 32]', 24, 10, 96),
    ('DLA', '', 'hidden_layer_sizes=[96],activation_func=relu', 33, 7, 50),
    ('DLA', '', 'hidden_layer_sizes=[200, 72, 40, 24],activation_func=tanh', 50, 5, 77),
    ('DLA', '', 'hidden_layer_sizes=[64, 32],norm=None,activation_func=selu', 20, 16, 40),
    ('DLA', '', 'hidden_layer_sizes=[640, 100],activation_func=sigmoid', 700, 10, 33),
    ('IPWrank', 'propensity_estimator_json=%s' % IPW_JSON, 'hidden_layer_sizes=[128, 64]', 40, 10, 128),
    ('IPWrank', 'propensity_estimator_json=%s,loss_func=sigmoid_loss' % IPW_JSON, 'hidden_layer_sizes=[64, 32]', 24, 4, 65),
    ('NavieAlgorithm', '', 'hidden_layer_sizes=[64, 32]', 24, 9, 64),
    ('RegressionEM', '', 'hidden_layer_sizes=[512, 256, 128]', 220, 10, 64),
    ('RegressionEM', '', 'hidden_layer_sizes=[48]', 17, 3, 100),
    ('PairDebias', '', 'hidden_layer_sizes=[256, 128]', 60, 10, 128),
    ('PairDebias', '', 'hidden_layer_sizes=[64, 32]', 24, 8, 31),
    # edges: one-document lists (four lists per workgroup, one row each), two-document lists, a single list, one hidden
    # layer narrower than an MFMA unit, a hidden layer close to what LDS can hold
    ('DLA', '', 'hidden_layer_sizes=[32, 16]', 12, 1, 37),
    ('IPWrank', 'propensity_estimator_json=%s' % IPW_JSON, 'hidden_layer_sizes=[32]', 12, 2, 9),
    ('DLA', '', 'hidden_layer_sizes=[64, 32]', 24, 10, 1),
    ('RegressionEM', '', 'hidden_layer_sizes=[8]', 5, 6, 20),
    ('DLA', '', 'hidden_layer_sizes=[768, 16]', 40, 10, 24),
    # more than 256 workgroups of the 16-row kernel: the 32-row kernel (chain32.hpp) -- three lists of 10 per workgroup
    # at the three data-set widths, four lists of 8, two lists of 16, a last workgroup with one list of its three
    ('DLA', '', 'hidden_layer_sizes=[512, 256, 128]', 136, 10, 700),
    ('RegressionEM', '', 'hidden_layer_sizes=[512, 256, 128]', 700, 10, 260),
    ('PairDebias', '', 'hidden_layer_sizes=[512, 256, 128]', 220, 10, 301),
    ('IPWrank', 'propensity_estimator_json=%s' % IPW_JSON, 'hidden_layer_sizes=[128, 64]', 40, 8, 520),
    ('DLA', '', 'hidden_layer_sizes=[64, 32]', 24, 16, 258),
    # a last round of 32-row workgroups that would be mostly empty: whole rounds of 256 x 32 rows, then the remaining
    # lists as a launch of 16-row workgroups (TAIL: lists in that launch)
    ('DLA', '', 'hidden_layer_sizes=[512, 256, 128]', 136, 10, 1000),
    ('RegressionEM', '', 'hidden_layer_sizes=[128, 64]', 220, 10, 769),
    ('PairDebias', '', 'hidden_layer_sizes=[256, 128]', 60, 10, 1024),
    ('IPWrank', 'propensity_estimator_json=%s' % IPW_JSON, 'hidden_layer_sizes=[128, 64]', 40, 8, 1300),
    ('DLA', '', 'hidden_layer_sizes=[64, 32]', 24, 16, 600),
]
TAIL = {1000: 232, 769: 1, 1024: 256, 1300: 276, 600: 88}


@pytest.mark.parametrize('algo,hp,mhp,F,L,B', CASES)
def test_fused_step_equals_the_per_launch_plan(algo, hp, mhp, F, L, B):
    from ultra_amd.learning_algorithm.base_algorithm import DEVICE_BATCH_KEY
    data = SyntheticData(max(300, B), L, F, seed=11, pad_frac=0.2 if L > 2 else 0.0)
    click = CASCADE if algo == 'PairDebias' else PBM
    exp = settings(algo, hp, mhp, Lmax=L, cutoff=L, feed_hparams='click_model_json=%s' % click)
    m1, feed = make(algo, data, exp, B)
    m2, _ = make(algo, data, exp, B)
    m2.fused_step = False
    assert m1.model.fused_supported(B, L)
    assert m1.model.fused_rows(B, L) == (32 if B > 256 else 16)
    assert m1.model.fused_tail_lists(B, L) == TAIL.get(B, 0)
    rng = np.random.RandomState(2)
    for step in range(3):
        feed_dict, _ = feed.get_batch(data, check_validation=True)
        batch = feed_dict[DEVICE_BATCH_KEY]
        u = torch.tensor(rng.uniform(size=(B, L)).astype(np.float32)).cuda()
        m1.metric_uniforms = m2.metric_uniforms = u
        if algo == 'RegressionEM':  # same Bernoulli draws on both sides: the step counter keys them
            m2.step_ctr = m1.step_ctr
        l1, _, s1 = m1.step(None, feed_dict, False)
        sc1 = m1._train_bufs['scores'].clone()
        g1 = m1._train_bufs['grads'].clone()
        l2, _, s2 = m2.step(None, feed_dict, False)
        sc2 = m2._train_bufs['scores'].clone()
        g2 = m2._train_bufs['grads'].clone()
        assert relerr(sc1.cpu().numpy(), sc2.cpu().numpy()) < 2e-6, step
        if step == 0:  # same parameters on both sides: the raw gradients are comparable
            n = m1.n_params
            assert relerr_floor(g1[:n].cpu().numpy(), g2[:n].cpu().numpy(), 1e-6) < 2e-5
        a, b = s1.as_dict(), s2.as_dict()
        assert a.keys() == b.keys()
        for k in a:
            assert abs(a[k] - b[k]) <= 2e-5 * max(abs(b[k]), 1e-2) or (np.isnan(a[k]) and np.isnan(b[k])) or a[k] == b[k], (step, k, a[k], b[k])
        assert abs(float(l1) - float(l2)) <= 1e-5 * abs(float(l2)) or (np.isnan(float(l1)) and np.isnan(float(l2)))
    assert relerr_floor(m1.model.params.cpu().numpy(), m2.model.params.cpu().numpy(), 1e-4) < 2e-5
    for k, v in m1._extra_state().items():
        w = m2._extra_state()[k]
        ok = torch.isfinite(w)
        assert relerr(v[ok].cpu().numpy(), w[ok].cpu().numpy()) < 2e-5, k


def test_fused_step_refuses_without_a_parked_loss():
    from ultra_amd import _lib
    data = SyntheticData(50, 10, 24, seed=1)
    exp = settings('DLA')
    m, _ = make('DLA', data, exp, 16)
    tb = m._train_buffers(16)
    docids = torch.zeros((10, 16), dtype=torch.int32, device='cuda')
    from ultra_amd.ranking_model.DNN import feature_row_stats
    table = torch.zeros((8, 24), dtype=torch.float32, device='cuda')
    stats = feature_row_stats(table, 24)
    with pytest.raises(_lib.UltrError, match='no loss request parked'):
        m.model.train_fused(table, stats, 8, docids, tb['scores'], tb['grads'])


def test_fused_step_metric_sums_come_with_the_update_or_a_flush():
    """The fused step's metric rider stops at its partial rows: the sums are formed by the next update launch, or by
    ultr_metrics_tail_flush (what a data-parallel step does before its collective)."""
    from ultra_amd import _lib
    data = SyntheticData(200, 10, 24, seed=5, pad_frac=0.1)
    exp = settings('DLA', '', 'hidden_layer_sizes=[64, 32]')
    outs = []
    for flush in (False, True):
        m, feed = make('DLA', data, exp, 64)
        m.metric_uniforms = torch.tensor(np.random.RandomState(1).uniform(size=(64, 10)).astype(np.float32)).cuda()
        feed_dict, _ = feed.get_batch(data, check_validation=True)
        m.step_local(feed_dict)
        tb = m._train_bufs
        assert tb['fused']
        met = tb['bucket'][m.n_total + m.n_stats:m.n_total + m.n_stats + m.train_plan.n_out]
        torch.cuda.synchronize()
        assert float(met.abs().sum()) == 0.0          # first step: nothing has formed the sums yet
        if flush:
            _lib.check(_lib.lib().ultr_metrics_tail_flush(_lib.current_stream()), 'flush')
            torch.cuda.synchronize()
            early = met.cpu().numpy().copy()
            assert early.max() > 0.0
        _, _, summary = m.step_apply()
        torch.cuda.synchronize()
        late = met.cpu().numpy().copy()
        if flush:
            assert np.array_equal(early, late)         # the update launch had nothing left to add
        outs.append((late, summary.as_dict()))
    assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1]


def test_fused_not_supported_shapes_fall_back():
    data = SyntheticData(50, 20, 24, seed=1)
    exp = settings('DLA', Lmax=20, cutoff=20)
    m, feed = make('DLA', data, exp, 16)
    assert not m.model.fused_supported(16, 20)  # lists longer than 16: the per-launch plan
    feed_dict, _ = feed.get_batch(data, check_validation=True)
    loss, _, _ = m.step(None, feed_dict, False)
    assert np.isfinite(float(loss))


def test_transposed_weights_follow_external_parameter_writes():
    """The fused step's dgrad GEMMs stream transposed weight copies that the update launch keeps current; a write to the
    parameters from outside (checkpoint restore, a test poking values) must make the next step rebuild them."""
    from ultra_amd.learning_algorithm.base_algorithm import DEVICE_BATCH_KEY
    data = SyntheticData(200, 10, 24, seed=4, pad_frac=0.1)
    exp = settings('DLA', '', 'hidden_layer_sizes=[64, 48, 32]')
    m1, feed = make('DLA', data, exp, 64)
    m2, _ = make('DLA', data, exp, 64)
    m2.fused_step = False
    rng = np.random.RandomState(3)
    for step in range(4):
        feed_dict, _ = feed.get_batch(data, check_validation=True)
        u = torch.tensor(rng.uniform(size=(64, 10)).astype(np.float32)).cuda()
        m1.metric_uniforms = m2.metric_uniforms = u
        if step == 2:  # both models jump to new parameters
            new = torch.randn_like(m1.model.params) * 0.05
            new[m1.n_params:] = 0
            mask = m1.model.params != 0
            m1.model.params.copy_(torch.where(mask, new, m1.model.params))
            m2.model.params.copy_(m1.model.params)
        m1.step(None, feed_dict, False)
        m2.step(None, feed_dict, False)
        assert relerr_floor(m1.model.params.cpu().numpy(), m2.model.params.cpu().numpy(), 1e-4) < 2e-5, step


def test_transposed_weights_follow_a_non_fused_update_between_fused_steps():
    """The C update calls write the parameters in place without touching torch's version counter: a step whose update
    went through ultr_clip_adagrad / 2 / 3 (here: fused_step switched off for one step) must leave the fused step's
    streamed weight copies marked stale, so that the next fused step of the same (B, L) rebuilds them."""
    data = SyntheticData(200, 10, 24, seed=6, pad_frac=0.1)
    exp = settings('DLA', '', 'hidden_layer_sizes=[64, 48, 32]')
    m1, feed = make('DLA', data, exp, 64)
    m2, _ = make('DLA', data, exp, 64)
    m2.fused_step = False
    rng = np.random.RandomState(5)
    for step in range(5):
        feed_dict, _ = feed.get_batch(data, check_validation=True)
        u = torch.tensor(rng.uniform(size=(64, 10)).astype(np.float32)).cuda()
        m1.metric_uniforms = m2.metric_uniforms = u
        m1.fused_step = step not in (1, 3)   # fused, per-launch, fused, per-launch, fused on the same workspace
        m1.step(None, feed_dict, False)
        assert m1._train_bufs['fused'] == m1.fused_step
        m2.step(None, feed_dict, False)
        assert relerr_floor(m1.model.params.cpu().numpy(), m2.model.params.cpu().numpy(), 1e-4) < 2e-5, step
