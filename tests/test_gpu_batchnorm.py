"""GPU tests (-m gpu) of norm='batch' (reference DNN.py:70-84, base_ranking_model.py:67-70: one keras
BatchNormalization per layer, called with training=is_training) against the float64 restatement in
oracle/torch_graph.py: training steps use the batch statistics over all B*L rows of the step (pad rows included, as
zero rows), the evaluation step the moving statistics -- which the reference's training step never updates, so they
are mean 0 / variance 1.  Parity unpinned by the reference (no fixtures for the TF arithmetic), 1e-5 against the oracle."""
import numpy as np
import pytest
import torch

from oracle import torch_graph as TG
from tests.util import SyntheticData, relerr
from tests.test_gpu_step import settings, make, torch_feed, reference_for, tie_uniforms, compare_params, IPW_JSON

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('algo,kind,mhp,F,L,B', [
    ('DLA', 'dla', 'hidden_layer_sizes=[64, 32],norm=batch', 24, 10, 96),
    ('DLA', 'dla', 'hidden_layer_sizes=[512, 256, 128],norm=batch', 136, 10, 256),
    ('DLA', 'dla', 'hidden_layer_sizes=[40],norm=batch,activation_func=tanh', 17, 7, 33),
    ('IPWrank', 'ipw', 'hidden_layer_sizes=[64, 32],norm=batch,activation_func=relu', 24, 10, 64),
    ('NavieAlgorithm', 'naive', 'hidden_layer_sizes=[48, 24],norm=batch', 20, 20, 40),   # lists longer than 16
])
def test_batch_norm_training_steps_match_the_reference_graph(algo, kind, mhp, F, L, B):
    from ultra_amd.learning_algorithm.base_algorithm import DEVICE_BATCH_KEY
    from ultra_amd.utils.propensity_estimator import RandomizedPropensityEstimator
    data = SyntheticData(300, L, F, seed=3, pad_frac=0.2)
    hp = 'propensity_estimator_json=%s' % IPW_JSON if algo == 'IPWrank' else ''
    exp = settings(algo, hp, mhp, Lmax=L, cutoff=L)
    model, feed = make(algo, data, exp, B)
    assert not model.model.fused_supported(B, L)        # batch statistics couple the rows: the per-layer plan
    keys = {'dla': ('learning_rate', 'max_gradient_norm', 'loss_func', 'logits_to_prob', 'propensity_learning_rate',
                    'ranker_loss_weight', 'l2_loss', 'max_propensity_weight', 'grad_strategy'),
            'ipw': ('learning_rate', 'max_gradient_norm', 'loss_func', 'l2_loss', 'grad_strategy'),
            'naive': ('learning_rate', 'max_gradient_norm', 'loss_func', 'l2_loss', 'grad_strategy')}[kind]
    hpd = {k: getattr(model.hparams, k) for k in keys}
    kw = {'norm': 'batch', 'activation': model.model.hparams.activation_func}
    if kind == 'dla':
        pp = model.prop_params.cpu().numpy().astype(np.float64)
        kw.update(prop_W=pp[:L, None], prop_b=pp[L:])
    ref = reference_for(kind, model, L, hpd, **kw)
    est = RandomizedPropensityEstimator(IPW_JSON) if kind == 'ipw' else None
    rng = np.random.RandomState(1)
    for step in range(3):
        fd, _ = feed.get_batch(data, check_validation=True)
        batch = fd[DEVICE_BATCH_KEY]
        t_in = torch_feed(data, batch)
        u = tie_uniforms(model, rng, B, L)
        loss, _, summary = model.step(None, fd, False)
        pw = None
        if kind == 'ipw':
            clicks = batch.labels.t().cpu().numpy()
            pwn = np.array([est.getPropensityForOneList(list(c)) for c in clicks])
            pw = [torch.tensor(pwn[:, l].astype(np.float64)) for l in range(L)]
        ref_loss, ref_sum = ref.step(*t_in, pw_lists=pw, u_pred=u)
        assert abs(float(loss) - ref_loss) < 1e-5 * abs(ref_loss), step
        got = summary.as_dict()
        for tag, val in ref_sum.items():
            key = tag.replace(' ', '_')
            if 'output_score' in key or key not in got:
                continue
            assert abs(got[key] - val) <= 2e-5 * max(abs(val), 1e-2), (step, key, got[key], val)
    compare_params(model, ref, 3e-5)


def test_batch_norm_evaluation_uses_the_initial_moving_statistics():
    from ultra_amd.learning_algorithm.base_algorithm import DEVICE_BATCH_KEY
    L, F, B = 10, 24, 50
    data = SyntheticData(200, L, F, seed=4, pad_frac=0.1)
    exp = settings('DLA', '', 'hidden_layer_sizes=[64, 32],norm=batch', Lmax=L, cutoff=L)
    model, feed = make('DLA', data, exp, B)
    fd, _ = feed.get_batch(data, check_validation=True)
    model.step(None, fd, False)                      # one training step: the moving statistics must NOT move
    batch = fd[DEVICE_BATCH_KEY]
    table, d, _ = torch_feed(data, batch)
    ps = TG.params_from_numpy([{k: v.astype(np.float64) for k, v in p.items()}
                               for p in model.model.unflatten(model.model.params)])
    with torch.no_grad():
        want_eval = TG.get_ranking_scores(ps, table, d[:L], 'elu', 'batch_eval').numpy()
        want_train = TG.get_ranking_scores(ps, table, d[:L], 'elu', 'batch').numpy()
    got_eval = model.ranking_model(L, batch, is_training=False).t().cpu().numpy()
    got_train = model.ranking_model(L, batch, is_training=True).t().cpu().numpy()
    assert relerr(got_eval, want_eval) < 1e-5
    assert relerr(got_train, want_train) < 1e-5
    assert relerr(got_eval, want_train) > 1e-3           # the two modes do differ
    _, scores, _ = model.step(None, fd, True)            # the evaluation step of the plugin API
    assert relerr(scores, want_eval) < 1e-5


def test_batch_norm_reference_build_api():
    from ultra_amd.ranking_model.DNN import DNN
    m = DNN('hidden_layer_sizes=[32, 16],norm=batch')
    g = torch.Generator().manual_seed(0)
    inputs = [torch.randn((8, 12), generator=g).cuda() for _ in range(5)]
    out_train = m.build(inputs, is_training=True)
    out_eval = m.build(inputs, is_training=False)
    ps = TG.params_from_numpy([{k: v.astype(np.float64) for k, v in p.items()} for p in m.unflatten(m.params)])
    cpu = [x.cpu().double() for x in inputs]
    with torch.no_grad():
        wt = torch.cat(TG.dnn_build(ps, cpu, 'elu', 'batch'), dim=1).numpy()
        we = torch.cat(TG.dnn_build(ps, cpu, 'elu', 'batch_eval'), dim=1).numpy()
    assert relerr(torch.cat(out_train, dim=1).cpu().numpy(), wt) < 1e-5
    assert relerr(torch.cat(out_eval, dim=1).cpu().numpy(), we) < 1e-5
    assert 'ranking_model/layer_norm_1/gamma:0' in m.model_parameters
