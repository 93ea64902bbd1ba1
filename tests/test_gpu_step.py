"""GPU parity tests (-m gpu) of the whole training step behind the plugin API: ClickSimulationFeed.get_batch
-> Algorithm.step, against the op-for-op torch-CPU restatement of the reference graphs (oracle/torch_graph.py,
float64) on the SAME sampled batches and the same initial variables.  Tolerance: 1e-5 relative (fp32), with
the accumulation over steps stated where it applies."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import torch_graph as TG
from oracle import ultr_oracle as O
from tests.util import GOLDEN, SyntheticData, relerr, relerr_floor

pytestmark = pytest.mark.gpu

PBM = os.path.join(GOLDEN, 'pbm_0.1_1.0_4_1.0.json')
CASCADE = os.path.join(GOLDEN, 'cascade_0.1_1.0_4_1.0.json')
IPW_JSON = os.path.join(GOLDEN, 'randomized_pbm_0.1_1.0_4_1.0.json')


def settings(algo, algo_hparams='', model_hparams='hidden_layer_sizes=[64, 32]', Lmax=10, cutoff=10, feed_hparams=None):
    return {
        'train_input_feed': 'ultra_amd.input_layer.ClickSimulationFeed',
        'train_input_hparams': feed_hparams if feed_hparams is not None else 'click_model_json=%s' % PBM,
        'valid_input_feed': 'ultra_amd.input_layer.DirectLabelFeed', 'valid_input_hparams': '',
        'ranking_model': 'ultra_amd.ranking_model.DNN', 'ranking_model_hparams': model_hparams,
        'learning_algorithm': 'ultra_amd.learning_algorithm.' + algo, 'learning_algorithm_hparams': algo_hparams,
        'metrics': ['mrr', 'ndcg'], 'metrics_topn': [3, 5, 10], 'objective_metric': 'ndcg_10',
        'max_candidate_num': Lmax, 'selection_bias_cutoff': cutoff, 'seed': 1,
    }


def make(algo, data, exp, B):
    from ultra_amd.utils import find_class, metrics
    metrics.RankingMetricKey.MAX_LABEL = 4.0
    model = find_class(exp['learning_algorithm'])(data, exp, False)
    feed = find_class(exp['train_input_feed'])(model, B, exp['train_input_hparams'], None)
    return model, feed


def tie_uniforms(model, rng, B, L=10):
    """Clicked PAD documents tie at PADDING_SCORE; the reference breaks ties with fresh uniforms
    (metric_utils.py:76-81), so both sides get the same ones."""
    u = rng.uniform(size=(B, L)).astype(np.float32)
    model.metric_uniforms = torch.tensor(u).cuda()
    return torch.tensor(u)


def torch_feed(data, batch):
    docids = batch.docids.cpu().numpy()
    clicks = batch.labels.cpu().numpy()
    table = torch.tensor(data.features.astype(np.float64))
    d = [torch.tensor(docids[l].astype(np.int64)) for l in range(docids.shape[0])]
    c = [torch.tensor(clicks[l].astype(np.float64)) for l in range(docids.shape[0])]
    return table, d, c


def reference_for(kind, model, L, hparams, **kw):
    ps = [{k: v.astype(np.float64) for k, v in p.items()} for p in model.model.unflatten(model.model.params)]
    return TG.TorchReferenceStep(kind, ps, L, hparams, max_label=4.0, **kw)


def compare_params(model, ref, tol):
    got = model.model.unflatten(model.model.params)
    for j, p in enumerate(ref.params):
        for k in ('W', 'b', 'gamma', 'beta'):
            assert relerr_floor(got[j][k].reshape(p[k].shape), p[k].detach().numpy(), 1e-3) < tol, (j, k)


# the last case is BASELINE.json's headline shape (MSLR-10K: 136 features, DNN[512, 256, 128], 256 lists of 10): the
# compile-time specialised (FAST) instantiation of the fused step against the float64 graph directly
@pytest.mark.parametrize('hp,B,F,mhp', [('', 256, 24, None), ('loss_func=sigmoid_loss,l2_loss=0.001', 96, 24, None),
                                        ('logits_to_prob=sigmoid,max_propensity_weight=4,ranker_loss_weight=2.0', 96, 24, None),
                                        ('grad_strategy=sgd,learning_rate=0.01,max_gradient_norm=0.0', 64, 24, None),
                                        ('', 256, 136, 'hidden_layer_sizes=[512, 256, 128]')])
def test_dla_steps_match_reference_graph(hp, B, F, mhp):
    from ultra_amd.learning_algorithm.base_algorithm import DEVICE_BATCH_KEY
    data = SyntheticData(400, 10, F, seed=3, pad_frac=0.2)
    exp = settings('DLA', hp) if mhp is None else settings('DLA', hp, mhp)
    model, feed = make('DLA', data, exp, B)
    hpd = {k: getattr(model.hparams, k) for k in ('learning_rate', 'max_gradient_norm', 'loss_func', 'logits_to_prob',
                                                   'propensity_learning_rate', 'ranker_loss_weight', 'l2_loss',
                                                   'max_propensity_weight', 'grad_strategy')}
    pp = model.prop_params.cpu().numpy().astype(np.float64)
    ref = reference_for('dla', model, 10, hpd, prop_W=pp[:10, None], prop_b=pp[10:])
    rng = np.random.RandomState(1)
    for step in range(3):
        feed_dict, info = feed.get_batch(data, check_validation=True)
        batch = feed_dict[DEVICE_BATCH_KEY]
        assert (batch.labels.sum(dim=0) > 0).all()
        t_in = torch_feed(data, batch)
        u = tie_uniforms(model, rng, B)
        loss, _, summary = model.step(None, feed_dict, False)
        assert model._train_bufs['fused'] == model._fused_ok(batch)  # lists of 10: the fused step unless switched off
        ref_loss, ref_sum = ref.step(*t_in, u_pred=u)
        assert abs(float(loss) - ref_loss) < 1e-5 * abs(ref_loss), step
        got = summary.as_dict()
        for tag, val in ref_sum.items():
            key = tag.replace(' ', '_')
            assert key in got, key
            assert abs(got[key] - val) <= 2e-5 * max(abs(val), 1e-2), (step, key, got[key], val)
        assert got['Learning_Rate'] == model.hparams.learning_rate
    tol = 3e-5  # three Adagrad steps of fp32 arithmetic against float64
    compare_params(model, ref, tol)
    assert relerr(model.prop_params.cpu().numpy()[:10], ref.prop_W.detach().numpy()[:, 0]) < tol
    assert abs(model.prop_params[10].item() - ref.prop_b.item()) < tol
    assert model.global_step.eval() == 3
    assert len(info['input_list']) == B and len(info['click_list'][0]) == 10


def test_numpy_reference_style_feed_gives_the_same_step():
    """The reference's own feed format (float64 features, float32-typed doc ids, per-position arrays, q1)."""
    from ultra_amd.learning_algorithm.base_algorithm import DEVICE_BATCH_KEY
    data = SyntheticData(200, 10, 16, seed=5, pad_frac=0.3)
    exp = settings('DLA')
    m1, feed = make('DLA', data, exp, 64)
    m2, _ = make('DLA', data, exp, 64)
    feed_dict, _ = feed.get_batch(data, True)
    batch = feed_dict[DEVICE_BATCH_KEY]
    docids, clicks = batch.docids.cpu().numpy(), batch.labels.cpu().numpy()
    # compact the table like click_simulation_feed.py:98-104,140-143: only the batch's docs, pad id = len(table)
    n_rows = data.features.shape[0]
    used = np.unique(docids[docids < n_rows])
    remap = np.full(n_rows + 1, len(used), dtype=np.int64)
    remap[used] = np.arange(len(used))
    np_feed = {m2.letor_features.name: data.features[used].astype(np.float64)}
    for l in range(10):
        np_feed[m2.docid_inputs[l].name] = remap[docids[l]].astype(np.float32)
        np_feed[m2.labels[l].name] = clicks[l].astype(np.float32)
    l1, _, s1 = m1.step(None, feed_dict, False)
    l2, _, s2 = m2.step(None, np_feed, False)
    assert abs(float(l1) - float(l2)) < 1e-6 * abs(float(l1))
    assert relerr(m2.model.params.cpu().numpy(), m1.model.params.cpu().numpy()) < 1e-6
    a, b = s1.as_dict(), s2.as_dict()
    assert a.keys() == b.keys() and all(abs(a[k] - b[k]) < 1e-5 for k in a)


def test_ipw_steps_match_reference_graph():
    from ultra_amd.learning_algorithm.base_algorithm import DEVICE_BATCH_KEY
    data = SyntheticData(300, 10, 20, seed=7, pad_frac=0.1)
    exp = settings('IPWrank', 'propensity_estimator_json=%s' % IPW_JSON)
    model, feed = make('IPWrank', data, exp, 128)
    ref = reference_for('ipw', model, 10, {})
    ipw = model.propensity_estimator.IPW_list
    for step in range(3):
        feed_dict, _ = feed.get_batch(data, True)
        batch = feed_dict[DEVICE_BATCH_KEY]
        table, d, c = torch_feed(data, batch)
        clicks = batch.labels.cpu().numpy()
        pw = [torch.tensor([O.propensity_for_one_list(ipw, list(clicks[:, b]))[l] for b in range(clicks.shape[1])])
              for l in range(10)]
        u = tie_uniforms(model, np.random.RandomState(step), 128)
        loss, _, summary = model.step(None, feed_dict, False)
        ref_loss, ref_sum = ref.step(table, d, c, pw_lists=pw, u_pred=u)
        assert abs(float(loss) - ref_loss) < 1e-5 * abs(ref_loss)
        got = summary.as_dict()
        for tag, val in ref_sum.items():
            assert abs(got[tag.replace(' ', '_')] - val) <= 2e-5 * max(abs(val), 1e-2), tag
        assert 'Propensity_weights_0' in got
    compare_params(model, ref, 3e-5)


@pytest.mark.parametrize('hp,ranker,ranker_hp', [
    ('', 'DNN', 'hidden_layer_sizes=[64, 32]'),
    ('loss_func=sigmoid_cross_entropy,l2_loss=0.001', 'DNN', 'hidden_layer_sizes=[64, 32]'),
    ('', 'Linear', ''),
    ('grad_strategy=sgd,learning_rate=0.01', 'Linear', 'norm=none')])
def test_naive_steps_match_reference_graph(hp, ranker, ranker_hp):
    """NavieAlgorithm (clicks as labels, no debiasing) on the DNN and on the Linear scorer (navie_algorithm.py,
    Linear.py): same reference graph as IPW with unit propensity weights."""
    from ultra_amd.learning_algorithm.base_algorithm import DEVICE_BATCH_KEY
    data = SyntheticData(300, 10, 20, seed=11, pad_frac=0.1)
    exp = settings('NavieAlgorithm', hp, model_hparams=ranker_hp)
    exp['ranking_model'] = 'ultra_amd.ranking_model.' + ranker
    model, feed = make('NavieAlgorithm', data, exp, 128)
    hpd = dict(kv.split('=') for kv in hp.split(',')) if hp else {}
    hpd = {k: (v if k in ('loss_func', 'grad_strategy') else float(v)) for k, v in hpd.items()}
    ref = reference_for('naive', model, 10, hpd, norm='none' if 'norm=none' in ranker_hp else 'layer')
    if ranker == 'Linear':
        assert sorted(model.model.model_parameters)[-2:] == ['ranking_model/linear_W_0:0', 'ranking_model/linear_b_0:0']
    for step in range(3):
        feed_dict, _ = feed.get_batch(data, True)
        table, d, c = torch_feed(data, feed_dict[DEVICE_BATCH_KEY])
        u = tie_uniforms(model, np.random.RandomState(step), 128)
        loss, _, summary = model.step(None, feed_dict, False)
        ref_loss, ref_sum = ref.step(table, d, c, u_pred=u)
        assert abs(float(loss) - ref_loss) < 1e-5 * abs(ref_loss)
        got = summary.as_dict()
        assert set(t.replace(' ', '_') for t in ref_sum) | {'Learning_Rate'} == set(got), (sorted(ref_sum), sorted(got))
        for tag, val in ref_sum.items():
            assert abs(got[tag.replace(' ', '_')] - val) <= 2e-5 * max(abs(val), 1e-2), tag
    compare_params(model, ref, 3e-5)


def test_rem_steps_match_reference_graph():
    from ultra_amd.learning_algorithm.base_algorithm import DEVICE_BATCH_KEY
    data = SyntheticData(300, 10, 20, seed=9)
    exp = settings('RegressionEM', 'EM_step_size=0.1')
    model, feed = make('RegressionEM', data, exp, 128)
    ref = reference_for('rem', model, 10, {'EM_step_size': 0.1})
    rng = np.random.RandomState(0)
    for step in range(3):
        feed_dict, _ = feed.get_batch(data, True)
        table, d, c = torch_feed(data, feed_dict[DEVICE_BATCH_KEY])
        u = rng.uniform(size=(10, 128)).astype(np.float32)
        model.uniforms = torch.tensor(u).cuda()
        ut = tie_uniforms(model, rng, 128)
        loss, _, summary = model.step(None, feed_dict, False)
        ref_loss, ref_sum = ref.step(table, d, c, uniforms=torch.tensor(u.T.astype(np.float64)), u_pred=ut)
        assert abs(float(loss) - ref_loss) < 1e-5 * abs(ref_loss)
        got = summary.as_dict()
        assert abs(got['Gradient_Norm'] - ref_sum['Gradient Norm']) < 1e-5 * ref_sum['Gradient Norm']
        for tag in ('ndcg_10', 'Weighted_ndcg_10', 'mrr_3'):
            assert abs(got[tag] - ref_sum[tag]) < 2e-5
    compare_params(model, ref, 3e-5)
    assert relerr(model.propensity.cpu().numpy(), ref.propensity.numpy()[0]) < 1e-5
    assert abs(model.sigmoid_prob_b.item() - ref.sigmoid_prob_b.item()) < 3e-5 * max(abs(ref.sigmoid_prob_b.item()), 1e-2)
    assert abs(got['Examination_Probability_0'] - 0.9) < 0.2


def test_pd_steps_match_reference_graph():
    from ultra_amd.learning_algorithm.base_algorithm import DEVICE_BATCH_KEY
    data = SyntheticData(300, 10, 20, seed=11)
    exp = settings('PairDebias', '', feed_hparams='click_model_json=%s' % CASCADE)
    model, feed = make('PairDebias', data, exp, 128)
    ref = reference_for('pd', model, 10, {})
    for step in range(3):
        feed_dict, _ = feed.get_batch(data, True)
        batch = feed_dict[DEVICE_BATCH_KEY]
        assert (batch.labels.sum(dim=0) == 1).all()  # cascade: exactly one click per accepted list
        table, d, c = torch_feed(data, batch)
        ut = tie_uniforms(model, np.random.RandomState(step), 128)
        loss, _, summary = model.step(None, feed_dict, False)
        ref_loss, ref_sum = ref.step(table, d, c, u_pred=ut)
        assert abs(float(loss) - ref_loss) < 1e-5 * abs(ref_loss)
        got = summary.as_dict()
        assert abs(got['Gradient_Norm'] - ref_sum['Gradient Norm']) < 1e-5 * ref_sum['Gradient Norm']
        assert 'Weighted_ndcg_10' not in got and abs(got['ndcg_5'] - ref_sum['ndcg_5']) < 2e-5
    compare_params(model, ref, 3e-5)
    assert relerr(model.t_plus.cpu().numpy(), ref.t_plus.numpy()[0]) < 1e-5
    assert relerr(model.t_minus.cpu().numpy(), ref.t_minus.numpy()[0]) < 1e-5


def test_eval_step_and_direct_label_feed():
    from ultra_amd.utils import find_class, merge_TFSummary
    data = SyntheticData(150, 14, 20, seed=13, pad_frac=0.4)
    exp = settings('DLA', Lmax=14, cutoff=10)
    model, _ = make('DLA', data, exp, 64)
    vfeed = find_class(exp['valid_input_feed'])(model, 64, '', None)
    it, sums, sizes, all_scores = 0, [], [], []
    while it < len(data.initial_list):
        feed_dict, info = vfeed.get_next_batch(it, data, check_validation=False)
        _, scores, summary = model.step(None, feed_dict, True)
        sums.append(summary)
        sizes.append(len(info['input_list']))
        all_scores.append(scores)
        it += sizes[-1]
    assert sizes == [64, 64, 22]
    scores = np.concatenate(all_scores, 0)
    assert scores.shape == (150, 14)
    ps = [{k: v.astype(np.float64) for k, v in p.items()} for p in model.model.unflatten(model.model.params)]
    n_rows = data.features.shape[0]
    docids = np.where(data.initial_list < 0, n_rows, data.initial_list)
    ref, _ = O.dnn_forward(ps, O.gather_rows(data.features.astype(np.float64), docids.reshape(-1)))
    ref = ref.reshape(150, 14)
    assert relerr(scores, ref) < 1e-5
    merged = merge_TFSummary(sums, sizes).as_dict()
    labels = np.where(data.initial_list < 0, 0, data.labels).astype(np.float64)
    masked = O.remove_padding(ref, docids, n_rows)
    assert abs(merged['ndcg_10'] - O.ndcg(labels, masked, None, 10)) < 2e-5
    assert abs(merged['mrr_3'] - O.mrr(labels, masked)) < 2e-5
    assert list(merged) == ['mrr_3', 'mrr_5', 'mrr_10', 'ndcg_3', 'ndcg_5', 'ndcg_10']


@pytest.mark.parametrize('model_json', ['pbm_0.1_1.0_4_1.0.json', 'ubm_0.1_1_4_1.0.json', 'trust_bias.json'])
def test_python_rng_feed_is_bit_exact_with_the_reference_stream(model_json):
    """device_rng=False: the batch and the post-call state of Python's `random` equal what the reference's
    get_batch produces/leaves (click_simulation_feed.py:131-137; q2), for click models of every device kind."""
    from ultra_amd.learning_algorithm.base_algorithm import DEVICE_BATCH_KEY
    import json
    data = SyntheticData(120, 10, 8, seed=17, pad_frac=0.2)
    path = os.path.join(os.path.dirname(PBM), model_json)
    exp = settings('DLA', feed_hparams='click_model_json=%s,device_rng=False' % path)
    model, feed = make('DLA', data, exp, 200)
    g = json.load(open(path))
    random.seed(1234)
    feed_dict, info = feed.get_batch(data, check_validation=True)
    after = random.random()
    random.seed(1234)
    u = [random.random() for _ in range(200 * 11 * 8)]
    idxs, clicks, n_cand = O.get_batch_sequential(data.initial_list, data.labels, 10, 200, g['model_name'],
                                                  g['click_prob'], g['exam_prob'], u, True, g.get('extra'))
    assert [x[0] for x in info['rank_list_idxs']] == idxs
    assert np.array_equal(feed_dict[DEVICE_BATCH_KEY].labels.cpu().numpy().T, np.array(clicks, dtype=np.float32))
    assert after == u[n_cand * 11]  # exactly n_cand * (1 + L) draws were consumed


def test_checkpoint_roundtrip(tmp_path):
    data = SyntheticData(100, 10, 12, seed=19)
    exp = settings('RegressionEM')
    m1, feed = make('RegressionEM', data, exp, 32)
    for _ in range(2):
        m1.step(None, feed.get_batch(data, True)[0], False)
    path = m1.saver.save(None, str(tmp_path / 'RegressionEM.ckpt'), global_step=m1.global_step)
    m2, _ = make('RegressionEM', data, exp, 32)
    m2.saver.restore(None, path)
    assert torch.equal(m1.model.params, m2.model.params) and torch.equal(m1.acc, m2.acc)
    assert torch.equal(m1.propensity, m2.propensity) and m2.global_step.eval() == 2


def test_error_behaviour():
    from ultra_amd.utils import find_class
    data = SyntheticData(50, 6, 12, seed=23)
    exp = settings('DLA', Lmax=10, cutoff=10)
    model, feed = make('DLA', data, exp, 8)
    with pytest.raises(ValueError):  # lists shorter than rank_list_size (click_simulation_feed.py:124-126)
        feed.get_batch(data, True)
    with pytest.raises(NameError):   # loss_func=pairwise_loss cannot build in the reference either
        find_class(exp['learning_algorithm'])(data, settings('DLA', 'loss_func=pairwise_loss'), False)
    with pytest.raises(ImportError):
        find_class('ultra_amd.ranking_model.NoSuchModel')


def test_prefetching_feed_yields_the_same_batches_and_steps():
    """prefetch=true assembles batch t+1 inside the last launch of step t's backward pass (into the same buffers);
    the sequence of batches and the trained parameters must be exactly those of the plain feed."""
    from ultra_amd.learning_algorithm.base_algorithm import DEVICE_BATCH_KEY
    data = SyntheticData(300, 10, 20, seed=21, pad_frac=0.1)
    runs = []
    for prefetch in ('false', 'true'):
        exp = settings('DLA', feed_hparams='click_model_json=%s,seed=77,prefetch=%s' % (PBM, prefetch))
        model, feed = make('DLA', data, exp, 96)
        batches = []
        for step in range(4):
            feed_dict, info = feed.get_batch(data, True)
            b = feed_dict[DEVICE_BATCH_KEY]
            batches.append((b.docids.clone(), b.labels.clone(), list(x[0] for x in info['rank_list_idxs'])))
            model.step(None, feed_dict, False)
        # two calls with no step in between: the parked request is launched by the second call
        f1, _ = feed.get_batch(data, True)
        d1 = f1[DEVICE_BATCH_KEY].docids.clone()
        f2, _ = feed.get_batch(data, True)
        d2 = f2[DEVICE_BATCH_KEY].docids.clone()
        runs.append((batches, d1, d2, model.model.params.clone(), model.prop_params.clone()))
        del feed
    (b0, d10, d20, p0, q0), (b1, d11, d21, p1, q1) = runs
    for (da, ca, ia), (db, cb, ib) in zip(b0, b1):
        assert torch.equal(da, db) and torch.equal(ca, cb) and ia == ib
    assert not torch.equal(b0[0][0], b0[1][0])  # and they are different batches
    assert torch.equal(d10, d11) and torch.equal(d20, d21) and not torch.equal(d10, d20)
    assert torch.equal(p0, p1) and torch.equal(q0, q1)


def test_online_simulation_feed_python_stream_and_training():
    """DeterministicOnlineSimulationFeed: (a) device_rng=False reproduces the reference's get_batch from the same
    random.random() stream -- one index draw per slot, irrelevant lists dropped, lists re-ranked by the current model,
    clicks re-sampled per list until one lands (deterministic_online_simulation_feed.py:106-270) -- and leaves the
    stream where the reference would; (b) the default device stream trains a model through the usual step()."""
    from ultra_amd.learning_algorithm.base_algorithm import DEVICE_BATCH_KEY, DeviceBatch
    from ultra_amd.input_layer import DeterministicOnlineSimulationFeed
    from ultra_amd.input_layer.device_dataset import DeviceDataset
    import json
    Lmax, Ls, B = 14, 10, 96
    data = SyntheticData(150, Lmax, 12, seed=31, pad_frac=0.3)
    data.labels[::9] = 0.0                                   # some lists without any relevant document
    exp = settings('DLA', Lmax=Lmax, cutoff=Ls)
    model, _ = make('DLA', data, exp, B)
    g = json.load(open(PBM))
    feed = DeterministicOnlineSimulationFeed(model, B, 'click_model_json=%s,device_rng=False' % PBM, None)
    random.seed(99)
    feed_dict, info = feed.get_batch(data, check_validation=True)
    after = random.random()
    # the reference's loop on the same stream
    random.seed(99)
    picks = [int(random.random() * 150) for _ in range(B)]
    assert info['rank_list_idxs'] == picks
    kept = [i for i in picks if sum(0 if data.initial_list[i][x] < 0 else data.labels[i][x] for x in range(Lmax)) != 0]
    assert 0 < len(kept) < B
    n_rows = data.features.shape[0]
    docids = np.where(data.initial_list[kept] < 0, n_rows, data.initial_list[kept])
    labels = np.where(data.initial_list[kept] < 0, 0, data.labels[kept])
    dd = DeviceDataset.of(data, model.device)
    d_dev = torch.tensor(docids.T.copy(), dtype=torch.int32, device='cuda')
    scores = model.ranking_model(Lmax, DeviceBatch(dd.table, dd.stats, dd.n_rows, d_dev, None)).cpu().numpy().T
    want_d, want_c = docids.copy(), labels.astype(np.float32).copy()
    for i in range(len(kept)):
        n = int((docids[i] < n_rows).nonzero()[0].max()) + 1
        order = sorted(range(n), key=lambda k: scores[i][k], reverse=True)
        shown = [float(labels[i][order[j]]) for j in range(min(n, Ls))]
        for _ in range(101):
            clicks = O.sample_clicks_one_list(g['model_name'], g['click_prob'], g['exam_prob'], shown,
                                              [random.random() for _ in shown])
            if sum(clicks) > 0:
                break
        want_d[i, :n] = docids[i][order]
        want_c[i, :n] = 0
        want_c[i, :len(clicks)] = clicks
    batch = feed_dict[DEVICE_BATCH_KEY]
    assert batch.docids.shape == (Lmax, len(kept))
    assert np.array_equal(batch.docids.cpu().numpy().T, want_d)
    assert np.array_equal(batch.labels.cpu().numpy().T, want_c)
    assert after == random.random()                          # the stream was left exactly where the reference leaves it
    assert [list(x) for x in info['input_list']] == docids.tolist()
    assert torch.equal(feed_dict[model.docid_inputs[3].name], batch.docids[3])
    # (b) device stream: a few training steps, every fed list has a click among the shown ranks, the model moves
    feed = DeterministicOnlineSimulationFeed(model, B, 'click_model_json=%s,seed=5' % PBM, None)
    before = model.model.params.clone()
    seen = []
    for _ in range(3):
        feed_dict, info = feed.get_batch(data, check_validation=True)
        b = feed_dict[DEVICE_BATCH_KEY]
        assert (b.labels[:Ls].sum(0) >= 1).all() and (b.labels[Ls:] == 0).all()
        seen.append(b.labels.clone())
        loss, _, _ = model.step(None, feed_dict, False)
        assert np.isfinite(float(loss))
    assert not torch.equal(before, model.model.params)
    assert not torch.equal(seen[0][:, :8], seen[1][:, :8])
    nb, _ = feed.get_next_batch(140, data, check_validation=False)
    assert nb[DEVICE_BATCH_KEY].docids.shape == (Lmax, 10)
    # the stochastic feed: same contract, the ranking is a Plackett-Luce draw (tau scales the scores)
    from ultra_amd.input_layer import StochasticOnlineSimulationFeed
    sfeed = StochasticOnlineSimulationFeed(model, B, 'click_model_json=%s,tau=0.5,seed=5' % PBM, None)
    random.seed(4)
    fa, ia = sfeed.get_batch(data, check_validation=True)
    da = fa[DEVICE_BATCH_KEY].docids.clone()
    random.seed(4)
    fb, ib = sfeed.get_batch(data, check_validation=True)                   # same lists, next draw of the rankings
    db = fb[DEVICE_BATCH_KEY].docids
    assert ia['rank_list_idxs'] == ib['rank_list_idxs'] and not torch.equal(da, db)
    assert torch.equal(da.sort(dim=0).values, db.sort(dim=0).values)        # the same documents per list
    assert (fb[DEVICE_BATCH_KEY].labels[:Ls].sum(0) >= 1).all()
    loss, _, _ = model.step(None, fb, False)
    assert np.isfinite(float(loss))


@pytest.mark.parametrize('model_json', ['pbm_0.1_1.0_4_1.0.json', 'ubm_0.1_1_4_1.0.json', 'trust_bias.json'])
def test_dynamic_bias_eta_schedule(model_json):
    """dynamic_bias_eta_change / dynamic_bias_step_interval (click_simulation_feed.py:171-178): every `interval` batches
    eta grows and setExamProb rebuilds the examination table from the BUILT-IN table ** eta (q5) -- the tables on the
    device must follow, for every click-model kind, and the clicks sampled afterwards must come from the new tables."""
    import json
    from ultra_amd.utils import click_models as cm
    from ultra_amd.learning_algorithm.base_algorithm import DEVICE_BATCH_KEY
    path = os.path.join(os.path.dirname(PBM), model_json)
    data = SyntheticData(200, 10, 12, seed=41)
    exp = settings('DLA', feed_hparams='click_model_json=%s,dynamic_bias_eta_change=0.5,dynamic_bias_step_interval=2,'
                                       'device_rng=False' % path)
    model, feed = make('DLA', data, exp, 64)
    ref = cm.loadModelFromJson(json.load(open(path)))
    random.seed(5)
    for batch_no in range(1, 5):
        feed.get_batch(data, check_validation=True)
        if batch_no % 2 == 0:
            ref.eta += 0.5
            ref.setExamProb(ref.eta)
        want = ref.device_tables(10)[1]
        assert feed.click_model.eta == ref.eta
        assert feed.exam.cpu().tolist() == want, (model_json, batch_no)
    assert feed.click_model.eta == 2.0
    # a batch under the new tables, against the oracle on the same stream
    g = {'model_name': ref.model_name, 'click_prob': ref.click_prob, 'exam_prob': ref.exam_prob, 'extra': getattr(ref, 'extra', None)}
    random.seed(77)
    feed_dict, info = feed.get_batch(data, check_validation=True)
    random.seed(77)
    u = [random.random() for _ in range(64 * 11 * 12)]
    idxs, clicks, _ = O.get_batch_sequential(data.initial_list, data.labels, 10, 64, g['model_name'], g['click_prob'],
                                             g['exam_prob'], u, True, g['extra'])
    assert [x[0] for x in info['rank_list_idxs']] == idxs
    assert np.array_equal(feed_dict[DEVICE_BATCH_KEY].labels.cpu().numpy().T, np.array(clicks, dtype=np.float32))


@pytest.mark.parametrize('algo', ['DLA', 'RegressionEM', 'PairDebias'])
def test_hipgraph_replay_equals_eager_steps(algo):
    """bench.py times the whole step (prefetching feed + forward + riding loss + backward + riding metrics and assembly +
    fused tail) as ONE captured hipGraph.  Replaying that graph N times must leave exactly the parameters, optimiser state
    and loss that N eager steps leave: the Philox step counter lives on the device, so every replay draws a fresh batch."""
    data = SyntheticData(300, 10, 24, seed=13, pad_frac=0.1)
    results = []
    for use_graph in (False, True, 'split'):
        exp = settings(algo, feed_hparams='click_model_json=%s,seed=11,prefetch=true' % PBM)
        model, feed = make(algo, data, exp, 128)

        def step():
            input_feed, _ = feed.get_batch(data, check_validation=True)
            return model.step(None, input_feed, False)
        for _ in range(2):
            last = step()
        if use_graph == 'split':
            # the data-parallel launch path of bench.py: one graph up to the collective, one after it (the all-reduce
            # itself, a no-op on one rank, is issued between the two replays)
            model.enable_graph_replay(True)
            last = step()
            torch.cuda.synchronize()
            g_local, g_apply = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_local):
                input_feed, _ = feed.get_batch(data, check_validation=True)
                model.step_local(input_feed)
            with torch.cuda.graph(g_apply):
                last = model.step_apply()
            for _ in range(4):
                g_local.replay()
                model.step_reduce()
                g_apply.replay()
        elif use_graph:
            model.enable_graph_replay(True)
            last = step()
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                last = step()                       # recorded, not executed
            for _ in range(4):
                graph.replay()
        else:
            for _ in range(5):
                last = step()
        torch.cuda.synchronize()
        extra = {'DLA': lambda: model.prop_params, 'RegressionEM': lambda: model.propensity,
                 'PairDebias': lambda: model.t_plus}[algo]().clone()
        results.append((model.model.params.clone(), model.acc.clone(), extra, float(last[0])))
        del feed
    p0, a0, e0, l0 = results[0]
    for p1, a1, e1, l1 in results[1:]:
        assert torch.equal(p0, p1) and torch.equal(a0, a1) and torch.equal(e0, e1)
        assert l0 == l1 and np.isfinite(l0)


def test_rem_draws_fresh_bernoulli_labels_every_step_without_the_device_sampler():
    """device_rng=False: no kernel advances a device step counter, so regression-EM must key its Bernoulli labels
    (regression_EM.py:29-39,158: fresh tf.random_uniform per session.run) on its own step count.  Frozen variables
    (learning rate 0, EM step 0) and the same batch twice: d_scores = sigmoid(score + b) - label differ iff the draws do."""
    data = SyntheticData(80, 10, 24, seed=2)
    exp = settings('RegressionEM', 'learning_rate=0.0,EM_step_size=0.0',
                   feed_hparams='click_model_json=%s,device_rng=false' % PBM)
    m, feed = make('RegressionEM', data, exp, 32)
    assert m.step_ctr is None
    random.seed(5)
    fd, _ = feed.get_batch(data, check_validation=True)
    m.step(None, fd, False)
    d0 = m._train_bufs['d_scores'].clone()
    m.step(None, fd, False)
    d1 = m._train_bufs['d_scores'].clone()
    assert not torch.equal(d0, d1)
    # with the device sampler the feed's counter is the key, shared by every feed built on the model
    m2, feed2 = make('RegressionEM', data, settings('RegressionEM'), 32)
    assert m2.step_ctr is feed2.step_ctr


def test_a_step_result_read_after_its_slot_was_reused_raises():
    data = SyntheticData(80, 10, 24, seed=2)
    m, feed = make('DLA', data, settings('DLA'), 16)
    fd, _ = feed.get_batch(data, check_validation=True)
    first_loss, _, _ = m.step(None, fd, False)
    _, _, second_summary = m.step(None, fd, False)
    for _ in range(62):
        loss, _, _ = m.step(None, fd, False)
    assert np.isfinite(float(first_loss))          # 63 steps later: still its own row
    m.step(None, fd, False)
    m.step(None, fd, False)                        # the second step's slot has been handed out again
    with pytest.raises(RuntimeError, match='more than 64 steps'):
        second_summary.as_dict()
    assert np.isfinite(float(loss))
