"""CPU suite (-m "not gpu"): pins the oracle against the reference-generated golden vectors, checks the
two independent oracle derivations against each other, the host-side plugin logic, and that the C-ABI
library exports every symbol include/ultr_hip.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import random
import re

import numpy as np
import pytest
import torch

from oracle import ultr_oracle as O
from oracle import torch_graph as TG
from tests.util import GOLDEN, load_json, SyntheticData, relerr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- click models: oracle vs goldens generated from the reference module -------------------------
def test_click_oracle_matches_reference_goldens():
    g = load_json('click_goldens.json')
    n = 0
    for model in g['models']:
        for case in model['cases']:
            got = O.sample_clicks_one_list(model['model_name'], model['click_prob'], model['exam_prob'],
                                           case['labels'], case['uniforms'], model.get('extra'))
            assert [float(c) for c in got] == case['clicks'], (model['name'], case)
            n += 1
    assert n == 320 and {m['model_name'] for m in g['models']} == {
        'position_biased_model', 'cascade_model', 'user_browsing_model', 'trust_biased_model', 'click_matrix_model'}
    pbm = [m for m in g['models'] if m['name'] == 'pbm_json'][0]
    random.seed(0)
    u = [random.random() for _ in range(10)]
    assert O.sample_clicks_one_list('position_biased_model', pbm['click_prob'], pbm['exam_prob'],
                                    [4, 0, 3, 4, 2, 1, 0, 0, 1, 2], u) == g['survey_seed0_clicks']


def test_click_tables_match_reference():
    g = load_json('click_goldens.json')
    for t in g['click_prob_tables']:
        assert O.click_prob_table(*t['args']) == t['click_prob']
    for t in g['exam_tables']:
        assert O.pbm_exam_table(t['eta']) == t['exam_prob']
    # the reference's example JSONs hold setClickProb(0.1, 1.0, 4) rounded to 2 decimals
    assert np.allclose(O.click_prob_table(0.1, 1.0, 4), load_json('pbm_0.1_1.0_4_1.0.json')['click_prob'], atol=1e-12)
    assert load_json('pbm_0.1_1.0_4_1.0.json')['exam_prob'] == O.pbm_exam_table(1.0)


def test_host_click_models_match_goldens():
    from ultra_amd.utils import click_models as cm
    g = load_json('click_goldens.json')
    for model in g['models']:
        desc = {'model_name': model['model_name'], 'eta': model['eta'],
                'click_prob': model['click_prob'], 'exam_prob': model['exam_prob']}
        if model.get('extra') is not None:
            desc['extra'] = model['extra']
        m = cm.loadModelFromJson(desc)
        for case in model['cases']:
            random.seed(case['seed'])
            clicks, exam_p, click_p = m.sampleClicksForOneList(list(case['labels']))
            assert [float(c) for c in clicks] == case['clicks']
            assert [float(x) for x in exam_p] == case['exam_p'] and click_p == case['click_p']
    for t in g['click_prob_tables']:
        assert cm.PositionBiasedModel(*t['args'], 1.0).click_prob == t['click_prob']
    for t in g['exam_tables']:
        assert cm.PositionBiasedModel(0.1, 1.0, 4, t['eta']).exam_prob == t['exam_prob']
    # an unknown model name falls back to the position-biased model (click_models.py:8)
    assert cm.loadModelFromJson({'model_name': 'nonesuch', 'eta': 1, 'click_prob': [0.1], 'exam_prob': [1.0]}
                                ).model_name == 'position_biased_model'


def test_device_click_tables_hold_the_reference_thresholds():
    """What the feeds upload (device_tables) against the products sampleClick compares a uniform with: the exam_p and
    click_p the reference returned for every golden case, multiplied in float64 exactly as click_models.py does."""
    from ultra_amd.utils import click_models as cm
    g = load_json('click_goldens.json')
    for model in g['models']:
        desc = {k: model[k] for k in ('model_name', 'eta', 'click_prob', 'exam_prob')}
        if model.get('extra') is not None:
            desc['extra'] = model['extra']
        m = cm.loadModelFromJson(desc)
        kind, exam, n_exam, cp = m.device_tables(10)
        assert kind == {'position_biased_model': 0, 'cascade_model': 1, 'user_browsing_model': 4,
                        'trust_biased_model': 3, 'click_matrix_model': 3}[model['model_name']]
        if model['model_name'] == 'trust_biased_model':
            assert n_exam == 10 and len(exam) == len(model['click_prob']) * 10
            for case in model['cases']:
                for r, lab in enumerate(case['labels'][:10]):
                    grade = min(max(int(lab), 0), len(model['click_prob']) - 1)
                    assert exam[grade * n_exam + r] == case['exam_p'][r] * case['click_p'][r]
            wide = m.device_tables(15)     # the trust term keeps changing beyond the examination table
            assert wide[2] == 15 and wide[1][14] == m.getExamProb(14) * m.getClickProb(14, 0)
        elif model['model_name'] == 'user_browsing_model':
            assert n_exam == 10 and len(exam) == 100
            assert all(exam[r * 10 + d] == model['exam_prob'][r][d] for r in range(10) for d in range(r + 1))
        elif model['model_name'] == 'click_matrix_model':
            assert n_exam == 10 and exam == [x for row in model['click_prob'] for x in row]
            with pytest.raises(IndexError):
                m.device_tables(11)
    with pytest.raises(ValueError):
        cm.loadModelFromJson({'model_name': 'user_browsing_model', 'eta': 1, 'click_prob': [0.1, 1.0],
                              'exam_prob': [[1.0], [1.0]]}).device_tables(10)


def test_hparams_match_reference_goldens():
    from ultra_amd.utils.hparams import HParams
    g = load_json('hparams_goldens.json')
    for c in g['cases']:
        h = HParams(**g['defaults'][c['group']])
        try:
            h.parse(c['string'])
            res = {'values': h.values()}
        except Exception as e:  # noqa: BLE001
            res = {'error': type(e).__name__}
        assert res == {k: c[k] for k in ('values', 'error') if k in c}, c['string']


def test_propensity_lookup():
    from ultra_amd.utils.propensity_estimator import RandomizedPropensityEstimator
    est = RandomizedPropensityEstimator(os.path.join(GOLDEN, 'randomized_pbm_0.1_1.0_4_1.0.json'))
    ipw = load_json('randomized_pbm_0.1_1.0_4_1.0.json')['IPW_list']
    clicks = [1, 0, 0, 2, 0, 0, 0, 0, 0, 1]
    assert est.getPropensityForOneList(clicks) == O.propensity_for_one_list(ipw, clicks)
    assert est.getPropensityForOneList(clicks)[3] == ipw[3] and est.getPropensityForOneList(clicks)[1] == 0.0
    long = [1] * 45  # beyond the table -> last entry (propensity_estimator.py:37-40)
    assert est.getPropensityForOneList(long)[44] == ipw[-1]
    # IPW_list ~ exam_0 / exam_r for the PBM it was estimated from
    exam = load_json('pbm_0.1_1.0_4_1.0.json')['exam_prob']
    assert abs(ipw[1] - exam[0] / exam[1]) < 5e-3


def test_find_class_and_settings_names():
    from ultra_amd.utils import find_class
    assert find_class('ultra_amd.learning_algorithm.DLA').__name__ == 'DLA'
    assert find_class('ultra_amd.learning_algorithm.IPWrank').__name__ == 'IPWrank'
    assert find_class('ultra_amd.learning_algorithm.RegressionEM').__name__ == 'RegressionEM'
    assert find_class('ultra_amd.learning_algorithm.PairDebias').__name__ == 'PairDebias'
    assert find_class('ultra_amd.learning_algorithm.NavieAlgorithm').__name__ == 'NavieAlgorithm'
    assert find_class('ultra_amd.ranking_model.DNN').__name__ == 'DNN'
    assert find_class('ultra_amd.ranking_model.Linear').__name__ == 'Linear'
    assert find_class('ultra_amd.input_layer.ClickSimulationFeed').__name__ == 'ClickSimulationFeed'
    assert find_class('ultra_amd.input_layer.DirectLabelFeed').__name__ == 'DirectLabelFeed'
    with pytest.raises(ImportError):
        find_class('ultra_amd.learning_algorithm.NoSuchAlgo')


def test_raw_data_toy_both_formats():
    from ultra_amd.utils import read_data, metrics
    for sub in ('toy_ultra', 'toy_libsvm'):
        d = read_data(os.path.join(GOLDEN, sub) + '/', 'train')
        assert d.feature_size == 136 and len(d.qids) == 20 and d.rank_list_size == 9  # 27 queries, 7 invalid (data_utils.py:290)
        assert metrics.RankingMetricKey.MAX_LABEL == 2.0
        n_before = len(d.features)
        d.pad(9)
        assert len(d.features) == n_before + 1 and d.features[-1] == [0.0] * 136
        assert all(len(x) == 9 for x in d.initial_list)
        assert d.initial_list[0][:4] == [0, 1, 2, 3] and d.initial_list[0][4:] == [-1] * 5
    # ULTRA format shifts feature ids by one more than libsvm: same values, one column apart
    a = read_data(os.path.join(GOLDEN, 'toy_ultra') + '/', 'train')
    b = read_data(os.path.join(GOLDEN, 'toy_libsvm') + '/', 'train')
    assert sorted(a.labels[0]) == sorted(float(x) for x in b.labels[0])  # ULTRA lists are in SVMrank order


def test_ranklist_output(tmp_path):
    """TREC output (data_utils.py:383-501): descending score, ties in initial order, pads dropped, duplicates
    removed and unmentioned positions appended."""
    from ultra_amd.utils import generate_ranklist, generate_ranklist_by_scores, output_ranklist

    class D(object):
        qids = ['q1', 'q2']
        dids = ['d0', 'd1', 'd2', 'd3', 'd4']
        initial_list = [[0, 1, 2], [3, 4, -1]]
    scores = [[0.5, 2.0, 0.5], [1.0, 3.0, 9.0]]
    by_score = generate_ranklist_by_scores(D, scores)
    assert by_score == {'q1': [('d1', 2.0), ('d0', 0.5), ('d2', 0.5)], 'q2': [('d4', 3.0), ('d3', 1.0)]}
    assert generate_ranklist(D, [[2, 2, 0], [1, 0, 2]]) == {'q1': ['d2', 'd0', 'd1'], 'q2': ['d4', 'd3']}
    with pytest.raises(ValueError):
        generate_ranklist(D, [[0, 1, 2]])
    with pytest.raises(ValueError):
        generate_ranklist_by_scores(D, [[0.1, 0.2], [1.0, 2.0, 3.0]])
    output_ranklist(D, scores, str(tmp_path) + '/', 'valid')
    lines = open(str(tmp_path / 'valid.ranklist')).read().splitlines()
    assert lines == ['q1 Q0 d1 1 2.0 Model', 'q1 Q0 d0 2 0.5 Model', 'q1 Q0 d2 3 0.5 Model',
                     'q2 Q0 d4 1 3.0 Model', 'q2 Q0 d3 2 1.0 Model']


def test_summary_merge():
    from ultra_amd.utils import merge_TFSummary, Summary, SummaryValue
    s1 = Summary(value=[SummaryValue('ndcg_10', 0.5), SummaryValue('mrr_3', 1.0)])
    s2 = Summary(fetch=lambda: {'ndcg_10': 1.0, 'mrr_3': 0.0})
    m = merge_TFSummary([s1, s2], [1, 3]).as_dict()
    assert abs(m['ndcg_10'] - 0.875) < 1e-12 and abs(m['mrr_3'] - 0.25) < 1e-12
    assert Summary(fetch=lambda: {'Rank Loss': 1.0}).value[0].tag == 'Rank_Loss'


# ---- Philox: known-answer test (Random123 kat_vectors: philox4x32-10) -----------------------------
def test_philox_known_answers():
    z = O.philox4x32(np.zeros((1, 4), np.uint32), (0, 0))[0]
    assert [hex(int(x)) for x in z] == ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']
    f = O.philox4x32(np.full((1, 4), 0xFFFFFFFF, np.uint32), (0xFFFFFFFF, 0xFFFFFFFF))[0]
    assert [hex(int(x)) for x in f] == ['0x408f276d', '0x41c83b0e', '0xa20bc7c6', '0x6d5451fd']
    p = O.philox4x32(np.array([[0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344]], np.uint32),
                     (0xa4093822, 0x299f31d0))[0]
    assert [hex(int(x)) for x in p] == ['0xd16cfe09', '0x94fdcceb', '0x5001e420', '0x24126ea1']


def test_u53_is_mt_res53():
    a, b = np.uint32(0xDEADBEEF), np.uint32(0x12345678)
    assert O.u53(a, b) == ((0xDEADBEEF >> 5) * 67108864.0 + (0x12345678 >> 6)) / 9007199254740992.0


# ---- feed semantics --------------------------------------------------------------------------------
def test_sequential_feed_consumes_draws_like_the_reference():
    data = SyntheticData(50, 10, 8, seed=3)
    g = load_json('pbm_0.1_1.0_4_1.0.json')
    rng = np.random.RandomState(0)
    u = rng.random_sample(20000)
    idxs, clicks, n_cand = O.get_batch_sequential(data.initial_list, data.labels, 10, 16, g['model_name'],
                                                  g['click_prob'], g['exam_prob'], u, True)
    assert len(idxs) == 16 and n_cand >= 16 and all(sum(c) > 0 for c in clicks)
    # candidate k uses draws [11k, 11k+11)
    k = 0
    got = []
    while len(got) < 16:
        i = int(u[11 * k] * 50)
        c = O.sample_clicks_one_list(g['model_name'], g['click_prob'], g['exam_prob'],
                                     list(data.labels[i]), u[11 * k + 1:11 * k + 11])
        if sum(c) > 0:
            got.append((i, c))
        k += 1
    assert k == n_cand and [x[0] for x in got] == idxs


# ---- two independent derivations of the training graphs -----------------------------------------
def _np_params(rng, F, hidden):
    ps = O.init_dnn_params(rng, F, hidden)
    for p in ps:  # non-trivial LayerNorm parameters so their gradients are exercised
        p['gamma'] = rng.uniform(0.5, 1.5, size=p['gamma'].shape)
        p['beta'] = rng.uniform(-0.3, 0.3, size=p['beta'].shape)
    return ps


@pytest.mark.parametrize('act,norm', [('elu', 'layer'), ('relu', 'none'), ('tanh', 'layer'), ('selu', 'layer'),
                                      ('sigmoid', 'none')])
def test_dnn_closed_form_gradients_match_autograd(act, norm):
    rng = np.random.RandomState(1)
    F, hidden, M = 12, [16, 8], 40
    ps = _np_params(rng, F, hidden)
    x = rng.uniform(-1, 1, size=(M, F))
    d_out = rng.normal(size=(M, 1))
    scores, cache = O.dnn_forward(ps, x, act, norm)
    grads = O.dnn_backward(ps, cache, d_out, act, norm)
    tp = TG.params_from_numpy(ps)
    out = torch.cat(TG.dnn_build(tp, [torch.tensor(x)], act, norm), dim=0)
    assert relerr(out.detach().numpy(), scores) < 1e-12
    (out * torch.tensor(d_out)).sum().backward()
    for j, p in enumerate(tp):
        for k in ('W', 'b') + (('gamma', 'beta') if norm == 'layer' else ()):
            assert relerr(grads[j][k], p[k].grad.numpy()) < 1e-10, (j, k)


def _batch(rng, B, L, Q=40, F=10):
    data = SyntheticData(Q, L, F, seed=5, pad_frac=0.3)
    lists = rng.randint(0, Q, size=B)
    docids = np.where(data.initial_list[lists] < 0, data.features.shape[0], data.initial_list[lists])
    clicks = (rng.uniform(size=(B, L)) < 0.3).astype(np.float64)
    clicks[clicks.sum(1) == 0, 0] = 1.0
    return data, docids, clicks


@pytest.mark.parametrize('loss_func,prob', [('softmax_loss', 'softmax'), ('sigmoid_loss', 'softmax'),
                                            ('softmax_loss', 'sigmoid')])
def test_dla_oracle_matches_torch_graph(loss_func, prob):
    rng = np.random.RandomState(2)
    B, L, F = 24, 6, 10
    data, docids, clicks = _batch(rng, B, L, F=F)
    ps = _np_params(rng, F, [16, 8])
    prop_W, prop_b = rng.uniform(-0.5, 0.5, size=(L, 1)), rng.uniform(-0.5, 0.5, size=(1,))
    hpar = {'loss_func': loss_func, 'logits_to_prob': prob, 'max_propensity_weight': 5 if prob == 'sigmoid' else -1}
    ref = TG.TorchReferenceStep('dla', ps, L, hpar, prop_W=prop_W, prop_b=prop_b, with_metrics=False)
    table = torch.tensor(data.features.astype(np.float64))
    d_t = [torch.tensor(docids[:, l].astype(np.int64)) for l in range(L)]
    c_t = [torch.tensor(clicks[:, l]) for l in range(L)]
    loss, summ = ref.step(table, d_t, c_t)
    # numpy side
    x = O.gather_rows(data.features.astype(np.float64), docids.T.reshape(-1))
    scores, cache = O.dnn_forward(ps, x)
    out = scores.reshape(L, B).T
    r = O.dla_losses(out, clicks, prop_W, prop_b, loss_func, prob, hpar['max_propensity_weight'])
    assert abs(r['rank_loss'] - summ['Rank Loss']) < 1e-10 * max(1, abs(summ['Rank Loss']))
    assert abs(r['exam_loss'] - summ['Exam Loss']) < 1e-10 * max(1, abs(summ['Exam Loss']))
    grads = O.dnn_backward(ps, cache, r['d_output'].T.reshape(-1, 1))
    flat = [grads[j][k] for j in range(len(ps)) for k in ('W', 'b', 'gamma', 'beta')]
    clipped, nrm = O.clip_by_global_norm(flat, 5.0)
    for a, b in zip(clipped, ref.grads['ranker']):
        # the output bias gradient is identically 0 for the softmax loss (shift invariance): compare on the
        # scale of the whole gradient, not of the tensor itself
        assert np.linalg.norm(a - b.numpy()) < 1e-9 * max(np.linalg.norm(b.numpy()), 1e-6 * min(nrm, 5.0))
    dcl, _ = O.clip_by_global_norm([r['d_prop_W'], r['d_prop_b']], 5.0)
    for a, b in zip(dcl, ref.grads['denoise']):
        assert relerr(a, b.numpy()) < 1e-9


def test_rem_and_pd_oracles_match_torch_graph():
    rng = np.random.RandomState(4)
    B, L, F = 20, 5, 10
    data, docids, clicks = _batch(rng, B, L, F=F)
    ps = _np_params(rng, F, [16, 8])
    table = torch.tensor(data.features.astype(np.float64))
    d_t = [torch.tensor(docids[:, l].astype(np.int64)) for l in range(L)]
    c_t = [torch.tensor(clicks[:, l]) for l in range(L)]
    x = O.gather_rows(data.features.astype(np.float64), docids.T.reshape(-1))
    scores, cache = O.dnn_forward(ps, x)
    out = scores.reshape(L, B).T
    # REM
    u = rng.uniform(size=(B, L))
    ref = TG.TorchReferenceStep('rem', ps, L, {'max_gradient_norm': 0.0}, with_metrics=False)
    loss, _ = ref.step(table, d_t, c_t, uniforms=torch.tensor(u))
    l2, d_out, new_p, y = O.rem_step(out, clicks, np.full((1, L), 0.9), 0.05, u)
    assert abs(l2 - loss) < 1e-12 and np.array_equal(y, ref.ranker_labels.numpy())
    assert relerr(new_p, ref.propensity.numpy()) < 1e-12
    g = O.dnn_backward(ps, cache, d_out.T.reshape(-1, 1))
    assert relerr(g[0]['W'], ref.grads['ranker'][0].numpy()) < 1e-9
    assert abs(d_out.sum() - float(ref.grads['ranker'][-1])) < 1e-10
    # PD
    ref = TG.TorchReferenceStep('pd', ps, L, {'max_gradient_norm': 0.0}, with_metrics=False)
    ref.t_plus = torch.tensor(rng.uniform(0.5, 1.5, size=(1, L)))
    ref.t_minus = torch.tensor(rng.uniform(0.5, 1.5, size=(1, L)))
    tp0, tm0 = ref.t_plus.numpy().copy(), ref.t_minus.numpy().copy()
    loss, _ = ref.step(table, d_t, c_t)
    l3, d_out, ntp, ntm, _ = O.pd_step(out, clicks, tp0, tm0, 0.05, 1)
    assert abs(l3[0, 0] - loss) < 1e-10 * max(1.0, abs(loss))
    assert relerr(ntp, ref.t_plus.numpy()) < 1e-10 and relerr(ntm, ref.t_minus.numpy()) < 1e-10
    g = O.dnn_backward(ps, cache, d_out.T.reshape(-1, 1))
    assert relerr(g[1]['W'], ref.grads['ranker'][4].numpy()) < 1e-9


def test_metric_oracles_agree_and_known_answers():
    rng = np.random.RandomState(7)
    B, L = 30, 10
    labels = rng.choice(3, size=(B, L)).astype(np.float64)
    scores = np.round(rng.normal(size=(B, L)), 1)  # rounded -> plenty of exact ties
    u = rng.uniform(size=(B, L)).astype(np.float32)
    w = rng.uniform(0.1, 2.0, size=(B, 1))
    for topn in (3, 5, 10):
        a = O.ndcg(labels, scores, None, topn, u, u)
        b = float(TG.ndcg(torch.tensor(labels), torch.tensor(scores), None, topn, torch.tensor(u), torch.tensor(u)))
        assert abs(a - b) < 1e-12
        a = O.ndcg(labels, scores, w, topn, u, u)
        b = float(TG.ndcg(torch.tensor(labels), torch.tensor(scores), torch.tensor(w), topn, torch.tensor(u),
                          torch.tensor(u)))
        assert abs(a - b) < 1e-12
        a = O.err(labels, scores, 2.0, w, topn, u)
        b = float(TG.err(torch.tensor(labels), torch.tensor(scores), 2.0, torch.tensor(w), topn, torch.tensor(u)))
        assert abs(a - b) < 1e-12
    assert abs(O.mrr(labels, scores, w, u) - float(TG.mrr(torch.tensor(labels), torch.tensor(scores),
                                                          torch.tensor(w), torch.tensor(u)))) < 1e-12
    # hand-checkable: labels [0,2,1], scores rank doc1 > doc2 > doc0
    lab = np.array([[0.0, 2.0, 1.0]])
    sc = np.array([[0.1, 0.9, 0.5]])
    assert abs(O.ndcg(lab, sc, None, 3) - 1.0) < 1e-12
    sc2 = np.array([[0.9, 0.1, 0.5]])  # order doc0, doc2, doc1
    dcg = 0 / np.log(2) + 1 / np.log(3) + 3 / np.log(4)
    idcg = 3 / np.log(2) + 1 / np.log(3)
    assert abs(O.ndcg(lab, sc2, None, 3) - dcg / idcg) < 1e-12
    assert abs(O.mrr(lab, sc2) - 0.5) < 1e-12
    # ties: stable-argsort(u) then lower-index-first top_k
    perm = O.sort_permutation(np.array([[1.0, 1.0, 2.0, 1.0]]), np.array([[0.7, 0.2, 0.9, 0.5]]))
    assert perm.tolist() == [[2, 1, 3, 0]]


def test_optimizer_oracle_known_answers():
    th, acc = O.adagrad_update(np.array([1.0]), np.array([0.1]), np.array([0.3]), 0.05)
    assert abs(acc[0] - 0.19) < 1e-15 and abs(th[0] - (1.0 - 0.05 * 0.3 / np.sqrt(0.19))) < 1e-15
    cl, nrm = O.clip_by_global_norm([np.array([3.0]), np.array([4.0])], 2.5)
    assert abs(nrm - 5.0) < 1e-15 and np.allclose([cl[0][0], cl[1][0]], [1.5, 2.0])
    cl, _ = O.clip_by_global_norm([np.array([0.3])], 5.0)
    assert cl[0][0] == 0.3
    # L = 2 softmax loss by hand
    o = np.array([[0.0, np.log(3.0)]])
    loss, d, tot = O.softmax_loss(o, np.array([[1.0, 0.0]]), np.array([[1.0, 2.0]]))
    wl = np.array([1.0000001, 2e-7])
    assert abs(loss - (-(wl[0] * np.log(0.25) + wl[1] * np.log(0.75)) / wl.sum())) < 1e-12


# ---- the C ABI: library loads and exports exactly what the header declares ----------------------------
def test_library_exports_every_declared_symbol():
    from ultra_amd import _lib
    header = open(os.path.join(ROOT, 'include', 'ultr_hip.h')).read()
    declared = set(re.findall(r'\b(ultr_[a-z0-9_]+)\s*\(', header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), name
    lib = _lib.lib()
    assert lib.ultr_version() == 100
    # argument validation happens before any launch, so it can be exercised without a GPU
    assert lib.ultr_loss_workspace_floats(256, 10) == 8 + 16 * 101 + 576  # + the far rows of the two-level arrival
    assert lib.ultr_click_pbm(None, 1, 1, None, 1, None, 1, None, 0, 0, None, None) == -1
    assert b'null pointer' in lib.ultr_last_error()


def _scorer_desc(F, hidden, dtype=0):
    """The descriptor DNN.py builds for the reference's default scorer (elu, layer normalisation)."""
    from ultra_amd import _lib
    d = _lib.UltrDnnDesc()
    sizes = list(hidden) + [1]
    d.n_layers, d.in_features, d.activation, d.norm, d.ln_eps, d.compute_dtype = len(sizes), F, _lib.ACT['elu'], _lib.NORM_LAYER, 1e-3, dtype
    off, k = 0, F
    for j, n in enumerate(sizes):
        ldw = 1 if j == len(sizes) - 1 else (n + 3) // 4 * 4
        d.sizes[j] = n
        for name, length in (('W', k * ldw), ('b', n), ('gamma', k), ('beta', k)):
            getattr(d, 'off_' + name)[j] = off
            off += (length + 63) // 64 * 64
        k = n
    d.n_params = off
    return d


def test_fused_plan_splits_large_batches_into_whole_rounds_and_a_tail():
    """Host-side plan only (no launch): 32-row workgroups in whole rounds of 256, the rest on 16-row workgroups when
    that rest fits one round of them (fused.hip: make_fused_plan)."""
    from ultra_amd import _lib
    lib = _lib.lib()
    d = _scorer_desc(136, [512, 256, 128])
    rows = lambda B, L: lib.ultr_dnn_fused_rows(ctypes.byref(d), B, L)
    tail = lambda B, L: lib.ultr_dnn_fused_tail_lists(ctypes.byref(d), B, L)
    assert (rows(256, 10), tail(256, 10)) == (16, 0)
    assert (rows(257, 10), tail(257, 10)) == (32, 0)       # 86 workgroups of three lists: less than a round
    assert (rows(768, 10), tail(768, 10)) == (32, 0)       # exactly one round
    assert (rows(769, 10), tail(769, 10)) == (32, 1)
    assert (rows(1024, 10), tail(1024, 10)) == (32, 256)
    assert (rows(1025, 10), tail(1025, 10)) == (32, 0)     # 257 lists left: more than a round of 16-row workgroups
    assert (rows(4096, 10), tail(4096, 10)) == (32, 256)   # BASELINE config 5: 1280 x 3 lists + 256 x 1
    assert (rows(1300, 8), tail(1300, 8)) == (32, 276)     # four lists of 8 per 32 rows, two per 16
    assert (rows(600, 16), tail(600, 16)) == (32, 88)
    assert lib.ultr_dnn_fused_workspace_bytes(ctypes.byref(d), 4096, 10) > 0
    b = _scorer_desc(136, [512, 256, 128], dtype=_lib.DTYPE_BF16)   # the bf16 step keeps the 16-row kernel
    assert (lib.ultr_dnn_fused_rows(ctypes.byref(b), 4096, 10), lib.ultr_dnn_fused_tail_lists(ctypes.byref(b), 4096, 10)) == (16, 0)
    assert lib.ultr_dnn_fused_rows(ctypes.byref(d), 64, 17) == 0       # lists longer than 16: the per-launch plan


def test_product_does_not_import_the_oracle():
    pat = re.compile(r'^\s*(from|import)\s+[\w.]*oracle[\w.]*(\s|$)', re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'ultra_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not pat.search(src), os.path.join(dirpath, f)
                assert 'ultr_oracle' not in src and 'torch_graph' not in src, os.path.join(dirpath, f)


def test_main_flags_and_checkpoint_index(tmp_path):
    """ultra_amd.main without a GPU: the reference's flag names and defaults (main.py:24-51), the mapping of the reference's
    own `ultra.*` class names onto the drop-in classes, and the checkpoint index that stands in for
    tf.train.get_checkpoint_state (main.py:68-74)."""
    import json
    from ultra_amd import main as M
    f = M.Flags(batch_size=32, test_only=True)
    assert (f.batch_size, f.test_only, f.steps_per_checkpoint, f.selection_bias_cutoff, f.max_train_iteration) == \
        (32, True, 50, 10, 10000)
    assert f.data_dir == './tests/data/' and f.train_data_prefix == 'train' and f.max_list_cutoff == 0
    with pytest.raises(AttributeError):
        M.Flags(no_such_flag=1)
    assert M._drop_in('ultra.learning_algorithm.DLA') == 'ultra_amd.learning_algorithm.DLA'
    assert M._drop_in('my_pkg.MyAlgo') == 'my_pkg.MyAlgo'
    assert M._checkpoint_state(str(tmp_path)) is None
    ckpt = str(tmp_path / 'ultra.learning_algorithm.DLA.ckpt-50')
    json.dump({'model_checkpoint_path': ckpt}, open(str(tmp_path / 'checkpoint'), 'w'))
    assert M._checkpoint_state(str(tmp_path)) is None          # the index names a file that is not there
    open(ckpt + '.pt', 'wb').close()
    assert M._checkpoint_state(str(tmp_path)) == ckpt
    # every feed / algorithm / ranker the reference's example settings name resolves to a drop-in class by dotted name
    from ultra_amd.utils import find_class
    for name in ('ultra.input_layer.ClickSimulationFeed', 'ultra.input_layer.DirectLabelFeed',
                 'ultra.input_layer.DeterministicOnlineSimulationFeed', 'ultra.input_layer.StochasticOnlineSimulationFeed',
                 'ultra.learning_algorithm.DLA', 'ultra.learning_algorithm.IPWrank', 'ultra.learning_algorithm.RegressionEM',
                 'ultra.learning_algorithm.PairDebias', 'ultra.learning_algorithm.NavieAlgorithm',
                 'ultra.ranking_model.DNN', 'ultra.ranking_model.Linear'):
        assert find_class(M._drop_in(name)).__name__ == name.rsplit('.', 1)[1]
