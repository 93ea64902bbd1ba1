"""GPU pipeline tests (-m gpu): the reference's own CPU-runnable case and "NDCG@10 after a fixed step count".

* BASELINE config 0: example/toy offline_exp_pipeline -- DNN + IPW on the toy libsvm data with PBM clicks
  (tests/test_settings/dnn.json-style settings), train feed drawing from Python's `random` exactly like the
  reference's ClickSimulationFeed, validation through DirectLabelFeed + merge_TFSummary (main.py:116-167).
* BASELINE config 1 at full shape (F = 136, DNN[512,256,128], batch 256, DLA + PBM): after a fixed number of steps on
  identical batches the device model and the op-for-op reference graph (float64) report the same NDCG@10 on the
  true labels (north_star: "reference-matching NDCG@10 after a fixed step count").
"""
import os
import random

import numpy as np
import pytest
import torch

from oracle import ultr_oracle as O
from tests.util import GOLDEN, SyntheticData
from tests.test_gpu_step import (IPW_JSON, PBM, compare_params, make, reference_for, settings, tie_uniforms,
                                 torch_feed)

pytestmark = pytest.mark.gpu


def _validate(model, data, exp, batch):
    """main.py:153-167: get_next_batch over the set, step(forward_only=True), merge_TFSummary weighted by size."""
    from ultra_amd.utils import find_class, merge_TFSummary
    vfeed = find_class(exp['valid_input_feed'])(model, batch, exp['valid_input_hparams'], None)
    it, sums, sizes, scores = 0, [], [], []
    while it < len(data.initial_list):
        feed_dict, info = vfeed.get_next_batch(it, data, check_validation=False)
        _, s, summary = model.step(None, feed_dict, True)
        sums.append(summary)
        sizes.append(len(info['input_list']))
        scores.append(s)
        it += sizes[-1]
    return merge_TFSummary(sums, sizes).as_dict(), np.concatenate(scores, 0)


def _reference_validation(ref, features, initial_list, labels, act='elu'):
    """NDCG@10 / MRR of the reference graph's current parameters over the whole set, true labels, pads masked."""
    ps = [{k: v.detach().numpy() for k, v in p.items()} for p in ref.params]
    n_rows = features.shape[0] - 1                       # the last row is the zero pad row
    docids = np.where(initial_list < 0, n_rows, initial_list)
    scores, _ = O.dnn_forward(ps, O.gather_rows(features.astype(np.float64), docids.reshape(-1)))
    scores = scores.reshape(docids.shape)
    masked = O.remove_padding(scores, docids, n_rows)
    lab = np.where(initial_list < 0, 0, labels).astype(np.float64)
    return {'ndcg_10': O.ndcg(lab, masked, None, 10), 'ndcg_3': O.ndcg(lab, masked, None, 3), 'mrr_10': O.mrr(lab, masked)}


class _ArrayView(object):
    """Raw_data (lists of lists) seen as the arrays the oracle side indexes."""

    def __init__(self, d, Lmax):
        self.features = np.asarray(d.features, dtype=np.float32)
        self.initial_list = np.full((len(d.initial_list), Lmax), -1, dtype=np.int32)
        self.labels = np.zeros((len(d.initial_list), Lmax), dtype=np.float32)
        for i, (ids, labs) in enumerate(zip(d.initial_list, d.labels)):
            self.initial_list[i, :len(ids)] = ids
            self.labels[i, :min(len(labs), Lmax)] = labs[:Lmax]


def test_toy_offline_pipeline_dnn_ipw_libsvm_pbm():
    from ultra_amd.learning_algorithm.base_algorithm import DEVICE_BATCH_KEY
    from ultra_amd.utils import read_data, metrics
    train = read_data(os.path.join(GOLDEN, 'toy_libsvm') + '/', 'train')
    Lmax = train.rank_list_size
    train.pad(Lmax)
    exp = settings('IPWrank', 'propensity_estimator_json=%s' % IPW_JSON, model_hparams='hidden_layer_sizes=[512, 256, 128]',
                   Lmax=Lmax, cutoff=Lmax, feed_hparams='click_model_json=%s,device_rng=False' % PBM)
    model, feed = make('IPWrank', train, exp, 16)
    metrics.RankingMetricKey.MAX_LABEL = 2.0             # what loading the toy data sets (data_utils.py:92)
    view = _ArrayView(train, Lmax)
    ref = reference_for('ipw', model, Lmax, {})
    ref.max_label = 2.0
    ipw = model.propensity_estimator.IPW_list
    random.seed(42)
    before, _ = _validate(model, train, exp, 8)
    for step in range(12):
        feed_dict, info = feed.get_batch(train, True)
        batch = feed_dict[DEVICE_BATCH_KEY]
        table, d, c = torch_feed(view, batch)
        clicks = batch.labels.cpu().numpy()
        assert clicks.sum(axis=0).min() >= 1             # check_validation: every list has a click
        pw = [torch.tensor([O.propensity_for_one_list(ipw, list(clicks[:, b]))[l] for b in range(clicks.shape[1])])
              for l in range(Lmax)]
        u = tie_uniforms(model, np.random.RandomState(step), 16, Lmax)
        loss, _, summary = model.step(None, feed_dict, False)
        ref_loss, ref_sum = ref.step(table, d, c, pw_lists=pw, u_pred=u)
        assert abs(float(loss) - ref_loss) < 1e-5 * abs(ref_loss), (step, float(loss), ref_loss)
        got = summary.as_dict()
        assert abs(got['ndcg_10'] - ref_sum['ndcg_10']) < 2e-5 and abs(got['Weighted_ndcg_10'] - ref_sum['Weighted_ndcg_10']) < 2e-5
    compare_params(model, ref, 3e-5)
    after, scores = _validate(model, train, exp, 8)
    assert scores.shape == (len(train.initial_list), Lmax)
    want = _reference_validation(ref, view.features, view.initial_list, view.labels)
    for tag in ('ndcg_10', 'ndcg_3', 'mrr_10'):
        assert abs(after[tag] - want[tag]) < 5e-5, (tag, after[tag], want[tag])
    assert after['ndcg_10'] != before['ndcg_10']        # the model moved


def test_dla_ndcg10_after_fixed_step_count_matches_reference():
    from ultra_amd.learning_algorithm.base_algorithm import DEVICE_BATCH_KEY
    STEPS, B, F = 40, 256, 136
    data = SyntheticData(800, 10, F, seed=5)
    rng = np.random.RandomState(5)
    w = rng.normal(size=F) / np.sqrt(F)                  # labels that can be learnt: a noisy linear relevance signal
    rel = data.features[:-1] @ w * 3.0 + rng.normal(size=data.features.shape[0] - 1) * 0.3
    cuts = np.quantile(rel, [0.52, 0.84, 0.97, 0.99])
    data.labels = np.searchsorted(cuts, rel).astype(np.float32).reshape(800, 10)
    exp = settings('DLA', '', model_hparams='hidden_layer_sizes=[512, 256, 128]')
    model, feed = make('DLA', data, exp, B)
    pp = model.prop_params.cpu().numpy().astype(np.float64)
    ref = reference_for('dla', model, 10, {}, with_metrics=False, prop_W=pp[:10, None], prop_b=pp[10:])
    before, _ = _validate(model, data, exp, 256)
    for step in range(STEPS):
        feed_dict, _ = feed.get_batch(data, True)
        table, d, c = torch_feed(data, feed_dict[DEVICE_BATCH_KEY])
        loss, _, _ = model.step(None, feed_dict, False)
        ref_loss, _ = ref.step(table, d, c)
        assert abs(float(loss) - ref_loss) < 2e-5 * abs(ref_loss), (step, float(loss), ref_loss)
    after, _ = _validate(model, data, exp, 256)
    want = _reference_validation(ref, data.features, data.initial_list, data.labels)
    assert after['ndcg_10'] > before['ndcg_10'] + 0.02, (before['ndcg_10'], after['ndcg_10'])  # it learnt from clicks
    assert abs(after['ndcg_10'] - want['ndcg_10']) < 1e-4, (after['ndcg_10'], want['ndcg_10'])
    assert abs(after['ndcg_3'] - want['ndcg_3']) < 2e-4



def test_main_train_validate_checkpoint_and_test(tmp_path, capsys):
    """ultra_amd.main: the reference's own settings file layout (tests/test_settings/dnn.json: `ultra.*` class names,
    objective_metric ndcg_10) driven through train() -- feed, steps, validation every steps_per_checkpoint steps, the
    best-checkpoint rule (main.py:180-194) -- then test() from that checkpoint: merged summary and TREC rank lists."""
    import json
    from ultra_amd import main as M
    exp = {
        'train_input_feed': 'ultra.input_layer.ClickSimulationFeed', 'train_input_hparams': 'click_model_json=%s,seed=3' % PBM,
        'valid_input_feed': 'ultra.input_layer.DirectLabelFeed', 'valid_input_hparams': '',
        'test_input_feed': 'ultra.input_layer.DirectLabelFeed', 'test_input_hparams': '',
        'ranking_model': 'ultra.ranking_model.DNN', 'ranking_model_hparams': 'hidden_layer_sizes=[64, 32]',
        'learning_algorithm': 'ultra.learning_algorithm.DLA', 'learning_algorithm_hparams': '',
        'metrics': ['mrr', 'ndcg'], 'metrics_topn': [3, 5, 10], 'objective_metric': 'ndcg_10', 'seed': 1}
    flags = M.Flags(data_dir=os.path.join(GOLDEN, 'toy_ultra') + '/', model_dir=str(tmp_path / 'model'),
                    output_dir=str(tmp_path / 'out') + '/', batch_size=32, steps_per_checkpoint=8, max_train_iteration=30,
                    test_while_train=True)
    history = M.train(json.loads(json.dumps(exp)), flags)
    out = capsys.readouterr().out
    assert [h['step'] for h in history] == [8, 16, 24, 32]          # stops at the first interval past max_train_iteration
    assert 'Users can only see the top 9 documents' in out and 'global step 8 learning rate 0.0500' in out
    assert all(np.isfinite(h['loss']) and set(h['valid']) >= {'ndcg_10', 'mrr_3'} and 'test' in h for h in history)
    # a checkpoint is written exactly when validation ndcg_10 improves on the best so far
    best, want_saved = None, []
    for h in history:
        better = best is None or best < h['valid']['ndcg_10']
        want_saved.append(better)
        best = h['valid']['ndcg_10'] if better else best
    assert [h['saved'] is not None for h in history] == want_saved and want_saved[0]
    last_saved = [h for h in history if h['saved']][-1]
    assert os.path.exists(last_saved['saved'] + '.pt')
    # test(): a forward-only model restored from that checkpoint reproduces the validation numbers of that interval on the
    # test split the training loop evaluated, and writes one TREC line per (query, document)
    flags.test_only = True
    summary, scores = M.test(json.loads(json.dumps(exp)), flags)
    assert 'Reading model parameters from' in capsys.readouterr().out
    for tag, v in last_saved['test'].items():
        assert abs(summary[tag] - v) < 1e-6, (tag, summary[tag], v)
    lines = open(os.path.join(flags.output_dir, 'test.ranklist')).read().strip().split('\n')
    from ultra_amd.utils import read_data
    test_set = read_data(flags.data_dir, 'test')
    assert len(scores) == len(test_set.initial_list) and len(lines) == sum(len(x) for x in test_set.initial_list)


def test_dla_ndcg10_on_a_held_out_split_after_1000_steps():
    """NDCG@10 on a held-out synthetic split after the fixed step count of the reference's MSLR pipeline
    (1000, example/MSLR_WEB10k/offline_exp_pipeline.sh:79), against the op-for-op reference graph (float64) run from the
    same initial weights on the same batches.  fp32 and float64 trajectories drift apart slowly over a thousand Adagrad
    steps, so the bound here is on the metric (and on the loss of the last step), not on every parameter."""
    import copy
    import time
    from ultra_amd.learning_algorithm.base_algorithm import DEVICE_BATCH_KEY
    STEPS, B, F = 1000, 256, 136
    data = SyntheticData(800, 10, F, seed=7)
    rng = np.random.RandomState(7)
    w = rng.normal(size=F) / np.sqrt(F)
    rel = data.features[:-1] @ w * 3.0 + rng.normal(size=data.features.shape[0] - 1) * 0.3
    cuts = np.quantile(rel, [0.52, 0.84, 0.97, 0.99])
    data.labels = np.searchsorted(cuts, rel).astype(np.float32).reshape(800, 10)
    train, held = copy.copy(data), copy.copy(data)
    train.initial_list, train.labels, train.qids = data.initial_list[:600], data.labels[:600], data.qids[:600]
    held.initial_list, held.labels, held.qids = data.initial_list[600:], data.labels[600:], data.qids[600:]
    exp = settings('DLA', '', model_hparams='hidden_layer_sizes=[512, 256, 128]')
    model, feed = make('DLA', train, exp, B)
    pp = model.prop_params.cpu().numpy().astype(np.float64)
    ref = reference_for('dla', model, 10, {}, with_metrics=False, prop_W=pp[:10, None], prop_b=pp[10:])
    before, _ = _validate(model, held, exp, 200)
    t0 = time.time()
    for step in range(STEPS):
        feed_dict, _ = feed.get_batch(train, True)
        table, d, c = torch_feed(train, feed_dict[DEVICE_BATCH_KEY])
        loss, _, _ = model.step(None, feed_dict, False)
        ref_loss, _ = ref.step(table, d, c)
    print('1000 steps on both sides: %.1f s' % (time.time() - t0))
    assert abs(float(loss) - ref_loss) < 2e-3 * abs(ref_loss), (float(loss), ref_loss)
    after, _ = _validate(model, held, exp, 200)
    want = _reference_validation(ref, held.features, held.initial_list, held.labels)
    print('held-out ndcg_10: before %.4f, after %.4f, reference %.4f' % (before['ndcg_10'], after['ndcg_10'], want['ndcg_10']))
    assert after['ndcg_10'] > before['ndcg_10'] + 0.05                      # it learnt to rank unseen queries from clicks
    assert abs(after['ndcg_10'] - want['ndcg_10']) < 2e-3, (after['ndcg_10'], want['ndcg_10'])
    assert abs(after['ndcg_3'] - want['ndcg_3']) < 4e-3
