"""GPU parity tests (-m gpu) at BASELINE.json's FULL configuration sizes.

The float64 oracle is too slow for 40,960-row batches, so the big cases are checked through size-independent
properties of the path: linearity of the backward pass in d_scores, shard additivity (the sum of the
un-normalised gradients of two half batches is the gradient of the whole batch -- the identity the data-parallel
all-reduce relies on), determinism, sortedness/idempotence of the metric sort, NDCG of an ideal ranking = 1.  The
dataset-shaped configs that DO fit the oracle (Yahoo F=700 + REM, Istella-S F=220 + PairDebias on cascade clicks,
with the real [512, 256, 128] scorer) run one full step against the op-for-op reference graph.
"""
import numpy as np
import pytest
import torch

from oracle import ultr_oracle as O
from tests.util import SyntheticData, relerr
from tests.test_gpu_step import (CASCADE, PBM, compare_params, make, reference_for, settings, tie_uniforms,
                                 torch_feed)

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def _big_dnn(F, B, L, seed):
    from ultra_amd.ranking_model import DNN
    from ultra_amd.ranking_model.DNN import feature_row_stats
    g = torch.Generator(device='cuda').manual_seed(seed)
    n_rows = 60000
    ldx = (F + 3) // 4 * 4
    table = torch.zeros((n_rows, ldx), dtype=torch.float32, device='cuda')
    table[:, :F] = torch.rand((n_rows, F), device='cuda', generator=g) * 2 - 1
    stats = feature_row_stats(table, F)
    docids = torch.randint(0, n_rows + 1, (L, B), device='cuda', generator=g, dtype=torch.int32)  # n_rows = pad row
    model = DNN('hidden_layer_sizes=[512, 256, 128]')
    model.create_variables(F, 'cuda', seed=seed)
    return model, table, stats, n_rows, docids, g


@pytest.mark.parametrize('F,B', [(136, 4096), (700, 1024)])  # MSLR-30K per-GPU batch; Yahoo width
def test_backward_is_linear_and_shard_additive_at_full_size(F, B):
    L = 10
    model, table, stats, n_rows, docids, g = _big_dnn(F, B, L, seed=F)
    d1 = torch.randn((L, B), device='cuda', generator=g)
    d2 = torch.randn((L, B), device='cuda', generator=g)

    def grads(dids, d):
        model.forward_lists(table, stats, n_rows, dids)
        out = torch.zeros(model.n_total, dtype=torch.float32, device='cuda')
        model.backward(d.contiguous(), out)
        return out[:model.desc.n_params].double()

    g1, g2, g12 = grads(docids, d1), grads(docids, d2), grads(docids, d1 + d2)
    scale = g12.norm().item()
    assert scale > 0 and torch.isfinite(g12).all()
    assert (g1 + g2 - g12).norm().item() < RTOL * scale                 # linearity in d_scores
    assert torch.equal(grads(docids, d1 + d2), g12)                      # deterministic: fixed summation order
    h = B // 2                                                           # two "ranks" with half the lists each
    ga = grads(docids[:, :h].contiguous(), (d1 + d2)[:, :h])
    gb = grads(docids[:, h:].contiguous(), (d1 + d2)[:, h:])
    assert (ga + gb - g12).norm().item() < RTOL * scale                 # shard additivity (what DP all-reduces)


@pytest.mark.parametrize('B', [4096, 1024])  # per-layer forward kernels; all layers in one launch (chain kernel)
def test_scores_do_not_depend_on_batch_composition(B):
    """A list's scores are a function of its own documents only: scoring B lists at once or in two halves, or
    with the lists permuted, gives the same numbers (bit for bit: per-row arithmetic has a fixed order within a
    launch plan; the plan changes with the row count at M = 16384, dnn.hip)."""
    L = 10
    model, table, stats, n_rows, docids, g = _big_dnn(136, B, L, seed=7)
    docids[3, ::97] = n_rows                                             # some pad documents for certain
    full = model.forward_lists(table, stats, n_rows, docids).clone()
    a = model.forward_lists(table, stats, n_rows, docids[:, :B // 2].contiguous()).clone()
    b = model.forward_lists(table, stats, n_rows, docids[:, B // 2:].contiguous()).clone()
    assert torch.equal(torch.cat([a, b], dim=1), full)
    perm = torch.randperm(B, device='cuda', generator=g)
    assert torch.equal(model.forward_lists(table, stats, n_rows, docids[:, perm].contiguous()), full[:, perm])
    pad = docids == n_rows                                               # the zero pad row scores like a zero document
    zero_scores = full[pad]
    assert zero_scores.numel() > 0 and (zero_scores == zero_scores[0]).all()


def test_metric_sort_properties_at_full_size():
    from ultra_amd.utils import metrics as M
    L, B = 10, 4096
    g = torch.Generator(device='cuda').manual_seed(3)
    scores = torch.randn((L, B), device='cuda', generator=g)
    scores[3] = scores[2]                                                # exact ties in every list
    u = torch.rand((B, L), device='cuda', generator=g)
    perm = M.sort_by_scores_indices(scores.t().contiguous(), u, L).long()  # [B, L]
    assert torch.equal(perm.sort(dim=1).values, torch.arange(L, device='cuda').expand(B, L))  # a permutation
    s_bl = scores.t()
    sorted_scores = s_bl.gather(1, perm)
    assert (sorted_scores[:, 1:] <= sorted_scores[:, :-1]).all()        # sortedness
    tie = sorted_scores[:, 1:] == sorted_scores[:, :-1]
    su = u.gather(1, perm)
    assert (su[:, 1:][tie] >= su[:, :-1][tie]).all()                    # ties: uniform ascending (metric_utils.py:76-86)
    again = M.sort_by_scores_indices(sorted_scores.contiguous(), su.contiguous(), L).long()
    assert torch.equal(again, torch.arange(L, device='cuda').expand(B, L))  # idempotence
    # NDCG of the ideal ranking is exactly 1 for every list with a relevant document, MRR is 1 when the top is relevant
    labels = torch.randint(0, 5, (L, B), device='cuda', generator=g).float()
    plan = M.MetricPlan(['mrr', 'ndcg'], [3, 5, 10])
    out = torch.zeros(plan.n_out, device='cuda')
    ws = M.metrics_workspace(B, plan, 'cuda')
    docids = torch.zeros((L, B), dtype=torch.int32, device='cuda')
    M.batch_metrics(plan, labels.clone(), labels, docids, 10, None, out, ws, None, max_label=4.0)
    vals = dict((tag, out[slot].item()) for tag, slot in plan.tags(False))
    frac = (labels.sum(dim=0) > 0).float().mean().item()
    for k in (3, 5, 10):
        assert abs(vals['ndcg_%d' % k] - frac) < 1e-6, (k, vals, frac)


def _one_step_vs_reference(algo, kind, F, click_json, B, extra_check=None):
    from ultra_amd.learning_algorithm.base_algorithm import DEVICE_BATCH_KEY
    data = SyntheticData(500, 10, F, seed=F, pad_frac=0.1)
    exp = settings(algo, '', model_hparams='hidden_layer_sizes=[512, 256, 128]',
                   feed_hparams='click_model_json=%s' % click_json)
    model, feed = make(algo, data, exp, B)
    ref = reference_for(kind, model, 10, {})
    rng = np.random.RandomState(F)
    for step in range(2):
        feed_dict, _ = feed.get_batch(data, True)
        batch = feed_dict[DEVICE_BATCH_KEY]
        table, d, c = torch_feed(data, batch)
        ut = tie_uniforms(model, rng, B)
        kw = {}
        if kind == 'rem':  # TF's graph RNG is not reproducible: both sides get the same Bernoulli uniforms (q15)
            u = rng.uniform(size=(10, B)).astype(np.float32)
            model.uniforms = torch.tensor(u).cuda()
            kw['uniforms'] = torch.tensor(u.T.astype(np.float64))
        loss, _, summary = model.step(None, feed_dict, False)
        ref_loss, ref_sum = ref.step(table, d, c, u_pred=ut, **kw)
        assert abs(float(loss) - ref_loss) < 1e-5 * abs(ref_loss), (step, float(loss), ref_loss)
        got = summary.as_dict()
        assert abs(got['Gradient_Norm'] - ref_sum['Gradient Norm']) < 2e-5 * ref_sum['Gradient Norm']
        assert abs(got['ndcg_10'] - ref_sum['ndcg_10']) < 2e-5
    compare_params(model, ref, 3e-5)
    return model, ref


def test_yahoo_shape_regression_em_step():
    """BASELINE config 3: F = 700, DNN[512,256,128] + regression-EM, PBM clicks."""
    model, ref = _one_step_vs_reference('RegressionEM', 'rem', 700, PBM, 128)
    assert relerr(model.propensity.cpu().numpy(), ref.propensity.numpy()[0]) < 1e-5


def test_istella_shape_pairwise_debias_step():
    """BASELINE config 4: F = 220, DNN[512,256,128] + PairDebias on cascade clicks."""
    model, ref = _one_step_vs_reference('PairDebias', 'pd', 220, CASCADE, 128)
    assert relerr(model.t_plus.cpu().numpy(), ref.t_plus.numpy()[0]) < 1e-5
    assert relerr(model.t_minus.cpu().numpy(), ref.t_minus.numpy()[0]) < 1e-5
