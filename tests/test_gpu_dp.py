"""GPU tests (-m gpu) of the PRODUCT's data-parallel step with two ranks' worth of state on one GPU: two algorithm
instances with world = 2 and rank = 0 / 1 forced (no process group), B lists each; `step_local` on both, the reduced
range of the two buckets (`_train_bufs['reduce']`) summed by hand and written back to both (what the one RCCL all-reduce does),
`step_apply` on both -- against ONE instance stepping the 2B-list batch.  Clicks / doc ids bit-exact (slot s of rank r
is global slot r*B + s of the Philox stream), parameters, EM state, loss and every summary to 1e-6 relative with a
small absolute floor on the first step (same variables on both sides, fp32 sums in another order), 2e-5 on the later
steps (the rounding differences of the first update feed back through exp / softmax).  Reference: the single-process graph of dla.py:237-255,
regression_EM.py:151-155, pairwise_debias.py:162-171, ipw_rank.py:120-125 evaluated on the whole batch."""
import numpy as np
import pytest
import torch

from tests.util import SyntheticData, relerr_floor
from tests.test_gpu_step import settings, CASCADE, IPW_JSON, PBM

pytestmark = pytest.mark.gpu


def make_rank(algo, data, exp, B, world, rank):
    from ultra_amd.utils import find_class, metrics
    metrics.RankingMetricKey.MAX_LABEL = 4.0
    model = find_class(exp['learning_algorithm'])(data, exp, False)
    model.world, model.rank = world, rank  # forced: what torch.distributed would have reported
    feed = find_class(exp['train_input_feed'])(model, B, exp['train_input_hparams'], None)
    return model, feed


CASES = [
    ('DLA', '', 'hidden_layer_sizes=[64, 32]', 24, 10, 48),
    ('DLA', '', 'hidden_layer_sizes=[512, 256, 128]', 136, 10, 48),
    ('IPWrank', 'propensity_estimator_json=%s' % IPW_JSON, 'hidden_layer_sizes=[64, 32]', 24, 10, 48),
    ('RegressionEM', '', 'hidden_layer_sizes=[64, 32]', 24, 10, 48),
    ('PairDebias', '', 'hidden_layer_sizes=[64, 32]', 24, 10, 48),
    ('DLA', '', 'hidden_layer_sizes=[64, 32]', 24, 20, 48),   # lists longer than 16: the per-launch plan
    # BASELINE config 4 (MSLR shape, 4096 lists per step): the whole step -- feed, step, update -- at full size; the
    # 4096-list instance runs the per-launch plan, its 2048-list shards too (both beyond the fused step's row limit)
    ('DLA', '', 'hidden_layer_sizes=[512, 256, 128]', 136, 10, 2048),
]


@pytest.mark.parametrize('algo,hp,mhp,F,L,B', CASES)
def test_two_ranks_on_one_gpu_equal_one_rank_with_the_whole_batch(algo, hp, mhp, F, L, B):
    from ultra_amd.learning_algorithm.base_algorithm import DEVICE_BATCH_KEY
    data = SyntheticData(300 if B < 1000 else 3000, L, F, seed=5, pad_frac=0.2)
    click = CASCADE if algo == 'PairDebias' else PBM
    exp = settings(algo, hp, mhp, Lmax=L, cutoff=L, feed_hparams='click_model_json=%s,seed=77' % click)
    whole, feed_w = make_rank(algo, data, exp, 2 * B, 1, 0)
    ranks = [make_rank(algo, data, exp, B, 2, r) for r in range(2)]
    rng = np.random.RandomState(9)
    for step in range(3):
        tol = 1e-6 if step == 0 else 2e-5
        u = torch.tensor(rng.uniform(size=(2 * B, L)).astype(np.float32)).cuda()
        fw, _ = feed_w.get_batch(data, check_validation=True)
        bw = fw[DEVICE_BATCH_KEY]
        whole.metric_uniforms = u
        feeds = []
        for r, (m, feed) in enumerate(ranks):
            fr, _ = feed.get_batch(data, check_validation=True)
            br = fr[DEVICE_BATCH_KEY]
            # the shards ARE the halves of the whole batch, bit for bit
            assert torch.equal(br.docids, bw.docids[:, r * B:(r + 1) * B]), (step, r)
            assert torch.equal(br.labels, bw.labels[:, r * B:(r + 1) * B]), (step, r)
            m.metric_uniforms = u[r * B:(r + 1) * B].contiguous()
            feeds.append(fr)
        lw, _, sw = whole.step(None, fw, False)
        for (m, _), fr in zip(ranks, feeds):
            m.step_local(fr)
        total = ranks[0][0]._train_bufs['reduce'] + ranks[1][0]._train_bufs['reduce']  # what the collective would leave
        outs = []
        for m, _ in ranks:
            m._train_bufs['reduce'].copy_(total)
            outs.append(m.step_apply())
        want = sw.as_dict()
        for r, (lr, _, sr) in enumerate(outs):
            got = sr.as_dict()
            assert got.keys() == want.keys()
            for k in want:
                if 'output_score' in k:
                    continue  # rank-local by definition (navie_algorithm.py:93-119 summarises the local batch)
                a, b = float(got[k]), float(want[k])
                assert (np.isnan(a) and np.isnan(b)) or abs(a - b) <= tol * abs(b) + 2 * tol, (step, r, k, a, b)
            assert abs(float(lr) - float(lw)) <= tol * abs(float(lw)) + tol, (step, r)
        for r, (m, _) in enumerate(ranks):
            assert relerr_floor(m.model.params.cpu().numpy(), whole.model.params.cpu().numpy(), 1e-3) < 2 * tol, (step, r)
            for k, v in m._extra_state().items():
                w = whole._extra_state()[k]
                ok = torch.isfinite(w)
                assert relerr_floor(v[ok].cpu().numpy(), w[ok].cpu().numpy(), 1e-3) < 2 * tol, (step, r, k)
    # both replicas end bit-identical to each other: they applied the same bucket to the same variables
    assert torch.equal(ranks[0][0].model.params, ranks[1][0].model.params)


def test_ipw_propensity_weight_summary_is_a_maximum_not_a_mean():
    """ipw_rank.py:120-125 publishes reduce_max of the weights per position: with one rank seeing a click at a position
    and the other not, the summed bucket must still read the table entry (not half of it)."""
    from ultra_amd.learning_algorithm.base_algorithm import DEVICE_BATCH_KEY
    data = SyntheticData(100, 10, 24, seed=5)
    exp = settings('IPWrank', 'propensity_estimator_json=%s' % IPW_JSON)
    m, feed = make_rank('IPWrank', data, exp, 16, 2, 0)
    fr, _ = feed.get_batch(data, check_validation=True)
    batch = fr[DEVICE_BATCH_KEY]
    batch.labels.zero_()
    batch.labels[0].fill_(1.0)   # clicks at position 0 only (this rank; the imagined other rank saw none anywhere)
    m.step_local(fr)
    _, _, s = m.step_apply()     # the bucket holds this rank's contribution + zeros
    d = s.as_dict()
    table = m.ipw_table.cpu().numpy()
    assert d['Propensity_weights_0'] == pytest.approx(table[0], rel=1e-6)
    assert d['Propensity_weights_3'] == 0.0


def test_second_feed_on_a_model_shares_its_step_counter():
    """Regression-EM keys its Bernoulli labels on the device step counter of the feed; a second feed built on the same
    model must not replace that counter (the model would read one that the training feed no longer advances)."""
    data = SyntheticData(100, 10, 24, seed=5)
    exp = settings('RegressionEM')
    m, feed = make_rank('RegressionEM', data, exp, 16, 1, 0)
    ctr = m.step_ctr
    _, feed2 = None, type(feed)(m, 8, exp['train_input_hparams'], None)
    assert m.step_ctr is ctr and feed2.step_ctr is ctr and feed.step_ctr is ctr
    before = int(ctr.item())
    fr, _ = feed.get_batch(data, check_validation=True)
    m.step(None, fr, False)
    torch.cuda.synchronize()
    assert int(ctr.item()) > before


def test_whole_step_at_4096_lists_against_the_float64_oracle():
    """BASELINE config 4: feed -> step -> update at 4096 lists of 10 (MSLR width, DNN[512,256,128]) on the device against
    the float64 restatement of the reference graph; loss and summaries 1e-5, parameters 3e-5 after the update."""
    from ultra_amd.learning_algorithm.base_algorithm import DEVICE_BATCH_KEY
    from tests.test_gpu_step import reference_for, torch_feed, tie_uniforms, compare_params, make
    B, L, F = 4096, 10, 136
    data = SyntheticData(3000, L, F, seed=6, pad_frac=0.1)
    exp = settings('DLA', '', 'hidden_layer_sizes=[512, 256, 128]', Lmax=L, cutoff=L)
    m, feed = make('DLA', data, exp, B)
    hpd = {k: getattr(m.hparams, k) for k in ('learning_rate', 'max_gradient_norm', 'loss_func', 'logits_to_prob',
                                               'propensity_learning_rate', 'ranker_loss_weight', 'l2_loss',
                                               'max_propensity_weight', 'grad_strategy')}
    pp = m.prop_params.cpu().numpy().astype(np.float64)
    ref = reference_for('dla', m, L, hpd, prop_W=pp[:L, None], prop_b=pp[L:])
    fd, _ = feed.get_batch(data, check_validation=True)
    batch = fd[DEVICE_BATCH_KEY]
    assert batch.B == B and (batch.labels.sum(dim=0) > 0).all()
    u = tie_uniforms(m, np.random.RandomState(1), B, L)
    loss, _, summary = m.step(None, fd, False)
    ref_loss, ref_sum = ref.step(*torch_feed(data, batch), u_pred=u)
    assert abs(float(loss) - ref_loss) < 1e-5 * abs(ref_loss)
    got = summary.as_dict()
    for tag, val in ref_sum.items():
        key = tag.replace(' ', '_')
        assert abs(got[key] - val) <= 2e-5 * max(abs(val), 1e-2), (key, got[key], val)
    compare_params(m, ref, 3e-5)
