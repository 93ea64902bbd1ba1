"""Data-parallel path on CPU: world_size 2 over gloo.

The GPU kernels cannot run here, so the ranks compute their shard with the oracle; what is under test is the
product's DP contract (ultra_amd/parallel.py): (1) rank r's sampling slots are the global slots r*B .. r*B+B-1 of
the Philox stream, so two ranks x B lists draw exactly the lists and clicks of one rank x 2B; (2) summing the
un-normalised [gradient | statistics] bucket with ONE all-reduce and dividing once reproduces the full-batch
listwise softmax loss and gradient (base_algorithm.py:242-251), which averaging per-rank losses does not.
"""
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ultr_oracle as O  # noqa: E402
from tests.util import GOLDEN, SyntheticData  # noqa: E402
from ultra_amd import parallel  # noqa: E402

B, L, F, SEED, STEP = 12, 6, 9, 77, 5


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _click_tables():
    g = json.load(open(os.path.join(GOLDEN, 'pbm_0.1_1.0_4_1.0.json')))
    return g['model_name'], g['click_prob'], g['exam_prob']


def _sample(data, slot_base, n):
    name, cp, ep = _click_tables()
    idx, clicks, _ = O.get_batch_slots(data.initial_list, data.labels, L, n, name, cp, ep, SEED, STEP, slot_base=slot_base)
    return np.array(idx), np.array(clicks, dtype=np.float64)


def _shard_terms(data, w, idx, clicks, pw):
    """Un-normalised pieces of the IPW softmax loss for a linear scorer w: (sum of CE*S_b, sum wl, d numerator / d w)."""
    docs = data.initial_list[idx][:, :L]
    x = np.where(docs[..., None] >= 0, data.features[np.maximum(docs, 0)], 0.0).astype(np.float64)  # [n, L, F]
    scores = x @ w
    loss, d_out, total = O.softmax_loss(scores, clicks.astype(np.float64), pw)
    return loss * total, total, np.einsum('bl,blf->f', d_out * total, x)


def _worker(rank, world, port, out_path):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
    try:
        data = SyntheticData(40, L, F, seed=3, pad_frac=0.2)
        w = np.random.default_rng(0).normal(size=F)
        assert parallel.world_and_rank() == (world, rank)
        idx, clicks = _sample(data, parallel.slot_base(rank, B), B)
        pw = np.ones((B, L))
        num, den, gnum = _shard_terms(data, w, idx, clicks, pw)
        # the step bucket: [gradient | statistics | local], only the first n_reduce floats are summed
        bucket = torch.tensor(np.concatenate([gnum, [num, den], [float(rank + 1)]]), dtype=torch.float64)
        parallel.all_reduce_bucket(bucket, F + 2, world)
        gathered = [None] * world
        torch.distributed.all_gather_object(gathered, (idx.tolist(), clicks.tolist()))
        if rank == 0:
            json.dump({'bucket': bucket.tolist(), 'shards': gathered, 'local_loss': num / den}, open(out_path, 'w'))
        else:
            assert bucket[-1].item() == rank + 1  # the local tail is not reduced
    finally:
        torch.distributed.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_ranks_equal_one_big_batch(tmp_path):
    out_path = str(tmp_path / 'dp.json')
    mp.spawn(_worker, args=(2, _free_port(), out_path), nprocs=2, join=True)
    got = json.load(open(out_path))
    data = SyntheticData(40, L, F, seed=3, pad_frac=0.2)
    w = np.random.default_rng(0).normal(size=F)
    # (1) the two shards are the two halves of the single-rank 2B batch, bit for bit
    idx, clicks = _sample(data, 0, 2 * B)
    assert got['shards'][0][0] + got['shards'][1][0] == idx.tolist()
    assert got['shards'][0][1] + got['shards'][1][1] == clicks.tolist()
    # (2) sum-then-divide equals the full batch; the bucket tail (local values) stayed rank-local
    num, den, gnum = _shard_terms(data, w, idx, clicks, np.ones((2 * B, L)))
    bucket = np.array(got['bucket'])
    np.testing.assert_allclose(bucket[F] / bucket[F + 1], num / den, rtol=1e-12)
    np.testing.assert_allclose(bucket[:F] / bucket[F + 1], gnum / den, rtol=1e-10, atol=1e-14)
    assert bucket[-1] == 1.0
    # ... and differs from averaging normalised per-rank losses (why the kernels emit un-normalised terms)
    assert abs(got['local_loss'] - num / den) > 1e-9


def _sync_worker(rank, world, port, out_path):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from types import SimpleNamespace
        from ultra_amd.learning_algorithm.base_algorithm import BaseAlgorithm, HostVariable

        class Replica(object):  # the attributes sync_replicas touches, on CPU tensors (the kernels need a GPU)
            sync_replicas = BaseAlgorithm.sync_replicas
            _replica_state = BaseAlgorithm._replica_state

            def _extra_state(self):
                return {'propensity': self.propensity}
        r = Replica()
        g = torch.Generator().manual_seed(100 + rank)  # replicas start APART (per-rank seeds)
        r.world, r.rank, r.device = world, rank, torch.device('cpu')
        r.model = SimpleNamespace(params=torch.randn(50, generator=g), _wt_state=('stale',))
        r.acc = torch.rand(50, generator=g)
        r.propensity = torch.rand(7, generator=g)
        r.global_step = HostVariable(10 * (rank + 1))
        r.sync_replicas()
        assert r.model._wt_state is None and r._replicas_synced
        state = (r.model.params.tolist(), r.acc.tolist(), r.propensity.tolist(), r.global_step.value)
        gathered = [None] * world
        torch.distributed.all_gather_object(gathered, state)
        if rank == 0:
            json.dump(gathered, open(out_path, 'w'))
    finally:
        torch.distributed.destroy_process_group()


@pytest.mark.timeout(120)
def test_replicas_take_rank_zeros_variables(tmp_path):
    out_path = str(tmp_path / 'sync.json')
    mp.spawn(_sync_worker, args=(2, _free_port(), out_path), nprocs=2, join=True)
    a, b = json.load(open(out_path))
    assert a == b
    g = torch.Generator().manual_seed(100)
    assert a[0] == torch.randn(50, generator=g).tolist() and a[3] == 10


def test_single_process_is_a_no_op():
    b = torch.arange(6, dtype=torch.float32)
    assert parallel.world_and_rank() == (1, 0)
    assert torch.equal(parallel.all_reduce_bucket(b.clone(), 4), b)
    assert parallel.slot_base(3, 256) == 768
