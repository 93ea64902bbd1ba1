"""Simulated-click training feed, sampled on the device.

Reference: ultra/input_layer/click_simulation_feed.py -- hparams :45-56, per-list click sampling
:75-105, get_batch :107-180 (rejection sampling of lists with >= 1 click, position-major re-index,
dynamic-eta schedule :171-178), get_next_batch :182-242, get_data_by_index :244-300.

Same constructor, hparam string, method signatures and feed-dict keys.  The interpreted per-list
loop is replaced by ``ultr_batch_assemble``: B list slots are sampled in parallel, each slot retrying
on its own Philox stream until its list has a click (distributionally the reference's i.i.d.
rejection sampling; order-free, so N data-parallel ranks x B lists equal one rank x N*B lists).
``device_rng=False`` instead consumes Python's global ``random`` stream exactly as the reference
does (1 + rank_list_size draws per candidate, rejected candidates included) and reproduces the
reference batch bit for bit.

The feed values are DEVICE tensors: ``letor_features:0`` is the whole resident feature table, doc ids
index it directly and the pad id is its row count (the reference's convention: pad = len(letor_features)).
"""
import json
import random

import numpy as np
import torch

from .. import _lib, parallel
from ..learning_algorithm.base_algorithm import DeviceBatch, DEVICE_BATCH_KEY
from ..utils import click_models as cm
from ..utils import hparams as hp
from .base_input_feed import BaseInputFeed
from .device_dataset import DeviceDataset, LazyLists


class ClickSimulationFeed(BaseInputFeed):
    def __init__(self, model, batch_size, hparam_str, session=None):
        self.hparams = hp.HParams(
            click_model_json='./example/ClickModel/pbm_0.1_1.0_4_1.0.json',
            oracle_mode=False,               # feed relevance labels instead of simulated clicks
            dynamic_bias_eta_change=0.0,     # eta increment for dynamic bias severity, 0 = off
            dynamic_bias_step_interval=1000,
            device_rng=True,                 # (new) False: consume Python's random stream, bit-exact
            seed=0,                          # (new) Philox key of the device stream
            prefetch=False)                  # (new) assemble batch t+1 during the backward pass of step t
        print('Create simluated clicks feed')
        print(hparam_str)
        self.hparams.parse(hparam_str)
        self.click_model = None
        if not self.hparams.oracle_mode:
            with open(self.hparams.click_model_json) as fin:
                self.click_model = cm.loadModelFromJson(json.load(fin))
        self.start_index = 0
        self.count = 1
        self.rank_list_size = model.rank_list_size
        self.feature_size = model.feature_size
        self.batch_size = batch_size
        self.model = model
        self.global_batch_count = 0
        self.device = model.device
        if self.rank_list_size > _lib.ULTR_MAX_LIST:
            raise ValueError('rank_list_size %d > %d' % (self.rank_list_size, _lib.ULTR_MAX_LIST))
        self.world, self.rank = getattr(model, 'world', 1), getattr(model, 'rank', 0)
        self._upload_tables()
        # The device step counter keys the Philox draws of every assembled batch.  Regression-EM keys its Bernoulli
        # labels on the same counter, so a model that has one shares it with ALL its feeds: the first feed creates it,
        # later feeds (a second training feed, a feed built for validation) advance the same counter instead of
        # replacing it -- a replaced counter would leave the model reading one that no batch advances any more.
        # Only a feed that advances the counter hands it over (the device sampler, mode 0): with device_rng=False no
        # kernel ever bumps it, and the model must key its draws on its own step count instead.
        shared = getattr(model, 'step_ctr', None)
        self.step_ctr = shared if shared is not None else torch.zeros(1, dtype=torch.int32, device=self.device)
        if hasattr(model, 'step_ctr') and shared is None and self.hparams.device_rng:
            model.step_ctr = self.step_ctr
        self.info = torch.zeros(_lib.ULTR_FEED_INFO_INTS, dtype=torch.int32, device=self.device)
        self._out = {}
        # prefetch: (data set, batch size, check_validation) the buffers were last filled / parked for
        self._prefetched = None
        self.ctx = _lib.context_of(model)  # where this feed's next-batch requests are parked

    # The library parks ONE assembly request per request context (include/ultr_hip.h, Conventions): the feed parks in
    # its model's context (``self.ctx``), where the model's backward pass or update finds it.  Several feeds may serve
    # one model, so the context remembers whose request was parked last: a feed drops only its own request, and
    # launches (never drops) one that may belong to another feed of the same model.
    def _drop_parked(self):
        lib = _lib.lib()
        if self.ctx.assemble_owner == id(self):
            lib.ultr_batch_assemble_cancel()
        else:  # another feed's next batch (or nothing at all): it runs now, on its own
            _lib.check(lib.ultr_batch_assemble_flush(_lib.current_stream()), 'ultr_batch_assemble_flush')
        self.ctx.assemble_owner = None

    def __del__(self):
        try:
            if self._prefetched is not None and self.ctx.assemble_owner == id(self):
                lib = _lib.lib()
                was = lib.ultr_pending_bind(self.ctx.handle)
                lib.ultr_batch_assemble_cancel()  # the parked request points into buffers that go away with us
                lib.ultr_pending_bind(was)
                self.ctx.assemble_owner = None
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def _upload_tables(self):
        if self.click_model is None:
            self.kind, exam, self.n_exam, cp = _lib.CLICK_ORACLE, [1.0], 1, [1.0]
        else:
            self.kind, exam, self.n_exam, cp = self.click_model.device_tables(self.rank_list_size)
        self.exam = torch.tensor(exam, dtype=torch.float64, device=self.device)
        self.click_prob = torch.tensor(cp, dtype=torch.float64, device=self.device)

    def _buffers(self, B):
        o = self._out.get(B)
        if o is None:
            L = self.rank_list_size
            o = {'docids': torch.empty((L, B), dtype=torch.int32, device=self.device),
                 'clicks': torch.empty((L, B), dtype=torch.float32, device=self.device),
                 'idx': torch.empty(B, dtype=torch.int32, device=self.device)}
            self._out[B] = o
        return o

    def _check_lists(self, data_set):
        if len(data_set.initial_list[0]) < self.rank_list_size:
            raise ValueError("Input ranklist length must be no less than the required list size,"
                             " %d != %d." % (len(data_set.initial_list[0]), self.rank_list_size))

    def _assemble(self, dd, o, B, mode, uniforms=None, n_cand=0, slot_start=0, check_validation=True):
        _lib.check(_lib.lib().ultr_batch_assemble(
            _lib.ptr(dd.initial_list), _lib.ptr(dd.labels), dd.Q, dd.Lmax, self.rank_list_size, B, self.kind,
            _lib.ptr(self.exam), self.n_exam, _lib.ptr(self.click_prob), self.click_prob.numel(), mode,
            int(self.hparams.seed) & 0xFFFFFFFFFFFFFFFF, 0, _lib.ptr(self.step_ctr) if mode == 0 else None,
            parallel.slot_base(self.rank, B), _lib.ptr(uniforms), n_cand, slot_start, 1 if check_validation else 0,
            self.MAX_SAMPLE_ROUND_NUM, dd.n_rows, _lib.ptr(o['docids']), _lib.ptr(o['clicks']), _lib.ptr(o['idx']),
            _lib.ptr(self.info), _lib.current_stream()), 'ultr_batch_assemble')

    def _assemble_deferred(self, dd, o, B, check_validation):
        """Park the assembly of the NEXT batch: the backward pass of the coming step carries it inside its last
        launch (``ultr_batch_assemble_deferred``); if no backward pass comes first, the next call launches it."""
        _lib.check(_lib.lib().ultr_batch_assemble_deferred(
            _lib.ptr(dd.initial_list), _lib.ptr(dd.labels), dd.Q, dd.Lmax, self.rank_list_size, B, self.kind,
            _lib.ptr(self.exam), self.n_exam, _lib.ptr(self.click_prob), self.click_prob.numel(),
            int(self.hparams.seed) & 0xFFFFFFFFFFFFFFFF, 0, _lib.ptr(self.step_ctr), parallel.slot_base(self.rank, B),
            1 if check_validation else 0, self.MAX_SAMPLE_ROUND_NUM, dd.n_rows, _lib.ptr(o['docids']),
            _lib.ptr(o['clicks']), _lib.ptr(o['idx']), _lib.ptr(self.info), _lib.current_stream()),
            'ultr_batch_assemble_deferred')

    def _sample_python_stream(self, dd, o, B, check_validation):
        """Consume random.random() exactly like click_simulation_feed.py:131-137 (q2): per candidate one
        draw for the list index, then rank_list_size click draws (none in oracle mode)."""
        L = self.rank_list_size
        per = 1 if self.click_model is None else 1 + L
        state = random.getstate()
        rs = np.random.RandomState()
        rs.set_state(('MT19937', np.array(state[1][:624], dtype=np.uint32), state[1][624]))
        consumed_total, filled = 0, 0
        while filled < B:
            n_cand = max(1024, 4 * (B - filled))
            draws = rs.random_sample(n_cand * per).reshape(n_cand, per)  # same doubles as random.random()
            u = np.zeros((n_cand, 1 + L), dtype=np.float64)
            u[:, :per] = draws
            self._assemble(dd, o, B, 1, torch.from_numpy(u).to(self.device), n_cand, filled, check_validation)
            consumed, filled = self.info[:2].cpu().tolist()
            consumed_total += consumed
            if filled < B:  # everything was consumed; continue with fresh draws
                continue
            # rewind to the exact number of draws the reference would have made
            rs2 = np.random.RandomState()
            rs2.set_state(('MT19937', np.array(state[1][:624], dtype=np.uint32), state[1][624]))
            rs2.random_sample(consumed_total * per)
            st = rs2.get_state()
            random.setstate((state[0], tuple(int(x) for x in st[1]) + (int(st[2]),), state[2]))
        return consumed_total

    def _feed(self, dd, docids, clicks):
        input_feed = {self.model.letor_features.name: dd.table,
                      DEVICE_BATCH_KEY: DeviceBatch(dd.table, dd.stats, dd.n_rows, docids, clicks)}
        for l in range(self.rank_list_size):
            input_feed[self.model.docid_inputs[l].name] = docids[l]
            input_feed[self.model.labels[l].name] = clicks[l]
        return input_feed

    def get_batch(self, data_set, check_validation=False):
        self._check_lists(data_set)
        dd = DeviceDataset.of(data_set, self.device)
        B = self.batch_size
        o = self._buffers(B)
        if self.hparams.device_rng and self.hparams.prefetch and self.hparams.dynamic_bias_eta_change == 0:
            # Prefetching feed: the batch returned now was assembled during the previous step's backward pass, into
            # these same buffers (nothing reads a batch after the weight gradients of its step); this call only
            # parks the request for the batch after it.  Host-side views of a batch (input_list, click_list, the
            # per-position feed entries) must therefore be read before the step that follows this call.
            key = (id(dd), B, bool(check_validation))
            self.ctx.bind()
            if self._prefetched != key:
                self._drop_parked()  # a request this feed parked for other arguments is dropped
                self._assemble(dd, o, B, 0, check_validation=check_validation)
            self._assemble_deferred(dd, o, B, check_validation)
            self.ctx.assemble_owner = id(self)
            self._prefetched = key
        elif self.hparams.device_rng:
            self._assemble(dd, o, B, 0, check_validation=check_validation)
        else:
            self._sample_python_stream(dd, o, B, check_validation)
        input_feed = self._feed(dd, o['docids'], o['clicks'])
        info_map = {'rank_list_idxs': LazyLists(o['idx'].view(-1, 1)),
                    'input_list': LazyLists(o['docids'].t()),
                    'click_list': LazyLists(o['clicks'].t()),
                    'letor_features': dd.table}
        self.global_batch_count += 1
        if self.hparams.dynamic_bias_eta_change != 0 and not self.hparams.oracle_mode:
            if self.global_batch_count % self.hparams.dynamic_bias_step_interval == 0:
                self.click_model.eta += self.hparams.dynamic_bias_eta_change
                self.click_model.setExamProb(self.click_model.eta)  # rebuilt from the built-in table (q5)
                self.exam.copy_(torch.tensor(self.click_model.device_tables(self.rank_list_size)[1], dtype=torch.float64))
                print('Dynamically change bias severity eta to %.3f' % self.click_model.eta)
        return input_feed, info_map

    # -- ordered batches (validation): clicks for lists index..index+batch_size ---------------------
    def _ordered(self, dd, idx, check_validation):
        L = self.rank_list_size
        docids, labs = dd.lists_batch(idx, L)
        if self.click_model is None:
            clicks = labs
        else:
            n = idx.numel()
            lab_nl = labs.t().contiguous()
            out = torch.empty_like(lab_nl)
            u = torch.tensor([[random.random() for _ in range(L)] for _ in range(n)], dtype=torch.float64,
                             device=self.device)
            _lib.check(_lib.lib().ultr_click_sample(
                self.kind, _lib.ptr(lab_nl), n, L, _lib.ptr(self.exam), self.n_exam, _lib.ptr(self.click_prob),
                self.click_prob.numel(), _lib.ptr(u), 0, 0, _lib.ptr(out), _lib.current_stream()), 'ultr_click_sample')
            clicks = out.t().contiguous()
        if check_validation:
            keep = clicks.sum(dim=0) != 0
            docids, clicks = docids[:, keep].contiguous(), clicks[:, keep].contiguous()
        return docids, clicks

    def get_next_batch(self, index, data_set, check_validation=False):
        self._check_lists(data_set)
        dd = DeviceDataset.of(data_set, self.device)
        n = min(self.batch_size, dd.Q - index)
        idx = torch.arange(index, index + n, device=self.device)
        docids, clicks = self._ordered(dd, idx, check_validation)
        return self._feed(dd, docids, clicks), {'input_list': LazyLists(docids.t()), 'click_list': LazyLists(clicks.t())}

    def get_data_by_index(self, data_set, index, check_validation=False):
        self._check_lists(data_set)
        dd = DeviceDataset.of(data_set, self.device)
        docids, clicks = self._ordered(dd, torch.tensor([index], device=self.device), check_validation)
        return self._feed(dd, docids, clicks), {'input_list': LazyLists(docids.t()), 'click_list': LazyLists(clicks.t())}
