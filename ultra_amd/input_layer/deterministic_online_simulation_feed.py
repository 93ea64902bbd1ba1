"""Online simulation feed: lists are re-ranked by the CURRENT model and clicked on the device.

Reference: ultra/input_layer/deterministic_online_simulation_feed.py -- hparams :46-57, true-label list
preparation :86-104, simulate_clicks_online :106-194 (score with ``model.output``, re-rank every list by score,
sample clicks on the top ``rank_list_size`` of the result, re-sampling while a list has no click), get_batch
:196-270, get_next_batch :272-337, get_data_by_index :339-400.

Same constructor, hparam string, method signatures and feed-dict keys.  The reference scores the batch with a
``session.run`` round trip and re-ranks in an interpreted per-list loop; here the scores come from the same DNN
kernels the training step uses (``BaseAlgorithm.ranking_model`` over ``max_candidate_num`` positions) and
``ultr_online_rerank_click`` does the rest in one launch, so nothing leaves the device.  List indices are drawn
from Python's ``random`` exactly like the reference (one draw per slot, :219-221; lists without a relevant
document are dropped, not re-drawn, :94-96).  Click draws come from the device Philox stream; with
``device_rng=False`` they are taken from Python's ``random`` in the reference's order instead (the re-ranked
labels make one trip to the host: a compatibility mode, as slow as the reference).

Result interleaving (``need_interleave`` algorithms, :77-84, 141-154) is not on the device path.
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


class DeterministicOnlineSimulationFeed(BaseInputFeed):
    BANNER = 'Create online deterministic simluation feed'
    STOCHASTIC = False

    def _default_hparams(self):
        return hp.HParams(
            click_model_json='./example/ClickModel/pbm_0.1_1.0_4_1.0.json',
            oracle_mode=False,               # feed relevance labels instead of simulated clicks
            dynamic_bias_eta_change=0.0,     # eta increment for dynamic bias severity, 0 = off
            dynamic_bias_step_interval=1000,
            device_rng=True,                 # (new) False: click draws from Python's random stream, bit-exact
            seed=0)                          # (new) Philox key of the device stream

    def __init__(self, model, batch_size, hparam_str, session=None):
        self.hparams = self._default_hparams()
        print(self.BANNER)
        print(hparam_str)
        self.hparams.parse(hparam_str)
        with open(self.hparams.click_model_json) as fin:
            self.click_model = cm.loadModelFromJson(json.load(fin))
        self.start_index = 0
        self.count = 1
        self.rank_list_size = model.rank_list_size
        self.max_candidate_num = model.max_candidate_num
        self.feature_size = model.feature_size
        self.batch_size = batch_size
        self.model = model
        self.session = session
        self.global_batch_count = 0
        self.device = model.device
        self.need_interleave = bool(getattr(model.hparams, 'need_interleave', False))
        if self.need_interleave:
            raise NotImplementedError('result interleaving (need_interleave) is not on the MI355X path')
        if self.rank_list_size > _lib.ULTR_MAX_LIST:
            raise ValueError('rank_list_size %d > %d' % (self.rank_list_size, _lib.ULTR_MAX_LIST))
        self.world, self.rank = getattr(model, 'world', 1), getattr(model, 'rank', 0)
        self._upload_tables()
        self.draw_step = 0

    def _upload_tables(self):
        if self.hparams.oracle_mode:
            self.kind, exam, self.n_exam, cp = _lib.CLICK_ORACLE, [1.0], 1, [1.0]
        else:
            self.kind, exam, self.n_exam, cp = self.click_model.device_tables(self.rank_list_size)
        self.exam = torch.tensor(exam, dtype=torch.float64, device=self.device)
        self.click_prob = torch.tensor(cp, dtype=torch.float64, device=self.device)

    def _check_lists(self, data_set):
        if len(data_set.initial_list[0]) < self.rank_list_size:
            raise ValueError("Input ranklist length must be no less than the required list size,"
                             " %d != %d." % (len(data_set.initial_list[0]), self.rank_list_size))

    @staticmethod
    def _has_relevant(data_set, Lmax):
        """Per list: does prepare_true_labels_with_index keep it under check_validation (:90-96)?"""
        cached = getattr(data_set, '_ultra_amd_relevant', None)
        if cached is None or len(cached) != len(data_set.initial_list):
            cached = [sum(0 if ids[x] < 0 else labs[x] for x in range(min(Lmax, len(ids)))) != 0
                      for ids, labs in zip(data_set.initial_list, data_set.labels)]
            data_set._ultra_amd_relevant = cached
        return cached

    def _python_stream_clicks(self, relabels, check_validation):
        """device_rng=False: sampleClicksForOneList per list, in list order, on Python's random (:173-180)."""
        L = self.rank_list_size
        lab = relabels.t().cpu().numpy()                                       # [B][Lmax] re-ranked true labels
        lens = self._lens.cpu().numpy()
        out = np.zeros_like(lab)
        for i in range(lab.shape[0]):
            shown = [float(x) for x in lab[i, :min(L, int(lens[i]))]]
            click_list, _, _ = self.click_model.sampleClicksForOneList(shown)
            sample_count = 0
            while check_validation and sum(click_list) == 0 and sample_count < self.MAX_SAMPLE_ROUND_NUM:
                click_list, _, _ = self.click_model.sampleClicksForOneList(shown)
                sample_count += 1
            out[i, :len(click_list)] = click_list
        return torch.from_numpy(out.T.copy()).to(self.device)

    def simulate_clicks_online(self, dd, docids, labels, check_validation=False):
        """:106-194 on device tensors [Lmax][B]; returns the re-ranked doc ids and the clicks."""
        Lmax, B = docids.shape
        batch = DeviceBatch(dd.table, dd.stats, dd.n_rows, docids, labels)
        scores = self.model.ranking_model(Lmax, batch)                         # the current model, eval mode (:119-120)
        new_docids, clicks = torch.empty_like(docids), torch.empty_like(labels)
        python_stream = not self.hparams.device_rng and not self.hparams.oracle_mode
        # compatibility mode: the kernel only re-ranks (oracle mode copies the re-ranked labels), clicks follow on the host
        kind = _lib.CLICK_ORACLE if python_stream else self.kind
        args = (_lib.ptr(docids), _lib.ptr(labels), _lib.ptr(scores), B, Lmax, self.rank_list_size, dd.n_rows, kind,
                _lib.ptr(self.exam), self.n_exam, _lib.ptr(self.click_prob), self.click_prob.numel(), None,
                int(self.hparams.seed) & 0xFFFFFFFFFFFFFFFF, self.draw_step, None, parallel.slot_base(self.rank, B),
                1 if check_validation else 0, self.MAX_SAMPLE_ROUND_NUM, _lib.ptr(new_docids), _lib.ptr(clicks), None,
                _lib.current_stream())
        if self.STOCHASTIC:  # the ranking itself is a Plackett-Luce draw from the scores
            _lib.check(_lib.lib().ultr_online_sample_click(float(self.hparams.tau), None, *args), 'ultr_online_sample_click')
        else:
            _lib.check(_lib.lib().ultr_online_rerank_click(*args), 'ultr_online_rerank_click')
        self.draw_step += 1
        if python_stream:
            L = self.rank_list_size
            pos = torch.arange(1, Lmax + 1, device=self.device, dtype=torch.int32)[:, None]
            self._lens = ((docids < dd.n_rows).to(torch.int32) * pos).max(dim=0).values  # 1 + last real document
            # ranks >= L of the kernel's output are already 0; the first L rows hold the re-ranked true labels
            full = clicks.clone()
            full[:L] = self._python_stream_clicks(clicks[:L], check_validation)[:L]
            clicks = full
        return new_docids, clicks

    def _feed(self, dd, docids, clicks):
        input_feed = {self.model.letor_features.name: dd.table,
                      DEVICE_BATCH_KEY: DeviceBatch(dd.table, dd.stats, dd.n_rows, docids, clicks)}
        for l in range(self.max_candidate_num):
            input_feed[self.model.docid_inputs[l].name] = docids[l]
            input_feed[self.model.labels[l].name] = clicks[l]
        return input_feed

    def _batch(self, data_set, picks, check_validation):
        self._check_lists(data_set)
        dd = DeviceDataset.of(data_set, self.device)
        if check_validation:  # lists without a relevant document are dropped, the batch gets smaller (:94-96)
            ok = self._has_relevant(data_set, self.max_candidate_num)
            picks = [i for i in picks if ok[i]]
        idx = torch.tensor(picks, dtype=torch.int64, device=self.device)
        docids, labels = dd.lists_batch(idx, self.max_candidate_num)
        if len(picks):
            new_docids, clicks = self.simulate_clicks_online(dd, docids, labels, check_validation)
        else:
            new_docids, clicks = docids, labels
        # info_map holds the lists as they were BEFORE re-ranking, and the true labels under 'click_list' (:254-259)
        info_map = {'input_list': LazyLists(docids.t()), 'click_list': LazyLists(labels.t())}
        return self._feed(dd, new_docids, clicks), info_map, dd

    def get_batch(self, data_set, check_validation=False):
        length = len(data_set.initial_list)
        picks = [int(random.random() * length) for _ in range(self.batch_size)]  # :219-221
        input_feed, info_map, dd = self._batch(data_set, picks, check_validation)
        info_map['rank_list_idxs'] = picks
        info_map['letor_features'] = dd.table
        self.global_batch_count += 1
        if self.hparams.dynamic_bias_eta_change != 0:
            if self.global_batch_count % self.hparams.dynamic_bias_step_interval == 0:
                self.click_model.eta += self.hparams.dynamic_bias_eta_change
                self.click_model.setExamProb(self.click_model.eta)
                if not self.hparams.oracle_mode:
                    self.exam.copy_(torch.tensor(self.click_model.device_tables(self.rank_list_size)[1],
                                                 dtype=torch.float64))
                print('Dynamically change bias severity eta to %.3f' % self.click_model.eta)
        return input_feed, info_map

    def get_next_batch(self, index, data_set, check_validation=False):
        n = min(self.batch_size, len(data_set.initial_list) - index)
        input_feed, info_map, _ = self._batch(data_set, list(range(index, index + n)), check_validation)
        return input_feed, info_map

    def get_data_by_index(self, data_set, index, check_validation=False):
        input_feed, info_map, _ = self._batch(data_set, [index], check_validation)
        return input_feed, info_map
