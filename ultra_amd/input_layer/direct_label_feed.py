"""True-label feed (validation / test) from the device-resident data set.

Reference: ultra/input_layer/direct_label_feed.py -- hparams :43-47, list preparation :61-80,
get_batch :82-144, get_next_batch :146-207, get_data_by_index :209-265.  Evaluation only (off the
training hot path): batches are cut out of the resident tensors with a few device index ops.
"""
import random

import torch

from ..learning_algorithm.base_algorithm import DeviceBatch, DEVICE_BATCH_KEY
from ..utils import hparams as hp
from .base_input_feed import BaseInputFeed
from .device_dataset import DeviceDataset, LazyLists


class DirectLabelFeed(BaseInputFeed):
    def __init__(self, model, batch_size, hparam_str, session=None):
        self.hparams = hp.HParams(use_max_candidate_num=True)
        self.hparams.parse(hparam_str)
        self.start_index = 0
        self.count = 1
        self.rank_list_size = model.max_candidate_num if self.hparams.use_max_candidate_num else model.rank_list_size
        self.feature_size = model.feature_size
        self.batch_size = batch_size
        self.model = model
        self.device = model.device
        print('Create direct label feed with list size %d with feature size %d' %
              (self.rank_list_size, self.feature_size))

    def _batch(self, data_set, idx, check_validation):
        if len(data_set.initial_list[0]) < self.rank_list_size:
            raise ValueError("Input ranklist length must be no less than the required list size,"
                             " %d != %d." % (len(data_set.initial_list[0]), self.rank_list_size))
        dd = DeviceDataset.of(data_set, self.device)
        docids, labels = dd.lists_batch(idx, self.rank_list_size)
        if check_validation:  # drop lists without a positive label (:70-71)
            keep = labels.sum(dim=0) != 0
            docids, labels = docids[:, keep].contiguous(), labels[:, keep].contiguous()
        feed = {self.model.letor_features.name: dd.table,
                DEVICE_BATCH_KEY: DeviceBatch(dd.table, dd.stats, dd.n_rows, docids, labels)}
        for l in range(self.rank_list_size):
            feed[self.model.docid_inputs[l].name] = docids[l]
            feed[self.model.labels[l].name] = labels[l]
        return feed, {'input_list': LazyLists(docids.t()), 'click_list': LazyLists(labels.t())}, dd

    def get_batch(self, data_set, check_validation=False):
        length = len(data_set.initial_list)
        picks = [int(random.random() * length) for _ in range(self.batch_size)]  # :105-107
        feed, info, dd = self._batch(data_set, torch.tensor(picks, device=self.device), check_validation)
        info['rank_list_idxs'] = picks
        info['letor_features'] = dd.table
        return feed, info

    def get_next_batch(self, index, data_set, check_validation=False):
        n = min(self.batch_size, len(data_set.initial_list) - index)
        feed, info, _ = self._batch(data_set, torch.arange(index, index + n, device=self.device), check_validation)
        return feed, info

    def get_data_by_index(self, data_set, index, check_validation=False):
        feed, info, _ = self._batch(data_set, torch.tensor([index], device=self.device), check_validation)
        return feed, info
