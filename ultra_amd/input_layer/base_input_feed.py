"""Plugin base class for input feeds (reference: ultra/input_layer/base_input_feed.py:23-106)."""
from abc import ABC, abstractmethod


class BaseInputFeed(ABC):
    MAX_SAMPLE_ROUND_NUM = 100  # :28 -- here: bound of the per-slot rejection loop on the device

    @staticmethod
    def preprocess_data(data_set, hparam_str, exp_settings):
        return

    @abstractmethod
    def __init__(self, model, batch_size, hparam_str, session):
        pass

    @abstractmethod
    def get_batch(self, data_set, check_validation=False):
        """Random batch for training -> (input_feed, info_map)."""
        pass

    @abstractmethod
    def get_next_batch(self, index, data_set, check_validation=False):
        """The batch starting at list ``index`` (validation / test) -> (input_feed, info_map)."""
        pass

    @abstractmethod
    def get_data_by_index(self, data_set, index, check_validation=False):
        pass
