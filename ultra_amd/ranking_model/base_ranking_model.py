"""Plugin base class for ranking models (reference: ultra/ranking_model/base_ranking_model.py).

The reference maps hparam strings to TF callables (ACT_FUNC_DIC :59-65, NORM_FUNC_DIC :67-70,
INITIALIZER_DIC :72-74); here the same keys select enum values of the HIP kernels.
"""
from abc import ABC, abstractmethod

from .. import _lib


class ActivationFunctions(object):
    ELU = 'elu'
    RELU = 'relu'
    SELU = 'selu'
    TANH = 'tanh'
    SIGMOID = 'sigmoid'


class NormalizationFunctions(object):
    BATCH = 'batch'
    LAYER = 'layer'


class Initializer(object):
    CONSTANT = 'constant'


class BaseRankingModel(ABC):
    # hparam value -> kernel enum (ULTR_ACT_*); any other string leaves act_func = None as in the reference
    ACT_FUNC_DIC = dict(_lib.ACT)
    # 'layer' -> fused LayerNorm in the GEMM prologue; 'batch' is recognised but not on the device path
    NORM_FUNC_DIC = {NormalizationFunctions.BATCH: 'batch', NormalizationFunctions.LAYER: 'layer'}
    # tf.constant_initializer(0.001) (:73)
    INITIALIZER_DIC = {Initializer.CONSTANT: 0.001}

    model_parameters = {}

    @abstractmethod
    def __init__(self, hparams_str=None, **kwargs):
        pass

    @abstractmethod
    def build(self, input_list, noisy_params=None, noise_rate=0.05, is_training=False, **kwargs):
        """input_list: list of [B, F] device tensors (one per list position) -> list of [B, 1] scores."""
        pass
