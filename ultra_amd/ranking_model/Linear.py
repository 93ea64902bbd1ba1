"""The linear list scorer on MI355X: LayerNorm on the features, one [F, 1] matmul, bias.

Reference: ultra/ranking_model/Linear.py -- hparams :24-28 (initializer, norm, output_size), graph :52-76
(variables linear_W_0 / linear_b_0 and layer_norm_0).  It is the DNN with no hidden layer, so it runs in the same
kernels (``ultr_dnn_forward`` / ``ultr_dnn_backward`` with n_layers = 1: the gather + LayerNorm + dot-product output
pass and its backward); only the hparams and the variable names differ.
"""
from ..utils import hparams as hp
from .. import _lib
from .base_ranking_model import BaseRankingModel
from .DNN import DNN


class Linear(DNN):
    def __init__(self, hparams_str):
        DNN.__init__(self, '')
        self.hparams = hp.HParams(
            initializer='None',
            norm='layer',
            output_size=1)
        self.hparams.parse(hparams_str)
        # the shared variable/launch code reads these two; a linear scorer has neither hidden layers nor activation
        self.hparams.hidden_layer_sizes = []      # plain attributes, not parseable hyper-parameters
        self.hparams.activation_func = 'elu'
        self.initializer = BaseRankingModel.INITIALIZER_DIC.get(self.hparams.initializer)

    def _name_variables(self):
        self.model_parameters = {}
        ent = self.slices[0]
        o, ln = ent['W']
        self.model_parameters['ranking_model/linear_W_0:0'] = self.params[o:o + ln].view(ent['K'], 1)
        o, ln = ent['b']
        self.model_parameters['ranking_model/linear_b_0:0'] = self.params[o:o + ln]
        if self.desc.norm == _lib.NORM_LAYER:
            o, ln = ent['gamma']
            self.model_parameters['ranking_model/layer_norm_0/gamma:0'] = self.params[o:o + ln]
            o, ln = ent['beta']
            self.model_parameters['ranking_model/layer_norm_0/beta:0'] = self.params[o:o + ln]
