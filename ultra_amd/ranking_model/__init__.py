from .base_ranking_model import BaseRankingModel, ActivationFunctions, NormalizationFunctions, Initializer
from .DNN import DNN
from .Linear import Linear
