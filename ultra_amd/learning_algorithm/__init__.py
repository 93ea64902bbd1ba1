from .base_algorithm import BaseAlgorithm
from .dla import DLA
from .ipw_rank import IPWrank
from .regression_EM import RegressionEM
from .pairwise_debias import PairDebias
from .navie_algorithm import NavieAlgorithm, NaiveAlgorithm
