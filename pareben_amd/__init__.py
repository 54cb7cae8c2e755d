"""pareben_amd -- MI355X-native cross-validation hot path of parEBEN.

Public surface (mirrors the reference's R names):
  CrossValidate, LocalSearch, BuildGrid, GetLambdaMax, AssignToFolds   (host side, R/*.R)
  EBelasticNet.Gaussian / EBelasticNet.Binomial           (refit at the optimum, EBEN_orig/R/*.R)
  Context, fit_gaussian, fit_binomial                      (thin wrappers over the C ABI)
"""
from .grid import BuildGrid, GetLambdaMax, AssignToFolds, summarise_cv
from .cv import CrossValidate
from .local import LocalSearch
from ._lib import Context, cv_grid_multi, multi_last_stats, fit_gaussian, fit_binomial, ParebenError, load as load_library
from .eben import EBelasticNet, pt

__all__ = ["CrossValidate", "LocalSearch", "BuildGrid", "GetLambdaMax", "AssignToFolds", "summarise_cv",
           "Context", "cv_grid_multi", "multi_last_stats", "fit_gaussian", "fit_binomial", "EBelasticNet", "pt", "ParebenError", "load_library"]
