"""pareben_amd -- MI355X-native cross-validation hot path of parEBEN.

Public surface (mirrors the reference's R names):
  CrossValidate, BuildGrid, GetLambdaMax, AssignToFolds   (host side, R/*.R)
  Context, fit_gaussian                                    (thin wrappers over the C ABI)
"""
from .grid import BuildGrid, GetLambdaMax, AssignToFolds, summarise_cv
from .cv import CrossValidate
from ._lib import Context, fit_gaussian, ParebenError, load as load_library

__all__ = ["CrossValidate", "BuildGrid", "GetLambdaMax", "AssignToFolds", "summarise_cv",
           "Context", "fit_gaussian", "ParebenError", "load_library"]
