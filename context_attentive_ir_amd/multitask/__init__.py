from .cars import CARS
from .mmtensor import M_MATCH_TENSOR
from .mnsrf import MNSRF

__all__ = ["CARS", "M_MATCH_TENSOR", "MNSRF"]
