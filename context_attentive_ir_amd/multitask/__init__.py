from .cars import CARS
from .mmtensor import M_MATCH_TENSOR

__all__ = ["CARS", "M_MATCH_TENSOR"]
