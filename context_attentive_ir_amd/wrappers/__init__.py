from .ranker import Ranker
from .multitask import Multitask

__all__ = ["Ranker", "Multitask"]
