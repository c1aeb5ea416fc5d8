from .ranker import Ranker
from .multitask import Multitask
from .common import GraphedUpdate

__all__ = ["Ranker", "Multitask", "GraphedUpdate"]
