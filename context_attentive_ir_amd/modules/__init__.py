from .embeddings import Embeddings
from .maxout import Maxout

__all__ = ["Embeddings", "Maxout"]
