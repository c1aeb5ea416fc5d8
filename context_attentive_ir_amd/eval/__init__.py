from .ltorank import MAP, MRR, precision_at_k, rank_candidates
from .validate import validate_official

__all__ = ["MAP", "MRR", "precision_at_k", "rank_candidates", "validate_official"]
