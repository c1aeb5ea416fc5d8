from .ltorank import MAP, MRR, precision_at_k, rank_candidates

__all__ = ["MAP", "MRR", "precision_at_k", "rank_candidates"]
