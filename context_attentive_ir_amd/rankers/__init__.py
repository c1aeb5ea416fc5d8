from .esm import ESM
from .mtensor import MatchTensor
from .drmm import DRMM
from .duet import DUET

__all__ = ["ESM", "MatchTensor", "DRMM", "DUET"]
