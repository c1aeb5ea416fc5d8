from .batchify import ranker_batchify, session_batchify, flat_examples  # noqa: F401
from .samplers import length_sorted_batches, session_length_batches, flat_indices  # noqa: F401
from .stream import PrefetchingBatchStream  # noqa: F401
from .session_stream import SyntheticSessionCorpus, WireLayout  # noqa: F401
