from .batchify import ranker_batchify, session_batchify, flat_examples  # noqa: F401
