"""trainer.py imports DataManager but never uses it (its call is commented out, trainer.py:44-45): import-only stub."""


class DataManager:  # pragma: no cover
    def __init__(self, *a, **k):
        raise NotImplementedError("classification data managers are outside the retrieval hot path (SURVEY.md section 2a)")
