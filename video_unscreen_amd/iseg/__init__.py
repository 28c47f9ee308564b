from .agent import ISegAgent  # noqa
