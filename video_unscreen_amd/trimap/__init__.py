from .agent import TrimapAgent  # noqa
