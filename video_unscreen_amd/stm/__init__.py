from .agent import STMAgent  # noqa
