from .agent import ColorFilteringAgent  # noqa
