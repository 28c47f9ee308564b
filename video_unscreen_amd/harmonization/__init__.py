from .agent import HarmonizationAgent  # noqa
