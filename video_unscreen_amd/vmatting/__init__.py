from .agent import VMattingAgent  # noqa
