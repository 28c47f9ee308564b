from .agent_human import HumanSegAgent  # noqa
