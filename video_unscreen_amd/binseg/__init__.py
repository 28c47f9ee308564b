from .agent import SegAgent  # noqa
from .agent_human import HumanSegAgent  # noqa
