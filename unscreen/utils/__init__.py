from video_unscreen_amd.utils import *  # noqa
from video_unscreen_amd.utils.fileio import *  # noqa
