from video_unscreen_amd.binseg import HumanSegAgent, SegAgent  # noqa
