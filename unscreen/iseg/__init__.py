from video_unscreen_amd.iseg import ISegAgent  # noqa
