from video_unscreen_amd.stm import STMAgent  # noqa
