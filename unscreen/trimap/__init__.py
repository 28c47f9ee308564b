from video_unscreen_amd.trimap import TrimapAgent  # noqa
