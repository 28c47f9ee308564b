from video_unscreen_amd.colorfiltering import ColorFilteringAgent  # noqa
