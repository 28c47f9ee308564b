from video_unscreen_amd.harmonization import HarmonizationAgent  # noqa
