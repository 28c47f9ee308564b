from video_unscreen_amd.vmatting import VMattingAgent  # noqa
