from .fgfuncs import get_bg, get_fg, is_pixel_inrange  # noqa
from .imgprocess import get_target_size, inv_pad_resize_size, pad_resize_size  # noqa
from .maskprocess import (dilate_mask, erode_mask, exist_foreground, get_outer_boundary,  # noqa
                          remove_invalid_objects)
