from .fgfuncs import composite_fgbg, fuse_fgbg, get_bg, get_fg, is_pixel_inrange  # noqa
from .imgprocess import (adaptive_resize, color_correct, get_target_size, inv_pad_resize_size, pad_resize_size, regionfill, rescale_fg,  # noqa
                         shift_fg)
from .maskprocess import (dilate_mask, erode_mask, exist_foreground, get_outer_boundary,  # noqa
                          remove_invalid_objects)
