from .fgfuncs import composite_fgbg, fuse_fgbg, get_bg, get_fg, get_fg_naive, get_fg_with_colorremove, is_pixel_inrange  # noqa
from .imgprocess import (adaptive_resize, add_blackborder, color_correct, get_center, get_mask, get_target_size, imnormalize, inv_pad_resize,  # noqa
                         inv_pad_resize_size, pad_resize, pad_resize_size, regionfill, remove_blackborder, rescale_fg, shift_fg)
from .maskprocess import (build_score_map, dilate_mask, erode_mask, exist_foreground, get_fgbox, get_outer_boundary, get_score_map,  # noqa
                          remove_invalid_objects)
