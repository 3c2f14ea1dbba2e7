"""Net builders: lower the reference's dense nets into libcsm355 layer programs."""
from .isnet import build_isnet  # noqa: F401
from .leres import build_leres  # noqa: F401
from .rtmdet import build_rtmdet, RTMDetConfig  # noqa: F401
from .refine import build_refine  # noqa: F401
from .inpaint import build_inpaint_context, build_inpaint_grid  # noqa: F401
from .disparity import build_semantics, build_disparity  # noqa: F401
from .zoedepth_head import build_zoe_head  # noqa: F401
from .dpt_beit import build_dpt_beit, DPTBeitConfig, resample_rel_table  # noqa: F401
