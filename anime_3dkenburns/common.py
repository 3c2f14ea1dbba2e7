"""reference module path anime_3dkenburns.common (common.py:59-248): operators by their reference names"""
from cartoonsegmentation_amd.ops import (process_shift, process_autozoom, fill_disocclusion, render_pointcloud,  # noqa: F401
                                         spatial_filter, depth_to_points)
