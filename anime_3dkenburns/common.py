"""reference module path anime_3dkenburns.common (common.py:59-248): operators by their reference names"""
from cartoonsegmentation_amd.ops import process_shift, fill_disocclusion, render_pointcloud, spatial_filter, depth_to_points  # noqa: F401
