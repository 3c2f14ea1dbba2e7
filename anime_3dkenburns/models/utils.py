"""reference module path anime_3dkenburns.models.utils (models/utils.py:9-315)"""
from cartoonsegmentation_amd.ops import render_pointcloud, spatial_filter, depth_to_points  # noqa: F401
