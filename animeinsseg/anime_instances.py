"""reference module path animeinsseg.anime_instances (run_segmentation.ipynb cell 0 imports get_color from here)"""
from cartoonsegmentation_amd.anime_instances import AnimeInstances, get_color  # noqa: F401
