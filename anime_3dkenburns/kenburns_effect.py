"""reference module path anime_3dkenburns.kenburns_effect"""
from cartoonsegmentation_amd.kenburns import (KenBurnsPipeline, KenBurnsConfig, npyframes2video, build_kenburns_cfg,  # noqa: F401
                                              depth_adjustment_animesseg)
