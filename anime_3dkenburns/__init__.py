"""Drop-in import surface of the reference package `anime_3dkenburns` (reference anime_3dkenburns/__init__.py:1):
    from anime_3dkenburns import KenBurnsPipeline, KenBurnsConfig, npyframes2video
resolves to the MI355X implementation (cartoonsegmentation_amd.kenburns on libcsm355)."""
from cartoonsegmentation_amd.kenburns import KenBurnsPipeline, KenBurnsConfig, npyframes2video  # noqa: F401
