"""Drop-in import surface of the reference package `animeinsseg` (reference animeinsseg/__init__.py):
    from animeinsseg import AnimeInsSeg, AnimeInstances
resolves to the MI355X implementation in cartoonsegmentation_amd (libcsm355)."""
from cartoonsegmentation_amd.anime_instances import AnimeInstances  # noqa: F401
from cartoonsegmentation_amd.segmentation import AnimeInsSeg, VALID_REFINEMETHODS  # noqa: F401
