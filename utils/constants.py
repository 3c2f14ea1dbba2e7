"""reference utils/constants.py:1-82 -- the constants the hot path and its callers read"""
import torch

CATEGORIES = [{"id": 0, "name": "object", "isthing": 1}]
IMAGE_ID_ZFILL = 12


from cartoonsegmentation_amd.anime_instances import Colors, colors, get_color  # noqa: E402,F401  (utils/constants.py:44-63)

DEFAULT_DEVICE = 'cuda' if torch.cuda.is_available() else 'cpu'
DEFAULT_DETECTOR_CKPT = 'models/AnimeInstanceSegmentation/rtmdetl_e60.ckpt'
DEFAULT_DEPTHREFINE_CKPT = 'models/AnimeInstanceSegmentation/kenburns_depth_refinenet.ckpt'
DEFAULT_INPAINTNET_CKPT = 'models/AnimeInstanceSegmentation/kenburns_inpaintnet.ckpt'
DEPTH_ZOE_CKPT = 'models/AnimeInstanceSegmentation/ZoeD_M12_N.pt'
