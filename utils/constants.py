"""reference utils/constants.py:1-82 -- the constants the hot path and its callers read"""
import torch

CATEGORIES = [{"id": 0, "name": "object", "isthing": 1}]
IMAGE_ID_ZFILL = 12


class Colors:
    """instance colours of the notebook / draw_instances: utils/constants.py:44-57 (hex table -> BGR tuples)"""
    _HEX = ('FF1010', '10FF10', 'FFF010', '100FFF', '0018EC', 'FF3838', 'FF9D97', 'FF701F', 'FFB21D', 'CFD231', '48F90A', '92CC17',
            '3DDB86', '1A9334', '00D4BB', '2C99A8', '00C2FF', '344593', '6473FF', '0018EC', '8438FF', '520085', 'CB38FF', 'FF95C8',
            'FF37C7')

    def __init__(self):
        self.palette = [tuple(int(h[i:i + 2], 16) for i in (0, 2, 4)) for h in self._HEX]
        self.n = len(self.palette)

    def __call__(self, i, bgr=True):
        r, g, b = self.palette[int(i) % self.n]
        return (b, g, r) if bgr else (r, g, b)


colors = Colors()


def get_color(idx):
    return 255 if idx == -1 else colors(idx)


DEFAULT_DEVICE = 'cuda' if torch.cuda.is_available() else 'cpu'
DEFAULT_DETECTOR_CKPT = 'models/AnimeInstanceSegmentation/rtmdetl_e60.ckpt'
DEFAULT_DEPTHREFINE_CKPT = 'models/AnimeInstanceSegmentation/kenburns_depth_refinenet.ckpt'
DEFAULT_INPAINTNET_CKPT = 'models/AnimeInstanceSegmentation/kenburns_inpaintnet.ckpt'
DEPTH_ZOE_CKPT = 'models/AnimeInstanceSegmentation/ZoeD_M12_N.pt'
