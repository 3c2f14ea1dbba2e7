"""reference utils/effects.py:143-181 -- bokeh_blur on the MI355X (kernel_bokeh restated in imageops.hip)"""
import numpy as np

from cartoonsegmentation_amd import ops


def bokeh_blur(img, depth, num_samples=32, lightness_factor=10, depth_factor=2, use_cuda=False, focal_plane=None):
    """same signature and result type as the reference: numpy uint8 in -> numpy uint8 out (device tensors stay on the device)"""
    out = ops.bokeh_blur(img, depth, num_samples, lightness_factor, depth_factor, use_cuda, focal_plane)
    return out.cpu().numpy() if isinstance(img, np.ndarray) else out
