import numpy as np
import torch
from PIL import Image


def save_image(tensor, fp, **kwargs):
    """[3,H,W] / [1,H,W] / [H,W] float image in [0,1] -> 8-bit PNG (rounding as torchvision: x*255 + 0.5, clamp)."""
    t = tensor.detach().to("cpu", torch.float32)
    if t.dim() == 4:
        t = t[0]
    if t.dim() == 2:
        t = t[None]
    arr = t.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
    if arr.shape[2] == 1:
        arr = arr[..., 0]
    Image.fromarray(np.ascontiguousarray(arr)).save(fp)
