"""oracle/postprocess.py -- CPU restatement of the reference's image post-processing.

TEST INFRASTRUCTURE ONLY.  Nothing under portal_amd/ may import this.

average_images follows src/main.rs:645-722 exactly: S_TO_L[c] = c*c (u16), per-channel u32 sums over
the images, integer division by the image count, L_TO_S[l] = ((l as f32).sqrt() + 0.5) as u8, alpha 255.
"""
import numpy as np

S_TO_L = (np.arange(256, dtype=np.uint32) ** 2).astype(np.uint16)
L_TO_S = (np.sqrt(np.arange(65026, dtype=np.float32)) + np.float32(0.5)).astype(np.uint8)  # f32 sqrt, f32 add, truncation


def average_images(images):
    """images: list of (H, W, 4) uint8 arrays -> (H, W, 4) uint8."""
    if len(images) == 1:
        return images[0].copy()  # the reference returns the single image untouched (incl. its alpha)
    acc = np.zeros(images[0].shape[:2] + (3,), np.uint32)
    for im in images:
        assert im.shape == images[0].shape
        acc += S_TO_L[im[..., :3]].astype(np.uint32)
    mean = acc // np.uint32(len(images))
    out = np.empty_like(images[0])
    out[..., :3] = L_TO_S[mean]
    out[..., 3] = 255
    return out
