"""Synthetic frame generators for the bench and the frame stream.

S1 is the reference's benchmark gradient (nu_scaler_core/src/benchmark.rs:188-207),
frame k of a stream being S1 with the column index rotated by k (1 px/frame motion).
These are product-side data generators (numpy / torch); the oracle has its own.
"""
from __future__ import annotations

import numpy as np


def gradient_frame(width: int, height: int, shift: int = 0) -> np.ndarray:
    """(h, w, 4) uint8: r = xs*255/w, g = y*255/h, b = (xs+y)*255/(w+h), a = 255, integer division."""
    xs = (np.arange(width, dtype=np.int64) + int(shift)) % width
    ys = np.arange(height, dtype=np.int64)
    img = np.empty((height, width, 4), dtype=np.uint8)
    img[..., 0] = (xs * 255 // width)[None, :]
    img[..., 1] = (ys * 255 // height)[:, None]
    img[..., 2] = (xs[None, :] + ys[:, None]) * 255 // (width + height)
    img[..., 3] = 255
    return img


def _splitmix64(n_words: int, seed: int) -> np.ndarray:
    idx = np.arange(1, n_words + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def noise_frame(width: int, height: int, seed: int = 0x5EED) -> np.ndarray:
    """S3: uniform u8 RGBA from splitmix64(seed), 8 little-endian bytes per draw."""
    n = width * height * 4
    words = _splitmix64((n + 7) // 8, seed)
    return words.view(np.uint8)[:n].reshape(height, width, 4).copy()


def box_frame(width: int, height: int, rgba=(255, 0, 0, 255)) -> np.ndarray:
    """S4: centred half-size box (nu_scaler_py/test_interpolator.py:23-34)."""
    img = np.zeros((height, width, 4), dtype=np.uint8)
    ch, cw, hh, hw = height // 2, width // 2, height // 4, width // 4
    img[ch - hh:ch + hh, cw - hw:cw + hw] = rgba
    return img


def gradient_stream_torch(n_frames: int, width: int, height: int, device, first: int = 0):
    """(n, h, w, 4) uint8 torch tensor on `device`: frames first .. first+n-1 of the S1 stream."""
    import torch

    k = torch.arange(first, first + n_frames, device=device, dtype=torch.int64)[:, None]
    xs = (torch.arange(width, device=device, dtype=torch.int64)[None, :] + k) % width  # (n, w)
    ys = torch.arange(height, device=device, dtype=torch.int64)  # (h,)
    out = torch.empty((n_frames, height, width, 4), dtype=torch.uint8, device=device)
    out[..., 0] = (xs * 255 // width)[:, None, :].to(torch.uint8)
    out[..., 1] = (ys * 255 // height)[None, :, None].to(torch.uint8)
    out[..., 2] = ((xs[:, None, :] + ys[None, :, None]) * 255 // (width + height)).to(torch.uint8)
    out[..., 3] = 255
    return out


def noise_stream_torch(n_frames: int, width: int, height: int, device, seed: int = 0x5EED):
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return torch.randint(0, 256, (n_frames, height, width, 4), dtype=torch.uint8, device=device, generator=g)
