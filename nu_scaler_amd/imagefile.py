"""Image-file front end of the upscaler: `upscale_image_file` as in the legacy crate
(Nu_scale/src/upscale/mod.rs:307-338 and :341-356) plus the matching interpolation helper.

The reference decodes with the `image` crate and converts to RGBA8 (`image::open(..)?.to_rgba8()`);
here a small PNG codec (8-bit, non-interlaced, colour types 0/2/3/4/6) on zlib + numpy does that,
because nothing else in the image can read PNGs.  Pixels go through the HIP path only
(`PyWgpuUpscaler` / `WgpuFrameInterpolator`): no CPU resampling here.
"""
from __future__ import annotations

import struct
import zlib

from .interpolator import WgpuFrameInterpolator
from .upscaler import PyWgpuUpscaler

_SIG = b"\x89PNG\r\n\x1a\n"


def _np():
    import numpy as np
    return np


def _unfilter(raw: bytes, w: int, h: int, bpp: int):
    """Undo the five PNG row filters (PNG spec section 9).  bpp = bytes per complete pixel."""
    np = _np()
    stride = w * bpp
    if len(raw) < h * (stride + 1):
        raise ValueError("PNG: truncated image data")
    rows = np.frombuffer(raw, np.uint8, h * (stride + 1)).reshape(h, stride + 1)
    out = np.empty((h, stride), np.uint8)
    zero = np.zeros(stride, np.uint8)
    for y in range(h):
        ft = int(rows[y, 0])
        line = rows[y, 1:]
        prev = out[y - 1] if y else zero
        if ft == 0:
            out[y] = line
        elif ft == 1:  # Sub: running sum per channel, modulo 256
            out[y] = np.cumsum(line.reshape(w, bpp), axis=0, dtype=np.uint8).reshape(stride)
        elif ft == 2:  # Up
            out[y] = line + prev
        elif ft in (3, 4):  # Average / Paeth: sequential along the row, vectorised over the pixel's bytes
            cur = np.zeros((w + 1, bpp), np.int32)  # cur[0] = the zero pixel left of the row
            up = np.zeros((w + 1, bpp), np.int32)
            up[1:] = prev.reshape(w, bpp)
            ln = line.reshape(w, bpp).astype(np.int32)
            if ft == 3:
                for x in range(w):
                    cur[x + 1] = (ln[x] + ((cur[x] + up[x + 1]) >> 1)) & 255
            else:
                for x in range(w):
                    a, b, c = cur[x], up[x + 1], up[x]
                    pa, pb, pc = np.abs(b - c), np.abs(a - c), np.abs(a + b - 2 * c)
                    pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))
                    cur[x + 1] = (ln[x] + pred) & 255
            out[y] = cur[1:].astype(np.uint8).reshape(stride)
        else:
            raise ValueError(f"PNG: unknown filter type {ft}")
    return out


def read_png(path: str):
    """-> (width, height, RGBA8 bytes), what `image::open(path)?.to_rgba8()` yields for 8-bit PNGs."""
    np = _np()
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != _SIG:
        raise ValueError(f"{path}: not a PNG file")
    pos, idat, hdr, palette, trns = 8, [], None, None, None
    while pos + 8 <= len(data):
        (n,) = struct.unpack(">I", data[pos:pos + 4])
        typ, body = data[pos + 4:pos + 8], data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"PLTE":
            palette = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif typ == b"tRNS":
            trns = np.frombuffer(body, np.uint8)
        elif typ == b"IDAT":
            idat.append(body)
        elif typ == b"IEND":
            break
    if hdr is None:
        raise ValueError(f"{path}: PNG without IHDR")
    w, h, depth, ctype, _, _, interlace = hdr
    if depth != 8 or interlace != 0 or ctype not in (0, 2, 3, 4, 6):
        raise ValueError(f"{path}: unsupported PNG flavour (depth {depth}, colour type {ctype}, interlace {interlace})")
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    px = _unfilter(zlib.decompress(b"".join(idat)), w, h, ch).reshape(h, w, ch)
    rgba = np.full((h, w, 4), 255, np.uint8)
    if ctype == 6:
        rgba = px
    elif ctype == 2:
        rgba[..., :3] = px
    elif ctype == 0:
        rgba[..., :3] = px
    elif ctype == 4:
        rgba[..., :3] = px[..., :1]
        rgba[..., 3] = px[..., 1]
    else:
        if palette is None:
            raise ValueError(f"{path}: palette PNG without PLTE")
        rgba[..., :3] = palette[px[..., 0]]
        if trns is not None:
            alpha = np.full(256, 255, np.uint8)
            alpha[:len(trns)] = trns
            rgba[..., 3] = alpha[px[..., 0]]
    return w, h, np.ascontiguousarray(rgba).tobytes()


def write_png(path: str, width: int, height: int, rgba: bytes) -> None:
    """8-bit RGBA, filter 0 on every row."""
    np = _np()
    if len(rgba) != width * height * 4:
        raise ValueError("write_png: buffer size does not match width * height * 4")
    rows = np.zeros((height, width * 4 + 1), np.uint8)
    rows[:, 1:] = np.frombuffer(rgba, np.uint8).reshape(height, width * 4)

    def chunk(typ: bytes, body: bytes) -> bytes:
        return struct.pack(">I", len(body)) + typ + body + struct.pack(">I", zlib.crc32(typ + body) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(_SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, 8, 6, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(rows.tobytes(), 6)) + chunk(b"IEND", b""))


# quality -> algorithm when none is given (Nu_scale/src/upscale/mod.rs:295-303)
_QUALITY_TO_ALGORITHM = {"ultra": "lanczos3", "quality": "bicubic", "balanced": "bicubic", "performance": "bilinear"}


def output_size(width: int, height: int, scale_factor: float):
    """`(input_width as f32 * scale_factor) as u32` (Nu_scale/src/upscale/mod.rs:320-321)."""
    np = _np()
    s = np.float32(scale_factor)
    return int(np.float32(width) * s), int(np.float32(height) * s)


def upscale_image_file(input_path: str, output_path: str, technology: str = "fallback", quality: str = "quality",
                       scale_factor: float = 2.0, algorithm: str | None = None, device: int = 0) -> tuple[int, int]:
    """Nu_scale/src/upscale/mod.rs:307-338: load, upscale, save.  Returns the output size.

    `technology`: "fallback" (the algorithm menu), "fsr" (the FSR1-style pass pair), "none"
    (pass-through copy, as PassThroughUpscaler).  "dlss" has no equivalent on this hardware and is an
    error rather than a silent substitute.
    """
    w, h, px = read_png(input_path)
    tech = str(technology).lower()
    if tech == "none":
        write_png(output_path, w, h, px)
        return w, h
    if tech == "dlss":
        raise ValueError("technology 'dlss' is not available on this device")
    if tech not in ("fallback", "fsr", "wgpu"):
        raise ValueError(f"unknown technology '{technology}'")
    ow, oh = output_size(w, h, scale_factor)
    if ow == 0 or oh == 0:
        raise ValueError("scale factor gives an empty output image")
    if tech == "fsr":
        alg = "fsr1"
    else:
        alg = (algorithm or _QUALITY_TO_ALGORITHM.get(str(quality).lower(), "bicubic")).lower()
    up = PyWgpuUpscaler(quality, alg, device=device)
    up.initialize(w, h, ow, oh)
    write_png(output_path, ow, oh, up.upscale(px))
    return ow, oh


def interpolate_image_files(path_a: str, path_b: str, output_path: str, time_t: float = 0.5, estimate_flow: bool = False,
                            device: int = 0) -> tuple[int, int]:
    """The in-between frame of two equally sized images (wgpu_interpolator.rs:215-491); with
    `estimate_flow` the Horn-Schunck front end supplies the motion field, otherwise zero flow as the
    reference's live path."""
    w, h, a = read_png(path_a)
    wb, hb, b = read_png(path_b)
    if (w, h) != (wb, hb):
        raise ValueError(f"frame sizes differ: {w}x{h} vs {wb}x{hb}")
    it = WgpuFrameInterpolator(device=device)
    flow = None
    if estimate_flow:
        from .flow import FlowEstimator
        flow = FlowEstimator(device=device).estimate(a, b, w, h)
    out = it.interpolate_py(a, b, w, h, time_t=time_t, flow=flow) if flow is not None else it.interpolate_py(a, b, w, h, time_t=time_t)
    write_png(output_path, w, h, out)
    return w, h
