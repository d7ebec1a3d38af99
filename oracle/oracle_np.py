"""numpy twin of nus_oracle.c -- TEST INFRASTRUCTURE ONLY.

An independent restatement (vectorised float32, same expression order) used to
cross-check the C oracle and to generate fixtures.  Citations as in nus_oracle.c.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def _as_u8_trunc(v: np.ndarray) -> np.ndarray:
    """Rust `f32 as u8`: truncate toward zero, saturating."""
    v = np.nan_to_num(v, nan=0.0)
    return np.clip(np.trunc(v), 0, 255).astype(np.uint8)


def nearest(img: np.ndarray, ow: int, oh: int) -> np.ndarray:
    """Nu_scale/src/upscale/common.rs:188-198."""
    ih, iw = img.shape[:2]
    sx = np.minimum((np.arange(ow, dtype=np.uint64) * np.uint64(iw)) // np.uint64(ow), iw - 1).astype(np.int64)
    sy = np.minimum((np.arange(oh, dtype=np.uint64) * np.uint64(ih)) // np.uint64(oh), ih - 1).astype(np.int64)
    return img[sy[:, None], sx[None, :]]


def _bilinear_coords(n_in: int, n_out: int, clamp_max: bool):
    o = np.arange(n_out, dtype=f32)
    s = o * f32(n_in) / f32(n_out)
    if clamp_max:
        s = np.minimum(s, f32(n_in) - f32(1.0))
    i0 = np.floor(s).astype(np.int64)
    i1 = np.minimum(i0 + 1, n_in - 1)
    d = (s - i0.astype(f32)).astype(f32)
    return i0, i1, d


def bilinear(img: np.ndarray, ow: int, oh: int) -> np.ndarray:
    """Nu_scale/src/upscale/common.rs:199-231 (CPU form, THE bilinear oracle)."""
    ih, iw = img.shape[:2]
    x0, x1, dx = _bilinear_coords(iw, ow, True)
    y0, y1, dy = _bilinear_coords(ih, oh, True)
    p = img.astype(f32)
    dx = dx[None, :, None]
    dy = dy[:, None, None]
    one = f32(1.0)
    p00 = p[y0[:, None], x0[None, :]]
    p10 = p[y0[:, None], x1[None, :]]
    p01 = p[y1[:, None], x0[None, :]]
    p11 = p[y1[:, None], x1[None, :]]
    top = p00 * (one - dx) + p10 * dx
    bottom = p01 * (one - dx) + p11 * dx
    value = top * (one - dy) + bottom * dy
    return _as_u8_trunc(np.clip(value, f32(0), f32(255)))


def bilinear_wgsl(img: np.ndarray, ow: int, oh: int) -> np.ndarray:
    """nu_scaler_core/src/upscale/mod.rs:209-263 (WGSL form, diff-only)."""
    ih, iw = img.shape[:2]
    x0, x1, dx = _bilinear_coords(iw, ow, False)
    y0, y1, dy = _bilinear_coords(ih, oh, False)
    p = img.astype(f32) / f32(255.0)
    dx = dx[None, :, None]
    dy = dy[:, None, None]
    one = f32(1.0)
    c00 = p[y0[:, None], x0[None, :]]
    c10 = p[y0[:, None], x1[None, :]]
    c01 = p[y1[:, None], x0[None, :]]
    c11 = p[y1[:, None], x1[None, :]]
    c0 = c00 * (one - dx) + c10 * dx
    c1 = c01 * (one - dx) + c11 * dx
    v = c0 * (one - dy) + c1 * dy
    return np.trunc(np.clip(v, f32(0), f32(1)) * f32(255.0)).astype(np.uint8)


def _sinc(t: np.ndarray) -> np.ndarray:
    a = (t * f32(np.pi)).astype(f32)
    with np.errstate(invalid="ignore", divide="ignore"):
        r = (np.sin(a) / a).astype(f32)
    return np.where(t == 0, f32(1.0), r).astype(f32)


def lanczos3_kernel(x: np.ndarray) -> np.ndarray:
    x = x.astype(f32)
    v = (_sinc(x) * _sinc(x / f32(3.0))).astype(f32)
    return np.where(np.abs(x) < 3.0, v, f32(0.0)).astype(f32)


def resize_axis(n_in: int, n_out: int, support: float = 3.0, kernel=lanczos3_kernel):
    """Tap windows of image-0.24.9 vertical_sample/horizontal_sample (PARITY UNPINNED)."""
    ratio = f32(n_in) / f32(n_out)
    sratio = ratio if ratio >= 1 else f32(1.0)
    src_support = f32(support) * sratio
    taps = []
    for o in range(n_out):
        c = (f32(o) + f32(0.5)) * ratio
        left = int(np.floor(c - src_support))
        left = min(max(left, 0), n_in - 1)
        right = int(np.ceil(c + src_support))
        right = min(max(right, left + 1), n_in)
        c = c - f32(0.5)
        idx = np.arange(left, right)
        w = kernel((idx.astype(f32) - c) / sratio)
        s = f32(0.0)
        for v in w:          # sequential f32 sum, as the reference accumulates
            s = f32(s + v)
        w = (w / s).astype(f32)
        taps.append((left, w))
    return taps


def lanczos3(img: np.ndarray, ow: int, oh: int) -> np.ndarray:
    """image-0.24.9 imageops::resize(.., Lanczos3): vertical pass to f32, then
    horizontal pass, clamp, round half away from zero.  PARITY UNPINNED."""
    ih, iw = img.shape[:2]
    if (iw, ih) == (ow, oh):
        return img.copy()
    src = img.astype(f32)
    tmp = np.zeros((oh, iw, 4), dtype=f32)
    for oy, (left, w) in enumerate(resize_axis(ih, oh)):
        acc = np.zeros((iw, 4), dtype=f32)
        for i, wi in enumerate(w):
            acc = (acc + src[left + i] * wi).astype(f32)
        tmp[oy] = acc
    out = np.zeros((oh, ow, 4), dtype=np.uint8)
    for ox, (left, w) in enumerate(resize_axis(iw, ow)):
        acc = np.zeros((oh, 4), dtype=f32)
        for i, wi in enumerate(w):
            acc = (acc + tmp[:, left + i] * wi).astype(f32)
        acc = np.clip(acc, f32(0), f32(255))
        # f32::round -- half away from zero (values are non-negative here)
        out[:, ox] = np.floor(acc.astype(np.float64) + 0.5).astype(np.uint8)
    return out


def _sample_trunc(frame: np.ndarray, x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """nu_scaler_core/src/interpolation/mod.rs:467-510."""
    h, w = frame.shape[:2]
    x = np.clip(x, f32(0), f32(w - 1)).astype(f32)
    y = np.clip(y, f32(0), f32(h - 1)).astype(f32)
    x0 = np.floor(x).astype(np.int64)
    y0 = np.floor(y).astype(np.int64)
    x1 = np.minimum(x0 + 1, w - 1)
    y1 = np.minimum(y0 + 1, h - 1)
    xf = (x - x0.astype(f32))[..., None].astype(f32)
    yf = (y - y0.astype(f32))[..., None].astype(f32)
    p = frame.astype(f32)
    one = f32(1.0)
    top = p[y0, x0] * (one - xf) + p[y0, x1] * xf
    bottom = p[y1, x0] * (one - xf) + p[y1, x1] * xf
    value = top * (one - yf) + bottom * yf
    return _as_u8_trunc(value)


def warp_blend(a: np.ndarray, b: np.ndarray, flow, t: float) -> np.ndarray:
    """Geometry: shaders/warp_blend.wgsl:25-43; rounding: interpolation/mod.rs:407-411."""
    h, w = a.shape[:2]
    t = f32(t)
    one = f32(1.0)
    xs = np.arange(w, dtype=f32)[None, :].repeat(h, 0)
    ys = np.arange(h, dtype=f32)[:, None].repeat(w, 1)
    if flow is None:
        fx = np.zeros((h, w), dtype=f32)
        fy = np.zeros((h, w), dtype=f32)
    else:
        fx = flow[..., 0].astype(f32)
        fy = flow[..., 1].astype(f32)
    pa = _sample_trunc(a, xs - t * fx, ys - t * fy).astype(f32)
    pb = _sample_trunc(b, xs + (one - t) * fx, ys + (one - t) * fy).astype(f32)
    return _as_u8_trunc((one - t) * pa + t * pb)


def gen_py_gradient(w: int, h: int) -> np.ndarray:
    """S2: the gradient of nu_scaler_core/upscale_test.py:13-33 (data generator)."""
    x = np.linspace(0, 1, w)
    y = np.linspace(0, 1, h)
    X, Y = np.meshgrid(x, y)
    img = np.zeros((h, w, 4), dtype=np.uint8)
    img[:, :, 0] = X * 255
    img[:, :, 1] = Y * 255
    img[:, :, 2] = ((X + Y) / 2) * 255
    img[:, :, 3] = 255
    return img


# ---- FSR1-style EASU + RCAS (nu_scaler_core/src/upscale/fsr.rs:24-260), f32 throughout ----

_F = np.float32


def _fsr_fetch(img: np.ndarray, x: np.ndarray, y: np.ndarray) -> np.ndarray:
    h, w = img.shape[:2]
    return img[np.clip(y, 0, h - 1), np.clip(x, 0, w - 1), :3].astype(_F) / _F(255.0)


def _fsr_cubic(d: np.ndarray) -> np.ndarray:
    d2 = d * d
    d3 = d * d2
    near = _F(2.0) - _F(1.5) * d - _F(0.5) * d3 + d2
    far = _F(0.0) - _F(0.5) * d + _F(2.5) * d2 - d3
    return np.where(d <= 1, near, np.where(d <= 2, far, _F(0.0))).astype(_F)


def _pack_trunc(v: np.ndarray) -> np.ndarray:
    return (np.minimum(np.maximum(v, _F(0.0)), _F(1.0)) * _F(255.0)).astype(np.uint32).astype(np.uint8)


def fsr_easu(img: np.ndarray, ow: int, oh: int, sharpness: float = 0.0) -> np.ndarray:
    ih, iw = img.shape[:2]
    s = _F(sharpness)
    gx, gy = np.meshgrid(np.arange(ow), np.arange(oh))
    cx = (gx.astype(_F) + _F(0.5)) * (_F(iw) / _F(ow))
    cy = (gy.astype(_F) + _F(0.5)) * (_F(ih) / _F(oh))
    ix, iy = cx.astype(np.int32), cy.astype(np.int32)
    fx, fy = cx - np.floor(cx), cy - np.floor(cy)
    up, dn = _fsr_fetch(img, ix, iy - 1), _fsr_fetch(img, ix, iy + 1)
    lf, rt = _fsr_fetch(img, ix - 1, iy), _fsr_fetch(img, ix + 1, iy)
    a, b = np.abs(up - dn), np.abs(lf - rt)
    vgx = (a[..., 0] + a[..., 1] + a[..., 2]) / _F(3.0)
    vgy = (b[..., 0] + b[..., 1] + b[..., 2]) / _F(3.0)
    dx, dy = vgx + _F(0.0001), vgy + _F(0.0001)
    ln = np.sqrt(dx * dx + dy * dy)
    dx, dy = dx / ln, dy / ln
    wx = np.abs(dx) / (np.abs(dx) + np.abs(dy))
    wy = _F(1.0) - wx
    acc = np.zeros((oh, ow, 3), _F)
    accw = np.zeros((oh, ow), _F)
    for y in range(4):
        for x in range(4):
            c = _fsr_fetch(img, ix - 1 + x, iy - 1 + y)
            dist = np.abs((_F(x) - fx) * wx + (_F(y) - fy) * wy)
            wgt = _fsr_cubic(dist)
            acc = acc + c * wgt[..., None]
            accw = accw + wgt
    col = acc / np.maximum(accw, _F(0.0001))[..., None]
    if s > _F(0.001):
        col = col * (_F(1.0) - s) + _fsr_fetch(img, ix, iy) * s
    out = np.empty((oh, ow, 4), np.uint8)
    out[..., :3] = _pack_trunc(col)
    out[..., 3] = 255
    return out


def fsr_rcas(img: np.ndarray, sharpness: float) -> np.ndarray:
    h, w = img.shape[:2]
    gx, gy = np.meshgrid(np.arange(w), np.arange(h))
    c = _fsr_fetch(img, gx, gy)
    t, b = _fsr_fetch(img, gx, gy - 1), _fsr_fetch(img, gx, gy + 1)
    l, r = _fsr_fetch(img, gx - 1, gy), _fsr_fetch(img, gx + 1, gy)

    def luma(p):
        return p[..., 0] * _F(0.299) + p[..., 1] * _F(0.587) + p[..., 2] * _F(0.114)

    lc, lt, lb, ll, lr = luma(c), luma(t), luma(b), luma(l), luma(r)
    mn = np.minimum(lc, np.minimum(np.minimum(lt, lb), np.minimum(ll, lr)))
    mx = np.maximum(lc, np.maximum(np.maximum(lt, lb), np.maximum(ll, lr)))
    st = np.minimum(np.maximum((mx - mn - _F(0.0)) / (_F(0.2) - _F(0.0)), _F(0.0)), _F(1.0))
    strength = _F(sharpness) * (_F(1.0) - st * st * (_F(3.0) - _F(2.0) * st))
    lap = _F(4.0) * c - t - b - l - r
    out = np.empty((h, w, 4), np.uint8)
    out[..., :3] = _pack_trunc(c + lap * strength[..., None])
    out[..., 3] = 255
    return out
