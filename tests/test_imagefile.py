"""PNG front end + CLI.  CPU: the codec round-trips and reads the
reference's own fixture files like the tests' independent reader; GPU: the commands produce the
oracle's pixels."""
import os
import struct
import zlib

import numpy as np
import pytest

from _png import read_png as read_png_independent
from conftest import GOLDEN


def _write_filtered_png(path, img, filter_types):
    """Encoder used only here: applies the given PNG filter type per row (cycled)."""
    h, w, ch = img.shape
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[ch]
    raw = bytearray()
    prev = np.zeros(w * ch, np.int32)
    for y in range(h):
        ft = filter_types[y % len(filter_types)]
        cur = img[y].reshape(-1).astype(np.int32)
        left = np.concatenate([np.zeros(ch, np.int32), cur[:-ch]])
        ul = np.concatenate([np.zeros(ch, np.int32), prev[:-ch]])
        if ft == 0:
            enc = cur
        elif ft == 1:
            enc = cur - left
        elif ft == 2:
            enc = cur - prev
        elif ft == 3:
            enc = cur - ((left + prev) >> 1)
        else:
            pa, pb, pc = np.abs(prev - ul), np.abs(left - ul), np.abs(left + prev - 2 * ul)
            enc = cur - np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
        raw += bytes([ft]) + (enc & 255).astype(np.uint8).tobytes()
        prev = cur

    def chunk(t, b):
        return struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(bytes(raw))) + chunk(b"IEND", b""))


@pytest.mark.parametrize("ch", [1, 2, 3, 4])
def test_png_reader_all_filters_and_colour_types(nsc, tmp_path, ch):
    from nu_scaler_amd import imagefile
    rng = np.random.default_rng(ch)
    img = rng.integers(0, 256, (13, 17, ch), dtype=np.uint8)
    p = str(tmp_path / "f.png")
    _write_filtered_png(p, img, [0, 1, 2, 3, 4])
    w, h, px = imagefile.read_png(p)
    got = np.frombuffer(px, np.uint8).reshape(h, w, 4)
    assert (w, h) == (17, 13)
    assert np.array_equal(got, read_png_independent(p))
    if ch >= 3:
        assert np.array_equal(got[..., :3], img[..., :3])
    else:
        assert np.array_equal(got[..., 0], img[..., 0]) and np.array_equal(got[..., 1], img[..., 0])
    assert np.array_equal(got[..., 3], img[..., ch - 1] if ch in (2, 4) else np.full((13, 17), 255))


def test_png_roundtrip_and_reference_fixtures(nsc, tmp_path):
    from nu_scaler_amd import imagefile
    for name in ("ref_test_input.png", "ref_test_output.png", "ref_interp_half.png"):
        w, h, px = imagefile.read_png(os.path.join(GOLDEN, name))
        want = read_png_independent(os.path.join(GOLDEN, name))
        assert want.shape == (h, w, 4) and px == want.tobytes()
        p = str(tmp_path / name)
        imagefile.write_png(p, w, h, px)
        assert imagefile.read_png(p) == (w, h, px)
        assert np.array_equal(read_png_independent(p), want)
    with pytest.raises(ValueError, match="not a PNG"):
        bad = tmp_path / "bad.png"
        bad.write_bytes(b"hello")
        imagefile.read_png(str(bad))
    with pytest.raises(ValueError, match="buffer size"):
        imagefile.write_png(str(tmp_path / "x.png"), 4, 4, b"\0" * 10)


def test_output_size_truncates_like_the_reference(nsc):
    from nu_scaler_amd import imagefile
    # (w as f32 * scale) as u32 -- Nu_scale/src/upscale/mod.rs:320-321
    assert imagefile.output_size(320, 240, 2.0) == (640, 480)
    assert imagefile.output_size(321, 241, 1.5) == (481, 361)
    assert imagefile.output_size(100, 50, 1.33) == (133, 66)


def test_cli_parser_and_error_exit(nsc, tmp_path, capsys):
    from nu_scaler_amd import cli
    a = cli.build_parser().parse_args(["upscale", "a.png", "b.png", "--algorithm", "lanczos3", "--scale", "1.5"])
    assert (a.command, a.algorithm, a.scale, a.tech, a.quality) == ("upscale", "lanczos3", 1.5, "fallback", "quality")
    assert cli.main(["upscale", str(tmp_path / "missing.png"), str(tmp_path / "o.png")]) == 1
    assert "error" in capsys.readouterr().err


@pytest.mark.gpu
def test_cli_upscale_and_interpolate_match_oracle(nsc, oracle_mod, tmp_path):
    from nu_scaler_amd import cli, imagefile
    src = os.path.join(GOLDEN, "ref_test_input.png")
    img = read_png_independent(src)
    h, w = img.shape[:2]
    cases = [(["--algorithm", "bilinear"], lambda: oracle_mod.bilinear(img, 2 * w, 2 * h), 0),
             (["--algorithm", "nearest", "--scale", "1.5"], lambda: oracle_mod.nearest(img, int(w * 1.5), int(h * 1.5)), 0),
             (["--quality", "ultra"], lambda: oracle_mod.lanczos3(img, 2 * w, 2 * h), 1),
             (["--quality", "balanced"], lambda: oracle_mod.resize(img, 2 * w, 2 * h, oracle_mod.FILTER_CATMULLROM), 1),
             (["--tech", "fsr", "--quality", "ultra"], lambda: oracle_mod.fsr1(img, 2 * w, 2 * h, 0.0, 0.8), 0),
             (["--tech", "none"], lambda: img, 0)]
    for extra, want, tol in cases:
        out = str(tmp_path / "o.png")
        assert cli.main(["upscale", src, out] + extra) == 0
        got, exp = read_png_independent(out), want()
        assert got.shape == exp.shape
        assert int(np.abs(got.astype(int) - exp.astype(int)).max()) <= tol, extra
    # the reference's bilinear fixture through the file front end (top-left quadrant is what it holds)
    out = str(tmp_path / "bl.png")
    imagefile.upscale_image_file(src, out, "fallback", "performance", 2.0)
    ref = read_png_independent(os.path.join(GOLDEN, "ref_test_output.png"))
    assert np.array_equal(read_png_independent(out)[:h, :w], ref[:h, :w])
    # interpolate: the two boxes of the reference's test_interpolator.py -> its interp_half.png
    a, b = oracle_mod.gen_box(64, 64, (255, 0, 0, 255)), oracle_mod.gen_box(64, 64, (0, 0, 255, 255))
    pa, pb, po = (str(tmp_path / n) for n in ("a.png", "b.png", "mid.png"))
    imagefile.write_png(pa, 64, 64, a.tobytes())
    imagefile.write_png(pb, 64, 64, b.tobytes())
    assert cli.main(["interpolate", pa, pb, po]) == 0
    assert np.array_equal(read_png_independent(po), oracle_mod.warp_blend(a, b, None, 0.5))
    assert cli.main(["interpolate", pa, pb, po, "--flow", "--t", "0.25"]) == 0
    assert read_png_independent(po).shape == (64, 64, 4)
    assert cli.main(["upscale", src, out, "--tech", "dlss"]) == 1


def _native_cli(nsc):
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(nsc._capi.LIB_PATH)), "bin", "nu_scaler_cli")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", nsc._capi.CSRC_DIR, "../bin/nu_scaler_cli"], check=True, capture_output=True)
    return exe


@pytest.mark.parametrize("ch", [1, 2, 3, 4])
def test_native_cli_png_codec(nsc, tmp_path, ch):
    """nu_scaler_cli (C++ on the C ABI + zlib): its PNG decoder and encoder against the independent reader, every
    filter type and colour type; usage and error exits.  No GPU involved."""
    import subprocess
    exe = _native_cli(nsc)
    rng = np.random.default_rng(10 + ch)
    img = rng.integers(0, 256, (13, 17, ch), dtype=np.uint8)
    src, dst = str(tmp_path / "f.png"), str(tmp_path / "g.png")
    _write_filtered_png(src, img, [0, 1, 2, 3, 4])
    r = subprocess.run([exe, "png-copy", src, dst], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(read_png_independent(dst), read_png_independent(src))
    assert subprocess.run([exe, "--help"], capture_output=True).returncode == 0
    assert subprocess.run([exe], capture_output=True).returncode == 2
    r = subprocess.run([exe, "upscale", str(tmp_path / "missing.png"), dst], capture_output=True, text=True)
    assert r.returncode == 1 and "error" in r.stderr
    bad = tmp_path / "bad.png"
    bad.write_bytes(b"hello")
    r = subprocess.run([exe, "png-copy", str(bad), dst], capture_output=True, text=True)
    assert r.returncode == 1 and "not a PNG" in r.stderr


@pytest.mark.gpu
def test_native_cli_upscale_and_interpolate_match_oracle(nsc, oracle_mod, tmp_path):
    import subprocess
    from nu_scaler_amd import imagefile
    exe = _native_cli(nsc)
    src = os.path.join(GOLDEN, "ref_test_input.png")
    img = read_png_independent(src)
    h, w = img.shape[:2]
    cases = [(["--algorithm", "bilinear"], lambda: oracle_mod.bilinear(img, 2 * w, 2 * h), 0),
             (["--algorithm", "nearest", "--scale", "1.5"], lambda: oracle_mod.nearest(img, int(w * 1.5), int(h * 1.5)), 0),
             (["--quality", "ultra"], lambda: oracle_mod.lanczos3(img, 2 * w, 2 * h), 1),
             (["--tech", "fsr", "--quality", "ultra"], lambda: oracle_mod.fsr1(img, 2 * w, 2 * h, 0.0, 0.8), 0),
             (["--tech", "none"], lambda: img, 0)]
    for extra, want, tol in cases:
        out = str(tmp_path / "o.png")
        r = subprocess.run([exe, "upscale", src, out] + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        got, exp = read_png_independent(out), want()
        assert got.shape == exp.shape and int(np.abs(got.astype(int) - exp.astype(int)).max()) <= tol, extra
    a, b = oracle_mod.gen_box(64, 64, (255, 0, 0, 255)), oracle_mod.gen_box(64, 64, (0, 0, 255, 255))
    pa, pb, po = (str(tmp_path / n) for n in ("a.png", "b.png", "mid.png"))
    imagefile.write_png(pa, 64, 64, a.tobytes())
    imagefile.write_png(pb, 64, 64, b.tobytes())
    assert subprocess.run([exe, "interpolate", pa, pb, po], capture_output=True).returncode == 0
    assert np.array_equal(read_png_independent(po), oracle_mod.warp_blend(a, b, None, 0.5))
    assert subprocess.run([exe, "interpolate", pa, pb, po, "--flow", "--t", "0.25"], capture_output=True).returncode == 0
    assert subprocess.run([exe, "upscale", src, po, "--tech", "dlss"], capture_output=True).returncode == 1
