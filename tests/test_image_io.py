"""Image readers and writers of the product library against independent decoders (numpy RGBE
decoder in tools/make_probe.py, PIL for PNG)."""
import os

import numpy as np
import pytest
from PIL import Image

import path_tracer_amd as pta
from common import read_hdr_independent, scene_path
from make_probe import probe_pixels, write_hdr


@pytest.mark.parametrize("name", ["probe_64.hdr", "probe_64_rle.hdr"])
def test_hdr_reader(name):
    s = pta.Scene(pta.DEVICE_NONE)
    s.load_env(scene_path(name))
    got = s.get_env()                      # stored bottom row first
    want = read_hdr_independent(scene_path(name))[::-1]
    assert got.shape == (64, 64, 3)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert got.max() > 10.0  # the sun


def test_hdr_flat_and_rle_decode_alike(tmp_path):
    img = probe_pixels(40, seed=5)
    img[3:9, 5:30] = [1.5, 0.25, 0.0]  # long runs for the RLE coder
    img[10, :] = 0.0                   # zero exponent
    write_hdr(str(tmp_path / "a.hdr"), img, rle=False)
    write_hdr(str(tmp_path / "b.hdr"), img, rle=True)
    out = []
    for f in ("a.hdr", "b.hdr"):
        s = pta.Scene(pta.DEVICE_NONE)
        s.load_env(str(tmp_path / f))
        out.append(s.get_env())
    assert np.array_equal(out[0], out[1])
    # RGBE keeps 8 bits relative to the largest channel of a texel (mantissas are truncated)
    assert np.all(np.abs(out[0][::-1] - img) <= img.max(-1, keepdims=True) / 128 + 1e-6)
    assert not out[0][::-1][10].any()


def test_hdr_errors(tmp_path):
    s = pta.Scene(pta.DEVICE_NONE)
    with pytest.raises(pta.PtError) as e:
        s.load_env(str(tmp_path / "missing.hdr"))
    assert e.value.code == pta.ERR_IO
    (tmp_path / "bad.hdr").write_bytes(b"P6\n1 1\n255\nabc")
    with pytest.raises(pta.PtError):
        s.load_env(str(tmp_path / "bad.hdr"))
    (tmp_path / "short.hdr").write_bytes(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 4 +X 4\n" + b"\x01" * 10)
    with pytest.raises(pta.PtError):
        s.load_env(str(tmp_path / "short.hdr"))


def _texture_via_mtl(tmp_path, png_name):
    (tmp_path / "t.mtl").write_text("newmtl m\nKd 1 1 1\nmap_Kd %s\n" % png_name)
    (tmp_path / "t.obj").write_text("mtllib t.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nusemtl m\nf 1/1 2/2 3/3\n")
    s = pta.Scene(pta.DEVICE_NONE)
    assert s.load(str(tmp_path / "t.obj"))
    return s.get_texture(0)


def test_png_reader_on_the_reference_textures(tmp_path):
    for name in ("default.png", "test.png"):
        import shutil
        shutil.copy(scene_path(name), tmp_path / name)
        got = _texture_via_mtl(tmp_path, name)
        want = np.asarray(Image.open(scene_path(name)).convert("RGB"), np.uint8)
        assert np.array_equal(got, want), name


@pytest.mark.parametrize("mode,bits", [("L", 8), ("LA", 8), ("RGB", 8), ("RGBA", 8), ("P", 8), ("P", 4), ("P", 2), ("1", 1), ("I;16", 16)])
def test_png_reader_colour_types(tmp_path, mode, bits):
    rng = np.random.default_rng(11)
    w, h = 23, 17
    if mode == "I;16":
        arr = rng.integers(0, 65536, (h, w), dtype=np.uint16)
        img = Image.fromarray(arr)
        want = np.repeat((arr >> 8).astype(np.uint8)[..., None], 3, -1)
    elif mode == "P":
        arr = rng.integers(0, 1 << bits, (h, w), dtype=np.uint8)
        img = Image.fromarray(arr, "P")
        pal = rng.integers(0, 256, 768, dtype=np.uint8)
        img.putpalette(pal.tolist())
        want = pal.reshape(256, 3)[arr]
    elif mode == "1":
        arr = rng.integers(0, 2, (h, w), dtype=np.uint8)
        img = Image.fromarray(arr * 255).convert("1")
        want = np.repeat((arr * 255)[..., None], 3, -1)
    else:
        ch = len(mode)
        arr = rng.integers(0, 256, (h, w, ch) if ch > 1 else (h, w), dtype=np.uint8)
        img = Image.fromarray(arr, mode)
        want = np.asarray(img.convert("RGB") if mode in ("L", "RGB") else Image.fromarray(arr[..., :-1] if ch > 2 else arr[..., 0]).convert("RGB"), np.uint8)
    path = tmp_path / "x.png"
    kw = {"bits": bits} if mode == "P" and bits < 8 else {}
    img.save(path, **kw)
    got = _texture_via_mtl(tmp_path, "x.png")
    assert np.array_equal(got, want)


def test_interlaced_png_is_refused_and_scene_still_loads(tmp_path, capfd):
    # hand-made IHDR with the interlace flag set
    import struct
    import zlib

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 2, 2, 8, 2, 0, 0, 1)) + chunk(b"IDAT", zlib.compress(b"\0" * 14)) + chunk(b"IEND", b"")
    (tmp_path / "i.png").write_bytes(png)
    (tmp_path / "t.mtl").write_text("newmtl m\nmap_Kd i.png\n")
    (tmp_path / "t.obj").write_text("mtllib t.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nusemtl m\nf 1/1 2/1 3/1\n")
    s = pta.Scene(pta.DEVICE_NONE)
    assert s.load(str(tmp_path / "t.obj"))
    assert "Unable to load texture" in capfd.readouterr().err
    m = s.get_material(0)
    assert m.textured == 1 and m.tex_width == 0
