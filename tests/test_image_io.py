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


def _png_chunk(tag, data):
    import struct
    import zlib
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def _adam7_png(samples, depth, ctype, palette=None):
    """An Adam7-interlaced PNG of `samples` (h x w x channels integers of `depth` bits), written here because
    Pillow only reads them: seven reduced images, rows packed most significant bit first, filter type 0 or 1."""
    import struct
    import zlib
    h, w, ch = samples.shape
    raw = bytearray()
    for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
        sub = samples[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        for j, row in enumerate(sub):
            vals = row.reshape(-1)
            if depth == 16:
                line = b"".join(struct.pack(">H", int(v)) for v in vals)
            elif depth == 8:
                line = bytes(int(v) for v in vals)
            else:
                bits = "".join(format(int(v), "0%db" % depth) for v in vals)
                bits += "0" * (-len(bits) % 8)
                line = bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))
            bpp = max(ch * depth // 8, 1)
            if j % 2:  # every other row with the Sub filter
                line = bytes((line[i] - (line[i - bpp] if i >= bpp else 0)) & 255 for i in range(len(line)))
                raw += b"\x01" + line
            else:
                raw += b"\x00" + line
    png = b"\x89PNG\r\n\x1a\n" + _png_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1))
    if palette is not None:
        png += _png_chunk(b"PLTE", bytes(palette))
    return png + _png_chunk(b"IDAT", zlib.compress(bytes(raw))) + _png_chunk(b"IEND", b"")


@pytest.mark.parametrize("w,h", [(1, 1), (3, 2), (5, 9), (23, 17), (64, 40)])
@pytest.mark.parametrize("ctype,depth", [(2, 8), (6, 8), (0, 8), (4, 8), (3, 4), (3, 8), (0, 1), (0, 16), (2, 16)])
def test_interlaced_png(tmp_path, w, h, ctype, depth):
    """Adam7 files (the reference's stbi_load reads them, src/cs488.h:839-846), including sizes at which some of the
    seven passes are empty.  The file written here is first read back with Pillow, an independent decoder."""
    rng = np.random.default_rng(w * 100 + h + depth + ctype)
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    samples = rng.integers(0, 1 << depth, (h, w, ch))
    palette = rng.integers(0, 256, 3 << depth, dtype=np.uint8) if ctype == 3 else None
    (tmp_path / "i.png").write_bytes(_adam7_png(samples, depth, ctype, palette))
    if ctype == 3:
        want = palette.reshape(-1, 3)[samples[..., 0]]
    else:
        v8 = (samples >> 8 if depth == 16 else samples * {1: 255, 2: 85, 4: 17, 8: 1}[depth]).astype(np.uint8)
        want = np.repeat(v8[..., :1], 3, -1) if ch < 3 else v8[..., :3]
    if depth <= 8:  # (Pillow keeps 16-bit grey as I;16 and scales it its own way; the 8-bit forms pin the writer above)
        assert np.array_equal(np.asarray(Image.open(tmp_path / "i.png").convert("RGBA").convert("RGB") if ch in (2, 4)
                                         else Image.open(tmp_path / "i.png").convert("RGB"), np.uint8), want)
    got = _texture_via_mtl(tmp_path, "i.png")
    assert np.array_equal(got, want)


def test_png_with_an_unknown_interlace_method_is_refused_and_scene_still_loads(tmp_path, capfd):
    # hand-made IHDR with interlace method 2, which PNG does not define
    import struct
    import zlib
    png = b"\x89PNG\r\n\x1a\n" + _png_chunk(b"IHDR", struct.pack(">IIBBBBB", 2, 2, 8, 2, 0, 0, 2)) + _png_chunk(b"IDAT", zlib.compress(b"\0" * 14)) + _png_chunk(b"IEND", b"")
    (tmp_path / "i.png").write_bytes(png)
    (tmp_path / "t.mtl").write_text("newmtl m\nmap_Kd i.png\n")
    (tmp_path / "t.obj").write_text("mtllib t.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nusemtl m\nf 1/1 2/1 3/1\n")
    s = pta.Scene(pta.DEVICE_NONE)
    assert s.load(str(tmp_path / "t.obj"))
    assert "Unable to load texture" in capfd.readouterr().err
    m = s.get_material(0)
    assert m.textured == 1 and m.tex_width == 0


@pytest.mark.parametrize("fmt,mode,kw", [
    ("tga", "RGB", {}), ("tga", "RGBA", {}), ("tga", "L", {}), ("tga", "P", {}),
    ("tga", "RGB", {"compression": "tga_rle"}), ("tga", "RGBA", {"compression": "tga_rle"}), ("tga", "L", {"compression": "tga_rle"}),
    ("tga", "P", {"compression": "tga_rle"}),
    ("bmp", "RGB", {}), ("bmp", "RGBA", {}), ("bmp", "L", {}), ("bmp", "P", {}), ("bmp", "1", {}),
])
def test_tga_and_bmp_textures(tmp_path, fmt, mode, kw):
    """map_Kd files in the two other formats scene files use (the reference's stbi_load reads them, src/cs488.h:839-846):
    files written by Pillow, read through the .mtl loader, compared with Pillow's own decoding to RGB (alpha dropped).
    The reference holds no such file, so parity with its decoder is not pinned beyond this."""
    rng = np.random.default_rng(5)
    w, h = 37, 21  # odd sizes: BMP rows are padded to four bytes
    if mode == "P":
        img = Image.fromarray(rng.integers(0, 256, (h, w), dtype=np.uint8), "P")
        img.putpalette(list(rng.integers(0, 256, 768, dtype=np.uint8)))
    elif mode == "1":
        img = Image.fromarray((rng.integers(0, 2, (h, w), dtype=np.uint8) * 255), "L").convert("1")
    else:
        n = {"L": 1, "RGB": 3, "RGBA": 4}[mode]
        arr = rng.integers(0, 256, (h, w, n), dtype=np.uint8)
        # runs, so that a run-length encoder has something to encode
        arr[3:9, 5:30] = arr[3, 5]
        img = Image.fromarray(arr[..., 0] if n == 1 else arr, mode)
    name = "tex." + fmt
    img.save(tmp_path / name, **kw)
    got = _texture_via_mtl(tmp_path, name)
    back = Image.open(tmp_path / name)
    want = np.asarray(back.convert("RGBA").convert("RGB") if back.mode in ("RGBA", "LA") else back.convert("RGB"), np.uint8)
    if back.mode == "RGBA":  # alpha is dropped, not composited
        want = np.asarray(back, np.uint8)[..., :3]
    assert got.shape == want.shape and np.array_equal(got, want)


def test_bottom_up_and_top_down_tga_and_16_bit_pixels(tmp_path):
    """Hand-written headers for what Pillow does not write: a bottom-up TGA (descriptor bit 5 clear), 16-bit TGA pixels
    (1-5-5-5, scaled as x * 255 / 31), a top-down 32-bit BMP with bit masks and a 16-bit 5-6-5 BMP."""
    import struct
    w, h = 5, 3
    rng = np.random.default_rng(9)
    px = rng.integers(0, 1 << 15, (h, w), dtype=np.uint16)
    for top_down in (False, True):
        hdr = struct.pack("<BBBHHBHHHHBB", 0, 0, 2, 0, 0, 0, 0, 0, w, h, 16, 0x20 if top_down else 0)
        (tmp_path / "t16.tga").write_bytes(hdr + px.astype("<u2").tobytes())
        got = _texture_via_mtl(tmp_path, "t16.tga")
        rows = px if top_down else px[::-1]
        want = np.stack([(((rows >> 10) & 31).astype(np.uint32) * 255) // 31, (((rows >> 5) & 31).astype(np.uint32) * 255) // 31,
                         ((rows & 31).astype(np.uint32) * 255) // 31], -1).astype(np.uint8)
        assert np.array_equal(got, want)
    # 32-bit BMP, top-down (negative height), BI_BITFIELDS with R and B swapped against the default
    v = rng.integers(0, 1 << 32, (h, w), dtype=np.uint64).astype(np.uint32)
    masks = (0x000000FF, 0x0000FF00, 0x00FF0000)
    body = v.astype("<u4").tobytes()
    head = struct.pack("<2sIHHI", b"BM", 14 + 40 + 12 + len(body), 0, 0, 14 + 40 + 12)
    dib = struct.pack("<IiiHHIIiiII", 40, w, -h, 1, 32, 3, len(body), 0, 0, 0, 0) + struct.pack("<III", *masks)
    (tmp_path / "m.bmp").write_bytes(head + dib + body)
    got = _texture_via_mtl(tmp_path, "m.bmp")
    want = np.stack([(v >> 0) & 255, (v >> 8) & 255, (v >> 16) & 255], -1).astype(np.uint8)
    assert np.array_equal(got, want)
    # 16-bit 5-6-5 BMP, bottom-up, rows padded to four bytes
    p16 = rng.integers(0, 1 << 16, (h, w), dtype=np.uint16)
    stride = ((w * 16 + 31) // 32) * 4
    rows = b"".join(r.astype("<u2").tobytes().ljust(stride, b"\0") for r in p16[::-1])
    head = struct.pack("<2sIHHI", b"BM", 14 + 40 + 12 + len(rows), 0, 0, 14 + 40 + 12)
    dib = struct.pack("<IiiHHIIiiII", 40, w, h, 1, 16, 3, len(rows), 0, 0, 0, 0) + struct.pack("<III", 0xF800, 0x07E0, 0x001F)
    (tmp_path / "p.bmp").write_bytes(head + dib + rows)
    got = _texture_via_mtl(tmp_path, "p.bmp")
    r5, g6, b5 = (p16 >> 11) & 31, (p16 >> 5) & 63, p16 & 31
    want = np.stack([(r5 << 3) | (r5 >> 2), (g6 << 2) | (g6 >> 4), (b5 << 3) | (b5 >> 2)], -1).astype(np.uint8)
    assert np.array_equal(got, want)


def test_broken_tga_and_bmp_are_refused_and_the_scene_still_loads(tmp_path):
    import struct
    (tmp_path / "cut.bmp").write_bytes(b"BM" + b"\0" * 30)
    (tmp_path / "rle.bmp").write_bytes(struct.pack("<2sIHHI", b"BM", 70, 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, 2, 2, 1, 8, 1, 0, 0, 0, 0, 0) + b"\0" * 16)
    (tmp_path / "cut.tga").write_bytes(struct.pack("<BBBHHBHHHHBB", 0, 0, 10, 0, 0, 0, 0, 0, 4, 4, 24, 0) + b"\x83\x01\x02")
    for name in ("cut.bmp", "rle.bmp", "cut.tga"):
        (tmp_path / "t.mtl").write_text("newmtl m\nKd 1 1 1\nmap_Kd %s\n" % name)
        (tmp_path / "t.obj").write_text("mtllib t.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nusemtl m\nf 1/1 2/2 3/3\n")
        s = pta.Scene(pta.DEVICE_NONE)
        assert s.load(str(tmp_path / "t.obj")) and s.num_triangles() == 1  # renders untextured, as after a failed stbi_load


@pytest.mark.parametrize("kind", ["random", "smooth", "two_colours", "interlaced", "offset_frame", "transparent"])
def test_gif_textures(tmp_path, kind):
    """First frame of a GIF as a texture (the reference's stbi_load reads them, src/cs488.h:839-846): files written by
    Pillow -- long and short LZW strings, table overflow and clear codes, interlaced rows --, or assembled here for a
    frame smaller than the screen and a transparent colour; compared with the palette look-up."""
    rng = np.random.default_rng(5)
    w, h = 67, 45
    pal = rng.integers(0, 256, 768, dtype=np.uint8)
    if kind == "random":
        idx = rng.integers(0, 256, (h, w), dtype=np.uint8)
    elif kind == "smooth":
        idx = ((np.arange(w)[None, :] // 5 + np.arange(h)[:, None] // 7) % 256).astype(np.uint8)  # long repeated strings
        w, h = 400, 300
        idx = np.resize(idx, (h, w))
    elif kind == "two_colours":
        idx = rng.integers(0, 2, (h, w), dtype=np.uint8)
    else:
        idx = rng.integers(0, 200, (h, w), dtype=np.uint8)
    img = Image.fromarray(idx, "P")
    img.putpalette(pal.tolist())
    path = tmp_path / "g.gif"
    want = pal.reshape(256, 3)[idx]
    if kind == "interlaced":
        img.save(path, interlace=1)
    elif kind in ("offset_frame", "transparent"):
        # re-pack Pillow's image block into a file with a larger screen / a graphic control extension
        import struct
        img.save(path, interlace=0)
        raw = path.read_bytes()
        at = raw.index(b"\x2c", 13 + 768)   # the image descriptor follows the global colour table (and extensions)
        while raw[at + 1:at + 9] != struct.pack("<HHHH", 0, 0, w, h):
            at = raw.index(b"\x2c", at + 1)
        block = raw[at + 10:]
        if kind == "offset_frame":
            W, H, ox, oy = w + 9, h + 6, 4, 3
            head = b"GIF89a" + struct.pack("<HHBBB", W, H, 0xF7, 0, 0) + bytes(pal)
            path.write_bytes(head + b"\x2c" + struct.pack("<HHHHB", ox, oy, w, h, 0) + block)
            full = np.zeros((H, W, 3), np.uint8)
            full[oy:oy + h, ox:ox + w] = want
            want = full
        else:
            head = b"GIF89a" + struct.pack("<HHBBB", w, h, 0xF7, 0, 0) + bytes(pal)
            gce = b"\x21\xf9\x04" + struct.pack("<BHB", 1, 0, 17) + b"\x00"   # colour 17 is transparent
            path.write_bytes(head + gce + b"\x2c" + struct.pack("<HHHHB", 0, 0, w, h, 0) + block)
            want = want.copy()
            want[idx == 17] = 0
    else:
        img.save(path)
    if kind not in ("offset_frame", "transparent"):
        assert np.array_equal(np.asarray(Image.open(path).convert("RGB"), np.uint8), want)  # Pillow wrote what we think
    got = _texture_via_mtl(tmp_path, "g.gif")
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.parametrize("mode", ["RGB", "L", "I;16"])
def test_pnm_textures(tmp_path, mode):
    rng = np.random.default_rng(6)
    w, h = 31, 22
    if mode == "I;16":
        arr = rng.integers(0, 65536, (h, w), dtype=np.uint16)
        (tmp_path / "p.pgm").write_bytes(b"P5\n# a comment\n%d %d\n65535\n" % (w, h) + arr.astype(">u2").tobytes())
        want = np.repeat((arr >> 8).astype(np.uint8)[..., None], 3, -1)
        name = "p.pgm"
    else:
        arr = rng.integers(0, 256, (h, w, 3) if mode == "RGB" else (h, w), dtype=np.uint8)
        name = "p.ppm" if mode == "RGB" else "p.pgm"
        Image.fromarray(arr, mode).save(tmp_path / name)
        want = arr if mode == "RGB" else np.repeat(arr[..., None], 3, -1)
    got = _texture_via_mtl(tmp_path, name)
    assert np.array_equal(got, want)


def test_corrupt_texture_files_are_refused_or_decoded_never_fatal(tmp_path):
    """Truncated and bit-flipped files of every texture format: the scene still loads (the reference prints "Unable to
    load texture" and goes on, src/cs488.h:843-845).  (The same mutations, 7 800 files, ran clean through the decoders
    built with AddressSanitizer and UBSan on the host.)"""
    rng = np.random.default_rng(9)
    w, h = 24, 18
    idx = rng.integers(0, 256, (h, w), dtype=np.uint8)
    pimg = Image.fromarray(idx, "P")
    pimg.putpalette(rng.integers(0, 256, 768, dtype=np.uint8).tolist())
    cimg = Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8), "RGB")
    seeds = tmp_path / "seeds"
    seeds.mkdir()
    pimg.save(seeds / "a.gif")
    pimg.save(seeds / "b.gif", interlace=1)
    cimg.save(seeds / "c.ppm")
    cimg.save(seeds / "d.tga")
    cimg.save(seeds / "e.tga", compression="tga_rle")
    cimg.save(seeds / "f.bmp")
    pimg.save(seeds / "g.bmp")
    cimg.save(seeds / "h.jpg")
    cimg.save(seeds / "i.jpg", progressive=True)
    cimg.save(seeds / "j.png")
    (seeds / "k.png").write_bytes(_adam7_png(rng.integers(0, 256, (h, w, 3)), 8, 2))
    (tmp_path / "t.obj").write_text("mtllib t.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nusemtl m\nf 1/1 2/2 3/3\n")
    for seed in sorted(seeds.iterdir()):
        raw = seed.read_bytes()
        for k in range(40):
            b = bytearray(raw)
            if k % 3 == 0:
                b = b[:rng.integers(1, len(b))]
            elif k % 3 == 1:
                for _ in range(rng.integers(1, 6)):
                    b[rng.integers(0, len(b))] = rng.integers(0, 256)
            else:
                b[rng.integers(0, min(len(b), 64))] = rng.integers(0, 256)
            name = "m" + seed.suffix
            (tmp_path / name).write_bytes(bytes(b))
            (tmp_path / "t.mtl").write_text("newmtl m\nKd 1 1 1\nmap_Kd %s\n" % name)
            s = pta.Scene(pta.DEVICE_NONE)
            assert s.load(str(tmp_path / "t.obj"))
            s.close()
