#!/usr/bin/env python3
"""Writes a synthetic light probe as a Radiance .hdr file (angular-map layout).

The reference's demo probe (uffizi_probe.hdr) is not distributed with it; this seeded analytic
sky + sun stands in for it.  The file uses the flat (non-RLE) RGBE scanline form unless --rle.
"""
import argparse

import numpy as np


def probe_pixels(size, seed=1):
    """float32 RGB, (size, size, 3), first row = top of the image."""
    rng = np.random.default_rng(seed)
    j, i = np.mgrid[0:size, 0:size]
    u = (i + 0.5) / size * 2 - 1
    v = 1 - (j + 0.5) / size * 2          # top row has v = +1
    r = np.sqrt(u * u + v * v)
    theta = np.pi * r                      # angular map: angle from +z... (only a pattern here)
    up = v / np.maximum(r, 1e-6) * np.sin(np.minimum(theta, np.pi))
    sky = np.stack([0.35 + 0.25 * up, 0.45 + 0.3 * up, 0.7 + 0.3 * up], -1)
    ground = np.stack([0.25 + 0 * up, 0.2 + 0 * up, 0.15 + 0 * up], -1)
    img = np.where((up > 0)[..., None], sky, ground)
    sun = np.exp(-((u - 0.3) ** 2 + (v - 0.45) ** 2) / 0.002)
    img = img + sun[..., None] * np.array([40.0, 36.0, 28.0])
    img = img * (1.0 + 0.05 * rng.standard_normal((size, size, 1)))
    return np.maximum(img, 0).astype(np.float32)


def float_to_rgbe(img):
    m = img.max(axis=-1)
    e = np.zeros(m.shape, np.int32)
    nz = m > 1e-32
    mant, ex = np.frexp(m[nz])
    scale = np.zeros_like(m)
    scale[nz] = mant * 256.0 / m[nz]
    e[nz] = ex + 128
    rgb = np.clip((img * scale[..., None]).astype(np.int32), 0, 255)
    out = np.concatenate([rgb, e[..., None]], -1).astype(np.uint8)
    out[~nz] = 0
    return out


def rle_scanline(row):
    """New-style RLE of one scanline (row: (w, 4) uint8)."""
    w = row.shape[0]
    out = bytearray([2, 2, (w >> 8) & 0xFF, w & 0xFF])
    for c in range(4):
        data = row[:, c]
        x = 0
        while x < w:
            run = 1
            while x + run < w and run < 127 and data[x + run] == data[x]:
                run += 1
            if run >= 4:
                out += bytes([128 + run, int(data[x])])
                x += run
            else:
                start = x
                x += 1
                while x < w and x - start < 128:
                    if x + 3 < w and data[x] == data[x + 1] == data[x + 2] == data[x + 3]:
                        break
                    x += 1
                out += bytes([x - start]) + data[start:x].tobytes()
    return bytes(out)


def write_hdr(path, img, rle=False):
    h, w = img.shape[:2]
    rgbe = float_to_rgbe(img)
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w))
        if rle:
            for y in range(h):
                f.write(rle_scanline(rgbe[y]))
        else:
            f.write(rgbe.tobytes())


def decode_rgbe(rgbe):
    """Independent decoder used by the tests: mantissa * 2^(e-136), zero when e == 0."""
    e = rgbe[..., 3].astype(np.int32)
    scale = np.where(e != 0, np.ldexp(1.0, e - 136), 0.0).astype(np.float32)
    return (rgbe[..., :3].astype(np.float32) * scale[..., None]).astype(np.float32)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--rle", action="store_true")
    a = ap.parse_args()
    write_hdr(a.out, probe_pixels(a.size, a.seed), a.rle)
    print("wrote", a.out)
