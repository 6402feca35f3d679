#!/usr/bin/env python3
"""Procedural stand-in for the Sponza atrium (the reference's media/sponza.obj is not distributed).

A seeded generator of a two-storey colonnaded courtyard: displaced stone floor and walls, fluted
columns, arches, hanging curtains (translucent, `Op`), metal vases (`Ns 100`, `Ro`), glass spheres
(material name starting with "glass"), a textured floor (`map_Kd`), and an emissive lamp (`Ke`).
It writes a Wavefront .obj/.mtl pair that uses only what the reference's loader understands:
triangles, one `v/t/n` reference per corner, lines shorter than 79 bytes.  The default detail gives
about 150 k triangles; `--detail` scales the tessellation.

This is a synthetic scene with the same kinds of geometry and materials; it is NOT the Crytek model.
"""
import argparse
import os

import numpy as np


class Mesh:
    def __init__(self):
        self.v, self.vn, self.vt, self.groups = [], [], [], []  # groups: (material, faces[n,3] of global indices)
        self.count = 0

    def add(self, material, pos, nrm, uv, faces):
        """pos/nrm (n,3), uv (n,2), faces (m,3) local indices; every corner uses index i for v, vt and vn."""
        self.v.append(pos.astype(np.float64))
        self.vn.append(nrm.astype(np.float64))
        self.vt.append(uv.astype(np.float64))
        self.groups.append((material, faces.astype(np.int64) + self.count))
        self.count += pos.shape[0]


def grid_faces(nu, nv):
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    a = (i * (nv + 1) + j).ravel()
    b = a + (nv + 1)
    return np.concatenate([np.stack([a, b, b + 1], 1), np.stack([a, b + 1, a + 1], 1)], 0)


def patch(mesh, material, origin, du, dv, nu, nv, bump=None, uv_scale=1.0):
    """Tessellated parallelogram origin + s*du + t*dv, displaced along its normal by bump(s, t)."""
    s, t = np.meshgrid(np.linspace(0, 1, nu + 1), np.linspace(0, 1, nv + 1), indexing="ij")
    n = np.cross(du, dv)
    n = n / np.linalg.norm(n)
    p = origin + s[..., None] * du + t[..., None] * dv
    nrm = np.broadcast_to(n, p.shape).copy()
    if bump is not None:
        h = bump(s, t)
        p = p + h[..., None] * n
        hs = np.gradient(h, axis=0) * nu / np.linalg.norm(du)
        ht = np.gradient(h, axis=1) * nv / np.linalg.norm(dv)
        nrm = n - hs[..., None] * du / np.linalg.norm(du) - ht[..., None] * dv / np.linalg.norm(dv)
        nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    uv = np.stack([s * uv_scale, t * uv_scale], -1)
    mesh.add(material, p.reshape(-1, 3), nrm.reshape(-1, 3), uv.reshape(-1, 2), grid_faces(nu, nv))


def revolve(mesh, material, base, profile_r, profile_y, nseg, flutes=0, flute_depth=0.0):
    """Surface of revolution about the y axis through `base`; optional fluting of the radius."""
    ang = np.linspace(0, 2 * np.pi, nseg + 1)
    r = np.asarray(profile_r)[None, :] * (1.0 - flute_depth * (0.5 + 0.5 * np.cos(flutes * ang))[:, None])
    y = np.asarray(profile_y)[None, :] + 0 * r
    x, z = r * np.cos(ang)[:, None], r * np.sin(ang)[:, None]
    p = np.stack([x, y, z], -1) + base
    dr = np.gradient(np.asarray(profile_r))
    dy = np.gradient(np.asarray(profile_y))
    nx = dy[None, :] * np.cos(ang)[:, None]
    nz = dy[None, :] * np.sin(ang)[:, None]
    ny = -dr[None, :] + 0 * nx
    nrm = np.stack([nx, ny, nz], -1)
    nrm /= np.maximum(np.linalg.norm(nrm, axis=-1, keepdims=True), 1e-9)
    uv = np.stack(np.meshgrid(np.linspace(0, 4, nseg + 1), np.linspace(0, 4, len(profile_r)), indexing="ij"), -1)
    mesh.add(material, p.reshape(-1, 3), nrm.reshape(-1, 3), uv.reshape(-1, 2), grid_faces(nseg, len(profile_r) - 1))


def sphere(mesh, material, centre, radius, nseg, nring, squash=1.0):
    th = np.linspace(1e-3, np.pi - 1e-3, nring + 1)
    revolve(mesh, material, np.asarray(centre) - [0, 0, 0], radius * np.sin(th), -radius * squash * np.cos(th), nseg)


def arch(mesh, material, c0, c1, y, radius, thickness, nseg, nring):
    """Half torus-like band spanning from c0 to c1 (points at height y), bulging upward."""
    c0, c1 = np.asarray(c0, float), np.asarray(c1, float)
    mid = 0.5 * (c0 + c1)
    half = 0.5 * np.linalg.norm(c1 - c0)
    axis = (c1 - c0) / (2 * half)
    side = np.cross(axis, [0, 1, 0])
    a, b = np.meshgrid(np.linspace(0, np.pi, nseg + 1), np.linspace(0, 2 * np.pi, nring + 1), indexing="ij")
    spine = mid + np.array([0, y, 0]) + (-np.cos(a))[..., None] * half * axis + (np.sin(a) * radius)[..., None] * np.array([0, 1, 0])
    out = (-np.cos(a))[..., None] * axis + np.sin(a)[..., None] * np.array([0, 1, 0])
    nrm = np.cos(b)[..., None] * out + np.sin(b)[..., None] * side
    p = spine + thickness * nrm
    uv = np.stack([a / np.pi * 3, b / np.pi], -1)
    mesh.add(material, p.reshape(-1, 3), nrm.reshape(-1, 3), uv.reshape(-1, 2), grid_faces(nseg, nring))


def noise_fn(rng, amp, freq):
    ph = rng.random((4, 3)) * 2 * np.pi
    k = rng.integers(1, 4, (4, 2)) * freq

    def f(s, t):
        h = 0 * s
        for i in range(4):
            h = h + np.sin(k[i, 0] * s * 2 * np.pi + ph[i, 0]) * np.sin(k[i, 1] * t * 2 * np.pi + ph[i, 1]) / (i + 1)
        return amp * h
    return f


MATERIALS = """newmtl floor
Ka 0.2 0.2 0.2
Kd 0.85 0.82 0.78
Ks 0 0 0
Ns 5
map_Kd sponza_floor.png

newmtl stone
Ka 0.2 0.2 0.2
Kd 0.72 0.66 0.55
Ks 0 0 0
Ns 5

newmtl brick
Ka 0.2 0.2 0.2
Kd 0.60 0.42 0.33
Ks 0 0 0
Ns 5

newmtl curtainred
Ka 0.1 0.0 0.0
Kd 0.70 0.10 0.08
Ks 0 0 0
Ns 5
Op 0.70

newmtl curtaingreen
Ka 0.0 0.1 0.0
Kd 0.12 0.55 0.15
Ks 0 0 0
Ns 5
Op 0.55

newmtl curtainblue
Ka 0.0 0.0 0.1
Kd 0.10 0.18 0.65
Ks 0 0 0
Ns 5

newmtl vase
Ka 0.1 0.1 0.1
Kd 0.55 0.45 0.20
Ks 0.9 0.8 0.5
Ns 100
Ro 0.30
Ri 1.80

newmtl glassorb
Ka 0 0 0
Kd 1 1 1
Ks 1 1 1
Ns 5

newmtl lamp
Ka 0.6 0.6 0.6
Kd 0.8 0.8 0.8
Ks 0 0 0
Ke 12.0 11.0 9.0
Ns 5
"""


def build(seed=7, detail=1.0):
    rng = np.random.default_rng(seed)
    m = Mesh()
    d = lambda n: max(2, int(round(n * detail)))  # noqa: E731
    L, W, H = 2.4, 1.0, 1.3            # half length (x), half width (z), height of the courtyard
    y0 = -0.5
    # floor and the four walls (displaced relief)
    patch(m, "floor", np.array([-L, y0, W]), np.array([2 * L, 0, 0]), np.array([0, 0, -2 * W]), d(160), d(72),
          noise_fn(rng, 0.004, 6), uv_scale=12.0)
    patch(m, "brick", np.array([-L, y0, -W]), np.array([2 * L, 0, 0]), np.array([0, H, 0]), d(160), d(56), noise_fn(rng, 0.012, 10))
    patch(m, "brick", np.array([L, y0, W]), np.array([-2 * L, 0, 0]), np.array([0, H, 0]), d(160), d(56), noise_fn(rng, 0.012, 10))
    patch(m, "stone", np.array([-L, y0, W]), np.array([0, 0, -2 * W]), np.array([0, H, 0]), d(72), d(56), noise_fn(rng, 0.010, 8))
    patch(m, "stone", np.array([L, y0, -W]), np.array([0, 0, 2 * W]), np.array([0, H, 0]), d(72), d(56), noise_fn(rng, 0.010, 8))
    # gallery slabs between the storeys
    for zs in (-1, 1):
        patch(m, "stone", np.array([-L, y0 + 0.62, zs * W]), np.array([2 * L, 0, 0]), np.array([0, 0, -zs * 0.32]), d(96), d(8),
              noise_fn(rng, 0.003, 5))
    # colonnades: two sides, two storeys
    xs = np.linspace(-L + 0.3, L - 0.3, 9)
    prof_y = np.concatenate([[0, 0.03, 0.05], np.linspace(0.06, 0.52, d(22)), [0.54, 0.57, 0.6]])
    prof_r = np.concatenate([[0.075, 0.075, 0.055], np.linspace(0.05, 0.042, d(22)), [0.05, 0.07, 0.07]])
    for zs in (-1, 1):
        z = zs * (W - 0.32)
        for storey, yb in enumerate((y0, y0 + 0.64)):
            sc = 1.0 if storey == 0 else 0.8
            for x in xs:
                revolve(m, "stone", np.array([x, yb, z]), prof_r * sc, prof_y * (1.0 if storey == 0 else 0.9), d(36), flutes=12,
                        flute_depth=0.12)
            for xa, xb in zip(xs[:-1], xs[1:]):
                arch(m, "stone", [xa, 0, z], [xb, 0, z], yb + (0.6 if storey == 0 else 0.54), 0.12 * sc, 0.035 * sc, d(28), d(12))
    # curtains hanging between some columns (wavy sheets)
    for k, (mat, xi) in enumerate((("curtainred", 1), ("curtaingreen", 3), ("curtainblue", 5), ("curtainred", 6))):
        zs = -1 if k % 2 == 0 else 1
        z = zs * (W - 0.36)
        wave = noise_fn(rng, 0.03, 3)
        patch(m, mat, np.array([xs[xi] + 0.06, y0 + 0.62, z]), np.array([xs[xi + 1] - xs[xi] - 0.12, 0, 0]), np.array([0, -0.5, 0]),
              d(40), d(40), wave)
    # vases, glass orbs, lamp
    for k, x in enumerate(np.linspace(-L + 0.6, L - 0.6, 4)):
        vy = np.linspace(0, 0.28, d(40))
        vr = 0.045 + 0.05 * np.sin(np.linspace(0.2, 2.9, d(40))) ** 2
        revolve(m, "vase", np.array([x, y0, 0.35 * (-1) ** k]), vr, vy, d(48))
        sphere(m, "glassorb", [x + 0.3, y0 + 0.09, -0.3 * (-1) ** k], 0.085, d(56), d(28))
    patch(m, "lamp", np.array([-0.5, y0 + H - 0.02, 0.25]), np.array([1.0, 0, 0]), np.array([0, 0, -0.5]), d(8), d(4))
    return m


def write(mesh, obj_path):
    base = os.path.splitext(os.path.basename(obj_path))[0]
    folder = os.path.dirname(os.path.abspath(obj_path))
    with open(os.path.join(folder, base + ".mtl"), "w") as f:
        f.write(MATERIALS)
    # small checker/noise texture for the floor
    from PIL import Image
    rng = np.random.default_rng(3)
    j, i = np.mgrid[0:64, 0:64]
    tex = 150 + 60 * ((i // 8 + j // 8) % 2) + rng.integers(-12, 12, (64, 64))
    Image.fromarray(np.clip(np.stack([tex, tex * 0.95, tex * 0.85], -1), 0, 255).astype(np.uint8)).save(
        os.path.join(folder, "sponza_floor.png"))
    v = np.concatenate(mesh.v)
    vn = np.concatenate(mesh.vn)
    vt = np.concatenate(mesh.vt)
    ntri = 0
    with open(obj_path, "w") as f:
        f.write("# procedural courtyard (stand-in for sponza.obj)\nmtllib %s.mtl\n" % base)
        f.write("".join("v %.6f %.6f %.6f\n" % tuple(p) for p in v))
        f.write("".join("vt %.5f %.5f\n" % tuple(p) for p in vt))
        f.write("".join("vn %.5f %.5f %.5f\n" % tuple(p) for p in vn))
        for material, faces in mesh.groups:
            f.write("usemtl %s\n" % material)
            g = faces + 1
            f.write("".join("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (a, a, a, b, b, b, c, c, c) for a, b, c in g))
            ntri += faces.shape[0]
    return ntri


# camera used by the benchmark: inside the courtyard, looking down its length
CAMERA = dict(eye=(-2.0, 0.05, 0.15), lookat=(1.0, -0.1, -0.05), up=(0.0, 1.0, 0.0))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out", help="output .obj path (the .mtl and the floor texture go next to it)")
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--detail", type=float, default=1.0)
    a = ap.parse_args()
    n = write(build(a.seed, a.detail), a.out)
    print("wrote %s with %d triangles" % (a.out, n))
