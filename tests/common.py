"""Shared helpers of the test-suite: scene paths, paired oracle/GPU scene set-up, ray batches."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = os.path.join(ROOT, "tests", "golden", "scenes")
GOLDEN = os.path.join(ROOT, "tests", "golden")


# Scene files of the reference that differ from another one of its files only in the material lines named here
# (verified with diff against /root/reference/media when the fixtures were taken): written next to the base file
# on first use instead of being stored a second time.
DERIVED_SCENES = {
    "teapot-glass.obj": ("teapot.obj", {"mtllib teapot.mtl": "mtllib teapot-glass.mtl", "usemtl default": "usemtl glass"}),
    "teapot-metal.obj": ("teapot.obj", {"mtllib teapot.mtl": "mtllib teapot-metal.mtl"}),
    "testObj-metal.obj": ("testObj.obj", {"mtllib testObj.mtl": "mtllib testObj-metal.mtl"}),
}


def scene_path(name):
    path = os.path.join(SCENES, name)
    if name in DERIVED_SCENES and not os.path.exists(path):
        base, swaps = DERIVED_SCENES[name]
        lines = open(os.path.join(SCENES, base), "rb").read().split(b"\n")
        found = 0
        for i, ln in enumerate(lines):
            key = ln.rstrip(b"\r").decode("latin-1")
            if key in swaps:
                lines[i] = swaps[key].encode()
                found += 1
        assert found == len(swaps), (name, found)
        tmp = "%s.%d.tmp" % (path, os.getpid())
        with open(tmp, "wb") as f:
            f.write(b"\n".join(lines))
        os.replace(tmp, path)
    return path


def read_hdr_independent(path):
    """Flat or RLE Radiance file -> float32 (h, w, 3), first row first; numpy-only decoder."""
    from make_probe import decode_rgbe
    data = open(path, "rb").read()
    head, _, rest = data.partition(b"\n\n")
    line, _, body = rest.partition(b"\n")
    parts = line.split()
    h, w = int(parts[1]), int(parts[3])
    if len(body) == w * h * 4:
        rgbe = np.frombuffer(body, np.uint8).reshape(h, w, 4)
    else:
        rgbe = np.zeros((h, w, 4), np.uint8)
        pos = 0
        for y in range(h):
            assert body[pos] == 2 and body[pos + 1] == 2 and (body[pos + 2] << 8 | body[pos + 3]) == w
            pos += 4
            for c in range(4):
                x = 0
                while x < w:
                    n = body[pos]; pos += 1
                    if n > 128:
                        n -= 128
                        rgbe[y, x:x + n, c] = body[pos]; pos += 1
                    else:
                        rgbe[y, x:x + n, c] = np.frombuffer(body[pos:pos + n], np.uint8); pos += n
                    x += n
    return decode_rgbe(rgbe)


def load_oracle_scene(obj, accel, env=None, textures=True):
    """Oracle scene for an .obj under tests/golden/scenes (textures decoded with PIL)."""
    import orc
    o = orc.Oracle()
    n = o.load_obj(scene_path(obj))
    assert n > 0, obj
    if textures:
        from PIL import Image
        for i, m in enumerate(o.materials()):
            if m.textured:
                img = np.asarray(Image.open(m.tex_file.decode()).convert("RGB"), np.uint8)
                o.set_texture(i, img)
    if env is not None:
        o.set_env(read_hdr_independent(scene_path(env)))
    o.build(accel)
    return o


def load_gpu_scene(obj, accel, env=None, device=0):
    import path_tracer_amd as pta
    s = pta.Scene(device)
    assert s.load(scene_path(obj)), obj
    if env is not None:
        s.load_env(scene_path(env))
    s.pre_calc(accel)
    return s


def mesh_bounds(pos):
    p = pos.reshape(-1, 3)
    return p.min(0), p.max(0)


def ray_batch(pos, n_random=2000, n_vertex=300, seed=3, eye=(0.0, 0.0, 1.5)):
    """Ray batch for closest-hit parity: incoherent random rays through the scene's bounds, rays
    from the eye aimed exactly at mesh vertices and edge midpoints, rays with zero direction
    components, and rays starting on geometry (origins on triangle planes)."""
    rng = np.random.default_rng(seed)
    lo, hi = mesh_bounds(pos)
    c, ext = (lo + hi) / 2, (hi - lo).max()
    rays = []
    o = c + (rng.random((n_random, 3)) - 0.5) * ext * 2.2
    tgt = lo + rng.random((n_random, 3)) * (hi - lo)
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays.append(np.concatenate([o, d], 1))
    tris = pos.reshape(-1, 3, 3)
    pick = rng.integers(0, tris.shape[0], n_vertex)
    verts = tris[pick, rng.integers(0, 3, n_vertex)]
    mids = (tris[pick, 0] + tris[pick, 1]) * 0.5
    e = np.asarray(eye, np.float64)
    for tg in (verts, mids):
        d = tg - e
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        rays.append(np.concatenate([np.broadcast_to(e, d.shape), d], 1))
    # axis-parallel rays (zero direction components) through random points of the bounds
    m = 120
    for axis in range(3):
        o = lo + rng.random((m, 3)) * (hi - lo)
        o[:, axis] = lo[axis] - 0.3 * ext
        d = np.zeros((m, 3)); d[:, axis] = 1.0
        rays.append(np.concatenate([o, d], 1))
        o2 = o.copy(); o2[:, axis] = hi[axis] + 0.3 * ext
        rays.append(np.concatenate([o2, -d], 1))
    # bounce-like rays: origin on a triangle (its centroid), random direction
    pick = rng.integers(0, tris.shape[0], n_vertex)
    o = tris[pick].mean(1)
    d = rng.standard_normal((n_vertex, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays.append(np.concatenate([o, d], 1))
    return np.concatenate(rays, 0).astype(np.float32)


def rmse(a, b):
    return float(np.sqrt(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2)))


class DeviceArray:
    """A device allocation of the HIP runtime the library itself uses (no PyTorch: its own copy of the
    runtime cannot open the GPU once the system one has), for tests that hand device pointers to the C ABI."""
    _hip = None

    @classmethod
    def hip(cls):
        if cls._hip is None:
            import ctypes
            import path_tracer_amd as pta
            pta.lib()  # loads libamdhip64 as a dependency; the name below then resolves to that same copy
            cls._hip = ctypes.CDLL("libamdhip64.so")
        return cls._hip

    def __init__(self, array):
        import ctypes
        self.host = np.ascontiguousarray(array)
        self.ptr = ctypes.c_void_p()
        assert self.hip().hipMalloc(ctypes.byref(self.ptr), ctypes.c_size_t(max(self.host.nbytes, 4))) == 0
        assert self.hip().hipMemcpy(self.ptr, self.host.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(self.host.nbytes), 1) == 0

    def download(self):
        import ctypes
        out = np.empty_like(self.host)
        assert self.hip().hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), self.ptr, ctypes.c_size_t(out.nbytes), 2) == 0
        return out

    def free(self):
        if self.ptr:
            self.hip().hipFree(self.ptr)
            self.ptr = None
