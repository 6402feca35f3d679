"""ctypes binding of the CPU oracle (oracle/libpt_oracle.so).

Test infrastructure only: imported by tests/, the smoke check and bench.py's cpu_baseline leg,
never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(ROOT, "oracle", "libpt_oracle.so")

BVH, KD = 0, 1
LAMBERTIAN, METAL, GLASS = 0, 1, 2
DEFAULT_SEED = 0xCAFEF00DD15EA5E5
N_COUNTERS = 24


class OrcMaterial(C.Structure):
    _fields_ = [
        ("name", C.c_char * 96),
        ("type", C.c_int),
        ("eta", C.c_float),
        ("Ka", C.c_float * 3),
        ("Kd", C.c_float * 3),
        ("Ks", C.c_float * 3),
        ("Ke", C.c_float * 3),
        ("Ns", C.c_float),
        ("opacity", C.c_float),
        ("roughness", C.c_float),
        ("textured", C.c_int),
        ("tex_file", C.c_char * 512),
    ]


def build_oracle():
    """Compile the oracle if the shared object is missing or older than its source."""
    src = os.path.join(ROOT, "oracle", "pt_oracle.cpp")
    if not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build_oracle()
        L = C.CDLL(LIB_PATH)
        L.orc_create.restype = C.c_void_p
        for name in ("orc_destroy", "orc_single_triangle", "orc_set_tris", "orc_get_tris", "orc_get_material",
                     "orc_set_material", "orc_set_env", "orc_set_camera", "orc_set_max_level", "orc_build",
                     "orc_bvh_dump", "orc_kd_dump", "orc_kd_import", "orc_trace", "orc_camera_rays", "orc_radiance", "orc_ibl",
                     "orc_reset", "orc_get_image", "orc_pcg32"):
            getattr(L, name).restype = None
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _u64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


class Oracle:
    """One oracle scene: mesh + materials + accelerator + camera + progressive buffers."""

    def __init__(self):
        self.L = lib()
        self.h = C.c_void_p(self.L.orc_create())
        self.width, self.height = 512, 384

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- scene ----
    def load_obj(self, path):
        return self.L.orc_load_obj(self.h, os.fsencode(path))

    def single_triangle(self):
        self.L.orc_single_triangle(self.h)

    def set_tris(self, pos, nrm, uv=None, mat=None):
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 9)
        nrm = np.ascontiguousarray(nrm, np.float32).reshape(-1, 9)
        n = pos.shape[0]
        uvp = _fp(np.ascontiguousarray(uv, np.float32)) if uv is not None else None
        matp = _ip(np.ascontiguousarray(mat, np.int32)) if mat is not None else None
        self.L.orc_set_tris(self.h, n, _fp(pos), _fp(nrm), uvp, matp)

    def num_tris(self):
        return self.L.orc_num_tris(self.h)

    def num_materials(self):
        return self.L.orc_num_materials(self.h)

    def get_tris(self):
        n = self.num_tris()
        pos = np.zeros((n, 9), np.float32)
        nrm = np.zeros((n, 9), np.float32)
        uv = np.zeros((n, 6), np.float32)
        mat = np.zeros(n, np.int32)
        self.L.orc_get_tris(self.h, _fp(pos), _fp(nrm), _fp(uv), _ip(mat))
        return pos, nrm, uv, mat

    def get_material(self, i):
        m = OrcMaterial()
        self.L.orc_get_material(self.h, i, C.byref(m))
        return m

    def materials(self):
        return [self.get_material(i) for i in range(self.num_materials())]

    def set_material(self, i, **kw):
        m = self.get_material(i) if i < self.num_materials() else OrcMaterial(
            eta=1.0, Kd=(C.c_float * 3)(0.9, 0.9, 0.9), opacity=1.0, roughness=0.6)
        for k, v in kw.items():
            if k in ("Ka", "Kd", "Ks", "Ke"):
                v = (C.c_float * 3)(*v)
            if k == "name":
                v = v if isinstance(v, bytes) else v.encode()
            setattr(m, k, v)
        self.L.orc_set_material(self.h, i, C.byref(m))

    def set_texture(self, mat, rgb_u8):
        rgb_u8 = np.ascontiguousarray(rgb_u8, np.uint8)
        hh, ww = rgb_u8.shape[:2]
        return self.L.orc_set_texture(self.h, mat, rgb_u8.ctypes.data_as(C.POINTER(C.c_uint8)), ww, hh)

    def set_env(self, rgb_f32):
        rgb_f32 = np.ascontiguousarray(rgb_f32, np.float32)
        hh, ww = rgb_f32.shape[:2]
        self.L.orc_set_env(self.h, _fp(rgb_f32), ww, hh)

    def set_camera(self, eye=(0, 0, 1.5), lookat=(0, 0, 0), up=(0, 1, 0), fov=45.0, film=0.032):
        e = np.asarray(eye, np.float32)
        l = np.asarray(lookat, np.float32)
        u = np.asarray(up, np.float32)
        self.L.orc_set_camera(self.h, _fp(e), _fp(l), _fp(u), C.c_float(fov), C.c_float(film))

    def set_max_level(self, lvl):
        self.L.orc_set_max_level(self.h, int(lvl))

    def build(self, accel=BVH):
        self.L.orc_build(self.h, int(accel))

    # ---- accelerator dumps ----
    def import_kd(self, tree):
        """Uses a kd-tree built elsewhere (the dict Scene.get_tree() returns: the renderer's host builder) in place
        of this oracle's quadratic builder; for meshes too large for the latter."""
        n = len(tree["left"])
        arrs = [np.ascontiguousarray(tree[k], np.int32) for k in ("left", "right", "first", "count")]
        box = np.ascontiguousarray(tree["box"], np.float32)
        refs = np.ascontiguousarray(tree["refs"] if len(tree["refs"]) else np.zeros(1), np.int32)
        self.L.orc_kd_import(self.h, n, *[_ip(a) for a in arrs], _fp(box), _ip(refs))

    def bvh_dump(self):
        n = self.L.orc_bvh_num_nodes(self.h)
        r = self.L.orc_bvh_num_refs(self.h)
        d = dict(is_leaf=np.zeros(n, np.int32), left=np.zeros(n, np.int32), right=np.zeros(n, np.int32),
                 first=np.zeros(n, np.int32), count=np.zeros(n, np.int32), box=np.zeros((n, 6), np.float32),
                 refs=np.zeros(max(r, 1), np.int32))
        self.L.orc_bvh_dump(self.h, _ip(d["is_leaf"]), _ip(d["left"]), _ip(d["right"]), _ip(d["first"]),
                            _ip(d["count"]), _fp(d["box"]), _ip(d["refs"]))
        d["refs"] = d["refs"][:r]
        return d

    def kd_dump(self):
        n = self.L.orc_kd_num_nodes(self.h)
        r = self.L.orc_kd_num_refs(self.h)
        d = dict(left=np.zeros(n, np.int32), right=np.zeros(n, np.int32), first=np.zeros(n, np.int32),
                 count=np.zeros(n, np.int32), box=np.zeros((n, 6), np.float32), refs=np.zeros(max(r, 1), np.int32))
        self.L.orc_kd_dump(self.h, _ip(d["left"]), _ip(d["right"]), _ip(d["first"]), _ip(d["count"]),
                           _fp(d["box"]), _ip(d["refs"]))
        d["refs"] = d["refs"][:r]
        return d

    # ---- queries ----
    def trace(self, rays, want_geometry=False):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        n = rays.shape[0]
        tri = np.zeros(n, np.int32)
        t = np.zeros(n, np.float32)
        bary = np.zeros((n, 2), np.float32)
        cnt = np.zeros(N_COUNTERS, np.uint64)
        if want_geometry:
            P = np.zeros((n, 3), np.float32)
            N = np.zeros((n, 3), np.float32)
            bf = np.zeros(n, np.int32)
            self.L.orc_trace(self.h, n, _fp(rays), _ip(tri), _fp(t), _fp(bary), _fp(P), _fp(N), _ip(bf), _u64p(cnt))
            return dict(tri=tri, t=t, bary=bary, P=P, N=N, backface=bf, counters=cnt)
        self.L.orc_trace(self.h, n, _fp(rays), _ip(tri), _fp(t), _fp(bary), None, None, None, _u64p(cnt))
        return dict(tri=tri, t=t, bary=bary, counters=cnt)

    def tie_set(self, ray, t_ref, cap=64):
        ray = np.ascontiguousarray(ray, np.float32).reshape(6)
        out = np.zeros(cap, np.int32)
        n = self.L.orc_tie_set(self.h, _fp(ray), C.c_float(t_ref), _ip(out), cap)
        return out[:min(n, cap)]

    def camera_rays(self, width, height, jitter=False, state=DEFAULT_SEED):
        st = np.array([state], np.uint64)
        rays = np.zeros((height * width, 6), np.float32)
        self.L.orc_camera_rays(self.h, width, height, 1 if jitter else 0, _u64p(st), _fp(rays))
        return rays, int(st[0])

    def radiance(self, rays, state=DEFAULT_SEED):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        st = np.array([state], np.uint64)
        rgb = np.zeros((rays.shape[0], 3), np.float32)
        self.L.orc_radiance(self.h, rays.shape[0], _fp(rays), _u64p(st), _fp(rgb))
        return rgb, int(st[0])

    def ibl(self, dirs):
        dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        rgb = np.zeros_like(dirs)
        self.L.orc_ibl(self.h, dirs.shape[0], _fp(dirs), _fp(rgb))
        return rgb

    # ---- progressive rendering ----
    def reset(self, width=512, height=384):
        self.width, self.height = width, height
        self.L.orc_reset(self.h, width, height)

    def render(self, spp=10, frames=1, threads=1, state=DEFAULT_SEED):
        st = np.array([state], np.uint64)
        cnt = np.zeros(N_COUNTERS, np.uint64)
        total = self.L.orc_render(self.h, spp, frames, threads, _u64p(st), _u64p(cnt))
        return total, int(st[0]), cnt

    def image(self, which="frame"):
        img = np.zeros((self.height, self.width, 3), np.float32)
        self.L.orc_get_image(self.h, 0 if which == "accum" else 1, _fp(img))
        return img


def pcg32(n, state=DEFAULT_SEED):
    st = np.array([state], np.uint64)
    u = np.zeros(n, np.uint32)
    f = np.zeros(n, np.float32)
    lib().orc_pcg32(_u64p(st), n, u.ctypes.data_as(C.POINTER(C.c_uint32)), _fp(f))
    return u, f, int(st[0])
