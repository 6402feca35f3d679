"""path_tracer_amd -- Python binding of libpt_amd.so, the MI355X path-tracing library.

The class below mirrors the reference's renderer interface (TriangleMesh::load, Scene::preCalc,
Scene::Pathtrace, Scene::intersect, src/cs488.h) on top of the C ABI in include/pt_amd.h.  It is
a thin ctypes layer: all rendering happens in HIP kernels inside the shared library, and importing
this package fails loudly when the library has not been built.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PT_AMD_LIB") or os.path.join(_HERE, "libpt_amd.so")  # the variable is for A/B measurements of builds

BVH, KD = 0, 1
LAMBERTIAN, METAL, GLASS = 0, 1, 2
DEVICE_NONE = -1

PT_OK = 0
ERR_INVALID_ARG, ERR_IO, ERR_INVALID_OBJ, ERR_NOT_READY, ERR_HIP, ERR_QUEUE_OVERFLOW, ERR_TREE_DEPTH, ERR_UNSUPPORTED = (
    -1, -2, -3, -4, -5, -6, -7, -8)
ERR_NO_MEMORY, ERR_RCCL = -9, -10

OPT_SEED, OPT_MAX_LEVEL, OPT_COUNT_TESTS, OPT_PROFILE, OPT_MAX_BATCH, OPT_QUEUE_FACTOR, OPT_LDS_NODES, OPT_TRAVERSAL, OPT_TAIL_LEVEL, OPT_BUILD_THREADS, OPT_OBJ_EXTENDED, OPT_BLOCKS_PER_CU, OPT_STREAMS, OPT_TREE_IN_LDS = (
    range(1, 15))
OPT_TAIL_LEVEL_USED = 15
CNT_RAYS, CNT_BOX_TESTS, CNT_TRI_TESTS, CNT_SAMPLES, CNT_NAN_SAMPLES, CNT_HITS, CNT_MAX_QUEUE, CNT_REPAIRED, CNT_TAIL_RAYS = range(9)
CNT_WAVE_STEPS, CNT_WAVE_ROUNDS, CNT_STREAM_SWAPS = 9, 10, 11
CNT_COUNT = 12
TIMER_RAYGEN, TIMER_TRACE, TIMER_SHADE, TIMER_ACCUM, TIMER_TAIL = range(5)
TIMER_COUNT = 5
TRACE_STATIC_MAX_RAYS = 1 << 17

# every symbol include/pt_amd.h declares
EXPORTS = [
    "pt_create", "pt_destroy", "pt_last_error", "pt_set_stream", "pt_set_option", "pt_get_option", "pt_wait_frames",
    "pt_load_obj", "pt_single_triangle", "pt_load_env", "pt_set_mesh", "pt_set_materials", "pt_set_texture",
    "pt_set_env", "pt_num_triangles", "pt_num_materials", "pt_get_mesh", "pt_get_material", "pt_get_texture",
    "pt_get_env", "pt_build", "pt_get_tree", "pt_set_camera", "pt_set_resolution", "pt_reset", "pt_render",
    "pt_render_split", "pt_resolve", "pt_total_samples", "pt_synchronize", "pt_download", "pt_accum_device_ptr",
    "pt_use_external_accum", "pt_save_png", "pt_save_pfm", "pt_trace_batch", "pt_camera_rays", "pt_get_counters",
    "pt_reset_counters", "pt_get_timers", "pt_trace_batch_static", "pt_camera_samples", "pt_group_create", "pt_group_destroy", "pt_group_last_error",
    "pt_group_size", "pt_group_render", "pt_group_reset", "pt_group_synchronize", "pt_group_wait_frames", "pt_group_set_cadence", "pt_group_stats", "pt_frames_in_flight", "pt_group_frames_in_flight", "pt_record_begin", "pt_record_frame", "pt_record_end", "pt_encode_gif",
]


class PtMaterial(C.Structure):
    _fields_ = [
        ("name", C.c_char * 96),
        ("type", C.c_int32),
        ("eta", C.c_float),
        ("Ka", C.c_float * 3),
        ("Kd", C.c_float * 3),
        ("Ks", C.c_float * 3),
        ("Ke", C.c_float * 3),
        ("Ns", C.c_float),
        ("opacity", C.c_float),
        ("roughness", C.c_float),
        ("textured", C.c_int32),
        ("tex_width", C.c_int32),
        ("tex_height", C.c_int32),
    ]


class PtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libpt_amd error %d: %s" % (code, message))
        self.code = code


_lib = None


def lib():
    """Loads libpt_amd.so; there is no fallback when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: build it with `make -C path_tracer_amd/csrc` "
                              "(or __graft_entry__.build())" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.pt_last_error.restype = C.c_char_p
        L.pt_accum_device_ptr.restype = C.c_void_p
        L.pt_destroy.restype = None
        L.pt_group_destroy.restype = None
        L.pt_group_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def encode_gif(path, frames, delay_cs=1):
    """Animated GIF from 8-bit RGB frames (n x height x width x 3, rows top-down) with the encoder of
    the recording entries; needs no GPU."""
    frames = np.ascontiguousarray(frames, np.uint8)
    if frames.ndim != 4 or frames.shape[3] != 3:
        raise ValueError("frames must be n x height x width x 3")
    n, h, w, _ = frames.shape
    rc = lib().pt_encode_gif(os.fsencode(path), frames.ctypes.data_as(C.POINTER(C.c_uint8)), n, w, h, int(delay_cs))
    if rc != 0:
        raise PtError(rc, "pt_encode_gif failed for %s" % path)


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class Scene:
    """One renderer context: mesh + materials + environment + accelerator + camera + film."""

    def __init__(self, device=0):
        self.L = lib()
        self.h = C.c_void_p()
        rc = self.L.pt_create(int(device), C.byref(self.h))
        if rc != PT_OK:
            raise PtError(rc, "pt_create(%d) failed: no usable GPU (this library has no CPU path)" % device)
        self.width = self.height = 0

    # ---- plumbing ----
    def _check(self, rc):
        if rc < 0:
            raise PtError(rc, (self.L.pt_last_error(self.h) or b"").decode(errors="replace"))
        return rc

    def close(self):
        group = getattr(self, "_group", None)
        if group is not None:  # a member of a group is destroyed after the group
            group.close()
        if getattr(self, "h", None):
            self.L.pt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, opt, value):
        self._check(self.L.pt_set_option(self.h, opt, C.c_uint64(int(value))))

    def get_option(self, opt):
        v = C.c_uint64()
        self._check(self.L.pt_get_option(self.h, opt, C.byref(v)))
        return v.value

    def set_stream(self, hip_stream):
        self._check(self.L.pt_set_stream(self.h, C.c_void_p(hip_stream)))

    # ---- scene ----
    def load(self, obj_path):
        """TriangleMesh::load; returns False (and leaves the scene empty) like the reference on a bad file."""
        rc = self.L.pt_load_obj(self.h, os.fsencode(obj_path))
        if rc in (ERR_INVALID_OBJ, ERR_IO):
            return False
        self._check(rc)
        return True

    def create_single_triangle(self):
        self._check(self.L.pt_single_triangle(self.h))

    def load_env(self, hdr_path):
        self._check(self.L.pt_load_env(self.h, os.fsencode(hdr_path)))

    def set_mesh(self, pos, nrm, uv=None, mat=None):
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 9)
        nrm = np.ascontiguousarray(nrm, np.float32).reshape(-1, 9)
        uvp = _fp(np.ascontiguousarray(uv, np.float32)) if uv is not None else None
        matp = _ip(np.ascontiguousarray(mat, np.int32)) if mat is not None else None
        self._check(self.L.pt_set_mesh(self.h, pos.shape[0], _fp(pos), _fp(nrm), uvp, matp))

    def set_materials(self, mats):
        arr = (PtMaterial * len(mats))(*mats)
        self._check(self.L.pt_set_materials(self.h, len(mats), arr))

    def set_texture(self, material, rgb_u8):
        rgb_u8 = np.ascontiguousarray(rgb_u8, np.uint8)
        hh, ww = rgb_u8.shape[:2]
        self._check(self.L.pt_set_texture(self.h, material, rgb_u8.ctypes.data_as(C.POINTER(C.c_uint8)), ww, hh))

    def set_env(self, rgb_f32):
        if rgb_f32 is None:
            self._check(self.L.pt_set_env(self.h, None, 0, 0))
            return
        rgb_f32 = np.ascontiguousarray(rgb_f32, np.float32)
        hh, ww = rgb_f32.shape[:2]
        self._check(self.L.pt_set_env(self.h, _fp(rgb_f32), ww, hh))

    def num_triangles(self):
        return self._check(self.L.pt_num_triangles(self.h))

    def num_materials(self):
        return self._check(self.L.pt_num_materials(self.h))

    def get_mesh(self):
        n = self.num_triangles()
        pos = np.zeros((n, 9), np.float32)
        nrm = np.zeros((n, 9), np.float32)
        uv = np.zeros((n, 6), np.float32)
        mat = np.zeros(n, np.int32)
        self._check(self.L.pt_get_mesh(self.h, _fp(pos), _fp(nrm), _fp(uv), _ip(mat)))
        return pos, nrm, uv, mat

    def get_material(self, i):
        m = PtMaterial()
        self._check(self.L.pt_get_material(self.h, i, C.byref(m)))
        return m

    def materials(self):
        return [self.get_material(i) for i in range(self.num_materials())]

    def get_texture(self, i):
        m = self.get_material(i)
        out = np.zeros((m.tex_height, m.tex_width, 3), np.uint8)
        self._check(self.L.pt_get_texture(self.h, i, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def get_env(self):
        w, h = C.c_int(), C.c_int()
        self._check(self.L.pt_get_env(self.h, None, C.byref(w), C.byref(h)))
        out = np.zeros((h.value, w.value, 3), np.float32)
        if out.size:
            self._check(self.L.pt_get_env(self.h, _fp(out), None, None))
        return out

    # ---- accelerator ----
    def pre_calc(self, accel=KD):
        """Scene::preCalc: build the accelerator on the CPU and upload everything."""
        self._check(self.L.pt_build(self.h, int(accel)))

    def get_tree(self):
        n, r, d = C.c_int(), C.c_int(), C.c_int()
        self._check(self.L.pt_get_tree(self.h, C.byref(n), C.byref(r), C.byref(d), None, None, None, None, None, None))
        out = dict(left=np.zeros(n.value, np.int32), right=np.zeros(n.value, np.int32), first=np.zeros(n.value, np.int32),
                   count=np.zeros(n.value, np.int32), box=np.zeros((n.value, 6), np.float32),
                   refs=np.zeros(max(r.value, 1), np.int32), max_depth=d.value)
        self._check(self.L.pt_get_tree(self.h, None, None, None, _ip(out["left"]), _ip(out["right"]), _ip(out["first"]),
                                       _ip(out["count"]), _fp(out["box"]), _ip(out["refs"])))
        out["refs"] = out["refs"][:r.value]
        return out

    # ---- camera and film ----
    def set_camera(self, eye=(0, 0, 1.5), lookat=(0, 0, 0), up=(0, 1, 0), fov=45.0, film=0.032):
        e = (C.c_float * 3)(*eye)
        l = (C.c_float * 3)(*lookat)
        u = (C.c_float * 3)(*up)
        self._check(self.L.pt_set_camera(self.h, e, l, u, C.c_float(fov), C.c_float(film)))

    def set_resolution(self, width=512, height=384):
        self._check(self.L.pt_set_resolution(self.h, width, height))
        self.width, self.height = width, height

    def reset(self):
        self._check(self.L.pt_reset(self.h))

    # ---- hot path ----
    def pathtrace(self, spp=10):
        """Scene::Pathtrace: one progressive frame of `spp` samples per pixel (asynchronous)."""
        self._check(self.L.pt_render(self.h, int(spp)))

    def pathtrace_split(self, spp_frame, first, count):
        self._check(self.L.pt_render_split(self.h, int(spp_frame), int(first), int(count)))

    def resolve(self):
        self._check(self.L.pt_resolve(self.h))

    def total_samples(self):
        return self._check(self.L.pt_total_samples(self.h))

    def synchronize(self):
        self._check(self.L.pt_synchronize(self.h))

    def _image(self, which):
        img = np.zeros((self.height, self.width, 3), np.float32)
        self._check(self.L.pt_download(self.h, which, _fp(img)))
        return img

    def accumulation_buffer(self):
        return self._image(0)

    def frame_buffer(self):
        return self._image(1)

    def accum_device_ptr(self):
        return self.L.pt_accum_device_ptr(self.h)

    def wait_frames(self, keep=0):
        """Blocks until at most `keep` of the frames queued by pathtrace / pathtrace_split are unfinished."""
        self._check(self.L.pt_wait_frames(self.h, int(keep)))

    def frames_in_flight(self):
        """Queued frames that have not ended on the GPU yet (pt_frames_in_flight); never waits."""
        return self._check(self.L.pt_frames_in_flight(self.h))

    def use_external_accum(self, device_ptr):
        self._check(self.L.pt_use_external_accum(self.h, C.c_void_p(device_ptr)))

    def save_png(self, path, which=1):
        self._check(self.L.pt_save_png(self.h, which, os.fsencode(path)))

    def save_pfm(self, path, which=1):
        self._check(self.L.pt_save_pfm(self.h, which, os.fsencode(path)))

    # ---- recording (the window's F key, src/cs488.h:255-266, 2297-2310) ----
    def record_begin(self, path, delay_cs=1):
        self._check(self.L.pt_record_begin(self.h, os.fsencode(path), int(delay_cs)))

    def record_frame(self):
        self._check(self.L.pt_record_frame(self.h))

    def record_end(self):
        self._check(self.L.pt_record_end(self.h))

    # ---- parity entry ----
    def intersect(self, rays):
        """Scene::intersect for a batch of rays (n x 6: origin, direction)."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        n = rays.shape[0]
        tri = np.zeros(n, np.int32)
        t = np.zeros(n, np.float32)
        bary = np.zeros((n, 2), np.float32)
        self._check(self.L.pt_trace_batch(self.h, n, _fp(rays), _ip(tri), _fp(t), _fp(bary)))
        return dict(tri=tri, t=t, bary=bary)

    def intersect_static(self, d_rays, d_tri, d_t, d_bary=None, n=None):
        """pt_trace_batch_static: device addresses (ints) of n x 6 float rays and of the outputs; asynchronous
        on the context's stream and free of allocations."""
        self._check(self.L.pt_trace_batch_static(self.h, int(n), C.c_void_p(d_rays), C.c_void_p(d_tri), C.c_void_p(d_t),
                                                 C.c_void_p(d_bary) if d_bary else None))

    def camera_rays(self):
        rays = np.zeros((self.width * self.height, 6), np.float32)
        self._check(self.L.pt_camera_rays(self.h, _fp(rays)))
        return rays

    def camera_samples(self, sample=0):
        """The jittered camera rays raygen_kernel generates for one global sample index (W*H x 6, pixel order)."""
        rays = np.zeros((self.width * self.height, 6), np.float32)
        self._check(self.L.pt_camera_samples(self.h, int(sample), _fp(rays)))
        return rays

    # ---- measurement ----
    def counters(self):
        out = (C.c_uint64 * CNT_COUNT)()
        self._check(self.L.pt_get_counters(self.h, out))
        return [int(x) for x in out]

    def reset_counters(self):
        self._check(self.L.pt_reset_counters(self.h))

    def timers(self):
        ms = (C.c_double * TIMER_COUNT)()
        n = (C.c_uint64 * TIMER_COUNT)()
        self._check(self.L.pt_get_timers(self.h, ms, n))
        return [float(x) for x in ms], [int(x) for x in n]


class Group:
    """Several Scenes of this process, one per GPU, rendering one image: every frame's samples are split over
    the members and summed onto scenes[0] with one RCCL reduce (pt_group_*).  Set the members up alike first."""

    def __init__(self, scenes):
        self.L = lib()
        self.scenes = list(scenes)
        arr = (C.c_void_p * len(self.scenes))(*[s.h for s in self.scenes])
        self.h = C.c_void_p()
        rc = self.L.pt_group_create(len(self.scenes), arr, C.byref(self.h))
        if rc != PT_OK:
            raise PtError(rc, (self.L.pt_last_error(self.scenes[0].h) or b"").decode(errors="replace") if self.scenes else "empty group")
        for s in self.scenes:
            s._group = self

    def _check(self, rc):
        if rc < 0:
            raise PtError(rc, (self.L.pt_group_last_error(self.h) or b"").decode(errors="replace"))
        return rc

    def pathtrace(self, spp=10):
        self._check(self.L.pt_group_render(self.h, int(spp)))

    def reset(self):
        self._check(self.L.pt_group_reset(self.h))

    def synchronize(self):
        self._check(self.L.pt_group_synchronize(self.h))

    def wait_frames(self, keep=0):
        """Returns once at most `keep` queued frames are unfinished on any member (pt_group_wait_frames)."""
        self._check(self.L.pt_group_wait_frames(self.h, int(keep)))

    def set_cadence(self, frames_per_reduce):
        """Frames added into one per-frame buffer before it is reduced (pt_group_set_cadence)."""
        self._check(self.L.pt_group_set_cadence(self.h, int(frames_per_reduce)))

    def stats(self):
        """(frames queued, reduces issued, host threads queueing the members' launches)."""
        f, r, t = C.c_uint64(), C.c_uint64(), C.c_int()
        self._check(self.L.pt_group_stats(self.h, C.byref(f), C.byref(r), C.byref(t)))
        return f.value, r.value, t.value

    def close(self):
        if getattr(self, "h", None):
            self.L.pt_group_destroy(self.h)
            self.h = None
            for s in self.scenes:
                s._group = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
