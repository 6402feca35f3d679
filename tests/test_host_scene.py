"""Host side of the product library without a GPU: .obj/.mtl loader and the two accelerator
builders, compared with the oracle's independent restatement of the reference (bit for bit), and
the loader's documented behaviours."""
import os

import numpy as np
import pytest

import orc
import path_tracer_amd as pta
from common import DERIVED_SCENES, SCENES, scene_path

# the stored scene files, and those written from them on first use (same bytes as the reference's own files)
OBJS = sorted(set(f for f in os.listdir(SCENES) if f.endswith(".obj")) | set(DERIVED_SCENES))
# meshes whose tree comparisons are run one by one below: the oracle's kd builder recounts every triangle for every
# candidate plane like the reference's (quadratic per node), which takes seconds at 12 k triangles and minutes at 69 k
LARGE = {"teapot.obj", "teapot-glass.obj", "teapot-metal.obj", "testObj.obj", "testObj-glass.obj", "testObj-metal.obj", "bunny.obj"}


def bits(a):
    return a.view(np.uint32) if a.dtype == np.float32 else a


def host_scene(obj=None):
    s = pta.Scene(pta.DEVICE_NONE)
    if obj:
        assert s.load(scene_path(obj))
    return s


@pytest.mark.parametrize("obj", OBJS)
def test_loader_matches_oracle(obj):
    o = orc.Oracle()
    n = o.load_obj(scene_path(obj))
    s = host_scene(obj)
    assert n == s.num_triangles() > 0
    for a, b in zip(o.get_tris(), s.get_mesh()):
        assert np.array_equal(bits(a), bits(b))
    assert o.num_materials() == s.num_materials()
    for ma, mb in zip(o.materials(), s.materials()):
        for f in ("name", "type", "eta", "Ns", "opacity", "roughness", "textured"):
            assert getattr(ma, f) == getattr(mb, f), f
        for f in ("Ka", "Kd", "Ks", "Ke"):
            assert list(getattr(ma, f)) == list(getattr(mb, f)), f


@pytest.mark.parametrize("obj", [o for o in OBJS if o not in LARGE])
@pytest.mark.parametrize("accel", [orc.BVH, orc.KD])
def test_builders_match_oracle(obj, accel):
    compare_trees(obj, accel)


@pytest.mark.parametrize("obj,accel", [("testObj.obj", orc.BVH), ("testObj.obj", orc.KD), ("testObj-glass.obj", orc.BVH),
                                       ("testObj-glass.obj", orc.KD), ("bunny.obj", orc.BVH)])
def test_builders_match_oracle_on_the_larger_meshes(obj, accel):
    """testObj (11 950 and 10 846 triangles, five materials) both ways; the 69 451-triangle bunny (no mtllib, no
    normals) for the BVH only, its kd-tree being out of the quadratic builder's reach."""
    t = compare_trees(obj, accel)
    if obj == "bunny.obj":
        assert len(t["left"]) > 30000


def test_teapot_bvh_matches_oracle():
    compare_trees("teapot.obj", orc.BVH)


def test_teapot_kd_sweep_builder_matches_quadratic_oracle():
    """The product finds each kd split by binary search in sorted extents; the oracle recounts
    every triangle for every candidate like the reference.  Same tree, node for node."""
    t = compare_trees("teapot.obj", orc.KD)
    assert len(t["left"]) == 126015 and t["max_depth"] == 33


def compare_trees(obj, accel):
    o = orc.Oracle()
    o.load_obj(scene_path(obj))
    o.build(accel)
    s = host_scene(obj)
    s.pre_calc(accel)
    t = s.get_tree()
    if accel == orc.BVH:
        d = o.bvh_dump()
        leaf = d["is_leaf"] == 1
        assert np.array_equal(np.where(leaf, -1, d["left"]), t["left"])
        assert np.array_equal(np.where(leaf, -1, d["right"]), t["right"])
    else:
        d = o.kd_dump()
        assert np.array_equal(d["left"], t["left"]) and np.array_equal(d["right"], t["right"])
    assert np.array_equal(d["count"], t["count"])
    assert np.array_equal(bits(d["box"]), bits(t["box"]))
    for i in np.nonzero(d["count"])[0]:
        assert np.array_equal(d["refs"][d["first"][i]:d["first"][i] + d["count"][i]],
                              t["refs"][t["first"][i]:t["first"][i] + t["count"][i]]), i
    # depth = longest root-to-leaf path
    depth = np.zeros(len(t["left"]), np.int32)
    for i in range(len(t["left"])):
        for c in (t["left"][i], t["right"][i]):
            if c >= 0:
                depth[c] = depth[i] + 1
    assert depth.max() == t["max_depth"]
    return t


def test_material_rules():
    s = host_scene("cornellbox-glass.obj")
    m = {x.name: x for x in s.materials()}
    assert b"glass\n" in m, "names keep their newline"
    g = m[b"glass\n"]
    assert g.type == pta.GLASS and g.eta == 1.5 and g.Ns == 100.0  # the name beats Ns == 100
    s = host_scene("cornellbox-metal.obj")
    mt = {x.name: x for x in s.materials()}[b"metal\n"]
    assert mt.type == pta.METAL and abs(mt.roughness - 0.7) < 1e-7 and abs(mt.eta - 1.2) < 1e-7
    s = host_scene("cornellbox-alpha.obj")
    a = {x.name: x for x in s.materials()}
    assert abs(a[b"metalred\n"].opacity - 0.2) < 1e-7 and abs(a[b"metalblue\n"].opacity - 0.4) < 1e-7
    assert a[b"metalred\n"].type == pta.LAMBERTIAN
    assert list(a[b"light\n"].Ke) == [7.5, 7.5, 7.5]


def test_mesh_without_mtl_gets_one_default_material():
    s = host_scene("smallsphere.obj")
    assert s.num_materials() == 1
    m = s.get_material(0)
    assert m.type == pta.LAMBERTIAN and abs(m.Kd[0] - 0.9) < 1e-7 and m.opacity == 1.0 and abs(m.roughness - 0.6) < 1e-7
    assert not s.get_mesh()[3].any()


def test_missing_normals_give_face_normals_and_texcoords_need_a_texture():
    s = host_scene("cornellbox.obj")
    pos, nrm, uv, mat = s.get_mesh()
    p = pos.reshape(-1, 3, 3)
    n = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    assert np.allclose(nrm.reshape(-1, 3, 3), n[:, None, :], atol=1e-6)
    assert not uv.any()  # "f a/1 b/1 c/1" but no textured material
    t = host_scene("textured.obj")
    assert t.get_mesh()[2].any() and t.get_material(0).textured == 1


def test_bad_files(tmp_path):
    s = pta.Scene(pta.DEVICE_NONE)
    assert s.load(str(tmp_path / "nope.obj")) is False
    empty = tmp_path / "empty.obj"
    empty.write_text("# nothing\nv 0 0 0\n")
    assert s.load(str(empty)) is False and s.num_triangles() == 0
    s.create_single_triangle()  # the fallback of src/main.cpp
    pos, nrm, uv, mat = s.get_mesh()
    assert pos.tolist() == [[-0.5, -0.5, 0, 0.5, -0.5, 0, 0, 0.5, 0]]
    assert nrm.reshape(3, 3).tolist() == [[0, 0, 1]] * 3
    assert uv.tolist() == [[0, 0, 0, 1, 1, 0]]
    bad = tmp_path / "bad.obj"
    bad.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 7\n")
    assert s.load(str(bad)) is False  # index outside the vertex table: refused, never read out of bounds


def test_loader_quirks(tmp_path):
    """Only the first three vertices of a face are used; a face without a normal index turns
    normals off for the whole mesh; lines are consumed in 79-byte pieces; usemtl before mtllib
    still resolves because libraries are read in a first pass."""
    (tmp_path / "q.mtl").write_text("newmtl a\nKd 1 0 0\nnewmtl glassy\nKd 0 1 0\nRi 2.5\n")
    obj = tmp_path / "q.obj"
    obj.write_text("usemtl glassy\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 2\n"
                   "f 1//1 2//1 3//1 4//1\nusemtl a\nf 1 3 4\nmtllib q.mtl\n")
    for loader in ("product", "oracle"):
        if loader == "product":
            s = pta.Scene(pta.DEVICE_NONE)
            assert s.load(str(obj))
            pos, nrm, uv, mat = s.get_mesh()
            mats = s.materials()
        else:
            o = orc.Oracle()
            assert o.load_obj(str(obj)) == 2
            pos, nrm, uv, mat = o.get_tris()
            mats = o.materials()
        assert pos.shape[0] == 2 and mat.tolist() == [1, 0]
        assert pos[0].tolist() == [0, 0, 0, 1, 0, 0, 1, 1, 0]
        assert np.allclose(nrm.reshape(-1, 3), [0, 0, 1])  # geometric normals (second face has no vn)
        assert mats[1].type == pta.GLASS and mats[1].eta == 1.5  # "glassy" starts with glass; Ri ignored
    long_line = "v 0.125 0.25 0.5" + " " * 70 + "\n"
    obj2 = tmp_path / "long.obj"
    obj2.write_text(long_line + "v 1 0 0\nv 0 1 0\nf 1 2 3\n")
    s = pta.Scene(pta.DEVICE_NONE)
    o = orc.Oracle()
    assert s.load(str(obj2)) and o.load_obj(str(obj2)) == 1
    assert np.array_equal(s.get_mesh()[0], o.get_tris()[0])


def test_host_only_context_refuses_gpu_work():
    s = host_scene("cornellbox.obj")
    s.pre_calc(pta.BVH)
    for call in (lambda: s.set_resolution(8, 8), lambda: s.pathtrace(1), lambda: s.intersect(np.zeros((1, 6), np.float32)),
                 lambda: s.counters(), lambda: s.synchronize()):
        with pytest.raises(pta.PtError) as e:
            call()
        assert e.value.code == pta.ERR_NOT_READY


def test_parallel_builders_give_the_sequential_trees():
    """Large subtrees near the root are built by other threads and spliced in; node numbering, boxes and
    leaf references must be those of the single-threaded recursion (a procedural mesh large enough for
    several levels of concurrent subtrees)."""
    import path_tracer_amd as pta
    rng = np.random.default_rng(11)
    n = 60000
    centre = rng.random((n, 1, 3)).astype(np.float32) * 4 - 2
    pos = (centre + (rng.random((n, 3, 3)).astype(np.float32) - 0.5) * 0.05).reshape(n, 9)
    nrm = np.tile(np.array([0, 0, 1], np.float32), (n, 3)).reshape(n, 9)
    trees = {}
    for accel in (pta.BVH, pta.KD):
        for threads in (1, 0, 8):
            s = pta.Scene(pta.DEVICE_NONE)
            s.set_mesh(pos, nrm)
            s.set_option(pta.OPT_BUILD_THREADS, threads)
            s.pre_calc(accel)
            trees[(accel, threads)] = s.get_tree()
        ref = trees[(accel, 1)]
        for threads in (0, 8):
            got = trees[(accel, threads)]
            assert got["max_depth"] == ref["max_depth"]
            for key in ("left", "right", "first", "count", "box", "refs"):
                assert np.array_equal(got[key], ref[key]), (accel, threads, key)


def test_obj_extensions_are_opt_in(tmp_path):
    """A quad with negative indices: the reference's reader takes the first three corners of a face and
    positive indices only (src/cs488.h:1059, 944-946); PT_OPT_OBJ_EXTENDED fans polygons out and resolves
    indices that count back from the end of the tables."""
    import path_tracer_amd as pta
    obj = tmp_path / "quad.obj"
    obj.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0.5 2 0\n"
                   "f 1 2 3 4 5\n"
                   "f -5 -4 -3 -2\n")
    s = pta.Scene(pta.DEVICE_NONE)
    assert not s.load(str(obj))            # -5 is not a valid index for the reference's reader
    s.set_option(pta.OPT_OBJ_EXTENDED, 1)
    assert s.load(str(obj))
    assert s.num_triangles() == 3 + 2
    pos = s.get_mesh()[0].reshape(-1, 3, 3)
    quad = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0.5, 2, 0]], np.float32)
    assert np.array_equal(pos[0], quad[[0, 1, 2]]) and np.array_equal(pos[1], quad[[0, 2, 3]]) and np.array_equal(pos[2], quad[[0, 3, 4]])
    assert np.array_equal(pos[3], quad[[0, 1, 2]]) and np.array_equal(pos[4], quad[[0, 2, 3]])
    # the default reader still takes three corners of the pentagon
    s2 = pta.Scene(pta.DEVICE_NONE)
    obj2 = tmp_path / "penta.obj"
    obj2.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0.5 2 0\nf 1 2 3 4 5\n")
    assert s2.load(str(obj2)) and s2.num_triangles() == 1


def test_gif_encoder_round_trips_through_an_independent_decoder(tmp_path):
    """The recording's container (src/cs488.h:255-266, 2297-2310 hand frames to a GIF writer): frame count,
    size, delay and loop flag as written; pixels within one step of the fixed 6 x 7 x 6 palette, exact for
    palette colours; an image large and noisy enough to fill the LZW table several times decodes too."""
    from PIL import Image
    import path_tracer_amd as pta
    rng = np.random.default_rng(7)
    h, w = 37, 53
    frames = np.zeros((4, h, w, 3), np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    frames[0, ..., 0] = xx * 255 // (w - 1)
    frames[0, ..., 1] = yy * 255 // (h - 1)
    frames[0, ..., 2] = 128
    frames[1] = rng.integers(0, 256, (h, w, 3))
    frames[2] = 255
    frames[3, ..., 0], frames[3, ..., 1], frames[3, ..., 2] = 51, 85, 204   # palette levels 1/5, 2/6, 4/5
    path = str(tmp_path / "t.gif")
    pta.encode_gif(path, frames, delay_cs=3)
    im = Image.open(path)
    assert im.format == "GIF" and im.size == (w, h) and im.n_frames == 4
    assert im.info["duration"] == 30 and im.info["loop"] == 0
    for f in range(4):
        im.seek(f)
        got = np.asarray(im.convert("RGB")).astype(int)
        err = np.abs(got - frames[f].astype(int))
        assert err[..., 0].max() <= 51 and err[..., 1].max() <= 43 and err[..., 2].max() <= 51
        if f >= 2:
            assert err.max() == 0
    assert abs(float(np.asarray(Image.open(path).convert("RGB")).mean()) - float(frames[0].mean())) < 2.0
    big = rng.integers(0, 256, (1, 300, 400, 3)).astype(np.uint8)
    pta.encode_gif(str(tmp_path / "b.gif"), big)
    got = np.asarray(Image.open(tmp_path / "b.gif").convert("RGB")).astype(int)
    assert got.shape == (300, 400, 3) and np.abs(got - big[0].astype(int)).max() <= 51
    with pytest.raises(pta.PtError):
        pta.encode_gif(str(tmp_path / "no_such_dir" / "x.gif"), frames)


def _texture_through_the_loader(tmp_path, file_name):
    (tmp_path / "m.mtl").write_text("newmtl a\nKd 1 1 1\nmap_Kd %s\n" % file_name)
    (tmp_path / "m.obj").write_text("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nvn 0 0 1\nusemtl a\nf 1/1/1 2/2/1 3/3/1\n")
    s = pta.Scene(pta.DEVICE_NONE)
    assert s.load(str(tmp_path / "m.obj"))
    try:
        return s.get_texture(0), s.materials()[0]
    finally:
        s.close()


@pytest.mark.parametrize("shape", [(45, 70), (64, 64), (17, 9), (120, 201)])
def test_jpeg_textures_decode_like_an_independent_decoder(tmp_path, shape):
    """map_Kd files go through stbi_load in the reference (src/cs488.h:839-846), which reads baseline and
    progressive JPEG.  No file of the reference holds a JPEG texture, so the check is against Pillow's
    decoder on files written here: sequential and progressive, 4:4:4 / 4:2:2 / 4:2:0, optimised tables,
    restart intervals, grey.  Decoders differ in the rounding of the inverse DCT, hence the small bar
    (parity with stb_image itself is unpinned)."""
    from PIL import Image
    h, w = shape
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([xx * 255 // (w - 1), yy * 255 // (h - 1), (xx + yy) * 255 // (w + h - 2)], -1).astype(np.uint8)
    img[h // 4:h // 2, w // 5:w // 2] = [220, 30, 60]
    img[::7, ::5] = 255
    variants = {
        "q95_444": dict(quality=95, subsampling=0),
        "q75_420": dict(quality=75, subsampling=2),
        "q85_422": dict(quality=85, subsampling=1),
        "prog_444": dict(quality=90, subsampling=0, progressive=True),
        "prog_420": dict(quality=80, subsampling=2, progressive=True),
        "optimised": dict(quality=90, subsampling=0, optimize=True),
        "restarts": dict(quality=85, subsampling=2, restart_marker_blocks=3),
        "prog_restarts": dict(quality=85, subsampling=1, progressive=True, restart_marker_rows=1),
    }
    for name, kw in variants.items():
        path = tmp_path / (name + ".jpg")
        Image.fromarray(img).save(path, "JPEG", **kw)
        got, mat = _texture_through_the_loader(tmp_path, name + ".jpg")
        want = np.asarray(Image.open(path).convert("RGB")).astype(int)
        assert got.shape == want.shape and mat.textured
        err = np.abs(got.astype(int) - want)
        assert err.max() <= 4 and err.mean() < 0.3, (name, int(err.max()), float(err.mean()))
    Image.fromarray(img[..., 0], "L").save(tmp_path / "grey.jpg", "JPEG", quality=90)
    got, _ = _texture_through_the_loader(tmp_path, "grey.jpg")
    want = np.asarray(Image.open(tmp_path / "grey.jpg").convert("RGB")).astype(int)
    assert np.abs(got.astype(int) - want).max() <= 2


def test_broken_jpeg_texture_is_reported_and_left_empty(tmp_path, capfd):
    """A texture that cannot be read leaves the material without pixels and prints the reference's
    message (src/cs488.h:843); a truncated or non-JPEG file must not crash the decoder."""
    from PIL import Image
    img = (np.arange(64 * 64 * 3) % 251).astype(np.uint8).reshape(64, 64, 3)
    Image.fromarray(img).save(tmp_path / "ok.jpg", "JPEG", quality=80)
    blob = (tmp_path / "ok.jpg").read_bytes()
    for name, data in (("cut.jpg", blob[:len(blob) // 3]), ("noise.jpg", b"\xff\xd8" + bytes(range(256)) * 4), ("empty.jpg", b"")):
        (tmp_path / name).write_bytes(data)
        if name == "cut.jpg":  # a truncated scan still yields the rows that were there
            tex, mat = _texture_through_the_loader(tmp_path, name)
            assert mat.textured and tex.shape == (64, 64, 3)
            continue
        with pytest.raises(pta.PtError):   # the material stays textured but holds no pixels
            _texture_through_the_loader(tmp_path, name)
        assert "Unable to load texture" in capfd.readouterr().err
