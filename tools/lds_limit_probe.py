"""Scenes of 150-256 triangles with up to 32 materials, where the staged tree, the stacks and the shading geometry
compete for a workgroup's LDS in the fused launch.  usage: lds_limit_probe.py [PT_OPT_LDS_NODES]"""
import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import path_tracer_amd as pta
def write_scene(path, n_tris, n_mats):
    # a lit box of random small triangles inside [-1,1]^3 with n_mats diffuse materials and one light
    rng = np.random.default_rng(3)
    with open(path.replace(".obj", ".mtl"), "w") as f:
        for m in range(n_mats):
            f.write("newmtl m%d\nKd %.3f %.3f %.3f\n" % ((m,) + tuple(rng.uniform(0.2, 0.9, 3))))
            if m == 0: f.write("Ke 8 8 8\n")
    with open(path, "w") as f:
        f.write("mtllib %s\n" % os.path.basename(path).replace(".obj", ".mtl"))
        k = 1
        for t in range(n_tris):
            c = rng.uniform(-0.8, 0.8, 3)
            for _ in range(3):
                p = c + rng.uniform(-0.25, 0.25, 3)
                f.write("v %.5f %.5f %.5f\n" % tuple(p))
            f.write("usemtl m%d\nf %d %d %d\n" % (t % n_mats, k, k + 1, k + 2)); k += 3
for n_tris, n_mats in ((200, 8), (256, 32), (256, 4), (150, 32)):
    for accel in (pta.BVH, pta.KD):
        p = "/tmp/lds_probe_%d_%d.obj" % (n_tris, n_mats)
        write_scene(p, n_tris, n_mats)
        s = pta.Scene(0)
        assert s.load(p)
        s.pre_calc(accel)
        if len(sys.argv) > 1:
            try:
                s.set_option(pta.OPT_LDS_NODES, int(sys.argv[1]))
            except pta.PtError as e:
                print(n_tris, n_mats, "kd" if accel == pta.KD else "bvh", "LDS_NODES refused:", e, flush=True)
        s.set_resolution(128, 96)
        try:
            for _ in range(3): s.pathtrace(4)
            s.synchronize()
            img = s.accumulation_buffer()
            print(n_tris, n_mats, "kd" if accel == pta.KD else "bvh", "ok, in LDS:", s.get_option(pta.OPT_TREE_IN_LDS), "mean %.4f" % img.mean(), flush=True)
        except pta.PtError as e:
            print(n_tris, n_mats, "kd" if accel == pta.KD else "bvh", "FAILED:", e, flush=True)
        s.close()
