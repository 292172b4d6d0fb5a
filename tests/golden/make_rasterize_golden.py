"""Generates tests/golden/body_mesh.npz from the reference's own rasteriser fixtures.

Run in the build container only (needs /root/reference):
    python tests/golden/make_rasterize_golden.py

Inputs (reference data, not source code):
  my_utils/standard_rasterize_cuda/data/obj/body.obj        mesh fed to demo_vert_visibility.py:12-22
  my_utils/standard_rasterize_cuda/data/obj/body_vis.obj    output of visibility.get_visibility   (h=w=512, verts*0.8)
  my_utils/standard_rasterize_cuda/data/obj/body_vis_z.obj  output of visibility.get_visibility_z
The *_vis objs store the per-vertex visibility flag as the vertex colour (write_obj_with_colors).
"""
import os
import numpy as np

REF = "/root/reference/my_utils/standard_rasterize_cuda/data/obj"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "body_mesh.npz")


def read_obj(path):
    v, c, f = [], [], []
    with open(path) as fh:
        for line in fh:
            t = line.split()
            if not t:
                continue
            if t[0] == "v":
                v.append([float(x) for x in t[1:4]])
                if len(t) >= 7:
                    c.append(float(t[4]))
            elif t[0] == "f":
                f.append([int(x.split("/")[0]) - 1 for x in t[1:4]])
    return np.asarray(v, np.float64), np.asarray(c, np.float32), np.asarray(f, np.int32)


def main():
    v, _, f = read_obj(os.path.join(REF, "body.obj"))
    v1, vis, f1 = read_obj(os.path.join(REF, "body_vis.obj"))
    v2, visz, f2 = read_obj(os.path.join(REF, "body_vis_z.obj"))
    assert (f == f1).all() and (f == f2).all()
    # the demo scales by 0.8 in float32 (helpers.Mesh loads float32 tensors)
    v32 = v.astype(np.float32)
    assert np.array_equal((v32 * np.float32(0.8)).astype(np.float32), v1.astype(np.float32))
    np.savez_compressed(OUT, vertices=v32, faces=f, vis=vis.astype(np.uint8), vis_z=visz.astype(np.uint8))
    print("wrote", OUT, v32.shape, f.shape, vis.mean(), visz.mean())


if __name__ == "__main__":
    main()
