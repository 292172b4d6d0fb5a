"""Condition-render pipeline around the rasteriser: projected mesh -> 6-channel condition image.

Mirrors the pieces of the reference that exist in-tree:
  vertex_normals / batch_orth_proj       model/mesh_and_3d_helpers.py:5-50
  NDC -> pixel transform, buffer init    my_utils/standard_rasterize_cuda/visibility.py:38-44
  8-bit quantisation of the renders      my_utils/visualize_flame_overlay.py:29-31  (floor(clamp*255)/255)
  [-1,1] scaling and channel order       plots/generate_random_samples.py:22-30, :188-189  (cat(texture, normal))
The FLAME layer and the spherical-harmonics texture shading live in the absent `photometric_optimization` submodule
(parity unpinned, out of scope): the per-vertex "texture" attribute is therefore an INPUT here.
Kernels: gif_vertex_normals_f32 (gather, deterministic) and gif_rasterize_colors_f32, both behind the C ABI.
"""
import numpy as np
import torch

from . import _lib
from . import standard_rasterize as sr


def _topology_csr(faces_cpu: np.ndarray):
    """vertex -> (face, corner) entries ordered like the reference's index_add_ passes: corner 1, 2, 0; faces ascending."""
    F = faces_cpu.shape[0]
    V = int(faces_cpu.max()) + 1 if F else 0
    ents, verts = [], []
    for rank, corner in enumerate((1, 2, 0)):
        ents.append(np.arange(F, dtype=np.int64) * 4 + corner)
        verts.append(faces_cpu[:, corner].astype(np.int64))
    ents, verts = np.concatenate(ents), np.concatenate(verts)
    order = np.argsort(verts, kind="stable")  # stable: keeps (pass, face) order inside a vertex
    counts = np.bincount(verts, minlength=V)
    return ents[order].astype(np.int32), counts


def vertex_normals(vertices, faces):
    """[B,V,3] float32, faces [F,3] or [B,F,3] (same topology for every sample) -> unit normals [B,V,3]."""
    assert vertices.ndimension() == 3 and vertices.shape[2] == 3
    if faces.ndimension() == 3:
        faces = faces[0]
    if not vertices.is_cuda:
        raise _lib.GifHipError("vertex_normals needs device tensors (no CPU fallback)")
    B, V, _ = vertices.shape
    # the topology tables live ON the faces tensor (attribute), validated by its in-place version counter: no global
    # cache keyed by an address that could be recycled
    cached = getattr(faces, "_gif_csr", None)
    if cached is None or cached[0] != (faces._version, V, str(vertices.device)):
        f_cpu = faces.detach().cpu().numpy().astype(np.int64)
        if f_cpu.size and (f_cpu.min() < 0 or f_cpu.max() >= V):
            raise _lib.GifHipError(f"vertex_normals: face index out of range [0,{V})")
        ent, counts = _topology_csr(f_cpu)
        off = np.zeros(V + 1, np.int32)
        off[1:len(counts) + 1] = np.cumsum(counts)[:V]
        off[len(counts) + 1:] = off[len(counts)]
        dev = vertices.device
        cached = ((faces._version, V, str(dev)),
                  (faces.to(torch.int32).contiguous(), torch.from_numpy(off).to(dev), torch.from_numpy(ent).to(dev)))
        faces._gif_csr = cached
    f32, off, ent = cached[1]
    verts = vertices.contiguous().float()
    out = torch.empty_like(verts)
    lib = _lib.load()
    with torch.cuda.device(verts.device):  # launch on the operands' device and its current stream
        _lib.check(lib.gif_vertex_normals_f32(verts.data_ptr(), f32.data_ptr(), off.data_ptr(), ent.data_ptr(), out.data_ptr(),
                                              B, V, f32.shape[0], torch.cuda.current_stream().cuda_stream), "vertex_normals")
    return out


def batch_orth_proj(X, camera):
    """Orthographic camera [s, tx, ty]: s * (X.xy + t), z scaled by s  (mesh_and_3d_helpers.py:40-50)."""
    camera = camera.clone().view(-1, 1, 3)
    X_trans = torch.cat([X[:, :, :2] + camera[:, :, 1:], X[:, :, 2:]], 2)
    return camera[:, :, 0:1] * X_trans


def rasterize_attributes(vertices_ndc, faces, attributes, h, w):
    """Barycentric interpolation of per-vertex attributes [B,V,3] over the z-buffered mesh -> images [B,3,h,w], plus
    the coverage mask [B,1,h,w].  vertices_ndc: x,y in [-1,1], any z (visibility.py:38-44 conventions)."""
    B = vertices_ndc.shape[0]
    if faces.ndimension() == 2:
        faces = faces[None].expand(B, -1, -1)
    v = sr.to_image_space(vertices_ndc.float(), h, w)
    fv = sr.face_vertices(v, faces)
    fc = sr.face_vertices(attributes.float().contiguous(), faces)
    depth, tri, img = sr.new_buffers(B, h, w, vertices_ndc.device)
    sr.standard_rasterize_colors(fv, fc, depth, tri, img, h, w)
    return img.permute(0, 3, 1, 2).contiguous(), (tri >= 0)[:, None]


def quantize_8bit(img01):
    """floor(clamp(x,0,1)*255)/255 — the reference's render post-processing (visualize_flame_overlay.py:29-31)."""
    return torch.floor(img01.clamp(0, 1) * 255) / 255.0


def render_condition(vertices_ndc, faces, vertex_texture, h=256, w=256):
    """6-channel generator condition: cat(texture render, normal render) in [-1,1] (generate_random_samples.py:22-30,
    :188-189).  Normals are mapped to [0,1] as n*0.5+0.5 before the 8-bit quantisation (normal-map convention)."""
    normals = vertex_normals(vertices_ndc, faces)
    normal_img, _ = rasterize_attributes(vertices_ndc, faces, normals * 0.5 + 0.5, h, w)
    tex_img, _ = rasterize_attributes(vertices_ndc, faces, vertex_texture, h, w)
    normal_img = quantize_8bit(normal_img) * 2 - 1
    tex_img = quantize_8bit(tex_img) * 2 - 1
    return torch.cat((tex_img, normal_img), dim=1)


class FlameConditionRenderer:
    """callable flame_batch [N, >=159] -> (rend_flm, norma_map_img), both [N,3,h,w] in [-1,1]: the role
    OverLayViz.get_rendered_mesh + the [-1,1] scaling play in InterpolatedTextureLoss.get_image_and_textures
    (loss_functions/losses.py:184-215).  `flame` is the FLAME layer (absent submodule: injected; gif_amd.data.SyntheticFlame for
    synthetic runs), `vertex_texture` [V,3] in [0,1] replaces the SH-lit albedo render (out of scope)."""

    def __init__(self, flame, faces, vertex_texture, h=256, w=256):
        self.flame, self.faces, self.vertex_texture, self.h, self.w = flame, faces, vertex_texture, h, w

    def vertices(self, flame_batch):
        """(FLAME vertices, vertices in NDC with the y flip of stg2_generator.py:368, camera)."""
        shape, exp = flame_batch[:, 0:100], flame_batch[:, 100:150]
        pose, cam = flame_batch[:, 150:156], flame_batch[:, 156:159]
        verts, _, _ = self.flame(shape_params=shape, expression_params=exp, pose_params=pose)
        trans = batch_orth_proj(verts, cam)
        return verts, torch.cat([trans[:, :, :1], -trans[:, :, 1:]], 2), cam

    def __call__(self, flame_batch):
        _, v_ndc, _ = self.vertices(flame_batch)
        tex = self.vertex_texture[None].expand(v_ndc.shape[0], -1, -1)
        cond = render_condition(v_ndc, self.faces, tex, self.h, self.w)
        return cond[:, :3], cond[:, 3:]
