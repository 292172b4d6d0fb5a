"""Drop-in for the reference's pybind module `standard_rasterize_cuda` and its Python helpers.

Mirrors (same names, argument order, in-place semantics, return value):
  standard_rasterize / standard_rasterize_colors   my_utils/standard_rasterize_cuda/standard_rasterize_cuda.cpp:26-40, :59-75
  face_vertices / get_visibility / get_visibility_z my_utils/standard_rasterize_cuda/visibility.py:9-100
The kernels are gif_amd/csrc/rasterize.hip (HIP, gfx950) behind the C ABI gif_rasterize[_colors]_f32 / _f64 (the reference
dispatches both floating types).
Like the reference's CHECK_INPUT (.cpp:21-23) every tensor must be a contiguous device tensor, else an
exception is raised; buffers are caller-allocated, caller-initialised, mutated in place and returned.
"""
import torch

from . import _lib, ops


def _check(t, name, dtype):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")  # wording of the reference's AT_CHECK
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    if t.dtype != dtype:
        raise _lib.GifHipError(f"{name} must be {dtype}, got {t.dtype}")


def _float_dtype(face_vertices):
    """float32 or float64, taken from face_vertices like the reference's AT_DISPATCH_FLOATING_TYPES(face_vertices.type(), ...)
    (standard_rasterize_cuda_kernel.cu:252,295); every other floating buffer must match it."""
    dt = face_vertices.dtype if isinstance(face_vertices, torch.Tensor) else None
    if dt not in (torch.float32, torch.float64):
        raise _lib.GifHipError(f"face_vertices must be float32 or float64, got {dt}")
    return dt


def standard_rasterize(face_vertices, depth_buffer, triangle_buffer, baryw_buffer, height, width):
    ft = _float_dtype(face_vertices)
    _check(face_vertices, "face_vertices", ft)
    _check(depth_buffer, "depth_buffer", ft)
    _check(triangle_buffer, "triangle_buffer", torch.int32)
    _check(baryw_buffer, "baryw_buffer", ft)
    ops.rasterize(face_vertices, depth_buffer, triangle_buffer, baryw_buffer, int(height), int(width))
    return [depth_buffer, triangle_buffer, baryw_buffer]


def standard_rasterize_colors(face_vertices, face_colors, depth_buffer, triangle_buffer, images, height, width):
    ft = _float_dtype(face_vertices)
    _check(face_vertices, "face_vertices", ft)
    _check(face_colors, "face_colors", ft)
    _check(depth_buffer, "depth_buffer", ft)
    _check(triangle_buffer, "triangle_buffer", torch.int32)
    _check(images, "images", ft)
    if face_colors.shape != face_vertices.shape:
        raise _lib.GifHipError("face_colors must have the shape of face_vertices [B,F,3,3]")
    ops.rasterize(face_vertices, depth_buffer, triangle_buffer, images, int(height), int(width), face_colors=face_colors)
    return [depth_buffer, triangle_buffer, images]


def face_vertices(vertices, faces):
    """[B,V,3], [B,F,3] -> [B,F,3,3]  (visibility.py:9-27)."""
    assert vertices.ndimension() == 3 and faces.ndimension() == 3
    assert vertices.shape[0] == faces.shape[0] and vertices.shape[2] == 3 and faces.shape[2] == 3
    bs, nv = vertices.shape[:2]
    offs = (torch.arange(bs, dtype=torch.int64, device=vertices.device) * nv)[:, None, None]
    return vertices.reshape(bs * nv, 3)[faces.long() + offs].contiguous()


def to_image_space(vertices, h, w):
    """NDC [-1,1] -> pixel units, z shifted to min 1 over the whole batch (visibility.py:38-40)."""
    v = vertices.clone()
    v[..., 0] = v[..., 0] * w / 2 + w / 2
    v[..., 1] = v[..., 1] * h / 2 + h / 2
    v[..., 2] = v[..., 2] - v[..., 2].min() + 1
    return v


def new_buffers(bz, h, w, device):
    depth = torch.zeros([bz, h, w], device=device).float() + 1e6
    tri = torch.zeros([bz, h, w], device=device).int() - 1
    bary = torch.zeros([bz, h, w, 3], device=device).float()
    return depth, tri, bary


def get_visibility(vertices, triangles, h, w, print_time=False):
    """Per-vertex visibility from the face-index buffer (visibility.py:29-60)."""
    bz, device = vertices.shape[0], vertices.device
    v = to_image_space(vertices, h, w)
    depth, tri, bary = new_buffers(bz, h, w, device)
    standard_rasterize(face_vertices(v, triangles), depth, tri, bary, h, w)
    vert_vis = torch.zeros([bz, vertices.shape[1]], device=device)
    tri = tri.reshape(bz, -1)
    for i in range(bz):
        vis_tri = torch.unique(tri[i])[1:].long()  # first unique value is the -1 background, as in the reference
        vert_vis[i, torch.unique(triangles[i, vis_tri, :].flatten().long())] = 1.0
    return vert_vis


def get_visibility_z(vertices, triangles, h, w, print_time=False):
    """Per-vertex visibility from a bilinear depth test (visibility.py:62-100), vectorised."""
    bz, device = vertices.shape[0], vertices.device
    v = to_image_space(vertices, h, w)
    depth, tri, bary = new_buffers(bz, h, w, device)
    standard_rasterize(face_vertices(v, triangles), depth, tri, bary, h, w)
    zrange = v[..., -1].max() - v[..., -1].min()
    x, y, z = v[..., 0], v[..., 1], v[..., 2]
    fx, fy, cx, cy = torch.floor(x).long(), torch.floor(y).long(), torch.ceil(x).long(), torch.ceil(y).long()
    bi = torch.arange(bz, device=device)[:, None].expand_as(fx)
    ul, ur, dl, dr = depth[bi, fy, fx], depth[bi, fy, cx], depth[bi, cy, fx], depth[bi, cy, cx]
    yd, xd = y - torch.floor(y), x - torch.floor(x)
    d = ul * (1 - xd) * (1 - yd) + ur * xd * (1 - yd) + dl * (1 - xd) * yd + dr * xd * yd
    return (z < d + zrange * 0.02).float()
