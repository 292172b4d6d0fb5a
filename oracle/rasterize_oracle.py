"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by gif_amd/).

numpy/ctypes front-end of oracle/rasterize_ref.c plus a numpy restatement of the reference's
Python helpers around the rasteriser:
  face_vertices      my_utils/standard_rasterize_cuda/visibility.py:9-27
  get_visibility     visibility.py:29-60
  get_visibility_z   visibility.py:62-100
Parity: PINNED by tests/golden/body_mesh.npz (the reference's body_vis.obj / body_vis_z.obj).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int32)
        _LIB.oracle_rasterize.argtypes = [fp, fp, ip, fp] + [ctypes.c_int] * 4
        _LIB.oracle_rasterize.restype = None
        _LIB.oracle_rasterize_colors.argtypes = [fp, fp, fp, ip, fp] + [ctypes.c_int] * 4
        _LIB.oracle_rasterize_colors.restype = None
        dp = ctypes.POINTER(ctypes.c_double)
        _LIB.oracle_rasterize_f64.argtypes = [dp, dp, ip, dp] + [ctypes.c_int] * 4
        _LIB.oracle_rasterize_f64.restype = None
        _LIB.oracle_rasterize_colors_f64.argtypes = [dp, dp, dp, ip, dp] + [ctypes.c_int] * 4
        _LIB.oracle_rasterize_colors_f64.restype = None
    return _LIB


def _f(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _i(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def _d(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def standard_rasterize(face_verts, depth, tri, bary, h, w):
    """In-place on C-contiguous float32 (or float64) / int32 numpy buffers, like standard_rasterize_cuda.cpp:26-40."""
    assert face_verts.dtype in (np.float32, np.float64) and face_verts.flags.c_contiguous
    assert depth.dtype == face_verts.dtype and bary.dtype == face_verts.dtype
    B, F = face_verts.shape[:2]
    if face_verts.dtype == np.float64:
        lib().oracle_rasterize_f64(_d(face_verts), _d(depth), _i(tri), _d(bary), B, F, h, w)
    else:
        lib().oracle_rasterize(_f(face_verts), _f(depth), _i(tri), _f(bary), B, F, h, w)
    return depth, tri, bary


def standard_rasterize_colors(face_verts, face_colors, depth, tri, images, h, w):
    assert face_verts.dtype in (np.float32, np.float64) and face_colors.dtype == face_verts.dtype
    assert depth.dtype == face_verts.dtype and images.dtype == face_verts.dtype
    B, F = face_verts.shape[:2]
    if face_verts.dtype == np.float64:
        lib().oracle_rasterize_colors_f64(_d(face_verts), _d(face_colors), _d(depth), _i(tri), _d(images), B, F, h, w)
    else:
        lib().oracle_rasterize_colors(_f(face_verts), _f(face_colors), _f(depth), _i(tri), _f(images), B, F, h, w)
    return depth, tri, images


def face_vertices(vertices, faces):
    """[B,V,3], [B,F,3] -> [B,F,3,3]  (visibility.py:9-27)."""
    B, V = vertices.shape[:2]
    flat = vertices.reshape(B * V, 3)
    idx = faces.astype(np.int64) + (np.arange(B, dtype=np.int64) * V)[:, None, None]
    return np.ascontiguousarray(flat[idx])


def to_image_space(vertices, h, w):
    """NDC [-1,1] -> pixel units, z shifted so that min z = 1 (visibility.py:38-40); float32 arithmetic."""
    v = vertices.astype(np.float32).copy()
    v[..., 0] = v[..., 0] * np.float32(w) / np.float32(2) + np.float32(w / 2)
    v[..., 1] = v[..., 1] * np.float32(h) / np.float32(2) + np.float32(h / 2)
    v[..., 2] = v[..., 2] - v[..., 2].min() + np.float32(1)
    return v


def new_buffers(B, h, w):
    depth = np.zeros((B, h, w), np.float32) + np.float32(1e6)
    tri = np.zeros((B, h, w), np.int32) - 1
    bary = np.zeros((B, h, w, 3), np.float32)
    return depth, tri, bary


def get_visibility(vertices, triangles, h, w):
    B = vertices.shape[0]
    v = to_image_space(vertices, h, w)
    depth, tri, bary = new_buffers(B, h, w)
    standard_rasterize(face_vertices(v, triangles), depth, tri, bary, h, w)
    vis = np.zeros((B, vertices.shape[1]), np.float32)
    for i in range(B):
        t = np.unique(tri[i].reshape(-1))
        t = t[1:] if t[0] < 0 else t  # reference drops the first unique value (-1), visibility.py:55
        vis[i, np.unique(triangles[i, t].reshape(-1))] = 1.0
    return vis, (depth, tri, bary)


def get_visibility_z(vertices, triangles, h, w):
    B = vertices.shape[0]
    v = to_image_space(vertices, h, w)
    depth, tri, bary = new_buffers(B, h, w)
    standard_rasterize(face_vertices(v, triangles), depth, tri, bary, h, w)
    zrange = v[..., 2].max() - v[..., 2].min()
    vis = np.zeros((B, vertices.shape[1]), np.float32)
    for i in range(B):
        x, y, z = v[i, :, 0], v[i, :, 1], v[i, :, 2]
        fx, fy = np.floor(x).astype(np.int64), np.floor(y).astype(np.int64)
        cx, cy = np.ceil(x).astype(np.int64), np.ceil(y).astype(np.int64)
        ul, ur = depth[i, fy, fx], depth[i, fy, cx]
        dl, dr = depth[i, cy, fx], depth[i, cy, cx]
        yd, xd = y - np.floor(y), x - np.floor(x)
        d = ul * (1 - xd) * (1 - yd) + ur * xd * (1 - yd) + dl * (1 - xd) * yd + dr * xd * yd
        vis[i, z < d + zrange * np.float32(0.02)] = 1.0
    return vis, (depth, tri, bary)
