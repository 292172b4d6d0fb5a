"""Texture stealing on MI355X — drop-in for FlameTextureSpace (model/stg2_generator.py:336-421), SURVEY §8(f) row 2.

compute_texture_map() keeps the reference signature and return value (texture image [B,C,256,256], bool visibility mask
[B,1,256,256]) and runs as one HIP kernel (gif_texture_map_f32) with a HIP backward w.r.t. the source image (the
texture-interpolation loss differentiates through it, loss_functions/losses.py:147-176).  forward() needs the FLAME
layer of the absent photometric_optimization submodule: it is accepted as a constructor argument (any callable returning
vertices) and raises a clear error when missing — parity of that part is unpinned / out of scope.
"""
import numpy as np
import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib
from . import render


class _TextureMapFn(Function):
    @staticmethod
    def forward(ctx, img, verts, normals, cam, tmap, tfaces, tbc, T):
        lib = _lib.load()
        if not img.is_cuda:
            raise _lib.GifHipError("compute_texture_map needs device tensors (no CPU fallback)")
        img = img.contiguous().float()
        verts, normals, cam = verts.contiguous().float(), normals.contiguous().float(), cam.contiguous().float().view(-1, 3)
        B, C, H, W = img.shape
        V = verts.shape[1]
        tex = torch.empty((B, C, T, T), device=img.device, dtype=torch.float32)
        mask = torch.empty((B, 1, T, T), device=img.device, dtype=torch.uint8)
        with torch.cuda.device(img.device):  # launch on the operands' device and its current stream
            _lib.check(lib.gif_texture_map_f32(img.data_ptr(), verts.data_ptr(), normals.data_ptr(), cam.data_ptr(),
                                               tmap.data_ptr(), tfaces.data_ptr(), tbc.data_ptr(), tex.data_ptr(),
                                               mask.data_ptr(), B, C, H, W, V, T, torch.cuda.current_stream().cuda_stream),
                       "texture_map")
        ctx.save_for_backward(verts, normals, cam, tmap, tfaces, tbc)
        ctx.dims = (B, C, H, W, V, T)
        mask = mask.view(torch.bool)  # the kernel writes 0 / 1 bytes; the reference returns a bool mask (stg2_generator.py:415)
        ctx.mark_non_differentiable(mask)
        return tex, mask

    @staticmethod
    @once_differentiable  # raw kernel launch: a double backward must fail loudly, not return a history-free gradient
    def backward(ctx, gtex, _gmask):
        verts, normals, cam, tmap, tfaces, tbc = ctx.saved_tensors
        B, C, H, W, V, T = ctx.dims
        lib = _lib.load()
        gtex = gtex.contiguous()
        gimg = torch.empty((B, C, H, W), device=gtex.device, dtype=torch.float32)
        with torch.cuda.device(gtex.device):
            _lib.check(lib.gif_texture_map_bwd_f32(gtex.data_ptr(), verts.data_ptr(), normals.data_ptr(), cam.data_ptr(),
                                                   tmap.data_ptr(), tfaces.data_ptr(), tbc.data_ptr(), gimg.data_ptr(), B, C, H,
                                                   W, V, T, torch.cuda.current_stream().cuda_stream), "texture_map_bwd")
        return gimg, None, None, None, None, None, None, None


class FlameTextureSpace(nn.Module):
    TEX = 256  # the reference hard-codes a 256x256 texture grid (:402, :414)

    def __init__(self, texture_data, data_un_normalizer, flame=None, faces=None):
        """texture_data: dict with x_coords, y_coords, valid_pixel_ids, valid_pixel_3d_faces, valid_pixel_b_coords
        (reference :348-353).  flame / faces: the FLAME layer and its triangle list (absent submodule) — optional."""
        super().__init__()
        self.texture_data = texture_data
        self.data_un_normalizer = data_un_normalizer
        self.flame = flame
        self.faces = faces
        x = np.asarray(texture_data.get('x_coords')).astype('int')
        y = np.asarray(texture_data.get('y_coords')).astype('int')
        ids = np.asarray(texture_data.get('valid_pixel_ids')).astype('int')
        tmap = -np.ones(self.TEX * self.TEX, np.int32)
        tmap[y[ids] * self.TEX + x[ids]] = np.arange(len(ids), dtype=np.int32)  # later entries win, like index assignment
        self.register_buffer('texel_map', torch.from_numpy(tmap), persistent=False)
        self.register_buffer('valid_pixel_3d_faces',
                             torch.from_numpy(np.asarray(texture_data.get('valid_pixel_3d_faces')).astype('int32')),
                             persistent=False)
        self.register_buffer('valid_pixel_b_coords',
                             torch.from_numpy(np.asarray(texture_data.get('valid_pixel_b_coords')).astype('float32')),
                             persistent=False)

    def forward(self, source_img, flame_params_full):
        if self.flame is None or self.faces is None:
            raise _lib.GifHipError("FlameTextureSpace.forward needs a FLAME layer (photometric_optimization submodule, not "
                                   "part of this repository): pass flame=/faces= or call compute_texture_map directly")
        if self.data_un_normalizer is not None:
            flame_params_full = self.data_un_normalizer(flame_params_full)
        shape, expression = flame_params_full[:, 0:100], flame_params_full[:, 100:150]
        pose, camera_params = flame_params_full[:, 150:156], flame_params_full[:, 156:159]
        verts, _, _ = self.flame(shape_params=shape, expression_params=expression, pose_params=pose)
        trans_verts = render.batch_orth_proj(verts, camera_params)
        trans_verts = torch.cat([trans_verts[:, :, :1], -trans_verts[:, :, 1:]], 2)  # :368
        vertex_normals = render.vertex_normals(trans_verts, self.faces)
        return self.compute_texture_map(source_img, verts, vertex_normals, camera_params=camera_params)

    def compute_texture_map(self, source_img, target_mesh_v, vertex_normals, camera_params):
        return _TextureMapFn.apply(source_img, target_mesh_v, vertex_normals, camera_params, self.texel_map,
                                   self.valid_pixel_3d_faces, self.valid_pixel_b_coords, self.TEX)
