"""ORACLE — TEST INFRASTRUCTURE ONLY.  numpy restatement of model/mesh_and_3d_helpers.py:5-50 (vertex_normals with its
three sequential index_add_ passes, batch_orth_proj).  Pinned against the imported reference in the build container
(tests/test_oracle_mesh.py, marker `reference`) and against tests/golden/mesh_golden.npz everywhere."""
import numpy as np


def vertex_normals(vertices, faces):
    """vertices [B,V,3] float32, faces [B,F,3] or [F,3] -> [B,V,3]; float32 arithmetic, reference summation order."""
    vertices = np.asarray(vertices, np.float32)
    B, V, _ = vertices.shape
    if faces.ndim == 2:
        faces = np.repeat(faces[None], B, 0)
    out = np.zeros((B, V, 3), np.float32)
    for b in range(B):
        vf = vertices[b][faces[b]]  # [F,3,3]
        n = np.zeros((V, 3), np.float32)
        # index_add_ order of the reference: corner 1, corner 2, corner 0 (:27-32)
        np.add.at(n, faces[b][:, 1], np.cross(vf[:, 2] - vf[:, 1], vf[:, 0] - vf[:, 1]).astype(np.float32))
        np.add.at(n, faces[b][:, 2], np.cross(vf[:, 0] - vf[:, 2], vf[:, 1] - vf[:, 2]).astype(np.float32))
        np.add.at(n, faces[b][:, 0], np.cross(vf[:, 1] - vf[:, 0], vf[:, 2] - vf[:, 0]).astype(np.float32))
        ln = np.maximum(np.sqrt((n * n).sum(1, keepdims=True)), np.float32(1e-6))  # F.normalize(eps=1e-6) :34
        out[b] = n / ln
    return out


def batch_orth_proj(X, camera):
    cam = np.asarray(camera, np.float32).reshape(-1, 1, 3)
    Xt = np.concatenate([X[:, :, :2] + cam[:, :, 1:], X[:, :, 2:]], 2)
    return cam[:, :, 0:1] * Xt
