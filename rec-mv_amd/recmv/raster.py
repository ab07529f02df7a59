"""First-hit mesh rasteriser on the HIP kernel of csrc/rasterize_meshes.hip.

Stands where the reference uses pytorch3d's `MeshRasterizer` inside `maskRender` to get the `Fragments` that
`utils.FindSurfacePs` turns into visible canonical surface points (engineer/networks/OptimGarmentNetwork.py:742-767,
RasterizationSettings :2336-2347, utils/FindSurfacePs.py:7-37).  Names and output layout are pytorch3d's:
`pix_to_face [N,H,W,K]` (packed face index, -1 = empty), `zbuf`, `bary_coords [N,H,W,K,3]`, `dists`, with K = 1 —
the only value the reference uses.  No CPU path: without librecmv_hip.so the import of `_lib` fails.
"""
from collections import namedtuple

import torch

from . import _lib as L

Fragments = namedtuple("Fragments", ["pix_to_face", "zbuf", "bary_coords", "dists"])


def rasterize_meshes(face_verts, mesh_first_face, mesh_num_faces, image_size, blur_radius=0.0, faces_per_pixel=1,
                     perspective_correct=True, clip_barycentric_coords=False, cull_backfaces=False,
                     max_faces_per_mesh=None):
    """face_verts [F,3,3] f32 CUDA: (x_ndc, y_ndc, z_view) of each face's corners, meshes packed back to back;
    mesh_first_face / mesh_num_faces int64 [N] on the device.  image_size = (H, W).  Returns `Fragments`.
    `max_faces_per_mesh` (python int) avoids reading mesh_num_faces back; default = F."""
    if faces_per_pixel != 1:
        raise ValueError("rasterize_meshes: faces_per_pixel must be 1 (the reference's setting, "
                         "OptimGarmentNetwork.py:2343)")
    if clip_barycentric_coords:
        raise ValueError("rasterize_meshes: clip_barycentric_coords=True is not provided (reference: False, :2345)")
    L.require_cuda(face_verts, "face_verts")
    L.require_contiguous(face_verts, "face_verts")
    if face_verts.dtype != torch.float32 or face_verts.dim() != 3 or tuple(face_verts.shape[1:]) != (3, 3):
        raise ValueError("face_verts must be float32 of shape [F,3,3]")
    for t, name in ((mesh_first_face, "mesh_first_face"), (mesh_num_faces, "mesh_num_faces")):
        L.require_cuda(t, name)
        L.require_contiguous(t, name)
        if t.dtype != torch.int64:
            raise ValueError(name + " must be int64")
    if mesh_first_face.numel() != mesh_num_faces.numel():
        raise ValueError("mesh_first_face and mesh_num_faces must have one entry per mesh")
    H, W = int(image_size[0]), int(image_size[1])
    N, F = mesh_first_face.numel(), face_verts.shape[0]
    dev = face_verts.device
    pix_to_face = torch.empty((N, H, W, 1), dtype=torch.int64, device=dev)
    zbuf = torch.empty((N, H, W, 1), dtype=torch.float32, device=dev)
    bary = torch.empty((N, H, W, 1, 3), dtype=torch.float32, device=dev)
    dists = torch.empty((N, H, W, 1), dtype=torch.float32, device=dev)
    lib = L.lib()
    with torch.cuda.device(dev):
        nbytes = int(lib.recmv_rasterize_meshes_workspace_bytes(N, H, W, F))
        ws = torch.empty(max(nbytes, 64), dtype=torch.uint8, device=dev)
        L.check(lib.recmv_rasterize_meshes(L.ptr(face_verts), L.ptr(mesh_first_face), L.ptr(mesh_num_faces), N, F,
                                           F if max_faces_per_mesh is None else int(max_faces_per_mesh), H, W,
                                           float(blur_radius), int(bool(perspective_correct)),
                                           int(bool(cull_backfaces)), L.ptr(pix_to_face), L.ptr(zbuf), L.ptr(bary),
                                           L.ptr(dists), L.ptr(ws), ws.numel(), L.stream_ptr(dev)),
                "rasterize_meshes")
    return Fragments(pix_to_face, zbuf, bary, dists)


class MeshRasterizer:
    """`MeshRasterizer(cameras, raster_settings)(meshes)` of the reference's maskRender, for a batch of meshes that
    share one face table (one deformed garment per frame): verts [N,V,3] world space, faces [F,3] int64."""

    def __init__(self, cameras, image_size, blur_radius=0.0, perspective_correct=True, cull_backfaces=False):
        self.cameras = cameras
        self.image_size = (int(image_size[0]), int(image_size[1]))           # (H, W)
        self.blur_radius = float(blur_radius)
        self.perspective_correct = bool(perspective_correct)
        self.cull_backfaces = bool(cull_backfaces)

    def transform(self, verts):
        """World -> (NDC x, NDC y, view z): MeshRasterizer.transform of pytorch3d 0.4.0 keeps the view-space depth."""
        return self.cameras.transform_points_ndc(verts)

    def __call__(self, verts, faces):
        N, V = verts.shape[0], verts.shape[1]
        F = faces.shape[0]
        ndc = self.transform(verts.reshape(-1, 3)).view(N, V, 3)
        face_verts = ndc[:, faces.reshape(-1)].reshape(N * F, 3, 3).contiguous()
        first = torch.arange(N, device=verts.device, dtype=torch.int64) * F
        num = torch.full((N,), F, device=verts.device, dtype=torch.int64)
        return rasterize_meshes(face_verts, first, num, self.image_size, self.blur_radius, 1,
                                self.perspective_correct, False, self.cull_backfaces, max_faces_per_mesh=F)
