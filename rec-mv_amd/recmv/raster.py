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
    with L.device_guard(dev):
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


# ------------------------------------------------------------------------------------------- points
PointFragments = namedtuple("PointFragments", ["idx", "zbuf", "dists"])


class _RasterizePoints(torch.autograd.Function):
    """pytorch3d `_RasterizePoints` (renderer/points/rasterize_points.py): idx is not differentiable, dists and zbuf
    are, w.r.t. the packed NDC points."""

    @staticmethod
    def forward(ctx, points, cloud_first, cloud_num, H, W, radius, K, max_points):
        L.require_cuda(points, "points")
        L.require_contiguous(points, "points")
        if points.dtype != torch.float32 or points.dim() != 2 or points.shape[1] != 3:
            raise ValueError("points must be float32 of shape [P,3]")
        for t, name in ((cloud_first, "cloud_first_point"), (cloud_num, "cloud_num_points")):
            L.require_cuda(t, name)
            L.require_contiguous(t, name)
            if t.dtype != torch.int64:
                raise ValueError(name + " must be int64")
        N, P = cloud_first.numel(), points.shape[0]
        dev = points.device
        idx = torch.empty((N, H, W, K), dtype=torch.int32, device=dev)
        zbuf = torch.empty((N, H, W, K), dtype=torch.float32, device=dev)
        dists = torch.empty((N, H, W, K), dtype=torch.float32, device=dev)
        lib = L.lib()
        with L.device_guard(dev):
            nbytes = int(lib.recmv_rasterize_points_workspace_bytes(N, H, W, P, float(radius)))
            if nbytes < 0:
                raise ValueError("rasterize_points: bad sizes / radius")
            ws = torch.empty(max(nbytes, 64), dtype=torch.uint8, device=dev)
            L.check(lib.recmv_rasterize_points(L.ptr(points), L.ptr(cloud_first), L.ptr(cloud_num), N, P,
                                               P if max_points is None else int(max_points), H, W, float(radius),
                                               int(K), L.ptr(idx), L.ptr(zbuf), L.ptr(dists), L.ptr(ws), ws.numel(),
                                               L.stream_ptr(dev)), "rasterize_points")
        ctx.save_for_backward(points, idx, cloud_first, cloud_num)
        ctx.radius, ctx.max_points = float(radius), (P if max_points is None else int(max_points))
        ctx.mark_non_differentiable(idx)
        return idx, zbuf, dists

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, _g_idx, g_zbuf, g_dists):
        points, idx, cloud_first, cloud_num = ctx.saved_tensors
        N, H, W, K = idx.shape
        g_points = torch.empty_like(points)
        with L.device_guard(points.device):
            L.check(L.lib().recmv_rasterize_points_backward(
                L.ptr(points), L.ptr(cloud_first), L.ptr(cloud_num), L.ptr(idx),
                L.ptr(g_dists.contiguous()) if g_dists is not None else None,
                L.ptr(g_zbuf.contiguous()) if g_zbuf is not None else None, N, points.shape[0], ctx.max_points, H, W,
                ctx.radius, K, L.ptr(g_points), L.stream_ptr(points.device)), "rasterize_points_backward")
        return g_points, None, None, None, None, None, None, None


def rasterize_points(points, cloud_first_point, cloud_num_points, image_size, radius, points_per_pixel=8,
                     max_points_per_cloud=None):
    """points [P,3] f32 CUDA (x_ndc, y_ndc, z_view), clouds packed back to back.  Returns `PointFragments`
    (idx int32 [N,H,W,K], zbuf, dists), differentiable w.r.t. `points` through dists / zbuf."""
    H, W = int(image_size[0]), int(image_size[1])
    return PointFragments(*_RasterizePoints.apply(points, cloud_first_point, cloud_num_points, H, W, float(radius),
                                                  int(points_per_pixel), max_points_per_cloud))


class _AlphaComposite(torch.autograd.Function):
    """pytorch3d `alpha_composite` on fragment-major tensors: idx / alphas [N,H,W,K], features [C,P] -> [N,C,H,W]."""

    @staticmethod
    def forward(ctx, idx, alphas, features, radius2=0.0):
        for t, name in ((idx, "idx"), (alphas, "alphas"), (features, "features")):
            L.require_cuda(t, name)
            L.require_contiguous(t, name)
        if idx.dtype != torch.int32 or alphas.dtype != torch.float32 or features.dtype != torch.float32:
            raise ValueError("alpha_composite: idx int32, alphas / features float32")
        if idx.shape != alphas.shape or idx.dim() != 4 or features.dim() != 2:
            raise ValueError("alpha_composite: idx / alphas [N,H,W,K], features [C,P]")
        N, H, W, K = idx.shape
        C, P = features.shape
        images = torch.empty((N, C, H, W), dtype=torch.float32, device=idx.device)
        with L.device_guard(idx.device):
            L.check(L.lib().recmv_alpha_composite_forward(L.ptr(idx), L.ptr(alphas), L.ptr(features), N, H, W, K, C, P,
                                                          float(radius2), L.ptr(images), L.stream_ptr(idx.device)),
                    "alpha_composite_forward")
        ctx.save_for_backward(idx, alphas, features)
        ctx.radius2 = float(radius2)
        return images

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_images):
        idx, alphas, features = ctx.saved_tensors
        N, H, W, K = idx.shape
        C, P = features.shape
        g_alphas = torch.empty_like(alphas)
        g_features = torch.empty_like(features) if ctx.needs_input_grad[2] else None
        with L.device_guard(idx.device):
            L.check(L.lib().recmv_alpha_composite_backward(
                L.ptr(idx), L.ptr(alphas), L.ptr(features), L.ptr(g_images.contiguous()), N, H, W, K, C, P,
                ctx.radius2, L.ptr(g_alphas), L.ptr(g_features) if g_features is not None else None,
                L.stream_ptr(idx.device)), "alpha_composite_backward")
        return None, g_alphas, g_features, None


def alpha_composite(idx, alphas, features):
    """pytorch3d `alpha_composite` on fragment-major tensors (idx / alphas [N,H,W,K], packed lists)."""
    return _AlphaComposite.apply(idx, alphas, features, 0.0)


def alpha_composite_dists(idx, dists, radius, features):
    """The compositor on the rasteriser's squared distances: opacities 1 - dists / radius^2 are formed inside the
    kernels (and their chain rule in the backward) instead of by four full-size element-wise passes."""
    return _AlphaComposite.apply(idx, dists, features, float(radius) * float(radius))


class PointsRendererWithFrags_Split:
    """`PointsRendererWithFrags_Split(PointsRasterizer, AlphaCompositor)` of the reference (model/CameraMine.py:347-415)
    for N clouds with the same number of points: every frame's cloud is [upper garment vertices ; bottom garment
    vertices]; returns one alpha-composited silhouette per garment — the other garment's points take part in the
    compositing with feature 0, so they occlude — and the fragments."""

    def __init__(self, cameras, image_size, radius, points_per_pixel=50):
        self.cameras = cameras
        self.image_size = (int(image_size[0]), int(image_size[1]))
        self.radius = float(radius)
        self.points_per_pixel = int(points_per_pixel)

    def __call__(self, points, split_size):
        N, V = points.shape[0], points.shape[1]
        H, W = self.image_size
        dev = points.device
        ndc = self.cameras.transform_points_ndc(points.reshape(-1, 3)).contiguous()
        first = torch.arange(N, device=dev, dtype=torch.int64) * V
        num = torch.full((N,), V, device=dev, dtype=torch.int64)
        frags = rasterize_points(ndc, first, num, (H, W), self.radius, self.points_per_pixel, max_points_per_cloud=V)
        upper = (torch.arange(N * V, device=dev) % V) < int(split_size)            # :367-372
        features = torch.stack([upper, ~upper]).to(torch.float32).contiguous()
        # weights = 1 - dists / r^2 (CameraMine.py:361-362) fused into the compositor
        images = alpha_composite_dists(frags.idx, frags.dists, self.radius, features)          # [N,2,H,W]
        return [images[:, 0, :, :, None], images[:, 1, :, :, None]], frags
