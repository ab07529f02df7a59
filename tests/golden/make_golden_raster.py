"""Golden vectors for the rasteriser-side host logic, produced by the REAL reference Python code imported from
/root/reference (this container only; the fixtures travel):

  findsurface.npz  utils/FindSurfacePs.py:7-37 `FindSurfacePs` on (a) first-hit fragments of an MC sphere seen by the
                   reference camera convention (fragments from the C oracle rasteriser: pytorch3d itself is absent),
                   (b) synthetic K = 3 fragments with holes in the first layer (exercises the scatter-min path)
  camera_ndc.npz   model/CameraMine.py:210-300 `_get_sfm_calibration_matrix` (the reference's own override of the
                   pytorch3d helper): the 4x4 projection of a screen-space camera, and the NDC coordinates it gives

    python tests/golden/make_golden_raster.py
"""
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(HERE))
sys.path[:0] = [str(REPO / "rec-mv_amd"), str(REPO)]
import ref_loader  # noqa: E402

ref_loader.install()
from make_golden import save  # noqa: E402


def sphere_mesh(res=25, radius=0.45):
    from oracle import oracle as orc
    ax = torch.linspace(-0.6, 0.6, res)
    x, y, z = torch.meshgrid(ax, ax, ax, indexing="ij")
    sdf = (x * x + 1.3 * y * y + z * z).sqrt() - radius
    step = 1.2 / (res - 1)
    return orc.mc(sdf.contiguous(), step, step, step, -0.6, -0.6, -0.6, 0.0)


def main():
    from oracle import oracle as orc
    Fref = ref_loader.ref_module("utils.FindSurfacePs")
    Cref = ref_loader.ref_module("model.CameraMine")

    # ---------------------------------------------------------------- camera: the reference's calibration matrix
    W, H = 56, 72
    focal = torch.tensor([[1.8 * W, 1.7 * W]])
    pp = torch.tensor([[W / 2 - 0.5 + 3.25, H / 2 - 0.5 - 2.5]])
    K = Cref._get_sfm_calibration_matrix(1, "cpu", focal, pp, orthographic=False, image_size=torch.tensor([[W, H]]))
    R = torch.tensor([[[-1., 0., 0.], [0., 1., 0.], [0., 0., -1.]]])
    T = torch.tensor([[0.05, -0.02, 2.6]])
    g = torch.Generator().manual_seed(5)
    pts = 0.5 * torch.randn(64, 3, generator=g)
    view = pts @ R[0] + T[0]                                          # pytorch3d row-vector convention
    hom = torch.cat([view, torch.ones(64, 1)], 1) @ K[0].t()          # Transform3d._matrix = K^T, points @ matrix
    ndc = hom[:, :3] / hom[:, 3:4]
    # MeshRasterizer.transform (pytorch3d 0.4.0) keeps the view-space depth
    save("camera_ndc", W=W, H=H, focal=focal, pp=pp, R=R, T=T, K=K, pts=pts, ndc_xy=ndc[:, :2], view_z=view[:, 2])

    # ---------------------------------------------------------------- FindSurfacePs on first-hit fragments
    verts, faces = sphere_mesh()
    offs = torch.tensor([[0., 0., 0.], [0.07, -0.03, 0.2]])
    def_vs = verts[None] + offs[:, None]
    vflat = def_vs.reshape(-1, 3) @ R[0] + T[0]
    hom = torch.cat([vflat, torch.ones(vflat.shape[0], 1)], 1) @ K[0].t()
    vndc = torch.cat([hom[:, :2] / hom[:, 3:4], vflat[:, 2:3]], 1).view(2, -1, 3)
    F = faces.shape[0]
    fv = vndc[:, faces.reshape(-1)].reshape(-1, 3, 3)
    p2f, zbuf, bary, dists = orc.rasterize_meshes(fv, torch.tensor([0, F]), torch.tensor([F, F]), (H, W))

    class Frags:
        pass

    fr = Frags()
    fr.pix_to_face, fr.bary_coords = p2f, bary
    b, r, c, p0, finds = Fref.FindSurfacePs(verts, faces, fr)
    # (b) K = 3 with holes / non-inner first fragments
    g = torch.Generator().manual_seed(9)
    N3, H3, W3, K3 = 2, 12, 10, 3
    p2f3 = torch.randint(-1, 2 * F, (N3, H3, W3, K3), generator=g)
    bary3 = torch.rand(N3, H3, W3, K3, 3, generator=g) - 0.15        # some components negative -> not "inner"
    fr3 = Frags()
    fr3.pix_to_face, fr3.bary_coords = p2f3, bary3
    b3, r3, c3, p03, finds3 = Fref.FindSurfacePs(verts, faces, fr3)
    print("findsurface: %d hits (K=1), %d hits (K=3)" % (b.numel(), b3.numel()))
    save("findsurface", verts=verts, faces=faces, fv=fv, H=H, W=W, pix_to_face=p2f, bary=bary, batch=b, row=r, col=c,
         init=p0, finds=finds, pix_to_face3=p2f3, bary3=bary3, batch3=b3, row3=r3, col3=c3, init3=p03, finds3=finds3)


if __name__ == "__main__":
    main()
