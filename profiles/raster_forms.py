#!/usr/bin/env python3
"""The optimisation loop's rasterise phase in its two forms on the bench scene (run on the GPU box from the repository root):
    python3 profiles/raster_forms.py > profiles/r2_raster_forms.txt
512 poses drawn like the bench's images (two workgroups per compute unit, as in the solve), 50 rasterisations each, cycles by
s_memtime inside the kernel; then the same with one workgroup per compute unit (256 poses)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from nemo_amd import ops  # noqa: E402
from nemo_amd.utils import camera_position_from_spherical_angles  # noqa: E402

dev = torch.device("cuda:0")
mod, mem, clutter, v, f = bench.build_scene(dev)
cam = mod.camera()
for n in (512, 256):
    g = torch.Generator().manual_seed(7)
    u = torch.rand(n, 3, generator=g)
    az, el, th = u[:, 0] * 2 * math.pi, -math.pi / 6 + u[:, 1] * math.pi / 2, -math.pi / 6 + u[:, 2] * math.pi / 3
    campos = camera_position_from_spherical_angles(5.0, el, az, device=dev)
    theta = th.float().to(dev)
    res = {}
    for form in (0, 1):
        ops.raster_forms(mod.mesh, cam, campos, theta, form, reps=5)            # warm-up
        cyc, binning, keys = ops.raster_forms(mod.mesh, cam, campos, theta, form, reps=50)
        torch.cuda.synchronize()
        res[form] = (cyc.cpu(), binning.cpu(), keys.cpu())
    k0, k1 = res[0][2], res[1][2]
    empty = 0xFFFFFFFF
    fbits = 11                                                                   # 2^11 > 1904 faces
    cov0, cov1 = k0 != empty, k1 != empty
    face_diff = int(((k0 & ((1 << fbits) - 1)) != (k1 & ((1 << fbits) - 1)))[cov0 & cov1].sum())
    print(f"{n} poses ({'two' if n > 256 else 'one'} workgroup(s) per compute unit), mesh V={v.shape[0]} F={f.shape[0]}, map {bench.H}x{bench.W}:")
    print(f"  covered pixels per pose: face-per-lane {float(cov0.sum()) / n:.1f}, tile-binned {float(cov1.sum()) / n:.1f}; "
          f"pixels covered by one form only {int((cov0 ^ cov1).sum())}, covered by both with a different face {face_diff} (of {int(cov0.sum())})")
    print(f"  face-per-lane (raster_direct): {float(res[0][0].mean()):.0f} cycles per rasterisation (min {float(res[0][0].min()):.0f}, max {float(res[0][0].max()):.0f})")
    print(f"  tile-binned   (raster_tiled):  {float(res[1][0].mean()):.0f} cycles per rasterisation (min {float(res[1][0].min()):.0f}, max {float(res[1][0].max()):.0f}), "
          f"of which binning {float(res[1][1].mean()):.0f}")
