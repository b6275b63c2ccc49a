#!/usr/bin/env python
"""Benchmark of the render-and-compare pose solver (BASELINE.json configs[1]).

One "step" = one pass of the hot path over one batch of synthetic images already resident in HBM:
clutter score -> pose-grid scoring (144 pre-rendered poses) -> 300 Adam steps for every image (the solve
computes the similarity-table tiles its covered pixels need itself) -> read-out (+ an RCCL all-gather of
the [B,7] results when N > 1).  The CNN backbone is outside the path: feature maps are synthetic
(seeded), the vertex bank is random-init.

    python bench.py --gpus N --steps K --warmup W            # BASELINE configs[1], weak scaling
    python bench.py --gpus N --strong --steps K --warmup W   # BASELINE configs[2]: 1024 images x 8 hypotheses, strong scaling

N > 1 is launched by torch.distributed.run (one rank per GPU); every rank solves its own batch of
`--batch` images (weak scaling; with --strong its 1/N share of 1024 images), the loss normaliser uses the
global batch size, rank 0 prints ONE JSON line.  `value` = hypothesis-optimisation steps per second of the
whole job; images/s is in `images_per_sec`.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W, C = 80, 100, 128           # 640x800 image at stride 8, 128 channels
FOCAL = 3000.0 / 8
EPOCHS = 300
STREAM_BYTES_PER_HYP_STEP = C * H * W * 4 + H * W * 4     # a design that streams the feature map + clutter map once per step
HBM_PEAK_GBS = 8000.0
MFMA_F16_PEAK_TFLOPS = 2500.0
HALF_EXTENTS = (0.225, 0.5, 0.18)     # the synthetic cuboid: half of a car's 0.45 / 1.0 / 0.36, so that it fits the 100-pixel map at distance 5
STRONG_TOTAL_IMAGES, STRONG_HYPOTHESES = 1024, 8


def build_scene(dev, seed=0, half_extents=None, number_vertices=1000):
    import numpy as np
    import nemo_amd
    from nemo_amd.utils import meshelize, normalize_features
    # cuboid mesh (V=1118, F=1904), vertices rounded through the OFF text format like the reference's files
    hx, hy, hz = half_extents or HALF_EXTENTS
    v, f = meshelize((-hx, hx), (-hy, hy), (-hz, hz), number_vertices)
    v = np.array([[float("%.16f" % x) for x in row] for row in v], dtype=np.float32)
    g = torch.Generator().manual_seed(seed)
    mem = torch.nn.functional.normalize(torch.randn(v.shape[0], C, generator=g), dim=1)
    rows = torch.nn.functional.normalize(torch.randn(2560, C, generator=g), dim=1)
    clutter = normalize_features(torch.mean(rows, dim=0)).unsqueeze(0)
    cams = nemo_amd.PerspectiveCameras(focal_length=FOCAL, principal_point=((W // 2, H // 2),), image_size=((H, W),),
                                       in_ndc=False)
    rs = nemo_amd.RasterizationSettings(image_size=(H, W), blur_radius=0.0, faces_per_pixel=1, bin_size=0)
    mod = nemo_amd.MeshInterpolateModule(torch.from_numpy(v), torch.from_numpy(f.astype(np.int32)), mem,
                                         rasterizer=nemo_amd.MeshRasterizer(cameras=cams, raster_settings=rs)).to(dev)
    return mod, mem, clutter.to(dev), v, f


def synthetic_images(mod, clutter, B, first_seed, dev):
    """Feature maps of B synthetic images: the bank rendered at a seeded pose, clutter direction on the
    background, gaussian noise of norm ~0.5, L2-normalised over channels (the backbone's contract)."""
    from nemo_amd.utils import camera_position_from_spherical_angles
    g = torch.Generator().manual_seed(1000 + first_seed)
    u = torch.rand(B, 3, generator=g)
    az, el, th = u[:, 0] * 2 * math.pi, -math.pi / 6 + u[:, 1] * math.pi / 2, -math.pi / 6 + u[:, 2] * math.pi / 3
    campos = camera_position_from_spherical_angles(5.0, el, az, device=dev)
    with torch.no_grad():
        maps = mod(campos, th.float().to(dev))                                 # [B,C,H,W]
    bg = (maps.abs().sum(dim=1, keepdim=True) == 0).float()
    maps = maps + bg * clutter.view(1, C, 1, 1)
    maps = maps + 0.5 / math.sqrt(C) * torch.randn(maps.shape, generator=torch.Generator(device=dev).manual_seed(2000 + first_seed), device=dev)
    return torch.nn.functional.normalize(maps, dim=1).contiguous(), dict(azimuth=az, elevation=el, theta=th)


def cpu_baseline(mod, mem, clutter, v, f, fm0, maps_dev, cs, ths, n_steps=EPOCHS):
    """The oracle (PyTorch-CPU restatement of the reference's PyTorch3D CPU path) on this host's cores,
    on a bounded sample of the same workload: one whole image (grid scoring + all Adam steps) at B=1, about
    ten seconds of CPU work."""
    from oracle import refsolver as R
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 16)))
    cores = torch.get_num_threads()
    ref = R.MeshInterpolateModuleRef(v, f, mem, R.Camera(H, W, FOCAL))
    fm = fm0.cpu()[None]
    cl = clutter.cpu()
    maps = maps_dev.cpu()
    t0 = time.time()
    clu = R.clutter_score(fm, cl)
    C0, th0, _, _ = R.get_init_pos_rendered_dim0(maps, cs.cpu(), ths.cpu(), fm, clu)
    t_grid = time.time() - t0
    t0 = time.time()
    R.optimize_poses(ref, fm, clu, C0, th0, epochs=n_steps)
    t_opt = (time.time() - t0) / n_steps
    per_image = t_grid + EPOCHS * t_opt
    return dict(value=1.0 / t_opt, unit="hypothesis-steps/s", cores=cores, kind="port",
                sample=f"oracle/refsolver.py at B=1, {H}x{W}x{C}, V={v.shape[0]} F={f.shape[0]}: grid scoring of 1 image x "
                       f"{maps.shape[0]} poses ({t_grid:.2f} s) + {n_steps} Adam steps ({t_opt:.3f} s each); "
                       f"images/s = 1 / (grid + {EPOCHS} steps)",
                images_per_sec=1.0 / per_image)


def _median_ms(fn, reps=3):
    """fn() once un-timed, then `reps` times between device synchronisations -> (median ms, min ms)."""
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2], min(ts)


def _event_ms(fn, reps=3):
    """Median GPU time (ms) of fn() between two HIP events on the current stream."""
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


OTHER_CONFIGS = ("latency", "k8", "six_d", "classification", "clipped", "fast_inference")


def other_configs(dev, cfg, mod, clutter, maps, cs, ths, packed, batch=64, which=OTHER_CONFIGS):
    """The other BASELINE configs on this GPU, a few calls each, through the drop-in entry points (not the headline's pipelined
    loop): every figure is one call at a time, median of three.  Returns the `other_configs` object of the JSON line."""
    import numpy as np
    import nemo_amd
    from nemo_amd import batch_solve_pose as bsp
    from nemo_amd import ops
    from nemo_amd.utils import ConfigNode, default_pose3d_config, get_param_samples, meshelize, normalize_features
    out = {}
    hw = H * W
    fm64, _ = synthetic_images(mod, clutter, batch, first_seed=5000, dev=dev)

    # -- configs[1], one lone batch: what a caller of solve_pose waits for
    if "latency" in which:
        med, mn = _median_ms(lambda: nemo_amd.solve_pose(cfg, fm64, mod, clutter, cs, ths, maps, device=dev), reps=5)
        out["single_batch_latency_ms"] = {"median": med, "min": mn, "images": batch,
                                          "call": "nemo_amd.solve_pose (clutter + 144-pose grid + complete table + 300 steps + read-out)"}

    # -- configs[2]: 8 starting poses per image (the 8 best grid poses), 128 images per call = 1024 hypotheses in one solve launch
    if "k8" in which:
        n2 = 128
        fm2 = torch.cat([fm64, synthetic_images(mod, clutter, n2 - batch, first_seed=5100, dev=dev)[0]]) if n2 > batch else fm64
        med, _ = _median_ms(lambda: nemo_amd.solve_pose(cfg, fm2, mod, clutter, cs, ths, maps, device=dev, hypotheses=STRONG_HYPOTHESES))
        t_tab = _event_ms(lambda: ops.similarity_table(mod.mesh, fm2))
        out["configs[2] multi-hypothesis"] = {
            "workload": f"{n2} images x {STRONG_HYPOTHESES} starting poses x {EPOCHS} steps per call (solve_pose(hypotheses=8)), one GPU",
            "ms_per_call": med, "images_per_sec": n2 / med * 1e3, "hypothesis_steps_per_sec": n2 * STRONG_HYPOTHESES * EPOCHS / med * 1e3,
            "dominant_kernel": "solve_kernel on the images' complete similarity tables (the 8 hypotheses of an image share it)",
            "similarity_table_ms": t_tab,
            "similarity_table_hbm_frac": n2 * 4.0 * math.prod(ops.table_shape(mod.mesh, 1, hw)) / (t_tab * 1e-3) / 1e9 / HBM_PEAK_GBS}
        del fm2

    # -- configs[3]: 6-D, 576 poses (4 distances) x 25 principal-point samples, 64 images per call
    if "six_d" in which:
        cfg6 = default_pose3d_config()
        cfg6.inference.dist_sample = ConfigNode({"min": 3.0, "max": 9.0, "num": 4})
        cfg6.inference.px_sample = ConfigNode({"min": 0.25, "max": 0.75, "num": 5})
        cfg6.inference.py_sample = ConfigNode({"min": 0.25, "max": 0.75, "num": 5})
        az, el, th, di, px, py = get_param_samples(cfg6)
        maps6, cs6, ths6 = nemo_amd.get_pre_render_samples(mod, az, el, th, di)
        cand = torch.stack((torch.from_numpy(px).view(-1, 1).expand(-1, py.shape[0]) * W,
                            torch.from_numpy(py).view(1, -1).expand(px.shape[0], -1) * H)).view(2, -1).T.float().to(dev)
        call6 = lambda: nemo_amd.solve_pose(cfg6, fm64, mod, clutter, cs6, ths6, maps6, device=dev, principal=cand, dof=6)
        med, _ = _median_ms(call6)
        clu64 = ops.clutter_score(fm64, clutter)
        t_init = _event_ms(lambda: bsp.initial_poses(fm64, clu64, mod.rasterizer.cameras, cs6, ths6, maps6, principal=cand, dof=6))
        n_pose6, n_pr = int(cs6.shape[0]), int(cand.shape[0])
        flops6 = 2.0 * batch * n_pr * n_pose6 * C * hw
        out["configs[3] 6-D"] = {
            "workload": f"{batch} images, {n_pose6}-pose grid x {n_pr} principal-point samples, then {EPOCHS} steps over 6 parameters "
                        "(solve_pose(dof=6)), one GPU",
            "ms_per_call": med, "images_per_sec": batch / med * 1e3,
            "dominant_kernel": "grid scoring of the re-centred copies (grid_feat_recentred_kernel + grid_mfma_kernel)",
            "dominant_kernel_ms": t_init,
            "dominant_kernel_roofline": {"bound": "mfma", "achieved": flops6 / (t_init * 1e-3) / 1e12, "peak": MFMA_F16_PEAK_TFLOPS,
                                         "unit": "TFLOP/s", "frac": flops6 / (t_init * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS,
                                         "note": "useful flops 2 B P N C HW; three f16 piece products per useful one"}}
        del maps6, cs6, ths6

    # -- configs[4]: render-and-compare classification, 12 category cuboids in one multi-class bank, 128 images per call
    if "classification" in which:
        n_cat, n4 = 12, 128
        rng = np.random.RandomState(0)
        meshes = []
        for i in range(n_cat):
            ex = 0.14 + 0.12 * rng.rand(3)
            ex[1] += 0.15
            v_, f_ = meshelize((-ex[0], ex[0]), (-ex[1], ex[1]), (-ex[2], ex[2]), 900 + 20 * i)
            meshes.append((np.asarray(v_, dtype=np.float32), f_.astype(np.int32)))
        counts = [m_[0].shape[0] for m_ in meshes]
        g = torch.Generator().manual_seed(11)
        bank = nemo_amd.FeatureBankNeMo(input_size=C, num_pos=sum(counts), num_noise=0, mesh_n_list=counts)
        bank.load_memory(torch.nn.functional.normalize(torch.randn(sum(counts), C, generator=g), dim=1))
        solver = nemo_amd.MultiCategorySolver(cfg, [f"cat{i}" for i in range(n_cat)], meshes, bank, clutter.cpu(), mod.rasterizer,
                                              device=str(dev), dof=3)
        fm4 = torch.cat([synthetic_images(solver.modules[i % n_cat], clutter, 1, first_seed=5200 + i, dev=dev)[0] for i in range(16)])
        fm4 = fm4.repeat(n4 // 16, 1, 1, 1).contiguous()
        res = {}
        def call4():
            res["names"], res["scores"], _ = solver.classify(fm4)
        med, _ = _median_ms(call4)
        right = sum(res["names"][i] == f"cat{(i % 16) % n_cat}" for i in range(n4))
        out["configs[4] classification"] = {
            "workload": f"{n4} images x {n_cat} category meshes ({min(counts)}-{max(counts)} vertices each, one multi-class bank): 144-pose "
                        f"grid + {EPOCHS} steps per (image, category) pair, all {n4 * n_cat} pairs in one solve launch "
                        "(MultiCategorySolver.classify), one GPU",
            "ms_per_call": med, "images_per_sec": n4 / med * 1e3, "hypothesis_steps_per_sec": n4 * n_cat * EPOCHS / med * 1e3,
            "correct_of_synthetic": f"{right}/{n4}",
            "dominant_kernel": "solve_kernel with a mesh part per hypothesis (nemo_solve_features on the multi-mesh handle)"}
        del solver, fm4

    # -- the full-size cuboid of a car (twice the headline's extents: it overflows the 100-pixel map at distance 5)
    if "clipped" in which:
        hx, hy, hz = HALF_EXTENTS
        big, _, _, vb_, fb_ = build_scene(dev, half_extents=(2 * hx, 2 * hy, 2 * hz))
        mapsb, csb, thsb = nemo_amd.get_pre_render_samples(big, *get_param_samples(cfg)[:4])
        nb = 512
        fmb = torch.cat([synthetic_images(big, clutter, batch, first_seed=5300 + batch * k, dev=dev)[0] for k in range(nb // batch)])
        packedb = mapsb._nemo_packed_grid
        wsb = torch.empty(ops.solve_workspace_bytes(big.mesh, nb, hw), dtype=torch.uint8, device=dev)
        camb = big.camera()
        def callb():
            clu_, _, idx_, _ = ops.clutter_and_grid_score(fmb, clutter, packedb)
            ops.solve_from_features(big.mesh, camb, fmb, clu_, csb[idx_.long()], thsb[idx_.long()], workspace=wsb, global_batch=batch,
                                    steps=EPOCHS, lr_every=EPOCHS // 3)
        t_big = _event_ms(callb)
        with torch.no_grad():
            cov = float((mapsb.squeeze(1).abs().sum(dim=1) != 0).float().mean())
        out["clipped-object scene"] = {
            "workload": f"configs[1] with a cuboid of half extents {2 * hx}/{2 * hy}/{2 * hz} (V={vb_.shape[0]} F={fb_.shape[0]}): the object "
                        f"overflows the map at distance 5 (mean coverage of the grid poses {cov:.2f} of the map), {nb} images in one launch",
            "ms_per_launch": t_big, "images_per_sec": nb / t_big * 1e3, "hypothesis_steps_per_sec": nb * EPOCHS / t_big * 1e3}
        del big, mapsb, fmb, wsb

    # -- NeMo.fast_inference: per-pixel maximum over the vertices of the bank x feature-map product, mean over the box
    if "fast_inference" in which:
        nf = 256
        fmf = fm64.repeat(nf // batch, 1, 1, 1).contiguous()
        def callf():
            ops.vertex_similarity_from_features(mod.mesh, fmf, ops.clutter_score(fmf, clutter))
        t_f = _event_ms(callf)
        flops_f = 2.0 * nf * mod.mesh.V * C * hw
        out["fast_inference"] = {
            "workload": f"{nf} images, whole map as the box: clutter score + vertex-maximum similarity straight from the feature maps "
                        "(no table written)",
            "ms_per_call": t_f, "images_per_sec": nf / t_f * 1e3,
            "roofline": {"bound": "mfma", "achieved": flops_f / (t_f * 1e-3) / 1e12, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": flops_f / (t_f * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS,
                         "note": "useful flops 2 V C HW per image; three f16 piece products per useful one"}}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 256 (8 with --strong)")
    ap.add_argument("--warmup", type=int, default=None, help="default 16 (1 with --strong)")
    ap.add_argument("--batch", type=int, default=64, help="images per GPU (BASELINE config 2: 64)")
    ap.add_argument("--in-flight", type=int, default=None,
                    help="batches processed together: their kernels are launched once over all their images (a 64-image solve launch "
                         "occupies an eighth of the chip, and the hardware refills compute units best inside one large launch); every "
                         "batch keeps its own loss normaliser. Default: all of them in one launch up to 32 steps, 16 beyond")
    ap.add_argument("--hypotheses", type=int, default=1,
                    help="initial poses per image (the K best grid poses; BASELINE configs[2] uses 8); 1 = configs[1]")
    ap.add_argument("--streams", type=int, default=2, help="launch streams the groups alternate over")
    ap.add_argument("--complete-table", action="store_true",
                    help="compute the complete vertex x pixel similarity table in front of the solve (nemo_similarity_table + "
                         "nemo_solve) instead of letting the solve compute the tiles it needs (nemo_solve_features). Default: "
                         "tiles on demand with one or two hypotheses per image, the complete table with more (they share it)")
    ap.add_argument("--on-demand-table", action="store_true", help="tiles on demand inside the solve whatever the number of hypotheses")
    ap.add_argument("--no-bulk-tiles", action="store_true",
                    help="tiles on demand: every tile inside the loop (default: the first step's tiles and a ring of two pixels around "
                         "them are listed by a launch that only looks and computed by one kernel in front of the loop)")
    ap.add_argument("--strong", action="store_true",
                    help=f"BASELINE configs[2]: {STRONG_TOTAL_IMAGES} images x {STRONG_HYPOTHESES} hypotheses in total, sharded over the ranks "
                         "(strong scaling: a step is one pass over all of them); overrides --batch, --hypotheses and --in-flight")
    ap.add_argument("--separate-clutter", action="store_true",
                    help="clutter score and grid scoring as two calls (nemo_clutter_score + nemo_grid_score_packed) instead of one read of "
                         "the feature maps for both (nemo_clutter_grid_score_packed, the default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the extra measurements after the timed region (the other BASELINE configs, one lone batch, repeats)")
    ap.add_argument("--other-configs", default=",".join(OTHER_CONFIGS),
                    help="which of the extra measurements to run (comma list of " + ", ".join(OTHER_CONFIGS) + ")")
    ap.add_argument("--repeats", type=int, default=5, help="un-scored repetitions of the timed region reported as min / median")
    ap.add_argument("--no-stagger", action="store_true",
                    help="let the groups on different launch streams start together (default: a group's clutter / grid / table "
                         "kernels start when the previous group's have finished, i.e. beside the previous group's solve)")
    args = ap.parse_args()

    if args.steps is None:
        args.steps = 8 if args.strong else 256
    if args.warmup is None:
        args.warmup = 1 if args.strong else 16
    if args.in_flight is None:
        args.in_flight = args.steps if args.steps <= 32 else 16
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch through python -m torch.distributed.run --nproc-per-node N")
    if args.strong:
        if STRONG_TOTAL_IMAGES % world:
            raise SystemExit(f"--strong shards {STRONG_TOTAL_IMAGES} images: the number of ranks ({world}) must divide it")
        args.batch, args.hypotheses, args.in_flight = STRONG_TOTAL_IMAGES // world, STRONG_HYPOTHESES, 1
    if args.complete_table and args.on_demand_table:
        raise SystemExit("--complete-table and --on-demand-table exclude each other")
    if not args.on_demand_table and args.hypotheses > 2:
        args.complete_table = True          # the hypotheses of an image share its table; on demand each would compute its own tiles
    # NEMO_BENCH_REHEARSAL=1: several ranks share the visible GPUs and talk over gloo, to rehearse the N > 1
    # code path on a one-GPU box (RCCL refuses two ranks on one device).  Never set by the driver.
    rehearsal = os.environ.get("NEMO_BENCH_REHEARSAL") == "1"
    dev = torch.device(f"cuda:{local_rank % torch.cuda.device_count() if rehearsal else local_rank}")
    torch.cuda.set_device(dev)
    import torch.distributed as dist
    # under torch.distributed.run (RANK set) the process group is created even for a single rank, so that the gather
    # of the results takes the RCCL path whatever the number of ranks
    grouped = world > 1 or "RANK" in os.environ
    if grouped:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import nemo_amd
    from nemo_amd import ops
    from nemo_amd.utils import default_pose3d_config, get_param_samples
    cfg = default_pose3d_config()
    mod, mem, clutter, v, f = build_scene(dev)
    az, el, th, di, _, _ = get_param_samples(cfg)
    maps, cs, ths = nemo_amd.get_pre_render_samples(mod, az, el, th, di)     # one-off, untimed (like the reference's init)
    pre = maps.squeeze(1).contiguous()
    packed = getattr(maps, "_nemo_packed_grid", None) or ops.grid_pack(pre)   # one-off, like the grid itself
    B = args.batch
    cam = mod.camera()
    G = max(1, args.in_flight)
    # G batches of B images are in flight together: their kernels are launched once over G*B images
    # (every batch keeps its own loss normaliser B), so that a 300-step solve launch has G*B workgroups
    # for the 256 CUs instead of B.
    parts = [synthetic_images(mod, clutter, B, first_seed=(rank * G + g) * B, dev=dev) for g in range(G)]   # G * B distinct images
    fm_g = torch.cat([p_[0] for p_ in parts]) if G > 1 else parts[0][0]
    gts = parts[0][1]
    del parts
    # Two launch streams, alternating groups: the front kernels of the next group (clutter, grid, table) fill the
    # compute units that the workgroups of a solve launch leave as they finish at different times.
    K_HYP = max(1, args.hypotheses)
    hyp_image = torch.arange(G * B, device=dev, dtype=torch.int32).repeat_interleave(K_HYP)
    n_streams = max(1, min(args.streams, -(-args.steps // G)))      # no more streams (and sets of buffers) than groups
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
    # per launch stream: room for the similarity tables (complete, or the tiles the solve computes for itself) + feature fragments
    work = [torch.empty(ops.solve_workspace_bytes(mod.mesh, G * B, H * W), dtype=torch.uint8, device=dev) for _ in range(n_streams)]
    tables = [ops.workspace_tables(mod.mesh, w_, G * B, H * W) for w_ in work]
    gathered = [[torch.empty(G * B, 7, device=dev) for _ in range(world)] for _ in range(n_streams)] if grouped else None
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    kern_ms = {"clutter_score": 0.0, "grid_score": 0.0, "similarity_table": 0.0, "solve": 0.0}   # fused: all of the first two under grid_score
    grid_ws = [torch.empty(ops.grid_score_workspace_bytes(packed, G * B), dtype=torch.uint8, device=dev) for _ in range(n_streams)]

    # Staggered groups: the front kernels (clutter + grid scoring, complete table: HBM / matrix-core bound) of group k+1 wait
    # for the front kernels of group k, so they run beside solve(k) (vector-issue / latency bound) instead of beside front(k).
    # (Measured and dropped: one stream for all front kernels and one of higher priority for all solves. The front kernels
    # then run ahead into the middle of a solve launch, both slow down: 35 k images/s against 50 k.)
    front_ev = [None]
    stagger = not args.no_stagger and n_streams > 1

    trace = [] if os.environ.get("NEMO_BENCH_TRACE") == "1" else None     # diagnostic: per-group events of the timed region

    def front_part(n, slot, timed):
        """clutter score, grid scoring and the choice of the starting pose(s); the complete table if asked for."""
        x = fm_g[:n]
        if timed:
            ev[0].record()
        if args.separate_clutter:
            clu = ops.clutter_score(x, clutter)
            if timed:
                ev[1].record()
            loss, idx, _ = ops.grid_score(pre, x, clu, packed=packed)
        else:
            if timed:
                ev[1].record()
            clu, loss, idx, _ = ops.clutter_and_grid_score(x, clutter, packed, workspace=grid_ws[slot])
        h2i = None
        if K_HYP > 1:       # the K best grid poses of every image, best first (as solve_pose(..., hypotheses=K))
            idx = torch.sort(loss, dim=1, stable=True)[1][:, :K_HYP].reshape(-1)
            h2i = hyp_image[:n * K_HYP]
        idx = idx.long()
        if timed:
            ev[2].record()
        tab = ops.similarity_table(mod.mesh, x, out=tables[slot][:n]) if args.complete_table else None
        return x, clu, idx, h2i, tab

    def solve_part(g, n, fr, slot, timed, host_out, gather):
        """300 Adam steps per hypothesis, read-out, gather over the ranks, copy to the host."""
        x, clu, idx, h2i, tab = fr
        skw = dict(global_batch=B * world, dof=3, steps=EPOCHS, lr=0.05, betas=(0.4, 0.6), lr_gamma=0.2, lr_every=EPOCHS // 3)
        if args.complete_table:
            pose, _ = ops.solve(mod.mesh, cam, tab, clu, cs[idx], ths[idx], hyp_to_image=h2i, **skw)
        else:           # the table tiles are computed inside the solve, when a covered pixel first needs one
            pose, _ = ops.solve_from_features(mod.mesh, cam, x, clu, cs[idx], ths[idx], hyp_to_image=h2i, workspace=work[slot],
                                               bulk_first_step=not args.no_bulk_tiles, **skw)
        if timed:
            ev[4].record()
        done = None
        if trace is not None and host_out is not None:
            done = torch.cuda.Event(enable_timing=True)
            done.record()
        if K_HYP > 1:       # keep the best-scoring hypothesis of every image
            pose = pose.view(n, K_HYP, 8)
            pose = pose[torch.arange(n, device=dev), pose[:, :, 6].argmin(dim=1)]
        sph = ops.spherical_angles(pose[:, 0:3])
        res = torch.cat([sph[:, 2:3], sph[:, 1:2], pose[:, 3:4], sph[:, 0:1], pose[:, 4:7]], dim=1)   # az, el, th, d, px, py, score
        if grouped and gather:
            res = res.contiguous()
            if rehearsal:                                     # gloo gathers host tensors
                part = [torch.empty(n, 7) for _ in range(world)]
                dist.all_gather(part, res.cpu())
                res = torch.cat(part).to(dev)
            elif g == G:
                dist.all_gather(gathered[slot], res)
                res = torch.cat(gathered[slot])
            else:
                part = [torch.empty(n, 7, device=dev) for _ in range(world)]
                dist.all_gather(part, res)
                res = torch.cat(part)
        if host_out is not None:                    # the results go to (pinned) host memory per group
            host_out.copy_(res, non_blocking=True)
        return res, done

    def one_group(g, timed=False, host_out=None, slot=0, gather=True):
        """g batches (g <= G) through the whole path; returns [world * g * B, 7] (rank-major).
        slot: which stream's buffers to use (the caller has made that stream current)."""
        n = g * B
        tracing = trace is not None and host_out is not None
        tr = [torch.cuda.Event(enable_timing=True) for _ in range(2)] if tracing else None
        if stagger and front_ev[0] is not None:
            torch.cuda.current_stream().wait_event(front_ev[0])
        if tracing:
            tr[0].record()
        fr = front_part(n, slot, timed)
        if stagger:
            front_ev[0] = torch.cuda.Event()
            front_ev[0].record()
        if tracing:
            tr[1].record()
        if timed:
            ev[3].record()
        res, done = solve_part(g, n, fr, slot, timed, host_out, gather)
        if tracing:
            trace.append((g, slot, tr + [done]))
        return res

    def sync():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    # un-timed: per-kernel durations of one group (HIP events on the launch stream), then the warm-up batches
    n_probe = 3
    for i in range(n_probe):
        one_group(G, timed=True)
        torch.cuda.current_stream().synchronize()
        if i > 0:
            for j, k in enumerate(list(kern_ms)):
                kern_ms[k] += ev[j].elapsed_time(ev[j + 1]) / (n_probe - 1)
    groups = [G] * (args.steps // G) + ([args.steps % G] if args.steps % G else [])
    # un-timed set-up: every launch stream sees one group of every size the timed region will give it, so that the
    # memory pools torch keeps per stream are filled before the clock starts (a first allocation on a stream is a
    # device-wide synchronisation in the middle of the pipeline)
    def run_group(k, g, host_out=None):
        with torch.cuda.stream(streams[k % n_streams]):
            return one_group(g, host_out=host_out, slot=k % n_streams)

    for slot, g in sorted({(k % n_streams, g) for k, g in enumerate(groups)}):
        run_group(slot, g)
    sync()
    done, k = 0, 0
    while done < args.warmup:
        g = min(G, args.warmup - done)
        run_group(k, g)
        done += g
        k += 1
    host_res = [torch.empty(world * g * B, 7).pin_memory() for g in groups]
    sync()
    if trace is not None:
        trace_t0 = torch.cuda.Event(enable_timing=True)
        trace_t0.record()
    # timed: exactly `steps` batches of B images per GPU, G at a time; every result is copied to the host
    t0 = time.perf_counter()
    for k, (g, out) in enumerate(zip(groups, host_res)):
        run_group(k, g, host_out=out)
    sync()
    elapsed = time.perf_counter() - t0
    if trace is not None and rank == 0:
        for g, slot, tr in trace:
            print(f"trace: group of {g} on stream {slot}: front starts {trace_t0.elapsed_time(tr[0]):.2f} ms, solve queued behind "
                  f"{trace_t0.elapsed_time(tr[1]):.2f} ms, solve ends {trace_t0.elapsed_time(tr[2]):.2f} ms", file=sys.stderr)
        print(f"trace: timed region {elapsed * 1e3:.2f} ms", file=sys.stderr)
    res = host_res[0][:B].clone()
    if rank == 0 and os.environ.get("NEMO_BENCH_DUMP"):       # diagnostic: the gathered rows of the first group, rank-major
        import numpy as np
        np.save(os.environ["NEMO_BENCH_DUMP"], host_res[0].numpy())
    # the same region again (not scored): its spread
    rep_ms = []
    if not args.no_other_configs:
        for _ in range(max(0, args.repeats)):
            sync()
            t1 = time.perf_counter()
            for k, (g, out) in enumerate(zip(groups, host_res)):
                run_group(k, g, host_out=out)
            sync()
            rep_ms.append((time.perf_counter() - t1) * 1e3)
    if grouped:
        tmax = torch.tensor([elapsed], device="cpu" if rehearsal else dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax)

    if rank == 0:
        host = res
        # accuracy of the run against the synthetic ground truth (sanity, rank 0's own images)
        from nemo_amd.utils import camera_position_from_spherical_angles, pose_error
        errs = [pose_error(dict(azimuth=float(gts["azimuth"][i]), elevation=float(gts["elevation"][i]), theta=float(gts["theta"][i])),
                           dict(azimuth=float(host[i, 0]), elevation=float(host[i, 1]), theta=float(host[i, 2]))) for i in range(B)]
        med_err = sorted(errs)[len(errs) // 2] * 180 / math.pi
        n_hyp_steps = args.steps * B * world * EPOCHS * K_HYP
        n_img, n_hyp = G * B, G * B * K_HYP
        fm_bytes = C * H * W * 4

        # ---- what one solve launch has to move (un-timed census of one group on this GPU) ----
        # per hypothesis-step: one 16-byte record per covered pixel (three table entries + the clutter score); per hypothesis:
        # the table tiles it computes (written once, 4 KB each) and the feature columns of the pixel tiles they lie in (C x 32 x 4 B);
        # once per launch: the vertex bank's matrix-core operand.  Covered pixels are counted at the final poses.
        one = one_group(G, gather=False)                         # [n_img, 7] az, el, th, d, px, py, score
        torch.cuda.synchronize()
        loc = one[:n_img]
        with torch.no_grad():
            campos = camera_position_from_spherical_angles(loc[:, 3], loc[:, 1], loc[:, 0], device=dev)
            covered = 0
            for i0 in range(0, n_img, 64):
                covered += int((mod(campos[i0:i0 + 64], loc[i0:i0 + 64, 2].contiguous()).abs().sum(dim=1) != 0).sum())
        covered *= K_HYP                                         # every hypothesis of an image taken as its best one
        tiles = pixel_tiles = None
        solve_algo = EPOCHS * covered * 16
        if not args.complete_table:
            tables[0][:n_img].fill_(float("nan"))
            one_group(G, gather=False)
            torch.cuda.synchronize()
            tiles = pixel_tiles = 0
            for i0 in range(0, n_img, 64):                       # (in slices: the masks of a whole group would take tens of GB)
                filled = ~torch.isnan(tables[0][i0:min(i0 + 64, n_img)])      # [., HW/128, V/4, 128, 4]
                tiles += int(filled.sum(dtype=torch.int64)) // 1024
                pixel_tiles += int(filled.any(dim=4).any(dim=2).view(filled.shape[0], -1, 4, 32).any(dim=3).sum())
                del filled
            bank_bytes = -(-v.shape[0] // 32) * 32 * C * 2 * 2   # two f16 pieces per value
            solve_algo += tiles * 4096 + pixel_tiles * C * 32 * 4 + bank_bytes
        solve_ms = kern_ms["solve"]
        achieved = solve_algo / (solve_ms * 1e-3) / 1e9

        # ---- counters of the same launch shape from the committed rocprofv3 --pmc passes (bench.py cannot profile itself) ----
        traffic = valu = grid_mfma_busy = None
        import glob
        counters, exact = None, False
        for cpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]_solve_counters*.json")), reverse=True):
            with open(cpath) as fh:
                cj = json.load(fh)
            if cj.get("adam_steps") != EPOCHS or cj.get("on_demand_table") != (not args.complete_table):
                continue
            if not args.complete_table and bool(cj.get("tile_bulk")) != (not args.no_bulk_tiles):
                continue
            if cj.get("hypotheses_per_launch") == n_hyp:
                counters, exact, csrc = cj, True, os.path.relpath(cpath, ROOT)
                break
            if counters is None:
                counters, csrc = cj, os.path.relpath(cpath, ROOT)
        if counters is not None:
            # the same kernel on the same scene does the same work per hypothesis: a profile of another launch size is scaled
            cj, scale = counters, n_hyp / counters["hypotheses_per_launch"]
            how = "" if exact else f", scaled from a launch of {cj['hypotheses_per_launch']} hypotheses"
            # the solve call = the loop and, with the first step's tiles in bulk, the launch that lists them and the kernel that computes them
            calls = [cj] + [cj[k] for k in ("footprint", "tile_bulk") if cj.get(k) and not args.no_bulk_tiles]
            traffic = sum(c_["fetch_size_kb_per_launch"] + c_["write_size_kb_per_launch"] for c_ in calls) * 1024.0 * scale
            insts = sum(c_["sq_insts_valu_per_launch"] for c_ in calls) * scale
            # vector issue: a SIMD issues one wave-instruction per 2 cycles (64 lanes over a 32-wide unit)
            peak_issue = cj["compute_units"] * 4 * cj["shader_clock_ghz"] * 1e9 / 2
            valu = {"wave_instructions_per_launch": insts, "peak_per_sec": peak_issue,
                    "frac": insts / (solve_ms * 1e-3) / peak_issue,
                    "model": f"{cj['compute_units']} CUs x 4 SIMDs x {cj['shader_clock_ghz']} GHz / 2 cycles per wave-instruction",
                    "source": f"{csrc} (rocprofv3 --pmc SQ_INSTS_VALU{how})"}
            traffic_source = f"{csrc} (rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE, separate passes, bytes per launch{how})"
            grid_mfma_busy = cj.get("grid_mfma", {}).get("mfma_busy_frac")
        else:
            traffic_source = None

        # ---- the kernels in front of the solve: algorithmic bytes / flops of one launch over n_img images ----
        front = {}
        if args.separate_clutter:
            front["clutter_score"] = {"bound": "hbm", "algorithmic": n_img * (fm_bytes + H * W * 4)}     # maps read, scores written
        n_poses = int(pre.shape[0])
        # grid scoring: 2 B N C HW useful flops (the matrix cores execute three f16 piece products for each)
        front["grid_score"] = {"bound": "mfma", "algorithmic": 2.0 * n_img * n_poses * C * H * W}
        if args.complete_table:
            table_bytes = 4 * math.prod(ops.table_shape(mod.mesh, 1, H * W))
            front["similarity_table"] = {"bound": "hbm", "algorithmic": n_img * (table_bytes + fm_bytes)}   # table written once, maps read once
        other = {}
        for k, d in front.items():
            rate = d["algorithmic"] / (kern_ms[k] * 1e-3)
            if d["bound"] == "hbm":
                other[k] = {"bound": "hbm", "achieved": rate / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rate / 1e9 / HBM_PEAK_GBS,
                            "algorithmic_bytes_per_launch": d["algorithmic"]}
            else:
                other[k] = {"bound": "mfma", "achieved": rate / 1e12, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": rate / 1e12 / MFMA_F16_PEAK_TFLOPS, "algorithmic_flops_per_launch": d["algorithmic"],
                            "matrix_pipe_busy_frac_of_grid_mfma_kernel": grid_mfma_busy,
                            "note": "useful flops 2 B N C HW; the two-piece f16 operands cost three matrix-core products per useful one, and the "
                                    "time includes the pass that turns the feature maps into operand fragments (and, unless --separate-clutter, "
                                    "scores the clutter vectors) and the finalize pass"}
        # The solve call is bound by vector-instruction issue and by the latencies between its barriers, not by HBM: the
        # primary figure is the issue rate (when the counter profile of this launch shape is at hand), the HBM figure the
        # secondary one.
        hbm_roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "algorithmic_bytes_per_launch": solve_algo,
                    "note": "algorithmic bytes of the implemented design: 300 steps x covered pixels x 16 B (three table entries + clutter "
                            "score per covered pixel) + computed table tiles x 4 KB + touched pixel tiles x 16 KB of feature map + the "
                            "bank operand, / the launch time measured with HIP events on the launch stream"}
        roof = {"kernel": "solve_kernel" if args.complete_table or args.no_bulk_tiles else
                          "solve_kernel (the solve call: the loop, the launch that lists its first step's tiles and tile_bulk_kernel)"}
        if valu is not None:
            roof.update({"bound": "valu_issue", "achieved": valu["wave_instructions_per_launch"] / (solve_ms * 1e-3) / 1e9,
                         "peak": valu["peak_per_sec"] / 1e9, "unit": "G wave-instructions/s", "frac": valu["frac"],
                         "model": valu["model"], "source": valu["source"], "hbm": hbm_roof})
        else:
            roof.update(hbm_roof)
        roof.update({"traffic": traffic, "traffic_source": traffic_source,
                     "census": {"covered_pixels_per_step": covered, "table_tiles_computed": tiles, "pixel_tiles_touched": pixel_tiles,
                                "table_fraction_computed": None if tiles is None else tiles * 1024 / (n_img * math.prod(ops.table_shape(mod.mesh, 1, H * W)))},
                     "note": "bound = the resource that limits the kernel: vector-instruction issue (SQ_INSTS_VALU of the committed counter "
                             "pass of this launch shape / the launch time measured here / 256 CUs x 4 SIMDs x clock / 2 cycles per "
                             "wave-instruction); the rest of the time is barriers, one-lane sections and LDS / gather latency. `hbm` is "
                             "the same launch against the HBM roofline on the implemented design's algorithmic bytes; `traffic` the "
                             "fabric-side bytes per launch from the FETCH_SIZE + WRITE_SIZE passes"})
        hx, hy, hz = HALF_EXTENTS
        out = {
            "metric": "pose_opt_steps_per_sec", "value": n_hyp_steps / elapsed, "unit": "hypothesis-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "images_per_sec": args.steps * B * world / elapsed,
            "config": {"workload": (f"multi-hypothesis solve (BASELINE configs[2]): {STRONG_TOTAL_IMAGES} synthetic 640x800 images x "
                                    f"{STRONG_HYPOTHESES} initial poses in total, sharded over the GPUs, " if args.strong else
                                    "omni_nemo_pose_3d car (BASELINE configs[1]): batch 64 synthetic 640x800 images per GPU, ") +
                                   f"80x100x128 feature maps, cuboid mesh V={v.shape[0]} F={f.shape[0]} with half extents {hx}/{hy}/{hz} (half of a "
                                   "car's, so that it fits the 100-pixel map at distance 5), 144-pose pre-rendered grid scoring + "
                                   "300 Adam steps per hypothesis, random-init vertex bank",
                       "batch_per_gpu": B, "global_batch": B * world, "grid_poses": n_poses, "adam_steps": EPOCHS,
                       "batches_in_flight": G, "launch_streams": n_streams, "staggered_groups": stagger,
                       "similarity_table": "complete, in front of the solve" if args.complete_table else
                                           ("tiles on demand inside the solve" if args.no_bulk_tiles else
                                            "tiles on demand: the first step's (and a two-pixel ring's) by one kernel in front of the loop, the rest inside it"),
                       "clutter_score": "its own pass over the feature maps" if args.separate_clutter else "scored by the pass that builds the grid scoring's operand fragments (timed under grid_score)",
                       "hypotheses_per_image": K_HYP,
                       "partition": (f"images sharded over {world} GPU(s), all-gather of [B,7] results" if world > 1 else "single GPU") +
                                    (f" (process group: {dist.get_backend()}, world size {dist.get_world_size()})" if grouped else "")},
            "kernel_ms_per_launch": dict(kern_ms), "hypotheses_per_launch": n_hyp,
            "median_pose_error_deg_vs_synthetic_gt": med_err,
            "roofline": roof,
            "speedup_vs_streaming_design": n_hyp * EPOCHS * STREAM_BYTES_PER_HYP_STEP / (solve_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "front_kernel_rooflines": other,
        }
        if rep_ms:
            srt = sorted(rep_ms)
            out["repeats"] = {"n": len(rep_ms), "ms": [round(x, 3) for x in rep_ms], "min_ms": srt[0], "median_ms": srt[len(srt) // 2],
                              "median_value": n_hyp_steps / (srt[len(srt) // 2] * 1e-3),
                              "note": "the timed region run again after the scored one (same steps, same buffers), rank 0's clock"}
        if not args.no_other_configs and world == 1:
            out["other_configs"] = other_configs(dev, cfg, mod, clutter, maps, cs, ths, packed, batch=B,
                                                 which=tuple(x for x in args.other_configs.split(",") if x))
        if not args.no_cpu_baseline and world == 1:      # the CPU leg is timed at N = 1 only
            out["cpu_baseline"] = cpu_baseline(mod, mem, clutter, v, f, fm_g[0], pre, cs, ths)
        print(json.dumps(out))
    if grouped:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
