"""Several category meshes against one multi-class feature bank, solved in ONE launch.

What the reference does category by category:
  * pose with a known category and a multi-class FeatureBankNeMo (its memory = the per-mesh vertex blocks back to
    back, nemo/models/feature_banks.py:133-163 with mesh_n_list; scripts/inference.py:29-96 loops over the categories,
    one model and one batch of batch_solve_pose.solve_pose per category);
  * 3D-aware classification: every image is solved against every category mesh and the smallest final loss names the
    class (nemo/models/nemo_cls.py:262-282, scripts/classification_accuracy.py:86-87).

Here the category meshes live back to back in one mesh handle (ops.MeshHandle.set_parts) whose vertex memory is the
bank's memory_pos as it is.  One clutter launch, one grid scoring per category (each category has its own
pre-rendered grid) and ONE solve launch over all (image, category) pairs (which computes the similarity-table tiles
its covered pixels need itself), every pair a hypothesis with its own mesh (ops.solve(..., hyp_to_part=...)).  Every pair's result is what
batch_solve_pose.solve_pose gives for that image on that category's own module with the same `global_batch`
(bit for bit: a hypothesis only reads its own mesh, its own image and the loss normaliser).
"""
import time

import numpy as np
import torch

from . import ops
from .batch_solve_pose import (_adam_from_cfg, _clutter_rows, get_pre_render_samples, initial_poses, pack_predictions,
                               score_boxes_from_bbox)
from .mesh_interpolate_module import MeshInterpolateModuleMesh
from .utils import get_param_samples


def _shape_only(maps):
    """A tensor of maps' shape without its storage (one element, stride 0) that carries maps' packed grid: what the solver
    passes where the reference passes feature_pre_rendered once the f32 maps themselves are not needed any more."""
    stand_in = torch.zeros(1, device=maps.device, dtype=maps.dtype).expand(maps.shape)
    stand_in._nemo_packed_grid = maps._nemo_packed_grid
    stand_in._nemo_shape_only = True
    return stand_in


class MultiCategorySolver:
    """categories: names; meshes: [(vertices [V_i,3], faces [F_i,3])] in the bank's order; memory_bank: a
    FeatureBankNeMo built with mesh_n_list (or anything with .memory_pos / .vertex_block(i)); clutter_bank [K, C];
    rasterizer: the shared MeshRasterizer (camera + raster settings); dof 3 or 6 (`principal_samples` [P, 2] in
    feature-map pixels for dof 6); keep_f32_grids: keep the pre-rendered f32 maps beside their packed form (`grids[i][0]`
    is otherwise a storage-free stand-in that only carries the packed grid)."""

    def __init__(self, cfg, categories, meshes, memory_bank, clutter_bank, rasterizer, device="cuda:0", dof=3,
                 principal_samples=None, keep_f32_grids=False):
        if len(categories) != len(meshes) or not categories:
            raise ValueError("one mesh per category")
        self.cfg, self.categories, self.dof = cfg, list(categories), int(dof)
        self.device = torch.device(device)
        self.rasterizer = rasterizer
        self.clutter_bank = clutter_bank.to(self.device)
        counts = [int(np.asarray(v).shape[0]) for v, _ in meshes]
        if int(memory_bank.memory_pos.shape[0]) != sum(counts):
            raise ValueError(f"the bank holds {int(memory_bank.memory_pos.shape[0])} vertex rows, the meshes {sum(counts)} vertices")
        # one module per category (its own handle): pre-rendering, and the per-category reference path
        self.modules = [MeshInterpolateModuleMesh(torch.as_tensor(v), torch.as_tensor(f), memory_bank.vertex_block(i).detach().cpu(),
                                                  rasterizer).to(self.device) for i, (v, f) in enumerate(meshes)]
        # all categories back to back in one handle: vertices as the modules hold them (render axes), faces shifted
        offs = np.concatenate([[0], np.cumsum(counts)[:-1]])
        verts = torch.cat([m.verts for m in self.modules]).to(self.device)
        faces = torch.cat([m.faces.to(torch.int32) + int(o) for m, o in zip(self.modules, offs)]).to(self.device)
        self.mesh = ops.MeshHandle(verts, faces)
        self.mesh.set_memory(memory_bank.memory_pos.detach().float().to(self.device))
        self.mesh.set_parts(counts, [int(m.faces.shape[0]) for m in self.modules])
        az, el, th, di, _, _ = get_param_samples(cfg)
        self.grids = [get_pre_render_samples(m, az, el, th, di) for m in self.modules]
        if not keep_f32_grids:
            # the scoring reads the packed (matrix-core) form only: drop the f32 maps (590 MB per category at 80 x 100 x 128) and
            # keep a storage-free stand-in of the same shape that carries the packed grid
            self.grids = [(_shape_only(maps), cs, ths) if getattr(maps, "_nemo_packed_grid", None) is not None else (maps, cs, ths)
                          for maps, cs, ths in self.grids]
        self.principal_samples = None if principal_samples is None else torch.as_tensor(principal_samples).float().to(self.device)
        if self.dof == 6 and self.principal_samples is None:
            raise ValueError("dof 6 needs principal_samples")

    def camera(self):
        return self.modules[0].camera()

    def solve_pairs(self, feature_map, pair_image, pair_category, global_batch=None, bbox=None, down_sample_rate=8):
        """Refines the pose of category pair_category[j] in image pair_image[j] for every pair j, all pairs in one solve
        launch.  -> (list of prediction dicts in pair order (as batch_solve_pose.solve_pose returns them), device
        tensor [n_pairs, 8] of the raw solve rows)."""
        cfg = self.cfg
        dev = self.device
        fm = feature_map.to(dev)
        b, c = fm.shape[0], fm.shape[1]
        pair_image = torch.as_tensor(pair_image, dtype=torch.long)
        pair_category = torch.as_tensor(pair_category, dtype=torch.long)
        n_pairs = int(pair_image.numel())
        adam = _adam_from_cfg(cfg)
        epochs = int(cfg.inference.epochs)
        times = {}
        t0 = time.time()
        cats = [cat for cat in sorted(set(pair_category.tolist())) if 0 <= cat < len(self.grids)]   # others: the solve answers with a NaN row
        rows_of = {cat: torch.nonzero(pair_category == cat).reshape(-1) for cat in cats}
        whole = {cat: rows_of[cat].numel() == b and bool(torch.equal(pair_image[rows_of[cat]], torch.arange(b))) for cat in cats}
        packs = {cat: getattr(self.grids[cat][0], "_nemo_packed_grid", None) for cat in cats}
        # Categories that see every image, in order, share one conversion of the maps into the scoring's operand
        # fragments (3-D: the batch is scored as it is): the first of them scores the clutter vectors in the same pass.
        shared = [cat for cat in cats if whole[cat] and packs[cat] is not None and self.dof != 6]
        frag_ws, grid_scores = None, {}
        if shared:
            first = shared[0]
            frag_ws = torch.empty(ops.grid_score_workspace_bytes(packs[first], b), dtype=torch.uint8, device=dev)
            clutter_score, l_, i_, m_ = ops.clutter_and_grid_score(fm, _clutter_rows(self.clutter_bank, c), packs[first], workspace=frag_ws)
            grid_scores[first] = (l_, i_, m_)
        else:
            clutter_score = ops.clutter_score(fm, _clutter_rows(self.clutter_bank, c))
        t1 = time.time()
        times["pre_compute_time"] = t1 - t0
        # initial poses: one grid scoring per category, over the images paired with it
        init_C = torch.zeros(n_pairs, 3, device=dev)
        init_theta = torch.zeros(n_pairs, device=dev)
        init_principal = torch.zeros(n_pairs, 2, device=dev) if self.dof == 6 else None
        cams = self.rasterizer.cameras
        for cat in cats:
            rows = rows_of[cat]
            imgs = pair_image[rows]
            maps, cs, ths = self.grids[cat]
            if whole[cat]:                       # every image, in order: no copies
                fm_c, clu_c = fm, clutter_score
                if cat in shared and cat not in grid_scores:
                    grid_scores[cat] = ops.grid_score_on_fragments(packs[cat], frag_ws, b, clutter_score)
            else:
                fm_c, clu_c = fm[imgs.to(dev)].contiguous(), clutter_score[imgs.to(dev)].contiguous()
            C_, th_, pr_ = initial_poses(fm_c, clu_c, cams, cs, ths, maps,
                                         principal=self.principal_samples if self.dof == 6 else None, dof=self.dof,
                                         grid_scores=grid_scores.get(cat))
            init_C[rows.to(dev)], init_theta[rows.to(dev)] = C_, th_
            if init_principal is not None:
                init_principal[rows.to(dev)] = pr_
        cams._N = b
        t2 = time.time()
        times["pre_rendering_time"] = t2 - t1
        boxes = None
        if bbox is not None:
            boxes = score_boxes_from_bbox(bbox, b, 1, down_sample_rate)[pair_image]
        skw = dict(hyp_to_image=pair_image.to(torch.int32), hyp_to_part=pair_category.to(torch.int32),
                   global_batch=b if global_batch is None else int(global_batch), dof=6 if self.dof == 6 else 3,
                   steps=epochs, lr_every=epochs // 3 if epochs >= 3 else 0, score_boxes=boxes, **adam)
        if ops.on_demand_table_supported(self.mesh, self.camera(), with_parts=True):
            # only the table tiles a pair's covered pixels need are computed (inside the solve): an image's table over
            # all categories' vertices is never built
            pose, _ = ops.solve_from_features(self.mesh, self.camera(), fm, clutter_score, init_C, init_theta, init_principal, **skw)
        else:
            table = ops.similarity_table(self.mesh, fm)
            pose, _ = ops.solve(self.mesh, self.camera(), table, clutter_score, init_C, init_theta, init_principal, **skw)
        sph = ops.spherical_angles(pose[:, 0:3])
        host = torch.cat([pose, sph], dim=1).cpu().numpy()
        times["optimization_time"] = time.time() - t2
        return pack_predictions(host, n_pairs, 1, times), pose

    def solve_labelled(self, feature_map, labels, **kw):
        """Pose of every image with the mesh of its own category (labels [B] = category index)."""
        labels = torch.as_tensor(labels, dtype=torch.long).reshape(-1)
        preds, _ = self.solve_pairs(feature_map, torch.arange(labels.numel()), labels, **kw)
        return preds

    max_workspace_bytes = 64 << 30      # classify() splits the batch so that one launch's similarity-tile workspace stays below this

    def classify(self, feature_map, **kw):
        """Render-and-compare classification: every image against every category.
        -> (category name per image, final losses [B, n_categories] (numpy), predictions [B][n_categories]).
        The batch is taken in slices of as many images as fit the workspace budget (the tile workspace holds room for
        every category's vertices per image); every slice is solved with the loss normaliser of the whole batch, so the
        slicing does not change any result."""
        b, n_cat = feature_map.shape[0], len(self.categories)
        hw = int(feature_map.shape[2] * feature_map.shape[3])
        per_image = ops.solve_workspace_bytes(self.mesh, 1, hw) or 4 * int(np.prod(ops.table_shape(self.mesh, 1, hw)))
        step = max(1, min(b, int(self.max_workspace_bytes // max(per_image, 1))))
        step = -(-b // -(-b // step))            # slices of equal size
        kw.setdefault("global_batch", b)
        preds, scores = [], []
        for i0 in range(0, b, step):
            n = min(step, b - i0)
            pair_image = torch.arange(n).repeat_interleave(n_cat)
            pair_category = torch.arange(n_cat).repeat(n)
            bbox = kw.get("bbox")
            kw_i = dict(kw, bbox=None if bbox is None else bbox[i0:i0 + n])
            p_, pose = self.solve_pairs(feature_map[i0:i0 + n], pair_image, pair_category, **kw_i)
            preds += p_
            scores.append(pose[:, 6].reshape(n, n_cat).cpu().numpy())
        scores = np.concatenate(scores)
        names = [self.categories[j] for j in scores.argmin(axis=1)]
        return names, scores, [preds[i * n_cat:(i + 1) * n_cat] for i in range(b)]
