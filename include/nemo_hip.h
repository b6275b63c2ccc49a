/*
 * nemo_hip.h - C ABI of the MI355X (gfx950) render-and-compare pose solver.
 *
 * Every entry point is what a binding of the reference's hot path would call: plain pointers
 * and sizes, no torch types.  All data pointers are DEVICE pointers owned by the caller unless
 * a comment says "host".  `stream` is a hipStream_t passed as void* (NULL = default stream);
 * work is enqueued on it and the call returns without synchronising.  Return value: 0 on
 * success, a negative NEMO_ERR_* code otherwise; nemo_last_error() then describes the failure
 * (thread-local string).  No entry point except nemo_mesh_create / nemo_mesh_set_memory /
 * nemo_mesh_destroy allocates or frees device memory.
 *
 * Conventions (PyTorch3D 0.7.1 semantics as used by the reference, see each citation):
 *   - mesh vertices are already in render axes, i.e. after (x,y,z)->(x,z,-y)
 *     (reference nemo/utils/mesh.py:162-170);
 *   - a camera is (campos[3], theta) looking at the origin, up = +y, R = look_at(C) @ Rz(theta),
 *     T = -R^T C (nemo/utils/mesh.py:222-226), screen-space pinhole with focal length `focal`
 *     and principal point (px,py) in feature-map pixels on an H x W map (nemo/models/nemo.py:258-280);
 *   - feature maps are float32 [*, C, H, W] contiguous (H*W is written HW below).
 */
#ifndef NEMO_HIP_H
#define NEMO_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NEMO_OK               0
#define NEMO_ERR_INVALID     -1  /* bad argument (null pointer, non-positive size, ...)        */
#define NEMO_ERR_UNSUPPORTED -2  /* size outside what the kernels are built for                */
#define NEMO_ERR_HIP         -3  /* a HIP runtime call failed                                  */
#define NEMO_ERR_STATE       -4  /* e.g. mesh has no vertex memory yet                         */

typedef struct nemo_mesh_s* nemo_mesh_t;

/* Pinhole camera shared by all poses of a call (reference: PerspectiveCameras(in_ndc=False,
 * image_size=((H,W),), focal_length=3000/8, principal_point=((W//2,H//2),)) built at
 * nemo/models/nemo.py:258-280 and RasterizationSettings(blur_radius=0, faces_per_pixel=1,
 * bin_size=0) from config/models/nemo_runtime.yaml:47-54). */
typedef struct {
    int32_t H, W;      /* feature-map height, width                              */
    float   focal;     /* focal length in feature-map pixels                     */
    float   px, py;    /* principal point used when a call passes principal=NULL */
} nemo_camera;

/* Adam + ExponentialLR exactly as the reference configures them (torch.optim.Adam lr .05,
 * betas (.4,.6), eps 1e-8; ExponentialLR gamma .2 ticking every epochs//3 steps:
 * nemo/models/batch_solve_pose.py:306-338, config/omni_nemo_pose_3d.yaml:53-65). */
typedef struct {
    int32_t steps;       /* cfg.inference.epochs                                       */
    int32_t lr_every;    /* scheduler ticks after every lr_every-th step (epochs // 3) */
    double  lr, beta1, beta2, eps, lr_gamma;
} nemo_adam;

const char* nemo_last_error(void);

/* Library / build info string (host). */
const char* nemo_version(void);

/* ---- mesh handle ------------------------------------------------------------------------
 * Replaces the state of MeshInterpolateModuleMesh.__init__ / update_memory
 * (nemo/models/mesh_interpolate_module.py:92-150): the mesh and the per-vertex feature memory.
 * The handle keeps its own device copies (verts [V,3] f32, faces [F,3] i32, memory [V,C] f32
 * and its transpose [C,V]).  face_memory[f,j,:] of the reference == memory[faces[f,j],:]
 * (nemo/utils/mesh.py:173-174) and is never materialised.
 * nemo_mesh_create copies verts / faces synchronously (null-stream copies to the device and to the host, where the
 * mesh is checked for being a closed solid): the caller's buffers must be complete on the device before the call,
 * i.e. work that produces them on a non-blocking stream has to be synchronised first.  On failure nothing stays
 * allocated. */
int nemo_mesh_create(const float* verts, const int32_t* faces, int32_t V, int32_t F, nemo_mesh_t* out);
int nemo_mesh_set_memory(nemo_mesh_t mesh, const float* memory /*[V,C]*/, int32_t C, void* stream);
int nemo_mesh_destroy(nemo_mesh_t mesh);
/* A handle may hold several meshes back to back - the layout of a multi-class FeatureBankNeMo, whose memory is the
 * per-mesh vertex blocks concatenated (nemo/models/feature_banks.py:133-163, offsets as in
 * cu_layers/CuNeMo/gather_features.py:31; the reference keeps one MeshInterpolateModule per category instead,
 * nemo/models/nemo_cls.py:156-168).  Create it from the concatenated vertices and the concatenated faces (face
 * indices into the concatenated vertex list), then name the parts: part l = the next vert_counts[l] vertices and
 * the next face_counts[l] faces (HOST arrays of n_parts entries; every face must stay inside its part's vertices).
 * nemo_similarity_table then covers all meshes at once and nemo_solve_parts renders one part per hypothesis. */
int nemo_mesh_set_parts(nemo_mesh_t mesh, const int32_t* vert_counts, const int32_t* face_counts, int32_t n_parts);
int nemo_mesh_info(nemo_mesh_t mesh, int32_t* V, int32_t* F, int32_t* C);

/* ---- camera / projection (diagnostic + parity) -------------------------------------------
 * campos_to_R_T (nemo/utils/mesh.py:222-226) + MeshRasterizer.transform: world -> view -> NDC.
 * out_rt  [n,12] : R row-major (9) then T (3)                      (may be NULL)
 * out_proj[n,V,5]: view X, Y, Z, then NDC x, y of every vertex     (may be NULL) */
int nemo_project(nemo_mesh_t mesh, const nemo_camera* cam, const float* campos /*[n,3]*/,
                 const float* theta /*[n]*/, const float* principal /*[n,2] or NULL*/, int32_t n,
                 float* out_rt, float* out_proj, void* stream);

/* ---- render ------------------------------------------------------------------------------
 * MeshInterpolateModuleMesh.forward (nemo/models/mesh_interpolate_module.py:171-217) =
 * rasterize (nemo/utils/mesh.py:73-93, PyTorch3D naive rasteriser, nearest face per pixel) +
 * interpolate_face_attributes (nemo/utils/mesh.py:98-148, mode='bilinear').
 * Any output may be NULL.
 * out_maps        [n,C,H,W] f32 : interpolated vertex features, 0 on background
 * out_pix_to_face [n,H,W]   i32 : face index within the mesh, -1 = background
 *                                 (the reference's packed index is i*F + this value)
 * out_bary        [n,H,W,3] f32 : perspective-correct barycentrics, -1 on background
 * out_zbuf        [n,H,W]   f32 : interpolated view depth, -1 on background */
int nemo_render(nemo_mesh_t mesh, const nemo_camera* cam, const float* campos, const float* theta,
                const float* principal, int32_t n, float* out_maps, int32_t* out_pix_to_face,
                float* out_bary, float* out_zbuf, void* stream);

/* forward_interpolate with mode='nearest' (nemo/utils/mesh.py:59-70,104-109): the same rasterisation, but a covered
 * pixel takes the feature of the face's vertex with the largest barycentric coordinate (the first one on ties, as
 * argmax picks it) instead of the barycentric blend.  The reference forms the weight of that vertex as
 * softmax + (1 - softmax) in f32, which is 1 or 1 - 2^-24; here it is exactly 1.  Its backward
 * (nemo_render_nearest_backward) is the reference's straight-through estimator: the gradient w.r.t. the
 * barycentrics is that of softmax(barycentrics). */
int nemo_render_nearest(nemo_mesh_t mesh, const nemo_camera* cam, const float* campos, const float* theta,
                        const float* principal, int32_t n, float* out_maps, void* stream);

/* Backward of nemo_render's out_maps w.r.t. the camera, for callers that differentiate
 * MeshInterpolateModule.forward themselves (what torch autograd does through PyTorch3D in the
 * reference).  grad_maps [n,C,H,W] -> grad_campos [n,3], grad_theta [n], grad_principal [n,2]
 * (NULL = not wanted).  The rasterisation is recomputed from the pose. */
int nemo_render_backward(nemo_mesh_t mesh, const nemo_camera* cam, const float* campos,
                         const float* theta, const float* principal, int32_t n,
                         const float* grad_maps, float* grad_campos, float* grad_theta,
                         float* grad_principal, void* stream);
/* The same for maps rendered by nemo_render_nearest (mode='nearest', nemo/utils/mesh.py:59-70): the weights are
 * softmax(b) + (one_hot - softmax(b)).detach(), so the gradient reaches the barycentrics b through the softmax's
 * Jacobian, gb_j <- s_j (gb_j - sum_i gb_i s_i), and then takes the same chain to the camera. */
int nemo_render_nearest_backward(nemo_mesh_t mesh, const nemo_camera* cam, const float* campos,
                                 const float* theta, const float* principal, int32_t n,
                                 const float* grad_maps, float* grad_campos, float* grad_theta,
                                 float* grad_principal, void* stream);

/* ---- clutter score -------------------------------------------------------------------------
 * batch_solve_pose.py:214-225: 1x1 conv of the feature map with each clutter vector, max over
 * the K vectors.  feat [B,C,HW], clutter [K,C] -> out [B,HW]. */
int nemo_clutter_score(const float* feat, const float* clutter, int32_t B, int32_t C, int32_t HW,
                       int32_t K, float* out, void* stream);

/* ---- re-centring resample ---------------------------------------------------------------------
 * align_no_centered (batch_solve_pose.py:65-90, with tensor_linspace nemo/utils/general.py:11-40):
 * shifts / rescales feature maps (or clutter score maps, C = 1) so that an object seen with principal
 * point p_src at distance dist_src looks as if centred at p_tgt at distance dist_tgt; bilinear
 * grid_sample with align_corners = false; border = 1 is padding_mode='border', 0 is 'zeros'.
 * src [n,C,h1,w1]; dist_src, dist_tgt [n]; p_src, p_tgt [n,2] (x, y); out [n,C,h0,w0]. */
int nemo_align_maps(const float* src, const float* dist_src, const float* dist_tgt, const float* p_src,
                    const float* p_tgt, int32_t n, int32_t C, int32_t h1, int32_t w1, int32_t h0, int32_t w0,
                    int32_t border, float* out, void* stream);

/* ---- pose-grid scoring ---------------------------------------------------------------------
 * get_init_pos_rendered_dim0 (batch_solve_pose.py:128-164) with loss_fg_bg (:31-34):
 *   obj[b,n,p] = sum_c P[n,c,p] * feat[b,c,p]
 *   loss[b,n]  = 1 - ( mean_p max(obj, clu[b,p]) - mean_p clu[b,p] )      (clu==NULL: 1 - mean_p obj)
 *   argmin[b]  = first n attaining min_n loss[b,n]
 * pre [N,C,HW] (the reference's feature_pre_rendered squeezed), feat [B,C,HW], clu [B,HW].
 * out_loss [B,N] (may be NULL), out_argmin [B] i32, out_min [B] (may be NULL).
 * workspace: device scratch of at least nemo_grid_score_workspace(N,B,HW) bytes. */
size_t nemo_grid_score_workspace(int32_t N, int32_t B, int32_t HW);
int nemo_grid_score(const float* pre, const float* feat, const float* clu, int32_t N, int32_t B,
                    int32_t C, int32_t HW, float* out_loss, int32_t* out_argmin, float* out_min,
                    void* workspace, size_t workspace_bytes, void* stream);

/* nemo_grid_score for callers that cannot allocate device memory or hand over a workspace (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  The
 * per-tile partial sums go to a fixed array of NEMO_STATIC_SCRATCH_BYTES that the library holds on every GPU
 * it runs on until the process ends; the call allocates and frees nothing.  Sizes whose partial sums
 * (nemo_grid_score_workspace(N, B, HW) bytes) exceed the array are refused with NEMO_ERR_UNSUPPORTED.  All
 * calls on one GPU share that GPU's array: calls that overlap in time on one GPU (different streams or host
 * threads) race on it; calls that follow each other on one stream do not.
 * The device that owns `stream` must be the current device (NEMO_ERR_STATE otherwise: each GPU holds its own copy
 * of the array).  The first call on a device looks up the array's address there (the runtime may load the library's
 * code object at that moment): make one ordinary, eager call per device before any call under graph capture or
 * inside a serving loop; every later call only launches kernels. */
#define NEMO_STATIC_SCRATCH_BYTES (16u << 20)
int nemo_grid_score_static(const float* pre, const float* feat, const float* clu, int32_t N, int32_t B,
                           int32_t C, int32_t HW, float* out_loss, int32_t* out_argmin, float* out_min,
                           void* stream);

/* The same scoring on the matrix cores (v_mfma_f32_32x32x16_f16 on two-piece f16 operands: f32-grade
 * accuracy).  The pre-rendered grid is packed once, next to its rendering (batch_solve_pose.py:43-62),
 * into pixel-major fragment order; every call then converts the batch's feature maps to the same
 * order inside the workspace and runs one [N x C].[C x B] product per pixel.  C must be 16, 32, 64 or
 * 128 (nemo_grid_pack_bytes returns 0 otherwise: use nemo_grid_score).
 * packed: device buffer of nemo_grid_pack_bytes(N, C, HW) bytes, filled by nemo_grid_pack: the fragments and, behind
 * them, one bit per pixel = "some pose has a non-zero value here".  With a clutter map a pixel no pose covers is not
 * multiplied: it adds max(0, clu) to every loss, which is what the product of its all-zero fragments gives (bit-identical
 * sums).  The clutter map is expected to be the score of the same feature maps (a NaN feature at such a pixel then
 * reaches the losses through its NaN clutter score); without a clutter map every pixel is multiplied.
 * Outputs as nemo_grid_score; workspace of nemo_grid_score_packed_workspace(N, B, C, HW) bytes.
 * Input contract of the matrix-core forms (this one, nemo_grid_score_recentred and nemo_similarity_table for 16, 32,
 * 64 or 128 channels): feature values of magnitude up to 255 (the backbone's maps are unit vectors per pixel,
 * nemo/models/keypoint_representation_net.py:385-386); larger magnitudes are clamped to that range, NaN stays NaN. */
size_t nemo_grid_pack_bytes(int32_t N, int32_t C, int32_t HW);
int nemo_grid_pack(const float* pre, int32_t N, int32_t C, int32_t HW, void* packed, void* stream);
size_t nemo_grid_score_packed_workspace(int32_t N, int32_t B, int32_t C, int32_t HW);
int nemo_grid_score_packed(const void* packed, const float* feat, const float* clu, int32_t N, int32_t B,
                           int32_t C, int32_t HW, float* out_loss, int32_t* out_argmin, float* out_min,
                           void* workspace, size_t workspace_bytes, void* stream);

/* nemo_clutter_score followed by nemo_grid_score_packed with that clutter score, from ONE read of the feature maps
 * (batch_solve_pose.py:214-225 and :128-164 both read the whole batch): the pass that turns the maps into
 * matrix-core operand fragments also scores the clutter vectors.  clutter [K,C]; out_clutter_score [B,HW] is written
 * (bit for bit what nemo_clutter_score gives) and used; the other outputs and the workspace
 * (nemo_grid_score_packed_workspace bytes) as nemo_grid_score_packed.  More than 4 clutter vectors: the two steps
 * run one after the other inside the call. */
int nemo_clutter_grid_score_packed(const void* packed, const float* feat, const float* clutter, int32_t K, int32_t N, int32_t B,
                                   int32_t C, int32_t HW, float* out_clutter_score, float* out_loss, int32_t* out_argmin,
                                   float* out_min, void* workspace, size_t workspace_bytes, void* stream);

/* Scoring the same batch against several pose grids (one per category, scripts/inference.py:29-96 and
 * nemo_cls.py:262-282 score every category's grid on the same feature maps): nemo_grid_score_packed and
 * nemo_clutter_grid_score_packed leave the batch's operand fragments in the first nemo_grid_fragments_bytes(B, C, HW)
 * bytes of their workspace; nemo_grid_score_fragments scores another packed grid (same C and HW) on them without
 * reading the maps again.  Outputs as nemo_grid_score_packed, bit for bit; workspace (the partial sums):
 * nemo_grid_score_fragments_workspace(N, B, HW) bytes. */
size_t nemo_grid_fragments_bytes(int32_t B, int32_t C, int32_t HW);
size_t nemo_grid_score_fragments_workspace(int32_t N, int32_t B, int32_t HW);
int nemo_grid_score_fragments(const void* packed, const void* fragments, const float* clu, int32_t N, int32_t B, int32_t C,
                              int32_t HW, float* out_loss, int32_t* out_argmin, float* out_min, void* workspace,
                              size_t workspace_bytes, void* stream);

/* Scoring of re-centred copies of a batch, the 6-D initialisation loop of batch_solve_pose.py:264-295 (every
 * principal-point sample re-centres the whole batch with align_no_centered, :65-90, and scores it against the
 * grid): virtual image bv is source image bv % B_src resampled with (dist_src, dist_tgt, p_src, p_tgt)[bv] as
 * nemo_align_maps does, and scored like nemo_grid_score_packed - but the resampled feature maps are never written:
 * the fragment kernel samples the source maps directly.  feat [B_src,C,h1*w1], clu [B_src,h1*w1] or NULL,
 * packed = nemo_grid_pack of the [N,C,h0*w0] grid; outputs [n_virtual,N], [n_virtual], [n_virtual].  Same numbers
 * as nemo_align_maps followed by nemo_grid_score_packed, bit for bit. */
size_t nemo_grid_score_recentred_workspace(int32_t N, int32_t n_virtual, int32_t C, int32_t h0, int32_t w0);
int nemo_grid_score_recentred(const void* packed, const float* feat, const float* clu, const float* dist_src,
                              const float* dist_tgt, const float* p_src, const float* p_tgt, int32_t N, int32_t B_src,
                              int32_t n_virtual, int32_t C, int32_t h1, int32_t w1, int32_t h0, int32_t w0,
                              int32_t border, float* out_loss, int32_t* out_argmin, float* out_min, void* workspace,
                              size_t workspace_bytes, void* stream);

/* ---- vertex x pixel similarity table --------------------------------------------------------
 * table[b,v,p] = sum_c memory[v,c] * feat[b,c,p]: the reference's kpt_score_map
 * (nemo/models/solve_pose.py:120-124) and the matmul of NeMo.fast_inference
 * (nemo/models/nemo.py:350), computed once per image on the matrix cores so that the
 * optimisation loop touches three scalars per covered pixel instead of 4 x C floats.
 * feat [B,C,HW] -> table (f32) laid out [B][PT = ceil(HW/128)][VG = ceil(V/4)][128][4]: element
 *   (b,v,p) at (((b * PT + p / 128) * VG + v / 4) * 128 + p % 128) * 4 + v % 4;
 * pixels past HW and vertices past V (padding) are written as zeros.
 * nemo_similarity_table_elems returns the number of floats per image.  The matrix-core result
 * registers map onto whole 16-byte elements of this layout, each pixel tile is one contiguous
 * block, and the optimisation loop finds the values of neighbouring vertices for neighbouring
 * pixels in the same cache lines. */
size_t nemo_similarity_table_elems(nemo_mesh_t mesh, int32_t HW);
int nemo_similarity_table(nemo_mesh_t mesh, const float* feat, int32_t B, int32_t HW, float* table,
                          void* stream);

/* NeMo.fast_inference (nemo/models/nemo.py:327-360) on a table: per image, over the pixel
 * rectangle rows [r0,r1) x cols [c0,c1):  mean_p max( max_v table[b,v,p], clu[b,p] ).
 * boxes [B,4] i32 (r0,r1,c0,c1) host or NULL for the whole map; out [B]. */
int nemo_vertex_similarity(nemo_mesh_t mesh, const float* table, const float* clu, const int32_t* boxes_dev,
                           int32_t B, int32_t H, int32_t W, float* out, void* stream);

/* The same similarity straight from the feature maps (no table is written or read): the vertex x pixel products of
 * nemo_similarity_table are formed tile by tile on the matrix cores and reduced to the per-pixel maximum over the
 * vertices inside the kernel; only one partial sum per (image, 128-pixel tile) reaches memory.  Same values as
 * nemo_similarity_table + nemo_vertex_similarity up to the order of the final sum over pixels.  feat [B,C,HW],
 * clu [B,HW] (nemo_clutter_score), boxes_dev [B,4] i32 device (r0,r1,c0,c1) or NULL; workspace: device scratch of
 * nemo_vertex_similarity_workspace(B, HW) bytes.  C must be 16, 32, 64 or 128 (NEMO_ERR_UNSUPPORTED otherwise). */
size_t nemo_vertex_similarity_workspace(int32_t B, int32_t HW);
int nemo_vertex_similarity_features(nemo_mesh_t mesh, const float* feat, const float* clu, const int32_t* boxes_dev,
                                    int32_t B, int32_t H, int32_t W, float* out, void* workspace,
                                    size_t workspace_bytes, void* stream);

/* ---- pose optimisation -----------------------------------------------------------------------
 * The Adam loop of batch_solve_pose.solve_pose (batch_solve_pose.py:306-338), all `steps`
 * iterations inside one launch, one workgroup per hypothesis:
 *   render at (C,theta[,principal]) -> obj = <interpolated features, feat> per pixel ->
 *   loss_fg_bg with clutter -> backward to the pose parameters -> Adam -> LR schedule.
 * n_hyp hypotheses; hypothesis h refines against image hyp_to_image[h] (NULL: image h).
 * table [n_img] images in the layout of nemo_similarity_table, clu [n_img,HW].
 * The loss of the reference is a mean over the whole batch (global_batch images x HW pixels),
 * which sets the gradient scale seen by Adam's eps: pass the size of the batch the images
 * belong to (a shard of a larger batch passes the larger batch's size).
 * dof = 3: C and theta are optimised (4 scalars), principal fixed (batch_solve_pose.py:312-315)
 * dof = 6: C, theta and principal are optimised (:308-311).
 * score_boxes [n_hyp,4] i32 (row0,row1,col0,col1) or NULL: crop of the map on which the reported
 *                     score is computed (object_loss_fg_bg, batch_solve_pose.py:36-41,354-357).
 * out_pose [n_hyp,8]: cx, cy, cz, theta, px, py (after the last update), score (= per-image
 *                     loss_fg_bg of the LAST forward, i.e. before the last update, as
 *                     batch_solve_pose.py:345-359 computes it), loss of the first forward.
 * out_loss_hist [n_hyp,steps] per-image loss at every step (may be NULL). */
int nemo_solve(nemo_mesh_t mesh, const nemo_camera* cam, const float* table, const float* clu,
               const int32_t* hyp_to_image, const float* init_campos /*[n_hyp,3]*/,
               const float* init_theta /*[n_hyp]*/, const float* init_principal /*[n_hyp,2] or NULL*/,
               const int32_t* score_boxes, int32_t n_hyp, int32_t n_img, int32_t global_batch, int32_t dof,
               const nemo_adam* opt, float* out_pose, float* out_loss_hist, void* stream);

/* nemo_solve on a handle of several meshes (nemo_mesh_set_parts): hypothesis h refines the pose of mesh
 * hyp_to_part[h] (device, [n_hyp]) against image hyp_to_image[h]; `table` is the similarity table of the whole
 * handle.  One launch does what the reference does category by category: the per-category pose solves of
 * scripts/inference.py:29-96 with a multi-class bank, and render-and-compare classification, where every image is
 * solved against every category mesh and the smallest final loss names the class (nemo/models/nemo_cls.py:262-282).
 * A hypothesis whose part index names no part gets a NaN row. */
int nemo_solve_parts(nemo_mesh_t mesh, const nemo_camera* cam, const float* table, const float* clu,
                     const int32_t* hyp_to_image, const int32_t* hyp_to_part, const float* init_campos,
                     const float* init_theta, const float* init_principal, const int32_t* score_boxes, int32_t n_hyp,
                     int32_t n_img, int32_t global_batch, int32_t dof, const nemo_adam* opt, float* out_pose,
                     float* out_loss_hist, void* stream);

/* ---- keypoint-initialised solver (nemo/models/solve_pose.py) ----------------------------------
 * get_corr_pytorch (solve_pose.py:27-82) on the similarity table of ONE image:
 *   corr[i,a,b] = sum_v vis[i,v] * table[v, pixel(round((kp[i,v] + (px[a], py[b])) / down_sample_rate))]
 * where a keypoint counts only if the rounded pixel satisfies 0 <= x < W-1 and 0 <= y < H-1 (the
 * reference's bounds), round = half to even, and kp + sample is the f64 sum rounded to f32 (the
 * reference adds the f64 sample to the f32 tensor in place).
 * kp_coords [n_pose,V,2] f32, kp_vis [n_pose,V] f32, px/py samples f64 (device), out [n_pose,n_px,n_py]. */
int nemo_keypoint_corr(nemo_mesh_t mesh, const float* table, const float* kp_coords, const float* kp_vis,
                       const double* px_samples, const double* py_samples, int32_t n_pose, int32_t n_px,
                       int32_t n_py, float down_sample_rate, int32_t H, int32_t W, float* out, void* stream);

/* The per-proposal Adam loop of solve_pose.solve_pose (solve_pose.py:220-324): like nemo_solve, but
 * the translation is the reference's 2-D flow: the map rendered at the camera's own principal point
 * is resampled bilinearly at pixel + flow / translation_scale (flow_warp.py:6-46: grid_sample,
 * align_corners=True, border padding) before the comparison with the image.  Parameters: C, theta and,
 * when optimize_translation != 0, the flow.  init_flow [n_hyp,2] (NULL: zero).  out_pose [n_hyp,8] as
 * nemo_solve with columns 4,5 = the flow.  Every proposal of an image is one hypothesis
 * (hyp_to_image), so the reference's sequential loop over proposals becomes one launch. */
int nemo_solve_flow(nemo_mesh_t mesh, const nemo_camera* cam, const float* table, const float* clu,
                    const int32_t* hyp_to_image, const float* init_campos, const float* init_theta,
                    const float* init_flow, int32_t n_hyp, int32_t n_img, int32_t global_batch,
                    int32_t optimize_translation, float translation_scale, const nemo_adam* opt,
                    float* out_pose, float* out_loss_hist, void* stream);

/* ---- the optimisation loop straight from the feature maps ------------------------------------------------------
 * nemo_solve / nemo_solve_parts without a precomputed table (no counterpart in the reference, which recomputes the
 * whole rendered map and its product with the image at every step, batch_solve_pose.py:320-338).  The loop reads
 * table entry (vertex, pixel) only where a face of that vertex covers the pixel during some step - about one entry
 * in a hundred - so the kernel computes the table itself, one tile of 32 vertices x 32 pixels at a time, the first
 * time a covered pixel needs it, with the same matrix-core product as nemo_similarity_table (every entry it reads
 * is bit for bit what nemo_similarity_table would have written; results equal nemo_similarity_table + nemo_solve
 * bit for bit).  feat [n_img,C,HW] are the images' feature maps; workspace is device memory of
 * nemo_solve_features_workspace(mesh, n_img, HW) bytes that need not be initialised: room for the tables in the
 * layout of nemo_similarity_table (n_img * nemo_similarity_table_elems floats; only the computed tiles are written,
 * the rest stays undefined).  hyp_to_part may be NULL (whole handle).  Several hypotheses may refine against one
 * image.  C must be 16, 32, 64 or 128 (the workspace size is 0 and the call returns NEMO_ERR_UNSUPPORTED otherwise:
 * use nemo_similarity_table + nemo_solve).
 * The tiles of the first step are not computed one by one inside the loop: a launch that only looks takes every
 * hypothesis's first step as far as its covered pixels and lists their tiles (and those of the pixels two columns /
 * rows around them, which the next steps are likely to reach), one kernel computes the listed tiles, and the loop
 * starts with them present.  The lists (one bit per tile and hypothesis) live behind the tables in the workspace:
 * nemo_solve_features_workspace leaves room for two hypotheses per image; a call with more hypotheses than the
 * workspace has room for, or with a workspace that holds the tables alone, computes every tile inside the loop.
 * Results are the same bit for bit either way. */
size_t nemo_solve_features_workspace(nemo_mesh_t mesh, int32_t n_img, int32_t HW);
/* 1 when nemo_solve_features takes this mesh on an H x W map (channel count, and the presence maps of the tiles fit
 * the workgroup's LDS beside the rasteriser's buffers: meshes of a few thousand vertices per part), else 0: then use
 * nemo_similarity_table + nemo_solve.  with_parts != 0: for calls with hyp_to_part (the largest part counts). */
int nemo_solve_features_supported(nemo_mesh_t mesh, int32_t H, int32_t W, int32_t with_parts);
int nemo_solve_features(nemo_mesh_t mesh, const nemo_camera* cam, const float* feat, void* workspace, size_t workspace_bytes,
                        const float* clutter_score, const int32_t* hyp_to_image, const int32_t* hyp_to_part,
                        const float* init_campos, const float* init_theta, const float* init_principal,
                        const int32_t* score_boxes, int32_t n_hyp, int32_t n_img, int32_t global_batch, int32_t dof,
                        const nemo_adam* opt, float* out_pose, float* out_loss_hist, void* stream);

/* Diagnostic (no counterpart in the reference): the rasterise phase of the optimisation loop in two forms on the same
 * poses, timed in shader cycles inside the kernel.  form 0 = the form the solver uses (a face per lane walks the pixel
 * centres of its bounding box, LDS atomic min on depth | face keys); form 1 = faces binned to 8 x 8-pixel tiles in LDS,
 * a wave per tile, lanes = pixels, no atomics on the keys.  One workgroup per pose, `reps` rasterisations each.
 * out_cycles [2 n]: per pose the average cycles of one rasterisation, then (form 1) of its binning part;
 * out_keys [n, H*W] u32 or NULL: the key buffer after the last one (0xFFFFFFFF = not covered, else quantised depth
 * above the face index).  Form 1 keeps 4096 (face, tile) pairs per pose in LDS and drops the rest (the reference's
 * meshes need about 1300); meshes of 65536 faces and more are refused. */
int nemo_raster_forms(nemo_mesh_t mesh, const nemo_camera* cam, const float* campos, const float* theta, int32_t n,
                      int32_t form, int32_t reps, float* out_cycles, uint32_t* out_keys, void* stream);

/* camera_position_to_spherical_angle (nemo/utils/mesh.py:50-56) on device:
 * campos [n,3] -> out [n,3] = distance, elevation, azimuth. */
int nemo_spherical_angles(const float* campos, int32_t n, float* out, void* stream);

/* Training-side keypoint annotation, get_one_standard of nemo/models/project_kp.py:175-247 for one mesh per
 * pose (PackedRaster, mesh_mode 'single', type 'near', :132-143): per pose the mesh vertices projected to
 * feature-map pixels, out_kp [n,V,2] = (y, x) (clamped to the map when restrict_to_boundary != 0), and
 * out_vis [n,V] u8 = 1 where the distance map (per-vertex distance to cam_loc = -inverse(R) T, interpolated
 * over the rasterised faces) read at the nearest pixel equals the vertex's own distance within dist_thr.
 * principal [n,2] in feature-map pixels or NULL.  rt [n,12] or NULL: the caller's own R (row-major, PyTorch3D
 * convention X_view = X R + T) followed by T per pose, used instead of campos / theta (the R=, T= keyword
 * arguments of PackedRaster.__call__, :127-129); campos and theta may then be NULL.  The distance map is never
 * materialised. */
int nemo_project_keypoints(nemo_mesh_t mesh, const nemo_camera* cam, const float* campos, const float* theta,
                           const float* rt, const float* principal, int32_t n, float dist_thr,
                           int32_t restrict_to_boundary, float* out_kp, uint8_t* out_vis, void* stream);

/* ---- feature-bank bookkeeping of multi-object batches (the reference's CuNeMo extension) ----
 * An image n holds up to M objects: sample_indexs[n,i] = mesh index of object i (-1: none); its K padded
 * vertex slots are the objects' vertices back to back, object i starting at slot sample_shifts[n,i];
 * input_valid[n] = slots in use.  The bank holds the L meshes back to back, mesh l starting at
 * bank_shifts[l]; KK = bank rows.  All index arrays i32, device memory.
 *
 * nemo_gather_features replaces _C.gather_features (cu_layers/CuNeMo/csrc/gather_features/
 * gather_features.cu:24-68,101-150, bound in csrc/ext.cpp):
 *   forward_mapping[n,k] = bank row of slot (n,k), -1 for k >= input_valid[n];
 *   out_feat[v,:] = sum over the slots mapped to v of weight[n,k] * feat[n,k,:];  out_weight[v] = sum of the
 *   weights.  Every output element is written (no zero-fill needed).  The sums run in a fixed order
 *   (image, then object) instead of the reference's atomicAdd order.  Precondition as in the reference's
 *   wrapper (gather_features.py:26-30): sample_shifts are the running sums of the objects' vertex counts. */
int nemo_gather_features(const float* feat, const float* weight, const int32_t* input_valid,
                         const int32_t* bank_shifts, const int32_t* sample_shifts, const int32_t* sample_indexs,
                         int32_t N, int32_t M, int32_t K, int32_t C, int32_t L, int32_t KK, float* out_feat,
                         float* out_weight, int32_t* forward_mapping, void* stream);
/* _C.gather_features_backward (gather_features.cu:71-98,153-190): per slot
 *   gi_weight[n,k] = g_weight[v] + <g_feat[v,:], feat[n,k,:]>,  gi_feat[n,k,:] = weight[n,k] * g_feat[v,:],
 * zeros for padding slots; every output element is written. */
int nemo_gather_features_backward(const float* g_feat, const float* g_weight, const float* feat,
                                  const float* weight, const int32_t* forward_mapping, int32_t N, int32_t K,
                                  int32_t C, float* gi_feat, float* gi_weight, void* stream);
/* _C.gather_idx (csrc/mask_weight/mask_weight.cu:10-56,93-135): per slot the object's mesh index, the
 * vertex index inside that mesh, the mesh's first bank row (bank_starts[mesh]) and its vertex count
 * (sample_n_verts[n,i]); zeros for padding slots.  Outputs [N,K] i32. */
int nemo_gather_idx(const int32_t* input_valid, const int32_t* sample_shifts, const int32_t* sample_indexs,
                    const int32_t* sample_n_verts, const int32_t* bank_starts, int32_t N, int32_t M, int32_t K,
                    int32_t* object_idx, int32_t* source_vert_idx, int32_t* verts_start, int32_t* num_verts,
                    void* stream);
/* _C.mask_weight (mask_weight.cu:59-90,139-177): out[s,k] for selected slot s and bank column k (K columns)
 *   = verts_dist_weight[object_idx[s], source_vert_idx[s], k - verts_start[s]]  for verts_start[s] <= k < verts_end[s],
 *   = weight_c for k >= clutter_start, = weight_x otherwise.  verts_dist_weight [n_obj,V,V] f32. */
int nemo_mask_weight(const int32_t* object_idx, const int32_t* source_vert_idx, const int32_t* verts_start,
                     const int32_t* verts_end, const float* verts_dist_weight, float weight_x, float weight_c,
                     int32_t clutter_start, int32_t S, int32_t K, int32_t V, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NEMO_HIP_H */
