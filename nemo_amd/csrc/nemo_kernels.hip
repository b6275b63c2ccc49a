// HIP kernels (gfx950) and C ABI of the render-and-compare pose solver.  See include/nemo_hip.h
// for the contract of every entry point and the reference lines each one replaces.
//
// Kernel inventory
//   solve_kernel            one workgroup per pose hypothesis, all Adam steps inside the launch:
//                           pose -> project vertices (LDS) -> rasterise faces into an LDS depth/face
//                           key buffer (64-bit atomic min) -> per-pixel score + loss + reverse chain
//                           -> deterministic wave/block reduction -> Adam.  No inter-workgroup traffic.
//   render_kernel           same front end, writes fragments and the interpolated feature maps.
//   render_backward_kernel  same front end, consumes a feature-map gradient.
//   clutter_kernel          1x1 conv with the clutter vectors, max over vectors.
//   grid_score_kernel       dense pose-grid scoring (partials per pixel tile) + grid_finalize_kernel.
//   table_gemm_kernel       vertex x pixel similarity table on the f32 matrix cores (in table_gemm.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include <stdlib.h>
#include <new>
#include <map>
#include <mutex>
#include <array>
#include <vector>
#include <utility>

#include "../../include/nemo_hip.h"
#include "raster.h"
#include "resample.h"

#pragma clang fp contract(off)

using namespace nemo;

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(NEMO_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                  \
    } while (0)

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

extern "C" const char* nemo_last_error(void) { return g_err; }
extern "C" const char* nemo_version(void) { return "nemo_hip 0.1 (gfx950, f32)"; }

// ------------------------------------------------------------------------------------------
// mesh handle
// ------------------------------------------------------------------------------------------
struct nemo_mesh_s {
    int V, F, C;
    float* verts;      // [V,3]
    int32_t* faces;    // [F,3]
    float* memory;     // [V,C]
    float* memory_t;   // [C,V]
    void* bank_frag;   // vertex bank as f16-piece MFMA fragments (table_gemm.hip), or null when C does not qualify
    int32_t* faces_outward;   // [F,3]: the faces wound counter-clockwise seen from outside, or null (see orient_faces)
    bool whole_outward;       // every face of faces_outward is wound outward (false when only some parts of the handle are closed solids)
    float aabb[6];            // xmin, ymin, zmin, xmax, ymax, zmax of the vertices
    int device;
    // several meshes back to back in one handle (a multi-class bank): part l = vertices [v0, v0 + nv), faces [f0, f0 + nf)
    std::vector<float> host_verts;      // copies kept for nemo_mesh_set_parts
    std::vector<int32_t> host_faces;
    std::vector<nemo::MeshPart> parts;
    nemo::MeshPart* d_parts;            // device copy of `parts`, or null
};

// Watertight-and-orientable test of a triangle mesh and outward re-winding of its faces (host, once per
// mesh).  Vertices are welded by exact position first (the reference's cuboids duplicate the vertices
// along the edges of their six sides, data/create_cuboid_mesh.py:22-97).  Succeeds only when every edge of
// the welded mesh is shared by exactly two faces, the windings can be made consistent across every edge
// and every connected component encloses a non-zero volume; then out[f] is face f with two indices
// swapped where needed so that (v1-v0) x (v2-v0) points out of the solid.  The optimisation kernel uses
// it to skip faces that look away from a camera outside the solid: such a face can never be the nearest
// one at a pixel centre that a front face covers.
static bool orient_faces(const std::vector<float>& v, const std::vector<int32_t>& f, int V, int F, std::vector<int32_t>* out) {
    std::map<std::array<uint32_t, 3>, int> weld;
    std::vector<int> canon(V);
    for (int i = 0; i < V; ++i) {
        std::array<uint32_t, 3> key;
        for (int k = 0; k < 3; ++k) { float x = v[3 * i + k] + 0.0f; memcpy(&key[k], &x, 4); }       // +0.0f: -0 and +0 weld
        canon[i] = weld.emplace(key, i).first->second;
    }
    struct Use { int face; bool forward; };
    std::map<std::pair<int, int>, std::vector<Use>> edges;
    for (int t = 0; t < F; ++t) {
        int c[3];
        for (int k = 0; k < 3; ++k) {
            const int idx = f[3 * t + k];
            if (idx < 0 || idx >= V) return false;
            c[k] = canon[idx];
        }
        if (c[0] == c[1] || c[1] == c[2] || c[0] == c[2]) return false;
        for (int k = 0; k < 3; ++k) {
            const int a = c[k], b = c[(k + 1) % 3];
            edges[{a < b ? a : b, a < b ? b : a}].push_back({t, a < b});
        }
    }
    std::vector<std::vector<std::pair<int, bool>>> adj(F);      // neighbour, "same direction on the shared edge"
    for (auto& e : edges) {
        if (e.second.size() != 2) return false;
        const Use &p = e.second[0], &q = e.second[1];
        adj[p.face].push_back({q.face, p.forward == q.forward});
        adj[q.face].push_back({p.face, p.forward == q.forward});
    }
    std::vector<int> flip(F, -1), comp(F, -1);
    int ncomp = 0;
    for (int s0 = 0; s0 < F; ++s0) {
        if (flip[s0] >= 0) continue;
        std::vector<int> stack{s0};
        flip[s0] = 0; comp[s0] = ncomp;
        while (!stack.empty()) {
            const int t = stack.back(); stack.pop_back();
            for (auto& nb : adj[t]) {
                // consistent windings run along a shared edge in opposite directions
                const int want = nb.second ? 1 - flip[t] : flip[t];
                if (flip[nb.first] < 0) { flip[nb.first] = want; comp[nb.first] = ncomp; stack.push_back(nb.first); }
                else if (flip[nb.first] != want) return false;
            }
        }
        ++ncomp;
    }
    std::vector<double> vol(ncomp, 0.0), scale(ncomp, 0.0);
    for (int t = 0; t < F; ++t) {
        const int i0 = f[3 * t], i1 = f[3 * t + (flip[t] ? 2 : 1)], i2 = f[3 * t + (flip[t] ? 1 : 2)];
        const double a[3] = {v[3 * i0], v[3 * i0 + 1], v[3 * i0 + 2]}, b[3] = {v[3 * i1], v[3 * i1 + 1], v[3 * i1 + 2]},
                     c[3] = {v[3 * i2], v[3 * i2 + 1], v[3 * i2 + 2]};
        const double d = a[0] * (b[1] * c[2] - b[2] * c[1]) - a[1] * (b[0] * c[2] - b[2] * c[0]) + a[2] * (b[0] * c[1] - b[1] * c[0]);
        vol[comp[t]] += d;
        scale[comp[t]] += d < 0 ? -d : d;
    }
    for (int k = 0; k < ncomp; ++k)
        if (!(scale[k] > 0.0) || !((vol[k] < 0 ? -vol[k] : vol[k]) > 1e-9 * scale[k])) return false;
    out->assign(f.begin(), f.end());
    for (int t = 0; t < F; ++t)
        if ((flip[t] != 0) != (vol[comp[t]] < 0)) std::swap((*out)[3 * t + 1], (*out)[3 * t + 2]);
    return true;
}

// Layout of the similarity table: [image][pixel tile = ceil(HW / 128)][vertex group = ceil(V / 4)][128][4]:
// element (v, p) of an image at ((p / 128 * VG + v / 4) * 128 + p % 128) * 4 + v % 4.  The producer's
// accumulator registers map onto whole 16-byte elements and each pixel tile is one contiguous block;
// the values of a vertex and of its neighbours in the vertex order, for neighbouring pixels, share
// cache lines.
constexpr int kTabTile = 128, kTabShift = 7;
__host__ __device__ inline size_t table_image_elems(int V, int HW) {
    return (size_t)((HW + kTabTile - 1) >> kTabShift) * ((V + 3) >> 2) * kTabTile * 4;
}
__device__ inline int table_offset(int vg_n, int v, int p) {          // within one image; fits 32 bits for maps <= 128 x 128
    return ((((p >> kTabShift) * vg_n + (v >> 2)) << kTabShift) + (p & (kTabTile - 1))) * 4 + (v & 3);
}

namespace nemo {
void launch_slot_map(const int32_t* input_valid, const int32_t* bank_shifts, const int32_t* sample_shifts, const int32_t* sample_indexs,
                     int N, int M, int K, int32_t* forward_mapping, hipStream_t st);
void launch_gather_features(const float* feat, const float* weight, const int32_t* bank_shifts, const int32_t* sample_shifts,
                            const int32_t* sample_indexs, const int32_t* forward_mapping, int N, int M, int K, int C, int L, int KK,
                            float* out_feat, float* out_weight, hipStream_t st);
void launch_gather_features_backward(const float* g_feat, const float* g_weight, const float* feat, const float* weight,
                                     const int32_t* forward_mapping, int NK, int C, float* gi_feat, float* gi_weight, hipStream_t st);
void launch_gather_idx(const int32_t* input_valid, const int32_t* sample_shifts, const int32_t* sample_indexs,
                       const int32_t* sample_n_verts, const int32_t* bank_starts, int N, int M, int K, int32_t* object_idx,
                       int32_t* source_vert_idx, int32_t* verts_start, int32_t* num_verts, hipStream_t st);
void launch_mask_weight(const int32_t* object_idx, const int32_t* source_vert_idx, const int32_t* verts_start, const int32_t* verts_end,
                        const float* verts_dist_weight, float weight_x, float weight_c, int clutter_start, int S, int K, int V,
                        float* out, hipStream_t st);
bool split_table_supported(int C);
size_t split_bank_bytes(int V, int C);
void launch_split_bank(const float* memory, void* frag, int V, int C, hipStream_t st);
int launch_grid_mfma_recentred(const void* packed, const float* feat_src, const float* clu, int N, int B, int C, int Bsrc, int h1, int w1,
                               int h0, int w0, int border, const float* dist_src, const float* dist_tgt, const float* p_src,
                               const float* p_tgt, void* featp, float* partial, hipStream_t st);
int launch_split_table(const void* frag, const float* feat, float* table, int V, int C, int HW, int B, hipStream_t st);
int launch_vertex_max(const void* frag, const float* feat, const float* clu, const int32_t* boxes, int V, int C, int H, int Wd, int B,
                      float* partial, float* out, hipStream_t st);
int split_bank_blocks(int V);
}

__global__ void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
    __shared__ float tile[32][33];
    int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    int tx = threadIdx.x, ty = threadIdx.y;
    for (int j = ty; j < 32; j += 8) {
        int r = by + j, c = bx + tx;
        if (r < rows && c < cols) tile[j][tx] = in[(size_t)r * cols + c];
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        int c = bx + j, r = by + tx;
        if (r < rows && c < cols) out[(size_t)c * rows + r] = tile[tx][j];
    }
}

// Depth-key quantisation range of a set of vertices with bounding box `aabb` (see depth_key32): the bounding
// radius about the mesh origin is the farthest corner of the box.
static void depth_range(const float* aabb, int face_bits, float* margin, float* scale, float* qmax) {
    float r2 = 0.0f;
    for (int c = 0; c < 8; ++c) {
        float d2 = 0.0f;
        for (int k = 0; k < 3; ++k) { const float x = aabb[(c >> k & 1) ? 3 + k : k]; d2 += x * x; }
        r2 = d2 > r2 ? d2 : r2;
    }
    *margin = 1.02f * sqrtf(r2) + 1e-6f;
    *qmax = (float)((1u << (32 - face_bits)) - 2u);
    *scale = *qmax / (2.0f * *margin);
}

static void box_of(const std::vector<float>& v, int v0, int nv, float* aabb) {
    for (int k = 0; k < 3; ++k) { aabb[k] = __builtin_inff(); aabb[3 + k] = -__builtin_inff(); }
    for (int i = v0; i < v0 + nv; ++i)
        for (int k = 0; k < 3; ++k) {
            aabb[k] = v[3 * i + k] < aabb[k] ? v[3 * i + k] : aabb[k];
            aabb[3 + k] = v[3 * i + k] > aabb[3 + k] ? v[3 * i + k] : aabb[3 + k];
        }
}

static int mesh_create_impl(nemo_mesh_s* m, const float* verts, const int32_t* faces, int32_t V, int32_t F) {
    HIP_TRY(hipGetDevice(&m->device));
    HIP_TRY(hipMalloc((void**)&m->verts, sizeof(float) * 3 * V));
    HIP_TRY(hipMalloc((void**)&m->faces, sizeof(int32_t) * 3 * F));
    // synchronous copies on the null stream: the caller's buffers must be complete on the device before the call
    HIP_TRY(hipMemcpy(m->verts, verts, sizeof(float) * 3 * V, hipMemcpyDeviceToDevice));
    HIP_TRY(hipMemcpy(m->faces, faces, sizeof(int32_t) * 3 * F, hipMemcpyDeviceToDevice));
    m->host_verts.resize((size_t)3 * V);
    m->host_faces.resize((size_t)3 * F);
    HIP_TRY(hipMemcpy(m->host_verts.data(), verts, sizeof(float) * 3 * V, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(m->host_faces.data(), faces, sizeof(int32_t) * 3 * F, hipMemcpyDeviceToHost));
    box_of(m->host_verts, 0, V, m->aabb);
    std::vector<int32_t> oriented;
    if (orient_faces(m->host_verts, m->host_faces, V, F, &oriented)) {
        HIP_TRY(hipMalloc((void**)&m->faces_outward, sizeof(int32_t) * 3 * F));
        HIP_TRY(hipMemcpy(m->faces_outward, oriented.data(), sizeof(int32_t) * 3 * F, hipMemcpyHostToDevice));
        m->whole_outward = true;
    }
    return NEMO_OK;
}

extern "C" int nemo_mesh_create(const float* verts, const int32_t* faces, int32_t V, int32_t F, nemo_mesh_t* out) {
    if (!verts || !faces || !out || V <= 0 || F <= 0) return fail(NEMO_ERR_INVALID, "nemo_mesh_create: bad argument");
    nemo_mesh_s* m = new (std::nothrow) nemo_mesh_s();
    if (!m) return fail(NEMO_ERR_INVALID, "out of host memory");
    m->V = V; m->F = F; m->C = 0; m->memory = nullptr; m->memory_t = nullptr; m->bank_frag = nullptr; m->faces_outward = nullptr;
    m->verts = nullptr; m->faces = nullptr; m->d_parts = nullptr; m->whole_outward = false;
    int rc = NEMO_ERR_INVALID;
    try { rc = mesh_create_impl(m, verts, faces, V, F); } catch (const std::bad_alloc&) { rc = fail(NEMO_ERR_INVALID, "out of host memory"); }
    if (rc) { (void)nemo_mesh_destroy(m); return rc; }       // frees whatever was allocated before the failure
    *out = m;
    return NEMO_OK;
}

// Several meshes in one handle: part l = the next vert_counts[l] vertices and the next face_counts[l] faces (host
// arrays).  Every face of a part must index vertices of that part.
static int mesh_set_parts_impl(nemo_mesh_t m, const int32_t* vert_counts, const int32_t* face_counts, int32_t n_parts) {
    std::vector<nemo::MeshPart> parts((size_t)n_parts);
    int v0 = 0, f0 = 0;
    for (int l = 0; l < n_parts; ++l) {
        if (vert_counts[l] <= 0 || face_counts[l] <= 0) return fail(NEMO_ERR_INVALID, "nemo_mesh_set_parts: part %d is empty", l);
        parts[l].v0 = v0; parts[l].nv = vert_counts[l]; parts[l].f0 = f0; parts[l].nf = face_counts[l];
        v0 += vert_counts[l]; f0 += face_counts[l];
        if (v0 > m->V || f0 > m->F) return fail(NEMO_ERR_INVALID, "nemo_mesh_set_parts: the parts exceed the mesh");
    }
    if (v0 != m->V || f0 != m->F)
        return fail(NEMO_ERR_INVALID, "nemo_mesh_set_parts: the parts hold %d vertices / %d faces, the mesh %d / %d", v0, f0, m->V, m->F);
    // Every part is treated as the mesh it is in a handle of its own: its own depth-key format (face bits from its own
    // face count) and its own closed-solid test (vertices are welded within the part only: two category meshes may share
    // a corner position without sharing an edge).
    std::vector<int32_t> outward(m->host_faces);
    bool any_outward = false, all_outward = true;
    for (int l = 0; l < n_parts; ++l) {
        nemo::MeshPart& q = parts[l];
        for (int t = 3 * q.f0; t < 3 * (q.f0 + q.nf); ++t)
            if (m->host_faces[t] < q.v0 || m->host_faces[t] >= q.v0 + q.nv)
                return fail(NEMO_ERR_INVALID, "nemo_mesh_set_parts: face %d of part %d uses vertex %d of another part", t / 3 - q.f0, l,
                            m->host_faces[t]);
        box_of(m->host_verts, q.v0, q.nv, q.aabb);
        int fb = 9;
        while ((1 << fb) <= q.nf) ++fb;
        q.face_bits = fb;
        depth_range(q.aabb, fb, &q.depth_margin, &q.depth_scale, &q.depth_qmax);
        std::vector<float> pv(m->host_verts.begin() + 3 * (size_t)q.v0, m->host_verts.begin() + 3 * (size_t)(q.v0 + q.nv));
        std::vector<int32_t> pf((size_t)3 * q.nf), po;
        for (int t = 0; t < 3 * q.nf; ++t) pf[t] = m->host_faces[3 * (size_t)q.f0 + t] - q.v0;
        q.outward = orient_faces(pv, pf, q.nv, q.nf, &po) ? 1 : 0;
        all_outward = all_outward && q.outward;
        if (q.outward) {
            any_outward = true;
            for (int t = 0; t < 3 * q.nf; ++t) outward[3 * (size_t)q.f0 + t] = po[t] + q.v0;
        }
    }
    HIP_TRY(hipSetDevice(m->device));
    if (any_outward) {
        if (!m->faces_outward) HIP_TRY(hipMalloc((void**)&m->faces_outward, sizeof(int32_t) * 3 * (size_t)m->F));
        HIP_TRY(hipMemcpy(m->faces_outward, outward.data(), sizeof(int32_t) * 3 * (size_t)m->F, hipMemcpyHostToDevice));
    } else if (m->faces_outward) {
        (void)hipFree(m->faces_outward);
        m->faces_outward = nullptr;
    }
    nemo::MeshPart* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, sizeof(nemo::MeshPart) * n_parts));
    if (hipMemcpy(d, parts.data(), sizeof(nemo::MeshPart) * n_parts, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(d);
        return fail(NEMO_ERR_HIP, "nemo_mesh_set_parts: copying the part table failed");
    }
    if (m->d_parts) (void)hipFree(m->d_parts);
    m->d_parts = d;
    m->parts = parts;
    m->whole_outward = any_outward && all_outward;
    return NEMO_OK;
}

extern "C" int nemo_mesh_set_parts(nemo_mesh_t m, const int32_t* vert_counts, const int32_t* face_counts, int32_t n_parts) {
    if (!m || !vert_counts || !face_counts || n_parts <= 0) return fail(NEMO_ERR_INVALID, "nemo_mesh_set_parts: bad argument");
    try { return mesh_set_parts_impl(m, vert_counts, face_counts, n_parts); }
    catch (const std::bad_alloc&) { return fail(NEMO_ERR_INVALID, "out of host memory"); }
}

extern "C" int nemo_mesh_set_memory(nemo_mesh_t m, const float* memory, int32_t C, void* stream) {
    if (!m || !memory || C <= 0) return fail(NEMO_ERR_INVALID, "nemo_mesh_set_memory: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (m->C != C) {
        if (m->memory) { HIP_TRY(hipFree(m->memory)); m->memory = nullptr; }
        if (m->memory_t) { HIP_TRY(hipFree(m->memory_t)); m->memory_t = nullptr; }
        HIP_TRY(hipMalloc((void**)&m->memory, sizeof(float) * (size_t)m->V * C));
        HIP_TRY(hipMalloc((void**)&m->memory_t, sizeof(float) * (size_t)m->V * C));
        if (m->bank_frag) { HIP_TRY(hipFree(m->bank_frag)); m->bank_frag = nullptr; }
        if (nemo::split_table_supported(C)) HIP_TRY(hipMalloc(&m->bank_frag, nemo::split_bank_bytes(m->V, C)));
        m->C = C;
    }
    HIP_TRY(hipMemcpyAsync(m->memory, memory, sizeof(float) * (size_t)m->V * C, hipMemcpyDeviceToDevice, st));
    dim3 grid((C + 31) / 32, (m->V + 31) / 32), block(32, 8);
    hipLaunchKernelGGL(transpose_kernel, grid, block, 0, st, m->memory, m->memory_t, m->V, C);
    if (m->bank_frag) nemo::launch_split_bank(m->memory, m->bank_frag, m->V, C, st);
    HIP_TRY(hipGetLastError());
    return NEMO_OK;
}

extern "C" int nemo_mesh_destroy(nemo_mesh_t m) {
    if (!m) return NEMO_OK;
    if (m->verts) (void)hipFree(m->verts);
    if (m->faces) (void)hipFree(m->faces);
    if (m->memory) (void)hipFree(m->memory);
    if (m->memory_t) (void)hipFree(m->memory_t);
    if (m->bank_frag) (void)hipFree(m->bank_frag);
    if (m->faces_outward) (void)hipFree(m->faces_outward);
    if (m->d_parts) (void)hipFree(m->d_parts);
    delete m;
    return NEMO_OK;
}

extern "C" int nemo_mesh_info(nemo_mesh_t m, int32_t* V, int32_t* F, int32_t* C) {
    if (!m) return fail(NEMO_ERR_INVALID, "nemo_mesh_info: null mesh");
    if (V) *V = m->V;
    if (F) *F = m->F;
    if (C) *C = m->C;
    return NEMO_OK;
}

// ------------------------------------------------------------------------------------------
// LDS plan of the rasterising kernels (layout and front end: raster.h)
// ------------------------------------------------------------------------------------------
static int plan_layout(int H, int W, int V, int F, int threads, LdsLayout* L) {
    if (H > kMaxSide || W > kMaxSide || F >= kMaxFaces)
        return fail(NEMO_ERR_UNSUPPORTED, "map %dx%d / %d faces exceed the rasteriser's limits (%d x %d pixels, %d faces)", H, W, F,
                    kMaxSide, kMaxSide, kMaxFaces - 1);
    *L = make_layout(H, W, V, F, true, threads / 64);
    if (L->total > kMaxLds) *L = make_layout(H, W, V, F, false, threads / 64);
    if (L->total > kMaxLds)
        return fail(NEMO_ERR_UNSUPPORTED,
                    "map %dx%d with %d vertices needs %d bytes of LDS per workgroup (limit %d)", H, W, V,
                    L->total, kMaxLds);
    return NEMO_OK;
}

// Layout and key format of the optimisation loop (make_solve_layout, depth_key32).
static int plan_solve(int H, int W, int V, int F, int threads, const float* aabb, RasterCtx* rc, int vblocks = 0) {
    if (H > kMaxSide || W > kMaxSide || F >= kMaxFaces)
        return fail(NEMO_ERR_UNSUPPORTED, "map %dx%d / %d faces exceed the rasteriser's limits (%d x %d pixels, %d faces)", H, W, F,
                    kMaxSide, kMaxSide, kMaxFaces - 1);
    const bool packable = V <= kPackedFaceVerts;
    rc->L = make_solve_layout(H, W, V, F, packable, threads / 64, vblocks);
    if (!packable || rc->L.total > kMaxLds) rc->L = make_solve_layout(H, W, V, F, false, threads / 64, vblocks);
    if (rc->L.total > kMaxLds)
        return fail(NEMO_ERR_UNSUPPORTED,
                    "map %dx%d with %d vertices needs %d bytes of LDS per workgroup (limit %d)", H, W, V,
                    rc->L.total, kMaxLds);
    int fb = 9;                                    // at most 23 bits of depth: they are read from a float's mantissa
    while ((1 << fb) <= F) ++fb;                   // 2^fb > F: no key equals the empty marker
    rc->face_bits = fb;
    depth_range(aabb, fb, &rc->depth_margin, &rc->depth_scale, &rc->depth_qmax);
    return NEMO_OK;
}

// ------------------------------------------------------------------------------------------
// projection (diagnostic / parity)
// ------------------------------------------------------------------------------------------
__global__ void project_kernel(const float* __restrict__ verts, int V, CamConst cc, const float* __restrict__ campos,
                               const float* __restrict__ theta, const float* __restrict__ principal, float def_px,
                               float def_py, float* __restrict__ out_rt, float* __restrict__ out_proj) {
    __shared__ PoseCam pc;
    const int n = blockIdx.x;
    if (threadIdx.x == 0) {
        float C[3] = {campos[3 * n], campos[3 * n + 1], campos[3 * n + 2]};
        float px = principal ? principal[2 * n] : def_px, py = principal ? principal[2 * n + 1] : def_py;
        pose_setup(cc, C, theta[n], px, py, &pc);
        if (out_rt) {
            for (int i = 0; i < 9; ++i) out_rt[12 * n + i] = pc.R[i];
            for (int i = 0; i < 3; ++i) out_rt[12 * n + 9 + i] = pc.T[i];
        }
    }
    __syncthreads();
    if (!out_proj) return;
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
        float Xv[3], xn, yn;
        project_vertex(cc, pc.R, pc.T, pc.bx, pc.by, verts[3 * v], verts[3 * v + 1], verts[3 * v + 2], Xv, &xn, &yn);
        float* o = out_proj + ((size_t)n * V + v) * 5;
        o[0] = Xv[0]; o[1] = Xv[1]; o[2] = Xv[2]; o[3] = xn; o[4] = yn;
    }
}

static int check_cam(const nemo_camera* cam) {
    if (!cam || cam->H <= 0 || cam->W <= 0 || !(cam->focal > 0.0f)) return fail(NEMO_ERR_INVALID, "bad camera");
    return NEMO_OK;
}

extern "C" int nemo_project(nemo_mesh_t m, const nemo_camera* cam, const float* campos, const float* theta,
                            const float* principal, int32_t n, float* out_rt, float* out_proj, void* stream) {
    if (n == 0) return NEMO_OK;
    if (!m || !campos || !theta || n < 0) return fail(NEMO_ERR_INVALID, "nemo_project: bad argument");
    if (check_cam(cam)) return NEMO_ERR_INVALID;
    CamConst cc = make_cam_const(cam->H, cam->W, cam->focal);
    hipLaunchKernelGGL(project_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, m->verts, m->V, cc, campos, theta,
                       principal, cam->px, cam->py, out_rt, out_proj);
    HIP_TRY(hipGetLastError());
    return NEMO_OK;
}

// ------------------------------------------------------------------------------------------
// render
// ------------------------------------------------------------------------------------------
struct RenderArgs {
    RasterCtx rc;
    const float* memory_t;   // [C,V]
    int C;
    const float* campos; const float* theta; const float* principal;
    float def_px, def_py;
    float* out_maps; int32_t* out_p2f; float* out_bary; float* out_zbuf;
    int nearest;             // out_maps takes the feature of the face's vertex with the largest barycentric (mode='nearest')
};

template <bool FL>
__global__ __launch_bounds__(kRasterThreads) void render_kernel(RenderArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Lds s = carve(smem, a.rc.L);
    const int n = blockIdx.x;
    const int H = a.rc.cc.H, W = a.rc.cc.W, HW = H * W, V = a.rc.V;
    raster_prologue(a.rc, s);
    if (threadIdx.x == 0) {
        float C[3] = {a.campos[3 * n], a.campos[3 * n + 1], a.campos[3 * n + 2]};
        float px = a.principal ? a.principal[2 * n] : a.def_px, py = a.principal ? a.principal[2 * n + 1] : a.def_py;
        pose_setup(a.rc.cc, C, a.theta[n], px, py, s.pc);
    }
    __syncthreads();
    project_vertices(a.rc, s);
    raster_faces<FL, true>(a.rc, s);
    for (int p = threadIdx.x; p < HW; p += blockDim.x) {
        unsigned long long key = s.zbuf[p];
        int f = -1;
        float b[3] = {-1.0f, -1.0f, -1.0f}, pz = -1.0f;
        int i0 = 0, i1 = 0, i2 = 0;
        if (key != kEmptyKey) {
            f = (int)(key & 0xFFFFFFFFull);
            load_face<FL>(a.rc, s, f, &i0, &i1, &i2);
            FaceVerts fv = face_verts(s, i0, i1, i2);
            int yi = p / W, xi = p - yi * W;
            face_eval(fv, s.xs[xi], s.ys[yi], b, &pz);
        }
        if (a.out_p2f) a.out_p2f[(size_t)n * HW + p] = f;
        if (a.out_zbuf) a.out_zbuf[(size_t)n * HW + p] = pz;
        if (a.out_bary) {
            float* ob = a.out_bary + ((size_t)n * HW + p) * 3;
            ob[0] = b[0]; ob[1] = b[1]; ob[2] = b[2];
        }
        if (a.out_maps) {
            float* om = a.out_maps + (size_t)n * a.C * HW + p;
            if (f >= 0 && a.nearest) {
                // the first of the largest barycentrics, as argmax over the last axis picks it
                const int iv = b[0] >= b[1] ? (b[0] >= b[2] ? i0 : i2) : (b[1] >= b[2] ? i1 : i2);
                for (int c = 0; c < a.C; ++c) om[(size_t)c * HW] = a.memory_t[(size_t)c * V + iv];
            } else if (f >= 0) {
                for (int c = 0; c < a.C; ++c) {
                    const float* mt = a.memory_t + (size_t)c * V;
                    om[(size_t)c * HW] = (b[0] * mt[i0] + b[1] * mt[i1]) + b[2] * mt[i2];
                }
            } else {
                for (int c = 0; c < a.C; ++c) om[(size_t)c * HW] = 0.0f;
            }
        }
    }
}

template <typename K>
static int set_lds_limit(K kernel, int bytes) {
    HIP_TRY(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    return NEMO_OK;
}

static int render_impl(nemo_mesh_t m, const nemo_camera* cam, const float* campos, const float* theta,
                       const float* principal, int32_t n, float* out_maps, int32_t* out_p2f, float* out_bary,
                       float* out_zbuf, int nearest, void* stream) {
    if (n == 0) return NEMO_OK;
    if (!m || !campos || !theta || n < 0) return fail(NEMO_ERR_INVALID, "nemo_render: bad argument");
    if (check_cam(cam)) return NEMO_ERR_INVALID;
    if (out_maps && !m->memory_t) return fail(NEMO_ERR_STATE, "nemo_render: mesh has no vertex memory (call nemo_mesh_set_memory)");
    RenderArgs a;
    a.rc.verts = m->verts; a.rc.faces = m->faces; a.rc.V = m->V; a.rc.F = m->F;
    a.rc.outward = 0;
    a.rc.cc = make_cam_const(cam->H, cam->W, cam->focal);
    int rcode = plan_layout(cam->H, cam->W, m->V, m->F, kRasterThreads, &a.rc.L);
    if (rcode) return rcode;
    a.memory_t = m->memory_t; a.C = m->C;
    a.campos = campos; a.theta = theta; a.principal = principal; a.def_px = cam->px; a.def_py = cam->py;
    a.out_maps = out_maps; a.out_p2f = out_p2f; a.out_bary = out_bary; a.out_zbuf = out_zbuf;
    a.nearest = nearest;
    auto kern = a.rc.L.faces >= 0 ? render_kernel<true> : render_kernel<false>;
    if (set_lds_limit(kern, a.rc.L.total)) return NEMO_ERR_HIP;
    hipLaunchKernelGGL(kern, dim3(n), dim3(kRasterThreads), a.rc.L.total, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return NEMO_OK;
}

extern "C" int nemo_render(nemo_mesh_t m, const nemo_camera* cam, const float* campos, const float* theta,
                           const float* principal, int32_t n, float* out_maps, int32_t* out_p2f, float* out_bary,
                           float* out_zbuf, void* stream) {
    return render_impl(m, cam, campos, theta, principal, n, out_maps, out_p2f, out_bary, out_zbuf, 0, stream);
}

extern "C" int nemo_render_nearest(nemo_mesh_t m, const nemo_camera* cam, const float* campos, const float* theta,
                                   const float* principal, int32_t n, float* out_maps, void* stream) {
    if (!out_maps) return fail(NEMO_ERR_INVALID, "nemo_render_nearest: out_maps is required");
    return render_impl(m, cam, campos, theta, principal, n, out_maps, nullptr, nullptr, nullptr, 1, stream);
}

// ------------------------------------------------------------------------------------------
// render backward
// ------------------------------------------------------------------------------------------
struct RenderBwdArgs {
    RasterCtx rc;
    const float* memory_t; int C;
    const float* campos; const float* theta; const float* principal;
    float def_px, def_py;
    const float* grad_maps;
    float* g_campos; float* g_theta; float* g_principal;
    int nearest;        // the maps were rendered with mode='nearest': the gradient passes through a softmax over the barycentrics
};

template <bool FL>
__global__ __launch_bounds__(kRasterThreads) void render_backward_kernel(RenderBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Lds s = carve(smem, a.rc.L);
    const int n = blockIdx.x;
    const int H = a.rc.cc.H, W = a.rc.cc.W, HW = H * W, V = a.rc.V;
    raster_prologue(a.rc, s);
    if (threadIdx.x == 0) {
        float C[3] = {a.campos[3 * n], a.campos[3 * n + 1], a.campos[3 * n + 2]};
        float px = a.principal ? a.principal[2 * n] : a.def_px, py = a.principal ? a.principal[2 * n + 1] : a.def_py;
        pose_setup(a.rc.cc, C, a.theta[n], px, py, s.pc);
    }
    __syncthreads();
    project_vertices(a.rc, s);
    raster_faces<FL, true>(a.rc, s);
    unsigned int* seg;
    const int cnt = compact_covered(a.rc, s, &seg);
    PixelGrad acc;
    for (int i = 0; i < 9; ++i) acc.M[i] = 0.f;
    acc.S[0] = acc.S[1] = acc.S[2] = 0.f; acc.gbx = acc.gby = 0.f;
    const ViewInv vinv = view_inv(a.rc.cc, s);
    for (int i = threadIdx.x & 63; i < cnt; i += 64) {
        int xi, yi, f, i0, i1, i2;
        unpack_entry(seg[i], &xi, &yi, &f);
        const int p = yi * W + xi;
        load_face<FL>(a.rc, s, f, &i0, &i1, &i2);
        FaceVerts fv = face_verts(s, i0, i1, i2);
        float gb[3] = {0.f, 0.f, 0.f};
        const float* gm = a.grad_maps + (size_t)n * a.C * HW + p;
        for (int c = 0; c < a.C; ++c) {
            float g = gm[(size_t)c * HW];
            const float* mt = a.memory_t + (size_t)c * V;
            gb[0] += g * mt[i0]; gb[1] += g * mt[i1]; gb[2] += g * mt[i2];
        }
        if (a.nearest) {
            // mode='nearest' (nemo/utils/mesh.py:59-70): the weights are softmax(b) + (one_hot - softmax(b)).detach(), so
            // the gradient w.r.t. the barycentrics is that of the softmax: gb_j <- s_j (gb_j - sum_i gb_i s_i)
            float bb[3], pz;
            face_eval(fv, s.xs[xi], s.ys[yi], bb, &pz);
            const float mx = fmaxf(fmaxf(bb[0], bb[1]), bb[2]);
            const float e0 = expf(bb[0] - mx), e1 = expf(bb[1] - mx), e2 = expf(bb[2] - mx);
            const float inv = 1.0f / ((e0 + e1) + e2);
            const float s0 = e0 * inv, s1 = e1 * inv, s2 = e2 * inv;
            const float dot = (gb[0] * s0 + gb[1] * s1) + gb[2] * s2;
            gb[0] = s0 * (gb[0] - dot); gb[1] = s1 * (gb[1] - dot); gb[2] = s2 * (gb[2] - dot);
        }
        float X0[3], X1[3], X2[3];
        view_coords(vinv, fv, X0, X1, X2);
        pixel_backward(a.rc.cc, fv, X0, X1, X2, s.xs[xi], s.ys[yi], gb, &acc);
    }
    float vals[14];
    for (int i = 0; i < 9; ++i) vals[i] = acc.M[i];
    vals[9] = acc.S[0]; vals[10] = acc.S[1]; vals[11] = acc.S[2]; vals[12] = acc.gbx; vals[13] = acc.gby;
    block_sum<14>(vals, s.red, s.tot);
    if (threadIdx.x == 0) {
        float gC[3], gth, gpx, gpy;
        pose_backward(a.rc.cc, *s.pc, s.tot, s.tot + 9, s.tot[12], s.tot[13], gC, &gth, &gpx, &gpy);
        if (a.g_campos) { a.g_campos[3 * n] = gC[0]; a.g_campos[3 * n + 1] = gC[1]; a.g_campos[3 * n + 2] = gC[2]; }
        if (a.g_theta) a.g_theta[n] = gth;
        if (a.g_principal) { a.g_principal[2 * n] = gpx; a.g_principal[2 * n + 1] = gpy; }
    }
}

static int render_backward_impl(nemo_mesh_t m, const nemo_camera* cam, const float* campos, const float* theta,
                                const float* principal, int32_t n, const float* grad_maps, float* g_campos,
                                float* g_theta, float* g_principal, int nearest, void* stream) {
    if (n == 0) return NEMO_OK;
    if (!m || !campos || !theta || !grad_maps || n < 0) return fail(NEMO_ERR_INVALID, "nemo_render_backward: bad argument");
    if (check_cam(cam)) return NEMO_ERR_INVALID;
    if (!m->memory_t) return fail(NEMO_ERR_STATE, "nemo_render_backward: mesh has no vertex memory");
    RenderBwdArgs a;
    a.rc.verts = m->verts; a.rc.faces = m->faces; a.rc.V = m->V; a.rc.F = m->F;
    a.rc.outward = 0;
    a.rc.cc = make_cam_const(cam->H, cam->W, cam->focal);
    int rcode = plan_layout(cam->H, cam->W, m->V, m->F, kRasterThreads, &a.rc.L);
    if (rcode) return rcode;
    a.memory_t = m->memory_t; a.C = m->C;
    a.campos = campos; a.theta = theta; a.principal = principal; a.def_px = cam->px; a.def_py = cam->py;
    a.grad_maps = grad_maps; a.g_campos = g_campos; a.g_theta = g_theta; a.g_principal = g_principal;
    a.nearest = nearest;
    auto kern = a.rc.L.faces >= 0 ? render_backward_kernel<true> : render_backward_kernel<false>;
    if (set_lds_limit(kern, a.rc.L.total)) return NEMO_ERR_HIP;
    hipLaunchKernelGGL(kern, dim3(n), dim3(kRasterThreads), a.rc.L.total, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return NEMO_OK;
}

extern "C" int nemo_render_backward(nemo_mesh_t m, const nemo_camera* cam, const float* campos, const float* theta,
                                    const float* principal, int32_t n, const float* grad_maps, float* g_campos,
                                    float* g_theta, float* g_principal, void* stream) {
    return render_backward_impl(m, cam, campos, theta, principal, n, grad_maps, g_campos, g_theta, g_principal, 0, stream);
}

extern "C" int nemo_render_nearest_backward(nemo_mesh_t m, const nemo_camera* cam, const float* campos, const float* theta,
                                            const float* principal, int32_t n, const float* grad_maps, float* g_campos,
                                            float* g_theta, float* g_principal, void* stream) {
    return render_backward_impl(m, cam, campos, theta, principal, n, grad_maps, g_campos, g_theta, g_principal, 1, stream);
}

// ------------------------------------------------------------------------------------------
// clutter score
// ------------------------------------------------------------------------------------------
__global__ void clutter_kernel(const float* __restrict__ feat, const float* __restrict__ clutter, int C, int HW, int K,
                               float* __restrict__ out) {
    extern __shared__ float cl[];   // [K*C]
    for (int i = threadIdx.x; i < K * C; i += blockDim.x) cl[i] = clutter[i];
    __syncthreads();
    const int b = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const float* f = feat + (size_t)b * C * HW + p;
    float best = 0.0f;
    for (int k = 0; k < K; ++k) {
        float acc = 0.0f;
        const float* ck = cl + k * C;
#pragma unroll 8
        for (int c = 0; c < C; ++c) acc = __builtin_fmaf(f[(size_t)c * HW], ck[c], acc);
        best = (k == 0) ? acc : max_nan(best, acc);     // torch.max over the clutter vectors: a NaN score stays NaN
    }
    out[(size_t)b * HW + p] = best;
}

extern "C" int nemo_clutter_score(const float* feat, const float* clutter, int32_t B, int32_t C, int32_t HW, int32_t K,
                                  float* out, void* stream) {
    if (B == 0) return NEMO_OK;
    if (!feat || !clutter || !out || B < 0 || C <= 0 || HW <= 0 || K <= 0)
        return fail(NEMO_ERR_INVALID, "nemo_clutter_score: bad argument");
    if ((size_t)K * C * 4 > 64 * 1024) return fail(NEMO_ERR_UNSUPPORTED, "nemo_clutter_score: K*C too large");
    dim3 grid((HW + 255) / 256, B), block(256);
    hipLaunchKernelGGL(clutter_kernel, grid, block, (size_t)K * C * 4, (hipStream_t)stream, feat, clutter, C, HW, K, out);
    HIP_TRY(hipGetLastError());
    return NEMO_OK;
}

// ------------------------------------------------------------------------------------------
// re-centring resample (align_no_centered)
// ------------------------------------------------------------------------------------------
// out[n,c,y,x] = bilinear sample (align_corners = false) of src[n,c] at the grid
//   gx(x) = (1 - t_x) * x_min[n] + t_x * x_max[n],  t_x = x / (w0 - 1)     (same for y)
// with  x_min/max = 2 ps_x / w1 -/+ w0 / (w1 * rate) - 2 pt_x / w0,  rate = dist_source / dist_target.
// border != 0: coordinates are clamped to the source (padding_mode='border'), else taps outside read 0.
// src_count: map n reads source image n % src_count (several re-centred copies of one batch in one launch)
__global__ void align_kernel(const float* __restrict__ src, const float* __restrict__ dist_src,
                             const float* __restrict__ dist_tgt, const float* __restrict__ p_src,
                             const float* __restrict__ p_tgt, int C, int h1, int w1, int h0, int w0, int border, int src_count,
                             float* __restrict__ out) {
    const int n = blockIdx.y;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= h0 * w0) return;
    const int y = q / w0, x = q - y * w0;
    const AlignTaps t = align_taps(dist_src[n], dist_tgt[n], p_src[2 * n], p_src[2 * n + 1], p_tgt[2 * n], p_tgt[2 * n + 1], h1, w1, h0,
                                   w0, border, x, y);
    const float* sp = src + (size_t)(n % src_count) * C * h1 * w1;
    float* op = out + (size_t)n * C * h0 * w0 + q;
    for (int c = 0; c < C; ++c) op[(size_t)c * h0 * w0] = align_sample(sp + (size_t)c * h1 * w1, t);
}

extern "C" int nemo_align_maps(const float* src, const float* dist_src, const float* dist_tgt, const float* p_src,
                               const float* p_tgt, int32_t n, int32_t C, int32_t h1, int32_t w1, int32_t h0, int32_t w0,
                               int32_t border, float* out, void* stream) {
    if (n == 0) return NEMO_OK;
    if (!src || !dist_src || !dist_tgt || !p_src || !p_tgt || !out || n < 0 || C <= 0 || h1 <= 0 || w1 <= 0 || h0 <= 0 || w0 <= 0)
        return fail(NEMO_ERR_INVALID, "nemo_align_maps: bad argument");
    if (n > 65535) return fail(NEMO_ERR_UNSUPPORTED, "nemo_align_maps: more than 65535 maps per call");
    dim3 grid((h0 * w0 + 255) / 256, n);
    hipLaunchKernelGGL(align_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dist_src, dist_tgt, p_src, p_tgt, C, h1,
                       w1, h0, w0, border, n, out);
    HIP_TRY(hipGetLastError());
    return NEMO_OK;
}

// ------------------------------------------------------------------------------------------
// dense pose-grid scoring
// ------------------------------------------------------------------------------------------
constexpr int GS_TN = 12, GS_TB = 16, GS_PIX = 256;

// partial[ptile][b][n] = sum over the tile's pixels of max(obj, clu) (or obj when clu == nullptr)
__global__ __launch_bounds__(GS_PIX) void grid_score_kernel(const float* __restrict__ pre, const float* __restrict__ feat,
                                                           const float* __restrict__ clu, int N, int B, int C, int HW,
                                                           float* __restrict__ partial) {
    __shared__ float red[4][GS_TN * GS_TB];
    const int p = blockIdx.x * GS_PIX + threadIdx.x;
    const int n0 = blockIdx.y * GS_TN, b0 = blockIdx.z * GS_TB;
    const bool valid = p < HW;
    const int pc = valid ? p : HW - 1;
    float acc[GS_TN][GS_TB];
#pragma unroll
    for (int i = 0; i < GS_TN; ++i)
#pragma unroll
        for (int j = 0; j < GS_TB; ++j) acc[i][j] = 0.0f;
    const float* pp[GS_TN];
    const float* fp[GS_TB];
#pragma unroll
    for (int i = 0; i < GS_TN; ++i) pp[i] = pre + (size_t)min(n0 + i, N - 1) * C * HW + pc;
#pragma unroll
    for (int j = 0; j < GS_TB; ++j) fp[j] = feat + (size_t)min(b0 + j, B - 1) * C * HW + pc;
    for (int c = 0; c < C; ++c) {
        float pv[GS_TN], fv[GS_TB];
#pragma unroll
        for (int i = 0; i < GS_TN; ++i) pv[i] = pp[i][(size_t)c * HW];
#pragma unroll
        for (int j = 0; j < GS_TB; ++j) fv[j] = fp[j][(size_t)c * HW];
#pragma unroll
        for (int i = 0; i < GS_TN; ++i)
#pragma unroll
            for (int j = 0; j < GS_TB; ++j) acc[i][j] = __builtin_fmaf(pv[i], fv[j], acc[i][j]);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < GS_TB; ++j) {
        float cl = 0.0f;
        if (clu) cl = clu[(size_t)min(b0 + j, B - 1) * HW + pc];
#pragma unroll
        for (int i = 0; i < GS_TN; ++i) {
            float v = clu ? max_nan(acc[i][j], cl) : acc[i][j];
            v = valid ? v : 0.0f;
            v = wave_sum(v);
            if (lane == 0) red[wave][i * GS_TB + j] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x < GS_TN * GS_TB) {
        int i = threadIdx.x / GS_TB, j = threadIdx.x % GS_TB;
        if (n0 + i < N && b0 + j < B) {
            float v = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
            partial[((size_t)blockIdx.x * B + (b0 + j)) * N + (n0 + i)] = v;
        }
    }
}

// loss[b][n] = 1 - (sum_tiles partial / HW - sum_p clu / HW); argmin over n (first minimum).
__global__ void grid_finalize_kernel(const float* __restrict__ partial, const float* __restrict__ clu, int N, int B, int HW,
                                     int ntiles, float* __restrict__ out_loss, int32_t* __restrict__ out_argmin,
                                     float* __restrict__ out_min) {
    __shared__ float sred[256];
    __shared__ int ired[256];
    __shared__ float clu_mean;
    const int b = blockIdx.x, t = threadIdx.x;
    float s = 0.0f;
    if (clu)
        for (int p = t; p < HW; p += 256) s += clu[(size_t)b * HW + p];
    sred[t] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) sred[t] += sred[t + o];
        __syncthreads();
    }
    if (t == 0) clu_mean = sred[0] / (float)HW;
    __syncthreads();
    float best = __builtin_inff();
    int besti = 0x7FFFFFFF;
    for (int n = t; n < N; n += 256) {
        float acc = 0.0f;
        for (int k = 0; k < ntiles; ++k) acc += partial[((size_t)k * B + b) * N + n];
        float loss = 1.0f - (acc / (float)HW - clu_mean);
        if (out_loss) out_loss[(size_t)b * N + n] = loss;
        if (loss < best || (loss == best && n < besti)) { best = loss; besti = n; }
    }
    sred[t] = best; ired[t] = besti;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) {
            float ob = sred[t + o]; int oi = ired[t + o];
            if (ob < sred[t] || (ob == sred[t] && oi < ired[t])) { sred[t] = ob; ired[t] = oi; }
        }
        __syncthreads();
    }
    if (t == 0) {
        out_argmin[b] = ired[0] == 0x7FFFFFFF ? 0 : ired[0];
        if (out_min) out_min[b] = sred[0];
    }
}

extern "C" size_t nemo_grid_score_workspace(int32_t N, int32_t B, int32_t HW) {
    if (N <= 0 || B <= 0 || HW <= 0) return 0;
    size_t tiles = (size_t)(HW + GS_PIX - 1) / GS_PIX;
    return tiles * (size_t)B * (size_t)N * sizeof(float);
}

extern "C" int nemo_grid_score(const float* pre, const float* feat, const float* clu, int32_t N, int32_t B, int32_t C,
                               int32_t HW, float* out_loss, int32_t* out_argmin, float* out_min, void* workspace,
                               size_t workspace_bytes, void* stream) {
    if (B == 0) return NEMO_OK;
    if (!pre || !feat || !out_argmin || N <= 0 || B < 0 || C <= 0 || HW <= 0)
        return fail(NEMO_ERR_INVALID, "nemo_grid_score: bad argument");
    size_t need = nemo_grid_score_workspace(N, B, HW);
    if (!workspace || workspace_bytes < need)
        return fail(NEMO_ERR_INVALID, "nemo_grid_score: workspace of %zu bytes required, %zu given", need, workspace_bytes);
    int tiles = (HW + GS_PIX - 1) / GS_PIX;
    dim3 grid(tiles, (N + GS_TN - 1) / GS_TN, (B + GS_TB - 1) / GS_TB);
    if (grid.y > 65535 || grid.z > 65535) return fail(NEMO_ERR_UNSUPPORTED, "nemo_grid_score: grid too large");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(grid_score_kernel, grid, dim3(GS_PIX), 0, st, pre, feat, clu, N, B, C, HW, (float*)workspace);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(grid_finalize_kernel, dim3(B), dim3(256), 0, st, (const float*)workspace, clu, N, B, HW, tiles,
                       out_loss, out_argmin, out_min);
    HIP_TRY(hipGetLastError());
    return NEMO_OK;
}

// ---- the same scoring for callers that cannot allocate ----
// The per-tile partial sums go to a fixed device array instead of a caller workspace.  In host code the name
// `scratch` is only a handle for hipGetSymbolAddress: its device address is looked up once per device (each GPU
// holds its own copy of the array at its own address) and the kernels always take that looked-up pointer.
static __device__ __align__(256) float scratch[NEMO_STATIC_SCRATCH_BYTES / sizeof(float)];
constexpr int kMaxDevices = 64;
static float* g_scratch_ptr[kMaxDevices] = {};
static std::mutex g_scratch_mutex;

// The array's address on the device that will run the call: the device that owns `stream` (the current device for the
// null stream).  The first call on a device looks the address up (with that device current; the runtime may load the
// code object then, which is why that first call must not come under graph capture); later calls only read the cache.
static int static_scratch(hipStream_t stream, float** out) {
    int cur = -1;
    HIP_TRY(hipGetDevice(&cur));
    int dev = cur;
    if (stream) HIP_TRY(hipStreamGetDevice(stream, &dev));
    if (dev < 0 || dev >= kMaxDevices) return fail(NEMO_ERR_UNSUPPORTED, "device ordinal %d outside 0..%d", dev, kMaxDevices - 1);
    if (dev != cur)
        return fail(NEMO_ERR_STATE, "nemo_grid_score_static: the stream belongs to device %d, the current device is %d (make the "
                    "stream's device current: every GPU holds its own copy of the scratch array)", dev, cur);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(NEMO_ERR_HIP, "hipGetSymbolAddress returned a null address for the static scratch array");
        g_scratch_ptr[dev] = (float*)p;
    }
    *out = g_scratch_ptr[dev];
    return NEMO_OK;
}

extern "C" int nemo_grid_score_static(const float* pre, const float* feat, const float* clu, int32_t N, int32_t B, int32_t C,
                                      int32_t HW, float* out_loss, int32_t* out_argmin, float* out_min, void* stream) {
    if (B == 0) return NEMO_OK;
    if (!pre || !feat || !out_argmin || N <= 0 || B < 0 || C <= 0 || HW <= 0)
        return fail(NEMO_ERR_INVALID, "nemo_grid_score_static: bad argument");
    const size_t need = nemo_grid_score_workspace(N, B, HW);
    if (need > (size_t)NEMO_STATIC_SCRATCH_BYTES)
        return fail(NEMO_ERR_UNSUPPORTED,
                    "nemo_grid_score_static: %d poses x %d images x %d pixels need %zu bytes of partial sums, the static scratch "
                    "array holds %zu (use nemo_grid_score with a workspace)", N, B, HW, need, (size_t)NEMO_STATIC_SCRATCH_BYTES);
    float* partial = nullptr;
    const int rc = static_scratch((hipStream_t)stream, &partial);
    if (rc) return rc;
    // every partial sum the finalize kernel reads is written by the scoring kernel of this call: nothing to clear
    return nemo_grid_score(pre, feat, clu, N, B, C, HW, out_loss, out_argmin, out_min, partial, need, stream);
}

// ---- matrix-core form (kernels in grid_mfma.hip) ----
namespace nemo {
bool grid_mfma_supported(int C);
int grid_chunks(int HW);
size_t grid_pack_bytes(int N, int C, int HW);
size_t grid_feat_bytes(int B, int C, int HW);
int launch_grid_pack(const float* pre, void* packed, int N, int C, int HW, hipStream_t st);
int launch_grid_mfma(const void* packed, const float* feat, const float* clu, int N, int B, int C, int HW, void* featp,
                     float* partial, hipStream_t st);
int launch_grid_mfma_clutter(const void* packed, const float* feat, const float* clutter, int K, int N, int B, int C, int HW,
                             float* out_clu, void* featp, float* partial, hipStream_t st);
int launch_grid_mfma_fragments(const void* packed, const void* featp, const float* clu, int N, int B, int C, int HW, float* partial,
                               hipStream_t st);
}

extern "C" size_t nemo_grid_pack_bytes(int32_t N, int32_t C, int32_t HW) {
    if (N <= 0 || C <= 0 || HW <= 0 || !nemo::grid_mfma_supported(C)) return 0;
    return nemo::grid_pack_bytes(N, C, HW);
}

extern "C" int nemo_grid_pack(const float* pre, int32_t N, int32_t C, int32_t HW, void* packed, void* stream) {
    if (!pre || !packed || N <= 0 || C <= 0 || HW <= 0) return fail(NEMO_ERR_INVALID, "nemo_grid_pack: bad argument");
    if (!nemo::grid_mfma_supported(C))
        return fail(NEMO_ERR_UNSUPPORTED, "nemo_grid_pack: %d channels (the matrix-core form takes 16, 32, 64 or 128)", C);
    if (nemo::launch_grid_pack(pre, packed, N, C, HW, (hipStream_t)stream))
        return fail(NEMO_ERR_UNSUPPORTED, "nemo_grid_pack: grid too large for one launch");
    HIP_TRY(hipGetLastError());
    return NEMO_OK;
}


extern "C" size_t nemo_grid_score_packed_workspace(int32_t N, int32_t B, int32_t C, int32_t HW) {
    if (N <= 0 || B <= 0 || C <= 0 || HW <= 0 || !nemo::grid_mfma_supported(C)) return 0;
    return align256(nemo::grid_feat_bytes(B, C, HW)) + (size_t)nemo::grid_chunks(HW) * B * N * sizeof(float);
}

extern "C" int nemo_grid_score_packed(const void* packed, const float* feat, const float* clu, int32_t N, int32_t B, int32_t C,
                                      int32_t HW, float* out_loss, int32_t* out_argmin, float* out_min, void* workspace,
                                      size_t workspace_bytes, void* stream) {
    if (B == 0) return NEMO_OK;
    if (!packed || !feat || !out_argmin || N <= 0 || B < 0 || C <= 0 || HW <= 0)
        return fail(NEMO_ERR_INVALID, "nemo_grid_score_packed: bad argument");
    if (!nemo::grid_mfma_supported(C))
        return fail(NEMO_ERR_UNSUPPORTED, "nemo_grid_score_packed: %d channels (takes 16, 32, 64 or 128)", C);
    const size_t need = nemo_grid_score_packed_workspace(N, B, C, HW);
    if (!workspace || workspace_bytes < need)
        return fail(NEMO_ERR_INVALID, "nemo_grid_score_packed: workspace of %zu bytes required, %zu given", need, workspace_bytes);
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)((char*)workspace + align256(nemo::grid_feat_bytes(B, C, HW)));
    if (nemo::launch_grid_mfma(packed, feat, clu, N, B, C, HW, workspace, partial, st))
        return fail(NEMO_ERR_UNSUPPORTED, "nemo_grid_score_packed: batch too large for one launch");
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(grid_finalize_kernel, dim3(B), dim3(256), 0, st, (const float*)partial, clu, N, B, HW, nemo::grid_chunks(HW),
                       out_loss, out_argmin, out_min);
    HIP_TRY(hipGetLastError());
    return NEMO_OK;
}

extern "C" size_t nemo_grid_fragments_bytes(int32_t B, int32_t C, int32_t HW) {
    if (B <= 0 || C <= 0 || HW <= 0 || !nemo::grid_mfma_supported(C)) return 0;
    return nemo::grid_feat_bytes(B, C, HW);
}

extern "C" size_t nemo_grid_score_fragments_workspace(int32_t N, int32_t B, int32_t HW) {
    if (N <= 0 || B <= 0 || HW <= 0) return 0;
    return (size_t)nemo::grid_chunks(HW) * B * N * sizeof(float);
}

extern "C" int nemo_grid_score_fragments(const void* packed, const void* fragments, const float* clu, int32_t N, int32_t B, int32_t C,
                                         int32_t HW, float* out_loss, int32_t* out_argmin, float* out_min, void* workspace,
                                         size_t workspace_bytes, void* stream) {
    if (B == 0) return NEMO_OK;
    if (!packed || !fragments || !out_argmin || N <= 0 || B < 0 || C <= 0 || HW <= 0)
        return fail(NEMO_ERR_INVALID, "nemo_grid_score_fragments: bad argument");
    if (!nemo::grid_mfma_supported(C))
        return fail(NEMO_ERR_UNSUPPORTED, "nemo_grid_score_fragments: %d channels (takes 16, 32, 64 or 128)", C);
    const size_t need = nemo_grid_score_fragments_workspace(N, B, HW);
    if (!workspace || workspace_bytes < need)
        return fail(NEMO_ERR_INVALID, "nemo_grid_score_fragments: workspace of %zu bytes required, %zu given", need, workspace_bytes);
    hipStream_t st = (hipStream_t)stream;
    if (nemo::launch_grid_mfma_fragments(packed, fragments, clu, N, B, C, HW, (float*)workspace, st))
        return fail(NEMO_ERR_UNSUPPORTED, "nemo_grid_score_fragments: batch too large for one launch");
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(grid_finalize_kernel, dim3(B), dim3(256), 0, st, (const float*)workspace, clu, N, B, HW, nemo::grid_chunks(HW),
                       out_loss, out_argmin, out_min);
    HIP_TRY(hipGetLastError());
    return NEMO_OK;
}

extern "C" int nemo_clutter_grid_score_packed(const void* packed, const float* feat, const float* clutter, int32_t K, int32_t N, int32_t B,
                                              int32_t C, int32_t HW, float* out_clutter_score, float* out_loss, int32_t* out_argmin,
                                              float* out_min, void* workspace, size_t workspace_bytes, void* stream) {
    if (B == 0) return NEMO_OK;
    if (!packed || !feat || !clutter || !out_clutter_score || !out_argmin || K <= 0 || N <= 0 || B < 0 || C <= 0 || HW <= 0)
        return fail(NEMO_ERR_INVALID, "nemo_clutter_grid_score_packed: bad argument");
    if (!nemo::grid_mfma_supported(C))
        return fail(NEMO_ERR_UNSUPPORTED, "nemo_clutter_grid_score_packed: %d channels (takes 16, 32, 64 or 128)", C);
    const size_t need = nemo_grid_score_packed_workspace(N, B, C, HW);
    if (!workspace || workspace_bytes < need)
        return fail(NEMO_ERR_INVALID, "nemo_clutter_grid_score_packed: workspace of %zu bytes required, %zu given", need, workspace_bytes);
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)((char*)workspace + align256(nemo::grid_feat_bytes(B, C, HW)));
    const int rc = nemo::launch_grid_mfma_clutter(packed, feat, clutter, K, N, B, C, HW, out_clutter_score, workspace, partial, st);
    if (rc < 0) return fail(NEMO_ERR_UNSUPPORTED, "nemo_clutter_grid_score_packed: batch too large for one launch");
    if (rc > 0) {       // more clutter vectors than the fused pass takes: the two steps one after the other
        const int r1 = nemo_clutter_score(feat, clutter, B, C, HW, K, out_clutter_score, stream);
        if (r1) return r1;
        return nemo_grid_score_packed(packed, feat, out_clutter_score, N, B, C, HW, out_loss, out_argmin, out_min, workspace,
                                      workspace_bytes, stream);
    }
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(grid_finalize_kernel, dim3(B), dim3(256), 0, st, (const float*)partial, (const float*)out_clutter_score, N, B, HW,
                       nemo::grid_chunks(HW), out_loss, out_argmin, out_min);
    HIP_TRY(hipGetLastError());
    return NEMO_OK;
}

// Pose-grid scoring of re-centred copies of a batch without writing the copies (6-D initialisation):
// virtual image bv = source image bv % B_src resampled with (dist_src, dist_tgt, p_src, p_tgt)[bv].
extern "C" size_t nemo_grid_score_recentred_workspace(int32_t N, int32_t n_virtual, int32_t C, int32_t h0, int32_t w0) {
    const size_t base = nemo_grid_score_packed_workspace(N, n_virtual, C, h0 * w0);
    return base ? align256(base) + (size_t)n_virtual * h0 * w0 * sizeof(float) : 0;
}

extern "C" int nemo_grid_score_recentred(const void* packed, const float* feat, const float* clu, const float* dist_src,
                                         const float* dist_tgt, const float* p_src, const float* p_tgt, int32_t N, int32_t B_src,
                                         int32_t n_virtual, int32_t C, int32_t h1, int32_t w1, int32_t h0, int32_t w0,
                                         int32_t border, float* out_loss, int32_t* out_argmin, float* out_min, void* workspace,
                                         size_t workspace_bytes, void* stream) {
    if (n_virtual == 0) return NEMO_OK;
    if (!packed || !feat || !dist_src || !dist_tgt || !p_src || !p_tgt || !out_argmin || N <= 0 || B_src <= 0 || n_virtual < 0 ||
        C <= 0 || h1 <= 0 || w1 <= 0 || h0 <= 0 || w0 <= 0)
        return fail(NEMO_ERR_INVALID, "nemo_grid_score_recentred: bad argument");
    if (!nemo::grid_mfma_supported(C))
        return fail(NEMO_ERR_UNSUPPORTED, "nemo_grid_score_recentred: %d channels (takes 16, 32, 64 or 128)", C);
    if (n_virtual > 65535) return fail(NEMO_ERR_UNSUPPORTED, "nemo_grid_score_recentred: more than 65535 copies per call");
    const int HW = h0 * w0;
    const size_t need = nemo_grid_score_recentred_workspace(N, n_virtual, C, h0, w0);
    if (!workspace || workspace_bytes < need)
        return fail(NEMO_ERR_INVALID, "nemo_grid_score_recentred: workspace of %zu bytes required, %zu given", need, workspace_bytes);
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)((char*)workspace + align256(nemo::grid_feat_bytes(n_virtual, C, HW)));
    float* clu_v = nullptr;
    if (clu) {          // the clutter maps are small: re-centred into the workspace by the one-channel resample
        clu_v = (float*)((char*)workspace + align256(nemo_grid_score_packed_workspace(N, n_virtual, C, HW)));
        dim3 grid((HW + 255) / 256, n_virtual);
        hipLaunchKernelGGL(align_kernel, grid, dim3(256), 0, st, clu, dist_src, dist_tgt, p_src, p_tgt, 1, h1, w1, h0, w0, border, B_src,
                           clu_v);
    }
    if (nemo::launch_grid_mfma_recentred(packed, feat, clu_v, N, n_virtual, C, B_src, h1, w1, h0, w0, border, dist_src, dist_tgt, p_src,
                                         p_tgt, workspace, partial, st))
        return fail(NEMO_ERR_UNSUPPORTED, "nemo_grid_score_recentred: batch too large for one launch");
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(grid_finalize_kernel, dim3(n_virtual), dim3(256), 0, st, (const float*)partial, (const float*)clu_v, N, n_virtual,
                       HW, nemo::grid_chunks(HW), out_loss, out_argmin, out_min);
    HIP_TRY(hipGetLastError());
    return NEMO_OK;
}

// ------------------------------------------------------------------------------------------
// vertex similarity (fast_inference)
// ------------------------------------------------------------------------------------------
__global__ void vertex_similarity_kernel(const float* __restrict__ table, const float* __restrict__ clu,
                                         const int32_t* __restrict__ boxes, int V, int H, int W, float* __restrict__ out) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    int r0 = 0, r1 = H, c0 = 0, c1 = W;
    if (boxes) { r0 = boxes[4 * b]; r1 = boxes[4 * b + 1]; c0 = boxes[4 * b + 2]; c1 = boxes[4 * b + 3]; }
    r0 = max(r0, 0); c0 = max(c0, 0); r1 = min(r1, H); c1 = min(c1, W);
    const int bw = max(c1 - c0, 0), bh = max(r1 - r0, 0), np = bw * bh, HW = H * W;
    float s = 0.0f;
    for (int q = threadIdx.x; q < np; q += blockDim.x) {
        int p = (r0 + q / bw) * W + c0 + q % bw;
        const float* t = table + (size_t)b * table_image_elems(V, HW);
        float best = -__builtin_inff();
        for (int v = 0; v < V; ++v) best = max_nan(best, t[table_offset((V + 3) >> 2, v, p)]);
        s += max_nan(best, clu[(size_t)b * HW + p]);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[b] = (((red[0] + red[1]) + red[2]) + red[3]) / (float)np;
}

extern "C" int nemo_vertex_similarity(nemo_mesh_t m, const float* table, const float* clu, const int32_t* boxes, int32_t B,
                                      int32_t H, int32_t W, float* out, void* stream) {
    if (B == 0) return NEMO_OK;
    if (!m || !table || !clu || !out || B < 0 || H <= 0 || W <= 0) return fail(NEMO_ERR_INVALID, "nemo_vertex_similarity: bad argument");
    hipLaunchKernelGGL(vertex_similarity_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, table, clu, boxes, m->V, H, W, out);
    HIP_TRY(hipGetLastError());
    return NEMO_OK;
}

// The same similarity straight from the feature maps: the vertex x pixel products are formed tile by tile on the matrix
// cores and reduced to the per-pixel maximum over the vertices before anything is stored (table_gemm.hip).
extern "C" size_t nemo_vertex_similarity_workspace(int32_t B, int32_t HW) {
    if (B <= 0 || HW <= 0) return 0;
    return align256(sizeof(float) * (size_t)B * ((HW + 127) / 128));
}

extern "C" int nemo_vertex_similarity_features(nemo_mesh_t m, const float* feat, const float* clu, const int32_t* boxes, int32_t B,
                                               int32_t H, int32_t W, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    if (B == 0) return NEMO_OK;
    if (!m || !feat || !clu || !out || !workspace || B < 0 || H <= 0 || W <= 0)
        return fail(NEMO_ERR_INVALID, "nemo_vertex_similarity_features: bad argument");
    if (!m->memory) return fail(NEMO_ERR_STATE, "nemo_vertex_similarity_features: the mesh has no vertex memory");
    if (!m->bank_frag)
        return fail(NEMO_ERR_UNSUPPORTED, "nemo_vertex_similarity_features: %d channels (takes 16, 32, 64 or 128: use "
                    "nemo_similarity_table + nemo_vertex_similarity)", m->C);
    if (workspace_bytes < nemo_vertex_similarity_workspace(B, H * W))
        return fail(NEMO_ERR_INVALID, "nemo_vertex_similarity_features: workspace of %zu bytes, %zu needed", workspace_bytes,
                    nemo_vertex_similarity_workspace(B, H * W));
    if (nemo::launch_vertex_max(m->bank_frag, feat, clu, boxes, m->V, m->C, H, W, B, (float*)workspace, out, (hipStream_t)stream))
        return fail(NEMO_ERR_UNSUPPORTED, "nemo_vertex_similarity_features: batch too large for one launch");
    HIP_TRY(hipGetLastError());
    return NEMO_OK;
}

// ------------------------------------------------------------------------------------------
// pose optimisation
// ------------------------------------------------------------------------------------------
#ifdef NEMO_STAMPS
#define STAMP(i) do { __syncthreads(); if (tid == 0) { long long t_ = clock64(); stamp_acc[i] += t_ - stamp_t; stamp_t = t_; } } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

struct OptState {
    float P[6];                 // parameter values (mirror of the registers of wave 0)
    float first_loss, last_loss;
    float base_max, base_clu, box_max, box_clu, box_n;   // per-image constants of the loss (see solve_kernel)
    float out_lo[3], out_hi[3];  // the mesh's bounding box with a margin: a camera beyond it is outside the solid
};

struct SolveArgs {
    RasterCtx rc;
    float* table; const float* clu; const int32_t* hyp_to_image;
    const float* init_campos; const float* init_theta; const float* init_principal;
    const int32_t* score_boxes;
    // table tiles computed on demand (ONDEMAND): the images' feature maps [n_img, C, HW] and the vertex bank as matrix-core
    // fragments (the two operands of the table product) and what it takes to address them; `table` is then written by the kernel
    const float* feat; const uint4* bank; size_t apiece; const BankMeta* meta; int nks, nfb;
    const MeshPart* parts; const int32_t* hyp_to_part; int n_parts;   // a handle of several meshes: hypothesis h rasterises part hyp_to_part[h]
    // ONDEMAND with the first step's tiles computed in bulk: per hypothesis one row of mask words per 32-pixel tile naming
    // the vertex blocks whose tiles tile_bulk_kernel computes (written by the FOOTPRINT launch, read by the loop's set-up)
    unsigned int* need; int npt32; int dilate;
    int table_vg;                        // vertex groups of 4 per pixel tile of the table (all vertices of the handle)
    float def_px, def_py;
    float tscale, inv_tscale;    // warp mode: the reference's translation_scale (flow units per pixel) and its inverse
    int n_img; float inv_norm; int dof;
    int steps, lr_every;
    double lr, beta1, beta2, eps, gamma;
    float* out_pose; float* out_hist;
};

// WARP: the keypoint-initialised solver's translation model (solve_pose.py:258-276): parameters 4, 5 are a
// 2-D flow; the map rendered at the default principal point is resampled bilinearly at pixel + flow / scale
// (flow_warp.py: grid_sample, align_corners=True, border padding) before it is compared with the image.
// Two workgroups per compute unit (four waves per SIMD, 128 registers): while one hypothesis is in a serial
// section (pose set-up, the gradient chain, Adam) or waits on LDS, the other one's waves issue.
// FOOTPRINT (with ONDEMAND): the launch in front of the loop that only looks: it takes the first step as far as the
// covered entries and records the table tiles they read (SolveArgs::need) for tile_bulk_kernel; nothing else is written.
template <bool FL, bool WARP, bool ONDEMAND = false, bool FOOTPRINT = false>
__global__ __launch_bounds__(kSolveThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void solve_kernel(SolveArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Lds s = carve(smem, a.rc.L);
    // Runs of eight consecutive hypotheses go to workgroups whose block ids are 8 apart: workgroups are dealt
    // round-robin over the 8 XCDs, so the hypotheses of one image (consecutive when an image has several) run at the
    // same time behind the same L2 and share the lines of its similarity table.
    int h = blockIdx.x;
    if (h < (int)(gridDim.x & ~63u)) { const int b = h & 63; h = (h & ~63) + (b & 7) * 8 + (b >> 3); }
    const int tid = threadIdx.x;
    const int H = a.rc.cc.H, W = a.rc.cc.W, HW = H * W;
    const int img = a.hyp_to_image ? a.hyp_to_image[h] : h;
    // The mesh this hypothesis renders: the whole handle, or one of its parts (uniform over the workgroup).
    RasterCtx rc = a.rc;
    int part = 0;
    if (a.hyp_to_part) {
        part = a.hyp_to_part[h];
        if ((unsigned)part < (unsigned)a.n_parts) {
            const MeshPart q = a.parts[part];
            rc.verts = a.rc.verts + 3 * (size_t)q.v0; rc.faces = a.rc.faces + 3 * (size_t)q.f0; rc.vbase = q.v0;
            rc.V = q.nv; rc.F = q.nf; rc.outward = __builtin_amdgcn_readfirstlane(q.outward);
            // (uniform values, pinned to scalar registers: they live over the whole step loop)
            auto uni = [](float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); };
#pragma unroll
            for (int k = 0; k < 6; ++k) rc.aabb[k] = uni(q.aabb[k]);
            rc.face_bits = __builtin_amdgcn_readfirstlane(q.face_bits);
            rc.depth_margin = uni(q.depth_margin); rc.depth_scale = uni(q.depth_scale); rc.depth_qmax = uni(q.depth_qmax);
        }
    }
    const int V = rc.V, vbase = rc.vbase;
    if ((unsigned)img >= (unsigned)a.n_img || (unsigned)part >= (unsigned)a.n_parts) {      // a hypothesis that names no image / no mesh: NaN row, nothing read
        if (FOOTPRINT) return;
        if (tid < 8) a.out_pose[(size_t)h * 8 + tid] = __builtin_nanf("");
        if (a.out_hist)
            for (int k = tid; k < a.steps; k += blockDim.x) a.out_hist[(size_t)h * a.steps + k] = __builtin_nanf("");
        return;
    }
    const int vg_n = a.table_vg;
    const float* tab = a.table + (size_t)img * ((size_t)((HW + kTabTile - 1) >> kTabShift) * vg_n * kTabTile * 4);
    const float* clu = a.clu + (size_t)img * HW;
    solve_prologue(rc, s);
    TileCtx tc = {};
    if (ONDEMAND) {
        tc.feat = a.feat + (size_t)img * (a.nks * 16) * HW; tc.HW = HW;
        tc.table = a.table + (size_t)img * ((size_t)((HW + kTabTile - 1) >> kTabShift) * vg_n * kTabTile * 4);
        tc.bank = a.bank; tc.apiece = a.apiece; tc.unscale = a.meta->unscale; tc.nks = a.nks;
        tc.vg_n = vg_n; tc.vb_first = vbase >> 5; tc.nvb = ((vbase + V - 1) >> 5) - tc.vb_first + 1;
        tc.nfb = a.nfb;
        build_face_block_masks<FL>(rc, s, tc);
        if (!FOOTPRINT && a.need) preset_table_tiles(rc, s, tc, a.need + (size_t)h * a.npt32 * rc.L.mask_words, a.npt32);
    }

    // Adam's per-step constants (torch single-tensor path: bias corrections and step size in double, then
    // cast): step_size = -lr_k / (1 - beta1^(k+1)), sqrt(1 - beta2^(k+1)); lr_k follows the exponential schedule.
    float2* sched = (float2*)(smem + rc.L.sched);
    for (int k = tid; k < (FOOTPRINT ? 0 : min(a.steps, kSchedSteps)); k += blockDim.x) {
        const double lr_k = a.lr * (a.lr_every > 0 ? pow(a.gamma, (double)(k / a.lr_every)) : 1.0);
        const double bc1 = 1.0 - pow(a.beta1, (double)(k + 1)), bc2 = 1.0 - pow(a.beta2, (double)(k + 1));
        sched[k] = make_float2((float)(-(lr_k / bc1)), (float)sqrt(bc2));
    }
    // optimiser state lives in LDS; the first lanes of wave 0 touch it
    OptState* os = (OptState*)s.opt;
    const int np = a.dof == 6 ? 6 : 4;
    if (tid == 0) {
        s.qcount[0] = (unsigned)H; s.qcount[1] = 0u;       // row range of the covered pixels (see raster_direct)
        s.qcount[2] = kSolveThreads / 64;                  // next face task
        s.qcount[3] = 0u;                                  // length of the front-face list
        os->P[0] = a.init_campos[3 * h]; os->P[1] = a.init_campos[3 * h + 1]; os->P[2] = a.init_campos[3 * h + 2];
        os->P[3] = a.init_theta[h];
        os->P[4] = a.init_principal ? a.init_principal[2 * h] : (WARP ? 0.0f : a.def_px);
        os->P[5] = a.init_principal ? a.init_principal[2 * h + 1] : (WARP ? 0.0f : a.def_py);
        os->first_loss = 0.0f; os->last_loss = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float m = 1e-3f * (rc.aabb[3 + k] - rc.aabb[k]) + 1e-6f;
            os->out_lo[k] = rc.aabb[k] - m; os->out_hi[k] = rc.aabb[3 + k] + m;
        }
    }
    // per-image constants of the loss: sum_p max(0, clu) (what an uncovered image scores) and sum_p clu
    // They are parked in the optimiser state (dedicated LDS words written once by thread 0): s.tot is
    // reused by every later reduction.
    if (!FOOTPRINT) {
        float v2[2] = {0.0f, 0.0f};
        for (int p = tid; p < HW; p += blockDim.x) { float cl = clu[p]; v2[0] += fmaxf(0.0f, cl); v2[1] += cl; }
        block_sum<2>(v2, s.red, s.tot);
        if (tid == 0) {
            os->base_max = s.tot[0]; os->base_clu = s.tot[1];
            os->box_max = s.tot[0]; os->box_clu = s.tot[1]; os->box_n = (float)HW;
        }
        __syncthreads();
    }
    // optional crop for the reported score (rows [br0,br1) x cols [bc0,bc1) of the map)
    int br0 = 0, br1 = H, bc0 = 0, bc1 = W;
    if (!FOOTPRINT && a.score_boxes) {
        br0 = max(a.score_boxes[4 * h], 0); br1 = min(a.score_boxes[4 * h + 1], H);
        bc0 = max(a.score_boxes[4 * h + 2], 0); bc1 = min(a.score_boxes[4 * h + 3], W);
        __syncthreads();
        float v2[2] = {0.0f, 0.0f};
        const int bw = max(bc1 - bc0, 0), bh = max(br1 - br0, 0);
        for (int q = tid; q < bw * bh; q += blockDim.x) {
            float cl = clu[(br0 + q / bw) * W + bc0 + q % bw];
            v2[0] += fmaxf(0.0f, cl); v2[1] += cl;
        }
        block_sum<2>(v2, s.red, s.tot);
        if (tid == 0) { os->box_max = s.tot[0]; os->box_clu = s.tot[1]; os->box_n = (float)(bw * bh); }
        __syncthreads();
    }
    // Optimiser state in registers: lane k of wave 0 owns parameter k (value, first and second moment); the
    // value is mirrored in os->P for the other readers (flow warp, final output).
    float Pk = tid < 6 ? os->P[tid] : 0.0f, mom1 = 0.0f, mom2 = 0.0f;
    const float fHW = (float)HW;
    const unsigned int fmask = (1u << rc.face_bits) - 1u;
    (void)fmask;
    // world vertices of this thread, resident in registers when the mesh is small enough
    const bool verts_resident = V <= kVertsPerThread * (int)blockDim.x;
    float wv[kVertsPerThread][3];
#pragma unroll
    for (int k = 0; k < kVertsPerThread; ++k) {
        const int v = tid + k * blockDim.x;
        const bool ok = verts_resident && v < V;
        wv[k][0] = ok ? rc.verts[3 * v] : 0.0f; wv[k][1] = ok ? rc.verts[3 * v + 1] : 0.0f; wv[k][2] = ok ? rc.verts[3 * v + 2] : 0.0f;
    }
#ifdef NEMO_STAMPS
    long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long stamp_t = clock64();
#endif

    for (int step = 0; step < a.steps; ++step) {
        if (tid < 64) {
            // serial section: seven waves of this hypothesis wait for it, so it goes ahead of the other
            // hypothesis's waves on this SIMD
            __builtin_amdgcn_s_setprio(3);
            float P[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) P[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Pk), k));
            if (tid == 0) {
                pose_setup<false>(rc.cc, P, P[3], WARP ? a.def_px : P[4], WARP ? a.def_py : P[5], s.pc);
                // back faces may be skipped while the camera is outside the solid (outside its bounding box, with a margin)
                bool outside = false;
#pragma unroll
                for (int k = 0; k < 3; ++k) outside = outside | (P[k] < os->out_lo[k]) | (P[k] > os->out_hi[k]);
                s.pc->cull = rc.outward && outside;
                // depth range of the keys: the vertices lie within the bounding radius of the mesh origin, whose view depth is T_z
                const float z0 = s.pc->T[2] - rc.depth_margin;
                s.pc->qz0 = z0;
                s.pc->qc = 8388608.0f - z0 * rc.depth_scale;
            }
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
        STAMP(0);
        if (verts_resident) project_vertices_resident(rc, s, wv); else project_vertices(rc, s);
        STAMP(1);
        // The rasterise phase goes ahead of the other hypothesis's waves on this compute unit (whose other phases wait on LDS and
        // gathers most of the time): two workgroups that start together otherwise move through their phases side by side and
        // compete for the same resource in each.  Launches of one or two full rounds of workgroups take 5 % less time with
        // it (512 hypotheses 9.0 -> 8.5 ms, 1024: 16.7 -> 15.8); launches whose last round is partly empty are unchanged.
        __builtin_amdgcn_s_setprio(1);
        raster_direct<FL>(rc, s);
        __builtin_amdgcn_s_setprio(0);
#ifdef NEMO_STAMPS_P1
        if (tid == 0 && a.out_hist) for (int k = 0; k < 6; ++k) a.out_hist[(size_t)h * a.steps + 16 + k] += s.tot[k] / (float)a.steps;
#endif
        STAMP(7);
#ifdef NEMO_STAMPS_P1
        if (tid == 0 && a.out_hist) { a.out_hist[(size_t)h * a.steps + 11] += (float)s.qcount[0] / (float)a.steps; a.out_hist[(size_t)h * a.steps + 12] += (float)s.qcount[1] / (float)a.steps; }
#endif
        STAMP(2);
        float vals[16];
        const bool last = step == a.steps - 1;
        const int row0 = min((int)s.qcount[0], H), row1 = max(min((int)s.qcount[1], H), row0);
        if (WARP) {
            // Target-centred pass over the whole map: a target pixel blends the scores of its four source
            // pixels in the rendered map; the gradient flows back to every covered source's face and to
            // the flow itself.  (One image at a time and a handful of proposals: simplicity over speed.)
            // Sample positions in torch's own float arithmetic (flow_warp.py + grid_sample, align_corners=True):
            // normalised grid = linspace(-1, 1, n)[i] + (flow / scale) / ((n - 1) / 2), pixel = ((g + 1) / 2) (n - 1).
            // With a zero or whole-pixel flow the position lands within an ulp of a pixel centre, where the
            // bilinear surface has a kink: which one-sided slope the flow gradient sees is decided by this
            // rounding, in the reference as here.
            const float tscale = a.tscale;
            const float fnx = (os->P[4] / tscale) / (((float)W - 1.0f) / 2.0f), fny = (os->P[5] / tscale) / (((float)H - 1.0f) / 2.0f);
            const float stepx = 2.0f / (float)(W - 1), stepy = 2.0f / (float)(H - 1);
            __syncthreads();                    // everyone has read the row range and the flow
            if (tid == 0) { s.qcount[0] = (unsigned)H; s.qcount[1] = 0u; s.qcount[2] = kSolveThreads / 64; s.qcount[3] = 0u; }
            PixelGrad acc;
#pragma unroll
            for (int i = 0; i < 9; ++i) acc.M[i] = 0.f;
            acc.S[0] = acc.S[1] = acc.S[2] = 0.f; acc.gbx = acc.gby = 0.f;
            float dmax = 0.0f, dbox = 0.0f, gfx = 0.0f, gfy = 0.0f;
            const ViewInv vinv = view_inv(rc.cc, s);
            const float cam_a = rc.cc.a;
            for (int pt = tid; pt < HW; pt += blockDim.x) {
                const int yt = pt / W, xt = pt - yt * W;
                const float lx = xt < W / 2 ? -1.0f + stepx * (float)xt : 1.0f - stepx * (float)(W - 1 - xt);
                const float ly = yt < H / 2 ? -1.0f + stepy * (float)yt : 1.0f - stepy * (float)(H - 1 - yt);
                const float fxr = (((lx + fnx) + 1.0f) / 2.0f) * (float)(W - 1), fyr = (((ly + fny) + 1.0f) / 2.0f) * (float)(H - 1);
                const float fx = fminf(fmaxf(fxr, 0.0f), (float)(W - 1)), fy = fminf(fmaxf(fyr, 0.0f), (float)(H - 1));
                const bool x_free = fxr > 0.0f && fxr < (float)(W - 1), y_free = fyr > 0.0f && fyr < (float)(H - 1);
                const float x0f = floorf(fx), y0f = floorf(fy);
                const float ax = fx - x0f, ay = fy - y0f;
                const int x0 = (int)x0f, y0 = (int)y0f, x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
                const int qx[4] = {x0, x1, x0, x1}, qy[4] = {y0, y0, y1, y1};
                const float wq[4] = {(1.0f - ax) * (1.0f - ay), ax * (1.0f - ay), (1.0f - ax) * ay, ax * ay};
                unsigned int key[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) key[k] = s.zkey[qy[k] * W + qx[k]];
                if (key[0] == kEmptyKey32 && key[1] == kEmptyKey32 && key[2] == kEmptyKey32 && key[3] == kEmptyKey32) continue;
                const float cl = clu[pt];
                float sk[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                float dd[4][3];
                int vi[4][3];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (key[k] == kEmptyKey32) continue;    // a source of weight 0 still enters the flow gradient
                    load_face2<FL>(rc, s, (int)(key[k] & fmask), &vi[k][0], &vi[k][1], &vi[k][2]);
#pragma unroll
                    for (int i = 0; i < 3; ++i) dd[k][i] = tab[table_offset(vg_n, vbase + vi[k][i], pt)];
                    sk[k] = pixel_score(s.verts[vi[k][0]], s.verts[vi[k][1]], s.verts[vi[k][2]], s.xs[qx[k]], s.ys[qy[k]], dd[k]);
                }
                const float obj = (wq[0] * sk[0] + wq[1] * sk[1]) + (wq[2] * sk[2] + wq[3] * sk[3]);
                const float dlt = max_nan(obj, cl) - fmaxf(0.0f, cl);
                dmax += dlt;
                if (last && yt >= br0 && yt < br1 && xt >= bc0 && xt < bc1) dbox += dlt;
                const float mk = obj > cl ? 1.0f : (obj == cl ? 0.5f : (__builtin_isunordered(obj, cl) ? obj + cl : 0.0f));
                if (mk == 0.0f) continue;
                const float g = -mk * a.inv_norm;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (key[k] == kEmptyKey32 || wq[k] == 0.0f) continue;
                    (void)pixel_fused(cam_a, vinv.ia, vinv.bx, vinv.by, s.verts[vi[k][0]], s.verts[vi[k][1]], s.verts[vi[k][2]],
                                      s.xs[qx[k]], s.ys[qy[k]], dd[k], -__builtin_inff(), mk * a.inv_norm * wq[k], &acc);
                }
                if (x_free) gfx += g * ((1.0f - ay) * (sk[1] - sk[0]) + ay * (sk[3] - sk[2]));
                if (y_free) gfy += g * ((1.0f - ax) * (sk[2] - sk[0]) + ax * (sk[3] - sk[1]));
            }
#pragma unroll
            for (int i = 0; i < 9; ++i) vals[i] = acc.M[i];
            vals[9] = acc.S[0]; vals[10] = acc.S[1]; vals[11] = acc.S[2];
            vals[12] = gfx * a.inv_tscale; vals[13] = gfy * a.inv_tscale; vals[14] = dmax; vals[15] = dbox;
        } else {
        int first_chunk;
        bool tile_missing = false;
        const int cnt = (ONDEMAND && !FOOTPRINT) ? compact_covered32<true>(rc, s, row0, row1, &first_chunk, tc.nfb, &tile_missing)
                                                 : compact_covered32(rc, s, row0, row1, &first_chunk);
        if (FOOTPRINT) {
            collect_table_tiles(rc, s, tc, first_chunk, cnt, a.need + (size_t)h * a.npt32 * rc.L.mask_words, a.dilate);
            return;
        }
        if (ONDEMAND && __any(tile_missing)) ensure_table_tiles(rc, s, tc, first_chunk, cnt);
#ifdef NEMO_STAMPS_P1
        if ((tid & 63) == 0 && a.out_hist) atomicAdd(&a.out_hist[(size_t)h * a.steps + 13], (float)cnt / (float)a.steps);
#endif
        STAMP(3);

        {
            PixelGrad acc;
#pragma unroll
            for (int i = 0; i < 9; ++i) acc.M[i] = 0.f;
            acc.S[0] = acc.S[1] = acc.S[2] = 0.f; acc.gbx = acc.gby = 0.f;
            float dmax = 0.0f;      // sum over covered pixels of max(obj, clu) - max(0, clu)
            float dbox = 0.0f;      // the same over the score crop (last step only)
            const ViewInv vinv = view_inv(rc.cc, s);
            // The covered entries of this wave, UNR per lane at a time, as straight-line code: the LDS
            // reads, the table / clutter gathers and the arithmetic of the UNR entries interleave.  A lane
            // whose batch runs past the end re-reads its first entry with weight 0.
            constexpr int UNR = 1;
            const float cam_a = rc.cc.a;
            static_assert(UNR == 1, "the entry slots are walked one chunk per trip");
            unsigned int* slot = s.zkey + first_chunk * 64 + (tid & 63);        // entry j of this wave: chunk first_chunk + (j >> 6) * waves
            const int slot_step = 64 * rc.L.waves;
            for (int ib = tid & 63; ib < cnt; ib += 64 * UNR, slot += slot_step) {
                int px[UNR], py[UNR], vi[UNR][3];
                float dd[UNR][3], cc_[UNR], wg[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    wg[u] = 1.0f;
                    int f;
                    unpack_entry(*slot, &px[u], &py[u], &f);
                    *slot = kEmptyKey32;                 // the slot is a key again: empty for the next step
                    load_face2<FL>(rc, s, f, &vi[u][0], &vi[u][1], &vi[u][2]);
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int p = py[u] * W + px[u];
                    dd[u][0] = tab[table_offset(vg_n, vbase + vi[u][0], p)];
                    dd[u][1] = tab[table_offset(vg_n, vbase + vi[u][1], p)];
                    dd[u][2] = tab[table_offset(vg_n, vbase + vi[u][2], p)];
                    cc_[u] = clu[p];
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const float4 v0 = s.verts[vi[u][0]], v1 = s.verts[vi[u][1]], v2 = s.verts[vi[u][2]];
                    const float obj = pixel_fused(cam_a, vinv.ia, vinv.bx, vinv.by, v0, v1, v2, s.xs[px[u]], s.ys[py[u]], dd[u],
                                                  cc_[u], wg[u] * a.inv_norm, &acc);
                    const float dlt = (max_nan(obj, cc_[u]) - fmaxf(0.0f, cc_[u])) * wg[u];
                    dmax += dlt;
                    if (last && py[u] >= br0 && py[u] < br1 && px[u] >= bc0 && px[u] < bc1) dbox += dlt;
                }
            }
#pragma unroll
            for (int i = 0; i < 9; ++i) vals[i] = acc.M[i];
            vals[9] = acc.S[0]; vals[10] = acc.S[1]; vals[11] = acc.S[2];
            vals[12] = acc.gbx; vals[13] = acc.gby; vals[14] = dmax; vals[15] = dbox;
        }
        }
        STAMP(4);
        block_sum16(vals, s.red, s.tot);
        STAMP(5);
        if (WARP)       // the key buffer was not consumed by a compaction: empty the rows that were drawn
            for (int p = row0 * W + tid; p < row1 * W; p += blockDim.x) s.zkey[p] = kEmptyKey32;

        // Wave 0: lane 0 turns the sums into the pose gradient, the six values are handed to lanes 0-5 through
        // readlane, and each of those lanes makes torch.optim.Adam's update (single-tensor path: lerp for the
        // first moment, mul + addcmul for the second, addcdiv) of its own parameter.
        if (tid < 64) {
            __builtin_amdgcn_s_setprio(3);      // serial section again; the priority is dropped after the next pose set-up
            float g[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            if (tid == 0) {
                // every thread has read the row range (a barrier ago): back to "no rows", first face task = number of waves
                if (!WARP) { s.qcount[0] = (unsigned)H; s.qcount[1] = 0u; s.qcount[2] = kSolveThreads / 64; s.qcount[3] = 0u; }
                // the loss value is only reported (first step, last step, history): the update needs the gradient alone
                if (step == 0 || last || a.out_hist) {
                    const float loss = 1.0f - ((os->base_max + s.tot[14]) / fHW - os->base_clu / fHW);
                    if (step == 0) os->first_loss = loss;
                    if (last) os->last_loss = a.score_boxes ? 1.0f - ((os->box_max + s.tot[15]) / os->box_n - os->box_clu / os->box_n) : loss;
                    if (a.out_hist) a.out_hist[(size_t)h * a.steps + step] = loss;
                }
                pose_backward<false>(rc.cc, *s.pc, s.tot, s.tot + 9, s.tot[12], s.tot[13], g, &g[3], &g[4], &g[5]);
                if (WARP) { g[4] = s.tot[12]; g[5] = s.tot[13]; }       // the flow's own gradient
            }
            float gk = 0.0f;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const float t_ = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g[k]), 0));
                gk = tid == k ? t_ : gk;
            }
            if (tid < np) {
                float step_size, bc2s;
                if (step < kSchedSteps) { const float2 sc = sched[step]; step_size = sc.x; bc2s = sc.y; }
                else {
                    const double lr_k = a.lr * (a.lr_every > 0 ? pow(a.gamma, (double)(step / a.lr_every)) : 1.0);
                    step_size = (float)(-(lr_k / (1.0 - pow(a.beta1, (double)(step + 1)))));
                    bc2s = (float)sqrt(1.0 - pow(a.beta2, (double)(step + 1)));
                }
                const float w1 = (float)(1.0 - a.beta1), b2f = (float)a.beta2, w2 = (float)(1.0 - a.beta2), epsf = (float)a.eps;
                const float diff = gk - mom1;
                mom1 = w1 < 0.5f ? mom1 + w1 * diff : gk - diff * (1.0f - w1);
                mom2 = mom2 * b2f + (w2 * gk) * gk;
                const float denom = sqrtf(mom2) / bc2s + epsf;
                Pk = Pk + step_size * (mom1 / denom);
                os->P[tid] = Pk;
            }
        }
        STAMP(6);
        // the next iteration's first barrier orders these LDS accesses before s.tot / s.pc are reused
    }
    if (tid == 0) {
        float* o = a.out_pose + (size_t)h * 8;
        for (int k = 0; k < 6; ++k) o[k] = os->P[k];
        o[6] = os->last_loss; o[7] = os->first_loss;

#ifdef NEMO_STAMPS
        if (a.out_hist) for (int k = 0; k < 8; ++k) a.out_hist[(size_t)h * a.steps + k] = (float)stamp_acc[k] / (float)a.steps;
#endif
    }
}

// The table tiles the first step of every hypothesis reads (listed by the FOOTPRINT launch), computed in bulk in front
// of the loop: a wave per (hypothesis, 32-pixel tile) with a non-empty row of the list, the same routine and operand
// values as the tiles the loop computes for itself (raster.h::table_tiles_nks), with registers to spare for all the
// channel steps of a bank fragment in flight and thousands of waves to hide the reads of the feature columns behind.
constexpr int kBulkWaves = 4;
// The list also takes the tiles of the pixels this many columns / rows around every covered one: the pose moves by a
// fraction of a degree per step, and a tile computed here costs a third of one computed inside the loop, where seven
// waves wait for the eighth.  Measured on 512 hypotheses of the 80 x 100 x 128 scene (loop + bulk kernel + the launch
// that looks, ms): no ring 8.51 + 0.59 + 0.06, 2 pixels 8.18 + 0.75 + 0.10, 3 pixels 8.08 + 0.83 + 0.11, 4 pixels
// 8.05 + 0.91 + 0.11; without any of it the loop takes 9.60.
constexpr int kBulkRing = 2;
template <int NKS>
__global__ __launch_bounds__(64 * kBulkWaves) void tile_bulk_kernel(SolveArgs a, int mw) {
    // one-dimensional grid, a hypothesis's groups of kBulkWaves pixel tiles side by side (neighbouring workgroups read neighbouring columns)
    const int groups = (a.npt32 + kBulkWaves - 1) / kBulkWaves;
    const int h = blockIdx.x / groups, pt = (blockIdx.x % groups) * kBulkWaves + (threadIdx.x >> 6);
    if (pt >= a.npt32) return;
    const unsigned int* nm = a.need + ((size_t)h * a.npt32 + pt) * mw;
    unsigned int any = 0u;
    for (int w = 0; w < mw; ++w) any |= nm[w];
    if (!__builtin_amdgcn_readfirstlane((int)any)) return;
    const int HW = a.rc.cc.H * a.rc.cc.W;
    const int img = a.hyp_to_image ? a.hyp_to_image[h] : h;
    int v0 = 0, nv = a.rc.V;
    if (a.hyp_to_part) {
        const int part = a.hyp_to_part[h];
        if ((unsigned)part >= (unsigned)a.n_parts) return;
        v0 = a.parts[part].v0; nv = a.parts[part].nv;
    }
    if ((unsigned)img >= (unsigned)a.n_img) return;
    const int vg_n = a.table_vg, vb_first = v0 >> 5, nvb = ((v0 + nv - 1) >> 5) - vb_first + 1;
    const float* feat = a.feat + (size_t)img * (NKS * 16) * HW;
    float* table = a.table + (size_t)img * ((size_t)((HW + kTabTile - 1) >> kTabShift) * vg_n * kTabTile * 4);
    const int lane = threadIdx.x & 63, lk = lane >> 5, px = lane & 31;
    const int p = pt * 32 + px;
    const bool pok = p < HW;
    const float* col = feat + (pok ? p : HW - 1) + (size_t)(lk * 8) * HW;
    const uint4* ap = a.bank + (size_t)vb_first * NKS * 64 + lane;
    const int g0 = vb_first * 8 + lk;
    float4* out = reinterpret_cast<float4*>(table) + ((size_t)(pt >> 2) * vg_n + g0) * 128 + (pt & 3) * 32 + px;
    table_tiles_nks<NKS, (NKS < 4 ? NKS : 4)>(col, HW, pok, ap, a.apiece, out, a.meta->unscale, g0, vg_n, nvb, nm);
}

// Diagnostic: the rasterise phase of the optimisation loop in its two forms on the same poses, timed in shader cycles.
// form 0: raster_direct (face per lane, bounding-box walk, LDS atomic min); form 1: raster_tiled (faces binned to 8 x 8
// tiles, a wave per tile, lanes = pixels).  One workgroup per pose, `reps` rasterisations each; out_cycles[pose] = average
// cycles of one (for form 1, out_cycles[n + pose] = the binning part of it); out_keys [n, HW] = the key buffer of the last.
struct RasterFormsArgs {
    RasterCtx rc; const float* campos; const float* theta; float def_px, def_py;
    int form, reps, bins_off, bins_cap;
    float* out_cycles; unsigned int* out_keys; int n;
};
template <bool FL>
__global__ __launch_bounds__(kSolveThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void raster_forms_kernel(RasterFormsArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Lds s = carve(smem, a.rc.L);
    const RasterCtx& rc = a.rc;
    const int h = blockIdx.x, tid = threadIdx.x, H = rc.cc.H, W = rc.cc.W, HW = H * W;
    solve_prologue(rc, s);
    const int TX = (W + 7) >> kBinShift, TY = (H + 7) >> kBinShift, NT = TX * TY;
    TileBins tb;
    tb.cnt = (unsigned int*)(smem + a.bins_off); tb.cur = tb.cnt + NT + 1; tb.ent = (unsigned short*)(tb.cur + NT); tb.cap = a.bins_cap;
    if (tid == 0) {
        float P[3] = {a.campos[3 * h], a.campos[3 * h + 1], a.campos[3 * h + 2]};
        pose_setup<false>(rc.cc, P, a.theta[h], a.def_px, a.def_py, s.pc);
        bool outside = false;
        for (int k = 0; k < 3; ++k) {
            const float m = 1e-3f * (rc.aabb[3 + k] - rc.aabb[k]) + 1e-6f;
            outside = outside | (P[k] < rc.aabb[k] - m) | (P[k] > rc.aabb[3 + k] + m);
        }
        s.pc->cull = rc.outward && outside;
        const float z0 = s.pc->T[2] - rc.depth_margin;
        s.pc->qz0 = z0; s.pc->qc = 8388608.0f - z0 * rc.depth_scale;
        s.qcount[0] = (unsigned)H; s.qcount[1] = 0u; s.qcount[2] = kSolveThreads / 64; s.qcount[3] = 0u;
    }
    __syncthreads();
    project_vertices(rc, s);
    long long total = 0, binning = 0;
    for (int rep = 0; rep < a.reps; ++rep) {
        const long long t0 = clock64();
        long long tb_end = t0;
        if (a.form == 0) raster_direct<FL>(rc, s); else raster_tiled<FL>(rc, s, tb, &tb_end);
        const long long t1 = clock64();
        total += t1 - t0; binning += tb_end - t0;
        if (rep == a.reps - 1 && a.out_keys)
            for (int p = tid; p < HW; p += blockDim.x) a.out_keys[(size_t)h * HW + p] = s.zkey[p];
        __syncthreads();
        for (int p = tid; p < HW; p += blockDim.x) s.zkey[p] = kEmptyKey32;
        if (tid == 0) { s.qcount[0] = (unsigned)H; s.qcount[1] = 0u; s.qcount[2] = kSolveThreads / 64; s.qcount[3] = 0u; }
        __syncthreads();
    }
    if (tid == 0) {
        a.out_cycles[h] = (float)total / (float)a.reps;
        a.out_cycles[a.n + h] = (float)binning / (float)a.reps;
    }
}

extern "C" int nemo_raster_forms(nemo_mesh_t m, const nemo_camera* cam, const float* campos, const float* theta, int32_t n,
                                 int32_t form, int32_t reps, float* out_cycles, uint32_t* out_keys, void* stream) {
    if (!m || !campos || !theta || !out_cycles || n <= 0 || reps <= 0 || (form != 0 && form != 1))
        return fail(NEMO_ERR_INVALID, "nemo_raster_forms: bad argument");
    if (check_cam(cam)) return NEMO_ERR_INVALID;
    if (m->F >= (1 << 16)) return fail(NEMO_ERR_UNSUPPORTED, "nemo_raster_forms: meshes below 65536 faces");
    RasterFormsArgs a;
    a.rc.verts = m->verts; a.rc.faces = m->faces_outward ? m->faces_outward : m->faces; a.rc.V = m->V; a.rc.F = m->F;
    a.rc.outward = m->faces_outward != nullptr && m->whole_outward;
    for (int k = 0; k < 6; ++k) a.rc.aabb[k] = m->aabb[k];
    a.rc.cc = make_cam_const(cam->H, cam->W, cam->focal);
    int rcode = plan_solve(cam->H, cam->W, m->V, m->F, kSolveThreads, m->aabb, &a.rc);
    if (rcode) return rcode;
    const int NT = ((cam->W + 7) >> nemo::kBinShift) * ((cam->H + 7) >> nemo::kBinShift);
    a.bins_off = a.rc.L.total;
    a.bins_cap = 4096;
    const int lds = a.bins_off + 4 * (2 * NT + 1) + 2 * a.bins_cap;
    if (lds > kMaxLds) return fail(NEMO_ERR_UNSUPPORTED, "nemo_raster_forms: %d bytes of LDS", lds);
    a.campos = campos; a.theta = theta; a.def_px = cam->px; a.def_py = cam->py;
    a.form = form; a.reps = reps; a.out_cycles = out_cycles; a.out_keys = out_keys; a.n = n;
    auto kern = a.rc.L.faces >= 0 ? raster_forms_kernel<true> : raster_forms_kernel<false>;
    if (set_lds_limit(kern, lds)) return NEMO_ERR_HIP;
    hipLaunchKernelGGL(kern, dim3(n), dim3(kSolveThreads), lds, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return NEMO_OK;
}

// The sizes the optimisation loop's LDS plan is made for: the whole handle, or its largest part.
static void solve_plan_sizes(const nemo_mesh_s* m, bool use_parts, int* lay_v, int* lay_f, int* lay_blocks) {
    *lay_v = m->V; *lay_f = m->F; *lay_blocks = ((m->V - 1) >> 5) + 1;
    if (!use_parts) return;
    *lay_v = *lay_f = *lay_blocks = 0;
    for (const nemo::MeshPart& q : m->parts) {
        *lay_v = q.nv > *lay_v ? q.nv : *lay_v; *lay_f = q.nf > *lay_f ? q.nf : *lay_f;
        const int nb = ((q.v0 + q.nv - 1) >> 5) - (q.v0 >> 5) + 1;
        *lay_blocks = nb > *lay_blocks ? nb : *lay_blocks;
    }
}

// `feat` non-null: the similarity table is computed on demand by the kernel (`table` = workspace that need not be
// initialised); feat null: `table` is the complete table of nemo_similarity_table.
static int solve_impl(nemo_mesh_t m, const nemo_camera* cam, const float* feat, float* table, const float* clu,
                      const int32_t* hyp_to_image, const float* init_campos, const float* init_theta,
                      const float* init_principal, const int32_t* score_boxes, int32_t n_hyp, int32_t n_img,
                      int32_t global_batch, int32_t dof, const nemo_adam* opt, float* out_pose, float* out_hist,
                      const int32_t* hyp_to_part, void* stream, size_t workspace_bytes = 0) {
    if (n_hyp == 0) return NEMO_OK;
    if (!m || !table || !clu || !init_campos || !init_theta || !opt || !out_pose || n_hyp < 0 || n_img <= 0 || global_batch <= 0)
        return fail(NEMO_ERR_INVALID, "nemo_solve: bad argument");
    if (check_cam(cam)) return NEMO_ERR_INVALID;
    if (dof != 3 && dof != 4 && dof != 6) return fail(NEMO_ERR_INVALID, "nemo_solve: dof must be 3, 4 or 6");
    if (opt->steps < 0) return fail(NEMO_ERR_INVALID, "nemo_solve: negative step count");
    SolveArgs a;
    a.rc.verts = m->verts; a.rc.faces = m->faces_outward ? m->faces_outward : m->faces; a.rc.V = m->V; a.rc.F = m->F;
    a.rc.outward = m->faces_outward != nullptr && m->whole_outward;
    for (int k = 0; k < 6; ++k) a.rc.aabb[k] = m->aabb[k];
    a.rc.cc = make_cam_const(cam->H, cam->W, cam->focal);
    // LDS plan: for the whole handle, or for its largest part when every hypothesis renders one part
    int lay_v, lay_f, lay_blocks;
    a.parts = nullptr; a.hyp_to_part = nullptr; a.n_parts = 1;
    if (hyp_to_part) {
        if (m->parts.empty()) return fail(NEMO_ERR_STATE, "nemo_solve_parts: the mesh has no parts (call nemo_mesh_set_parts)");
        a.parts = m->d_parts; a.hyp_to_part = hyp_to_part; a.n_parts = (int)m->parts.size();
    }
    solve_plan_sizes(m, hyp_to_part != nullptr, &lay_v, &lay_f, &lay_blocks);
    const int HW = cam->H * cam->W;
    a.feat = feat; a.bank = nullptr; a.apiece = 0; a.meta = nullptr; a.nks = 0;
    a.nfb = (lay_f + (1 << kFaceBlockShift) - 1) >> kFaceBlockShift;
    int vblocks = 0;
    if (feat) {
        if (!m->bank_frag)
            return fail(NEMO_ERR_UNSUPPORTED, "nemo_solve_features: %d channels (the on-demand table takes 16, 32, 64 or 128: use "
                        "nemo_similarity_table + nemo_solve)", m->C);
        // (pixel tile, face block) pairs travel as two 16-bit halves of a word
        if (((HW + 31) >> 5) >= 0xFFFF || a.nfb > 0xFFFF) return fail(NEMO_ERR_UNSUPPORTED, "nemo_solve_features: map or mesh too large");
        vblocks = lay_blocks;
        a.meta = (const BankMeta*)m->bank_frag;
        a.bank = (const uint4*)((const char*)m->bank_frag + SG_META_BYTES);
        a.nks = m->C / 16;
        a.apiece = (size_t)nemo::split_bank_blocks(m->V) * a.nks * 64;
    }
    int rcode = plan_solve(cam->H, cam->W, lay_v, lay_f, kSolveThreads, m->aabb, &a.rc, vblocks);
    if (rcode) return rcode;
    a.table_vg = (m->V + 3) >> 2;
    a.table = table; a.clu = clu; a.hyp_to_image = hyp_to_image;
    a.init_campos = init_campos; a.init_theta = init_theta; a.init_principal = init_principal;
    a.score_boxes = score_boxes;
    a.def_px = cam->px; a.def_py = cam->py;
    a.n_img = n_img;
    a.inv_norm = 1.0f / ((float)global_batch * (float)(cam->H * cam->W));
    a.dof = dof;
    a.steps = opt->steps; a.lr_every = opt->lr_every;
    a.lr = opt->lr; a.beta1 = opt->beta1; a.beta2 = opt->beta2; a.eps = opt->eps; a.gamma = opt->lr_gamma;
    a.out_pose = out_pose; a.out_hist = out_hist;
    a.tscale = 1.0f; a.inv_tscale = 0.0f;
    a.need = nullptr; a.npt32 = (HW + 31) >> 5; a.dilate = 0;
    if (feat && opt->steps > 0) {
        // First step's tiles in bulk, when the workspace has room for the lists behind the tables: a launch that only looks
        // (first step up to the covered entries) lists them, one kernel computes them, the loop starts with them present.
        const size_t table_bytes = align256((size_t)n_img * table_image_elems(m->V, HW) * sizeof(float));
        const size_t need_bytes = (size_t)n_hyp * a.npt32 * a.rc.L.mask_words * sizeof(unsigned int);
        if (workspace_bytes >= table_bytes + need_bytes) {
            a.need = (unsigned int*)((char*)table + table_bytes);
            a.dilate = kBulkRing;
            HIP_TRY(hipMemsetAsync(a.need, 0, need_bytes, (hipStream_t)stream));
            auto look = a.rc.L.faces >= 0 ? solve_kernel<true, false, true, true> : solve_kernel<false, false, true, true>;
            if (set_lds_limit(look, a.rc.L.total)) return NEMO_ERR_HIP;
            hipLaunchKernelGGL(look, dim3(n_hyp), dim3(kSolveThreads), a.rc.L.total, (hipStream_t)stream, a);
            HIP_TRY(hipGetLastError());
            const long long nblk = (long long)n_hyp * ((a.npt32 + kBulkWaves - 1) / kBulkWaves);
            if (nblk > 0x7FFFFFFFLL) return fail(NEMO_ERR_UNSUPPORTED, "nemo_solve_features: too many hypotheses for one call");
            const dim3 bg((unsigned)nblk);
            const int mw = a.rc.L.mask_words;
            switch (a.nks) {
                case 1: hipLaunchKernelGGL(tile_bulk_kernel<1>, bg, dim3(64 * kBulkWaves), 0, (hipStream_t)stream, a, mw); break;
                case 2: hipLaunchKernelGGL(tile_bulk_kernel<2>, bg, dim3(64 * kBulkWaves), 0, (hipStream_t)stream, a, mw); break;
                case 4: hipLaunchKernelGGL(tile_bulk_kernel<4>, bg, dim3(64 * kBulkWaves), 0, (hipStream_t)stream, a, mw); break;
                default: hipLaunchKernelGGL(tile_bulk_kernel<8>, bg, dim3(64 * kBulkWaves), 0, (hipStream_t)stream, a, mw); break;
            }
            HIP_TRY(hipGetLastError());
        }
    }
    auto kern = feat ? (a.rc.L.faces >= 0 ? solve_kernel<true, false, true> : solve_kernel<false, false, true>)
                     : (a.rc.L.faces >= 0 ? solve_kernel<true, false> : solve_kernel<false, false>);
    if (set_lds_limit(kern, a.rc.L.total)) return NEMO_ERR_HIP;
    hipLaunchKernelGGL(kern, dim3(n_hyp), dim3(kSolveThreads), a.rc.L.total, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return NEMO_OK;
}

extern "C" int nemo_solve(nemo_mesh_t m, const nemo_camera* cam, const float* table, const float* clu,
                          const int32_t* hyp_to_image, const float* init_campos, const float* init_theta,
                          const float* init_principal, const int32_t* score_boxes, int32_t n_hyp, int32_t n_img,
                          int32_t global_batch, int32_t dof, const nemo_adam* opt, float* out_pose, float* out_hist,
                          void* stream) {
    return solve_impl(m, cam, nullptr, const_cast<float*>(table), clu, hyp_to_image, init_campos, init_theta, init_principal, score_boxes,
                      n_hyp, n_img, global_batch, dof, opt, out_pose, out_hist, nullptr, stream);
}

extern "C" int nemo_solve_parts(nemo_mesh_t m, const nemo_camera* cam, const float* table, const float* clu,
                                const int32_t* hyp_to_image, const int32_t* hyp_to_part, const float* init_campos,
                                const float* init_theta, const float* init_principal, const int32_t* score_boxes, int32_t n_hyp,
                                int32_t n_img, int32_t global_batch, int32_t dof, const nemo_adam* opt, float* out_pose,
                                float* out_hist, void* stream) {
    if (!hyp_to_part) return fail(NEMO_ERR_INVALID, "nemo_solve_parts: hyp_to_part is required (nemo_solve renders the whole handle)");
    return solve_impl(m, cam, nullptr, const_cast<float*>(table), clu, hyp_to_image, init_campos, init_theta, init_principal, score_boxes,
                      n_hyp, n_img, global_batch, dof, opt, out_pose, out_hist, hyp_to_part, stream);
}

extern "C" int nemo_solve_features_supported(nemo_mesh_t m, int32_t H, int32_t W, int32_t with_parts) {
    if (!m || H <= 0 || W <= 0 || !m->bank_frag || (with_parts && m->parts.empty())) return 0;
    if (H > kMaxSide || W > kMaxSide || m->F >= kMaxFaces) return 0;
    int lay_v, lay_f, lay_blocks;
    solve_plan_sizes(m, with_parts != 0, &lay_v, &lay_f, &lay_blocks);
    const int nfb = (lay_f + (1 << kFaceBlockShift) - 1) >> kFaceBlockShift;
    if (((H * W + 31) >> 5) >= 0xFFFF || nfb > 0xFFFF) return 0;
    const bool packable = lay_v <= kPackedFaceVerts;
    LdsLayout L = make_solve_layout(H, W, lay_v, lay_f, packable, kSolveThreads / 64, lay_blocks);
    if (!packable || L.total > kMaxLds) L = make_solve_layout(H, W, lay_v, lay_f, false, kSolveThreads / 64, lay_blocks);
    return L.total <= kMaxLds ? 1 : 0;
}

extern "C" size_t nemo_solve_features_workspace(nemo_mesh_t m, int32_t n_img, int32_t HW) {
    if (!m || n_img <= 0 || HW <= 0 || !m->bank_frag) return 0;
    // the tables, and behind them the lists of the first step's tiles for up to two hypotheses per image
    const size_t mw = (size_t)((((m->V - 1) >> 5) + 1) + 31) / 32;
    return align256((size_t)n_img * table_image_elems(m->V, HW) * sizeof(float)) +
           align256((size_t)2 * n_img * ((HW + 31) >> 5) * mw * sizeof(unsigned int));
}

extern "C" int nemo_solve_features(nemo_mesh_t m, const nemo_camera* cam, const float* feat, void* workspace, size_t workspace_bytes,
                                   const float* clu, const int32_t* hyp_to_image, const int32_t* hyp_to_part,
                                   const float* init_campos, const float* init_theta, const float* init_principal,
                                   const int32_t* score_boxes, int32_t n_hyp, int32_t n_img, int32_t global_batch, int32_t dof,
                                   const nemo_adam* opt, float* out_pose, float* out_hist, void* stream) {
    if (!feat || !workspace) return fail(NEMO_ERR_INVALID, "nemo_solve_features: feat and workspace are required");
    // at least the tables; with the room nemo_solve_features_workspace names behind them, the first step's tiles are computed in bulk
    if (m && m->bank_frag && n_img > 0 && cam) {
        const size_t tables = align256((size_t)n_img * table_image_elems(m->V, cam->H * cam->W) * sizeof(float));
        if (workspace_bytes < tables)
            return fail(NEMO_ERR_INVALID, "nemo_solve_features: workspace of at least %zu bytes required (%zu with the lists of the first "
                        "step's tiles), %zu given", tables, nemo_solve_features_workspace(m, n_img, cam->H * cam->W), workspace_bytes);
    }
    return solve_impl(m, cam, feat, (float*)workspace, clu, hyp_to_image, init_campos, init_theta, init_principal, score_boxes, n_hyp, n_img,
                      global_batch, dof, opt, out_pose, out_hist, hyp_to_part, stream, workspace_bytes);
}

// Refinement of the keypoint-initialised solver (solve_pose.py:220-324): every proposal of an image is one
// hypothesis; parameters C, theta and, when optimize_translation is set, the 2-D flow.
extern "C" int nemo_solve_flow(nemo_mesh_t m, const nemo_camera* cam, const float* table, const float* clu,
                               const int32_t* hyp_to_image, const float* init_campos, const float* init_theta,
                               const float* init_flow, int32_t n_hyp, int32_t n_img, int32_t global_batch,
                               int32_t optimize_translation, float translation_scale, const nemo_adam* opt,
                               float* out_pose, float* out_hist, void* stream) {
    if (n_hyp == 0) return NEMO_OK;
    if (!m || !table || !clu || !init_campos || !init_theta || !opt || !out_pose || n_hyp < 0 || n_img <= 0 || global_batch <= 0)
        return fail(NEMO_ERR_INVALID, "nemo_solve_flow: bad argument");
    if (check_cam(cam)) return NEMO_ERR_INVALID;
    if (!(translation_scale > 0.0f)) return fail(NEMO_ERR_INVALID, "nemo_solve_flow: translation_scale must be positive");
    if (opt->steps < 0) return fail(NEMO_ERR_INVALID, "nemo_solve_flow: negative step count");
    SolveArgs a;
    a.rc.verts = m->verts; a.rc.faces = m->faces_outward ? m->faces_outward : m->faces; a.rc.V = m->V; a.rc.F = m->F;
    a.rc.outward = m->faces_outward != nullptr && m->whole_outward;
    for (int k = 0; k < 6; ++k) a.rc.aabb[k] = m->aabb[k];
    a.rc.cc = make_cam_const(cam->H, cam->W, cam->focal);
    int rcode = plan_solve(cam->H, cam->W, m->V, m->F, kSolveThreads, m->aabb, &a.rc);
    if (rcode) return rcode;
    a.table = const_cast<float*>(table); a.clu = clu; a.hyp_to_image = hyp_to_image;
    a.feat = nullptr; a.bank = nullptr; a.apiece = 0; a.meta = nullptr; a.nks = 0; a.nfb = 0;
    a.parts = nullptr; a.hyp_to_part = nullptr; a.n_parts = 1; a.table_vg = (m->V + 3) >> 2;
    a.init_campos = init_campos; a.init_theta = init_theta; a.init_principal = init_flow;
    a.score_boxes = nullptr;
    a.def_px = cam->px; a.def_py = cam->py;
    a.tscale = translation_scale; a.inv_tscale = 1.0f / translation_scale;
    a.n_img = n_img;
    a.inv_norm = 1.0f / ((float)global_batch * (float)(cam->H * cam->W));
    a.dof = optimize_translation ? 6 : 4;
    a.steps = opt->steps; a.lr_every = opt->lr_every;
    a.lr = opt->lr; a.beta1 = opt->beta1; a.beta2 = opt->beta2; a.eps = opt->eps; a.gamma = opt->lr_gamma;
    a.out_pose = out_pose; a.out_hist = out_hist;
    auto kern = a.rc.L.faces >= 0 ? solve_kernel<true, true> : solve_kernel<false, true>;
    if (set_lds_limit(kern, a.rc.L.total)) return NEMO_ERR_HIP;
    hipLaunchKernelGGL(kern, dim3(n_hyp), dim3(kSolveThreads), a.rc.L.total, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return NEMO_OK;
}

// ------------------------------------------------------------------------------------------
// keypoint correlation (get_corr_pytorch, solve_pose.py:27-82)
// ------------------------------------------------------------------------------------------
// corr[i, a, b] = sum_v vis[i, v] * table[v, pixel(round((kp[i, v] + (px[a], py[b])) / dsr))], keypoints that
// land outside [0, W-1) x [0, H-1) do not count.  One workgroup per (pose i, a, b).
__global__ __launch_bounds__(256) void keypoint_corr_kernel(const float* __restrict__ table, const float* __restrict__ kp,
                                                            const float* __restrict__ vis, const double* __restrict__ pxs,
                                                            const double* __restrict__ pys, int V, int npx, int npy,
                                                            float dsr, int H, int W, float* __restrict__ out) {
    __shared__ float red[4];
    const int b = blockIdx.x % npy, a_ = (blockIdx.x / npy) % npx, i = blockIdx.x / (npx * npy);
    const double ox = pxs[a_], oy = pys[b];
    const int vg_n = (V + 3) >> 2;
    float sum = 0.0f;
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
        // the reference adds the f64 sample to the f32 coordinate in place: f64 sum rounded to f32
        const float x = (float)((double)kp[((size_t)i * V + v) * 2] + ox), y = (float)((double)kp[((size_t)i * V + v) * 2 + 1] + oy);
        const float cx = rintf(x / dsr), cy = rintf(y / dsr);
        const bool in = !(cx < 0.0f) && !(cx >= (float)(W - 1)) && !(cy < 0.0f) && !(cy >= (float)(H - 1)) && cx == cx && cy == cy;
        if (in) sum += vis[(size_t)i * V + v] * table[table_offset(vg_n, v, (int)cy * W + (int)cx)];
    }
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

extern "C" int nemo_keypoint_corr(nemo_mesh_t m, const float* table, const float* kp_coords, const float* kp_vis,
                                  const double* px_samples, const double* py_samples, int32_t n_pose, int32_t n_px,
                                  int32_t n_py, float down_sample_rate, int32_t H, int32_t W, float* out, void* stream) {
    if (n_pose == 0 || n_px == 0 || n_py == 0) return NEMO_OK;
    if (!m || !table || !kp_coords || !kp_vis || !px_samples || !py_samples || !out || n_pose < 0 || n_px < 0 || n_py < 0 ||
        H <= 1 || W <= 1 || !(down_sample_rate > 0.0f))
        return fail(NEMO_ERR_INVALID, "nemo_keypoint_corr: bad argument");
    const long long nwg = (long long)n_pose * n_px * n_py;
    if (nwg > 0x7FFFFFFFLL) return fail(NEMO_ERR_UNSUPPORTED, "nemo_keypoint_corr: too many samples for one launch");
    hipLaunchKernelGGL(keypoint_corr_kernel, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, table, kp_coords, kp_vis,
                       px_samples, py_samples, m->V, n_px, n_py, down_sample_rate, H, W, out);
    HIP_TRY(hipGetLastError());
    return NEMO_OK;
}

// ------------------------------------------------------------------------------------------
// spherical read-out
// ------------------------------------------------------------------------------------------
__global__ void spherical_kernel(const float* __restrict__ campos, int n, float* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = campos[3 * i], y = campos[3 * i + 1], z = campos[3 * i + 2];
    float d = sqrtf((x * x + y * y) + z * z);
    const double pi = 3.14159265358979323846;
    // atan(x/z) % pi + pi * (x < 0), python-style remainder (result in [0, pi))
    float at = (float)atan((double)(x / z));
    float r = at - floorf(at / (float)pi) * (float)pi;
    if (r < 0.0f) r += (float)pi;
    if (r >= (float)pi) r -= (float)pi;
    float azim = r + (x < 0.0f ? (float)pi : 0.0f);
    float elev = (float)asin((double)(y / d));
    out[3 * i] = d; out[3 * i + 1] = elev; out[3 * i + 2] = azim;
}

extern "C" int nemo_spherical_angles(const float* campos, int32_t n, float* out, void* stream) {
    if (n == 0) return NEMO_OK;
    if (!campos || !out || n < 0) return fail(NEMO_ERR_INVALID, "nemo_spherical_angles: bad argument");
    hipLaunchKernelGGL(spherical_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, campos, n, out);
    HIP_TRY(hipGetLastError());
    return NEMO_OK;
}

// ------------------------------------------------------------------------------------------
// similarity table (kernel in table_gemm.hip)
// ------------------------------------------------------------------------------------------
namespace nemo {
int launch_table_gemm(const float* memT, const float* feat, float* table, int V, int C, int HW, int B, hipStream_t st);
}

extern "C" size_t nemo_similarity_table_elems(nemo_mesh_t m, int32_t HW) {
    if (!m || HW <= 0) return 0;
    return table_image_elems(m->V, HW);
}

extern "C" int nemo_similarity_table(nemo_mesh_t m, const float* feat, int32_t B, int32_t HW, float* table, void* stream) {
    if (B == 0) return NEMO_OK;
    if (!m || !feat || !table || B < 0 || HW <= 0) return fail(NEMO_ERR_INVALID, "nemo_similarity_table: bad argument");
    if (!m->memory_t) return fail(NEMO_ERR_STATE, "nemo_similarity_table: mesh has no vertex memory");
    const int rc = m->bank_frag ? nemo::launch_split_table(m->bank_frag, feat, table, m->V, m->C, HW, B, (hipStream_t)stream)
                                : nemo::launch_table_gemm(m->memory_t, feat, table, m->V, m->C, HW, B, (hipStream_t)stream);
    if (rc)
        return fail(NEMO_ERR_UNSUPPORTED, "nemo_similarity_table: too many tiles for one launch");
    HIP_TRY(hipGetLastError());
    return NEMO_OK;
}

// ------------------------------------------------------------------------------------------
// training-side keypoint annotation: projected vertices + visibility (project_kp.py:175-247)
// ------------------------------------------------------------------------------------------
// get_one_standard for one mesh per pose: every vertex is projected to feature-map pixels (y, x), the mesh is
// rasterised, the "distance" map (per-vertex distance to cam_loc, interpolated with the barycentrics) is read
// at the pixel nearest to the projected vertex, and the vertex is visible when that value equals its own
// distance within dist_thr.  Here the map never exists: after the bit-exact rasteriser has filled the key
// buffer in LDS, each vertex looks up its one pixel, recomputes that pixel's barycentrics and interpolates the
// three distances.  cam_loc follows the reference's formula -inverse(R) T (with the orthonormal R, -R^T T);
// it is the camera centre only for symmetric R, and the test is self-consistent either way because map and
// vertex use the same point.
struct KeypointVisArgs {
    RasterCtx rc;
    const float* campos; const float* theta; const float* principal;
    const float* rt;             // optional [n, 12]: R (row-major, X_view = X R + T) and T given by the caller instead of campos / theta
    float def_px, def_py, focal, dist_thr;
    int clamp_to_image;
    float* out_kp; uint8_t* out_vis;
};

__device__ inline float kp_distance(const float* cl, float x, float y, float z) {
    const float dx = cl[0] - x, dy = cl[1] - y, dz = cl[2] - z;
    return sqrtf((dx * dx + dy * dy) + dz * dz);
}

template <bool FL>
__global__ __launch_bounds__(kRasterThreads) void keypoint_vis_kernel(KeypointVisArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Lds s = carve(smem, a.rc.L);
    const int n = blockIdx.x;
    const int H = a.rc.cc.H, W = a.rc.cc.W, V = a.rc.V;
    raster_prologue(a.rc, s);
    const float px = a.principal ? a.principal[2 * n] : a.def_px, py = a.principal ? a.principal[2 * n + 1] : a.def_py;
    if (threadIdx.x == 0) {
        if (a.rt) {
            for (int i = 0; i < 9; ++i) s.pc->R[i] = a.rt[12 * (size_t)n + i];
            for (int i = 0; i < 3; ++i) s.pc->T[i] = a.rt[12 * (size_t)n + 9 + i];
            s.pc->bx = (a.rc.cc.half_w - px) / a.rc.cc.s; s.pc->by = (a.rc.cc.half_h - py) / a.rc.cc.s;
        } else {
            float C[3] = {a.campos[3 * n], a.campos[3 * n + 1], a.campos[3 * n + 2]};
            pose_setup(a.rc.cc, C, a.theta[n], px, py, s.pc);
        }
    }
    __syncthreads();
    project_vertices(a.rc, s);
    raster_faces<FL, true>(a.rc, s);
    float R[9], T[3], cl[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = s.pc->R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) T[i] = s.pc->T[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) cl[i] = -((R[i] * T[0] + R[3 + i] * T[1]) + R[6 + i] * T[2]);      // -(R^T T)_i
    const float fW = (float)W, fH = (float)H;
    for (int v = threadIdx.x; v < V; v += blockDim.x) {
        const float X0 = a.rc.verts[3 * v], X1 = a.rc.verts[3 * v + 1], X2 = a.rc.verts[3 * v + 2];
        float Xv[3], xn, yn;
        project_vertex(a.rc.cc, R, T, s.pc->bx, s.pc->by, X0, X1, X2, Xv, &xn, &yn);
        // screen point of the pinhole (f X + p Z) / Z, mirrored about the principal point (project_kp.py:186-188)
        float x = 2.0f * px - (a.focal * Xv[0] + px * Xv[2]) / Xv[2];
        float y = 2.0f * py - (a.focal * Xv[1] + py * Xv[2]) / Xv[2];
        if (a.clamp_to_image) {
            x = fmaxf(fminf(x, fW - 1.0f), 0.0f);
            y = fmaxf(fminf(y, fH - 1.0f), 0.0f);
        }
        a.out_kp[((size_t)n * V + v) * 2] = y;
        a.out_kp[((size_t)n * V + v) * 2 + 1] = x;
        // grid_sample(mode='nearest', align_corners=False) at grid = p / size * 2 - 1: pixel index rint(((g + 1) size - 1) / 2)
        const float gx = x / fW * 2.0f - 1.0f, gy = y / fH * 2.0f - 1.0f;
        const float fx = rintf(((gx + 1.0f) * fW - 1.0f) / 2.0f), fy = rintf(((gy + 1.0f) * fH - 1.0f) / 2.0f);
        float sampled = 0.0f;                                   // zero padding outside the map, 0 on the background
        if (fx >= 0.0f && fx <= fW - 1.0f && fy >= 0.0f && fy <= fH - 1.0f) {
            const int xi = (int)fx, yi = (int)fy;
            const unsigned long long key = s.zbuf[yi * W + xi];
            if (key != kEmptyKey) {
                int i0, i1, i2;
                load_face<FL>(a.rc, s, (int)(key & 0xFFFFFFFFull), &i0, &i1, &i2);
                const FaceVerts fv = face_verts(s, i0, i1, i2);
                float b[3], pz;
                face_eval(fv, s.xs[xi], s.ys[yi], b, &pz);
                const float d0 = kp_distance(cl, a.rc.verts[3 * i0], a.rc.verts[3 * i0 + 1], a.rc.verts[3 * i0 + 2]);
                const float d1 = kp_distance(cl, a.rc.verts[3 * i1], a.rc.verts[3 * i1 + 1], a.rc.verts[3 * i1 + 2]);
                const float d2 = kp_distance(cl, a.rc.verts[3 * i2], a.rc.verts[3 * i2 + 1], a.rc.verts[3 * i2 + 2]);
                sampled = (b[0] * d0 + b[1] * d1) + b[2] * d2;
            }
        }
        const float own = kp_distance(cl, X0, X1, X2);
        a.out_vis[(size_t)n * V + v] = fabsf(sampled - own) < a.dist_thr ? 1 : 0;
    }
}

extern "C" int nemo_project_keypoints(nemo_mesh_t m, const nemo_camera* cam, const float* campos, const float* theta,
                                      const float* rt, const float* principal, int32_t n, float dist_thr,
                                      int32_t restrict_to_boundary, float* out_kp, uint8_t* out_vis, void* stream) {
    if (n == 0) return NEMO_OK;
    if (!m || (!rt && (!campos || !theta)) || !out_kp || !out_vis || n < 0)
        return fail(NEMO_ERR_INVALID, "nemo_project_keypoints: bad argument");
    if (check_cam(cam)) return NEMO_ERR_INVALID;
    KeypointVisArgs a;
    a.rc.verts = m->verts; a.rc.faces = m->faces; a.rc.V = m->V; a.rc.F = m->F;
    a.rc.outward = 0;
    for (int k = 0; k < 6; ++k) a.rc.aabb[k] = m->aabb[k];
    a.rc.face_bits = 0; a.rc.depth_margin = 0.0f; a.rc.depth_scale = 0.0f; a.rc.depth_qmax = 0.0f;
    a.rc.cc = make_cam_const(cam->H, cam->W, cam->focal);
    int rcode = plan_layout(cam->H, cam->W, m->V, m->F, kRasterThreads, &a.rc.L);
    if (rcode) return rcode;
    a.campos = campos; a.theta = theta; a.rt = rt; a.principal = principal; a.def_px = cam->px; a.def_py = cam->py;
    a.focal = cam->focal; a.dist_thr = dist_thr; a.clamp_to_image = restrict_to_boundary != 0;
    a.out_kp = out_kp; a.out_vis = out_vis;
    auto kern = a.rc.L.faces >= 0 ? keypoint_vis_kernel<true> : keypoint_vis_kernel<false>;
    if (set_lds_limit(kern, a.rc.L.total)) return NEMO_ERR_HIP;
    hipLaunchKernelGGL(kern, dim3(n), dim3(kRasterThreads), a.rc.L.total, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return NEMO_OK;
}

// ------------------------------------------------------------------------------------------
// feature-bank bookkeeping of multi-object batches (kernels in bank_ops.hip)
// ------------------------------------------------------------------------------------------
extern "C" int nemo_gather_features(const float* feat, const float* weight, const int32_t* input_valid, const int32_t* bank_shifts,
                                    const int32_t* sample_shifts, const int32_t* sample_indexs, int32_t N, int32_t M, int32_t K,
                                    int32_t C, int32_t L, int32_t KK, float* out_feat, float* out_weight,
                                    int32_t* forward_mapping, void* stream) {
    if (N < 0 || M <= 0 || K < 0 || C <= 0 || L <= 0 || KK < 0) return fail(NEMO_ERR_INVALID, "nemo_gather_features: bad size");
    if ((long long)N * K > 0x7FFFFFFFLL) return fail(NEMO_ERR_UNSUPPORTED, "nemo_gather_features: N * K too large");
    if (!bank_shifts || !sample_shifts || !sample_indexs || (KK > 0 && (!out_feat || !out_weight)) ||
        (N * K > 0 && (!feat || !weight || !input_valid || !forward_mapping)))
        return fail(NEMO_ERR_INVALID, "nemo_gather_features: null argument");
    if (N * K > 0) nemo::launch_slot_map(input_valid, bank_shifts, sample_shifts, sample_indexs, N, M, K, forward_mapping, (hipStream_t)stream);
    if (KK > 0)
        nemo::launch_gather_features(feat, weight, bank_shifts, sample_shifts, sample_indexs, forward_mapping, N, M, K, C, L, KK, out_feat,
                                     out_weight, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return NEMO_OK;
}

extern "C" int nemo_gather_features_backward(const float* g_feat, const float* g_weight, const float* feat, const float* weight,
                                             const int32_t* forward_mapping, int32_t N, int32_t K, int32_t C, float* gi_feat,
                                             float* gi_weight, void* stream) {
    if (N < 0 || K < 0 || C <= 0) return fail(NEMO_ERR_INVALID, "nemo_gather_features_backward: bad size");
    if ((long long)N * K > 0x7FFFFFFFLL) return fail(NEMO_ERR_UNSUPPORTED, "nemo_gather_features_backward: N * K too large");
    if (N * K == 0) return NEMO_OK;
    if (!g_feat || !g_weight || !feat || !weight || !forward_mapping || !gi_feat || !gi_weight)
        return fail(NEMO_ERR_INVALID, "nemo_gather_features_backward: null argument");
    nemo::launch_gather_features_backward(g_feat, g_weight, feat, weight, forward_mapping, N * K, C, gi_feat, gi_weight, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return NEMO_OK;
}

extern "C" int nemo_gather_idx(const int32_t* input_valid, const int32_t* sample_shifts, const int32_t* sample_indexs,
                               const int32_t* sample_n_verts, const int32_t* bank_starts, int32_t N, int32_t M, int32_t K,
                               int32_t* object_idx, int32_t* source_vert_idx, int32_t* verts_start, int32_t* num_verts, void* stream) {
    if (N < 0 || M <= 0 || K < 0) return fail(NEMO_ERR_INVALID, "nemo_gather_idx: bad size");
    if ((long long)N * K > 0x7FFFFFFFLL) return fail(NEMO_ERR_UNSUPPORTED, "nemo_gather_idx: N * K too large");
    if (N * K == 0) return NEMO_OK;
    if (!input_valid || !sample_shifts || !sample_indexs || !sample_n_verts || !bank_starts || !object_idx || !source_vert_idx ||
        !verts_start || !num_verts)
        return fail(NEMO_ERR_INVALID, "nemo_gather_idx: null argument");
    nemo::launch_gather_idx(input_valid, sample_shifts, sample_indexs, sample_n_verts, bank_starts, N, M, K, object_idx, source_vert_idx,
                            verts_start, num_verts, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return NEMO_OK;
}

extern "C" int nemo_mask_weight(const int32_t* object_idx, const int32_t* source_vert_idx, const int32_t* verts_start,
                                const int32_t* verts_end, const float* verts_dist_weight, float weight_x, float weight_c,
                                int32_t clutter_start, int32_t S, int32_t K, int32_t V, float* out, void* stream) {
    if (S < 0 || K < 0 || V <= 0) return fail(NEMO_ERR_INVALID, "nemo_mask_weight: bad size");
    if (S == 0 || K == 0) return NEMO_OK;
    if (!object_idx || !source_vert_idx || !verts_start || !verts_end || !verts_dist_weight || !out)
        return fail(NEMO_ERR_INVALID, "nemo_mask_weight: null argument");
    nemo::launch_mask_weight(object_idx, source_vert_idx, verts_start, verts_end, verts_dist_weight, weight_x, weight_c, clutter_start, S,
                             K, V, out, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return NEMO_OK;
}
