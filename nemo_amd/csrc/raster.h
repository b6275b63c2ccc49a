// Workgroup-level rasteriser front end shared by the solve / render / render-backward kernels.
// One workgroup (8 waves) owns one camera pose; everything lives in LDS.  The kernels are bound by
// vector-instruction issue, so the structure is about executing as few wave-instructions as possible.
//
// Common pieces:
//   project_vertices   every vertex -> NDC xy, view depth and its reciprocal (one float4 record per vertex);
//                      project_vertices_resident keeps the world vertices in registers over all steps
//   compact_covered    each wave scans its own interleaved 64-pixel chunks of the key buffer (only the rows
//                      the faces can touch), resets them for the next step and compacts the covered
//                      (face, pixel) entries into its own segment: the pixel work that follows runs on
//                      dense lanes, and because a wave only ever reads its own segment in a fixed order
//                      every float sum is reproducible.
//
// Bit-exact rasteriser (render kernels: what a caller sees as an image):
//   raster_pass1       one face per lane, registers only.  A conservative bounding box, then a cheap
//                      conservative edge-sign test (fma arithmetic with an error margin) over its pixels;
//                      survivors are recorded in a 64-bit mask.  The wave reserves queue space with one
//                      DPP scan + one LDS atomic per class and the lanes write (face, pixel) pairs: faces
//                      nearer than the object centre go to the near half of the queue, the rest to the
//                      far half.
//   raster_pass2       the queue is consumed densely, one pair per lane: exact bounding-box test, exact
//                      barycentrics and depth, 64-bit LDS atomic min on (depth bits << 32 | face).  The far
//                      half runs after the near half and skips a pair whose face cannot beat the depth
//                      already stored.  Every decision that can change the winner uses the exact
//                      arithmetic of geom.h, so the result equals a serial scan over the faces.
//
// Rasteriser of the optimisation loop (no image leaves the kernel):
//   raster_direct      when faces that look away can be skipped (closed solid, camera outside) the ones that
//                      look at the camera are listed first; faces are then handed out in tasks of 64 from
//                      an LDS counter; a lane orients its face's edge functions
//                      and walks the pixel centres of its bounding box with incrementally stepped edge
//                      functions and depth; 32-bit atomic min on (quantised depth << face bits | face).  No
//                      queue, no second pass; the covered entries are compacted in place (compact_covered32).
#pragma once
#include "geom.h"
#include "split_f16.h"

#pragma clang fp contract(off)

namespace nemo {

constexpr int kMaxLds = 160 * 1024;
constexpr int kRasterThreads = 512;           // render kernels: 2 waves per SIMD, up to 256 registers each
constexpr int kSolveThreads = 512;            // optimisation kernel (1024 threads would cap it at 128 registers: spills in the serial sections)
constexpr int kSchedSteps = 512;              // optimiser steps whose Adam constants are tabulated in LDS
constexpr int kMaxSide = 128;                // queue entry = xi | yi << 7 | face << 14
constexpr int kPixBits = 14;
constexpr int kMaxFaces = 1 << (32 - kPixBits);

__device__ inline unsigned int pack_entry(int xi, int yi, int f) {
    return (unsigned)xi | ((unsigned)yi << 7) | ((unsigned)f << kPixBits);
}
__device__ inline void unpack_entry(unsigned int e, int* xi, int* yi, int* f) {
    *xi = e & 127; *yi = (e >> 7) & 127; *f = e >> kPixBits;
}

struct LdsLayout {
    int zbuf;                   // u64 [HW]
    int verts;                  // float4 [V]: NDC x, NDC y, view depth Z, 1/Z
    int faces;                  // int4 [F]: three vertex indices (or -1: faces are read from global memory)
    int xs, ys;                 // f32 [W], [H]: NDC centre of pixel column xi / row yi
    int queue;                  // u32 [qcap]: pair queue, later the per-wave covered segments
    int qcap;                   // entries
    int seg;                    // entries per wave segment (multiple of 64)
    int waves;                  // waves of the workgroup the layout is for
    int red;                    // f32 [16 values][16 waves]
    int tot;                    // f32 [16]
    int pc;                     // PoseCam
    int opt;                    // optimiser state (solve kernel) + queue counter
    int sched;                  // float2 [kSchedSteps]: per-step Adam step size and sqrt of the second bias correction
    int flist;                  // u16 [F]: faces that look at the camera this step (optimisation loop; -1: not kept)
    // optimisation loop computing the similarity table on demand (-1 / 0: not kept)
    int mask_words;             // 32-bit words of a vertex-block mask (one bit per 32-vertex block the mesh can span)
    int present;                // u32 [32-pixel tiles][mask_words]: bit (pixel tile, vertex block) = that tile of the table has been computed
    int present_fb, fb_words;   // u32 [fb_words]: bit (pixel tile, block of faces) = every table tile a face of the block can read there is present
    int fbmask;                 // u32 [face blocks][mask_words]: the vertex blocks the faces of a face block use
    int wmask;                  // u32 [waves][mask_words]: per wave, the vertex blocks to compute for one pixel tile
    int od_words;               // words from `present` on (cleared at set-up)
    int total;
};

inline int align16(int x) { return (x + 15) & ~15; }

inline LdsLayout make_layout(int H, int W, int V, int F, bool faces_in_lds, int waves) {
    LdsLayout L;
    L.waves = waves;
    const int HW = H * W;
    int o = 0;
    L.zbuf = o; o = align16(o + 8 * HW);
    L.verts = o; o = align16(o + 16 * V);
    if (faces_in_lds) { L.faces = o; o = align16(o + 16 * F); } else L.faces = -1;
    L.xs = o; o = align16(o + 4 * W);
    L.ys = o; o = align16(o + 4 * H);
    const int chunks = (HW + 63) / 64;
    L.seg = ((chunks + waves - 1) / waves) * 64;
    L.qcap = L.seg * waves;
    if (L.qcap < 8192) L.qcap = 8192;
    L.queue = o; o = align16(o + 4 * L.qcap);
    L.red = o; o = align16(o + 4 * 16 * 16);
    L.tot = o; o = align16(o + 4 * 16);
    L.pc = o; o = align16(o + (int)sizeof(PoseCam));
    L.opt = o; o = align16(o + 256);
    L.sched = o; o = align16(o + 8 * kSchedSteps);
    L.flist = -1;
    L.mask_words = 0; L.present = L.present_fb = L.fbmask = L.wmask = -1; L.fb_words = 0; L.od_words = 0;
    L.total = o;
    return L;
}

// Layout of the optimisation loop: 32-bit keys (quantised depth above the face index), the covered entries
// are compacted in place into the key buffer (no queue) and a face is three 16-bit vertex indices.  Half the
// size of the layout above, so that two workgroups share a compute unit.
constexpr int kPackedFaceVerts = 1 << 16;     // faces are kept in LDS only when the vertex indices fit 16 bits
constexpr int kFaceBlockShift = 5;            // faces per face block of the on-demand presence map: 32
// vblocks > 0: with the presence maps of the on-demand similarity table, for a mesh that spans at most that many 32-vertex blocks.
inline LdsLayout make_solve_layout(int H, int W, int V, int F, bool faces_in_lds, int waves, int vblocks = 0) {
    LdsLayout L;
    L.waves = waves;
    const int chunks = (H * W + 63) / 64;
    int o = 0;
    L.zbuf = o; o = align16(o + 4 * 64 * chunks);      // whole chunks: entries may land in the padding of the last one
    L.verts = o; o = align16(o + 16 * V);
    if (faces_in_lds) { L.faces = o; o = align16(o + 8 * F); } else L.faces = -1;
    L.xs = o; o = align16(o + 4 * W);
    L.ys = o; o = align16(o + 4 * H);
    L.queue = L.zbuf; L.qcap = 0; L.seg = 0;
    L.red = o; o = align16(o + 4 * 16 * 16);
    L.tot = o; o = align16(o + 4 * 16);
    L.pc = o; o = align16(o + (int)sizeof(PoseCam));
    L.opt = o; o = align16(o + 256);
    L.sched = o; o = align16(o + 8 * kSchedSteps);
    if (F < (1 << 16)) { L.flist = o; o = align16(o + 2 * F); } else L.flist = -1;
    L.mask_words = 0; L.present = L.present_fb = L.fbmask = L.wmask = -1; L.fb_words = 0; L.od_words = 0;
    if (vblocks > 0) {
        const int npt32 = (H * W + 31) / 32, nfb = (F + (1 << kFaceBlockShift) - 1) >> kFaceBlockShift;
        L.mask_words = (vblocks + 31) / 32;
        L.fb_words = (npt32 * nfb + 31) / 32;
        L.present = o;
        L.present_fb = L.present + 4 * npt32 * L.mask_words;
        L.fbmask = L.present_fb + 4 * L.fb_words;
        L.wmask = L.fbmask + 4 * nfb * L.mask_words;
        L.od_words = npt32 * L.mask_words + L.fb_words + nfb * L.mask_words + waves * L.mask_words;
        o = align16(o + 4 * L.od_words);
    }
    L.total = o;
    return L;
}

// One mesh of a handle that holds several meshes back to back (a multi-class bank): vertices [v0, v0 + nv), faces
// [f0, f0 + nf) whose indices all lie in that vertex range, and the per-mesh constants of the optimisation loop.
struct MeshPart {
    int v0, nv, f0, nf;
    float aabb[6];
    float depth_margin, depth_scale, depth_qmax;
    int face_bits;
    int outward;                // this part's faces in the handle's outward-wound copy are wound outward (closed solid)
};

struct RasterCtx {
    const float* verts;         // first vertex of the mesh (or of the part) that is rasterised
    const int32_t* faces;       // its faces; their vertex indices count from vertex `vbase` of the handle
    int vbase = 0;              // subtracted from every face index (0 unless a part of a larger handle is rasterised)
    int V, F;
    int outward;                // faces are wound counter-clockwise seen from outside a closed solid (optimisation loop only)
    float aabb[6];              // bounding box of the vertices
    // optimisation loop: key = quantised depth << face_bits | face (see depth_key32)
    int face_bits;              // 2^face_bits > F, at least 9
    float depth_margin;         // view depths lie within T_z +- depth_margin (bounding radius of the vertices, plus slack)
    float depth_scale;          // quantisation steps per unit of depth
    float depth_qmax;           // largest quantised depth
    CamConst cc;
    LdsLayout L;
};

constexpr unsigned int kEmptyKey32 = 0xFFFFFFFFu;

struct Lds {
    unsigned long long* zbuf;
    unsigned int* zkey; // the same buffer as the 32-bit keys of the optimisation loop
    float4* verts;
    int4* faces;        // LDS copy or nullptr
    uint2* faces2;      // the same area as packed faces (optimisation loop): x = i0 | i1 << 16, y = i2
    float *xs, *ys, *red, *tot;
    unsigned int* queue;
    unsigned int* qcount;    // [4]: render kernels: pairs in the near / far half; optimisation loop: row range, task counter, front faces
    unsigned short* flist;   // optimisation loop: front-face list or nullptr
    unsigned int *present, *present_fb, *fbmask, *wmask;   // on-demand similarity table (see LdsLayout) or nullptr
    PoseCam* pc;
    char* opt;
};

__device__ inline Lds carve(char* base, const LdsLayout& L) {
    Lds s;
    s.zbuf = (unsigned long long*)(base + L.zbuf);
    s.zkey = (unsigned int*)(base + L.zbuf);
    s.verts = (float4*)(base + L.verts);
    s.faces = L.faces >= 0 ? (int4*)(base + L.faces) : nullptr;
    s.faces2 = L.faces >= 0 ? (uint2*)(base + L.faces) : nullptr;
    s.xs = (float*)(base + L.xs); s.ys = (float*)(base + L.ys);
    s.red = (float*)(base + L.red); s.tot = (float*)(base + L.tot);
    s.queue = (unsigned int*)(base + L.queue);
    s.pc = (PoseCam*)(base + L.pc);
    s.opt = base + L.opt;
    s.qcount = (unsigned int*)(base + L.opt + 240);
    s.flist = L.flist >= 0 ? (unsigned short*)(base + L.flist) : nullptr;
    s.present = L.present >= 0 ? (unsigned int*)(base + L.present) : nullptr;
    s.present_fb = L.present >= 0 ? (unsigned int*)(base + L.present_fb) : nullptr;
    s.fbmask = L.present >= 0 ? (unsigned int*)(base + L.fbmask) : nullptr;
    s.wmask = L.present >= 0 ? (unsigned int*)(base + L.wmask) : nullptr;
    return s;
}

// ---- reductions (fixed order => bitwise reproducible) ------------------------------------------
template <int CTRL>
__device__ inline float dpp_add(float v) {
    int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true);
    return v + __int_as_float(t);
}

// Sum within each row of 16 lanes; every lane of the row returns the row total.
__device__ inline float row16_sum(float v) {
    v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);   // row_half_mirror
    v = dpp_add<0x140>(v);   // row_mirror
    return v;
}

// Sum over the 64 lanes of a wave; every lane returns the total.
__device__ inline float wave_sum(float v) {
    v = row16_sum(v);
    float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return ((r0 + r1) + r2) + r3;
}

// Block total of NV (<= 16) values per thread for a 16-wave workgroup.  red: LDS [NV][16], tot: LDS [NV].
// After the call tot[k] holds the total, visible to every thread.
template <int NV>
__device__ inline void block_sum(const float* vals, float* red, float* tot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        float sv = wave_sum(vals[k]);
        if (lane == 0) red[k * 16 + wave] = sv;
    }
    __syncthreads();
    if (threadIdx.x < NV * 16) {      // one 16-lane row per value: lanes = the waves' partials (zero padded)
        float sv = row16_sum(red[threadIdx.x]);
        if ((threadIdx.x & 15) == 0) tot[threadIdx.x >> 4] = sv;
    }
    __syncthreads();
}

// ---- front end ------------------------------------------------------------------------------------
// One-time per-workgroup set-up: pixel-centre tables, the face list, an empty key buffer.  Ends with a barrier.
__device__ inline void raster_prologue(const RasterCtx& rc, const Lds& s) {
    const int H = rc.cc.H, W = rc.cc.W;
    for (int i = threadIdx.x; i < W; i += blockDim.x) s.xs[i] = pix_to_ndc(W - 1 - i, W, H);
    for (int i = threadIdx.x; i < H; i += blockDim.x) s.ys[i] = pix_to_ndc(H - 1 - i, H, W);
    if (s.faces)
        for (int i = threadIdx.x; i < rc.F; i += blockDim.x)
            s.faces[i] = make_int4(rc.faces[3 * i], rc.faces[3 * i + 1], rc.faces[3 * i + 2], 0);
    for (int p = threadIdx.x; p < H * W; p += blockDim.x) s.zbuf[p] = kEmptyKey;
    if (threadIdx.x == 0) { s.qcount[0] = 0; s.qcount[1] = 0; }
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s.red[i] = 0.0f;
    __syncthreads();      // the reduction scratch must be zero before any wave writes a partial into it
}

// The same for the layout of the optimisation loop (make_solve_layout).
__device__ inline void solve_prologue(const RasterCtx& rc, const Lds& s) {
    const int H = rc.cc.H, W = rc.cc.W;
    for (int i = threadIdx.x; i < W; i += blockDim.x) s.xs[i] = pix_to_ndc(W - 1 - i, W, H);
    for (int i = threadIdx.x; i < H; i += blockDim.x) s.ys[i] = pix_to_ndc(H - 1 - i, H, W);
    if (s.faces2)
        for (int i = threadIdx.x; i < rc.F; i += blockDim.x)
            s.faces2[i] = make_uint2((unsigned)(rc.faces[3 * i] - rc.vbase) | ((unsigned)(rc.faces[3 * i + 1] - rc.vbase) << 16),
                                     (unsigned)(rc.faces[3 * i + 2] - rc.vbase));
    const int slots = ((H * W + 63) >> 6) << 6;
    for (int p = threadIdx.x; p < slots; p += blockDim.x) s.zkey[p] = kEmptyKey32;
    if (s.present)
        for (int i = threadIdx.x; i < rc.L.od_words; i += blockDim.x) s.present[i] = 0u;
    if (threadIdx.x == 0) { s.qcount[0] = 0; s.qcount[1] = 0; }
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s.red[i] = 0.0f;
    __syncthreads();
}

template <bool FACES_LDS>
__device__ inline void load_face2(const RasterCtx& rc, const Lds& s, int f, int* i0, int* i1, int* i2) {
    if (FACES_LDS) { const uint2 t = s.faces2[f]; *i0 = t.x & 0xFFFFu; *i1 = t.x >> 16; *i2 = t.y; }
    else { *i0 = rc.faces[3 * f + 0] - rc.vbase; *i1 = rc.faces[3 * f + 1] - rc.vbase; *i2 = rc.faces[3 * f + 2] - rc.vbase; }
}

template <bool FACES_LDS>
__device__ inline void load_face(const RasterCtx& rc, const Lds& s, int f, int* i0, int* i1, int* i2) {
    if (FACES_LDS) { const int4 t = s.faces[f]; *i0 = t.x; *i1 = t.y; *i2 = t.z; }
    else { *i0 = rc.faces[3 * f + 0]; *i1 = rc.faces[3 * f + 1]; *i2 = rc.faces[3 * f + 2]; }
}

__device__ inline FaceVerts face_verts(const Lds& s, int i0, int i1, int i2) {
    FaceVerts fv;
    const float4 a = s.verts[i0], b = s.verts[i1], c = s.verts[i2];
    fv.x0 = a.x; fv.y0 = a.y; fv.z0 = a.z;
    fv.x1 = b.x; fv.y1 = b.y; fv.z1 = b.z;
    fv.x2 = c.x; fv.y2 = c.y; fv.z2 = c.z;
    return fv;
}

// View-space X, Y of the face's vertices recovered from their NDC position and depth
// (x_ndc = a X / Z + bx); used by the backward chain only.
struct ViewInv { float ia, bx, by; };
__device__ inline ViewInv view_inv(const CamConst& cc, const Lds& s) {
    ViewInv v; v.ia = 1.0f / cc.a; v.bx = s.pc->bx; v.by = s.pc->by; return v;
}
__device__ inline void view_coords(const ViewInv& vi, const FaceVerts& fv, float* X0, float* X1, float* X2) {
    const float ia = vi.ia, bx = vi.bx, by = vi.by;
    X0[0] = (fv.x0 - bx) * fv.z0 * ia; X0[1] = (fv.y0 - by) * fv.z0 * ia; X0[2] = fv.z0;
    X1[0] = (fv.x1 - bx) * fv.z1 * ia; X1[1] = (fv.y1 - by) * fv.z1 * ia; X1[2] = fv.z1;
    X2[0] = (fv.x2 - bx) * fv.z2 * ia; X2[1] = (fv.y2 - by) * fv.z2 * ia; X2[2] = fv.z2;
}

// Pose camera in s.pc must be visible (barrier before).  Ends with a barrier.
__device__ inline void project_vertices(const RasterCtx& rc, const Lds& s) {
    float R[9], T[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = s.pc->R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) T[i] = s.pc->T[i];
    const float bx = s.pc->bx, by = s.pc->by;
    for (int v = threadIdx.x; v < rc.V; v += blockDim.x) {
        float Xv[3], xn, yn;
        project_vertex(rc.cc, R, T, bx, by, rc.verts[3 * v], rc.verts[3 * v + 1], rc.verts[3 * v + 2], Xv, &xn, &yn);
        s.verts[v] = make_float4(xn, yn, Xv[2], 1.0f / Xv[2]);
    }
    __syncthreads();
}

// Projection for the optimisation loop: the world vertices of a thread stay in registers over all steps
// (wv[k] = vertex threadIdx.x + k * blockDim.x), reciprocal instead of division.  Ends with a barrier.
constexpr int kVertsPerThread = 4;
#pragma clang fp contract(fast)
__device__ inline void project_vertices_resident(const RasterCtx& rc, const Lds& s, const float (*wv)[3]) {
    float R[9], T[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = s.pc->R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) T[i] = s.pc->T[i];
    const float bx = s.pc->bx, by = s.pc->by, a = rc.cc.a;
#pragma unroll
    for (int k = 0; k < kVertsPerThread; ++k) {
        const int v = threadIdx.x + k * blockDim.x;
        if (v < rc.V) {
            const float X = wv[k][0] * R[0] + wv[k][1] * R[3] + wv[k][2] * R[6] + T[0];
            const float Y = wv[k][0] * R[1] + wv[k][1] * R[4] + wv[k][2] * R[7] + T[1];
            const float Z = wv[k][0] * R[2] + wv[k][1] * R[5] + wv[k][2] * R[8] + T[2];
            const float iz = __builtin_amdgcn_rcpf(Z);
            s.verts[v] = make_float4(a * X * iz + bx, a * Y * iz + by, Z, iz);
        }
    }
    __syncthreads();
}
#pragma clang fp contract(off)

// Inclusive prefix sum of one int per lane over the wave (DPP within 16-lane rows, row totals via readlane).
// Sixteen values per lane -> lane l of the wave returns the wave total of value (l & 15).  Butterfly over
// the 16 lanes of a row in which every step halves the values a lane carries: at the step for index bit
// b a lane keeps the half of its values whose bit b equals its own lane bit b and adds the same half
// received from a partner that differs in lane bit b and agrees on the higher ones (row_mirror = xor 15,
// row_half_mirror = xor 7, quad swaps = xor 2, xor 1).  15 DPP adds instead of 64, then two cross-row
// shuffles.  The order of the additions is fixed, so the result is reproducible.
template <int CTRL>
__device__ inline float dpp_get(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ inline float wave_sum16_transposed(const float* vals) {
    const int lane = threadIdx.x & 63;
    float a[8], b4[4], c2[2];
    const bool h3 = lane & 8, h2 = lane & 4, h1 = lane & 2, h0 = lane & 1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float keep = h3 ? vals[8 + j] : vals[j], send = h3 ? vals[j] : vals[8 + j];
        a[j] = keep + dpp_get<0x140>(send);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float keep = h2 ? a[4 + j] : a[j], send = h2 ? a[j] : a[4 + j];
        b4[j] = keep + dpp_get<0x141>(send);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float keep = h1 ? b4[2 + j] : b4[j], send = h1 ? b4[j] : b4[2 + j];
        c2[j] = keep + dpp_get<0x4E>(send);
    }
    float r = (h0 ? c2[1] : c2[0]) + dpp_get<0xB1>(h0 ? c2[0] : c2[1]);
    r += __shfl_xor(r, 16);
    r += __shfl_xor(r, 32);
    return r;
}

// Block total of exactly 16 values per thread (same contract as block_sum<16>).
__device__ inline void block_sum16(const float* vals, float* red, float* tot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float sv = wave_sum16_transposed(vals);
    if (lane < 16) red[lane * 16 + wave] = sv;
    __syncthreads();
    if (threadIdx.x < 256) {          // one 16-lane row per value: lanes = the waves' partials (zero padded)
        float t = row16_sum(red[threadIdx.x]);
        if ((threadIdx.x & 15) == 0) tot[threadIdx.x >> 4] = t;
    }
    __syncthreads();
}

// Minimum / maximum of an int over the 64 lanes of a wave; every lane returns the result.
template <int CTRL>
__device__ inline int dpp_min(int v) { return min(v, __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false)); }
template <int CTRL>
__device__ inline int dpp_max(int v) { return max(v, __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false)); }
__device__ inline int wave_min(int v) {
    v = dpp_min<0xB1>(v); v = dpp_min<0x4E>(v); v = dpp_min<0x141>(v); v = dpp_min<0x140>(v);
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ inline int wave_max(int v) {
    v = dpp_max<0xB1>(v); v = dpp_max<0x4E>(v); v = dpp_max<0x141>(v); v = dpp_max<0x140>(v);
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

template <int CTRL>
__device__ inline int dpp_shr_add(int v) {
    return v + __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
__device__ inline int wave_inclusive_scan(int v, int* total) {
    v = dpp_shr_add<0x111>(v);   // row_shr:1
    v = dpp_shr_add<0x112>(v);   // row_shr:2
    v = dpp_shr_add<0x114>(v);   // row_shr:4
    v = dpp_shr_add<0x118>(v);   // row_shr:8
    const int r0 = __builtin_amdgcn_readlane(v, 15), r1 = __builtin_amdgcn_readlane(v, 31),
              r2 = __builtin_amdgcn_readlane(v, 47), r3 = __builtin_amdgcn_readlane(v, 63);
    const int row = (threadIdx.x & 63) >> 4;
    v += row == 0 ? 0 : (row == 1 ? r0 : (row == 2 ? r0 + r1 : r0 + r1 + r2));
    *total = r0 + r1 + r2 + r3;
    return v;
}

// Coverage test of one (face, pixel) pair and depth-buffer update in the reference's arithmetic, bit for bit.
// (EXACT is always true: the optimisation loop has its own rasteriser, raster_direct.)
template <bool EXACT>
__device__ inline void eval_pair(int face, const FaceVerts& fv, int xi, int yi, int W, const Lds& s) {
    static_assert(EXACT, "the queue rasteriser only exists in its exact form");
    const float xf = s.xs[xi], yf = s.ys[yi];
    const float xmin = fminf(fminf(fv.x0, fv.x1), fv.x2), xmax = fmaxf(fmaxf(fv.x0, fv.x1), fv.x2);
    const float ymin = fminf(fminf(fv.y0, fv.y1), fv.y2), ymax = fmaxf(fmaxf(fv.y0, fv.y1), fv.y2);
    if (xf > xmax || xf < xmin || yf > ymax || yf < ymin) return;     // exact bounding-box rejection
    eval_and_min(face, fv, yi * W + xi, xf, yf, s.zbuf);
}

// Key buffer must be empty and s.qcount[0..1] == 0 on entry; ends with a barrier.
template <bool FACES_LDS, bool EXACT>
__device__ inline void raster_pass1(const RasterCtx& rc, const Lds& s) {
    const int H = rc.cc.H, W = rc.cc.W;
    const int half = rc.L.qcap >> 1;
    // approximate pixel centres: x(j) = x_org + j * dx in the flipped index j (error of a few ulp of the
    // extent, covered by the margin of the conservative tests)
    const float rx = ndc_range(W, H), ry = ndc_range(H, W);
    const float dx = rx / (float)W, dy = ry / (float)H;
    const float x_org = -0.5f * rx + 0.5f * dx, y_org = -0.5f * ry + 0.5f * dy;
    const float inv_dx = (float)W / rx, inv_dy = (float)H / ry;
    const float z_mid3 = 3.0f * s.pc->T[2];          // three times the view depth of the object origin
#ifdef NEMO_STAMPS_P1
    long long tq0 = clock64(), tq[4] = {0, 0, 0, 0};
#define P1STAMP(i) do { long long t_ = clock64(); tq[i] += t_ - tq0; tq0 = t_; } while (0)
#else
#define P1STAMP(i) do { } while (0)
#endif
    for (int base = 0; base < rc.F; base += blockDim.x) {
        const int f = base + threadIdx.x;
        FaceVerts fv = {};
        bool ok = false, zpos = false, near = false;
        int jx0 = 0, jy0 = 0, nx = 0, ny = 0;
        float A0 = 0, B0 = 0, C0 = 0, A1 = 0, B1 = 0, C1 = 0, A2 = 0, B2 = 0, C2 = 0, tol = 0;
        if (f < rc.F) {
            int i0, i1, i2;
            load_face<FACES_LDS>(rc, s, f, &i0, &i1, &i2);
            fv = face_verts(s, i0, i1, i2);
            const float zmax = fmaxf(fmaxf(fv.z0, fv.z1), fv.z2);
            const float farea = edge_fn(fv.x0, fv.y0, fv.x1, fv.y1, fv.x2, fv.y2);
            ok = !(zmax < 0.0f) && !(farea <= kEps && farea >= -kEps) && (farea == farea);
            const float xmin = fminf(fminf(fv.x0, fv.x1), fv.x2), xmax = fmaxf(fmaxf(fv.x0, fv.x1), fv.x2);
            const float ymin = fminf(fminf(fv.y0, fv.y1), fv.y2), ymax = fmaxf(fmaxf(fv.y0, fv.y1), fv.y2);
            // conservative pixel range (flipped indices); 1e-3 pixel of slack, the exact test is in pass 2
            const float tx0 = (xmin - x_org) * inv_dx - 1e-3f, tx1 = (xmax - x_org) * inv_dx + 1e-3f;
            const float ty0 = (ymin - y_org) * inv_dy - 1e-3f, ty1 = (ymax - y_org) * inv_dy + 1e-3f;
            ok = ok && !(tx1 < 0.0f) && !(ty1 < 0.0f) && !(tx0 > (float)(W - 1)) && !(ty0 > (float)(H - 1));
            jx0 = clamp_index(ceilf(tx0), W, 0); jy0 = clamp_index(ceilf(ty0), H, 0);
            const int jx1 = clamp_index(floorf(tx1), W, W - 1), jy1 = clamp_index(floorf(ty1), H, H - 1);
            nx = jx1 - jx0 + 1; ny = jy1 - jy0 + 1;
            ok = ok && nx > 0 && ny > 0;
            zpos = fv.z0 > 0.0f && fv.z1 > 0.0f && fv.z2 > 0.0f;
            near = (fv.z0 + fv.z1) + fv.z2 < z_mid3;
            // edge functions as planes e_i(x,y) = A_i x + B_i y + C_i, oriented so that inside means >= 0
            const float sgn = farea < 0.0f ? -1.0f : 1.0f;
            A0 = sgn * (fv.y2 - fv.y1); B0 = -sgn * (fv.x2 - fv.x1); C0 = -(A0 * fv.x1 + B0 * fv.y1);
            A1 = sgn * (fv.y0 - fv.y2); B1 = -sgn * (fv.x0 - fv.x2); C1 = -(A1 * fv.x2 + B1 * fv.y2);
            A2 = sgn * (fv.y1 - fv.y0); B2 = -sgn * (fv.x1 - fv.x0); C2 = -(A2 * fv.x0 + B2 * fv.y0);
            const float ext = fmaxf(xmax - xmin, ymax - ymin);
            const float mag = fmaxf(fmaxf(fmaxf(fabsf(xmin), fabsf(xmax)), fmaxf(fabsf(ymin), fabsf(ymax))), 1.5f);
            tol = 2e-5f * ext * mag;      // >> the rounding error of either evaluation (a few 1e-7 * ext * mag)
            if (!(tol == tol) || !zpos) tol = __builtin_inff();   // odd geometry: keep every pixel of the box
        }
        P1STAMP(0);
        // Candidate pixels, one row of the box at a time: inside means e_i(x, y) >= -tol for the three
        // oriented edge planes, i.e. a_i * t + s_i >= 0 in box-column units t, which is an interval of t.
        // mask bit (r * 8 + c) = column c of row r may be inside.
        unsigned int mask_lo = 0, mask_hi = 0;        // rows 0-3, rows 4-7
        int cnt_near = 0, cnt_far = 0;
        if (ok) {
            if (nx <= 8 && ny <= 8) {
                const float x0 = __builtin_fmaf((float)jx0, dx, x_org);
                const float a0 = A0 * dx, a1 = A1 * dx, a2 = A2 * dx;
                const float i0 = -__builtin_amdgcn_rcpf(a0), i1 = -__builtin_amdgcn_rcpf(a1), i2 = -__builtin_amdgcn_rcpf(a2);
                const float k0 = __builtin_fmaf(A0, x0, C0) + tol, k1 = __builtin_fmaf(A1, x0, C1) + tol,
                            k2 = __builtin_fmaf(A2, x0, C2) + tol;
                const float ninf = -__builtin_inff(), pinf = __builtin_inff();
                const float tiny = 1e-20f;
                for (int r = 0; r < ny; ++r) {
                    const float yf = __builtin_fmaf((float)(jy0 + r), dy, y_org);
                    const float s0 = __builtin_fmaf(B0, yf, k0), s1 = __builtin_fmaf(B1, yf, k1), s2 = __builtin_fmaf(B2, yf, k2);
                    const float q0 = s0 * i0, q1 = s1 * i1, q2 = s2 * i2;      // root of a_i * t + s_i
                    float lo = fmaxf(fmaxf(a0 > tiny ? q0 : ninf, a1 > tiny ? q1 : ninf), a2 > tiny ? q2 : ninf);
                    float hi = fminf(fminf(a0 < -tiny ? q0 : pinf, a1 < -tiny ? q1 : pinf), a2 < -tiny ? q2 : pinf);
                    // an edge parallel to the row decides the whole row
                    const bool dead = (fabsf(a0) <= tiny && s0 < 0.0f) || (fabsf(a1) <= tiny && s1 < 0.0f) ||
                                      (fabsf(a2) <= tiny && s2 < 0.0f);
                    lo = fmaxf(ceilf(lo - 1e-3f), 0.0f);
                    hi = fminf(floorf(hi + 1e-3f), (float)(nx - 1));
                    if (!dead && lo <= hi) {
                        const int cl = (int)lo, ch = (int)hi;
                        const unsigned int bits = ((2u << ch) - 1u) & ~((1u << cl) - 1u);
                        if (r < 4) mask_lo |= bits << (r * 8); else mask_hi |= bits << ((r - 4) * 8);
                    }
                }
            } else {
                // close-up face: evaluate its box in place (exact path), no queueing
                for (int jy = jy0; jy < jy0 + ny; ++jy)
                    for (int jx = jx0; jx < jx0 + nx; ++jx) eval_pair<EXACT>(f, fv, W - 1 - jx, H - 1 - jy, W, s);
            }
        }
        {
            const int n = __popc(mask_lo) + __popc(mask_hi);
            cnt_near = near ? n : 0;
            cnt_far = near ? 0 : n;
        }
        P1STAMP(1);
        int total = 0, incl = 0;
        incl = wave_inclusive_scan(cnt_near | (cnt_far << 16), &total);
        if (total) {
            unsigned int bn = 0, bf = 0;
            if ((threadIdx.x & 63) == 0) {
                if (total & 0xFFFF) bn = atomicAdd(&s.qcount[0], (unsigned)(total & 0xFFFF));
                if (total >> 16) bf = atomicAdd(&s.qcount[1], (unsigned)(total >> 16));
            }
            bn = __builtin_amdgcn_readfirstlane(bn);
            bf = __builtin_amdgcn_readfirstlane(bf);
            const int excl = incl - (cnt_near | (cnt_far << 16));
            unsigned int slot = near ? bn + (unsigned)(excl & 0xFFFF) : bf + (unsigned)(excl >> 16);
            const int off = near ? 0 : half;
#pragma unroll
            for (int halfsel = 0; halfsel < 2; ++halfsel) {
                unsigned int m = halfsel ? mask_hi : mask_lo;
                while (m) {
                    const int k = __ffs((int)m) - 1;
                    m &= m - 1;
                    const int xi = W - 1 - (jx0 + (k & 7)), yi = H - 1 - (jy0 + (k >> 3) + 4 * halfsel);
                    if (slot < (unsigned)half) s.queue[off + slot] = pack_entry(xi, yi, f);
                    else eval_pair<EXACT>(f, fv, xi, yi, W, s);   // queue half full: evaluate in place
                    ++slot;
                }
            }
        }
        P1STAMP(2);
    }
#ifdef NEMO_STAMPS_P1
    if ((threadIdx.x & 63) == 0 && threadIdx.x < 512) { float* d = (float*)(s.red); (void)d; }
    if (threadIdx.x == 64) { s.tot[0] = (float)tq[0]; s.tot[1] = (float)tq[1]; s.tot[2] = (float)tq[2]; }
#endif
    __syncthreads();
}

// Rasteriser of the optimisation loop: every face walks the pixel centres of its bounding box with
// incrementally evaluated edge functions (planes in the affine pixel-centre coordinates) and depth,
// no queue and no second pass.  Same decisions as the exact rule except for pixel centres within
// rounding distance (1e-7 of the face's extent) of an edge.  A face is typically 1-2 pixels, a box 2-6
// centres.  A lane steps two centres of its face per trip, straight-line code; the lanes of a wave leave
// the loop when the largest box of the wave is done.  Faces with a vertex behind the camera take
// the exact rule.  Key buffer must be empty on entry; ends with a barrier.
//
// Keys are 32 bits: the depth, quantised over the range the object can occupy this step (view depth of the
// mesh origin +- the bounding radius), sits above the face index.  With 2000 faces that leaves 21 bits, a
// step of about 1e-6 of the object's size: two faces compete for a pixel centre only when one occludes the
// other (neighbours on a surface never both contain it), and then their depths differ by far more; closer
// than a step, the smaller face index wins, as it does at equal depth with full-precision keys.
// The quantised value is read from the mantissa of 2^23 + q, so it costs one fused multiply-add.
#pragma clang fp contract(fast)
__device__ inline unsigned int depth_key32(const RasterCtx& rc, float z0, float pz, int face) {
    const float q = fminf(fmaxf((pz - z0) * rc.depth_scale, 0.0f), rc.depth_qmax);
    return ((unsigned int)q << rc.face_bits) | (unsigned int)face;
}

struct WalkState {
    int n, nx, p, c, row_skip, face;
    float e0, e1, e2, su, a0, a1, a2, asu, r0, r1, r2, rsu, numer;
};

struct WalkGrid { int H, W; float dx, dy, x_org, y_org, inv_dx, inv_dy; };

// Straight-line: every quantity is computed for every lane and the pixel count n is zeroed for faces
// that are skipped.  Returns true for a face that needs the exact rule (a vertex behind the camera).
__device__ inline bool walk_setup(const WalkGrid& g, int f, bool valid, bool cull, float zscale, const FaceVerts& fv, WalkState* w, int* box) {
    const int H = g.H, W = g.W;
    const float zmax = fmaxf(fmaxf(fv.z0, fv.z1), fv.z2), zmin = fminf(fminf(fv.z0, fv.z1), fv.z2);
    // signed area = e0 evaluated at vertex 0
    const float A0 = fv.y2 - fv.y1, B0 = fv.x1 - fv.x2;
    const float A1 = fv.y0 - fv.y2, B1 = fv.x2 - fv.x0;
    const float A2 = fv.y1 - fv.y0, B2 = fv.x0 - fv.x1;
    const float farea = A0 * (fv.x0 - fv.x1) + B0 * (fv.y0 - fv.y1);
    bool ok = valid && !(zmax < 0.0f) && (fabsf(farea) > kEps);           // false for NaN
    // An outward-wound face seen from outside the solid looks at the camera when its signed area is positive.
    ok = ok && (!cull || farea > 0.0f);
    const float sgn = farea < 0.0f ? -1.0f : 1.0f;                          // edge functions oriented: inside means all positive
    const float xmin = fminf(fminf(fv.x0, fv.x1), fv.x2), xmax = fmaxf(fmaxf(fv.x0, fv.x1), fv.x2);
    const float ymin = fminf(fminf(fv.y0, fv.y1), fv.y2), ymax = fmaxf(fmaxf(fv.y0, fv.y1), fv.y2);
    const float tx0 = (xmin - g.x_org) * g.inv_dx - 1e-3f, tx1 = (xmax - g.x_org) * g.inv_dx + 1e-3f;
    const float ty0 = (ymin - g.y_org) * g.inv_dy - 1e-3f, ty1 = (ymax - g.y_org) * g.inv_dy + 1e-3f;
    // fmaxf / fminf drop NaNs; a NaN box is rejected through farea
    const int jx0 = (int)fminf(fmaxf(ceilf(tx0), 0.0f), (float)(W - 1)), jy0 = (int)fminf(fmaxf(ceilf(ty0), 0.0f), (float)(H - 1));
    const int jx1 = (int)fminf(fmaxf(floorf(tx1), -1.0f), (float)(W - 1)), jy1 = (int)fminf(fmaxf(floorf(ty1), -1.0f), (float)(H - 1));
    const int bx = jx1 - jx0 + 1, by = jy1 - jy0 + 1;
    ok = ok && bx > 0 && by > 0 && tx0 <= (float)(W - 1) && ty0 <= (float)(H - 1);
    const bool slow = ok && !(zmin > 0.0f);
    box[0] = jx0; box[1] = jy0; box[2] = jx1; box[3] = jy1;
    w->face = f;
    w->nx = bx;
    w->n = (ok && !slow) ? bx * by : 0;
    w->row_skip = bx - W;
    w->c = 0;
    w->p = (H - 1 - jy0) * W + (W - 1 - jx0);          // pixel of the box origin; flipped indices grow downwards in p
    const float xo = (float)jx0 * g.dx + g.x_org, yo = (float)jy0 * g.dy + g.y_org;
    // e_i(x, y) = (x - xa)(yb - ya) - (y - ya)(xb - xa): value at the origin, step per column, step per row
    w->e0 = sgn * ((xo - fv.x1) * A0 + (yo - fv.y1) * B0);
    w->e1 = sgn * ((xo - fv.x2) * A1 + (yo - fv.y2) * B1);
    w->e2 = sgn * ((xo - fv.x0) * A2 + (yo - fv.y0) * B2);
    const float sdx = sgn * g.dx, sdy = sgn * g.dy;
    w->a0 = A0 * sdx; w->a1 = A1 * sdx; w->a2 = A2 * sdx;
    // a row step also rewinds the nx columns walked
    const float back = (float)bx;
    w->r0 = B0 * sdy - back * w->a0; w->r1 = B1 * sdy - back * w->a1; w->r2 = B2 * sdy - back * w->a2;
    // u_i = e_i z_j z_k; corrected barycentrics u_i / sum(u); depth z0 z1 z2 (e0 + e1 + e2) / sum(u)
    const float z12 = fv.z1 * fv.z2, z02 = fv.z0 * fv.z2, z01 = fv.z0 * fv.z1;
    w->su = w->e0 * z12 + w->e1 * z02 + w->e2 * z01;
    w->asu = w->a0 * z12 + w->a1 * z02 + w->a2 * z01;
    w->rsu = w->r0 * z12 + w->r1 * z02 + w->r2 * z01;
    w->numer = (fv.z0 * z12) * fabsf(farea) * zscale;      // depth in quantisation steps = numer / su
    return slow;
}

// One pixel centre of the walk: test, depth, key, then step to the next centre (k = index of this centre).
// qc = 2^23 - (start of the depth range in steps): the sum is 2^23 + q, whose mantissa is the integer q.
__device__ inline void walk_step(WalkState& w, int k, unsigned int* zkey, float qc, int face_bits) {
    const float lo = fminf(fminf(w.e0, w.e1), w.e2);
    const float t = __builtin_fmaf(w.numer, __builtin_amdgcn_rcpf(w.su), qc);   // su > 0 inside: every term has the sign of its e_i
    if (lo > 0.0f && k < w.n) atomicMin(&zkey[w.p], (__float_as_uint(t) << face_bits) | (unsigned int)w.face);
    ++w.c;
    const bool wrap = w.c == w.nx;
    const float wf = wrap ? 1.0f : 0.0f;
    w.e0 = w.e0 + w.a0 + wf * w.r0; w.e1 = w.e1 + w.a1 + wf * w.r1; w.e2 = w.e2 + w.a2 + wf * w.r2;
    w.su = w.su + w.asu + wf * w.rsu;
    w.p = w.p - 1 + (wrap ? w.row_skip : 0);
    w.c = wrap ? 0 : w.c;
}

// When back faces can be skipped (closed solid, camera outside) a first pass over all faces lists the ones
// that look at the camera (signed screen area > 0), so that the walk below runs on lanes that all hold a
// face with work: about half the faces of a solid look away.  The list order depends on which wave adds
// its faces first, which the atomic-min key buffer does not care about.
//
// Work is handed out in tasks of 64 consecutive (listed) faces (one per lane): a wave starts with task = its
// index and takes the next one from an LDS counter (s.qcount[2], which the caller keeps at the number
// of waves between steps), so waves whose faces have small boxes pick up more tasks.  The caller keeps
// s.qcount[3] (length of the list) at 0 between steps.
template <bool FACES_LDS>
__device__ inline void raster_direct(const RasterCtx& rc, const Lds& s) {
    WalkGrid g;
    g.H = rc.cc.H; g.W = rc.cc.W;
    const float rx = ndc_range(g.W, g.H), ry = ndc_range(g.H, g.W);
    g.dx = rx / (float)g.W; g.dy = ry / (float)g.H;
    g.x_org = -0.5f * rx + 0.5f * g.dx; g.y_org = -0.5f * ry + 0.5f * g.dy;
    g.inv_dx = (float)g.W / rx; g.inv_dy = (float)g.H / ry;
#ifdef NEMO_STAMPS_P1
    long long tq0 = clock64(), tq[4] = {0, 0, 0, 0};
    int iters = 0;
#endif
    const int lane = threadIdx.x & 63;
    const bool cull = s.pc->cull != 0;          // set with the pose: outward faces and the camera outside the solid
    const float qz0 = s.pc->qz0, qc = s.pc->qc; // depth range of this step
    const int face_bits = rc.face_bits;
    const bool listed = cull && s.flist != nullptr;
    int nfaces = rc.F;
    if (listed) {
        for (int base = 0; base < rc.F; base += blockDim.x) {
            const int f = base + threadIdx.x;
            bool front = false;
            if (f < rc.F) {
                int i0, i1, i2;
                load_face2<FACES_LDS>(rc, s, f, &i0, &i1, &i2);
                const float2 a = *reinterpret_cast<const float2*>(&s.verts[i0]), b = *reinterpret_cast<const float2*>(&s.verts[i1]),
                             c = *reinterpret_cast<const float2*>(&s.verts[i2]);
                front = (c.y - b.y) * (a.x - b.x) + (b.x - c.x) * (a.y - b.y) > 0.0f;      // the signed area of walk_setup
            }
            const unsigned long long m = __ballot(front);
            if (m) {
                unsigned int off = 0;
                if (lane == 0) off = atomicAdd(&s.qcount[3], (unsigned)__popcll(m));
                off = __builtin_amdgcn_readfirstlane(off);
                if (front) s.flist[off + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0))] = (unsigned short)f;
            }
        }
        __syncthreads();
        nfaces = (int)s.qcount[3];
    }
#ifdef NEMO_STAMPS_P1
    { long long t_ = clock64(); tq[2] = t_ - tq0; tq0 = t_; }
    int ntasks_done = 0;
#endif
    const int ntask = (nfaces + 63) >> 6;
    int row_lo = g.H, row_hi = -1;              // pixel rows (yi) this thread's faces may touch
    int task = threadIdx.x >> 6;
    // the vertex records of the next task are requested before the pixel walk of the current one
    FaceVerts nfv;
    int nf = 0;
    {
        const int idx = task * 64 + lane;
        int i0 = 0, i1 = 0, i2 = 0;
        if (idx < nfaces) {
            nf = listed ? (int)s.flist[idx] : idx;
            load_face2<FACES_LDS>(rc, s, nf, &i0, &i1, &i2);
        }
        nfv = face_verts(s, i0, i1, i2);
    }
    while (task < ntask) {
        const bool valid = task * 64 + lane < nfaces;
        const int f = nf;
        const FaceVerts fv = nfv;
        unsigned int nt = 0;
        if (lane == 0) nt = atomicAdd(&s.qcount[2], 1u);
        const int next = (int)__builtin_amdgcn_readfirstlane(nt);
        {
            const int idx = next * 64 + lane;
            int i0 = 0, i1 = 0, i2 = 0;
            if (idx < nfaces) {
                nf = listed ? (int)s.flist[idx] : idx;
                load_face2<FACES_LDS>(rc, s, nf, &i0, &i1, &i2);
            }
            nfv = face_verts(s, i0, i1, i2);
        }
        WalkState w;
        int box[4];
        const bool slow = walk_setup(g, f, valid, cull, rc.depth_scale, fv, &w, box);
        if (w.n > 0 || slow) {
            const int r0 = g.H - 1 - box[3], r1 = g.H - 1 - box[1];
            row_lo = min(row_lo, r0); row_hi = max(row_hi, r1);
        }
        if (__any(slow)) {                       // rare: a face crossing the camera plane
            if (slow)
                for (int jy = box[1]; jy <= box[3]; ++jy)
                    for (int jx = box[0]; jx <= box[2]; ++jx) {
                        const int xi = g.W - 1 - jx, yi = g.H - 1 - jy;
                        const float xf = s.xs[xi], yf = s.ys[yi];
                        const float xmin = fminf(fminf(fv.x0, fv.x1), fv.x2), xmax = fmaxf(fmaxf(fv.x0, fv.x1), fv.x2);
                        const float ymin = fminf(fminf(fv.y0, fv.y1), fv.y2), ymax = fmaxf(fmaxf(fv.y0, fv.y1), fv.y2);
                        if (xf > xmax || xf < xmin || yf > ymax || yf < ymin) continue;
                        float b[3], pz;
                        face_eval(fv, xf, yf, b, &pz);
                        if (pz < 0.0f || !(b[0] > 0.0f && b[1] > 0.0f && b[2] > 0.0f)) continue;
                        atomicMin(&s.zkey[yi * g.W + xi], depth_key32(rc, qz0, pz, f));
                    }
        }
        P1STAMP(0);
        for (int k = 0; __any(k < w.n); k += 2) {
#ifdef NEMO_STAMPS_P1
            ++iters;
#endif
            walk_step(w, k, s.zkey, qc, face_bits);
            walk_step(w, k + 1, s.zkey, qc, face_bits);
        }
        P1STAMP(1);
#ifdef NEMO_STAMPS_P1
        ++ntasks_done;
#endif
        task = next;
    }
#ifdef NEMO_STAMPS_P1
    if (threadIdx.x == 64) { s.tot[0] = (float)tq[0]; s.tot[1] = (float)tq[1]; s.tot[2] = (float)iters; s.tot[3] = (float)tq[2]; s.tot[4] = (float)ntasks_done; s.tot[5] = (float)nfaces; }
#endif
    // rows that can hold covered pixels: s.qcount[0] = first, s.qcount[1] = last + 1 (the caller keeps
    // them at H and 0 between steps)
    row_lo = wave_min(row_lo); row_hi = wave_max(row_hi);
    if (lane == 0 && row_hi >= row_lo) {
        atomicMin(&s.qcount[0], (unsigned)row_lo);
        atomicMax(&s.qcount[1], (unsigned)(row_hi + 1));
    }
    __syncthreads();
}

// ---- the tile-binned form of the same rasteriser (diagnostic: nemo_raster_forms) -----------------------------------
// Faces binned to 8 x 8-pixel tiles in LDS, then a wave per tile with lanes = the tile's 64 pixels and one loop trip per
// face of the tile's bin: the shape that suits triangles of tens of pixels.  A lane owns its pixel, so the depth test
// runs in a register and the key buffer is written with plain stores (no atomics); against that, every (face, tile)
// pair repeats the face's whole set-up on all 64 lanes.  The reference's meshes project to 2 - 3 pixel centres per
// face, where the face-per-lane walk above wins by a wide margin; this form is kept to measure that (it is not used by
// the solver).  Same arithmetic per (face, pixel) as walk_setup / walk_step except that the edge functions are
// evaluated at the pixel instead of stepped to it, so a centre within rounding distance of an edge may differ.
struct TileBins {
    unsigned int* cnt;      // [tiles + 1]: faces per tile, then (after the scan) the first entry of every tile
    unsigned int* cur;      // [tiles]: fill cursors
    unsigned short* ent;    // [cap]: face indices, tile by tile
    int cap;
};
constexpr int kBinShift = 3;       // 8 x 8 pixels

template <bool FACES_LDS>
__device__ inline void raster_tiled(const RasterCtx& rc, const Lds& s, const TileBins& tb, long long* t_bin) {
    WalkGrid g;
    g.H = rc.cc.H; g.W = rc.cc.W;
    const float rx = ndc_range(g.W, g.H), ry = ndc_range(g.H, g.W);
    g.dx = rx / (float)g.W; g.dy = ry / (float)g.H;
    g.x_org = -0.5f * rx + 0.5f * g.dx; g.y_org = -0.5f * ry + 0.5f * g.dy;
    g.inv_dx = (float)g.W / rx; g.inv_dy = (float)g.H / ry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const bool cull = s.pc->cull != 0;
    const float qc = s.pc->qc;
    const int face_bits = rc.face_bits;
    const int TX = (g.W + 7) >> kBinShift, TY = (g.H + 7) >> kBinShift, NT = TX * TY;
    for (int t = threadIdx.x; t <= NT; t += blockDim.x) tb.cnt[t] = 0u;
    __syncthreads();
    // pass 1 and 2 over the faces: count per tile, then fill.  A face's pixel box as in walk_setup.
    for (int pass = 0; pass < 2; ++pass) {
        for (int f = threadIdx.x; f < rc.F; f += blockDim.x) {
            int i0, i1, i2;
            load_face2<FACES_LDS>(rc, s, f, &i0, &i1, &i2);
            const FaceVerts fv = face_verts(s, i0, i1, i2);
            WalkState w; int box[4];
            const bool slow = walk_setup(g, f, true, cull, rc.depth_scale, fv, &w, box);
            if (w.n <= 0 || slow) continue;
            // box is in flipped indices j (pixel = side - 1 - j)
            const int x0 = g.W - 1 - box[2], x1 = g.W - 1 - box[0], y0 = g.H - 1 - box[3], y1 = g.H - 1 - box[1];
            for (int ty = y0 >> kBinShift; ty <= y1 >> kBinShift; ++ty)
                for (int tx = x0 >> kBinShift; tx <= x1 >> kBinShift; ++tx) {
                    const int t = ty * TX + tx;
                    if (pass == 0) atomicAdd(&tb.cnt[t], 1u);
                    else { const unsigned int slot = atomicAdd(&tb.cur[t], 1u); if ((int)slot < tb.cap) tb.ent[slot] = (unsigned short)f; }
                }
        }
        __syncthreads();
        if (pass == 0) {
            if (threadIdx.x < 64) {           // exclusive scan of the NT counters by one wave: a run of consecutive tiles per lane
                const int per = (NT + 63) >> 6, t0 = lane * per;
                unsigned int c[4] = {0u, 0u, 0u, 0u};            // NT <= 256 (maps up to 128 x 128)
                int sum = 0;
                for (int k = 0; k < per; ++k) { c[k] = t0 + k < NT ? tb.cnt[t0 + k] : 0u; sum += (int)c[k]; }
                int total;
                unsigned int run = (unsigned int)(wave_inclusive_scan(sum, &total) - sum);
                for (int k = 0; k < per; ++k)
                    if (t0 + k < NT) { tb.cnt[t0 + k] = run; tb.cur[t0 + k] = run; run += c[k]; }
                if (lane == 0) tb.cnt[NT] = (unsigned int)total;
            }
            __syncthreads();
        }
    }
    if (t_bin) *t_bin = clock64();
    // a wave per tile (tile t belongs to wave t mod waves), lanes = pixels
    for (int t = wave; t < NT; t += nwaves) {
        const unsigned int e0 = tb.cnt[t], e1 = min(tb.cnt[t + 1], (unsigned)tb.cap);
        if (e1 <= e0) continue;
        const int xi = (t % TX) * 8 + (lane & 7), yi = (t / TX) * 8 + (lane >> 3);
        const bool inmap = xi < g.W && yi < g.H;
        const float xo = (float)(g.W - 1 - xi) * g.dx + g.x_org, yo = (float)(g.H - 1 - yi) * g.dy + g.y_org;
        unsigned int best = kEmptyKey32;
#pragma unroll 4
        for (unsigned int e = e0; e < e1; ++e) {      // (unrolled: the LDS reads of four faces are in flight together)
            const int f = tb.ent[e];
            int i0, i1, i2;
            load_face2<FACES_LDS>(rc, s, f, &i0, &i1, &i2);
            const FaceVerts fv = face_verts(s, i0, i1, i2);
            const float A0 = fv.y2 - fv.y1, B0 = fv.x1 - fv.x2, A1 = fv.y0 - fv.y2, B1 = fv.x2 - fv.x0, A2 = fv.y1 - fv.y0, B2 = fv.x0 - fv.x1;
            const float farea = A0 * (fv.x0 - fv.x1) + B0 * (fv.y0 - fv.y1);
            const float sgn = farea < 0.0f ? -1.0f : 1.0f;
            const float q0 = sgn * ((xo - fv.x1) * A0 + (yo - fv.y1) * B0), q1 = sgn * ((xo - fv.x2) * A1 + (yo - fv.y2) * B1),
                        q2 = sgn * ((xo - fv.x0) * A2 + (yo - fv.y0) * B2);
            const float z12 = fv.z1 * fv.z2, z02 = fv.z0 * fv.z2, z01 = fv.z0 * fv.z1;
            const float su = q0 * z12 + q1 * z02 + q2 * z01;
            const float numer = (fv.z0 * z12) * fabsf(farea) * rc.depth_scale;
            const float tq = __builtin_fmaf(numer, __builtin_amdgcn_rcpf(su), qc);
            const unsigned int key = (__float_as_uint(tq) << face_bits) | (unsigned int)f;
            if (fminf(fminf(q0, q1), q2) > 0.0f) best = min(best, key);
        }
        if (inmap && best != kEmptyKey32) s.zkey[yi * g.W + xi] = best;
    }
    __syncthreads();
}
#pragma clang fp contract(off)

// Evaluates n queued (face, pixel) pairs, kPairBatch per thread at a time so that the dependent LDS
// reads (entry -> face record -> vertex records) of several pairs are in flight together.  FAR: pairs
// of the far half, skipped when the face's nearest vertex is clearly behind the depth already stored.
constexpr int kPairBatch = 2;
template <bool FACES_LDS, bool EXACT, bool FAR>
__device__ inline void eval_queue(const RasterCtx& rc, const Lds& s, const unsigned int* q, int n) {
    const int W = rc.cc.W;
    for (int i0 = threadIdx.x; i0 < n; i0 += kPairBatch * blockDim.x) {
        int xi[kPairBatch], yi[kPairBatch], f[kPairBatch], v0[kPairBatch], v1[kPairBatch], v2[kPairBatch];
        bool ok[kPairBatch];
#pragma unroll
        for (int u = 0; u < kPairBatch; ++u) {
            const int i = i0 + u * blockDim.x;
            ok[u] = i < n;
            unpack_entry(ok[u] ? q[i] : 0u, &xi[u], &yi[u], &f[u]);
        }
#pragma unroll
        for (int u = 0; u < kPairBatch; ++u) load_face<FACES_LDS>(rc, s, f[u], &v0[u], &v1[u], &v2[u]);
        FaceVerts fv[kPairBatch];
        unsigned long long cur[kPairBatch];
#pragma unroll
        for (int u = 0; u < kPairBatch; ++u) {
            fv[u] = face_verts(s, v0[u], v1[u], v2[u]);
            if (FAR) cur[u] = s.zbuf[yi[u] * W + xi[u]];
        }
#pragma unroll
        for (int u = 0; u < kPairBatch; ++u) {
            if (!ok[u]) continue;
            if (FAR) {
                // the interpolated depth of a covered pixel is >= zmin * (1 - 5e-7): a face whose nearest
                // vertex is clearly behind the stored depth cannot win this pixel
                const float zmin = fminf(fminf(fv[u].z0, fv[u].z1), fv[u].z2);
                if (cur[u] != kEmptyKey && zmin * 0.999999f > __uint_as_float((unsigned)(cur[u] >> 32))) continue;
            }
            eval_pair<EXACT>(f[u], fv[u], xi[u], yi[u], W, s);
        }
    }
}

// Ends with a barrier and with both queue counters reset.
template <bool FACES_LDS, bool EXACT>
__device__ inline void raster_pass2(const RasterCtx& rc, const Lds& s) {
    const int half = rc.L.qcap >> 1;
    const int n_near = min((int)s.qcount[0], half), n_far = min((int)s.qcount[1], half);
    eval_queue<FACES_LDS, EXACT, false>(rc, s, s.queue, n_near);
    __syncthreads();
    eval_queue<FACES_LDS, EXACT, true>(rc, s, s.queue + half, n_far);
    __syncthreads();
    if (threadIdx.x == 0) { s.qcount[0] = 0; s.qcount[1] = 0; }
}

template <bool FACES_LDS, bool EXACT>
__device__ inline void raster_faces(const RasterCtx& rc, const Lds& s) {
    raster_pass1<FACES_LDS, EXACT>(rc, s);
    raster_pass2<FACES_LDS, EXACT>(rc, s);
}

// Each wave compacts the covered pixels of its own chunks (chunk c = 64 consecutive pixels, owned by
// wave c % 16) into its segment of the queue area and resets the key buffer.  Returns the number of
// entries in this wave's segment (pack_entry format).  Ends with a barrier.
__device__ inline int compact_covered(const RasterCtx& rc, const Lds& s, unsigned int** seg_out, int row0 = 0, int row1 = -1) {
    const int HW = rc.cc.H * rc.cc.W;
    if (row1 < 0) row1 = rc.cc.H;
    const int c0 = (row0 * rc.cc.W) >> 6, c1 = (row1 * rc.cc.W + 63) >> 6;      // chunks holding rows [row0, row1)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned int* seg = s.queue + wave * rc.L.seg;
    const float inv_w = 1.0f / (float)rc.cc.W;
    int count = 0;
    constexpr int CB = 4;                       // chunks in flight per wave: their key reads overlap
    for (int cb = c0 + wave; cb < c1; cb += CB * rc.L.waves) {
        unsigned long long key[CB];
#pragma unroll
        for (int u = 0; u < CB; ++u) {
            const int c = cb + u * rc.L.waves;
            const int p = c * 64 + lane;
            key[u] = (c < c1 && p < HW) ? s.zbuf[p] : kEmptyKey;
        }
#pragma unroll
        for (int u = 0; u < CB; ++u) {
            const int p = (cb + u * rc.L.waves) * 64 + lane;
            const bool cov = key[u] != kEmptyKey;
            const unsigned long long m = __ballot(cov);
            if (cov) {
                s.zbuf[p] = kEmptyKey;
                const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                const int yi = (int)(((float)p + 0.5f) * inv_w);       // exact for p < 2^22
                seg[count + rank] = pack_entry(p - yi * rc.cc.W, yi, (int)(key[u] & 0xFFFFFFFFull));
            }
            count += __popcll(m);
        }
    }
    __syncthreads();
    *seg_out = seg;
    return count;
}

// ---- similarity-table tiles computed on demand (optimisation loop) -------------------------------------------
// The loop reads table entry (vertex, pixel) only where a face of that vertex covers the pixel, about one entry
// in a hundred over 300 steps, so a wave computes the table itself, a tile of 32 vertices x 32 pixels at a time,
// the first time one of its covered pixels needs it: the same matrix-core product as the table kernel
// (split_table_kernel: two-piece f16 operands, piece pairs (l,h) (h,l) (h,h) per 16-channel step, f32
// accumulation), with the same operand values in the same order, so an entry is bit for bit what the table kernel
// would have written.  The vertex bank's operand arrives as ready fragments from the mesh handle; the image's is
// read from the feature map and split in registers (scale_clamp, split8, as the table kernel's first phase does),
// once per call: all the tiles a pixel tile needs are computed together.
struct TileCtx {
    const float* feat;        // feature map of this hypothesis's image, [C, HW]
    int HW;
    float* table;             // its table ([pixel tile of 128][vertex group of 4][128][4])
    const uint4* bank;        // the vertex bank as f16-piece fragments (first piece; the second follows at `apiece`)
    size_t apiece;
    float unscale;
    int nks;                  // 16-channel steps (C / 16)
    int vg_n;
    int vb_first, nvb;        // the 32-vertex blocks the rendered mesh (part) spans
    int nfb;                  // face blocks of the mesh (part): the face-block presence map's row length in bits
};

// NA: channel steps of the bank's operand requested together (every request is a round trip to the L2: the loop, short of
// registers, keeps two in flight; the kernel that computes the predicted tiles in bulk asks for more).
template <int NKS, int NA_ = 2>
__device__ inline void table_tiles_nks(const float* __restrict__ col, size_t HW, bool pok, const uint4* __restrict__ ap, size_t apiece,
                                       float4* __restrict__ out, float us, int g0, int vg_n, int nvb, const unsigned int* vbmask) {
    // the image's operand: all channel steps of the 32 pixels, both pieces, resident for the whole call (all of it
    // requested at once: the eight values of a channel step turn into the eight registers of its two pieces)
    uint4 b[NKS][SG_NP];
    constexpr int NB = NKS;
#pragma unroll
    for (int k0 = 0; k0 < NKS; k0 += NB) {
        float x[NB][8];
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) x[i][j] = col[(size_t)((k0 + i) * 16 + j) * HW];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) x[i][j] = pok ? scale_clamp(x[i][j]) : 0.0f;
            split8(x[i], b[k0 + i]);
        }
    }
    constexpr int NA = NKS < NA_ ? NKS : NA_;   // channel steps of the bank's operand in flight
    for (int w = 0; w * 32 < nvb; ++w) {
        unsigned int todo = (unsigned)__builtin_amdgcn_readfirstlane((int)vbmask[w]);
        while (todo) {
            const int vb = w * 32 + __ffs((int)todo) - 1;
            todo &= todo - 1;
            const uint4* a = ap + (size_t)vb * NKS * 64;
            f32x16 acc = {0};
#pragma unroll
            for (int k0 = 0; k0 < NKS; k0 += NA) {
                uint4 ah[NA], al[NA];
#pragma unroll
                for (int i = 0; i < NA; ++i) { ah[i] = a[(size_t)(k0 + i) * 64]; al[i] = a[apiece + (size_t)(k0 + i) * 64]; }
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_frag(al[i]), as_frag(b[k0 + i][0]), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_frag(ah[i]), as_frag(b[k0 + i][1]), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_frag(ah[i]), as_frag(b[k0 + i][0]), acc, 0, 0, 0);
                }
            }
            // result register quad j of a lane = vertices 32 vb + 8 j + 4 lk .. + 3 of pixel px: one 16-byte table element
            float4* o = out + (size_t)vb * 8 * 128;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (g0 + vb * 8 + 2 * j < vg_n) o[(size_t)(2 * j) * 128] = make_float4(acc[4 * j] * us, acc[4 * j + 1] * us, acc[4 * j + 2] * us, acc[4 * j + 3] * us);
        }
    }
}

// The tiles (vertex block vb_first + v, pixel tile pt32) for every bit v set in vbmask (words in LDS).  A function of
// its own (not inlined): its registers (the image's operand, the accumulators) are then not part of the optimisation
// loop's allocation, which has none to spare; it is called off the loop's common path, and marked cold so that what
// the loop must save around the call is saved on that path and not reloaded in every phase of every step.
__device__ __attribute__((noinline, cold)) void table_tiles(const float* feat, const uint4* bank, float* table, int HW, int nks, int vg_n,
                                                      size_t apiece, float unscale, int vb_first, int nvb, int pt32,
                                                      const unsigned int* vbmask) {
    const int lane = threadIdx.x & 63, lk = lane >> 5, px = lane & 31;
    const int p = pt32 * 32 + px;
    const bool pok = p < HW;
    const float* col = feat + (pok ? p : HW - 1) + (size_t)(lk * 8) * HW;
    const uint4* ap = bank + (size_t)vb_first * nks * 64 + lane;
    const int g0 = vb_first * 8 + lk;            // vertex group of register quad 0 of the first block
    float4* out = reinterpret_cast<float4*>(table) + ((size_t)(pt32 >> 2) * vg_n + g0) * 128 + (pt32 & 3) * 32 + px;
    switch (nks) {
        case 1: table_tiles_nks<1>(col, HW, pok, ap, apiece, out, unscale, g0, vg_n, nvb, vbmask); break;
        case 2: table_tiles_nks<2>(col, HW, pok, ap, apiece, out, unscale, g0, vg_n, nvb, vbmask); break;
        case 4: table_tiles_nks<4>(col, HW, pok, ap, apiece, out, unscale, g0, vg_n, nvb, vbmask); break;
        default: table_tiles_nks<8>(col, HW, pok, ap, apiece, out, unscale, g0, vg_n, nvb, vbmask); break;
    }
}

// Optimisation loop: the same compaction for 32-bit keys, in place.  Chunk c (64 consecutive pixels) belongs to
// wave c % waves for the whole launch; a wave scans its chunks of the touched rows and writes its j-th covered
// entry over the j-th slot of those same chunks (slot j lives at chunk first + (j >> 6) * waves, lane j & 63),
// which it has already read: the key buffer doubles as the entry list and no other wave touches these chunks
// before the next step.  The caller consumes the entries with the same slot arithmetic and leaves every slot
// empty again.  Returns the number of entries; *first_chunk = the wave's first chunk.  No barrier.
// DETECT (on-demand similarity table): while pixel and face of an entry are at hand, its (32-pixel tile, face block)
// presence bit is tested too; *missing = some entry of this wave lacks a tile (the caller then runs ensure_table_tiles,
// otherwise it has nothing to do: the common case after the first steps).
template <bool DETECT = false>
__device__ inline int compact_covered32(const RasterCtx& rc, const Lds& s, int row0, int row1, int* first_chunk, int nfb = 0,
                                        bool* missing = nullptr) {
    const int W = rc.cc.W, nw = rc.L.waves;
    const int c0 = (row0 * W) >> 6, c1 = (row1 * W + 63) >> 6;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned int fmask = (1u << rc.face_bits) - 1u;
    const float inv_w = 1.0f / (float)W;
    const int cw = c0 + ((wave - c0 % nw) + nw) % nw;       // first chunk >= c0 owned by this wave
    int count = 0;
    bool absent_any = false;
    constexpr int CB = 4;                       // chunks in flight per wave: their key reads overlap
    for (int cb = cw; cb < c1; cb += CB * nw) {
        unsigned int key[CB];
#pragma unroll
        for (int u = 0; u < CB; ++u) {
            const int c = cb + u * nw;
            key[u] = c < c1 ? s.zkey[c * 64 + lane] : kEmptyKey32;      // the padding of the last chunk stays empty
        }
#pragma unroll
        for (int u = 0; u < CB; ++u) {
            const int p = (cb + u * nw) * 64 + lane;
            const bool cov = key[u] != kEmptyKey32;
            const unsigned long long m = __ballot(cov);
            if (cov) {
                s.zkey[p] = kEmptyKey32;        // before the entry writes below: LDS accesses of a wave stay in order
                const int j = count + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
                const int yi = (int)(((float)p + 0.5f) * inv_w);       // exact for p < 2^22
                s.zkey[(cw + (j >> 6) * nw) * 64 + (j & 63)] = pack_entry(p - yi * W, yi, (int)(key[u] & fmask));
                if (DETECT) {
                    const unsigned int bit = ((unsigned)p >> 5) * (unsigned)nfb + ((key[u] & fmask) >> kFaceBlockShift);
                    absent_any |= !((s.present_fb[bit >> 5] >> (bit & 31u)) & 1u);
                }
            }
            count += __popcll(m);
        }
    }
    *first_chunk = cw;
    if (DETECT) *missing = absent_any;
    return count;
}

// On-demand table, set-up (once per workgroup, after solve_prologue): the vertex blocks every face block uses.
// Ends with a barrier.
template <bool FACES_LDS>
__device__ inline void build_face_block_masks(const RasterCtx& rc, const Lds& s, const TileCtx& tc) {
    for (int f = threadIdx.x; f < rc.F; f += blockDim.x) {
        int vi[3];
        load_face2<FACES_LDS>(rc, s, f, &vi[0], &vi[1], &vi[2]);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const unsigned int vb = (unsigned)(((rc.vbase + vi[k]) >> 5) - tc.vb_first);
            atomicOr(&s.fbmask[(f >> kFaceBlockShift) * rc.L.mask_words + (vb >> 5)], 1u << (vb & 31u));
        }
    }
    __syncthreads();
}

// On-demand table: a pass of each wave over its own compacted entries (dense lanes, four entries per lane in
// flight, straight-line while nothing is missing).  One presence bit per (pixel tile, face block) says that every
// table tile a face of that block can read at a pixel of that tile is there: the tiles (pixel tile, vertex block)
// for all vertex blocks the face block uses.  Where bits are clear the wave takes one pixel tile at a time: it sets
// the bits of all the face blocks missing there, gathers their vertex blocks, drops those whose tiles exist
// (per-tile presence map) and computes the rest in one call.  A 32-pixel tile lies inside one chunk, so only the
// wave that owns its pixels ever asks for, computes and reads its tiles: no barrier, no other wave waits.
//
// A function of its own (not inlined), with every operand passed by value: the pass then leaves the register
// allocation of the optimisation loop as it is without it.
//
// COLLECT: nothing is computed; the vertex blocks a pixel tile lacks are recorded in `need` (global memory, one row of
// mw words per 32-pixel tile of this hypothesis) for the kernel that computes the first step's tiles in bulk
// (tile_bulk_kernel).  (dx, dy): the entries are taken as if they covered the pixel that many columns / rows away (entries
// that leave the map are skipped): called once per displacement of a small ring, this also lists the tiles the following
// steps are likely to read, as faces move by fractions of a pixel per step.  Displaced pixels may lie in another wave's
// chunks, so the presence words are updated atomically.
typedef __attribute__((address_space(3))) unsigned int lds_u32;

template <bool COLLECT = false>
__device__ inline void ensure_table_tiles_call(lds_u32* zkey, lds_u32* present, lds_u32* present_fb, lds_u32* fbmask,
                                                                  lds_u32* wmask, int W, int waves, int mw, int nfb, int first_chunk, int cnt,
                                                                  const float* feat, const uint4* bank, float* table, int HW, int nks, int vg_n,
                                                                  size_t apiece, float unscale, int vb_first, int nvb,
                                                                  unsigned int* need = nullptr, int H = 0, int dx = 0, int dy = 0) {
    constexpr unsigned int kNone = 0xFFFFFFFFu;
    constexpr int U = 4;
    const int lane = threadIdx.x & 63;
    const int slot_step = 64 * waves;
    const lds_u32* first = zkey + first_chunk * 64 + lane;       // always inside the key buffer
    const lds_u32* slot = first;
    wmask += (threadIdx.x >> 6) * mw;
    bool computed = false;
    // (trip count uniform over the wave: the rare path below needs every lane)
    for (int ib0 = 0; ib0 < cnt; ib0 += 64 * U, slot += U * slot_step) {
        unsigned int code[U];
        bool miss = false;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            bool ok = ib0 + lane + 64 * u < cnt;
            const unsigned int e = *(ok ? slot + u * slot_step : first);    // unconditional read; a lane past the end ...
            int px, py, f;
            unpack_entry(ok ? e : 0u, &px, &py, &f);                        // ... looks at pixel 0 / face 0 and reports nothing
            if (COLLECT) {
                px += dx; py += dy;
                ok = ok && px >= 0 && px < W && py >= 0 && py < H;
                if (!ok) { px = 0; py = 0; }
            }
            const unsigned int pt = (unsigned)(py * W + px) >> 5, fb = (unsigned)f >> kFaceBlockShift;
            const unsigned int bit = pt * (unsigned)nfb + fb;
            const bool absent = ok && !((present_fb[bit >> 5] >> (bit & 31u)) & 1u);
            code[u] = absent ? (pt << 16 | fb) : kNone;
            miss |= absent;
        }
        if (__builtin_expect(!__any(miss), 1)) continue;
        for (;;) {                                   // rare: one round per pixel tile with something missing
            unsigned int mine = kNone;
#pragma unroll
            for (int u = 0; u < U; ++u) mine = min(mine, code[u]);
            const unsigned long long pend = __ballot(mine != kNone);
            if (!pend) break;
            const unsigned int pt = (unsigned)__builtin_amdgcn_readlane((int)mine, __ffsll((long long)pend) - 1) >> 16;
            for (int w = lane; w < mw; w += 64) wmask[w] = 0u;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (code[u] != kNone && (code[u] >> 16) == pt) {
                    const unsigned int fb = code[u] & 0xFFFFu, bit = pt * (unsigned)nfb + fb;
                    code[u] = kNone;
                    __hip_atomic_fetch_or(&present_fb[bit >> 5], 1u << (bit & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    for (int w = 0; w < mw; ++w)
                        __hip_atomic_fetch_or(&wmask[w], fbmask[fb * mw + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            bool some = false;
            for (int w = lane; w < mw; w += 64) {
                unsigned int have;
                if (COLLECT) have = __hip_atomic_fetch_or(&present[pt * mw + w], wmask[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else { have = present[pt * mw + w]; present[pt * mw + w] = have | wmask[w]; }
                const unsigned int lack = wmask[w] & ~have;
                wmask[w] = lack;
                some |= lack != 0u;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            if (COLLECT) {
                for (int w = lane; w < mw; w += 64) {
                    const unsigned int nd = wmask[w];
                    if (nd) atomicOr(&need[pt * mw + w], nd);
                }
            } else if (__any(some)) {
                table_tiles(feat, bank, table, HW, nks, vg_n, apiece, unscale, vb_first, nvb, (int)pt, (const unsigned int*)wmask);
                computed = true;
            }
        }
    }
    if (computed) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");      // the tile stores have landed before the wave reads them
}

__device__ inline void ensure_table_tiles(const RasterCtx& rc, const Lds& s, const TileCtx& tc, int first_chunk, int cnt) {
    ensure_table_tiles_call((lds_u32*)s.zkey, (lds_u32*)s.present, (lds_u32*)s.present_fb, (lds_u32*)s.fbmask, (lds_u32*)s.wmask, rc.cc.W,
                            rc.L.waves, rc.L.mask_words, tc.nfb, first_chunk, cnt, tc.feat, tc.bank, tc.table, tc.HW, tc.nks, tc.vg_n,
                            tc.apiece, tc.unscale, tc.vb_first, tc.nvb);
}

// The tiles the covered entries read, recorded instead of computed (see ensure_table_tiles_call<COLLECT>): at their own
// pixels and, with ring > 0, at the eight pixels `ring` columns / rows away.
__device__ inline void collect_table_tiles(const RasterCtx& rc, const Lds& s, const TileCtx& tc, int first_chunk, int cnt,
                                           unsigned int* need, int ring) {
    for (int k = 4; k < (ring > 0 ? 13 : 5); ++k) {      // k % 9 runs over a 3 x 3 pattern, its centre (4) first and once
        const int j = k % 9, dx = (j % 3 - 1) * ring, dy = (j / 3 - 1) * ring;
        ensure_table_tiles_call<true>((lds_u32*)s.zkey, (lds_u32*)s.present, (lds_u32*)s.present_fb, (lds_u32*)s.fbmask, (lds_u32*)s.wmask,
                                      rc.cc.W, rc.L.waves, rc.L.mask_words, tc.nfb, first_chunk, cnt, tc.feat, tc.bank, tc.table, tc.HW,
                                      tc.nks, tc.vg_n, tc.apiece, tc.unscale, tc.vb_first, tc.nvb, need, rc.cc.H, dx, dy);
    }
}

// The loop starts from presence maps that already hold the tiles computed in bulk: `need` (one row of mw words per
// 32-pixel tile) is the list tile_bulk_kernel worked from.  A (pixel tile, face block) bit is set where every vertex
// block the face block uses is present.  Call after build_face_block_masks; ends with a barrier.
__device__ inline void preset_table_tiles(const RasterCtx& rc, const Lds& s, const TileCtx& tc, const unsigned int* need, int npt32) {
    const int mw = rc.L.mask_words, nfb = tc.nfb;
    for (int i = threadIdx.x; i < npt32 * mw; i += blockDim.x) s.present[i] = need[i];
    __syncthreads();
    const int nbits = npt32 * nfb;
    for (int wi = threadIdx.x; wi * 32 < nbits; wi += blockDim.x) {
        unsigned int word = 0u;
        for (int b = 0; b < 32 && wi * 32 + b < nbits; ++b) {
            const int bit = wi * 32 + b, pt = bit / nfb, fb = bit - pt * nfb;
            unsigned int lacks = 0u, uses = 0u;
            for (int w = 0; w < mw; ++w) {
                const unsigned int m = s.fbmask[fb * mw + w];
                uses |= m; lacks |= m & ~s.present[pt * mw + w];
            }
            if (uses && !lacks) word |= 1u << b;
        }
        s.present_fb[wi] = word;
    }
    __syncthreads();
}

}  // namespace nemo
