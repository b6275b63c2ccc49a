// Device geometry shared by every kernel that rasterises: pose -> (R, T), vertex projection,
// pixel centres, the per-(face, pixel) coverage test and the reverse-mode chain from a pixel's
// barycentric gradient back to the pose parameters.
//
// Arithmetic contract: everything that decides WHICH face covers a pixel is float32 with one
// IEEE rounding per operation, in exactly the association written here (this translation unit
// is built with -ffp-contract=off, and division / sqrt are the correctly rounded forms), and
// sin/cos are evaluated in float64 and rounded once.  The same sequence evaluated on a CPU gives
// the same bits, which is what makes face-index buffers comparable bit for bit.
//
// Semantics restated from PyTorch3D 0.7.1 as the reference uses it:
//   look_at_rotation / campos_to_R_T            reference nemo/utils/mesh.py:222-226
//   PerspectiveCameras(in_ndc=False) + NDC map  reference nemo/models/nemo.py:258-280
//   RasterizeMeshesNaive, blur 0, K = 1, perspective-correct barycentrics
//                                               reference nemo/utils/mesh.py:73-93
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "split_f16.h"      // max_nan

#pragma clang fp contract(off)

namespace nemo {

constexpr float kEps = 1e-8f;          // rasteriser kEpsilon
constexpr float kNormEps = 1e-5f;      // F.normalize eps used by look_at_rotation
constexpr unsigned long long kEmptyKey = 0xFFFFFFFFFFFFFFFFull;

struct CamConst {
    int H, W;
    float a;        // focal / s,  s = min(H,W)/2
    float s;
    float half_w, half_h;
};

__host__ __device__ inline CamConst make_cam_const(int H, int W, float focal) {
    CamConst c;
    c.H = H; c.W = W;
    c.s = (float)(H < W ? H : W) / 2.0f;
    c.a = focal / c.s;
    c.half_w = (float)W / 2.0f;
    c.half_h = (float)H / 2.0f;
    return c;
}

// Per-hypothesis camera: everything the projection, the rasteriser and the backward chain need.
struct PoseCam {
    float R[9];            // row-major, X_view = X_world . R + T  (row vectors)
    float T[3];
    float bx, by;          // x_ndc = a*X/Z + bx,  bx = (W/2 - px)/s
    // saved by the forward for the backward chain
    float C[3];
    float ax[3], ay[3], az[3];   // look-at axes (columns of R0)
    float nv, nw, nu;            // clamped norms of -C, up x z, z x x
    float sn, cs;                // sin, cos of theta
    int degenerate;
    int cull;                    // optimisation loop: faces looking away from the camera may be skipped this step
    float qz0, qc;               // optimisation loop: start of the quantised depth range, and 2^23 - start in steps
};

__device__ inline void normalize3(float x, float y, float z, float* o, float* n_out) {
    float n = sqrtf((x * x + y * y) + z * z);
    n = n > kNormEps ? n : (n != n ? n : kNormEps);
    o[0] = x / n; o[1] = y / n; o[2] = z / n;
    *n_out = n;
}

// The same with one reciprocal instead of three divisions (optimisation loop: a couple of ulp).
__device__ inline void normalize3_fast(float x, float y, float z, float* o, float* n_out) {
    float n = sqrtf((x * x + y * y) + z * z);
    n = n > kNormEps ? n : (n != n ? n : kNormEps);
    const float inv = __builtin_amdgcn_rcpf(n);
    o[0] = x * inv; o[1] = y * inv; o[2] = z * inv;
    *n_out = n;
}

__device__ inline void cross3(const float* a, const float* b, float* o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

__device__ inline void sincos32(float x, float* s, float* c) {
    double sd, cd;
    sincos((double)x, &sd, &cd);
    *s = (float)sd; *c = (float)cd;
}

// sin and cos in f32 arithmetic, about 1 ulp: Cody-Waite reduction by pi/2 and the Cephes minimax
// polynomials on [-pi/4, pi/4].  For the optimisation loop, where theta stays within a few turns.
__device__ inline void sincos_short(float x, float* s, float* c) {
    const float k = rintf(x * 0.63661977236758134f);
    float r = __builtin_fmaf(k, -1.5707962513e+00f, x);
    r = __builtin_fmaf(k, -7.5497894159e-08f, r);
    r = __builtin_fmaf(k, -5.3903029534e-15f, r);
    const float r2 = r * r;
    const float sp = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f), r2 * r, r);
    const float cp = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f),
                                    r2 * r2, __builtin_fmaf(-0.5f, r2, 1.0f));
    const int q = (int)k & 3;
    const float ss = (q & 1) ? cp : sp, cc = (q & 1) ? sp : cp;
    *s = q >= 2 ? -ss : ss;
    *c = (q == 1 || q == 2) ? -cc : cc;
}

// EXACT: sin/cos rounded once from f64 (what the rendered images are pinned to); otherwise sincos_short.
template <bool EXACT = true>
__device__ inline void pose_setup(const CamConst& cc, const float* C, float theta, float px, float py,
                                  PoseCam* pc) {
    const float up[3] = {0.0f, 1.0f, 0.0f};
    float v[3] = {0.0f - C[0], 0.0f - C[1], 0.0f - C[2]};
    if (EXACT) normalize3(v[0], v[1], v[2], pc->az, &pc->nv); else normalize3_fast(v[0], v[1], v[2], pc->az, &pc->nv);
    float w[3];
    cross3(up, pc->az, w);
    if (EXACT) normalize3(w[0], w[1], w[2], pc->ax, &pc->nw); else normalize3_fast(w[0], w[1], w[2], pc->ax, &pc->nw);
    float u[3];
    cross3(pc->az, pc->ax, u);
    if (EXACT) normalize3(u[0], u[1], u[2], pc->ay, &pc->nu); else normalize3_fast(u[0], u[1], u[2], pc->ay, &pc->nu);
    pc->degenerate = 0;
    if (fabsf(pc->ax[0]) <= 5e-3f && fabsf(pc->ax[1]) <= 5e-3f && fabsf(pc->ax[2]) <= 5e-3f) {
        float r[3], nr;
        cross3(pc->ay, pc->az, r);
        normalize3(r[0], r[1], r[2], pc->ax, &nr);
        pc->degenerate = 1;
    }
    float sn, cs;
    if (EXACT) sincos32(theta, &sn, &cs); else sincos_short(theta, &sn, &cs);
    pc->sn = sn; pc->cs = cs;
    for (int i = 0; i < 3; ++i) {
        float r0 = pc->ax[i], r1 = pc->ay[i], r2 = pc->az[i];
        pc->R[i * 3 + 0] = r0 * cs + r1 * sn;
        pc->R[i * 3 + 1] = r1 * cs - r0 * sn;
        pc->R[i * 3 + 2] = r2;
    }
    for (int j = 0; j < 3; ++j)
        pc->T[j] = -((pc->R[0 + j] * C[0] + pc->R[3 + j] * C[1]) + pc->R[6 + j] * C[2]);
    if (EXACT) { pc->bx = (cc.half_w - px) / cc.s; pc->by = (cc.half_h - py) / cc.s; }
    else { const float is = __builtin_amdgcn_rcpf(cc.s); pc->bx = (cc.half_w - px) * is; pc->by = (cc.half_h - py) * is; }
    pc->C[0] = C[0]; pc->C[1] = C[1]; pc->C[2] = C[2];
}

// World vertex -> view coordinates and NDC xy.
__device__ inline void project_vertex(const CamConst& cc, const float* R, const float* T, float bx, float by,
                                      float X0, float X1, float X2, float* Xv, float* xn, float* yn) {
    Xv[0] = ((X0 * R[0] + X1 * R[3]) + X2 * R[6]) + T[0];
    Xv[1] = ((X0 * R[1] + X1 * R[4]) + X2 * R[7]) + T[1];
    Xv[2] = ((X0 * R[2] + X1 * R[5]) + X2 * R[8]) + T[2];
    *xn = (cc.a * Xv[0]) / Xv[2] + bx;
    *yn = (cc.a * Xv[1]) / Xv[2] + by;
}

// NDC coordinate of pixel index i (already flipped: row H-1-yi / column W-1-xi) on an axis of
// S1 pixels whose other axis has S2 pixels (PyTorch3D PixToNonSquareNdc).
__device__ inline float ndc_range(int S1, int S2) {
    float r = 2.0f;
    if (S1 > S2) r = ((float)S1 * r) / (float)S2;
    return r;
}
__device__ inline float pix_to_ndc(int i, int S1, int S2) {
    float r = ndc_range(S1, S2);
    float off = r / 2.0f;
    return -off + (r * (float)i + off) / (float)S1;
}

__device__ inline float edge_fn(float px, float py, float ax, float ay, float bx, float by) {
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

struct FaceVerts {
    float x0, y0, z0, x1, y1, z1, x2, y2, z2;
};

// Barycentrics (perspective corrected) and depth of pixel (xf,yf) in a face.
__device__ inline void face_eval(const FaceVerts& f, float xf, float yf, float* b, float* pz) {
    float area = edge_fn(f.x2, f.y2, f.x0, f.y0, f.x1, f.y1) + kEps;
    float w0 = edge_fn(xf, yf, f.x1, f.y1, f.x2, f.y2) / area;
    float w1 = edge_fn(xf, yf, f.x2, f.y2, f.x0, f.y0) / area;
    float w2 = edge_fn(xf, yf, f.x0, f.y0, f.x1, f.y1) / area;
    float t0 = w0 * f.z1 * f.z2;
    float t1 = f.z0 * w1 * f.z2;
    float t2 = f.z0 * f.z1 * w2;
    float den = (t0 + t1) + t2;
    den = den > kEps ? den : (den != den ? den : kEps);
    b[0] = t0 / den; b[1] = t1 / den; b[2] = t2 / den;
    *pz = (b[0] * f.z0 + b[1] * f.z1) + b[2] * f.z2;
}

__device__ inline unsigned long long depth_key(float pz, int face) {
    return ((unsigned long long)__float_as_uint(pz) << 32) | (unsigned int)face;
}

// Clamp a float pixel index to [0, n-1]; NaN maps to `nan_val`.
__device__ inline int clamp_index(float v, int n, int nan_val) {
    if (!(v == v)) return nan_val;
    if (v <= 0.0f) return 0;
    if (v >= (float)(n - 1)) return n - 1;
    return (int)v;
}

// Full coverage test of one (face, pixel) pair and depth-buffer update: atomic min on
// (depth bits << 32 | face), i.e. nearest depth wins and ties go to the smaller face index, exactly as
// a serial scan over the faces in order would leave it.
__device__ inline void eval_and_min(int face, const FaceVerts& f, int p, float xf, float yf, unsigned long long* zbuf) {
    float b[3], pz;
    face_eval(f, xf, yf, b, &pz);
    if (pz < 0.0f) return;
    if (!(b[0] > 0.0f && b[1] > 0.0f && b[2] > 0.0f)) return;
    atomicMin(&zbuf[p], depth_key(pz, face));
}

// ------------------------------------------------------------------------------------------
// Backward.  Per covered pixel the gradient of the loss w.r.t. the three barycentrics (gb) is
// pushed through the perspective correction and the edge functions to the NDC position and depth
// of the face's three vertices, through the projection to their view coordinates, and folded
// into 14 running sums: M = sum Xv^T (x) gXv (9), S = sum gXv (3), gbx, gby.  With
// X_view = X.R + T and R orthonormal, dL/dR = R.(M - T^T (x) S) and dL/dT = S, so no per-vertex
// gradient buffer and no atomics are needed.
// ------------------------------------------------------------------------------------------
struct PixelGrad {
    float M[9];
    float S[3];
    float gbx, gby;
};

__device__ inline void pixel_backward(const CamConst& cc, const FaceVerts& f, const float* Xv0, const float* Xv1,
                                      const float* Xv2, float xf, float yf, const float* gb, PixelGrad* acc) {
    // recompute the forward intermediates
    float area = edge_fn(f.x2, f.y2, f.x0, f.y0, f.x1, f.y1) + kEps;
    float e0 = edge_fn(xf, yf, f.x1, f.y1, f.x2, f.y2);
    float e1 = edge_fn(xf, yf, f.x2, f.y2, f.x0, f.y0);
    float e2 = edge_fn(xf, yf, f.x0, f.y0, f.x1, f.y1);
    float w0 = e0 / area, w1 = e1 / area, w2 = e2 / area;
    float t0 = w0 * f.z1 * f.z2, t1 = f.z0 * w1 * f.z2, t2 = f.z0 * f.z1 * w2;
    float den_raw = (t0 + t1) + t2;
    bool clamped = !(den_raw > kEps);
    float den = clamped ? kEps : den_raw;
    float b0 = t0 / den, b1 = t1 / den, b2 = t2 / den;
    // b_i = t_i / den
    float inv_den = 1.0f / den;
    float dot = gb[0] * b0 + gb[1] * b1 + gb[2] * b2;
    float gden = clamped ? 0.0f : -dot * inv_den;
    float gt0 = gb[0] * inv_den + gden, gt1 = gb[1] * inv_den + gden, gt2 = gb[2] * inv_den + gden;
    // t0 = w0 z1 z2, t1 = z0 w1 z2, t2 = z0 z1 w2
    float gw0 = gt0 * f.z1 * f.z2, gw1 = gt1 * f.z0 * f.z2, gw2 = gt2 * f.z0 * f.z1;
    float gz0 = gt1 * w1 * f.z2 + gt2 * f.z1 * w2;
    float gz1 = gt0 * w0 * f.z2 + gt2 * f.z0 * w2;
    float gz2 = gt0 * w0 * f.z1 + gt1 * f.z0 * w1;
    // w_i = e_i / area
    float inv_area = 1.0f / area;
    float ge0 = gw0 * inv_area, ge1 = gw1 * inv_area, ge2 = gw2 * inv_area;
    float garea = -(gw0 * w0 + gw1 * w1 + gw2 * w2) * inv_area;
    // edge(p; a, b): d/da.x = p.y - b.y, d/da.y = b.x - p.x, d/db.x = -(p.y - a.y), d/db.y = p.x - a.x,
    //               d/dp.x = b.y - a.y, d/dp.y = -(b.x - a.x)
    float gx0 = 0.f, gy0 = 0.f, gx1 = 0.f, gy1 = 0.f, gx2 = 0.f, gy2 = 0.f;
    // e0 = edge(p; v1, v2)
    gx1 += ge0 * (yf - f.y2); gy1 += ge0 * (f.x2 - xf); gx2 += ge0 * -(yf - f.y1); gy2 += ge0 * (xf - f.x1);
    // e1 = edge(p; v2, v0)
    gx2 += ge1 * (yf - f.y0); gy2 += ge1 * (f.x0 - xf); gx0 += ge1 * -(yf - f.y2); gy0 += ge1 * (xf - f.x2);
    // e2 = edge(p; v0, v1)
    gx0 += ge2 * (yf - f.y1); gy0 += ge2 * (f.x1 - xf); gx1 += ge2 * -(yf - f.y0); gy1 += ge2 * (xf - f.x0);
    // area = edge(v2; v0, v1) + eps
    gx0 += garea * (f.y2 - f.y1); gy0 += garea * (f.x1 - f.x2);
    gx1 += garea * -(f.y2 - f.y0); gy1 += garea * (f.x2 - f.x0);
    gx2 += garea * (f.y1 - f.y0); gy2 += garea * -(f.x1 - f.x0);
    // x_ndc = a X / Z + bx, y_ndc = a Y / Z + by, depth = Z
    const float* Xs[3] = {Xv0, Xv1, Xv2};
    float gxs[3] = {gx0, gx1, gx2}, gys[3] = {gy0, gy1, gy2}, gzs[3] = {gz0, gz1, gz2};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float* X = Xs[i];
        float iz = 1.0f / X[2];
        float gX = gxs[i] * cc.a * iz;
        float gY = gys[i] * cc.a * iz;
        float gZ = gzs[i] - (gX * X[0] + gY * X[1]) * iz;
        acc->M[0] += X[0] * gX; acc->M[1] += X[0] * gY; acc->M[2] += X[0] * gZ;
        acc->M[3] += X[1] * gX; acc->M[4] += X[1] * gY; acc->M[5] += X[1] * gZ;
        acc->M[6] += X[2] * gX; acc->M[7] += X[2] * gY; acc->M[8] += X[2] * gZ;
        acc->S[0] += gX; acc->S[1] += gY; acc->S[2] += gZ;
        acc->gbx += gxs[i];
        acc->gby += gys[i];
    }
}

// Forward + backward of one covered pixel for the optimisation loop, straight-line (values only need the
// loss tolerance, not bit parity: reciprocals are the 1-ulp hardware form and fma contraction is allowed).
// v_i = (x_ndc, y_ndc, Z, 1/Z) of the face's vertices, all in front of the camera.  With pixel-relative
// vectors (dx_i, dy_i) = pixel - v_i the edge functions are cross products, e0 = dy1 dx2 - dx1 dy2 ...;
// u_i = e_i z_j z_k, b_i = u_i / sum(u) (the 1/area factor cancels and is never formed),
// obj = sum b_i d_i.  Backward: gu_i = g (d_i - obj) / sum(u).  The view-space gradient of a vertex is
// accumulated through q = Z * gXv = (a gx, a gy, Z gz - a (gx xi + gy eta)) with (xi, eta) = (X/Z, Y/Z):
// M += (xi, eta, 1)^T (x) q,  S += q / Z.  weight = 1/normaliser for a real entry, 0 for a padding
// entry (which must still be a valid covered pixel so that every intermediate is finite).
#pragma clang fp contract(fast)
__device__ inline float pixel_fused(float a, float ia, float bx, float by, const float4& v0, const float4& v1,
                                    const float4& v2, float xf, float yf, const float* d, float cl, float weight,
                                    PixelGrad* acc) {
    const float dx0 = xf - v0.x, dy0 = yf - v0.y, dx1 = xf - v1.x, dy1 = yf - v1.y, dx2 = xf - v2.x, dy2 = yf - v2.y;
    const float e0 = dy1 * dx2 - dx1 * dy2, e1 = dy2 * dx0 - dx2 * dy0, e2 = dy0 * dx1 - dx0 * dy1;
    const float zt0 = v1.z * v2.z, zt1 = v0.z * v2.z, zt2 = v0.z * v1.z;
    const float u0 = e0 * zt0, u1 = e1 * zt1, u2 = e2 * zt2;
    const float inv_su = __builtin_amdgcn_rcpf((u0 + u1) + u2);
    const float obj = ((u0 * d[0] + u1 * d[1]) + u2 * d[2]) * inv_su;
    // sub-gradient of max(obj, cl); a NaN score (or clutter score) poisons the gradient as it does in autograd
    const float mk = obj > cl ? 1.0f : (obj == cl ? 0.5f : (__builtin_isunordered(obj, cl) ? obj + cl : 0.0f));
    const float G = -mk * weight * inv_su;
    const float gu0 = G * (d[0] - obj), gu1 = G * (d[1] - obj), gu2 = G * (d[2] - obj);
    const float ge0 = gu0 * zt0, ge1 = gu1 * zt1, ge2 = gu2 * zt2;
    const float h0 = gu0 * e0, h1 = gu1 * e1, h2 = gu2 * e2;
    const float gz[3] = {h1 * v2.z + h2 * v1.z, h0 * v2.z + h2 * v0.z, h0 * v1.z + h1 * v0.z};
    const float gx[3] = {ge2 * dy1 - ge1 * dy2, ge0 * dy2 - ge2 * dy0, ge1 * dy0 - ge0 * dy1};
    const float gy[3] = {ge1 * dx2 - ge2 * dx1, ge2 * dx0 - ge0 * dx2, ge0 * dx1 - ge1 * dx0};
    const float4 vs[3] = {v0, v1, v2};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float xi = (vs[i].x - bx) * ia, eta = (vs[i].y - by) * ia;
        const float qx = gx[i] * a, qy = gy[i] * a;
        const float qz = vs[i].z * gz[i] - (qx * xi + qy * eta);
        acc->M[0] += xi * qx;  acc->M[1] += xi * qy;  acc->M[2] += xi * qz;
        acc->M[3] += eta * qx; acc->M[4] += eta * qy; acc->M[5] += eta * qz;
        acc->M[6] += qx;       acc->M[7] += qy;       acc->M[8] += qz;
        acc->S[0] += qx * vs[i].w; acc->S[1] += qy * vs[i].w; acc->S[2] += qz * vs[i].w;
        acc->gbx += gx[i];
        acc->gby += gy[i];
    }
    return obj;
}
#pragma clang fp contract(off)

// Forward only: the interpolated score sum b_i d_i of a pixel centre inside (or on the rim of) a face.
#pragma clang fp contract(fast)
__device__ inline float pixel_score(const float4& v0, const float4& v1, const float4& v2, float xf, float yf, const float* d) {
    const float dx0 = xf - v0.x, dy0 = yf - v0.y, dx1 = xf - v1.x, dy1 = yf - v1.y, dx2 = xf - v2.x, dy2 = yf - v2.y;
    const float e0 = dy1 * dx2 - dx1 * dy2, e1 = dy2 * dx0 - dx2 * dy0, e2 = dy0 * dx1 - dx0 * dy1;
    const float u0 = e0 * (v1.z * v2.z), u1 = e1 * (v0.z * v2.z), u2 = e2 * (v0.z * v1.z);
    return ((u0 * d[0] + u1 * d[1]) + u2 * d[2]) * __builtin_amdgcn_rcpf((u0 + u1) + u2);
}
#pragma clang fp contract(off)

// From the reduced sums to the gradient of (C[3], theta, px, py).  EXACT = false (optimisation loop): the
// divisions by the three norms and by the NDC scale are reciprocals.
template <bool EXACT = true>
__device__ inline void pose_backward(const CamConst& cc, const PoseCam& pc, const float* M, const float* S, float gbx,
                                     float gby, float* gC, float* gtheta, float* gpx, float* gpy) {
    // dL/dR = R (M - T^T (x) S),  dL/dT = S
    float A[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) A[i * 3 + j] = M[i * 3 + j] - pc.T[i] * S[j];
    float GR[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            GR[i * 3 + j] = pc.R[i * 3 + 0] * A[0 + j] + pc.R[i * 3 + 1] * A[3 + j] + pc.R[i * 3 + 2] * A[6 + j];
    float gc[3] = {0.f, 0.f, 0.f};
    // T_j = -sum_i R_ij C_i
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            GR[i * 3 + j] += -pc.C[i] * S[j];
            gc[i] += -pc.R[i * 3 + j] * S[j];
        }
    const float inu = EXACT ? 1.0f / pc.nu : __builtin_amdgcn_rcpf(pc.nu), inw = EXACT ? 1.0f / pc.nw : __builtin_amdgcn_rcpf(pc.nw),
                inv_ = EXACT ? 1.0f / pc.nv : __builtin_amdgcn_rcpf(pc.nv), is = EXACT ? 1.0f / cc.s : __builtin_amdgcn_rcpf(cc.s);
    // R = R0 Rz(theta): R[i][0] = r0 c + r1 s, R[i][1] = r1 c - r0 s, R[i][2] = r2
    float gth = 0.f;
    float G0[9];   // dL/dR0, R0 = [ax ay az] as columns
    for (int i = 0; i < 3; ++i) {
        float r0 = pc.ax[i], r1 = pc.ay[i];
        float g0 = GR[i * 3 + 0], g1 = GR[i * 3 + 1];
        G0[i * 3 + 0] = g0 * pc.cs - g1 * pc.sn;
        G0[i * 3 + 1] = g0 * pc.sn + g1 * pc.cs;
        G0[i * 3 + 2] = GR[i * 3 + 2];
        gth += g0 * (-r0 * pc.sn + r1 * pc.cs) + g1 * (-r1 * pc.sn - r0 * pc.cs);
    }
    float Gx[3] = {G0[0], G0[3], G0[6]}, Gy[3] = {G0[1], G0[4], G0[7]}, Gz[3] = {G0[2], G0[5], G0[8]};
    // y = u/|u|, u = z x x
    {
        float d = Gy[0] * pc.ay[0] + Gy[1] * pc.ay[1] + Gy[2] * pc.ay[2];
        float Gu[3];
        for (int i = 0; i < 3; ++i) {
            const float t_ = pc.nu > kNormEps ? (Gy[i] - pc.ay[i] * d) : Gy[i];
            Gu[i] = EXACT ? t_ / pc.nu : t_ * inu;
        }
        float t1[3], t2[3];
        cross3(pc.ax, Gu, t1);   // dL/dz += x x Gu
        cross3(Gu, pc.az, t2);   // dL/dx += Gu x z
        for (int i = 0; i < 3; ++i) { Gz[i] += t1[i]; Gx[i] += t2[i]; }
    }
    // x = w/|w|, w = up x z
    {
        const float up[3] = {0.f, 1.f, 0.f};
        float d = Gx[0] * pc.ax[0] + Gx[1] * pc.ax[1] + Gx[2] * pc.ax[2];
        float Gw[3];
        for (int i = 0; i < 3; ++i) {
            const float t_ = pc.nw > kNormEps ? (Gx[i] - pc.ax[i] * d) : Gx[i];
            Gw[i] = EXACT ? t_ / pc.nw : t_ * inw;
        }
        float t[3];
        cross3(Gw, up, t);       // dL/dz += Gw x up
        for (int i = 0; i < 3; ++i) Gz[i] += t[i];
    }
    // z = v/|v|, v = -C
    {
        float d = Gz[0] * pc.az[0] + Gz[1] * pc.az[1] + Gz[2] * pc.az[2];
        for (int i = 0; i < 3; ++i) {
            const float t_ = pc.nv > kNormEps ? (Gz[i] - pc.az[i] * d) : Gz[i];
            const float Gv = EXACT ? t_ / pc.nv : t_ * inv_;
            gc[i] += -Gv;
        }
    }
    gC[0] = gc[0]; gC[1] = gc[1]; gC[2] = gc[2];
    *gtheta = gth;
    *gpx = EXACT ? -gbx / cc.s : -gbx * is;
    *gpy = EXACT ? -gby / cc.s : -gby * is;
}

}  // namespace nemo
