// Pose-grid scoring on the gfx950 matrix cores.
//
//   obj[n, b, p] = sum_c pre[n, c, p] * feat[b, c, p]          (batch_solve_pose.py:128-164)
//   partial[chunk, b, n] = sum over the chunk's pixels of max(obj, clu[b, p])   (or obj without clutter)
//
// For one pixel this is a [N x C] . [C x B] product: poses are the rows, images the columns, channels
// the reduction.  Both operands are kept pixel-major in MFMA fragment order as two-piece f16 values
// (split_f16.h), so a wave reads each fragment as one contiguous kilobyte:
//
//   packed grid   [p][row block of 32 poses][k-step of 16 channels][piece][lane]   built once per grid,
//                 followed by one bit per pixel: some pose covers it (grid_mask_kernel)
//   feature frags [p][column block of 32 images][k-step][piece][lane]              built once per batch
//
// A wave owns one column block (32 images) and walks the pixels of its chunk; for every pixel and row
// block it runs 3 x C/16 v_mfma_f32_32x32x16_f16 and folds the 32 x 32 result tile into its running
// sums with one max and one add per element.  The eight waves of a workgroup share the pose fragments
// through LDS, so the packed grid is read once per 256 images.  The per-chunk sums go to the same partial buffer layout
// as the VALU kernel's, so the deterministic finalize (sum over chunks, first-minimum arg-min) is shared.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "split_f16.h"
#include "resample.h"

namespace nemo {

constexpr int GM_CHUNK = 32;        // pixels per workgroup
constexpr int GM_RBG = 6;           // row blocks (of 32 poses) accumulated per launch slice at most (16 result registers each; nine spill)

// ---------------------------------------------------------------------------------------------
// one-off: pre-rendered maps [N, C, HW] f32 -> packed fragments
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grid_pack_kernel(const float* __restrict__ pre, uint4* __restrict__ packed, int N, int C,
                                                        int HW, int nrb) {
    const int nks = C / 16;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)HW * nrb * nks * 64;
    if (t >= total) return;
    const int p = (int)(t % HW);                       // consecutive threads read consecutive pixels
    const int r = (int)(t / HW);
    const int lane = r & 63, ks = (r >> 6) % nks, rb = (r >> 6) / nks;
    const int n = rb * 32 + (lane & 31), k0 = ks * 16 + 8 * (lane >> 5);
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = n < N ? scale_clamp(pre[((size_t)n * C + k0 + j) * HW + p]) : 0.0f;
    uint4 pcs[SG_NP];
    split8(x, pcs);
    const size_t base = (((size_t)p * nrb + rb) * nks + ks) * 2;
    packed[(base + 0) * 64 + lane] = pcs[0];
    packed[(base + 1) * 64 + lane] = pcs[1];
}

// Behind the packed fragments: one bit per pixel = some pose of the grid has a non-zero value there (any channel).  A pixel
// no pose covers contributes max(0, clutter) to every loss (its pose fragments are all zero), which the scoring adds
// without reading anything.  One workgroup per 64 pixels, its four waves share the rows (pose x channel) of the maps.
__global__ __launch_bounds__(256) void grid_mask_kernel(const float* __restrict__ pre, unsigned int* __restrict__ mask, int rows, int HW) {
    __shared__ unsigned int found[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = blockIdx.x * 64 + lane;
    if (threadIdx.x < 2) found[threadIdx.x] = 0u;
    __syncthreads();
    bool any = false;
    if (p < HW) {
        for (int r0 = wave; r0 < rows; r0 += 4 * 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = r0 + 4 * u;
                if (r < rows) any |= pre[(size_t)r * HW + p] != 0.0f;        // true for NaN as well
            }
            if (__all(any)) break;
        }
    }
    const unsigned long long m = __ballot(any);
    if (lane == 0) { atomicOr(&found[0], (unsigned int)m); atomicOr(&found[1], (unsigned int)(m >> 32)); }
    __syncthreads();
    if (threadIdx.x < 2 && blockIdx.x * 2 + threadIdx.x < (unsigned)((HW + 31) / 32)) mask[blockIdx.x * 2 + threadIdx.x] = found[threadIdx.x];
}

// ---------------------------------------------------------------------------------------------
// per batch: feature maps [B, C, HW] f32 -> fragments.  One workgroup per (32 pixels, k-step, column
// block): the 32 images x 16 channels x 32 pixels brick goes through LDS so that both the reads (along
// pixels) and the writes (one kilobyte per pixel and piece) are contiguous.
// ---------------------------------------------------------------------------------------------
constexpr int GF_PX = 32;
constexpr int GF_IMG_STRIDE = 16 * (GF_PX + 1) + 1;     // odd: the 32 images of a fragment read hit 32 different banks

__global__ __launch_bounds__(256) void grid_feat_kernel(const float* __restrict__ feat, uint4* __restrict__ featp, int B, int C,
                                                        int HW, int ncb) {
    __shared__ float brick[32 * GF_IMG_STRIDE];
    const int nks = C / 16;
    const int p0 = blockIdx.x * GF_PX, ks = blockIdx.y, cb = blockIdx.z;
    const int px = threadIdx.x & 31;
    const int p = p0 + px;
    const int pc = p < HW ? p : HW - 1;
    // 64 brick rows per thread, 16 loads in flight at a time (addresses clamped, values masked)
#pragma unroll 1
    for (int r0 = threadIdx.x >> 5; r0 < 32 * 16; r0 += 8 * 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int r = r0 + 8 * u, img = r >> 4, ch = r & 15;
            const int b = min(cb * 32 + img, B - 1);
            v[u] = feat[((size_t)b * C + ks * 16 + ch) * HW + pc];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int r = r0 + 8 * u, img = r >> 4, ch = r & 15;
            brick[img * GF_IMG_STRIDE + ch * (GF_PX + 1) + px] = (cb * 32 + img < B && p < HW) ? v[u] : 0.0f;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int img = lane & 31, kq = lane >> 5;
    for (int q = wave; q < GF_PX; q += 4) {
        const int p = p0 + q;
        if (p >= HW) break;
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = scale_clamp(brick[img * GF_IMG_STRIDE + (8 * kq + j) * (GF_PX + 1) + q]);
        uint4 pcs[SG_NP];
        split8(x, pcs);
        const size_t base = (((size_t)p * ncb + cb) * nks + ks) * 2;
        featp[(base + 0) * 64 + lane] = pcs[0];
        featp[(base + 1) * 64 + lane] = pcs[1];
    }
}

// The same pass also scoring the clutter vectors (batch_solve_pose.py:214-225: 1x1 convolution with the K clutter
// vectors, max over K): the feature maps are read once for both.  One workgroup per (32 pixels, column block) walks
// the k-steps; thread t keeps the running dot products of image t / 8 at pixels t % 8 + 8 j, channels in ascending
// order with one fused multiply-add each, i.e. the very sums of clutter_kernel.
template <int K>
__global__ __launch_bounds__(256) void grid_feat_clutter_kernel(const float* __restrict__ feat, const float* __restrict__ clutter,
                                                                uint4* __restrict__ featp, float* __restrict__ out_clu, int B, int C,
                                                                int HW, int ncb) {
    __shared__ float brick[32 * GF_IMG_STRIDE];
    __shared__ float cl[K * 128];
    const int nks = C / 16;
    const int p0 = blockIdx.x * GF_PX, cb = blockIdx.y;
    const int px = threadIdx.x & 31;
    const int p = p0 + px;
    const int pc = p < HW ? p : HW - 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cimg = threadIdx.x >> 3, cpx = threadIdx.x & 7;
    for (int i = threadIdx.x; i < K * C; i += 256) cl[i] = clutter[i];
    float acc[4][K];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < K; ++k) acc[j][k] = 0.0f;
#pragma unroll 1
    for (int ks = 0; ks < nks; ++ks) {
#pragma unroll 1
        for (int r0 = threadIdx.x >> 5; r0 < 32 * 16; r0 += 8 * 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int r = r0 + 8 * u, img = r >> 4, ch = r & 15;
                const int b = min(cb * 32 + img, B - 1);
                v[u] = feat[((size_t)b * C + ks * 16 + ch) * HW + pc];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int r = r0 + 8 * u, img = r >> 4, ch = r & 15;
                brick[img * GF_IMG_STRIDE + ch * (GF_PX + 1) + px] = (cb * 32 + img < B && p < HW) ? v[u] : 0.0f;
            }
        }
        __syncthreads();
        {
            const int img = lane & 31, kq = lane >> 5;
            for (int q = wave; q < GF_PX; q += 4) {
                const int pq = p0 + q;
                if (pq >= HW) break;
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = scale_clamp(brick[img * GF_IMG_STRIDE + (8 * kq + j) * (GF_PX + 1) + q]);
                uint4 pcs[SG_NP];
                split8(x, pcs);
                const size_t base = (((size_t)pq * ncb + cb) * nks + ks) * 2;
                featp[(base + 0) * 64 + lane] = pcs[0];
                featp[(base + 1) * 64 + lane] = pcs[1];
            }
        }
#pragma unroll
        for (int ch = 0; ch < 16; ++ch) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float f = brick[cimg * GF_IMG_STRIDE + ch * (GF_PX + 1) + cpx + 8 * j];
#pragma unroll
                for (int k = 0; k < K; ++k) acc[j][k] = __builtin_fmaf(f, cl[k * C + ks * 16 + ch], acc[j][k]);
            }
        }
        __syncthreads();
    }
    const int b = cb * 32 + cimg;
    if (b < B) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pj = p0 + cpx + 8 * j;
            float best = acc[j][0];
#pragma unroll
            for (int k = 1; k < K; ++k) best = max_nan(best, acc[j][k]);
            if (pj < HW) out_clu[(size_t)b * HW + pj] = best;
        }
    }
}

// The same for re-centred copies of a batch (6-D initialisation, batch_solve_pose.py:264-295): virtual image
// bv is source image bv % Bsrc resampled with its own re-centring parameters (resample.h).  The resampled maps
// are never written: the four taps of a (virtual image, pixel) are worked out once per brick row group and
// the brick is filled with the bilinear values.
struct RecentreArgs {
    const float* dist_src; const float* dist_tgt; const float* p_src; const float* p_tgt;     // per virtual image
    int Bsrc, h1, w1, h0, w0, border;
};

// One workgroup per (32 pixels, column block of 32 virtual images) walks the k-steps: a thread owns four (image, pixel)
// pairs, works out their taps once and samples the 16 channels of every k-step with them (the taps do not depend on the
// channel: eight times fewer tap computations than a workgroup per k-step, and this pass was bound by them).
__global__ __launch_bounds__(256) void grid_feat_recentred_kernel(const float* __restrict__ feat, uint4* __restrict__ featp, int B,
                                                                  int C, int ncb, RecentreArgs ra) {
    __shared__ float brick[32 * GF_IMG_STRIDE];
    const int nks = C / 16, HW = ra.h0 * ra.w0, HW1 = ra.h1 * ra.w1;
    const int p0 = blockIdx.x * GF_PX, cb = blockIdx.y;
    const int px = threadIdx.x & 31, ig = threadIdx.x >> 5;
    const int p = p0 + px;
    const int pc = p < HW ? p : HW - 1;
    const int ty = pc / ra.w0, tx = pc - ty * ra.w0;
    AlignTaps taps[4];
    const float* src[4];
    bool live[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int img = ig + 8 * q;
        const int b = min(cb * 32 + img, B - 1);
        taps[q] = align_taps(ra.dist_src[b], ra.dist_tgt[b], ra.p_src[2 * b], ra.p_src[2 * b + 1], ra.p_tgt[2 * b], ra.p_tgt[2 * b + 1],
                             ra.h1, ra.w1, ra.h0, ra.w0, ra.border, tx, ty);
        src[q] = feat + (size_t)(b % ra.Bsrc) * C * HW1;
        live[q] = cb * 32 + img < B && p < HW;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll 1
    for (int ks = 0; ks < nks; ++ks) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float* sp = src[q] + (size_t)(ks * 16) * HW1;
            float v[16];
#pragma unroll
            for (int ch = 0; ch < 16; ++ch) v[ch] = align_sample(sp + (size_t)ch * HW1, taps[q]);
#pragma unroll
            for (int ch = 0; ch < 16; ++ch) brick[(ig + 8 * q) * GF_IMG_STRIDE + ch * (GF_PX + 1) + px] = live[q] ? v[ch] : 0.0f;
        }
        __syncthreads();
        {
            const int img = lane & 31, kq = lane >> 5;
            for (int q = wave; q < GF_PX; q += 4) {
                const int pq = p0 + q;
                if (pq >= HW) break;
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = scale_clamp(brick[img * GF_IMG_STRIDE + (8 * kq + j) * (GF_PX + 1) + q]);
                uint4 pcs[SG_NP];
                split8(x, pcs);
                const size_t base = (((size_t)pq * ncb + cb) * nks + ks) * 2;
                featp[(base + 0) * 64 + lane] = pcs[0];
                featp[(base + 1) * 64 + lane] = pcs[1];
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// scoring
// ---------------------------------------------------------------------------------------------
// Workgroup = 8 waves = 8 column blocks (256 images) of one pixel chunk.  The pose fragments of a
// (pixel, row block) stage - NKS * 2 KB, the same for every column block - are fetched once by the whole
// workgroup and staged in LDS (double buffered); every wave multiplies them with its own image fragments,
// which it holds in registers for the pixel.  The fetch of stage s+2 is issued before the MFMAs of stage
// s, into one of two register sets, and written to LDS after the MFMAs of stage s+1.
constexpr int GM_WAVES = 8;

// NR = row blocks of this launch slice (compile time, so that the running sums, the LDS buffer and the
// register set of every stage are addressed statically: the stage loop is unrolled over two pixels).
template <int NKS, int NR>
__global__ __launch_bounds__(64 * GM_WAVES) void grid_mfma_kernel(const uint4* __restrict__ packed, const uint4* __restrict__ featp,
                                                                 const float* __restrict__ clu, int N, int B, int HW, int nrb,
                                                                 int ncb, int rb0, float unscale, float* __restrict__ partial,
                                                                 const unsigned int* __restrict__ covered) {
    static_assert(GM_CHUNK == 32, "one word of the coverage mask per chunk");
    constexpr int STAGE_U4 = NKS * 2 * 64;                       // uint4 per stage
    constexpr int PER_THREAD = (STAGE_U4 + 64 * GM_WAVES - 1) / (64 * GM_WAVES);
    __shared__ uint4 abuf[2][STAGE_U4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Workgroups that share a pixel chunk (the same pose fragments, different image blocks) get block ids 8 apart:
    // workgroups are dealt round-robin over the 8 XCDs, so they run at the same time behind the same L2 and the packed
    // grid comes from HBM once per chunk instead of once per workgroup.  id = ((chunk / 8) * ncbg + cbg) * 8 + chunk % 8.
    const int ncbg = (ncb + GM_WAVES - 1) / GM_WAVES;
    const int chunk = ((int)blockIdx.x / 8 / ncbg) * 8 + (int)blockIdx.x % 8, cbg = ((int)blockIdx.x / 8) % ncbg;
    if (chunk * GM_CHUNK >= HW) return;          // padding of the last group of eight chunks (uniform over the workgroup)
    const int cb = cbg * GM_WAVES + wave;
    const bool wave_on = cb < ncb;                               // idle waves still help with the staging
    const int cbc = wave_on ? cb : ncb - 1;
    const int b = cbc * 32 + (lane & 31), lk = lane >> 5;
    const bool bvalid = wave_on && b < B;
    const int p_begin = chunk * GM_CHUNK, p_end = min(p_begin + GM_CHUNK, HW);
    // The pixels of the chunk that go through the matrix cores, as a bit mask (uniform over the workgroup): all of them, or,
    // with a clutter map, the ones some pose of the grid covers.  Without a clutter map nothing is skipped: a NaN feature at
    // a pixel no pose covers must still reach the sums (0 x NaN), with one it does through its clutter score.
    const int npx_all = p_end - p_begin;
    unsigned int px_mask = npx_all >= 32 ? 0xFFFFFFFFu : ((1u << npx_all) - 1u);
    if (covered && clu) px_mask &= (unsigned int)__builtin_amdgcn_readfirstlane((int)covered[chunk]);
    const int nstage = __popc(px_mask) * NR;
    unsigned int fetch_left = px_mask, work_left = px_mask;     // pixels not yet fetched / multiplied (their lowest bit is next)
    int fetch_px = 0, work_px = -1;                             // offsets in the chunk of the pixel being fetched / multiplied

    struct Staged { uint4 v[PER_THREAD]; };                      // one thread's share of a stage, held in registers
    Staged qa = {}, qb = {};                                     // register sets 0 and 1
    auto fetch = [&](int st) {                                   // called for st = 0, 1, 2, ... in this order
        if (st % NR == 0) { fetch_px = __ffs((int)fetch_left) - 1; fetch_left &= fetch_left - 1u; }
        const uint4* src = packed + (((size_t)(p_begin + fetch_px) * nrb + rb0 + st % NR) * NKS) * 2 * 64;
        Staged q = {};
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) { const int e = tid + i * 64 * GM_WAVES; if (e < STAGE_U4) q.v[i] = src[e]; }
        return q;
    };
    auto park = [&](int buf, const Staged& q) {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) { const int e = tid + i * 64 * GM_WAVES; if (e < STAGE_U4) abuf[buf][e] = q.v[i]; }
    };

    f32x16 lacc[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) lacc[r] = (f32x16){0};
    f16x8 bh[NKS], bl[NKS];
    float cl = -__builtin_inff();
    // pixels [from, to) of the chunk that no pose covers: what the matrix instructions would return for them is +0
    auto uncovered = [&](int from, int to) {
        for (int q = from; q < to; ++q) {
            const float cq = bvalid ? clu[(size_t)b * HW + p_begin + q] : -__builtin_inff();
            const float v = max_nan(0.0f * unscale, cq);
#pragma unroll
            for (int r = 0; r < NR; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) lacc[r][i] += v;
        }
    };

    if (nstage > 0) { qa = fetch(0); park(0, qa); }
    if (nstage > 1) qb = fetch(1);
    __syncthreads();
    for (int st0 = 0; st0 < nstage; st0 += 2 * NR) {
#pragma unroll
        for (int u = 0; u < 2 * NR; ++u) {
            const int st = st0 + u;
            if (st >= nstage) continue;                          // workgroup-uniform (odd pixel count: the tail of the last pair)
            const int r = u % NR, buf = u & 1;                   // compile-time after unrolling (2 * NR is even)
            // stage st is in LDS buffer `buf`; register set 1 - buf holds stage st + 1; set `buf` is free
            if (st + 2 < nstage) { if (buf == 0) qa = fetch(st + 2); else qb = fetch(st + 2); }
            if (r == 0) {                                        // new pixel: this wave's image fragments
                const int nxt = __ffs((int)work_left) - 1;
                work_left &= work_left - 1u;
                if (nxt > work_px + 1) uncovered(work_px + 1, nxt);      // the sums take the pixels in ascending order
                work_px = nxt;
                const int p = p_begin + nxt;
                const uint4* bp = featp + (((size_t)p * ncb + cbc) * NKS) * 2 * 64 + lane;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) { bh[ks] = as_frag(bp[(ks * 2 + 0) * 64]); bl[ks] = as_frag(bp[(ks * 2 + 1) * 64]); }
                cl = (clu && bvalid) ? clu[(size_t)b * HW + p] : -__builtin_inff();
            }
            f32x16 S = {0};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const f16x8 ah = as_frag(abuf[buf][(ks * 2 + 0) * 64 + lane]), al = as_frag(abuf[buf][(ks * 2 + 1) * 64 + lane]);
                S = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[ks], S, 0, 0, 0);
                S = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[ks], S, 0, 0, 0);
                S = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[ks], S, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) lacc[r][i] += max_nan(S[i] * unscale, cl);        // cl = -inf without a clutter map
            if (st + 1 < nstage) { if (buf == 0) park(1, qb); else park(0, qa); }
            __syncthreads();
        }
    }
    if (work_px + 1 < npx_all) uncovered(work_px + 1, npx_all);
    if (!bvalid) return;
    float* out = partial + ((size_t)chunk * B + b) * N;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int n = (rb0 + r) * 32 + (i & 3) + 8 * (i >> 2) + 4 * lk;          // C/D layout of the 32x32 tile
            if (n < N) out[n] = lacc[r][i];
        }
    }
}

bool grid_mfma_supported(int C) { return C == 16 || C == 32 || C == 64 || C == 128; }
int grid_chunks(int HW) { return (HW + GM_CHUNK - 1) / GM_CHUNK; }
static size_t grid_pack_fragment_bytes(int N, int C, int HW) { return (size_t)HW * ((N + 31) / 32) * (C / 16) * 2 * 64 * 16; }
static size_t grid_mask_bytes(int HW) { return (((size_t)(HW + 31) / 32) * 4 + 15) & ~(size_t)15; }
size_t grid_pack_bytes(int N, int C, int HW) { return grid_pack_fragment_bytes(N, C, HW) + grid_mask_bytes(HW); }
size_t grid_feat_bytes(int B, int C, int HW) { return (size_t)HW * ((B + 31) / 32) * (C / 16) * 2 * 64 * 16; }

int launch_grid_pack(const float* pre, void* packed, int N, int C, int HW, hipStream_t st) {
    const int nrb = (N + 31) / 32;
    const long long total = (long long)HW * nrb * (C / 16) * 64;
    const long long nwg = (total + 255) / 256;
    if (nwg > 0x7FFFFFFFLL) return -1;
    hipLaunchKernelGGL(grid_pack_kernel, dim3((unsigned)nwg), dim3(256), 0, st, pre, (uint4*)packed, N, C, HW, nrb);
    hipLaunchKernelGGL(grid_mask_kernel, dim3((unsigned)((HW + 63) / 64)), dim3(256), 0, st, pre,
                       (unsigned int*)((char*)packed + grid_pack_fragment_bytes(N, C, HW)), N * C, HW);
    return 0;
}

template <int NKS, int NR>
static void launch_mfma_slice(const void* packed, const void* featp, const float* clu, int N, int B, int HW, int rb0, float* partial,
                              hipStream_t st) {
    const int nrb = (N + 31) / 32, ncb = (B + 31) / 32;
    const dim3 grid((unsigned)(((grid_chunks(HW) + 7) / 8) * 8 * ((ncb + GM_WAVES - 1) / GM_WAVES)));
    const unsigned int* covered = (const unsigned int*)((const char*)packed + grid_pack_fragment_bytes(N, NKS * 16, HW));
    hipLaunchKernelGGL((grid_mfma_kernel<NKS, NR>), grid, dim3(64 * GM_WAVES), 0, st, (const uint4*)packed, (const uint4*)featp, clu, N,
                       B, HW, nrb, ncb, rb0, 1.0f / (SG_FEAT_SCALE * SG_FEAT_SCALE), partial, covered);
}

// All row blocks of the grid, GM_RBG at a time (one launch per slice; 144 poses are one slice).
template <int NKS>
static void launch_mfma_nks(const void* packed, const void* featp, const float* clu, int N, int B, int HW, float* partial,
                            hipStream_t st) {
    const int nrb = (N + 31) / 32;
    for (int rb0 = 0; rb0 < nrb; rb0 += GM_RBG) {
        switch (nrb - rb0 < GM_RBG ? nrb - rb0 : GM_RBG) {
            case 1: launch_mfma_slice<NKS, 1>(packed, featp, clu, N, B, HW, rb0, partial, st); break;
            case 2: launch_mfma_slice<NKS, 2>(packed, featp, clu, N, B, HW, rb0, partial, st); break;
            case 3: launch_mfma_slice<NKS, 3>(packed, featp, clu, N, B, HW, rb0, partial, st); break;
            case 4: launch_mfma_slice<NKS, 4>(packed, featp, clu, N, B, HW, rb0, partial, st); break;
            case 5: launch_mfma_slice<NKS, 5>(packed, featp, clu, N, B, HW, rb0, partial, st); break;
            default: launch_mfma_slice<NKS, 6>(packed, featp, clu, N, B, HW, rb0, partial, st); break;
        }
    }
}

constexpr int kFusedClutterMax = 4;         // clutter vectors the fused fragment + clutter pass takes

template <int NKS>
static void launch_mfma_nks(const void* packed, const void* featp, const float* clu, int N, int B, int HW, float* partial, hipStream_t st);

// Clutter scores [B, HW] of K <= kFusedClutterMax clutter vectors and the pose-grid losses from one read of the maps.
// Returns 1 when the shape is not one the fused pass takes (the caller then runs the two steps one after the other).
int launch_grid_mfma_clutter(const void* packed, const float* feat, const float* clutter, int K, int N, int B, int C, int HW,
                             float* out_clu, void* featp, float* partial, hipStream_t st) {
    const int ncb = (B + 31) / 32;
    if (K < 1 || K > kFusedClutterMax || C > 128) return 1;
    if (ncb > 65535) return -1;
    const dim3 grid((HW + GF_PX - 1) / GF_PX, ncb);
    uint4* fp = (uint4*)featp;
    switch (K) {
        case 1: hipLaunchKernelGGL(grid_feat_clutter_kernel<1>, grid, dim3(256), 0, st, feat, clutter, fp, out_clu, B, C, HW, ncb); break;
        case 2: hipLaunchKernelGGL(grid_feat_clutter_kernel<2>, grid, dim3(256), 0, st, feat, clutter, fp, out_clu, B, C, HW, ncb); break;
        case 3: hipLaunchKernelGGL(grid_feat_clutter_kernel<3>, grid, dim3(256), 0, st, feat, clutter, fp, out_clu, B, C, HW, ncb); break;
        default: hipLaunchKernelGGL(grid_feat_clutter_kernel<4>, grid, dim3(256), 0, st, feat, clutter, fp, out_clu, B, C, HW, ncb); break;
    }
    switch (C / 16) {
        case 1: launch_mfma_nks<1>(packed, featp, out_clu, N, B, HW, partial, st); break;
        case 2: launch_mfma_nks<2>(packed, featp, out_clu, N, B, HW, partial, st); break;
        case 4: launch_mfma_nks<4>(packed, featp, out_clu, N, B, HW, partial, st); break;
        case 8: launch_mfma_nks<8>(packed, featp, out_clu, N, B, HW, partial, st); break;
        default: return -2;
    }
    return 0;
}

// Scoring alone, on fragments built by an earlier call for the same maps.
int launch_grid_mfma_fragments(const void* packed, const void* featp, const float* clu, int N, int B, int C, int HW, float* partial,
                               hipStream_t st) {
    if ((B + 31) / 32 > 65535) return -1;
    switch (C / 16) {
        case 1: launch_mfma_nks<1>(packed, featp, clu, N, B, HW, partial, st); break;
        case 2: launch_mfma_nks<2>(packed, featp, clu, N, B, HW, partial, st); break;
        case 4: launch_mfma_nks<4>(packed, featp, clu, N, B, HW, partial, st); break;
        case 8: launch_mfma_nks<8>(packed, featp, clu, N, B, HW, partial, st); break;
        default: return -2;
    }
    return 0;
}

// featp: grid_feat_bytes(B, C, HW) bytes of scratch; partial: grid_chunks(HW) * B * N floats.
int launch_grid_mfma(const void* packed, const float* feat, const float* clu, int N, int B, int C, int HW, void* featp,
                     float* partial, hipStream_t st) {
    const int ncb = (B + 31) / 32, nrb = (N + 31) / 32;
    if (ncb > 65535) return -1;
    (void)nrb;
    hipLaunchKernelGGL(grid_feat_kernel, dim3((HW + GF_PX - 1) / GF_PX, C / 16, ncb), dim3(256), 0, st, feat, (uint4*)featp, B, C, HW,
                       ncb);
    switch (C / 16) {
        case 1: launch_mfma_nks<1>(packed, featp, clu, N, B, HW, partial, st); break;
        case 2: launch_mfma_nks<2>(packed, featp, clu, N, B, HW, partial, st); break;
        case 4: launch_mfma_nks<4>(packed, featp, clu, N, B, HW, partial, st); break;
        case 8: launch_mfma_nks<8>(packed, featp, clu, N, B, HW, partial, st); break;
        default: return -2;
    }
    return 0;
}

// The same with the feature fragments built from re-centred copies: B virtual images of Bsrc source maps
// (h1 x w1), scored on the h0 x w0 grid maps; clu [B, h0 * w0] already re-centred (or null).
int launch_grid_mfma_recentred(const void* packed, const float* feat_src, const float* clu, int N, int B, int C, int Bsrc, int h1, int w1,
                               int h0, int w0, int border, const float* dist_src, const float* dist_tgt, const float* p_src,
                               const float* p_tgt, void* featp, float* partial, hipStream_t st) {
    const int ncb = (B + 31) / 32, HW = h0 * w0;
    if (ncb > 65535) return -1;
    RecentreArgs ra;
    ra.dist_src = dist_src; ra.dist_tgt = dist_tgt; ra.p_src = p_src; ra.p_tgt = p_tgt;
    ra.Bsrc = Bsrc; ra.h1 = h1; ra.w1 = w1; ra.h0 = h0; ra.w0 = w0; ra.border = border;
    hipLaunchKernelGGL(grid_feat_recentred_kernel, dim3((HW + GF_PX - 1) / GF_PX, ncb), dim3(256), 0, st, feat_src, (uint4*)featp, B, C,
                       ncb, ra);
    switch (C / 16) {
        case 1: launch_mfma_nks<1>(packed, featp, clu, N, B, HW, partial, st); break;
        case 2: launch_mfma_nks<2>(packed, featp, clu, N, B, HW, partial, st); break;
        case 4: launch_mfma_nks<4>(packed, featp, clu, N, B, HW, partial, st); break;
        case 8: launch_mfma_nks<8>(packed, featp, clu, N, B, HW, partial, st); break;
        default: return -2;
    }
    return 0;
}

}  // namespace nemo
