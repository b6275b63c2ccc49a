// Pose-grid scoring on the gfx950 matrix cores.
//
//   obj[n, b, p] = sum_c pre[n, c, p] * feat[b, c, p]          (batch_solve_pose.py:128-164)
//   partial[chunk, b, n] = sum over the chunk's pixels of max(obj, clu[b, p])   (or obj without clutter)
//
// For one pixel this is a [N x C] . [C x B] product: poses are the rows, images the columns, channels
// the reduction.  Both operands are kept pixel-major in MFMA fragment order as two-piece f16 values
// (split_f16.h), so a wave reads each fragment as one contiguous kilobyte:
//
//   packed grid   [p][row block of 32 poses][k-step of 16 channels][piece][lane]   built once per grid
//   feature frags [p][column block of 32 images][k-step][piece][lane]              built once per batch
//
// A wave owns one column block (32 images) and walks the pixels of its chunk; for every pixel and row
// block it runs 3 x C/16 v_mfma_f32_32x32x16_f16 and folds the 32 x 32 result tile into its running
// sums with one max and one add per element.  The per-chunk sums go to the same partial buffer layout
// as the VALU kernel's, so the deterministic finalize (sum over chunks, first-minimum arg-min) is shared.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "split_f16.h"

namespace nemo {

constexpr int GM_CHUNK = 32;        // pixels per workgroup
constexpr int GM_RBG = 5;           // row blocks (of 32 poses) accumulated per launch slice

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

// ---------------------------------------------------------------------------------------------
// scoring
// ---------------------------------------------------------------------------------------------
template <int NKS>
__global__ __launch_bounds__(256) void grid_mfma_kernel(const uint4* __restrict__ packed, const uint4* __restrict__ featp,
                                                        const float* __restrict__ clu, int N, int B, int HW, int nrb, int ncb,
                                                        float unscale, float* __restrict__ partial) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunk = blockIdx.x, cb = blockIdx.y * 4 + wave, rb0 = blockIdx.z * GM_RBG;
    if (cb >= ncb) return;                               // no barriers in this kernel
    const int nr = min(GM_RBG, nrb - rb0);
    const int b = cb * 32 + (lane & 31), lk = lane >> 5;
    const bool bvalid = b < B;
    const int p_begin = chunk * GM_CHUNK, p_end = min(p_begin + GM_CHUNK, HW);

    f32x16 lacc[GM_RBG];
#pragma unroll
    for (int r = 0; r < GM_RBG; ++r) lacc[r] = (f32x16){0};

    for (int p = p_begin; p < p_end; ++p) {
        f16x8 bh[NKS], bl[NKS];
        const uint4* bp = featp + (((size_t)p * ncb + cb) * NKS) * 2 * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) { bh[ks] = as_frag(bp[(ks * 2 + 0) * 64]); bl[ks] = as_frag(bp[(ks * 2 + 1) * 64]); }
        const float cl = (clu && bvalid) ? clu[(size_t)b * HW + p] : -__builtin_inff();
#pragma unroll
        for (int r = 0; r < GM_RBG; ++r) {
            if (r < nr) {
                const uint4* ap = packed + (((size_t)p * nrb + rb0 + r) * NKS) * 2 * 64 + lane;
                f32x16 S = {0};
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const f16x8 ah = as_frag(ap[(ks * 2 + 0) * 64]), al = as_frag(ap[(ks * 2 + 1) * 64]);
                    S = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[ks], S, 0, 0, 0);
                    S = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[ks], S, 0, 0, 0);
                    S = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[ks], S, 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) lacc[r][i] += fmaxf(S[i] * unscale, cl);      // cl = -inf without a clutter map
            }
        }
    }
    if (!bvalid) return;
    float* out = partial + ((size_t)chunk * B + b) * N;
#pragma unroll
    for (int r = 0; r < GM_RBG; ++r) {
        if (r < nr) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int n = (rb0 + r) * 32 + (i & 3) + 8 * (i >> 2) + 4 * lk;      // C/D layout of the 32x32 tile
                if (n < N) out[n] = lacc[r][i];
            }
        }
    }
}

bool grid_mfma_supported(int C) { return C == 16 || C == 32 || C == 64 || C == 128; }
int grid_chunks(int HW) { return (HW + GM_CHUNK - 1) / GM_CHUNK; }
size_t grid_pack_bytes(int N, int C, int HW) { return (size_t)HW * ((N + 31) / 32) * (C / 16) * 2 * 64 * 16; }
size_t grid_feat_bytes(int B, int C, int HW) { return (size_t)HW * ((B + 31) / 32) * (C / 16) * 2 * 64 * 16; }

int launch_grid_pack(const float* pre, void* packed, int N, int C, int HW, hipStream_t st) {
    const int nrb = (N + 31) / 32;
    const long long total = (long long)HW * nrb * (C / 16) * 64;
    const long long nwg = (total + 255) / 256;
    if (nwg > 0x7FFFFFFFLL) return -1;
    hipLaunchKernelGGL(grid_pack_kernel, dim3((unsigned)nwg), dim3(256), 0, st, pre, (uint4*)packed, N, C, HW, nrb);
    return 0;
}

template <int NKS>
static void launch_mfma_nks(const void* packed, const void* featp, const float* clu, int N, int B, int HW, float* partial,
                            hipStream_t st) {
    const int nrb = (N + 31) / 32, ncb = (B + 31) / 32;
    dim3 grid(grid_chunks(HW), (ncb + 3) / 4, (nrb + GM_RBG - 1) / GM_RBG);
    hipLaunchKernelGGL(grid_mfma_kernel<NKS>, grid, dim3(256), 0, st, (const uint4*)packed, (const uint4*)featp, clu, N, B, HW, nrb,
                       ncb, 1.0f / (SG_FEAT_SCALE * SG_FEAT_SCALE), partial);
}

// featp: grid_feat_bytes(B, C, HW) bytes of scratch; partial: grid_chunks(HW) * B * N floats.
int launch_grid_mfma(const void* packed, const float* feat, const float* clu, int N, int B, int C, int HW, void* featp,
                     float* partial, hipStream_t st) {
    const int ncb = (B + 31) / 32, nrb = (N + 31) / 32;
    if (ncb > 65535 || (ncb + 3) / 4 > 65535 || (nrb + GM_RBG - 1) / GM_RBG > 65535) return -1;
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

}  // namespace nemo
