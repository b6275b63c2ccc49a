// Vertex x pixel similarity table on the gfx950 f32 matrix cores.
//
//   table[b, v, p] = sum_c memory[v, c] * feat[b, c, p]
//
// the reference's kpt_score_map (nemo/models/solve_pose.py:120-124) and the matmul of
// NeMo.fast_inference (nemo/models/nemo.py:350).  Both operands are read "k-major":
// A = memory^T [C, V] (the transpose is kept by the mesh handle) and B = feat[b] [C, HW], so the
// lanes of every fragment load touch consecutive addresses and no LDS staging is needed.
// v_mfma_f32_32x32x2_f32 accumulates an exact f32 fma chain over c (k order 0,1,2,...), i.e. the
// same numerics as a scalar float loop.
//
// Tiling: one wave owns a 64 x 64 output tile (2 x 2 MFMA tiles, 64 accumulator registers); a
// workgroup is 4 waves in a 2 x 2 arrangement = 128 vertices x 128 pixels.  Workgroups that share
// a pixel tile of one image are consecutive in the per-XCD order so its feature columns stay in
// that XCD's L2.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace nemo {

constexpr int TG_BM = 128, TG_BN = 128;

__global__ __launch_bounds__(256) void table_gemm_kernel(const float* __restrict__ memT, const float* __restrict__ feat,
                                                         float* __restrict__ table, int V, int C, int HW, int nvt,
                                                         int npt, int nwg) {
    // XCD-aware remap: hardware deals consecutive workgroup ids round-robin over the 8 XCDs; give
    // each XCD a contiguous range of logical tiles (bijective for any nwg).
    const int orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    const int vt = w % nvt;
    const int rest = w / nvt;
    const int pt = rest % npt;
    const int b = rest / npt;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int v0 = vt * TG_BM + wr * 64, p0 = pt * TG_BN + wc * 64;
    const int lr = lane & 31, lk = lane >> 5;

    const float* fb = feat + (size_t)b * C * HW;
    const int va0 = v0 + lr, va1 = v0 + 32 + lr;
    const int pb0 = p0 + lr, pb1 = p0 + 32 + lr;
    const bool va0_ok = va0 < V, va1_ok = va1 < V, pb0_ok = pb0 < HW, pb1_ok = pb1 < HW;
    const int va0c = va0_ok ? va0 : 0, va1c = va1_ok ? va1 : 0, pb0c = pb0_ok ? pb0 : 0, pb1c = pb1_ok ? pb1 : 0;

    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
    // K loop in chunks of 8 channels (4 MFMA k-steps), register double-buffered: the loads of chunk
    // i+1 are in flight while the 16 MFMAs of chunk i issue.
    const float* qa0 = memT + va0c;
    const float* qa1 = memT + va1c;
    const float* qb0 = fb + pb0c;
    const float* qb1 = fb + pb1c;
    float a0[4], a1[4], b0[4], b1[4], na0[4], na1[4], nb0[4], nb1[4];
    const int nfull = C / 8;
    auto load_chunk = [&](int k, float* x0, float* x1, float* y0, float* y1) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t kk = (size_t)(k + 2 * u + lk);
            x0[u] = qa0[kk * V]; x1[u] = qa1[kk * V];
            y0[u] = qb0[kk * HW]; y1[u] = qb1[kk * HW];
        }
    };
    auto mfma_chunk = [&](const float* x0, const float* x1, const float* y0, const float* y1) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float p0 = va0_ok ? x0[u] : 0.0f, p1 = va1_ok ? x1[u] : 0.0f;
            const float q0 = pb0_ok ? y0[u] : 0.0f, q1 = pb1_ok ? y1[u] : 0.0f;
            acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(p0, q0, acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(p0, q1, acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(p1, q0, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(p1, q1, acc11, 0, 0, 0);
        }
    };
    int k = 0;
    if (nfull > 0) {
        load_chunk(0, a0, a1, b0, b1);
        int i = 0;
        for (; i + 2 <= nfull; i += 2) {
            load_chunk((i + 1) * 8, na0, na1, nb0, nb1);
            mfma_chunk(a0, a1, b0, b1);
            if (i + 2 < nfull) load_chunk((i + 2) * 8, a0, a1, b0, b1);
            mfma_chunk(na0, na1, nb0, nb1);
        }
        if (i < nfull) mfma_chunk(a0, a1, b0, b1);
        k = nfull * 8;
    }
    for (; k < C; k += 2) {
        const int kk = k + lk;
        const bool kok = kk < C;
        const int kc = kok ? kk : 0;
        const float x0 = (kok && va0_ok) ? memT[(size_t)kc * V + va0c] : 0.0f;
        const float x1 = (kok && va1_ok) ? memT[(size_t)kc * V + va1c] : 0.0f;
        const float y0 = (kok && pb0_ok) ? fb[(size_t)kc * HW + pb0c] : 0.0f;
        const float y1 = (kok && pb1_ok) ? fb[(size_t)kc * HW + pb1c] : 0.0f;
        acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y0, acc00, 0, 0, 0);
        acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y1, acc01, 0, 0, 0);
        acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y0, acc10, 0, 0, 0);
        acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y1, acc11, 0, 0, 0);
    }

    // C/D layout of the 32x32 tile: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float* tb = table + (size_t)b * V * HW;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * lk;
        const int r0 = v0 + row, r1 = v0 + 32 + row;
        if (r0 < V) {
            if (pb0_ok) tb[(size_t)r0 * HW + pb0] = acc00[reg];
            if (pb1_ok) tb[(size_t)r0 * HW + pb1] = acc01[reg];
        }
        if (r1 < V) {
            if (pb0_ok) tb[(size_t)r1 * HW + pb0] = acc10[reg];
            if (pb1_ok) tb[(size_t)r1 * HW + pb1] = acc11[reg];
        }
    }
}

int launch_table_gemm(const float* memT, const float* feat, float* table, int V, int C, int HW, int B, hipStream_t st) {
    const int nvt = (V + TG_BM - 1) / TG_BM, npt = (HW + TG_BN - 1) / TG_BN;
    const long long nwg = (long long)nvt * npt * B;
    if (nwg > 0x7FFFFFFFLL) return -1;
    hipLaunchKernelGGL(table_gemm_kernel, dim3((unsigned)nwg), dim3(256), 0, st, memT, feat, table, V, C, HW, nvt, npt,
                       (int)nwg);
    return 0;
}

}  // namespace nemo
