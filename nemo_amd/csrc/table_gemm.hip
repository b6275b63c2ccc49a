// Vertex x pixel similarity table on the gfx950 matrix cores.
//
//   table[b, v, p] = sum_c memory[v, c] * feat[b, c, p]
//
// the reference's kpt_score_map (nemo/models/solve_pose.py:120-124) and the matmul of
// NeMo.fast_inference (nemo/models/nemo.py:350).
//
// Two kernels:
//
//  * split_table_kernel (C = 16, 32, 64 or 128): every f32 operand, scaled by a power of two into the
//    comfortable part of the f16 range, is written as a sum of two f16 pieces x = h + l (l the
//    round-to-nearest f16 of what h left over: 22 significant bits together), and the product is
//    accumulated in f32 by v_mfma_f32_32x32x16_f16 over the three piece pairs hh, hl, lh.  The dropped
//    pair ll is below f32 resolution; measured against an f64 product the result has the rms error of
//    an f32 matmul.  The f16 pipe has 16 x the rate of the f32 pipe, so the three products cost 3/16 of
//    the f32 matrix-core time and the kernel is bound by writing the table to HBM.  The vertex bank is
//    split once per nemo_mesh_set_memory into MFMA fragment order; the feature tile of a workgroup
//    (128 pixels x C channels) is split once into LDS and reused for every vertex block.  Features are
//    unit vectors per pixel by the reference's contract (F.normalize in the backbone wrapper); a
//    component of magnitude above 255.8 would leave the f16 range after scaling and is clamped.
//
//  * table_gemm_kernel (any C): v_mfma_f32_32x32x2_f32, an exact f32 fma chain over c.  Both
//    operands are read "k-major": A = memory^T [C, V] (the transpose is kept by the mesh handle)
//    and B = feat[b] [C, HW], so the lanes of every fragment load touch consecutive addresses and
//    no LDS staging is needed.  One wave owns a 64 x 64 output tile (2 x 2 MFMA tiles); a workgroup
//    is 4 waves = 128 vertices x 128 pixels; workgroups that share a pixel tile of one image are
//    consecutive in the per-XCD order so its feature columns stay in that XCD's L2.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "split_f16.h"


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

    // C/D layout of the 32x32 tile: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5):
    // registers 4j..4j+3 of a lane are four consecutive vertices of one pixel, one 16-byte element of
    // the table ([b][pixel tile][vertex group of 4][128 pixels][4]); columns past HW hold zeros and
    // the vertices that pad the last group are written as zeros.
    const int vg_n = (V + 3) >> 2;
    float4* tb = reinterpret_cast<float4*>(table) + ((size_t)b * npt + pt) * vg_n * TG_BN;
    const int ca = wc * 64 + lr, cb = ca + 32;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int g0 = (v0 + 8 * j + 4 * lk) >> 2, g1 = g0 + 8;
        if (g0 < vg_n) {
            tb[(size_t)g0 * TG_BN + ca] = make_float4(acc00[4 * j], acc00[4 * j + 1], acc00[4 * j + 2], acc00[4 * j + 3]);
            tb[(size_t)g0 * TG_BN + cb] = make_float4(acc01[4 * j], acc01[4 * j + 1], acc01[4 * j + 2], acc01[4 * j + 3]);
        }
        if (g1 < vg_n) {
            tb[(size_t)g1 * TG_BN + ca] = make_float4(acc10[4 * j], acc10[4 * j + 1], acc10[4 * j + 2], acc10[4 * j + 3]);
            tb[(size_t)g1 * TG_BN + cb] = make_float4(acc11[4 * j], acc11[4 * j + 1], acc11[4 * j + 2], acc11[4 * j + 3]);
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


// ---------------------------------------------------------------------------------------------
// split-f16 path
// ---------------------------------------------------------------------------------------------
int split_bank_blocks(int V);
constexpr int SG_PT = 128;          // pixels per workgroup
constexpr int SG_THREADS = 256;     // 4 waves, each a 32-vertex block x all 128 pixels at a time

__global__ void bank_absmax_kernel(const float* __restrict__ memory, size_t n, BankMeta* meta) {
    float m = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float a = fabsf(memory[i]);
        if (a < __builtin_inff()) m = fmaxf(m, a);       // NaN / inf do not pick the scale
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(&meta->absmax_bits, __float_as_uint(m));
}

// Vertex bank -> MFMA A-operand fragments: frag[((piece * nvb + vb) * nks + ks) * 64 + lane] holds
// rows vb*32 + (lane & 31), channels ks*16 + 8*(lane >> 5) .. +7; rows >= V are zero.  The bank is
// scaled by the power of two that brings its largest magnitude into [128, 256).
__global__ void split_bank_kernel(const float* __restrict__ memory, BankMeta* meta, uint4* __restrict__ frag, int V, int C,
                                  int nvb) {
    const float amax = __uint_as_float(meta->absmax_bits);
    int e = 0;
    if (amax > 0.0f) { (void)frexpf(amax, &e); e = 8 - e; }            // amax * 2^e in [128, 256)
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
    const float scale = ldexpf(1.0f, e);
    const int nks = C / 16;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { meta->scale = scale; meta->unscale = ldexpf(1.0f, -e) / SG_FEAT_SCALE; }
    if (i >= nvb * nks * 64) return;
    const int lane = i & 63, ks = (i >> 6) % nks, vb = (i >> 6) / nks;
    const int row = vb * 32 + (lane & 31), k0 = ks * 16 + 8 * (lane >> 5);
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = row < V ? memory[(size_t)row * C + k0 + j] * scale : 0.0f;
    uint4 pcs[SG_NP];
    split8(x, pcs);
#pragma unroll
    for (int pc = 0; pc < SG_NP; ++pc) frag[((size_t)(pc * nvb + vb) * nks + ks) * 64 + lane] = pcs[pc];
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Stores register quads [J0, J1) of the four accumulator tiles of one wave (32 vertices x 4 blocks of 32
// pixels) into its pixel tile of the table (tile: [vertex group][128 pixels] 16-byte elements).  A quad is
// four consecutive vertices of one pixel = one element, so a store instruction writes two contiguous
// 512-byte runs (vertex groups 2j and 2j + 1 of the block), and the 16 stores of a wave-tile fill 16
// contiguous kilobytes.  CHECK = false: the caller guarantees that all 32 rows exist, and the stores are
// straight-line code.  Plain stores: measured faster than non-temporal ones for this pattern.
template <int J0, int J1, bool CHECK>
__device__ inline void store_quads(f32x4* __restrict__ tile, const f32x16 (&c)[4], int v0, int vg_n, int lk, int px, float us) {
#pragma unroll
    for (int j = J0; j < J1; ++j) {
        const int g = (v0 >> 2) + 2 * j + lk;
        if (!CHECK || g < vg_n) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                tile[g * SG_PT + t * 32 + px] = (f32x4){c[t][4 * j] * us, c[t][4 * j + 1] * us, c[t][4 * j + 2] * us, c[t][4 * j + 3] * us};
        }
    }
}

template <int NKS>
struct SplitWave {
    const uint4* bank_frag; const char* bbase; f32x4* tile;
    size_t apiece; int vg_n, lane, lk, px; float unscale;
    static constexpr int C = NKS * 16;
    static constexpr int row_bytes = C * 2 + 16;      // +16: consecutive pixels start 4 banks apart
    static constexpr int piece_bytes = SG_PT * row_bytes;
    static constexpr int QPK = (4 + NKS - 1) / NKS;   // result register quads stored per k-step

    template <int KS>
    __device__ inline void store_slice(const f32x16 (&o)[4], int v0) const {
        constexpr int J0 = KS * QPK < 4 ? KS * QPK : 4, J1 = (KS + 1) * QPK < 4 ? (KS + 1) * QPK : 4;
        store_quads<J0, J1, false>(tile, o, v0, vg_n, lk, px, unscale);
    }

    // One 32 x 128 result tile: vertex block vb against the whole pixel tile, so a bank fragment is fetched
    // by exactly one wave of the workgroup.  The A fragments are requested two k-steps ahead of
    // their MFMAs.  STORE_PREV: the 16 stores of the previous tile (o, all of it inside the table) are
    // spread over the k-steps, each slice issued after the A request, so that waiting for the fragments
    // never waits for a store (gfx9 counts loads and stores in one in-order counter).
    template <bool STORE_PREV>
    __device__ inline void tile_mma(int vb, f32x16 (&acc)[4], const f32x16 (&o)[4], int prev_v0) const {
        const uint4* ap = bank_frag + ((size_t)vb * NKS) * 64 + lane;
        constexpr int PF = NKS > 1 ? 2 : 1;      // k-steps of A fragments in flight: an L2 round trip outlasts one step's MFMAs
        uint4 na[PF][SG_NP];
#pragma unroll
        for (int d = 0; d < PF; ++d)
#pragma unroll
            for (int pc = 0; pc < SG_NP; ++pc) na[d][pc] = ap[pc * apiece + (size_t)(d < NKS ? d : 0) * 64];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            f16x8 a[SG_NP], b[4][SG_NP];
#pragma unroll
            for (int pc = 0; pc < SG_NP; ++pc) a[pc] = as_frag(na[ks % PF][pc]);
            if (ks + PF < NKS) {
#pragma unroll
                for (int pc = 0; pc < SG_NP; ++pc) na[ks % PF][pc] = ap[pc * apiece + (size_t)(ks + PF) * 64];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int pc = 0; pc < SG_NP; ++pc)
                    b[t][pc] = as_frag(*reinterpret_cast<const uint4*>(bbase + pc * piece_bytes + t * 32 * row_bytes + ks * 32));
            if (STORE_PREV) {
                switch (ks) {          // ks is a compile-time constant after unrolling
                    case 0: store_slice<0>(o, prev_v0); break;
                    case 1: store_slice<1>(o, prev_v0); break;
                    case 2: store_slice<2>(o, prev_v0); break;
                    case 3: store_slice<3>(o, prev_v0); break;
                    case 4: store_slice<4>(o, prev_v0); break;
                    case 5: store_slice<5>(o, prev_v0); break;
                    case 6: store_slice<6>(o, prev_v0); break;
                    default: store_slice<7>(o, prev_v0); break;
                }
            }
            // piece pairs in increasing weight: (l,h) (h,l) (h,h)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int ia = q == 0 ? 1 : 0, ib = q == 1 ? 1 : 0;
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ia], b[t][ib], acc[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);        // keep the k-steps apart: registers for one step's operands only
        }
    }
};

// NKS = C / 16 k-steps, fully unrolled.
template <int NKS>
__global__ __launch_bounds__(SG_THREADS) void split_table_kernel(const BankMeta* __restrict__ meta,
                                                                    const uint4* __restrict__ bank_frag,
                                                                    const float* __restrict__ feat, float* __restrict__ table,
                                                                    int V, int nvb, int HW, int npt) {
    extern __shared__ __align__(16) char sg_lds[];
    using W = SplitWave<NKS>;
    constexpr int C = W::C, row_bytes = W::row_bytes, piece_bytes = W::piece_bytes;
    const int pt = blockIdx.x % npt, b = blockIdx.x / npt;
    const int p0 = pt * SG_PT;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float* fb = feat + (size_t)b * C * HW;

    // ---- phase 1: feature tile -> f16 pieces in LDS, [piece][pixel][channel] ----
    {
        const int pl = t & (SG_PT - 1);
        const int p = p0 + pl;
        const bool pok = p < HW;
        const float* col = fb + (pok ? p : HW - 1);
        constexpr int KG_STEP = SG_THREADS / SG_PT, NKG = C / 8;
        constexpr int NIT = (NKG + KG_STEP - 1) / KG_STEP;
        float x[NIT][8];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int kg = t / SG_PT + it * KG_STEP;
            const int kgc = kg < NKG ? kg : NKG - 1;
#pragma unroll
            for (int j = 0; j < 8; ++j) x[it][j] = col[(size_t)(kgc * 8 + j) * HW];
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int kg = t / SG_PT + it * KG_STEP;
            if (kg < NKG) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    x[it][j] = pok ? scale_clamp(x[it][j]) : 0.0f;
                uint4 pcs[SG_NP];
                split8(x[it], pcs);
#pragma unroll
                for (int pc = 0; pc < SG_NP; ++pc)
                    *reinterpret_cast<uint4*>(sg_lds + pc * piece_bytes + pl * row_bytes + kg * 16) = pcs[pc];
            }
        }
    }
    __syncthreads();

    // ---- phase 2: this wave's vertex blocks (wave, wave + 4, ...) against the resident pixel tile ----
    W w;
    w.unscale = meta->unscale;
    w.bank_frag = bank_frag; w.lane = lane; w.lk = lane >> 5; w.px = lane & 31;
    w.bbase = sg_lds + (lane & 31) * row_bytes + w.lk * 16;
    w.vg_n = (V + 3) >> 2;
    w.tile = reinterpret_cast<f32x4*>(table) + ((size_t)b * npt + pt) * w.vg_n * SG_PT;
    w.apiece = (size_t)nvb * NKS * 64;

    constexpr int WAVES = SG_THREADS / 64;
    int vb = wave;
    if (vb >= nvb) return;
    // The vertex list ends less than 64 rows before the end of the last block pair, so only the last block
    // of a wave can hang over it, and the last block is stored by the checked epilogue: every block stored
    // inside the pipelined loop is complete.  Pixel columns past HW exist in the table (zero features give zeros).
    f32x16 o[4] = {};
    {
        f32x16 acc[4] = {};
        w.template tile_mma<false>(vb, acc, o, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = acc[k];
    }
    int prev_v0 = vb * 32;
    for (;;) {
        vb += WAVES;
        if (vb >= nvb) break;
        f32x16 acc[4] = {};
        w.template tile_mma<true>(vb, acc, o, prev_v0);
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = acc[k];
        prev_v0 = vb * 32;
    }
    store_quads<0, 4, true>(w.tile, o, prev_v0, w.vg_n, w.lk, w.px, w.unscale);
}

// ---------------------------------------------------------------------------------------------
// NeMo.fast_inference without the table (nemo/models/nemo.py:327-360):
//   out[b] = mean over the pixels p of a box of max( max_v <memory[v], feat[b,:,p]>, clu[b,p] )
// The same product as split_table_kernel (same operands, same instruction sequence per entry, so every vertex x
// pixel value is what the table would hold), but a wave keeps only the running maximum over the vertices of each of
// its pixels: 16 of a block's 32 rows sit in one lane's result registers, the other 16 in lane + 32.  The four waves'
// maxima meet in LDS; nothing but one partial sum per (image, pixel tile) leaves the workgroup.
// ---------------------------------------------------------------------------------------------
template <int NKS>
__global__ __launch_bounds__(SG_THREADS) void vertex_max_kernel(const BankMeta* __restrict__ meta, const uint4* __restrict__ bank_frag,
                                                                const float* __restrict__ feat, const float* __restrict__ clu,
                                                                const int32_t* __restrict__ boxes, int V, int nvb, int H, int Wd, int npt,
                                                                float* __restrict__ partial) {
    extern __shared__ __align__(16) char sg_lds[];
    using W = SplitWave<NKS>;
    constexpr int C = W::C, row_bytes = W::row_bytes, piece_bytes = W::piece_bytes;
    constexpr int WAVES = SG_THREADS / 64;
    __shared__ float wmax[WAVES][SG_PT];
    __shared__ float wsum[WAVES];
    const int HW = H * Wd;
    const int pt = blockIdx.x % npt, b = blockIdx.x / npt;
    const int p0 = pt * SG_PT;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int r0 = 0, r1 = H, c0 = 0, c1 = Wd;
    if (boxes) { r0 = max(boxes[4 * b], 0); r1 = min(boxes[4 * b + 1], H); c0 = max(boxes[4 * b + 2], 0); c1 = min(boxes[4 * b + 3], Wd); }
    // a pixel tile that lies outside the box rows contributes nothing (uniform over the workgroup)
    if (r1 <= r0 || c1 <= c0 || p0 >= r1 * Wd || p0 + SG_PT <= r0 * Wd) {
        if (t == 0) partial[(size_t)b * npt + pt] = 0.0f;
        return;
    }
    const float* fb = feat + (size_t)b * C * HW;
    {       // feature tile -> f16 pieces in LDS, exactly as split_table_kernel's first phase
        const int pl = t & (SG_PT - 1);
        const int p = p0 + pl;
        const bool pok = p < HW;
        const float* col = fb + (pok ? p : HW - 1);
        constexpr int KG_STEP = SG_THREADS / SG_PT, NKG = C / 8;
        constexpr int NIT = (NKG + KG_STEP - 1) / KG_STEP;
        float x[NIT][8];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int kg = t / SG_PT + it * KG_STEP;
            const int kgc = kg < NKG ? kg : NKG - 1;
#pragma unroll
            for (int j = 0; j < 8; ++j) x[it][j] = col[(size_t)(kgc * 8 + j) * HW];
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int kg = t / SG_PT + it * KG_STEP;
            if (kg < NKG) {
#pragma unroll
                for (int j = 0; j < 8; ++j) x[it][j] = pok ? scale_clamp(x[it][j]) : 0.0f;
                uint4 pcs[SG_NP];
                split8(x[it], pcs);
#pragma unroll
                for (int pc = 0; pc < SG_NP; ++pc)
                    *reinterpret_cast<uint4*>(sg_lds + pc * piece_bytes + pl * row_bytes + kg * 16) = pcs[pc];
            }
        }
    }
    __syncthreads();
    W w;
    w.unscale = meta->unscale;
    w.bank_frag = bank_frag; w.lane = lane; w.lk = lane >> 5; w.px = lane & 31;
    w.bbase = sg_lds + (lane & 31) * row_bytes + w.lk * 16;
    w.vg_n = (V + 3) >> 2;
    w.tile = nullptr;
    w.apiece = (size_t)nvb * NKS * 64;
    float best[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) best[k] = -__builtin_inff();
    const f32x16 none[4] = {};
    for (int vb = wave; vb < nvb; vb += WAVES) {
        f32x16 acc[4] = {};
        w.template tile_mma<false>(vb, acc, none, 0);
        // rows of this lane: vb * 32 + (i & 3) + 8 * (i >> 2) + 4 * lk; rows past V are padding (zero bank rows), not vertices
        const int row0 = vb * 32 + 4 * w.lk;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const bool real = row0 + (i & 3) + 8 * (i >> 2) < V;
#pragma unroll
            for (int k = 0; k < 4; ++k) best[k] = real ? max_nan(best[k], acc[k][i]) : best[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float other = __shfl_xor(best[k], 32);
        best[k] = max_nan(best[k], other);
        if (w.lk == 0) wmax[wave][k * 32 + w.px] = best[k];
    }
    __syncthreads();
    float sum = 0.0f;
    if (t < SG_PT) {
        const int p = p0 + t;
        const int y = p / Wd, x = p - y * Wd;
        if (p < HW && y >= r0 && y < r1 && x >= c0 && x < c1) {
            float m = wmax[0][t];
#pragma unroll
            for (int k = 1; k < WAVES; ++k) m = max_nan(m, wmax[k][t]);
            sum = max_nan(m * w.unscale, clu[(size_t)b * HW + p]);          // the scale is a power of two: it commutes with the maximum
        }
    }
    // fixed-order sum over the 128 pixel lanes (two waves), then over the two waves
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane == 0) wsum[wave] = sum;
    __syncthreads();
    if (t == 0) partial[(size_t)b * npt + pt] = wsum[0] + wsum[1];
}

// out[b] = (sum of the image's pixel-tile partial sums, in tile order) / number of box pixels
__global__ void vertex_max_finalize_kernel(const float* __restrict__ partial, const int32_t* __restrict__ boxes, int B, int H, int Wd,
                                           int npt, float* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int r0 = 0, r1 = H, c0 = 0, c1 = Wd;
    if (boxes) { r0 = max(boxes[4 * b], 0); r1 = min(boxes[4 * b + 1], H); c0 = max(boxes[4 * b + 2], 0); c1 = min(boxes[4 * b + 3], Wd); }
    const int np = max(r1 - r0, 0) * max(c1 - c0, 0);
    float s = 0.0f;
    for (int k = 0; k < npt; ++k) s += partial[(size_t)b * npt + k];
    out[b] = s / (float)np;          // an empty box: 0 / 0 = NaN, as the mean of an empty crop is in torch
}

template <int NKS>
static void launch_vertex_max_nks(const void* buf, const float* feat, const float* clu, const int32_t* boxes, int V, int H, int Wd, int B,
                                  int npt, float* partial, hipStream_t st) {
    const BankMeta* meta = (const BankMeta*)buf;
    const uint4* frag = (const uint4*)((const char*)buf + SG_META_BYTES);
    constexpr int lds = SG_NP * SG_PT * (NKS * 16 * 2 + 16);
    static bool attr_done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_done[dev]) {
        (void)hipFuncSetAttribute((const void*)vertex_max_kernel<NKS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done[dev] = true;
    }
    hipLaunchKernelGGL(vertex_max_kernel<NKS>, dim3((unsigned)(npt * B)), dim3(SG_THREADS), lds, st, meta, frag, feat, clu, boxes, V,
                       split_bank_blocks(V), H, Wd, npt, partial);
}

// partial: B * ceil(H * W / 128) floats of scratch.
int launch_vertex_max(const void* frag, const float* feat, const float* clu, const int32_t* boxes, int V, int C, int H, int Wd, int B,
                      float* partial, float* out, hipStream_t st) {
    const int npt = (H * Wd + SG_PT - 1) / SG_PT;
    if ((long long)npt * B > 0x7FFFFFFFLL) return -1;
    switch (C / 16) {
        case 1: launch_vertex_max_nks<1>(frag, feat, clu, boxes, V, H, Wd, B, npt, partial, st); break;
        case 2: launch_vertex_max_nks<2>(frag, feat, clu, boxes, V, H, Wd, B, npt, partial, st); break;
        case 4: launch_vertex_max_nks<4>(frag, feat, clu, boxes, V, H, Wd, B, npt, partial, st); break;
        case 8: launch_vertex_max_nks<8>(frag, feat, clu, boxes, V, H, Wd, B, npt, partial, st); break;
        default: return -2;
    }
    hipLaunchKernelGGL(vertex_max_finalize_kernel, dim3((B + 255) / 256), dim3(256), 0, st, (const float*)partial, boxes, B, H, Wd, npt, out);
    return 0;
}

bool split_table_supported(int C) { return C == 16 || C == 32 || C == 64 || C == 128; }
int split_bank_blocks(int V) { return 2 * ((V + 63) / 64); }                          // 32-row blocks, even
size_t split_bank_bytes(int V, int C) { return SG_META_BYTES + (size_t)SG_NP * split_bank_blocks(V) * 32 * C * 2; }

void launch_split_bank(const float* memory, void* buf, int V, int C, hipStream_t st) {
    BankMeta* meta = (BankMeta*)buf;
    uint4* frag = (uint4*)((char*)buf + SG_META_BYTES);
    const int nvb = split_bank_blocks(V);
    const int n = nvb * (C / 16) * 64;
    (void)hipMemsetAsync(meta, 0, SG_META_BYTES, st);
    hipLaunchKernelGGL(bank_absmax_kernel, dim3(64), dim3(256), 0, st, memory, (size_t)V * C, meta);
    hipLaunchKernelGGL(split_bank_kernel, dim3((n + 255) / 256), dim3(256), 0, st, memory, meta, frag, V, C, nvb);
}

template <int NKS>
static void launch_split_nks(const void* buf, const float* feat, float* table, int V, int HW, int B, int npt, hipStream_t st) {
    const BankMeta* meta = (const BankMeta*)buf;
    const uint4* frag = (const uint4*)((const char*)buf + SG_META_BYTES);
    constexpr int lds = SG_NP * SG_PT * (NKS * 16 * 2 + 16);
    static bool attr_done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_done[dev]) {
        (void)hipFuncSetAttribute((const void*)split_table_kernel<NKS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done[dev] = true;
    }
    hipLaunchKernelGGL(split_table_kernel<NKS>, dim3((unsigned)(npt * B)), dim3(SG_THREADS), lds, st, meta, frag, feat, table, V,
                       split_bank_blocks(V), HW, npt);
}

int launch_split_table(const void* frag, const float* feat, float* table, int V, int C, int HW, int B, hipStream_t st) {
    const int npt = (HW + SG_PT - 1) / SG_PT;
    if ((long long)npt * B > 0x7FFFFFFFLL) return -1;
    switch (C / 16) {
        case 1: launch_split_nks<1>(frag, feat, table, V, HW, B, npt, st); break;
        case 2: launch_split_nks<2>(frag, feat, table, V, HW, B, npt, st); break;
        case 4: launch_split_nks<4>(frag, feat, table, V, HW, B, npt, st); break;
        case 8: launch_split_nks<8>(frag, feat, table, V, HW, B, npt, st); break;
        default: return -2;
    }
    return 0;
}

}  // namespace nemo
