// Two-piece f16 representation of f32 operands for the gfx950 matrix cores.
//
// A value x, scaled by a power of two into the comfortable part of the f16 range, is written as
// x = h + l with h = f16(x) and l = f16(x - h) (round to nearest even; x - h is exact in f32), 22
// significant bits together.  A product of two such values accumulated in f32 as h h' + h l' + l h'
// (v_mfma_f32_32x32x16_f16, three instructions instead of one) reproduces x x' to a few 1e-7 relative:
// the dropped l l' term is below f32 resolution.  Measured on unit-vector operands against an f64
// product, a 128-term dot product has the rms error of an f32 matmul.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nemo {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));     // one MFMA operand fragment
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int SG_NP = 2;                        // f16 pieces per f32 value
constexpr float SG_FEAT_SCALE = 256.0f;         // feature maps are unit vectors per pixel: scaled values stay far inside f16
constexpr float SG_FEAT_LIMIT = 65504.0f;       // largest finite f16: scaled values beyond it are clamped

// A NaN stays a NaN (fminf / fmaxf alone would turn it into the limit), as in the f32 paths.
__device__ inline float scale_clamp(float x) {
    const float y = x * SG_FEAT_SCALE;
    return y != y ? y : fminf(fmaxf(y, -SG_FEAT_LIMIT), SG_FEAT_LIMIT);
}

// torch.max(a, b) semantics: NaN if either operand is NaN (fmaxf alone returns the other operand).
__device__ inline float max_nan(float a, float b) { return __builtin_isunordered(a, b) ? a + b : fmaxf(a, b); }

__device__ inline void split2(float x0, float x1, uint32_t* h, uint32_t* l) {
    const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
    const _Float16 l0 = (_Float16)(x0 - (float)h0), l1 = (_Float16)(x1 - (float)h1);
    union { f16x2 v; uint32_t u; } ph, pl;
    ph.v = (f16x2){h0, h1}; pl.v = (f16x2){l0, l1};
    *h = ph.u; *l = pl.u;
}

// Eight (already scaled) values -> out[0] = the eight h pieces, out[1] = the eight l pieces.
__device__ inline void split8(const float x[8], uint4 out[SG_NP]) {
    uint32_t h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split2(x[2 * j], x[2 * j + 1], &h[j], &l[j]);
    out[0] = make_uint4(h[0], h[1], h[2], h[3]);
    out[1] = make_uint4(l[0], l[1], l[2], l[3]);
}

// Header of the vertex bank's fragment buffer (first 256 bytes): largest |memory| as float bits, the power-of-two
// scale applied to the bank, and the factor that undoes both scales on a product.
struct BankMeta { uint32_t absmax_bits; float scale; float unscale; };
constexpr size_t SG_META_BYTES = 256;

__device__ inline f16x8 as_frag(const uint4& u) {
    union { uint4 u; f16x8 f; } c; c.u = u; return c.f;
}

}  // namespace nemo
