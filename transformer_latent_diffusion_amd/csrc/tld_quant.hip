// tld_quant.hip -- MX-fp8 (OCP e4m3 elements, one E8M0 scale per 32 K-elements) quantisation of a GEMM operand.
//
// BASELINE config C4 runs the QKV / MLP GEMMs on v_mfma_scale_f32_32x32x64_f8f6f4 (tld_gemm.hip, F8 = true).  The
// reference has no fp8 path: parity of this mode is stated against the fp32 oracle at a looser
// tolerance.  Scheme (OCP Microscaling v1.0): per row and per block of 32 consecutive K-elements,
//   X = 2^(floor(log2(amax)) - 8)          (8 = emax of e4m3; stored as the E8M0 byte  floor(log2 amax) - 8 + 127)
//   q = e4m3_rne(clamp(v / X, -448, 448))
// so the block's largest element lands in [256, 512) -> saturating at 448 as the specification prescribes.
// Scales are written in the layout the GEMM's tile DMA wants: [K / 128][rows][4].
#include "tld_common.h"
#include <cmath>
#include <cstring>

namespace tld {

namespace {

__global__ __launch_bounds__(256) void quant_mx8_kernel(const bf16* __restrict__ in, uint8_t* __restrict__ out,
                                                        uint8_t* __restrict__ scale, int M, int K) {
    // a thread owns 8 consecutive elements, four adjacent lanes one 32-element block (K % 32 == 0, so a quad never
    // straddles two rows)
    const long g = (long)blockIdx.x * 256 + threadIdx.x;
    const int gpr = K >> 3;                                   // 8-element groups per row
    if (g >= (long)M * gpr) return;
    const int row = (int)(g / gpr), k0 = (int)(g - (long)row * gpr) * 8;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(in + (size_t)row * K + k0);
    float f[8], amax = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { f[e] = (float)v[e]; amax = fmaxf(amax, fabsf(f[e])); }
    amax = fmaxf(amax, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(amax), 0xB1, 0xf, 0xf, true)));   // lane ^ 1
    amax = fmaxf(amax, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(amax), 0x4E, 0xf, 0xf, true)));   // lane ^ 2
    const int e_amax = (int)((__float_as_uint(amax) >> 23) & 0xffu);          // biased exponent of the block maximum
    const int e8 = e_amax > 8 ? e_amax - 8 : 0;                               // E8M0 byte of X
    const float inv = __uint_as_float((unsigned)(254 - e8) << 23);            // 1 / X, exact
    int w0 = 0, w1 = 0;
    auto q2 = [&](float a, float b, int old, bool hi) {
        a = fminf(fmaxf(a * inv, -448.f), 448.f);
        b = fminf(fmaxf(b * inv, -448.f), 448.f);
        return hi ? __builtin_amdgcn_cvt_pk_fp8_f32(a, b, old, true) : __builtin_amdgcn_cvt_pk_fp8_f32(a, b, old, false);
    };
    w0 = q2(f[0], f[1], w0, false); w0 = q2(f[2], f[3], w0, true);
    w1 = q2(f[4], f[5], w1, false); w1 = q2(f[6], f[7], w1, true);
    *reinterpret_cast<uint2*>(out + (size_t)row * K + k0) = make_uint2((unsigned)w0, (unsigned)w1);
    if ((threadIdx.x & 3) == 0) scale[((size_t)(k0 >> 7) * M + row) * 4 + ((k0 >> 5) & 3)] = (uint8_t)e8;
}

}  // namespace

void launch_quant_mx8(const bf16* in, uint8_t* out, uint8_t* scale, int M, int K, hipStream_t s) {
    const long groups = (long)M * (K >> 3);
    hipLaunchKernelGGL(quant_mx8_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, in, out, scale, M, K);
}

// ---- host side (weights, once at finalize) --------------------------------------------------------------------------
static uint8_t e4m3_rne(float f) {
    const uint8_t sign = std::signbit(f) ? 0x80 : 0;
    float a = fabsf(f);
    if (!(a == a)) return sign | 0x7f;                         // NaN
    if (a >= 448.f) return sign | 0x7e;                        // saturate (the caller clamps anyway)
    if (a < 0.015625f) return sign | (uint8_t)lrintf(a * 512.f);      // subnormals: multiples of 2^-9 (8 -> 2^-6, code 0x08)
    int e;
    const float m = frexpf(a, &e);                             // a = m 2^e, m in [0.5, 1)
    int q = (int)lrintf(m * 16.f) - 8;                         // 3 mantissa bits of 2 m - 1
    int ex = e - 1;
    if (q == 8) { q = 0; ex += 1; }
    if (ex > 8 || (ex == 8 && q > 6)) return sign | 0x7e;
    return sign | (uint8_t)(((ex + 7) << 3) | q);
}

void quant_mx8_host(const float* w, int rows, int K, uint8_t* out, uint8_t* scale) {
    for (int r = 0; r < rows; ++r)
        for (int k0 = 0; k0 < K; k0 += 32) {
            float amax = 0.f;
            for (int e = 0; e < 32; ++e) amax = fmaxf(amax, fabsf(w[(size_t)r * K + k0 + e]));
            uint32_t bits; memcpy(&bits, &amax, 4);
            const int e_amax = (int)((bits >> 23) & 0xffu);
            const int e8 = e_amax > 8 ? e_amax - 8 : 0;
            const uint32_t ib = (uint32_t)(254 - e8) << 23;
            float inv; memcpy(&inv, &ib, 4);
            for (int e = 0; e < 32; ++e) {
                float v = w[(size_t)r * K + k0 + e] * inv;
                v = v < -448.f ? -448.f : (v > 448.f ? 448.f : v);
                out[(size_t)r * K + k0 + e] = e4m3_rne(v);
            }
            scale[((size_t)(k0 >> 7) * rows + r) * 4 + ((k0 >> 5) & 3)] = (uint8_t)e8;
        }
}

}  // namespace tld
