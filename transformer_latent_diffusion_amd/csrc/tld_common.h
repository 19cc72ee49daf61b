// tld_common.h -- shared device/host declarations of the gfx950 denoising engine.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <mutex>
#include <stdint.h>

namespace tld {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// Residual stream dtype: bf16, as in the reference's own bf16 mode (x never leaves bf16 there either).  All
// statistics, softmax and residual ADDS are computed in fp32 and rounded once on store.  Measured vs the fp32
// reference: forward rel-rms 5-7e-3, 35-step CFG-6 trajectory 1.4e-2 (tolerances 2e-2 / 6e-2; the reference's
// own bf16 path is at 0.8-1.0e-2 per forward).  -DTLD_RESID_FP32 keeps x in fp32 (2.7e-3 / 4.6e-3, ~4 % slower).
#ifdef TLD_RESID_FP32
typedef float resid_t;
#else
#define TLD_RESID_BF16 1
typedef __bf16 resid_t;
#endif

constexpr int kWave = 64;
constexpr int kHeadDim = 64;
constexpr float kLnEps = 1e-5f;

// Full-wave (64-lane) all-reduce sum without LDS traffic: four DPP adds inside each 16-lane row
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror), then the gfx950 half-exchange instructions
// v_permlane16_swap / v_permlane32_swap for the two cross-row steps.  Every lane ends with the total.
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    const int s = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true);
    return v + __int_as_float(s);
}
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_add<0xB1>(v);      // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);      // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);     // row_half_mirror
    v = dpp_add<0x140>(v);     // row_mirror
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

// Sum within each 32-lane half of the wave (every lane ends with its half's total).
__device__ __forceinline__ float half_sum(float v) {
    v = dpp_add<0xB1>(v);
    v = dpp_add<0x4E>(v);
    v = dpp_add<0x141>(v);
    v = dpp_add<0x140>(v);
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// erf by Abramowitz-Stegun 7.1.28:  1 - (1 + a1 x + ... + a6 x^6)^-16  (|error| <= 3e-7; 9e-7 measured in fp32
// through the GELU).  One transcendental (v_rcp) + six FMAs + four squarings; the 7.1.26 form used before
// needed v_rcp AND v_exp, and quarter-rate transcendentals were ~40 % of the fused depthwise epilogue.
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    float p = fmaf(0.0000430638f, ax, 0.0002765672f);
    p = fmaf(p, ax, 0.0001520143f);
    p = fmaf(p, ax, 0.0092705272f);
    p = fmaf(p, ax, 0.0422820123f);
    p = fmaf(p, ax, 0.0705230784f);
    p = fmaf(p, ax, 1.0f);
    float r = __builtin_amdgcn_rcpf(p);
    r *= r; r *= r; r *= r; r *= r;
    return copysignf(1.0f - r, x);
}

// exact-erf GELU (nn.GELU default): 0.5 x (1 + erf(x / sqrt 2)).  `precise` uses the libm-grade erff
// (conditioning path); the fast form is for outputs that are rounded to bf16 anyway (rel 2^-9).
__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float gelu_erf_fast(float x) {
    return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f));
}
// two-lane form (v_pk_* arithmetic; v_rcp per component)
__device__ __forceinline__ f32x2 gelu_erf_fast2(f32x2 x) {
    const f32x2 z = x * 0.70710678118654752440f;
    const f32x2 az = __builtin_elementwise_abs(z);
    f32x2 p = __builtin_elementwise_fma(f32x2{0.0000430638f, 0.0000430638f}, az, f32x2{0.0002765672f, 0.0002765672f});
    p = __builtin_elementwise_fma(p, az, f32x2{0.0001520143f, 0.0001520143f});
    p = __builtin_elementwise_fma(p, az, f32x2{0.0092705272f, 0.0092705272f});
    p = __builtin_elementwise_fma(p, az, f32x2{0.0422820123f, 0.0422820123f});
    p = __builtin_elementwise_fma(p, az, f32x2{0.0705230784f, 0.0705230784f});
    p = __builtin_elementwise_fma(p, az, f32x2{1.0f, 1.0f});
    f32x2 r = {__builtin_amdgcn_rcpf(p[0]), __builtin_amdgcn_rcpf(p[1])};
    r *= r; r *= r; r *= r; r *= r;
    const f32x2 e = {copysignf(1.0f - r[0], z[0]), copysignf(1.0f - r[1], z[1])};       // erf(z)
    const f32x2 hx = x * 0.5f;
    return __builtin_elementwise_fma(hx, e, hx);
}
// GELU of 2y given y (the fused depthwise epilogue halves the conv weights and bias on the host -- exact -- so the conv
// delivers y = x / 2):  GELU(x) = y (1 + erf(sqrt2 y)), with sqrt2^i folded into the A&S coefficients.  Two packed
// multiplies per pair fewer than gelu_erf_fast2(x).
#ifndef TLD_GELU_POLY4
#define TLD_GELU_POLY4 2      // 2 (round 4): division-free clamp(y R(y^2)), |GELU error| <= 1.7e-4 incl. the negative tail; fused up-projection 194.6 -> 192.3 us same-box
#endif                        // 1: erfc by Abramowitz-Stegun 7.1.27 (four coefficients, ^-4, |erf error| <= 5e-4; O(1e-3) absolute on the negative tail);
                              // 0: 7.1.28 (six coefficients, ^-16, 3e-7).  See DESIGN.md 4.1 for the parity they were judged by.
__device__ __forceinline__ f32x2 gelu_erf_fast2_half(f32x2 y) {
#if TLD_GELU_POLY4 == 2
    // division-free: erf(sqrt2 y) ~ clamp(y R(y^2), -1, 1), R of degree 6 in y^2 (weighted minimax fit on |y| <= 1.98, beyond which the even
    // polynomial grows and the clamp takes over): |GELU error| <= 1.7e-4 everywhere.  11 packed slots per pair against 18 (two quarter-rate v_rcp).
    const f32x2 u = y * y;
    // (Horner; the Estrin form -- three levels instead of six dependent FMAs -- measured no faster: 197.2 us either way)
    f32x2 r = __builtin_elementwise_fma(f32x2{3.952182666e-04f, 3.952182666e-04f}, u, f32x2{-6.822538060e-03f, -6.822538060e-03f});
    r = __builtin_elementwise_fma(r, u, f32x2{5.043737174e-02f, 5.043737174e-02f});
    r = __builtin_elementwise_fma(r, u, f32x2{-2.115262865e-01f, -2.115262865e-01f});
    r = __builtin_elementwise_fma(r, u, f32x2{5.651242001e-01f, 5.651242001e-01f});
    r = __builtin_elementwise_fma(r, u, f32x2{-1.035131935e+00f, -1.035131935e+00f});
    r = __builtin_elementwise_fma(r, u, f32x2{1.591872892e+00f, 1.591872892e+00f});
    f32x2 e = y * r;
    e[0] = __builtin_amdgcn_fmed3f(e[0], -1.0f, 1.0f); e[1] = __builtin_amdgcn_fmed3f(e[1], -1.0f, 1.0f);
    return __builtin_elementwise_fma(y, e, y);
#endif
    const f32x2 ay = __builtin_elementwise_abs(y);
#if TLD_GELU_POLY4
    // erfc(sqrt2 |y|) = (1 + b1 |y| + b2 |y|^2 + b3 |y|^3 + b4 |y|^4)^-4,  b_i = a_i sqrt2^i of A&S 7.1.27: four packed FMAs and two
    // squarings instead of six and four.  The GELU's absolute error is <= 2.5e-4 |x| (x = 2 y), i.e. below the bf16 rounding of the
    // stored activation (2^-9 relative) wherever |GELU| is not itself tiny, and O(1e-3) absolute on the negative tail where the
    // exact value is ~ -1e-3 .. -1e-2: invisible in the forward / trajectory rel-rms (profiles/r03_parity_report.md).
    f32x2 p = __builtin_elementwise_fma(f32x2{0.312432f, 0.312432f}, ay, f32x2{0.00274923f, 0.00274923f});
    p = __builtin_elementwise_fma(p, ay, f32x2{0.460778f, 0.460778f});
    p = __builtin_elementwise_fma(p, ay, f32x2{0.393707156f, 0.393707156f});
    p = __builtin_elementwise_fma(p, ay, f32x2{1.0f, 1.0f});
    f32x2 r4 = {__builtin_amdgcn_rcpf(p[0]), __builtin_amdgcn_rcpf(p[1])};
    r4 *= r4; r4 *= r4;
    return __builtin_elementwise_fma(-ay, r4, y + ay);
#else
    f32x2 p = __builtin_elementwise_fma(f32x2{0.0003445104f, 0.0003445104f}, ay, f32x2{0.001564500341f, 0.001564500341f});
    p = __builtin_elementwise_fma(p, ay, f32x2{0.0006080572f, 0.0006080572f});
    p = __builtin_elementwise_fma(p, ay, f32x2{0.02622101059f, 0.02622101059f});
    p = __builtin_elementwise_fma(p, ay, f32x2{0.0845640246f, 0.0845640246f});
    p = __builtin_elementwise_fma(p, ay, f32x2{0.09973469393f, 0.09973469393f});
    p = __builtin_elementwise_fma(p, ay, f32x2{1.0f, 1.0f});
    f32x2 r = {__builtin_amdgcn_rcpf(p[0]), __builtin_amdgcn_rcpf(p[1])};
    r *= r; r *= r; r *= r; r *= r;                                                      // erfc(sqrt2 |y|)
    // y (1 + erf(sqrt2 y)) = (y + |y|) - |y| erfc(sqrt2 |y|)  for either sign of y: no copysign / select needed
    return __builtin_elementwise_fma(-ay, r, y + ay);
#endif
}
// (A transcendental-free degree-9 polynomial erf was measured ~4 % SLOWER end to end: its 10-deep dependent FMA
// chain is latency-bound at the 2 waves/SIMD of the fused GEMM epilogue.)

__device__ __forceinline__ float2 rs_load2(const resid_t* p) {
#ifdef TLD_RESID_BF16
    const bf16x2 v = *reinterpret_cast<const bf16x2*>(p);
    return make_float2((float)v[0], (float)v[1]);
#else
    return *reinterpret_cast<const float2*>(p);
#endif
}
__device__ __forceinline__ void rs_store2(resid_t* p, float2 v) {
#ifdef TLD_RESID_BF16
    bf16x2 o; o[0] = (bf16)v.x; o[1] = (bf16)v.y;
    *reinterpret_cast<bf16x2*>(p) = o;
#else
    *reinterpret_cast<float2*>(p) = v;
#endif
}
__device__ __forceinline__ float4 rs_load4(const resid_t* p) {
#ifdef TLD_RESID_BF16
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
#else
    return *reinterpret_cast<const float4*>(p);
#endif
}
// four residual values as they sit in memory (prefetch registers: half the space of float4 in the bf16 build) and their fp32 view
#ifdef TLD_RESID_BF16
typedef bf16x4 resid4_t;
__device__ __forceinline__ float4 rs_widen4(resid4_t v) { return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]); }
#else
typedef float4 resid4_t;
__device__ __forceinline__ float4 rs_widen4(resid4_t v) { return v; }
#endif
__device__ __forceinline__ resid4_t rs_raw4(const resid_t* p) { return *reinterpret_cast<const resid4_t*>(p); }
__device__ __forceinline__ void rs_store4(resid_t* p, float4 v) {
#ifdef TLD_RESID_BF16
    bf16x4 o; o[0] = (bf16)v.x; o[1] = (bf16)v.y; o[2] = (bf16)v.z; o[3] = (bf16)v.w;
    *reinterpret_cast<bf16x4*>(p) = o;
#else
    *reinterpret_cast<float4*>(p) = v;
#endif
}

// MX-fp8 output of a row producer (fp8 GEMM mode): the four values of this lane and those of the 7 other lanes of its aligned
// 8-lane group form one 32-element block (the row kernels' 4-features-per-lane layout and the depthwise kernels' channel quads
// both have that shape).  Returns the packed e4m3 dword; *e8 is the block's E8M0 byte (same in all 8 lanes).  Same arithmetic as
// quant_mx8_kernel (tld_quant.hip): X = 2^(floor(log2 amax) - 8), saturating conversion.
__device__ __forceinline__ unsigned mx8_pack4(float v0, float v1, float v2, float v3, int* e8_out) {
    // block maximum of |v| on the BIT PATTERNS (monotonic for non-negative floats): unsigned max needs no canonicalisation and takes the DPP operand
    // directly (v_max_u32_dpp) -- the float form compiled to v_mov_dpp + v_max + a canonicalising v_max per step (round 6: 12 -> 9 instructions per 4 values)
    const unsigned a0 = __float_as_uint(v0) & 0x7fffffffu, a1 = __float_as_uint(v1) & 0x7fffffffu;
    const unsigned a2 = __float_as_uint(v2) & 0x7fffffffu, a3 = __float_as_uint(v3) & 0x7fffffffu;
    unsigned am = a0 > a1 ? a0 : a1;
    const unsigned am2 = a2 > a3 ? a2 : a3;
    am = am > am2 ? am : am2;
    auto dmax = [](unsigned x, unsigned y) { return x > y ? x : y; };
    am = dmax(am, (unsigned)__builtin_amdgcn_update_dpp(0, (int)am, 0xB1, 0xf, 0xf, true));     // lane ^ 1
    am = dmax(am, (unsigned)__builtin_amdgcn_update_dpp(0, (int)am, 0x4E, 0xf, 0xf, true));     // lane ^ 2
    am = dmax(am, (unsigned)__builtin_amdgcn_update_dpp(0, (int)am, 0x141, 0xf, 0xf, true));    // 8-lane mirror
    const int e_amax = (int)(am >> 23);
    const int e8 = e_amax > 8 ? e_amax - 8 : 0;
    const float inv = __uint_as_float((unsigned)(254 - e8) << 23);
    auto cl = [&](float v) { return fminf(fmaxf(v * inv, -448.f), 448.f); };
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(cl(v0), cl(v1), w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(cl(v2), cl(v3), w, true);
    *e8_out = e8;
    return (unsigned)w;
}
// scale byte address of (row, first K-element k of a 32-block) in the GEMM's [K/128][rows][4] layout
__device__ __forceinline__ size_t mx8_scale_index(int k, size_t row, size_t rows) { return ((size_t)(k >> 7) * rows + row) * 4 + ((k >> 5) & 3); }

// value the residual stream will hold after a store (what a later LayerNorm of the stored row sees)
__device__ __forceinline__ float rs_round(float v) { return (float)(resid_t)v; }

#ifndef TLD_DW_GELU
#define TLD_DW_GELU gelu_erf_fast
#endif

// The MLP hidden activation (201 MB per layer, written by the up-projection epilogue) is stored nontemporal: a plain
// store leaves 128 KB of dirty lines per tile in the XCD's 4 MB L2 and pushes the A / W tiles its 32 workgroups
// share back out to the fabric (PMC: L2-miss traffic 1.7x the algorithmic bytes).  Measured, same box:
//   hidden NT            +1.4 % end to end (its consumer, the down GEMM, gains most: 184 -> 174 us)
//   q|k, v^T NT as well  -0.9 % (attention then misses L2 on its inputs: 50 -> 54 us) -> those stay plain
//   8-B NT stores in the row kernels: attention 50 -> 110 us (partial lines are not combined) -> plain
//   residual read-modify-write of the down GEMM NT: no change -> plain
// -DTLD_NT_STORES=0 restores plain stores everywhere.
#ifndef TLD_NT_STORES
#define TLD_NT_STORES 1
#endif
#if TLD_NT_STORES
#define TLD_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define TLD_STORE(ptr, val) (*(ptr) = (val))
#endif
// ---- launch descriptors ------------------------------------------------------------------------

enum GemmEpilogue {
    EPI_F32 = 0,         // C fp32 [M,N]                         (debug / small tables)
    EPI_QKV = 1,         // q,k -> bf16 [M,2d] ; v -> bf16 transposed per (sample, head): [B,H,64,Ntok]
    EPI_BIAS_BF16 = 2,   // bf16(C + bias[n]) -> [M,N]           (MLP up projection)
    EPI_BIAS_RESID = 3,  // x[m,n] += C + bias[n] (resid_t)      (MLP down projection)
    EPI_QKV_LN = 5,      // EPI_QKV with LayerNorm-1 folded in (A = raw residual stream, see GemmParams::ln_stats)
    EPI_UP_DWCONV32 = 4, // the same on a 32 x 32 token grid (a tile = 8 image rows): interior rows here, seam rows by launch_dwconv_seam (round 4; a separate
                         //   instantiation so that the 16 x 16 kernel's register allocation stays what it was)
    EPI_UP_DWCONV2 = 6,  // bf16(C + bias) -> depthwise 3x3 + bias + GELU over the tile's 16x16 image -> [M,N]  (MLP up projection fused with
                         // the depthwise conv: token-pair image in LDS, packed-bf16 taps on v_dot2c_f32_bf16; needs ntok == 256, BN == 256.
                         // Value 4 was the first form of this epilogue, retired in round 4.)
    EPI_QKV_ATTN = 7,    // QKV GEMM (LayerNorm-1 folded in) + the head's whole self-attention in the epilogue: W rows are permuted to
                         // [head][q_h | k_h | v_h] so that a 256 x 192 tile = everything (sample, head) needs; q, k, v^T go from the
                         // accumulators to LDS, softmax(q k^T / 8) v runs there, and only att[256 x 64] is written (out_bf16, ldo = d).
                         // Needs ntok == 256, N = 3 d = heads x 192, K % 128 == 0.  (tld/transformer_blocks.py:51-59 + 24-48)
};

// Launch-side caches are PER DEVICE: hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the current device only, and engines on
// different devices may live in one process (every ABI entry point runs under a device guard).
// Host threads may drive different devices (or the same one) for the first time concurrently, so the flags are atomic and a flag is
// published only AFTER the attribute call it guards has returned: a racing thread either sees the bit (the attribute is set) or sets the
// attribute itself once more, which is harmless.
struct PerDeviceOnce {
    std::atomic<unsigned long long> mask{0};
    template <class F> void run(F&& set_attribute) {     // set_attribute() runs (at least) once per device, before any launch that needs it
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned long long bit = 1ull << (dev & 63);
        if (mask.load(std::memory_order_acquire) & bit) return;
        set_attribute();
        mask.fetch_or(bit, std::memory_order_release);
    }
};
struct PerDeviceMax {                                // largest value requested so far on the current device
    // The attribute call and the publication of the new maximum are ONE critical section: two threads that first-launch the same kernel
    // on one device with different sizes could otherwise leave the attribute at the smaller value while the cache says the larger one
    // (round-5 advisor finding).  The fast path -- the cached value already covers the request -- stays lock-free.
    std::atomic<int> v[64] = {};
    std::mutex mu;
    template <class F> void run(int want, F&& set_attribute) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::atomic<int>& a = v[dev & 63];
        if (a.load(std::memory_order_acquire) >= want) return;
        std::lock_guard<std::mutex> lk(mu);
        if (a.load(std::memory_order_relaxed) >= want) return;       // another thread raised it while this one waited
        set_attribute();
        a.store(want, std::memory_order_release);
    }
};
// device that owns a device pointer (debug hooks that take raw pointers and no engine); -1 if unknown
inline int ptr_device(const void* p) {
    hipPointerAttribute_t a;
    if (p && hipPointerGetAttributes(&a, p) == hipSuccess) return a.device;
    (void)hipGetLastError();
    return -1;
}
struct PtrDeviceGuard {                              // runs the scope on the pointer's device, restores the caller's device after
    int prev = -1; bool switched = false;
    explicit PtrDeviceGuard(const void* p) {
        const int dev = ptr_device(p);
        if (dev >= 0 && hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~PtrDeviceGuard() { if (switched) (void)hipSetDevice(prev); }
    PtrDeviceGuard(const PtrDeviceGuard&) = delete;
    PtrDeviceGuard& operator=(const PtrDeviceGuard&) = delete;
};
inline int device_cu_count() {
    static int cus[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    int& c = cus[dev & 63];
    if (!c) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        c = n;
    }
    return c;
}

struct GemmParams {
    const bf16* A; int lda;       // [M,K] row-major, K contiguous
    const bf16* W; int ldw;       // [N,K] row-major (nn.Linear weight layout)
    int M, N, K;
    float* c_f32; int ldc;        // EPI_F32
    bf16* out_bf16; int ldo;      // EPI_QKV (q|k, ldo = 2d) / EPI_BIAS_BF16
    bf16* vt;                     // EPI_QKV
    int ntok, d;                  // EPI_QKV
    const float* bias;            // EPI_BIAS_*
    const float* dw_b;            // EPI_UP_DWCONV2: HALVED depthwise bias [N]  (the epilogue's GELU takes x / 2)
    uint32_t* dw_seam;            // EPI_UP_DWCONV32: the epilogue computes the six interior rows of its 8 image rows (+ the image's own top / bottom row) and leaves
                                  //   rows 0, 1, 6, 7 of the hidden tensor, in its token-pair image format, here for launch_dwconv_seam:
                                  //   [M / 256 tiles][N / 256 column tiles][64 pair-rows][256 channels] dwords
    const uint32_t* dw_wpk;       // EPI_UP_DWCONV2: HALVED depthwise taps as packed bf16 pairs [3 window rows][4 kinds][N]:
                                  //   kinds (lo, hi): (0, w0), (w1, w2) for even output columns; (w0, w1), (w2, 0) for odd ones
    resid_t* resid; int ldr;      // EPI_BIAS_RESID
    // LayerNorm-1 folded into the QKV GEMM (EPI_QKV_LN): the producers of the residual stream (embed, EPI_BIAS_RESID)
    // leave per-row partial sums (sum x, sum x^2) of the ROUNDED values, one slot per 96-column group (slot index =
    // column / 96, the same for 192- and 384-wide tiles, so results do not depend on the tile shape chosen for a
    // batch size); the QKV GEMM multiplies the raw residual by bf16(gamma1 (.) Wqkv) and its epilogue applies
    // rstd_m (acc - mean_m c1[n]) + b1[n].
    float2* stats_out;            // EPI_BIAS_RESID: [M][kLnSlots] partials out (null: none)
    const float2* ln_stats;       // EPI_QKV_LN: [M][kLnSlots] partials of the A rows
    int ln_slots;                 // EPI_QKV_LN: slots to sum per row (even, <= kLnSlots)
    const float* ln_b1;           // EPI_QKV_LN: [N] beta1 . Wqkv^T   (ln_c1 below holds the column sums)
    // EPI_UP_DWCONV2 / EPI_BIAS_BF16 with LayerNorm-3 folded in: A is the raw bf16 residual stream, W = bf16(gamma3 (.) Wup),
    // bias = up_b + beta3 . Wup^T, and the image write applies  rstd_m (acc - mean_m c1[n]) + bias[n]
    const float2* row_stats;      // [M] (mean, rstd) per row; null: A is already normalized
    const float* ln_c1;           // [N] column sums of the gamma-scaled bf16 weights
    // MX-fp8 operands (f8 != 0): A and W hold OCP e4m3 bytes ([M,K] / [N,K], K contiguous, lda / ldw in elements = bytes),
    // a_scale / w_scale one E8M0 byte per 32 K-elements, laid out [K/128][rows][4] so that a tile's scales of one 128-wide
    // K-step are contiguous.  K % 128 == 0, M % 4 == 0, N % 4 == 0.  Epilogues: EPI_F32, EPI_QKV, EPI_BIAS_BF16, EPI_BIAS_RESID.
    int f8;
    const uint8_t* a_scale;
    const uint8_t* w_scale;
    // Implicit 3x3 convolution (conv != 0; VAE decoder, tld_vae.hip): the A operand is never materialised.  Row m of the GEMM is
    // output pixel (b, y, x) of a channels-last [B, cv_h, cv_w, *] image, K = 9 cv_cin with k = tap * cv_cin + c (tap = 3 ky + kx,
    // W laid out [N][3][3][cv_cin]), and K-step k of a tile is DMA'd from the 128-byte channel slice of source pixel
    // (y + ky - 1, x + kx - 1) -- or of ((y + ky - 1) >> 1, (x + kx - 1) >> 1) in a half-resolution source when cv_up (nearest
    // 2x upsampling folded into the addressing).  A points at the activation BUFFER, whose first cv_data_off bytes are
    // zeros (>= 2 cv_cin: one whole pixel): taps outside the image read that zero pixel, so there is no border code in
    // the kernel.
    // cv_cin % 64 == 0; buffer size < 4 GiB; epilogues EPI_F32, EPI_BIAS_BF16, EPI_BIAS_RESID; 256- and 128-wide tiles.
    // Block-diagonal batching of the W operand (w_batch_rows != 0; EPI_F32 / EPI_BIAS_BF16 only): the A rows are w_batch_rows-row
    // groups stacked into one tall matrix, and group g multiplies ITS OWN W matrix at byte offset g * w_batch_stride_bytes from W
    // (attention inside the VAE decoder: scores_b = Q_b K_b^T and O_b = P_b V_b for all samples in one launch each).
    // w_batch_rows % 256 == 0 (a tile never straddles groups); the offset stays inside the 32-bit DMA offsets.
    int w_batch_rows;
    unsigned w_batch_stride_bytes;
    int conv, cv_h, cv_w, cv_up, cv_cin;
    unsigned cv_data_off;
    // conv epilogues EPI_BIAS_BF16 / EPI_BIAS_RESID: GroupNorm statistics of the OUTPUT for its consumer, fused into the epilogue
    // (null: none).  partial[(sample, 256-pixel chunk)][group] = (sum, sum of squares) of the output values (bias-to-bf16 epilogue: before their rounding), gn_cpg channels per
    // group, gn_hw pixels per sample.  Requires gn_hw % 256 == 0 (a tile never straddles samples), N % 128 == 0, gn_cpg % 4 == 0.
    float2* gn_partial;
    int gn_groups, gn_cpg, gn_hw;
    int ksplit;                   // > 1 (EPI_F32, bf16, no conv): split-K -- K is the length of ONE split, split s multiplies columns [s K, (s + 1) K) of A and W (lda / ldw
                                  //   = the full row pitch) and writes its fp32 product to c_f32 + s M ldc; the caller sums the slices in a fixed order
    int xcd_ngroups;              // > 1: XCDs form a (8 / G) x G grid over (tile-rows, tile-column groups); needs ntn % G == 0
    // Transposed-operand form (launch_gemm_tn; the weight gradients dW = dY^T X of the training step): A [k rows][lda] and W [k rows][ldw] are
    // both row-major with the CONTRACTION index as the row, C[M, N] = sum_k A[k][m] W[k][n].  tn_ktotal = rows of both operands; K = rows per
    // split (multiple of 64); w_batch_rows != 0: output rows [j w_batch_rows, (j + 1) w_batch_rows) are split j = operand rows [j K, min((j + 1) K,
    // tn_ktotal)) against the same A columns (split-K into fp32 slices, summed by the caller in a fixed order).
    int tn_ktotal;
    int half_tail;                // ring K loop: split the tiles of a partly filled last round by ROWS between two workgroups (set by the launcher)
    int dbg_epi;                  // experiment knob, builds with -DTLD_DBG_EPI only (TLD_EPI_DBG bit mask, see tld_gemm.hip)
};

void launch_gemm(const GemmParams& p, int epilogue, hipStream_t s);
// EPI_UP_DWCONV2 on 256 x 128 tiles with 4-wave workgroups (tld_updw.hip): bitwise equal to the 8-wave kernel, taken by launch_gemm for launches of at most one tile per CU
bool updw_pp_supported(const GemmParams& p);
void launch_updw_pp(const GemmParams& p, hipStream_t s);
bool splitk_pp_supported(const GemmParams& p);          // EPI_F32 with ksplit > 1 on the same 4-wave K loop (bitwise the slices of gemm256p_kernel<128, EPI_F32>)
void launch_splitk_pp(const GemmParams& p, hipStream_t s);
bool down_pp_supported(const GemmParams& p);            // EPI_BIAS_RESID (default class) on 128 x 192 tiles with 4-wave workgroups: bitwise gemm256p_kernel<192 | 384, EPI_BIAS_RESID>
bool down_pp_fits(const GemmParams& p);                 // ... at most one 64- or 128-row tile per CU
void launch_down_pp(const GemmParams& p, hipStream_t s);
// the two rows on either side of every tile seam of a 32 x 32 image (dw_grid = 32): depthwise 3x3 + GELU on the seam rows the fused epilogue left
void launch_dwconv_seam(const uint32_t* seam, const uint32_t* dw_wpk, const float* dw_b_half, bf16* out, int ldo, int batch, int channels, hipStream_t s);
void launch_gemm_tn(const GemmParams& p, hipStream_t s);        // EPI_F32, 256 x 256 tiles: M % 256 == 0, N % 256 == 0, K % 64 == 0, tn_ktotal % 64 == 0

// thread-local message behind tld_last_error() (tld_engine.hip); for the other host translation units
void set_last_error(const char* msg);

// MX-fp8 quantisation of a GEMM operand (tld_quant.hip): e4m3 elements [rows, K] + E8M0 block scales [K/128][rows][4]
void launch_quant_mx8(const bf16* in, uint8_t* out, uint8_t* scale, int M, int K, hipStream_t s);
void quant_mx8_host(const float* w, int rows, int K, uint8_t* out, uint8_t* scale);
constexpr int kLnSlots = 8;
// slots an EPI_BIAS_RESID launch of width N writes for every batch size (0: shape not supported -> keep the LN kernel)
int gemm_resid_stat_slots(int N);


// self-attention over ntok tokens, head_dim 64: softmax(q k^T / 8) v -> att bf16 [M, d]
void launch_attention(const bf16* qk, const bf16* vt, bf16* att, int batch, int ntok, int heads,
                      hipStream_t s);

// depthwise 3x3 (zero pad) + bias + exact GELU on channels-last [B, g, g, C] bf16
// (w9c_half / bias_half: the same tables times 0.5, used by the spatially tiled kernel's half-argument GELU)
// out8 / scale8 non-null (spatially tiled kernel only): the result goes out as e4m3 [M, C] + E8M0 scales [C/128][M][4]
// (fp8 GEMM mode) instead of bf16
void launch_dwconv_gelu(const bf16* in, bf16* out, const float* w9c /*[9][C]*/, const float* bias, const float* w9c_half,
                        const float* bias_half, int batch, int grid, int channels, hipStream_t s, uint8_t* out8 = nullptr,
                        uint8_t* scale8 = nullptr);

struct EmbedParams {
    const float* x;               // [B,C,S,S] fp32
    const float* conv_w;          // [pd, C*p*p]
    const float* conv_b;          // [pd]
    const float* ln1_w; const float* ln1_b;   // [pd]
    const float* lin_wt;          // [pd, d]  (transposed nn.Linear weight)
    const bf16* lin_w_hl;         // optional [2][d][pd]: the nn.Linear weight [d, pd] split into bf16 hi | lo halves (matrix-pipe form; null: VALU form)
    const float* lin_b;           // [d]
    const float* ln2_w; const float* ln2_b;   // [d]
    const float* pos;             // [N, d]
    resid_t* tok;                 // [B*N, d]
    float2* stats_out;            // optional [B*N][kLnSlots]: slot 0 <- (sum, sum of squares) of the rounded row, slot 1 <- 0
    int batch, src_batch;         // model sample b reads latent b % src_batch (CFG doubling without a copy)
    int C, S, p, grid, pd, d, ntok;
};
void launch_embed(const EmbedParams& p, hipStream_t s);

// LayerNorm rows: x [M,d] -> bf16 normalized-affine [M,d]
void launch_layernorm_bf16(const resid_t* x, const float* g, const float* b, bf16* out, int M, int d,
                           hipStream_t s);
// the same rows straight into the fp8 GEMM's operand format: e4m3 [M,d] + E8M0 scales [d/128][M][4]  (d % 256 == 0)
bool layernorm_mx8_supported(int d);
void launch_layernorm_mx8(const resid_t* x, const float* g, const float* b, uint8_t* out8, uint8_t* scale8, int M, int d,
                          hipStream_t s);

// Fused row kernel of one decoder block's middle:
//   x += att (self-attention residual);  cross-attention to the 2 conditioning tokens computed from
//   row statistics and per-sample folded query vectors;  x += cross;  xn3 = LN3(x) in bf16.
struct CrossRowParams {
    resid_t* x;                   // [M,d] out (and in, when x_in is null)
    const resid_t* x_in;          // optional separate input stream [src_batch*ntok, d] (CFG layer-0 sharing)
    const bf16* att;              // [M,d]  (or [src_batch*ntok, d] with x_in)
    int src_batch;                // with x_in: model sample b reads x_in / att of sample b % src_batch
    const float* wq;              // [T, H, d]  gamma2 * (Wq_h^T k_t[h] / 8) for this layer (per token row)
    const float* bwq;             // [T, H]     sum_j beta2[j] * (Wq_h^T k_t[h] / 8)[j]
    const float* v; int v_ld;     // [T, v_ld]  cross-attention values (per token row) for this layer
    const int* noise_row;         // [B] token row of each sample's noise token
    const int* label_row;         // [B] token row of each sample's label token
    const float* ln2_w; const float* ln2_b;
    const float* ln3_w; const float* ln3_b;
    bf16* xn3;                    // [M,d]  LN3(x) for the up-projection (null when ln3_stats is used instead)
    float2* ln3_stats;            // optional [M]: (mean, rstd) of the ROUNDED new residual row -- LN3 is then applied
                                  // inside the fused up-projection's epilogue and xn3 is never written
    uint8_t* xn3_f8;              // optional (fp8 GEMM mode, 4-features-per-lane kernel): LN3(x) as e4m3 [M,d] instead of xn3 ...
    uint8_t* xn3_s8;              //   ... and its E8M0 block scales [d/128][M][4]
    float* sa_out;                // optional debug dump of x + att  [M,d]
    int batch, ntok, d, heads;
};
void launch_cross_row(const CrossRowParams& p, hipStream_t s);
bool cross_row_supports_ln3_stats(int d);
// split-K finish of the low-latency down projection: x += bias + sum of the fp32 slices (+ LayerNorm-1 partial sums); d = 768 / 384
bool splitk_resid_supported(int d);
void launch_splitk_resid(const float* parts, int nsplit, size_t slice_stride, const float* bias, resid_t* x, float2* stats_out, int M, int d, hipStream_t s);

struct TailParams {
    const resid_t* tok;           // [B*N, d]
    const float* w;               // [pd, d]
    const bf16* w_hl;             // optional [2][pd][d]: w split into bf16 hi | lo halves (matrix-pipe form of the kernel; null: VALU form)
    const float* b;               // [pd]
    float* out;                   // [B,C,S,S] fp32
    int batch, C, S, p, grid, pd, d, ntok;
};
void launch_tail(const TailParams& p, hipStream_t s);

// sampler elementwise: CFG combine + multistep update (see schedule.py)
struct UpdateParams {
    const float* x0_2b;           // [2B, img] model output (cond rows first, then uncond)
    float* x_t;                   // [B, img]  in/out
    float* x0_prev;               // [B, img]  in/out
    float* x0_out;                // [B, img]  CFG-combined prediction (always written)
    float* trace_x0; float* trace_xt;  // optional [B, img]
    float g, a, b, c, c1, c2, sharp, bright;
    int final_step;               // 1: only combine (+shifts on channels 3 and 0), no update
    int batch, img, chan_stride, C;
};
void launch_update(const UpdateParams& p, hipStream_t s);

// ---- conditioning path (fp32) ------------------------------------------------------------------
// out[t, n] = act(sum_k in[t,k] W[n,k] + b[n]);  act: 0 none, 1 exact GELU
void launch_linear_f32(const float* in, int ldi, const float* W, const float* b, float* out, int ldo,
                       int T, int K, int N, int act, hipStream_t s);
void launch_sinusoid(const float* sigma, const float* angular, float* out, int T, int half, hipStream_t s);
void launch_layernorm_f32(const float* x, const float* g, const float* b, float* out, int M, int d,
                          hipStream_t s);
// raw[t,h,j] = (1/8) sum_c k[t, h*64+c] Wq[h*64+c, j];  wq = raw * gamma[j];  bwq[t,h] = sum_j beta[j] raw
void launch_linear_f32_layers(const float* in, int ldi, const float* const* Wl, float* out, size_t out_lstride,
                              int ldo, int T, int K, int N, int layers, hipStream_t s);
void launch_wq_layers(const float* k, size_t k_lstride, int ldk, const float* const* Wql, const float* const* gammal,
                      const float* const* betal, float* wq, size_t wq_lstride, float* bwq, size_t bwq_lstride, int T,
                      int heads, int d, int layers, hipStream_t s);
// dst_f32 <- src (fp32/bf16/f16) and back
void launch_cast_to_f32(const void* src, int dtype, float* dst, int64_t n, hipStream_t s);
void launch_cast_from_f32(const float* src, void* dst, int dtype, int64_t n, hipStream_t s);
void launch_iota(int* dst, int n, int base, hipStream_t s);
void launch_fill_bf16(bf16* dst, int64_t n, uint32_t seed, float scale, hipStream_t s);

}  // namespace tld
