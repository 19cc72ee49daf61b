// tld_cond.hip -- the conditioning path, always fp32 (sigma * angular speeds up to 2*pi*1000 has no
// meaningful bf16 phase; SURVEY.md section 7 "noise-embedding precision").
//
//   sinusoid_kernel     SinusoidalEmbedding.forward                      tld/transformer_blocks.py:17-21
//   linear_f32_kernel   fourier_feats[1], [3], label_proj, kv_linear     tld/denoiser.py:107-109,114; transformer_blocks.py:65,71
//   layernorm_f32       Denoiser.norm on the 2-token condition           tld/denoiser.py:113,122
//   wq_kernel           folds cross-attention's q_linear into per-(token,head) d-vectors (see tld_rows.hip)
//   cast kernels        I/O dtype conversion at the C-ABI edge
//
// These run on "token rows": the set of distinct conditioning tokens of a call.  A plain forward has
// one noise row and one label row per sample; the sampler has one noise row per timestep and one label
// row per prompt plus the shared zero-label row, all prepared once before the 35-step loop.
#include "tld_common.h"
#include <hip/hip_fp16.h>

namespace tld {

namespace {

__global__ void sinusoid_kernel(const float* __restrict__ sigma, const float* __restrict__ angular,
                                float* __restrict__ out, int T, int half) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * half) return;
    const int t = i / half, k = i - t * half;
    const float ph = angular[k] * sigma[t];
    out[(size_t)t * 2 * half + k] = sinf(ph);
    out[(size_t)t * 2 * half + half + k] = cosf(ph);
}

// block: 8 token rows x 64 output columns; 4 waves, each wave 16 columns; lanes stride over K.
constexpr int LR = 8;
// blockIdx.z = layer for the batched launches: W then comes from the pointer table Wl and out advances by out_lstride
__global__ __launch_bounds__(256) void linear_f32_kernel(const float* __restrict__ in, int ldi,
                                                         const float* __restrict__ W,
                                                         const float* const* __restrict__ Wl,
                                                         const float* __restrict__ bias,
                                                         float* __restrict__ out, size_t out_lstride, int ldo, int T,
                                                         int K, int N, int act) {
    if (Wl) { W = Wl[blockIdx.z]; out += (size_t)blockIdx.z * out_lstride; }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xs = reinterpret_cast<float*>(smem);          // [LR][K]
    const int t0 = blockIdx.y * LR;
    for (int i = threadIdx.x; i < LR * K; i += 256) {
        const int r = i / K, k = i - r * K;
        const int t = t0 + r;
        xs[i] = t < T ? in[(size_t)t * ldi + k] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int c = 0; c < 16; ++c) {
        const int n = blockIdx.x * 64 + wid * 16 + c;
        if (n >= N) break;
        float acc[LR];
#pragma unroll
        for (int r = 0; r < LR; ++r) acc[r] = 0.f;
        const float* wrow = W + (size_t)n * K;
        for (int k = lane; k < K; k += 64) {
            const float w = wrow[k];
#pragma unroll
            for (int r = 0; r < LR; ++r) acc[r] += xs[r * K + k] * w;
        }
#pragma unroll
        for (int r = 0; r < LR; ++r) acc[r] = wave_sum(acc[r]);
        if (lane < LR) {
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < LR; ++r) if (lane == r) v = acc[r];
            const int t = t0 + lane;
            if (t < T) {
                v += bias ? bias[n] : 0.f;
                if (act == 1) v = gelu_erf(v);
                out[(size_t)t * ldo + n] = v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void layernorm_f32_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ g,
                                                            const float* __restrict__ b,
                                                            float* __restrict__ out, int M, int d) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float s = 0.f;
    for (int k = lane; k < d; k += 64) s += x[(size_t)row * d + k];
    const float mean = wave_sum(s) / (float)d;
    float q = 0.f;
    for (int k = lane; k < d; k += 64) { const float c = x[(size_t)row * d + k] - mean; q += c * c; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + kLnEps);
    for (int k = lane; k < d; k += 64)
        out[(size_t)row * d + k] = (x[(size_t)row * d + k] - mean) * rstd * g[k] + b[k];
}

// grid (heads, ceil(T/8)); thread j strides over d; k rows (64 features of this head) in LDS.
// blockIdx.z = layer: per-layer pointers from tables (Wq, gamma, beta), strided inputs / outputs
__global__ __launch_bounds__(256) void wq_kernel(const float* __restrict__ kmat, size_t k_lstride, int ldk,
                                                 const float* const* __restrict__ Wql,
                                                 const float* const* __restrict__ gammal,
                                                 const float* const* __restrict__ betal, float* __restrict__ wq,
                                                 size_t wq_lstride, float* __restrict__ bwq, size_t bwq_lstride, int T,
                                                 int H, int d) {
    const float* __restrict__ Wq = Wql[blockIdx.z];
    const float* __restrict__ gamma = gammal[blockIdx.z];
    const float* __restrict__ beta = betal[blockIdx.z];
    kmat += (size_t)blockIdx.z * k_lstride; wq += (size_t)blockIdx.z * wq_lstride; bwq += (size_t)blockIdx.z * bwq_lstride;
    __shared__ float ks[LR][64];
    __shared__ float red[4][LR];
    const int h = blockIdx.x, t0 = blockIdx.y * LR;
    for (int i = threadIdx.x; i < LR * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        const int t = t0 + r;
        ks[r][c] = t < T ? kmat[(size_t)t * ldk + h * 64 + c] * 0.125f : 0.f;
    }
    __syncthreads();
    float bacc[LR];
#pragma unroll
    for (int r = 0; r < LR; ++r) bacc[r] = 0.f;
    for (int j = threadIdx.x; j < d; j += 256) {
        float acc[LR];
#pragma unroll
        for (int r = 0; r < LR; ++r) acc[r] = 0.f;
        for (int c = 0; c < 64; ++c) {
            const float w = Wq[(size_t)(h * 64 + c) * d + j];
#pragma unroll
            for (int r = 0; r < LR; ++r) acc[r] += ks[r][c] * w;
        }
        const float gj = gamma[j], bj = beta[j];
#pragma unroll
        for (int r = 0; r < LR; ++r) {
            const int t = t0 + r;
            if (t < T) wq[((size_t)t * H + h) * d + j] = acc[r] * gj;
            bacc[r] += acc[r] * bj;
        }
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < LR; ++r) {
        const float tot = wave_sum(bacc[r]);
        if (lane == 0) red[wid][r] = tot;
    }
    __syncthreads();
    if (threadIdx.x < LR) {
        const int t = t0 + threadIdx.x;
        if (t < T) bwq[(size_t)t * H + h] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    }
}

__global__ void iota_kernel(int* __restrict__ dst, int n, int base) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = base + i;
}

__global__ void fill_bf16_kernel(bf16* __restrict__ dst, int64_t n, uint32_t seed, float scale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t h = (uint32_t)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    dst[i] = (bf16)(((float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale);
}

template <typename T>
__global__ void cast_to_f32_kernel(const T* __restrict__ src, float* __restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (float)src[i];
}
template <typename T>
__global__ void cast_from_f32_kernel(const float* __restrict__ src, T* __restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (T)src[i];
}

}  // namespace

void launch_sinusoid(const float* sigma, const float* angular, float* out, int T, int half, hipStream_t s) {
    const int n = T * half;
    hipLaunchKernelGGL(sinusoid_kernel, dim3((n + 255) / 256), dim3(256), 0, s, sigma, angular, out, T, half);
}

void launch_linear_f32(const float* in, int ldi, const float* W, const float* b, float* out, int ldo, int T,
                       int K, int N, int act, hipStream_t s) {
    dim3 grid((N + 63) / 64, (T + LR - 1) / LR);
    hipLaunchKernelGGL(linear_f32_kernel, grid, dim3(256), LR * K * sizeof(float), s, in, ldi, W,
                       (const float* const*)nullptr, b, out, (size_t)0, ldo, T, K, N, act);
}

// the same input rows through `layers` weight matrices in one launch: out[l] = in . W[l]^T (no bias)
void launch_linear_f32_layers(const float* in, int ldi, const float* const* Wl, float* out, size_t out_lstride,
                              int ldo, int T, int K, int N, int layers, hipStream_t s) {
    dim3 grid((N + 63) / 64, (T + LR - 1) / LR, layers);
    hipLaunchKernelGGL(linear_f32_kernel, grid, dim3(256), LR * K * sizeof(float), s, in, ldi, (const float*)nullptr,
                       Wl, (const float*)nullptr, out, out_lstride, ldo, T, K, N, 0);
}

void launch_layernorm_f32(const float* x, const float* g, const float* b, float* out, int M, int d,
                          hipStream_t s) {
    hipLaunchKernelGGL(layernorm_f32_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, g, b, out, M, d);
}

void launch_wq_layers(const float* k, size_t k_lstride, int ldk, const float* const* Wql, const float* const* gammal,
                      const float* const* betal, float* wq, size_t wq_lstride, float* bwq, size_t bwq_lstride, int T,
                      int heads, int d, int layers, hipStream_t s) {
    dim3 grid(heads, (T + LR - 1) / LR, layers);
    hipLaunchKernelGGL(wq_kernel, grid, dim3(256), 0, s, k, k_lstride, ldk, Wql, gammal, betal, wq, wq_lstride, bwq,
                       bwq_lstride, T, heads, d);
}

void launch_iota(int* dst, int n, int base, hipStream_t s) {
    hipLaunchKernelGGL(iota_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dst, n, base);
}

void launch_fill_bf16(bf16* dst, int64_t n, uint32_t seed, float scale, hipStream_t s) {
    hipLaunchKernelGGL(fill_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dst, n, seed, scale);
}

void launch_cast_to_f32(const void* src, int dtype, float* dst, int64_t n, hipStream_t s) {
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (dtype == 1) hipLaunchKernelGGL(cast_to_f32_kernel<bf16>, grid, block, 0, s, (const bf16*)src, dst, n);
    else if (dtype == 2) hipLaunchKernelGGL(cast_to_f32_kernel<__half>, grid, block, 0, s, (const __half*)src, dst, n);
    else hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, s);
}

void launch_cast_from_f32(const float* src, void* dst, int dtype, int64_t n, hipStream_t s) {
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (dtype == 1) hipLaunchKernelGGL(cast_from_f32_kernel<bf16>, grid, block, 0, s, src, (bf16*)dst, n);
    else if (dtype == 2) hipLaunchKernelGGL(cast_from_f32_kernel<__half>, grid, block, 0, s, src, (__half*)dst, n);
    else hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, s);
}

}  // namespace tld
