// gemm_4phase.hip -- round 6 experiment: the ring of gemm_8phase.hip with MERGED phases (two 16-MFMA phases per K-tile, see the K loop).
// gemm_8phase.hip -- standalone bring-up of the counted-vmcnt half-tile ring (the "256^2 8-phase" structure of the local gfx950 HIP
// guide, cdna_hip_programming.md section 5) for  C[M,N] = A[M,K] . W[N,K]^T,  bf16 in / fp32 accumulate.  The shipped kernel's K loop
// (csrc/tld_gemm.hip) grew out of this file; it stays as the experiment + race screen for sync-structure edits.
//
//   tile 256 x 256 x 64, 8 waves as 2 (M) x 4 (N), wave tile 128 x 64, persistent workgroups (one per CU)
//   LDS: 8 half-tile slots of 16 KiB = 2 K-tiles x {B0, A0, B1, A1}.  A half-tile is 128 operand rows x 64 K (128-byte rows, 16-B chunk
//        index XOR ((row >> 1) & 7): conflict-free for the ds_read_b128 lane groups of both MFMA shapes), filled by 16 global_load_lds
//        pieces of 1 KiB (8 full 128-B lines each), two per wave.
//        Half-tile A_qa = the rows {wr * 128 + qa * 64 + [0, 64)} of the tile, B_qb = the W rows {wc * 64 + qb * 32 + [0, 32)}: quadrant
//        (qa, qb) of EVERY wave's 128 x 64 output reads exactly half-tiles A_qa and B_qb, and the wave tiles stay contiguous.
//   K loop: 8 phases per iteration (2 K-tiles).  Phase p: [ds_read the quadrant's operand subtile | stage ONE half-tile (2 glds per wave)]
//        s_barrier, lgkmcnt(0), 8 x v_mfma_f32_32x32x16_bf16 (one C quadrant x K = 64) at raised priority, s_barrier.
//        Quadrant order per K-tile: (0,0) (0,1) (1,1) (1,0); reads: B0+A0 | B1 | A1 | none (B0 stays in registers).
//        Waves 4-7 run one barrier behind waves 0-3, so the two waves of a SIMD alternate between the read and the MFMA half of a phase.
//   Prefetch: the half-tile stream runs 6 phases ahead of its first read: phase P stages stream element P + 6.
//        s_waitcnt vmcnt(6) in phases 4 and 8 only (three half-tiles stay in flight; never 0 in the main loop); a half-tile is read no
//        earlier than the phase after the wait that retired it.  WAR: a slot is restaged >= 2 phases after its last read, except B0
//        (1 phase), whose reads are issued first and retired by lgkmcnt(8) before the reading phase's first barrier.
//        The stream continues across tile boundaries (the next tile's first 7 half-tiles land under the epilogue).
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o gemm_8phase tools/ubench/gemm_8phase.hip && ./gemm_8phase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
#include <type_traits>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HT = 128 * BK * 2;                 // half-tile bytes (16 KiB)
constexpr int NSLOT = 8;
constexpr int SCR_OFF = NSLOT * HT;              // per-wave epilogue scratch behind the ring: 8 x 4 KiB
constexpr int LDS_BYTES = SCR_OFF + 8 * 4096;    // 160 KiB

__device__ __forceinline__ bf16x8 read_frag(const char* ht, int row, int kchunk) {
    return *reinterpret_cast<const bf16x8*>(ht + row * 128 + ((kchunk ^ ((row >> 1) & 7)) << 4));
}

// dbg: 1 = no fragment reads after the first K-tile, 2 = no tile DMA in the loop, 4 = no MFMAs
template <int dbg>
__global__ __launch_bounds__(512) void gemm8p_kernel(const bf16* __restrict__ A, const bf16* __restrict__ W, bf16* __restrict__ C,
                                                      int M, int N, int K, int store, unsigned long long* trace, int epi_work) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int grp = wr;                               // waves 4-7 run one barrier behind
    const int l31 = lane & 31, hi = lane >> 5;
    const int ntn = N / BN, ntiles = (M / BM) * ntn;
    const int nk = K / BK;

    // static XCD-aware schedule (as the shipped kernel): XCD x owns a contiguous eighth of the row-major tile order
    const int nblocks = gridDim.x;
    const int bid = blockIdx.x, xcd = bid & 7, lidx = bid >> 3;
    const int per_xcd_blocks = (nblocks + 7 - xcd) / 8;
    const int q8 = ntiles >> 3, r8 = ntiles & 7;
    const int xbase = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int xcount = q8 + (xcd < r8 ? 1 : 0);
    const int my_tiles = lidx < xcount ? (xcount - lidx + per_xcd_blocks - 1) / per_xcd_blocks : 0;
    if (my_tiles == 0) return;
    auto tile_coords = [&](int i, int& m0, int& n0) {
        const int tile = xbase + lidx + i * per_xcd_blocks;
        const int tm = tile / ntn;
        m0 = tm * BM; n0 = (tile - tm * ntn) * BN;
    };

    // DMA source offsets of this lane's two pieces per half-tile (32-bit; the half's row offset and the K offset are uniform)
    unsigned voffA[2], voffB[2];
    auto set_offsets = [&](int m0, int n0) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = (wid * 2 + q) * 8 + (ln >> 3);                 // row inside the half-tile image
            const unsigned c16 = (unsigned)(((ln & 7) ^ ((r >> 1) & 7)) * 16);
            voffA[q] = (unsigned)(m0 + (r >> 6) * 128 + (r & 63)) * (unsigned)(K * 2) + c16;      // + qa * 64 rows (uniform)
            voffB[q] = (unsigned)(n0 + (r >> 5) * 64 + (r & 31)) * (unsigned)(K * 2) + c16;       // + qb * 32 rows (uniform)
        }
    };
    // stream element i of a K-tile: 0 = B0, 1 = A0, 2 = B1, 3 = A1
    auto stage_ht = [&](int i, int tk, int slot) {
        const bool isA = i & 1;
        const int half = i >> 1;                                          // B0 A0 -> 0, B1 A1 -> 1
        const char* base = reinterpret_cast<const char*>(isA ? A : W) + (size_t)tk * (BK * 2)
                           + (size_t)half * (isA ? 64 : 32) * (size_t)(K * 2);
        char* dst = smem + slot * HT + wid * 2048;
        __builtin_amdgcn_global_load_lds((gptr_t)(base + (isA ? voffA[0] : voffB[0])), (lptr_t)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(base + (isA ? voffA[1] : voffB[1])), (lptr_t)(dst + 1024), 16, 0, 0);
    };

    int m0, n0;
    tile_coords(0, m0, n0);
    set_offsets(m0, n0);
    // prologue: K-tile 0 (4 half-tiles) + the first three of K-tile 1
#pragma unroll
    for (int h = 0; h < 7; ++h) stage_ht(h & 3, h >> 2, h);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    bool started = false;
    for (int it = 0; it < my_tiles; ++it) {
        int m0n = 0, n0n = 0;
        const bool has_next = it + 1 < my_tiles;
        if (has_next) tile_coords(it + 1, m0n, n0n);

        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        bf16x8 fa[2][4], fb0[4], fb1[4];                 // A quadrant subtile [ii][ks], B subtiles [ks]
        // dbg bit 3 (8): s_memtime trace of workgroup 0's first 6 tiles, 8 stamps per (tile, wave)
        auto stamp = [&](int slot) {
            if constexpr ((dbg & 8) != 0) {
                if (blockIdx.x == 0 && it < 6 && lane == 0) trace[(it * 8 + wid) * 8 + slot] = __builtin_amdgcn_s_memtime();
            }
        };
        stamp(0);
        if (grp) __builtin_amdgcn_s_barrier();           // stagger in
        // ---- merged phases: TWO phases of 16 MFMAs per K-tile instead of four of 8 (half the barriers).
        //   PA: read B0, A0, B1 (16 fragments) | stage A1 of the NEXT K-tile (2 pieces)          -> quadrants (0,0), (0,1)
        //   PB: read A1 (8 fragments)          | stage B0, A0, B1 of K-tile + 2 (6 pieces)       -> quadrants (1,1), (1,0)
        //   Fragment reads are retired (lgkmcnt 0) BEFORE the barrier that closes an R half: a slot is then free for anyone's DMA as soon as both groups
        //   have passed that barrier -- B0 / A0 / B1 of K-tile t are restaged (for t + 2) in PB(t), A1 of K-tile t - 1 (for t + 1) in PA(t).
        //   Counted waits before the barrier closing an M half: PA(t) needs A1(t) -- issued in PA(t - 1), 6 + 2 younger pieces -> vmcnt(8);
        //   PB(t) needs B0 / A0 / B1 of t + 1 -- issued in PB(t - 1), 2 + 6 younger pieces -> vmcnt(8).
        auto ktile = [&](int tk, auto ec) {
            constexpr int e = decltype(ec)::value;
            const char* kt = smem + e * 4 * HT;
            const bool drain = !has_next && tk >= nk - 2;
            // ---------------- PA
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fb0[ks] = read_frag(kt + 0 * HT, wc * 32 + l31, ks * 2 + hi);
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) fa[ii][ks] = read_frag(kt + 1 * HT, wr * 64 + ii * 32 + l31, ks * 2 + hi);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fb1[ks] = read_frag(kt + 2 * HT, wc * 32 + l31, ks * 2 + hi);
            {
                int t2 = tk + 1; bool doit = !(dbg & 2);
                if (t2 >= nk) { t2 -= nk; doit = doit && has_next; }
                if (doit) stage_ht(3, t2, (1 - e) * 4 + 3);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (!(dbg & 4)) {
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii) acc[ii][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb0[ks], fa[ii][ks], acc[ii][0], 0, 0, 0);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii) acc[ii][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb1[ks], fa[ii][ks], acc[ii][1], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
            }
            if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // ---------------- PB
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) fa[ii][ks] = read_frag(kt + 3 * HT, wr * 64 + ii * 32 + l31, ks * 2 + hi);
            {
                int t2 = tk + 2; bool doit = !(dbg & 2);
                if (t2 >= nk) {
                    if (tk == nk - 2 && has_next) set_offsets(m0n, n0n);
                    t2 -= nk; doit = doit && has_next;
                }
                if (doit) { stage_ht(0, t2, e * 4 + 0); stage_ht(1, t2, e * 4 + 1); stage_ht(2, t2, e * 4 + 2); }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (!(dbg & 4)) {
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii) acc[2 + ii][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb1[ks], fa[ii][ks], acc[2 + ii][1], 0, 0, 0);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii) acc[2 + ii][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb0[ks], fa[ii][ks], acc[2 + ii][0], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
            }
            if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        };
        for (int tk = 0; tk < nk; tk += 2) {
            ktile(tk, std::integral_constant<int, 0>{});
            ktile(tk + 1, std::integral_constant<int, 1>{});
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next tile's staged half-tiles (own pieces), before this tile's stores are issued
        if (!grp) __builtin_amdgcn_s_barrier();          // stagger out
        stamp(5);

        if (store >= 5) {
            // store modes 5 / 6 (plain / nontemporal): no LDS round trip.  Swapped accumulators: lane (l31, hi) holds, per 32 x 32 tile and
            // register quad rq, the 4 columns 8 rq + 4 hi + [0, 4) of row l31 -- 8 bytes.  One v_permlane32_swap per packed dword pair
            // exchanges quads between the two lane halves, after which lane (l31, hi) owns the 8 consecutive columns 16 k + 8 hi + [0, 8)
            // (k = 0, 1): 16-byte stores, 32 bytes per row per instruction.
            const int row0 = m0 + wr * 128, col0 = n0 + wc * 64;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2) {
                        unsigned x[2], y[2];
#pragma unroll
                        for (int w2 = 0; w2 < 2; ++w2) {
                            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
                            bf16x2 a, b;
                            a[0] = (bf16)acc[i][j2][(2 * k2) * 4 + 2 * w2]; a[1] = (bf16)acc[i][j2][(2 * k2) * 4 + 2 * w2 + 1];
                            b[0] = (bf16)acc[i][j2][(2 * k2 + 1) * 4 + 2 * w2]; b[1] = (bf16)acc[i][j2][(2 * k2 + 1) * 4 + 2 * w2 + 1];
                            const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
                            x[w2] = r[0]; y[w2] = r[1];
                        }
                        const u32x4 v = {x[0], x[1], y[0], y[1]};
                        u32x4* dst = reinterpret_cast<u32x4*>(C + (size_t)(row0 + i * 32 + l31) * N + col0 + j2 * 32 + 16 * k2 + 8 * hi);
                        if (store == 6) __builtin_nontemporal_store(v, dst); else *dst = v;
                    }
        } else if (store) {
            // swapped accumulators: lane = token row, registers = 4 consecutive columns per quad.  One 32-row x 64-col bf16 slab at a
            // time through the wave's scratch (128-B pitch, chunks XOR (row & 7)), then whole 128-B rows with 16-B stores.
            char* ws = smem + SCR_OFF + wid * 4096;
            const int row0 = m0 + wr * 128, col0 = n0 + wc * 64;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        if (store == 3) continue;
                        const int cl = j2 * 32 + 8 * rq + 4 * hi;
                        bf16x4 pk;
#pragma unroll
                        for (int e2 = 0; e2 < 4; ++e2) {
                            float v = acc[i][j2][rq * 4 + e2];
                            for (int wk = 0; wk < epi_work; ++wk) v = fmaf(v, 1.0001f, 0.0001f * v);   // emulated elementwise epilogue work (cf. gemm_2wg.hip)
                            pk[e2] = (bf16)v;
                        }
                        *reinterpret_cast<bf16x4*>(ws + l31 * 128 + ((((cl >> 3) ^ (l31 & 7)) << 4) | ((cl & 7) << 1))) = pk;
                    }
#pragma unroll
                for (int itr = 0; itr < 4; ++itr) {
                    const int idx = itr * 64 + lane;
                    const int rl = idx >> 3, ch = idx & 7;
                    // store modes (attribution): 1 = normal, 2 = LDS transpose only, 3 = global stores only (no LDS round trip), 4 = nontemporal stores
                    u32x4 v;
                    if (store == 3) v = u32x4{(unsigned)__float_as_uint(acc[i][0][itr]), (unsigned)__float_as_uint(acc[i][1][itr]), 0u, 0u};
                    else v = *reinterpret_cast<const u32x4*>(ws + rl * 128 + ((ch ^ (rl & 7)) << 4));
                    u32x4* dst = reinterpret_cast<u32x4*>(C + (size_t)(row0 + i * 32 + rl) * N + col0 + ch * 8);
                    if (store == 2) { if (v[0] == 0x12345678u && v[3] == 0x9abcdef0u) *dst = v; }
                    else if (store == 4) __builtin_nontemporal_store(v, dst);
                    else *dst = v;
                }
            }
        } else if (acc[0][0][0] == 12345.678f || acc[3][1][5] == 12345.678f || acc[1][1][3] == 777.f || acc[2][0][9] == 777.f) {
            C[0] = (bf16)1.f;                            // (keeps the accumulators alive)
        }
        stamp(6);
        m0 = m0n; n0 = n0n;
    }
}

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
    const int screen_runs = argc > 1 ? atoi(argv[1]) : 6;
    const int check_store = argc > 2 ? atoi(argv[2]) : 1;
    struct Shape { const char* name; int M, N, K; } shapes[] = {{"256 ", 256, 256, 256}, {"512 ", 512, 512, 512}, {"4k  ", 4096, 4096, 4096},
                                                                  {"qkv ", 32768, 2304, 768}, {"up  ", 32768, 3072, 768}, {"down", 32768, 768, 3072},
                                                                  {"8k  ", 8192, 8192, 8192}};
    typedef void (*kern_t)(const bf16*, const bf16*, bf16*, int, int, int, int, unsigned long long*, int);
    kern_t kerns[9] = {gemm8p_kernel<0>, gemm8p_kernel<1>, gemm8p_kernel<2>, nullptr, nullptr, gemm8p_kernel<5>, nullptr, nullptr, gemm8p_kernel<8>};
    unsigned long long* dtrace = nullptr;
    hipMalloc(&dtrace, 6 * 8 * 8 * 8);
    for (kern_t k : kerns) if (k) hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kerns[0]));
    printf("gemm8p_kernel: %d registers per thread, %zu bytes scratch, %d B LDS\n", fa.numRegs, (size_t)fa.localSizeBytes, LDS_BYTES);
    int ncu = 256;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, 0) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount; }
    for (auto& s : shapes) {
        const size_t na = (size_t)s.M * s.K, nw = (size_t)s.N * s.K, nc = (size_t)s.M * s.N;
        std::vector<uint16_t> ha(na), hw(nw);
        uint32_t st = 12345u;
        auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.0f - 1.0f; };     // uniform [-1, 1)
        for (auto& v : ha) v = f2bf(rnd());
        for (auto& v : hw) v = f2bf(rnd());
        bf16 *dA, *dW, *dC;
        hipMalloc(&dA, na * 2); hipMalloc(&dW, nw * 2); hipMalloc(&dC, nc * 2);
        hipMemcpy(dA, ha.data(), na * 2, hipMemcpyHostToDevice);
        hipMemcpy(dW, hw.data(), nw * 2, hipMemcpyHostToDevice);
        const int ntiles = (s.M / BM) * (s.N / BN);
        const int grid = ntiles < ncu ? ntiles : ncu;
        auto launch = [&](int store, int dbg) { hipLaunchKernelGGL(kerns[dbg], dim3(grid), dim3(512), LDS_BYTES, 0, dA, dW, dC, s.M, s.N, s.K, store & 15, dtrace, store >> 4); };       // store >> 4: emulated epilogue work
        // correctness (sampled fp64 reference, asymmetric random operands) + race screen (bitwise-identical output over repeated runs)
        std::vector<uint16_t> hc(nc), hc0;
        double worst = 0.0;
        int mismatching_runs = 0;
        for (int run = 0; run < screen_runs; ++run) {
            hipMemset(dC, 0xff, nc * 2);
            launch(check_store, 0);
            if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed: %s\n", s.name, hipGetErrorString(hipGetLastError())); return 1; }
            hipMemcpy(hc.data(), dC, nc * 2, hipMemcpyDeviceToHost);
            if (run == 0) {
                hc0 = hc;
                const int nsamp = 4000;
                for (int t = 0; t < nsamp; ++t) {
                    const size_t m = ((size_t)t * 7919 + 13) % s.M, n = ((size_t)t * 104729 + 7) % s.N;
                    double ref = 0.0;
                    for (int k = 0; k < s.K; ++k) ref += (double)bf2f(ha[m * s.K + k]) * (double)bf2f(hw[n * s.K + k]);
                    const double got = bf2f(hc[m * s.N + n]);
                    worst = fmax(worst, fabs(got - ref) / (fabs(ref) + 0.02 * sqrt((double)s.K)));
                }
            } else if (memcmp(hc.data(), hc0.data(), nc * 2) != 0) ++mismatching_runs;
        }
        printf("%s M=%d N=%d K=%d: max rel err %.2e over 4000 samples; %d of %d repeat runs differ bitwise\n", s.name, s.M, s.N, s.K, worst,
               mismatching_runs, screen_runs - 1);
        if (s.M < 1024) { hipFree(dA); hipFree(dW); hipFree(dC); continue; }
        struct { const char* label; int store, dbg, zeros; } modes[] = {{"full (with stores)", 1, 0, 0}, {"no stores         ", 0, 0, 0}, {"no stores, zeros  ", 0, 0, 1},
                                                                        {"no stores, no DMA ", 0, 2, 0}, {"no stores, no frag", 0, 1, 0}, {"DMA + barriers only", 0, 5, 0},
                                                                        {"nontemporal stores", 4, 0, 0}};
        for (auto& md : modes) {
            if (md.zeros) { hipMemset(dA, 0, na * 2); hipMemset(dW, 0, nw * 2); }
            else { hipMemcpy(dA, ha.data(), na * 2, hipMemcpyHostToDevice); hipMemcpy(dW, hw.data(), nw * 2, hipMemcpyHostToDevice); }
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            for (int i = 0; i < 3; ++i) launch(md.store, md.dbg);
            hipEventRecord(e0, 0);
            const int iters = 20;
            for (int i = 0; i < iters; ++i) launch(md.store, md.dbg);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / iters, tf = 2.0 * s.M * s.N * s.K / (us * 1e-6) / 1e12;
            printf("  %s %s: %8.1f us  %7.1f TFLOP/s (%.1f%% of 2500)\n", s.name, md.label, us, tf, tf / 25.0);
        }
        if (s.K == 768 || s.M == 4096) {
            for (int store = 0; store < 2; ++store) {
                hipMemcpy(dA, ha.data(), na * 2, hipMemcpyHostToDevice); hipMemcpy(dW, hw.data(), nw * 2, hipMemcpyHostToDevice);
                hipMemset(dtrace, 0, 6 * 8 * 8 * 8);
                for (int i = 0; i < 3; ++i) launch(store, 8);
                hipDeviceSynchronize();
                unsigned long long ht[6 * 8 * 8];
                hipMemcpy(ht, dtrace, sizeof(ht), hipMemcpyDeviceToHost);
                printf("  trace %s store=%d (workgroup 0; cycles since the tile-0 start of wave 0): per tile, waves 0 and 4: start | p4 wait in/out | p8 wait in/out | K loop end | epilogue end\n", s.name, store);
                const unsigned long long t0 = ht[0];
                for (int it = 0; it < 6; ++it)
                    for (int w = 0; w < 8; w += 4) {
                        const unsigned long long* r = ht + (it * 8 + w) * 8;
                        if (!r[0]) continue;
                        printf("    tile %d w%d: %7lld | %7lld %7lld (+%lld) | %7lld %7lld (+%lld) | %7lld | %7lld\n", it, w, (long long)(r[0] - t0), (long long)(r[1] - t0), (long long)(r[2] - t0),
                               (long long)(r[2] - r[1]), (long long)(r[3] - t0), (long long)(r[4] - t0), (long long)(r[4] - r[3]), (long long)(r[5] - t0), (long long)(r[6] - t0));
                    }
            }
        }
        hipFree(dA); hipFree(dW); hipFree(dC);
    }
    return 0;
}
