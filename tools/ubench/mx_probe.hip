// probe of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 operands): which (lane, byte) of the A / B operands is which
// (row, k), and which operand elements a lane's scale byte applies to.  Single wave, host-driven one-hot experiments.
// build: hipcc --offload-arch=gfx950 -O3 mx_probe.hip -o mx_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int OPA, int OPB>
__global__ void k(const int* a, const int* b, const int* sa, const int* sb, float* c) {
    const int l = threadIdx.x;
    i32x8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = a[l * 8 + i]; bv[i] = b[l * 8 + i]; }
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 0, 0, OPA, sa[l], OPB, sb[l]);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        c[row * 32 + col] = acc[r];
    }
}

struct Dev {
    int *a, *b, *sa, *sb; float* c;
    std::vector<int> ha, hb, hsa, hsb; std::vector<float> hc;
    Dev() : ha(512), hb(512), hsa(64), hsb(64), hc(1024) {
        hipMalloc(&a, 2048); hipMalloc(&b, 2048); hipMalloc(&sa, 256); hipMalloc(&sb, 256); hipMalloc(&c, 4096);
    }
    void fill(std::vector<int>& v, unsigned char byte) { memset(v.data(), byte, v.size() * 4); }
    void setbyte(std::vector<int>& v, int lane, int j, unsigned char byte) { ((unsigned char*)v.data())[lane * 32 + j] = byte; }
    template <int OPA, int OPB> void run() {
        hipMemcpy(a, ha.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 2048, hipMemcpyHostToDevice);
        hipMemcpy(sa, hsa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(sb, hsb.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL((k<OPA, OPB>), dim3(1), dim3(64), 0, 0, a, b, sa, sb, c);
        hipMemcpy(hc.data(), c, 4096, hipMemcpyDeviceToHost);
    }
};

int main() {
    Dev d;
    const unsigned char ONE = 0x38;                      // e4m3 1.0
    auto unit_scales = [&]() { for (int i = 0; i < 64; ++i) { d.hsa[i] = 0x7f7f7f7f; d.hsb[i] = 0x7f7f7f7f; } };
    // 1. rows of A bytes: B = ones
    unit_scales();
    d.fill(d.hb, ONE);
    int bad_rows = 0;
    for (int L = 0; L < 64; ++L)
        for (int j = 0; j < 32; j += 5) {
            d.fill(d.ha, 0); d.setbyte(d.ha, L, j, ONE);
            d.run<0, 0>();
            int row = -1, cnt = 0;
            for (int r = 0; r < 32; ++r) if (d.hc[r * 32] != 0.f) { row = r; ++cnt; }
            if (cnt != 1 || row != (L & 31)) { ++bad_rows; if (bad_rows < 6) printf("A lane %d byte %d -> row %d (cnt %d, val %g)\n", L, j, row, cnt, row >= 0 ? d.hc[row * 32] : 0.f); }
        }
    printf("[1] A (lane, byte) -> row == lane %% 32 : %s\n", bad_rows ? "NO" : "yes");
    // 2. k equivalence between A and B
    printf("[2] for A(lane La, byte j): B (lane, byte) with the same k   (expect same half, same byte)\n");
    for (int La : {0, 32})
        for (int j : {0, 1, 7, 8, 15, 16, 31}) {
            d.fill(d.ha, 0); d.setbyte(d.ha, La, j, ONE);
            printf("  A(%2d,%2d):", La, j);
            for (int Lb : {0, 32})
                for (int jb = 0; jb < 32; ++jb) {
                    d.fill(d.hb, 0); d.setbyte(d.hb, Lb, jb, ONE);
                    d.run<0, 0>();
                    if (d.hc[(La & 31) * 32 + (Lb & 31)] != 0.f) printf(" B(%d,%d)", Lb, jb);
                }
            printf("\n");
        }
    // 3. scale_a: lane Ls, byte bs doubled; A = one-hot, B = ones: which A elements are doubled (opsel 0 .. 3)
    printf("[3] scale_a byte semantics: A one-hot (lane, byte); scale VGPR of lane Ls has byte bs = 2.0; result 2 means scaled\n");
    d.fill(d.hb, ONE);
    auto probe_scale = [&](int opsel, int Ls, int bs) {
        unit_scales();
        ((unsigned char*)d.hsa.data())[Ls * 4 + bs] = 0x80;
        printf("  opsel_a %d, scale lane %2d byte %d: scaled A elements:", opsel, Ls, bs);
        int shown = 0;
        for (int L = 0; L < 64; ++L)
            for (int j = 0; j < 32; ++j) {
                d.fill(d.ha, 0); d.setbyte(d.ha, L, j, ONE);
                if (opsel == 0) d.run<0, 0>(); else if (opsel == 1) d.run<1, 0>(); else if (opsel == 2) d.run<2, 0>(); else d.run<3, 0>();
                if (d.hc[(L & 31) * 32] == 2.f && shown++ < 40) printf(" (%d,%d)", L, j);
            }
        printf("  [%d total]\n", shown);
    };
    probe_scale(0, 0, 0); probe_scale(0, 32, 0); probe_scale(0, 5, 0); probe_scale(0, 0, 1);
    probe_scale(1, 0, 1); probe_scale(2, 0, 2); probe_scale(2, 32, 2); probe_scale(3, 0, 3); probe_scale(2, 0, 0);
    // 4. scale_b likewise (B one-hot, A ones)
    printf("[4] scale_b: B one-hot; scale VGPR of lane Ls byte bs = 2.0\n");
    d.fill(d.ha, ONE);
    auto probe_scale_b = [&](int opsel, int Ls, int bs) {
        unit_scales();
        ((unsigned char*)d.hsb.data())[Ls * 4 + bs] = 0x80;
        printf("  opsel_b %d, scale lane %2d byte %d: scaled B elements:", opsel, Ls, bs);
        int shown = 0;
        for (int L = 0; L < 64; ++L)
            for (int j = 0; j < 32; ++j) {
                d.fill(d.hb, 0); d.setbyte(d.hb, L, j, ONE);
                if (opsel == 0) d.run<0, 0>(); else d.run<0, 2>();
                if (d.hc[L & 31] == 2.f && shown++ < 40) printf(" (%d,%d)", L, j);
            }
        printf("  [%d total]\n", shown);
    };
    probe_scale_b(0, 0, 0); probe_scale_b(0, 32, 0); probe_scale_b(2, 0, 2);
    return 0;
}
