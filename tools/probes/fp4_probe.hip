// What the mix_mx4 kernels rely on (round 6), measured before they were written:
//  1. v_cvt_scalef32_pk_fp4_f32 (__builtin_amdgcn_cvt_scalef32_pk_fp4_f32(old, a, b, scale, byte)): which byte / nibble receives which argument,
//     and whether `scale` divides (quantise) or multiplies;
//  2. v_mfma_scale_f32_16x16x128_f8f6f4 with cbsz = blgp = 4 (e2m1 operands in registers 0-3 of the eight): lane (i = l & 15, g = l >> 4)
//     supplies the 32 values of row / column i in k-block g -- the element order INSIDE the block only has to be the same for A and B --, the
//     E8M0 scale of that block comes from the lane's own scale register, byte op_sel (0..3).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/fp4_probe.hip -o tools/probes/fp4_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void cvt_kernel(const float* x, float scale, unsigned* out) {
    unsigned w = 0xffffffffu;
    w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, x[0], x[1], scale, 0);
    w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, x[2], x[3], scale, 2);
    out[0] = w;
}
// A, B: [16][128] nibbles packed as [16 rows][4 blocks][16 bytes]; sa, sb: [16][4] scale bytes, placed in byte `sel` of the scale register
template <int SEL>
__global__ void mfma_kernel(const unsigned char* A, const unsigned char* B, const unsigned char* sa, const unsigned char* sb, float* D) {
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    const int4 av = *reinterpret_cast<const int4*>(A + (i * 4 + g) * 16), bv = *reinterpret_cast<const int4*>(B + (i * 4 + g) * 16);
    const v8i_t a8 = v8i_t{av.x, av.y, av.z, av.w, 0, 0, 0, 0}, b8 = v8i_t{bv.x, bv.y, bv.z, bv.w, 0, 0, 0, 0};
    const int ra = ((int)sa[i * 4 + g] << (8 * SEL)) | (SEL ? 0x7f : 0x7f00), rb = ((int)sb[i * 4 + g] << (8 * SEL)) | (SEL ? 0x7f : 0x7f00);
    f32x4 c = f32x4{0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, c, 4, 4, SEL, ra, SEL, rb);
    for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + i] = c[r];      // C/D layout of every 16x16 MFMA: col = l & 15, row = 4 (l >> 4) + r
}
static const float kFp4[16] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f, -0.f, -0.5f, -1.f, -1.5f, -2.f, -3.f, -4.f, -6.f};
int main() {
    float hx[4] = {0.5f, 1.5f, -3.f, 6.f};
    float* dx; unsigned* dout;
    CK(hipMalloc(&dx, 16)); CK(hipMalloc(&dout, 4));
    CK(hipMemcpy(dx, hx, 16, hipMemcpyHostToDevice));
    for (float sc : {1.f, 2.f, 0.5f}) {
        hipLaunchKernelGGL(cvt_kernel, dim3(1), dim3(1), 0, 0, dx, sc, dout);
        unsigned w; CK(hipMemcpy(&w, dout, 4, hipMemcpyDeviceToHost));
        printf("cvt_scalef32_pk_fp4_f32(0xffffffff; (0.5, 1.5) -> byte 0, (-3, 6) -> byte 2; scale %.1f) = %08x : byte 0 = [lo %g, hi %g], byte 2 = [lo %g, hi %g]\n", sc, w,
               kFp4[w & 15], kFp4[(w >> 4) & 15], kFp4[(w >> 16) & 15], kFp4[(w >> 20) & 15]);
    }
    // MFMA: random nibbles and random scale bytes in [120, 134]
    srand(7);
    std::vector<unsigned char> A(16 * 64), B(16 * 64), sa(64), sb(64);
    for (auto& v : A) v = rand() & 0xff;
    for (auto& v : B) v = rand() & 0xff;
    for (auto& v : sa) v = 120 + rand() % 15;
    for (auto& v : sb) v = 120 + rand() % 15;
    unsigned char *dA, *dB, *dsa, *dsb; float* dD;
    CK(hipMalloc(&dA, A.size())); CK(hipMalloc(&dB, B.size())); CK(hipMalloc(&dsa, 64)); CK(hipMalloc(&dsb, 64)); CK(hipMalloc(&dD, 1024));
    CK(hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dsa, sa.data(), 64, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb.data(), 64, hipMemcpyHostToDevice));
    auto nib = [](const std::vector<unsigned char>& M, int row, int k) { const unsigned char b = M[row * 64 + k / 2]; return kFp4[(k & 1) ? (b >> 4) : (b & 15)]; };
    int bad_total = 0;
    for (int sel = 0; sel < 2; ++sel) {
        if (sel == 0) hipLaunchKernelGGL(mfma_kernel<0>, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
        else hipLaunchKernelGGL(mfma_kernel<1>, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
        std::vector<float> D(256);
        CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
        int bad = 0; double worst = 0;
        for (int m = 0; m < 16; ++m)
            for (int n = 0; n < 16; ++n) {
                double ref = 0;
                for (int g = 0; g < 4; ++g) {
                    double s = 0;
                    for (int k = 0; k < 32; ++k) s += (double)nib(A, m, 32 * g + k) * nib(B, n, 32 * g + k);
                    ref += s * ldexp(1.0, sa[m * 4 + g] - 127) * ldexp(1.0, sb[n * 4 + g] - 127);
                }
                const double d = fabs(ref - D[m * 16 + n]);
                worst = fmax(worst, d / fmax(1.0, fabs(ref)));
                if (d > 1e-5 * fmax(1.0, fabs(ref))) ++bad;
            }
        printf("mfma_scale 16x16x128, e2m1 x e2m1, block scales from the lanes' own registers (byte %d, op_sel %d): %d of 256 outputs off (worst relative %.2e)\n", sel, sel, bad, worst);
        bad_total += bad;
    }
    printf("fp4 probe: %s\n", bad_total ? "FAILED" : "ok");
    return bad_total ? 1 : 0;
}
