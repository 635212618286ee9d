// Probe of the gfx950 block-scaled fp8 MFMA (v_mfma_scale_f32_16x16x128_f8f6f4) as the carrier of the two low-precision correction
// terms of a split product (fp16 hi*hi + [ra8*wh8 + ah8*rw8] on MX-fp8):
//   (1) operand / scale layout check against a scalar reference computed in the same kernel launch,
//   (2) issue rate of the scaled K = 128 fp8 MFMA vs the K = 32 fp16 MFMA (chip-wide, independent accumulators).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/mx_probe.hip -o tools/probes/mx_probe.bin && tools/probes/mx_probe.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// OCP e4m3fn decode (no inf; 0x7f / 0xff = NaN)
__host__ __device__ inline float e4m3(unsigned char b) {
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -v : v;
}

// A [16][128] fp8, B [128][16] fp8 (stored B^T [16][128]), scale bytes sa [16][4], sb [16][4] (E8M0: 2^(byte - 127)) -> C [16][16]
__global__ void check(const unsigned char* A, const unsigned char* Bt, const unsigned char* sa, const unsigned char* sb, float* C, float* Cref) {
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    v8i a, b;
    for (int w = 0; w < 8; ++w) {
        a[w] = *reinterpret_cast<const int*>(A + i * 128 + 32 * g + 4 * w);        // lane (i, g): row i, k = 32 g .. 32 g + 31
        b[w] = *reinterpret_cast<const int*>(Bt + i * 128 + 32 * g + 4 * w);       // lane (j = i, g): column j, same k range
    }
    const int scale_a = sa[i * 4 + g], scale_b = sb[i * 4 + g];                    // this lane's block scale in byte 0 (opsel 0)
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);
    for (int r = 0; r < 4; ++r) C[(4 * g + r) * 16 + i] = c[r];                    // C/D: col = l & 15, row = 4 (l >> 4) + reg
    // scalar reference: thread l computes 4 elements
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * g + r, col = i;
        double acc = 0.0;
        for (int k = 0; k < 128; ++k)
            acc += (double)e4m3(A[row * 128 + k]) * ldexp(1.0, (int)sa[row * 4 + k / 32] - 127) * (double)e4m3(Bt[col * 128 + k]) * ldexp(1.0, (int)sb[col * 4 + k / 32] - 127);
        Cref[row * 16 + col] = (float)acc;
    }
}

// Which k values does lane L's scale apply to?  A (or B) data: lane group g, 16-byte half h of its 32 bytes = 2^(2g + h) (8 distinct
// powers of two), the other operand = 1; for every L the scale is 2.0 on lane L only (all four bytes of the register).  The increase of
// C over the all-ones-scale result, divided by 16, is the bit mask of the (g, h) pieces that scale multiplies.
__global__ void scale_map(float* out) {      // out [2][64][16][16]
    const int l = threadIdx.x, g = l >> 4;
    const int ones = 0x38383838;               // e4m3 1.0 x4
    v8i a, b;
    for (int w = 0; w < 8; ++w) {
        const int e = 7 + 2 * g + (w >> 2);    // exponent field of 2^(2g + h)
        a[w] = (e << 3) * 0x01010101;
        b[w] = ones;
    }
    for (int which = 0; which < 2; ++which)
        for (int L = 0; L < 64; ++L) {
            const int sc = (l == L) ? 0x80808080 : 0x7f7f7f7f;
            f32x4 c = {0.f, 0.f, 0.f, 0.f};
            if (which == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sc, 0, 0x7f7f7f7f);
            else c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b, a, c, 0, 0, 0, 0x7f7f7f7f, 0, sc);
            for (int r = 0; r < 4; ++r) out[((which * 64 + L) * 16 + 4 * g + r) * 16 + (l & 15)] = c[r];
        }
}

template <int MODE>     // 0: fp16 16x16x32, 1: scaled fp8 16x16x128
__global__ __launch_bounds__(256) void rate(int iters, float* out) {
    f32x4 c[4];
    for (int j = 0; j < 4; ++j) c[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    v8i a, b;
    for (int w = 0; w < 8; ++w) { a[w] = 0x38383838 + threadIdx.x * 0x01010101 * (w & 1); b[w] = 0x30303030 + w; }
    f16x8 ha, hb;
    for (int w = 0; w < 8; ++w) { ha[w] = (_Float16)(0.01f * (threadIdx.x % 13 + w)); hb[w] = (_Float16)(0.02f * (w + 1)); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (MODE == 0) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c[j], 0, 0, 0);
            else c[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c[j], 0, 0, 0, 120, 0, 125);
        }
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];
    if (s == 12345.678f) out[0] = s;
}

// host reference under a hypothesis about the operand layout: kmap(g, p) = the k index of byte p (0..31) of lane group g
static double try_layout(const std::vector<unsigned char>& A, const std::vector<unsigned char>& Bt, const std::vector<unsigned char>& sa,
                         const std::vector<unsigned char>& sb, const float* C, int hyp, bool scales) {
    double worst = 0, mag = 0;
    for (int row = 0; row < 16; ++row)
        for (int col = 0; col < 16; ++col) {
            double acc = 0;
            for (int g = 0; g < 4; ++g)
                for (int p = 0; p < 32; ++p) {
                    // the kernel loaded byte p of lane group g from memory column 32 g + p; the hypothesis says which k that byte IS
                    // (only matters for pairing A and B bytes, which share the mapping -> any bijection gives the same dot product),
                    // so what can differ is which scale applies: hyp 0: scale index = g; hyp 1: scale index = p / 8 (interleaved)
                    const int kmem = 32 * g + p;
                    const int sidx = hyp == 0 ? g : (p / 8);
                    const double fa = scales ? ldexp(1.0, (int)sa[row * 4 + sidx] - 127) : 1.0, fb = scales ? ldexp(1.0, (int)sb[col * 4 + sidx] - 127) : 1.0;
                    acc += (double)e4m3(A[row * 128 + kmem]) * fa * (double)e4m3(Bt[col * 128 + kmem]) * fb;
                }
            worst = fmax(worst, fabs(acc - C[row * 16 + col])); mag = fmax(mag, fabs(acc));
        }
    return worst / mag;
}

int main() {
    std::vector<unsigned char> A(16 * 128), Bt(16 * 128), sa(64), sb(64);
    srand(3);
    for (auto& v : A) v = (unsigned char)(rand() & 0xff);
    for (auto& v : Bt) v = (unsigned char)(rand() & 0xff);
    for (auto& v : A) if ((v & 0x7f) == 0x7f) v ^= 1;          // no NaN encodings
    for (auto& v : Bt) if ((v & 0x7f) == 0x7f) v ^= 1;
    for (auto& v : sa) v = (unsigned char)(120 + rand() % 12);
    for (auto& v : sb) v = (unsigned char)(118 + rand() % 12);
    unsigned char *dA, *dB, *dsa, *dsb; float *dC, *dR;
    CK(hipMalloc(&dA, A.size())); CK(hipMalloc(&dB, Bt.size())); CK(hipMalloc(&dsa, 64)); CK(hipMalloc(&dsb, 64));
    CK(hipMalloc(&dC, 1024)); CK(hipMalloc(&dR, 1024));
    CK(hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bt.data(), Bt.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dsa, sa.data(), 64, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb.data(), 64, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dC, dR);
    CK(hipDeviceSynchronize());
    float C[256], R[256];
    CK(hipMemcpy(C, dC, 1024, hipMemcpyDeviceToHost)); CK(hipMemcpy(R, dR, 1024, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0;
    for (int i = 0; i < 256; ++i) { worst = fmax(worst, fabs((double)C[i] - R[i])); scale = fmax(scale, fabs((double)R[i])); }
    printf("layout check: max |mfma - reference| = %.3e (reference magnitude up to %.3e)  -> %s\n", worst, scale, worst <= 1e-5 * scale ? "LAYOUT CONFIRMED" : "MISMATCH");
    printf("  sample C[0][0..3] = %.6g %.6g %.6g %.6g | ref %.6g %.6g %.6g %.6g\n", C[0], C[1], C[2], C[3], R[0], R[1], R[2], R[3]);
    printf("  relative mismatch by hypothesis (with the lane's scale as given): per-lane-group scale %.3e | interleaved %.3e | scales ignored %.3e\n",
           try_layout(A, Bt, sa, sb, C, 0, true), try_layout(A, Bt, sa, sb, C, 1, true), try_layout(A, Bt, sa, sb, C, 0, false));
    {   // second launch with every scale = 127 (1.0): isolates the data layout from the scale semantics
        std::vector<unsigned char> one(64, 127);
        CK(hipMemcpy(dsa, one.data(), 64, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, one.data(), 64, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dC, dR);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(C, dC, 1024, hipMemcpyDeviceToHost)); CK(hipMemcpy(R, dR, 1024, hipMemcpyDeviceToHost));
        double w2 = 0, m2 = 0;
        for (int i = 0; i < 256; ++i) { w2 = fmax(w2, fabs((double)C[i] - R[i])); m2 = fmax(m2, fabs((double)R[i])); }
        printf("  all scales = 1.0: max |mfma - reference| / magnitude = %.3e  (C[0][0] %.6g ref %.6g)\n", w2 / m2, C[0], R[0]);
        // scale semantics: A scale byte 128 (2.0) on every lane, B 127
        std::vector<unsigned char> two(64, 128);
        CK(hipMemcpy(dsa, two.data(), 64, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dC, dR);
        CK(hipDeviceSynchronize());
        float C2[256];
        CK(hipMemcpy(C2, dC, 1024, hipMemcpyDeviceToHost));
        printf("  scale_a byte 128 on all lanes: C[0][0] = %.6g (x%.3f of the unscaled)\n", C2[0], C2[0] / C[0]);
    }
    {
        float* dM; CK(hipMalloc(&dM, 2 * 64 * 256 * 4));
        hipLaunchKernelGGL(scale_map, dim3(1), dim3(64), 0, 0, dM);
        CK(hipDeviceSynchronize());
        std::vector<float> M(2 * 64 * 256);
        CK(hipMemcpy(M.data(), dM, M.size() * 4, hipMemcpyDeviceToHost));
        const double base = 16.0 * 255;
        for (int which = 0; which < 2; ++which) {
            printf("  scale_%c: lane -> (row / column index, bit mask of the k-blocks) its byte-0 scale applies to:", which == 0 ? 'a' : 'b');
            for (int L = 0; L < 64; ++L) {
                int idx = -1, blk = -1;
                for (int r = 0; r < 16 && idx < 0; ++r)
                    for (int c = 0; c < 16; ++c) {
                        const double v = which == 0 ? M[((which * 64 + L) * 16 + r) * 16 + c] : M[((which * 64 + L) * 16 + c) * 16 + r];
                        const double d = v - base;
                        if (fabs(d) > 0.5) { idx = r; blk = (int)lround(d / 16.0); break; }      // bit 2g + h set: the scale multiplies half h of lane group g
                    }
                if (L % 16 == 0) printf("\n     lanes %2d-%2d:", L, L + 15);
                printf(" (%d,0x%02x)", idx, blk);
            }
            printf("\n");
        }
    }
    // issue rate: 256 CUs x 4 waves/SIMD... launch 2048 workgroups of 256 threads (8 per CU), 4 independent accumulators per wave
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(2048), dim3(256), 0, 0, iters, dC);
            else hipLaunchKernelGGL(rate<1>, dim3(2048), dim3(256), 0, 0, iters, dC);
            hipEventRecord(e1); CK(hipDeviceSynchronize());
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double mfmas = 2048.0 * 4 * iters * 4;
            const double flop = mfmas * 2.0 * 16 * 16 * (mode == 0 ? 32 : 128);
            printf("%s: %.2f ms, %.1f TFLOP/s, %.2f ns per MFMA per SIMD-slot (%.1f Ginstr/s chip-wide)\n", mode == 0 ? "fp16 16x16x32       " : "scaled fp8 16x16x128",
                   ms, flop / ms / 1e9, ms * 1e6 / (mfmas / (256 * 4)), mfmas / ms / 1e6);
        }
    }
    return 0;
}
