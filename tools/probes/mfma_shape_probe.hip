// Does the MFMA shape matter on a power-limited chip?  Same bf16 FLOPs issued as 16x16x32 (the shape the library uses) and as
// 32x32x16 (half the operand-register reads per flop), random operands held in registers, every CU busy with two waves per SIMD;
// reports TFLOP/s and the effective shader clock (cycle counter vs the constant 100-MHz counter) of each.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/mfma_shape_probe.hip -o tools/probes/mfma_shape_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));

__device__ long long g_clk[4];

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void burn(const uint4* __restrict__ src, float* sink, int iters) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    bf16x8_t a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint4 u = src[(tid * 8 + i) & 0xffff], v = src[(tid * 8 + 4 + i) & 0xffff];
        a[i] = *reinterpret_cast<bf16x8_t*>(&u);
        b[i] = *reinterpret_cast<bf16x8_t*>(&v);
    }
    const long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    float out = 0.f;
    if constexpr (SHAPE == 16) {
        f32x4 c[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i + (i >> 2)) & 3], c[i], 0, 0, 0);      // 8 x 16384 FLOP
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) out += c[i][0] + c[i][3];
    } else if constexpr (SHAPE == 128) {      // block-scaled fp8 16x16x128 (random e4m3 bytes with the exponent kept mid-range)
        v8i a8[2], b8[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                a8[i][w] = (int)((__builtin_bit_cast(uint4, a[2 * i + (w >> 2)])[w & 3] & 0x87878787u) | 0x38383838u);
                b8[i][w] = (int)((__builtin_bit_cast(uint4, b[2 * i + (w >> 2)])[w & 3] & 0x87878787u) | 0x30303030u);
            }
        f32x4 c[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)      // 8 x 65536 FLOP
                c[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8[i & 1], b8[(i >> 1) & 1], c[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) out += c[i][0] + c[i][3];
    } else {
        f32x16 c[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[(i + 1) & 3], c[i], 0, 0, 0);                  // 4 x 32768 FLOP
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) out += c[i][0] + c[i][15];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_clk[0] = __builtin_readcyclecounter() - t0; g_clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
    if (out == 123.456f) sink[tid] = out;
}

template <int SHAPE>
int run(const uint4* src, float* sink, const char* what) {
    const int wgs = 256 * 2, iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((burn<SHAPE>), dim3(wgs), dim3(256), 0, 0, src, sink, iters);
        hipEventRecord(e1); CK(hipDeviceSynchronize());
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long clk[4]; CK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof clk));
        const double flop = (double)wgs * 4 * iters * 8 * (SHAPE == 128 ? 65536.0 : 16384.0);
        printf("%-17s %8.2f ms  %7.1f TFLOP/s  effective clock %.0f MHz\n", what, ms, flop / (ms * 1e-3) * 1e-12, clk[0] / (clk[1] * 0.01));
    }
    return 0;
}

int main() {
    std::vector<unsigned> h(65536 * 4);
    srand(1);
    for (auto& v : h) {      // random finite bf16 pairs (exponents around 1.0)
        const unsigned lo = 0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15), hi = 0x3f00 + (rand() & 0xff) + ((rand() & 1) << 15);
        v = lo | (hi << 16);
    }
    uint4* src; float* sink;
    CK(hipMalloc(&src, h.size() * 4)); CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMalloc(&sink, 256 * 512 * 4));
    if (run<16>(src, sink, "16x16x32")) return 1;
    if (run<32>(src, sink, "32x32x16")) return 1;
    if (run<16>(src, sink, "16x16x32")) return 1;
    if (run<32>(src, sink, "32x32x16")) return 1;
    if (run<128>(src, sink, "mx fp8 16x16x128")) return 1;
    return run<16>(src, sink, "16x16x32");
}
