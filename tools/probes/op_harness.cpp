// Bare C++ caller of libfs2_hip.so's fs2_op_conv_gemm (no Python / torch in the process): decoder FFN size, mx arithmetic.
//   g++ -O2 -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ tools/probes/op_harness.cpp -o tools/probes/op_harness.bin -L fastspeech2_amd -lfs2_hip -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,'$ORIGIN/../../fastspeech2_amd'
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "fs2.h"
int main(int argc, char** argv) {
    const int R = 30208, C = 384, N = 1024, k = 9;
    const int prec = argc > 1 ? atoi(argv[1]) : FS2_PREC_MIX_MX;
    std::vector<float> hx((size_t)R * C), hw((size_t)N * C * k), hb(N, 0.1f);
    for (auto& v : hx) { float s = 0; for (int i = 0; i < 12; ++i) s += (float)rand() / RAND_MAX; v = s - 6.f; }
    for (auto& v : hw) v = ((float)rand() / RAND_MAX * 2.f - 1.f) / sqrtf((float)C * k);
    float *x, *w, *b, *y;
    hipMalloc(&x, hx.size() * 4); hipMalloc(&w, hw.size() * 4); hipMalloc(&b, N * 4); hipMalloc(&y, (size_t)R * N * 4);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice);
    fs2_set_option("FS2_BM", 256);
    fs2_op_gemm_args a = {};
    a.struct_size = sizeof a; a.R = R; a.C = C; a.N = N; a.ktaps = k; a.precision = prec; a.x = x; a.w = w; a.bias = b; a.act_post = 1; a.y = y; a.ln_eps = 1e-5f;
    for (int i = 0; i < 3; ++i) { int rc = fs2_op_conv_gemm(nullptr, &a); if (rc) { printf("rc %d %s\n", rc, fs2_last_error(nullptr)); return 1; } }
    hipDeviceSynchronize();
    std::vector<float> hy(1024);
    hipMemcpy(hy.data(), y, 4096, hipMemcpyDeviceToHost);
    printf("y[0..3] = %g %g %g %g\n", hy[0], hy[1], hy[2], hy[3]);
    return 0;
}
