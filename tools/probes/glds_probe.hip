#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
// each wave-instruction: lane j -> LDS base + 16*j ; source per lane arbitrary (here: reversed order inside each 1 KB)
__global__ void k(const unsigned* __restrict__ src, unsigned* dst, int nchunk) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = wave; c < nchunk; c += 4) {     // chunk = 1 KB = 256 dwords
        const unsigned* g = src + c * 256 + (63 - lane) * 4;       // permuted source
        __builtin_amdgcn_global_load_lds(g, (lds_void*)(lds + c * 256), 16, 0, 0);
    }
    __syncthreads();
    for (int i = tid; i < nchunk * 256; i += 256) dst[i] = lds[i];
}
int main() {
    const int nchunk = 16, n = nchunk * 256;
    std::vector<unsigned> h(n), o(n);
    for (int i = 0; i < n; ++i) h[i] = i;
    unsigned *d, *e;
    hipMalloc(&d, n * 4); hipMalloc(&e, n * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(e, 0xff, n * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), n * 4, 0, d, e, nchunk);
    hipError_t err = hipDeviceSynchronize();
    hipMemcpy(o.data(), e, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int c = 0; c < nchunk; ++c) for (int j = 0; j < 64; ++j) for (int w = 0; w < 4; ++w) {
        unsigned want = c * 256 + (63 - j) * 4 + w;
        if (o[c * 256 + j * 4 + w] != want) { if (bad < 5) printf("mismatch c%d lane%d w%d got %u want %u\n", c, j, w, o[c*256+j*4+w], want); ++bad; }
    }
    printf("glds probe: err=%d bad=%d (first words %u %u %u %u %u)\n", (int)err, bad, o[0], o[1], o[4], o[252], o[256]);
    return bad != 0;
}
