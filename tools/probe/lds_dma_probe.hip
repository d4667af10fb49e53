// GPU box probe: where do the lanes of `buffer_load_dwordx4 ... lds` / `global_load_lds_dwordx4` land in LDS, does the destination
// reach beyond 64 KiB, and is soffset part of the buffer range check?  hipcc --offload-arch=gfx950 -O2 -o lds_dma_probe lds_dma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef const __attribute__((address_space(1))) void* gvp_t;
typedef __attribute__((address_space(3))) void* lvp_t;

__global__ void probe(const float* src, int n_floats, float* out, int lds_off, int perm, int soff) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 40 * 1024; i += 64) smem[i] = -1.0f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, n_floats * 4, 0x00020000);
    char* dst = reinterpret_cast<char*>(smem) + lds_off;
    const int l2 = perm ? (lane ^ 5) : lane;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lvp_t)dst, 16, l2 * 16, soff, 0, 0);
    __builtin_amdgcn_global_load_lds((gvp_t)(src + 1024 + l2 * 4), (lvp_t)(dst + 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 512; i += 64) out[i] = smem[lds_off / 4 + i];
    for (int i = lane; i < 512; i += 64) out[512 + i] = smem[(lds_off & 0xffff) / 4 + i];       // where a 16-bit destination would have put it
}

int main() {
    const int N = 4096;
    std::vector<float> h(N);
    for (int i = 0; i < N; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, N * 4); hipMalloc(&o, 1024 * 4);
    hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    struct { int off, perm, soff, n; } cases[] = {{0, 0, 0, N}, {0, 1, 0, N}, {0, 0, 512, N}, {70 * 1024, 0, 0, N}, {0, 0, 0, 128}, {0, 0, 256, 128 + 64}};
    for (auto c : cases) {
        probe<<<1, 64, 160 * 1024>>>(d, c.n, o, c.off, c.perm, c.soff);
        std::vector<float> r(1024);
        hipMemcpy(r.data(), o, 1024 * 4, hipMemcpyDeviceToHost);
        printf("lds_off %d perm %d soffset %d records %d floats (%s)\n", c.off, c.perm, c.soff, c.n, hipGetErrorString(hipGetLastError()));
        printf("  buffer->lds, first 24 floats at the destination: "); for (int i = 0; i < 24; ++i) printf("%g ", r[i]); printf("\n");
        printf("  buffer->lds, floats 120..135 / 248..255:          "); for (int i = 120; i < 136; ++i) printf("%g ", r[i]); printf("| "); for (int i = 248; i < 256; ++i) printf("%g ", r[i]); printf("\n");
        printf("  global->lds (+1 KiB), first 12:                   "); for (int i = 256; i < 268; ++i) printf("%g ", r[i]); printf("\n");
        if (c.off > 65535) { printf("  at (destination & 0xffff), first 8:               "); for (int i = 512; i < 520; ++i) printf("%g ", r[i]); printf(" | +1 KiB: "); for (int i = 768; i < 776; ++i) printf("%g ", r[i]); printf("\n"); }
    }
    return 0;
}
