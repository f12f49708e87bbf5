// Probe: does global_load_lds_dwordx4 / _dword honour an M0 LDS address above 64 KiB on gfx950 (160 KiB LDS)?
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe/lds_dma_hi.hip -o tools/probe/bin/lds_dma_hi ; prints one line per destination.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ __launch_bounds__(64) void probe(const unsigned* src, unsigned* out, unsigned dst_off, int wide) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 160 * 1024 / 4 - 64; i += 64) ((unsigned*)smem)[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)smem) + dst_off;
    if (wide) { const unsigned* g = src + 4 * lane; asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(base) : "memory"); }
    else      { const unsigned* g = src + lane;     asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(g), "s"(base) : "memory"); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int n = wide ? 256 : 64;
    for (int i = lane; i < n; i += 64) out[i] = ((const unsigned*)(smem + dst_off))[i];
    // where did it land if not there? report the first word index holding src[0]
    if (lane == 0) {
        int found = -1;
        for (int i = 0; i < 160 * 1024 / 4 - 64; ++i) if (((const unsigned*)smem)[i] == src[0]) { found = i * 4; break; }
        out[256] = (unsigned)found;
    }
}
int main() {
    const int lds = 160 * 1024 - 256;
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    std::vector<unsigned> h(256);
    for (int i = 0; i < 256; ++i) h[i] = 0x1000u + i;
    unsigned *src, *out;
    hipMalloc(&src, 1024); hipMalloc(&out, 2048);
    hipMemcpy(src, h.data(), 1024, hipMemcpyHostToDevice);
    int bad = 0;
    for (int wide = 1; wide >= 0; --wide)
        for (unsigned off : {0u, 32768u, 65536u - 1024u, 65536u, 65536u + 4096u, 98304u, 131072u, 150u * 1024u}) {
            hipMemset(out, 0, 2048);
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), lds, 0, src, out, off, wide);
            std::vector<unsigned> o(257);
            hipError_t e = hipMemcpy(o.data(), out, 257 * 4, hipMemcpyDeviceToHost);
            int n = wide ? 256 : 64, ok = 1;
            for (int i = 0; i < n; ++i) ok &= (o[i] == h[i]);
            printf("LDS_DMA_HI %s dst=%u: %s (first copy of src[0] at byte %d) %s\n", wide ? "dwordx4" : "dword", off, ok ? "OK" : "WRONG", (int)o[256], e == hipSuccess ? "" : hipGetErrorString(e));
            bad += !ok;
        }
    printf("LDS_DMA_HI %s\n", bad ? "SOME DESTINATIONS WRONG" : "all destinations honoured");
    return 0;
}
