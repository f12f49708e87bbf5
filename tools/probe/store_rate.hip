// Probe (round 4): what does a 256 x 256 bf16 tile's C store cost a CU, and does the ADDRESS FORM of the store matter?
// The GEMM's full-tile epilogue issues 16 x global_store_dwordx4 per wave (8 waves, 128 KB per tile); in-kernel stamps (round 3) put the
// epilogue at 3.4-8.2 us per tile and "skipping its stores" at -5.5 us: ~14 B/clk per CU, far below the 64 B/clk read path. A store moves its
// address and data registers to the memory pipeline: 8 + 16 bytes per lane with a 64-bit VGPR address, 4 + 16 with an SGPR base + 32-bit VGPR
// offset (saddr form) or a buffer descriptor. This probe issues the same bursts in each form and stamps issue time and completion time per wave.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/store_rate.hip -o tools/probe/bin/store_rate && tools/probe/bin/store_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// MODE 0: 64-bit vaddr, plain | 1: 64-bit vaddr, nt | 2: saddr + voffset, plain | 3: saddr + voffset, nt | 4: buffer_store (srsrc + voffset), plain | 5: buffer nt
// PAT 0: the GEMM epilogue's pattern: one instruction = 16 rows x 64 B (lane (r = lane & 15, g = lane >> 4) writes 16 B at row r, byte 16 g), row stride `ldc_bytes`
// PAT 1: one instruction = 1 KiB contiguous
template <int MODE, int PAT>
__global__ __launch_bounds__(512, 2) void k(unsigned char* out, long ldc_bytes, unsigned long long* stamps, int bursts, int gap_iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char* tile = out + (size_t)blockIdx.x * 256 * ldc_bytes;              // a 256-row x 512-byte tile per workgroup
    // wave (wr = wave >> 2, wc = wave & 3) owns rows {i * 128 + wr * 64 ..}, byte columns wc * 64 (+ j * 256)
    const int wr = wave >> 2, wc = wave & 3;
    u32x4 v = {(unsigned)lane, (unsigned)wave, 0x3f803f80u, 0x40004000u};
    unsigned long long t_issue = 0, t_done = 0;
    __builtin_amdgcn_s_barrier();
    for (int b = 0; b < bursts; ++b) {
        unsigned long long t0, t1, t2;
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            long off;
            if (PAT == 0) {
                const int i = s >> 3, mt = (s >> 1) & 3, j = s & 1;
                const int row = i * 128 + wr * 64 + mt * 16 + (lane & 15);
                off = (long)row * ldc_bytes + j * 256 + wc * 64 + (lane >> 4) * 16;
            } else if (PAT == 1) {
                off = (long)(wave * 16 + s) * 1024 + lane * 16 + (long)(b & 1) * 0;
            } else if (PAT >= 2 && PAT <= 5) {
                // R rows per instruction, 1024 / R contiguous bytes per row, rows `ldc_bytes` apart: PAT 2: R = 8 (128 B = one line per row), 3: R = 4, 4: R = 2, 5: R = 1 (with the row stride)
                constexpr int R = PAT == 2 ? 8 : PAT == 3 ? 4 : PAT == 4 ? 2 : 1;
                constexpr int LPR = 64 / R;                          // lanes per row
                const int row = (wave * 16 + s) * R + lane / LPR;      // 8 waves x 16 instructions x R rows <= 1024 rows: the tile region is 256 rows, wrap into it
                off = (long)(row & 255) * ldc_bytes + (long)(row >> 8) * 1024 + (lane % LPR) * 16;
            } else if (PAT == 7) {
                // the attention forward's epilogue (T21 form): one instruction = 32 rows x 32 B (lane l: row l & 31, bytes 16 (l >> 5) .. + 15 of a 32-byte group), 8 groups per 256-byte row
                const int row = wave * 32 + (lane & 31);
                off = (long)row * ldc_bytes + (s & 7) * 32 + (lane >> 5) * 16 + (long)(s >> 3) * 256;
            } else {
                // PAT 6: the GEMM pattern's shape (16 rows x 64 B per instruction) with the rows only 64 B apart (one 1 KiB run): lines, not pages
                off = (long)(wave * 16 + s) * 1024 + (lane & 15) * 64 + (lane >> 4) * 16;
            }
            v.z += s;
            if (MODE == 0) { unsigned char* p = tile + off; asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
            else if (MODE == 1) { unsigned char* p = tile + off; asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory"); }
            else if (MODE == 2) { unsigned o32 = (unsigned)off; asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(o32), "v"(v), "s"(tile) : "memory"); }
            else if (MODE == 3) { unsigned o32 = (unsigned)off; asm volatile("global_store_dwordx4 %0, %1, %2 nt" ::"v"(o32), "v"(v), "s"(tile) : "memory"); }
            else {
                const auto rs = __builtin_amdgcn_make_buffer_rsrc(tile, 0, 0x7fffffff, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, (int)off, 0, MODE == 5 ? 2 : 0);
            }
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
        t_issue += t1 - t0; t_done += t2 - t0;
        for (int g = 0; g < gap_iters; ++g) asm volatile("s_sleep 8");
        __builtin_amdgcn_s_barrier();
    }
    if (lane == 0) { stamps[((size_t)blockIdx.x * 8 + wave) * 2] = t_issue; stamps[((size_t)blockIdx.x * 8 + wave) * 2 + 1] = t_done; }
}

template <int MODE, int PAT>
static void run(const char* name, unsigned char* out, long ldc, unsigned long long* st, int blocks) {
    const int bursts = 64;
    std::vector<unsigned long long> h(blocks * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, PAT>), dim3(blocks), dim3(512), 0, 0, out, ldc, st, 4, 50);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, PAT>), dim3(blocks), dim3(512), 0, 0, out, ldc, st, bursts, 50);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> iss, don;
    for (int w = 0; w < blocks * 8; ++w) { iss.push_back(h[2 * w] / (double)bursts); don.push_back(h[2 * w + 1] / (double)bursts); }
    std::sort(iss.begin(), iss.end()); std::sort(don.begin(), don.end());
    // per burst a CU's 8 waves store 128 KiB: B/clk/CU = 131072 / (slowest wave's completion)
    printf("%-34s issue median %7.0f max %7.0f | done median %7.0f max %7.0f cycles per 16-store burst -> %5.1f B/clk/CU (128 KiB / median done)  [%0.2f ms]\n", name,
           iss[iss.size() / 2], iss.back(), don[don.size() / 2], don.back(), 131072.0 / don[don.size() / 2], ms);
}

// The same bursts as LOADS (the residual epilogues read R in the epilogue's pattern): PAT 0 = 16 rows x 64 B per instruction, PAT 2 = 8 rows x 128 B
template <int PAT>
__global__ __launch_bounds__(512, 2) void kl(const unsigned char* in, long ldc_bytes, unsigned long long* stamps, int bursts, int gap_iters, unsigned* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char* tile = in + (size_t)blockIdx.x * 256 * ldc_bytes;
    const int wr = wave >> 2, wc = wave & 3;
    unsigned long long t_issue = 0, t_done = 0;
    unsigned acc = 0;
    __builtin_amdgcn_s_barrier();
    for (int b = 0; b < bursts; ++b) {
        unsigned long long t0, t1, t2;
        u32x4 v[16];
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            long off;
            if (PAT == 0) {
                const int i = s >> 3, mt = (s >> 1) & 3, j = s & 1;
                const int row = i * 128 + wr * 64 + mt * 16 + (lane & 15);
                off = (long)row * ldc_bytes + j * 256 + wc * 64 + (lane >> 4) * 16;
            } else {
                const int row = (wave * 16 + s) * 8 + lane / 8;
                off = (long)(row & 255) * ldc_bytes + (long)(row >> 8) * 1024 + (lane % 8) * 16;
            }
            const unsigned char* p = tile + off + (size_t)(b & 7) * 0;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[s]) : "v"(p) : "memory");
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2), "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                     "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])::"memory");
#pragma unroll
        for (int s = 0; s < 16; ++s) acc ^= v[s].x;
        t_issue += t1 - t0; t_done += t2 - t0;
        for (int g = 0; g < gap_iters; ++g) asm volatile("s_sleep 8");
        __builtin_amdgcn_s_barrier();
    }
    if (acc == 0x12345678u) sink[0] = acc;
    if (lane == 0) { stamps[((size_t)blockIdx.x * 8 + wave) * 2] = t_issue; stamps[((size_t)blockIdx.x * 8 + wave) * 2 + 1] = t_done; }
}

template <int PAT>
static void runl(const char* name, unsigned char* out, long ldc, unsigned long long* st, int blocks) {
    const int bursts = 64;
    std::vector<unsigned long long> h(blocks * 16);
    unsigned* sink; hipMalloc(&sink, 64);
    hipLaunchKernelGGL((kl<PAT>), dim3(blocks), dim3(512), 0, 0, out, ldc, st, 4, 50, sink);
    hipLaunchKernelGGL((kl<PAT>), dim3(blocks), dim3(512), 0, 0, out, ldc, st, bursts, 50, sink);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> iss, don;
    for (int w = 0; w < blocks * 8; ++w) { iss.push_back(h[2 * w] / (double)bursts); don.push_back(h[2 * w + 1] / (double)bursts); }
    std::sort(iss.begin(), iss.end()); std::sort(don.begin(), don.end());
    printf("%-34s issue median %7.0f max %7.0f | done median %7.0f max %7.0f cycles per 16-load burst -> %5.1f B/clk/CU\n", name,
           iss[iss.size() / 2], iss.back(), don[don.size() / 2], don.back(), 131072.0 / don[don.size() / 2]);
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 256;
    const long ldc = 6144;                                   // bytes: a [M][3072] bf16 C
    unsigned char* out; unsigned long long* st;
    hipMalloc(&out, (size_t)blocks * 256 * 49152 + (1 << 20)); hipMalloc(&st, blocks * 16 * 8);
    printf("blocks %d (one 8-wave workgroup each), 16 x dwordx4 per wave per burst, bursts separated by sleep + barrier\n", blocks);
    run<0, 0>("gemm rows, vaddr64, plain", out, ldc, st, blocks);
    run<1, 0>("gemm rows, vaddr64, nt", out, ldc, st, blocks);
    run<2, 0>("gemm rows, saddr+voff32, plain", out, ldc, st, blocks);
    run<3, 0>("gemm rows, saddr+voff32, nt", out, ldc, st, blocks);
    run<4, 0>("gemm rows, buffer, plain", out, ldc, st, blocks);
    run<5, 0>("gemm rows, buffer, nt", out, ldc, st, blocks);
    run<1, 2>("8 rows x 128 B, nt", out, ldc, st, blocks);
    run<1, 3>("4 rows x 256 B, nt", out, ldc, st, blocks);
    run<1, 4>("2 rows x 512 B, nt", out, ldc, st, blocks);
    run<1, 5>("1 row x 1 KiB (row stride), nt", out, ldc, st, blocks);
    run<1, 6>("16 x 64 B pieces of one 1 KiB run, nt", out, ldc, st, blocks);
    run<1, 1>("1 KiB contiguous, vaddr64, nt", out, ldc, st, blocks);
    run<3, 1>("1 KiB contiguous, saddr+voff32, nt", out, ldc, st, blocks);
    run<1, 7>("attention O: 32 rows x 32 B, nt", out, ldc, st, blocks);
    run<0, 7>("attention O: 32 rows x 32 B, plain", out, ldc, st, blocks);
    run<0, 7>("attention O, row stride 48 KiB, plain", out, 49152, st, blocks);
    run<0, 3>("4 rows x 256 B, row stride 48 KiB", out, 49152, st, blocks);
    runl<0>("LOADS gemm rows (16 x 64 B)", out, ldc, st, blocks);
    runl<2>("LOADS 8 rows x 128 B", out, ldc, st, blocks);
    return 0;
}
