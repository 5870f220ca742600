// What does FETCH_SIZE count -- and what does the memory system move -- when an LDS-DMA gathers HALF lines?  
// The stride-2 phase-plane kernels gather 64-B pieces (one pixel's 32-channel chunk) whose neighbours in a plane sit 512 B apart; the
// other half of each 128-B line belongs to the next channel chunk, which the kernel stages nine K-steps later.  The committed PMC traffic
// of those kernels doubles FETCH_SIZE as MI355X_MICROARCH.md prescribes for wide coalesced reads -- is that right for this access shape?
//   mode 0: every wave instruction reads 1 KB contiguous (64 lanes x 16 B): whole 128-B lines, each once;
//   mode 1: 4 lanes read one 64-B piece, pieces 256 B apart: the FIRST half of every second line, each line touched once;
//   mode 2: as mode 1, and the same workgroup reads the OTHER half of the same lines `lag` iterations (lag x 16 KB of lines) later
//           (what chunk c + 1 of the stride-2 kernel does to the lines chunk c touched);
//   mode 3: 8 lanes read both halves of a line as two adjacent 64-B pieces of one instruction (the "chunk pair" staging).
// Buffer >> the 256-MB Infinity Cache, every byte read once per sweep.  Prints algorithmic bytes, time and GB/s; run under
//   rocprofv3 --pmc FETCH_SIZE   (and a second time with  --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum  where the part lists them)
// to get the counter per kernel: the ratio counter / algorithmic bytes is the correction factor for that access shape, and the time tells
// whether the un-read half of a line crosses the fabric (mode 1 as slow per piece as mode 0 per line) or not.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/fetch_probe tools/fetch_probe.hip && tools/bin/fetch_probe [GiB, default 2]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int NT = 256, INFLIGHT = 8;

// each workgroup owns a contiguous slab; a wave instruction covers `span` bytes of it (1 KB of lines in mode 0 / 3: 1 KB read; 4 KB of lines in
// modes 1 / 2: 1 KB read)
template <int MODE>
__global__ __launch_bounds__(NT) void probe(const unsigned char* __restrict__ buf, size_t slab, int iters, float* sink, int lag) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[NT / 64 * INFLIGHT * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char* base = buf + (size_t)blockIdx.x * slab;
    unsigned char* dst = lds + wave * INFLIGHT * 1024;
    size_t off;
    size_t span;
    if (MODE == 0 || MODE == 3) { off = (size_t)lane * 16; span = 1024; }             // contiguous (mode 3: lanes 0-3 first half, 4-7 second half of a line: the same addresses)
    else { off = (size_t)(lane >> 2) * 256 + (lane & 3) * 16; span = 4096; }          // 64-B pieces, 256 B apart
    const int nw = NT / 64;
    // mode 2: the slab in blocks of `lag` iterations -- first halves of a block's lines, then their second halves (lag x 16 KB of lines per workgroup in between)
    const int blk = MODE == 2 ? lag : iters;
    for (int b0 = 0; b0 < iters; b0 += blk) {
        for (int sweep = 0; sweep < (MODE == 2 ? 2 : 1); ++sweep) {
            const size_t half = (MODE == 2 && sweep == 1) ? 64 : 0;                   // the second sweep takes the other half of every line
            for (int it = b0; it < b0 + blk && it < iters; it += INFLIGHT) {
#pragma unroll
                for (int j = 0; j < INFLIGHT; ++j) {
                    const size_t o = ((size_t)(it + j) * nw + wave) * span + off + half;
                    __builtin_amdgcn_global_load_lds(GPTR(base + o), LPTR(dst + j * 1024), 16, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
    }
    __syncthreads();
    if (sink && threadIdx.x == 0) sink[blockIdx.x] = (float)lds[0];
}

template <int MODE>
void run(const char* what, const unsigned char* buf, size_t bytes, float* sink, int lag = 8) {
    const int nblk = 2048;
    const size_t slab = bytes / nblk;
    const size_t span = (MODE == 0 || MODE == 3) ? 1024 : 4096;
    int iters = (int)(slab / (span * (NT / 64)));
    iters -= iters % INFLIGHT;
    const double algo = (double)nblk * iters * (NT / 64) * 1024.0 * (MODE == 2 ? 2 : 1);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(probe<MODE>, dim3(nblk), dim3(NT), 0, 0, buf, slab, iters, sink, lag);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("mode %d  %-66s algorithmic %8.1f MB  lines touched %8.1f MB  %7.3f ms  %7.1f GB/s algorithmic  %7.1f GB/s of touched lines\n", MODE, what,
           algo / 1e6, (double)nblk * iters * (NT / 64) * span / 1e6, best, algo / best / 1e6, (double)nblk * iters * (NT / 64) * span / best / 1e6);
}

int main(int argc, char** argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 2.0;
    const size_t bytes = (size_t)(gib * (1ull << 30)) / (2048 * 4096 * 4 * 8) * (2048 * 4096 * 4 * 8);
    unsigned char* buf; float* sink;
    CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&sink, 2048 * 4));
    CHECK(hipMemset(buf, 1, bytes));
    CHECK(hipDeviceSynchronize());
    printf("buffer %.2f GiB (Infinity Cache: 256 MB)\n", bytes / double(1ull << 30));
    run<0>("whole lines, 1 KB contiguous per wave instruction", buf, bytes, sink);
    run<1>("64-B pieces 256 B apart: first half of every second line", buf, bytes, sink);
    run<2>("the same + the other halves 8 iterations (128 KB of lines per workgroup) later", buf, bytes, sink, 8);
    run<2>("the same + the other halves 64 iterations (1 MB per workgroup) later", buf, bytes, sink, 64);
    run<3>("both halves of a line as adjacent pieces of one instruction", buf, bytes, sink);
    CHECK(hipFree(buf)); CHECK(hipFree(sink));
    return 0;
}
