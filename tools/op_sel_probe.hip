// Can the packed-fp32 op_sel hazard of DESIGN.md 4.14 be shown outside the conv kernel?
// There, `v_pk_add_f32 d, a, v[p-1:p] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]` (the HIGH register of the pair routed into the LOW lane) read 0.0 in lanes
// 48..63 on some executions when it was the first reader of a ds_bpermute result, with a second workgroup on the CU keeping the matrix pipe and the
// LDS busy.  This probe isolates that instruction sequence:
//   per iteration and wave:  v101 <- poison;  ds_bpermute_b32 v101 <- lane (l & 48)'s value;  [wait];  d = a - {v101, v101}  three ways:
//     mode 0: v_pk_add_f32 ... v[100:101] op_sel:[0,1]          (high register into the low lane: the suspect)
//     mode 1: v_pk_add_f32 ... v[100:101] op_sel_hi:[1,0] with the pivot returned into v100 (low register into the high lane: what the fix compiles to)
//     mode 2: v_mov_b32 v100, v101 first, then a plain pair (the full-tile path of the old code)
//   and checks both halves of d in every lane.  Half of the waves of every workgroup run the check; the other half run an MFMA + ds_read loop as
//   the neighbour.  `gap` s_nop's between the s_waitcnt and the packed add (0 = back to back).
// Prints mismatches per mode and, for the first few, lane, half, expected and got.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/op_sel_probe tools/op_sel_probe.hip && tools/bin/op_sel_probe [iterations, default 20000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Rec { int lane, half, iter, blk; float want, got; };

template <int MODE, int GAP>
__global__ __launch_bounds__(256, 2) void probe(int iters, unsigned* nbad, Rec* recs, float* sink) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)i * 0.25f;
    __syncthreads();
    if (wave & 1) {
        // the neighbour: MFMAs fed by ds_read_b128, as the other resident workgroup's main loop does
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
            const bf16x8 a = *(const bf16x8*)(lds + ((it * 64 + lane) * 4 & 8188)), b = *(const bf16x8*)(lds + (((it + 7) * 64 + lane) * 4 & 8188));
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
        if (sink && s == 12345.678f) sink[blockIdx.x] = s;
        return;
    }
    const int addr = (lane & 48) * 4;                     // ds_bpermute address: lane (l & 48)
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        const float mine = (float)(lane + 1) * 0.5f + (float)(it & 1023);        // the value each lane offers; lanes 0, 16, 32, 48 are the pivots
        const f32x2 a = {mine * 3.0f + 1.0f, mine * 5.0f - 2.0f};
        f32x2 d;
        if (MODE == 0) {
            asm volatile("v_mov_b32 v100, 0x7fc00000\n\tv_mov_b32 v101, 0x7fc00000\n\t"          // poison (NaN): a stale read shows as NaN, a zero read as a - 0
                         "ds_bpermute_b32 v101, %2, %3\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         ".rept %4\n\ts_nop 0\n\t.endr\n\t"
                         "v_pk_add_f32 %0, %1, v[100:101] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                         : "=&v"(d) : "v"(a), "v"(addr), "v"(mine), "n"(GAP) : "v100", "v101", "memory");
        } else if (MODE == 1) {
            asm volatile("v_mov_b32 v100, 0x7fc00000\n\tv_mov_b32 v101, 0x7fc00000\n\t"
                         "ds_bpermute_b32 v100, %2, %3\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         ".rept %4\n\ts_nop 0\n\t.endr\n\t"
                         "v_pk_add_f32 %0, %1, v[100:101] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                         : "=&v"(d) : "v"(a), "v"(addr), "v"(mine), "n"(GAP) : "v100", "v101", "memory");
        } else {
            asm volatile("v_mov_b32 v100, 0x7fc00000\n\tv_mov_b32 v101, 0x7fc00000\n\t"
                         "ds_bpermute_b32 v101, %2, %3\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         ".rept %4\n\ts_nop 0\n\t.endr\n\t"
                         "v_mov_b32 v100, v101\n\t"
                         "v_pk_add_f32 %0, %1, v[100:101] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                         : "=&v"(d) : "v"(a), "v"(addr), "v"(mine), "n"(GAP) : "v100", "v101", "memory");
        }
        const float piv = (float)((lane & 48) + 1) * 0.5f + (float)(it & 1023);
        for (int h = 0; h < 2; ++h) {
            const float want = a[h] - piv;
            if (!(d[h] == want)) {
                ++bad;
                const unsigned k = atomicAdd(nbad + 1, 1u);
                if (k < 16) recs[k] = Rec{lane, h, it, (int)blockIdx.x, want, d[h]};
            }
        }
    }
    if (bad) atomicAdd(nbad, bad);
}

template <int MODE, int GAP>
void run(const char* what, int iters) {
    unsigned* nbad; Rec* recs; float* sink;
    CHECK(hipMalloc(&nbad, 8)); CHECK(hipMalloc(&recs, 16 * sizeof(Rec))); CHECK(hipMalloc(&sink, 4096 * 4));
    CHECK(hipMemset(nbad, 0, 8)); CHECK(hipMemset(recs, 0, 16 * sizeof(Rec)));
    const int nblk = 2048;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<MODE, GAP>), dim3(nblk), dim3(256), 0, 0, iters, nbad, recs, sink);
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned h[2]; Rec r[16];
    CHECK(hipMemcpy(h, nbad, 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(r, recs, sizeof(r), hipMemcpyDeviceToHost));
    const double checks = (double)nblk * 2 * 64 * 2 * iters;
    printf("%-64s gap %d: %u wrong of %.3g half-results (%.1f ms)\n", what, GAP, h[0], checks, ms);
    for (unsigned i = 0; i < h[1] && i < 6; ++i)
        printf("      block %d iteration %d lane %d half %d: want %g got %g\n", r[i].blk, r[i].iter, r[i].lane, r[i].half, r[i].want, r[i].got);
    CHECK(hipFree(nbad)); CHECK(hipFree(recs)); CHECK(hipFree(sink));
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 0>("op_sel:[0,1] (high register -> low lane), first reader", iters);
        run<0, 4>("op_sel:[0,1] (high register -> low lane), first reader", iters);
        run<1, 0>("op_sel_hi:[1,0] (low register -> high lane), first reader", iters);
        run<2, 0>("v_mov_b32 first, then a plain pair", iters);
    }
    return 0;
}
