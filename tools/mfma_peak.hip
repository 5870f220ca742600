// Bare-MFMA micro-benchmark for gfx950: the matrix-pipe rate this chip sustains with operands in registers, on all-zero
// and on random operands (the clock under MFMA load depends on the data), for the four instructions the library uses.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_peak tools/mfma_peak.hip && tools/bin/mfma_peak [seconds-per-case]
// Each wave keeps NACC independent accumulator chains; WPS waves per SIMD; every CU busy.  SURVEY.md 8(d) asks for this
// figure beside the vendor peak the roofline fractions are quoted against.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// KIND 0: v_mfma_f32_16x16x32_bf16, 1: v_mfma_f32_32x32x16_bf16, 2: v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3, scales 2^0),
// 3: v_mfma_scale_f32_16x16x128_f8f6f4
template <int KIND, int NACC>
__global__ __launch_bounds__(256) void mfma_loop(const uint32_t* __restrict__ operands, float* __restrict__ sink, int iters) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t raw[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) raw[i] = operands[(size_t)(tid & 4095) * 16 + i];
    float total = 0.f;
    const int one = 127;                             // E8M0 scale 2^0 for both operands of the scaled forms
    if constexpr (KIND == 0 || KIND == 3) {
        f32x4 acc[NACC];
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j] = f32x4{(float)j, 0.f, 0.f, 0.f};      // distinct chains: nothing to merge
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < NACC; ++j) {
                if constexpr (KIND == 0) {
                    bf16x8 a, b;
                    __builtin_memcpy(&a, &raw[0], 16); __builtin_memcpy(&b, &raw[4 + 4 * (j & 1)], 16);
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
                } else {
                    i32x8 a, b;
                    __builtin_memcpy(&a, &raw[0], 32); __builtin_memcpy(&b, &raw[8], 32);
                    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(acc[j]) : "v"(a), "v"(b), "v"(one));
                }
            }
        }
        asm volatile("s_nop 15\n s_nop 15" ::: "memory");   // the asm MFMAs carry no compiler-inserted hazard nops
#pragma unroll
        for (int j = 0; j < NACC; ++j) total += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    } else {
        f32x16 acc[NACC];
#pragma unroll
        for (int j = 0; j < NACC; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = (float)(j + r);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < NACC; ++j) {
                if constexpr (KIND == 1) {
                    bf16x8 a, b;
                    __builtin_memcpy(&a, &raw[0], 16); __builtin_memcpy(&b, &raw[4 + 4 * (j & 1)], 16);
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
                } else {
                    i32x8 a, b;
                    __builtin_memcpy(&a, &raw[0], 32); __builtin_memcpy(&b, &raw[8], 32);
                    asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]" : "+v"(acc[j]) : "v"(a), "v"(b), "v"(one));
                }
            }
        }
        asm volatile("s_nop 15\n s_nop 15" ::: "memory");
#pragma unroll
        for (int j = 0; j < NACC; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) total += acc[j][r];
    }
    if (total == 1.2345678f) sink[tid] = total;      // keeps the chains alive, never true in practice
}

static uint32_t rng_state = 12345u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state; }
static uint16_t bf16_of(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }

template <int KIND, int NACC>
static void run(const char* name, double flop_per_mfma, int wps, bool random, double seconds, uint32_t* d_ops, float* d_sink) {
    std::vector<uint32_t> h(4096 * 16, 0u);
    if (random) {
        for (size_t i = 0; i < h.size(); ++i) {
            if (KIND <= 1) {             // two bf16 values, roughly normal(0, 1)
                auto g = [] { float s = 0.f; for (int k = 0; k < 6; ++k) s += (float)(rnd() >> 8) / 16777216.f; return (s - 3.f) * 1.41f; };
                h[i] = (uint32_t)bf16_of(g()) | ((uint32_t)bf16_of(g()) << 16);
            } else {                     // four e4m3 bytes: random sign and mantissa, exponent field 4..10 (no NaN 0x7f pattern)
                uint32_t w = 0;
                for (int k = 0; k < 4; ++k) { uint32_t r = rnd() >> 8; uint32_t byte = ((r & 1u) << 7) | ((4u + (r >> 1) % 7u) << 3) | ((r >> 8) & 7u); w |= byte << (8 * k); }
                h[i] = w;
            }
        }
    }
    CHECK(hipMemcpy(d_ops, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * wps;                         // 256 threads = 4 waves = one per SIMD; wps blocks per CU
    int iters = 20000;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((mfma_loop<KIND, NACC>), dim3(blocks), dim3(256), 0, 0, d_ops, d_sink, 2000);
    CHECK(hipDeviceSynchronize());
    double best = 0.0, total_ms = 0.0, last = 0.0;
    // calibrate one launch to ~50 ms, then repeat until `seconds` have been spent under load (the clock settles after ~0.1 s)
    for (int rep = 0; rep < 1000 && total_ms < seconds * 1e3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((mfma_loop<KIND, NACC>), dim3(blocks), dim3(256), 0, 0, d_ops, d_sink, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        total_ms += ms;
        const double tf = flop_per_mfma * NACC * (double)iters * blocks * 4 / (ms * 1e-3) / 1e12;
        if (rep == 0 && ms < 40.f) { iters = (int)(iters * 50.0 / (ms > 0.5f ? ms : 0.5f)); continue; }
        if (tf > best) best = tf;
        last = tf;
    }
    printf("%-38s %d wave(s)/SIMD %-6s  sustained %7.1f TFLOP/s  (best launch %7.1f)\n", name, wps, random ? "random" : "zeros", last, best);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 1.5;
    uint32_t* d_ops; float* d_sink;
    CHECK(hipMalloc(&d_ops, 4096 * 16 * 4)); CHECK(hipMalloc(&d_sink, 1 << 22));
    for (int wps = 1; wps <= 2; ++wps)
        for (int random = 0; random <= 1; ++random) {
            run<0, 8>("v_mfma_f32_16x16x32_bf16", 2.0 * 16 * 16 * 32, wps, random, seconds, d_ops, d_sink);
            run<1, 4>("v_mfma_f32_32x32x16_bf16", 2.0 * 32 * 32 * 16, wps, random, seconds, d_ops, d_sink);
            run<2, 4>("v_mfma_scale_f32_32x32x64_f8f6f4 e4m3", 2.0 * 32 * 32 * 64, wps, random, seconds, d_ops, d_sink);
            run<3, 8>("v_mfma_scale_f32_16x16x128_f8f6f4 e4m3", 2.0 * 16 * 16 * 128, wps, random, seconds, d_ops, d_sink);
        }
    return 0;
}
