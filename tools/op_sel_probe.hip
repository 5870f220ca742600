// Can the packed-fp32 op_sel hazard of DESIGN.md 4.14 be shown outside the conv kernel?
// There, `v_pk_add_f32 d, a, v[p-1:p] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]` (the HIGH register of the pair routed into the LOW lane) read 0.0 in lanes
// 48..63 on some executions when it was the first reader of a ds_bpermute result, with a second workgroup on the CU keeping the matrix pipe and the
// LDS busy.  This probe isolates that instruction sequence:
//   per iteration and wave:  v101 <- poison;  ds_bpermute_b32 v101 <- lane (l & 48)'s value;  [wait];  d = a - {v101, v101}  three ways:
//     mode 0: v_pk_add_f32 ... v[100:101] op_sel:[0,1]          (high register into the low lane: the suspect)
//     mode 1: v_pk_add_f32 ... v[100:101] op_sel_hi:[1,0] with the pivot returned into v100 (low register into the high lane: what the fix compiles to)
//     mode 2: v_mov_b32 v100, v101 first, then a plain pair (the full-tile path of the old code)
//   and checks both halves of d in every lane.  Every wave alternates between checking and an MFMA + ds_read loop; the two waves that share a SIMD
//   (one from each resident workgroup) start in opposite phases, so the packed add issues beside the other wave's MFMAs.  `gap` s_nop's between the s_waitcnt and the packed add (0 = back to back).
// Prints mismatches per mode and, for the first few, lane, half, expected and got.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/op_sel_probe tools/op_sel_probe.hip && tools/bin/op_sel_probe [iterations, default 20000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// the register pair under test: -DPLO=132 -DPHI=133 (default v[100:101]; the conv kernel's pairs were v[130:131] .. v[144:145])
#ifndef PLO
#define PLO 100
#define PHI 101
#endif
#define STR_(x) #x
#define STR(x) STR_(x)
#define RLO "v" STR(PLO)
#define RHI "v" STR(PHI)
#define RPAIR "v[" STR(PLO) ":" STR(PHI) "]"
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Rec { int lane, half, iter, blk; float want, got; };

template <int MODE, int GAP, int NEIGH>
__global__ __launch_bounds__(256, 2) void probe(int iters, unsigned* nbad, Rec* recs, float* sink) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)i * 0.25f;
    __syncthreads();
    // Wave k of a workgroup runs on SIMD k; the two workgroups resident on a CU are (first fill) 256 apart in blockIdx.  Every wave alternates between
    // an MFMA + ds_read_b128 phase (the neighbour's main loop) and a checking phase, and co-resident waves of one SIMD start in opposite phases, so a
    // packed add of one wave issues beside the other wave's MFMAs -- the situation of the conv epilogue.
    const int start = ((blockIdx.x >> 8) ^ wave) & 1;
    const int addr = (lane & 48) * 4;                     // ds_bpermute address: lane (l & 48)
    constexpr int PHASES = 16;
    const int per = iters / PHASES;
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned bad = 0;
    for (int phase = 0; phase < PHASES; ++phase) {
        if ((phase + start) & 1) {
            for (int it = 0; it < per; ++it) {
                if (NEIGH == 4) {                 // dense MFMAs whose A / B operands live in v[132:135] / v[136:139] -- the registers the conv kernel keeps its fragments in, and the pair under test
                    const bf16x8 a = *(const bf16x8*)(lds + ((it * 64 + lane) * 4 & 8188)), b = *(const bf16x8*)(lds + (((it + 7) * 64 + lane) * 4 & 8188));
                    __builtin_amdgcn_s_setprio(1);
                    asm volatile("v_mov_b32 v132, %8\n\tv_mov_b32 v133, %9\n\tv_mov_b32 v134, %10\n\tv_mov_b32 v135, %11\n\t"
                                 "v_mov_b32 v136, %12\n\tv_mov_b32 v137, %13\n\tv_mov_b32 v138, %14\n\tv_mov_b32 v139, %15\n\t"
                                 ".rept 4\n\t"
                                 "v_mfma_f32_16x16x32_bf16 %0, v[132:135], v[136:139], %0\n\tv_mfma_f32_16x16x32_bf16 %1, v[132:135], v[136:139], %1\n\t"
                                 "v_mfma_f32_16x16x32_bf16 %2, v[132:135], v[136:139], %2\n\tv_mfma_f32_16x16x32_bf16 %3, v[132:135], v[136:139], %3\n\t"
                                 "v_mfma_f32_16x16x32_bf16 %4, v[132:135], v[136:139], %4\n\tv_mfma_f32_16x16x32_bf16 %5, v[132:135], v[136:139], %5\n\t"
                                 "v_mfma_f32_16x16x32_bf16 %6, v[132:135], v[136:139], %6\n\tv_mfma_f32_16x16x32_bf16 %7, v[132:135], v[136:139], %7\n\t"
                                 ".endr\n\t"
                                 : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                                 : "v"(((const int*)&a)[0]), "v"(((const int*)&a)[1]), "v"(((const int*)&a)[2]), "v"(((const int*)&a)[3]),
                                   "v"(((const int*)&b)[0]), "v"(((const int*)&b)[1]), "v"(((const int*)&b)[2]), "v"(((const int*)&b)[3])
                                 : "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139");
                    __builtin_amdgcn_s_setprio(0);
                } else if (NEIGH == 3) {                 // a dense main loop: 32 MFMAs back to back at raised priority, operands fetched once per 32 (the conv kernel's K-step)
                    const bf16x8 a = *(const bf16x8*)(lds + ((it * 64 + lane) * 4 & 8188)), b = *(const bf16x8*)(lds + (((it + 7) * 64 + lane) * 4 & 8188));
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
                    __builtin_amdgcn_s_setprio(0);
                } else if (NEIGH != 1) {          // the neighbour's main loop: MFMAs fed by ds_read_b128
                    const bf16x8 a = *(const bf16x8*)(lds + ((it * 64 + lane) * 4 & 8188)), b = *(const bf16x8*)(lds + (((it + 7) * 64 + lane) * 4 & 8188));
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
                }
                if (NEIGH == 1 || NEIGH == 2) {    // the neighbour's epilogue: DPP row sums (quad_perm, row_half_mirror, row_mirror, bound_ctrl) as in vt_row16_sum
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        float x = acc[i][1] + (float)it;
                        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
                        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));
                        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));
                        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));
                        acc[i][1] = x * 1e-3f;
                    }
                }
            }
            continue;
        }
        for (int it0 = 0; it0 < per; ++it0) {
            const int it = phase * per + it0;
            const float mine = (float)(lane + 1) * 0.5f + (float)(it & 1023);        // the value each lane offers; lanes 0, 16, 32, 48 are the pivots
            const f32x2 a = {mine * 3.0f + 1.0f, mine * 5.0f - 2.0f};
            f32x2 d;
            float piv = (float)((lane & 48) + 1) * 0.5f + (float)(it & 1023);
            if (MODE == 0) {          // pivot through the LDS into the HIGH register, op_sel:[0,1] first reader
                asm volatile("v_mov_b32 " RLO ", 0x7fc00000\n\tv_mov_b32 " RHI ", 0x7fc00000\n\t"          // poison (NaN) in both halves of the pair
                             "ds_bpermute_b32 " RHI ", %2, %3\n\t"
                             "s_waitcnt lgkmcnt(0)\n\t"
                             ".rept %4\n\ts_nop 0\n\t.endr\n\t"
                             "v_pk_add_f32 %0, %1, " RPAIR " op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                             : "=&v"(d) : "v"(a), "v"(addr), "v"(mine), "n"(GAP) : RLO, RHI, "memory");
            } else if (MODE == 1) {   // control: pivot into the LOW register, op_sel_hi:[1,0]
                asm volatile("v_mov_b32 " RLO ", 0x7fc00000\n\tv_mov_b32 " RHI ", 0x7fc00000\n\t"
                             "ds_bpermute_b32 " RLO ", %2, %3\n\t"
                             "s_waitcnt lgkmcnt(0)\n\t"
                             ".rept %4\n\ts_nop 0\n\t.endr\n\t"
                             "v_pk_add_f32 %0, %1, " RPAIR " op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                             : "=&v"(d) : "v"(a), "v"(addr), "v"(mine), "n"(GAP) : RLO, RHI, "memory");
            } else if (MODE == 2) {   // control: v_mov_b32 first, then a plain pair
                asm volatile("v_mov_b32 " RLO ", 0x7fc00000\n\tv_mov_b32 " RHI ", 0x7fc00000\n\t"
                             "ds_bpermute_b32 " RHI ", %2, %3\n\t"
                             "s_waitcnt lgkmcnt(0)\n\t"
                             ".rept %4\n\ts_nop 0\n\t.endr\n\t"
                             "v_mov_b32 " RLO ", " RHI "\n\t"
                             "v_pk_add_f32 %0, %1, " RPAIR " neg_lo:[0,1] neg_hi:[0,1]\n\t"
                             : "=&v"(d) : "v"(a), "v"(addr), "v"(mine), "n"(GAP) : RLO, RHI, "memory");
            } else if (MODE == 3) {   // no LDS at all: the pivot (this lane's own value) moved into the HIGH register, 0.0 in the low one, op_sel:[0,1]
                piv = mine;
                asm volatile("v_mov_b32 " RLO ", 0\n\tv_mov_b32 " RHI ", %2\n\t"
                             ".rept %3\n\ts_nop 0\n\t.endr\n\t"
                             "v_pk_add_f32 %0, %1, " RPAIR " op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                             : "=&v"(d) : "v"(a), "v"(mine), "n"(GAP) : RLO, RHI, "memory");
            } else {                  // as mode 3 without the neg modifiers (d = a + pivot)
                piv = -mine;
                asm volatile("v_mov_b32 " RLO ", 0\n\tv_mov_b32 " RHI ", %2\n\t"
                             ".rept %3\n\ts_nop 0\n\t.endr\n\t"
                             "v_pk_add_f32 %0, %1, " RPAIR " op_sel:[0,1]\n\t"
                             : "=&v"(d) : "v"(a), "v"(mine), "n"(GAP) : RLO, RHI, "memory");
            }
            for (int h = 0; h < 2; ++h) {
                const float want = a[h] - piv;
                if (!(d[h] == want)) {
                    ++bad;
                    const unsigned k = atomicAdd(nbad + 1, 1u);
                    if (k < 16) recs[k] = Rec{lane, h, it, (int)blockIdx.x, want, d[h]};
                }
            }
        }
    }
    float sacc = 0.f;
    for (int i = 0; i < 8; ++i) sacc += acc[i][0] + acc[i][3];
    if (sink && sacc == 12345.678f) sink[blockIdx.x] = sacc;
    if (bad) atomicAdd(nbad, bad);
}

template <int MODE, int GAP, int NEIGH = 0>
void run(const char* what, int iters) {
    unsigned* nbad; Rec* recs; float* sink;
    CHECK(hipMalloc(&nbad, 8)); CHECK(hipMalloc(&recs, 16 * sizeof(Rec))); CHECK(hipMalloc(&sink, 4096 * 4));
    CHECK(hipMemset(nbad, 0, 8)); CHECK(hipMemset(recs, 0, 16 * sizeof(Rec)));
    const int nblk = 2048;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<MODE, GAP, NEIGH>), dim3(nblk), dim3(256), 0, 0, iters, nbad, recs, sink);
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned h[2]; Rec r[16];
    CHECK(hipMemcpy(h, nbad, 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(r, recs, sizeof(r), hipMemcpyDeviceToHost));
    const double checks = (double)nblk * 4 * 64 * 2 * (iters / 2);
    printf("pair " RPAIR " neighbour %s %-64s gap %d: %u wrong of %.3g half-results (%.1f ms)\n", NEIGH == 0 ? "MFMA" : NEIGH == 1 ? "DPP " : NEIGH == 2 ? "MFMA+DPP" : NEIGH == 3 ? "dense MFMA" : "dense MFMA on v[132:139]", what, GAP, h[0], checks, ms);
    for (unsigned i = 0; i < h[1] && i < 6; ++i)
        printf("      block %d iteration %d lane %d half %d: want %g got %g\n", r[i].blk, r[i].iter, r[i].lane, r[i].half, r[i].want, r[i].got);
    CHECK(hipFree(nbad)); CHECK(hipFree(recs)); CHECK(hipFree(sink));
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 0>("op_sel:[0,1] (high register -> low lane), ds_bpermute pivot", iters);
        run<0, 4>("op_sel:[0,1] (high register -> low lane), ds_bpermute pivot", iters);
        run<3, 0>("op_sel:[0,1], pivot by v_mov_b32 (no LDS), 0.0 in the low register", iters);
        run<4, 0>("op_sel:[0,1] without neg modifiers, pivot by v_mov_b32", iters);
        run<1, 0>("op_sel_hi:[1,0] (low register -> high lane), ds_bpermute pivot", iters);
        run<2, 0>("v_mov_b32 first, then a plain pair", iters);
        run<0, 0, 1>("op_sel:[0,1] (high register -> low lane), ds_bpermute pivot", iters);
        run<3, 0, 1>("op_sel:[0,1], pivot by v_mov_b32 (no LDS), 0.0 in the low register", iters);
        run<3, 0, 2>("op_sel:[0,1], pivot by v_mov_b32 (no LDS), 0.0 in the low register", iters);
        run<1, 0, 2>("op_sel_hi:[1,0] (low register -> high lane), ds_bpermute pivot", iters);
        run<3, 0, 3>("op_sel:[0,1], pivot by v_mov_b32 (no LDS), 0.0 in the low register", iters / 4);
        run<0, 0, 3>("op_sel:[0,1] (high register -> low lane), ds_bpermute pivot", iters / 4);
        run<1, 0, 3>("op_sel_hi:[1,0] (low register -> high lane), ds_bpermute pivot", iters / 4);
        run<3, 0, 4>("op_sel:[0,1], pivot by v_mov_b32 (no LDS), 0.0 in the low register", iters / 4);
        run<0, 0, 4>("op_sel:[0,1] (high register -> low lane), ds_bpermute pivot", iters / 4);
    }
    return 0;
}
