// Microbenchmark for the register-stationary fused kernel's ceiling on gfx950 ("a register-pressure-matched MFMA-only
// microbenchmark to state the attainable ceiling").  NOT part of libnmfx; built and run by scripts/run_ubench.sh on the GPU box.
//
// One workgroup of 4 waves per CU, one wave per SIMD, 512 v_mfma_f32_32x32x2_f32 per "tile" with the register footprint of fused_kernel<256, ...>:
//   128 stationary operands in AGPRs, two 16-register S accumulators in VGPRs (two dependent chains of 128 MFMAs: the first product), eight 16-register
//   output accumulators in AGPRs used round-robin (the second product).
// On top of that skeleton, one ingredient of the real tile body at a time:
//   LDS      0 none | 1 the kernel's reads (one ds_read_b128 per 4 MFMAs in P1/P2, two ds_read_b128 per 8 MFMAs in P3/P4) | 2 P3/P4 as 8 ds_read_b32 (round 5)
//   NVALU    full-rate VALU instructions per tile (v_fma_f32), spread evenly behind the MFMAs
//   NTRANS   transcendental instructions per tile (v_rcp_f32)
//   NPK      packed fp32 instructions per tile (v_pk_fma_f32)
//   NACC     v_accvgpr_read_b32 per tile
//   BAR      s_barrier per tile
//   DEP      1: the first product's MFMAs as TWO dependent chains (the kernel) | 0: spread over the eight independent accumulators as well
//   GROUP    the fillers are issued GROUP at a time (behind every GROUP-th of the MFMAs they would otherwise follow one by one): is the cost per instruction
//            or per interruption of the MFMA stream?
//   NVM      global loads per tile (buffer_load_dword from an L2-resident array, results unused until the end of the tile), GROUP at a time: what the V tile's 32
//            loads and the 16 LDS-DMA rows of the real kernel cost as instructions in the stream
//   WHERE    0: the VALU fillers behind all 512 MFMAs | 1: all of them inside the first product (the dependent chains) | 2: all inside the second product
// Output: one JSON object per variant: ms, TFLOP/s, fraction of the 157.3 TFLOP/s datasheet peak, shader cycles per MFMA (s_memtime) and the shader clock
// that the s_memtime / s_memrealtime ratio implies.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int LDS, int NVALU, int NTRANS, int NPK, int NACC, int BAR, int DEP, int WHERE = 0, int GROUP = 1, int NVM = 0>
__global__ __launch_bounds__(256, 1) void skel(int ntiles, float *sink, unsigned long long *clk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int LDY = 260, BUF = 64 * LDY;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    for (int i = tid; i < 2 * BUF; i += 256) lds[i] = 1.0f + (float)(i & 7);
    __syncthreads();
    float xreg[128];
#pragma unroll
    for (int s = 0; s < 128; ++s) { xreg[s] = 1.0f + 0.001f * (float)(s + lane); asm volatile("" : "+a"(xreg[s])); }
    f32x16 acc[8], sacc[2];
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[k][e] = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("" : "+a"(acc[k]));
    float fv = 1.0f + (float)lane, ft = 2.0f + (float)lane, ar = 0.0f;
    float2 fp = {1.0f, 2.0f};
    float fvs[8], fts[8];
    float2 fps[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { fvs[q] = fv + (float)q; fts[q] = ft + (float)q; fps[q] = fp; }
    const float c1 = 0.999f;
    const __amdgpu_buffer_rsrc_t gsrd = __builtin_amdgcn_make_buffer_rsrc((void *)(sink + 64 + 2048 * (blockIdx.x & 7)), 0, 16384 * 4, 0x00020000);
    const int gvoff = lane * 4;
    float vsum = 0.0f;
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int t = 0; t < ntiles; ++t) {
        if (BAR) __syncthreads();
        const float *Yt = lds + (t & 1) * BUF;
        int mf = 0;   // MFMAs issued so far in this tile (compile time after unrolling)
        int nv = 0, nt = 0, np = 0, na = 0, nm = 0;
        float vm[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) vm[q] = 0.0f;
        auto fill = [&]() {   // the fillers due behind MFMA number mf
            ++mf;
            if ((WHERE == 1 && mf > 256) || (WHERE == 2 && mf <= 256)) { __builtin_amdgcn_sched_barrier(0); return; }
            constexpr int SPAN = WHERE == 0 ? 512 : 256;                 // the fillers are spread over this many MFMAs
            const int mfl = WHERE == 2 ? mf - 256 : mf;
            // (independent registers per filler of a group: a group is GROUP issue slots, not a dependent chain)
            if (NVALU) while (nv < NVALU && (long)mfl * NVALU / SPAN >= nv + GROUP) { for (int q = 0; q < GROUP; ++q) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(fvs[q & 7]) : "v"(c1)); ++nv; } }
            if (NTRANS) while (nt < NTRANS && (long)mfl * NTRANS / SPAN >= nt + GROUP) { for (int q = 0; q < GROUP; ++q) { asm volatile("v_rcp_f32 %0, %0" : "+v"(fts[q & 7])); ++nt; } }
            if (NPK) while (np < NPK && (long)mfl * NPK / SPAN >= np + GROUP) { for (int q = 0; q < GROUP; ++q) { asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(fps[q & 7]) : "v"(fp)); ++np; } }
            if (NVM) while (nm < NVM && (long)mfl * NVM / SPAN >= nm + GROUP) { for (int q = 0; q < GROUP; ++q) { vm[nm & 31] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gsrd, gvoff, 256 * (nm & 31), 0)); ++nm; } }
            if (NACC && na < NACC && (long)mfl * NACC / SPAN > na) { asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(ar) : "a"(xreg[na & 127])); asm volatile("" :: "v"(ar)); ++na; }
            __builtin_amdgcn_sched_barrier(0);
        };
        // ---- first product: 2 x 128 MFMAs, LDS operand of the next group fetched before this group's MFMAs (as the kernel does)
        auto g1_read = [&](int ph, int g) -> float4 {
            if (LDS) return *reinterpret_cast<const float4 *>(Yt + (32 * ph + l31) * LDY + 8 * g + 4 * h);
            return float4{1.0f, 1.0f, 1.0f, 1.0f};
        };
        float4 a_cur = g1_read(0, 0);
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
#pragma unroll
            for (int g = 0; g < 32; ++g) {
                float4 a_nxt = a_cur;
                if (g + 1 < 32) a_nxt = g1_read(ph, g + 1);
                else if (ph == 0) a_nxt = g1_read(1, 0);
                const float av[4] = {a_cur.x, a_cur.y, a_cur.z, a_cur.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (DEP) {
                        if (g == 0 && e == 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=v"(sacc[ph]) : "v"(av[e]), "a"(xreg[4 * g + e]));
                        else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(sacc[ph]) : "v"(av[e]), "a"(xreg[4 * g + e]));
                    } else {
                        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[(4 * g + e) & 7]) : "v"(av[e]), "a"(xreg[4 * g + e]));
                    }
                    fill();
                }
                a_cur = a_nxt;
            }
        }
        if (DEP) asm volatile("s_nop 15\n\ts_nop 3");
        // ---- second product: 32 steps x 8 MFMAs, the next step's LDS operands fetched before this step's MFMAs
        auto g2_read = [&](int st, float (&y)[8]) {
            const int jb = st >> 4, reg = st & 15;
            const int row = 32 * jb + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            if (LDS == 1) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float4 t4 = *reinterpret_cast<const float4 *>(Yt + row * LDY + 128 * c + 4 * l31);
                    y[4 * c] = t4.x; y[4 * c + 1] = t4.y; y[4 * c + 2] = t4.z; y[4 * c + 3] = t4.w;
                }
            } else if (LDS == 2) {
#pragma unroll
                for (int kb = 0; kb < 8; ++kb) y[kb] = Yt[row * LDY + 32 * kb + l31];
            } else {
#pragma unroll
                for (int kb = 0; kb < 8; ++kb) y[kb] = 1.0f;
            }
        };
        float y_cur[8], y_nxt[8];
        g2_read(0, y_cur);
#pragma unroll
        for (int st = 0; st < 32; ++st) {
            if (st + 1 < 32) g2_read(st + 1, y_nxt);
            const float rr = DEP ? sacc[st >> 4][st & 15] : 1.0f;
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[kb]) : "v"(y_cur[kb]), "v"(rr));
                fill();
            }
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) y_cur[kb] = y_nxt[kb];
        }
        if (NVM) {
#pragma unroll
            for (int q = 0; q < 32; ++q) asm volatile("" :: "v"(vm[q]));   // the loads must happen; their values are not needed
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = fv + ft + fp.x + fp.y + ar;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += fvs[q] + fts[q] + fps[q].x + fps[q].y;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][7];
    if (DEP) s += sacc[0][0] + sacc[1][3];
    if (s == 12345.678f) sink[0] = s;   // keep everything alive
    if (blockIdx.x == 0 && tid == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

struct Row { const char *name; double ms, tf; unsigned long long cyc, wall; };

template <int LDS, int NVALU, int NTRANS, int NPK, int NACC, int BAR, int DEP, int WHERE = 0, int GROUP = 1, int NVM = 0>
Row run(const char *name, int ntiles, float *sink, unsigned long long *clk, int ncu) {
    auto k = skel<LDS, NVALU, NTRANS, NPK, NACC, BAR, DEP, WHERE, GROUP, NVM>;
    const size_t shm = 2 * 64 * 260 * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k, dim3(ncu), dim3(256), shm, 0, ntiles / 4, sink, clk);   // warm-up (clock ramp)
    CK(hipDeviceSynchronize());
    double best = 1e30;
    unsigned long long h[2] = {0, 0};
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(k, dim3(ncu), dim3(256), shm, 0, ntiles, sink, clk);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) { best = ms; CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost)); }
    }
    const double flop = (double)ncu * 4.0 * (double)ntiles * 512.0 * 4096.0;
    Row r{name, best, flop / (best * 1e-3) / 1e12, h[0], h[1]};
    const double cyc_per_mfma = (double)h[0] / ((double)ntiles * 512.0);
    printf("{\"variant\": \"%s\", \"ms\": %.4f, \"TFLOPs\": %.2f, \"frac_of_157.3\": %.4f, \"s_memtime_ticks\": %llu, \"s_memrealtime_ticks\": %llu, "
           "\"memtime_ticks_per_mfma\": %.3f, \"implied_MHz_if_memtime_is_shader_clock\": %.1f, \"mfma_per_us_per_simd\": %.2f}\n",
           name, best, r.tf, r.tf / 157.3, h[0], h[1], cyc_per_mfma, h[1] ? (double)h[0] / (double)h[1] * 100.0 : 0.0, (double)ntiles * 512.0 / (best * 1e3));
    fflush(stdout);
    return r;
}

int main(int argc, char **argv) {
    const int ntiles = argc > 1 ? atoi(argv[1]) : 3000;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("{\"device\": \"%s\", \"CUs\": %d, \"clockRate_kHz\": %d, \"tiles\": %d}\n", prop.gcnArchName, ncu, prop.clockRate, ntiles);
    float *sink;
    unsigned long long *clk;
    CK(hipMalloc(&sink, 64 + 4 * (64 + 8 * 2048 + 16384))); CK(hipMemset(sink, 0, 64 + 4 * (64 + 8 * 2048 + 16384))); CK(hipMalloc(&clk, 64));
    //    LDS NVALU NTRANS NPK NACC BAR DEP
    run<0, 0, 0, 0, 0, 0, 0>("mfma only, 8 independent accumulators", ntiles, sink, clk, ncu);
    run<0, 0, 0, 0, 0, 0, 1>("mfma only, first product as two dependent chains (the kernel's dependency structure)", ntiles, sink, clk, ncu);
    run<1, 0, 0, 0, 0, 0, 1>("+ LDS reads, b128 in both products (round 6)", ntiles, sink, clk, ncu);
    run<2, 0, 0, 0, 0, 0, 1>("+ LDS reads, second product as 8 ds_read_b32 per step (round 5)", ntiles, sink, clk, ncu);
    run<1, 0, 0, 0, 0, 1, 1>("+ LDS b128 + one s_barrier per tile", ntiles, sink, clk, ncu);
    run<1, 64, 0, 0, 0, 0, 1>("+ LDS b128 + 64 v_fma_f32 per tile", ntiles, sink, clk, ncu);
    run<1, 128, 0, 0, 0, 0, 1>("+ LDS b128 + 128 v_fma_f32 per tile", ntiles, sink, clk, ncu);
    run<1, 256, 0, 0, 0, 0, 1>("+ LDS b128 + 256 v_fma_f32 per tile", ntiles, sink, clk, ncu);
    run<1, 0, 64, 0, 0, 0, 1>("+ LDS b128 + 64 v_rcp_f32 per tile", ntiles, sink, clk, ncu);
    run<1, 0, 128, 0, 0, 0, 1>("+ LDS b128 + 128 v_rcp_f32 per tile", ntiles, sink, clk, ncu);
    run<1, 0, 0, 64, 0, 0, 1>("+ LDS b128 + 64 v_pk_fma_f32 per tile", ntiles, sink, clk, ncu);
    run<1, 0, 0, 128, 0, 0, 1>("+ LDS b128 + 128 v_pk_fma_f32 per tile", ntiles, sink, clk, ncu);
    run<1, 0, 0, 0, 64, 0, 1>("+ LDS b128 + 64 v_accvgpr_read per tile", ntiles, sink, clk, ncu);
    run<1, 128, 0, 0, 0, 0, 1, 1>("+ LDS b128 + 128 v_fma_f32 per tile, ALL inside the first product (dependent MFMA chains)", ntiles, sink, clk, ncu);
    run<1, 128, 0, 0, 0, 0, 1, 2>("+ LDS b128 + 128 v_fma_f32 per tile, ALL inside the second product (independent accumulators)", ntiles, sink, clk, ncu);
    run<1, 128, 0, 0, 0, 0, 0, 1>("+ LDS b128 + 128 v_fma_f32 per tile, all inside the first half, first product on independent accumulators", ntiles, sink, clk, ncu);
    run<1, 0, 64, 0, 0, 0, 1, 1>("+ LDS b128 + 64 v_rcp_f32 per tile, all inside the first product", ntiles, sink, clk, ncu);
    run<1, 0, 64, 0, 0, 0, 1, 2>("+ LDS b128 + 64 v_rcp_f32 per tile, all inside the second product", ntiles, sink, clk, ncu);
    run<1, 128, 0, 0, 0, 0, 1, 0, 2>("+ LDS b128 + 128 v_fma_f32 per tile in groups of 2 (behind every 8th MFMA)", ntiles, sink, clk, ncu);
    run<1, 128, 0, 0, 0, 0, 1, 0, 4>("+ LDS b128 + 128 v_fma_f32 per tile in groups of 4 (behind every 16th MFMA)", ntiles, sink, clk, ncu);
    run<1, 128, 0, 0, 0, 0, 1, 0, 8>("+ LDS b128 + 128 v_fma_f32 per tile in groups of 8 (behind every 32nd MFMA)", ntiles, sink, clk, ncu);
    run<1, 0, 64, 0, 0, 0, 1, 0, 2>("+ LDS b128 + 64 v_rcp_f32 per tile in groups of 2", ntiles, sink, clk, ncu);
    run<1, 0, 64, 0, 0, 0, 1, 0, 4>("+ LDS b128 + 64 v_rcp_f32 per tile in groups of 4", ntiles, sink, clk, ncu);
    run<1, 0, 0, 0, 0, 0, 1, 0, 1, 32>("+ LDS b128 + 32 global loads per tile, one by one", ntiles, sink, clk, ncu);
    run<1, 0, 0, 0, 0, 0, 1, 0, 2, 32>("+ LDS b128 + 32 global loads per tile in groups of 2 (the kernel's V loads)", ntiles, sink, clk, ncu);
    run<1, 0, 0, 0, 0, 0, 1, 0, 8, 32>("+ LDS b128 + 32 global loads per tile in groups of 8", ntiles, sink, clk, ncu);
    run<1, 0, 0, 0, 0, 0, 1, 2, 2, 32>("+ LDS b128 + 32 global loads per tile in groups of 2, all inside the second product", ntiles, sink, clk, ncu);
    run<1, 0, 64, 64, 0, 1, 1, 0, 2>("round-6 KL W-step tile, its 64 trans + 64 packed in groups of 2 each", ntiles, sink, clk, ncu);
    run<1, 0, 64, 64, 0, 1, 1, 0, 4>("round-6 KL W-step tile, its 64 trans + 64 packed in groups of 4 each", ntiles, sink, clk, ncu);
    run<1, 0, 64, 64, 0, 1, 1>("round-6 KL W-step tile without its global loads: LDS b128 + 64 trans + 64 packed + barrier", ntiles, sink, clk, ncu);
    run<2, 64, 64, 0, 64, 1, 1>("round-5 KL W-step tile without its global loads: LDS b32 + 64 trans + 64 fma + 64 accvgpr moves + barrier", ntiles, sink, clk, ncu);
    return 0;
}
