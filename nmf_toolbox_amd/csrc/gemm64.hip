// The small products of the euclidean Gram form with float64 accumulation, on the fp64 matrix core (v_mfma_f64_16x16x4_f64, gfx950):
//     P = W * (H*H')          (nmf.m:149-150 in the form of SURVEY A.2: V_hat*H' = W*(H*H'))
// The K-long fp32 accumulation chain of this product -- not the storage of W -- is what put H-fixed problems past the 1e-5 contract (round 4: 1.06e-5 ... 1.7e-5
// on W; the chain's rounding differs from iteration to iteration and the W update amplifies it; scripts/emu_precision.py).  m*K*K multiply-adds: a few
// per cent of the n-long contractions next to it, so it is simply done in double, with the float64 master copy of W as the left operand.
#include <cstdlib>
#include <type_traits>
#include <utility>
#include "nmfx_internal.h"

namespace nmfx {

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int G64_BM = 64, G64_BN = 32, G64_BK = 16;
constexpr int G64_LD = 80;   // LDS row stride in doubles: 16 (mod 32), so the two k-rows a 32-lane group of ds_read_b64 touches fall on disjoint bank halves

// C(i, j) = sum_k A(i, k) * B(k, j);  A(i, k) = A[i + lda*k] (float64 or fp32), B(k, j) = B[k + ldb*j] (fp32 or float64), C[i + ldc*j] (float64 and / or fp32)
// 4 waves, each a 32 x 16 block of the 64 x 32 tile (two MFMA blocks along i).  The MFMA is fed transposed (first operand = B', second = A) so that the 16 lanes of a
// result register run along i, the contiguous dimension of C.
// These products are small (cnmf at C4: 4096 x 512 x 512) and the kernel is latency-bound, not MFMA-bound: the PMC pass of round 5's first version (64 x 64 tiles, one
// tile prefetched) shows the matrix core busy 13 % and the waves waiting 63 % of their cycles (profiles/r5_20_c4_pmc.md).  Measured at the engine's shapes
// (scripts/bench_gemm64.py, profiles/r5_24_gemm64_tiles.txt): 64 x 64 tiles 77 / 104 us (4096 x 512 x 512 / 16384 x 256 x 256), 64 x 32 tiles -- twice the workgroups,
// four and more per CU -- 65 / 69 us, 32 x 32 tiles 81 / 78 us; a second tile in flight (two register sets: the tile that goes to LDS at the end of a trip was
// requested a trip earlier) is worth 1-2 us on top.  33 TFLOP/s of the fp64 matrix core's 78; C4's product 106 -> 71 us inside the iteration.
template <bool A64, bool B64>
__global__ __launch_bounds__(256) void gemm64_kernel(const void *__restrict__ Ap, long lda, const void *__restrict__ Bp, long ldb, long M, long N, long Kc,
                                                      double *__restrict__ C64, float *__restrict__ C32, long ldc) {
    constexpr int NB = 1, MB = 2, PF = 2;                  // the 64 x 32 tile with two tiles in flight: the fastest of the (MB, NB, PF) variants tried, see above
    constexpr int BN = 32 * NB, BPT = BN * G64_BK / 256;   // B elements per thread and tile
    constexpr int LDB = NB == 2 ? G64_LD : 48;             // 48 = 16 (mod 32) as well
    constexpr int BM = 32 * MB, APT = BM * G64_BK / 256, LDA = MB == 2 ? G64_LD : 48;
    __shared__ double As[2][G64_BK * LDA], Bs[2][G64_BK * LDB];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w & 1, wn = w >> 1;
    const long i0 = (long)blockIdx.x * BM, j0 = (long)blockIdx.y * BN;
    // loaders: A -- thread (row = tid & 63, k = 4*(tid >> 6) + u); B -- thread (col = tid / (16 / BPT), k = BPT*(tid % (16 / BPT)) + u)
    const int a_row = tid % BM, a_k = APT * (tid / BM);
    const int b_col = tid / (G64_BK / BPT), b_k = BPT * (tid % (G64_BK / BPT));
    const bool a_ok = i0 + a_row < M, b_ok = j0 + b_col < N;
    double ra[2][APT], rb[2][BPT];
    auto gload = [&](long k0, int set) {
#pragma unroll
        for (int u = 0; u < APT; ++u) {
            const long ka = k0 + a_k + u;
            double va = 0.0;
            if (a_ok && ka < Kc) va = A64 ? static_cast<const double *>(Ap)[i0 + a_row + lda * ka] : (double)static_cast<const float *>(Ap)[i0 + a_row + lda * ka];
            ra[set][u] = va;
        }
#pragma unroll
        for (int u = 0; u < BPT; ++u) {
            const long kb = k0 + b_k + u;
            double vb = 0.0;
            if (b_ok && kb < Kc) vb = B64 ? static_cast<const double *>(Bp)[kb + ldb * (j0 + b_col)] : (double)static_cast<const float *>(Bp)[kb + ldb * (j0 + b_col)];
            rb[set][u] = vb;
        }
    };
    auto lstore = [&](int buf, int set) {
#pragma unroll
        for (int u = 0; u < APT; ++u) As[buf][(a_k + u) * LDA + a_row] = ra[set][u];
#pragma unroll
        for (int u = 0; u < BPT; ++u) Bs[buf][(b_k + u) * LDB + b_col] = rb[set][u];
    };
    f64x4 acc[MB][NB];
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[a][b][e] = 0.0;
    const int nk = (int)((Kc + G64_BK - 1) / G64_BK);
    gload(0, 0);
    lstore(0, 0);
    if (PF == 2 && nk > 1) gload((long)G64_BK, 1);
    __syncthreads();
    const int l15 = lane & 15, lk = lane >> 4;
    auto trip = [&](int k1, auto set_far_c) {   // tile k1 is in LDS buffer k1 & 1; tile k1 + 1 in register set (k1 + 1) & 1; tile k1 + 2 is requested into set k1 & 1
        constexpr int set_far = decltype(set_far_c)::value, set_next = set_far ^ 1;
        const int buf = k1 & 1;
        if (PF == 2) { if (k1 + 2 < nk) gload((long)(k1 + 2) * G64_BK, set_far); }
        else if (k1 + 1 < nk) gload((long)(k1 + 1) * G64_BK, set_next);   // PF == 1: one tile ahead, one register set in use at a time
#pragma unroll
        for (int kk = 0; kk < G64_BK / 4; ++kk) {
            double af[MB], bf[NB];
#pragma unroll
            for (int a = 0; a < MB; ++a) af[a] = As[buf][(4 * kk + lk) * LDA + 16 * MB * wm + 16 * a + l15];
#pragma unroll
            for (int b = 0; b < NB; ++b) bf[b] = Bs[buf][(4 * kk + lk) * LDB + 16 * NB * wn + 16 * b + l15];
#pragma unroll
            for (int a = 0; a < MB; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[b], af[a], acc[a][b], 0, 0, 0);   // D(row = j, col = i)
        }
        // careful with the order: set_far was requested above and must not be overwritten before it is stored -- it is stored in the NEXT trip (as set_next there)
        if (k1 + 1 < nk) lstore(buf ^ 1, set_next);   // (the other LDS buffer: last read one trip ago, behind the barrier below)
        __syncthreads();
    };
    for (int kt = 0; kt < nk; kt += 2) {   // two trips per turn: the register sets swap roles by NAME (no dynamic register indexing)
        trip(kt, std::integral_constant<int, 0>{});
        if (kt + 1 < nk) trip(kt + 1, std::integral_constant<int, 1>{});
    }
    // acc[a][b][e]: column (lane & 15) -> i, row (lane >> 4) + 4 e -> j
#pragma unroll
    for (int a = 0; a < MB; ++a) {
        const long i = i0 + 16 * MB * wm + 16 * a + l15;
        if (i >= M) continue;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const long j = j0 + 16 * NB * wn + 16 * b + lk + 4 * e;
                if (j >= N) continue;
                const double v = acc[a][b][e];
                if (C64) C64[i + ldc * j] = v;
                if (C32) C32[i + ldc * j] = (float)v;
            }
    }
}

// ---- the engine's own shape: A = float64 master copy of W (m x Kc, m large), B = a Gram matrix in fp32 (Kc x N, Kc = N = K*T <= 512) ---------------------------------
// Round 6.  The tiled kernel above re-stages BOTH operands through LDS every 16 k and meets at a barrier each time: 8 MFMAs per wave between barriers, 33 TFLOP/s.
// Here a workgroup keeps the WHOLE k-extent of a 32-column panel of B in LDS, converted to float64 once (32 x Kc x 8 B = 132 KB at Kc = 512: one workgroup of eight
// waves per CU; two below Kc = 256), and streams its rows of A from global memory straight into MFMA operand registers -- no LDS for A, no barrier after the
// prologue, every wave on its own with the A operands of the next 16-k group in flight.  The contraction order inside a group is free as long as both operands
// agree: lane (i or j = lane & 15, q = lane >> 4) takes k = 16 g + 4 q + s in step s, so its four B values of a group are two ds_read_b128 of the k-contiguous LDS
// column (column stride = 2 (mod 4) doubles: the 16 lanes of a read phase fall on 16 disjoint bank quads), and its A values are 8-byte loads whose 16-lane groups
// cover 128 contiguous bytes.  Workgroups that share a row chunk of A (one per column panel) are consecutive on ONE XCD (the XCDs take workgroups round-robin): A
// leaves HBM once.
// What was tried on the way (profiles/r6_20_gemm64_variants.txt; all between 42 and 48 TFLOP/s at 4096 x 512 x 512): B as fp32 in LDS with the conversion in the loop
// (a v_cvt_f64_f32 between MFMAs costs the interruption); 64-column panels (half the L2 reads of A); 4 / 8 / 16 waves per workgroup; one to six groups of A in flight.
// Two things the compiler does to such a loop had to be switched off first, and neither moved the number either: with a branch around the prefetch loads its
// s_waitcnt placement gives up counting and drains the queue every group; without a sched_barrier behind them it sinks every load down to its first use.  The kernel
// is ONE round of workgroups: its prologue (the panel) and its epilogue (16.8 MB of float64 stores at config 4) run on every CU at the same time with nothing to hide
// under, and the fp64 MFMA itself reaches 63 ... 72 TFLOP/s on this part, not the 78.6 of the data sheet (scripts/ubench_mfma64.hip, profiles/r6_20_ubench_mfma64.jsonl).
template <typename F, int... I>
__device__ __forceinline__ void for_each_set(std::integer_sequence<int, I...>, F f) { (f(std::integral_constant<int, I>{}), ...); }

template <int MB, int NW, int PD, int NB>
__global__ __launch_bounds__(64 * NW) void gemm64_panel_kernel(const double *__restrict__ A, long lda, const float *__restrict__ B, long ldb, long M, long N, int Kc, int ldk,
                                                            int n_panels, double *__restrict__ C64, float *__restrict__ C32, long ldc) {
    extern __shared__ __attribute__((aligned(16))) double Bs[];   // Bs[j * ldk + k], j < 16 * NB, converted ONCE: a v_cvt_f64_f32 between MFMAs costs the interruption, not the instruction
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, q = lane >> 4;
    const long id = blockIdx.x, xcd = id % 8, r = id / 8;
    const long panel = r % n_panels, chunk = xcd + 8 * (r / n_panels);
    const long i0 = chunk * (16 * NW * MB) + (long)w * (16 * MB), j0 = panel * (16 * NB);
    if (chunk * (16 * NW * MB) >= M) return;   // (the chunk count is padded to a multiple of 8; the whole workgroup leaves)
    {   // prologue: the panel, k-contiguous on both sides
        const int k4n = Kc / 4;
        for (int idx = tid; idx < 16 * NB * k4n; idx += 64 * NW) {
            const int j = idx / k4n, k4 = idx - j * k4n;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j0 + j < N) {
                const float *src = B + ldb * (j0 + j) + 4 * k4;
                if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) v = *reinterpret_cast<const float4 *>(src);
                else v = make_float4(src[0], src[1], src[2], src[3]);
            }
            double *dst = &Bs[j * ldk + 4 * k4];
            *reinterpret_cast<double2 *>(dst) = make_double2((double)v.x, (double)v.y);
            *reinterpret_cast<double2 *>(dst + 2) = make_double2((double)v.z, (double)v.w);
        }
    }
    __syncthreads();
    const int ng = Kc / 16;
    const double *ap[MB];
#pragma unroll
    for (int a = 0; a < MB; ++a) {
        const long i = i0 + 16 * a + l15;
        ap[a] = A + (i < M ? i : M - 1) + lda * (4 * q);   // rows past the end are computed from row M-1 and dropped
    }
    constexpr int NS = PD + 1;   // register sets of A operands: the group being multiplied + PD groups in flight
    double av[NS][4][MB];
    auto load_a = [&](int g, auto set_c) {
        constexpr int set = decltype(set_c)::value;
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
            for (int a = 0; a < MB; ++a) av[set][s2][a] = ap[a][lda * (16 * (long)g + s2)];
    };
    f64x4 acc[MB][NB];
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[a][b][e] = 0.0;
    const double *bp = Bs + l15 * ldk + 4 * q;
    auto group = [&](int g, auto cur_c) {
        constexpr int cur = decltype(cur_c)::value, far = (cur + PD) % NS;
        load_a(g + PD < ng ? g + PD : ng - 1, std::integral_constant<int, far>{});   // unconditional: the last PD groups re-request the last one (see above)
        __builtin_amdgcn_sched_barrier(0);
        double bf[4][NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const double2 lo = *reinterpret_cast<const double2 *>(bp + 16 * b * ldk + 16 * g), hi = *reinterpret_cast<const double2 *>(bp + 16 * b * ldk + 16 * g + 2);
            bf[0][b] = lo.x; bf[1][b] = lo.y; bf[2][b] = hi.x; bf[3][b] = hi.y;
        }
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
            for (int a = 0; a < MB; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[s2][b], av[cur][s2][a], acc[a][b], 0, 0, 0);   // D(row = j, col = i)
    };
    for_each_set(std::make_integer_sequence<int, PD>{}, [&](auto c) { load_a(decltype(c)::value < ng ? decltype(c)::value : ng - 1, c); });
    for (int g = 0; g < ng; g += NS)   // NS groups per turn, ng a multiple of NS (the launcher checks): the register sets rotate by NAME (no dynamic register indexing)
        for_each_set(std::make_integer_sequence<int, NS>{}, [&](auto c) { group(g + decltype(c)::value, c); });
    // acc[a][b][e]: column (lane & 15) -> i, row (lane >> 4) + 4 e -> j
#pragma unroll
    for (int a = 0; a < MB; ++a) {
        const long i = i0 + 16 * a + l15;
        if (i >= M) continue;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const long j = j0 + 16 * b + q + 4 * e;
                if (j >= N) continue;
                const double v = acc[a][b][e];
                if (C64) C64[i + ldc * j] = v;
                if (C32) C32[i + ldc * j] = (float)v;
            }
    }
}
constexpr int G64_PANEL_KC_MAX = 512;
template <int MB, int NW, int PD, int NB>
nmfx_status launch_panel(hipStream_t st, const double *A, long lda, const float *B, long ldb, long M, long N, long Kc, double *C64, float *C32, long ldc) {
    const int ldk = (int)Kc + 2;   // doubles: 2*ldk words = 4 (mod 8) -- the 16 lanes of a read phase fall on 16 disjoint bank quads
    const int n_panels = (int)((N + 16 * NB - 1) / (16 * NB));
    const long chunks = ((M + 16 * NW * MB - 1) / (16 * NW * MB) + 7) / 8 * 8;
    const size_t lds = sizeof(double) * 16 * NB * (size_t)ldk;
    static LdsAttrOnce lds_attr;
    TRY(lds_attr.set(reinterpret_cast<const void *>(gemm64_panel_kernel<MB, NW, PD, NB>), (int)(sizeof(double) * 16 * NB * (G64_PANEL_KC_MAX + 2))));
    hipLaunchKernelGGL((gemm64_panel_kernel<MB, NW, PD, NB>), dim3((unsigned)(chunks * n_panels)), dim3(64 * NW), lds, st, A, lda, B, ldb, M, N, (int)Kc, ldk, n_panels, C64, C32, ldc);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

}  // namespace

nmfx_status gemm64(hipStream_t st, long M, long N, long Kc, const double *A64, const float *A32, long lda, const double *B64, const float *B32, long ldb,
                   double *C64, float *C32, long ldc) {
    if (M <= 0 || N <= 0) return NMFX_OK;
    if ((!A64 && !A32) || (!B64 && !B32) || (!C64 && !C32) || Kc <= 0) { set_error("gemm64: bad arguments"); return NMFX_ERR_INVALID; }
    const void *A = A64 ? static_cast<const void *>(A64) : static_cast<const void *>(A32);
    const void *B = B64 ? static_cast<const void *>(B64) : static_cast<const void *>(B32);
    static const bool no_panel = getenv("NMFX_GEMM64_TILED") != nullptr;   // A/B: the round-5 kernel for every shape
    if (A64 && B32 && !no_panel && Kc % 32 == 0 && Kc <= G64_PANEL_KC_MAX && M >= 256) {   // (Kc / 16 groups, two register sets)
        // eight waves share a panel (a workgroup per CU at Kc = 512, two below 256); 32-row waves where that still gives every CU a workgroup, 16-row waves otherwise
        // (profiles/r6_20_gemm64_variants.txt: 4096 x 512 x 512 44.4 us / 48 TFLOP/s against 64.8 / 33 of the tiled kernel, 8192 x 128 x 128 9.4 against 13.0; the fp64
        // MFMA alone reaches 63-72 TFLOP/s on this part (scripts/ubench_mfma64.hip), and one round of workgroups leaves its prologue and its 16.8 MB of stores exposed)
        const long panels = (N + 31) / 32;
        if (((M + 255) / 256) * panels >= 256) return launch_panel<2, 8, 1, 2>(st, A64, lda, B32, ldb, M, N, Kc, C64, C32, ldc);
        return launch_panel<1, 8, 1, 2>(st, A64, lda, B32, ldb, M, N, Kc, C64, C32, ldc);
    }
    dim3 grid((unsigned)((M + G64_BM - 1) / G64_BM), (unsigned)((N + G64_BN - 1) / G64_BN));
    if (A64 && B64) hipLaunchKernelGGL((gemm64_kernel<true, true>), grid, dim3(256), 0, st, A, lda, B, ldb, M, N, Kc, C64, C32, ldc);
    else if (A64) hipLaunchKernelGGL((gemm64_kernel<true, false>), grid, dim3(256), 0, st, A, lda, B, ldb, M, N, Kc, C64, C32, ldc);
    else if (B64) hipLaunchKernelGGL((gemm64_kernel<false, true>), grid, dim3(256), 0, st, A, lda, B, ldb, M, N, Kc, C64, C32, ldc);
    else hipLaunchKernelGGL((gemm64_kernel<false, false>), grid, dim3(256), 0, st, A, lda, B, ldb, M, N, Kc, C64, C32, ldc);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

}  // namespace nmfx

extern "C" nmfx_status nmfx_gemm64(void *stream, int64_t M, int64_t N, int64_t Kc, const double *A64, const float *A32, int64_t lda, const double *B64,
                                   const float *B32, int64_t ldb, double *C64, float *C32, int64_t ldc) {
    return nmfx::gemm64(static_cast<hipStream_t>(stream), M, N, Kc, A64, A32, lda, B64, B32, ldb, C64, C32, ldc);
}
