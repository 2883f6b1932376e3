// The fused two-stage MFMA kernel template (instantiated by the fused_*.hip translation units through fused_launch.h).
#pragma once
#include "nmfx_internal.h"

namespace nmfx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int FT_ROWS = 128;  // stationary rows per workgroup (32 per wave)
constexpr int FT_C = 64;      // streamed rows (contraction tile of the second product) per step

// packed fp32 VALU (gfx90a+): two lanes' worth of work per instruction on a 64-bit register pair.  hipcc scalarises <2 x float> arithmetic whose operands are
// assembled from scalars, so these are asm.  The hazard recogniser does not look inside inline asm: a non-transcendental VALU instruction that reads a register
// written by v_rcp / v_log one instruction earlier needs one wait state on gfx940+ (LLVM: hasTransForwardingHazard) -- the `_t` forms carry it themselves
__device__ __forceinline__ f32x2 pk_mul_t(f32x2 a, f32x2 b) { f32x2 d; asm("s_nop 0\n\tv_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ f32x2 pk_fma_t(f32x2 a, f32x2 b, f32x2 c) { f32x2 d; asm("s_nop 0\n\tv_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ f32x2 pk_fnma(f32x2 a, f32x2 b, f32x2 c) { f32x2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }   // c - a.*b
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { f32x2 d; asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }

// Wait states between an asm MFMA and the first VALU / store that reads its result: the hazard recogniser cannot see into the asm, and a 16-pass MFMA needs 18.
// The accumulator is an in/out operand of the waiting asm, so that NOTHING that reads it can be moved in front of the wait (a pure builtin such as v_rcp has no other
// reason to stay behind an asm volatile).  Where another MFMA follows the chain anyway, the wait is part of THAT MFMA's asm statement (its 16 passes + `s_nop 3`);
// this full form is for the cost-only passes, where nothing follows.  What the kernel cannot prevent -- a register copy of the tuple that hipcc itself places
// right behind an asm MFMA -- is checked in the built objects: scripts/mfma_asm_lint.py (tests/test_abi_and_host.py)
__device__ __forceinline__ void mfma_settle_full(f32x16 &t) { asm volatile("s_nop 15\n\ts_nop 3" : "+v"(t)); }
__device__ __forceinline__ constexpr int rowmap(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }
// raw buffer descriptor (gfx950): base, stride 0, num_records bytes, 32-bit raw format
__device__ __forceinline__ i32x4 make_srd(const void *base, unsigned bytes) {
    const unsigned long long b = (unsigned long long)base;
    i32x4 s;
    s.x = (int)(unsigned)b;
    s.y = (int)((b >> 32) & 0xffffu);
    s.z = (int)bytes;
    s.w = 0x00020000;
    return s;
}

// FUNC: 0 R=V, no S | 1 R=V, S only for the euclidean cost | 2 R=V./S (KL) | 3 R=V./S + KL cost
//       4 IS (nmf.m:155-156,186-187,212): TWO element maps per pass, A = V./S.^2 and B = 1./S, two accumulator sets (K <= 192)
//       5 alpha-beta, alpha ~= 0 (nmf.m:162-163,193-194,214): A = V.^alpha .* S.^(beta-1), B = S.^(alpha+beta-1); D holds V.^alpha
//       6 R = S - V (the residual) + euclidean cost: the gradients of nmfsc.m:144-148,194-200 in ONE contraction, dH = W'*(W*H - V) /
//         dW = (W*H - V)*H', instead of the difference of two separately rounded products (which cancels as the fit improves)
//       7 / 8 (cost-only form, W-step form): the first product in SEVERAL launches over column blocks of a factor wider than 256 (KL with K > 256,
//         nmf.m:152-153,183-184 have no K limit).  Every launch starts its S tile from the partial sums of the launches before it (p.Sin, read like the
//         V tile; nullptr = zeros) and contracts its own <= 256 components: 7 stores the raw partial S (p.Rout), 8 is the last block and goes on
//         like 3: R = V./S (+ KL cost terms), R stored to p.Rout for the numerator passes; 10 is the last block of a euclidean chain: the residual
//         sum (V - S).^2 of the accumulated S (functor 1's terms), nothing stored
//       1 in the cost-only form with p.Rout: the raw S = V_hat is stored as well (cnmfsc.m:269); 21: the same pass storing the RESIDUAL S - V instead -- all the
//         sparse-H step of cnmfsc consumes of V_hat (cnmfsc.m:160-168: dH = pos - neg = sum_t W_t' * lshift_t(V_hat - V)): its Q product then streams ONE m x n
//         operand instead of V_hat and V, and the difference is formed from the fp32 S in registers, before it is rounded to memory
//       11 / 12 (IS) and 13 / 14 (alpha-beta, alpha ~= 0): the two element maps of functors 4 / 5 as TWO single-map passes, for 192 < K <= 256 where a second
//         accumulator set no longer fits the register file (nmf.m:154-164,185-195 have no K limit).  11: A = V./S.^2 (+ the IS cost terms), 12: B = 1./S,
//         13: A = V.^alpha .* S.^(beta-1) (+ the alpha-beta cost terms; D holds V.^alpha), 14: B = S.^(alpha+beta-1).  Each is the KL pass with another map:
//         S is formed twice (8*m*n*K per half-iteration instead of 6) but V_hat never reaches HBM
//       17: the DUAL form of the alpha-beta divergence (alpha == 0; nmf.m:124-128,159-160,190-191): numerators A = V.^(-1) .* S.^beta.  (Its denominators
//         V.^(beta-1) * H' do not involve S: functor 0 on a precomputed V.^(beta-1).)  No cost terms: the reference's cost divides by alpha*beta = 0
//       15 / 16 (W-step form, second product on): 11 / 13 that ALSO leave the second map's values B in HBM (p.Rout, m x n) -- 1./S and S.^(alpha+beta-1) are
//         by-products of the first map, so storing them costs one buffer_store per element -- for a no-first-product pass (functor 0 with D = that buffer) to
//         contract: 4 + 2 = 6*m*n*K per W step instead of 8.  The stores ride behind the MFMAs of the second product; the LDS-DMA rows of the next tile are
//         waited for right behind P2, while nothing but loads is in flight (stores and loads do not retire in order relative to each other)
// RAG: p.R / p.Cn need not be multiples of 128 / 64.  Stationary rows past R load zeros, keep their (garbage, row-local) results to
// themselves and are neither stored nor costed; streamed indices past the end arrive as zero rows (buffer bounds) and their R
// elements and cost terms are masked to zero, so they add nothing to the second product.  Two extra VALU ops per element.
// TT > 1 (cnmf, W-step form only): the contraction index k' = (p, k), p = 0..TT-1, k = 0..K/TT-1, addresses H(k, j - t) with
// t = TT-1-p (cnmf.m:188 / RFD.m:36-38: V_hat = sum_t W_t * rshift_t(H)), i.e. streamed row j is the K floats that start at column
// j-(TT-1) of H -- consecutive rows OVERLAP in memory.  The LDS tile therefore holds FT_C + TT-1 columns of H (row stride KH + 4) and
// row c of the tile starts at LDS row c; nothing is replicated.  p.Y must be preceded by TT-1 readable columns (zeros, or the left
// halo of a column shard).  X / out slice t = TT-1-p lives at xs_t / os_t.
template <int K, bool D_RC, int FUNC, bool DO_G2, int EPI, bool RAG = false, int TT = 1>
__global__ __launch_bounds__(256, ((K <= 128 && FUNC != 4 && FUNC != 5) ? 2 : 1)) void fused_kernel(const FusedParams p) {   // K <= 128 fits two workgroups per CU (256 VGPRs, 2 x 68 KB LDS)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    static_assert(TT == 1 || (D_RC && EPI == 0 && (K / TT) % 32 == 0 && K % TT == 0), "TT > 1: W-step form, K/TT a multiple of 32");
    if (p.run_if && *p.run_if == 0) return;   // conditional launch (the explicit cost pass behind the Gram-form cost): uniform, one scalar load
    constexpr int KH = K / TT;             // floats per column of H
    constexpr int LDY = KH + 4;            // LDS row stride (one column of H per row)
    constexpr int NKB = K / 32;
    constexpr int TROWS = FT_C + TT - 1;   // LDS rows per tile
    constexpr int BUF = TROWS * LDY;
    constexpr bool NEED_S = FUNC != 0;
    constexpr bool S_IN = FUNC == 7 || FUNC == 8 || FUNC == 10 || FUNC == 19 || FUNC == 20;   // S accumulated over several launches (column blocks of a wide factor)
    constexpr int MF = FUNC == 8 ? 3 : ((FUNC == 10 || FUNC == 21) ? 1 : FUNC);   // the element map
    static_assert(!S_IN || (!DO_G2 && D_RC && TT == 1), "partial-S passes: first product only, W-step form");
    constexpr bool DUAL = FUNC == 4 || FUNC == 5;   // two element maps, two accumulator sets
    // element maps that never look at V (1./S, S.^(a+b-1)): no V loads are issued at all.  They must not merely be left unused: hipcc drops dead loads, and the
    // tile-top wait below counts on exactly 32 V loads standing behind the DMA rows in the in-order counter -- with fewer, `vmcnt(32)` returns before the rows
    // have landed and the tile is read half-written (found in round 4 as run-to-run differences of IS with K = 256 on three shards)
    constexpr bool NO_V = FUNC == 12 || FUNC == 14;
    constexpr int NVL = (!D_RC && !RAG) ? 8 : 32;   // V-load instructions per tile (load_d_piece)
    // H-step form (D_RC false): the two sources of the second product's MFMA change places, which TRANSPOSES the accumulator -- acc[kb][reg] = O(k <-> lane, r <->
    // register) instead of O(k <-> register, r <-> lane).  The output of this form is K-contiguous (H, or a K x n slab), so a lane then owns 4 consecutive k of a column and
    // 32 lanes write 512 contiguous bytes of it; with r on the lanes every store instruction scattered 64 four-byte pieces at stride K, and once NMFX_G2_VEC had
    // changed which k a register holds the L2 no longer merged them: 1.2e9 B written per H-step pass at C3 for 2.0e8 B of H and its master (profiles/r6_10_*_pmc.md)
    // EPI 2 = EPI 0 of the W-step form with the transposed accumulator as well: the no-first-product passes whose output is K-contiguous -- the H-step numerators
    // of the euclidean paths as transposed products, (V'*W)' and cnmf's Q = (V'*W_flat)' (launch_fused picks it from the output strides)
    constexpr bool SWAP = (!D_RC || EPI == 2) && NMFX_G2_VEC;
    static_assert(EPI != 2 || (D_RC && DO_G2 && FUNC == 0 && TT == 1), "EPI 2: W-step form, no first product");
    constexpr bool STB = FUNC == 15 || FUNC == 16;                                   // first map + store of the second map's values
    // 19 / 20: the LAST block of an IS / alpha-beta chain over a factor wider than 256 (like 8 for KL): the accumulated S goes through map 11 / 13 and both maps'
    // values are stored
    constexpr int EF = (FUNC == 15 || FUNC == 19) ? 11 : ((FUNC == 16 || FUNC == 20) ? 13 : FUNC);   // the element map to run
    static_assert(!STB || (DO_G2 && D_RC && EPI == 0 && TT == 1), "functors 15 / 16: W-step form with the second product");
    // functors 11 / 13 in the cost-only form: BOTH element maps' values go to HBM (p.Rout the first, p.Rout2 the second) -- the S pass of IS / alpha-beta cnmf, whose
    // numerator passes (functor 0 on either buffer) have no room for a first product next to K*T = 512 accumulators
    constexpr bool ST2 = (EF == 11 || EF == 13) && !DO_G2 && D_RC;
    constexpr bool BQ = STB || ST2;                                                  // the second map's values of a tile are kept in bq[] on their way out
    // first-product-only passes wait for the next tile's DMA rows right behind P2 -- they went out during P1 -- instead of at the next tile top, where
    // the R / S stores of this tile would stand between them and the V loads in the in-order counter and get waited for as well (an HBM write round
    // trip per tile: c4kl's S pass)
    constexpr bool EARLY = !DO_G2 && NEED_S;
    // the asm form of the first product (NMFX_G1_ASM).  Chain kernels (S_IN): the chain starts from zero like everywhere else and the partial sums of the blocks before
    // are ADDED once it has settled (SADD) -- with the partial S as the chain's initial value the accumulator tuple is loop-carried through the prefetch registers,
    // and hipcc re-homed it with `v_mov_b64` copies right behind an asm MFMA, i.e. read it 1-2 wait states after an instruction that delivers 18 later: every
    // K > 256 test failed on the hardware (round 6; scripts/chain_kernel_check.hip reproduces it, scripts/mfma_asm_lint.py finds it in the objects)
    constexpr bool G1A = NMFX_G1_ASM;
    constexpr bool SADD = S_IN && G1A;
    constexpr bool PK = NMFX_KL_MODE == 2 && (MF == 2 || MF == 3) && !DUAL && EF == FUNC;   // the packed KL map, in bursts per double pair (emap_burst below)
    constexpr int NU = (DUAL || EF == 11 || EF == 13) ? 8 : ((MF == 3 && NMFX_KL_MODE == 1) ? 6 : 4);   // micro-ops per element of the one-by-one element maps (everything but PK)
    static_assert(!DUAL || (K <= 192 && TT == 1), "dual-map kernels: K <= 192 (two accumulator sets + the stationary operand must fit 512 VGPRs: 501 at K = 192, spills at 224)");
    constexpr int NG = K / 8;              // ds_read_b128 groups (4 MFMAs each) per half of the first product
    constexpr int ROWS_PER_WAVE = (TROWS + 3) / 4;  // LDS rows each wave moves per tile
    // LDS float offset of contraction index kq (a multiple of 4 or of 32, never straddling a block of KH) relative to the row of streamed index c
    auto kofs = [](int kq) constexpr -> int { return (kq / KH) * LDY + (kq % KH); };
    // second product: which k (inside its block of KH) MFMA lb of a block carries on A-row i -- see NMFX_G2_VEC
    constexpr int G2_KB = KH / 32, G2_NFULL = KH / 128, G2_WREM = G2_KB % 4;
    auto g2_kloc = [](int lb, int i) constexpr -> int {
        if (!NMFX_G2_VEC) return 32 * lb + i;
        return lb < 4 * G2_NFULL ? 128 * (lb / 4) + 4 * i + (lb % 4) : 128 * G2_NFULL + G2_WREM * i + (lb - 4 * G2_NFULL);
    };

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const long r_wg0 = (long)blockIdx.x * FT_ROWS;
    const long r0 = r_wg0 + 32 * w;   // this wave's rows r0 .. r0+31
    const long r = r0 + l31;
    const long cbeg = (long)blockIdx.y * p.c_per_split;
    const long cend = cbeg + p.c_per_split < p.Cn ? cbeg + p.c_per_split : p.Cn;
    const int ntiles = RAG ? (int)((cend - cbeg + FT_C - 1) / FT_C) : (int)((cend - cbeg) / FT_C);
    const bool row_ok = !RAG || r < p.R;

    // stationary operand: B-port register s holds X(r, k = 8*(s>>2) + 4*h + (s&3))
    float xreg[NEED_S ? K / 2 : 1];
    if (NEED_S) {
#pragma unroll
        for (int s = 0; s < K / 2; ++s) {
            const int kq = 8 * (s >> 2) + (s & 3);                     // + 4*h: stays inside a block of KH
            xreg[s] = row_ok ? p.X[r * p.xs_r + (long)(kq % KH + 4 * h) * p.xs_k + (long)(TT - 1 - kq / KH) * p.xs_t] : 0.0f;
        }
    }

    f32x16 acc[DO_G2 ? NKB : 1];
    f32x16 acc2[(DO_G2 && DUAL) ? NKB : 1];   // second contraction of the dual-map divergences (the denominators)
#pragma unroll
    for (int kb = 0; kb < (DO_G2 ? NKB : 1); ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[kb][e] = 0.0f;
#pragma unroll
    for (int kb = 0; kb < ((DO_G2 && DUAL) ? NKB : 1); ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[kb][e] = 0.0f;
    const float ab_e1 = p.ab_beta - 1.0f, ab_e2 = p.ab_alpha + p.ab_beta - 1.0f, ab_kappa = (p.ab_alpha + p.ab_beta) == 0.0f ? 0.0f : p.ab_beta / (p.ab_alpha + p.ab_beta);   // FUNC 5 (alpha + beta == 0: the caller reproduces the reference's division by zero)

    // ---- loads.  f32 MFMA shares the SIMD with VALU (every VALU instruction in the loop costs MFMA time), so all per-tile
    // addressing is wave-uniform (SGPR buffer descriptor + SGPR offset) plus ONE per-lane byte offset computed here.
    // Streamed tile: LDS-DMA (buffer_load_dwordx4 ... lds), one 1-KiB row of K floats per wave-instruction, wave w moves rows
    // w, w+4, ...  Issued as inline asm so hipcc does not see an LDS write (with the builtin it drains vmcnt(0) before the next
    // ds_read and serialises the DMA latency into every tile); completion = s_waitcnt vmcnt(0) + barrier at the tile top.
    const float *const Yz = p.Y + (TT == 1 ? (long)blockIdx.z * p.yz_stride : 0L);   // grid.z: K-wide column blocks of wider streamed rows (one launch for a Kw > 256 wide factor)
    const int ystride = p.y_stride > 0 ? (int)p.y_stride : K;   // floats between streamed rows in memory (> K: a K-wide column block of wider rows)
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)lds;
    const unsigned y_voff = (unsigned)lane * 16u;
    auto dma_row = [&](const i32x4 ysrd, int b, int c) {   // c-th row of this wave: tile row w + 4c
        const int row = w + 4 * c;
        if (TROWS % 4 != 0 && row >= TROWS) return;            // wave-uniform
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((b * BUF + row * LDY) * 4));
        const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)(row * (TT > 1 ? KH : ystride) * 4));
        unsigned keep;
        if (lane < KH / 4)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(y_voff), "s"(ysrd), "s"(dst), "s"(soff) : "memory");
    };
    // streamed rows left in tile t (FT_C except in the last, partial tile of a ragged extent): rows past them read as zero
    auto tile_rows = [&](int t) -> int {
        if (!RAG) return FT_C;
        const long left = cend - (cbeg + (long)t * FT_C);
        return left >= FT_C ? FT_C : (left > 0 ? (int)left : 0);
    };
    auto y_srd = [&](int t) {
        if (TT > 1) return make_srd(p.Y + (cbeg + (long)t * FT_C - (TT - 1)) * KH, (unsigned)((tile_rows(t) + TT - 1) * KH * 4));
        return make_srd(Yz + (cbeg + (long)t * FT_C) * ystride, (unsigned)(tile_rows(t) > 0 ? ((tile_rows(t) - 1) * ystride + K) * 4 : 0));
    };

    // V tile of step t: d[jb*16 + reg] = V(r, c = c0 + 32*jb + rowmap(reg, h))
    float d[32];
    const unsigned d_voff = D_RC ? (unsigned)((r + 4 * h * p.ldd) * 4) : (unsigned)((p.ldd * (r - r_wg0) + 4 * h) * 4);
    const long wg_rows = RAG ? (p.R - r_wg0 < FT_ROWS ? p.R - r_wg0 : (long)FT_ROWS) : (long)FT_ROWS;   // stationary rows this workgroup really has
    const __amdgpu_buffer_rsrc_t d_srd_fixed =
        __builtin_amdgcn_make_buffer_rsrc((void *)(p.D + p.ldd * r_wg0), 0, (int)(unsigned)(wg_rows * p.ldd * 4), 0x00020000);   // !D_RC
    auto d_srd = [&](int t) {
        if (D_RC) return __builtin_amdgcn_make_buffer_rsrc((void *)(p.D + p.ldd * (cbeg + (long)t * FT_C)), 0, (int)(unsigned)(tile_rows(t) * p.ldd * 4), 0x00020000);
        return d_srd_fixed;
    };
    // piece i of 16: D_RC two dwords (columns c0 + 32*jb + {(reg&3) + 8*(reg>>2)}), else half of the 8 float4 (one per even i)
    auto load_d_piece = [&](const __amdgpu_buffer_rsrc_t srd, int t, int i) {
        if (FUNC == 7 || NO_V) return;
        if (D_RC) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int e = 2 * i + u, jb = e >> 4, reg = e & 15;
                const int soff = (int)(p.ldd * (32 * jb + (reg & 3) + 8 * (reg >> 2)) * 4);
                d[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd, d_voff, soff, 0));
            }
        } else if ((i & 1) == 0) {
            const int f = i >> 1, jb = f >> 2, q = f & 3;
            const int soff = (int)((cbeg + (long)t * FT_C + 32 * jb + 8 * q) * 4);
            if (!RAG) {   // tile-aligned extents: the four floats are one 16-byte load (8 VMEM instructions a tile instead of 32)
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                // (the WHOLE vector is cast: a bit_cast applied to one element of an ext_vector value reads element 0 for every index -- hipcc 7.2, -O3; the same trap
                // as in the R store below)
                const f32x4 q4 = __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(srd, d_voff, soff, 0));
                d[jb * 16 + 4 * q + 0] = q4.x; d[jb * 16 + 4 * q + 1] = q4.y; d[jb * 16 + 4 * q + 2] = q4.z; d[jb * 16 + 4 * q + 3] = q4.w;
                return;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                d[jb * 16 + 4 * q + e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd, d_voff, soff + 4 * e, 0));
        }
    };

    // partial S of the same tile (S_IN), same register layout as d[]: a descriptor of zero bytes (no p.Sin) reads zeros, so the first launch
    // of a chain runs the same instruction stream
    float sin_[S_IN ? 32 : 1];
    float sin_n[(S_IN && NMFX_G1_ASM) ? 32 : 1];   // SADD: the next tile's partial S (sin_ stays live until it has been added at the end of P2)
    auto s_srd = [&](int t) {
        return __builtin_amdgcn_make_buffer_rsrc((void *)(p.Sin ? p.Sin + p.ldd * (cbeg + (long)t * FT_C) : p.D), 0, p.Sin ? (int)(unsigned)(tile_rows(t) * p.ldd * 4) : 0, 0x00020000);
    };
    auto load_s_piece = [&](const __amdgpu_buffer_rsrc_t srd, int i, bool nxt = false) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = 2 * i + u, jb = e >> 4, reg = e & 15;
            const float sv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(srd, d_voff, (int)(p.ldd * (32 * jb + (reg & 3) + 8 * (reg >> 2)) * 4), 0));
            if (S_IN && NMFX_G1_ASM && nxt) sin_n[(S_IN && NMFX_G1_ASM) ? e : 0] = sv;
            else sin_[S_IN ? e : 0] = sv;
        }
    };

    double cost = 0.0;
    if (ntiles > 0) {
        const i32x4 ys = y_srd(0);
#pragma unroll
        for (int c = 0; c < ROWS_PER_WAVE; ++c) dma_row(ys, 0, c);
        const __amdgpu_buffer_rsrc_t ds0 = d_srd(0);
#pragma unroll
        for (int i = 0; i < 16; ++i) load_d_piece(ds0, 0, i);
        if (S_IN) {
            const __amdgpu_buffer_rsrc_t ss0 = s_srd(0);
#pragma unroll
            for (int i = 0; i < 16; ++i) load_s_piece(ss0, i);
        }
    }
    for (int t = 0; t < ntiles; ++t) {
        const int b = t & 1;
        const int tn = t + 1 < ntiles ? t + 1 : t;           // the last tile re-fetches itself: keeps the body branch-free
        {
            // own DMA rows of tile t have landed.  With a first product the 32 V loads of this tile were issued AFTER its DMA
            // rows (loads return in order), and V is first needed in P2, a whole P1 (>= 3 us of MFMAs) later: leave them in
            // flight instead of draining to vmcnt(0), which exposed one HBM round trip (~2 us of a 13.6 us tile at K = 256) per tile.
            // hipcc places its own, conservative vmcnt waits before the first use of d[] (it does not count the asm DMA loads).
            // EARLY: the DMA rows of tile t > 0 were waited for behind P2 of tile t-1 (see there)
            if (EARLY || STB) { if (t == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            else if (NEED_S && !NO_V) {
                // (NVL = the V-load INSTRUCTIONS of a tile: 32 dword loads, or 8 dwordx4 in the H-step form on tile-aligned extents -- the count this wait hangs on)
                if (NVL == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                     // everyone's rows landed; buffer b^1 is free again
        }
        const float *Yt = lds + b * BUF;
        const i32x4 ysn = y_srd(tn);
        const __amdgpu_buffer_rsrc_t dsn = d_srd(tn);
        const __amdgpu_buffer_rsrc_t ssn = S_IN ? s_srd(tn) : dsn;
        int dma_c = 0;                                        // rows of tile tn issued so far (compile-time after unrolling)
        auto dma_some = [&](int upto) {                       // issue rows until `upto` have been issued
#pragma unroll
            for (int c = 0; c < ROWS_PER_WAVE; ++c)
                if (c >= dma_c && c < upto) dma_row(ysn, b ^ 1, c);
            dma_c = upto > dma_c ? upto : dma_c;
        };

        // ---- software-interleaved tile body -------------------------------------------------------------------------
        // P1  first product, half 0     || LDS-DMA of the next tile, one row every other group
        // P2  first product, half 1     || element map of half 0
        // P3  second product, half 0    || element map of half 1
        // P4  second product, half 1    || V loads of the next tile, two per step
        // A wave issues in order and a 32x32x2 f32 MFMA occupies the pipe for 64 cycles, so whatever sits between two MFMAs
        // in program order must finish (dependency latencies included) inside 64 cycles or the matrix pipe idles.  The
        // element maps are therefore cut into micro-ops placed behind chosen MFMAs (the KL map: bursts of 4-6 instructions that fit that shadow, emap_burst; the
        // other divergences: one micro-op at a time, emap_u), LDS operands are fetched one step ahead, and a sched_barrier after every MFMA pins this order (left
        // alone, hipcc clusters the VALU/VMEM work: probe runs lost 9-19 % of the MFMA rate that way).
        f32x16 sacc[2];
        f32x16 sacc2[DUAL ? 2 : 1];                           // the second map's tile (B)
        float bq[BQ ? 2 : 1][BQ ? 16 : 1];                    // the second map's values of this tile, on their way to p.Rout (ST2: p.Rout2)
        const __amdgpu_buffer_rsrc_t rs_b = STB ? __builtin_amdgcn_make_buffer_rsrc((void *)(p.Rout + p.ldd * (cbeg + (long)t * FT_C)), 0, (int)(unsigned)(tile_rows(t) * p.ldd * 4), 0x00020000)
                                                : d_srd_fixed;
        float tc = 0.0f;
        float ts = 0.0f;                                      // KL_MODE 1: sum of S - q.*S (natural units; tc is in log2 units)
        f32x2 tc2[2] = {{0.0f, 0.0f}, {0.0f, 0.0f}}, ts2[2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};   // KL_MODE 2: the same two sums per pair of a double pair, even / odd elements
        f32x2 es2[2], er2[2], eq2[2];                         // ... and the map's state for the two pairs of the double pair in flight
        const int cvh = RAG ? tile_rows(t) - 4 * h : 0;       // streamed index 32*jb + (reg&3) + 8*(reg>>2) + 4*h of this tile is real iff its h-free part < cvh
        float es[2], er[2], eq[2];                            // element-map pipeline state: S value, reciprocal / quotient, third temporary
        auto emap_u = [&](int jb, int reg, int u) {           // micro-op u of element (jb, reg); R + divergence terms, nmf.m:152,206-215
            const int sl = reg & 1;
            if (FUNC == 7) return;                            // the raw partial S is what gets stored
            const float v = NO_V ? 0.0f : d[jb * 16 + reg];
            const bool live = !RAG || (32 * jb + (reg & 3) + 8 * (reg >> 2)) < cvh;   // streamed index inside the matrix
            if (FUNC == 4) {                                  // IS: B = 1./S, A = V./S.^2, cost terms q - ln(q) with q = V./S (the -1 per element: caller)
                if (u == 0) { es[sl] = sacc[jb][reg]; er[sl] = __builtin_amdgcn_rcpf(es[sl]); }
                if (u == 1) eq[sl] = v * er[sl];
                if (u == 2) sacc2[jb][reg] = live ? er[sl] : 0.0f;
                if (u == 3) sacc[jb][reg] = live ? eq[sl] * er[sl] : 0.0f;
                if (u == 4) er[sl] = __builtin_amdgcn_logf(eq[sl]);                  // log2(q)
                if (u == 5) tc = live ? tc + eq[sl] : tc;
                if (u == 6) tc = live ? fmaf(-0.6931471805599453f, er[sl], tc) : tc;
                if (u == 5 || u == 6) asm volatile("" : "+v"(tc));
            } else if (FUNC == 5) {                           // alpha-beta: B = S.^(a+b-1), A = V.^a .* S.^(b-1); cost terms S.*(A - b/(a+b)*B)
                // x.^0 == 1 also for x == 0 and x == Inf (MATLAB): log2(S) is clamped to +-3e38 (one v_med3_f32), so an exponent of exactly 0
                // (beta == 1, or alpha + beta == 1) gives 2^(+-0) = 1 where 0 * (-Inf) would be NaN; any other exponent still overflows to the same 0 / Inf
                if (u == 0) { es[sl] = sacc[jb][reg]; er[sl] = __builtin_amdgcn_fmed3f(__builtin_amdgcn_logf(es[sl]), -3.0e38f, 3.0e38f); }   // log2(S)
                if (u == 1) eq[sl] = ab_e1 * er[sl];
                if (u == 2) eq[sl] = __builtin_amdgcn_exp2f(eq[sl]);                 // S.^(b-1)
                if (u == 3) er[sl] = __builtin_amdgcn_exp2f(ab_e2 * er[sl]);         // S.^(a+b-1)
                if (u == 4) { eq[sl] = v * eq[sl]; sacc[jb][reg] = live ? eq[sl] : 0.0f; }
                if (u == 5) sacc2[jb][reg] = live ? er[sl] : 0.0f;
                if (u == 6) eq[sl] = fmaf(-ab_kappa, er[sl], eq[sl]);
                if (u == 7) { tc = live ? fmaf(es[sl], eq[sl], tc) : tc; asm volatile("" : "+v"(tc)); }
            } else if (EF == 11) {                            // IS, numerators only: A = V./S.^2, cost terms q - ln(q) (functor 4 without its B tile)
                if (u == 0) { es[sl] = sacc[jb][reg]; er[sl] = __builtin_amdgcn_rcpf(es[sl]); }
                if (u == 1) eq[sl] = v * er[sl];
                if (BQ && u == 2) bq[BQ ? jb : 0][BQ ? reg : 0] = er[sl];           // B = 1./S
                if (u == 3) sacc[jb][reg] = live ? eq[sl] * er[sl] : 0.0f;
                if (u == 4) er[sl] = __builtin_amdgcn_logf(eq[sl]);                  // log2(q)
                if (u == 5) tc = live ? tc + eq[sl] : tc;
                if (u == 6) tc = live ? fmaf(-0.6931471805599453f, er[sl], tc) : tc;
                if (u == 5 || u == 6) asm volatile("" : "+v"(tc));
            } else if (FUNC == 12) {                          // IS, denominators only: B = 1./S
                if (u == 0) sacc[jb][reg] = live ? __builtin_amdgcn_rcpf(sacc[jb][reg]) : 0.0f;
            } else if (EF == 13) {                            // alpha-beta, numerators only: A = V.^a .* S.^(b-1), cost terms S.*(A - b/(a+b)*B) (functor 5 without its B tile)
                if (u == 0) { es[sl] = sacc[jb][reg]; er[sl] = __builtin_amdgcn_fmed3f(__builtin_amdgcn_logf(es[sl]), -3.0e38f, 3.0e38f); }   // log2(S), see functor 5
                if (u == 1) eq[sl] = ab_e1 * er[sl];
                if (u == 2) eq[sl] = __builtin_amdgcn_exp2f(eq[sl]);                 // S.^(b-1)
                if (u == 3) er[sl] = __builtin_amdgcn_exp2f(ab_e2 * er[sl]);         // S.^(a+b-1)
                if (u == 4) { eq[sl] = v * eq[sl]; sacc[jb][reg] = live ? eq[sl] : 0.0f; }
                if (BQ && u == 5) bq[BQ ? jb : 0][BQ ? reg : 0] = er[sl];           // B = S.^(a+b-1)
                if (u == 6) eq[sl] = fmaf(-ab_kappa, er[sl], eq[sl]);
                if (u == 7) { tc = live ? fmaf(es[sl], eq[sl], tc) : tc; asm volatile("" : "+v"(tc)); }
            } else if (FUNC == 17) {                          // alpha-beta, dual form (alpha == 0): A = S.^beta ./ V
                if (u == 0) er[sl] = __builtin_amdgcn_fmed3f(__builtin_amdgcn_logf(sacc[jb][reg]), -3.0e38f, 3.0e38f);   // log2(S), clamped as in functor 5
                if (u == 1) eq[sl] = __builtin_amdgcn_exp2f(p.ab_beta * er[sl]);                                         // S.^beta
                if (u == 2) er[sl] = __builtin_amdgcn_rcpf(v);
                if (u == 3) sacc[jb][reg] = live ? eq[sl] * er[sl] : 0.0f;
            } else if (FUNC == 14) {                          // alpha-beta, denominators only: B = S.^(a+b-1)
                if (u == 0) er[sl] = __builtin_amdgcn_fmed3f(__builtin_amdgcn_logf(sacc[jb][reg]), -3.0e38f, 3.0e38f);
                if (u == 1) sacc[jb][reg] = live ? __builtin_amdgcn_exp2f(ab_e2 * er[sl]) : 0.0f;
            } else if (FUNC == 6) {                           // residual: R = S - V, cost terms (S - V).^2   (nmfsc.m:139,148)
                if (u == 0) { const float e = sacc[jb][reg] - v; tc = live ? fmaf(e, e, tc) : tc; sacc[jb][reg] = live ? e : 0.0f; }
            } else if (MF >= 2) {
                if (u == 0) { es[sl] = sacc[jb][reg]; er[sl] = __builtin_amdgcn_rcpf(es[sl]); }
                if (u == 1) { er[sl] = v * er[sl]; sacc[jb][reg] = live ? er[sl] : 0.0f; }      // q = V ./ V_hat
                if (MF == 3 && NMFX_KL_MODE == 0) {
                    if (u == 2) er[sl] = __builtin_amdgcn_logf(er[sl]);   // log2(q)
                    if (u == 3) tc = live ? fmaf(v, er[sl], tc) : tc;   // sum(V_hat - V) is added in closed form by the caller (see FusedParams)
                }
                if (MF == 3 && NMFX_KL_MODE == 1) {
                    if (u == 2) eq[sl] = fmaf(-er[sl], es[sl], es[sl]);   // S - q.*S
                    if (u == 3) er[sl] = __builtin_amdgcn_logf(er[sl]);   // log2(q)
                    if (u == 4) { tc = live ? fmaf(v, er[sl], tc) : tc; asm volatile("" : "+v"(tc)); }
                    if (u == 5) { ts = live ? ts + eq[sl] : ts; asm volatile("" : "+v"(ts)); }
                }
            } else {
                if (u == 0) {
                    if (MF == 1) {   // nmf.m:208
                        const float e = sacc[jb][reg] - v;
                        tc = live ? fmaf(e, e, tc) : tc;
                        if (FUNC == 21) sacc[jb][reg] = e;   // the residual S - V is what this pass leaves in HBM (dead positions: never stored -- row_ok / the store's buffer bounds)
                    }
                    if (DO_G2 || MF != 1) sacc[jb][reg] = live ? v : 0.0f;   // (cost-only form: S stays, for the optional store below)
                }
            }
            if (((MF == 1 || FUNC == 6) && u == 0) || (MF == 3 && NMFX_KL_MODE == 0 && u == 3)) asm volatile("" : "+v"(tc));   // keep the cost terms in place
        };
        // The packed KL map (PK) in BURSTS.  What an element-map instruction costs the MFMA stream is the interruption, not the instruction: on the skeleton of this
        // tile (scripts/ubench_mfma.hip, profiles/r6_ubench.md) 128 v_fma_f32 spread one by one take 8.7 points of the MFMA rate, the same 128 in groups of eight 2.8;
        // 64 v_rcp_f32 3.1 points one by one, 2.6 in groups of four -- as long as a group fits the 64-cycle shadow of the MFMA in front of it (a transcendental is 16
        // cycles, a packed or plain fp32 instruction 4).  So the map of a half-tile runs per DOUBLE PAIR (4 elements) in three bursts of 48-64 cycles:
        //   b = 0   4 x v_rcp                                              b = 1   2 x v_pk_mul (q), 2 x v_pk_fma (S - q.*S), 2 x v_log
        //   b = 2   2 x v_log, 2 x v_pk_fma (V.*log q), 2 x v_pk_add        (functor 2, no cost terms: b = 0 and the two v_pk_mul)
        // twelve interruptions per half-tile instead of 64.  sacc[jb][reg .. reg + 3] and d[.. + reg ..] are adjacent registers: the pairs cost no moves.
        constexpr int NBD = MF == 3 ? 3 : 2;                  // bursts per double pair
        auto emap_burst = [&](int jb, int dp, int b) {
            const int reg = 4 * dp;
            const bool lv[4] = {!RAG || (32 * jb + (reg & 3) + 8 * (reg >> 2)) < cvh, !RAG || (32 * jb + ((reg + 1) & 3) + 8 * ((reg + 1) >> 2)) < cvh,
                                !RAG || (32 * jb + ((reg + 2) & 3) + 8 * ((reg + 2) >> 2)) < cvh, !RAG || (32 * jb + ((reg + 3) & 3) + 8 * ((reg + 3) >> 2)) < cvh};
            const f32x2 vv[2] = {{NO_V ? 0.0f : d[jb * 16 + reg], NO_V ? 0.0f : d[jb * 16 + reg + 1]}, {NO_V ? 0.0f : d[jb * 16 + reg + 2], NO_V ? 0.0f : d[jb * 16 + reg + 3]}};
            if (b == 0) {
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    es2[pr].x = sacc[jb][reg + 2 * pr]; es2[pr].y = sacc[jb][reg + 2 * pr + 1];
                    er2[pr].x = __builtin_amdgcn_rcpf(es2[pr].x); er2[pr].y = __builtin_amdgcn_rcpf(es2[pr].y);
                }
            }
            if (b == 1) {
                // q = V ./ V_hat.  (Always the `_t` forms behind a transcendental, however far back it was issued in the source: with few MFMAs per phase (K <= 64), in
                // the cost-only form, or when hipcc sinks a v_rcp to its use, producer and consumer end up adjacent -- round 6's first burst version returned NaN there)
                er2[0] = pk_mul_t(vv[0], er2[0]);
                er2[1] = pk_mul_t(vv[1], er2[1]);
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    sacc[jb][reg + 2 * pr] = lv[2 * pr] ? er2[pr].x : 0.0f;
                    sacc[jb][reg + 2 * pr + 1] = lv[2 * pr + 1] ? er2[pr].y : 0.0f;
                }
                if (MF == 3) {
                    eq2[0] = pk_fnma(er2[0], es2[0], es2[0]);                                 // S - q.*S  (= S - V up to the rounding of q, see NMFX_KL_MODE)
                    eq2[1] = pk_fnma(er2[1], es2[1], es2[1]);
                    er2[0].x = __builtin_amdgcn_logf(er2[0].x); er2[0].y = __builtin_amdgcn_logf(er2[0].y);   // log2(q), first pair
                }
            }
            if (b == 2 && MF == 3) {
                er2[1].x = __builtin_amdgcn_logf(er2[1].x); er2[1].y = __builtin_amdgcn_logf(er2[1].y);
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const f32x2 t = pk_fma_t(vv[pr], er2[pr], tc2[pr]);
                    const f32x2 u = pk_add(ts2[pr], eq2[pr]);
                    if (RAG) {
                        tc2[pr].x = lv[2 * pr] ? t.x : tc2[pr].x; tc2[pr].y = lv[2 * pr + 1] ? t.y : tc2[pr].y;
                        ts2[pr].x = lv[2 * pr] ? u.x : ts2[pr].x; ts2[pr].y = lv[2 * pr + 1] ? u.y : ts2[pr].y;
                    } else { tc2[pr] = t; ts2[pr] = u; }
                    asm volatile("" : "+v"(tc2[pr]), "+v"(ts2[pr]));
                }
            }
        };
        // fillers behind the i-th MFMA of a phase with M MFMAs that hosts the 16 elements x NU micro-ops of half jb: slot i runs
        // micro-ops [16*NU*i/M, 16*NU*(i+1)/M) in element order (NU = 4: one every other MFMA at K = 256, one each at 128, two at 64, four at 32)
        auto emap_fill = [&](int jb, int i, int M) {
            if (PK) {   // 4 double pairs x NBD bursts: burst q sits behind MFMA (2q + 1) * M / (2 * 4 * NBD) of the phase
#pragma unroll
                for (int q = 0; q < 4 * NBD; ++q)
                    if ((2 * q + 1) * M / (8 * NBD) == i) emap_burst(jb, q / NBD, q % NBD);
                return;
            }
            const int q0 = 16 * NU * i / M, q1 = 16 * NU * (i + 1) / M;
#pragma unroll
            for (int q = q0; q < q1; ++q) emap_u(jb, q / NU, q % NU);
        };
        auto g1_read = [&](int jb, int g) { return *reinterpret_cast<const float4 *>(Yt + (32 * jb + l31) * LDY + kofs(8 * g) + 4 * h); };
        if (NEED_S) {
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int e = 0; e < 16; ++e) sacc[jb][e] = (S_IN && !SADD) ? sin_[S_IN ? jb * 16 + e : 0] : 0.0f;   // (asm form: dead -- the chains start with C = 0)
            float4 a_cur = g1_read(0, 0);
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {                          // P1 (ph 0), P2 (ph 1)
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    float4 a_nxt = a_cur;
                    if (g + 1 < NG) a_nxt = g1_read(ph, g + 1);
                    else if (ph == 0) a_nxt = g1_read(1, 0);
                    auto one = [&](int e, float av) {
#if NMFX_G1_ASM
                      if (G1A) {
                        // asm: the S tile in architectural VGPRs (the element map reads it: out of AGPRs that costs one v_accvgpr_read per element) and the
                        // stationary operand in AGPRs (it is only ever an MFMA source; in v0-v127 it left too few VGPRs for the tiles and hipcc shuttled the V tile
                        // through AGPRs).  hipcc itself puts EVERY MFMA result of a kernel that may use more than 256 registers into AGPRs.  Back-to-back MFMAs on
                        // one accumulator need no wait states; what reads the tile afterwards is kept away from the last MFMA by mfma_settle() below
                        // (the dual-map kernels at K > 128 hold two output accumulator sets: with the stationary operand as well the AGPR half would overflow)
                        constexpr bool XA = K / 2 + (DO_G2 ? (DUAL ? 2 : 1) * NKB * 16 : 0) <= 256;
                        const bool first = g == 0 && e == 0 && (!S_IN || SADD);   // chain start: C = 0
                        if (ph == 1 && g == 0 && e == 0) {
                            // the first MFMA of the second chain, and behind it -- in the SAME statement -- what the first chain still needs: that MFMA's 16 passes +
                            // `s_nop 3` after the last MFMA of sacc[0], which rides along as an in/out operand (see mfma_settle_full)
                            if (XA) {
                                if (first) asm volatile("v_mfma_f32_32x32x2_f32 %0, %2, %3, 0\n\ts_nop 3" : "=&v"(sacc[1]), "+v"(sacc[0]) : "v"(av), "a"(xreg[0]));
                                else asm volatile("v_mfma_f32_32x32x2_f32 %0, %2, %3, %0\n\ts_nop 3" : "+v"(sacc[1]), "+v"(sacc[0]) : "v"(av), "a"(xreg[0]));
                            } else {
                                if (first) asm volatile("v_mfma_f32_32x32x2_f32 %0, %2, %3, 0\n\ts_nop 3" : "=&v"(sacc[1]), "+v"(sacc[0]) : "v"(av), "v"(xreg[0]));
                                else asm volatile("v_mfma_f32_32x32x2_f32 %0, %2, %3, %0\n\ts_nop 3" : "+v"(sacc[1]), "+v"(sacc[0]) : "v"(av), "v"(xreg[0]));
                            }
                            if (SADD) {   // the partial sums of the blocks before, half 0
#pragma unroll
                                for (int q = 0; q < 16; ++q) sacc[0][q] += sin_[S_IN ? q : 0];
                            }
                        } else if (XA) {
                            if (first) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=&v"(sacc[ph]) : "v"(av), "a"(xreg[4 * g + e]));
                            else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(sacc[ph]) : "v"(av), "a"(xreg[4 * g + e]));
                        } else {
                            if (first) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=&v"(sacc[ph]) : "v"(av), "v"(xreg[4 * g + e]));
                            else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(sacc[ph]) : "v"(av), "v"(xreg[4 * g + e]));
                        }
                      } else
#endif
                        sacc[ph] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, xreg[4 * g + e], sacc[ph], 0, 0, 0);
                        if (ph == 1) emap_fill(0, 4 * g + e, 4 * NG);
                        if (ph == 0 && e == 0) dma_some(((g + 1) * ROWS_PER_WAVE + NG - 1) / NG);
                        if (S_IN && ph == 1 && 4 * g + e < 16) load_s_piece(ssn, 4 * g + e, true);   // partial S of the next tile (builtin form: sin_ went into sacc at the tile top; asm form: into sin_n)
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    one(0, a_cur.x); one(1, a_cur.y); one(2, a_cur.z); one(3, a_cur.w);
                    a_cur = a_nxt;
                }
            }
        } else {
            // no first product: R = V, needed at once.  Both halves move into the R tile here, so d[] is free for the whole tile and the V loads of the
            // next tile go out during P3 -- a tile ahead of their use instead of half a tile (at K = 128 that half is 1.7 us, less than an HBM round trip
            // under load, and the wait at the next tile top exposed the rest)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) emap_u(0, reg, 0);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) emap_u(1, reg, 0);
        }
#if NMFX_G1_ASM
        if (NEED_S && G1A && !DO_G2) {   // cost-only form: nothing follows the second chain (with a second product, its first MFMA carries the wait: below)
            mfma_settle_full(sacc[1]);
            if (SADD) {
#pragma unroll
                for (int q = 0; q < 16; ++q) sacc[1][q] += sin_[S_IN ? 16 + q : 0];
            }
        }
#endif
        if (DO_G2) {
            if (STB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next tile's DMA rows (issued under P1) have landed; from here on stores are in flight too
            auto g2_read = [&](int jb, int reg, float (&y)[NKB]) {
                if (NMFX_G2_VEC) {   // vector reads: MFMA kb = pb*KB + lb carries k = pb*KH + g2_kloc(lb, lane) -- 4 consecutive floats per 128-float chunk and lane
                    const float *yrow = Yt + (32 * jb + rowmap(reg, h)) * LDY;
#pragma unroll
                    for (int pb = 0; pb < TT; ++pb) {
#pragma unroll
                        for (int c = 0; c < G2_NFULL; ++c) {
                            const float4 t4 = *reinterpret_cast<const float4 *>(yrow + pb * LDY + 128 * c + 4 * l31);
                            y[pb * G2_KB + 4 * c + 0] = t4.x; y[pb * G2_KB + 4 * c + 1] = t4.y; y[pb * G2_KB + 4 * c + 2] = t4.z; y[pb * G2_KB + 4 * c + 3] = t4.w;
                        }
                        if (G2_WREM == 2) {
                            const float2 t2 = *reinterpret_cast<const float2 *>(yrow + pb * LDY + 128 * G2_NFULL + 2 * l31);
                            y[pb * G2_KB + 4 * G2_NFULL + 0] = t2.x; y[pb * G2_KB + 4 * G2_NFULL + 1] = t2.y;
                        } else {
#pragma unroll
                            for (int j = 0; j < G2_WREM; ++j) y[pb * G2_KB + 4 * G2_NFULL + j] = yrow[pb * LDY + 128 * G2_NFULL + G2_WREM * l31 + j];
                        }
                    }
                    return;
                }
                const float *yrow = Yt + (32 * jb + rowmap(reg, h)) * LDY + l31;
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) y[kb] = yrow[kofs(32 * kb)];
            };
            float y_cur[NKB], y_nxt[NKB];
            g2_read(0, 0, y_cur);
#pragma unroll
            for (int st = 0; st < 32; ++st) {                        // P3 (st < 16) and P4
                const int jb = st >> 4, reg = st & 15;
                if (st + 1 < 32) g2_read((st + 1) >> 4, (st + 1) & 15, y_nxt);
                const float rr = sacc[jb][reg];
                const float rr2 = DUAL ? sacc2[jb][reg] : 0.0f;
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) {
#if NMFX_G1_ASM
                    if (NEED_S && G1A && st == 0 && kb == 0) {
                        // the first MFMA of the second product carries the wait for the second chain (sacc[1], in/out): its 16 passes + `s_nop 3` (see mfma_settle_full)
                        if (SWAP) asm volatile("v_mfma_f32_32x32x2_f32 %0, %2, %3, %0\n\ts_nop 3" : "+a"(acc[0]), "+v"(sacc[1]) : "v"(rr), "v"(y_cur[0]));
                        else asm volatile("v_mfma_f32_32x32x2_f32 %0, %2, %3, %0\n\ts_nop 3" : "+a"(acc[0]), "+v"(sacc[1]) : "v"(y_cur[0]), "v"(rr));
                    } else
#endif
                    acc[kb] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x2f32(rr, y_cur[kb], acc[kb], 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x2f32(y_cur[kb], rr, acc[kb], 0, 0, 0);
                    if (jb == 0 && NEED_S) emap_fill(1, (reg * NKB + kb) * (DUAL ? 2 : 1), 16 * NKB * (DUAL ? 2 : 1));   // element map of half 1 under the MFMAs of half 0
                    if (kb == 0 && !NEED_S) dma_some((st + 1) * ROWS_PER_WAVE / 24 < ROWS_PER_WAVE ? (st + 1) * ROWS_PER_WAVE / 24 : ROWS_PER_WAVE);   // no first product: the DMA rides here, done by step 24
                    if (kb == NKB / 2 && jb == (NEED_S ? 1 : 0)) load_d_piece(dsn, tn, reg);   // V tile of the next step, in flight under P4 (no first product: under P3 already)
                    if (STB && kb == 1 && (jb == 1 || reg >= 8)) {   // the B values: half 0's (mapped under P2) two per step in the second half of P3, half 1's (mapped under P3) one per step of P4
                        auto put = [&](int j2, int r2) {
                            const float bv = bq[STB ? j2 : 0][STB ? r2 : 0];
                            if (row_ok) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, bv), rs_b, d_voff, (int)(p.ldd * (32 * j2 + (r2 & 3) + 8 * (r2 >> 2)) * 4), 0);
                        };
                        if (jb == 0) { put(0, 2 * (reg - 8)); put(0, 2 * (reg - 8) + 1); }
                        else put(1, reg);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (DUAL) {
                        acc2[kb] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x2f32(rr2, y_cur[kb], acc2[kb], 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x2f32(y_cur[kb], rr2, acc2[kb], 0, 0, 0);
                        if (jb == 0) emap_fill(1, (reg * NKB + kb) * 2 + 1, 32 * NKB);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) y_cur[kb] = y_nxt[kb];
            }
        } else {
            // S_IN: the next tile's DMA rows went out during P1, only the 32 partial-S loads of P2 are younger: the rows have landed once at most those
            // are in flight (the R stores and V loads below would push the count past what s_waitcnt can express at the tile top)
            if (S_IN) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            else if (EARLY) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (PK) {
#pragma unroll
                for (int q = 0; q < 4 * NBD; ++q) emap_burst(1, q / NBD, q % NBD);
            } else {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg)
#pragma unroll
                    for (int u = 0; u < NU; ++u) emap_u(1, reg, u);
            }
            if (D_RC && ((MF >= 2 && MF <= 3) || FUNC == 7 || FUNC == 1 || FUNC == 21 || ST2) && p.Rout) {   // wave-uniform: this pass also leaves R = V./S in HBM (KL cnmf: the numerator passes read it)
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(p.Rout + p.ldd * (cbeg + (long)t * FT_C)), 0, (int)(unsigned)(tile_rows(t) * p.ldd * 4), 0x00020000);
                if (row_ok) {
#pragma unroll
                    for (int e = 0; e < 32; ++e) {
                        const int jb = e >> 4, reg = e & 15;
                        const float rv = sacc[jb][reg];   // (a bit_cast applied to the vector element itself reads element 0)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, rv), rs, d_voff, (int)(p.ldd * (32 * jb + (reg & 3) + 8 * (reg >> 2)) * 4), 0);
                    }
                }
            }
            if (ST2 && p.Rout2) {   // wave-uniform: the second map's values of this tile (masked like the first: a streamed index past the end stores nothing real)
                const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void *)(p.Rout2 + p.ldd * (cbeg + (long)t * FT_C)), 0, (int)(unsigned)(tile_rows(t) * p.ldd * 4), 0x00020000);
                if (row_ok) {
#pragma unroll
                    for (int e = 0; e < 32; ++e) {
                        const int jb = e >> 4, reg = e & 15;
                        const float bv = bq[BQ ? jb : 0][BQ ? reg : 0];
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, bv), rs2, d_voff, (int)(p.ldd * (32 * jb + (reg & 3) + 8 * (reg >> 2)) * 4), 0);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) load_d_piece(dsn, tn, i);
        }
        dma_some(ROWS_PER_WAVE);
        if (SADD) {
#pragma unroll
            for (int e = 0; e < 32; ++e) sin_[S_IN ? e : 0] = sin_n[(S_IN && NMFX_G1_ASM) ? e : 0];
        }
        // rows past R hold garbage (possibly NaN): theirs alone, never summed.  KL (modes 1 / 2): tc is in log2 units, ts in natural ones
        if (MF == 3 && NMFX_KL_MODE == 2) cost += row_ok ? ((double)tc2[0].x + (double)tc2[0].y + (double)tc2[1].x + (double)tc2[1].y) * 0.6931471805599453 + ((double)ts2[0].x + (double)ts2[0].y + (double)ts2[1].x + (double)ts2[1].y) : 0.0;
        else if (MF == 3 && NMFX_KL_MODE == 1) cost += row_ok ? (double)tc * 0.6931471805599453 + (double)ts : 0.0;
        else cost += row_ok ? (double)tc : 0.0;
    }

    // epilogue: acc[kb][reg] = O(k, r) with k = (kb / G2_KB)*KH + g2_kloc(kb % G2_KB, i), where i = rowmap(reg, h) and r = this lane's row -- or, SWAP, i = the lane and
    // r = r0 + rowmap(reg, h)
    auto e_kloc = [&](int kb, int reg) -> int { return g2_kloc(kb % G2_KB, SWAP ? l31 : rowmap(reg, h)); };
    auto e_row = [&](int reg) -> long { return SWAP ? r0 + rowmap(reg, h) : r; };
    auto e_ok = [&](int reg) -> bool { return !RAG || e_row(reg) < p.R; };
    if (DO_G2) {
        if (EPI == 0 || EPI == 2) {
            float *out = p.out + (long)blockIdx.y * p.slab_stride + (TT == 1 ? (long)blockIdx.z * p.oz_stride : 0L);
#pragma unroll
            for (int q = 0; q < 16 * NKB; ++q) {
                const int kb = SWAP ? q % NKB : q / 16, reg = SWAP ? q / NKB : q % 16;   // SWAP: the MFMAs innermost -- a lane's consecutive k
                if (e_ok(reg)) {
                    const long oi = e_row(reg) * p.os_r + (long)e_kloc(kb, reg) * p.os_k + (long)(TT - 1 - kb / G2_KB) * p.os_t;
                    out[oi] = acc[kb][reg];
                    if (DUAL) p.out2[(long)blockIdx.y * p.slab_stride + oi] = acc2[kb][reg];
                }
            }
        } else {
            // H(k, j=r) <- H .* (G ./ max(den + lambda, eps))      nmf.m:199   (den: matrix K x n, or per-row vector for KL)
            // (EPI 1 is the H-step form: SWAP whenever NMFX_G2_VEC.)  The common case -- float64 master, no fixed rows, the ratio rule -- in vectors: the four MFMAs of
            // chunk c hold k = 128*c + 4*lane + {0, 1, 2, 3} of column r(reg): 32 bytes of the master and 16 of H per lane, 1 KB / 512 B contiguous per half-wave
            bool lines_done = false;
            if (SWAP && !DUAL && G2_NFULL > 0 && p.H64 && !p.fix && !p.sqrt_rule) {   // (wave-uniform)
                lines_done = true;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    if (!e_ok(reg)) continue;
#pragma unroll
                    for (int c = 0; c < G2_NFULL; ++c) {
                        const int k0 = 128 * c + 4 * l31;
                        const long idx0 = (long)k0 + (long)K * e_row(reg);
                        double hv[4], dn[4];
#pragma unroll
                        for (int o = 0; o < 4; ++o) hv[o] = p.H64[idx0 + o];
#pragma unroll
                        for (int o = 0; o < 4; ++o) dn[o] = (p.den ? (double)p.den[idx0 + o] : p.denvec[k0 + o]) + (p.lam ? (double)p.lam[k0 + o] : 0.0);
#pragma unroll
                        for (int o = 0; o < 4; ++o) hv[o] = hv[o] * ((double)acc[4 * c + o][reg] / fmax(dn[o], 2.220446049250313e-16));
#pragma unroll
                        for (int o = 0; o < 4; ++o) p.H64[idx0 + o] = hv[o];
#pragma unroll
                        for (int o = 0; o < 4; ++o) p.Hio[idx0 + o] = (float)hv[o];
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 16 * NKB; ++q) {
                    const int kb = SWAP ? q % NKB : q / 16, reg = SWAP ? q / NKB : q % 16;
                    if (lines_done && kb < 4 * G2_NFULL) continue;                       // (the full chunks went out above; what is left is the K % 128 remainder)
                    const int k = e_kloc(kb, reg);   // (EPI 1: TT == 1, one block)
                    if (!e_ok(reg) || (p.fix && p.fix[k])) continue;
                    const long idx = (long)k + (long)K * e_row(reg);
                    const float lam = p.lam ? p.lam[k] : 0.0f;
                    if (p.H64) {   // float64 master copy of H: the update in double, both arrays written
                        const double hv = p.H64[idx];
                        double hn;
                        if (p.sqrt_rule) hn = sqrt(hv * (double)acc[kb][reg]);   // lnmf.m:76
                        else if (DUAL) {
                            double gn = (double)acc[kb][reg], gp = (double)acc2[kb][reg];
                            if (p.inv_exp != 1.0f) { gn = pow(gn, (double)p.inv_exp); gp = pow(gp, (double)p.inv_exp); }
                            hn = hv * (gn / fmax(gp + (double)lam, 2.220446049250313e-16));
                        } else hn = hv * ((double)acc[kb][reg] / fmax((p.den ? (double)p.den[idx] : p.denvec[k]) + (double)lam, 2.220446049250313e-16));
                        p.H64[idx] = hn;
                        p.Hio[idx] = (float)hn;
                        continue;
                    }
                    if (p.sqrt_rule) { p.Hio[idx] = sqrtf(p.Hio[idx] * acc[kb][reg]); continue; }   // lnmf.m:76
                    if (DUAL) {   // nmf.m:186-187,193-194 + 199: numerator and denominator both come out of this pass; outer .^(1/alpha) for alpha-beta
                        float gn = acc[kb][reg], gp = acc2[kb][reg];
                        if (p.inv_exp != 1.0f) { gn = powf(gn, p.inv_exp); gp = powf(gp, p.inv_exp); }
                        p.Hio[idx] = p.Hio[idx] * (gn / fmaxf(gp + lam, NMFX_EPS_F));
                        continue;
                    }
                    const float den = p.den ? p.den[idx] : (float)p.denvec[k];
                    p.Hio[idx] = p.Hio[idx] * (acc[kb][reg] / fmaxf(den + lam, NMFX_EPS_F));
                }
        }
    }
    if (p.cost_partials) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cost += __shfl_xor(cost, o);
        double *red = reinterpret_cast<double *>(lds);
        __syncthreads();
        if (lane == 0) red[w] = cost;
        __syncthreads();
        // KL, mode 0: the kernel sums V.*log2(V./V_hat); ln 2 is applied here, sum(V_hat) - sum(V) by the caller in closed form.  Modes 1 / 2: the complete
        // divergence sum(V.*log(V./V_hat) - V + V_hat) of nmf.m:210 / cnmf.m:243 over this workgroup's elements
        if (tid == 0) p.cost_partials[(long)blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1] + red[2] + red[3]) * ((MF == 3 && NMFX_KL_MODE == 0) ? 0.6931471805599453 : 1.0);
    }
}

}  // namespace nmfx
