// Pipelined fp32 MFMA GEMM kernel template (gfx950), instantiated by gemm_pipe.hip (whole-tile problems) and gemm_pipe_edge.hip
// (edge tiles, odd leading dimensions); the general kernel, the views and the host-side dispatch live in gemm.hip.
#pragma once
#include <type_traits>

#include "gemm_common.h"

namespace nmfx {

// =====================================================================================================================
// Pipelined variant for tile-aligned problems (the hot instantiations of cnmf / the materialised paths).
// Same tiling, LDS layout, MFMA roles and epilogue as gemm_kernel, but nothing is left outside the MFMA stream: a wave
// issues in order, so every global load (two k-tiles ahead, into one of two register sets), every element map + LDS store
// (one k-tile ahead) and every LDS operand read (one MFMA step ahead) is placed in the 64-cycle shadow of a specific MFMA
// (sched_barrier after each).  Tile indices past the end are clamped instead of branched on: the redundant loads hit L2
// and the redundant LDS stores land in the buffer nobody reads again.
// =====================================================================================================================
// VEC: float4 loads (16-byte aligned views whose contiguous dimension is a multiple of 4).  !VEC: four dword loads per chunk at
// immediate offsets and per-element masks, for odd leading dimensions (m = 513, 1025, ... spectrogram bins) and odd M / N / Kc.
// EDGE: tile edges in M / N and a short last k-tile are possible (row / k guards, guarded epilogue); !EDGE is the lean stream for
// problems made of whole tiles.
template <int BR, bool KC, bool PRO, bool VEC, bool EDGE>
struct PLoader {
    static constexpr int NCH = BR * BK / 4 / NTHREADS;     // float4 chunks per thread per k-tile: 4 (BR = 128) or 2 (BR = 64)
    static constexpr int LDS_STRIDE = BR + (KC ? 1 : 0);
    static constexpr int CPR = KC ? (BK / 4) : (BR / 4);
    static constexpr int LSTEP = NTHREADS / CPR;
    float4 x[2][NCH];
    float4 y[PRO ? 2 : 1][PRO ? NCH : 1];
    int okm[2];                  // bit p: chunk p of the set is inside the view (shift zero-fill otherwise)
    long offr[KC ? NCH : 1];
    int gr[KC ? NCH : 1];
    int c, q;
    int rowok;                   // bit p: the rows of chunk p exist (r < R); edge tiles of M / N that are not tile multiples
    int nval[2][VEC ? 1 : NCH];  // !VEC: leading elements of chunk p that are inside the matrix (0..4)
    int rval;                    // !VEC, RC: valid rows of this thread's 4-row chunk
    long eoff[(!VEC && !KC) ? 4 : 1];   // !VEC, RC: dec_r of each of the 4 rows (a stacked r-view with K % 4 != 0 changes t block inside a chunk)
    int eg[(!VEC && !KC) ? 4 : 1];
    long kend;                   // contraction indices >= kend read as zero (last k-tile of a Kc that is not a multiple of BK)
    int kt, kin;                 // KC stacked views: t block and offset inside it of THIS THREAD's chunk (kc_next + 4c) in the next tile to load
    int q32, r32;                // BK / blk, BK % blk: how (kt, kin) move per k-tile
    long kc_next;                // first contraction index of the next tile to load

    __device__ __forceinline__ void init(const OpView &v, int tid, int r_tile0, long kbeg, long R, long kend_) {
        c = tid % CPR;
        q = tid / CPR;
        kend = kend_;
        rowok = 0;
        if (KC) {
#pragma unroll
            for (int p = 0; p < NCH; ++p) {
                dec_r(v, r_tile0 + q + p * LSTEP, offr[p], gr[p]);
                rowok |= (r_tile0 + q + p * LSTEP < R) ? (1 << p) : 0;
            }
        } else {
            dec_r(v, r_tile0 + 4 * c, offr[0], gr[0]);
            rowok = (r_tile0 + 4 * c < R) ? 1 : 0;           // VEC: R % 4 == 0, a chunk of 4 rows is inside or outside as a whole
            const long left = R - (r_tile0 + 4 * c);
            rval = left >= 4 ? 4 : (left > 0 ? (int)left : 0);
            if (!VEC) {
#pragma unroll
                for (int e = 0; e < 4; ++e) dec_r(v, r_tile0 + 4 * c + e, eoff[(!VEC && !KC) ? e : 0], eg[(!VEC && !KC) ? e : 0]);
            }
        }
        kc_next = kbeg;
        kt = 0; kin = 0; q32 = 0; r32 = 0;
        if (KC && v.mode >= VIEW_HSTACK_KC) {
            // EDGE: any block length -- the (t, offset) pair is per thread (chunk kc_next + 4c) and walks with the tiles.
            // !EDGE: blk % BK == 0, a k-tile lies inside one block: the pair is tile-uniform (scalar registers, no VALU)
            const long kc0 = kbeg + (EDGE ? 4 * c : 0);
            kt = (int)(kc0 / v.blk); kin = (int)(kc0 - (long)kt * v.blk);
            q32 = BK / v.blk; r32 = BK - q32 * v.blk;
        }
    }
    // dec_k of this thread's chunk (contraction index kc_next + 4c) for the KC views
    __device__ __forceinline__ void kc_thread(const OpView &v, long &off, int &g) const {
        switch (v.mode) {
        case VIEW_HSTACK_KC: off = (long)kin - v.ld * kt; g = -kt; break;
        case VIEW_WSTACK_KC: off = (long)kin + v.tstride * kt; g = 0; break;
        case VIEW_XSHIFT_KC: off = (long)kin + v.ld * kt; g = -kt; break;
        default: off = kc_next; g = 0; break;   // VIEW_KC
        }
        if (!EDGE || v.mode < VIEW_HSTACK_KC) off += 4 * c;   // the per-thread pair of EDGE already contains the chunk offset
    }
    // issue the global load(s) of chunk P of the next tile into register set SET
    template <int SET, int P>
    __device__ __forceinline__ void issue(const OpView &v) {
        long off; int g;
        if (KC) {
            long ok; int gk;
            kc_thread(v, ok, gk);
            off = offr[P] + ok;
            g = gr[P] + gk;
            if (EDGE) { if (!((rowok >> P) & 1) || kc_next + 4 * c >= kend) g = -1; }   // Kc % 4 == 0: a chunk of 4 k is inside or outside as a whole
        } else {
            const long kc = kc_next + q + P * LSTEP;
            off = offr[0] + v.ld * kc;
            g = gr[0] + (v.mode == VIEW_HSTACK_RC ? (int)kc + v.goff : 0);
            if (EDGE) { if (!rowok || kc >= kend) g = -1; }
        }
        const bool ok = g >= 0;
        if (P == 0) okm[SET] = 0;
        okm[SET] |= ok ? (1 << P) : 0;
        if (VEC) {
            const long o = ok ? off : v.safe;   // out-of-view chunks read an element that always exists (OpView.safe) and are zeroed at commit
            x[SET][P] = *reinterpret_cast<const float4 *>(v.p + o);
            if (PRO) { if (v.p2) y[PRO ? SET : 0][PRO ? P : 0] = *reinterpret_cast<const float4 *>(v.p2 + o); }
        } else {
            // elements past the edge of a straddling chunk would be out of bounds on the last column: clamp each address.  In a
            // stacked k-view whose block length is not a multiple of 4 the chunk may also cross into the next t block: those
            // elements live at another address and under the next block's (stricter) shift guard -- validity stays a prefix.
            int lim;
            if (KC) { const long left = kend - (kc_next + 4 * c); lim = left >= 4 ? 4 : (left > 0 ? (int)left : 0); }
            else lim = rval;
            if (!ok) lim = 0;
            long eo1 = 1, eo2 = 2, eo3 = 3;
            int nv = lim;
            if (!KC) {   // rows of the chunk decoded one by one (their shift guards only get stricter along the chunk: validity stays a prefix)
                const long kc = kc_next + q + P * LSTEP;
                const int gk = v.mode == VIEW_HSTACK_RC ? (int)kc + v.goff : 0;
                eo1 = eoff[(!VEC && !KC) ? 1 : 0] - eoff[0];
                eo2 = eoff[(!VEC && !KC) ? 2 : 0] - eoff[0];
                eo3 = eoff[(!VEC && !KC) ? 3 : 0] - eoff[0];
                if (nv > 1 && eg[(!VEC && !KC) ? 1 : 0] + gk < 0) nv = 1;
                if (nv > 2 && eg[(!VEC && !KC) ? 2 : 0] + gk < 0) nv = 2;
                if (nv > 3 && eg[(!VEC && !KC) ? 3 : 0] + gk < 0) nv = 3;
            }
            if (KC && v.mode >= VIEW_HSTACK_KC) {
                const long dwrap = -(long)v.blk + (v.mode == VIEW_HSTACK_KC ? -v.ld : (v.mode == VIEW_WSTACK_KC ? v.tstride : v.ld));
                const bool gnext = v.mode == VIEW_WSTACK_KC || g - 1 >= 0;     // shift guard of block t + 1
                const int first_wrapped = v.blk - kin;                             // elements e >= first_wrapped belong to block t + 1
                if (first_wrapped <= 1) eo1 += dwrap;
                if (first_wrapped <= 2) eo2 += dwrap;
                if (first_wrapped <= 3) eo3 += dwrap;
                if (!gnext && first_wrapped < nv) nv = first_wrapped;
            }
            nval[SET][VEC ? 0 : P] = nv;
            const float *b1 = v.p + (nv > 0 ? off : v.safe);
            x[SET][P] = make_float4(b1[0], b1[nv > 1 ? eo1 : 0], b1[nv > 2 ? eo2 : 0], b1[nv > 3 ? eo3 : 0]);
            if (PRO) {
                if (v.p2) {
                    const float *b2 = v.p2 + (nv > 0 ? off : v.safe);
                    y[PRO ? SET : 0][PRO ? P : 0] = make_float4(b2[0], b2[nv > 1 ? eo1 : 0], b2[nv > 2 ? eo2 : 0], b2[nv > 3 ? eo3 : 0]);
                }
            }
        }
    }
    // advance to the following tile unless `last` (clamped re-load of the final tile)
    __device__ __forceinline__ void advance(const OpView &v, bool more) {
        if (!more) return;
        kc_next += BK;
        if (KC && v.mode >= VIEW_HSTACK_KC) {
            if (EDGE) { kt += q32; kin += r32; if (kin >= v.blk) { kin -= v.blk; ++kt; } }
            else { kin += BK; if (kin >= v.blk) { kin -= v.blk; ++kt; } }
        }
    }
    // element map + LDS store of chunk P of register set SET
    template <int SET, int P>
    __device__ __forceinline__ void commit(const OpView &v, float *S) {
        float4 t = x[SET][P];
        if (PRO) { if (v.func != NMFX_PRO_NONE) t = pro4<false>(v.func, t, y[PRO ? SET : 0][PRO ? P : 0], 0.f, 0.f); }
        if (!((okm[SET] >> P) & 1)) t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!VEC) {   // mask after the element map: 0/0 of padding must not become NaN in the tile
            const int nv = nval[SET][VEC ? 0 : P];
            if (nv < 4) t.w = 0.f;
            if (nv < 3) t.z = 0.f;
            if (nv < 2) t.y = 0.f;
            if (nv < 1) t.x = 0.f;
        }
        const int line = q + P * LSTEP;
        if (KC) {
            S[(4 * c + 0) * LDS_STRIDE + line] = t.x;
            S[(4 * c + 1) * LDS_STRIDE + line] = t.y;
            S[(4 * c + 2) * LDS_STRIDE + line] = t.z;
            S[(4 * c + 3) * LDS_STRIDE + line] = t.w;
        } else {
            *reinterpret_cast<float4 *>(&S[line * LDS_STRIDE + 4 * c]) = t;
        }
    }
};

template <int BM, int BN, bool A_KC, bool B_KC, bool PRO, bool VEC, bool EDGE>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_pipe_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using LA = PLoader<BM, A_KC, PRO, VEC, EDGE>;
    using LB = PLoader<BN, B_KC, PRO, VEC, EDGE>;
    constexpr int LDA_S = LA::LDS_STRIDE, LDB_S = LB::LDS_STRIDE;
    constexpr int A_SZ_AL = (BK * LDA_S + 3) & ~3, B_SZ_AL = (BK * LDB_S + 3) & ~3;
    constexpr int BUF_SZ = A_SZ_AL + B_SZ_AL;
    constexpr int MR = BM / 64, NR = BN / 64, NM = MR * NR;
    constexpr int NPIECE = LA::NCH + LB::NCH;            // 8, or 6 with a 64-wide tile
    static_assert(NPIECE <= 8, "piece schedule assumes <= 8 chunks per tile");

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wi0 = (wave & 1) * (BM / 2), wj0 = (wave >> 1) * (BN / 2);
    // XCD-aware tile order: workgroups are dealt to the 8 XCDs round-robin by linear id (x fastest) and each XCD has its own L2.  XCD k gets
    // the k-th CONTIGUOUS eighth of the (x fastest) tile order instead of every eighth tile, so the tiles an XCD runs together share their
    // B panel (all i tiles of a column block) and walk the contraction in step: a skinny product (Q = W_flat' * V: 4 x 128 tiles) then
    // fetches each column block of V into ONE L2 instead of four (HBM read per launch 1.11e9 -> see profiles/archive/r2_13_c4_pmc.md).
    unsigned bx = blockIdx.x, by = blockIdx.y;
    {
        const unsigned nt = gridDim.x * gridDim.y;
        if ((nt & 7u) == 0u && nt >= 16u) {
            const unsigned id = bx + gridDim.x * by;
            const unsigned tl = (id & 7u) * (nt >> 3) + (id >> 3);
            bx = tl % gridDim.x; by = tl / gridDim.x;
        }
    }
    const int i_tile0 = bx * BM, j_tile0 = by * BN;

    long kbeg = 0, kend = p.Kc;
    float *C = p.C;
    OpView vA = p.A, vB = p.B;
    if (p.zbatch > 0) {   // blockIdx.z = shift index t: same contraction on shifted operands, slab t
        const int z = blockIdx.z;
        vA.p += (long)z * p.zA_off;
        vB.p += (long)z * p.zB_off;
        vB.lim += z * p.zB_lim;
        vB.tstride += z * p.zB_tstride;
        C += (long)z * p.slab_stride;
    } else if (p.splitk > 1) {
        kbeg = (long)blockIdx.z * p.kc_per_split;
        kend = kbeg + p.kc_per_split < p.Kc ? kbeg + p.kc_per_split : p.Kc;
        C += (long)blockIdx.z * p.slab_stride;
    }
    const int ntiles = (int)((kend - kbeg + BK - 1) / BK);

    LA la; LB lb;
    la.init(vA, tid, i_tile0, kbeg, p.M, kend);
    lb.init(vB, tid, j_tile0, kbeg, p.N, kend);

    f32x16 acc[NR][MR];
#pragma unroll
    for (int a = 0; a < NR; ++a)
#pragma unroll
        for (int b = 0; b < MR; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.0f;

    // compile-time unrolled helpers over the chunk pieces: pieces 0..NCH_A-1 belong to A, the rest to B
    auto issue_piece = [&](auto set_c, auto piece_c) {
        constexpr int SET = decltype(set_c)::value, PC = decltype(piece_c)::value;
        if constexpr (PC < LA::NCH) la.template issue<SET, PC>(vA);
        else if constexpr (PC < NPIECE) lb.template issue<SET, PC - LA::NCH>(vB);
    };
    auto commit_piece = [&](auto set_c, auto piece_c, float *buf) {
        constexpr int SET = decltype(set_c)::value, PC = decltype(piece_c)::value;
        if constexpr (PC < LA::NCH) la.template commit<SET, PC>(vA, buf);
        else if constexpr (PC < NPIECE) lb.template commit<SET, PC - LA::NCH>(vB, buf + A_SZ_AL);
    };
    auto for_pieces = [&](auto f) {
        f(std::integral_constant<int, 0>{}); f(std::integral_constant<int, 1>{}); f(std::integral_constant<int, 2>{}); f(std::integral_constant<int, 3>{});
        f(std::integral_constant<int, 4>{}); f(std::integral_constant<int, 5>{}); f(std::integral_constant<int, 6>{}); f(std::integral_constant<int, 7>{});
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;

    if (ntiles > 0) {
        // prologue: tile 0 -> set 0 -> LDS buffer 0; tile 1 (or 0 again) -> set 1, left in flight
        for_pieces([&](auto pc) { issue_piece(S0{}, pc); });
        la.advance(vA, ntiles > 1); lb.advance(vB, ntiles > 1);
        for_pieces([&](auto pc) { issue_piece(S1{}, pc); });
        la.advance(vA, ntiles > 2); lb.advance(vB, ntiles > 2);
        for_pieces([&](auto pc) { commit_piece(S0{}, pc, smem); });
    }
    __syncthreads();

    // one k-tile: MFMAs on LDS buffer CUR; loads of tile t+2 into set CUR (its previous content, tile t, is already in LDS);
    // commits of set CUR^1 (tile t+1) into LDS buffer CUR^1
    auto tile = [&](auto cur_c, int t) {
        constexpr int CUR = decltype(cur_c)::value;
        const float *Ac = smem + CUR * BUF_SZ, *Bc = Ac + A_SZ_AL;
        float *Nb = smem + (CUR ^ 1) * BUF_SZ;
        float fa[2][NR], fb[2][MR];
#pragma unroll
        for (int a = 0; a < NR; ++a) fa[0][a] = Bc[h * LDB_S + wj0 + 32 * a + l31];
#pragma unroll
        for (int b = 0; b < MR; ++b) fb[0][b] = Ac[h * LDA_S + wi0 + 32 * b + l31];
        auto step = [&](auto kk_c) {
            constexpr int kk = decltype(kk_c)::value;
            constexpr int cb = kk & 1, nb = cb ^ 1;
#pragma unroll
            for (int j = 0; j < NM; ++j) {
                const int a = j / MR, b = j % MR;
                if (kk + 1 < BK / 2) {   // operand registers of step kk+1, one or two per MFMA slot
                    constexpr int per = (NR + MR + NM - 1) / NM;
#pragma unroll
                    for (int u = 0; u < per; ++u) {
                        const int o = j * per + u;
                        if (o < NR) fa[nb][o] = Bc[(2 * kk + 2 + h) * LDB_S + wj0 + 32 * o + l31];
                        else if (o < NR + MR) fb[nb][o - NR] = Ac[(2 * kk + 2 + h) * LDA_S + wi0 + 32 * (o - NR) + l31];
                    }
                }
                if (j == 0 && kk < 8) issue_piece(cur_c, std::integral_constant<int, (kk < 8 ? kk : 0)>{});
                if (j == NM - 1 && kk >= 8) commit_piece(std::integral_constant<int, CUR ^ 1>{}, std::integral_constant<int, (kk >= 8 ? kk - 8 : 0)>{}, Nb);
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cb][a], fb[cb][b], acc[a][b], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
        step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{}); step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{});
        step(std::integral_constant<int, 8>{}); step(std::integral_constant<int, 9>{}); step(std::integral_constant<int, 10>{}); step(std::integral_constant<int, 11>{});
        step(std::integral_constant<int, 12>{}); step(std::integral_constant<int, 13>{}); step(std::integral_constant<int, 14>{}); step(std::integral_constant<int, 15>{});
        la.advance(vA, t + 3 < ntiles); lb.advance(vB, t + 3 < ntiles);
        __syncthreads();
    };
    for (int t = 0; t < ntiles; t += 2) {
        tile(S0{}, t);
        if (t + 1 < ntiles) tile(S1{}, t + 1);
    }

    // epilogue: identical to gemm_kernel's (acc[a][b][e] = C[i][j], i = i_tile0+wi0+32b+l31, j = j_tile0+wj0+32a+(e&3)+8(e>>2)+4h)
    double part = 0.0;
#pragma unroll
    for (int a = 0; a < NR; ++a)
#pragma unroll
        for (int b = 0; b < MR; ++b) {
            const long i = i_tile0 + wi0 + 32 * b + l31;
            float pf = 0.0f;   // this block's 16 cost terms
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const long j = j_tile0 + wj0 + 32 * a + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (EDGE) { if (i >= p.M || j >= p.N) continue; }       // edge tiles
                float sv = acc[a][b][e];
                if (p.epi == EPI_COST) {
                    if (p.cost_ncols == 0 || j < p.cost_ncols) pf += div_term_f32(p.cost_div, p.Vref[i + p.ldv * j], sv);
                    if (p.store_c) C[i + p.ldc * j] = sv;
                } else {
                    if (p.accumulate) sv += C[i + p.ldc * j];
                    if (p.clamp0) sv = fmaxf(sv, 0.0f);
                    C[i + p.ldc * j] = sv;
                }
            }
            part += (double)pf;
        }
    if (p.epi == EPI_COST) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        double *red = reinterpret_cast<double *>(smem);
        __syncthreads();
        if (lane == 0) red[wave] = part;
        __syncthreads();
        if (tid == 0) p.cost_partials[(long)blockIdx.y * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    }
}

template <int BM, int BN, bool A_KC, bool B_KC, bool PRO, bool VEC, bool EDGE>
static nmfx_status launch_pipe_cfg(hipStream_t st, const GemmParams &p) {
    using LA = PLoader<BM, A_KC, PRO, VEC, EDGE>;
    using LB = PLoader<BN, B_KC, PRO, VEC, EDGE>;
    constexpr int A_SZ_AL = (BK * LA::LDS_STRIDE + 3) & ~3, B_SZ_AL = (BK * LB::LDS_STRIDE + 3) & ~3;
    const size_t lds = sizeof(float) * 2 * (A_SZ_AL + B_SZ_AL);
    auto kern = gemm_pipe_kernel<BM, BN, A_KC, B_KC, PRO, VEC, EDGE>;
    static LdsAttrOnce lds_attr;
    TRY(lds_attr.set(reinterpret_cast<const void *>(kern), (int)lds));
    dim3 grid((unsigned)((p.M + BM - 1) / BM), (unsigned)((p.N + BN - 1) / BN), (unsigned)(p.zbatch > 0 ? p.zbatch : (p.splitk > 1 ? p.splitk : 1)));
    hipLaunchKernelGGL(kern, grid, dim3(NTHREADS), lds, st, p);
    NMFX_HIP(hipGetLastError());
    return NMFX_OK;
}

template <int BM, int BN, bool VEC, bool EDGE>
static nmfx_status dispatch_pipe_t(hipStream_t st, const GemmParams &p) {
    const bool akc = is_kc(p.A.mode), bkc = is_kc(p.B.mode);
    const bool pro = p.A.func != NMFX_PRO_NONE || p.B.func != NMFX_PRO_NONE;
#define NMFX_PIPE(AK, BKC_)                                                                     \
    return pro ? launch_pipe_cfg<BM, BN, AK, BKC_, true, VEC, EDGE>(st, p) : launch_pipe_cfg<BM, BN, AK, BKC_, false, VEC, EDGE>(st, p)
    if (akc && bkc) { NMFX_PIPE(true, true); }
    if (akc) { NMFX_PIPE(true, false); }
    if (bkc) { NMFX_PIPE(false, true); }
    NMFX_PIPE(false, false);
#undef NMFX_PIPE
}

}  // namespace nmfx
