// fp64 matrix-core helper shared by the Riccati sweep (256-thread workgroups) and the per-knot kernels (single wave):
// v_mfma_f64_16x16x4_f64 — lane l feeds A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; it owns C rows (l>>4) + 4r, column l&15.
#pragma once
#include "hs_common.hpp"

namespace hs {

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef float f4_t __attribute__((ext_vector_type(4)));
// accumulator type and lane -> row map of the 16x16x4 matrix-core instruction per scalar type.  Lane l = (li = l & 15, lk = l >> 4) feeds
// A[i = li][k = lk], B[k = lk][j = li] in both; it owns column li of C and the rows  lk + 4 r  (v_mfma_f64_16x16x4_f64)  /  4 lk + r
// (v_mfma_f32_16x16x4_f32), r = 0..3  (tools/mfma_layout_test.hip prints both maps from the hardware).
template <class R> struct MfmaT;
template <> struct MfmaT<double> {
    using acc = d4_t;
    static HD int row(int lk, int r) { return lk + 4 * r; }
#ifndef HS_HOST_EMU
    static HD acc mma(double a, double b, acc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    // v_mfma_f64_4x4x4_4b_f64: four independent 4 x 4 x 4 products in 16 cycles (the 16 x 16 x 4 form takes 64; tools/probe/mfma_f64_4x4.hip measured
    // both and printed the lane maps).  Block b = (l >> 2) & 3; lane l feeds A_b[i = l & 3][k = l >> 4] and B_b[k = l >> 4][j = l & 3], owns C_b[l >> 4][l & 3].
    static HD double mma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
#endif
};
template <> struct MfmaT<float> {
    using acc = f4_t;
    static HD int row(int lk, int r) { return 4 * lk + r; }
#ifndef HS_HOST_EMU
    static HD acc mma(float a, float b, acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static HD float mma4(float, float, float c) { return c; }      // (no edge form in fp32: mfma_edge_tile is false for it)
#endif
};

// Several 16x16 output tiles of ONE wave, accumulated together: the k-loop is outermost so that the MFMAs issued back to
// back belong to different accumulators (a tile's own MFMAs form a dependent chain) and all operand loads of a k-step are in
// flight together.  A tile may chain a second product onto the same accumulator (A^T HA + C^T lC).
template <class R = double> struct MTileT {
    R* Cout; int ldc; const R* Cin; int ldcin; int i0, j0, M_, N_;
    const R* A; int lda; const R* B; int ldb; int K; bool TA;     // A(i,k) = TA ? A[k + lda*i] : A[i + lda*k] ; B(k,j) = TB ? B[j + ldb*k] : B[k + ldb*j] ; k >= K reads as 0
    const R* A2; int lda2; const R* B2; int ldb2; int K2;      // second product (always A2^T B2), K2 = 0: none
    bool TB = false;
    bool TC = false;      // C(i,j) (and Cin) at C[j + ldc*i] instead of C[i + ldc*j]
    bool mirror = false;  // also store C(i,j) at the transposed position (an off-diagonal tile of a symmetric result whose lower partner is not formed)
    R dadd = 0.0;    // added to the entries C(i,i) of the tile's block (i.e. where the absolute row equals the absolute column)
    // structured operand [I, s I] folded in as an addend (whole-body A = [I, dt I; A21, A22]: the products run over the lower rows only):
    //   tmode 1 (H [I, s I]):    C(i,j) += j < tsplit ? T(i,j) : tscale * T(i, j - tsplit)
    //   tmode 2 ([I, s I]^T HA): C(i,j) += i < tsplit ? T(i,j) : tscale * T(i - tsplit, j)          T(i,j) = T[i + ldt*j]
    const R* T = nullptr; int ldt = 0; int tmode = 0; int tsplit = 0; R tscale = 0.0;
    HD R top(int i, int j) const {
        if (tmode == 1) return j < tsplit ? T[i + ldt * j] : tscale * T[i + ldt * (j - tsplit)];
        if (tmode == 2) return i < tsplit ? T[i + ldt * j] : tscale * T[(i - tsplit) + ldt * j];
        return R(0.0);
    }
};
using MTile = MTileT<double>;
// EDGE TILES.  A tile with at most 4 rows (or 4 columns) inside the matrix - rows / columns 32..35 of the 36-wide whole-body blocks: 5 of the 9 tiles of a
// 36 x 36 product - wastes three quarters of a 16 x 16 x 4 instruction.  In fp64 it is formed by the 4 x 4 x 4 x 4-block instruction instead: the strip's (up to)
// four 4 x 4 blocks are the instruction's four blocks, same operands and operand reads, a quarter of the matrix-core time.
#ifndef MFMA_EDGE4
#define MFMA_EDGE4 1
#endif
template <class R> HD constexpr bool mfma_edge_tile(int M_, int N_, int i0, int j0) { return MFMA_EDGE4 && sizeof(R) == 8 && (M_ - i0 <= 4 || N_ - j0 <= 4); }
template <int NTL, int KMAX, int KMAX2, class R>
HD void mfma_tiles(int lane, const MTileT<R>* td) {
#ifdef HS_HOST_EMU
    if (lane != 0) return;     // the emulator lets lane 0 stand for the wave (plain loops); the lane mapping is verified on the GPU
    static R res[NTL][256];   // every tile is formed before any is stored, like the GPU path (in-place products)
    for (int t = 0; t < NTL; t++) {
        const MTileT<R>& T = td[t];
        for (int j = T.j0; j < T.j0 + 16 && j < T.N_; j++) for (int i = T.i0; i < T.i0 + 16 && i < T.M_; i++) {
            R s = T.Cin ? (T.TC ? T.Cin[j + T.ldcin * i] : T.Cin[i + T.ldcin * j]) : 0.0;
            s += T.top(i, j);
            for (int k = 0; k < T.K; k++) s += (T.TA ? T.A[k + T.lda * i] : T.A[i + T.lda * k]) * (T.TB ? T.B[j + T.ldb * k] : T.B[k + T.ldb * j]);
            for (int k = 0; k < T.K2; k++) s += T.A2[k + T.lda2 * i] * T.B2[k + T.ldb2 * j];
            res[t][(i - T.i0) + 16 * (j - T.j0)] = s;
        }
    }
    for (int t = 0; t < NTL; t++) {
        const MTileT<R>& T = td[t];
        for (int j = T.j0; j < T.j0 + 16 && j < T.N_; j++) for (int i = T.i0; i < T.i0 + 16 && i < T.M_; i++) {
            const R v = res[t][(i - T.i0) + 16 * (j - T.j0)] + ((i == j) ? T.dadd : 0.0);
            if (T.TC) T.Cout[j + T.ldc * i] = v; else T.Cout[i + T.ldc * j] = v;
            if (T.mirror) { if (T.TC) T.Cout[i + T.ldc * j] = v; else T.Cout[j + T.ldc * i] = v; }
        }
    }
#else
    const int li = lane & 15, lk = lane >> 4;
    const int c4 = lane & 3, bk = (lane >> 2) & 3, r4 = lane >> 4;      // lane maps of the 4-block instruction (edge tiles)
    typename MfmaT<R>::acc c[NTL];
    // per tile: (row, column) of the entries the lane feeds and owns.  Full tile: A row i0 + li, B column j0 + li, outputs (i0 + row(lk, r), j0 + li), r = 0..3.
    // Edge tile with <= 4 rows: block bk covers columns j0 + 4 bk ..; with <= 4 columns: rows i0 + 4 bk ..; one output per lane (c[t][0]).
#define HS_TILE_MAP(t) \
        const bool edge = mfma_edge_tile<R>(td[t].M_, td[t].N_, td[t].i0, td[t].j0), rs = td[t].M_ - td[t].i0 <= 4; \
        const int ia = edge ? td[t].i0 + (rs ? c4 : 4 * bk + c4) : td[t].i0 + li, jb = edge ? td[t].j0 + (rs ? 4 * bk + c4 : c4) : td[t].j0 + li; \
        const int kl = edge ? r4 : lk; (void)kl; (void)ia; (void)jb;
    _Pragma("unroll") for (int t = 0; t < NTL; t++) {
        HS_TILE_MAP(t)
        if (edge) {
            const int row = td[t].i0 + (rs ? r4 : 4 * bk + r4), j = jb;
            c[t][0] = (td[t].Cin && row < td[t].M_ && j < td[t].N_) ? (td[t].TC ? td[t].Cin[j + td[t].ldcin * row] : td[t].Cin[row + td[t].ldcin * j]) : 0.0;
            if (td[t].tmode != 0 && row < td[t].M_ && j < td[t].N_) c[t][0] += td[t].top(row, j);
        } else {
            const int j = td[t].j0 + li;
            _Pragma("unroll") for (int r = 0; r < 4; r++) { const int row = td[t].i0 + MfmaT<R>::row(lk, r); c[t][r] = (td[t].Cin && row < td[t].M_ && j < td[t].N_) ? (td[t].TC ? td[t].Cin[j + td[t].ldcin * row] : td[t].Cin[row + td[t].ldcin * j]) : 0.0;
                if (td[t].tmode != 0 && row < td[t].M_ && j < td[t].N_) c[t][r] += td[t].top(row, j); }
        }
    }
    _Pragma("unroll") for (int kg = 0; kg < KMAX / 4; kg++) {
        _Pragma("unroll") for (int t = 0; t < NTL; t++) if (4 * kg < td[t].K) {
            HS_TILE_MAP(t)
            const int i = ia, j = jb, k = 4 * kg + kl;
            const bool kv = k < td[t].K;
            const R a = (i < td[t].M_ && kv) ? (td[t].TA ? td[t].A[k + td[t].lda * i] : td[t].A[i + td[t].lda * k]) : 0.0;
            const R b = (j < td[t].N_ && kv) ? (td[t].TB ? td[t].B[j + td[t].ldb * k] : td[t].B[k + td[t].ldb * j]) : 0.0;
            if (edge) c[t][0] = MfmaT<R>::mma4(a, b, c[t][0]); else c[t] = MfmaT<R>::mma(a, b, c[t]);
        }
    }
    _Pragma("unroll") for (int kg = 0; kg < KMAX2 / 4; kg++) {
        _Pragma("unroll") for (int t = 0; t < NTL; t++) if (4 * kg < td[t].K2) {
            HS_TILE_MAP(t)
            const int i = ia, j = jb, k = 4 * kg + kl;
            const R a = (i < td[t].M_) ? td[t].A2[k + td[t].lda2 * i] : 0.0;
            const R b = (j < td[t].N_) ? td[t].B2[k + td[t].ldb2 * j] : 0.0;
            if (edge) c[t][0] = MfmaT<R>::mma4(a, b, c[t][0]); else c[t] = MfmaT<R>::mma(a, b, c[t]);
        }
    }
    _Pragma("unroll") for (int t = 0; t < NTL; t++) {
        HS_TILE_MAP(t)
        if (edge) {
            const int row = td[t].i0 + (rs ? r4 : 4 * bk + r4), j = jb;
            if (row < td[t].M_ && j < td[t].N_) { const R v = c[t][0] + ((row == j) ? td[t].dadd : 0.0); if (td[t].TC) td[t].Cout[j + td[t].ldc * row] = v; else td[t].Cout[row + td[t].ldc * j] = v;
                if (td[t].mirror) { if (td[t].TC) td[t].Cout[row + td[t].ldc * j] = v; else td[t].Cout[j + td[t].ldc * row] = v; } }
        } else {
            const int j = td[t].j0 + li;
            _Pragma("unroll") for (int r = 0; r < 4; r++) { const int row = td[t].i0 + MfmaT<R>::row(lk, r); if (row < td[t].M_ && j < td[t].N_) { const R v = c[t][r] + ((row == j) ? td[t].dadd : 0.0); if (td[t].TC) td[t].Cout[j + td[t].ldc * row] = v; else td[t].Cout[row + td[t].ldc * j] = v;
                if (td[t].mirror) { if (td[t].TC) td[t].Cout[row + td[t].ldc * j] = v; else td[t].Cout[j + td[t].ldc * row] = v; } } }
        }
    }
#undef HS_TILE_MAP
#endif
}

}  // namespace hs
