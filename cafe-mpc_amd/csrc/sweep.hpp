// Backward Riccati sweep and linear rollout: ONE WORKGROUP PER PROBLEM, sequential over knots and phases,
// value function (H, G) and the knot's A, B, C, D held in LDS for the whole sweep.
//
//   riccati_sweep   replaces MultiPhaseDDP::backward_sweep (MultiPhaseDDP.cpp:174-213), impact_aware_step (:499-503)
//                   and SinglePhase::backward_sweep (SinglePhase.cpp:323-391)
//   (k_sweep)       MultiPhaseDDP::backward_sweep_regularized (MultiPhaseDDP.cpp:136-165) wraps it in the retry loop
//   linear_rollout  MultiPhaseDDP::linear_rollout (MultiPhaseDDP.cpp:12-42) + SinglePhase::linear_rollout (SinglePhase.cpp:145-178)
// Quu^-1: the reference uses Eigen's pivoted LDLT of (Quu - 1e-9 I) and rejects on a negative pivot
// (SinglePhase.cpp:366-375).  Here: unpivoted Cholesky of the same shifted matrix; by Sylvester's law of inertia the
// accept/reject decision is the same (a non-positive pivot <=> not positive definite), the inverse agrees to rounding.
// lux is identically zero for every cost the reference ships (SinglePhaseInterface.cpp:47, MHPCCost.cpp) and is not stored.
//
// LDS layout: every matrix is column-major with an ODD padded leading dimension (37 for n-row, 13 for m/p-row
// matrices): with 8-byte elements a stride of 37 (=74 dwords) or 13 (=26 dwords) maps the 3x2 register tiles of a
// wave onto distinct bank pairs, so the transposed products (A^T HA, B^T HB, C^T lyy C ...) read LDS conflict-free.
// Each thread owns a 3x2 output tile (6 FMAs per 5 LDS reads).  The 12x12 Cholesky/inverse runs inside ONE wave
// (wave-level phases, no s_barrier) while the workgroup barrier count per knot stays at ~10.
#pragma once
#include "hs_types.hpp"

namespace hs {

constexpr int SW_NT = 256;
constexpr int SW_M = 12;    // control dimension bound of this build (whole-body phases); HKD (m=24) needs its own instantiation
constexpr int LDN = 37;     // padded leading dimension of n-row matrices
constexpr int LDM = 13;     // padded leading dimension of m-row / p-row matrices

struct SweepLds {
    double H[LDN * MAXN], A[LDN * MAXN], HA[LDN * MAXN], Qxx[LDN * MAXN];
    double B[LDN * SW_M], HB[LDN * SW_M];
    double Qux[LDM * MAXN], K[LDM * MAXN], C[LDM * MAXN], lC[LDM * MAXN];
    double D[LDM * SW_M], lD[LDM * SW_M], lyy[LDM * MAXP], Quu[LDM * SW_M], LQ[LDM * SW_M], Qi[LDM * SW_M];
    double G[MAXN], Gn[MAXN], Qx[MAXN], Qu[SW_M], dU[SW_M], ly[MAXP], def[MAXN], tmp[64];
    double dx[MAXN], dxn[MAXN], du[SW_M];
    double red[SW_NT];
    double dV1, dV2;
    unsigned long long t_last;
    int ok;
};

#define CM(M, i, j, ld) (M)[(i) + (ld) * (j)]

// optional in-kernel stamps (diagnostic builds only: -DSW_PROF): cycles per phase group, block 0 / thread 0
#if defined(SW_PROF) && !defined(HS_HOST_EMU)
__device__ unsigned long long g_sw_prof[16];
#define SW_STAMP(i) { if (blockIdx.x == 0 && threadIdx.x == 0) { unsigned long long t_ = clock64(); atomicAdd(&g_sw_prof[i], t_ - S.t_last); S.t_last = t_; } }
#define SW_STAMP0() { if (blockIdx.x == 0 && threadIdx.x == 0) S.t_last = clock64(); }
#else
#define SW_STAMP(i)
#define SW_STAMP0()
#endif

// One 3x2 register tile of  C(i,j) (+)= alpha * sum_t opA(i,t) * B(t,j) ;  opA(i,t) = TA ? A[t + lda*i] : A[i + lda*t]
template <bool TA>
HD void mm_tile(int tile, double* C, int ldc, const double* A, int lda, const double* B, int ldb, int M, int K, bool acc, double alpha) {
    const int nti = M / 3, ti = tile % nti, tj = tile / nti, i0 = 3 * ti, j0 = 2 * tj;
    double c00 = 0, c10 = 0, c20 = 0, c01 = 0, c11 = 0, c21 = 0;
    for (int t = 0; t < K; t++) {
        double a0, a1, a2;
        if (TA) { a0 = A[t + lda * i0]; a1 = A[t + lda * (i0 + 1)]; a2 = A[t + lda * (i0 + 2)]; }
        else { a0 = A[i0 + lda * t]; a1 = A[i0 + 1 + lda * t]; a2 = A[i0 + 2 + lda * t]; }
        const double b0 = B[t + ldb * j0], b1 = B[t + ldb * (j0 + 1)];
        c00 += a0 * b0; c10 += a1 * b0; c20 += a2 * b0; c01 += a0 * b1; c11 += a1 * b1; c21 += a2 * b1;
    }
    double* c0 = C + i0 + ldc * j0; double* c1 = c0 + ldc;
    if (acc) { c0[0] += alpha * c00; c0[1] += alpha * c10; c0[2] += alpha * c20; c1[0] += alpha * c01; c1[1] += alpha * c11; c1[2] += alpha * c21; }
    else { c0[0] = alpha * c00; c0[1] = alpha * c10; c0[2] = alpha * c20; c1[0] = alpha * c01; c1[1] = alpha * c11; c1[2] = alpha * c21; }
}
HD int ntiles(int M, int N) { return (M / 3) * (N / 2); }

// global (dense, ld = rows) -> LDS (padded ld); threads stride over columns with a fixed row
template <int NT> HD void ld_mat(int tid, double* dst, int ldd, const double* src, int rows, int cols) {
    const int i = tid % rows, j0 = tid / rows, js = NT / rows;
    if (j0 < js) for (int j = j0; j < cols; j += js) dst[i + ldd * j] = src[i + rows * j];
}
template <int NT> HD void st_mat(int tid, double* dst, const double* src, int lds_, int rows, int cols) {
    const int i = tid % rows, j0 = tid / rows, js = NT / rows;
    if (j0 < js) for (int j = j0; j < cols; j += js) dst[i + rows * j] = src[i + lds_ * j];
}

// One phase of the backward sweep for problem b. On entry S.G/S.H hold (Gprime, Hprime) (already through Px^T).
template <int NT>
HD bool riccati_phase(SweepLds& S, const PhaseDev& P, int b, double reg) {
    const int n = P.n, m = P.m, p = P.p, h = P.h;
    // terminal: G[h] = Phix + Gprime ; H[h] = Phixx + Hprime  (SinglePhase.cpp:326-327)
    HS_PHASE(NT, { const int i = tid % n, j0 = tid / n, js = NT / n; if (j0 < js) for (int j = j0; j < n; j += js) CM(S.H, i, j, LDN) += P.Phixx[(size_t)b * n * n + i + n * j]; }
             if (tid < n) { S.G[tid] += P.Phix[(size_t)b * n + tid]; P.G[((size_t)b * (h + 1) + h) * n + tid] = S.G[tid]; }
             if (tid == 0) { S.ok = 1; })
    for (int k = h - 1; k >= 0; k--) {
        const size_t kk = (size_t)b * h + k;
        SW_STAMP0()
        HS_PHASE(NT,
            ld_mat<NT>(tid, S.A, LDN, P.A + kk * n * n, n, n); ld_mat<NT>(tid, S.Qxx, LDN, P.lxx + kk * n * n, n, n);
            ld_mat<NT>(tid, S.B, LDN, P.B + kk * n * m, n, m); ld_mat<NT>(tid, S.Quu, LDM, P.luu + kk * m * m, m, m);
            if (p > 0) {
                ld_mat<NT>(tid, S.C, LDM, P.C + kk * p * n, p, n); ld_mat<NT>(tid, S.D, LDM, P.D + kk * p * m, p, m);
                ld_mat<NT>(tid, S.lyy, LDM, P.lyy + kk * p * p, p, p);
                if (tid < p) S.ly[tid] = P.ly[kk * p + tid];
            }
            if (tid < n) { S.Qx[tid] = P.lx[kk * n + tid]; S.def[tid] = P.Defect[((size_t)b * (h + 1) + k + 1) * n + tid]; }
            if (tid < m) S.Qu[tid] = P.lu[kk * m + tid];)
        SW_STAMP(0)
        // HA = H A ; HB = H B ; Gnext = G + H Defect[k+1] ; (p>0) lC = lyy C ; lD = lyy D
        {
            const int t1 = ntiles(n, n), t2 = t1 + ntiles(n, m), t3 = t2 + (p > 0 ? ntiles(p, n) : 0), t4 = t3 + (p > 0 ? ntiles(p, m) : 0);
            HS_PHASE(NT,
                for (int tile = tid; tile < t4; tile += NT) {
                    if (tile < t1) mm_tile<false>(tile, S.HA, LDN, S.H, LDN, S.A, LDN, n, n, false, 1.0);
                    else if (tile < t2) mm_tile<false>(tile - t1, S.HB, LDN, S.H, LDN, S.B, LDN, n, n, false, 1.0);
                    else if (tile < t3) mm_tile<false>(tile - t2, S.lC, LDM, S.lyy, LDM, S.C, LDM, p, p, false, 1.0);
                    else mm_tile<false>(tile - t3, S.lD, LDM, S.lyy, LDM, S.D, LDM, p, p, false, 1.0);
                }
                if (tid >= NT - n) { const int i = tid - (NT - n); double s = S.G[i]; for (int j = 0; j < n; j++) s += CM(S.H, i, j, LDN) * S.def[j]; S.Gn[i] = s; })
        }
        SW_STAMP(1)
        // Qxx += A^T HA (+ C^T lC) ; Qux = B^T HA (+ D^T lC) ; Quu += B^T HB (+ D^T lD) ; Qx += A^T Gn (+C^T ly) ; Qu += B^T Gn (+D^T ly)
        {
            const int t1 = ntiles(n, n), t2 = t1 + ntiles(m, n), t3 = t2 + ntiles(m, m);
            HS_PHASE(NT,
                for (int tile = tid; tile < t3; tile += NT) {
                    if (tile < t1) { mm_tile<true>(tile, S.Qxx, LDN, S.A, LDN, S.HA, LDN, n, n, true, 1.0); if (p > 0) mm_tile<true>(tile, S.Qxx, LDN, S.C, LDM, S.lC, LDM, n, p, true, 1.0); }
                    else if (tile < t2) { mm_tile<true>(tile - t1, S.Qux, LDM, S.B, LDN, S.HA, LDN, m, n, false, 1.0); if (p > 0) mm_tile<true>(tile - t1, S.Qux, LDM, S.D, LDM, S.lC, LDM, m, p, true, 1.0); }
                    else { mm_tile<true>(tile - t2, S.Quu, LDM, S.B, LDN, S.HB, LDN, m, n, true, 1.0); if (p > 0) mm_tile<true>(tile - t2, S.Quu, LDM, S.D, LDM, S.lD, LDM, m, p, true, 1.0); }
                }
                if (tid >= NT - n) {
                    const int i = tid - (NT - n); double s = 0; for (int t = 0; t < n; t++) s += CM(S.A, t, i, LDN) * S.Gn[t];
                    if (p > 0) for (int t = 0; t < p; t++) s += CM(S.C, t, i, LDM) * S.ly[t];
                    S.Qx[i] += s;
                } else if (tid >= NT - n - m) {
                    const int a = tid - (NT - n - m); double s = 0; for (int t = 0; t < n; t++) s += CM(S.B, t, a, LDN) * S.Gn[t];
                    if (p > 0) for (int t = 0; t < p; t++) s += CM(S.D, t, a, LDM) * S.ly[t];
                    S.Qu[a] += s;
                })
        }
        SW_STAMP(2)
        // regularisation (also on Qxx: quirk x); store Qu / Quu / Qux as the reference keeps them (callers read them)
        HS_PHASE(NT,
            if (tid < n) CM(S.Qxx, tid, tid, LDN) += reg;
            if (tid >= 64 && tid < 64 + m) { const int a = tid - 64; CM(S.Quu, a, a, LDM) += reg; P.Qu[kk * m + a] = S.Qu[a]; })
        HS_PHASE(NT,
            st_mat<NT>(tid, P.Quu + kk * m * m, S.Quu, LDM, m, m); st_mat<NT>(tid, P.Qux + kk * m * n, S.Qux, LDM, m, n);
            )
        SW_STAMP(3)
        // wave 0: Cholesky of (Quu - 1e-9 I) and the inverse, all inside one wave (no workgroup barrier)
        for (int j = 0; j < m; j++) {
            HS_WPHASE(if (tid >= j && tid < m) {
                double s = CM(S.Quu, tid, j, LDM) - ((tid == j) ? 1e-9 : 0.0);
                for (int t = 0; t < j; t++) s -= S.LQ[tid * LDM + t] * S.LQ[j * LDM + t];
                S.tmp[tid] = s;
            })
            HS_WPHASE(if (tid >= j && tid < m) {
                const double piv = S.tmp[j];
                if (tid == j && !(piv > 0.0)) S.ok = 0;
                const double d = sqrt(piv > 0.0 ? piv : 1.0);
                S.LQ[tid * LDM + j] = (tid == j) ? d : S.tmp[tid] / d;
            })
        }
        HS_WPHASE(if (tid < m) {   // Quu_inv column tid (symmetric; stored [row*LDM + col])
            const int c = tid;
            for (int i = 0; i < m; i++) { double s = (i == c) ? 1.0 : 0.0; for (int t = 0; t < i; t++) s -= S.LQ[i * LDM + t] * S.Qi[t * LDM + c]; S.Qi[i * LDM + c] = s / S.LQ[i * LDM + i]; }
            for (int i = m - 1; i >= 0; i--) { double s = S.Qi[i * LDM + c]; for (int t = i + 1; t < m; t++) s -= S.LQ[t * LDM + i] * S.Qi[t * LDM + c]; S.Qi[i * LDM + c] = s / S.LQ[i * LDM + i]; }
        })
        SW_STAMP(4)
        // symmetrise Qxx (threads >= 64, wave 0 was busy) — pairs (i<j)
        HS_PHASE(NT,
            for (int e = tid; e < n * n; e += NT) { const int i = e % n, j = e / n; if (i < j) { double s = (CM(S.Qxx, i, j, LDN) + CM(S.Qxx, j, i, LDN)) / 2; CM(S.Qxx, i, j, LDN) = s; CM(S.Qxx, j, i, LDN) = s; } })
        SW_STAMP(5)
        if (!S.ok) return false;
        // K = -Qi Qux ; dU = -Qi Qu
        HS_PHASE(NT,
            for (int tile = tid; tile < ntiles(m, n); tile += NT) mm_tile<false>(tile, S.K, LDM, S.Qi, LDM, S.Qux, LDM, m, m, false, -1.0);
            if (tid >= NT - m) { const int a = tid - (NT - m); double s = 0; for (int t = 0; t < m; t++) s += S.Qi[a * LDM + t] * S.Qu[t]; S.dU[a] = -s; })
        SW_STAMP(6)
        // H = Qxx + Qux^T K ; G = Qx + Qux^T dU ; dV ; store K, dU, G
        HS_PHASE(NT,
            for (int tile = tid; tile < ntiles(n, n); tile += NT) {
                const int nti = n / 3, ti = tile % nti, tj = tile / nti, i0 = 3 * ti, j0 = 2 * tj;
                for (int jj = 0; jj < 2; jj++) for (int ii = 0; ii < 3; ii++) CM(S.H, i0 + ii, j0 + jj, LDN) = CM(S.Qxx, i0 + ii, j0 + jj, LDN);
                mm_tile<true>(tile, S.H, LDN, S.Qux, LDM, S.K, LDM, n, m, true, 1.0);
            }
            if (tid >= NT - n) { const int i = tid - (NT - n); double s = S.Qx[i]; for (int t = 0; t < m; t++) s += CM(S.Qux, t, i, LDM) * S.dU[t]; S.G[i] = s; P.G[((size_t)b * (h + 1) + k) * n + i] = s; }
            else if (tid == NT - n - 1) { double dVk = 0; for (int t = 0; t < m; t++) dVk -= S.Qu[t] * S.dU[t]; S.dV1 -= dVk; S.dV2 += dVk; }
            else if (tid >= NT - n - 1 - m && tid < NT - n - 1) { const int a = tid - (NT - n - 1 - m); P.dU[kk * m + a] = S.dU[a]; })
        SW_STAMP(7)
        HS_PHASE(NT, st_mat<NT>(tid, P.K + kk * m * n, S.K, LDM, m, n);)
        SW_STAMP(8)
    }
    // G[0] += H[0] * Defect[0]   (SinglePhase.cpp:389)
    HS_PHASE(NT, if (tid < n) S.def[tid] = P.Defect[((size_t)b * (h + 1)) * n + tid];)
    HS_PHASE(NT, if (tid < n) { double s = S.G[tid]; for (int j = 0; j < n; j++) s += CM(S.H, tid, j, LDN) * S.def[j]; S.Gn[tid] = s; })
    HS_PHASE(NT, if (tid < n) { S.G[tid] = S.Gn[tid]; P.G[((size_t)b * (h + 1)) * n + tid] = S.Gn[tid]; }
             st_mat<NT>(tid, P.H0 + (size_t)b * n * n, S.H, LDN, n, n);)
    return true;
}

// full multi-phase backward sweep of problem b; returns success, writes dV into S.dV1/dV2
template <int NT>
HD bool riccati_sweep(SweepLds& S, const PhaseDev* ph, int nph, int b, double reg) {
    HS_PHASE(NT, if (tid == 0) { S.dV1 = 0.0; S.dV2 = 0.0; })
    for (int i = nph - 1; i >= 0; i--) {
        const PhaseDev& P = ph[i];
        const int n = P.n;
        if (i == nph - 1) {
            HS_PHASE(NT, for (int e = tid; e < LDN * n; e += NT) S.H[e] = 0.0; if (tid < n) S.G[tid] = 0.0;)
        } else {   // impact-aware step: (G,H) <- (Px^T G, Px^T H Px), Px = next_n x n  (MultiPhaseDDP.cpp:196-201)
            const int nn = P.next_n;
            HS_PHASE(NT, ld_mat<NT>(tid, S.A, LDN, P.Px + (size_t)b * nn * n, nn, n);)
            HS_PHASE(NT,
                for (int tile = tid; tile < ntiles(nn, n); tile += NT) mm_tile<false>(tile, S.HA, LDN, S.H, LDN, S.A, LDN, nn, nn, false, 1.0);
                if (tid >= NT - n) { const int i2 = tid - (NT - n); double s = 0; for (int t = 0; t < nn; t++) s += CM(S.A, t, i2, LDN) * S.G[t]; S.Gn[i2] = s; })
            HS_PHASE(NT,
                for (int tile = tid; tile < ntiles(n, n); tile += NT) mm_tile<true>(tile, S.H, LDN, S.A, LDN, S.HA, LDN, n, nn, false, 1.0);
                if (tid >= NT - n) S.G[tid - (NT - n)] = S.Gn[tid - (NT - n)];)
        }
        if (!riccati_phase<NT>(S, P, b, reg)) return false;
    }
    return true;
}

// linear rollout of problem b (eps = 1 in solve).  Returns dV_1, dV_2 in S.dV1/dV2.  (dense ld = rows layouts here:
// the matvecs read rows with unit stride across threads)
template <int NT>
HD void linear_rollout(SweepLds& S, const PhaseDev* ph, int nph, int b, double eps) {
    HS_PHASE(NT, if (tid == 0) { S.dV1 = 0.0; S.dV2 = 0.0; } if (tid < MAXN) S.dxn[tid] = 0.0;)
    for (int i = 0; i < nph; i++) {
        const PhaseDev& P = ph[i];
        const int n = P.n, m = P.m, h = P.h;
        if (i > 0) {   // dx_init = Px * dX_end(prev)   (MultiPhaseDDP.cpp:27-30); S.dx holds prev terminal dX
            const PhaseDev& Pp = ph[i - 1]; const int np = Pp.n;
            HS_PHASE(NT, for (int e = tid; e < n * np; e += NT) S.A[e] = Pp.Px[(size_t)b * n * np + e];)
            HS_PHASE(NT, if (tid < n) { double s = 0; for (int t = 0; t < np; t++) s += CM(S.A, tid, t, n) * S.dx[t]; S.dxn[tid] = s; })
        }
        // dX[0] = dx_init + eps * Defect[0]
        HS_PHASE(NT, if (tid < n) { double v = S.dxn[tid] + eps * P.Defect[((size_t)b * (h + 1)) * n + tid]; S.dx[tid] = v; P.dX[((size_t)b * (h + 1)) * n + tid] = v; })
        for (int k = 0; k < h; k++) {
            const size_t kk = (size_t)b * h + k;
            HS_PHASE(NT,
                for (int e = tid; e < n * n; e += NT) { S.A[e] = P.A[kk * n * n + e]; S.Qxx[e] = P.lxx[kk * n * n + e]; }
                for (int e = tid; e < n * m; e += NT) { S.B[e] = P.B[kk * n * m + e]; S.K[e] = P.K[kk * m * n + e]; }
                for (int e = tid; e < m * m; e += NT) S.Quu[e] = P.luu[kk * m * m + e];
                if (tid < n) { S.Qx[tid] = P.lx[kk * n + tid]; S.def[tid] = P.Defect[((size_t)b * (h + 1) + k + 1) * n + tid]; }
                if (tid < m) { S.Qu[tid] = P.lu[kk * m + tid]; S.dU[tid] = P.dU[kk * m + tid]; })
            HS_PHASE(NT, if (tid < m) { double s = eps * S.dU[tid]; for (int j = 0; j < n; j++) s += CM(S.K, tid, j, m) * S.dx[j]; S.du[tid] = s; })
            HS_PHASE(NT,
                if (tid < n) {
                    double s = 0; for (int j = 0; j < n; j++) s += CM(S.A, tid, j, n) * S.dx[j];
                    double s2 = 0; for (int j = 0; j < m; j++) s2 += CM(S.B, tid, j, n) * S.du[j];
                    double v = s + s2 + eps * S.def[tid];
                    S.dxn[tid] = v; P.dX[((size_t)b * (h + 1) + k + 1) * n + tid] = v;
                    double q = 0; for (int j = 0; j < n; j++) q += CM(S.Qxx, tid, j, n) * S.dx[j];
                    S.red[tid] = S.dx[tid] * q;            // dx^T lxx dx contributions
                    S.red[64 + tid] = S.Qx[tid] * S.dx[tid];
                } else if (tid >= 128 && tid < 128 + m) {
                    int a = tid - 128; double q = 0; for (int j = 0; j < m; j++) q += CM(S.Quu, a, j, m) * S.du[j];
                    S.red[tid] = S.du[a] * q; S.red[64 + tid] = S.Qu[a] * S.du[a];
                })
            HS_PHASE(NT, if (tid == 0) {
                double a1 = 0, a2 = 0, b1 = 0, b2 = 0;
                for (int j = 0; j < n; j++) { a1 += S.red[64 + j]; a2 += S.red[j]; }
                for (int j = 0; j < m; j++) { b1 += S.red[64 + 128 + j]; b2 += S.red[128 + j]; }
                S.dV1 += a1 + b1; S.dV2 += a2; S.dV2 += b2;      // (+ du^T lux dx with lux == 0)
            } if (tid < n) S.dx[tid] = S.dxn[tid];)
        }
        // terminal: dV_1 += Phix . dx ; dV_2 += dx^T Phixx dx
        HS_PHASE(NT, for (int e = tid; e < n * n; e += NT) S.Qxx[e] = P.Phixx[(size_t)b * n * n + e];)
        HS_PHASE(NT, if (tid < n) { double q = 0; for (int j = 0; j < n; j++) q += CM(S.Qxx, tid, j, n) * S.dx[j]; S.red[tid] = S.dx[tid] * q; S.red[64 + tid] = P.Phix[(size_t)b * n + tid] * S.dx[tid]; })
        HS_PHASE(NT, if (tid == 0) { double a1 = 0, a2 = 0; for (int j = 0; j < n; j++) { a1 += S.red[64 + j]; a2 += S.red[j]; } S.dV1 += a1; S.dV2 += a2; })
    }
}

}  // namespace hs
