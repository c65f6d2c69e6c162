// Backward Riccati sweep and linear rollout: ONE WORKGROUP PER PROBLEM, sequential over knots and phases,
// value function (H, G) and the knot's A, B, C, D held in LDS for the whole sweep.
//
//   riccati_sweep   replaces MultiPhaseDDP::backward_sweep (MultiPhaseDDP.cpp:174-213), impact_aware_step (:499-503)
//                   and SinglePhase::backward_sweep (SinglePhase.cpp:323-391)
//   riccati_regularized  MultiPhaseDDP::backward_sweep_regularized (MultiPhaseDDP.cpp:136-165)
//   linear_rollout  MultiPhaseDDP::linear_rollout (MultiPhaseDDP.cpp:12-42) + SinglePhase::linear_rollout (SinglePhase.cpp:145-178)
// Quu^-1: the reference uses Eigen's pivoted LDLT of (Quu - 1e-9 I) and rejects on a negative pivot
// (SinglePhase.cpp:366-375).  Here: unpivoted Cholesky of the same shifted matrix; by Sylvester's law of inertia the
// accept/reject decision is the same (a non-positive pivot <=> not positive definite), the inverse agrees to rounding.
// lux is identically zero for every cost the reference ships (SinglePhaseInterface.cpp:47, MHPCCost.cpp) and is not stored.
#pragma once
#include "hs_types.hpp"

namespace hs {

constexpr int SW_NT = 256;
constexpr int SW_M = 12;   // control dimension bound of this build (whole-body phases); HKD (m=24) needs its own instantiation

struct SweepLds {
    double H[MAXN * MAXN], A[MAXN * MAXN], HA[MAXN * MAXN], Qxx[MAXN * MAXN];
    double B[MAXN * SW_M], HB[MAXN * SW_M], Qux[SW_M * MAXN], K[SW_M * MAXN];
    double C[MAXP * MAXN], D[MAXP * SW_M], lyy[MAXP * MAXP], lC[MAXP * MAXN], lD[MAXP * SW_M];
    double Quu[SW_M * SW_M], LQ[SW_M * SW_M], Qi[SW_M * SW_M];
    double G[MAXN], Gn[MAXN], Qx[MAXN], Qu[SW_M], dU[SW_M], ly[MAXP], def[MAXN], tmp[64];
    double dx[MAXN], dxn[MAXN], du[SW_M];
    double red[SW_NT];
    double dV1, dV2;
    int ok;
};

// column-major helpers: M(i,j) = M[i + ld*j]
#define CM(M, i, j, ld) (M)[(i) + (ld) * (j)]

template <int NT> HD void ld_block(double* dst, const double* src, int n) { HS_PHASE(NT, for (int i = tid; i < n; i += NT) dst[i] = src[i];) }

// One phase of the backward sweep for problem b. On entry S.G/S.H hold (Gprime, Hprime) (already through Px^T).
template <int NT>
HD bool riccati_phase(SweepLds& S, const PhaseDev& P, int b, double reg) {
    const int n = P.n, m = P.m, p = P.p, h = P.h;
    // terminal: G[h] = Phix + Gprime ; H[h] = Phixx + Hprime  (SinglePhase.cpp:326-327)
    HS_PHASE(NT, for (int i = tid; i < n * n; i += NT) S.H[i] += P.Phixx[(size_t)b * n * n + i];
             if (tid < n) { S.G[tid] += P.Phix[(size_t)b * n + tid]; P.G[((size_t)b * (h + 1) + h) * n + tid] = S.G[tid]; }
             if (tid == 0) { S.ok = 1; })
    for (int k = h - 1; k >= 0; k--) {
        const size_t kk = (size_t)b * h + k;
        HS_PHASE(NT,
            for (int i = tid; i < n * n; i += NT) { S.A[i] = P.A[kk * n * n + i]; S.Qxx[i] = P.lxx[kk * n * n + i]; }
            for (int i = tid; i < n * m; i += NT) S.B[i] = P.B[kk * n * m + i];
            for (int i = tid; i < m * m; i += NT) S.Quu[i] = P.luu[kk * m * m + i];
            if (p > 0) {
                for (int i = tid; i < p * n; i += NT) S.C[i] = P.C[kk * p * n + i];
                for (int i = tid; i < p * m; i += NT) S.D[i] = P.D[kk * p * m + i];
                for (int i = tid; i < p * p; i += NT) S.lyy[i] = P.lyy[kk * p * p + i];
                if (tid < p) S.ly[tid] = P.ly[kk * p + tid];
            }
            if (tid < n) { S.Qx[tid] = P.lx[kk * n + tid]; S.def[tid] = P.Defect[((size_t)b * (h + 1) + k + 1) * n + tid]; }
            if (tid < m) S.Qu[tid] = P.lu[kk * m + tid];)
        // Gnext = G + H * Defect[k+1]
        HS_PHASE(NT, if (tid < n) { double s = S.G[tid]; for (int j = 0; j < n; j++) s += CM(S.H, tid, j, n) * S.def[j]; S.Gn[tid] = s; })
        // HA = H A ; HB = H B
        HS_PHASE(NT,
            for (int e = tid; e < n * n; e += NT) { int i = e % n, j = e / n; double s = 0; for (int t = 0; t < n; t++) s += CM(S.H, i, t, n) * CM(S.A, t, j, n); S.HA[e] = s; }
            for (int e = tid; e < n * m; e += NT) { int i = e % n, j = e / n; double s = 0; for (int t = 0; t < n; t++) s += CM(S.H, i, t, n) * CM(S.B, t, j, n); S.HB[e] = s; })
        // Qxx += A^T HA ; Qux = B^T HA ; Quu += B^T HB ; Qx += A^T Gn ; Qu += B^T Gn
        HS_PHASE(NT,
            for (int e = tid; e < n * n; e += NT) { int i = e % n, j = e / n; double s = 0; for (int t = 0; t < n; t++) s += CM(S.A, t, i, n) * CM(S.HA, t, j, n); S.Qxx[e] += s; }
            for (int e = tid; e < m * n; e += NT) { int a = e % m, j = e / m; double s = 0; for (int t = 0; t < n; t++) s += CM(S.B, t, a, n) * CM(S.HA, t, j, n); S.Qux[e] = s; }
            for (int e = tid; e < m * m; e += NT) { int a = e % m, c = e / m; double s = 0; for (int t = 0; t < n; t++) s += CM(S.B, t, a, n) * CM(S.HB, t, c, n); S.Quu[e] += s; }
            if (tid < n) { double s = 0; for (int t = 0; t < n; t++) s += CM(S.A, t, tid, n) * S.Gn[t]; S.Qx[tid] += s; }
            else if (tid >= 64 && tid < 64 + m) { int a = tid - 64; double s = 0; for (int t = 0; t < n; t++) s += CM(S.B, t, a, n) * S.Gn[t]; S.Qu[a] += s; })
        if (p > 0) {   // output (GRF) terms, SinglePhase.cpp:353-360
            HS_PHASE(NT,
                for (int e = tid; e < p * n; e += NT) { int i = e % p, j = e / p; double s = 0; for (int t = 0; t < p; t++) s += CM(S.lyy, i, t, p) * CM(S.C, t, j, p); S.lC[e] = s; }
                for (int e = tid; e < p * m; e += NT) { int i = e % p, j = e / p; double s = 0; for (int t = 0; t < p; t++) s += CM(S.lyy, i, t, p) * CM(S.D, t, j, p); S.lD[e] = s; })
            HS_PHASE(NT,
                for (int e = tid; e < n * n; e += NT) { int i = e % n, j = e / n; double s = 0; for (int t = 0; t < p; t++) s += CM(S.C, t, i, p) * CM(S.lC, t, j, p); S.Qxx[e] += s; }
                for (int e = tid; e < m * n; e += NT) { int a = e % m, j = e / m; double s = 0; for (int t = 0; t < p; t++) s += CM(S.D, t, a, p) * CM(S.lC, t, j, p); S.Qux[e] += s; }
                for (int e = tid; e < m * m; e += NT) { int a = e % m, c = e / m; double s = 0; for (int t = 0; t < p; t++) s += CM(S.D, t, a, p) * CM(S.lD, t, c, p); S.Quu[e] += s; }
                if (tid < n) { double s = 0; for (int t = 0; t < p; t++) s += CM(S.C, t, tid, p) * S.ly[t]; S.Qx[tid] += s; }
                else if (tid >= 64 && tid < 64 + m) { int a = tid - 64; double s = 0; for (int t = 0; t < p; t++) s += CM(S.D, t, a, p) * S.ly[t]; S.Qu[a] += s; })
        }
        // regularisation (also on Qxx: quirk x), store Qu/Quu/Qux, shifted Cholesky of Quu
        HS_PHASE(NT,
            if (tid < n) CM(S.Qxx, tid, tid, n) += reg;
            if (tid < m) { CM(S.Quu, tid, tid, m) += reg; P.Qu[kk * m + tid] = S.Qu[tid]; })
        HS_PHASE(NT,
            for (int i = tid; i < m * m; i += NT) P.Quu[kk * m * m + i] = S.Quu[i];
            for (int i = tid; i < m * n; i += NT) P.Qux[kk * m * n + i] = S.Qux[i];)
        for (int j = 0; j < m; j++) {   // left-looking Cholesky of (Quu - 1e-9 I), row-major LQ
            HS_PHASE(NT, if (tid >= j && tid < m) {
                double s = CM(S.Quu, tid, j, m) - ((tid == j) ? 1e-9 : 0.0);
                for (int t = 0; t < j; t++) s -= S.LQ[tid * m + t] * S.LQ[j * m + t];
                S.tmp[tid] = s;
            })
            HS_PHASE(NT, if (tid >= j && tid < m) {
                double piv = S.tmp[j];
                if (tid == j && !(piv > 0.0)) S.ok = 0;
                double d = sqrt(piv > 0.0 ? piv : 1.0);
                S.LQ[tid * m + j] = (tid == j) ? d : S.tmp[tid] / d;
            })
        }
        if (!S.ok) return false;
        HS_PHASE(NT, if (tid < m) {   // Quu_inv column tid (row-major Qi, symmetric)
            const int c = tid;
            for (int i = 0; i < m; i++) { double s = (i == c) ? 1.0 : 0.0; for (int t = 0; t < i; t++) s -= S.LQ[i * m + t] * S.Qi[t * m + c]; S.Qi[i * m + c] = s / S.LQ[i * m + i]; }
            for (int i = m - 1; i >= 0; i--) { double s = S.Qi[i * m + c]; for (int t = i + 1; t < m; t++) s -= S.LQ[t * m + i] * S.Qi[t * m + c]; S.Qi[i * m + c] = s / S.LQ[i * m + i]; }
        })
        // symmetrise Qxx ; K = -Qi Qux ; dU = -Qi Qu
        HS_PHASE(NT,
            for (int e = tid; e < n * n; e += NT) { int i = e % n, j = e / n; if (i < j) { double s = (CM(S.Qxx, i, j, n) + CM(S.Qxx, j, i, n)) / 2; CM(S.Qxx, i, j, n) = s; CM(S.Qxx, j, i, n) = s; } }
            for (int e = tid; e < m * n; e += NT) { int a = e % m, j = e / m; double s = 0; for (int t = 0; t < m; t++) s += S.Qi[a * m + t] * CM(S.Qux, t, j, m); S.K[e] = -s; }
            if (tid < m) { double s = 0; for (int t = 0; t < m; t++) s += S.Qi[tid * m + t] * S.Qu[t]; S.dU[tid] = -s; })
        // G = Qx + Qux^T dU ; H = Qxx + Qux^T K ; dV
        HS_PHASE(NT,
            for (int e = tid; e < n * n; e += NT) { int i = e % n, j = e / n; double s = CM(S.Qxx, i, j, n); for (int t = 0; t < m; t++) s += CM(S.Qux, t, i, m) * CM(S.K, t, j, m); S.H[e] = s; }
            if (tid < n) { double s = S.Qx[tid]; for (int t = 0; t < m; t++) s += CM(S.Qux, t, tid, m) * S.dU[t]; S.G[tid] = s; P.G[((size_t)b * (h + 1) + k) * n + tid] = s; }
            if (tid == 64) { double dVk = 0; for (int t = 0; t < m; t++) dVk -= S.Qu[t] * S.dU[t]; S.dV1 -= dVk; S.dV2 += dVk; }
            for (int i = tid; i < m * n; i += NT) P.K[kk * m * n + i] = S.K[i];
            if (tid >= 128 && tid < 128 + m) P.dU[kk * m + tid - 128] = S.dU[tid - 128];)
    }
    // G[0] += H[0] * Defect[0]   (SinglePhase.cpp:389)
    HS_PHASE(NT, if (tid < n) S.def[tid] = P.Defect[((size_t)b * (h + 1)) * n + tid];)
    HS_PHASE(NT, if (tid < n) { double s = S.G[tid]; for (int j = 0; j < n; j++) s += CM(S.H, tid, j, n) * S.def[j]; S.Gn[tid] = s; })
    HS_PHASE(NT, if (tid < n) { S.G[tid] = S.Gn[tid]; P.G[((size_t)b * (h + 1)) * n + tid] = S.Gn[tid]; }
             for (int i = tid; i < n * n; i += NT) P.H0[(size_t)b * n * n + i] = S.H[i];)
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
            HS_PHASE(NT, for (int e = tid; e < n * n; e += NT) S.H[e] = 0.0; if (tid < n) S.G[tid] = 0.0;)
        } else {   // impact-aware step: (G,H) <- (Px^T G, Px^T H Px), Px = next_n x n  (MultiPhaseDDP.cpp:196-201)
            const int nn = P.next_n;
            ld_block<NT>(S.A, P.Px + (size_t)b * nn * n, nn * n);
            HS_PHASE(NT,
                for (int e = tid; e < nn * n; e += NT) { int i2 = e % nn, j = e / nn; double s = 0; for (int t = 0; t < nn; t++) s += CM(S.H, i2, t, nn) * CM(S.A, t, j, nn); S.HA[e] = s; }
                if (tid < n) { double s = 0; for (int t = 0; t < nn; t++) s += CM(S.A, t, tid, nn) * S.G[t]; S.Gn[tid] = s; })
            HS_PHASE(NT,
                for (int e = tid; e < n * n; e += NT) { int i2 = e % n, j = e / n; double s = 0; for (int t = 0; t < nn; t++) s += CM(S.A, t, i2, nn) * CM(S.HA, t, j, nn); S.Qxx[e] = s; })
            HS_PHASE(NT, for (int e = tid; e < n * n; e += NT) S.H[e] = S.Qxx[e]; if (tid < n) S.G[tid] = S.Gn[tid];)
        }
        if (!riccati_phase<NT>(S, P, b, reg)) return false;
    }
    return true;
}

// linear rollout of problem b (eps = 1 in solve).  Returns dV_1, dV_2 in S.dV1/dV2.
template <int NT>
HD void linear_rollout(SweepLds& S, const PhaseDev* ph, int nph, int b, double eps) {
    HS_PHASE(NT, if (tid == 0) { S.dV1 = 0.0; S.dV2 = 0.0; } if (tid < MAXN) S.dxn[tid] = 0.0;)
    for (int i = 0; i < nph; i++) {
        const PhaseDev& P = ph[i];
        const int n = P.n, m = P.m, h = P.h;
        if (i > 0) {   // dx_init = Px * dX_end(prev)   (MultiPhaseDDP.cpp:27-30); S.dx holds prev terminal dX
            const PhaseDev& Pp = ph[i - 1]; const int np = Pp.n;
            ld_block<NT>(S.A, Pp.Px + (size_t)b * n * np, n * np);
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
        ld_block<NT>(S.Qxx, P.Phixx + (size_t)b * n * n, n * n);
        HS_PHASE(NT, if (tid < n) { double q = 0; for (int j = 0; j < n; j++) q += CM(S.Qxx, tid, j, n) * S.dx[j]; S.red[tid] = S.dx[tid] * q; S.red[64 + tid] = P.Phix[(size_t)b * n + tid] * S.dx[tid]; })
        HS_PHASE(NT, if (tid == 0) { double a1 = 0, a2 = 0; for (int j = 0; j < n; j++) { a1 += S.red[64 + j]; a2 += S.red[j]; } S.dV1 += a1; S.dV2 += a2; })
    }
}

}  // namespace hs
