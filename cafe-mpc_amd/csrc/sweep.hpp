// Backward Riccati sweep and linear rollout: ONE WORKGROUP PER PROBLEM, sequential over knots and phases,
// value function (H, G) and the knot's A, B, C, D held in LDS for the whole sweep.
//
//   riccati_sweep   replaces MultiPhaseDDP::backward_sweep (MultiPhaseDDP.cpp:174-213), impact_aware_step (:499-503)
//                   and SinglePhase::backward_sweep (SinglePhase.cpp:323-391)
//   (k_sweep)       MultiPhaseDDP::backward_sweep_regularized (MultiPhaseDDP.cpp:136-165) wraps it in the retry loop
//   linear_rollout  MultiPhaseDDP::linear_rollout (MultiPhaseDDP.cpp:12-42) + SinglePhase::linear_rollout (SinglePhase.cpp:145-178)
// Quu^-1: Eigen's pivoted LDLT of (Quu + reg I - 1e-9 I), PD test = no negative pivot, Quu_inv = LDLT.solve(I), then
// K = -Quu_inv Qux, dU = -Quu_inv Qu exactly as SinglePhase.cpp:366-380 does (ldlt_inverse_w below restates
// Eigen/src/Cholesky/LDLT.h ldlt_inplace<Lower>::unblocked + _solve_impl for one wave).
// lux is identically zero for every cost the reference ships (SinglePhaseInterface.cpp:47, MHPCCost.cpp) and is not stored.
//
// MI355X structure (dims are template constants: whole body 36/12/12):
//  * LDS: every matrix column-major with an ODD padded leading dimension (37 / 13): the 16x16 operand tiles of a wave then hit
//    distinct bank pairs also in the transposed products (A^T HA, B^T HB, C^T lyy C ...).
//  * every matrix product of a knot runs on the fp64 matrix cores (v_mfma_f64_16x16x4_f64), 16x16 output tiles dealt over the
//    four waves of the workgroup (hs_mfma.hpp); the mat-vec chains ride on the lanes of the waves that carry fewer tiles.
//  * the NEXT knot's record (A, lxx, B, C, D, luu, lyy, vectors) is prefetched from HBM into 20 registers per thread
//    while the current knot computes, and committed to LDS at the top of the next iteration: HBM latency is off the
//    sequential critical path.
//  * the 12x12 (24x24) LDLT + inverse runs inside wave 0: rows in registers, pivots and multipliers by lane broadcast.
#pragma once
#include <cstddef>
#include <limits>
#include <type_traits>
#include "hs_types.hpp"
#include "hs_mfma.hpp"

namespace hs {

constexpr int SW_NT = 256;
constexpr int SW_SET_WB = 0, SW_SET_HKD = 1;      // model sets a sweep kernel is instantiated for: {whole body, SRB} / {kinodynamic, SRB}
constexpr int SW_N = 36;    // largest state dimension of any model
constexpr int SW_PRE = 20;  // prefetch registers per thread
#ifndef SW_UNROLL_N
#define SW_UNROLL_N 4
#endif
#define HS_STR_(x) #x
#define HS_UNROLL_N(n) _Pragma(HS_STR_(unroll n))
#ifndef SW_LDLT_NB
#define SW_LDLT_NB ((M <= 12 || sizeof(R) == 4) ? 2 : 1)
#endif
#ifndef SW_MINB
#define SW_MINB 2
#endif
#ifndef SW_LDLT_WDW
#define SW_LDLT_WDW 1          // the explicit inverse as W^T D^-1 W on the matrix cores (W = L^-1 P) instead of the backward solves
#endif
#ifndef SW_OVERLAP_LDLT
#define SW_OVERLAP_LDLT 1      // the factorisation of Quu (wave 0) side by side with the Qxx / Qux tiles (waves 1..3): see riccati_phase
#endif

// LDS working set of one Riccati step for a model with dims (N, M, PY); leading dimensions padded to rows + 1
// (conflict-free column access).  All three instantiations are views over the same raw LDS block (SweepLds).
// R: the scalar the sweep computes in - double, or float for the fp32 handles of hsddp_create_ex (fp32 LQ records, v_mfma_f32_16x16x4_f32).
// Two views over one raw LDS block.  Riccati step: H (value-function Hessian of knot k+1, then Qxx, then H of knot k, all in place), the stored
// rows of A and B, HA / HB, Qux, C, D, lyy, Quu; K shares storage with lC and the (negated) inverse of Quu with lD (dead before they are born).
// lxx never sits in LDS: it goes from the prefetch registers straight into H (see riccati_phase).  Whole body: 6 264 doubles = 50.1 KB, three
// workgroups per CU.
template <int N, int M, int PY, class R = double> struct SweepLdsT {
    static constexpr int LDN = N + 1, LDM = (M > PY ? M : PY) + 1, PYd = PY > 0 ? PY : 1;
    static constexpr int AR = rec_arows(N), LDA = AR + 1, A0 = N - AR;      // A, B: the stored (lower AR) rows, i.e. rows A0.. of the full matrices
    R H[LDN * N], A[LDA * N], HA[LDN * N];
    R B[LDA * M], HB[LDN * M];
    R Qux[LDM * N];
    union { R K[LDM * N]; R lC[PY > 0 ? LDM * N : 1]; };
    R C[PY > 0 ? LDM * N : 1], D[PY > 0 ? LDM * M : 1];
    union { R LQ[LDM * M]; R lD[PY > 0 ? LDM * M : 1]; };
    R lyy[PY > 0 ? LDM * PY : 1], Quu[LDM * M];
    R G[N], Gn[N], Qx[N], Qu[M], dU[M], ly[PYd], def[N];
};
// Linear rollout: two sets of (stored rows of A, lxx, stored rows of B, K, luu, vectors) alternate between knots, dense leading dimensions
template <int N, int M, int PY, class R = double> struct LinLdsT {
    static constexpr int AR = rec_arows(N), A0 = N - AR;
    R A[2][AR * N], Q[2][N * N], B[2][AR * M], K[2][M * N], U[2][M * M], v[2][256];     // v: lx | lu at 64 | dU at 128 | Defect[k+1] at 192
    R dx[N], dxn[N], du[M];
};
// what survives a phase boundary: value-function gradient / state deviation handed to the neighbouring phase, dV, status
struct SweepCtl { double dV1, dV2, xfer[SW_N]; unsigned long long t_last; int ok; };
constexpr size_t sw_max(size_t a, size_t b) { return a > b ? a : b; }
template <class R> constexpr size_t sweep_bytes_wb() { return sw_max(sw_max(sizeof(SweepLdsT<36, 12, 12, R>), sizeof(LinLdsT<36, 12, 12, R>)), sw_max(sizeof(SweepLdsT<12, 12, 0, R>), sizeof(LinLdsT<12, 12, 0, R>))); }
template <class R> constexpr size_t sweep_bytes_hkd() { return sw_max(sw_max(sizeof(SweepLdsT<24, 24, 0, R>), sizeof(LinLdsT<24, 24, 0, R>)), sw_max(sizeof(SweepLdsT<12, 12, 0, R>), sizeof(LinLdsT<12, 12, 0, R>))); }
// one raw block per (scalar type, model set); every view of a set fits its block
template <class R, int SET> struct SweepLdsRaw { R raw[(SET == 0 ? sweep_bytes_wb<R>() : sweep_bytes_hkd<R>()) / sizeof(R)]; SweepCtl c; };
using SweepLds = SweepLdsRaw<double, 0>;       // whole-body + SRB phases, fp64
using SweepLdsHkd = SweepLdsRaw<double, 1>;    // kinodynamic + SRB phases, fp64
using SweepLds32 = SweepLdsRaw<float, 1>;      // kinodynamic + SRB phases of the fp32 handles

#define CM(M, i, j, ld) (M)[(i) + (ld) * (j)]

// optional in-kernel stamps (diagnostic builds only: -DSW_PROF): cycles per phase group, block 0 / thread 0
#if defined(SW_PROF) && !defined(HS_HOST_EMU)
__device__ unsigned long long g_sw_prof[48];      // [0,16): phase stamps of thread 0 ; [16 + 4 p + w]: wave w's own time in MFMA phase p (start of the phase to its arrival at the barrier)
#define SW_STAMP(i) { if (blockIdx.x == 0 && threadIdx.x == 0) { unsigned long long t_ = clock64(); atomicAdd(&g_sw_prof[i], t_ - SWC.t_last); SWC.t_last = t_; } }
#define SW_STAMP0() { if (blockIdx.x == 0 && threadIdx.x == 0) SWC.t_last = clock64(); }
#define SW_WSTAMP(p) { if (blockIdx.x == 0 && (tid & 63) == 0) atomicAdd(&g_sw_prof[16 + 4 * (p) + (tid >> 6)], clock64() - SWC.t_last); }
#else
#define SW_WSTAMP(p)
#define SW_STAMP(i)
#define SW_STAMP0()
#endif

// per-thread prefetch registers (the host lane emulator keeps one row per emulated thread)
#ifdef HS_HOST_EMU
#define SW_PRE_DECL static R pre_all_[SW_NT][SW_PRE];
#define PRE(r) pre_all_[tid][r]
#else
#define SW_PRE_DECL R pre_[SW_PRE];
#define PRE(r) pre_[r]
#endif

// Sum over the 4 lanes of an aligned quad in the order (p0 + p1) + (p2 + p3), by two DPP quad permutes (no LDS traffic); every
// lane of the quad receives the sum.  The linear rollout is a chain of mat-vec products: splitting each row over a quad cuts the
// dependent multiply-add chain of a knot by four.
#ifndef HS_HOST_EMU
template <int CTRL> HD double dpp_quad(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true), hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
HD double quad_sum(double v) { v += dpp_quad<0xB1>(v); v += dpp_quad<0x4E>(v); return v; }     // quad_perm [1,0,3,2], then [2,3,0,1]
template <int CTRL> HD float dpp_quad(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true)); }
HD float quad_sum(float v) { v += dpp_quad<0xB1>(v); v += dpp_quad<0x4E>(v); return v; }
#endif
// SW_QUAD_ROWS(CNT, PARTIAL, FINISH): for each output o < CNT, PARTIAL computes `partial` and `partial2` from (o, part = 0..3), FINISH
// consumes `total` / `total2` = (p0 + p1) + (p2 + p3).  GPU: the four lanes of quad o (tid = 4 o + part), FINISH on the part-0 lane;
// emulator: thread o does all four parts itself.  SW_QS: lane stride between the lanes that run FINISH.
#ifdef HS_HOST_EMU
#define SW_QS 1
#define SW_QUAD_ROWS(CNT, PARTIAL, FINISH) if (tid < (CNT)) { const int o = tid; R p_[4]; R p2_[4]; for (int part = 0; part < 4; part++) { R partial = 0; R partial2 = 0; PARTIAL p_[part] = partial; p2_[part] = partial2; } \
    const R total = (p_[0] + p_[1]) + (p_[2] + p_[3]); const R total2 = (p2_[0] + p2_[1]) + (p2_[2] + p2_[3]); (void)total2; FINISH }
#else
#define SW_QS 4
#define SW_QUAD_ROWS(CNT, PARTIAL, FINISH) if ((tid >> 2) < (CNT)) { const int o = tid >> 2; const int part = tid & 3; R partial = 0; R partial2 = 0; PARTIAL \
    const R total = quad_sum(partial); const R total2 = quad_sum(partial2); (void)total2; if (part == 0) FINISH }
#endif

// ---- wave 0: Eigen's pivoted LDLT and the explicit inverse ------------------------------------------------------------------
// What the reference does per knot (SinglePhase.cpp:366-375): Eigen::LDLT<DMat>::compute(Quu - 1e-9 I), isPositive(), solve(I).
// Eigen 3.3's ldlt_inplace<Lower>::unblocked (third-party, restated from its published source; the CPU checker holds the same
// restatement):
//   step k: the pivot is the FIRST largest |diagonal| among rows k.. of the working matrix - and since a left-looking step only
//   ever rewrites column k, those diagonals are still the ORIGINAL ones: the pivot order is a selection sort of the original
//   |diagonal| (with Eigen's swap dynamics deciding exact ties);  symmetric swap;  temp_j = D_j L(k,j);  A(k,k) -= L(k,:) temp;
//   A(k+1:,k) = (A(k+1:,k) - L(k+1:,:) temp) / A(k,k);  sign bookkeeping.  Only the lower triangle of the input is read.
//   solve: x = P b; unit-lower forward substitution; x_i = |D_i| > 1/DBL_MAX ? x_i / D_i : 0; backward substitution; x = P^T x.
// Wave mapping: lane i owns ROW i of the PERMUTED matrix.  The pivot order comes first (each lane ranks its own diagonal entry; an
// exact tie - symmetric problems - sends the wave through the literal selection sort instead).  Then the left-looking steps with
// the row's multipliers in registers: L(k,j) and the pivot travel by constant-lane broadcasts (v_readlane), every entry sees the
// same sequence of multiply-adds as in Eigen's loops.  The multipliers go to LDS once, and lanes 0..M-1 each solve one column of
// the identity.  Output: NI = -(A + diag_add I)^-1 (the sign the gains need), column-major with leading dimension LD.
// Scratch: Lw >= M*M doubles, iw >= M ints.
// PART: 0 = the whole routine as ONE wave-level phase of wave 0 (ldlt_inverse_w); 1 = pivot order + factorisation, 2 = the solves - called by the
// lanes of wave 0 from inside two consecutive workgroup phases (SW_OVERLAP_LDLT: the other waves form the Qxx / Qux tiles and symmetrise Qxx
// meanwhile); the factor (Lw) and the pivot order (iw) cross from part 1 to part 2 through LDS behind the workgroup barrier between them.
// W1 / W2 (PART 2 with SW_LDLT_WDW, >= M*M each): where the columns of W = L^-1 P and of -D^-1 W are laid for the matrix-core product NI = W^T (-D^-1 W)
template <int M, int LD, class R, int PART>
HD void ldlt_parts(int tid, const R* A, R diag_add, R* NI, R* Lw, int* iw, int* ok, R* W1 = nullptr, R* W2 = nullptr) {
    const R TOL = R(1.0) / std::numeric_limits<R>::max();
#ifdef HS_HOST_EMU
    if (PART == 2) return;         // (the emulator's part 1 is the whole routine)
    if (tid == 0) {      // the emulator has no lanes to broadcast between: Eigen's loops as they stand, with physical swaps
        R m[M * M]; int tr[M]; int sign = 0; (void)Lw; (void)iw;
        for (int j = 0; j < M; j++) for (int i = 0; i < M; i++) m[i + M * j] = A[i + LD * j] + (i == j ? diag_add : 0.0);
        auto Mx = [&](int i, int j) -> R& { return m[i + M * j]; };
        R temp[M];
        for (int k = 0; k < M; k++) {
            int big = k; R best = std::fabs(Mx(k, k));
            for (int i = k + 1; i < M; i++) if (std::fabs(Mx(i, i)) > best) { best = std::fabs(Mx(i, i)); big = i; }
            tr[k] = big;
            if (k != big) {
                for (int j = 0; j < k; j++) { const R t = Mx(k, j); Mx(k, j) = Mx(big, j); Mx(big, j) = t; }
                for (int i = big + 1; i < M; i++) { const R t = Mx(i, k); Mx(i, k) = Mx(i, big); Mx(i, big) = t; }
                { const R t = Mx(k, k); Mx(k, k) = Mx(big, big); Mx(big, big) = t; }
                for (int i = k + 1; i < big; i++) { const R t = Mx(i, k); Mx(i, k) = Mx(big, i); Mx(big, i) = t; }
            }
            for (int j = 0; j < k; j++) temp[j] = Mx(j, j) * Mx(k, j);
            { R s = 0; for (int j = 0; j < k; j++) s += Mx(k, j) * temp[j]; Mx(k, k) -= s; }
            for (int i = k + 1; i < M; i++) { R t = 0; for (int j = 0; j < k; j++) t += Mx(i, j) * temp[j]; Mx(i, k) -= t; }
            const R akk = Mx(k, k); const bool valid = std::fabs(akk) > 0.0;
            if (valid) for (int i = k + 1; i < M; i++) Mx(i, k) /= akk;
            if (sign == 1) { if (akk < 0) sign = 2; } else if (sign == -1) { if (akk > 0) sign = 2; } else if (sign == 0) { if (akk > 0) sign = 1; else if (akk < 0) sign = -1; }
        }
        if (!(sign == 1 || sign == 0)) *ok = 0;
        for (int c = 0; c < M; c++) {
            R x[M]; for (int i = 0; i < M; i++) x[i] = (i == c) ? 1.0 : 0.0;
            for (int k = 0; k < M; k++) if (tr[k] != k) { const R t = x[k]; x[k] = x[tr[k]]; x[tr[k]] = t; }
            for (int i = 0; i < M; i++) { R s = x[i]; for (int j = 0; j < i; j++) s -= Mx(i, j) * x[j]; x[i] = s; }
            for (int i = 0; i < M; i++) { const R d = Mx(i, i); x[i] = (std::fabs(d) > TOL) ? x[i] / d : 0.0; }
            for (int i = M - 1; i >= 0; i--) { R s = x[i]; for (int j = i + 1; j < M; j++) s -= Mx(j, i) * x[j]; x[i] = s; }
            for (int k = M - 1; k >= 0; k--) if (tr[k] != k) { const R t = x[k]; x[k] = x[tr[k]]; x[tr[k]] = t; }
            for (int i = 0; i < M; i++) NI[i + LD * c] = -x[i];
        }
    }
#else
    static_assert(M <= 32, "one row per lane");
#if defined(SW_PROF)
#define SW_LSTAMP(i) { if (blockIdx.x == 0 && tid == 0) { const unsigned long long t_ = clock64(); atomicAdd(&g_sw_prof[i], t_ - tl_); tl_ = t_; } }
#else
#define SW_LSTAMP(i)
#endif
    {
#if defined(SW_PROF)
        unsigned long long tl_ = clock64();
#endif
        const bool act = tid < M; const int me = act ? tid : M - 1;       // idle lanes mirror the last row (all 64 lanes run the broadcasts)
        if (PART != 2) {
        // ---- pivot order
        R dg[M];
        _Pragma("unroll") for (int j = 0; j < M; j++) dg[j] = fabs(A[j + LD * j] + diag_add);
        const R mine = fabs(A[me + LD * me] + diag_add);
        int rank = 0; bool tie = false;
        _Pragma("unroll") for (int j = 0; j < M; j++) { const bool eq = (dg[j] == mine) && (j != me); rank += (dg[j] > mine || (eq && j < me)) ? 1 : 0; tie = tie || eq; }
        if (__builtin_amdgcn_ballot_w64(tie && act) == 0ull) { if (act) iw[rank] = me; }
        else {      // exact ties (symmetric problems: every kinodynamic knot): Eigen's selection with swaps, literally.  Lane p holds the entry at POSITION p
            // of the working diagonal as (original index, sorted positions [gt, ge) its value class occupies): the maxima are taken class by class, so
            // at step k the largest remaining value is the class with gt <= k < ge - no reduction; its FIRST position >= k by ballot, swap with k
            int gt = 0, ge = 0;
            _Pragma("unroll") for (int j = 0; j < M; j++) { gt += (dg[j] > mine) ? 1 : 0; ge += (dg[j] >= mine) ? 1 : 0; }
            int ixp = me; if (!act) { gt = M; ge = M; }
            _Pragma("unroll") for (int k = 0; k < M; k++) {
                const unsigned long long at = __builtin_amdgcn_ballot_w64(tid >= k && gt <= k && k < ge);
                const int big = (int)__builtin_ctzll(at);
                if (big != k) {      // (uniform: a position that already holds an element of the current class stays as it is)
                    const int ik = __builtin_amdgcn_readlane(ixp, k), gk = __builtin_amdgcn_readlane(gt, k), ek = __builtin_amdgcn_readlane(ge, k);
                    const int ib = __builtin_amdgcn_readlane(ixp, big), gb = __builtin_amdgcn_readlane(gt, big), eb = __builtin_amdgcn_readlane(ge, big);
                    ixp = (tid == k) ? ib : ((tid == big) ? ik : ixp); gt = (tid == k) ? gb : ((tid == big) ? gk : gt); ge = (tid == k) ? eb : ((tid == big) ? ek : ge);
                }
            }
            if (act) iw[tid] = ixp;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        SW_LSTAMP(9)
        }
        int pv[M]; int myrank = 0;
        _Pragma("unroll") for (int k = 0; k < M; k++) { pv[k] = iw[k]; myrank = (pv[k] == me) ? k : myrank; }
        if (PART != 2) {
        const int prow = iw[me];
        // ---- row `me` of the permuted matrix (lower triangle of the input only)
        R arow[M];
        _Pragma("unroll") for (int j = 0; j < M; j++) { const int r = prow > pv[j] ? prow : pv[j], c = prow > pv[j] ? pv[j] : prow; arow[j] = A[r + LD * c] + ((j == me) ? diag_add : 0.0); }
        // ---- LDL^T of the permuted matrix, right-looking with the row in registers (the structure of chol_r): at step j every lane holds
        //      its column-j entry w = D_j L(me,j) of the current Schur complement in a[j]; the pivot D_j = w of lane j and the entries of
        //      the other rows travel by constant-lane broadcasts: a[k] -= L(me,j) * (D_j L(k,j)).  Same pivots and multipliers as Eigen's
        //      left-looking loops (A(k,k) -= L temp, A21 = (A21 - A20 temp) / A(k,k)) up to the association of the partial sums; quotients
        //      by the pivot as multiplications by its reciprocal.  Multipliers and reciprocal pivots go straight to LDS (row me of Lw).
        SW_LSTAMP(10)
        bool anyneg = false;
        _Pragma("unroll") for (int j = 0; j < M; j++) {
            const R d = hs_readlane(arow[j], j);
            R rd = hs_rcp(d); rd = rd * (R(2.0) - d * rd); rd = rd * (R(2.0) - d * rd);     // 1/d to the last bit or two (the IEEE division sequence is five times as long and sits on the chain from pivot to pivot)
            rd = (fabs(d) > TOL) ? rd : 0.0;
            anyneg = anyneg || (d < 0.0);
            const R lij = (fabs(d) > 0.0) ? arow[j] * rd : arow[j];
            // multiplier, or the reciprocal pivot on the diagonal.  One predicated store for the 12 x 12 blocks (k_sweep 16.5 -> 16.2 ms); the 24 x 24
            // blocks of the kinodynamic model measure better with the two-branch form (k_sweep32 418.8 against 441.6 ms per five launches)
            if (M <= 12) { if (act && me >= j) Lw[me * M + j] = (me == j) ? rd : lij; }
            else if (act) { if (me > j) Lw[me * M + j] = lij; else if (me == j) Lw[j * M + j] = rd; }
            _Pragma("unroll") for (int k = j + 1; k < M; k++) arow[k] -= lij * hs_readlane(arow[j], k);
        }
        SW_LSTAMP(11)
        if (tid == 0 && anyneg) *ok = 0;       // isPositive(): no negative pivot (Eigen's sign bookkeeping ends in PositiveSemiDef / ZeroSign exactly then)
        }
        if (PART == 1) return;
        // ---- column `me` of the inverse: P e_me is the unit vector at position myrank.  The factor row of the next step is fetched
        //      (broadcast reads) while the current row's multiply-add chain runs: a single wave has nothing else to hide the LDS latency behind.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        constexpr int NB = SW_LDLT_NB;      // (the 24-row factor of the kinodynamic model leaves no registers for a second row buffer)
        R y[M], lr[NB][M];
        if (NB == 1) { lr[0][0] = 0.0; }
        _Pragma("unroll") for (int k = 0; k < M; k++) {
            if (NB == 2) { if (k + 1 < M) { _Pragma("unroll") for (int j = 0; j <= k; j++) lr[(k + 1) % NB][j] = Lw[(k + 1) * M + j]; } }
            else { _Pragma("unroll") for (int j = 0; j < k; j++) lr[0][j] = Lw[k * M + j]; }
            HS_CBAR();
            R sacc = (k == myrank) ? 1.0 : 0.0;
            _Pragma("unroll") for (int j = 0; j < k; j++) sacc -= lr[k % NB][j] * y[j];
            HS_PIN(sacc);       // (the row's chain is evaluated HERE: otherwise every row's loads are issued first and their values spill)
            y[k] = sacc;
        }
        SW_LSTAMP(12)
#if SW_LDLT_WDW
        if (PART == 2 && sizeof(R) == 8) {      // (fp64 only: in fp32 the product form loses a digit against the substitutions - measured on the kinodynamic trot: |dU| error 5e-3 against 5e-4)
            // Quu^-1 = P^T L^-T D^-1 L^-1 P = W^T D^-1 W with W = L^-1 P, whose column `me` this lane has just formed: instead of M^2/2 more dependent
            // multiply-adds per lane (the backward solves) the columns go to LDS once and ONE matrix-core product per 16 x 16 tile forms the (negated)
            // inverse.  Same factors and pivots as LDLT.solve(I) (SinglePhase.cpp:375); the sums run over k in the matrix cores' order.
            if (act) { _Pragma("unroll") for (int k = 0; k < M; k++) { W1[k + M * me] = y[k]; W2[k + M * me] = -(y[k] * Lw[k * M + k]); } }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            constexpr int TMq = (M + 15) / 16;
            MTileT<R> td[TMq * TMq];
            _Pragma("unroll") for (int t = 0; t < TMq * TMq; t++) td[t] = MTileT<R>{NI, LD, nullptr, 0, 16 * (t % TMq), 16 * (t / TMq), M, M, W1, M, W2, M, M, true, nullptr, 0, nullptr, 0, 0};
            mfma_tiles<TMq * TMq, (M + 3) / 4 * 4, 0, R>(tid, td);
            SW_LSTAMP(13)
            return;
        }
#endif
        _Pragma("unroll") for (int k = 0; k < M; k++) y[k] = y[k] * Lw[k * M + k];
        _Pragma("unroll") for (int k = M - 1; k >= 0; k--) {
            if (NB == 2) { if (k > 0) { _Pragma("unroll") for (int j = k; j < M; j++) lr[(k - 1) % NB][j] = Lw[j * M + (k - 1)]; } }
            else { _Pragma("unroll") for (int j = k + 1; j < M; j++) lr[0][j] = Lw[j * M + k]; }
            HS_CBAR();
            R sacc = y[k];
            _Pragma("unroll") for (int j = k + 1; j < M; j++) sacc -= lr[k % NB][j] * y[j];
            HS_PIN(sacc);
            y[k] = sacc;
        }
        if (act) { _Pragma("unroll") for (int k = 0; k < M; k++) NI[pv[k] + LD * me] = -y[k]; }
        SW_LSTAMP(13)
    }
#endif
}
template <int M, int LD, class R>
HD void ldlt_inverse_w(const R* A, R diag_add, R* NI, R* Lw, int* iw, int* ok) {
    HS_WPHASE(ldlt_parts<M, LD, R, 0>(tid, A, diag_add, NI, Lw, iw, ok);)
}
// global (dense, ld = rows) <-> LDS (padded ld); threads stride over columns with a fixed row
template <int NT, class TD, class TS> HD void ld_mat(int tid, TD* dst, int ldd, const TS* src, int rows, int cols) {
    const int i = tid % rows, j0 = tid / rows, js = NT / rows;
    if (j0 < js) for (int j = j0; j < cols; j += js) dst[i + ldd * j] = src[i + rows * j];
}
template <int NT, class TD, class TS> HD void st_mat(int tid, TD* dst, const TS* src, int lds_, int rows, int cols) {
    const int i = tid % rows, j0 = tid / rows, js = NT / rows;
    if (j0 < js) for (int j = j0; j < cols; j += js) dst[i + rows * j] = src[i + lds_ * j];
}

// ---- prefetch of one knot's record (backward sweep): RL::rounds rounds of 256 elements, each inside ONE sub-array,
//      plus one round for the vectors [lx(N) lu(M) ly(PY) Defect[k+1](N)]
// lxx (rounds rA .. rA + rQ - 1) travels on its own: it is added to Qxx straight from the registers AFTER the products (SW_RICCATI_LXX), so the
// Riccati step needs no LDS buffer for it; its next fetch is issued right there.
#define SW_RICCATI_FETCH(kk_, k_) { \
    const HS_GLOBAL R* rec_ = grec + (kk_) * (size_t)RL::size + tid; \
    _Pragma("unroll") for (int r = 0; r < RL::rounds; r++) if (r < RL::rA || r >= RL::rA + RL::rQ) PRE(r) = rec_[NT * r];     /* one base pointer, constant offsets */ \
    PRE(RL::rounds) = (tid < N + M + PY) ? rec_[RL::oLx] : (tid < 2 * N + M + PY) ? (R)gDefect[((size_t)b * (h + 1) + (k_) + 1) * N + tid - N - M - PY] : R(0.0); }
#define SW_RICCATI_FETCH_LXX(kk_) { \
    const HS_GLOBAL R* rec_ = grec + (kk_) * (size_t)RL::size + tid; \
    _Pragma("unroll") for (int r = RL::rA; r < RL::rA + RL::rQ; r++) PRE(r) = rec_[NT * r]; }
#define SW_RICCATI_LXX() { \
    _Pragma("unroll") for (int r = 0; r < RL::rQ; r++) { const int e = tid + NT * r; if (e < N * N) S.H[(e % N) + LDN * (e / N)] += PRE(RL::rA + r); } }
#define SW_RICCATI_COMMIT() { \
    constexpr int PYd = PY > 0 ? PY : 1; \
    _Pragma("unroll") for (int r = 0; r < RL::rA; r++) { const int e = tid + NT * r; if (e < AR * N) S.A[(e % AR) + LDA * (e / AR)] = PRE(r); } \
    _Pragma("unroll") for (int r = 0; r < RL::rB; r++) { const int e = tid + NT * r; if (e < AR * M) S.B[(e % AR) + LDA * (e / AR)] = PRE(RL::rA + RL::rQ + r); } \
    _Pragma("unroll") for (int r = 0; r < RL::rC; r++) { const int e = tid + NT * r; if (e < PY * N) S.C[(e % PYd) + LDM * (e / PYd)] = PRE(RL::rA + RL::rQ + RL::rB + r); } \
    _Pragma("unroll") for (int r = 0; r < RL::rD; r++) { const int e = tid + NT * r; if (e < PY * M) S.D[(e % PYd) + LDM * (e / PYd)] = PRE(RL::rA + RL::rQ + RL::rB + RL::rC + r); } \
    _Pragma("unroll") for (int r = 0; r < RL::rLuu; r++) { const int e = tid + NT * r; if (e < M * M) S.Quu[(e % M) + LDM * (e / M)] = PRE(RL::rA + RL::rQ + RL::rB + RL::rC + RL::rD + r); } \
    _Pragma("unroll") for (int r = 0; r < RL::rLyy; r++) { const int e = tid + NT * r; if (e < PY * PY) S.lyy[(e % PYd) + LDM * (e / PYd)] = PRE(RL::rA + RL::rQ + RL::rB + RL::rC + RL::rD + RL::rLuu + r); } \
    if (tid < N) S.Qx[tid] = PRE(RL::rounds); else if (tid < N + M) S.Qu[tid - N] = PRE(RL::rounds); else if (tid < N + M + PY) S.ly[tid - N - M] = PRE(RL::rounds); \
    else if (tid < 2 * N + M + PY) S.def[tid - N - M - PY] = PRE(RL::rounds); }

// the LQ records of a phase in the precision the sweep computes in
template <class R> HD const HS_GLOBAL R* rec_of(const PhaseDev& P);
template <> HD const HS_GLOBAL double* rec_of<double>(const PhaseDev& P) { return P.rec; }
template <> HD const HS_GLOBAL float* rec_of<float>(const PhaseDev& P) { return P.rec32; }

// Dealing order of a tile list: the full 16 x 16 tiles first, the edge tiles (hs_mfma.hpp: a quarter of the matrix-core time) after them, so that a
// round-robin deal hands every wave its share of both kinds.  E::edge(e) classifies list entry e; returns the ix-th entry in that order.
template <class E> HD constexpr int deal_order(int n, int ix) {
    int c = 0;
    for (int e = 0; e < n; e++) if (!E::edge(e)) { if (c == ix) return e; c++; }
    for (int e = 0; e < n; e++) if (E::edge(e)) { if (c == ix) return e; c++; }
    return ix;
}
// MFMA tile lists of the two matrix phases of a Riccati step, dealt round-robin over the 4 waves.  W is a template
// parameter so that every tile's kind and offsets are compile-time constants after unrolling.
// DEAL 1 (SW_OVERLAP_LDLT): wave 0 forms what only Quu needs - the HB and lD tiles - and goes on to the Quu tiles inside the same phase; waves 1..3 share
// the HA and lC tiles
template <int W, int N, int M, int PY, class R, int DEAL = 0>
HD void sweep_tiles1(SweepLdsT<N, M, PY, R>& S, int lane, R dt) {
    constexpr int LDN = SweepLdsT<N, M, PY, R>::LDN, LDM = SweepLdsT<N, M, PY, R>::LDM, AR = SweepLdsT<N, M, PY, R>::AR, LDA = SweepLdsT<N, M, PY, R>::LDA, A0 = SweepLdsT<N, M, PY, R>::A0;
    constexpr int TN = (N + 15) / 16, TM = (M + 15) / 16, TP = (PY + 15) / 16, TPd = TP > 0 ? TP : 1;
    constexpr int t1 = TN * TN, t2 = t1 + TN * TM, t3 = t2 + TP * TN, t4 = t3 + TP * TM;
    // tiles dealt round-robin, t = W + 4 q - except for the whole-body sizes (16 tiles), where the per-wave stamps (tools/microbench.py, -DSW_PROF)
    // put wave 3 (HA, HA, HB(32,0), lD) 1.3 k cycles behind wave 0 (HA, HA, HA, lC): the lD tile goes to wave 0
    constexpr bool WBT = (DEAL == 0 && t4 == 16 && t3 == 15);
    constexpr int nA = t1 + (t3 - t2), nB = (t2 - t1) + (t4 - t3);      // DEAL 1: tiles of waves 1..3 (HA, lC) / of wave 0 (HB, lD)
    constexpr int NTL = DEAL == 1 ? (W == 0 ? nB : (nA - (W - 1) + 2) / 3) : WBT ? (W == 0 ? 5 : W == 3 ? 3 : 4) : (t4 - W + 3) / 4;
    if (NTL <= 0) return;
    MTileT<R> td[NTL > 0 ? NTL : 1];
    _Pragma("unroll") for (int q = 0; q < NTL; q++) {
        struct E1 { static constexpr bool edge(int e) { return e < t1 ? mfma_edge_tile<R>(N, N, 16 * (e % TN), 16 * (e / TN)) : mfma_edge_tile<R>(PY, N, 16 * ((e - t1) % TPd), 16 * ((e - t1) / TPd)); } };
        const int ix = DEAL == 1 && W > 0 ? deal_order<E1>(nA, (W - 1) + 3 * q) : (W - 1) + 3 * q;
        const int t = DEAL == 1 ? (W == 0 ? (q < t2 - t1 ? t1 + q : t3 + (q - (t2 - t1))) : (ix < t1 ? ix : t2 + (ix - t1))) : (WBT && W == 0 && q == 4) ? 15 : W + 4 * q;
        // HA = H A = H(:, A0:) A_low (+ H [I, dt I] when the upper rows of A are the forward-Euler identities) ; HB = H(:, A0:) B_low
        if (t < t1) { td[q] = MTileT<R>{S.HA, LDN, nullptr, 0, 16 * (t % TN), 16 * (t / TN), N, N, S.H + LDN * A0, LDN, S.A, LDA, AR, false, nullptr, 0, nullptr, 0, 0};
                      if (A0 > 0) { td[q].T = S.H; td[q].ldt = LDN; td[q].tmode = 1; td[q].tsplit = A0; td[q].tscale = dt; } }
        else if (t < t2) td[q] = MTileT<R>{S.HB, LDN, nullptr, 0, 16 * ((t - t1) % TN), 16 * ((t - t1) / TN), N, M, S.H + LDN * A0, LDN, S.B, LDA, AR, false, nullptr, 0, nullptr, 0, 0};
        else if (t < t3) td[q] = MTileT<R>{S.lC, LDM, nullptr, 0, 16 * ((t - t2) % TPd), 16 * ((t - t2) / TPd), PY, N, S.lyy, LDM, S.C, LDM, PY, false, nullptr, 0, nullptr, 0, 0};
        else td[q] = MTileT<R>{S.lD, LDM, nullptr, 0, 16 * ((t - t3) % TPd), 16 * ((t - t3) / TPd), PY, M, S.lyy, LDM, S.D, LDM, PY, false, nullptr, 0, nullptr, 0, 0};
    }
    mfma_tiles<(NTL > 0 ? NTL : 1), ((AR > PY ? AR : PY) + 3) / 4 * 4, 0, R>(lane, td);
}
// DEAL 0: every tile round-robin over the four waves.  DEAL 1 (SW_OVERLAP_LDLT): the Quu tiles on wave 0 - which goes straight on to the factorisation of
// Quu while waves 1..3 share the Qxx and Qux tiles (the factorisation needs nothing else of this phase)
template <int W, int N, int M, int PY, class R, int DEAL = 0>
HD void sweep_tiles2(SweepLdsT<N, M, PY, R>& S, int lane, R reg, R dt) {
    constexpr int LDN = SweepLdsT<N, M, PY, R>::LDN, LDM = SweepLdsT<N, M, PY, R>::LDM, AR = SweepLdsT<N, M, PY, R>::AR, LDA = SweepLdsT<N, M, PY, R>::LDA, A0 = SweepLdsT<N, M, PY, R>::A0;
    constexpr int TN = (N + 15) / 16, TM = (M + 15) / 16;
    // Qxx = lxx + A^T H A + C^T lyy C is symmetric: only the tiles on and above the block diagonal are formed (6 of 9 for the whole
    // body), the symmetrisation step of the reference (SinglePhase.cpp:376) fills the rest
    constexpr int t1 = TN * (TN + 1) / 2, t2 = t1 + TM * TN, t3 = t2 + TM * TM;
    constexpr int NTL = DEAL == 0 ? (t3 - W + 3) / 4 : DEAL == 2 ? (t3 - t2 - W + 3) / 4 : (W == 0 ? t3 - t2 : (t2 - (W - 1) + 2) / 3);      // DEAL 2: the Quu tiles alone, round-robin
    if (NTL <= 0) return;
    MTileT<R> td[NTL > 0 ? NTL : 1];
    _Pragma("unroll") for (int q = 0; q < NTL; q++) {
        struct E2 { static constexpr bool edge(int e) {
            if (e < t1) { int c = 0; for (int jj = 0; jj < TN; jj++) for (int ii = 0; ii <= jj; ii++) { if (c == e) return mfma_edge_tile<R>(N, N, 16 * ii, 16 * jj); c++; } return false; }
            return mfma_edge_tile<R>(M, N, 16 * ((e - t1) % TM), 16 * ((e - t1) / TM)); } };
        const int t = DEAL == 0 ? W + 4 * q : DEAL == 2 ? t2 + W + 4 * q : (W == 0 ? t2 + q : deal_order<E2>(t2, (W - 1) + 3 * q));
        int bi = 0, bj = 0;     // t-th pair (bi <= bj) in column order
        { int c = 0; for (int jj = 0; jj < TN; jj++) for (int ii = 0; ii <= jj; ii++) { if (c == t) { bi = ii; bj = jj; } c++; } }
        // A^T HA = A_low^T HA(A0:, :) (+ [I, dt I]^T HA(:A0, :)) ; B^T HA = B_low^T HA(A0:, :) ; B^T HB = B_low^T HB(A0:, :)
        if (t < t1) { td[q] = MTileT<R>{S.H, LDN, nullptr, 0, 16 * bi, 16 * bj, N, N, S.A, LDA, S.HA + A0, LDN, AR, true, S.C, LDM, S.lC, LDM, PY}; if (bi == bj) td[q].dadd = reg; else td[q].mirror = true;     // into the H block (dead since phase 1); regularisation on Qxx as well: quirk x; an upper tile also fills its mirror image (the symmetrisation of SinglePhase.cpp:376 for the tiles whose lower partner is not formed)
                      if (A0 > 0) { td[q].T = S.HA; td[q].ldt = LDN; td[q].tmode = 2; td[q].tsplit = A0; td[q].tscale = dt; } }
        else if (t < t2) td[q] = MTileT<R>{S.Qux, LDM, nullptr, 0, 16 * ((t - t1) % TM), 16 * ((t - t1) / TM), M, N, S.B, LDA, S.HA + A0, LDN, AR, true, S.D, LDM, S.lC, LDM, PY};
        else { td[q] = MTileT<R>{S.Quu, LDM, S.Quu, LDM, 16 * ((t - t2) % TM), 16 * ((t - t2) / TM), M, M, S.B, LDA, S.HB + A0, LDN, AR, true, S.D, LDM, S.lD, LDM, PY}; if ((t - t2) % TM == (t - t2) / TM) td[q].dadd = reg; }
    }
    mfma_tiles<(NTL > 0 ? NTL : 1), (AR + 3) / 4 * 4, (PY + 3) / 4 * 4, R>(lane, td);
}

// H = Qxx + Qux^T K : the TN x TN tiles dealt round-robin, a wave's tiles interleaved (their short accumulation chains overlap)
template <int W, int N, int M, int PY, class R>
HD void sweep_tiles3(SweepLdsT<N, M, PY, R>& S, int lane) {
    constexpr int LDN = SweepLdsT<N, M, PY, R>::LDN, LDM = SweepLdsT<N, M, PY, R>::LDM;
    constexpr int TN = (N + 15) / 16;
    constexpr int NTL = (TN * TN - W + 3) / 4;
    if (NTL <= 0) return;
    MTileT<R> td[NTL > 0 ? NTL : 1];
    _Pragma("unroll") for (int q = 0; q < NTL; q++) {
        struct E3 { static constexpr bool edge(int e) { return mfma_edge_tile<R>(N, N, 16 * (e % TN), 16 * (e / TN)); } };
        const int t = deal_order<E3>(TN * TN, W + 4 * q);
        td[q] = MTileT<R>{S.H, LDN, S.H, LDN, 16 * (t % TN), 16 * (t / TN), N, N, S.Qux, LDM, S.K, LDM, M, true, nullptr, 0, nullptr, 0, 0};     // in place: H holds Qxx
    }
    mfma_tiles<(NTL > 0 ? NTL : 1), (M + 3) / 4 * 4, 0, R>(lane, td);
}

// K = (-Quu_inv) Qux : TM x TN tiles dealt round-robin
template <int W, int N, int M, int PY, class R>
HD void sweep_tilesK(SweepLdsT<N, M, PY, R>& S, int lane) {
    constexpr int LDM = SweepLdsT<N, M, PY, R>::LDM;
    constexpr int TN = (N + 15) / 16, TM = (M + 15) / 16;
    constexpr int NTL = (TM * TN - W + 3) / 4;
    if (NTL <= 0) return;
    MTileT<R> td[NTL > 0 ? NTL : 1];
    _Pragma("unroll") for (int q = 0; q < NTL; q++) {
        const int t = W + 4 * q;
        td[q] = MTileT<R>{S.K, LDM, nullptr, 0, 16 * (t % TM), 16 * (t / TM), M, N, S.LQ, LDM, S.Qux, LDM, M, false, nullptr, 0, nullptr, 0, 0};
    }
    mfma_tiles<(NTL > 0 ? NTL : 1), (M + 3) / 4 * 4, 0, R>(lane, td);
}

// One phase of the backward sweep for problem b. On entry S.G/S.H hold (Gprime, Hprime) (already through Px^T).
template <int NT, int N, int M, int PY, class R, class LDS>
HD bool riccati_phase(LDS& SS, const PhaseDev& P, int b, R reg) {
    using RL = RecLayout<N, M, PY>; using ST = SweepLdsT<N, M, PY, R>;
    static_assert(NT == 256 && RL::rounds + 1 <= SW_PRE && N <= SW_N && 2 * N + M + PY <= NT && 64 + M <= NT - N - 1 - M && N <= 64 && M <= 64, "sweep limits");
    // lanes of the mat-vec chains that ride along with the tile phases (balance measured with the per-wave stamps, -DSW_PROF)
    constexpr bool WBS = (N == 36 && M == 12 && PY == 12);
    // (measured and rejected for the 24 x 24 blocks of the kinodynamic model, where wave 0 carries 8 of the 12 phase-1 tiles on top of a 17 k-cycle factorisation: without the
    // overlap the one-phase factorisation holds more live values than the 96 / 168-register occupancy points of those kernels allow - 529 spilled registers, k_sweep32 197 against 76 ms)
    constexpr bool OVL = SW_OVERLAP_LDLT != 0;
    constexpr int GN0 = WBS ? 64 : 0, G0 = WBS ? 64 : NT - N, DV0 = WBS ? 64 + N : NT - N - 1, DU0 = WBS ? 64 + N + 1 : NT - N - 1 - M;
    constexpr int LDN = ST::LDN, LDM = ST::LDM, AR = ST::AR, LDA = ST::LDA, A0 = ST::A0;
    static_assert(sizeof(ST) <= sizeof(SS.raw), "the view fits the raw LDS block of its model set");
    ST& S = *reinterpret_cast<ST*>(SS.raw); SweepCtl& SWC = SS.c;
    constexpr int TN = (N + 15) / 16, TM = (M + 15) / 16, TP = (PY + 15) / 16;   // 16x16 MFMA tiles per dimension
    const int h = P.h; const R dtR = (R)P.dt;
    // trajectory pointers of the phase, read ONCE: a descriptor field fetched inside the knot loop is a vector load whose wait
    // (vmcnt(0)) would also drain the record prefetch that is meant to stay in flight for a whole knot
    const auto grec = rec_of<R>(P); const auto gDefect = P.Defect; const auto gQu = P.Qu; const auto gQuu = P.Quu; const auto gQux = P.Qux;
    const auto gK = P.K; const auto gdU = P.dU; const auto gG = P.G;
    SW_PRE_DECL
    // terminal: G[h] = Phix + Gprime ; H[h] = Phixx + Hprime  (SinglePhase.cpp:326-327); prefetch knot h-1
    HS_PHASE(NT, { const int i = tid % N, j0 = tid / N; if (j0 < NT / N) for (int j = j0; j < N; j += NT / N) CM(S.H, i, j, LDN) += P.Phixx[(size_t)b * N * N + i + N * j]; }
             if (tid < N) { const R g = SWC.xfer[tid] + P.Phix[(size_t)b * N + tid]; S.G[tid] = g; gG[((size_t)b * (h + 1) + h) * N + tid] = g; }
             if (tid == 0) { SWC.ok = 1; }
             SW_RICCATI_FETCH((size_t)b * h + (h - 1), h - 1) SW_RICCATI_FETCH_LXX((size_t)b * h + (h - 1)))
    for (int k = h - 1; k >= 0; k--) {
        const size_t kk = (size_t)b * h + k;
        SW_STAMP0()
        HS_PHASE_L(NT, SW_RICCATI_COMMIT() if (k > 0) SW_RICCATI_FETCH(kk - 1, k - 1))
        SW_STAMP(0)
        // phase 1 (matrix cores): HA = H A (TN x TN tiles) ; HB = H B (TN x TM) ; lC = lyy C (TP x TN) ; lD = lyy D (TP x TM), dealt
        // round-robin over the 4 waves (whole body: 30 MFMAs per wave) ; Gnext = G + H Defect[k+1]
        HS_PHASE_L(NT, {
            const int w = tid >> 6, lane = tid & 63;
            if constexpr (OVL && M > 12) {
            // 24 x 24 control blocks (kinodynamic model): HB and Quu are four tiles each - on wave 0 alone they made it 9.9 k cycles against the others' 1.4-2.0 k
            // (per-wave stamps).  Every phase-1 tile round-robin here, the Quu tiles over the four waves in a short phase of their own, then the factorisation beside
            // the Qxx / Qux tiles as for the small blocks
            switch (w) { case 0: sweep_tiles1<0, N, M, PY, R>(S, lane, dtR); break; case 1: sweep_tiles1<1, N, M, PY, R>(S, lane, dtR); break;
                         case 2: sweep_tiles1<2, N, M, PY, R>(S, lane, dtR); break; default: sweep_tiles1<3, N, M, PY, R>(S, lane, dtR); }
            } else if constexpr (OVL) {
            // wave 0: HB, lD - what only Quu needs - and, without a workgroup barrier (its own tiles), the Quu tiles themselves: Quu += B^T HB + D^T lD with the
            // regularisation on the diagonal; waves 1..3: HA, lC
            switch (w) { case 0: sweep_tiles1<0, N, M, PY, R, 1>(S, lane, dtR);
#ifndef HS_HOST_EMU
                                 __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
                                 sweep_tiles2<0, N, M, PY, R, 1>(S, lane, reg, dtR); break;
                         case 1: sweep_tiles1<1, N, M, PY, R, 1>(S, lane, dtR); break;
                         case 2: sweep_tiles1<2, N, M, PY, R, 1>(S, lane, dtR); break; default: sweep_tiles1<3, N, M, PY, R, 1>(S, lane, dtR); }
            } else {
            switch (w) { case 0: sweep_tiles1<0, N, M, PY, R>(S, lane, dtR); break; case 1: sweep_tiles1<1, N, M, PY, R>(S, lane, dtR); break;
                         case 2: sweep_tiles1<2, N, M, PY, R>(S, lane, dtR); break; default: sweep_tiles1<3, N, M, PY, R>(S, lane, dtR); }
            }
            // (whole body: the Gnext chain rides on wave 1 - per-wave stamps show it 2.5 k cycles ahead of wave 0, which carries three of the nine HA
            // tiles; the 24 / 12-row models deal two tiles to every wave and keep it on wave 0)
            if (tid >= GN0 && tid < GN0 + N) { const int i = tid - GN0; R s = S.G[i]; _Pragma("unroll 6") for (int j = 0; j < N; j++) s += CM(S.H, i, j, LDN) * S.def[j]; S.Gn[i] = s; }
            SW_WSTAMP(0)
        })
        if constexpr (OVL && M > 12) {
            HS_PHASE_L(NT, { const int w = tid >> 6, lane = tid & 63;
                switch (w) { case 0: sweep_tiles2<0, N, M, PY, R, 2>(S, lane, reg, dtR); break; case 1: sweep_tiles2<1, N, M, PY, R, 2>(S, lane, reg, dtR); break;
                             case 2: sweep_tiles2<2, N, M, PY, R, 2>(S, lane, reg, dtR); break; default: sweep_tiles2<3, N, M, PY, R, 2>(S, lane, reg, dtR); } })
        }
        SW_STAMP(1)
        if constexpr (OVL) {
        // phase 2, two jobs side by side.  Wave 0 (its Quu tiles are done: phase 1): Eigen's pivoted LDLT of (Quu - 1e-9 I), pivot order + factorisation
        // (SinglePhase.cpp:366-372).  Waves 1..3: Qxx - lxx =
        // A^T HA + C^T lC (into the H block), Qux = B^T HA + D^T lC, and the chains Qx += A^T Gn + C^T ly, Qu += B^T Gn + D^T ly.  The factorisation
        // needs nothing else of this phase, so its 5.5 k cycles run under the other waves' tiles instead of behind them.
        // Scratch of the LDLT: the HB block (only the Quu tiles read it, and they are wave 0's own).
        HS_PHASE_L(NT, {
            const int w = tid >> 6, lane = tid & 63;
            switch (w) { case 0: break; case 1: sweep_tiles2<1, N, M, PY, R, 1>(S, lane, reg, dtR); break;
                         case 2: sweep_tiles2<2, N, M, PY, R, 1>(S, lane, reg, dtR); break; default: sweep_tiles2<3, N, M, PY, R, 1>(S, lane, reg, dtR); }
            if (tid >= 128 && tid < 128 + N) {
                const int i = tid - 128; R s = 0;
                if (A0 > 0) s = (i < A0) ? S.Gn[i] : dtR * S.Gn[i - A0];       // [I, dt I]^T Gn(:A0): the upper rows of the whole-body A
                _Pragma("unroll 6") for (int t = 0; t < AR; t++) s += CM(S.A, t, i, LDA) * S.Gn[A0 + t];
                if (PY > 0) { _Pragma("unroll 6") for (int t = 0; t < PY; t++) s += CM(S.C, t, i, LDM) * S.ly[t]; }
                S.Qx[i] += s;
            } else if (tid >= 192 && tid < 192 + M) {
                const int a = tid - 192; R s = 0;
                _Pragma("unroll 6") for (int t = 0; t < AR; t++) s += CM(S.B, t, a, LDA) * S.Gn[A0 + t];
                if (PY > 0) { _Pragma("unroll 6") for (int t = 0; t < PY; t++) s += CM(S.D, t, a, LDM) * S.ly[t]; }
                S.Qu[a] += s;
            }
            if (w == 0) ldlt_parts<M, LDM, R, 1>(lane, S.Quu, R(-1e-9), S.LQ, S.HB, reinterpret_cast<int*>(S.HB + M * M), &SWC.ok);
            SW_WSTAMP(1)
        })
        SW_STAMP(2)
        // wave 0: the solves, LQ = -Quu_inv = -LDLT.solve(I) (SinglePhase.cpp:375) ; waves 1..3: symmetrise Qxx (SinglePhase.cpp:376)
        HS_PHASE_L(NT,
            if (tid < 64) ldlt_parts<M, LDM, R, 2>(tid, S.Quu, R(-1e-9), S.LQ, S.HB, reinterpret_cast<int*>(S.HB + M * M), &SWC.ok, S.HA, S.A);      // (HA, A: dead since phase 2)
            else for (int e = tid - 64; e < TN * 256; e += NT - 64) {      // inside a diagonal tile both halves were formed: average them (the upper tiles mirrored themselves when they were stored)
                const int i = 16 * (e >> 8) + (e & 15), j = 16 * (e >> 8) + ((e >> 4) & 15);
                if (i < j && j < N) { const R s = (CM(S.H, i, j, LDN) + CM(S.H, j, i, LDN)) / 2; CM(S.H, i, j, LDN) = s; CM(S.H, j, i, LDN) = s; }
            })
        SW_STAMP(5)
        } else {
        // phase 2: Qxx - lxx = A^T HA + C^T lC (TN x TN, into the H block) ; Qux = B^T HA + D^T lC (TM x TN) ; Quu += B^T HB + D^T lD (TM x TM), round-robin ;
        // Qx += A^T Gn + C^T ly ; Qu += B^T Gn + D^T ly
        HS_PHASE_L(NT, {
            const int w = tid >> 6, lane = tid & 63;
            switch (w) { case 0: sweep_tiles2<0, N, M, PY, R>(S, lane, reg, dtR); break; case 1: sweep_tiles2<1, N, M, PY, R>(S, lane, reg, dtR); break;
                         case 2: sweep_tiles2<2, N, M, PY, R>(S, lane, reg, dtR); break; default: sweep_tiles2<3, N, M, PY, R>(S, lane, reg, dtR); }
            // the two mat-vec chains ride on waves 2 and 3, which carry two tiles each in this phase (waves 0 and 1: three)
            if (tid >= 128 && tid < 128 + N) {
                const int i = tid - 128; R s = 0;
                if (A0 > 0) s = (i < A0) ? S.Gn[i] : dtR * S.Gn[i - A0];       // [I, dt I]^T Gn(:A0): the upper rows of the whole-body A
                _Pragma("unroll 6") for (int t = 0; t < AR; t++) s += CM(S.A, t, i, LDA) * S.Gn[A0 + t];
                if (PY > 0) { _Pragma("unroll 6") for (int t = 0; t < PY; t++) s += CM(S.C, t, i, LDM) * S.ly[t]; }
                S.Qx[i] += s;
            } else if (tid >= 192 && tid < 192 + M) {
                const int a = tid - 192; R s = 0;
                _Pragma("unroll 6") for (int t = 0; t < AR; t++) s += CM(S.B, t, a, LDA) * S.Gn[A0 + t];
                if (PY > 0) { _Pragma("unroll 6") for (int t = 0; t < PY; t++) s += CM(S.D, t, a, LDM) * S.ly[t]; }
                S.Qu[a] += s;
            }
            SW_WSTAMP(1)
        })
        SW_STAMP(2)
        // (the regularisation of Quu and Qxx went onto the diagonals with the tiles' stores) ; store Qu / Quu / Qux as the reference keeps them
        HS_PHASE_L(NT, if (tid < M) gQu[kk * M + tid] = S.Qu[tid];
                   st_mat<NT>(tid, gQuu + kk * M * M, S.Quu, LDM, M, M); st_mat<NT>(tid, gQux + kk * M * N, S.Qux, LDM, M, N);)
        SW_STAMP(3)
        // wave 0: Eigen's pivoted LDLT of (Quu - 1e-9 I), positivity test, LQ = -Quu_inv = -LDLT.solve(I) (SinglePhase.cpp:366-375);
        // meanwhile the other waves symmetrise Qxx (next phase).  Scratch: the HA and HB blocks (dead since phase 2).
        ldlt_inverse_w<M, LDM, R>(S.Quu, R(-1e-9), S.LQ, S.HA, reinterpret_cast<int*>(S.HB), &SWC.ok);
        SW_STAMP(4)
        HS_PHASE_L(NT,
            if (tid >= 64) for (int e = tid - 64; e < N * N; e += NT - 64) {
                const int i = e % N, j = e / N;
                if (i < j) {     // inside a diagonal tile both halves were formed: average them; elsewhere mirror the upper tile
                    const R s = (i / 16 == j / 16) ? (CM(S.H, i, j, LDN) + CM(S.H, j, i, LDN)) / 2 : CM(S.H, i, j, LDN);
                    CM(S.H, i, j, LDN) = s; CM(S.H, j, i, LDN) = s;
                }
            })
        SW_STAMP(5)
        }
        if (!SWC.ok) return false;
        SW_STAMP(6)
        // K = -Quu_inv Qux on the matrix cores (TM x TN tiles over the waves) ; dU = -Quu_inv Qu on the last lanes   (SinglePhase.cpp:379-380)
        HS_PHASE_L(NT,
            { const int w = tid >> 6, lane = tid & 63;
              switch (w) { case 0: sweep_tilesK<0, N, M, PY, R>(S, lane); break; case 1: sweep_tilesK<1, N, M, PY, R>(S, lane); break;
                           case 2: sweep_tilesK<2, N, M, PY, R>(S, lane); break; default: sweep_tilesK<3, N, M, PY, R>(S, lane); } }
            if (tid >= NT - M) { const int i = tid - (NT - M); R s = 0; _Pragma("unroll") for (int t = 0; t < M; t++) s += CM(S.LQ, i, t, LDM) * S.Qu[t]; S.dU[i] = s; }
            // lxx joins Qxx here, from the registers it was prefetched into (symmetric by construction; the symmetrisation of SinglePhase.cpp:376
            // has acted on the products): Qxx = lxx + A^T H A + C^T lyy C + reg I is complete before the next phase reads it
            SW_RICCATI_LXX() if (k > 0) SW_RICCATI_FETCH_LXX(kk - 1) SW_WSTAMP(2))
        // H = Qxx + Qux^T K ; G = Qx + Qux^T dU ; dV ; store K, dU, G
        HS_PHASE_L(NT,
            { const int w = tid >> 6, lane = tid & 63;      // H = Qxx + Qux^T K on the matrix cores: 9 tiles over 4 waves
              switch (w) { case 0: sweep_tiles3<0, N, M, PY, R>(S, lane); break; case 1: sweep_tiles3<1, N, M, PY, R>(S, lane); break;
                           case 2: sweep_tiles3<2, N, M, PY, R>(S, lane); break; default: sweep_tiles3<3, N, M, PY, R>(S, lane); } }
            // (whole body: G, dV and the store of dU on wave 1 - two tiles and the shortest time in this phase; wave 3 carried them 1 k cycles behind)
            if constexpr (!WBS && M > 12) {
                // (24-row models: every wave carries one H tile, and the G chain - 24 dependent multiply-adds per lane - put wave 3 at 10.1 k cycles against 5-6 k:
                // each row on a quad of lanes, the four partial sums added by DPP as in the linear rollout)
                constexpr int CH = (M + 3) / 4;
                SW_QUAD_ROWS(N, { _Pragma("unroll") for (int jj = 0; jj < CH; jj++) { const int t = part * CH + jj; if (t < M) partial += CM(S.Qux, t, o, LDM) * S.dU[t]; } },
                             { const R s = S.Qx[o] + total; S.G[o] = s; gG[((size_t)b * (h + 1) + k) * N + o] = s; })
            }
            if ((WBS || M <= 12) && tid >= G0 && tid < G0 + N) { const int i = tid - G0; R s = S.Qx[i]; _Pragma("unroll") for (int t = 0; t < M; t++) s += CM(S.Qux, t, i, LDM) * S.dU[t]; S.G[i] = s; gG[((size_t)b * (h + 1) + k) * N + i] = s; }
            else if (tid == DV0) { R dVk = 0; _Pragma("unroll") for (int t = 0; t < M; t++) dVk -= S.Qu[t] * S.dU[t]; SWC.dV1 -= dVk; SWC.dV2 += dVk; }
            else if (tid >= DU0 && tid < DU0 + M) { const int a = tid - DU0; gdU[kk * M + a] = S.dU[a]; } SW_WSTAMP(3))
        SW_STAMP(7)
        if constexpr (OVL) {
        // stores: the gains, and Qu / Quu / Qux as the reference keeps them (Quu carries the regularisation, SinglePhase.cpp:364-365) - by every wave alike,
        // so that the count of stores between the record prefetch and its commit is the same on all of them
        HS_PHASE_L(NT, st_mat<NT>(tid, gK + kk * M * N, S.K, LDM, M, N); if (tid < M) gQu[kk * M + tid] = S.Qu[tid];
                   st_mat<NT>(tid, gQuu + kk * M * M, S.Quu, LDM, M, M); st_mat<NT>(tid, gQux + kk * M * N, S.Qux, LDM, M, N);)
        } else {
        HS_PHASE_L(NT, st_mat<NT>(tid, gK + kk * M * N, S.K, LDM, M, N);)
        }
        SW_STAMP(8)
    }
    // G[0] += H[0] * Defect[0]   (SinglePhase.cpp:389)
    HS_PHASE(NT, if (tid < N) S.def[tid] = gDefect[((size_t)b * (h + 1)) * N + tid];)
    HS_PHASE(NT, if (tid < N) { R s = S.G[tid]; for (int j = 0; j < N; j++) s += CM(S.H, tid, j, LDN) * S.def[j]; S.Gn[tid] = s; })
    HS_PHASE(NT, if (tid < N) { SWC.xfer[tid] = S.Gn[tid]; gG[((size_t)b * (h + 1)) * N + tid] = S.Gn[tid]; }
             st_mat<NT>(tid, P.H0 + (size_t)b * N * N, S.H, LDN, N, N);)
    return true;
}

// full multi-phase backward sweep of problem b (phases may differ in dimension: WB 36/12/12, HKD 24/24/0, SRB 12/12/0);
// returns success, writes dV into S.c.dV1/dV2.  H of the phase being processed sits at the start of the raw block with
// ld n+1 in every view; the gradient G crosses phase boundaries through S.c.xfer.
template <int NT, class R, int SET, class LDS>
HD bool riccati_sweep(LDS& S, const PhaseDev* ph, int nph, int b, R reg) {
    HS_PHASE(NT, if (tid == 0) { S.c.dV1 = 0.0; S.c.dV2 = 0.0; })
    for (int i = nph - 1; i >= 0; i--) {
        const PhaseDev& P = ph[i];
        const int n = P.n;
        if (i == nph - 1) {
            HS_PHASE(NT, for (int e = tid; e < (n + 1) * n; e += NT) S.raw[e] = 0.0; if (tid < n) S.c.xfer[tid] = 0.0;)
        } else {   // impact-aware step: (G,H) <- (Px^T G, Px^T H Px), Px = nn x n, nn <= n  (MultiPhaseDDP.cpp:196-201); once per phase boundary
            const int nn = P.next_n, ldh = nn + 1;
            R* Hn = S.raw;                          // H of the later phase: nn x nn, ld nn+1
            R* Px = S.raw + (n + 1) * n;            // scratch in the A / HA regions of the current view (beyond Hn because n >= nn)
            R* HP = S.raw + 2 * (n + 1) * n;
            const auto* Pxg = P.Px + (size_t)b * nn * n;
            R gnew = 0.0;
            HS_PHASE(NT, for (int e = tid; e < nn * n; e += NT) Px[e] = Pxg[e];)
            HS_PHASE(NT,
                for (int e = tid; e < nn * n; e += NT) { const int r = e % nn, c = e / nn; R s = 0; for (int t = 0; t < nn; t++) s += Hn[r + ldh * t] * Px[t + nn * c]; HP[e] = s; }
                if (tid >= NT - n) { const int i2 = tid - (NT - n); R s = 0; for (int t = 0; t < nn; t++) s += Px[t + nn * i2] * S.c.xfer[t]; S.raw[3 * (n + 1) * n + i2] = s; })
            (void)gnew;
            HS_PHASE(NT,
                for (int e = tid; e < n * n; e += NT) { const int r = e % n, c = e / n; R s = 0; for (int t = 0; t < nn; t++) s += Px[t + nn * r] * HP[t + nn * c]; S.raw[r + (n + 1) * c] = s; }
                if (tid >= NT - n) S.c.xfer[tid - (NT - n)] = S.raw[3 * (n + 1) * n + tid - (NT - n)];)
        }
        bool ok;
        // (the kernels are instantiated per model SET: the 24-row factor of the kinodynamic model needs twice the registers of the 12-row
        //  ones, and a kernel that carries both spills in every instantiation)
        if constexpr (SET == SW_SET_WB) {
            if (P.model == HSDDP_MODEL_WB) ok = riccati_phase<NT, 36, 12, 12, R>(S, P, b, reg); else ok = riccati_phase<NT, 12, 12, 0, R>(S, P, b, reg);
        } else {
            if (P.model == HSDDP_MODEL_SRB) ok = riccati_phase<NT, 12, 12, 0, R>(S, P, b, reg); else ok = riccati_phase<NT, 24, 24, 0, R>(S, P, b, reg);
        }
        if (!ok) return false;
    }
    return true;
}

// ---- linear rollout: forward over phases/knots; next knot prefetched into registers (dense ld = rows layouts) ----
//   rounds: A (rA) | lxx (rQ) | B (rB) | K (rK) | luu (rLuu) | [lx(N) lu(M) dU(M) Defect[k+1](N)]
#define SW_LIN_FETCH(kk_, k_) { \
    const HS_GLOBAL R* rec_ = grec + (kk_) * (size_t)RL::size + tid; \
    _Pragma("unroll") for (int r = 0; r < RL::rA; r++) PRE(r) = rec_[RL::oA + NT * r]; \
    _Pragma("unroll") for (int r = 0; r < RL::rQ; r++) PRE(RL::rA + r) = rec_[RL::oLxx + NT * r]; \
    _Pragma("unroll") for (int r = 0; r < RL::rB; r++) PRE(RL::rA + RL::rQ + r) = rec_[RL::oB + NT * r]; \
    _Pragma("unroll") for (int r = 0; r < rK; r++) { const int e = tid + NT * r; PRE(RL::rA + RL::rQ + RL::rB + r) = (e < M * N) ? (R)gK[(kk_) * M * N + e] : R(0.0); } \
    _Pragma("unroll") for (int r = 0; r < RL::rLuu; r++) PRE(RL::rA + RL::rQ + RL::rB + rK + r) = rec_[RL::oLuu + NT * r]; \
    PRE(RL::rA + RL::rQ + RL::rB + rK + RL::rLuu) = (tid < N + M) ? rec_[RL::oLx] : (tid < N + 2 * M) ? (R)gdU[(kk_) * M + tid - N - M] \
            : (tid < 2 * N + 2 * M) ? (R)gDefect[((size_t)b * (h + 1) + (k_) + 1) * N + tid - N - 2 * M] : R(0.0); }
#define SW_LIN_COMMIT(p_) { \
    R* A_ = S.A[p_]; R* Q_ = S.Q[p_]; R* B_ = S.B[p_]; R* K_ = S.K[p_]; R* U_ = S.U[p_]; R* v_ = S.v[p_]; \
    _Pragma("unroll") for (int r = 0; r < RL::rA; r++) { const int e = tid + NT * r; if (e < AR * N) A_[e] = PRE(r); } \
    _Pragma("unroll") for (int r = 0; r < RL::rQ; r++) { const int e = tid + NT * r; if (e < N * N) Q_[e] = PRE(RL::rA + r); } \
    _Pragma("unroll") for (int r = 0; r < RL::rB; r++) { const int e = tid + NT * r; if (e < AR * M) B_[e] = PRE(RL::rA + RL::rQ + r); } \
    _Pragma("unroll") for (int r = 0; r < rK; r++) { const int e = tid + NT * r; if (e < N * M) K_[e] = PRE(RL::rA + RL::rQ + RL::rB + r); } \
    _Pragma("unroll") for (int r = 0; r < RL::rLuu; r++) { const int e = tid + NT * r; if (e < M * M) U_[e] = PRE(RL::rA + RL::rQ + RL::rB + rK + r); } \
    { const R w_ = PRE(RL::rA + RL::rQ + RL::rB + rK + RL::rLuu); \
      if (tid < N) v_[tid] = w_; else if (tid < N + M) v_[64 + tid - N] = w_; \
      else if (tid < N + 2 * M) v_[128 + tid - N - M] = w_; else if (tid < 2 * N + 2 * M) v_[192 + tid - N - 2 * M] = w_; } }

// one phase of the linear rollout; on entry S.c.xfer holds dx_init (Px * dX_end of the previous phase, or 0), on exit dX_end.
// Two LDS sets (A, lxx, B, K, luu and the vectors) alternate between knots, so a knot costs two barriers: du = eps dU + K dx, then
// dx+ = A dx + B du + eps defect together with the commit of the next knot's record into the other set.  The contributions to
// dV_1 / dV_2 stay in registers (one partial sum per lane) and are added up once per phase.
template <int NT, int N, int M, int PY, class R, class LDS>
HD void linear_phase(LDS& SS, const PhaseDev& P, int b, R eps) {
    using RL = RecLayout<N, M, PY>; using ST = LinLdsT<N, M, PY, R>;
    constexpr int rK = rec_rnd(M * N) / 256, AR = ST::AR, A0 = ST::A0;
    static_assert(RL::rA + RL::rQ + RL::rB + rK + RL::rLuu + 1 <= SW_PRE && 2 * N + 2 * M <= NT && 4 * N <= 192 && 4 * M <= NT && 192 + M <= NT && N <= 64 && M <= 64, "prefetch registers / lane maps");
    const R dtR = (R)P.dt;
    static_assert(offsetof(ST, dx) >= 2 * NT * sizeof(R), "the partial-sum scratch must not reach dx");
    static_assert(sizeof(ST) <= sizeof(SS.raw), "the view fits the raw LDS block of its model set");
    ST& S = *reinterpret_cast<ST*>(SS.raw); SweepCtl& SWC = SS.c;
    const int h = P.h;
    const auto grec = rec_of<R>(P); const auto gDefect = P.Defect; const auto gK = P.K; const auto gdU = P.dU; const auto gdX = P.dX; const auto gKdX = P.KdX;   // read once (see riccati_phase)
    SW_PRE_DECL
#ifdef HS_HOST_EMU
    static R acc1_all_[NT], acc2_all_[NT];
    for (int t = 0; t < NT; t++) { acc1_all_[t] = 0.0; acc2_all_[t] = 0.0; }
#define ACC1 acc1_all_[tid]
#define ACC2 acc2_all_[tid]
#else
    R acc1_ = 0.0, acc2_ = 0.0;
#define ACC1 acc1_
#define ACC2 acc2_
#endif
    // dX[0] = dx_init + eps * Defect[0]
    HS_PHASE(NT, if (tid < N) { R v = SWC.xfer[tid] + eps * gDefect[((size_t)b * (h + 1)) * N + tid]; S.dx[tid] = v; gdX[((size_t)b * (h + 1)) * N + tid] = v; }
             SW_LIN_FETCH((size_t)b * h, 0))
    HS_PHASE_L(NT, SW_LIN_COMMIT(0) if (1 < h) SW_LIN_FETCH((size_t)b * h + 1, 1))
    for (int k = 0; k < h; k++) {
        const size_t kk = (size_t)b * h + k;
        const int p = k & 1;
        const R* A_ = S.A[p]; const R* Q_ = S.Q[p]; const R* B_ = S.B[p]; const R* K_ = S.K[p]; const R* U_ = S.U[p];
        const R* Qx_ = S.v[p]; const R* Qu_ = S.v[p] + 64; const R* dU_ = S.v[p] + 128; const R* def_ = S.v[p] + 192;
        const R* dxc = p ? S.dxn : S.dx; R* dxw = p ? S.dx : S.dxn;
        // du = eps dU + K dx : row o by the four lanes of quad o
        HS_PHASE_L(NT, SW_QUAD_ROWS(M, {
            constexpr int CH = (N + 3) / 4; R s = 0;
            _Pragma("unroll") for (int jj = 0; jj < CH; jj++) { const int j = part * CH + jj; if (j < N) s += CM(K_, o, j, M) * dxc[j]; }
            partial = s; }, { S.du[o] = eps * dU_[o] + total; gKdX[kk * M + o] = total; }))      // K dX also goes out: the rollout's u = ubar + eps (dU + K dX)
        HS_PHASE_L(NT,
            // dx+ = [A B] [dx; du] + eps defect and q = lxx dx : row o by quad o, a quarter of the N + M (resp. N) terms per lane
            SW_QUAD_ROWS(N, {
                constexpr int T = N + M; constexpr int CH = (T + 3) / 4; constexpr int CQ = (N + 3) / 4; R s = 0; R q = 0;
                if (o >= A0) {      // a stored row of [A B]
                    _Pragma("unroll") for (int jj = 0; jj < CH; jj++) { const int t = part * CH + jj; if (t < N) s += CM(A_, o - A0, t, AR) * dxc[t]; else if (t < T) s += CM(B_, o - A0, t - N, AR) * S.du[t - N]; }
                } else if (part == 0) s = dxc[o] + dtR * dxc[A0 + o];      // upper rows of the whole-body A = [I, dt I], B = 0
                _Pragma("unroll") for (int jj = 0; jj < CQ; jj++) { const int j = part * CQ + jj; if (j < N) q += CM(Q_, o, j, N) * dxc[j]; }
                partial = s; partial2 = q; }, {
                const R v = total + eps * def_[o];
                dxw[o] = v; gdX[((size_t)b * (h + 1) + k + 1) * N + o] = v;
                ACC2 += dxc[o] * total2;          // dx^T lxx dx
                ACC1 += Qx_[o] * dxc[o]; })
            if (tid >= 192 && tid < 192 + M) {
                const int a = tid - 192; R q = 0;
                _Pragma("unroll") for (int j = 0; j < M; j++) q += CM(U_, a, j, M) * S.du[j];
                ACC2 += S.du[a] * q; ACC1 += Qu_[a] * S.du[a];      // (+ du^T lux dx with lux == 0)
            }
            if (k + 1 < h) { SW_LIN_COMMIT(1 - p) if (k + 2 < h) SW_LIN_FETCH(kk + 2, k + 2) })
    }
    const R* dxe = (h & 1) ? S.dxn : S.dx;
    // terminal: dV_1 += Phix . dx ; dV_2 += dx^T Phixx dx ; then the per-lane partial sums of the whole phase
    HS_PHASE(NT, for (int e = tid; e < N * N; e += NT) S.Q[0][e] = P.Phixx[(size_t)b * N * N + e];)
    HS_PHASE(NT, if (tid % SW_QS == 0 && tid / SW_QS < N) { const int o = tid / SW_QS;      // on the lanes that carry the x-part partial sums
                     R q = 0; for (int j = 0; j < N; j++) q += CM(S.Q[0], o, j, N) * dxe[j]; ACC2 += dxe[o] * q; ACC1 += P.Phix[(size_t)b * N + o] * dxe[o]; })
    R* scr = SS.raw;     // 2 x NT partial sums (the matrices are no longer needed; dx / dxn live beyond the first 2 NT doubles of every view)
    HS_PHASE(NT, scr[tid] = ACC1; scr[NT + tid] = ACC2;)
    HS_PHASE(NT, if (tid == 0) { R a1 = 0, a2 = 0; for (int j = 0; j < N; j++) { a1 += scr[SW_QS * j]; a2 += scr[NT + SW_QS * j]; }
                                 R b1 = 0, b2 = 0; for (int j = 0; j < M; j++) { b1 += scr[192 + j]; b2 += scr[NT + 192 + j]; }
                                 SWC.dV1 += a1 + b1; SWC.dV2 += a2; SWC.dV2 += b2; }
             if (tid >= 64 && tid < 64 + N) SWC.xfer[tid - 64] = dxe[tid - 64];)
#undef ACC1
#undef ACC2
}

// linear rollout of problem b (eps = 1 in solve).  Returns dV_1, dV_2 in S.c.dV1/dV2.
template <int NT, class R, int SET, class LDS>
HD void linear_rollout(LDS& S, const PhaseDev* ph, int nph, int b, R eps) {
    HS_PHASE(NT, if (tid == 0) { S.c.dV1 = 0.0; S.c.dV2 = 0.0; } if (tid < SW_N) S.c.xfer[tid] = 0.0;)
    for (int i = 0; i < nph; i++) {
        const PhaseDev& P = ph[i];
        if (i > 0) {   // dx_init = Px * dX_end(prev)   (MultiPhaseDDP.cpp:27-30); xfer holds the previous phase's terminal dX
            const PhaseDev& Pp = ph[i - 1]; const int np = Pp.n, n = P.n;
            const auto* Pxg = Pp.Px + (size_t)b * n * np;
            HS_PHASE(NT, if (tid < n) { R s = 0; for (int t = 0; t < np; t++) s += Pxg[tid + n * t] * S.c.xfer[t]; S.raw[tid] = s; })
            HS_PHASE(NT, if (tid < n) S.c.xfer[tid] = S.raw[tid];)
        }
        if constexpr (SET == SW_SET_WB) {
            if (P.model == HSDDP_MODEL_WB) linear_phase<NT, 36, 12, 12, R>(S, P, b, eps); else linear_phase<NT, 12, 12, 0, R>(S, P, b, eps);
        } else {
            if (P.model == HSDDP_MODEL_SRB) linear_phase<NT, 12, 12, 0, R>(S, P, b, eps); else linear_phase<NT, 24, 24, 0, R>(S, P, b, eps);
        }
    }
}

}  // namespace hs
