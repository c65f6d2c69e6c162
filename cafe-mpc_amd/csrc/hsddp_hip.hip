// libhsddp_hip.so — MI355X (gfx950) implementation of include/hsddp.h.
//
// Host driver of the batched HS-DDP solve: MultiPhaseDDP<T>::solve (HSDDPSolver/source/MultiPhaseDDP.cpp:216-447)
// restructured as a per-problem state machine evaluated with masks over the whole batch, so that thousands of
// independent problems advance through rollout / LQ approximation / Riccati sweep / line search in lock-step
// kernel launches; the host reads back one 16-byte activity counter per line-search launch and per inner iteration to stop early.
// Kernels: wb_knot.hpp (one wavefront per knot), sweep.hpp (one workgroup per problem).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <cmath>
#include "hsddp.h"
#include "hs_types.hpp"
#include "hs_host.hpp"
#include "wb_knot.hpp"
#include "wb_quad.hpp"
#include "srb_knot.hpp"
#include "hkd_knot.hpp"
#include "sweep.hpp"

using namespace hs;

// the host's control flow reads four counters the stream delivers: synchronise, check the launches before it, and refuse counters that never arrived
#define SYNC_COUNTERS() do { HIPCK(hipStreamSynchronize(h->stream)); HIPCK(hipGetLastError()); if (h->h_counters[0] < 0 || h->h_counters[1] < 0 || h->h_counters[2] < 0 || h->h_counters[3] < 0) { fprintf(stderr, "[hsddp_hip] counters not delivered (%s:%d)\n", __FILE__, __LINE__); return HSDDP_ENODEV; } } while (0)
#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[hsddp_hip] %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); return HSDDP_ENODEV; } } while (0)

struct SlotArrays { double *cost, *dsq, *ming, *maxh; };
static long long g_dev_allocs = 0;      // device allocations made by this library (hsddp_debug_malloc_count: the MPC-tick test watches it)
#define hipMalloc(...) (++g_dev_allocs, hipMalloc(__VA_ARGS__))
#ifdef ROLL_WPE
#define ROLL_ATTR __attribute__((amdgpu_waves_per_eu(ROLL_WPE, ROLL_WPE)))
#else
#define ROLL_ATTR
#endif
// LQ workgroup: 128 threads = two waves per knot (tangent rounds side by side, then column solves || cost partials), capped at 256
// registers so that the four knots a CU holds in LDS make two waves per SIMD.  LQ_NT=64 is the one-wave variant.
#ifndef LQ_NT
#define LQ_NT 128
#endif
#ifndef LQ_WPE
#define LQ_WPE 2
#endif
#define LQ_ATTR __attribute__((amdgpu_waves_per_eu(LQ_WPE, LQ_WPE)))

// ------------------------------------------------------------------------------------------------ kernels
enum { MASK_NONE = 0, MASK_LS = 1, MASK_INNER = 2, MASK_OUTER = 3, MASK_LS_OK = 4, MASK_COMMIT = 5 };
__device__ inline bool masked_out(const ProbState& s, int mask) {
    if (mask == MASK_LS) return !s.ls_active;
    if (mask == MASK_INNER) return !s.inner_active;
    if (mask == MASK_OUTER) return !s.outer_active;
    if (mask == MASK_LS_OK) return !s.ls_success;
    if (mask == MASK_COMMIT) return !s.need_commit;
    return false;
}

// Single shooting from phase `first` on, by ONE wave: every knot takes the state its predecessor simulated (SinglePhase.cpp:187,211-220).
// Used for the young phases the receding-horizon update creates (SS_set empty, MHPCProblem.cpp:340-351; first > 0, stops at the next
// phase with shooting nodes) and for the whole horizon when option.MS is false (MultiPhaseDDP.cpp:65-68; first = 0, descriptors with
// every shooting flag cleared).  On entry the state to start from is in L.xnext (whole-body phase) / in Xsim[0] of the phase (SRB, HKD).
template <bool WBM, class LDS> __device__ __forceinline__ void rollout_chain(LDS& L, PhaseC* ph, int nph, int first, const ModelDev& md, int b, int nslots, double eps, const OptDev& opt, SlotOut so,
                                              size_t slot_base, int* fail, bool wr) {
    for (int pj = first; pj < nph && !ph[pj].shooting; pj++) {
        PhaseC& Q = ph[pj]; PhaseC* Qn = pj + 1 < nph ? &ph[pj + 1] : nullptr; const size_t s0 = slot_base + Q.slot0;
        if (WBM && Q.model == HSDDP_MODEL_WB) {
            if constexpr (WBM) {
            for (int kq = 0; kq < Q.h; kq++) wb_rollout_knot<64>(L, Q, md, b, kq, eps, opt.ReB_active, nullptr, so, s0 + kq, fail, true, wr);
            wb_rollout_terminal<64>(L, Q, Qn, md, b, eps, opt.AL_active, so, s0 + Q.h, true, wr);
            }
        } else if (Q.model == HSDDP_MODEL_SRB) {      // (reads the simulated state back from Xsim: not available to probes, see hsddp_solve)
            SrbLds& Ls = *reinterpret_cast<SrbLds*>(&L);
            for (int kq = 0; kq < Q.h; kq++) srb_rollout_knot<64>(Ls, Q, b, kq, eps, opt.ReB_active, nullptr, so, s0 + kq, fail, true, wr);
            srb_rollout_terminal<64>(Ls, Q, Qn, b, eps, so, s0 + Q.h, true, wr);
        } else {
            HkdLds& Lh = *reinterpret_cast<HkdLds*>(&L);
            for (int kq = 0; kq < Q.h; kq++) hkd_rollout_knot<64>(Lh, Q, b, kq, eps, opt.ReB_active, nullptr, so, s0 + kq, fail, true, wr);
            hkd_rollout_terminal<64>(Lh, Q, Qn, md, b, eps, opt.AL_active, so, s0 + Q.h, true, wr);
        }
    }
}

// Step lengths of one launch.  Ordinary launches carry one (eps[0], or the problem's own ls_eps when from_state is set: the commit of a
// batched line search); a PROBE launch carries the candidates eps[0..n-1] of MultiPhaseDDP::line_search (MultiPhaseDDP.cpp:95-133) that
// are still to be tried: grid = candidates x problems x slots, candidate c only leaves the per-slot partials of its merit function in
// slice c of the slot arrays - except candidate `writer` (the last of the search), which also writes the trajectories like an ordinary trial.
constexpr int MAXCAND = 12;
struct EpsList { double e[MAXCAND]; int n, writer, from_state; };
// Probe launches: grid = candidates x units.  Workgroups go round-robin over the 8 XCDs (one L2 each); the candidates of a unit read the same
// trajectories and differ in eps only.  Unit u lives on XCD u % 8 and its candidates occupy consecutive dispatch slots of that XCD, so one of them
// brings the lines into that L2 and the others find them there (candidate-major order streamed the whole ensemble from HBM once per candidate:
// probe launch of the quad kernel 11.4 -> 9.8 ms).  QUAD_CAND_MAJOR 1 = the old order.
#ifndef QUAD_CAND_MAJOR
#define QUAD_CAND_MAJOR 0
#endif
__device__ __forceinline__ void cand_unit(int per, int g, int& c, int& r) {
#if QUAD_CAND_MAJOR
    c = g / per; r = g - c * per;
#else
    const int ncand = gridDim.x / per;
    const int full = (per >> 3) << 3;
    if (g < full * ncand) { const int G = g / (8 * ncand), rem = g - G * 8 * ncand; c = rem >> 3; r = G * 8 + (rem & 7); }
    else { const int nt = per - full, g2 = g - full * ncand; c = g2 / nt; r = full + (g2 - c * nt); }
#endif
}

// LDS of the kernels instantiated WITHOUT the whole-body model (kinodynamic / single-rigid-body handles): 8 KB instead of 16 / 40 KB, so that
// their small knots are not held to the whole-body kernels' two waves per SIMD
union RedLds { HkdLds h; SrbLds s; };
// slot_list / nlist: the slots this launch covers (null: all nslots) - when the lane-quad kernel takes the whole-body running knots of a launch,
// the one-wave programs only see the rest (terminal knots with their reset maps, single-rigid-body knots); unit_knots: running knots among them
#define ROLL_ARGS const PhaseDev* ph_, int nph, const int* slot_phase, const int* slot_k, int nslots, int batch, ModelDev md, EpsList el, OptDev opt, const double* x0, SlotArrays sa, \
                  const ProbState* st, int mask, int* fail, unsigned long long* units, const int* slot_list, int nlist, int unit_knots
template <bool WBM, class LDS> __device__ __forceinline__ void rollout_body(LDS& L, ROLL_ARGS) {
    PhaseC* ph = (PhaseC*)ph_;   // descriptors: constant memory, scalar loads
    const int per = batch * nlist;
    int c, r; cand_unit(per, blockIdx.x, c, r);
    const int b = r / nlist, si = r - b * nlist;
    if (masked_out(st[b], mask)) return;
    if (si == 0 && threadIdx.x == 0) atomicAdd(units, (unsigned long long)unit_knots);     // knots this launch rolls out (measurement only)
    const int s = slot_list != nullptr ? slot_list[si] : si;
    const int pi = slot_phase[s], k = slot_k[s];
    PhaseC& P = ph[pi];
    const size_t cbase = (size_t)c * batch * nslots;            // slice of candidate c in the slot arrays
    SlotOut so{sa.cost, sa.dsq, sa.ming, sa.maxh};
    const size_t slot_base = cbase + (size_t)b * nslots, slot = slot_base + s;
    const double eps = el.from_state ? st[b].ls_eps : el.e[c];
    const bool wr = (c == el.writer);
    fail += (size_t)c * batch;
    int chain_first = -1;
    if (!ph[0].shooting) {       // single shooting over the whole horizon (option.MS = false): the wave of slot 0 walks every phase
        if (s != 0) return;
        const int n0 = ph[0].n;
        if (WBM && ph[0].model == HSDDP_MODEL_WB) { if constexpr (WBM) { HS_PHASE(64, if (tid < 36) L.xnext[tid] = x0[(size_t)b * 36 + tid];) } }
        else { HS_PHASE(64, if (tid < n0) ph[0].Xsim[(size_t)b * (ph[0].h + 1) * n0 + tid] = x0[(size_t)b * n0 + tid];) }
        chain_first = 0;
    } else {
        if (!P.shooting) return;     // a phase without shooting nodes is rolled sequentially by the wave of its predecessor's terminal knot
        PhaseC* Pn = pi + 1 < nph ? &ph[pi + 1] : nullptr;
        if (k == P.h) chain_first = pi + 1;
        if (P.model == HSDDP_MODEL_HKD) {
            HkdLds& Lh = *reinterpret_cast<HkdLds*>(&L);
            if (k < P.h) hkd_rollout_knot<64>(Lh, P, b, k, eps, opt.ReB_active, pi == 0 ? x0 : nullptr, so, slot, fail, false, wr);
            else hkd_rollout_terminal<64>(Lh, P, Pn, md, b, eps, opt.AL_active, so, slot, false, wr);
        } else if (P.model == HSDDP_MODEL_SRB) {   // reduced-model tail of the MHPC horizon: a few hundred flops per knot, reuses the whole-body LDS block
            SrbLds& Ls = *reinterpret_cast<SrbLds*>(&L);
            if (k < P.h) srb_rollout_knot<64>(Ls, P, b, k, eps, opt.ReB_active, pi == 0 ? x0 : nullptr, so, slot, fail, false, wr);
            else srb_rollout_terminal<64>(Ls, P, Pn, b, eps, so, slot, false, wr);
        } else if constexpr (WBM) {
            if (k < P.h) wb_rollout_knot<64>(L, P, md, b, k, eps, opt.ReB_active, pi == 0 ? x0 : nullptr, so, slot, fail, false, wr);
            else wb_rollout_terminal<64>(L, P, Pn, md, b, eps, opt.AL_active, so, slot, false, wr);
        }
    }
    // single-shooting phases from here on (young phases behind a terminal knot, or the whole horizon): ONE call site, one copy of the code
    if (chain_first >= 0 && chain_first < nph && !ph[chain_first].shooting) rollout_chain<WBM>(L, ph, nph, chain_first, md, b, nslots, eps, opt, so, slot_base, fail, wr);
}
#define ROLL_PASS ph_, nph, slot_phase, slot_k, nslots, batch, md, el, opt, x0, sa, st, mask, fail, units, slot_list, nlist, unit_knots
__global__ void __launch_bounds__(64) ROLL_ATTR k_rollout(ROLL_ARGS) { __shared__ WbCore L; rollout_body<true>(L, ROLL_PASS); }
// handles without whole-body phases (has_hkd): no whole-body code, 8 KB of LDS, no register cap
// (measured on config 5, waves per SIMD rollout / LQ: none/none 111.6 k it/s, 3/- 117.7 k, 4/- 121.9 k, 4/5 122.1 k, 5/5 122.5 k; 8 KB of LDS allow five)
#ifndef ROLL_HKD_WPE
#define ROLL_HKD_WPE 4
#endif
#ifdef ROLL_HKD_WPE
#define ROLL_HKD_ATTR __attribute__((amdgpu_waves_per_eu(ROLL_HKD_WPE, ROLL_HKD_WPE)))
#else
#define ROLL_HKD_ATTR
#endif
__global__ void __launch_bounds__(64) ROLL_HKD_ATTR k_rollout_hkd(ROLL_ARGS) { __shared__ RedLds L; rollout_body<false>(L, ROLL_PASS); }

// The whole-body running knots of the phases with shooting nodes on LANE QUADS (wb_quad.hpp): one lane per leg, sixteen problems of the same
// (candidate, knot) per wave.  grid = candidates x knots of the list x ceil(batch / 16); qslots: the slots this kernel owns.
#ifndef QUAD_WPE
#define QUAD_WPE 1      // waves per SIMD the quad kernel is compiled for (1: up to 512 registers, nothing in scratch; 2: 256 registers)
#endif
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(QUAD_WPE, QUAD_WPE)))
k_rollout_quad(const PhaseDev* ph_, const int* slot_phase, const int* slot_k, const int* qslots, int nq, int nslots, int batch, ModelDev md, EpsList el, OptDev opt, const double* x0,
               SlotArrays sa, const ProbState* st, int mask, int* fail, unsigned long long* units, const int* plist, int nlist) {
    PhaseC* ph = (PhaseC*)ph_;
    // plist / nlist: the problems this launch is for (null: all of the batch, `mask` picks).  A probe launch of a line search or a commit launch only
    // concerns some problems; packed sixteen to a wave from the list the deciding kernel left behind, its waves are full whatever the share is
    const int nprob = plist != nullptr ? nlist : batch;
    const int nbg = (nprob + 15) >> 4;
    const int per = nq * nbg;
    int c, r; cand_unit(per, blockIdx.x, c, r);
    const int qi = r / nbg, bg = r - qi * nbg;
    const int s = qslots[qi], pi = slot_phase[s], k = slot_k[s];
    const int ix = bg * 16 + (threadIdx.x >> 2);
    const int b = ix < nprob ? (plist != nullptr ? plist[ix] : ix) : batch;
    const bool active = b < batch && !masked_out(st[b < batch ? b : 0], mask);
    {   // knots this launch rolls out (measurement only): one atomic per wave
        const unsigned long long m = __ballot(active && (threadIdx.x & 3) == 0);
        if (threadIdx.x == 0 && m != 0) atomicAdd(units, (unsigned long long)__popcll(m));
    }
    if (!active) return;      // (a quad leaves or stays as a whole: the cross-lane steps below need all four lanes)
    const double eps = el.from_state ? st[b].ls_eps : el.e[c];
    const QuadOut q = wbq_rollout_knot<QD>(ph[pi], md, b, k, eps, opt.ReB_active, pi == 0 ? x0 : nullptr, c == el.writer);
    if ((threadIdx.x & 3) == 0) {
        const size_t slot = ((size_t)c * batch + b) * nslots + s;
        sa.cost[slot] = q.cost; sa.dsq[slot] = q.dsq; sa.ming[slot] = q.ming; sa.maxh[slot] = 0.0;
        if (q.bad) fail[(size_t)c * batch + b] = 1;
    }
}

// Batched line search, decision step (MultiPhaseDDP::line_search, MultiPhaseDDP.cpp:108-131, for the candidates of one probe launch): per
// problem the candidates are examined IN ORDER - reduction of the slice's partials, merit, Armijo test - exactly as if they had been
// rolled out one after the other; the first accepted one ends the search.  ls_eps = the step whose trajectories the problem must hold
// afterwards (the accepted one, else the last one tried: quirk vi); need_commit = those trajectories still have to be written (the
// launch's writer was another candidate).
__global__ void __launch_bounds__(64) k_ls_pick(int nslots, int batch, SlotArrays sp, SlotArrays sa, EpsList el, int last_chunk, ProbState* st, OptDev opt, const int* fail, int* counters,
                                               int* ls_list, int* commit_list) {
    const int b = blockIdx.x, tid = threadIdx.x;
    ProbState& s = st[b];
    if (!s.ls_active) return;
    __shared__ double rc[64], rd[64], rg[64], rh[64];
    __shared__ int accepted;
    if (tid == 0) accepted = -1;
    __syncthreads();
    const size_t per = (size_t)batch * nslots;
    for (int c = 0; c < el.n; c++) {
        double cs = 0, d = 0, g = 0, hh = 0;
        for (int i = tid; i < nslots; i += 64) { const size_t j = c * per + (size_t)b * nslots + i; cs += sp.cost[j]; d += sp.dsq[j]; g = fmin(g, sp.ming[j]); hh = fmax(hh, sp.maxh[j]); }
        rc[tid] = cs; rd[tid] = d; rg[tid] = g; rh[tid] = hh;
        __syncthreads();
        for (int o = 32; o > 0; o >>= 1) { if (tid < o) { rc[tid] += rc[tid + o]; rd[tid] += rd[tid + o]; rg[tid] = fmin(rg[tid], rg[tid + o]); rh[tid] = fmax(rh[tid], rh[tid + o]); } __syncthreads(); }
        if (tid == 0) {
            const double eps = el.e[c];
            s.ls_total++;
            const bool rollout_success = fail[(size_t)c * batch + b] == 0;
            s.actual_cost = rc[0]; s.max_pconstr = rg[0]; s.max_tconstr = rh[0];
            s.feas = sqrt(rd[0]);
            s.merit = s.actual_cost + s.merit_rho * s.feas;
            const double exp_cost_change = eps * s.dV_1 + 0.5 * eps * eps * s.dV_2;
            const double exp_merit_change = exp_cost_change - eps * s.merit_rho * s.feas_prev;
            s.ls_eps = eps;
            if ((s.merit <= s.merit_prev + opt.gamma * exp_merit_change) && rollout_success) { s.ls_success = 1; s.ls_active = 0; accepted = c; }
        }
        __syncthreads();
        if (accepted >= 0) break;
    }
    // trajectories on the device: the writer's (last candidate of the last chunk).  They are the right ones iff the search ends on it; its
    // per-slot partials then become the problem's current ones (the next iteration's compute_cost / feasibility reduce the base arrays).
    const bool ends_here = accepted >= 0 || last_chunk;
    const int final_c = accepted >= 0 ? accepted : el.n - 1;
    const bool commit = ends_here && final_c != el.writer;
    if (ends_here && !commit) {
        for (int i = tid; i < nslots; i += 64) {
            const size_t j = final_c * per + (size_t)b * nslots + i, d = (size_t)b * nslots + i;
            sa.cost[d] = sp.cost[j]; sa.dsq[d] = sp.dsq[j]; sa.ming[d] = sp.ming[j]; sa.maxh[d] = sp.maxh[j];
        }
    }
    if (tid == 0) {
        s.need_commit = commit ? 1 : 0;
        if (commit) commit_list[atomicAdd(&counters[3], 1)] = b;
        if (s.ls_active) ls_list[atomicAdd(&counters[2], 1)] = b;
    }
}

#define LQ_ARGS const PhaseDev* ph_, int nph, const int* slot_phase, const int* slot_k, int nslots, ModelDev md, OptDev opt, const ProbState* st, int mask, int use_cache, unsigned long long* units
template <bool WBM, int NT, class LDS> __device__ __forceinline__ void lq_body(LDS& L, LQ_ARGS) {
    PhaseC* ph = (PhaseC*)ph_;   // descriptors: constant memory, scalar loads
    const int b = blockIdx.x / nslots, s = blockIdx.x % nslots;
    if (masked_out(st[b], mask)) return;
    if (s == 0 && threadIdx.x == 0) atomicAdd(units, (unsigned long long)nslots - nph);
    const int pi = slot_phase[s], k = slot_k[s];
    PhaseC& P = ph[pi];
    if (P.model == HSDDP_MODEL_HKD) {
        HkdLds& Lh = *reinterpret_cast<HkdLds*>(&L);
        if (k < P.h) hkd_lq_knot<NT>(Lh, P, b, k, opt.ReB_active); else hkd_lq_terminal<NT>(Lh, P, pi + 1 < nph ? &ph[pi + 1] : nullptr, md, b, opt.AL_active);
        return;
    }
    if (P.model == HSDDP_MODEL_SRB) {
        SrbLds& Ls = *reinterpret_cast<SrbLds*>(&L);
        if (k < P.h) srb_lq_knot<NT>(Ls, P, b, k, opt.ReB_active); else srb_lq_terminal<NT>(Ls, P, pi + 1 < nph ? &ph[pi + 1] : nullptr, b);
        return;
    }
    if constexpr (WBM) {
        if (k < P.h) wb_lq_knot<NT>(L, P, md, b, k, opt.ReB_active, use_cache != 0);
        else wb_lq_terminal<NT>(L, P, pi + 1 < nph ? &ph[pi + 1] : nullptr, md, b, opt.AL_active);
    }
}
#define LQ_PASS ph_, nph, slot_phase, slot_k, nslots, md, opt, st, mask, use_cache, units
__global__ void __launch_bounds__(LQ_NT) LQ_ATTR k_lq(LQ_ARGS) { __shared__ WbLqLds L; lq_body<true, LQ_NT>(L, LQ_PASS); }
// handles without whole-body phases: one wave per knot, 8 KB of LDS, no register cap
#ifndef LQ_HKD_WPE
#define LQ_HKD_WPE 5
#endif
#ifdef LQ_HKD_WPE
#define LQ_HKD_ATTR __attribute__((amdgpu_waves_per_eu(LQ_HKD_WPE, LQ_HKD_WPE)))
#else
#define LQ_HKD_ATTR
#endif
__global__ void __launch_bounds__(64) LQ_HKD_ATTR k_lq_hkd(LQ_ARGS) { __shared__ RedLds L; lq_body<false, 64>(L, LQ_PASS); }

// cost-only refresh from stored g / h with the CURRENT ReB / AL parameters (SinglePhase::compute_cost, SinglePhase.cpp:236-262)
__global__ void __launch_bounds__(64) k_cost(const PhaseDev* ph_, const int* slot_phase, const int* slot_k, int nslots, OptDev opt, SlotArrays sa,
                                            const ProbState* st, int mask) {
    PhaseC* ph = (PhaseC*)ph_;   // descriptors: constant memory, scalar loads
    const int b = blockIdx.x / nslots, s = blockIdx.x % nslots;
    if (masked_out(st[b], mask)) return;
    __shared__ double bar[MAXG];
    PhaseC& P = ph[slot_phase[s]]; const int k = slot_k[s]; const int tid = threadIdx.x;
    const size_t slot = (size_t)b * nslots + s;
    if (k < P.h) {
        const size_t kk = (size_t)b * P.h + k;
        for (int c = tid; c < P.ng; c += 64) { size_t gi = kk * P.ng + c; bar[c] = P.eps[gi] * reb_barrier(P.g[gi], P.delta[gi]); }
        __syncthreads();
        if (tid == 0) {
            double l = P.lbase[kk];
            if (opt.ReB_active) {
                int offs[5], sz[5]; const int nobj = constraint_objects(P, offs, sz);
                for (int gI = 0; gI < nobj; gI++) { double c = 0; for (int i = 0; i < sz[gI]; i++) c += bar[offs[gI] + i]; l += P.dt * c; }
            }
            P.l[kk] = l; sa.cost[slot] = l;
        }
    } else if (tid == 0) {
        double Phi = P.Phibase[b], c = 0;
        for (int i = 0; i < P.nt; i++) { double hh = P.th[(size_t)b * P.nt + i]; c += 0.5 * P.sigma[(size_t)b * P.nt + i] * hh * hh; c += P.lambda[(size_t)b * P.nt + i] * hh; }
        if (opt.AL_active && P.nt > 0) Phi += c;
        P.Phi[b] = Phi; sa.cost[slot] = Phi;
    }
}

template <class R, int SET, class LDS>
__device__ __forceinline__ void sweep_body(LDS& S, const PhaseDev* ph, int nph, const OptDev& opt, ProbState* st, double fixed_reg, int regularized,
                                           int do_linear, double lin_eps, int* success_out) {
    const int b = blockIdx.x;
    bool success = false;
    if (regularized) {   // MultiPhaseDDP::backward_sweep_regularized (MultiPhaseDDP.cpp:136-165)
        double reg = st[b].reg; int iter = 0;
        while (true) {
            iter++;
            success = riccati_sweep<SW_NT, R, SET>(S, ph, nph, b, (R)reg);
            if (success) break;
            reg = fmax(reg * opt.update_regularization, 1e-3);
            if (reg > 1e2) break;
        }
        reg = reg / 20; if (reg < 1e-6) reg = 0;
        if (threadIdx.x == 0) { st[b].reg = reg; st[b].reg_total += iter; st[b].bs_ok = success ? 1 : 0; }
    } else {
        success = riccati_sweep<SW_NT, R, SET>(S, ph, nph, b, (R)fixed_reg);
        if (threadIdx.x == 0 && success_out) success_out[b] = success ? 1 : 0;
    }
    if (success && do_linear) linear_rollout<SW_NT, R, SET>(S, ph, nph, b, (R)lin_eps);
    __syncthreads();
    if (threadIdx.x == 0) { st[b].dV_1 = S.c.dV1; st[b].dV_2 = S.c.dV2; }
}
// One sweep kernel per (scalar type, model set): the 24-row factor of the kinodynamic model needs twice the registers of the 12-row ones,
// and a kernel that carries both spills in every instantiation (whole-body kernel alone: 230 registers, no scratch, 150 SGPR spills;
// with the 24/24/0 phases in the same kernel: 256 + scratch, 850 SGPR spills).
#ifndef SW_WB_WAVES
#define SW_WB_WAVES 3      // waves per SIMD the whole-body sweep is compiled for: 50 KB of LDS per workgroup -> three workgroups per CU (168 registers)
#endif
#define SWEEP_KERNEL(NAME, R_, SET_, LDS_, MINW_) \
__global__ void __launch_bounds__(SW_NT, MINW_) NAME(const PhaseDev* ph_, int nph, OptDev opt, ProbState* st, int mask, double fixed_reg, int regularized, \
                                                     int do_linear, double lin_eps, int* success_out, unsigned long long* units, int nknots) { \
    const PhaseDev* ph = ph_;   /* (the sweep keeps generic descriptor reads: scalar copies of its fields only add SGPR spills there) */ \
    if (masked_out(st[blockIdx.x], mask)) return; \
    if (threadIdx.x == 0) atomicAdd(units, (unsigned long long)nknots); \
    __shared__ LDS_ S; \
    sweep_body<R_, SET_>(S, ph, nph, opt, st, fixed_reg, regularized, do_linear, lin_eps, success_out); \
}
#define LINEAR_KERNEL(NAME, R_, SET_, LDS_, MINW_) \
__global__ void __launch_bounds__(SW_NT, MINW_) NAME(const PhaseDev* ph_, int nph, ProbState* st, double eps) { \
    const PhaseDev* ph = ph_; \
    __shared__ LDS_ S; \
    linear_rollout<SW_NT, R_, SET_>(S, ph, nph, blockIdx.x, (R_)eps); \
    __syncthreads(); \
    if (threadIdx.x == 0) { st[blockIdx.x].dV_1 = S.c.dV1; st[blockIdx.x].dV_2 = S.c.dV2; } \
}
SWEEP_KERNEL(k_sweep, double, SW_SET_WB, SweepLds, SW_WB_WAVES)           // whole-body (+ SRB tail) phases
#ifndef SW_HKD_WAVES
#define SW_HKD_WAVES 3      // 51 KB of LDS -> three workgroups per CU; measured on config 5 in fp64 (k_sweep_hkd per launch): 2 waves per SIMD 161.8 ms, 3: 115.9
#endif
SWEEP_KERNEL(k_sweep_hkd, double, SW_SET_HKD, SweepLdsHkd, SW_HKD_WAVES)      // kinodynamic 24/24/0 phases, fp64
// fp32 handles (hsddp_create_ex): fp32 LQ records, every product of the Riccati step on v_mfma_f32_16x16x4_f32, an LDS block a third the size
// (kinodynamic 24/24/0 and single-rigid-body phases only: SinglePhase.cpp:565-567, HKDModel.h:33-61)
#ifndef SW_F32_WAVES
#define SW_F32_WAVES 5      // measured on config 5 (k_sweep32 per launch): 3 waves per SIMD 98.3 ms (no spill), 4: 83.6, 5: 76.5 (96 registers), 6: 191
#endif
SWEEP_KERNEL(k_sweep32, float, SW_SET_HKD, SweepLds32, SW_F32_WAVES)
LINEAR_KERNEL(k_linear, double, SW_SET_WB, SweepLds, SW_WB_WAVES)
LINEAR_KERNEL(k_linear_hkd, double, SW_SET_HKD, SweepLdsHkd, SW_HKD_WAVES)
LINEAR_KERNEL(k_linear32, float, SW_SET_HKD, SweepLds32, SW_F32_WAVES)

// receding-horizon shift of one phase (include/hsddp.h hsddp_warm_start_phase / hsddp_reconfigure): one workgroup per (problem, destination knot).
// Trajectories as SinglePhase::pop_front x shift + push_back_default do (SinglePhase.cpp:513-528, TrajectoryManagement.cpp:130-228); the per-knot
// ReB parameters travel with their knots, a pushed knot copies the last knot's (PathConstraintBase::pop_front / push_back,
// ConstraintsBase.h:296-306: reset_params() is a no-op, :192); the AL parameters of the terminal constraint stay with the phase (:375).
__global__ void __launch_bounds__(256) k_warm_start(PhaseDev D, PhaseDev S, int has_src, int shift) {
    const int b = blockIdx.y, k = blockIdx.x, n = D.n, m = D.m, tid = threadIdx.x;
    const size_t dx = ((size_t)b * (D.h + 1) + k) * n, du = ((size_t)b * D.h + k);
    const int ks = k + shift;
    for (int i = tid; i < n; i += blockDim.x) {
        double v = 0.0;
        if (has_src) v = (ks <= S.h) ? S.Xbar[((size_t)b * (S.h + 1) + ks) * n + i] : S.X[((size_t)b * (S.h + 1) + S.h) * n + i];
        D.Xbar[dx + i] = v; D.X[dx + i] = v; D.dX[dx + i] = 0.0;
    }
    if (k < D.h) {
        const bool cs = has_src && ks < S.h; const size_t su = (size_t)b * S.h + ks;
        for (int i = tid; i < m; i += blockDim.x) { const double v = cs ? S.Ubar[su * m + i] : 0.0; D.Ubar[du * m + i] = v; D.U[du * m + i] = v; D.dU[du * m + i] = 0.0; D.KdX[du * m + i] = 0.0; }
        for (int i = tid; i < m * n; i += blockDim.x) D.K[du * m * n + i] = cs ? S.K[su * m * n + i] : 0.0;
        if (has_src && S.ng == D.ng && S.h > 0) {
            const size_t sg = (size_t)b * S.h + (ks < S.h ? ks : S.h - 1);
            for (int i = tid; i < D.ng; i += blockDim.x) { D.eps[du * D.ng + i] = S.eps[sg * S.ng + i]; D.delta[du * D.ng + i] = S.delta[sg * S.ng + i]; }
        }
    }
    if (k == 0 && has_src && S.nt == D.nt) for (int i = tid; i < D.nt; i += blockDim.x) { D.sigma[(size_t)b * D.nt + i] = S.sigma[(size_t)b * S.nt + i]; D.lambda[(size_t)b * D.nt + i] = S.lambda[(size_t)b * S.nt + i]; }
}

// MHPC_Command_lcmt packing (include/hsddp.h): one workgroup per control step, fp64 -> fp32 on the device
__global__ void __launch_bounds__(256) k_pack_command(const PhaseDev* ph, const int* step_phase, const int* step_k, int n_steps, int b, double t0, double dt,
                                                     const float* status, unsigned int* out) {
    const int s = blockIdx.x; const PhaseDev& P = ph[step_phase[s]]; const int k = step_k[s];
    const double* X = P.Xbar + ((size_t)b * (P.h + 1) + k) * 36; const size_t kk = (size_t)b * P.h + k;
    constexpr int NF = 15; const int wd[NF] = {1, 12, 3, 3, 12, 3, 3, 12, 12, 432, 12, 144, 432, 4, 4};
    if (s == 0 && threadIdx.x == 0) out[0] = (unsigned int)n_steps;
    size_t off = 1;
    for (int f = 0; f < NF; f++) {
        unsigned int* dst = out + off + (size_t)s * wd[f];
        for (int e = threadIdx.x; e < wd[f]; e += blockDim.x) {
            float v = 0.f; bool is_int = false; int iv = 0;
            switch (f) {
                case 0: v = (float)(t0 + s * dt); break;
                case 1: v = (float)P.Ubar[kk * 12 + e]; break;
                case 2: v = (float)X[3 + e]; break;
                case 3: v = (float)X[e]; break;
                case 4: v = (float)X[6 + e]; break;
                case 5: v = (float)X[18 + e]; break;
                case 6: v = (float)X[21 + e]; break;
                case 7: v = (float)X[24 + e]; break;
                case 8: v = (float)P.Y[kk * 12 + e]; break;
                case 9: v = (float)P.K[kk * 432 + e]; break;
                case 10: v = (float)P.Qu[kk * 12 + e]; break;
                case 11: v = (float)P.Quu[kk * 144 + e]; break;
                case 12: v = (float)P.Qux[kk * 432 + e]; break;
                case 13: is_int = true; iv = P.contact[e]; break;
                default: v = status ? status[step_phase[s] * 4 + e] : 0.f; break;
            }
            dst[e] = is_int ? (unsigned int)iv : __float_as_uint(v);
        }
        off += (size_t)n_steps * wd[f];
    }
}

// K dX of every whole-body control knot from the CURRENT gains (step API: a backward sweep without the linear rollout behind it leaves new gains
// next to the old search direction dX; SinglePhase::hybrid_rollout would apply the new K to it, SinglePhase.cpp:196-200)
__global__ void __launch_bounds__(64) k_refresh_kdx(const PhaseDev* ph, int nph) {
    const int b = blockIdx.y, tid = threadIdx.x;
    for (int pi = 0; pi < nph; pi++) {
        const PhaseDev& P = ph[pi];
        if (P.model != HSDDP_MODEL_WB) continue;
        for (int k = blockIdx.x; k < P.h; k += gridDim.x) {
            const size_t kk = (size_t)b * P.h + k, kx = ((size_t)b * (P.h + 1) + k) * 36;
            if (tid < 12) { double s = 0; for (int j = 0; j < 36; j++) s += P.K[kk * 432 + tid + 12 * j] * P.dX[kx + j]; P.KdX[kk * 12 + tid] = s; }
        }
    }
}

// X -> Xbar, U -> Ubar, Defect -> Defect_bar (Trajectory::update_nominal_vals, TrajectoryManagement.cpp:122-127)
__global__ void k_update_nominal(const PhaseDev* ph, int nph, const ProbState* st, int mask) {
    const int b = blockIdx.y;
    if (masked_out(st[b], mask)) return;
    for (int pi = 0; pi < nph; pi++) {
        const PhaseDev& P = ph[pi];
        const size_t nx = (size_t)(P.h + 1) * P.n, nu = (size_t)P.h * P.m;
        for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nx; i += (size_t)gridDim.x * blockDim.x) {
            P.Xbar[b * nx + i] = P.X[b * nx + i]; P.Defect_bar[b * nx + i] = P.Defect[b * nx + i];
        }
        for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nu; i += (size_t)gridDim.x * blockDim.x) P.Ubar[b * nu + i] = P.U[b * nu + i];
    }
}

// ReB / AL parameter updates (ConstraintsBase.h:194-209, 375-391) for problems flagged by k_outer_end (ls_success reused as flag? no: own mask)
__global__ void k_update_params(const PhaseDev* ph, int nph, OptDev opt, const ProbState* st, const int* do_update) {
    const int b = blockIdx.y;
    if (!do_update[b]) return;
    for (int pi = 0; pi < nph; pi++) {
        const PhaseDev& P = ph[pi];
        if (opt.ReB_active) {
            const size_t tot = (size_t)P.h * P.ng;
            for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
                const size_t gi = (size_t)b * tot + i; const int c = (int)(i % P.ng);
                if (P.g[gi] > -opt.pconstr_thresh) continue;
                const int grp = constraint_group(P, c);
                P.eps[gi] *= opt.update_ReB;
                P.delta[gi] = fmax(P.delta[gi] * opt.update_relax, P.reb_init[grp][1]);
            }
        }
        if (opt.AL_active && blockIdx.x == 0 && (int)threadIdx.x < P.nt) {
            const size_t ti = (size_t)b * P.nt + threadIdx.x; const double hh = P.th[ti];
            if (!(fabs(hh) < opt.tconstr_thresh)) {
                if (fabs(hh) > 0.005) { P.sigma[ti] = fmin(P.sigma[ti] * opt.update_penalty, P.al_init[2]); }
                else P.lambda[ti] += hh * P.sigma[ti];
            }
        }
    }
}

// Per-problem control: reduction of the per-slot partials + the scalar logic of MultiPhaseDDP::solve / line_search.
enum { EV_REDUCE_ONLY = 0, EV_INIT, EV_OUTER_BEGIN, EV_INNER_BEGIN, EV_PRE_LS, EV_LS_TRIAL, EV_POST_LS, EV_OUTER_END, EV_TIMEOUT };
// history buffers of MultiPhaseDDP (cost / dyn_feas / eqn_feas / ineq_feas_buffer, std::vector<float>, MultiPhaseDDP.h:133-136): [batch][4][cap]
struct HistDev { float* buf; int cap; };
__device__ inline void hist_push(const HistDev& hd, int b, ProbState& s) {      // MultiPhaseDDP.cpp:258-261, 382-385
    const float tc = (float)s.max_tconstr, pc = (float)s.max_pconstr;
    if (hd.buf != nullptr && s.hist_n < hd.cap) {
        float* p = hd.buf + (size_t)b * 4 * hd.cap + s.hist_n;
        p[0] = (float)s.actual_cost; p[hd.cap] = (float)s.feas; p[2 * hd.cap] = tc; p[3 * hd.cap] = pc;
    }
    s.hist_n++;
    s.info_tconstr = tc; s.info_pconstr = pc;      // get_terminal / get_path_constraint_violation() = buffer.back() (MultiPhaseDDP.h:81-83)
}
__global__ void __launch_bounds__(64) k_eval(int mode, int nslots, SlotArrays sa, ProbState* st, OptDev opt, double eps, const int* fail, int* do_update,
                                            int* counters, int iter_ou_host, HistDev hd, int* ls_list) {
    const int b = blockIdx.x, tid = threadIdx.x;
    ProbState& s = st[b];
    __shared__ double rc[64], rd[64], rg[64], rh[64];
    bool need_reduce = (mode == EV_REDUCE_ONLY) || (mode == EV_INIT) || (mode == EV_INNER_BEGIN && s.inner_active) || (mode == EV_LS_TRIAL && s.ls_active);
    double cost = 0, dsq = 0, ming = 0, maxh = 0;
    if (need_reduce) {
        double c = 0, d = 0, g = 0, h = 0;
        for (int i = tid; i < nslots; i += 64) { size_t j = (size_t)b * nslots + i; c += sa.cost[j]; d += sa.dsq[j]; g = fmin(g, sa.ming[j]); h = fmax(h, sa.maxh[j]); }
        rc[tid] = c; rd[tid] = d; rg[tid] = g; rh[tid] = h;
        __syncthreads();
        for (int o = 32; o > 0; o >>= 1) { if (tid < o) { rc[tid] += rc[tid + o]; rd[tid] += rd[tid + o]; rg[tid] = fmin(rg[tid], rg[tid + o]); rh[tid] = fmax(rh[tid], rh[tid + o]); } __syncthreads(); }
        cost = rc[0]; dsq = rd[0]; ming = rg[0]; maxh = rh[0];
    }
    if (tid != 0) return;
    // end of an inner iteration that ran to its last line (MultiPhaseDDP.cpp:382-385): the entry is buffered by the NEXT control step, so
    // that a max_cputime stop at the reference's last checkpoint (:376-380, before the push) can still drop it
    if (s.push_pending) { if (mode != EV_TIMEOUT) hist_push(hd, b, s); s.push_pending = 0; }
    switch (mode) {
    case EV_REDUCE_ONLY: s.actual_cost = cost; s.feas = sqrt(dsq); s.max_pconstr = ming; s.max_tconstr = maxh; break;
    case EV_INIT:   // MultiPhaseDDP.cpp:218-263
        s.actual_cost = cost; s.feas = sqrt(dsq); s.max_pconstr = ming; s.max_tconstr = maxh;
        s.iter = 0; s.ls_total = 0; s.status = 0; s.reg = 0;
        s.outer_active = 1; s.inner_active = 0; s.ls_active = 0; s.ls_success = 0; s.iter_in = 0; s.iter_ou = 0;
        s.hist_n = 0; s.push_pending = 0; hist_push(hd, b, s);
        break;
    case EV_OUTER_BEGIN:   // :267-276
        if (s.outer_active) { s.iter_ou++; s.max_tconstr_prev = s.max_tconstr; s.max_pconstr_prev = s.max_pconstr; s.reg = 0; s.iter_in = 0; s.inner_active = 1; }
        break;
    case EV_INNER_BEGIN:   // :280-285  (compute_cost; feas)
        if (s.inner_active) { s.actual_cost = cost; s.feas = sqrt(dsq); s.iter_in++; s.iter++; s.ls_success = 0; }
        break;
    case EV_PRE_LS:        // :315-349
        s.ls_success = 0; s.need_commit = 0;
        if (s.inner_active) {
            if (!s.bs_ok) { s.status = 1; s.inner_active = 0; s.outer_active = 0; break; }   // bad_solve
            double dV_abs = fabs(s.dV_1 + 0.5 * s.dV_2);
            s.merit_rho = (s.feas > opt.dynamics_feas_thresh) ? dV_abs / ((1 - opt.merit_scale) * s.feas) + opt.merit_offset : 0;
            s.merit = s.actual_cost + s.merit_rho * s.feas;
            s.cost_prev = s.actual_cost; s.merit_prev = s.merit; s.feas_prev = s.feas;
            if ((dV_abs < opt.cost_thresh) && (s.feas <= opt.dynamics_feas_thresh)) { s.inner_active = 0; s.ls_active = 0; }
            else { s.ls_active = 1; s.ls_success = 0; }
        }
        break;
    case EV_LS_TRIAL:      // MultiPhaseDDP::line_search body (:108-131)
        if (s.ls_active) {
            s.ls_total++;
            bool rollout_success = fail[b] == 0;
            s.actual_cost = cost; s.max_pconstr = ming; s.max_tconstr = maxh;
            s.feas = sqrt(dsq);
            s.merit = s.actual_cost + s.merit_rho * s.feas;
            double exp_cost_change = eps * s.dV_1 + 0.5 * eps * eps * s.dV_2;
            double exp_merit_change = exp_cost_change - eps * s.merit_rho * s.feas_prev;
            if ((s.merit <= s.merit_prev + opt.gamma * exp_merit_change) && rollout_success) { s.ls_success = 1; s.ls_active = 0; s.ls_eps = eps; }
        }
        break;
    case EV_POST_LS:       // :356-385
        if (s.inner_active) {
            s.ls_active = 0;
            if (!s.ls_success) { s.actual_cost = s.cost_prev; s.merit = s.merit_prev; }
            if ((fabs((s.cost_prev - s.actual_cost) / s.cost_prev) < opt.cost_thresh) && (s.feas <= opt.dynamics_feas_thresh)) s.inner_active = 0;
            else { s.push_pending = 1; if (s.iter_in >= opt.max_DDP_iter) s.inner_active = 0; }
        }
        break;
    case EV_OUTER_END:     // :394-426
        do_update[b] = 0;
        if (s.outer_active) {
            s.inner_active = 0;
            if (s.max_tconstr < opt.tconstr_thresh && fabs(s.max_pconstr) < opt.pconstr_thresh && s.feas <= opt.dynamics_feas_thresh) s.outer_active = 0;
            else if (fabs(s.max_tconstr - s.max_tconstr_prev) < 0.0001 && fabs(s.max_pconstr - s.max_pconstr_prev) < 0.0001 && s.feas <= opt.dynamics_feas_thresh) s.outer_active = 0;
            else { do_update[b] = 1; if (iter_ou_host >= opt.max_AL_iter) s.outer_active = 0; }
        }
        break;
    case EV_TIMEOUT:
        if (s.outer_active || s.inner_active) { s.status = 2; s.outer_active = 0; s.inner_active = 0; s.ls_active = 0; }
        break;
    }
    if (counters) { if (s.inner_active) atomicAdd(&counters[0], 1); if (s.outer_active) atomicAdd(&counters[1], 1); if (s.ls_active) { const int ix = atomicAdd(&counters[2], 1); if (ls_list) ls_list[ix] = b; } }
}

// ------------------------------------------------------------------------------------------------ host side
#if !defined(__HIP_DEVICE_COMPILE__)   // the device pass sees address-space qualified descriptor pointers (HS_GLOBAL)
struct DevBuf { void* p = nullptr; size_t bytes = 0; };

struct hsddp_handle {
    int nph = 0, batch = 0, device = 0, nslots = 0;
    bool has_hkd = false;             // any kinodynamic phase: the sweep kernels instantiated for {HKD, SRB} serve the handle
    bool f32 = false;                 // HSDDP_PREC_F32: fp32 LQ records + fp32 Riccati sweep / linear rollout (kinodynamic and SRB phases)
    std::vector<PhaseDev> ph;         // host copy (device pointers inside)
    PhaseDev* d_ph = nullptr;
    PhaseDev* d_ph_ss = nullptr;      // the same descriptors with every shooting flag cleared: what option.MS = false rolls out (MultiPhaseDDP.cpp:65-68)
    int *d_slot_phase = nullptr, *d_slot_k = nullptr, *d_fail = nullptr, *d_do_update = nullptr, *d_counters = nullptr, *d_success = nullptr;
    int *d_ls_list = nullptr, *d_commit_list = nullptr;      // problems still searching / needing a commit rollout, in the order the deciding kernel met them (compact grids for the quad kernel)
    int *d_qslots = nullptr, *d_oslots = nullptr; int nq = 0, n_other = 0, other_knots = 0;      // slots of the lane-quad kernel (whole-body running knots of phases with shooting nodes) / the rest
    bool ls_speculate = true;         // HSDDP_LS_SPECULATE=0: the full step of every line search is rolled out on its own (see hsddp_solve)
    int ls_chunk = MAXCAND;           // candidates per probe launch (HSDDP_LS_CHUNK): problems that accept inside a chunk skip the later chunks, at one more launch + decision step per chunk
    bool ls_probe_first = false;      // what the previous search of this handle suggests for the next one
    bool quad = true;                 // the lane-quad kernel takes its slots of every multiple-shooting rollout launch (HSDDP_QUAD=0: the one-wave programs everywhere)
    int* h_counters = nullptr;        // pinned
    SlotArrays sp{}; int sp_cands = 0;       // slot partials of the candidates of a batched line-search launch: [sp_cands][batch][nslots] (allocated on first use)
    bool probe_ok = true;                    // every phase without shooting nodes is a whole-body phase (their chain keeps its state in LDS: probes need no trajectory store)
    unsigned long long* d_units = nullptr;   // knots processed by k_rollout / k_lq / k_sweep launches since the last reset (measurement)
    float* d_hist = nullptr; int hist_cap = 0;      // history buffers [batch][4][hist_cap]
    unsigned int* d_cmd = nullptr; size_t cmd_words = 0; int* d_cmd_map = nullptr; int cmd_steps = 0; float* d_cmd_status = nullptr;   // export staging (kept across calls)
    ProbState* d_st = nullptr;
    double* d_x0 = nullptr;
    SlotArrays sa{};
    ModelDev md{};
    std::vector<void*> allocs;        // buffers that live as long as the handle
    std::vector<void*> gen_allocs;    // phase storage made by hsddp_create (one hipMalloc per array); released by the first hsddp_reconfigure
    struct Arena { char* base = nullptr; size_t cap = 0, used = 0; } arena[2];      // phase storage of hsddp_reconfigure: the new window is laid out
    int cur_arena = -1;               //   in one arena while the other (or gen_allocs) still holds the old window; -1: none in use yet
    std::vector<std::vector<char>> staged;   // host copies of the uploads of the last reconfigure (kept until the next one: async H2D sources)
    int slots_cap = 0, nph_cap = 0;   // capacity of the slot / descriptor tables
    hipStream_t stream = nullptr;
    float solve_ms = 0;
    bool cache_valid = false;         // every problem has been rolled out since its trajectories were last set from outside: P.kc matches X, U
    // kernel timing
    std::vector<std::string> kname; std::vector<double> kms; std::vector<long long> kcnt;
    struct Ev { hipEvent_t a, b; int id; };
    std::vector<Ev> pending;
    std::vector<hipEvent_t> pool;
    bool timing = true;
};

static int kid(hsddp_handle* h, const char* n) {
    for (size_t i = 0; i < h->kname.size(); i++) if (h->kname[i] == n) return (int)i;
    h->kname.push_back(n); h->kms.push_back(0); h->kcnt.push_back(0); return (int)h->kname.size() - 1;
}
static hipEvent_t get_event(hsddp_handle* h) { if (!h->pool.empty()) { hipEvent_t e = h->pool.back(); h->pool.pop_back(); return e; } hipEvent_t e; hipEventCreate(&e); return e; }
struct Timed {
    hsddp_handle* h; int id; hipEvent_t a, b;
    Timed(hsddp_handle* h_, const char* n) : h(h_) { if (h->timing) { id = kid(h, n); a = get_event(h); b = get_event(h); hipEventRecord(a, h->stream); } }
    ~Timed() { if (h->timing) { hipEventRecord(b, h->stream); h->pending.push_back({a, b, id}); } }
};
static void drain_events(hsddp_handle* h) {
    for (auto& e : h->pending) { float ms = 0; hipEventSynchronize(e.b); hipEventElapsedTime(&ms, e.a, e.b); h->kms[e.id] += ms; h->kcnt[e.id]++; h->pool.push_back(e.a); h->pool.push_back(e.b); }
    h->pending.clear();
}

template <class T> static int dalloc(hsddp_handle* h, T** out, size_t count, bool zero = true) {
    void* p = nullptr; size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) { fprintf(stderr, "[hsddp_hip] hipMalloc(%zu) failed: %s\n", bytes, hipGetErrorString(e)); return HSDDP_ENOMEM; }
    h->allocs.push_back(p);
    if (zero && (e = hipMemset(p, 0, bytes)) != hipSuccess) { fprintf(stderr, "[hsddp_hip] hipMemset(%zu) failed: %s\n", bytes, hipGetErrorString(e)); return HSDDP_ENODEV; }
    *out = (T*)p; return HSDDP_OK;
}

// record 0 -> records 1..count-1 by doubling device-to-device copies (log2(count) calls instead of `count`)
static hipError_t dev_replicate(void* base, size_t one, size_t count) {
    size_t have = 1;
    while (have < count) {
        size_t n = std::min(have, count - have);
        hipError_t e = hipMemcpy((char*)base + have * one, base, n * one, hipMemcpyDeviceToDevice); if (e != hipSuccess) return e;
        have += n;
    }
    return hipSuccess;
}

// slots of a window by rollout program: the lane-quad kernel owns the running knots of whole-body phases with shooting nodes, the one-wave
// programs everything else (terminal knots and their reset maps, single-rigid-body / kinodynamic knots, phases without shooting nodes)
static void split_slots(const std::vector<PhaseDev>& ph, const std::vector<int>& sp, const std::vector<int>& sk, std::vector<int>& qs, std::vector<int>& os, int& other_knots) {
    qs.clear(); os.clear(); other_knots = 0;
    for (size_t s = 0; s < sp.size(); s++) {
        const PhaseDev& P = ph[sp[s]];
        if (P.model == HSDDP_MODEL_WB && P.shooting && sk[s] < P.h) qs.push_back((int)s);
        else { os.push_back((int)s); if (sk[s] < P.h) other_knots++; }
    }
}

static OptDev to_dev(const hsddp_option_t& o) {
    OptDev d; d.alpha = o.alpha; d.gamma = o.gamma; d.update_penalty = o.update_penalty; d.update_relax = o.update_relax;
    d.update_regularization = o.update_regularization; d.update_ReB = o.update_ReB; d.max_DDP_iter = o.max_DDP_iter; d.max_AL_iter = o.max_AL_iter;
    d.cost_thresh = o.cost_thresh; d.tconstr_thresh = o.tconstr_thresh; d.pconstr_thresh = o.pconstr_thresh; d.dynamics_feas_thresh = o.dynamics_feas_thresh;
    d.merit_scale = o.merit_scale; d.merit_offset = o.merit_offset; d.AL_active = o.AL_active; d.ReB_active = o.ReB_active; d.MS = o.MS; return d;
}

extern "C" {

const char* hsddp_backend_name(void) { return "hip-gfx950"; }

void hsddp_destroy(hsddp_handle_t* h) {
    if (!h) return;
    hipSetDevice(h->device);
    if (h->stream) hipStreamSynchronize(h->stream);
    drain_events(h);
    for (void* p : h->allocs) hipFree(p);
    for (void* p : h->gen_allocs) hipFree(p);
    for (auto& a : h->arena) if (a.base) hipFree(a.base);
    if (h->d_hist) hipFree(h->d_hist);
    { double* p[4] = {h->sp.cost, h->sp.dsq, h->sp.ming, h->sp.maxh}; for (auto q : p) if (q) hipFree(q); }
    if (h->d_cmd) hipFree(h->d_cmd);
    if (h->d_cmd_map) hipFree(h->d_cmd_map);
    if (h->d_cmd_status) hipFree(h->d_cmd_status);
    for (auto e : h->pool) hipEventDestroy(e);
    if (h->h_counters) hipHostFree(h->h_counters);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
}

// HIP memory policy of setup_phase (hs_host.hpp): every call checked, the first failure is remembered
struct HipMem {
    hsddp_handle* h; hipError_t err = hipSuccess;
    void fail(hipError_t e, const char* what, size_t bytes) { if (err == hipSuccess) { err = e; fprintf(stderr, "[hsddp_hip] %s(%zu) failed: %s\n", what, bytes, hipGetErrorString(e)); } }
    void* alloc(size_t bytes) {
        void* p = nullptr; bytes = std::max<size_t>(bytes, 8);
        hipError_t e = hipMalloc(&p, bytes); if (e != hipSuccess) { fail(e, "hipMalloc", bytes); return nullptr; }
        h->gen_allocs.push_back(p);
        e = hipMemset(p, 0, bytes); if (e != hipSuccess) fail(e, "hipMemset", bytes);
        return p;
    }
    void upload(void* dst, const void* src, size_t bytes) { hipError_t e = hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice); if (e != hipSuccess) fail(e, "hipMemcpy H2D", bytes); }
    void replicate(void* base, size_t one, size_t count) { hipError_t e = dev_replicate(base, one, count); if (e != hipSuccess) fail(e, "hipMemcpy D2D", one * count); }
};

int hsddp_create_ex(hsddp_handle_t** out, int n_phases, const hsddp_phase_desc_t* phases, const hsddp_model_param_t* mp, int batch, int device, int precision) {
    if (!out || n_phases <= 0 || !phases || batch <= 0 || (precision != HSDDP_PREC_F64 && precision != HSDDP_PREC_F32)) return HSDDP_EINVAL;
    for (int i = 0; i < n_phases; i++) {
        if (precision == HSDDP_PREC_F32 && phases[i].model == HSDDP_MODEL_WB) { fprintf(stderr, "[hsddp_hip] HSDDP_PREC_F32 covers kinodynamic (HKD) and single-rigid-body phases; whole-body phases need fp64\n"); return HSDDP_ENOTSUP; }
        if (phases[i].model != HSDDP_MODEL_WB && phases[i].model != HSDDP_MODEL_SRB && phases[i].model != HSDDP_MODEL_HKD) return HSDDP_EINVAL;
        if (i > 0 && !phase_chain_ok(phases[i - 1].model, phases[i].model)) { fprintf(stderr, "[hsddp_hip] phase %d: the reference has no reset map from model %d to model %d (MHPCReset.cpp:4-52, HKDReset.h)\n", i, phases[i - 1].model, phases[i].model); return HSDDP_ENOTSUP; }
        if (!phases[i].shooting && i == 0) { fprintf(stderr, "[hsddp_hip] phase 0 must have shooting nodes (single shooting over the whole horizon is option.MS = 0)\n"); return HSDDP_ENOTSUP; }
        if (phases[i].horizon <= 0) return HSDDP_EINVAL;
    }
    int ndev = 0; if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device || device < 0) { fprintf(stderr, "[hsddp_hip] no HIP device %d\n", device); return HSDDP_ENODEV; }
    HIPCK(hipSetDevice(device));
    hsddp_handle* h = new hsddp_handle();
    h->nph = n_phases; h->batch = batch; h->device = device; h->f32 = precision == HSDDP_PREC_F32;
    // from here on every failure goes through hsddp_destroy(h): nothing allocated so far is leaked
#define CREATE_CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[hsddp_hip] %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); hsddp_destroy(h); return HSDDP_ENODEV; } } while (0)
    CREATE_CK(hipStreamCreate(&h->stream));
    double pd = mp ? mp->psi_dyn : 3.1415, pk = mp ? mp->psi_kin : M_PI;
    h->md = {cos(pd), sin(pd), cos(pk), sin(pk)};
    h->ph.resize(n_phases);
    std::vector<int> sp, sk;
    const size_t B = batch;
    int rc = 0;
    HipMem mem{h};
    for (int i = 0; i < n_phases && !rc; i++) {
        rc = setup_phase(mem, phases[i], i + 1 < n_phases ? &phases[i + 1] : nullptr, i == n_phases - 1, B, h->ph[i], (int)sp.size(), h->f32);
        if (!rc && mem.err != hipSuccess) rc = (mem.err == hipErrorOutOfMemory) ? HSDDP_ENOMEM : HSDDP_ENODEV;
        for (int k = 0; k <= phases[i].horizon; k++) { sp.push_back(i); sk.push_back(k); }
    }
    h->nslots = (int)sp.size();
    h->slots_cap = h->nslots + 16; h->nph_cap = n_phases + 8;      // slack: a receding-horizon update adds or drops a phase (one slot) now and then
    if (!rc) rc |= dalloc(h, &h->d_ph, h->nph_cap); if (!rc) rc |= dalloc(h, &h->d_ph_ss, h->nph_cap);
    if (!rc) rc |= dalloc(h, &h->d_slot_phase, h->slots_cap); if (!rc) rc |= dalloc(h, &h->d_slot_k, h->slots_cap);
    if (!rc) rc |= dalloc(h, &h->d_qslots, h->slots_cap); if (!rc) rc |= dalloc(h, &h->d_oslots, h->slots_cap);
    if (!rc) rc |= dalloc(h, &h->d_fail, B * MAXCAND); if (!rc) rc |= dalloc(h, &h->d_do_update, B); if (!rc) rc |= dalloc(h, &h->d_counters, 4); if (!rc) rc |= dalloc(h, &h->d_ls_list, B); if (!rc) rc |= dalloc(h, &h->d_commit_list, B); if (!rc) rc |= dalloc(h, &h->d_success, B);
    if (!rc) rc |= dalloc(h, &h->d_st, B); if (!rc) rc |= dalloc(h, &h->d_x0, B * h->ph[0].n); if (!rc) rc |= dalloc(h, &h->d_units, 8);
    if (!rc) rc |= dalloc(h, &h->sa.cost, B * h->slots_cap); if (!rc) rc |= dalloc(h, &h->sa.dsq, B * h->slots_cap);
    if (!rc) rc |= dalloc(h, &h->sa.ming, B * h->slots_cap); if (!rc) rc |= dalloc(h, &h->sa.maxh, B * h->slots_cap);
    if (rc) { hsddp_destroy(h); return rc; }
    for (int i = 0; i < n_phases; i++) if (!phases[i].shooting && phases[i].model != HSDDP_MODEL_WB) h->probe_ok = false;
    for (int i = 0; i < n_phases; i++) if (phases[i].model == HSDDP_MODEL_HKD) h->has_hkd = true;
    std::vector<PhaseDev> ss = h->ph; for (auto& q : ss) q.shooting = 0;
    CREATE_CK(hipMemcpy(h->d_ph, h->ph.data(), sizeof(PhaseDev) * n_phases, hipMemcpyHostToDevice));
    CREATE_CK(hipMemcpy(h->d_ph_ss, ss.data(), sizeof(PhaseDev) * n_phases, hipMemcpyHostToDevice));
    CREATE_CK(hipMemcpy(h->d_slot_phase, sp.data(), sp.size() * 4, hipMemcpyHostToDevice));
    CREATE_CK(hipMemcpy(h->d_slot_k, sk.data(), sk.size() * 4, hipMemcpyHostToDevice));
    {
        std::vector<int> qs, os; split_slots(h->ph, sp, sk, qs, os, h->other_knots);
        h->nq = (int)qs.size(); h->n_other = (int)os.size();
        if (h->nq) CREATE_CK(hipMemcpy(h->d_qslots, qs.data(), qs.size() * 4, hipMemcpyHostToDevice));
        if (h->n_other) CREATE_CK(hipMemcpy(h->d_oslots, os.data(), os.size() * 4, hipMemcpyHostToDevice));
        const char* e = getenv("HSDDP_QUAD"); h->quad = !(e && e[0] == '0');
        const char* e2 = getenv("HSDDP_LS_SPECULATE"); h->ls_speculate = !(e2 && e2[0] == '0');
        const char* e3 = getenv("HSDDP_LS_CHUNK"); if (e3 && atoi(e3) >= 1) h->ls_chunk = std::min(atoi(e3), MAXCAND);
    }
    CREATE_CK(hipHostMalloc((void**)&h->h_counters, 4 * sizeof(int))); for (int q = 0; q < 4; q++) h->h_counters[q] = 0;
    CREATE_CK(hipDeviceSynchronize());
#undef CREATE_CK
    *out = h; return HSDDP_OK;
}
int hsddp_create(hsddp_handle_t** out, int n_phases, const hsddp_phase_desc_t* phases, const hsddp_model_param_t* mp, int batch, int device) {
    return hsddp_create_ex(out, n_phases, phases, mp, batch, device, HSDDP_PREC_F64);
}
int hsddp_precision(hsddp_handle_t* h) { return (h && h->f32) ? HSDDP_PREC_F32 : HSDDP_PREC_F64; }

// setup_phase memory policies of hsddp_reconfigure: (1) SizeMem walks the layout without touching memory, (2) ArenaMem lays the window out in
// one arena by bumping a pointer; zero-fill, uploads and replications are queued and issued afterwards on the handle's stream
struct SizeMem {
    size_t used = 0;
    void* alloc(size_t bytes) { bytes = (std::max<size_t>(bytes, 8) + 255) / 256 * 256; void* p = (void*)(uintptr_t)(0x1000 + used); used += bytes; return p; }
    void upload(void*, const void*, size_t) {}
    void replicate(void*, size_t, size_t) {}
};
struct ArenaMem {
    hsddp_handle* h; hsddp_handle::Arena* a;
    struct Up { void* dst; size_t idx, bytes; }; std::vector<Up> ups;
    struct Rep { void* base; size_t one, count; }; std::vector<Rep> reps;
    void* alloc(size_t bytes) { bytes = (std::max<size_t>(bytes, 8) + 255) / 256 * 256; if (a->used + bytes > a->cap) return nullptr; void* p = a->base + a->used; a->used += bytes; return p; }
    void upload(void* dst, const void* src, size_t bytes) { h->staged.emplace_back((const char*)src, (const char*)src + bytes); ups.push_back({dst, h->staged.size() - 1, bytes}); }
    void replicate(void* base, size_t one, size_t count) { reps.push_back({base, one, count}); }
    hipError_t flush() {
        hipError_t e = hipMemsetAsync(a->base, 0, a->used, h->stream); if (e != hipSuccess) return e;
        for (auto& u : ups) { e = hipMemcpyAsync(u.dst, h->staged[u.idx].data(), u.bytes, hipMemcpyHostToDevice, h->stream); if (e != hipSuccess) return e; }
        for (auto& r : reps) {
            size_t have = 1;
            while (have < r.count) { size_t n = std::min(have, r.count - have); e = hipMemcpyAsync((char*)r.base + have * r.one, r.base, n * r.one, hipMemcpyDeviceToDevice, h->stream); if (e != hipSuccess) return e; have += n; }
        }
        return hipSuccess;
    }
};

int hsddp_reconfigure(hsddp_handle_t* h, int n_phases, const hsddp_phase_desc_t* phases, const int* src_phase, const int* shift) {
    if (!h || n_phases <= 0 || !phases || !src_phase || !shift) return HSDDP_EINVAL;
    for (int i = 0; i < n_phases; i++) {
        if (phases[i].model != HSDDP_MODEL_WB && phases[i].model != HSDDP_MODEL_SRB && phases[i].model != HSDDP_MODEL_HKD) return HSDDP_EINVAL;
        if (i > 0 && !phase_chain_ok(phases[i - 1].model, phases[i].model)) return HSDDP_ENOTSUP;
        if ((!phases[i].shooting && i == 0) || phases[i].horizon <= 0 || shift[i] < 0 || src_phase[i] >= h->nph) return HSDDP_EINVAL;
        if (src_phase[i] >= 0 && h->ph[src_phase[i]].model != phases[i].model) return HSDDP_EINVAL;
        if (h->f32 && phases[i].model == HSDDP_MODEL_WB) return HSDDP_ENOTSUP;
    }
    if (phases[0].model != h->ph[0].model) return HSDDP_EINVAL;      // (the initial-condition buffer is sized for the first phase's model)
    HIPCK(hipSetDevice(h->device));
    const size_t B = h->batch;
    // 1. size of the new window, arena to build it in (grown only when the window outgrows it: the first ticks)
    std::vector<PhaseDev> np(n_phases);
    std::vector<int> sp, sk;
    {
        SizeMem sz;
        for (int i = 0; i < n_phases; i++) { int rc = setup_phase(sz, phases[i], i + 1 < n_phases ? &phases[i + 1] : nullptr, i == n_phases - 1, B, np[i], 0, h->f32); if (rc) return rc; }
        const int g = (h->cur_arena == 0) ? 1 : 0;
        auto& A = h->arena[g];
        if (sz.used > A.cap) {
            if (A.base) { HIPCK(hipStreamSynchronize(h->stream)); HIPCK(hipFree(A.base)); A.base = nullptr; A.cap = 0; }
            const size_t want = sz.used + sz.used / 8;
            if (hipMalloc((void**)&A.base, want) != hipSuccess) { fprintf(stderr, "[hsddp_hip] hipMalloc(%zu) failed (reconfigure arena)\n", want); return HSDDP_ENOMEM; }
            A.cap = want;
        }
        A.used = 0;
        h->staged.clear();
        ArenaMem mem{h, &A};
        for (int i = 0; i < n_phases; i++) {
            int rc = setup_phase(mem, phases[i], i + 1 < n_phases ? &phases[i + 1] : nullptr, i == n_phases - 1, B, np[i], (int)sp.size(), h->f32); if (rc) return rc;
            for (int k = 0; k <= phases[i].horizon; k++) { sp.push_back(i); sk.push_back(k); }
        }
        HIPCK(mem.flush());
        // 2. tables that depend on the slot / phase count
        if ((int)sp.size() > h->slots_cap || n_phases > h->nph_cap) {
            fprintf(stderr, "[hsddp_hip] reconfigure: %zu slots / %d phases exceed the capacity the handle was created with (%d / %d)\n", sp.size(), n_phases, h->slots_cap, h->nph_cap);
            return HSDDP_ENOTSUP;
        }
        // 3. warm start, device to device, old window -> new window (trajectories, ReB parameters, AL parameters)
        for (int i = 0; i < n_phases; i++) {
            const bool has = src_phase[i] >= 0;
            hipLaunchKernelGGL(k_warm_start, dim3(np[i].h + 1, h->batch), dim3(256), 0, h->stream, np[i], has ? h->ph[src_phase[i]] : np[i], has ? 1 : 0, shift[i]);
        }
        std::vector<PhaseDev> ss = np; for (auto& q : ss) q.shooting = 0;
        HIPCK(hipMemcpyAsync(h->d_ph, np.data(), sizeof(PhaseDev) * n_phases, hipMemcpyHostToDevice, h->stream));
        HIPCK(hipMemcpyAsync(h->d_ph_ss, ss.data(), sizeof(PhaseDev) * n_phases, hipMemcpyHostToDevice, h->stream));
        HIPCK(hipMemcpyAsync(h->d_slot_phase, sp.data(), sp.size() * 4, hipMemcpyHostToDevice, h->stream));
        HIPCK(hipMemcpyAsync(h->d_slot_k, sk.data(), sk.size() * 4, hipMemcpyHostToDevice, h->stream));
        std::vector<int> qs, os; int other_knots = 0; split_slots(np, sp, sk, qs, os, other_knots);
        if (!qs.empty()) HIPCK(hipMemcpyAsync(h->d_qslots, qs.data(), qs.size() * 4, hipMemcpyHostToDevice, h->stream));
        if (!os.empty()) HIPCK(hipMemcpyAsync(h->d_oslots, os.data(), os.size() * 4, hipMemcpyHostToDevice, h->stream));
        h->nq = (int)qs.size(); h->n_other = (int)os.size(); h->other_knots = other_knots;
        HIPCK(hipStreamSynchronize(h->stream));      // (the host vectors above are the copy sources; the old window is no longer read after this point)
        // 4. the new window becomes the handle's
        if (!h->gen_allocs.empty()) { for (void* p : h->gen_allocs) hipFree(p); h->gen_allocs.clear(); }      // storage of hsddp_create: first tick only
        h->cur_arena = g;
    }
    h->ph = np; h->nph = n_phases; h->nslots = (int)sp.size();
    h->probe_ok = true; for (int i = 0; i < n_phases; i++) if (!phases[i].shooting && phases[i].model != HSDDP_MODEL_WB) h->probe_ok = false;
    h->has_hkd = false; for (int i = 0; i < n_phases; i++) if (phases[i].model == HSDDP_MODEL_HKD) h->has_hkd = true;
    h->cache_valid = false;
    if (h->d_cmd_status) { hipFree(h->d_cmd_status); h->d_cmd_status = nullptr; }      // sized by the phase count
    HIPCK(hipGetLastError());
    return HSDDP_OK;
}

int hsddp_set_initial_condition(hsddp_handle_t* h, const double* x0) {
    if (!h || !x0) return HSDDP_EINVAL;
    HIPCK(hipSetDevice(h->device));
    HIPCK(hipMemcpy(h->d_x0, x0, (size_t)h->batch * h->ph[0].n * 8, hipMemcpyHostToDevice));
    return HSDDP_OK;
}

int hsddp_set_nominal(hsddp_handle_t* h, int phase, const double* Xbar, const double* Ubar, int per_problem) {
    if (!h || phase < 0 || phase >= h->nph) return HSDDP_EINVAL;
    HIPCK(hipSetDevice(h->device));
    PhaseDev& P = h->ph[phase]; const size_t sx = (size_t)(P.h + 1) * P.n, su = (size_t)P.h * P.m, B = h->batch;
    h->cache_valid = false;
    if (Xbar) {
        HIPCK(hipMemcpy(P.Xbar, Xbar, (per_problem ? B : 1) * sx * 8, hipMemcpyHostToDevice));
        if (!per_problem) HIPCK(dev_replicate(P.Xbar, sx * 8, B));
        HIPCK(hipMemcpy(P.X, P.Xbar, B * sx * 8, hipMemcpyDeviceToDevice));
    }
    if (Ubar) {
        HIPCK(hipMemcpy(P.Ubar, Ubar, (per_problem ? B : 1) * su * 8, hipMemcpyHostToDevice));
        if (!per_problem) HIPCK(dev_replicate(P.Ubar, su * 8, B));
        HIPCK(hipMemcpy(P.U, P.Ubar, B * su * 8, hipMemcpyDeviceToDevice));
    }
    HIPCK(hipMemset(P.K, 0, B * P.h * P.m * P.n * 8)); HIPCK(hipMemset(P.dU, 0, B * su * 8)); HIPCK(hipMemset(P.KdX, 0, B * su * 8)); HIPCK(hipMemset(P.dX, 0, B * sx * 8));
    return HSDDP_OK;
}

int hsddp_set_control_knot(hsddp_handle_t* h, int phase, int k, const double* u) {
    if (!h || phase < 0 || phase >= h->nph || k < 0 || k >= h->ph[phase].h) return HSDDP_EINVAL;
    HIPCK(hipSetDevice(h->device));
    PhaseDev& P = h->ph[phase]; const size_t pitch = (size_t)P.h * P.m * 8, w = (size_t)P.m * 8;
    HIPCK(hipStreamSynchronize(h->stream));
    for (double* dst : {(double*)P.Ubar, (double*)P.U}) {
        if (u) HIPCK(hipMemcpy2D(dst + (size_t)k * P.m, pitch, u, w, w, h->batch, hipMemcpyHostToDevice));
        else HIPCK(hipMemset2D(dst + (size_t)k * P.m, pitch, 0, w, h->batch));
    }
    h->cache_valid = false;
    return HSDDP_OK;
}

// ---- launch helpers
enum { UNIT_ROLLOUT = 0, UNIT_LQ = 1, UNIT_SWEEP = 2, UNIT_PROBE = 3 };
static HistDev hist_of(hsddp_handle* h) { return HistDev{h->d_hist, h->hist_cap}; }
static void launch_rollout_list(hsddp_handle* h, const EpsList& el, const SlotArrays& sa, const OptDev& o, int mask, const char* name, int unit = UNIT_ROLLOUT, const int* plist = nullptr, int nlist = 0) {
    Timed t(h, name);
    hipMemsetAsync(h->d_fail, 0, (size_t)h->batch * el.n * sizeof(int), h->stream);
    if (h->quad && o.MS && h->nq > 0) {      // whole-body running knots on lane quads, the rest (terminal knots, single-rigid-body tail) on the one-wave programs
        const int nbg = ((plist ? nlist : h->batch) + 15) / 16;
        hipLaunchKernelGGL(k_rollout_quad, dim3((unsigned)((size_t)el.n * h->nq * nbg)), dim3(64), 0, h->stream, h->d_ph, h->d_slot_phase, h->d_slot_k, h->d_qslots, h->nq, h->nslots, h->batch,
                           h->md, el, o, h->d_x0, sa, h->d_st, mask, h->d_fail, h->d_units + unit, plist, nlist);
        if (h->n_other > 0)
            hipLaunchKernelGGL(k_rollout, dim3((unsigned)((size_t)el.n * h->batch * h->n_other)), dim3(64), 0, h->stream, h->d_ph, h->nph, h->d_slot_phase, h->d_slot_k,
                               h->nslots, h->batch, h->md, el, o, h->d_x0, sa, h->d_st, mask, h->d_fail, h->d_units + unit, h->d_oslots, h->n_other, h->other_knots);
        return;
    }
    hipLaunchKernelGGL(h->has_hkd ? k_rollout_hkd : k_rollout, dim3((unsigned)((size_t)el.n * h->batch * h->nslots)), dim3(64), 0, h->stream, o.MS ? h->d_ph : h->d_ph_ss, h->nph, h->d_slot_phase, h->d_slot_k,
                       h->nslots, h->batch, h->md, el, o, h->d_x0, sa, h->d_st, mask, h->d_fail, h->d_units + unit, (const int*)nullptr, h->nslots, h->nslots - h->nph);
}
static void launch_rollout(hsddp_handle* h, double eps, const OptDev& o, int mask, bool eps_from_state = false, const int* plist = nullptr, int nlist = 0) {
    EpsList el{}; el.e[0] = eps; el.n = 1; el.writer = 0; el.from_state = eps_from_state ? 1 : 0;
    launch_rollout_list(h, el, h->sa, o, mask, "k_rollout", UNIT_ROLLOUT, plist, nlist);
    if (mask == MASK_NONE) h->cache_valid = true;     // masked launches only refresh problems whose cache was valid already
}
// slot arrays for the candidates of a probe launch, grown on demand (never inside an MPC tick once they exist)
static int ensure_probe_arrays(hsddp_handle* h, int cands) {
    if (cands <= h->sp_cands) return HSDDP_OK;
    double** p[4] = {&h->sp.cost, &h->sp.dsq, &h->sp.ming, &h->sp.maxh};
    HIPCK(hipStreamSynchronize(h->stream));
    for (auto q : p) { if (*q) HIPCK(hipFree(*q)); *q = nullptr; }
    h->sp_cands = 0;
    for (auto q : p) HIPCK(hipMalloc((void**)q, (size_t)cands * h->batch * h->slots_cap * sizeof(double)));
    h->sp_cands = cands; return HSDDP_OK;
}
static void launch_lq(hsddp_handle* h, const OptDev& o, int mask) {
    Timed t(h, "k_lq");
    if (h->has_hkd) hipLaunchKernelGGL(k_lq_hkd, dim3((unsigned)((size_t)h->batch * h->nslots)), dim3(64), 0, h->stream, h->d_ph, h->nph, h->d_slot_phase, h->d_slot_k, h->nslots, h->md, o, h->d_st, mask,
                                       0, h->d_units + UNIT_LQ);
    else hipLaunchKernelGGL(k_lq, dim3((unsigned)((size_t)h->batch * h->nslots)), dim3(LQ_NT), 0, h->stream, h->d_ph, h->nph, h->d_slot_phase, h->d_slot_k, h->nslots, h->md, o, h->d_st, mask,
                            h->cache_valid ? 1 : 0, h->d_units + UNIT_LQ);
}
static void launch_cost(hsddp_handle* h, const OptDev& o, int mask) {
    Timed t(h, "k_cost");
    hipLaunchKernelGGL(k_cost, dim3((unsigned)((size_t)h->batch * h->nslots)), dim3(64), 0, h->stream, h->d_ph, h->d_slot_phase, h->d_slot_k, h->nslots, o, h->sa, h->d_st, mask);
}
static void launch_sweep(hsddp_handle* h, const OptDev& o, int mask, double reg, int regularized, int do_linear, double lin_eps, int* succ) {
    Timed t(h, "k_sweep");
    if (h->f32) hipLaunchKernelGGL(k_sweep32, dim3(h->batch), dim3(SW_NT), 0, h->stream, h->d_ph, h->nph, o, h->d_st, mask, reg, regularized, do_linear, lin_eps, succ, h->d_units + UNIT_SWEEP, h->nslots - h->nph);
    else if (h->has_hkd) hipLaunchKernelGGL(k_sweep_hkd, dim3(h->batch), dim3(SW_NT), 0, h->stream, h->d_ph, h->nph, o, h->d_st, mask, reg, regularized, do_linear, lin_eps, succ, h->d_units + UNIT_SWEEP, h->nslots - h->nph);
    else hipLaunchKernelGGL(k_sweep, dim3(h->batch), dim3(SW_NT), 0, h->stream, h->d_ph, h->nph, o, h->d_st, mask, reg, regularized, do_linear, lin_eps, succ, h->d_units + UNIT_SWEEP, h->nslots - h->nph);
}
static void launch_eval(hsddp_handle* h, int mode, const OptDev& o, double eps, bool count, int iter_ou) {
    Timed t(h, "k_eval");
    if (count) hipMemsetAsync(h->d_counters, 0, 4 * sizeof(int), h->stream);
    hipLaunchKernelGGL(k_eval, dim3(h->batch), dim3(64), 0, h->stream, mode, h->nslots, h->sa, h->d_st, o, eps, h->d_fail, h->d_do_update, count ? h->d_counters : nullptr, iter_ou, hist_of(h), h->d_ls_list);
    if (count) { for (int q = 0; q < 4; q++) h->h_counters[q] = -1; hipMemcpyAsync(h->h_counters, h->d_counters, 4 * sizeof(int), hipMemcpyDeviceToHost, h->stream); }      // -1: "not delivered" (SYNC_COUNTERS)
}
static void launch_update_nominal(hsddp_handle* h, int mask) {
    Timed t(h, "k_update_nominal");
    hipLaunchKernelGGL(k_update_nominal, dim3(8, h->batch), dim3(256), 0, h->stream, h->d_ph, h->nph, h->d_st, mask);
}

int hsddp_solve(hsddp_handle_t* h, const hsddp_option_t* opt, float max_cputime_ms) {
    if (!h || !opt) return HSDDP_EINVAL;
    HIPCK(hipSetDevice(h->device));
    const OptDev o = to_dev(*opt);
    {   // history buffers: one entry after the initial rollout + one per completed inner iteration (MultiPhaseDDP.cpp:258-261, 382-385)
        const long long want = 1 + (long long)std::max(opt->max_AL_iter, 0) * std::max(opt->max_DDP_iter, 0);
        const int cap = (int)std::min<long long>(want, 4096);
        if (cap > h->hist_cap) {
            if (h->d_hist) { HIPCK(hipStreamSynchronize(h->stream)); HIPCK(hipFree(h->d_hist)); h->d_hist = nullptr; h->hist_cap = 0; }
            HIPCK(hipMalloc((void**)&h->d_hist, (size_t)h->batch * 4 * cap * sizeof(float))); h->hist_cap = cap;
        }
    }
    auto t0 = std::chrono::high_resolution_clock::now();
    auto elapsed = [&]() { return std::chrono::duration<float, std::milli>(std::chrono::high_resolution_clock::now() - t0).count(); };
    const bool budget = max_cputime_ms < 1e5f;
    hipError_t sync_err = hipSuccess;      // every host decision below reads counters a synchronise has to deliver: a failed one ends the solve
    auto timeup = [&]() { if (!budget) return false; if ((sync_err = hipStreamSynchronize(h->stream)) != hipSuccess) return true; float e = elapsed(); return e > max_cputime_ms || fabsf(e - max_cputime_ms) <= 1e-6f; };
    bool timed_out = false;
    int n_inner_est = h->batch;      // problems in the inner loop (upper bound until the first counted step): only steers where the full step of a line search is rolled out
    // initial rollout (MultiPhaseDDP.cpp:238-241)
    launch_rollout(h, 0.0, o, MASK_NONE);
    launch_update_nominal(h, MASK_NONE);
    launch_eval(h, EV_INIT, o, 0.0, false, 0);
    for (int iter_ou = 1; iter_ou <= opt->max_AL_iter && !timed_out; iter_ou++) {
        launch_eval(h, EV_OUTER_BEGIN, o, 0.0, false, iter_ou);
        for (int iter_in = 1; iter_in <= opt->max_DDP_iter && !timed_out; iter_in++) {
            if (iter_in == 1 && iter_ou > 1) launch_cost(h, o, MASK_INNER);   // parameters changed: refresh costs
            launch_eval(h, EV_INNER_BEGIN, o, 0.0, false, iter_ou);
            if (timeup()) { if (sync_err != hipSuccess) HIPCK(sync_err); timed_out = true; break; }
            launch_lq(h, o, MASK_INNER);
            if (timeup()) { if (sync_err != hipSuccess) HIPCK(sync_err); timed_out = true; break; }
            launch_sweep(h, o, MASK_INNER, 0.0, 1, o.MS ? 1 : 0, 1.0, nullptr);      // (linear rollout only with multiple shooting, MultiPhaseDDP.cpp:326-329)
            if (timeup()) { if (sync_err != hipSuccess) HIPCK(sync_err); timed_out = true; break; }
            // ---- line search (MultiPhaseDDP::line_search, MultiPhaseDDP.cpp:95-133): the step lengths 1, alpha, alpha^2, ... > 1e-3
            std::vector<double> steps; for (double e = 1.0; e > 1e-3; e *= opt->alpha) { steps.push_back(e); if (!(opt->alpha < 1.0) || steps.size() > 4096) break; }
            size_t next;
            // Where does the full step go?  Normally it is rolled out on its own and WRITES its trajectories (most searches end there).  When the
            // previous search of this handle saw most problems reject it (past convergence every search walks the whole ladder), its 11 KB per knot
            // of trajectories and contact-solve cache are written only to be overwritten: it then rides as candidate 0 of the probe launch and a
            // problem that does accept it gets the commit rollout every other accepted probe gets.  Same trials in the same order, same counts, same
            // trajectories either way (test_line_search_speculation_is_invisible); only the schedule differs.
            const bool probe_first = h->ls_speculate && h->ls_probe_first && o.MS && h->probe_ok;
            int n_search0 = std::max(n_inner_est, 1);
            if (probe_first) {
                launch_eval(h, EV_PRE_LS, o, 0.0, true, iter_ou);
                SYNC_COUNTERS();
                n_search0 = std::max(h->h_counters[2], 1); next = 0;
            } else {
                launch_eval(h, EV_PRE_LS, o, 0.0, false, iter_ou);
                // the full step first, on its own: most searches end here, and it is the one whose trajectories are most likely to stay
                launch_rollout(h, steps[0], o, MASK_LS);
                launch_eval(h, EV_LS_TRIAL, o, steps[0], true, iter_ou);
                SYNC_COUNTERS();
                next = 1;
                h->ls_probe_first = 8 * (long long)(n_search0 - h->h_counters[2]) < n_search0;      // fewer than one problem in eight took the full step: speculate next time
            }
            if (!(o.MS && h->probe_ok)) {      // single shooting through SRB / HKD phases hands its state over in memory: one trial per launch
                while (h->h_counters[2] != 0 && next < steps.size()) {
                    launch_rollout(h, steps[next], o, MASK_LS);
                    launch_eval(h, EV_LS_TRIAL, o, steps[next], true, iter_ou);
                    SYNC_COUNTERS(); next++;
                }
            }
            // every remaining candidate of the problems still searching in ONE launch (chunks of MAXCAND): candidates x problems x knots run
            // side by side as probes, the last candidate of the search also writes its trajectories (what a failed search leaves behind,
            // quirk vi); k_ls_pick then walks the candidates in order per problem, and only problems that accepted an earlier candidate
            // need one more rollout of their own step (the commit)
            while (h->h_counters[2] != 0 && next < steps.size()) {
                EpsList el{}; el.n = (int)std::min<size_t>(h->ls_chunk, steps.size() - next);
                for (int c = 0; c < el.n; c++) el.e[c] = steps[next + c];
                const bool last_chunk = next + el.n == steps.size();
                el.writer = last_chunk ? el.n - 1 : -1; el.from_state = 0;
                { int rc = ensure_probe_arrays(h, (int)std::min<size_t>(MAXCAND, steps.size())); if (rc) return rc; }      // (sized for the longest chunk of this ladder at once)
                launch_rollout_list(h, el, h->sp, o, MASK_LS, "k_ls_probe", UNIT_PROBE, h->d_ls_list, h->h_counters[2]);      // (the counted step before left the list of problems still searching)
                {
                    Timed t(h, "k_eval");
                    hipMemsetAsync(h->d_counters, 0, 4 * sizeof(int), h->stream);
                    hipLaunchKernelGGL(k_ls_pick, dim3(h->batch), dim3(64), 0, h->stream, h->nslots, h->batch, h->sp, h->sa, el, last_chunk ? 1 : 0, h->d_st, o, h->d_fail, h->d_counters, h->d_ls_list, h->d_commit_list);
                    for (int q = 0; q < 4; q++) h->h_counters[q] = -1;
                    hipMemcpyAsync(h->h_counters, h->d_counters, 4 * sizeof(int), hipMemcpyDeviceToHost, h->stream);
                }
                SYNC_COUNTERS();
                if (h->h_counters[3] != 0) launch_rollout(h, 0.0, o, MASK_COMMIT, true, h->d_commit_list, h->h_counters[3]);      // problems that accepted a probe: their own step, for real
                if (probe_first && next == 0) h->ls_probe_first = 8 * (long long)h->h_counters[3] < n_search0;      // one in eight or more needed a commit: back to writing the full step
                next += el.n;
            }
            launch_update_nominal(h, MASK_LS_OK);
            launch_eval(h, EV_POST_LS, o, 0.0, true, iter_ou);
            if (timeup()) { if (sync_err != hipSuccess) HIPCK(sync_err); timed_out = true; break; }
            SYNC_COUNTERS();
            n_inner_est = h->h_counters[0];
            if (h->h_counters[0] == 0) break;
        }
        if (timed_out) break;
        launch_eval(h, EV_OUTER_END, o, 0.0, true, iter_ou);
        {
            Timed t(h, "k_update_params");
            hipLaunchKernelGGL(k_update_params, dim3(8, h->batch), dim3(256), 0, h->stream, h->d_ph, h->nph, o, h->d_st, h->d_do_update);
        }
        SYNC_COUNTERS();
        n_inner_est = h->h_counters[1];
        if (h->h_counters[1] == 0) break;
    }
    if (timed_out) launch_eval(h, EV_TIMEOUT, o, 0.0, false, 0);
    HIPCK(hipStreamSynchronize(h->stream));
    h->solve_ms = elapsed();
    drain_events(h);
    HIPCK(hipGetLastError());
    return HSDDP_OK;
}

// ---- step API (MultiPhaseDDP public methods)
int hsddp_hybrid_rollout(hsddp_handle_t* h, double eps, const hsddp_option_t* opt) {
    if (!h || !opt) return HSDDP_EINVAL;
    HIPCK(hipSetDevice(h->device)); OptDev o = to_dev(*opt);
    launch_rollout(h, eps, o, MASK_NONE); launch_eval(h, EV_REDUCE_ONLY, o, 0.0, false, 0);
    HIPCK(hipStreamSynchronize(h->stream)); drain_events(h); HIPCK(hipGetLastError()); return HSDDP_OK;
}
int hsddp_compute_cost(hsddp_handle_t* h, const hsddp_option_t* opt) {
    if (!h || !opt) return HSDDP_EINVAL; HIPCK(hipSetDevice(h->device)); OptDev o = to_dev(*opt);
    launch_cost(h, o, MASK_NONE); launch_eval(h, EV_REDUCE_ONLY, o, 0.0, false, 0);
    HIPCK(hipStreamSynchronize(h->stream)); drain_events(h); HIPCK(hipGetLastError()); return HSDDP_OK;
}
int hsddp_LQ_approximation(hsddp_handle_t* h, const hsddp_option_t* opt) {
    if (!h || !opt) return HSDDP_EINVAL; HIPCK(hipSetDevice(h->device)); OptDev o = to_dev(*opt);
    launch_lq(h, o, MASK_NONE); HIPCK(hipStreamSynchronize(h->stream)); drain_events(h); HIPCK(hipGetLastError()); return HSDDP_OK;
}
int hsddp_backward_sweep(hsddp_handle_t* h, double regularization, int* success) {
    if (!h) return HSDDP_EINVAL; HIPCK(hipSetDevice(h->device)); OptDev o{};
    launch_sweep(h, o, MASK_NONE, regularization, 0, 0, 0.0, h->d_success);
    if (!h->has_hkd && !h->f32) hipLaunchKernelGGL(k_refresh_kdx, dim3(64, h->batch), dim3(64), 0, h->stream, h->d_ph, h->nph);      // the new gains on the standing search direction
    HIPCK(hipStreamSynchronize(h->stream)); drain_events(h); HIPCK(hipGetLastError());
    if (success) HIPCK(hipMemcpy(success, h->d_success, h->batch * sizeof(int), hipMemcpyDeviceToHost));
    return HSDDP_OK;
}
int hsddp_linear_rollout(hsddp_handle_t* h, double eps, const hsddp_option_t* opt) {
    (void)opt; if (!h) return HSDDP_EINVAL; HIPCK(hipSetDevice(h->device));
    { Timed t(h, "k_linear");
      if (h->f32) hipLaunchKernelGGL(k_linear32, dim3(h->batch), dim3(SW_NT), 0, h->stream, h->d_ph, h->nph, h->d_st, eps);
      else if (h->has_hkd) hipLaunchKernelGGL(k_linear_hkd, dim3(h->batch), dim3(SW_NT), 0, h->stream, h->d_ph, h->nph, h->d_st, eps);
      else hipLaunchKernelGGL(k_linear, dim3(h->batch), dim3(SW_NT), 0, h->stream, h->d_ph, h->nph, h->d_st, eps); }
    HIPCK(hipStreamSynchronize(h->stream)); drain_events(h); HIPCK(hipGetLastError()); return HSDDP_OK;
}
int hsddp_update_nominal_trajectory(hsddp_handle_t* h) {
    if (!h) return HSDDP_EINVAL; HIPCK(hipSetDevice(h->device));
    launch_update_nominal(h, MASK_NONE); HIPCK(hipStreamSynchronize(h->stream)); drain_events(h); return HSDDP_OK;
}
static int read_states(hsddp_handle* h, std::vector<ProbState>& st) {
    st.resize(h->batch); HIPCK(hipSetDevice(h->device));
    HIPCK(hipMemcpy(st.data(), h->d_st, sizeof(ProbState) * h->batch, hipMemcpyDeviceToHost)); return HSDDP_OK;
}
int hsddp_get_exp_cost_change(hsddp_handle_t* h, double* dV_1, double* dV_2) {
    if (!h || !dV_1 || !dV_2) return HSDDP_EINVAL;
    std::vector<ProbState> st; int rc = read_states(h, st); if (rc) return rc;
    for (int b = 0; b < h->batch; b++) { dV_1[b] = st[b].dV_1; dV_2[b] = st[b].dV_2; } return HSDDP_OK;
}
int hsddp_measure_dynamics_feasibility(hsddp_handle_t* h, double* feas) {
    if (!h || !feas) return HSDDP_EINVAL;
    std::vector<ProbState> st; int rc = read_states(h, st); if (rc) return rc;
    for (int b = 0; b < h->batch; b++) feas[b] = st[b].feas; return HSDDP_OK;
}
int hsddp_get_info(hsddp_handle_t* h, hsddp_info_t* info) {
    if (!h || !info) return HSDDP_EINVAL;
    std::vector<ProbState> st; int rc = read_states(h, st); if (rc) return rc;
    for (int b = 0; b < h->batch; b++) {
        info[b].actual_cost = st[b].actual_cost; info[b].dyn_feas = st[b].feas; info[b].max_tconstr = st[b].info_tconstr; info[b].max_pconstr = st[b].info_pconstr;
        info[b].n_iters = st[b].iter; info[b].n_ls_iters = st[b].ls_total; info[b].n_reg_iters = st[b].reg_total; info[b].status = st[b].status;
    }
    return HSDDP_OK;
}

int hsddp_get_history(hsddp_handle_t* h, int problem, int cap, float* cost, float* dyn_feas, float* eqn_feas, float* ineq_feas, int* n) {
    if (!h || problem < 0 || problem >= h->batch || cap < 0 || !n) return HSDDP_EINVAL;
    HIPCK(hipSetDevice(h->device));
    ProbState st; HIPCK(hipMemcpy(&st, h->d_st + problem, sizeof(ProbState), hipMemcpyDeviceToHost));
    const int have = std::min(st.hist_n, h->hist_cap); *n = have;
    const int m = std::min(have, cap);
    float* dst[4] = {cost, dyn_feas, eqn_feas, ineq_feas};
    for (int q = 0; q < 4 && m > 0; q++) if (dst[q]) HIPCK(hipMemcpy(dst[q], h->d_hist + ((size_t)problem * 4 + q) * h->hist_cap, m * sizeof(float), hipMemcpyDeviceToHost));
    return HSDDP_OK;
}

int hsddp_field_shape(hsddp_handle_t* h, int phase, int field, int* count, int* elems) {
    if (!h || phase < 0 || phase >= h->nph || field < 0 || field >= HSDDP_F_COUNT) return HSDDP_EINVAL;
    int stride; field_dev(h->ph[phase], field, *count, *elems, stride); return HSDDP_OK;
}
int hsddp_get_field(hsddp_handle_t* h, int phase, int field, int b0, int nb, double* dst) {
    if (!h || phase < 0 || phase >= h->nph || field < 0 || field >= HSDDP_F_COUNT || b0 < 0 || nb < 0 || b0 + nb > h->batch || !dst) return HSDDP_EINVAL;
    HIPCK(hipSetDevice(h->device));
    int count, elems, stride; const double* src = field_dev(h->ph[phase], field, count, elems, stride);
    const size_t sz = (size_t)count * elems;
    if (h->f32 && stride != elems && sz > 0 && nb > 0) {      // a field inside the LQ record of an fp32 handle: floats on the device
        const PhaseDev& P = h->ph[phase];
        const int off = field == HSDDP_F_A ? P.oA : field == HSDDP_F_B ? P.oB : field == HSDDP_F_C ? P.oC : field == HSDDP_F_D ? P.oD : field == HSDDP_F_LX ? P.oLx : field == HSDDP_F_LU ? P.oLu :
                        field == HSDDP_F_LY ? P.oLy : field == HSDDP_F_LXX ? P.oLxx : field == HSDDP_F_LUU ? P.oLuu : P.oLyy;
        std::vector<float> tmp(sz * nb);
        HIPCK(hipMemcpy2D(tmp.data(), (size_t)elems * 4, P.rec32 + (size_t)b0 * count * stride + off, (size_t)stride * 4, (size_t)elems * 4, (size_t)nb * count, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < tmp.size(); i++) dst[i] = tmp[i];
        return HSDDP_OK;
    }
    if (!src) { memset(dst, 0, sz * nb * 8); return HSDDP_OK; }
    if (sz == 0 || nb == 0) return HSDDP_OK;
    if (wb_structured(h->ph[phase], field)) {      // the record holds the lower 18 rows only: the identities of the upper rows are filled in here
        const int st = field == HSDDP_F_A ? 648 : 216;
        std::vector<double> tmp((size_t)st * nb * count);
        HIPCK(hipMemcpy2D(tmp.data(), (size_t)st * 8, src + (size_t)b0 * count * stride, (size_t)stride * 8, (size_t)st * 8, (size_t)nb * count, hipMemcpyDeviceToHost));
        for (size_t r = 0; r < (size_t)nb * count; r++) wb_expand_ab(field, h->ph[phase].dt, tmp.data() + r * st, dst + r * elems);
        return HSDDP_OK;
    }
    if (stride == elems) HIPCK(hipMemcpy(dst, src + (size_t)b0 * sz, sz * nb * 8, hipMemcpyDeviceToHost));
    else HIPCK(hipMemcpy2D(dst, (size_t)elems * 8, src + (size_t)b0 * count * stride, (size_t)stride * 8, (size_t)elems * 8, (size_t)nb * count, hipMemcpyDeviceToHost));
    return HSDDP_OK;
}
float hsddp_get_solve_time_ms(hsddp_handle_t* h) { return h ? h->solve_ms : 0.f; }

int hsddp_warm_start_phase(hsddp_handle_t* dst, int dphase, hsddp_handle_t* src, int sphase, int shift) {
    if (!dst || dphase < 0 || dphase >= dst->nph || shift < 0) return HSDDP_EINVAL;
    const PhaseDev& D = dst->ph[dphase];
    dst->cache_valid = false;
    const bool has = src != nullptr && sphase >= 0;
    if (has && (sphase >= src->nph || src->batch != dst->batch || src->device != dst->device || src->ph[sphase].model != D.model)) return HSDDP_EINVAL;
    HIPCK(hipSetDevice(dst->device));
    if (has) HIPCK(hipStreamSynchronize(src->stream));
    hipLaunchKernelGGL(k_warm_start, dim3(D.h + 1, dst->batch), dim3(256), 0, dst->stream, D, has ? src->ph[sphase] : D, has ? 1 : 0, shift);
    HIPCK(hipStreamSynchronize(dst->stream));
    return HSDDP_OK;
}

int hsddp_export_mpc_command(hsddp_handle_t* h, int problem, int n_steps, double mpc_time, double dt, const float* status_times, unsigned int* out) {
    if (!h || problem < 0 || problem >= h->batch || n_steps <= 0 || !out) return HSDDP_EINVAL;
    std::vector<int> sp, sk;   // control knot k -> (phase, k_rel)   (MHPCProblemData::get_index)
    for (int i = 0; i < h->nph && (int)sp.size() < n_steps; i++) {
        if (h->ph[i].model != HSDDP_MODEL_WB) break;
        for (int k = 0; k < h->ph[i].h && (int)sp.size() < n_steps; k++) { sp.push_back(i); sk.push_back(k); }
    }
    if ((int)sp.size() < n_steps) return HSDDP_EINVAL;
    HIPCK(hipSetDevice(h->device));
    const size_t words = 1 + (size_t)n_steps * HSDDP_CMD_WORDS_PER_STEP;
    // staging buffers live in the handle (an MPC tick must not pay hipMalloc / hipFree): grown on demand, freed by hsddp_destroy
    if (words > h->cmd_words) { if (h->d_cmd) HIPCK(hipFree(h->d_cmd)); h->d_cmd = nullptr; h->cmd_words = 0; HIPCK(hipMalloc((void**)&h->d_cmd, words * 4)); h->cmd_words = words; }
    if (n_steps > h->cmd_steps) { if (h->d_cmd_map) HIPCK(hipFree(h->d_cmd_map)); h->d_cmd_map = nullptr; h->cmd_steps = 0; HIPCK(hipMalloc((void**)&h->d_cmd_map, 2 * (size_t)n_steps * sizeof(int))); h->cmd_steps = n_steps; }
    if (status_times && !h->d_cmd_status) HIPCK(hipMalloc((void**)&h->d_cmd_status, (size_t)h->nph * 4 * sizeof(float)));
    HIPCK(hipMemcpyAsync(h->d_cmd_map, sp.data(), n_steps * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCK(hipMemcpyAsync(h->d_cmd_map + n_steps, sk.data(), n_steps * sizeof(int), hipMemcpyHostToDevice, h->stream));
    if (status_times) HIPCK(hipMemcpyAsync(h->d_cmd_status, status_times, (size_t)h->nph * 4 * sizeof(float), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_pack_command, dim3(n_steps), dim3(256), 0, h->stream, h->d_ph, h->d_cmd_map, h->d_cmd_map + n_steps, n_steps, problem, mpc_time, dt,
                       status_times ? h->d_cmd_status : nullptr, h->d_cmd);
    HIPCK(hipMemcpyAsync(out, h->d_cmd, words * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCK(hipStreamSynchronize(h->stream));
    return HSDDP_OK;
}

int hsddp_export_solver_info(hsddp_handle_t* h, int problem, unsigned int* out) {
    if (!h || problem < 0 || problem >= h->batch || !out) return HSDDP_EINVAL;
    HIPCK(hipSetDevice(h->device));
    ProbState st; HIPCK(hipMemcpy(&st, h->d_st + problem, sizeof(ProbState), hipMemcpyDeviceToHost));
    const int iv[3] = {st.iter, st.ls_total, st.reg_total};
    const float fv[5] = {h->solve_ms, (float)st.actual_cost, (float)st.feas, (float)st.info_pconstr, (float)st.info_tconstr};
    memcpy(out, iv, sizeof(iv)); memcpy(out + 3, fv, sizeof(fv));
    return HSDDP_OK;
}

// knots processed by the launches of a kernel family since the last hsddp_reset_kernel_times (masked launches skip problems: the
// roofline figure of bench.py divides algorithmic bytes by what a launch really processed)
int hsddp_get_kernel_units(hsddp_handle_t* h, const char* name, long long* units) {
    if (!h || !name || !units) return HSDDP_EINVAL;
    const int id = !strcmp(name, "k_rollout") ? UNIT_ROLLOUT : !strcmp(name, "k_lq") ? UNIT_LQ : !strcmp(name, "k_sweep") ? UNIT_SWEEP : !strcmp(name, "k_ls_probe") ? UNIT_PROBE : -1;
    if (id < 0) return HSDDP_EINVAL;
    HIPCK(hipSetDevice(h->device));
    unsigned long long v = 0; HIPCK(hipMemcpy(&v, h->d_units + id, sizeof(v), hipMemcpyDeviceToHost));
    *units = (long long)v; return HSDDP_OK;
}

int hsddp_get_kernel_times(hsddp_handle_t* h, int max_n, double* ms, long long* launches, char* names, int names_cap) {
    if (!h) return 0;
    int n = std::min<int>(max_n, (int)h->kname.size()); int pos = 0;
    for (int i = 0; i < n; i++) {
        ms[i] = h->kms[i]; launches[i] = h->kcnt[i];
        int len = (int)h->kname[i].size();
        if (pos + len + 1 < names_cap) { memcpy(names + pos, h->kname[i].c_str(), len + 1); pos += len + 1; }
    }
    if (pos < names_cap) names[pos] = 0;
    return n;
}
#if defined(LQ_PROF) || defined(ROLL_PROF)
int hsddp_debug_lq_prof(unsigned long long* out16, int reset) {
    hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_lq_prof), 16 * sizeof(unsigned long long));
    if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_lq_prof), z, sizeof(z)); }
    return 0;
}
#endif
#ifdef QUAD_PROF
int hsddp_debug_quad_prof(unsigned long long* out24, int reset) {
    hipMemcpyFromSymbol(out24, HIP_SYMBOL(g_quad_prof), 24 * sizeof(unsigned long long));
    if (reset) { unsigned long long z[24] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_quad_prof), z, sizeof(z)); }
    return 0;
}
#endif
#ifdef SW_PROF
int hsddp_debug_sweep_prof(unsigned long long* out48, int reset) {
    hipMemcpyFromSymbol(out48, HIP_SYMBOL(g_sw_prof), 48 * sizeof(unsigned long long));
    if (reset) { unsigned long long z[48] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_sw_prof), z, sizeof(z)); }
    return 0;
}
#endif
long long hsddp_debug_malloc_count(void) { return g_dev_allocs; }
int hsddp_reset_kernel_times(hsddp_handle_t* h) {
    if (!h) return HSDDP_EINVAL;
    for (auto& v : h->kms) v = 0; for (auto& v : h->kcnt) v = 0;
    HIPCK(hipSetDevice(h->device)); HIPCK(hipMemset(h->d_units, 0, 8 * sizeof(unsigned long long)));
    return HSDDP_OK;
}

}  // extern "C"
#endif
