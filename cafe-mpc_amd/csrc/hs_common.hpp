// Common definitions for the HIP HS-DDP kernels (gfx950).
//
// SPMD-phase style: a kernel body is a sequence of PHASES; inside a phase every lane/thread runs the
// same lambda on its own id and communicates with other lanes only through LDS, phases are separated
// by a workgroup barrier.  On the GPU a phase is `body(threadIdx.x); __syncthreads();`.
// With -DHS_HOST_EMU (tests/_emu only: a CPU lane-emulator used to debug kernel logic in a container
// that has no GPU — never built into or loaded by the product) a phase is a plain loop over ids.
#pragma once
#include <cstddef>
#include <cmath>
#include <cstdint>

#ifdef HS_HOST_EMU
#define HD inline
#define HDH inline
#define HS_SHARED static thread_local
#define HS_PHASE(NT, ...) { for (int tid = 0; tid < (NT); ++tid) { __VA_ARGS__ } }
// wave-level phase: only the first 64 threads of the workgroup run it, ordered by a wave barrier (no s_barrier)
#define HS_WPHASE(...) { for (int tid = 0; tid < 64; ++tid) { __VA_ARGS__ } }
#define HS_PHASE_L(NT, ...) HS_PHASE(NT, __VA_ARGS__)
// values a lane keeps in registers from one phase to a later one: the emulator keeps one copy per lane
#define HS_NLANES(NT) (NT)
#define HS_LANE(tid) (tid)
template <int NT> inline void hs_phase_sync_all() {}
// wave-level phase of wave W of a multi-wave workgroup, tid = lane 0..63 (the emulator runs the waves' phases in program order)
#define HS_WPHASE_W(W, ...) { for (int tid = 0; tid < 64; ++tid) { __VA_ARGS__ } }
#else
#include <hip/hip_runtime.h>
#define HD __device__ __forceinline__
#define HDH __host__ __device__ inline
#define HS_SHARED __shared__
// The per-knot workgroups (NT == 64 or 128) only ever hand data over through LDS inside a kernel: its phase boundary is the LDS-only
// barrier (see HS_PHASE_L below) - __syncthreads() would also drain every outstanding global store / prefetch at each of the
// ~60 phase boundaries of a knot.  Larger workgroups keep the full barrier.
template <int NT> __device__ __forceinline__ void hs_phase_sync() { __syncthreads(); }
template <> __device__ __forceinline__ void hs_phase_sync<64>() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
template <> __device__ __forceinline__ void hs_phase_sync<128>() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }   // the per-knot kernels, one or two waves
template <int NT> __device__ __forceinline__ void hs_phase_sync_all() { hs_phase_sync<NT>(); }     // a bare workgroup barrier between wave-level phase sequences
#define HS_PHASE(NT, ...) { { int tid = threadIdx.x; asm volatile("" : "+v"(tid)); __builtin_assume(tid >= 0); if (tid < (NT)) { __VA_ARGS__ } } hs_phase_sync<(NT)>(); }
// values a lane keeps in registers from one phase to a later one (the emulator keeps one copy per lane)
#define HS_NLANES(NT) 1
#define HS_LANE(tid) 0
// wave-level phase: executed by wave 0 only; a wave runs in lock-step and its LDS operations complete in program
// order, so the only thing to prevent is compiler motion across the phase boundary.
// LDS-only phase boundary: raw s_barrier behind an lgkmcnt(0) wait.  __syncthreads() carries a workgroup release fence,
// for which hipcc emits s_waitcnt vmcnt(0) whenever global stores may be pending - that would also drain the
// prefetch loads that are meant to stay in flight across the knot.  Phases that only hand data over through LDS use this.
// (tid is laundered through an empty asm so that index arithmetic derived from it is NOT hoisted out of the knot loop and
// kept live in hundreds of registers / scratch slots)
#define HS_PHASE_L(NT, ...) { { int tid = threadIdx.x; asm volatile("" : "+v"(tid)); __builtin_assume(tid >= 0); if (tid < (NT)) { __VA_ARGS__ } } asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
// wave-level phase of wave W of a multi-wave workgroup, tid = lane 0..63: the waves of a workgroup can run DIFFERENT phase sequences
// side by side between two workgroup barriers as long as they touch disjoint LDS
#define HS_WPHASE_W(W, ...) { if ((threadIdx.x >> 6) == (W)) { int tid = threadIdx.x & 63; asm volatile("" : "+v"(tid)); __builtin_assume(tid >= 0 && tid < 64); { __VA_ARGS__ } } __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
#define HS_WPHASE(...) { if (threadIdx.x < 64) { int tid = threadIdx.x; asm volatile("" : "+v"(tid)); { __VA_ARGS__ } } __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
#endif

// Trajectory pointers live in descriptors that the kernels read from memory, so the compiler cannot see that they point to global
// memory and would emit FLAT loads / stores (which also count on lgkmcnt: every LDS wait then drains them).  HS_GLOBAL pins the
// address space in device code; on the host (and in the lane emulator) it is an ordinary pointer of the same size.
// The descriptors themselves are written by the host before a launch and never by a kernel: reading them through the CONSTANT
// address space (HS_CONST) lets the compiler fetch their fields with scalar loads and keep them across memory clobbers instead
// of re-reading each field with an exposed vector load after every phase boundary.
#if !defined(HS_HOST_EMU) && defined(__HIP_DEVICE_COMPILE__)
#define HS_GLOBAL __attribute__((address_space(1)))
#define HS_CONST __attribute__((address_space(4)))
#else
#define HS_GLOBAL
#define HS_CONST
#endif
// generic pointer that the optimiser knows to point into constant memory (address-space inference follows the double cast)
#define HS_AS_CONST(T, p) ((const T*)(const HS_CONST T*)(p))

// compiler-only memory barrier: stops the scheduler from hoisting a whole unrolled recurrence's LDS loads ahead of it
// (hundreds of live registers); emits no instruction
#define HS_CBAR() asm volatile("" ::: "memory")
// same, and the value x must have been computed by this point (pins a dependent chain between two batches of loads)
#ifdef HS_HOST_EMU
#define HS_PIN(x)
#define HS_PIN_S(x)
#else
#define HS_PIN(x) asm volatile("" : "+v"(x) :: "memory")
// a wave-uniform value that must sit in a scalar register here (keeps a select chain over descriptor fields from turning into one branch per field)
#define HS_PIN_S(x) asm volatile("" : "+s"(x))
#endif

namespace hs {

constexpr int WAVE = 64;

// Sparsity pattern of a triangular factor.  NZ::nz(i, k), k < i: whether L(i,k) can be non-zero (a compile-time pattern; updates with a structurally zero multiplier are not issued).
struct DenseNZ { static constexpr bool nz(int, int) { return true; } };
constexpr double GRAV = 9.81;

// forward-mode scalar: value + one tangent
struct Dual {
    double v, d;
    HD Dual() : v(0), d(0) {}
    HD Dual(double a) : v(a), d(0) {}
    HD Dual(double a, double b) : v(a), d(b) {}
};
HD Dual operator+(Dual a, Dual b) { return Dual(a.v + b.v, a.d + b.d); }
HD Dual operator-(Dual a, Dual b) { return Dual(a.v - b.v, a.d - b.d); }
HD Dual operator-(Dual a) { return Dual(-a.v, -a.d); }
HD Dual operator*(Dual a, Dual b) { return Dual(a.v * b.v, a.v * b.d + a.d * b.v); }
HD Dual operator*(double a, Dual b) { return Dual(a * b.v, a * b.d); }
HD Dual operator*(Dual a, double b) { return Dual(a.v * b, a.d * b); }
HD Dual operator+(Dual a, double b) { return Dual(a.v + b, a.d); }
HD Dual operator-(Dual a, double b) { return Dual(a.v - b, a.d); }
#ifdef HS_HOST_EMU
HD void sincos_(double a, double& s, double& c) { s = sin(a); c = cos(a); }
#else
HD void sincos_(double a, double& s, double& c) { sincos(a, &s, &c); }      // one argument reduction for both
#endif
HD void sincos_(Dual a, Dual& s, Dual& c) { double sv, cv; sincos_(a.v, sv, cv); s = Dual(sv, cv * a.d); c = Dual(cv, -sv * a.d); }
HD double val(double a) { return a; }
HD double val(Dual a) { return a.v; }
HD double tang(double) { return 0.0; }
HD double tang(Dual a) { return a.d; }
template <class S> HD S mk(double v, bool seed);
template <> HD double mk<double>(double v, bool) { return v; }
template <> HD Dual mk<Dual>(double v, bool seed) { return Dual(v, seed ? 1.0 : 0.0); }

HD double hs_fma(double a, double b, double c) { return fma(a, b, c); }
// 1/sqrt(x): one reciprocal-square-root instead of a square root plus two divisions in every Cholesky column
#ifdef HS_HOST_EMU
HD double hs_rsqrt(double x) { return 1.0 / std::sqrt(x); }
#else
HD double hs_rsqrt(double x) { return rsqrt(x); }
#endif

#ifndef HS_HOST_EMU
HD double hs_rcp(double x) { return __builtin_amdgcn_rcp(x); }
HD float hs_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
HD float hs_readlane(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }
// broadcast of lane `src` (compile-time constant after unrolling -> v_readlane_b32 x2 into an SGPR pair)
HD double hs_readlane(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
// Same factorisation as chol_s (identical sequence of FMAs per entry, bit-identical result) but right-looking with row i
// of the matrix in the REGISTERS of lane i: the pivot and the column being eliminated travel by lane broadcast, so a
// column costs a dependent chain of a few instructions instead of an LDS round trip plus a workgroup barrier.
// Must be called with all 64 lanes active.  Lo may alias A.
template <int N, int LD, class NZ = DenseNZ>
HD void chol_r(const double* A, int sr, int sk, double* Lo, double* rd, double diag_add, int tid, int* ok = nullptr) {   // A(i,k) = A[i*sr + k*sk]
    const int row = tid < N ? tid : N - 1;          // idle lanes mirror the last row (never written back)
    double a[N];
    _Pragma("unroll")
    for (int k = 0; k < N; k++) a[k] = A[row * sr + k * sk];
    double rown = 0.0;
    _Pragma("unroll")
    for (int j = 0; j < N; j++) {
        double piv = hs_readlane(a[j], j) + diag_add;
        if (ok != nullptr) { const bool good = piv > 0.0; if (!good && tid == j) *ok = 0; piv = good ? piv : 1.0; }   // not positive definite: flag, keep going on a dummy pivot
        const double r = hs_rsqrt(piv);
        const double lij = a[j] * r;
        a[j] = lij;
        rown = (tid == j) ? r : rown;
        _Pragma("unroll")
        for (int k = j + 1; k < N; k++) if (NZ::nz(k, j)) a[k] -= lij * hs_readlane(lij, k);
    }
    if (tid < N) {
        rd[tid] = rown;
        _Pragma("unroll")
        for (int k = 0; k < N - 1; k++) if (k < tid) Lo[tid * LD + k] = a[k];
    }
}
// value of lane `src` (any lane, per-lane choice) — ds_bpermute moves 32 bits per instruction
HD double hs_bperm(double v, int src) {
    const int lo = __builtin_amdgcn_ds_bpermute(4 * src, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(4 * src, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
HD float hs_bperm(float v, int src) { return __int_as_float(__builtin_amdgcn_ds_bpermute(4 * src, __float_as_int(v))); }
// Cholesky of the 18 x 18 whole-body mass matrix in the legs-first order (rows 3l..3l+2: leg l, rows 12..17: floating base; no entries
// between different legs), row i in lane i, in place.  The four 3 x 3 leg blocks do not depend on each other: their three column steps run
// for all legs AT ONCE (pivot and multipliers travel inside a leg by ds_bpermute), so the dependent chain is 3 + 12 short + 6 column steps
// instead of 18 full ones; the base rows then take the twelve leg columns in order (the same sequence of multiply-adds per entry as the
// column-by-column chol_r with the same pattern: bit-identical factor) and the dense 6 x 6 base block follows.  All 64 lanes active.
HD void chol_wb18(double* M, double* rd, int tid) {
    const int row = tid < 18 ? tid : 17, leg = row < 12 ? row / 3 : 3, jl = row - 3 * leg;
    double al[3], a[18];
    _Pragma("unroll") for (int k = 0; k < 3; k++) al[k] = M[row * 18 + 3 * leg + k];
    _Pragma("unroll") for (int k = 0; k < 18; k++) a[k] = M[row * 18 + k];
    double rown = 0.0, rj[3], lj[3];
    _Pragma("unroll")
    for (int j = 0; j < 3; j++) {        // leg blocks, column j of every leg
        const double r = hs_rsqrt(al[j]);                       // meaningful on the pivot lanes (jl == j)
        rown = (tid < 12 && jl == j) ? r : rown;
        rj[j] = r;
        const double lij = al[j] * hs_bperm(r, 3 * leg + j);    // L(row, 3 leg + j), rows below the pivot
        lj[j] = lij; al[j] = lij;
        _Pragma("unroll")
        for (int k = j + 1; k < 3; k++) al[k] -= lij * hs_bperm(lij, 3 * leg + k);
    }
    _Pragma("unroll")
    for (int c = 0; c < 12; c++) {       // base rows (lanes 12..17) against the leg columns, in column order
        const int l = c / 3, j = c % 3;
        const double lic = a[c] * hs_readlane(rj[j], c);
        a[c] = lic;
        _Pragma("unroll")
        for (int k = j + 1; k < 3; k++) a[3 * l + k] -= lic * hs_readlane(lj[j], 3 * l + k);
        _Pragma("unroll")
        for (int k = 12; k < 18; k++) a[k] -= lic * hs_readlane(lic, k);
    }
    _Pragma("unroll")
    for (int j = 12; j < 18; j++) {      // base block
        const double r = hs_rsqrt(hs_readlane(a[j], j));
        const double lij = a[j] * r;
        a[j] = lij;
        rown = (tid == j) ? r : rown;
        _Pragma("unroll")
        for (int k = j + 1; k < 18; k++) a[k] -= lij * hs_readlane(lij, k);
    }
    if (tid < 12) {
        rd[tid] = rown;
        _Pragma("unroll") for (int k = 0; k < 2; k++) if (k < jl) M[row * 18 + 3 * leg + k] = al[k];
    } else if (tid < 18) {
        rd[tid] = rown;
        _Pragma("unroll") for (int k = 0; k < 17; k++) if (k < tid) M[tid * 18 + k] = a[k];
    }
}
#endif

template <class S> struct V3 { S x, y, z; };
template <class S> HD V3<S> operator+(V3<S> a, V3<S> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class S> HD V3<S> operator-(V3<S> a, V3<S> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class S> HD V3<S> scale(S s, V3<S> a) { return {s * a.x, s * a.y, s * a.z}; }
template <class S> HD V3<S> scaled(double s, V3<S> a) { return {s * a.x, s * a.y, s * a.z}; }
template <class S> HD V3<S> cross(V3<S> a, V3<S> b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// cross with a constant (double) vector
template <class S> HD V3<S> crossc(V3<S> a, double bx, double by, double bz) { return {a.y * bz - a.z * by, a.z * bx - a.x * bz, a.x * by - a.y * bx}; }
template <class S> HD V3<S> ccross(double ax, double ay, double az, V3<S> b) { return {ay * b.z - az * b.y, az * b.x - ax * b.z, ax * b.y - ay * b.x}; }

// axis rotations: rot<AX>(c,s,w) = R_AX(theta) w ; rotT = R^T w
template <int AX, class S, class T> HD V3<S> rot(T c, T s, V3<S> w) {
    if (AX == 0) return {w.x, c * w.y - s * w.z, s * w.y + c * w.z};
    if (AX == 1) return {c * w.x + s * w.z, w.y, c * w.z - s * w.x};
    return {c * w.x - s * w.y, s * w.x + c * w.y, w.z};
}
template <int AX, class S, class T> HD V3<S> rotT(T c, T s, V3<S> w) {
    if (AX == 0) return {w.x, c * w.y + s * w.z, c * w.z - s * w.y};
    if (AX == 1) return {c * w.x - s * w.z, w.y, s * w.x + c * w.z};
    return {c * w.x + s * w.y, c * w.y - s * w.x, w.z};
}

}  // namespace hs
