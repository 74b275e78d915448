// tdlo_devcommon.h -- device-side helpers shared by the kernel translation units.
#pragma once
#include "tdlo_internal.h"
#include "../../include/trackdlo_hip.h"

namespace tdlo {

template <typename T> struct alignas(16) V4 { T x, y, z, w; };

__device__ __forceinline__ float tmin(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ double tmin(double a, double b) { return ::fmin(a, b); }
__device__ __forceinline__ float tabs(float a) { return __builtin_fabsf(a); }
__device__ __forceinline__ double tabs(double a) { return __builtin_fabs(a); }

// Address-space casts.  Pointers reach the kernels through the FrameDev descriptor, so the compiler
// only sees generic (flat) pointers and would emit flat_load + full waits.  Node data and parameters
// are never written by the kernel that reads them, so they are read through the constant address
// space (wave-uniform index => s_load into SGPRs, no VALU/LDS/VMEM cost); the cloud through global.
#define TDLO_AS_CONST(TYPE, p) ((const __attribute__((address_space(4))) TYPE *)(uintptr_t)(p))
#define TDLO_AS_GLOBAL(TYPE, p) ((const __attribute__((address_space(1))) TYPE *)(uintptr_t)(p))
#define TDLO_AS_GLOBAL_RW(TYPE, p) ((__attribute__((address_space(1))) TYPE *)(uintptr_t)(p))

template <typename T> struct Num;
template <> struct Num<float> {
    static __device__ __forceinline__ float inf() { return __builtin_huge_valf(); }
    static __device__ __forceinline__ float exp2(float x) { return __builtin_amdgcn_exp2f(x); }
    static __device__ __forceinline__ float sqrt(float x) { return __builtin_sqrtf(x); }
    // one instruction each (v_sqrt_f32, 1 ulp; v_rcp_f32 + one Newton step): the correctly rounded forms cost ~12 VALU
    // instructions apiece, and a point's distances / normalisation carry a 1e-7 relative error either way
    static __device__ __forceinline__ float sqrt_fast(float x) { return __builtin_amdgcn_sqrtf(x); }
    static __device__ __forceinline__ float rcp_fast(float x) { const float r = __builtin_amdgcn_rcpf(x); return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r); }
    static __device__ __forceinline__ unsigned long long bits(float x) { return (unsigned long long)__float_as_uint(x); }
    static __device__ __forceinline__ double from_bits(unsigned long long b) { return (double)__uint_as_float((unsigned)b); }
};
template <> struct Num<double> {
    static __device__ __forceinline__ double inf() { return __builtin_huge_val(); }
    // 2^x in 17 instructions: n = rint(x), f = x - n in [-1/2, 1/2] (exact), 2^f by the Taylor polynomial of degree 13 in f (coefficients
    // ln2^k / k!, truncation 4e-18, 0.8 ulp with the rounding of the Horner steps), scaled by v_ldexp_f64 (which also delivers the denormals and the
    // exact zero below 2^-1075 that the E-step's node window relies on).  The library's exp2 is the same scheme, but its Horner steps are
    // v_fmac_f64 INTO the coefficient registers, so every evaluation re-creates its nine coefficients (v_mov_b64 each) and ends in two range
    // compares and three selects: 32 instructions -- two thirds of a (point, node) pair of the fp64 E-step.  Here the steps are the
    // three-address v_fma_f64 (inline asm; the coefficients stay in registers across the loop) and the range handling is the ldexp's own.
    static __device__ __forceinline__ double exp2(double x) {
        const double n = __builtin_rint(x), f = x - n;
        double p;
        // (one asm statement: between separate ones the compiler pads every dependent pair with an s_nop, not knowing what the first one wrote)
        asm("v_fma_f64 %0, %1, %2, %3\n\tv_fma_f64 %0, %1, %0, %4\n\tv_fma_f64 %0, %1, %0, %5\n\tv_fma_f64 %0, %1, %0, %6\n\t"
            "v_fma_f64 %0, %1, %0, %7\n\tv_fma_f64 %0, %1, %0, %8\n\tv_fma_f64 %0, %1, %0, %9\n\tv_fma_f64 %0, %1, %0, %10\n\t"
            "v_fma_f64 %0, %1, %0, %11\n\tv_fma_f64 %0, %1, %0, %12\n\tv_fma_f64 %0, %1, %0, %13\n\tv_fma_f64 %0, %1, %0, %14"
            : "=&v"(p)
            : "v"(f), "v"(0x1.816193166d0f9p-40), "v"(0x1.c3bd650fc2986p-36), "v"(0x1.e8cac7351bb25p-32), "v"(0x1.e4cf5158b8ecap-28),
              "v"(0x1.b5253d395e7c4p-24), "v"(0x1.62c0223a5c824p-20), "v"(0x1.ffcbfc588b0c7p-17), "v"(0x1.430912f86c787p-13),
              "v"(0x1.5d87fe78a6731p-10), "v"(0x1.3b2ab6fba4e77p-7), "v"(0x1.c6b08d704a0c0p-5), "v"(0x1.ebfbdff82c58fp-3), "v"(0x1.62e42fefa39efp-1));
        p = __builtin_fma(f, p, 1.0);
        return __builtin_ldexp(p, (int)n);
    }
    static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
    static __device__ __forceinline__ double sqrt_fast(double x) { return ::sqrt(x); }
    static __device__ __forceinline__ double rcp_fast(double x) { return 1.0 / x; }
    static __device__ __forceinline__ unsigned long long bits(double x) { return (unsigned long long)__double_as_longlong(x); }
    static __device__ __forceinline__ double from_bits(unsigned long long b) { return __longlong_as_double((long long)b); }
};

// Squared node-to-point distance of the visibility pre-pass (trackdlo_node.cpp:257-277), written out as the fused operations it is computed in: three
// kernels form it (k_node_min_dist, k_node_min_dist_direct, the depth -> cloud team kernel's own pre-pass) and must give the same bits.
__device__ __forceinline__ double node_point_d2(double yx, double yy, double yz, double x, double y, double z) {
    const double dx = yx - x, dy = yy - y, dz = yz - z;
    return __builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx));
}

// LDS hand-off between lanes of ONE wave: DS operations of a wave execute in order, so only the
// compiler has to be kept from reordering across the point.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// value of the first active lane, as a wave-uniform (SGPR) operand
__device__ __forceinline__ float bcast_first(float v) { return __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v))); }
__device__ __forceinline__ double bcast_first(double v) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Wave-wide min / max of NON-NEGATIVE values (+inf allowed), wave-uniform result.  fp32: the IEEE bit patterns of
// non-negative floats order like unsigned integers, so the reduction runs on v_min_u32 / v_max_u32 with DPP operands
// (row_shr 1, 2, 4, 8 inside each row of 16 lanes, then row_bcast 15 / 31 across rows; total in lane 63): 6 VALU
// instructions and no LDS traffic, against 6 dependent ds_bpermute round trips for the __shfl_xor butterfly.
template <bool MAX> __device__ __forceinline__ unsigned wave_minmax_u32(unsigned u) {
    constexpr int idn = MAX ? 0 : -1;                      // identity: lanes without a source keep it
#define TDLO_DPP_STEP(CTRL, ROWMASK) do { \
        const unsigned o_ = (unsigned)__builtin_amdgcn_update_dpp(idn, (int)u, CTRL, ROWMASK, 0xf, false); \
        u = MAX ? (u > o_ ? u : o_) : (u < o_ ? u : o_); } while (0)
    TDLO_DPP_STEP(0x111, 0xf);      // row_shr:1
    TDLO_DPP_STEP(0x112, 0xf);      // row_shr:2
    TDLO_DPP_STEP(0x114, 0xf);      // row_shr:4
    TDLO_DPP_STEP(0x118, 0xf);      // row_shr:8   -> lane 15 of every row holds the row's result
    TDLO_DPP_STEP(0x142, 0xa);      // row_bcast:15 into rows 1 and 3
    TDLO_DPP_STEP(0x143, 0xc);      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's result
#undef TDLO_DPP_STEP
    return (unsigned)__builtin_amdgcn_readlane((int)u, 63);
}
__device__ __forceinline__ float wave_min_nonneg(float v) { return __uint_as_float(wave_minmax_u32<false>(__float_as_uint(v))); }
__device__ __forceinline__ float wave_max_nonneg(float v) { return __uint_as_float(wave_minmax_u32<true>(__float_as_uint(v))); }
// Two or three wave-wide maxima of unsigned keys at once (round 4): v_permlane32_swap folds the halves of TWO values with one exchange,
// v_permlane16_swap the row pairs, four DPP steps finish inside the rows -- rows 0 / 2 / 1 end with the maxima of a / b / c in their lane 15.
// 13 instructions for three reductions (10 for two) instead of 7 each.  A minimum of non-negative floats is the maximum of the complemented bits.
__device__ __forceinline__ unsigned fold32_max(unsigned x, unsigned y) {
    const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    return r[0] > r[1] ? r[0] : r[1];
}
__device__ __forceinline__ unsigned fold16_max(unsigned x, unsigned y) {
    const auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    return r[0] > r[1] ? r[0] : r[1];
}
__device__ __forceinline__ unsigned rows_max_u32(unsigned u) {          // the maximum of every row of 16 lanes in its lane 15
#define TDLO_DPP_STEP(CTRL) do { const unsigned o_ = (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, CTRL, 0xf, 0xf, false); u = u > o_ ? u : o_; } while (0)
    TDLO_DPP_STEP(0x111); TDLO_DPP_STEP(0x112); TDLO_DPP_STEP(0x114); TDLO_DPP_STEP(0x118);      // row_shr:1, 2, 4, 8
#undef TDLO_DPP_STEP
    return u;
}
// (max a, min b) of non-negative floats, wave-uniform
__device__ __forceinline__ void wave_max_min_nonneg(float a, float b, float &amax, float &bmin) {
    const unsigned x = fold32_max(__float_as_uint(a), ~__float_as_uint(b));
    const unsigned z = rows_max_u32(fold16_max(x, x));
    amax = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)z, 15));
    bmin = __uint_as_float(~(unsigned)__builtin_amdgcn_readlane((int)z, 47));
}
// (min a, max b, max c) of non-negative floats, wave-uniform
__device__ __forceinline__ void wave_min_max_max_nonneg(float a, float b, float c, float &amin, float &bmax, float &cmax) {
    const unsigned x = fold32_max(~__float_as_uint(a), __float_as_uint(b)), y = fold32_max(__float_as_uint(c), __float_as_uint(c));
    const unsigned z = rows_max_u32(fold16_max(x, y));
    amin = __uint_as_float(~(unsigned)__builtin_amdgcn_readlane((int)z, 15));
    cmax = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)z, 31));
    bmax = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)z, 47));
}
// fp64: the same butterfly as wave_sum below (v_permlane32_swap / v_permlane16_swap, then DPP rotations inside the rows), result in every
// lane; min / max are exact, so the order does not matter (the former __shfl_xor tree cost the fp64 E-step 24 LDS round trips per batch)
template <bool MAX, int CTRL> __device__ __forceinline__ double dpp_minmax_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false), hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    const double o = __hiloint2double(hi, lo);
    return MAX ? (v > o ? v : o) : (v < o ? v : o);
}
template <bool MAX> __device__ __forceinline__ double wave_minmax_f64(double v) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto l32 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), h32 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    const double a = __hiloint2double((int)h32[0], (int)l32[0]), b = __hiloint2double((int)h32[1], (int)l32[1]);
    double z = MAX ? (a > b ? a : b) : (a < b ? a : b);
    const unsigned zl = (unsigned)__double2loint(z), zh = (unsigned)__double2hiint(z);
    const auto l16 = __builtin_amdgcn_permlane16_swap(zl, zl, false, false), h16 = __builtin_amdgcn_permlane16_swap(zh, zh, false, false);
    const double c = __hiloint2double((int)h16[0], (int)l16[0]), d = __hiloint2double((int)h16[1], (int)l16[1]);
    z = MAX ? (c > d ? c : d) : (c < d ? c : d);
    z = dpp_minmax_f64<MAX, 0x128>(z);      // row_ror:8
    z = dpp_minmax_f64<MAX, 0x124>(z);      // row_ror:4
    z = dpp_minmax_f64<MAX, 0x122>(z);      // row_ror:2
    z = dpp_minmax_f64<MAX, 0x121>(z);      // row_ror:1
    return z;
}
__device__ __forceinline__ double wave_min_nonneg(double v) { return wave_minmax_f64<false>(v); }
__device__ __forceinline__ double wave_max_nonneg(double v) { return wave_minmax_f64<true>(v); }
__device__ __forceinline__ void wave_max_min_nonneg(double a, double b, double &amax, double &bmin) { amax = wave_max_nonneg(a); bmin = wave_min_nonneg(b); }
__device__ __forceinline__ void wave_min_max_max_nonneg(double a, double b, double c, double &amin, double &bmax, double &cmax) { amin = wave_min_nonneg(a); bmax = wave_max_nonneg(b); cmax = wave_max_nonneg(c); }

// wave-wide sum, the total in every lane.  The same pairs in the same order as an xor-32, 16, 8, 4, 2, 1 shuffle tree (identical bits), but
// without the LDS crossbar: v_permlane32_swap / v_permlane16_swap (gfx950) lay the two halves / the rows 16 apart of the value side by
// side, DPP rotations inside the rows do the rest (a rotation by 8, 4, 2, 1 of a value that already has that period pairs lane l with l ^ 8, ...).
template <int CTRL> __device__ __forceinline__ double dpp_rot_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false), hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto l32 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), h32 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    double z = __hiloint2double((int)h32[0], (int)l32[0]) + __hiloint2double((int)h32[1], (int)l32[1]);
    const unsigned zl = (unsigned)__double2loint(z), zh = (unsigned)__double2hiint(z);
    const auto l16 = __builtin_amdgcn_permlane16_swap(zl, zl, false, false), h16 = __builtin_amdgcn_permlane16_swap(zh, zh, false, false);
    z = __hiloint2double((int)h16[0], (int)l16[0]) + __hiloint2double((int)h16[1], (int)l16[1]);
    z += dpp_rot_f64<0x128>(z);      // row_ror:8
    z += dpp_rot_f64<0x124>(z);      // row_ror:4
    z += dpp_rot_f64<0x122>(z);      // row_ror:2
    z += dpp_rot_f64<0x121>(z);      // row_ror:1
    return z;
}

// Folding butterflies (gfx950 v_permlane32_swap / v_permlane16_swap): fold32(x, y) leaves x[l] + x[l + 32] in lanes 0..31 and y[l - 32] + y[l] in
// lanes 32..63 -- ONE exchange folds the halves of TWO values; fold16(x, y) does the same for rows 16 lanes apart: rows 0 and 2 receive x's row
// pairs (0, 1) and (2, 3), rows 1 and 3 y's.  With x == y every lane receives its pair's sum (a plain xor-32 / xor-16 butterfly step).
__device__ __forceinline__ float fold32(float x, float y) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float fold16(float x, float y) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ double fold32(double x, double y) {
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double fold16(double x, double y) {
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}

// block-wide sum for kBlock threads; scratch holds >= 4 doubles; result valid in every thread
__device__ __forceinline__ double block_sum(double v, double *scratch) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

// ---- the E-step's sums [P1 | Rx | Ry | Rz | Q] (trackdlo.cpp:386-389 in residual form) as 64-bit fixed-point accumulators ----------------
// The conversion happens at the finest grain: one wave's share of one 64-point batch.  Everything after it -- the wave's running sums over
// its batches, the workgroup's sum over its waves, the atomics of the workgroups into one of kAccRows replica rows -- is INTEGER addition,
// which is associative: the totals depend neither on the order in which workgroups arrive nor on how the batches are dealt out to waves
// and workgroups.  Batches, repetitions and any launch geometry give the same bits; and the M-step fetches kAccRows short rows instead
// of one row per workgroup (98 rows = 80 KB through one CU took 3 us at N = 50 000; 512 rows needed a reduction kernel of their own).
// Resolution 2^-sh: sh is chosen per quantity from the cloud size and an ASSUMED extent D of the scene so that totals stay below 2^62 and a
// batch's share below 2^51 (prepare_frame): shP = min(60 - ceil(log2 N), 44) for P1, shP - ld for R, shP - 2 ld for Q, 2^ld >= D.  At
// N = 50 000 that is 6e-14 on P1 ~ 1000 -- finer than the fp32 rows it replaces by seven decimal digits.  It is NOT uniformly "fp64 level":
// at N = 2 000 000 shP falls to 39 and Q, with 2 ld bits less, resolves ~2^-35 m^2 -- about fp32-level relative accuracy once sigma2 is
// small.  That is ample for what Q feeds (sigma2, gated at 1e-3 relative in fp32 mode, 1e-7 in fp64 mode at the sizes the gates are tested
// on); a caller who needs more at multi-million-point clouds splits the cloud (the shards' exponents follow their own N).
// The extent is a heuristic: every converted value is therefore CHECKED against FrameDev::acc_lim in the E-step (exact conversion below 2^51,
// totals below 2^62, NaN fails the comparison) and a violation ends the registration with TDLO_E_NUMERIC -- never a wrapped-around integer.
__host__ __device__ inline int acc_stride(int M) { return 4 * M + 2; }
// (the sums the M-step reads were formed by the E-step in front of it with IterState::sh_boost extra digits for R, twice as many for Q: the word is read
//  before the M-step's own set_iter_consts replaces it)
__device__ __forceinline__ int acc_shift(const FrameDev &f, int i) {
    const int M = f.M, b = TDLO_AS_GLOBAL(IterState, f.st)->sh_boost;
    return i < M ? f.acc_sh[0] : (i < 4 * M ? f.acc_sh[1] + b : f.acc_sh[2] + 2 * b);
}
// double -> fixed point, round to nearest, |v * 2^sh| < 2^51 (checked by the caller against FrameDev::acc_lim): the sum v * 2^sh + 1.5 * 2^52 has unit spacing, so its low mantissa bits ARE the integer -- one FMA and
// one 64-bit subtraction instead of the ~12 instructions of a double -> int64 conversion
__device__ __forceinline__ double acc_scale(int sh) { return __hiloint2double((1023 + sh) << 20, 0); }
__device__ __forceinline__ long long acc_fix(double v, double scale) {
    return __double_as_longlong(::fma(v, scale, 6755399441055744.0)) - 0x4338000000000000ll;
}
__device__ __forceinline__ void acc_add(long long *row, int i, long long iv) {
    if (iv != 0) __hip_atomic_fetch_add(row + i, iv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // exact zeros (nodes outside the window) add nothing
}
__device__ __forceinline__ int acc_rows_used(const FrameDev &f) { const int r = f.acc_rows; return (r == 2 || r == 4) ? r : kAccRows; }
__device__ __forceinline__ const long long *acc_rows(const FrameDev &f, int it) { return f.acc + (size_t)(it & 1) * kAccRows * acc_stride(f.M); }
// element i of the sums of iteration `it`: the replica rows are added as integers (exact), one conversion
__device__ __forceinline__ double acc_read(const FrameDev &f, const long long *rows, int i) {
    const auto a = TDLO_AS_GLOBAL(long long, rows);
    const int st = acc_stride(f.M);
    long long s = 0;
#pragma unroll
    for (int r = 0; r < kAccRows; ++r) s += a[(size_t)r * st + i];
    return ::ldexp((double)s, -acc_shift(f, i));
}
// the same without a load that depends on the iteration counter (both parities are fetched, one is kept): for the one-workgroup M-steps,
// whose first memory round trip is on the critical path of every iteration
// (ROWS: the rows the E-step used, FrameDev::acc_rows -- the others are zero; a kernel instantiated for fewer rows fetches less in the round trip it waits for)
template <int ROWS = kAccRows>
__device__ __forceinline__ double acc_read_both(const FrameDev &f, int i, int it) {
    const auto a = TDLO_AS_GLOBAL(long long, f.acc);
    const int st = acc_stride(f.M);
    long long s0 = 0, s1 = 0;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) { s0 += a[(size_t)r * st + i]; s1 += a[(size_t)(kAccRows + r) * st + i]; }
    return ::ldexp((double)((it & 1) ? s1 : s0), -acc_shift(f, i));
}
// ... and when the launch was TOLD the iteration's parity (the host counts the iterations it enqueues: mstep_parity_hint in tdlo_mstep_chain.hip), that
// parity's rows alone -- half the lines of the round trip every iteration waits for.  The kernel checks the word against the device's counter afterwards.
template <int ROWS = kAccRows>
__device__ __forceinline__ double acc_read_par(const FrameDev &f, int i, int par) {
    const auto a = TDLO_AS_GLOBAL(long long, f.acc) + (size_t)par * kAccRows * acc_stride(f.M);
    const int st = acc_stride(f.M);
    long long s = 0;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) s += a[(size_t)r * st + i];
    return ::ldexp((double)s, -acc_shift(f, i));
}
// the rows of the other parity are cleared for the next E-step (nobody touches them while an M-step runs)
template <int NT> __device__ __forceinline__ void acc_clear_other(const FrameDev &f, int it, int t) {
    long long *z = f.acc + (size_t)((it + 1) & 1) * kAccRows * acc_stride(f.M);
    const int n = kAccRows * acc_stride(f.M);
    for (int i = t; i < n; i += NT) z[i] = 0;
}

__device__ __forceinline__ void set_iter_consts(const FrameDev &f, IterState *st, double sigma2, double Nc) {
    // c of trackdlo.cpp:300 (or c' of :378 when visibility weighting is active)
    const double tp = 2.0 * M_PI * sigma2, rtp = ::sqrt(tp);
    double c = tp * rtp * f.mu / (1.0 - f.mu);                 // (2 pi sigma2)^(3/2): one square root instead of pow() at the end of every M-step
    c = f.vis_branch ? c / Nc : c * (double)f.M / Nc;
    st->sigma2 = sigma2;
    st->Nc = Nc;
    st->k2 = -1.4426950408889634 / (2.0 * sigma2);
    st->c_norm = c;
    // the E-step's node window (FrameDev::win_e32): E / |k2| = E 2 ln2 sigma2
    st->rwin32 = f.win_e32 * 1.3862943611198906 * sigma2;
    st->rwin64 = f.win_e64 * 1.3862943611198906 * sigma2;
    // fp64 mode: the resolution of the fixed-point sums follows sigma.  FrameDev::acc_sh is sized for point-node distances up to twice the chain's
    // length (a wave's share of R_m = sum p (x - y_m) is bounded by 64 D); but a normalised membership times its distance is at most about
    // d_nearest + 0.61 sigma (p <= exp(-arc^2 / 2 sigma2), |x - y_m| <= d_nearest + arc), so once sigma is centimetres the shares are a hundredth of that
    // bound and the sums can carry ld - ld_eff more digits, D_eff = 2 (0.4 m + 2 sigma) -- four times the prune's 0.1 m for nodes that have moved.  The
    // E-step still CHECKS every share against the (finer) limit.  With lambda sigma2 ~ 1e-5 (lambda = 1 without the LLE term) the coarse resolution, 2^-39 m
    // per share on a chain of 460 nodes, had put nodes 1e-8 m from the oracle's (profiles/r05_fuzz.log); fp32 mode's tile sums carry 1e-7 relative anyway.
    int boost = 0;
    if (f.precision == TDLO_PREC_F64 && !f.acc_boost_off && sigma2 > 0.0) {
        const int ld = f.acc_sh[0] - f.acc_sh[1];
        const double deff = 2.0 * (0.4 + 2.0 * ::sqrt(sigma2));
        int lde = 0;                                   // (D_eff >= 0.8: 2^0 is the smallest extent there is)
        while (lde < ld && ::ldexp(1.0, lde) < deff) ++lde;
        boost = ld - lde;
    }
    st->sh_boost = boost;
}


// ---- shared by the E-step kernels (tdlo_device.hip: one point per lane; tdlo_estep2.hip: two) ----------------------------------------------
// four consecutive nodes {x, y, z, coord} by ONE scalar load (s_load_dwordx16; two of them in fp64)
template <typename T> struct Node4 { T v[16]; };
template <typename T> __device__ __forceinline__ Node4<T> load_node4(const void *nodes, int m0) {
    typedef T vec16 __attribute__((ext_vector_type(16)));
    typedef vec16 __attribute__((aligned(16))) vec16a;
    const vec16 r = *(const __attribute__((address_space(4))) vec16a *)((uintptr_t)nodes + (size_t)m0 * 4 * sizeof(T));
    Node4<T> o;
#pragma unroll
    for (int i = 0; i < 16; ++i) o.v[i] = r[i];
    return o;
}

// Geodesic membership of node m for a point whose nearest pair is (lo, hi):
//   m <= lo : (coord_lo - coord_m + |x - y_lo|)^2      m >= hi : (coord_m - coord_hi + |x - y_hi|)^2
//   lo < m < hi (only when hi - lo == 2, the reference's end-node quirk): 0            (:332-350)
template <typename T>
__device__ __forceinline__ T geo_arg(int m, int lo, int hi, T cm, T c_lo, T d_lo, T c_hi, T d_hi) {
    const T t_lo = (c_lo - cm) + d_lo;
    const T t_hi = (cm - c_hi) + d_hi;
    T t = (m <= lo) ? t_lo : T(0);
    t = (m >= hi) ? t_hi : t;
    return t * t;
}

// The same value when hi == lo + 1 (every point but the rare end-node case of :313-321, where the node between lo and
// hi keeps the zero of :305): coord is non-decreasing, so c_lo - cm = |cm - c_lo| for m <= lo and cm - c_hi = |cm - c_hi|
// for m > lo -- one compare, two selects, |a - b| + d instead of both branches and two compare/select pairs.
template <typename T>
__device__ __forceinline__ T geo_arg_adj(int m, int lo, T cm, T c_lo, T d_lo, T c_hi, T d_hi) {
    const bool s = m <= lo;
    const T c = s ? c_lo : c_hi, d = s ? d_lo : d_hi;
    const T t = tabs(cm - c) + d;
    return t * t;
}

// wave-wide sum of a 64-bit integer without the LDS crossbar: v_permlane32_swap / v_permlane16_swap put the two halves (row pairs) of the
// value side by side, four DPP rotations finish inside the rows; every lane returns the total (integer addition: any order)
template <int CTRL> __device__ __forceinline__ long long dpp_i64(long long v) {
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, 0xf, 0xf, false), hi = __builtin_amdgcn_update_dpp(0, (int)(v >> 32), CTRL, 0xf, 0xf, false);
    return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ long long wave_sum_i64(long long v) {
    const unsigned lo = (unsigned)v, hi = (unsigned)((unsigned long long)v >> 32);
    const auto l32 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), h32 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    long long z = (long long)(((unsigned long long)h32[0] << 32) | l32[0]) + (long long)(((unsigned long long)h32[1] << 32) | l32[1]);       // lanes l and l ^ 32
    const unsigned zl = (unsigned)z, zh = (unsigned)((unsigned long long)z >> 32);
    const auto l16 = __builtin_amdgcn_permlane16_swap(zl, zl, false, false), h16 = __builtin_amdgcn_permlane16_swap(zh, zh, false, false);
    z = (long long)(((unsigned long long)h16[0] << 32) | l16[0]) + (long long)(((unsigned long long)h16[1] << 32) | l16[1]);                 // rows r and r ^ 1
    z += dpp_i64<0x128>(z);      // row_ror:8
    z += dpp_i64<0x124>(z);      // row_ror:4
    z += dpp_i64<0x122>(z);      // row_ror:2
    z += dpp_i64<0x121>(z);      // row_ror:1
    return z;
}

// the value of the lane 8 away in the same row of 16 (DPP row_ror:8)
__device__ __forceinline__ float row_ror8(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, true)); }
__device__ __forceinline__ double row_ror8(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x128, 0xf, 0xf, true), hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x128, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}


__device__ __forceinline__ double fast_rcp(double v) {
    double r = __builtin_amdgcn_rcp(v);
    r = fma(fma(-v, r, 1.0), r, r);
    r = fma(fma(-v, r, 1.0), r, r);
    return r;
}

// State-space form of the kernel G of trackdlo.cpp:233 (Matern-3/2 in the chain coordinate; see tdlo_mstep_chain.hip): the link
// over a gap h >= 0 between consecutive nodes, o = {Phi11, Phi12, Phi21, Phi22, Q11, Q12, Q22, 0},
//   Phi = e^-x [[1 + x, h], [-s^2 h, 1 - x]],  x = s h,  s = sqrt2 / beta,
//   Q   = Pinf - Phi Pinf Phi^T,  Pinf = sf2 diag(1, s^2),  sf2 = 1 / (2 sqrt2 beta):
//   Q11 = sf2 (1 - e^-2x (1 + 2x + 2x^2)),  Q12 = 2 sf2 s^3 h^2 e^-2x,  Q22 = sf2 s^2 (1 - e^-2x (1 - 2x + 2x^2)).
// For small gaps 1 - e^-2x (1 + 2x + 2x^2) is O(x^3) out of terms of size 1; the series e^-2x sum_{n >= 3} (2x)^n / n! has
// only positive terms (Q22 likewise: e^-2x (4x + sum_{n >= 3})), so every entry keeps full relative accuracy.
__device__ __forceinline__ void chain_link(double beta, double h, double *o) {
    const double s = ::sqrt(2.0) / beta, sf2 = 1.0 / (2.0 * ::sqrt(2.0) * beta);
    const double x = s * h, e = ::exp(-x), e2 = e * e;
    o[0] = e * (1.0 + x); o[1] = e * h; o[2] = -s * s * h * e; o[3] = e * (1.0 - x);
    double u11, u22;
    if (x < 1.0) {
        const double tt = 2.0 * x;
        double term = tt * tt * tt / 6.0, sum = term;
#pragma unroll
        for (int n = 4; n < 34; ++n) { term *= tt * (1.0 / (double)n); sum += term; }      // tt < 2: the 34th term is below 2^34 / 34! = 6e-29 of the first
        u11 = e2 * sum; u22 = e2 * (4.0 * x + sum);
    } else {
        u11 = 1.0 - e2 * (1.0 + 2.0 * x + 2.0 * x * x); u22 = 1.0 - e2 * (1.0 - 2.0 * x + 2.0 * x * x);
    }
    o[4] = sf2 * u11; o[5] = 2.0 * sf2 * s * s * s * h * h * e2; o[6] = sf2 * s * s * u22; o[7] = 0.0;
}

typedef double mfma_d4 __attribute__((ext_vector_type(4)));     // accumulator of v_mfma_f64_16x16x4_f64

// ---- results into pinned host memory (FrameDev::host_out / host_prog) -------------------------------------------------------------------
// Called by the 64 lanes of WAVE 0 of a one-workgroup M-step, after a workgroup barrier behind the stores to f.Yout and after lane 0 has
// updated *st (fresh == true), or at the kernel's early exit from a registration that is already done (fresh == false: only a registration
// that ended on an error is reported again -- the E-step's range check, an empty cloud --, a finished one has been reported by the M-step that
// finished it).  Yout and *st are read back through agent-scope loads (this CU's vector cache may hold lines of *st from the kernel's first
// loads), stored to the host with plain stores, fenced at system scope, then the progress word goes out.
static_assert(offsetof(IterState, N) == 72 && offsetof(IterState, it) == 76 && offsetof(IterState, done) == 80 && offsetof(IterState, status) == 88 &&
              sizeof(IterState) <= 13 * 8, "host_publish reads {N, it}, {done, converged}, {status, retry_pending} as 64-bit words");
__device__ __forceinline__ void host_publish(const FrameDev &f, IterState *st, int lane, bool fresh) {
    if (!f.host_prog) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // lane 0's stores to *st have been performed
    const unsigned long long *sw = (const unsigned long long *)st;
    int it = 0, done = 0, status = 0;
    if (lane == 0) {
        const unsigned long long w_it = __hip_atomic_load(sw + offsetof(IterState, it) / 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // {N, it}
        const unsigned long long w_dn = __hip_atomic_load(sw + offsetof(IterState, done) / 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // {done, converged}
        const unsigned long long w_st = __hip_atomic_load(sw + offsetof(IterState, status) / 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // {status, retry_pending}
        it = (int)(w_it >> 32); done = (int)(unsigned)w_dn; status = (int)(unsigned)w_st;
    }
    it = __builtin_amdgcn_readfirstlane(it); done = __builtin_amdgcn_readfirstlane(done); status = __builtin_amdgcn_readfirstlane(status);
    if (!fresh && status == 0) return;
    if (done) {
        const int n = 3 * f.M, so = (int)((const double *)st - f.Yout);
        const unsigned long long *yo = (const unsigned long long *)f.Yout;
        unsigned long long *ho = (unsigned long long *)f.host_out;
        for (int i = lane; i < n; i += 64) ho[i] = __hip_atomic_load(yo + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        constexpr int nst = (int)((sizeof(IterState) + 7) / 8);
        if (lane < nst) ho[so + lane] = __hip_atomic_load(sw + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    }
    if (lane == 0)
        __hip_atomic_store(f.host_prog, ((unsigned long long)f.host_epoch << 32) | ((unsigned long long)(done ? 1u : 0u) << 31) | (unsigned)(it & 0x7fffffff),
                           __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- spin-ahead loop (FrameDev::spin_on): bounded wait of ONE thread for a word of `sync` to take a value (agent scope); false: 2 s passed
__device__ __forceinline__ bool spin_wait_word(const unsigned *word, unsigned want) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {
        __builtin_amdgcn_s_sleep(1);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) return false;
    }
    return true;
}

// ---- one-shot exchange of the N-split: peer-written inboxes (xGMI peer stores on a multi-GPU node), system scope ---------
// payload: relaxed system-scope stores -> release fence (system) -> flag store; reader: relaxed poll of the flag ->
// acquire fence (system) -> relaxed system-scope loads of the payload.
typedef __attribute__((address_space(1))) unsigned long long xch_word;
__device__ __forceinline__ xch_word *xch_ptr(unsigned long long *p) { return (xch_word *)(uintptr_t)p; }
__device__ __forceinline__ void xch_store(xch_word *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void xch_store_f64(xch_word *p, double v) { xch_store(p, (unsigned long long)__double_as_longlong(v)); }
__device__ __forceinline__ unsigned long long xch_load(const xch_word *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ double xch_load_f64(const xch_word *p) { return __longlong_as_double((long long)xch_load(p)); }
__device__ __forceinline__ void xch_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); }
__device__ __forceinline__ void xch_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, ""); }
constexpr unsigned long long kXchSpinTicks = 200000000ull;      // 2 s of the 100 MHz real-time clock: a peer that is this late is gone
// A rank whose registration ended on an error of its OWN shard (the E-step's range check: TDLO_E_NUMERIC) leaves the M-step before the exchange;
// it still raises its flag, with this mark in the tag, so that its peers leave with the same status instead of sitting out the time limit
// (tags proper carry the iteration number + 1 < 2^30 in their low word, the once-per-call exchange 0x80000000)
constexpr unsigned long long kXchErrMark = 0x40000000ull;
// waits until *flag == tag; false when the time limit passed
__device__ __forceinline__ bool xch_wait(const xch_word *flag, unsigned long long tag) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (xch_load(flag) != tag) {
        __builtin_amdgcn_s_sleep(2);
        if (__builtin_amdgcn_s_memrealtime() - t0 > kXchSpinTicks) return false;
    }
    return true;
}
// the same for the per-iteration sums: 1 = arrived, 0 = time limit, -1 = the peer reported an error of its own (kXchErrMark)
__device__ __forceinline__ int xch_wait_sums(const xch_word *flag, unsigned long long tag) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        const unsigned long long v = xch_load(flag);
        if (v == tag) return 1;
        if (v == (tag | kXchErrMark)) return -1;
        __builtin_amdgcn_s_sleep(2);
        if (__builtin_amdgcn_s_memrealtime() - t0 > kXchSpinTicks) return 0;
    }
}
// a finished registration's M-step in the one-shot exchange: if it finished on TDLO_E_NUMERIC, tell the peers (lane = peer)
__device__ __forceinline__ void xch_post_error(const FrameDev &f, const IterState *st, int t) {
    if (f.xch_nranks < 1 || st->status != TDLO_E_NUMERIC) return;
    const int R = f.xch_nranks, me = f.xch_rank, it = st->it, par = it & 1;
    const unsigned long long tag = ((unsigned long long)f.xch_epoch << 32) | (unsigned)(it + 1) | kXchErrMark;
    if (t < R) xch_store(xch_ptr(f.xch_inbox[t]) + xch_off_flag_sums(R) + par * R + me, tag);
}

}  // namespace tdlo
