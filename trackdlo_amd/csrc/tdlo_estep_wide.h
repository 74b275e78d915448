// The fp64 E-step's memberships, normalisation and column sums (trackdlo.cpp:332-389) for ONE 64-point batch whose node window is WIDE -- round 6.
//
// k_estep maps thread = point: a lane walks the window's nodes for its point, the memberships go through an LDS tile to be summed per node.  While sigma is
// still decimetres (the first iterations of a registration from sigma2 = 0) the window is the whole chain -- 300 nodes at BASELINE.json's configs[4] -- and a
// 24-row tile holds a twelfth of it: every later chunk of the window RECOMPUTES its memberships (68 vector instructions per node and batch, of which 17 + 17
// are the two evaluations of 2^x).  Here the mapping is turned round for such a batch: LANE = NODE (lane l holds nodes wlo + l, wlo + 64 + l, ... of the
// window, up to WCH of them), the batch's points come by one after the other as LDS broadcasts, a node's column sums P1_m, sum_n P_mn (x_n - o) stay in its
// lane's REGISTERS for the whole batch -- no tile, no transposition, every membership evaluated once.  What a point needs from all lanes is its denominator
// (:354-383): four points at a time (their four Horner chains of 2^x interleaved in one asm statement), the lanes' partial sums are folded so that each row of 16 lanes ends up with one point's total (fold32, fold16, four DPP
// steps: 21 instructions for four points), the reciprocal is taken there and handed to the wave by v_readlane.
// Per (64 nodes, point): geodesic argument 2 - 7, two multiplies, 2^x 17, the sum 1, the normalised membership and its four accumulations 5 = 27 - 32
// against 68; per point ~20 more (record reads, fold, reciprocal, the points' part of Q).  Measured (MI355X, N = 200 000, M = 300, scripts/gpu_estep_wide_trace.sh):
// the first E-steps 222 -> 144 us, a call's E-steps 1.53 -> 1.34 ms, the call 2.55 -> 2.34 ms (19.6 -> 21.4 k it/s); from 129 nodes on (FrameDev::estep_wide_min:
// narrower windows measured slower than the tile form -- the per-batch fixed work).  Both forms issue ~8 clocks per fp64 instruction, and 3 125 batches on 1 024
// SIMDs leave the fullest with 4 where 3.05 is the mean: what is left is instruction count and that quantisation.  The tail is the thread = point form's: residual R = s + (o - y) P1,
// the nodes' part of Q from the same sums, conversion to 64-bit fixed point at the grain of this one batch (so the totals stay independent of the launch
// geometry), range checks, ds_add_u64 into the workgroup's accumulators.
// The arithmetic is that of the thread = point form up to the order of the additions (a point's denominator is summed per lane, then across lanes; a node's
// column sums point by point) and the reciprocal (v_rcp_f64 + two Newton steps instead of a division): differences of 1e-16 relative, far inside the
// mode's 1e-9 m / 1e-7.
#pragma once
namespace tdlo {

// lo, hi: the point's nearest pair (:313-329); c_*, d_*: their chain coordinates and the point's distances to them; min_lo / max_hi: the smallest lo / largest
// hi of the batch's points, adj: no pair has the end-node gap.  rec: >= 64 x kWideRec doubles of this wave's LDS (the membership tile's place: unused here).
constexpr int kWideRec = 10;      // doubles per point record: c_lo d_lo | c_hi d_hi | (lo, hi) uu | ux uy | uz -

__device__ __forceinline__ double rcp_newton(double t) {          // 1 / t to the last place or the one before it (t > 0, normal)
    double r = __builtin_amdgcn_rcp(t);
    double e = __builtin_fma(-t, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-t, r, 1.0);
    return __builtin_fma(r, e, r);
}
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v) {
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ double readlane_pair(double v, int src) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}


// Four values of 2^x at once: Num<double>::exp2 (tdlo_devcommon.h) with the four Horner chains INTERLEAVED in one asm statement.  One evaluation is twelve
// dependent v_fma_f64 in a row, and the fp64 E-step of a long chain runs two waves per SIMD (its LDS): a dependent fp64 instruction issues every ~8 clocks where
// an independent one takes 4.  The same operations per value: the same bits.
__device__ __forceinline__ void exp2x4(const double (&x)[4], double (&out)[4]) {
    double n[4], f[4], p[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { n[k] = __builtin_rint(x[k]); f[k] = x[k] - n[k]; }
    asm("v_fma_f64 %0, %4, %8, %9\n\tv_fma_f64 %1, %5, %8, %9\n\tv_fma_f64 %2, %6, %8, %9\n\tv_fma_f64 %3, %7, %8, %9\n\t"
            "v_fma_f64 %0, %4, %0, %10\n\tv_fma_f64 %1, %5, %1, %10\n\tv_fma_f64 %2, %6, %2, %10\n\tv_fma_f64 %3, %7, %3, %10\n\t"
            "v_fma_f64 %0, %4, %0, %11\n\tv_fma_f64 %1, %5, %1, %11\n\tv_fma_f64 %2, %6, %2, %11\n\tv_fma_f64 %3, %7, %3, %11\n\t"
            "v_fma_f64 %0, %4, %0, %12\n\tv_fma_f64 %1, %5, %1, %12\n\tv_fma_f64 %2, %6, %2, %12\n\tv_fma_f64 %3, %7, %3, %12\n\t"
            "v_fma_f64 %0, %4, %0, %13\n\tv_fma_f64 %1, %5, %1, %13\n\tv_fma_f64 %2, %6, %2, %13\n\tv_fma_f64 %3, %7, %3, %13\n\t"
            "v_fma_f64 %0, %4, %0, %14\n\tv_fma_f64 %1, %5, %1, %14\n\tv_fma_f64 %2, %6, %2, %14\n\tv_fma_f64 %3, %7, %3, %14\n\t"
            "v_fma_f64 %0, %4, %0, %15\n\tv_fma_f64 %1, %5, %1, %15\n\tv_fma_f64 %2, %6, %2, %15\n\tv_fma_f64 %3, %7, %3, %15\n\t"
            "v_fma_f64 %0, %4, %0, %16\n\tv_fma_f64 %1, %5, %1, %16\n\tv_fma_f64 %2, %6, %2, %16\n\tv_fma_f64 %3, %7, %3, %16\n\t"
            "v_fma_f64 %0, %4, %0, %17\n\tv_fma_f64 %1, %5, %1, %17\n\tv_fma_f64 %2, %6, %2, %17\n\tv_fma_f64 %3, %7, %3, %17\n\t"
            "v_fma_f64 %0, %4, %0, %18\n\tv_fma_f64 %1, %5, %1, %18\n\tv_fma_f64 %2, %6, %2, %18\n\tv_fma_f64 %3, %7, %3, %18\n\t"
            "v_fma_f64 %0, %4, %0, %19\n\tv_fma_f64 %1, %5, %1, %19\n\tv_fma_f64 %2, %6, %2, %19\n\tv_fma_f64 %3, %7, %3, %19\n\t"
            "v_fma_f64 %0, %4, %0, %20\n\tv_fma_f64 %1, %5, %1, %20\n\tv_fma_f64 %2, %6, %2, %20\n\tv_fma_f64 %3, %7, %3, %20"
        : "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3])
        : "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]),
          "v"(0x1.816193166d0f9p-40), "v"(0x1.c3bd650fc2986p-36), "v"(0x1.e8cac7351bb25p-32), "v"(0x1.e4cf5158b8ecap-28), "v"(0x1.b5253d395e7c4p-24), "v"(0x1.62c0223a5c824p-20), "v"(0x1.ffcbfc588b0c7p-17), "v"(0x1.430912f86c787p-13), "v"(0x1.5d87fe78a6731p-10), "v"(0x1.3b2ab6fba4e77p-7), "v"(0x1.c6b08d704a0c0p-5), "v"(0x1.ebfbdff82c58fp-3), "v"(0x1.62e42fefa39efp-1));
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = __builtin_ldexp(__builtin_fma(f[k], p[k], 1.0), (int)n[k]);
}

template <int WCH, bool VIS>
__device__ __forceinline__ void estep_wide_batch(int lane, int wlo, int whi, int n0, int N, bool adj, int min_lo, int max_hi,
                                                 double x, double y, double z, int lo, int hi, double c_lo, double d_lo, double c_hi, double d_hi,
                                                 double k2, double cn, const V4<double> *nodesL, const double *lvL, double *rec, long long *accL,
                                                 double scP, double scR, double scQ, double limP, double limR, double limQ, double limQn,
                                                 long long &accQ, bool &acc_ok) {
    const double ox = bcast_first(x), oy = bcast_first(y), oz = bcast_first(z);
    {
        const double ux = x - ox, uy = y - oy, uz = z - oz;
        double *r = rec + lane * kWideRec;
        r[0] = c_lo; r[1] = d_lo; r[2] = c_hi; r[3] = d_hi;
        ((int *)(r + 4))[0] = lo; ((int *)(r + 4))[1] = hi;
        r[5] = ux * ux + uy * uy + uz * uz;
        r[6] = ux; r[7] = uy; r[8] = uz;
    }
    const int W = whi - wlo + 1;
    const int nch = (W + 63) >> 6;                      // <= WCH (the caller's test)
    // this lane's nodes: chain coordinate and visibility term (a lane behind the window's end computes on the window's last node; its memberships are set to 0)
    double cmw[WCH], lvw[WCH];
#pragma unroll
    for (int c = 0; c < WCH; ++c) {
        const int m = wlo + 64 * c + lane;
        const int mc = (c < nch && m <= whi) ? m : whi;
        cmw[c] = nodesL[mc].w;
        lvw[c] = VIS ? lvL[mc] : 0.0;
    }
    const bool off_last = wlo + 64 * (nch - 1) + lane > whi;             // this lane has no node in the window's last chunk
    double P1[WCH], Sx[WCH], Sy[WCH], Sz[WCH];
#pragma unroll
    for (int c = 0; c < WCH; ++c) { P1[c] = 0; Sx[c] = 0; Sy[c] = 0; Sz[c] = 0; }
    wave_lds_sync();
    const int row = lane >> 4;
    const int rowpt = ((row & 1) << 1) | (row >> 1);                     // the point (of a group of four) whose total this lane's row receives: 0, 2, 1, 3
    for (int g = 0; g < 16; ++g) {
        double p[4][WCH], s[4];
        double pc_lo[4], pd_lo[4], pc_hi[4], pd_hi[4];
        int plo[4], phi[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double *r = rec + (4 * g + j) * kWideRec;              // (every lane the same address: LDS broadcasts)
            pc_lo[j] = r[0]; pd_lo[j] = r[1]; pc_hi[j] = r[2]; pd_hi[j] = r[3];
            plo[j] = ((const int *)(r + 4))[0]; phi[j] = ((const int *)(r + 4))[1];
            s[j] = 0;
        }
#pragma unroll
        for (int c = 0; c < WCH; ++c) {
#pragma unroll
            for (int j = 0; j < 4; ++j) p[j][c] = 0;
            if (c < nch) {                                               // wave-uniform
                const int mfirst = wlo + 64 * c;
                const double cm = cmw[c];
                double e[4], pv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    double t;
                    if (adj && mfirst + 63 <= min_lo) t = (pc_lo[j] - cm) + pd_lo[j];           // every node of the chunk at or below every point's lo
                    else if (adj && mfirst >= max_hi) t = (cm - pc_hi[j]) + pd_hi[j];           // ... at or above every point's hi
                    else {                                                                       // geo_arg (tdlo_devcommon.h), the end-node gap's zero included
                        const int m = mfirst + lane;
                        const double t_lo = (pc_lo[j] - cm) + pd_lo[j], t_hi = (cm - pc_hi[j]) + pd_hi[j];
                        t = (m <= plo[j]) ? t_lo : 0.0;
                        t = (m >= phi[j]) ? t_hi : t;
                    }
                    e[j] = (t * t) * k2;
                    if (VIS) e[j] += lvw[c];
                }
                exp2x4(e, pv);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (c == nch - 1 && off_last) pv[j] = 0.0;
                    p[j][c] = pv[j];
                    s[j] += pv[j];
                }
            }
        }
        // the four points' denominators: rows 0, 1, 2, 3 receive the totals of points 0, 2, 1, 3
        double tot = fold16(fold32(s[0], s[1]), fold32(s[2], s[3]));
        tot += dpp_f64<0x128>(tot);      // row_ror:8
        tot += dpp_f64<0x124>(tot);      // row_ror:4
        tot += dpp_f64<0x122>(tot);      // row_ror:2
        tot += dpp_f64<0x121>(tot);      // row_ror:1
        const int jn = 4 * g + rowpt;
        const bool pvalid = n0 + jn < N;
        const double inv = pvalid ? rcp_newton(tot + cn) : 0.0;
        {   // the points' part of Q: Pt1_n |x_n - o|^2, once per point (the first lane of its row)
            const double uu = rec[jn * kWideRec + 5];
            const double qv = (lane & 15) == 0 ? inv * (tot * uu) : 0.0;
            acc_ok &= __builtin_fabs(qv) < limQ; accQ += acc_fix(qv, scQ);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double invj = readlane_pair(inv, j == 0 ? 0 : (j == 1 ? 32 : (j == 2 ? 16 : 48)));
            const double *r = rec + (4 * g + j) * kWideRec;
            const double ux = r[6], uy = r[7], uz = r[8];
#pragma unroll
            for (int c = 0; c < WCH; ++c) {
                if (c < nch) {
                    const double w = p[j][c] * invj;
                    P1[c] += w;
                    Sx[c] = __builtin_fma(w, ux, Sx[c]); Sy[c] = __builtin_fma(w, uy, Sy[c]); Sz[c] = __builtin_fma(w, uz, Sz[c]);
                }
            }
        }
    }
    // ---- tail: this batch's share of the node's sums, residual form, 64-bit fixed point
    typedef __attribute__((address_space(3))) long long lds_i64;
#pragma unroll
    for (int c = 0; c < WCH; ++c) {
        if (c < nch) {
            const int m = wlo + 64 * c + lane;
            const bool mine = m <= whi;
            const V4<double> ym = nodesL[mine ? m : whi];
            lds_i64 *acn = (lds_i64 *)(accL + (size_t)(mine ? m : whi) * 4);
            const double w0 = P1[c];
            acc_ok &= !mine || __builtin_fabs(w0) < limP;
            if (mine) __hip_atomic_fetch_add(acn + 0, acc_fix(w0, scP), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double d = k == 0 ? ox - ym.x : (k == 1 ? oy - ym.y : oz - ym.z), a = k == 0 ? Sx[c] : (k == 1 ? Sy[c] : Sz[c]);
                const double val = ::fma(d, w0, a);
                acc_ok &= !mine || __builtin_fabs(val) < limR;
                if (mine) __hip_atomic_fetch_add(acn + 1 + k, acc_fix(val, scR), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const double dq = mine ? d * (a + val) : 0.0;
                acc_ok &= __builtin_fabs(dq) < limQn; accQ += acc_fix(dq, scQ);
            }
        }
    }
    wave_lds_sync();
}
}  // namespace tdlo
