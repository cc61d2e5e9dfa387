// Internal device code shared by the blocked Cholesky (mvf_solve.hip) and the pivoted panel factorisation (mvf_minnorm.hip):
// the Newton reciprocal of the pivot steps and the quad-interleaved triangular substitution.
#pragma once
#include "mvf_solve.h"

namespace mvf {

constexpr int NB = CHOL_NB;
constexpr int LDU = CHOL_NB + 1;  // LDS stride of the published columns

__device__ __forceinline__ double rcp_nr2(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = fma(fma(-x, y, 1.0), y, y);
    return fma(fma(-x, y, 1.0), y, y);
}


// ---- triangular substitution x L^T = a with FOUR lanes (one DPP quad) per row (see trsm_panel_kernel) ----
template <int O>
__device__ __forceinline__ double quad_bcast(double v) {
    constexpr int ctrl = O | (O << 2) | (O << 4) | (O << 6);  // quad_perm: every lane of the quad reads lane O
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), ctrl, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), ctrl, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

constexpr int TLQ = 18;           // doubles per (column, lane-of-quad) slot: 16 used, the pad de-phases the four lanes' banks
constexpr int TLC = 4 * TLQ + 2;  // doubles per column c (the odd multiple of 2 keeps the transposing store conflict-free)

// step C with the operands of step C + 1 (their LDS reads) issued before its arithmetic: the chain of a step is
// multiply - two DPP moves - multiply-subtract, and an LDS round trip per step would triple it
template <int C>
__device__ __forceinline__ void trsm_steps(double (&x)[16], const double* __restrict__ ls_rho, const double* __restrict__ rd,
                                           const double2 (&cur)[8], double rdc) {
    constexpr int G = C / 4, O = C % 4;
    double2 nxt[8];
    double rdn = 0.0;
    if constexpr (C + 1 < NB) {
        const double2* ln = reinterpret_cast<const double2*>(ls_rho + (C + 1) * TLC);
#pragma unroll
        for (int h = (C + 1) / 8; h < 8; ++h) nxt[h] = ln[h];
        rdn = rd[C + 1];
    }
    const double xc = quad_bcast<O>(x[G] * rdc);  // meaningful on the owning lane (rho == O), read from it
#pragma unroll
    for (int h = G / 2; h < 8; ++h) {
        x[2 * h] = fma(-xc, cur[h].x, x[2 * h]);
        x[2 * h + 1] = fma(-xc, cur[h].y, x[2 * h + 1]);
    }
    if constexpr (C + 1 < NB) trsm_steps<C + 1>(x, ls_rho, rd, nxt, rdn);
}


// a / l as a * (1 / l) with one correction step: the quotient to within an ulp of the correctly rounded one
__device__ __forceinline__ double div_by(double a, double l, double rl) {
    const double q0 = a * rl;
    return fma(fma(-q0, l, a), rl, q0);
}

}  // namespace mvf
