// Differential-geometry evaluators of the learned field  v(x) = sum_m K(x, c_m) C[m, :]
//
// Reference: dynamo `Jacobian_rkhs_gaussian` + `compute_{acceleration,curvature,curl,torsion,divergence}`; in-tree
// twins spateo/tdr/morphometrics/morphofield_dg/GPVectorField.py:143-190 (Jacobian) and :12-121 (evaluators), which
// run one Python-level `_con_K` call per cell.  Here: ONE fused kernel; a lane owns CPT query points, control points
// and coefficients are broadcast from LDS, v (3) and J (3x3) are accumulated in float64 in registers and every
// requested quantity is derived in registers, so HBM traffic is 16 B in + the requested outputs per query.
//   J[f][i] = -2 beta sum_m K_m C[m, f] (x - c_m)_i                      (GPVectorField.py:176,190 with pre_scale = 1)
//   div = tr J;  curl = [J21 - J12, J02 - J20, J10 - J01];  a = J v
//   curvature (formula 2) = (a (v.v) - v (v.a)) / |v|^4;  torsion = v (a . J a) / (|v|^2 |a|^2)
#include "mvf_common.h"

namespace mvf {

constexpr int EVAL_CHUNK = 512;

// v_out = alpha * (K @ C) + A q + b  (q = the UNSCALED, centred query point as passed in x4);  J_out = jmul * J.
// Identity for the sparsevfc field; the affine part carries the GP variant's norm_dict scaling and rigid transform
// (spateo/tdr/morphometrics/morphofield/gaussian_process.py:102-127, GPVectorField.py:158-159,190).
struct EvalAffine {
    double alpha[3], jmul;  // alpha per output component (the GP variant's scale_fixed may be per axis)
    double A[9];
    double b[3];
};

template <typename T, int CPT>
__global__ __launch_bounds__(256) void eval_kernel(const T* __restrict__ x4, int64_t n, const T* __restrict__ ctrl4,
                                                   int64_t m, T s, double jscale /* -2 beta / s */, EvalAffine af,
                                                   const double* __restrict__ C, int flags, double* __restrict__ v_out,
                                                   double* __restrict__ jac, double* __restrict__ div,
                                                   double* __restrict__ curl, double* __restrict__ acc_out,
                                                   double* __restrict__ curv, double* __restrict__ tors,
                                                   double* __restrict__ jdet) {
    using V4T = typename Vec4<T>::type;
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[EVAL_CHUNK * (sizeof(V4T) + 4 * sizeof(double))];
    V4T* sc = reinterpret_cast<V4T*>(smem_raw);
    double4* sC = reinterpret_cast<double4*>(smem_raw + EVAL_CHUNK * sizeof(V4T));

    const int64_t base = ((int64_t)blockIdx.x * 256) * CPT + threadIdx.x;
    T px[CPT], py[CPT], pz[CPT];
    double q0[CPT], q1[CPT], q2[CPT];
    double v[CPT][3], J[CPT][3][3];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const int64_t i = base + (int64_t)c * 256;
        V4T xv = (i < n) ? reinterpret_cast<const V4T*>(x4)[i] : V4T{0, 0, 0, 0};
        q0[c] = (double)xv.x, q1[c] = (double)xv.y, q2[c] = (double)xv.z;
        px[c] = xv.x * s, py[c] = xv.y * s, pz[c] = xv.z * s;
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            v[c][f] = 0.0;
#pragma unroll
            for (int i2 = 0; i2 < 3; ++i2) J[c][f][i2] = 0.0;
        }
    }

    for (int64_t m0 = 0; m0 < m; m0 += EVAL_CHUNK) {
        const int mc = (int)min((int64_t)EVAL_CHUNK, m - m0);
        __syncthreads();
        for (int j = threadIdx.x; j < EVAL_CHUNK; j += 256) {
            if (j < mc) {
                V4T cv = reinterpret_cast<const V4T*>(ctrl4)[m0 + j];
                sc[j] = V4T{cv.x * s, cv.y * s, cv.z * s, 0};
                const double* cp = C + (m0 + j) * 3;
                sC[j] = double4{cp[0], cp[1], cp[2], 0.0};
            } else {
                sc[j] = V4T{0, 0, 0, 0};
                sC[j] = double4{0.0, 0.0, 0.0, 0.0};
            }
        }
        __syncthreads();
        const int mc_pad = (mc + 1) & ~1;
#pragma unroll 2
        for (int j = 0; j < mc_pad; ++j) {
            const V4T cv = sc[j];
            const double4 cc = sC[j];
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                const T dx = px[c] - cv.x, dy = py[c] - cv.y, dz = pz[c] - cv.z;
                const double k = (double)kernel_value(px[c], py[c], pz[c], cv.x, cv.y, cv.z);
                const double t0 = k * cc.x, t1 = k * cc.y, t2 = k * cc.z;
                const double ddx = (double)dx, ddy = (double)dy, ddz = (double)dz;
                v[c][0] += t0, v[c][1] += t1, v[c][2] += t2;
                J[c][0][0] = fma(t0, ddx, J[c][0][0]), J[c][0][1] = fma(t0, ddy, J[c][0][1]), J[c][0][2] = fma(t0, ddz, J[c][0][2]);
                J[c][1][0] = fma(t1, ddx, J[c][1][0]), J[c][1][1] = fma(t1, ddy, J[c][1][1]), J[c][1][2] = fma(t1, ddz, J[c][1][2]);
                J[c][2][0] = fma(t2, ddx, J[c][2][0]), J[c][2][1] = fma(t2, ddy, J[c][2][1]), J[c][2][2] = fma(t2, ddz, J[c][2][2]);
            }
        }
    }

#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const int64_t q = base + (int64_t)c * 256;
        if (q >= n) continue;
        double Jm[3][3];
#pragma unroll
        for (int f = 0; f < 3; ++f)
#pragma unroll
            for (int i = 0; i < 3; ++i) Jm[f][i] = J[c][f][i] * jscale * af.jmul;
        const double v0 = af.alpha[0] * v[c][0] + af.A[0] * q0[c] + af.A[1] * q1[c] + af.A[2] * q2[c] + af.b[0];
        const double v1 = af.alpha[1] * v[c][1] + af.A[3] * q0[c] + af.A[4] * q1[c] + af.A[5] * q2[c] + af.b[1];
        const double v2 = af.alpha[2] * v[c][2] + af.A[6] * q0[c] + af.A[7] * q1[c] + af.A[8] * q2[c] + af.b[2];
        if (flags & MVF_EVAL_V) {
            v_out[q * 3 + 0] = v0, v_out[q * 3 + 1] = v1, v_out[q * 3 + 2] = v2;
        }
        if (flags & MVF_EVAL_JAC) {
#pragma unroll
            for (int f = 0; f < 3; ++f)
#pragma unroll
                for (int i = 0; i < 3; ++i) jac[(int64_t)(f * 3 + i) * n + q] = Jm[f][i];
        }
        if (flags & MVF_EVAL_DIV) div[q] = Jm[0][0] + Jm[1][1] + Jm[2][2];
        if (flags & MVF_EVAL_JDET)
            jdet[q] = Jm[0][0] * (Jm[1][1] * Jm[2][2] - Jm[1][2] * Jm[2][1]) -
                      Jm[0][1] * (Jm[1][0] * Jm[2][2] - Jm[1][2] * Jm[2][0]) +
                      Jm[0][2] * (Jm[1][0] * Jm[2][1] - Jm[1][1] * Jm[2][0]);
        if (flags & MVF_EVAL_CURL) {
            curl[q * 3 + 0] = Jm[2][1] - Jm[1][2];
            curl[q * 3 + 1] = Jm[0][2] - Jm[2][0];
            curl[q * 3 + 2] = Jm[1][0] - Jm[0][1];
        }
        if (flags & (MVF_EVAL_ACC | MVF_EVAL_CURV | MVF_EVAL_TORS)) {
            const double a0 = Jm[0][0] * v0 + Jm[0][1] * v1 + Jm[0][2] * v2;
            const double a1 = Jm[1][0] * v0 + Jm[1][1] * v1 + Jm[1][2] * v2;
            const double a2 = Jm[2][0] * v0 + Jm[2][1] * v1 + Jm[2][2] * v2;
            if (flags & MVF_EVAL_ACC) {
                acc_out[q * 3 + 0] = a0, acc_out[q * 3 + 1] = a1, acc_out[q * 3 + 2] = a2;
            }
            const double vv = v0 * v0 + v1 * v1 + v2 * v2;
            if (flags & MVF_EVAL_CURV) {
                const double va = v0 * a0 + v1 * a1 + v2 * a2;
                const double nv = sqrt(vv);
                const double den = (nv * nv) * (nv * nv);  // ||v||^4 as norm(v)**4
                curv[q * 3 + 0] = (a0 * vv - v0 * va) / den;
                curv[q * 3 + 1] = (a1 * vv - v1 * va) / den;
                curv[q * 3 + 2] = (a2 * vv - v2 * va) / den;
            }
            if (flags & MVF_EVAL_TORS) {
                const double Ja0 = Jm[0][0] * a0 + Jm[0][1] * a1 + Jm[0][2] * a2;
                const double Ja1 = Jm[1][0] * a0 + Jm[1][1] * a1 + Jm[1][2] * a2;
                const double Ja2 = Jm[2][0] * a0 + Jm[2][1] * a1 + Jm[2][2] * a2;
                const double aJa = a0 * Ja0 + a1 * Ja1 + a2 * Ja2;
                const double aa = a0 * a0 + a1 * a1 + a2 * a2;
                const double den = vv * aa;  // ||v a^T||_F^2
                tors[q * 3 + 0] = v0 * aJa / den;
                tors[q * 3 + 1] = v1 * aJa / den;
                tors[q * 3 + 2] = v2 * aJa / den;
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------
// Trajectory integration  dx/dt = v(x)  (morphopath; reference: spateo/tdr/morphometrics/morphofield/trajectory.py:11-117,
// which hands the field to dynamo's `fate`).  One launch integrates every start point through ALL time steps with
// classical RK4: a lane owns one trajectory (position in float64 registers), the control points and coefficients are
// staged in LDS once (or per evaluation in chunks when they do not fit), and the n_out sampled positions are written
// as traj[cell][t][3].  No host round trip and no intermediate tensor per step.
// ----------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void integrate_kernel(const T* __restrict__ x4, int64_t n, const T* __restrict__ ctrl4,
                                                        int64_t m, T s, EvalAffine af, const double* __restrict__ C,
                                                        double dt, int substeps, int n_out, int chunk,
                                                        double* __restrict__ traj) {
    using V4T = typename Vec4<T>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];
    V4T* sc = reinterpret_cast<V4T*>(smem_dyn);
    double4* sC = reinterpret_cast<double4*>(smem_dyn + (size_t)chunk * sizeof(V4T));
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n;
    const int nchunks = (int)((m + chunk - 1) / chunk);

    auto stage = [&](int64_t m0) {
        const int mc = (int)min((int64_t)chunk, m - m0);
        for (int j = threadIdx.x; j < chunk; j += 256) {
            if (j < mc) {
                const V4T cv = reinterpret_cast<const V4T*>(ctrl4)[m0 + j];
                sc[j] = V4T{cv.x * s, cv.y * s, cv.z * s, 0};
                const double* cp = C + (m0 + j) * 3;
                sC[j] = double4{cp[0], cp[1], cp[2], 0.0};
            } else {
                sc[j] = V4T{0, 0, 0, 0};
                sC[j] = double4{0.0, 0.0, 0.0, 0.0};
            }
        }
    };
    // v(q) for this lane's position q (block-synchronous: every lane takes the same path through the barriers)
    auto field = [&](double q0, double q1, double q2, double& v0, double& v1, double& v2) {
        const T px = (T)q0 * s, py = (T)q1 * s, pz = (T)q2 * s;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
        for (int cidx = 0; cidx < nchunks; ++cidx) {
            if (nchunks > 1) {
                __syncthreads();
                stage((int64_t)cidx * chunk);
                __syncthreads();
            }
            const int mc = (int)min((int64_t)chunk, m - (int64_t)cidx * chunk);
#pragma unroll 4
            for (int j = 0; j < mc; ++j) {
                const V4T cv = sc[j];
                const double4 cc = sC[j];
                const double k = (double)kernel_value(px, py, pz, cv.x, cv.y, cv.z);
                a0 = fma(k, cc.x, a0), a1 = fma(k, cc.y, a1), a2 = fma(k, cc.z, a2);
            }
        }
        v0 = af.alpha[0] * a0 + af.A[0] * q0 + af.A[1] * q1 + af.A[2] * q2 + af.b[0];
        v1 = af.alpha[1] * a1 + af.A[3] * q0 + af.A[4] * q1 + af.A[5] * q2 + af.b[1];
        v2 = af.alpha[2] * a2 + af.A[6] * q0 + af.A[7] * q1 + af.A[8] * q2 + af.b[2];
    };

    if (nchunks == 1) {
        stage(0);
        __syncthreads();
    }
    double x0 = 0.0, x1 = 0.0, x2 = 0.0;
    if (live) {
        const V4T xv = reinterpret_cast<const V4T*>(x4)[i];
        x0 = (double)xv.x, x1 = (double)xv.y, x2 = (double)xv.z;
        double* o = traj + (size_t)i * n_out * 3;
        o[0] = x0, o[1] = x1, o[2] = x2;
    }
    const double h = dt / substeps;
    for (int t = 1; t < n_out; ++t) {
        for (int ss = 0; ss < substeps; ++ss) {
            double k1x, k1y, k1z, k2x, k2y, k2z, k3x, k3y, k3z, k4x, k4y, k4z;
            field(x0, x1, x2, k1x, k1y, k1z);
            field(x0 + 0.5 * h * k1x, x1 + 0.5 * h * k1y, x2 + 0.5 * h * k1z, k2x, k2y, k2z);
            field(x0 + 0.5 * h * k2x, x1 + 0.5 * h * k2y, x2 + 0.5 * h * k2z, k3x, k3y, k3z);
            field(x0 + h * k3x, x1 + h * k3y, x2 + h * k3z, k4x, k4y, k4z);
            x0 += h / 6.0 * (k1x + 2.0 * k2x + 2.0 * k3x + k4x);
            x1 += h / 6.0 * (k1y + 2.0 * k2y + 2.0 * k3y + k4y);
            x2 += h / 6.0 * (k1z + 2.0 * k2z + 2.0 * k3z + k4z);
        }
        if (live) {
            double* o = traj + ((size_t)i * n_out + t) * 3;
            o[0] = x0, o[1] = x1, o[2] = x2;
        }
    }
}

}  // namespace mvf

using namespace mvf;

extern "C" int mvf_eval_affine(const void* x4, int64_t n, const void* ctrl4, int64_t m, double beta, const double* C,
                               const double* affine /* host: alpha[3], jmul, A[9] row-major, b[3]; NULL = identity */,
                               int flags, double* v, double* jac, double* div, double* curl, double* acc, double* curv,
                               double* tors, double* jdet, mvf_dtype dtype, void* stream) {
    MVF_REQUIRE(n >= 0 && m >= 0, "mvf_eval: bad shape");
    MVF_REQUIRE(beta > 0.0 && std::isfinite(beta), "mvf_eval: beta must be finite and > 0");
    if (n == 0) return 0;
    MVF_REQUIRE(x4 && (m == 0 || (ctrl4 && C)), "mvf_eval: null input");
    MVF_REQUIRE(!(flags & MVF_EVAL_V) || v, "mvf_eval: v requested but null");
    MVF_REQUIRE(!(flags & MVF_EVAL_JAC) || jac, "mvf_eval: jac requested but null");
    MVF_REQUIRE(!(flags & MVF_EVAL_DIV) || div, "mvf_eval: div requested but null");
    MVF_REQUIRE(!(flags & MVF_EVAL_CURL) || curl, "mvf_eval: curl requested but null");
    MVF_REQUIRE(!(flags & MVF_EVAL_ACC) || acc, "mvf_eval: acc requested but null");
    MVF_REQUIRE(!(flags & MVF_EVAL_CURV) || curv, "mvf_eval: curv requested but null");
    MVF_REQUIRE(!(flags & MVF_EVAL_TORS) || tors, "mvf_eval: tors requested but null");
    MVF_REQUIRE(!(flags & MVF_EVAL_JDET) || jdet, "mvf_eval: jdet requested but null");
    EvalAffine af;
    af.alpha[0] = af.alpha[1] = af.alpha[2] = 1.0, af.jmul = 1.0;
    for (int i = 0; i < 9; ++i) af.A[i] = 0.0;
    for (int i = 0; i < 3; ++i) af.b[i] = 0.0;
    if (affine) {
        for (int i = 0; i < 3; ++i) af.alpha[i] = affine[i];
        af.jmul = affine[3];
        for (int i = 0; i < 9; ++i) af.A[i] = affine[4 + i];
        for (int i = 0; i < 3; ++i) af.b[i] = affine[13 + i];
    }
    hipStream_t st = (hipStream_t)stream;
    const double s = std::sqrt(beta * LOG2E);
    // (x - c) = (scaled difference) / s, with s as the kernel rounds it
    const double jscale = -2.0 * beta / ((dtype == MVF_F32) ? (double)(float)s : s);
    // Two query points per lane amortise the LDS broadcast of a control point; a grid-sized launch (64^3 = 262 144 queries:
    // 512 workgroups = 2 per CU, 2 waves per SIMD) then leaves the float64 pipe short of independent work between its
    // dependent chains (0.30 of the float64 VALU peak) - below one million queries one point per lane
    // and twice the workgroups.
    const bool one = n < (int64_t)(1 << 20);
    dim3 grid((unsigned)cdiv(n, 256 * (one ? 1 : 2)));
    if (dtype == MVF_F32) {
        if (one)
            hipLaunchKernelGGL((eval_kernel<float, 1>), grid, dim3(256), 0, st, (const float*)x4, n, (const float*)ctrl4,
                               m, (float)s, jscale, af, C, flags, v, jac, div, curl, acc, curv, tors, jdet);
        else
            hipLaunchKernelGGL((eval_kernel<float, 2>), grid, dim3(256), 0, st, (const float*)x4, n, (const float*)ctrl4,
                               m, (float)s, jscale, af, C, flags, v, jac, div, curl, acc, curv, tors, jdet);
    } else if (dtype == MVF_F64) {
        if (one)
            hipLaunchKernelGGL((eval_kernel<double, 1>), grid, dim3(256), 0, st, (const double*)x4, n,
                               (const double*)ctrl4, m, s, jscale, af, C, flags, v, jac, div, curl, acc, curv, tors, jdet);
        else
            hipLaunchKernelGGL((eval_kernel<double, 2>), grid, dim3(256), 0, st, (const double*)x4, n,
                               (const double*)ctrl4, m, s, jscale, af, C, flags, v, jac, div, curl, acc, curv, tors, jdet);
    } else {
        return set_error("mvf_eval: bad dtype %d", (int)dtype);
    }
    MVF_LAUNCH_CHECK();
    return 0;
}

extern "C" int mvf_eval(const void* x4, int64_t n, const void* ctrl4, int64_t m, double beta, const double* C,
                        int flags, double* v, double* jac, double* div, double* curl, double* acc, double* curv,
                        double* tors, double* jdet, mvf_dtype dtype, void* stream) {
    return mvf_eval_affine(x4, n, ctrl4, m, beta, C, nullptr, flags, v, jac, div, curl, acc, curv, tors, jdet, dtype,
                           stream);
}

extern "C" int mvf_integrate(const void* x4, int64_t n, const void* ctrl4, int64_t m, double beta, const double* C,
                             const double* affine, double dt, int substeps, int n_out, double* traj, mvf_dtype dtype,
                             void* stream) {
    MVF_REQUIRE(n >= 0 && m >= 0 && n_out >= 1 && substeps >= 1, "mvf_integrate: bad shape");
    MVF_REQUIRE(beta > 0.0 && std::isfinite(beta) && std::isfinite(dt), "mvf_integrate: bad beta / dt");
    if (n == 0) return 0;
    MVF_REQUIRE(x4 && traj && (m == 0 || (ctrl4 && C)), "mvf_integrate: null pointer");
    EvalAffine af;
    af.alpha[0] = af.alpha[1] = af.alpha[2] = 1.0, af.jmul = 1.0;
    for (int i = 0; i < 9; ++i) af.A[i] = 0.0;
    for (int i = 0; i < 3; ++i) af.b[i] = 0.0;
    if (affine) {
        for (int i = 0; i < 3; ++i) af.alpha[i] = affine[i];
        af.jmul = affine[3];
        for (int i = 0; i < 9; ++i) af.A[i] = affine[4 + i];
        for (int i = 0; i < 3; ++i) af.b[i] = affine[13 + i];
    }
    hipStream_t st = (hipStream_t)stream;
    const double s = std::sqrt(beta * LOG2E);
    const size_t per = (dtype == MVF_F32 ? 16 : 32) + 32;  // bytes of LDS per staged control point
    const int cap = (int)((144 * 1024) / per);               // leave headroom below the 160 KiB of a CU
    const int chunk = (int)std::max<int64_t>(1, std::min<int64_t>(m, cap));
    const size_t lds = (size_t)chunk * per;
    dim3 grid((unsigned)cdiv(n, 256));
    if (dtype == MVF_F32) {
        MVF_CHECK_HIP(hipFuncSetAttribute((const void*)integrate_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)lds));
        hipLaunchKernelGGL(integrate_kernel<float>, grid, dim3(256), lds, st, (const float*)x4, n, (const float*)ctrl4, m,
                           (float)s, af, C, dt, substeps, n_out, chunk, traj);
    } else if (dtype == MVF_F64) {
        MVF_CHECK_HIP(hipFuncSetAttribute((const void*)integrate_kernel<double>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(integrate_kernel<double>, grid, dim3(256), lds, st, (const double*)x4, n,
                           (const double*)ctrl4, m, s, af, C, dt, substeps, n_out, chunk, traj);
    } else {
        return set_error("mvf_integrate: bad dtype %d", (int)dtype);
    }
    MVF_LAUNCH_CHECK();
    return 0;
}
