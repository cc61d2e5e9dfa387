// M-step assembly:  G = U^T diag(P) U  (m x m)  and  R = U^T diag(P) Y  (m x 3),  U = con_K(x, ctrl, beta) (n x m)
//
// Reference: `UP = U.T * repmat(P.T, M, 1); lhs = UP.dot(U) + ...; rhs = UP.dot(Y)` of dynamo SparseVFC
// (call sites spateo/tdr/morphometrics/morphofield/sparsevfc.py:189-198).
//
// Design (MI355X first, not a GEMM on a materialised U):
//   * G is an n-long reduction SYRK; at m >= 500 it is MFMA-bound (2 n m^2 flops vs 16 n bytes of input).  The kernel
//     values are either regenerated per lane (gram_f64acc_kernel: the lane keeps ITS control points in registers and
//     reads the step's cells with one broadcast ds_read_b128 from a 4 KiB LDS stage; 6 VALU + 1 v_exp per operand) or
//     streamed from a cache in MFMA-operand layout built once per fit (gram_cached_kernel, the default: U is constant
//     across EM iterations).  A[i][k] = P_n K(x_n, c_i), B[k][j] = K(x_n, c_j), n = the cell k of the step.
//   * Only tile pairs ti <= tj of the symmetric G are computed (algorithmic flops n m (m+1)).
//   * Precision: v_mfma_f64_16x16x4_f64 accumulation in BOTH modes - float32 mode generates / stores the kernel values
//     in float32 and widens them (a product of two float32 values is exact in float64: G is the exact Gram matrix of
//     the float32 kernel values); per-slice float64 partial tiles go to a workspace and are summed in a fixed order
//     -> deterministic.  (An all-float32 MFMA arm, 256-cell chains folded into float64, was measured in round 1 -
//     87 TF, ~1e-9 relative noise in G that the solve amplifies to percent level - and removed in round 2.)
//   * Work decomposition: job = (tile pair, cell slice); blockIdx = slice * npairs + pair so that concurrently
//     resident workgroups stream the same cell slice (L2 / MALL hits on the only global input).
#include "mvf_common.h"
#include <type_traits>

namespace mvf {

constexpr int GT = 128;      // Gram tile edge per workgroup (4 waves, 64 x 64 per wave)
constexpr int GCHUNK = 256;  // cells per LDS stage

typedef double f64x4 __attribute__((ext_vector_type(4)));

struct GramPlan {
    int nt = 0, npairs = 0;
    int njobs = 0, edge2 = 0;              // cached kernel: jobs per slice (edge2: narrow last-column tiles run two per job)
    int64_t slice_len = 0, nslices = 0;    // Gram jobs
    int64_t phase_slices = 0, nphases = 1; // slices per launch; the partial-tile buffer holds one phase
    int64_t rslice_len = 0, rslices = 0;   // rhs jobs
    int rcolblocks = 0;
    size_t gram_bytes = 0, rhs_bytes = 0;
};

constexpr int RHS_CPT = 2;                  // control points per lane in the rhs kernel
constexpr int RHS_COLS = 256 * RHS_CPT;     // control points per rhs workgroup

// Resident workgroup slots of the Gram kernels: 2 workgroups (8 waves) per CU (register-limited), 256 CUs on MI355X.
static int gram_slots() { return 2 * device_cu_count(); }

static GramPlan make_plan(int64_t n, int64_t m, mvf_dtype dtype) {
    GramPlan p;
    p.nt = (int)cdiv(m, GT);
    p.npairs = p.nt * (p.nt + 1) / 2;
    // float64 cached kernel: when the last tile column has <= 64 live control points (m = 3000: 56) its nt tiles run TWO per
    // job (4 row blocks x 4 column blocks per wave = the 16 MFMAs per k-step of every other job), see gram_cached_kernel
    p.edge2 = (dtype == MVF_F64 && p.nt >= 2 && m - (int64_t)(p.nt - 1) * GT <= GT / 2) ? 1 : 0;
    p.njobs = p.edge2 ? p.npairs - p.nt + (p.nt + 1) / 2 : p.npairs;
    // Jobs = tile pairs x cell slices.  All jobs cost the same, so the launch runs in ceil(jobs / slots) rounds of
    // (slice_len + OVERHEAD) cell-times, OVERHEAD = the 128 KB partial tile a job writes and the reduce kernel reads
    // back, expressed in cells of MFMA work (measured ~150: at 50 k cells x 500 control points 1024-cell slices beat
    // 256-cell ones by 15 % although both fill the chip).  Pick the slice count with the smallest modelled time.  (At
    // 1 M cells x 300 pairs the naive 4 slices = 1200 jobs on 512 slots lost 22 % to the tail.)
    // SHORT slices also keep the workgroups that stream the same panels together: the ~24 tile pairs that share a
    // panel start a slice in step and drift apart while they run, and once they are further apart than the 256 MB MALL
    // holds (10 - 20 k cells of all panels) every one of them fetches the panel from HBM again.  Measured at 8 M cells x
    // 3000 (profiles/r02_gram_slice_len.md): float64 51.1 TF with 276 k-cell slices, 53.0 / 54.3 / 54.9 / 56.4 TF at
    // 131 k / 65 k / 32 k / 16 k, 61.1 at 8 k, 60.8 at 4 k; float32 65.6 -> 66.2 / 66.7 / 67.1 / 67.5 and 67.8 at 8 k.  So slices
    // are capped at 8 k cells; the partial-tile buffer stays within 10 % of the size of the kernel-value cache (at most
    // 20 GB; at least 3 GB) and is reused by several launches when the capped slices need more.
    const int64_t max_chunks = cdiv(n, GCHUNK);
    const double cache_bytes = (double)n * (double)m * (dtype == MVF_F64 ? 8.0 : 4.0);
    const double budget_bytes = std::min(20e9, std::max(3e9, 0.1 * cache_bytes));  // (1 / 2 / 4 phases A/B-ed in round 5: no difference)
    const int64_t budget_jobs = std::max<int64_t>(p.npairs, (int64_t)(budget_bytes / (GT * GT * sizeof(double))));
    const int64_t fit = std::max<int64_t>(1, budget_jobs / p.npairs);  // slices whose partial tiles fit the buffer
    constexpr int64_t SL_CAP = 8192;
    const int64_t s_cap = std::min<int64_t>(max_chunks, std::max<int64_t>(1, cdiv(n, SL_CAP)));
    const double slots = (double)gram_slots();
    constexpr double OVERHEAD = 150.0;
    auto model = [&](int64_t s, int64_t& ns) {
        const int64_t sl = cdiv(cdiv(n, s), GCHUNK) * GCHUNK;
        ns = cdiv(n, sl);
        return std::ceil((double)ns * p.npairs / slots) * ((double)sl + OVERHEAD);
    };
    int64_t best_s = s_cap, ns = 0;
    const int64_t s_fit = std::min<int64_t>(fit, max_chunks);
    if (s_cap <= s_fit) {
        // one launch: the slice count in [s_cap, s_fit] with the smallest modelled time; among plans within 0.5 % of it the
        // one with the most rounds (many short rounds even out per-XCD speed differences better than a few long ones)
        double best_t = 1e300;
        for (int64_t s = s_cap; s <= s_fit; ++s) best_t = std::min(best_t, model(s, ns));
        for (int64_t s = s_fit; s >= s_cap; --s)
            if (model(s, ns) <= best_t * 1.005) {
                best_s = ns;
                break;
            }
    } else {
        // the capped slices need more partial tiles than the buffer holds: several launches ("phases") reuse it, each
        // folded into G before the next one starts (8 M cells x 3000: 977 slices of 8 k cells in 2 (float64) / 4 (float32)
        // phases).  Searching shorter slices / more phases by a tail model measured WORSE (56 instead of 61 TF in float64):
        // every phase boundary drains the chip.
        // Slices up to 10 % longer than the cap are accepted if that saves a launch (977 = 4 x 244 + 1 slices would
        // otherwise take a fifth phase for one slice).
        int64_t s = s_cap;
        const int64_t nph0 = cdiv(s_cap, fit);
        if (nph0 > 1 && (nph0 - 1) * fit * 10 >= s_cap * 9) s = (nph0 - 1) * fit;
        model(s, ns);
        best_s = ns;
    }
    int64_t sl = cdiv(cdiv(n, best_s), GCHUNK) * GCHUNK;
    const int64_t slice_knob = debug_opt(DBG_SLICE_LEN);  // developer option (probes, tests of the multi-phase plan)
    if (slice_knob > 0) sl = std::max<int64_t>(GCHUNK, slice_knob / GCHUNK * GCHUNK);
    p.slice_len = sl;
    p.nslices = std::max<int64_t>(1, cdiv(n, sl));
    p.nphases = cdiv(p.nslices, fit);
    p.phase_slices = cdiv(p.nslices, p.nphases);
    p.gram_bytes = (size_t)p.phase_slices * p.npairs * GT * GT * sizeof(double);
    p.rcolblocks = (int)cdiv(m, RHS_COLS);
    int64_t want_r = std::max<int64_t>(1, cdiv(1024, std::max(1, p.rcolblocks)));
    int64_t rsl = cdiv(cdiv(n, want_r), GCHUNK) * GCHUNK;
    rsl = std::max<int64_t>(rsl, GCHUNK);
    p.rslice_len = rsl;
    p.rslices = std::max<int64_t>(1, cdiv(n, rsl));
    p.rhs_bytes = (size_t)p.rslices * m * 4 * sizeof(double);
    return p;
}

__device__ __forceinline__ void decode_pair(int pair, int nt, int& ti, int& tj) {
    int p = pair, t = 0;
    while (p >= nt - t) {
        p -= nt - t;
        ++t;
    }
    ti = t;
    tj = t + p;
}

// ----------------------------------------------------------------------------------------------------------------
// float64-ACCUMULATE MFMA Gram kernel (v_mfma_f64_16x16x4_f64: A[i = l&15][k = l>>4], B[k = l>>4][j = l&15];
// C/D: col = l & 15, row = (l >> 4) + 4 * reg).  TIn = double: the float64 mode.  TIn = float: the default float32
// mode - operands are generated in float32 (6 VALU + v_exp_f32) and widened; a product of two float32 values is exact
// in float64, so G is the EXACT Gram matrix of the float32-rounded kernel values (error ~1e-16 sqrt(n)), whereas an
// all-float32 MFMA kernel (v_mfma_f32_32x32x2_f32, measured in round 1 at 87 TF and removed) leaves ~1e-9 relative noise
// that the ill-conditioned solve amplifies to percent level (DESIGN.md "Why the float32 mode accumulates in float64").
// ----------------------------------------------------------------------------------------------------------------
template <typename TIn>
__global__ __launch_bounds__(256, 2) void gram_f64acc_kernel(const typename Vec4<TIn>::type* __restrict__ x4,
                                                             const TIn* __restrict__ P, int64_t n,
                                                             const typename Vec4<TIn>::type* __restrict__ ctrl4,
                                                             int64_t m, TIn s, int nt, int npairs, int64_t slice_len,
                                                             int64_t slice0, double* __restrict__ partial) {
    using V4 = typename Vec4<TIn>::type;
    __shared__ V4 cells[2][GCHUNK];
    const TIn PAD = sizeof(TIn) == 4 ? (TIn)1.0e18f : (TIn)1.0e150;  // padded control points sit "at infinity": K == 0 exactly

    const int pair = blockIdx.x % npairs;
    const int64_t slice = blockIdx.x / npairs;  // within this launch's phase; slice0 + slice = the cell slice
    int ti, tj;
    decode_pair(pair, nt, ti, tj);
    const int64_t n0 = (slice0 + slice) * slice_len;
    const int64_t n1 = min(n, n0 + slice_len);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;

    TIn ax[4], ay[4], az[4], bx[4], by[4], bz[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int64_t ia = (int64_t)ti * GT + wi * 64 + a * 16 + (lane & 15);
        const int64_t ib = (int64_t)tj * GT + wj * 64 + a * 16 + (lane & 15);
        if (ia < m) {
            const V4 c = ctrl4[ia];
            ax[a] = c.x * s, ay[a] = c.y * s, az[a] = c.z * s;
        } else {
            ax[a] = ay[a] = az[a] = PAD;
        }
        if (ib < m) {
            const V4 c = ctrl4[ib];
            bx[a] = c.x * s, by[a] = c.y * s, bz[a] = c.z * s;
        } else {
            bx[a] = by[a] = bz[a] = PAD;
        }
    }

    f64x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};

    auto load_cell = [&](int64_t i) -> V4 {
        if (i < n1) {
            const V4 xv = x4[i];
            return V4{xv.x * s, xv.y * s, xv.z * s, P[i]};
        }
        return V4{0, 0, 0, 0};
    };

    const int nchunks = (int)((n1 - n0 + GCHUNK - 1) / GCHUNK);
    V4 stage = load_cell(n0 + tid);
    cells[0][tid] = stage;
    const int quarter = lane >> 4;

    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();
        if (c + 1 < nchunks) stage = load_cell(n0 + (int64_t)(c + 1) * GCHUNK + tid);
        const V4* cb = cells[c & 1];
#pragma unroll 2
        for (int st = 0; st < GCHUNK / 4; ++st) {
            const V4 cell = cb[4 * st + quarter];
            double fa[4], fb[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                // P K: exact product in float64
                fa[a] = (double)kernel_value(cell.x, cell.y, cell.z, ax[a], ay[a], az[a]) * (double)cell.w;
                fb[a] = (double)kernel_value(cell.x, cell.y, cell.z, bx[a], by[a], bz[a]);
            }
#ifdef MVF_PROBE_SETPRIO_RECOMPUTE
            __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, 0);
#ifdef MVF_PROBE_SETPRIO_RECOMPUTE
            __builtin_amdgcn_s_setprio(0);
#endif
        }
        if (c + 1 < nchunks) cells[(c + 1) & 1][tid] = stage;
    }

    double* out = partial + ((size_t)slice * npairs + pair) * (size_t)(GT * GT);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wi * 64 + a * 16 + quarter + 4 * r;
                const int col = wj * 64 + b * 16 + (lane & 15);
                out[row * GT + col] = acc[a][b][r];
            }
}

// ----------------------------------------------------------------------------------------------------------------
// Cached-U variant (both cell dtypes).  U = con_K(x, ctrl) does not change across EM iterations - only P does - so when
// HBM has room (4 n M bytes: 96 GB at 8 M x 3000) the float32 kernel values are materialised ONCE per fit, in the
// blocked layout  Ublk[cb][cell][16]  (cb = control point / 16), which is exactly the v_mfma_f64_16x16x4 operand
// shape: for one MFMA block and one k-step the 64 lanes read 4 cells x 16 control points = 256 contiguous bytes.
// Measured motivation (tools/mfma_peak2.hip): VALU work is NOT hidden behind f64 MFMAs on gfx950 - a pure f64 MFMA
// stream runs 77.8 TF, the same stream with 8 VALU ops per MFMA 59.4 TF - so the exp/convert work per operand of the
// recompute kernel is additive.  Here the loop holds only 9 coalesced dword loads, 8 converts and 4 f64 multiplies
// per 16 MFMAs; the values are bit-identical to kernel_value() in every other kernel.
// ----------------------------------------------------------------------------------------------------------------
constexpr int UB = 16;  // control points per cached block

template <typename T>
__global__ __launch_bounds__(256) void ublk_build_kernel(const typename Vec4<T>::type* __restrict__ x4, int64_t n,
                                                         int64_t n_pad, const typename Vec4<T>::type* __restrict__ ctrl4,
                                                         int64_t m, int64_t m_pad, T s, int cb_per_block,
                                                         T* __restrict__ ublk) {
    using V4 = typename Vec4<T>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char ublk_smem[];
    V4* sctrl = reinterpret_cast<V4*>(ublk_smem);  // cb_per_block * 16 scaled control points
    const int64_t cb0 = (int64_t)blockIdx.y * cb_per_block;
    const int ncb = (int)min((int64_t)cb_per_block, m_pad / UB - cb0);
    for (int j = threadIdx.x; j < ncb * UB; j += 256) {
        const int64_t c = cb0 * UB + j;
        if (c < m) {
            const V4 cv = ctrl4[c];
            sctrl[j] = V4{cv.x * s, cv.y * s, cv.z * s, 1};
        } else {
            sctrl[j] = V4{0, 0, 0, 0};  // w = 0 marks a padded control point
        }
    }
    __syncthreads();
    // Store pattern (round 6): a wave owns 64 consecutive cells; per cached block that is one contiguous tile of 64 cells x 16
    // control points.  Lane l of store instruction q writes the 16 bytes at tile offset (64 q + l) * 16 - every instruction
    // covers 1 KB of whole 128-byte lines (the first version gave each lane ONE cell, so an instruction touched a quarter
    // (float32) / an eighth (float64) of 64 different lines: 1.2 / 0.6 TB/s at 8 M x 3000, 84 / 324 ms per fit).  A lane then
    // serves Q different cells and PER consecutive control points of every block; same values, same places.
    constexpr int PER = 16 / sizeof(T);  // elements per 16-byte store
    constexpr int Q = UB / PER;          // store instructions per block and wave (4 float32 / 8 float64)
    constexpr int LPC = UB / PER;        // lanes per cell inside one instruction
    typedef T vec_t __attribute__((ext_vector_type(PER)));
    const int lane = threadIdx.x & 63;
    const int64_t wbase = (int64_t)blockIdx.x * 256 + (threadIdx.x & ~63);  // first cell of this wave (n_pad is a multiple of 256)
    if (wbase >= n_pad) return;
    const int j0 = (lane % LPC) * PER;   // this lane's control points inside a block: j0 .. j0 + PER - 1
    T px[Q], py[Q], pz[Q];
    bool live[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int64_t i = wbase + (64 / LPC) * q + lane / LPC;
        live[q] = i < n;
        px[q] = py[q] = pz[q] = T(0);
        if (live[q]) {
            const V4 xv = x4[i];
            px[q] = xv.x * s, py[q] = xv.y * s, pz[q] = xv.z * s;
        }
    }
    for (int b = 0; b < ncb; ++b) {
        V4 cv[PER];
#pragma unroll
        for (int jj = 0; jj < PER; ++jj) cv[jj] = sctrl[b * UB + j0 + jj];
        vec_t* dst = reinterpret_cast<vec_t*>(ublk + ((cb0 + b) * n_pad + wbase) * UB) + lane;
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            vec_t o;
#pragma unroll
            for (int jj = 0; jj < PER; ++jj) {
                const T k = kernel_value(px[q], py[q], pz[q], cv[jj].x, cv[jj].y, cv[jj].z);
                o[jj] = (live[q] && cv[jj].w != 0) ? k : T(0);
            }
            __builtin_nontemporal_store(o, dst + 64 * q);
        }
    }
}

#ifndef MVF_UG
#define MVF_UG 2
#endif
constexpr int UG = MVF_UG;  // k-steps (of 4 cells) per software-pipeline group

#ifndef MVF_CACHED_WPS
#define MVF_CACHED_WPS 2
#endif
// Software pipeline of the cached kernel: a ring of NBUF operand groups of UGT k-steps each; the loads of group
// g + NBUF - 1 are issued before the MFMAs of group g.  float: 2 groups x 2 k-steps (20 + 20 operand VGPRs next to
// the 128 accumulator VGPRs).  double: operands are twice as wide, and 2 x 2 spills (256 VGPRs + scratch, 36 TF
// measured): 3 groups x 1 k-step keeps the same prefetch distance (2 k-steps) in 66 instead of 88 VGPRs.
#ifndef MVF_DBL_UG
#define MVF_DBL_UG 1
#endif
#ifndef MVF_DBL_NBUF
#define MVF_DBL_NBUF 2
#endif
template <typename T> struct CachedPipe { static constexpr int UGT = UG, NBUF = 2; };
template <> struct CachedPipe<double> { static constexpr int UGT = MVF_DBL_UG, NBUF = MVF_DBL_NBUF; };

// One wave's share of a tile: NA x NB blocks of 16 x 16 (TRI: only the blocks b >= a of a square arrangement) whose
// first row block is `rb0` and first column block `cb0` (absolute 16-wide block indices into Ublk); results go to
// out[(orow0 + ...) * GT + ocol0 + ...] of the 128 x 128 partial tile.
// DW >= 0 (diagonal tile, balanced): NA = 2, NB = 8; this wave takes row blocks DW and 7 - DW of the tile and, of each, only
// the blocks on or above the diagonal (8 - DW and DW + 1 of them: 9 for every wave - the four waves of a diagonal tile finish
// together and execute 36 of the 64 blocks; same registers as the full 2 x 8 shape, of which it is a sub-shape).
// PAIR2 (NA = 4, NB = 4): the row blocks a = 0, 1 belong to one tile (rb0, out) and a = 2, 3 to ANOTHER tile of the same tile
// column (rb1, out1) - two narrow last-column tiles in one job of full length.
template <typename T, int NA, int NB, bool TRI, int DW = -1, bool PAIR2 = false>
__device__ __forceinline__ void cached_block(const T* __restrict__ ublk, const T* __restrict__ P, int64_t n,
                                             int64_t n_pad, int64_t n0, int64_t n1, int64_t rb0, int64_t cb0,
                                             double* __restrict__ out, int orow0, int ocol0, int64_t rb1 = 0,
                                             double* __restrict__ out1 = nullptr) {
    constexpr int UGT = CachedPipe<T>::UGT, NBUF = CachedPipe<T>::NBUF;
    static_assert(DW < 0 || (NA == 2 && NB == 8 && !TRI && DW < 4), "balanced diagonal shape is a sub-shape of 2 x 8");
    static_assert(!PAIR2 || (NA == 4 && NB == 4 && !TRI && DW < 0), "paired edge shape: 2 + 2 row blocks x 4 column blocks");
    const int lane = threadIdx.x & 63;
    const int li = lane & 15, lk = lane >> 4;
    // block (a, b) is computed iff ...
    auto blk_live = [](int a, int b) constexpr {
        if (DW >= 0) return b >= (a == 0 ? DW : 7 - DW);
        return !TRI || b >= a;
    };
    // row block of operand a relative to rb0, and its first row in the partial tile relative to orow0
    auto rblk = [](int a) constexpr { return DW >= 0 ? (a == 0 ? DW : 7 - DW) : a; };

    const T* pa[NA];
    const T* pb[NB];
#pragma unroll
    for (int a = 0; a < NA; ++a)
        pa[a] = ublk + ((PAIR2 ? (a < 2 ? rb0 + a : rb1 + (a - 2)) : rb0 + rblk(a)) * n_pad + n0 + lk) * UB + li;
#pragma unroll
    for (int b = 0; b < NB; ++b) pb[b] = ublk + ((cb0 + b) * n_pad + n0 + lk) * UB + li;

    f64x4 acc[NA][NB];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};

    T ua[NBUF][UGT][NA], ub[NBUF][UGT][NB], pp[NBUF][UGT];
    const int ngroups = (int)((n1 - n0) / (4 * UGT));  // slices are multiples of 256 cells

    // Address arithmetic is VALU work too (64-bit adds) and VALU time is additive to f64 MFMA time: the operand
    // pointers and the P pointer advance ONCE per superblock of SUPER groups (4 KB per pointer); inside it every load
    // uses a compile-time immediate offset.  P is read branch-free: cached rows of padded cells are zero, so any finite
    // P does; only the remainder loop clamps its index.
    constexpr int SUPER = 4096 / (UGT * 4 * UB * (int)sizeof(T));  // groups per 4 KB of one pointer's stream
    const T* pP = P + n0 + lk;
    const int pmax = (int)min((int64_t)0x3fffffff, n - 1 - n0 - lk);  // last valid index from pP (may be < 0: P[n-1])
    auto load_group = [&](int64_t gbase, int s, bool in_tail, T(&A)[UGT][NA], T(&B)[UGT][NB], T(&Pq)[UGT]) {
        // group gbase + s; `s` is a compile-time constant after unrolling, gbase advances once per superblock
#pragma unroll
        for (int q = 0; q < UGT; ++q) {
            const int64_t off = (gbase * UGT) * (4 * UB) + (s * UGT + q) * (4 * UB);
#ifdef MVF_PROBE_NO_LOAD
#pragma unroll
            for (int a = 0; a < NA; ++a) A[q][a] = (T)(0.5 + lane * 1e-3 + off * 1e-9);
#pragma unroll
            for (int b = 0; b < NB; ++b) B[q][b] = (T)(0.25 + lane * 1e-3);
#else
#pragma unroll
            for (int a = 0; a < NA; ++a) A[q][a] = pa[a][off];
#pragma unroll
            for (int b = 0; b < NB; ++b) B[q][b] = pb[b][off];
#endif
#if defined(MVF_PROBE_NO_LOAD) || defined(MVF_PROBE_NO_P)
            Pq[q] = T(1);
#else
            const int64_t poff = (gbase * UGT) * 4 + (s * UGT + q) * 4;
            Pq[q] = in_tail ? pP[min((int)poff, pmax)] : pP[poff];
#endif
        }
    };
    auto compute_group = [&](const T(&A)[UGT][NA], const T(&B)[UGT][NB], const T(&Pq)[UGT]) {
#pragma unroll
        for (int q = 0; q < UGT; ++q) {
            double fa[NA], fb[NB];
            const double pd = (double)Pq[q];
#pragma unroll
            for (int a = 0; a < NA; ++a) {
#ifdef MVF_PROBE_NO_P
                fa[a] = (double)A[q][a];
#else
                fa[a] = (double)A[q][a] * pd;  // exact in float64
#endif
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) fb[b] = (double)B[q][b];
#ifdef MVF_PROBE_NO_MFMA
#pragma unroll
            for (int a = 0; a < NA; ++a) acc[a][a][0] += fa[a] + fb[a];
#else
            // the two waves of a SIMD run out of phase (one converting / multiplying the next operands, one issuing
            // MFMAs): raising the priority of the MFMA cluster keeps the matrix pipe fed (+2.5 % measured)
#ifndef MVF_PROBE_NO_SETPRIO
            __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
            for (int a = 0; a < NA; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    if (blk_live(a, b))
                        acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, 0);
#ifndef MVF_PROBE_NO_SETPRIO
            __builtin_amdgcn_s_setprio(0);
#endif
#endif
        }
    };

    static_assert(SUPER % NBUF == 0, "ring slots must be static inside a superblock");
    constexpr int AHEAD = NBUF - 1;
    // main part: whole superblocks whose cells all exist (no index clamp on P)
    const int64_t live = max((int64_t)0, min(n1, n) - n0);
    const int ng_main = (int)(min((int64_t)ngroups, live / (4 * UGT)) / SUPER) * SUPER;
#pragma unroll
    for (int s = 0; s < AHEAD; ++s)
        if (s < ng_main) load_group(0, s, false, ua[s], ub[s], pp[s]);
    for (int g0 = 0; g0 < ng_main; g0 += SUPER) {
#pragma unroll
        for (int s = 0; s < SUPER; ++s) {
            if (g0 + s + AHEAD < ng_main)
                load_group(g0, s + AHEAD, false, ua[(s + AHEAD) % NBUF], ub[(s + AHEAD) % NBUF], pp[(s + AHEAD) % NBUF]);
            compute_group(ua[s % NBUF], ub[s % NBUF], pp[s % NBUF]);
            __builtin_amdgcn_sched_barrier(0);  // keep the unrolled groups in program order (else the loads of all
                                                // SUPER groups are hoisted and the kernel spills)
        }
    }
    // remainder (< SUPER groups, plus the padded cells of the last slice): unpipelined, P index clamped
    for (int g = ng_main; g < ngroups; ++g) {
        load_group(g, 0, true, ua[0], ub[0], pp[0]);
        compute_group(ua[0], ub[0], pp[0]);
    }

#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
            if (blk_live(a, b)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = orow0 + (PAIR2 ? (a & 1) : rblk(a)) * 16 + lk + 4 * r;
                    const int col = ocol0 + b * 16 + li;
                    double* o = (PAIR2 && a >= 2) ? out1 : out;
                    __builtin_nontemporal_store(acc[a][b][r], &o[row * GT + col]);  // read once, by the reduction
                }
            }
}

template <typename T>
__global__ __launch_bounds__(256, MVF_CACHED_WPS) void gram_cached_kernel(const T* __restrict__ ublk, const T* __restrict__ P,
                                                             int64_t n, int64_t n_pad, int64_t m, int nt, int npairs,
                                                             int64_t slice_len, int64_t slice0,
                                                             double* __restrict__ partial, int njobs, int edge2) {
    // Off-diagonal tile: wave tile 32 x 128 (2 row blocks x 8 column blocks of 16) - the four waves stack in the row
    // direction and all read the same 8 column panels (L1 hits); per k-step a lane does 10 loads and only TWO
    // v_mul_f64 (P K, kept exact) for 16 MFMAs - the f64 VALU work is what steals MFMA time on gfx950.
    // Work that the reduction never reads is not computed:
    //  * last tile column with <= 64 live control points (m = 3000: 56 of 128): 2 x 4 blocks per wave;
    //  * diagonal tile: only blocks on or above the diagonal - waves 0, 1 take the upper-right 64 x 64 quadrant
    //    (2 x 4 blocks each), waves 2, 3 the upper triangles of the two diagonal quadrants (10 of 16 blocks each).
    constexpr int TB = GT / UB;  // 8 blocks per tile side
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int ti, tj, pair;
    int64_t slice;
    if constexpr (sizeof(T) == 4) {
        // (the float32 instantiation keeps its job decode to the letter: its hot loop's register allocation moves with it)
        pair = blockIdx.x % npairs;
        slice = blockIdx.x / npairs;
        decode_pair(pair, nt, ti, tj);
    } else {
        // edge2 (float64, last tile column with <= 64 live control points): the jobs of a slice are the tile pairs of the
        // leading (nt - 1) x (nt - 1) triangle, then the nt tiles of the last column TWO per job - rows 2e and 2e + 1, 2 + 2 row
        // blocks x 4 column blocks per wave: the same 16 MFMAs per k-step as every other job, so the jobs of a slice still run
        // in step (the half-length edge jobs of the float32 instantiation cost the float64 one 4 %: it lives on that
        // lock-step), and the dead half of those tiles is not computed (12 of 300 jobs at m = 3000).
        const int job = blockIdx.x % njobs;
        slice = blockIdx.x / njobs;
        if (edge2 && job >= npairs - nt) {
            const int e = job - (npairs - nt);
            ti = 2 * e, tj = nt - 1;
            if (ti + 1 < nt) {
                const int64_t n0p = (slice0 + slice) * slice_len;
                const int64_t n1p = min(n_pad, n0p + slice_len);
                auto slot = [&](int t) { return t * nt - t * (t - 1) / 2 + (tj - t); };
                double* oa = partial + ((size_t)slice * npairs + slot(ti)) * (size_t)(GT * GT);
                double* ob = partial + ((size_t)slice * npairs + slot(ti + 1)) * (size_t)(GT * GT);
                cached_block<T, 4, 4, false, -1, true>(ublk, P, n, n_pad, n0p, n1p, (int64_t)ti * TB + 2 * wave,
                                                       (int64_t)tj * TB, oa, 32 * wave, 0, (int64_t)(ti + 1) * TB + 2 * wave, ob);
                return;
            }
        } else if (edge2) {
            decode_pair(job, nt - 1, ti, tj);
        } else {
            decode_pair(job, nt, ti, tj);
        }
        pair = ti * nt - ti * (ti - 1) / 2 + (tj - ti);
    }
    const int64_t n0 = (slice0 + slice) * slice_len;  // `slice` indexes the partial tile inside this launch's phase
    const int64_t n1 = min(n_pad, n0 + slice_len);
    double* out = partial + ((size_t)slice * npairs + pair) * (size_t)(GT * GT);
    const int64_t rb = (int64_t)ti * TB, cb = (int64_t)tj * TB;
    // float32: work the reduction never reads is not computed - the edge shape (2 x 4: last tile column with <= 64 live
    // control points) and the balanced diagonal shape (every wave 9 of the 36 blocks on or above the diagonal; rounds
    // 1 - 2 used a 4 x 4 triangular body for waves 2, 3: 10 blocks against 8 for waves 0, 1) - both sub-shapes of the 2 x 8
    // wave tile.  Same-box A/B (profiles/r03_gram_diag_ab.md): float32 67.13 -> 67.36 TF at 8 M cells.
    // float64: every tile in full.  Measured twice now: round 2's triangular body cost registers (-5 %); round 3's
    // register-neutral sub-shapes lose 4 % at 8 M cells (60.5 -> 58.2 TF) and nothing at 1 M - the float64 kernel lives
    // on the tile pairs of a slice streaming their panels in step, and jobs of unequal length drift apart.
#ifdef MVF_PROBE_NO_SKIP
    constexpr bool skip = false;
#else
    constexpr bool skip = sizeof(T) == 4;
#endif
    // probe arms for the float64 instantiation (profiles/r03_gram_diag_ab.md): 1 = edge shape only, 2 = diagonal shape only
#ifndef MVF_PROBE_F64_SHAPES
#define MVF_PROBE_F64_SHAPES 0
#endif
    constexpr bool use_edge = skip || (MVF_PROBE_F64_SHAPES & 1);
    constexpr bool use_diag = skip || (MVF_PROBE_F64_SHAPES & 2);
    if (ti != tj || !use_diag) {
        const int64_t live_cols = m - (int64_t)tj * GT;  // > 0
        if (use_edge && ti != tj && live_cols <= 4 * UB)
            cached_block<T, 2, 4, false>(ublk, P, n, n_pad, n0, n1, rb + 2 * wave, cb, out, 32 * wave, 0);
        else
            cached_block<T, 2, 8, false>(ublk, P, n, n_pad, n0, n1, rb + 2 * wave, cb, out, 32 * wave, 0);
    } else {
        switch (wave) {
            case 0: cached_block<T, 2, 8, false, 0>(ublk, P, n, n_pad, n0, n1, rb, cb, out, 0, 0); break;
            case 1: cached_block<T, 2, 8, false, 1>(ublk, P, n, n_pad, n0, n1, rb, cb, out, 0, 0); break;
            case 2: cached_block<T, 2, 8, false, 2>(ublk, P, n, n_pad, n0, n1, rb, cb, out, 0, 0); break;
            default: cached_block<T, 2, 8, false, 3>(ublk, P, n, n_pad, n0, n1, rb, cb, out, 0, 0); break;
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------
// rhs:  R[j, :] = sum_n K(x_n, c_j) P_n y_n   (VALU kernel; a lane owns RHS_CPT control points, cells broadcast
// from LDS; float64 accumulation)
// ----------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rhs_kernel(const T* __restrict__ x4, const T* __restrict__ P,
                                                  const T* __restrict__ y4, int64_t n, const T* __restrict__ ctrl4,
                                                  int64_t m, T s, int64_t slice_len, double* __restrict__ rpart) {
    using V4T = typename Vec4<T>::type;
    __shared__ V4T sx[GCHUNK];
    __shared__ V4T sw[GCHUNK];
    const int64_t slice = blockIdx.y;
    const int64_t n0 = slice * slice_len, n1 = min(n, n0 + slice_len);
    T cx[RHS_CPT], cy[RHS_CPT], cz[RHS_CPT];
    double r0[RHS_CPT], r1[RHS_CPT], r2[RHS_CPT];
#pragma unroll
    for (int c = 0; c < RHS_CPT; ++c) {
        const int64_t j = (int64_t)blockIdx.x * RHS_COLS + c * 256 + threadIdx.x;
        if (j < m) {
            const V4T cv = reinterpret_cast<const V4T*>(ctrl4)[j];
            cx[c] = cv.x * s, cy[c] = cv.y * s, cz[c] = cv.z * s;
        } else {
            cx[c] = cy[c] = cz[c] = T(0);
        }
        r0[c] = r1[c] = r2[c] = 0.0;
    }
    for (int64_t c0 = n0; c0 < n1; c0 += GCHUNK) {
        __syncthreads();
        const int64_t i = c0 + threadIdx.x;
        if (i < n1) {
            const V4T xv = reinterpret_cast<const V4T*>(x4)[i];
            const V4T yv = reinterpret_cast<const V4T*>(y4)[i];
            const T p = P[i];
            sx[threadIdx.x] = V4T{xv.x * s, xv.y * s, xv.z * s, 0};
            sw[threadIdx.x] = V4T{yv.x * p, yv.y * p, yv.z * p, 0};
        } else {
            sx[threadIdx.x] = V4T{0, 0, 0, 0};
            sw[threadIdx.x] = V4T{0, 0, 0, 0};  // zero weight
        }
        __syncthreads();
        // float64 accumulation on purpose: rounding noise in R is NOT of the form U^T P (dY), so the ill-conditioned
        // solve amplifies it by 1/sigma_min(U) (a float32 accumulation here cost 1e-2 in V)
#pragma unroll 4
        for (int q = 0; q < GCHUNK; ++q) {
            const V4T xv = sx[q];
            const V4T wv = sw[q];
#pragma unroll
            for (int c = 0; c < RHS_CPT; ++c) {
                const double k = (double)kernel_value(xv.x, xv.y, xv.z, cx[c], cy[c], cz[c]);
                r0[c] = fma(k, (double)wv.x, r0[c]);
                r1[c] = fma(k, (double)wv.y, r1[c]);
                r2[c] = fma(k, (double)wv.z, r2[c]);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < RHS_CPT; ++c) {
        const int64_t j = (int64_t)blockIdx.x * RHS_COLS + c * 256 + threadIdx.x;
        if (j < m) {
            double4* o = reinterpret_cast<double4*>(rpart + ((size_t)slice * m + j) * 4);
            *o = double4{r0[c], r1[c], r2[c], 0.0};
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------
// deterministic reductions of the per-slice partials
// ----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gram_reduce_kernel(const double* __restrict__ partial, int64_t nslices, int nt,
                                                          int npairs, int64_t m, double* __restrict__ G, int accumulate) {
    const int pair = blockIdx.y;
    int ti, tj;
    decode_pair(pair, nt, ti, tj);
    const int e = blockIdx.x * 256 + threadIdx.x;  // element of the 128 x 128 tile
    const int row = e / GT, col = e % GT;
    const int64_t i = (int64_t)ti * GT + row, j = (int64_t)tj * GT + col;
    if (i >= m || j >= m) return;
    if (ti == tj && col < row) return;  // diagonal tiles: keep the upper triangle, mirror it -> G exactly symmetric
    // eight interleaved partial sums (slice s goes to sum s % 8), combined pairwise in a fixed order: deterministic, and
    // eight loads in flight instead of one dependent chain (78 us -> ~12 us for the 391 slices of a 50 k x 100 fit)
    const double* p = partial + (size_t)pair * (GT * GT) + e;
    const size_t stride = (size_t)npairs * (GT * GT);
    double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    int64_t s = 0;
    for (; s + 8 <= nslices; s += 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] += p[(s + q) * stride];
    }
    for (int q = 0; s < nslices; ++s, ++q) a[q] += p[s * stride];
    double acc = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    if (accumulate) acc += G[i * m + j];  // later phases continue the sum
    G[i * m + j] = acc;
    if (i != j) G[j * m + i] = acc;
}

__global__ __launch_bounds__(256) void rhs_reduce_kernel(const double* __restrict__ rpart, int64_t rslices, int64_t m,
                                                         double* __restrict__ R /* m x 3 */) {
    // 16 control points x 16 slice lanes per workgroup (a serial loop over ~1000 slices per output was latency bound:
    // 78 us at M = 500); the combination order is fixed, so the result is deterministic.
    __shared__ double sm[16][16][3];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t j = (int64_t)blockIdx.x * 16 + tx;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    if (j < m) {
        for (int64_t s = ty; s < rslices; s += 16) {
            const double4 v = *reinterpret_cast<const double4*>(rpart + ((size_t)s * m + j) * 4);
            a0 += v.x, a1 += v.y, a2 += v.z;
        }
    }
    sm[ty][tx][0] = a0, sm[ty][tx][1] = a1, sm[ty][tx][2] = a2;
    __syncthreads();
    if (threadIdx.x < 48) {
        const int o = threadIdx.x / 3, comp = threadIdx.x % 3;
        const int64_t jo = (int64_t)blockIdx.x * 16 + o;
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += sm[q][o][comp];
        if (jo < m) R[jo * 3 + comp] = t;
    }
}

}  // namespace mvf

using namespace mvf;

extern "C" size_t mvf_gram_workspace_bytes(int64_t n, int64_t m, mvf_dtype dtype) {
    if (n <= 0 || m <= 0) return 0;
    const GramPlan p = make_plan(n, m, dtype);
    return align_up(p.gram_bytes, 256) + align_up(p.rhs_bytes, 256);
}

extern "C" int mvf_gram_stages(int stages, const void* x4, const void* P, const void* y4, int64_t n,
                               const void* ctrl4, int64_t m, double beta, double* G, double* R, void* workspace,
                               size_t workspace_bytes, mvf_dtype dtype, void* stream) {
    MVF_REQUIRE(n >= 0 && m >= 0, "mvf_gram: bad shape");
    MVF_REQUIRE(beta >= 0.0 && std::isfinite(beta), "mvf_gram: beta must be finite and >= 0");
    MVF_REQUIRE(dtype == MVF_F32 || dtype == MVF_F64, "mvf_gram: bad dtype %d", (int)dtype);
    MVF_REQUIRE(stages > 0 && stages <= 15, "mvf_gram: bad stage mask %d", stages);
    if (m == 0) return 0;
    MVF_REQUIRE(!(stages & MVF_GRAM_STAGE_REDUCE) || G, "mvf_gram: null G");
    MVF_REQUIRE(!(stages & MVF_GRAM_STAGE_REDUCE_RHS) || R, "mvf_gram: null R");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        if (stages & MVF_GRAM_STAGE_REDUCE) MVF_CHECK_HIP(hipMemsetAsync(G, 0, sizeof(double) * m * m, st));
        if (stages & MVF_GRAM_STAGE_REDUCE_RHS) MVF_CHECK_HIP(hipMemsetAsync(R, 0, sizeof(double) * m * 3, st));
        return 0;
    }
    MVF_REQUIRE(x4 && P && ctrl4, "mvf_gram: null input");
    MVF_REQUIRE(!(stages & MVF_GRAM_STAGE_RHS) || y4, "mvf_gram: null y4");
    const GramPlan p = make_plan(n, m, dtype);
    const size_t need = align_up(p.gram_bytes, 256) + align_up(p.rhs_bytes, 256);
    MVF_REQUIRE(workspace && workspace_bytes >= need, "mvf_gram: workspace too small (%zu < %zu)", workspace_bytes, need);
    MVF_REQUIRE((int64_t)p.phase_slices * p.npairs < (1LL << 31), "mvf_gram: too many jobs");
    MVF_REQUIRE(p.rslices <= 65535, "mvf_gram: rhs slice count too large");
    double* gpart = (double*)workspace;
    double* rpart = (double*)((char*)workspace + align_up(p.gram_bytes, 256));
    const double s = std::sqrt(beta * LOG2E);
    dim3 rgrid((unsigned)p.rcolblocks, (unsigned)p.rslices);
    if (stages & MVF_GRAM_STAGE_TILES) {
        MVF_REQUIRE(p.nphases == 1 || G, "mvf_gram: null G (the tile stage reduces all but its last phase)");
        for (int64_t ph = 0; ph < p.nphases; ++ph) {
            const int64_t s0 = ph * p.phase_slices, ns = std::min(p.phase_slices, p.nslices - s0);
            const unsigned njobs = (unsigned)(ns * p.npairs);
            if (dtype == MVF_F32)
                hipLaunchKernelGGL(gram_f64acc_kernel<float>, dim3(njobs), dim3(256), 0, st, (const float4*)x4,
                                   (const float*)P, n, (const float4*)ctrl4, m, (float)s, p.nt, p.npairs, p.slice_len, s0,
                                   gpart);
            else
                hipLaunchKernelGGL(gram_f64acc_kernel<double>, dim3(njobs), dim3(256), 0, st, (const double4*)x4,
                                   (const double*)P, n, (const double4*)ctrl4, m, s, p.nt, p.npairs, p.slice_len, s0,
                                   gpart);
            if (ph + 1 < p.nphases)  // the buffer is reused: fold this phase into G now
                hipLaunchKernelGGL(gram_reduce_kernel, dim3(GT * GT / 256, (unsigned)p.npairs), dim3(256), 0, st, gpart,
                                   ns, p.nt, p.npairs, m, G, ph > 0 ? 1 : 0);
        }
        MVF_LAUNCH_CHECK();
    }
    if (stages & MVF_GRAM_STAGE_RHS) {
        if (dtype == MVF_F32)
            hipLaunchKernelGGL(rhs_kernel<float>, rgrid, dim3(256), 0, st, (const float*)x4, (const float*)P,
                               (const float*)y4, n, (const float*)ctrl4, m, (float)s, p.rslice_len, rpart);
        else
            hipLaunchKernelGGL(rhs_kernel<double>, rgrid, dim3(256), 0, st, (const double*)x4, (const double*)P,
                               (const double*)y4, n, (const double*)ctrl4, m, s, p.rslice_len, rpart);
        MVF_LAUNCH_CHECK();
    }
    if (stages & MVF_GRAM_STAGE_REDUCE) {
        // the last phase's partial tiles (all of them when there is one phase), added to what the tile stage folded in
        const int64_t s0 = (p.nphases - 1) * p.phase_slices;
        hipLaunchKernelGGL(gram_reduce_kernel, dim3(GT * GT / 256, (unsigned)p.npairs), dim3(256), 0, st, gpart,
                           p.nslices - s0, p.nt, p.npairs, m, G, p.nphases > 1 ? 1 : 0);
        MVF_LAUNCH_CHECK();
    }
    if (stages & MVF_GRAM_STAGE_REDUCE_RHS) {
        hipLaunchKernelGGL(rhs_reduce_kernel, dim3((unsigned)cdiv(m, 16)), dim3(256), 0, st, rpart, p.rslices, m, R);
        MVF_LAUNCH_CHECK();
    }
    return 0;
}

static inline int64_t ublk_npad(int64_t n) { return cdiv(n, GCHUNK) * GCHUNK; }
static inline int64_t ublk_mpad(int64_t m) { return cdiv(m, GT) * GT; }

extern "C" size_t mvf_ublk_bytes(int64_t n, int64_t m, mvf_dtype dtype) {
    if (n <= 0 || m <= 0) return 0;
    return (size_t)ublk_npad(n) * (size_t)ublk_mpad(m) * (dtype == MVF_F64 ? sizeof(double) : sizeof(float));
}

extern "C" int mvf_ublk_build(const void* x4, int64_t n, const void* ctrl4, int64_t m, double beta, void* ublk,
                              size_t ublk_bytes, mvf_dtype dtype, void* stream) {
    MVF_REQUIRE(n > 0 && m > 0, "mvf_ublk_build: need n > 0 and m > 0");
    MVF_REQUIRE(dtype == MVF_F32 || dtype == MVF_F64, "mvf_ublk_build: bad dtype %d", (int)dtype);
    MVF_REQUIRE(beta >= 0.0 && std::isfinite(beta), "mvf_ublk_build: beta must be finite and >= 0");
    MVF_REQUIRE(x4 && ctrl4 && ublk, "mvf_ublk_build: null pointer");
    MVF_REQUIRE(ublk_bytes >= mvf_ublk_bytes(n, m, dtype), "mvf_ublk_build: buffer too small (%zu < %zu)", ublk_bytes,
                mvf_ublk_bytes(n, m, dtype));
    hipStream_t st = (hipStream_t)stream;
    const int64_t n_pad = ublk_npad(n), m_pad = ublk_mpad(m);
    const int cb_per_block = 32;  // 512 control points (8 KiB of LDS) per workgroup column
    dim3 grid((unsigned)(n_pad / 256), (unsigned)cdiv(m_pad / UB, cb_per_block));
    MVF_REQUIRE(grid.y <= 65535, "mvf_ublk_build: m too large");
    const double s = std::sqrt(beta * LOG2E);
    if (dtype == MVF_F32)
        hipLaunchKernelGGL(ublk_build_kernel<float>, grid, dim3(256), cb_per_block * UB * sizeof(float4), st,
                           (const float4*)x4, n, n_pad, (const float4*)ctrl4, m, m_pad, (float)s, cb_per_block,
                           (float*)ublk);
    else
        hipLaunchKernelGGL(ublk_build_kernel<double>, grid, dim3(256), cb_per_block * UB * sizeof(double4), st,
                           (const double4*)x4, n, n_pad, (const double4*)ctrl4, m, m_pad, s, cb_per_block,
                           (double*)ublk);
    MVF_LAUNCH_CHECK();
    return 0;
}

extern "C" int mvf_gram_cached(int stages, const void* ublk, const void* x4, const void* P, const void* y4, int64_t n,
                               const void* ctrl4, int64_t m, double beta, double* G, double* R, void* workspace,
                               size_t workspace_bytes, mvf_dtype dtype, void* stream) {
    MVF_REQUIRE(n > 0 && m > 0, "mvf_gram_cached: need n > 0 and m > 0");
    MVF_REQUIRE(dtype == MVF_F32 || dtype == MVF_F64, "mvf_gram_cached: bad dtype %d", (int)dtype);
    MVF_REQUIRE(stages > 0 && stages <= 15, "mvf_gram_cached: bad stage mask %d", stages);
    MVF_REQUIRE(ublk && P, "mvf_gram_cached: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (stages & MVF_GRAM_STAGE_TILES) {
        const GramPlan p = make_plan(n, m, dtype);
        const size_t need = align_up(p.gram_bytes, 256) + align_up(p.rhs_bytes, 256);
        MVF_REQUIRE(workspace && workspace_bytes >= need, "mvf_gram_cached: workspace too small (%zu < %zu)",
                    workspace_bytes, need);
        MVF_REQUIRE((int64_t)p.phase_slices * p.npairs < (1LL << 31), "mvf_gram_cached: too many jobs");
        MVF_REQUIRE(p.nphases == 1 || G, "mvf_gram_cached: null G (the tile stage reduces all but its last phase)");
        for (int64_t ph = 0; ph < p.nphases; ++ph) {
            const int64_t s0 = ph * p.phase_slices, ns = std::min(p.phase_slices, p.nslices - s0);
            const unsigned njobs = (unsigned)(ns * p.npairs);
            if (dtype == MVF_F32)
                hipLaunchKernelGGL(gram_cached_kernel<float>, dim3(njobs), dim3(256), 0, st, (const float*)ublk,
                                   (const float*)P, n, ublk_npad(n), m, p.nt, p.npairs, p.slice_len, s0,
                                   (double*)workspace, p.npairs, 0);
            else
                hipLaunchKernelGGL(gram_cached_kernel<double>, dim3((unsigned)(ns * p.njobs)), dim3(256), 0, st,
                                   (const double*)ublk, (const double*)P, n, ublk_npad(n), m, p.nt, p.npairs, p.slice_len, s0,
                                   (double*)workspace, p.njobs, p.edge2);
            if (ph + 1 < p.nphases)
                hipLaunchKernelGGL(gram_reduce_kernel, dim3(GT * GT / 256, (unsigned)p.npairs), dim3(256), 0, st,
                                   (const double*)workspace, ns, p.nt, p.npairs, m, G, ph > 0 ? 1 : 0);
        }
        MVF_LAUNCH_CHECK();
    }
    const int rest = stages & (MVF_GRAM_STAGE_RHS | MVF_GRAM_STAGE_REDUCE | MVF_GRAM_STAGE_REDUCE_RHS);
    if (rest)
        return mvf_gram_stages(rest, x4, P, y4, n, ctrl4, m, beta, G, R, workspace, workspace_bytes, dtype, stream);
    return 0;
}

extern "C" int mvf_gram(const void* x4, const void* P, const void* y4, int64_t n, const void* ctrl4, int64_t m,
                        double beta, double* G, double* R, void* workspace, size_t workspace_bytes, mvf_dtype dtype,
                        void* stream) {
    return mvf_gram_stages(MVF_GRAM_STAGE_TILES | MVF_GRAM_STAGE_RHS | MVF_GRAM_STAGE_REDUCE | MVF_GRAM_STAGE_REDUCE_RHS, x4,
                           P, y4, n, ctrl4, m,
                           beta, G, R, workspace, workspace_bytes, dtype, stream);
}
