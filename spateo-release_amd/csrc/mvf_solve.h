// Internal (not part of the C ABI): the blocked Cholesky of mvf_solve.hip, shared with the minimum-norm solve.
#pragma once
#include "mvf_common.h"

namespace mvf {

constexpr int CHOL_NB = 64;

struct CholPlan {
    int64_t mp, mr;  // padded order (multiple of 64); rows of the trapezoidal work matrix (mp + 64 rhs rows)
    int nb, nbr;
    double *W, *Cp, *Yw, *rdiag, *scal;  // scal[0] = mean diagonal, scal[1] = shift * mean diagonal (what was added)
};

size_t chol_workspace_bytes(int64_t m, int nrhs);
void chol_layout(int64_t m, int nrhs, void* workspace, CholPlan* pl);
int chol_factor(hipStream_t st, const double* G, const double* K, double ls2, double shift, const double* R, int64_t m,
                int nrhs, void* workspace, CholPlan* pl, int* info);
int chol_factor_mat(hipStream_t st, const double* A, int64_t ld, double shift, int64_t m, void* workspace, CholPlan* pl,
                    int* info);
size_t chol_inv_workspace_bytes(int64_t m);
int chol_factor_mat_inv(hipStream_t st, const double* A, int64_t ld, int64_t m, void* workspace, CholPlan* pl, int* info,
                        int inverse, const int* order = nullptr, const int* oflag = nullptr);

}  // namespace mvf
