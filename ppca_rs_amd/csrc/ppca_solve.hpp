// ppca_solve.hpp -- the per-sample solve stage of the split pipeline (ppca_generic.hip): argument block shared by the
// solver kernels of ppca_generic.hip and the batched blocked solver of ppca_solve4.hip.  Not installed.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace ppca {

struct SolveArgs {
    double *G;        // [n][kp] in: packed Gram; out (EM): w P packed
    double *Bz;       // [n][k+1] in: b (k); out (EM): [w z | w]
    const double *xx; // [n]
    const double *mc; // [n]
    const double *w;  // [n] or nullptr
    int64_t n;
    int k;
    const double *model;  // device model buffer: sigma^2 and ln sigma are read on the device
    double *sc;       // [n][4]: sq, dev, w*llk, nonempty  (EM: sq/dev filled; post: only llk)
    int em;
    double *llks;     // post (nullable): per-sample llk
    double *states;   // post (nullable): [n][k]
    double *covs;     // post (nullable): [n][k][k]
    int need_sigma;   // post: some consumer reads the packed Sigma the solver leaves in G (covariances, their diagonals);
                      // 0: the lane-per-sample solver skips the k inverse columns (llk, states, smooth, extrapolate)
};

// Blocked in-place inversion on the fp64 MFMA, several samples per wave (ppca_solve4.hip): 17 <= k <= 64.
bool solve4_covers(int k);
hipError_t launch_solve4(const SolveArgs &a, int n_cu, hipStream_t s);

}  // namespace ppca
