// em_train.cpp -- a compiled host driving the EM hot path through the C-ABI alone (no Python, no torch).
//
// What the reference's PPCATrainer.train does (python/ppca_rs/__init__.py:33-67) in terms of the boundary a
// Rust maintainer would bind: context -> device-resident dataset -> `iterate` loop with the
// llk of the input model as a by-product -> model back on the host.
//
//   g++ -O2 -std=c++17 -Iinclude examples/em_train.cpp -Lppca_rs_amd -lppca_hip -Wl,-rpath,$PWD/ppca_rs_amd -o examples/em_train
//   examples/em_train [n_samples] [d] [k] [n_iters]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "ppca_hip.h"

#define CHECK(call)                                                                     \
    do {                                                                                \
        int rc_ = (call);                                                               \
        if (rc_ != PPCA_OK) {                                                           \
            std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ppca_last_error()); \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? std::atoll(argv[1]) : 200000;
    const int d = argc > 2 ? std::atoi(argv[2]) : 256, k = argc > 3 ? std::atoi(argv[3]) : 10;
    const int iters = argc > 4 ? std::atoi(argv[4]) : 8;
    if (ppca_abi_version() != PPCA_ABI_VERSION) {
        std::fprintf(stderr, "ABI mismatch\n");
        return 1;
    }
    std::printf("path kind for (d=%d, k=%d): %d\n", d, k, ppca_path_kind(d, k));

    std::mt19937_64 rng(1234);
    std::normal_distribution<double> nd;
    std::vector<double> c_true((size_t)d * k), mean_true(d), c0((size_t)d * k), mean0(d, 0.0);
    for (double &v : c_true) v = nd(rng);
    for (double &v : mean_true) v = nd(rng);
    for (double &v : c0) v = nd(rng);  // PPCAModel::init: C0 ~ N(0,1), mean 0, sigma 1 (ppca_model.rs:51-70)

    ppca_ctx *ctx = nullptr;
    CHECK(ppca_ctx_create(-1, nullptr, &ctx));
    ppca_synth_spec spec{};
    spec.n_rows = n;
    spec.d = d;
    spec.k = k;
    spec.sigma = 0.1;
    spec.mask_prob = 0.3;
    spec.seed = 1013;
    spec.transform = c_true.data();
    spec.mean = mean_true.data();
    ppca_dataset *ds = nullptr;
    CHECK(ppca_dataset_generate(ctx, &spec, &ds));

    ppca_model *cur = nullptr, *next = nullptr;
    CHECK(ppca_model_create(ctx, d, k, 1.0, c0.data(), mean0.data(), &cur));
    CHECK(ppca_model_alloc(ctx, d, k, &next));
    double prev = -INFINITY;
    for (int it = 0; it < iters; ++it) {
        double llk = 0.0;
        CHECK(ppca_em_step(ctx, ds, cur, nullptr, next, &llk));  // PPCAModel::iterate (ppca_model.rs:277-393)
        std::printf("Masked PPCA iteration %d: llk=%.6f\n", it, llk / (double)n);
        if (!(llk >= prev - 1e-9 * std::fabs(llk))) {  // EM never decreases the likelihood (:263-265)
            std::fprintf(stderr, "llk decreased\n");
            return 2;
        }
        prev = llk;
        std::swap(cur, next);
    }
    double sigma = 0.0;
    std::vector<double> c_fit((size_t)d * k), mean_fit(d);
    CHECK(ppca_model_download(cur, &sigma, c_fit.data(), mean_fit.data()));
    std::printf("fitted isotropic noise %.6f (truth 0.1)\n", sigma);
    CHECK(ppca_model_free(cur));
    CHECK(ppca_model_free(next));
    CHECK(ppca_dataset_free(ds));
    CHECK(ppca_ctx_destroy(ctx));
    return 0;
}
