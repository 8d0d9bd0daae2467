// ppca_llk.hip -- the log-likelihood sweep (PPCAModel::llk / llks, ppca/src/ppca_model.rs:124-159, on
// OutputCovariance::quadratic_form / covariance_log_det, ppca/src/output_covariance.rs:115-142) as its own kernel for
// the fused shapes (d <= 256, k <= 10).  It is the pass a mixture step runs once per component and per iteration
// (PPCAMix::iterate_with_prior, ppca/src/mix.rs:283-288) and the trainers' metric.
//
// pass_kernel<K, false, ...> serves this request with its general machinery: a 32-sample tile, the solver on 32 of 64
// lanes in every one of four waves, z by two substitutions, and room for the posterior / reconstruction outputs.  The
// llk of a sample needs only  m, |x~|^2, ln det M  and  quad = b^T M^-1 b = |L^-1 b|^2:
//   * TWO tiles per round: P1 / P2 of tile A, P1 / P2 of tile B (the x~ tile is free again after a tile's P2: nothing
//     reads it later in this pass), then ONE solver step for the 64 samples on all 64 lanes of wave 0 -- the other
//     waves have no share in it (there are no M^-1 columns to compute) and wait at the barrier;
//   * factorisation + the forward substitution only;
//   * the next tile's rows are requested as soon as the registers of the previous one are staged, so they travel
//     under a whole P2.
// Per-sample arithmetic is that of pass_kernel (same staging, same int8-sliced Gram behind the same guard, same
// Cholesky), so the llks agree with it to rounding of the final sum (quad is summed from the forward unknowns in the
// same order).  The guard's fallback is pass_kernel<K, false, 4, false>.
#include <atomic>
#include <cstdlib>

#include "ppca_device.hpp"

namespace ppca {

template <int K>
struct CfgL {
    using c = Cfg<K>;
    static constexpr int KP = c::KP, NTP = c::NTP, B = c::B, DP = c::DP, XS = c::XS, CS = c::CS;
    static constexpr int GS = 16 * NTP + 16 + 1;  // [G (16 NTP) | b partial of dims 0-127 (16)], 2 B rows
    static constexpr int BS = 17;                 // b partial of dims 128-255, 2 B rows
    static constexpr int OFF_X = 0;
    static constexpr int OFF_C = OFF_X + B * XS;
    static constexpr int OFF_G = OFF_C + DP * CS;
    static constexpr int OFF_B1 = OFF_G + 2 * B * GS;
    static constexpr int OFF_M = OFF_B1 + 2 * B * BS;   // mask words of the tile being contracted, B x 4 u64
    static constexpr int OFF_XX = OFF_M + B * 4;        // |x~|^2 of the 2 B samples
    static constexpr int OFF_MC = OFF_XX + 2 * B;       // observed counts, 2 B ints
    static constexpr int OFF_R = OFF_MC + B;            // cross-wave scratch
    static constexpr int LDS_DOUBLES = OFF_R + 16;
};

template <int K>
__global__ __launch_bounds__(256) void llk2_kernel(PassArgs p) {
    using cfg = CfgL<K>;
    constexpr int KP = cfg::KP, NTP = cfg::NTP, B = cfg::B, XS = cfg::XS, CS = cfg::CS, GS = cfg::GS, BS = cfg::BS;
    constexpr int NW = 4, RPW = B / NW, DPS = cfg::DP / 2, STEPS = DPS / 4;
    static_assert(NTP <= NW && QS == 8, "int8 Gram: one wave per packed-column tile, 8 digit slices");
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *Xs = sm + cfg::OFF_X;
    double *Cs = sm + cfg::OFF_C;
    double *Gs = sm + cfg::OFF_G;
    double *B1 = sm + cfg::OFF_B1;
    unsigned long long *Ms = reinterpret_cast<unsigned long long *>(sm + cfg::OFF_M);
    double *xxs = sm + cfg::OFF_XX;
    int *mcnt = reinterpret_cast<int *>(sm + cfg::OFF_MC);

    if (p.qflag) {  // qprep's dynamic-range guard: pass_kernel<K, false, 4, false> runs instead
        int unsafe = 0;
#pragma unroll
        for (int t = 0; t < NTP; ++t) unsafe |= p.qflag[t];
        if (unsafe) return;
    }
    const int tid = threadIdx.x, lane_entry = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int d = p.d;
    const int64_t n = p.n;
    const double *mC = p.model + MODEL_HDR;
    const double *mMean = mC + (int64_t)d * K;
    const double s2 = p.model[1], lnsig = p.model[2];
    for (int idx = tid; idx < cfg::DP * CS; idx += 256) {
        int j = idx / CS, a = idx - j * CS;
        Cs[idx] = (j < d && a < K) ? mC[(int64_t)j * K + a] : 0.0;
    }
    double mu[4];  // staging lane map: lane l holds dims 128 h + 2 l + e (element q = 2 h + e) of a row
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int j = 128 * (q >> 1) + 2 * lane_entry + (q & 1);
        mu[q] = (j < d) ? mMean[j] : 0.0;
    }
    bool dim_ok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) dim_ok[q] = 128 * (q >> 1) + 2 * lane_entry + (q & 1) < d;
    unsigned long long dimmask[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) dimmask[q] = __builtin_amdgcn_ballot_w64(dim_ok[q]);

    const int64_t ntiles = (n + B - 1) / B;
    const int64_t tiles_per_wg = (ntiles + gridDim.x - 1) / gridDim.x;
    const int64_t tile_begin = (int64_t)blockIdx.x * tiles_per_wg;
    const int64_t tile_end = tile_begin + tiles_per_wg < ntiles ? tile_begin + tiles_per_wg : ntiles;
    const int64_t nleft = n - tile_begin * B;
    const int nrel = (int)(nleft < (1 << 30) ? nleft : (1 << 30));
    const double *Xwg = p.X + tile_begin * B * p.ldx;
    double xr[RPW][4];
    auto load_tile = [&](int64_t tile) {  // unconditional (rows clamped to real ones; validity applied when staged)
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
            const int rel = (int)(tile - tile_begin) * B + wave * RPW + r;
            const int rc = nrel > 0 ? (rel < nrel ? rel : nrel - 1) : 0;
            const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<double *>(Xwg + (int64_t)rc * p.ldx), 0, nrel > 0 ? d * (int)sizeof(double) : 0, 0x00020000);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                typedef unsigned u4_t __attribute__((ext_vector_type(4)));
                const u4_t v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, lane_entry * 16, 1024 * h, 0);
                xr[r][2 * h] = __longlong_as_double(((long long)v[1] << 32) | v[0]);
                xr[r][2 * h + 1] = __longlong_as_double(((long long)v[3] << 32) | v[2]);
            }
        }
    };
    const __amdgpu_buffer_rsrc_t qrsrc = __builtin_amdgcn_make_buffer_rsrc(p.qtab, 0, (int)qtab_bytes<K>(), 0x00020000);
    const bool gram_wave = NTP >= NW || wave < NTP;
    auto load_pair = [&](i4_t(&dst)[2][4], int sl0) {
        int qbase = wave * QS * 4 * 1024;
        asm volatile("" : "+s"(qbase));
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                typedef unsigned u4_t __attribute__((ext_vector_type(4)));
                const u4_t v = __builtin_amdgcn_raw_buffer_load_b128(qrsrc, lane_entry * 16, qbase + ((sl0 + u) * 4 + kc) * 1024, 0);
                dst[u][kc] = i4_t{(int)v[0], (int)v[1], (int)v[2], (int)v[3]};
            }
    };

    // ---- P1: stage one tile (slot 0 / 1 of the round): x~ into the LDS tile, mask words, |x~|^2 and counts per sample
    auto stage_tile = [&](int64_t t, int lane, int slot) {
        int st_wlo = 0, st_whi = 0, st_m = 0, st_xlo = 0, st_xhi = 0;
        static_for<RPW>([&](auto r_tag) {
            constexpr int r = decltype(r_tag)::value;
            const int ri = wave * RPW + r;
            const bool row_ok = t < tile_end && (int)(t - tile_begin) * B + ri < nrel;  // wave-uniform
            double xt[4];
            unsigned long long bal[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double v = xr[r][q];
                // observed <=> finite (dataset.rs:19-22): |v| < inf straight into an SGPR pair
                bal[q] = __builtin_amdgcn_fcmp(__builtin_fabs(v), __builtin_inf(), 4) & (row_ok ? dimmask[q] : 0ull);
                xt[q] = keep_if(v - mu[q], bal[q]);  // select, never multiply (utils.rs:118-127)
            }
            double pc_xx = 0.0;
            int pc_m = 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                auto weave = [&](unsigned ev, unsigned od) {
                    unsigned long long re, ro;
                    asm("s_bitreplicate_b64_b32 %0, %1" : "=s"(re) : "s"(ev));
                    asm("s_bitreplicate_b64_b32 %0, %1" : "=s"(ro) : "s"(od));
                    return (re & 0x5555555555555555ull) | (ro & 0xAAAAAAAAAAAAAAAAull);
                };
                const unsigned long long b0 = bal[2 * h], b1 = bal[2 * h + 1];
                const unsigned long long w0 = weave((unsigned)b0, (unsigned)b1);
                const unsigned long long w1 = weave((unsigned)(b0 >> 32), (unsigned)(b1 >> 32));
                if (h == 0) {
                    st_wlo = writelane_s<4 * r>(st_wlo, (int)(unsigned)w0);
                    st_whi = writelane_s<4 * r>(st_whi, (int)(unsigned)(w0 >> 32));
                    st_wlo = writelane_s<4 * r + 1>(st_wlo, (int)(unsigned)w1);
                    st_whi = writelane_s<4 * r + 1>(st_whi, (int)(unsigned)(w1 >> 32));
                } else {
                    st_wlo = writelane_s<4 * r + 2>(st_wlo, (int)(unsigned)w0);
                    st_whi = writelane_s<4 * r + 2>(st_whi, (int)(unsigned)(w0 >> 32));
                    st_wlo = writelane_s<4 * r + 3>(st_wlo, (int)(unsigned)w1);
                    st_whi = writelane_s<4 * r + 3>(st_whi, (int)(unsigned)(w1 >> 32));
                }
                typedef double d2_t __attribute__((ext_vector_type(2)));
                *reinterpret_cast<d2_t *>(Xs + ri * XS + 128 * h + 2 * lane) = d2_t{xt[2 * h], xt[2 * h + 1]};
                pc_xx += xt[2 * h] * xt[2 * h];
                pc_xx += xt[2 * h + 1] * xt[2 * h + 1];
                pc_m += __popcll(b0) + __popcll(b1);
            }
            st_m = writelane<r>(st_m, pc_m);
            const long long tb = __double_as_longlong(wave_total(pc_xx));
            st_xlo = writelane<r>(st_xlo, (int)tb);  // (wave-uniform by v_readlane: the padded form of the lane write)
            st_xhi = writelane<r>(st_xhi, (int)(tb >> 32));
        });
        const unsigned long long myw = ((unsigned long long)(unsigned)st_whi << 32) | (unsigned)st_wlo;
        if (lane < 4 * RPW) Ms[wave * 4 * RPW + lane] = myw;
        if (lane < RPW) {
            xxs[slot * B + wave * RPW + lane] = __longlong_as_double(((long long)st_xhi << 32) | (unsigned)st_xlo);
            mcnt[slot * B + wave * RPW + lane] = st_m;
        }
    };
    // ---- P2: [G | b] of the staged tile into rows slot * B .. of the exchange buffers
    auto contract_tile = [&](int lane, int slot) {
        const int l15 = lane & 15, l4 = lane >> 4;
        const int colb = (l15 < K) ? l15 : K;
        const int rt = wave & 1, kq = wave >> 1;
        const int si = 16 * rt + l15;
        d4_t accb = d4_t{0, 0, 0, 0};
        const double *xrow = Xs + si * XS + DPS * kq + l4;
        const double *cpc = Cs + (DPS * kq + l4) * CS + colb;
        i4_t af[2][4], qbA[2][4], qbB[2][4];
        double v[2][4];
        auto group = [&](const i4_t(*qb)[4], bool first) {
#pragma unroll
            for (int rt2 = 0; rt2 < 2; ++rt2) {
                i4_t ia[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    ia[u] = i4_t{0, 0, 0, 0};
#pragma unroll
                    for (int kc = 0; kc < 4; ++kc)
                        ia[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[rt2][kc], qb[u][kc], ia[u], 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int part = ia[1][r] * 128 + ia[0][r];
                    v[rt2][r] = first ? (double)part : v[rt2][r] * 16384.0 + (double)part;
                }
            }
        };
        unsigned long long mwd[2][4];
#pragma unroll
        for (int rt2 = 0; rt2 < 2; ++rt2)
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) mwd[rt2][kc] = Ms[(16 * rt2 + l15) * 4 + kc];
        __builtin_amdgcn_sched_barrier(0);
        load_pair(qbA, 6);
        load_pair(qbB, 4);
        const double qs = gram_wave ? p.qscale[16 * wave + l15] : 0.0;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rt2 = 0; rt2 < 2; ++rt2)
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                const unsigned bits = (unsigned)(mwd[rt2][kc] >> (16 * l4)) & 0xFFFFu;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    af[rt2][kc][u] = (int)((((bits >> (4 * u)) & 0xFu) * 0x00204081u) & 0x01010101u);
            }
        {
            // b = X~ C first: its 32 fp64 MFMAs cover the arrival of the digit table
            constexpr int CH = 4;
            double axb[2][CH], cbb[2][CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                axb[0][u] = xrow[4 * u];
                cbb[0][u] = cpc[4 * u * CS];
            }
#pragma unroll
            for (int c = 0; c < STEPS / CH; ++c) {
                if (c + 1 < STEPS / CH) {
#pragma unroll
                    for (int u = 0; u < CH; ++u) {
                        axb[(c + 1) & 1][u] = xrow[4 * ((c + 1) * CH + u)];
                        cbb[(c + 1) & 1][u] = cpc[4 * ((c + 1) * CH + u) * CS];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < CH; ++u) accb = mfma(axb[c & 1][u], cbb[c & 1][u], accb);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        group(qbA, true);   // digits {7,6}
        load_pair(qbA, 2);
        group(qbB, false);  // digits {5,4}
        load_pair(qbB, 0);
        group(qbA, false);  // digits {3,2}
        group(qbB, false);  // digits {1,0}
        if (gram_wave) {
#pragma unroll
            for (int rt2 = 0; rt2 < 2; ++rt2)
#pragma unroll
                for (int r = 0; r < 4; ++r)  // C/D map of the 16x16 integer MFMA: row = 4 (lane >> 4) + reg
                    Gs[(slot * B + 16 * rt2 + 4 * l4 + r) * GS + 16 * wave + l15] = v[rt2][r] * qs;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // the K-split partials of b, summed by the solver in a fixed order (p0 + p1)
            if (kq == 0) Gs[(slot * B + 16 * rt + l4 + 4 * r) * GS + 16 * NTP + l15] = accb[r];
            else B1[(slot * B + 16 * rt + l4 + 4 * r) * BS + l15] = accb[r];
        }
    };

    double run_llk = 0.0, run_w = 0.0;  // wave 0: running sums of its lane's samples
    if (tile_begin < tile_end) load_tile(tile_begin);
    __syncthreads();
    for (int64_t tile = tile_begin; tile < tile_end; tile += 2) {
        int lane = lane_entry;
        asm volatile("" : "+v"(lane));
        stage_tile(tile, lane, 0);
        load_tile(tile + 1);  // travels under the contraction below
        __syncthreads();
        contract_tile(lane, 0);
        __syncthreads();
        stage_tile(tile + 1, lane, 1);
        load_tile(tile + 2);
        __syncthreads();
        contract_tile(lane, 1);
        __syncthreads();
        if (wave == 0) {
            // ---- P3: lane i < 32 -> sample i of tile, lane i >= 32 -> sample i - 32 of tile + 1
            const int slot = lane >> 5, i = lane & (B - 1);
            const int64_t t = tile + slot;
            const int64_t row = t * B + i;
            const bool mine = t < tile_end && row < n;
            const double *g0 = Gs + lane * GS;
            const double *b1 = B1 + lane * BS;
            const double wgt = mine ? (p.w ? p.w[row] : 1.0) : 0.0;
            const int m = mcnt[lane];
            const double xx = xxs[lane];
            Posterior<K> post;
            double pm;
            int pe;
            post.factor([&](int e) { return g0[e]; }, s2, pm, pe);
            const double quad = post.forward_quad([&](int a) { return g0[16 * NTP + a] + b1[a]; });
            const double lk = sample_llk(xx, quad, Posterior<K>::logdet(pm, pe), s2, lnsig, m, K);
            run_llk += wgt * lk;
            run_w += wgt;
            if (p.llks && mine) p.llks[row] = lk;
        }
        __syncthreads();
    }
    if (wave == 0) {
        const double v2 = wave_sum(run_llk), v3 = wave_sum(run_w);
        if (lane_entry == 0) {
            double *sc = p.scal_part + (int64_t)blockIdx.x * 8;
            sc[SC_SQERR] = 0.0;
            sc[SC_DEVSQ] = 0.0;
            sc[SC_LLK] = v2;
            sc[SC_SUMW] = v3;
            sc[SC_NONEMPTY] = 0.0;
            sc[5] = 0.0;
            sc[6] = 0.0;
            sc[7] = 0.0;
        }
    }
}

template <int K>
static hipError_t launch_llk2_t(int grid, const PassArgs &a, hipStream_t s) {
    const size_t lds = sizeof(double) * CfgL<K>::LDS_DOUBLES;
    static std::atomic<unsigned long long> done{0ull};
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&llk2_kernel<K>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        done.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL((llk2_kernel<K>), dim3(grid), dim3(256), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_llk2(int k, int grid, const PassArgs &a, hipStream_t s) {
    switch (k) {
#ifdef PPCA_DEV_K10
        case 10: return launch_llk2_t<10>(grid, a, s);
#else
        case 1: return launch_llk2_t<1>(grid, a, s);
        case 2: return launch_llk2_t<2>(grid, a, s);
        case 3: return launch_llk2_t<3>(grid, a, s);
        case 4: return launch_llk2_t<4>(grid, a, s);
        case 5: return launch_llk2_t<5>(grid, a, s);
        case 6: return launch_llk2_t<6>(grid, a, s);
        case 7: return launch_llk2_t<7>(grid, a, s);
        case 8: return launch_llk2_t<8>(grid, a, s);
        case 9: return launch_llk2_t<9>(grid, a, s);
        case 10: return launch_llk2_t<10>(grid, a, s);
#endif
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ppca
