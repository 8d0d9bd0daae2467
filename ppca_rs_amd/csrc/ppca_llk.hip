// ppca_llk.hip -- the log-likelihood sweep (PPCAModel::llk / llks, ppca/src/ppca_model.rs:124-159, on
// OutputCovariance::quadratic_form / covariance_log_det, ppca/src/output_covariance.rs:115-142) as its own kernel for
// the fused shapes (d <= 256, k <= 10).  It is the pass a mixture step runs once per component and per iteration
// (PPCAMix::iterate_with_prior, ppca/src/mix.rs:283-288) and the trainers' metric.
//
// pass_kernel<K, false, ...> serves this request with its general machinery: a 32-sample tile, the solver on 32 of 64
// lanes in every one of four waves, z by two substitutions, and room for the posterior / reconstruction outputs.  The
// llk of a sample needs only  m, |x~|^2, ln det M  and  quad = b^T M^-1 b = |L^-1 b|^2, and with one wave per SIMD
// every instruction of a wave -- scalar ones included -- costs an issue slot of ~4 cycles (measured: the staging of a
// tile took 4.07 cycles per instruction of any kind), so this kernel is built to issue few of them:
//   * TWO tiles per round: P1 / P2 of tile A, P1 / P2 of tile B (the x~ tile is free again after a tile's P2: nothing
//     reads it later in this pass), then ONE solver step for the 64 samples on all 64 lanes of wave 0 -- factorisation
//     + the forward substitution only; the observed counts are popcounts of the mask words, taken there;
//   * the wave's slice of the int8 digit table (8 slices x 4 k-chunks x 16 bytes per lane = 128 registers; there are
//     no statistics accumulators in this pass) is loaded ONCE per workgroup and stays in registers: no table traffic
//     and no load latency inside the tile loop;
//   * staging without scalar work: the finite test |x| < lim (lim = +inf, or -1 for the padding past d) gives the wave
//     mask that is both the select predicate and, as it comes, mask word 2 h + e of the row (qprep orders the table
//     rows to match); a tile's rows are addressed through ONE buffer descriptor with a scalar offset per row (rows
//     past the end read as zeros and are never somebody's sample); the eight per-lane |x~|^2 partials of a wave's rows
//     are summed together (v_permlane32_swap / v_permlane16_swap fold two rows per add, then four DPP steps for the
//     last two registers) instead of one six-step reduction per row;
//   * the next tile's rows are requested as soon as the registers of the previous one are staged, so they travel
//     under a whole P2.
// Per-sample arithmetic is that of pass_kernel except for the ORDER of the |x~|^2 sum (same int8-sliced Gram behind
// the same guard, same Cholesky), so the llks agree with it to rounding.  The guard's fallback is
// pass_kernel<K, false, 4, false>.
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
    static constexpr int OFF_M = OFF_B1 + 2 * B * BS;   // mask words of the round's two tiles, 2 B x 4 u64
    static constexpr int OFF_XX = OFF_M + 2 * B * 4;    // |x~|^2 of the 2 B samples
    static constexpr int LDS_DOUBLES = OFF_XX + 2 * B;
};

template <int K>
__global__ __launch_bounds__(256) void llk2_kernel(PassArgs p) {
    using cfg = CfgL<K>;
    constexpr int KP = cfg::KP, NTP = cfg::NTP, B = cfg::B, XS = cfg::XS, CS = cfg::CS, GS = cfg::GS, BS = cfg::BS;
    constexpr int NW = 4, RPW = B / NW, DPS = cfg::DP / 2, STEPS = DPS / 4;
    static_assert(NTP <= NW && QS == 8 && RPW == 8, "int8 Gram: one wave per packed-column tile, 8 digit slices; 8 rows per wave");
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *Xs = sm + cfg::OFF_X;
    double *Cs = sm + cfg::OFF_C;
    double *Gs = sm + cfg::OFF_G;
    double *B1 = sm + cfg::OFF_B1;
    unsigned long long *Ms = reinterpret_cast<unsigned long long *>(sm + cfg::OFF_M);
    double *xxs = sm + cfg::OFF_XX;

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
    // staging lane map: lane l holds dims 128 h + 2 l + e (element q = 2 h + e) of a row; observed <=> |x| < lim
    double mu[4], lim[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = 128 * (q >> 1) + 2 * lane_entry + (q & 1);
        mu[q] = (j < d) ? mMean[j] : 0.0;
        lim[q] = (j < d) ? __builtin_inf() : -1.0;
    }

    const int64_t ntiles = (n + B - 1) / B;
    const int64_t tiles_per_wg = (ntiles + gridDim.x - 1) / gridDim.x;
    const int64_t tile_begin = (int64_t)blockIdx.x * tiles_per_wg;
    const int64_t tile_end = tile_begin + tiles_per_wg < ntiles ? tile_begin + tiles_per_wg : ntiles;
    const int64_t nleft = n - tile_begin * B;
    const int nrel = (int)(nleft < (1 << 30) ? nleft : (1 << 30));  // rows from the workgroup's first row to the end
    const double *Xwg = p.X + tile_begin * B * p.ldx;
    const int rowbytes = d * (int)sizeof(double);  // rows are contiguous (ldx == d)
#ifdef PPCA_PHASE_TIMING
    long long tph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = clock64();
#define LLK_STAMP(i) { long long tn = clock64(); tph[i] += tn - tlast; tlast = tn; }
#else
#define LLK_STAMP(i)
#endif
    double xr[RPW][4];
    // One descriptor per tile (base = its first row, extent = its real rows): the row is a scalar offset, the lane
    // a constant VGPR, the half an immediate.  Rows past n read as zeros ("observed", but no lane's sample).
    auto load_tile = [&](int64_t tile) {
#ifdef LLK2_DIAG_RESIDENT  // measurement only: every tile re-reads the workgroup's first one (L2-resident: the sweep without HBM)
        const int rel0 = 0;
#else
        const int rel0 = (int)(tile - tile_begin) * B;
#endif
        int cnt = nrel - rel0;
        cnt = __builtin_amdgcn_readfirstlane(cnt < 0 ? 0 : (cnt > B ? B : cnt));  // (keeps the descriptor scalar)
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<double *>(Xwg + (int64_t)rel0 * p.ldx), 0, cnt * rowbytes, 0x00020000);
        const int wbase = wave * RPW * rowbytes;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                typedef unsigned u4_t __attribute__((ext_vector_type(4)));
                const u4_t v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, lane_entry * 16, wbase + r * rowbytes + 1024 * h, 0);
                xr[r][2 * h] = __longlong_as_double(((long long)v[1] << 32) | v[0]);
                xr[r][2 * h + 1] = __longlong_as_double(((long long)v[3] << 32) | v[2]);
            }
        }
    };
    // The wave's slice of the digit table, resident: fragment (slice sl, k-chunk kc) = 16 bytes per lane.  Waves
    // without a column tile (k' <= 48) read past the table (zeros), run the same MFMAs and skip only the store.
    const bool gram_wave = NTP >= NW || wave < NTP;
    i4_t qt[QS][4];
    {
        const __amdgpu_buffer_rsrc_t qrsrc = __builtin_amdgcn_make_buffer_rsrc(p.qtab, 0, (int)qtab_bytes<K>(), 0x00020000);
        const int qbase = wave * QS * 4 * 1024;
#pragma unroll
        for (int sl = 0; sl < QS; ++sl)
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                typedef unsigned u4_t __attribute__((ext_vector_type(4)));
                const u4_t v = __builtin_amdgcn_raw_buffer_load_b128(qrsrc, lane_entry * 16, qbase + (sl * 4 + kc) * 1024, 0);
                qt[sl][kc] = i4_t{(int)v[0], (int)v[1], (int)v[2], (int)v[3]};
            }
    }
    const double qs = gram_wave ? p.qscale[16 * wave + (lane_entry & 15)] : 0.0;

    // ---- P1: stage one tile (slot 0 / 1 of the round): x~ into the LDS tile, mask words, |x~|^2 per sample
    auto stage_tile = [&](int lane, int slot) {
        int st_wlo = 0, st_whi = 0;
        double pxx[RPW];
        static_for<RPW>([&](auto r_tag) {
            constexpr int r = decltype(r_tag)::value;
            const int ri = wave * RPW + r;
            double xt[4];
            static_for<4>([&](auto q_tag) {
                constexpr int q = decltype(q_tag)::value;
                const double v = xr[r][q];
                const bool ob = __builtin_fabs(v) < lim[q];  // finite (dataset.rs:19-22) and a real dimension
                xt[q] = ob ? v - mu[q] : 0.0;                // select, never multiply (utils.rs:118-127)
                // mask word q of the row = this ballot as it comes (bit l <-> dim 128 h + 2 l + e), into lane 4 r + q
                writelane_mask<4 * r + q>(st_wlo, st_whi, __builtin_amdgcn_ballot_w64(ob));
            });
            typedef double d2_t __attribute__((ext_vector_type(2)));
            *reinterpret_cast<d2_t *>(Xs + ri * XS + 2 * lane) = d2_t{xt[0], xt[1]};
            *reinterpret_cast<d2_t *>(Xs + ri * XS + 128 + 2 * lane) = d2_t{xt[2], xt[3]};
            pxx[r] = xt[0] * xt[0] + xt[1] * xt[1] + xt[2] * xt[2] + xt[3] * xt[3];
        });
        const unsigned long long myw = ((unsigned long long)(unsigned)st_whi << 32) | (unsigned)st_wlo;
        if (lane < 4 * RPW) Ms[(slot * B + wave * RPW) * 4 + lane] = myw;
        store_row_sums(pxx, lane, xxs + slot * B + wave * RPW);
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
        // A = mask bytes: lane (sample = 16 rt2 + l15, k = 64 kc + 16 l4 .. +15 of word kc); 4 bits -> 4 bytes by
        // one multiply: (x * 0x204081) & 0x01010101 puts bit i of x into byte i
        i4_t af[2][4];
        unsigned long long mwd[2][4];
#pragma unroll
        for (int rt2 = 0; rt2 < 2; ++rt2)
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) mwd[rt2][kc] = Ms[(slot * B + 16 * rt2 + l15) * 4 + kc];
        {
            // b = X~ C first: its operands are requested four k-steps ahead, and the mask words arrive under it
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
        LLK_STAMP(4)
#pragma unroll
        for (int rt2 = 0; rt2 < 2; ++rt2)
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                const unsigned bits = (unsigned)(mwd[rt2][kc] >> (16 * l4)) & 0xFFFFu;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    af[rt2][kc][u] = (int)((((bits >> (4 * u)) & 0xFu) * 0x00204081u) & 0x01010101u);
            }
        // digit pairs high to low: exact integer sums (|sum| <= 2^14 per digit, two digits per i32), folded into the
        // running fp64 value by Horner in 128^2
        double v[2][4];
#pragma unroll
        for (int g = 0; g < QS / 2; ++g) {
            const int sl = QS - 2 - 2 * g;
#pragma unroll
            for (int rt2 = 0; rt2 < 2; ++rt2) {
                i4_t ia[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    ia[u] = i4_t{0, 0, 0, 0};
#pragma unroll
                    for (int kc = 0; kc < 4; ++kc)
                        ia[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[rt2][kc], qt[sl + u][kc], ia[u], 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int part = ia[1][r] * QBASE + ia[0][r];
                    v[rt2][r] = g == 0 ? (double)part : v[rt2][r] * (double)(QBASE * QBASE) + (double)part;
                }
            }
        }
        LLK_STAMP(5)
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
        LLK_STAMP(7)
        stage_tile(lane, 0);
        load_tile(tile + 1);  // travels under the contraction below
        LLK_STAMP(0)
        __syncthreads();
        LLK_STAMP(1)
        contract_tile(lane, 0);
        LLK_STAMP(2)
        __syncthreads();
        LLK_STAMP(1)
        stage_tile(lane, 1);
        load_tile(tile + 2);
        LLK_STAMP(0)
        __syncthreads();
        LLK_STAMP(1)
        contract_tile(lane, 1);
        LLK_STAMP(2)
        __syncthreads();
        LLK_STAMP(1)
        if (wave == 0) {
            // ---- P3: lane i < 32 -> sample i of tile, lane i >= 32 -> sample i - 32 of tile + 1
            const int slot = lane >> 5, i = lane & (B - 1);
            const int64_t t = tile + slot;
            const int64_t row = t * B + i;
            const bool mine = t < tile_end && row < n;
            const double *g0 = Gs + lane * GS;
            const double *b1 = B1 + lane * BS;
            const double wgt = mine ? (p.w ? p.w[row] : 1.0) : 0.0;
            const unsigned long long *mw = Ms + lane * 4;
            const int m = __popcll(mw[0]) + __popcll(mw[1]) + __popcll(mw[2]) + __popcll(mw[3]);
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
        LLK_STAMP(3)
        __syncthreads();
    }
#ifdef PPCA_PHASE_TIMING
    if (p.dbg && tid == 0)
        for (int i = 0; i < 16; ++i) p.dbg[(int64_t)blockIdx.x * 16 + i] = (double)tph[i];
#endif
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

// ---------------------------------------------------------------------------------------------------------------------
// llk8_kernel -- the same sweep as an EIGHT-wave workgroup (two waves per SIMD, 256 registers each; round 4).  One wave per
// SIMD issues a vector instruction every ~5-8 cycles (dependent chains, LDS waits); a second wave fills those slots.  The
// work of a tile is split eight ways instead of four: four staged rows per wave; the int8 Gram as (packed-column tile, row
// tile) = 4 x 2 wave units, each with ITS column tile's digit-table slice resident (128 registers, as in llk2_kernel); b = X~ C
// as (row tile, quarter of the dimensions) = 2 x 4 units, the four K-split partials summed by the solver in a fixed order.
// The solver step (64 samples of the round's two tiles on the 64 lanes of wave 0) is unchanged.
template <int K>
struct CfgL8 {
    using c = Cfg<K>;
    static constexpr int KP = c::KP, NTP = c::NTP, B = c::B, DP = c::DP, XS = c::XS, CS = c::CS;
    static constexpr int GS = 16 * NTP + 16 + 1;  // [G (16 NTP) | b partial of dims 0-63 (16)], 2 B rows
    static constexpr int BS = 17;                 // b partials of dims 64-127, 128-191, 192-255: 3 x 2 B rows
    static constexpr int OFF_X = 0;
    static constexpr int OFF_C = OFF_X + B * XS;
    static constexpr int OFF_G = OFF_C + DP * CS;
    static constexpr int OFF_B1 = OFF_G + 2 * B * GS;
    // mask words and |x~|^2 of the round's two tiles -- and a SECOND copy for the round's first tile (block 2): the first tile of round
    // r + 1 is staged while the solver step of round r still reads round r's (round 6); block of (slot, parity): see stage_tile
    static constexpr int OFF_M = OFF_B1 + 3 * 2 * B * BS;  // 3 B x 4 u64
    static constexpr int OFF_XX = OFF_M + 3 * B * 4;       // 3 B
    static constexpr int OFF_MU = OFF_XX + 3 * B;          // the mean (DP doubles, zero past d): re-read by the staging of every tile
    static constexpr int OFF_FLAG = OFF_MU + DP;           // hand-off counter of the solver wave's rows (llk8_run)
    static constexpr int LDS_DOUBLES = OFF_FLAG + 2;
    static_assert(LDS_DOUBLES * 8 <= 160 * 1024, "LDS budget (llk8)");
};

// Four per-lane partial sums (one register per row of a wave's four staged rows) -> the four row totals into out[0..3]
// (store_row_sums for four rows): halves, then 16-lane rows, then four DPP steps.
// only < 0: all four rows are stored; otherwise only row `only` (the others' registers hold zeros: llk8_run's hand-off of the solver
// wave's rows, one to each of four other waves -- the SAME tree of additions for the row as when its own wave sums it with its three
// neighbours, so a sample's |x~|^2 does not depend on who staged it)
__device__ __forceinline__ void store_row_sums4(const double (&pxx)[4], int lane, double *out, int only = -1) {
    const double u0 = fold_halves(pxx[0], pxx[1]);  // lanes 0-31: row 0, lanes 32-63: row 1
    const double u1 = fold_halves(pxx[2], pxx[3]);  // lanes 0-31: row 2, lanes 32-63: row 3
    double v = fold_rows(u0, u1);                   // 16-lane row rho holds data row: 0 -> 0, 1 -> 2, 2 -> 1, 3 -> 3
    v += dpp_f64<0xB1, 0xF>(v);
    v += dpp_f64<0x4E, 0xF>(v);
    v += dpp_f64<0x141, 0xF>(v);
    v += dpp_f64<0x140, 0xF>(v);
    if ((lane & 15) == 0) {
        const int rho = lane >> 4, row = 2 * (rho & 1) + (rho >> 1);
        if (only < 0 || row == only) out[row] = v;
    }
}

// The sweep over the tiles [tile_begin, tile_end) of p's rows by ONE workgroup (the body of llk8_kernel; mix_llk8_kernel below walks
// several (model, run of tiles) units with it).  scal: where the workgroup's scalars go (nullable).
// OUT (round 6): 0 the llk sweep; 1 / 2 the N x d output passes PPCAModel::smooth / extrapolate (ppca_model.rs:237-261 on :454-463:
// C z + mean for every dimension / for the masked ones, the observed values passed through) on the same sweep -- the solver step also
// runs the back substitution (z = M^-1 b into the sample's dead b-partial slots), then every wave forms eight of the round's 64 output
// rows with the lane map of the staging (lane l: dims 128 h + 2 l, + 1: its two rows of C in registers, z broadcast from LDS, K
// multiply-adds per element) and stores them as whole 16-byte pieces, 1 KB per instruction; extrapolate re-reads the rows (L2: they
// were staged a round ago) and selects by the mask words still in LDS -- bit-exact pass-through.  The waves drop their table slices
// for that phase and request them again behind it, as the solver wave always did.  Needs an even d (16-byte pieces).  The four-wave
// pass_kernel<K, false> (5.3 / 5.7 ms at N = 4 M) formed the outputs on the fp64 MFMA, moved them through LDS and stored 8 bytes per lane.
template <int K, int OUT = 0>
__device__ __forceinline__ void llk8_run(const PassArgs &p, double *sm, const int64_t tile_begin, const int64_t tile_end, double *scal) {
    using cfg = CfgL8<K>;
    constexpr int KP = cfg::KP, NTP = cfg::NTP, B = cfg::B, XS = cfg::XS, CS = cfg::CS, GS = cfg::GS, BS = cfg::BS;
    constexpr int NW = 8, RPW = B / NW, DPQ = cfg::DP / 4, STEPS = DPQ / 4;
    static_assert(NTP <= 4 && QS == 8 && RPW == 4, "int8 Gram: (column tile, row tile) wave units, 8 digit slices; 4 rows per wave");
    double *Xs = sm + cfg::OFF_X;
    double *Cs = sm + cfg::OFF_C;
    double *Gs = sm + cfg::OFF_G;
    double *B1 = sm + cfg::OFF_B1;
    unsigned long long *Ms = reinterpret_cast<unsigned long long *>(sm + cfg::OFF_M);
    double *xxs = sm + cfg::OFF_XX;

    const int tid = threadIdx.x, lane_entry = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int d = p.d;
    const int64_t n = p.n;
    const double *mC = p.model + MODEL_HDR;
    const double *mMean = mC + (int64_t)d * K;
    const double s2 = p.model[1], lnsig = p.model[2];
    for (int idx = tid; idx < cfg::DP * CS; idx += 512) {
        int j = idx / CS, a = idx - j * CS;
        Cs[idx] = (j < d && a < K) ? mC[(int64_t)j * K + a] : 0.0;
    }
    for (int idx = tid; idx < cfg::DP; idx += 512) sm[cfg::OFF_MU + idx] = idx < d ? mMean[idx] : 0.0;
    unsigned *hand = reinterpret_cast<unsigned *>(sm + cfg::OFF_FLAG);
    if (tid == 0) *hand = 0u;

    const int64_t nleft = n - tile_begin * B;
    const int nrel = (int)(nleft < (1 << 30) ? nleft : (1 << 30));
    const double *Xwg = p.X + tile_begin * B * p.ldx;
    const int rowbytes = d * (int)sizeof(double);
    double xr[RPW][4];
    auto load_tile = [&](int64_t tile) {
        const int rel0 = (int)(tile - tile_begin) * B;
        int cnt = nrel - rel0;
        cnt = __builtin_amdgcn_readfirstlane(cnt < 0 ? 0 : (cnt > B ? B : cnt));
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<double *>(Xwg + (int64_t)rel0 * p.ldx), 0, cnt * rowbytes, 0x00020000);
        const int wbase = wave * RPW * rowbytes;
#pragma unroll
        for (int r = 0; r < RPW; ++r) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                typedef unsigned u4_t __attribute__((ext_vector_type(4)));
                const u4_t v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, lane_entry * 16, wbase + r * rowbytes + 1024 * h, 0);
                xr[r][2 * h] = __longlong_as_double(((long long)v[1] << 32) | v[0]);
                xr[r][2 * h + 1] = __longlong_as_double(((long long)v[3] << 32) | v[2]);
            }
        }
    };
    // (round 6) One dword of every 128-byte line of the wave's four rows of a tile, a round before the rows themselves are requested:
    // since the first tile of round r + 1 is staged in round r's solver phase its rows are wanted a solver step EARLIER than before,
    // and the register set that receives them is free only one contraction ahead -- less than an HBM round trip (the phase table
    // showed wave 0 standing ~4 k cycles at its hand-off).  The touch brings the lines into the XCD's L2 (one instruction, one
    // register whose value only keeps the load alive); the 16-byte requests that follow are L2 hits.
    int pf_acc = 0;
    auto touch_tile = [&](int64_t tile) {
        const int rel0 = (int)(tile - tile_begin) * B;
        int cnt = nrel - rel0;
        cnt = __builtin_amdgcn_readfirstlane(cnt < 0 ? 0 : (cnt > B ? B : cnt));
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<double *>(Xwg + (int64_t)rel0 * p.ldx), 0, cnt * rowbytes, 0x00020000);
        pf_acc |= (int)__builtin_amdgcn_raw_buffer_load_b32(xrsrc, lane_entry * 128, wave * RPW * rowbytes, 0);
    };
    // Contraction units.  LLK8_ROLES = 0 (rounds 4-5): every wave one Gram unit (packed-column tile ct, row tile rt) + one b unit (row
    // tile rtb, quarter kq of the dimensions).  LLK8_ROLES = 1 (round 6): waves 0-3 the Gram of column tile ct = wave for BOTH row tiles
    // (one table slice, two units), waves 4-7 b of row tile rtb for one HALF of the dimensions (two of the old units in one loop: two
    // K-split partials instead of four).  The same MFMAs per SIMD (wave w and w + 4 share one); what it buys: the solver step runs on a
    // b wave (wave 4), which holds no table slice -- the solver wave of rounds 4-5 dropped its slice for the solver step and requested
    // the 32 KB again behind it, ~2 k cycles of request issue on the round's critical path (phase table of the LLK8_TIMING build).
#ifndef LLK8_ROLES
#define LLK8_ROLES 1
#endif
    constexpr bool ROLES = LLK8_ROLES != 0;
    constexpr int SOLVER = ROLES ? 4 : 0;
    const int ct = wave & 3, rt = wave >> 2;
    const int rtb = wave & 1, kq = wave >> 1;
    const bool is_gram = !ROLES || wave < 4, is_b = !ROLES || wave >= 4;
    const bool gram_wave = is_gram && ct < NTP;
    // the unit's slice of the digit table, resident (128 registers).  The solver wave cannot hold it next to the packed
    // factor (110 registers): it drops the slice for its solver step and requests it again right after (32 loads from
    // L2 that travel under the next staging) -- a spill through scratch would wait on the row loads in flight instead.
    i4_t qt[QS][4];
    auto load_table = [&]() {
        const __amdgpu_buffer_rsrc_t qrsrc = __builtin_amdgcn_make_buffer_rsrc(p.qtab, 0, (int)qtab_bytes<K>(), 0x00020000);
        const int qbase = ct * QS * 4 * 1024;  // (a tile past the table reads zeros)
#pragma unroll
        for (int sl = 0; sl < QS; ++sl)
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                typedef unsigned u4_t __attribute__((ext_vector_type(4)));
                const u4_t v = __builtin_amdgcn_raw_buffer_load_b128(qrsrc, lane_entry * 16, qbase + (sl * 4 + kc) * 1024, 0);
                qt[sl][kc] = i4_t{(int)v[0], (int)v[1], (int)v[2], (int)v[3]};
            }
    };
    // The slice's registers declared dead (no instruction): hipcc's liveness is not path-sensitive -- it does not know that the wave in
    // the solver branch (wave 4) is never a Gram wave, and kept the 128 registers alive across the solver's 110 (hundreds of spills).
    auto kill_table = [&]() {
#pragma unroll
        for (int sl = 0; sl < QS; ++sl)
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) asm volatile("" : "=v"(qt[sl][kc]));
    };
    if (is_gram) load_table();
    else kill_table();
    const double qs = gram_wave ? p.qscale[16 * ct + (lane_entry & 15)] : 0.0;

    auto stage_tile = [&](int lane, int slot, int par) {
        const int mr = slot == 1 ? B : (par ? 2 * B : 0);  // block of the tile's mask words / |x~|^2
        // the lane's four means and limits (observed <=> |x| < lim: +inf for a real dimension, -1 past d), rebuilt per tile from
        // the LDS copy of the mean: as loop invariants they would sit in 16 registers across the solver step
        double mu[4], lim[4];
        {
            typedef double d2_t __attribute__((ext_vector_type(2)));
            const d2_t m0 = *reinterpret_cast<const d2_t *>(sm + cfg::OFF_MU + 2 * lane);
            const d2_t m1 = *reinterpret_cast<const d2_t *>(sm + cfg::OFF_MU + 128 + 2 * lane);
            mu[0] = m0[0]; mu[1] = m0[1]; mu[2] = m1[0]; mu[3] = m1[1];
#pragma unroll
            for (int q = 0; q < 4; ++q) lim[q] = (128 * (q >> 1) + 2 * lane + (q & 1) < d) ? __builtin_inf() : -1.0;
        }
        int st_wlo = 0, st_whi = 0;
        double pxx[RPW];
        static_for<RPW>([&](auto r_tag) {
            constexpr int r = decltype(r_tag)::value;
            const int ri = wave * RPW + r;
            double xt[4];
            static_for<4>([&](auto q_tag) {
                constexpr int q = decltype(q_tag)::value;
                const double v = xr[r][q];
                const bool ob = __builtin_fabs(v) < lim[q];
                xt[q] = ob ? v - mu[q] : 0.0;
                writelane_mask<4 * r + q>(st_wlo, st_whi, __builtin_amdgcn_ballot_w64(ob));
            });
            typedef double d2_t __attribute__((ext_vector_type(2)));
            *reinterpret_cast<d2_t *>(Xs + ri * XS + 2 * lane) = d2_t{xt[0], xt[1]};
            *reinterpret_cast<d2_t *>(Xs + ri * XS + 128 + 2 * lane) = d2_t{xt[2], xt[3]};
            pxx[r] = xt[0] * xt[0] + xt[1] * xt[1] + xt[2] * xt[2] + xt[3] * xt[3];
        });
        const unsigned long long myw = ((unsigned long long)(unsigned)st_whi << 32) | (unsigned)st_wlo;
        if (lane < 4 * RPW) Ms[(mr + wave * RPW) * 4 + lane] = myw;
        store_row_sums4(pxx, lane, xxs + mr + wave * RPW);
    };
    auto contract_tile = [&](int lane, int slot, int par) {
        const int mr = slot == 1 ? B : (par ? 2 * B : 0);
        const int l15 = lane & 15, l4 = lane >> 4;
        // b = X~ C on v_mfma_f64_4x4x4 (four independent 4 x 4 x 4 blocks per instruction: block b = lane bits 2-3 = samples
        // 4 b .. 4 b + 3 of the row tile; A[i][k] in lane 16 k + 4 b + i, B[k][j] in lane 16 k + 4 b + j, D[i][j] in lane
        // 16 i + 4 b + j).  The part issues it every 17.3 cycles for 512 flop against 105 for the 2 048 of v_mfma_f64_16x16x4
        // (profiles/r04/mfma_peak.txt), and K = 10 columns are three groups of 4 instead of one padded tile of 16; the A operand
        // is the one the 16x16x4 form read, the B operand three LDS reads per k-step (the same address in the four blocks).
        constexpr int NCB = (K + 3) / 4;
        if (is_b) {
            // the unit: row tile rb, dimensions [dim0, dim0 + 4 NST); its partial of b goes to the row's b slots (part 0) or to B1
            const int rb = rtb;
            const int part = ROLES ? (wave - 4) >> 1 : kq;
            constexpr int NST = ROLES ? 2 * STEPS : STEPS;
            const int dim0 = (ROLES ? 2 * DPQ : DPQ) * part;
            double accb[NCB];
            int ccol[NCB];
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
                accb[c] = 0.0;
                ccol[c] = 4 * c + (lane & 3) < K ? 4 * c + (lane & 3) : K;  // (column K of the tile of C is zero)
            }
            const double *xrow = Xs + (16 * rb + l15) * XS + dim0 + l4;
            const double *cpc = Cs + (dim0 + l4) * CS;
            constexpr int CH = 2;
            double axb[2][CH], cbb[2][CH][NCB];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                axb[0][u] = xrow[4 * u];
#pragma unroll
                for (int cc = 0; cc < NCB; ++cc) cbb[0][u][cc] = cpc[4 * u * CS + ccol[cc]];
            }
#pragma unroll
            for (int c = 0; c < NST / CH; ++c) {
                if (c + 1 < NST / CH) {
#pragma unroll
                    for (int u = 0; u < CH; ++u) {
                        axb[(c + 1) & 1][u] = xrow[4 * ((c + 1) * CH + u)];
#pragma unroll
                        for (int cc = 0; cc < NCB; ++cc) cbb[(c + 1) & 1][u][cc] = cpc[4 * ((c + 1) * CH + u) * CS + ccol[cc]];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < CH; ++u)
#pragma unroll
                    for (int cc = 0; cc < NCB; ++cc)
                        accb[cc] = __builtin_amdgcn_mfma_f64_4x4x4f64(axb[c & 1][u], cbb[c & 1][u][cc], accb[cc], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            // the K-split partials of b, summed by the solver in a fixed order
            const int row = slot * B + 16 * rb + 4 * ((lane >> 2) & 3) + l4;  // D[i][j] of block b: lane 16 i + 4 b + j
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
                const int col = 4 * c + (lane & 3);
                if (part == 0) Gs[row * GS + 16 * NTP + col] = accb[c];
                else B1[((part - 1) * 2 * B + row) * BS + col] = accb[c];
            }
        }
        if (is_gram) {
            // Gram units of this wave: column tile ct, row tile rg
            static_for<ROLES ? 2 : 1>([&](auto u_tag) {
                const int rg = ROLES ? decltype(u_tag)::value : rt;
                unsigned long long mwd[4];
#pragma unroll
                for (int kc = 0; kc < 4; ++kc) mwd[kc] = Ms[(mr + 16 * rg + l15) * 4 + kc];
                i4_t af[4];
#pragma unroll
                for (int kc = 0; kc < 4; ++kc) {
                    const unsigned bits = (unsigned)(mwd[kc] >> (16 * l4)) & 0xFFFFu;
#pragma unroll
                    for (int u = 0; u < 4; ++u) af[kc][u] = (int)((((bits >> (4 * u)) & 0xFu) * 0x00204081u) & 0x01010101u);
                }
                double v[4];
#pragma unroll
                for (int g = 0; g < QS / 2; ++g) {
                    const int sl = QS - 2 - 2 * g;
                    i4_t ia[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        ia[u] = i4_t{0, 0, 0, 0};
#pragma unroll
                        for (int kc = 0; kc < 4; ++kc) ia[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[kc], qt[sl + u][kc], ia[u], 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int part = ia[1][r] * QBASE + ia[0][r];
                        v[r] = g == 0 ? (double)part : v[r] * (double)(QBASE * QBASE) + (double)part;
                    }
                }
                if (gram_wave) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) Gs[(slot * B + 16 * rg + 4 * l4 + r) * GS + 16 * ct + l15] = v[r] * qs;
                }
            });
        }
    };

    // One row of the x~ tile that holds RAW values (the solver wave's hand-off): centred, masked and filed exactly as stage_tile does
    // it for the row's own wave; r4 = the row's place among that wave's four.
    auto stage_one = [&](int lane, int ri, int r4, int mr) {
        typedef double d2_t __attribute__((ext_vector_type(2)));
        const d2_t m0 = *reinterpret_cast<const d2_t *>(sm + cfg::OFF_MU + 2 * lane);
        const d2_t m1 = *reinterpret_cast<const d2_t *>(sm + cfg::OFF_MU + 128 + 2 * lane);
        const double mu[4] = {m0[0], m0[1], m1[0], m1[1]};
        const d2_t v0 = *reinterpret_cast<const d2_t *>(Xs + ri * XS + 2 * lane);
        const d2_t v1 = *reinterpret_cast<const d2_t *>(Xs + ri * XS + 128 + 2 * lane);
        const double v[4] = {v0[0], v0[1], v1[0], v1[1]};
        double xt[4];
        unsigned long long bal[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double lim = (128 * (q >> 1) + 2 * lane + (q & 1) < d) ? __builtin_inf() : -1.0;
            const bool ob = __builtin_fabs(v[q]) < lim;
            xt[q] = ob ? v[q] - mu[q] : 0.0;
            bal[q] = __builtin_amdgcn_ballot_w64(ob);
        }
        *reinterpret_cast<d2_t *>(Xs + ri * XS + 2 * lane) = d2_t{xt[0], xt[1]};
        *reinterpret_cast<d2_t *>(Xs + ri * XS + 128 + 2 * lane) = d2_t{xt[2], xt[3]};
        if (lane < 4) Ms[(mr + ri) * 4 + lane] = lane == 0 ? bal[0] : lane == 1 ? bal[1] : lane == 2 ? bal[2] : bal[3];
        const double px = xt[0] * xt[0] + xt[1] * xt[1] + xt[2] * xt[2] + xt[3] * xt[3];
        const double pxx[4] = {r4 == 0 ? px : 0.0, r4 == 1 ? px : 0.0, r4 == 2 ? px : 0.0, r4 == 3 ? px : 0.0};
        store_row_sums4(pxx, lane, xxs + mr + (ri - r4), r4);
    };
    unsigned handed = 0u;  // rounds whose solver wave has handed its rows of the next first tile over

    // Round r: [its first tile was staged during round r - 1's solver step] contract A, stage B, contract B, then the solver step of
    // the round's 64 samples on wave 0 WHILE the other seven waves stage the first tile of round r + 1 (wave 0 stages its four rows
    // behind its solver step).  Until round 6 the other waves stood at the barrier during the solver step (a sixth of the round by
    // ablation, LLK8_EXP_NOSOLVE: 2.36 -> 1.97 ms at N = 4 M) and everybody staged the first tile afterwards (another sixth,
    // LLK8_EXP_NOSTAGE0: 1.98 ms).
#ifdef LLK8_TIMING  // (diagnostic build: per-phase cycle sums of wave 0 -- and the overlapped staging of wave 1 -- into the scalars)
    long long tq[6] = {0, 0, 0, 0, 0, 0}, tl = clock64();
#define L8_STAMP(i) { __builtin_amdgcn_sched_barrier(0); const long long tn = clock64(); tq[i] += tn - tl; tl = tn; __builtin_amdgcn_sched_barrier(0); }
#else
#define L8_STAMP(i)
#endif
    double run_llk = 0.0, run_w = 0.0;
    int par = 0;  // which copy of the first tile's mask words / |x~|^2 the round reads
    if (tile_begin < tile_end) load_tile(tile_begin);
    __syncthreads();
    if (tile_begin < tile_end) {
        stage_tile(lane_entry, 0, par);
        load_tile(tile_begin + 1);
    }
    __syncthreads();
    for (int64_t tile = tile_begin; tile < tile_end; tile += 2) {
        int lane = lane_entry;
        asm volatile("" : "+v"(lane));
        L8_STAMP(5)
#ifndef LLK8_NO_TOUCH
        touch_tile(tile + 2);
        touch_tile(tile + 3);
#endif
        contract_tile(lane, 0, par);
        L8_STAMP(0)
        __syncthreads();
        L8_STAMP(4)
        stage_tile(lane, 1, par);
        load_tile(tile + 2);
        L8_STAMP(1)
        __syncthreads();
        L8_STAMP(4)
        contract_tile(lane, 1, par);
        L8_STAMP(0)
        __syncthreads();
        L8_STAMP(4)
        const bool more = tile + 2 < tile_end;  // (wave-uniform)
        if (wave == SOLVER) {
            if (more) {  // this wave's four rows of the next round's first tile: RAW into the (free) x~ tile, one for each of four other waves
                typedef double d2_t __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int r = 0; r < RPW; ++r) {
                    *reinterpret_cast<d2_t *>(Xs + (SOLVER * RPW + r) * XS + 2 * lane) = d2_t{xr[r][0], xr[r][1]};
                    *reinterpret_cast<d2_t *>(Xs + (SOLVER * RPW + r) * XS + 128 + 2 * lane) = d2_t{xr[r][2], xr[r][3]};
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(hand, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                load_tile(tile + 3);
            }
#ifdef LLK8_EXP_NOSOLVE  // (timing experiment, results wrong: the round without the arithmetic of its one-wave solver step)
          if (p.d < 0) {
#else
          {
#endif
            const int slot = lane >> 5, i = lane & (B - 1);
            const int mrs = slot == 1 ? B : (par ? 2 * B : 0);
            const int64_t t = tile + slot;
            const int64_t row = t * B + i;
            const bool mine = t < tile_end && row < n;
            const double *g0 = Gs + lane * GS;
            const double *b1 = B1 + lane * BS;
            const double wgt = mine ? (p.w ? p.w[row] : 1.0) : 0.0;
            const unsigned long long *mw = Ms + (mrs + i) * 4;
            const int m = __popcll(mw[0]) + __popcll(mw[1]) + __popcll(mw[2]) + __popcll(mw[3]);
            const double xx = xxs[mrs + i];
            Posterior<K> post;
            double pm;
            int pe;
            post.factor([&](int e) { return g0[e]; }, s2, pm, pe);
            if constexpr (OUT == 0) {
                const double quad = post.forward_quad([&](int a) { return ROLES ? g0[16 * NTP + a] + b1[a] : ((g0[16 * NTP + a] + b1[a]) + b1[2 * B * BS + a]) + b1[4 * B * BS + a]; });
                const double lk = sample_llk(xx, quad, lean_log(pm) + (double)pe * LN_2, s2, lnsig, m, K);
                run_llk += wgt * lk;
                run_w += wgt;
                if (p.llks && mine) p.llks[row] = lk;
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!ROLES) load_table();
                else kill_table();  // (LLK8_ROLES = 1: the solver wave is a b wave and holds no slice)
            } else {
                double z[K], quad, zz;
                post.solve([&](int a) { return ROLES ? g0[16 * NTP + a] + b1[a] : ((g0[16 * NTP + a] + b1[a]) + b1[2 * B * BS + a]) + b1[4 * B * BS + a]; }, z, quad, zz);
                double *zr = B1 + lane * BS;  // (the first b partial of the sample is dead: its z goes there)
#pragma unroll
                for (int a = 0; a < K; ++a) zr[a] = z[a];
                (void)xx; (void)m; (void)wgt; (void)pm; (void)pe;
                if constexpr (ROLES) kill_table();
            }
          }
        }
#ifndef LLK8_EXP_NOSTAGE0  // (timing experiment, results wrong: the rounds without the staging of their first tile)
        if (more) {  // the first tile of the next round (its rows have been in registers since this round's second staging)
            ++handed;
            if (wave != SOLVER) {
                stage_tile(lane, 0, par ^ 1);
                load_tile(tile + 3);
#ifndef LLK8_HAND_WAVES
#define LLK8_HAND_WAVES (LLK8_ROLES ? 0x1765 : 0x4321)  // nibble r: the wave that stages row r of the solver wave (its SIMD's other wave is spared)
#endif
                const int hrow = wave == ((LLK8_HAND_WAVES >> 0) & 15) ? 0 : wave == ((LLK8_HAND_WAVES >> 4) & 15) ? 1
                               : wave == ((LLK8_HAND_WAVES >> 8) & 15) ? 2 : wave == ((LLK8_HAND_WAVES >> 12) & 15) ? 3 : -1;
                if (hrow >= 0) {  // ... and one of the solver wave's rows
                    for (;;) {
                        const unsigned seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(hand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                        if ((int)(seen - handed) >= 0) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                    asm volatile("" ::: "memory");
                    stage_one(lane, SOLVER * RPW + hrow, hrow, par ? 0 : 2 * B);
                }
            }
        }
#endif
        L8_STAMP(2)
        if constexpr (OUT != 0) {
            __syncthreads();
            // (a fresh opaque copy of the lane index: everything the phase derives from it -- addresses, the re-read of the rows -- is computed
            //  HERE, not hoisted above the barrier into the contraction, whose registers are full: hipcc did, and spilled the next tile's rows)
            int lo = lane_entry;
            asm volatile("" : "+v"(lo));
            typedef double d2_t __attribute__((ext_vector_type(2)));
            typedef unsigned u4_t __attribute__((ext_vector_type(4)));
            constexpr int NR = 2 * RPW;  // rows of the round per wave
            // Order of the phase: [extrapolate: the re-read of the wave's rows] -> the 2 x NR output pieces into registers -> the
            // stores -> the Gram waves' request of their table slices (requested in front of the stores -- so that the slices do
            // not wait for the stores' acknowledgement on the one in-order counter -- measured the same and, with the contraction
            // split by roles, no longer fits the registers).
            u4_t ov[NR][2];
            if constexpr (OUT == 2) {
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int si = NR * wave + r;
                    const int64_t row = (tile + (si >> 5)) * B + (si & (B - 1));
                    const __amdgpu_buffer_rsrc_t xr1 = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<double *>(p.X + (row < n ? row : n - 1) * p.ldx), 0, rowbytes, 0x00020000);
                    ov[r][0] = __builtin_amdgcn_raw_buffer_load_b128(xr1, lo * 16, 0, 0);
                    ov[r][1] = __builtin_amdgcn_raw_buffer_load_b128(xr1, lo * 16, 1024, 0);
                }
            }
            // the halves h in [H0, H1) of the wave's NR rows: smooth takes both at once (z read once per row); extrapolate one after the
            // other (its 2 x NR re-read pieces are in registers meanwhile: both halves' rows of C beside them do not fit)
            auto form = [&](auto h0_tag, auto h1_tag) {
                constexpr int H0 = decltype(h0_tag)::value, H1 = decltype(h1_tag)::value;
                double c[2][2][K];
                d2_t mu2[2];
#pragma unroll
                for (int h = H0; h < H1; ++h) {
                    const double *cr = Cs + (128 * h + 2 * lo) * CS;
#pragma unroll
                    for (int a = 0; a < K; ++a) {
                        c[h][0][a] = cr[a];
                        c[h][1][a] = cr[CS + a];
                    }
                    mu2[h] = *reinterpret_cast<const d2_t *>(sm + cfg::OFF_MU + 128 * h + 2 * lo);
                }
                int chain = 0;
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int si = NR * wave + r;  // sample of the round: slot si >> 5, row si & 31 of its tile
                    // (the row's z address goes through an opaque statement that also takes the previous row's result: hipcc otherwise
                    //  requests the z of all eight rows at once -- 160 registers -- and spills the table slice it is about to reload)
                    int zo = si * BS;
                    asm volatile("" : "+v"(zo) : "v"(chain));
                    const double *zr = B1 + zo;
                    double o[2][2];
#pragma unroll
                    for (int h = H0; h < H1; ++h) {
                        o[h][0] = mu2[h][0];
                        o[h][1] = mu2[h][1];
                    }
#pragma unroll
                    for (int a = 0; a < K; ++a) {
                        const double za = zr[a];  // (one address for the wave: a broadcast read)
#pragma unroll
                        for (int h = H0; h < H1; ++h) {
                            o[h][0] = __builtin_fma(za, c[h][0][a], o[h][0]);
                            o[h][1] = __builtin_fma(za, c[h][1][a], o[h][1]);
                        }
                    }
#pragma unroll
                    for (int h = H0; h < H1; ++h) {
                        if constexpr (OUT == 2) {
                            const int mro = (si >> 5) ? B : (par ? 2 * B : 0);
                            const unsigned long long w0 = Ms[(mro + (si & (B - 1))) * 4 + 2 * h], w1 = Ms[(mro + (si & (B - 1))) * 4 + 2 * h + 1];
                            const double x0 = __longlong_as_double(((long long)ov[r][h][1] << 32) | ov[r][h][0]);
                            const double x1 = __longlong_as_double(((long long)ov[r][h][3] << 32) | ov[r][h][2]);
                            o[h][0] = ((w0 >> lo) & 1ull) ? x0 : o[h][0];
                            o[h][1] = ((w1 >> lo) & 1ull) ? x1 : o[h][1];
                        }
                        const long long b0 = __double_as_longlong(o[h][0]), b1v = __double_as_longlong(o[h][1]);
                        ov[r][h] = u4_t{(unsigned)b0, (unsigned)(b0 >> 32), (unsigned)b1v, (unsigned)(b1v >> 32)};
                        chain = (int)b0;
                    }
                }
            };
            if constexpr (OUT == 1) {
                form(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
            } else {
                form(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
                __builtin_amdgcn_sched_barrier(0);
                form(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int si = NR * wave + r;
                const int64_t t = tile + (si >> 5);
                const int64_t row = t * B + (si & (B - 1));
                const bool ok = t < tile_end && row < n;  // (wave-uniform)
                const __amdgpu_buffer_rsrc_t orow = __builtin_amdgcn_make_buffer_rsrc(
                    p.recon + (ok ? row : 0) * (int64_t)d, 0, ok ? rowbytes : 0, 0x00020000);  // (lanes past d, rows past n: dropped)
                __builtin_amdgcn_raw_buffer_store_b128(ov[r][0], orow, lo * 16, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(ov[r][1], orow, lo * 16, 1024, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (is_gram) load_table();
            else kill_table();
        }
        __syncthreads();
        L8_STAMP(3)
        par ^= 1;
    }
#ifdef LLK8_TIMING
    if (wave == 1 && lane_entry == 0) sm[cfg::OFF_FLAG + 1] = (double)tq[2];
    __syncthreads();
#endif
    if (pf_acc == 0x7FF12345 && p.llks) p.llks[0] = 0.0;  // (never: the upper dword pattern of no finite double this code produces; keeps the touches alive)
    if (wave == SOLVER && scal) {
        const double v2 = wave_sum(run_llk), v3 = wave_sum(run_w);
        if (lane_entry == 0) {
            double *sc = scal;
            sc[SC_SQERR] = 0.0;
            sc[SC_DEVSQ] = 0.0;
            sc[SC_LLK] = v2;
            sc[SC_SUMW] = v3;
            sc[SC_NONEMPTY] = 0.0;
            sc[5] = 0.0;
            sc[6] = 0.0;
            sc[7] = 0.0;
#ifdef LLK8_TIMING
            sc[SC_SQERR] = (double)tq[0];      // contractions (wave 0)
            sc[SC_DEVSQ] = (double)tq[1];      // second staging (wave 0)
            sc[SC_NONEMPTY] = (double)tq[2];   // hand-off + solver + table request (wave 0)
            sc[5] = (double)tq[3];             // wait at the round's last barrier (wave 0)
            sc[6] = (double)tq[4];             // waits at the other barriers (wave 0)
            sc[7] = sm[cfg::OFF_FLAG + 1];     // overlapped staging of the next first tile (wave 1)
#endif
        }
    }
}

template <int K>
__global__ __launch_bounds__(512) void llk8_kernel(PassArgs p) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    if (p.qflag) {  // qprep's dynamic-range guard: pass_kernel<K, false, 4, false> runs instead
        int unsafe = 0;
#pragma unroll
        for (int t = 0; t < CfgL8<K>::NTP; ++t) unsafe |= p.qflag[t];
        if (unsafe) return;
    }
    constexpr int B = CfgL8<K>::B;
    const int64_t ntiles = (p.n + B - 1) / B;
    const int64_t tiles_per_wg = (ntiles + gridDim.x - 1) / gridDim.x;
    const int64_t tile_begin = (int64_t)blockIdx.x * tiles_per_wg;
    const int64_t tile_end = tile_begin + tiles_per_wg < ntiles ? tile_begin + tiles_per_wg : ntiles;
    llk8_run<K>(p, sm, tile_begin, tile_end, p.scal_part + (int64_t)blockIdx.x * 8);
}

template <int K, int OUT>
__global__ __launch_bounds__(512) void recon8_kernel(PassArgs p) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    if (p.qflag) {  // qprep's dynamic-range guard: pass_kernel<K, false, 4, false> runs instead
        int unsafe = 0;
#pragma unroll
        for (int t = 0; t < CfgL8<K>::NTP; ++t) unsafe |= p.qflag[t];
        if (unsafe) return;
    }
    constexpr int B = CfgL8<K>::B;
    const int64_t ntiles = (p.n + B - 1) / B;
    const int64_t tiles_per_wg = (ntiles + gridDim.x - 1) / gridDim.x;
    const int64_t tile_begin = (int64_t)blockIdx.x * tiles_per_wg;
    const int64_t tile_end = tile_begin + tiles_per_wg < ntiles ? tile_begin + tiles_per_wg : ntiles;
    llk8_run<K, OUT>(p, sm, tile_begin, tile_end, p.scal_part + (int64_t)blockIdx.x * 8);
}

// ---------------------------------------------------------------------------------------------------------------------
// mix_llk8_kernel -- the log-likelihood sweeps of ALL components of a mixture in ONE launch (PPCAMix::llks / the responsibilities
// of PPCAMix::iterate_with_prior, ppca/src/mix.rs:137-149, :283-288: llks of every component over every sample).  Round 5 ran one
// llk8_kernel launch per component: X came from HBM once per component (16.8 KB per sample and iteration at K = 8 against the 2.1 KB
// of one read).  Here a unit of work is (component, run of tiles) and the units are dealt so that the workgroups of ONE XCD (the
// blocks b with equal b mod 8 share an XCD's L2: observed placement, used for speed only) walk the SAME runs for the different
// components at the same time: a tile comes from HBM once and from that XCD's L2 for the other components.  Per unit the kernel is
// llk8_kernel's body (its component's C tile into LDS, its wave's slice of that component's digit table into registers), so the llks
// are bit-identical to the per-component sweeps.  A component whose table tripped the dynamic-range guard is skipped here and served
// by the fp64 instantiation of pass_kernel behind the same flag (launched per component by the host: returns at once otherwise).
template <int K>
__global__ __launch_bounds__(512) void mix_llk8_kernel(MixLlkArgs m) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    constexpr int B = CfgL8<K>::B, NTP = CfgL8<K>::NTP;
    const int W = gridDim.x, X = (W % 8 == 0) ? 8 : 1, Wx = W / X;
    const int x = blockIdx.x % X, j = blockIdx.x / X;
    const int64_t ntiles = (m.n + B - 1) / B;
    const int R = X * m.runs_per_xcd;
    int64_t tiles_per_run = (ntiles + R - 1) / R;
    tiles_per_run += tiles_per_run & 1;  // (rounds are two tiles: an even run keeps every run's pairs those of its neighbours)
    const int units = m.nm * m.runs_per_xcd;
    for (int q = j; q < units; q += Wx) {
        const int c = q % m.nm, run = x + X * (q / m.nm);
        const int64_t tile_begin = (int64_t)run * tiles_per_run;
        const int64_t tile_end = tile_begin + tiles_per_run < ntiles ? tile_begin + tiles_per_run : ntiles;
        if (tile_begin >= tile_end) continue;
        PassArgs p{};
        p.X = m.X;
        p.ldx = m.ldx;
        p.n = m.n;
        p.d = m.d;
        p.model = m.model[c];
        fused_qtab_view(m.tab[c], p);
        p.llks = m.llks[c];
        int unsafe = 0;
#pragma unroll
        for (int t = 0; t < NTP; ++t) unsafe |= p.qflag[t];
        if (unsafe) continue;
        __syncthreads();  // (the previous unit's readers of the C tile / mean are done)
        llk8_run<K>(p, sm, tile_begin, tile_end, nullptr);
    }
}

static bool llk8_enabled() {  // PPCA_LLK8=0: the four-wave llk2_kernel (A/B runs)
    static const bool v = [] {
        const char *e = getenv("PPCA_LLK8");
        return !(e && atoi(e) == 0);
    }();
    return v;
}

template <int K>
static hipError_t launch_llk8_t(int grid, const PassArgs &a, hipStream_t s) {
    const size_t lds = sizeof(double) * CfgL8<K>::LDS_DOUBLES;
    static std::atomic<unsigned long long> done{0ull};
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&llk8_kernel<K>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        done.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL((llk8_kernel<K>), dim3(grid), dim3(512), lds, s, a);
    return hipGetLastError();
}

template <int K>
static hipError_t launch_llk2_t(int grid, const PassArgs &a, hipStream_t s) {
    if (llk8_enabled()) return launch_llk8_t<K>(grid, a, s);
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

template <int K>
static hipError_t launch_mix_llk8_t(int grid, const MixLlkArgs &a, hipStream_t s) {
    const size_t lds = sizeof(double) * CfgL8<K>::LDS_DOUBLES;
    static std::atomic<unsigned long long> done{0ull};
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&mix_llk8_kernel<K>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        done.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL((mix_llk8_kernel<K>), dim3(grid), dim3(512), lds, s, a);
    return hipGetLastError();
}

template <int K, int OUT>
static hipError_t launch_recon8_t(int grid, const PassArgs &a, hipStream_t s) {
    const size_t lds = sizeof(double) * CfgL8<K>::LDS_DOUBLES;
    static std::atomic<unsigned long long> done{0ull};
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&recon8_kernel<K, OUT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        done.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL((recon8_kernel<K, OUT>), dim3(grid), dim3(512), lds, s, a);
    return hipGetLastError();
}
// smooth / extrapolate on the eight-wave sweep: recon modes 0 / 1 with an even d and nothing else asked of the pass (PPCA_RECON8=0:
// the four-wave pass_kernel, A/B runs)
bool recon8_covers(const PassArgs &a) {
    static const bool on = [] {
        const char *e = getenv("PPCA_RECON8");
        return !(e && atoi(e) == 0);
    }();
    return on && llk8_enabled() && a.recon && !a.states && !a.covs && !a.llks && (a.recon_mode == 0 || a.recon_mode == 1) && (a.d & 1) == 0 &&
           a.ldx == a.d;
}
hipError_t launch_recon8(int k, int grid, const PassArgs &a, hipStream_t s) {
#define PPCA_R8(KK) case KK: return a.recon_mode == 0 ? launch_recon8_t<KK, 1>(grid, a, s) : launch_recon8_t<KK, 2>(grid, a, s);
    switch (k) {
#ifdef PPCA_DEV_K10
        PPCA_R8(10)
#else
        PPCA_R8(1) PPCA_R8(2) PPCA_R8(3) PPCA_R8(4) PPCA_R8(5) PPCA_R8(6) PPCA_R8(7) PPCA_R8(8) PPCA_R8(9) PPCA_R8(10)
#endif
        default: return hipErrorInvalidValue;
    }
#undef PPCA_R8
}

// Runs of tiles per XCD group of workgroups for a launch of `grid` workgroups over nm components: the smallest count that deals
// every workgroup the same number of (component, run) units.
int mix_llk_runs_per_xcd(int grid, int nm) {
    const int X = (grid % 8 == 0) ? 8 : 1, Wx = grid / X;
    int a = nm, b = Wx;
    while (b) {
        const int t = a % b;
        a = b;
        b = t;
    }
    return Wx / a;  // Wx / gcd(nm, Wx)
}

bool mix_llk8_available() { return llk8_enabled(); }

hipError_t launch_mix_llk8(int k, int grid, const MixLlkArgs &a, hipStream_t s) {
    switch (k) {
#ifdef PPCA_DEV_K10
        case 10: return launch_mix_llk8_t<10>(grid, a, s);
#else
        case 1: return launch_mix_llk8_t<1>(grid, a, s);
        case 2: return launch_mix_llk8_t<2>(grid, a, s);
        case 3: return launch_mix_llk8_t<3>(grid, a, s);
        case 4: return launch_mix_llk8_t<4>(grid, a, s);
        case 5: return launch_mix_llk8_t<5>(grid, a, s);
        case 6: return launch_mix_llk8_t<6>(grid, a, s);
        case 7: return launch_mix_llk8_t<7>(grid, a, s);
        case 8: return launch_mix_llk8_t<8>(grid, a, s);
        case 9: return launch_mix_llk8_t<9>(grid, a, s);
        case 10: return launch_mix_llk8_t<10>(grid, a, s);
#endif
        default: return hipErrorInvalidValue;
    }
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
