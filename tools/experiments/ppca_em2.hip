// ppca_em2.hip -- the EM pass of the fused shapes (d <= 256, k <= 10) with TWO 32-sample tiles per round, so that the
// per-sample solver (P3) runs on all 64 lanes of a wave instead of 32 (pass_kernel's waves solve a 32-sample tile with
// half their lanes and repeat the factorisation of every sample in all four waves: 5.2 k of its 27.8 k cycles per tile).
//
// What stood in the way: cross = X~^T [wz | w] (P4a) needs the x~ tile AFTER the solver, the [G | b] contraction (P2)
// needs it BEFORE, and LDS holds one 66 KB x~ tile, not two.  Here the older tile's P4a operands -- 32 doubles per
// lane, the wave's 64 dims x 32 samples in MFMA A-operand layout -- are read into registers right after its P2
// (while x~ is still there), the second tile is staged into the same LDS tile, and after the joint solver step P4a of
// tile A runs from registers, P4a of tile B from LDS.  [G | b] of the 64 samples shrinks to one buffer (the int8 Gram
// has no K-split partial) and the solver writes [wP | wz | w] over it IN PLACE (one extra barrier between its loads
// and its stores).  Per-sample arithmetic is pass_kernel's; the accumulation order over samples is unchanged.
//
// MEASURED (round 2, N = 10 M, d = 256, k = 10): 56.7 EM it/s against 72 for pass_kernel, statistics equal to it (llk
// to 13 digits) -- OPT-IN (PPCA_EM2=1), kept as the record of the experiment.  The solver step does halve, but the
// register file, not LDS, is what the two-tile round runs out of: 200 accumulator registers sit in the AGPR half, and
// the arch half (256) has to hold the 64 stashed operand registers next to the solver's ~150 (factor 110, z 20,
// column 20) plus addressing; hipcc spills 71 dwords to scratch and moves 660 values per round through the 56 spare
// AGPRs.  The stash has no other home: 32 doubles per lane are 64 KB per workgroup, LDS has 15 KB left.
//
// Replaces, like pass_kernel: infer (ppca/src/ppca_model.rs:221-227), cross moment (:281-293), second moments
// (:294-306), noise 4-tuple (:328-358), llk (:142-149).  Int8-sliced Gram only; the guard's fallback is
// pass_kernel<K, true, 4, false>.
#include <atomic>
#include <cstdlib>

#include "ppca_device.hpp"

namespace ppca {

template <int K>
struct CfgE2 {
    using c = Cfg<K>;
    static constexpr int KP = c::KP, NTP = c::NTP, NTM = c::NTM, B = c::B, DP = c::DP, XS = c::XS, CS = c::CS;
    static constexpr int RS = 16 * NTM + 1;  // [G | b0] -> [wP | wz | w] row stride, 2 B rows
    static constexpr int BS = 17;            // b partial of dims 128-255, 2 B rows
    static constexpr int OFF_X = 0;
    static constexpr int OFF_C = OFF_X + B * XS;
    static constexpr int OFF_GW = OFF_C + DP * CS;
    static constexpr int OFF_B1 = OFF_GW + 2 * B * RS;
    static constexpr int OFF_M = OFF_B1 + 2 * B * BS;   // mask words: 4 tile slots x B x 4 u64
    static constexpr int OFF_MC = OFF_M + 4 * B * 4;    // observed counts: 4 slots x B ints
    static constexpr int OFF_R = OFF_MC + 2 * B;        // cross-wave scratch (16)
    static constexpr int OFF_L = OFF_R + 16;            // running scalars: sq[4 waves][2 B] | dev | llk | w | ne | pm | px (2 B each)
    static constexpr int LDS_DOUBLES = OFF_L + 4 * 2 * B + 6 * 2 * B;
};

template <int K>
__global__ __launch_bounds__(256) void em2_kernel(PassArgs p) {
    using cfg = CfgE2<K>;
    constexpr int KP = cfg::KP, NTP = cfg::NTP, NTM = cfg::NTM, B = cfg::B, XS = cfg::XS, CS = cfg::CS, RS = cfg::RS, BS = cfg::BS;
    constexpr int NW = 4, RPW = B / NW, DPS = cfg::DP / 2, STEPS = DPS / 4, RT = 16 / NW, DW = cfg::DP / NW;
    constexpr int PADS = 16 * NTP - KP;
    constexpr int SMALL_COLS = K + 1 - PADS;
    constexpr bool SPLIT = SMALL_COLS > 0 && SMALL_COLS <= 4 && PADS > 0;  // see pass_kernel
    constexpr int NTMB = SPLIT ? NTP : NTM;
    constexpr int STAGE_PIECES = 7;
    static_assert(NTP <= NW && QS == 8, "int8 Gram: one wave per packed-column tile, 8 digit slices");
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *Xs = sm + cfg::OFF_X;
    double *Cs = sm + cfg::OFF_C;
    double *GW = sm + cfg::OFF_GW;
    double *B1 = sm + cfg::OFF_B1;
    unsigned long long *Ms = reinterpret_cast<unsigned long long *>(sm + cfg::OFF_M);
    int *mcnt = reinterpret_cast<int *>(sm + cfg::OFF_MC);
    double *red = sm + cfg::OFF_R;
    double *scl = sm + cfg::OFF_L;

    if (p.qflag) {  // qprep's dynamic-range guard: pass_kernel<K, true, 4, false> runs instead
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
    constexpr int L_DEV = NW * 2 * B, L_LLK = L_DEV + 2 * B, L_W = L_DEV + 4 * B, L_NE = L_DEV + 6 * B, L_PM = L_DEV + 8 * B,
                  L_PX = L_DEV + 10 * B;
    for (int idx = tid; idx < L_DEV + 12 * B; idx += 256) scl[idx] = (idx >= L_PM && idx < L_PX) ? 1.0 : 0.0;
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
    d4_t accM[RT][NTM];
    double accS[RT];
    d4_t accX[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
#pragma unroll
        for (int t = 0; t < NTM; ++t) accM[r][t] = d4_t{0, 0, 0, 0};
        accS[r] = 0.0;
        accX[r] = d4_t{0, 0, 0, 0};
    }
    const double inv_s2 = 1.0 / s2;

    const int64_t ntiles = (n + B - 1) / B;
    const int64_t tiles_per_wg = (ntiles + gridDim.x - 1) / gridDim.x;
    const int64_t tile_begin = (int64_t)blockIdx.x * tiles_per_wg;
    const int64_t tile_end = tile_begin + tiles_per_wg < ntiles ? tile_begin + tiles_per_wg : ntiles;
    const int64_t nleft = n - tile_begin * B;
    const int nrel = (int)(nleft < (1 << 30) ? nleft : (1 << 30));
    const double *Xwg = p.X + tile_begin * B * p.ldx;
    double xr[RPW][4];
    auto load_row = [&](int64_t tile, int r) {  // unconditional (rows clamped to real ones; validity applied when staged)
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

    // ---- P1 in pieces (so that it can sit between the MFMAs of a P4b): per row, per half h: two "classify + centre"
    // pieces and one "file" piece; piece 6 = count + weighted |x~|^2
    int st_wlo = 0, st_whi = 0, st_m = 0;
    double xx_run = 0.0;
    double pc_xt0 = 0.0, pc_xt1 = 0.0, pc_xx = 0.0;
    unsigned long long pc_b0 = 0ull, pc_b1 = 0ull;
    int pc_m = 0;
    auto stage_begin = [&]() { st_wlo = st_whi = st_m = 0; };
    auto stage_piece = [&](int64_t t, int lane, auto r_tag, auto p_tag) {
        constexpr int r = decltype(r_tag)::value, P = decltype(p_tag)::value;
        const int ri = wave * RPW + r;
        if constexpr (P == 0 || P == 1 || P == 3 || P == 4) {
            constexpr int q = (P < 2) ? P : P - 1, e = q & 1;
            if constexpr (q == 0) {
                pc_xx = 0.0;
                pc_m = 0;
            }
            const bool row_ok = t < tile_end && (int)(t - tile_begin) * B + ri < nrel;  // wave-uniform
            const double v = xr[r][q];
            const unsigned long long bal = __builtin_amdgcn_fcmp(__builtin_fabs(v), __builtin_inf(), 4) & (row_ok ? dimmask[q] : 0ull);
            const double xt = keep_if(v - mu[q], bal);  // select, never multiply (utils.rs:118-127)
            if constexpr (e == 0) {
                pc_b0 = bal;
                pc_xt0 = xt;
            } else {
                pc_b1 = bal;
                pc_xt1 = xt;
            }
        } else if constexpr (P == 2 || P == 5) {
            constexpr int h = (P == 2) ? 0 : 1;
            auto weave = [&](unsigned ev, unsigned od) {
                unsigned long long re, ro;
                asm("s_bitreplicate_b64_b32 %0, %1" : "=s"(re) : "s"(ev));
                asm("s_bitreplicate_b64_b32 %0, %1" : "=s"(ro) : "s"(od));
                return (re & 0x5555555555555555ull) | (ro & 0xAAAAAAAAAAAAAAAAull);
            };
            const unsigned long long w0 = weave((unsigned)pc_b0, (unsigned)pc_b1);
            const unsigned long long w1 = weave((unsigned)(pc_b0 >> 32), (unsigned)(pc_b1 >> 32));
            st_wlo = writelane_s<4 * r + 2 * h>(st_wlo, (int)(unsigned)w0);
            st_whi = writelane_s<4 * r + 2 * h>(st_whi, (int)(unsigned)(w0 >> 32));
            st_wlo = writelane_s<4 * r + 2 * h + 1>(st_wlo, (int)(unsigned)w1);
            st_whi = writelane_s<4 * r + 2 * h + 1>(st_whi, (int)(unsigned)(w1 >> 32));
            typedef double d2_t __attribute__((ext_vector_type(2)));
            *reinterpret_cast<d2_t *>(Xs + ri * XS + 128 * h + 2 * lane) = d2_t{pc_xt0, pc_xt1};
            pc_xx += pc_xt0 * pc_xt0;
            pc_xx += pc_xt1 * pc_xt1;
            pc_m += __popcll(pc_b0) + __popcll(pc_b1);
        } else if constexpr (P == 6) {
            st_m = writelane<r>(st_m, pc_m);
            const int64_t row = t * B + ri;
            const double wr = p.w ? p.w[row < n ? row : n - 1] : 1.0;  // wave-uniform (scalar load)
            xx_run += wr * pc_xx;
        }
    };
    auto stage_end = [&](int64_t t, int lane) {
        const int slot4 = (int)((t - tile_begin) & 3);
        const unsigned long long myw = ((unsigned long long)(unsigned)st_whi << 32) | (unsigned)st_wlo;
        if (lane < 4 * RPW) Ms[slot4 * 4 * B + wave * 4 * RPW + lane] = myw;
        if (lane < RPW) mcnt[slot4 * B + wave * RPW + lane] = st_m;
    };
    auto stage_plain = [&](int64_t t, int lane) {
        stage_begin();
        static_for<RPW>([&](auto r_tag) {
            static_for<STAGE_PIECES>([&](auto p_tag) { stage_piece(t, lane, r_tag, p_tag); });
        });
        stage_end(t, lane);
    };

    // ---- P2: [G | b] of the tile in the x~ buffer -> rows slot * B .. of the exchange buffer
    auto contract_tile = [&](int64_t t, int lane, int slot) {
        const int l15 = lane & 15, l4 = lane >> 4;
        const int colb = (l15 < K) ? l15 : K;
        const int rt = wave & 1, kq = wave >> 1;
        const int si = 16 * rt + l15;
        const unsigned long long *Msc = Ms + (int)((t - tile_begin) & 3) * 4 * B;
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
            for (int kc = 0; kc < 4; ++kc) mwd[rt2][kc] = Msc[(16 * rt2 + l15) * 4 + kc];
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
        group(qbA, true);   // digits {7,6}
        load_pair(qbA, 2);
        group(qbB, false);  // digits {5,4}
        load_pair(qbB, 0);
        {
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
        group(qbA, false);  // digits {3,2}
        group(qbB, false);  // digits {1,0}
        if (gram_wave) {
#pragma unroll
            for (int rt2 = 0; rt2 < 2; ++rt2)
#pragma unroll
                for (int r = 0; r < 4; ++r)  // C/D map of the 16x16 integer MFMA: row = 4 (lane >> 4) + reg
                    GW[(slot * B + 16 * rt2 + 4 * l4 + r) * RS + 16 * wave + l15] = v[rt2][r] * qs;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // the K-split partials of b, summed by the solver in a fixed order (p0 + p1)
            if (kq == 0) GW[(slot * B + 16 * rt + l4 + 4 * r) * RS + 16 * NTP + l15] = accb[r];
            else B1[(slot * B + 16 * rt + l4 + 4 * r) * BS + l15] = accb[r];
        }
    };

    // ---- P4b of one tile (rows slot * B .. of the W buffer, mask words of the tile), optionally with the staging of
    // tile `tnext` between the MFMAs
    auto mask_side = [&](int64_t t, int lane, int slot, auto stage_tag, int64_t tnext) {
        constexpr bool STAGE = decltype(stage_tag)::value;
        const int l15 = lane & 15, l4 = lane >> 4;
        const unsigned long long *Msc = Ms + (int)((t - tile_begin) & 3) * 4 * B;
        const double *Wsc = GW + slot * B * RS;
        if constexpr (STAGE) stage_begin();
        unsigned long long mwc = Msc[l4 * 4 + (DW * wave) / 64];
        double bwc[NTMB], bsc = 0.0;
#pragma unroll
        for (int tt = 0; tt < NTMB; ++tt) bwc[tt] = Wsc[l4 * RS + 16 * tt + l15];
        if constexpr (SPLIT) bsc = Wsc[l4 * RS + 16 * NTP + PADS + (lane & 3)];
        static_for<8>([&](auto s_tag) {
            constexpr int s = decltype(s_tag)::value;
            unsigned long long mwn = 0ull;
            double bwn[NTMB], bsn = 0.0;
            constexpr int PER_R = NTMB + (SPLIT ? 1 : 0);
            constexpr int SLOTS = RT * PER_R;
            static_for<SLOTS>([&](auto i_tag) {
                constexpr int i = decltype(i_tag)::value, r = i / PER_R, tt = i % PER_R;
                const int sh = ((DW * wave) & 63) + 16 * r;
                const int am_hi = __builtin_amdgcn_sbfe((int)(unsigned)(mwc >> (sh & 32)), (sh & 31) + l15, 1) & 0x3FF00000;
                const double am = __hiloint2double(am_hi, 0);
                if constexpr (tt < NTMB) {
                    accM[r][tt] = mfma(am, bwc[tt], accM[r][tt]);
                } else {
                    accS[r] = __builtin_amdgcn_mfma_f64_4x4x4f64(am, bsc, accS[r], 0, 0, 0);
                }
                if constexpr (STAGE && s < RPW) {
                    constexpr int P0 = i * STAGE_PIECES / SLOTS, P1 = (i + 1) * STAGE_PIECES / SLOTS;
                    static_for<P1 - P0>([&](auto o_tag) {
                        stage_piece(tnext, lane, s_tag, std::integral_constant<int, P0 + decltype(o_tag)::value>{});
                    });
                }
                if constexpr (i == SLOTS * 3 / 4 && s + 1 < 8) {
                    const int smp = 4 * (s + 1) + l4;
                    mwn = Msc[smp * 4 + (DW * wave) / 64];
#pragma unroll
                    for (int u = 0; u < NTMB; ++u) bwn[u] = Wsc[smp * RS + 16 * u + l15];
                    if constexpr (SPLIT) bsn = Wsc[smp * RS + 16 * NTP + PADS + (lane & 3)];
                }
            });
            if constexpr (s + 1 < 8) {
                mwc = mwn;
#pragma unroll
                for (int u = 0; u < NTMB; ++u) bwc[u] = bwn[u];
                bsc = bsn;
            }
        });
        if constexpr (STAGE) stage_end(tnext, lane);
    };

    if (tile_begin < tile_end) {
#pragma unroll
        for (int r = 0; r < RPW; ++r) load_row(tile_begin, r);
        stage_plain(tile_begin, lane_entry);
    }
    __syncthreads();
    for (int64_t tile = tile_begin; tile < tile_end; tile += 2) {
        int lane = lane_entry;
        asm volatile("" : "+v"(lane));
        const int l15 = lane & 15, l4 = lane >> 4;
        // ---- tile A: [G | b]; its rows of tile B are requested first and travel under the contraction
#pragma unroll
        for (int r = 0; r < RPW; ++r) load_row(tile + 1, r);
        contract_tile(tile, lane, 0);
        // the x~-side operands of tile A's P4a, while the x~ tile still holds A: dims DW wave + 16 r + l15, sample 4 s + l4
        double stash[8][RT];
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int r = 0; r < RT; ++r) stash[s][r] = Xs[(4 * s + l4) * XS + DW * wave + 16 * r + l15];
        __syncthreads();
        // ---- tile B: stage, [G | b]
        stage_plain(tile + 1, lane);
        __syncthreads();
        contract_tile(tile + 1, lane, 1);
        __syncthreads();
        // ---- P3: lane i < 32 -> sample i of tile A, lane i >= 32 -> sample i - 32 of tile B; every wave factors every
        // sample, the waves share the independent columns of M^-1, wave 0 also owns z, llk and the scalars
        {
            const int slot = lane >> 5, i = lane & (B - 1);
            const int64_t t = tile + slot;
            const int64_t row = t * B + i;
            const bool mine = t < tile_end && row < n;
            double *wrow = GW + lane * RS;
            const double wgt = mine ? (p.w ? p.w[row] : 1.0) : 0.0;
            const int m = mcnt[(int)((t - tile_begin) & 3) * B + i];
            Posterior<K> post;
            double z[K], quad, zz;
            post.load([&](int e) { return wrow[e]; }, s2);
#pragma unroll
            for (int a = 0; a < K; ++a) z[a] = wrow[16 * NTP + a] + B1[lane * BS + a];
            const double sq_run = scl[wave * 2 * B + lane];
            __syncthreads();  // every wave holds [G | b] of its lane's sample: the rows may be overwritten
            double sc_sq = 0.0, sc_dev = 0.0, sc_llk = 0.0, sc_w = 0.0, sc_ne = 0.0;
            double pm;
            int pe;
            post.factor_loaded(pm, pe);
            post.solve_loaded(z, quad, zz);
            double trpart = 0.0;
#pragma unroll
            for (int c = 0; c < K; ++c) {
                if (column_owner(K, c, NW) != wave) continue;
                // P = z z^T + Sigma, Sigma = sigma^2 M^-1 (ppca_model.rs:437-439), weighted
                trpart += post.minv_column(c, [&](int a, int cc, double v) { wrow[tri(a, cc)] = wgt * (z[a] * z[cc] + s2 * v); });
            }
            // tr(C_o Sigma C_o^T) = <Sigma, G> = s2 (K - s2 tr M^-1)  (:345); all-masked samples are filtered out (:333)
            if (m > 0) sc_sq -= wgt * s2 * s2 * trpart;
            if (wave == 0) {
                const double run_dev = scl[L_DEV + lane], run_llk = scl[L_LLK + lane], run_w = scl[L_W + lane], run_ne = scl[L_NE + lane];
                const double run_pm = scl[L_PM + lane], run_px = scl[L_PX + lane];
                double *zrow = wrow + 16 * NTP;  // W row = [w P (K') | pads | w z (K) | w | 0..]
#pragma unroll
                for (int a = 0; a < 16; ++a) zrow[a] = (a < K) ? wgt * z[a] : (a == K ? wgt : 0.0);
                if constexpr (SPLIT) {
#pragma unroll
                    for (int a = 0; a < PADS; ++a) wrow[KP + a] = (a < K) ? wgt * z[a] : wgt;
                }
                if (m > 0) {
                    sc_sq += wgt * s2 * (double)K;
                    sc_dev += wgt * (0.0 - quad - s2 * zz);  // |x~ - C_o z|^2 minus |x~|^2, added in the epilogue (:346)
                    sc_ne += mine ? 1.0 : 0.0;
                }
                const double lk0 = sample_llk_nolog(0.0, quad, inv_s2, lnsig, m, K);
                if (p.w) {
                    if (!p.no_llk) sc_llk += wgt * (m > 0 ? lk0 - 0.5 * Posterior<K>::logdet(pm, pe) : 0.0);
                } else {
                    const bool use = m > 0 && mine;  // wgt is 1 for real rows
                    sc_llk += use ? lk0 : 0.0;
                    int e;
                    scl[L_PM + lane] = frexp(run_pm * (use ? pm : 1.0), &e);
                    scl[L_PX + lane] = run_px + (double)(e + (use ? pe : 0));
                }
                sc_w += wgt;
                scl[L_DEV + lane] = run_dev + sc_dev;
                scl[L_LLK + lane] = run_llk + sc_llk;
                scl[L_W + lane] = run_w + sc_w;
                scl[L_NE + lane] = run_ne + sc_ne;
            }
            scl[wave * 2 * B + lane] = sq_run + sc_sq;
        }
        __syncthreads();
        // ---- P4a: cross / sumx += X~^T [wz | w]; tile A from the stashed operands, tile B from the x~ tile; the next
        // round's first tile is requested one row per k-step behind the MFMAs of tile B
        {
            const double *WA = GW, *WB = GW + B * RS;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const double bz = WA[(4 * s + l4) * RS + 16 * NTP + l15];
#pragma unroll
                for (int r = 0; r < RT; ++r) accX[r] = mfma(stash[s][r], bz, accX[r]);
            }
            double bzb[2], axb[2][RT];
            bzb[0] = WB[l4 * RS + 16 * NTP + l15];
#pragma unroll
            for (int r = 0; r < RT; ++r) axb[0][r] = Xs[l4 * XS + DW * wave + 16 * r + l15];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (s < RPW) load_row(tile + 2, s);
                if (s + 1 < 8) {
                    const int smp = 4 * (s + 1) + l4;
                    bzb[(s + 1) & 1] = WB[smp * RS + 16 * NTP + l15];
#pragma unroll
                    for (int r = 0; r < RT; ++r) axb[(s + 1) & 1][r] = Xs[smp * XS + DW * wave + 16 * r + l15];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < RT; ++r) accX[r] = mfma(axb[s & 1][r], bzb[s & 1], accX[r]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();  // the x~ tile is free
        // ---- P4b of tile A with the staging of the next round's first tile between its MFMAs, then P4b of tile B
        mask_side(tile, lane, 0, std::true_type{}, tile + 2);
        mask_side(tile + 1, lane, 1, std::false_type{}, tile + 2);
        __syncthreads();
    }

    // ---------------------------------------------------------------- epilogue
    {
        const int lane = lane_entry, l15 = lane & 15, l4 = lane >> 4;
        const double sq_w = wave_sum(scl[wave * 2 * B + lane]);
        const double xx_w = wave_sum(xx_run);
        if (lane == 0) {
            red[wave] = sq_w;
            red[NW + wave] = xx_w;
        }
        __syncthreads();
        StatsLayout L(d, K);
        double *out = p.part + (int64_t)blockIdx.x * L.len;
        if (wave == 0) {
            double v0 = 0.0, xx_tot = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) v0 += red[w];
#pragma unroll
            for (int w = 0; w < NW; ++w) xx_tot += red[NW + w];
            double sc_llk = scl[L_LLK + lane];
            sc_llk -= 0.5 * (log(scl[L_PM + lane]) + scl[L_PX + lane] * LN_2);
            const double v1 = wave_sum(scl[L_DEV + lane]), v2 = wave_sum(sc_llk), v3 = wave_sum(scl[L_W + lane]),
                         v4 = wave_sum(scl[L_NE + lane]);
            if (lane == 0) {
                double *sc = out + L.scalars;
                sc[SC_SQERR] = v0;
                sc[SC_DEVSQ] = v1 + xx_tot;
                sc[SC_LLK] = v2 - 0.5 * inv_s2 * xx_tot;
                sc[SC_SUMW] = v3;
                sc[SC_NONEMPTY] = v4;
                sc[5] = 0.0;
                sc[6] = 0.0;
                sc[7] = 0.0;
            }
        }
#pragma unroll
        for (int r = 0; r < RT; ++r) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int dim = DW * wave + 16 * r + l4 + 4 * q;  // C/D row of v_mfma_f64_16x16x4
                if (dim >= d) continue;
#pragma unroll
                for (int t = 0; t < NTP; ++t) {
                    const int c = 16 * t + l15;
                    if (c < KP) out[L.S + (int64_t)dim * KP + c] = accM[r][t][q];
                    if constexpr (SPLIT) {
                        if (t == NTP - 1 && c >= KP) {
                            const int a = c - KP;
                            if (a < K) out[L.U + (int64_t)dim * K + a] = accM[r][t][q];
                            else if (a == K) out[L.totals + dim] = accM[r][t][q];
                        }
                    }
                }
                if (l15 < K) {
                    if constexpr (!SPLIT) out[L.U + (int64_t)dim * K + l15] = accM[r][NTP][q];
                    out[L.cross + (int64_t)dim * K + l15] = accX[r][q];
                } else if (l15 == K) {
                    if constexpr (!SPLIT) out[L.totals + dim] = accM[r][NTP][q];
                    out[L.sumx + dim] = accX[r][q];
                }
            }
            if constexpr (SPLIT) {  // 4x4x4 group: D lane = 16 i + 4 block + j
                const int dim = DW * wave + 16 * r + 4 * ((lane >> 2) & 3) + (lane >> 4);
                const int a = PADS + (lane & 3);
                if (dim < d) {
                    if (a < K) out[L.U + (int64_t)dim * K + a] = accS[r];
                    else if (a == K) out[L.totals + dim] = accS[r];
                }
            }
        }
    }
}

template <int K>
static hipError_t launch_em2_t(int grid, const PassArgs &a, hipStream_t s) {
    const size_t lds = sizeof(double) * CfgE2<K>::LDS_DOUBLES;
    static std::atomic<unsigned long long> done{0ull};
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&em2_kernel<K>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        done.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL((em2_kernel<K>), dim3(grid), dim3(256), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_em2(int k, int grid, const PassArgs &a, hipStream_t s) {
    switch (k) {
#ifdef PPCA_DEV_K10
        case 10: return launch_em2_t<10>(grid, a, s);
#else
        case 2: return launch_em2_t<2>(grid, a, s);
        case 3: return launch_em2_t<3>(grid, a, s);
        case 4: return launch_em2_t<4>(grid, a, s);
        case 5: return launch_em2_t<5>(grid, a, s);
        case 6: return launch_em2_t<6>(grid, a, s);
        case 7: return launch_em2_t<7>(grid, a, s);
        case 8: return launch_em2_t<8>(grid, a, s);
        case 9: return launch_em2_t<9>(grid, a, s);
        case 10: return launch_em2_t<10>(grid, a, s);
#endif
        default: return hipErrorInvalidValue;
    }
}

}  // namespace ppca
