// ppca_kernels.hip -- gfx950 (CDNA4) kernels of the PPCA EM hot path.
//
// pass_kernel<K, EM>: ONE streaming pass over the sample matrix X that fuses what
// the reference does in five sweeps (reference = viodotcom/ppca_rs):
//   infer            ppca/src/ppca_model.rs:221-227 (infer_one :195-208)
//   cross moment     :281-293
//   second moments   :294-306   (the reference's d sequential scans over N)
//   noise 4-tuple    :328-358
//   llk              :142-149
// Persistent workgroups (one per CU) walk tiles of 32 samples:
//   P1  coalesced row loads (prefetched one tile ahead in registers) ->
//       x~ = observed ? x - mean : 0 into an LDS tile, mask words by wave ballot
//   P2  [G | b] = [Mask | X~] (32 x 256) . [vech(c c^T) | C] (256 x (k'+k))
//       on v_mfma_f64_16x16x4_f64 (true dense contractions over the 256 dims)
//   P3  per-sample k x k SPD solve, one lane per sample, registers only:
//       z, Sigma, ln det, quadratic form -> llk, noise terms, w P, w z
//   P4  S/U/totals (256 x (k'+k+1)) += Mask^T . [wP | wz | w],
//       cross/sumx (256 x (k+1))   += X~^T  . [wz | w]   on fp64 MFMA,
//       accumulators persistent in registers across all tiles of the workgroup
// and end with one deterministic per-workgroup partial that reduce_partials sums
// in fixed order.  No atomics: results are bit-reproducible for a given grid.
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "ppca_device.hpp"

namespace ppca {

// ------------------------------------------------------------------ int8-sliced Gram operand
// G_i = sum_j m_ij vech(c_j c_j^T) has one EXACT operand (the 0/1 mask), so the other one can be
// split into QS signed QB-bit digits (QB = 8 since round 3: balanced base 256; 7 in rounds 1-2) of a per-column
// fixed-point representation and contracted on v_mfma_i32_16x16x64_i8 with exact integer accumulation:
//   Q[j][c] = c_ja c_jb  ~  2^E_c 2^-(QB QS - 2) sum_s 256^s dig_s[j][c],   dig_s in [-128, 127],
// E_c = exponent of max_j |Q[j][c]| (|Q| < 2^E_c).  QS balanced digits span (-0.502, +0.498) 256^QS, so
// the integers are kept below 2^(QB QS - 2): with QS = 8 that is 62 bits under the column maximum -- every product
// whose magnitude is within 2^9 of its column's largest is carried with its full fp64 mantissa.
// qtab: [NTP][QS][4 k-chunks][64 lanes][16 bytes]; lane = 16 (dim/16 % 4) + (col % 16), byte = dim % 16
// qscale: [64] dequantisation multipliers 2^(E_c - (QB QS - 2)).
// One workgroup per packed-column tile t (256 threads).  Step 1, thread = dim: maxima of the tile's 16 columns
// (wave-level maximum first -- non-negative doubles order like their bit patterns -- then one LDS atomic per wave)
// -> qscale.  Step 2, thread = (k-chunk, lane): 16 dims x one column, its 16 digits of every slice as one 16-byte
// store per slice.  (Two launches before: a single-workgroup maximum over all 55 columns took 21 us per iteration.)
//
// Dynamic-range guard.  One scale per column means that a sample which masks the rows carrying the column maximum
// gets a Gram summed from entries truncated at 2^(E_c - 63) each.  Per model (no knowledge of the masks) two
// rigorous bounds are available, with eps = max_c 2^(E_c - 63), m' = observed rows with a non-zero c_j:
//   forward:   |dG|_F <= K m' eps  and  lambda_min(M_i) >= sigma^2  =>  |dM^-1| / |M^-1| <= K d eps / sigma^2
//   backward:  |G_i|_2 >= tr(G_i) / K >= m' r_min / K  (r_min = smallest non-zero |c_j|^2)
//                                                      =>  |dG|_F / |G_i|_2 <= K^2 eps / r_min
// The int8 Gram is used when the forward bound is below 1e-8 OR the backward bound is below 2^-40 (fp64
// accumulation itself sits at ~2^-50 of |G_i|); otherwise -- or when a product is not finite / >= 1e300, which the
// fixed-point form cannot carry -- qflag[tile] is raised and the launcher's fp64-MFMA instantiation of the pass
// runs instead (both are enqueued, each returns at once unless the flag selects it: no host round trip).
// With unit-scale C (column maxima ~ 2^4) the forward test passes down to sigma ~ 7e-4 and the backward test down to
// row norms |c_j|^2 ~ 2e-4 of the scale (7-bit digits: 1e-2 and 5e-2): a trained model with a small sigma and a few
// weakly loaded dimensions stays on the int8 engine.
constexpr double QGUARD_FWD = 1.0e-8;
constexpr double QGUARD_BWD = 9.094947017729282e-13;  // 2^-40

// (the body: C(j, a) reads the transform -- from the model buffer, or from the LDS copy finalize_qprep_kernel has just computed)
template <int K, class CF>
__device__ __forceinline__ void qprep_body(CF C, double s2m, int d, double *qscale, signed char *qtab, int *qflag) {
    constexpr int KP = Cfg<K>::KP;
    __shared__ unsigned long long cmax[16];
    __shared__ double scale[16];
    __shared__ unsigned long long rmin_bits;
    __shared__ int bad;
    const int t = blockIdx.x, j = threadIdx.x;
    if (j < 16) cmax[j] = 0ull;
    if (j == 0) {
        rmin_bits = 0x7FF0000000000000ull;  // +inf: no non-zero row
        bad = 0;
    }
    __syncthreads();
    double cj[K];
#pragma unroll
    for (int a = 0; a < K; ++a) cj[a] = (j < d) ? C(j, a) : 0.0;
    {
        double rn = 0.0;
#pragma unroll
        for (int a = 0; a < K; ++a) rn += cj[a] * cj[a];
        // non-negative doubles order like their bit patterns; NaN / inf rows are caught by the product test below
        if (rn > 0.0 && rn < 1.0e300) atomicMin(&rmin_bits, (unsigned long long)__double_as_longlong(rn));
    }
    int notfin = 0;
    double q[16];
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) {
        const int c = 16 * t + cc;
        int a = 0;
        while ((a + 1) * (a + 2) / 2 <= c) ++a;
        const int b = c - a * (a + 1) / 2;
        double ca = 0.0, cb = 0.0;
#pragma unroll
        for (int u = 0; u < K; ++u) {  // (register arrays are indexed by compile-time constants only)
            ca = (u == a) ? cj[u] : ca;
            cb = (u == b) ? cj[u] : cb;
        }
        q[cc] = (c < KP) ? fabs(ca * cb) : 0.0;
        if (!(q[cc] < 1.0e300)) {
            q[cc] = 0.0;
            notfin = 1;
        }
    }
    if (notfin) atomicOr(&bad, 1);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)  // sixteen independent butterflies per level: the shuffle latency overlaps
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) q[cc] = fmax(q[cc], __shfl_xor(q[cc], o, 64));
    if ((j & 63) == 0) {
#pragma unroll
        for (int cc = 0; cc < 16; ++cc)
            if (q[cc] > 0.0) atomicMax(&cmax[cc], (unsigned long long)__double_as_longlong(q[cc]));
    }
    __syncthreads();
    if (j < 16) {
        int e = 0;
        const double mx = __longlong_as_double((long long)cmax[j]);
        if (mx > 0.0) (void)frexp(mx, &e);
        const double sc = ldexp(1.0, e - (QB * QS - 2));
        scale[j] = sc;
        qscale[16 * t + j] = sc;
        if (mx > 0.0) {
            const double eps = 0.5 * sc;  // rounding bound of one entry: 2^(E_c - 63)
            const double rmin = __longlong_as_double((long long)rmin_bits);
            const bool fwd = (double)K * (double)d * eps <= QGUARD_FWD * s2m;
            const bool bwd = (double)(K * K) * eps <= QGUARD_BWD * rmin;
            if (!(fwd || bwd)) atomicOr(&bad, 1);
        }
    }
    __syncthreads();
    if (j == 0) qflag[t] = bad;
    if constexpr (K <= FUSED_MAX_K) {
        // Slots 8.. belong to the fused pass's second stage (reduce_wguard_kernel) ONLY in the fused layout: at k = 16 this kernel
        // has nine blocks and block 8 files its own tile's verdict in qflag[8] -- round 4's unconditional reset of slots 8 / 9 by
        // block 0 raced with it and could clear the flag of the tile holding column pairs (15, 8..15) (advisor, round 4).
        // Tiles this state size does not have are cleared too: ppca_em_last_guard reads the first four flags whatever the k of
        // the last pass, and the buffer may have held a larger model's (or the two-kernel pass's) flags.
        static_assert(Cfg<K>::NTP <= 8, "slots 8 / 9 must not alias a tile flag");
        // (slots 8..15 -- QF_* in ppca_internal.hpp -- are reduce_wguard_kernel's: every one is written before it is read, except
        //  its ticket counter, which the host zeroes with the buffer and the kernel's last workgroup resets)
        if (j == 0 && t == 0)
            for (int u = (K * (K + 1) / 2 + 15) / 16; u < 8; ++u) qflag[u] = 0;
    }
    // zero-padded copy of C: ppca_em9.hip (and em8's -DE8_C_GLOBAL experiment) read the B operands of b = X~ C from it.  Only
    // in the layout of fused_qtab_layout (k <= FUSED_MAX_K: the copy sits behind the largest table); the callers with their
    // own, exactly sized tables (ppca_em16.hip's, k = 11..16) have no room behind them -- round 4 found that the hard way:
    // the copy went over the two-kernel pass's workspace.
    if constexpr (K <= FUSED_MAX_K) {  // (every block writes its share: block 0 alone was the kernel's tail)
        constexpr int NB = Cfg<K>::NTP;
        double *cp = reinterpret_cast<double *>(qtab + qtab_bytes<FUSED_MAX_K>());
        for (int idx = 256 * t + j; idx < FUSED_MAX_D * (K + 1); idx += 256 * NB) {
            const int jj = idx / (K + 1), a2 = idx - jj * (K + 1);
            cp[idx] = (jj < d && a2 < K) ? C(jj, a2) : 0.0;
        }
        // ... and C in the operand order of em9_kernel's b = X~ C on v_mfma_f64_4x4x4 (ppca_internal.hpp, CPB_DOUBLES): one
        // contiguous 256-byte block per (dimension half, step, column group); the kernel copies the first two groups into LDS
        double *cb = cp + FUSED_MAX_D * (FUSED_MAX_K + 1);
        constexpr int NCGB = (K + 3) / 4;
        for (int idx = 256 * t + j; idx < 2 * 16 * NCGB * 32; idx += 256 * NB) {
            const int e = idx & 31, blk = idx >> 5;
            const int c = blk % NCGB, q = (blk / NCGB) & 15, kq = blk / (NCGB * 16);
            const int i = e & 3, kb = (e >> 2) & 1, kk = e >> 3;
            const int dim = 128 * kq + 32 * (q >> 2) + 16 * kb + 4 * (q & 3) + kk, col = 4 * c + i;
            cb[idx] = (dim < d && col < K) ? C(dim, col) : 0.0;
        }
    }
    const int lane = j & 63, kc = j >> 6;
    const int c = 16 * t + (lane & 15);
    int a = 0;
    while ((a + 1) * (a + 2) / 2 <= c) ++a;
    const int b = c - a * (a + 1) / 2;
    const int j0 = 64 * kc + 16 * (lane >> 4);
    int ex = 0;
    (void)frexp(scale[lane & 15], &ex);  // scale = 2^(E - (QB QS - 2)) = 0.5 * 2^(ex)
    const int shift = -(ex - 1);         // multiply by 2^(QB QS - 2 - E)
    union { signed char b8[QS][16]; i4_t v[QS]; } dg;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        // k index j0 + jj of the contraction <-> bit (j0 + jj) % 64 of mask word (j0 + jj) / 64 of a sample; the
        // staging leaves the finite-test ballots as they come (lane l of ballot 2 h + e tested dim 128 h + 2 l + e),
        // so word q = 2 h + e, bit l is dim 128 h + 2 l + e: the order of a sum is free, the table follows the masks
        const int kk = j0 + jj;
        const int jd = 128 * (kk >> 7) + 2 * (kk & 63) + ((kk >> 6) & 1);
        double q = 0.0;
        if (c < KP && jd < d) q = C(jd, a) * C(jd, b);
        if (!(fabs(q) < 1.0e300)) q = 0.0;
        long long I = llrint(ldexp(q, shift));  // |I| <= 2^(QB QS - 2)
#pragma unroll
        for (int sl = 0; sl < QS; ++sl) {
            const int dig = (int)((I + QBASE / 2) & (QBASE - 1)) - QBASE / 2;
            I = (I - dig) >> QB;
            dg.b8[sl][jj] = (signed char)dig;
        }
    }
#pragma unroll
    for (int sl = 0; sl < QS; ++sl)
        reinterpret_cast<i4_t *>(qtab)[(((size_t)t * QS + sl) * 4 + kc) * 64 + lane] = dg.v[sl];
}
template <int K>
__global__ __launch_bounds__(256) void qprep_kernel(const double *model, int d, double *qscale, signed char *qtab, int *qflag) {
    qprep_body<K>([=](int j, int a) { return model[MODEL_HDR + (int64_t)j * K + a]; }, model[1], d, qscale, qtab, qflag);
}
// ... of several models in one launch (the components of a mixture): grid (tiles, models), each model's block of fused_qtab_bytes()
template <int K>
__global__ __launch_bounds__(256) void qprep_multi_kernel(MixTabArgs m) {
    const int c = blockIdx.y;
    const double *model = m.model[c];
    PassArgs t{};
    fused_qtab_view(m.tab[c], t);
    qprep_body<K>([=](int j, int a) { return model[MODEL_HDR + (int64_t)j * K + a]; }, model[1], m.d, t.qscale, t.qtab, t.qflag);
}

// NW = waves per workgroup: 4 (one wave per SIMD, 512 registers each) or 8 (two waves per SIMD,
// 256 registers each, every wave owning half as many accumulator tiles).
// GI8: Gram on the int8 MFMA (wave t <-> packed-column tile t) instead of fp64 MFMA.
// GATHER: the pass honours PassArgs::rows (sample i = physical row rows[i]); the fp64-Gram instantiations always do
// (they are the fallback of both forms), the tuned int8 instantiation only as its own variant, so that the
// un-gathered hot kernel carries no trace of it.
#ifndef PPCA_P4_INT8
#define PPCA_P4_INT8 0
#endif
template <int K, bool EM, int NW, bool GI8, bool GATHER = false>
__global__ __launch_bounds__(64 * NW) void pass_kernel(PassArgs p) {
    using cfg = Cfg<K>;
    constexpr int KP = cfg::KP, NTP = cfg::NTP, NTM = cfg::NTM, B = cfg::B, XS = cfg::XS, CS = cfg::CS,
                  GS = cfg::GS, WS = cfg::WS;
    constexpr int THREADS = 64 * NW;
    constexpr int RPW = B / NW;          // rows staged per wave in P1
    constexpr int KS = NW / 2;           // K-splits of the [G | b] contraction (x 2 row tiles = NW waves)
    constexpr int DPS = cfg::DP / KS;    // dims per split
    constexpr int STEPS = DPS / 4;       // MFMA k-steps per split
    constexpr int WPS = DPS / 64;        // mask words per split
    constexpr int RT = 16 / NW;          // accumulator row tiles (16 dims each) per wave in P4
    constexpr int DW = cfg::DP / NW;     // dims owned by a wave in P4
    // Mask-side columns of P4: k' (vech P) + k (w z) + 1 (w).  When they overflow the k' tiles by at most four
    // columns (k = 10: 66 = 4 x 16 + 2) the first PADS of [w z | w] ride in the padding of the last vech tile and
    // the remaining SMALL_COLS go through v_mfma_f64_4x4x4_4b (four 4 x 4 x 4 blocks, 18 cycles) instead of a
    // fifth 16-column tile of 64-cycle MFMAs: 16 + 4 instead of 20 big MFMAs per k-step and wave.
    constexpr int PADS = 16 * NTP - KP;
    constexpr int SMALL_COLS = K + 1 - PADS;
    constexpr bool SPLIT = EM && NW == 4 && SMALL_COLS > 0 && SMALL_COLS <= 4 && PADS > 0;
    constexpr int NTMB = SPLIT ? NTP : NTM;  // big (16-column) mask-side tiles
    // P4I8: the mask-side statistics contraction S, U, totals += Mask^T [wP | wz | w] on the int8 MFMA, two tiles
    // (64 samples = one full K depth) at a time: see P4b below.
    // Built and parity-green (tools/fuzz_gpu.py: 3e-13), measured at 72.7 vs 76.0 EM it/s for the fp64 form, so it is
    // compiled in only with -DPPCA_P4_INT8=1: its 1.1 k cycles of int8 MFMAs per tile replace 8.8 k of fp64 MFMAs,
    // but staging (2.5 k), digit cutting (3.0 k) and the fold with its barriers (6.9 k) no longer run in the shadow of
    // anything (phase timing, N = 2 M) -- 12.4 k against the fp64 form's 12.0 k for the same work.
    constexpr bool P4I8 = EM && NW == 4 && GI8 && (PPCA_P4_INT8 != 0);
    constexpr int NC = KP + K + 1;         // statistic columns [wP | wz | w]
    constexpr int NCT = (NC + 15) / 16;    // 16-column tiles of them (<= NTM)
    constexpr int NCOL = 16 * NCT;
    constexpr int QW = 7;                  // signed 8-bit digits of a 56-bit fixed-point form
    constexpr int QHEAD = 3;               // binary orders kept free above the first tile's column maximum
    static_assert(!P4I8 || (NCT <= NTM && QW * 2 * NCOL * 16 <= B * GS * 8 && (NCOL / 2) <= 64),
                  "digit planes of a tile fit one [G | b] buffer; one wave digitises NCOL / 2 items");
    static_assert(NW == 4 || NW == 8, "4 or 8 waves");
    static_assert(!GI8 || (NW == 4 && NTP <= 4), "int8 Gram: one wave per packed-column tile");
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *Xs = sm + cfg::OFF_X;
    double *Cs = sm + cfg::OFF_C;
    double *Gp = sm + cfg::OFF_G;
    double *Ws = sm + cfg::OFF_W;
    unsigned long long *Ms = reinterpret_cast<unsigned long long *>(sm + cfg::OFF_M);
    double *xxs = sm + cfg::OFF_S;

    const int *prows = p.rows;
    const double *pw = p.w;
    const int *pndev = p.n_dev;
    const int64_t pn = p.n;
    int fb_mode = 0;  // second stage of a guarded EM pass: reduce_wguard_kernel's verdict (Gram flags or the W-side check)
    if (!GI8 && p.runflag) {
        fb_mode = *p.runflag;
        if (fb_mode == 0) return;
    } else if (p.qflag) {  // Gram engine chosen per model by qprep's dynamic-range guard: exactly one of the two variants runs
        int unsafe = 0;
#pragma unroll
        for (int t = 0; t < NTP; ++t) unsafe |= p.qflag[t];
        if (GI8 == (unsafe != 0)) return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: row bases stay in SGPRs
    const int l15 = lane & 15, l4 = lane >> 4;
    const int d = p.d;
    const int64_t n = pndev ? (int64_t)*pndev : pn;
    const double *mC = p.model + MODEL_HDR;
    const double *mMean = mC + (int64_t)d * K;
    const double s2 = p.model[1], lnsig = p.model[2];

    for (int idx = tid; idx < cfg::DP * CS; idx += THREADS) {
        int j = idx / CS, a = idx - j * CS;
        Cs[idx] = (j < d && a < K) ? mC[(int64_t)j * K + a] : 0.0;
    }
    for (int idx = tid; idx < B * WS; idx += THREADS) Ws[idx] = 0.0;

    double muo[4];  // means in the output pass's lane map (dim = 64 wave + 16 c4 + lane % 16)
    if constexpr (!EM) {
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const int j = 64 * (wave & 3) + 16 * c4 + l15;
            muo[c4] = (j < d) ? mMean[j] : 0.0;
        }
    }
    // Staging lane map: a row arrives as two 16-byte loads per lane, lane l holding dims 128 h + 2 l + e
    // (element q = 2 h + e): half as many memory and LDS-store instructions as 8-byte loads of dims 64 q + l.
    double mu[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int j = 128 * (q >> 1) + 2 * lane + (q & 1);
        mu[q] = (j < d) ? mMean[j] : 0.0;
    }
    // (a, b) of packed column c = 16 t + (lane & 15); pad columns point at the zero column K
    int pa[NTP], pb[NTP];
#pragma unroll
    for (int t = 0; t < NTP; ++t) {
        int c = 16 * t + l15;
        int a = 0;
        while ((a + 1) * (a + 2) / 2 <= c) ++a;
        int b = c - a * (a + 1) / 2;
        if (c >= KP) { a = K; b = K; }
        pa[t] = a;
        pb[t] = b;
    }
    const int colb = (l15 < K) ? l15 : K;

    d4_t accM[RT][NTM];
    double accS[RT];  // SPLIT: the 4-column group, D lane = 16 i + 4 block + j (dim 4 block + i of the row tile, column j)
    d4_t accX[RT];
    if constexpr (EM) {
#pragma unroll
        for (int r = 0; r < RT; ++r) {
#pragma unroll
            for (int t = 0; t < NTM; ++t) accM[r][t] = d4_t{0, 0, 0, 0};
            accS[r] = 0.0;
            accX[r] = d4_t{0, 0, 0, 0};
        }
    }
    // unweighted EM pass: sum_i ln det M_i = ln prod_i det M_i, so each solver lane keeps a running
    // (mantissa, exponent) product and takes ONE logarithm at the end of the kernel (a per-tile fp64 log
    // cost 2.2k of ~35k cycles per tile: its constants live in scratch at this register pressure)
    double *scl = sm + cfg::OFF_L;
    // PAIRS: the M^-1 columns are computed two at a time, by lane i (column 2p) and lane i + 32 (column 2p + 1)
    constexpr bool PAIRS = EM && NW == 4 && K >= 2;
    constexpr int SQW = PAIRS ? 2 * B : B;  // sq slots per wave
    constexpr int L_DEV = NW * SQW, L_LLK = L_DEV + B, L_W = L_DEV + 2 * B, L_NE = L_DEV + 3 * B, L_PM = L_DEV + 4 * B,
                  L_PX = L_DEV + 5 * B;
    for (int idx = tid; idx < L_DEV + 6 * B; idx += THREADS) scl[idx] = (idx >= L_PM && idx < L_PX) ? 1.0 : 0.0;
    const double inv_s2 = 1.0 / s2;

    const int64_t ntiles = (n + B - 1) / B;
    double xr[RPW][4];
    // Row loads are unconditional (clamped to real rows; nothing between issue and first use, so a whole
    // tile's loads stay in flight); out-of-range rows / dims are masked when consumed in P1.
    // observed <=> |x| < lim: +inf for a real dimension (finite test, dataset.rs:19-22), -1 for the padding past d
    double lim[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) lim[q] = (128 * (q >> 1) + 2 * lane + (q & 1) < d) ? __builtin_inf() : -1.0;
    // Each workgroup walks a CONTIGUOUS run of tiles (consecutive 64 KB pieces of X share pages, unlike a
    // grid-strided walk that starts every tile 16 MB further on).
    const int64_t tiles_per_wg = (ntiles + gridDim.x - 1) / gridDim.x;
    int64_t tile_begin = (int64_t)blockIdx.x * tiles_per_wg;
    int64_t tile_end = tile_begin + tiles_per_wg < ntiles ? tile_begin + tiles_per_wg : ntiles;
    if (!GI8 && fb_mode == 2) {
        // Only the slices of the flagged workgroups of the guarded launch (same grid: slice g = tiles [g tpw, (g + 1) tpw)), each
        // split over gridDim.x / n_flagged workgroups of this one: a workgroup's run stays inside ONE slice, so everything below
        // (descriptors relative to the run's first row, weights, gather list) is the plain pass's.
        const int nfl = p.runflag[QF_NFLAGGED - QF_MODE];
        const int wps = (int)gridDim.x / nfl, f = (int)blockIdx.x / wps, sub = (int)blockIdx.x - f * wps;
        if (f >= nfl) {
            tile_begin = tile_end = 0;
        } else {
            const int64_t s0 = (int64_t)p.who[f] * tiles_per_wg, s1 = s0 + tiles_per_wg < ntiles ? s0 + tiles_per_wg : ntiles;
            const int64_t per = (s1 - s0 + wps - 1) / wps;
            tile_begin = s0 + (int64_t)sub * per;
            tile_end = tile_begin + per < s1 ? tile_begin + per : s1;
            if (tile_begin > tile_end) tile_begin = tile_end;
        }
    }
    const int64_t nleft = n - tile_begin * B;
    const int nrel = (int)(nleft < (1 << 30) ? nleft : (1 << 30));  // rows from the workgroup's first row to the end
    const double *Xwg = p.X + tile_begin * B * p.ldx;
    // real rows of the workgroup's own tiles, counted from its first row
    const int64_t own = (tile_end - tile_begin) * B;
    const int nmine = tile_end > tile_begin ? (int)(own < nleft ? own : nleft) : 0;
    constexpr bool CAN_GATHER = EM && NW == 4 && (GATHER || !GI8);
    const int *rows_wg = (CAN_GATHER && prows) ? prows + tile_begin * B : nullptr;
    const int lane_entry = lane;
    // Row loads are buffer loads through ONE descriptor per tile (base = the tile's first row, extent = its real rows;
    // rows are contiguous, ldx == d): the row is a scalar offset, the lane offset one constant VGPR, the half an
    // immediate -- no vector address arithmetic and no per-row scalar work.  Rows past n read as zeros: "observed",
    // but weighted with zero or skipped by every consumer.  The gathered pass (PassArgs::rows: sample i = physical
    // row rows[i]) builds a descriptor per row from one scalar load.
    const int rowbytes = d * (int)sizeof(double);
    auto tile_rsrc = [&](int64_t tile) {
        const int rel0 = (int)(tile - tile_begin) * B;
        int cnt = nrel - rel0;
        cnt = __builtin_amdgcn_readfirstlane(cnt < 0 ? 0 : (cnt > B ? B : cnt));  // (keeps the descriptor scalar)
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(Xwg + (int64_t)rel0 * p.ldx), 0, cnt * rowbytes, 0x00020000);
    };
    auto load_row = [&](const __amdgpu_buffer_rsrc_t &trs, int64_t tile, int r) {
        typedef unsigned u4_t __attribute__((ext_vector_type(4)));
        if constexpr (CAN_GATHER) {
            if (rows_wg) {
                const int rel = (int)(tile - tile_begin) * B + wave * RPW + r;
                const int rc = rel < nrel ? rel : nrel - 1;
                const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<double *>(p.X + (int64_t)rows_wg[rc] * p.ldx), 0, rowbytes, 0x00020000);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const u4_t v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, lane_entry * 16, 1024 * h, 0);
                    xr[r][2 * h] = __longlong_as_double(((long long)v[1] << 32) | v[0]);
                    xr[r][2 * h + 1] = __longlong_as_double(((long long)v[3] << 32) | v[2]);
                }
                return;
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {  // validity is applied by the consumers (P1: lim; P3 / outputs: row < n)
            const u4_t v = __builtin_amdgcn_raw_buffer_load_b128(trs, lane_entry * 16, (wave * RPW + r) * rowbytes + 1024 * h, 0);
            xr[r][2 * h] = __longlong_as_double(((long long)v[1] << 32) | v[0]);
            xr[r][2 * h + 1] = __longlong_as_double(((long long)v[3] << 32) | v[2]);
        }
    };
    auto load_tile = [&](int64_t tile) {
        const __amdgpu_buffer_rsrc_t trs = tile_rsrc(tile);
#pragma unroll
        for (int r = 0; r < RPW; ++r) load_row(trs, tile, r);
    };
    if (tile_begin < tile_end) load_tile(tile_begin);
    // int8 Gram: the slice table of this wave's column tile (QS x 4 fragments of 16 B per lane, L2-resident)
    // streams through two register sets in four digit pairs, high to low.  The table does not depend on the
    // tile, so pair {7,6} of the NEXT tile is requested at the start of P4 and is long there when P2 begins;
    // {5,4} is requested first thing in P2 (behind the mask-byte expansion), {3,2} and {1,0} as soon as a
    // register set is free, and they land behind the fp64 b = X~ C loop.  Buffer loads: one scalar resource for the table, a scalar offset per fragment
    // and one lane offset register.  (The wave's table base is made opaque per use: 32 loop-invariant
    // scalar offsets would be hoisted, overflow the SGPR file and come back through v_readlane + s_nop 4.)
    static_assert(!GI8 || QS == 8, "digit grouping below assumes 8 slices");
    const __amdgpu_buffer_rsrc_t qrsrc = __builtin_amdgcn_make_buffer_rsrc(p.qtab, 0, (int)qtab_bytes<K>(), 0x00020000);
    // Waves without a column tile (k' <= 48) run the same loads (out of range of the table: zeros) and MFMAs
    // and skip only the final store: a run-time condition around the loads would make the fragment registers
    // look live around the whole tile loop.
    const bool gram_wave = GI8 && (NTP >= NW || wave < NTP);  // compile-time true when every wave owns a tile
    constexpr bool PREFETCH_A = EM;  // (no P4 in the output passes: they request {7,6} at the start of P2 as well)
    i4_t qbA[2][4];
    auto load_pair = [&](i4_t(&dst)[2][4], int sl0) {
        int qbase = wave * QS * 4 * 1024;
        asm volatile("" : "+s"(qbase));
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                typedef unsigned u4_t __attribute__((ext_vector_type(4)));
                // (the k-chunk rides in the instruction's immediate offset: one scalar offset per slice, not per fragment)
                const u4_t v = __builtin_amdgcn_raw_buffer_load_b128(qrsrc, lane_entry * 16 + kc * 1024, qbase + (sl0 + u) * 4096, 0);
                dst[u][kc] = i4_t{(int)v[0], (int)v[1], (int)v[2], (int)v[3]};
            }
    };
    // Output passes: no statistics accumulators, so the wave's whole slice of the table (QS x 4 fragments = 128
    // registers) is loaded once per workgroup and stays resident -- no table traffic inside the tile loop.
    constexpr bool RESIDENT = GI8 && !EM;
    i4_t qt[RESIDENT ? QS : 1][4];
    if constexpr (RESIDENT) {
#pragma unroll
        for (int sl = 0; sl < QS; sl += 2) load_pair(*reinterpret_cast<i4_t(*)[2][4]>(&qt[sl]), sl);
    }
    if constexpr (GI8) {
        if (PREFETCH_A) load_pair(qbA, 6);
    }
#ifdef PPCA_PHASE_TIMING
    long long tph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = clock64();
#define PPCA_STAMP(i) { long long tn = clock64(); tph[i] += tn - tlast; tlast = tn; }
#elif defined(PPCA_MARKS)  // tools/devbuild.py -DPPCA_MARKS --asm: phase boundaries as comments in the ISA listing
#define PPCA_STAMP(i) asm volatile("; PPCA_MARK " #i ::: "memory");
#else
#define PPCA_STAMP(i)
#endif
    // ---- P1 as three pieces, so that the EM pass can run it for tile t+1 inside tile t's P4.
    // Wave-uniform results (mask words, popcounts, row sums) are gathered into the lane that will store
    // them -- lane 4r+q keeps mask word q of row r, lane r keeps xx_r / m_r (v_writelane drops each scalar
    // into its lane: 32 "lane == c" compare masks would overflow the SGPR file) -- so the whole wave does
    // ONE compact store per array.
    int st_wlo = 0, st_whi = 0;
    int st_mb[4] = {0, 0, 0, 0};  // P4I8: this lane's four dims over the wave's eight staged samples (row r at bit 7 - r)
    auto stage_begin = [&]() {
        st_wlo = st_whi = 0;
        st_mb[0] = st_mb[1] = st_mb[2] = st_mb[3] = 0;
    };
    // One row = 7 small pieces that P4 spreads over a k-step: per half h of the row, two "classify + centre" pieces
    // (elements 2h, 2h+1) and one "file" piece (mask words, x~ pair, squares); 6 = the row's |x~|^2.
    // (the EM pass needs only the weighted SUM of the |x~_i|^2 -- sigma^2 and the total llk are linear in it --
    //  so it keeps one running per-lane sum, reduced once per kernel; the output passes keep the eight per-lane
    //  partials of a wave's rows and sum them together in stage_end)
    constexpr int STAGE_PIECES = 7;
    double xx_run = 0.0;
    double pxx[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    static_assert(EM || RPW == 8, "output passes: eight staged rows per wave (store_row_sums)");
    double pc_xt0 = 0.0, pc_xt1 = 0.0, pc_xx = 0.0;
    unsigned long long pc_b0 = 0ull, pc_b1 = 0ull;
    auto stage_piece = [&](int64_t t, int lane, auto r_tag, auto p_tag) {
        constexpr int r = decltype(r_tag)::value, P = decltype(p_tag)::value;
        const int ri = wave * RPW + r;
        if constexpr (P == 0 || P == 1 || P == 3 || P == 4) {  // classify + centre element q = 2 h + e
            constexpr int q = (P < 2) ? P : P - 1, e = q & 1;
            if constexpr (q == 0) {
                pc_xx = 0.0;
            }
            // Rows past n (the clamped last row again) and the tile staged behind the workgroup's last P4 (another
            // workgroup's, or none) are staged like any other: every consumer weighs them with zero or skips them.
            // The compare's wave mask IS the ballot and feeds the select directly -- no scalar instruction between.
            const double v = xr[r][q];
            const bool ob = __builtin_fabs(v) < lim[q];
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(ob);
            const double xt = ob ? v - mu[q] : 0.0;  // select, never multiply (utils.rs:118-127)
            if constexpr (e == 0) {
                pc_b0 = bal;
                pc_xt0 = xt;
            } else {
                pc_b1 = bal;
                pc_xt1 = xt;
            }
        } else if constexpr (P == 2 || P == 5) {  // file half h: mask words, x~ pair, sums
            constexpr int h = (P == 2) ? 0 : 1;
            // mask word 2 h + e of the row = the ballot of element 2 h + e (bit l <-> dim 128 h + 2 l + e); the
            // readers index it that way (P2: any order of a sum, qprep lays the table out to match; P4b / output
            // pass: word by the parity of the dimension)
            if constexpr (P4I8) {
                file_mask<4 * r + 2 * h>(st_wlo, st_whi, st_mb[2 * h], pc_b0);
                file_mask<4 * r + 2 * h + 1>(st_wlo, st_whi, st_mb[2 * h + 1], pc_b1);
            } else {
                writelane_mask<4 * r + 2 * h>(st_wlo, st_whi, pc_b0);
                writelane_mask<4 * r + 2 * h + 1>(st_wlo, st_whi, pc_b1);
            }
            typedef double d2_t __attribute__((ext_vector_type(2)));
            *reinterpret_cast<d2_t *>(Xs + ri * XS + 128 * h + 2 * lane) = d2_t{pc_xt0, pc_xt1};  // 16-byte aligned
            pc_xx += pc_xt0 * pc_xt0;
            pc_xx += pc_xt1 * pc_xt1;
        } else if constexpr (P == 6) {
            if constexpr (!EM) pxx[r] = pc_xx;  // the per-sample |x~|^2: the eight rows are summed together in stage_end
            if constexpr (EM) {
                // wave-uniform: a real row of one of THIS workgroup's tiles (32-bit compare on the scalar unit)
                const bool mine = (int)(t - tile_begin) * B + ri < nmine;
                const double wr = mine ? (pw ? pw[t * B + ri] : 1.0) : 0.0;  // (scalar load)
                xx_run += wr * pc_xx;
            }
        }
    };
    auto stage_row = [&](int64_t t, int lane, auto r_tag) {
        static_for<STAGE_PIECES>([&](auto p_tag) { stage_piece(t, lane, r_tag, p_tag); });
    };
    auto stage_end = [&](int lane, int par) {
        const unsigned long long myw = ((unsigned long long)(unsigned)st_whi << 32) | (unsigned)st_wlo;
        if (lane < 4 * RPW) Ms[par * 4 * B + wave * 4 * RPW + lane] = myw;
        if constexpr (!EM) store_row_sums(pxx, lane, xxs + wave * RPW);
        if constexpr (P4I8) {
            // sample masks per dimension for the int8 contraction: byte (4 par + wave) of dimension j's 64-bit word =
            // its observed flags over this wave's eight samples of the group's tile `par`
            unsigned char *mbb = reinterpret_cast<unsigned char *>(sm + cfg::OFF_MB);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                mbb[(128 * (q >> 1) + 2 * lane + (q & 1)) * 8 + 4 * par + wave] = (unsigned char)st_mb[q];
        }
    };
    if constexpr (EM) {
        if (tile_begin < tile_end) {
            stage_begin();
            static_for<RPW>([&](auto r_tag) { stage_row(tile_begin, lane, r_tag); });
            stage_end(lane, 0);
        }
        __syncthreads();
    }
    for (int64_t tile = tile_begin; tile < tile_end; ++tile) {
        // The lane index is made opaque once per tile: everything derived from it (LDS addresses, shift
        // counts, column maps) is then recomputed per tile -- a few integer ops -- instead of being
        // hoisted out of the loop by LICM and parked in (spilled) registers for the whole kernel.
        int lane = lane_entry;
        asm volatile("" : "+v"(lane));
        const int l15 = lane & 15, l4 = lane >> 4;
        const int colb = (l15 < K) ? l15 : K;
        // ------------------------------------------------------------ P1
        // EM: the tile was staged behind the previous tile's P4 (or before the loop for the first one)
        if constexpr (!EM) {
            stage_begin();
            static_for<RPW>([&](auto r_tag) { stage_row(tile, lane, r_tag); });
            PPCA_STAMP(5)
            stage_end(lane, 0);
            __syncthreads();
        }
        const unsigned long long *Msc = Ms + (EM ? (int)((tile - tile_begin) & 1) * 4 * B : 0);
        PPCA_STAMP(0)
        // ------------------------------------------------------------ P2
        {
            const int rt = wave & 1, kq = wave >> 1;
            const int si = 16 * rt + l15;
            // fp64 Gram: the lane's dims DPS kq + 4 s + l4 all sit in ONE mask word (parity l4 & 1), bits 2 s apart
            const unsigned long long mwsel = Msc[si * 4 + 2 * ((DPS * kq) >> 7) + (l4 & 1)];
            const int mwbit = (((DPS * kq) & 127) >> 1) + (l4 >> 1);
            d4_t acc[NTM];
#pragma unroll
            for (int t = 0; t < NTM; ++t) acc[t] = d4_t{0, 0, 0, 0};
            // lane-constant base pointers; after full unrolling every LDS read below is
            // base + immediate offset (no per-step address arithmetic)
            const double *xrow = Xs + si * XS + DPS * kq + l4;
            const double *crow = Cs + (DPS * kq + l4) * CS;
            const double *cpa[NTP], *cpb[NTP];
#pragma unroll
            for (int t = 0; t < NTP; ++t) {
                cpa[t] = crow + pa[t];
                cpb[t] = crow + pb[t];
            }
            const double *cpc = crow + colb;
            // A = mask bytes: lane (sample = 16 rt + l15, dims 64 kc + 16 l4 .. +15); 4 bits -> 4 bytes
            // by one multiply: (x * 0x204081) & 0x01010101 puts bit i of x into byte i.
            i4_t af[GI8 ? 2 : 1][4];
            double v[2][4];
            // one digit pair: contract, then fold the (exact) integer digit sums -- |sum| <= 2^14, so two
            // digits fit one i32 with room to spare -- into the running fp64 value, Horner in 128^2
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
                        const int part = ia[1][r] * QBASE + ia[0][r];
                        v[rt2][r] = first ? (double)part : v[rt2][r] * (double)(QBASE * QBASE) + (double)part;
                    }
                }
            };
            double qs = 0.0;
            i4_t qbB[2][4];
            if constexpr (GI8) {
                {
                    // (the mask words are requested from LDS first, so that their latency runs under the issue of
                    //  the table loads)
                    unsigned long long mwd[2][4];
#pragma unroll
                    for (int rt2 = 0; rt2 < 2; ++rt2)
#pragma unroll
                        for (int kc = 0; kc < 4; ++kc) mwd[rt2][kc] = Msc[(16 * rt2 + l15) * 4 + kc];
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (!RESIDENT) {
                        if (!PREFETCH_A) load_pair(qbA, 6);
                        load_pair(qbB, 4);
                    }
                    if (gram_wave) qs = p.qscale[16 * wave + l15];
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
                    PPCA_STAMP(12)
                    if constexpr (RESIDENT) {
                        group(qt + 6, true);
                        group(qt + 4, false);
                    } else {
                        group(qbA, true);   // digits {7,6}: requested during the previous P4
                        load_pair(qbA, 2);
                        group(qbB, false);  // digits {5,4}
                        load_pair(qbB, 0);
                    }
                }
            }
            PPCA_STAMP(13)
            if constexpr (GI8) {
                // b = X~ C alone: operands of the next four k-steps are requested before the current four
                // MFMAs issue (hipcc otherwise reads each pair right before its MFMAs and waits on LDS)
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
                    __builtin_amdgcn_sched_barrier(0);  // (without the fences the reads sink back to their uses)
#pragma unroll
                    for (int u = 0; u < CH; ++u) acc[NTP] = mfma(axb[c & 1][u], cbb[c & 1][u], acc[NTP]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int s = 0; s < (GI8 ? 0 : STEPS); ++s) {
                const double ax = xrow[4 * s];
                if constexpr (!GI8) {
                    const double am = ((mwsel >> (mwbit + 2 * s)) & 1ull) ? 1.0 : 0.0;
#pragma unroll
                    for (int t = 0; t < NTP; ++t)
                        acc[t] = mfma(am, cpa[t][4 * s * CS] * cpb[t][4 * s * CS], acc[t]);
                }
                acc[NTP] = mfma(ax, cpc[4 * s * CS], acc[NTP]);
            }
            PPCA_STAMP(14)
            if constexpr (GI8) {
                if constexpr (RESIDENT) {
                    group(qt + 2, false);
                    group(qt + 0, false);
                } else {
                    group(qbA, false);  // digits {3,2}
                    group(qbB, false);  // digits {1,0}
                }
                if (gram_wave) {
#pragma unroll
                    for (int rt2 = 0; rt2 < 2; ++rt2)
#pragma unroll
                        for (int r = 0; r < 4; ++r)  // C/D map of the 16x16 integer MFMA: row = 4 (lane >> 4) + reg
                            Gp[(16 * rt2 + 4 * l4 + r) * GS + 16 * wave + l15] = v[rt2][r] * qs;
                }
            }
            // K-split partials -> two buffers, summed in a fixed order (deterministic):
            // G = (p0 [+ p2]) + (p1 [+ p3]); the bracketed terms are added in place by their owner
            double *g = Gp + (kq & 1) * B * GS;
            constexpr int T0 = GI8 ? NTP : 0;  // int8 Gram: only the b tile comes from the fp64 accumulators
            if constexpr (P4I8) {
                // (the second buffer holds digit planes: the second K-half of b goes to its own compact array)
                double *p1 = sm + cfg::OFF_P1;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (kq == 0) g[(16 * rt + l4 + 4 * r) * GS + 16 * NTP + l15] = acc[NTP][r];
                    else if (l15 < K + 1) p1[(16 * rt + l4 + 4 * r) * (K + 1) + l15] = acc[NTP][r];
                }
            } else if (kq < 2) {
#pragma unroll
                for (int t = T0; t < NTM; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) g[(16 * rt + l4 + 4 * r) * GS + 16 * t + l15] = acc[t][r];
            }
            if constexpr (KS == 4) {
                __syncthreads();
                if (kq >= 2) {
#pragma unroll
                    for (int t = T0; t < NTM; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) g[(16 * rt + l4 + 4 * r) * GS + 16 * t + l15] += acc[t][r];
                }
            }
        }
        __syncthreads();
        PPCA_STAMP(1)
        if constexpr (!EM) load_tile(tile + 1);  // unconditional: rows are clamped, see load_tile
        // ------------------------------------------------------------ P3
        // Every wave factors every sample (lane = sample, redundantly, in parallel) and the waves
        // share the independent columns of M^-1; wave 0 also owns z, llk and the scalars.
        if (PAIRS || lane < B) {
            const int i = lane & (B - 1);
            const int hi = PAIRS ? lane >> 5 : 0;  // which column of a pair this half of the wave solves for
            const int64_t row = tile * B + i;
            const double *g0 = Gp + i * GS;
            const double *g1 = g0 + B * GS;
            const double wgt = (row < n) ? (pw ? pw[row] : 1.0) : 0.0;
            // observed count of the sample: popcount of its four mask words (the padding past d is never set)
            const int m = __popcll(Msc[i * 4]) + __popcll(Msc[i * 4 + 1]) + __popcll(Msc[i * 4 + 2]) + __popcll(Msc[i * 4 + 3]);
            double *wrow = Ws + i * WS;
            double sc_sq = 0.0, sc_dev = 0.0, sc_llk = 0.0, sc_w = 0.0, sc_ne = 0.0;  // this tile's terms
            const double sq_run = scl[wave * SQW + (PAIRS ? lane : i)];
            Posterior<K> post;
            double pm;
            int pe;
            post.factor([&](int e) { return GI8 ? g0[e] : g0[e] + g1[e]; }, s2, pm, pe);
            PPCA_STAMP(8)
            double z[K], quad, zz;
            const double *p1row = sm + cfg::OFF_P1 + i * (K + 1);
            post.solve([&](int a) { return g0[16 * NTP + a] + (P4I8 ? p1row[a] : g1[16 * NTP + a]); }, z, quad, zz);
            PPCA_STAMP(9)
            double trpart = 0.0;
            // llk / llks / states / smooth / extrapolate need z only: the posterior covariance (the M^-1 columns)
            // is computed when somebody reads it -- covariances out, or the covariance diagonals
            const bool need_cov = EM || p.covs != nullptr || (p.recon != nullptr && p.recon_mode >= 2);
            if constexpr (PAIRS) {
#pragma unroll
                for (int pp = 0; pp < (K + 1) / 2; ++pp) {
                    if (pair_owner(K, pp, NW) != wave) continue;
                    const int c0 = 2 * pp;
                    const double zc = (hi && c0 + 1 < K) ? z[c0 + 1 < K ? c0 + 1 : c0] : z[c0];
                    // P = z z^T + Sigma, Sigma = sigma^2 M^-1 (ppca_model.rs:437-439), weighted;
                    // tri(t, c0) + 1 = tri(t, c0 + 1)
                    trpart += post.minv_column_pair(c0, hi, [&](int t, double v, bool ok) {
                        if (ok && c0 + hi < K) wrow[tri(t, c0) + hi] = wgt * (z[t] * zc + s2 * v);
                    });
                }
            }
#pragma unroll
            for (int c = 0; c < (PAIRS ? 0 : K); ++c) {
                if (column_owner(K, c, NW) != wave || !need_cov) continue;
                if constexpr (EM) {
                    // P = z z^T + Sigma, Sigma = sigma^2 M^-1 (ppca_model.rs:437-439), weighted
                    trpart += post.minv_column(
                        c, [&](int a, int cc, double v) { wrow[tri(a, cc)] = wgt * (z[a] * z[cc] + s2 * v); });
                } else {
                    trpart += post.minv_column(c, [&](int a, int cc, double v) {
                        const double sv = s2 * v;
                        wrow[K + tri(a, cc)] = sv;  // post mode: W row = [z (K) | Sigma packed (K')]
                    });
                }
            }
            PPCA_STAMP(10)
            if constexpr (EM) {
                // tr(C_o Sigma C_o^T) = <Sigma, G> = s2 (K - s2 tr M^-1)  (:345); each wave carries the
                // share of its columns.  All-masked samples are filtered out of the noise sums (:333).
                if (m > 0) sc_sq -= wgt * s2 * s2 * trpart;
            }
            if (wave == 0 && hi == 0) {
                const double xx = EM ? 0.0 : xxs[i];  // EM: the |x~|^2 terms are added once, in the epilogue
                // running sums: requested at the top of the block, needed at its end
                const double run_dev = scl[L_DEV + i], run_llk = scl[L_LLK + i], run_w = scl[L_W + i], run_ne = scl[L_NE + i];
                const double run_pm = scl[L_PM + i], run_px = scl[L_PX + i];
                if constexpr (EM) {
                    double *zrow = wrow + 16 * NTP;  // W row = [w P (K') | 0.. | w z (K) | w | 0..]
#pragma unroll
                    for (int a = 0; a < K; ++a) zrow[a] = wgt * z[a];
                    zrow[K] = wgt;
                    if constexpr (SPLIT && !P4I8) {  // the leading [w z | w] columns again, in the padding behind vech(P)
#pragma unroll
                        for (int a = 0; a < PADS; ++a) wrow[KP + a] = (a < K) ? wgt * z[a] : wgt;
                    }
                    if (m > 0) {
                        sc_sq += wgt * s2 * (double)K;
                        sc_dev += wgt * (xx - quad - s2 * zz);  // |x~ - C_o z|^2  (:346)
                        sc_ne += (row < n) ? 1.0 : 0.0;
                    }
                } else {
#pragma unroll
                    for (int a = 0; a < K; ++a) wrow[a] = z[a];
                }
                if constexpr (EM) {
                    const double lk0 = sample_llk_nolog(xx, quad, inv_s2, lnsig, m, K);
                    if (pw) {
                        if (!p.no_llk) sc_llk += wgt * (m > 0 ? lk0 - 0.5 * Posterior<K>::logdet(pm, pe) : 0.0);
                    } else {
                        const bool use = m > 0 && row < n;  // wgt is 1 for real rows
                        sc_llk += use ? lk0 : 0.0;
                        int e;
                        scl[L_PM + i] = frexp(run_pm * (use ? pm : 1.0), &e);
                        scl[L_PX + i] = run_px + (double)(e + (use ? pe : 0));
                    }
                    sc_w += wgt;
                } else {
                    const double lk = sample_llk(xx, quad, Posterior<K>::logdet(pm, pe), s2, lnsig, m, K);
                    sc_llk += wgt * lk;
                    sc_w += wgt;
                    if (p.llks && row < n) p.llks[row] = lk;
                }
                scl[L_DEV + i] = run_dev + sc_dev;
                scl[L_LLK + i] = run_llk + sc_llk;
                scl[L_W + i] = run_w + sc_w;
                scl[L_NE + i] = run_ne + sc_ne;
            }
            if constexpr (EM) scl[wave * SQW + (PAIRS ? lane : i)] = sq_run + sc_sq;
        }
        PPCA_STAMP(11)
        __syncthreads();
        PPCA_STAMP(2)
        if constexpr (!EM) {
            // InferredMasked outputs (src/python_bindings.rs:211-234): the tile's states (32 x k) and covariances
            // (32 x k x k) are contiguous in the output arrays, so they are written from the W rows by the whole
            // workgroup with consecutive lanes on consecutive addresses (not 8 bytes per lane 8 k^2 bytes apart)
            if (p.states) {
                for (int idx = tid; idx < B * K; idx += THREADS) {
                    const int i = idx / K, a = idx - i * K;
                    const int64_t row = tile * B + i;
                    if (row < n) p.states[row * K + a] = Ws[i * WS + a];
                }
            }
            if (p.covs) {
                for (int idx = tid; idx < B * K * K; idx += THREADS) {
                    const int i = idx / (K * K), rem = idx - i * (K * K), a = rem / K, c = rem - a * K;
                    const int64_t row = tile * B + i;
                    if (row < n) p.covs[row * (K * K) + rem] = Ws[i * WS + K + (a >= c ? tri(a, c) : tri(c, a))];
                }
            }
        }
        if constexpr (EM) {
            // -------------------------------------------------------- P4
            // (a) cross/sumx += X~^T [wz | w]: the only reader of the x~ tile; the next tile's rows are
            //     requested one per k-step behind these MFMAs (a burst of all of them stalls the wave
            //     ~2.9k cycles on the memory queue: a CU drains ~10 B/cycle)
            PPCA_STAMP(6)
            {
                // operands of step s+1 are read from LDS before the MFMAs of step s issue (fenced: hipcc otherwise
                // sinks the reads to their uses and waits on LDS in front of every step)
                double bzb[2], axb[2][RT];
                const __amdgpu_buffer_rsrc_t trs = tile_rsrc(tile + 1);
                bzb[0] = Ws[l4 * WS + 16 * NTP + l15];
#pragma unroll
                for (int r = 0; r < RT; ++r) axb[0][r] = Xs[l4 * XS + DW * wave + 16 * r + l15];
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    if (s < RPW) load_row(trs, tile + 1, s);  // unconditional; two rows per step measured slower
                    if (s + 1 < 8) {
                        const int smp = 4 * (s + 1) + l4;
                        bzb[(s + 1) & 1] = Ws[smp * WS + 16 * NTP + l15];
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
            PPCA_STAMP(4)
            // (b) S/U/totals += Mask^T [wP | wz | w], with the staging (P1) of the next tile's rows between the MFMAs
            stage_begin();
            if constexpr (GI8 && !P4I8) {
                // the next tile's first digit pair (the table is tile-independent); holding the second pair
                // across P4 as well spills 73 registers
                if (PREFETCH_A) load_pair(qbA, 6);
            }
            if constexpr (P4I8) {

                // S, U, totals += Mask^T [wP | wz | w] on the int8 MFMA.  The mask operand is exactly 0 / 1; the rows of
                // [wP | wz | w] are cut into QW signed 8-bit digits of a per-column fixed-point form and contracted
                // with exact integer accumulation over a GROUP of two tiles (64 samples = the full K depth of
                // v_mfma_i32_16x16x64_i8); the 7 integer sums of an entry are recombined exactly (three digits per
                // i32) and folded into the fp64 accumulators once per group.  Scales: the group's scale of a column is
                // 2^QHEAD above the first tile's column maximum; the second tile uses it when its own maximum fits
                // (the common case), otherwise the two tiles are contracted one after the other, each with its own
                // scale (same code, the other half's mask bytes zeroed).  A non-finite column poisons its scale, so
                // NaN / inf reach the statistics as they would through fp64.
                // fp64 form: 16 + 4 MFMAs of 64 cycles per 4 samples and wave; here 7 x 20 of 16 cycles per 64.
                const int rel = (int)(tile - tile_begin);
                const int half = rel & 1;
                // digit planes of the group's first tile: second [G | b] buffer; of its second tile: the first buffer
                // (free once that tile's P3 has read its Gram).  Kept as offsets from ONE LDS base: a select between two
                // pointers would lose the address space (flat loads).
                unsigned char *wq_base = reinterpret_cast<unsigned char *>(Gp);
                constexpr int WQ_FIRST = B * GS * 8;  // byte offset of the first tile's planes
                int *Ex = reinterpret_cast<int *>(sm + cfg::OFF_E);  // [2][NCOL] column exponents (group / second tile), flag
                int *viol = Ex + 2 * NCOL;
                {
                    // ---- digitise this tile: item = (column c, 16-sample chunk), NCOL / 2 items per wave
                    const bool active = lane < NCOL / 2;
                    const int it = (NCOL / 2) * wave + (active ? lane : 0);
                    const int c = it >> 1, chunk = it & 1;
                    const bool cvalid = c < NC;
                    const int src = cvalid ? (c < KP ? c : 16 * NTP + (c - KP)) : 0;
                    double wv[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) wv[j] = Ws[(16 * chunk + j) * WS + src];
                    double m = 0.0;
                    bool fin = true;
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        wv[j] = cvalid ? wv[j] : 0.0;
                        const double av = __builtin_fabs(wv[j]);
                        fin = fin && (av < __builtin_inf());
                        m = __builtin_fmax(m, av);
                    }
                    // the other chunk of the column sits in the neighbouring lane
                    m = __builtin_fmax(m, dpp_f64<0xB1, 0xF>(m));
                    const int finw = __builtin_amdgcn_update_dpp(0, fin ? 1 : 0, 0xB1, 0xF, 0xF, true);
                    fin = fin && finw != 0;
                    int e = m > 0.0 ? __builtin_amdgcn_frexp_exp(m) : -900;  // |w| < 2^e
                    e = e < -900 ? -900 : e;
                    int E;
                    if (half == 0) {
                        E = fin ? e + QHEAD : 100000;
                        if (active && chunk == 0) {
                            Ex[c] = E;
                            Ex[NCOL + c] = E;
                        }
                        if (tid == 0) *viol = 0;
                    } else {
                        const int Eg = Ex[c];
                        const bool over = !fin || e > Eg;  // (a poisoned group scale stays poisoned)
                        E = (Eg > 5000) ? Eg : (fin ? (e > Eg ? e : Eg) : 100000);
                        if (active && chunk == 0) Ex[NCOL + c] = E;
                        if (active && over && Eg <= 5000) *viol = 1;
                    }
                    const double qsc = __builtin_ldexp(1.0, 54 - (E > 5000 ? 0 : E));
                    unsigned wlo[16], whi[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        // I = rint(w 2^(54 - E)), |I| < 2^54, as two's complement (hi, lo); + 0x80 per byte with carries,
                        // then ^ 0x80 per byte: bytes 0..6 are the signed digits, I = sum d_k 256^k exactly
                        const double xr = __builtin_rint(wv[j] * qsc);
                        const double xh = __builtin_floor(xr * 2.3283064365386963e-10);  // 2^-32
                        const int hi = (int)xh;
                        const unsigned lo = (unsigned)__builtin_fma(xh, -4294967296.0, xr);
                        unsigned long long u = (((unsigned long long)(unsigned)hi << 32) | lo) + 0x0080808080808080ull;
                        u ^= 0x0080808080808080ull;
                        wlo[j] = (unsigned)u;
                        whi[j] = (unsigned)(u >> 32);
                    }
                    // 4 x 4 byte transposes: plane k of samples 4 g .. 4 g + 3 = bytes k of their four words
                    unsigned pl[QW][4];
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        auto tr4 = [&](const unsigned *w, unsigned *o0, unsigned *o1, unsigned *o2, unsigned *o3) {
                            const unsigned t0 = __builtin_amdgcn_perm(w[1], w[0], 0x05010400u), t1 = __builtin_amdgcn_perm(w[1], w[0], 0x07030602u);
                            const unsigned u0 = __builtin_amdgcn_perm(w[3], w[2], 0x05010400u), u1 = __builtin_amdgcn_perm(w[3], w[2], 0x07030602u);
                            *o0 = __builtin_amdgcn_perm(u0, t0, 0x05040100u);
                            *o1 = __builtin_amdgcn_perm(u0, t0, 0x07060302u);
                            *o2 = __builtin_amdgcn_perm(u1, t1, 0x05040100u);
                            if (o3) *o3 = __builtin_amdgcn_perm(u1, t1, 0x07060302u);
                        };
                        tr4(wlo + 4 * g4, &pl[0][g4], &pl[1][g4], &pl[2][g4], &pl[3][g4]);
                        tr4(whi + 4 * g4, &pl[4][g4], &pl[5][g4], &pl[6][g4], nullptr);
                    }
                    if (active) {
                        unsigned char *wq = wq_base + (half == 0 ? WQ_FIRST : 0);
#pragma unroll
                        for (int sl = 0; sl < QW; ++sl)
                            *reinterpret_cast<i4_t *>(wq + ((sl * 2 + chunk) * NCOL + c) * 16) =
                                i4_t{(int)pl[sl][0], (int)pl[sl][1], (int)pl[sl][2], (int)pl[sl][3]};
                    }
                }
                PPCA_STAMP(15)
                // staging (P1) of the next tile: its rows, requested during P4a, have had the digit work to arrive; the
                // row registers are dead before the contraction starts (the mask words / sample masks gathered here
                // are stored at the end, after the contraction has read the current ones)
                __builtin_amdgcn_sched_barrier(0);
                static_for<RPW>([&](auto r_tag) { stage_row(tile + 1, lane, r_tag); });
                __builtin_amdgcn_sched_barrier(0);
                PPCA_STAMP(5)
                const bool two = half == 1;
                if (two || tile + 1 == tile_end) {  // (uniform) the group is complete
                    __syncthreads();                // digit planes, exponents, flag
                    const bool split = two && *viol != 0;
                    const unsigned long long *Mb = reinterpret_cast<const unsigned long long *>(sm + cfg::OFF_MB);
                    const int npass = split ? 2 : 1;
#pragma unroll 1
                    for (int pass = 0; pass < npass; ++pass) {  // (one copy of the code: 140 MFMAs + their folds)
                        const int sel = split ? pass : -1;      // < 0: both tiles under the group scale; 0 / 1: one tile
                        i4_t af[RT];
#pragma unroll
                        for (int r = 0; r < RT; ++r) {
                            const unsigned long long mbits = Mb[DW * wave + 16 * r + l15];
                            unsigned f = (unsigned)(mbits >> (16 * l4)) & 0xFFFFu;
                            const int lh = l4 >> 1;
                            const bool keep = sel < 0 ? (two || lh == 0) : lh == sel;
                            f = keep ? f : 0u;
                            // rows were shifted in first-to-last: row r of a byte at bit 7 - r; after the bit reversal
                            // samples 0..7 of the chunk sit at bits 24..31, samples 8..15 at bits 16..23, ascending
                            const unsigned g = __builtin_bitreverse32(f);
                            af[r][0] = (int)((((g >> 24) & 0xFu) * 0x00204081u) & 0x01010101u);
                            af[r][1] = (int)((((g >> 28) & 0xFu) * 0x00204081u) & 0x01010101u);
                            af[r][2] = (int)((((g >> 16) & 0xFu) * 0x00204081u) & 0x01010101u);
                            af[r][3] = (int)((((g >> 20) & 0xFu) * 0x00204081u) & 0x01010101u);
                        }
                        const unsigned char *wq = wq_base + ((l4 >> 1) == 0 ? WQ_FIRST : 0);
                        const int *Eh = Ex + (sel == 1 ? NCOL : 0);
#pragma unroll
                        for (int t = 0; t < NCT; ++t) {
                            const int c = 16 * t + l15;
                            const int E = Eh[c];
                            // 2^(E - 54), or a NaN for a poisoned column (built from bits: as an arithmetic select the
                            // compiler would carry the NaN case through every accumulator update)
                            const double fsc = __hiloint2double(E > 5000 ? 0x7FF80000 : (E + 969) << 20, 0);
                            i4_t bq[QW];
#pragma unroll
                            for (int sl = 0; sl < QW; ++sl)
                                bq[sl] = *reinterpret_cast<const i4_t *>(wq + ((sl * 2 + (l4 & 1)) * NCOL + c) * 16);
                            static_assert(QW == 7 && RT % 2 == 0, "mfma_i8_x7, row tiles in pairs");
#pragma unroll
                            for (int r2 = 0; r2 < RT; r2 += 2) {
                                // two row tiles at a time: 14 MFMAs in flight, then eight independent fold chains
                                i4_t ia[2][QW];
                                mfma_i8_x7(af[r2], bq, ia[0]);
                                mfma_i8_x7(af[r2 + 1], bq, ia[1]);
#pragma unroll
                                for (int u = 0; u < 2; ++u)
#pragma unroll
                                    for (int q = 0; q < 4; ++q) {  // D row = 4 l4 + q (dim 16 r + 4 l4 + q), column l15
                                        const int i1 = (((ia[u][2][q] << 8) + ia[u][1][q]) << 8) + ia[u][0][q];  // two v_lshl_add
                                        const int i2 = (((ia[u][5][q] << 8) + ia[u][4][q]) << 8) + ia[u][3][q];
                                        const double v = ((double)ia[u][6][q] * 16777216.0 + (double)i2) * 16777216.0 + (double)i1;
                                        accM[r2 + u][t][q] = __builtin_fma(v, fsc, accM[r2 + u][t][q]);
                                    }
                            }
                        }
                    }
                    __syncthreads();  // the sample masks and digit planes are free (the staging below rewrites the masks)
                }
                if (PREFETCH_A) load_pair(qbA, 6);  // the next tile's first digit pair of the Gram table
            } else {
                // mask operand of dims DW wave + 16 r + l15: word 2 (dim / 128) + parity, bit (dim % 128) / 2
                const int mword = 2 * ((DW * wave) >> 7) + (l15 & 1);
                unsigned long long mwc = Msc[l4 * 4 + mword];
                double bwc[NTMB], bsc = 0.0;
    #pragma unroll
                for (int t = 0; t < NTMB; ++t) bwc[t] = Ws[l4 * WS + 16 * t + l15];
                if constexpr (SPLIT) bsc = Ws[l4 * WS + 16 * NTP + PADS + (lane & 3)];
                // One staging piece follows each MFMA.  Measured (tools/ubench_shadow.hip): v_mfma_f64 holds the
                // SIMD's VALU port for its 64 cycles -- no VALU instruction of this wave overlaps it, only LDS,
                // SALU and memory instructions do -- so this interleave hides the staging's LDS writes and the
                // row-load latency, not its ALU work.  The B operands of step s+1 are read from LDS during step s.
                // B operand of the 4x4x4 blocks: lane = 16 k + 4 block + j -> W[sample 4 s + k][column j of the group];
                // its A operand (lane = 16 k + 4 block + i -> dim 4 block + i, sample k) is the SAME register as the
                // 16x16x4 tiles' (row = lane % 16, k = lane / 16): no extra mask expansion.
                static_for<8>([&](auto s_tag) {
                    constexpr int s = decltype(s_tag)::value;
                    unsigned long long mwn = 0ull;
                    double bwn[NTMB], bsn = 0.0;
                    constexpr int PER_R = NTMB + (SPLIT ? 1 : 0);  // MFMAs per row tile and k-step
                    constexpr int SLOTS = RT * PER_R;
                    static_for<SLOTS>([&](auto i_tag) {
                        constexpr int i = decltype(i_tag)::value, r = i / PER_R, t = i % PER_R;
                        // am = bit ? 1.0 : 0.0 in two ops: sign-extended 1-bit field (0 / -1) & high word of 1.0
                        const int sh = (((DW * wave) & 127) >> 1) + 8 * r;
                        const int am_hi = __builtin_amdgcn_sbfe((int)(unsigned)(mwc >> (sh & 32)), (sh & 31) + (l15 >> 1), 1) & 0x3FF00000;
                        const double am = __hiloint2double(am_hi, 0);
                        if constexpr (t < NTMB) {
                            accM[r][t] = mfma(am, bwc[t], accM[r][t]);
                        } else {
                            accS[r] = __builtin_amdgcn_mfma_f64_4x4x4f64(am, bsc, accS[r], 0, 0, 0);
                        }
                        if constexpr (s < RPW) {  // the row's pieces spread evenly over the step's MFMAs
                            constexpr int P0 = i * STAGE_PIECES / SLOTS, P1 = (i + 1) * STAGE_PIECES / SLOTS;
                            static_for<P1 - P0>([&](auto o_tag) {
                                stage_piece(tile + 1, lane, s_tag, std::integral_constant<int, P0 + decltype(o_tag)::value>{});
                            });
                        }
                        if constexpr (i == SLOTS * 3 / 4 && s + 1 < 8) {
                            const int smp = 4 * (s + 1) + l4;
                            mwn = Msc[smp * 4 + mword];
    #pragma unroll
                            for (int tt = 0; tt < NTMB; ++tt) bwn[tt] = Ws[smp * WS + 16 * tt + l15];
                            if constexpr (SPLIT) bsn = Ws[smp * WS + 16 * NTP + PADS + (lane & 3)];
                        }
                    });
                    if constexpr (s + 1 < 8) {
                        mwc = mwn;
    #pragma unroll
                        for (int t = 0; t < NTMB; ++t) bwc[t] = bwn[t];
                        bsc = bsn;
                    }
                });
            }
            stage_end(lane, (int)((tile + 1 - tile_begin) & 1));
#ifdef PPCA_PHASE_TIMING
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // diagnostic: charge the prefetch wait to P4
#endif
            PPCA_STAMP(7)
        } else if (p.recon) {
            // Output pass on the fp64 MFMA: out (32 samples x 256 dims) = A (32 x kk) . B (kk x 256), wave w owning
            // dims 64 w .. 64 w + 63 (four 16-dim tiles) of both 16-sample row tiles.
            //   smooth / extrapolate (ppca_model.rs:454-463): A = z (K states), B = C^T; + mean
            //   covariance diagonals (:485-508, :542-577): A = vech(Sigma_i) (K' entries), B_j = c_ja c_jb,
            //   doubled off the diagonal (c_j^T Sigma c_j over the packed half); + sigma^2
            // (the scalar form spent 2 K LDS reads per output element, 2 K^2 for the diagonals)
            static_assert(EM || NW == 4, "output pass: four waves x four dim tiles");
            constexpr int CT = 4;
            const bool dg = p.recon_mode >= 2;
            // Loads and stores share one in-order counter (vmcnt): a load consumed after a store waits for that
            // store to be acknowledged by memory.  So every load of this pass -- the observed values extrapolate
            // passes through bit-exactly -- is issued HERE, before any store of the tile, and the stores at the
            // end run back to back.
            const bool extra = p.recon_mode == 1, zero_obs = p.recon_mode == 3;
            constexpr int ROWS = B / NW;
            double xin[ROWS][4];
#pragma unroll
            for (int rr = 0; rr < ROWS; ++rr)
#pragma unroll
                for (int q = 0; q < 4; ++q) xin[rr][q] = 0.0;
            if (extra) {
#pragma unroll
                for (int rr = 0; rr < ROWS; ++rr) {
                    const int64_t row = tile * B + wave + NW * rr;
                    const double *xrow = p.X + (row < n ? row : n - 1) * p.ldx;
#pragma unroll
                    for (int q = 0; q < 4; ++q) xin[rr][q] = xrow[64 * q + lane < d ? 64 * q + lane : d - 1];
                }
            }
            d4_t oacc[2][CT];
#pragma unroll
            for (int rt2 = 0; rt2 < 2; ++rt2)
#pragma unroll
                for (int c4 = 0; c4 < CT; ++c4) oacc[rt2][c4] = d4_t{0, 0, 0, 0};
            const double *crow4[CT];
#pragma unroll
            for (int c4 = 0; c4 < CT; ++c4) crow4[c4] = Cs + (64 * wave + 16 * c4 + l15) * CS;
            if (!dg) {
#pragma unroll
                for (int s = 0; s < (K + 3) / 4; ++s) {
                    const int kx = 4 * s + l4;             // state index of this lane's operands
                    const int ka = kx < K ? kx : K - 1;    // A: any finite value (B is zero there)
                    const int kb = kx < K ? kx : K;        // B: column K of the C tile is all zeros
                    double bo[CT];
#pragma unroll
                    for (int c4 = 0; c4 < CT; ++c4) bo[c4] = crow4[c4][kb];
#pragma unroll
                    for (int rt2 = 0; rt2 < 2; ++rt2) {
                        const double ao = Ws[(16 * rt2 + l15) * WS + ka];
#pragma unroll
                        for (int c4 = 0; c4 < CT; ++c4) oacc[rt2][c4] = mfma(ao, bo[c4], oacc[rt2][c4]);
                    }
                }
            } else {
#pragma unroll
                for (int s = 0; s < (KP + 3) / 4; ++s) {
                    const int e = 4 * s + l4;              // packed index (a >= b) of this lane's operands
                    int ea = 0;
                    while ((ea + 1) * (ea + 2) / 2 <= e) ++ea;
                    int eb = e - ea * (ea + 1) / 2;
                    const double fac = (e < KP) ? (ea == eb ? 1.0 : 2.0) : 0.0;
                    if (e >= KP) { ea = K; eb = K; }       // zero column
                    const int ka = e < KP ? e : KP - 1;
                    double bo[CT];
#pragma unroll
                    for (int c4 = 0; c4 < CT; ++c4) bo[c4] = fac * crow4[c4][ea] * crow4[c4][eb];
#pragma unroll
                    for (int rt2 = 0; rt2 < 2; ++rt2) {
                        const double ao = Ws[(16 * rt2 + l15) * WS + K + ka];
#pragma unroll
                        for (int c4 = 0; c4 < CT; ++c4) oacc[rt2][c4] = mfma(ao, bo[c4], oacc[rt2][c4]);
                    }
                }
            }
            // C/D map of v_mfma_f64_16x16x4: row (sample) = l4 + 4 r, column (dim) = l15.  The results go through
            // the x~ tile (free after P2 in the output passes) and leave as whole rows: per row and wave four
            // 512-byte stores under a wave-uniform row test, the observed values of extrapolate re-read the same
            // way (bit-exact pass-through).  (Storing straight from the accumulator layout put each 8-byte store
            // in its own exec-masked branch with a vmcnt wait: up to 40x slower, depending on hipcc's mood.)
#pragma unroll
            for (int c4 = 0; c4 < CT; ++c4) {
                const double add = dg ? s2 : muo[c4];
#pragma unroll
                for (int rt2 = 0; rt2 < 2; ++rt2)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        Xs[(16 * rt2 + l4 + 4 * r) * XS + 64 * wave + 16 * c4 + l15] = oacc[rt2][c4][r] + add;
            }
            __syncthreads();
#pragma unroll
            for (int rr = 0; rr < ROWS; ++rr) {
                const int ri = wave + NW * rr;
                const int64_t row = tile * B + ri;
                const bool row_ok = row < n;  // wave-uniform
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int j = 64 * q + lane;
                    const bool obs = (Msc[ri * 4 + 2 * (q >> 1) + (lane & 1)] >> (32 * (q & 1) + (lane >> 1))) & 1ull;
                    double out = Xs[ri * XS + j];
                    if (extra) out = obs ? xin[rr][q] : out;
                    if (zero_obs) out = obs ? 0.0 : out;
                    if (row_ok && j < d) p.recon[row * (int64_t)d + j] = out;
                }
            }
        }
        __syncthreads();
        PPCA_STAMP(3)
    }
#ifdef PPCA_PHASE_TIMING
    if (p.dbg && tid == 0)
        for (int i = 0; i < 16; ++i) p.dbg[(int64_t)blockIdx.x * 16 + i] = (double)tph[i];
#endif

    // ------------------------------------------------------------ epilogue
    // scalars: deterministic reduction over the 32 solver lanes of each wave, then over the waves
    {
        const double sq_w = wave_sum(lane < SQW ? scl[wave * SQW + lane] : 0.0);
        const double xx_w = wave_sum(xx_run);
        if (lane == 0) {  // xxs is free after the last tile
            xxs[wave] = sq_w;
            xxs[NW + wave] = xx_w;
        }
    }
    __syncthreads();
    if (wave == 0) {
        double v0 = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) v0 += xxs[w];
        double xx_tot = 0.0;  // sum_i w_i |x~_i|^2 (EM pass)
#pragma unroll
        for (int w = 0; w < NW; ++w) xx_tot += xxs[NW + w];
        const int li = lane < B ? lane : 0;
        double sc_llk = scl[L_LLK + li];
        if constexpr (EM) sc_llk -= 0.5 * (log(scl[L_PM + li]) + scl[L_PX + li] * LN_2);
        double v1 = wave_sum(lane < B ? scl[L_DEV + li] : 0.0), v2 = wave_sum(lane < B ? sc_llk : 0.0),
               v3 = wave_sum(lane < B ? scl[L_W + li] : 0.0), v4 = wave_sum(lane < B ? scl[L_NE + li] : 0.0);
        if (lane == 0) {
            double *sc;
            if constexpr (EM) {
                StatsLayout L(d, K);
                sc = p.part + (int64_t)blockIdx.x * L.len + L.scalars;
            } else {
                sc = p.scal_part + (int64_t)blockIdx.x * 8;
            }
            sc[SC_SQERR] = v0;
            sc[SC_DEVSQ] = EM ? v1 + xx_tot : v1;
            sc[SC_LLK] = EM ? v2 - 0.5 * inv_s2 * xx_tot : v2;
            sc[SC_SUMW] = v3;
            sc[SC_NONEMPTY] = v4;
            sc[5] = 0.0;
            sc[6] = 0.0;
            sc[7] = 0.0;
        }
    }
    if constexpr (EM) {
        StatsLayout L(d, K);
        double *out = p.part + (int64_t)blockIdx.x * L.len;
#pragma unroll
        for (int r = 0; r < RT; ++r) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int dim = DW * wave + 16 * r + l4 + 4 * q;  // C/D row of v_mfma_f64_16x16x4
                if constexpr (P4I8) {
                    const int dimi = DW * wave + 16 * r + 4 * l4 + q;  // C/D row of v_mfma_i32_16x16x64_i8
                    if (dimi < d) {
#pragma unroll
                        for (int t = 0; t < NCT; ++t) {
                            const int c = 16 * t + l15, a = c - KP;
                            if (c < KP) out[L.S + (int64_t)dimi * KP + c] = accM[r][t][q];
                            else if (a < K) out[L.U + (int64_t)dimi * K + a] = accM[r][t][q];
                            else if (a == K) out[L.totals + dimi] = accM[r][t][q];
                        }
                    }
                    if (dim < d) {
                        if (l15 < K) out[L.cross + (int64_t)dim * K + l15] = accX[r][q];
                        else if (l15 == K) out[L.sumx + dim] = accX[r][q];
                    }
                    continue;
                }
                if (dim >= d) continue;
#pragma unroll
                for (int t = 0; t < NTP; ++t) {
                    const int c = 16 * t + l15;
                    if (c < KP) out[L.S + (int64_t)dim * KP + c] = accM[r][t][q];
                    if constexpr (SPLIT) {  // [w z | w] columns 0 .. PADS-1 sit behind vech(P) in the last tile
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
            if constexpr (SPLIT && !P4I8) {  // 4x4x4 group: D lane = 16 i + 4 block + j
                const int dim = DW * wave + 16 * r + 4 * ((lane >> 2) & 3) + (lane >> 4);
                const int a = PADS + (lane & 3);  // index in [w z | w]
                if (dim < d) {
                    if (a < K) out[L.U + (int64_t)dim * K + a] = accS[r];
                    else if (a == K) out[L.totals + dim] = accS[r];
                }
            }
        }
    }
}

// ------------------------------------------------------------------ guard of the int8 mask-side statistics
// em8_kernel cuts the rows [wP | wz | w] into a fixed-point form with ONE exponent per column and workgroup that only rises
// (ppca_em8.hip).  A row far above its neighbours -- an outlier sample, or a heavy sample weight -- lifts the exponents of
// its workgroup, and the rows after it are cut far below their own resolution: harmless for every sum the large row is
// part of, but a dimension that is MASKED in the large row sums only the coarsely cut ones (measured: one row at 1e6 x
// the others, 6 000 rows on one workgroup: S_j off by 1e-3 .. 1e-2 in those dimensions).  Whether that matters is a
// property of the REDUCED statistic, so it is decided after the reduction: every workgroup reports, per column c, a bound
// e_c of the rounding it added to any sum of that column (4 sqrt(rows) quanta per flush window: ~14 sigma of independent
// roundings; rows cut to nothing show up as a small sum instead), and the pass is repeated on the fp64 engine when some
// diagonal entry S_j,aa of an observed dimension (a sum of non-negative terms z_a^2 + Sigma_aa) is not at least 2^34 x
// sum_workgroups e_c.  The off-diagonal and [wz | w] columns ride on that test: their rows scale with the same sample
// magnitudes (|P_ab| <= sqrt(P_aa P_bb), z_a^2 <= P_aa).
constexpr double WGUARD_TOL = 5.820766091346741e-11;  // 2^-34

// (round 5) The verdict used to send the WHOLE pass to the fp64 engine: one outlier row in ten million tripled the step.  The
// bound is a sum over workgroups and almost all of it comes from the few whose exponents the large rows lifted, so the
// second stage recomputes only those: with S_min,a = the smallest |S_j,aa| over the observed dimensions, workgroup g is
// flagged when e_aa[g] > S_min,a 2^-34 / (2 grid) for some a -- what is left un-flagged then sums to at most half the
// bound every diagonal entry has to clear.  The fp64 instantiation of the pass then walks only the flagged workgroups'
// slices (runs of tiles: ppca_em9.hip deals them as this kernel recomputes them), each split over grid / n_flagged of its
// workgroups, and the reduction is redone from the un-flagged partials + the fallback's (launch_em_fallback).  A model that
// tripped the Gram guard, or more than half the grid flagged: the whole pass again, as before.
//
// One launch does the reduction AND the verdict: workgroups [0, ceil(len / 64)) sum the statistics (out[e] = the fixed order of
// reduce_partials_kernel, bit-identical to it), two more sum the columns of the bounds; every workgroup takes a ticket after
// its sums have left (sc1 stores drained by an explicit s_waitcnt vmcnt(0) in the storing wave, then the workgroup barrier, then
// the ticket atomic; the last workgroup reads them back with sc1 loads), the holder of the last ticket runs the check.
template <int K>
__device__ __forceinline__ void reduce_wguard_body(const double *part, int64_t len, double *out, const GuardArgs &g) {
    constexpr int KP = Cfg<K>::KP, NTP = Cfg<K>::NTP;
    __shared__ double red[4][64];
    __shared__ int last_s;
    const int tid = threadIdx.x;
    const int nstat = (int)((len + 63) / 64);
    {
        const bool bounds = (int)blockIdx.x >= nstat;  // the two extra workgroups: es[c] = sum_g errb[g][c]
        const int gq = tid >> 6, l = tid & 63;
        const int64_t e = bounds ? (int64_t)(blockIdx.x - nstat) * 64 + l : (int64_t)blockIdx.x * 64 + l;
        const int64_t elen = bounds ? W_GUARD_NCOL : len;
        const double *src = bounds ? g.errb : part;
        const int per = (g.grid + 3) / 4;
        const int p0 = gq * per, p1 = (p0 + per < g.grid) ? p0 + per : g.grid;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        if (e < elen && src) {
            int q = p0;
            // sixteen partials requested before the first is added (a thread's four running sums take them in the order of the
            // four-at-a-time loop below: the result does not change): the launch is bound by the bytes it keeps in flight
            for (; q + 16 <= p1; q += 16) {
                double v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = src[(int64_t)(q + u) * elen + e];
#pragma unroll
                for (int u = 0; u < 16; u += 4) {
                    s0 += v[u];
                    s1 += v[u + 1];
                    s2 += v[u + 2];
                    s3 += v[u + 3];
                }
            }
            for (; q + 4 <= p1; q += 4) {
                s0 += src[(int64_t)q * elen + e];
                s1 += src[(int64_t)(q + 1) * elen + e];
                s2 += src[(int64_t)(q + 2) * elen + e];
                s3 += src[(int64_t)(q + 3) * elen + e];
            }
            for (; q < p1; ++q) s0 += src[(int64_t)q * elen + e];
        }
        red[gq][l] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (gq == 0) {
            // Handed to the last workgroup WITHOUT a release / acquire pair (311 agent-scope releases -- an L2 write-back each -- cost
            // ~10 us of this 20 us kernel): every store of the handed-off values is an sc1 store (written through to memory),
            // drained by the storing wave before the ticket, and every load of them below is an sc1 load (served past the L1 and
            // the reader's own L2) -- the form MI355X_MICROARCH.md lists as valid in place of the fences.
            if (e < elen) __hip_atomic_store((bounds ? g.es : out) + e, (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // The drain is written out: a workgroup-scope release fence emits NO s_waitcnt vmcnt on gfx950 (advisor, round 5: the built
            // code object had store -> s_barrier -> ticket atomic, nothing ordering the sc1 store's completion before the ticket).
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (tid == 0) last_s = atomicAdd(&g.qflag[QF_TICKET], 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (!last_s) return;
    auto ld = [](const double *q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };

    // ---------------------------------------------------------------- the verdict (one workgroup)
    __shared__ double es[W_GUARD_NCOL];
    __shared__ unsigned long long smin[K];
    __shared__ int nflag_s;
    __shared__ unsigned char fl[1024];
    int gram = 0;
#pragma unroll
    for (int u = 0; u < NTP; ++u) gram |= g.qflag[u];
    auto finish = [&](int mode, int verdict, int nflag, int nrows2) {
        if (tid == 0) {
            g.qflag[QF_MODE] = mode;
            g.qflag[QF_WVERDICT] = verdict;
            g.qflag[QF_NFLAGGED] = nflag;
            g.qflag[QF_NROWS2] = nrows2;
            g.qflag[QF_GVERDICT] = gram ? 1 : 0;
            g.qflag[QF_TICKET] = 0;  // (the next launch's election)
        }
    };
    auto flag_all = [&]() {
        for (int w = tid; w < g.grid; w += 256) g.wgflag[w] = 1;
    };
    if (gram) {  // the int8 kernel returned at once: its partials (and `out`) hold nothing
        flag_all();
        finish(1, 0, g.grid, 0);
        return;
    }
    if (!g.errb) {
        finish(0, 0, 0, 0);
        return;
    }
    if (tid < W_GUARD_NCOL) es[tid] = ld(g.es + tid);
    if (tid < K) smin[tid] = 0x7FF0000000000000ull;  // + inf
    if (tid == 0) nflag_s = 0;
    __syncthreads();
    int unsafe = 0;
    const StatsLayout L(g.d, K);
    for (int j = tid; j < g.d; j += 256) {
        if (ld(out + L.totals + j) != 0.0) {
#pragma unroll
            for (int a = 0; a < K; ++a) {
                const double S = fabs(ld(out + L.S + (int64_t)j * KP + tri(a, a)));
                if (S * WGUARD_TOL < es[tri(a, a)]) unsafe = 1;  // (a NaN statistic compares false: it propagates as through fp64)
            }
        }
    }
    unsafe = __syncthreads_or(unsafe);
    if (!unsafe) {
        finish(0, 0, 0, 0);
        return;
    }
    // which workgroups' cuts the bound is made of (the cold path from here on)
    for (int j = tid; j < g.d; j += 256) {
        if (ld(out + L.totals + j) != 0.0) {
#pragma unroll
            for (int a = 0; a < K; ++a) {
                const double S = fabs(ld(out + L.S + (int64_t)j * KP + tri(a, a)));
                if (S == S) atomicMin(&smin[a], (unsigned long long)__double_as_longlong(S));  // (non-negative doubles order like their bits)
            }
        }
    }
    __syncthreads();
    for (int w = tid; w < g.grid; w += 256) {
        int f = 0;
#pragma unroll
        for (int a = 0; a < K; ++a) {
            const double sm_a = __longlong_as_double((long long)smin[a]);
            if (g.errb[(int64_t)w * W_GUARD_NCOL + tri(a, a)] * (2.0 * (double)g.grid) > sm_a * WGUARD_TOL) f = 1;
        }
        g.wgflag[w] = f;
        if (w < 1024) fl[w] = (unsigned char)f;  // (the list below is built from LDS: no global round trip inside the workgroup)
        if (f) atomicAdd(&nflag_s, 1);
    }
    __syncthreads();
    const int nflag = nflag_s;
    if (nflag == 0 || 2 * nflag > g.grid || g.grid > 1024) {
        __syncthreads();
        flag_all();
        finish(1, 1, g.grid, 0);
        return;
    }
    // the flagged workgroups, ascending (one thread: at most grid / 2 entries), and the rows of their slices
    if (tid == 0) {
        const int64_t n = g.n_dev ? (int64_t)*g.n_dev : g.n;
        const int64_t ntiles = (n + FUSED_TILE - 1) / FUSED_TILE;
        const int64_t per_wg = (ntiles + g.grid - 1) / g.grid * FUSED_TILE;
        int at = 0;
        int64_t rows = 0;
        for (int w = 0; w < g.grid; ++w) {
            if (fl[w]) {
                g.who[at++] = w;
                const int64_t r0 = (int64_t)w * per_wg, r1 = r0 + per_wg < n ? r0 + per_wg : n;
                rows += r1 > r0 ? r1 - r0 : 0;
            }
        }
        nflag_s = (int)(rows < 0x7FFFFFFF ? rows : 0x7FFFFFFF);
    }
    __syncthreads();
    finish(2, 1, nflag, nflag_s);
}
template <int K>
__global__ __launch_bounds__(256) void reduce_wguard_kernel(const double *part, int64_t len, double *out, GuardArgs g) {
    reduce_wguard_body<K>(part, len, out, g);
}
// ... of the nm guarded component passes of a mixture step in one launch: grid (blocks, components); every component has its own
// partials, bounds, guard words (ticket counter included) and verdict
template <int K>
__global__ __launch_bounds__(256) void reduce_wguard_multi_kernel(MixReduceArgs m) {
    const int c = blockIdx.y;
    reduce_wguard_body<K>(m.part[c], m.len, m.out[c], m.g[c]);
}

// Second reduction of a guarded EM pass (behind the mode flag): the un-flagged workgroups' partials of the int8 kernel + the
// partials of the fp64 fallback, each in the fixed order of reduce_partials_kernel.
__global__ __launch_bounds__(256) void reduce_fallback_kernel(const double *part, const double *part2, const int *wgflag, int grid_parts,
                                                              int64_t len, double *out, const int *run_if) {
    __shared__ double red[4][64];
    if (*run_if == 0) return;
    const int gq = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int64_t e = (int64_t)blockIdx.x * 64 + l;
    const int per = (grid_parts + 3) / 4;
    const int p0 = gq * per, p1 = (p0 + per < grid_parts) ? p0 + per : grid_parts;
    double s0 = 0.0, s1 = 0.0;
    if (e < len) {
        for (int q = p0; q < p1; ++q) {
            if (!wgflag[q]) s0 += part[(int64_t)q * len + e];
            s1 += part2[(int64_t)q * len + e];
        }
    }
    red[gq][l] = s0 + s1;
    __syncthreads();
    if (gq == 0 && e < len) out[e] = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
}


// out[e] = sum over workgroup partials in a fixed order (deterministic): four threads per element
// each sum a contiguous quarter of the partials, the quarters are combined as (q0 + q1) + (q2 + q3).
__global__ __launch_bounds__(256) void reduce_partials_kernel(const double *part, int grid_parts, int64_t len,
                                                             double *out, int accumulate, const int *run_if) {
    __shared__ double red[4][64];
    if (run_if && *run_if == 0) return;
    const int g = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int64_t e = (int64_t)blockIdx.x * 64 + l;
    const int per = (grid_parts + 3) / 4;
    const int p0 = g * per, p1 = (p0 + per < grid_parts) ? p0 + per : grid_parts;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (e < len) {
        int q = p0;
        for (; q + 4 <= p1; q += 4) {  // four independent chains, recombined in a fixed order
            s0 += part[(int64_t)q * len + e];
            s1 += part[(int64_t)(q + 1) * len + e];
            s2 += part[(int64_t)(q + 2) * len + e];
            s3 += part[(int64_t)(q + 3) * len + e];
        }
        for (; q < p1; ++q) s0 += part[(int64_t)q * len + e];
    }
    red[g][l] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && e < len) {
        const double v = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
        out[e] = accumulate ? out[e] + v : v;
    }
}

// M-step finalisation (ppca_model.rs:307-322, :360-377) -- one workgroup.  write: this workgroup stores the new model; cn_lds /
// s2_lds (nullable): the new transform (row-major [d][K]) and sigma^2 also into LDS (finalize_qprep_kernel: every workgroup
// finalises redundantly, then builds its tile of the new model's slice table from there).
template <int K>
__device__ __forceinline__ void finalize_body(const double *stats, const double *min, double *mout, int d, double tau, int has_ig,
                                              double alpha, double beta, bool write, double *cn_lds, double *s2_lds) {
    constexpr int KP = K * (K + 1) / 2;
    StatsLayout L(d, K);
    __shared__ double red[256];
    const int tid = threadIdx.x;
    double ts = 0.0;
    for (int j = tid; j < d; j += 256) ts += stats[L.totals + j];
    red[tid] = ts;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    const double totsum = red[0];
    // deviations_square_sum comes from the identity |x~ - C z|^2 = |x~|^2 - b^T z - s2 |z|^2 (one pass): when the
    // residual is ~1e-16 of |x~|^2 (noise-free low-rank data) the difference can round below zero where the reference's
    // explicitly formed residual (ppca_model.rs:337-346) cannot; a negative total is clamped to 0 (sigma = 0), never NaN
    const double sq = stats[L.scalars + SC_SQERR], dv = fmax(stats[L.scalars + SC_DEVSQ], -stats[L.scalars + SC_SQERR]);
    const double s2new = has_ig ? ((sq + dv) / 2.0 + beta) / (totsum / 2.0 + alpha + 1.0) : (sq + dv) / totsum;
    const double *Cold = min + MODEL_HDR;
    const double *Mold = Cold + (int64_t)d * K;
    double *Cnew = mout + MODEL_HDR;
    double *Mnew = Cnew + (int64_t)d * K;
    for (int j = tid; j < d; j += 256) {
        double S[KP], rhs[K], cn[K];
#pragma unroll
        for (int e = 0; e < KP; ++e) S[e] = stats[L.S + (int64_t)j * KP + e];
        double cz = 0.0;
#pragma unroll
        for (int a = 0; a < K; ++a) {
            rhs[a] = stats[L.cross + (int64_t)j * K + a];
            cn[a] = Cold[(int64_t)j * K + a];  // keep the old row if the system is singular (:313-321)
            cz += cn[a] * stats[L.U + (int64_t)j * K + a];
        }
        row_solve<K>(S, tau, rhs, cn);
        if (cn_lds) {
#pragma unroll
            for (int a = 0; a < K; ++a) cn_lds[j * K + a] = cn[a];
        }
        if (write) {
#pragma unroll
            for (int a = 0; a < K; ++a) Cnew[(int64_t)j * K + a] = cn[a];
            const double tot = stats[L.totals + j];
            const double totdev = stats[L.sumx + j] - cz;  // sum_i w_i m_ij (x_ij - c_j.z_i - mu_j)  (:338-347)
            Mnew[j] = (tot > 0.0 ? totdev / tot : 0.0) + Mold[j];  // :373-377
        }
    }
    const double sig = sqrt(s2new);  // :389
    if (tid == 0) {
        if (write) {
            mout[0] = sig;
            mout[1] = sig * sig;
            mout[2] = log(sig);
            mout[3] = 0.0;
        }
        if (s2_lds) *s2_lds = sig * sig;  // (what the model buffer holds: the passes read sigma^2 from there)
    }
}
template <int K>
__global__ __launch_bounds__(256) void finalize_kernel(const double *stats, const double *min, double *mout, int d,
                                                       double tau, int has_ig, double alpha, double beta) {
    finalize_body<K>(stats, min, mout, d, tau, has_ig, alpha, beta, true, nullptr, nullptr);
}
// The plain EM step's finalisation AND the next pass's qprep_kernel in one launch: grid = the packed-column tiles of the slice
// table; every workgroup finalises (d row systems, one per thread: redundant and concurrent), workgroup 0 stores the model, then
// each builds its tile of the table, its guard flag and its share of the padded copy of C from the new transform in LDS.
template <int K>
__global__ __launch_bounds__(256) void finalize_qprep_kernel(const double *stats, const double *min, double *mout, int d, double tau,
                                                             int has_ig, double alpha, double beta, double *qscale, signed char *qtab,
                                                             int *qflag) {
    __shared__ double cn[FUSED_MAX_D * K];
    __shared__ double s2n;
    finalize_body<K>(stats, min, mout, d, tau, has_ig, alpha, beta, blockIdx.x == 0, cn, &s2n);
    __syncthreads();
    qprep_body<K>([&](int j, int a) { return cn[j * K + a]; }, s2n, d, qscale, qtab, qflag);
}
// ... of the nm components of a mixture step in one launch: grid (tiles, components)
template <int K>
__global__ __launch_bounds__(256) void finalize_qprep_multi_kernel(MixFinalArgs m) {
    __shared__ double cn[FUSED_MAX_D * K];
    __shared__ double s2n;
    const int c = blockIdx.y;
    finalize_body<K>(m.stats[c], m.min[c], m.mout[c], m.d, m.tau, m.has_ig, m.alpha, m.beta, blockIdx.x == 0, cn, &s2n);
    __syncthreads();
    PassArgs t{};
    fused_qtab_view(m.tab[c], t);
    qprep_body<K>([&](int j, int a) { return cn[j * K + a]; }, s2n, m.d, t.qscale, t.qtab, t.qflag);
}

// ------------------------------------------------------------------ synthetic data
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ uint64_t rng_bits(uint64_t seed, uint64_t stream, uint64_t row, uint64_t col) {
    return mix64(mix64(mix64(seed ^ (stream * 0xD1342543DE82EF95ull)) + row) + col * 0xA0761D6478BD642Full);
}
__device__ __forceinline__ double rng_u01(uint64_t bits) { return ((double)(bits >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
__device__ __forceinline__ double rng_normal(uint64_t seed, uint64_t stream, uint64_t row, uint64_t col) {
    double u1 = rng_u01(rng_bits(seed, stream, row, 2 * col));
    double u2 = rng_u01(rng_bits(seed, stream, row, 2 * col + 1));
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

__global__ void synth_latent_kernel(double *z, int64_t row_offset, int64_t n_rows, int k, uint64_t seed) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_rows * k) return;
    int64_t i = idx / k;
    int a = (int)(idx - i * k);
    z[idx] = rng_normal(seed, 1, (uint64_t)(row_offset + i), (uint64_t)a);
}

__global__ void synth_data_kernel(const double *c, const double *mean, const double *z, double *x, int64_t row_offset,
                                  int64_t n_rows, int d, int k, double sigma, double mask_prob, int mask_kind,
                                  int mask_run, uint64_t seed) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_rows * d) return;
    int64_t i = idx / d;
    int j = (int)(idx - i * d);
    const uint64_t grow = (uint64_t)(row_offset + i);
    double v = mean[j] + sigma * rng_normal(seed, 2, grow, (uint64_t)j);
    for (int a = 0; a < k; ++a) v += c[(int64_t)j * k + a] * z[i * k + a];
    bool masked;
    if (mask_kind == 0) {
        masked = rng_u01(rng_bits(seed, 3, grow, (uint64_t)j)) < mask_prob;
    } else {
        int start = (int)(rng_u01(rng_bits(seed, 4, grow, 0)) * d);
        int off = j - start;
        if (off < 0) off += d;
        masked = off < mask_run;
    }
    x[idx] = masked ? __builtin_nan("") : v;
}

// present[j] = 1 if any sample has a finite value in dim j (Dataset::empty_dimensions, dataset.rs:194-222)
__global__ void column_presence_kernel(const double *X, int64_t ldx, int64_t n, int d, int *present) {
    int j = blockIdx.y * blockDim.x + threadIdx.x;
    if (j >= d) return;
    int64_t rows_per = (n + gridDim.x - 1) / gridDim.x;
    int64_t r0 = (int64_t)blockIdx.x * rows_per, r1 = r0 + rows_per < n ? r0 + rows_per : n;
    int any = 0;
    for (int64_t r = r0; r < r1; ++r) any |= __builtin_isfinite(X[r * ldx + j]) ? 1 : 0;
    if (any) atomicOr(&present[j], 1);
}

// dst[i] = src[i] if finite, NaN otherwise: masked entries leave the device in the canonical form of masked_vector
// (dataset.rs:64-72; +-inf inputs are masked, so they come back NaN too)
__global__ void canon_copy_kernel(const double *src, double *dst, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i + 1 < n) {
        typedef double d2_t __attribute__((ext_vector_type(2)));
        d2_t v = *reinterpret_cast<const d2_t *>(src + i);
        v[0] = __builtin_isfinite(v[0]) ? v[0] : __builtin_nan("");
        v[1] = __builtin_isfinite(v[1]) ? v[1] : __builtin_nan("");
        *reinterpret_cast<d2_t *>(dst + i) = v;
    } else if (i < n) {
        dst[i] = __builtin_isfinite(src[i]) ? src[i] : __builtin_nan("");
    }
}
// X[rows[i]][:] *= factor (bench / test hook: outlier rows in a device-resident dataset; non-finite entries stay masked)
__global__ void scale_rows_kernel(double *X, int64_t ldx, int d, const int64_t *rows, int64_t n_rows, double factor) {
    const int64_t i = blockIdx.x;
    if (i >= n_rows) return;
    for (int j = threadIdx.x; j < d; j += blockDim.x) X[rows[i] * ldx + j] *= factor;
}
hipError_t launch_scale_rows(double *X, int64_t ldx, int d, const int64_t *rows_dev, int64_t n_rows, double factor, hipStream_t s) {
    if (n_rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)n_rows), dim3(256), 0, s, X, ldx, d, rows_dev, n_rows, factor);
    return hipGetLastError();
}
hipError_t launch_canon_copy(const double *src, double *dst, int64_t n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const int64_t pairs = (n + 1) / 2;
    hipLaunchKernelGGL(canon_copy_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, s, src, dst, n);
    return hipGetLastError();
}

__global__ void fill_kernel(double *p, int64_t n, double v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ void mfma_probe_kernel(const double *a, const double *b, double *out) {
    const int lane = threadIdx.x;
    d4_t acc = {0, 0, 0, 0};
    // A[i][kk] at lane = i + 16 kk ; B[kk][j] at lane = j + 16 kk
    acc = mfma(a[(lane & 15) * 4 + (lane >> 4)], b[(lane >> 4) * 16 + (lane & 15)], acc);
    for (int r = 0; r < 4; ++r) out[((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc[r];
}

// ------------------------------------------------------------------ mixture helpers
// mix.rs:283-295 (log-softmax of llk_c + log pi_c) and :304-309 (ln w_i + log posterior)
__global__ void mix_posteriors_kernel(const double *llk, const double *logw, const double *w, int64_t n, int nm,
                                      double *u, double *lse, double *logpost) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double mx = -INFINITY;
    for (int c = 0; c < nm; ++c) mx = fmax(mx, llk[(int64_t)c * n + i] + logw[c]);
    double s = 0.0;
    for (int c = 0; c < nm; ++c) s += exp(llk[(int64_t)c * n + i] + logw[c] - mx);
    const double ln = log(s);
    if (lse) lse[i] = mx + ln;
    const double wi = w ? w[i] : 1.0;
    const double lw = wi > 0.0 ? log(wi) : -INFINITY;
    for (int c = 0; c < nm; ++c) {
        double lp = llk[(int64_t)c * n + i] + logw[c] - mx - ln;
        if (logpost) logpost[i * nm + c] = lp;
        if (u) u[(int64_t)c * n + i] = lw + lp;
    }
}

// The same + what the mixture step needs of u and lse afterwards, so that no further sweep over them is launched: per block of 256
// samples the maxima of u_c (NaNs skipped, mix.rs:312-315) and the sum of w_i lse_i, into bpart[nm + 1][gridDim.x] (fixed order: lane
// butterflies, then the four waves in index order).  nm <= MIX_MAX.
__global__ __launch_bounds__(256) void mix_posteriors2_kernel(const double *llk, const double *logw, const double *w, int64_t n, int nm,
                                                               double *u, double *lse, double *bpart) {
    __shared__ double red[4][MIX_MAX + 1];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool in = i < n;
    double lv[MIX_MAX], mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < MIX_MAX; ++c) {
        lv[c] = (in && c < nm) ? llk[(int64_t)c * n + i] + logw[c] : -INFINITY;
        if (c < nm) mx = fmax(mx, lv[c]);
    }
    double sm = 0.0;
#pragma unroll
    for (int c = 0; c < MIX_MAX; ++c)
        if (c < nm) sm += exp(lv[c] - mx);
    const double ln = log(sm);
    const double wi = in ? (w ? w[i] : 1.0) : 0.0;
    const double lw = wi > 0.0 ? log(wi) : -INFINITY;
    if (in) lse[i] = mx + ln;
    double part = in ? (w ? (mx + ln) * wi : mx + ln) : 0.0;  // (launch_reduce_sum's term: v w or v)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    if (lane == 0) red[wave][MIX_MAX] = part;
#pragma unroll
    for (int c = 0; c < MIX_MAX; ++c) {
        if (c < nm) {
            double uc = -INFINITY;
            if (in) {
                const double v = lw + (lv[c] - mx - ln);
                u[(int64_t)c * n + i] = v;
                uc = (v == v) ? v : -INFINITY;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) uc = fmax(uc, __shfl_xor(uc, o, 64));
            if (lane == 0) red[wave][c] = uc;
        }
    }
    __syncthreads();
    if (threadIdx.x < nm) {
        const int c = threadIdx.x;
        bpart[(int64_t)c * gridDim.x + blockIdx.x] = fmax(fmax(red[0][c], red[1][c]), fmax(red[2][c], red[3][c]));
    } else if (threadIdx.x == MIX_MAX) {
        bpart[(int64_t)nm * gridDim.x + blockIdx.x] = (red[0][MIX_MAX] + red[1][MIX_MAX]) + (red[2][MIX_MAX] + red[3][MIX_MAX]);
    }
}
// Second stage: workgroup c < nm: maxima[c] = max of bpart[c][0 .. nb); workgroup nm: *llk_out = sum of bpart[nm][0 .. nb) -- thread t
// takes the blocks t, t + 256, ... in order, then a fixed tree.
__global__ __launch_bounds__(256) void mix_stage2_kernel(const double *bpart, int nb, int nm, double *maxima, double *llk_out) {
    __shared__ double red[256];
    const int c = blockIdx.x, t = threadIdx.x;
    const bool is_max = c < nm;
    double acc = is_max ? -INFINITY : 0.0;
    for (int b = t; b < nb; b += 256) {
        const double v = bpart[(int64_t)c * nb + b];
        acc = is_max ? fmax(acc, v) : acc + v;
    }
    red[t] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) red[t] = is_max ? fmax(red[t], red[t + o]) : red[t] + red[t + o];
        __syncthreads();
    }
    if (t == 0) {
        if (is_max) maxima[c] = red[0];
        else *llk_out = red[0];
    }
}

template <bool MAX>
__global__ void reduce_stage_kernel(const double *v, const double *w, int64_t n, double *out, const int *n_dev = nullptr) {
    __shared__ double red[256];
    const int tid = threadIdx.x;
    if (n_dev) n = *n_dev;  // (the count was produced on the device; the grid is sized for an upper bound)
    double acc = MAX ? -INFINITY : 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < n; i += (int64_t)gridDim.x * 256) {
        double x = v[i];
        if (MAX) {
            if (x == x) acc = fmax(acc, x);  // NaNs are skipped (mix.rs:312-315)
        } else {
            acc += w ? x * w[i] : x;
        }
    }
    red[tid] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] = MAX ? fmax(red[tid], red[tid + o]) : red[tid] + red[tid + o];
        __syncthreads();
    }
    if (tid == 0) out[blockIdx.x] = red[0];
}

__global__ void exp_shift_kernel(const double *v, const double *mx, int64_t n, double *out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = exp(v[i] - *mx);
}

// ------------------------------------------------------------------ responsibility-sparse component passes
// Per mixture component the sample weights are exp(u_i - max) (mix.rs:320-323): the largest is exactly 1.  Every
// statistic is a sum of weight x per-sample term, so a sample whose weight is below 2^-200 (6e-61) of the largest
// moves no statistic: its term is >= 147 binary orders below the fp64 resolution of the sum it would join (the
// exponential underflows to exactly zero only past 2^-1074; at d = 256 most samples of the other clusters sit far
// below either bound).  Such samples are dropped from the component's pass.
// select_*: ascending list of the rows with a non-zero weight and their weights, in three deterministic steps
// (per-block counts, exclusive scan of the counts, ordered scatter); the EM pass then gathers those rows.
constexpr int SEL_BLOCK = 256;
constexpr double SEL_MIN_WEIGHT = 6.223015277861142e-61;  // 2^-200
__global__ __launch_bounds__(SEL_BLOCK) void select_count_kernel(const double *v, const double *shift, int64_t n, int *counts) {
    const int64_t i = (int64_t)blockIdx.x * SEL_BLOCK + threadIdx.x;
    const bool keep = i < n && exp(v[i] - *shift) > SEL_MIN_WEIGHT;
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(keep);
    __shared__ int wc[SEL_BLOCK / 64];
    if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
}
// counts[0 .. nblocks) -> exclusive offsets in place; counts[nblocks] = total.  One workgroup.
__global__ __launch_bounds__(1024) void select_scan_kernel(int *counts, int nblocks) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (nblocks + 1023) / 1024;
    const int b0 = t * per, b1 = (b0 + per < nblocks) ? b0 + per : nblocks;
    int s = 0;
    for (int b = b0; b < b1; ++b) s += counts[b];
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {  // inclusive scan of the per-thread sums
        const int add = (t >= o) ? part[t - o] : 0;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    int run = part[t] - s;
    for (int b = b0; b < b1; ++b) {
        const int c = counts[b];
        counts[b] = run;
        run += c;
    }
    if (t == 1023) counts[nblocks] = part[1023];
}
__global__ __launch_bounds__(SEL_BLOCK) void select_scatter_kernel(const double *v, const double *shift, int64_t n,
                                                                    const int *offsets, int *rows, double *wout) {
    const int64_t i = (int64_t)blockIdx.x * SEL_BLOCK + threadIdx.x;
    const double w = i < n ? exp(v[i] - *shift) : 0.0;
    const bool keep = w > SEL_MIN_WEIGHT;
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(keep);
    __shared__ int wc[SEL_BLOCK / 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) wc[wave] = __popcll(bal);
    __syncthreads();
    int base = offsets[blockIdx.x];
    for (int u = 0; u < wave; ++u) base += wc[u];
    if (keep) {
        const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
        rows[pos] = (int)i;
        wout[pos] = w;
    }
}
hipError_t launch_select_positive(const double *v, const double *shift_dev, int64_t n, int *counts, int *rows, double *wout,
                                  hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const int nblocks = (int)((n + SEL_BLOCK - 1) / SEL_BLOCK);
    hipLaunchKernelGGL(select_count_kernel, dim3(nblocks), dim3(SEL_BLOCK), 0, s, v, shift_dev, n, counts);
    hipLaunchKernelGGL(select_scan_kernel, dim3(1), dim3(1024), 0, s, counts, nblocks);
    hipLaunchKernelGGL(select_scatter_kernel, dim3(nblocks), dim3(SEL_BLOCK), 0, s, v, shift_dev, n, counts, rows, wout);
    return hipGetLastError();
}
int select_blocks(int64_t n) { return (int)((n + SEL_BLOCK - 1) / SEL_BLOCK); }

// The three selection steps for nm components at once (blockIdx.y = component; component c's arrays at c * n, its counts at
// c * (nb + 1)), the scatter also leaving the block's sum of kept weights (fixed order), and a last step that sums those per component.
__global__ __launch_bounds__(SEL_BLOCK) void select_count_multi_kernel(const double *u, const double *shift, int64_t n, int nb, int *counts) {
    const int c = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * SEL_BLOCK + threadIdx.x;
    const bool keep = i < n && exp(u[(int64_t)c * n + i] - shift[c]) > SEL_MIN_WEIGHT;
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(keep);
    __shared__ int wc[SEL_BLOCK / 64];
    if ((threadIdx.x & 63) == 0) wc[threadIdx.x >> 6] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) counts[(int64_t)c * (nb + 1) + blockIdx.x] = wc[0] + wc[1] + wc[2] + wc[3];
}
__global__ __launch_bounds__(1024) void select_scan_multi_kernel(int *counts_all, int nblocks) {
    int *counts = counts_all + (int64_t)blockIdx.x * (nblocks + 1);
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (nblocks + 1023) / 1024;
    const int b0 = t * per, b1 = (b0 + per < nblocks) ? b0 + per : nblocks;
    int s = 0;
    for (int b = b0; b < b1; ++b) s += counts[b];
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int add = (t >= o) ? part[t - o] : 0;
        __syncthreads();
        part[t] += add;
        __syncthreads();
    }
    int run = part[t] - s;
    for (int b = b0; b < b1; ++b) {
        const int c = counts[b];
        counts[b] = run;
        run += c;
    }
    if (t == 1023) counts[nblocks] = part[1023];
}
__global__ __launch_bounds__(SEL_BLOCK) void select_scatter_multi_kernel(const double *u, const double *shift, int64_t n, int nb, const int *counts,
                                                                          int *rows, double *wout, double *wpart) {
    const int c = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * SEL_BLOCK + threadIdx.x;
    const double w = i < n ? exp(u[(int64_t)c * n + i] - shift[c]) : 0.0;
    const bool keep = w > SEL_MIN_WEIGHT;
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(keep);
    __shared__ int wc[SEL_BLOCK / 64];
    __shared__ double ws[SEL_BLOCK / 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double sw = keep ? w : 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sw += __shfl_xor(sw, o, 64);
    if (lane == 0) {
        wc[wave] = __popcll(bal);
        ws[wave] = sw;
    }
    __syncthreads();
    int base = counts[(int64_t)c * (nb + 1) + blockIdx.x];
    for (int q = 0; q < wave; ++q) base += wc[q];
    if (keep) {
        const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
        rows[(int64_t)c * n + pos] = (int)i;
        wout[(int64_t)c * n + pos] = w;
    }
    if (threadIdx.x == 0) wpart[(int64_t)c * nb + blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}
__global__ __launch_bounds__(256) void mix_wsum_kernel(const double *wpart, int nb, const int *counts, double *sums_out, int *used_out) {
    __shared__ double red[256];
    const int c = blockIdx.x, t = threadIdx.x;
    double acc = 0.0;
    for (int b = t; b < nb; b += 256) acc += wpart[(int64_t)c * nb + b];
    red[t] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) red[t] += red[t + o];
        __syncthreads();
    }
    if (t == 0) {
        sums_out[c] = red[0];
        used_out[c] = counts[(int64_t)c * (nb + 1) + nb];
    }
}
hipError_t launch_select_multi(const double *u, const double *shift_dev, int64_t n, int nm, int *counts, int *rows, double *wout, double *wpart,
                               double *sums_out, int *used_out, hipStream_t s) {
    if (n <= 0 || nm <= 0) return hipSuccess;
    const int nb = (int)((n + SEL_BLOCK - 1) / SEL_BLOCK);
    hipLaunchKernelGGL(select_count_multi_kernel, dim3(nb, nm), dim3(SEL_BLOCK), 0, s, u, shift_dev, n, nb, counts);
    hipLaunchKernelGGL(select_scan_multi_kernel, dim3(nm), dim3(1024), 0, s, counts, nb);
    hipLaunchKernelGGL(select_scatter_multi_kernel, dim3(nb, nm), dim3(SEL_BLOCK), 0, s, u, shift_dev, n, nb, (const int *)counts, rows, wout, wpart);
    hipLaunchKernelGGL(mix_wsum_kernel, dim3(nm), dim3(256), 0, s, (const double *)wpart, nb, (const int *)counts, sums_out, used_out);
    return hipGetLastError();
}
hipError_t launch_mix_posteriors2(const double *llk, const double *logw_dev, const double *w, int64_t n, int nm, double *u, double *lse,
                                  double *bpart, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (nm > MIX_MAX) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mix_posteriors2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, llk, logw_dev, w, n, nm, u, lse, bpart);
    return hipGetLastError();
}
hipError_t launch_mix_stage2(const double *bpart, int64_t n, int nm, double *maxima, double *llk_out, hipStream_t s) {
    const int nb = (int)((n + 255) / 256);
    hipLaunchKernelGGL(mix_stage2_kernel, dim3(nm + 1), dim3(256), 0, s, bpart, nb, nm, maxima, llk_out);
    return hipGetLastError();
}

// ------------------------------------------------------------------ launchers
int fused_grid(int64_t n, int n_cu) {
    int64_t tiles = (n + FUSED_TILE - 1) / FUSED_TILE;
    if (tiles < 1) tiles = 1;
    return (int)(tiles < n_cu ? tiles : n_cu);
}

size_t fused_lds_bytes(int k) {
    switch (k) {
#define PPCA_CASE(KK) \
    case KK:          \
        return sizeof(double) * Cfg<KK>::LDS_DOUBLES;
        PPCA_CASE(1) PPCA_CASE(2) PPCA_CASE(3) PPCA_CASE(4) PPCA_CASE(5) PPCA_CASE(6) PPCA_CASE(7) PPCA_CASE(8)
        PPCA_CASE(9) PPCA_CASE(10)
#undef PPCA_CASE
    }
    return 0;
}

// hipFuncSetAttribute is per device: remember which devices have seen it for this instantiation.
template <int K, bool EM, int NW, bool GI8, bool GATHER = false>
static hipError_t pass_attr(size_t lds) {
    static std::atomic<unsigned long long> done{0ull};
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&pass_kernel<K, EM, NW, GI8, GATHER>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    done.fetch_or(bit, std::memory_order_release);
    return hipSuccess;
}

template <int K, bool EM, int NW, bool GI8, bool GATHER = false>
static hipError_t launch_pass_t(int grid, const PassArgs &a, hipStream_t s) {
    const size_t lds = sizeof(double) * Cfg<K>::LDS_DOUBLES;
    if (hipError_t e = pass_attr<K, EM, NW, GI8, GATHER>(lds); e != hipSuccess) return e;
    hipLaunchKernelGGL((pass_kernel<K, EM, NW, GI8, GATHER>), dim3(grid), dim3(64 * NW), lds, s, a);
    return hipGetLastError();
}

size_t fused_qtab_bytes() { return qtab_bytes<FUSED_MAX_K>() + (72 + FUSED_MAX_D * (FUSED_MAX_K + 1) + CPB_DOUBLES) * sizeof(double); }
void fused_qtab_layout(void *base, PassArgs &a) { fused_qtab_view(base, a); }  // [64 scales | 8 doubles of guard flags | digit table | ...]

// Gram engine of the fused passes: 0 = int8-sliced MFMA behind the dynamic-range guard, with the fp64-MFMA instantiation
// as its on-device fallback (default); 1 = fp64 MFMA always (PPCA_GRAM_FP64=1: a safe choice, so a run-time switch);
// 2 = int8 WITHOUT the guard -- only in builds compiled with -DPPCA_NO_GRAM_GUARD (measurements; never a run-time
// switch: it would silently disable a parity guard).
static int gram_mode() {
    static const int v = [] {
        const char *e = getenv("PPCA_GRAM_FP64");
        if (e && atoi(e) == 1) return 1;
#ifdef PPCA_NO_GRAM_GUARD
        return 2;
#else
        return 0;
#endif
    }();
    return v;
}

// PPCA_EM8=0: the four-wave pass_kernel for the EM pass (A/B runs against the eight-wave role-split kernel)
static bool em8_enabled() {
    static const bool v = [] {
        const char *e = getenv("PPCA_EM8");
        return !(e && atoi(e) == 0);
    }();
    return v;
}

// The eight-wave kernel with the per-sample solve pipelined across tiles (ppca_em9.hip; default since the end of round 4:
// 98.0 against 96.1 EM it/s, every GPU test green on both).  PPCA_EM9=0: ppca_em8.hip's kernel.
static bool em9_enabled() {
    static const bool v = [] {
        const char *e = getenv("PPCA_EM9");
        return !(e && atoi(e) == 0);
    }();
    return v;
}

static bool llk2_enabled() {
    static const bool v = [] {
        const char *e = getenv("PPCA_LLK2");
        return !(e && atoi(e) == 0);
    }();
    return v;
}

// Slice table + guard flags of the current model (device-side, no host sync), then the pass: the int8 variant and,
// behind the guard, the fp64 variant -- each returns at once unless qflag selects it.
template <int K, bool EM>
static hipError_t launch_pass_guarded(int grid, PassArgs a, hipStream_t s) {
#ifdef PPCA_DEV_K10
    const int mode = 2;  // kernel-tuning builds instantiate the int8 variants only
#else
    const int mode = gram_mode();
    if (mode == 1) {
        a.qflag = nullptr;
        return launch_pass_t<K, EM, 4, false>(grid, a, s);
    }
#endif
    if (!a.skip_qprep) hipLaunchKernelGGL((qprep_kernel<K>), dim3(Cfg<K>::NTP), dim3(256), 0, s, a.model, a.d, a.qscale, a.qtab, a.qflag);
    auto int8_pass = [&](const PassArgs &b) {
        if constexpr (!EM) {  // llk / llks alone: the two-tile sweep (ppca_llk.hip) unless PPCA_LLK2=0
            if (!b.states && !b.covs && !b.recon && llk2_enabled()) return launch_llk2(K, grid, b, s);
            if (recon8_covers(b)) return launch_recon8(K, grid, b, s);  // smooth / extrapolate on the eight-wave sweep (round 6)
        }
        if constexpr (EM) {
            if (em8_enabled() && em9_enabled() && em9_covers(K)) return launch_em9(K, grid, b, s);  // ... with the solve pipelined across tiles (PPCA_EM8=0 switches both eight-wave kernels off)
            if (em8_enabled() && em8_covers(K)) return launch_em8(K, grid, b, s);  // eight waves, two roles (ppca_em8.hip)
            if (b.rows) return launch_pass_t<K, EM, 4, true, true>(grid, b, s);
        }
        return launch_pass_t<K, EM, 4, true>(grid, b, s);
    };
    if (mode == 2) {
        a.qflag = nullptr;
        return int8_pass(a);
    }
#ifdef PPCA_DEV_K10
    return hipErrorInvalidValue;
#else
    if (hipError_t e = int8_pass(a); e != hipSuccess) return e;
    if constexpr (EM) return hipSuccess;  // the fp64 variant of an EM pass runs behind launch_em_wguard, after the reduction
    return launch_pass_t<K, EM, 4, false>(grid, a, s);
#endif
}

#ifdef PPCA_DEV_K10
// kernel-tuning builds (tools/devbuild.py): only the k = 10 int8-Gram variants are instantiated
#define PPCA_DISPATCH_K(k, EXPR)                         \
    switch (k) {                                         \
        case 10: { constexpr int KK = 10; EXPR; } break; \
        default: return hipErrorInvalidValue;            \
    }
#else
#define PPCA_DISPATCH_K(k, EXPR)                         \
    switch (k) {                                         \
        case 1: { constexpr int KK = 1; EXPR; } break;   \
        case 2: { constexpr int KK = 2; EXPR; } break;   \
        case 3: { constexpr int KK = 3; EXPR; } break;   \
        case 4: { constexpr int KK = 4; EXPR; } break;   \
        case 5: { constexpr int KK = 5; EXPR; } break;   \
        case 6: { constexpr int KK = 6; EXPR; } break;   \
        case 7: { constexpr int KK = 7; EXPR; } break;   \
        case 8: { constexpr int KK = 8; EXPR; } break;   \
        case 9: { constexpr int KK = 9; EXPR; } break;   \
        case 10: { constexpr int KK = 10; EXPR; } break; \
        default: return hipErrorInvalidValue;            \
    }
#endif

hipError_t launch_pass_em(int k, int grid, const PassArgs &a, hipStream_t s) {
    // (the eight-wave fp64-Gram instantiation of pass_kernel that PPCA_FUSED_WAVES=8 used to select is gone: it ignored
    //  PassArgs::rows, and the eight-wave kernel of this library is ppca_em8.hip)
    PPCA_DISPATCH_K(k, return (launch_pass_guarded<KK, true>(grid, a, s)));
    return hipErrorInvalidValue;
}
hipError_t launch_reduce_wguard(int k, const double *part, int64_t len, double *stats, const GuardArgs &g, hipStream_t s, bool *applies_out) {
    *applies_out = false;
#ifdef PPCA_DEV_K10
    return launch_reduce_partials(part, g.grid, len, stats, s);  // kernel-tuning builds instantiate the int8 variants only
#else
    if (gram_mode() != 0) return launch_reduce_partials(part, g.grid, len, stats, s);  // engine pinned: nothing to decide
    *applies_out = true;
    GuardArgs h = g;
    if (!(em8_enabled() && em8_covers(k))) h.errb = nullptr;  // (only em8_kernel / em9_kernel cut their rows)
    const int blocks = (int)((len + 63) / 64) + (W_GUARD_NCOL + 63) / 64;
    PPCA_DISPATCH_K(k, {
        hipLaunchKernelGGL((reduce_wguard_kernel<KK>), dim3(blocks), dim3(256), 0, s, part, len, stats, h);
        return hipGetLastError();
    });
    return hipErrorInvalidValue;
#endif
}
hipError_t launch_em_fallback(int k, int grid, PassArgs a, const GuardArgs &g, const double *part, double *part2, int64_t len, double *stats,
                              hipStream_t s) {
#ifdef PPCA_DEV_K10
    return hipSuccess;
#else
    a.runflag = g.qflag + QF_MODE;
    a.who = g.who;
    a.part = part2;
    a.errb = nullptr;
    PPCA_DISPATCH_K(k, {
        if (hipError_t e = launch_pass_t<KK, true, 4, false>(grid, a, s); e != hipSuccess) return e;
    });
    const int blocks = (int)((len + 63) / 64);
    hipLaunchKernelGGL(reduce_fallback_kernel, dim3(blocks), dim3(256), 0, s, part, (const double *)part2, (const int *)g.wgflag, grid, len, stats,
                       (const int *)(g.qflag + QF_MODE));
    return hipGetLastError();
#endif
}

int fused_gram_mode() {
#ifdef PPCA_DEV_K10
    return 2;
#else
    return gram_mode();
#endif
}
hipError_t launch_pass_post_fp64(int k, int grid, const PassArgs &a, hipStream_t s) {
#ifdef PPCA_DEV_K10
    return hipSuccess;
#else
    PPCA_DISPATCH_K(k, return (launch_pass_t<KK, false, 4, false>(grid, a, s)));
    return hipErrorInvalidValue;
#endif
}
hipError_t launch_qprep_multi(int k, const MixTabArgs &a, hipStream_t s) {
    PPCA_DISPATCH_K(k, hipLaunchKernelGGL((qprep_multi_kernel<KK>), dim3(Cfg<KK>::NTP, a.nm), dim3(256), 0, s, a));
    return hipGetLastError();
}
hipError_t launch_finalize_qprep_multi(int k, const MixFinalArgs &a, hipStream_t s) {
    PPCA_DISPATCH_K(k, hipLaunchKernelGGL((finalize_qprep_multi_kernel<KK>), dim3(Cfg<KK>::NTP, a.nm), dim3(256), 0, s, a));
    return hipGetLastError();
}
hipError_t launch_reduce_wguard_multi(int k, const MixReduceArgs &a, int grid_parts, hipStream_t s) {
#ifdef PPCA_DEV_K10
    return hipErrorInvalidValue;
#else
    (void)grid_parts;
    MixReduceArgs h = a;
    if (!(em8_enabled() && em8_covers(k)))
        for (int c = 0; c < h.nm; ++c) h.g[c].errb = nullptr;  // (only em8_kernel / em9_kernel cut their rows)
    const int blocks = (int)((a.len + 63) / 64) + (W_GUARD_NCOL + 63) / 64;
    PPCA_DISPATCH_K(k, hipLaunchKernelGGL((reduce_wguard_multi_kernel<KK>), dim3(blocks, a.nm), dim3(256), 0, s, h));
    return hipGetLastError();
#endif
}

hipError_t launch_pass_post(int k, int grid, const PassArgs &a, hipStream_t s) {
    PPCA_DISPATCH_K(k, return (launch_pass_guarded<KK, false>(grid, a, s)));
    return hipErrorInvalidValue;
}

hipError_t launch_gram_guard(int k, const PassArgs &a, hipStream_t s, int *forced) {
#ifdef PPCA_DEV_K10
    *forced = 0;
    return hipSuccess;
#else
    const int mode = gram_mode();
    *forced = mode == 1 ? 1 : (mode == 2 ? 0 : -1);
    if (mode != 0) return hipSuccess;
    PPCA_DISPATCH_K(k, hipLaunchKernelGGL((qprep_kernel<KK>), dim3(Cfg<KK>::NTP), dim3(256), 0, s, a.model, a.d, a.qscale,
                                          a.qtab, a.qflag));
    return hipGetLastError();
#endif
}
// slice table + guard flags for the state sizes of ppca_em16.hip
hipError_t launch_qprep16(int k, const double *model, int d, double *qscale, signed char *qtab, int *qflag, hipStream_t s) {
    switch (k) {
#define PPCA_Q16(KK)                                                                                                  \
    case KK:                                                                                                          \
        hipLaunchKernelGGL((qprep_kernel<KK>), dim3(Cfg<KK>::NTP), dim3(256), 0, s, model, d, qscale, qtab, qflag); \
        break;
        PPCA_Q16(11) PPCA_Q16(12) PPCA_Q16(13) PPCA_Q16(14) PPCA_Q16(15) PPCA_Q16(16)
#undef PPCA_Q16
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
int fused_gram_tiles(int k) { return (k * (k + 1) / 2 + 15) / 16; }

hipError_t launch_reduce_partials(const double *part, int grid_parts, int64_t len, double *out, hipStream_t s, int accumulate,
                                  const int *run_if) {
    int blocks = (int)((len + 63) / 64);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(blocks), dim3(256), 0, s, part, grid_parts, len, out, accumulate, run_if);
    return hipGetLastError();
}

hipError_t launch_finalize(int k, int d, const double *stats, const double *model_in, double *model_out, double tau,
                           int has_ig, double alpha, double beta, hipStream_t s) {
    PPCA_DISPATCH_K(k, hipLaunchKernelGGL((finalize_kernel<KK>), dim3(1), dim3(256), 0, s, stats, model_in, model_out,
                                          d, tau, has_ig, alpha, beta));
    return hipGetLastError();
}

hipError_t launch_finalize_qprep(int k, int d, const double *stats, const double *model_in, double *model_out, double tau, int has_ig,
                                 double alpha, double beta, const PassArgs &tab, hipStream_t s) {
    PPCA_DISPATCH_K(k, hipLaunchKernelGGL((finalize_qprep_kernel<KK>), dim3(Cfg<KK>::NTP), dim3(256), 0, s, stats, model_in, model_out, d,
                                          tau, has_ig, alpha, beta, tab.qscale, tab.qtab, tab.qflag));
    return hipGetLastError();
}

hipError_t launch_synth(const double *c_dev, const double *mean_dev, double *z_work, double *x_out, int64_t row_offset,
                        int64_t n_rows, int d, int k, double sigma, double mask_prob, int mask_kind, int mask_run,
                        uint64_t seed, hipStream_t s) {
    if (n_rows <= 0) return hipSuccess;
    int64_t nz = n_rows * k;
    if (nz > 0)
        hipLaunchKernelGGL(synth_latent_kernel, dim3((unsigned)((nz + 255) / 256)), dim3(256), 0, s, z_work, row_offset,
                           n_rows, k, seed);
    int64_t nx = n_rows * d;
    hipLaunchKernelGGL(synth_data_kernel, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, s, c_dev, mean_dev, z_work,
                       x_out, row_offset, n_rows, d, k, sigma, mask_prob, mask_kind, mask_run, seed);
    return hipGetLastError();
}

hipError_t launch_column_presence(const double *X, int64_t ldx, int64_t n, int d, int *present, hipStream_t s) {
    if (n <= 0 || d <= 0) return hipSuccess;
    int gx = (int)((n + 4095) / 4096);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(column_presence_kernel, dim3(gx, (d + 255) / 256), dim3(256), 0, s, X, ldx, n, d, present);
    return hipGetLastError();
}

hipError_t launch_fill(double *p, int64_t n, double v, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n, v);
    return hipGetLastError();
}

// raw operand/result registers of one v_mfma_i32_16x16x64_i8: a, b: [64 lanes][16 bytes]; out: [64][4]
__global__ void mfma_i8_probe_kernel(const int *a, const int *b, int *out) {
    const int lane = threadIdx.x;
    i4_t av, bv, acc = {0, 0, 0, 0};
    for (int u = 0; u < 4; ++u) {
        av[u] = a[lane * 4 + u];
        bv[u] = b[lane * 4 + u];
    }
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bv, acc, 0, 0, 0);
    for (int u = 0; u < 4; ++u) out[lane * 4 + u] = acc[u];
}
hipError_t launch_mfma_i8_probe(const int *a, const int *b, int *out, hipStream_t s) {
    hipLaunchKernelGGL(mfma_i8_probe_kernel, dim3(1), dim3(64), 0, s, a, b, out);
    return hipGetLastError();
}

hipError_t launch_mfma_probe(const double *a16x4, const double *b4x16, double *out16x16, hipStream_t s) {
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(1), dim3(64), 0, s, a16x4, b4x16, out16x16);
    return hipGetLastError();
}

hipError_t launch_mix_posteriors(const double *llk, const double *logw_dev, const double *w, int64_t n, int nm,
                                 double *u, double *lse, double *logpost, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(mix_posteriors_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, llk, logw_dev, w, n,
                       nm, u, lse, logpost);
    return hipGetLastError();
}

// two-stage deterministic reductions; work must hold >= 1024 doubles
hipError_t launch_reduce_max(const double *v, int64_t n, double *out_scalar, double *work, hipStream_t s) {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((reduce_stage_kernel<true>), dim3(blocks), dim3(256), 0, s, v, (const double *)nullptr, n, work, (const int *)nullptr);
    hipLaunchKernelGGL((reduce_stage_kernel<true>), dim3(1), dim3(256), 0, s, work, (const double *)nullptr,
                       (int64_t)blocks, out_scalar, (const int *)nullptr);
    return hipGetLastError();
}
hipError_t launch_reduce_sum(const double *v, const double *w, int64_t n, double *out_scalar, double *work,
                             hipStream_t s, const int *n_dev) {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((reduce_stage_kernel<false>), dim3(blocks), dim3(256), 0, s, v, w, n, work, n_dev);
    hipLaunchKernelGGL((reduce_stage_kernel<false>), dim3(1), dim3(256), 0, s, work, (const double *)nullptr,
                       (int64_t)blocks, out_scalar, (const int *)nullptr);
    return hipGetLastError();
}

// shifts of the mixture's component weights (mix.rs:312-323): a maximum that is not finite (no sample with a finite
// ln w + log r: an empty shard set, or all weights zero) becomes 0 -- on the device, after the all-reduce(MAX)
__global__ void mix_shift_kernel(double *mx, int nm) {
    for (int c = threadIdx.x; c < nm; c += blockDim.x) {
        const double v = mx[c];
        mx[c] = (v - v == 0.0) ? v : 0.0;
    }
}
hipError_t launch_mix_shift(double *mx, int nm, hipStream_t s) {
    hipLaunchKernelGGL(mix_shift_kernel, dim3(1), dim3(256), 0, s, mx, nm);
    return hipGetLastError();
}
// new log-weights (mix.rs:324-325, :335): logsum_c = ln(sum_c) + shift_c, then robust_log_softmax (mix.rs:14-18); one
// workgroup; out[0 .. nm) the log-weights, out[nm] = *llk (so that ONE copy brings both to the host).  Any number of
// components: out[] itself holds the logsums between the two sweeps.
__global__ void mix_logweights_kernel(const double *sums, const double *shift, const double *llk, int nm, double *out) {
    __shared__ double red[2];
    const int t = threadIdx.x;
    for (int c = t; c < nm; c += blockDim.x) out[c] = log(sums[c]) + shift[c];
    __syncthreads();
    if (t == 0) {  // (one thread, index order: the result does not depend on the launch shape)
        double mx = -INFINITY;
        for (int i = 0; i < nm; ++i) mx = fmax(mx, out[i]);
        double sm = 0.0;
        for (int i = 0; i < nm; ++i) sm += exp(out[i] - mx);
        red[0] = mx;
        red[1] = log(sm);
        out[nm] = llk ? *llk : 0.0;
    }
    __syncthreads();
    const double mx = red[0], ln = red[1];
    for (int c = t; c < nm; c += blockDim.x) out[c] = out[c] - mx - ln;
}
hipError_t launch_mix_logweights(const double *sums, const double *shift, const double *llk, int nm, double *out, hipStream_t s) {
    hipLaunchKernelGGL(mix_logweights_kernel, dim3(1), dim3(256), 0, s, sums, shift, llk, nm, out);
    return hipGetLastError();
}
// Posterior-weighted accumulation over mixture components (mix.rs:404-423, :447-461, :489-505):
//   out (first ? = : +=) exp(logpost[i][c]) * (dev ? a + (dev - mean)^2 : a)
__global__ void mix_accumulate_kernel(double *out, const double *a, const double *dev, const double *mean,
                                      const double *logpost, int c, int nm, int64_t n, int d, int first) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * d) return;
    const int64_t i = idx / d;
    const double wgt = exp(logpost[i * nm + c]);
    double v = a[idx];
    if (dev) {
        const double t = dev[idx] - mean[idx];
        v += t * t;
    }
    out[idx] = first ? wgt * v : out[idx] + wgt * v;
}
hipError_t launch_mix_accumulate(double *out, const double *a, const double *dev, const double *mean,
                                 const double *logpost, int c, int nm, int64_t n, int d, int first, hipStream_t s) {
    if (n <= 0 || d <= 0) return hipSuccess;
    hipLaunchKernelGGL(mix_accumulate_kernel, dim3((unsigned)((n * d + 255) / 256)), dim3(256), 0, s, out, a, dev, mean,
                       logpost, c, nm, n, d, first);
    return hipGetLastError();
}
hipError_t launch_exp_shift(const double *v, const double *max_dev, int64_t n, double *out, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(exp_shift_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, v, max_dev, n, out);
    return hipGetLastError();
}

}  // namespace ppca
