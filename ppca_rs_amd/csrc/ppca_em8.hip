// ppca_em8.hip -- the EM pass (E-step + every M-step reduction in ONE sweep over X, as pass_kernel) as an
// EIGHT-wave workgroup, two waves per SIMD, with two roles:
//
//   front waves 0-3   P1 staging of a tile's rows, P2 [G | b] (int8-sliced Gram + fp64-MFMA b = X~ C), P3 the
//                     per-sample k x k solve, P4a the x~-side statistics cross / sumx on the fp64 MFMA
//   back waves 4-7    P4b: S / U / totals (256 x 66) += Mask^T [wP | wz | w] on the INT8 MFMA: rows of [wP | wz | w] cut
//                     into seven signed bytes of a fixed-point form, 64 samples (two tiles) per v_mfma_i32_16x16x64_i8,
//                     the group's exact digit sums recombined into 24-bit pieces and added to fp64 accumulators
//                     (round 3: int64 accumulators, E8_ACC_F64=0)
//
// Why this split (round 3):  v_mfma_f64 shares the SIMD's vector port with every other vector
// instruction of every wave (tools/ubench_shadow.hip), and P4b was 8.8 k of a tile's 13.9 k fp64-MFMA cycles.  With
// one exact operand (the 0/1 mask) P4b does not need the fp64 pipe at all: on the int8 MFMA it is 70 x 16 cycles of
// the MATRIX pipe per tile plus ~700 integer vector instructions (digit cutting, recombination) -- work that leaves
// the fp64 port alone and so can run on the second wave of each SIMD in the gaps of the front role's dependent fp64
// chains (the Cholesky pivots, LDS operand waits).  The round-2 role split failed because its second role WAS fp64
// MFMAs; the round-2 int8 P4b failed because it ran on the same wave as everything else.  Registers: the back role
// holds the 160 accumulator registers (int64, so no fp64 conversion on the per-group path) and nothing of the solver;
// the front role holds the solver and none of the mask-side accumulators: both fit 256.
//
// Fixed-point form.  Column c of [wP | wz | w] has ONE exponent E_c per workgroup: I = rint(w 2^(50 - E_c)), |I| < 2^50,
// cut by one fused multiply-add onto the magic constant 1.5 2^52 + 0x808080808080 (the mantissa then IS the biased
// integer; xor with the constant's own bits leaves the balanced bytes).  E_c = exponent of the first tile's column
// maximum + 6.  A later tile whose entry does not fit raises a flag (any exponent field other than 0x433 after the
// add: too large, infinite or NaN alike); the back waves then contract what is pending under the old exponents,
// flush the int64 accumulators into the workgroup's partial (x 2^(E_c - 50), fp64), raise the exponents and cut the
// tile again -- a cold path, taken once at the first tile and whenever a column outgrows its scale.  The digit sums of a
// group are exact; the roundings are the cut itself (<= 2^-45 of the column's first-tile maximum per entry), one fp64
// addition per 24-bit piece and one multiplication per flush.  A non-finite column poisons its exponent: NaN reaches the
// statistics as through fp64.  Accumulators are also flushed every 100 groups (int64 form: 64 samples x 2^50 per group
// stay below 2^63).  Whether the cut was fine enough for the data at hand is checked on the REDUCED statistics by
// wguard_kernel (ppca_kernels.hip) from the per-column rounding bounds this role reports (p.errb).
//
// Hand-off (all in LDS): the front writes the tile's [wP | wz | w] rows in P3, ONE workgroup barrier per tile; the back
// cuts them into digit planes (even tile of a group: its own region; odd tile: the [G | b] buffer, free after P3) and
// contracts a group when its second tile is there.  The front waits on a counter of finished back iterations before
// it overwrites [G | b] or the rows (normally long satisfied: the back's ~4 k cycles per tile sit beside ~16 k of
// front work).  Per-dimension sample masks of the last three tiles are kept (16 bytes per dimension), so the staging
// of tile t+1 never touches what the contraction of tiles t-1, t reads.
//
// Replaces in the reference: infer (ppca/src/ppca_model.rs:221-227), the cross moment (:281-293), the d second-moment
// scans (:294-306), the noise 4-tuple (:328-358) and llk (:142-149) -- as pass_kernel, whose per-sample arithmetic,
// tile order and summation orders the front role keeps (cross, sumx, the scalars and the llk are bit-identical to it).
#include <atomic>
#include <cstdlib>

#include "ppca_device.hpp"

namespace ppca {

constexpr int E8_QW = 7;        // signed bytes per entry
constexpr int E8_F = 50;        // |I| < 2^F
constexpr int E8_HEAD = 6;      // binary orders kept free above the column maximum of the tile that set the scale
constexpr int E8_POISON = 100000;
constexpr int E8_EMIN = -900, E8_EMAX = 1000;
constexpr int E8_FLUSH_GROUPS = 100;
#ifndef E8_BARRIER_SLEEP
#define E8_BARRIER_SLEEP __builtin_amdgcn_s_sleep(1);
#endif
#ifndef E8_BACK_PRIO
#define E8_BACK_PRIO 0
#endif
#ifndef E8_FRONT_PRIO
#define E8_FRONT_PRIO 0
#endif
#ifndef E8_GS_PAD
#define E8_GS_PAD 18  // row stride of [G | b] / W rows = 16 NTP + 18 doubles: even, so that the solver's lane-per-sample accesses
                      // pair up into 16-byte LDS operations that spread over all banks (measured: 17 costs 2 %)
#endif
#ifndef E8_BARRIER_RTN
#define E8_BARRIER_RTN 0  // (measured: within the run-to-run noise)
#endif
#ifndef E8_EARLY_DIGDONE
#define E8_EARLY_DIGDONE 1
#endif
#ifndef E8_DECOUPLED
#define E8_DECOUPLED 0  // 1: the roles meet through LDS counters only; 0: a workgroup barrier per tile after P3 (round 3)
#endif
#ifndef E8_SHARED_FACTOR
#define E8_SHARED_FACTOR 0  // 0: every front wave factors every sample (rounds 1-3)
#endif

// Diagnostic counters of the back role (tests prove with them that the cold path and the periodic flush ran): [0] tiles cut
// again after a rescale of the fixed-point exponents (beyond a workgroup's first), [1] periodic flushes of the int64
// accumulators, [2] the largest number of tiles one workgroup walked, [3] launches.  One atomic each per workgroup.
__device__ unsigned long long e8_counters[4];

template <int K>
struct Cfg8 {
    using c = Cfg<K>;
    static constexpr int KP = c::KP, NTP = c::NTP, B = c::B, DP = c::DP, XS = c::XS, CS = c::CS;
    static constexpr int NC = KP + K + 1;        // statistic columns [wP | wz | w]
    static constexpr int NCT = (NC + 15) / 16;   // 16-column tiles of them
    static constexpr int NCOL = 16 * NCT;
    // [G | b] and the [wP | wz | w] rows share ONE buffer (row stride GS): a sample's Gram and b partial are dead once the
    // four front waves hold them in registers (a front barrier inside P3), its W row is written after that.
    //   as [G | b]:  G (16 NTP, K' used) | b partial of dims 0-127 (16) | pad
    //   as W row:    wP (K') ..          | wz (K) | w | 0 ..
    static constexpr int GS = 16 * NTP + E8_GS_PAD;
    static constexpr int WS = GS;
    static constexpr int BS = K + 1;             // b partial of dims 128-255
    static constexpr int PLANE_BYTES = E8_QW * 2 * NCOL * 16;  // digit planes of one tile: [plane][16-sample chunk][column][16 B]
    static constexpr int OFF_X = 0;
    static constexpr int OFF_C = OFF_X + B * XS;
    static constexpr int OFF_G = OFF_C + DP * CS;
    static constexpr int OFF_W = OFF_G;
    static constexpr int OFF_B1 = OFF_G + B * GS;
    static constexpr int OFF_M = OFF_B1 + B * BS;         // mask words, two parities x B x 4 u64
    static constexpr int OFF_MB = OFF_M + 2 * B * 4;      // sample masks per dimension: DP x 4 u32 (slot = tile % 3)
    static constexpr int OFF_S = OFF_MB + DP * 2;         // cross-wave scratch
    static constexpr int OFF_L = OFF_S + 2 * B;           // running scalars: sq[4][2 B] | dev | llk | w | ne | pm | px
    static constexpr int OFF_P0 = OFF_L + 14 * B;         // digit planes of a group's first tile
    static constexpr int OFF_P1 = OFF_P0 + PLANE_BYTES / 8;  // ... and of its second
    static constexpr int OFF_E = OFF_P1 + PLANE_BYTES / 8;  // column exponents (NCOL ints), flags
    static constexpr int OFF_BAR = OFF_E + NCOL / 2 + 4;  // counters: front barrier, back barrier, tiles digitised, violation stamp
    static constexpr int OFF_MU = OFF_BAR + 4;            // the mean (DP doubles, zero past d): re-read by the staging of every tile
    static constexpr int OFF_K = OFF_MU + DP;             // model scalars: sigma^2, 1 / sigma^2, ln sigma (re-read per tile)
    static constexpr int OFF_EB = OFF_K + 4;              // rounding bounds of the cut, per column (wguard_kernel)
    static constexpr int LDS_DOUBLES = OFF_EB + NCOL;
    static_assert(NCOL / 2 <= 64, "one back wave digitises NCOL / 2 (column, chunk) items");
    static_assert(LDS_DOUBLES * 8 <= 160 * 1024, "LDS budget");
};

// Barrier among the four waves of one role on a monotonic LDS counter.  A wave's LDS operations execute in order, so
// its add follows its stores; the others read only after seeing the count.
__device__ __forceinline__ void role_barrier(unsigned *ctr, unsigned &target, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    target += 4;
#if E8_BARRIER_RTN
    // the add returns the count it found: the LAST wave to arrive -- the one the others wait for -- leaves without a poll
    unsigned found = 0u;
    if (lane == 0) found = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if ((int)((unsigned)__builtin_amdgcn_readfirstlane(found) + 1u - target) >= 0) {
        asm volatile("" ::: "memory");
        return;
    }
#else
    if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
    for (;;) {
        const unsigned seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if ((int)(seen - target) >= 0) break;
        E8_BARRIER_SLEEP
    }
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void wait_counter(const unsigned *ctr, unsigned need) {
#if defined(E8_ONLY_FRONT) || defined(E8_ONLY_BACK)  // timing experiments with one role absent: nothing to wait for
    return;
#endif
    for (;;) {
        const unsigned seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if ((int)(seen - need) >= 0) break;
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}

#ifdef PPCA_PHASE_TIMING
#define E8_STAMP(i) { long long tn = clock64(); tph[i] += tn - tlast; tlast = tn; }
#define E8_FINE(i) { long long tn = clock64(); tfine[i] += tn - tfl; tfl = tn; }  // per-wave table (PPCA_PHASE_TIMING)
#elif defined(PPCA_MARKS)  // tools/devbuild.py -DPPCA_MARKS --asm: phase boundaries as comments in the ISA listing
#define E8_STAMP(i) asm volatile("; E8_MARK " #i ::: "memory");
#define E8_FINE(i)
#else
#define E8_STAMP(i)
#define E8_FINE(i)
#endif

// WEIGHTED: the dataset carries sample weights (PassArgs::w).  A template parameter because the weighted pass takes a
// logarithm per sample and tile (the unweighted one multiplies the determinants up and takes one per lane per kernel):
// the constants of that logarithm were what the register allocator spilled in the un-weighted hot kernel.
template <int K, bool GATHER, bool WEIGHTED>
__global__ __launch_bounds__(512) void em8_kernel(PassArgs p) {
    using cfg = Cfg8<K>;
    constexpr int KP = cfg::KP, NTP = cfg::NTP, B = cfg::B, XS = cfg::XS, CS = cfg::CS, GS = cfg::GS, BS = cfg::BS,
                  WS = cfg::WS, NC = cfg::NC, NCT = cfg::NCT, NCOL = cfg::NCOL;
    constexpr int NF = 4;              // waves per role
    constexpr int RPW = B / NF;        // rows staged per front wave
    constexpr int DPS = cfg::DP / 2;   // dims per K-split of b = X~ C
    constexpr int STEPS = DPS / 4;
    constexpr int RT = 16 / NF;        // 16-dim row tiles per wave in P4
    constexpr int DW = cfg::DP / NF;   // dims owned by a wave in P4
    constexpr int QW = E8_QW;
    static_assert(NTP <= NF, "int8 Gram: one front wave per packed-column tile");
    static_assert(QS == 8, "digit grouping assumes 8 slices");
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *Xs = sm + cfg::OFF_X;
    double *Cs = sm + cfg::OFF_C;
    double *Gs = sm + cfg::OFF_G;
    double *B1 = sm + cfg::OFF_B1;
    double *Ws = sm + cfg::OFF_W;
    unsigned long long *Ms = reinterpret_cast<unsigned long long *>(sm + cfg::OFF_M);
    unsigned *Mb = reinterpret_cast<unsigned *>(sm + cfg::OFF_MB);
    double *xxs = sm + cfg::OFF_S;
    double *scl = sm + cfg::OFF_L;
    int *Ex = reinterpret_cast<int *>(sm + cfg::OFF_E);
    unsigned *ctr = reinterpret_cast<unsigned *>(sm + cfg::OFF_BAR);
    unsigned *fbar = ctr, *bbar = ctr + 1, *digdone = ctr + 2, *vstamp = ctr + 3, *wready = ctr + 4, *itdone = ctr + 5;

    if (p.qflag) {  // qprep's dynamic-range guard: the fp64-Gram pass_kernel runs instead
        int unsafe = 0;
#pragma unroll
        for (int t = 0; t < NTP; ++t) unsafe |= p.qflag[t];
        if (unsafe) return;
    }
    const int tid = threadIdx.x, lane_entry = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool front = wave8 < NF;
    const int wave = wave8 & (NF - 1);  // index within the role
    const int d = p.d;
    const int64_t n = p.n_dev ? (int64_t)*p.n_dev : p.n;
    const double *mC = p.model + MODEL_HDR;
    const double *mMean = mC + (int64_t)d * K;
    const double s2_k = p.model[1], lnsig_k = p.model[2];

    for (int idx = tid; idx < cfg::DP * CS; idx += 512) {
        int j = idx / CS, a = idx - j * CS;
        Cs[idx] = (j < d && a < K) ? mC[(int64_t)j * K + a] : 0.0;
    }
    constexpr bool PAIRS = K >= 2;
    constexpr int SQW = PAIRS ? 2 * B : B;  // sq slots per front wave
    constexpr int L_DEV = NF * SQW, L_LLK = L_DEV + B, L_W = L_DEV + 2 * B, L_NE = L_DEV + 3 * B, L_PM = L_DEV + 4 * B,
                  L_PX = L_DEV + 5 * B;
    static_assert(L_DEV + 6 * B <= 14 * B, "scalar slots");
    for (int idx = tid; idx < L_DEV + 6 * B; idx += 512) scl[idx] = (idx >= L_PM && idx < L_PX) ? 1.0 : 0.0;
    for (int idx = tid; idx < cfg::DP * 4; idx += 512) Mb[idx] = 0u;
    for (int idx = tid; idx < cfg::DP; idx += 512) sm[cfg::OFF_MU + idx] = idx < d ? mMean[idx] : 0.0;
#ifdef E8_ONLY_BACK
    for (int idx = tid; idx < B * GS; idx += 512) Gs[idx] = 1.0;  // (something finite for the back role to cut)
#endif
    if (tid < 8) ctr[tid] = 0u;
    if (tid < NCOL) sm[cfg::OFF_EB + tid] = 0.0;
    if (tid == 0) {
        sm[cfg::OFF_K] = s2_k;
        sm[cfg::OFF_K + 1] = 1.0 / s2_k;
        sm[cfg::OFF_K + 2] = lnsig_k;
    }

    const int64_t ntiles = (n + B - 1) / B;
    const int64_t tiles_per_wg = (ntiles + gridDim.x - 1) / gridDim.x;
    const int64_t tile_begin = (int64_t)blockIdx.x * tiles_per_wg;
    const int64_t tile_end = tile_begin + tiles_per_wg < ntiles ? tile_begin + tiles_per_wg : ntiles;
#ifdef PPCA_PHASE_TIMING
    long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = clock64();
    long long tfine[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tfl = tlast;
#endif
    __syncthreads();

#ifdef E8_ONLY_FRONT
    if (!front) return;
#endif
#ifdef E8_ONLY_BACK
    if (front) return;
#endif
    if (!front) {
        // =========================================================== back role: P4b on the int8 MFMA
#ifndef E8_ACC_F64
#define E8_ACC_F64 1  // 1: the group's exact 24-bit pieces are added to fp64 accumulators (one rounding per piece: three per group
                      //    of 64 samples, where a plain fp64 sum takes 64); 0: to int64 accumulators (exact; rounds 3)
#endif
#if E8_ACC_F64
        typedef double acc_t;
#else
        typedef long long acc_t;
#endif
        acc_t accM[RT][NCT][4];
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int t = 0; t < NCT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) accM[r][t][q] = (acc_t)0;
        // (measured: the back role at a higher priority than the front costs 5 % -- it is off the front's critical path
        //  and only has to fill the gaps)
        if (E8_BACK_PRIO) __builtin_amdgcn_s_setprio(E8_BACK_PRIO);
        unsigned bbar_target = 0u;
        unsigned attempt = 0u;   // digitise attempts so far (the violation stamp of the current one)
        int pending = 0;         // 1: the previous tile's planes wait in P0 for their partner
        int have_scale = 0, flushed = 0, groups = 0;
        int n_rescale = 0, n_flush = 0;  // (wave-uniform: diagnostic counters)
        // Rounding bound of the cut, per column this lane covers (c = 16 t + l15), for wguard_kernel (ppca_kernels.hip): every
        // flush window adds 4 sqrt(rows of the window) quanta 2^(E_c - F) of the exponents it was cut under.
        double *ebs = sm + cfg::OFF_EB;  // (in LDS: touched at flushes only, by the role's first wave)
        int rows_win = 0;
        unsigned char *smb = reinterpret_cast<unsigned char *>(sm);
        constexpr int P0_BYTES = cfg::OFF_P0 * 8, PG_BYTES = cfg::OFF_P1 * 8;
        StatsLayout L(d, K);
        double *out = p.part + (int64_t)blockIdx.x * L.len;

        // [wP | wz | w] column c of W row -> its slot
        auto wsrc = [&](int c) { return c < KP ? c : 16 * NTP + (c - KP); };

        // ---- digit planes of the current tile's rows under the exponents Ex; returns (wave-uniform) whether an entry
        // of a live column did not fit.  Item = (column c, 16-sample chunk): NCOL / 2 items per wave.
        auto digitise = [&](int lane, int dst_bytes) -> bool {
            asm volatile("" : "+v"(lane));  // (addresses recomputed here, not hoisted and parked across the other phases)
            const bool active = lane < NCOL / 2;
            const int it = (NCOL / 2) * wave + (active ? lane : 0);
            const int c = it >> 1, chunk = it & 1;
            const bool cvalid = c < NC;
            const int src = cvalid ? wsrc(c) : 0;
            const int E = Ex[c];
            const bool poisoned = E > 5000;
            const double qsc = __hiloint2double((1023 + E8_F - (poisoned ? 0 : E)) << 20, 0);
            const double magic = __hiloint2double(0x43388080, (int)0x80808080);
            unsigned bad = 0u;
            unsigned pl[QW][4];
            const double *wsrcp = Ws + (16 * chunk) * WS + src;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                // four samples at a time (the live set stays small next to the 160 accumulator registers)
                unsigned wlo[4], whi[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // mantissa of w 2^(F - E) + magic = 2^51 + 0x808080808080 + I: xor with the constant's own bits leaves
                    // bytes 0..5 = the balanced digits and bits 48..51 = the top digit (4-bit two's complement)
                    const double wv = wsrcp[(4 * g4 + j) * WS];
                    const double v = __builtin_fma(cvalid ? wv : 0.0, qsc, magic);
                    const unsigned lo = (unsigned)__double2loint(v) ^ 0x80808080u;
                    const unsigned hi = (unsigned)__double2hiint(v) ^ 0x43388080u;
                    bad |= hi;  // any exponent field other than 0x433: the entry does not fit (or is not finite)
                    const int top = __builtin_amdgcn_sbfe((int)hi, 16, 4);
                    wlo[j] = lo;
                    whi[j] = __builtin_amdgcn_perm((unsigned)top, hi, 0x0C040100u);  // [d4, d5, d6, 0]
                }
                // 4 x 4 byte transposes: plane k of samples 4 g .. 4 g + 3 = bytes k of their four words
                auto tr4 = [&](const unsigned *w, unsigned *o0, unsigned *o1, unsigned *o2, unsigned *o3) {
                    const unsigned t0 = __builtin_amdgcn_perm(w[1], w[0], 0x05010400u), t1 = __builtin_amdgcn_perm(w[1], w[0], 0x07030602u);
                    const unsigned u0 = __builtin_amdgcn_perm(w[3], w[2], 0x05010400u), u1 = __builtin_amdgcn_perm(w[3], w[2], 0x07030602u);
                    *o0 = __builtin_amdgcn_perm(u0, t0, 0x05040100u);
                    *o1 = __builtin_amdgcn_perm(u0, t0, 0x07060302u);
                    *o2 = __builtin_amdgcn_perm(u1, t1, 0x05040100u);
                    if (o3) *o3 = __builtin_amdgcn_perm(u1, t1, 0x07060302u);
                };
                tr4(wlo, &pl[0][g4], &pl[1][g4], &pl[2][g4], &pl[3][g4]);
                tr4(whi, &pl[4][g4], &pl[5][g4], &pl[6][g4], nullptr);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (active) {
                unsigned char *wq = smb + dst_bytes;
#pragma unroll
                for (int sl = 0; sl < QW; ++sl)
                    *reinterpret_cast<i4_t *>(wq + ((sl * 2 + chunk) * NCOL + c) * 16) =
                        i4_t{(int)pl[sl][0], (int)pl[sl][1], (int)pl[sl][2], (int)pl[sl][3]};
            }
            const bool mine = active && cvalid && !poisoned && (bad >> 20) != 0u;
            return __builtin_amdgcn_ballot_w64(mine) != 0ull;
        };

        // ---- new exponents from the current tile's column maxima (cold path)
        auto rescale = [&](int lane) {
            asm volatile("" : "+v"(lane));
            const bool active = lane < NCOL / 2;
            const int it = (NCOL / 2) * wave + (active ? lane : 0);
            const int c = it >> 1, chunk = it & 1;
            const bool cvalid = c < NC;
            const int src = cvalid ? wsrc(c) : 0;
            double m = 0.0;
            bool fin = true;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const double av = __builtin_fabs(cvalid ? Ws[(16 * chunk + j) * WS + src] : 0.0);
                fin = fin && (av < __builtin_inf());
                m = __builtin_fmax(m, av);
            }
            m = __builtin_fmax(m, dpp_f64<0xB1, 0xF>(m));  // the other chunk of the column sits in the neighbouring lane
            const int finw = __builtin_amdgcn_update_dpp(0, fin ? 1 : 0, 0xB1, 0xF, 0xF, true);
            fin = fin && finw != 0;
            const int Eold = have_scale ? Ex[c] : E8_EMIN;
            int Enew = Eold;
            if (m > 0.0) {
                int e = __builtin_amdgcn_frexp_exp(m) + E8_HEAD;  // |w| < 2^(e - HEAD)
                e = e < E8_EMIN ? E8_EMIN : e;
                Enew = e > Eold ? e : Eold;
            }
            if (!fin || Enew > E8_EMAX) Enew = E8_POISON;
            if (Eold > 5000) Enew = Eold;  // (a poisoned column stays poisoned)
            if (active && chunk == 0) Ex[c] = Enew;
        };

        // ---- accumulators -> the workgroup's partial (x 2^(E - F)); C/D row of v_mfma_i32_16x16x64_i8 = 4 (lane / 16) + reg
        auto emit = [&](int lane, bool accumulate, bool clear) {
            asm volatile("" : "+v"(lane));
            const int l15 = lane & 15, l4 = lane >> 4;
            const double win = 4.0 * __builtin_sqrt((double)rows_win);
            rows_win = 0;
#pragma unroll
            for (int t = 0; t < NCT; ++t) {
                const int c = 16 * t + l15, a = c - KP;
                const int E = have_scale ? Ex[c] : 0;
#ifndef E8_NO_ERRB  // (A/B builds)
                if (wave == 0 && l4 == 0 && have_scale && E <= 5000) ebs[c] += win * __hiloint2double((1023 + E - E8_F) << 20, 0);
#endif
                const double fsc = __hiloint2double(E > 5000 ? 0x7FF80000 : (1023 + E - E8_F) << 20, 0);
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int dim = DW * wave + 16 * r + 4 * l4 + q;
                        double v = have_scale ? (double)accM[r][t][q] * fsc : 0.0;
                        if (E > 5000) v = fsc;  // poisoned column: NaN whatever the integers hold
                        if (clear) accM[r][t][q] = (acc_t)0;
                        if (dim < d && c < NC) {
                            double *dst = c < KP ? out + L.S + (int64_t)dim * KP + c
                                                 : (a < K ? out + L.U + (int64_t)dim * K + a : out + L.totals + dim);
                            *dst = accumulate ? *dst + v : v;
                        }
                    }
            }
        };

        // ---- one contraction: the group in [P0 | G-region planes] (both == false: P0 alone), sample masks of slots
        // slot_first / slot_second; digit sums folded into the int64 accumulators
        auto contract = [&](int lane, bool both, int slot_first, int slot_second) {
            asm volatile("" : "+v"(lane));
            const int l15 = lane & 15, l4 = lane >> 4, lh = l4 >> 1;
            i4_t af[RT];
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const unsigned word = Mb[(DW * wave + 16 * r + l15) * 4 + (lh ? slot_second : slot_first)];
                unsigned f = (word >> (16 * (l4 & 1))) & 0xFFFFu;
                f = (both || lh == 0) ? f : 0u;
                // rows were shifted in first-to-last: row r of a byte at bit 7 - r; after the bit reversal samples 0..7
                // of the chunk sit at bits 24..31, samples 8..15 at bits 16..23, ascending
                const unsigned g = __builtin_bitreverse32(f);
                af[r][0] = (int)((((g >> 24) & 0xFu) * 0x00204081u) & 0x01010101u);
                af[r][1] = (int)((((g >> 28) & 0xFu) * 0x00204081u) & 0x01010101u);
                af[r][2] = (int)((((g >> 16) & 0xFu) * 0x00204081u) & 0x01010101u);
                af[r][3] = (int)((((g >> 20) & 0xFu) * 0x00204081u) & 0x01010101u);
            }
            const unsigned char *wq = smb + (lh == 0 ? P0_BYTES : PG_BYTES) + ((l4 & 1) * NCOL + l15) * 16;
            constexpr int PSTRIDE = 2 * NCOL * 16;  // bytes between digit planes
            // The planes go through in three batches per column tile -- {0,1,2}, {3,4,5}, {6}: 24-bit pieces of the sums,
            // each added to the int64 accumulators on its own -- one block = one (column tile, batch, row tile).  The
            // blocks are software-pipelined two deep: the MFMAs of block i+1 are issued before the sums of block i are
            // folded, so the matrix pipe works while the integers are recombined (12 + 12 result registers in flight).
            constexpr int NBLK = NCT * 3 * RT;
            i4_t dd[2][3];
            i4_t bb[3];
            auto issue = [&](auto i_tag) {
                constexpr int i = decltype(i_tag)::value, t = i / (3 * RT), batch = (i / RT) % 3, r = i % RT;
                const unsigned char *wt = wq + t * 256 + 3 * batch * PSTRIDE;
                if constexpr (r == 0) {  // the batch's B operands, shared by its four row tiles
                    bb[0] = *reinterpret_cast<const i4_t *>(wt);
                    if constexpr (batch < 2) {
                        bb[1] = *reinterpret_cast<const i4_t *>(wt + PSTRIDE);
                        bb[2] = *reinterpret_cast<const i4_t *>(wt + 2 * PSTRIDE);
                    }
                }
                if constexpr (batch < 2) mfma_i8_x3(af[r], bb[0], bb[1], bb[2], dd[i & 1][0], dd[i & 1][1], dd[i & 1][2]);
                else mfma_i8_x1(af[r], bb[0], dd[i & 1][0]);
            };
            auto fold = [&](auto i_tag) {
                constexpr int i = decltype(i_tag)::value, t = i / (3 * RT), batch = (i / RT) % 3, r = i % RT;
                const i4_t *d3 = dd[i & 1];
#pragma unroll
                for (int q = 0; q < 4; ++q) {  // D row = 4 l4 + q (dim 16 r + 4 l4 + q), column l15
#if E8_ACC_F64
                    // the piece (|.| < 2^30) converts exactly; piece x 2^(24 batch) is exact; ONE rounding in the add.
                    // (v_lshl_add_u32 by hand: hipcc reassociates the Horner form into two shifts and a three-way add)
                    if constexpr (batch < 2) {
                        int pc;
                        asm("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(pc) : "v"(d3[2][q]), "v"(d3[1][q]));
                        asm("v_lshl_add_u32 %0, %1, 8, %2" : "=v"(pc) : "v"(pc), "v"(d3[0][q]));
                        if constexpr (batch == 0) accM[r][t][q] += (double)pc;
                        else accM[r][t][q] = __builtin_fma((double)pc, 0x1p24, accM[r][t][q]);
                    } else {
                        accM[r][t][q] = __builtin_fma((double)d3[0][q], 0x1p48, accM[r][t][q]);
                    }
#else
                    if constexpr (batch == 0) accM[r][t][q] += (long long)((((d3[2][q] << 8) + d3[1][q]) << 8) + d3[0][q]);  // |.| < 2^30
                    else if constexpr (batch == 1) accM[r][t][q] += (long long)((((d3[2][q] << 8) + d3[1][q]) << 8) + d3[0][q]) << 24;
                    else accM[r][t][q] += (long long)d3[0][q] << 48;
#endif
                }
            };
            issue(std::integral_constant<int, 0>{});
            static_for<NBLK>([&](auto i_tag) {
                constexpr int i = decltype(i_tag)::value;
                if constexpr (i + 1 < NBLK) issue(std::integral_constant<int, i + 1>{});
                fold(i_tag);
            });
        };

        for (int64_t tile = tile_begin; tile < tile_end; ++tile) {
            int lane = lane_entry;
            asm volatile("" : "+v"(lane));
            const int rel = (int)(tile - tile_begin);
            const bool last = tile + 1 == tile_end;
            const int slot_cur = rel % 3, slot_prev = (rel + 2) % 3;
            // front: P3(tile) done in all four waves -> the tile's W rows are final.  A counter, not a workgroup barrier: the
            // front never waits for this role here (the contraction of a group is two tiles' work on every second
            // tile: behind a barrier the front stood ~2.6 k cycles on those tiles)
#if E8_DECOUPLED
            wait_counter(wready, 4u * (unsigned)(rel + 1));
#else
            __syncthreads();
#endif
            E8_STAMP(5)
            E8_FINE(0)
#pragma unroll 1
            for (;;) {  // normally one trip
                ++attempt;
                const bool bad = !have_scale || digitise(lane, pending ? PG_BYTES : P0_BYTES);
                if (bad && lane == 0) __hip_atomic_store(vstamp, attempt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                E8_FINE(1)
                role_barrier(bbar, bbar_target, lane_entry);
                E8_STAMP(6)
                E8_FINE(2)
                const bool viol = __builtin_amdgcn_readfirstlane(__hip_atomic_load(vstamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == attempt;
                // What is contracted now: a fitting tile completes its group (or is the last tile: alone); a tile that does
                // not fit sends what is pending in alone, under the old exponents -- then (cold path) the integers leave
                // for the partial, the exponents rise and the tile is cut again.
                if (!viol) {  // every back wave has read the tile's rows for the last time: the front may overwrite them
                    if (lane_entry == 0) __hip_atomic_fetch_add(digdone, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    rows_win += B;
                }
                const bool con = pending || (!viol && last);
                if (con) {
                    contract(lane, !viol && pending, pending ? slot_prev : slot_cur, slot_cur);
                    ++groups;
                    E8_FINE(3)
                }
                pending = (!viol && !con) ? 1 : 0;
                if (viol ? have_scale != 0 : groups >= E8_FLUSH_GROUPS) {
                    emit(lane, flushed != 0, true);
                    flushed = 1;
                    groups = 0;
                    n_flush += viol ? 0 : 1;
                }
                if (!viol) break;
                n_rescale += have_scale;
                role_barrier(bbar, bbar_target, lane_entry);  // every wave has read the old exponents
                rescale(lane);
                have_scale = 1;
                role_barrier(bbar, bbar_target, lane_entry);
            }
            // this wave has read the sample masks and planes of its iteration for the last time (the front waits for this
            // before it stages the tile that reuses the oldest mask slot)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane_entry == 0) __hip_atomic_fetch_add(itdone, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            E8_STAMP(7)
        }
        emit(lane_entry, flushed != 0, false);
        if (p.errb && wave == 0 && lane_entry < 16) {
#pragma unroll
            for (int t = 0; t < NCT; ++t) p.errb[(int64_t)blockIdx.x * W_GUARD_NCOL + 16 * t + lane_entry] = ebs[16 * t + lane_entry];
        }
        if (tid == 256) {
            if (n_rescale) atomicAdd(&e8_counters[0], (unsigned long long)n_rescale);
            if (n_flush) atomicAdd(&e8_counters[1], (unsigned long long)n_flush);
            atomicMax(&e8_counters[2], (unsigned long long)(tile_end > tile_begin ? tile_end - tile_begin : 0));
            if (blockIdx.x == 0) atomicAdd(&e8_counters[3], 1ull);
        }
#ifdef PPCA_PHASE_TIMING
        if (p.dbg && tid == 256)
            for (int i = 5; i < 8; ++i) p.dbg[(int64_t)blockIdx.x * 16 + 8 + i] = (double)tph[i];
        if (p.dbg && lane_entry == 0)
            for (int i = 0; i < 16; ++i) p.dbg[(int64_t)gridDim.x * 16 + ((int64_t)blockIdx.x * 8 + wave8) * 16 + i] = (double)tfine[i];
#endif
        return;
    }

    // =============================================================== front role
    if (E8_FRONT_PRIO) __builtin_amdgcn_s_setprio(E8_FRONT_PRIO);
    unsigned fbar_target = 0u;
    d4_t accX[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) accX[r] = d4_t{0, 0, 0, 0};
    const double inv_s2_k = 1.0 / s2_k;
    double xr[RPW][4];
    const int64_t nleft = n - tile_begin * B;
    const int nrel = (int)(nleft < (1 << 30) ? nleft : (1 << 30));  // rows from the workgroup's first row to the end
    const double *Xwg = p.X + tile_begin * B * p.ldx;
    const int64_t own = (tile_end - tile_begin) * B;
    const int nmine = tile_end > tile_begin ? (int)(own < nleft ? own : nleft) : 0;
    const int *rows_wg = (GATHER && p.rows) ? p.rows + tile_begin * B : nullptr;
    const int rowbytes = d * (int)sizeof(double);
    // one buffer descriptor per tile (base = its first row, extent = its real rows): the row is a scalar offset, the
    // lane offset one constant VGPR, the half an immediate; rows past n read as zeros
    auto tile_rsrc = [&](int64_t tile) {
        const int rel0 = (int)(tile - tile_begin) * B;
        int cnt = nrel - rel0;
        cnt = __builtin_amdgcn_readfirstlane(cnt < 0 ? 0 : (cnt > B ? B : cnt));
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(Xwg + (int64_t)rel0 * p.ldx), 0, cnt * rowbytes, 0x00020000);
    };
    auto load_row = [&](const __amdgpu_buffer_rsrc_t &trs, int64_t tile, int r) {
        typedef unsigned u4_t __attribute__((ext_vector_type(4)));
        if constexpr (GATHER) {
            if (rows_wg) {
                const int rel = (int)(tile - tile_begin) * B + wave * RPW + r;
                const int rc = rel < nrel ? rel : nrel - 1;
                const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<double *>(p.X + (int64_t)rows_wg[rc < 0 ? 0 : rc] * p.ldx), 0, rc < 0 ? 0 : rowbytes, 0x00020000);
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
        for (int h = 0; h < 2; ++h) {
            const u4_t v = __builtin_amdgcn_raw_buffer_load_b128(trs, lane_entry * 16, (wave * RPW + r) * rowbytes + 1024 * h, 0);
            xr[r][2 * h] = __longlong_as_double(((long long)v[1] << 32) | v[0]);
            xr[r][2 * h + 1] = __longlong_as_double(((long long)v[3] << 32) | v[2]);
        }
    };
    const __amdgpu_buffer_rsrc_t qrsrc = __builtin_amdgcn_make_buffer_rsrc(p.qtab, 0, (int)qtab_bytes<K>(), 0x00020000);
    const bool gram_wave = NTP >= NF || wave < NTP;
    i4_t qbA[2][4];
    auto load_pair = [&](i4_t(&dst)[2][4], int sl0) {
        int qbase = wave * QS * 4 * 1024;
        asm volatile("" : "+s"(qbase));
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                typedef unsigned u4_t __attribute__((ext_vector_type(4)));
                const u4_t v = __builtin_amdgcn_raw_buffer_load_b128(qrsrc, lane_entry * 16 + kc * 1024, qbase + (sl0 + u) * 4096, 0);
                dst[u][kc] = i4_t{(int)v[0], (int)v[1], (int)v[2], (int)v[3]};
            }
    };
    // ---- P1: one tile = RPW rows per front wave.  The finite-test ballots ARE the mask words (word 2 h + e of a row,
    // bit l <-> dim 128 h + 2 l + e; qprep orders the digit table to match); each lane also shifts its own bit of every
    // ballot into st_mb (v_addc with the ballot as carry-in): after the wave's eight rows, byte = this dimension over
    // those samples (row r at bit 7 - r) -- the A operand of the back role's contraction.
    double mu[4], lim[4];  // (rebuilt per tile in stage_tile)
    int st_wlo = 0, st_whi = 0;
    int st_mb[4] = {0, 0, 0, 0};
    double xx_run = 0.0;  // sum_i w_i |x~_i|^2 of this wave's rows (sigma^2 and the llk are linear in it)
    auto stage_row = [&](int64_t t, int lane, auto r_tag) {
        constexpr int r = decltype(r_tag)::value;
        const int ri = wave * RPW + r;
        double pc_xx = 0.0;
        static_for<2>([&](auto h_tag) {
            constexpr int h = decltype(h_tag)::value;
            double xt[2];
            unsigned long long bal[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const double v = xr[r][2 * h + e];
                const bool ob = __builtin_fabs(v) < lim[2 * h + e];
                bal[e] = __builtin_amdgcn_ballot_w64(ob);
                xt[e] = ob ? v - mu[2 * h + e] : 0.0;  // select, never multiply (utils.rs:118-127)
            }
            file_mask<4 * r + 2 * h>(st_wlo, st_whi, st_mb[2 * h], bal[0]);
            file_mask<4 * r + 2 * h + 1>(st_wlo, st_whi, st_mb[2 * h + 1], bal[1]);
            typedef double d2_t __attribute__((ext_vector_type(2)));
            *reinterpret_cast<d2_t *>(Xs + ri * XS + 128 * h + 2 * lane) = d2_t{xt[0], xt[1]};
            pc_xx += xt[0] * xt[0];
            pc_xx += xt[1] * xt[1];
        });
        // wave-uniform: a real row of one of THIS workgroup's tiles
        const bool mine = (int)(t - tile_begin) * B + ri < nmine;
        const double wr = mine ? (WEIGHTED ? p.w[t * B + ri] : 1.0) : 0.0;  // (scalar load)
        xx_run += wr * pc_xx;
    };
    // staging lane map: lane l holds dims 128 h + 2 l + e (element q = 2 h + e) of a row.  The lane's four means and
    // limits (observed <=> |x| < lim: +inf for a real dimension -- the finite test of dataset.rs:19-22 --, -1 for the
    // padding past d) are rebuilt per tile from the LDS copy of the mean: as loop invariants they sat in 16 registers
    // across the solver and were what the register allocator spilled.
    auto stage_tile = [&](int64_t t, int lane) {
        {
            typedef double d2_t __attribute__((ext_vector_type(2)));
            const d2_t m0 = *reinterpret_cast<const d2_t *>(sm + cfg::OFF_MU + 2 * lane);
            const d2_t m1 = *reinterpret_cast<const d2_t *>(sm + cfg::OFF_MU + 128 + 2 * lane);
            mu[0] = m0[0]; mu[1] = m0[1]; mu[2] = m1[0]; mu[3] = m1[1];
#pragma unroll
            for (int q = 0; q < 4; ++q) lim[q] = (128 * (q >> 1) + 2 * lane + (q & 1) < d) ? __builtin_inf() : -1.0;
        }
        const int rel = (int)(t - tile_begin);
        st_wlo = st_whi = 0;
        st_mb[0] = st_mb[1] = st_mb[2] = st_mb[3] = 0;
        static_for<RPW>([&](auto r_tag) { stage_row(t, lane, r_tag); });
        const unsigned long long myw = ((unsigned long long)(unsigned)st_whi << 32) | (unsigned)st_wlo;
        if (lane < 4 * RPW) Ms[(rel & 1) * 4 * B + wave * 4 * RPW + lane] = myw;
        unsigned char *mbb = reinterpret_cast<unsigned char *>(Mb);
        const int slot = rel % 3;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            mbb[(128 * (q >> 1) + 2 * lane + (q & 1)) * 16 + 4 * slot + wave] = (unsigned char)st_mb[q];
    };

    if (tile_begin < tile_end) {
        const __amdgpu_buffer_rsrc_t trs = tile_rsrc(tile_begin);
#pragma unroll
        for (int r = 0; r < RPW; ++r) load_row(trs, tile_begin, r);
        load_pair(qbA, 6);
        stage_tile(tile_begin, lane_entry);
    }
    role_barrier(fbar, fbar_target, lane_entry);

    for (int64_t tile = tile_begin; tile < tile_end; ++tile) {
        int lane = lane_entry;
        asm volatile("" : "+v"(lane));
        const int l15 = lane & 15, l4 = lane >> 4;
        const int colb = (l15 < K) ? l15 : K;
        const int rel = (int)(tile - tile_begin);
        const unsigned long long *Msc = Ms + (rel & 1) * 4 * B;
        // ------------------------------------------------------------ P2: [G | b] of the tile
        {
            const int rt = wave & 1, kq = wave >> 1;
            const int si = 16 * rt + l15;
            d4_t accb = d4_t{0, 0, 0, 0};
            const double *xrow = Xs + si * XS + DPS * kq + l4;
            const double *cpc = Cs + (DPS * kq + l4) * CS + colb;
#if E8_EARLY_DIGDONE
            // the count of tiles the back role has cut, requested HERE and looked at where [G | b] is stored: by then it is
            // almost always enough, and the poll (an LDS round trip behind everything this wave has queued: ~0.7 k cycles per
            // tile in the phase table) is skipped
            const unsigned dd_early = __hip_atomic_load(digdone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
            i4_t af[2][4];
            double v[2][4];
            // one digit pair: contract, then fold the exact integer digit sums (|sum| <= 2^14) into the running fp64
            // value, Horner in 128^2
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
            {
                unsigned long long mwd[2][4];
#pragma unroll
                for (int rt2 = 0; rt2 < 2; ++rt2)
#pragma unroll
                    for (int kc = 0; kc < 4; ++kc) mwd[rt2][kc] = Msc[(16 * rt2 + l15) * 4 + kc];
                __builtin_amdgcn_sched_barrier(0);
                load_pair(qbB, 4);
                if (gram_wave) qs = p.qscale[16 * wave + l15];
                __builtin_amdgcn_sched_barrier(0);
                // A = mask bytes: lane (sample 16 rt2 + l15, k-chunk kc, 16 l4 .. +15 of it); 4 bits -> 4 bytes by one
                // multiply: (x * 0x204081) & 0x01010101 puts bit i of x into byte i
#pragma unroll
                for (int rt2 = 0; rt2 < 2; ++rt2)
#pragma unroll
                    for (int kc = 0; kc < 4; ++kc) {
                        const unsigned bits = (unsigned)(mwd[rt2][kc] >> (16 * l4)) & 0xFFFFu;
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            af[rt2][kc][u] = (int)((((bits >> (4 * u)) & 0xFu) * 0x00204081u) & 0x01010101u);
                    }
                group(qbA, true);   // digits {7,6}: requested during the previous tile's P4a
                load_pair(qbA, 2);
                group(qbB, false);  // digits {5,4}
                load_pair(qbB, 0);
            }
            {
#ifdef E8_C_GLOBAL
                // EXPERIMENT: the B operands from the zero-padded copy of C in global memory (L1 / L2) instead of the LDS tile
                // (would free 22.5 KB of LDS), requested two chunks of four k-steps ahead
                constexpr int CH = 4, NCH = STEPS / CH;
                const double *cg = p.cpad + (DPS * kq + l4) * CS + colb;
                double axb[2][CH], cbb[3][CH];
#pragma unroll
                for (int c0 = 0; c0 < 2; ++c0)
#pragma unroll
                    for (int u = 0; u < CH; ++u) cbb[c0][u] = cg[4 * (c0 * CH + u) * CS];
#pragma unroll
                for (int u = 0; u < CH; ++u) axb[0][u] = xrow[4 * u];
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if (c + 1 < NCH) {
#pragma unroll
                        for (int u = 0; u < CH; ++u) axb[(c + 1) & 1][u] = xrow[4 * ((c + 1) * CH + u)];
                    }
                    if (c + 2 < NCH) {
#pragma unroll
                        for (int u = 0; u < CH; ++u) cbb[(c + 2) % 3][u] = cg[4 * ((c + 2) * CH + u) * CS];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < CH; ++u) accb = mfma(axb[c & 1][u], cbb[c % 3][u], accb);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#else
                // b = X~ C: operands of the next four k-steps are requested before the current four MFMAs issue
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
#endif
            group(qbA, false);  // digits {3,2}
            group(qbB, false);  // digits {1,0}
            E8_STAMP(0)
            E8_FINE(0)
            // [G | b] shares its buffer with the previous tile's W rows: wait until the back role has cut them (long done:
            // the cut is the first thing the back does after the workgroup barrier)
#if E8_EARLY_DIGDONE
            if ((int)((unsigned)__builtin_amdgcn_readfirstlane(dd_early) - 4u * (unsigned)rel) < 0)
#endif
                wait_counter(digdone, 4u * (unsigned)rel);
            E8_FINE(1)
            if (gram_wave) {
#pragma unroll
                for (int rt2 = 0; rt2 < 2; ++rt2)
#pragma unroll
                    for (int r = 0; r < 4; ++r)  // C/D map of the 16x16 integer MFMA: row = 4 (lane >> 4) + reg
                        Gs[(16 * rt2 + 4 * l4 + r) * GS + 16 * wave + l15] = v[rt2][r] * qs;
            }
            // the two K-split partials of b are summed by the solver in a fixed order (p0 + p1)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (kq == 0) Gs[(16 * rt + l4 + 4 * r) * GS + 16 * NTP + l15] = accb[r];
                else if (l15 < K + 1) B1[(16 * rt + l4 + 4 * r) * BS + l15] = accb[r];
            }
        }
        E8_FINE(2)
        role_barrier(fbar, fbar_target, lane_entry);
        E8_STAMP(1)
        E8_FINE(3)
        // ------------------------------------------------------------ P3: per-sample k x k solve
        // Every front wave factors every sample (lane = sample, lanes 32-63 mirror 0-31) and the waves share the
        // independent columns of M^-1, two per instruction stream (lane i: column 2p, lane i + 32: column 2p + 1);
        // wave 0 also owns z, llk and the scalars.  (The rows of W are free: the back role has cut the previous
        // tile's -- the counter above.)
        if (PAIRS || lane < B) {
            // (the model scalars come back from LDS per tile: as loop invariants they -- and what hipcc derives from them --
            //  were parked in spilled vector registers across the whole tile loop)
            const double s2 = sm[cfg::OFF_K], inv_s2 = sm[cfg::OFF_K + 1], lnsig = sm[cfg::OFF_K + 2];
            const int i = lane & (B - 1);
            const int hi = PAIRS ? lane >> 5 : 0;
            const int64_t row = tile * B + i;
            const double *g0 = Gs + i * GS;
            const double *b1 = B1 + i * BS;
            const double wgt = (row < n) ? (WEIGHTED ? p.w[row] : 1.0) : 0.0;
            const int m = __popcll(Msc[i * 4]) + __popcll(Msc[i * 4 + 1]) + __popcll(Msc[i * 4 + 2]) + __popcll(Msc[i * 4 + 3]);
            double *wrow = Ws + i * WS;
            double sc_sq = 0.0, sc_dev = 0.0, sc_llk = 0.0, sc_w = 0.0, sc_ne = 0.0;
            Posterior<K> post;
            double pm = 1.0;
            int pe = 0;
            double z[K], quad = 0.0, zz = 0.0;
#if E8_SHARED_FACTOR
            // ONE front wave -- rotating with the tile, so that every SIMD carries the solver once in four tiles -- factors
            // the tile's samples and solves for z; the factor, z and the by-products (quad, |z|^2, det M) go back into the
            // sample's [G | b] row, where the other waves pick up what their columns of M^-1 need.  (Rounds 1-3: every
            // front wave factored every sample: 3 x (328 + 140) vector instructions per tile on a port that is the
            // kernel's bound.)
            constexpr int XB = 16 * NTP + 12;  // free slots of the row (the b partial used K + 1 <= 11 of its 16)
            if (wave == (rel & (NF - 1))) {
                Posterior<K> fac;
                double y[K], fq, fz, fm;
                int fe;
                fac.load([&](int e) { return g0[e]; }, s2);
#pragma unroll
                for (int a = 0; a < K; ++a) y[a] = g0[16 * NTP + a] + b1[a];
                fac.factor_loaded(fm, fe);
                fac.solve_loaded(y, fq, fz);
                if (hi == 0) {
                    double *g0w = Gs + i * GS;
#pragma unroll
                    for (int e = 0; e < KP; ++e) g0w[e] = fac.L[e];
#pragma unroll
                    for (int a = 0; a < K; ++a) g0w[16 * NTP + a] = y[a];
                    g0w[XB] = fq;
                    g0w[XB + 1] = fz;
                    g0w[XB + 2] = fm;
                    g0w[XB + 3] = (double)fe;
                }
            }
            role_barrier(fbar, fbar_target, lane_entry);
#pragma unroll
            for (int e = 0; e < KP; ++e) post.L[e] = g0[e];
#pragma unroll
            for (int a = 0; a < K; ++a) z[a] = g0[16 * NTP + a];
            quad = g0[XB];
            zz = g0[XB + 1];
            pm = g0[XB + 2];
            pe = (int)g0[XB + 3];
            // the W rows go where [G | b] is: every front wave holds its operands before any of them writes a row
            role_barrier(fbar, fbar_target, lane_entry);
#else
            post.load([&](int e) { return g0[e]; }, s2);
#pragma unroll
            for (int a = 0; a < K; ++a) z[a] = g0[16 * NTP + a] + b1[a];
            // the W rows go where [G | b] is: every front wave holds its operands before any of them writes a row
            E8_FINE(4)
            role_barrier(fbar, fbar_target, lane_entry);
            E8_FINE(5)
            post.factor_loaded(pm, pe);
            E8_FINE(6)
            post.solve_loaded(z, quad, zz);
            E8_FINE(7)
#endif
            double trpart = 0.0;
            if constexpr (PAIRS) {
#pragma unroll
                for (int pp = 0; pp < (K + 1) / 2; ++pp) {
                    if (pair_owner(K, pp, NF) != wave) continue;
                    const int c0 = 2 * pp;
                    const double zc = (hi && c0 + 1 < K) ? z[c0 + 1 < K ? c0 + 1 : c0] : z[c0];
                    // P = z z^T + Sigma, Sigma = sigma^2 M^-1 (ppca_model.rs:437-439), weighted
                    trpart += post.minv_column_pair(c0, hi, [&](int t, double v, bool ok) {
                        if (ok && c0 + hi < K) wrow[tri(t, c0) + hi] = wgt * (z[t] * zc + s2 * v);
                    });
                }
            } else {
#pragma unroll
                for (int c = 0; c < K; ++c) {
                    if (column_owner(K, c, NF) != wave) continue;
                    trpart += post.minv_column(c, [&](int a, int cc, double v) { wrow[tri(a, cc)] = wgt * (z[a] * z[cc] + s2 * v); });
                }
            }
            E8_FINE(8)
            // tr(C_o Sigma C_o^T) = <Sigma, G> = s2 (K - s2 tr M^-1)  (:345); all-masked samples are filtered out (:333)
            if (m > 0) sc_sq -= wgt * s2 * s2 * trpart;
            if (hi == 0) {
                // W row = [w P (K') | 0.. | w z (K) | w | 0..].  EVERY front wave writes the [w z | w] part (the same values:
                // each has solved for z): P4a then reads what the wave itself wrote, and no front wave waits for another
                // between P3 and P4a.
                double *zrow = wrow + 16 * NTP;
#pragma unroll
                for (int a = 0; a < K; ++a) zrow[a] = wgt * z[a];
                zrow[K] = wgt;
            }
            if (wave == 0 && hi == 0) {
                const double run_dev = scl[L_DEV + i], run_llk = scl[L_LLK + i], run_w = scl[L_W + i], run_ne = scl[L_NE + i];
                const double run_pm = scl[L_PM + i], run_px = scl[L_PX + i];
                if (m > 0) {
                    sc_sq += wgt * s2 * (double)K;
                    sc_dev += wgt * (0.0 - quad - s2 * zz);  // |x~ - C_o z|^2 minus |x~|^2, added in the epilogue (:346)
                    sc_ne += (row < n) ? 1.0 : 0.0;
                }
                const double lk0 = sample_llk_nolog(0.0, quad, inv_s2, lnsig, m, K);
                if constexpr (WEIGHTED) {
                    if (!p.no_llk) sc_llk += wgt * (m > 0 ? lk0 - 0.5 * Posterior<K>::logdet(pm, pe) : 0.0);
                } else {
                    const bool use = m > 0 && row < n;  // wgt is 1 for real rows
                    sc_llk += use ? lk0 : 0.0;
                    int e;
                    scl[L_PM + i] = frexp(run_pm * (use ? pm : 1.0), &e);
                    scl[L_PX + i] = run_px + (double)(e + (use ? pe : 0));
                }
                sc_w += wgt;
                scl[L_DEV + i] = run_dev + sc_dev;
                scl[L_LLK + i] = run_llk + sc_llk;
                scl[L_W + i] = run_w + sc_w;
                scl[L_NE + i] = run_ne + sc_ne;
            }
            scl[wave * SQW + (PAIRS ? lane : i)] += sc_sq;  // (read here, not at the top: the value would sit in a spilled register across the solve)
        }
        E8_STAMP(2)
        E8_FINE(9)
        // this wave's part of the tile's W rows is final: the back role starts on them when all four have said so
#if E8_DECOUPLED
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane_entry == 0) __hip_atomic_fetch_add(wready, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
        __syncthreads();
#endif
        E8_FINE(10)
        // ------------------------------------------------------------ P4a: cross / sumx += X~^T [wz | w]
        // the only reader of the x~ tile; the next tile's rows are requested one per k-step behind the MFMAs
        {
            load_pair(qbA, 6);  // the next tile's first digit pair (the table does not depend on the tile)
#ifndef E8_P4A_AHEAD
#define E8_P4A_AHEAD 1  // k-steps the LDS operands are requested ahead of their MFMAs (measured: 2 and 3 change nothing -- 95.9 / 95.8 / 95.4 it/s -- and cost registers)
#endif
            constexpr int AH = E8_P4A_AHEAD, NB3 = AH + 1;
            double bzb[NB3], axb[NB3][RT];
            const __amdgpu_buffer_rsrc_t trs = tile_rsrc(tile + 1);
#pragma unroll
            for (int s0 = 0; s0 < AH; ++s0) {
                const int smp = 4 * s0 + l4;
                bzb[s0] = Ws[smp * WS + 16 * NTP + l15];
#pragma unroll
                for (int r = 0; r < RT; ++r) axb[s0][r] = Xs[smp * XS + DW * wave + 16 * r + l15];
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (s < RPW) load_row(trs, tile + 1, s);  // unconditional (rows past the end read as zeros)
                if (s + AH < 8) {
                    const int smp = 4 * (s + AH) + l4;
                    bzb[(s + AH) % NB3] = Ws[smp * WS + 16 * NTP + l15];
#pragma unroll
                    for (int r = 0; r < RT; ++r) axb[(s + AH) % NB3][r] = Xs[smp * XS + DW * wave + 16 * r + l15];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < RT; ++r) accX[r] = mfma(axb[s % NB3][r], bzb[s % NB3], accX[r]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        E8_FINE(11)
        role_barrier(fbar, fbar_target, lane_entry);  // the x~ tile is free
        E8_STAMP(3)
        E8_FINE(12)
        // ------------------------------------------------------------ P1 of the next tile
        // (its sample masks go into the slot of tile - 2: the back role's iterations up to tile - 1 must be over -- it
        //  normally is one digitise into iteration `tile`)
#if E8_DECOUPLED
        wait_counter(itdone, 4u * (unsigned)rel);
#endif
        stage_tile(tile + 1, lane);
        E8_FINE(13)
        role_barrier(fbar, fbar_target, lane_entry);
        E8_STAMP(4)
        E8_FINE(14)
    }
#ifdef PPCA_PHASE_TIMING
    if (p.dbg && tid == 0)
        for (int i = 0; i < 5; ++i) p.dbg[(int64_t)blockIdx.x * 16 + 8 + i] = (double)tph[i];
    if (p.dbg && lane_entry == 0)
        for (int i = 0; i < 16; ++i) p.dbg[(int64_t)gridDim.x * 16 + ((int64_t)blockIdx.x * 8 + wave8) * 16 + i] = (double)tfine[i];
#endif

    // ---------------------------------------------------------------- epilogue (front waves)
    {
        const int lane = lane_entry, l15 = lane & 15, l4 = lane >> 4;
        const double sq_w = wave_sum(lane < SQW ? scl[wave * SQW + lane] : 0.0);
        const double xx_w = wave_sum(xx_run);
        if (lane == 0) {
            xxs[wave] = sq_w;
            xxs[NF + wave] = xx_w;
        }
        role_barrier(fbar, fbar_target, lane_entry);
        StatsLayout L(d, K);
        double *out = p.part + (int64_t)blockIdx.x * L.len;
        if (wave == 0) {
            double v0 = 0.0, xx_tot = 0.0;
#pragma unroll
            for (int w = 0; w < NF; ++w) v0 += xxs[w];
#pragma unroll
            for (int w = 0; w < NF; ++w) xx_tot += xxs[NF + w];
            const int li = lane < B ? lane : 0;
            double sc_llk = scl[L_LLK + li];
            if constexpr (!WEIGHTED) sc_llk -= 0.5 * (log(scl[L_PM + li]) + scl[L_PX + li] * LN_2);
            const double v1 = wave_sum(lane < B ? scl[L_DEV + li] : 0.0), v2 = wave_sum(lane < B ? sc_llk : 0.0),
                         v3 = wave_sum(lane < B ? scl[L_W + li] : 0.0), v4 = wave_sum(lane < B ? scl[L_NE + li] : 0.0);
            if (lane == 0) {
                double *sc = out + L.scalars;
                sc[SC_SQERR] = v0;
                sc[SC_DEVSQ] = v1 + xx_tot;
                sc[SC_LLK] = v2 - 0.5 * inv_s2_k * xx_tot;
                sc[SC_SUMW] = v3;
                sc[SC_NONEMPTY] = v4;
                sc[5] = 0.0;
                sc[6] = 0.0;
                sc[7] = 0.0;
            }
        }
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int dim = DW * wave + 16 * r + l4 + 4 * q;  // C/D row of v_mfma_f64_16x16x4
                if (dim >= d) continue;
                if (l15 < K) out[L.cross + (int64_t)dim * K + l15] = accX[r][q];
                else if (l15 == K) out[L.sumx + dim] = accX[r][q];
            }
    }
}

// ------------------------------------------------------------------ launcher
template <int K, bool GATHER, bool WEIGHTED>
static hipError_t launch_em8_t(int grid, const PassArgs &a, hipStream_t s) {
    const size_t lds = sizeof(double) * Cfg8<K>::LDS_DOUBLES;
    static std::atomic<unsigned long long> done{0ull};
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&em8_kernel<K, GATHER, WEIGHTED>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        done.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL((em8_kernel<K, GATHER, WEIGHTED>), dim3(grid), dim3(512), lds, s, a);
    return hipGetLastError();
}

hipError_t em8_debug_counters(unsigned long long *out4, int reset, hipStream_t s) {
    if (hipError_t e = hipMemcpyFromSymbolAsync(out4, HIP_SYMBOL(e8_counters), sizeof(unsigned long long) * 4, 0, hipMemcpyDeviceToHost, s);
        e != hipSuccess)
        return e;
    if (hipError_t e = hipStreamSynchronize(s); e != hipSuccess) return e;
    if (reset) {
        const unsigned long long z[4] = {0, 0, 0, 0};
        if (hipError_t e = hipMemcpyToSymbolAsync(HIP_SYMBOL(e8_counters), z, sizeof(z), 0, hipMemcpyHostToDevice, s); e != hipSuccess) return e;
        return hipStreamSynchronize(s);
    }
    return hipSuccess;
}

bool em8_covers(int k) {
#ifdef PPCA_DEV_K10
    return k == 10;
#else
    return k >= 1 && k <= FUSED_MAX_K;
#endif
}

hipError_t launch_em8(int k, int grid, const PassArgs &a, hipStream_t s) {
    const bool gather = a.rows != nullptr;
#define PPCA_E8_CASE(KK) \
    case KK:             \
        return gather ? launch_em8_t<KK, true, true>(grid, a, s) : (a.w ? launch_em8_t<KK, false, true>(grid, a, s) : launch_em8_t<KK, false, false>(grid, a, s));
    switch (k) {
#ifdef PPCA_DEV_K10
        PPCA_E8_CASE(10)
#else
        PPCA_E8_CASE(1) PPCA_E8_CASE(2) PPCA_E8_CASE(3) PPCA_E8_CASE(4) PPCA_E8_CASE(5) PPCA_E8_CASE(6) PPCA_E8_CASE(7)
        PPCA_E8_CASE(8) PPCA_E8_CASE(9) PPCA_E8_CASE(10)
#endif
        default: return hipErrorInvalidValue;
    }
#undef PPCA_E8_CASE
}

}  // namespace ppca
