// ppca_em9.hip -- the EM pass of ppca_em8.hip (eight-wave workgroup, two roles: fp64 front, int8 statistics contraction in
// the back) with the per-sample SOLVE PIPELINED ACROSS TILES (round 4; the default EM kernel, PPCA_EM9=0 selects em8_kernel):
//
//   em8_kernel    every front wave factors every sample of the tile and solves for z (260 + 130 vector instructions on each
//                 of the four waves), then the waves share the columns of M^-1.  The redundancy is free in time -- the waves
//                 run in lockstep -- but the solve sits on every wave's chain: load, factor, solve, columns, scalars =
//                 4.7 k of a tile's ~20 k cycles.
//   em9_kernel    ONE front wave (rotating with the tile) factors tile t, solves for z and writes the factor, z and the
//                 [wz | w] part of the W rows back into the tile's [G | b] buffer, while the OTHER THREE form the columns
//                 of M^-1 and the wP part of the W rows of tile t - 1 from the factor its solver left there one tile
//                 earlier.  The two activities take about the same time (~3 k cycles), so the chain of a tile loses
//                 ~1.7 k cycles, and three of four waves skip the factorisation (1 400 vector instructions per tile).
//                 What it needs: the [G | b] / factor / W-row buffer TWICE (tile parity) -- the 21 KB come from the C tile,
//                 whose B operands of b = X~ C are read from a zero-padded copy of C in global memory instead (measured
//                 alone: - 2.3 %) --, the back role one tile later (the W rows of tile t - 1 are final after phase beta of
//                 tile t: the roles meet through LDS counters only, as in em8's E8_DECOUPLED form) and the per-dimension
//                 sample masks four tiles deep.
//
// Everything else -- staging, [G | b], the fixed-point form of the back role and its guard, the cross product, the
// epilogue -- is em8_kernel's, statement for statement; see ppca_em8.hip for the description.
//
// Round 5 ("layout B", E9_CLDS=1, the default): b = X~ C runs on v_mfma_f64_4x4x4 -- the instruction's four blocks are (two groups
// of four dimensions) x (two groups of four samples), so one MFMA covers 8 samples x 4 columns x 8 dimensions and K = 10 columns
// are three groups of four -- with the C^T operands of the first two column groups RESIDENT IN LDS in operand order; the 16 KB
// for them come from the digit planes packed to their real columns, the second K-half's partial of b in the row's own free
// slots, a 78-double row stride at k = 10, the mean re-read from L1 / L2 and a register running sum (Cfg9 below).  Measured
// 103.5 against 100.3 EM it/s at N = 10 M; with nt row loads and the second digit pair requested one phase early 104.2-104.8.
#include <atomic>
#include <cstdlib>

#include "ppca_device.hpp"

namespace ppca {

constexpr int E9_QW = 7;        // signed bytes per entry
constexpr int E9_F = 50;        // |I| < 2^F
constexpr int E9_HEAD = 6;      // binary orders kept free above the column maximum of the tile that set the scale
constexpr int E9_POISON = 100000;
constexpr int E9_EMIN = -900, E9_EMAX = 1000;
constexpr int E9_FLUSH_GROUPS = 100;
#ifndef E9_BARRIER_SLEEP
#define E9_BARRIER_SLEEP __builtin_amdgcn_s_sleep(1);
#endif
#ifndef E9_BACK_PRIO
#define E9_BACK_PRIO 0
#endif
#ifndef E9_FRONT_PRIO
#define E9_FRONT_PRIO 0
#endif
#ifndef E9_CLDS
#define E9_CLDS 1  // 1 (round 5, "layout B"; measured 103.7 against 100.3 EM it/s, gpurun_out/r5ab4; 0: round 4's layout): b = X~ C on the 4 x 4 x 4 form with the first two column groups of C RESIDENT IN LDS (16 KB; the third
                   // from L1 / L2), the LDS for it reclaimed at k = 10 from: digit planes packed to their real columns (6.1 KB), the second K-half
                   // partial of b in the row's free slots instead of its own array (2.8), the row stride 82 -> 78 (2), the mean re-read from L1 / L2
                   // instead of an LDS copy (2), the per-lane running sum of tr(C Sigma C^T) in a register instead of LDS (2)
#endif
#ifndef E9_B444
#define E9_B444 E9_CLDS
#endif
#if E9_CLDS && !E9_B444
#error "E9_CLDS is the 4 x 4 x 4 form of the b product"
#endif
// E9_B444 alone (C operands from L1 / L2): b = X~ C on v_mfma_f64_4x4x4 with the dimension and sample groups in the instruction's four blocks (round 5
                   // experiment, parity-green).  Measured (gpurun_out/r5ab1, r5ab2, r5t1, r5t2; profiles/r05/README.md): 100.0 against
                   // 100.7 EM it/s for the 16 x 16 x 4 form -- the loop's 96 MFMAs of 17 cycles take 2.84 k cycles per tile against
                   // 2.70 k for 32 of 64: the 48 loads of C per wave and tile queue behind the Gram's digit-table loads (one in-order
                   // vmcnt), deeper look-ahead changes nothing (E9_LAC=6: 99.9).  With the C operands from LDS (timing experiment
                   // E9_EXP_CLDS, results wrong) the loop takes 2.22 k and the launch 104.8 it/s -- but LDS has 1.7 KB free where 20 KB are
                   // needed: E9_CLDS reclaims 16 for two of the three column groups.
#ifndef E9_B_EARLYC
#define E9_B_EARLYC 1  // the first steps of C operands requested before the Gram's digit pairs (their L2 latency under the integer MFMAs)
#endif
#ifndef E9_X_AUX
#define E9_X_AUX 2  // cache policy of the row loads of X: nt (L2-served: the rows are read once and would only push the digit table and the
                    // third column group of C out of the 32 KB L1).  Measured on layout B, two rounds interleaved: 103.9-104.1 against
                    // 103.2-103.4 EM it/s (round 4's layout: +0.5 %, inside the spread)
#endif
#ifndef E9_Q_AUX
#define E9_Q_AUX 0  // ... and of the Gram's digit table (128 KB per tile and workgroup).  Measured with 2 (nt): 90.0 against 100.7 it/s
#endif
#ifndef E9_LAC
#define E9_LAC 4
#endif
#ifndef E9_QB_EARLY
#define E9_QB_EARLY 1  // the second digit pair {5,4} requested with {7,6} during the previous tile's P4a.  On round 4's layout it only moved the
                       // wait (first half of the Gram 2.40 -> 2.12 k cycles per tile, b loop 2.70 -> 2.87 k: 100.7 against 100.8 it/s); on layout
                       // B, whose b loop waits for 16 instead of 48 loads, + 0.3 % (with nt row loads 104.2-104.4 against 103.2-103.4)
#endif
#ifndef E9_HEAVY_BUDGET
#define E9_HEAVY_BUDGET 8  // tiles per flush window (200 tiles) whose rows above the scale go round the fixed-point form (heavy_add)
#endif
#ifndef E9_PLANES_PACKED
#define E9_PLANES_PACKED 1
#endif
#ifndef E9_GS_PAD
#define E9_GS_PAD 18  // row stride of [G | b] / W rows = 16 NTP + 18 doubles: even, so that the solver's lane-per-sample accesses
                      // pair up into 16-byte LDS operations that spread over all banks (measured: 17 costs 2 %)
#endif

#ifdef PPCA_PHASE_TIMING  // per-wave phase sums into PassArgs::dbg (printed by ppca_capi.hip with em8's column names)
#define E9_FINE(i) { __builtin_amdgcn_sched_barrier(0); long long tn = clock64(); tfine[i] += tn - tfl; tfl = tn; __builtin_amdgcn_sched_barrier(0); }
#else
#define E9_FINE(i)
#endif

// Diagnostic counters of the back role (tests prove with them that the cold path and the periodic flush ran): [0] tiles cut
// again after a rescale of the fixed-point exponents (beyond a workgroup's first), [1] periodic flushes of the int64
// accumulators, [2] the largest number of tiles one workgroup walked, [3] launches.  One atomic each per workgroup.
__device__ unsigned long long e9_counters[4];

template <int K>
struct Cfg9 {
    using c = Cfg<K>;
    static constexpr int KP = c::KP, NTP = c::NTP, B = c::B, DP = c::DP, XS = c::XS, CS = c::CS;
    static constexpr int NC = KP + K + 1;        // statistic columns [wP | wz | w]
    static constexpr int NCT = (NC + 15) / 16;   // 16-column tiles of them
    static constexpr int NCOL = 16 * NCT;
    // [G | b], then the factor + z, then the W row of a sample share ONE row (stride GS) of the buffer of the tile's parity:
    //   as [G | b]:      G (16 NTP, K' used)        | b partial of dims 0-127 (16)          | pad (2)
    //   after the solve: L (K')  .. z in free slots | w z (K) | w | z in free slots ..      | z
    //   as W row:        w P (K') ..                | w z (K) | w | ..
    static constexpr bool CLDS = E9_CLDS != 0;
    static constexpr int NCGB = (K + 3) / 4;                          // column groups of four of b = X~ C (4 x 4 x 4 form)
    static constexpr int NCL = CLDS ? (NCGB < 2 ? NCGB : 2) : 0;      // ... whose C operands are resident in LDS
    static constexpr int PADC = (CLDS && K == 10) ? 14 : E9_GS_PAD;   // width of the row's b area + pad (layout B squeezes k = 10)
    static constexpr int GS = 16 * NTP + PADC;
    static constexpr int WS = GS;
    static constexpr int BS = CLDS ? 0 : K + 1;  // b partial of dims 128-255 (layout B: in the row's free slots, p1slot below)
    // digit planes of one tile: [plane][16-sample chunk][column][16 B].  (round 5) NCP = the NC real columns, not the NCOL of the 16-column
    // tiles: the lanes of the last column tile that stand for columns >= NC read whatever follows (the next chunk row, or the words
    // behind the region): their digit sums are never emitted (emit: c < NC) -- 14 columns x 14 rows x 16 B x 2 regions = 6.1 KB of LDS.
    static constexpr int NCP = E9_PLANES_PACKED ? NC : NCOL;
    static constexpr int PLANE_BYTES = E9_QW * 2 * NCP * 16;
    static constexpr int OFF_X = 0;
    static constexpr int OFF_G = OFF_X + B * XS;          // two buffers: tile parity
    static constexpr int OFF_W = OFF_G;
    static constexpr int OFF_B1 = OFF_G + 2 * B * GS;
    static constexpr int OFF_M = OFF_B1 + B * BS;         // mask words, two parities x B x 4 u64
    static constexpr int OFF_MB = OFF_M + 2 * B * 4;      // sample masks per dimension: DP x 4 u32 (slot = tile % 4)
    static constexpr int OFF_S = OFF_MB + DP * 2;         // cross-wave scratch
    static constexpr int OFF_L = OFF_S + 2 * B;           // running scalars: sq[4][2 B] | dev | llk | w | ne | pm | px
    static constexpr int OFF_P0 = OFF_L + (CLDS ? 6 : 14) * B;  // digit planes of a group's first tile (layout B: sq lives in a register)
    static constexpr int OFF_P1 = OFF_P0 + PLANE_BYTES / 8;  // ... and of its second
    static constexpr int OFF_E = OFF_P1 + PLANE_BYTES / 8;  // column exponents (NCOL ints), flags
    static constexpr int OFF_BAR = OFF_E + NCOL / 2 + 4;  // counters: front barrier, back barrier, tiles digitised, violation stamp
    static constexpr int OFF_MU = OFF_BAR + 4;  // the mean (DP doubles, zero past d): re-read by the staging of every tile
    static constexpr int OFF_K = OFF_MU + (CLDS ? 0 : DP);  // model scalars: sigma^2, 1 / sigma^2, ln sigma (re-read per tile)
    static constexpr int OFF_EB = OFF_K + 4;              // rounding bounds of the cut, per column (wguard_kernel)
    static constexpr int OFF_XT = OFF_EB + NCOL;          // by-products of the solve (quad, |z|^2, det M) per sample, two tile parities
    static constexpr int OFF_CL = OFF_XT + 2 * B * 4;     // (layout B) C^T operands of the first NCL column groups: [K-half][step][group][32]
    static constexpr int LDS_DOUBLES = OFF_CL + 2 * 16 * NCL * 32;
    // where the second K-half's partial of b[a] waits for the solver (layout B): the row's free slots at that time -- what is left of
    // the b area behind the K columns of the first half, then the G part's unused packed columns
    static constexpr int p1slot(int a) { return a < PADC - K ? 16 * NTP + K + a : KP + (a - (PADC - K)); }
    static_assert(!CLDS || K - (PADC - K) <= 16 * NTP - KP, "free slots of a row hold the second partial of b");
    static_assert(NCOL / 2 <= 64, "one back wave digitises NCOL / 2 (column, chunk) items");
    static_assert(LDS_DOUBLES * 8 <= 160 * 1024, "LDS budget");
};

// Barrier among the four waves of one role on a monotonic LDS counter.  A wave's LDS operations execute in order, so
// its add follows its stores; the others read only after seeing the count.
__device__ __forceinline__ void role_barrier(unsigned *ctr, unsigned &target, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    target += 4;
    if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (;;) {
        const unsigned seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if ((int)(seen - target) >= 0) break;
        E9_BARRIER_SLEEP
    }
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void wait_counter(const unsigned *ctr, unsigned need) {
#if defined(E9_ONLY_FRONT) || defined(E9_ONLY_BACK)  // timing experiments with one role absent: nothing to wait for
    return;
#endif
    for (;;) {
        const unsigned seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if ((int)(seen - need) >= 0) break;
        __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}


// WEIGHTED: the dataset carries sample weights (PassArgs::w).  A template parameter because the weighted pass takes a
// logarithm per sample and tile (the unweighted one multiplies the determinants up and takes one per lane per kernel):
// the constants of that logarithm were what the register allocator spilled in the un-weighted hot kernel.
template <int K, bool GATHER, bool WEIGHTED>
__global__ __launch_bounds__(512) void em9_kernel(PassArgs p) {
    using cfg = Cfg9<K>;
    constexpr int KP = cfg::KP, NTP = cfg::NTP, B = cfg::B, XS = cfg::XS, CS = cfg::CS, GS = cfg::GS, BS = cfg::BS,
                  WS = cfg::WS, NC = cfg::NC, NCT = cfg::NCT, NCOL = cfg::NCOL, NCP = cfg::NCP;
    constexpr int NF = 4;              // waves per role
    constexpr int RPW = B / NF;        // rows staged per front wave
    constexpr int DPS = cfg::DP / 2;   // dims per K-split of b = X~ C
    constexpr int STEPS = DPS / 4;
    constexpr int RT = 16 / NF;        // 16-dim row tiles per wave in P4
    constexpr int DW = cfg::DP / NF;   // dims owned by a wave in P4
    constexpr int QW = E9_QW;
    static_assert(NTP <= NF, "int8 Gram: one front wave per packed-column tile");
    static_assert(QS == 8, "digit grouping assumes 8 slices");
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *Xs = sm + cfg::OFF_X;
    double *Gs = sm + cfg::OFF_G;
    double *B1 = sm + cfg::OFF_B1;
    double *Ws = sm + cfg::OFF_W;
    unsigned long long *Ms = reinterpret_cast<unsigned long long *>(sm + cfg::OFF_M);
    unsigned *Mb = reinterpret_cast<unsigned *>(sm + cfg::OFF_MB);
    double *xxs = sm + cfg::OFF_S;
    double *scl = sm + cfg::OFF_L;
    int *Ex = reinterpret_cast<int *>(sm + cfg::OFF_E);
    unsigned *ctr = reinterpret_cast<unsigned *>(sm + cfg::OFF_BAR);
    unsigned *fbar = ctr, *bbar = ctr + 1, *digdone = ctr + 2, *vstamp = ctr + 3, *wready = ctr + 4, *itdone = ctr + 5, *cbar = ctr + 6;

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

    constexpr bool PAIRS = K >= 2;
    constexpr int SQW = PAIRS ? 2 * B : B;  // sq slots per front wave
    constexpr bool CLDS = cfg::CLDS;
    constexpr int L_DEV = CLDS ? 0 : NF * SQW, L_LLK = L_DEV + B, L_W = L_DEV + 2 * B, L_NE = L_DEV + 3 * B, L_PM = L_DEV + 4 * B,
                  L_PX = L_DEV + 5 * B;
    static_assert(L_DEV + 6 * B <= (CLDS ? 6 : 14) * B, "scalar slots");
    for (int idx = tid; idx < L_DEV + 6 * B; idx += 512) scl[idx] = (idx >= L_PM && idx < L_PX) ? 1.0 : 0.0;
    for (int idx = tid; idx < cfg::DP * 4; idx += 512) Mb[idx] = 0u;
    if constexpr (!CLDS) {
        for (int idx = tid; idx < cfg::DP; idx += 512) sm[cfg::OFF_MU + idx] = idx < d ? mMean[idx] : 0.0;
    } else {  // the resident part of C^T in operand order: [K-half][step][group < NCL][32] of PassArgs::cpb's [..][group < NCGB][32]
        constexpr int NCL = cfg::NCL, NCGB = cfg::NCGB;
        for (int idx = tid; idx < 2 * 16 * NCL * 32; idx += 512) {
            const int e = idx & 31, blk = idx >> 5, c = blk % NCL, hq = blk / NCL;
            sm[cfg::OFF_CL + idx] = p.cpb[(hq * NCGB + c) * 32 + e];
        }
    }
#ifdef E9_ONLY_BACK
    for (int idx = tid; idx < 2 * B * GS; idx += 512) Gs[idx] = 1.0;  // (something finite for the back role to cut)
#endif
    if (tid < 8) ctr[tid] = 0u;
    if (tid < 4) Ex[NCOL + tid] = 0;  // (heavy-row words of the back role's cold path)
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
    long long tfine[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tfl = clock64();
#endif
    __syncthreads();
#ifdef E9_ONLY_FRONT
    if (!front) return;
#endif
#ifdef E9_ONLY_BACK
    if (front) return;
#endif

    if (!front) {
        // =========================================================== back role: P4b on the int8 MFMA
        typedef double acc_t;
        acc_t accM[RT][NCT][4];
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int t = 0; t < NCT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) accM[r][t][q] = (acc_t)0;
        // (measured: the back role at a higher priority than the front costs 5 % -- it is off the front's critical path
        //  and only has to fill the gaps)
        if (E9_BACK_PRIO) __builtin_amdgcn_s_setprio(E9_BACK_PRIO);
        unsigned bbar_target = 0u;
        unsigned attempt = 0u;   // digitise attempts so far (the violation stamp of the current one)
        int pending = 0;         // 1: the previous tile's planes wait in P0 for their partner
        int have_scale = 0, flushed = 0, groups = 0;
        int n_rescale = 0, n_flush = 0;  // (wave-uniform: diagnostic counters)
        int hbudget = E9_HEAVY_BUDGET;   // tiles of the current flush window that may still send rows round the fixed-point form
        // Rounding bound of the cut, per column this lane covers (c = 16 t + l15), for wguard_kernel (ppca_kernels.hip): every
        // flush window adds 4 sqrt(rows of the window) quanta 2^(E_c - F) of the exponents it was cut under.
        double *ebs = sm + cfg::OFF_EB;  // (in LDS: touched at flushes only, by the role's first wave)
        int rows_win = 0;
        unsigned char *smb = reinterpret_cast<unsigned char *>(sm);
        constexpr int P0_BYTES = cfg::OFF_P0 * 8, PG_BYTES = cfg::OFF_P1 * 8;
        StatsLayout L(d, K);
        double *out = p.part + (int64_t)blockIdx.x * L.len;

        // [wP | wz | w] column c of W row -> its slot
        const double *Wcur = Ws;  // the W rows of the tile being cut (buffer of its parity)
        auto wsrc = [&](int c) { return c < KP ? c : 16 * NTP + (c - KP); };

        // ---- digit planes of the current tile's rows under the exponents Ex; returns (wave-uniform) whether an entry
        // of a live column did not fit.  Item = (column c, 16-sample chunk): NCOL / 2 items per wave.
        // heavy_tag = true (cold path, round 6): the samples of the mask `heavy` are cut as zeros -- they go round the fixed-point form
        // (heavy_add below)
        auto digitise = [&](int lane, int dst_bytes, auto heavy_tag, unsigned heavy) -> bool {
            constexpr bool HEAVY = decltype(heavy_tag)::value;
            asm volatile("" : "+v"(lane));  // (addresses recomputed here, not hoisted and parked across the other phases)
            const bool active = lane < NCOL / 2;
            const int it = (NCOL / 2) * wave + (active ? lane : 0);
            const int c = it >> 1, chunk = it & 1;
            const unsigned hmine = HEAVY ? (heavy >> (16 * chunk)) & 0xFFFFu : 0u;
            const bool cvalid = c < NC;
            const int src = cvalid ? wsrc(c) : 0;
            const int E = Ex[c];
            const bool poisoned = E > 5000;
            const double qsc = __hiloint2double((1023 + E9_F - (poisoned ? 0 : E)) << 20, 0);
            const double magic = __hiloint2double(0x43388080, (int)0x80808080);
            unsigned bad = 0u;
            unsigned pl[QW][4];
            const double *wsrcp = Wcur + (16 * chunk) * WS + src;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                // four samples at a time (the live set stays small next to the 160 accumulator registers)
                unsigned wlo[4], whi[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // mantissa of w 2^(F - E) + magic = 2^51 + 0x808080808080 + I: xor with the constant's own bits leaves
                    // bytes 0..5 = the balanced digits and bits 48..51 = the top digit (4-bit two's complement)
                    const double wv = wsrcp[(4 * g4 + j) * WS];
                    const bool take = HEAVY ? cvalid && !((hmine >> (4 * g4 + j)) & 1u) : cvalid;
                    const double v = __builtin_fma(take ? wv : 0.0, qsc, magic);
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
            if (active && (cvalid || NCP == NCOL)) {
                unsigned char *wq = smb + dst_bytes;
#pragma unroll
                for (int sl = 0; sl < QW; ++sl)
                    *reinterpret_cast<i4_t *>(wq + ((sl * 2 + chunk) * NCP + c) * 16) =
                        i4_t{(int)pl[sl][0], (int)pl[sl][1], (int)pl[sl][2], (int)pl[sl][3]};
            }
            const bool mine = active && cvalid && !poisoned && (bad >> 20) != 0u;
            return __builtin_amdgcn_ballot_w64(mine) != 0ull;
        };

        // ---- new exponents from the current tile's column maxima (cold path)
        // robust (round 6: the workgroup's FIRST scale, when rows may go round the form): per column the largest magnitude that at
        // least `heavy_max` of the tile's 32 entries exceed -- the (heavy_max + 1)-th largest -- instead of the largest: outlier rows in
        // the scale-setting tile then do not set the scale; they do not fit it and go round like any later one.  (Clean rows: the
        // entries above it are within the form's 2^HEAD of headroom.  Fewer than heavy_max + 1 non-zero entries: the plain maximum.)
        auto rescale = [&](int lane, bool robust) {
            asm volatile("" : "+v"(lane));
            const bool active = lane < NCOL / 2;
            const int it = (NCOL / 2) * wave + (active ? lane : 0);
            const int c = it >> 1, chunk = it & 1;
            const bool cvalid = c < NC;
            const int src = cvalid ? wsrc(c) : 0;
            double m = 0.0;
            bool fin = true;
            if (!robust) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const double av = __builtin_fabs(cvalid ? Wcur[(16 * chunk + j) * WS + src] : 0.0);
                    fin = fin && (av < __builtin_inf());
                    m = __builtin_fmax(m, av);
                }
            } else {  // (cold: once per workgroup)
                const int hm = p.heavy_max;
                double cand = 0.0, mx = 0.0;
                for (int j = 0; j < 16; ++j) {
                    const double av = __builtin_fabs(cvalid ? Wcur[(16 * chunk + j) * WS + src] : 0.0);
                    fin = fin && (av < __builtin_inf());
                    mx = __builtin_fmax(mx, av);
                    int above = 0;  // entries of the column (both chunks) strictly above this one
                    for (int i = 0; i < 32; ++i) above += __builtin_fabs(cvalid ? Wcur[i * WS + src] : 0.0) > av ? 1 : 0;
                    if (above >= hm) cand = __builtin_fmax(cand, av);
                }
                cand = __builtin_fmax(cand, dpp_f64<0xB1, 0xF>(cand));
                mx = __builtin_fmax(mx, dpp_f64<0xB1, 0xF>(mx));
                m = cand > 0.0 ? cand : mx;
            }
            m = __builtin_fmax(m, dpp_f64<0xB1, 0xF>(m));  // the other chunk of the column sits in the neighbouring lane
            const int finw = __builtin_amdgcn_update_dpp(0, fin ? 1 : 0, 0xB1, 0xF, 0xF, true);
            fin = fin && finw != 0;
            const int Eold = have_scale ? Ex[c] : E9_EMIN;
            int Enew = Eold;
            if (m > 0.0) {
                int e = __builtin_amdgcn_frexp_exp(m) + E9_HEAD;  // |w| < 2^(e - HEAD)
                e = e < E9_EMIN ? E9_EMIN : e;
                Enew = e > Eold ? e : Eold;
            }
            if (!fin || Enew > E9_EMAX) Enew = E9_POISON;
            if (Eold > 5000) Enew = Eold;  // (a poisoned column stays poisoned)
            if (active && chunk == 0) Ex[c] = Enew;
        };

        // ---- (round 6, cold path) rows far above the scale go ROUND the fixed-point form.  Until round 5 a tile with an entry that did
        // not fit raised its columns' exponents for the rest of the workgroup's run: an outlier row -- a sample 1e6 x its neighbours,
        // a heavy sample weight -- had every later row cut far below its own resolution, the W-side guard then sent the workgroup's
        // whole slice (39 k rows at N = 10 M) to the fp64 engine: 10 such rows per million halved the rate.  Now the rows of the tile
        // that do not fit are found (heavy_bits: per (column, 16-sample chunk) item the samples whose entry leaves the form, OR-ed
        // over the items in LDS); if they are few and finite the tile is cut again WITHOUT them under the old exponents, and each of
        // them is added to the fp64 accumulators directly -- w P_i, w z_i, w x 2^(F - E_c), an exact scaling, into the accumulators of
        // the dimensions the sample observes (heavy_add: the lane's 80 accumulator entries, as emit() maps them).  That is what the
        // reference's fp64 sums do with such a row (ppca_model.rs:297-306); the dimensions MASKED in it keep their full resolution.
        // Many such rows in one tile are a change of scale: the exponents rise as before.
        int *hvw = Ex + NCOL;  // [0] the samples of the tile that do not fit, [1] "cannot go round"
        auto heavy_bits = [&](int lane) {
            asm volatile("" : "+v"(lane));
            const bool active = lane < NCOL / 2;
            const int it = (NCOL / 2) * wave + (active ? lane : 0);
            const int c = it >> 1, chunk = it & 1;
            const bool cvalid = c < NC;
            const int src = cvalid ? wsrc(c) : 0;
            const int E = Ex[c];
            const bool live = active && cvalid && E <= 5000;
            const double qsc = __hiloint2double((1023 + E9_F - (live ? E : 0)) << 20, 0);
            const double magic = __hiloint2double(0x43388080, (int)0x80808080);
            unsigned bits = 0u, nogo = 0u;
            for (int j = 0; j < 16; ++j) {
                const double wv = live ? Wcur[(16 * chunk + j) * WS + src] : 0.0;
                const double v = __builtin_fma(wv, qsc, magic);
                const unsigned hi = (unsigned)__double2hiint(v) ^ 0x43388080u;
                if ((hi >> 20) != 0u) {
                    bits |= 1u << (16 * chunk + j);
                    // not finite, or so far above the column's scale that w 2^(F - E) would overflow: the exponents have to rise
                    if (!(__builtin_fabs(wv) < __builtin_inf()) || __builtin_amdgcn_frexp_exp(wv) + E9_F - E > 1000) nogo = 1u;
                }
            }
            if (bits) __hip_atomic_fetch_or(reinterpret_cast<unsigned *>(hvw), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (nogo) __hip_atomic_fetch_or(reinterpret_cast<unsigned *>(hvw + 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        auto heavy_add = [&](int lane, unsigned hv, int slot) {
            asm volatile("" : "+v"(lane));
            const int l15 = lane & 15, l4 = lane >> 4;
            unsigned mw[RT][4];
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) mw[r][q] = Mb[(DW * wave + 16 * r + 4 * l4 + q) * 4 + slot];
            double sc[NCT];
            int src[NCT];
#pragma unroll
            for (int t = 0; t < NCT; ++t) {
                const int c = 16 * t + l15;
                const int E = Ex[c];
                sc[t] = (c < NC && E <= 5000) ? __hiloint2double((1023 + E9_F - E) << 20, 0) : 0.0;
                src[t] = c < NC ? wsrc(c) : 0;
            }
            for (unsigned rest = hv; rest != 0u; rest &= rest - 1u) {
                const int i = __builtin_ctz(rest);                 // sample of the tile (wave-uniform)
                const int bitpos = 8 * (i >> 3) + 7 - (i & 7);     // its bit in a dimension's sample mask (byte = staging wave, row r at bit 7 - r)
                double wv[NCT];
#pragma unroll
                for (int t = 0; t < NCT; ++t) wv[t] = Wcur[i * WS + src[t]] * sc[t];
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const bool obs = (mw[r][q] >> bitpos) & 1u;
#pragma unroll
                        for (int t = 0; t < NCT; ++t) accM[r][t][q] += obs ? wv[t] : 0.0;
                    }
            }
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
                if (wave == 0 && l4 == 0 && have_scale && E <= 5000) ebs[c] += win * __hiloint2double((1023 + E - E9_F) << 20, 0);
                const double fsc = __hiloint2double(E > 5000 ? 0x7FF80000 : (1023 + E - E9_F) << 20, 0);
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
            const unsigned char *wq = smb + (lh == 0 ? P0_BYTES : PG_BYTES) + ((l4 & 1) * NCP + l15) * 16;
            constexpr int PSTRIDE = 2 * NCP * 16;  // bytes between digit planes
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
            const int slot_cur = rel & 3, slot_prev = (rel + 3) & 3;
            Wcur = Ws + (rel & 1) * B * WS;  // the buffer of the tile's parity
            // front: P3(tile) done in all four waves -> the tile's W rows are final.  A counter, not a workgroup barrier: the
            // front never waits for this role here (the contraction of a group is two tiles' work on every second
            // tile: behind a barrier the front stood ~2.6 k cycles on those tiles)
            wait_counter(wready, 4u * (unsigned)(rel + 1));
#pragma unroll 1
            for (;;) {  // normally one trip
                ++attempt;
                const bool bad = !have_scale || digitise(lane, pending ? PG_BYTES : P0_BYTES, std::false_type{}, 0u);
                if (bad && lane == 0) __hip_atomic_store(vstamp, attempt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                role_barrier(bbar, bbar_target, lane_entry);
                bool viol = __builtin_amdgcn_readfirstlane(__hip_atomic_load(vstamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == attempt;
                bool hv_dirty = false;  // the two words are cleared behind the NEXT barrier of the role (every wave has read them by then)
                if (viol && have_scale && p.heavy_max > 0) {  // (cold) a few rows above the scale: round the fixed-point form
                    heavy_bits(lane);
                    role_barrier(bbar, bbar_target, lane_entry);
                    const unsigned hv = (unsigned)__builtin_amdgcn_readfirstlane(__hip_atomic_load(hvw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    const unsigned nogo = (unsigned)__builtin_amdgcn_readfirstlane(__hip_atomic_load(hvw + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    hv_dirty = true;
                    // (a window in which tile after tile has a few such rows -- weights spread over many orders, as in the first
                    //  iterations of a mixture -- is better served by ONE rise of the exponents than by a second cut of every tile)
                    if (hv != 0u && nogo == 0u && __builtin_popcount(hv) <= p.heavy_max && hbudget > 0) {
                        --hbudget;
                        (void)digitise(lane, pending ? PG_BYTES : P0_BYTES, std::true_type{}, hv);
                        heavy_add(lane, hv, slot_cur);
                        role_barrier(bbar, bbar_target, lane_entry);  // (every wave's planes of the tile are cut again)
                        if (wave == 0 && lane_entry < 2) hvw[lane_entry] = 0;
                        hv_dirty = false;
                        viol = false;
                    }
                }
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
                }
                pending = (!viol && !con) ? 1 : 0;
                if (viol ? have_scale != 0 : groups >= E9_FLUSH_GROUPS) {
                    emit(lane, flushed != 0, true);
                    flushed = 1;
                    groups = 0;
                    n_flush += viol ? 0 : 1;
                    if (!viol) hbudget = E9_HEAVY_BUDGET;
                }
                if (!viol) break;
                n_rescale += have_scale;
                role_barrier(bbar, bbar_target, lane_entry);  // every wave has read the old exponents
                if (hv_dirty && wave == 0 && lane_entry < 2) hvw[lane_entry] = 0;
                rescale(lane, !have_scale && p.heavy_max > 0);
                have_scale = 1;
                role_barrier(bbar, bbar_target, lane_entry);
            }
            // this wave has read the sample masks and planes of its iteration for the last time (the front waits for this
            // before it stages the tile that reuses the oldest mask slot)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane_entry == 0) __hip_atomic_fetch_add(itdone, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        emit(lane_entry, flushed != 0, false);
        if (p.errb && wave == 0 && lane_entry < 16) {
#pragma unroll
            for (int t = 0; t < NCT; ++t) p.errb[(int64_t)blockIdx.x * W_GUARD_NCOL + 16 * t + lane_entry] = ebs[16 * t + lane_entry];
        }
        if (tid == 256) {
            if (n_rescale) atomicAdd(&e9_counters[0], (unsigned long long)n_rescale);
            if (n_flush) atomicAdd(&e9_counters[1], (unsigned long long)n_flush);
            atomicMax(&e9_counters[2], (unsigned long long)(tile_end > tile_begin ? tile_end - tile_begin : 0));
            if (blockIdx.x == 0) atomicAdd(&e9_counters[3], 1ull);
        }
        return;
    }

    // =============================================================== front role
    if (E9_FRONT_PRIO) __builtin_amdgcn_s_setprio(E9_FRONT_PRIO);
    unsigned fbar_target = 0u;
    // cross / sumx accumulators of P4a on v_mfma_f64_4x4x4 (four independent 4 x 4 x 4 blocks per instruction: block b = lane bits
    // 2-3; A[i][k] in lane 16 k + 4 b + i, B[k][j] in lane 16 k + 4 b + j, D[i][j] in lane 16 i + 4 b + j -- probed on the part,
    // tools/mfma_peak).  That instruction issues every 17.3 cycles for 512 flop (72.8 TFLOP/s, profiles/r04/mfma_peak.txt) where
    // v_mfma_f64_16x16x4 takes 105 for 2 048 (48): the same x~ operand registers, the K + 1 columns as groups of 4.
    // (b = X~ C of P2 the same way measured SLOWER, 87.5 against 100.7 it/s: its B operands come from global memory -- three
    //  loads per k-step instead of one.)
    constexpr int NCG = (K + 1 + 3) / 4;
    double accX[RT][NCG];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < NCG; ++c) accX[r][c] = 0.0;
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
    // (round 6) The row list and the weights of a gathered / weighted pass used to be read where they are used -- one scalar load per
    // row in front of its row request inside P4a's MFMA loop, one per row inside the staging -- and a scalar load's wait is
    // lgkmcnt(0): it also drains the LDS operand reads / x~ stores in flight around it.  The gathered weighted pass of a mixture
    // component ran at 1.5 us per 1 000 rows against 0.95 for the plain pass.  Now lane r < RPW of one VECTOR load holds the physical
    // row (gidx) and the weight (gw) of the wave's row r of the tile requested / staged next, issued a whole trip before its use.
    int gidx = 0;
    double gw = 0.0;
    auto fetch_meta = [&](int64_t tile) {
        const int rel = (int)(tile - tile_begin) * B + wave * RPW + (lane_entry & (RPW - 1));
        const int rc = rel < nrel ? rel : nrel - 1;  // (nrel >= 1 wherever this is called)
        if constexpr (GATHER) {
            if (rows_wg) gidx = rows_wg[rc];
        }
        if constexpr (WEIGHTED) gw = p.w[tile_begin * B + rc];
    };
    auto load_row = [&](const __amdgpu_buffer_rsrc_t &trs, int64_t tile, int r) {
        typedef unsigned u4_t __attribute__((ext_vector_type(4)));
        if constexpr (GATHER) {
            if (rows_wg) {
                const int rel = (int)(tile - tile_begin) * B + wave * RPW + r;
                const int rc = rel < nrel ? rel : nrel - 1;
                const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<double *>(p.X + (int64_t)__builtin_amdgcn_readlane(gidx, r) * p.ldx), 0, rc < 0 ? 0 : rowbytes, 0x00020000);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const u4_t v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, lane_entry * 16, 1024 * h, E9_X_AUX);
                    xr[r][2 * h] = __longlong_as_double(((long long)v[1] << 32) | v[0]);
                    xr[r][2 * h + 1] = __longlong_as_double(((long long)v[3] << 32) | v[2]);
                }
                return;
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const u4_t v = __builtin_amdgcn_raw_buffer_load_b128(trs, lane_entry * 16, (wave * RPW + r) * rowbytes + 1024 * h, E9_X_AUX);
            xr[r][2 * h] = __longlong_as_double(((long long)v[1] << 32) | v[0]);
            xr[r][2 * h + 1] = __longlong_as_double(((long long)v[3] << 32) | v[2]);
        }
    };
    const __amdgpu_buffer_rsrc_t qrsrc = __builtin_amdgcn_make_buffer_rsrc(p.qtab, 0, (int)qtab_bytes<K>(), 0x00020000);
    const bool gram_wave = NTP >= NF || wave < NTP;
    i4_t qbA[2][4];
#if E9_QB_EARLY
    i4_t qbB[2][4];  // digits {5,4} requested with {7,6} during the previous tile's P4a (round 5): requested at the top of P2 they
                     // were ~600 cycles old when the second digit pair wanted them -- an L2 round trip is longer
#endif
    auto load_pair = [&](i4_t(&dst)[2][4], int sl0) {
        int qbase = wave * QS * 4 * 1024;
        asm volatile("" : "+s"(qbase));
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                typedef unsigned u4_t __attribute__((ext_vector_type(4)));
                const u4_t v = __builtin_amdgcn_raw_buffer_load_b128(qrsrc, lane_entry * 16 + kc * 1024, qbase + (sl0 + u) * 4096, E9_Q_AUX);
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
    double sq_run = 0.0;  // (layout B) this lane's running share of sum_i w_i tr(C_o Sigma_i C_o^T); layout A keeps it in LDS
    // (layout B) the lane's four means, requested from L1 / L2 with the next tile's rows (no LDS copy of the mean)
    double muq[4] = {0.0, 0.0, 0.0, 0.0};
    auto load_mu = [&]() {
        const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(mMean), 0, d * (int)sizeof(double), 0x00020000);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            typedef unsigned u2_t __attribute__((ext_vector_type(2)));
            const u2_t v2 = __builtin_amdgcn_raw_buffer_load_b64(mrs, lane_entry * 16 + 8 * (q & 1), 1024 * (q >> 1), 0);  // (past d: zeros)
            muq[q] = __longlong_as_double(((long long)v2[1] << 32) | v2[0]);
        }
    };
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
        double wr = 1.0;
        if constexpr (WEIGHTED) {  // (lane r of gw: fetch_meta)
            const long long gb = __double_as_longlong(gw);
            const int lo = __builtin_amdgcn_readlane((int)gb, r), hi = __builtin_amdgcn_readlane((int)(gb >> 32), r);
            wr = __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
        }
        wr = mine ? wr : 0.0;
        xx_run += wr * pc_xx;
    };
    // staging lane map: lane l holds dims 128 h + 2 l + e (element q = 2 h + e) of a row.  The lane's four means and
    // limits (observed <=> |x| < lim: +inf for a real dimension -- the finite test of dataset.rs:19-22 --, -1 for the
    // padding past d) are rebuilt per tile from the LDS copy of the mean: as loop invariants they sat in 16 registers
    // across the solver and were what the register allocator spilled.
    auto stage_tile = [&](int64_t t, int lane) {
        {
            typedef double d2_t __attribute__((ext_vector_type(2)));
            if constexpr (CLDS) {
                mu[0] = muq[0]; mu[1] = muq[1]; mu[2] = muq[2]; mu[3] = muq[3];
            } else {
                const d2_t m0 = *reinterpret_cast<const d2_t *>(sm + cfg::OFF_MU + 2 * lane);
                const d2_t m1 = *reinterpret_cast<const d2_t *>(sm + cfg::OFF_MU + 128 + 2 * lane);
                mu[0] = m0[0]; mu[1] = m0[1]; mu[2] = m1[0]; mu[3] = m1[1];
            }
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
        const int slot = rel & 3;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            mbb[(128 * (q >> 1) + 2 * lane + (q & 1)) * 16 + 4 * slot + wave] = (unsigned char)st_mb[q];
    };

    if (tile_begin < tile_end) {
        const __amdgpu_buffer_rsrc_t trs = tile_rsrc(tile_begin);
        if constexpr (GATHER || WEIGHTED) fetch_meta(tile_begin);
#pragma unroll
        for (int r = 0; r < RPW; ++r) load_row(trs, tile_begin, r);
        if constexpr (CLDS) load_mu();
        load_pair(qbA, 6);
#if E9_QB_EARLY
        load_pair(qbB, 4);
#endif
        stage_tile(tile_begin, lane_entry);
        if constexpr (GATHER || WEIGHTED) fetch_meta(tile_begin + 1);
    }
    role_barrier(fbar, fbar_target, lane_entry);

    // z of a solved sample lives in the row's free slots until the columns of its tile are formed: the unused columns of
    // the b partial, the pad, then the unused packed-column slots behind K'
    constexpr int LPER = (16 * NTP) / 32;  // factor entries per buffer row when the factor is kept entry-major (below)
    constexpr bool LMAJ = (16 * NTP) % 32 == 0 && B == 32 && (KP + (LPER > 0 ? LPER : 1) - 1) / (LPER > 0 ? LPER : 1) + 2 <= B;
    constexpr int LROWS = LMAJ ? (KP + LPER - 1) / LPER : 0;  // buffer rows the entry-major factor takes
    // z (a, sample i): first the unused columns of the row's b partial and the pad; what does not fit goes behind K' in the row
    // (row-major factor) or entry-major behind the factor's rows (entry-major factor: the G part of every row is taken)
    constexpr int PADC = cfg::PADC, ZIN = PADC - K - 1;  // z slots behind [w z | w] in the row's b area + pad (layout A: 17 - K)
    auto zslot = [](int a, int i) {
        if (a < ZIN) return i * GS + 16 * NTP + K + 1 + a;
        a -= ZIN;
        if (LMAJ) return (LROWS + a / (LPER > 0 ? LPER : 1)) * GS + (a % (LPER > 0 ? LPER : 1)) * 32 + i;
        return i * GS + KP + a;
    };
    static_assert(!LMAJ || LROWS + (K - ZIN > 0 ? (K - ZIN + LPER - 1) / (LPER > 0 ? LPER : 1) : 0) <= B, "entry-major z rows");
    static_assert(ZIN >= 0 && (LMAJ || ZIN + (16 * NTP - KP) >= K), "free slots of a row hold z");

    // Where entry e of sample i's factor waits for the column waves: entry-major over the G part of the buffer when that part is
    // a whole number of 32-double blocks wide and has rows enough (k = 6, 7, 10), else in the sample's own row.
    auto lslot = [](int e, int i) { return LMAJ ? (e / (LPER > 0 ? LPER : 1)) * GS + (e % (LPER > 0 ? LPER : 1)) * 32 + i : i * GS + e; };

    // One trip per tile plus ONE more: trip `rel` solves tile rel (one wave) and forms the columns of tile rel - 1 (the others).
    for (int64_t tile = tile_begin; tile <= tile_end; ++tile) {
        int lane = lane_entry;
        asm volatile("" : "+v"(lane));
        const int l15 = lane & 15, l4 = lane >> 4;
        const int colb = (l15 < K) ? l15 : K;
        const int rel = (int)(tile - tile_begin);
        const bool cur = tile < tile_end;  // (wave-uniform) there is a tile to stage / contract / solve in this trip
        const unsigned long long *Msc = Ms + (rel & 1) * 4 * B;
        double *Gcur = Gs + (rel & 1) * B * GS;         // [G | b] -> factor, z, [wz | w] of tile rel
        double *Gprev = Gs + ((rel + 1) & 1) * B * GS;  // factor, z of tile rel - 1 -> its W rows
        // ------------------------------------------------------------ P2: [G | b] of the tile
        if (cur) {
            const int rt = wave & 1, kq = wave >> 1;
#if E9_B444
            constexpr int NCGB = (K + 3) / 4;  // column groups of four
            double bsum[NCGB];
#else
            const int si = 16 * rt + l15;
            d4_t accb = d4_t{0, 0, 0, 0};
            const double *xrow = Xs + si * XS + DPS * kq + l4;
#endif
            // the count of tiles the back role has cut, requested HERE and looked at where [G | b] is stored: by then it is
            // almost always enough, and the poll (an LDS round trip behind everything this wave has queued: ~0.7 k cycles per
            // tile in the phase table) is skipped
            const unsigned dd_early = __hip_atomic_load(digdone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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
#if E9_B444
            // (operand pointers of the b product; its first LAC steps of C are requested before the Gram's digit pairs so that
            //  their L2 latency passes under the integer MFMAs)
                constexpr int NQ = DPS / 8;  // steps of 8 dimensions (two blocks of four)
                constexpr int LAC = E9_LAC, LAX = 2;  // look-ahead in steps: C (L2 latency), x~ (LDS)
                const int j4 = lane & 3, sb = (lane >> 2) & 1, kb = (lane >> 3) & 1;
                const double *xrow = Xs + (16 * rt + 4 * sb + j4) * XS + DPS * kq + 16 * kb + l4;
                const __amdgpu_buffer_rsrc_t crs =
                    __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(p.cpb), 0, CPB_DOUBLES * (int)sizeof(double), 0x00020000);
                const int cvo = ((2 * l4 + kb) * 4 + j4) * 8;
                // (soffset: one SGPR per 4 KB of the wave's half of the table, the rest in the instruction's 12-bit immediate -- left to
                //  the compiler every load had its own s_add)
                constexpr int NSO = (NQ * NCGB * 256 + 4095) / 4096;
                int cso[NSO];
#pragma unroll
                for (int u = 0; u < NSO; ++u) {
                    cso[u] = kq * (NQ * NCGB * 256) + 4096 * u;
                    asm volatile("" : "+s"(cso[u]));
                }
                double accb[NCGB][2];
#pragma unroll
                for (int c = 0; c < NCGB; ++c) accb[c][0] = accb[c][1] = 0.0;
                double cb[LAC + 1][NCGB], xb[LAX + 1][2];
                constexpr int NCLc = cfg::NCL;
                const double *clp = sm + cfg::OFF_CL + kq * (NQ * NCLc * 32) + (2 * l4 + kb) * 4 + j4;
                auto cload_lds = [&](auto q_tag) {  // (layout B) the resident column groups, requested with the x~ operands
                    constexpr int q = decltype(q_tag)::value;
#pragma unroll
                    for (int c = 0; c < NCLc; ++c) cb[q % (LAC + 1)][c] = clp[(q * NCLc + c) * 32];
                };
                auto cload = [&](auto q_tag) {
                    constexpr int q = decltype(q_tag)::value;
#pragma unroll
                    for (int c = NCLc; c < NCGB; ++c) {
                        typedef unsigned u2_t __attribute__((ext_vector_type(2)));
                        const int off = (q * NCGB + c) * 256;  // (a constant once the loop is unrolled)
#ifdef E9_EXP_CLDS  // (timing experiment, results wrong: what the loop would cost with its C operands in LDS)
                        cb[q % (LAC + 1)][c] = Xs[(cvo + off) / 8 + kq * 4096];
                        continue;
#endif
                        const u2_t v2 = __builtin_amdgcn_raw_buffer_load_b64(crs, cvo + (off & 4095), cso[off >> 12], 0);
                        cb[q % (LAC + 1)][c] = __longlong_as_double(((long long)v2[1] << 32) | v2[0]);
                    }
                };
                auto xload = [&](auto q_tag) {
                    constexpr int q = decltype(q_tag)::value;
#pragma unroll
                    for (int g = 0; g < 2; ++g) xb[q % (LAX + 1)][g] = xrow[8 * g * XS + 32 * (q >> 2) + 4 * (q & 3)];
                };
#endif
            double qs = 0.0;
#if !E9_QB_EARLY
            i4_t qbB[2][4];
#endif
            {
                unsigned long long mwd[2][4];
#pragma unroll
                for (int rt2 = 0; rt2 < 2; ++rt2)
#pragma unroll
                    for (int kc = 0; kc < 4; ++kc) mwd[rt2][kc] = Msc[(16 * rt2 + l15) * 4 + kc];
                __builtin_amdgcn_sched_barrier(0);
#if !E9_QB_EARLY
                load_pair(qbB, 4);
#endif
                if (gram_wave) qs = p.qscale[16 * wave + l15];
#if E9_B444 && E9_B_EARLYC
                static_for<LAC>([&](auto q_tag) { cload(q_tag); });
#endif
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
            E9_FINE(15)  // "-": mask bytes + digit pairs {7,6}, {5,4}
#if E9_B444
            {
                // b = X~ C on v_mfma_f64_4x4x4 (round 5).  One instruction = four independent 4 x 4 x 4 blocks; block = (kb, sb):
                // two groups of four DIMENSIONS x two groups of four SAMPLES, so one MFMA covers 8 samples x 4 columns x 8
                // dimensions.  A = C^T from the operand-ordered copy PassArgs::cpb (one 256-byte block per step and column
                // group, L1 / L2), B = x~^T from the LDS tile (lane 16 k + 8 kb + 4 sb + j: sample 8 g + 4 sb + j, dimension
                // 32 s + 16 kb + 4 t + k -- the 16 kb keeps the two dimension groups 16 doubles apart: conflict-free with the
                // 258-double row stride).  Per wave and tile 96 MFMAs of 17 cycles where the 16 x 16 x 4 form took 32 of
                // 105-142 (profiles/r04/mfma_peak.txt), the same 32 LDS reads, 48 instead of 32 loads of C, and the two
                // dimension-group partials summed across lane bit 3 at the end.
#if !E9_B_EARLYC
                static_for<LAC>([&](auto q_tag) { cload(q_tag); });
#endif
                static_for<LAX>([&](auto q_tag) {
                    xload(q_tag);
                    cload_lds(q_tag);
                });
                static_for<NQ>([&](auto q_tag) {
                    constexpr int q = decltype(q_tag)::value;
                    if constexpr (q + LAC < NQ) cload(std::integral_constant<int, q + LAC>{});
                    if constexpr (q + LAX < NQ) {
                        xload(std::integral_constant<int, q + LAX>{});
                        cload_lds(std::integral_constant<int, q + LAX>{});
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c = 0; c < NCGB; ++c)
#pragma unroll
                        for (int g = 0; g < 2; ++g)
                            accb[c][g] = __builtin_amdgcn_mfma_f64_4x4x4f64(cb[q % (LAC + 1)][c], xb[q % (LAX + 1)][g], accb[c][g], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                });
                // the two dimension groups: lane bit 3 (row_ror:8 inside a row of 16 lanes); a + b on both sides, bit-identical.
                // D[i][j] of block (kb, sb) sits in lane 16 i + 8 kb + 4 sb + j: after the sum lanes with kb = g keep sample group g,
                // so that lane (l4, l15) holds b[sample 16 rt + l15][column 4 c + l4].
#pragma unroll
                for (int c = 0; c < NCGB; ++c) {
                    const double t0 = accb[c][0] + dpp_f64<0x128, 0xF>(accb[c][0]);
                    const double t1 = accb[c][1] + dpp_f64<0x128, 0xF>(accb[c][1]);
                    bsum[c] = kb ? t1 : t0;
                }
            }
#else
            {
                // the 16 x 16 x 4 form of rounds 3-4 (kept for A/B runs): B operands from the zero-padded copy of C in global
                // memory (L1 / L2), requested two chunks of four k-steps ahead
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
#endif
            E9_FINE(4)   // "P3-load": the b loop
            group(qbA, false);  // digits {3,2}
            group(qbB, false);  // digits {1,0}
            E9_FINE(0)   // "P2": digit pairs {3,2}, {1,0}
            // the buffer of this parity held the W rows of tile rel - 2: wait until the back role has cut them (long done)
            const unsigned need_dd = rel >= 2 ? 4u * (unsigned)(rel - 1) : 0u;
            if ((int)((unsigned)__builtin_amdgcn_readfirstlane(dd_early) - need_dd) < 0) wait_counter(digdone, need_dd);
            E9_FINE(1)
            if (gram_wave) {
#pragma unroll
                for (int rt2 = 0; rt2 < 2; ++rt2)
#pragma unroll
                    for (int r = 0; r < 4; ++r)  // C/D map of the 16x16 integer MFMA: row = 4 (lane >> 4) + reg
                        if (!CLDS || 16 * wave + l15 < KP) Gcur[(16 * rt2 + 4 * l4 + r) * GS + 16 * wave + l15] = v[rt2][r] * qs;  // (sample-major: the solver's loads pair up into 16-byte reads; entry-major measured 2 % slower)
            }
            // the two K-split partials of b are summed by the solver in a fixed order (p0 + p1)
#if E9_B444
#pragma unroll
            for (int c = 0; c < NCGB; ++c) {
                if constexpr (CLDS) {
                    const int a = 4 * c + l4;
                    if (a < K) Gcur[(16 * rt + l15) * GS + (kq == 0 ? 16 * NTP + a : cfg::p1slot(a))] = bsum[c];
                } else {
                    if (kq == 0) Gcur[(16 * rt + l15) * GS + 16 * NTP + 4 * c + l4] = bsum[c];
                    else if (4 * c + l4 < K + 1) B1[(16 * rt + l15) * BS + 4 * c + l4] = bsum[c];
                }
            }
#else
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (kq == 0) Gcur[(16 * rt + l4 + 4 * r) * GS + 16 * NTP + l15] = accb[r];
                else if (l15 < K + 1) B1[(16 * rt + l4 + 4 * r) * BS + l15] = accb[r];
            }
#endif
        }
        E9_FINE(2)
#ifndef E9_EXP_NOB1  // (timing experiment, results wrong: what this barrier costs)
        role_barrier(fbar, fbar_target, lane_entry);
#endif
        // ------------------------------------------------------------ phase beta: solve tile rel | columns of tile rel - 1
        // ONE front wave (rel mod 4) factors the samples of tile rel (lane = sample, lanes 32-63 mirror 0-31), solves for z,
        // leaves factor, z and [wz | w] in the tile's rows and takes the scalars; the other three form the columns of M^-1 of
        // tile rel - 1 from what ITS solver left one trip earlier -- two columns per instruction stream (lane i: column 2p,
        // lane i + 32: column 2p + 1) -- and overwrite the factor with the wP part of the W rows.
        E9_FINE(3)
        const int solver = rel & (NF - 1);
        constexpr int SCAL_UNITS = 107;  // the scalars in the units of pair_owner_eq's column costs (1.07 k against 1.39 k cycles of pair 0)
        if (PAIRS || lane < B) {
            // (the model scalars come back from LDS per tile: as loop invariants they -- and what hipcc derives from them --
            //  were parked in spilled vector registers across the whole tile loop)
            const double s2 = sm[cfg::OFF_K], inv_s2 = sm[cfg::OFF_K + 1], lnsig = sm[cfg::OFF_K + 2];
            const int i = lane & (B - 1);
            const int hi = PAIRS ? lane >> 5 : 0;
            double sc_sq = 0.0;
            if (wave == solver) {
                if (cur) {
                    const int64_t row = tile * B + i;
                    double *g0 = Gcur + i * GS;
                    const double *b1 = B1 + i * BS;
                    const double wgt = (row < n) ? (WEIGHTED ? p.w[row] : 1.0) : 0.0;
                    const int m = __popcll(Msc[i * 4]) + __popcll(Msc[i * 4 + 1]) + __popcll(Msc[i * 4 + 2]) + __popcll(Msc[i * 4 + 3]);
                    Posterior<K> fac;
                    double pm = 1.0;
                    int pe = 0;
                    double z[K], quad = 0.0, zz = 0.0;
#ifdef E9_EXP_NOLOAD  // (timing experiments, results wrong: what each part of the solver's trip costs)
                    fac.load([&](int e) { return (double)(e + 1) * s2; }, s2);
#pragma unroll
                    for (int a = 0; a < K; ++a) z[a] = s2 + (double)a;
#else
                    fac.load([&](int e) { return g0[e]; }, s2);
#pragma unroll
                    for (int a = 0; a < K; ++a) z[a] = g0[16 * NTP + a] + (CLDS ? g0[cfg::p1slot(a)] : b1[a]);
#endif
                    E9_FINE(7)   // "solve" column of the table: the solver's loads
#ifndef E9_EXP_NOFACTOR
                    fac.factor_loaded(pm, pe);
#endif
                    E9_FINE(9)   // "scalars": its factorisation
#ifndef E9_EXP_NOSOLVE
                    fac.solve_loaded(z, quad, zz);
#endif
                    E9_FINE(10)  // "wg-barrier": its substitutions; "factor": its write-back
                    if (hi == 0) {
                        // the factor ENTRY-major over the G part of the tile's buffer (entry e of sample i at row e / 2, column
                        // 32 (e % 2) + i: the 32 lanes of one store are 32 consecutive doubles) -- sample-major, 82 doubles
                        // between the lanes, every store and every load of the three column waves ran four-way bank-conflicted
                        // (the solver's trip took 4.7 k cycles).  Safe: every lane of this wave has read its own [G | b] row
                        // before the first store is issued, and nobody else reads this buffer in this phase.
#ifndef E9_EXP_NOWRITE  // (timing experiment: what the write-back costs)
#pragma unroll
                        for (int e = 0; e < KP; ++e) Gcur[lslot(e, i)] = fac.L[e];
#else
                        Gcur[lslot(0, i)] = fac.L[0] + fac.L[KP - 1];
#endif
#pragma unroll
                        for (int a = 0; a < K; ++a) Gcur[zslot(a, i)] = z[a];
                        // W row = [w P (K') | .. | w z (K) | w | ..]: the [w z | w] part now (P4a of this trip reads it)
#pragma unroll
                        for (int a = 0; a < K; ++a) g0[16 * NTP + a] = wgt * z[a];
                        g0[16 * NTP + K] = wgt;
                        double *xt = sm + cfg::OFF_XT + ((rel & 1) * B + i) * 4;  // for the wave that takes the scalars next trip
                        xt[0] = quad;
                        xt[1] = zz;
                        xt[2] = pm;
                        xt[3] = (double)pe;
                    }
                }
            } else if (rel > 0) {
                const int cw = (wave - solver - 1) & (NF - 1);  // 0 .. 2: this wave's index among the three column waves
                const int64_t row = (tile - 1) * B + i;
                double *wrow = Gprev + i * GS;
                const double wgt = (row < n) ? (WEIGHTED ? p.w[row] : 1.0) : 0.0;
                const unsigned long long *Msp = Ms + ((rel + 1) & 1) * 4 * B;
                const int m = __popcll(Msp[i * 4]) + __popcll(Msp[i * 4 + 1]) + __popcll(Msp[i * 4 + 2]) + __popcll(Msp[i * 4 + 3]);
                Posterior<K> post;
                double z[K];
#pragma unroll
                for (int e = 0; e < KP; ++e) post.L[e] = Gprev[lslot(e, i)];  // (entry-major where the row width allows, see the solver)
#pragma unroll
                for (int a = 0; a < K; ++a) z[a] = Gprev[zslot(a, i)];
                // the wP entries go where the factor is: the three column waves hold it before any of them writes
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane_entry == 0) __hip_atomic_fetch_add(cbar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                wait_counter(cbar, 3u * (unsigned)rel);
                double trpart = 0.0;
                if constexpr (PAIRS) {
#pragma unroll
                    for (int pp = 0; pp < (K + 1) / 2; ++pp) {
                        if (pair_owner_eq(K, pp, NF - 1, SCAL_UNITS) != cw) continue;
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
                        if (column_owner_eq(K, c, NF - 1, SCAL_UNITS) != cw) continue;
                        trpart += post.minv_column(c, [&](int a, int cc, double v) { wrow[tri(a, cc)] = wgt * (z[a] * z[cc] + s2 * v); });
                    }
                }
                // tr(C_o Sigma C_o^T) = <Sigma, G> = s2 (K - s2 tr M^-1)  (:345); all-masked samples are filtered out (:333)
                if (m > 0) sc_sq -= wgt * s2 * s2 * trpart;
                if (cw == NF - 2 && hi == 0) {  // the last column wave (fewest columns) takes the tile's scalars
                    const double *xt = sm + cfg::OFF_XT + (((rel + 1) & 1) * B + i) * 4;
                    const double quad = xt[0], zz = xt[1], pm = xt[2];
                    const int pe = (int)xt[3];
                    double sc_dev = 0.0, sc_llk = 0.0, sc_w = 0.0, sc_ne = 0.0;
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
            }
            if constexpr (CLDS) sq_run += sc_sq;
            else scl[wave * SQW + (PAIRS ? lane : i)] += sc_sq;
        }
        if (wave == solver) { E9_FINE(6) } else { E9_FINE(8) }  // "factor": the solver's trip; "columns": a column wave's
        // the factor, z and [wz | w] of tile rel and the W rows of tile rel - 1 are final
#ifndef E9_EXP_NOB2  // (timing experiment, results wrong: what this barrier costs)
        role_barrier(fbar, fbar_target, lane_entry);
#endif
        E9_FINE(5)
        if (rel > 0 && lane_entry == 0) __hip_atomic_fetch_add(wready, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (!cur) break;
        // ------------------------------------------------------------ P4a: cross / sumx += X~^T [wz | w]
        // the only reader of the x~ tile; the next tile's rows are requested one per k-step behind the MFMAs
        {
            load_pair(qbA, 6);  // the next tile's first digit pair (the table does not depend on the tile)
            if constexpr (CLDS) load_mu();
#if E9_QB_EARLY
            load_pair(qbB, 4);  // ... and its second
#endif
#ifndef E9_P4A_AHEAD
#define E9_P4A_AHEAD 1  // k-steps the LDS operands are requested ahead of their MFMAs (measured: 2 and 3 change nothing -- 95.9 / 95.8 / 95.4 it/s -- and cost registers)
#endif
            constexpr int AH = E9_P4A_AHEAD, NB3 = AH + 1;
            double bzb[NB3][NCG], axb[NB3][RT];
            const __amdgpu_buffer_rsrc_t trs = tile_rsrc(tile + 1);
            // k = sample 4 s + l4 of the step; A: x~[sample][dim 16 r + (lane & 15)] (rows 4 b + i of block b: the operand the
            // 16x16x4 form read); B: [wz | w][sample][column 4 c + (lane & 3)], the same for the four blocks
            const int j4 = lane & 3;
#pragma unroll
            for (int s0 = 0; s0 < AH; ++s0) {
                const int smp = 4 * s0 + l4;
#pragma unroll
                for (int c = 0; c < NCG; ++c) bzb[s0][c] = Gcur[smp * WS + 16 * NTP + 4 * c + j4];
#pragma unroll
                for (int r = 0; r < RT; ++r) axb[s0][r] = Xs[smp * XS + DW * wave + 16 * r + l15];
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (s < RPW) load_row(trs, tile + 1, s);  // unconditional (rows past the end read as zeros)
                if (s + AH < 8) {
                    const int smp = 4 * (s + AH) + l4;
#pragma unroll
                    for (int c = 0; c < NCG; ++c) bzb[(s + AH) % NB3][c] = Gcur[smp * WS + 16 * NTP + 4 * c + j4];
#pragma unroll
                    for (int r = 0; r < RT; ++r) axb[(s + AH) % NB3][r] = Xs[smp * XS + DW * wave + 16 * r + l15];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int c = 0; c < NCG; ++c)
                        accX[r][c] = __builtin_amdgcn_mfma_f64_4x4x4f64(axb[s % NB3][r], bzb[s % NB3][c], accX[r][c], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        E9_FINE(11)
#ifndef E9_EXP_NOB3  // (timing experiment, results wrong: what this barrier costs)
        role_barrier(fbar, fbar_target, lane_entry);  // the x~ tile is free
#endif
        E9_FINE(12)
        // ------------------------------------------------------------ P1 of the next tile
        // (its sample masks go into the slot of tile rel - 3: the back role's iterations up to rel - 2 must be over)
        if (rel >= 2) wait_counter(itdone, 4u * (unsigned)(rel - 1));
        stage_tile(tile + 1, lane);
        if constexpr (GATHER || WEIGHTED) fetch_meta(tile + 2);  // (consumed one trip from here: P4a's row requests, the staging's weights)
        E9_FINE(13)
#ifndef E9_EXP_NOB4  // (timing experiment, results wrong: what this barrier costs)
        role_barrier(fbar, fbar_target, lane_entry);
#endif
        E9_FINE(14)
    }
#ifdef PPCA_PHASE_TIMING
    if (p.dbg && lane_entry == 0)
        for (int i = 0; i < 16; ++i) p.dbg[(int64_t)gridDim.x * 16 + ((int64_t)blockIdx.x * 8 + wave8) * 16 + i] = (double)tfine[i];
#endif

    // ---------------------------------------------------------------- epilogue (front waves)
    {
        const int lane = lane_entry, l15 = lane & 15, l4 = lane >> 4;
        const double sq_w = CLDS ? wave_sum(sq_run) : wave_sum(lane < SQW ? scl[wave * SQW + lane] : 0.0);
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
        for (int r = 0; r < RT; ++r) {
            const int dim = DW * wave + 16 * r + 4 * ((lane >> 2) & 3) + l4;  // D[i][j] of block b: lane 16 i + 4 b + j
            if (dim >= d) continue;
#pragma unroll
            for (int c = 0; c < NCG; ++c) {
                const int col = 4 * c + (lane & 3);
                if (col < K) out[L.cross + (int64_t)dim * K + col] = accX[r][c];
                else if (col == K) out[L.sumx + dim] = accX[r][c];
            }
        }
    }
}

// ------------------------------------------------------------------ launcher
template <int K, bool GATHER, bool WEIGHTED>
static hipError_t launch_em9_t(int grid, const PassArgs &a, hipStream_t s) {
    const size_t lds = sizeof(double) * Cfg9<K>::LDS_DOUBLES;
    static std::atomic<unsigned long long> done{0ull};
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&em9_kernel<K, GATHER, WEIGHTED>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        done.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL((em9_kernel<K, GATHER, WEIGHTED>), dim3(grid), dim3(512), lds, s, a);
    return hipGetLastError();
}

hipError_t em9_debug_counters(unsigned long long *out4, int reset, hipStream_t s) {
    if (hipError_t e = hipMemcpyFromSymbolAsync(out4, HIP_SYMBOL(e9_counters), sizeof(unsigned long long) * 4, 0, hipMemcpyDeviceToHost, s);
        e != hipSuccess)
        return e;
    if (hipError_t e = hipStreamSynchronize(s); e != hipSuccess) return e;
    if (reset) {
        const unsigned long long z[4] = {0, 0, 0, 0};
        if (hipError_t e = hipMemcpyToSymbolAsync(HIP_SYMBOL(e9_counters), z, sizeof(z), 0, hipMemcpyHostToDevice, s); e != hipSuccess) return e;
        return hipStreamSynchronize(s);
    }
    return hipSuccess;
}

bool em9_covers(int k) {
#ifdef PPCA_DEV_K10
    return k == 10;
#else
    return k >= 1 && k <= FUSED_MAX_K;
#endif
}

hipError_t launch_em9(int k, int grid, const PassArgs &a, hipStream_t s) {
    const bool gather = a.rows != nullptr;
#define PPCA_E8_CASE(KK) \
    case KK:             \
        return gather ? launch_em9_t<KK, true, true>(grid, a, s) : (a.w ? launch_em9_t<KK, false, true>(grid, a, s) : launch_em9_t<KK, false, false>(grid, a, s));
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
