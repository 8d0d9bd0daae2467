// ppca_device.hpp -- device-side helpers shared by the fused-pass kernels (ppca_kernels.hip, ppca_llk.hip):
// the LDS layout constants of a 32-sample tile, wave reductions on the DPP path, lane-targeted writes, compile-time
// loops, the fp64 MFMA wrapper and the constants of the int8-sliced Gram table.  Not installed.
#pragma once

#include <type_traits>
#include <utility>

#include "ppca_internal.hpp"

namespace ppca {

typedef double d4_t __attribute__((ext_vector_type(4)));

template <int K>
struct Cfg {
    static constexpr int KP = K * (K + 1) / 2;
    static constexpr int NTP = (KP + 15) / 16;  // column tiles holding vech(P)
    static constexpr int NTM = NTP + 1;         // + the [w z | w] tile
    static constexpr int B = FUSED_TILE;
    static constexpr int DP = FUSED_MAX_D;
    static constexpr int XS = DP + 2;           // x~ tile row stride (doubles): conflict-free A-operand reads
    static constexpr int CS = K + 1;            // C tile row stride; column K is all zeros
    static constexpr int GS = 16 * NTM + 1;     // [G | b] exchange row stride
    static constexpr int WS = 16 * NTM + 2;     // [wP | wz | w] row stride
    static constexpr int OFF_X = 0;
    static constexpr int OFF_C = OFF_X + B * XS;
    static constexpr int OFF_G = OFF_C + DP * CS;
    static constexpr int OFF_W = OFF_G + 2 * B * GS;
    static constexpr int OFF_M = OFF_W + B * WS;  // mask words, B x 4 u64
    static constexpr int OFF_S = OFF_M + 2 * B * 4;  // (two parities: the next tile is staged behind P4)
                                                     // then xx[B] doubles, m[B] ints
    // running per-solver-lane scalars (kept out of registers: at this pressure hipcc parks loop-carried
    // once-per-tile values in scratch and reloads them with an exposed vmcnt(0)):
    // sq[NW <= 8 waves][B] | dev[B] | llk[B] | sumw[B] | nonempty[B] | det mantissa[B] | det exponent[B]
    static constexpr int OFF_L = OFF_S + 2 * B;
    // int8 form of the mask-side statistics contraction (pass_kernel, P4I8): the second K-half of b moves out of the
    // second [G | b] buffer (which then holds a tile's digit planes) into P1 [B][K + 1]; column exponents of the
    // group's two tiles (2 x 16 NTM ints) and a flag word follow.  The per-dimension sample masks of a group
    // (256 x u64) live in the part of the L region that only the 8-wave variant uses.
    static constexpr int OFF_P1 = OFF_L + 22 * B;  // (sq has two slots per solver sample when the waves pair lanes)
    static constexpr int OFF_E = OFF_P1 + B * (K + 1);
    static constexpr int LDS_DOUBLES = OFF_E + 16 * NTM + 2;
    static constexpr int OFF_MB = OFF_L + 14 * B;  // 4 waves use L[0 .. 14 B); 256 u64 fit in [14 B, 22 B)
    static_assert(K > FUSED_MAX_K || LDS_DOUBLES * 8 <= 160 * 1024, "LDS budget");  // (k > 10: only KP / NTP are used, by qprep_kernel)
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Wave-wide sum on the DPP data path (VALU speed, no LDS crossbar): quad swaps, half-row and row mirrors,
// then row_bcast15 / row_bcast31 carry the row totals forward; lane 63 ends up with the total, which is
// returned as a wave-uniform value.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave_total(double v) {
    v += dpp_f64<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
    v += dpp_f64<0x141, 0xF>(v);  // row_half_mirror
    v += dpp_f64<0x140, 0xF>(v);  // row_mirror: every lane of a 16-lane row holds the row sum
    v += dpp_f64<0x142, 0xA>(v);  // row_bcast15 into rows 1 and 3
    v += dpp_f64<0x143, 0xC>(v);  // row_bcast31 into rows 2 and 3: lane 63 = total
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)b, 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// a (lanes 0-31 | 32-63), b -> lanes 0-31: a_lo + a_hi, lanes 32-63: b_lo + b_hi (v_permlane32_swap exchanges the
// upper half of its first operand with the lower half of its second)
__device__ __forceinline__ double fold_halves(double a, double b) {
    const long long ab = __double_as_longlong(a), bb = __double_as_longlong(b);
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)ab, (unsigned)bb, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(ab >> 32), (unsigned)(bb >> 32), false, false);
    return __longlong_as_double(((long long)hi[0] << 32) | lo[0]) + __longlong_as_double(((long long)hi[1] << 32) | lo[1]);
}
// rows of 16 lanes (r0 r1 r2 r3): -> rows 0, 2: a_even + a_odd, rows 1, 3: b_even + b_odd (v_permlane16_swap exchanges
// the odd rows of its first operand with the even rows of its second)
__device__ __forceinline__ double fold_rows(double a, double b) {
    const long long ab = __double_as_longlong(a), bb = __double_as_longlong(b);
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)ab, (unsigned)bb, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(ab >> 32), (unsigned)(bb >> 32), false, false);
    return __longlong_as_double(((long long)hi[0] << 32) | lo[0]) + __longlong_as_double(((long long)hi[1] << 32) | lo[1]);
}

// Eight per-lane partial sums (one register per row of a wave's eight staged rows) -> the eight row totals into
// out[0..7]: two rows folded per add across the wave halves, then across the 16-lane rows, then four DPP steps for
// the last two registers -- 42 vector instructions instead of eight six-step reductions (8 x 22).
__device__ __forceinline__ void store_row_sums(const double (&pxx)[8], int lane, double *out) {
    double u[4], w2[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i] = fold_halves(pxx[2 * i], pxx[2 * i + 1]);  // half h holds row 2 i + h
#pragma unroll
    for (int j = 0; j < 2; ++j) w2[j] = fold_rows(u[2 * j], u[2 * j + 1]);  // 16-lane row rho: row 4 j + 2 (rho & 1) + (rho >> 1)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        double v = w2[j];
        v += dpp_f64<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
        v += dpp_f64<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
        v += dpp_f64<0x141, 0xF>(v);  // row_half_mirror
        v += dpp_f64<0x140, 0xF>(v);  // row_mirror: every lane of a 16-lane row holds the row sum
        w2[j] = v;
    }
    if ((lane & 15) == 0) {
        const int rho = lane >> 4, r0 = 2 * (rho & 1) + (rho >> 1);
        out[r0] = w2[0];
        out[4 + r0] = w2[1];
    }
}

// dst[LANE] = sval (wave-uniform value into ONE lane of a VGPR): no compare mask, no select.  The lane
// select is an immediate: the instruction admits a single SGPR operand (constant-bus limit).
template <int LANE>
__device__ __forceinline__ int writelane(int dst, int sval) {
    // s_nop 1: gfx940+ needs 2 wait states between a VALU that writes an SGPR (the v_cmp ballot) and a
    // VALU that reads it; hipcc pads its own code but nothing inside an asm statement.
    asm("s_nop 1\n\tv_writelane_b32 %0, %1, %2" : "+v"(dst) : "s"(sval), "n"(LANE));
    return dst;
}
// As writelane, for a value produced by the SCALAR unit (no VALU-writes-SGPR hazard to pad).
template <int LANE>
__device__ __forceinline__ int writelane_s(int dst, int sval) {
    asm("v_writelane_b32 %0, %1, %2" : "+v"(dst) : "s"(sval), "n"(LANE));
    return dst;
}
// Both halves of a 64-bit wave mask (a v_cmp ballot) into lane LANE of two VGPRs; one s_nop covers the hazard for both.
template <int LANE>
__device__ __forceinline__ void writelane_mask(int &dlo, int &dhi, unsigned long long m) {
    asm("s_nop 1\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4"
        : "+v"(dlo), "+v"(dhi)
        : "s"((int)(unsigned)m), "s"((int)(unsigned)(m >> 32)), "n"(LANE));
}
// writelane_mask + the lane's own bit of the mask shifted into `bits` (bits = 2 bits + bit: v_addc with the mask as the
// carry-in), so that after a wave's eight staged rows `bits` holds the lane's dimension over those samples.
template <int LANE>
__device__ __forceinline__ void file_mask(int &dlo, int &dhi, int &bits, unsigned long long m) {
    asm("s_nop 1\n\tv_writelane_b32 %0, %3, %5\n\tv_writelane_b32 %1, %4, %5\n\tv_addc_co_u32_e64 %2, vcc, %2, %2, %6"
        : "+v"(dlo), "+v"(dhi), "+v"(bits)
        : "s"((int)(unsigned)m), "s"((int)(unsigned)(m >> 32)), "n"(LANE), "s"(m)
        : "vcc");
}
// mask bit of the lane ? v : 0.0 with the wave mask taken straight from its SGPR pair (a C++ select on
// (mask >> lane) & 1 would rebuild the predicate with vector shifts).
__device__ __forceinline__ double keep_if(double v, unsigned long long mask) {
    const long long b = __double_as_longlong(v);
    int lo, hi;
    asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(lo) : "v"((int)b), "s"(mask));
    asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(hi) : "v"((int)(b >> 32)), "s"(mask));
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ d4_t mfma(double a, double b, d4_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------ int8-sliced Gram operand (see ppca_kernels.hip)
constexpr int QS = 8;
// Digit width of the table: QB = 8 bits (balanced base 256, digits in [-128, 127]) since round 3 -- 62 bits below each
// column maximum instead of the 54 of the 7-bit digits of rounds 1-2, at the same eight slices: the dynamic-range guard
// admits 256 x smaller sigma^2 / row norms before it falls back on the fp64 engine.  Digit sums over 256 dims stay
// below 2^15, a pair (hi * 2^QB + lo) below 2^24: exact in i32.
constexpr int QB = 8;
constexpr int QBASE = 1 << QB;
typedef int i4_t __attribute__((ext_vector_type(4)));

// Seven independent v_mfma_i32_16x16x64_i8 (one A operand, seven B operands, C = 0) with the results in VGPRs: the
// builtin's results land in accumulation registers, and under this kernel's register pressure all seven shared one
// quad -- every MFMA waited for the previous result to be copied out.  The trailing s_nops cover the read-after-MFMA
// wait states of the last result (the compiler does not see inside the statement); the leading s_nop 1 covers a VALU
// write of an A / B operand register by the instruction just before the statement (hipcc pads nothing across the
// boundary: seen as plane-0 digit sums of the first block contracted under a half-written mask operand, 1e-10 of S, in
// ppca_em16.hip's instantiations for k = 11..15).
__device__ __forceinline__ void mfma_i8_x7(const i4_t &a, const i4_t (&b)[7], i4_t (&d)[7]) {
    asm volatile(
        "s_nop 1\n\t"
        "v_mfma_i32_16x16x64_i8 %0, %7, %8, 0\n\t"
        "v_mfma_i32_16x16x64_i8 %1, %7, %9, 0\n\t"
        "v_mfma_i32_16x16x64_i8 %2, %7, %10, 0\n\t"
        "v_mfma_i32_16x16x64_i8 %3, %7, %11, 0\n\t"
        "v_mfma_i32_16x16x64_i8 %4, %7, %12, 0\n\t"
        "v_mfma_i32_16x16x64_i8 %5, %7, %13, 0\n\t"
        "v_mfma_i32_16x16x64_i8 %6, %7, %14, 0\n\t"
        "s_nop 7"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6])
        : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]));
}

// Three / one independent v_mfma_i32_16x16x64_i8 with a shared A operand (C = 0), results in VGPRs (see mfma_i8_x7).
__device__ __forceinline__ void mfma_i8_x3(const i4_t &a, const i4_t &b0, const i4_t &b1, const i4_t &b2, i4_t &d0, i4_t &d1, i4_t &d2) {
    asm volatile(
        "s_nop 1\n\t"
        "v_mfma_i32_16x16x64_i8 %0, %3, %4, 0\n\t"
        "v_mfma_i32_16x16x64_i8 %1, %3, %5, 0\n\t"
        "v_mfma_i32_16x16x64_i8 %2, %3, %6, 0\n\t"
        "s_nop 7"
        : "=&v"(d0), "=&v"(d1), "=&v"(d2)
        : "v"(a), "v"(b0), "v"(b1), "v"(b2));
}
__device__ __forceinline__ void mfma_i8_x1(const i4_t &a, const i4_t &b0, i4_t &d0) {
    asm volatile(
        "s_nop 1\n\t"
        "v_mfma_i32_16x16x64_i8 %0, %1, %2, 0\n\t"
        "s_nop 7"
        : "=&v"(d0)
        : "v"(a), "v"(b0));
}

// ln(x) for a positive normal x with its polynomial constants read from constant memory at the point of use (scalar loads
// into SGPRs).  The library logarithm's constants are materialised in vector registers and hoisted out of the tile loop;
// under the register pressure of an eight-wave kernel they were what the allocator spilled (llk8_kernel: seven doubles
// reloaded from scratch inside the solver step, each reload waiting on the row loads in flight).  Method: x = f 2^e,
// f in [sqrt(1/2), sqrt(2)); s = (f - 1) / (f + 1), ln f = 2 s (1 + s^2/3 + s^4/5 + ...): s^2 <= 0.0295, ten terms leave
// < 1e-17 relative.
__constant__ double LEAN_LOG_C[10] = {1.0 / 3.0,  1.0 / 5.0,  1.0 / 7.0,  1.0 / 9.0,  1.0 / 11.0,
                                      1.0 / 13.0, 1.0 / 15.0, 1.0 / 17.0, 1.0 / 19.0, 1.0 / 21.0};
__device__ __forceinline__ double lean_log(double x) {
    int e = __builtin_amdgcn_frexp_exp(x);
    double f = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
    const bool low = f < 0.7071067811865476;
    f = low ? f + f : f;
    e = low ? e - 1 : e;
    const double den = f + 1.0;
    double r = __builtin_amdgcn_rcp(den);
    r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
    const double num = f - 1.0;
    double s = num * r;
    s = __builtin_fma(__builtin_fma(-den, s, num), r, s);  // one correction of the quotient
    const double z = s * s;
    double pz = LEAN_LOG_C[9];
#pragma unroll
    for (int i = 8; i >= 0; --i) pz = __builtin_fma(pz, z, LEAN_LOG_C[i]);
    const double lf = __builtin_fma(2.0 * s * z, pz, 2.0 * s);
    return __builtin_fma((double)e, LN_2, lf);
}

template <int K>
constexpr size_t qtab_bytes() { return (size_t)Cfg<K>::NTP * QS * 4 * 1024; }

// One model's block of fused_qtab_bytes(): [64 scales | 8 doubles of guard words | digit table (sized for k = FUSED_MAX_K) | zero-padded
// C | C in em9_kernel's operand order].  The host's fused_qtab_layout and the multi-component kernels (one block per mixture
// component, MixLlkArgs::tab) point a PassArgs at it with this.
__host__ __device__ inline void fused_qtab_view(void *base, PassArgs &a) {
    a.qscale = static_cast<double *>(base);
    a.qflag = reinterpret_cast<int *>(a.qscale + 64);
    a.qtab = reinterpret_cast<signed char *>(a.qscale + 72);
    a.cpad = reinterpret_cast<const double *>(a.qtab + qtab_bytes<FUSED_MAX_K>());  // (behind the largest table)
    a.cpb = a.cpad + FUSED_MAX_D * (FUSED_MAX_K + 1);
}


}  // namespace ppca
