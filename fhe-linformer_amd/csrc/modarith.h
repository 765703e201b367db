// 64-bit modular arithmetic shared by the HIP kernels and the host-side client code.
//
// Every residue on the hot path is an unsigned 64-bit integer in [0, q) with q < 2^61
// (55-bit first prime, 52-bit scaling primes, 60-bit special primes; parameters follow
// reference src/FHEController.cpp:6-31).  Three multiplication flavours are used:
//   * Shoup ("precomputed quotient") for constant operands: twiddles, basis-conversion
//     constants, q_l^{-1}, P^{-1}.       cost: 1 mulhi64 + 2 mullo64
//   * Barrett (2-word ratio floor(2^128/q)) for data x data products (ct x pt, tensor).
//   * 128-bit accumulate + one Barrett reduction for the fast-basis-conversion sums.
// The functions are exact: every op on this path is a mathematical function mod q, which is
// what makes GPU results comparable bit-for-bit with oracle/.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FHE_HD __host__ __device__ __forceinline__
#else
#define FHE_HD inline
#endif

namespace fhelin {

typedef uint64_t u64;
typedef uint32_t u32;
typedef unsigned __int128 u128;

#if defined(__HIP_DEVICE_COMPILE__)
// sign word of a 64-bit value through an asm v_ashrrev_i32, so that instruction selection cannot turn the mask
// arithmetic of the conditional subtractions below back into compare + v_cndmask (measured on gfx950, bench.py
// --micro: v_cndmask_b32 with a VCC mask issues ~5x slower than other VALU instructions)
__device__ __forceinline__ u64 sign_mask64(u64 t) {
    u32 s;
    asm("v_ashrrev_i32 %0, 31, %1" : "=v"(s) : "v"((u32)(t >> 32)));
    return ((u64)s << 32) | s;
}
#endif

FHE_HD u64 mulhi64(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (u64)(((u128)a * b) >> 64);
#endif
}

// ---- Shoup multiplication by a constant w with ws = floor(w * 2^64 / q) ---------------
// result in [0, 2q) for ANY 64-bit x (Harvey 2014); q < 2^63.
FHE_HD u64 mul_shoup_lazy(u64 x, u64 w, u64 ws, u64 q) {
    u64 h = mulhi64(x, ws);
    return x * w - h * q;
}
// x < 2q, q < 2^63 (so that x - q lies in [-q, q))  ->  x >= q ? x - q : x
FHE_HD u64 csub(u64 x, u64 q) {
#if defined(__HIP_DEVICE_COMPILE__)
    const u64 t = x - q;
    return t + (q & sign_mask64(t));
#else
    return x >= q ? x - q : x;
#endif
}
FHE_HD u64 mul_shoup(u64 x, u64 w, u64 ws, u64 q) { return csub(mul_shoup_lazy(x, w, ws, q), q); }
FHE_HD u64 add_mod(u64 a, u64 b, u64 q) { return csub(a + b, q); }
FHE_HD u64 sub_mod(u64 a, u64 b, u64 q) {  // a, b < q < 2^63
#if defined(__HIP_DEVICE_COMPILE__)
    const u64 t = a - b;
    return t + (q & sign_mask64(t));
#else
    return a >= b ? a - b : a + q - b;
#endif
}
FHE_HD u64 neg_mod(u64 a, u64 q) { return a ? q - a : 0; }

// ---- Barrett: ratio (r1:r0) = floor(2^128 / q) ------------------------------------------
struct Barrett {
    u64 q;
    u64 r0;  // low word of floor(2^128/q)
    u64 r1;  // high word
};

// reduce a 128-bit value (hi:lo) modulo q.  Requirements: q < 2^62 and hi < q (value < q * 2^64, so that the
// quotient fits 64 bits); the quotient estimate is short by at most 2, hence two conditional subtractions.
FHE_HD u64 barrett_reduce128(u64 lo, u64 hi, const Barrett& b) {
    // qhat = floor( (hi:lo) * (r1:r0) / 2^128 ), computed without the lowest partial product's low word
    u64 carry = mulhi64(lo, b.r0);
    // lo * r1
    u64 t_lo = lo * b.r1;
    u64 t_hi = mulhi64(lo, b.r1);
    u64 s1 = t_lo + carry;
    u64 c1 = t_hi + (s1 < t_lo);
    // hi * r0
    t_lo = hi * b.r0;
    t_hi = mulhi64(hi, b.r0);
    u64 s2 = s1 + t_lo;
    u64 c2 = t_hi + (s2 < t_lo);
    u64 qhat = hi * b.r1 + c1 + c2;
    u64 r = lo - qhat * b.q;  // < 3q
    r = csub(r, b.q);
    r = csub(r, b.q);
    return r;
}

// Montgomery reduction of a 128-bit value: (hi:lo) * 2^-64 mod q, canonical.  Requirements: q odd, hi < q (value < q * 2^64), qinv = q^-1
// mod 2^64.  m = lo * qinv makes m * q agree with the value in its low word, so (value - m q) / 2^64 = hi - mulhi64(m, q) exactly, in (-q, q).
// One mullo64 + one mulhi64 + a correction (~15 VALU instructions) where barrett_reduce128 takes three of each (~45): the basis conversions
// use it with their host-side constants stored times 2^64 (modulo the target), so that the result is the canonical residue of the plain sum.
FHE_HD u64 redc128(u64 lo, u64 hi, u64 q, u64 qinv) {
    const u64 m = lo * qinv;
    const u64 u = mulhi64(m, q);
    const u64 t = hi - u;
    return hi < u ? t + q : t;
}

FHE_HD u64 mul_mod(u64 a, u64 c, const Barrett& b) {
    return barrett_reduce128(a * c, mulhi64(a, c), b);
}

// 128-bit accumulator for sums of up to 16 products of 60-bit numbers.
struct Acc128 {
    u64 lo, hi;
};
FHE_HD void acc_mac(Acc128& a, u64 x, u64 y) {
    u64 pl = x * y;
    u64 ph = mulhi64(x, y);
    a.lo += pl;
    a.hi += ph + (a.lo < pl);
}

// ---- 30-bit split accumulation -------------------------------------------------------------
// Residues are < 2^60, so x = x1*2^30 + x0 with x0, x1 < 2^30 and every partial product is < 2^60: up to 8
// products can be summed per column in plain 64-bit registers with NO carry handling.  One multiply-accumulate
// is then exactly four v_mad_u64_u32 (vs ~16 VALU ops for a carried 64x64->128 MAC).
struct Acc30 {
    u64 s0, s1, s2;  // value = s0 + s1*2^30 + s2*2^60
};
FHE_HD void split30(u64 x, u32& lo, u32& hi) {
    lo = (u32)x & 0x3FFFFFFFu;
    hi = (u32)(x >> 30);
}
FHE_HD u64 pack30(u64 x) { return (x & 0x3FFFFFFFull) | ((x >> 30) << 32); }  // host: constants stored pre-split
FHE_HD void mac30(Acc30& a, u32 x0, u32 x1, u32 y0, u32 y1) {
    a.s0 += (u64)x0 * y0;
    a.s1 += (u64)x0 * y1;
    a.s1 += (u64)x1 * y0;
    a.s2 += (u64)x1 * y1;
}
// fold an Acc30 (<= 8 accumulated products) into a running 128-bit sum
FHE_HD void acc30_flush(const Acc30& a, u64& lo, u64& hi) {
    u64 t = lo + a.s0;
    hi += (t < lo);
    lo = t;
    t = lo + (a.s1 << 30);
    hi += (a.s1 >> 34) + (t < lo);
    lo = t;
    t = lo + (a.s2 << 60);
    hi += (a.s2 >> 4) + (t < lo);
    lo = t;
}

#if defined(__HIPCC__)
// ---- gfx950 multiply-add chains ---------------------------------------------------------------
// The NTT is bound by VALU issue slots: what counts is the NUMBER of vector instructions per
// butterfly.  v_mad_u64_u32 does a 32x32 multiply and a 64-bit add in one slot; chains of it (inline asm, so that
// instruction selection cannot re-split them into v_mul_lo/v_mul_hi/v_add3 triples) carry the low words of the
// Shoup product below.
__device__ __forceinline__ u64 mad32(u32 a, u32 b, u64 c) {  // a*b + c (mod 2^64)
    u64 d, cy;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(cy) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ u64 mul32(u32 a, u32 b) {  // a*b
    u64 d, cy;
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(cy) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ u64 mad32s(u32 a, u32 sb, u64 c) {  // a*sb + c with sb wave-uniform (SGPR operand)
    u64 d, cy;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(cy) : "v"(a), "s"(sb), "v"(c));
    return d;
}
// Shoup product with an APPROXIMATE quotient, for the lazy NTT path (primes below 2^53):
//   h~ = x1*s1 + hi32(x0*s1) + hi32(x1*s0)  drops x0*s0 and the low words of the cross products, so
//   h - 3 <= h~ <= h  with h = floor(x*ws/2^64), and  x*w - h~*q  lies in [0, 5q)  for ANY 64-bit x (5q < 2^64).
// nq = 2^64 - q (wave-uniform).  11 vector instructions: 2 v_mul_hi_u32, 7 v_mad_u64_u32, 2 adds.
__device__ __forceinline__ u64 mul_shoup_lazy5(u64 x, u64 w, u64 ws, u64 nq) {
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32), s0 = (u32)ws, s1 = (u32)(ws >> 32);
    const u64 h = mad32(x1, s1, (u64)__umulhi(x0, s1)) + (u64)__umulhi(x1, s0);
    const u32 w0 = (u32)w, w1 = (u32)(w >> 32), h0 = (u32)h, h1 = (u32)(h >> 32);
    const u32 n0 = (u32)nq, n1 = (u32)(nq >> 32);
    u64 c = mul32(x0, w1);  // cross terms: only their low 32 bits reach the result
    c = mad32(x1, w0, c);
    c = mad32s(h0, n1, c);
    c = mad32s(h1, n0, c);
    u64 r = mul32(x0, w0);
    r = mad32s(h0, n0, r);
    u32 rh;  // r += c << 32 touches the high word only (asm: keeps it one v_add_u32 instead of a 64-bit add)
    asm("v_add_u32 %0, %1, %2" : "=v"(rh) : "v"((u32)(r >> 32)), "v"((u32)c));
    return ((u64)rh << 32) | (u32)r;
}
// acc + x*w - h~*q  (mod 2^64), h~ as above: the product of mul_shoup_lazy5 added to `acc` for free — the chain's first
// multiply-add takes acc as its addend instead of 0.  A forward butterfly (X, Y) -> (X + T, X + 5q - T) then needs
// A = shoup_lazy5_add(Y, w, X) and B = (2X + 5q) - A: 11 + 1 + 2 = 14 vector instructions instead of 15.
__device__ __forceinline__ u64 mul_shoup_lazy5_add(u64 x, u64 w, u64 ws, u64 nq, u64 acc) {
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32), s0 = (u32)ws, s1 = (u32)(ws >> 32);
    const u64 h = mad32(x1, s1, (u64)__umulhi(x0, s1)) + (u64)__umulhi(x1, s0);
    const u32 w0 = (u32)w, w1 = (u32)(w >> 32), h0 = (u32)h, h1 = (u32)(h >> 32);
    const u32 n0 = (u32)nq, n1 = (u32)(nq >> 32);
    u64 c = mul32(x0, w1);
    c = mad32(x1, w0, c);
    c = mad32s(h0, n1, c);
    c = mad32s(h1, n0, c);
    u64 r = mad32(x0, w0, acc);
    r = mad32s(h0, n0, r);
    u32 rh;
    asm("v_add_u32 %0, %1, %2" : "=v"(rh) : "v"((u32)(r >> 32)), "v"((u32)c));
    return ((u64)rh << 32) | (u32)r;
}
// (x << 1) + c in one instruction
__device__ __forceinline__ u64 shl1_add(u64 x, u64 c) {
    u64 d;
    asm("v_lshl_add_u64 %0, %1, 1, %2" : "=v"(d) : "v"(x), "v"(c));
    return d;
}
__device__ __forceinline__ u64 csub_mask(u64 x, u64 m) { return csub(x, m); }  // x < 2m: x >= m ? x - m : x (mask form)
// x mod q up to one q:  x < 2^(bits(q)+11)  ->  [0, 2q).   sh = bits(q) - 10,  rr = floor(2^(bits(q)+22) / q);
// k~ = hi32((x >> sh) * rr) is floor(x/q) or one less (error terms < 2^-8).
__device__ __forceinline__ u64 reduce_lazy_2q(u64 x, u64 q, u32 sh, u32 rr) {
    const u32 k = __umulhi((u32)(x >> sh), rr);
    return x - (u64)k * q;
}
#endif

#if !defined(__HIP_DEVICE_COMPILE__)
// ---- host-only helpers ------------------------------------------------------------------
inline u64 h_mulmod(u64 a, u64 b, u64 q) { return (u64)(((u128)a * b) % q); }
inline u64 h_powmod(u64 a, u64 e, u64 q) {
    u64 r = 1 % q;
    a %= q;
    while (e) {
        if (e & 1) r = h_mulmod(r, a, q);
        a = h_mulmod(a, a, q);
        e >>= 1;
    }
    return r;
}
inline u64 h_invmod(u64 a, u64 q) { return h_powmod(a, q - 2, q); }  // q prime
inline u64 h_shoup(u64 w, u64 q) { return (u64)((((u128)w) << 64) / q); }
inline Barrett h_barrett(u64 q) {
    Barrett b;
    b.q = q;
    // floor(2^128 / q) = floor((2^128 - 1) / q) because q is odd and > 1
    u128 all = ~(u128)0;
    u128 r = all / q;
    b.r0 = (u64)r;
    b.r1 = (u64)(r >> 64);
    return b;
}
#endif

}  // namespace fhelin
