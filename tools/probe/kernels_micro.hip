// Instruction-rate probes for the 64-bit modular-integer hot path on gfx950.
// The NTT butterfly is ~10 32x32-bit multiplies plus ~20 32-bit add/compare/select VALU ops; whether the
// kernel is HBM- or VALU-bound depends on the issue rate of v_mul_lo_u32 / v_mul_hi_u32 / v_mad_u64_u32,
// which the CDNA4 guides do not list.  bench.py --micro runs these probes and reports ops/clk/CU so that
// DESIGN.md's roofline argument rests on measured numbers.
#include <hip/hip_runtime.h>
#include "../../fhe-linformer_amd/csrc/modarith.h"   // the butterflies being probed

namespace fhelin {
namespace {

template <int VARIANT>
__global__ __launch_bounds__(256) void mulbench_kernel(u64* out, int iters) {
    const u32 t = threadIdx.x + blockIdx.x * blockDim.x;
    u64 a0 = 0x9E3779B97F4A7C15ull * (t + 1), a1 = a0 ^ 0xD1B54A32D192ED03ull, a2 = a0 + 0x1234567, a3 = ~a0;
    u64 a4 = a0 * 3, a5 = a1 * 5, a6 = a2 * 7, a7 = a3 * 9;
    const u64 q = 0x0FFFFFFFFFFC0001ull;
    const u64 w = 0x0123456789ABCDEFull % q, ws = 0x1D1D1D1D1D1D1D1Dull;
    for (int i = 0; i < iters; ++i) {
        if (VARIANT == 0) {  // 8 independent v_mul_lo_u32
            u32 b0 = (u32)a0 * (u32)a1, b1 = (u32)a1 * (u32)a2, b2 = (u32)a2 * (u32)a3, b3 = (u32)a3 * (u32)a4;
            u32 b4 = (u32)a4 * (u32)a5, b5 = (u32)a5 * (u32)a6, b6 = (u32)a6 * (u32)a7, b7 = (u32)a7 * (u32)a0;
            a0 = b0; a1 = b1; a2 = b2; a3 = b3; a4 = b4; a5 = b5; a6 = b6; a7 = b7;
            a0 |= 1; a1 |= 1; a2 |= 1; a3 |= 1; a4 |= 1; a5 |= 1; a6 |= 1; a7 |= 1;
        } else if (VARIANT == 1) {  // 8 independent v_mul_hi_u32
            u32 b0 = __umulhi((u32)a0, (u32)a1), b1 = __umulhi((u32)a1, (u32)a2), b2 = __umulhi((u32)a2, (u32)a3);
            u32 b3 = __umulhi((u32)a3, (u32)a4), b4 = __umulhi((u32)a4, (u32)a5), b5 = __umulhi((u32)a5, (u32)a6);
            u32 b6 = __umulhi((u32)a6, (u32)a7), b7 = __umulhi((u32)a7, (u32)a0);
            a0 = b0 | 0x80000001u; a1 = b1 | 0x80000001u; a2 = b2 | 0x80000001u; a3 = b3 | 0x80000001u;
            a4 = b4 | 0x80000001u; a5 = b5 | 0x80000001u; a6 = b6 | 0x80000001u; a7 = b7 | 0x80000001u;
        } else if (VARIANT == 2) {  // 8 independent v_mad_u64_u32
            a0 = (u64)(u32)a0 * (u32)a1 + a2; a1 = (u64)(u32)a1 * (u32)a2 + a3; a2 = (u64)(u32)a2 * (u32)a3 + a4;
            a3 = (u64)(u32)a3 * (u32)a4 + a5; a4 = (u64)(u32)a4 * (u32)a5 + a6; a5 = (u64)(u32)a5 * (u32)a6 + a7;
            a6 = (u64)(u32)a6 * (u32)a7 + a0; a7 = (u64)(u32)a7 * (u32)a0 + a1;
        } else if (VARIANT == 3) {  // 4 Harvey forward butterflies (the NTT inner op)
            const u64 q2 = q << 1;
            u64 X, T;
            X = csub(a0, q2); T = mul_shoup_lazy(a1, w, ws, q); a0 = X + T; a1 = X - T + q2;
            X = csub(a2, q2); T = mul_shoup_lazy(a3, w, ws, q); a2 = X + T; a3 = X - T + q2;
            X = csub(a4, q2); T = mul_shoup_lazy(a5, w, ws, q); a4 = X + T; a5 = X - T + q2;
            X = csub(a6, q2); T = mul_shoup_lazy(a7, w, ws, q); a6 = X + T; a7 = X - T + q2;
        } else if (VARIANT == 4) {  // 8 independent v_fma_f64
            double d0 = __longlong_as_double((a0 & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);
            double d1 = __longlong_as_double((a1 & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);
            double e0 = d0, e1 = d1, e2 = d0 + 1, e3 = d1 + 1, e4 = d0 + 2, e5 = d1 + 2, e6 = d0 + 3, e7 = d1 + 3;
            for (int j = 0; j < 8; ++j) {
                e0 = __fma_rn(e0, d0, d1); e1 = __fma_rn(e1, d0, d1); e2 = __fma_rn(e2, d0, d1); e3 = __fma_rn(e3, d0, d1);
                e4 = __fma_rn(e4, d0, d1); e5 = __fma_rn(e5, d0, d1); e6 = __fma_rn(e6, d0, d1); e7 = __fma_rn(e7, d0, d1);
            }
            a0 ^= __double_as_longlong(e0 + e1 + e2 + e3 + e4 + e5 + e6 + e7);
        } else if (VARIANT == 5) {  // 8 independent 64-bit adds (2 VALU each)
            a0 += a1; a1 += a2; a2 += a3; a3 += a4; a4 += a5; a5 += a6; a6 += a7; a7 += a0;
        } else if (VARIANT == 6) {  // 8 independent mulhi64
            a0 = mulhi64(a0 | 1, a1) | 0x8000000000000001ull; a1 = mulhi64(a1, a2) | 0x8000000000000001ull;
            a2 = mulhi64(a2, a3) | 0x8000000000000001ull; a3 = mulhi64(a3, a4) | 0x8000000000000001ull;
            a4 = mulhi64(a4, a5) | 0x8000000000000001ull; a5 = mulhi64(a5, a6) | 0x8000000000000001ull;
            a6 = mulhi64(a6, a7) | 0x8000000000000001ull; a7 = mulhi64(a7, a0) | 0x8000000000000001ull;
        } else if (VARIANT == 7) {  // 8 independent mullo64
            a0 = (a0 * a1) | 1; a1 = (a1 * a2) | 1; a2 = (a2 * a3) | 1; a3 = (a3 * a4) | 1;
            a4 = (a4 * a5) | 1; a5 = (a5 * a6) | 1; a6 = (a6 * a7) | 1; a7 = (a7 * a0) | 1;
        } else if (VARIANT == 8) {  // 4 lazy butterflies: approximate-quotient Shoup product, no conditional subtraction
            const u64 q5 = 5 * q, nq = 0 - q;
            u64 T;
            T = mul_shoup_lazy5(a1, w ^ a0, ws ^ a2, nq); a1 = a0 + q5 - T; a0 += T;
            T = mul_shoup_lazy5(a3, w ^ a2, ws ^ a4, nq); a3 = a2 + q5 - T; a2 += T;
            T = mul_shoup_lazy5(a5, w ^ a4, ws ^ a6, nq); a5 = a4 + q5 - T; a4 += T;
            T = mul_shoup_lazy5(a7, w ^ a6, ws ^ a0, nq); a7 = a6 + q5 - T; a6 += T;
        } else if (VARIANT == 9) {  // 4 Harvey butterflies with per-butterfly (non-constant) twiddles
            const u64 q2 = q << 1;
            u64 X, T;
            X = csub(a0, q2); T = mul_shoup_lazy(a1, w ^ a0, ws ^ a2, q); a0 = X + T; a1 = X - T + q2;
            X = csub(a2, q2); T = mul_shoup_lazy(a3, w ^ a2, ws ^ a4, q); a2 = X + T; a3 = X - T + q2;
            X = csub(a4, q2); T = mul_shoup_lazy(a5, w ^ a4, ws ^ a6, q); a4 = X + T; a5 = X - T + q2;
            X = csub(a6, q2); T = mul_shoup_lazy(a7, w ^ a6, ws ^ a0, q); a6 = X + T; a7 = X - T + q2;
        }
    }
    out[t] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}


// Pure issue-rate probes: 16 independent instructions of one kind per loop trip, written in asm so that the
// compiler can neither fuse nor reorder them.  KIND: 0 v_mov_b32, 1 v_add_u32, 2 v_lshl_add_u64, 3 v_mad_u64_u32,
// 4 v_mul_lo_u32, 5 v_mul_hi_u32, 6 v_sub_co/v_subb_co pair (8 pairs), 7 v_cndmask_b32, 8 v_add3_u32,
// 9 v_xor_b32, 10 v_lshrrev_b64, 11 v_mad_u32_u24, 12 v_mul_hi_u32 interleaved with v_mad_u64_u32
#define REP16(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7) I(8) I(9) I(10) I(11) I(12) I(13) I(14) I(15)
template <int KIND>
__global__ __launch_bounds__(256) void issue_kernel(u64* out, int iters) {
    const u32 t = threadIdx.x + blockIdx.x * blockDim.x;
    u32 r[16];
    u64 d[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { r[i] = t * 2654435761u + i; d[i] = (u64)r[i] * 0x9E3779B97F4A7C15ull; }
    u32 a = t | 1, b = t * 3 + 7;
    u64 mask = __ballot(t & 1), mk[4] = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (KIND == 0) asm volatile("v_mov_b32 %0, %1" : "=v"(r[i]) : "v"(r[(i + 1) & 15]));
            if (KIND == 1) asm volatile("v_add_u32 %0, %1, %2" : "=v"(r[i]) : "v"(r[i]), "v"(a));
            if (KIND == 2) asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(d[i]) : "v"(d[i]), "v"(d[(i + 1) & 15]));
            if (KIND == 3) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(d[i]) : "v"(a), "v"(b), "v"(d[i]) : "vcc");
            if (KIND == 4) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(r[i]) : "v"(r[i]), "v"(a));
            if (KIND == 5) asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(r[i]) : "v"(r[i]), "v"(a));
            if (KIND == 6 && i < 8) asm volatile("v_sub_co_u32 %0, vcc, %2, %4\n\ts_nop 1\n\tv_subb_co_u32 %1, vcc, %3, %5, vcc" : "=&v"(r[2 * i]), "=&v"(r[2 * i + 1]) : "v"(r[2 * i]), "v"(r[2 * i + 1]), "v"(a), "v"(b) : "vcc");
            if (KIND == 7) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(r[i]) : "v"(r[i]), "v"(a) : "vcc");
            if (KIND == 8) asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(r[i]) : "v"(r[i]), "v"(a), "v"(b));
            if (KIND == 9) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(r[i]) : "v"(r[i]), "v"(a));
            if (KIND == 10) asm volatile("v_lshrrev_b64 %0, 3, %1" : "=v"(d[i]) : "v"(d[i]));
            if (KIND == 11) asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r[i]) : "v"(r[i]), "v"(a), "v"(b));
            if (KIND == 13) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r[i]) : "v"(r[i]), "v"(a), "s"(mask));
            if (KIND == 14) asm volatile("v_cmp_gt_u64_e64 %0, %1, %2" : "=s"(mk[i & 3]) : "v"(d[i]), "v"(d[(i + 1) & 15]));
            if (KIND == 15) asm volatile("v_ashrrev_i32 %0, 31, %1" : "=v"(r[i]) : "v"(r[i]));
            if (KIND == 16) asm volatile("v_and_b32 %0, %1, %2" : "=v"(r[i]) : "v"(r[i]), "v"(a));
            if (KIND == 17) asm volatile("v_min_u32 %0, %1, %2" : "=v"(r[i]) : "v"(r[i]), "v"(a));
            if (KIND == 18) asm volatile("v_sub_co_u32 %0, %1, %2, %3" : "=v"(r[i]), "=s"(mk[i & 3]) : "v"(r[i]), "v"(a));
            if (KIND == 19) asm volatile("v_add_co_u32 %0, vcc, %1, %2" : "=v"(r[i]) : "v"(r[i]), "v"(a) : "vcc");
            if (KIND == 20) asm volatile("v_addc_co_u32 %0, vcc, %1, %2, vcc" : "=v"(r[i]) : "v"(r[i]), "v"(a) : "vcc");
            if (KIND == 21) asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(r[i]) : "v"(r[(i + 5) & 15]), "v"(a));
            if (KIND == 22) asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(r[i]) : "v"(r[i]), "v"(a), "v"(b));
            if (KIND == 23) asm volatile("v_max_u32 %0, %1, %2" : "=v"(r[i]) : "v"(r[i]), "v"(a));
            if (KIND == 12) {
                if (i & 1) asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(r[i]) : "v"(r[i]), "v"(a));
                else asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(d[i]) : "v"(a), "v"(b), "v"(d[i]) : "vcc");
            }
        }
    }
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc ^= d[i] + r[i];
    out[t] = acc ^ mk[0] ^ mk[1] ^ mk[2] ^ mk[3];
}

}  // namespace

void launch_mulbench(u64* out, int iters, int variant, int blocks, hipStream_t s) {
    switch (variant) {
        case 0: hipLaunchKernelGGL((mulbench_kernel<0>), dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 1: hipLaunchKernelGGL((mulbench_kernel<1>), dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 2: hipLaunchKernelGGL((mulbench_kernel<2>), dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 3: hipLaunchKernelGGL((mulbench_kernel<3>), dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 4: hipLaunchKernelGGL((mulbench_kernel<4>), dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 5: hipLaunchKernelGGL((mulbench_kernel<5>), dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 6: hipLaunchKernelGGL((mulbench_kernel<6>), dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 7: hipLaunchKernelGGL((mulbench_kernel<7>), dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 8: hipLaunchKernelGGL((mulbench_kernel<8>), dim3(blocks), dim3(256), 0, s, out, iters); break;
        case 9: hipLaunchKernelGGL((mulbench_kernel<9>), dim3(blocks), dim3(256), 0, s, out, iters); break;
        default: break;
    }
#define ISSUE_CASE(K) case 100 + K: hipLaunchKernelGGL((issue_kernel<K>), dim3(blocks), dim3(256), 0, s, out, iters); break;
    switch (variant) {
        ISSUE_CASE(0) ISSUE_CASE(1) ISSUE_CASE(2) ISSUE_CASE(3) ISSUE_CASE(4) ISSUE_CASE(5) ISSUE_CASE(6) ISSUE_CASE(7)
        ISSUE_CASE(8) ISSUE_CASE(9) ISSUE_CASE(10) ISSUE_CASE(11) ISSUE_CASE(12) ISSUE_CASE(13) ISSUE_CASE(14) ISSUE_CASE(15)
        ISSUE_CASE(16) ISSUE_CASE(17) ISSUE_CASE(18) ISSUE_CASE(19) ISSUE_CASE(20) ISSUE_CASE(21) ISSUE_CASE(22) ISSUE_CASE(23)
        default: break;
    }
#undef ISSUE_CASE
}

}  // namespace fhelin

// ---- self-contained entry point of libfhelin_probe.so (tools/probe/Makefile): not part of the product library ----------------
extern "C" int fhelin_probe_microbench(int device, int variant, int iters, int blocks, float* ms) {
    if (!ms || iters < 1 || blocks < 1 || variant < 0 || (variant > 9 && (variant < 100 || variant > 123))) return 1;
    if (hipSetDevice(device) != hipSuccess) return 2;
    fhelin::u64* out = nullptr;
    hipEvent_t a, b;
    hipStream_t s;
    if (hipMalloc(&out, (size_t)blocks * 256 * sizeof(fhelin::u64)) != hipSuccess) return 3;
    (void)hipStreamCreate(&s);
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    fhelin::launch_mulbench(out, 8, variant, blocks, s);  // warm
    (void)hipEventRecord(a, s);
    fhelin::launch_mulbench(out, iters, variant, blocks, s);
    (void)hipEventRecord(b, s);
    const hipError_t e = hipEventSynchronize(b);
    (void)hipEventElapsedTime(ms, a, b);
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    (void)hipStreamDestroy(s);
    (void)hipFree(out);
    return e == hipSuccess ? 0 : 3;
}
