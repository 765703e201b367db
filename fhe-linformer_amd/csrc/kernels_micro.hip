// Instruction-rate probes for the 64-bit modular-integer hot path on gfx950.
// The NTT butterfly is ~10 32x32-bit multiplies plus ~20 32-bit add/compare/select VALU ops; whether the
// kernel is HBM- or VALU-bound depends on the issue rate of v_mul_lo_u32 / v_mul_hi_u32 / v_mad_u64_u32,
// which the CDNA4 guides do not list.  bench.py --micro runs these probes and reports ops/clk/CU so that
// DESIGN.md's roofline argument rests on measured numbers.
#include <hip/hip_runtime.h>
#include "kernels.h"

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
        }
    }
    out[t] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
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
        default: break;
    }
}

}  // namespace fhelin
