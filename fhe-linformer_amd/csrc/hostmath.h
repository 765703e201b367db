// Host-side number theory for the RNS-CKKS parameter layer: NTT-friendly prime selection,
// minimal primitive 2N-th roots, bit-reversed twiddle tables, RNS basis-conversion constants.
//
// The reference delegates all of this to OpenFHE's GenCryptoContext (reference
// src/FHEController.cpp:37, parameters :6-31); OpenFHE is not vendored, so the rules below are
// this engine's own specification (DESIGN.md "Parameter spec") and oracle/ restates them
// independently.
#pragma once
#include <cstdint>
#include <vector>
#include "modarith.h"

namespace fhelin {

bool is_prime_u64(u64 n);
// largest prime p < upper (exclusive) with p == 1 (mod m); 0 if none above 'lower'
u64 prev_prime_congruent(u64 upper, u64 m);
// smallest prime p > lower (exclusive) with p == 1 (mod m)
u64 next_prime_congruent(u64 lower, u64 m);
// minimal primitive (2N)-th root of unity mod q (two_n a power of two dividing q-1)
u64 min_primitive_root(u64 q, u64 two_n);
u32 bitrev32(u32 x, int bits);

struct PrimeChain {
    std::vector<u64> q;  // q[0] = first (55-bit) prime, q[1..L] scaling primes
    std::vector<u64> p;  // special primes
};
// Deterministic FLEXIBLEAUTO-style chain (DESIGN.md "Parameter spec"):
//   q_L   = largest prime == 1 mod 2N below 2^scale_bits
//   q_i   (i = L-1 .. 1) alternately the nearest unused prime below / above sf_i, sf_i = sf_{i+1}^2 / q_{i+1}
//   q_0   = largest unused prime == 1 mod 2N below 2^first_bits
//   p_j   = the k largest unused primes == 1 mod 2N below 2^special_bits
PrimeChain make_prime_chain(int log_n, int n_q, int first_bits, int scale_bits, int n_p, int special_bits);

// Twiddle tables for one modulus: psi_br[i] = psi^{bitrev(i)}, ipsi_br[i] = psi^{-bitrev(i)}, both with
// their Shoup companions, interleaved as (w, w') pairs so the kernels fetch 16 bytes per twiddle.
struct TwiddleTable {
    std::vector<u64> fwd;  // 2*N entries: fwd[2i] = w, fwd[2i+1] = w'
    std::vector<u64> inv;  // 2*N entries, same layout for psi^{-1}
    u64 n_inv, n_inv_s;          // N^{-1} and Shoup companion
    u64 w1_n_inv, w1_n_inv_s;    // ipsi_br[1] * N^{-1} (last GS stage merged with the scaling)
    u64 psi;
};
TwiddleTable make_twiddles(u64 q, int log_n);

}  // namespace fhelin
