#include "hostmath.h"
#include <algorithm>
#include <cmath>
#include <stdexcept>

namespace fhelin {

bool is_prime_u64(u64 n) {
    if (n < 2) return false;
    static const u64 small[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    for (u64 p : small) {
        if (n == p) return true;
        if (n % p == 0) return false;
    }
    u64 d = n - 1;
    int s = 0;
    while ((d & 1) == 0) {
        d >>= 1;
        ++s;
    }
    // deterministic Miller-Rabin for n < 2^64 with the first 12 prime bases
    for (u64 a : small) {
        u64 x = h_powmod(a, d, n);
        if (x == 1 || x == n - 1) continue;
        bool comp = true;
        for (int r = 1; r < s; ++r) {
            x = h_mulmod(x, x, n);
            if (x == n - 1) {
                comp = false;
                break;
            }
        }
        if (comp) return false;
    }
    return true;
}

u64 prev_prime_congruent(u64 upper, u64 m) {
    if (upper <= m + 1) return 0;
    u64 c = upper - 1;
    c -= (c - 1) % m;  // largest c <= upper-1 with c == 1 mod m
    for (; c > m; c -= m)
        if (is_prime_u64(c)) return c;
    return 0;
}

u64 next_prime_congruent(u64 lower, u64 m) {
    u64 c = lower + 1;
    u64 r = (c - 1) % m;
    if (r) c += m - r;  // smallest c >= lower+1 with c == 1 mod m
    for (;; c += m)
        if (is_prime_u64(c)) return c;
}

u32 bitrev32(u32 x, int bits) {
    u32 r = 0;
    for (int i = 0; i < bits; ++i) {
        r = (r << 1) | (x & 1);
        x >>= 1;
    }
    return r;
}

u64 min_primitive_root(u64 q, u64 two_n) {
    if ((q - 1) % two_n) throw std::runtime_error("q != 1 mod 2N");
    u64 e = (q - 1) / two_n;
    u64 root = 0;
    for (u64 x = 2; x < q; ++x) {
        u64 r = h_powmod(x, e, q);
        // order of r divides 2N (a power of two); it is exactly 2N iff r^N == -1
        if (h_powmod(r, two_n / 2, q) == q - 1) {
            root = r;
            break;
        }
    }
    if (!root) throw std::runtime_error("no primitive root found");
    // all primitive 2N-th roots are root^k, k odd: take the smallest as the canonical psi
    u64 r2 = h_mulmod(root, root, q);
    u64 cur = root, best = root;
    for (u64 k = 1; k < two_n; k += 2) {
        if (cur < best) best = cur;
        cur = h_mulmod(cur, r2, q);
    }
    return best;
}

PrimeChain make_prime_chain(int log_n, int n_q, int first_bits, int scale_bits, int n_p, int special_bits) {
    if (n_q < 1) throw std::runtime_error("need at least one Q prime");
    const u64 m = 2ull << log_n;
    PrimeChain pc;
    pc.q.assign(n_q, 0);
    std::vector<u64> used;
    auto is_used = [&](u64 v) { return std::find(used.begin(), used.end(), v) != used.end(); };
    const int L = n_q - 1;
    if (L >= 1) {
        u64 ql = prev_prime_congruent(1ull << scale_bits, m);
        if (!ql) throw std::runtime_error("no scaling prime");
        pc.q[L] = ql;
        used.push_back(ql);
        long double sf = (long double)ql;
        int cnt = 0;
        for (int i = L - 1; i >= 1; --i) {
            sf = sf * sf / (long double)pc.q[i + 1];
            u64 c;
            if ((cnt & 1) == 0) {
                u64 start = (u64)floorl(sf) + 1;  // exclusive upper bound
                c = prev_prime_congruent(start, m);
                while (c && is_used(c)) c = prev_prime_congruent(c, m);
            } else {
                u64 start = (u64)ceill(sf) - 1;  // exclusive lower bound
                c = next_prime_congruent(start, m);
                while (is_used(c)) c = next_prime_congruent(c, m);
            }
            if (!c) throw std::runtime_error("scaling prime search failed");
            pc.q[i] = c;
            used.push_back(c);
            ++cnt;
        }
    }
    {
        u64 c = prev_prime_congruent(1ull << first_bits, m);
        while (c && is_used(c)) c = prev_prime_congruent(c, m);
        if (!c) throw std::runtime_error("no first prime");
        pc.q[0] = c;
        used.push_back(c);
    }
    u64 c = 1ull << special_bits;
    for (int j = 0; j < n_p; ++j) {
        c = prev_prime_congruent(c, m);
        while (c && is_used(c)) c = prev_prime_congruent(c, m);
        if (!c) throw std::runtime_error("no special prime");
        pc.p.push_back(c);
        used.push_back(c);
    }
    return pc;
}

TwiddleTable make_twiddles(u64 q, int log_n) {
    const u64 n = 1ull << log_n;
    TwiddleTable t;
    t.psi = min_primitive_root(q, 2 * n);
    const u64 ipsi = h_invmod(t.psi, q);
    t.fwd.assign(2 * n, 0);
    t.inv.assign(2 * n, 0);
    u64 pw = 1, ipw = 1;
    for (u64 i = 0; i < n; ++i) {
        u32 r = bitrev32((u32)i, log_n);
        t.fwd[2 * r] = pw;
        t.fwd[2 * r + 1] = h_shoup(pw, q);
        t.inv[2 * r] = ipw;
        t.inv[2 * r + 1] = h_shoup(ipw, q);
        pw = h_mulmod(pw, t.psi, q);
        ipw = h_mulmod(ipw, ipsi, q);
    }
    t.n_inv = h_invmod(n % q, q);
    t.n_inv_s = h_shoup(t.n_inv, q);
    t.w1_n_inv = h_mulmod(t.inv[2], t.n_inv, q);
    t.w1_n_inv_s = h_shoup(t.w1_n_inv, q);
    return t;
}

}  // namespace fhelin
