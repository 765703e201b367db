"""Slot-level (plaintext, float64) restatement of the reference FHEController's composite circuit ops.
TEST INFRASTRUCTURE (part of oracle/): decrypted GPU results are compared with these within CKKS tolerance.

Each function follows the reference method of the same name in src/FHEController.cpp (line cited).  A
ciphertext is modelled as its slot vector; EvalRotate(c, i) is a LEFT shift by i (np.roll(x, -i)),
consistent with every packing layout in the reference."""
import math

import numpy as np


def rot(x, i):
    return np.roll(x, -i)


def _steps(slots):                       # `for (i = 0; i < log2(slots); i++)`  :832
    n = 0
    while n < math.log2(slots):
        n += 1
    return n


def rotsum(x, slots, padding):           # :829-837
    r = x.copy()
    for i in range(_steps(slots)):
        r = r + rot(r, padding * 2 ** i)
    return r


def repeat(x, slots, padding=1):         # :849-867
    r = x.copy()
    for i in range(_steps(slots)):
        r = r + rot(r, padding * -(2 ** i))
    return r


def mask_block(x, frm, to, v=1.0):       # :1207
    m = np.zeros_like(x)
    m[max(frm, 0):to] = v
    return x * m


def mask_mod_n(x, n, padding=0, v=1.0):  # :1249,:1262 (and mask_heads :1221, mask_heads_128 :1235 with v)
    m = np.zeros_like(x)
    m[padding::n] = v
    return x * m


def mask_first_n(x, n, v=1.0):           # :1275
    m = np.zeros_like(x)
    m[:n] = v
    return x * m


def matmul(rows, w, bias, slots, padding):       # matmulRE :869-899 / matmulCR :982-996
    out = []
    for r in rows:
        m = rotsum(r * w, slots, padding)
        if bias is not None:
            m = m + bias
        out.append(m)
    return out


def matmulRElarge(inputs, weights, bias, mask_val=1.0):   # :915-944
    out = []
    for x in inputs:
        res = None
        for j in range(len(weights) - 1, -1, -1):
            o = mask_first_n(rotsum(x * weights[j], 128, 128), 128, mask_val)
            if j == len(weights) - 1:
                res = o
            else:
                res = rot(rot(res, -64), -64) + o
        out.append(res + bias if bias is not None else res)
    return out


def matmulCRlarge(rows, weights, bias):          # :998-1026
    out = []
    for r in rows:
        res = rotsum(sum(r[j] * weights[j] for j in range(4)), 128, 1)
        out.append(res + bias if bias is not None else res)
    return out


def matmulScores(queries, key):                  # :1028-1058
    scores = [rotsum(q * key, 128, 1) for q in queries]
    v = 1 / 8.0 * (1 / 8.0)
    if len(scores) == 1:
        return mask_mod_n(scores[0], 128, 0, v)
    w = rot(mask_mod_n(scores[-1], 128, 0, v), -1)
    for i in range(len(scores) - 2, -1, -1):
        w = w + mask_mod_n(scores[i], 128, 0, v)
        if i > 0:
            w = rot(w, -1)
    return w


def wrapUpRepeated(v):                           # :1060-1068
    return sum(mask_block(x, 128 * i, 128 * (i + 1)) for i, x in enumerate(v))


def wrapUpExpanded(v):                           # :1070-1084
    m = mask_mod_n(v[-1], 128)
    if len(v) > 1:
        m = rot(m, -1)
    for i in range(len(v) - 2, -1, -1):
        m = m + mask_mod_n(v[i], 128)
        if i > 0:
            m = rot(m, -1)
    return m


def unwrapExpanded(c, n):                        # :1086-1100
    out = []
    for i in range(n):
        out.append(repeat(mask_mod_n(c, 128, 0), 128))
        if i < n - 1:
            c = rot(c, 1)
    return out


def unwrapScoresExpanded(c, n):                  # :1125-1140
    out = []
    for i in range(n):
        a = repeat(mask_mod_n(c, 128, 0), 64)
        b = repeat(mask_mod_n(c, 128, 64), 64)
        if i < n - 1:
            c = rot(c, 1)
        out.append(a + b)
    return out


def unwrap_512_in_4_128(c, index):               # :1142-1162
    s = index * 512
    return [repeat(mask_block(c, s + 128 * k, s + 128 * (k + 1)), 128, -128) for k in range(4)]


def wrap_containers(c, n):                       # :1193-1205
    r = c[0]
    for i in range(1, n):
        r = rot(r, -512) + c[i]
    return r


def generate_containers(inputs, bias):           # :1164-1191 (with slicing :1338-1357)
    out = []
    total = len(inputs)
    i = 0
    while i < total / 32.0:
        q = 32 if (i + 1) * 32 <= total else total - i * 32
        X, Y = i * 32, (i + 1) * 32
        sl = list(inputs) if Y - X >= total else list(inputs[X:min(Y, total)])
        sl.reverse()
        part = wrap_containers(sl, q)
        out.append(part + bias if bias is not None else part)
        i += 1
    return out


def unwrapRepeatedLarge(containers, input_number):   # :1102-1123
    out = []
    qs = []
    i = 0
    while i < input_number / 32.0:
        qs.append(32 if (i + 1) * 32 <= input_number else input_number - i * 32)
        i += 1
    for ci, c in enumerate(containers):
        for j in range(qs[ci]):
            out.append(unwrap_512_in_4_128(c, j))
    return out
