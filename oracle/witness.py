"""ORACLE (test infrastructure only).

Restates the request -> witness half of the reference provers:
  ChaChaProver.Prove  gnark/libraries/prover/impl/provers.go:116-173
  AESProver.Prove     gnark/libraries/prover/impl/provers.go:199-263
  bit packing         gnark/utils/bytes.go:12-56
and the ciphers the reference takes from golang.org/x/crypto/chacha20 (RFC 8439)
and Go's crypto/aes + cipher.NewCTR (FIPS-197 / SP 800-38A).
Witness order = public (struct field order, without the constant 1) then secret
(gnark frontend.NewWitness).
"""
from __future__ import annotations

import struct

# --------------------------------------------------------------------- ChaCha20
def _rotl(v, n):
    return ((v << n) & 0xFFFFFFFF) | (v >> (32 - n))


def _qr(s, a, b, c, d):
    s[a] = (s[a] + s[b]) & 0xFFFFFFFF; s[d] = _rotl(s[d] ^ s[a], 16)
    s[c] = (s[c] + s[d]) & 0xFFFFFFFF; s[b] = _rotl(s[b] ^ s[c], 12)
    s[a] = (s[a] + s[b]) & 0xFFFFFFFF; s[d] = _rotl(s[d] ^ s[a], 8)
    s[c] = (s[c] + s[d]) & 0xFFFFFFFF; s[b] = _rotl(s[b] ^ s[c], 7)


def chacha20_block(key: bytes, counter: int, nonce: bytes) -> bytes:
    st = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(struct.unpack("<8I", key)) + [counter & 0xFFFFFFFF] + list(struct.unpack("<3I", nonce))
    w = st[:]
    for _ in range(10):
        _qr(w, 0, 4, 8, 12); _qr(w, 1, 5, 9, 13); _qr(w, 2, 6, 10, 14); _qr(w, 3, 7, 11, 15)
        _qr(w, 0, 5, 10, 15); _qr(w, 1, 6, 11, 12); _qr(w, 2, 7, 8, 13); _qr(w, 3, 4, 9, 14)
    return struct.pack("<16I", *[(w[i] + st[i]) & 0xFFFFFFFF for i in range(16)])


# -------------------------------------------------------------------------- AES
def _make_sbox():
    sbox = [0] * 256
    p = q = 1
    while True:
        p = p ^ ((p << 1) & 0xFF) ^ (0x1B if p & 0x80 else 0)
        q ^= q << 1; q ^= q << 2; q ^= q << 4; q &= 0xFF
        if q & 0x80:
            q ^= 0x09
        x = q ^ _rotl8(q, 1) ^ _rotl8(q, 2) ^ _rotl8(q, 3) ^ _rotl8(q, 4)
        sbox[p] = (x ^ 0x63) & 0xFF
        if p == 1:
            break
    sbox[0] = 0x63
    return sbox


def _rotl8(x, s):
    return ((x << s) | (x >> (8 - s))) & 0xFF


SBOX = _make_sbox()


def _xt(a):
    return ((a << 1) ^ 0x1B) & 0xFF if a & 0x80 else (a << 1)


def aes_expand_key(key: bytes):
    nk = len(key) // 4
    nr = nk + 6
    w = [list(key[4 * i:4 * i + 4]) for i in range(nk)]
    rcon = 1
    for i in range(nk, 4 * (nr + 1)):
        t = w[i - 1][:]
        if i % nk == 0:
            t = [SBOX[t[1]] ^ rcon, SBOX[t[2]], SBOX[t[3]], SBOX[t[0]]]
            rcon = _xt(rcon)
        elif nk > 6 and i % nk == 4:
            t = [SBOX[b] for b in t]
        w.append([w[i - nk][j] ^ t[j] for j in range(4)])
    return [sum(w[4 * r:4 * r + 4], []) for r in range(nr + 1)], nr


def aes_encrypt_block(key: bytes, block: bytes) -> bytes:
    rk, nr = aes_expand_key(key)
    s = [block[i] ^ rk[0][i] for i in range(16)]
    for r in range(1, nr + 1):
        s = [SBOX[b] for b in s]
        s = [s[(i + 4 * (i % 4)) % 16] for i in range(16)]      # ShiftRows (column-major state)
        if r != nr:
            t = []
            for c in range(4):
                a = s[4 * c:4 * c + 4]
                x = a[0] ^ a[1] ^ a[2] ^ a[3]
                t += [a[0] ^ x ^ _xt(a[0] ^ a[1]), a[1] ^ x ^ _xt(a[1] ^ a[2]), a[2] ^ x ^ _xt(a[2] ^ a[3]), a[3] ^ x ^ _xt(a[3] ^ a[0])]
            s = t
        s = [s[i] ^ rk[r][i] for i in range(16)]
    return bytes(s)


# -------------------------------------------------------------------- requests
CHACHA_BLOCKS = 2   # gnark/circuits/chachaV3/common.go:8
AES_BLOCKS = 5      # gnark/circuits/aesV2/common.go:10


def _u32_bits(v):
    return [(v >> i) & 1 for i in range(32)]           # utils/bytes.go:32-48 (LSB first)


def chacha_witness(key: bytes, nonces, counters, data: bytes):
    """-> (witness ints [public..., secret...], output bytes).  provers.go:116-173"""
    if len(key) != 32:
        raise ValueError("key length must be 32: %d" % len(key))
    if len(nonces) != CHACHA_BLOCKS or len(counters) != CHACHA_BLOCKS:
        raise ValueError("blocks array length must be %d: %d" % (CHACHA_BLOCKS, len(nonces)))
    for i, n in enumerate(nonces):
        if len(n) != 12:
            raise ValueError("block[%d] nonce length must be 12: %d" % (i, len(n)))
    if len(data) != 64 * CHACHA_BLOCKS:
        raise ValueError("input length must be %d: %d" % (64 * CHACHA_BLOCKS, len(data)))
    out = bytearray()
    for b in range(CHACHA_BLOCKS):
        ks = chacha20_block(key, counters[b], nonces[b])
        out += bytes(x ^ y for x, y in zip(data[64 * b:64 * b + 64], ks))
    out = bytes(out)
    w = []
    for b in range(CHACHA_BLOCKS):                      # Counter [2][32]
        w += _u32_bits(counters[b])
    for b in range(CHACHA_BLOCKS):                      # Nonce [2][3][32]  (LE words)
        for k in range(3):
            w += _u32_bits(struct.unpack_from("<I", nonces[b], 4 * k)[0])
    for k in range(32):                                 # In [32][32] (BE words)
        w += _u32_bits(struct.unpack_from(">I", data, 4 * k)[0])
    for k in range(32):                                 # Out
        w += _u32_bits(struct.unpack_from(">I", out, 4 * k)[0])
    for k in range(8):                                  # Key [8][32] (LE words), secret
        w += _u32_bits(struct.unpack_from("<I", key, 4 * k)[0])
    return w, out


def aes_witness(key: bytes, nonces, counters, data: bytes):
    """-> (witness ints, output bytes).  provers.go:199-263"""
    if len(key) not in (16, 32):
        raise ValueError("key length must be 16 or 32: %d" % len(key))
    if len(nonces) != AES_BLOCKS or len(counters) != AES_BLOCKS:
        raise ValueError("blocks array length must be %d: %d" % (AES_BLOCKS, len(nonces)))
    for i, n in enumerate(nonces):
        if len(n) != 12:
            raise ValueError("block[%d] nonce length must be 12: %d" % (i, len(n)))
    if len(data) != 16 * AES_BLOCKS:
        raise ValueError("input length must be %d: %d" % (16 * AES_BLOCKS, len(data)))
    out = bytearray()
    for b in range(AES_BLOCKS):
        iv = bytes(nonces[b]) + struct.pack(">I", counters[b])
        ks = aes_encrypt_block(key, iv)
        out += bytes(x ^ y for x, y in zip(data[16 * b:16 * b + 16], ks))
    out = bytes(out)
    w = []
    for b in range(AES_BLOCKS):
        w += list(nonces[b])
    w += [int(c) for c in counters]
    w += list(data) + list(out) + list(key)
    return w, out
