"""ORACLE (test infrastructure only -- never imported by the product path).

Independent Python decoder for the gnark v0.14 R1CS binary artefacts the reference
prover loads in `gnark/libraries/prover/impl/library.go:112-125`
(`groth16.NewCS(ecc.BN254).ReadFrom`).  gnark itself is a pinned third-party module
(`gnark/go.mod:6`, github.com/consensys/gnark v0.14.0) that is absent from the
mount; the layout below was decoded from the reference's own artefacts
`resources/gnark/r1cs.{chacha20,aes128,aes256}` and is pinned by
their SHA-256 (`impl/library.go:38,43,48`).

The product has its own C++ decoder (zk-symmetric-crypto_amd/csrc/r1cs_reader.cpp);
the two are cross-checked in tests/.
"""
from __future__ import annotations

import hashlib
import struct
from dataclasses import dataclass, field

R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
R_MONT_INV = pow(1 << 256, -1, R_MOD)

TAG_HINT = 5309735
TAG_R1C = 5309736
TAG_COMMIT = 5309737
TAG_LOOKUP = 5309748

HINT_NBITS = 4115454955          # bits.nBits
HINT_COUNT = 2138922168          # logderivarg.countHint
HINT_RANDOMIZE = 1774611027      # hints.Randomize
HINT_BSB22 = 4156202267          # cs.Bsb22CommitmentComputePlaceholder

CIRCUIT_SHA256 = {  # gnark/libraries/prover/impl/library.go:38,43,48
    "chacha20": "517cdf7642c84dcb165a29f0accea413178fc11b21d8998461156a1acce6ebbd",
    "aes128": "8ef58f13706568eed44f9aa40b97da7c2f2a7f1bce47b7bfe52f7df860a049fb",
    "aes256": "7f589a6d1a24efa6d01f663c11b671e0abef196c15cadb93d2c6b226df7bf0dd",
}


# ----------------------------------------------------------------------------- CBOR
class Tagged:
    __slots__ = ("tag", "value")

    def __init__(self, tag, value):
        self.tag, self.value = tag, value

    def __repr__(self):
        return f"Tagged({self.tag}, {self.value!r})"


def cbor_decode(buf: bytes, pos: int = 0):
    """Minimal CBOR (RFC 8949) decoder: ints, bytes, text, arrays, maps, tags, simple."""
    ib = buf[pos]
    pos += 1
    major, info = ib >> 5, ib & 31
    if info < 24:
        val = info
    elif info == 24:
        val = buf[pos]; pos += 1
    elif info == 25:
        val = struct.unpack_from(">H", buf, pos)[0]; pos += 2
    elif info == 26:
        val = struct.unpack_from(">I", buf, pos)[0]; pos += 4
    elif info == 27:
        val = struct.unpack_from(">Q", buf, pos)[0]; pos += 8
    elif info == 31:
        val = None  # indefinite
    else:
        raise ValueError("bad cbor info")
    if major == 0:
        return val, pos
    if major == 1:
        return -1 - val, pos
    if major in (2, 3):
        if val is None:
            chunks = []
            while buf[pos] != 0xFF:
                c, pos = cbor_decode(buf, pos)
                chunks.append(c)
            pos += 1
            return (b"".join(chunks) if major == 2 else "".join(chunks)), pos
        raw = bytes(buf[pos:pos + val]); pos += val
        return (raw if major == 2 else raw.decode("utf-8")), pos
    if major == 4:
        out = []
        if val is None:
            while buf[pos] != 0xFF:
                v, pos = cbor_decode(buf, pos); out.append(v)
            return out, pos + 1
        for _ in range(val):
            v, pos = cbor_decode(buf, pos); out.append(v)
        return out, pos
    if major == 5:
        out = {}
        if val is None:
            while buf[pos] != 0xFF:
                k, pos = cbor_decode(buf, pos); v, pos = cbor_decode(buf, pos); out[k] = v
            return out, pos + 1
        for _ in range(val):
            k, pos = cbor_decode(buf, pos); v, pos = cbor_decode(buf, pos)
            out[k if not isinstance(k, list) else tuple(k)] = v
        return out, pos
    if major == 6:
        v, pos = cbor_decode(buf, pos)
        return Tagged(val, v), pos
    # major 7
    if info == 20:
        return False, pos
    if info == 21:
        return True, pos
    if info in (22, 23):
        return None, pos
    if info == 25:
        return struct.unpack(">e", struct.pack(">H", val))[0], pos
    if info == 26:
        return struct.unpack(">f", struct.pack(">I", val))[0], pos
    if info == 27:
        return struct.unpack(">d", struct.pack(">Q", val))[0], pos
    return val, pos


# -------------------------------------------------------------------------- intcomp
def intcomp_decode(words, W: int):
    """Decode one ronanh/intcomp stream as stored by gnark.

    words: list of W-bit unsigned ints (includes the 1-word trailer)."""
    mask = (1 << W) - 1
    out = []
    pos = 0
    body = len(words) - 1
    if body <= 0:
        return out
    n0 = words[0] & 0xFFFFFFFF
    if n0 != 0 and n0 % (4 * W) == 0:
        if W == 32:
            n_ints, n_words, first, p = words[0], words[1], words[2], 3
        else:
            n_ints, n_words, first, p = words[0] & 0xFFFFFFFF, words[0] >> 32, words[1], 2
        prev = first
        end = n_words
        while p < end:
            h = words[p] & 0xFFFFFFFF
            p += 1
            for sign, b in (((h >> 31) & 1, (h >> 24) & 127), ((h >> 23) & 1, (h >> 16) & 127),
                            ((h >> 15) & 1, (h >> 8) & 127), ((h >> 7) & 1, h & 127)):
                big = 0
                for k in range(b):
                    big |= words[p + k] << (W * k)
                p += b
                bm = (1 << b) - 1
                for k in range(W):
                    d = (big >> (b * k)) & bm
                    if sign:
                        d = (d >> 1) ^ -(d & 1)
                    prev = (prev + d) & mask
                    out.append(prev)
        assert len(out) == n_ints, (len(out), n_ints)
        pos = end
    if pos < body:
        if W == 32:
            n_ints, n_words, q = words[pos], words[pos + 1], pos + 2
        else:
            n_ints, n_words, q = words[pos] & 0xFFFFFFFF, words[pos] >> 32, pos + 1
        bs = bytearray()
        for w in words[q:pos + n_words]:
            bs += w.to_bytes(W // 8, "big")
        prev = 0
        bp = 0
        for _ in range(n_ints):
            v = 0
            shift = 0
            while True:
                c = bs[bp]; bp += 1
                v |= (c & 0x7F) << shift
                shift += 7
                if not (c & 0x80):
                    break
            prev = (prev + v) & mask
            out.append(prev)
    return out


def _read_stream(buf, pos, W):
    n = struct.unpack_from("<Q", buf, pos)[0]
    pos += 8
    fmt = "<%d%s" % (n, "I" if W == 32 else "Q")
    words = list(struct.unpack_from(fmt, buf, pos))
    pos += n * (W // 8)
    return intcomp_decode(words, W), pos


# ------------------------------------------------------------------------ container
@dataclass
class R1CS:
    name: str
    nb_constraints: int
    nb_internal: int
    public: list
    secret: list
    blueprints: list          # list of ("hint"|"r1c"|"lookup", payload)
    commitment_info: list     # list of dicts
    hint_names: dict
    levels: list              # list of lists of instruction ids
    blueprint_id: list
    constraint_offset: list
    wire_offset: list
    start_calldata: list
    calldata: list            # list of u32
    coeffs: list              # canonical ints (NOT Montgomery)
    coeffs_mont_raw: bytes = b""
    body: dict = field(default_factory=dict)

    @property
    def nb_public(self):
        return len(self.public)

    @property
    def nb_secret(self):
        return len(self.secret)

    @property
    def nb_wires(self):
        return len(self.public) + len(self.secret) + self.nb_internal

    @property
    def nb_instructions(self):
        return len(self.blueprint_id)

    def domain_size(self):
        n = 1
        while n < self.nb_constraints:
            n <<= 1
        return n


def parse_r1cs(data: bytes, name: str = "") -> R1CS:
    total_len, major, minor, patch = struct.unpack_from("<4Q", data, 0)
    assert total_len + 32 == len(data), "length prefix mismatch"
    assert (major, minor) == (0, 14), "unsupported gnark version %d.%d.%d" % (major, minor, patch)
    levels_len, instr_len, calldata_len, body_len = struct.unpack_from("<4Q", data, 32)
    pos = 64
    # levels
    end = pos + levels_len
    n_levels = struct.unpack_from("<Q", data, pos)[0]
    pos += 8
    levels = []
    for _ in range(n_levels):
        lvl, pos = _read_stream(data, pos, 32)
        levels.append(lvl)
    assert pos == end, ("levels", pos, end)
    # instructions
    end = pos + instr_len
    bp_id, pos = _read_stream(data, pos, 32)
    c_off, pos = _read_stream(data, pos, 32)
    w_off, pos = _read_stream(data, pos, 32)
    start_cd, pos = _read_stream(data, pos, 64)
    assert pos == end, ("instr", pos, end)
    # calldata
    end = pos + calldata_len
    n_cd = struct.unpack_from("<Q", data, pos)[0]
    pos += 8
    calldata = []
    mv = data
    for _ in range(n_cd):
        v = 0
        shift = 0
        while True:
            c = mv[pos]; pos += 1
            v |= (c & 0x7F) << shift
            shift += 7
            if not (c & 0x80):
                break
        calldata.append(v)
    assert pos == end, ("calldata", pos, end)
    # body
    body, bend = cbor_decode(data, pos)
    assert bend == pos + body_len, ("body", bend, pos + body_len)
    pos += body_len
    # coefficient table
    n_coeff = struct.unpack_from("<Q", data, pos)[0]
    pos += 8
    raw = bytes(data[pos:pos + 32 * n_coeff])
    coeffs = []
    for i in range(n_coeff):
        limbs = struct.unpack_from("<4Q", raw, 32 * i)
        m = limbs[0] | (limbs[1] << 64) | (limbs[2] << 128) | (limbs[3] << 192)
        coeffs.append(m * R_MONT_INV % R_MOD)
    pos += 32 * n_coeff
    assert pos == len(data), ("tail", pos, len(data))

    blueprints = []
    for bp in body["Blueprints"]:
        assert isinstance(bp, Tagged), bp
        if bp.tag == TAG_HINT:
            blueprints.append(("hint", None))
        elif bp.tag == TAG_R1C:
            blueprints.append(("r1c", None))
        elif bp.tag == TAG_LOOKUP:
            blueprints.append(("lookup", list(bp.value["EntriesCalldata"])))
        else:
            raise ValueError("unknown blueprint tag %d" % bp.tag)
    ci = body.get("CommitmentInfo")
    commitment_info = []
    if ci is not None:
        v = ci.value if isinstance(ci, Tagged) else ci
        for c in (v or []):
            commitment_info.append({
                "CommitmentIndex": c["CommitmentIndex"],
                "PrivateCommitted": list(c["PrivateCommitted"] or []),
                "NbPublicCommitted": c["NbPublicCommitted"],
                "PublicAndCommitmentCommitted": list(c["PublicAndCommitmentCommitted"] or []),
            })
    return R1CS(
        name=name,
        nb_constraints=body["NbConstraints"],
        nb_internal=body["NbInternalVariables"],
        public=list(body["Public"]),
        secret=list(body["Secret"]),
        blueprints=blueprints,
        commitment_info=commitment_info,
        hint_names={int(k): v for k, v in (body.get("MHintsDependencies") or {}).items()},
        levels=levels,
        blueprint_id=bp_id,
        constraint_offset=c_off,
        wire_offset=w_off,
        start_calldata=start_cd,
        calldata=calldata,
        coeffs=coeffs,
        coeffs_mont_raw=raw,
        body=body,
    )


def load_reference_r1cs(name: str, root: str = "/root/reference/resources/gnark") -> R1CS:
    import lzma
    import os
    p = os.path.join(root, "r1cs." + name)
    if os.path.exists(p):
        data = open(p, "rb").read()
    else:  # GPU box: xz fixtures committed under tests/golden
        here = os.path.dirname(os.path.abspath(__file__))
        data = lzma.open(os.path.join(here, "..", "tests", "golden", "r1cs.%s.xz" % name)).read()
    assert hashlib.sha256(data).hexdigest() == CIRCUIT_SHA256[name]
    return parse_r1cs(data, name)


def load_r1cs_bytes(name: str) -> bytes:
    import lzma
    import os
    p = "/root/reference/resources/gnark/r1cs." + name
    if os.path.exists(p):
        return open(p, "rb").read()
    here = os.path.dirname(os.path.abspath(__file__))
    return lzma.open(os.path.join(here, "..", "tests", "golden", "r1cs.%s.xz" % name)).read()
