"""Fiat-Shamir RNG and random-number plumbing of the reference, restated (oracle only).

* ChaCha block function + BlockRng word order: rand_chacha 0.3 (`ChaChaRng` = ChaCha20,
  dev-dependency Cargo.toml:36, third-party, absent) [UPSTREAM-RECALLED B-6/B-7].
* SimpleHashFiatShamirRng<Blake2s, ChaChaRng>: /root/reference src/rng.rs:54-79 (in-repo, exact).
* Fp256::rand rejection sampling and u128 -> F conversion: ark-ff 0.3 [UPSTREAM-RECALLED B-7].
Blake2s comes from hashlib (RFC 7693, 32-byte digest == blake2 0.9 `Blake2s`).
"""
import hashlib
import struct
from .fields import R_MOD, FR_MONT_RINV, FR_REPR_SHAVE_BITS

MASK32 = 0xFFFFFFFF


def _rotl(x, n):
    return ((x << n) & MASK32) | (x >> (32 - n))


def _qr(s, a, b, c, d):
    s[a] = (s[a] + s[b]) & MASK32; s[d] = _rotl(s[d] ^ s[a], 16)
    s[c] = (s[c] + s[d]) & MASK32; s[b] = _rotl(s[b] ^ s[c], 12)
    s[a] = (s[a] + s[b]) & MASK32; s[d] = _rotl(s[d] ^ s[a], 8)
    s[c] = (s[c] + s[d]) & MASK32; s[b] = _rotl(s[b] ^ s[c], 7)


def chacha_block(key_words, counter, stream, rounds):
    st = [0x61707865, 0x3320646e, 0x79622d32, 0x6b206574] + list(key_words) + [
        counter & MASK32, (counter >> 32) & MASK32, stream & MASK32, (stream >> 32) & MASK32]
    w = list(st)
    for _ in range(rounds // 2):
        _qr(w, 0, 4, 8, 12); _qr(w, 1, 5, 9, 13); _qr(w, 2, 6, 10, 14); _qr(w, 3, 7, 11, 15)
        _qr(w, 0, 5, 10, 15); _qr(w, 1, 6, 11, 12); _qr(w, 2, 7, 8, 13); _qr(w, 3, 4, 9, 14)
    return [(w[i] + st[i]) & MASK32 for i in range(16)]


class ChaChaRng:
    """rand_chacha::ChaCha{8,12,20}Rng::from_seed: 64-bit block counter from 0, stream 0,
    64-word (4-block) buffer consumed as little-endian u32 words; next_u64 = lo | hi << 32."""

    def __init__(self, seed32, rounds=20):
        assert len(seed32) == 32
        self.key = list(struct.unpack("<8I", bytes(seed32)))
        self.rounds = rounds
        self.counter = 0
        self.buf = []
        self.idx = 64

    def _refill(self):
        self.buf = []
        for _ in range(4):
            self.buf += chacha_block(self.key, self.counter, 0, self.rounds)
            self.counter += 1
        self.idx = 0

    def next_u32(self):
        if self.idx >= 64:
            self._refill()
        v = self.buf[self.idx]
        self.idx += 1
        return v

    def next_u64(self):
        # BlockRng::next_u64: aligned reads only occur here (all draws are u64 pairs)
        if self.idx >= 63:
            if self.idx == 63:
                lo = self.buf[63]
                self._refill()
                hi = self.buf[0]
                self.idx = 1
                return lo | (hi << 32)
            self._refill()
        lo, hi = self.buf[self.idx], self.buf[self.idx + 1]
        self.idx += 2
        return lo | (hi << 32)


def test_rng():
    """ark_std::test_rng(): StdRng (ChaCha12 in rand 0.8) from a fixed seed [B-7]."""
    seed = bytes([1, 0, 0, 0, 23, 0, 0, 0, 200, 1, 0, 0, 210, 30, 0, 0] + [0] * 16)
    return ChaChaRng(seed, rounds=12)


def fr_rand(rng):
    """Fp256::<FrParameters>::rand: 4 x next_u64, clear the top REPR_SHAVE_BITS = 1 bit, accept if
    < r; the accepted limbs ARE the Montgomery representation.  Returns the canonical value."""
    while True:
        limbs = [rng.next_u64() for _ in range(4)]
        limbs[3] &= (1 << (64 - FR_REPR_SHAVE_BITS)) - 1
        x = limbs[0] | (limbs[1] << 64) | (limbs[2] << 128) | (limbs[3] << 192)
        if x < R_MOD:
            return x * FR_MONT_RINV % R_MOD


def fr_bytes(x):
    """ToBytes for Fr: canonical value, 32 bytes little-endian."""
    return int(x % R_MOD).to_bytes(32, "little")


class SimpleHashFiatShamirRng:
    """SimpleHashFiatShamirRng<Blake2s, ChaChaRng> (src/rng.rs:18-80)."""

    def __init__(self, initial_bytes):
        self.seed = hashlib.blake2s(bytes(initial_bytes)).digest()     # rng.rs:54-66
        self.r = ChaChaRng(self.seed, 20)

    def absorb(self, new_bytes):                                       # rng.rs:71-79
        self.seed = hashlib.blake2s(bytes(new_bytes) + self.seed).digest()
        self.r = ChaChaRng(self.seed, 20)

    def next_u64(self):
        return self.r.next_u64()

    def rand_fr(self):
        return fr_rand(self.r)

    def rand_u128_as_fr(self):
        """`u128::rand(&mut fs_rng).into()` (src/lib.rs:290): lo = next_u64, hi = next_u64."""
        lo = self.r.next_u64()
        hi = self.r.next_u64()
        return (lo | (hi << 64)) % R_MOD
