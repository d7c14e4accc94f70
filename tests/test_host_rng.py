"""CPU-only: the random-number stack under the host drivers.
 - Blake2b-256 seed hashing (product C++) vs hashlib (RFC 7693 implementation) and the
   digests recorded in SURVEY.md 8(c);
 - ChaCha20 block function (product C++ and oracle Python) vs the RFC 7539 2.3.2 vector;
 - every rand-0.8.5 sampling routine: product C++ (ls_host_rng_script) vs the oracle's
   independent Python restatement, on random scripts.
The rand / rand_chacha crates are not in /root/reference: PARITY UNPINNED against the Rust
binaries' streams."""
import hashlib
import os
import random

import pytest
import struct

import numpy as np

import ls_amd
import ls_oracle as lo


def test_blake2b_seed_hash():
    for s in ["42", "43", "", "a", "x" * 127, "y" * 128, "z" * 129, "seed " * 100]:
        assert ls_amd.host_hash_seed(s) == hashlib.blake2b(s.encode(), digest_size=32).digest() == lo.hash_str(s)
    # SURVEY.md 8(c)
    assert ls_amd.host_hash_seed("42").hex() == "2f0039e93a27221fcf657fb877a1d4f60307106113e885096cb44a461cd0afbf"
    assert ls_amd.host_hash_seed("43").hex() == "947df21e471af47ec8d88b09c748a07c8abbbcc2ebc27d8fabc94efc815e3cc1"


def test_chacha20_block_rfc7539():
    # RFC 7539 section 2.3.2: key 00..1f, counter 1, nonce 00:00:00:09:00:00:00:4a:00:00:00:00
    key = [int.from_bytes(bytes(range(4 * i, 4 * i + 4)), "little") for i in range(8)]
    expect = [0xe4e7f110, 0x15593bd1, 0x1fdd0f50, 0xc47120a3, 0xc7f4d1c7, 0x0368c033, 0x9aaa2204, 0x4e6cd4c3,
              0x466482d2, 0x09aa9f07, 0x05d7c214, 0xa2028bd9, 0xd19c12b5, 0xb94e16de, 0xe883d0cb, 0x4e3c50a2]
    assert ls_amd.host_chacha20_block(key, 1, 0x09000000, 0x4a000000, 0) == expect
    assert lo.chacha20_block(key, 1, 0x09000000, 0x4a000000, 0) == expect


def test_rng_stream_layout():
    # first refill = blocks with counters 0..3 back to back; next_u64 = two words, low first
    seed = hashlib.blake2b(b"42", digest_size=32).digest()
    key = [int.from_bytes(seed[4 * i:4 * i + 4], "little") for i in range(8)]
    words = ls_amd.host_rng_script(seed, [(0, 0, 0)] * 130)
    exp = []
    for ctr in range(9):
        exp += lo.chacha20_block(key, ctr, 0, 0, 0)
    assert words == exp[:130]
    u64 = ls_amd.host_rng_script(seed, [(1, 0, 0)] * 40)
    assert u64 == [exp[2 * i] | (exp[2 * i + 1] << 32) for i in range(40)]
    # straddling a refill: 63 u32 draws leave one word; the next u64 takes it as the LOW half
    mixed = ls_amd.host_rng_script(seed, [(0, 0, 0)] * 63 + [(1, 0, 0), (0, 0, 0)])
    assert mixed[63] == exp[63] | (exp[64] << 32) and mixed[64] == exp[65]


def test_chacha20rng_word_stream_against_openssl():
    # An implementation neither the product nor the oracle shares: the OpenSSL command-line tool.  Encrypting zeros with
    # ChaCha20 under the seed as key, counter 0 and a zero nonce yields the keystream; ChaCha20Rng::from_seed (64-bit block
    # counter 0, stream id 0: the same initial state) must hand out exactly its little-endian 32-bit words, in order, across
    # several four-block refills -- for the seeds the examples derive from "42", "43", ... (examples/nqueens/src/main.rs:28-39).
    # Pins the generator core (block function, word order, block sequencing); rand's sampling algorithms on top of it stay
    # "parity unpinned".
    import shutil
    import subprocess
    if shutil.which("openssl") is None:
        pytest.skip("no openssl command-line tool")
    for seed_str in ("42", "43", "105", ""):
        seed = hashlib.blake2b(seed_str.encode(), digest_size=32).digest()
        assert ls_amd.host_hash_seed(seed_str) == seed
        n_words = 16 * 21 + 5                                    # five full refills (64 words each) and a bit
        r = subprocess.run(["openssl", "enc", "-chacha20", "-K", seed.hex(), "-iv", bytes(16).hex(), "-nosalt", "-nopad"],
                           input=bytes(4 * n_words + 60), capture_output=True, timeout=60)
        if r.returncode != 0:
            pytest.skip("this openssl has no chacha20 stream cipher: " + r.stderr.decode()[:200])
        ks = r.stdout
        exp = [int.from_bytes(ks[4 * i:4 * i + 4], "little") for i in range(n_words)]
        assert ls_amd.host_rng_script(seed, [(0, 0, 0)] * n_words) == exp          # the product's host generator
        rng = lo.ChaCha20Rng(seed)                                                 # the oracle's restatement
        assert [rng.next_u32() for _ in range(n_words)] == exp
        u64 = ls_amd.host_rng_script(seed, [(1, 0, 0)] * (n_words // 2))
        assert u64 == [exp[2 * i] | (exp[2 * i + 1] << 32) for i in range(n_words // 2)]


def _oracle_script(seed, ops, weights):
    rng = lo.ChaCha20Rng(seed)
    out = []
    for op, a, b in ops:
        if op == 0:
            out.append(rng.next_u32())
        elif op == 1:
            out.append(rng.next_u64())
        elif op == 2:
            out.append(lo.gen_range_incl(rng, a, b, 64))
        elif op == 3:
            out.append(lo.gen_range_incl(rng, a, b, 32))
        elif op == 4:
            out.append(lo.uniform_below(rng, a, 64))
        elif op == 5:
            out.append(lo.uniform_below(rng, a, 32))
        elif op == 6:
            v = list(range(a))
            lo.shuffle(rng, v)
            out += v
        elif op == 7:
            out += lo.index_sample(rng, a, b)
        elif op == 8:
            out.append(struct.unpack("<Q", struct.pack("<d", rng.gen_f64()))[0])
        elif op == 9:
            out += lo.index_sample_weighted(rng, list(weights[:a]), b)
    return out


def test_sampling_routines_match_oracle():
    pyr = random.Random(7)
    weights = [pyr.randint(1, 40) + 1e-4 for _ in range(300)]
    for trial in range(30):
        seed = bytes(pyr.randrange(256) for _ in range(32))
        ops = []
        for _ in range(40):
            op = pyr.choice([0, 1, 2, 3, 4, 5, 6, 7, 8, 9])
            if op == 2:
                lo_, hi_ = sorted((pyr.randrange(1 << 40), pyr.randrange(1 << 40)))
                ops.append((2, lo_, hi_))
            elif op == 3:
                lo_, hi_ = sorted((pyr.randrange(1 << 31), pyr.randrange(1 << 31)))
                ops.append((3, lo_, hi_))
            elif op == 4:
                ops.append((4, pyr.choice([1, 2, 5, 110, 1 << 33, (1 << 62) + 12345]), 0))
            elif op == 5:
                ops.append((5, pyr.choice([1, 2, 3, 6, 7, 1 << 20, (1 << 31) - 1]), 0))
            elif op == 6:
                ops.append((6, pyr.choice([1, 2, 8, 31, 100]), 0))
            elif op == 7:
                # covers floyd (amount < 12 and sparse), floyd+shuffle (>= 50), inplace (dense), rejection (>= 163 sparse)
                length, amount = pyr.choice([(31, 2), (8, 1), (204, 7), (51, 30), (300, 60), (4096, 12), (1000, 200),
                                             (200000, 170), (600000, 20), (5, 5)])
                ops.append((7, length, amount))
            elif op == 9:
                a = pyr.choice([1, 5, 51, 300])
                ops.append((9, a, pyr.randint(1, min(a, 15))))
            else:
                ops.append((op, 0, 0))
        got = ls_amd.host_rng_script(seed, ops, weights, cap=1 << 18)
        assert got == _oracle_script(seed, ops, weights), (trial, ops)


def test_weighted_sampler_prefilter_shape_matches_oracle():
    # the N-queens proposer's shape at n = 4096 (about 1100 conflicting columns, 204 picks): the product's sampler brackets
    # every key with a cheap log2, evaluates pow() only for the keys that can reach the top 204 and selects by histogram;
    # the oracle's Python restatement computes every key and sorts.  Same picks, same order, same generator state after.
    pyr = random.Random(99)
    for trial in range(6):
        L = pyr.choice([64, 400, 1100, 1137, 2000])
        weights = [pyr.randint(1, 9) + 1e-4 for _ in range(L)]
        seed = bytes(pyr.randrange(256) for _ in range(32))
        ops = [(9, L, max(1, L // pyr.choice([2, 5, 20]))), (1, 0, 0), (9, L, 1), (0, 0, 0)]
        got = ls_amd.host_rng_script(seed, ops, weights, cap=1 << 18)
        assert got == _oracle_script(seed, ops, weights), (trial, L)


def test_gen_range_is_in_range_and_covers():
    seed = bytes(range(32))
    vals = ls_amd.host_rng_script(seed, [(2, 3, 9)] * 2000)
    assert min(vals) == 3 and max(vals) == 9
    vals = ls_amd.host_rng_script(seed, [(7, 10, 10)])
    assert sorted(vals) == list(range(10))


def test_rng_stream_without_avx2_lookahead():
    # the refill has three code paths (AVX-512: sixteen blocks at once with a look-ahead buffer; AVX2: eight; otherwise
    # four): same words, including u64 draws that straddle a refill (the odd u32 draw shifts the phase)
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import hashlib, ls_amd;"
            "seed = hashlib.blake2b(b'42', digest_size=32).digest();"
            "print(ls_amd.host_rng_script(seed, ([(0, 0, 0)] * 300 + [(1, 0, 0)] * 100 + [(0, 0, 0)]) * 5))") % os.path.dirname(ls_amd.__path__[0])
    outs = []
    for extra in ({}, {"LS_HOST_NO_AVX512": "1"}, {"LS_HOST_NO_AVX2": "1"}):
        env = dict(os.environ, **extra)
        outs.append(subprocess.run([sys.executable, "-c", code], env=env, check=True, capture_output=True, text=True,
                                   timeout=120).stdout)
    assert outs[0] == outs[1] == outs[2] and len(outs[0]) > 1000
