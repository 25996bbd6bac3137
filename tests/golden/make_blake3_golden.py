#!/usr/bin/env python3
"""Golden vectors for the BLAKE3 restatements (oracle/blake3.hpp, csrc/blake3.cuh), made with an INDEPENDENT implementation:
the BLAKE3 team's C code as shipped inside LLVM (llvm_blake3_hasher_* in libLLVM-15, present in this image).  Inputs follow
the official test_vectors.json: byte i = i mod 251, at its input lengths (block, chunk and tree boundaries) plus a few more.
The three digests everybody knows by heart (empty input, "abc", one zero byte) are asserted first.

    python tests/golden/make_blake3_golden.py   ->  tests/golden/blake3.json
"""
import ctypes as C, json, os

L = C.CDLL("/usr/lib/x86_64-linux-gnu/libLLVM-15.so.1")


def b3(data):
    st = C.create_string_buffer(4096)  # sizeof(llvm_blake3_hasher) = 1912
    L.llvm_blake3_hasher_init(st)
    L.llvm_blake3_hasher_update(st, data, C.c_size_t(len(data)))
    out = C.create_string_buffer(32)
    L.llvm_blake3_hasher_finalize(st, out, C.c_size_t(32))
    return out.raw.hex()


assert b3(b"") == "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262"
assert b3(b"abc") == "6437b3ac38465133ffb63b75273a8db548c558465d79db03fd359c6cd5bd9d85"
assert b3(b"\x00") == "2d3adedff11b61f14c886e35afa036736dcd87a74d27b5c1510225d0f592e213"
LENGTHS = [0, 1, 2, 3, 4, 5, 6, 7, 8, 63, 64, 65, 127, 128, 129, 1023, 1024, 1025, 2048, 2049, 3072, 3073, 4096, 4097, 5120, 5121,
           6144, 6145, 7168, 7169, 8192, 8193, 16384, 31744, 102400,
           40, 72, 104, 440, 472, 1056]  # 32-byte state + 1 / 5 / 9 / 51 / 55 / 128 felts: the LMCS leaf message sizes
out = {"pattern": "byte i = i mod 251", "source": "llvm_blake3_hasher (LLVM 15's copy of the official C implementation)",
       "cases": [{"len": n, "hash": b3(bytes(i % 251 for i in range(n)))} for n in LENGTHS],
       "abc": b3(b"abc")}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "blake3.json")
json.dump(out, open(path, "w"), indent=1)
print("wrote", path, len(LENGTHS), "cases")
