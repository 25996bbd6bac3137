"""A generator of random programs for the TEST VM (miden-vm_amd/testing/core_trace.py) -- test infrastructure.  Programs are nests of JOIN / SPLIT /
LOOP over basic blocks made of stack-neutral GADGETS: a gadget pushes a few values (constants, copies of whatever is on the stack), runs random
operations over its own values only -- field, boolean, u32 (operands typed u32 by construction: constants, U32SPLIT limbs, results of u32
operations), memory, the hasher, conditional swaps, word swaps, EXT2MUL / EXPACC -- and drops what is left.  Preconditions the processor would
turn into an execution error (NOT / AND / OR on a non-binary value, u32 operations on a large element, division by zero, INV of zero, a failing
ASSERT) are avoided by tracking a type per value, so every program runs to HALT; the stack goes above sixteen (the overflow table) and blocks
span several operation batches with immediates."""
import numpy as np
from miden_vm_amd.testing import core_trace as CV

P = 0xFFFFFFFF00000001


def _typ(c):
    return "b" if c < 2 else "u" if c < 1 << 32 else "f"


def gadget(rng, budget=24):
    ops, reg = [], []          # reg: types of this gadget's own values, top first

    def push_const():
        k = int(rng.integers(0, 5))
        c = int([rng.integers(0, 2), rng.integers(0, 1 << 16), rng.integers(0, 1 << 32), rng.integers(0, P, dtype=np.uint64), (1 << 32) - 1][k])
        ops.append(("PUSH", c))
        reg.insert(0, _typ(c))

    for _ in range(int(rng.integers(1, 5))):
        push_const()
    for _ in range(int(rng.integers(1, budget))):
        r = len(reg)
        k = int(rng.integers(0, 30))
        if r >= 20:
            break
        if k == 0 or r < 2:
            push_const()
        elif k == 1:
            ops.append("PAD"); reg.insert(0, "b")
        elif k == 2:
            d = int(rng.choice([0, 1, 2, 3, 4, 5, 6, 7, 9, 11, 13, 15]))
            ops.append(f"DUP{d}"); reg.insert(0, reg[d] if d < r else "f")
        elif k == 3:
            ops.append("SWAP"); reg[0], reg[1] = reg[1], reg[0]
        elif k == 4 and r >= 3:
            d = int(rng.integers(2, min(r, 9)))
            ops.append(f"MOVUP{d}"); reg.insert(0, reg.pop(d))
        elif k == 5 and r >= 3:
            d = int(rng.integers(2, min(r, 9)))
            ops.append(f"MOVDN{d}"); reg.insert(d, reg.pop(0))
        elif k in (6, 7):
            ops.append("ADD" if k == 6 else "MUL"); reg[0:2] = ["f"]
        elif k == 8:
            ops.append(str(rng.choice(["NEG", "INCR"]))); reg[0] = "f"
        elif k == 9:
            c = int(rng.integers(1, P, dtype=np.uint64))
            ops += [("PUSH", c), "INV"]; reg.insert(0, "f")
        elif k == 10:
            ops.append("EQ"); reg[0:2] = ["b"]
        elif k == 11:
            ops.append("EQZ"); reg[0] = "b"
        elif k == 12 and reg[0] == "b":
            ops.append("NOT")
        elif k == 13 and reg[0] == "b" and reg[1] == "b":
            ops.append(str(rng.choice(["AND", "OR"]))); reg[0:2] = ["b"]
        elif k == 14:
            ops.append("U32SPLIT"); reg[0:1] = ["u", "u"]
        elif k in (15, 16, 17) and reg[0] in "bu" and reg[1] in "bu":
            name = str(rng.choice(["U32ADD", "U32SUB", "U32MUL", "U32ASSERT2", "U32AND", "U32XOR"]))
            ops.append(name)
            if name in ("U32AND", "U32XOR"):
                reg[0:2] = ["u"]
            elif name != "U32ASSERT2":
                reg[0:2] = ["u", "u"]
        elif k == 18 and r >= 3 and all(t in "bu" for t in reg[:3]):
            ops.append(str(rng.choice(["U32ADD3", "U32MADD"]))); reg[0:3] = ["u", "u"]
        elif k == 19 and reg[0] in "bu":
            ops += [("PUSH", int(rng.integers(1, 1 << 32))), "U32DIV"]; reg[0:1] = ["u", "u"]
        elif k == 20:
            addr = int(rng.choice([0, 1, 5, 100, 101, 4096, (1 << 20) + 3, (1 << 32) - 1]))
            ops += [("PUSH", addr), "MSTORE"]
        elif k == 21:
            addr = int(rng.choice([0, 1, 5, 100, 101, 4096, (1 << 20) + 3, 77]))
            ops += [("PUSH", addr), "MLOAD"]; reg.insert(0, "f")
        elif k == 22 and r >= 4:
            ops += [("PUSH", int(rng.choice([0, 4, 200, 4096, (1 << 20) + 4]))), "MSTOREW"]
        elif k == 23 and r >= 4:
            ops += [("PUSH", int(rng.choice([0, 4, 200, 4096, 1 << 24]))), "MLOADW"]; reg[0:4] = ["f"] * 4
        elif k == 24 and r + 12 <= 20:
            n = max(0, 12 - r)
            ops += ["PAD"] * n + ["HPERM"]; reg[0:0] = ["b"] * n; reg[0:12] = ["f"] * 12
        elif k == 25 and r >= 2:
            ops += [("PUSH", int(rng.integers(0, 2))), "CSWAP"]; reg[0], reg[1] = "f" if reg[0] != reg[1] else reg[0], "f" if reg[0] != reg[1] else reg[1]
        elif k == 26:
            ops += [("PUSH", 1), "ASSERT"]
        elif k == 27:
            ops.append(str(rng.choice(["CLK", "SDEPTH"]))); reg.insert(0, "f")
        elif k == 28 and r >= 4:
            ops.append("EXT2MUL"); reg[0:4] = ["f"] * 4
        elif k == 29 and r >= 8:
            ops.append("SWAPW"); reg[0:8] = reg[4:8] + reg[0:4]
    ops += ["DROP"] * len(reg)
    return ops


def block(rng, max_gadgets=5):
    ops = []
    for _ in range(int(rng.integers(1, max_gadgets + 1))):
        ops += gadget(rng)
    return CV.Span(ops or ["NOOP"])


def node(rng, depth):
    if depth == 0 or rng.random() < 0.35:
        return block(rng)
    k = int(rng.integers(0, 4))
    if k <= 1:
        return CV.Join(node(rng, depth - 1), node(rng, depth - 1))
    if k == 2:       # SPLIT on a pushed condition: either branch is stack-neutral
        return CV.Join(CV.Span(gadget(rng) + [("PUSH", int(rng.integers(0, 2)))]), CV.Split(node(rng, depth - 1), node(rng, depth - 1)))
    iters = int(rng.integers(1, 5))   # LOOP: a counter on the stack, neutral gadgets in the body, `iters` iterations
    body = CV.Span(gadget(rng, 10) + [("PUSH", 1), "ADD", "DUP0", ("PUSH", iters), "EQ", "NOT"])
    return CV.Join(CV.Span([("PUSH", 0)]), CV.Join(CV.Loop(body), CV.Span(["DROP"])))


def random_program(seed, depth=3):
    return node(np.random.default_rng(0x9406 + seed), depth)
