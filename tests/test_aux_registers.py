"""Aux REGISTER columns: extension-field accumulators that live in the aux trace (they depend on the challenges) next to the LogUp columns
and stay out of sigma -- the layout of the precompile prover's uint chiplets (precompiles-prover/src/uint/store_mul/mod.rs:118-121:
STORE_REG_ID, MUL_REG_ID, MUL_REG_S).  The reference computes them in each AIR's own `build_aux_trace`; here the recurrence
r[0] = 0, r[i + 1] = keep(i) r[i] + sum_j coeff_j(i) r_j[i] + build(i) is data of the lookup program ("MHLKP001" register tail), built by
the oracle (oracle/lookup.hpp) and by the device (csrc/logup.hip: a scan over affine maps).  Host only; device parity in
tests/test_gpu_lookup.py.

  ext_register_verifies_and_stays_out_of_sigma      the reference's own spike (precompiles-prover/src/tests/aux_register.rs) replayed: one
                                                    empty LogUp column, one Horner register acc' = acc beta + x; sigma = 0, every
                                                    constraint holds, the register is what Horner says
  a chain of registers with a live LogUp column     the multiplier's shape (uint/store_mul/mod.rs:304-440): S' = keep S + build with a
                                                    periodic keep, id' = id + S U + V reading S; against a plain Python evaluation
  proofs                                            oracle proofs of both AIRs verify with both verifiers; a forged register cell does not"""
import numpy as np
import pytest
import oracle_binding as ob
from __graft_entry__ import load_package

pkg = load_package()
from miden_vm_amd import dag, protocol  # noqa: E402

P = dag.P
FAST = dict(log_blowup=3, log_folding_arity=2, log_final_degree=2, folding_pow_bits=1, deep_pow_bits=2, num_queries=5, query_pow_bits=3)
RND = [(0x1234567890abcdef % P, 0x0fedcba987654321), (3141592653589793, 2718281828459045)]
MAX_MESSAGE_WIDTH, NUM_BUS_IDS = 18, 21                                 # precompiles-prover/src/relations.rs


def e_mul(a, b):
    return ((a[0] * b[0] + 7 * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def e_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def _air(b, lookup, name):
    def build(main, rnd):
        aux, fin = ob.lookup_build_aux(lookup, main, rnd, None)
        return aux, [int(fin[0]), int(fin[1])]
    return dag.Air(b, build, name)


def spike_air():
    """`SpikeAir` (tests/aux_register.rs:36-146): main = [x]; aux col 0 = the running sum of one LogUp column that emits nothing, aux col 1
    = the Horner register."""
    b = dag.AirBuilder(1, aux_width=2, num_randomness=2, num_aux_values=1, num_public=4)
    acc, acc_next, beta = b.aux(1), b.aux(1, 1), b.randomness(1)
    b.assert_zero_ext(b.is_first_row() * acc)
    b.assert_zero_ext(b.is_transition() * (acc_next - acc * beta - b.main(0)))
    lk = dag.LogUp(b, MAX_MESSAGE_WIDTH, NUM_BUS_IDS, closing="sigma_last_row", num_logup_cols=1)
    with lk.column():
        pass
    assert lk.register(lambda bb: bb.randomness(1), lambda bb: bb.main(0)) == 1
    lookup = lk.finish("aux_register_spike")
    return _air(b, lookup, "aux_register_spike"), lookup


CHAIN_PERIOD = 4
CHAIN_KEEP = [1, 1, 1, 0]                                               # S restarts after every fourth row (S_KEEP of the multiplier)


def chain_air():
    """The multiplier's register shape on a small AIR: main = [x, y, m]; periodic [keep, sel]; one live LogUp column (multiplicity m on
    the message (x, y)); registers S' = keep S + (x + beta y) and id' = id + S (sel y) + (x y + beta^2), id reading S; plus a register
    with an extension-field keep, h' = beta h + x (the spike's)."""
    b = dag.AirBuilder(3, aux_width=5, num_randomness=2, num_aux_values=1, num_public=4, periodic=[CHAIN_KEEP, [0, 1, 0, 1]])
    x, y, beta = b.main(0), b.main(1), b.randomness(1)
    keep, sel = b.periodic_value(0), b.periodic_value(1)
    s, s_next, idr, id_next, h, h_next = b.aux(2), b.aux(2, 1), b.aux(3), b.aux(3, 1), b.aux(4), b.aux(4, 1)
    for r in (s, idr, h):
        b.assert_zero_ext(b.is_first_row() * r)
    b.assert_zero_ext(b.is_transition() * (s_next - s * keep - (x + beta * y)))
    b.assert_zero_ext(b.is_transition() * (id_next - idr - s * (sel * y) - (x * y + beta * beta)))
    b.assert_zero_ext(b.is_transition() * (h_next - h * beta - x))
    lk = dag.LogUp(b, MAX_MESSAGE_WIDTH, NUM_BUS_IDS, closing="sigma_last_row", num_logup_cols=2)
    msg = lambda ch: ch.encode(3, [(lk.b if ch is lk.ch_c else lk.lb).main(0), (lk.b if ch is lk.ch_c else lk.lb).main(1)])   # noqa: E731
    with lk.column():
        pass
    with lk.column() as col:
        with col.group() as g:
            with g.batch((lk.b.const(1), lk.lb.const(1))) as bt:
                bt.insert((lk.b.main(2), lk.lb.main(2)), msg)
    r_s = lk.register(lambda bb: bb.periodic_value(0), lambda bb: bb.main(0) + bb.randomness(1) * bb.main(1))
    r_id = lk.register(None, lambda bb: bb.main(0) * bb.main(1) + bb.randomness(1) * bb.randomness(1),
                       terms=[(r_s, lambda bb: bb.periodic_value(1) * bb.main(1))])
    r_h = lk.register(lambda bb: bb.randomness(1), lambda bb: bb.main(0))
    assert (r_s, r_id, r_h) == (2, 3, 4)
    lookup = lk.finish("register_chain")
    return _air(b, lookup, "register_chain"), lookup


def chain_trace(n, seed=3):
    rng = np.random.default_rng(seed)
    t = np.zeros((n, 3), dtype=np.uint64)
    t[:, 0] = rng.integers(0, P, n, dtype=np.uint64)
    t[:, 1] = rng.integers(0, 1 << 32, n, dtype=np.uint64)
    t[:, 2] = rng.integers(0, 3, n, dtype=np.uint64)
    return t


def check(pair, main, aux=None, rnd=RND):
    air, lookup = pair
    built, fin = ob.lookup_build_aux(lookup, main, rnd, None)
    return ob.check_constraints(air, main, built if aux is None else aux, [int(fin[0]), int(fin[1])], [1, 2, 3, 4], rnd, None)


def test_ext_register_verifies_and_stays_out_of_sigma():
    air, lookup = spike = spike_air()
    assert (lookup.num_cols, lookup.num_regs, lookup.num_aux_cols) == (1, 1, 2)
    rng = np.random.default_rng(0x5217e)
    main = rng.integers(0, 1 << 32, (16, 1), dtype=np.uint64)
    aux, sigma = ob.lookup_build_aux(lookup, main, RND, None)
    assert aux.shape == (16, 4) and (int(sigma[0]), int(sigma[1])) == (0, 0), "the register must not pollute sigma"
    assert not aux[:, 0:2].any()
    reg = (0, 0)
    for r in range(16):                                                 # reg[0] = 0, reg[r + 1] = reg[r] beta + x[r]
        assert (int(aux[r, 2]), int(aux[r, 3])) == reg, r
        reg = e_add(e_mul(reg, RND[1]), (int(main[r, 0]), 0))
    assert check(spike, main) == (0, None)
    bad = aux.copy()
    bad[5, 2] = (int(bad[5, 2]) + 1) % P
    assert check(spike, main, bad)[0] != 0
    folded = dag.AirBuilder(1, aux_width=2, num_randomness=2, num_aux_values=1, num_public=4)      # without the bound the register is folded in
    lk = dag.LogUp(folded, MAX_MESSAGE_WIDTH, NUM_BUS_IDS, closing="sigma_last_row")
    with lk.column():
        pass
    with pytest.raises(AssertionError):
        lk.finish()


def test_a_chain_of_registers_next_to_a_live_logup_column():
    air, lookup = chain = chain_air()
    assert (lookup.num_cols, lookup.num_regs) == (2, 3)
    n = 64
    main = chain_trace(n)
    aux, sigma = ob.lookup_build_aux(lookup, main, RND, None)
    assert aux.shape == (n, 10) and (int(sigma[0]), int(sigma[1])) != (0, 0)
    beta = RND[1]
    s = idr = h = (0, 0)
    for r in range(n):
        assert [int(v) for v in aux[r, 4:10]] == [*s, *idr, *h], r
        x, y = (int(main[r, 0]), 0), (int(main[r, 1]), 0)
        keep, sel = CHAIN_KEEP[r % 4], [0, 1, 0, 1][r % 4]
        s_next = e_add((s[0] * keep % P, s[1] * keep % P), e_add(x, e_mul(beta, y)))
        idr = e_add(e_add(idr, e_mul(s, (sel * y[0] % P, 0))), e_add(e_mul(x, y), e_mul(beta, beta)))
        h = e_add(e_mul(h, beta), x)
        s = s_next
    assert check(chain, main) == (0, None)
    for col in (4, 7, 9):
        bad = aux.copy()
        bad[17, col] = (int(bad[17, col]) + 1) % P
        assert check(chain, main, bad)[0] != 0, col
    # the LogUp columns are what they are without the registers: the same program minus its tail gives the same first four aux words
    plain = dag.LookupBuilder(3, num_cols=2, num_randomness=2, periodic=[CHAIN_KEEP, [0, 1, 0, 1]])
    w = [int(v) for v in lookup.blob]
    tail = 1 + 3 + (3 + 2) + 3                                          # count | (keep, build, 0) | (keep, build, 1, j, u) | (keep, build, 0)
    plain_blob = np.array(w[:-tail], dtype=np.uint64)

    class _L:
        blob, num_aux_cols = plain_blob, 2
    aux2, sigma2 = ob.lookup_build_aux(_L, main, RND, None)
    assert (aux2 == aux[:, 0:4]).all() and (sigma2 == sigma).all() and plain.num_cols == 2


@pytest.mark.parametrize("make, trace", [(spike_air, lambda: np.random.default_rng(1).integers(0, 1 << 32, (32, 1), dtype=np.uint64)),
                                         (chain_air, lambda: chain_trace(64, seed=9))], ids=["spike", "chain"])
def test_proofs_with_registers_verify_and_forged_registers_do_not(make, trace):
    air, lookup = make()
    main = trace()
    root = [5, 6, 7, 8]
    st = protocol.challenger_state((0, 0, 0, 0))
    proof = ob.prove([air], [main], root, FAST, init_state=st)
    pre = protocol.protocol_pre_observe(FAST, root)
    ok_o, msg = ob.verify([air], proof["log_heights"], root, proof, FAST)
    assert ok_o, msg
    ok_p, _ = pkg.verify([air], proof["log_heights"], root, FAST, st, pre, proof["fields"], proof["commitments"])
    assert ok_p

    def forged_build(m, rnd):
        aux, fin = ob.lookup_build_aux(lookup, m, rnd, None)
        aux[3, 2 * lookup.num_cols] = (int(aux[3, 2 * lookup.num_cols]) + 1) % P
        return aux, [int(fin[0]), int(fin[1])]
    b2 = make()[0]
    b2.build_aux = forged_build
    bad = ob.prove([b2], [main], root, FAST, init_state=st)
    assert not ob.verify([b2], bad["log_heights"], root, bad, FAST)[0]
    assert not pkg.verify([b2], bad["log_heights"], root, FAST, st, pre, bad["fields"], bad["commitments"])[0]


def test_malformed_register_tails_are_refused_by_the_oracle_and_by_the_library():
    """The register tail of the "MHLKP001" blob (include/midenhip.h): a register that reads itself, a register that does not exist, a cycle,
    a truncated tail and trailing words are format errors on both sides (the library's parser runs host-side in `mh_jit_precompile`)."""
    _, lookup = chain_air()
    w = [int(v) for v in lookup.blob]
    tail = 1 + 3 + (3 + 2) + 3                                          # count | (keep, build, 0) | (keep, build, 1, j, u) | (keep, build, 0)
    head, regs = w[:-tail], w[-tail:]
    assert regs[0] == 3 and regs[4 + 2] == 1 and regs[4 + 3] == 0, "register 1 (id) reads register 0 (S)"
    main = chain_trace(8)

    class _L:
        num_aux_cols = 5

    def both_refuse(words):
        _L.blob = np.array(words, dtype=np.uint64)
        with pytest.raises(RuntimeError):
            ob.lookup_build_aux(_L, main, RND, None)
        with pytest.raises(pkg.MidenHipError):
            pkg.jit_precompile(_L.blob)
    both_refuse(head + regs[:4 + 3] + [1] + regs[4 + 4:])               # id reads itself
    both_refuse(head + regs[:4 + 3] + [7] + regs[4 + 4:])               # ... or a register that does not exist
    cyc = list(regs)
    cyc[1:4] = [cyc[1], cyc[2], 1]                                      # S gains a term ...
    cyc[4:4] = [1, regs[4 + 4]]                                         # ... reading id, which reads S
    both_refuse(head + cyc)
    both_refuse(head + regs[:-1])                                       # truncated
    both_refuse(head + regs + [0])                                      # trailing words
    _L.blob = np.array(head + regs, dtype=np.uint64)
    aux, _ = ob.lookup_build_aux(_L, main, RND, None)
    assert aux.shape == (8, 10) and pkg.jit_precompile(_L.blob) >= 1
