#!/usr/bin/env python3
"""Extract the reference's own known-answer vectors for the Poseidon2/Merkle hot path into
tests/golden/kat.json.  Run in the build container (needs /root/reference); the JSON is committed
because /root/reference does not exist on the GPU box.

Vectors:
  1. Poseidon2 permutation KAT .... crates/crypto/src/hash/algebraic_sponge/poseidon2/test.rs:7-39
  2. RELATION_DIGEST ............... air/src/config.rs:93-98  (= hash_elements([0] ++ ACE_ROOT))
  3. ACE_CIRCUIT_REGISTRY_ROOT ..... air/src/config.rs:103-108 (depth-3 merge tree over the leaves)
  4. ACE_CIRCUIT_REGISTRY_LEAVES ... air/src/config.rs:126-175 (leaves 6,7 = hash_elements([0xace,i]))
  5. EMPTY_SUBTREES ................ crates/crypto/src/merkle/empty_roots.rs:49-.. (chain of merge(x,x))
  6. Poseidon2 constants ........... .../poseidon2/constants.rs:18-211 (cross-check of p2_constants.inc)
  7. MASM recursive-verifier layout  crates/lib/core/asm/stark/constants.masm (quotient recomposition constants, FRI
     parameters, sizes of the OOD / aux-boundary / trace-row regions, per-AIR widths) and the production PCS parameters
     of air/src/config.rs:54-67: a second in-tree witness of how a Miden proof's streams are laid out.
  8. RPO hash_elements vectors ..... crates/crypto/src/hash/algebraic_sponge/rescue/rpo/tests.rs:241-267, 316-..
  9. AIR column layouts ............ air/src/constraints/snapshots/*_col_map_layout.snap (insta snapshots of the #[repr(C)] column
     structs: core, chiplets, hasher controller, bitwise, memory, ACE (+ read / eval overlays), kernel ROM) and the LogUp bus ids
     air/src/constraints/lookup/messages.rs:55-107 -- what the hand-ported AIRs' column tables are held to.
"""
import json, os, re
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

def ints(s):
    return [int(x, 16) if x.startswith("0x") else int(x) for x in re.findall(r"new_unchecked\((0x[0-9a-fA-F]+|\d+)\)", s)]

def const_block(src, name):
    m = re.search(r"const %s:[^=]*=\s*&?\[(.*?)\n\];" % name, src, re.S)
    assert m, name
    return ints(m.group(1))

def main():
    out = {}
    t = open(f"{REF}/crates/crypto/src/hash/algebraic_sponge/poseidon2/test.rs").read()
    body = t[t.index("fn permutation_test_vector"):t.index("fn test_poseidon2_permutation_basic")]
    exp = [int(x, 16) for x in re.findall(r"perm\[\d+\], Felt::new_unchecked\((0x[0-9a-f]+)\)", body)]
    assert len(exp) == 12
    out["permutation_kat"] = {"input": list(range(12)), "output": exp}
    cfg = open(f"{REF}/air/src/config.rs").read()
    out["relation_digest"] = const_block(cfg, "RELATION_DIGEST")
    out["ace_root"] = const_block(cfg, "ACE_CIRCUIT_REGISTRY_ROOT")
    leaves = const_block(cfg, "ACE_CIRCUIT_REGISTRY_LEAVES")
    assert len(leaves) == 32
    out["ace_leaves"] = [leaves[4*i:4*i+4] for i in range(8)]
    er = open(f"{REF}/crates/crypto/src/merkle/empty_roots.rs").read()
    es = const_block(er, "EMPTY_SUBTREES")
    assert len(es) == 1024
    out["empty_subtrees"] = [es[4*i:4*i+4] for i in range(256)]
    c = open(f"{REF}/crates/crypto/src/hash/algebraic_sponge/poseidon2/constants.rs").read()
    def hexblock(name):
        m = re.search(r"const %s:[^=]*=\s*\[(.*?)\];\n" % name, c, re.S)
        return [int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]+", m.group(1))]
    out["p2_constants"] = {k: hexblock(k) for k in ("MAT_DIAG", "ARK_EXT_INITIAL", "ARK_INT", "ARK_EXT_TERMINAL")}
    out["root_2_32"] = int(re.search(r"const ROOT_UNITY = (\d+)", open(f"{REF}/crates/lib/core/asm/stark/constants.masm").read()).group(1))
    masm = open(f"{REF}/crates/lib/core/asm/stark/constants.masm").read()
    def mconst(name):
        return int(re.search(r"^const %s = (\d+)" % name, masm, re.M).group(1))
    m = {k.lower(): mconst(k) for k in ("BLOWUP_FACTOR", "BLOWUP_FACTOR_LOG", "QUOTIENT_SHIFT_RATIO", "QUOTIENT_FIRST_SHIFT",
                                        "QUOTIENT_FIRST_WEIGHT", "LOG_FINAL_DEGREE", "FRI_FOLD_ARITY", "NUM_AUX_TRACE_COEFS")}
    m["ood_region_felts"] = mconst("AUX_BUS_BOUNDARY_PTR") - mconst("OOD_EVALUATIONS_PTR")
    m["aux_boundary_region_felts"] = mconst("AUXILIARY_ACE_INPUTS_PTR") - mconst("AUX_BUS_BOUNDARY_PTR")
    mm = re.search(r"main\s+= aligned\((\d+)\) \+ aligned\((\d+)\) \+ aligned\((\d+)\)", masm)
    m["main_widths"] = [int(x) for x in mm.groups()]
    mm = re.search(r"aux coords = aligned\((\d+) \* 2\) \+ aligned\((\d+) \* 2\) \+ aligned\((\d+) \* 2\)", masm)
    m["aux_widths_ef"] = [int(x) for x in mm.groups()]
    mm = re.search(r"quot\s+= (\d+) chunks \* 2 coordinates", masm)
    m["quotient_chunks"] = int(mm.group(1))
    mm = re.search(r"total per row = (\d+) EF slots = (\d+) felts", masm)
    m["ood_row_ef_slots"], m["ood_row_felts"] = int(mm.group(1)), int(mm.group(2))
    mm = re.search(r"The row therefore occupies (\d+) field\s*\n?##\s*elements", masm)
    m["trace_row_felts"] = int(mm.group(1))
    mm = re.search(r"maximal degree of the remainder polynomial\s*\n## that we allow is (\d+)", masm)
    m["max_remainder_degree"] = int(mm.group(1))
    out["masm_layout"] = m
    pcs = {}
    for k in ("LOG_BLOWUP", "LOG_FOLDING_ARITY", "LOG_FINAL_DEGREE", "FOLDING_POW_BITS", "DEEP_POW_BITS", "NUM_QUERIES", "QUERY_POW_BITS"):
        pcs[k.lower()] = int(re.search(r"const %s: \w+ = (\d+);" % k, cfg).group(1))
    out["pcs_params"] = pcs
    # 8. RPO: hash_elements([0, 1, .., i]) for i < 19 (rescue/rpo/tests.rs:241-267, EXPECTED :316-..): pins the Rescue Prime
    #    permutation (MDS, ARK1/ARK2, x^7 and x^(1/7)) that RPO and RPX share
    rt = open(f"{REF}/crates/crypto/src/hash/algebraic_sponge/rescue/rpo/tests.rs").read()
    ev = const_block(rt, "EXPECTED")
    assert len(ev) == 19 * 4
    out["rpo_hash_elements"] = [ev[4*i:4*i+4] for i in range(19)]
    # 9. column-layout snapshots and bus ids
    import glob
    layouts = {}
    for path in sorted(glob.glob(f"{REF}/air/src/constraints/snapshots/*_col_map_layout.snap")):
        name = re.search(r"tests__(\w+)_col_map_layout", path).group(1)
        body = open(path).read().split("---")[-1]
        flat = {}
        # `field: 3,`, `field: [3, 4,],`, `field: QuadFeltExpr(7, 8,),`, nested struct names are prefixes
        stack = []
        for line in body.splitlines():
            line = line.strip()
            mm = re.match(r"(\w+): (\w+) \{$", line) or re.match(r"^(\w+) \{$", line)
            if mm:
                stack.append(mm.group(1) if mm.lastindex == 2 else "")
                continue
            if line.startswith("}"):
                stack and stack.pop()
                continue
            mm = re.match(r"(\w+): (\d+),$", line)
            if mm:
                flat[".".join([x for x in stack if x] + [mm.group(1)])] = int(mm.group(2))
                continue
            mm = re.match(r"(\w+): (\[|QuadFeltExpr\()$", line)
            if mm:
                stack.append("@" + mm.group(1))
                flat[".".join([x for x in stack[:-1] if x] + [mm.group(1)])] = []
                continue
            mm = re.match(r"(\d+),$", line)
            if mm and stack and stack[-1].startswith("@"):
                flat[".".join([x for x in stack[:-1] if x] + [stack[-1][1:]])].append(int(mm.group(1)))
                continue
            if line in ("],", "),"):
                stack.pop()
        layouts[name] = flat
    out["col_maps"] = layouts
    msg = open(f"{REF}/air/src/constraints/lookup/messages.rs").read()
    enum = msg[msg.index("pub enum BusId"):msg.index("impl BusId")]
    out["bus_ids"] = {k: int(v) for k, v in re.findall(r"(\w+) = (\d+),", enum)}
    # 10. the VM's opcodes (core/src/operations/mod.rs:29-129)
    ops_src = open(f"{REF}/core/src/operations/mod.rs").read()
    out["opcodes"] = {k: int(v.replace("_", ""), 2) for k, v in re.findall(r"pub const (\w+): u8\s*= 0b([01_]+);", ops_src)}
    # 11. the recursive verifier's ACE circuit metadata (air/src/snapshots/miden_air__config__tests__relation_digest_matches_current_air.snap,
    #     test air/src/config.rs:383-454): num_inputs / num_eval_gates / stream_len of the six proof orders + the relation digest
    snap = open(f"{REF}/air/src/snapshots/miden_air__config__tests__relation_digest_matches_current_air.snap").read()
    orders = re.findall(r"(constraints_eval_\w+):\n  num_inputs: (\d+)\n  num_eval_gates: (\d+)\n  stream_len: (\d+)\n  commitment: \[([^\]]*)\]", snap)
    assert len(orders) == 6
    out["ace_circuit_snapshot"] = {"orders": {n: {"num_inputs": int(a), "num_eval_gates": int(b), "stream_len": int(c),
                                                   "commitment": [int(x) for x in d.split(",")]} for n, a, b, c, d in orders},
                                   "relation_digest": [int(x) for x in re.search(r"relation_digest: \[([^\]]*)\]", snap).group(1).split(",")]}
    # 12. operation batching: the nine `batch_ops_N` cases of core/src/mast/node/basic_block_node/tests.rs:12-270 with their insta
    #     snapshots (ops incl. padding NOOPs, indptr, padding flags, the 8 group slots, num_groups per batch)
    bt = open(f"{REF}/core/src/mast/node/basic_block_node/tests.rs").read()
    op_batches = []
    for k in range(1, 10):
        body = bt[bt.index(f"fn batch_ops_{k}()"):]
        body = body[:body.index("batch_and_hash_ops")]
        ops = []
        for name, arg, _num in re.findall(r"Operation::(\w+)(?:\((ONE|Felt::new_unchecked\((\d+)\))\))?", body):
            ops.append([name.upper(), 1 if arg == "ONE" else int(re.search(r"\d+", arg).group(0))] if arg else [name.upper()])
        snap = open(f"{REF}/core/src/mast/node/basic_block_node/snapshots/miden_core__mast__node__basic_block_node__tests__batch_ops_{k}.snap").read()
        batches = []
        for b in re.findall(r"OpBatch \{(.*?)num_groups: (\d+),", snap, re.S):
            text = b[0]
            bops = [[n.upper(), int(a)] if a else [n.upper()] for n, a in re.findall(r"^\s{12}(\w+)(?:\(\s*(\d+),\s*\))?,?$", text[text.index("ops: ["):text.index("indptr")], re.M)]
            nums = lambda key, nxt: [int(x) for x in re.findall(r"\d+", text[text.index(key):text.index(nxt)] if nxt else text[text.index(key):])]
            batches.append({"ops": bops, "indptr": nums("indptr: [", "padding"), "padding": re.findall(r"true|false", text[text.index("padding: ["):text.index("groups")]),
                            "groups": nums("groups: [", None), "num_groups": int(b[1])})
        op_batches.append({"ops": ops, "batches": batches})
    out["op_batches"] = op_batches
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote kat.json:", {k: (len(v) if hasattr(v, '__len__') else v) for k, v in out.items()})

if __name__ == "__main__":
    main()
