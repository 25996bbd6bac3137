#!/usr/bin/env python3
"""Extract the reference processor's per-operation snapshots into tests/golden/all_ops_snapshots.json.

`test_basic_block` (processor/src/fast/tests/all_ops.rs:12-134) runs the cross product of 17 stack-input vectors ([], [1], [1, 2], ..,
[1..16], top first) and 69 operation sequences through the reference processor and snapshots the resulting `StackOutputs` -- or the
error -- in processor/src/fast/tests/snapshots/*all_ops*test_basic_block*.snap (1173 files).  They are outputs of the REFERENCE for
(inputs, operations) pairs: what pins the operation semantics of the test VM (miden-vm_amd/testing/core_trace.py), which every AIR
test over executed programs leans on.  Run in the build container (needs /root/reference)."""
import glob, json, os, re
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    src = open(f"{REF}/processor/src/fast/tests/all_ops.rs").read()
    ops_block = src[src.index("#[values(", src.index("stack_inputs: Vec<Felt>")):src.index("operations: Vec<Operation>")]
    sequences = []
    for line in ops_block.splitlines():
        line = line.strip()
        if not line.startswith("vec!["):
            continue
        seq = []
        for name, arg in re.findall(r"Operation::(\w+)(?:\(Felt::from_u32\((\d+)\)\))?", line):
            seq.append([name.upper(), int(arg)] if arg else [name.upper()])
        sequences.append(seq)
    assert len(sequences) == 69, len(sequences)
    cases = []
    for path in sorted(glob.glob(f"{REF}/processor/src/fast/tests/snapshots/*all_ops*test_basic_block*.snap")):
        m = re.search(r"stack_inputs_(\d+)_.*operations_(\d+)_", os.path.basename(path))
        si, op = int(m.group(1)), int(m.group(2))
        body = open(path).read().split("---")[2]
        if body.lstrip().startswith("Ok("):
            out = [int(x) for x in re.findall(r"^\s+(\d+),$", body, re.M)]
            assert len(out) == 16, path
            cases.append({"inputs": si - 1, "ops": op - 1, "ok": out})
        else:
            err = re.search(r"err: (\w+)", body)
            cases.append({"inputs": si - 1, "ops": op - 1, "err": err.group(1) if err else re.search(r"Err\(\s*(\w+)", body).group(1)})
    assert len(cases) == 17 * 69, len(cases)
    out = {"source": "processor/src/fast/tests/snapshots/*all_ops*test_basic_block* (test: processor/src/fast/tests/all_ops.rs:12-134)",
           "stack_inputs": "case['inputs'] = n means [1, 2, .., n], top of the stack first", "sequences": sequences, "cases": cases}
    with open(os.path.join(HERE, "all_ops_snapshots.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    from collections import Counter
    print(len(cases), "cases;", Counter("ok" if "ok" in c else c["err"] for c in cases))


if __name__ == "__main__":
    main()
