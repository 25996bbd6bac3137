#!/usr/bin/env python3
"""Write the trace files of the reference-fixture cases (tools/ref_fixtures/README.md) and print the command lines.

A case is a list of (log_height, width, aux_cols) DummyMidenAir instances in INSTANCE order; its traces come from
tests/airs.py `dummy_trace(log_height, width, seed)` (numpy PCG64, column 0 all zero), so tests/test_ref_fixtures.py can
rebuild the same matrices without the files."""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

CASES = {
    # name: [(log_height, width, aux_cols, seed), ...]   -- production parameters (config::pcs_params()) for all
    "miden_6_11_2": [(6, 11, 2, 1)],
    "miden_10_51_8": [(10, 51, 8, 1)],
    "miden_shape_10_9_8": [(10, 51, 4, 3), (9, 22, 3, 4), (8, 16, 1, 5)],
    "miden_mixed_order": [(9, 16, 1, 5), (7, 51, 4, 3), (9, 22, 3, 4)],
}
OTHER_HASHERS_FOR = ("miden_6_11_2", "miden_shape_10_9_8")


def main():
    import airs as A
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "inputs")
    os.makedirs(out, exist_ok=True)
    lines = []
    for name, insts in CASES.items():
        specs = []
        for k, (lh, w, aux, seed) in enumerate(insts):
            path = os.path.join(out, f"{name}_{k}.bin")
            A.dummy_trace(lh, w, seed=seed).astype("<u8").tofile(path)
            specs.append(f"{lh}:{w}:{aux}:{path}")
        lines.append(f"cargo run --release -p midenhip-fixtures -- {os.path.join(out, 'ref_' + name + '.json')} " + " ".join(specs))
        if name in OTHER_HASHERS_FOR:  # the other four StarkConfigs of prove_stark on a couple of cases: ref_<case>@<hasher>.json
            for h in ("blake3", "keccak", "rpo", "rpx"):
                lines.append(f"cargo run --release -p midenhip-fixtures -- --hasher {h} "
                             f"{os.path.join(out, 'ref_' + name + '@' + h + '.json')} " + " ".join(specs))
    json.dump(CASES, open(os.path.join(out, "cases.json"), "w"))
    print("\n".join(lines))


if __name__ == "__main__":
    main()
