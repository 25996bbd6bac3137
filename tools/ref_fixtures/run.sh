#!/bin/bash
# One command for a maintainer with a Rust toolchain:  tools/ref_fixtures/run.sh /path/to/miden-vm
# Installs the fixture crate into the reference workspace, proves the cases with the REFERENCE prover and copies
# the resulting ref_*.json into tests/golden/, where tests/test_ref_fixtures.py picks them up.
set -euo pipefail
REF=${1:?usage: run.sh /path/to/miden-vm-checkout}
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
mkdir -p "$REF/benches/midenhip-fixtures/src/bin"
cp "$HERE/Cargo.toml" "$REF/benches/midenhip-fixtures/Cargo.toml"
cp "$HERE/src/main.rs" "$REF/benches/midenhip-fixtures/src/main.rs"
cp "$HERE/src/bin/export_dag.rs" "$REF/benches/midenhip-fixtures/src/bin/export_dag.rs"
grep -q 'benches/midenhip-fixtures' "$REF/Cargo.toml" || sed -i 's|"benches/miden-bench",|"benches/miden-bench",\n    "benches/midenhip-fixtures",|' "$REF/Cargo.toml"
python3 "$HERE/make_inputs.py" "$HERE/inputs" > "$HERE/inputs/commands.sh"
(cd "$REF" && bash "$HERE/inputs/commands.sh")
cp "$HERE"/inputs/ref_*.json "$ROOT/tests/golden/"
# the real Miden AIRs as constraint-DAG blobs: tests/test_miden_air_blobs.py picks them up
(cd "$REF" && cargo run --release -p midenhip-fixtures --bin export_dag -- "$HERE/inputs") && cp "$HERE"/inputs/miden_air_*.dag "$ROOT/tests/golden/"
echo "fixtures written to tests/golden/: now run  python -m pytest tests/test_ref_fixtures.py -q  (and -m gpu on an MI355X)"
