"""TEST-ONLY witness generators -- not part of the proving backend.

The reference builds execution traces in `processor/` (OUT OF SCOPE for this library, whose path starts at
`prove_stark`'s three matrices).  The modules here exist so that the hand-ported AIRs (core_air.py, chiplets_air.py,
miden_air.py) can be exercised on traces of executed programs without a Rust toolchain:

* `core_trace`     -- a small VM (SPAN / JOIN / SPLIT / LOOP, ~60 operations) emitting the 51-column core trace;
* `chiplets_trace` -- hasher controller, bitwise, memory, ACE, kernel-ROM segments and the Poseidon2 permutation requests;
* `precompile_trace` (round 6: moved out of `precompile_airs.py`) -- the witness side of the SECOND client: the trace generators,
  `*Requires` ledgers, `UintStore` / `EcStore` and the `Session` / `SessionTraces` front end of the precompile prover's twelve chiplets
  (`precompiles-prover/src/**/trace.rs`, `session/mod.rs`).

They are pinned to the reference processor CELL FOR CELL on the 17 programs of its own snapshot test that they can execute
(tests/test_ref_traces.py, processor/src/trace/parallel/tests.rs:320-450).  Their feature set is FROZEN: no CALL / SYSCALL / DYN /
DYNCALL / EXTERNAL, no advice-driven operations -- statements with those node types are tested from the reference's own snapshot
traces (tests/golden/ref_traces.json.gz).  Used by tests/, bench.py's `miden_real*` workloads and tools/ only; nothing in the
product path (csrc/, __init__.py, dag.py, protocol.py, sharding.py, the AIR modules, miden_statement.py) imports them
(tests/test_abi.py enforces it).  The place mirrors the reference's own `crates/lifted-stark/src/testing/`."""
