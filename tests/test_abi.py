"""CPU-only: the C-ABI library loads and exports every symbol include/midenhip.h declares; the
product path fails loudly without a GPU (no CPU fallback)."""
import os, re
import pytest
from __graft_entry__ import load_package, ROOT


def header_symbols():
    h = open(os.path.join(ROOT, "include", "midenhip.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(mh_[a-z0-9_]+)\s*\(", h)))


def test_library_exports_every_declared_symbol():
    pkg = load_package()
    if not os.path.exists(pkg.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = pkg.load_library()
    syms = header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in midenhip.h but not exported"
    assert sorted(pkg.EXPORTS) == syms


def test_rust_sys_declarations_cover_the_header():
    """bindings/rust/midenhip_sys.rs (the -sys layer of INTEGRATION.md's shim; uncompiled here, no Rust toolchain) must
    declare exactly the header's symbols."""
    src = open(os.path.join(ROOT, "bindings", "rust", "midenhip_sys.rs")).read()
    rs = sorted(set(re.findall(r"pub fn (mh_[a-z0-9_]+)\s*\(", src)))
    assert rs == header_symbols()


def test_no_cpu_fallback():
    pkg = load_package()
    lib = pkg.load_library()
    if lib.mh_device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(pkg.MidenHipError):
        pkg.Ctx(0)


def test_product_never_touches_oracle():
    # the oracle is test infrastructure: nothing under miden-vm_amd/ may reference it
    for dp, _, fs in os.walk(os.path.join(ROOT, "miden-vm_amd")):
        if "build" in dp:
            continue
        for f in fs:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".cuh", ".h", "Makefile")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle" not in src.lower(), (dp, f)


def test_configuration_ids_agree_everywhere():
    """MH_LMCS_* in the header, the Rust -sys constants, the ctypes layer and the reference's HashFunction mirror name the same five
    configurations with the same numbers."""
    h = open(os.path.join(ROOT, "include", "midenhip.h")).read()
    ids = {k: int(v) for k, v in re.findall(r"#define MH_LMCS_([A-Z0-9]+) (\d+)", h)}
    assert ids == {"POSEIDON2": 0, "BLAKE3": 1, "KECCAK": 2, "RPO": 3, "RPX": 4}
    rs = open(os.path.join(ROOT, "bindings", "rust", "midenhip_sys.rs")).read()
    assert {k: int(v) for k, v in re.findall(r"pub const MH_LMCS_([A-Z0-9]+): c_int = (\d+);", rs)} == ids
    pkg = load_package()
    assert {k.upper(): v for k, v in pkg.Ctx.LMCS.items()} == ids
    from miden_vm_amd import protocol
    assert set(protocol.HashFunction.LMCS.values()) == set(pkg.Ctx.LMCS)
    assert protocol.HashFunction.LMCS[protocol.ProvingOptions().hash_fn()] == "blake3"


def test_example_trace_generator_exports_its_entry_points():
    """examples/libkeccak_trace_device.so (built by __graft_entry__.build(): a client-side GPU trace generator over the public C ABI)
    loads without a GPU and exports what tests/test_gpu_precompile.py and bench.py call."""
    import ctypes
    so = os.path.join(ROOT, "examples", "libkeccak_trace_device.so")
    if not os.path.exists(so):
        pytest.skip("examples/ not built (run __graft_entry__.build())")
    lib = ctypes.CDLL(so)
    for sym in ("kt_keccak_round_trace", "kt_download", "kt_free"):
        assert hasattr(lib, sym), sym


def test_product_never_imports_the_test_generators():
    """miden-vm_amd/testing/ (the small VM and the chiplet trace builder: witness generators for tests and bench workloads) is not
    part of the proving backend: no product module may import it."""
    pkg_dir = os.path.join(ROOT, "miden-vm_amd")
    for f in os.listdir(pkg_dir):
        if f.endswith(".py"):
            src = open(os.path.join(pkg_dir, f)).read()
            assert not re.search(r"^\s*(from|import)\s+[\w.]*\b(testing|core_trace|chiplets_trace|precompile_trace)\b", src, re.M), f
            # ... and the witness side of the second client stays out of the AIR module it was split from (round 6)
            if f == "precompile_airs.py":
                assert not re.search(r"^(def \w+_traces?\b|def \w+_session\b|class \w*Requires\b|class (Session|SessionTraces|UintStore|EcStore)\b)", src, re.M), f
