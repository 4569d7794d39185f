"""CPU: the C-ABI library builds in-tree, loads, and exports every symbol include/dbhip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "dbhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dbhip_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from databend_amd import _lib
    assert os.path.exists(_lib.library_path()), "libdbhip.so not built (run __graft_entry__.build())"
    L = ctypes.CDLL(_lib.library_path())
    syms = header_symbols()
    assert len(syms) >= 45
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(_lib.SYMBOLS) == syms, set(_lib.SYMBOLS) ^ set(syms)
    assert L.dbhip_abi_version() == 6


def test_no_cpu_fallback_without_device():
    """Without a GPU dbhip_init must fail with DBHIP_ERR_NO_DEVICE; it must never 'work' on CPU."""
    from databend_amd import _lib
    L = _lib.load_library()
    n = ctypes.c_int32(-1)
    L.dbhip_device_count(ctypes.byref(n))
    if n.value == 0:
        assert L.dbhip_init(0) == _lib.ERR_NO_DEVICE
        assert b"no CPU fallback" in L.dbhip_last_error()


def test_product_does_not_reference_oracle():
    """The oracle is test infrastructure: nothing in the package may import/link it."""
    pkg = os.path.join(ROOT, "databend_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "liboracle" not in txt and "oracle_lib" not in txt and "orc_" not in txt, os.path.join(dp, f)


def test_run_time_specialisation_compiles_without_a_gpu():
    """dbhip_groupby_add_block_program's hiprtc path: the embedded device headers + a generated constexpr program must
    compile for gfx950 on the CPU box (the interpreter is the fallback, so a broken header would otherwise go unnoticed)."""
    import ctypes as C
    from databend_amd import _lib
    L = _lib.load_library()
    L.dbhip_jit_compile_check.restype = C.c_int64
    buf = C.create_string_buffer(1 << 16)
    size = L.dbhip_jit_compile_check(buf, C.c_int64(1 << 16))
    assert size > 4096, buf.value.decode(errors="replace")[:4000]


def test_specialised_kernels_are_cached_on_disk(tmp_path, monkeypatch):
    """a code object compiled once is found again by key (generated source + embedded headers + flags +
    target) — by a later table, a later process — instead of paying hiprtc again; DBHIP_JIT_CACHE_DIR=off disables."""
    import ctypes as C
    import time
    from databend_amd import _lib
    L = _lib.load_library()
    L.dbhip_jit_compile_check.restype = C.c_int64
    buf = C.create_string_buffer(1 << 16)
    monkeypatch.setenv("DBHIP_JIT_CACHE_DIR", str(tmp_path / "cache"))
    t0 = time.perf_counter()
    size = L.dbhip_jit_compile_check(buf, C.c_int64(1 << 16))
    t1 = time.perf_counter()
    assert size > 4096
    files = sorted(os.listdir(tmp_path / "cache"))
    assert len(files) == 1 and files[0].startswith("fagg_") and files[0].endswith(".co")
    assert os.path.getsize(tmp_path / "cache" / files[0]) == size
    again = L.dbhip_jit_compile_check(buf, C.c_int64(1 << 16))
    t2 = time.perf_counter()
    assert again == size and (t2 - t1) < (t1 - t0) / 5, (t1 - t0, t2 - t1)
    # a corrupted / foreign cache directory never breaks a compile: disabled cache still compiles
    monkeypatch.setenv("DBHIP_JIT_CACHE_DIR", "off")
    assert L.dbhip_jit_compile_check(buf, C.c_int64(1 << 16)) == size
    assert sorted(os.listdir(tmp_path / "cache")) == files


def test_rust_binding_file_matches_the_header():
    """bindings/dbhip_sys.rs (the `extern "C"` block + #[repr(C)] structs a Rust host links against) is generated from
    include/dbhip.h by tools/gen_rust_bindings.py: the committed file must be what the header produces today, bind every exported
    function exactly once, and carry the struct layouts of the ctypes mirror."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_rust_bindings as G
    text, funcs = G.render(open(os.path.join(ROOT, "include", "dbhip.h")).read())
    assert open(os.path.join(ROOT, "bindings", "dbhip_sys.rs")).read() == text, "stale bindings: run python tools/gen_rust_bindings.py"
    assert sorted(funcs) == header_symbols() and len(set(funcs)) == len(funcs)
    # struct fields in declaration order == the ctypes mirror's (_lib.py), so both bindings agree on the layout
    from databend_amd import _lib
    for rust_name, ct in (("dbhip_col", _lib.Col), ("dbhip_agg_desc", _lib.AggDesc), ("dbhip_expr_ins", _lib.ExprIns),
                          ("dbhip_agg_program", _lib.AggProgram), ("dbhip_pq_info", _lib.PqInfo)):
        body = re.search(r"pub struct %s \{(.*?)\}" % rust_name, text, flags=re.S).group(1)
        fields = [f.replace("r#", "") for f in re.findall(r"pub (\S+):", body)]
        assert fields == [f[0] for f in ct._fields_], (rust_name, fields)
    for const, val in (("DBHIP_T_DEC256", 17), ("DBHIP_ERR_UNSUPPORTED", 7), ("DBHIP_AGG_MAX", 3), ("DBHIP_ABI_VERSION", 6)):
        assert re.search(r"pub const %s: i32 = %d;" % (const, val), text), const


def test_the_shipped_library_reads_no_experiment_knob():
    """the development sweeps' knobs (grid sizes, kernel variants, thresholds — and the work-skipping
    DBHIP_FAGG_DEBUG of rounds 2-5, deleted) are compiled in only with -DDBHIP_EXPERIMENTS (csrc/runtime.h exp_env). The shipped binary
    names exactly the documented configuration variables, none of which skips work."""
    import subprocess
    from databend_amd import _lib
    out = subprocess.run(["strings", "-n", "6", _lib.library_path()], capture_output=True, text=True, check=True).stdout
    names = set(re.findall(r"\bDBHIP_[A-Z0-9_]{3,}\b", out))
    env_like = {n for n in names if not n.startswith(("DBHIP_T_", "DBHIP_ERR_", "DBHIP_EX_", "DBHIP_AGG_", "DBHIP_OP_", "DBHIP_CMP_", "DBHIP_VEC_", "DBHIP_ARG_",
                                                      "DBHIP_JOIN_", "DBHIP_OK", "DBHIP_ABI", "DBHIP_NULL", "DBHIP_H", "DBHIP_JIT\b"))}
    env_like -= {"DBHIP_JIT", "DBHIP_WAVE", "DBHIP_EXPERIMENTS", "DBHIP_REQUIRE", "DBHIP_CHECK", "DBHIP_TRY", "DBHIP_LAUNCH_CHECK", "DBHIP_POLL_CANCEL"}   # macro names inside the embedded JIT headers
    allowed = {"DBHIP_TRACE", "DBHIP_JIT_CACHE_DIR", "DBHIP_JIT_ARCH", "DBHIP_FAGG_JIT", "DBHIP_COMM_TIMEOUT_S", "DBHIP_CACHE_BYTES"}
    assert env_like <= allowed, sorted(env_like - allowed)
    assert "FAGG_DEBUG" not in out and "FA_X_SKIP" not in out


def test_the_rust_shim_source_calls_only_declared_functions():
    """bindings/shim (the Rust side as source; not compilable here) may only call what include/dbhip.h declares and
    what the generated FFI (bindings/dbhip_sys.rs) binds, with the same struct fields."""
    shim = os.path.join(ROOT, "bindings", "shim", "src")
    declared = set(header_symbols())
    sys_rs = open(os.path.join(ROOT, "bindings", "dbhip_sys.rs")).read()
    bound = set(re.findall(r"pub fn (dbhip_\w+)\(", sys_rs))
    types = set(re.findall(r"pub struct (dbhip_\w+)", sys_rs))
    files = [f for f in sorted(os.listdir(shim)) if f.endswith(".rs")]
    assert {"lib.rs", "device.rs", "scalar.rs", "aggregate.rs", "join.rs"} <= set(files)
    for f in files:
        src = re.sub(r"//[^\n]*", "", open(os.path.join(shim, f)).read())
        for name in set(re.findall(r"\b(dbhip_\w+)\s*\(", src)):
            assert name in declared and name in bound, (f, name)
        for name in set(re.findall(r"\b(dbhip_\w+)\s*\{", src)):   # struct literals
            assert name in types, (f, name)
    # the three trait impls the hot path dispatches through
    allsrc = "".join(open(os.path.join(shim, f)).read() for f in files)
    for needle in ("impl ScalarFunction for HipArith", "impl AccumulatingTransform for HipTransformPartialAggregate", "impl Join for HipInnerHashJoin",
                   "impl JoinStream for HipJoinStream"):
        assert needle in allsrc, needle
