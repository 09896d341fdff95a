"""include/colorid_hip.rs — the Rust side of the boundary, generated from the header by tools/gen_rust_bindings.py — names
every entry point the library exports, with the arity the header and the ctypes table give it, and is not stale."""
import importlib.util
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gen():
    spec = importlib.util.spec_from_file_location("gen_rust_bindings", os.path.join(ROOT, "tools", "gen_rust_bindings.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _rust_fns():
    src = open(os.path.join(ROOT, "include", "colorid_hip.rs")).read()
    fns = {}
    for name, args, ret in re.findall(r"pub fn (cid_\w+)\(([^)]*)\)( -> [^;]+)?;", src):
        fns[name] = ([a.strip() for a in args.split(",") if a.strip()], ret.strip())
    return fns, src


def test_the_committed_file_is_what_the_generator_writes():
    text, funcs = _gen().generate()
    assert open(os.path.join(ROOT, "include", "colorid_hip.rs")).read() == text
    assert len(funcs) >= 100


def test_every_exported_symbol_is_declared_with_its_arity():
    from colorid_amd import _lib
    fns, src = _rust_fns()
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "colorid_amd", "libcolorid_hip.so")], stdout=subprocess.PIPE, text=True, check=True).stdout
    exported = sorted(ln.split()[-1] for ln in out.splitlines() if ln.strip())
    assert sorted(fns) == exported
    assert sorted(fns) == sorted(_lib.SIGNATURES)
    for name, (restype, argtypes) in _lib.SIGNATURES.items():
        args, ret = fns[name]
        assert len(args) == len(argtypes), name
        assert (ret == "") == (restype is None), name
    for t in ("cid_ctx", "cid_index", "cid_kmerset", "cid_group", "cid_group_kmerset", "cid_fastq"):
        assert f"pub struct {t} {{ _private: [u8; 0] }}" in src


def test_pointer_spellings():
    fns, _ = _rust_fns()
    assert fns["cid_last_error"] == ([], "-> *const c_char")
    assert fns["cid_ctx_create"][0][-1].endswith(": *mut *mut cid_ctx")
    a = dict(x.split(": ") for x in fns["cid_group_search_count"][0])
    assert a["replicas"] == "*const *mut cid_index" and a["kmers"] == "*const u8" and a["hits"] == "*mut u64" and a["n_kmers"] == "usize"
    a = dict(x.split(": ") for x in fns["cid_readid_count_resident"][0])
    assert a["d_bases"] == "*const u8" and a["seq_off"] == "*const u64" and a["d_status"] == "*mut u8" and a["stride_d"] == "u32"
    assert fns["cid_kmerset_destroy"][1] == ""


def test_core_and_extended_modules_follow_the_headers_marks():
    """include/colorid_hip.h marks the stable core with CID_CORE (SURVEY.md §8b's calls + what the one-GPU command line runs on); the
    generated Rust file carries the same split as `mod core` / `mod extended`, re-exported flat"""
    hdr = open(os.path.join(ROOT, "include", "colorid_hip.h")).read()
    core_h = set(re.findall(r"^CID_CORE [\w \*]*?(cid_\w+)\(", hdr, re.M))
    src = open(os.path.join(ROOT, "include", "colorid_hip.rs")).read()
    core_rs = set(re.findall(r"pub fn (cid_\w+)\(", src[src.index("pub mod core {"):src.index("pub mod extended {")]))
    ext_rs = set(re.findall(r"pub fn (cid_\w+)\(", src[src.index("pub mod extended {"):]))
    assert core_h == core_rs and not (core_rs & ext_rs)
    for name in ("cid_ctx_create", "cid_index_create", "cid_index_put_rows", "cid_index_finalize", "cid_search_count", "cid_search_perfect", "cid_readid_count",
                 "cid_index_destroy", "cid_ctx_destroy", "cid_last_error"):            # SURVEY.md §8b's ten
        assert name in core_rs
    assert 40 <= len(core_rs) <= 50 and len(core_rs) + len(ext_rs) == len(_rust_fns()[0])
    assert "pub use self::core::*;" in src and "pub use self::extended::*;" in src
