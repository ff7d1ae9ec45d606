"""Drift guard for the Rust shim (bindings/rust/src/lib.rs), which cannot be compiled here (no rustc): its hand-written
`extern "C"` block is parsed and compared with the declarations of include/ldpc_hip.h -- every function it binds must
exist in the header with the same number of arguments and the same types (C type -> the Rust FFI type bindgen would
emit), and the same return type.  The crate-side items SURVEY.md 8(f1) asks of the shim are checked by name as well
(the traits of decoders/mod.rs:15-23 and noise/mod.rs:24-30 implemented, the accessors of
belief_propagation.rs:78-88, the generic decode of flip.rs:17-24, Display of flip.rs:50-54)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ldpc_hip.h")
SHIM = os.path.join(ROOT, "bindings", "rust", "src", "lib.rs")

SCALARS = {"size_t": "usize", "uint64_t": "u64", "uint32_t": "u32", "uint8_t": "u8", "double": "c_double", "int": "c_int",
           "char": "c_char", "void": "c_void"}


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def c_type_to_rust(ctype):
    """'const size_t *' -> '*const usize', 'ldpc_code **' -> '*mut *mut ldpc_code', 'uint64_t' -> 'u64'"""
    t = ctype.strip()
    stars = t.count("*")
    t = t.replace("*", " ").split()
    const = "const" in t
    base = [w for w in t if w != "const"]
    assert len(base) == 1, ctype
    base = SCALARS.get(base[0], base[0])
    if stars == 0:
        return base
    out = base
    for level in range(stars):
        # the innermost pointer carries the const of the pointee; outer levels are out-parameters (mut)
        out = ("*const " if const and level == 0 else "*mut ") + out
    return out


def header_functions():
    text = _strip_comments(open(HEADER).read())
    funcs = {}
    for m in re.finditer(r"\b((?:const\s+)?\w+\s*\**)\s*\b(ldpc_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        params = []
        if args.strip() != "void":
            for a in args.split(","):
                a = re.sub(r"(\w+)\s*\[[^\]]*\]", r"*\1", a.strip())      # array parameters decay to pointers
                mm = re.match(r"^(.*?)(\w+)\s*$", a.replace("*", " * "))
                ptype = mm.group(1)
                params.append(c_type_to_rust(ptype))
        rret = c_type_to_rust(ret)
        funcs[name] = ("()" if rret == "c_void" else rret, params)      # a C function returning void has no `->` in Rust
    return funcs


def shim_functions():
    text = _strip_comments(open(SHIM).read())
    block = re.search(r'extern\s+"C"\s*\{(.*?)\n\}', text, flags=re.S).group(1)
    funcs = {}
    for m in re.finditer(r"fn\s+(\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", block, flags=re.S):
        name, args, ret = m.group(1), m.group(2), (m.group(3) or "()").strip()
        params = []
        for a in [x.strip() for x in args.split(",") if x.strip()]:
            params.append(re.sub(r"\s+", " ", a.split(":", 1)[1].strip()))
        funcs[name] = (ret, params)
    return funcs


def test_c_type_mapping():
    assert c_type_to_rust("const size_t *") == "*const usize"
    assert c_type_to_rust("ldpc_code **") == "*mut *mut ldpc_code"
    assert c_type_to_rust("const ldpc_code *") == "*const ldpc_code"
    assert c_type_to_rust("uint64_t") == "u64"
    assert c_type_to_rust("const char *") == "*const c_char"


def test_every_bound_function_matches_the_header():
    hdr, shim = header_functions(), shim_functions()
    assert len(hdr) >= 70 and len(shim) >= 20, (len(hdr), len(shim))
    for name, (ret, params) in shim.items():
        assert name in hdr, "%s is bound by the shim but not declared in include/ldpc_hip.h" % name
        hret, hparams = hdr[name]
        assert ret == hret, (name, "return", ret, hret)
        assert len(params) == len(hparams), (name, "arity", len(params), len(hparams))
        for i, (a, b) in enumerate(zip(params, hparams)):
            assert a == b, (name, "argument %d" % i, a, b)


def test_the_shim_binds_the_whole_hot_path():
    shim = shim_functions()
    for name in ("ldpc_code_create", "ldpc_flip_create", "ldpc_flip_decode", "ldpc_flip_mc_run", "ldpc_bp_create", "ldpc_bp_correction_for",
                 "ldpc_bp_decode", "ldpc_bp_mc_run", "ldpc_css_create_with_logicals", "ldpc_css_bp_create", "ldpc_css_bp_mc_run",
                 "ldpc_bsc_sample_error_of_length", "ldpc_depolarizing_sample_error_of_length", "ldpc_erasure_sample_error_of_length",
                 "ldpc_code_has_codeword", "ldpc_last_error"):
        assert name in shim, name


def test_crate_side_api_surface():
    text = _strip_comments(open(SHIM).read())
    text = re.sub(r"\s+", " ", text)
    for needle in (
        "impl LinearDecoder for GpuBpDecoder",                                            # decoders/mod.rs:15-17
        "impl<'a> SyndromeDecoder<SparseBinSlice<'a>, SparseBinVec> for GpuBpDecoder",    # decoders/mod.rs:19-21
        "impl<'a> ClassicalSyndromeDecoder<'a> for GpuBpDecoder",                         # decoders/mod.rs:23
        "pub fn num_bits(&self) -> usize", "pub fn num_checks(&self) -> usize",           # belief_propagation.rs:78-84
        "pub fn has_zero_syndrome(&self, vector: SparseBinSlice) -> bool",                # belief_propagation.rs:86-88
        "pub struct GpuFlipDecoder<Code>", "Code: Borrow<LinearCode>",                    # flip.rs:7-9, 17-19
        "pub fn decode<T>(&self, message: &SparseBinVecBase<T>) -> SparseBinVec",         # flip.rs:21-24
        "impl<T> fmt::Display for GpuFlipDecoder<T>",                                     # flip.rs:50-54
        "impl NoiseModel for GpuBinarySymmetricChannel",                                  # noise/mod.rs:24-30
        "impl NoiseModel for GpuDepolarizingNoise", "impl NoiseModel for GpuErasureChannel",
        "fn sample_error_of_length<R: Rng>(&self, length: usize, rng: &mut R) -> Self::Error",
    ):
        assert needle in text, needle
