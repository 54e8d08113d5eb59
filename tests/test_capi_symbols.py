"""CPU-only: libmarlin_hip.so loads and exports every symbol include/marlin_hip.h declares
(no compute calls without a GPU), and the product fails loudly without a device."""
import os
import re
import pytest
import marlin_amd
from marlin_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "marlin_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(mh_[a-z0-9_]+)\s*\(", hdr)))


def test_header_symbols_exported_and_bound():
    lib = marlin_amd.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libmarlin_hip.so lacks %s" % n
        assert n in _lib.SYMBOLS, "ctypes binding lacks %s" % n
    assert sorted(_lib.SYMBOLS) == names


def _exported(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return sorted(l.split()[-1] for l in out.splitlines() if l.split()[-1].startswith("mh_"))


def test_the_product_library_exports_the_header_and_nothing_else():
    """the libraries are built with -fvisibility=hidden; `nm -D libmarlin_hip.so` lists exactly the entry
    points include/marlin_hip.h declares -- no debug hook, no self-test, no internal helper -- on both curves, and the hooks
    libraries (the same objects + testhooks.hip) add exactly what include/marlin_hip_testhooks.h declares."""
    names = _declared()
    hooks_hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "marlin_hip_testhooks.h")).read(), flags=re.S)
    hooks = sorted(set(re.findall(r"\b(mh_[a-z0-9_]+)\s*\(", hooks_hdr)))
    assert hooks == sorted(_lib.HOOK_SYMBOLS) and len(hooks) >= 4
    for lib, hk in (("libmarlin_hip.so", "libmarlin_hip_testhooks.so"), ("libmarlin_hip_bn254.so", "libmarlin_hip_bn254_testhooks.so")):
        got = _exported(os.path.join(ROOT, "marlin_amd", lib))
        assert got == names, (lib, set(got) ^ set(names))
        assert not [n for n in got if "debug" in n or "selftest" in n or "_test_" in n], lib
        assert _exported(os.path.join(ROOT, "marlin_amd", hk)) == sorted(names + hooks), hk


def test_no_device_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(marlin_amd.MarlinHipError):
        marlin_amd.init(0)


def test_product_does_not_use_oracle():
    pkg = os.path.join(ROOT, "marlin_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".cpp", ".inc")) or f == "Makefile":
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_rust_shim_binds_the_whole_header():
    """shim/marlin-hip-sys/src/ffi.rs (uncompiled Rust side of the boundary) declares exactly the header's entry points, and
    its tests/ffi_symbols.rs takes the address of each of them."""
    names = _declared()
    ffi = open(os.path.join(ROOT, "shim", "marlin-hip-sys", "src", "ffi.rs")).read()
    bound = sorted(set(re.findall(r"pub fn (mh_[a-z0-9_]+)\s*\(", ffi)))
    assert bound == names, (set(names) - set(bound), set(bound) - set(names))
    link = open(os.path.join(ROOT, "shim", "marlin-hip-sys", "tests", "ffi_symbols.rs")).read()
    for n in names:
        assert "%s as usize" % n in link, n


def _c_prototypes():
    """{name: (return type, [parameter types])} of include/marlin_hip.h, C spelling normalised ("const T *" forms)."""
    hdr = open(os.path.join(ROOT, "include", "marlin_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    hdr = re.sub(r"typedef\s+struct\s*\{.*?\}\s*\w+\s*;", "", hdr, flags=re.S)
    hdr = re.sub(r"typedef\s+int\s*\(\*\w+\)\s*\(.*?\)\s*;", "", hdr, flags=re.S)
    hdr = re.sub(r"typedef\s+struct\s+\w+\s*\*\s*\w+\s*;", "", hdr)
    out = {}
    for ret, name, args in re.findall(r"\b(int|const char\s*\*|mh_ctx_t)\s+(mh_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", hdr):
        params = []
        if args.strip() not in ("", "void"):
            for a in args.split(","):
                a = " ".join(a.replace("*", " * ").split())
                toks = a.split(" ")
                if toks[-1] != "*" and len(toks) > 1 and re.match(r"^[A-Za-z_]\w*$", toks[-1]) and toks[-1] not in ("int", "size_t", "char", "void", "double"):
                    toks = toks[:-1]                                  # the parameter's name
                params.append(" ".join(toks))
        out[name] = (" ".join(ret.replace("*", " * ").split()), params)
    return out


_C2RUST_BASE = {"int": "c_int", "size_t": "usize", "uint64_t": "u64", "uint32_t": "u32", "uint8_t": "u8", "char": "c_char", "void": "c_void",
                "double": "f64", "mh_r1cs_matrices": "mh_r1cs_matrices", "mh_verifier_key": "mh_verifier_key",
                "mh_allgather_fn": "mh_allgather_fn", "mh_alltoall_fn": "mh_alltoall_fn", "mh_allgather_dev_fn": "mh_allgather_dev_fn",
                "mh_fiat_shamir": "mh_fiat_shamir", "mh_ctx_t": "mh_ctx_t", "mh_group_t": "mh_group_t", "mh_group_fn": "mh_group_fn"}


def _c_to_rust(ctype):
    """`const uint64_t * const *` -> `*const *const u64`: pointers are read right to left, each with the constness of what
    it points to."""
    toks = ctype.split(" ")
    const_base = False
    if toks[0] == "const":
        const_base, toks = True, toks[1:]
    base, rest = _C2RUST_BASE[toks[0]], toks[1:]
    levels, pointee_const = [], const_base              # constness of the thing the next '*' points to
    i = 0
    while i < len(rest):
        assert rest[i] == "*", ctype
        levels.append("*const" if pointee_const else "*mut")
        pointee_const = i + 1 < len(rest) and rest[i + 1] == "const"
        i += 2 if pointee_const else 1
    return " ".join(levels[::-1] + [base]) if levels else base


_RUST2CTYPES = {"c_int": "c_int", "usize": "c_size_t", "u64": "c_uint64", "u32": "c_uint32", "u8": "c_ubyte", "f64": "c_double"}


def test_rust_and_ctypes_signatures_match_the_header():
    """not only the NAMES of the header's entry points but every ARGUMENT LIST -- count, order and the
    C <-> Rust type of each parameter, and the return type -- is compared between include/marlin_hip.h and the extern block of
    shim/marlin-hip-sys/src/ffi.rs (which no rustc has ever seen), and the parameter COUNT and every by-value scalar type with
    the ctypes table of marlin_amd/_lib.py (which every test calls through)."""
    import ctypes as C
    protos = _c_prototypes()
    assert sorted(protos) == _declared()
    ffi = open(os.path.join(ROOT, "shim", "marlin-hip-sys", "src", "ffi.rs")).read()
    ffi = re.sub(r"//[^\n]*", "", ffi)
    rust = {}
    for name, args, ret in re.findall(r"pub fn (mh_[a-z0-9_]+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+))?;", ffi):
        params = [" ".join(a.split(":", 1)[1].split()) for a in args.split(",") if a.strip()]
        rust[name] = (" ".join((ret or "()").split()), params)
    for name, (cret, cparams) in protos.items():
        rret, rparams = rust[name]
        assert rret == _c_to_rust(cret), (name, cret, rret)
        assert len(rparams) == len(cparams), (name, cparams, rparams)
        for k, (ct, rt) in enumerate(zip(cparams, rparams)):
            assert rt == _c_to_rust(ct), "%s parameter %d: C `%s` is Rust `%s`, ffi.rs says `%s`" % (name, k, ct, _c_to_rust(ct), rt)
        res, argtypes = _lib.SYMBOLS[name]
        assert len(argtypes) == len(cparams), (name, "ctypes binds %d parameters, the header declares %d" % (len(argtypes), len(cparams)))
        for k, (ct, at) in enumerate(zip(cparams, argtypes)):
            rt = _c_to_rust(ct)
            if rt in _RUST2CTYPES:                       # by-value scalars must agree exactly; pointers are void* / typed pointers in ctypes
                assert at is getattr(C, _RUST2CTYPES[rt]), (name, k, ct, at)
            else:
                assert at in (C.c_void_p, C.c_char_p) or issubclass(at, C._Pointer), (name, k, ct, at)
        assert res is (C.c_char_p if "char" in cret else (C.c_void_p if cret == "mh_ctx_t" else C.c_int)), (name, res)
    # the callback typedefs and the two structs, field for field
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "marlin_hip.h")).read(), flags=re.S)
    for tname, args in re.findall(r"typedef\s+int\s*\(\*(\w+)\)\s*\(([^)]*)\)\s*;", hdr):
        want = [_c_to_rust(" ".join(" ".join(a.replace("*", " * ").split()).split(" ")[:-1])) for a in args.split(",")]
        m = re.search(r"pub type %s\s*=\s*Option<unsafe extern \"C\" fn\(([^)]*)\)\s*->\s*c_int>;" % tname, ffi, flags=re.S)
        assert m, tname
        got = [" ".join(a.split(":", 1)[1].split()) for a in m.group(1).split(",") if a.strip()]
        assert got == want, (tname, want, got)


def _rust_call_sites(src):
    """[(name, argument count, line)] of every `mh_*(...)` CALL in a Rust source (declarations `fn mh_*` excluded): the
    argument list is split at top-level commas with brackets, strings and `|closure|` bars left alone."""
    src = re.sub(r"//[^\n]*", lambda m: " " * len(m.group(0)), src)
    out = []
    for m in re.finditer(r"\b(mh_[a-z0-9_]+)\s*\(", src):
        if re.search(r"\bfn\s+$", src[:m.start()]):
            continue
        i, depth, args, cur, in_str = m.end(), 1, [], "", False
        while depth:
            ch = src[i]
            if in_str:
                in_str = not (ch == '"' and src[i - 1] != "\\")
            elif ch == '"':
                in_str = True
            elif ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
                if depth == 0:
                    break
            elif ch == "," and depth == 1:
                args.append(cur)
                cur = ""
                i += 1
                continue
            cur += ch
            i += 1
        if cur.strip():
            args.append(cur)
        out.append((m.group(1), len(args), src.count("\n", 0, m.start()) + 1))
    return out


def test_rust_call_sites_pass_what_the_header_declares():
    """No rustc has seen the shim: at least every CALL of an entry point in it (src/, marlin-hip-sys/src/, both tests/ and the
    ark-poly / ark-ec patches) passes as many arguments as include/marlin_hip.h declares, and names an entry point that exists."""
    protos = _c_prototypes()
    files = []
    for d, _, fs in os.walk(os.path.join(ROOT, "shim")):
        files += [os.path.join(d, f) for f in fs if f.endswith((".rs", ".patch"))]
    assert len(files) >= 10, files
    seen = 0
    for f in sorted(files):
        src = open(f).read()
        if f.endswith(".patch"):
            src = "\n".join(l[1:] for l in src.split("\n") if l.startswith("+") and not l.startswith("+++"))
        for name, nargs, line in _rust_call_sites(src):
            if f.endswith("ffi_symbols.rs") and nargs == 0:
                continue                                    # `mh_x as usize`-style address-of uses carry no argument list
            assert name in protos, "%s:%d calls %s, which include/marlin_hip.h does not declare" % (os.path.relpath(f, ROOT), line, name)
            assert nargs == len(protos[name][1]), "%s:%d passes %d arguments to %s; the header declares %d" % (
                os.path.relpath(f, ROOT), line, nargs, name, len(protos[name][1]))
            seen += 1
    assert seen >= 25, seen


def _build_c_example(tmp_path):
    import subprocess
    exe = str(tmp_path / "prove_verify")
    lib_dir = os.path.join(ROOT, "marlin_amd")
    r = subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "prove_verify.c"), "-L" + lib_dir, "-lmarlin_hip", "-Wl,-rpath," + lib_dir, "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_example_builds_against_the_header_and_library(tmp_path):
    """include/marlin_hip.h is plain C99 and a C program that drives index -> prove -> serialize -> verify through it links
    against libmarlin_hip.so; without a device it stops at mh_init with the library's own message."""
    import subprocess
    exe = _build_c_example(tmp_path)
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe, "8"], capture_output=True, text=True)
        assert r.returncode == 1 and "mh_init" in r.stderr and "device" in r.stderr


@pytest.mark.gpu
def test_c_example_proves_and_verifies(tmp_path):
    import subprocess
    r = subprocess.run([_build_c_example(tmp_path), "12"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "verify = 1, verify(wrong input) = 0" in r.stdout


def test_rust_crates_have_no_dependency_cycle():
    """the patched ark-poly depended on marlin-hip, which depends on ark-poly -- cargo rejects that before
    compiling a line.  The hook now lives in the LEAF crate marlin-hip-sys: read the manifests / the patch and check the
    package graph (path dependencies and the [patch] redirections) for cycles."""
    shim = os.path.join(ROOT, "shim")

    def deps(path):
        out, on = [], False
        for line in open(path):
            t = line.strip()
            if t.startswith("["):
                on = t in ("[dependencies]", "[dev-dependencies]")
                continue
            m = re.match(r"^([A-Za-z0-9_-]+)\s*=", t)
            if on and m:
                out.append(m.group(1))
        return out
    graph = {"marlin-hip": deps(os.path.join(shim, "Cargo.toml")), "marlin-hip-sys": deps(os.path.join(shim, "marlin-hip-sys", "Cargo.toml"))}
    patch = open(os.path.join(shim, "vendor", "ark-poly-hip", "radix2_hip.patch")).read()
    added = re.findall(r"^\+([A-Za-z0-9_-]+)\s*=\s*\{", patch, flags=re.M)
    assert added == ["marlin-hip-sys"], added
    # upstream edges that matter (ark-* 0.3 manifests, recalled): none of these reaches ark-poly
    graph.update({"ark-poly": ["ark-ff", "ark-serialize", "ark-std"] + added, "ark-bls12-381": ["ark-ec", "ark-ff", "ark-std"],
                  "ark-bn254": ["ark-ec", "ark-ff", "ark-std"], "ark-ec": ["ark-ff", "ark-serialize", "ark-std"],
                  "ark-ff": ["ark-serialize", "ark-std"], "ark-serialize": ["ark-std"], "ark-std": [],
                  "ark-poly-commit": ["ark-poly", "ark-ec", "ark-ff", "ark-serialize", "ark-std", "ark-relations"],
                  "ark-relations": ["ark-ff", "ark-std"], "ark-marlin": ["ark-poly-commit", "ark-poly", "ark-relations", "ark-ff", "ark-serialize", "ark-std"]})
    assert "ark-poly" not in graph["marlin-hip-sys"] and "marlin-hip" not in graph["marlin-hip-sys"]
    state = {}

    def visit(n, stack):
        if state.get(n) == 2:
            return
        assert state.get(n) != 1, "dependency cycle: %s" % " -> ".join(stack + [n])
        state[n] = 1
        for d in graph.get(n, []):
            visit(d, stack + [n])
        state[n] = 2
    for n in list(graph):
        visit(n, [])


def test_mock_rccl_exports_what_the_native_transport_resolves():
    """tests/mock_rccl/libmock_rccl.so (test infrastructure, built by __graft_entry__.build(); the product loads it only through
    MH_RCCL_LIB) exports every entry point marlin_amd/csrc/rccl_native.h resolves with dlsym -- a renamed symbol on either side
    fails here, on the CPU, not in the GPU suite's multi-rank tests."""
    import ctypes as C
    mock = os.path.join(ROOT, "tests", "mock_rccl", "libmock_rccl.so")
    if not os.path.exists(mock):
        pytest.skip("tests/mock_rccl/libmock_rccl.so not built (python -c 'import __graft_entry__ as g; g.build()')")
    src = open(os.path.join(ROOT, "marlin_amd", "csrc", "rccl_native.h")).read()
    wanted = re.findall(r'RCCL_SYM\(\w+, "(nccl\w+)", (?:true|false)\)', src)
    assert len(wanted) >= 10 and "ncclAllToAll" in wanted and "ncclCommInitRank" in wanted
    lib = C.CDLL(mock)
    for name in wanted:
        assert hasattr(lib, name), "the stand-in lacks %s" % name
    # ... and the product never names the stand-in: it is reached through the environment variable only
    for dirpath, _, files in os.walk(os.path.join(ROOT, "marlin_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".h", ".cpp")):
                assert "libmock_rccl" not in open(os.path.join(dirpath, f)).read(), f
