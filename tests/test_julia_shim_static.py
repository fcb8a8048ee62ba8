"""Static checks of the reference-side binding (waveletsext.jl_amd/julia/) against include/waveletsext_hip.h.

There is no Julia in the image, so the shim cannot be executed; what can be checked without it:
  * julia/libwx.jl (the only file with `ccall`s) is exactly what tools/gen_julia_bindings.py generates from the header, and
    -- parsed independently of the generator -- every `ccall` names an exported symbol with the header's return type,
    argument count and argument types (Ptr{Float64} <-> double*, Int64 <-> int64_t, Cint <-> int, ...);
  * every call site `wx_name(...)` in julia/WaveletsExtHIP.jl passes as many arguments as the header's prototype has (one
    more, the element type first, for the `wx_name(T, ...)` forms), and no `ccall` hides outside libwx.jl;
  * every entry family of the header has a caller in the shim (the test hook excepted), so INTEGRATION.md's table has no row
    without a Julia method; the methods SURVEY 8(b) lists exist with a HIP-typed signature;
  * the ctypes table of waveletsext.jl_amd/_lib.py agrees with the header as well."""
import ctypes
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import abi_header  # noqa: E402
import gen_julia_bindings  # noqa: E402

JULIA = os.path.join(ROOT, "waveletsext.jl_amd", "julia")
HOOKS = set()          # (the dispatch override of the parity suite left the public header in round 5: csrc/wx_debug.h)


def _protos():
    return {p.name: p for p in abi_header.parse()}


def _strip_comments(src):
    out = []
    for line in src.splitlines():
        # a '#' inside a string literal does not occur in the shim except in messages without parentheses; cut at the
        # first '#' that is not inside double quotes
        q = False
        for i, ch in enumerate(line):
            if ch == '"':
                q = not q
            elif ch == "#" and not q:
                line = line[:i]
                break
        out.append(line)
    return "\n".join(out)


def _split_args(s):
    """top-level comma split of the text between a call's parentheses"""
    args, depth, cur, q = [], 0, "", False
    for ch in s:
        if ch == '"':
            q = not q
        if not q:
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
            elif ch == "," and depth == 0:
                args.append(cur.strip())
                cur = ""
                continue
        cur += ch
    if cur.strip():
        args.append(cur.strip())
    return args


def _calls(src, pattern=r"\bwx_[a-z0-9_]+"):
    """(name, [args]) of every call `name(...)` in src"""
    res = []
    for m in re.finditer(pattern + r"\(", src):
        name = m.group(0)[:-1]
        i, depth = m.end(), 1
        while depth:
            ch = src[i]
            depth += ch == "("
            depth -= ch == ")"
            i += 1
        res.append((name, _split_args(src[m.end():i - 1])))
    return res


def test_header_parses_completely():
    protos = _protos()
    txt = re.sub(r"/\*.*?\*/", "", open(abi_header.HEADER).read(), flags=re.S)
    declared = set(re.findall(r"\b(wx_[a-z0-9_]+)\s*\(", txt))
    assert declared == set(protos), declared ^ set(protos)
    assert len(protos) >= 120


def test_generated_bindings_are_up_to_date():
    cur = open(os.path.join(JULIA, "libwx.jl")).read()
    assert cur == gen_julia_bindings.generate(), "run python tools/gen_julia_bindings.py"


def test_every_ccall_matches_the_header(wx):
    protos = _protos()
    lib = ctypes.CDLL(wx.LIB_PATH)
    src = open(os.path.join(JULIA, "libwx.jl")).read()
    seen = set()
    for name, args in _calls(_strip_comments(src), r"\bccall"):
        m = re.match(r"\(:(\w+), LIB\)$", args[0])
        assert m, args[0]
        sym = m.group(1)
        assert sym in protos, "ccall of a symbol the header does not declare: " + sym
        assert hasattr(lib, sym), "not exported by the built library: " + sym
        p = protos[sym]
        assert args[1] == p.julia_ret, (sym, args[1], p.julia_ret)
        tup = args[2]
        assert tup.startswith("(") and tup.endswith(")")
        types = _split_args(tup[1:-1])
        assert types == p.julia_args, (sym, types, p.julia_args)
        assert len(args) - 3 == len(p.args), (sym, "values passed", len(args) - 3, "prototype", len(p.args))
        assert args[3:] == [a for _, a in p.args], (sym, args[3:])
        seen.add(sym)
    assert seen == set(protos), set(protos) - seen
    # the wrapper `name(args) = ccall(...)` forwards exactly its own parameters
    for m in re.finditer(r"^(wx_\w+)\(([^)]*)\) =\n    ccall\(\(:(\w+),", src, flags=re.M):
        assert m.group(1) == m.group(3)
        assert _split_args(m.group(2)) == [a for _, a in protos[m.group(1)].args]


def test_shim_call_sites_have_the_prototypes_arity():
    protos = _protos()
    fams = {}
    for p in protos.values():
        fam, suf = p.family
        if suf:
            fams.setdefault(fam, []).append(p)
    src = _strip_comments(open(os.path.join(JULIA, "WaveletsExtHIP.jl")).read())
    assert "ccall" not in src, "every ccall belongs in the generated libwx.jl"
    used = set()
    calls = _calls(src)
    assert len(calls) >= 70
    for name, args in calls:
        if name in fams:                       # wx_name(T, ...) -> wx_name_f64 / _f32
            assert args and re.match(r"^(T|Float64|Float32)$", args[0]), (name, args[:1])
            for p in fams[name]:
                assert len(args) - 1 == len(p.args), (name, len(args) - 1, len(p.args))
            if args[0] == "T":
                used.update(p.name for p in fams[name])
            else:
                used.add(name + ("_f64" if args[0] == "Float64" else "_f32"))
        else:
            assert name in protos, "call of an entry point the header does not declare: " + name
            assert len(args) == len(protos[name].args), (name, len(args), len(protos[name].args))
            used.add(name)
    missing = set(protos) - used - HOOKS
    assert not missing, "entry points without a caller in WaveletsExtHIP.jl: %s" % sorted(missing)


def test_shim_defines_the_reference_methods():
    """the method names of SURVEY 8(b), each with a HIP-typed data argument"""
    src = open(os.path.join(JULIA, "WaveletsExtHIP.jl")).read()
    # names generated by the @eval loops: the tuples of `for (f, f!, fall, ...) in ((:a, :a!, :aall, ...), ...)`
    generated = set()
    for tab in re.findall(r"^for \(f, f!, fall, \w+, \w+\) in \((.*?)\)\)\n", src, flags=re.S | re.M):
        for tup in re.findall(r"\(([^()]*)\)", tab + ")"):
            generated.update(re.findall(r":([a-z]+!?)", tup)[:3])
    explicit = set(re.findall(r"^\s*(?:@eval )?(?:function )?(?:WaveletsExt\.\w+\.)?([a-z_]+!?)\((?:\w+|x̂)::HIP", src, flags=re.M))
    have = generated | explicit
    need = """wpt wpt! iwpt iwpt! wpd wpd! iwpd iwpd! wptall iwptall wpdall iwpdall dwtall idwtall dwt idwt
              sdwt sdwt! isdwt isdwt! swpt swpt! iswpt iswpt! swpd swpd! iswpd iswpd! sdwtall isdwtall swptall iswptall
              swpdall iswpdall acdwt acdwt! iacdwt iacdwt! acwpt acwpt! iacwpt iacwpt! acwpd acwpd! iacwpd iacwpd!
              acdwtall iacdwtall acwptall iacwptall acwpdall iacwpdall getbasiscoef getbasiscoefall tree_costs
              bestbasistree bestbasistreeall bestbasis_treeselection energy_map discriminant_power noisest surethreshold
              relerrorthreshold denoiseall siwpdall""".split()
    missing = [n for n in need if n not in have]
    assert not missing, missing
    # `!` forms return the array they were given: each one ends in `return <first argument>`
    nbang = 0
    for m in re.finditer(r"^( *)(?:@eval )?function (\$?\w+!)\((\w+|x̂)::HIP.*?^\1end$", src, flags=re.S | re.M):
        if m.group(2) in ("bestbasistreeall!", "thresholdall!"):
            continue
        assert re.search(r"return %s\n%send$" % (re.escape(m.group(3)), m.group(1)), m.group(0)), m.group(2)
        nbang += 1
    assert nbang >= 9
    # no method may widen an argument the reference dispatches on (that is what makes an ambiguity): the transform methods
    # never use a Union of level and tree, and the `!` forms pin the dimensionality of the output they dispatch on
    assert "Union{Integer,BitVector}" not in src
    for m in re.finditer(r"function (\$?\w+!)\((\w+|x̂)::HIP\{T\}", src):
        assert m.group(1) in ("thresholdall!",), "`!` form without a dimension on its HIP argument: " + m.group(1)


def test_integration_table_covers_every_family():
    protos = _protos()
    txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    mentioned = set()
    for tok in re.findall(r"wx_[a-z0-9_]+(?:\{[^}]*\})?\*?", txt):
        base = re.sub(r"(\{[^}]*\}|\*)$", "", tok).rstrip("_")
        mentioned.add(base)
    for name, p in protos.items():
        fam, _ = p.family
        if name in HOOKS:
            continue
        assert any(fam == m or name == m or fam.startswith(m) for m in mentioned), "INTEGRATION.md never mentions " + fam
    for m in mentioned:
        assert any(n.startswith(m) for n in protos), "INTEGRATION.md names an entry point the header lacks: " + m


def test_ctypes_table_matches_the_header(wx):
    """waveletsext.jl_amd/_lib.py declares argtypes by hand: compare them with the header too"""
    from waveletsext_jl_amd import _lib
    code = {ctypes.c_void_p: "P", ctypes.c_int: "I", ctypes.c_int64: "L", ctypes.c_double: "D",
            ctypes.POINTER(ctypes.c_void_p): "PP"}
    L = _lib.lib()
    checked = 0
    for name, p in _protos().items():
        fn = getattr(L, name)
        if fn.argtypes is None:
            continue
        got = [code[t] for t in fn.argtypes]
        want = [abi_header.CTYPES[t][1] for t, _ in p.args]
        assert got == want, (name, got, want)
        checked += 1
    assert checked >= 100


def test_shim_blocks_and_brackets_balance():
    """no Julia to parse the file: at least every block opener has its `end` and every bracket closes (tokens inside
    strings, comments and brackets -- `x[1:end-1]`, generators -- do not count)"""
    for fname in ("WaveletsExtHIP.jl", "libwx.jl"):
        src = _strip_comments(open(os.path.join(JULIA, fname)).read())
        src = re.sub(r'"""(.|\n)*?"""', '""', src)
        src = re.sub(r'"(\\.|[^"\\])*"', '""', src)
        depth, stack, flat = 0, [], []
        for ch in src:
            if ch in "([{":
                stack.append(ch)
            elif ch in ")]}":
                assert stack and "([{".index(stack.pop()) == ")]}".index(ch), fname + ": unbalanced " + ch
            flat.append(ch if not stack and ch not in ")]}" else " ")
        assert not stack, fname
        toks = re.findall(r"(?<![\w!:.@])(function|if|for|while|begin|struct|module|let|do|try|quote|macro|end)(?![\w!])",
                          "".join(flat))
        opens = sum(t != "end" for t in toks)
        assert opens == toks.count("end"), (fname, opens, toks.count("end"))
        assert opens > (40 if fname == "WaveletsExtHIP.jl" else -1)
