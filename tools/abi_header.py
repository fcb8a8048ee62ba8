"""Parser of include/waveletsext_hip.h: the one description of the C ABI that the bindings are checked against
(tools/gen_julia_bindings.py generates julia/libwx.jl from it; tests/test_julia_shim_static.py and
tests/test_abi_cpu.py compare the Julia shim and the ctypes table with it)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "waveletsext_hip.h")

# C type (const dropped, whitespace normalised) -> (Julia ccall type, ctypes code used in waveletsext.jl_amd/_lib.py)
CTYPES = {
    "double *": ("Ptr{Float64}", "P"),
    "float *": ("Ptr{Float32}", "P"),
    "uint8_t *": ("Ptr{UInt8}", "P"),
    "int32_t *": ("Ptr{Int32}", "P"),
    "int64_t *": ("Ptr{Int64}", "P"),
    "void *": ("Ptr{Cvoid}", "P"),
    "void **": ("Ptr{Ptr{Cvoid}}", "PP"),
    "char *": ("Cstring", "S"),
    "int64_t": ("Int64", "L"),
    "int": ("Cint", "I"),
    "double": ("Float64", "D"),
    "void": ("Cvoid", "V"),
}


def _norm(t):
    t = re.sub(r"\bconst\b", " ", t)
    t = re.sub(r"\s+", " ", t.replace("*", " * ")).strip()
    t = re.sub(r"\* \*", "**", t)
    return t


class Proto:
    def __init__(self, name, ret, args):
        self.name, self.ret, self.args = name, ret, args      # args: list of (ctype, argname)

    @property
    def julia_ret(self):
        return CTYPES[self.ret][0]

    @property
    def julia_args(self):
        return [CTYPES[t][0] for t, _ in self.args]

    @property
    def family(self):
        """name without the element-type suffix, and the suffix ('' when the entry has none)"""
        m = re.match(r"(.*)_(f64|f32)$", self.name)
        return (m.group(1), m.group(2)) if m else (self.name, "")


def parse(path=HEADER):
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    txt = re.sub(r"^\s*#.*$", " ", txt, flags=re.M)
    txt = txt.replace('extern "C" {', " ")
    protos = []
    for stmt in txt.split(";"):
        m = re.search(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(wx_[a-z0-9_]+)\s*\((.*)\)\s*$", stmt.strip(), flags=re.S)
        if not m:
            continue
        ret, name, arglist = _norm(m.group(1)), m.group(2), m.group(3)
        args = []
        if arglist.strip() != "void":
            for a in arglist.split(","):
                am = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)$", a.strip(), flags=re.S)
                args.append((_norm(am.group(1)), am.group(2)))
        for t in [ret] + [t for t, _ in args]:
            if t not in CTYPES:
                raise ValueError("unmapped C type %r in %s" % (t, name))
        protos.append(Proto(name, ret, args))
    return protos


if __name__ == "__main__":
    for p in parse():
        print(p.name, p.julia_ret, p.julia_args)
