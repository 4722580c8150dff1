"""Generates the ctypes binding stub of INTEGRATION.md section 2 from include/gode.h, so the document cannot drift
from the header: every `typedef struct` becomes a ctypes.Structure with the same fields in the same order, every
`GODE_OP_*` kind is listed with its struct, and the stub checks itself against gode_sizeof() when executed.

  python scripts/gen_binding_stub.py            # prints the stub
  python scripts/gen_binding_stub.py --write    # rewrites the block between the markers in INTEGRATION.md

tests/test_integration_doc.py asserts that the committed block equals this output and executes it.
"""
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "gode.h")
DOC = os.path.join(REPO, "INTEGRATION.md")
BEGIN, END = "<!-- BEGIN GENERATED BINDING STUB (scripts/gen_binding_stub.py) -->", "<!-- END GENERATED BINDING STUB -->"

SCALARS = {"int32_t": "i32", "int64_t": "i64", "float": "f32", "int": "C.c_int"}


def parse_structs(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    out = []
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        name, body = m.group(3), m.group(2)
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            mm = re.match(r"^(const\s+)?(\w+)\s*(\*?)\s*(.*)$", decl)
            base, star, rest = mm.group(2), mm.group(3), mm.group(4)
            for d in rest.split(","):
                d = d.strip()
                ptr = bool(star) or d.startswith("*")
                d = d.lstrip("* ")
                arr = re.match(r"^(\w+)\[(\d+)\]$", d)
                if ptr:
                    ctype = "ptr"
                elif base in SCALARS:
                    ctype = SCALARS[base]
                else:
                    ctype = base          # nested struct, declared earlier in the header
                if arr:
                    fields.append((arr.group(1), f"{ctype} * {arr.group(2)}"))
                else:
                    fields.append((d, ctype))
        out.append((name, fields))
    return out


def parse_kinds(text):
    m = re.search(r"enum\s*\{\s*(GODE_OP_IGEMM.*?)\}", text, flags=re.S)
    return [(k, int(v)) for k, v in re.findall(r"(GODE_OP_\w+)\s*=\s*(\d+)", m.group(1))]


KIND_STRUCT = {"GODE_OP_IGEMM": "gode_igemm_op", "GODE_OP_WGRAD": "gode_wgrad_op",
               "GODE_OP_BN_FINALIZE": "gode_bn_finalize_op", "GODE_OP_BN_BWD": "gode_bn_bwd_op",
               "GODE_OP_ODE_FWD": "gode_ode_fwd_op", "GODE_OP_ODE_BWD": "gode_ode_bwd_op", "GODE_OP_BCE": "gode_bce_op",
               "GODE_OP_ADAM": "gode_adam_op", "GODE_OP_PACK": "gode_pack_op", "GODE_OP_ODERNN_FWD": "gode_odernn_fwd_op",
               "GODE_OP_ODERNN_BWD": "gode_odernn_bwd_op", "GODE_OP_BN_APPLY": "gode_bn_apply_op",
               "GODE_OP_COL2IM": "gode_col2im_op"}


def generate():
    text = open(HEADER).read()
    structs = parse_structs(text)
    kinds = parse_kinds(text)
    names = {n for n, _ in structs}
    lines = ["import ctypes as C", "", "i32, i64, f32, ptr = C.c_int32, C.c_int64, C.c_float, C.c_void_p", ""]
    for name, fields in structs:
        lines.append(f"class {name}(C.Structure):")
        body = ", ".join(f'("{f}", {t})' for f, t in fields)
        # wrap at ~110 columns
        row, rows = "    _fields_ = [", []
        for item in body.split("), ("):
            item = item if item.startswith("(") else "(" + item
            item = item if item.endswith(")") else item + ")"
            if len(row) + len(item) + 2 > 112:
                rows.append(row.rstrip())
                row = "                "
            row += item + ", "
        rows.append(row.rstrip(", ") + "]")
        lines += rows + [""]
    lines.append("KINDS = {0: gode_conv_geom, " + ", ".join(f"{v}: {KIND_STRUCT[k]}" for k, v in kinds) + "}")
    lines += ["", "",
              "def load(path=\"gan-ode_amd/lib/libgode.so\"):",
              "    lib = C.CDLL(path)",
              "    lib.gode_sizeof.argtypes = [C.c_int]",
              "    for kind, st in KINDS.items():            # the mirror must match the compiled ABI, field for field",
              "        assert lib.gode_sizeof(kind) == C.sizeof(st), (kind, st.__name__)",
              "    for fn in (\"gode_igemm\", \"gode_wgrad\", \"gode_bn_finalize\", \"gode_bn_bwd\", \"gode_bn_apply\", \"gode_col2im\", \"gode_ode_fwd\",",
              "               \"gode_ode_bwd\", \"gode_odernn_fwd\", \"gode_odernn_bwd\", \"gode_bce_logits\", \"gode_adam_l2\"):",
              "        getattr(lib, fn).argtypes = [ptr, ptr]          # (const op struct*, hipStream_t)",
              "        getattr(lib, fn).restype = C.c_int              # 0 ok, <0 GODE_E_*, >0 hipError_t",
              "    return lib"]
    missing = [s for s in KIND_STRUCT.values() if s not in names]
    assert not missing, missing
    return "\n".join(lines) + "\n"


def main():
    stub = generate()
    if "--write" in sys.argv:
        doc = open(DOC).read()
        i, j = doc.index(BEGIN), doc.index(END)
        doc = doc[:i] + BEGIN + "\n```python\n" + stub + "```\n" + doc[j:]
        open(DOC, "w").write(doc)
        print("rewrote", DOC)
    else:
        sys.stdout.write(stub)


if __name__ == "__main__":
    main()
