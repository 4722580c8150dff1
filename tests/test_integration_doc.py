"""CPU: INTEGRATION.md's binding stub is generated from include/gode.h, matches a fresh generation, executes, and every
ctypes struct in it has the size the compiled library reports (a maintainer following the document passes structs
the library reads correctly)."""
import ctypes as C
import importlib.util
import os
import re

from conftest import REPO

import gan_ode_amd._lib as L


def _gen():
    spec = importlib.util.spec_from_file_location("_stubgen", os.path.join(REPO, "scripts", "gen_binding_stub.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _doc_blocks():
    doc = open(os.path.join(REPO, "INTEGRATION.md")).read()
    g = _gen()
    stub = doc[doc.index(g.BEGIN) + len(g.BEGIN):doc.index(g.END)]
    stub = re.search(r"```python\n(.*?)```", stub, flags=re.S).group(1)
    rest = doc[doc.index(g.END):]
    example = re.search(r"```python\n(.*?)```", rest, flags=re.S).group(1)
    return stub, example


def test_binding_stub_is_the_generated_one():
    stub, _ = _doc_blocks()
    assert stub == _gen().generate(), "run: python scripts/gen_binding_stub.py --write"


def test_binding_stub_executes_and_matches_the_compiled_abi():
    stub, example = _doc_blocks()
    ns = {}
    exec(compile(stub, "INTEGRATION.md:stub", "exec"), ns)
    lib = ns["load"](L.LIB_PATH)                       # asserts sizeof(struct) == gode_sizeof(kind) for every kind
    assert lib.gode_version() == L.lib().gode_version()
    kinds = ns["KINDS"]
    assert set(kinds) == set(L._STRUCTS)
    for kind, st in kinds.items():
        mine = L._STRUCTS[kind]
        assert C.sizeof(st) == C.sizeof(mine) == lib.gode_sizeof(kind)
        # same field offsets as the mirror the product itself uses
        assert [(n, getattr(st, n).offset) for n, *_ in st._fields_] == \
               [(n, getattr(mine, n).offset) for n, *_ in mine._fields_][:len(st._fields_)], st.__name__
    # the example function compiles against the stub's names (it needs a GPU to run: tests/test_gpu_api.py)
    exec(compile(example, "INTEGRATION.md:example", "exec"), ns)
    assert callable(ns["sample_z_m"])
