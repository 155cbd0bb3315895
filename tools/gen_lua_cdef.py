#!/usr/bin/env python3
"""Regenerates the ffi.cdef block of depth-estimation_amd/lua/dfe_ffi.lua from include/dfe.h (every prototype of the C ABI),
so that the LuaJIT binding cannot drift from the header.  usage: python tools/gen_lua_cdef.py [--check]"""
import os, re, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
hdr = open(os.path.join(root, "include", "dfe.h")).read()
lua_path = os.path.join(root, "depth-estimation_amd", "lua", "dfe_ffi.lua")


def cdef_from_header(h):
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)          # comments
    h = re.sub(r"//[^\n]*", "", h)
    lines = [l for l in h.splitlines() if not l.strip().startswith("#") and 'extern "C"' not in l and l.strip() not in ("}",)]
    text = "\n".join(lines)
    out = ["typedef struct dfe_ctx dfe_ctx;"]
    # the struct types the prototypes use (dfe_radial_params, dfe_filter_layer): LuaJIT rejects the whole cdef when one is missing
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{([^}]*)\}\s*(\w+)\s*;", text):
        body = " ".join(re.sub(r"\s+", " ", m.group(2)).strip().split())
        out.append("typedef struct %s { %s } %s;" % (m.group(1), body, m.group(3)))
    for m in re.finditer(r"(?:^|\n)\s*((?:const\s+)?[A-Za-z_][A-Za-z0-9_]*(?:\s*\*)?\s+\*?dfe_[a-z0-9_]+\s*\([^;{}]*\))\s*;", text):
        proto = re.sub(r"\s+", " ", m.group(1)).strip()
        out.append(proto + ";")
    return "\n".join(out)


cdef = cdef_from_header(hdr)
lua = open(lua_path).read()
a, b = lua.index("ffi.cdef[[\n") + len("ffi.cdef[[\n"), lua.index("]]\n")
new = lua[:a] + cdef + "\n" + lua[b:]
if "--check" in sys.argv:
    sys.exit(0 if new == lua else 1)
open(lua_path, "w").write(new)
print("%d prototypes" % cdef.count(");"))
