#!/usr/bin/env python3
"""Linker version script from the public header(s): the shared library exports exactly the functions they declare.

usage: gen_exports.py [-DNAME ...] header.h [header2.h ...] > exports.map
(`#ifdef NAME ... #endif` blocks are kept only when -DNAME is given; include guards and __cplusplus blocks are left alone)"""
import re
import sys


def declared(path, defines):
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)

    def cond(m):
        return m.group(2) if m.group(1) in defines else ""
    src = re.sub(r"#ifdef (ZK_[A-Z_]+)\n(.*?)#endif", cond, src, flags=re.S)
    return re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{()]*\)\s*;", src)


def main():
    defines = {a[2:] for a in sys.argv[1:] if a.startswith("-D")}
    names = []
    for h in [a for a in sys.argv[1:] if not a.startswith("-D")]:
        for n in declared(h, defines):
            if n not in names:
                names.append(n)
    print("{\n  global:")
    for n in names:
        print("    %s;" % n)
    print("  local: *;\n};")


if __name__ == "__main__":
    main()
