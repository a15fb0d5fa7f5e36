"""Lists every C function the reference's Cython binding declares in its `cdef extern from "wholememory/..."` blocks —
i.e. every symbol `wholememory_binding.pyx` needs from the shared library it is linked against — into
tests/golden/reference_binding_symbols.txt (data: one name per line).  Runs in the BUILD container only, where
/root/reference exists; the list travels, the reference does not.

    python tests/golden/make_binding_symbols.py
"""
import os
import re

PYX = "/root/reference/python/pylibwholegraph/pylibwholegraph/binding/wholememory_binding.pyx"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_binding_symbols.txt")


def extern_functions(text):
    names, inside = set(), False
    for line in text.splitlines():
        if line.startswith("cdef extern from"):
            inside = '"wholememory/' in line
            continue
        if inside and line and not line[0].isspace():
            inside = False
        if inside:
            m = re.match(r"\s+cdef\s+[\w\s\*]+?\s+\**(\w+)\s*\(", line)
            if m:
                names.add(m.group(1))
    return sorted(names)


if __name__ == "__main__":
    names = extern_functions(open(PYX).read())
    with open(OUT, "w") as f:
        f.write("# C functions declared by the reference's wholememory_binding.pyx (cdef extern blocks), one per line\n")
        f.write("\n".join(names) + "\n")
    print(len(names), "symbols ->", OUT)
