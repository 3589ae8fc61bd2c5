"""Build step of oracle/_ref (test infrastructure only): read the marching-cubes case tables out of
the reference checkout, where they lie (src/kfusion/marching_cubes.cpp:86-343 triTable,
:344-354 numVertsTable), into the binary oracle/_ref/mc_tables.bin = int32 little-endian
triTable[256][16] followed by numVertsTable[256].  The binary is git-ignored and travels to the GPU
box like the other _ref artefacts; no text of the reference enters the repository.

usage: python extract_ref_mc_tables.py <reference root> <output file>"""
import re
import struct
import sys


def array_body(text, name):
    m = re.search(r"const\s+int\s+" + name + r"\s*(\[\d+\])+\s*=\s*\{", text)
    if not m:
        raise SystemExit("table %s not found" % name)
    depth, i = 1, m.end()
    while depth:
        depth += {"{": 1, "}": -1}.get(text[i], 0)
        i += 1
    body = re.sub(r"//[^\n]*|/\*.*?\*/", " ", text[m.end():i - 1], flags=re.S)
    return [int(t, 0) for t in re.findall(r"-?(?:0[xX][0-9a-fA-F]+|\d+)", body)]


def main(ref, out):
    text = open(ref + "/src/kfusion/marching_cubes.cpp").read()
    tri, nv = array_body(text, "triTable"), array_body(text, "numVertsTable")
    if len(tri) != 256 * 16 or len(nv) != 256:
        raise SystemExit("unexpected table sizes %d / %d" % (len(tri), len(nv)))
    for c in range(256):  # the two tables must agree: numVerts = entries before the first -1
        row = tri[16 * c:16 * c + 16]
        n = row.index(-1) if -1 in row else 16
        if n != nv[c]:
            raise SystemExit("case %d: triTable has %d vertices, numVertsTable says %d" % (c, n, nv[c]))
    with open(out, "wb") as f:
        f.write(struct.pack("<%di" % (256 * 16 + 256), *(tri + nv)))
    print("wrote", out)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
