#!/usr/bin/env python3
"""Extract the reference's bundled sparse datasets into small CSC fixtures.

Run IN THE BUILD CONTAINER (needs /root/reference); the outputs are committed:

    python tests/golden/make_fixtures.py

Reads  /root/reference/data/hawaiibirds.rda  (dgCMatrix 183 x 1183, nnz 30815)
       /root/reference/data/movielens.rda    (dgCMatrix 3867 x 610, nnz 75238)
and writes tests/golden/{hawaiibirds,movielens}.npz with int32 `p`, `i`, float64 `x`
and `shape`.  These are DATA files the reference's own tests use (BASELINE configs C1, C3);
no reference source is copied.

The .rda files are xz-compressed R serialization streams ("RDX3", XDR big-endian).  This
is a minimal reader of that public format: just enough SEXP types for an S4 dgCMatrix inside
a pairlist.
"""
import lzma
import os
import struct
import sys

import numpy as np

REF = "/root/reference/data"
OUT = os.path.dirname(os.path.abspath(__file__))


class XdrReader:
    def __init__(self, buf):
        self.b, self.o, self.refs = buf, 0, []

    def i32(self):
        v = struct.unpack_from(">i", self.b, self.o)[0]
        self.o += 4
        return v

    def raw(self, n):
        v = self.b[self.o:self.o + n]
        self.o += n
        return v

    def item(self):
        flags = self.i32()
        t = flags & 0xFF
        has_attr, has_tag = bool(flags & 0x200), bool(flags & 0x400)
        if t == 254:                       # NILVALUE_SXP
            return None
        if t == 253 or t == 242 or t == 241:   # global/empty/base env
            return ("env", t)
        if t == 255:                       # REFSXP
            return self.refs[(flags >> 8) - 1]
        if t == 1:                         # SYMSXP
            name = self.item()
            self.refs.append(("sym", name))
            return ("sym", name)
        if t == 2 or t == 6:               # LISTSXP / LANGSXP
            out = []
            while True:
                attr = self.item() if has_attr else None
                tag = self.item() if has_tag else None
                car = self.item()
                out.append((tag[1] if tag else None, car))
                flags = self.i32()
                t = flags & 0xFF
                has_attr, has_tag = bool(flags & 0x200), bool(flags & 0x400)
                if t == 254:
                    return ("list", out)
                if t not in (2, 6):
                    raise ValueError("unexpected pairlist tail type %d" % t)
        if t == 9:                         # CHARSXP
            n = self.i32()
            return None if n == -1 else self.raw(n).decode("utf-8", "replace")
        if t == 10 or t == 13:             # LGLSXP / INTSXP
            n = self.i32()
            v = np.frombuffer(self.raw(4 * n), dtype=">i4").astype(np.int32)
            return self._attr(v, has_attr)
        if t == 14:                        # REALSXP
            n = self.i32()
            v = np.frombuffer(self.raw(8 * n), dtype=">f8").astype(np.float64)
            return self._attr(v, has_attr)
        if t == 16 or t == 19:             # STRSXP / VECSXP
            n = self.i32()
            v = [self.item() for _ in range(n)]
            return self._attr(v, has_attr)
        if t == 25:                        # S4SXP: attributes carry the slots
            attrs = self.item() if has_attr else ("list", [])
            return ("s4", dict(attrs[1]))
        if t == 238:                       # ALTREP: (info, state, attr) -- expand compact seqs
            info, state, attr = self.item(), self.item(), self.item()
            cls = info[1][0][1][1] if info else ""
            if cls in ("compact_intseq", "compact_realseq"):
                n, start, step = (int(state[0]), state[1], state[2])
                return (start + step * np.arange(n)).astype(np.int32 if cls == "compact_intseq" else np.float64)
            if cls == "wrap_integer" or cls == "wrap_real" or cls == "wrap_string":
                return state[0] if isinstance(state, list) else state
            return state
        raise ValueError("unsupported SEXP type %d at offset %d" % (t, self.o))

    def _attr(self, v, has_attr):
        if has_attr:
            self.item()
        return v


def read_rda(path):
    buf = lzma.decompress(open(path, "rb").read())
    assert buf[:5] == b"RDX3\n" and buf[5:7] == b"X\n", buf[:8]
    r = XdrReader(buf)
    r.o = 7
    r.i32(); r.i32(); r.i32()              # format version, R version, min R version
    n = r.i32(); r.raw(n)                  # native encoding
    top = r.item()
    return dict(top[1])


def main():
    for name in ("hawaiibirds", "movielens"):
        objs = read_rda(os.path.join(REF, name + ".rda"))
        obj = objs[name] if name in objs else list(objs.values())[0]
        if isinstance(obj, list):          # movielens ships list(ratings=<dgCMatrix>, ...)? pick the S4
            obj = [o for o in obj if isinstance(o, tuple) and o[0] == "s4"][0]
        if isinstance(obj, tuple) and obj[0] == "list":
            obj = [c for _, c in obj[1] if isinstance(c, tuple) and c[0] == "s4"][0]
        slots = obj[1]
        dim = np.asarray(slots["Dim"], dtype=np.int32)
        p = np.asarray(slots["p"], dtype=np.int32)
        i = np.asarray(slots["i"], dtype=np.int32)
        x = np.asarray(slots["x"], dtype=np.float64)
        assert p.shape[0] == dim[1] + 1 and p[-1] == i.shape[0] == x.shape[0]
        np.savez_compressed(os.path.join(OUT, name + ".npz"), shape=dim, p=p, i=i, x=x)
        print(name, "Dim", tuple(dim), "nnz", x.shape[0], "x range", x.min(), x.max())


if __name__ == "__main__":
    sys.exit(main())
