"""Minimal reader for R serialization streams (.rda / .RDS, XDR format, versions 2 and 3).

Pure-Python XDR parser: it reads numbers, strings, lists and attributes and executes
nothing from the file.  Used only to turn the reference's stored data/outputs into
small numeric fixtures under tests/golden/ (see tools/make_golden.py).
"""
import gzip, bz2, lzma, struct, io
import numpy as np


def _open_any(path):
    raw = open(path, 'rb').read()
    if raw[:2] == b'\x1f\x8b':
        return gzip.decompress(raw)
    if raw[:3] == b'BZh':
        return bz2.decompress(raw)
    if raw[:6] == b'\xfd7zXZ\x00':
        return lzma.decompress(raw)
    return raw


class _R:
    def __init__(self, buf):
        self.b = io.BytesIO(buf)
        self.refs = []

    def i32(self):
        return struct.unpack('>i', self.b.read(4))[0]

    def f64(self):
        return struct.unpack('>d', self.b.read(8))[0]

    def length(self):
        n = self.i32()
        if n == -1:
            hi, lo = self.i32(), self.i32()
            n = (hi << 32) + lo
        return n

    def item(self):
        flags = self.i32()
        t = flags & 0xFF
        has_attr = bool(flags & (1 << 9))
        has_tag = bool(flags & (1 << 10))
        if t == 254:      # NILVALUE
            return None
        if t == 253:      # global env
            return '<globalenv>'
        if t == 255:      # REFSXP
            idx = flags >> 8
            if idx == 0:
                idx = self.i32()
            return self.refs[idx - 1]
        if t == 1:        # SYMSXP
            s = self.item()
            self.refs.append(s)
            return s
        if t in (2, 6):      # LISTSXP pairlist / LANGSXP
            out = []
            while True:
                attr = self.item() if has_attr else None
                tag = self.item() if has_tag else None
                car = self.item()
                out.append((tag, car))
                flags = self.i32()
                t2 = flags & 0xFF
                if t2 == 254:
                    break
                if t2 not in (2, 6):
                    raise ValueError('unexpected pairlist tail type %d' % t2)
                has_attr = bool(flags & (1 << 9))
                has_tag = bool(flags & (1 << 10))
            return out
        if t == 9:        # CHARSXP
            n = self.i32()
            if n == -1:
                return None
            return self.b.read(n).decode('utf-8', 'replace')
        if t == 10:       # LGLSXP
            n = self.length()
            v = np.frombuffer(self.b.read(4 * n), dtype='>i4').astype(np.int32)
        elif t == 13:     # INTSXP
            n = self.length()
            v = np.frombuffer(self.b.read(4 * n), dtype='>i4').astype(np.int32)
        elif t == 14:     # REALSXP
            n = self.length()
            v = np.frombuffer(self.b.read(8 * n), dtype='>f8').astype(np.float64)
        elif t == 16:     # STRSXP
            n = self.length()
            v = [self.item() for _ in range(n)]
        elif t == 19:     # VECSXP
            n = self.length()
            v = [self.item() for _ in range(n)]
        elif t == 238:    # ALTREP
            info = self.item(); state = self.item(); attr = self.item()
            cls = info[0][1] if isinstance(info, list) else info
            if cls == 'compact_intseq':
                n, start, step = int(state[0]), int(state[1]), int(state[2])
                return {'value': np.arange(n, dtype=np.int64) * step + start, 'attr': {}}
            if cls == 'compact_realseq':
                n, start, step = int(state[0]), state[1], state[2]
                return {'value': np.arange(n) * step + start, 'attr': {}}
            raise ValueError('unsupported ALTREP class %r' % (cls,))
        else:
            raise ValueError('unsupported SEXP type %d' % t)
        attrs = {}
        if has_attr:
            for tag, val in self.item():
                attrs[tag] = val
        return {'value': v, 'attr': attrs}


def _header(r):
    fmt = r.b.read(2)
    if fmt != b'X\n':
        raise ValueError('not an XDR serialization stream')
    version = r.i32(); r.i32(); r.i32()
    if version == 3:
        n = r.i32(); r.b.read(n)


def read_rds(path):
    r = _R(_open_any(path))
    _header(r)
    return r.item()


def read_rda(path):
    buf = _open_any(path)
    if buf[:5] not in (b'RDX2\n', b'RDX3\n'):
        raise ValueError('not an RDX2/3 file')
    r = _R(buf[5:])
    _header(r)
    return {tag: val for tag, val in r.item()}


def simplify(obj):
    """R object -> numpy array (honouring dim) / dict (named list) / list."""
    if not isinstance(obj, dict) or 'value' not in obj:
        return obj
    v, a = obj['value'], obj['attr']
    def unwrap(x):
        return x['value'] if isinstance(x, dict) and 'value' in x else x
    if isinstance(v, list):
        vals = [simplify(x) for x in v]
        names = unwrap(a.get('names'))
        if names is not None and len(names) == len(vals):
            return dict(zip(names, vals))
        return vals
    dim = unwrap(a.get('dim'))
    if dim is not None:
        return np.asarray(v).reshape([int(d) for d in dim], order='F')
    return v
