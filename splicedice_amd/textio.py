"""Host-side text I/O shared by the sub-commands: junction names, manifests, count/PS tables.

The file formats are the reference's inter-stage contract; nothing
here is on the device path.
"""
import numpy as np

STRAND_CODE = {"+": 0, "-": 1}
STRAND_SYM = ("+", "-")


def junction_name(j):
    """(chrom, left, right, strand) -> 'chrom:left-right:strand' (SPLICEDICE.py:312-314)."""
    return f"{j[0]}:{j[1]}-{j[2]}:{j[3]}"


def parse_junction_name(name):
    """Inverse of junction_name (counts_to_ps.py:53-56).  Chromosome names without ':' only,
    as in the reference (a 3-way split)."""
    chromosome, coords, strand = name.split(":")
    start, end = (int(x) for x in coords.split("-"))
    return (chromosome, start, end, strand)


def junction_arrays(junctions):
    """list of tuples -> (chrom_names_sorted, chrom_rank int32, left int32, right int32, strand int8).

    chrom_rank is the rank of the chromosome name under Python string sort, strand 0 = '+',
    1 = '-', which makes integer order on the device equal to the reference's tuple order
    (SPLICEDICE.py:96,237: chr1 < chr10 < chr2, '+' < '-').
    """
    names = sorted({j[0] for j in junctions})
    rank = {c: i for i, c in enumerate(names)}
    n = len(junctions)
    cr = np.fromiter((rank[j[0]] for j in junctions), dtype=np.int32, count=n)
    left = np.fromiter((j[1] for j in junctions), dtype=np.int64, count=n)
    right = np.fromiter((j[2] for j in junctions), dtype=np.int64, count=n)
    bad = [j for j in junctions if j[3] not in STRAND_CODE]
    if bad:
        raise ValueError(f"junction strand must be '+' or '-': {bad[0]!r}")
    if n and (left.min() < 0 or right.max() >= 2 ** 31 or (right < left).any()):
        raise ValueError("junction coordinates must satisfy 0 <= left <= right < 2**31")
    strand = np.fromiter((STRAND_CODE[j[3]] for j in junctions), dtype=np.int8, count=n)
    return names, cr, left.astype(np.int32), right.astype(np.int32), strand


def counts_to_int32(values, what):
    """float table cells -> int32; the engine sums integers (the reference's float sums of
    integer-valued counts are the same numbers)."""
    arr = np.asarray(values, dtype=np.float64)
    if arr.size and (not np.isfinite(arr).all() or (arr < 0).any() or (arr >= 2 ** 31).any()
                     or (arr != np.floor(arr)).any()):
        raise ValueError(f"{what}: counts must be non-negative integers below 2**31")
    return arr.astype(np.int32)


_DTYPE_CODE = {np.dtype(np.float32): 0, np.dtype(np.float64): 1, np.dtype(np.int32): 2}
_MODE_CODE = {".3f": 0, ".0f": 1, "repr": 2}


class NameTable:
    """Row names as ONE byte string + offsets -- the form the library's writers take them in -- behaving as a read-only
    sequence of str (names are decoded on demand).  A million names as a Python list cost ~1 s to build and ~0.3 s to
    re-encode per written table."""

    def __init__(self, blob, off):
        self.blob = bytes(blob)
        self.off = np.ascontiguousarray(off, dtype=np.int64)
        assert self.off.ndim == 1 and self.off.size >= 1 and int(self.off[-1]) - int(self.off[0]) == len(self.blob)

    def __len__(self):
        return self.off.size - 1

    def __getitem__(self, i):
        if isinstance(i, slice):
            lo, hi, step = i.indices(len(self))
            if step != 1:
                return [self[k] for k in range(lo, hi, step)]
            hi = max(hi, lo)
            base = int(self.off[0])
            return NameTable(self.blob[int(self.off[lo]) - base:int(self.off[hi]) - base], self.off[lo:hi + 1])
        n = len(self)
        i = int(i)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError(i)
        base = int(self.off[0])
        return self.blob[int(self.off[i]) - base:int(self.off[i + 1]) - base].decode()

    def __iter__(self):
        text, base = self.blob, int(self.off[0])
        o = (self.off - base).tolist()
        for k in range(len(o) - 1):
            yield text[o[k]:o[k + 1]].decode()

    def __eq__(self, other):
        return list(self) == list(other)

    def take(self, idx):
        """rows idx (an index array) as a new NameTable; the whole table in order is returned as it is"""
        idx = np.asarray(idx, dtype=np.int64)
        n = len(self)
        if idx.size == n and (n == 0 or (idx[0] == 0 and idx[-1] == n - 1 and np.all(np.diff(idx) == 1))):
            return self
        rel = self.off - self.off[0]
        lens = rel[idx + 1] - rel[idx]
        off = np.zeros(idx.size + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        src = np.repeat(rel[idx] - off[:-1], lens) + np.arange(off[-1], dtype=np.int64)
        return NameTable(np.frombuffer(self.blob, dtype=np.uint8)[src].tobytes(), off)

    def packed(self):
        """-> (blob, offsets starting at 0)"""
        return self.blob, (self.off - self.off[0] if self.off[0] else self.off)


def junction_names(chrom_names, chrom, left, right, strand):
    """NameTable of 'chrom:left-right:strand' (SPLICEDICE.py:312-314) for row-ordered junction arrays: chrom = index into
    chrom_names, strand = index into STRAND_SYM (sdice_junction_names)."""
    import ctypes as C
    from . import _ffi
    lib = _ffi.load()
    cblob, coff = _names_blob(chrom_names)
    ch, lf, rt = (np.ascontiguousarray(a, dtype=np.int32) for a in (chrom, left, right))
    sym = np.frombuffer("".join(STRAND_SYM).encode(), dtype=np.uint8)
    st = np.ascontiguousarray(sym[np.asarray(strand, dtype=np.int64)]) if ch.size else np.zeros(0, dtype=np.uint8)
    n = ch.size
    assert lf.size == rt.size == st.size == n
    longest = max((len(str(c).encode()) for c in chrom_names), default=0)
    cap = n * (longest + 26)
    out = np.empty(max(cap, 1), dtype=np.uint8)
    off = np.zeros(n + 1, dtype=np.int64)
    need = C.c_int64()
    _ffi.check(lib.sdice_junction_names(n, C.c_char_p(cblob), coff.ctypes.data_as(C.c_void_p), len(chrom_names),
                                        ch.ctypes.data_as(C.c_void_p), lf.ctypes.data_as(C.c_void_p), rt.ctypes.data_as(C.c_void_p),
                                        st.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), cap,
                                        off.ctypes.data_as(C.c_void_p), C.byref(need)), "sdice_junction_names")
    return NameTable(out[:need.value].tobytes(), off)


def trim():
    """free the text buffers the table writer keeps between calls (sdice_textio_trim)"""
    from . import _ffi
    _ffi.load().sdice_textio_trim()


def _names_blob(names):
    """names (NameTable or any sequence) -> (one byte string, int64 offsets [n + 1])"""
    if isinstance(names, NameTable):
        return names.packed()
    blobs = [str(nm).encode() for nm in names]
    off = np.zeros(len(blobs) + 1, dtype=np.int64)
    if blobs:
        np.cumsum([len(b) for b in blobs], out=off[1:])
    return b"".join(blobs), off


def write_table(path, header, names, data, mode, threads=0, append=False):
    """Write `header` + one 'name<TAB>values' line per row through the library's multithreaded
    formatter (sdice_write_table).  mode: '.3f' | '.0f' | 'repr' (numpy str of float32/float64).
    Byte-identical to the reference's per-element f-string writers."""
    import ctypes as C
    from . import _ffi
    lib = _ffi.load()
    data = np.ascontiguousarray(data)
    if data.dtype not in _DTYPE_CODE:
        raise TypeError(f"write_table: unsupported dtype {data.dtype}")
    n = len(names)
    s = data.shape[1] if data.ndim == 2 else 0
    assert data.shape[0] == n
    blob, off = _names_blob(names)
    rc = lib.sdice_write_table(str(path).encode(), header.encode(), n, s, C.c_char_p(blob), off.ctypes.data_as(C.c_void_p),
                               data.ctypes.data_as(C.c_void_p), _DTYPE_CODE[data.dtype],
                               _MODE_CODE[mode] | (0x100 if append else 0), int(threads))
    _ffi.check(rc, "sdice_write_table")


def write_columns(path, header, names, columns, modes, threads=0, suffixes=None):
    """Like write_table for column-major data: `columns` is a list of 1-D arrays (float32 / float64
    / int32, one value per row), `modes` one of '.3f' | '.0f' | 'repr' per column
    (sdice_write_columns).  `suffixes`: one ready-made string per row, written after the last numeric
    column (sdice_write_columns_sfx)."""
    import ctypes as C
    from . import _ffi
    lib = _ffi.load()
    n = len(names)
    cols = [np.ascontiguousarray(c) for c in columns]
    for c in cols:
        if c.dtype not in _DTYPE_CODE or c.shape != (n,):
            raise TypeError(f"write_columns: unsupported column {c.dtype} {c.shape}")
    blob, off = _names_blob(names)
    ptrs = (C.c_void_p * len(cols))(*[c.ctypes.data for c in cols])
    dts = np.array([_DTYPE_CODE[c.dtype] for c in cols], dtype=np.int32)
    mds = np.array([_MODE_CODE[m] for m in modes], dtype=np.int32)
    if suffixes is not None:
        assert len(suffixes) == n
        sb = [str(x).encode() for x in suffixes]
        soff = np.zeros(n + 1, dtype=np.int64)
        if n:
            np.cumsum([len(b) for b in sb], out=soff[1:])
        sblob = b"".join(sb)
        rc = lib.sdice_write_columns_sfx(str(path).encode(), header.encode(), n, C.c_char_p(blob),
                                         off.ctypes.data_as(C.c_void_p), len(cols), ptrs, dts.ctypes.data_as(C.c_void_p),
                                         mds.ctypes.data_as(C.c_void_p), C.c_char_p(sblob), soff.ctypes.data_as(C.c_void_p),
                                         int(threads))
        _ffi.check(rc, "sdice_write_columns_sfx")
        return
    rc = lib.sdice_write_columns(str(path).encode(), header.encode(), n, C.c_char_p(blob), off.ctypes.data_as(C.c_void_p),
                                 len(cols), ptrs, dts.ctypes.data_as(C.c_void_p), mds.ctypes.data_as(C.c_void_p),
                                 int(threads))
    _ffi.check(rc, "sdice_write_columns")


def interval_overlaps(ev_group, ev_a, ev_b, grp_ptr, lo, hi, threads=0):
    """-> (ptr[n_events+1], idx): for every event the intervals of its group (ev_group, -1 = none) that contain
    ev_a or ev_b, in interval order (sdice_interval_overlaps; compareSampleSets.py:246-252)."""
    import ctypes as C
    from . import _ffi
    lib = _ffi.load()
    g = np.ascontiguousarray(ev_group, dtype=np.int32)
    a = np.ascontiguousarray(ev_a, dtype=np.int64)
    b = np.ascontiguousarray(ev_b, dtype=np.int64)
    gp = np.ascontiguousarray(grp_ptr, dtype=np.int64)
    lo = np.ascontiguousarray(lo, dtype=np.int64)
    hi = np.ascontiguousarray(hi, dtype=np.int64)
    assert g.size == a.size == b.size and gp.size >= 1 and lo.size == hi.size == gp[-1]
    ptr = np.zeros(g.size + 1, dtype=np.int64)
    vp = lambda x: x.ctypes.data_as(C.c_void_p)
    args = (g.size, vp(g), vp(a), vp(b), gp.size - 1, vp(gp), vp(lo), vp(hi), vp(ptr))
    _ffi.check(lib.sdice_interval_overlaps(*args, None, 0, int(threads)), "sdice_interval_overlaps")
    idx = np.zeros(max(int(ptr[-1]), 1), dtype=np.int64)
    _ffi.check(lib.sdice_interval_overlaps(*args, vp(idx), idx.size, int(threads)), "sdice_interval_overlaps")
    return ptr, idx[: int(ptr[-1])]


def write_clusters(path, names, row_ptr, col, threads=0):
    """`_allClusters.tsv`: name<TAB>comma-joined neighbour names per row (sdice_write_clusters)."""
    import ctypes as C
    from . import _ffi
    lib = _ffi.load()
    n = len(names)
    blob, off = _names_blob(names)
    rp = np.ascontiguousarray(row_ptr, dtype=np.int64)
    cl = np.ascontiguousarray(col, dtype=np.int32)
    assert rp.size == n + 1
    rc = lib.sdice_write_clusters(str(path).encode(), n, C.c_char_p(blob), off.ctypes.data_as(C.c_void_p),
                                  rp.ctypes.data_as(C.c_void_p), cl.ctypes.data_as(C.c_void_p), int(threads))
    _ffi.check(rc, "sdice_write_clusters")


def write_junction_bed(path, chrom_names, chrom, left, right, strand, threads=0):
    """`_junctions.bed` (SPLICEDICE.py:316-321) from the row-ordered junction arrays: chrom = index into
    chrom_names, strand = index into STRAND_SYM (sdice_write_junction_bed)."""
    import ctypes as C
    from . import _ffi
    lib = _ffi.load()
    blob, off = _names_blob(chrom_names)
    ch = np.ascontiguousarray(chrom, dtype=np.int32)
    lf = np.ascontiguousarray(left, dtype=np.int32)
    rt = np.ascontiguousarray(right, dtype=np.int32)
    sym = np.frombuffer("".join(STRAND_SYM).encode(), dtype=np.uint8)
    st = np.ascontiguousarray(sym[np.asarray(strand, dtype=np.int64)]) if ch.size else np.zeros(0, dtype=np.uint8)
    assert ch.size == lf.size == rt.size == st.size
    rc = lib.sdice_write_junction_bed(str(path).encode(), ch.size, C.c_char_p(blob), off.ctypes.data_as(C.c_void_p),
                                      len(chrom_names), ch.ctypes.data_as(C.c_void_p), lf.ctypes.data_as(C.c_void_p),
                                      rt.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p), int(threads))
    _ffi.check(rc, "sdice_write_junction_bed")


def read_table_numeric(path, dtype=np.float32, threads=0, as_table=False):
    """-> (header_line, names list, data[n, s]) through the library's mmap + multithreaded parser
    (numpy semantics: text -> float64 -> dtype).  as_table: the names as a NameTable (no Python string per row)."""
    import ctypes as C
    from . import _ffi
    lib = _ffi.load()
    dtype = np.dtype(dtype)
    code = {np.dtype(np.float32): 0, np.dtype(np.float64): 1}[dtype]
    h = C.c_void_p()
    n, s, nb, hb = C.c_int64(), C.c_int32(), C.c_int64(), C.c_int64()
    _ffi.check(lib.sdice_table_open(str(path).encode(), C.byref(h), C.byref(n), C.byref(s), C.byref(nb), C.byref(hb)),
               "sdice_table_open")
    try:
        header = C.create_string_buffer(max(1, hb.value))
        names = C.create_string_buffer(max(1, nb.value))
        off = np.zeros(n.value + 1, dtype=np.int64)
        data = np.empty((n.value, s.value), dtype=dtype)
        _ffi.check(lib.sdice_table_read(h, header, names, off.ctypes.data_as(C.c_void_p), data.ctypes.data_as(C.c_void_p),
                                        code, int(threads)), "sdice_table_read")
    finally:
        lib.sdice_table_close(h)
    if as_table:
        return header.raw[:hb.value].decode(), NameTable(names.raw[:nb.value], off), data
    text = names.raw[:nb.value].decode()
    name_list = [text[off[i]:off[i + 1]] for i in range(n.value)] if text.isascii() else \
        [names.raw[off[i]:off[i + 1]].decode() for i in range(n.value)]
    return header.raw[:hb.value].decode(), name_list, data


def read_table(path, strip_lines=False):
    """'cluster<TAB>s0<TAB>s1...' table -> (header_line, names list, rows list of str lists)."""
    names, rows = [], []
    with open(path) as fh:
        header = fh.readline()
        for line in fh:
            row = (line.strip() if strip_lines else line.rstrip()).split("\t")
            names.append(row[0])
            rows.append(row[1:])
    return header, names, rows
