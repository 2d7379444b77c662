"""`splicedice quant`: manifest of junction files -> clusters, inclusion counts, PS table.

Drop-in for the reference sub-command (splicedice/SPLICEDICE.py: add_parser :377-402,
run_with :404-409): same flags, same stdout banners, byte-identical output files
`<prefix>_junctions.bed`, `_allClusters.tsv`, `_inclusionCounts.tsv`, `_allPS.tsv`
(+ `_drimTable.tsv` with --drim).

What moved to the GPU (through the C ABI, include/sdice.h):
  getClusters + junctionIndex sort (SPLICEDICE.py:230-255, :96)  -> sdice_cluster
  calculatePsi (SPLICEDICE.py:297-310)                           -> sdice_ps (+ sdice_mark_low)
Text parsing and formatting stay on the host, as in the reference.
"""
from time import time

import numpy as np

from . import _stages, juncio, textio
from .engine import Context


class Sample:
    """One manifest row (SPLICEDICE.py:11-46): name, path, sniffed file type, metadata, condition."""

    def __init__(self, row):
        self.name = row[0]
        self.filename = row[1]
        upper = self.filename.upper()
        if upper.endswith(".BED"):
            self.type = "bed"
            with open(self.filename) as bedfile:
                info = bedfile.readline().split("\t")[3].split(";")
                if info[0].startswith("e:") and info[1].startswith("o:"):
                    self.type = "splicedicebed"
        elif upper.endswith("SJ.OUT.TAB"):
            self.type = "SJ"
        elif upper.endswith(".BAM"):
            self.type = "bam"
        elif upper.endswith("LEAFCUTTER.JUNC"):
            self.type = "leafcutter"
        else:
            self.type = "unknown"
        self.metadata = row[2]
        self.condition = row[3]


class Timer:
    """Stage stopwatch printing [h:mm:ss.ss] like SPLICEDICE.py:48-66."""

    def __init__(self):
        self.start = time()
        self.checkpoint = self.start

    @staticmethod
    def _fmt(passed):
        hours = int(passed // 3600)
        minutes = int((passed % 3600) // 60)
        seconds = passed % 60
        return f"[{hours}:{minutes:02d}:{seconds:02.2f}]"

    def total(self):
        return self._fmt(time() - self.start)

    def check(self):
        now = time()
        passed = now - self.checkpoint
        self.checkpoint = now
        return self._fmt(passed)


def parse_manifest(path):
    manifest = []
    with open(path, "r") as fh:
        for line in fh:
            manifest.append(Sample(line.rstrip().split("\t")))
    return manifest


class Quant:
    """The quant pipeline; attribute names follow the reference class (SPLICEDICE.py:71-131)."""

    def __init__(self, manifest_filename, output_prefix, args, ctx=None):
        self.args = args
        self.manifestFilename = manifest_filename
        self.outputPrefix = output_prefix
        own_ctx = ctx is None
        from . import mgpu
        self.L = mgpu.launcher()            # (reads the torchrun environment before any GPU call)
        self.ctx = ctx if ctx is not None else Context(self.L.local_rank)
        try:
            self._run()
        finally:
            if own_ctx:
                self.ctx.close()

    def _run(self):
        timer = Timer()
        print("Parsing manifest...")
        self.manifest = parse_manifest(self.manifestFilename)
        print("\tDone", timer.check())

        print(f"Getting all junctions from {len(self.manifest)} files...")
        # one multithreaded pass per file (csrc/juncio.cpp) instead of the reference's two Python
        # passes; get_all_junctions / get_junction_counts below state the same rules in Python
        with _stages.stage("parse"):
            chrom_names, junc, parsed = juncio.ingest(self.manifest, self.args, self.ctx)
        print("\tDone", timer.check())

        print(f"Finding clusters from {junc[0].size} junctions...")
        with _stages.stage("cluster"):
            row_of, self.row_ptr, self.col = self.ctx.cluster(*junc)
        rows = [np.empty_like(a) for a in junc]
        for dst, src in zip(rows, junc):
            dst[row_of] = src                                  # row order (SPLICEDICE.py:96)
        self.rows = tuple(rows)
        self.chrom_names = chrom_names
        # 'chrom:left-right:strand' per row (SPLICEDICE.py:312-314), kept as one byte string + offsets for the writers
        self.names = textio.junction_names(chrom_names, *self.rows)
        print("\tDone", timer.check())

        # Several ranks (python -m torch.distributed.run ... -m splicedice_amd quant): parsing and clustering
        # are repeated on every rank (host-bound, deterministic); the junction rows are then cut by
        # shard.shard_plan and every rank computes and FORMATS its own rows of the two big tables;
        # rank 0 writes the small files and stitches the parts (mgpu.py).
        L = self.L
        n = len(self.names)
        if L.world > 1:
            from . import shard
            self.part = shard.shard_plan(self.row_ptr, self.col, L.world)[L.rank]
        else:
            self.part = dict(own_lo=0, own_hi=n, ext_lo=0, ext_hi=n)

        print("Writing cluster file...")
        if L.root:
            with _stages.stage("format+write"):
                self.write_clusters()
        print("\tDone", timer.check())

        print("Writing junction bed file...")
        if L.root:
            with _stages.stage("format+write"):
                self.write_junction_bed()
        print("\tDone", timer.check())

        print("Gathering junction counts...")
        with _stages.stage("parse"):
            self.counts, self.low = juncio.gather_counts(self.manifest, parsed, self.rows, self.args)
        print("\tDone", timer.check())

        print("Writing inclusion counts...")
        with _stages.stage("format+write"):
            self.write_inclusions()
        print("\tDone", timer.check())

        print("Calculating PS values...")
        lo, hi, elo, ehi = (self.part[k] for k in ("own_lo", "own_hi", "ext_lo", "ext_hi"))
        if L.world > 1:
            from . import shard
            rp, cl = shard.local_csr(self.row_ptr, self.col, self.part)
            s = self.counts.shape[1]
            self.psi = self.ctx.ps(np.ascontiguousarray(self.counts[elo:ehi]), rp, cl)[lo - elo: hi - elo] \
                if hi > lo else np.zeros((0, s), np.float32)
            low = self.low[(self.low // s >= lo) & (self.low // s < hi)] - lo * s if self.low.size else self.low
        elif hasattr(self.ctx, "ps_dev"):
            # the HIP engine, stage by stage (the host entry point sdice_ps does the same three steps in one call)
            ctx = self.ctx
            with _stages.stage("h2d"):
                d_counts = ctx.to_device(self.counts, np.int32)
                d_rp = ctx.to_device(self.row_ptr, np.int64)
                d_col = ctx.to_device(self.col if self.col.size else np.zeros(1, np.int32), np.int32)
                d_ps = ctx.empty(self.counts.shape, np.float32)
            with _stages.stage("kernels"):
                if self.counts.size:
                    ctx.ps_dev(d_counts, d_rp, d_col, None, d_ps)
                ctx.sync()
            with _stages.stage("d2h"):
                self.psi = d_ps.to_host()
            for a in (d_counts, d_rp, d_col, d_ps):
                a.free()
            low = self.low
        else:
            self.psi = self.ctx.ps(self.counts, self.row_ptr, self.col)
            low = self.low
        if self.args.lowCoverageNan and low.size:
            self.psi = self.ctx.mark_low(self.psi, low)
        print("\tDone", timer.check())

        print("Writing PS values...")
        with _stages.stage("format+write"):
            self.write_all_psi()
        print("\tDone", timer.check())

        if self.args.drim:
            print("Writing drim table...")
            if L.root:
                self.write_drim_table()
            print("\tDone", timer.check())

        print("All done", timer.total())

    # ------------------------------------------------------------------ writers (SPLICEDICE.py:316-370)
    def write_junction_bed(self):
        textio.write_junction_bed(f"{self.outputPrefix}_junctions.bed", self.chrom_names, *self.rows)

    def write_clusters(self):
        textio.write_clusters(f"{self.outputPrefix}_allClusters.tsv", self.names, self.row_ptr, self.col)

    def _sample_header(self, first):
        return first + "\t" + "\t".join(s.name for s in self.manifest) + "\n"

    def _write_rows(self, path, data_own, mode):
        """this rank's rows [own_lo, own_hi) of one table; with several ranks: part files, stitched by rank 0"""
        L, lo, hi = self.L, self.part["own_lo"], self.part["own_hi"]
        header = self._sample_header("cluster")
        if L.world == 1:
            textio.write_table(path, header, self.names, data_own, mode)
            return
        textio.write_table(L.part(path), header if L.root else "", self.names[lo:hi], data_own, mode)
        L.stitch(path)

    def write_inclusions(self):
        # f'{x:.0f}' per cell (SPLICEDICE.py:340) through the library's multithreaded formatter
        lo, hi = self.part["own_lo"], self.part["own_hi"]
        self._write_rows(f"{self.outputPrefix}_inclusionCounts.tsv", self.counts if self.L.world == 1 else self.counts[lo:hi],
                         ".0f")

    def write_all_psi(self):
        # f'{x:.3f}' per cell (SPLICEDICE.py:353)
        self._write_rows(f"{self.outputPrefix}_allPS.tsv", self.psi, ".3f")

    def write_drim_table(self):
        counts_str = self.counts.astype(np.float32).astype("str")
        with open(f"{self.outputPrefix}_drimTable.tsv", "w") as out:
            out.write("gene\tfeature_id\t" + "\t".join(s.name for s in self.manifest) + "\n")
            rp, col = self.row_ptr, self.col
            for r, name in enumerate(self.names):
                for other in [r] + list(col[rp[r]:rp[r + 1]]):
                    print(f"cl_{r}_{name}", f"{self.names[other]}_{r}", "\t".join(counts_str[other]), sep="\t", file=out)


def add_parser(parser):
    """Flag surface of SPLICEDICE.py:377-402."""
    parser.add_argument("--manifest", "-m", action="store", required=True,
                        help="tab-separated list of samples with file paths")
    parser.add_argument("--output_prefix", "-o", action="store", required=True,
                        help="prefix for output filenames")
    parser.add_argument("--maxLength", type=int, default=50000, help="maximum splice junction size")
    parser.add_argument("--minLength", type=int, default=50, help="minimum splice junction size")
    parser.add_argument("--minOverhang", type=int, default=5,
                        help="minimum overlap on reads to support splice junction")
    parser.add_argument("--drim", action="store_true", help="create table for use by DRIMSeq")
    parser.add_argument("--noMultimap", action="store_true",
                        help="use only reads that uniquely map to one location")
    parser.add_argument("--filter", default="gtag_only", choices=["gtag_only"],
                        help="donor and acceptor intron sequences to include.")
    parser.add_argument("--minUnique", type=int, default=5,
                        help="minimum number of unique reads to support splice junction")
    parser.add_argument("--lowCoverageNan", action="store_true",
                        help="Report NaN for splicing events with coverage below minUnique")
    parser.add_argument("--minEntropy", type=float, default=1,
                        help="Shannon's diversity index associated with a junction, minumum required for "
                             "inclusion [Default 1]")


def run_with(args):
    Quant(args.manifest, args.output_prefix, args)


if __name__ == "__main__":
    import argparse
    p = argparse.ArgumentParser(description="Percent-Spliced quantification (MI355X engine).")
    add_parser(p)
    run_with(p.parse_args())
