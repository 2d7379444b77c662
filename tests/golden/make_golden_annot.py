#!/usr/bin/env python3
"""A second GTF for the annotated compare_sample_sets table whose exons DO border tested events, by RUNNING
THE REFERENCE (build container only; same statsmodels shim as make_golden.py).  The first fixture
(compare/anno.gtf) only exercises the gene-interval scan; this one also exercises the known-junction join:
one junction shared by two transcripts of one gene, one shared by two genes, genes on the other strand,
two genes on identical coordinates, an event whose chromosome+strand has no gene at all.

    python tests/golden/make_golden_annot.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (puts /root/reference on sys.path)


def main():
    MG.install_statsmodels_shim()
    import splicedice.compareSampleSets as CSS
    d = os.path.join(HERE, "compare")
    gtf = os.path.join(d, "anno_hits.gtf")

    def rec(kind, a, b, strand, gene, tid=None, chrom="chr1"):
        t = f' transcript_id "{tid}";' if tid else ""
        return f'{chrom}\tt\t{kind}\t{a}\t{b}\t.\t{strand}\t.\tgene_id "ID_{gene}";{t} gene_name "{gene}";\n'

    with open(gtf, "w") as fh:
        fh.write("# events are chr1:(1000+10r)-(2000+10r):+ ; an exon ending at 1000+10r and one starting at 2001+10r border event r\n")
        fh.write(rec("gene", 900, 2600, "+", "GENEA"))
        fh.write(rec("transcript", 900, 2600, "+", "GENEA", "TA1"))
        fh.write(rec("exon", 900, 1000, "+", "GENEA", "TA1"))          # event 0
        fh.write(rec("exon", 2001, 2100, "+", "GENEA", "TA1"))
        fh.write(rec("transcript", 900, 2600, "+", "GENEA", "TA2"))     # same junction, second transcript, same gene
        fh.write(rec("exon", 950, 1000, "+", "GENEA", "TA2"))
        fh.write(rec("exon", 2001, 2600, "+", "GENEA", "TA2"))
        fh.write(rec("gene", 1000, 2300, "+", "GENEC"))                 # a second gene on the same junction as event 3
        fh.write(rec("transcript", 1000, 2300, "+", "GENEC", "TC1"))
        fh.write(rec("exon", 1005, 1030, "+", "GENEC", "TC1"))
        fh.write(rec("exon", 2031, 2300, "+", "GENEC", "TC1"))
        fh.write(rec("gene", 900, 2600, "+", "GENED"))                  # identical coordinates to GENEA
        fh.write(rec("transcript", 900, 2600, "+", "GENED", "TD1"))
        fh.write(rec("exon", 1001, 1030, "+", "GENED", "TD1"))          # event 3 again: two genes, one junction
        fh.write(rec("exon", 2031, 2040, "+", "GENED", "TD1"))
        fh.write(rec("exon", 2100, 2600, "+", "GENED", "TD1"))          # and a junction no event has
        fh.write(rec("gene", 1000, 4000, "-", "GENEM"))                 # other strand: never listed
        fh.write(rec("transcript", 1000, 4000, "-", "GENEM", "TM1"))
        fh.write(rec("exon", 1000, 1010, "-", "GENEM", "TM1"))
        fh.write(rec("exon", 2011, 2500, "-", "GENEM", "TM1"))
        fh.write(rec("gene", 3000, 4200, "+", "GENEB"))
        fh.write(rec("gene", 100, 200, "+", "GENEZ", chrom="chr9"))
    MG.quiet(CSS.run_with, MG.ns(psiSPLICEDICE=os.path.join(d, "in_allPS.tsv"), manifest1=os.path.join(d, "m1.tsv"),
                                  manifest2=os.path.join(d, "m2.tsv"), annotation=gtf,
                                  outputFile=os.path.join(d, "expected_out_gtf_hits.tsv")))
    print(open(os.path.join(d, "expected_out_gtf_hits.tsv")).read()[:600])


if __name__ == "__main__":
    main()
