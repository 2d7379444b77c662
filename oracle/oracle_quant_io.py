"""TEST INFRASTRUCTURE (oracle): the reference's two junction-file passes of `splicedice quant`,
restated loop for loop -- getAllJunctions (SPLICEDICE.py:147-228) and getJunctionCounts
(SPLICEDICE.py:257-295).  The product parses junction files in C++ (splicedice_amd/csrc/juncio.cpp);
only tests import this module (they pin it to the reference-generated goldens, then use it to check
the C++ parser).  Nothing under splicedice_amd/ may import it.
"""
import numpy as np

STRAND_SYMBOL = {"0": "0", "1": "+", "2": "-", "+": "+", "-": "-"}   # SPLICEDICE.py:150
VALID_MOTIFS = {"gtag_only": {1, 2}, "gc_at": {1, 2, 3, 4, 5}, "all": {0, 1, 2, 3, 4, 5, 6}}   # :153
BED_LIKE = ("bed", "splicedicebed", "leafcutter")


def _sj_score(row, no_multimap):
    return int(row[6]) if no_multimap else int(row[6]) + int(row[7])


def get_all_junctions(manifest, args):
    """Union of junctions passing the per-type filters (SPLICEDICE.py:147-228)."""
    valid_motifs = VALID_MOTIFS[args.filter]
    junctions = set()
    for sample in manifest:
        with open(sample.filename, "r") as fh:
            if sample.type == "SJ":
                for line in fh:
                    row = line.rstrip().split("\t")
                    left, right = int(row[1]) - 1, int(row[2])
                    strand = STRAND_SYMBOL[row[3]]
                    length = right - left
                    if (args.minLength < length < args.maxLength and strand != "0"
                            and _sj_score(row, args.noMultimap) >= args.minUnique
                            and int(row[4]) in valid_motifs):
                        junctions.add((row[0], left, right, strand))
            elif sample.type == "splicedicebed":
                for line in fh:
                    row = line.rstrip().split("\t")
                    score = int(row[4])
                    info = [x.split(":") for x in row[3].split(";")]
                    left, right = int(row[1]), int(row[2])
                    length = right - left
                    if info[3][1] == "?":      # un-annotated junctions must earn their place (:194-203)
                        if score < args.minUnique:
                            continue
                        if length > args.maxLength or length < args.minLength:
                            continue
                        if int(info[1][1]) < args.minOverhang:
                            continue
                        if float(info[0][1]) < args.minEntropy or float(info[0][2]) < args.minEntropy:
                            continue
                    if row[5] in ("+", "-"):
                        junctions.add((row[0], left, right, row[5]))
            elif sample.type in ("bed", "leafcutter"):
                for line in fh:
                    row = line.rstrip().split("\t")
                    if int(row[4]) < args.minUnique:
                        continue
                    left, right = int(row[1]), int(row[2])
                    length = right - left
                    if length > args.maxLength or length < args.minLength:
                        continue
                    if row[5] in ("+", "-"):
                        junctions.add((row[0], left, right, row[5]))
    return junctions


def get_junction_counts(manifest, index, args):
    """counts int32 [N, S] + `low` flat indices (SPLICEDICE.py:257-295).  No score filter
    here; a later line for the same junction overwrites an earlier one."""
    n, s = len(index), len(manifest)
    counts = np.zeros((n, s), dtype=np.int64)
    low = []
    for si, sample in enumerate(manifest):
        with open(sample.filename, "r") as fh:
            if sample.type in BED_LIKE:
                for line in fh:
                    row = line.rstrip().split("\t")
                    r = index.get((row[0], int(row[1]), int(row[2]), row[5]))
                    if r is not None:
                        score = int(row[4])
                        counts[r, si] = score
                        if args.lowCoverageNan and score < args.minUnique:
                            low.append(r * s + si)
            elif sample.type == "SJ":
                for line in fh:
                    row = line.rstrip().split("\t")
                    r = index.get((row[0], int(row[1]) - 1, int(row[2]), {"0": "0", "1": "+", "2": "-"}[row[3]]))
                    if r is not None:
                        counts[r, si] = _sj_score(row, args.noMultimap)
    if counts.size and (counts.min() < 0 or counts.max() >= 2 ** 31):
        raise ValueError("junction counts must be non-negative and below 2**31")
    return counts.astype(np.int32), np.asarray(low, dtype=np.int64)
