"""`splicedice` console entry: same sub-command names as the reference dispatcher
(splicedice/__main__.py:17-61).  Modules are imported lazily, so the four accelerated
commands do not need pysam/statsmodels just to start.

Sub-commands outside the accelerated path (bam_to_junc_bed, intron_coverage,
subset, select) are registered by name so that scripts get
a clear message instead of an argparse "invalid choice".
"""
import argparse
import importlib
import sys

ACCELERATED = {
    "quant": "splicedice_amd.quant",
    "counts_to_ps": "splicedice_amd.counts_to_ps",
    "compare_sample_sets": "splicedice_amd.compare_sample_sets",
    "pairwise": "splicedice_amd.pairwise",
    "similarity": "splicedice_amd.similarity",
    "findOutliers": "splicedice_amd.find_outliers",
    "ir_table": "splicedice_amd.ir_table",
}
NOT_BUILT = ["bam_to_junc_bed", "intron_coverage", "subset", "select"]


def _not_built(name):
    def run(_args):
        print(f"splicedice {name}: not part of the MI355X engine (hot path only: "
              f"{', '.join(ACCELERATED)}); use the reference implementation for this step.", file=sys.stderr)
        sys.exit(2)
    return run


def build_parser():
    parser = argparse.ArgumentParser(prog="splicedice", description="splicedice on MI355X (gfx950)")
    subparsers = parser.add_subparsers(title="subcommands", dest="command")
    for name, modname in ACCELERATED.items():
        sub = subparsers.add_parser(name)
        module = importlib.import_module(modname)
        module.add_parser(sub)
        sub.set_defaults(main=module.run_with)
    for name in NOT_BUILT:
        sub = subparsers.add_parser(name, add_help=False)
        sub.add_argument("rest", nargs=argparse.REMAINDER)
        sub.set_defaults(main=_not_built(name))
    return parser


def main(argv=None):
    parser = build_parser()
    args = parser.parse_args(argv)
    if not hasattr(args, "main"):
        parser.print_usage()      # no sub-command: splicedice/__main__.py:58-61
        return
    args.main(args)


if __name__ == "__main__":
    main()
