"""Mirror of the reference binary `centroid_fold` (src/bin/centroid_fold.rs:13-207):

    python -m rna_algos_amd.bin.centroid_fold -i FASTA -o DIR [-g GAMMA] [-c] [-t N]

bpp matrices come from the GPU (one batch), the gamma-centroid folds from
rnamc_centroid_fold; one file `centroid_threshold={gamma}.fa` per gamma (2^-7 .. 2^10
when -g is absent) holding `>{index}\\n{dot-bracket}` records joined by `\\n`."""
import argparse
import os
import sys

from ..centroid_fold import MAX_POW_2, MIN_POW_2, centroid_fold, centroid_fold_multi, get_fold_str
from ..mccaskill_algo import Context, mccaskill_algo_batch
from ..utils import FoldScoreSets, NoTablesError, read_fasta, set_default_tables
from .mccaskill_algo import fmt_f32


def write_centroid_fold(mats, recs, centroid_threshold, path):
    """src/bin/centroid_fold.rs:165-195"""
    parts = []
    for rna_id, ((_, seq), m) in enumerate(zip(recs, mats)):
        fold = centroid_fold(m, len(seq), centroid_threshold)
        parts.append(f">{rna_id}\n" + get_fold_str(fold, len(seq)))
    with open(path, "w") as fh:
        fh.write("\n".join(parts))


def main(argv=None):
    ap = argparse.ArgumentParser(prog="centroid_fold")
    ap.add_argument("-i", "--input_file_path", required=True)
    ap.add_argument("-o", "--output_dir_path", required=True)
    ap.add_argument("-g", "--centroid_threshold", type=float, default=None)
    ap.add_argument("-t", "--num_threads", type=int, default=0)
    ap.add_argument("-c", "--uses_contra_model", action="store_true")
    ap.add_argument("--synthetic-tables", type=int, default=None, metavar="SEED",
                    help="NOT the reference's parameters: seeded synthetic tables (testing only). "
                         "Without it $RNAMC_TABLES must name a table file dumped from the "
                         "rna-ss-params crate")
    args = ap.parse_args(argv)
    if args.synthetic_tables is not None:
        set_default_tables(FoldScoreSets.synthetic(args.synthetic_tables))
        print(f"warning: SYNTHETIC scoring tables (seed {args.synthetic_tables}): the output is "
              "not comparable with the reference's", file=sys.stderr)
    recs = read_fasta(args.input_file_path)
    fold_score_sets = FoldScoreSets.new(0.0)
    try:
        fold_score_sets.transfer()
    except NoTablesError as e:
        print(f"error: {e}", file=sys.stderr)
        return 2
    mats, _ = mccaskill_algo_batch([s for _, s in recs], args.uses_contra_model, False,
                                   fold_score_sets)
    os.makedirs(args.output_dir_path, exist_ok=True)
    if args.centroid_threshold is not None:
        gammas = [args.centroid_threshold]
    else:
        gammas = [2.0 ** k for k in range(MIN_POW_2, MAX_POW_2 + 1)]
    if len(gammas) == 1 and max(len(s) for _, s in recs) < 512:
        # one threshold, short records: the host fold (a launch per anti-diagonal would cost more)
        path = os.path.join(args.output_dir_path, f"centroid_threshold={fmt_f32(gammas[0])}.fa")
        write_centroid_fold(mats, recs, gammas[0], path)
        return 0
    # every threshold of a record in one device sweep (bit-identical to the host fold)
    ctx = Context(fold_score_sets)
    folds = [centroid_fold_multi(ctx, m, len(seq), gammas) for (_, seq), m in zip(recs, mats)]
    ctx.close()
    for x, g in enumerate(gammas):
        path = os.path.join(args.output_dir_path, f"centroid_threshold={fmt_f32(g)}.fa")
        parts = [f">{rna_id}\n" + get_fold_str(folds[rna_id][x], len(seq))
                 for rna_id, (_, seq) in enumerate(recs)]
        with open(path, "w") as fh:
            fh.write("\n".join(parts))
    return 0


if __name__ == "__main__":
    sys.exit(main())
