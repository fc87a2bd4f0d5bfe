"""Mirror of the reference binary `durbin_algo` (src/bin/durbin_algo.rs:6-92):

    python -m rna_algos_amd.bin.durbin_algo -i FASTA -o OUT [-t N]

Same output text: a `# Format = ...` header, then per pair of records `\\n\\n>{id1},{id2}\\n`
followed by `i,j,p ` triples of the match probabilities p > 0 (positions without the pseudo
bases).  The reference iterates a hash map of pairs (order unspecified); here pairs come in
ascending (id1, id2).  All pairs go to the GPU as one batch; `-t` is accepted and ignored."""
import argparse
import sys

import numpy as np

from ..durbin_algo import AlignScores, durbin_algo_batch, with_pseudo_bases
from ..utils import read_fasta
from .mccaskill_algo import fmt_f32

HEADER = ("# Format = >{RNA sequence id 1},{RNA sequence id 2} {line break} {nucleotide 1}, "
          "{nucleotide 2}, {nucletide matching probability} ...")


def main(argv=None):
    ap = argparse.ArgumentParser(prog="durbin_algo")
    ap.add_argument("-i", "--input_file_path", required=True)
    ap.add_argument("-o", "--output_file_path", required=True)
    ap.add_argument("-t", "--num_threads", type=int, default=0)
    args = ap.parse_args(argv)
    seqs = [with_pseudo_bases(s) for _, s in read_fasta(args.input_file_path)]
    align_scores = AlignScores.new(0.0)
    align_scores.transfer()
    pairs = [(a, b) for a in range(len(seqs)) for b in range(a + 1, len(seqs))]
    mats = durbin_algo_batch(seqs, pairs, align_scores)
    buf = [HEADER]
    for (a, b), m in zip(pairs, mats):
        buf.append(f"\n\n>{a},{b}\n")
        ii, jj = np.nonzero(m > 0.0)
        buf.append("".join(f"{i - 1},{j - 1},{fmt_f32(m[i, j])} " for i, j in zip(ii, jj)))
    with open(args.output_file_path, "w") as fh:
        fh.write("".join(buf))
    return 0


if __name__ == "__main__":
    sys.exit(main())
